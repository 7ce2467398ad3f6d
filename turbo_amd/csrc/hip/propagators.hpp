// Device-side interval propagators for the ternary constraint network `x = y op z`.
//
// Replaces the bodies of `PIR::deduce(i)` / `PIR::ask(i)` that the reference calls at
// gpu_dive_and_solve.hpp:302,312,334 and barebones_dive_and_solve.hpp:931,944,977 (the bodies
// themselves live in lattice-land/lala-pc v1.2.8, which is not part of the reference tree).
// Semantics are documented in DESIGN.md ("Propagator rules") and must stay bit-identical to
// oracle/oracle.c -- the parity tests compare fixpoints, which are order independent because every
// rule below is monotone and contracting.
//
// CDNA4 notes: integer VALU only (no MFMA: nothing here is a contraction).  A propagator is evaluated
// in two phases so that a wave executing mixed operators diverges only in the cheap "compute the
// candidate bounds" switch and reconverges for the memory phase (LDS atomics / change detection).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tb {

constexpr int NINF = INT32_MIN;
constexpr int PINF = INT32_MAX;

enum Op : int { OP_ADD = 0, OP_MUL = 1, OP_TDIV = 2, OP_TMOD = 3, OP_MIN = 4, OP_MAX = 5, OP_EQ = 6, OP_LEQ = 7 };

struct Itv { int lb, ub; };

// Candidate bounds produced by one propagator: dom(v) is intersected with [l,u].
struct Cand {
  int xl = NINF, xu = PINF, yl = NINF, yu = PINF, zl = NINF, zu = PINF;
  bool ent = false;  // ask(): entailed by the loaded box
};

__device__ __forceinline__ bool is_inf(int a) { return a == NINF || a == PINF; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int sat_add(int a, int b) { return __builtin_elementwise_add_sat(a, b); }  // v_add_i32 clamp
__device__ __forceinline__ int neg_ext(int a) { return a == NINF ? PINF : (a == PINF ? NINF : -a); }
__device__ __forceinline__ int clamp64(long long v) { return v >= (long long)PINF ? PINF : (v <= (long long)NINF ? NINF : (int)v); }

// lower / upper bound of a sum from the two lower / upper bounds (+-inf absorbing, saturating)
__device__ __forceinline__ int add_lo(int a, int b) {
  int r = sat_add(a, b);
  r = (a == PINF || b == PINF) ? PINF : r;
  return (a == NINF || b == NINF) ? NINF : r;
}
__device__ __forceinline__ int add_hi(int a, int b) {
  int r = sat_add(a, b);
  r = (a == NINF || b == NINF) ? NINF : r;
  return (a == PINF || b == PINF) ? PINF : r;
}
__device__ __forceinline__ int mul_ext(int a, int b) {
  if (a == 0 || b == 0) return 0;
  if (is_inf(a) || is_inf(b)) return ((a < 0) != (b < 0)) ? NINF : PINF;
  return clamp64((long long)a * (long long)b);
}
// floor / ceil division of finite 32-bit operands (b != 0; INT32_MIN never occurs: it is -inf)
__device__ __forceinline__ int div_floor(int a, int b) {
  int q = a / b, r = a - q * b;
  return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}
__device__ __forceinline__ int div_ceil(int a, int b) {
  int q = a / b, r = a - q * b;
  return (r != 0 && ((r < 0) == (b < 0))) ? q + 1 : q;
}
__device__ __forceinline__ long long lmin(long long a, long long b) { return a < b ? a : b; }
__device__ __forceinline__ long long lmax(long long a, long long b) { return a > b ? a : b; }
__device__ __forceinline__ long long labs64(long long a) { return a < 0 ? -a : a; }

__device__ __forceinline__ void meet(int& l, int& u, int nl, int nu) { l = imax(l, nl); u = imin(u, nu); }

// Phase 1: candidate bounds + entailment, from the loaded domains only (no memory access).
__device__ __forceinline__ Cand evaluate(int op, const Itv X, const Itv Y, const Itv Z) {
  Cand c;
  switch (op) {
    case OP_ADD: {
      c.xl = add_lo(Y.lb, Z.lb); c.xu = add_hi(Y.ub, Z.ub);
      c.yl = add_lo(X.lb, neg_ext(Z.ub)); c.yu = add_hi(X.ub, neg_ext(Z.lb));
      c.zl = add_lo(X.lb, neg_ext(Y.ub)); c.zu = add_hi(X.ub, neg_ext(Y.lb));
      break;
    }
    case OP_MUL: {
      int c0 = mul_ext(Y.lb, Z.lb), c1 = mul_ext(Y.lb, Z.ub), c2 = mul_ext(Y.ub, Z.lb), c3 = mul_ext(Y.ub, Z.ub);
      c.xl = imin(imin(c0, c1), imin(c2, c3)); c.xu = imax(imax(c0, c1), imax(c2, c3));
      if (X.lb > 0 || X.ub < 0) {  // a non-zero product has non-zero factors
        if (Y.lb == 0) c.yl = 1;
        if (Y.ub == 0) c.yu = -1;
        if (Z.lb == 0) c.zl = 1;
        if (Z.ub == 0) c.zu = -1;
      }
      bool x_fin = !is_inf(X.lb) && !is_inf(X.ub);
      if (x_fin && (Z.lb > 0 || Z.ub < 0) && !is_inf(Z.lb) && !is_inf(Z.ub)) {
        int lo = imin(imin(div_ceil(X.lb, Z.lb), div_ceil(X.lb, Z.ub)), imin(div_ceil(X.ub, Z.lb), div_ceil(X.ub, Z.ub)));
        int hi = imax(imax(div_floor(X.lb, Z.lb), div_floor(X.lb, Z.ub)), imax(div_floor(X.ub, Z.lb), div_floor(X.ub, Z.ub)));
        meet(c.yl, c.yu, lo, hi);
      }
      if (x_fin && (Y.lb > 0 || Y.ub < 0) && !is_inf(Y.lb) && !is_inf(Y.ub)) {
        int lo = imin(imin(div_ceil(X.lb, Y.lb), div_ceil(X.lb, Y.ub)), imin(div_ceil(X.ub, Y.lb), div_ceil(X.ub, Y.ub)));
        int hi = imax(imax(div_floor(X.lb, Y.lb), div_floor(X.lb, Y.ub)), imax(div_floor(X.ub, Y.lb), div_floor(X.ub, Y.ub)));
        meet(c.zl, c.zu, lo, hi);
      }
      break;
    }
    case OP_TDIV:
    case OP_TMOD: {
      int zl = Z.lb, zu = Z.ub;  // the divisor is never 0
      if (zl == 0) { zl = 1; c.zl = 1; }
      if (zu == 0) { zu = -1; c.zu = -1; }
      if (zl > zu) break;
      bool z_fin = !is_inf(zl) && !is_inf(zu);
      bool y_fin = !is_inf(Y.lb) && !is_inf(Y.ub);
      bool z_nz = (zl > 0 || zu < 0);
      if (op == OP_TDIV) {
        if (z_nz && z_fin && y_fin) {
          int q0 = Y.lb / zl, q1 = Y.lb / zu, q2 = Y.ub / zl, q3 = Y.ub / zu;
          c.xl = imin(imin(q0, q1), imin(q2, q3)); c.xu = imax(imax(q0, q1), imax(q2, q3));
        } else if (y_fin) {
          long long m = lmax(labs64(Y.lb), labs64(Y.ub));
          c.xl = clamp64(-m); c.xu = clamp64(m);
        }
        if (!is_inf(X.lb) && !is_inf(X.ub) && z_fin) {
          long long p0 = (long long)X.lb * zl, p1 = (long long)X.lb * zu, p2 = (long long)X.ub * zl, p3 = (long long)X.ub * zu;
          long long m = lmax(labs64(zl), labs64(zu)) - 1;
          c.yl = clamp64(lmin(lmin(p0, p1), lmin(p2, p3)) - m); c.yu = clamp64(lmax(lmax(p0, p1), lmax(p2, p3)) + m);
        }
      } else {
        int m = z_fin ? clamp64(lmax(labs64(zl), labs64(zu)) - 1) : PINF;
        if (Y.lb >= 0) { c.xl = 0; c.xu = imin(m, Y.ub); }
        else if (Y.ub <= 0) { c.xl = imax(neg_ext(m), Y.lb); c.xu = 0; }
        else { c.xl = neg_ext(m); c.xu = m; }
        if (y_fin && Y.lb == Y.ub && z_fin && zl == zu) { int r = Y.lb % zl; meet(c.xl, c.xu, r, r); }
      }
      break;
    }
    case OP_MIN: {
      c.xl = imin(Y.lb, Z.lb); c.xu = imin(Y.ub, Z.ub);
      c.yl = X.lb; c.zl = X.lb;
      if (Y.lb > X.ub) c.zu = X.ub;
      if (Z.lb > X.ub) c.yu = X.ub;
      break;
    }
    case OP_MAX: {
      c.xl = imax(Y.lb, Z.lb); c.xu = imax(Y.ub, Z.ub);
      c.yu = X.ub; c.zu = X.ub;
      if (Y.ub < X.lb) c.zl = X.lb;
      if (Z.ub < X.lb) c.yl = X.lb;
      break;
    }
    case OP_EQ: {
      bool disjoint = Y.ub < Z.lb || Y.lb > Z.ub;
      bool same = Y.lb == Y.ub && Z.lb == Z.ub && Y.lb == Z.lb;
      if (X.lb >= 1) {
        c.yl = Z.lb; c.yu = Z.ub; c.zl = Y.lb; c.zu = Y.ub;
        c.ent = same;
      } else if (X.ub <= 0) {
        if (Y.lb == Y.ub) {
          if (Z.lb == Y.lb) c.zl = add_lo(Y.lb, 1);
          if (Z.ub == Y.lb) c.zu = add_hi(Y.lb, -1);
        }
        if (Z.lb == Z.ub) {
          if (Y.lb == Z.lb) c.yl = add_lo(Z.lb, 1);
          if (Y.ub == Z.lb) c.yu = add_hi(Z.lb, -1);
        }
        c.ent = disjoint;
      } else {
        if (disjoint) c.xu = 0;
        else if (same) c.xl = 1;
      }
      return c;
    }
    case OP_LEQ: {
      if (X.lb >= 1) {
        c.yu = Z.ub; c.zl = Y.lb;
        c.ent = Y.ub <= Z.lb;
      } else if (X.ub <= 0) {
        c.yl = add_lo(Z.lb, 1); c.zu = add_hi(Y.ub, -1);
        c.ent = Y.lb > Z.ub;
      } else {
        if (Y.ub <= Z.lb) c.xl = 1;
        else if (Y.lb > Z.ub) c.xu = 0;
      }
      return c;
    }
    default: break;
  }
  // arithmetic operators: entailed iff all three are assigned and the relation holds
  if (X.lb == X.ub && Y.lb == Y.ub && Z.lb == Z.ub && !is_inf(X.lb) && !is_inf(Y.lb) && !is_inf(Z.lb)) {
    long long x = X.lb, y = Y.lb, z = Z.lb;
    switch (op) {
      case OP_ADD: c.ent = (x == y + z); break;
      case OP_MUL: c.ent = (x == y * z); break;
      case OP_TDIV: c.ent = (z != 0 && x == (long long)(Y.lb / (Z.lb == 0 ? 1 : Z.lb))); break;
      case OP_TMOD: c.ent = (z != 0 && x == (long long)(Y.lb % (Z.lb == 0 ? 1 : Z.lb))); break;
      case OP_MIN: c.ent = (x == (y < z ? y : z)); break;
      case OP_MAX: c.ent = (x == (y > z ? y : z)); break;
      default: break;
    }
  }
  return c;
}

}  // namespace tb
