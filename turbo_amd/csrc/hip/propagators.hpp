// Device-side interval propagators for the ternary constraint network `x = y op z`.
//
// Replaces the bodies of `PIR::deduce(i)` / `PIR::ask(i)` that the reference calls at
// gpu_dive_and_solve.hpp:302,312,334 and barebones_dive_and_solve.hpp:931,944,977 (the bodies
// themselves live in lattice-land/lala-pc v1.2.8, which is not part of the reference tree).
// Semantics are documented in DESIGN.md ("Propagator rules") and must stay bit-identical to
// oracle/oracle.c -- the parity tests compare fixpoints, which are order independent because every
// rule below is monotone and contracting.
//
// CDNA4 notes: integer VALU only (no MFMA: nothing here is a contraction).  The sweep is VALU-issue
// bound (first profile: 134 VALU + 95 SALU per wave-propagation, LDS 1 % busy), so the rules are written
// to minimise VALU instructions:
//   * the shim rewrites each bytecode into a packed record whose word0 carries a pack-time CLASS and the
//     set of classes present in its 64-record slice; the slice set is read into an SGPR, so every class
//     body sits behind a SCALAR branch -- a wave pays only for the classes it contains and there is no
//     exec-mask manipulation; inside a body all lanes compute and the lanes of that class keep the result;
//   * predicates are combined with logical operators on already computed values (they become s_and/s_or on
//     lane masks, not VALU), values with v_cndmask, sums with v_add_i32/v_sub_i32 clamp (saturating).
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
__device__ __forceinline__ int sel(bool c, int a, int b) { return c ? a : b; }
__device__ __forceinline__ int sat_add(int a, int b) { return __builtin_elementwise_add_sat(a, b); }  // v_add_i32 clamp
__device__ __forceinline__ int sat_sub(int a, int b) { return __builtin_elementwise_sub_sat(a, b); }  // v_sub_i32 clamp
__device__ __forceinline__ int neg_ext(int a) { return a == NINF ? PINF : (a == PINF ? NINF : -a); }
__device__ __forceinline__ int clamp64(long long v) { return v >= (long long)PINF ? PINF : (v <= (long long)NINF ? NINF : (int)v); }

// Bounds of sums / differences: an infinite bound on the side that matters absorbs, the rest saturates.
__device__ __forceinline__ int add_lo(int a, int b) { return sel(a == NINF || b == NINF, NINF, sat_add(a, b)); }  // lb(A+B)
__device__ __forceinline__ int add_hi(int a, int b) { return sel(a == PINF || b == PINF, PINF, sat_add(a, b)); }  // ub(A+B)
__device__ __forceinline__ int sub_lo(int a, int b) { return sel(a == NINF || b == PINF, NINF, sat_sub(a, b)); }  // lb(A-B) from lb(A), ub(B)
__device__ __forceinline__ int sub_hi(int a, int b) { return sel(a == PINF || b == NINF, PINF, sat_sub(a, b)); }  // ub(A-B) from ub(A), lb(B)

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

// The three operators with divisions (rare; divergent code is acceptable here).
__device__ __forceinline__ Cand evaluate_heavy(int op, const Itv X, const Itv Y, const Itv Z) {
  Cand c;
  if (op == OP_MUL) {
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
  } else {  // OP_TDIV, OP_TMOD
    int zl = Z.lb, zu = Z.ub;  // the divisor is never 0
    if (zl == 0) { zl = 1; c.zl = 1; }
    if (zu == 0) { zu = -1; c.zu = -1; }
    if (zl <= zu) {
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
    }
  }
  // entailed iff all three are assigned (finite) and the relation holds
  if (X.lb == X.ub && Y.lb == Y.ub && Z.lb == Z.ub && !is_inf(X.lb) && !is_inf(Y.lb) && !is_inf(Z.lb)) {
    long long x = X.lb, y = Y.lb, z = Z.lb;
    if (op == OP_MUL) c.ent = (x == y * z);
    else if (op == OP_TDIV) c.ent = (z != 0 && x == (long long)(Y.lb / (Z.lb == 0 ? 1 : Z.lb)));
    else c.ent = (z != 0 && x == (long long)(Y.lb % (Z.lb == 0 ? 1 : Z.lb)));
  }
  return c;
}

// `x = y * z` on NON-NEGATIVE operands (r05): products and quotients are monotone, so the hull of the corner products is [y.lb * z.lb, y.ub * z.ub] and the hull of the corner
// quotients [ceil(x.lb / z.ub), floor(x.ub / z.lb)] -- two multiplications and four unsigned divisions where evaluate_heavy walks every sign case (four 64-bit corner products,
// sixteen signed divisions with floor / ceil fix-ups).  Same candidate bounds as evaluate_heavy on every box with x.lb, y.lb, z.lb >= 0 that is not already empty: an infinite
// upper bound (INT32_MAX) saturates like mul_ext (0 * inf = 0), the quotient rules apply under the same conditions (x finite, divisor finite and > 0).
__device__ __forceinline__ int mul_sat_nn(int a, int b) {  // a, b >= 0
  const unsigned long long p = (unsigned long long)(unsigned)a * (unsigned long long)(unsigned)b;
  return p >= (unsigned long long)PINF ? PINF : (int)p;
}
__device__ __forceinline__ unsigned udiv1(int a) { return a > 1 ? (unsigned)a : 1u; }  // a divisor that is only used when it is positive
__device__ __forceinline__ Cand evaluate_mul_nn(const Itv X, const Itv Y, const Itv Z) {
  Cand c;
  c.xl = mul_sat_nn(Y.lb, Z.lb); c.xu = mul_sat_nn(Y.ub, Z.ub);
  if (X.lb > 0) {  // a non-zero product has non-zero factors
    if (Y.lb == 0) c.yl = 1;
    if (Y.ub == 0) c.yu = -1;
    if (Z.lb == 0) c.zl = 1;
    if (Z.ub == 0) c.zu = -1;
  }
  const bool x_fin = X.ub != PINF;
  const unsigned uxl = (unsigned)X.lb, uxu = (unsigned)imax(X.ub, 0);
  bool dy = x_fin && Z.lb > 0 && Z.ub != PINF, dz = x_fin && Y.lb > 0 && Y.ub != PINF;
  // Would a quotient narrow anything?  ceil(x.lb / z.ub) > y.lb <=> x.lb > y.lb * z.ub, floor(x.ub / z.lb) < y.ub <=> x.ub < y.ub * z.lb (and the same with y and z exchanged: the
  // same two products).  Two multiplications answer for the four divisions, ~25 VALU instructions each, which a sweep near its fixpoint -- most sweeps -- would compute to
  // learn nothing: a candidate that does not narrow is dropped by the caller's meet (r05: the synthetic network has one product in nearly every 64-record slice).
  {
    const unsigned long long p_ylzu = (unsigned long long)(unsigned)Y.lb * (unsigned long long)(unsigned)Z.ub, p_yuzl = (unsigned long long)(unsigned)Y.ub * (unsigned long long)(unsigned)Z.lb;
    dy = dy && ((unsigned long long)uxl > p_ylzu || (unsigned long long)uxu < p_yuzl);
    dz = dz && ((unsigned long long)uxl > p_yuzl || (unsigned long long)uxu < p_ylzu);
  }
  if (__builtin_amdgcn_ballot_w64(dy) != 0ull) {
    const unsigned du = udiv1(Z.ub), dl = udiv1(Z.lb);
    const int lo = (int)((uxl + du - 1u) / du), hi = (int)(uxu / dl);
    if (dy) { c.yl = imax(c.yl, lo); c.yu = imin(c.yu, hi); }
  }
  if (__builtin_amdgcn_ballot_w64(dz) != 0ull) {
    const unsigned du = udiv1(Y.ub), dl = udiv1(Y.lb);
    const int lo = (int)((uxl + du - 1u) / du), hi = (int)(uxu / dl);
    if (dz) { c.zl = imax(c.zl, lo); c.zu = imin(c.zu, hi); }
  }
  c.ent = X.lb == X.ub && Y.lb == Y.ub && Z.lb == Z.ub && x_fin && Y.ub != PINF && Z.ub != PINF && X.lb == c.xl;  // (c.xl is y.lb * z.lb, saturated: a finite x never equals the saturated value)
  return c;
}

// ---- packed records -----------------------------------------------------------------------------------
// word0 = class | original op << 12 | (set of classes present in the 64-record slice) << 16 ; words 1-3 = x,y,z.
// A comparison whose truth variable is a constant of the root store (TCN has no constants, only singleton
// variables: common_solving.hpp:743-771) gets its own class: `y <= z`, `y > z`, `y = z`, `y != z`.

enum Class : int {
  K_HEAVY = 0,  // MUL, TDIV, TMOD
  K_ADD = 1, K_MIN = 2, K_MAX = 3,
  K_EQ_R = 4, K_LEQ_R = 5,  // reified: x is a variable
  K_EQ_T = 6, K_EQ_F = 7,   // x is the constant true / false: y = z, y != z
  K_LEQ_T = 8, K_LEQ_F = 9  // y <= z, y > z
};

constexpr int CLASS_SET_MASK = 0x3ff;  // word0 bits 16-25: the classes present in the slice; bits 26-31: operand kinds (kernels.hpp)

__host__ __device__ inline int class_of(int op, bool x_const, int x_value) {
  switch (op) {
    case OP_ADD: return K_ADD;
    case OP_MIN: return K_MIN;
    case OP_MAX: return K_MAX;
    case OP_EQ: return x_const ? (x_value >= 1 ? K_EQ_T : K_EQ_F) : K_EQ_R;
    case OP_LEQ: return x_const ? (x_value >= 1 ? K_LEQ_T : K_LEQ_F) : K_LEQ_R;
    default: return K_HEAVY;
  }
}

// NNF: compile the fast path for products of non-negative operands (evaluate_mul_nn) into the heavy branch.  Only the kernels of networks in GLOBAL memory with the caller's record order
// (store layouts 3 and 5: the synthetic 100k x 500k network, where one product sits in almost every slice) take it: measured r05, same box -- synthetic wac1 +17 %, event +25 %, ac1 +16 %;
// but wordpress7_500 wac1 -15 %, ac1 -8 %, event -2 % when it is compiled into the LDS-resident kernels too (their five product slices gain, everything else pays for the registers),
// and as an out-of-line function for class-pure heavy slices of the LDS sweeps +0.6 % there against -7 % on accap_a3, which has no product at all (a call in the loop changes its
// register allocation): the LDS-resident sweeps keep the general rule.
template <int NNF = 0>
__device__ __forceinline__ Cand evaluate_packed(int w0, const Itv X, const Itv Y, const Itv Z);
// The rules for lanes holding ARBITRARY records (not the 64 records of one slice): the slice-level class set of word0 is
// replaced by "every class may be present", which sends evaluate_packed down its per-lane path.
template <int NNF = 0>
__device__ __forceinline__ Cand evaluate_single(int w0, const Itv X, const Itv Y, const Itv Z) {
  return evaluate_packed<NNF>((w0 & 0xffff) | (CLASS_SET_MASK << 16), X, Y, Z);
}

template <int NNF>
__device__ __forceinline__ Cand evaluate_packed(int w0, const Itv X, const Itv Y, const Itv Z) {
  Cand c;
  int cls = w0 & 0xf;  // (bits 4-9: which narrowings of the record's operands have a reader outside its slice, kernels.hpp: mark_successors)
  const int present = (__builtin_amdgcn_readfirstlane(w0) >> 16) & CLASS_SET_MASK;
  bool ent = false;
  // Class-pure slices (the records are sorted by class, engine.hip: to_internal): the commonest classes get a body
  // without any per-lane class predicate.
  if (present == (1 << K_LEQ_T)) {  // y <= z
    c.yu = Z.ub; c.zl = Y.lb;
    c.ent = Y.ub <= Z.lb;
    return c;
  }
  if (present == (1 << K_EQ_R)) {  // x = (y = z), x a variable
    const bool t = X.lb >= 1, f = X.ub <= 0, u = !t && !f;
    const bool ys = Y.lb == Y.ub, zs = Z.lb == Z.ub;
    const bool disjoint = Y.ub < Z.lb || Y.lb > Z.ub;
    const bool same = ys && zs && Y.lb == Z.lb;
    const bool fy = f && ys, fz = f && zs;
    c.xl = sel(u && same, 1, c.xl);
    c.xu = sel(u && disjoint, 0, c.xu);
    c.yl = sel(t, Z.lb, sel(fz && Y.lb == Z.lb, sat_add(Z.lb, 1), c.yl));
    c.yu = sel(t, Z.ub, sel(fz && Y.ub == Z.lb, sat_sub(Z.lb, 1), c.yu));
    c.zl = sel(t, Y.lb, sel(fy && Z.lb == Y.lb, sat_add(Y.lb, 1), c.zl));
    c.zu = sel(t, Y.ub, sel(fy && Z.ub == Y.lb, sat_sub(Y.lb, 1), c.zu));
    c.ent = (t && same) || (f && disjoint);
    return c;
  }
  if (present == (1 << K_LEQ_R)) {  // x = (y <= z), x a variable
    const bool t = X.lb >= 1, f = X.ub <= 0, u = !t && !f;
    const bool le = Y.ub <= Z.lb, gt = Y.lb > Z.ub;
    c.xl = sel(u && le, 1, c.xl);
    c.xu = sel(u && gt, 0, c.xu);
    c.yu = sel(t, Z.ub, c.yu);
    c.zl = sel(t, Y.lb, c.zl);
    c.yl = sel(f, add_lo(Z.lb, 1), c.yl);
    c.zu = sel(f, add_hi(Y.ub, -1), c.zu);
    c.ent = (t && le) || (f && gt);
    return c;
  }
  if (present == (1 << K_ADD)) {  // x = y + z
    c.xl = add_lo(Y.lb, Z.lb); c.xu = add_hi(Y.ub, Z.ub);
    c.yl = sub_lo(X.lb, Z.ub); c.yu = sub_hi(X.ub, Z.lb);
    c.zl = sub_lo(X.lb, Y.ub); c.zu = sub_hi(X.ub, Y.lb);
    c.ent = X.lb == X.ub && Y.lb == Y.ub && Z.lb == Z.ub && (long long)X.lb == (long long)Y.lb + (long long)Z.lb;
    return c;
  }
  if (present == (1 << K_MIN) || present == (1 << K_MAX)) {  // x = min(y, z) / x = max(y, z)
    const bool fixed = X.lb == X.ub && Y.lb == Y.ub && Z.lb == Z.ub;
    if (present == (1 << K_MIN)) {
      const int mnl = imin(Y.lb, Z.lb);
      c.xl = mnl; c.xu = imin(Y.ub, Z.ub);
      c.yl = X.lb; c.zl = X.lb;
      c.yu = sel(Z.lb > X.ub, X.ub, c.yu);
      c.zu = sel(Y.lb > X.ub, X.ub, c.zu);
      c.ent = fixed && X.lb == mnl;
    } else {
      const int mxl = imax(Y.lb, Z.lb);
      c.xl = mxl; c.xu = imax(Y.ub, Z.ub);
      c.yu = X.ub; c.zu = X.ub;
      c.yl = sel(Z.ub < X.lb, X.lb, c.yl);
      c.zl = sel(Y.ub < X.lb, X.lb, c.zl);
      c.ent = fixed && X.lb == mxl;
    }
    return c;
  }
  // Mixed slice (class boundaries only, once the records are sorted).  The per-lane class predicates below are loop
  // invariant for the caller's wave-local loop; hoisted, they would sit in ~28 SGPRs across the whole loop and push
  // the common class-pure paths into SGPR spills (v_readlane / v_writelane are VALU work).  The empty asm makes the
  // class look freshly written, so they are recomputed here, where they are needed.
  asm volatile("" : "+v"(cls));
  // predicates shared by the comparison classes
  const bool xt = X.lb >= 1, xf = X.ub <= 0;
  if (NNF == 1 && (present & ((1 << K_LEQ_T) | (1 << K_LEQ_F) | (1 << K_LEQ_R))) == (1 << K_LEQ_T)) {
    // (r05, the kernels of stores in global memory, whose records keep the caller's order and whose slices therefore mix classes as a rule: the synthetic network's
    //  comparisons are all `y <= z` and `y != z` with constant truth values -- the bodies for just those, without the reified and negated cases, are a third as long)
    const bool t = cls == K_LEQ_T;
    c.yu = sel(t, Z.ub, c.yu);
    c.zl = sel(t, Y.lb, c.zl);
    ent = t && Y.ub <= Z.lb;
  } else if (present & ((1 << K_LEQ_T) | (1 << K_LEQ_F) | (1 << K_LEQ_R))) {
    // x = (y <= z); T/F: x is a constant of that value
    const bool t = cls == K_LEQ_T || (cls == K_LEQ_R && xt);
    const bool f = cls == K_LEQ_F || (cls == K_LEQ_R && xf);
    const bool u = cls == K_LEQ_R && !xt && !xf;
    const bool le = Y.ub <= Z.lb, gt = Y.lb > Z.ub;
    c.xl = sel(u && le, 1, c.xl);
    c.xu = sel(u && gt, 0, c.xu);
    c.yu = sel(t, Z.ub, c.yu);
    c.zl = sel(t, Y.lb, c.zl);
    c.yl = sel(f, add_lo(Z.lb, 1), c.yl);
    c.zu = sel(f, add_hi(Y.ub, -1), c.zu);
    ent = (t && le) || (f && gt);
  }
  if (NNF == 1 && (present & ((1 << K_EQ_T) | (1 << K_EQ_F) | (1 << K_EQ_R))) == (1 << K_EQ_F)) {
    const bool f = cls == K_EQ_F;  // y != z
    const bool fy = f && Y.lb == Y.ub, fz = f && Z.lb == Z.ub;
    c.yl = sel(fz && Y.lb == Z.lb, sat_add(Z.lb, 1), c.yl);
    c.yu = sel(fz && Y.ub == Z.lb, sat_sub(Z.lb, 1), c.yu);
    c.zl = sel(fy && Z.lb == Y.lb, sat_add(Y.lb, 1), c.zl);
    c.zu = sel(fy && Z.ub == Y.lb, sat_sub(Y.lb, 1), c.zu);
    ent = ent || (f && (Y.ub < Z.lb || Y.lb > Z.ub));
  } else if (present & ((1 << K_EQ_T) | (1 << K_EQ_F) | (1 << K_EQ_R))) {
    // x = (y = z)
    const bool t = cls == K_EQ_T || (cls == K_EQ_R && xt);
    const bool f = cls == K_EQ_F || (cls == K_EQ_R && xf);
    const bool u = cls == K_EQ_R && !xt && !xf;
    const bool ys = Y.lb == Y.ub, zs = Z.lb == Z.ub;
    const bool disjoint = Y.ub < Z.lb || Y.lb > Z.ub;
    const bool same = ys && zs && Y.lb == Z.lb;
    const bool fy = f && ys, fz = f && zs;
    c.xl = sel(u && same, 1, c.xl);
    c.xu = sel(u && disjoint, 0, c.xu);
    c.yl = sel(t, Z.lb, sel(fz && Y.lb == Z.lb, sat_add(Z.lb, 1), c.yl));
    c.yu = sel(t, Z.ub, sel(fz && Y.ub == Z.lb, sat_sub(Z.lb, 1), c.yu));
    c.zl = sel(t, Y.lb, sel(fy && Z.lb == Y.lb, sat_add(Y.lb, 1), c.zl));
    c.zu = sel(t, Y.ub, sel(fy && Z.ub == Y.lb, sat_sub(Y.lb, 1), c.zu));
    ent = ent || (t && same) || (f && disjoint);
  }
  if (present & ((1 << K_ADD) | (1 << K_MIN) | (1 << K_MAX))) {
    const bool ka = cls == K_ADD, kmin = cls == K_MIN, kmax = cls == K_MAX;
    const bool fixed = X.lb == X.ub && Y.lb == Y.ub && Z.lb == Z.ub;
    if (present & (1 << K_ADD)) {
      c.xl = sel(ka, add_lo(Y.lb, Z.lb), c.xl); c.xu = sel(ka, add_hi(Y.ub, Z.ub), c.xu);
      c.yl = sel(ka, sub_lo(X.lb, Z.ub), c.yl); c.yu = sel(ka, sub_hi(X.ub, Z.lb), c.yu);
      c.zl = sel(ka, sub_lo(X.lb, Y.ub), c.zl); c.zu = sel(ka, sub_hi(X.ub, Y.lb), c.zu);
      ent = ent || (ka && fixed && (long long)X.lb == (long long)Y.lb + (long long)Z.lb);
    }
    if (present & ((1 << K_MIN) | (1 << K_MAX))) {
      const int mnl = imin(Y.lb, Z.lb), mnu = imin(Y.ub, Z.ub), mxl = imax(Y.lb, Z.lb), mxu = imax(Y.ub, Z.ub);
      c.xl = sel(kmin, mnl, sel(kmax, mxl, c.xl));
      c.xu = sel(kmin, mnu, sel(kmax, mxu, c.xu));
      c.yl = sel(kmin || (kmax && Z.ub < X.lb), X.lb, c.yl);
      c.zl = sel(kmin || (kmax && Y.ub < X.lb), X.lb, c.zl);
      c.yu = sel(kmax || (kmin && Z.lb > X.ub), X.ub, c.yu);
      c.zu = sel(kmax || (kmin && Y.lb > X.ub), X.ub, c.zu);
      ent = ent || (fixed && ((kmin && X.lb == mnl) || (kmax && X.lb == mxl)));
    }
  }
  if (present & (1 << K_HEAVY)) {
    // (r05) The heavy lanes of most networks are products of non-negative operands -- coefficient x Boolean terms of linear constraints, price x quantity; the synthetic
    // 100k x 500k network's 8 147 products sit one to a 64-record slice, so that EVERY slice of a sweep walked evaluate_heavy for a single lane.  When every heavy lane of
    // the wave is such a product (one vote) they take evaluate_mul_nn; a TDIV / TMOD or a factor that may be negative sends the wave's heavy lanes down the general rule.
    const bool heavy = cls == K_HEAVY;
    const bool nn = heavy && ((w0 >> 12) & 0x7) == OP_MUL && X.lb >= 0 && Y.lb >= 0 && Z.lb >= 0 && X.lb <= X.ub && Y.lb <= Y.ub && Z.lb <= Z.ub;
    if (NNF == 1 && __builtin_amdgcn_ballot_w64(heavy && !nn) == 0ull) {
      if (heavy) { c = evaluate_mul_nn(X, Y, Z); ent = c.ent; }
    } else if (heavy) {
      c = evaluate_heavy((w0 >> 12) & 0x7, X, Y, Z);
      ent = c.ent;
    }
  }
  c.ent = ent;
  return c;
}

}  // namespace tb
