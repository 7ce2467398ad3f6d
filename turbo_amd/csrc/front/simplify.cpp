// TCN simplifier (front-end, host side).
//
// Behavioural counterpart of the preprocessing loop of the reference
// (include/common_solving.hpp:537-585: root propagation, meet_equivalence_classes, algebraic_simplify,
// eliminate_entailed_constraints, i_cse, eliminate_useless_variables, deinterpret + re-interpret) -- the
// reference delegates every step to lala's `Simplifier`, which is not part of its tree, so this is an own
// implementation of the same steps over the flat `{op,x,y,z}` network:
//   1. take the root fixpoint computed by the caller (the GPU engine in production: the propagators stay
//      single-sourced) and intersect it into the store;
//   2. equivalence classes (union-find): `1 = (y = z)`, `x = y + 0`, `x = y * 1`, `x = min/max(y, y)`,
//      `b = (y = 1)` over a 0/1 variable, equal (op, y, z) pairs (common subexpressions);
//   3. drop entailed propagators and duplicates;
//   4. drop variables that no propagator, objective, strategy or output needs;
//   5. renumber (constants 0, 1, 2 first) and keep the original -> simplified map so that solutions print in the
//      original variables (common_solving.hpp:848-851).
// The simplified network has the same optimum and every solution of it expands to a solution of the original.
#include <algorithm>
#include <map>
#include <numeric>
#include <tuple>

#include "tcn.hpp"

namespace turbo_front {

namespace {

constexpr int32_t NINF = TB_NINF, PINF = TB_PINF;

struct UnionFind {
  std::vector<int32_t> parent;
  explicit UnionFind(size_t n) : parent(n) { std::iota(parent.begin(), parent.end(), 0); }
  int32_t find(int32_t v) {
    while (parent[(size_t)v] != v) { parent[(size_t)v] = parent[(size_t)parent[(size_t)v]]; v = parent[(size_t)v]; }
    return v;
  }
};

inline bool is_single(const tb_itv& d) { return d.lb == d.ub && d.lb != NINF && d.lb != PINF; }

// sound entailment test (same rules as the engine's `ask`)
bool entailed(const tb_prop& p, const std::vector<tb_itv>& st) {
  const tb_itv X = st[(size_t)p.x], Y = st[(size_t)p.y], Z = st[(size_t)p.z];
  switch (p.op) {
    case TB_EQ: return (X.lb >= 1 && is_single(Y) && is_single(Z) && Y.lb == Z.lb) || (X.ub <= 0 && (Y.ub < Z.lb || Y.lb > Z.ub));
    case TB_LEQ: return (X.lb >= 1 && Y.ub <= Z.lb) || (X.ub <= 0 && Y.lb > Z.ub);
    default: break;
  }
  if (!is_single(X) || !is_single(Y) || !is_single(Z)) return false;
  const int64_t x = X.lb, y = Y.lb, z = Z.lb;
  switch (p.op) {
    case TB_ADD: return x == y + z;
    case TB_MUL: return x == y * z;
    case TB_TDIV: return z != 0 && x == y / z;
    case TB_TMOD: return z != 0 && x == y % z;
    case TB_MIN: return x == std::min(y, z);
    case TB_MAX: return x == std::max(y, z);
    default: return false;
  }
}

}  // namespace

void simplify_tcn(TCN& t, const tb_itv* root_fixpoint, SimplifyInfo* info) {
  const size_t V = t.store.size();
  SimplifyInfo inf;
  inf.original_vars = (int32_t)V;
  inf.original_props = (int32_t)t.props.size();
  if (!t.simplified) {  // keep the very first network: solutions are expanded back to it
    t.original_store = t.store;
    t.original_props = t.props;
    t.expand_var.resize(V);
    std::iota(t.expand_var.begin(), t.expand_var.end(), 0);
    t.expand_const.assign(V, 0);
    t.simplified = true;
  }
  std::vector<tb_itv> st = t.store;
  if (root_fixpoint)
    for (size_t v = 0; v < V; ++v) {
      st[v].lb = std::max(st[v].lb, root_fixpoint[v].lb);
      st[v].ub = std::min(st[v].ub, root_fixpoint[v].ub);
      if (st[v].lb > st[v].ub) t.trivially_unsat = true;
    }
  if (t.trivially_unsat) { if (info) *info = inf; return; }

  UnionFind uf(V);
  auto unite = [&](int32_t a, int32_t b) -> bool {
    a = uf.find(a); b = uf.find(b);
    if (a == b) return false;
    // representative: a constant if there is one, otherwise the lower index
    const bool ac = is_single(st[(size_t)a]), bc = is_single(st[(size_t)b]);
    if ((bc && !ac) || (ac == bc && b < a)) std::swap(a, b);
    uf.parent[(size_t)b] = a;
    st[(size_t)a].lb = std::max(st[(size_t)a].lb, st[(size_t)b].lb);
    st[(size_t)a].ub = std::min(st[(size_t)a].ub, st[(size_t)b].ub);
    if (st[(size_t)a].lb > st[(size_t)a].ub) t.trivially_unsat = true;
    ++inf.merged_variables;
    return true;
  };
  auto is_const = [&](int32_t v, int64_t c) { const tb_itv d = st[(size_t)uf.find(v)]; return d.lb == d.ub && d.lb == c; };
  auto fix = [&](int32_t v, int32_t c) {
    tb_itv& d = st[(size_t)uf.find(v)];
    d.lb = std::max(d.lb, c); d.ub = std::min(d.ub, c);
    if (d.lb > d.ub) t.trivially_unsat = true;
  };

  std::vector<tb_prop> props = t.props;
  bool changed = true;
  int rounds = 0;
  while (changed && !t.trivially_unsat && rounds++ < 16) {
    changed = false;
    // --- algebraic simplification -> equivalences
    for (tb_prop& p : props) {
      p.x = uf.find(p.x); p.y = uf.find(p.y); p.z = uf.find(p.z);
      switch (p.op) {
        case TB_ADD:
          if (is_const(p.z, 0)) changed |= unite(p.x, p.y);
          else if (is_const(p.y, 0)) changed |= unite(p.x, p.z);
          break;
        case TB_MUL:
          if (is_const(p.z, 1)) changed |= unite(p.x, p.y);
          else if (is_const(p.y, 1)) changed |= unite(p.x, p.z);
          else if (is_const(p.y, 0) || is_const(p.z, 0)) fix(p.x, 0);
          break;
        case TB_MIN:
        case TB_MAX:
          if (p.y == p.z) changed |= unite(p.x, p.y);
          break;
        case TB_EQ: {
          const tb_itv X = st[(size_t)p.x];
          if (X.lb >= 1) changed |= unite(p.y, p.z);
          else if (p.y == p.z) fix(p.x, 1);
          else if (X.ub > 0) {  // b = (y = 1) over a 0/1 variable: b is y
            const tb_itv Y = st[(size_t)p.y], Z = st[(size_t)p.z];
            if (is_const(p.z, 1) && Y.lb >= 0 && Y.ub <= 1 && X.lb >= 0 && X.ub <= 1) changed |= unite(p.x, p.y);
            else if (is_const(p.y, 1) && Z.lb >= 0 && Z.ub <= 1 && X.lb >= 0 && X.ub <= 1) changed |= unite(p.x, p.z);
          }
          break;
        }
        case TB_LEQ:
          if (p.y == p.z) fix(p.x, 1);
          break;
        default: break;
      }
    }
    // --- common subexpressions: equal (op, y, z) define equal x
    std::map<std::tuple<int32_t, int32_t, int32_t>, int32_t> seen;
    for (tb_prop& p : props) {
      p.x = uf.find(p.x); p.y = uf.find(p.y); p.z = uf.find(p.z);
      int32_t y = p.y, z = p.z;
      if (p.op == TB_ADD || p.op == TB_MUL || p.op == TB_MIN || p.op == TB_MAX || p.op == TB_EQ) { if (z < y) std::swap(y, z); }
      auto key = std::make_tuple(p.op, y, z);
      auto it = seen.find(key);
      if (it == seen.end()) seen.emplace(key, p.x);
      else if (uf.find(it->second) != p.x) { if (unite(it->second, p.x)) { changed = true; ++inf.cse_merges; } }
    }
  }
  if (t.trivially_unsat) { if (info) *info = inf; return; }

  // --- rewrite, drop entailed and duplicate propagators
  std::vector<tb_itv> rep_store = st;
  for (size_t v = 0; v < V; ++v) rep_store[v] = st[(size_t)uf.find((int32_t)v)];
  std::vector<tb_prop> kept;
  {
    std::map<std::tuple<int32_t, int32_t, int32_t, int32_t>, char> dup;
    for (tb_prop p : props) {
      p.x = uf.find(p.x); p.y = uf.find(p.y); p.z = uf.find(p.z);
      if (entailed(p, rep_store)) { ++inf.entailed_props; continue; }
      {  // tautologies left behind by the merges: y = y, y <= y, x = x + 0, x = x * 1, x = min/max(x, x), b = (b = 1)
        const tb_itv X = rep_store[(size_t)p.x];
        const auto konst = [&](int32_t v, int32_t c) { return rep_store[(size_t)v].lb == c && rep_store[(size_t)v].ub == c; };
        bool taut = false;
        if ((p.op == TB_EQ || p.op == TB_LEQ) && p.y == p.z && X.lb >= 1) taut = true;
        if (p.op == TB_ADD && ((p.x == p.y && konst(p.z, 0)) || (p.x == p.z && konst(p.y, 0)))) taut = true;
        if (p.op == TB_MUL && ((p.x == p.y && konst(p.z, 1)) || (p.x == p.z && konst(p.y, 1)))) taut = true;
        if ((p.op == TB_MIN || p.op == TB_MAX) && p.x == p.y && p.y == p.z) taut = true;
        if (p.op == TB_EQ && X.lb >= 0 && X.ub <= 1 && ((p.x == p.y && konst(p.z, 1)) || (p.x == p.z && konst(p.y, 1)))) taut = true;
        if (taut) { ++inf.entailed_props; continue; }
      }
      if (!dup.emplace(std::make_tuple(p.op, p.x, p.y, p.z), 1).second) { ++inf.duplicate_props; continue; }
      kept.push_back(p);
    }
  }

  // --- which representatives are still needed
  std::vector<char> needed(V, 0);
  for (int32_t c = 0; c < 3 && (size_t)c < V; ++c) needed[(size_t)uf.find(c)] = 1;  // the interned constants 0, 1, 2 stay
  for (const tb_prop& p : kept) { needed[(size_t)p.x] = needed[(size_t)p.y] = needed[(size_t)p.z] = 1; }
  if (t.obj_var >= 0) needed[(size_t)uf.find(t.obj_var)] = 1;
  if (t.goal_var >= 0) needed[(size_t)uf.find(t.goal_var)] = 1;
  // strategy variables are kept only if some propagator still constrains them (branching on a free variable
  // cannot change the outcome; its value is its lower bound)
  std::vector<int32_t> new_id(V, -1);
  std::vector<tb_itv> new_store;
  std::vector<std::string> new_names;
  // constants first, in the conventional order 0, 1, 2
  for (int32_t c = 0; c < 3 && (size_t)c < V; ++c) {
    const int32_t r = uf.find(c);
    if (new_id[(size_t)r] < 0) { new_id[(size_t)r] = (int32_t)new_store.size(); new_store.push_back(rep_store[(size_t)r]); new_names.push_back(t.names[(size_t)r]); }
  }
  for (size_t v = 0; v < V; ++v) {
    const int32_t r = uf.find((int32_t)v);
    if ((size_t)r != v || !needed[v] || new_id[v] >= 0) continue;
    new_id[v] = (int32_t)new_store.size();
    new_store.push_back(rep_store[v]);
    new_names.push_back(t.names[v]);
  }
  inf.eliminated_variables = (int32_t)V - (int32_t)new_store.size();

  // --- compose the expansion map: original variable -> simplified variable, or a constant
  for (size_t o = 0; o < t.expand_var.size(); ++o) {
    const int32_t cur = t.expand_var[o];
    if (cur < 0) continue;  // already a constant from an earlier pass
    const int32_t r = uf.find(cur);
    if (new_id[(size_t)r] >= 0) t.expand_var[o] = new_id[(size_t)r];
    else { t.expand_var[o] = -1; t.expand_const[o] = rep_store[(size_t)r].lb; }  // free variable: lower corner of its domain
  }

  for (tb_prop& p : kept) { p.x = new_id[(size_t)p.x]; p.y = new_id[(size_t)p.y]; p.z = new_id[(size_t)p.z]; }
  for (Strategy& s : t.strategies) {
    std::vector<int32_t> vs;
    for (int32_t v : s.vars) {
      const int32_t n = new_id[(size_t)uf.find(v)];
      if (n >= 0 && std::find(vs.begin(), vs.end(), n) == vs.end()) vs.push_back(n);
    }
    s.emptied = !s.vars.empty() && vs.empty();
    s.vars.swap(vs);
  }
  // a strategy whose variables all disappeared must not turn into "whole store"
  t.strategies.erase(std::remove_if(t.strategies.begin(), t.strategies.end(), [](const Strategy& s) { return s.emptied; }), t.strategies.end());
  if (t.obj_var >= 0) t.obj_var = new_id[(size_t)uf.find(t.obj_var)];
  if (t.goal_var >= 0) t.goal_var = new_id[(size_t)uf.find(t.goal_var)];
  t.store.swap(new_store);
  t.names.swap(new_names);
  t.props.swap(kept);
  t.flatten_strategies();
  inf.simplified_vars = (int32_t)t.store.size();
  inf.simplified_props = (int32_t)t.props.size();
  if (info) *info = inf;
}

void expand_solution(const TCN& t, const tb_itv* simplified, tb_itv* original_out) {
  if (!t.simplified) {
    std::copy(simplified, simplified + t.store.size(), original_out);
    return;
  }
  for (size_t o = 0; o < t.expand_var.size(); ++o) {
    if (t.expand_var[o] >= 0) original_out[o] = simplified[(size_t)t.expand_var[o]];
    else original_out[o] = tb_itv{t.expand_const[o], t.expand_const[o]};
  }
}

}  // namespace turbo_front
