// FlatZinc model -> ternary constraint network (TCN): every propagator is `x = y op z` over
// 32-bit interval domains, every constant is a (singleton) variable, comparisons are only `=` and
// `<=` with negation written `0 = (y op z)` -- the conventions of the reference's TCN
// (include/common_solving.hpp:520-527,739-771).  The lowering itself is this project's own: the
// reference delegates to lala's `ternarize`/`normalize`, which are not part of the reference tree.
//
// Design choices that matter for the GPU fixpoint:
//   * n-ary sums / and / or are lowered to BALANCED trees (depth log n), so a Jacobi-style parallel
//     sweep needs O(log n) iterations per constraint instead of O(n);
//   * `sum - sum' <= c` is lowered to `P = S + N` plus a unary bound on S (no multiplication by -1);
//   * reified `var == const` booleans are shared (hash-consed) between element constraints.
#include "tcn.hpp"

#include <algorithm>
#include <climits>
#include <functional>
#include <sstream>

namespace turbo_front {

namespace {

constexpr int32_t NINF = TB_NINF, PINF = TB_PINF;
constexpr int64_t FIN_MIN = (int64_t)INT32_MIN + 1, FIN_MAX = (int64_t)INT32_MAX - 1;

inline bool is_inf(int32_t a) { return a == NINF || a == PINF; }
inline int32_t clamp64(int64_t v) { return v >= PINF ? PINF : (v <= NINF ? NINF : (int32_t)v); }
inline int32_t neg_ext(int32_t a) { return a == NINF ? PINF : (a == PINF ? NINF : -a); }
inline int32_t add_lo(int32_t a, int32_t b) {
  if (a == NINF || b == NINF) return NINF;
  if (a == PINF || b == PINF) return PINF;
  return clamp64((int64_t)a + b);
}
inline int32_t add_hi(int32_t a, int32_t b) {
  if (a == PINF || b == PINF) return PINF;
  if (a == NINF || b == NINF) return NINF;
  return clamp64((int64_t)a + b);
}
inline int32_t mul_ext(int32_t a, int32_t b) {
  if (a == 0 || b == 0) return 0;
  if (is_inf(a) || is_inf(b)) return ((a < 0) != (b < 0)) ? NINF : PINF;
  return clamp64((int64_t)a * b);
}

using Ranges = std::vector<std::pair<int64_t, int64_t>>;

class Lowerer {
 public:
  explicit Lowerer(const Model& m) : m_(m) {}

  TCN run() {
    ZERO = konst(0); ONE = konst(1); TWO = konst(2);  // common_solving.hpp:521
    declare();
    for (const Constraint& c : m_.constraints) {
      std::vector<Expr> args = c.args;
      post_pred(c.name, args, nullptr);
    }
    t_.parsed_constraints = (int32_t)m_.constraints.size();
    goal();
    search_annotations();
    Strategy dflt;  // default strategy over the whole store, common_solving.hpp:640-650
    dflt.var_order = TB_FIRST_FAIL; dflt.val_order = TB_VAL_MIN;
    t_.strategies.push_back(dflt);
    t_.flatten_strategies();
    return std::move(t_);
  }

 private:
  const Model& m_;
  TCN t_;
  int32_t ZERO = -1, ONE = -1, TWO = -1;
  std::map<int64_t, int32_t> konst_cache_;
  std::map<std::string, Term> scalars_;
  std::map<std::string, std::vector<Term>> arrays_;
  std::map<std::string, std::vector<Expr>> set_arrays_;
  std::map<std::pair<int32_t, int32_t>, int32_t> eq_cache_;

  [[noreturn]] void fail(const std::string& msg) const { throw LowerError("FlatZinc model error: " + msg); }

  // ---------------------------------------------------------------- variables
  int32_t new_var(int64_t lb, int64_t ub, const std::string& name = "") {
    t_.store.push_back(tb_itv{clamp64(lb), clamp64(ub)});
    t_.names.push_back(name);
    if (lb > ub) t_.trivially_unsat = true;
    return (int32_t)t_.store.size() - 1;
  }
  int32_t new_bool() { return new_var(0, 1); }
  int32_t konst(int64_t v) {
    if (v < FIN_MIN || v > FIN_MAX) fail("integer constant " + std::to_string(v) + " does not fit the 32-bit interval domain");
    auto it = konst_cache_.find(v);
    if (it != konst_cache_.end()) return it->second;
    int32_t id = new_var(v, v);
    konst_cache_[v] = id;
    return id;
  }
  int32_t as_var(const Term& t) { return t.is_const ? konst(t.value) : t.var; }
  tb_itv dom(int32_t v) const { return t_.store[v]; }
  tb_itv dom(const Term& t) const {
    if (t.is_const) return tb_itv{clamp64(t.value), clamp64(t.value)};
    return t_.store[t.var];
  }
  void restrict(int32_t v, int64_t lb, int64_t ub) {
    tb_itv& d = t_.store[v];
    if (lb > d.lb) d.lb = clamp64(lb);
    if (ub < d.ub) d.ub = clamp64(ub);
    if (d.lb > d.ub) t_.trivially_unsat = true;
  }
  void restrict(const Term& t, int64_t lb, int64_t ub) {
    if (t.is_const) { if (t.value < lb || t.value > ub) t_.trivially_unsat = true; }
    else restrict(t.var, lb, ub);
  }
  void prop(int32_t op, int32_t x, int32_t y, int32_t z) { t_.props.push_back(tb_prop{op, x, y, z}); }

  // ---------------------------------------------------------------- expression resolution
  static bool has_ann(const std::vector<Expr>& anns, const char* name) {
    for (const Expr& a : anns) if ((a.kind == Expr::ID || a.kind == Expr::CALL) && a.name == name) return true;
    return false;
  }

  Term resolve(const Expr& e) {
    switch (e.kind) {
      case Expr::INT: return Term::konst(e.ival);
      case Expr::BOOL: return Term::konst(e.ival);
      case Expr::ID: {
        auto it = scalars_.find(e.name);
        if (it == scalars_.end()) fail("unknown identifier `" + e.name + "`");
        return it->second;
      }
      case Expr::INDEX: {
        auto it = arrays_.find(e.name);
        if (it == arrays_.end()) fail("unknown array `" + e.name + "`");
        if (e.ival < 1 || e.ival > (int64_t)it->second.size()) fail("index out of bounds in `" + e.name + "`");
        return it->second[(size_t)e.ival - 1];
      }
      case Expr::CALL: {  // nested predicate: its truth value (lala accepts `int_eq(b, int_le(0,y))`)
        int32_t r = new_bool();
        Term rt = Term::variable(r);
        std::vector<Expr> args = e.args;
        post_pred(e.name, args, &rt);
        return rt;
      }
      default: fail("expected a scalar expression");
    }
  }

  std::vector<Term> resolve_array(const Expr& e) {
    std::vector<Term> out;
    if (e.kind == Expr::ARRAY) {
      out.reserve(e.args.size());
      for (const Expr& a : e.args) out.push_back(resolve(a));
      return out;
    }
    if (e.kind == Expr::ID) {
      auto it = arrays_.find(e.name);
      if (it == arrays_.end()) fail("unknown array `" + e.name + "`");
      return it->second;
    }
    fail("expected an array expression");
  }

  std::vector<int64_t> const_array(const Expr& e) {
    std::vector<Term> ts = resolve_array(e);
    std::vector<int64_t> out;
    out.reserve(ts.size());
    for (const Term& t : ts) {
      if (!t.is_const) fail("expected an array of constants");
      out.push_back(t.value);
    }
    return out;
  }

  Ranges resolve_set(const Expr& e) {
    Ranges r;
    if (e.kind == Expr::RANGE) { if (e.ival <= e.ival2) r.push_back({e.ival, e.ival2}); return r; }
    if (e.kind == Expr::SETLIT) {
      std::vector<int64_t> vals;
      for (const Expr& a : e.args) {
        if (a.kind == Expr::INT) vals.push_back(a.ival);
        else if (a.kind == Expr::RANGE) { for (int64_t v = a.ival; v <= a.ival2; ++v) vals.push_back(v); }
        else fail("expected integers in a set literal");
      }
      std::sort(vals.begin(), vals.end());
      vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
      for (int64_t v : vals) {
        if (!r.empty() && r.back().second + 1 == v) r.back().second = v;
        else r.push_back({v, v});
      }
      return r;
    }
    if (e.kind == Expr::ID) {
      auto it = m_.params.find(e.name);
      if (it == m_.params.end()) fail("unknown set `" + e.name + "`");
      return resolve_set(it->second);
    }
    if (e.kind == Expr::INDEX) {
      auto it = set_arrays_.find(e.name);
      if (it == set_arrays_.end() || e.ival < 1 || e.ival > (int64_t)it->second.size()) fail("bad set array access `" + e.name + "`");
      return resolve_set(it->second[(size_t)e.ival - 1]);
    }
    fail("expected a set expression");
  }

  // ---------------------------------------------------------------- declarations
  void add_output(const std::string& name, bool is_array, bool is_bool, const std::vector<Expr>& anns, const std::vector<Term>& terms) {
    OutputItem o;
    o.name = name; o.is_array = is_array; o.is_bool = is_bool;
    if (is_array) {
      for (const Expr& a : anns)
        if (a.kind == Expr::CALL && a.name == "output_array" && !a.args.empty() && a.args[0].kind == Expr::ARRAY)
          for (const Expr& d : a.args[0].args) {
            if (d.kind == Expr::RANGE) o.dims.push_back({d.ival, d.ival2});
            else if (d.kind == Expr::ID || d.kind == Expr::SETLIT) { Ranges rr = resolve_set(d); o.dims.push_back(rr.empty() ? std::make_pair<int64_t, int64_t>(1, 0) : std::make_pair(rr.front().first, rr.back().second)); }
          }
      if (o.dims.empty()) o.dims.push_back({1, (int64_t)terms.size()});
    }
    for (const Term& t : terms) o.vars.push_back(as_var(t));
    t_.outputs.push_back(std::move(o));
  }

  void declare() {
    // scalar parameters first (they may be used as array sizes / values)
    for (const auto& kv : m_.params) {
      const Expr& v = kv.second;
      if (v.kind == Expr::INT || v.kind == Expr::BOOL) scalars_[kv.first] = Term::konst(v.ival);
    }
    for (const Model::Item& item : m_.order) {
      if (item.kind == Model::Item::VAR) {
        const VarDecl& v = m_.vars[item.index];
        if (v.is_set_var) fail("set variables are not supported (`" + v.name + "`)");
        ++t_.parsed_variables;
        Term t;
        if (v.has_init) {
          t = resolve(v.init);
          if (v.has_dom) restrict(t, v.lb, v.ub);
        } else {
          int64_t lb = v.has_dom ? v.lb : (int64_t)NINF, ub = v.has_dom ? v.ub : (int64_t)PINF;
          t = Term::variable(new_var(lb, ub, v.name));
        }
        scalars_[v.name] = t;
        if (!v.set_values.empty()) holes(t, v.set_values);
        if (has_ann(v.anns, "output_var")) add_output(v.name, false, v.is_bool, v.anns, {t});
      } else {
        const ArrayDecl& a = m_.arrays[item.index];
        if (a.elem_set) {
          if (a.is_var) fail("arrays of set variables are not supported (`" + a.name + "`)");
          set_arrays_[a.name] = a.elems;
          continue;
        }
        std::vector<Term> terms;
        if (a.has_init) {
          terms.reserve(a.elems.size());
          for (const Expr& e : a.elems) terms.push_back(resolve(e));
        } else {
          if (!a.is_var) fail("parameter array `" + a.name + "` has no value");
          for (int64_t i = 1; i <= a.size; ++i) {  // lala extension: fresh variables (test_data/bug5.fzn)
            int64_t lb = a.elem_has_dom ? a.elem_lb : (int64_t)NINF, ub = a.elem_has_dom ? a.elem_ub : (int64_t)PINF;
            terms.push_back(Term::variable(new_var(lb, ub, a.name + "[" + std::to_string(i) + "]")));
            ++t_.parsed_variables;
          }
        }
        if (has_ann(a.anns, "output_array")) add_output(a.name, true, a.elem_bool, a.anns, terms);
        arrays_[a.name] = std::move(terms);
      }
    }
  }

  void holes(const Term& t, std::vector<int64_t> values) {  // `var {1,2,4,5}: x`
    std::sort(values.begin(), values.end());
    values.erase(std::unique(values.begin(), values.end()), values.end());
    for (size_t i = 0; i + 1 < values.size(); ++i)
      for (int64_t h = values[i] + 1; h < values[i + 1]; ++h) {
        if (values[i + 1] - values[i] > 4096) fail("domain with a hole wider than 4096 values");
        if (t.is_const) { if (t.value == h) t_.trivially_unsat = true; }
        else prop(TB_EQ, ZERO, t.var, konst(h));
      }
  }

  // ---------------------------------------------------------------- building blocks
  int32_t scaled(int64_t coef, int32_t v) {  // coef * v, coef > 0
    if (coef == 1) return v;
    tb_itv D = dom(v);
    int32_t c = clamp64(coef);
    int32_t p0 = mul_ext(c, D.lb), p1 = mul_ext(c, D.ub);
    int32_t s = new_var(std::min(p0, p1), std::max(p0, p1));
    prop(TB_MUL, s, konst(coef), v);
    return s;
  }
  // balanced reduction; if `target` >= 0 the root of the tree is `target`
  int32_t tree(int32_t op, std::vector<int32_t> vs, int32_t target = -1) {
    if (vs.empty()) fail("internal: empty reduction");
    auto node = [&](int32_t a, int32_t b) {
      tb_itv A = dom(a), B = dom(b);
      int32_t r;
      if (op == TB_ADD) r = new_var(add_lo(A.lb, B.lb), add_hi(A.ub, B.ub));
      else if (op == TB_MIN) r = new_var(std::min(A.lb, B.lb), std::min(A.ub, B.ub));
      else r = new_var(std::max(A.lb, B.lb), std::max(A.ub, B.ub));
      prop(op, r, a, b);
      return r;
    };
    if (vs.size() == 1) {
      if (target >= 0 && target != vs[0]) prop(TB_EQ, ONE, target, vs[0]);
      return target >= 0 ? target : vs[0];
    }
    while (vs.size() > 2) {
      std::vector<int32_t> next;
      for (size_t i = 0; i + 1 < vs.size(); i += 2) next.push_back(node(vs[i], vs[i + 1]));
      if (vs.size() % 2) next.push_back(vs.back());
      vs.swap(next);
    }
    if (target < 0) return node(vs[0], vs[1]);
    prop(op, target, vs[0], vs[1]);
    return target;
  }

  struct Lin { int32_t var = -1; int64_t k = 0; };  // value = var + k  (var == -1: the constant k)

  Lin linear(const std::vector<int64_t>& coefs, const std::vector<Term>& xs) {
    if (coefs.size() != xs.size()) fail("linear constraint with mismatched array sizes");
    Lin out;
    std::vector<std::pair<int32_t, int64_t>> terms;  // (var, coef), first-occurrence order
    std::map<int32_t, size_t> pos;
    for (size_t i = 0; i < xs.size(); ++i) {
      if (coefs[i] == 0) continue;
      if (xs[i].is_const) { out.k += coefs[i] * xs[i].value; continue; }
      auto it = pos.find(xs[i].var);
      if (it == pos.end()) { pos[xs[i].var] = terms.size(); terms.push_back({xs[i].var, coefs[i]}); }
      else terms[it->second].second += coefs[i];
    }
    std::vector<int32_t> P, N;
    for (auto& tc : terms) {
      if (tc.second > 0) P.push_back(scaled(tc.second, tc.first));
      else if (tc.second < 0) N.push_back(scaled(-tc.second, tc.first));
    }
    int32_t p = P.empty() ? -1 : tree(TB_ADD, P);
    int32_t n = N.empty() ? -1 : tree(TB_ADD, N);
    if (p < 0 && n < 0) return out;
    if (n < 0) { out.var = p; return out; }
    tb_itv Nd = dom(n);
    if (p < 0) {  // S = -N  <=>  0 = S + N
      out.var = new_var(neg_ext(Nd.ub), neg_ext(Nd.lb));
      prop(TB_ADD, ZERO, out.var, n);
      return out;
    }
    tb_itv Pd = dom(p);  // S = P - N  <=>  P = S + N
    out.var = new_var(add_lo(Pd.lb, neg_ext(Nd.ub)), add_hi(Pd.ub, neg_ext(Nd.lb)));
    prop(TB_ADD, p, out.var, n);
    return out;
  }

  void set_truth(const Term* r, bool truth) {
    if (r == nullptr) { if (!truth) t_.trivially_unsat = true; }
    else restrict(*r, truth ? 1 : 0, truth ? 1 : 0);
  }

  // r <=> (a == b), hash-consed
  int32_t reif_eq(int32_t a, int32_t b) {
    std::pair<int32_t, int32_t> key = std::minmax(a, b);
    auto it = eq_cache_.find(key);
    if (it != eq_cache_.end()) return it->second;
    tb_itv A = dom(a), B = dom(b);
    int32_t r;
    if (A.ub < B.lb || A.lb > B.ub) r = ZERO;
    else if (A.lb == A.ub && B.lb == B.ub) r = ONE;
    else { r = new_bool(); prop(TB_EQ, r, a, b); }
    eq_cache_[key] = r;
    return r;
  }

  int32_t negation(int32_t b) {  // nb = (b == 0)
    return reif_eq(b, ZERO);
  }

  enum Cmp { EQ, NE, LE, LT, GE, GT };

  // compare `S + k` (a linear form) with the constant c
  void lin_cmp(Cmp cmp, std::vector<int64_t> coefs, const std::vector<Term>& xs, int64_t c, const Term* r) {
    if (cmp == GE || cmp == GT) {  // sum >= c  <=>  -sum <= -c
      for (auto& a : coefs) a = -a;
      c = -c;
      cmp = (cmp == GE) ? LE : LT;
    }
    if (cmp == LT) { cmp = LE; c -= 1; }
    Lin lin = linear(coefs, xs);
    int64_t rhs = c - lin.k;
    if (lin.var < 0) {
      bool truth = cmp == EQ ? (0 == rhs) : (cmp == NE ? (0 != rhs) : (0 <= rhs));
      set_truth(r, truth);
      return;
    }
    int32_t S = lin.var;
    if (rhs > FIN_MAX || rhs < FIN_MIN) {  // outside the representable range: constant truth value
      bool truth = cmp == EQ ? false : (cmp == NE ? true : (rhs > FIN_MAX));
      set_truth(r, truth);
      return;
    }
    switch (cmp) {
      case LE:
        if (!r) restrict(S, NINF, rhs);
        else prop(TB_LEQ, as_var(*r), S, konst(rhs));
        break;
      case EQ:
        if (!r) restrict(S, rhs, rhs);
        else prop(TB_EQ, as_var(*r), S, konst(rhs));
        break;
      case NE:
        if (!r) prop(TB_EQ, ZERO, S, konst(rhs));
        else prop(TB_EQ, as_var(*r), reif_eq(S, konst(rhs)), ZERO);
        break;
      default: break;
    }
  }

  void cmp2(Cmp cmp, const Term& a, const Term& b, const Term* r) {
    if (cmp == GE) return cmp2(LE, b, a, r);
    if (cmp == GT) return cmp2(LT, b, a, r);
    if (a.is_const && b.is_const) {
      bool truth = cmp == EQ ? a.value == b.value : cmp == NE ? a.value != b.value : cmp == LE ? a.value <= b.value : a.value < b.value;
      set_truth(r, truth);
      return;
    }
    if (!r) {
      switch (cmp) {
        case EQ:
          if (a.is_const) restrict(b, a.value, a.value);
          else if (b.is_const) restrict(a, b.value, b.value);
          else prop(TB_EQ, ONE, a.var, b.var);
          return;
        case NE: prop(TB_EQ, ZERO, as_var(a), as_var(b)); return;
        case LE:
          if (a.is_const) restrict(b, a.value, PINF);
          else if (b.is_const) restrict(a, NINF, b.value);
          else prop(TB_LEQ, ONE, a.var, b.var);
          return;
        case LT:
          if (a.is_const) restrict(b, a.value + 1, PINF);
          else if (b.is_const) restrict(a, NINF, b.value - 1);
          else prop(TB_LEQ, ZERO, b.var, a.var);  // a < b  <=>  not (b <= a)
          return;
        default: return;
      }
    }
    int32_t rv = as_var(*r);
    switch (cmp) {
      case EQ: prop(TB_EQ, rv, as_var(a), as_var(b)); return;
      case NE: prop(TB_EQ, rv, reif_eq(as_var(a), as_var(b)), ZERO); return;
      case LE: prop(TB_LEQ, rv, as_var(a), as_var(b)); return;
      case LT:
        if (b.is_const) { prop(TB_LEQ, rv, as_var(a), konst(b.value - 1)); return; }
        if (a.is_const) { prop(TB_LEQ, rv, konst(a.value + 1), as_var(b)); return; }
        {
          int32_t nb = new_bool();  // nb = (b <= a); r = not nb
          prop(TB_LEQ, nb, b.var, a.var);
          prop(TB_EQ, rv, nb, ZERO);
        }
        return;
      default: return;
    }
  }

  // r <=> x in S
  void set_in(const Term& x, const Ranges& S, const Term* r) {
    if (S.empty()) { set_truth(r, false); return; }
    if (x.is_const) {
      bool in = false;
      for (auto& rg : S) in |= (x.value >= rg.first && x.value <= rg.second);
      set_truth(r, in);
      return;
    }
    if (!r) {
      restrict(x, S.front().first, S.back().second);
      for (size_t i = 0; i + 1 < S.size(); ++i) {
        if (S[i + 1].first - S[i].second > 4096) fail("set_in with a hole wider than 4096 values");
        for (int64_t h = S[i].second + 1; h < S[i + 1].first; ++h) prop(TB_EQ, ZERO, x.var, konst(h));
      }
      return;
    }
    std::vector<int32_t> ins;
    for (auto& rg : S) {
      if (rg.first == rg.second) { ins.push_back(reif_eq(x.var, konst(rg.first))); continue; }
      int32_t l = new_bool(), u = new_bool(), in = new_bool();
      prop(TB_LEQ, l, konst(rg.first), x.var);
      prop(TB_LEQ, u, x.var, konst(rg.second));
      prop(TB_MIN, in, l, u);
      ins.push_back(in);
    }
    tree(TB_MAX, ins, as_var(*r));
  }

  void element(const Term& idx, const std::vector<Term>& arr, const Term& x) {
    int64_t n = (int64_t)arr.size();
    restrict(idx, 1, n);
    tb_itv I = dom(idx);
    if (I.lb > I.ub) { t_.trivially_unsat = true; return; }
    int32_t lo = PINF, hi = NINF;  // x lies in the hull of the reachable entries
    for (int64_t i = I.lb; i <= I.ub; ++i) { tb_itv D = dom(arr[(size_t)i - 1]); lo = std::min(lo, D.lb); hi = std::max(hi, D.ub); }
    restrict(x, lo, hi);
    if (idx.is_const) { cmp2(EQ, x, arr[(size_t)idx.value - 1], nullptr); return; }
    int32_t xv = as_var(x);
    for (int64_t i = I.lb; i <= I.ub; ++i) {  // (idx = i) -> (x = arr[i])
      int32_t bi = reif_eq(idx.var, konst(i));
      int32_t ci = reif_eq(xv, as_var(arr[(size_t)i - 1]));
      if (bi == ZERO || ci == ONE) continue;
      if (ci == ZERO) { restrict(bi, 0, 0); continue; }  // idx != i
      prop(TB_LEQ, ONE, bi, ci);
    }
  }

  std::vector<int32_t> vars_of(const std::vector<Term>& ts) {
    std::vector<int32_t> vs;
    vs.reserve(ts.size());
    for (const Term& t : ts) vs.push_back(as_var(t));
    return vs;
  }

  // ---------------------------------------------------------------- predicates
  static bool ends_with(const std::string& s, const char* suf) {
    size_t n = std::char_traits<char>::length(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
  }

  // Post predicate `name(args)`; r == nullptr: it must hold; otherwise r <=> name(args).
  void post_pred(std::string name, std::vector<Expr>& args, const Term* r) {
    Term rt;
    if (ends_with(name, "_reif")) {
      if (r) fail("nested reified predicate `" + name + "`");
      name.resize(name.size() - 5);
      rt = resolve(args.back());
      args.pop_back();
      r = &rt;
    } else if (ends_with(name, "_imp")) {  // half reification: b -> p
      name.resize(name.size() - 4);
      Term b = resolve(args.back());
      args.pop_back();
      int32_t full = new_bool();
      Term ft = Term::variable(full);
      post_pred(name, args, &ft);
      prop(TB_LEQ, ONE, as_var(b), full);
      return;
    }
    auto need = [&](size_t n) { if (args.size() != n) fail("predicate `" + name + "` expects " + std::to_string(n) + " arguments"); };

    // ---- comparisons
    static const std::map<std::string, Cmp> cmp2_names = {
        {"int_eq", EQ}, {"int_ne", NE}, {"int_le", LE}, {"int_lt", LT}, {"int_ge", GE}, {"int_gt", GT},
        {"bool_eq", EQ}, {"bool_ne", NE}, {"bool_le", LE}, {"bool_lt", LT}, {"bool2int", EQ}};
    auto c2 = cmp2_names.find(name);
    if (c2 != cmp2_names.end()) { need(2); Term a = resolve(args[0]), b = resolve(args[1]); cmp2(c2->second, a, b, r); return; }
    static const std::map<std::string, Cmp> lin_names = {
        {"int_lin_eq", EQ}, {"int_lin_ne", NE}, {"int_lin_le", LE}, {"int_lin_lt", LT}, {"int_lin_ge", GE}, {"int_lin_gt", GT},
        {"bool_lin_eq", EQ}, {"bool_lin_le", LE}};
    auto cl = lin_names.find(name);
    if (cl != lin_names.end()) {
      need(3);
      std::vector<int64_t> coefs = const_array(args[0]);
      std::vector<Term> xs = resolve_array(args[1]);
      Term c = resolve(args[2]);
      if (!c.is_const) {  // sum - c  cmp  0
        coefs.push_back(-1); xs.push_back(c);
        lin_cmp(cl->second, coefs, xs, 0, r);
      } else lin_cmp(cl->second, coefs, xs, c.value, r);
      return;
    }
    if (name == "set_in") { need(2); Term x = resolve(args[0]); set_in(x, resolve_set(args[1]), r); return; }

    // ---- Boolean connectives with a result argument (functional, not reified)
    if (name == "bool_not") {
      need(2);
      Term a = resolve(args[0]), b = resolve(args[1]);
      if (r) { cmp2(NE, a, b, r); return; }
      if (a.is_const) { restrict(b, 1 - a.value, 1 - a.value); return; }
      if (b.is_const) { restrict(a, 1 - b.value, 1 - b.value); return; }
      prop(TB_EQ, b.var, a.var, ZERO);
      return;
    }
    if (name == "bool_clause") {
      need(2);
      std::vector<int32_t> pos = vars_of(resolve_array(args[0])), neg = vars_of(resolve_array(args[1]));
      if (r) {
        int32_t P = pos.empty() ? ZERO : tree(TB_MAX, pos), N = neg.empty() ? ONE : tree(TB_MIN, neg);
        prop(TB_LEQ, as_var(*r), N, P);
        return;
      }
      if (pos.empty() && neg.empty()) { t_.trivially_unsat = true; return; }
      if (neg.empty()) { if (pos.size() == 1) restrict(pos[0], 1, 1); else tree(TB_MAX, pos, ONE); return; }
      if (pos.empty()) { if (neg.size() == 1) restrict(neg[0], 0, 0); else tree(TB_MIN, neg, ZERO); return; }
      prop(TB_LEQ, ONE, tree(TB_MIN, neg), tree(TB_MAX, pos));  // not (all pos false and all neg true)
      return;
    }
    if (r) fail("predicate `" + name + "` cannot be reified by this front-end");

    if (name == "bool_and" || name == "bool_or" || name == "int_min" || name == "int_max") {
      need(3);
      Term a = resolve(args[0]), b = resolve(args[1]), c = resolve(args[2]);
      prop((name == "bool_and" || name == "int_min") ? TB_MIN : TB_MAX, as_var(c), as_var(a), as_var(b));
      return;
    }
    if (name == "bool_xor") {
      Term a = resolve(args[0]), b = resolve(args[1]);
      Term c = args.size() == 3 ? resolve(args[2]) : Term::konst(1);
      if (c.is_const) cmp2(c.value ? NE : EQ, a, b, nullptr);
      else cmp2(NE, a, b, &c);
      return;
    }
    if (name == "array_bool_and" || name == "array_bool_or") {
      need(2);
      std::vector<int32_t> as = vars_of(resolve_array(args[0]));
      Term c = resolve(args[1]);
      bool is_and = name == "array_bool_and";
      if (as.empty()) { restrict(c, is_and ? 1 : 0, is_and ? 1 : 0); return; }
      tree(is_and ? TB_MIN : TB_MAX, as, as_var(c));
      return;
    }
    if (name == "array_bool_xor") {  // odd parity: sum = 2k + 1
      need(1);
      std::vector<Term> as = resolve_array(args[0]);
      std::vector<int64_t> ones(as.size(), 1);
      Lin s = linear(ones, as);
      if (s.var < 0) { if ((s.k & 1) == 0) t_.trivially_unsat = true; return; }
      int64_t n = (int64_t)as.size();
      int32_t k = new_var(-n, n), twok = new_var(-2 * n, 2 * n);
      prop(TB_MUL, twok, TWO, k);
      prop(TB_ADD, s.var, twok, konst(1 - s.k));  // s.var + s.k = 2k + 1
      return;
    }

    // ---- arithmetic
    if (name == "int_plus" || name == "int_minus" || name == "int_times" || name == "int_div" || name == "int_mod") {
      need(3);
      int32_t a = as_var(resolve(args[0])), b = as_var(resolve(args[1])), c = as_var(resolve(args[2]));
      if (name == "int_plus") prop(TB_ADD, c, a, b);
      else if (name == "int_minus") prop(TB_ADD, a, c, b);  // c = a - b  <=>  a = c + b
      else if (name == "int_times") prop(TB_MUL, c, a, b);
      else if (name == "int_div") prop(TB_TDIV, c, a, b);
      else prop(TB_TMOD, c, a, b);
      return;
    }
    if (name == "int_abs") {
      need(2);
      int32_t a = as_var(resolve(args[0])), b = as_var(resolve(args[1]));
      tb_itv A = dom(a);
      int32_t na = new_var(neg_ext(A.ub), neg_ext(A.lb));
      prop(TB_ADD, ZERO, a, na);
      prop(TB_MAX, b, a, na);
      restrict(b, 0, PINF);
      return;
    }
    if (name == "int_negate") {
      need(2);
      prop(TB_ADD, ZERO, as_var(resolve(args[0])), as_var(resolve(args[1])));
      return;
    }

    // ---- element
    if (name == "array_int_element" || name == "array_bool_element" || name == "array_var_int_element" || name == "array_var_bool_element") {
      need(3);
      Term idx = resolve(args[0]);
      std::vector<Term> arr = resolve_array(args[1]);
      Term x = resolve(args[2]);
      element(idx, arr, x);
      return;
    }
    fail("unsupported predicate `" + name + "`");
  }

  // ---------------------------------------------------------------- solve item
  void goal() {
    t_.goal = (int32_t)m_.solve.goal;
    if (m_.solve.goal == Solve::SATISFY) return;
    Term o = resolve(m_.solve.objective);
    int32_t ov = as_var(o);
    t_.goal_var = ov;
    if (m_.solve.goal == Solve::MINIMIZE) { t_.obj_var = ov; return; }
    // maximize x  ->  minimize __MINIMIZE_OBJ with __MINIMIZE_OBJ = -x   (common_solving.hpp:489-510)
    tb_itv D = dom(ov);
    int32_t mo = new_var(neg_ext(D.ub), neg_ext(D.lb), "__MINIMIZE_OBJ");
    prop(TB_ADD, ZERO, ov, mo);
    t_.obj_var = mo;
  }

  void add_search(const Expr& a) {
    if (a.kind != Expr::CALL) return;
    if (a.name == "seq_search") {
      if (a.args.size() == 1 && a.args[0].kind == Expr::ARRAY)
        for (const Expr& s : a.args[0].args) add_search(s);
      return;
    }
    if (a.name != "int_search" && a.name != "bool_search") return;  // other annotations are ignored
    if (a.args.size() < 3) fail("search annotation `" + a.name + "` expects at least 3 arguments");
    Strategy s;
    s.vars = vars_of(resolve_array(a.args[0]));
    const std::string& vo = a.args[1].name;
    const std::string& vl = a.args[2].name;
    if (vo == "input_order") s.var_order = TB_INPUT_ORDER;
    else if (vo == "first_fail") s.var_order = TB_FIRST_FAIL;
    else if (vo == "anti_first_fail") s.var_order = TB_ANTI_FIRST_FAIL;
    else if (vo == "smallest") s.var_order = TB_SMALLEST;
    else if (vo == "largest") s.var_order = TB_LARGEST;
    else fail("unsupported variable order `" + vo + "`");
    if (vl == "indomain_min" || vl == "indomain") s.val_order = TB_VAL_MIN;
    else if (vl == "indomain_max") s.val_order = TB_VAL_MAX;
    else if (vl == "indomain_split") s.val_order = TB_VAL_SPLIT;
    else if (vl == "indomain_reverse_split") s.val_order = TB_VAL_REVERSE_SPLIT;
    else fail("unsupported value order `" + vl + "`");
    if (s.vars.empty()) return;  // an empty list would mean "whole store" to the engine
    t_.strategies.push_back(std::move(s));
  }

  void search_annotations() {
    for (const Expr& a : m_.solve.anns) add_search(a);
  }
};

}  // namespace

void TCN::flatten_strategies() {
  f_var_order.clear(); f_val_order.clear(); f_off.clear(); f_vars.clear();
  f_off.push_back(0);
  for (const Strategy& s : strategies) {
    f_var_order.push_back(s.var_order);
    f_val_order.push_back(s.val_order);
    f_vars.insert(f_vars.end(), s.vars.begin(), s.vars.end());
    f_off.push_back((int32_t)f_vars.size());
  }
  if (f_vars.empty()) f_vars.push_back(0);  // keep data() non-null
}

TCN lower_to_tcn(const Model& m) {
  Lowerer l(m);
  TCN t = l.run();
  // A constraint that is false on constants leaves no propagator behind; besides the flag, make the network itself
  // unsatisfiable for callers that only look at the arrays: `0 = (0 = 0)` over the interned constant 0.
  if (t.trivially_unsat && !t.store.empty()) t.props.push_back(tb_prop{TB_EQ, 0, 0, 0});
  return t;
}

}  // namespace turbo_front
