// XCSP3-core reader: XML instance -> FlatZinc text, which then goes through the same parser and lowering as a
// .fzn input (one lowering path).  Stands in for lala-parsing's `parse_xcsp3` (common_solving.hpp:409-413; the
// library is absent from the reference tree), for the constraint forms listed below.
//
//   variables    <var>, <array> (uniform domain: ranges and value lists; multi-dimensional sizes)
//   variables    ... arrays with per-cell <domain for="..."> (and for="others")
//   constraints  <intension>, <extension> (supports / conflicts, `*`), <regular>, <mdd>, <allDifferent> (also <except>, several
//                lists, <matrix>), <allEqual>, <ordered> (also <lengths>), <lex> (lists or <matrix>), <sum> (constant or variable
//                coefficients), <count>, <nValues> (also <except>), <cardinality> (constant, interval or variable occurrences,
//                closed or not), <minimum>, <maximum>, <element> (list or <matrix>), <channel> (one or two lists), <noOverlap> (one or
//                several dimensions), <cumulative> (constant or variable lengths and heights, optional <ends>), <binPacking>
//                (<condition>, <limits> or <loads>), <knapsack>, <circuit> (sub-circuit semantics, optional <size>), <instantiation>,
//                <clause>, <slide>, <group> with %i and %... arguments, <block>; conditions with a value, a variable, an interval (in / notin
//                a..b) or a set (in / notin {..})
//   objectives   <minimize> / <maximize> of type expression, sum, product, minimum, maximum, nValues (optional <coeffs>)
//
// Decompositions are the textbook ones (time-indexed cumulative, pairwise allDifferent, tuple-wise tables); the
// only thing the reference pins for this format is the objective of benchmarks/test_data/cumulative.xml.
#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <sstream>

#include "fzn_ast.hpp"
#include "tcn.hpp"

namespace turbo_front {
namespace {

[[noreturn]] void fail(const std::string& m) { throw ParseError("XCSP3: " + m); }

// ---- a very small XML reader (elements, attributes, text; comments and <? ?> are skipped) -------------------
struct Xml {
  std::string name, text;
  std::map<std::string, std::string> attr;
  std::vector<std::unique_ptr<Xml>> kids;
  const Xml* child(const std::string& n) const {
    for (auto& k : kids) if (k->name == n) return k.get();
    return nullptr;
  }
  std::string get(const std::string& a, const std::string& dflt = "") const {
    auto it = attr.find(a);
    return it == attr.end() ? dflt : it->second;
  }
};

struct XmlParser {
  const std::string& s;
  size_t p = 0;
  explicit XmlParser(const std::string& t) : s(t) {}
  void skip_misc() {
    for (;;) {
      while (p < s.size() && std::isspace((unsigned char)s[p])) ++p;
      if (s.compare(p, 4, "<!--") == 0) { size_t e = s.find("-->", p); if (e == std::string::npos) fail("unterminated comment"); p = e + 3; }
      else if (s.compare(p, 2, "<?") == 0) { size_t e = s.find("?>", p); if (e == std::string::npos) fail("unterminated declaration"); p = e + 2; }
      else if (s.compare(p, 2, "<!") == 0) { size_t e = s.find('>', p); if (e == std::string::npos) fail("unterminated declaration"); p = e + 1; }
      else return;
    }
  }
  static std::string unescape(std::string t) {
    static const std::pair<const char*, const char*> ents[] = {{"&lt;", "<"}, {"&gt;", ">"}, {"&amp;", "&"}, {"&quot;", "\""}, {"&apos;", "'"}};
    for (auto& e : ents) for (size_t q; (q = t.find(e.first)) != std::string::npos;) t.replace(q, std::strlen(e.first), e.second);
    return t;
  }
  std::unique_ptr<Xml> element() {
    skip_misc();
    if (p >= s.size() || s[p] != '<') fail("expected an element");
    ++p;
    auto n = std::make_unique<Xml>();
    while (p < s.size() && !std::isspace((unsigned char)s[p]) && s[p] != '>' && s[p] != '/') n->name += s[p++];
    for (;;) {
      while (p < s.size() && std::isspace((unsigned char)s[p])) ++p;
      if (p >= s.size()) fail("unterminated tag <" + n->name);
      if (s[p] == '/') { if (p + 1 >= s.size() || s[p + 1] != '>') fail("bad tag end"); p += 2; return n; }
      if (s[p] == '>') { ++p; break; }
      std::string a;
      while (p < s.size() && s[p] != '=' && !std::isspace((unsigned char)s[p])) a += s[p++];
      while (p < s.size() && std::isspace((unsigned char)s[p])) ++p;
      if (p >= s.size() || s[p] != '=') fail("attribute `" + a + "` without value");
      ++p;
      while (p < s.size() && std::isspace((unsigned char)s[p])) ++p;
      const char q = s[p];
      if (q != '"' && q != '\'') fail("attribute value must be quoted");
      size_t e = s.find(q, p + 1);
      if (e == std::string::npos) fail("unterminated attribute value");
      n->attr[a] = unescape(s.substr(p + 1, e - p - 1));
      p = e + 1;
    }
    for (;;) {  // content
      size_t lt = s.find('<', p);
      if (lt == std::string::npos) fail("missing </" + n->name + ">");
      n->text += unescape(s.substr(p, lt - p));
      n->text += ' ';
      p = lt;
      if (s.compare(p, 4, "<!--") == 0) { skip_misc(); continue; }
      if (s.compare(p, 2, "</") == 0) {
        size_t e = s.find('>', p);
        if (e == std::string::npos) fail("unterminated end tag");
        p = e + 1;
        return n;
      }
      n->kids.push_back(element());
    }
  }
};

std::vector<std::string> words(const std::string& t) {
  std::vector<std::string> out;
  std::istringstream is(t);
  for (std::string w; is >> w;) out.push_back(w);
  return out;
}

bool is_int(const std::string& w) {
  if (w.empty()) return false;
  size_t i = (w[0] == '-' || w[0] == '+') ? 1 : 0;
  if (i >= w.size()) return false;
  for (; i < w.size(); ++i) if (!std::isdigit((unsigned char)w[i])) return false;
  return true;
}

// "1 3 5..7 2x3" -> integers (ranges expanded, `vxk` = v repeated k times)
std::vector<int64_t> int_list(const std::string& t) {
  std::vector<int64_t> out;
  for (const std::string& w : words(t)) {
    size_t dd = w.find("..");
    size_t xx = w.find('x');
    if (dd != std::string::npos) {
      const int64_t a = std::stoll(w.substr(0, dd)), b = std::stoll(w.substr(dd + 2));
      if (b - a > 10000000) fail("range too large in a value list");
      for (int64_t v = a; v <= b; ++v) out.push_back(v);
    } else if (xx != std::string::npos && xx > 0) {
      const int64_t v = std::stoll(w.substr(0, xx)), k = std::stoll(w.substr(xx + 1));
      for (int64_t i = 0; i < k; ++i) out.push_back(v);
    } else if (is_int(w)) out.push_back(std::stoll(w));
    else fail("expected an integer, got `" + w + "`");
  }
  return out;
}

// A value of the translation: a constant or a FlatZinc variable name with known bounds.
struct Val {
  bool is_const = true, is_bool = false;
  int64_t c = 0;
  std::string var;
  int64_t lo = 0, hi = 0;
  std::string str() const { return is_const ? std::to_string(c) : var; }
};
Val konst(int64_t c) { Val v; v.c = c; v.lo = v.hi = c; return v; }

struct ArrayInfo {
  std::vector<int64_t> dims;
  std::vector<std::string> elems;  // flattened FlatZinc names
};

struct Translator {
  std::ostringstream decl, cons;
  std::string solve = "solve satisfy;\n";
  std::map<std::string, Val> vars;
  std::map<std::string, ArrayInfo> arrays;
  int tmp = 0;

  static int64_t sat(__int128 v) { return v > INT32_MAX ? INT32_MAX : (v < INT32_MIN ? INT32_MIN : (int64_t)v); }

  Val fresh_int(int64_t lo, int64_t hi) {
    Val v; v.is_const = false; v.var = "X_T" + std::to_string(tmp++); v.lo = sat(lo); v.hi = sat(hi);
    decl << "var " << v.lo << ".." << v.hi << ": " << v.var << ";\n";
    return v;
  }
  Val fresh_bool() {
    Val v; v.is_const = false; v.is_bool = true; v.var = "X_B" + std::to_string(tmp++); v.lo = 0; v.hi = 1;
    decl << "var bool: " << v.var << ";\n";
    return v;
  }

  // ---- variables
  void domain_text(const std::string& t, std::string* fz, int64_t* lo, int64_t* hi) {
    const auto ws = words(t);
    if (ws.empty()) fail("empty domain");
    if (ws.size() == 1 && ws[0].find("..") != std::string::npos) {
      const size_t dd = ws[0].find("..");
      *lo = std::stoll(ws[0].substr(0, dd)); *hi = std::stoll(ws[0].substr(dd + 2));
      *fz = std::to_string(*lo) + ".." + std::to_string(*hi);
      return;
    }
    std::vector<int64_t> vals = int_list(t);
    std::sort(vals.begin(), vals.end());
    vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
    *lo = vals.front(); *hi = vals.back();
    if ((int64_t)vals.size() == *hi - *lo + 1) { *fz = std::to_string(*lo) + ".." + std::to_string(*hi); return; }
    *fz = "{";
    for (size_t i = 0; i < vals.size(); ++i) *fz += (i ? "," : "") + std::to_string(vals[i]);
    *fz += "}";
  }
  static std::string flat_name(std::string id) {
    for (char& c : id) if (!std::isalnum((unsigned char)c) && c != '_') c = '_';
    return id;
  }
  void variables(const Xml& vs) {
    for (auto& k : vs.kids) {
      const std::string id = k->get("id");
      if (id.empty()) fail("variable without id");
      if (!k->get("as").empty()) fail("`as` attribute is not supported");
      if (k->name == "var") {
        std::string fz; int64_t lo, hi;
        domain_text(k->text, &fz, &lo, &hi);
        Val v; v.is_const = false; v.var = flat_name(id); v.lo = lo; v.hi = hi;
        decl << "var " << fz << ": " << v.var << " :: output_var;\n";
        vars[id] = v;
        id_of[v.var] = id;
      } else if (k->name == "array") {
        ArrayInfo info;
        const std::string sz = k->get("size");
        for (size_t i = 0; i < sz.size(); ++i)
          if (sz[i] == '[') { size_t e = sz.find(']', i); info.dims.push_back(std::stoll(sz.substr(i + 1, e - i - 1))); i = e; }
        if (info.dims.empty()) fail("array without size");
        int64_t total = 1;
        for (int64_t d : info.dims) total *= d;
        if (total <= 0 || total > 10000000) fail("bad array size");
        std::string fz; int64_t lo, hi;
        // per-cell domains (<domain for="x[0] x[2..3]"> ... </domain>, for="others"): the array is declared over the hull, every cell
        // gets its own set_in
        std::vector<std::pair<std::string, std::string>> cell_domains;  // (for, domain text)
        if (!k->kids.empty()) {
          lo = INT64_MAX; hi = INT64_MIN;
          for (auto& dkid : k->kids) {
            if (dkid->name != "domain") fail("unknown element <" + dkid->name + "> in <array>");
            std::string dfz; int64_t dlo, dhi;
            domain_text(dkid->text, &dfz, &dlo, &dhi);
            lo = std::min(lo, dlo); hi = std::max(hi, dhi);
            cell_domains.push_back({dkid->get("for"), dkid->text});
          }
          fz = std::to_string(lo) + ".." + std::to_string(hi);
        } else domain_text(k->text, &fz, &lo, &hi);
        const std::string base = flat_name(id);
        const bool holes = fz[0] == '{';
        decl << "array [1.." << total << "] of var " << (holes ? std::to_string(lo) + ".." + std::to_string(hi) : fz) << ": " << base << " :: output_array([";
        for (size_t d = 0; d < info.dims.size(); ++d) decl << (d ? ", " : "") << "0.." << info.dims[d] - 1;
        decl << "]);\n";
        for (int64_t i = 0; i < total; ++i) {
          const std::string en = base + "[" + std::to_string(i + 1) + "]";
          info.elems.push_back(en);
          // canonical id of the cell: x[i][j]
          std::string cid = id;
          int64_t rem = i;
          std::vector<int64_t> idx(info.dims.size());
          for (size_t d = info.dims.size(); d-- > 0;) { idx[d] = rem % info.dims[d]; rem /= info.dims[d]; }
          for (int64_t q : idx) cid += "[" + std::to_string(q) + "]";
          Val v; v.is_const = false; v.var = en; v.lo = lo; v.hi = hi;
          vars[cid] = v;
          id_of[en] = cid;
          if (holes) cons << "constraint set_in(" << en << ", " << fz << ");\n";
        }
        arrays[id] = info;
        if (!cell_domains.empty()) {
          std::map<std::string, bool> done;
          const std::pair<std::string, std::string>* others = nullptr;
          for (auto& cd : cell_domains) {
            if (cd.first == "others") { others = &cd; continue; }
            std::string dfz; int64_t dlo, dhi;
            domain_text(cd.second, &dfz, &dlo, &dhi);
            for (const std::string& w : words(cd.first)) {
              std::vector<Val> cells; expand_ref(w, &cells);
              for (auto& c : cells) { done[c.var] = true; narrow_cell(c, dfz, dlo, dhi); }
            }
          }
          for (auto& en : info.elems) {
            if (done.count(en)) continue;
            if (!others) fail("array `" + id + "`: a cell has no domain");
            std::string dfz; int64_t dlo, dhi;
            domain_text(others->second, &dfz, &dlo, &dhi);
            narrow_cell(vars_by_elem(en), dfz, dlo, dhi);
          }
        }
      } else fail("unknown declaration <" + k->name + ">");
    }
  }

  // a cell of an array declared over the hull of its cells' domains gets its own domain
  void narrow_cell(const Val& c, const std::string& dfz, int64_t dlo, int64_t dhi) {
    if (dfz[0] == '{') cons << "constraint set_in(" << c.var << ", " << dfz << ");\n";
    else cons << "constraint int_le(" << dlo << ", " << c.var << ");\nconstraint int_le(" << c.var << ", " << dhi << ");\n";
    // (looked up by name: one scan of `vars` per cell made per-cell array domains quadratic)
    auto id = id_of.find(c.var);
    if (id != id_of.end()) { Val& v = vars[id->second]; v.lo = dlo; v.hi = dhi; auto e = elem_cache.find(c.var); if (e != elem_cache.end()) e->second = v; }
  }
  // "x[]" "x[1..3]" "x[][0]" "y" "3" -> values
  void expand_ref(const std::string& w, std::vector<Val>* out) {
    if (is_int(w)) { out->push_back(konst(std::stoll(w))); return; }
    if (w.find("..") != std::string::npos && w.find('[') == std::string::npos) { for (int64_t v : int_list(w)) out->push_back(konst(v)); return; }
    if (w.find('x') != std::string::npos && std::isdigit((unsigned char)w[0]) && w.find('[') == std::string::npos) { for (int64_t v : int_list(w)) out->push_back(konst(v)); return; }
    const size_t br = w.find('[');
    if (br == std::string::npos) {
      auto it = vars.find(w);
      if (it == vars.end()) { auto at = arrays.find(w); if (at != arrays.end()) { for (auto& e : at->second.elems) out->push_back(vars_by_elem(e)); return; } fail("unknown variable `" + w + "`"); }
      out->push_back(it->second);
      return;
    }
    const std::string id = w.substr(0, br);
    auto at = arrays.find(id);
    if (at == arrays.end()) fail("unknown array `" + id + "`");
    const ArrayInfo& a = at->second;
    std::vector<std::pair<int64_t, int64_t>> sel;
    for (size_t i = br; i < w.size();) {
      if (w[i] != '[') fail("bad index in `" + w + "`");
      size_t e = w.find(']', i);
      if (e == std::string::npos) fail("bad index in `" + w + "`");
      const std::string in = w.substr(i + 1, e - i - 1);
      const size_t d = sel.size();
      if (d >= a.dims.size()) fail("too many indices in `" + w + "`");
      if (in.empty()) sel.push_back({0, a.dims[d] - 1});
      else if (in.find("..") != std::string::npos) { size_t dd = in.find(".."); sel.push_back({std::stoll(in.substr(0, dd)), std::stoll(in.substr(dd + 2))}); }
      else sel.push_back({std::stoll(in), std::stoll(in)});
      i = e + 1;
    }
    while (sel.size() < a.dims.size()) sel.push_back({0, a.dims[sel.size()] - 1});
    std::vector<int64_t> idx(sel.size());
    for (size_t d = 0; d < sel.size(); ++d) { idx[d] = sel[d].first; if (sel[d].first < 0 || sel[d].second >= a.dims[d]) fail("index out of range in `" + w + "`"); }
    for (;;) {
      int64_t flat = 0;
      for (size_t d = 0; d < idx.size(); ++d) flat = flat * a.dims[d] + idx[d];
      out->push_back(vars_by_elem(a.elems[(size_t)flat]));
      size_t d = idx.size();
      while (d-- > 0) { if (++idx[d] <= sel[d].second) break; idx[d] = sel[d].first; if (d == 0) return; }
      if (idx.empty()) return;
    }
  }
  Val vars_by_elem(const std::string& elem) {
    // elem = base[k]; find by scanning is too slow: rebuild from the canonical map lazily
    auto it = elem_cache.find(elem);
    if (it != elem_cache.end()) return it->second;
    for (auto& kv : vars) elem_cache[kv.second.var] = kv.second;
    it = elem_cache.find(elem);
    if (it == elem_cache.end()) fail("internal: unknown element " + elem);
    return it->second;
  }
  std::map<std::string, Val> elem_cache;

  std::vector<Val> val_list(const std::string& t) {
    std::vector<Val> out;
    for (const std::string& w : words(t)) expand_ref(w, &out);
    return out;
  }

  // ---- functional expressions: eq(add(x,3),y)
  struct Tok { const std::string& s; size_t p = 0; };
  Val expr(Tok& t, bool want_bool = false) {
    while (t.p < t.s.size() && std::isspace((unsigned char)t.s[t.p])) ++t.p;
    std::string id;
    while (t.p < t.s.size() && t.s[t.p] != '(' && t.s[t.p] != ')' && t.s[t.p] != ',' && !std::isspace((unsigned char)t.s[t.p])) id += t.s[t.p++];
    while (t.p < t.s.size() && std::isspace((unsigned char)t.s[t.p])) ++t.p;
    if (t.p >= t.s.size() || t.s[t.p] != '(') {
      std::vector<Val> v;
      expand_ref(id, &v);
      if (v.size() != 1) fail("`" + id + "` is not a single value");
      return v[0];
    }
    ++t.p;
    std::vector<Val> a;
    for (;;) {
      while (t.p < t.s.size() && std::isspace((unsigned char)t.s[t.p])) ++t.p;
      if (t.p < t.s.size() && t.s[t.p] == ')') { ++t.p; break; }
      a.push_back(expr(t));
      while (t.p < t.s.size() && std::isspace((unsigned char)t.s[t.p])) ++t.p;
      if (t.p < t.s.size() && t.s[t.p] == ',') ++t.p;
      else if (t.p < t.s.size() && t.s[t.p] == ')') { ++t.p; break; }
      else fail("bad expression near `" + t.s.substr(t.p, 20) + "`");
    }
    return apply(id, a);
  }

  Val as_bool(const Val& v) {  // integer 0/1 seen as a Boolean literal
    if (v.is_const || v.is_bool) return v;
    Val b = fresh_bool();
    cons << "constraint int_eq_reif(" << v.str() << ", 1, " << b.var << ");\n";  // v in 0..1 by construction of callers
    cons << "constraint int_le(0, " << v.str() << ");\nconstraint int_le(" << v.str() << ", 1);\n";
    return b;
  }
  Val as_int(const Val& v) {
    if (!v.is_bool || v.is_const) return v;
    Val i = fresh_int(0, 1);
    cons << "constraint int_eq_reif(" << i.var << ", 1, " << v.var << ");\n";
    return i;
  }
  std::string lit(const Val& v) { return v.is_const ? (v.c != 0 ? "true" : "false") : v.var; }

  Val binary_arith(const std::string& fz, Val a, Val b, int64_t lo, int64_t hi) {
    a = as_int(a); b = as_int(b);
    Val r = fresh_int(lo, hi);
    cons << "constraint " << fz << "(" << a.str() << ", " << b.str() << ", " << r.var << ");\n";
    return r;
  }
  Val cmp(const std::string& fz, Val a, Val b) {  // reified comparison
    a = as_int(a); b = as_int(b);
    Val r = fresh_bool();
    cons << "constraint " << fz << "_reif(" << a.str() << ", " << b.str() << ", " << r.var << ");\n";
    return r;
  }
  static void mul_bounds(const Val& a, const Val& b, int64_t* lo, int64_t* hi) {
    const __int128 c[4] = {(__int128)a.lo * b.lo, (__int128)a.lo * b.hi, (__int128)a.hi * b.lo, (__int128)a.hi * b.hi};
    *lo = sat(*std::min_element(c, c + 4)); *hi = sat(*std::max_element(c, c + 4));
  }

  Val apply(const std::string& f, std::vector<Val> a) {
    auto need = [&](size_t n) { if (a.size() != n) fail("operator `" + f + "` expects " + std::to_string(n) + " arguments"); };
    auto fold = [&](auto&& op) { if (a.empty()) fail("operator `" + f + "` without arguments"); Val r = a[0]; for (size_t i = 1; i < a.size(); ++i) r = op(r, a[i]); return r; };
    if (f == "neg") { need(1); Val x = as_int(a[0]); if (x.is_const) return konst(-x.c); return binary_arith("int_minus", konst(0), x, -x.hi, -x.lo); }
    if (f == "abs") { need(1); Val x = as_int(a[0]); if (x.is_const) return konst(std::llabs(x.c)); Val r = fresh_int(0, std::max(std::llabs(x.lo), std::llabs(x.hi))); cons << "constraint int_abs(" << x.str() << ", " << r.var << ");\n"; return r; }
    if (f == "add") return fold([&](Val x, Val y) { if (x.is_const && y.is_const) return konst(x.c + y.c); return binary_arith("int_plus", x, y, x.lo + y.lo, x.hi + y.hi); });
    if (f == "sub") { need(2); if (a[0].is_const && a[1].is_const) return konst(a[0].c - a[1].c); return binary_arith("int_minus", a[0], a[1], a[0].lo - a[1].hi, a[0].hi - a[1].lo); }
    if (f == "mul") return fold([&](Val x, Val y) { if (x.is_const && y.is_const) return konst(x.c * y.c); int64_t lo, hi; mul_bounds(x, y, &lo, &hi); return binary_arith("int_times", x, y, lo, hi); });
    if (f == "sqr") { need(1); int64_t lo, hi; mul_bounds(a[0], a[0], &lo, &hi); return binary_arith("int_times", a[0], a[0], std::max<int64_t>(0, lo), hi); }
    if (f == "div") { need(2); const int64_t m = std::max(std::llabs(a[0].lo), std::llabs(a[0].hi)); return binary_arith("int_div", a[0], a[1], -m, m); }
    if (f == "mod") { need(2); const int64_t m = std::max(std::llabs(a[1].lo), std::llabs(a[1].hi)); return binary_arith("int_mod", a[0], a[1], -m, m); }
    if (f == "min") return fold([&](Val x, Val y) { return binary_arith("int_min", x, y, std::min(x.lo, y.lo), std::min(x.hi, y.hi)); });
    if (f == "max") return fold([&](Val x, Val y) { return binary_arith("int_max", x, y, std::max(x.lo, y.lo), std::max(x.hi, y.hi)); });
    if (f == "dist") { need(2); Val d = apply("sub", {a[0], a[1]}); return apply("abs", {d}); }
    if (f == "lt") { need(2); return cmp("int_lt", a[0], a[1]); }
    if (f == "le") { need(2); return cmp("int_le", a[0], a[1]); }
    if (f == "gt") { need(2); return cmp("int_lt", a[1], a[0]); }
    if (f == "ge") { need(2); return cmp("int_le", a[1], a[0]); }
    if (f == "ne") { need(2); return cmp("int_ne", a[0], a[1]); }
    if (f == "eq") {
      if (a.size() < 2) fail("eq needs two arguments");
      std::vector<Val> parts;
      for (size_t i = 1; i < a.size(); ++i) parts.push_back(cmp("int_eq", a[0], a[i]));
      return parts.size() == 1 ? parts[0] : apply("and", parts);
    }
    if (f == "not") { need(1); Val x = as_bool(a[0]); if (x.is_const) return konst(x.c ? 0 : 1); Val r = fresh_bool(); cons << "constraint bool_not(" << x.var << ", " << r.var << ");\n"; Val rr = r; rr.is_bool = true; return rr; }
    if (f == "and" || f == "or") {
      if (a.empty()) fail(f + " without arguments");
      std::vector<std::string> lits;  // (as_bool may post constraints of its own: before the line is started)
      for (auto& x : a) lits.push_back(lit(as_bool(x)));
      Val r = fresh_bool();
      cons << "constraint array_bool_" << f << "([";
      for (size_t i = 0; i < lits.size(); ++i) cons << (i ? ", " : "") << lits[i];
      cons << "], " << r.var << ");\n";
      return r;
    }
    if (f == "xor") { return fold([&](Val x, Val y) { const std::string lx = lit(as_bool(x)), ly = lit(as_bool(y)); Val r = fresh_bool(); cons << "constraint bool_xor(" << lx << ", " << ly << ", " << r.var << ");\n"; return r; }); }
    if (f == "iff") { need(2); const std::string lx = lit(as_bool(a[0])), ly = lit(as_bool(a[1])); Val r = fresh_bool(); cons << "constraint bool_eq_reif(" << lx << ", " << ly << ", " << r.var << ");\n"; return r; }
    if (f == "imp") { need(2); const std::string lx = lit(as_bool(a[0])), ly = lit(as_bool(a[1])); Val r = fresh_bool(); cons << "constraint bool_le_reif(" << lx << ", " << ly << ", " << r.var << ");\n"; return r; }
    if (f == "if") {  // if(c, t, e) = r:  c -> r = t, not c -> r = e
      need(3);
      Val c = as_bool(a[0]), t = as_int(a[1]), e = as_int(a[2]);
      Val r = fresh_int(std::min(t.lo, e.lo), std::max(t.hi, e.hi));
      Val et = cmp("int_eq", r, t), ee = cmp("int_eq", r, e);
      cons << "constraint bool_clause([" << et.var << "], [" << lit(c) << "]);\n";
      cons << "constraint bool_clause([" << ee.var << ", " << lit(c) << "], []);\n";
      return r;
    }
    fail("unsupported operator `" + f + "`");
  }
  void post_true(const Val& b) {
    if (b.is_const) { if (!b.c) cons << "constraint bool_eq(true, false);\n"; return; }
    cons << "constraint bool_eq(" << b.var << ", true);\n";
  }

  // (op, operand) conditions of sum / minimum / maximum / cumulative
  struct Cond { std::string op; Val rhs; std::vector<int64_t> set; };  // op in / notin: `set` holds the values
  Cond condition(const std::string& t) {
    std::string s = t;
    for (char& c : s) if (c == '(' || c == ')' || c == ',' || c == '{' || c == '}') c = ' ';
    const auto ws = words(s);
    if (ws.size() < 2) fail("unsupported condition `" + t + "`");
    Cond c; c.op = ws[0];
    if (c.op == "in" || c.op == "notin") {
      for (size_t i = 1; i < ws.size(); ++i) for (int64_t v : int_list(ws[i])) c.set.push_back(v);
      if (c.set.size() > 100000) fail("condition set too large");
      return c;
    }
    if (ws.size() != 2) fail("unsupported condition `" + t + "`");
    std::vector<Val> v; expand_ref(ws[1], &v);
    if (v.size() != 1) fail("condition operand must be a single value");
    c.rhs = v[0];
    return c;
  }
  // lhs (in | notin) set, as a Boolean
  void post_membership(const Val& lhs, const Cond& c) {
    std::vector<Val> eqs;
    for (int64_t v : c.set) eqs.push_back(cmp("int_eq", lhs, konst(v)));
    if (c.op == "in") post_true(eqs.empty() ? konst(0) : (eqs.size() == 1 ? eqs[0] : apply("or", eqs)));
    else for (auto& e : eqs) post_true(apply("not", {e}));
  }
  // a linear expression as one value (fresh variable when it has more than one term)
  Val linear_value(const std::vector<int64_t>& coef, const std::vector<Val>& xs) {
    std::vector<Val> terms;
    for (size_t i = 0; i < xs.size(); ++i) terms.push_back(coef[i] == 1 ? as_int(xs[i]) : apply("mul", {konst(coef[i]), xs[i]}));
    if (terms.empty()) return konst(0);
    return terms.size() == 1 ? terms[0] : apply("add", terms);
  }
  void post_linear(std::vector<int64_t> coef, std::vector<Val> xs, const Cond& c) {
    if (c.op == "in" || c.op == "notin") { post_membership(linear_value(coef, xs), c); return; }
    int64_t k = 0;
    if (c.rhs.is_const) k = c.rhs.c; else { coef.push_back(-1); xs.push_back(c.rhs); }
    // fold constants of the list into the right-hand side
    std::vector<int64_t> cc; std::vector<std::string> vv;
    for (size_t i = 0; i < xs.size(); ++i) { Val x = as_int(xs[i]); if (x.is_const) k -= coef[i] * x.c; else { cc.push_back(coef[i]); vv.push_back(x.var); } }
    static const std::map<std::string, std::string> fz = {{"le", "int_lin_le"}, {"lt", "int_lin_lt"}, {"ge", "int_lin_ge"}, {"gt", "int_lin_gt"}, {"eq", "int_lin_eq"}, {"ne", "int_lin_ne"}};
    auto it = fz.find(c.op);
    if (it == fz.end()) fail("unsupported condition operator `" + c.op + "`");
    if (cc.empty()) {  // 0 (op) k
      const bool ok = c.op == "le" ? 0 <= k : c.op == "lt" ? 0 < k : c.op == "ge" ? 0 >= k : c.op == "gt" ? 0 > k : c.op == "eq" ? 0 == k : 0 != k;
      if (!ok) post_true(konst(0));
      return;
    }
    cons << "constraint " << it->second << "([";
    for (size_t i = 0; i < cc.size(); ++i) cons << (i ? ", " : "") << cc[i];
    cons << "], [";
    for (size_t i = 0; i < vv.size(); ++i) cons << (i ? ", " : "") << vv[i];
    cons << "], " << k << ");\n";
  }
  void post_cond(const Val& lhs, const Cond& c) { post_linear({1}, {lhs}, c); }

  // `%i`: the i-th argument of the <args> element; `%...`: the arguments behind the highest numbered %i of the whole template (`rest`: its index; all of them
  // when the template names none).
  static std::string subst(std::string t, const std::vector<std::string>& args, size_t rest) {
    if (t.find("%...") != std::string::npos) {
      std::string tail;
      for (size_t i = rest; i < args.size(); ++i) tail += args[i] + " ";
      for (size_t q; (q = t.find("%...")) != std::string::npos;) t.replace(q, 4, tail);
    }
    for (size_t i = args.size(); i-- > 0;) {
      const std::string key = "%" + std::to_string(i);
      for (size_t q; (q = t.find(key)) != std::string::npos;) t.replace(q, key.size(), args[i]);
    }
    return t;
  }
  static size_t numbered_params(const Xml& n) {  // 1 + the highest i of a %i in the template, 0 when there is none
    size_t k = 0;
    for (size_t p = 0; (p = n.text.find('%', p)) != std::string::npos; ++p)
      if (p + 1 < n.text.size() && std::isdigit((unsigned char)n.text[p + 1])) k = std::max(k, (size_t)std::stoul(n.text.substr(p + 1)) + 1);
    for (auto& c : n.kids) k = std::max(k, numbered_params(*c));
    return k;
  }
  static std::unique_ptr<Xml> clone_subst(const Xml& n, const std::vector<std::string>& args, size_t rest) {
    auto c = std::make_unique<Xml>();
    c->name = n.name; c->attr = n.attr; c->text = subst(n.text, args, rest);
    for (auto& k : n.kids) c->kids.push_back(clone_subst(*k, args, rest));
    return c;
  }
  static std::string text_of(const Xml& n, const std::string& child) {
    const Xml* c = n.child(child);
    return c ? c->text : std::string();
  }

  // ---- helpers shared by several constraints
  std::map<std::string, std::string> id_of;  // FlatZinc name -> XCSP3 id of a declared variable (slide / cardinality re-parse ids)
  std::string xcsp_name(const Val& v) {
    if (v.is_const) return std::to_string(v.c);
    auto it = id_of.find(v.var);
    if (it == id_of.end()) fail("internal: `" + v.var + "` has no XCSP3 id");
    return it->second;
  }
  // table constraint over `xs`: tuples of integers or `*`
  void post_table(const std::vector<Val>& xs, const std::vector<std::vector<std::string>>& tuples, bool support) {
    std::vector<Val> rows;
    for (auto& tp : tuples) {
      std::vector<Val> lits;
      bool impossible = false;
      for (size_t j = 0; j < xs.size(); ++j) {
        if (tp[j] == "*") continue;
        if (!is_int(tp[j])) fail("table values must be integers or *");
        const int64_t v = std::stoll(tp[j]);
        if (support && (v < xs[j].lo || v > xs[j].hi)) { impossible = true; break; }
        lits.push_back(cmp(support ? "int_eq" : "int_ne", xs[j], konst(v)));
      }
      if (impossible) continue;
      if (support) rows.push_back(lits.empty() ? konst(1) : (lits.size() == 1 ? lits[0] : apply("and", lits)));
      else if (lits.empty()) post_true(konst(0));
      else post_true(lits.size() == 1 ? lits[0] : apply("or", lits));
    }
    if (support) post_true(rows.empty() ? konst(0) : (rows.size() == 1 ? rows[0] : apply("or", rows)));
  }
  // "(a,b,c)(d,e,f)" -> rows of values
  std::vector<std::vector<Val>> matrix_rows(const std::string& text) {
    std::vector<std::vector<Val>> rows;
    std::string t = text, cur;
    if (t.find('(') == std::string::npos) {  // a 2-dimensional array reference: x[][]
      for (const std::string& w : words(t)) {
        const size_t br = w.find('[');
        auto at = arrays.find(br == std::string::npos ? w : w.substr(0, br));
        if (at == arrays.end() || at->second.dims.size() != 2) fail("matrix must be tuples or a 2-dimensional array");
        std::vector<Val> all; expand_ref(w, &all);
        const size_t cols = (size_t)at->second.dims[1];
        if (all.size() % cols != 0) fail("matrix: partial rows are not supported");
        for (size_t i = 0; i < all.size(); i += cols) rows.emplace_back(all.begin() + (long)i, all.begin() + (long)(i + cols));
      }
      return rows;
    }
    for (char& c : t) if (c == ',') c = ' ';
    for (size_t i = 0; i < t.size(); ++i) {
      if (t[i] == '(') cur.clear();
      else if (t[i] == ')') rows.push_back(val_list(cur));
      else cur += t[i];
    }
    return rows;
  }
  // x (op) y lexicographically, op in lt le gt ge
  void lex_pair(std::vector<Val> x, std::vector<Val> y, const std::string& op) {
    if (x.size() != y.size()) fail("lex: lists differ in length");
    if (op == "gt" || op == "ge") std::swap(x, y);
    const bool strict = op == "lt" || op == "gt";
    // for every i: (x_0 = y_0 and ... and x_{i-1} = y_{i-1}) -> x_i <= y_i ; strict: additionally some x_i != y_i
    std::vector<Val> eq_prefix;
    for (size_t i = 0; i < x.size(); ++i) {
      Val le = cmp("int_le", x[i], y[i]);
      if (eq_prefix.empty()) post_true(le);
      else {
        cons << "constraint bool_clause([" << lit(le) << "], [";
        for (size_t j = 0; j < eq_prefix.size(); ++j) cons << (j ? ", " : "") << lit(eq_prefix[j]);
        cons << "]);\n";
      }
      eq_prefix.push_back(cmp("int_eq", x[i], y[i]));
    }
    if (strict) {
      std::vector<Val> ne;
      for (auto& e : eq_prefix) ne.push_back(apply("not", {e}));
      post_true(ne.empty() ? konst(0) : (ne.size() == 1 ? ne[0] : apply("or", ne)));
    }
  }
  void lex_chain(const std::vector<std::vector<Val>>& rows, const std::string& op) {
    for (size_t i = 0; i + 1 < rows.size(); ++i) lex_pair(rows[i], rows[i + 1], op);
  }
  // number of cells equal to v, as a value
  Val count_eq(const std::vector<Val>& xs, const Val& v) {
    std::vector<Val> hits;
    for (auto& x : xs) hits.push_back(as_int(cmp("int_eq", x, v)));
    if (hits.empty()) return konst(0);
    return hits.size() == 1 ? hits[0] : apply("add", hits);
  }
  // number of distinct values taken by xs (not counting those of `except`), as a value
  Val n_values(const std::vector<Val>& xs, const std::vector<int64_t>& except) {
    int64_t lo = INT64_MAX, hi = INT64_MIN;
    for (auto& x : xs) { lo = std::min(lo, x.lo); hi = std::max(hi, x.hi); }
    if (xs.empty()) return konst(0);
    if (hi - lo > 100000) fail("nValues: value range too large");
    std::vector<Val> used;
    for (int64_t v = lo; v <= hi; ++v) {
      if (std::find(except.begin(), except.end(), v) != except.end()) continue;
      std::vector<Val> any;
      for (auto& x : xs) if (x.lo <= v && v <= x.hi) any.push_back(cmp("int_eq", x, konst(v)));
      if (any.empty()) continue;
      used.push_back(as_int(any.size() == 1 ? any[0] : apply("or", any)));
    }
    if (used.empty()) return konst(0);
    return used.size() == 1 ? used[0] : apply("add", used);
  }

  void constraint(const Xml& n) {
    const std::string& k = n.name;
    if (k == "block") { for (auto& c : n.kids) constraint(*c); return; }
    if (k == "group") {
      const Xml* tmpl = nullptr;
      for (auto& c : n.kids) if (c->name != "args") { tmpl = c.get(); break; }
      if (!tmpl) fail("group without a constraint template");
      const size_t rest = numbered_params(*tmpl);
      for (auto& c : n.kids) if (c->name == "args") constraint(*clone_subst(*tmpl, words(c->text), rest));
      return;
    }
    if (k == "intension") {
      const std::string src = n.child("function") ? n.child("function")->text : n.text;
      Tok t{src};
      post_true(as_bool(expr(t)));
      return;
    }
    if (k == "allDifferent") {
      auto pairwise = [&](const std::vector<Val>& xs, const std::vector<int64_t>& except) {
        for (size_t i = 0; i < xs.size(); ++i)
          for (size_t j = i + 1; j < xs.size(); ++j) {
            if (except.empty()) { cons << "constraint int_ne(" << xs[i].str() << ", " << xs[j].str() << ");\n"; continue; }
            // different, or both equal to an excepted value
            std::vector<Val> ok{cmp("int_ne", xs[i], xs[j])};
            for (int64_t e : except) if (xs[i].lo <= e && e <= xs[i].hi) ok.push_back(cmp("int_eq", xs[i], konst(e)));
            post_true(ok.size() == 1 ? ok[0] : apply("or", ok));
          }
      };
      std::vector<int64_t> except = n.child("except") ? int_list(n.child("except")->text) : std::vector<int64_t>();
      if (const Xml* m = n.child("matrix")) {  // every row and every column
        std::vector<std::vector<Val>> rows = matrix_rows(m->text);
        for (auto& r : rows) pairwise(r, except);
        for (size_t j = 0; !rows.empty() && j < rows[0].size(); ++j) { std::vector<Val> col; for (auto& r : rows) { if (r.size() != rows[0].size()) fail("allDifferent: ragged matrix"); col.push_back(r[j]); } pairwise(col, except); }
        return;
      }
      std::vector<const Xml*> lists;
      for (auto& c : n.kids) if (c->name == "list") lists.push_back(c.get());
      if (lists.size() >= 2) {  // the lists, seen as tuples, differ pairwise
        std::vector<std::vector<Val>> ts;
        for (auto* l : lists) ts.push_back(val_list(l->text));
        for (size_t a = 0; a < ts.size(); ++a)
          for (size_t b = a + 1; b < ts.size(); ++b) {
            if (ts[a].size() != ts[b].size()) fail("allDifferent: lists differ in length");
            std::vector<Val> ne;
            for (size_t q = 0; q < ts[a].size(); ++q) ne.push_back(cmp("int_ne", ts[a][q], ts[b][q]));
            post_true(ne.empty() ? konst(0) : (ne.size() == 1 ? ne[0] : apply("or", ne)));
          }
        return;
      }
      pairwise(val_list(lists.size() == 1 ? lists[0]->text : n.text), except);
      return;
    }
    if (k == "ordered") {
      std::vector<Val> xs = val_list(text_of(n, "list"));
      std::string op = words(text_of(n, "operator")).empty() ? "le" : words(text_of(n, "operator"))[0];
      if (n.child("lengths")) {  // x_i + l_i (op) x_{i+1}
        std::vector<Val> len = val_list(text_of(n, "lengths"));
        if (len.size() + 1 != xs.size()) fail("ordered: lengths must have one entry less than the list");
        for (size_t i = 0; i + 1 < xs.size(); ++i) { Cond c; c.op = op; c.rhs = xs[i + 1]; post_cond(apply("add", {xs[i], len[i]}), c); }
        return;
      }
      for (size_t i = 0; i + 1 < xs.size(); ++i) {
        if (op == "le") cons << "constraint int_le(" << xs[i].str() << ", " << xs[i + 1].str() << ");\n";
        else if (op == "lt") cons << "constraint int_lt(" << xs[i].str() << ", " << xs[i + 1].str() << ");\n";
        else if (op == "ge") cons << "constraint int_le(" << xs[i + 1].str() << ", " << xs[i].str() << ");\n";
        else if (op == "gt") cons << "constraint int_lt(" << xs[i + 1].str() << ", " << xs[i].str() << ");\n";
        else fail("unsupported order `" + op + "`");
      }
      return;
    }
    if (k == "sum") {
      std::vector<Val> xs = val_list(text_of(n, "list"));
      std::vector<int64_t> coef(xs.size(), 1);
      if (n.child("coeffs")) {
        std::vector<Val> cv = val_list(n.child("coeffs")->text);
        if (cv.size() != xs.size()) fail("sum: coeffs and list differ in length");
        for (size_t i = 0; i < xs.size(); ++i) {
          if (cv[i].is_const) coef[i] = cv[i].c;
          else xs[i] = apply("mul", {cv[i], xs[i]});  // a variable coefficient: the term is a product
        }
      }
      post_linear(coef, xs, condition(text_of(n, "condition")));
      return;
    }
    if (k == "minimum" || k == "maximum") {
      std::vector<Val> xs = val_list(text_of(n, "list"));
      Val m = apply(k == "minimum" ? "min" : "max", xs);
      post_cond(m, condition(text_of(n, "condition")));
      return;
    }
    if (k == "instantiation") {
      std::vector<Val> xs = val_list(text_of(n, "list"));
      std::vector<int64_t> vs = int_list(text_of(n, "values"));
      if (xs.size() != vs.size()) fail("instantiation: list and values differ in length");
      for (size_t i = 0; i < xs.size(); ++i) cons << "constraint int_eq(" << xs[i].str() << ", " << vs[i] << ");\n";
      return;
    }
    if (k == "element" && n.child("matrix")) {  // m[i][j] = v
      const Xml* m = n.child("matrix");
      std::vector<std::vector<Val>> rows = matrix_rows(m->text);
      if (rows.empty()) fail("element: empty matrix");
      std::vector<Val> iv = val_list(text_of(n, "index")), vv = val_list(text_of(n, "value"));
      if (iv.size() != 2 || vv.size() != 1) fail("element on a matrix needs two indices and one value");
      const int64_t sr = m->get("startRowIndex").empty() ? 0 : std::stoll(m->get("startRowIndex")), sc = m->get("startColIndex").empty() ? 0 : std::stoll(m->get("startColIndex"));
      const int64_t cols = (int64_t)rows[0].size();
      Cond c0; c0.op = "ge"; c0.rhs = konst(sc); Cond c1; c1.op = "le"; c1.rhs = konst(sc + cols - 1);
      post_cond(iv[1], c0); post_cond(iv[1], c1);
      Val flat = apply("add", {apply("mul", {konst(cols), apply("sub", {iv[0], konst(sr)})}), apply("sub", {iv[1], konst(sc)}), konst(1)});
      bool all_const = true;
      std::vector<Val> cells;
      for (auto& r : rows) { if ((int64_t)r.size() != cols) fail("element: ragged matrix"); for (auto& x : r) { cells.push_back(x); all_const &= x.is_const; } }
      cons << "constraint " << (all_const ? "array_int_element(" : "array_var_int_element(") << flat.str() << ", [";
      for (size_t i = 0; i < cells.size(); ++i) cons << (i ? ", " : "") << cells[i].str();
      cons << "], " << vv[0].str() << ");\n";
      return;
    }
    if (k == "element") {
      const Xml* l = n.child("list");
      if (!l) fail("element without list");
      std::vector<Val> xs = val_list(l->text);
      const int64_t start = l->get("startIndex").empty() ? 0 : std::stoll(l->get("startIndex"));
      std::vector<Val> iv = val_list(text_of(n, "index")), vv = val_list(text_of(n, "value"));
      if (vv.size() != 1) fail("element needs one value");
      if (iv.empty()) {  // membership: value occurs in the list
        std::vector<Val> eqs;
        for (auto& x : xs) eqs.push_back(cmp("int_eq", x, vv[0]));
        post_true(apply("or", eqs));
        return;
      }
      Val idx1 = apply("add", {iv[0], konst(1 - start)});  // FlatZinc arrays start at 1
      bool all_const = true;
      for (auto& x : xs) all_const &= x.is_const;
      cons << "constraint " << (all_const ? "array_int_element(" : "array_var_int_element(") << idx1.str() << ", [";
      for (size_t i = 0; i < xs.size(); ++i) cons << (i ? ", " : "") << xs[i].str();
      cons << "], " << vv[0].str() << ");\n";
      return;
    }
    if (k == "extension") {
      std::vector<Val> xs = val_list(text_of(n, "list"));
      const bool support = n.child("supports") != nullptr;
      const Xml* tb = support ? n.child("supports") : n.child("conflicts");
      if (!tb) fail("extension without supports/conflicts");
      if (xs.size() == 1) {  // unary table: a set of values
        std::vector<int64_t> vals = int_list(tb->text);
        if (support) {
          cons << "constraint set_in(" << xs[0].str() << ", {";
          for (size_t i = 0; i < vals.size(); ++i) cons << (i ? "," : "") << vals[i];
          cons << "});\n";
        } else for (int64_t v : vals) cons << "constraint int_ne(" << xs[0].str() << ", " << v << ");\n";
        return;
      }
      std::vector<std::vector<std::string>> tuples;
      {
        std::string t = tb->text, cur;
        for (char& c : t) if (c == ',') c = ' ';
        for (size_t i = 0; i < t.size(); ++i) {
          if (t[i] == '(') cur.clear();
          else if (t[i] == ')') { tuples.push_back(words(cur)); if (tuples.back().size() != xs.size()) fail("tuple arity differs from the scope"); }
          else cur += t[i];
        }
      }
      post_table(xs, tuples, support);
      return;
    }
    if (k == "cumulative") {
      std::vector<Val> s = val_list(text_of(n, "origins")), len = val_list(text_of(n, "lengths")), h = val_list(text_of(n, "heights"));
      if (n.child("machines")) fail("cumulative with <machines> is not supported");
      if (s.size() != len.size() || s.size() != h.size()) fail("cumulative: origins, lengths and heights differ in length");
      if (n.child("ends")) {  // e_i = s_i + l_i
        std::vector<Val> e = val_list(text_of(n, "ends"));
        if (e.size() != s.size()) fail("cumulative: ends differ in length");
        for (size_t i = 0; i < s.size(); ++i) { Cond c; c.op = "eq"; c.rhs = e[i]; post_cond(apply("add", {s[i], len[i]}), c); }
      }
      const Cond c = condition(text_of(n, "condition"));
      int64_t t0 = INT64_MAX, t1 = INT64_MIN;
      for (size_t i = 0; i < s.size(); ++i) { if (len[i].hi <= 0 || (h[i].is_const && h[i].c == 0)) continue; t0 = std::min(t0, s[i].lo); t1 = std::max(t1, s[i].hi + len[i].hi - 1); }
      if (t0 > t1) return;
      if (t1 - t0 > 100000) fail("cumulative: horizon too long for the time-indexed decomposition");
      for (int64_t t = t0; t <= t1; ++t) {  // sum_i h_i * [s_i <= t < s_i + l_i]  (cond)  limit
        std::vector<int64_t> coef; std::vector<Val> terms;
        for (size_t i = 0; i < s.size(); ++i) {
          if (len[i].hi <= 0 || (h[i].is_const && h[i].c == 0) || s[i].lo > t || s[i].hi + len[i].hi - 1 < t) continue;
          Val a = cmp("int_le", s[i], konst(t));
          Val b = len[i].is_const ? cmp("int_le", konst(t - len[i].c + 1), s[i]) : cmp("int_lt", konst(t), apply("add", {s[i], len[i]}));
          Val run = as_int(apply("and", {a, b}));
          if (h[i].is_const) { terms.push_back(run); coef.push_back(h[i].c); }
          else { terms.push_back(apply("mul", {run, h[i]})); coef.push_back(1); }
        }
        if (!terms.empty()) post_linear(coef, terms, c);
      }
      return;
    }
    if (k == "allEqual") {
      std::vector<Val> xs = val_list(n.child("list") ? n.child("list")->text : n.text);
      for (size_t i = 0; i + 1 < xs.size(); ++i) cons << "constraint int_eq(" << xs[i].str() << ", " << xs[i + 1].str() << ");\n";
      return;
    }
    if (k == "count") {  // number of list cells whose value is one of <values>, compared by the condition
      std::vector<Val> xs = val_list(text_of(n, "list")), vals = val_list(text_of(n, "values"));
      if (vals.empty()) fail("count without values");
      std::vector<Val> hits;
      for (auto& x : xs) {
        std::vector<Val> any;
        for (auto& v : vals) any.push_back(cmp("int_eq", x, v));
        hits.push_back(as_int(any.size() == 1 ? any[0] : apply("or", any)));
      }
      post_linear(std::vector<int64_t>(hits.size(), 1), hits, condition(text_of(n, "condition")));
      return;
    }
    if (k == "noOverlap") {  // constant or variable lengths: in some dimension, s_i + l_i <= s_j  or  s_j + l_j <= s_i
      if (text_of(n, "origins").find('(') != std::string::npos) {
        std::vector<std::vector<Val>> o = matrix_rows(text_of(n, "origins")), len = matrix_rows(text_of(n, "lengths"));
        if (o.size() != len.size()) fail("noOverlap: origins and lengths differ in length");
        for (size_t i = 0; i < o.size(); ++i)
          for (size_t j = i + 1; j < o.size(); ++j) {
            if (o[i].size() != o[j].size() || len[i].size() != o[i].size() || len[j].size() != o[j].size()) fail("noOverlap: dimensions differ");
            std::vector<Val> apart;
            for (size_t d = 0; d < o[i].size(); ++d) {
              apart.push_back(cmp("int_le", apply("add", {o[i][d], len[i][d]}), o[j][d]));
              apart.push_back(cmp("int_le", apply("add", {o[j][d], len[j][d]}), o[i][d]));
            }
            post_true(apply("or", apart));
          }
        return;
      }
      std::vector<Val> o = val_list(text_of(n, "origins")), len = val_list(text_of(n, "lengths"));
      if (o.size() != len.size()) fail("noOverlap: origins and lengths differ in length");
      for (size_t i = 0; i < o.size(); ++i)
        for (size_t j = i + 1; j < o.size(); ++j) {
          Val a = cmp("int_le", apply("add", {o[i], len[i]}), o[j]), b = cmp("int_le", apply("add", {o[j], len[j]}), o[i]);
          cons << "constraint bool_clause([" << lit(a) << ", " << lit(b) << "], []);\n";
        }
      return;
    }
    if (k == "channel") {  // one list: x[i] = j  <=>  x[j] = i ; two lists: x[i] = j  <=>  y[j] = i
      std::vector<const Xml*> lists;
      for (auto& c : n.kids) if (c->name == "list") lists.push_back(c.get());
      if (lists.empty() || lists.size() > 2 || n.child("value")) fail("this form of channel is not supported");
      std::vector<Val> x = val_list(lists[0]->text), y = lists.size() == 2 ? val_list(lists[1]->text) : x;
      const int64_t sx = lists[0]->get("startIndex").empty() ? 0 : std::stoll(lists[0]->get("startIndex"));
      const int64_t sy = lists.size() == 2 && !lists[1]->get("startIndex").empty() ? std::stoll(lists[1]->get("startIndex")) : (lists.size() == 2 ? 0 : sx);
      if (x.size() != y.size()) fail("channel: lists differ in length");
      for (size_t i = 0; i < x.size(); ++i)
        for (size_t j = 0; j < y.size(); ++j) {
          Val a = cmp("int_eq", x[i], konst((int64_t)j + sy)), b = cmp("int_eq", y[j], konst((int64_t)i + sx));
          cons << "constraint bool_eq(" << lit(a) << ", " << lit(b) << ");\n";
        }
      return;
    }
    if (k == "regular" || k == "mdd") {
      // a layered automaton: one state variable between two consecutive positions, (q_{t-1}, x_t, q_t) in the transition table
      std::vector<Val> xs = val_list(text_of(n, "list"));
      std::vector<std::vector<std::string>> trans;
      std::map<std::string, int64_t> id;
      auto state = [&](const std::string& nm) { auto it = id.find(nm); if (it != id.end()) return it->second; const int64_t v = (int64_t)id.size(); id[nm] = v; return v; };
      {
        std::string t = text_of(n, "transitions"), cur;
        for (char& c : t) if (c == ',') c = ' ';
        for (size_t i = 0; i < t.size(); ++i) {
          if (t[i] == '(') cur.clear();
          else if (t[i] == ')') { auto w = words(cur); if (w.size() != 3) fail(k + ": a transition is (state, value, state)"); trans.push_back(w); }
          else cur += t[i];
        }
      }
      if (trans.empty()) fail(k + " without transitions");
      std::vector<std::vector<std::string>> tuples;
      std::map<int64_t, bool> is_src, is_dst;
      for (auto& w : trans) { const int64_t a = state(w[0]), b = state(w[2]); is_src[a] = true; is_dst[b] = true; tuples.push_back({std::to_string(a), w[1], std::to_string(b)}); }
      std::vector<int64_t> start, fin;
      if (k == "regular") {
        for (auto& w : words(text_of(n, "start"))) start.push_back(state(w));
        for (auto& w : words(text_of(n, "final"))) fin.push_back(state(w));
        if (start.size() != 1) fail("regular needs one start state");
      } else {  // mdd: the root is the node no transition leads to, the terminal the one none leaves
        for (auto& kv : id) { if (!is_dst.count(kv.second)) start.push_back(kv.second); if (!is_src.count(kv.second)) fin.push_back(kv.second); }
        if (start.size() != 1) fail("mdd needs exactly one root");
      }
      const int64_t nq = (int64_t)id.size();
      std::vector<Val> q;
      for (size_t t = 0; t <= xs.size(); ++t) q.push_back(fresh_int(0, nq - 1));
      cons << "constraint int_eq(" << q[0].var << ", " << start[0] << ");\n";
      for (size_t t = 0; t < xs.size(); ++t) post_table({q[t], xs[t], q[t + 1]}, tuples, true);
      Cond c; c.op = "in"; c.set = fin;
      post_membership(q[xs.size()], c);
      return;
    }
    if (k == "lex") {
      std::vector<std::vector<Val>> rows;
      std::string op = words(text_of(n, "operator")).empty() ? "le" : words(text_of(n, "operator"))[0];
      if (const Xml* m = n.child("matrix")) {
        std::vector<std::vector<Val>> mat = matrix_rows(m->text);
        lex_chain(mat, op);
        std::vector<std::vector<Val>> cols(mat.empty() ? 0 : mat[0].size());
        for (auto& r : mat) { if (r.size() != cols.size()) fail("lex: ragged matrix"); for (size_t j = 0; j < r.size(); ++j) cols[j].push_back(r[j]); }
        lex_chain(cols, op);
        return;
      }
      for (auto& c : n.kids) if (c->name == "list") rows.push_back(val_list(c->text));
      lex_chain(rows, op);
      return;
    }
    if (k == "nValues") {
      std::vector<Val> xs = val_list(text_of(n, "list"));
      std::vector<int64_t> except = n.child("except") ? int_list(n.child("except")->text) : std::vector<int64_t>();
      post_cond(n_values(xs, except), condition(text_of(n, "condition")));
      return;
    }
    if (k == "cardinality") {
      std::vector<Val> xs = val_list(text_of(n, "list"));
      const Xml* vs = n.child("values");
      if (!vs || !n.child("occurs")) fail("cardinality needs <values> and <occurs>");
      std::vector<Val> vals = val_list(vs->text);
      // an occurrence is a constant, a variable, or an interval a..b
      std::vector<std::string> occ;
      for (const std::string& w : words(n.child("occurs")->text)) {
        const size_t x = w.find('x');
        if (x != std::string::npos && std::isdigit((unsigned char)w[0]) && w.find('[') == std::string::npos) {  // compact form: 2x3 or 0..1x3
          const int64_t times = std::stoll(w.substr(x + 1));
          for (int64_t q = 0; q < times; ++q) occ.push_back(w.substr(0, x));
        } else if (w.find('[') != std::string::npos) { std::vector<Val> v; expand_ref(w, &v); for (auto& q : v) occ.push_back(xcsp_name(q)); }
        else occ.push_back(w);
      }
      if (occ.size() != vals.size()) fail("cardinality: values and occurs differ in length");
      for (size_t j = 0; j < vals.size(); ++j) {
        Val cnt = count_eq(xs, vals[j]);
        const size_t dd = occ[j].find("..");
        if (dd != std::string::npos) {
          Cond lo; lo.op = "ge"; lo.rhs = konst(std::stoll(occ[j].substr(0, dd)));
          Cond hi; hi.op = "le"; hi.rhs = konst(std::stoll(occ[j].substr(dd + 2)));
          post_cond(cnt, lo); post_cond(cnt, hi);
        } else {
          std::vector<Val> ov; expand_ref(occ[j], &ov);
          if (ov.size() != 1) fail("cardinality: bad occurrence `" + occ[j] + "`");
          Cond c; c.op = "eq"; c.rhs = ov[0];
          post_cond(cnt, c);
        }
      }
      if (vs->get("closed") == "true")
        for (auto& x : xs) { std::vector<Val> any; for (auto& v : vals) any.push_back(cmp("int_eq", x, v)); post_true(any.size() == 1 ? any[0] : apply("or", any)); }
      return;
    }
    if (k == "knapsack") {
      std::vector<Val> xs = val_list(text_of(n, "list"));
      std::vector<int64_t> w = int_list(text_of(n, "weights")), pf = int_list(text_of(n, "profits"));
      if (w.size() != xs.size() || pf.size() != xs.size()) fail("knapsack: list, weights and profits differ in length");
      std::vector<const Xml*> conds;
      for (auto& c : n.kids) if (c->name == "condition" || c->name == "limit") conds.push_back(c.get());
      if (conds.size() != 2) fail("knapsack needs two conditions (weights, then profits)");
      // (a <limit> holds a bare value or variable: the weight may not exceed it)
      auto cond_of = [&](const Xml* c) { std::string t = c->text; t.erase(0, t.find_first_not_of(" \t\r\n")); return condition((c->name == "limit" && !t.empty() && t[0] != '(') ? "(le," + t + ")" : t); };
      post_linear(w, xs, cond_of(conds[0]));
      post_linear(pf, xs, cond_of(conds[1]));
      return;
    }
    if (k == "binPacking") {
      std::vector<Val> xs = val_list(text_of(n, "list"));
      std::vector<int64_t> sz = int_list(text_of(n, "sizes"));
      if (sz.size() != xs.size()) fail("binPacking: list and sizes differ in length");
      int64_t b0 = INT64_MAX, b1 = INT64_MIN;
      for (auto& x : xs) { b0 = std::min(b0, x.lo); b1 = std::max(b1, x.hi); }
      const Xml* lim = n.child("limits"); const Xml* loads = n.child("loads");
      std::vector<Val> per_bin = lim ? val_list(lim->text) : (loads ? val_list(loads->text) : std::vector<Val>());
      if (lim || loads) { b0 = 0; b1 = (int64_t)per_bin.size() - 1; }
      if (b1 - b0 > 10000) fail("binPacking: too many bins");
      for (int64_t b = b0; b <= b1; ++b) {
        std::vector<int64_t> coef; std::vector<Val> lits;
        for (size_t i = 0; i < xs.size(); ++i) { if (xs[i].lo > b || xs[i].hi < b) continue; lits.push_back(as_int(cmp("int_eq", xs[i], konst(b)))); coef.push_back(sz[i]); }
        Cond c;
        if (lim) { c.op = "le"; c.rhs = per_bin[(size_t)(b - b0)]; }
        else if (loads) { c.op = "eq"; c.rhs = per_bin[(size_t)(b - b0)]; }
        else c = condition(text_of(n, "condition"));
        post_linear(coef, lits, c);
      }
      return;
    }
    if (k == "clause") {
      Val r = fresh_bool();
      std::vector<Val> lits;
      for (const std::string& w : words(n.child("list") ? n.child("list")->text : n.text)) {
        if (w.compare(0, 4, "not(") == 0) { std::vector<Val> v; expand_ref(w.substr(4, w.size() - 5), &v); lits.push_back(apply("not", {v[0]})); }
        else { std::vector<Val> v; expand_ref(w, &v); for (auto& q : v) lits.push_back(q); }
      }
      post_true(lits.size() == 1 ? as_bool(lits[0]) : apply("or", lits));
      (void)r;
      return;
    }
    if (k == "circuit") {
      // successor representation with sub-circuit semantics: x_i = i leaves i out; the others form ONE cycle (of <size> nodes when
      // given).  Order encoding: the root is the first node that is in the cycle, every other node of the cycle sits one step behind
      // its successor unless the successor is the root: pos[x_i] = pos[i] + 1 in 1..n-1, pos[root] = 0.
      const Xml* l = n.child("list");
      std::vector<Val> xs = val_list(l ? l->text : n.text);
      const int64_t st = l && !l->get("startIndex").empty() ? std::stoll(l->get("startIndex")) : 0;
      const int64_t nn = (int64_t)xs.size();
      for (size_t i = 0; i < xs.size(); ++i)
        for (size_t j = i + 1; j < xs.size(); ++j) cons << "constraint int_ne(" << xs[i].str() << ", " << xs[j].str() << ");\n";
      for (auto& x : xs) { Cond lo; lo.op = "ge"; lo.rhs = konst(st); Cond hi; hi.op = "le"; hi.rhs = konst(st + nn - 1); post_cond(x, lo); post_cond(x, hi); }
      std::vector<Val> in, root, pos;
      for (int64_t i = 0; i < nn; ++i) in.push_back(cmp("int_ne", xs[(size_t)i], konst(i + st)));
      for (int64_t i = 0; i < nn; ++i) {
        std::vector<Val> c{in[(size_t)i]};
        for (int64_t j = 0; j < i; ++j) c.push_back(apply("not", {in[(size_t)j]}));
        root.push_back(c.size() == 1 ? c[0] : apply("and", c));
        pos.push_back(fresh_int(0, nn - 1));
      }
      post_true(apply("or", in));  // (a circuit, not the empty one)
      for (int64_t i = 0; i < nn; ++i) {
        // root -> pos = 0 ; in and not root -> pos >= 1
        Val p0 = cmp("int_eq", pos[(size_t)i], konst(0));
        Val not_p0 = apply("not", {p0});
        cons << "constraint bool_clause([" << lit(p0) << "], [" << lit(root[(size_t)i]) << "]);\n";
        cons << "constraint bool_clause([" << lit(root[(size_t)i]) << ", " << lit(not_p0) << "], [" << lit(in[(size_t)i]) << "]);\n";
        for (int64_t j = 0; j < nn; ++j) {
          if (j == i || xs[(size_t)i].lo > j + st || xs[(size_t)i].hi < j + st) continue;
          // x_i = j and j is not the root -> pos_j = pos_i + 1
          Val e = cmp("int_eq", xs[(size_t)i], konst(j + st));
          Val step = cmp("int_eq", pos[(size_t)j], apply("add", {pos[(size_t)i], konst(1)}));
          cons << "constraint bool_clause([" << lit(step) << ", " << lit(root[(size_t)j]) << "], [" << lit(e) << "]);\n";
        }
      }
      if (n.child("size")) {
        std::vector<Val> cnt;
        for (auto& b : in) cnt.push_back(as_int(b));
        Cond c; c.op = "eq"; std::vector<Val> sv = val_list(text_of(n, "size")); if (sv.size() != 1) fail("circuit: bad size"); c.rhs = sv[0];
        post_linear(std::vector<int64_t>(cnt.size(), 1), cnt, c);
      }
      return;
    }
    if (k == "slide") {
      // the template constraint applied to every window of the list (arity = number of distinct %i of the template)
      const Xml* l = n.child("list");
      const Xml* tmpl = nullptr;
      for (auto& c : n.kids) if (c->name != "list") { tmpl = c.get(); break; }
      if (!l || !tmpl) fail("slide needs a list and a constraint template");
      std::vector<std::string> items;
      for (auto& v : val_list(l->text)) items.push_back(v.is_const ? std::to_string(v.c) : xcsp_name(v));
      const size_t arity = numbered_params(*tmpl);
      if (arity == 0) fail("slide: the template has no %i argument");
      const int64_t offset = l->get("offset").empty() ? 1 : std::stoll(l->get("offset"));
      if (offset < 1) fail("slide: offset must be at least 1");
      const bool circular = n.get("circular") == "true";
      for (size_t a = 0; circular ? a < items.size() : a + arity <= items.size(); a += (size_t)offset) {
        std::vector<std::string> args;
        for (size_t q = 0; q < arity; ++q) args.push_back(items[(a + q) % items.size()]);
        constraint(*clone_subst(*tmpl, args, arity));
      }
      return;
    }
    fail("unsupported constraint <" + k + ">");
  }

  void objective(const Xml& n) {
    const bool minimize = n.name == "minimize";
    const std::string type = n.get("type", "expression");
    Val obj;
    if (type == "expression") { Tok t{n.text}; obj = as_int(expr(t)); }
    else {
      std::vector<Val> xs = val_list(n.child("list") ? n.child("list")->text : n.text);
      std::vector<int64_t> coef(xs.size(), 1);
      if (n.child("coeffs")) { coef = int_list(n.child("coeffs")->text); if (coef.size() != xs.size()) fail("objective: coeffs and list differ in length"); }
      std::vector<Val> terms;
      for (size_t i = 0; i < xs.size(); ++i) terms.push_back(coef[i] == 1 ? xs[i] : apply("mul", {konst(coef[i]), xs[i]}));
      if (type == "sum") obj = terms.size() == 1 ? as_int(terms[0]) : apply("add", terms);
      else if (type == "product") obj = terms.size() == 1 ? as_int(terms[0]) : apply("mul", terms);
      else if (type == "nValues") obj = n_values(terms, {});
      else if (type == "minimum") obj = apply("min", terms);
      else if (type == "maximum") obj = apply("max", terms);
      else fail("unsupported objective type `" + type + "`");
    }
    if (obj.is_const) { Val v = fresh_int(obj.c, obj.c); obj = v; }
    solve = std::string("solve ") + (minimize ? "minimize " : "maximize ") + obj.var + ";\n";
  }

  std::string run(const Xml& root) {
    if (root.name != "instance") fail("root element must be <instance>");
    if (const Xml* v = root.child("variables")) variables(*v);
    if (const Xml* c = root.child("constraints")) for (auto& k : c->kids) constraint(*k);
    if (const Xml* o = root.child("objectives")) {
      if (o->kids.size() > 1) fail("several objectives are not supported");
      if (!o->kids.empty()) objective(*o->kids[0]);
    }
    return decl.str() + cons.str() + solve;
  }
};

}  // namespace

std::string xcsp3_to_flatzinc(const std::string& xml_text) {
  XmlParser xp(xml_text);
  std::unique_ptr<Xml> root = xp.element();
  Translator tr;
  return tr.run(*root);
}

}  // namespace turbo_front
