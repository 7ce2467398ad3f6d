// XCSP3-core reader: XML instance -> FlatZinc text, which then goes through the same parser and lowering as a
// .fzn input (one lowering path).  Stands in for lala-parsing's `parse_xcsp3` (common_solving.hpp:409-413; the
// library is absent from the reference tree), for the constraint forms listed below.
//
//   variables    <var>, <array> (uniform domain: ranges and value lists; multi-dimensional sizes)
//   constraints  <intension>, <extension> (supports / conflicts, `*`), <allDifferent>, <sum>, <cumulative>
//                (constant lengths and heights), <element>, <minimum>, <maximum>, <ordered>, <instantiation>,
//                <allEqual>, <count>, <noOverlap> (one dimension), <channel> (one or two lists),
//                <group> with %i arguments, <block>
//   objectives   <minimize> / <maximize> of type expression, sum, minimum, maximum (optional <coeffs>)
//
// Decompositions are the textbook ones (time-indexed cumulative, pairwise allDifferent, tuple-wise tables); the
// only thing the reference pins for this format is the objective of benchmarks/test_data/cumulative.xml.
#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <fstream>
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
      } else if (k->name == "array") {
        if (!k->kids.empty()) fail("arrays with per-cell <domain> are not supported");
        ArrayInfo info;
        const std::string sz = k->get("size");
        for (size_t i = 0; i < sz.size(); ++i)
          if (sz[i] == '[') { size_t e = sz.find(']', i); info.dims.push_back(std::stoll(sz.substr(i + 1, e - i - 1))); i = e; }
        if (info.dims.empty()) fail("array without size");
        int64_t total = 1;
        for (int64_t d : info.dims) total *= d;
        if (total <= 0 || total > 10000000) fail("bad array size");
        std::string fz; int64_t lo, hi;
        domain_text(k->text, &fz, &lo, &hi);
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
          if (holes) cons << "constraint set_in(" << en << ", " << fz << ");\n";
        }
        arrays[id] = info;
      } else fail("unknown declaration <" + k->name + ">");
    }
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
      Val r = fresh_bool();
      cons << "constraint array_bool_" << f << "([";
      for (size_t i = 0; i < a.size(); ++i) cons << (i ? ", " : "") << lit(as_bool(a[i]));
      cons << "], " << r.var << ");\n";
      return r;
    }
    if (f == "xor") { return fold([&](Val x, Val y) { Val r = fresh_bool(); cons << "constraint bool_xor(" << lit(as_bool(x)) << ", " << lit(as_bool(y)) << ", " << r.var << ");\n"; return r; }); }
    if (f == "iff") { need(2); Val r = fresh_bool(); cons << "constraint bool_eq_reif(" << lit(as_bool(a[0])) << ", " << lit(as_bool(a[1])) << ", " << r.var << ");\n"; return r; }
    if (f == "imp") { need(2); Val r = fresh_bool(); cons << "constraint bool_le_reif(" << lit(as_bool(a[0])) << ", " << lit(as_bool(a[1])) << ", " << r.var << ");\n"; return r; }
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
  struct Cond { std::string op; Val rhs; };
  Cond condition(const std::string& t) {
    std::string s = t;
    for (char& c : s) if (c == '(' || c == ')' || c == ',') c = ' ';
    const auto ws = words(s);
    if (ws.size() != 2) fail("unsupported condition `" + t + "`");
    Cond c; c.op = ws[0];
    std::vector<Val> v; expand_ref(ws[1], &v);
    if (v.size() != 1) fail("condition operand must be a single value");
    c.rhs = v[0];
    return c;
  }
  void post_linear(std::vector<int64_t> coef, std::vector<Val> xs, const Cond& c) {
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

  static std::string subst(std::string t, const std::vector<std::string>& args) {
    for (size_t i = args.size(); i-- > 0;) {
      const std::string key = "%" + std::to_string(i);
      for (size_t q; (q = t.find(key)) != std::string::npos;) t.replace(q, key.size(), args[i]);
    }
    if (t.find("%...") != std::string::npos) {
      std::string all;
      for (auto& a : args) all += a + " ";
      for (size_t q; (q = t.find("%...")) != std::string::npos;) t.replace(q, 4, all);
    }
    return t;
  }
  static std::unique_ptr<Xml> clone_subst(const Xml& n, const std::vector<std::string>& args) {
    auto c = std::make_unique<Xml>();
    c->name = n.name; c->attr = n.attr; c->text = subst(n.text, args);
    for (auto& k : n.kids) c->kids.push_back(clone_subst(*k, args));
    return c;
  }
  static std::string text_of(const Xml& n, const std::string& child) {
    const Xml* c = n.child(child);
    return c ? c->text : std::string();
  }

  void constraint(const Xml& n) {
    const std::string& k = n.name;
    if (k == "block") { for (auto& c : n.kids) constraint(*c); return; }
    if (k == "group") {
      const Xml* tmpl = nullptr;
      for (auto& c : n.kids) if (c->name != "args") { tmpl = c.get(); break; }
      if (!tmpl) fail("group without a constraint template");
      for (auto& c : n.kids) if (c->name == "args") constraint(*clone_subst(*tmpl, words(c->text)));
      return;
    }
    if (k == "intension") {
      const std::string src = n.child("function") ? n.child("function")->text : n.text;
      Tok t{src};
      post_true(as_bool(expr(t)));
      return;
    }
    if (k == "allDifferent") {
      if (n.child("except") || n.child("matrix")) fail("allDifferent with <except>/<matrix> is not supported");
      std::vector<Val> xs = val_list(n.child("list") ? n.child("list")->text : n.text);
      for (size_t i = 0; i < xs.size(); ++i)
        for (size_t j = i + 1; j < xs.size(); ++j) cons << "constraint int_ne(" << xs[i].str() << ", " << xs[j].str() << ");\n";
      return;
    }
    if (k == "ordered") {
      std::vector<Val> xs = val_list(text_of(n, "list"));
      std::string op = words(text_of(n, "operator")).empty() ? "le" : words(text_of(n, "operator"))[0];
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
      if (n.child("coeffs")) { coef = int_list(n.child("coeffs")->text); if (coef.size() != xs.size()) fail("sum: coeffs and list differ in length"); }
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
      std::vector<Val> rows;
      for (auto& tp : tuples) {
        std::vector<Val> lits;
        for (size_t j = 0; j < xs.size(); ++j) {
          if (tp[j] == "*") continue;
          if (!is_int(tp[j])) fail("table values must be integers or *");
          lits.push_back(cmp(support ? "int_eq" : "int_ne", xs[j], konst(std::stoll(tp[j]))));
        }
        if (support) rows.push_back(lits.empty() ? konst(1) : (lits.size() == 1 ? lits[0] : apply("and", lits)));
        else if (lits.empty()) post_true(konst(0));
        else post_true(lits.size() == 1 ? lits[0] : apply("or", lits));
      }
      if (support) post_true(rows.empty() ? konst(0) : (rows.size() == 1 ? rows[0] : apply("or", rows)));
      return;
    }
    if (k == "cumulative") {
      std::vector<Val> s = val_list(text_of(n, "origins")), len = val_list(text_of(n, "lengths")), h = val_list(text_of(n, "heights"));
      if (n.child("ends") || n.child("machines")) fail("cumulative with <ends>/<machines> is not supported");
      if (s.size() != len.size() || s.size() != h.size()) fail("cumulative: origins, lengths and heights differ in length");
      for (size_t i = 0; i < s.size(); ++i) if (!len[i].is_const || !h[i].is_const) fail("cumulative: variable lengths or heights are not supported");
      const Cond c = condition(text_of(n, "condition"));
      int64_t t0 = INT64_MAX, t1 = INT64_MIN;
      for (size_t i = 0; i < s.size(); ++i) { if (len[i].c <= 0 || h[i].c == 0) continue; t0 = std::min(t0, s[i].lo); t1 = std::max(t1, s[i].hi + len[i].c - 1); }
      if (t0 > t1) return;
      if (t1 - t0 > 100000) fail("cumulative: horizon too long for the time-indexed decomposition");
      for (int64_t t = t0; t <= t1; ++t) {  // sum_i h_i * [s_i <= t < s_i + l_i]  (cond)  limit
        std::vector<int64_t> coef; std::vector<Val> lits;
        for (size_t i = 0; i < s.size(); ++i) {
          if (len[i].c <= 0 || h[i].c == 0 || s[i].lo > t || s[i].hi + len[i].c - 1 < t) continue;
          Val a = cmp("int_le", s[i], konst(t)), b = cmp("int_le", konst(t - len[i].c + 1), s[i]);
          lits.push_back(as_int(apply("and", {a, b})));
          coef.push_back(h[i].c);
        }
        if (!lits.empty()) post_linear(coef, lits, c);
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
    if (k == "noOverlap") {  // one dimension, constant or variable lengths: s_i + l_i <= s_j  or  s_j + l_j <= s_i
      std::vector<Val> o = val_list(text_of(n, "origins")), len = val_list(text_of(n, "lengths"));
      if (o.size() != len.size()) fail("noOverlap: origins and lengths differ in length");
      if (text_of(n, "origins").find('(') != std::string::npos) fail("noOverlap in several dimensions is not supported");
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
