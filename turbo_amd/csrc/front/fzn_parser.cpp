// Recursive-descent FlatZinc parser (own implementation; the reference delegates to lala-parsing's
// PEG grammar, include/common_solving.hpp:404-439, which is not part of the reference tree).
#include "fzn_ast.hpp"

#include <cctype>
#include <cstdlib>

namespace turbo_front {
namespace {

struct Tok {
  enum Kind { END, IDENT, INT, STR, PUNCT } kind = END;
  std::string text;  // IDENT / STR / PUNCT
  int64_t ival = 0;
};

class Lexer {
 public:
  explicit Lexer(const std::string& s) : s_(s) { advance(); }
  const Tok& peek() const { return cur_; }
  Tok next() { Tok t = cur_; advance(); return t; }
  bool is_punct(const char* p) const { return cur_.kind == Tok::PUNCT && cur_.text == p; }
  bool is_ident(const char* p) const { return cur_.kind == Tok::IDENT && cur_.text == p; }
  bool accept_punct(const char* p) { if (is_punct(p)) { advance(); return true; } return false; }
  bool accept_ident(const char* p) { if (is_ident(p)) { advance(); return true; } return false; }
  void expect_punct(const char* p) {
    if (!accept_punct(p)) fail(std::string("expected `") + p + "`");
  }
  void expect_ident(const char* p) {
    if (!accept_ident(p)) fail(std::string("expected `") + p + "`");
  }
  [[noreturn]] void fail(const std::string& msg) const {
    throw ParseError("FlatZinc parse error at line " + std::to_string(line_) + ": " + msg + " (got `" +
                     (cur_.kind == Tok::END ? std::string("<eof>") : (cur_.kind == Tok::INT ? std::to_string(cur_.ival) : cur_.text)) + "`)");
  }

 private:
  void skip_ws() {
    for (;;) {
      while (pos_ < s_.size() && std::isspace((unsigned char)s_[pos_])) { if (s_[pos_] == '\n') ++line_; ++pos_; }
      if (pos_ < s_.size() && s_[pos_] == '%') { while (pos_ < s_.size() && s_[pos_] != '\n') ++pos_; continue; }
      break;
    }
  }
  void advance() {
    skip_ws();
    cur_ = Tok();
    if (pos_ >= s_.size()) { cur_.kind = Tok::END; return; }
    char c = s_[pos_];
    if (std::isalpha((unsigned char)c) || c == '_') {
      size_t b = pos_;
      while (pos_ < s_.size() && (std::isalnum((unsigned char)s_[pos_]) || s_[pos_] == '_')) ++pos_;
      cur_.kind = Tok::IDENT; cur_.text = s_.substr(b, pos_ - b);
      return;
    }
    if (std::isdigit((unsigned char)c) || ((c == '-' || c == '+') && pos_ + 1 < s_.size() && std::isdigit((unsigned char)s_[pos_ + 1]))) {
      size_t b = pos_;
      ++pos_;
      while (pos_ < s_.size() && std::isdigit((unsigned char)s_[pos_])) ++pos_;
      // a float literal `1.5` (but not the range `1..5`)
      if (pos_ + 1 < s_.size() && s_[pos_] == '.' && std::isdigit((unsigned char)s_[pos_ + 1]))
        throw ParseError("FlatZinc parse error at line " + std::to_string(line_) + ": floating point literals are not supported");
      cur_.kind = Tok::INT;
      cur_.ival = std::strtoll(s_.substr(b, pos_ - b).c_str(), nullptr, 10);
      return;
    }
    if (c == '"') {
      size_t b = ++pos_;
      while (pos_ < s_.size() && s_[pos_] != '"') ++pos_;
      cur_.kind = Tok::STR; cur_.text = s_.substr(b, pos_ - b);
      if (pos_ < s_.size()) ++pos_;
      return;
    }
    cur_.kind = Tok::PUNCT;
    if (c == ':' && pos_ + 1 < s_.size() && s_[pos_ + 1] == ':') { cur_.text = "::"; pos_ += 2; return; }
    if (c == '.' && pos_ + 1 < s_.size() && s_[pos_ + 1] == '.') { cur_.text = ".."; pos_ += 2; return; }
    cur_.text = std::string(1, c);
    ++pos_;
  }
  const std::string& s_;
  size_t pos_ = 0;
  int line_ = 1;
  Tok cur_;
};

class Parser {
 public:
  explicit Parser(const std::string& text) : lx_(text) {}

  Model parse() {
    Model m;
    while (lx_.peek().kind != Tok::END) parse_item(m);
    if (!m.has_solve) throw ParseError("FlatZinc parse error: missing solve item");
    return m;
  }

 private:
  Lexer lx_;

  int64_t expect_int() {
    if (lx_.peek().kind != Tok::INT) lx_.fail("expected an integer");
    return lx_.next().ival;
  }
  std::string expect_name() {
    if (lx_.peek().kind != Tok::IDENT) lx_.fail("expected an identifier");
    return lx_.next().text;
  }

  // expr := int | bool | ident | ident[int] | ident(args) | [ exprs ] | { ints } | int..int | "str"
  Expr parse_expr() {
    Expr e;
    const Tok& t = lx_.peek();
    if (t.kind == Tok::INT) {
      e.kind = Expr::INT; e.ival = lx_.next().ival;
      if (lx_.accept_punct("..")) { e.kind = Expr::RANGE; e.ival2 = expect_int(); }
      return e;
    }
    if (t.kind == Tok::STR) { e.kind = Expr::STRING; e.name = lx_.next().text; return e; }
    if (t.kind == Tok::IDENT) {
      std::string id = lx_.next().text;
      if (id == "true" || id == "false") { e.kind = Expr::BOOL; e.ival = (id == "true"); return e; }
      if (lx_.accept_punct("(")) {
        e.kind = Expr::CALL; e.name = id;
        if (!lx_.accept_punct(")")) {
          do { e.args.push_back(parse_expr()); } while (lx_.accept_punct(","));
          lx_.expect_punct(")");
        }
        return e;
      }
      if (lx_.accept_punct("[")) {
        e.kind = Expr::INDEX; e.name = id; e.ival = expect_int();
        lx_.expect_punct("]");
        return e;
      }
      e.kind = Expr::ID; e.name = id;
      return e;
    }
    if (lx_.accept_punct("[")) {
      e.kind = Expr::ARRAY;
      if (!lx_.accept_punct("]")) {
        do { if (lx_.is_punct("]")) break; e.args.push_back(parse_expr()); } while (lx_.accept_punct(","));
        lx_.expect_punct("]");
      }
      return e;
    }
    if (lx_.accept_punct("{")) {
      e.kind = Expr::SETLIT;
      if (!lx_.accept_punct("}")) {
        do { e.args.push_back(parse_expr()); } while (lx_.accept_punct(","));
        lx_.expect_punct("}");
      }
      return e;
    }
    lx_.fail("expected an expression");
  }

  std::vector<Expr> parse_annotations() {
    std::vector<Expr> anns;
    while (lx_.accept_punct("::")) anns.push_back(parse_expr());
    return anns;
  }

  // Scalar type after an optional `var`.  Fills the domain fields.
  struct Type { bool is_bool = false, is_set = false, has_dom = false; int64_t lb = 0, ub = 0; std::vector<int64_t> values; };
  Type parse_type() {
    Type ty;
    if (lx_.accept_ident("set")) { lx_.expect_ident("of"); ty = parse_type(); ty.is_set = true; return ty; }
    if (lx_.accept_ident("bool")) { ty.is_bool = true; ty.has_dom = true; ty.lb = 0; ty.ub = 1; return ty; }
    if (lx_.accept_ident("int")) return ty;
    if (lx_.is_ident("float")) throw ParseError("FlatZinc parse error: float variables are not supported");
    if (lx_.peek().kind == Tok::INT) {
      ty.lb = expect_int(); lx_.expect_punct(".."); ty.ub = expect_int(); ty.has_dom = true;
      return ty;
    }
    if (lx_.accept_punct("{")) {
      ty.has_dom = true;
      if (!lx_.accept_punct("}")) {
        do { ty.values.push_back(expect_int()); } while (lx_.accept_punct(","));
        lx_.expect_punct("}");
      }
      if (ty.values.empty()) { ty.lb = 1; ty.ub = 0; }
      else { ty.lb = ty.values.front(); ty.ub = ty.values.front(); for (auto v : ty.values) { if (v < ty.lb) ty.lb = v; if (v > ty.ub) ty.ub = v; } }
      return ty;
    }
    lx_.fail("expected a type");
  }

  void parse_item(Model& m) {
    if (lx_.accept_ident("predicate")) {  // declarations of solver predicates carry no information
      while (lx_.peek().kind != Tok::END && !lx_.is_punct(";")) lx_.next();
      lx_.expect_punct(";");
      return;
    }
    if (lx_.accept_ident("constraint")) {
      Expr e = parse_expr();
      if (e.kind != Expr::CALL) lx_.fail("expected a predicate call after `constraint`");
      Constraint c;
      c.name = e.name; c.args = std::move(e.args);
      c.anns = parse_annotations();
      lx_.expect_punct(";");
      m.constraints.push_back(std::move(c));
      return;
    }
    if (lx_.accept_ident("solve")) {
      m.solve.anns = parse_annotations();
      if (lx_.accept_ident("satisfy")) m.solve.goal = Solve::SATISFY;
      else if (lx_.accept_ident("minimize")) { m.solve.goal = Solve::MINIMIZE; m.solve.objective = parse_expr(); }
      else if (lx_.accept_ident("maximize")) { m.solve.goal = Solve::MAXIMIZE; m.solve.objective = parse_expr(); }
      else lx_.fail("expected satisfy, minimize or maximize");
      lx_.expect_punct(";");
      m.has_solve = true;
      return;
    }
    if (lx_.accept_ident("array")) {
      ArrayDecl a;
      lx_.expect_punct("[");
      int64_t lo = expect_int(); lx_.expect_punct(".."); int64_t hi = expect_int();
      lx_.expect_punct("]");
      if (lo != 1) lx_.fail("array index sets must start at 1");
      a.size = hi < 0 ? 0 : hi;
      lx_.expect_ident("of");
      a.is_var = lx_.accept_ident("var");
      Type ty = parse_type();
      a.elem_bool = ty.is_bool; a.elem_set = ty.is_set; a.elem_has_dom = ty.has_dom; a.elem_lb = ty.lb; a.elem_ub = ty.ub;
      lx_.expect_punct(":");
      a.name = expect_name();
      a.anns = parse_annotations();
      if (lx_.accept_punct("=")) {
        Expr init = parse_expr();
        if (init.kind != Expr::ARRAY) lx_.fail("expected an array literal");
        a.has_init = true; a.elems = std::move(init.args);
      }
      lx_.expect_punct(";");
      m.order.push_back({Model::Item::ARRAY, m.arrays.size()});
      m.arrays.push_back(std::move(a));
      return;
    }
    if (lx_.accept_ident("var")) {
      Type ty = parse_type();
      VarDecl v;
      v.is_bool = ty.is_bool; v.is_set_var = ty.is_set; v.has_dom = ty.has_dom; v.lb = ty.lb; v.ub = ty.ub; v.set_values = std::move(ty.values);
      lx_.expect_punct(":");
      v.name = expect_name();
      v.anns = parse_annotations();
      if (lx_.accept_punct("=")) { v.has_init = true; v.init = parse_expr(); }
      lx_.expect_punct(";");
      m.order.push_back({Model::Item::VAR, m.vars.size()});
      m.vars.push_back(std::move(v));
      return;
    }
    // scalar parameter: `int: n = 3;`  `bool: b = true;`  `set of int: s = 1..3;`  `1..5: k = 2;`
    {
      (void)parse_type();
      lx_.expect_punct(":");
      std::string name = expect_name();
      (void)parse_annotations();
      lx_.expect_punct("=");
      Expr val = parse_expr();
      lx_.expect_punct(";");
      m.params[name] = std::move(val);
    }
  }
};

}  // namespace

Model parse_flatzinc(const std::string& text) {
  Parser p(text);
  return p.parse();
}

}  // namespace turbo_front
