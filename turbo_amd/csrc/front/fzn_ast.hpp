// FlatZinc abstract syntax used by the front-end.
// Vocabulary: the subset exercised by the reference's fixtures, including the
// non-standard forms lala's parser accepts: nested predicate calls `int_eq(b, int_le(0,y))`,
// indexed access `varr[1]`, constants in search arrays, arrays of variables without initialiser.
#pragma once

#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace turbo_front {

struct Expr {
  enum Kind { INT, BOOL, ID, ARRAY, RANGE, SETLIT, CALL, INDEX, STRING };
  Kind kind = INT;
  int64_t ival = 0;   // INT / BOOL value, RANGE lower bound, INDEX index
  int64_t ival2 = 0;  // RANGE upper bound
  std::string name;   // ID, CALL name, INDEX array name, STRING text
  std::vector<Expr> args;  // ARRAY / SETLIT elements, CALL arguments
};

struct VarDecl {
  std::string name;
  bool is_bool = false;
  bool is_set_var = false;  // `var set of ...` (unsupported, only unsolved_bugs_data/valve6.fzn)
  bool has_dom = false;
  int64_t lb = 0, ub = 0;
  std::vector<int64_t> set_values;  // `var {1,2,4,5}: x`
  std::vector<Expr> anns;
  bool has_init = false;
  Expr init;
};

struct ArrayDecl {
  std::string name;
  bool is_var = false;
  bool elem_bool = false;
  bool elem_set = false;
  bool elem_has_dom = false;
  int64_t elem_lb = 0, elem_ub = 0;
  int64_t size = 0;
  std::vector<Expr> anns;
  bool has_init = false;
  std::vector<Expr> elems;
};

struct Constraint {
  std::string name;
  std::vector<Expr> args;
  std::vector<Expr> anns;
};

struct Solve {
  enum Goal { SATISFY = 0, MINIMIZE = 1, MAXIMIZE = 2 };
  Goal goal = SATISFY;
  Expr objective;
  std::vector<Expr> anns;
};

struct Model {
  // declaration order matters for output and for variable numbering
  struct Item { enum Kind { VAR, ARRAY } kind; size_t index; };
  std::vector<Item> order;
  std::vector<VarDecl> vars;
  std::vector<ArrayDecl> arrays;
  std::map<std::string, Expr> params;  // scalar int/bool/set parameters
  std::vector<Constraint> constraints;
  Solve solve;
  bool has_solve = false;
};

struct ParseError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

Model parse_flatzinc(const std::string& text);

}  // namespace turbo_front
