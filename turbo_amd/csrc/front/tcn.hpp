// Ternary constraint network produced by the front-end and consumed by the engine's C-ABI.
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/turbo_hip.h"
#include "fzn_ast.hpp"

namespace turbo_front {

struct Term {  // an operand: a constant or a TCN variable
  bool is_const = true;
  int64_t value = 0;  // constant value
  int32_t var = -1;   // TCN variable id
  static Term konst(int64_t v) { Term t; t.is_const = true; t.value = v; return t; }
  static Term variable(int32_t v) { Term t; t.is_const = false; t.var = v; return t; }
};

struct OutputItem {
  std::string name;
  bool is_array = false;
  bool is_bool = false;
  std::vector<std::pair<int64_t, int64_t>> dims;  // output_array([1..2,1..2])
  std::vector<int32_t> vars;                      // TCN variables (constants are interned)
};

struct Strategy {
  int32_t var_order = TB_INPUT_ORDER;
  int32_t val_order = TB_VAL_MIN;
  std::vector<int32_t> vars;  // empty = whole store
  bool emptied = false;       // simplifier: every variable of the strategy was eliminated
};

struct TCN {
  std::vector<tb_itv> store;
  std::vector<tb_prop> props;
  std::vector<std::string> names;  // per TCN variable ("" for temporaries / constants)
  std::vector<Strategy> strategies;
  std::vector<OutputItem> outputs;
  int32_t obj_var = -1;   // variable to minimise
  int32_t goal = 0;       // Solve::Goal of the model
  int32_t goal_var = -1;  // variable named in the solve item
  bool trivially_unsat = false;
  int32_t parsed_variables = 0, parsed_constraints = 0;

  // simplifier state: the network as first lowered, and how to expand a solution of the current network to it
  bool simplified = false;
  std::vector<tb_itv> original_store;
  std::vector<tb_prop> original_props;
  std::vector<int32_t> expand_var;    // original variable -> current variable, or -1
  std::vector<int32_t> expand_const;  // value of an eliminated original variable

  // flattened strategy arrays (rebuilt by flatten_strategies)
  std::vector<int32_t> f_var_order, f_val_order, f_off, f_vars;
  void flatten_strategies();
};

struct LowerError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

TCN lower_to_tcn(const Model& m);
// XCSP3-core instance (XML text) -> FlatZinc text (xcsp3_reader.cpp)
std::string xcsp3_to_flatzinc(const std::string& xml_text);

struct SimplifyInfo {
  int32_t original_vars = 0, original_props = 0, simplified_vars = 0, simplified_props = 0;
  int32_t merged_variables = 0, cse_merges = 0, entailed_props = 0, duplicate_props = 0, eliminated_variables = 0;
};
// In-place simplification (simplify.cpp).  `root_fixpoint` (may be null) is the propagated root store of the
// CURRENT network, computed by the caller.
void simplify_tcn(TCN& t, const tb_itv* root_fixpoint, SimplifyInfo* info);
// Solution of the current network -> store over the variables of the network as first lowered.
void expand_solution(const TCN& t, const tb_itv* simplified, tb_itv* original_out);

}  // namespace turbo_front
