// C-ABI of the front-end (include/turbo_front.h).
#include "../../../include/turbo_front.h"

#include <cstring>
#include <algorithm>
#include <fstream>
#include <map>
#include <memory>
#include <random>
#include <sstream>

#include "tcn.hpp"

using namespace turbo_front;

struct tf_model {
  TCN tcn;
  std::string fcn_stats;  // analyze_cn (common_solving.hpp:669-704) of the model as parsed, one `key=value` per line
};

namespace {

void set_err(char* err, int32_t err_len, const std::string& msg) {
  if (!err || err_len <= 0) return;
  std::strncpy(err, msg.c_str(), (size_t)err_len - 1);
  err[err_len - 1] = '\0';
}

// Statistics of the constraint network before ternarisation (analyze_cn, common_solving.hpp:669-704), on the
// FlatZinc model: symbols are predicate names, a variable occurrence is a reference from a constraint argument.
std::string analyze_model(const Model& m) {
  std::map<std::string, int64_t> var_size;  // declared variables (arrays: number of cells)
  for (const VarDecl& v : m.vars) var_size[v.name] = 1;
  for (const ArrayDecl& a : m.arrays) if (a.is_var) var_size[a.name] = a.size;
  int64_t n_vars = 0;
  for (auto& kv : var_size) n_vars += kv.second;
  int64_t occurrences = 0;
  std::map<std::string, int64_t> symbols, degree_hist;
  struct Walk {
    const std::map<std::string, int64_t>& vs;
    int64_t count(const Expr& e) const {
      int64_t n = 0;
      if (e.kind == Expr::ID) { auto it = vs.find(e.name); if (it != vs.end()) n += it->second; }
      else if (e.kind == Expr::INDEX) { if (vs.count(e.name)) n += 1; }
      for (const Expr& a : e.args) n += count(a);
      return n;
    }
  } walk{var_size};
  for (const Constraint& c : m.constraints) {
    symbols[c.name]++;
    int64_t deg = 0;
    for (const Expr& a : c.args) deg += walk.count(a);
    occurrences += deg;
    degree_hist["('" + c.name + "', " + std::to_string(deg) + ")"]++;
  }
  auto dict = [](const std::map<std::string, int64_t>& d, bool quote) {
    std::string s = "{";
    bool first = true;
    for (auto& kv : d) { s += (first ? "" : ", ") + (quote ? "'" + kv.first + "'" : kv.first) + ": " + std::to_string(kv.second); first = false; }
    return s + "}";
  };
  std::string out;
  out += "fcn_variables=" + std::to_string(n_vars) + "\n";
  out += "fcn_constraints=" + std::to_string(m.constraints.size()) + "\n";
  out += "fcn_var_occurrences=" + std::to_string(occurrences) + "\n";
  out += "fcn_histogram_symbols=\"" + dict(symbols, true) + "\"\n";
  out += "fcn_histogram_constraints_degree=\"" + dict(degree_hist, false) + "\"\n";
  return out;
}

tf_model* build(const std::string& text, char* err, int32_t err_len) {
  try {
    Model m = parse_flatzinc(text);
    std::unique_ptr<tf_model> out(new tf_model);
    out->tcn = lower_to_tcn(m);
    out->fcn_stats = analyze_model(m);
    return out.release();
  } catch (const std::exception& e) {
    set_err(err, err_len, e.what());
    return nullptr;
  }
}

std::string value_text(const tb_itv& d, bool is_bool) {
  // A solution box may leave a variable unassigned when every propagator is already entailed
  // (barebones:971-993): every point of the box is a solution and we print its lower corner.
  int64_t v = d.lb;
  if (is_bool) return v != 0 ? "true" : "false";
  return std::to_string(v);
}

}  // namespace

extern "C" {

tf_model* tf_load_fzn(const char* path, char* err, int32_t err_len) {
  std::ifstream in(path);
  if (!in) { set_err(err, err_len, std::string("Could not open input file ") + (path ? path : "(null)")); return nullptr; }
  std::stringstream ss;
  ss << in.rdbuf();
  return build(ss.str(), err, err_len);
}

tf_model* tf_load_fzn_string(const char* text, char* err, int32_t err_len) { return build(text ? text : "", err, err_len); }

tf_model* tf_load_xcsp3_string(const char* xml, char* err, int32_t err_len) {
  std::string fzn;
  try {
    fzn = xcsp3_to_flatzinc(xml ? xml : "");
  } catch (const std::exception& e) {
    set_err(err, err_len, e.what());
    return nullptr;
  }
  return build(fzn, err, err_len);
}

tf_model* tf_load_xcsp3(const char* path, char* err, int32_t err_len) {
  std::ifstream in(path);
  if (!in) { set_err(err, err_len, std::string("Could not open input file ") + (path ? path : "(null)")); return nullptr; }
  std::stringstream ss;
  ss << in.rdbuf();
  return tf_load_xcsp3_string(ss.str().c_str(), err, err_len);
}

int32_t tf_xcsp3_to_fzn(const char* xml, char* buf, int32_t buf_len, char* err, int32_t err_len) {
  std::string fzn;
  try {
    fzn = xcsp3_to_flatzinc(xml ? xml : "");
  } catch (const std::exception& e) {
    set_err(err, err_len, e.what());
    return -1;
  }
  if (buf && buf_len > 0) {
    const size_t n = std::min((size_t)buf_len - 1, fzn.size());
    std::memcpy(buf, fzn.data(), n);
    buf[n] = '\0';
  }
  return (int32_t)fzn.size();
}

void tf_free(tf_model* m) { delete m; }

int32_t tf_num_vars(const tf_model* m) { return (int32_t)m->tcn.store.size(); }
int32_t tf_num_props(const tf_model* m) { return (int32_t)m->tcn.props.size(); }
const tb_itv* tf_store(const tf_model* m) { return m->tcn.store.data(); }
const tb_prop* tf_props(const tf_model* m) { return m->tcn.props.data(); }
int32_t tf_num_strategies(const tf_model* m) { return (int32_t)m->tcn.strategies.size(); }
const int32_t* tf_strat_var_order(const tf_model* m) { return m->tcn.f_var_order.data(); }
const int32_t* tf_strat_val_order(const tf_model* m) { return m->tcn.f_val_order.data(); }
const int32_t* tf_strat_off(const tf_model* m) { return m->tcn.f_off.data(); }
const int32_t* tf_strat_vars(const tf_model* m) { return m->tcn.f_vars.data(); }

int32_t tf_push_eps_strategy(tf_model* m, int32_t var_order, int32_t val_order) {
  if (var_order < TB_INPUT_ORDER || var_order > TB_LARGEST || val_order < TB_VAL_MIN || val_order > TB_VAL_REVERSE_SPLIT) return -1;
  Strategy s;
  s.var_order = var_order; s.val_order = val_order;
  // The EPS strategy ranges over the variables of the model's first search annotation (whole store if none).
  if (m->tcn.strategies.size() > 1) s.vars = m->tcn.strategies.front().vars;
  m->tcn.strategies.insert(m->tcn.strategies.begin(), s);
  m->tcn.flatten_strategies();
  return 0;
}

int32_t tf_obj_var(const tf_model* m) { return m->tcn.obj_var; }
int32_t tf_goal(const tf_model* m) { return m->tcn.goal; }
int32_t tf_goal_var(const tf_model* m) { return m->tcn.goal_var; }
int32_t tf_trivially_unsat(const tf_model* m) { return m->tcn.trivially_unsat ? 1 : 0; }
int32_t tf_parsed_variables(const tf_model* m) { return m->tcn.parsed_variables; }
int32_t tf_parsed_constraints(const tf_model* m) { return m->tcn.parsed_constraints; }

int64_t tf_objective_of(const tf_model* m, const tb_itv* store) {
  if (m->tcn.goal_var < 0) return 0;
  const tb_itv d = store[m->tcn.goal_var];
  return m->tcn.goal == 2 ? d.ub : d.lb;
}

int32_t tf_simplify(tf_model* m, const tb_itv* root_fixpoint, int32_t* stats_out) {
  SimplifyInfo info;
  try {
    simplify_tcn(m->tcn, root_fixpoint, &info);
  } catch (const std::exception&) {
    return -1;
  }
  if (stats_out) {
    const int32_t v[9] = {info.original_vars, info.original_props, info.simplified_vars, info.simplified_props, info.merged_variables,
                          info.cse_merges, info.entailed_props, info.duplicate_props, info.eliminated_variables};
    std::memcpy(stats_out, v, sizeof(v));
  }
  return 0;
}
int32_t tf_original_num_vars(const tf_model* m) { return (int32_t)(m->tcn.simplified ? m->tcn.original_store.size() : m->tcn.store.size()); }
int32_t tf_original_num_props(const tf_model* m) { return (int32_t)(m->tcn.simplified ? m->tcn.original_props.size() : m->tcn.props.size()); }
const tb_itv* tf_original_store(const tf_model* m) { return m->tcn.simplified ? m->tcn.original_store.data() : m->tcn.store.data(); }
const tb_prop* tf_original_props(const tf_model* m) { return m->tcn.simplified ? m->tcn.original_props.data() : m->tcn.props.data(); }
int32_t tf_expand_solution(const tf_model* m, const tb_itv* store, tb_itv* original_out) {
  expand_solution(m->tcn, store, original_out);
  return 0;
}

int32_t tf_format_solution(const tf_model* m, const tb_itv* store_in, char* buf, int32_t buf_len) {
  std::vector<tb_itv> expanded;
  const tb_itv* store = store_in;
  if (m->tcn.simplified) {  // output items name variables of the network as first lowered
    expanded.resize(m->tcn.original_store.size());
    expand_solution(m->tcn, store_in, expanded.data());
    store = expanded.data();
  }
  std::string out;
  for (const OutputItem& o : m->tcn.outputs) {
    if (!o.is_array) {
      out += o.name + " = " + value_text(store[o.vars[0]], o.is_bool) + ";\n";
      continue;
    }
    out += o.name + " = array" + std::to_string(o.dims.size()) + "d(";
    for (auto& d : o.dims) out += std::to_string(d.first) + ".." + std::to_string(d.second) + ", ";
    out += "[";
    for (size_t i = 0; i < o.vars.size(); ++i) {
      if (i) out += ", ";
      out += value_text(store[o.vars[i]], o.is_bool);
    }
    out += "]);\n";
  }
  if (buf && buf_len > 0) {
    size_t n = std::min((size_t)buf_len - 1, out.size());
    std::memcpy(buf, out.data(), n);
    buf[n] = '\0';
  }
  return (int32_t)out.size();
}

const char* tf_fcn_statistics(const tf_model* m) { return m->fcn_stats.c_str(); }

int32_t tf_shuffle_strategy(tf_model* m, int32_t strategy, uint64_t seed) {
  if (strategy < 0 || strategy >= (int32_t)m->tcn.strategies.size()) return -1;
  Strategy& s = m->tcn.strategies[(size_t)strategy];
  if (s.vars.empty()) {  // whole store: materialise it so that it can be permuted
    for (int32_t v = 0; v < (int32_t)m->tcn.store.size(); ++v) s.vars.push_back(v);
  }
  std::mt19937 gen((uint32_t)seed);  // std::mt19937 random_generator(config.seed), common_solving.hpp:632
  std::shuffle(s.vars.begin(), s.vars.end(), gen);
  s.var_order = TB_INPUT_ORDER;
  m->tcn.flatten_strategies();
  return 0;
}

const char* tf_var_name(const tf_model* m, int32_t var) {
  if (var < 0 || var >= (int32_t)m->tcn.names.size()) return "";
  return m->tcn.names[(size_t)var].c_str();
}

}  // extern "C"
