/*
 * turbo_front.h -- C-ABI of the host front-end (libturbo_front.so): FlatZinc -> ternary constraint
 * network (TCN).  Prerequisite of the hot path, not the hot path itself.
 *
 * Stands in for `AbstractDomains::preprocess()` of the reference (include/common_solving.hpp:605-637)
 * restricted to its `-disable_simplify` pipeline (common_solving.hpp:520-535): parse_flatzinc
 * (lala-parsing, absent), maximize->minimize rewrite (common_solving.hpp:489-510), ternarize with the
 * constants {0,1,2} pre-interned (common_solving.hpp:521), default first_fail/indomain_min strategy
 * over the whole store (common_solving.hpp:640-650).
 */
#ifndef TURBO_FRONT_H
#define TURBO_FRONT_H

#include <stdint.h>
#include "turbo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tf_model tf_model;

/* Parse + lower.  Returns NULL on error and writes a message into err (if non-NULL). */
tf_model* tf_load_fzn(const char* path, char* err, int32_t err_len);
tf_model* tf_load_fzn_string(const char* text, char* err, int32_t err_len);
/* XCSP3-core input (common_solving.hpp:409-413, `parse_xcsp3` of lala-parsing): the instance is rewritten to FlatZinc
 * (tf_xcsp3_to_fzn returns the text length, -1 on error) and then takes the same path as a .fzn file. */
tf_model* tf_load_xcsp3(const char* path, char* err, int32_t err_len);
tf_model* tf_load_xcsp3_string(const char* xml, char* err, int32_t err_len);
int32_t tf_xcsp3_to_fzn(const char* xml, char* buf, int32_t buf_len, char* err, int32_t err_len);
void tf_free(tf_model* m);

int32_t tf_num_vars(const tf_model* m);
int32_t tf_num_props(const tf_model* m);
const tb_itv* tf_store(const tf_model* m);   /* root store, tf_num_vars entries */
const tb_prop* tf_props(const tf_model* m);  /* tf_num_props entries, 16-B aligned */

/* strategies, flattened like tb_solve expects; the default whole-store strategy is last */
int32_t tf_num_strategies(const tf_model* m);
const int32_t* tf_strat_var_order(const tf_model* m);
const int32_t* tf_strat_val_order(const tf_model* m);
const int32_t* tf_strat_off(const tf_model* m);   /* n_strats + 1 entries */
const int32_t* tf_strat_vars(const tf_model* m);

/* Insert an EPS strategy at position 0 (split->push_eps_strategy, common_solving.hpp:652-667). */
int32_t tf_push_eps_strategy(tf_model* m, int32_t var_order, int32_t val_order);

int32_t tf_obj_var(const tf_model* m);        /* TCN variable to MINIMISE, -1 for satisfy */
int32_t tf_goal(const tf_model* m);           /* 0 satisfy, 1 minimize, 2 maximize (as written in the model) */
int32_t tf_goal_var(const tf_model* m);       /* TCN variable named in the solve item (-1 for satisfy) */
int32_t tf_trivially_unsat(const tf_model* m);/* store was bot after interpretation (cpu_solving.hpp:14) */
int32_t tf_parsed_variables(const tf_model* m);
int32_t tf_parsed_constraints(const tf_model* m);

/* Objective value as the reference prints it (statistics.hpp:378-388): lb of the goal var when minimising,
 * ub when maximising. */
int64_t tf_objective_of(const tf_model* m, const tb_itv* store);

/* Format a solution like lala's SolverOutput (output_var / output_array annotations), without the
 * `----------` separator.  Returns the number of bytes needed (excluding NUL). */
int32_t tf_format_solution(const tf_model* m, const tb_itv* store, char* buf, int32_t buf_len);

/*
 * TCN simplifier: the preprocessing loop of common_solving.hpp:537-585 (root fixpoint -> equivalence classes,
 * algebraic simplification, entailed-constraint elimination, common subexpressions, useless variables), in place.
 * `root_fixpoint` is the propagated root store of the CURRENT network (tf_num_vars entries) computed by the caller
 * (the GPU engine: `tb_propagate`), or NULL.  May be called repeatedly (propagate, simplify, propagate, ...).
 * stats_out (may be NULL) receives 9 int32: original vars/props, simplified vars/props, merged variables,
 * common-subexpression merges, entailed propagators, duplicate propagators, eliminated variables.
 * After the call tf_store/tf_props/... describe the simplified network; tf_format_solution / tf_objective_of /
 * tf_expand_solution take solutions of the simplified network.
 */
int32_t tf_simplify(tf_model* m, const tb_itv* root_fixpoint, int32_t* stats_out);
int32_t tf_original_num_vars(const tf_model* m);
int32_t tf_original_num_props(const tf_model* m);
const tb_itv* tf_original_store(const tf_model* m);
const tb_prop* tf_original_props(const tf_model* m);
/* solution of the current network -> store over the variables of the network as first lowered */
int32_t tf_expand_solution(const tf_model* m, const tb_itv* store, tb_itv* original_out);

/* Statistics of the model before ternarisation (analyze_cn, common_solving.hpp:669-704): `key=value` lines
 * (fcn_variables, fcn_constraints, fcn_var_occurrences, fcn_histogram_symbols, fcn_histogram_constraints_degree). */
const char* tf_fcn_statistics(const tf_model* m);

/* `-eps_var_order random`: strategy `strategy` becomes INPUT_ORDER over its variables shuffled with
 * std::mt19937(seed) (split->shuffle_random_strategies, common_solving.hpp:632-633). */
int32_t tf_shuffle_strategy(tf_model* m, int32_t strategy, uint64_t seed);

/* name of a TCN variable ("" for temporaries), for debugging */
const char* tf_var_name(const tf_model* m, int32_t var);

#ifdef __cplusplus
}
#endif
#endif
