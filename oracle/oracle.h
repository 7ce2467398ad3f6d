/*
 * oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the reference's sequential propagate-and-search path
 * (ptal/turbo, include/cpu_solving.hpp:8-48) over the ternary constraint
 * network (TCN) that crosses the drop-in boundary, with the decision / rope /
 * incumbent rules of include/barebones_dive_and_solve.hpp:187-405,656-886,903-1031
 * so that the oracle and the HIP engine explore the same tree.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Nothing under turbo_amd/ links, imports or executes it.
 *
 * PARITY PINNING: the arithmetic of `deduce`/`ask` lives in lattice-land/lala-pc
 * @ v1.2.8 (reference CMakeLists.txt:46-50), which is NOT vendored in
 * /root/reference and not present in this container.  The propagator rules
 * below are therefore a restatement of the published interval (bounds)
 * propagation rules for `x = y op z`; at the deduce level parity is UNPINNED.
 * At the end-to-end level the oracle IS pinned: it reproduces every expected
 * objective of the reference's own known-answer table
 * (benchmarks/test_list.csv:1-32, driven by test_turbo.sh:34-67) --
 * see tests/test_oracle_golden.py.
 */
#ifndef TURBO_ORACLE_H
#define TURBO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NINF INT32_MIN /* -infinity sentinel (lala ZLB bottom), common_solving.hpp:45-54 */
#define ORC_PINF INT32_MAX /* +infinity sentinel */

/* TCN operators `x = y op z` (reference: bytecode {op,x,y,z}, common_solving.hpp:739-742).
 * Only `=` and `<=` comparisons exist; `!=`/`>` are `0 = (y op z)` (common_solving.hpp:743-747). */
enum {
  ORC_ADD = 0,
  ORC_MUL = 1,
  ORC_TDIV = 2, /* truncating division (FlatZinc int_div) */
  ORC_TMOD = 3, /* truncating modulo   (FlatZinc int_mod) */
  ORC_MIN = 4,
  ORC_MAX = 5,
  ORC_EQ = 6,
  ORC_LEQ = 7
};

/* variable orders / value orders: barebones_dive_and_solve.hpp:193-221, 362-387 */
enum { ORC_INPUT_ORDER = 0, ORC_FIRST_FAIL = 1, ORC_ANTI_FIRST_FAIL = 2, ORC_SMALLEST = 3, ORC_LARGEST = 4 };
enum { ORC_VAL_MIN = 0, ORC_VAL_MAX = 1, ORC_VAL_SPLIT = 2, ORC_VAL_REVERSE_SPLIT = 3 };

typedef struct { int32_t lb, ub; } orc_itv;
typedef struct { int32_t op, x, y, z; } orc_prop;

typedef struct {
  int32_t subproblems_power;    /* d: 2^d EPS subproblems, solved in index order; 0 = plain DFS (cpu_solving.hpp) */
  int32_t has_eps_strategy;     /* strategy 0 is the EPS (dive-only) strategy, barebones:747-750 */
  int32_t use_fixed_bound;      /* 1: satisfaction search under the constant constraint obj <= fixed_bound (canonical-solution pass) */
  int32_t fixed_bound;
  uint64_t stop_after_n_nodes;  /* 0 = no limit (config.hpp -cutnodes) */
  uint64_t stop_after_n_solutions; /* satisfaction problems only; 0 = all */
  uint64_t timeout_ms;          /* 0 = none */
  int32_t leaf_requires_assignment; /* which leaf rule: 0 = barebones' (a node whose propagators are all entailed is a solution, barebones:988-993);
                                       1 = the `gpu` / `cpu` paths' (... and every variable is assigned: is_extractable<AtomicExtraction>,
                                       gpu_dive_and_solve.hpp:333-338, cpu_solving.hpp:33-40; otherwise the search keeps branching) */
  int32_t reserved;
} orc_config;

typedef struct {
  uint64_t nodes, fails, solutions, fixpoint_iterations, num_deductions;
  uint64_t eps_num_subproblems, eps_solved_subproblems, eps_skipped_subproblems;
  int32_t depth_max, exhaustive;
  int32_t best_bound;           /* objective of best solution (ORC_PINF if none) */
  int32_t best_subproblem;      /* index of the subproblem that produced it (-1 if none) */
  double solve_seconds;
} orc_stats;

/* Apply propagator p once on `store`.  Returns 1 if any bound moved.  Sets *failed if an
 * empty domain is seen or produced. */
int orc_deduce(const orc_prop* p, orc_itv* store, int* failed);

/* 1 iff propagator p is entailed (true for every valuation of the box `store`). */
int orc_ask(const orc_prop* p, const orc_itv* store);

/* One node: Gauss-Seidel fixpoint of all propagators (cpu_solving.hpp:26), then the
 * entailment test (cpu_solving.hpp:34 / barebones:971-982).
 * Returns 1 if the node failed.  *all_entailed is meaningful only when not failed. */
int orc_propagate(int32_t n_vars, orc_itv* store, int32_t n_props, const orc_prop* props,
                  uint64_t* iterations, uint64_t* deductions, int* all_entailed);

/* Full propagate-and-search (branch and bound when obj_var >= 0, always minimising).
 * Strategies are flattened: strategy s branches on strat_vars[strat_off[s] .. strat_off[s+1])
 * (an empty range = every variable of the store, barebones:242-243).
 * Returns 0 on success. */
int orc_solve(const orc_config* cfg, int32_t n_vars, const orc_itv* root_store,
              int32_t n_props, const orc_prop* props,
              int32_t n_strats, const int32_t* strat_var_order, const int32_t* strat_val_order,
              const int32_t* strat_off, const int32_t* strat_vars,
              int32_t obj_var, orc_itv* best_store_out, int32_t* has_solution_out, orc_stats* stats_out);

/* Test aid: copy every accepted solution leaf of the next orc_solve calls into buf (capacity solutions of n_vars
 * intervals); orc_solution_sink_count() = leaves seen (may exceed capacity).  NULL switches it off. */
void orc_set_solution_sink(orc_itv* buf, int64_t capacity);
int64_t orc_solution_sink_count(void);

/* Test aids: failed flag of every node of the next orc_solve calls (node i -> failed_flags[i]); a copy of the store the
 * search was working on when it returned.  NULL switches them off. */
void orc_set_node_trace(unsigned char* failed_flags, int64_t capacity);
void orc_set_last_store_sink(orc_itv* buf);

/* Test aid: replay one root-to-node path reported by the HIP engine (tb_session_debug_path) and return the store under its last node
 * (see oracle.c).  *mismatch_out = -1 when every recorded decision is the one this oracle takes on the replayed store. */
typedef struct { int32_t var, child; orc_itv children[2]; int32_t objective_ub; } orc_path_decision;
typedef struct { uint64_t subproblem; int32_t dive_levels_left, depth, decisions, last_objective_ub; } orc_path_header;
/* Test aid: the path the next orc_solve calls stand on when they return (NULL switches it off); feeding it to orc_replay_path
 * must give back the store of orc_set_last_store_sink -- the CPU-side check of the replay itself. */
void orc_set_path_sink(orc_path_header* hdr, orc_path_decision* decisions, int32_t capacity);
int orc_replay_path(const orc_config* cfg, int32_t n_vars, const orc_itv* root_store, int32_t n_props, const orc_prop* props,
                    int32_t n_strats, const int32_t* strat_var_order, const int32_t* strat_val_order, const int32_t* strat_off, const int32_t* strat_vars,
                    int32_t obj_var, uint64_t subproblem, int32_t dive_levels_left, int32_t n_decisions, const orc_path_decision* decisions,
                    int32_t last_objective_ub, orc_itv* store_out, int32_t* failed_out, int32_t* mismatch_out);

#ifdef __cplusplus
}
#endif
#endif
