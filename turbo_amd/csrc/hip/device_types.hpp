// Plain-old-data shared between the host shim and the kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/turbo_hip.h"

namespace tb {

// lala LightBranch (barebones_dive_and_solve.hpp:135,355-393): one entry of the decision stack.
struct Decision {
  int var;
  int cur;        // current child index (-1 before the first `next()`)
  int2 child[2];  // the two children intervals
  int rope[2];    // O(1) backtracking: depth to jump to after each child (barebones:388-393)
};

// Grid-level words shared by every workgroup (barebones GridData, :409-453): the only inter-workgroup state.
struct Ctrl {
  unsigned long long next_subproblem;  // work queue over the EPS index space (barebones:418)
  unsigned long long first_sol_idx;    // canonical pass: lowest subproblem index holding a solution
  int best_bound;                      // appx_best_bound (barebones:426), monotone min
  int foreign_bound;                   // incumbent imported from other GPUs (host writes it)
  int stop;                            // host stop request (UnifiedData::stop, barebones:64)
  int gpu_stop;                        // raised by a workgroup (solution limit reached / unbounded objective)
  int blocks_done;                     // number of workgroups that left the kernel
  int error;                           // device-side error code (decision stack overflow ...)
  unsigned long long solutions;        // satisfaction: global solution counter for -n
  unsigned long long sol_ticket;       // streaming: next sequence number of the solution ring
};

// Solution ring in pinned host memory (streaming, gpu_dive_and_solve.hpp:100-132 re-done without a print lock):
// a producer takes a ticket, waits until `ticket - consumed < slots`, copies its store into slot ticket % slots and
// publishes seq[slot] = ticket + 1; the host consumes in ticket order.
struct SolutionRing {
  unsigned long long* consumed;  // host -> device: number of solutions the host has taken
  unsigned long long* seq;       // [slots] device -> host
  int2* data;                    // [slots][n_vars]
  int slots;                     // 0 = streaming off
};

// Per-workgroup statistics (Statistics<>, statistics.hpp:134-154), reduced on the host.
struct BlockStats {
  unsigned long long nodes, fails, solutions, fixpoint_iterations, num_deductions;
  unsigned long long eps_solved, eps_skipped, store_writes;
  long long timers[TB_NUM_TIMERS];  // wall-clock ticks
  long long best_time;              // tick at which the best solution was found
  int depth_max, exhaustive, num_blocks_done, best_bound;
  long long best_sub;               // subproblem index that produced best_store (-1: none)
  int why, pad_why;                 // debugging: reasons that cleared `exhaustive` (bit mask)
};

struct DevProblem {
  int n_vars, n_props, n_strats, obj_var;
  const int4* props;
  const int2* root_store;
  const int* strat_var_order;
  const int* strat_val_order;
  const int* strat_off;
  const int* strat_vars;
  // event-driven fixpoint: variable -> 64-propagator slices adjacency (CSR), built by the shim
  const int4* adj_head; // [n_vars] {degree, first slice, second slice, offset of the remaining slices in adj}
  const int* adj;       // remaining slice ids of the variables read by more than two slices
  int n_slices;         // ceil(n_props / 64)
  int dirty_words;      // ceil(n_slices / 32)
  int vext;             // int2 elements of a store slab (even): intervals, Boolean words, one "not entailed" byte per slice
  int n_int;            // variables stored as int2 {lb,ub}; variables >= n_int are 2-bit Booleans (COMPACT layout), else n_int = n_vars
  int unent_off;        // byte offset of the per-slice bytes in a slab
  int chg_cap;          // capacity of one change list (entries)
  // configuration
  int fixpoint;            // 0 AC1, 1 WAC1, 2 event-driven WAC1
  int wac1_threshold;
  int subproblems_power;
  int has_eps_strategy;
  int use_fixed_bound, fixed_bound;
  int mem_kind;            // tb_mem_kind
  int debug;               // ablation knobs for profiling (tb_config.reserved[0]); 0 in production
  int snapshot_levels;     // >= 1
  int max_depth;           // capacity of the decision stack
  unsigned long long sub_lo, sub_hi;  // this device's slice of the EPS index space
  unsigned long long cut_nodes;       // 0 = none
  unsigned long long stop_after_n_solutions;
  long long deadline_ticks;           // watchdog (wall_clock64 units); 0 = none
  // per-workgroup buffers in HBM
  int2* g_store;     // [B][V]  working store (GLOBAL mode only)
  int2* g_snap;      // [B][L][V] snapshot stack
  int2* g_best;      // [B][V]
  Decision* g_dec;   // [B][max_depth]
  BlockStats* g_stats;
  Ctrl* ctrl;
  SolutionRing ring;
};

}  // namespace tb
