/*
 * turbo_hip.h -- C-ABI of the MI355X-native dive-and-solve engine (libturbo_hip.so).
 *
 * This is the drop-in boundary for ptal/turbo's GPU solving path.  The reference has no FFI
 * today (header-only templates in one translation unit, src/turbo.cpp:4-18); the cut is made
 * after `CP<Itv>::preprocess()` and before the kernel launch, i.e. it replaces
 *   include/gpu_dive_and_solve.hpp:671-676  (configure_and_run: memory config + launch + wait)
 *   include/barebones_dive_and_solve.hpp:479-497 (configure_gpu_barebones, launch, wait, reduce)
 *   include/memory_gpu.hpp:27-84,174-196    (MemoryConfig, wait_solving_ends)
 * What crosses is the preprocessed ternary constraint network (TCN) as plain arrays; what
 * comes back is the best store, the statistics and the exhaustive flag.
 * INTEGRATION.md shows the reference-side binding.
 *
 * Conventions: POD only; the caller owns every host buffer; the callee owns all device memory;
 * every entry point returns 0 on success or a negative tb_error, never throws, never exits,
 * writes nothing to stdout.  tb_last_error() gives a message for the calling thread.
 */
#ifndef TURBO_HIP_H
#define TURBO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TB_NINF INT32_MIN /* -inf sentinel of Interval<ZLB<int>> (common_solving.hpp:45-54, TURBO_ITV_BITS=32) */
#define TB_PINF INT32_MAX /* +inf sentinel */

/* `x = y op z` operators of the PIR bytecode (common_solving.hpp:739-771). */
enum tb_op { TB_ADD = 0, TB_MUL = 1, TB_TDIV = 2, TB_TMOD = 3, TB_MIN = 4, TB_MAX = 5, TB_EQ = 6, TB_LEQ = 7, TB_NUM_OPS = 8 };

/* lala VariableOrder / ValueOrder as used by barebones_dive_and_solve.hpp:193-221,362-387
 * (RANDOM is INPUT_ORDER over an already shuffled variable list, common_solving.hpp:632-633). */
enum tb_var_order { TB_INPUT_ORDER = 0, TB_FIRST_FAIL = 1, TB_ANTI_FIRST_FAIL = 2, TB_SMALLEST = 3, TB_LARGEST = 4 };
enum tb_val_order { TB_VAL_MIN = 0, TB_VAL_MAX = 1, TB_VAL_SPLIT = 2, TB_VAL_REVERSE_SPLIT = 3 };

/* memory_gpu.hpp:18-22.  Where a workgroup's working store (and the propagator records) live: GLOBAL = a slab in HBM per workgroup; STORE_SHARED =
 * the store in LDS, records streamed from L2; TCN_SHARED = store and records in LDS (planned for the plain sweeps on small networks only: the
 * event-driven fixpoint and the compact layouts keep their records in L2, measured faster -- more workgroups per CU). */
enum tb_mem_kind { TB_MEM_GLOBAL = 0, TB_MEM_STORE_SHARED = 1, TB_MEM_TCN_SHARED = 2 };

enum tb_error {
  TB_OK = 0,
  TB_ERR_INVALID = -1,   /* bad argument */
  TB_ERR_NO_DEVICE = -2, /* no gfx950 device / HIP runtime failure at init */
  TB_ERR_HIP = -3,       /* HIP runtime error while running (see tb_last_error) */
  TB_ERR_OOM = -4,
  TB_ERR_STATE = -5,     /* session used out of order */
  TB_ERR_DEPTH = -6      /* the search went deeper than the decision stack (tb_config.decision_stack_depth) */
};

typedef struct { int32_t lb, ub; } tb_itv;          /* VStore element: 8 B */
typedef struct { int32_t op, x, y, z; } tb_prop;    /* PIR bytecode_type: 16 B, 16-B aligned */

/* The scalars of Configuration<> (include/config.hpp:35-59) that reach the GPU path. */
/* tb_config.fixpoint = 3 (`-fp auto`): the event-driven fixpoint from this many propagators on.  Measured r04 on the reference's regression instances (2 M nodes, full
 * grid, scripts/r04_auto_threshold.py): sweeps win at 127 and 161 propagators (4.7e7 / 5.4e7 against 3.0e7 nodes/s), the event fixpoint from 342 on (pat7 2.1x, triangular9 1.3x,
 * accap_a3 961 propagators 1.5x, pennies5 1.45x; bug4, 1009 propagators, is the exception: 0.75x).  r02 had set 2048. */
#define TB_AUTO_EVENT_MIN_PROPS 320

typedef struct {
  uint64_t timeout_ms;              /* -t; 0 = none */
  uint64_t or_nodes;                /* -or: number of workgroups; 0 = auto (barebones:538-546) */
  uint64_t subproblems_factor;      /* -subfactor (default 300), barebones:550-555 */
  uint64_t stop_after_n_nodes;      /* -cutnodes, per workgroup (barebones:1024); 0 = no limit */
  uint64_t stop_after_n_solutions;  /* -n, satisfaction problems only; 0 = all */
  uint64_t wac1_threshold;          /* -wac1_threshold (barebones:939) */
  uint64_t stop_after_n_nodes_total;/* node budget of the whole search, all workgroups of all linked GPUs together (0 = none): fixed total
                                       work for strong-scaling measurements.  Every device counts its own nodes (batches of 32 per workgroup); linked
                                       devices are summed in rank 0's cell by their pollers, once per poll period */
  int32_t subproblems_power;        /* -sub; -1 = auto.  At most 2^28 - 65536 subproblems per GPU (its share is served through a 28-bit queue
                                       word; the reference's 64-bit counter has no limit, its default is 300 per workgroup); at most 2^40 in all */
  int32_t fixpoint;                 /* 0 = AC1, 1 = WAC1 (config.hpp:22-25), 2 = event-driven WAC1 (this engine), 3 = automatic: event-driven from
                                       TB_AUTO_EVENT_MIN_PROPS propagators on, WAC1 below (a sweep over a few slices is cheaper than any
                                       bookkeeping); all of them compute the same fixpoint at every node, hence the same search tree */
  int32_t only_global_memory;       /* -globalmem */
  int32_t verbose;
  int32_t has_eps_strategy;         /* strategy 0 is the EPS strategy (barebones:434,747-750) */
  int32_t threads_per_block;        /* 0 = auto (reference: CMakeLists.txt:93 fixes 256) */
  int32_t device;                   /* HIP device ordinal */
  int32_t rank, world_size;         /* EPS index space sharding across GPUs (block-cyclic, see tb_eps_global_index); 0/1 = single GPU */
  int32_t use_fixed_bound;          /* 1: first-solution search under obj <= fixed_bound, lowest subproblem wins (canonical pass) */
  int32_t fixed_bound;
  int32_t deterministic;            /* tb_solve only: after B&B, run the canonical pass so the returned solution is the
                                       DFS-first optimal one (bit-identical to the sequential oracle) */
  int32_t snapshot_levels;          /* per-workgroup snapshot stack depth in HBM; 0 = auto, 1 = reference behaviour (recompute from subproblem root) */
  int32_t stream_solutions;         /* sessions only: hand solutions to the host while the kernel runs (tb_session_next_solution) */
  int32_t entailed_prop_removal;    /* AC1 / WAC1: skip, below the node that proved it, every 64-propagator slice whose propagators are
                                       all entailed (the reference's build option TURBO_NO_ENTAILED_PROP_REMOVAL=OFF, CMakeLists.txt:28;
                                       default 0 like the reference).  The event-driven fixpoint always does it. */
  int32_t eps_chunk_log2;           /* multi-GPU: the 2^d subproblems are dealt to the GPUs in chunks of 2^k consecutive indices, k = this
                                       value (0 = one by one, the default: best balance; clamped to subproblems_power) */
  int32_t decision_stack_depth;     /* size of one segment of a workgroup's decision stack, rounded up to a power of two; 0 = auto (16384).
                                       The reference grows a block's stack on demand (barebones:401-403); here a workgroup whose search goes
                                       deeper takes further segments (up to 16 in all) from a per-session pool in HBM, in the kernel.  If that
                                       is not enough the search ends with TB_ERR_DEPTH; tb_solve and the CLI then run it again with 8x larger segments */
  int32_t poll_period_us;           /* wall-clock period at which the kernel looks at the host mailbox and at the peers' words; 0 = 100 us */
  int32_t reserved[3];              /* 0 in production.  Tuning / test knobs read by the engine (the device-side ones -- ablations, timers,
                                       0x20000, 0x400000 -- only in a -DTB_TUNING build: they sit in the hot loops):
                                       [0] bit mask -- 0x1/0x2/0x4/0x8 and bits 8-15: sweep ablations of scripts/ablate.py (results are
                                           not fixpoints); 0x10000 in-kernel phase timers; 0x20000 keep running entailed slices;
                                           0x40000 event mode accepts < 4 workgroups per CU in LDS; 0x80000 never / 0x100000 always
                                           use the compact (2-bit Boolean) store layout of the event kernels (0x100000 with the sweeps and
                                           without entailed_prop_removal: the sweeps on that layout, an opt-in); 0x10000000 always / 0x20000000
                                           never pack integer variables as 16-bit bounds on top of it (COMPACT16); both bits: always / 0x80000000 never
                                           keep the integers at most 255 wide as two bytes relative to their root lower bound (COMPACT8); 0x40000000 keep the non-Boolean singletons
                                           of the root in the slab (by default the compact layouts carry their values in the records); 0x200000 keep the caller's
                                           propagator order instead of sorting the records by class; 0x400000 count slice
                                           runs instead of propagator evaluations; 0x800000 test aid: keep the store every workgroup stopped
                                           on (tb_session_debug_last_store); 0x1000000 no work stealing between linked GPUs (A/B runs, tests);
                                           0x2000000 do not propagate the root at session creation (every subproblem re-derives its fixpoint, as in r01);
                                       [1] capacity of the event change list; [2] cap on workgroups per CU */
  int32_t leaf_requires_assignment; /* which leaf rule the search follows.  0 (`-arch barebones`): a node is a solution as soon as every propagator is
                                       entailed (barebones_dive_and_solve.hpp:988-993) -- the solution may be a box with unassigned variables.
                                       1 (`-arch gpu`, and the reference's `-arch cpu`): ... and every variable of the store is assigned
                                       (`is_extractable<AtomicExtraction>`, gpu_dive_and_solve.hpp:333-338, cpu_solving.hpp:33-40); an all-entailed
                                       node with an open variable is an inner node, the search keeps branching below it.  (Takes the struct's former tail
                                       padding: sizeof(tb_config) is unchanged, a caller that zero-initialises gets the barebones rule.) */
} tb_config;

/* Statistics<> (include/statistics.hpp:134-154) + TimingStatistics (statistics.hpp:13-29). */
typedef struct {
  uint64_t nodes, fails, solutions, fixpoint_iterations, num_deductions;
  uint64_t eps_num_subproblems, eps_solved_subproblems, eps_skipped_subproblems, num_blocks_done;
  int64_t timers_ns[11];            /* indexed by tb_timer, summed over workgroups like statistics.hpp:72-77 */
  int64_t cumulative_time_block_ns;
  int64_t kernel_ns;                /* wall time of the persistent kernel (HIP events) */
  uint64_t store_writes;            /* number of narrowed bounds written by deduce (roofline write term) */
  int32_t depth_max, num_blocks, threads_per_block, exhaustive;
  int32_t mem_kind, shared_bytes, subproblems_power, best_bound;
  int32_t best_subproblem, interrupted;
  int32_t reserved[2];
  /* multi-GPU balance (this rank): */
  uint64_t eps_local_subproblems;   /* size of this rank's block-cyclic share of the 2^d subproblems */
  uint64_t eps_stolen_subproblems;  /* subproblems this GPU took over from other GPUs' queues (xGMI work stealing) */
  int64_t wait_time_ns;             /* summed over workgroups: time without a subproblem (looking / waiting for work on other GPUs) */
  int64_t min_block_ns, max_block_ns; /* first and last workgroup to leave the kernel (the reference's first_block_idle_time is the min) */
  uint64_t active_lane_evaluations; /* num_deductions counts wave iterations x wave width, as the reference does (barebones:958-960); this is the
                                       same count without the idle lanes of partly filled slices (class padding, the network's last slice) */
  int64_t prof_ns[4];               /* tuning build with the in-kernel phase timers on (reserved[0] & 0x10000), summed over workgroups, 0 otherwise -- indexed by
                                       tb_prof: phases of THIS engine, kept apart from the reference's timers (r04 lent them four of those keys) */
} tb_stats;
enum tb_prof { TB_PROF_SEEDING = 0, TB_PROF_ROUNDS = 1, TB_PROF_SNAPSHOT_PUSH = 2, TB_PROF_VARIABLE_SELECTION = 3, TB_NUM_PROF = 4 };

enum tb_timer { /* enum class Timer, statistics.hpp:13-29 (same order, 11 timers) */
  TB_T_OVERALL = 0, TB_T_PREPROCESSING = 1, TB_T_SEARCH = 2, TB_T_FIXPOINT = 3, TB_T_TRANSFER_CPU2GPU = 4,
  TB_T_TRANSFER_GPU2CPU = 5, TB_T_SELECT_FP_FUNCTIONS = 6, TB_T_WAIT_CPU = 7, TB_T_DIVE = 8,
  TB_T_LATEST_BEST_OBJ_FOUND = 9, TB_T_FIRST_BLOCK_IDLE = 10, TB_NUM_TIMERS = 11
};

typedef struct {
  char name[256];
  int32_t compute_units, lds_bytes_per_cu, wavefront_size, clock_khz;
  int64_t total_global_mem;
  int32_t is_gfx950, xcc_count;
} tb_device_info;

/* Library identification and per-thread error text. */
const char* tb_version(void);
const char* tb_last_error(void);
int tb_device_count(void);
int tb_get_device_info(int device, tb_device_info* out);

/* How the 2^d EPS subproblems (barebones:413-418) are dealt to the GPUs of a node -- host arithmetic only, usable
 * without a device.  Block-cyclic: chunk c (2^k consecutive indices) belongs to rank c % world_size; a rank numbers its own
 * subproblems j = 0, 1, ... in index order.  Static and balanced; on top of it an idle GPU takes over the upper half of
 * the fullest peer queue (work stealing over xGMI), so every GPU pulls from what is in effect one node-wide queue, like
 * every block of the reference pulls from one grid-wide counter (barebones:877-884).
 * tb_eps_local_count: number of subproblems of `rank`; tb_eps_global_index: global index of its j-th one. */
int tb_eps_local_count(int32_t subproblems_power, int32_t chunk_log2, int32_t rank, int32_t world_size, uint64_t* count_out);
int tb_eps_global_index(int32_t subproblems_power, int32_t chunk_log2, int32_t rank, int32_t world_size, uint64_t j, uint64_t* index_out);

/*
 * One search node for a batch of independent stores: block-parallel fixpoint of all propagators
 * (AC1 / WAC1) followed by the entailment test.  Replaces the device function `propagate`
 * (gpu_dive_and_solve.hpp:287-368, barebones_dive_and_solve.hpp:903-1031) minus bookkeeping.
 * One store per workgroup.  stores_inout holds n_stores * n_vars intervals and is overwritten
 * with the fixpoint.  Any *_out pointer may be NULL.
 */
int tb_propagate(const tb_config* cfg, int32_t n_vars, int32_t n_props, const tb_prop* props,
                 int32_t n_stores, tb_itv* stores_inout, int32_t* failed_out, int32_t* all_entailed_out,
                 uint64_t* iterations_out, uint64_t* deductions_out, int64_t* kernel_ns_out);

/*
 * Full dive-and-solve.  Replaces gpu_dive_and_solve (gpu_dive_and_solve.hpp:680-700 after preprocess)
 * and barebones_dive_and_solve (barebones_dive_and_solve.hpp:479-497).  Always minimises obj_var
 * (obj_var = -1: satisfaction), like barebones (common_solving.hpp:277-279).
 * Strategies are flattened: strategy s branches on strat_vars[strat_off[s] .. strat_off[s+1]); an empty
 * range means every variable of the store (barebones:242-243).
 * Blocks until the search ends, the timeout expires, or *host_stop_flag becomes non-zero
 * (set by the caller's SIGINT logic, common_solving.hpp:56-104).
 */
int tb_solve(const tb_config* cfg, int32_t n_vars, const tb_itv* root_store,
             int32_t n_props, const tb_prop* props,
             int32_t n_strats, const int32_t* strat_var_order, const int32_t* strat_val_order,
             const int32_t* strat_off, const int32_t* strat_vars,
             int32_t obj_var, volatile int32_t* host_stop_flag,
             tb_itv* best_store_out, int32_t* has_solution_out, tb_stats* stats_out);

/*
 * Asynchronous form of tb_solve (what a multi-GPU host uses: one session per device / process).
 * create: uploads the TCN, sizes LDS/HBM buffers (replaces configure_memory / configure_gpu_barebones,
 * gpu_dive_and_solve.hpp:534-584, barebones:527-606).  start: launches the persistent kernel.
 * poll: non-blocking; returns the device's current incumbent bound and whether the kernel finished.
 * push_bound: imports a foreign incumbent (the only payload exchanged between GPUs).
 * stop: asks every workgroup to stop at its next node.  finish: waits, reduces the workgroups
 * (replaces reduce_blocks, barebones:1033-1067) and copies the results out.
 */
typedef struct tb_session tb_session;
int tb_session_create(const tb_config* cfg, int32_t n_vars, const tb_itv* root_store,
                      int32_t n_props, const tb_prop* props,
                      int32_t n_strats, const int32_t* strat_var_order, const int32_t* strat_val_order,
                      const int32_t* strat_off, const int32_t* strat_vars,
                      int32_t obj_var, tb_session** out);
int tb_session_start(tb_session* s);
/* What create decided (replaces the printouts of configure_gpu_barebones, barebones:527-606): grid, memory kind, 2^d.
 * Ranks of one search compare subproblems_power before they link: the shares only tile the index space if d agrees. */
typedef struct {
  int32_t num_blocks, threads_per_block, mem_kind, shared_bytes, subproblems_power, eps_chunk_log2, snapshot_levels, decision_stack_depth;
  uint64_t eps_local_subproblems;
  int32_t kernel_event, kernel_opt; /* which kernel start() launches: event-driven fixpoint or sweeps; its option flag (event: 0 plain store,
                                     * 1 COMPACT, 2 COMPACT16, 3 plain store in global memory with its most-read intervals in LDS, 4 COMPACT8; sweeps: 0 plain,
                                     * 1 entailed-slice removal, 2 COMPACT, 4 COMPACT16, 6 the hot tier, 10 workgroup teams: the workgroups of an XCD share stores in
                                     * global memory, one subproblem per team) */
} tb_plan;
int tb_session_plan(tb_session* s, tb_plan* plan_out);
/*
 * Multi-GPU wiring, between create and start.  Every session owns one 192-byte cell in fine-grained device memory: its
 * work-queue word and the incumbent bound imported from the other GPUs -- the only state another GPU touches
 * (barebones GridData::next_subproblem / appx_best_bound, :418,426).  Once the sessions of a node are linked, their
 * kernels exchange the bound (atomicMin of one int32 into every peer's cell) and rebalance work (CAS on a peer's queue
 * word) directly over xGMI; the host is not involved.  A session that is not linked to some rank still works: it then
 * relies on tb_session_poll / tb_session_push_bound (host relay) for the bound and on its static share for the work.
 *   same process:     tb_session_link_peer(a, b) in both directions (enables peer access between the two devices);
 *   other process:    tb_session_export_peer -> 64-byte handle (hipIpcMemHandle_t), sent by whatever means the processes
 *                     share (bench.py: torch.distributed all_gather over RCCL), tb_session_import_peer on the other side.
 * export / link fail with TB_ERR_HIP when a cell could not be placed in fine-grained memory (cross-GPU atomics on it would not
 * be coherent while kernels run): the caller then uses the host relay.
 * tb_session_arm resets the device-side state of a search (queue, bounds, counters); start does it itself when the
 * caller has not.  With linked sessions every rank arms, then all ranks synchronise, then every rank starts -- and all
 * ranks synchronise again after finish before the next arm: a peer must not touch a cell that is being reset.
 */
typedef struct { unsigned char bytes[64]; } tb_peer_handle;
int tb_session_export_peer(tb_session* s, tb_peer_handle* handle_out);
int tb_session_import_peer(tb_session* s, int32_t peer_rank, const tb_peer_handle* handle);
int tb_session_link_peer(tb_session* s, tb_session* peer);
/* Drop every cell linked / imported so far (a group that could not be linked completely falls back to the host relay as a whole). */
int tb_session_unlink_peers(tb_session* s);
int tb_session_arm(tb_session* s);
/* Remaining work of this GPU as of the kernel's last poll: subproblems not yet handed to a workgroup (its own share and
 * what it took from others), and how many it has taken from / lost to other GPUs so far. */
int tb_session_progress(tb_session* s, uint64_t* remaining_out, uint64_t* stolen_in_out, uint64_t* stolen_out_out);
/* Test aid (tb_config.reserved[0] & 0x800000): the store workgroup `workgroup` was working on when it left the kernel. */
int tb_session_debug_last_store(tb_session* s, int32_t workgroup, tb_itv* store_out);
/* Test aid (same knob): where workgroup `workgroup` stood when it left the kernel -- its subproblem, how much of the dive was left, the decisions on
 * its stack (variable in the caller's numbering, both children, the child taken, the objective's upper bound in force when the decision was taken) and
 * the bound in force at its last node.  With it a checker can replay the path from the root: root -> dive along the bits of `subproblem` -> the
 * decisions, and must arrive at the store of tb_session_debug_last_store (barebones_dive_and_solve.hpp:675-714,752-864 is what is being replayed).
 * At most `capacity` decisions are written (and only those of the first decision-stack segment). */
typedef struct {
  uint64_t subproblem;
  int32_t dive_levels_left;   /* > 0: the workgroup stopped while diving; the decisions below are then stale */
  int32_t depth;              /* decisions on the stack */
  int32_t decisions;          /* entries written to decisions_out */
  int32_t last_objective_ub;  /* INT32_MAX: no bound imposed yet in this subproblem */
  int32_t last_node_failed;
  int32_t had_work;           /* 0: it left because no subproblem was left */
  int32_t nodes;
  int32_t reserved;
} tb_debug_path;
typedef struct {
  int32_t var, child;         /* child: 0 / 1, the one being explored */
  tb_itv children[2];
  int32_t objective_ub;       /* INT32_MAX: none */
} tb_debug_decision;
int tb_session_debug_path(tb_session* s, int32_t workgroup, tb_debug_path* path_out, int32_t capacity, tb_debug_decision* decisions_out);
int tb_session_poll(tb_session* s, int32_t* local_best_out, int32_t* done_out);
int tb_session_push_bound(tb_session* s, int32_t bound);
int tb_session_stop(tb_session* s);
/*
 * Solution streaming (cfg.stream_solutions = 1), the producer/consumer protocol of the reference's `gpu` path
 * (gpu_dive_and_solve.hpp:100-132,334-345: `-i`, `-a`, satisfaction `-n k`) over a ring of pinned host buffers:
 * every solution of a satisfaction problem (at most stop_after_n_solutions of them), and every solution that improves
 * the device-wide incumbent of an optimisation problem, is handed over while the kernel runs.  A workgroup waits for
 * a free slot, so the caller drains the ring while polling, and once more after `done`.
 * *has_out = 1: store_out (n_vars intervals) holds the next solution, *objective_out its objective (lb of obj_var).
 * Solutions of concurrent workgroups may arrive out of objective order; a printer keeps the improving ones.
 */
int tb_session_next_solution(tb_session* s, tb_itv* store_out, int32_t* objective_out, int32_t* has_out);
int tb_session_finish(tb_session* s, tb_itv* best_store_out, int32_t* has_solution_out, tb_stats* stats_out);
void tb_session_destroy(tb_session* s);

#ifdef __cplusplus
}
#endif
#endif
