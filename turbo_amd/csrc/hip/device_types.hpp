// Plain-old-data shared between the host shim and the kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/turbo_hip.h"

#ifdef TB_TUNING
#define TB_DBG_WORDS 32
#else
#define TB_DBG_WORDS 4
#endif

namespace tb {

// lala LightBranch (barebones_dive_and_solve.hpp:135,355-393): one entry of the decision stack.
struct Decision {
  int var;
  int cur;        // current child index (-1 before the first `next()`)
  int2 child[2];  // the two children intervals
  int rope[2];    // O(1) backtracking: depth to jump to after each child (barebones:388-393)
};

// Grid-level words shared by every workgroup of ONE device (barebones GridData, :409-453), plain device memory touched
// with agent-scope atomics only.  The first 16 bytes are what thread 0 of every workgroup reads at every node: one load.
struct alignas(16) Ctrl {
  int best_bound;                      // appx_best_bound (barebones:426), monotone min
  int foreign_bound;                   // incumbent imported from other GPUs (peers' cells / the host mailbox)
  int stop;                            // bit 0: host stop request (UnifiedData::stop, barebones:64); bit 1: raised by a workgroup
                                       // (solution limit reached / unbounded objective), on this or on a peer device
  unsigned next_poll;                  // low 32 bits of the wall-clock tick at which the mailbox / peer cell is polled next
  unsigned long long first_sol_idx;    // canonical pass: lowest subproblem index holding a solution
  unsigned long long solutions;        // satisfaction: global solution counter for -n
  unsigned long long sol_ticket;       // streaming: next sequence number of the solution ring
  int blocks_done;                     // number of workgroups that left the kernel
  int error;                           // device-side error code (1: decision stack overflow)
  // node budget of the whole search (tb_config.stop_after_n_nodes_total): every workgroup of THIS device adds its nodes here in
  // batches (agent scope); with linked GPUs the device's poller folds what is new into rank 0's cell once per poll period --
  // one system-scope atomic per device and period instead of one per workgroup and batch crossing xGMI
  unsigned long long nodes_local;
  unsigned long long nodes_folded;     // the part of nodes_local already added to PeerCell::nodes_total of rank 0
  int dec_pool_next;                   // next free segment of DevProblem::dec_pool (decision stacks grow on demand, barebones:401-403)
  int steal_lock;                      // one workgroup of this device at a time looks for work on the other GPUs
#ifdef TB_TRAP_SEED
  int trap[48];                        // debugging aid (r06): what a workgroup saw when it met a decision or a change-list entry that cannot be (kernels.hpp: trap_report)
#endif
};
constexpr int STOP_HOST = 1, STOP_GPU = 2;

// One contiguous run of LOCAL subproblem indices of rank `owner` (see eps_global_index below).
struct QueueDesc {
  unsigned long long j_base;
  int owner, pad;
};

// The only state another GPU touches (barebones GridData::next_subproblem + appx_best_bound, :418,426, made
// multi-device): one cell per session in FINE-GRAINED device memory, mapped by every peer of the node (same process:
// hipDeviceEnablePeerAccess; other processes: hipIpcOpenMemHandle), accessed with system-scope atomics over xGMI.
//   queue  = gen:8 | next:28 | hi:28 -- the work queue over desc[gen & 7]: a fetch is one atomicAdd of 1 << 28, a thief
//            lowers `hi` with a CAS, the owner installs a new range (stolen from a peer) by bumping gen.
//   bound  = incumbent found by the peers: they atomicMin into it; the local poller folds it into Ctrl::foreign_bound.
struct alignas(64) PeerCell {
  unsigned long long queue;
  int bound;
  int stealing;   // 1 while a range taken from a peer's queue is in transit to this device's `queue` (not while it merely looks around)
  int stop;       // a peer reached the solution limit / proved the objective unbounded
  int waiting;    // workgroups of this device currently waiting for work (diagnostic)
  unsigned long long stolen_in, stolen_out;  // subproblems moved into / out of this device (diagnostic)
  unsigned long long nodes_total;            // rank 0's cell only: nodes explored by all GPUs (tb_config.stop_after_n_nodes_total)
  unsigned long long pad[2];
  QueueDesc desc[8];
};
static_assert(sizeof(PeerCell) == 192, "PeerCell is mapped by other processes: its layout is part of the protocol");
// 28-bit `next` / `hi`: a GPU serves at most 2^28 - 65536 subproblems of its share through one queue word (plan_launch refuses
// more; the margin covers the workgroups that may overshoot `next` at the same moment).  The reference's 64-bit counter has no
// such limit, but 2^28 subproblems per GPU is 500 times its own default of 300 per workgroup.
constexpr int Q_BITS = 28;
constexpr unsigned long long Q_MASK = (1ull << Q_BITS) - 1ull;
__host__ __device__ inline unsigned long long q_pack(unsigned gen, unsigned long long next, unsigned long long hi) {
  return ((unsigned long long)(gen & 0xffu) << (2 * Q_BITS)) | ((next & Q_MASK) << Q_BITS) | (hi & Q_MASK);
}
__host__ __device__ inline unsigned q_gen(unsigned long long w) { return (unsigned)(w >> (2 * Q_BITS)) & 0xffu; }
__host__ __device__ inline unsigned long long q_next(unsigned long long w) { return (w >> Q_BITS) & Q_MASK; }
__host__ __device__ inline unsigned long long q_hi(unsigned long long w) { return w & Q_MASK; }

// EPS index space across GPUs: block-cyclic.  The 2^d subproblems are cut into chunks of 2^k consecutive indices;
// rank g of G owns the chunks c with c % G == g and numbers its own subproblems j = 0, 1, ... in index order.
// (G = 1: j is the global index.)  Static, needs no communication, and statistically balanced because neighbouring
// subproblems -- which share most of their dive path and tend to be equally hard -- go to different GPUs.
__host__ __device__ inline unsigned long long eps_global_index(unsigned long long j, int k, int g, int G) {
  return G == 1 ? j : ((((j >> k) * (unsigned long long)G + (unsigned long long)g) << k) | (j & ((1ull << k) - 1ull)));
}
// smallest local index of rank g whose global index is >= t
__host__ __device__ inline unsigned long long eps_local_lower_bound(unsigned long long t, int k, int g, int G) {
  if (G == 1) return t;
  const unsigned long long c = t >> k, m = c % (unsigned long long)G;
  const unsigned long long d = ((unsigned long long)g + (unsigned long long)G - m) % (unsigned long long)G;
  if (d == 0) return (((c / (unsigned long long)G)) << k) | (t & ((1ull << k) - 1ull));
  return ((c + d) / (unsigned long long)G) << k;
}
// number of subproblems of rank g (d = subproblems_power)
__host__ __device__ inline unsigned long long eps_local_count(int d, int k, int g, int G) {
  return eps_local_lower_bound(1ull << d, k, g, G);
}

// Workgroup teams (store layout 5, r05): the workgroups resident on ONE XCD search one subproblem together on ONE store in global memory, which then lives in that
// XCD's 4 MB of L2 instead of 32 stores of 800 KB thrashing it (synthetic 100k x 500k: three of four gathers missed L2).  One TeamCtl per XCD, plain device memory
// touched with agent-scope atomics only.
struct alignas(64) TeamCtl {
  unsigned members;     // workgroups registered (team formation at kernel start)
  unsigned arrive;      // barrier: arrivals of the current generation
  unsigned flags;       // barrier: OR of the members' contributions
  unsigned gen;         // barrier generation
  unsigned result[2];   // merged flags of generation g in result[g & 1]
  unsigned pad[2];      // pad[0]: blockIdx of the team's leader (whose slabs of g_store / g_snap the team works on), written at team formation
  unsigned long long bcast[4];  // leader -> members (subproblem, bound, the final verdict), read between two barriers
};
struct alignas(64) TeamGrid {
  unsigned registered;      // workgroups of the grid that have joined a team
  unsigned xcd_members[8];  // workgroups registered per XCD (HW_REG_XCC_ID)
  unsigned pad[7];
  TeamCtl team[64];         // XCD x * split + k: up to eight teams per XCD (DevProblem::team_split)
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
  unsigned long long stolen;        // subproblems this workgroup took from other GPUs' queues
  unsigned long long active_evals;  // num_deductions without the idle lanes of partly filled slices
  long long timers[TB_NUM_TIMERS];  // wall-clock ticks
  long long best_time;              // tick at which the best solution was found
  long long wait_ticks;             // time spent without a subproblem (waiting for / looking for work on other GPUs)
  int depth_max, exhaustive, num_blocks_done, best_bound;
  long long best_sub;               // subproblem index that produced best_store (-1: none)
  int why, pad_why;                 // debugging: reasons that cleared `exhaustive` (bit mask)
#ifdef TB_TUNING
  unsigned reg[72];                 // tuning build: how often a wave passed each region marker of the search kernel (kernels.hpp: TB_REGION)
  long long prof[TB_NUM_PROF];      // tuning build, knob 0x10000: thread 0's wall clock in the engine's own phases (tb_stats.prof_ns)
#endif
  int dbg[TB_DBG_WORDS];            // tuning build: census of the event fixpoint / first violation found by its self-check (production keeps 4 words: the
                                    // block sits in every workgroup's LDS, where 1280-byte granules decide how many workgroups a CU holds)
};

// Test aid (tb_config.reserved[0] & 0x800000, tb_session_debug_path): where a workgroup stood when it left the kernel.
struct PathHeader {
  unsigned long long sub_idx;  // global index of its subproblem
  int remaining;               // levels of the dive still to take (> 0: it stopped while diving)
  int depth;                   // decisions on its stack
  int last_obj_ub;             // upper bound last imposed on the objective in this subproblem (INT32_MAX: none)
  int failed;                  // the last node failed
  int has_work;                // 0: it left because no subproblem was left
  int nodes;
};

struct DevProblem {
  int n_vars, n_props, n_strats, obj_var;
  const int4* props;
  const int2* root_store;
  const int* strat_var_order;
  const int* strat_val_order;
  const int* strat_off;
  const int* strat_vars;
  // event-driven fixpoint: variable -> 64-propagator slices adjacency, built by the shim (engine.hip: pack_var_adj, pack_succ)
  const int4* var_adj;  // [2 * n_vars] per variable 16 halfwords: number of reader slices, the first 11, their interests (2 bits each:
                        // 1 = woken when the lower bound rises, 2 = when the upper bound falls), offset of the others in adj_rest
  const int* adj_rest;  // reader slices beyond the eleventh: slice | interest << 30
  const int4* succ;     // [padded n_props] per record {x, y, z: up to two OTHER slices reading the operand, 16 bits each, 0xffff = none;
                        //  w: bit k = operand k has more of them (walk var_adj), bits 4..15 = interests of the packed ones}
  const int2* slice_info;  // [n_slices] event kernels, read with scalar loads: {word0 of the slice's records (class set, operand kinds, flags),
                           //  lanes holding a propagator | 0x100: lean implication records (engine.hip: pack_succ)}
  const int* slice_real;  // [n_slices] event kernels: lanes of the slice that hold a propagator (the engine pads every class to whole slices)
  const int2* cond2;      // [padded n_props] event kernels, COMPACT layouts (r05, engine.hip: pack_cond2): for a channelling record c = (val = v) whose truth variable is only read, as far as
                          // "c became false" goes, by the implications b_i <= c of ONE index variable: {reference of that index, lowest | highest << 16 position i}; x < 0: none
#ifdef TB_TUNING
  unsigned* slice_census;  // tuning build, knob 0x400000 + verbose: [2 * n_slices] runs of every slice, and those that narrowed nothing (whole grid)
#endif
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
  int max_depth;           // capacity of one segment of the decision stack (a power of two); a workgroup starts with one segment
  int max_depth_log2;
  int dec_pool_segments;   // segments in dec_pool, handed out on demand (Ctrl::dec_pool_next)
  int rank, world;         // this device among the GPUs of the search
  int chunk_log2;          // k of the block-cyclic partition
  int poll_ticks;          // wall-clock ticks between two polls of the mailbox / peer cell
  int root_fixpoint;       // 1: root_store is already the fixpoint of the root node (propagated once at session creation): the first
                           // node of every subproblem then has nothing to propagate instead of re-deriving it from the caller's store
  int steal;               // 1: a device whose queue is empty takes work from its peers (0: tb_config.reserved[0] & 0x1000000, A/B runs and tests)
  int leaf_assign;         // tb_config.leaf_requires_assignment: an all-entailed node is a solution only when every variable of the slab is assigned (gpu_dive_and_solve.hpp:337)
  unsigned long long cut_nodes;       // 0 = none
  unsigned long long cut_nodes_total; // 0 = none: budget of all workgroups of all GPUs together
  unsigned long long stop_after_n_solutions;
  long long deadline_ticks;           // watchdog (wall_clock64 units); 0 = none
  // per-workgroup buffers in HBM
  int2* g_store;     // [B][V]  working store (GLOBAL mode only)
  int2* g_snap;      // [B][L][V] snapshot stack
  int2* g_best;      // [B][V]
  unsigned* g_dirty; // [B][2][dirty_words] workgroup teams with the event fixpoint (layout 5, r06): the dirty bitmaps of a team (its leader's slab); nullptr otherwise
  int2* g_last;      // [B][V] test aid (tb_config.reserved[0] & 0x800000): the store of each workgroup when it left the kernel
  int* g_path_ub;    // [B][max_depth] same test aid: the objective's upper bound in force when decision i of the stack was taken
  PathHeader* g_path_hdr;  // [B] same test aid
  Decision* g_dec;   // [B][max_depth] first segment of every workgroup's decision stack
  Decision* dec_pool;  // [dec_pool_segments][max_depth] further segments, taken by the workgroups whose search goes deeper
  BlockStats* g_stats;
  Ctrl* ctrl;
  TeamGrid* teams;           // store layout 5: the per-XCD team control blocks (nullptr otherwise)
  int team_all, team_split;  // store layout 5.  team_all, test aid (TB_TEAM_ALL=1): the whole grid is ONE team whatever the XCDs (agent-scope accesses are coherent device-wide: slower,
                             // same results); team_split (TB_TEAM_SPLIT, 1 / 2 / 4): teams per XCD -- more independent searches, fewer members per barrier, k stores in the XCD's L2
  int team_relaxed;           // team barrier with relaxed agent-scope atomics and an explicit wait for the wave's own memory operations instead of acq_rel fences (TB_TEAM_RELAXED=1)
  int team_join_ticks;        // wall-clock ticks a workgroup waits for the whole grid to register before it gives up (kernels.hpp: team_join; 10 s, TB_TEAM_JOIN_MS)
  PeerCell* cell;            // this device's cell
  PeerCell* const* peers;    // [world] cells of every rank (peers[rank] == cell; nullptr = not reachable), device array
  SolutionRing ring;
  // Basic-block execution counts of an INSTRUMENTED build (scripts/instr_blocks.py rewrites the compiled assembly of the headline kernels: one s_atomic_add per straight-line
  // segment into this array, one copy per XCD, BLK_COUNT_STRIDE bytes apart).  No compiled C++ code reads it: the rewritten kernels fetch it from the kernel arguments by
  // its offset.  nullptr unless TB_BLOCK_COUNTS is set when the session is created.
  unsigned* blk_counts;
};
constexpr size_t BLK_COUNT_STRIDE = 1u << 16;  // bytes per XCD copy (16 384 segments)

}  // namespace tb
