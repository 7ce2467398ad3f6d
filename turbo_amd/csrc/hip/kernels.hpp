// HIP kernels of the dive-and-solve engine for gfx950 (MI355X, CDNA4).
//
// One EPS subproblem per workgroup at a time; the variable-domain store of the subproblem lives in
// LDS (int2 {lb,ub} per variable, ds_read_b64 / ds_max_i32 / ds_min_i32; Booleans as 2 bits in the COMPACT
// layout of the event kernels), the propagator bytecodes are 16-byte records read one per lane (coalesced
// global_load_dwordx4, or ds_read_b128 when they fit in LDS too).  Sweeping fixpoints (AC1 / WAC1): the "has
// changed" flag is reduced per wave with a ballot and published through one LDS word, one s_barrier separates
// two sweeps.  Event-driven fixpoint: rounds over a bitmap of dirty 64-propagator slices, one s_barrier per round.
//
// Mirrors, without their data structures:
//   gpu_barebones_solve   include/barebones_dive_and_solve.hpp:620-901   (workgroup main loop)
//   propagate             include/barebones_dive_and_solve.hpp:903-1031, include/gpu_dive_and_solve.hpp:287-368
//   split / push_decision include/barebones_dive_and_solve.hpp:187-405
#pragma once

#include "device_types.hpp"
#include "propagators.hpp"

// Build-time variants of the 256-thread event kernel (experiments): waves per SIMD the register allocator is asked for,
// software prefetch of the next slice's records (a loss below 5 waves' worth of registers: the prefetched records spill).
#ifndef TB_EVENT_WAVES_256
#define TB_EVENT_WAVES_256 7  // (r03: 6 -- trains15's slab then left room for six workgroups per CU whatever the registers; with the change list trimmed to the LDS granule, seven fit: +4.7 %)
#endif
#ifndef TB_EVENT_WAVES_C8
#define TB_EVENT_WAVES_C8 6  // two-wave workgroups on COMPACT8 slabs: LDS holds eleven or twelve of them per CU (trains15), so 6 waves per SIMD -- 80 VGPRs
#endif
#ifndef TB_EVENT_WAVES
#define TB_EVENT_WAVES 7  // (r06, measured again: 6 -- 80 VGPRs, 6 instead of 12 VGPR spills, 49 instead of 58 SGPRs spilled to lanes, twelve workgroups per CU -- is 3.7 % slower on wordpress7_500)
#endif
// Which parts of an event kernel's node are functions of their own (bit 0: the fixpoint, bit 1: node bookkeeping, bit 2: variable
// selection).  Measured on wordpress7_500: a call costs more than it saves here -- the callee-saved registers go through scratch
// memory on every call and return (3.37e7 nodes/s inlined, 2.74e7 with all three outlined) -- so the default is 0.
#ifndef TB_OUTLINE
#define TB_OUTLINE 0
#endif
// Event kernels: fetch the next slice's successor record (4 registers) while the current slice runs.  Measured on wordpress7_500, same
// box: 3.51e7 nodes/s without, 2.59e7 with -- loads return in order, so every `s_waitcnt vmcnt(0)` the compiler places inside a run
// (it cannot count across the branches of the bodies) also waits for the prefetch that was just issued.  Off.
#ifndef TB_PREGATHER
#define TB_PREGATHER 1  // the sweeps over stores in global memory gather a slice's operands one slice ahead (kernels.hpp: fixpoint, PREG)
#endif
#ifndef TB_DYNAMIC_DEEP
#define TB_DYNAMIC_DEEP 1  // also the WAC1 sweeps of the LDS-resident 1024-thread kernels, from DYN_MIN_PROPS propagators on (0: A/B)
#endif
#ifndef TB_TEAM_DYNAMIC
#define TB_TEAM_DYNAMIC 1  // the waves of a team workgroup take their slices from a counter in LDS (kernels.hpp: fixpoint, DYN)
#endif
#ifndef TB_TEAM_LOCAL_FENCE
#define TB_TEAM_LOCAL_FENCE 1  // (0: the WAC1 pass of a team does not wait for its narrowings before the next local pass -- A/B)
#endif
#ifndef TB_TEAM_WG_PER_CU
#define TB_TEAM_WG_PER_CU 1  // resident 1024-thread workgroups per CU in the team kernel (2: the registers are capped at 64 per lane)
#endif
#ifndef TB_SC_PREFETCH
#define TB_SC_PREFETCH 0
#endif
// (r06, measured and dropped: the same for the event kernel of stores in global memory (hot tier), both records of a wave's next slice fetched one run ahead -- there every
//  load of a run is a global load and returns are in order, so the waits inside a run cost nothing extra: synthetic 100k x 500k 4.00e4 -> 3.97e4 nodes/s, same box.  The records
//  hit the L2; what the kernel waits for is the memory system serving random gathers.)

#if TB_OUTLINE & 1
#define TB_FIX_ATTR __noinline__
#else
#define TB_FIX_ATTR __forceinline__
#endif
#if TB_OUTLINE & 2
#define TB_NODE_ATTR __noinline__
#else
#define TB_NODE_ATTR __forceinline__
#endif
#if TB_OUTLINE & 4
#define TB_SPLIT_ATTR __noinline__
#else
#define TB_SPLIT_ATTR __forceinline__
#endif

namespace tb {

#define TB_RLX __ATOMIC_RELAXED
#define TB_WG __HIP_MEMORY_SCOPE_WORKGROUP
#define TB_AGENT __HIP_MEMORY_SCOPE_AGENT
#define TB_SYS __HIP_MEMORY_SCOPE_SYSTEM

constexpr int MAX_WAVES = 16;          // 1024 threads
constexpr int MAX_DEC_SEGS = 15;       // extra decision-stack segments a workgroup can take from the pool (16 x max_depth decisions in all)
constexpr int NODE_BATCH = 32;         // nodes a workgroup explores between two updates of the node-wide node counter
constexpr int WAVE_WATCHDOG_PERIOD = 1024;  // wave-local iterations between two looks at the deadline / abort flag

// Host-pinned page shared with the host thread (replaces the managed-memory flags of
// barebones UnifiedData::stop, barebones:64, and the 100 ms wait loop of memory_gpu.hpp:174-196).
// One workgroup per poll period (DevProblem::poll_ticks of wall clock, elected through Ctrl::next_poll) reads the
// host -> device words and refreshes the device -> host ones: the traffic over PCIe does not depend on the grid size
// nor on how long a node takes.
struct Mailbox {
  int stop;           // host -> device
  int foreign_bound;  // host -> device: incumbent found by another GPU (host relay; peers normally write PeerCell::bound directly)
  int local_best;     // device -> host: incumbent found on this GPU (written on improvement, refreshed by every poll)
  int polls;          // device -> host: number of polls so far (liveness)
  unsigned long long progress;  // device -> host: PeerCell::queue at the last poll (remaining work of this GPU)
  unsigned long long pad;
};

// Workgroup control block, first bytes of the dynamic LDS segment.
constexpr int DYN_MIN_PROPS = 32 * 16 * 64;  // slices on demand in the LDS-resident WAC1 sweeps (fixpoint: DYN): from 32 slices per wave on

struct alignas(16) BlockShared {
  int bot;        // VStore::is_bot
  int abort;      // the watchdog fired (must follow `bot`: the hot loops read both with one 8-byte load, dead_node)
  int flag[3];    // "some domain changed during sweep k", rotating so one barrier per sweep is enough
  int unent[3];   // "some propagator is not entailed", computed in the same sweep
  int leaf, stop, depth, remaining;
  int cur_strategy, next_unassigned, snap_strategy, snap_next_unassigned;
  int best_bound;  // best objective found by this workgroup (BlockData::best_bound, barebones:116)
  int found, sol, skip, open_vars;  // open_vars: the all-assigned scan of an all-entailed node found a variable that is not (leaf rule of the `gpu` path)
  int new_depth, ev_all, chg_count[2], team_res;  // event mode: "run every slice" request, change-list fill; team_res: merged flags of the last team barrier (layout 5)
  unsigned long long sub_idx;  // global index of the current subproblem
  unsigned long long sub_j;    // its index in the local numbering of rank sub_owner (eps_global_index)
  int sub_owner, sub_gen;      // rank whose share it belongs to; generation of the queue range it was fetched from
  int has_work, witness;       // witness: index of a propagator found un-entailed (fixpoint_event), -1 = none
  long long ticket;  // streaming: sequence number of the solution being handed to the host, -1 if none
  Decision* dec_seg[MAX_DEC_SEGS];  // segments 1.. of this workgroup's decision stack (segment 0 is its slab in g_dec)
  int n_dec_seg, team;        // team (layout 5): member index | team size << 12 | control block (XCD x split + k) << 24
  long long t_start, t_mark;  // thread 0's clocks (kernel start, last phase boundary): LDS, not registers that live through every loop
  long long t_dive;           // start of the current dive (0: not diving)
  int last_obj_ub, team_gen;  // upper bound last imposed on the objective in this subproblem (PINF: none): test aid, tb_session_debug_path; team_gen: barriers passed (layout 5)
  int team_slab, pad_slab;    // layout 5: the workgroup whose slabs of g_store / g_snap the team works on (its leader's blockIdx)
  unsigned long long red_key[MAX_WAVES];
  int red_first[MAX_WAVES];
  BlockStats bs;  // written by thread 0 only
};

constexpr int SH_BYTES = (int)((sizeof(BlockShared) + 15) / 16 * 16);

__device__ __forceinline__ bool dead_node(const BlockShared& sh) {
  static_assert(offsetof(BlockShared, abort) == offsetof(BlockShared, bot) + 4 && offsetof(BlockShared, bot) % 8 == 0, "bot and abort are read as one 8-byte word");
  return __hip_atomic_load(reinterpret_cast<const long long*>(&sh.bot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0ll;
}

// Device-side tuning / profiling knobs (tb_config.reserved[0], see include/turbo_hip.h).  They sit in the hot loops,
// so a production build compiles them out; build with -DTB_TUNING (make hip EXTRA_HIPFLAGS=-DTB_TUNING) for
// scripts/ablate.py, scripts/phase_probe.py and the like.
#ifdef TB_TUNING
__device__ __forceinline__ int knobs(const DevProblem& P) { return P.debug; }
#else
__device__ __forceinline__ constexpr int knobs(const DevProblem&) { return 0; }
#endif

// Tuning build: bits 8-15 of the knob word select ONE phase of the event kernels to execute twice (all of them are idempotent), so
// that the difference in SQ_INSTS_VALU / SQ_INSTS_SALU to an undoubled run is that phase's instruction count
// (scripts/phase_budget.py): 1 seeding from the change list, 2 round scan, 3 a whole slice run (fetch + dispatch + set-up + one
// evaluation pass + the early-out of the marks), 4 witness test, 5 variable-selection scan, 6 restore copy, 7 dirty-bitmap clear,
// 8 best-store copy, 9 the prologue of a run outside the doubled part (scalar info load, record base, failure check), 10 the end of a round.  (0x1: successor marks, 0x4: snapshot push, 0x8: one evaluation pass -- older knobs.)
// For the instruction budget of profiles/ the doubled phase is chosen at COMPILE time (-DTB_DOUBLE_PHASE=n, `make phases`: one library per
// phase, each the production kernels with exactly that phase executed twice): run-time knobs in the hot loops changed the kernel they were
// meant to measure (14.5k VALU per node with them against 11.3k without).  Phases 11 / 12 / 13 / 14 = the older knobs 0x1 (marks), 0x4
// (snapshot push), 0x8 (one evaluation pass), 0x400000 (count slice runs instead of wave iterations).
#ifndef TB_DOUBLE_PHASE
#define TB_DOUBLE_PHASE 0
#endif
#ifdef TB_TUNING
__device__ __forceinline__ int pk(const DevProblem& P) { return P.debug; }
#else
__device__ __forceinline__ constexpr int pk(const DevProblem&) {
  return TB_DOUBLE_PHASE == 11 ? 0x1 : (TB_DOUBLE_PHASE == 12 ? 0x4 : (TB_DOUBLE_PHASE == 13 ? 0x8 : (TB_DOUBLE_PHASE == 14 ? 0x400000 : (TB_DOUBLE_PHASE << 8))));
}
#endif
__device__ __forceinline__ int reps_of(const DevProblem& P, int phase) { return ((pk(P) >> 8) & 0xff) == phase ? 2 : 1; }

// Regions of the search kernel for the instruction budget (scripts/region_budget.py): a -DTB_REGION_MARKERS build is the production kernel with an
// assembler comment at each point -- static instruction counts between two markers, in layout order -- and the tuning build counts how often a
// wave passes each point (BlockStats::reg; wave-uniform points only).
// In-kernel phase timers of the tuning build (knob 0x10000): the engine's own phases, apart from the reference's timers (BlockStats::prof, tb_stats.prof_ns)
#ifdef TB_TUNING
#define TB_PROF_ADD(bs, which, dt) ((bs).prof[which] += (dt))
#else
#define TB_PROF_ADD(bs, which, dt) ((void)(dt))
#endif
#if defined(TB_REGION_MARKERS)
#define TB_REGION(id) asm volatile("; TBREGION " #id)
#elif defined(TB_TUNING)
#define TB_REGION(id) do { if ((threadIdx.x & 63) == 0) (void)__hip_atomic_fetch_add(&sh.bs.reg[id], 1u, TB_RLX, TB_WG); } while (0)
#else
#define TB_REGION(id) do { } while (0)
#endif

// LDS pointers across a call boundary: a pointer argument of a non-inlined function is GENERIC (flat_load / flat_atomic, both
// counters, an aperture check per access) unless its address space travels with it.  The outlined functions below take 32-bit LDS
// offsets and rebuild address-space-3 pointers, from which the compiler infers ds_* instructions for everything inlined under them.
#define TB_LDS __attribute__((address_space(3)))
template <class T> __device__ __forceinline__ unsigned lds_off(T* p) { return (unsigned)(size_t)(TB_LDS T*)p; }
template <class T> __device__ __forceinline__ T* lds_ptr(unsigned off) { return (T*)(TB_LDS T*)(size_t)off; }
// A pointer read from the problem description in memory is generic as well (flat_load: counted by vmcnt AND lgkmcnt, so every wait
// for an LDS access also waits for the record fetch in flight).  glob() says "this one is global memory": global_load, vmcnt only.
#define TB_GLB __attribute__((address_space(1)))
#define TB_CST __attribute__((address_space(4)))
// ... and the problem description itself is constant for the whole launch: through the constant address space its fields are
// fetched with scalar loads (s_load_dword) where they are used, like kernel arguments.
struct DevProblem;
__device__ __forceinline__ const DevProblem& constant_problem(const DevProblem* p);
template <class T> __device__ __forceinline__ const T* cst(const T* p) { return (const T*)(TB_CST const T*)(size_t)p; }  // read-only for the whole launch: scalar loads
template <class T> __device__ __forceinline__ T* glob(T* p) { return (T*)(TB_GLB T*)(size_t)p; }  // (through an integer: a generic -> global -> generic cast pair folds away)

__device__ __forceinline__ const DevProblem& constant_problem(const DevProblem* p) { return *(const DevProblem*)(TB_CST const DevProblem*)(size_t)p; }
// The watchdog's test (once per 1024 wave-local iterations / 256 rounds or sweeps).  (r04 tried fetching the deadline through an opaque constant-address-space
// pointer so that it would not be hoisted and spilled: neutral on the event kernels, and WRONG for the kernels that take the problem description by value --
// its address is then not in constant memory: HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION the first time a watchdog branch ran.)
__device__ __forceinline__ bool deadline_passed(const DevProblem& P) {
  return P.deadline_ticks != 0 && wall_clock64() > P.deadline_ticks;
}

// Wave votes straight on the lane mask a comparison leaves in an SGPR pair (HIP's __any / __ballot take an int: the bool is first
// turned into 0/1 with a v_cndmask and compared again -- two VALU instructions per vote on an issue-bound kernel).
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// The lane mask of "x != 0" as ONE v_cmp.  (A ballot of a bool that is itself a combination of lane masks is re-materialised by the
// compiler as v_cndmask 0/1 + v_cmp to clear the lanes outside exec: the hot bodies keep their predicates as integers and vote once.)
__device__ __forceinline__ unsigned long long mask_nz(unsigned x) {
  unsigned long long m;
  asm volatile("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(m) : "v"(x));
  return m;
}

// Threads of the workgroup.  TB != 0: a compile-time constant -- the two-wave event kernels of the search are only ever launched with exactly
// 128 threads (engine.hip: plan_launch), and with the size known the strides, trip counts and alignment cases of every block-wide copy and scan fold
// away instead of being hoisted into SGPRs that stay live (and spill) through the whole persistent loop.
template <int TB> __device__ __forceinline__ int block_threads() { return TB != 0 ? TB : (int)blockDim.x; }
// A value the optimiser may not reason about: what is computed from it stays where it is written.  The persistent kernel is one huge loop nest, and
// everything loop invariant -- per-lane addresses of the block-wide copies, `1 << lane`, trip counts -- is otherwise hoisted to its top and kept (in
// scratch, or in lanes of a spill register) for the whole search, at the expense of the registers the hot loops need.
__device__ __forceinline__ int here(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int here_s(int x) { asm volatile("" : "+s"(x)); return x; }  // ... a wave-uniform one
struct DevProblem;
__device__ __forceinline__ bool deadline_passed(const DevProblem& P);

__device__ __forceinline__ int ld(const int* p) { return __hip_atomic_load(p, TB_RLX, TB_WG); }
__device__ __forceinline__ void st(int* p, int v) { __hip_atomic_store(p, v, TB_RLX, TB_WG); }

// ---- software bounds build (-DTB_BOUNDS, `make bounds`; r05) --------------------------------------------------------------------------------
// GPU AddressSanitizer is not available on the pool, and r04 saw ONE unexplained memory fault in ~40 traced runs.  The kernels form LDS and global addresses
// from host-packed fields (16-bit word indexes, COMPACT8 bases, 16-bit slice ids, 28-bit queue words): in this build every such index goes through TB_IDX,
// which compares it with the limit the host computed for THIS launch (BoundsLimits, hipMemcpyToSymbol before the launch), records the first offender
// {site, index, limit, workgroup, thread}, asks the grid to stop and substitutes index 0 -- a report instead of a fault (or of a silent LDS over-read:
// out-of-range LDS accesses do not fault at all).  tb_session_finish / tb_propagate turn a report into TB_ERR_HIP.  Production builds compile TB_IDX to
// the index itself.  Sites: 1 interval of a plain / COMPACT / HOT slab, 2 word of a COMPACT16 slab, 3 word of a COMPACT8 integer, 4-6 load_dom of COMPACT8 /
// COMPACT16 / COMPACT, 7-8 narrowing of an integer, 9 Boolean word, 10 slice id into a dirty bitmap, 11 variable into var_adj, 12 adj_rest entry, 13 slice
// of a run (slice_info / records / successor records), 14 word indexes of a lean implication record, 15 decision-stack entry, 16 strategy tables,
// 17 snapshot level, 18 entailment mark of a slice, 19 propagator index (witness, sweeps), 20 change-list entry, 21 all-assigned scan, 22 Boolean column of a run.
#ifdef TB_BOUNDS
struct BoundsLimits {
  int store_words;      // 32-bit words of a workgroup's slab (vext * 2): domains + entailment marks
  int slab_vars;        // variables of the slab (DevProblem::n_vars)
  int n_slices, records;  // slices; records = n_slices * 64 (the arrays are padded to whole slices)
  int adj_vars;         // entries of var_adj / 2
  int adj_rest;         // entries of adj_rest
  int strats, strat_total;
  int snapshot_levels;
  int mark_words;       // 32-bit words of entailment marks behind the domains (event: dirty_words; sweeps: bytes / 4 rounded up)
  int chg_cap;
  Ctrl* ctrl;           // the search's grid words (nullptr: batch propagation)
};
struct BoundsReport { unsigned hits; int site, index, limit, workgroup, thread; };
__device__ BoundsLimits g_bl;
__device__ BoundsReport g_br;
__device__ __noinline__ void bounds_hit(int site, int idx, int limit) {
  if (atomicAdd(&g_br.hits, 1u) == 0u) { g_br.site = site; g_br.index = idx; g_br.limit = limit; g_br.workgroup = (int)blockIdx.x; g_br.thread = (int)threadIdx.x; }
  if (g_bl.ctrl != nullptr) (void)__hip_atomic_fetch_or(&g_bl.ctrl->stop, STOP_HOST, TB_RLX, TB_AGENT);
}
__device__ __forceinline__ int bounds_idx(int site, int idx, int limit) {
  if ((unsigned)idx >= (unsigned)limit) { bounds_hit(site, idx, limit); return 0; }
  return idx;
}
#define TB_IDX(site, idx, lim) bounds_idx(site, (int)(idx), g_bl.lim)
#define TB_IDX_N(site, idx, n) bounds_idx(site, (int)(idx), (int)(n))
__device__ __forceinline__ int g_bl_total() { return g_bl.strat_total > 0 ? g_bl.strat_total : 1; }
// an index of 8-byte intervals (int2) into the slab: both of its words must lie inside
#define TB_CHECK_ITV(site, v) do { (v) = bounds_idx(site, 2 * (int)(v) + 1, g_bl.store_words) >> 1; } while (0)
#else
#define TB_CHECK_ITV(site, v) do { } while (0)
#define TB_IDX(site, idx, lim) (idx)
#define TB_IDX_N(site, idx, n) (idx)
__device__ __forceinline__ constexpr int g_bl_total() { return 1; }
#endif

#ifdef TB_TRAP_SEED
// Debugging aid (r06): the first workgroup that meets something impossible writes what it saw into Ctrl::trap and stops the search (tb_session_finish prints it).
__device__ __noinline__ void trap_report(const DevProblem& P, BlockShared& sh, int code, const int* words, int n) {
  Ctrl* c = glob(P.ctrl);
  if (__hip_atomic_exchange(&c->error, 3, TB_RLX, TB_AGENT) != 0) return;
  c->trap[0] = code; c->trap[1] = (int)blockIdx.x; c->trap[2] = (int)sh.bs.nodes; c->trap[3] = sh.depth; c->trap[4] = sh.new_depth; c->trap[5] = sh.remaining; c->trap[6] = sh.n_dec_seg; c->trap[7] = n;
  for (int i = 0; i < n && i < 40; ++i) c->trap[8 + i] = words[i];
  (void)__hip_atomic_fetch_or(&c->stop, STOP_HOST, TB_RLX, TB_AGENT);
}
#endif

// ---- workgroup teams (store layout 5, r05) -----------------------------------------------------------------------------------------------
// The workgroups resident on one XCD search ONE subproblem together, on ONE store in global memory (device_types.hpp: TeamCtl):
//   * the store is read with agent-scope relaxed loads (`global_load ... sc1`: served by the XCD's L2, never by a CU's L1) and narrowed with agent-scope
//     atomic max / min -- coherent for every CU whatever its placement, and L2 hits for the members of a team because they share that L2;
//   * a sweep is PARTITIONED: member m of M evaluates the slices m, m + M, ... of every wave's share; the has-changed / not-entailed / failed / aborted
//     flags of the members are merged by one team barrier per sweep (team_sync);
//   * everything else is REPLICATED: every member runs the same control code on the same store -- same variable selection, same decision stack (a copy
//     each), same backtracking -- so no decision has to be communicated; what is not a function of the store (the subproblem fetched from the queue, the
//     incumbent read from the grid words, a stop request) is read by the team's leader (member 0) and broadcast through TeamCtl::bcast before a barrier;
//   * block copies of the store (root restore, snapshot push / restore) are striped over the members; statistics are kept by the leader.
// Teams are formed at kernel start from where the workgroups actually are (HW_REG_XCC_ID), not from an assumed dispatch order.
constexpr unsigned TEAM_CHANGED = 1u, TEAM_UNENT = 2u, TEAM_BOT = 4u, TEAM_ABORT = 8u, TEAM_STOP = 16u, TEAM_WORK = 32u;
__device__ __forceinline__ int team_member(const BlockShared& sh) { return sh.team & 0xfff; }
__device__ __forceinline__ int team_size(const BlockShared& sh) { return (sh.team >> 12) & 0xfff; }
__device__ __forceinline__ int team_xcd(const BlockShared& sh) { return (sh.team >> 24) & 0x3f; }  // the team's slot: XCD x split + k
__device__ __forceinline__ TeamCtl* team_ctl(const DevProblem& P, const BlockShared& sh) { return &glob(P.teams)->team[team_xcd(sh)]; }

// Team barrier (uniform call, every thread of every member): returns the OR of the members' `contrib` (thread 0's value counts).
// Before it, every wave waits for its own outstanding memory operations (the narrowings are non-returning atomics: `vmcnt` counts them), the arrival is
// an agent-scope acq_rel atomic; after it, agent-scope loads see everything the members wrote before they arrived.  Sense reversal by generation: the last
// arriver collects the flags, clears the counter and publishes generation + 1; result[g & 1] cannot be overwritten before every member has read it
// (generation g + 2 needs every member's arrival at g + 2).
__device__ __forceinline__ unsigned team_sync(const DevProblem& P, BlockShared& sh, unsigned contrib) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    TeamCtl* t = team_ctl(P, sh);
    const unsigned g = (unsigned)sh.team_gen, M = (unsigned)team_size(sh);
    unsigned res = contrib;
    if (M > 1) {
      if (contrib) (void)__hip_atomic_fetch_or(&t->flags, contrib, TB_RLX, TB_AGENT);
      // (relaxed form: the store is only ever touched with agent-scope atomics and loads, which no cache holds dirty or stale -- every wave has waited for its own above)
      const unsigned old = P.team_relaxed ? __hip_atomic_fetch_add(&t->arrive, 1u, TB_RLX, TB_AGENT) : __hip_atomic_fetch_add(&t->arrive, 1u, __ATOMIC_ACQ_REL, TB_AGENT);
      if (old == M - 1) {
        res = __hip_atomic_exchange(&t->flags, 0u, TB_RLX, TB_AGENT);
        __hip_atomic_store(&t->result[g & 1], res, TB_RLX, TB_AGENT);
        __hip_atomic_store(&t->arrive, 0u, TB_RLX, TB_AGENT);
        if (P.team_relaxed) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __hip_atomic_store(&t->gen, g + 1, TB_RLX, TB_AGENT); }
        else __hip_atomic_store(&t->gen, g + 1, __ATOMIC_RELEASE, TB_AGENT);
      } else {
        bool lost = false;
        for (unsigned spins = 1; __hip_atomic_load(&t->gen, TB_RLX, TB_AGENT) == g; ++spins) {
          __builtin_amdgcn_s_sleep(2);
          // (a team that has fallen apart -- a member that never arrives -- must not outlive the search's deadline: everybody left gives up, as after a watchdog abort)
          if ((spins & 4095u) == 0u && deadline_passed(P)) { lost = true; break; }
        }
        if (!P.team_relaxed) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        res = lost ? (contrib | TEAM_STOP | TEAM_ABORT) : __hip_atomic_load(&t->result[g & 1], TB_RLX, TB_AGENT);
        if (lost) { sh.abort = 1; sh.stop = 1; }
      }
    }
    sh.team_gen = (int)(g + 1);
    sh.team_res = (int)res;
  }
  __syncthreads();
  return (unsigned)sh.team_res;
}
// Leader -> members: four 64-bit words, written by the leader's thread 0 BEFORE a team_sync and read by anybody after it.
__device__ __forceinline__ void team_post(const DevProblem& P, const BlockShared& sh, int slot, unsigned long long v) {
  __hip_atomic_store(&team_ctl(P, sh)->bcast[slot], v, TB_RLX, TB_AGENT);
}
__device__ __forceinline__ unsigned long long team_read(const DevProblem& P, const BlockShared& sh, int slot) {
  return __hip_atomic_load(&team_ctl(P, sh)->bcast[slot], TB_RLX, TB_AGENT);
}
// Join the team of the XCD this workgroup runs on (thread 0; kernel start).  The plan launches at most one workgroup per CU, so normally the whole grid is resident and
// everybody registers within microseconds; after that the member counts are final.  The launch is an ordinary one, though: another process on the same GPU, a CU mask
// or a concurrent persistent kernel can keep part of the grid from becoming resident.  The wait is therefore bounded by a wall-clock limit of its own
// (DevProblem::team_join_ticks, 10 s; independent of timeout_ms): whoever runs into it poisons the registration word (TEAM_JOIN_POISON), so that workgroups that
// start later do not form teams with members that have left, reports Ctrl::error = 2 (tb_session_finish: TB_ERR_STATE "team formation failed") and goes on as a team
// of one that has been told to stop -- no team barrier ever waits for anybody.
constexpr unsigned TEAM_JOIN_POISON = 0x80000000u;
__device__ __forceinline__ void team_join(const DevProblem& P, BlockShared& sh) {
  TeamGrid* G = glob(P.teams);
  const unsigned xcc = P.team_all ? 0u : ((unsigned)__builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((4 - 1) << 11)) & 7u);
  const unsigned split = P.team_all ? 1u : (unsigned)(P.team_split > 1 ? P.team_split : 1);  // 1, 2, 4 or 8
  const unsigned mx = __hip_atomic_fetch_add(&G->xcd_members[xcc], 1u, TB_RLX, TB_AGENT);  // my arrival on this XCD: dealt to its teams in turn
  const unsigned k = mx % split, m = mx / split;
  // the team's slabs (store, snapshot stack) are those of its LEADER's workgroup: a slot below num_blocks whatever the grid size (r05 used XCD x split + k, which runs up to
  // 63 and indexed past g_store / g_snap for grids smaller than that -- `-or 8`, or the 2e8 / n_vars cap of a very large network)
  if (m == 0) __hip_atomic_store(&G->team[xcc * split + k].pad[0], (unsigned)blockIdx.x, TB_RLX, TB_AGENT);
  unsigned seen = __hip_atomic_fetch_add(&G->registered, 1u, __ATOMIC_RELEASE, TB_AGENT) + 1u;
  const long long t_join = wall_clock64();
  for (unsigned spins = 1; (seen & ~TEAM_JOIN_POISON) < gridDim.x && !(seen & TEAM_JOIN_POISON); ++spins) {
    __builtin_amdgcn_s_sleep(8);
    if ((spins & 1023u) == 0u && (deadline_passed(P) || wall_clock64() - t_join > (long long)P.team_join_ticks)) {
      (void)__hip_atomic_fetch_or(&G->registered, TEAM_JOIN_POISON, TB_RLX, TB_AGENT);
    }
    seen = __hip_atomic_load(&G->registered, __ATOMIC_ACQUIRE, TB_AGENT);
  }
  sh.team_gen = 0; sh.team_res = 0;
  if (seen & TEAM_JOIN_POISON) {  // the grid never became co-resident: give up loudly instead of hanging
    __hip_atomic_store(&glob(P.ctrl)->error, 2, TB_RLX, TB_AGENT);
    (void)__hip_atomic_fetch_or(&glob(P.ctrl)->stop, STOP_HOST, TB_RLX, TB_AGENT);
    sh.abort = 1; sh.stop = 1;
    sh.team = (int)(0u | (1u << 12) | ((xcc * split + k) << 24));
    sh.team_slab = (int)blockIdx.x;
    return;
  }
  const unsigned Mx = __hip_atomic_load(&G->xcd_members[xcc], TB_RLX, TB_AGENT);
  const unsigned M = Mx / split + (k < Mx % split ? 1u : 0u);
  sh.team = (int)(m | (M << 12) | ((xcc * split + k) << 24));
  sh.team_slab = (int)__hip_atomic_load(&G->team[xcc * split + k].pad[0], TB_RLX, TB_AGENT);
}

// Store slab of a workgroup: [ni x int2 {lb,ub}] [Boolean words: 16 variables x 2 bits] [one byte per slice].
// Variables >= ni are the Boolean ones of the COMPACT layout (root domain within 0..1): bit 2k of their word says
// "lb raised to 1", bit 2k+1 "ub lowered to 0" -- narrowing is a `ds_or` (monotone like max/min), 3 = empty.
// Without COMPACT ni = n_vars and there are no Boolean words.
// Layouts of a workgroup's store slab (template parameter C of everything below):
//   0  PLAIN      [n_vars x int2 {lb,ub}]
//   1  COMPACT    [n_int x int2][Boolean words: 16 variables x 2 bits][entailment bits]   -- variables >= ni are 2-bit Booleans
//   2  COMPACT16  [n_int x u32 {lb:16 | ub:16 << 16}][Boolean words][entailment bits]      -- as COMPACT, and every other variable has a
//                 root domain within -32768..32767: half the bytes per integer, twice the workgroups per CU for a network like trains15
// A 16-bit bound cannot be narrowed with one 32-bit atomic max / min (the other half shares the word), so COMPACT16 narrows with a
// compare-and-swap loop; narrowings are two orders of magnitude rarer than reads.
//   3  HOT        the PLAIN layout of a store in global memory whose first HOT_VARS intervals live in LDS instead (r04; the 1024-thread kernels of
//                 networks too large for LDS, e.g. the synthetic 100k x 500k one: a store of 800 KB per workgroup is gathered at random -- three of four
//                 gathers miss the XCD's L2 and each moves a 64-byte line for 8 useful bytes -- while the 160 KB of LDS of its CU sit idle).  The host
//                 numbers the most-read variables first (engine.hip: renumber_by_reads); an access picks its address space per lane and goes out as one
//                 FLAT instruction (the hardware routes each lane by aperture).  The slab in global memory keeps room for all variables: block copies
//                 (snapshots, best store) gather the hot part from LDS.
//   4  COMPACT8   [n_wide x u32 {lb:16 | ub:16 << 16}][n_narrow x u16 {lb - base : 8 | ub - base : 8}][Boolean words][entailment bits]  (r04)
//                 as COMPACT16, and an integer whose root domain is at most 255 wide takes two BYTES: its bounds relative to the root lower bound
//                 (`base`).  The base travels with every reference to the variable -- an operand field of a record, an entry of a strategy list, the
//                 objective, a decision -- in bits 16-30 of the field (base + 16384; bits 0-15: the variable), so a load is still one ds_read_b32 and a
//                 few VALU, and a rule computes on absolute values as everywhere else.  trains15: 4939 of its 5011 integers are narrow, the slab goes from
//                 20.7 KB to 11.6 KB and two-wave workgroups fit eleven to a CU instead of four-wave ones seven.  `ni` of this layout is
//                 n_int | n_wide << 16 (the wide integers are numbered first, then the narrow ones, then the Booleans).
//   5  TEAM       the PLAIN layout of a store in global memory shared by the workgroups of an XCD (r05, see "workgroup teams" above): agent-scope loads and atomics.
constexpr int C8_BASE_BIAS = 16384;
template <int C> __device__ __forceinline__ int ni_int(int ni) { return C == 4 ? (ni & 0xffff) : ni; }   // integer variables of the slab
__device__ __forceinline__ int ni_wide(int ni) { return (int)((unsigned)ni >> 16); }                      // COMPACT8: the wide ones
// the variable of a reference (COMPACT8: without its base; other layouts: the reference itself).  Not for constants (sign bit set).
template <int C> __device__ __forceinline__ int var_of(int f) { return C == 4 ? (f & 0xffff) : f; }
constexpr int HOT_VARS = 19456;  // 152 KB of intervals, right behind the control block (with the bitmaps and the change list: 159 KB of the 160)
template <class T>
__device__ __forceinline__ T* hot_or_cold(T* cold, int v) {  // (generic pointers on both sides: a flat access)
  T* hot = const_cast<T*>(reinterpret_cast<const T*>(lds_ptr<int2>((unsigned)SH_BYTES) + v));
  return v < HOT_VARS ? hot : cold;
}
template <int C>
__device__ __forceinline__ Itv load_int(const int2* store, int v) {  // an integer (non-Boolean) variable of the layout
  Itv d;
  if (C == 3) {
    v = TB_IDX(1, v, slab_vars);
    const long long raw = __hip_atomic_load(reinterpret_cast<const long long*>(hot_or_cold(store + v, v)), TB_RLX, TB_WG);
    d.lb = (int)(raw & 0xffffffffll);
    d.ub = (int)(raw >> 32);
    return d;
  }
  if (C == 5) {
    TB_CHECK_ITV(1, v);
    const long long raw = __hip_atomic_load(reinterpret_cast<const long long*>(store + v), TB_RLX, TB_AGENT);  // global_load_dwordx2 sc1: the XCD's L2
    d.lb = (int)(raw & 0xffffffffll);
    d.ub = (int)(raw >> 32);
    return d;
  }
  if (C == 2) {
    v = TB_IDX(2, v, store_words);
    const unsigned w = __hip_atomic_load(reinterpret_cast<const unsigned*>(store) + v, TB_RLX, TB_WG);
    d.lb = (int)(short)(w & 0xffffu);
    d.ub = (int)w >> 16;
    return d;
  }
  TB_CHECK_ITV(1, v);
  const long long raw = __hip_atomic_load(reinterpret_cast<const long long*>(store + v), TB_RLX, TB_WG);
  d.lb = (int)(raw & 0xffffffffll);
  d.ub = (int)(raw >> 32);
  return d;
}
// COMPACT8: an integer variable from its reference `f` (variable | (base + 16384) << 16); `nw`: the wide integers of the slab.
// A narrow variable v sits in halfword v + nw of the slab (every wide one before it takes two).
__device__ __forceinline__ Itv load_int8(const int2* store, int nw, int f, unsigned* raw = nullptr) {
  const int v = f & 0xffff, base = ((f >> 16) & 0x7fff) - C8_BASE_BIAS;
  const bool wide = v < nw;
  const int h = v + nw;
  const unsigned w = __hip_atomic_load(reinterpret_cast<const unsigned*>(store) + TB_IDX(3, wide ? v : (h >> 1), store_words), TB_RLX, TB_WG);
  if (raw) *raw = w;
  const unsigned pair = w >> ((h & 1) * 16);
  Itv d;
  d.lb = wide ? (int)(short)(w & 0xffffu) : base + (int)(pair & 0xffu);
  d.ub = wide ? (int)w >> 16 : base + (int)((pair >> 8) & 0xffu);
  return d;
}
// first 32-bit word of the Boolean words of a COMPACT / COMPACT16 slab
template <int C>
__device__ __forceinline__ int bool_word0(int ni) { return C == 4 ? ((ni & 0xffff) + ni_wide(ni) + 1) >> 1 : (C == 2 ? ni : 2 * ni); }
template <int C>
__device__ __forceinline__ unsigned* bool_words(int2* store, int ni) { return reinterpret_cast<unsigned*>(store) + bool_word0<C>(ni); }
template <int C>
__device__ __forceinline__ const unsigned* bool_words(const int2* store, int ni) { return reinterpret_cast<const unsigned*>(store) + bool_word0<C>(ni); }

// An operand field with the sign bit set is not a variable but the VALUE of a constant the layout keeps out of the slab (engine.hip: Layout,
// operand_field; COMPACT layouts only): no memory access -- the load below goes to word 0 and is discarded by a select -- and nothing to
// narrow: a rule that would move a constant has emptied it, and the caller raises the failure flag on the empty candidate.
__device__ __forceinline__ int field_value(int v) { return (int)((unsigned)v << 1) >> 1; }
template <int C>
__device__ __forceinline__ Itv load_dom(const int2* store, int ni, int v, unsigned* seen = nullptr) {  // (`seen`, COMPACT8: the word that was read)
  if (C == 0) return load_int<0>(store, v);
  if (C == 3) return load_int<3>(store, v);
  if (C == 5) return load_int<5>(store, v);
  const bool isk = v < 0;
  const int kv = field_value(v);
  if (C == 4) {
    // one 4-byte load whatever the kind: a wide integer's word, the word holding a narrow one's two bytes, or the Boolean word
    const int id = v & 0xffff, nw = ni_wide(ni), n_i = ni & 0xffff, base = ((v >> 16) & 0x7fff) - C8_BASE_BIAS;
    const bool isb = id >= n_i, wide = id < nw;
    const int b = id - n_i, h = id + nw;
    const unsigned w = __hip_atomic_load(reinterpret_cast<const unsigned*>(store) + TB_IDX(4, isk ? 0 : (isb ? bool_word0<4>(ni) + (b >> 4) : (wide ? id : (h >> 1))), store_words), TB_RLX, TB_WG);
    const unsigned bits = (w >> ((b & 15) * 2)) & 3u, pair = w >> ((h & 1) * 16);
    if (seen) *seen = w;
    Itv d;
    d.lb = isk ? kv : (isb ? (int)(bits & 1u) : (wide ? (int)(short)(w & 0xffffu) : base + (int)(pair & 0xffu)));
    d.ub = isk ? kv : (isb ? 1 - (int)(bits >> 1) : (wide ? (int)w >> 16 : base + (int)((pair >> 8) & 0xffu)));
    return d;
  }
  if (C == 2) {
    // one 4-byte load whatever the kind: the integer's packed bounds, or the Boolean word holding the variable's two bits
    const bool isb = v >= ni;
    const int b = v - ni;
    const unsigned w = __hip_atomic_load(reinterpret_cast<const unsigned*>(store) + TB_IDX(5, isk ? 0 : (isb ? ni + (b >> 4) : v), store_words), TB_RLX, TB_WG);
    const unsigned bits = (w >> ((b & 15) * 2)) & 3u;
    Itv d;
    d.lb = isk ? kv : (isb ? (int)(bits & 1u) : (int)(short)(w & 0xffffu));
    d.ub = isk ? kv : (isb ? 1 - (int)(bits >> 1) : (int)w >> 16);
    return d;
  }
  // COMPACT: one 8-byte load whatever the kind of the variable -- an interval, or the pair of Boolean words holding its
  // two bits (the Boolean words start at store + ni, which is 8-byte aligned) -- and selects instead of branches: the
  // three gathers of a propagator are in flight together and a wave with mixed operands does not serialise them.
  Itv d;
  const bool isb = v >= ni;
  const int b = v - ni;
  int idx = isk ? 0 : (isb ? ni + (b >> 5) : v);
  TB_CHECK_ITV(6, idx);
  const long long raw = __hip_atomic_load(reinterpret_cast<const long long*>(store + idx), TB_RLX, TB_WG);
  const int lo = (int)(raw & 0xffffffffll), hi = (int)(raw >> 32);
  const unsigned word = (unsigned)(((b >> 4) & 1) ? hi : lo);
  const unsigned bits = (word >> ((b & 15) * 2)) & 3u;
  d.lb = isk ? kv : (isb ? (int)(bits & 1u) : lo);
  d.ub = isk ? kv : (isb ? 1 - (int)(bits >> 1) : hi);
  return d;
}
// COMPACT16: narrow one half of an integer's word (compare-and-swap: the other bound lives in the same word)
__device__ __forceinline__ void cas_raise_lb16(unsigned* w, int val) {
  val = val > 32767 ? 32767 : val;  // (a bound beyond the representable range empties the domain: the caller has raised the failure flag)
  unsigned old = __hip_atomic_load(w, TB_RLX, TB_WG);
  while ((int)(short)(old & 0xffffu) < val) {
    const unsigned nw = (old & 0xffff0000u) | ((unsigned)val & 0xffffu);
    if (__hip_atomic_compare_exchange_strong(w, &old, nw, TB_RLX, TB_RLX, TB_WG)) break;
  }
}
__device__ __forceinline__ void cas_lower_ub16(unsigned* w, int val) {
  val = val < -32768 ? -32768 : val;
  unsigned old = __hip_atomic_load(w, TB_RLX, TB_WG);
  while (((int)old >> 16) > val) {
    const unsigned nw = (old & 0xffffu) | ((unsigned)val << 16);
    if (__hip_atomic_compare_exchange_strong(w, &old, nw, TB_RLX, TB_RLX, TB_WG)) break;
  }
}
// COMPACT8: narrow one byte of a narrow integer (`sh`: bit position of the byte in the word; the word holds two variables)
__device__ __forceinline__ void cas_raise_lb8(unsigned* w, int sh, int rel) {
  const unsigned r = (unsigned)(rel < 0 ? 0 : (rel > 255 ? 255 : rel));  // (beyond the root domain: the domain is empty and the caller has raised the failure flag)
  unsigned old = __hip_atomic_load(w, TB_RLX, TB_WG);
  while (((old >> sh) & 0xffu) < r) {
    const unsigned nw = (old & ~(0xffu << sh)) | (r << sh);
    if (__hip_atomic_compare_exchange_strong(w, &old, nw, TB_RLX, TB_RLX, TB_WG)) break;
  }
}
__device__ __forceinline__ void cas_lower_ub8(unsigned* w, int sh, int rel) {
  const unsigned r = (unsigned)(rel < 0 ? 0 : (rel > 255 ? 255 : rel));
  unsigned old = __hip_atomic_load(w, TB_RLX, TB_WG);
  while (((old >> sh) & 0xffu) > r) {
    const unsigned nw = (old & ~(0xffu << sh)) | (r << sh);
    if (__hip_atomic_compare_exchange_strong(w, &old, nw, TB_RLX, TB_RLX, TB_WG)) break;
  }
}
// COMPACT8: an integer variable through its reference (load_int8)
__device__ __forceinline__ void raise_int_lb8(int2* store, int nw, int f, int val) {
  const int v = f & 0xffff, h = v + nw;
  unsigned* const words = reinterpret_cast<unsigned*>(store);
  if (v < nw) { cas_raise_lb16(words + TB_IDX(8, v, store_words), val); return; }
  cas_raise_lb8(words + TB_IDX(8, h >> 1, store_words), (h & 1) * 16, val - (((f >> 16) & 0x7fff) - C8_BASE_BIAS));
}
__device__ __forceinline__ void lower_int_ub8(int2* store, int nw, int f, int val) {
  const int v = f & 0xffff, h = v + nw;
  unsigned* const words = reinterpret_cast<unsigned*>(store);
  if (v < nw) { cas_lower_ub16(words + TB_IDX(8, v, store_words), val); return; }
  cas_lower_ub8(words + TB_IDX(8, h >> 1, store_words), (h & 1) * 16 + 8, val - (((f >> 16) & 0x7fff) - C8_BASE_BIAS));
}
template <int C>
__device__ __forceinline__ void raise_int_lb(int2* store, int v, int val) {
  if (C == 2) { cas_raise_lb16(reinterpret_cast<unsigned*>(store) + TB_IDX(7, v, store_words), val); return; }
  if (C == 3) v = TB_IDX(7, v, slab_vars); else TB_CHECK_ITV(7, v);
  if (C == 5) { (void)__hip_atomic_fetch_max(&store[v].x, val, TB_RLX, TB_AGENT); return; }
  if (C == 3) { (void)__hip_atomic_fetch_max(&hot_or_cold(store + v, v)->x, val, TB_RLX, TB_WG); return; }
  (void)__hip_atomic_fetch_max(&store[v].x, val, TB_RLX, TB_WG);
}
template <int C>
__device__ __forceinline__ void lower_int_ub(int2* store, int v, int val) {
  if (C == 2) { cas_lower_ub16(reinterpret_cast<unsigned*>(store) + TB_IDX(7, v, store_words), val); return; }
  if (C == 3) v = TB_IDX(7, v, slab_vars); else TB_CHECK_ITV(7, v);
  if (C == 5) { (void)__hip_atomic_fetch_min(&store[v].y, val, TB_RLX, TB_AGENT); return; }
  if (C == 3) { (void)__hip_atomic_fetch_min(&hot_or_cold(store + v, v)->y, val, TB_RLX, TB_WG); return; }
  (void)__hip_atomic_fetch_min(&store[v].y, val, TB_RLX, TB_WG);
}
// an integer variable through a reference, in every layout (`ni`: DevProblem::n_int)
template <int C>
__device__ __forceinline__ Itv load_ivar(const int2* store, int ni, int f) {
  if (C == 4) return load_int8(store, ni_wide(ni), f);
  return load_int<C>(store, f);
}
template <int C>
__device__ __forceinline__ void raise_ivar_lb(int2* store, int ni, int f, int val) {
  if (C == 4) { raise_int_lb8(store, ni_wide(ni), f, val); return; }
  raise_int_lb<C>(store, f, val);
}
template <int C>
__device__ __forceinline__ void lower_ivar_ub(int2* store, int ni, int f, int val) {
  if (C == 4) { lower_int_ub8(store, ni_wide(ni), f, val); return; }
  lower_int_ub<C>(store, f, val);
}
template <int C>
__device__ __forceinline__ void raise_lb(int2* store, int ni, int v, int val) {
  if (C == 3) { raise_int_lb<3>(store, v, val); return; }
  if (C == 5) { raise_int_lb<5>(store, v, val); return; }
  if (C && v < 0) return;  // a constant kept out of the slab (load_dom)
  if (C && var_of<C>(v) >= ni_int<C>(ni)) {
    const int b = var_of<C>(v) - ni_int<C>(ni);
    if (val >= 1) (void)__hip_atomic_fetch_or(reinterpret_cast<unsigned*>(store) + TB_IDX(9, bool_word0<C>(ni) + (b >> 4), store_words), 1u << ((b & 15) * 2), TB_RLX, TB_WG);
    return;
  }
  raise_ivar_lb<C>(store, ni, v, val);
}
template <int C>
__device__ __forceinline__ void lower_ub(int2* store, int ni, int v, int val) {
  if (C == 3) { lower_int_ub<3>(store, v, val); return; }
  if (C == 5) { lower_int_ub<5>(store, v, val); return; }
  if (C && v < 0) return;
  if (C && var_of<C>(v) >= ni_int<C>(ni)) {
    const int b = var_of<C>(v) - ni_int<C>(ni);
    if (val <= 0) (void)__hip_atomic_fetch_or(reinterpret_cast<unsigned*>(store) + TB_IDX(9, bool_word0<C>(ni) + (b >> 4), store_words), 2u << ((b & 15) * 2), TB_RLX, TB_WG);
    return;
  }
  lower_ivar_ub<C>(store, ni, v, val);
}

// COMPACT8: both bounds of one operand in ONE compare-and-swap loop -- two LDS round trips per narrowed variable where raise_lb + lower_ub take four.
// (Measured and dropped: starting the swap from the word the pass has read -- one trip -- keeps three more words live per lane and switches on the operand
//  kind: 84 -> 128 B of scratch at 80 VGPRs, trains15 4.34e7 -> 3.67e7 nodes/s.)  `cl` / `cu`: which bounds to move.
__device__ __forceinline__ void narrow_var8(int2* store, int ni, int f, int nl, int nu, bool cl, bool cu) {
  if (f < 0 || !(cl | cu)) return;  // a constant kept out of the slab
  const int v = f & 0xffff, n_i = ni & 0xffff, nw = ni_wide(ni);
  unsigned* const words = reinterpret_cast<unsigned*>(store);
  if (v >= n_i) {
    const int b = v - n_i;
    const unsigned bits = ((cl && nl >= 1) ? 1u : 0u) | ((cu && nu <= 0) ? 2u : 0u);
    if (bits) (void)__hip_atomic_fetch_or(words + TB_IDX(9, bool_word0<4>(ni) + (b >> 4), store_words), bits << ((b & 15) * 2), TB_RLX, TB_WG);
    return;
  }
  (void)TB_IDX(8, v < nw ? v : ((v + nw) >> 1), store_words);
  unsigned old = __hip_atomic_load(words + (v < nw ? v : ((v + nw) >> 1)), TB_RLX, TB_WG);
  if (v < nw) {
    const int tl = nl > 32767 ? 32767 : nl, tu = nu < -32768 ? -32768 : nu;
    for (;;) {
      const int cur_l = (int)(short)(old & 0xffffu), cur_u = (int)old >> 16;
      const int l = (cl && tl > cur_l) ? tl : cur_l, u = (cu && tu < cur_u) ? tu : cur_u;
      if (l == cur_l && u == cur_u) break;
      if (__hip_atomic_compare_exchange_strong(words + v, &old, ((unsigned)l & 0xffffu) | ((unsigned)u << 16), TB_RLX, TB_RLX, TB_WG)) break;
    }
    return;
  }
  const int h = v + nw, sh = (h & 1) * 16, base = ((f >> 16) & 0x7fff) - C8_BASE_BIAS;
  const int rl = nl - base, ru = nu - base;
  const unsigned tl = (unsigned)(rl < 0 ? 0 : (rl > 255 ? 255 : rl)), tu = (unsigned)(ru < 0 ? 0 : (ru > 255 ? 255 : ru));  // (beyond the root domain: empty, the caller has raised the flag)
  for (;;) {
    const unsigned cur_l = (old >> sh) & 0xffu, cur_u = (old >> (sh + 8)) & 0xffu;
    const unsigned l = (cl && tl > cur_l) ? tl : cur_l, u = (cu && tu < cur_u) ? tu : cur_u;
    if (l == cur_l && u == cur_u) break;
    if (__hip_atomic_compare_exchange_strong(words + (h >> 1), &old, (old & ~(0xffffu << sh)) | (l << sh) | (u << (sh + 8)), TB_RLX, TB_RLX, TB_WG)) break;
  }
}

// Per-thread counters kept in registers for the whole kernel and reduced once at the end.
struct ThreadCounters {
  unsigned writes = 0;  // narrowed bounds written by this lane since the last flush (flush_writes)
};
// The propagation counters of a workgroup live in LDS (BlockShared::bs): a wave adds its evaluations with one DS atomic per
// fixpoint call (lane 0), the per-lane write counters are folded in before they can wrap -- no 64-bit VALU arithmetic per run.
__device__ __forceinline__ void add_deductions(BlockShared& sh, unsigned long long n) {
  (void)__hip_atomic_fetch_add(&sh.bs.num_deductions, n, TB_RLX, TB_WG);
}
__device__ __forceinline__ void add_active(BlockShared& sh, unsigned long long n) {
  (void)__hip_atomic_fetch_add(&sh.bs.active_evals, n, TB_RLX, TB_WG);
}
__device__ __forceinline__ void flush_writes(BlockShared& sh, ThreadCounters& tc, bool force) {
  if (force || wave_any(tc.writes > (1u << 25))) {  // (64 lanes x 2^25 still fits the 32-bit wave sum)
    unsigned w = tc.writes;
    for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off, 64);
    if ((threadIdx.x & 63) == 0) (void)__hip_atomic_fetch_add(&sh.bs.store_writes, (unsigned long long)w, TB_RLX, TB_WG);
    tc.writes = 0;
  }
}

// One propagator application: load 3 domains, evaluate, write the narrowed bounds.
// `un` is only meaningful when the sweep it belongs to changed nothing.
// `act` = the lane holds a propagator (the whole wave calls this: the class test and the write tail are
// wave-uniform branches, so a sweep over propagators that change nothing -- the common case near the
// fixpoint -- executes no predicated store at all).
// Record of an idle lane (slice tail): the cheapest class, three immediates -> no memory access, entailed.
__device__ __forceinline__ int4 idle_record() { return make_int4(K_LEQ_T, 0, 0, 0); }

// Event mode bookkeeping of callers outside a fixpoint (decisions, objective bound, replay): the changed
// variable goes to a small list in LDS that the next fixpoint expands into the dirty bitmap.
struct ChangeList {
  int* list;
  int* count;
  int cap;
};
__device__ __forceinline__ void append_change(const ChangeList& cl, int v) {
  const int pos = __hip_atomic_fetch_add(cl.count, 1, TB_RLX, TB_WG);
  if (pos < cl.cap) cl.list[pos] = v;  // an overflowing list degrades to "run every slice"
}

// The three operand domains of a record, gathered ahead of its evaluation (the sweeps over a store in global memory: `fixpoint`, TB_PREGATHER).
struct Operands { Itv X, Y, Z; };
template <int C>
__device__ __forceinline__ Operands gather_operands(int2* store, const int ni, const int4 pr) {
  return Operands{load_dom<C>(store, ni, pr.y), load_dom<C>(store, ni, pr.z), load_dom<C>(store, ni, pr.w)};
}

template <bool EVENT, int C, bool PRE = false>
__device__ __forceinline__ void apply(const int4 pr, const bool act, int2* store, const int ni, int* bot, bool& changed, bool& un, ThreadCounters& tc, const int dbg = 0,
                                      int* narrowed = nullptr, const Operands* pre = nullptr) {
  const int w0 = pr.x;
  if (dbg == 0 && ((__builtin_amdgcn_readfirstlane(w0) >> 16) & CLASS_SET_MASK) == (1 << K_LEQ_T)) {
    // Class-pure slice of `y <= z` (x is the constant true; two thirds of wordpress7_500 after sorting the records by
    // class): two gathers instead of three, no candidate bookkeeping for x, one comparison per bound.
    const Itv Y = PRE ? pre->Y : load_dom<C>(store, ni, pr.z), Z = PRE ? pre->Z : load_dom<C>(store, ni, pr.w);
    const bool ny = Z.ub < Y.ub, nz = Y.lb > Z.lb;                    // y.ub := z.ub, z.lb := y.lb
    const bool empty_in = (Y.lb > Y.ub) | (Z.lb > Z.ub);
    const bool touched = act & (ny | nz | empty_in);
    if (wave_any(touched)) {
      if (touched) {
        if (empty_in | (Y.lb > Z.ub)) st(bot, 1);                     // y.lb > new y.ub, or z.ub < new z.lb
        if (!empty_in) {
          if (ny) lower_ub<C>(store, ni, pr.z, Z.ub);
          if (nz) raise_lb<C>(store, ni, pr.w, Y.lb);
          tc.writes += (unsigned)ny + (unsigned)nz;
          changed = true;
          if (EVENT) *narrowed = ((int)ny << 3) | ((int)nz << 4);  // y.ub lowered, z.lb raised
        }
      }
    }
    un |= act & !(Y.ub <= Z.lb);
    return;
  }
  // three gathers issued back to back, one s_waitcnt (the LDS is ~1 % busy: gathers are cheap, VALU is not)
  Itv X{0, 1}, Y{0, 1}, Z{0, 1};
  if (PRE) { X = pre->X; Y = pre->Y; Z = pre->Z; }
  else if (!(dbg & 2)) { X = load_dom<C>(store, ni, pr.y); Y = load_dom<C>(store, ni, pr.z); Z = load_dom<C>(store, ni, pr.w); }
  Cand c;
  if (!(dbg & 1)) c = evaluate_packed<((C == 3 || C == 5) ? 1 : 0)>(w0, X, Y, Z); else c.ent = (pr.y != 0x7fffffff);
  if (dbg & 4) { un |= act & !c.ent; return; }
  const bool empty_in = (X.lb > X.ub) | (Y.lb > Y.ub) | (Z.lb > Z.ub);
  const bool cx = (c.xl > X.lb) | (c.xu < X.ub), cy = (c.yl > Y.lb) | (c.yu < Y.ub), cz = (c.zl > Z.lb) | (c.zu < Z.ub);
  const bool touched = act & (cx | cy | cz | empty_in);
  if (wave_any(touched)) {  // wave-uniform: rare once the sweep is close to the fixpoint
    if (touched) {
      const int nxl = imax(c.xl, X.lb), nxu = imin(c.xu, X.ub);
      const int nyl = imax(c.yl, Y.lb), nyu = imin(c.yu, Y.ub);
      const int nzl = imax(c.zl, Z.lb), nzu = imin(c.zu, Z.ub);
      if (empty_in | (nxl > nxu) | (nyl > nyu) | (nzl > nzu)) st(bot, 1);
      if (!empty_in) {
        int k = 0;
        if (nxl != X.lb) { raise_lb<C>(store, ni, pr.y, nxl); ++k; }
        if (nxu != X.ub) { lower_ub<C>(store, ni, pr.y, nxu); ++k; }
        if (nyl != Y.lb) { raise_lb<C>(store, ni, pr.z, nyl); ++k; }
        if (nyu != Y.ub) { lower_ub<C>(store, ni, pr.z, nyu); ++k; }
        if (nzl != Z.lb) { raise_lb<C>(store, ni, pr.w, nzl); ++k; }
        if (nzu != Z.ub) { lower_ub<C>(store, ni, pr.w, nzu); ++k; }
        tc.writes += (unsigned)k;
        changed |= (k != 0);
        if (EVENT) *narrowed = (int)(nxl != X.lb) | ((int)(nxu != X.ub) << 1) | ((int)(nyl != Y.lb) << 2) | ((int)(nyu != Y.ub) << 3) | ((int)(nzl != Z.lb) << 4) | ((int)(nzu != Z.ub) << 5);
      }
    }
  }
  un |= act & !c.ent;
}

// Block-parallel fixpoint of all propagators + entailment test, fused.
// (BlockAsynchronousFixpointGPU::fixpoint + warp_fixpoint + the `ask` scan: barebones:920-982.)
// Precondition: the store is ready and a barrier separates its last write from this call.
// Returns the number of block-level sweeps.  After the call (which ends with a barrier):
//   sh.bot          the node failed
//   *all_entailed   no propagator is un-entailed (only meaningful when !sh.bot)
// `slice_unent`: one byte per 64-propagator slice behind the store ("some propagator of the slice is not entailed").
// With RM (tb_config.entailed_prop_removal) a slice whose byte is 0 is skipped for the whole subtree -- entailed-propagator removal
// (FixpointSubsetGPU::select, gpu_dive_and_solve.hpp:334, barebones:984; a build option of the reference, off by
// default) at the granularity of a wave's slice instead of a compacted index array.
template <bool RM, int C, bool DEEP = false>
__device__ __forceinline__ int fixpoint(const DevProblem& P, BlockShared& sh, int2* store, const int4* props,
                                        unsigned char* slice_unent, ThreadCounters& tc, bool& all_entailed) {
  const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63;
  const int n = P.n_props;
  const bool wac1 = P.fixpoint == 1 && n > P.wac1_threshold;
  constexpr bool rm = RM;  // compiled in or out: the headline sweep pays nothing for the option
  const int dbg = knobs(P) & 0xff, force_sweeps = (knobs(P) >> 8) & 0xff;  // profiling knobs, 0 in production
  if (tid == 0) { st(&sh.flag[0], 0); st(&sh.unent[0], 0); if (TB_TEAM_DYNAMIC && (C == 5 || (TB_DYNAMIC_DEEP && DEEP))) { st(&sh.chg_count[0], 0); st(&sh.chg_count[1], 0); } }
  __syncthreads();
  int it = 0, k = 0;
  unsigned wave_evals = 0;  // wave-uniform: slice evaluations of this wave (x 64 = deduce calls, barebones:958-960)
  unsigned long long wave_active = 0;  // ... and the propagators they held (the network's last slice is partly filled)
  for (;;) {
    k = it % 3;
    bool changed = false, un = false;
    // The bytecode of the next 64-propagator slice is fetched while the current one is evaluated (software prefetch: the 16-B records come from L2 unless
    // they were staged in LDS) -- and, in the WAC1 sweeps of the 1024-thread kernels (DEEP), those of the next THREE slices of the wave: an evaluation issues
    // ~40 VALU + ~50 SALU, an L2 round trip under load is several times that, and with the 4 waves per SIMD of those kernels a wave waited for its record
    // (wordpress7_500: VALU 46 % busy, 54 % of the wave-cycles parked).  r04, same box: wordpress7_500 wac1 1.97e6 -> 2.06e6 nodes/s, the synthetic 100k x 500k network
    // 7.99e10 -> 8.24e10 propagations/s.  Not for the plain AC1 sweeps (-3 %) nor for the kernels of smaller workgroups, where the 18 more VGPRs cost waves
    // (accap_a3 -8 %).  The loop is unrolled three times so that the three records need no rotation.
    // (the record array is padded to whole slices with idle records, and the prefetch index is clamped to the last
    //  slice instead of being predicated: an unconditional load lets the wait sink to the first use)
    const int last_base = ((n - 1) >> 6) << 6;
    // (layout 5, workgroup teams: member m of M takes the slices m, m + M, ... of every wave's share -- TS is the stride of the whole team)
    const int TS = C == 5 ? T * team_size(sh) : T;
    const int member_base = C == 5 ? T * team_member(sh) : 0;
    const int wave_base = __builtin_amdgcn_readfirstlane(tid - lane) + member_base;  // wave-uniform: slice addressing stays in SGPRs
    // Operands gathered one slice ahead (PREG; r05): with the store in global memory (hot tier, workgroup teams) a slice is a dependent chain record -> three gathers ->
    // compare -> atomics, the gathers an L2 round trip each time, and a CU has its 16 waves and no more to hide it behind.  The next slice's operands are requested before
    // the current slice is evaluated.  What they read may be older than what the current slice is about to write: a wider domain, from which the rules derive a weaker but
    // valid narrowing (the store only shrinks, and the writes are atomic min / max) and possibly a `changed` that changes nothing; the sweep that ends the fixpoint wrote
    // nothing, so everything it read was current, and its entailment flags are exact.  Same fixpoint, same tree.
    // Measured r05 (profiles/r05_pregather_ab.txt, same box): teams, plain sweeps 1.115e11 -> 1.225e11 propagations/s and 1.296e4 -> 1.364e4 nodes/s; not the WAC1 sweeps (a
    // wave's first local pass on older operands costs more passes than the overlap saves: teams -3 % nodes/s, hot tier -7 %), not the hot tier's plain sweeps (-1 %).
    constexpr bool PREG = TB_PREGATHER && !RM && C == 5;
    // one slice of a plain (AC1) sweep
    auto ac1_step = [&](const int4 pr, const int base, const Operands* g = nullptr) {
      const int i = base + lane;
      const bool act = i < n;
      if (PREG) { apply<false, C, PREG>(pr, act, store, P.n_int, &sh.bot, changed, un, tc, 0, nullptr, g); return; }
      if (rm) {
        if (slice_unent[base >> 6] == 0) return;
        bool ch = false, un_i = false;
        apply<false, C>(pr, act, store, P.n_int, &sh.bot, ch, un_i, tc, dbg);
        ++wave_evals;  // iterations x active propagators (gpu_dive_and_solve.hpp:304-306)
        wave_active += (unsigned)imin(64, n - base);
        changed |= ch; un |= un_i;
        if (!wave_any(ch) && !wave_any(un_i) && lane == 0) slice_unent[base >> 6] = 0;
        return;
      }
      apply<false, C>(pr, act, store, P.n_int, &sh.bot, changed, un, tc, dbg);
    };
    // WAC1: a wave iterates its 64 propagators to a local fixpoint before moving on (config.cpp:26,
    // warp_fixpoint at barebones:955; the wave is 64 wide on CDNA).
    auto wac1_step = [&](const int4 pr, const int base) {
      const int i = base + lane;
      const bool act = i < n;
      if (rm && slice_unent[base >> 6] == 0) return;
      for (unsigned local_iters = 1;; ++local_iters) {
        bool ch = false, un_i = false;
        apply<false, C>(pr, act, store, P.n_int, &sh.bot, ch, un_i, tc, dbg);
        ++wave_evals;
        wave_active += (unsigned)imin(64, n - base);
        if (!wave_any(ch)) {
          un |= un_i;
          if (rm && !wave_any(un_i) && lane == 0) slice_unent[base >> 6] = 0;  // 1 -> 0 only: entailment is monotone below a node
          break;
        }
        changed = true;
#if TB_TEAM_LOCAL_FENCE
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#else
        // (a team's store is read and narrowed with agent-scope atomics: the next pass may read it before this pass's narrowings have landed -- an older, wider domain, which
        //  is sound (PREG above) -- instead of waiting a round trip to the L2 for their acknowledgement)
        if (C != 5) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#endif
        if (ld(&sh.bot)) break;
        // watchdog inside the wave-local loop: a slowly converging pair in one slice (x < y < x over 2^31 values) never
        // reaches the block-level check below (wave-uniform counter: scalar work, once per 1024 iterations)
        if ((local_iters % WAVE_WATCHDOG_PERIOD) == 0) {
          if (lane == 0 && deadline_passed(P)) st(&sh.abort, 1);
          if (ld(&sh.abort)) break;
        }
      }
    };
    // Slices handed out on demand (DYN; r05, teams): the waves of a workgroup take the workgroup's slices of the sweep -- the same ones, in the same order -- from a counter in
    // LDS instead of every 16th each.  A sweep ends when its slowest wave does, and slices differ: a WAC1 wave iterates where something moves, a class-pure slice of sums
    // costs several times one of `<=`, a product may divide.  A wave keeps the records of its next slices in flight as before, so it holds two or three slices it cannot
    // give away: that is the grain of the balance.
    // Measured r05 (profiles/r05_dyn_ab.txt, same box, synthetic 100k x 500k in teams): WAC1 sweeps 1.196e11 -> 1.324e11 propagations/s, 1.955e4 -> 2.214e4 nodes/s.  Not the
    // plain sweeps: the rate rises there too (1.535e11 -> 1.577e11) but the order in which a sweep meets the slices is no longer the same from sweep to sweep, the fixpoint
    // takes 16 % more evaluations, and the nodes per second fall (2.10e4 -> 1.87e4): they keep every 16th slice per wave.
    constexpr bool DYN = TB_TEAM_DYNAMIC && (C == 5 || (TB_DYNAMIC_DEEP && DEEP));
    // The LDS-resident 1024-thread WAC1 sweeps take it when a wave has enough slices for the balance to outweigh the three slices every wave takes beyond the end
    // (same box, -fp wac1: wordpress7_500, 45 slices per wave, 2.07e6 -> 2.18e6 nodes/s; trains15, 10 per wave, 6.36e6 -> 6.28e6; accap_a3 5.44e7 -> 5.37e7).
    const bool dyn = DYN && wac1 && T == 1024 && (C == 5 || n >= DYN_MIN_PROPS);
    int* const next_step = &sh.chg_count[it & 1];  // (the event fixpoint's change-list fill: unused by the sweeps; two counters, the idle one is reset during the sweep)
    auto take = [&]() -> int {  // base of the next slice of this workgroup, wave-uniform (>= n: none left)
      int v = 0;
      if (lane == 0) v = __hip_atomic_fetch_add(next_step, 1, TB_RLX, TB_WG);
      v = __builtin_amdgcn_readfirstlane(v);
      return (v >> 4) * TS + member_base + (v & 15) * 64;  // (16 waves per workgroup: the team plan is 1024 threads, engine.hip: team_mode)
    };
    if (PREG && !wac1 && n > 0) {
      int4 pr_next = props[imin(wave_base, last_base) + lane], pr_next2 = props[imin(wave_base + TS, last_base) + lane];
      Operands g_next = gather_operands<C>(store, P.n_int, pr_next);
      for (int base = wave_base; base < n; base += TS) {
        const int4 pr = pr_next;
        const Operands g = g_next;
        pr_next = pr_next2;
        g_next = gather_operands<C>(store, P.n_int, pr_next);
        pr_next2 = props[imin(base + 2 * TS, last_base) + lane];
        ac1_step(pr, base, &g);
      }
    } else if (!wac1) {
      int4 pr_next = idle_record();
      if (n > 0) pr_next = props[imin(wave_base, last_base) + lane];
      for (int base = wave_base; base < n; base += TS) {
        const int4 pr = pr_next;
        pr_next = props[imin(base + TS, last_base) + lane];
        ac1_step(pr, base);
      }
    } else if (!DEEP) {
      int4 pr_next = props[imin(wave_base, last_base) + lane];
      for (int base = wave_base; base < n; base += TS) {
        const int4 pr = pr_next;
        pr_next = props[imin(base + TS, last_base) + lane];
        wac1_step(pr, base);
      }
    } else {
      // (one loop for both ways of dealing the slices -- every 16th per wave, or on demand: as a second loop the hand-out cost the plain sweeps of the same kernel 9 % through
      //  its register allocation, wordpress7_500 -fp ac1 6.34e11 -> 5.74e11 propagations/s)
      int b0 = dyn ? take() : wave_base, b1 = dyn ? take() : wave_base + TS, b2 = dyn ? take() : wave_base + 2 * TS;
      int4 q0 = props[imin(b0, last_base) + lane], q1 = props[imin(b1, last_base) + lane], q2 = props[imin(b2, last_base) + lane];  // (n > 0: WAC1 runs above its threshold)
      for (;;) {
        if (b0 >= n) break;
        { const int4 pr = q0; const int base = b0; b0 = dyn ? take() : b0 + 3 * TS; q0 = props[imin(b0, last_base) + lane]; wac1_step(pr, base); }
        if (b1 >= n) break;
        { const int4 pr = q1; const int base = b1; b1 = dyn ? take() : b1 + 3 * TS; q1 = props[imin(b1, last_base) + lane]; wac1_step(pr, base); }
        if (b2 >= n) break;
        { const int4 pr = q2; const int base = b2; b2 = dyn ? take() : b2 + 3 * TS; q2 = props[imin(b2, last_base) + lane]; wac1_step(pr, base); }
      }
    }
    const bool any_changed = wave_any(changed), any_un = wave_any(un);
    if (lane == 0) {
      if (any_changed) st(&sh.flag[k], 1);
      if (any_un) st(&sh.unent[k], 1);
    }
    if (tid == 0) {
      const int k1 = (k + 1) % 3;
      st(&sh.flag[k1], 0); st(&sh.unent[k1], 0);
      if (TB_TEAM_DYNAMIC && (C == 5 || (TB_DYNAMIC_DEEP && DEEP))) st(&sh.chg_count[(it + 1) & 1], 0);  // (nobody touches the next sweep's counter before the barrier below)
      // watchdog: a pathological network (x < y < x over 2^31 values) must not outlive the deadline
      if ((it & 255) == 255 && deadline_passed(P)) st(&sh.abort, 1);
    }
    __syncthreads();
    if constexpr (C == 5) {
      // one team barrier per sweep: did ANY member change something / see an un-entailed propagator / fail / run out of time?
      const unsigned mine = (ld(&sh.flag[k]) ? TEAM_CHANGED : 0u) | (ld(&sh.unent[k]) ? TEAM_UNENT : 0u) | (ld(&sh.bot) ? TEAM_BOT : 0u) | (ld(&sh.abort) ? TEAM_ABORT : 0u);
      const unsigned all = team_sync(P, sh, mine);
      if (tid == 0) {
        st(&sh.flag[k], (all & TEAM_CHANGED) ? 1 : 0); st(&sh.unent[k], (all & TEAM_UNENT) ? 1 : 0);
        if (all & TEAM_BOT) st(&sh.bot, 1);
        if (all & TEAM_ABORT) st(&sh.abort, 1);
      }
      __syncthreads();
    }
    ++it;
    if (force_sweeps) { if (it >= force_sweeps) break; else continue; }
    if (!ld(&sh.flag[k]) || dead_node(sh)) break;
  }
  if (!wac1 && !rm && tid == 0 && (C != 5 || team_member(sh) == 0)) { add_deductions(sh, (unsigned long long)it * (unsigned long long)n); add_active(sh, (unsigned long long)it * (unsigned long long)n); }  // barebones:934 (a team: its leader counts)
  if (lane == 0 && wave_evals != 0) { add_deductions(sh, 64ull * wave_evals); add_active(sh, wave_active); }
  all_entailed = !ld(&sh.unent[k]);
  return it;
}

// Event-driven WAC1 (tb_config.fixpoint = 2): state shared by its functions.  Same fixpoint, same wave-local
// iteration as `fixpoint`, but only the 64-propagator slices that read a narrowed variable are evaluated (one
// dirty bitmap in LDS, filled through the variable -> slices adjacency on every narrowing; see fixpoint_event).
// A node starts from the slices of the variables its caller changed, because backtracking restores the
// parent's PROPAGATED store from the HBM snapshot stack.  Entailment is kept as one byte per slice, stored
// right behind the store so that snapshots carry it: a slice that did not run has not changed status.
// This is the role of FixpointSubsetGPU / entailed-propagator removal in the reference (gpu_dive_and_solve.hpp:334,
// barebones:984, off by default there), re-thought for wave64 slices instead of a compacted index array.
struct EventState {
  unsigned* dirty;        // LDS: bitmap of the slices that must (re)run
  int* list;              // LDS: variables changed outside the fixpoint (decision, bound, replay)
  const int4* succ;       // successor records: DevProblem::succ, or their copy in LDS (TCN_SHARED)
  unsigned char* unent;   // behind the store (LDS or HBM slab): one byte per slice for the sweeps (entailed-slice removal), one BIT per slice (32-bit words) for the event fixpoint
  int words, cap;
  const unsigned* own;    // workgroup teams (layout 5, fixpoint_event_team): LDS table [waves of the workgroup][words], bit s & 31 of word s >> 5 = slice s is this wave's
};

// `ev`: EV_LB / EV_UB bits (what happened to v); stored in the two top bits of the entry
__device__ __forceinline__ void note_change(BlockShared& sh, const EventState& es, int v, int ev) {
  const ChangeList cl{es.list, &sh.chg_count[0], es.cap};
  append_change(cl, v | (ev << 30));
}

// (SC: the scope of the bitmap -- a workgroup's LDS, or, for the event fixpoint of a workgroup team (r06), the team's bitmap in global memory: agent scope)
template <int SC = TB_WG>
__device__ __forceinline__ void mark_slice(unsigned* dirty, int t) {
  t = TB_IDX(10, t, n_slices);
  (void)__hip_atomic_fetch_or(&dirty[t >> 5], 1u << (t & 31), TB_RLX, SC);
}

// Events a variable can undergo: EV_LB its lower bound was raised, EV_UB its upper bound was lowered.  A reader slice is woken
// up only by the events it declared an interest in (engine.hip: interest_of): `b1 <= b2` does not care that b1 became false.
constexpr int EV_LB = 1, EV_UB = 2;

// Up to two successor slices packed with the record (16-bit ids, 0xffff = none; `interest`: 2 bits each).  True when something was marked.
template <int SC = TB_WG>
__device__ __forceinline__ bool mark_packed(unsigned* dirty, unsigned packed, int interest, int ev) {
  const unsigned s0 = packed & 0xffffu, s1 = packed >> 16;
  const bool m0 = s0 != 0xffffu && (interest & ev) != 0, m1 = s1 != 0xffffu && ((interest >> 2) & ev) != 0;
  if (m0) mark_slice<SC>(dirty, (int)s0);
  if (m1) mark_slice<SC>(dirty, (int)s1);
  return m0 | m1;
}

// Mark the slices reading variable v that are interested in the events `ev` (0: nothing to do for this lane), except `self`,
// from the variable's 32-byte adjacency record (DevProblem::var_adj: halfword 0 = number of reader slices, halfwords 1-11 =
// the first eleven, word 6 = their interests, word 7 = offset of the others in DevProblem::adj_rest).  One L2 round trip
// whatever the degree up to 11; returns true when the list is longer (mark_tail).
template <int SC = TB_WG>
__device__ __forceinline__ bool mark_var(const DevProblem& P, unsigned* dirty, int v, int self, int ev, int& deg_out, int& off_out, bool& did) {
  int4 a = make_int4(0, 0, 0, 0), b = a;
  if (ev) { v = TB_IDX(11, v, adj_vars); a = glob(P.var_adj)[2 * (size_t)v]; b = glob(P.var_adj)[2 * (size_t)v + 1]; }
  const unsigned w[8] = {(unsigned)a.x, (unsigned)a.y, (unsigned)a.z, (unsigned)a.w, (unsigned)b.x, (unsigned)b.y, (unsigned)b.z, (unsigned)b.w};
  const int deg = (int)(w[0] & 0xffffu);
#pragma unroll
  for (int j = 0; j < 11; ++j) {
    if (!wave_any(j < deg)) break;  // wave-uniform: most variables have four or five readers, not eleven
    const int hw = j + 1;
    const int t = (int)((hw & 1) ? (w[hw >> 1] >> 16) : (w[hw >> 1] & 0xffffu));
    if (j < deg && t != self && ((w[6] >> (2 * j)) & (unsigned)ev)) { mark_slice<SC>(dirty, t); did = true; }
  }
  deg_out = deg;
  off_out = (int)w[7];
  return deg > 11;
}
// the tail of the lists longer than 11, cooperatively: one lane at a time is broadcast, the 64 lanes stride over its list
template <int SC = TB_WG>
__device__ __forceinline__ bool mark_tail(const DevProblem& P, unsigned* dirty, unsigned long long mask, int deg, int off, int ev, int self) {
  const int lane = here(threadIdx.x) & 63;
  bool did = false;
  while (mask) {
    const int l = __builtin_ctzll(mask);
    mask &= mask - 1;
    const int d = __builtin_amdgcn_readlane(deg, l), o = __builtin_amdgcn_readlane(off, l), e = __builtin_amdgcn_readlane(ev, l);
    for (int j = lane; j < d - 11; j += 64) {
      const int t = glob(P.adj_rest)[TB_IDX(12, o + j, adj_rest)];
      if ((t & 0x3fffffff) != self && ((t >> 30) & e)) { mark_slice<SC>(dirty, t & 0x3fffffff); did = true; }
    }
  }
  return did;
}

// ---- event-driven fixpoint: one slice run ------------------------------------------------------------------------------
//
// A run iterates the 64 propagators of a slice to their local fixpoint.  `eval(ch, un, nar)` is one iteration: it
// narrows the store, reports whether this lane changed something (ch), whether its propagator is not entailed (un) and
// which bounds it narrowed (nar: bits 2k / 2k+1 = lower bound raised / upper bound lowered of operand k, 0 x, 1 y, 2 z).  Everything that does not change between two iterations -- LDS
// addresses, bit positions, the value of a constant operand -- is computed once per run by the caller of run_slice.
struct RunEnv {
  const DevProblem& P;
  BlockShared& sh;
  unsigned* nxt;            // dirty bitmap of the next round
  unsigned* unent;          // per-slice "some propagator is not entailed" bits
  int s;                    // the slice
};

template <class Eval>
__device__ __forceinline__ unsigned run_slice(const RunEnv& E, int& nar_all, Eval&& eval) {
  const DevProblem& P = E.P;
  const int lane = threadIdx.x & 63, s = E.s;
  unsigned wave_iters = 0;  // wave-uniform
  for (;;) {
    bool ch = false, un_i = false;
    int nar = 0;
    eval(ch, un_i, nar);
    ++wave_iters;
    nar_all |= nar;  // (a cooperative body reaches the slice's fixpoint in one pass: it narrows, reports nar and no change)
    if (!wave_any(ch)) {
      // The byte only ever goes 1 -> 0 below a node (entailment is monotone).
      if (!wave_any(un_i) && lane == 0) { const int sq = here_s(__builtin_amdgcn_readfirstlane(s)); (void)__hip_atomic_fetch_and(&E.unent[sq >> 5], ~(1u << (sq & 31)), TB_RLX, TB_WG); }  // (mask and word formed here: hoisted, they were spilled around every run)
      break;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    if (ld(&E.sh.bot)) break;
    if ((wave_iters % WAVE_WATCHDOG_PERIOD) == 0) {  // watchdog inside the wave-local loop (see fixpoint)
      if (lane == 0 && deadline_passed(P)) st(&E.sh.abort, 1);
      if (ld(&E.sh.abort)) break;
    }
  }
  if (pk(P) & 0x8) { bool ch = false, un_i = false; int nar = 0; eval(ch, un_i, nar); nar_all |= nar; }  // tuning: cost of one iteration
  return wave_iters;
}

// Successors of one run: the OTHER slices reading a variable the run narrowed, if they care about that kind of narrowing,
// run in the next round.  `nar_all`, per lane: bits 2k / 2k+1 = lower bound raised / upper bound lowered of operand k
// (0 x, 1 y, 2 z).  In a round based fixpoint nobody looks at the marks before the barrier, so they are issued once per
// run, not after each narrowing.  Up to two successors per operand travel with the record (DevProblem::succ) and need no
// memory access; the others come from the variable's 32-byte adjacency record, one L2 round trip for the whole wave.
// Returns true (wave-uniform) when something was marked.
template <int C>
__device__ __forceinline__ bool mark_successors(const DevProblem& P, BlockShared& sh, unsigned* nxt, int s, const int4 pr, const int4 sc, int nar_all, int* census = nullptr) {
  // word0 bits 4-9 (pack_props): bits 2k, 2k + 1 = operand k has a reader outside this slice -- a narrowing of an operand private to the slice wakes nobody.  (r05 kept three
  // "private" flags and rebuilt this mask per lane and run: 14 VALU of every run by the block counts of r06, a dozen more than the two it takes now.)
  const int e6 = nar_all & (pr.x >> 4) & 63;
  if (!wave_any(e6 != 0)) return false;
  int ex = e6 & 3, ey = (e6 >> 2) & 3, ez = (e6 >> 4) & 3;
  TB_REGION(39);
  bool did = false;
#ifdef TB_TUNING
  if (census != nullptr) {  // knob 0x10000: runs with something to mark [12], lanes with something to mark [15], ... through the adjacency records [13] / lanes [16], ... with a tail [14]
    const unsigned long long m0 = wave_ballot((ex | ey | ez) != 0);
    const unsigned long long m1 = wave_ballot((ex & ((sc.w & 1) | ((sc.w >> 15) & 2))) || (ey & (((sc.w >> 1) & 1) | ((sc.w >> 16) & 2))) || (ez & (((sc.w >> 2) & 1) | ((sc.w >> 17) & 2))));
    if ((threadIdx.x & 63) == 0) { census[12] += 1; census[15] += __builtin_popcountll(m0); if (m1) { census[13] += 1; census[16] += __builtin_popcountll(m1); } }
  }
#endif
  // events whose interested readers all sit in the record's two slots are served from there; the others walk the variable's record
  const int ovx = (sc.w & 1) | ((sc.w >> 15) & 2), ovy = ((sc.w >> 1) & 1) | ((sc.w >> 16) & 2), ovz = ((sc.w >> 2) & 1) | ((sc.w >> 17) & 2);
  constexpr int SC = C == 5 ? TB_AGENT : TB_WG;  // (layout 5: the bitmap of a workgroup team lives in global memory)
  if (ex & ~ovx) did |= mark_packed<SC>(nxt, (unsigned)sc.x, sc.w >> 4, ex & ~ovx);
  if (ey & ~ovy) did |= mark_packed<SC>(nxt, (unsigned)sc.y, sc.w >> 8, ey & ~ovy);
  if (ez & ~ovz) did |= mark_packed<SC>(nxt, (unsigned)sc.z, sc.w >> 12, ez & ~ovz);
  ex &= ovx; ey &= ovy; ez &= ovz;
  if (wave_any((ex | ey | ez) != 0)) {
    TB_REGION(40);
#ifdef TB_TUNING
    const long long t_var = census != nullptr ? clock64() : 0;  // [21]: time in this branch; [22], [23]: lanes marking their x / their y or z through it
    if (census != nullptr) {
      const unsigned long long lx = wave_ballot(ex != 0), lyz = wave_ballot((ey | ez) != 0);
      if ((threadIdx.x & 63) == 0) { census[22] += __builtin_popcountll(lx); census[23] += __builtin_popcountll(lyz); }
    }
#endif
    int dx = 0, dy = 0, dz = 0, ox = 0, oy = 0, oz = 0;
    const bool tx = mark_var<SC>(P, nxt, var_of<C>(pr.y), s, ex, dx, ox, did);
    const bool ty = mark_var<SC>(P, nxt, var_of<C>(pr.z), s, ey, dy, oy, did);
    const bool tz = mark_var<SC>(P, nxt, var_of<C>(pr.w), s, ez, dz, oz, did);
    const unsigned long long mx = wave_ballot(tx), my = wave_ballot(ty), mz = wave_ballot(tz);
    bool dt = false;
#ifdef TB_TUNING
    if (census != nullptr) {  // degree histogram of the variables marked through their adjacency record: <= 4 [17], 5-6 [18], 7-11 [19], more [20]
      if ((mx | my | mz) && (threadIdx.x & 63) == 0) census[14] += 1;
      const int degs[3] = {ex ? dx : 0, ey ? dy : 0, ez ? dz : 0};
      for (int q = 0; q < 3; ++q)
        if (degs[q] > 0) (void)__hip_atomic_fetch_add(&census[degs[q] <= 4 ? 17 : (degs[q] <= 6 ? 18 : (degs[q] <= 11 ? 19 : 20))], 1, TB_RLX, TB_WG);
    }
#endif
    if (mx) dt |= mark_tail<SC>(P, nxt, mx, dx, ox, ex, s);
    if (my) dt |= mark_tail<SC>(P, nxt, my, dy, oy, ey, s);
    if (mz) dt |= mark_tail<SC>(P, nxt, mz, dz, oz, ez, s);
    did |= dt;
#ifdef TB_TUNING
    if (census != nullptr) { const long long t_ = clock64(); if ((threadIdx.x & 63) == 0) census[21] += (int)((t_ - t_var) >> 4); }
#endif
  }
  TB_REGION(41);
  return wave_any(did);
}

// A 2-bit Boolean of the COMPACT layout as an LDS word address and a bit position (computed once per run).
struct BoolRef { unsigned* word; int shift; };
template <int C>
__device__ __forceinline__ BoolRef bool_ref(int2* store, int ni, int v, bool act) {
  const int b = act ? v - ni_int<C>(ni) : 0;  // idle lanes of a padded slice look at the first Boolean and touch nothing (a Boolean's reference is the variable in every layout)
  BoolRef r;
  r.word = reinterpret_cast<unsigned*>(store) + TB_IDX(22, bool_word0<C>(ni) + (b >> 4), store_words);
  r.shift = (b & 15) * 2;
  return r;
}
__device__ __forceinline__ unsigned bool_bits(const BoolRef r) { return (__hip_atomic_load(r.word, TB_RLX, TB_WG) >> r.shift) & 3u; }
__device__ __forceinline__ void bool_or(const BoolRef r, unsigned bits) { (void)__hip_atomic_fetch_or(r.word, bits << r.shift, TB_RLX, TB_WG); }

// ---- lean run of a class-pure slice ----------------------------------------------------------------------------------------------------
//
// The generic `apply` pays for generality at every pass: infinity cases and saturation in every sum, six candidate bounds
// whatever the class, an emptiness check of all three inputs, per-bound counters.  A slice whose 64 records have ONE class and whose
// operands all have finite root domains within +-2^29 (engine.hip: slice_infos, bit 9 of the info word -- domains only shrink, so
// this holds in the whole tree) needs none of that: plain 32-bit arithmetic cannot overflow, a class body touches only the bounds
// its rule can move, and an empty input needs no test -- whoever emptied a domain raised the failure flag when it did.  Same rules,
// same fixpoint as `evaluate_packed`; what changed is kept as per-lane event bits (six ballots per narrowing pass kept the scalar unit
// busier than it pays).  One pass is ~25 VALU instead of ~90.
// Returns the wave iterations; `nar_all`: bits 2k / 2k+1 = lower bound raised / upper bound lowered of operand k (0 x, 1 y, 2 z).
// COMPACT layout: `kinds` (2 bits per operand, word0 bits 26-31 of the slice: 1 all integer variables, 2 all 2-bit Booleans, else mixed) picks
// the cheapest way to read an operand -- a Boolean column is one ds_read_b32 and three VALU per pass, its word address and bit position
// computed once per run.
template <int C>
struct LeanOperand {
  int v;            // variable (idle lanes: 0)
  BoolRef b;        // COMPACT Boolean column: word and bit position
};
template <int C>
__device__ __forceinline__ LeanOperand<C> lean_operand(int2* store, int ni, int v, bool act, int kind) {
  LeanOperand<C> o;
  o.v = act ? v : 0;
  if (C && kind == 2) o.b = bool_ref<C>(store, ni, v, act);
  return o;
}
template <int C>
__device__ __forceinline__ Itv lean_load(int2* store, int ni, const LeanOperand<C>& o, int kind, unsigned* raw = nullptr) {
  if (C && kind == 2) { const unsigned bits = bool_bits(o.b); Itv d; d.lb = (int)(bits & 1u); d.ub = 1 - (int)(bits >> 1); return d; }
  if (C == 4) return kind != 1 ? load_dom<4>(store, ni, o.v, raw) : load_int8(store, ni_wide(ni), o.v, raw);
  if (C && kind != 1) return load_dom<C>(store, ni, o.v);
  return load_ivar<C>(store, ni, o.v);
}
// Pseudo-class of the lean run (never in a record): a slice of `x = y * z` whose operands are all non-negative and finite with products below 2^30 (engine.hip:
// slice_infos, bit 0x800 of the info word).  The generic heavy rule (propagators.hpp: evaluate_heavy) pays for every sign case: four 64-bit corner products, sixteen
// signed divisions with floor / ceil fix-ups and the TDIV / TMOD code beside it -- 1018 VALU per pass, and the 15 `Price_i = price_i * occupied_i` of wordpress7_500 sit
// on the objective's critical path (1.5 passes per node: 16 % of a node's VALU, r04 region budget).  On non-negative operands products and quotients are monotone:
// the hull of the corner products is [y.lb * z.lb, y.ub * z.ub], of the quotients [ceil(x.lb / z.ub), floor(x.ub / z.lb)] -- two unsigned multiplications and four
// unsigned divisions, same bounds as the generic rule on every store that is not already failed.
constexpr int K_MUL_NN = 10;
__device__ __forceinline__ unsigned umax1(int a) { return a > 1 ? (unsigned)a : 1u; }  // a divisor that may be garbage on an idle or failed lane

// CLS >= 0: the class is a compile-time constant (the pass loop then holds one class body and no chain of scalar compares); -1: `cls_dyn`.
template <int C, int CLS>
__device__ __forceinline__ unsigned lean_class_run_t(const RunEnv& E, const int cls_dyn, const int kinds, const int4 pr, const bool act, int2* store, const int ni, unsigned& run_writes, unsigned& wave_writes, int& nar_all) {
  const int cls = CLS >= 0 ? CLS : cls_dyn;
  const int lane = here(threadIdx.x) & 63, s = E.s;
  const int kx = kinds & 3, ky = (kinds >> 2) & 3, kz = (kinds >> 4) & 3;  // wave-uniform
  const LeanOperand<C> ox = lean_operand<C>(store, ni, pr.y, act, kx), oy = lean_operand<C>(store, ni, pr.z, act, ky), oz = lean_operand<C>(store, ni, pr.w, act, kz);
  const int vx = ox.v, vy = oy.v, vz = oz.v;  // idle lanes look at variable 0 (a constant) and move nothing
  // What moved during the run is kept per lane (bits 2k / 2k+1 = lower bound raised / upper bound lowered of operand k, a write count), in every layout.
  // (r03 kept it as six lane masks in SGPR pairs for the COMPACT kernels, where this function is a side path -- 1.4 of 51 runs per node of wordpress7_500 -- because
  //  the per-lane form cost that kernel 4 % through register allocation then.  r04, with the persistent loop's invariants out of its registers, those masks were what
  //  spilled around every run: per-lane bits everywhere, 4.76e7 -> 4.93e7 nodes/s same box; the !LANE_BITS form stays for A/B.)
  constexpr bool LANE_BITS = true;
  int nar_bits = 0;
  unsigned lane_writes = 0;
  unsigned long long mxl = 0, mxu = 0, myl = 0, myu = 0, mzl = 0, mzu = 0;  // (!LANE_BITS) lanes that moved each bound during the run
  unsigned iters = 0;
  // a column of constants (kind 3: singletons of the root, in the slab or carried by the record) is read once: it can only change by failing
  Itv KX{0, 0}, KY{0, 0}, KZ{0, 0};
  if (kx == 3) KX = load_dom<C>(store, ni, vx);
  if (ky == 3) KY = load_dom<C>(store, ni, vy);
  if (kz == 3) KZ = load_dom<C>(store, ni, vz);
  for (;;) {
    const Itv X = kx == 3 ? KX : lean_load<C>(store, ni, ox, kx), Y = ky == 3 ? KY : lean_load<C>(store, ni, oy, ky), Z = kz == 3 ? KZ : lean_load<C>(store, ni, oz, kz);
    int xl = X.lb, xu = X.ub, yl = Y.lb, yu = Y.ub, zl = Z.lb, zu = Z.ub;  // the new bounds: start from the current ones
    bool ent;
    if (cls == K_ADD) {
      xl = imax(xl, Y.lb + Z.lb); xu = imin(xu, Y.ub + Z.ub);
      yl = imax(yl, X.lb - Z.ub); yu = imin(yu, X.ub - Z.lb);
      zl = imax(zl, X.lb - Y.ub); zu = imin(zu, X.ub - Y.lb);
      ent = X.lb == X.ub && Y.lb == Y.ub && Z.lb == Z.ub;
    } else if (cls == K_MUL_NN) {
      xl = imax(xl, (int)((unsigned)Y.lb * (unsigned)Z.lb)); xu = imin(xu, (int)((unsigned)Y.ub * (unsigned)Z.ub));
      const bool xnz = X.lb > 0;  // a non-zero product has non-zero factors
      yl = sel(xnz && Y.lb == 0, 1, yl); yu = sel(xnz && Y.ub == 0, -1, yu);
      zl = sel(xnz && Z.lb == 0, 1, zl); zu = sel(xnz && Z.ub == 0, -1, zu);
      const unsigned uxl = (unsigned)imax(X.lb, 0), uxu = (unsigned)imax(X.ub, 0);
      // A narrowing pre-test before the divisions (r05 had it in the product rule of the global-memory kernels only): ceil(x.lb / z.ub) > y.lb <=> x.lb > y.lb * z.ub and
      // floor(x.ub / z.lb) < y.ub <=> x.ub < y.ub * z.lb, and the same two cross products answer for z.  Near the fixpoint, where most passes happen, no lane needs a
      // quotient: two multiplications instead of four unsigned divisions (~30 VALU each, for the whole wave).  r06 block counts: 430 VALU per product run of wordpress7_500.
      const unsigned p1 = (unsigned)Y.lb * (unsigned)Z.ub, p2 = (unsigned)Y.ub * (unsigned)Z.lb;  // (products of a flagged slice stay below 2^30)
      const bool need_y = Z.lb > 0 && (uxl > p1 || uxu < p2), need_z = Y.lb > 0 && (uxl > p2 || uxu < p1);
      if (wave_any(act && need_y)) {  // y within x / z (0 not in z)
        const unsigned dzu = umax1(Z.ub), dzl = umax1(Z.lb);
        const int lo = (int)((uxl + dzu - 1u) / dzu), hi = (int)(uxu / dzl);
        yl = sel(Z.lb > 0, imax(yl, lo), yl); yu = sel(Z.lb > 0, imin(yu, hi), yu);
      }
      if (wave_any(act && need_z)) {  // z within x / y
        const unsigned dyu = umax1(Y.ub), dyl = umax1(Y.lb);
        const int lo = (int)((uxl + dyu - 1u) / dyu), hi = (int)(uxu / dyl);
        zl = sel(Y.lb > 0, imax(zl, lo), zl); zu = sel(Y.lb > 0, imin(zu, hi), zu);
      }
      ent = X.lb == X.ub && Y.lb == Y.ub && Z.lb == Z.ub && X.lb == (int)((unsigned)Y.lb * (unsigned)Z.lb);
    } else if (cls == K_LEQ_T) {
      yu = imin(yu, Z.ub); zl = imax(zl, Y.lb);
      ent = Y.ub <= Z.lb;
    } else if (cls == K_LEQ_F) {
      yl = imax(yl, Z.lb + 1); zu = imin(zu, Y.ub - 1);
      ent = Y.lb > Z.ub;
    } else if (cls == K_EQ_T) {
      yl = imax(yl, Z.lb); yu = imin(yu, Z.ub); zl = imax(zl, Y.lb); zu = imin(zu, Y.ub);
      ent = Y.lb == Y.ub && Z.lb == Z.ub;
    } else if (cls == K_EQ_F) {
      const bool ys = Y.lb == Y.ub, zs = Z.lb == Z.ub;
      yl = sel(zs && Y.lb == Z.lb, Z.lb + 1, yl); yu = sel(zs && Y.ub == Z.lb, Z.lb - 1, yu);
      zl = sel(ys && Z.lb == Y.lb, Y.lb + 1, zl); zu = sel(ys && Z.ub == Y.lb, Y.lb - 1, zu);
      ent = Y.ub < Z.lb || Y.lb > Z.ub;
    } else if (cls == K_MIN) {
      xl = imax(xl, imin(Y.lb, Z.lb)); xu = imin(xu, imin(Y.ub, Z.ub));
      yl = imax(yl, X.lb); zl = imax(zl, X.lb);
      yu = sel(Z.lb > X.ub, imin(yu, X.ub), yu); zu = sel(Y.lb > X.ub, imin(zu, X.ub), zu);
      ent = X.lb == X.ub && Y.lb == Y.ub && Z.lb == Z.ub;
    } else if (cls == K_MAX) {
      xl = imax(xl, imax(Y.lb, Z.lb)); xu = imin(xu, imax(Y.ub, Z.ub));
      yu = imin(yu, X.ub); zu = imin(zu, X.ub);
      yl = sel(Z.ub < X.lb, imax(yl, X.lb), yl); zl = sel(Y.ub < X.lb, imax(zl, X.lb), zl);
      ent = X.lb == X.ub && Y.lb == Y.ub && Z.lb == Z.ub;
    } else {  // K_LEQ_R, K_EQ_R: x is the truth value of the comparison
      const bool t = X.lb >= 1, f = X.ub <= 0, u = !t && !f;
      if (cls == K_LEQ_R) {
        const bool le = Y.ub <= Z.lb, gt = Y.lb > Z.ub;
        xl = sel(u && le, 1, xl); xu = sel(u && gt, 0, xu);
        yu = sel(t, imin(yu, Z.ub), yu); zl = sel(t, imax(zl, Y.lb), zl);
        yl = sel(f, imax(yl, Z.lb + 1), yl); zu = sel(f, imin(zu, Y.ub - 1), zu);
        ent = (t && le) || (f && gt);
      } else {
        const bool ys = Y.lb == Y.ub, zs = Z.lb == Z.ub;
        const bool disjoint = Y.ub < Z.lb || Y.lb > Z.ub, same = ys && zs && Y.lb == Z.lb;
        xl = sel(u && same, 1, xl); xu = sel(u && disjoint, 0, xu);
        yl = sel(t, imax(yl, Z.lb), sel(f && zs && Y.lb == Z.lb, Z.lb + 1, yl));
        yu = sel(t, imin(yu, Z.ub), sel(f && zs && Y.ub == Z.lb, Z.lb - 1, yu));
        zl = sel(t, imax(zl, Y.lb), sel(f && ys && Z.lb == Y.lb, Y.lb + 1, zl));
        zu = sel(t, imin(zu, Y.ub), sel(f && ys && Z.ub == Y.lb, Y.lb - 1, zu));
        ent = (t && same) || (f && disjoint);
      }
    }
    ++iters;
    const bool cxl = xl != X.lb, cxu = xu != X.ub, cyl = yl != Y.lb, cyu = yu != Y.ub, czl = zl != Z.lb, czu = zu != Z.ub;
    const bool moved = act && (cxl | cxu | cyl | cyu | czl | czu);
    if (!wave_any(moved)) {
      // quiet pass: the local fixpoint is reached; the slice's "not entailed" bit only ever goes 1 -> 0 below a node
      if (!wave_any(act && !ent) && lane == 0) { const int sq = here_s(__builtin_amdgcn_readfirstlane(s)); (void)__hip_atomic_fetch_and(&E.unent[sq >> 5], ~(1u << (sq & 31)), TB_RLX, TB_WG); }
      break;
    }
    if (moved) {
      if ((xl > xu) | (yl > yu) | (zl > zu)) st(&E.sh.bot, 1);
      if constexpr (C == 4) {
        narrow_var8(store, ni, vx, xl, xu, cxl, cxu);
        narrow_var8(store, ni, vy, yl, yu, cyl, cyu);
        narrow_var8(store, ni, vz, zl, zu, czl, czu);
      } else {
        if (cxl) raise_lb<C>(store, ni, vx, xl);
        if (cxu) lower_ub<C>(store, ni, vx, xu);
        if (cyl) raise_lb<C>(store, ni, vy, yl);
        if (cyu) lower_ub<C>(store, ni, vy, yu);
        if (czl) raise_lb<C>(store, ni, vz, zl);
        if (czu) lower_ub<C>(store, ni, vz, zu);
      }
    }
    if constexpr (LANE_BITS) {
      // what this lane moved in this pass as per-lane bits: kept off the scalar unit, the busiest resource of a network like accap_a3 (84 %;
      // six ballots, six 64-bit ORs and six bit counts per narrowing pass are ~25 scalar instructions): accap_a3 6.4e7 -> 7.8e7 nodes/s
      const int moved_bits = !act ? 0 : ((cxl ? 1 : 0) | (cxu ? 2 : 0) | (cyl ? 4 : 0) | (cyu ? 8 : 0) | (czl ? 16 : 0) | (czu ? 32 : 0));
      nar_bits |= moved_bits;
      lane_writes += (unsigned)__builtin_popcount((unsigned)moved_bits);
    } else {
      const unsigned long long bxl = wave_ballot(act && cxl), bxu = wave_ballot(act && cxu), byl = wave_ballot(act && cyl), byu = wave_ballot(act && cyu),
                               bzl = wave_ballot(act && czl), bzu = wave_ballot(act && czu);
      mxl |= bxl; mxu |= bxu; myl |= byl; myu |= byu; mzl |= bzl; mzu |= bzu;
      wave_writes += (unsigned)(__builtin_popcountll(bxl) + __builtin_popcountll(bxu) + __builtin_popcountll(byl) + __builtin_popcountll(byu) +
                                __builtin_popcountll(bzl) + __builtin_popcountll(bzu));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    if (ld(&E.sh.bot)) break;
    if ((iters % WAVE_WATCHDOG_PERIOD) == 0) {  // watchdog inside the wave-local loop (see fixpoint)
      if (lane == 0 && deadline_passed(E.P)) st(&E.sh.abort, 1);
      if (ld(&E.sh.abort)) break;
    }
  }
  if constexpr (LANE_BITS) {
    nar_all = nar_bits;
    run_writes += lane_writes;
  } else if ((mxl | mxu | myl | myu | mzl | mzu) != 0ull) {
    const unsigned long long me = 1ull << lane;
    nar_all = ((mxl & me) ? 1 : 0) | ((mxu & me) ? 2 : 0) | ((myl & me) ? 4 : 0) | ((myu & me) ? 8 : 0) | ((mzl & me) ? 16 : 0) | ((mzu & me) ? 32 : 0);
  }
  return iters;
}
// The class of a slice is wave-uniform and fixed for the run: outside the COMPACT kernels (where this function is a side path and its size is paid in
// registers by everything around it) the pass loop is instantiated per class -- a slice of `max` or of reified `<=` no longer walks a chain of
// seven scalar compares and branches in every pass (the scalar unit is the busiest resource of a network like accap_a3).
template <int C>
__device__ __forceinline__ unsigned lean_class_run(const RunEnv& E, const int cls, const int kinds, const int4 pr, const bool act, int2* store, const int ni, unsigned& run_writes, unsigned& wave_writes, int& nar_all) {
#ifndef TB_LEAN_SWITCH_C1
// The COMPACT event kernels (C == 1): 0 = one pass loop with the class as a run-time value (r03-r05: the lean class run is a side path of the headline's branch and bound, 1.4 of 37
// runs per node, and its size is paid in registers by everything around it); 1 = a pass loop per class as in the other layouts; 2 = a pass loop of its own for sums only.
// r06, same box, two passes each (profiles/r06_ab_lean_switch.txt): the proof search of the `sharded_search` record (11.5 lean class runs of 33 per node, sums most of them)
// takes 0.814 s with 0, 0.789 s with 1, 0.781 s with 2; the headline step 6.126 / 6.111 / 6.110e7 nodes/s (-0.25 %: inside the box-to-box spread).  2 is the default.
#define TB_LEAN_SWITCH_C1 2
#endif
  if constexpr (C == 1 && !TB_LEAN_SWITCH_C1) return lean_class_run_t<C, -1>(E, cls, kinds, pr, act, store, ni, run_writes, wave_writes, nar_all);
  else if constexpr (C == 1 && TB_LEAN_SWITCH_C1 == 2) {
    if (cls == K_ADD) return lean_class_run_t<C, K_ADD>(E, cls, kinds, pr, act, store, ni, run_writes, wave_writes, nar_all);
    return lean_class_run_t<C, -1>(E, cls, kinds, pr, act, store, ni, run_writes, wave_writes, nar_all);
  } else {
    // (specialising the operand kinds of the commonest signatures as well -- sums of three variables, `c = y + z`, min / max over Booleans -- was
    //  measured: accap_a3 +0.6 %, trains15 -1 %; not kept)
    switch (cls) {
      case K_ADD: return lean_class_run_t<C, K_ADD>(E, cls, kinds, pr, act, store, ni, run_writes, wave_writes, nar_all);
      case K_MIN: return lean_class_run_t<C, K_MIN>(E, cls, kinds, pr, act, store, ni, run_writes, wave_writes, nar_all);
      case K_MAX: return lean_class_run_t<C, K_MAX>(E, cls, kinds, pr, act, store, ni, run_writes, wave_writes, nar_all);
      case K_LEQ_R: return lean_class_run_t<C, K_LEQ_R>(E, cls, kinds, pr, act, store, ni, run_writes, wave_writes, nar_all);
      case K_EQ_R: return lean_class_run_t<C, K_EQ_R>(E, cls, kinds, pr, act, store, ni, run_writes, wave_writes, nar_all);
      case K_LEQ_T: return lean_class_run_t<C, K_LEQ_T>(E, cls, kinds, pr, act, store, ni, run_writes, wave_writes, nar_all);
      default: return lean_class_run_t<C, -1>(E, cls, kinds, pr, act, store, ni, run_writes, wave_writes, nar_all);  // K_LEQ_F, K_EQ_T, K_EQ_F: rare
    }
  }
}

// Slice signatures with a dedicated run (word0 >> 16 of the slice's records: class set | operand kinds << 10; see pack_props).
// kinds per operand: 0 mixed, 1 integer variables, 2 Booleans of the COMPACT layout, 3 constants.
constexpr unsigned kinds(unsigned kx, unsigned ky, unsigned kz) { return (kx | (ky << 2) | (kz << 4)) << 10; }
constexpr unsigned KEY_LEQT_BB = (1u << K_LEQ_T) | kinds(3, 2, 2);   // b1 <= b2 (implication between two Booleans)
constexpr unsigned KEY_EQR_BIC = (1u << K_EQ_R) | kinds(2, 1, 3);    // b = (y = k)
constexpr unsigned KEY_LEQR_BIC = (1u << K_LEQ_R) | kinds(2, 1, 3);  // b = (y <= k)

// Tuning build, knob 0x10000: wave 0 splits its time in the rounds into record fetch / slice body / successor marks / barrier
// (core-clock ticks >> 4 in BlockStats::dbg[0..3], runs and rounds in dbg[4..5]; printed by tb_session_finish when verbose).
#ifdef TB_TUNING
#define TB_PROF_MARK(slot) do { if (prof && wave == 0) { const long long t_ = clock64(); if (lane == 0) sh.bs.dbg[slot] += (int)((t_ - tprof) >> 4); tprof = t_; } } while (0)
#define TB_PROF_COUNT(slot) do { if (prof && wave == 0 && lane == 0) sh.bs.dbg[slot] += 1; } while (0)
#else
#define TB_PROF_MARK(slot) do { } while (0)
#define TB_PROF_COUNT(slot) do { } while (0)
#endif

// LDS bytes of the two dirty bitmaps of the event-driven fixpoint (current round, next round).
__host__ __device__ inline size_t dirty_region_bytes(int dirty_words) { return (((size_t)dirty_words * 8 + 15) / 16) * 16; }

// Event-driven WAC1 (tb_config.fixpoint = 2): rounds over a dirty set of 64-propagator slices.
//
// Round r runs every slice whose bit is set in bitmap r & 1; a slice that narrows a variable marks the OTHER slices
// reading it in bitmap (r + 1) & 1; one s_barrier separates two rounds; the fixpoint is reached when a round marks
// nothing.  Slices are owned statically: slice s belongs to wave s % nw, which clears the bit before it loads the
// domains and is the only one to do so -- no claim, no atomic with a return value, no "busy" counter.
// Why rounds: a slice fed by many variables that one long cascade narrows one after the other (the terms of the
// objective's sum, each read by the same element-constraint slice) is run once per round in which any of them moved,
// not once per narrowing as it is when idle waves grab marked slices immediately (measured on wordpress7_500 with 4
// waves: 250 slice runs per node asynchronously against 82 with a single wave).
// Inside a round a wave still iterates each slice to its local fixpoint (WAC1), and a slice sees the narrowings other
// waves have already made in the same round.
template <int C, int TB = 0>
__device__ __forceinline__ int fixpoint_event(const DevProblem& P, BlockShared& sh, int2* store, const int4* props,
                                              const EventState& es, ThreadCounters& tc, bool& all_entailed) {
  const int tid = here(threadIdx.x), T = block_threads<TB>(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = T >> 6;
  const int n = P.n_props, W = es.words, S = P.n_slices;
  const bool prof = (knobs(P) & 0x10000) != 0;
  long long tp0 = 0;
  if (tid == 0 && prof) tp0 = wall_clock64();
  // ---- initial dirty set (bitmap 0): everything, or the slices of the variables changed since the last fixpoint
  TB_REGION(2);
  const int cnt = ld(&sh.chg_count[0]);
  const bool all = ld(&sh.ev_all) != 0 || cnt > es.cap;
  // Entailed-slice removal: a slice whose 64 propagators were all entailed when it last ran stays entailed in the
  // whole subtree (domains only shrink), so it is dropped when its turn comes.  The bytes are undefined before the root pass.
  const bool root_pass = ld(&sh.ev_all) != 0;
  const bool drop_entailed = !root_pass && !(knobs(P) & 0x20000);
  unsigned* bm0 = es.dirty;
  unsigned* ubits = reinterpret_cast<unsigned*>(es.unent);  // bit s: some propagator of slice s was not entailed when it last ran
  if (all) {
    for (int i = tid; i < W; i += T) {
      const int left = S - i * 32;
      bm0[i] = left >= 32 ? 0xffffffffu : ((1u << left) - 1u);
    }
    if (root_pass) for (int i = tid; i < W; i += T) ubits[i] = bm0[i];  // nothing is known to be entailed yet (every valid bit set)
  } else {
    for (int rep = reps_of(P, 1); rep > 0; --rep)
    for (int e0 = wave * 64; e0 < cnt; e0 += T) {  // one lane per entry; long lists are finished cooperatively
      TB_REGION(3);
      const int e = e0 + lane;
      const int entry = e < cnt ? es.list[TB_IDX(20, e, chg_cap)] : 0;
      int ev = e < cnt ? ((entry >> 30) & 3) : 0;
#ifdef TB_TRAP_SEED
      // (debugging aid, r06: the intermittent memory fault of r04 is a change-list entry whose variable is -1 .. -128 -- var_adj + 32 GiB - 4 KiB; catch it before the load)
      if (ev != 0 && (unsigned)(entry & 0x3fffffff) >= (unsigned)P.n_vars) {
        { const int w[4] = {entry, e, cnt, es.list[1 - (e & 1)]}; trap_report(P, sh, 3, w, 4); }
        ev = 0;
      }
#endif
      int deg = 0, off = 0;
      bool did = false;
      const bool more = mark_var(P, bm0, entry & 0x3fffffff, -1, ev, deg, off, did);
      const unsigned long long mm = wave_ballot(more);
      if (mm) (void)mark_tail(P, bm0, mm, deg, off, ev, -1);
    }
  }
  TB_REGION(59);
  if (tid == 0) { st(&sh.unent[0], 0); st(&sh.flag[0], 0); st(&sh.flag[1], 0); }
  __syncthreads();
  if (tid == 0) { st(&sh.ev_all, 0); st(&sh.chg_count[0], 0); }
  if (tid == 0 && prof) { const long long t = wall_clock64(); TB_PROF_ADD(sh.bs, TB_PROF_SEEDING, t - tp0); tp0 = t; }  // profiling: seeding
  // ---- rounds
  // bits of a bitmap word owned by this wave: slices s with s % nw == wave (nw divides 32)
  // (nw is 1, 2, 4, 8 or 16: the pattern 0...010...01 with a one every nw bits, shifted to this wave's residue -- r05 built it with a 16-iteration scalar loop, 210 SALU per
  //  node and wave of accap_a3 by the block counts of r06, 4 % of the kernel's scalar instructions)
  const unsigned own = (0xffffffffu / ((1u << nw) - 1u)) << wave;
  int rounds = 0;
  unsigned wave_iters_total = 0;  // wave-uniform
  unsigned wave_active_total = 0; // wave-uniform: the same, times the propagators of each slice (idle lanes of padded slices not counted)
  unsigned wave_writes = 0;       // wave-uniform: narrowed bounds counted on lane masks (s_bcnt1), credited to lane 0 at the end
#ifdef TB_TUNING
  long long tprof = prof ? clock64() : 0;
#endif
  for (;; ++rounds) {
    const int k = rounds % 3;
    TB_REGION(4);
    unsigned* cur = es.dirty + (rounds & 1) * W;
    unsigned* nxt = es.dirty + ((rounds + 1) & 1) * W;
    // "this wave marked something for the next round" goes straight to the round's LDS flag (lane 0, one store per marking run): accumulated in a register it was
    // a 64-bit lane mask that lived -- and was spilled and reloaded -- through every run of the round
    auto note_marked = [&](bool did_any) { if (did_any && lane == 0) st(&sh.flag[k], 1); };
#if defined(TB_TUNING) || TB_DOUBLE_PHASE
    if (reps_of(P, 2) > 1)
      for (int base = 0; base < W; base += 64) {  // the scan once more, without clearing anything
        const int wi = base + lane;
        unsigned w = wi < W ? (__hip_atomic_load(&cur[wi], TB_RLX, TB_WG) & own) : 0u;
        if (w != 0) (void)__hip_atomic_fetch_and(&cur[wi], ~0u, TB_RLX, TB_WG);
        if (drop_entailed && w != 0) w &= __hip_atomic_load(&ubits[wi], TB_RLX, TB_WG);
        unsigned long long nz = wave_ballot(w != 0);
        asm volatile("" :: "s"(nz));
      }
#endif
#ifdef TB_TUNING
    if (prof && wave == 0) {  // census of the rounds by the number of slices that will run: 1, 2, 3-4, 5-8, 9-16, more -> dbg[24..29]; per wave maximum dbg[30]
      int total = 0, mx = 0;
      for (int base = 0; base < W; base += 64) {
        const int wi = base + lane;
        unsigned wa = wi < W ? __hip_atomic_load(&cur[wi], TB_RLX, TB_WG) : 0u;
        if (drop_entailed && wi < W) wa &= __hip_atomic_load(&ubits[wi], TB_RLX, TB_WG);
        int c = __popc(wa), c0 = __popc(wa & own);
        for (int off = 32; off > 0; off >>= 1) { c += __shfl_xor(c, off, 64); c0 += __shfl_xor(c0, off, 64); }
        total += c; mx += c0;
      }
      if (lane == 0 && total > 0) { sh.bs.dbg[total == 1 ? 24 : (total == 2 ? 25 : (total <= 4 ? 26 : (total <= 8 ? 27 : (total <= 16 ? 28 : 29))))] += 1; sh.bs.dbg[30] += total; sh.bs.dbg[31] += mx; }
    }
    if (prof) __syncthreads();  // (the other waves must not clear their bits under the census)
#endif
    for (int base = 0; base < W; base += 64) {
      const int wi = base + lane;
      unsigned w = wi < W ? (__hip_atomic_load(&cur[wi], TB_RLX, TB_WG) & own) : 0u;
      if (w != 0) (void)__hip_atomic_fetch_and(&cur[wi], ~w, TB_RLX, TB_WG);  // mine, cleared before any domain is loaded
      // entailed-slice removal: a slice whose 64 propagators were all entailed when it last ran is not even looked at
      if (drop_entailed && w != 0) w &= __hip_atomic_load(&ubits[wi], TB_RLX, TB_WG);
      // my slices of this round, one after the other: wave-uniform iteration over the set bits of the words held by the lanes.
      // The successor record of the NEXT slice is fetched while the current one runs (4 registers; every kind of run needs that
      // record, the lean implication runs nothing else): the L2 round trip of a run's first load was a third of its duration.
      unsigned long long nz = wave_ballot(w != 0);
      unsigned word = 0;
      int wl = 0;
      auto next_slice = [&]() -> int {
        while (word == 0) {
          if (nz == 0) return -1;
          wl = __builtin_ctzll(nz);
          nz &= nz - 1;
          word = (unsigned)__builtin_amdgcn_readlane((int)w, wl);
        }
        const int b = __builtin_ctz(word);
        word &= word - 1;
        return (base + wl) * 32 + b;
      };
      int s = next_slice();
#if TB_SC_PREFETCH
      int4 sc_cur = make_int4(0, 0, 0, 0);
      if (s >= 0) sc_cur = (es.succ + (size_t)s * 64)[lane];
#endif
      while (s >= 0) {
        TB_REGION(5);
#if TB_SC_PREFETCH
        const int s_next = next_slice();
#endif
        s = TB_IDX(13, s, n_slices);
#if TB_SC_PREFETCH
        const int4 sc = sc_cur;
        if (s_next >= 0) sc_cur = (es.succ + (size_t)s_next * 64)[lane];
#endif
        // (r06, measured and dropped: the table in reverse order right below the records, addressed from the record pointer this loop holds anyway -- one dependent scalar
        //  load fewer per run, ~100 scalar instructions more per node for the 64-bit address: wordpress7_500 6.112e7 -> 6.111e7 nodes/s, trains15 +0.5 %, accap_a3 -0.2 %)
        const int2 info = cst(P.slice_info)[s];
#if !TB_SC_PREFETCH
        const int4* const succ_slice = es.succ + (size_t)s * 64;  // uniform base + lane: no 64-bit VALU address arithmetic
#endif
        if (dead_node(sh)) break;  // the node failed in another wave
#if defined(TB_TUNING) || TB_DOUBLE_PHASE
        if (reps_of(P, 9) > 1) {  // (tuning: the part of a run's prologue that sits outside the doubled run -- scalar info load, record base, failure check)
          const DevProblem* p2 = &P; asm volatile("" : "+s"(p2));
          const int2 i2 = cst(p2->slice_info)[s];
          const int4* b2 = es.succ + (size_t)s * 64;
          asm volatile("" :: "s"(i2.x), "s"(i2.y), "s"(b2));
          if (dead_node(sh)) break;
        }
#endif
        for (int rep = reps_of(P, 3); rep > 0; --rep) {
          const bool act = lane < (info.y & 0xff);
#if !TB_SC_PREFETCH
          const int4 sc = succ_slice[lane];
#endif
          if (C && (info.y & 0x100)) {
            // ---- lean implication run: `y <= z` on two 2-bit Booleans (bit 0: lb raised to 1, bit 1: ub lowered to 0) from the successor
            // record alone (engine.hip: pack_succ): z.ub = 0 forces y.ub = 0, y.lb = 1 forces z.lb = 1; entailed once y.ub = 0 or z.lb = 1.
            unsigned* const words = reinterpret_cast<unsigned*>(store);  // (word indexes from the start of the slab, whatever the layout)
            unsigned* const wy = words + TB_IDX(14, (unsigned)sc.x & 0xffffu, store_words);
            unsigned* const wz = words + TB_IDX(14, (unsigned)sc.x >> 16, store_words);
            const int ys = (sc.w >> 19) & 30, zs = (sc.w >> 23) & 30;
            const unsigned am = act ? ~0u : 0u;  // idle lanes of a padded slice see nothing to do
            unsigned acc = 0u;  // what this lane narrowed during the run: bit 1 its y (upper bound lowered), bit 0 its z (lower bound raised)
            TB_PROF_MARK(0);
            unsigned iters = 0;
            unsigned run_writes_lean = 0;  // per lane
            for (;;) {
              TB_REGION(6);
              // predicates as integers on the raw bit pairs (bit 0 of yb / zb: lb raised, bit 1: ub lowered), one vote each
              const unsigned yb = __hip_atomic_load(wy, TB_RLX, TB_WG) >> ys, zb = __hip_atomic_load(wz, TB_RLX, TB_WG) >> zs;
              const unsigned ny = zb & ~yb & 2u & am, nz = yb & ~zb & 1u & am;             // y.ub := 0 / z.lb := 1
              const unsigned bad = ((yb & (yb >> 1)) | (zb & (zb >> 1)) | (yb & (zb >> 1))) & 1u & am;  // an empty domain, or y true and z false
              ++iters;
              const unsigned long long m_bad = mask_nz(bad);
              if (mask_nz(ny | nz | bad) == 0ull) {
                // quiet pass: the local fixpoint is reached; the slice's "not entailed" bit only ever goes 1 -> 0 below a node
                if (mask_nz(~(yb >> 1) & ~zb & 1u & am) == 0ull && lane == 0)
                  { const int sq = here_s(__builtin_amdgcn_readfirstlane(s)); (void)__hip_atomic_fetch_and(&ubits[sq >> 5], ~(1u << (sq & 31)), TB_RLX, TB_WG); }
                break;
              }
              if (m_bad != 0ull) { if (lane == 0) st(&sh.bot, 1); break; }  // the node fails: what this pass would still write is moot
              if (ny) (void)__hip_atomic_fetch_or(wy, 2u << ys, TB_RLX, TB_WG);
              if (nz) (void)__hip_atomic_fetch_or(wz, 1u << zs, TB_RLX, TB_WG);
              run_writes_lean += (ny >> 1) + nz;
              acc |= ny | nz;
              if (info.y & 0x1000) {
                // (r05) No variable of this slice is the y of one record and the z of another (engine.hip: slice_infos): what this pass wrote enables no other lane of the
                // slice, so there is nothing for a confirmation pass to find -- whoever else narrows one of these variables marks the slice for the next round.  Entailed
                // now = y false or z true, counting what was just written.  (wordpress7_500: 16 of a node's 29 implication runs narrow something; each saved a pass.)
                if (mask_nz(~((yb | ny) >> 1) & ~(zb | nz) & 1u & am) == 0ull && lane == 0)
                  { const int sq = here_s(__builtin_amdgcn_readfirstlane(s)); (void)__hip_atomic_fetch_and(&ubits[sq >> 5], ~(1u << (sq & 31)), TB_RLX, TB_WG); }
                break;
              }
              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
              if (ld(&sh.bot)) break;
            }
            TB_REGION(7);
            {  // (tuning build: the class / useless-run filters of the census, as for the other runs below)
              const int want = (knobs(P) >> 28) & 15;
              if (rep == reps_of(P, 3) && (want == 0 || want - 1 == K_LEQ_T) && (!(knobs(P) & 0x40) || mask_nz(acc) == 0ull)) {
                wave_iters_total += (pk(P) & 0x400000) ? 1u : iters;
                wave_active_total += iters * (unsigned)__builtin_popcountll(wave_ballot(act));  // (the lanes that hold a record: the vote is live anyway, the count from the info word was a spilled SGPR)
              }
            }
            tc.writes += run_writes_lean;
#ifdef TB_TUNING
            if (P.slice_census != nullptr && lane == 0) { atomicAdd(&glob(P.slice_census)[2 * s], 1u); if (mask_nz(acc) == 0ull) atomicAdd(&glob(P.slice_census)[2 * s + 1], 1u); }
#endif
#ifdef TB_TUNING
            if (prof && wave == 0) {
              const long long t_ = clock64();
              if (lane == 0) { sh.bs.dbg[1] += (int)((t_ - tprof) >> 4); sh.bs.dbg[6] += (int)((t_ - tprof) >> 4); sh.bs.dbg[11] += 1; }
              tprof = t_;
            }
#endif
            if (pk(P) & 0x8) {  // tuning: one more (quiet) pass
              const unsigned yb = (__hip_atomic_load(wy, TB_RLX, TB_WG) >> ys) & 3u, zb = (__hip_atomic_load(wz, TB_RLX, TB_WG) >> zs) & 3u;
              const unsigned long long again = wave_ballot(act && ((zb & ~yb & 2u) | (yb & ~zb & 1u)) != 0u);
              asm volatile("" :: "s"(again));
            }
            for (int mrep = (pk(P) & 0x1) ? 2 : 1; mrep > 0; --mrep)
            if (mask_nz(acc) != 0ull) {
              TB_REGION(8);
              // successors: the slots hold the readers interested in exactly these events (y.ub lowered / z.lb raised), pre-filtered
              const bool my_ny = (acc & 2u) != 0u, my_nz = (acc & 1u) != 0u;
              unsigned ty = my_ny ? (unsigned)sc.y : 0xffffffffu;
              const unsigned tz = my_nz ? (unsigned)sc.z : 0xffffffffu;
              // conditional wake-up (pack_succ, bit 30 of w): y's only reader that cares about "became false" is the chain record `y = (Y = kv)`, whose
              // rule reacts to a false y only when kv sits on a bound of Y -- one LDS read here instead of a whole channelling run there.  (Should a
              // bound of Y move onto kv later, whoever moves it wakes that slice.)
              const bool cond = my_ny && ((sc.w >> 30) & 1) != 0;
              if (wave_any(cond)) {
                TB_REGION(9);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // my write of y, then this read: the chain run does the same with roles reversed
                const Itv Yc = load_ivar<C>(store, P.n_int, (cond ? (int)(ty >> 16) : 0) | (C == 4 ? C8_BASE_BIAS << 16 : 0));  // (COMPACT8: kv is relative to Y's base, pack_succ)
                const int kv = (int)(short)((((unsigned)sc.w >> 3) & 0x3fffu) | ((((unsigned)sc.w >> 28) & 3u) << 14));
                if (cond) ty = (kv == Yc.lb || kv == Yc.ub) ? (ty | 0xffff0000u) : 0xffffffffu;
              }
              TB_REGION(45);
              bool did = false;
              if ((ty & 0xffffu) != 0xffffu) { mark_slice(nxt, (int)(ty & 0xffffu)); did = true; }
              if ((ty >> 16) != 0xffffu) { mark_slice(nxt, (int)(ty >> 16)); did = true; }
              if ((tz & 0xffffu) != 0xffffu) { mark_slice(nxt, (int)(tz & 0xffffu)); did = true; }
              if ((tz >> 16) != 0xffffu) { mark_slice(nxt, (int)(tz >> 16)); did = true; }
              const int ey = (my_ny && ((sc.w >> 17) & 1)) ? EV_UB : 0, ez = (my_nz && ((sc.w >> 2) & 1)) ? EV_LB : 0;
              if (wave_any((ey | ez) != 0)) {  // more than two interested readers: the variable's adjacency record
                TB_REGION(10);
                const int base_w = bool_word0<C>(P.n_int), n_i = ni_int<C>(P.n_int);  // first Boolean word of the slab
                const int vy = n_i + ((((int)((unsigned)sc.x & 0xffffu)) - base_w) << 4) + (ys >> 1);
                const int vz = n_i + ((((int)((unsigned)sc.x >> 16)) - base_w) << 4) + (zs >> 1);
                int dy = 0, dz = 0, oy = 0, oz = 0;
                const bool tly = mark_var(P, nxt, vy, s, ey, dy, oy, did);
                const bool tlz = mark_var(P, nxt, vz, s, ez, dz, oz, did);
                const unsigned long long my = wave_ballot(tly), mz = wave_ballot(tlz);
                if (my) did |= mark_tail(P, nxt, my, dy, oy, ey, s);
                if (mz) did |= mark_tail(P, nxt, mz, dz, oz, ez, s);
              }
              TB_REGION(46);
              note_marked(wave_any(did));
            }
            TB_REGION(11);
            TB_PROF_MARK(2);
            TB_PROF_COUNT(4);
            continue;
          }
          TB_REGION(12);
          const int4 pr_first = props[s * 64 + lane];  // the arrays are padded to whole slices
#if defined(TB_TUNING) || TB_DOUBLE_PHASE
          int4 pr_again = pr_first;
          if (rep == 1 && reps_of(P, 3) > 1) { const int4* pp = props + (s * 64 + lane); asm volatile("" : "+v"(pp)); pr_again = *pp; if (dead_node(sh)) break; }
          const int4 pr = pr_again;
#else
          const int4 pr = pr_first;
#endif
          const RunEnv E{P, sh, nxt, ubits, s};
          const unsigned key = (unsigned)info.x >> 16;  // wave-uniform: scalar dispatch
          unsigned wave_iters;
          unsigned run_writes = 0;  // per lane, folded into the 64-bit counter once per run
          int nar_all = 0;          // operands this lane narrowed during the run
          TB_PROF_MARK(0);
          if (C && key == KEY_LEQT_BB) {
            // y <= z on two Booleans, straight on their 2-bit encodings (bit 0: lb raised to 1, bit 1: ub lowered to 0):
            // z.ub = 0 forces y.ub = 0, y.lb = 1 forces z.lb = 1; entailed once y.ub = 0 or z.lb = 1.
            TB_REGION(42);
            const BoolRef ry = bool_ref<C>(store, P.n_int, pr.z, act), rz = bool_ref<C>(store, P.n_int, pr.w, act);
            wave_iters = run_slice(E, nar_all, [&](bool& ch, bool& un_i, int& nar) {
              const unsigned yb = bool_bits(ry), zb = bool_bits(rz);
              const bool ny = (zb & 2u) && !(yb & 2u), nz = (yb & 1u) && !(zb & 1u);
              const bool empty_in = yb == 3u || zb == 3u;
              if (wave_any(act & (ny | nz | empty_in))) {
                if (act) {
                  if (empty_in | ((yb & 1u) && (zb & 2u))) st(&sh.bot, 1);
                  if (!empty_in) {
                    if (ny) bool_or(ry, 2u);
                    if (nz) bool_or(rz, 1u);
                    run_writes += (unsigned)ny + (unsigned)nz;
                    ch = ny | nz;
                    nar = ((int)ny << 3) | ((int)nz << 4);  // y.ub lowered, z.lb raised
                  }
                }
              }
              un_i = act & !((yb & 2u) || (zb & 1u));
            });
          } else if (C && key == KEY_EQR_BIC && !(knobs(P) & 0x4000000)) {
            // Channelling propagators b_i = (y = k_i) -- the index of an element constraint against its positions, its value
            // against the table.  Evaluated one by one, a false b_i only strips k_i when it sits exactly on a bound of y, so
            // a run of m excluded values costs m wave iterations.  Here the lanes that share a variable y (the records are
            // sorted by y: a slice holds a few groups of neighbouring lanes) compute the fixpoint of their rules jointly, all
            // groups at once: every lane carries its group's bounds, a ballot masked with the group's lanes says whether some
            // false b_i sits on a bound, and the bounds step over the excluded values in registers -- no memory traffic; then
            // every b_i outside the new bounds becomes false and, if y is assigned, its b_i true.  Same fixpoint, one pass.
            TB_REGION(13);
            const BoolRef rx = bool_ref<C>(store, P.n_int, pr.y, act);
            const int yv = act ? pr.z : 0;
            const int w0u = info.x;
            const bool single_pass = ((w0u >> 11) & 1) != 0;
            // Prepared by the host with the record (pack_succ): the constant's value in the z slots, the first and last lane of
            // my group (the lanes between two changes of y), and for the slice whether every group's constants are consecutive
            // integers in lane order -- lane g_start + (v - k0) then holds value v, and "how far do the excluded values reach from
            // this bound" is a bit scan over the ballot of the false b_i instead of a walk.
            const int kc = sc.z;
            const int g_start = (sc.w >> 21) & 63, g_last = ((sc.w >> 27) & 31) | (((sc.w >> 19) & 1) << 5);
            const unsigned long long gmask = act ? (((2ull << g_last) - 1ull) & ~((1ull << g_start) - 1ull)) : 0ull;
            const bool writer = act && lane == g_start;
            const bool dense = ((w0u >> 15) & 1) != 0;
            const int k0 = kc - (lane - g_start), k_last = k0 + (g_last - g_start);  // (meaningful when dense)
            // Eligible chains (engine.hip: Chains, slice_info 0x400): the record of value v of my y is record s * 64 + lane + (v - kc), whatever slice it
            // is in, and the chain's other slices are not among y's listed readers: the lane that writes y wakes the ones holding the values the
            // bound moved over (and the new bound itself: its b may be false, or y assigned) -- the others have nothing to do.
            const bool by_range = (info.y & 0x400) != 0;
            bool chain_marked = false;
            wave_iters = run_slice(E, nar_all, [&](bool& ch, bool& un_i, int& nar) {
              TB_REGION(14);
              const unsigned xb = bool_bits(rx);
              const Itv Y = load_ivar<C>(store, P.n_int, yv);
              const bool t = act && (xb & 1u), f = act && (xb & 2u), u = act && xb == 0u;
              bool again = false;
              int lb = Y.lb, ub = Y.ub;
              for (unsigned long long tm = wave_ballot(t); tm; tm &= tm - 1) {  // y = k for every true b (normally at most one per group)
                const int l = __builtin_ctzll(tm);
                const int k = __builtin_amdgcn_readlane(kc, l);
                if ((gmask >> l) & 1ull) { lb = lb > k ? lb : k; ub = ub < k ? ub : k; }
              }
              if (dense) {
                const unsigned long long fm = wave_ballot(f) & gmask;
                if (lb <= ub && lb >= k0 && lb <= k_last) {
                  const unsigned long long open_up = ~(fm >> (g_start + (lb - k0)));  // first lane from lb's upwards whose b is not false
                  lb += open_up ? __builtin_ctzll(open_up) : 64;
                }
                if (lb <= ub && ub >= k0 && ub <= k_last) {
                  const unsigned long long open_dn = ~(fm << (63 - (g_start + (ub - k0))));
                  ub -= open_dn ? __builtin_clzll(open_dn) : 64;
                }
              } else {
                for (;;) {  // excluded values on the bounds, one step at a time, every group at its own pace
                  const unsigned long long ml = wave_ballot(f && kc == lb), mu = wave_ballot(f && kc == ub);
                  const bool live = lb <= ub;
                  const bool al = live && (ml & gmask) != 0ull, au = live && (mu & gmask) != 0ull;
                  if (!wave_any(al | au)) break;
                  if (al) lb = sat_add(lb, 1);
                  if (au) ub = sat_sub(ub, 1);
                }
              }
              const bool outside = kc < lb || kc > ub, hit = lb == ub && kc == lb;
              // (r06, dropped: leaving the pass after one vote when no lane has anything to write -- about half of the passes -- ADDS instructions, 7 307 -> 7 416 VALU per
              // node on wordpress7_500, and costs 5.5 % nodes/s, same box: the stores below are already skipped by their own execution masks, and the extra vote
              // and its live ranges are paid by every pass.  profiles/r06_ab_chan_early.txt)
              const bool bad = wave_any(act && (xb == 3u || lb > ub));
              un_i = act;
              if (!bad) {
                const bool set0 = u && outside, set1 = u && hit;
                // There is no confirmation pass to notice that ANOTHER wave emptied a domain I narrow in the same round (two slices
                // of one y walking its bounds towards each other, a b made true elsewhere while I make it false): look at what
                // my own atomics left behind.
                if (set0 | set1) {
                  const unsigned mine = set1 ? 1u : 2u;
                  const unsigned before = (__hip_atomic_fetch_or(rx.word, mine << rx.shift, TB_RLX, TB_WG) >> rx.shift) & 3u;
                  if ((before | mine) == 3u) st(&sh.bot, 1);
                }
                const bool cyl = writer && lb != Y.lb, cyu = writer && ub != Y.ub;
                if (cyl | cyu) {
                  if constexpr (C == 4) narrow_var8(store, P.n_int, yv, lb, ub, cyl, cyu);
                  else {
                    if (cyl) raise_ivar_lb<C>(store, P.n_int, yv, lb);
                    if (cyu) lower_ivar_ub<C>(store, P.n_int, yv, ub);
                  }
                  const Itv now = load_ivar<C>(store, P.n_int, yv);
                  if (now.lb > now.ub) st(&sh.bot, 1);
                  if (by_range) {
                    const int rec = s * 64 + lane - kc, q_max = P.n_slices - 1;
                    if (cyl) for (int q = imax((rec + Y.lb) >> 6, 0), qe = imin((rec + lb) >> 6, q_max); q <= qe; ++q) if (q != s) { mark_slice(nxt, q); chain_marked = true; }
                    if (cyu) for (int q = imax((rec + ub) >> 6, 0), qe = imin((rec + Y.ub) >> 6, q_max); q <= qe; ++q) if (q != s) { mark_slice(nxt, q); chain_marked = true; }
                  }
                }
                // The other half of the conditional wake-up (lean implication run above): a lane that makes a b false wakes this slice only if it then
                // finds b's value on a bound of y.  If it read y before the write just above, it found nothing -- so the lane whose value has just
                // BECOME a bound looks at its b once more, after that write: of the two, at least one sees the other's store (LDS operations of a wave
                // execute in order, and the writer's come before this read in the same instruction stream).  A b found false now means one more pass.
                if (wave_any(cyl | cyu)) {
                  TB_REGION(15);
                  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (relaxed accesses to different addresses: the compiler may not hoist the read)
                  const bool became_bound = act && !f && !set0 && ((lb != Y.lb && kc == lb) || (ub != Y.ub && kc == ub));
                  if (became_bound && (bool_bits(rx) & 2u)) again = true;
                }
                TB_REGION(47);
                run_writes += (unsigned)(set0 | set1) + (unsigned)cyl + (unsigned)cyu;
                // (y is reported by the lane that wrote it, or by every lane of the group when y's readers are dealt out over their records)
                const bool y_rep = writer || ((sc.w >> 20) & 1) != 0;
                nar = (int)set1 | ((int)set0 << 1) | ((y_rep && lb != Y.lb) ? 4 : 0) | ((y_rep && ub != Y.ub) ? 8 : 0);  // b true / false, y.lb / y.ub
                un_i = act && !(((t || set1) && hit) || ((f || set0) && outside));
              } else if (lane == 0) st(&sh.bot, 1);
              // When no truth variable occurs twice in the slice (word0 bit 11, pack_props) the joint fixpoint is reached and no
              // confirmation pass is needed; otherwise a b made false for one lane may still have to act through another one.
              ch = again || (single_pass ? false : nar != 0);
            });
            TB_REGION(16);
            if (by_range) note_marked(wave_any(chain_marked));
          } else if (C && (key == KEY_EQR_BIC || key == KEY_LEQR_BIC)) {
            // b = (y = k) / b = (y <= k): Boolean truth variable, integer y, constant k (read once per run)
            TB_REGION(17);
            const bool is_eq = key == KEY_EQR_BIC;
            const BoolRef rx = bool_ref<C>(store, P.n_int, pr.y, act);
            const int yv = act ? pr.z : 0;
            const int kc = sc.z;  // the constant's value travels in the z slots of the successor record (pack_succ)
            wave_iters = run_slice(E, nar_all, [&](bool& ch, bool& un_i, int& nar) {
              const unsigned xb = bool_bits(rx);
              const Itv Y = load_ivar<C>(store, P.n_int, yv);
              const bool t = (xb & 1u) != 0, f = (xb & 2u) != 0, u = !t && !f;
              const bool empty_in = (xb == 3u) | (Y.lb > Y.ub);
              bool set1, set0, ent;
              int nyl = Y.lb, nyu = Y.ub;
              if (is_eq) {  // wave-uniform
                const bool disjoint = Y.ub < kc || Y.lb > kc, same = Y.lb == Y.ub && Y.lb == kc;
                set1 = u && same; set0 = u && disjoint;
                nyl = t ? imax(Y.lb, kc) : ((f && Y.lb == kc) ? sat_add(kc, 1) : Y.lb);
                nyu = t ? imin(Y.ub, kc) : ((f && Y.ub == kc) ? sat_sub(kc, 1) : Y.ub);
                ent = (t && same) || (f && disjoint);
              } else {
                const bool le = Y.ub <= kc, gt = Y.lb > kc;
                set1 = u && le; set0 = u && gt;
                nyu = t ? imin(Y.ub, kc) : Y.ub;
                nyl = f ? imax(Y.lb, add_lo(kc, 1)) : Y.lb;
                ent = (t && le) || (f && gt);
              }
              const bool cyl = nyl != Y.lb, cyu = nyu != Y.ub;
              if (wave_any(act & (set1 | set0 | cyl | cyu | empty_in))) {
                if (act) {
                  if (empty_in | (nyl > nyu)) st(&sh.bot, 1);  // (a narrowed constant is an empty y: same condition)
                  if (!empty_in) {
                    if (set1 | set0) bool_or(rx, set1 ? 1u : 2u);
                    if constexpr (C == 4) narrow_var8(store, P.n_int, yv, nyl, nyu, cyl, cyu);
                    else {
                      if (cyl) raise_ivar_lb<C>(store, P.n_int, yv, nyl);
                      if (cyu) lower_ivar_ub<C>(store, P.n_int, yv, nyu);
                    }
                    const int kw = (int)(set1 | set0) + (int)cyl + (int)cyu;
                    run_writes += (unsigned)kw;
                    ch = kw != 0;
                    nar = (int)set1 | ((int)set0 << 1) | ((int)cyl << 2) | ((int)cyu << 3);
                  }
                }
              }
              un_i = act & !ent;
            });
          } else if (info.y & 0x800) {
            TB_REGION(66);
            wave_iters = lean_class_run_t<C, K_MUL_NN>(E, K_MUL_NN, (int)(key >> 10), pr, act, store, P.n_int, run_writes, wave_writes, nar_all);
          } else if (info.y & 0x200) {
            TB_REGION(43);
            wave_iters = lean_class_run<C>(E, __builtin_ctz(key & CLASS_SET_MASK), (int)(key >> 10), pr, act, store, P.n_int, run_writes, wave_writes, nar_all);
          } else {
            TB_REGION(44);
            wave_iters = run_slice(E, nar_all, [&](bool& ch, bool& un_i, int& nar) {
              apply<true, C>(pr, act, store, P.n_int, &sh.bot, ch, un_i, tc, 0, &nar);
            });
          }
#ifdef TB_TUNING
          if (prof && wave == 0) {  // body time by kind of run: dbg[6] Boolean implications, [7] joint channelling, [8] b = (y ~ k), [9] generic; dbg[10], dbg[11]: generic / implication runs
            const long long t_ = clock64();
            const int kind = (C && key == KEY_LEQT_BB) ? 6 : ((C && key == KEY_EQR_BIC && !(knobs(P) & 0x4000000)) ? 7 : ((C && (key == KEY_EQR_BIC || key == KEY_LEQR_BIC)) ? 8 : 9));
            if (lane == 0) { sh.bs.dbg[1] += (int)((t_ - tprof) >> 4); sh.bs.dbg[kind] += (int)((t_ - tprof) >> 4); if (kind == 9) sh.bs.dbg[10] += 1; if (kind == 6) sh.bs.dbg[11] += 1; }
            tprof = t_;
          }
#endif
          if (C != 0 && (info.y & 0x2000) && wave_any((nar_all & 2) != 0)) {
            // (r05) A channelling lane has just made c = (val = v) FALSE, and c's only readers that care are the implications b_i <= c of an element constraint
            // (engine.hip: pack_cond2).  Their rule can only make b_i false for the positions i whose table entry is v -- which the index's own channelling does as
            // soon as i leaves the index's domain.  So when no such position lies inside the index's bounds, or the index is assigned (the one b that is not false
            // is then TRUE, and its becoming true wakes these implications by itself), the wake-up would be a run that narrows nothing: one 8-byte record and one
            // domain read here instead.  wordpress7_500: 12 of a node's 25 implication runs were of that kind (r04 census).
            TB_REGION(67);
            const int2 cw = glob(P.cond2)[s * 64 + lane];
            if ((nar_all & 2) != 0 && cw.x >= 0) {
              const Itv I = load_ivar<C>(store, P.n_int, cw.x);
              if (I.lb == I.ub || I.ub < (cw.y & 0xffff) || I.lb > (int)((unsigned)cw.y >> 16)) nar_all &= ~2;
            }
          }
          TB_REGION(18);
#ifdef TB_TUNING
          note_marked(mark_successors<C>(P, sh, nxt, s, pr, sc, nar_all, (prof && wave == 0) ? sh.bs.dbg : nullptr));
#else
          note_marked(mark_successors<C>(P, sh, nxt, s, pr, sc, nar_all));
#endif
          TB_REGION(19);
          TB_PROF_MARK(2);
          TB_PROF_COUNT(4);
          if (pk(P) & 0x1) note_marked(mark_successors<C>(P, sh, nxt, s, pr, sc, nar_all));  // tuning: cost of the marks (idempotent)
          tc.writes += run_writes;
          {  // profiling (tuning build): 0x400000 counts slice runs instead of iterations; bits 28-31 = 1 + class to count only that class (11 = mixed slices)
            const int want = (knobs(P) >> 28) & 15;
            const unsigned cm = key & CLASS_SET_MASK;
            const int cls_of_slice = (cm & (cm - 1)) ? 10 : __builtin_ctz(cm | 0x400u);
            const bool useless = !wave_any(nar_all != 0);  // the run narrowed nothing
#ifdef TB_TUNING
            if (P.slice_census != nullptr && lane == 0) { atomicAdd(&glob(P.slice_census)[2 * s], 1u); if (useless) atomicAdd(&glob(P.slice_census)[2 * s + 1], 1u); }
#endif
            if (rep == reps_of(P, 3) && (want == 0 || want - 1 == cls_of_slice) && (!(knobs(P) & 0x40) || useless)) {  // 0x40: only the runs that narrowed nothing
              wave_iters_total += (pk(P) & 0x400000) ? 1u : wave_iters;
              wave_active_total += wave_iters * (unsigned)__builtin_popcountll(wave_ballot(act));
            }
          }
        }
#if TB_SC_PREFETCH
        s = s_next;
#else
        s = next_slice();  // (found after the run: nothing of the next slice lives through this one)
#endif
      }
    }
    for (int rep = reps_of(P, 10); rep > 0; --rep) {  // (tuning: the end of a round twice -- flag, reset, barrier)
    TB_REGION(20);
    if (tid == 0) {
      st(&sh.flag[(k + 1) % 3], 0);
      if ((rounds & 255) == 255 && deadline_passed(P)) st(&sh.abort, 1);
    }
    __syncthreads();  // the narrowings and the marks of this round are visible to everybody
    }
    TB_PROF_MARK(3);
    TB_PROF_COUNT(5);
    if (!ld(&sh.flag[k]) || dead_node(sh)) break;
  }
  TB_REGION(21);
  if (lane == 0) tc.writes += wave_writes;
  if (lane == 0 && wave_iters_total != 0) { add_deductions(sh, 64ull * wave_iters_total); add_active(sh, (unsigned long long)wave_active_total); }
  if (tid == 0 && prof) { const long long t = wall_clock64(); TB_PROF_ADD(sh.bs, TB_PROF_ROUNDS, t - tp0); tp0 = t; }  // profiling: rounds
  // leave both bitmaps empty for the next node (they are not after a failure)
  for (int rep = reps_of(P, 7); rep > 0; --rep)
  for (int i = tid; i < 2 * W; i += T) __hip_atomic_store(&es.dirty[i], 0u, TB_RLX, TB_WG);
  // ---- is every propagator entailed (the node is a solution, barebones:971-993)?
  // A byte says "some propagator of the slice was not entailed when the slice last ran".  With event filtering a slice may
  // sleep through a narrowing that cannot make it propagate but can make it entailed (`b1 <= b2` once b1 is false), so a
  // set byte is an upper bound.  One genuinely un-entailed propagator settles the question: the workgroup keeps such a
  // WITNESS (BlockShared::witness, a propagator index) and wave 0 re-evaluates just that one -- the common case, no scan of
  // the bytes at all.  When the witness has become entailed, the slices whose byte is set are examined one by one (their
  // byte is corrected on the way) until a new witness turns up or none is left.
  if (wave == 0 && !dead_node(sh)) {
    TB_REGION(22);
    auto unentailed_lanes = [&](int first_prop, bool whole_slice) -> unsigned long long {  // wave-uniform result
      const int i = whole_slice ? first_prop + lane : first_prop;
      const bool act = i < n;
      const int4 pr = props[TB_IDX(19, act ? i : 0, records)];
      const Itv X = load_dom<C>(store, P.n_int, pr.y), Y = load_dom<C>(store, P.n_int, pr.z), Z = load_dom<C>(store, P.n_int, pr.w);
      // (the witness is ONE record, the same in every lane: its own class as the slice's class set sends it down that class's body instead of every body behind selects)
      const Cand c = whole_slice ? evaluate_single<((C == 3 || C == 5) ? 1 : 0)>(pr.x, X, Y, Z)
                                 : evaluate_packed<((C == 3 || C == 5) ? 1 : 0)>((pr.x & 0xffff) | ((1 << (__builtin_amdgcn_readfirstlane(pr.x) & 0xf)) << 16), X, Y, Z);
      return wave_ballot(act && !c.ent);
    };
    int wit = __builtin_amdgcn_readfirstlane(ld(&sh.witness));
    if (reps_of(P, 4) > 1 && wit >= 0) { const unsigned long long again = unentailed_lanes(wit, false); asm volatile("" :: "s"(again)); }
    bool confirmed = wit >= 0 && unentailed_lanes(wit, false) != 0;
    for (int base = 0; !confirmed && base < W; base += 64) {
      const int wi = base + lane;
      const unsigned word = wi < W ? __hip_atomic_load(&ubits[wi], TB_RLX, TB_WG) : 0u;
      for (unsigned long long m = wave_ballot(word != 0); m && !confirmed; m &= m - 1) {
        const int wl = __builtin_ctzll(m);
        for (unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)word, wl); bits && !confirmed; bits &= bits - 1) {
          TB_REGION(23);
          const int sl = (base + wl) * 32 + __builtin_ctz(bits);
          const unsigned long long un = unentailed_lanes(sl * 64, true);
          if (un) { wit = sl * 64 + __builtin_ctzll(un); confirmed = true; }
          else if (lane == 0) (void)__hip_atomic_fetch_and(&ubits[sl >> 5], ~(1u << (sl & 31)), TB_RLX, TB_WG);  // every propagator of the slice is entailed now
        }
      }
    }
    if (lane == 0) { st(&sh.witness, confirmed ? wit : -1); st(&sh.unent[0], confirmed ? 1 : 0); }
  }
  __syncthreads();
  TB_REGION(24);
  all_entailed = !ld(&sh.unent[0]);
  return rounds + 1;
}

// ---- event-driven fixpoint of a workgroup TEAM (store layout 5; r06) ---------------------------------------------------------------------------
// The event fixpoint on the hot tier (layout 3: one 1024-thread workgroup per subproblem, an 800 KB store per workgroup of the synthetic 100k x 500k network) is the one kernel
// of this engine that is limited by the memory side: 256 stores do not fit the L2s (hit rate 0.43, 35 B of fabric reads per propagation).  Here the workgroups of a team search ONE
// subproblem on ONE store in global memory, as for the sweeps (solve_kernel_team), and share the ROUNDS of the event fixpoint:
//   * the two dirty bitmaps live in global memory (DevProblem::g_dirty, the slab of the team's leader), touched with agent-scope atomics like the store;
//   * slice s belongs to wave (s mod G) of the team's G = members x 16 waves (EventState::own: the bits of every bitmap word that are this wave's, built once per kernel in LDS),
//     which clears the bit before it loads a domain and is the only one to do so -- static ownership, as inside a workgroup;
//   * a round ends with ONE team barrier that merges "somebody marked a slice for the next round / failed / ran out of time": every member takes the same decision to go on;
//   * every member applies the same decisions to the shared store; whoever's atomic moved the bound notes the change in ITS list, and every member seeds from its own;
//   * the all-entailed test (witness) is evaluated by every member on the shared store -- the same propagator, the same answer.
// Runs are the generic ones (`apply`): a store in global memory keeps the caller's record order, whose slices hold several classes.  Same fixpoint, same tree.
__device__ __forceinline__ int fixpoint_event_team(const DevProblem& P, BlockShared& sh, int2* store, const int4* props, const EventState& es, ThreadCounters& tc, bool& all_entailed) {
  constexpr int C = 5;
  const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = T >> 6;
  const int M = team_size(sh), m = team_member(sh);
  const int n = P.n_props, W = es.words, S = P.n_slices;
  unsigned* const dirty = glob(es.dirty);
  unsigned* const ubits = glob(reinterpret_cast<unsigned*>(es.unent));  // bit s: some propagator of slice s was not entailed when it last ran (behind the team's store)
  const unsigned* const own = es.own + wave * W;
  const int cnt = ld(&sh.chg_count[0]);
  const bool root_pass = ld(&sh.ev_all) != 0;
  const bool all = root_pass || cnt > es.cap;
  const bool drop_entailed = !root_pass;
  // ---- initial dirty set (bitmap 0)
  if (all) {
    for (int i = m * T + tid; i < W; i += M * T) {
      const int left = S - i * 32;
      const unsigned v = left >= 32 ? 0xffffffffu : ((1u << left) - 1u);
      __hip_atomic_store(&dirty[i], v, TB_RLX, TB_AGENT);
      if (root_pass) __hip_atomic_store(&ubits[i], v, TB_RLX, TB_AGENT);
    }
  } else {
    // (every member applies the same decisions to the shared store, but only the one whose atomic actually moved a bound notes the change -- the others find it moved already:
    //  the members' change lists are SUBSETS whose union is complete, so every member seeds its whole list; a mark is an idempotent OR)
    for (int e0 = wave * 64; e0 < cnt; e0 += T) {
      const int e = e0 + lane;
      const int entry = e < cnt ? es.list[TB_IDX(20, e, chg_cap)] : 0;
      const int ev = e < cnt ? ((entry >> 30) & 3) : 0;
      int deg = 0, off = 0;
      bool did = false;
      const bool more = mark_var<TB_AGENT>(P, dirty, entry & 0x3fffffff, -1, ev, deg, off, did);
      const unsigned long long mm = wave_ballot(more);
      if (mm) (void)mark_tail<TB_AGENT>(P, dirty, mm, deg, off, ev, -1);
    }
  }
  if (tid == 0) { st(&sh.unent[0], 0); st(&sh.flag[0], 0); st(&sh.flag[1], 0); st(&sh.flag[2], 0); }
  (void)team_sync(P, sh, 0u);  // the seeds of every member are in the bitmap (every wave has waited for its own atomics)
  if (tid == 0) { st(&sh.ev_all, 0); st(&sh.chg_count[0], 0); }
  int rounds = 0;
  unsigned wave_iters_total = 0, wave_active_total = 0;
  for (;; ++rounds) {
    const int k = rounds % 3;
    unsigned* const cur = dirty + (rounds & 1) * W;
    unsigned* const nxt = dirty + ((rounds + 1) & 1) * W;
    bool marked = false;  // wave-uniform
    for (int base = 0; base < W; base += 64) {
      const int wi = base + lane;
      unsigned w = wi < W ? (__hip_atomic_load(&cur[wi], TB_RLX, TB_AGENT) & own[wi]) : 0u;
      if (w != 0) (void)__hip_atomic_fetch_and(&cur[wi], ~w, TB_RLX, TB_AGENT);  // mine, cleared before any domain is loaded
      if (drop_entailed && w != 0) w &= __hip_atomic_load(&ubits[wi], TB_RLX, TB_AGENT);  // entailed-slice removal
      unsigned long long nz = wave_ballot(w != 0);
      unsigned word = 0;
      int wl = 0;
      auto next_slice = [&]() -> int {
        while (word == 0) {
          if (nz == 0) return -1;
          wl = __builtin_ctzll(nz);
          nz &= nz - 1;
          word = (unsigned)__builtin_amdgcn_readlane((int)w, wl);
        }
        const int b = __builtin_ctz(word);
        word &= word - 1;
        return (base + wl) * 32 + b;
      };
      for (int s = next_slice(); s >= 0; s = next_slice()) {
        if (dead_node(sh)) break;  // the node failed in another wave of this member (the other members learn it at the round's barrier)
        s = TB_IDX(13, s, n_slices);
        const int4 pr = props[s * 64 + lane];  // (the arrays are padded to whole slices)
        const int4 sc = (es.succ + (size_t)s * 64)[lane];
        const bool act = s * 64 + lane < n;
        int nar_all = 0;
        unsigned iters = 0;
        for (;;) {  // the slice's local fixpoint (WAC1)
          bool ch = false, un_i = false;
          int nar = 0;
          apply<true, C>(pr, act, store, P.n_int, &sh.bot, ch, un_i, tc, 0, &nar);
          ++iters;
          nar_all |= nar;
          if (!wave_any(ch)) {
            if (!wave_any(un_i) && lane == 0) (void)__hip_atomic_fetch_and(&ubits[s >> 5], ~(1u << (s & 31)), TB_RLX, TB_AGENT);  // 1 -> 0 only: entailment is monotone below a node
            break;
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the narrowings are agent-scope atomics on the team's store: acknowledged before the next pass reads it)
          if (ld(&sh.bot)) break;
          if ((iters % WAVE_WATCHDOG_PERIOD) == 0) {
            if (lane == 0 && deadline_passed(P)) st(&sh.abort, 1);
            if (ld(&sh.abort)) break;
          }
        }
        marked |= mark_successors<C>(P, sh, nxt, s, pr, sc, nar_all);
        wave_iters_total += iters;
        wave_active_total += iters * (unsigned)__builtin_popcountll(wave_ballot(act));
      }
    }
    // ---- end of the round: one team barrier
    if (marked && lane == 0) st(&sh.flag[k], 1);
    if (tid == 0 && (rounds & 255) == 255 && deadline_passed(P)) st(&sh.abort, 1);
    __syncthreads();
    const unsigned mine = (ld(&sh.flag[k]) ? TEAM_CHANGED : 0u) | (ld(&sh.bot) ? TEAM_BOT : 0u) | (ld(&sh.abort) ? TEAM_ABORT : 0u);
    const unsigned agreed = team_sync(P, sh, mine);
    if (tid == 0) {
      st(&sh.flag[(k + 1) % 3], 0);
      if (agreed & TEAM_BOT) st(&sh.bot, 1);
      if (agreed & TEAM_ABORT) st(&sh.abort, 1);
    }
    __syncthreads();
    if (!(agreed & TEAM_CHANGED) || (agreed & (TEAM_BOT | TEAM_ABORT))) break;
  }
  if (lane == 0 && wave_iters_total != 0) { add_deductions(sh, 64ull * wave_iters_total); add_active(sh, (unsigned long long)wave_active_total); }
  // leave both bitmaps empty for the next node (they are not after a failure); nobody reads them after the last barrier, and the next barrier of the node orders these stores
  // before the next node's seeds
  for (int i = m * T + tid; i < 2 * W; i += M * T) __hip_atomic_store(&dirty[i], 0u, TB_RLX, TB_AGENT);
  // ---- is every propagator entailed?  (the witness of fixpoint_event, evaluated by every member on the shared store)
  if (wave == 0 && !dead_node(sh)) {
    auto unentailed_lanes = [&](int first_prop, bool whole_slice) -> unsigned long long {
      const int i = whole_slice ? first_prop + lane : first_prop;
      const bool act = i < n;
      const int4 pr = props[TB_IDX(19, act ? i : 0, records)];
      const Itv X = load_dom<C>(store, P.n_int, pr.y), Y = load_dom<C>(store, P.n_int, pr.z), Z = load_dom<C>(store, P.n_int, pr.w);
      const Cand c = whole_slice ? evaluate_single<1>(pr.x, X, Y, Z) : evaluate_packed<1>((pr.x & 0xffff) | ((1 << (__builtin_amdgcn_readfirstlane(pr.x) & 0xf)) << 16), X, Y, Z);
      return wave_ballot(act && !c.ent);
    };
    int wit = __builtin_amdgcn_readfirstlane(ld(&sh.witness));
    bool confirmed = wit >= 0 && unentailed_lanes(wit, false) != 0;
    for (int base = 0; !confirmed && base < W; base += 64) {
      const int wi = base + lane;
      const unsigned word = wi < W ? __hip_atomic_load(&ubits[wi], TB_RLX, TB_AGENT) : 0u;
      for (unsigned long long mk = wave_ballot(word != 0); mk && !confirmed; mk &= mk - 1) {
        const int wl = __builtin_ctzll(mk);
        for (unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)word, wl); bits && !confirmed; bits &= bits - 1) {
          const int sl = (base + wl) * 32 + __builtin_ctz(bits);
          const unsigned long long un = unentailed_lanes(sl * 64, true);
          if (un) { wit = sl * 64 + __builtin_ctzll(un); confirmed = true; }
          else if (lane == 0) (void)__hip_atomic_fetch_and(&ubits[sl >> 5], ~(1u << (sl & 31)), TB_RLX, TB_AGENT);  // every propagator of the slice is entailed now
        }
      }
    }
    if (lane == 0) { st(&sh.witness, confirmed ? wit : -1); st(&sh.unent[0], confirmed ? 1 : 0); }
  }
  __syncthreads();
  all_entailed = !ld(&sh.unent[0]);
  return rounds + 1;
}

// The event fixpoint as a function of its own: its registers are allocated for ITS loops, not for everything the persistent
// search loop keeps alive around it.
struct FpResult { int rounds; int all_entailed; unsigned writes; };
// SL: the store (and the entailment bits behind it) lives in LDS, `store_ref` is then its LDS offset; otherwise the store is the
// workgroup's slab in global memory and `gstore` points to it.
template <int C, int MEM, int TB = 0>
static __device__ TB_FIX_ATTR FpResult fixpoint_event_call(const DevProblem* Pp, unsigned sh_off, unsigned store_off, int2* gstore, unsigned props_off, const int4* gprops,
                                                     unsigned dirty_off, unsigned list_off, unsigned writes_in) {
  BlockShared& sh = *lds_ptr<BlockShared>(sh_off);
  int2* store = MEM >= TB_MEM_STORE_SHARED ? lds_ptr<int2>(store_off) : glob(gstore);
  const int4* props = MEM == TB_MEM_TCN_SHARED ? lds_ptr<const int4>(props_off) : glob(gprops);
  EventState es;
  es.dirty = lds_ptr<unsigned>(dirty_off); es.list = lds_ptr<int>(list_off);
  es.succ = glob(constant_problem(Pp).succ);  // (outlined variants keep the successor records in global memory)
  es.unent = reinterpret_cast<unsigned char*>(store) + constant_problem(Pp).unent_off;
  es.words = constant_problem(Pp).dirty_words; es.cap = constant_problem(Pp).chg_cap;
  ThreadCounters tc;
  tc.writes = writes_in;
  bool ae = false;
  FpResult r;
  r.rounds = fixpoint_event<C, TB>(constant_problem(Pp), sh, store, props, es, tc, ae);
  r.all_entailed = ae ? 1 : 0;
  r.writes = tc.writes;
  return r;
}

// ---- small helpers -------------------------------------------------------------------------------


// Block-wide copy of n intervals.  Both sides are 16-byte aligned (slabs are laid out in multiples of 2
// intervals), so the body moves 16 B per lane with four independent loads in flight: a snapshot of a
// 25k-variable store is ~12 memory round trips per thread instead of ~100.
// Block copies of a HOT store (layout 3): the first HOT_VARS intervals of the working store live in LDS.  HS: the source is the working store,
// HD: the destination is.
template <int TB, bool HS, bool HD>
__device__ __forceinline__ void copy_store_hot(int2* dst, const int2* src, int n) {
  const int T = block_threads<TB>(), tid = here(threadIdx.x);
  int4* d4 = reinterpret_cast<int4*>(dst);
  const int4* s4 = reinterpret_cast<const int4*>(src);
  int4* h4 = reinterpret_cast<int4*>(lds_ptr<int2>((unsigned)SH_BYTES));
  const int n4 = n >> 1;  // (slabs are laid out in multiples of two intervals)
  for (int i = tid; i < n4; i += T) {
    const bool hot = i < HOT_VARS / 2;
    const int4 v = (HS && hot) ? h4[i] : s4[i];
    if (HD && hot) h4[i] = v; else d4[i] = v;
  }
}
template <int TB>
__device__ __forceinline__ void copy_store(int2* dst, const int2* src, int n);
// Layout 5 (workgroup teams): copies between the team's store and a private slab, rows [lo, hi).  The team store is READ with agent-scope loads (another CU narrowed it: this CU's
// L1 may hold stale lines) and WRITTEN with agent-scope stores (they land in the XCD's L2, where the members' agent-scope loads find them).
__device__ __forceinline__ void team_copy_out(int2* dst, const int2* store, int lo, int hi) {
  for (int i = lo + (int)threadIdx.x; i < hi; i += (int)blockDim.x) {
    const long long raw = __hip_atomic_load(reinterpret_cast<const long long*>(store + i), TB_RLX, TB_AGENT);
    dst[i] = make_int2((int)(raw & 0xffffffffll), (int)(raw >> 32));
  }
}
__device__ __forceinline__ void team_copy_in(int2* store, const int2* src, int lo, int hi) {
  for (int i = lo + (int)threadIdx.x; i < hi; i += (int)blockDim.x) {
    const int2 d = src[i];
    __hip_atomic_store(reinterpret_cast<long long*>(store + i), (long long)(((unsigned long long)(unsigned)d.y << 32) | (unsigned long long)(unsigned)d.x), TB_RLX, TB_AGENT);
  }
}
// the working store of layout C written to / filled from a slab in global memory
template <int C, int TB>
__device__ __forceinline__ void store_out(int2* dst, const int2* store, int n) {
  if constexpr (C == 5) team_copy_out(dst, store, 0, n);
  else if constexpr (C == 3) copy_store_hot<TB, true, false>(dst, store, n); else copy_store<TB>(dst, store, n);
}
template <int C, int TB>
__device__ __forceinline__ void store_in(int2* store, const int2* src, int n) {
  if constexpr (C == 5) team_copy_in(store, src, 0, n);
  else if constexpr (C == 3) copy_store_hot<TB, false, true>(store, src, n); else copy_store<TB>(store, src, n);
}
template <int TB = 0>
__device__ __forceinline__ void copy_store(int2* dst, const int2* src, int n) {
  const int T = block_threads<TB>(), tid = here(threadIdx.x);  // (per-lane addresses computed here, at each copy)
  if ((reinterpret_cast<size_t>(dst) | reinterpret_cast<size_t>(src)) & 15) {  // odd-sized caller buffers (best store, tb_propagate batches)
    for (int i = tid; i < n; i += T) dst[i] = src[i];
    return;
  }
  const int n4 = n >> 1;
  int4* d4 = reinterpret_cast<int4*>(dst);
  const int4* s4 = reinterpret_cast<const int4*>(src);
  int i = tid;
  for (; i + 3 * T < n4; i += 4 * T) {
    const int4 a = s4[i], b = s4[i + T], c = s4[i + 2 * T], d = s4[i + 3 * T];
    d4[i] = a; d4[i + T] = b; d4[i + 2 * T] = c; d4[i + 3 * T] = d;
  }
  for (; i < n4; i += T) d4[i] = s4[i];
  if ((n & 1) && tid == 0) dst[n - 1] = src[n - 1];
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
  for (int off = 32; off > 0; off >>= 1) {
    unsigned long long o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
  for (int off = 32; off > 0; off >>= 1) {
    int o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}

// Thread 0 only: VStore::embed of one interval (decisions, objective bound).
template <int C>
__device__ __forceinline__ int embed0(int2* store, int ni, int* bot, int v, int lb, int ub) {  // returns EV_LB / EV_UB bits
  Itv d = load_dom<C>(store, ni, v);
  int ev = 0;
  if (lb > d.lb) { raise_lb<C>(store, ni, v, lb); d.lb = lb; ev |= EV_LB; }
  if (ub < d.ub) { lower_ub<C>(store, ni, v, ub); d.ub = ub; ev |= EV_UB; }
  if (d.lb > d.ub) st(bot, 1);
  return ev;
}
// embed0 + event bookkeeping: the slices reading v must run in the first sweep of the next fixpoint
template <bool EVENT, int C>
__device__ __forceinline__ void embed0_mark(const DevProblem& P, BlockShared& sh, const EventState& es, int2* store, int* bot, int v, int lb, int ub) {
  const int ev = embed0<C>(store, P.n_int, bot, v, lb, ub);
  if (EVENT && ev) note_change(sh, es, var_of<C>(v), ev);
}

// Key to MINIMISE for each variable order (barebones:193-221); ties resolve to the lowest index
// because the index sits in the low half of the packed 64-bit key (barebones:322-338).
__device__ __forceinline__ unsigned order_key(int var_order, const Itv d) {
  switch (var_order) {
    case TB_FIRST_FAIL: return (unsigned)d.ub - (unsigned)d.lb;
    case TB_ANTI_FIRST_FAIL: return ~((unsigned)d.ub - (unsigned)d.lb);
    case TB_SMALLEST: return (unsigned)d.lb ^ 0x80000000u;
    case TB_LARGEST: return ~((unsigned)d.ub ^ 0x80000000u);
    default: return 0u;  // input order
  }
}

// Entry i of a workgroup's decision stack: segment 0 is the workgroup's slab, the others were taken from the pool when the
// search went deeper (the reference reallocates its vector, barebones:401-403; here a segment is 16 384 decisions by default).
__device__ __forceinline__ Decision& dec_at(const DevProblem& P, BlockShared& sh, Decision* dec, int i) {
  const int k = i >> P.max_depth_log2;
  (void)TB_IDX_N(15, i, (1 + sh.n_dec_seg) << P.max_depth_log2);
  return k == 0 ? dec[i] : sh.dec_seg[k - 1][i & (P.max_depth - 1)];
}

// Thread 0 only (barebones:355-405).
template <int C>
__device__ __forceinline__ bool push_decision(const DevProblem& P, BlockShared& sh, Decision* dec, const int2* store, int val_order, int var) {
  const int depth = sh.depth;
  if (depth + 1 >= ((1 + sh.n_dec_seg) << P.max_depth_log2)) {  // grow: one more segment from the pool
    bool grown = false;
    if (sh.n_dec_seg < MAX_DEC_SEGS && P.dec_pool_segments > 0) {
      const int idx = __hip_atomic_fetch_add(&glob(P.ctrl)->dec_pool_next, 1, TB_RLX, TB_AGENT);
      if (idx < P.dec_pool_segments) { sh.dec_seg[sh.n_dec_seg++] = P.dec_pool + (size_t)idx * (size_t)P.max_depth; grown = true; }
    }
    if (!grown) { __hip_atomic_store(&glob(P.ctrl)->error, 1, TB_RLX, TB_AGENT); return false; }
  }
  Decision d;
  const Itv dom = load_dom<C>(store, P.n_int, var);
  d.var = var;
  d.cur = -1;
  const int mid = (int)((long long)dom.lb + ((long long)dom.ub - (long long)dom.lb) / 2);
  switch (val_order) {
    case TB_VAL_MIN: d.child[0] = make_int2(dom.lb, dom.lb); d.child[1] = make_int2(dom.lb + 1, dom.ub); break;
    case TB_VAL_MAX: d.child[0] = make_int2(dom.ub, dom.ub); d.child[1] = make_int2(dom.lb, dom.ub - 1); break;
    case TB_VAL_SPLIT: d.child[0] = make_int2(dom.lb, mid); d.child[1] = make_int2(mid + 1, dom.ub); break;
    default: d.child[0] = make_int2(mid + 1, dom.ub); d.child[1] = make_int2(dom.lb, mid); break;
  }
  d.rope[0] = depth + 1;
  if (depth > 0) { const Decision& up = dec_at(P, sh, dec, depth - 1); d.rope[1] = up.rope[up.cur]; }
  else d.rope[1] = -1;
  dec_at(P, sh, dec, depth) = d;
#ifdef TB_TRAP_SEED
  if (var < 0) { const int w[10] = {d.var, d.cur, d.child[0].x, d.child[0].y, d.child[1].x, d.child[1].y, d.rope[0], d.rope[1], depth, val_order}; trap_report(P, sh, 1, w, 10); }
#endif
  sh.depth = depth + 1;
  return true;
}

// Block-parallel variable selection.  Replaces the three-barrier lattice fixpoint loops of
// input_order_split / lattice_smallest_split (barebones:240-349) by one strided scan, a wave-level
// min reduction (DPP/bpermute shuffles) and one LDS round per strategy.
// Ends with a barrier; sh.found tells whether a decision was pushed at sh.depth-1.
template <int C, int TB = 0>
__device__ __forceinline__ void split(const DevProblem& P, BlockShared& sh, Decision* dec, const int2* store) {
  const int tid = here(threadIdx.x), T = block_threads<TB>(), lane = tid & 63, wave = tid >> 6, nw = T >> 6;
  for (;;) {
    TB_REGION(33);
    const int s = sh.cur_strategy;  // uniform: read after a barrier
    if (s >= P.n_strats) { if (tid == 0) sh.found = 0; __syncthreads(); return; }
    const int off = glob(P.strat_off)[TB_IDX(16, s, strats)];
    int n = glob(P.strat_off)[s + 1] - off;
    if (n > 0) (void)TB_IDX_N(16, off + n - 1, g_bl_total());
    const bool in_store = (n == 0);
    if (in_store) n = P.n_vars;
    const int vo = glob(P.strat_var_order)[s];
    unsigned long long best = ~0ull;
    int first = n;
    for (int rep = reps_of(P, 5); rep > 0; --rep) {
    best = ~0ull; first = n;
    // four candidates per thread and round: the index gathers, then the domain gathers, are issued together
    // (a store in global memory costs one L2 round trip per dependent load, not per variable)
    for (int i0 = sh.next_unassigned + tid; i0 < n; i0 += 4 * T) {
      int v[4];
      Itv d[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * T;
        v[k] = i < n ? (in_store ? i : glob(P.strat_vars)[off + i]) : -1;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        d[k] = Itv{0, 0};
        if (v[k] >= 0) d[k] = load_dom<C>(store, P.n_int, v[k]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * T;
        if (d[k].lb != d[k].ub && !is_inf(d[k].lb) && !is_inf(d[k].ub)) {
          const unsigned long long key = ((unsigned long long)order_key(vo, d[k]) << 32) | (unsigned)i;
          best = key < best ? key : best;
          first = i < first ? i : first;
        }
      }
    }
    best = wave_min_u64(best);
    first = wave_min_i32(first);
    if (lane == 0) { sh.red_key[wave] = best; sh.red_first[wave] = first; }
    __syncthreads();
    if (rep > 1) __syncthreads();  // (tuning: the scan is about to run again and rewrite red_key)
    }
    if (tid == 0) {
      TB_REGION(34);
      for (int w = 1; w < nw; ++w) {
        best = sh.red_key[w] < best ? sh.red_key[w] : best;
        first = sh.red_first[w] < first ? sh.red_first[w] : first;
      }
      sh.next_unassigned = first;
      if (best != ~0ull) {
        const int i = (int)(best & 0xffffffffu);
        const int v = in_store ? i : glob(P.strat_vars)[off + i];
        sh.found = push_decision<C>(P, sh, dec, store, glob(P.strat_val_order)[s], v) ? 1 : 0;
        if (!sh.found) sh.stop = 1;
        sh.skip = 1;  // leave the strategy loop
      } else {
        sh.cur_strategy = s + 1;
        sh.next_unassigned = 0;
        sh.skip = 0;
      }
    }
    TB_REGION(58);
    __syncthreads();
    if (sh.skip) return;
  }
}

// Streaming: hand the solution in `store` to the host through the ring (GridData::produce_solution,
// gpu_dive_and_solve.hpp:100-114, without the print lock: a ticket orders the producers, the host consumes in
// ticket order).  Uniform call; sh.ticket was taken by thread 0.
template <int TB = 0, int C = 0>
__device__ __forceinline__ void produce_solution(const DevProblem& P, BlockShared& sh, const int2* store, Mailbox* mbox) {
  const int tid = threadIdx.x;
  const unsigned long long ticket = (unsigned long long)sh.ticket;
  const SolutionRing& r = P.ring;
  if (tid == 0) {
    // wait for a free slot; a stop request (host or device) drops the solution: nobody is listening any more
    while (ticket - __hip_atomic_load(r.consumed, __ATOMIC_ACQUIRE, TB_SYS) >= (unsigned long long)r.slots) {
      if (__hip_atomic_load(&mbox->stop, TB_RLX, TB_SYS) != 0 || __hip_atomic_load(&glob(P.ctrl)->stop, TB_RLX, TB_AGENT) != 0 ||
          (P.deadline_ticks != 0 && wall_clock64() > P.deadline_ticks)) { sh.ticket = -1; break; }
      __builtin_amdgcn_s_sleep(64);
    }
  }
  __syncthreads();
  if (sh.ticket >= 0) {
    const int slot = (int)(ticket % (unsigned long long)r.slots);
    store_out<C, TB>(r.data + (size_t)slot * P.vext, store, P.vext);
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&r.seq[slot], ticket + 1ull, __ATOMIC_RELEASE, TB_SYS);
  }
  __syncthreads();
}

// ---- device <-> host / device <-> device words (thread 0 only) ---------------------------------------

__device__ __forceinline__ void raise_gpu_stop(const DevProblem& P);

struct Hot { int best, foreign, stop; unsigned next_poll; };
// The 16 hot bytes of Ctrl: two 8-byte agent-scope loads (they bypass the non-coherent vector L1), one wait.
__device__ __forceinline__ Hot load_hot(const Ctrl* c) {
  const unsigned long long a = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&c->best_bound), TB_RLX, TB_AGENT);
  const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&c->stop), TB_RLX, TB_AGENT);
  Hot h;
  h.best = (int)(a & 0xffffffffull); h.foreign = (int)(a >> 32);
  h.stop = (int)(b & 0xffffffffull); h.next_poll = (unsigned)(b >> 32);
  return h;
}

// One poll of everything that lives outside this device: the host mailbox (stop request, relayed incumbent) and this
// device's peer cell (incumbent and stop raised by the other GPUs over xGMI).  Time based: whoever notices that the
// poll is due and wins the CAS on Ctrl::next_poll does it, every other workgroup goes on.
__device__ __forceinline__ void poll_outside(const DevProblem& P, Mailbox* mbox, long long now) {
  Ctrl* c = glob(P.ctrl);
  PeerCell* me = glob(P.cell);
  int stop = __hip_atomic_load(&mbox->stop, TB_RLX, TB_SYS) != 0 ? STOP_HOST : 0;
  int fb = __hip_atomic_load(&mbox->foreign_bound, TB_RLX, TB_SYS);
  const int cb = __hip_atomic_load(&me->bound, TB_RLX, TB_SYS);
  fb = cb < fb ? cb : fb;
  if (__hip_atomic_load(&me->stop, TB_RLX, TB_SYS) != 0) stop |= STOP_GPU;
  if (P.deadline_ticks != 0 && now > P.deadline_ticks) stop |= STOP_HOST;
  if (fb != PINF) (void)__hip_atomic_fetch_min(&c->foreign_bound, fb, TB_RLX, TB_AGENT);
  if (stop) (void)__hip_atomic_fetch_or(&c->stop, stop, TB_RLX, TB_AGENT);
  if (P.cut_nodes_total != 0 && P.peers != nullptr) {
    // node budget of the linked group: fold what this device explored since the last poll into rank 0's cell -- one system-scope
    // atomic per device and poll period.  (fetch_max: two pollers overlapping in time fold disjoint parts.)
    const unsigned long long mine = __hip_atomic_load(&c->nodes_local, TB_RLX, TB_AGENT);
    const unsigned long long old = __hip_atomic_fetch_max(&c->nodes_folded, mine, TB_RLX, TB_AGENT);
    PeerCell* root = glob(P.peers)[0] != nullptr ? glob(glob(P.peers)[0]) : me;
    unsigned long long total;
    if (mine > old) total = __hip_atomic_fetch_add(&root->nodes_total, mine - old, TB_RLX, TB_SYS) + (mine - old);
    else total = __hip_atomic_load(&root->nodes_total, TB_RLX, TB_SYS);
    if (total >= P.cut_nodes_total) raise_gpu_stop(P);
  }
  // device -> host: the incumbent (two improvements may reach the host out of order; the refresh repairs it within
  // one period) and the remaining work
  __hip_atomic_store(&mbox->local_best, __hip_atomic_load(&c->best_bound, TB_RLX, TB_AGENT), TB_RLX, TB_SYS);
  __hip_atomic_store(&mbox->progress, __hip_atomic_load(&me->queue, TB_RLX, TB_SYS), TB_RLX, TB_SYS);
  __hip_atomic_store(&mbox->polls, __hip_atomic_load(&mbox->polls, TB_RLX, TB_SYS) + 1, TB_RLX, TB_SYS);
}
__device__ __forceinline__ void maybe_poll(const DevProblem& P, Mailbox* mbox, const Hot& h, long long now) {
  if ((int)((unsigned)now - h.next_poll) >= 0) {
    unsigned expect = h.next_poll;
    if (__hip_atomic_compare_exchange_strong(&glob(P.ctrl)->next_poll, &expect, (unsigned)now + (unsigned)P.poll_ticks, TB_RLX, TB_RLX, TB_AGENT))
      poll_outside(P, mbox, now);
  }
}

// A better incumbent was found on this device: tell the host and every other GPU (4 bytes per peer over xGMI -- the
// only payload the GPUs exchange during the search, barebones:426 made multi-device).
__device__ __forceinline__ void publish_bound(const DevProblem& P, Mailbox* mbox, int obj) {
  __hip_atomic_store(&mbox->local_best, obj, TB_RLX, TB_SYS);
  if (P.peers != nullptr)
    for (int r = 0; r < P.world; ++r) {
      PeerCell* pc = glob(glob(P.peers)[r]);
      if (r != P.rank && pc != nullptr) (void)__hip_atomic_fetch_min(&pc->bound, obj, TB_RLX, TB_SYS);
    }
}
// Solution limit reached / objective unbounded: every GPU stops.
__device__ __forceinline__ void raise_gpu_stop(const DevProblem& P) {
  (void)__hip_atomic_fetch_or(&glob(P.ctrl)->stop, STOP_GPU, TB_RLX, TB_AGENT);
  if (P.peers != nullptr)
    for (int r = 0; r < P.world; ++r) {
      PeerCell* pc = glob(glob(P.peers)[r]);
      if (r != P.rank && pc != nullptr) __hip_atomic_store(&pc->stop, 1, TB_RLX, TB_SYS);
    }
}

// ---- work queue over the EPS index space (barebones:718-741,877-884 made multi-device; thread 0 only) -----------------
//
// Each device owns a block-cyclic share of the 2^d subproblems (eps_global_index) and serves it through its cell's
// `queue` word.  A device whose queue runs dry takes the upper half of the fullest peer queue with one CAS over xGMI
// and installs it as its own next range: dynamic balance without the host, like the single grid-wide counter of the
// reference but hierarchical.  Every subproblem index is, at any time, in exactly one queue range or in the hands of
// exactly one workgroup.

// Take the upper half of the fullest peer queue.  1: a range was installed, 0: nothing now (try again), -1: every
// queue of the node is empty and nobody is moving work -- the search is over for this workgroup.
// One workgroup of a device at a time (Ctrl::steal_lock, device-local).  PeerCell::stealing is raised only WHILE A RANGE IS IN
// TRANSIT -- from just before the CAS that takes it out of the victim's queue until it is published in the thief's -- not while a
// device merely looks around: idle devices scanning each other at the end of the search never see one another as busy.
__device__ __forceinline__ int steal_work(const DevProblem& P, BlockStats& bs) {
  PeerCell* me = glob(P.cell);
  int zero = 0;
  if (!__hip_atomic_compare_exchange_strong(&glob(P.ctrl)->steal_lock, &zero, 1, __ATOMIC_ACQUIRE, TB_RLX, TB_AGENT)) return 0;  // a sibling is at it
  const unsigned long long mine = __hip_atomic_load(&me->queue, __ATOMIC_ACQUIRE, TB_SYS);
  if (q_next(mine) < q_hi(mine)) { __hip_atomic_store(&glob(P.ctrl)->steal_lock, 0, __ATOMIC_RELEASE, TB_AGENT); return 1; }  // refilled meanwhile
  // Scan the peers: in-transit flags, then queues, then the flags again.  A peer that moves a range -- raises its flag, empties a
  // third GPU's queue, publishes the range in its own, lowers the flag -- can slip between the reads of ONE pass (its own queue
  // read before the range was published, its flag after it was lowered); "the node is out of work" therefore needs two
  // consecutive passes that saw nothing.
  int best = -1;
  unsigned long long best_av = 0;
  bool busy = false;
  for (int pass = 0; pass < 2 && !busy; ++pass) {
    for (int r = 0; r < P.world; ++r) {
      PeerCell* pc = glob(glob(P.peers)[r]);
      if (r != P.rank && pc != nullptr && __hip_atomic_load(&pc->stealing, __ATOMIC_ACQUIRE, TB_SYS) != 0) busy = true;
    }
    for (int r = 0; r < P.world; ++r) {
      PeerCell* pc = glob(glob(P.peers)[r]);
      if (r == P.rank || pc == nullptr) continue;
      const unsigned long long w = __hip_atomic_load(&pc->queue, __ATOMIC_ACQUIRE, TB_SYS);
      const unsigned long long av = q_hi(w) > q_next(w) ? q_hi(w) - q_next(w) : 0ull;
      if (av > best_av) { best = r; best_av = av; }
      if (av > 0) busy = true;
    }
    for (int r = 0; r < P.world; ++r) {
      PeerCell* pc = glob(glob(P.peers)[r]);
      if (r != P.rank && pc != nullptr && __hip_atomic_load(&pc->stealing, __ATOMIC_ACQUIRE, TB_SYS) != 0) busy = true;
    }
  }
  int result = busy ? 0 : -1;
  if (best >= 0) {
    PeerCell* v = glob(glob(P.peers)[best]);
    result = 0;
    __hip_atomic_store(&me->stealing, 1, __ATOMIC_RELEASE, TB_SYS);  // visible before the victim's queue shrinks
    for (int tries = 0; tries < 8; ++tries) {
      unsigned long long w = __hip_atomic_load(&v->queue, __ATOMIC_ACQUIRE, TB_SYS);
      const unsigned long long nx = q_next(w), hi = q_hi(w);
      if (hi <= nx) break;
      const unsigned long long take = (hi - nx + 1) / 2;  // the last one too: the victim's kernel may not be running (yet, or any more)
      const unsigned g = q_gen(w);
      // the descriptor of generation g is stable while its range is not exhausted, and the CAS only succeeds on that range
      const unsigned long long vbase = __hip_atomic_load(&v->desc[g & 7].j_base, TB_RLX, TB_SYS);
      const int vowner = __hip_atomic_load(&v->desc[g & 7].owner, TB_RLX, TB_SYS);
      if (!__hip_atomic_compare_exchange_strong(&v->queue, &w, q_pack(g, nx, hi - take), __ATOMIC_ACQ_REL, TB_RLX, TB_SYS)) continue;
      // [hi - take, hi) of the victim's range is mine now
      const unsigned g2 = (q_gen(mine) + 1u) & 0xffu;
      __hip_atomic_store(&me->desc[g2 & 7].j_base, vbase + (hi - take), TB_RLX, TB_SYS);
      __hip_atomic_store(&me->desc[g2 & 7].owner, vowner, TB_RLX, TB_SYS);
      __hip_atomic_store(&me->queue, q_pack(g2, 0, take), __ATOMIC_RELEASE, TB_SYS);
      (void)__hip_atomic_fetch_add(&me->stolen_in, take, TB_RLX, TB_SYS);
      (void)__hip_atomic_fetch_add(&v->stolen_out, take, TB_RLX, TB_SYS);
      bs.stolen += take;
      result = 1;
      break;
    }
    __hip_atomic_store(&me->stealing, 0, __ATOMIC_RELEASE, TB_SYS);
  }
  __hip_atomic_store(&glob(P.ctrl)->steal_lock, 0, __ATOMIC_RELEASE, TB_AGENT);
  return result;
}

// Fetch the next subproblem of this device (sh.sub_idx / sub_j / sub_owner / sub_gen).  False: there is no work left
// anywhere (or a stop was requested while waiting for some).
__device__ __forceinline__ bool next_subproblem(const DevProblem& P, BlockShared& sh, Mailbox* mbox) {
  PeerCell* me = glob(P.cell);
  long long t_wait = 0;
  bool got = false;
  for (;;) {
    // (look before adding: a waiting workgroup must not push `next` towards the end of its 28-bit field)
    const unsigned long long seen = __hip_atomic_load(&me->queue, TB_RLX, TB_SYS);
    if (q_next(seen) < q_hi(seen)) {
      const unsigned long long old = __hip_atomic_fetch_add(&me->queue, 1ull << Q_BITS, __ATOMIC_ACQUIRE, TB_SYS);
      const unsigned long long nx = q_next(old), hi = q_hi(old);
      if (nx < hi) {
        const unsigned g = q_gen(old);
        const unsigned long long base = __hip_atomic_load(&me->desc[g & 7].j_base, TB_RLX, TB_SYS);
        const int owner = __hip_atomic_load(&me->desc[g & 7].owner, TB_RLX, TB_SYS);
        sh.sub_j = base + nx; sh.sub_owner = owner; sh.sub_gen = (int)g;
        sh.sub_idx = eps_global_index(base + nx, P.chunk_log2, owner, P.world);
        got = true;
        break;
      }
    }
    if (P.world <= 1 || P.peers == nullptr || !P.steal) break;
    const long long now = wall_clock64();
    if (t_wait == 0) { t_wait = now; (void)__hip_atomic_fetch_add(&me->waiting, 1, TB_RLX, TB_SYS); }
    const int r = steal_work(P, sh.bs);
    if (r < 0) break;
    if (r > 0) continue;
    const Hot h = load_hot(glob(P.ctrl));
    if (h.stop != 0) break;
    maybe_poll(P, mbox, h, now);
    __builtin_amdgcn_s_sleep(127);
  }
  if (t_wait != 0) { sh.bs.wait_ticks += wall_clock64() - t_wait; (void)__hip_atomic_fetch_add(&me->waiting, -1, TB_RLX, TB_SYS); }
  return got;
}

// A leaf was met `remaining` levels above the subproblem: every index of its subtree [s, s + 2^remaining) shares it
// (barebones:718-741).  Jump this device's queue over the part of the subtree that belongs to the range the index came
// from.  Accounting: the index itself counts as skipped, and so does every index the jump removes from the queue --
// each subproblem is counted exactly once (solved or skipped) however the ranges were cut and moved between the GPUs;
// with one workgroup this is the reference's `next_idx - idx`.
__device__ __forceinline__ void skip_subtree(const DevProblem& P, BlockShared& sh) {
  PeerCell* me = glob(P.cell);
  const int r = sh.remaining;
  const unsigned long long e = ((sh.sub_idx >> r) + 1ull) << r;  // first global index behind the subtree
  const unsigned long long je = eps_local_lower_bound(e, P.chunk_log2, sh.sub_owner, P.world);
  unsigned long long jumped = 0;
  for (;;) {
    unsigned long long w = __hip_atomic_load(&me->queue, __ATOMIC_ACQUIRE, TB_SYS);
    if ((int)q_gen(w) != sh.sub_gen) break;  // that range is exhausted: nothing left to jump over
    const unsigned long long base = __hip_atomic_load(&me->desc[sh.sub_gen & 7].j_base, TB_RLX, TB_SYS);
    const unsigned long long nx = q_next(w), hi = q_hi(w);
    unsigned long long target = je - base;
    target = target < hi ? target : hi;
    if (nx >= target) break;
    if (__hip_atomic_compare_exchange_strong(&me->queue, &w, q_pack((unsigned)sh.sub_gen, target, hi), __ATOMIC_ACQ_REL, TB_RLX, TB_SYS)) { jumped = target - nx; break; }
  }
  sh.bs.eps_skipped += 1ull + jumped;
}

// ---- one search node (barebones:903-1031) ---------------------------------------------------------

struct NodeTimers { long long t_last; };

template <bool EVENT, int C, bool RM, int MEM, int TB = 0, bool DEEP = false>
__device__ __forceinline__ void propagate_node_impl(const DevProblem& P, BlockShared& sh, int2* store, const int4* props, const EventState& es,
                                                    int2* best_store, Mailbox* mbox, ThreadCounters& tc) {
  const int tid = threadIdx.x;
  BlockStats& bs = sh.bs;
  // layout 5 (workgroup teams): every member runs this function on the same store and reaches the same leaf / solution verdict; what touches the grid words, the host or
  // the statistics is the LEADER's business, and the leader's stop decision is what every member follows (merged by one team barrier below)
  const bool lead = C != 5 || team_member(sh) == 0;
  TB_REGION(1);
  // (thread 0's clock at the phase boundary lives in LDS, sh.t_mark: a register pair held through the fixpoint ended up in scratch)
  if (tid == 0) { const long long t0 = wall_clock64(); bs.timers[TB_T_SEARCH] += t0 - sh.t_mark; sh.t_mark = t0; }
  bool all_entailed = false;
  int iters;
  if constexpr (EVENT && C == 5) {
    iters = fixpoint_event_team(P, sh, store, props, es, tc, all_entailed);
  } else if constexpr (EVENT) {
    constexpr bool SL = MEM >= TB_MEM_STORE_SHARED, PL = MEM == TB_MEM_TCN_SHARED;
    const FpResult r = fixpoint_event_call<C, MEM, TB>(&P, lds_off(&sh), SL ? lds_off(store) : 0u, SL ? nullptr : store, PL ? lds_off(props) : 0u, PL ? nullptr : props,
                                                   lds_off(es.dirty), lds_off(es.list), tc.writes);
    iters = r.rounds; all_entailed = r.all_entailed != 0; tc.writes = r.writes;
  } else iters = fixpoint<RM, C, DEEP>(P, sh, store, props, es.unent, tc, all_entailed);
  flush_writes(sh, tc, false);
  const bool aborted = ld(&sh.abort) != 0;
  const bool failed = !aborted && ld(&sh.bot) != 0;
  if (aborted) all_entailed = false;
#ifdef TB_TUNING
  // Self-check of the event-driven fixpoint (tuning build, tb_config.reserved[0] & 0x1000000): after a node that did not fail,
  // every propagator is evaluated once more with the generic rules, without writing.  BlockStats::why bit 8: some propagator
  // could still narrow (a wake-up was missed); bit 9: a slice is flagged all-entailed but one of its propagators is not.
  // pad_why keeps 1 + the first offending slice.
  if (EVENT && (knobs(P) & 0x1000000) && !failed && !aborted) {
    const int lane = threadIdx.x & 63;
    for (int s = threadIdx.x >> 6; s < P.n_slices; s += block_threads<TB>() >> 6) {
      const int i = s * 64 + lane;
      const bool act = lane < glob(P.slice_real)[s];  // (idle padding at the end of a class is not a propagator)
      const int4 pr = props[i];
      const Itv X = load_dom<C>(store, P.n_int, pr.y), Y = load_dom<C>(store, P.n_int, pr.z), Z = load_dom<C>(store, P.n_int, pr.w);
      const Cand c = evaluate_packed(pr.x, X, Y, Z);
      const bool narrows = act && ((c.xl > X.lb) | (c.xu < X.ub) | (c.yl > Y.lb) | (c.yu < Y.ub) | (c.zl > Z.lb) | (c.zu < Z.ub));
      const bool un = act && !c.ent;
      int bad = 0;
      if (wave_any(narrows)) bad |= 1 << 8;
      if (wave_any(un) && !((reinterpret_cast<const unsigned*>(es.unent)[s >> 5] >> (s & 31)) & 1u)) bad |= 1 << 9;
      if (bad) {
        int zero = 0;
        bool first = false;
        if (lane == 0) {
          (void)__hip_atomic_fetch_or(&sh.bs.why, bad, TB_RLX, TB_WG);
          first = __hip_atomic_compare_exchange_strong(&sh.bs.pad_why, &zero, s + 1, TB_RLX, TB_RLX, TB_WG);
        }
        first = __builtin_amdgcn_readfirstlane((int)first) != 0;
        const unsigned long long m = wave_ballot(narrows);
        if (first && m && lane == __builtin_ctzll(m)) {
          int* d = sh.bs.dbg;
          d[0] = lane; d[1] = pr.x; d[2] = pr.y; d[3] = pr.z; d[4] = pr.w; d[5] = X.lb; d[6] = X.ub; d[7] = Y.lb; d[8] = Y.ub; d[9] = Z.lb; d[10] = Z.ub; d[11] = (int)sh.bs.nodes;
        }
      }
    }
    __syncthreads();
  }
#endif
  // The leaf rule of the reference's `gpu` and `cpu` paths (gpu_dive_and_solve.hpp:333-338, cpu_solving.hpp:33-40): "no propagator is active" makes a node a
  // solution only if the store is extractable as well -- every variable assigned (hybrid_dive_and_solve.hpp:531); otherwise it is an inner node and the
  // search keeps branching below it.  barebones accepts the box (barebones:988-993).  Only all-entailed nodes pay for the scan: one pass over the slab.
  if (P.leaf_assign && !failed && all_entailed) {  // uniform
    bool open = false;
    for (int v = tid; v < P.n_vars; v += block_threads<TB>()) {  // (P.n_vars: the variables of the slab; constants kept out of it are assigned)
      (void)TB_IDX(21, v, slab_vars);
      const Itv d = load_dom<C>(store, P.n_int, C == 4 ? (v | (C8_BASE_BIAS << 16)) : v);  // (COMPACT8: any base will do for "lb == ub")
      open |= d.lb != d.ub;
    }
    if (wave_any(open) && (tid & 63) == 0) st(&sh.open_vars, 1);
    __syncthreads();
  }
  if (tid == 0) {
    TB_REGION(25);
    const long long t1 = wall_clock64();
    bs.timers[TB_T_FIXPOINT] += t1 - sh.t_mark;
    sh.t_mark = t1;
    int leaf = failed ? 1 : 0, sol = 0;
    bool stream = false;
    if (P.leaf_assign && ld(&sh.open_vars) != 0) { all_entailed = false; st(&sh.open_vars, 0); }
    if (!failed && all_entailed) {
      TB_REGION(62);
      leaf = 1;
      if (P.obj_var >= 0) {
        const int obj = load_dom<C>(store, P.n_int, P.obj_var).lb;
        if (sh.best_bound > obj && (!P.use_fixed_bound || obj <= P.fixed_bound)) {  // barebones:994
          sh.best_bound = obj;
          sol = 1;
          if (!P.use_fixed_bound && lead) {
            const int old = __hip_atomic_fetch_min(&glob(P.ctrl)->best_bound, obj, TB_RLX, TB_AGENT);  // appx_best_bound.meet
            if (obj < old) {
              publish_bound(P, mbox, obj);
              stream = P.ring.slots != 0;  // best_has_changed && is_printing_intermediate_sol (gpu_dive_and_solve.hpp:341-344)
            }
          }
        }
      } else {
        // A leaf met while diving is reached by every workgroup whose subproblem lies below it (barebones:736-739):
        // only the leftmost one reports the solution, so that `-a` / `-n k` enumerate each solution leaf once.
        sol = (sh.remaining > 0 && (sh.sub_idx & ((1ull << sh.remaining) - 1ull)) != 0ull) ? 0 : 1;
      }
      if (sol && lead) {
        bs.solutions++;
        bs.best_sub = (long long)sh.sub_idx;
        bs.best_time = t1 - sh.t_start;
        if (P.use_fixed_bound) {
          __hip_atomic_fetch_min(&glob(P.ctrl)->first_sol_idx, sh.sub_idx, TB_RLX, TB_AGENT);
          sh.stop = 1;
        } else if (P.obj_var < 0 && (P.stop_after_n_solutions != 0 || P.ring.slots != 0)) {
          const unsigned long long nsol = __hip_atomic_fetch_add(&glob(P.ctrl)->solutions, 1ull, TB_RLX, TB_AGENT) + 1;
          stream = P.ring.slots != 0 && (P.stop_after_n_solutions == 0 || nsol <= P.stop_after_n_solutions);
          if (P.stop_after_n_solutions != 0 && nsol >= P.stop_after_n_solutions) {  // common_solving.hpp:858-867
            bs.exhaustive = 0;
            sh.stop = 1;
            raise_gpu_stop(P);
          }
        }
      }
    }
    TB_REGION(64);
    sh.ticket = stream ? (long long)__hip_atomic_fetch_add(&glob(P.ctrl)->sol_ticket, 1ull, TB_RLX, TB_AGENT) : -1ll;
    sh.leaf = leaf;
    sh.sol = sol;
    if (lead) {
    bs.fixpoint_iterations += (unsigned long long)iters;
    bs.nodes++;
    bs.fails += failed ? 1 : 0;
    bs.depth_max = sh.depth > bs.depth_max ? sh.depth : bs.depth_max;
    // stopping conditions (barebones:1024-1029): one 16-byte look at the grid words per node; the mailbox and the peer
    // cell are polled on a wall-clock period by whichever workgroup notices that the poll is due
    bool must_stop = (P.cut_nodes != 0 && bs.nodes >= P.cut_nodes);
    if (P.cut_nodes_total != 0 && (bs.nodes % NODE_BATCH) == 0) {
      TB_REGION(63);
      // the budget of the whole search: counted per device (agent scope); a single GPU (or one without linked peers, which was
      // given its own share of the budget) checks its own count, linked GPUs are summed by their pollers (poll_outside)
      const unsigned long long mine = __hip_atomic_fetch_add(&glob(P.ctrl)->nodes_local, (unsigned long long)NODE_BATCH, TB_RLX, TB_AGENT) + NODE_BATCH;
      if (P.peers == nullptr && mine >= P.cut_nodes_total) raise_gpu_stop(P);
    }
    TB_REGION(65);
    const Hot hot = load_hot(glob(P.ctrl));
    maybe_poll(P, mbox, hot, t1);
    if (hot.stop != 0) must_stop = true;
    if (P.use_fixed_bound && __hip_atomic_load(&glob(P.ctrl)->first_sol_idx, TB_RLX, TB_AGENT) < sh.sub_idx) { sh.stop = 1; }
    if (aborted) { must_stop = true; (void)__hip_atomic_fetch_or(&glob(P.ctrl)->stop, STOP_HOST, TB_RLX, TB_AGENT); }
    if (must_stop) { bs.exhaustive = 0; sh.stop = 1; bs.why |= 4 | (aborted ? 8 : 0) | ((P.cut_nodes != 0 && bs.nodes >= P.cut_nodes) ? 16 : 0); }
    }  // lead
  }
  if constexpr (C == 5) {
    // the leader's verdict on stopping is everybody's (a member that ran out of time counts too); also the point after which nobody reads this node's store any more
    const unsigned all = team_sync(P, sh, ((lead && sh.stop) || ld(&sh.abort)) ? TEAM_STOP : 0u);
    if (tid == 0 && (all & TEAM_STOP)) sh.stop = 1;
  }
  __syncthreads();
  TB_REGION(26);
  if (sh.sol) {  // uniform (and the same on every member of a team)
    TB_REGION(27);
    if (lead) {
    // (the workgroup's slab of g_best is located here, where a solution is kept: not a pointer that lives through every round of every node)
    if (best_store == nullptr) best_store = glob(P.g_best) + (size_t)here_s(blockIdx.x) * P.vext;
    for (int rep = reps_of(P, 8); rep > 1; --rep) store_out<C, TB>(best_store, store, P.vext);
    store_out<C, TB>(best_store, store, P.vext);
    __syncthreads();
    if (sh.ticket >= 0) produce_solution<TB, C>(P, sh, store, mbox);
    }
    if constexpr (C == 5) (void)team_sync(P, sh, 0u);  // the store stays as it is until the leader has copied it
  }
  TB_REGION(48);
}

// Event kernels: the node (fixpoint + bookkeeping + best-store copy) and the variable selection are functions of their own -- one
// copy of each instead of one per call site, registers allocated for their own loops, and the persistent search loop keeps only
// a handful of values alive across them.  (The sweeping kernels stay inlined: their problem description is a by-value kernel
// argument, which a call would have to copy to memory.)
template <int C, int MEM, int TB = 0>
static __device__ TB_NODE_ATTR unsigned propagate_node_event(const DevProblem* Pp, unsigned sh_off, unsigned store_off, int2* gstore, unsigned props_off, const int4* gprops,
                                                      unsigned dirty_off, unsigned list_off, int2* best_store, Mailbox* mbox, unsigned writes) {
  BlockShared& sh = *lds_ptr<BlockShared>(sh_off);
  int2* store = MEM >= TB_MEM_STORE_SHARED ? lds_ptr<int2>(store_off) : glob(gstore);
  const int4* props = MEM == TB_MEM_TCN_SHARED ? lds_ptr<const int4>(props_off) : glob(gprops);
  if (best_store != nullptr) best_store = glob(best_store);
  EventState es;
  es.dirty = lds_ptr<unsigned>(dirty_off); es.list = lds_ptr<int>(list_off);
  es.succ = glob(constant_problem(Pp).succ);  // (outlined variants keep the successor records in global memory)
  es.unent = reinterpret_cast<unsigned char*>(store) + constant_problem(Pp).unent_off;
  es.words = constant_problem(Pp).dirty_words; es.cap = constant_problem(Pp).chg_cap;
  ThreadCounters tc;
  tc.writes = writes;
  propagate_node_impl<true, C, false, MEM, TB>(constant_problem(Pp), sh, store, props, es, best_store, glob(mbox), tc);
  return tc.writes;
}
template <int C, bool SL, int TB = 0>
static __device__ TB_SPLIT_ATTR void split_event(const DevProblem* Pp, unsigned sh_off, Decision* dec, unsigned store_off, const int2* gstore) {
  split<C, TB>(constant_problem(Pp), *lds_ptr<BlockShared>(sh_off), glob(dec), SL ? lds_ptr<const int2>(store_off) : glob(gstore));
}

template <bool EVENT, int C, bool RM, int MEM, int TB = 0, bool DEEP = false>
__device__ __forceinline__ void propagate_node(const DevProblem& P, BlockShared& sh, int2* store, const int4* props, const EventState& es,
                                               int2* best_store, Mailbox* mbox, ThreadCounters& tc) {
  constexpr bool SL = MEM >= TB_MEM_STORE_SHARED, PL = MEM == TB_MEM_TCN_SHARED;
  if constexpr (EVENT) tc.writes = propagate_node_event<C, MEM, TB>(&P, lds_off(&sh), SL ? lds_off(store) : 0u, SL ? nullptr : store, PL ? lds_off(props) : 0u, PL ? nullptr : props,
                                                                lds_off(es.dirty), lds_off(es.list), best_store, mbox, tc.writes);
  else propagate_node_impl<EVENT, C, RM, MEM, TB, DEEP>(P, sh, store, props, es, best_store, mbox, tc);
}
template <bool EVENT, int C, bool SL, int TB = 0>
__device__ __forceinline__ void split_node(const DevProblem& P, BlockShared& sh, Decision* dec, int2* store) {
  if constexpr (EVENT) split_event<C, SL, TB>(&P, lds_off(&sh), dec, SL ? lds_off(store) : 0u, SL ? nullptr : store);
  else split<C, TB>(P, sh, dec, store);
}

// ---- the persistent search kernel ----------------------------------------------------------------

// Thread 0, when the last level of the dive has been taken (or there is none): the dive's share of the time, and the EPS strategy
// (strategy 0, dive only) hands over to the others (barebones:747-750).
__device__ __forceinline__ void end_of_dive_timer(BlockShared& sh) {
  sh.bs.timers[TB_T_DIVE] += wall_clock64() - sh.t_dive;
  sh.t_dive = 0;
}
__device__ __forceinline__ void end_of_dive(const DevProblem& P, BlockShared& sh) {
  end_of_dive_timer(sh);
  if (P.has_eps_strategy) { sh.cur_strategy = sh.cur_strategy > 1 ? sh.cur_strategy : 1; sh.next_unassigned = 0; }
}


// The event-driven variant is latency bound: its 256-thread form asks the register allocator for 7 waves per
// SIMD (<= 72 VGPRs) so that 7 workgroups are resident per CU when their stores fit (wordpress7_500: 7 x 22.7 KB of
// LDS; measured 17.9 / 20.5 / 22.3 / 23.4e6 nodes/s with 4 / 5 / 6 / 7); the sweep variants are VALU bound and keep 4.
// OPT: the COMPACT store layout for the event kernels, entailed-slice removal for the sweeping ones.
// (r06, measured and dropped: stating the occupancy as an exact range -- amdgpu_waves_per_eu(7, 7) -- does not buy scalar registers.  The 70-94 SGPRs these kernels spill into
//  VGPR lanes (~500 v_readlane / v_writelane per node of wordpress7_500 by the block counts, 6 % of its vector issue) are the price of 7 waves per SIMD: 800 SGPRs / 7 = 114,
//  minus the 16 the trap handler reserves, rounded down to the granule of 16 = 96 including VCC and friends -- 88 for the allocator, whatever the attribute says.  Six waves get 102.)
template <int MEM, int TMAX, bool EVENT, int OPT>
__global__ void __launch_bounds__(TMAX, (EVENT && TMAX == 128) ? (OPT == 4 ? TB_EVENT_WAVES_C8 : TB_EVENT_WAVES) : ((EVENT && TMAX == 256) ? TB_EVENT_WAVES_256 : (TMAX == 256 ? 5 : 4))) solve_kernel(DevProblem by_value, const DevProblem* __restrict__ problem, Mailbox* mbox) {
  // The problem description is read through a pointer, not passed by value: as kernel arguments its ~70 scalars were all
  // hoisted into SGPRs for the whole persistent loop and 260 of them spilled through VGPR lanes (v_writelane / v_readlane,
  // VALU work on an issue-bound kernel); behind a pointer the compiler loads a field where it is used (scalar cache):
  // 137 spills, +4 % nodes/s on wordpress7_500.  The sweeping kernels keep the by-value arguments: their loops are short and
  // spill-free, and reloading fields inside them costs 2 %.
  // The problem description and the per-slice table are read with scalar loads.  They were written by the host (hipMemcpy) into
  // buffers whose addresses earlier launches of this process may have read through the scalar cache under another content; the
  // dispatch does not reliably invalidate that cache (observed: stale per-slice words on a re-used address, 14 of 30 searches of
  // pat11 went wrong), so every wave drops it once, here.
  __builtin_amdgcn_s_dcache_inv();
  // (the two-wave event kernels are only ever launched with exactly 128 threads: block_threads)
  constexpr int TB = (EVENT && TMAX == 128) ? 128 : 0;
  const DevProblem& P = EVENT ? constant_problem(problem) : by_value;  // (constant address space: every field is fetched with a scalar load where it is used)
  // OPT: event kernels -- the store layout (0 plain, 1 COMPACT, 2 COMPACT16, 3 HOT, 4 COMPACT8); sweeps -- bit 0 entailed-slice removal, bits 1-2 the layout
  constexpr int C = EVENT ? OPT : (OPT >> 1);
  constexpr bool RM = !EVENT && (OPT & 1) != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  BlockShared& sh = *reinterpret_cast<BlockShared*>(smem);
  const int tid = threadIdx.x, b = blockIdx.x;
  // LDS: [control block][store slab: vext x 8 B (STORE/TCN_SHARED) | the hot tier of a store in global memory (layout 3)][dirty bitmap][change list][bytecodes (TCN_SHARED)]
  const int VX = P.vext;
  static_assert(C != 3 || MEM == TB_MEM_GLOBAL, "the hot tier belongs to stores in global memory");
  const size_t store_bytes = MEM >= TB_MEM_STORE_SHARED ? (((size_t)VX * 8 + 15) / 16) * 16 : (C == 3 ? (size_t)HOT_VARS * 8 : 0);
  const size_t dirty_bytes = dirty_region_bytes(P.dirty_words) + (((size_t)P.chg_cap * 4 + 15) / 16) * 16;
  int2* store = MEM >= TB_MEM_STORE_SHARED ? reinterpret_cast<int2*>(smem + SH_BYTES) : glob(P.g_store) + (size_t)b * VX;
  EventState es;
  es.dirty = reinterpret_cast<unsigned*>(smem + SH_BYTES + store_bytes);
  es.list = reinterpret_cast<int*>(smem + SH_BYTES + store_bytes + dirty_region_bytes(P.dirty_words));
  es.unent = reinterpret_cast<unsigned char*>(store) + P.unent_off;
  es.words = P.dirty_words; es.cap = P.chg_cap;
  const int4* props = glob(P.props);
  es.succ = glob(P.succ);
  if (MEM == TB_MEM_TCN_SHARED) {
    int4* lprops = reinterpret_cast<int4*>(smem + SH_BYTES + store_bytes + dirty_bytes);
    for (int i = tid; i < P.n_slices * 64; i += block_threads<TB>()) lprops[i] = glob(P.props)[i];  // whole slices: the array is padded
    props = lprops;
    if (EVENT) {  // the successor records too: a run then starts without a trip to L2
      int4* lsucc = lprops + (size_t)P.n_slices * 64;
      for (int i = tid; i < P.n_slices * 64; i += block_threads<TB>()) lsucc[i] = glob(P.succ)[i];
      es.succ = lsucc;
    }
  }
  for (int i = tid; i < 2 * P.dirty_words; i += block_threads<TB>()) es.dirty[i] = 0;
  // This workgroup's slabs of the snapshot stack and of the decision stack are located where a node needs them (after its fixpoint), from an
  // opaque copy of the workgroup index: as loop invariants they were hoisted to the top of the search and spilled for its whole duration.
  auto snap_of = [&]() { return glob(P.g_snap) + (size_t)here_s(b) * P.snapshot_levels * P.vext; };
  auto dec_of = [&]() { return glob(P.g_dec) + (size_t)here_s(b) * P.max_depth; };
  BlockStats& bs = sh.bs;
  ThreadCounters tc;
  if (tid == 0) {
    for (int i = 0; i < TB_NUM_TIMERS; ++i) bs.timers[i] = 0;
    bs.nodes = bs.fails = bs.solutions = bs.fixpoint_iterations = bs.num_deductions = 0;
    bs.eps_solved = bs.eps_skipped = bs.store_writes = bs.stolen = bs.active_evals = 0;
    bs.wait_ticks = 0;
    bs.why = 0; bs.pad_why = 0;
    for (int i = 0; i < TB_DBG_WORDS; ++i) bs.dbg[i] = 0;
#ifdef TB_TUNING
    for (int i = 0; i < 72; ++i) bs.reg[i] = 0;
    for (int i = 0; i < TB_NUM_PROF; ++i) bs.prof[i] = 0;
#endif
    bs.depth_max = 0; bs.exhaustive = 1; bs.num_blocks_done = 0; bs.best_bound = PINF; bs.best_sub = -1; bs.best_time = 0;
    sh.stop = 0; sh.bot = 0; sh.leaf = 0; sh.depth = 0; sh.best_bound = PINF; sh.sol = 0; sh.found = 0; sh.skip = 0; sh.open_vars = 0;
    sh.n_dec_seg = 0;
    sh.abort = 0; sh.new_depth = 0; sh.ev_all = 0; sh.chg_count[0] = 0; sh.chg_count[1] = 0; sh.team_res = 0; sh.witness = -1;
    sh.t_start = sh.t_mark = wall_clock64();
    sh.has_work = next_subproblem(P, sh, mbox) ? 1 : 0;
  }
  __syncthreads();

  // B. dive-and-solve loop (barebones:656-886).  The dive (barebones:675-714) and the solve loop (barebones:752-864) share their body -- propagate, then
  // branch -- so they are ONE loop here with one call site of the node and of the variable selection (r03: the two inlined copies made the headline
  // kernel 112 KB of code for a dive that is 0.2 % of its time): a node of the dive applies no objective bound (gpu_dive_and_solve.hpp:370-372),
  // takes no snapshot, keeps no decision (the child is chosen by a bit of the subproblem index) and a leaf there skips the subtree.
  while (sh.has_work && !sh.stop) {
    // C. restore the root
    TB_REGION(38);
    store_in<C, TB>(store, glob(P.root_store), VX);  // the root slab is laid out like a workgroup slab
    if (EVENT && tid == 0) { sh.ev_all = P.root_fixpoint ? 0 : 1; sh.chg_count[0] = 0; }  // a root that is not a fixpoint: every slice runs once
    if (RM && !P.root_fixpoint) {  // nothing is known to be entailed yet (the event fixpoint does this in its root pass)
      __syncthreads();
      for (int s = tid; s < P.n_slices; s += block_threads<TB>()) es.unent[s] = 1;
    }
    if (tid == 0) {
      sh.cur_strategy = 0; sh.next_unassigned = 0; sh.depth = 0; sh.bot = 0;
      sh.remaining = P.subproblems_power; sh.leaf = 0;
      sh.last_obj_ub = PINF;
      sh.t_dive = wall_clock64();
      if (P.use_fixed_bound && __hip_atomic_load(&glob(P.ctrl)->first_sol_idx, TB_RLX, TB_AGENT) < sh.sub_idx) sh.stop = 1;
      if (P.subproblems_power == 0) end_of_dive(P, sh);
    }
    __syncthreads();
    bool exhausted = false;  // uniform: the subproblem's tree was searched to the end (barebones:866-870)
    while (!sh.stop) {
      TB_REGION(28);
      const bool diving = sh.remaining > 0;  // uniform: thread 0 last wrote it before a barrier
      if (!diving) {
        // I. tighten the objective with the incumbent (barebones:756-771)
        if (tid == 0 && P.obj_var >= 0) {
          TB_REGION(29);
          if (P.use_fixed_bound) { sh.last_obj_ub = P.fixed_bound; embed0_mark<EVENT, C>(P, sh, es, store, &sh.bot, P.obj_var, NINF, P.fixed_bound); }
          else {
            const unsigned long long bf = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&glob(P.ctrl)->best_bound), TB_RLX, TB_AGENT);
            int g = (int)(bf & 0xffffffffull);
            const int f = (int)(bf >> 32);  // Ctrl::foreign_bound
            g = f < g ? f : g;
            g = sh.best_bound < g ? sh.best_bound : g;
            if (g != PINF) {
              if (g == NINF) { TB_REGION(61); sh.stop = 1; raise_gpu_stop(P); }  // unbounded objective
              else { sh.last_obj_ub = g - 1; embed0_mark<EVENT, C>(P, sh, es, store, &sh.bot, P.obj_var, NINF, g - 1); }
            }
          }
        }
        TB_REGION(49);
        __syncthreads();
        if (sh.stop) break;
      }
      // II. propagate
      propagate_node<EVENT, C, RM, MEM, TB, (!EVENT && TMAX == 1024)>(P, sh, store, props, es, nullptr, mbox, tc);
      if (sh.stop) break;
      TB_REGION(30);
      int2* const snap = snap_of();
      Decision* const dec = dec_of();
      // III. branch
      if (!sh.leaf) {
        TB_REGION(31);
        const int d0 = sh.depth;
        const bool prof = (knobs(P) & 0x10000) != 0;
        long long tp = 0;
        if (prof && tid == 0) tp = wall_clock64();
        if (!diving) {
          TB_REGION(32);
          if (d0 < P.snapshot_levels) store_out<C, TB>(snap + (size_t)TB_IDX(17, d0, snapshot_levels) * VX, store, VX);  // d0 == 0: barebones:785-791
          if ((pk(P) & 0x4) && d0 < P.snapshot_levels) store_out<C, TB>(snap + (size_t)d0 * VX, store, VX);  // tuning: cost of the snapshot
          if (tid == 0 && d0 == 0) { sh.snap_strategy = sh.cur_strategy; sh.snap_next_unassigned = sh.next_unassigned; }
          __syncthreads();
        }
        TB_REGION(50);
        if (prof && tid == 0) { const long long t = wall_clock64(); TB_PROF_ADD(bs, TB_PROF_SNAPSHOT_PUSH, t - tp); tp = t; }  // profiling: snapshot push
        split_node<EVENT, C, (MEM >= TB_MEM_STORE_SHARED), TB>(P, sh, dec, store);
        TB_REGION(51);
        if (prof && tid == 0) TB_PROF_ADD(bs, TB_PROF_VARIABLE_SELECTION, wall_clock64() - tp);  // profiling: variable selection
        if (sh.stop) break;
        if (tid == 0) {
          TB_REGION(35);
          if (!sh.found) { sh.leaf = 1; bs.exhaustive = 0; bs.why |= diving ? 1 : 2; }  // unsplittable infinite domains (barebones:688-694)
          else if (diving) {
            --sh.remaining;
            --sh.depth;  // decisions are not recorded while diving
            const int bit = (int)((sh.sub_idx >> sh.remaining) & 1ull);
            embed0_mark<EVENT, C>(P, sh, es, store, &sh.bot, dec[0].var, dec[0].child[bit].x, dec[0].child[bit].y);
            if (sh.remaining == 0) end_of_dive(P, sh);
          } else {
            Decision& dd = dec_at(P, sh, dec, sh.depth - 1);
            const int c = ++dd.cur;
            embed0_mark<EVENT, C>(P, sh, es, store, &sh.bot, dd.var, dd.child[c].x, dd.child[c].y);
            // test aid (tb_session_debug_path): the objective bound in force when this decision was taken
            if (P.g_path_ub != nullptr && sh.depth <= P.max_depth) glob(P.g_path_ub)[(size_t)b * P.max_depth + (sh.depth - 1)] = sh.last_obj_ub;
          }
        }
        TB_REGION(52);
        __syncthreads();
      }
      if (sh.leaf) {
        // E. a leaf above the subproblem: skip the whole subtree (barebones:718-741)
        if (diving) { TB_REGION(60); if (tid == 0) skip_subtree(P, sh); break; }
        // IV. backtrack: rope jump, then restore the deepest snapshot and replay (barebones:812-863)
        TB_REGION(36);
        const int dcur = sh.depth;  // stable: last written before a barrier
        if (dcur == 0) { exhausted = true; break; }
        if (tid == 0) { const Decision& dl = dec_at(P, sh, dec, dcur - 1); sh.new_depth = dl.rope[dl.cur]; }
        __syncthreads();
        const int depth = sh.new_depth;
        if (depth == -1) { exhausted = true; break; }
        const int lvl = (depth - 1) < (P.snapshot_levels - 1) ? (depth - 1) : (P.snapshot_levels - 1);
        (void)TB_IDX(17, lvl, snapshot_levels);
        for (int rep = reps_of(P, 6); rep > 0; --rep) store_in<C, TB>(store, snap + (size_t)lvl * VX, VX);
        if (tid == 0) { sh.bot = 0; sh.depth = depth; }
        __syncthreads();
        // re-apply decisions[lvl .. depth-2].current(): distinct decisions may hit the same variable, the
        // atomic min/max make the order irrelevant (the reference loops to a fixpoint, barebones:839-851)
        for (int i = lvl + tid; i < depth - 1; i += block_threads<TB>()) {
          const Decision& di = dec_at(P, sh, dec, i);
          const int2 ch = di.child[di.cur];
          raise_lb<C>(store, P.n_int, di.var, ch.x);
          lower_ub<C>(store, P.n_int, di.var, ch.y);
          if (EVENT) note_change(sh, es, var_of<C>(di.var), EV_LB | EV_UB);
        }
        __syncthreads();
        if (tid == 0) {
          TB_REGION(37);
          Decision& dd = dec_at(P, sh, dec, depth - 1);
#ifdef TB_TRAP_SEED
          if (dd.var < 0 || dd.var >= P.n_vars + 100000 || dd.cur < -1 || dd.cur > 0) {
            int w[40];
            const int* a = reinterpret_cast<const int*>(&dd);
            for (int q = 0; q < 8; ++q) w[q] = a[q];
            const int* bq = reinterpret_cast<const int*>(&dec_at(P, sh, dec, dcur - 1));
            for (int q = 0; q < 8; ++q) w[8 + q] = bq[q];
            w[16] = dcur; w[17] = depth; w[18] = lvl;
            const int* cq = reinterpret_cast<const int*>(&dec_at(P, sh, dec, depth > 1 ? depth - 2 : 0));
            for (int q = 0; q < 8; ++q) w[19 + q] = cq[q];
            const int* dq = reinterpret_cast<const int*>(&dec_at(P, sh, dec, depth));
            for (int q = 0; q < 8; ++q) w[27 + q] = dq[q];
            w[35] = (int)(reinterpret_cast<size_t>(&dd) & 0xffffffffu); w[36] = (int)(reinterpret_cast<size_t>(dec) & 0xffffffffu);
            trap_report(P, sh, 2, w, 37);
          }
#endif
          const int c = ++dd.cur;
          embed0_mark<EVENT, C>(P, sh, es, store, &sh.bot, dd.var, dd.child[c].x, dd.child[c].y);
          sh.cur_strategy = sh.snap_strategy;
          sh.next_unassigned = sh.snap_next_unassigned;
        }
        TB_REGION(53);
        __syncthreads();
      }
    }
    TB_REGION(54);
    if (tid == 0) {
      if (sh.t_dive != 0) end_of_dive_timer(sh);  // (stopped, or met a leaf, while diving)
      if (exhausted && !sh.stop) bs.eps_solved += 1;
    }
    // G. next subproblem (barebones:877-884)
    if (tid == 0 && !sh.stop) {
      TB_REGION(55);
      const long long t = wall_clock64();
      bs.timers[TB_T_SEARCH] += t - sh.t_mark;
      sh.has_work = next_subproblem(P, sh, mbox) ? 1 : 0;
      sh.t_mark = wall_clock64();  // time spent waiting for work is not search time (BlockStats::wait_ticks)
    }
    TB_REGION(56);
    __syncthreads();
  }
  TB_REGION(57);
  // test aid (tb_config.reserved[0] & 0x800000, tb_session_debug_path): where this workgroup stood when it left
  if (P.g_path_hdr != nullptr && tid == 0) {
    PathHeader h;
    h.sub_idx = sh.sub_idx; h.remaining = sh.remaining; h.depth = sh.depth; h.last_obj_ub = sh.last_obj_ub; h.failed = sh.bot;
    h.has_work = sh.has_work; h.nodes = (int)bs.nodes;
    glob(P.g_path_hdr)[b] = h;
  }
  if (P.g_last != nullptr) store_out<C, TB>(glob(P.g_last) + (size_t)b * VX, store, VX);  // test aid: the store this workgroup stopped on

  // fold what is left of the per-lane write counters into the workgroup's statistics
  __syncthreads();
  flush_writes(sh, tc, true);
  __syncthreads();
  if (tid == 0) {
    bs.best_bound = sh.best_bound;
    const int stopped = __hip_atomic_load(&glob(P.ctrl)->stop, TB_RLX, TB_AGENT) & STOP_HOST;
    if (!(P.cut_nodes != 0 && bs.nodes >= P.cut_nodes) && !stopped) bs.num_blocks_done = 1;  // barebones:889-891
    const long long t_end = wall_clock64();
    bs.timers[TB_T_FIRST_BLOCK_IDLE] = t_end - sh.t_start;
    bs.timers[TB_T_OVERALL] = t_end - sh.t_start;
    bs.timers[TB_T_LATEST_BEST_OBJ_FOUND] = bs.best_time;
    glob(P.g_stats)[b] = bs;
    __hip_atomic_fetch_add(&glob(P.ctrl)->blocks_done, 1, TB_RLX, TB_AGENT);
  }
}

// ---- the persistent search kernel of workgroup teams (store layout 5, r05) ---------------------------------------------------------------------
// One subproblem per TEAM (the workgroups of one XCD) at a time, on one store in global memory -- see "workgroup teams" above.  Same dive-and-solve loop as
// solve_kernel (barebones:656-886): the sweeps are partitioned over the members, the control is replicated, the leader talks to the queue, the grid words
// and the host.  Sweeping fixpoints only (AC1 / WAC1), 1024 threads, the PLAIN layout.
// Between "the last read of this node's store" and "the first write for the next node" stands a team barrier: a member that is still selecting a variable must
// not see the decision a faster member has already applied.
// (a template only so that it is instantiated by the translation unit that launches it, kernel_units.inc: unit 5)
// EVENT (r06): the members share the rounds of the event-driven fixpoint (fixpoint_event_team) instead of partitioned sweeps.
template <int WG_PER_CU, bool EVENT = false>
__global__ void __launch_bounds__(1024, 4 * WG_PER_CU) solve_kernel_team(DevProblem P, Mailbox* mbox) {
  __builtin_amdgcn_s_dcache_inv();
  constexpr int C = 5;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  BlockShared& sh = *reinterpret_cast<BlockShared*>(smem);
  const int tid = threadIdx.x, b = blockIdx.x, VX = P.vext;
  EventState es{};  // (unused by the sweeps)
  BlockStats& bs = sh.bs;
  ThreadCounters tc;
  if (tid == 0) {
    for (int i = 0; i < TB_NUM_TIMERS; ++i) bs.timers[i] = 0;
    bs.nodes = bs.fails = bs.solutions = bs.fixpoint_iterations = bs.num_deductions = 0;
    bs.eps_solved = bs.eps_skipped = bs.store_writes = bs.stolen = bs.active_evals = 0;
    bs.wait_ticks = 0;
    bs.why = 0; bs.pad_why = 0;
    for (int i = 0; i < TB_DBG_WORDS; ++i) bs.dbg[i] = 0;
#ifdef TB_TUNING
    for (int i = 0; i < 72; ++i) bs.reg[i] = 0;
    for (int i = 0; i < TB_NUM_PROF; ++i) bs.prof[i] = 0;
#endif
    bs.depth_max = 0; bs.exhaustive = 1; bs.num_blocks_done = 0; bs.best_bound = PINF; bs.best_sub = -1; bs.best_time = 0;
    sh.stop = 0; sh.bot = 0; sh.leaf = 0; sh.depth = 0; sh.best_bound = PINF; sh.sol = 0; sh.found = 0; sh.skip = 0; sh.open_vars = 0;
    sh.n_dec_seg = 0; sh.has_work = 0; sh.remaining = 0; sh.sub_idx = 0;
    sh.abort = 0; sh.new_depth = 0; sh.ev_all = 0; sh.chg_count[0] = 0; sh.chg_count[1] = 0; sh.witness = -1;
    sh.t_start = sh.t_mark = wall_clock64(); sh.t_dive = 0; sh.last_obj_ub = PINF;
    team_join(P, sh);
  }
  __syncthreads();
  const int M = team_size(sh), m = team_member(sh), slab = sh.team_slab;
  const bool lead = m == 0;
  // the team's store and snapshot stack: the slabs of its leader's workgroup (every workgroup has slabs; a team uses one set); the leader's best store is its own slab
  int2* const store = glob(P.g_store) + (size_t)slab * VX;
  int2* const snap = glob(P.g_snap) + (size_t)slab * P.snapshot_levels * VX;
  if constexpr (EVENT) {
    // LDS behind the control block: [this workgroup's ownership table: waves x words][change list]; the dirty bitmaps are the team's, in global memory
    const int W = P.dirty_words, nw = (int)blockDim.x >> 6, G = M * nw;
    unsigned* own_tab = reinterpret_cast<unsigned*>(smem + SH_BYTES);
    for (int q = tid; q < nw * W; q += (int)blockDim.x) {
      const int wv = q / W, wi = q - wv * W, g = m * nw + wv;
      unsigned bits = 0;
      for (int bq = (((g - (wi * 32) % G) % G) + G) % G; bq < 32; bq += G) bits |= 1u << bq;  // slices s = 32 wi + bq with s mod G == g
      own_tab[q] = bits;
    }
    es.own = own_tab;
    es.dirty = glob(P.g_dirty) + (size_t)slab * 2 * (size_t)W;
    es.list = reinterpret_cast<int*>(smem + SH_BYTES + (((size_t)nw * (size_t)W * 4 + 15) / 16) * 16);
    es.unent = reinterpret_cast<unsigned char*>(store) + P.unent_off;
    es.succ = glob(P.succ);
    es.words = W; es.cap = P.chg_cap;
    // (the team's bitmaps start empty: the leader's slab is cleared by its members before the first barrier below)
    for (int q = m * (int)blockDim.x + tid; q < 2 * W; q += M * (int)blockDim.x) __hip_atomic_store(&glob(es.dirty)[q], 0u, TB_RLX, TB_AGENT);
    __syncthreads();
  }
  Decision* const dec = glob(P.g_dec) + (size_t)b * P.max_depth;  // (a copy of the decision stack per member: the control is replicated)
  const int4* const props = P.props;
  // rows of a block copy that are this member's (even boundaries: 16-byte granules stay whole)
  const int chunk = (((VX + M - 1) / M) + 1) & ~1;
  const int row_lo = m * chunk < VX ? m * chunk : VX, row_hi = (m + 1) * chunk < VX ? (m + 1) * chunk : VX;

  // leader: fetch a subproblem; everybody: learn which (uniform call)
  auto fetch = [&]() {
    if (tid == 0 && lead) {
      const bool got = !sh.stop && next_subproblem(P, sh, mbox);
      team_post(P, sh, 0, sh.sub_idx);
      team_post(P, sh, 1, got ? 1ull : 0ull);
    }
    (void)team_sync(P, sh, 0u);
    if (tid == 0) { sh.sub_idx = team_read(P, sh, 0); sh.has_work = (int)team_read(P, sh, 1); }
    (void)team_sync(P, sh, 0u);  // (the slots may be written again after this)
  };
  fetch();
  bool stopped_searching = false;  // the team was stopped in the middle of a subproblem (test aid: the path header)

  while (sh.has_work && !sh.stop) {
    // C. restore the root, striped over the members
    team_copy_in(store, glob(P.root_store), row_lo, row_hi);
    if (tid == 0) {
      sh.cur_strategy = 0; sh.next_unassigned = 0; sh.depth = 0; sh.bot = 0;
      sh.remaining = P.subproblems_power; sh.leaf = 0;
      sh.last_obj_ub = PINF;
      sh.t_dive = wall_clock64();
      if (P.subproblems_power == 0) end_of_dive(P, sh);
      if (EVENT) { sh.ev_all = P.root_fixpoint ? 0 : 1; sh.chg_count[0] = 0; }  // a root that is not a fixpoint: every slice runs once
    }
    (void)team_sync(P, sh, 0u);
    bool exhausted = false;
    while (!sh.stop) {
      const bool diving = sh.remaining > 0;
      if (!diving && P.obj_var >= 0) {
        // I. the objective bound: read from the grid words by the leader, imposed by everybody (barebones:756-771)
        if (tid == 0 && lead) {
          int g = PINF;
          if (P.use_fixed_bound) g = P.fixed_bound == PINF ? PINF : P.fixed_bound + 1;  // (the members subtract one)
          else {
            const unsigned long long bf = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&glob(P.ctrl)->best_bound), TB_RLX, TB_AGENT);
            g = (int)(bf & 0xffffffffull);
            const int f = (int)(bf >> 32);
            g = f < g ? f : g;
            g = sh.best_bound < g ? sh.best_bound : g;
            if (g == NINF) raise_gpu_stop(P);  // unbounded objective
          }
          team_post(P, sh, 2, (unsigned long long)(unsigned)g);
        }
        (void)team_sync(P, sh, 0u);
        if (tid == 0) {
          const int g = (int)(unsigned)team_read(P, sh, 2);
          if (g == NINF) sh.stop = 1;
          else if (g != PINF) { sh.last_obj_ub = g - 1; embed0_mark<EVENT, C>(P, sh, es, store, &sh.bot, P.obj_var, NINF, g - 1); }
        }
        (void)team_sync(P, sh, 0u);
        if (sh.stop) break;
      }
      // II. propagate (the sweeps are partitioned, the verdict is everybody's; ends with the team's agreement on stopping)
      propagate_node_impl<EVENT, C, false, TB_MEM_GLOBAL, 0, true>(P, sh, store, props, es, nullptr, mbox, tc);
      if (sh.stop) break;
      // III. branch
      if (!sh.leaf) {
        const int d0 = sh.depth;
        if (!diving) {
          if (d0 < P.snapshot_levels) team_copy_out(snap + (size_t)d0 * VX, store, row_lo, row_hi);
          if (tid == 0 && d0 == 0) { sh.snap_strategy = sh.cur_strategy; sh.snap_next_unassigned = sh.next_unassigned; }
        }
        split<C>(P, sh, dec, store);
        {  // nobody reads this node's store after this barrier; a member whose decision stack could not grow stops everybody
          const unsigned all = team_sync(P, sh, sh.stop ? TEAM_STOP : 0u);
          if (tid == 0 && (all & TEAM_STOP)) sh.stop = 1;
          __syncthreads();
        }
        if (sh.stop) break;
        if (tid == 0) {
          if (!sh.found) { sh.leaf = 1; if (lead) { bs.exhaustive = 0; bs.why |= diving ? 1 : 2; } }
          else if (diving) {
            --sh.remaining;
            --sh.depth;
            const int bit = (int)((sh.sub_idx >> sh.remaining) & 1ull);
            embed0_mark<EVENT, C>(P, sh, es, store, &sh.bot, dec[0].var, dec[0].child[bit].x, dec[0].child[bit].y);
            if (sh.remaining == 0) end_of_dive(P, sh);
          } else {
            Decision& dd = dec_at(P, sh, dec, sh.depth - 1);
            const int c = ++dd.cur;
            embed0_mark<EVENT, C>(P, sh, es, store, &sh.bot, dd.var, dd.child[c].x, dd.child[c].y);
            // test aid (tb_session_debug_path): the objective bound in force when this decision was taken (the leader's copy of the stack is the one handed out)
            if (P.g_path_ub != nullptr && lead && sh.depth <= P.max_depth) glob(P.g_path_ub)[(size_t)b * P.max_depth + (sh.depth - 1)] = sh.last_obj_ub;
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      if (sh.leaf) {
        if (diving) { if (tid == 0 && lead) skip_subtree(P, sh); break; }
        // IV. backtrack (barebones:812-863): rope jump, restore the deepest snapshot (striped), replay, right child
        const int dcur = sh.depth;
        if (dcur == 0) { exhausted = true; break; }
        if (tid == 0) { const Decision& dl = dec_at(P, sh, dec, dcur - 1); sh.new_depth = dl.rope[dl.cur]; }
        __syncthreads();
        const int depth = sh.new_depth;
        if (depth == -1) { exhausted = true; break; }
        const int lvl = (depth - 1) < (P.snapshot_levels - 1) ? (depth - 1) : (P.snapshot_levels - 1);
        team_copy_in(store, snap + (size_t)lvl * VX, row_lo, row_hi);
        if (tid == 0) { sh.bot = 0; sh.depth = depth; }
        (void)team_sync(P, sh, 0u);  // the whole store is the snapshot before anybody narrows it again
        for (int i = lvl + tid; i < depth - 1; i += (int)blockDim.x) {
          const Decision& di = dec_at(P, sh, dec, i);
          const int2 ch = di.child[di.cur];
          raise_lb<C>(store, P.n_int, di.var, ch.x);
          lower_ub<C>(store, P.n_int, di.var, ch.y);
          if (EVENT) note_change(sh, es, var_of<C>(di.var), EV_LB | EV_UB);
        }
        __syncthreads();
        if (tid == 0) {
          Decision& dd = dec_at(P, sh, dec, depth - 1);
          const int c = ++dd.cur;
          embed0_mark<EVENT, C>(P, sh, es, store, &sh.bot, dd.var, dd.child[c].x, dd.child[c].y);
          sh.cur_strategy = sh.snap_strategy;
          sh.next_unassigned = sh.snap_next_unassigned;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
    if (tid == 0) {
      if (sh.t_dive != 0) end_of_dive_timer(sh);
      if (exhausted && !sh.stop && lead) bs.eps_solved += 1;
    }
    stopped_searching = sh.stop != 0;  // (uniform: last written before a barrier) -- the fetch below reports "no work" after a stop
    fetch();
  }
  __syncthreads();
  // test aid (tb_config.reserved[0] & 0x800000, tb_session_debug_path): where the TEAM stood when it left -- reported in its leader's slot; the other members report "no work"
  // (they hold copies of the same decision stack, and nobody counts their nodes)
  if (P.g_path_hdr != nullptr && tid == 0) {
    PathHeader h;
    h.sub_idx = sh.sub_idx; h.remaining = sh.remaining; h.depth = sh.depth; h.last_obj_ub = sh.last_obj_ub; h.failed = sh.bot;
    h.has_work = (lead && (sh.has_work || stopped_searching)) ? 1 : 0; h.nodes = (int)bs.nodes;
    glob(P.g_path_hdr)[b] = h;
  }
  if (P.g_last != nullptr && lead) store_out<C, 0>(glob(P.g_last) + (size_t)b * VX, store, VX);  // test aid: the store the team stopped on (the leader's slot)
  // "did this workgroup finish its work" (barebones:889-891) is the TEAM's verdict: only the leader counts nodes, so only it can tell a -cutnodes stop from the end of the queue
  if (tid == 0 && lead) {
    const int stopped = __hip_atomic_load(&glob(P.ctrl)->stop, TB_RLX, TB_AGENT) & STOP_HOST;
    team_post(P, sh, 3, (!(P.cut_nodes != 0 && bs.nodes >= P.cut_nodes) && !stopped) ? 1ull : 0ull);
  }
  (void)team_sync(P, sh, 0u);
  flush_writes(sh, tc, true);
  __syncthreads();
  if (tid == 0) {
    bs.best_bound = lead ? sh.best_bound : PINF;
    if (team_read(P, sh, 3) != 0ull && !sh.abort) bs.num_blocks_done = 1;
    const long long t_end = wall_clock64();
    bs.timers[TB_T_FIRST_BLOCK_IDLE] = t_end - sh.t_start;
    bs.timers[TB_T_OVERALL] = t_end - sh.t_start;
    bs.timers[TB_T_LATEST_BEST_OBJ_FOUND] = bs.best_time;
    glob(P.g_stats)[b] = bs;
    __hip_atomic_fetch_add(&glob(P.ctrl)->blocks_done, 1, TB_RLX, TB_AGENT);
  }
}

template <int UNUSED = 0>
__global__ void clock_kernel(long long* out) { *out = wall_clock64(); }

// ---- batch propagation kernel: one store per workgroup (tb_propagate) ----------------------------

struct PropagateOut {
  int failed, all_entailed;
  unsigned long long iterations, deductions, writes;
};

// `stores` holds n_stores slabs of P.vext intervals each (the layout of a workgroup slab, encoded by the host).
template <int MEM, int TMAX, bool EVENT, int OPT>
__global__ void __launch_bounds__(TMAX, 4) propagate_kernel(DevProblem P, int2* stores, PropagateOut* out, int n_stores) {
  __builtin_amdgcn_s_dcache_inv();  // (see solve_kernel)
  // OPT: event kernels -- the store layout (0 plain, 1 COMPACT, 2 COMPACT16, 3 HOT, 4 COMPACT8); sweeps -- bit 0 entailed-slice removal, bits 1-2 the layout
  constexpr int C = EVENT ? OPT : (OPT >> 1);
  constexpr bool RM = !EVENT && (OPT & 1) != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  BlockShared& sh = *reinterpret_cast<BlockShared*>(smem);
  const int tid = threadIdx.x, V = P.n_vars;
  const int VX = P.vext;
  const size_t store_bytes = MEM >= TB_MEM_STORE_SHARED ? (((size_t)VX * 8 + 15) / 16) * 16 : 0;
  const size_t dirty_bytes = dirty_region_bytes(P.dirty_words) + (((size_t)P.chg_cap * 4 + 15) / 16) * 16;
  EventState es;
  es.dirty = reinterpret_cast<unsigned*>(smem + SH_BYTES + store_bytes);
  es.list = reinterpret_cast<int*>(smem + SH_BYTES + store_bytes + dirty_region_bytes(P.dirty_words));
  es.words = P.dirty_words; es.cap = P.chg_cap;
  for (int i = tid; i < 2 * P.dirty_words; i += blockDim.x) es.dirty[i] = 0;
  const int4* props = P.props;
  es.succ = P.succ;
  if (MEM == TB_MEM_TCN_SHARED) {
    int4* lprops = reinterpret_cast<int4*>(smem + SH_BYTES + store_bytes + dirty_bytes);
    for (int i = tid; i < P.n_slices * 64; i += blockDim.x) lprops[i] = P.props[i];  // whole slices: the array is padded
    props = lprops;
    if (EVENT) {
      int4* lsucc = lprops + (size_t)P.n_slices * 64;
      for (int i = tid; i < P.n_slices * 64; i += blockDim.x) lsucc[i] = P.succ[i];
      es.succ = lsucc;
    }
  }
  for (int s = blockIdx.x; s < n_stores; s += gridDim.x) {
    int2* gstore = stores + (size_t)s * VX;
    // GLOBAL mode works in place on the caller's slab
    int2* store = MEM >= TB_MEM_STORE_SHARED ? reinterpret_cast<int2*>(smem + SH_BYTES) : gstore;
    es.unent = reinterpret_cast<unsigned char*>(store) + P.unent_off;
    ThreadCounters tc;
    if (tid == 0) { sh.bot = 0; sh.abort = 0; sh.bs.num_deductions = 0; sh.bs.store_writes = 0; sh.witness = -1; }
    __syncthreads();
    if (MEM >= TB_MEM_STORE_SHARED) { copy_store(store, gstore, VX); __syncthreads(); }
    if (RM) for (int q = tid; q < P.n_slices; q += blockDim.x) es.unent[q] = 1;
    for (int i = tid; i < V; i += blockDim.x) { const Itv d = load_dom<C>(store, P.n_int, i); if (d.lb > d.ub) st(&sh.bot, 1); }
    __syncthreads();
    bool all_entailed = false;
    int iters = 0;
    if (EVENT) { if (tid == 0) { sh.ev_all = 1; sh.chg_count[0] = 0; sh.chg_count[1] = 0; } __syncthreads(); }
    if (!ld(&sh.bot)) {
      if constexpr (EVENT) iters = fixpoint_event<C>(P, sh, store, props, es, tc, all_entailed);
      else iters = fixpoint<RM, C, (TMAX == 1024)>(P, sh, store, props, es.unent, tc, all_entailed);
    }
    if (MEM >= TB_MEM_STORE_SHARED) copy_store(gstore, store, VX);
    flush_writes(sh, tc, true);
    __syncthreads();
    if (tid == 0) {
      PropagateOut o;
      o.failed = ld(&sh.abort) ? -1 : ld(&sh.bot);
      o.all_entailed = (!o.failed && all_entailed) ? 1 : 0;
      o.iterations = (unsigned long long)iters;
      o.deductions = sh.bs.num_deductions;
      o.writes = sh.bs.store_writes;
      out[s] = o;
    }
    __syncthreads();
  }
}

}  // namespace tb
