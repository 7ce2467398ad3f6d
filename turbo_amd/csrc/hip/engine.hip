// libturbo_hip.so -- host shim of the MI355X dive-and-solve engine (C-ABI in include/turbo_hip.h).
//
// Replaces the host halves of the reference's GPU path:
//   include/memory_gpu.hpp:27-84        MemoryConfig (what lives in LDS)             -> plan_launch()
//   include/barebones_dive_and_solve.hpp:527-606  configure_gpu_barebones           -> plan_launch()
//   include/barebones_dive_and_solve.hpp:479-497  launch / wait / reduce_blocks      -> tb_session_*
//   include/memory_gpu.hpp:174-196      wait_solving_ends (100 ms poll)             -> tb_solve()
// No managed memory and no device-side malloc: every buffer is sized on the host and the only
// host<->device traffic during the search is one pinned 16-byte mailbox.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <iterator>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "kernel_units.hpp"
#ifdef TB_SINGLE_TU
#include "kernel_units.inc"
#endif

using namespace tb;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                                       \
  do {                                                                                                      \
    hipError_t _e = (expr);                                                                                 \
    if (_e != hipSuccess) {                                                                                 \
      if (_e == hipErrorOutOfMemory) return fail(TB_ERR_OOM, std::string(#expr) + ": " + hipGetErrorString(_e)); \
      return fail(TB_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                           \
    }                                                                                                       \
  } while (0)

thread_local int g_adj_sizes[2] = {1 << 30, 1 << 30};  // entries of var_adj / 2 and of adj_rest of the tables uploaded last by this thread (bounds build)

struct DeviceCaps {
  int cus = 0, lds_per_cu = 0, wall_khz = 100000;
  size_t free_mem = 0, total_mem = 0;
  std::string arch;
};

int query_caps(int device, DeviceCaps* caps) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(TB_ERR_NO_DEVICE, "no HIP device is visible: the MI355X engine has no CPU fallback");
  if (device < 0 || device >= n) return fail(TB_ERR_INVALID, "device ordinal out of range");
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  caps->cus = prop.multiProcessorCount;
  caps->arch = prop.gcnArchName;
  int lds = 0;
  if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) != hipSuccess || lds <= 0) lds = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (lds <= 0) lds = 64 * 1024;
  caps->lds_per_cu = lds;
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) caps->wall_khz = khz;
  HIP_TRY(hipMemGetInfo(&caps->free_mem, &caps->total_mem));
  return TB_OK;
}

// ---- launch planning -----------------------------------------------------------------------------

constexpr long TEAM_AC1_SORT_WINDOW = 1024;  // records per window of the class sort for a team's plain sweeps (to_internal): one workgroup's 16 slices
struct LaunchPlan {
  int threads = 256, tmax = 256;
  int blocks_per_cu = 1, num_blocks = 1;
  int mem_kind = TB_MEM_GLOBAL;
  int shared_bytes = 0;
  int subproblems_power = 0;
  int snapshot_levels = 1;
  int max_depth = 16384;
  int n_slices = 0, dirty_words = 0, vext = 0, chg_cap = 64;
  int kernel_event = 0, kernel_opt = 0;  // template flags of the kernels this plan launches (solve and root propagation alike)
  int compact = 0, n_int = 0, unent_off = 0;  // store layout (Layout below)
  bool hot = false;  // store in global memory with its first HOT_VARS intervals in LDS (kernels.hpp: layout 3); the search kernel only
  bool team = false; // one store per XCD shared by the workgroups resident on it (kernels.hpp: layout 5, solve_kernel_team); the search kernel only
  int prop_opt() const { return (hot || team) ? 0 : kernel_opt; }  // template flag of the batch-propagation kernel on the same slabs (the plain layout)
};

// Store layout of a session.  COMPACT: the variables whose root domain lies within 0..1 are renumbered behind the
// others and kept as 2 bits each (kernels.hpp: load_dom / raise_lb / lower_ub); everything the engine works on --
// propagators, strategies, objective, adjacency -- is expressed in the internal numbering, and stores are
// converted at the boundary (encode_slab / decode_slab).  Without COMPACT the numbering is the caller's.
//
// Constants out of the slab (COMPACT layouts): a non-Boolean variable that is the same finite singleton in every store of the batch -- TCN has
// no constants, only singleton variables (common_solving.hpp:743-771); wordpress7_500 has 847 of them among its 1644 integers -- is numbered
// behind the Booleans and takes no room in the slab at all: the records that read it carry its VALUE in the operand field (sign bit set,
// kernels.hpp: load_dom), the strategy lists skip it, decode_slab puts it back.  Narrowing a constant is a failure; whoever tries raises the flag.
struct Layout {
  bool renumbered = false;  // plain layout under a locality permutation (renumber_for_locality: an experiment, TB_GLOBAL_RENUMBER)
  bool compact = false;
  bool c16 = false;  // COMPACT16: the non-Boolean variables as two 16-bit bounds in one word (kernels.hpp: load_dom<2>)
  // COMPACT8 (kernels.hpp: layout 4): on top of COMPACT16, an integer whose domain is at most 255 wide over the whole batch takes two bytes, its bounds
  // relative to `base` (the lowest lower bound of the batch).  The wide integers are numbered first (internal ids 0 .. n_wide - 1, a 32-bit word each),
  // the narrow ones behind them (halfword id + n_wide of the slab).
  bool c8 = false;
  int n_wide = 0;
  std::vector<int> base;  // [n_int] (0 for the wide ones)
  // what the device calls a reference to variable i: the id, and for a narrow integer of COMPACT8 its base in bits 16-30
  int ref(int i) const { return (c8 && i >= n_wide && i < n_int) ? (i | ((base[(size_t)i] + 16384) << 16)) : i; }
  int dev_n_int() const { return c8 ? (int)((unsigned)n_int | ((unsigned)n_wide << 16)) : n_int; }  // DevProblem::n_int (unsigned: n_wide may reach 2^15 and beyond)
  int n_vars = 0, n_int = 0, n_bool = 0;
  int n_out = 0;               // constants kept out of the slab: internal ids n_int + n_bool .. n_vars - 1
  std::vector<int> out_value;  // [n_out] their values
  int n_slab() const { return n_int + n_bool; }
  std::vector<int> perm, inv;  // perm[caller's id] = internal id, inv = its inverse
  int bool_words() const { return (n_bool + 15) / 16; }
  int bool_word0() const { return c8 ? (n_int + n_wide + 1) / 2 : (c16 ? n_int : 2 * n_int); }  // first Boolean word, in 32-bit words from the start of the slab
  int unent_off() const { return (bool_word0() + bool_words()) * 4; }
  // slab size in 8-byte units: the domains, then the "not entailed" marks of the slices -- a byte per slice for the sweeps with entailed-slice
  // removal (every wave writes its own), a bit per slice for the event-driven fixpoint (dirty_words 32-bit words)
  int vext(int n_slices, bool event) const {
    const size_t marks = event ? (size_t)((n_slices + 31) / 32) * 4 : (size_t)std::max(n_slices, 4);
    return (int)((((size_t)unent_off() + std::max<size_t>(marks, 4) + 15) / 16) * 2);
  }
};

// (`pinned`: a variable the kernels address by index outside the records -- the objective -- stays in the slab; `outs`: constants leave it)
Layout make_layout(int32_t n_vars, int32_t n_stores, const tb_itv* stores, bool compact, bool want_c16 = false, bool outs = false, int32_t pinned = -1, bool want_c8 = false) {
  Layout L;
  L.n_vars = n_vars; L.compact = compact;
  L.perm.resize((size_t)n_vars); L.inv.resize((size_t)n_vars);
  std::vector<char> is_bool((size_t)n_vars, 0), is_out((size_t)n_vars, 0);
  if (compact) {
    int n_b = 0;
    for (int32_t v = 0; v < n_vars; ++v) {
      bool b = n_stores > 0;
      for (int32_t k = 0; k < n_stores && b; ++k) { const tb_itv d = stores[(size_t)k * (size_t)n_vars + (size_t)v]; b = d.lb >= 0 && d.ub <= 1; }
      is_bool[(size_t)v] = b ? 1 : 0;
      n_b += b ? 1 : 0;
    }
    if (outs && n_b > 0)
      for (int32_t v = 0; v < n_vars; ++v) {
        if (is_bool[(size_t)v] || v == pinned) continue;
        const tb_itv d0 = stores[v];
        bool c = d0.lb == d0.ub && d0.lb > -(1 << 30) && d0.lb < (1 << 30);
        for (int32_t k = 1; k < n_stores && c; ++k) { const tb_itv d = stores[(size_t)k * (size_t)n_vars + (size_t)v]; c = d.lb == d0.lb && d.ub == d0.ub; }
        is_out[(size_t)v] = c ? 1 : 0;
      }
  }
  // hull of every variable over the batch
  auto hull_of = [&](int32_t v) {
    tb_itv h = stores[v];
    for (int32_t k = 1; k < n_stores; ++k) { const tb_itv d = stores[(size_t)k * (size_t)n_vars + (size_t)v]; h.lb = std::min(h.lb, d.lb); h.ub = std::max(h.ub, d.ub); }
    return h;
  };
  // COMPACT8: all integers within 16 bits, at least one of them narrow (bases within +-16383, ids within 16 bits)
  std::vector<char> is_narrow((size_t)n_vars, 0);
  bool c8 = compact && want_c8 && n_stores > 0 && n_vars < 0xffff;
  if (c8) {
    int narrow = 0;
    for (int32_t v = 0; v < n_vars && c8; ++v) {
      if (is_bool[(size_t)v] || is_out[(size_t)v]) continue;
      const tb_itv h = hull_of(v);
      if (h.lb < -32768 || h.ub > 32767 || h.lb > h.ub) { c8 = false; break; }
      if (h.ub - h.lb <= 255 && h.lb >= -16383 && h.lb <= 16382) { is_narrow[(size_t)v] = 1; ++narrow; }
    }
    if (narrow == 0) c8 = false;
    if (!c8) std::fill(is_narrow.begin(), is_narrow.end(), 0);
  }
  int next = 0;
  if (c8) for (int32_t v = 0; v < n_vars; ++v) if (!is_bool[(size_t)v] && !is_out[(size_t)v] && !is_narrow[(size_t)v]) L.perm[(size_t)v] = next++;
  L.n_wide = c8 ? next : 0;
  for (int32_t v = 0; v < n_vars; ++v) if (!is_bool[(size_t)v] && !is_out[(size_t)v] && (!c8 || is_narrow[(size_t)v])) L.perm[(size_t)v] = next++;
  L.n_int = next;
  for (int32_t v = 0; v < n_vars; ++v) if (is_bool[(size_t)v]) L.perm[(size_t)v] = next++;
  L.n_bool = next - L.n_int;
  for (int32_t v = 0; v < n_vars; ++v) if (is_out[(size_t)v]) { L.perm[(size_t)v] = next++; L.out_value.push_back(stores[v].lb); }
  L.n_out = n_vars - L.n_int - L.n_bool;
  for (int32_t v = 0; v < n_vars; ++v) L.inv[(size_t)L.perm[(size_t)v]] = v;
  if (L.n_bool == 0) L.compact = false;
  if (L.compact && c8) {
    L.c8 = true;
    L.base.assign((size_t)L.n_int, 0);
    for (int32_t v = 0; v < n_vars; ++v) if (is_narrow[(size_t)v]) L.base[(size_t)L.perm[(size_t)v]] = hull_of(v).lb;
  } else if (c8) {  // (no Boolean at all: the layout falls back to the plain one, in the caller's numbering)
    L.n_wide = 0;
    int nx = 0;
    for (int32_t v = 0; v < n_vars; ++v) if (!is_bool[(size_t)v] && !is_out[(size_t)v]) L.perm[(size_t)v] = nx++;
    for (int32_t v = 0; v < n_vars; ++v) L.inv[(size_t)L.perm[(size_t)v]] = v;
  }
  if (L.compact && want_c16 && !L.c8) {  // every non-Boolean variable of the slab within -32768..32767 in every store of the batch
    bool ok = true;
    for (int32_t v = 0; v < n_vars && ok; ++v) {
      if (is_bool[(size_t)v] || is_out[(size_t)v]) continue;
      for (int32_t k = 0; k < n_stores && ok; ++k) { const tb_itv d = stores[(size_t)k * (size_t)n_vars + (size_t)v]; ok = d.lb >= -32768 && d.ub <= 32767; }
    }
    L.c16 = ok;
  }
  return L;
}

// caller's store -> slab (zero-initialised by the caller of this function, `vext * 8` bytes)
void encode_slab(const Layout& L, const tb_itv* orig, unsigned char* slab) {
  tb_itv* ints = reinterpret_cast<tb_itv*>(slab);
  unsigned* ints16 = reinterpret_cast<unsigned*>(slab);
  unsigned short* halves = reinterpret_cast<unsigned short*>(slab);
  unsigned* words = reinterpret_cast<unsigned*>(slab) + L.bool_word0();
  for (int v = 0; v < L.n_vars; ++v) {
    const int i = L.perm[(size_t)v];
    if (i < L.n_int) {
      if (L.c8 && i >= L.n_wide) {
        // (a store of the batch that is already empty, or outside its own hull, cannot happen: the hull is taken over these very stores)
        const int lo = std::min(255, std::max(0, orig[v].lb - L.base[(size_t)i])), hi = std::min(255, std::max(0, orig[v].ub - L.base[(size_t)i]));
        halves[i + L.n_wide] = (unsigned short)(lo | (hi << 8));
      } else if (L.c16 || L.c8) ints16[i] = ((unsigned)orig[v].lb & 0xffffu) | ((unsigned)orig[v].ub << 16);
      else ints[i] = orig[v];
      continue;
    }
    if (i >= L.n_slab()) continue;  // a constant kept out of the slab
    const int b = i - L.n_int;
    const unsigned bits = (orig[v].lb >= 1 ? 1u : 0u) | (orig[v].ub <= 0 ? 2u : 0u);
    words[b >> 4] |= bits << ((b & 15) * 2);
  }
}
void decode_slab(const Layout& L, const unsigned char* slab, tb_itv* orig_out) {
  const tb_itv* ints = reinterpret_cast<const tb_itv*>(slab);
  const unsigned* ints16 = reinterpret_cast<const unsigned*>(slab);
  const unsigned short* halves = reinterpret_cast<const unsigned short*>(slab);
  const unsigned* words = reinterpret_cast<const unsigned*>(slab) + L.bool_word0();
  for (int i = 0; i < L.n_vars; ++i) {
    tb_itv d;
    if (i < L.n_int) {
      if (L.c8 && i >= L.n_wide) { const unsigned h = halves[i + L.n_wide]; d.lb = L.base[(size_t)i] + (int)(h & 0xffu); d.ub = L.base[(size_t)i] + (int)(h >> 8); }
      else if (L.c16 || L.c8) { d.lb = (int)(short)(ints16[i] & 0xffffu); d.ub = (int)ints16[i] >> 16; }
      else d = ints[i];
    }
    else if (i >= L.n_slab()) { d.lb = d.ub = L.out_value[(size_t)(i - L.n_slab())]; }
    else { const int b = i - L.n_int; const unsigned bits = (words[b >> 4] >> ((b & 15) * 2)) & 3u; d.lb = (int)(bits & 1u); d.ub = 1 - (int)(bits >> 1); }
    orig_out[L.inv[(size_t)i]] = d;
  }
}

// Experiment for stores in global memory ("renumber the variables so that a slice's gathers fall in fewer 64-byte lines"):
// breadth-first order over the variable / propagator graph in the caller's propagator order -- the operands of consecutive propagators get
// neighbouring ids, eight intervals to a line.  Plain layout only; off unless TB_GLOBAL_RENUMBER is set (measured r04 on the synthetic 100k x 500k
// network, whose operands are uniform random: see DESIGN.md section 7).
void renumber_for_locality(Layout* L, int32_t n_props, const tb_prop* props) {
  const int V = L->n_vars;
  std::vector<int> order((size_t)V, -1);
  int next = 0;
  for (int32_t i = 0; i < n_props; ++i)
    for (int v : {props[i].x, props[i].y, props[i].z})
      if (order[(size_t)v] < 0) order[(size_t)v] = next++;
  for (int v = 0; v < V; ++v) if (order[(size_t)v] < 0) order[(size_t)v] = next++;
  L->perm = order;
  for (int v = 0; v < V; ++v) L->inv[(size_t)order[(size_t)v]] = v;
  L->renumbered = true;
}

// Hot tier of a store in global memory (kernels.hpp: layout 3): the variables are numbered by how many propagator operands read them, most read
// first, so that the HOT_VARS intervals that live in LDS are the ones gathered most often (ties keep the caller's order).
void renumber_by_reads(Layout* L, int32_t n_props, const tb_prop* props) {
  const int V = L->n_vars;
  std::vector<int> reads((size_t)V, 0), order((size_t)V);
  for (int32_t i = 0; i < n_props; ++i) { reads[(size_t)props[i].x]++; reads[(size_t)props[i].y]++; reads[(size_t)props[i].z]++; }
  for (int v = 0; v < V; ++v) order[(size_t)v] = v;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return reads[(size_t)a] > reads[(size_t)b]; });
  for (int i = 0; i < V; ++i) { L->perm[(size_t)order[(size_t)i]] = i; L->inv[(size_t)i] = order[(size_t)i]; }
  L->renumbered = true;
}

inline size_t align16(size_t x) { return (x + 15) / 16 * 16; }
// LDS is handed out in granules of 320 dwords on gfx950 (160 KiB / 128): a workgroup asking for 12 336 B occupies 12 800 B, and 12 -- not 13 -- of
// them share a CU, whatever hipOccupancyMaxActiveBlocksPerMultiprocessor answers.  Measured r04 (wordpress7_500, 128-thread event workgroups): 12 336 B
// and 11 568 B per workgroup run at 4.27e7 and 4.35e7 nodes/s, 11 504 B (nine granules: 14 workgroups per CU) at 4.64e7.
// (1/128 of a CU's LDS on the CDNA parts: 1280 B of gfx950's 160 KiB, 512 B of gfx942's 64 KiB -- taken from the device, not assumed)
inline size_t lds_granule(const DeviceCaps& caps) { return std::max<size_t>(128, (size_t)caps.lds_per_cu / 128); }
inline size_t lds_footprint(const DeviceCaps& caps, size_t bytes) { const size_t g = lds_granule(caps); return (bytes + g - 1) / g * g; }

// Which memory holds what (the MemoryKind decision of memory_gpu.hpp:56-83 with CDNA4 numbers:
// 160 KiB of LDS per CU, wave64, at most 32 waves per CU) and how many workgroups to launch
// (barebones:530-546: occupancy x CUs, capped by -or).
int plan_launch(const tb_config& cfg, const DeviceCaps& caps, const Layout& lay, int n_props, LaunchPlan* plan, bool search = true) {
  const int n_vars = lay.n_vars;
  LaunchPlan p;
  int T = cfg.threads_per_block;
  // Full sweeps (AC1/WAC1) want a wide workgroup: the sweep is throughput bound.  The event-driven fixpoint
  // runs a few slices per sweep and is latency bound: many small workgroups per CU hide it better.
  const bool event = cfg.fixpoint == 2;
  // (event mode on a very large network, e.g. the 100k x 500k synthetic one: every node touches thousands of slices, so
  //  wide workgroups win again -- measured 5.2e10 against 4.2e10 propagations/s)
  const bool auto_threads = T == 0;
  if (T == 0) T = event ? (n_props >= 262144 ? 1024 : 256) : (n_props >= 16384 ? 1024 : (n_props >= 2048 ? 512 : 256));
  if (auto_threads && !event && !cfg.only_global_memory && (n_props + 63) / 64 >= 32) {
    // A store that leaves room for a single workgroup per CU: make that workgroup wide enough to fill the CU's 16 wave
    // slots (trains15 simplified, 87 KB store: 4.9e6 -> 7.3e6 nodes/s going from 512 to 1024 threads).
    const size_t slab = align16((size_t)lay.vext((n_props + 63) / 64, event) * 8) + dirty_region_bytes(((n_props + 63) / 64 + 31) / 32) + 4096 + SH_BYTES;
    while (T < 1024 && slab <= (size_t)caps.lds_per_cu && std::min<size_t>((size_t)caps.lds_per_cu / lds_footprint(caps, slab), (size_t)(2048 / T)) * (size_t)(T / 64) < 16) T *= 2;
  }
  bool small_wg = false;
  if (auto_threads && T == 256 && !cfg.only_global_memory && (event || n_props < 2048)) {
    // What a CU needs is subproblems in flight (and the fewer waves meet at a barrier, the less they wait for each other): when 10
    // slabs fit in LDS, workgroups of two waves beat 7 of four -- accap_a3, 14 per CU: 4.5e7 -> 7.0e7 nodes/s (event), 3.1e7 -> 5.3e7 (wac1);
    // wordpress7_500 with its constants out of the slab (Layout), same box: 7 x 256 threads 3.38e7, 8 x 128 3.22e7, 10 x 128 3.71e7,
    // 12 x 128 3.95e7.  trains15 (6 slabs fit) keeps its four waves.
    // (with the shortest change list the planner may pick, see below)
    const size_t slab = align16((size_t)lay.vext((n_props + 63) / 64, event) * 8) + dirty_region_bytes(((n_props + 63) / 64 + 31) / 32) + align16(32 * 4) + SH_BYTES;
    if (lds_footprint(caps, slab) * 10 <= (size_t)caps.lds_per_cu) { T = 128; small_wg = true; }
  }
  if (T != 64 && T != 128 && T != 256 && T != 512 && T != 1024) return fail(TB_ERR_INVALID, "threads_per_block must be 64, 128, 256, 512 or 1024");
  p.threads = T;
  // (the event kernels have an instantiation of their own for workgroups of at most two waves: 7 waves per SIMD -- 72 VGPRs -- there,
  //  6 -- 84 VGPRs -- for the 256-thread ones, whose workgroups per CU are limited by LDS before registers: trains15 2.88e7 -> 3.12e7 nodes/s)
  // (the 128-thread instantiation has its workgroup size compiled in, kernels.hpp: block_threads -- 64-thread event workgroups take the 256 one)
  p.tmax = (event && T == 128) ? 128 : (T <= 256 ? 256 : 1024);
  const size_t lds = (size_t)caps.lds_per_cu;
  // slab = domains + one entailment byte per slice, rounded to an even number of intervals so that every slab of a
  // stack (stores, snapshots) starts 16-byte aligned and copies as 16-byte words
  const int n_slices = (n_props + 63) / 64, dirty_words = (n_slices + 31) / 32, vext = lay.vext(n_slices, event);
  p.n_slices = n_slices; p.dirty_words = dirty_words; p.vext = vext;
  p.compact = lay.compact ? (lay.c8 ? 4 : (lay.c16 ? 2 : 1)) : 0; p.n_int = lay.dev_n_int(); p.unent_off = lay.unent_off();
  // store slab = domains + one entailment byte per 64-propagator slice; the dirty bitmap and the change list of the
  // event-driven fixpoint always live in LDS (an overflowing change list falls back to running every slice)
  // (256 entries: a node is entered with a handful of changed variables -- the decision, the objective bound, the decisions replayed
  //  above the deepest snapshot; a longer list only costs LDS that a sixth workgroup per CU can use, trains15)
  p.chg_cap = std::min(256, std::max(64, n_vars / 4));
  const size_t fixed = SH_BYTES;
  int bpc_max = std::min(cfg.reserved[2] > 0 ? cfg.reserved[2] : (small_wg ? 16 : 8), 2048 / T);  // 32 waves per CU (reserved[2]: tuning knob)
  if (bpc_max < 1) bpc_max = 1;
  if (event && !cfg.only_global_memory) {
    // The change list is the one elastic part of an event workgroup's LDS: a node is entered with a handful of changed variables (the decision,
    // the objective bound, the decisions replayed above the deepest snapshot), and an overflowing list only costs that node a pass over every
    // slice.  So its capacity (256 entries by default) gives way, down to 32 entries, when that puts one more workgroup on a CU -- LDS comes in
    // granules (lds_footprint), and what the registers allow is 7 waves per SIMD for both event kernels: wordpress7_500 12 -> 14 workgroups per CU
    // (+8.7 % nodes/s), trains15 6 -> 7 (+4.7 %).
    const size_t base = fixed + align16((size_t)vext * 8) + dirty_region_bytes(dirty_words);
    const int reg_cap = std::max(1, 28 / std::max(1, T / 64));
    auto per_cu = [&](int cap) { return std::min<int>({bpc_max, reg_cap, (int)((size_t)caps.lds_per_cu / lds_footprint(caps, base + align16((size_t)cap * 4)))}); };
    int best_cap = p.chg_cap;
    for (int cap : {192, 128, 96, 64, 48, 32})
      if (cap < best_cap && per_cu(cap) > per_cu(best_cap)) best_cap = cap;
    p.chg_cap = best_cap;
  }
  if (cfg.reserved[1] > 0) p.chg_cap = cfg.reserved[1];  // tuning knob
  const size_t dirty_b = dirty_region_bytes(dirty_words) + align16((size_t)p.chg_cap * 4);
  // (event mode keeps the successor records next to the bytecodes: 32 bytes per propagator)
  const size_t store_b = align16((size_t)vext * 8) + dirty_b, props_b = (size_t)n_slices * 64 * (event ? 32 : 16);  // padded to whole slices
  // Hot tier (r04): a plain store in global memory, 1024-thread workgroups (one per CU, whose LDS would otherwise hold a few KB of bitmaps): the first
  // HOT_VARS intervals live in LDS.  (TB_NO_HOT_TIER: A/B runs.)
  auto hot_tier = [&]() {
    if (!search || lay.compact || T != 1024 || n_vars <= HOT_VARS || cfg.entailed_prop_removal || std::getenv("TB_NO_HOT_TIER") != nullptr) return false;
    return fixed + (size_t)HOT_VARS * 8 + dirty_b <= lds;
  };
  // Workgroup teams (r05; kernels.hpp layout 5): the sweeps of a network whose store lives in global memory, 1024-thread workgroups, one per CU -- the workgroups of an
  // XCD share a store, which then sits in that XCD's L2.  (TB_TEAM=0 switches it off -- the hot tier then --, TB_TEAM=1 takes it wherever it is possible.)
  auto team_mode = [&]() {
    const char* e = std::getenv("TB_TEAM");
    // (r06: the event fixpoint in teams -- fixpoint_event_team -- is opt-in, TB_TEAM_EVENT=1, until it has been measured against the hot tier)
    const char* ee = std::getenv("TB_TEAM_EVENT");
    if (event && !(ee != nullptr && ee[0] == '1')) return false;
    if (!search || lay.compact || T != 1024 || cfg.entailed_prop_removal || (e != nullptr && e[0] == '0')) return false;
    // r05, same box, synthetic 100k x 500k with the product fast path in: hot tier 9.8e10 propagations/s (wac1) / 8.9e10 (ac1); four teams per XCD 1.09e11 / 1.11e11,
    // eight 1.10e11 / 1.07e11, two 1.02e11 / 1.05e11 -- so teams are the plan wherever the hot tier was (profiles/r05_team_ab.txt)
    return (e != nullptr && e[0] == '1') || n_vars > HOT_VARS;
  };
  // (teams with the event fixpoint: LDS holds the workgroup's ownership table -- 16 waves x dirty_words -- and the change list; the bitmaps are the team's, in global memory)
  const size_t team_event_lds = fixed + align16((size_t)16 * (size_t)dirty_words * 4) + align16((size_t)p.chg_cap * 4);
  if (cfg.only_global_memory) {
    p.mem_kind = TB_MEM_GLOBAL; p.blocks_per_cu = bpc_max; p.shared_bytes = (int)(fixed + dirty_b);
    if (team_mode()) { p.team = true; p.blocks_per_cu = TB_TEAM_WG_PER_CU; if (event) p.shared_bytes = (int)team_event_lds; }
    else if (hot_tier()) { p.hot = true; p.blocks_per_cu = 1; p.shared_bytes = (int)(fixed + (size_t)HOT_VARS * 8 + dirty_b); }
  } else if (!event && !lay.compact && lds_footprint(caps, fixed + store_b + props_b) * (size_t)bpc_max <= lds) {
    // (records in LDS: the plain sweeps on small networks only.  The event kernels and the compact layouts have no such instantiation:
    //  their records come out of L2 fast enough, see below, and a third memory kind for them is a quarter of the library's compile time.)
    p.mem_kind = TB_MEM_TCN_SHARED; p.blocks_per_cu = bpc_max; p.shared_bytes = (int)(fixed + store_b + props_b);
  // (event mode, measured on accap_a3: the records in LDS at the price of 3 workgroups per CU instead of 7 -- 1.39e7 against 4.50e7 nodes/s.
  //  The records come out of L2 fast enough; what a CU needs is subproblems in flight.)
  } else if (lds_footprint(caps, fixed + store_b) * (size_t)bpc_max <= lds) {
    p.mem_kind = TB_MEM_STORE_SHARED; p.blocks_per_cu = bpc_max; p.shared_bytes = (int)(fixed + store_b);
  } else if (fixed + store_b <= lds && !(event && (int)(lds / lds_footprint(caps, fixed + store_b)) < 4 && !(cfg.reserved[0] & 0x40000))) {
    // (event mode: a store that leaves fewer than 4 workgroups per CU goes to global memory instead: what a CU needs is
    //  subproblems in flight -- trains15, compact slab of 50 KB: 1.77e7 nodes/s with 3 workgroups per CU in LDS, 2.04e7 with 7
    //  working on slabs in global memory (L2 / Infinity Cache resident); wordpress7_500 r02: 1.2-1.5x.  choose_layout first tries
    //  the COMPACT layout, which usually brings the store back into LDS)
    p.mem_kind = TB_MEM_STORE_SHARED; p.blocks_per_cu = (int)(lds / lds_footprint(caps, fixed + store_b)); p.shared_bytes = (int)(fixed + store_b);
  } else {
    p.mem_kind = TB_MEM_GLOBAL; p.blocks_per_cu = bpc_max; p.shared_bytes = (int)(fixed + dirty_b);
    if (team_mode()) { p.team = true; p.blocks_per_cu = TB_TEAM_WG_PER_CU; if (event) p.shared_bytes = (int)team_event_lds; }
    else if (hot_tier()) { p.hot = true; p.blocks_per_cu = 1; p.shared_bytes = (int)(fixed + (size_t)HOT_VARS * 8 + dirty_b); }
  }
  long long blocks = (long long)p.blocks_per_cu * caps.cus;
  if (cfg.or_nodes != 0) blocks = std::min<long long>(blocks, (long long)cfg.or_nodes);
  blocks = std::min<long long>(blocks, std::max<long long>(1, 200000000ll / std::max(1, n_vars)));  // barebones:584
  p.num_blocks = (int)std::max<long long>(1, blocks);
  // II. number of subproblems: 2^d >= subfactor x blocks (x GPUs) (barebones:550-555)
  p.subproblems_power = cfg.subproblems_power;
  if (p.subproblems_power < 0) {
    const unsigned long long world = (unsigned long long)std::max(1, cfg.world_size);
    const unsigned long long target = std::max<unsigned long long>(1, cfg.subproblems_factor) * (unsigned long long)p.num_blocks * world;
    int d = 0;
    while (d < 40 && (1ull << d) < target) ++d;
    p.subproblems_power = d;
  }
  {
    // a GPU's share of the index space is served through a 28-bit queue word (device_types.hpp: PeerCell::queue)
    const int world = std::max(1, cfg.world_size);
    if (p.subproblems_power > 40 || eps_local_count(p.subproblems_power, 0, 0, world) > Q_MASK - (1ull << 16))
      return fail(TB_ERR_INVALID, "subproblems_power is too large: at most 2^28 - 65536 subproblems per GPU");
  }
  // snapshot stack: as deep as 1/4 of the free HBM allows (288 GB per GPU makes copying cheaper than
  // recomputing from the subproblem root)
  // one segment of a workgroup's decision stack (a power of two: dec_at splits an index with a shift); deeper searches take
  // further segments from a pool (tb_session_create), up to 16 segments per workgroup
  p.max_depth = 16;
  while (p.max_depth < (cfg.decision_stack_depth > 0 ? std::min(cfg.decision_stack_depth, 1 << 26) : 16384)) p.max_depth *= 2;
  int L = cfg.snapshot_levels;
  if (L <= 0) {
    const size_t per_level = (size_t)p.num_blocks * (size_t)std::max(1, vext) * 8;
    size_t budget = caps.free_mem / 4;  // of what is free NOW: a second session on the same device sizes itself on what the first left
    L = (int)std::min<size_t>(256, std::max<size_t>(1, budget / std::max<size_t>(1, per_level)));
  }
  p.snapshot_levels = std::max(1, std::min(L, p.max_depth));
  p.kernel_event = event ? 1 : 0;
  // sweeps: entailed-slice removal (bit 0) or a compact layout (2: COMPACT, 4: COMPACT16) -- not both, to keep the number of kernels down
  p.kernel_opt = p.team ? (event ? TEAM_EVENT_OPT : TEAM_SWEEP_OPT) : (p.hot ? (event ? HOT_EVENT_OPT : HOT_SWEEP_OPT) : (event ? p.compact : (cfg.entailed_prop_removal != 0 ? 1 : p.compact * 2)));
  *plan = p;
  return TB_OK;
}

// tb_config.fixpoint = 3: the engine chooses (same fixpoint, same tree either way)
void resolve_fixpoint(tb_config* cfg, int32_t n_props) {
  if (cfg->fixpoint == 3) cfg->fixpoint = n_props >= TB_AUTO_EVENT_MIN_PROPS ? 2 : 1;
}

int validate_network(int32_t n_vars, const tb_itv* store, int32_t n_props, const tb_prop* props) {
  if (n_vars < 0 || n_props < 0 || (n_vars > 0 && !store) || (n_props > 0 && !props)) return fail(TB_ERR_INVALID, "null or negative-sized network");
  if ((size_t)n_props > (size_t)0xfffe * 64) return fail(TB_ERR_INVALID, "more than 4 194 176 propagators: slice ids are 16 bits wide in the adjacency records");
  for (int32_t i = 0; i < n_props; ++i) {
    const tb_prop& p = props[i];
    if (p.op < 0 || p.op >= TB_NUM_OPS) return fail(TB_ERR_INVALID, "propagator " + std::to_string(i) + ": unknown operator");
    if (p.x < 0 || p.x >= n_vars || p.y < 0 || p.y >= n_vars || p.z < 0 || p.z >= n_vars) return fail(TB_ERR_INVALID, "propagator " + std::to_string(i) + ": variable out of range");
  }
  return TB_OK;
}

// Variable -> slices adjacency of the event-driven fixpoint: slice s = propagators [64 s, 64 s + 64).  Constants never change,
// so they get an empty list.  Each reader comes with its INTEREST in the two kinds of event a variable can undergo (1: its
// lower bound was raised, 2: its upper bound was lowered): `y <= z` with a constant-true truth value can only narrow
// something after y.lb rose or z.ub fell, `y > z` after y.ub fell or z.lb rose; every other propagator reacts to both.  A
// narrowing wakes up only the readers interested in it.
struct Reader { int slice, interest; };
struct Adjacency {
  std::vector<std::vector<Reader>> lists;  // per variable: the slices reading it (ascending, no duplicates)
};

inline int interest_of(int cls, int operand) {  // operand: 0 x, 1 y, 2 z
  if (cls == K_LEQ_T) return operand == 1 ? 1 : (operand == 2 ? 2 : 3);
  if (cls == K_LEQ_F) return operand == 1 ? 2 : (operand == 2 ? 1 : 3);
  return 3;
}

Adjacency build_adjacency(int32_t n_vars, int32_t n_props, const tb_prop* props, const std::vector<char>& is_const, const std::vector<int>& value, bool filter) {
  Adjacency a;
  a.lists.resize((size_t)n_vars);
  for (int32_t i = 0; i < n_props; ++i) {
    if (props[i].op < 0) continue;  // idle padding (to_internal)
    const int s = i / 64;
    const bool xc = is_const[(size_t)props[i].x] != 0;
    const int cls = class_of(props[i].op, xc, xc ? value[(size_t)props[i].x] : 0);
    const int vs[3] = {props[i].x, props[i].y, props[i].z};
    for (int k = 0; k < 3; ++k) {
      const int v = vs[k];
      if (is_const[(size_t)v]) continue;
      const int in = filter ? interest_of(cls, k) : 3;
      std::vector<Reader>& l = a.lists[(size_t)v];
      if (!l.empty() && l.back().slice == s) l.back().interest |= in; else l.push_back(Reader{s, in});
    }
  }
  return a;
}

// Rewrite the caller's bytecodes into the engine's packed records (propagators.hpp): word0 = pack-time class (bits 0-3) |
// "a narrowing of operand k has a reader outside the record's slice" bits 4-9 | original op << 12 | classes present in the 64-record slice << 16.
// An operand that is a constant kept out of the slab (internal id >= n_slab, Layout) is replaced by its value with the sign bit set.
inline int operand_field(int v, int n_slab, const std::vector<int>& value) {
  return v >= n_slab ? (int)(0x80000000u | ((unsigned)value[(size_t)v] & 0x7fffffffu)) : v;
}
inline bool field_is_value(int f) { return f < 0; }
inline int field_value(int f) { return (int)((unsigned)f << 1) >> 1; }
std::vector<int4> pack_props(int32_t n_props, const tb_prop* props, const std::vector<char>& is_const, const std::vector<int>& value,
                             const Adjacency& adj, int n_int, int n_slab) {
  // padded to whole slices with idle records (never narrow, always entailed): the kernels may load any lane of a slice
  std::vector<int4> out(((size_t)n_props + 63) / 64 * 64, make_int4(K_LEQ_T, 0, 0, 0));
  for (int32_t base = 0; base < n_props; base += 64) {
    int32_t end = std::min(n_props, base + 64);
    for (int32_t i = base; i < end; ++i) if (props[i].op < 0) { end = i; break; }  // idle padding fills the rest of the slice
    const int s = base / 64;
    int present = 0;
    for (int32_t i = base; i < end; ++i) {
      const tb_prop& p = props[i];
      const bool xc = is_const[(size_t)p.x] != 0;
      const int cls = class_of(p.op, xc, xc ? value[(size_t)p.x] : 0);
      present |= 1 << cls;
      int report = 0;  // bits 2k, 2k + 1: a narrowing of operand k (lower bound raised / upper bound lowered) has a reader outside this slice
      const int vs[3] = {p.x, p.y, p.z};
      for (int k = 0; k < 3; ++k) {
        const std::vector<Reader>& l = adj.lists[(size_t)vs[k]];
        if (!(l.empty() || (l.size() == 1 && l[0].slice == s))) report |= 3 << (2 * k);
      }
      out[(size_t)i] = make_int4(cls | (report << 4) | (p.op << 12), operand_field(p.x, n_slab, value), operand_field(p.y, n_slab, value), operand_field(p.z, n_slab, value));
    }
    // operand kinds of the slice (bits 26-31, two per operand): 1 all integer variables, 2 all Booleans of the COMPACT layout
    // (internal id >= n_int), 3 all constants, 0 mixed.  The event kernels have dedicated runs for the commonest signatures.
    unsigned kinds = 0;
    for (int k = 0; k < 3; ++k) {
      int seen = 0;
      for (int32_t i = base; i < end; ++i) {
        const int v = k == 0 ? props[i].x : (k == 1 ? props[i].y : props[i].z);
        seen |= is_const[(size_t)v] ? 4 : (v >= n_int ? 2 : 1);
      }
      const unsigned code = seen == 1 ? 1u : (seen == 2 ? 2u : (seen == 4 ? 3u : 0u));
      kinds |= code << (2 * k);
    }
    // The joint evaluation of a channelling slice takes the lanes between two changes of y for a group (pack_succ hands every
    // lane its group's first and last lane): records sorted by y guarantee it; a slice where some y comes back after another one
    // (record sort disabled by a test knob) is left to the generic run.
    if (((unsigned)present | (kinds << 10)) == KEY_EQR_BIC) {
      std::vector<int> seen_y;
      bool contiguous = true;
      for (int32_t i = base; i < end && contiguous; ++i) {
        if (i > base && props[i].y == props[i - 1].y) continue;
        if (std::find(seen_y.begin(), seen_y.end(), props[i].y) != seen_y.end()) contiguous = false;
        seen_y.push_back(props[i].y);
      }
      if (!contiguous) kinds = 0;
    }
    // bit 11: no truth variable x occurs twice in the slice (the joint evaluation of `b_i = (y = k_i)` slices is then complete in one pass)
    int distinct_x = 1;
    {
      std::vector<int> xs;
      for (int32_t i = base; i < end; ++i) xs.push_back(props[i].x);
      std::sort(xs.begin(), xs.end());
      if (std::adjacent_find(xs.begin(), xs.end()) != xs.end()) distinct_x = 0;
    }
    // bit 15 (channelling slices `b = (y = k)` only): inside every group of records sharing y the constants are consecutive
    // integers in lane order -- the bounds of y then move over excluded values with a bit scan (kernels.hpp: KEY_EQR_BIC)
    int dense = 0;
    if (((unsigned)present | (kinds << 10)) == KEY_EQR_BIC) {
      dense = 1;
      for (int32_t a = base; a < end && dense;) {
        const long long k0 = value[(size_t)props[a].z];
        if (k0 > 0x7fffff00ll) dense = 0;
        int32_t b = a + 1;
        for (; b < end && props[b].y == props[a].y; ++b)
          if ((long long)value[(size_t)props[b].z] != k0 + (b - a)) dense = 0;
        a = b;
      }
    }
    for (int32_t i = base; i < end; ++i) out[(size_t)i].x |= (present << 16) | (int)(kinds << 26) | (distinct_x << 11) | (dense << 15);  // same in the 64 records of a slice
  }
  return out;
}

// Per variable: 32-byte adjacency record of the event-driven fixpoint (device_types.hpp: DevProblem::var_adj) + overflow list.
// 16 halfwords: [0] number of reader slices, [1..11] the first eleven, [12..13] their interests (2 bits each), [14..15] offset
// of the others in the overflow list, whose entries are slice | interest << 30.
void pack_var_adj(const Adjacency& adj, std::vector<int4>* heads, std::vector<int>* rest) {
  const size_t V = adj.lists.size();
  std::vector<unsigned> w(std::max<size_t>(1, V) * 8, 0);
  rest->clear();
  for (size_t v = 0; v < V; ++v) {
    const std::vector<Reader>& l = adj.lists[v];
    unsigned* h = w.data() + v * 8;
    unsigned short hw[16] = {0};
    hw[0] = (unsigned short)std::min<size_t>(l.size(), 0xffffu);
    unsigned interest = 0;
    for (size_t k = 0; k < l.size() && k < 11; ++k) { hw[1 + k] = (unsigned short)l[k].slice; interest |= (unsigned)l[k].interest << (2 * k); }
    for (int k = 0; k < 6; ++k) h[k] = (unsigned)hw[2 * k] | ((unsigned)hw[2 * k + 1] << 16);
    h[6] = interest;
    h[7] = (unsigned)rest->size();
    for (size_t k = 11; k < l.size(); ++k) rest->push_back(l[k].slice | (l[k].interest << 30));
  }
  if (rest->empty()) rest->push_back(0);
  heads->resize(std::max<size_t>(1, V) * 2);
  std::memcpy(heads->data(), w.data(), heads->size() * sizeof(int4));
}

// Chains (r04): the channelling records `b_i = (y = k_i)` of one variable y, sorted by k, are a CHAIN of consecutive records over one or more
// slices (a 500-value index of an element constraint: eight slices).  When the chain's constants are consecutive integers, the record of value v
// is first + (v - k_first): a narrowing of y from [a, b] to [a', b'] concerns exactly the chain slices holding the values a..a' and b'..b -- the
// others hold values that are still inside the bounds (nothing to do) or were outside before (their b are false already).  For an eligible chain
// the chain's own slices are therefore not listed among y's readers in the channelling records: the lane that writes y marks them by that
// arithmetic (kernels.hpp: KEY_EQR_BIC run, slice_info 0x400).  Everybody else who narrows y still wakes every reader through var_adj.
struct Chains {
  std::vector<int> first, last;  // per variable: record range of its eligible chain, -1 = none
  std::vector<char> slice_ok;    // per slice: every group of the slice belongs to an eligible chain (the kernel's flag is per slice)
  bool in_chain(int y, int slice) const { return first[(size_t)y] >= 0 && slice >= first[(size_t)y] / 64 && slice <= last[(size_t)y] / 64; }
};
Chains find_chains(int32_t n_vars, int32_t n_props, const tb_prop* props, const std::vector<int4>& records, const std::vector<int>& value, const tb_itv* root, bool enable) {
  Chains c;
  c.first.assign((size_t)std::max(1, n_vars), -1); c.last.assign((size_t)std::max(1, n_vars), -1);
  const int n_slices = (n_props + 63) / 64;
  c.slice_ok.assign((size_t)std::max(1, n_slices), 0);
  if (!enable || root == nullptr) return c;
  std::vector<int> count((size_t)std::max(1, n_vars), 0);
  std::vector<char> bad((size_t)std::max(1, n_vars), 0);
  auto chain_slice = [&](int sl) { return (size_t)sl * 64 < records.size() && ((unsigned)records[(size_t)sl * 64].x >> 16) == KEY_EQR_BIC && (((unsigned)records[(size_t)sl * 64].x >> 15) & 1u); };
  for (int sl = 0; sl < n_slices; ++sl) {
    if (!chain_slice(sl)) continue;
    const int32_t base = sl * 64;
    int32_t end = std::min(n_props, base + 64);
    for (int32_t i = base; i < end; ++i) if (props[i].op < 0) { end = i; break; }
    for (int32_t i = base; i < end; ++i) {
      const int y = props[i].y;
      if (c.first[(size_t)y] < 0) c.first[(size_t)y] = i;
      if (c.last[(size_t)y] >= 0 && c.last[(size_t)y] != i - 1) bad[(size_t)y] = 1;  // the records of one y are not contiguous
      c.last[(size_t)y] = i;
      count[(size_t)y]++;
    }
  }
  for (int y = 0; y < n_vars; ++y) {
    if (c.first[(size_t)y] < 0) continue;
    const int f = c.first[(size_t)y], l = c.last[(size_t)y];
    const long long k0 = value[(size_t)props[f].z];
    bool ok = !bad[(size_t)y] && count[(size_t)y] == l - f + 1;
    for (int i = f; i <= l && ok; ++i) ok = (long long)value[(size_t)props[i].z] == k0 + (i - f);
    // (a root domain reaching far beyond the chain's values would make the range arithmetic wake slices of other chains, or overflow)
    ok = ok && (long long)root[y].lb >= k0 - 64 && (long long)root[y].ub <= k0 + (l - f) + 64;
    if (!ok) { c.first[(size_t)y] = c.last[(size_t)y] = -1; }
  }
  // the flag the kernel sees is per slice: a slice qualifies when all its groups do, a chain when all its slices do
  for (bool changed = true; changed;) {
    changed = false;
    for (int sl = 0; sl < n_slices; ++sl) {
      bool ok = chain_slice(sl);
      const int32_t base = sl * 64;
      for (int32_t i = base; ok && i < std::min(n_props, base + 64) && props[i].op >= 0; ++i) ok = c.first[(size_t)props[i].y] >= 0;
      c.slice_ok[(size_t)sl] = ok ? 1 : 0;
    }
    for (int y = 0; y < n_vars; ++y) {
      if (c.first[(size_t)y] < 0) continue;
      bool ok = true;
      for (int sl = c.first[(size_t)y] / 64; sl <= c.last[(size_t)y] / 64; ++sl) ok = ok && c.slice_ok[(size_t)sl];
      if (!ok) { c.first[(size_t)y] = c.last[(size_t)y] = -1; changed = true; }
    }
  }
  return c;
}

// Successor slices of each record's operands for the event-driven fixpoint (device_types.hpp: DevProblem::succ): x, y, z =
// up to two OTHER slices reading the operand (16-bit ids, 0xffff = none); w: bits 4 + 2 (2 k + j) = interest of the j-th packed
// successor of operand k; bit k = the slices interested in a RAISED LOWER bound of operand k do not all fit (walk var_adj for
// that event), bit 16 + k = the same for a LOWERED UPPER bound.  The two slots go to the event they can cover completely: a
// Boolean of wordpress7_500 typically has five readers, all of them interested in "became true" and one in "became false" --
// per reader that is three slots too few, per event the common one needs no memory access at all.
// Channelling slices (KEY_EQR_BIC, evaluated jointly by the lanes that share y): every lane of a group knows what happened to y,
// so y's readers are dealt out over the group's lanes, two each (bit 20 of w: "y is reported by every lane of its group"), when
// they all fit; a y with ten readers then needs no walk either.
//
// Implication slices of the COMPACT layout (KEY_LEQT_BB: `b1 <= b2` on two 2-bit Booleans, `lean` = their store words fit 16 bits): the run
// works from this record alone.  x = LDS word index (from the start of the store slab) of y's Boolean word | z's << 16; bits 20-23 / 24-27 of
// w = y's / z's bit position / 2.  y can only ever be narrowed from above by such a propagator, z only from below, so the y slots hold the
// readers interested in "upper bound lowered", the z slots those interested in "lower bound raised" -- already filtered: no interest check
// at run time -- and bit 17 / bit 2 say that more than two were interested (walk the variable's adjacency record).
std::vector<int4> pack_succ(int32_t n_props, const tb_prop* props, const Adjacency& adj, const std::vector<int4>& records, const std::vector<int>& value, bool deal_groups,
                            const Chains& chains, int n_int = 0, bool lean = false, int bool_word0 = 0, bool cond_wake = false, const Layout* c8 = nullptr) {
  std::vector<int4> out(((size_t)n_props + 63) / 64 * 64, make_int4(-1, -1, -1, 0));
  // Conditional wake-up of a chain by "b became false" (lean implication records, `cond_wake`): the rule of `b = (y = k)` reacts to a false b only when k
  // sits on a bound of y, and the lane that lowers b can test that itself (one LDS read) -- when b is the truth variable of exactly one chain record.
  std::vector<int> chan_rec;  // per variable: its only record in a channelling slice, -1 none, -2 several
  if (lean && cond_wake) {
    chan_rec.assign(adj.lists.size(), -1);
    for (int32_t i = 0; i < n_props; ++i) {
      if (props[i].op < 0 || ((unsigned)records[(size_t)(i / 64) * 64].x >> 16) != KEY_EQR_BIC) continue;
      int& r = chan_rec[(size_t)props[i].x];
      r = r == -1 ? i : -2;
    }
  }
  std::vector<int> dealt((size_t)n_props, -1);  // index inside its group of a lane whose y slots are dealt
  if (deal_groups)
    for (int32_t base = 0; base < n_props; base += 64) {
      if (((unsigned)records[(size_t)base].x >> 16) != KEY_EQR_BIC) continue;
      int32_t end = std::min(n_props, base + 64);
      for (int32_t i = base; i < end; ++i) if (props[i].op < 0) { end = i; break; }
      for (int32_t a = base; a < end;) {
        int32_t b = a + 1;
        while (b < end && props[b].y == props[a].y) ++b;
        size_t others = 0;
        // (the chain's own slices are left out only when THIS slice wakes them by value range -- slice_info 0x400, chains.slice_ok: a y may have records in a
        //  non-dense channelling slice next to its eligible chain, and that slice's run wakes nobody by arithmetic)
        const bool by_range = chains.slice_ok[(size_t)(base / 64)] != 0;
        for (const Reader& r : adj.lists[(size_t)props[a].y]) others += (r.slice != base / 64 && !(by_range && chains.in_chain(props[a].y, r.slice))) ? 1 : 0;
        if (others > 2 && others <= 2 * (size_t)(b - a))
          for (int32_t i = a; i < b; ++i) dealt[(size_t)i] = i - a;
        a = b;
      }
    }
  for (int32_t i = 0; i < n_props; ++i) {
    if (props[i].op < 0) continue;  // idle padding: nothing to wake
    const int s = i / 64;
    const int vs[3] = {props[i].x, props[i].y, props[i].z};
    unsigned packed[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
    int flags = 0;
    for (int k = 0; k < 3; ++k) {
      if (k == 1 && dealt[(size_t)i] >= 0) {
        unsigned o[2] = {0xffffu, 0xffffu};
        int in[2] = {0, 0}, n = 0, idx = 0;
        for (const Reader& r : adj.lists[(size_t)vs[k]]) {
          if (r.slice == s || (chains.slice_ok[(size_t)s] && chains.in_chain(vs[k], r.slice))) continue;
          if (idx / 2 == dealt[(size_t)i]) { o[n] = (unsigned)r.slice; in[n] = r.interest; ++n; }
          ++idx;
        }
        packed[k] = (o[1] << 16) | o[0];
        flags |= (in[0] << (4 + 2 * (2 * k))) | (in[1] << (4 + 2 * (2 * k + 1))) | (1 << 20);
        continue;
      }
      // (y of a record of an eligible chain: the chain's slices are woken by value range, see Chains)
      const bool chained = k == 1 && chains.slice_ok[(size_t)s] && chains.first[(size_t)vs[k]] >= 0;
      std::vector<Reader> others;
      for (const Reader& r : adj.lists[(size_t)vs[k]]) if (r.slice != s && !(chained && chains.in_chain(vs[k], r.slice))) others.push_back(r);
      int n_lb = 0, n_ub = 0;
      for (const Reader& r : others) { n_lb += (r.interest & 1) ? 1 : 0; n_ub += (r.interest & 2) ? 1 : 0; }
      // which event gets the slots when both do not fit: the one that fits; the upper bound if both would
      int first = 3;
      if (others.size() > 2) first = n_ub <= 2 ? 2 : (n_lb <= 2 ? 1 : 0);
      unsigned o[2] = {0xffffu, 0xffffu};
      int in[2] = {0, 0};
      int n = 0;
      for (int pass = 0; pass < 2 && first != 0; ++pass)
        for (const Reader& r : others) {
          const bool wanted = (r.interest & first) != 0;
          if ((pass == 0) != wanted || n >= 2) continue;
          o[n] = (unsigned)r.slice; in[n] = r.interest; ++n;
        }
      auto covered = [&](int ev) {  // every other reader interested in `ev` sits in a slot
        for (const Reader& r : others) {
          if (!(r.interest & ev)) continue;
          bool found = false;
          for (int j = 0; j < n; ++j) found |= o[j] == (unsigned)r.slice;
          if (!found) return false;
        }
        return true;
      };
      if (!covered(1)) flags |= 1 << k;
      if (!covered(2)) flags |= 1 << (16 + k);
      packed[k] = (o[1] << 16) | o[0];
      flags |= (in[0] << (4 + 2 * (2 * k))) | (in[1] << (4 + 2 * (2 * k + 1)));
    }
    out[(size_t)i] = make_int4((int)packed[0], (int)packed[1], (int)packed[2], flags);
    if (lean && ((unsigned)records[(size_t)(i / 64) * 64].x >> 16) == KEY_LEQT_BB) {
      unsigned slots[3] = {0, 0xffffffffu, 0xffffffffu};
      int fl = 0;
      for (int k = 1; k < 3; ++k) {
        const int ev = k == 1 ? 2 : 1;  // y: upper bound lowered; z: lower bound raised
        unsigned o[2] = {0xffffu, 0xffffu};
        int n = 0;
        bool over = false;
        for (const Reader& r : adj.lists[(size_t)vs[k]]) {
          if (r.slice == s || !(r.interest & ev)) continue;
          if (n < 2) o[n++] = (unsigned)r.slice; else over = true;
        }
        if (over) { o[0] = o[1] = 0xffffu; fl |= k == 1 ? (1 << 17) : (1 << 2); }
        slots[k] = (o[1] << 16) | o[0];
        // y's only reader interested in "became false" is the chain record `y = (Y = kv)`: slot 0 = that slice, slot 1 = Y's index among the
        // integers of the slab, bits 3-16 and 28-29 of w = kv (16 bits, signed), bit 30 = the flag; the run marks the slice only when kv is on a
        // bound of Y at that moment (whoever moves a bound of Y onto kv later wakes that slice for it)
        if (k == 1 && n == 1 && !over && !chan_rec.empty() && chan_rec[(size_t)vs[1]] >= 0) {
          const int r = chan_rec[(size_t)vs[1]];
          const int Y = props[r].y;
          int kv = value[(size_t)props[r].z];
          if (c8 != nullptr && Y >= c8->n_wide && Y < c8->n_int) kv -= c8->base[(size_t)Y];  // COMPACT8: the run compares with the bytes of a narrow Y as they are stored
          if ((unsigned)(r / 64) == o[0] && Y >= 0 && Y < n_int && Y < 0xffff && kv >= -32768 && kv <= 32767) {
            slots[1] = o[0] | ((unsigned)Y << 16);
            fl |= (int)((((unsigned)kv & 0x3fffu) << 3) | ((((unsigned)kv >> 14) & 3u) << 28) | (1u << 30));
          }
        }
      }
      const int by = vs[1] - n_int, bz = vs[2] - n_int;
      slots[0] = (unsigned)(bool_word0 + (by >> 4)) | ((unsigned)(bool_word0 + (bz >> 4)) << 16);
      fl |= ((by & 15) << 20) | ((bz & 15) << 24);
      out[(size_t)i] = make_int4((int)slots[0], (int)slots[1], (int)slots[2], fl);
    }
  }
  // Idle lanes of a lean implication slice (class padding, the tail of the last slice): the run forms both word addresses from the record before
  // it masks anything, so they point at the first Boolean word of the slab (never written through: the lane is inactive) instead of 0xffff words
  // past its start -- outside the workgroup's LDS allocation, or past the end of g_store for the last workgroups of a compact slab in global memory.
  if (lean)
    for (size_t base = 0; base < out.size(); base += 64) {
      if (base >= records.size() || ((unsigned)records[base].x >> 16) != KEY_LEQT_BB) continue;
      for (size_t i = base; i < base + 64; ++i)
        if (i >= (size_t)n_props || props[i].op < 0) out[i] = make_int4((int)((unsigned)bool_word0 | ((unsigned)bool_word0 << 16)), -1, -1, 0);
    }
  // Reified comparisons against a constant with a Boolean truth variable (KEY_EQR_BIC, KEY_LEQR_BIC: dedicated runs that never
  // report anything about z): the z slots, useless for a constant, carry its VALUE, and in the channelling slices bits 21-26 /
  // 27-31 + 19 of w carry the first / last lane of the record's group -- nothing left to gather or to shuffle at run time.
  for (int32_t base = 0; base < n_props; base += 64) {
    const unsigned key = (unsigned)records[(size_t)base].x >> 16;
    if (key != KEY_EQR_BIC && key != KEY_LEQR_BIC) continue;
    int32_t end = std::min(n_props, base + 64);
    for (int32_t i = base; i < end; ++i) if (props[i].op < 0) { end = i; break; }
    for (int32_t a = base; a < end;) {
      int32_t b = a + 1;
      while (b < end && key == KEY_EQR_BIC && props[b].y == props[a].y) ++b;
      for (int32_t i = a; i < b; ++i) {
        int4& r = out[(size_t)i];
        r.z = value[(size_t)props[i].z];
        const unsigned g0 = (unsigned)(a - base), g1 = (unsigned)(b - 1 - base);
        r.w = (int)(((unsigned)r.w & 0x0017ffffu) | (g0 << 21) | ((g1 & 31u) << 27) | ((g1 >> 5) << 19));
      }
      a = b;
    }
  }
  return out;
}

// (r05) Conditional wake-up of an element constraint's implications by "c became false" (DevProblem::cond2; kernels.hpp: the block before mark_successors).
// An element constraint val = table[idx] is lowered to channelling records b_i = (idx = i), c_v = (val = v) and implications b_i <= c_table[i].  When a channelling lane makes
// c_v false, the implication slices holding b_i <= c_v run to make those b_i false -- and find them false already whenever the positions of v lie outside idx's bounds
// (idx's own chain has seen to it, or will before the fixpoint ends), or idx is assigned (the one b that is not false is true, and ITS becoming true wakes the implications).
// Eligible: c is the truth variable of exactly this channelling record; every reader slice of c interested in "upper bound lowered" is a lean implication slice; every
// implication with z = c has for y the truth variable of exactly one channelling record of ONE index variable, an integer of the slab; positions within 0..65535.
std::vector<int2> pack_cond2(int32_t n_props, const tb_prop* props, const Adjacency& adj, const std::vector<int4>& records, const std::vector<int>& value, const Layout& lay,
                             std::vector<char>* slice_has) {
  const size_t padded = ((size_t)n_props + 63) / 64 * 64;
  std::vector<int2> out(std::max<size_t>(padded, 64), make_int2(-1, 0));
  slice_has->assign(std::max<size_t>(1, padded / 64), 0);
  const size_t V = adj.lists.size();
  auto key_of_slice = [&](int sl) { return (size_t)sl * 64 < records.size() ? (unsigned)records[(size_t)sl * 64].x >> 16 : 0u; };
  std::vector<int> chan_rec(V, -1);
  std::vector<std::vector<int>> impl_of_z(V);
  for (int32_t i = 0; i < n_props; ++i) {
    if (props[i].op < 0) continue;
    const unsigned key = key_of_slice(i / 64);
    if (key == KEY_EQR_BIC) { int& r = chan_rec[(size_t)props[i].x]; r = r == -1 ? i : -2; }
    else if (key == KEY_LEQT_BB) impl_of_z[(size_t)props[i].z].push_back(i);
  }
  for (int32_t i = 0; i < n_props; ++i) {
    if (props[i].op < 0 || key_of_slice(i / 64) != KEY_EQR_BIC) continue;
    const int c = props[i].x;
    // (a table-like fan-in only: where c has a single implication -- trains15's b1 <= b2 between two truth variables -- the test costs what the run it spares does: measured -1 %)
    if (chan_rec[(size_t)c] != i || impl_of_z[(size_t)c].size() < 4) continue;
    bool ok = true;
    for (const Reader& r : adj.lists[(size_t)c])
      if (r.slice != i / 64 && (r.interest & 2) && key_of_slice(r.slice) != KEY_LEQT_BB) { ok = false; break; }
    int idx = -1, pmin = 0x7fffffff, pmax = -0x7fffffff;
    for (size_t q = 0; q < impl_of_z[(size_t)c].size() && ok; ++q) {
      const int b = props[impl_of_z[(size_t)c][q]].y;
      const int rb = chan_rec[(size_t)b];
      if (rb < 0) { ok = false; break; }
      const int I = props[rb].y, pos = value[(size_t)props[rb].z];
      if (idx == -1) idx = I; else if (idx != I) ok = false;
      pmin = std::min(pmin, pos); pmax = std::max(pmax, pos);
    }
    if (!ok || idx < 0 || idx >= lay.n_int || pmin < 0 || pmax > 0xffff) continue;
    out[(size_t)i] = make_int2(lay.ref(idx), pmin | (pmax << 16));
    (*slice_has)[(size_t)(i / 64)] = 1;
  }
  return out;
}

// Per slice, for the event kernels (read with scalar loads): x = word0 of the slice's first record (its slice-uniform part: class set,
// operand kinds, flags), y = lanes that hold a propagator | 0x100 when the slice's successor records carry the lean implication encoding.
// 0x200: the slice has one class (not the heavy one) and every operand of every record has a finite root domain within +-2^29 -- the plain
// layout then runs it with lean_plain_run (kernels.hpp): 32-bit sums and differences of two such bounds cannot overflow, anywhere in the tree.
// 0x400: the slice's channelling groups belong to eligible chains (Chains): the lane that writes y wakes the chain's other slices by value range.
std::vector<int2> slice_infos(const std::vector<int4>& packed, const std::vector<int>& real, int n_slices, bool lean, const tb_itv* root, bool plain_lean, const Chains& chains) {
  std::vector<int2> info((size_t)std::max(1, n_slices), make_int2(0, 0));
  for (int s = 0; s < n_slices; ++s) {
    const int w0 = (size_t)s * 64 < packed.size() ? packed[(size_t)s * 64].x : 0;
    const bool is_lean = lean && ((unsigned)w0 >> 16) == KEY_LEQT_BB;
    const int n_real = (size_t)s < real.size() ? real[(size_t)s] : 0;
    const unsigned classes = ((unsigned)w0 >> 16) & CLASS_SET_MASK;
    bool finite = plain_lean && root != nullptr && n_real > 0 && classes != 0 && (classes & (classes - 1)) == 0 && classes != (1u << K_HEAVY);
    for (int l = 0; l < n_real && finite; ++l) {
      const int4 r = packed[(size_t)s * 64 + (size_t)l];
      for (int v : {r.y, r.z, r.w}) {
        const tb_itv d = field_is_value(v) ? tb_itv{field_value(v), field_value(v)} : root[v];
        if (d.lb < -(1 << 29) || d.ub > (1 << 29)) finite = false;
      }
    }
    // 0x800 (r05): every record is `x = y * z` over non-negative finite operands whose products stay below 2^30 -- the lean product run (kernels.hpp: K_MUL_NN)
    bool mul_nn = plain_lean && root != nullptr && n_real > 0 && classes == (1u << K_HEAVY) && !std::getenv("TB_NO_LEAN_MUL");
    for (int l = 0; l < n_real && mul_nn; ++l) {
      const int4 r = packed[(size_t)s * 64 + (size_t)l];
      tb_itv d[3];
      int k = 0;
      for (int v : {r.y, r.z, r.w}) d[k++] = field_is_value(v) ? tb_itv{field_value(v), field_value(v)} : root[v];
      mul_nn = ((r.x >> 12) & 7) == TB_MUL && d[0].lb >= 0 && d[1].lb >= 0 && d[2].lb >= 0 && d[0].ub <= (1 << 30) && d[1].ub <= (1 << 30) && d[2].ub <= (1 << 30) &&
               (long long)d[1].ub * (long long)d[2].ub <= (1ll << 30);
    }
    // 0x1000 (r05): a lean implication slice in which no variable is the y of one record and the z of another: a narrowing pass enables no other lane, the run needs no
    // confirmation pass (kernels.hpp: lean implication run)
    bool single = is_lean && n_real > 0 && !std::getenv("TB_NO_SINGLE_IMPL");
    if (single) {
      std::vector<int> ys, zs;
      for (int l = 0; l < n_real; ++l) { ys.push_back(packed[(size_t)s * 64 + (size_t)l].z); zs.push_back(packed[(size_t)s * 64 + (size_t)l].w); }
      std::sort(ys.begin(), ys.end()); std::sort(zs.begin(), zs.end());
      std::vector<int> both;
      std::set_intersection(ys.begin(), ys.end(), zs.begin(), zs.end(), std::back_inserter(both));
      single = both.empty();
    }
    const bool chain_ok = (size_t)s < chains.slice_ok.size() && chains.slice_ok[(size_t)s];
    info[(size_t)s] = make_int2(w0, n_real | (is_lean ? 0x100 : 0) | (finite ? 0x200 : 0) | (chain_ok ? 0x400 : 0) | (mul_nn ? 0x800 : 0) | (single ? 0x1000 : 0));
  }
  return info;
}

// Constants of a batch of stores: variables that are the same finite singleton in every store.
void find_constants(int32_t n_vars, int32_t n_stores, const tb_itv* stores, std::vector<char>* is_const, std::vector<int>* value) {
  is_const->assign((size_t)n_vars, 1);
  value->assign((size_t)n_vars, 0);
  for (int32_t v = 0; v < n_vars; ++v) {
    const tb_itv d0 = stores[v];
    if (d0.lb != d0.ub || d0.lb == TB_NINF || d0.lb == TB_PINF) { (*is_const)[(size_t)v] = 0; continue; }
    (*value)[(size_t)v] = d0.lb;
    for (int32_t s = 1; s < n_stores; ++s) {
      const tb_itv d = stores[(size_t)s * (size_t)n_vars + (size_t)v];
      if (d.lb != d0.lb || d.ub != d0.ub) { (*is_const)[(size_t)v] = 0; break; }
    }
  }
}

// Kernel selection -> translation unit (kernel_units.hpp): the host shim instantiates no kernel itself.
#define TB_ROUTE(unit, call)                                                                    \
  ((unit) == 1 ? unit1_##call : (unit) == 2 ? unit2_##call : (unit) == 3 ? unit3_##call : (unit) == 4 ? unit4_##call : (unit) == 5 ? unit5_##call : \
   (unit) == 6 ? unit6_##call : (unit) == 7 ? unit7_##call : (unit) == 8 ? unit8_##call : unit9_##call)
int hip_rc(int e, const char* what) {
  if (e == 0) return TB_OK;
  if (e < 0) return fail(TB_ERR_INVALID, std::string(what) + ": no such kernel instantiation");
  if ((hipError_t)e == hipErrorOutOfMemory) return fail(TB_ERR_OOM, std::string(what) + ": " + hipGetErrorString((hipError_t)e));
  return fail(TB_ERR_HIP, std::string(what) + ": " + hipGetErrorString((hipError_t)e));
}
int launch_solve(const KernelSel& k, const KernelGrid& g, const DevProblem& P, const DevProblem* dP, Mailbox* mbox) {
  const int u = kernel_unit(true, k);
  return hip_rc(TB_ROUTE(u, launch_solve(k, g, P, dP, mbox)), "search kernel launch");
}
int launch_prop(const KernelSel& k, const KernelGrid& g, const DevProblem& P, int2* stores, PropagateOut* out, int n_stores) {
  const int u = kernel_unit(false, k);
  return hip_rc(TB_ROUTE(u, launch_prop(k, g, P, stores, out, n_stores)), "propagation kernel launch");
}

// Sets the dynamic-LDS limit of the kernel that will run and returns how many of its workgroups a CU holds
// (register / LDS limited).  A persistent kernel gains nothing from queued workgroups, so the grid is capped.
int prepare_kernel(bool solve, int mem, int tmax, bool event, int opt, int bytes, int threads, int* max_blocks_per_cu) {
  const KernelSel k{mem, tmax, event, opt};
  const int u = kernel_unit(solve, k);
  return hip_rc(TB_ROUTE(u, prepare(k, bytes, threads, max_blocks_per_cu)), "hipFuncSetAttribute / occupancy");
}

// Layout + launch plan of a network.  The COMPACT layout is chosen for the event-driven fixpoint when it brings a
// store that would otherwise sit in global memory into LDS (tb_config.reserved[0]: 0x80000 never, 0x100000 always).
int choose_layout(const tb_config& cfg, const DeviceCaps& caps, int32_t n_vars, int32_t n_stores, const tb_itv* stores, int32_t n_props,
                  Layout* lay, LaunchPlan* plan, int32_t pinned = -1, bool search = true) {
  *lay = make_layout(n_vars, n_stores, stores, false);
  int rc = plan_launch(cfg, caps, *lay, n_props, plan, search);  // (search = false, tb_propagate: no hot tier -- the batch kernel works on plain slabs and would only lose workgroups per CU to it)
  // The sweeps take a compact layout only when it is forced (and never together with entailed-slice removal): measured r03, a sweep
  // evaluates every propagator whatever the layout, and the decode costs more than the extra workgroups bring -- wordpress7_500 WAC1
  // 2.10e6 nodes/s plain (256 x 1024) against 1.81e6 compact (1280 x 256); trains15 6.77e6 against 7.44e6.
  const bool sweeps_opt_in = cfg.fixpoint != 2 && !cfg.entailed_prop_removal && (cfg.reserved[0] & 0x100000);
  if (rc != TB_OK || (cfg.fixpoint != 2 && !sweeps_opt_in) || (cfg.reserved[0] & 0x80000)) return rc;
#ifdef TB_TUNING  // the run census (0x400000) selects a class with bits 28-31: not layout knobs then
  const int lk = (cfg.reserved[0] & 0x400000) ? 0 : cfg.reserved[0];
#else
  const int lk = cfg.reserved[0];
#endif
  const bool outs = !(lk & 0x40000000);  // constants out of the slab (0x40000000: keep them in, A/B runs and tests)
  Layout lc = make_layout(n_vars, n_stores, stores, true, false, outs, pinned);
  if (!lc.compact) return rc;
  LaunchPlan pc;
  if ((rc = plan_launch(cfg, caps, lc, n_props, &pc)) != TB_OK) return rc;
  // COMPACT16 (reserved[0] & 0x10000000 always when eligible, 0x20000000 never): half the bytes per integer variable.  Taken when it
  // puts more workgroups on a CU than COMPACT does, or brings the slab into LDS at all -- trains15: 50 KB -> 29 KB per workgroup.
  if (!(lk & 0x20000000) || (lk & 0x30000000) == 0x30000000) {
    Layout l16 = make_layout(n_vars, n_stores, stores, true, true, outs, pinned);
    LaunchPlan p16;
    if (l16.c16 && plan_launch(cfg, caps, l16, n_props, &p16) == TB_OK) {
      const bool better = (pc.mem_kind == TB_MEM_GLOBAL && p16.mem_kind != TB_MEM_GLOBAL) ||
                          (pc.mem_kind != TB_MEM_GLOBAL && p16.mem_kind != TB_MEM_GLOBAL && p16.blocks_per_cu > pc.blocks_per_cu);
      if (better || (lk & 0x30000000) == 0x10000000) { lc = std::move(l16); pc = p16; }
    }
  }
  // COMPACT8 (event kernels; both COMPACT16 bits set: always when eligible, sign bit: never): two bytes per narrow integer.  Taken when it puts more
  // workgroups on a CU -- trains15: 20.7 KB -> 11.6 KB per slab, seven four-wave workgroups -> eleven two-wave ones.
  if (cfg.fixpoint == 2 && !(lk & (int)0x80000000u)) {
    Layout l8 = make_layout(n_vars, n_stores, stores, true, false, outs, pinned, true);
    LaunchPlan p8;
    if (l8.c8 && plan_launch(cfg, caps, l8, n_props, &p8) == TB_OK && p8.mem_kind == TB_MEM_STORE_SHARED && p8.tmax <= 256) {
      const bool better = pc.mem_kind == TB_MEM_GLOBAL || p8.blocks_per_cu * p8.threads > pc.blocks_per_cu * pc.threads || (p8.threads < pc.threads && p8.blocks_per_cu * p8.threads * 4 >= pc.blocks_per_cu * pc.threads * 3);
      if (better || (lk & 0x30000000) == 0x30000000) { lc = std::move(l8); pc = p8; }
    }
  }
  const bool forced = (cfg.reserved[0] & 0x100000) != 0;
  // ... and when both layouts end up in global memory, the compact slab is taken if it is at most two thirds of the plain one: the
  // workgroups' slabs then stay closer to the CUs (trains15: 2.04e7 nodes/s against 1.93e7)
  const bool smaller_in_global = plan->mem_kind == TB_MEM_GLOBAL && pc.mem_kind == TB_MEM_GLOBAL && !cfg.only_global_memory && pc.vext * 3 < plan->vext * 2;
  if (forced || (plan->mem_kind == TB_MEM_GLOBAL && pc.mem_kind != TB_MEM_GLOBAL) || smaller_in_global) { *lay = std::move(lc); *plan = pc; }
  return TB_OK;
}

// The caller's network in the internal numbering of a layout.
struct InternalNet {
  std::vector<tb_itv> store;  // first store of the batch, internal order
  std::vector<tb_prop> props;
};
// Class of a record as the record sort sees it (constants = singletons of the first store).
inline int sort_class(const tb_prop& q, const tb_itv* store) {
  const tb_itv d = store[q.x];
  const bool xc = d.lb == d.ub && d.lb != TB_NINF && d.lb != TB_PINF;
  return class_of(q.op, xc, xc ? d.lb : 0);
}
// Records of the class-sorted array when every class but the last is padded to whole slices (to_internal with `pad`).
int32_t padded_count(int32_t n_props, const tb_prop* props, const tb_itv* store) {
  long long cnt[16] = {0};
  for (int32_t i = 0; i < n_props; ++i) cnt[sort_class(props[i], store) & 15]++;
  int last = -1;
  for (int c = 0; c < 16; ++c) if (cnt[c]) last = c;
  long long total = 0;
  for (int c = 0; c < 16; ++c) total += c == last ? cnt[c] : (cnt[c] + 63) / 64 * 64;
  return (int32_t)std::min<long long>(total, 0x7fffffff);
}
InternalNet to_internal(const Layout& L, const tb_itv* store, int32_t n_props, const tb_prop* props, bool keep_order, bool event = false, bool pad = false, long window = 0) {
  InternalNet n;
  n.store.resize((size_t)L.n_vars);
  for (int v = 0; v < L.n_vars; ++v) n.store[(size_t)L.perm[(size_t)v]] = store[v];
  n.props.resize((size_t)n_props);
  for (int32_t i = 0; i < n_props; ++i) {
    const tb_prop& q = props[i];
    n.props[(size_t)i] = tb_prop{q.op, L.perm[(size_t)q.x], L.perm[(size_t)q.y], L.perm[(size_t)q.z]};
  }
  // Record order is the engine's choice (the fixpoint does not depend on it): a stable sort by pack-time class (and by
  // operator inside the heavy class) makes almost every 64-record slice class-pure, so a slice evaluation runs one
  // class body instead of two or three.  Measured on wordpress7_500 (583 of 718 slices mixed `b = (y = z)` with
  // `y <= z`): WAC1 2.25e11 -> 3.3e11 propagations/s and 1.09 -> 1.5e6 nodes/s, event mode 4.9 -> 6.5e6 nodes/s.
  // The sort is stable, so the records of one constraint stay together inside their class.
  // Not for stores in global memory: those runs are bound by the memory system, not by VALU work, and the caller's
  // order carries locality (the 100k x 500k synthetic network is emitted in topological order: sorted, a node needs
  // 20 % more sweeps and the event worklist twice the evaluations; trains15 on compact slabs in global memory: 2.03e7 nodes/s in the
  // caller's order, 1.96e7 sorted).
  auto key = [&](const tb_prop& q) {
    const tb_itv d = n.store[(size_t)q.x];
    const bool xc = d.lb == d.ub && d.lb != TB_NINF && d.lb != TB_PINF;
    return class_of(q.op, xc, xc ? d.lb : 0) * 16 + q.op;
  };
  if (keep_order) {
    // (r05; `window`, TB_GLOBAL_SORT_WINDOW=W overrides) the caller's order kept at the scale of W records, the class sort applied inside each window: slices become class-pure
    // (two gathers and one comparison for `y <= z` instead of every class body behind selects) while the topological order of the stream survives at window granularity.
    // Synthetic 100k x 500k in workgroup teams, same box (profiles/r05_window_team_ab.txt): the plain sweeps 1.23e11 -> 1.52e11 propagations/s and 1.37e4 -> 2.08e4 nodes/s at
    // W = 1024 (one workgroup's 16 slices; 512: 1.74e4, 2048: 1.86e4, 8192: 1.41e4) -- the team kernel is VALU bound, a mixed slice pays every class body --; the WAC1 sweeps lose
    // (1.78e4 -> 1.57e4 nodes/s: a definition and the constraints on it no longer share a slice, the local passes find less), the event fixpoint too (3.66e4 -> 2.36e4).
    const char* e = std::getenv("TB_GLOBAL_SORT_WINDOW");
    const long W = e != nullptr ? std::atol(e) : window;
    if (W >= 128)
      for (size_t a = 0; a < n.props.size(); a += (size_t)W)
        std::stable_sort(n.props.begin() + (long)a, n.props.begin() + (long)std::min(n.props.size(), a + (size_t)W), [&](const tb_prop& x, const tb_prop& y) { return key(x) < key(y); });
    return n;
  }
  // Event-driven fixpoint: inside the reified comparisons against a constant (`b = (y = k)`, `b = (y <= k)`: the channelling of
  // element constraints) the records of one variable y are made contiguous, so that a slice holds few distinct y and the
  // kernel can evaluate the lanes sharing a variable jointly (kernels.hpp: KEY_EQR_BIC).
  auto key2 = [&](const tb_prop& q) -> long long {
    const long long k1 = key(q);
    if (!event || (q.op != TB_EQ && q.op != TB_LEQ)) return k1 << 32;
    const tb_itv dz = n.store[(size_t)q.z];
    const bool zc = dz.lb == dz.ub && dz.lb != TB_NINF && dz.lb != TB_PINF;
    return (k1 << 32) | (zc ? (long long)q.y + 1 : 0);
  };
  // ... and inside one y by the constant, so that a group's lanes are in value order (dense groups: bit scans instead of walks)
  auto key3 = [&](const tb_prop& q) -> long long {
    if (!event || (q.op != TB_EQ && q.op != TB_LEQ)) return 0;
    const tb_itv dz = n.store[(size_t)q.z];
    return (dz.lb == dz.ub && dz.lb != TB_NINF && dz.lb != TB_PINF) ? (long long)dz.lb : 0;
  };
  std::stable_sort(n.props.begin(), n.props.end(), [&](const tb_prop& a, const tb_prop& c) {
    const long long ka = key2(a), kc = key2(c);
    return ka != kc ? ka < kc : key3(a) < key3(c);
  });
  // (Measured and rejected, r06: sums grouped by the kinds of their operands -- constant / Boolean / integer -- so that a slice's columns are of one kind and take the cheap loads:
  //  the proof search of the sharded_search record gains 2.3 %, the headline step loses 4.1 % and trains15 4.5 % (more evaluations per node: the sums of one constraint no longer
  //  share a slice).  profiles/r06_ab_sort_kinds.txt)
  // (Measured and rejected, r03: re-ordering the records inside a class for fewer reader slices per variable -- the order with the fewest
  //  (variable, slice) incidences among the caller's and the sorts by x, y, z.  wordpress7_500: sorting its 30 017 implications `y <= z` by y
  //  brings 3.51 reader slices per variable down to 2.30, and the search from 4.15e7 to 3.49e7 nodes/s with 55 % more evaluations per node:
  //  the readers of the few z that head hundreds of implications, contiguous in the caller's order, end up in thirty slices, and those are
  //  the variables that move.  An unweighted incidence count is the wrong cost; the caller's order stays.)
  // Event-driven fixpoint: every class starts on a slice boundary (idle records, op < 0, fill the slice the previous class
  // ends in), so that no slice mixes two classes: a mixed slice takes the generic run with every class body it holds --
  // a third of trains15's runs were on its nine class-straddling slices.
  if (pad && !n.props.empty()) {
    std::vector<tb_prop> padded;
    padded.reserve(n.props.size() + 64 * 12);
    int cur = key(n.props[0]) / 16;
    for (const tb_prop& q : n.props) {
      const int c = key(q) / 16;
      if (c != cur) { while (padded.size() % 64 != 0) padded.push_back(tb_prop{-1, 0, 0, 0}); cur = c; }
      padded.push_back(q);
    }
    n.props.swap(padded);
  }
  return n;
}

struct DevBuffers {
  std::vector<void*> ptrs;
  ~DevBuffers() { for (void* p : ptrs) if (p) (void)hipFree(p); }
  template <class Tp>
  int alloc(Tp** out, size_t count) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(16, count * sizeof(Tp)));
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? TB_ERR_OOM : TB_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    ptrs.push_back(p);
    *out = static_cast<Tp*>(p);
    return TB_OK;
  }
};

// device wall clock "now" (the in-kernel watchdogs compare against it)
int device_now(hipStream_t stream, long long* d_now, long long* now_out) {
  clock_kernel<0><<<1, 1, 0, stream>>>(d_now);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(stream));
  HIP_TRY(hipMemcpy(now_out, d_now, sizeof(long long), hipMemcpyDeviceToHost));
  return TB_OK;
}

// Software bounds build (kernels.hpp: TB_BOUNDS): the limits of one launch go to the device before it, a report comes back after it.
#ifdef TB_BOUNDS
int bounds_arm(const DevProblem& P, const LaunchPlan& plan, int strats, int strat_total, const int adj_sizes[2], Ctrl* ctrl) {
  BoundsLimits L{};
  L.store_words = plan.vext * 2; L.slab_vars = P.n_vars; L.n_slices = std::max(1, plan.n_slices); L.records = std::max(1, plan.n_slices) * 64;
  L.adj_vars = std::max(1, adj_sizes[0]); L.adj_rest = std::max(1, adj_sizes[1]);
  L.strats = std::max(1, strats); L.strat_total = strat_total; L.snapshot_levels = std::max(1, plan.snapshot_levels);
  L.mark_words = std::max(1, plan.vext * 2 - plan.unent_off / 4); L.chg_cap = std::max(1, plan.chg_cap); L.ctrl = ctrl;
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_bl), &L, sizeof(L)));
  return TB_OK;
}
int bounds_collect(const char* what) {
  BoundsReport r{};
  HIP_TRY(hipMemcpyFromSymbol(&r, HIP_SYMBOL(g_br), sizeof(r)));
  if (r.hits == 0) return TB_OK;
  const BoundsReport zero{};
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_br), &zero, sizeof(zero)));
  return fail(TB_ERR_HIP, std::string("bounds build: ") + what + ": " + std::to_string(r.hits) + " out-of-range indexes; first: site " + std::to_string(r.site) + ", index " +
                              std::to_string(r.index) + ", limit " + std::to_string(r.limit) + ", workgroup " + std::to_string(r.workgroup) + ", thread " + std::to_string(r.thread));
}
#endif

}  // namespace

// ---- session -------------------------------------------------------------------------------------

struct tb_session {
  tb_config cfg{};
  DeviceCaps caps;
  Layout lay;
  LaunchPlan plan;
  DevProblem P{};
  DevProblem* d_P = nullptr;  // device copy of P, refreshed at every start (the search kernel reads the problem through a pointer)
  DevBuffers bufs;
  Mailbox* mbox_host = nullptr;
  Mailbox* mbox_dev = nullptr;
  // solution ring (streaming): one pinned host block = [consumed | seq[slots] | data[slots][n_vars]]
  unsigned char* ring_host = nullptr;
  int ring_slots = 0;
  unsigned long long ring_next = 0;  // next ticket the host expects
  hipStream_t stream = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  int32_t n_vars = 0, obj_var = -1;
  bool started = false, finished = false, armed = false;
  // multi-GPU: this session's cell (fine-grained device memory) and the cells of the other ranks as this device sees them
  PeerCell* cell = nullptr;
  bool cell_fine_grained = false;
  std::vector<PeerCell*> peer_cells;   // [world]; [rank] = cell; nullptr = not linked
  std::vector<void*> ipc_mapped;       // hipIpcOpenMemHandle mappings to close
  PeerCell** d_peers = nullptr;        // device copy of peer_cells
  unsigned long long local_count = 0;  // size of this rank's share of the index space
  int host_best = TB_PINF;
  long long* d_now = nullptr;
  int adj_sizes[2] = {1 << 30, 1 << 30}, strat_total = 0;  // bounds build: limits of this session's tables
  std::chrono::steady_clock::time_point t_start;
  ~tb_session() {
    if (ev_start) (void)hipEventDestroy(ev_start);
    if (ev_stop) (void)hipEventDestroy(ev_stop);
    if (stream) (void)hipStreamDestroy(stream);
    for (void* m : ipc_mapped) (void)hipIpcCloseMemHandle(m);
    if (cell) (void)hipFree(cell);
    if (mbox_host) (void)hipHostFree(mbox_host);
    if (ring_host) (void)hipHostFree(ring_host);
  }
  unsigned long long* ring_consumed() const { return reinterpret_cast<unsigned long long*>(ring_host); }
  unsigned long long* ring_seq() const { return reinterpret_cast<unsigned long long*>(ring_host + 64); }
  unsigned char* ring_data() const { return ring_host + 64 + align16((size_t)ring_slots * 8); }  // [slots] slabs of plan.vext intervals
};

// Event-driven fixpoint with sorted records: plan for the class-padded record array (to_internal) when the store stays out of
// global memory with it.  Returns the number of records the kernels will see.
int32_t plan_records(const tb_config& cfg, const DeviceCaps& caps, int32_t n_vars, int32_t n_stores, const tb_itv* stores, int32_t n_props,
                     const tb_prop* props, Layout* lay, LaunchPlan* plan, int32_t pinned = -1) {
  if (cfg.fixpoint != 2 || n_props == 0 || plan->mem_kind == TB_MEM_GLOBAL || (cfg.reserved[0] & (0x200000 | 0x20))) return n_props;
  const int32_t n_pad = padded_count(n_props, props, stores);
  // A slice that mixes two classes takes the generic run (7000 cycles at 7 waves per SIMD on trains15, against 1500 for a class-pure
  // one).  r03 padded only where the padding was at least 1/16 of the records (wordpress7_500 measured -1 % with it then: the extra
  // slices added runs); with r04's wake-up filters the runs of the padded network are the cheaper ones everywhere, same box:
  // wordpress7_500 4.95 -> 5.21e7 nodes/s, trains15 3.58 -> 3.67e7, accap_a3 (padded either way) 8.4e7.  Knob 0x20 switches it off.
  if (n_pad == n_props || (n_pad + 63) / 64 >= 0xfffe) return n_props;
  Layout l2;
  LaunchPlan p2;
  if (choose_layout(cfg, caps, n_vars, n_stores, stores, n_pad, &l2, &p2, pinned) != TB_OK || p2.mem_kind == TB_MEM_GLOBAL) return n_props;
  // (the extra slices lengthen the bitmaps: not when that costs a workgroup per CU -- LDS comes in granules, plan_launch)
  if (!(cfg.reserved[0] & 0x10) && p2.blocks_per_cu * p2.threads < plan->blocks_per_cu * plan->threads) return n_props;
  *lay = std::move(l2);
  *plan = p2;
  return n_pad;
}
// Lanes of every slice that hold a propagator (DevProblem::slice_real), sized for the plan.
std::vector<int> real_lanes(const std::vector<tb_prop>& props, int n_slices) {
  std::vector<int> r((size_t)std::max(1, n_slices), 0);
  for (size_t i = 0; i < props.size(); ++i)
    if (props[i].op >= 0 && i / 64 < r.size()) r[i / 64]++;
  return r;
}

// The records as the device reads them: in the COMPACT8 layout every operand that is a narrow integer becomes a reference carrying its base
// (Layout::ref); the host-side analyses (adjacency, chains, successor records) work on the plain ids.
std::vector<int4> device_records(const std::vector<int4>& packed, const Layout& lay) {
  if (!lay.c8) return packed;
  std::vector<int4> out = packed;
  for (int4& r : out) {
    if (r.y >= 0) r.y = lay.ref(r.y);
    if (r.z >= 0) r.z = lay.ref(r.z);
    if (r.w >= 0) r.w = lay.ref(r.w);
  }
  return out;
}

// Tables of the event-driven fixpoint (successor records, slice infos, variable adjacency): built from the packed records and uploaded.
// `root`: the first store of the batch in the internal numbering (finite-domain test of the plain lean runs).
int upload_event_tables(DevBuffers& bufs, DevProblem& P, const tb_config& cfg, const LaunchPlan& plan, const Layout& lay, int32_t n_rec,
                        const std::vector<tb_prop>& net_props, const Adjacency& adj, const std::vector<int4>& packed, const std::vector<int>& value,
                        const tb_itv* root) {
  int rc;
  const std::vector<int> real = real_lanes(net_props, plan.n_slices);
  int* d_real = nullptr;
  if ((rc = bufs.alloc(&d_real, real.size())) != TB_OK) return rc;
  HIP_TRY(hipMemcpy(d_real, real.data(), real.size() * sizeof(int), hipMemcpyHostToDevice));
  P.slice_real = d_real;
#ifdef TB_TUNING
  P.slice_census = nullptr;
  if ((cfg.reserved[0] & 0x400000) && cfg.verbose) {
    unsigned* d_census = nullptr;
    if ((rc = bufs.alloc(&d_census, (size_t)plan.n_slices * 2)) != TB_OK) return rc;
    HIP_TRY(hipMemset(d_census, 0, (size_t)plan.n_slices * 2 * sizeof(unsigned)));
    P.slice_census = d_census;
  }
#endif
  // the lean implication records address Boolean words by a 16-bit word index inside the slab
  const bool lean = lay.compact && std::getenv("TB_NO_LEAN") == nullptr && (size_t)lay.bool_word0() + (size_t)lay.bool_words() <= 0x10000;  // (TB_NO_LEAN: A/B runs)
  // (TB_NO_CHAIN_RANGE / TB_NO_COND_WAKE: A/B runs of the two r04 wake-up filters)
  const bool joint = !(cfg.reserved[0] & 0x4000000);  // channelling slices take the joint run
  const Chains chains = find_chains((int32_t)adj.lists.size(), n_rec, net_props.data(), packed, value, root, lay.compact && joint && std::getenv("TB_NO_CHAIN_RANGE") == nullptr);
  std::vector<int4> succ = pack_succ(n_rec, net_props.data(), adj, packed, value, joint, chains, lay.n_int, lean, lay.bool_word0(), joint && std::getenv("TB_NO_COND_WAKE") == nullptr, lay.c8 ? &lay : nullptr);
  succ.resize((size_t)plan.n_slices * 64, make_int4(-1, -1, -1, 0));
  int4* d_succ = nullptr;
  if ((rc = bufs.alloc(&d_succ, succ.size())) != TB_OK) return rc;
  if (!succ.empty()) HIP_TRY(hipMemcpy(d_succ, succ.data(), succ.size() * sizeof(int4), hipMemcpyHostToDevice));
  P.succ = d_succ;
  std::vector<int2> info = slice_infos(packed, real, plan.n_slices, lean, root, std::getenv("TB_NO_LEAN") == nullptr, chains);
  {  // conditional wake-up of element implications by "c became false" (TB_NO_COND2: A/B runs)
    std::vector<char> has;
    std::vector<int2> cond2(std::max<size_t>(64, (size_t)plan.n_slices * 64), make_int2(-1, 0));
    if (lean && joint && std::getenv("TB_NO_COND2") == nullptr) {
      cond2 = pack_cond2(n_rec, net_props.data(), adj, packed, value, lay, &has);
      cond2.resize(std::max<size_t>(64, (size_t)plan.n_slices * 64), make_int2(-1, 0));
      for (int q = 0; q < plan.n_slices && (size_t)q < has.size(); ++q) if (has[(size_t)q]) info[(size_t)q].y |= 0x2000;
    }
    int2* d_cond2 = nullptr;
    if ((rc = bufs.alloc(&d_cond2, cond2.size())) != TB_OK) return rc;
    HIP_TRY(hipMemcpy(d_cond2, cond2.data(), cond2.size() * sizeof(int2), hipMemcpyHostToDevice));
    P.cond2 = d_cond2;
  }
  if (std::getenv("TB_DUMP_SLICES") != nullptr)  // debugging aid
    for (int q = 0; q < plan.n_slices; ++q) {
      int prefix = 0, classed = 0;
      for (int l = 0; l < 64; ++l) { const bool c = ((unsigned)packed[(size_t)q * 64 + l].x >> 16) != 0u; classed += c ? 1 : 0; if (c && prefix == l) ++prefix; }
      std::fprintf(stderr, "%% slice %d: key %#x w0 %#x real %d lean %d finite %d chain %d | lanes with a class set %d, as a prefix %d\n", q, (unsigned)info[(size_t)q].x >> 16, (unsigned)info[(size_t)q].x,
                   info[(size_t)q].y & 0xff, (info[(size_t)q].y >> 8) & 1, (info[(size_t)q].y >> 9) & 1, (info[(size_t)q].y >> 10) & 1, classed, prefix);
    }
  int2* d_info = nullptr;
  if ((rc = bufs.alloc(&d_info, info.size())) != TB_OK) return rc;
  HIP_TRY(hipMemcpy(d_info, info.data(), info.size() * sizeof(int2), hipMemcpyHostToDevice));
  P.slice_info = d_info;
  std::vector<int4> heads; std::vector<int> rest;
  pack_var_adj(adj, &heads, &rest);
  g_adj_sizes[0] = (int)(heads.size() / 2); g_adj_sizes[1] = (int)rest.size();  // (bounds build: limits of var_adj / adj_rest for the launch that follows)
  int4* d_heads = nullptr; int* d_rest = nullptr;
  if ((rc = bufs.alloc(&d_heads, heads.size())) != TB_OK) return rc;
  if ((rc = bufs.alloc(&d_rest, rest.size())) != TB_OK) return rc;
  HIP_TRY(hipMemcpy(d_heads, heads.data(), heads.size() * sizeof(int4), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_rest, rest.data(), rest.size() * sizeof(int), hipMemcpyHostToDevice));
  P.var_adj = d_heads; P.adj_rest = d_rest;
  return TB_OK;
}

extern "C" {

const char* tb_version(void) { return "turbo-hip 0.1.0 (gfx950)"; }
const char* tb_last_error(void) { return g_last_error.c_str(); }

int tb_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int tb_get_device_info(int device, tb_device_info* out) {
  if (!out) return fail(TB_ERR_INVALID, "null output");
  DeviceCaps caps;
  int rc = query_caps(device, &caps);
  if (rc != TB_OK) return rc;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  std::memset(out, 0, sizeof(*out));
  std::snprintf(out->name, sizeof(out->name), "%s (%s)", prop.name, prop.gcnArchName);
  out->compute_units = caps.cus;
  out->lds_bytes_per_cu = caps.lds_per_cu;
  out->wavefront_size = prop.warpSize;
  out->clock_khz = prop.clockRate;
  out->total_global_mem = (int64_t)prop.totalGlobalMem;
  out->is_gfx950 = std::strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
  out->xcc_count = 0;
  return TB_OK;
}

int tb_eps_local_count(int32_t subproblems_power, int32_t chunk_log2, int32_t rank, int32_t world_size, uint64_t* count_out) {
  if (subproblems_power < 0 || subproblems_power > 62 || chunk_log2 < 0 || world_size < 1 || rank < 0 || rank >= world_size || !count_out)
    return fail(TB_ERR_INVALID, "bad partition arguments");
  *count_out = eps_local_count(subproblems_power, std::min(chunk_log2, subproblems_power), rank, world_size);
  return TB_OK;
}

int tb_eps_global_index(int32_t subproblems_power, int32_t chunk_log2, int32_t rank, int32_t world_size, uint64_t j, uint64_t* index_out) {
  uint64_t n = 0;
  int rc = tb_eps_local_count(subproblems_power, chunk_log2, rank, world_size, &n);
  if (rc != TB_OK) return rc;
  if (!index_out || j >= n) return fail(TB_ERR_INVALID, "local subproblem index out of range");
  *index_out = eps_global_index(j, std::min(chunk_log2, subproblems_power), rank, world_size);
  return TB_OK;
}

int tb_propagate(const tb_config* cfg_in, int32_t n_vars, int32_t n_props, const tb_prop* props,
                 int32_t n_stores, tb_itv* stores_inout, int32_t* failed_out, int32_t* all_entailed_out,
                 uint64_t* iterations_out, uint64_t* deductions_out, int64_t* kernel_ns_out) {
  if (!cfg_in) return fail(TB_ERR_INVALID, "null config");
  if (n_stores < 0 || (n_stores > 0 && !stores_inout)) return fail(TB_ERR_INVALID, "null stores");
  tb_config cfg = *cfg_in;
  resolve_fixpoint(&cfg, n_props);
  int rc = validate_network(n_vars, stores_inout, n_props, props);
  if (rc != TB_OK) return rc;
  if (n_stores == 0) return TB_OK;
  DeviceCaps caps;
  if ((rc = query_caps(cfg.device, &caps)) != TB_OK) return rc;
  Layout lay;
  LaunchPlan plan;
  if ((rc = choose_layout(cfg, caps, n_vars, n_stores, stores_inout, n_props, &lay, &plan, -1, false)) != TB_OK) return rc;
  const int32_t n_rec = plan_records(cfg, caps, n_vars, n_stores, stores_inout, n_props, props, &lay, &plan);  // records the kernels see
  const size_t VX = (size_t)plan.vext, slab_bytes = VX * 8;

  DevBuffers bufs;
  DevProblem P{};
  int4* d_props = nullptr; int2* d_stores = nullptr; PropagateOut* d_out = nullptr;
  if ((rc = bufs.alloc(&d_props, (size_t)plan.n_slices * 64)) != TB_OK) return rc;
  if ((rc = bufs.alloc(&d_stores, (size_t)n_stores * VX)) != TB_OK) return rc;
  if ((rc = bufs.alloc(&d_out, (size_t)n_stores)) != TB_OK) return rc;
  {
    std::vector<char> c0, is_const((size_t)n_vars);
    std::vector<int> v0, value((size_t)n_vars);
    find_constants(n_vars, n_stores, stores_inout, &c0, &v0);
    for (int v = 0; v < n_vars; ++v) { is_const[(size_t)lay.perm[(size_t)v]] = c0[(size_t)v]; value[(size_t)lay.perm[(size_t)v]] = v0[(size_t)v]; }
    const InternalNet net = to_internal(lay, stores_inout, n_props, props, (cfg.reserved[0] & 0x200000) != 0 || plan.mem_kind == TB_MEM_GLOBAL, cfg.fixpoint == 2 && !(cfg.reserved[0] & 0x8000000), n_rec != n_props);
    if ((int32_t)net.props.size() != n_rec) return fail(TB_ERR_INVALID, "internal: record padding does not match its plan");
    const Adjacency adj = build_adjacency(n_vars, n_rec, net.props.data(), is_const, value, !(cfg.reserved[0] & 0x80));
    std::vector<int4> packed = pack_props(n_rec, net.props.data(), is_const, value, adj, lay.n_int, lay.n_slab());
    packed.resize((size_t)plan.n_slices * 64, make_int4(K_LEQ_T, 0, 0, 0));
    if (n_rec) { const std::vector<int4> dev = device_records(packed, lay); HIP_TRY(hipMemcpy(d_props, dev.data(), dev.size() * sizeof(int4), hipMemcpyHostToDevice)); }
    // hull of the batch, internal numbering: what "finite domains" means for a batch of stores
    std::vector<tb_itv> hull((size_t)std::max(1, n_vars));
    for (int v = 0; v < n_vars; ++v) {
      tb_itv h = stores_inout[v];
      for (int32_t k = 1; k < n_stores; ++k) { const tb_itv d = stores_inout[(size_t)k * (size_t)n_vars + (size_t)v]; h.lb = std::min(h.lb, d.lb); h.ub = std::max(h.ub, d.ub); }
      hull[(size_t)lay.perm[(size_t)v]] = h;
    }
    if ((rc = upload_event_tables(bufs, P, cfg, plan, lay, n_rec, net.props, adj, packed, value, hull.data())) != TB_OK) return rc;
  }
  P.n_slices = plan.n_slices; P.dirty_words = plan.dirty_words; P.vext = plan.vext; P.chg_cap = plan.chg_cap;
  P.n_int = plan.n_int; P.unent_off = plan.unent_off;
  std::vector<unsigned char> slabs((size_t)n_stores * slab_bytes, 0);
  for (int32_t k = 0; k < n_stores; ++k) encode_slab(lay, stores_inout + (size_t)k * (size_t)n_vars, slabs.data() + (size_t)k * slab_bytes);
  HIP_TRY(hipMemcpy(d_stores, slabs.data(), slabs.size(), hipMemcpyHostToDevice));
  P.n_vars = lay.n_slab(); P.n_props = n_rec; P.props = d_props;  // (the variables of the slab)
  P.fixpoint = cfg.fixpoint; P.wac1_threshold = (int)std::min<uint64_t>(cfg.wac1_threshold, 0x7fffffffu);
  P.mem_kind = plan.mem_kind; P.debug = cfg.reserved[0];
  const bool event = cfg.fixpoint == 2;
  const int compact = plan.prop_opt();  // the batch kernel's fourth template flag
  {
    int occ = 0;
    if ((rc = prepare_kernel(false, plan.mem_kind, plan.tmax, event, compact, plan.shared_bytes, plan.threads, &occ)) != TB_OK) return rc;
    if (occ > 0) plan.num_blocks = std::min(plan.num_blocks, occ * caps.cus);
  }
  hipStream_t stream;
  HIP_TRY(hipStreamCreate(&stream));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  P.deadline_ticks = 0;
  if (cfg.timeout_ms != 0) {  // same watchdog as the search kernel: a slowly converging network must not outlive -t
    long long* d_now = nullptr;
    long long now = 0;
    if ((rc = bufs.alloc(&d_now, 1)) != TB_OK) return rc;
    if ((rc = device_now(stream, d_now, &now)) != TB_OK) return rc;
    P.deadline_ticks = now + (long long)cfg.timeout_ms * (long long)caps.wall_khz;
  }
  const int grid = std::min(n_stores, plan.num_blocks);
#ifdef TB_BOUNDS
  { const int none[2] = {1 << 30, 1 << 30}; if ((rc = bounds_arm(P, plan, 1, 0, event ? g_adj_sizes : none, nullptr)) != TB_OK) return rc; }
#endif
  HIP_TRY(hipEventRecord(e0, stream));
  if ((rc = launch_prop(KernelSel{plan.mem_kind, plan.tmax, event, compact}, KernelGrid{grid, plan.threads, plan.shared_bytes, stream}, P, d_stores, d_out, n_stores)) != TB_OK) return rc;
  HIP_TRY(hipEventRecord(e1, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  if (kernel_ns_out) *kernel_ns_out = (int64_t)((double)ms * 1e6);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(stream);
#ifdef TB_BOUNDS
  if ((rc = bounds_collect("tb_propagate")) != TB_OK) return rc;
#endif
  std::vector<PropagateOut> outs((size_t)n_stores);
  HIP_TRY(hipMemcpy(outs.data(), d_out, (size_t)n_stores * sizeof(PropagateOut), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(slabs.data(), d_stores, slabs.size(), hipMemcpyDeviceToHost));
  for (int32_t k = 0; k < n_stores; ++k) decode_slab(lay, slabs.data() + (size_t)k * slab_bytes, stores_inout + (size_t)k * (size_t)n_vars);
  for (int32_t s = 0; s < n_stores; ++s) {
    if (failed_out) failed_out[s] = outs[(size_t)s].failed;
    if (all_entailed_out) all_entailed_out[s] = outs[(size_t)s].all_entailed;
    if (iterations_out) iterations_out[s] = outs[(size_t)s].iterations;
    if (deductions_out) deductions_out[s] = outs[(size_t)s].deductions;
  }
  return TB_OK;
}

int tb_session_create(const tb_config* cfg_in, int32_t n_vars, const tb_itv* root_store,
                      int32_t n_props, const tb_prop* props,
                      int32_t n_strats, const int32_t* strat_var_order, const int32_t* strat_val_order,
                      const int32_t* strat_off, const int32_t* strat_vars,
                      int32_t obj_var, tb_session** out) {
  if (!cfg_in || !out) return fail(TB_ERR_INVALID, "null argument");
  *out = nullptr;
  int rc = validate_network(n_vars, root_store, n_props, props);
  if (rc != TB_OK) return rc;
  if (n_strats < 0 || (n_strats > 0 && (!strat_var_order || !strat_val_order || !strat_off))) return fail(TB_ERR_INVALID, "null strategies");
  if (obj_var < -1 || obj_var >= n_vars) return fail(TB_ERR_INVALID, "objective variable out of range");
  int32_t total_svars = n_strats > 0 ? strat_off[n_strats] : 0;
  for (int32_t s = 0; s < n_strats; ++s) {
    if (strat_off[s] > strat_off[s + 1] || strat_off[s] < 0) return fail(TB_ERR_INVALID, "strategy offsets must be non-decreasing");
    if (strat_var_order[s] < TB_INPUT_ORDER || strat_var_order[s] > TB_LARGEST) return fail(TB_ERR_INVALID, "unknown variable order");
    if (strat_val_order[s] < TB_VAL_MIN || strat_val_order[s] > TB_VAL_REVERSE_SPLIT) return fail(TB_ERR_INVALID, "unknown value order");
  }
  if (total_svars > 0 && !strat_vars) return fail(TB_ERR_INVALID, "null strategy variables");
  for (int32_t i = 0; i < total_svars; ++i)
    if (strat_vars[i] < 0 || strat_vars[i] >= n_vars) return fail(TB_ERR_INVALID, "strategy variable out of range");
  if (cfg_in->world_size > 1 && (cfg_in->rank < 0 || cfg_in->rank >= cfg_in->world_size)) return fail(TB_ERR_INVALID, "rank out of range");

  std::unique_ptr<tb_session> s(new tb_session);
  s->cfg = *cfg_in;
  resolve_fixpoint(&s->cfg, n_props);
  s->n_vars = n_vars; s->obj_var = obj_var;
  if ((rc = query_caps(s->cfg.device, &s->caps)) != TB_OK) return rc;
  if ((rc = choose_layout(s->cfg, s->caps, n_vars, 1, root_store, n_props, &s->lay, &s->plan, obj_var)) != TB_OK) return rc;
  int32_t n_rec = plan_records(s->cfg, s->caps, n_vars, 1, root_store, n_props, props, &s->lay, &s->plan, obj_var);  // records the kernels see
  {
    // cap the grid by what is actually resident (registers, LDS): queued workgroups of a persistent kernel only
    // add tail latency; re-plan so that the subproblem count follows the real workgroup count
    int occ = 0;
    if ((rc = prepare_kernel(true, s->plan.mem_kind, s->plan.tmax, s->plan.kernel_event != 0, s->plan.kernel_opt, s->plan.shared_bytes, s->plan.threads, &occ)) != TB_OK) return rc;
    // (the occupancy query does not round the LDS request up to its allocation granule: 13 workgroups of 12 336 B "fit" a CU that holds 12)
    if (s->plan.shared_bytes > 0) occ = std::min<int>(occ > 0 ? occ : 1 << 20, (int)((size_t)s->caps.lds_per_cu / lds_footprint(s->caps, (size_t)s->plan.shared_bytes)));
    if (occ > 0 && (long long)occ * s->caps.cus < (long long)s->plan.num_blocks) {
      tb_config capped = s->cfg;
      capped.or_nodes = (uint64_t)occ * (uint64_t)s->caps.cus;
      if ((rc = choose_layout(capped, s->caps, n_vars, 1, root_store, n_rec, &s->lay, &s->plan, obj_var)) != TB_OK) return rc;
      if (n_rec != n_props && s->plan.mem_kind == TB_MEM_GLOBAL) {  // (the padded array no longer fits next to the capped grid: plan the plain one)
        n_rec = n_props;
        if ((rc = choose_layout(capped, s->caps, n_vars, 1, root_store, n_rec, &s->lay, &s->plan, obj_var)) != TB_OK) return rc;
      }
      if ((rc = prepare_kernel(true, s->plan.mem_kind, s->plan.tmax, s->plan.kernel_event != 0, s->plan.kernel_opt, s->plan.shared_bytes, s->plan.threads, &occ)) != TB_OK) return rc;
    }
  }
  if (s->plan.hot) renumber_by_reads(&s->lay, n_props, props);
  else if (s->plan.mem_kind == TB_MEM_GLOBAL && !s->lay.compact && n_rec == n_props && std::getenv("TB_GLOBAL_RENUMBER") != nullptr) renumber_for_locality(&s->lay, n_props, props);
  const LaunchPlan& plan = s->plan;
  const Layout& lay = s->lay;
  DevProblem& P = s->P;
  const size_t B = (size_t)plan.num_blocks;
  const size_t VX = (size_t)plan.vext;

  // strategies in the internal numbering; with a renumbered store a whole-store strategy (empty list) becomes the
  // explicit list of all variables in the caller's order, so that ties still resolve to the caller's lowest index
  std::vector<int32_t> i_off((size_t)n_strats + 1, 0), i_vars;
  for (int32_t k = 0; k < n_strats; ++k) {
    i_off[(size_t)k] = (int32_t)i_vars.size();
    // (a constant kept out of the slab keeps its position in the list -- positions break ties -- as -1: assigned, never a candidate)
    auto entry = [&](int32_t v) { const int i = lay.perm[(size_t)v]; return i >= lay.n_slab() ? -1 : lay.ref(i); };  // (COMPACT8: a reference carries the base)
    if (strat_off[k] == strat_off[k + 1] && (lay.compact || lay.renumbered)) for (int32_t v = 0; v < n_vars; ++v) i_vars.push_back(entry(v));
    else for (int32_t j = strat_off[k]; j < strat_off[k + 1]; ++j) i_vars.push_back(entry(strat_vars[j]));
  }
  i_off[(size_t)n_strats] = (int32_t)i_vars.size();
  total_svars = (int32_t)i_vars.size();
  const int32_t i_obj = obj_var >= 0 ? lay.ref(lay.perm[(size_t)obj_var]) : -1;

  s->strat_total = total_svars;
  int4* d_props = nullptr; int2* d_root = nullptr; int *d_vo = nullptr, *d_vl = nullptr, *d_off = nullptr, *d_sv = nullptr;
  if ((rc = s->bufs.alloc(&d_props, (size_t)plan.n_slices * 64)) != TB_OK) return rc;
  if ((rc = s->bufs.alloc(&d_root, VX)) != TB_OK) return rc;
  if ((rc = s->bufs.alloc(&d_vo, (size_t)n_strats)) != TB_OK) return rc;
  if ((rc = s->bufs.alloc(&d_vl, (size_t)n_strats)) != TB_OK) return rc;
  if ((rc = s->bufs.alloc(&d_off, (size_t)n_strats + 1)) != TB_OK) return rc;
  if ((rc = s->bufs.alloc(&d_sv, (size_t)total_svars)) != TB_OK) return rc;
  {
    const InternalNet net = to_internal(lay, root_store, n_props, props, (s->cfg.reserved[0] & 0x200000) != 0 || plan.mem_kind == TB_MEM_GLOBAL, s->cfg.fixpoint == 2 && !(s->cfg.reserved[0] & 0x8000000), n_rec != n_props,
                                        plan.team && s->cfg.fixpoint == 0 ? TEAM_AC1_SORT_WINDOW : 0);
    if ((int32_t)net.props.size() != n_rec) return fail(TB_ERR_INVALID, "internal: record padding does not match its plan");
    std::vector<char> is_const;
    std::vector<int> value;
    find_constants(n_vars, 1, net.store.data(), &is_const, &value);  // constants = singleton variables of the root store
    const Adjacency adj = build_adjacency(n_vars, n_rec, net.props.data(), is_const, value, !(s->cfg.reserved[0] & 0x80));
    std::vector<int4> packed = pack_props(n_rec, net.props.data(), is_const, value, adj, lay.n_int, lay.n_slab());
    packed.resize((size_t)plan.n_slices * 64, make_int4(K_LEQ_T, 0, 0, 0));
    if (n_rec) { const std::vector<int4> dev = device_records(packed, lay); HIP_TRY(hipMemcpy(d_props, dev.data(), dev.size() * sizeof(int4), hipMemcpyHostToDevice)); }
    if ((rc = upload_event_tables(s->bufs, s->P, s->cfg, s->plan, s->lay, n_rec, net.props, adj, packed, value, net.store.data())) != TB_OK) return rc;
    s->adj_sizes[0] = g_adj_sizes[0]; s->adj_sizes[1] = g_adj_sizes[1];
  }
  s->P.n_slices = s->plan.n_slices; s->P.dirty_words = s->plan.dirty_words; s->P.vext = s->plan.vext; s->P.chg_cap = s->plan.chg_cap;
  s->P.n_int = s->plan.n_int; s->P.unent_off = s->plan.unent_off;
  {
    std::vector<unsigned char> slab(std::max<size_t>(16, VX * 8), 0);
    if (n_vars) encode_slab(lay, root_store, slab.data());
    HIP_TRY(hipMemcpy(d_root, slab.data(), VX * 8, hipMemcpyHostToDevice));
  }
  if (n_strats) {
    HIP_TRY(hipMemcpy(d_vo, strat_var_order, (size_t)n_strats * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_vl, strat_val_order, (size_t)n_strats * 4, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMemcpy(d_off, i_off.data(), ((size_t)n_strats + 1) * 4, hipMemcpyHostToDevice));
  if (total_svars) HIP_TRY(hipMemcpy(d_sv, i_vars.data(), (size_t)total_svars * 4, hipMemcpyHostToDevice));

  if (plan.mem_kind == TB_MEM_GLOBAL) { if ((rc = s->bufs.alloc(&P.g_store, B * VX)) != TB_OK) return rc; }
  if ((rc = s->bufs.alloc(&P.g_snap, B * (size_t)plan.snapshot_levels * VX)) != TB_OK) return rc;
  if ((rc = s->bufs.alloc(&P.g_best, B * VX)) != TB_OK) return rc;
  if ((rc = s->bufs.alloc(&P.g_dec, B * (size_t)plan.max_depth)) != TB_OK) return rc;
  {
    // Pool of further decision-stack segments, handed out on demand (the reference grows a block's stack when it is full,
    // barebones:401-403): a quarter as many segments as workgroups, within 1/16 of the memory that is free now.
    size_t segs = std::max<size_t>(MAX_DEC_SEGS + 1, B / 4);
    size_t free_now = 0, total = 0;
    if (hipMemGetInfo(&free_now, &total) != hipSuccess) { (void)hipGetLastError(); free_now = s->caps.free_mem; }
    const size_t seg_bytes = (size_t)plan.max_depth * sizeof(Decision);
    segs = std::min(segs, (free_now / 16) / std::max<size_t>(1, seg_bytes));
    P.dec_pool = nullptr; P.dec_pool_segments = 0;
    if (segs > 0 && s->bufs.alloc(&P.dec_pool, segs * (size_t)plan.max_depth) == TB_OK) P.dec_pool_segments = (int)std::min<size_t>(segs, 0x7fffffff);
  }
  if ((rc = s->bufs.alloc(&P.g_stats, B)) != TB_OK) return rc;
  if ((rc = s->bufs.alloc(&P.ctrl, 1)) != TB_OK) return rc;
  if ((rc = s->bufs.alloc(&s->d_now, 1)) != TB_OK) return rc;

  P.n_vars = lay.n_slab(); P.n_props = n_rec; P.n_strats = n_strats; P.obj_var = i_obj;  // (the variables of the slab)
  P.props = d_props; P.root_store = d_root;
  P.strat_var_order = d_vo; P.strat_val_order = d_vl; P.strat_off = d_off; P.strat_vars = d_sv;
  P.fixpoint = s->cfg.fixpoint;
  P.wac1_threshold = (int)std::min<uint64_t>(s->cfg.wac1_threshold, 0x7fffffffu);
  P.subproblems_power = plan.subproblems_power;
  P.has_eps_strategy = s->cfg.has_eps_strategy;
  P.use_fixed_bound = s->cfg.use_fixed_bound; P.fixed_bound = s->cfg.fixed_bound;
  P.mem_kind = plan.mem_kind; P.snapshot_levels = plan.snapshot_levels; P.max_depth = plan.max_depth; P.debug = s->cfg.reserved[0];
  P.max_depth_log2 = 0;
  while ((1 << P.max_depth_log2) < plan.max_depth) ++P.max_depth_log2;
  // this rank's block-cyclic share of the 2^d subproblems (device_types.hpp: eps_global_index)
  P.world = std::max(1, s->cfg.world_size); P.rank = P.world > 1 ? s->cfg.rank : 0;
  P.chunk_log2 = std::max(0, std::min(s->cfg.eps_chunk_log2, plan.subproblems_power));
  s->local_count = eps_local_count(plan.subproblems_power, P.chunk_log2, P.rank, P.world);
  P.poll_ticks = (int)std::min<long long>(0x3fffffff, (long long)(s->cfg.poll_period_us > 0 ? s->cfg.poll_period_us : 100) * (long long)s->caps.wall_khz / 1000);
  if (P.poll_ticks < 1) P.poll_ticks = 1;
  P.steal = (s->cfg.reserved[0] & 0x1000000) ? 0 : 1;
  P.leaf_assign = s->cfg.leaf_requires_assignment ? 1 : 0;
  P.teams = nullptr;
  P.team_all = (std::getenv("TB_TEAM_ALL") != nullptr && std::getenv("TB_TEAM_ALL")[0] == '1') ? 1 : 0;
  { const char* e = std::getenv("TB_TEAM_SPLIT"); const int k = e ? std::atoi(e) : 4; P.team_split = (k == 1 || k == 2 || k == 4 || k == 8) ? k : 4; }  // four teams per XCD by default
  P.team_relaxed = (std::getenv("TB_TEAM_RELAXED") != nullptr && std::getenv("TB_TEAM_RELAXED")[0] == '0') ? 0 : 1;  // (TB_TEAM_RELAXED=0: acq_rel fences around the barrier, -3 .. -4 %)
  { const char* e = std::getenv("TB_TEAM_JOIN_MS"); const long long ms = e ? std::max(1, std::atoi(e)) : 10000; P.team_join_ticks = (int)std::min<long long>(0x7fffffff, ms * (long long)s->caps.wall_khz); }
  if (plan.team && (rc = s->bufs.alloc(&P.teams, 1)) != TB_OK) return rc;
  P.g_dirty = nullptr;
  if (plan.team && plan.kernel_event && (rc = s->bufs.alloc(&P.g_dirty, B * 2 * (size_t)plan.dirty_words)) != TB_OK) return rc;
  P.blk_counts = nullptr;
  if (std::getenv("TB_BLOCK_COUNTS") != nullptr) {  // (instrumented build, scripts/instr_blocks.py: per-XCD arrays of basic-block execution counts, zeroed here, written out by tb_session_finish)
    if ((rc = s->bufs.alloc(&P.blk_counts, 8 * BLK_COUNT_STRIDE / sizeof(unsigned))) != TB_OK) return rc;
    HIP_TRY(hipMemset(P.blk_counts, 0, 8 * BLK_COUNT_STRIDE));
  }
  // the cell other GPUs reach over xGMI: fine-grained device memory (coherent at system scope while kernels run)
  {
    void* c = nullptr;
    if (hipExtMallocWithFlags(&c, sizeof(PeerCell), hipDeviceMallocFinegrained) == hipSuccess) s->cell_fine_grained = true;
    else {
      (void)hipGetLastError();
      if (P.world > 1 && s->cfg.verbose) std::fprintf(stderr, "%% fine-grained device memory is not available: GPUs exchange through the host only\n");
      HIP_TRY(hipMalloc(&c, sizeof(PeerCell)));
    }
    s->cell = static_cast<PeerCell*>(c);
    HIP_TRY(hipMemset(s->cell, 0, sizeof(PeerCell)));
    if (std::getenv("TB_PRINT_PTRS") != nullptr)  // (debugging aid: which buffer a faulting address belongs to)
      std::fprintf(stderr, "%% ptrs B=%zu VX=%zu L=%d depth=%d: props %p root %p g_store %p g_snap %p [%zu B] g_best %p [%zu B] g_dec %p [%zu B] dec_pool %p [%d segs] g_stats %p ctrl %p blk_counts %p cell %p succ %p var_adj %p adj_rest %p slice_info %p cond2 %p\n",
                   B, VX, plan.snapshot_levels, plan.max_depth, (const void*)P.props, (const void*)P.root_store, (void*)P.g_store, (void*)P.g_snap, B * (size_t)plan.snapshot_levels * VX * 8, (void*)P.g_best, B * VX * 8,
                   (void*)P.g_dec, B * (size_t)plan.max_depth * sizeof(Decision), (void*)P.dec_pool, P.dec_pool_segments, (void*)P.g_stats, (void*)P.ctrl, (void*)P.blk_counts, (void*)s->cell,
                   (const void*)P.succ, (const void*)P.var_adj, (const void*)P.adj_rest, (const void*)P.slice_info, (const void*)P.cond2);
    s->peer_cells.assign((size_t)P.world, nullptr);
    s->peer_cells[(size_t)P.rank] = s->cell;
    if ((rc = s->bufs.alloc(&s->d_peers, (size_t)P.world)) != TB_OK) return rc;
    P.cell = s->cell; P.peers = nullptr;  // set at arm time, when some peer is linked
  }
  if (s->cfg.reserved[0] & 0x800000) {  // test aids: tb_session_debug_last_store, tb_session_debug_path
    if ((rc = s->bufs.alloc(&P.g_last, B * VX)) != TB_OK) return rc;
    if ((rc = s->bufs.alloc(&P.g_path_ub, B * (size_t)plan.max_depth)) != TB_OK) return rc;
    if ((rc = s->bufs.alloc(&P.g_path_hdr, B)) != TB_OK) return rc;
    HIP_TRY(hipMemset(P.g_path_hdr, 0, B * sizeof(PathHeader)));
  }
  P.cut_nodes = s->cfg.stop_after_n_nodes;
  P.cut_nodes_total = s->cfg.stop_after_n_nodes_total;
  P.stop_after_n_solutions = s->cfg.stop_after_n_solutions;

  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->mbox_host), sizeof(Mailbox), hipHostMallocMapped));
  std::memset(s->mbox_host, 0, sizeof(Mailbox));
  s->mbox_host->foreign_bound = TB_PINF; s->mbox_host->local_best = TB_PINF;
  HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&s->mbox_dev), s->mbox_host, 0));
  if (s->cfg.stream_solutions && !s->cfg.use_fixed_bound) {
    s->ring_slots = 8;
    const size_t bytes = 64 + align16((size_t)s->ring_slots * 8) + (size_t)s->ring_slots * std::max<size_t>(2, VX) * sizeof(tb_itv);
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->ring_host), bytes, hipHostMallocMapped));
    std::memset(s->ring_host, 0, bytes);
    unsigned char* dev = nullptr;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), s->ring_host, 0));
    P.ring.consumed = reinterpret_cast<unsigned long long*>(dev);
    P.ring.seq = reinterpret_cast<unsigned long long*>(dev + 64);
    P.ring.data = reinterpret_cast<int2*>(dev + 64 + align16((size_t)s->ring_slots * 8));
    P.ring.slots = s->ring_slots;
  }
  HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&s->ev_start));
  HIP_TRY(hipEventCreate(&s->ev_stop));
  // Propagate the root once, here: every subproblem starts from the root (barebones:665-672), and its fixpoint -- the same
  // for all of them -- would otherwise be re-derived from the caller's store at the first node of each of the 2^d dives.
  // The node is still visited and counted; it just finds nothing left to do.  (tb_config.reserved[0] & 0x2000000 keeps the
  // caller's store, for A/B runs.)  An inconsistent root is left as it is: every subproblem then fails on its first node.
  if (n_props > 0 && !(s->cfg.reserved[0] & 0x2000000)) {
    const bool event = s->plan.kernel_event != 0; const int opt = s->plan.prop_opt();
    int occ = 0;
    if ((rc = prepare_kernel(false, plan.mem_kind, plan.tmax, event, opt, plan.shared_bytes, plan.threads, &occ)) != TB_OK) return rc;
    PropagateOut* d_out = nullptr;
    if ((rc = s->bufs.alloc(&d_out, 1)) != TB_OK) return rc;
    std::vector<unsigned char> original(std::max<size_t>(16, VX * 8));
    HIP_TRY(hipMemcpy(original.data(), d_root, VX * 8, hipMemcpyDeviceToHost));
    DevProblem Q = P;
    Q.deadline_ticks = 0;
    if (s->cfg.timeout_ms != 0) {
      long long now = 0;
      if ((rc = device_now(s->stream, s->d_now, &now)) != TB_OK) return rc;
      Q.deadline_ticks = now + (long long)s->cfg.timeout_ms * (long long)s->caps.wall_khz;
    }
#ifdef TB_BOUNDS
    if ((rc = bounds_arm(Q, plan, std::max(1, n_strats), s->strat_total, s->adj_sizes, nullptr)) != TB_OK) return rc;
#endif
    if ((rc = launch_prop(KernelSel{plan.mem_kind, plan.tmax, event, opt}, KernelGrid{1, plan.threads, plan.shared_bytes, s->stream}, Q, d_root, d_out, 1)) != TB_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
#ifdef TB_BOUNDS
    if ((rc = bounds_collect("root fixpoint of tb_session_create")) != TB_OK) return rc;
#endif
    PropagateOut o;
    HIP_TRY(hipMemcpy(&o, d_out, sizeof(o), hipMemcpyDeviceToHost));
    if (o.failed == 0) P.root_fixpoint = 1;
    else HIP_TRY(hipMemcpy(d_root, original.data(), VX * 8, hipMemcpyHostToDevice));
  }
  *out = s.release();
  return TB_OK;
}

int tb_session_plan(tb_session* s, tb_plan* plan_out) {
  if (!s || !plan_out) return fail(TB_ERR_INVALID, "null argument");
  plan_out->num_blocks = s->plan.num_blocks; plan_out->threads_per_block = s->plan.threads;
  plan_out->mem_kind = s->plan.mem_kind; plan_out->shared_bytes = s->plan.shared_bytes;
  plan_out->subproblems_power = s->plan.subproblems_power; plan_out->eps_chunk_log2 = s->P.chunk_log2;
  plan_out->snapshot_levels = s->plan.snapshot_levels; plan_out->decision_stack_depth = s->plan.max_depth;
  plan_out->eps_local_subproblems = s->local_count;
  plan_out->kernel_event = s->plan.kernel_event; plan_out->kernel_opt = s->plan.kernel_opt;
  return TB_OK;
}

int tb_session_export_peer(tb_session* s, tb_peer_handle* handle_out) {
  if (!s || !handle_out) return fail(TB_ERR_INVALID, "null argument");
  static_assert(sizeof(hipIpcMemHandle_t) <= sizeof(tb_peer_handle), "tb_peer_handle is too small for hipIpcMemHandle_t");
  // The kernels CAS / fetch_add the queue word and atomicMin the bound of this cell from other GPUs while kernels run here:
  // that is only coherent in fine-grained memory.  Without it the cell is not handed out -- the callers fall back to the host
  // relay with static shares (distributed.link_group, turbo -gpus).
  if (!s->cell_fine_grained) return fail(TB_ERR_HIP, "this session's cell is not in fine-grained device memory (hipExtMallocWithFlags failed): it cannot be shared with other GPUs");
  HIP_TRY(hipSetDevice(s->cfg.device));
  hipIpcMemHandle_t h;
  HIP_TRY(hipIpcGetMemHandle(&h, s->cell));
  std::memset(handle_out, 0, sizeof(*handle_out));
  std::memcpy(handle_out->bytes, &h, sizeof(h));
  return TB_OK;
}

int tb_session_import_peer(tb_session* s, int32_t peer_rank, const tb_peer_handle* handle) {
  if (!s || !handle) return fail(TB_ERR_INVALID, "null argument");
  if (s->started && !s->finished) return fail(TB_ERR_STATE, "session is running");
  if (peer_rank < 0 || peer_rank >= s->P.world || peer_rank == s->P.rank) return fail(TB_ERR_INVALID, "peer rank out of range");
  // test knob: what an 8-GPU box does when the IPC import misbehaves on one rank -- this process refuses to map its peers' cells, and the group must fall back as a whole
  // (distributed.link_group: all or nothing) to the host relay with static shares (tests/test_gpu_multi.py)
  if (std::getenv("TB_FAIL_IMPORT") != nullptr) return fail(TB_ERR_HIP, "hipIpcOpenMemHandle: refused (TB_FAIL_IMPORT is set: a test of the fallback)");
  HIP_TRY(hipSetDevice(s->cfg.device));
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle->bytes, sizeof(h));
  void* p = nullptr;
  HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  s->ipc_mapped.push_back(p);
  s->peer_cells[(size_t)peer_rank] = static_cast<PeerCell*>(p);
  s->armed = false;
  return TB_OK;
}

int tb_session_link_peer(tb_session* s, tb_session* peer) {
  if (!s || !peer || s == peer) return fail(TB_ERR_INVALID, "null or identical sessions");
  if (s->started && !s->finished) return fail(TB_ERR_STATE, "session is running");
  if (peer->P.world != s->P.world || peer->P.rank == s->P.rank) return fail(TB_ERR_INVALID, "the sessions do not belong to the same group");
  if (peer->plan.subproblems_power != s->plan.subproblems_power || peer->P.chunk_log2 != s->P.chunk_log2)
    return fail(TB_ERR_INVALID, "the sessions were planned with different subproblem counts: pass the same subproblems_power to every rank");
  if (!s->cell_fine_grained || !peer->cell_fine_grained)
    return fail(TB_ERR_HIP, "a session's cell is not in fine-grained device memory (hipExtMallocWithFlags failed): cross-GPU atomics on it would not be coherent");
  HIP_TRY(hipSetDevice(s->cfg.device));
  if (peer->cfg.device != s->cfg.device) {
    int can = 0;
    HIP_TRY(hipDeviceCanAccessPeer(&can, s->cfg.device, peer->cfg.device));
    if (!can) return fail(TB_ERR_HIP, "device " + std::to_string(s->cfg.device) + " cannot access device " + std::to_string(peer->cfg.device) + " (no xGMI / PCIe peer path)");
    const hipError_t e = hipDeviceEnablePeerAccess(peer->cfg.device, 0);
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(TB_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
    (void)hipGetLastError();
  }
  s->peer_cells[(size_t)peer->P.rank] = peer->cell;
  s->armed = false;
  return TB_OK;
}

// Forget every peer cell linked or imported so far: a group that could only be linked in part falls back, as a whole, to the host relay with
// per-rank shares of the node budget (a partly linked rank would otherwise count the group's budget in a cell nobody else adds to).
int tb_session_unlink_peers(tb_session* s) {
  if (!s) return fail(TB_ERR_INVALID, "null session");
  if (s->started && !s->finished) return fail(TB_ERR_STATE, "session is running");
  HIP_TRY(hipSetDevice(s->cfg.device));
  for (void* m : s->ipc_mapped) (void)hipIpcCloseMemHandle(m);
  s->ipc_mapped.clear();
  for (int r = 0; r < s->P.world; ++r) if (r != s->P.rank) s->peer_cells[(size_t)r] = nullptr;
  s->armed = false;
  return TB_OK;
}

// Device-side state of one search: queue = this rank's whole share, no incumbent, counters at zero.
int tb_session_arm(tb_session* s) {
  if (!s) return fail(TB_ERR_INVALID, "null session");
  if (s->started && !s->finished) return fail(TB_ERR_STATE, "session is running");
  HIP_TRY(hipSetDevice(s->cfg.device));
  s->host_best = TB_PINF;
  std::memset(s->mbox_host, 0, sizeof(Mailbox));
  s->mbox_host->foreign_bound = TB_PINF; s->mbox_host->local_best = TB_PINF;
  if (s->ring_host) {
    std::memset(s->ring_host, 0, 64 + align16((size_t)s->ring_slots * 8));
    s->ring_next = 0;
  }
  Ctrl c{};
  c.first_sol_idx = ~0ull;
  c.best_bound = TB_PINF; c.foreign_bound = TB_PINF;
  HIP_TRY(hipMemcpy(s->P.ctrl, &c, sizeof(c), hipMemcpyHostToDevice));
  PeerCell pc;
  std::memset(&pc, 0, sizeof(pc));
  pc.queue = q_pack(0, 0, s->local_count);
  pc.bound = TB_PINF;
  pc.desc[0].j_base = 0; pc.desc[0].owner = s->P.rank;
  HIP_TRY(hipMemcpy(s->cell, &pc, sizeof(pc), hipMemcpyHostToDevice));
  bool linked = false;
  for (int r = 0; r < s->P.world; ++r) linked |= r != s->P.rank && s->peer_cells[(size_t)r] != nullptr;
  if (linked) {
    HIP_TRY(hipMemcpy(s->d_peers, s->peer_cells.data(), sizeof(PeerCell*) * (size_t)s->P.world, hipMemcpyHostToDevice));
    s->P.peers = s->d_peers;
  } else s->P.peers = nullptr;
  HIP_TRY(hipMemset(s->P.g_stats, 0, sizeof(BlockStats) * (size_t)s->plan.num_blocks));
  s->armed = true;
  return TB_OK;
}

int tb_session_start(tb_session* s) {
  if (!s) return fail(TB_ERR_INVALID, "null session");
  if (s->started && !s->finished) return fail(TB_ERR_STATE, "session is running");
  HIP_TRY(hipSetDevice(s->cfg.device));
  // (re)start: a finished session can be started again on the same resident inputs
  int rc;
  if (!s->armed && (rc = tb_session_arm(s)) != TB_OK) return rc;
  s->armed = false;
  s->finished = false;
  // device wall clock "now": the first poll of the mailbox is due immediately (Ctrl::next_poll holds the low 32 bits of a
  // tick and is compared by signed difference), and the in-kernel watchdog fires at now + timeout + 2 s of margin
  long long now = 0;
  if ((rc = device_now(s->stream, s->d_now, &now)) != TB_OK) return rc;
  const unsigned first_poll = (unsigned)now;
  HIP_TRY(hipMemcpyAsync(&s->P.ctrl->next_poll, &first_poll, sizeof(first_poll), hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->P.deadline_ticks = s->cfg.timeout_ms != 0 ? now + (long long)(s->cfg.timeout_ms + 2000) * (long long)s->caps.wall_khz : 0;
  s->t_start = std::chrono::steady_clock::now();
  if (s->d_P == nullptr) { int rc2 = s->bufs.alloc(&s->d_P, 1); if (rc2 != TB_OK) return rc2; }
  HIP_TRY(hipMemcpyAsync(s->d_P, &s->P, sizeof(DevProblem), hipMemcpyHostToDevice, s->stream));
  const LaunchPlan& plan = s->plan;
#ifdef TB_BOUNDS
  if ((rc = bounds_arm(s->P, plan, std::max(1, s->P.n_strats), s->strat_total, s->adj_sizes, s->P.ctrl)) != TB_OK) return rc;
#endif
  HIP_TRY(hipEventRecord(s->ev_start, s->stream));
  if (plan.team) HIP_TRY(hipMemsetAsync(s->P.teams, 0, sizeof(TeamGrid), s->stream));
  if ((rc = launch_solve(KernelSel{plan.mem_kind, plan.tmax, plan.kernel_event != 0, plan.kernel_opt}, KernelGrid{plan.num_blocks, plan.threads, plan.shared_bytes, s->stream},
                         s->P, s->d_P, s->mbox_dev)) != TB_OK) return rc;
  HIP_TRY(hipEventRecord(s->ev_stop, s->stream));
  s->started = true;
  return TB_OK;
}

int tb_session_progress(tb_session* s, uint64_t* remaining_out, uint64_t* stolen_in_out, uint64_t* stolen_out_out) {
  if (!s || !s->started) return fail(TB_ERR_STATE, "session not started");
  const unsigned long long w = __atomic_load_n(&s->mbox_host->progress, __ATOMIC_RELAXED);
  if (remaining_out) *remaining_out = q_hi(w) > q_next(w) ? q_hi(w) - q_next(w) : 0;
  if (stolen_in_out || stolen_out_out) {
    PeerCell pc;
    HIP_TRY(hipSetDevice(s->cfg.device));
    // a side copy while the kernel runs: the cell is fine-grained memory, and this is a diagnostic
    hipError_t e = hipMemcpy(&pc, s->cell, sizeof(pc), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(TB_ERR_HIP, std::string("hipMemcpy(cell): ") + hipGetErrorString(e));
    if (stolen_in_out) *stolen_in_out = pc.stolen_in;
    if (stolen_out_out) *stolen_out_out = pc.stolen_out;
  }
  return TB_OK;
}

int tb_session_debug_last_store(tb_session* s, int32_t workgroup, tb_itv* store_out) {
  if (!s || !store_out) return fail(TB_ERR_INVALID, "null argument");
  if (!s->finished) return fail(TB_ERR_STATE, "session not finished");
  if (!s->P.g_last) return fail(TB_ERR_STATE, "the session was not created with the 0x800000 test knob");
  if (workgroup < 0 || workgroup >= s->plan.num_blocks) return fail(TB_ERR_INVALID, "workgroup out of range");
  HIP_TRY(hipSetDevice(s->cfg.device));
  const size_t VX = (size_t)s->plan.vext;
  std::vector<unsigned char> slab(std::max<size_t>(16, VX * 8));
  HIP_TRY(hipMemcpy(slab.data(), s->P.g_last + (size_t)workgroup * VX, VX * 8, hipMemcpyDeviceToHost));
  decode_slab(s->lay, slab.data(), store_out);
  return TB_OK;
}

int tb_session_debug_path(tb_session* s, int32_t workgroup, tb_debug_path* path_out, int32_t capacity, tb_debug_decision* decisions_out) {
  if (!s || !path_out || capacity < 0 || (capacity > 0 && !decisions_out)) return fail(TB_ERR_INVALID, "null argument");
  if (!s->finished) return fail(TB_ERR_STATE, "session not finished");
  if (!s->P.g_path_hdr) return fail(TB_ERR_STATE, "the session was not created with the 0x800000 test knob");
  if (workgroup < 0 || workgroup >= s->plan.num_blocks) return fail(TB_ERR_INVALID, "workgroup out of range");
  HIP_TRY(hipSetDevice(s->cfg.device));
  PathHeader h;
  HIP_TRY(hipMemcpy(&h, s->P.g_path_hdr + workgroup, sizeof(h), hipMemcpyDeviceToHost));
  std::memset(path_out, 0, sizeof(*path_out));
  path_out->subproblem = h.sub_idx; path_out->dive_levels_left = h.remaining; path_out->depth = h.depth;
  path_out->last_objective_ub = h.last_obj_ub; path_out->last_node_failed = h.failed != 0 ? 1 : 0; path_out->had_work = h.has_work; path_out->nodes = h.nodes;
  const int n = std::min(std::min(h.depth, s->plan.max_depth), capacity);  // (the first segment of the decision stack: deeper entries live in the pool)
  path_out->decisions = n;
  if (n > 0) {
    std::vector<Decision> dec((size_t)n);
    std::vector<int> ub((size_t)n);
    HIP_TRY(hipMemcpy(dec.data(), s->P.g_dec + (size_t)workgroup * (size_t)s->plan.max_depth, (size_t)n * sizeof(Decision), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ub.data(), s->P.g_path_ub + (size_t)workgroup * (size_t)s->plan.max_depth, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
      tb_debug_decision& d = decisions_out[i];
      d.var = s->lay.inv[(size_t)(s->lay.c8 ? (dec[(size_t)i].var & 0xffff) : dec[(size_t)i].var)];  // the caller's numbering (COMPACT8: a decision holds a reference)
      d.child = dec[(size_t)i].cur;
      d.children[0] = tb_itv{dec[(size_t)i].child[0].x, dec[(size_t)i].child[0].y};
      d.children[1] = tb_itv{dec[(size_t)i].child[1].x, dec[(size_t)i].child[1].y};
      d.objective_ub = ub[(size_t)i];
    }
  }
  return TB_OK;
}

int tb_session_poll(tb_session* s, int32_t* local_best_out, int32_t* done_out) {
  if (!s || !s->started) return fail(TB_ERR_STATE, "session not started");
  const int lb = __atomic_load_n(&s->mbox_host->local_best, __ATOMIC_RELAXED);
  s->host_best = std::min(s->host_best, lb);  // the mailbox word may be overwritten out of order; keep the minimum seen
  if (local_best_out) *local_best_out = s->host_best;
  if (done_out) {
    hipError_t e = hipEventQuery(s->ev_stop);
    if (e == hipSuccess) *done_out = 1;
    else if (e == hipErrorNotReady) *done_out = 0;
    else return fail(TB_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
  }
  return TB_OK;
}

int tb_session_push_bound(tb_session* s, int32_t bound) {
  if (!s) return fail(TB_ERR_INVALID, "null session");
  int cur = __atomic_load_n(&s->mbox_host->foreign_bound, __ATOMIC_RELAXED);
  if (bound < cur) __atomic_store_n(&s->mbox_host->foreign_bound, bound, __ATOMIC_RELEASE);
  return TB_OK;
}

int tb_session_stop(tb_session* s) {
  if (!s) return fail(TB_ERR_INVALID, "null session");
  __atomic_store_n(&s->mbox_host->stop, 1, __ATOMIC_RELEASE);
  return TB_OK;
}

int tb_session_next_solution(tb_session* s, tb_itv* store_out, int32_t* objective_out, int32_t* has_out) {
  if (!s || !has_out) return fail(TB_ERR_INVALID, "null argument");
  *has_out = 0;
  if (!s->started) return fail(TB_ERR_STATE, "session not started");
  if (!s->ring_host) return TB_OK;  // streaming is off: there is never anything to take
  int slot = (int)(s->ring_next % (unsigned long long)s->ring_slots);
  unsigned long long seq = __atomic_load_n(&s->ring_seq()[slot], __ATOMIC_ACQUIRE);
  if (seq != s->ring_next + 1) {
    // A producer that gave up on a stop request leaves a hole in the ticket sequence: once the kernel has ended nobody
    // will fill it, so deliver what later tickets left in the ring, oldest first.
    if (hipEventQuery(s->ev_stop) != hipSuccess) { (void)hipGetLastError(); return TB_OK; }
    unsigned long long best_seq = 0;
    for (int k = 0; k < s->ring_slots; ++k) {
      const unsigned long long q = __atomic_load_n(&s->ring_seq()[k], __ATOMIC_ACQUIRE);
      if (q > s->ring_next + 1 && (best_seq == 0 || q < best_seq)) { best_seq = q; slot = k; }
    }
    if (best_seq == 0) return TB_OK;
    seq = best_seq;
    s->ring_next = seq - 1;
  }
  std::vector<tb_itv> sol((size_t)std::max(1, s->n_vars));
  decode_slab(s->lay, s->ring_data() + (size_t)slot * (size_t)s->plan.vext * 8, sol.data());
  if (store_out && s->n_vars) std::memcpy(store_out, sol.data(), (size_t)s->n_vars * sizeof(tb_itv));
  if (objective_out) *objective_out = s->obj_var >= 0 ? sol[(size_t)s->obj_var].lb : 0;
  s->ring_next += 1;
  __atomic_store_n(s->ring_consumed(), s->ring_next, __ATOMIC_RELEASE);  // frees the slot for ticket ring_next + slots - 1
  *has_out = 1;
  return TB_OK;
}

int tb_session_finish(tb_session* s, tb_itv* best_store_out, int32_t* has_solution_out, tb_stats* stats_out) {
  if (!s || !s->started || s->finished) return fail(TB_ERR_STATE, "session not running");
  HIP_TRY(hipSetDevice(s->cfg.device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->finished = true;
#ifdef TB_BOUNDS
  { const int brc = bounds_collect("search kernel"); if (brc != TB_OK) return brc; }
#endif
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, s->ev_start, s->ev_stop));
  const size_t B = (size_t)s->plan.num_blocks, V = (size_t)s->n_vars;
  std::vector<BlockStats> bst(B);
  HIP_TRY(hipMemcpy(bst.data(), s->P.g_stats, B * sizeof(BlockStats), hipMemcpyDeviceToHost));
  Ctrl c{};
  HIP_TRY(hipMemcpy(&c, s->P.ctrl, sizeof(c), hipMemcpyDeviceToHost));
#ifdef TB_TRAP_SEED
  if (c.error == 3) {
    std::fprintf(stderr, "%% trap code %d: workgroup %d node %d depth %d new_depth %d dive-left %d segs %d;", c.trap[0], c.trap[1], c.trap[2], c.trap[3], c.trap[4], c.trap[5], c.trap[6]);
    for (int i = 0; i < c.trap[7] && i < 40; ++i) std::fprintf(stderr, " %d", c.trap[8 + i]);
    std::fprintf(stderr, "\n");
  }
#endif
  if (c.error == 2) return fail(TB_ERR_STATE, "team formation failed: the " + std::to_string(s->plan.num_blocks) + " workgroups of the team kernel did not all become resident within " +
                                "TB_TEAM_JOIN_MS (10 s) -- another process, a CU mask or a concurrent kernel holds part of the GPU; TB_TEAM=0 plans one workgroup per subproblem instead");
  if (c.error != 0) return fail(TB_ERR_DEPTH, "decision stack overflow: a workgroup went deeper than " + std::to_string((MAX_DEC_SEGS + 1) * (long long)s->plan.max_depth) +
                                " decisions, or the pool of " + std::to_string(s->P.dec_pool_segments) + " extra segments of " + std::to_string(s->plan.max_depth) +
                                " decisions ran out (tb_config.decision_stack_depth sets the segment size)");

  if (s->P.blk_counts != nullptr && std::getenv("TB_BLOCK_COUNTS") != nullptr) {  // instrumented build: the block counts of this search, summed over the XCDs, as raw uint64 to the file named
    std::vector<unsigned> raw(8 * BLK_COUNT_STRIDE / sizeof(unsigned));
    HIP_TRY(hipMemcpy(raw.data(), s->P.blk_counts, raw.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    const size_t n = BLK_COUNT_STRIDE / sizeof(unsigned);
    std::vector<unsigned long long> sum(n, 0ull);
    for (size_t x = 0; x < 8; ++x) for (size_t i = 0; i < n; ++i) sum[i] += raw[x * n + i];
    if (FILE* f = std::fopen(std::getenv("TB_BLOCK_COUNTS"), "wb")) { std::fwrite(sum.data(), sizeof(unsigned long long), n, f); std::fclose(f); }
    HIP_TRY(hipMemset(s->P.blk_counts, 0, 8 * BLK_COUNT_STRIDE));
  }
  // reduce_blocks (barebones:1033-1067): sum the statistics, pick the winning workgroup.
  tb_stats st;
  std::memset(&st, 0, sizeof(st));
  st.exhaustive = 1;
  const double ns_per_tick = 1e6 / (double)s->caps.wall_khz;
  long long best_block = -1;
  long long first_idle = -1, last_idle = 0, wait_ticks = 0;
#ifdef TB_TUNING
  if (s->P.slice_census != nullptr && s->cfg.verbose) {  // per-slice run census: the slices that run most, with their keys
    std::vector<unsigned> c((size_t)s->plan.n_slices * 2);
    if (hipMemcpy(c.data(), s->P.slice_census, c.size() * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess) {
      unsigned long long nodes = 0;
      for (size_t b = 0; b < B; ++b) nodes += bst[b].nodes;
      std::vector<int> order((size_t)s->plan.n_slices);
      for (int i = 0; i < s->plan.n_slices; ++i) order[(size_t)i] = i;
      std::sort(order.begin(), order.end(), [&](int a, int b2) { return c[(size_t)a * 2] > c[(size_t)b2 * 2]; });
      const double n = (double)std::max<unsigned long long>(1, nodes);
      for (int i = 0; i < std::min(std::getenv("TB_CENSUS_ROWS") ? std::atoi(std::getenv("TB_CENSUS_ROWS")) : 40, s->plan.n_slices); ++i) {
        const int q = order[(size_t)i];
        std::fprintf(stderr, "%% slice-census %3d: slice %4d runs/node %.3f useless %.3f\n", i, q, c[(size_t)q * 2] / n, c[(size_t)q * 2 + 1] / n);
      }
    }
  }
  if (s->cfg.verbose && std::getenv("TB_PRINT_REGIONS") != nullptr) {  // region census of the search kernel (kernels.hpp: TB_REGION; scripts/region_budget.py)
    unsigned long long r[72] = {0}, nodes = 0;
    for (size_t b = 0; b < B; ++b) { for (int i = 0; i < 72; ++i) r[i] += bst[b].reg[i]; nodes += bst[b].nodes; }
    std::fprintf(stderr, "%% regions nodes=%llu", nodes);
    for (int i = 0; i < 72; ++i) std::fprintf(stderr, " %d=%llu", i, r[i]);
    std::fprintf(stderr, "\n");
  }
  if ((s->cfg.reserved[0] & 0x10000) && s->cfg.verbose) {  // wave 0's time inside the rounds of the event fixpoint (kernels.hpp: TB_PROF_MARK)
    double d[32] = {0};
    unsigned long long nodes = 0;
    for (size_t b = 0; b < B; ++b) { for (int i = 0; i < 32; ++i) d[i] += (double)bst[b].dbg[i]; nodes += bst[b].nodes; }
    const double n = (double)std::max<unsigned long long>(1, nodes);
    std::fprintf(stderr, "%% event-profile (wave 0, core cycles per node): fetch %.0f body %.0f marks %.0f barrier %.0f; runs/node %.2f rounds/node %.2f\n",
                 16 * d[0] / n, 16 * d[1] / n, 16 * d[2] / n, 16 * d[3] / n, d[4] / n, d[5] / n);
    std::fprintf(stderr, "%% event-profile body by kind: implications %.0f (%.2f runs) channelling %.0f reified-const %.0f generic %.0f (%.2f runs)\n",
                 16 * d[6] / n, d[11] / n, 16 * d[7] / n, 16 * d[8] / n, 16 * d[9] / n, d[10] / n);
    std::fprintf(stderr, "%% event-profile marks per node: runs with marks %.2f (%.1f lanes), through adjacency records %.2f (%.1f lanes), with a tail %.2f\n",
                 d[12] / n, d[15] / n, d[13] / n, d[16] / n, d[14] / n);
    std::fprintf(stderr, "%% event-profile degree (reader slices) of the variables marked through their record, per node: <=4: %.1f  5-6: %.1f  7-11: %.1f  more: %.1f\n",
                 d[17] / n, d[18] / n, d[19] / n, d[20] / n);
    std::fprintf(stderr, "%% event-profile adjacency-record branch: %.0f cycles per node, lanes marking x %.1f, y or z %.1f\n", 16 * d[21] / n, d[22] / n, d[23] / n);
    std::fprintf(stderr, "%% event-profile rounds per node by slices that run: 1: %.2f  2: %.2f  3-4: %.2f  5-8: %.2f  9-16: %.2f  more: %.2f; slices per node %.1f, of wave 0 %.1f\n",
                 d[24] / n, d[25] / n, d[26] / n, d[27] / n, d[28] / n, d[29] / n, d[30] / n, d[31] / n);
  }
#endif
  for (size_t b = 0; b < B; ++b) {
    const BlockStats& x = bst[b];
    st.nodes += x.nodes; st.fails += x.fails; st.solutions += x.solutions;
    st.fixpoint_iterations += x.fixpoint_iterations; st.num_deductions += x.num_deductions;
    st.eps_solved_subproblems += x.eps_solved; st.eps_skipped_subproblems += x.eps_skipped;
    st.num_blocks_done += (uint64_t)x.num_blocks_done;
    st.store_writes += x.store_writes;
    st.eps_stolen_subproblems += x.stolen;
    st.active_lane_evaluations += x.active_evals;
    wait_ticks += x.wait_ticks;
    last_idle = std::max(last_idle, x.timers[TB_T_FIRST_BLOCK_IDLE]);
    st.depth_max = std::max(st.depth_max, x.depth_max);
    st.exhaustive = st.exhaustive && x.exhaustive;
    st.reserved[0] |= x.why;
    if (st.reserved[1] == 0) st.reserved[1] = x.pad_why;
#ifdef TB_TUNING
    if ((x.why & 0x300) && x.dbg[1] != 0 && s->cfg.verbose)
      std::fprintf(stderr, "%% self-check: workgroup %zu node %d slice %d lane %d word0 %#x x=%d [%d,%d] y=%d [%d,%d] z=%d [%d,%d]\n", b, x.dbg[11], x.pad_why - 1, x.dbg[0],
                   (unsigned)x.dbg[1], x.dbg[2], x.dbg[5], x.dbg[6], x.dbg[3], x.dbg[7], x.dbg[8], x.dbg[4], x.dbg[9], x.dbg[10]);
#endif
    for (int t = 0; t < TB_NUM_TIMERS; ++t)
      if (t != TB_T_FIRST_BLOCK_IDLE && t != TB_T_LATEST_BEST_OBJ_FOUND) st.timers_ns[t] += (int64_t)((double)x.timers[t] * ns_per_tick);
#ifdef TB_TUNING
    for (int t = 0; t < TB_NUM_PROF; ++t) st.prof_ns[t] += (int64_t)((double)x.prof[t] * ns_per_tick);
#endif
    st.cumulative_time_block_ns += (int64_t)((double)x.timers[TB_T_FIRST_BLOCK_IDLE] * ns_per_tick);
    if (first_idle < 0 || x.timers[TB_T_FIRST_BLOCK_IDLE] < first_idle) first_idle = x.timers[TB_T_FIRST_BLOCK_IDLE];
    if (x.solutions > 0) {
      bool better;
      if (best_block < 0) better = true;
      else {
        const BlockStats& y = bst[(size_t)best_block];
        if (s->obj_var < 0 || s->cfg.use_fixed_bound) better = x.best_sub < y.best_sub;  // lowest subproblem index wins
        else better = x.best_bound < y.best_bound || (x.best_bound == y.best_bound && x.best_sub < y.best_sub);
      }
      if (better) best_block = (long long)b;
    }
  }
  st.timers_ns[TB_T_FIRST_BLOCK_IDLE] = first_idle < 0 ? 0 : (int64_t)((double)first_idle * ns_per_tick);
  st.min_block_ns = st.timers_ns[TB_T_FIRST_BLOCK_IDLE];
  st.max_block_ns = (int64_t)((double)last_idle * ns_per_tick);
  st.wait_time_ns = (int64_t)((double)wait_ticks * ns_per_tick);
  st.eps_local_subproblems = s->local_count;
  st.best_bound = TB_PINF; st.best_subproblem = -1;
  if (best_block >= 0) {
    const BlockStats& w = bst[(size_t)best_block];
    st.best_bound = s->obj_var >= 0 ? w.best_bound : 0;
    st.best_subproblem = (int32_t)std::min<long long>(w.best_sub, 0x7fffffffll);
    st.timers_ns[TB_T_LATEST_BEST_OBJ_FOUND] = (int64_t)((double)w.best_time * ns_per_tick);
    if (best_store_out && V) {
      const size_t VX = (size_t)s->plan.vext;
      std::vector<unsigned char> slab(VX * 8);
      HIP_TRY(hipMemcpy(slab.data(), s->P.g_best + (size_t)best_block * VX, VX * 8, hipMemcpyDeviceToHost));
      decode_slab(s->lay, slab.data(), best_store_out);
    }
  }
  if (has_solution_out) *has_solution_out = best_block >= 0 ? 1 : 0;
  st.eps_num_subproblems = 1ull << s->plan.subproblems_power;
  st.kernel_ns = (int64_t)((double)ms * 1e6);
  st.num_blocks = s->plan.num_blocks; st.threads_per_block = s->plan.threads;
  st.mem_kind = s->plan.mem_kind; st.shared_bytes = s->plan.shared_bytes; st.subproblems_power = s->plan.subproblems_power;
  st.interrupted = (c.stop & STOP_HOST) ? 1 : 0;
  if (st.interrupted) st.exhaustive = 0;
  // a slice that was not fully consumed is not exhaustive either (stop raised by a workgroup)
  if (stats_out) *stats_out = st;
  return TB_OK;
}

void tb_session_destroy(tb_session* s) {
  if (!s) return;
  if (s->started && !s->finished) {
    __atomic_store_n(&s->mbox_host->stop, 1, __ATOMIC_RELEASE);
    (void)hipStreamSynchronize(s->stream);
  }
  delete s;
}

int tb_solve(const tb_config* cfg_in, int32_t n_vars, const tb_itv* root_store,
             int32_t n_props, const tb_prop* props,
             int32_t n_strats, const int32_t* strat_var_order, const int32_t* strat_val_order,
             const int32_t* strat_off, const int32_t* strat_vars,
             int32_t obj_var, volatile int32_t* host_stop_flag,
             tb_itv* best_store_out, int32_t* has_solution_out, tb_stats* stats_out) {
  if (!cfg_in) return fail(TB_ERR_INVALID, "null config");
  const auto t0 = std::chrono::steady_clock::now();
  auto run = [&](const tb_config& cfg, tb_itv* best, int32_t* has, tb_stats* st) -> int {
    tb_session* s = nullptr;
    int rc = tb_session_create(&cfg, n_vars, root_store, n_props, props, n_strats, strat_var_order, strat_val_order, strat_off, strat_vars, obj_var, &s);
    if (rc != TB_OK) return rc;
    rc = tb_session_start(s);
    if (rc != TB_OK) { tb_session_destroy(s); return rc; }
    // wait_solving_ends (memory_gpu.hpp:174-196): poll, stop on timeout or on the caller's flag
    int32_t done = 0;
    auto sleep_us = std::chrono::microseconds(50);
    while (!done) {
      rc = tb_session_poll(s, nullptr, &done);
      if (rc != TB_OK) break;
      if (done) break;
      const auto now = std::chrono::steady_clock::now();
      const uint64_t el = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(now - t0).count();
      if ((cfg.timeout_ms != 0 && el >= cfg.timeout_ms) || (host_stop_flag && *host_stop_flag)) tb_session_stop(s);
      std::this_thread::sleep_for(sleep_us);
      if (sleep_us < std::chrono::microseconds(2000)) sleep_us *= 2;
    }
    if (rc == TB_OK) rc = tb_session_finish(s, best, has, st);
    tb_session_destroy(s);
    return rc;
  };
  tb_config cfg = *cfg_in;
  cfg.stream_solutions = 0;  // nobody drains a ring in the blocking call: streaming belongs to the session API
  tb_stats st;
  int32_t has = 0;
  int rc = run(cfg, best_store_out, &has, &st);
  // The reference grows a workgroup's decision stack on demand (barebones:401-403); the stacks here are sized on the
  // host, so a search that outgrows them is run again with deeper ones.
  // (capped at 2^19 decisions per segment: beyond that the retry could only end in an allocation failure that hides the depth error)
  for (int depth = cfg.decision_stack_depth > 0 ? cfg.decision_stack_depth : 16384; rc == TB_ERR_DEPTH && depth < (1 << 19);) {
    depth *= 8;
    cfg.decision_stack_depth = depth;
    const int rc2 = run(cfg, best_store_out, &has, &st);
    if (rc2 == TB_ERR_OOM) return fail(TB_ERR_DEPTH, "decision stack overflow, and no memory for segments of " + std::to_string(depth) + " decisions");
    rc = rc2;
  }
  if (rc != TB_OK) return rc;
  // Canonical pass: the B&B above proved `best_bound` optimal; the DFS-first solution under the constant
  // constraint obj <= best_bound is unique, so the answer no longer depends on the race between workgroups.
  if (cfg.deterministic && !cfg.use_fixed_bound && obj_var >= 0 && has && st.exhaustive) {
    tb_config c2 = cfg;
    c2.use_fixed_bound = 1; c2.fixed_bound = st.best_bound; c2.stop_after_n_nodes = 0;
    if (cfg.timeout_ms != 0) {
      const uint64_t el = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
      c2.timeout_ms = el + 1 < cfg.timeout_ms ? cfg.timeout_ms : el + 1000;
    }
    tb_stats st2;
    int32_t has2 = 0;
    std::vector<tb_itv> best2((size_t)std::max(1, n_vars));
    rc = run(c2, best2.data(), &has2, &st2);
    if (rc != TB_OK) return rc;
    if (has2 && !st2.interrupted) {
      if (best_store_out) std::memcpy(best_store_out, best2.data(), (size_t)n_vars * sizeof(tb_itv));
      st.best_subproblem = st2.best_subproblem;
    }
    st.nodes += st2.nodes; st.fails += st2.fails; st.fixpoint_iterations += st2.fixpoint_iterations;
    st.num_deductions += st2.num_deductions; st.kernel_ns += st2.kernel_ns; st.store_writes += st2.store_writes;
    st.active_lane_evaluations += st2.active_lane_evaluations;
  }
  if (has_solution_out) *has_solution_out = has;
  if (stats_out) *stats_out = st;
  return TB_OK;
}

}  // extern "C"
