// `turbo` executable: the reference's command line and output protocol in front of the MI355X engine.
//
// Mirrors  src/turbo.cpp:22-56 (main, arch dispatch, exception -> exit 1),
//          include/barebones_dive_and_solve.hpp:464-518 (host side of the solve: preprocess, run, print),
//          include/common_solving.hpp:829-896 (solution / statistics printing),
//          include/statistics.hpp:338-412 (mzn-stat keys and separators).
// The solving itself is libturbo_hip.so through its C-ABI; there is no CPU solving path in this binary.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cinttypes>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/turbo_front.h"
#include "../../../include/turbo_hip.h"
#include "cli_options.hpp"

using namespace turbo_host;
using Clock = std::chrono::steady_clock;

namespace {

volatile int32_t g_stop_flag = 0;  // set by SIGINT / SIGTERM (common_solving.hpp:56-85)
void on_signal(int) { g_stop_flag = 1; }

struct Printer {
  bool on;
  void s(const char* k, const char* v) const { if (on) std::printf("%%%%%%mzn-stat: %s=\"%s\"\n", k, v); }
  void u(const char* k, uint64_t v) const { if (on) std::printf("%%%%%%mzn-stat: %s=%" PRIu64 "\n", k, v); }
  void i(const char* k, int v) const { if (on) std::printf("%%%%%%mzn-stat: %s=%d\n", k, v); }
  void d(const char* k, double v) const { if (on) std::printf("%%%%%%mzn-stat: %s=%lf\n", k, v != v ? 0.0 : v); }
  void end() const { if (on) std::printf("%%%%%%mzn-stat-end\n"); }
};

double to_sec(int64_t ns) { return (double)(ns / 1000 / 1000) / 1000.0; }  // statistics.hpp:321-323

void final_separator(uint64_t solutions, bool exhaustive, bool optimization) {  // statistics.hpp:394-412
  if (solutions > 0) { if (exhaustive) std::printf("==========\n"); }
  else if (exhaustive) std::printf("=====UNSATISFIABLE=====\n");
  else if (optimization) std::printf("=====UNBOUNDED=====\n");
  else std::printf("=====UNKNOWN=====\n");
}

void print_solve_statistics(const Printer& p, const Options&, const tb_stats& st, int n_vars, int n_props, int64_t preprocessing_ns, int64_t overall_ns) {
  const int nb = std::max(1, st.num_blocks);
  p.i("num_blocks", st.num_blocks);
  p.u("nodes", st.nodes);
  p.u("failures", st.fails);
  p.u("variables", (uint64_t)n_vars);
  p.u("propagators", (uint64_t)n_props);
  p.i("peakDepth", st.depth_max);
  p.d("initTime", to_sec(preprocessing_ns));
  p.d("solveTime", to_sec(overall_ns));
  p.u("num_solutions", st.solutions);
  p.u("eps_num_subproblems", st.eps_num_subproblems);
  p.u("eps_solved_subproblems", st.eps_solved_subproblems);
  p.u("eps_skipped_subproblems", st.eps_skipped_subproblems);
  p.u("num_blocks_done", st.num_blocks_done);
  p.u("fixpoint_iterations", st.fixpoint_iterations);
  p.u("num_deductions", st.num_deductions);
  p.d("cumulative_time_block_sec", to_sec(st.cumulative_time_block_ns));
  p.d("deductions_per_block_second", (double)(st.num_deductions / (uint64_t)nb) / to_sec(st.cumulative_time_block_ns));
  p.d("solve_time", to_sec(st.timers_ns[TB_T_OVERALL] / nb));
  p.d("search_time", to_sec(st.timers_ns[TB_T_SEARCH] / nb));
  p.d("fixpoint_time", to_sec(st.timers_ns[TB_T_FIXPOINT] / nb));
  p.d("transfer_cpu2gpu_time", to_sec(st.timers_ns[TB_T_TRANSFER_CPU2GPU] / nb));
  p.d("transfer_gpu2cpu_time", to_sec(st.timers_ns[TB_T_TRANSFER_GPU2CPU] / nb));
  p.d("select_fp_functions_time", to_sec(st.timers_ns[TB_T_SELECT_FP_FUNCTIONS] / nb));
  p.d("wait_cpu_time", to_sec(st.timers_ns[TB_T_WAIT_CPU] / nb));
  p.d("dive_time", to_sec(st.timers_ns[TB_T_DIVE] / nb));
  p.d("best_obj_time", to_sec(st.timers_ns[TB_T_LATEST_BEST_OBJ_FOUND]));
  p.d("first_block_idle_time", to_sec(st.timers_ns[TB_T_FIRST_BLOCK_IDLE]));
  // engine-specific keys: additions BEHIND the reference's block (statistics.hpp:338-371 keeps its sequence), never replacing a reference key
  p.u("eps_stolen_subproblems", st.eps_stolen_subproblems);  // work moved between GPUs
  p.d("wait_for_work_time", to_sec(st.wait_time_ns / nb));
  p.d("kernel_time", (double)st.kernel_ns * 1e-9);
  p.d("propagations_per_second", st.kernel_ns > 0 ? (double)st.num_deductions / ((double)st.kernel_ns * 1e-9) : 0.0);
  p.d("nodes_per_second", st.kernel_ns > 0 ? (double)st.nodes / ((double)st.kernel_ns * 1e-9) : 0.0);
}

void merge_stats(tb_stats& a, const tb_stats& b) {  // statistics.hpp:182-196 across GPUs
  a.nodes += b.nodes; a.fails += b.fails; a.solutions += b.solutions;
  a.fixpoint_iterations += b.fixpoint_iterations; a.num_deductions += b.num_deductions;
  a.eps_solved_subproblems += b.eps_solved_subproblems; a.eps_skipped_subproblems += b.eps_skipped_subproblems;
  a.num_blocks_done += b.num_blocks_done; a.store_writes += b.store_writes;
  a.depth_max = std::max(a.depth_max, b.depth_max);
  a.exhaustive = a.exhaustive && b.exhaustive;
  a.interrupted = a.interrupted || b.interrupted;
  a.num_blocks += b.num_blocks;
  for (int t = 0; t < TB_NUM_TIMERS; ++t)
    if (t != TB_T_FIRST_BLOCK_IDLE && t != TB_T_LATEST_BEST_OBJ_FOUND) a.timers_ns[t] += b.timers_ns[t];
  a.timers_ns[TB_T_FIRST_BLOCK_IDLE] = std::min(a.timers_ns[TB_T_FIRST_BLOCK_IDLE], b.timers_ns[TB_T_FIRST_BLOCK_IDLE]);
  a.cumulative_time_block_ns += b.cumulative_time_block_ns;
  a.kernel_ns = std::max(a.kernel_ns, b.kernel_ns);
  a.eps_local_subproblems += b.eps_local_subproblems; a.eps_stolen_subproblems += b.eps_stolen_subproblems;
  a.wait_time_ns += b.wait_time_ns;
  a.min_block_ns = std::min(a.min_block_ns, b.min_block_ns); a.max_block_ns = std::max(a.max_block_ns, b.max_block_ns);
}

tb_config make_config(const Options& o, bool has_eps) {
  tb_config c;
  std::memset(&c, 0, sizeof(c));
  c.timeout_ms = o.timeout_ms; c.or_nodes = o.or_nodes; c.subproblems_factor = o.subproblems_factor;
  c.stop_after_n_nodes = o.stop_after_n_nodes == UINT64_MAX ? 0 : o.stop_after_n_nodes;
  c.stop_after_n_solutions = o.stop_after_n_solutions; c.wac1_threshold = o.wac1_threshold;
  c.subproblems_power = o.subproblems_power; c.fixpoint = o.fixpoint == Fixpoint::AC1 ? 0 : (o.fixpoint == Fixpoint::WAC1 ? 1 : (o.fixpoint == Fixpoint::EVENT ? 2 : 3));
  c.only_global_memory = o.only_global_memory; c.verbose = o.verbose; c.has_eps_strategy = has_eps;
  c.threads_per_block = o.threads_per_block; c.device = 0; c.rank = 0; c.world_size = 1;
  c.deterministic = o.deterministic;
  c.entailed_prop_removal = o.entailed_removal ? 1 : 0;
  // the leaf rule follows the architecture, as in the reference: `-arch gpu` calls an all-entailed node a solution only when the store is extractable
  // (every variable assigned, gpu_dive_and_solve.hpp:333-338), `-arch barebones` accepts the box (barebones_dive_and_solve.hpp:988-993)
  c.leaf_requires_assignment = o.arch == Arch::GPU ? 1 : 0;
  return c;
}

std::string array_text(const std::vector<int32_t>& v) {  // statistics.hpp:31-43
  std::string s = "[";
  for (size_t i = 0; i < v.size(); ++i) s += (i ? ", " : "") + std::to_string(v[i]);
  return s + "]";
}

// analyze_tcn (common_solving.hpp:728-826): size of the ternary network and, unless -disable_network_analysis,
// its histograms.  Symbols: a comparison whose truth variable is the constant 0 counts as its negation, one whose
// truth variable is not assigned counts as reified as well.  A variable is "assigned" when lb = ub (the domain-size
// histogram is keyed by ub - lb + 1).
void analyze_tcn(const Options& o, const Printer& p, const std::string& prefix, const tf_model* m) {
  const int V = tf_num_vars(m), P = tf_num_props(m);
  p.u((prefix + "_variables").c_str(), (uint64_t)V);
  p.u((prefix + "_constraints").c_str(), (uint64_t)P);
  if (o.disable_network_analysis || !o.print_statistics) return;
  if (o.verbose) std::printf("%% Analyzing the ternary constraint network...\n");
  const tb_itv* store = tf_store(m);
  const tb_prop* props = tf_props(m);
  static const char* sym[] = {"+", "*", "/", "%", "min", "max", "=", "<="};
  std::map<std::string, uint64_t> ops, reified;
  std::vector<uint64_t> occ((size_t)V, 0);
  for (int i = 0; i < P; ++i) {
    const tb_prop& q = props[i];
    occ[(size_t)q.x]++; occ[(size_t)q.y]++; occ[(size_t)q.z]++;
    const tb_itv x = store[q.x];
    if (q.op == TB_EQ || q.op == TB_LEQ) {
      if (x.lb == x.ub) ops[x.lb == 0 ? (q.op == TB_EQ ? "!=" : ">") : sym[q.op]]++;
      else { ops[sym[q.op]]++; reified[sym[q.op]]++; }
    } else ops[sym[q.op]]++;
  }
  std::map<uint64_t, uint64_t> deg_assigned, deg_unassigned, dom_size;
  uint64_t assigned = 0, unbounded = 0, occ_assigned = 0, occ_unassigned = 0;
  for (int v = 0; v < V; ++v) {
    const tb_itv d = store[v];
    const bool inf = d.lb == TB_NINF || d.ub == TB_PINF;
    if (inf) unbounded++; else dom_size[(uint64_t)((int64_t)d.ub - (int64_t)d.lb + 1)]++;
    if (!inf && d.lb == d.ub) { assigned++; deg_assigned[occ[(size_t)v]]++; occ_assigned += occ[(size_t)v]; }
    else { deg_unassigned[occ[(size_t)v]]++; occ_unassigned += occ[(size_t)v]; }
  }
  auto dict_s = [](const std::map<std::string, uint64_t>& d) { std::string s = "{"; bool f = true; for (auto& kv : d) { s += (f ? "'" : ", '") + kv.first + "': " + std::to_string(kv.second); f = false; } return s + "}"; };
  auto dict_u = [](const std::map<uint64_t, uint64_t>& d) { std::string s = "{"; bool f = true; for (auto& kv : d) { s += (f ? "" : ", ") + std::to_string(kv.first) + ": " + std::to_string(kv.second); f = false; } return s + "}"; };
  p.u((prefix + "_assigned_variables").c_str(), assigned);
  p.u((prefix + "_unbounded_variables").c_str(), unbounded);
  p.u((prefix + "_unassigned_var_occurrences").c_str(), occ_unassigned);
  p.u((prefix + "_assigned_var_occurrences").c_str(), occ_assigned);
  p.s((prefix + "_histogram_symbols").c_str(), dict_s(ops).c_str());
  p.s((prefix + "_histogram_reified_predicates").c_str(), dict_s(reified).c_str());
  p.s((prefix + "_histogram_unassigned_vars_degree").c_str(), dict_u(deg_unassigned).c_str());
  p.s((prefix + "_histogram_assigned_vars_degree").c_str(), dict_u(deg_assigned).c_str());
  p.s((prefix + "_histogram_vars_dom_size").c_str(), dict_u(dom_size).c_str());
}

// The preprocessing loop of common_solving.hpp:537-585: root fixpoint (on the GPU, `tb_propagate`), then the
// network simplifier, until neither changes anything.  The propagators are not restated on the host.
bool simplify_network(const Options& o, const Printer& p, tf_model* m, std::string& err) {
  std::vector<int32_t> icse, algsimp, algsimp_eq, entailed, useless;
  tb_config c;
  std::memset(&c, 0, sizeof(c));
  c.fixpoint = 1; c.only_global_memory = o.only_global_memory; c.verbose = 0; c.world_size = 1;
  for (int round = 0; round < 16 && !tf_trivially_unsat(m); ++round) {
    const int n_vars = tf_num_vars(m), n_props = tf_num_props(m);
    std::vector<tb_itv> root(tf_store(m), tf_store(m) + n_vars);
    if (n_props > 0) {
      int32_t failed = 0;
      c.timeout_ms = o.timeout_ms;  // the root fixpoint obeys -t like the search does
      const int rc = tb_propagate(&c, n_vars, n_props, tf_props(m), 1, root.data(), &failed, nullptr, nullptr, nullptr, nullptr);
      if (rc != TB_OK) { err = tb_last_error(); return false; }
      // failed = 1: the root is inconsistent -- say so explicitly instead of relying on the kernel having left an empty
      // interval behind (variable 0 is the constant 0: [1, 0] is the empty interval the simplifier looks for);
      // failed = -1: the watchdog cut the fixpoint short; what was narrowed so far is still sound, go on with it
      if (failed == 1) { root[0].lb = 1; root[0].ub = 0; }
      if (failed == -1 && o.verbose) std::printf("%% The root fixpoint was interrupted by the timeout.\n");
    }
    int32_t st[9];
    if (tf_simplify(m, root.data(), st) != 0) { err = "the network simplifier failed"; return false; }
    icse.push_back(st[5]); algsimp.push_back(st[7]); algsimp_eq.push_back(st[4]); entailed.push_back(st[6]); useless.push_back(st[8]);
    if (st[4] == 0 && st[5] == 0 && st[6] == 0 && st[7] == 0 && st[8] == 0) break;
  }
  p.s("preprocessing_icse_eliminated_constraints", array_text(icse).c_str());
  p.s("preprocessing_algsimp_eliminated_constraints", array_text(algsimp).c_str());
  p.s("preprocessing_algsimp_eliminated_eq_constraints", array_text(algsimp_eq).c_str());
  p.s("preprocessing_entailment_eliminated_constraints", array_text(entailed).c_str());
  p.s("preprocessing_eliminated_variables", array_text(useless).c_str());
  if (o.verbose) std::printf("%% Formula simplified.\n");
  if (!tf_trivially_unsat(m)) {
    analyze_tcn(o, p, "preprocessed_tcn", m);
  }
  return true;
}

// Prints the solutions handed over while the kernels run (consume_solution, gpu_dive_and_solve.hpp:116-132).
struct SolutionPrinter {
  const tf_model* m;
  bool optimization;
  uint64_t limit;        // satisfaction: -n (0 = all)
  uint64_t printed = 0;
  int32_t last_obj = TB_PINF;
  void print(const tb_itv* store) {
    const int32_t need = tf_format_solution(m, store, nullptr, 0);
    std::string text((size_t)need + 1, '\0');
    tf_format_solution(m, store, text.data(), need + 1);
    std::fputs(text.c_str(), stdout);
    std::printf("----------\n");
    std::fflush(stdout);
    ++printed;
  }
  // Workgroups race: an optimisation solution is printed only if it improves on the last one printed.
  void offer(const tb_itv* store, int32_t obj) {
    if (optimization) { if (obj < last_obj) { last_obj = obj; print(store); } }
    else if (limit == 0 || printed < limit) print(store);
  }
  bool satisfied() const { return !optimization && limit != 0 && printed >= limit; }
};

// N GPUs of the node (N >= 1): one session per device, this thread relays the incumbent bound and the stop flag and,
// when streaming, prints the solutions as they arrive.  `segment` = tb_config.decision_stack_depth (0: the engine's default).
int solve_sessions_once(const Options& o, const tb_config& base, const tf_model* m, int segment, std::vector<tb_itv>& best, int32_t* has, tb_stats* out, SolutionPrinter* printer) {
  const int G = std::max(1, o.gpus);
  std::vector<tb_session*> ss((size_t)G, nullptr);
  int rc = TB_OK;
  auto create_all = [&](int sub_power) {
    for (int g = 0; g < G && rc == TB_OK; ++g) {
      tb_config c = base;
      c.device = o.devices[(size_t)g]; c.rank = g; c.world_size = G; c.deterministic = 0;
      c.stream_solutions = printer ? 1 : 0;
      c.decision_stack_depth = segment;
      if (sub_power >= 0) c.subproblems_power = sub_power;
      rc = tb_session_create(&c, tf_num_vars(m), tf_store(m), tf_num_props(m), tf_props(m), tf_num_strategies(m), tf_strat_var_order(m),
                             tf_strat_val_order(m), tf_strat_off(m), tf_strat_vars(m), tf_obj_var(m), &ss[(size_t)g]);
    }
  };
  create_all(-1);
  // The block-cyclic shares only tile the 2^d index space if every rank planned the same d (and chunking): the sessions size
  // themselves on their own device (free memory, occupancy), so compare -- and plan again with the largest d when they differ.
  if (rc == TB_OK && G > 1) {
    int dmin = 1 << 30, dmax = -1;
    bool same_chunk = true;
    tb_plan p0{};
    for (int g = 0; g < G && rc == TB_OK; ++g) {
      tb_plan pl{};
      rc = tb_session_plan(ss[(size_t)g], &pl);
      if (g == 0) p0 = pl;
      dmin = std::min(dmin, pl.subproblems_power); dmax = std::max(dmax, pl.subproblems_power);
      same_chunk = same_chunk && pl.eps_chunk_log2 == p0.eps_chunk_log2;
    }
    if (rc == TB_OK && (dmin != dmax || !same_chunk)) {
      if (o.verbose) std::printf("%% The GPUs planned 2^%d .. 2^%d subproblems: planning again with 2^%d on every GPU.\n", dmin, dmax, dmax);
      for (tb_session*& s : ss) { tb_session_destroy(s); s = nullptr; }
      create_all(dmax);
      for (int g = 0; g < G && rc == TB_OK; ++g) {
        tb_plan pl{};
        rc = tb_session_plan(ss[(size_t)g], &pl);
        if (rc == TB_OK && (pl.subproblems_power != dmax || pl.eps_chunk_log2 != std::max(0, std::min(base.eps_chunk_log2, dmax)))) {  // (the engine clamps the chunk to 0 .. d)
          std::cerr << "the GPUs of this search cannot agree on one subproblem count (2^" << pl.subproblems_power << " on GPU " << g << ", 2^" << dmax << " wanted)" << std::endl;
          for (tb_session* s : ss) tb_session_destroy(s);
          return TB_ERR_INVALID;
        }
      }
    }
  }
  // The kernels exchange the incumbent bound and rebalance work among themselves over xGMI (include/turbo_hip.h:
  // tb_session_link_peer); without a peer path between two devices the relay below (poll / push_bound) carries the bound.
  bool all_linked = true;
  for (int a = 0; a < G && rc == TB_OK; ++a)
    for (int b = 0; b < G && rc == TB_OK; ++b)
      if (a != b && tb_session_link_peer(ss[(size_t)a], ss[(size_t)b]) != TB_OK) {
        all_linked = false;
        if (o.verbose) std::printf("%% GPUs %d and %d are not linked (%s): the host relays the bound.\n", a, b, tb_last_error());
      }
  // (all or nothing: a partly linked group would count its node budget in a cell that not every GPU adds to)
  if (!all_linked) for (int g = 0; g < G && rc == TB_OK; ++g) rc = tb_session_unlink_peers(ss[(size_t)g]);
  for (int g = 0; g < G && rc == TB_OK; ++g) rc = tb_session_arm(ss[(size_t)g]);  // every cell is reset before any kernel can touch it
  for (int g = 0; g < G && rc == TB_OK; ++g) rc = tb_session_start(ss[(size_t)g]);
  std::vector<tb_itv> tmp(best.size());
  auto drain = [&]() {
    if (!printer) return;
    for (int g = 0; g < G && rc == TB_OK; ++g)
      for (;;) {
        int32_t got = 0, obj = 0;
        rc = tb_session_next_solution(ss[(size_t)g], tmp.data(), &obj, &got);
        if (rc != TB_OK || !got) break;
        printer->offer(tmp.data(), obj);
      }
  };
  const auto t0 = Clock::now();
  while (rc == TB_OK) {
    int32_t gbest = TB_PINF, all_done = 1;
    for (int g = 0; g < G; ++g) {
      int32_t b = TB_PINF, d = 0;
      rc = tb_session_poll(ss[(size_t)g], &b, &d);
      if (rc != TB_OK) break;
      gbest = std::min(gbest, b);
      all_done &= d;
    }
    if (rc != TB_OK) break;
    drain();
    if (rc != TB_OK || all_done) break;
    if (G > 1 && gbest != TB_PINF) for (int g = 0; g < G; ++g) tb_session_push_bound(ss[(size_t)g], gbest);
    const uint64_t el = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(Clock::now() - t0).count();
    if ((o.timeout_ms != 0 && el >= o.timeout_ms) || g_stop_flag || (printer && printer->satisfied())) for (int g = 0; g < G; ++g) tb_session_stop(ss[(size_t)g]);
    std::this_thread::sleep_for(std::chrono::microseconds(printer ? 200 : 500));
  }
  // reduce_blocks across the GPUs (barebones:1033-1067): best bound; ties go to the lowest SUBPROBLEM INDEX -- with block-cyclic
  // shares and work stealing the lowest rank does not hold the lowest indices
  bool first = true;
  int32_t best_bound = TB_PINF, best_sub = INT32_MAX;
  *has = 0;
  for (int g = 0; g < G && rc == TB_OK; ++g) {
    tb_stats st;
    int32_t h = 0;
    rc = tb_session_finish(ss[(size_t)g], tmp.data(), &h, &st);
    if (rc != TB_OK) break;
    const int32_t sub = st.best_subproblem >= 0 ? st.best_subproblem : INT32_MAX;
    if (h && (!*has || st.best_bound < best_bound || (st.best_bound == best_bound && sub < best_sub))) { *has = 1; best_bound = st.best_bound; best_sub = sub; best = tmp; }
    if (first) { *out = st; first = false; } else merge_stats(*out, st);
  }
  out->best_bound = best_bound;
  out->best_subproblem = *has && best_sub != INT32_MAX ? best_sub : -1;
  for (tb_session* s : ss) tb_session_destroy(s);
  return rc;
}

// The reference grows a block's decision stack on demand (barebones:401-403).  The engine does so in the kernel (segments from a
// pool); a search that outgrows even that ends with TB_ERR_DEPTH and is run again with 8x larger segments, as tb_solve does.
// (Not after solutions of a satisfaction problem were already printed: they would be printed twice.)
int solve_sessions(const Options& o, const tb_config& base, const tf_model* m, std::vector<tb_itv>& best, int32_t* has, tb_stats* out, SolutionPrinter* printer) {
  int segment = base.decision_stack_depth;
  int rc = solve_sessions_once(o, base, m, segment, best, has, out, printer);
  // (the first segments of all workgroups are one allocation of num_blocks x depth x 32 B: the retry stops at 2^19 decisions per segment --
  //  16 segments of that are 8 M decisions per workgroup -- instead of ending in an allocation failure that hides the depth error)
  for (int depth = segment > 0 ? segment : 16384; rc == TB_ERR_DEPTH && depth < (1 << 19) && (!printer || printer->optimization || printer->printed == 0);) {
    depth *= 8;
    if (o.verbose) std::printf("%% A decision stack overflowed: searching again with segments of %d decisions.\n", depth);
    rc = solve_sessions_once(o, base, m, depth, best, has, out, printer);
  }
  return rc;
}

}  // namespace

int main(int argc, char** argv) {
  Options o = parse_options(argc, argv);
  Printer p{o.print_statistics};
  if (o.print_statistics) std::printf("%%%%%%mzn-stat: command_line=\"%s\"\n", command_line_echo(o, argv[0]).c_str());
  if (o.arch == Arch::CPU || o.arch == Arch::HYBRID) {
    std::cerr << "-arch " << name_of(o.arch) << " is not provided by this build: it contains only the MI355X dive-and-solve engine (use -arch gpu or -arch barebones)." << std::endl;
    return EXIT_FAILURE;
  }
  const auto start = Clock::now();

  // preprocess (common_solving.hpp:605-637)
  auto has_ext = [&](const char* e) { const size_t n = std::strlen(e); return o.problem_path.size() >= n && o.problem_path.compare(o.problem_path.size() - n, n, e) == 0; };
  const bool is_fzn = has_ext(".fzn"), is_xcsp3 = has_ext(".xml");  // config.hpp:268-278
  if (!is_fzn && !is_xcsp3) {
    std::printf("ERROR: Unknown input format for the file %s [supported extension: .xml and .fzn].\n", o.problem_path.c_str());
    return EXIT_FAILURE;
  }
  char err[1024] = {0};
  tf_model* m = is_fzn ? tf_load_fzn(o.problem_path.c_str(), err, sizeof(err)) : tf_load_xcsp3(o.problem_path.c_str(), err, sizeof(err));
  if (!m) {
    std::cerr << "Could not parse input file." << std::endl;
    if (o.verbose) std::cerr << err << std::endl;
    return EXIT_FAILURE;
  }
  p.u("parsed_variables", (uint64_t)tf_parsed_variables(m));
  p.u("parsed_constraints", (uint64_t)tf_parsed_constraints(m));
  if (!o.disable_network_analysis && o.print_statistics) {  // analyze_cn, common_solving.hpp:608-610
    const std::string lines = tf_fcn_statistics(m);
    for (size_t b = 0, e; (e = lines.find('\n', b)) != std::string::npos; b = e + 1) std::printf("%%%%%%mzn-stat: %s\n", lines.substr(b, e - b).c_str());
  }
  p.s("abstract_domain", "pir_itv32_z");
  analyze_tcn(o, p, "tcn", m);
  if (!o.disable_simplify && !tf_trivially_unsat(m)) {
    std::string err_text;
    if (!simplify_network(o, p, m, err_text)) {
      std::cout.flush();
      std::cerr << "\n\tUnexpected exception:\n\t" << err_text << std::endl;
      tf_free(m);
      return EXIT_FAILURE;
    }
  }
  bool has_eps = false;
  if (o.eps_var_order != "default") {
    static const char* vo[] = {"input_order", "first_fail", "anti_first_fail", "smallest", "largest"};
    static const char* vl[] = {"min", "max", "split", "reverse_split"};
    int vi = -1, li = -1;
    for (int k = 0; k < 5; ++k) if (o.eps_var_order == vo[k]) vi = k;
    if (o.eps_var_order == "random") vi = 0;
    for (int k = 0; k < 4; ++k) if (o.eps_value_order == vl[k]) li = k;
    if (vi < 0) { std::printf("Unrecognized option `-eps_var_order %s`\n", o.eps_var_order.c_str()); return EXIT_FAILURE; }
    if (li < 0) { std::printf("Unrecognized option `-eps_value_order %s`\n", o.eps_value_order.c_str()); return EXIT_FAILURE; }
    tf_push_eps_strategy(m, vi, li);
    if (o.eps_var_order == "random") tf_shuffle_strategy(m, 0, o.seed);  // common_solving.hpp:632-633
    has_eps = true;
  }
  const int n_vars = tf_num_vars(m), n_props = tf_num_props(m);
  {
    // what the engine will run on this network (tb_config.fixpoint = 3 resolves to the event-driven fixpoint from TB_AUTO_EVENT_MIN_PROPS propagators
    // on, to WAC1 sweeps below: tb_plan.kernel_event / kernel_opt of the session): the event kernels always drop entailed slices,
    // the sweeps only with -entailed_removal
    const bool event = o.fixpoint == Fixpoint::EVENT || (o.fixpoint == Fixpoint::AUTO && n_props >= TB_AUTO_EVENT_MIN_PROPS);
    p.s("entailed_prop_removal", (event || o.entailed_removal) ? "by_slice_entailment" : "deactivated");
  }
  const int64_t preprocessing_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - start).count();
  p.d("preprocessing_time", to_sec(preprocessing_ns));
  p.end();

  const bool optimization = tf_goal(m) != 0;
  tb_stats st;
  std::memset(&st, 0, sizeof(st));
  st.exhaustive = 1; st.num_blocks = 1;
  if (tf_trivially_unsat(m)) {  // cp.iprop->is_bot() after preprocessing (barebones:474-478)
    final_separator(0, true, optimization);
    if (o.print_statistics) {
      print_config_statistics(o);
      print_solve_statistics(p, o, st, n_vars, n_props, preprocessing_ns, std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - start).count());
      p.end();
    }
    tf_free(m);
    return 0;
  }

  std::signal(SIGINT, on_signal);
  std::signal(SIGTERM, on_signal);
  tb_config cfg = make_config(o, has_eps);
  std::vector<tb_itv> best((size_t)std::max(1, n_vars));
  int32_t has = 0;
  int rc;
  // Streaming as in the reference's `gpu` path (is_printing_intermediate_sol, common_solving.hpp:838-840): `-i`/`-a`, and
  // satisfaction problems asked for more than one solution.  Otherwise only the final best solution is printed
  // (barebones:499-506), which keeps the answer of a satisfaction problem deterministic (lowest subproblem).
  const bool streaming = o.print_intermediate_solutions || (!optimization && o.stop_after_n_solutions != 1);
  SolutionPrinter printer{m, optimization, o.stop_after_n_solutions};
  if (o.gpus <= 1 && !streaming) {
    rc = tb_solve(&cfg, n_vars, tf_store(m), n_props, tf_props(m), tf_num_strategies(m), tf_strat_var_order(m), tf_strat_val_order(m),
                  tf_strat_off(m), tf_strat_vars(m), tf_obj_var(m), &g_stop_flag, best.data(), &has, &st);
  } else {
    rc = solve_sessions(o, cfg, m, best, &has, &st, streaming ? &printer : nullptr);
  }
  if (rc != TB_OK) {
    std::cout.flush();
    std::cerr << "\n\tUnexpected exception:\n\t" << tb_last_error() << std::endl;
    tf_free(m);
    return EXIT_FAILURE;
  }
  const int64_t overall_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - start).count();

  if (has) {
    // without streaming: the final best solution (barebones:499-506); with streaming it was printed when it arrived,
    // unless its hand-over was cut short by a stop request
    const bool pending = !streaming || (optimization ? best[(size_t)tf_obj_var(m)].lb < printer.last_obj : printer.printed == 0);
    if (pending) printer.print(best.data());
  }
  final_separator(st.solutions, st.exhaustive != 0, optimization);
  if (o.print_statistics) {
    static const char* mem[] = {"global", "store_shared", "tcn_shared"};
    p.s("memory_configuration", mem[std::min(2, std::max(0, st.mem_kind))]);
    p.u("shared_mem", (uint64_t)st.shared_bytes);
    p.u("store_mem", (uint64_t)n_vars * 8);
    p.u("propagator_mem", (uint64_t)n_props * 16);
    p.i("subproblems_power", st.subproblems_power);
    p.end();
    print_config_statistics(o);
    p.i("threads_per_block", st.threads_per_block);
    print_solve_statistics(p, o, st, n_vars, n_props, preprocessing_ns, overall_ns);
    if (optimization && has) std::printf("%%%%%%mzn-stat: objective=%" PRId64 "\n", tf_objective_of(m, best.data()));
    p.end();
  }
  tf_free(m);
  return 0;
}
