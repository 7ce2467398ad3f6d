#include "cli_options.hpp"

#include <algorithm>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <map>
#include <sstream>

namespace turbo_host {

const char* name_of(Arch a) {
  switch (a) {
    case Arch::CPU: return "cpu";
    case Arch::GPU: return "gpu";
    case Arch::BAREBONES: return "barebones";
    default: return "hybrid";
  }
}
const char* name_of(Fixpoint f) { return f == Fixpoint::AC1 ? "ac1" : (f == Fixpoint::WAC1 ? "wac1" : (f == Fixpoint::EVENT ? "event" : "auto")); }

void usage_and_exit(const std::string& program) {
  std::cout
      << "usage: " << program
      << " [-t 2000] [-a] [-n 10] [-i] [-f] [-s] [-v] [-arch <gpu|barebones>] [-p 48] [-or 48] [-sub 12] [-subfactor 300]"
         " [-fp <ac1|wac1|event|auto>] [-wac1_threshold 0] [-eps_var_order <input_order|first_fail|anti_first_fail|smallest|largest>]"
         " [-eps_value_order <min|max|split|reverse_split>] [-seed 0] [-cutnodes 0] [-disable_simplify] [-entailed_removal] [-globalmem]"
         " [-gpus 1] [-deterministic] [-threads 0] [-version 1.0.0] [-hardware \"...\"] fzninstance.fzn\n"
      << "\t-t / -timeout <ms>: timeout in milliseconds (-timeout overrides -t).\n"
      << "\t-a: all solutions (satisfaction) / intermediate solutions (optimisation); implies -n 0 -i.\n"
      << "\t-n <k>: stop after k solutions (satisfaction problems only).\n"
      << "\t-i: print intermediate solutions (not supported by the GPU architectures, a warning is printed).\n"
      << "\t-f: free search (accepted, search annotations are still followed).\n"
      << "\t-s: print statistics.  -v: verbose (repeatable).\n"
      << "\t-arch <gpu|barebones>: both run the MI355X dive-and-solve engine (gpu: a solution has every variable assigned and -i / -a / -n stream; barebones, the default: a node whose propagators are all entailed is a solution); cpu and hybrid are not provided by this build.\n"
      << "\t-fp <ac1|wac1|event|auto>: fixpoint strategy (default auto: event from 320 propagators on, wac1 below; event: wac1 that only re-evaluates the 64-propagator slices reading a narrowed variable -- same search tree, fastest; wac1, the reference's default: each wave reaches a local fixpoint over its 64 propagators in every sweep; ac1: plain sweeps).\n"
      << "\t-or / -p <n>: number of workgroups (default 0: automatic).\n"
      << "\t-sub <d>: 2^d subproblems (default -1: at least subfactor x workgroups).  -subfactor <f>: default 300.\n"
      << "\t-cutnodes <n>: stop a workgroup after n nodes (0: no limit).  -globalmem: keep the store in global memory.\n"
      << "\t-gpus <n>: shard the subproblems over n GPUs of the node (-devices a,b,..: which ones).  -deterministic: return the DFS-first optimal solution.\n";
  std::exit(EXIT_FAILURE);
}

namespace {

struct Args {
  std::vector<std::string> tok;
  size_t consumed = 0;
  bool has(const std::string& f) const { return std::find(tok.begin(), tok.end(), f) != tok.end(); }
  // value following the first occurrence of flag f ("" if absent or last)
  std::string value_of(const std::string& f) const {
    auto it = std::find(tok.begin(), tok.end(), f);
    if (it == tok.end() || ++it == tok.end()) return "";
    return *it;
  }
};

}  // namespace

Options parse_options(int argc, char** argv) {
  Options o;
  Args a;
  for (int i = 1; i < argc; ++i) a.tok.emplace_back(argv[i]);
  const std::string program = argc > 0 ? argv[0] : "turbo";
  if (a.has("-or") && a.has("-p")) {
    std::cerr << "The options -or and -p cannot be used at the same time" << std::endl;
    usage_and_exit(program);
  }
  auto u64 = [&](const char* flag, uint64_t& dst) {
    std::string v = a.value_of(flag);
    if (v.empty()) return;
    dst = std::strtoull(v.c_str(), nullptr, 10);
    a.consumed += 2;
  };
  auto i32 = [&](const char* flag, int& dst) {
    std::string v = a.value_of(flag);
    if (v.empty()) return;
    dst = std::atoi(v.c_str());
    a.consumed += 2;
  };
  auto boolean = [&](const char* flag, bool& dst) {
    dst = a.has(flag);
    if (dst) a.consumed += 1;
  };
  auto str = [&](const char* flag, std::string& dst) -> bool {
    std::string v = a.value_of(flag);
    if (v.empty()) return false;
    dst = v;
    a.consumed += 2;
    return true;
  };
  i32("-sub", o.subproblems_power);
  u64("-subfactor", o.subproblems_factor);
  u64("-p", o.or_nodes);
  u64("-or", o.or_nodes);
  u64("-t", o.timeout_ms);
  u64("-timeout", o.timeout_ms);
  u64("-stack", o.stack_kb);
  u64("-n", o.stop_after_n_solutions);
  uint64_t cut = 0;
  u64("-cutnodes", cut);
  o.stop_after_n_nodes = cut == 0 ? UINT64_MAX : cut;
  u64("-seed", o.seed);
  boolean("-i", o.print_intermediate_solutions);
  bool all = false;
  boolean("-a", all);
  if (all) { o.stop_after_n_solutions = 0; o.print_intermediate_solutions = true; }
  boolean("-f", o.free_search);
  o.verbose = (int)std::count(a.tok.begin(), a.tok.end(), std::string("-v"));
  a.consumed += (size_t)o.verbose;
  boolean("-ast", o.print_ast);
  boolean("-s", o.print_statistics);
  boolean("-globalmem", o.only_global_memory);
  boolean("-disable_simplify", o.disable_simplify);
  boolean("-entailed_removal", o.entailed_removal);
  boolean("-force_ternarize", o.force_ternarize);
  boolean("-disable_network_analysis", o.disable_network_analysis);
  boolean("-deterministic", o.deterministic);
  i32("-gpus", o.gpus);
  {
    std::string list;
    if (str("-devices", list)) {
      for (size_t b = 0; b <= list.size();) {
        const size_t e = std::min(list.find(',', b), list.size());
        try { o.devices.push_back(std::stoi(list.substr(b, e - b))); } catch (...) { std::cerr << "Bad device list -devices " << list << std::endl; std::exit(EXIT_FAILURE); }
        b = e + 1;
      }
      o.gpus = (int)o.devices.size();
    }
  }
  i32("-threads", o.threads_per_block);
  std::string s;
  if (str("-arch", s)) {
    static const std::map<std::string, Arch> archs = {{"cpu", Arch::CPU}, {"hybrid", Arch::HYBRID}, {"gpu", Arch::GPU}, {"barebones", Arch::BAREBONES}};
    auto it = archs.find(s);
    if (it == archs.end()) { std::cerr << "Unknown architecture -arch " << s << std::endl; std::exit(EXIT_FAILURE); }
    o.arch = it->second;
  }
  if (str("-fp", s)) {
    if (s == "ac1") o.fixpoint = Fixpoint::AC1;
    else if (s == "wac1") o.fixpoint = Fixpoint::WAC1;
    else if (s == "event") o.fixpoint = Fixpoint::EVENT;
    else if (s == "auto") o.fixpoint = Fixpoint::AUTO;
    else { std::cerr << "Unknown fixpoint -fp " << s << std::endl; std::exit(EXIT_FAILURE); }
  }
  u64("-wac1_threshold", o.wac1_threshold);
  bool has_var = str("-eps_var_order", o.eps_var_order);
  bool has_val = str("-eps_value_order", o.eps_value_order);
  if (has_var != has_val) {
    std::printf("-eps_var_order and -eps_value_order must be specified together.\n");
    std::exit(EXIT_FAILURE);
  }
  str("-version", o.version);
  str("-hardware", o.hardware);
  if (a.tok.size() <= a.consumed) usage_and_exit(program);  // the input file is the last token
  o.problem_path = a.tok.back();
  if (o.gpus < 1) o.gpus = 1;
  if (o.devices.empty()) for (int g = 0; g < o.gpus; ++g) o.devices.push_back(g);
  return o;
}

std::string command_line_echo(const Options& o, const char* program) {
  std::ostringstream s;
  s << program << " -t " << o.timeout_ms << " " << (o.print_intermediate_solutions ? "-a " : "") << "-n " << o.stop_after_n_solutions << " "
    << (o.print_intermediate_solutions ? "-i " : "") << (o.free_search ? "-f " : "") << (o.print_statistics ? "-s " : "") << (o.print_ast ? "-ast " : "");
  for (int i = 0; i < o.verbose; ++i) s << "-v ";
  if (o.arch != Arch::CPU) {
    s << "-arch " << name_of(o.arch) << " -or " << o.or_nodes << " -sub " << o.subproblems_power << " -subfactor " << o.subproblems_factor << " -stack " << o.stack_kb << " ";
    if (o.only_global_memory) s << "-globalmem ";
  } else {
    s << "-arch cpu -p " << o.or_nodes << " ";
  }
  if (o.disable_simplify) s << "-disable_simplify ";
  if (o.entailed_removal) s << "-entailed_removal ";
  if (o.force_ternarize) s << "-force_ternarize ";
  if (o.disable_network_analysis) s << "-disable_network_analysis ";
  s << "-fp " << name_of(o.fixpoint) << " ";
  if (o.fixpoint == Fixpoint::WAC1) s << "-wac1_threshold " << o.wac1_threshold << " ";
  s << "-seed " << o.seed << " -eps_var_order " << o.eps_var_order << " -eps_value_order " << o.eps_value_order << " ";
  if (!o.version.empty()) s << "-version " << o.version << " ";
  if (!o.hardware.empty()) s << "-hardware '" << o.hardware << "' ";
  s << "-cutnodes " << (o.stop_after_n_nodes == UINT64_MAX ? 0 : o.stop_after_n_nodes) << " " << o.problem_path;
  return s.str();
}

void print_config_statistics(const Options& o) {
  auto stat_s = [](const char* k, const std::string& v) { std::printf("%%%%%%mzn-stat: %s=\"%s\"\n", k, v.c_str()); };
  auto stat_u = [](const char* k, uint64_t v) { std::printf("%%%%%%mzn-stat: %s=%" PRIu64 "\n", k, v); };
  stat_s("problem_path", o.problem_path);
  stat_s("solver", "Turbo");
  stat_s("version", o.version.empty() ? "1.3.0-mi355x" : o.version);
  stat_s("hardware", o.hardware.empty() ? "unspecified" : o.hardware);
  stat_s("arch", name_of(o.arch));
  stat_s("fixpoint", name_of(o.fixpoint));
  stat_u("subproblems_factor", o.subproblems_factor);
  if (o.fixpoint == Fixpoint::WAC1) stat_u("wac1_threshold", o.wac1_threshold);
  stat_u("seed", o.seed);
  stat_s("eps_var_order", o.eps_var_order);
  stat_s("eps_value_order", o.eps_value_order);
  stat_s("free_search", o.free_search ? "yes" : "no");
  stat_u("or_nodes", o.or_nodes);
  stat_u("timeout_ms", o.timeout_ms);
  if (o.arch != Arch::CPU) stat_u("stack_size", o.stack_kb * 1000);
  stat_u("cutnodes", o.stop_after_n_nodes == UINT64_MAX ? 0 : o.stop_after_n_nodes);
}

}  // namespace turbo_host
