// Command line of the `turbo` executable: same flags, defaults and error behaviour as the reference
// (src/config.cpp:11-220, include/config.hpp:33-105), own implementation (table driven).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace turbo_host {

enum class Arch { CPU, GPU, BAREBONES, HYBRID };
enum class Fixpoint { AC1, WAC1, EVENT, AUTO };  // EVENT: this engine's event-driven WAC1; AUTO: EVENT from 2048 propagators on, else WAC1 (not reference flags)

struct Options {
  bool print_intermediate_solutions = false;  // -i / -a
  uint64_t stop_after_n_solutions = 1;        // -n
  uint64_t stop_after_n_nodes = UINT64_MAX;   // -cutnodes (0 printed when unlimited)
  bool free_search = false;                   // -f
  bool print_statistics = false;              // -s
  int verbose = 0;                            // -v (repeatable)
  bool print_ast = false;                     // -ast
  bool only_global_memory = false;            // -globalmem
  bool force_ternarize = false;
  bool disable_simplify = false;
  bool entailed_removal = false;              // -entailed_removal: the reference's TURBO_NO_ENTAILED_PROP_REMOVAL=OFF build, at run time
  bool disable_network_analysis = false;
  uint64_t timeout_ms = 0;                    // -t / -timeout
  uint64_t or_nodes = 0;                      // -or / -p
  int subproblems_power = -1;                 // -sub
  uint64_t subproblems_factor = 300;          // -subfactor
  uint64_t stack_kb = 0;                      // -stack (accepted, meaningless here: no device stack frames)
  Arch arch = Arch::BAREBONES;                // GPU build default (config.hpp:84-90)
  Fixpoint fixpoint = Fixpoint::AUTO;         // the engine's event-driven WAC1 (WAC1 sweeps on very small networks): same fixpoint at every node, hence the same tree, as the
                                              // reference's GPU default `-fp wac1` (config.hpp:91-97), which stays available like `-fp ac1`
  uint64_t wac1_threshold = 0;
  uint64_t seed = 0;
  std::string eps_var_order = "default", eps_value_order = "default";
  std::string problem_path, version, hardware;
  // extensions of this engine (not reference flags)
  int gpus = 1;                               // -gpus N: shard the EPS index space over N devices of the node
  std::vector<int> devices;                   // -devices a,b,..: the HIP device of each rank (default 0..N-1; an ordinal may repeat)
  bool deterministic = false;                 // -deterministic: canonical (DFS-first) optimal solution
  int threads_per_block = 0;                  // -threads
};

const char* name_of(Arch a);
const char* name_of(Fixpoint f);

// Prints the usage text and exits with EXIT_FAILURE (config.cpp:11-44).
[[noreturn]] void usage_and_exit(const std::string& program);

// Parses argv; exits like the reference on malformed input.
Options parse_options(int argc, char** argv);

// The echo printed as `%%%mzn-stat: command_line="..."` (config.hpp:168-207).
std::string command_line_echo(const Options& o, const char* program);

// The configuration block of the statistics (config.hpp:237-266).
void print_config_statistics(const Options& o);

}  // namespace turbo_host
