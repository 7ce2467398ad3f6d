// The kernels of the engine are ~90 instantiations of three templates (kernels.hpp); compiled in ONE translation unit they take 4-6 minutes of a single
// core.  They are dealt to nine translation units instead (unit_N.hip: `#define TB_UNIT N` + kernel_units.inc), each of which instantiates the kernels it
// dispatches and nothing else, so `make -j8` builds the library in the time of its slowest unit.  engine.hip (the host shim) routes a kernel selection
// {solve / batch propagation, workgroup width, event / sweeps, layout flag} to its unit through the three functions below and instantiates no kernel itself.
// The software bounds build (-DTB_BOUNDS: `__device__` report variables shared by host and kernels) and the per-phase libraries stay single translation units
// (-DTB_SINGLE_TU: engine.hip includes kernel_units.inc with every unit enabled).
#pragma once

#include "kernels.hpp"

namespace tb {

struct KernelSel { int mem, tmax; bool event; int opt; };           // tb_mem_kind, launch bound (128 / 256 / 1024), event-driven fixpoint, fourth template flag
struct KernelGrid { int blocks, threads, shared_bytes; hipStream_t stream; };

constexpr int HOT_EVENT_OPT = 3;    // kernel_opt of a search on the hot tier (event kernel: the layout itself)
constexpr int HOT_SWEEP_OPT = 6;    // ... sweeps: layout 3 << 1
constexpr int TEAM_SWEEP_OPT = 10;  // kernel_opt of a plan that searches in workgroup teams (sweeps, layout 5 << 1)
constexpr int TEAM_EVENT_OPT = 5;   // ... with the event-driven fixpoint (the layout itself; r06, TB_TEAM_EVENT=1)

// 1-5: the search kernel (1: 128-thread event, 2: 256-thread event, 3: 1024-thread event + hot tier, 4: 256-thread sweeps, 5: 1024-thread sweeps + hot tier + teams);
// 6-9: batch propagation (6: event <= 256, 7: event 1024, 8: sweeps <= 256, 9: sweeps 1024)
inline int kernel_unit(bool solve, const KernelSel& k) {
  if (solve) {
    if ((k.opt == HOT_EVENT_OPT || k.opt == TEAM_EVENT_OPT) && k.event) return 3;
    if ((k.opt == HOT_SWEEP_OPT || k.opt == TEAM_SWEEP_OPT) && !k.event) return 5;
    if (k.tmax == 128) return 1;
    if (k.tmax <= 256) return k.event ? 2 : 4;
    return k.event ? 3 : 5;
  }
  if (k.tmax <= 256) return k.event ? 6 : 8;
  return k.event ? 7 : 9;
}

// hipFuncSetAttribute(max dynamic LDS) + occupancy of the selected kernel; returns a hipError_t as int (0 = success)
#define TB_UNIT_DECL(N)                                                                                                                           \
  int unit##N##_prepare(const KernelSel& k, int bytes, int threads, int* max_blocks_per_cu);                                                     \
  int unit##N##_launch_solve(const KernelSel& k, const KernelGrid& g, const DevProblem& P, const DevProblem* dP, Mailbox* mbox);                 \
  int unit##N##_launch_prop(const KernelSel& k, const KernelGrid& g, const DevProblem& P, int2* stores, PropagateOut* out, int n_stores);
TB_UNIT_DECL(1) TB_UNIT_DECL(2) TB_UNIT_DECL(3) TB_UNIT_DECL(4) TB_UNIT_DECL(5) TB_UNIT_DECL(6) TB_UNIT_DECL(7) TB_UNIT_DECL(8) TB_UNIT_DECL(9)
#undef TB_UNIT_DECL

}  // namespace tb
