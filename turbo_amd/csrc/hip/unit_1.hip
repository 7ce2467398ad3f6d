// Translation unit 1 of the engine's kernels (kernel_units.hpp says which instantiations it holds).
#define TB_UNIT 1
#include "kernel_units.inc"
