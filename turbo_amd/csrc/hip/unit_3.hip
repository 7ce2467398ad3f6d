// Translation unit 3 of the engine's kernels (kernel_units.hpp says which instantiations it holds).
#define TB_UNIT 3
#include "kernel_units.inc"
