// Translation unit 2 of the engine's kernels (kernel_units.hpp says which instantiations it holds).
#define TB_UNIT 2
#include "kernel_units.inc"
