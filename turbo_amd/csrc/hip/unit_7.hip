// Translation unit 7 of the engine's kernels (kernel_units.hpp says which instantiations it holds).
#define TB_UNIT 7
#include "kernel_units.inc"
