// Translation unit 8 of the engine's kernels (kernel_units.hpp says which instantiations it holds).
#define TB_UNIT 8
#include "kernel_units.inc"
