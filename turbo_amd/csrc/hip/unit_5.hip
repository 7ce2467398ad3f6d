// Translation unit 5 of the engine's kernels (kernel_units.hpp says which instantiations it holds).
#define TB_UNIT 5
#include "kernel_units.inc"
