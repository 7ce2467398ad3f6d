// Translation unit 4 of the engine's kernels (kernel_units.hpp says which instantiations it holds).
#define TB_UNIT 4
#include "kernel_units.inc"
