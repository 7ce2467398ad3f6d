// Translation unit 6 of the engine's kernels (kernel_units.hpp says which instantiations it holds).
#define TB_UNIT 6
#include "kernel_units.inc"
