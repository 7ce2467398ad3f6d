// Translation unit 9 of the engine's kernels (kernel_units.hpp says which instantiations it holds).
#define TB_UNIT 9
#include "kernel_units.inc"
