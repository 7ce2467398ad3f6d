#!/bin/bash
# static census of the headline kernel alone (seconds instead of the whole engine): scripts/isa_quick.sh [extra hipcc flags]
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-atomic-optimizer-strategy=None -gline-tables-only -S --cuda-device-only -Iinclude "$@" -o build/h.s build/headline_only.hip 2>&1 | grep -v "hip-link" | head -20
python3 scripts/isa_stats.py build/h.s solve_kernel
