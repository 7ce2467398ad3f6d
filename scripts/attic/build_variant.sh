#!/bin/bash
# Build the engine from a snapshot of the CURRENT kernel sources into turbo_amd/lib/ab/<name>.so (same flags as `make hip`), for same-box A/B runs
# (scripts/pmc_ab.sh ab/<a>.so ab/<b>.so): scripts/build_variant.sh <name> [extra hipcc flags]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
d=build/$name/hip
rm -rf build/$name; mkdir -p $d turbo_amd/lib/ab
cp turbo_amd/csrc/hip/*.hpp turbo_amd/csrc/hip/engine.hip $d/
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-parameter -Wno-bitwise-instead-of-logical -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" -shared -o turbo_amd/lib/ab/$name.so $d/engine.hip
echo "built turbo_amd/lib/ab/$name.so"
