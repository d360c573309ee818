#!/bin/bash
# build a named variant of the library with extra -D flags (A/B timing only): tools/build_variant.sh name -DFOO=1 ...
name=$1; shift
mkdir -p gpurun_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include "$@" mobilesuperresolution_amd/csrc/sr_abi.hip -o gpurun_variants/lib_$name.so 2>&1 | grep -E "error|spill" | head
ls -la gpurun_variants/lib_$name.so
