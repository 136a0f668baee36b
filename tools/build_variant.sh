#!/bin/bash
# builds metalpathtracer_amd/lib/libmpt_hip_<name>.so with extra compiler flags:  tools/build_variant.sh <name> [-D...]
n=$1; shift
exec /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Imetalpathtracer_amd/csrc \
  -Wno-unused-function -Wno-unused-value -Wno-unused-result -Wno-pass-failed -fno-slp-vectorize "$@" -shared \
  -o metalpathtracer_amd/lib/libmpt_hip_$n.so metalpathtracer_amd/csrc/mpt_hip.hip
