#!/bin/bash
# build a variant of librydemu.so with extra -D switches: tools/build_variant.sh NAME [-DSPLITR_...=..] -> build/variants/NAME.so
ROOT=$(cd "$(dirname "$0")/.." && pwd); mkdir -p $ROOT/build/variants
n=$1; shift
cd $ROOT/pulser_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value "$@" rydemu.hip -o $ROOT/build/variants/$n.so
