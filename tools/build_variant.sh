#!/bin/bash
# an experimental build of the device library next to the product's: tools/build_variant.sh <name> <hipcc flags...> -> smartdenovo_amd/variants/libwtzmo_hip_<name>.so
# (git-ignored, travels to the GPU box with the snapshot; a GPU script swaps it in for one measurement: tools/with_variant.sh)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); N=$1; shift
mkdir -p $R/smartdenovo_amd/variants
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off "$@" -I$R/include -shared -fPIC -o $R/smartdenovo_amd/variants/libwtzmo_hip_$N.so $R/smartdenovo_amd/csrc/wtz_lib.cpp
