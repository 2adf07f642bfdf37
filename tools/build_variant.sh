#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags]: another build of the library for tools/ab.sh -> fisher_rast/ab_<name>.so
name=$1; shift
cd "$(dirname "$0")/../fisher-nerf-customized_amd/csrc"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize "$@" -o ../fisher_rast/ab_$name.so fisher_rast.hip fisher_occ.hip 2>&1 | grep -E "error" ; ls -la ../fisher_rast/ab_$name.so
