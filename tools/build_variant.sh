#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags]: another build of the library for tools/ab.sh -> tools/_build/ab_<name>.so (never the
# package directory).  Built with -DFR_AB: the experiment rig (FR_DEBUG_MODE / FR_GV / FR_VC / FR_GROUPS / FR_TILE_PRIO and the kernel
# generations only those switches select); add -UFR_AB for a variant of the plain product code.
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$root/tools/_build"
cd "$root/fisher-nerf-customized_amd/csrc"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -DFR_AB "$@" -o "$root/tools/_build/ab_$name.so" fisher_rast.hip fisher_occ.hip 2>&1 | grep -E "error" ; ls -la "$root/tools/_build/ab_$name.so"
