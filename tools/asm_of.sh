#!/bin/bash
# tools/asm_of.sh <mangled-name-prefix> [extra hipcc flags]: gfx950 assembly of one kernel of fisher_rast.hip -> /tmp/kernel.s
name=$1; shift
cd "$(dirname "$0")/../fisher-nerf-customized_amd/csrc"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize "$@" -S --cuda-device-only -o /tmp/all.s fisher_rast.hip 2>/dev/null
a=$(grep -n "^$name" /tmp/all.s | head -1 | cut -d: -f1)
b=$(awk -v a=$a 'NR>a && /^.Lfunc_end/ {print NR; exit}' /tmp/all.s)
sed -n "${a},${b}p" /tmp/all.s > /tmp/kernel.s
grep "$name" /tmp/all.s | grep "num_vgpr\|private_seg_size" | head -3
wc -l /tmp/kernel.s
