#!/bin/bash
# On the GPU box: per-kernel time of the default bench step (rocprofv3 --kernel-trace --stats).
#   tools/gpu_profile.sh <tag> [extra bench.py args]      -> gpurun_out/<tag>_kernel_stats.csv (+ the bench line)
# The program follows `--` directly (python3 itself: no env / bash / shebang hop under the profiler).
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats -d "$out" -o stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --cpu-views 0 "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats.csv && cut -d, -f1-4,8 gpurun_out/${tag}_kernel_stats.csv | head -24
