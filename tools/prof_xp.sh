#!/bin/bash
# headline line + per-kernel stats of the same command (kernel trace only)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python bench.py --steps 20 --warmup 4 --no-extras --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/xp_bench.json
cut -c1-400 gpurun_out/xp_bench.json
rm -rf gpurun_out/xp_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/xp_prof -o xp -- python3 bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/xp_prof.log 2>&1
f=$(find gpurun_out/xp_prof -name '*kernel_stats.csv' | head -1)
head -25 "$f" | cut -c1-220
cp "$f" gpurun_out/xp_kernel_stats.csv
find gpurun_out/xp_prof -name '*.db' -delete; find gpurun_out/xp_prof -name '*trace.csv' -delete
