#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline object refers to (run on the GPU box from the repo root):
#   1. kernel-trace + stats of the default bench command          -> gpurun_out/prof_stats
#   2. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, un-captured launches so that every dispatch is visible)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o st --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/prof_stats.log 2>&1
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o f --output-format csv -- python bench.py --steps 2 --warmup 1 --graph 0 --no-cpu --probe 0 > gpurun_out/prof_fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_write -o w --output-format csv -- python bench.py --steps 2 --warmup 1 --graph 0 --no-cpu --probe 0 > gpurun_out/prof_write.log 2>&1
echo write done
tail -1 gpurun_out/prof_stats.log
