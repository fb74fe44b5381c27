#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline object refers to (run on the GPU box from the repo root):
#   1. kernel-trace + stats of the default bench command          -> gpurun_out/prof_stats
#   2. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, un-captured launches so that every dispatch is visible)
#   3. PMC passes for the matrix-pipe and LDS evidence: SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES + GRBM_GUI_ACTIVE,
#      SQ_LDS_BANK_CONFLICT + SQ_LDS_IDX_ACTIVE  (counters only with --kernel-trace: gpurun refuses other trace domains with --pmc)
# usage: tools/collect_profiles.sh [extra bench.py arguments, e.g. --loss wiou+nwd]
set -e
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0  # the profiler may initialise HIP before Python can set it (ultralytics/hip/__init__.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_mfma gpurun_out/prof_lds
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o st --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu --secondary 0 "$@" > gpurun_out/prof_stats.log 2>&1
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o f --output-format csv -- python bench.py --steps 2 --warmup 1 --graph 0 --no-cpu --probe 0 --secondary 0 "$@" > gpurun_out/prof_fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_write -o w --output-format csv -- python bench.py --steps 2 --warmup 1 --graph 0 --no-cpu --probe 0 --secondary 0 "$@" > gpurun_out/prof_write.log 2>&1
echo write done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/prof_mfma -o m --output-format csv -- python bench.py --steps 2 --warmup 1 --graph 0 --no-cpu --probe 0 --secondary 0 "$@" > gpurun_out/prof_mfma.log 2>&1
echo mfma done
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/prof_lds -o l --output-format csv -- python bench.py --steps 2 --warmup 1 --graph 0 --no-cpu --probe 0 --secondary 0 "$@" > gpurun_out/prof_lds.log 2>&1
echo lds done
tail -1 gpurun_out/prof_stats.log
# BASELINE.json configs[3] and configs[4]: kernel stats of the LD model's training bench and of get_FPS.py on yolov8n-p2 at 1280x1280,
# batch 32 (the reference's protocol shortened to 20 + 100 forwards)
if [ "${DY_PROFILE_SECONDARY:-1}" = "1" ]; then
  rm -rf gpurun_out/prof_ld gpurun_out/prof_p2
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ld -o st --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu --probe 0 --secondary 0 --model yolov8n-LD-P2 > gpurun_out/prof_ld.log 2>&1
  echo ld done
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_p2 -o st --output-format csv -- python get_FPS.py --weights yolov8n-p2.yaml --batch 32 --imgs 1280 1280 --warmup 20 --testtime 100 > gpurun_out/prof_p2.log 2>&1
  echo p2 done
  tail -1 gpurun_out/prof_ld.log | cut -c1-160
  tail -1 gpurun_out/prof_p2.log
fi
