#!/bin/bash
# A/B timing of bench.py under environment switches, alternating on ONE box (run from the repo root on the GPU box):
#   tools/ab_bench.sh <rounds> "<ENV=..  ENV=..>" "<ENV=..>" ...     e.g.  tools/ab_bench.sh 2 "DY_BN_ACC=0" "DY_BN_ACC=1"
# prints one line per run: the variant, images/s and ms/step (bench.py --steps 30 --warmup 8 --no-cpu --probe 0)
rounds=$1; shift
for r in $(seq 1 "$rounds"); do
  for v in "$@"; do
    line=$(env $v python bench.py --steps ${STEPS:-30} --warmup 8 --no-cpu --probe 0 ${BENCH_ARGS:-} 2>/dev/null | grep '^{' | tail -1)
    echo "[$r] $v  $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(f\"{d['value']:.1f} img/s  {d['ms_per_step']:.3f} ms\")" "$line")"
  done
done
