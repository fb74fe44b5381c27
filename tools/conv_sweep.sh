#!/bin/bash
# usage: tools/conv_sweep.sh  -- times the conv shapes that dominate the DEAL-YOLO-N step (one launch each, batch 64)
for cfg in "64 64 3 1 160 160" "32 32 3 1 160 160" "16 16 3 1 160 160" "64 32 3 1 160 160" "64 64 1 1 160 160" "48 32 1 1 160 160" "128 128 3 1 40 40" "64 64 3 1 80 80" "32 64 3 2 160 160" "16 32 3 2 320 320"; do
  python tools/conv_bench.py fwd $cfg 2>/dev/null | tail -1
done
