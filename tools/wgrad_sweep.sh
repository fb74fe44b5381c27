#!/bin/bash
# times the wgrad shapes that dominate the DEAL-YOLO-N step (batch 64)
for cfg in "64 64 3 1 160 160" "32 64 3 1 160 160" "32 32 3 1 160 160" "16 16 3 1 160 160" "48 32 1 1 160 160" "64 64 1 1 160 160" "64 64 3 1 80 80" "8 16 3 2 640 640" "16 32 3 2 320 320" "64 128 3 2 80 80" "128 128 3 1 40 40"; do
  python tools/conv_bench.py wgrad $cfg 2>/dev/null | tail -1
done
