for cfg in "64 128 3 2 80 80" "128 64 3 1 40 40" "128 128 3 1 40 40" "128 256 3 2 40 40" "256 256 3 1 20 20" "192 128 1 1 40 40" "384 256 1 1 20 20" "256 128 1 1 20 20"; do
  python tools/conv_bench.py fwd $cfg 2>/dev/null | tail -1
done
