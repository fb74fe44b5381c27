#!/bin/bash
# all combinations of tools/graph_thread_stress.py, one process each
cd "$(dirname "$0")/.."
for mode in none loader pageable pinned alloc kernel; do for ms in legacy own; do for g in graph eager; do
  timeout -k 10 120 python tools/graph_thread_stress.py $mode $ms $g 2>&1 | grep RESULT || echo "FAILED $mode $ms $g"
done; done; done
