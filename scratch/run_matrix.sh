for mode in loader pageable pinned alloc kernel; do for ms in legacy own; do for g in graph eager; do
timeout -k 10 120 python scratch/stress_matrix.py $mode $ms $g 2>&1 | grep RESULT || echo "FAILED $mode $ms $g"
done; done; done
