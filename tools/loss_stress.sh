#!/bin/bash
# dy_detection_loss alone, re-run from restored state and compared bit for bit with its first result, in two processes, beside a third
# that keeps the GPU busy with full training steps (run on the GPU box from the repo root):
#   tools/loss_stress.sh ["ENV=.. ENV=.."]      e.g.  tools/loss_stress.sh "WIOU=1 NWD=0"      (ITERS / HEAVY: iterations / co-tenant steps)
# round 4: with four correctly rounded divisions in the dual-number quotient 4-7 % of the launches returned another derivative for one
# anchor (WIoU), with one reciprocal 0 of 40,000 (profiles/r04_loss_stress.txt)
B=64 S=640 python tools/loss_stress.py heavy1 ${HEAVY:-1500} > gpurun_out/ls_h.log 2>&1 &
sleep 6
for k in 1 2; do env $1 python tools/loss_stress.py p$k ${ITERS:-12000} > gpurun_out/ls_$k.log 2>&1 & done
wait
echo "$1 :: $(grep "loss executions" gpurun_out/ls_1.log | cut -c1-80) | $(grep "loss executions" gpurun_out/ls_2.log | cut -c1-80)"
