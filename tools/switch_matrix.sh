#!/bin/bash
# every switch (rounds 3-4) off / on one at a time (and a few together): the step must trace, capture and run
for e in "DY_X=1" "DY_BN_ACC=0" "DY_BN_WGRAD=0" "DY_SILU_FAST=0" "DY_STEM_DIRECT=0" "DY_BIAS_WGRAD=0" "DY_BN_RES=0" "DY_WGRAD_SPLIT=0" "DY_HEAD_ROWS=0" "DY_HEAD_DECODE=0" "DY_HEAD_APPLY=0" "DY_HEAD_CLS=0" "DY_ADD_ALIAS=0" "DY_SCALSEQ_ADD=0" "DY_BN_DGRED=1 DY_BN_DGRED_MAXC=32" "DY_HEAD_CLS=0 DY_HEAD_APPLY=0" "DY_SCALSEQ_ADD=0 DY_ADD_ALIAS=0" "DY_SIDE_WGRAD=1" "DY_SIDE_WGRAD=1 DY_HEAD_CLS=0 DY_HEAD_APPLY=0" "DY_HEAD_STREAMS=1" "DY_PLANAR=0" "DY_DP_BUCKETS=2" "DY_PLANAR_CV1=0" "DY_PLANAR_CV1=0 DY_SIDE_WGRAD=1" "DY_CONV_FW=0" "DY_CONV_FW=80" "DY_BN_ACC=0 DY_CONV_FW=80" "DY_UPSEG=0" "DY_UPSEG_TRAIN=0" "DY_UPSEG_TRAIN=0 DY_PLANAR_CV1=0" "DY_PLANAR=0 DY_UPSEG_TRAIN=1"; do
  out=$(env $e python bench.py --steps 3 --warmup 2 --batch 8 --imgsz 320 --lr 0.0005 --no-cpu --probe 0 2>&1 | tail -1 | cut -c1-140)
  echo "$e :: $out"
done
