#!/bin/bash
# interleaved A/B of one fc_bf16s knob on ONE box (isolated forward, N = 6400):  tools/gpu_ab_fc.sh VAR v1 v2
var=${1:-RELA_FC_XCD_MAP}; a=${2:-0}; b=${3:-1}
for r in 1 2 3; do
  for d in $a $b; do env $var=$d TAG=$var$d PRECISION=bf16x2 ITERS=40 python tools/time_forward.py; done
done
