#!/bin/bash
# interleaved A/B of fc_bf16s weight-prefetch depth on ONE box (isolated forward, N = 6400)
for r in 1 2 3; do
  for d in 1 2; do RELA_FC_BDEPTH=$d TAG=bdepth$d PRECISION=bf16x2 ITERS=40 python tools/time_forward.py; done
done
