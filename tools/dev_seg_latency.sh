#!/bin/bash
# single-window wall latency vs segment sizes (run on the GPU box)
for N in 1000000 100000; do
for SG in 8192 4096 2048 1024; do for SS in 4096 2048 1024 512; do
  echo -n "seg_gather=$SG seg_splat=$SS  "
  EINCM_SEG=$SG EINCM_SEG_SPLAT=$SS python3 tools/dev_trace_single.py $N 1
done; done; done
