#!/bin/bash
# batch throughput (bench workload) and single-window latency vs segment sizes (run on the GPU box)
for SG in 8192 4096 2048 1024; do for SS in 4096 2048; do
  echo -n "seg_gather=$SG seg_splat=$SS  bench: "
  EINCM_SEG=$SG EINCM_SEG_SPLAT=$SS python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g ev/s %.4f ms/step splat %.1f GB/s single %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d.get('eval_ms_single_window') or 0))"
done; done
for N in 1000000 100000; do for SG in 8192 4096 2048 1024; do
  echo -n "seg_gather=$SG  "; EINCM_SEG=$SG python3 tools/dev_trace_single.py $N 1
  echo -n "seg_gather=$SG  "; EINCM_SEG=$SG python3 tools/dev_trace_single.py $N 16
done; done
