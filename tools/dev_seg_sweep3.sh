#!/bin/bash
# longer single-chunk splat segments (coarser fixed point) and longer gather segments, balanced splitting (run on the GPU box)
for SG in 8192 16384; do for SS in 4096 8192 16384; do
  echo -n "seg_gather=$SG seg_splat=chunk=$SS  "
  EINCM_SEG=$SG EINCM_SEG_SPLAT=$SS EINCM_CHUNK=$SS python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('%.4g ev/s %.4f ms/step splat(bench) %.1f us single %.4f ms | stage splat %.1f gather %.1f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']*1e3, d.get('eval_ms_single_window') or 0, s['splat']*1e3, s['gather']*1e3))"
done; done
