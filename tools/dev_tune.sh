#!/bin/bash
cd $GRAFT_REPO_ROOT
for cap in 1536 2304 3072; do for segs in 2048 4096 8192; do for seg in 8192 16384 32768; do
  EINCM_WINCAP=$cap EINCM_SEG_SPLAT=$segs EINCM_SEG=$seg python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-latency > /tmp/b.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/b.json')); s=d['stage_ms_per_step']; print('cap $cap seg_s $segs seg_g $seg', 'ms/step %.3f'%d['ms_per_step'], 'splat %.4f gather %.4f'%(s['splat'],s['gather']))"
done; done; done
