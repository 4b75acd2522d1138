#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "2048 4096 2304" "4096 4096 2304" "4096 8192 2304" "2048 2048 2304" "4096 4096 3072" "4096 8192 3072" "4096 4096 1792"; do set -- $cfg
  EINCM_CHUNK=$1 EINCM_SEG_SPLAT=$2 EINCM_WINCAP=$3 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-latency > /tmp/b.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/b.json')); s=d['stage_ms_per_step']; print('chunk $1 seg_s $2 cap $3', 'ms/step %.3f'%d['ms_per_step'], 'splat %.4f gather %.4f'%(s['splat'],s['gather']))"
done
