#!/bin/bash
# A/B: alternative builds of the library (gpurun_out/libeincm_cap*.so are not shipped: gpurun_out/ is excluded) -> copy via tools/
cd $GRAFT_REPO_ROOT
cp edge-informed-contrast-maximization_amd/libeincm_hip.so /tmp/base.so
for v in base cap2304 cap3072; do
  if [ $v != base ]; then cp tools/variants/libeincm_$v.so edge-informed-contrast-maximization_amd/libeincm_hip.so; touch edge-informed-contrast-maximization_amd/libeincm_hip.so; fi
  for seg in 0 2048 4096; do
    EINCM_SEG=$seg python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency > /tmp/b.json 2>/dev/null
    python -c "
import json; d=json.load(open('/tmp/b.json')); s=d['stage_ms_per_step']; print('$v seg $seg', 'ms/step %.3f'%d['ms_per_step'], 'splat %.4f gather %.4f'%(s['splat'],s['gather']))"
  done
done
cp /tmp/base.so edge-informed-contrast-maximization_amd/libeincm_hip.so
