#!/bin/bash
# kernel trace of the 16x16-theta evaluations (dev tool): DEV_CFG row(s) of tools/dev_stages.py under rocprofv3 --kernel-trace --stats
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_tile
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export DEV_CFG=${DEV_CFG:-3}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/dev_stages.py > $OUT/run.log 2> $OUT/run.err
python3 - <<'PY'
import csv, glob, os
f = sorted(glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_tile/trace/*/*_kernel_stats.csv'), key=os.path.getmtime)[-1]
for r in list(csv.DictReader(open(f)))[:14]:
    print('%-60s calls %6s avg_us %9.1f pct %s' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
