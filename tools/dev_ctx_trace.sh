#!/bin/bash
# usage: tools/dev_ctx_trace.sh <lib.so> <tag>: kernel trace of tools/dev_ctx.py (multi-context) for a library build
LIB=$1; TAG=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/ctxtrace_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/dev_ctx.py $LIB > $OUT/out.txt 2> $OUT/err.txt
cat $OUT/out.txt
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:8]:
        print('$TAG', r['Name'][:40], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
find $OUT -name "*.csv" -size +3M -delete
