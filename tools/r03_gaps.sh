#!/bin/bash
# Round 3: where does a single-window evaluation spend its wall time?  Kernel traces (start / end per dispatch) of
# tools/dev_trace_single.py at the two latency targets; per-kernel durations and the idle gaps between the kernels of one evaluation.
# usage (GPU box): tools/r03_gaps.sh <tag>
set -u
TAG=${1:-base}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03/gaps_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 __graft_entry__.py > $OUT/build.log 2>&1
for cfg in "1000000 1" "1000000 16" "30000 16" "30000 1"; do
  set -- $cfg
  name=N$1_h$2
  python3 tools/dev_trace_single.py $1 $2 > $OUT/wall_$name.txt 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$name -- python3 tools/dev_trace_single.py $1 $2 > $OUT/traced_$name.txt 2>&1
  python3 tools/dev_trace_gaps.py $OUT/trace_$name > $OUT/gaps_$name.txt 2>&1
  find $OUT/trace_$name -name "*.csv" -size +2M -delete
  echo "== $name"; cat $OUT/wall_$name.txt; cat $OUT/gaps_$name.txt
done
