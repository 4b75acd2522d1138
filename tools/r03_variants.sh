#!/bin/bash
# Kernel durations of one evaluation (tools/dev_trace_single.py under rocprofv3 --kernel-trace) for variant builds of the library.
# usage (GPU box): tools/r03_variants.sh "<N> <h>" base name1 name2 ...     (names: tools/variants/libeincm_<name>.so; base = the product)
set -u
CFG=$1; shift
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/r03/var_$v
  rm -rf $OUT; mkdir -p $OUT
  if [ $v = base ]; then unset EINCM_LIB; else export EINCM_LIB=$GRAFT_REPO_ROOT/tools/variants/libeincm_$v.so; fi
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/dev_trace_single.py $CFG > $OUT/run.txt 2>&1
  echo "== $v ($CFG)"; grep wall $OUT/run.txt; python3 tools/dev_trace_gaps.py $OUT/trace 2>&1 | grep -v "^100"
  rm -rf $OUT/trace
done
