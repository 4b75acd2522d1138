#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + stats, then HBM PMC counters in their own passes.
# Usage: tools/profile_round.sh <tag>      outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency"
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
echo "pmc write rc=$?"
rocprofv3 --pmc TCC_EA0_ATOMIC_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_misc -- python3 $ARGS > $OUT/bench_pmc_misc.json 2> $OUT/pmc_misc.err
echo "pmc misc rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_valu -- python3 $ARGS > $OUT/bench_pmc_valu.json 2> $OUT/pmc_valu.err
echo "pmc valu rc=$?"
find $OUT -name "*.csv" | head -40
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merge small: drop the big per-dispatch CSVs except stats
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
