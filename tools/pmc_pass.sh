#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counter list...>   (one PMC pass over bench.py, summary to stdout)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-latency > $OUT/bench.json 2> $OUT/err.txt
echo "rc=$?"
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0].replace('eincm::','').replace('void ','')
        if n in ('k_splat','k_gather'): acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    print(k, {c: round(sum(v)/len(v)) for c,v in d.items()})
PY
find $OUT -name "*.csv" -size +4M -delete
