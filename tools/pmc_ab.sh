#!/bin/bash
# usage: tools/pmc_ab.sh <lib.so> <tag> <counters...>: one PMC pass over the bench workload (8 windows x 1e6 events) with the given
# library build; prints per-kernel averages.  Needs /tmp/dev_ab_wins.npz (written by tools/dev_ab.py).
LIB=$1; TAG=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cat > /tmp/pmc_child.py <<PY
import importlib, sys, os
sys.path.insert(0, '$GRAFT_REPO_ROOT')
import numpy as np
L = importlib.import_module('edge-informed-contrast-maximization_amd._lib')
L.LIB_PATH = os.path.abspath('$LIB')
engine = importlib.import_module('edge-informed-contrast-maximization_amd.engine')
z = np.load('/tmp/dev_ab_wins.npz'); B = 8
wins = [{k: z[f'{k}{b}'] for k in ('xs', 'ys', 'ts', 'edges', 'edge_ts', 'th')} for b in range(B)]
th = np.stack([w['th'] for w in wins])
p = engine.make_params(20., 35., 0., 0., 4)
with engine.Engine((260, 346), B * 1000000, max_refs=5, max_windows=B) as e:
    e.set_windows([(w['xs'], w['ys'], w['ts'], w['edges'], w['edge_ts']) for w in wins])
    for k in range(6):
        e.loss_grad(th * (1.0 + 0.01 * k), p)
PY
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 /tmp/pmc_child.py > $OUT/out.txt 2> $OUT/err.txt
echo "rc=$?"
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0].replace('eincm::','').replace('void ','')
        if n.startswith(('k_splat','k_gather','k_imgrad','k_stats_stream','k_final','k_theta')): acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in sorted(acc.items()):
    print('$TAG', k, {c: round(sum(v[1:])/max(len(v)-1,1)) for c,v in d.items()})
PY
find $OUT -name "*.csv" -size +4M -delete
