"""Summarise a rocprofv3 kernel trace of dev_trace_single.py: per-kernel mean duration and the idle gap in front of each
kernel inside one evaluation (last 100 evaluations).  usage: python3 tools/dev_trace_gaps.py <dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].split('::')[-1]) for r in csv.DictReader(open(f))]
rows.sort()
# an evaluation starts at k_theta*, or at a k_splat that no k_theta* precedes (2-DoF theta: the event kernels derive their windows themselves)
evals, cur = [], []
for s, e, n in rows:
    first = n.startswith('k_theta') or (n.startswith('k_splat') and not (cur and cur[-1][2].startswith('k_theta')))
    if first and cur:
        evals.append(cur); cur = []
    cur.append((s, e, n))
evals.append(cur)
evals = [ev for ev in evals if ev[0][2].startswith(('k_theta', 'k_splat'))]
nk = max(set(len(ev) for ev in evals), key=[len(ev) for ev in evals].count)       # the usual number of kernels per evaluation
evals = [ev for ev in evals if len(ev) == nk][-100:]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
span = []
for ev in evals:
    span.append(ev[-1][1] - ev[0][0])
    for i, (s, e, n) in enumerate(ev):
        dur[n].append(e - s)
        if i: gap[n].append(s - ev[i - 1][1])
print(f'{len(evals)} evaluations; GPU span first-kernel-start..last-kernel-end: mean {sum(span)/len(span)/1e3:.1f} us')
order = [n for _, _, n in evals[-1]]
for n in order:
    g = gap.get(n, [0])
    print(f'  {n:28s} dur {sum(dur[n])/len(dur[n])/1e3:7.1f} us   gap before {sum(g)/len(g)/1e3:6.1f} us')
print(f'  sum of durations {sum(sum(dur[n])/len(dur[n]) for n in order)/1e3:.1f} us, sum of gaps {sum(sum(gap[n])/len(gap[n]) for n in order if n in gap)/1e3:.1f} us')
