#!/usr/bin/env python3
"""Instruction mix of the two event kernels' inner loops from the gfx950 ISA (hipcc -save-temps).
usage: python tools/isa_mix.py <file.s> > profiles/rNN/isa_instruction_mix.md

For each kernel the event loop is located (the innermost loop that contains v_exp_f32), its basic blocks are listed, and the
blocks of the common case - every tap inside the LDS window - are summed per instruction class.  One trip of the loop handles
one event per lane (k_splat's loop is unrolled x3 by the source; the per-event figure divides by the events per trip)."""
import re
import sys
from collections import Counter, OrderedDict

CLASSES = OrderedDict([
    ('fp64 (add/mul/fma/rndne/cvt to or from f64)', re.compile(r'^v_(add|mul|fma|rndne|fract|min|max)_f64|^v_cvt_(f64_|[a-z0-9]+_f64)')),
    ('transcendental (exp/rcp/rsq/log/sqrt)', re.compile(r'^v_(exp|rcp|rsq|log|sqrt)_f32')),
    ('packed fp32 (v_pk_*)', re.compile(r'^v_pk_')),
    ('fp32 arithmetic', re.compile(r'^v_(add|sub|subrev|mul|fma|fmac|fmaak|fmamk|mac|mad|max|min|med3|rndne|fract)_f32|^v_(max3|min3)_f32')),
    ('convert (fp32 <-> int, no f64)', re.compile(r'^v_cvt_')),
    ('integer / address / select / move', re.compile(r'^v_(add|sub|subrev|mul|mad|lshl|lshr|ashr|and|or|xor|bfe|bfi|cndmask|mov|add3|lshl_add|lshl_or|and_or|or3|mad_u|mul_i|mul_u|min|max|med3|cmp|cmpx|readfirstlane|readlane|writelane|perm|alignbit|mbcnt|accvgpr)')),
    ('LDS (ds_*)', re.compile(r'^ds_')),
    ('global / flat memory', re.compile(r'^(global|flat|buffer|scratch)_')),
    ('scalar ALU / control (s_*)', re.compile(r'^s_')),
])


def classify(op):
    for name, rx in CLASSES.items():
        if rx.match(op):
            return name
    return 'other: ' + op


def kernels(text):
    for m in re.finditer(r'^(_ZN5eincm[^\n:]*):[^\n]*\n(.*?)\n\s*s_endpgm', text, re.S | re.M):
        yield m.group(1), m.group(2)


def blocks(body):
    cur, name, out = [], 'entry', []
    for line in body.splitlines():
        line = line.strip()
        m = re.match(r'^(\.LBB\d+_\d+):', line)
        if m:
            out.append((name, cur)); name, cur = m.group(1), []
            continue
        if not line or line.startswith(';') or line.startswith('.') or line.startswith('//'):
            continue
        op = line.split()[0]
        cur.append(op)
        if op.startswith('s_cbranch') or op == 's_branch':          # a branch ends the basic block: what follows is the fall-through path
            out.append((name, cur)); name, cur = name + '+', []
    out.append((name, cur))
    return out


def main():
    text = open(sys.argv[1]).read()
    want = [('k_splat<THETA_CONST, 0>  (2-DoF theta, the bench configuration)', 'k_splatILi1ELi0E', 'ds_add_u32', 3),
            ('k_gather<THETA_CONST, 0>  (2-DoF theta)', 'k_gatherILi1ELi0E', 'ds_read_b32', 1),
            ('k_splat<THETA_TILE, 0>  (pyramid levels >= 1, dense)', 'k_splatILi2ELi0E', 'ds_add_u32', 3),
            ('k_gather<THETA_TILE, 0>', 'k_gatherILi2ELi0E', 'ds_read_b32', 1)]
    print('# Instruction mix of the event kernels\' inner loops (gfx950 ISA, hipcc -O3, `tools/isa_mix.py`)\n')
    print('Counted from the disassembly: the basic blocks of the event loop that run in the common case (every tap of the event inside the LDS\n'
          'window).  "per event" = per lane-event and reference time, i.e. per warped event.  The out-of-window path (taps sent straight to HBM with\n'
          'the JAX wrap/drop rule) and the per-workgroup prologue / epilogue are excluded; they are in the PMC totals (`SQ_INSTS_VALU`).\n')
    for title, key, marker, events_per_trip in want:
        body = next((b for n, b in kernels(text) if key in n), None)
        if body is None:
            print(f'## {title}\n\nnot found\n'); continue
        bl = blocks(body)

        def has(ops, prefix, n):
            return sum(o.startswith(prefix) for o in ops) >= n
        # The source unrolls the event loop x3 (k_splat: renamed-register pipeline; k_gather: #pragma unroll 2 + remainder), so the
        # common-case blocks come in repeating groups; the MIDDLE group is the steady state.
        #   k_splat : [loads + fp64 warp + tap math (4 v_exp_f32)] [9 products + 9 ds_add_u32]
        #   k_gather: [loads + fp64 warp (>= 3 global_load)] [window reads (>= 6 ds_read)] [tap math + combination (3 v_exp_f32 + 2 v_rcp_f32)]
        if key.startswith('k_splat'):
            groups, cur = [], []
            for n, ops in bl:
                if has(ops, 'v_exp_f32', 4):
                    cur = [(n, ops)]
                elif cur and has(ops, 'ds_add_u32', 9):
                    groups.append(cur + [(n, ops)]); cur = []
        else:
            # k_gather's loop is not unrolled: [event loads + fp64 warp + window test] [9 window reads] [tap math + combination]
            # (+ for 2-DoF theta the fp32 accumulation, for the tile form the two i64 conversions and ds_add_u64)
            groups, cur, stage = [], [], 0
            for n, ops in bl:
                if stage == 0 and has(ops, 'global_load', 2) and sum('_f64' in o for o in ops) >= 8:
                    cur, stage = [(n, ops)], 1
                elif stage == 1 and has(ops, 'ds_read', 6):
                    cur.append((n, ops)); stage = 2
                elif stage == 2 and has(ops, 'v_exp_f32', 3):
                    cur.append((n, ops)); stage = 3
                elif stage == 3:
                    cur.append((n, ops)); groups.append(cur); cur, stage = [], 0
        fast = groups[len(groups) // 2] if groups else []
        events_per_trip = 1
        tot = Counter()
        for n, ops in fast:
            for o in ops:
                tot[classify(o)] += 1
        nvalu = sum(v for k, v in tot.items() if not k.startswith(('LDS', 'global', 'scalar', 'other')))
        print(f'## {title}\n')
        print(f'fast-path blocks: {", ".join(n for n, _ in fast)}; events per loop trip (unroll): {events_per_trip}\n')
        print('| class | instructions per warped event |')
        print('|---|---|')
        for k in list(CLASSES) + sorted(k for k in tot if k.startswith('other')):
            if tot.get(k):
                print(f'| {k} | {tot[k] / events_per_trip:.1f} |')
        print(f'| **VALU total** | **{nvalu / events_per_trip:.1f}** |\n')


if __name__ == '__main__':
    main()
