#!/usr/bin/env python3
"""Print the scheduling skeleton (waits, barriers, DMA, LDS reads, MFMAs) of a kernel's largest loop from a hipcc -S dump.
usage: isa_loop.py file.s mangled_name_substring [--all]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and key in l]
for start in starts:
    end = [i for i, l in enumerate(lines[start:]) if l.strip().startswith('s_endpgm')][0] + start
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i: loops.append((labels[m.group(1)], i))
    sel = [a for a in sys.argv if a.startswith('--loop=')]
    loops = sorted(set(loops))
    if sel:
        a, b = loops[int(sel[0][7:])]
    else:
        a, b = max(loops, key=lambda x: x[1] - x[0])
    print('loops:', loops)
    print(lines[start].split(':')[0], 'loop lines', a, b, 'len', b - a)
    run = None; n = 0
    def flush():
        global run, n
        if run: print(f'    {run} x{n}')
        run = None; n = 0
    for l in body[a:b]:
        t = l.strip().split(';')[0].strip()
        if not t or t.startswith('.'): continue
        op = t.split()[0]
        if op.startswith(('s_waitcnt', 's_barrier', 's_setprio')): k = t
        elif op.startswith('global_load_lds') or op.startswith('buffer_load'): k = op
        elif 'mfma' in op: k = 'MFMA'
        elif op.startswith('ds_read'): k = op
        elif op.startswith('ds_write'): k = op
        elif op.startswith('global_load'): k = op
        elif '--all' in sys.argv: k = 'other'
        else: continue
        if k == run: n += 1
        else:
            flush(); run = k; n = 1
    flush()
