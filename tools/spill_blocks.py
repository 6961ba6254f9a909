#!/usr/bin/env python3
"""Per-basic-block instruction / spill statistics of one kernel in a hipcc -S listing.
usage: tools/spill_blocks.py listing.s kernel_name_substring [min_instructions]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 50
start = [i for i, l in enumerate(lines) if key in l and re.match(r'^_Z\w+:', l)][0]
end = [i for i, l in enumerate(lines[start:]) if l.startswith('.Lfunc_end')][0] + start
cur = {'name': 'entry', 'n': 0, 'st': 0, 'ld': 0, 'accw': 0, 'accr': 0, 'valu': 0, 'ds': 0}
stats = []
for l in lines[start:end]:
    t = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', t):
        stats.append(cur)
        cur = {'name': t.split(':')[0], 'n': 0, 'st': 0, 'ld': 0, 'accw': 0, 'accr': 0, 'valu': 0, 'ds': 0}
        continue
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    cur['n'] += 1
    op = t.split()[0]
    if op.startswith('scratch_store'): cur['st'] += 1
    elif op.startswith('scratch_load'): cur['ld'] += 1
    elif op.startswith('v_accvgpr_write'): cur['accw'] += 1
    elif op.startswith('v_accvgpr_read'): cur['accr'] += 1
    elif op.startswith('v_'): cur['valu'] += 1
    elif op.startswith('ds_'): cur['ds'] += 1
stats.append(cur)
tot = {k: sum(s[k] for s in stats) for k in ('n', 'st', 'ld', 'accw', 'accr', 'valu', 'ds')}
for s in stats:
    if s['n'] >= minn:
        print(s)
print('total', tot)
