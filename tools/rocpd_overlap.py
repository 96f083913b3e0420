#!/usr/bin/env python3
"""How the kernels of a rocprofv3 --kernel-trace database share the device in time: python tools/rocpd_overlap.py db [skip]
For the dispatches after the first `skip`: the span, the time with 0 / 1 / 2+ kernels running, and per kernel name the time it
runs alone and next to others (two-lane execution: which phases of a forward still have the device to themselves)."""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 400
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
syms = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
names = dict(c.execute(f'select id, kernel_name from {syms}'))
rows = list(c.execute(f'select start, end, kernel_id from {disp} order by start'))[skip:]
ev = []
for i, (s, e, k) in enumerate(rows):
    ev.append((s, 1, i))
    ev.append((e, -1, i))
ev.sort()
running = set()
t_prev = ev[0][0]
by_level = collections.Counter()
alone = collections.Counter()
shared = collections.Counter()
for t, d, i in ev:
    dt = t - t_prev
    if dt > 0:
        by_level[min(len(running), 3)] += dt
        for j in running:
            (alone if len(running) == 1 else shared)[names[rows[j][2]].split('(')[0][:60]] += dt
    t_prev = t
    if d > 0:
        running.add(i)
    else:
        running.discard(i)
span = ev[-1][0] - ev[0][0]
print(f'{len(rows)} kernels, span {span / 1e3:.1f} us: idle {by_level[0] / span:.3f}, one kernel {by_level[1] / span:.3f}, '
      f'two {by_level[2] / span:.3f}, three or more {by_level[3] / span:.3f}')
for k in sorted(set(alone) | set(shared), key=lambda k: -(alone[k] + shared[k])):
    print(f'  {k:60s} alone {alone[k] / 1e3:9.1f} us  next to others {shared[k] / 1e3:9.1f} us')
