#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace database: python tools/rocpd_gaps.py db [skip]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 400
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
rows = list(c.execute(f'select start, end from {disp} order by start'))[skip:]
busy = sum(e - s for s, e in rows)
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
small = [g for g in gaps if g < 50000]
print(f'{len(rows)} kernels: busy {busy / 1e3:.1f} us, span {(rows[-1][1] - rows[0][0]) / 1e3:.1f} us, '
      f'gaps < 50 us: n={len(small)} total {sum(small) / 1e3:.1f} us avg {sum(small) / max(1, len(small)) / 1e3:.2f} us; '
      f'larger gaps: {len(gaps) - len(small)} total {sum(g for g in gaps if g >= 50000) / 1e3:.1f} us')
