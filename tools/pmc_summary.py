#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database (--pmc run)."""
import sqlite3
import sys


def main(db, pattern=''):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    pmc = [t for t in tabs if 'pmc_event' in t][0]
    info = [t for t in tabs if t.startswith('rocpd_info_pmc')][0]
    disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    q = (f'select s.kernel_name, i.name, count(*), avg(e.value) from {pmc} e join {info} i on e.pmc_id = i.id '
         f'join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id '
         f'group by s.kernel_name, i.name order by s.kernel_name, i.name')
    for name, cn, n, avg in c.execute(q):
        if pattern in name:
            print(f'{name[:70]:70s} {cn:32s} n={n:4d} avg={avg:16.1f}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
