#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd sqlite database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`) into the
plain-text per-kernel summary committed under profiles/ (name, calls, total/avg/min/max us, share)."""
import sqlite3
import sys


def short(name, n=110):
    name = name.replace('(anonymous namespace)::', '')
    return name if len(name) <= n else name[:n - 3] + '...'


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration), '
                     'max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x), max(grid_x) '
                     'from kernels group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f'# source: {db}', '# durations in microseconds (rocprofv3 kernel-trace, ns / 1000)',
             f'{"kernel":110s} {"calls":>6s} {"total_us":>12s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s} {"pct":>6s} '
             f'{"vgpr":>5s} {"agpr":>5s} {"sgpr":>5s} {"lds":>7s} {"wg":>5s}']
    for name, n, tot, avg, mn, mx, vg, ag, sg, lds, wg, grid in rows:
        lines.append(f'{short(name):110s} {n:6d} {tot / 1e3:12.1f} {avg / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} '
                     f'{100.0 * tot / total:6.2f} {vg or 0:5d} {ag or 0:5d} {sg or 0:5d} {lds or 0:7d} {wg or 0:5d}')
    text = '\n'.join(lines) + '\n'
    if out:
        open(out, 'w').write(text)
    else:
        sys.stdout.write(text)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
