"""Per-kernel PMC counter averages from a rocprofv3 --pmc run (rocpd sqlite).
    python tools/pmc_stats.py <results.db> [kernel-substring]"""
import sqlite3
import sys


def main(path, sub=''):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(counters_collection)')]
    # expected columns: ... kernel_name / name, counter_name, value
    kn = 'kernel_name' if 'kernel_name' in cols else 'name'
    rows = cur.execute(f'select {kn}, counter_name, count(*), avg(value), sum(value) from counters_collection group by {kn}, counter_name').fetchall()
    for k, c, n, a, s in sorted(rows):
        if sub in k:
            print(f'{k[:60]:60s} {c:32s} n={n:5d} avg={a:16.1f}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
