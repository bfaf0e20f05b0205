"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace into the familiar --stats table.

    python tools/rocpd_stats.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, top=40):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f'# kernel trace summary of {path}: {sum(r[1] for r in rows)} dispatches, {total/1e6:.3f} ms of kernel time')
    print(f'{"calls":>7} {"total_ms":>10} {"avg_us":>10} {"min_us":>9} {"max_us":>9} {"share":>7}  kernel')
    for n, c, s, a, mn, mx in rows[:top]:
        short = n if len(n) < 110 else n[:107] + '...'
        print(f'{c:7d} {s/1e6:10.3f} {a/1e3:10.1f} {mn/1e3:9.1f} {mx/1e3:9.1f} {s/total:7.2%}  {short}')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
