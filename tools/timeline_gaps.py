"""From a rocprofv3 kernel trace (rocpd sqlite) of the TWO-stream bench: how much of a step has NO kernel running, how much has
one, how much two or more (the st / ts streams overlapping), and the longest idle gaps with the kernels around them.
    python tools/timeline_gaps.py <results.db> [steps]"""
import sqlite3
import sys


def main(path, steps=5):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = cur.execute(f"select start, end, {name_col} from kernels order by start").fetchall()
    adam = [i for i, r in enumerate(rows) if 'adamw' in r[2]]
    if len(adam) < steps + 1:
        print('not enough steps in the trace', len(adam))
        return
    lo, hi = adam[-steps - 1] + 1, adam[-1] + 1           # the last `steps` steps: from after an AdamW launch to the end of a later one
    seg = rows[lo:hi]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    ev = sorted([(r[0], 1) for r in seg] + [(r[1], -1) for r in seg])
    depth, last, acc = 0, t0, {0: 0, 1: 0, 2: 0}
    for t, d in ev:
        acc[min(depth, 2)] += t - last
        last, depth = t, depth + d
    wall = t1 - t0
    print(f'# {path}: last {steps} steps, {len(seg)} kernels, {wall / steps / 1e6:.2f} ms per step (GPU timeline), kernel time {sum(r[1] - r[0] for r in seg) / steps / 1e6:.2f} ms per step')
    print(f'# no kernel running {acc[0] / wall:6.2%}   exactly one {acc[1] / wall:6.2%}   two or more {acc[2] / wall:6.2%}')
    gaps, end_so_far, prev = [], seg[0][1], seg[0][2]
    for s, e, n in seg[1:]:
        if s > end_so_far:
            gaps.append((s - end_so_far, prev, n))
        if e > end_so_far:
            end_so_far, prev = e, n
    gaps.sort(reverse=True)
    print(f'# {len(gaps)} idle gaps; the largest:')
    for g, a, b in gaps[:12]:
        print(f'  {g / 1e3:8.1f} us   after {a[:70]}   before {b[:70]}')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5)
