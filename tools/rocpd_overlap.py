"""How much of a (multi-stream) step actually overlaps: from a rocprofv3 rocpd kernel trace, the wall span covered by at least
one kernel, by two or more, the plain sum of kernel durations, and per kernel family its mean duration and the share of its
running time during which another kernel was resident too.

    python tools/rocpd_overlap.py /tmp/kt/..._results.db [skip_first_dispatches]
"""
import re
import sqlite3
import sys
from collections import defaultdict


def family(n):
    n = re.sub(r'^void ', '', n)
    return n.split('(')[0][:60]


def main(path, skip=0):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()[skip:]
    ev = []
    for i, (n, s, e) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, -1, i))
    ev.sort()
    active, last, busy1, busy2 = set(), None, 0, 0
    shared = defaultdict(float)          # per dispatch: time with company
    pair = defaultdict(float)            # (family a, family b) -> co-resident time
    fams = [family(n) for n, _, _ in rows]
    for t, d, i in ev:
        if last is not None and active:
            dt = t - last
            busy1 += dt
            if len(active) >= 2:
                busy2 += dt
                for j in active:
                    shared[j] += dt
                act = sorted(active)
                for x in range(len(act)):
                    for y in range(x + 1, len(act)):
                        pair[tuple(sorted((fams[act[x]], fams[act[y]])))] += dt
        if d == 1:
            active.add(i)
        else:
            active.discard(i)
        last = t
    total = sum(e - s for _, s, e in rows)
    span = rows[-1][2] - rows[0][1]
    print(f'# {len(rows)} dispatches: span {span/1e6:.2f} ms, >=1 kernel resident {busy1/1e6:.2f} ms, >=2 resident {busy2/1e6:.2f} ms, '
          f'sum of durations {total/1e6:.2f} ms')
    fam = defaultdict(lambda: [0, 0.0, 0.0])
    for i, (n, s, e) in enumerate(rows):
        f = fam[family(n)]
        f[0] += 1
        f[1] += e - s
        f[2] += shared[i]
    print(f'{"calls":>6} {"total_ms":>9} {"avg_us":>9} {"shared":>7}  kernel')
    for k, (c, tot, sh) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f'{c:6d} {tot/1e6:9.2f} {tot/c/1e3:9.1f} {sh/max(tot,1):7.1%}  {k}')
    print('# co-resident pairs (ms)')
    for (a, b), tt in sorted(pair.items(), key=lambda kv: -kv[1])[:25]:
        print(f'{tt/1e6:9.2f}  {a[:44]:44s} | {b[:44]}')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)

