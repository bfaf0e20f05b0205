"""Per-kernel roofline table from the round's profile passes: launch time (rocprofv3 kernel trace), MFMA instruction count, HBM bytes
(FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes) -> achieved TFLOP/s and TB/s, the roofline that bounds the kernel
(min(2500 TFLOP/s bf16 dense, intensity x 8 TB/s); HBM-bound kernels also against the ~5.5 TB/s that mixed read + write streams
reach on this part) and the fraction of it.

    python tools/roofline_report.py gpurun_out profiles/r03_kernel_stats_singlestream.txt > profiles/r03_roofline_table.txt
"""
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from pmc_table import parse        # noqa: E402

PEAK_TF, PEAK_TB, ACH_TB = 2500.0, 8.0, 5.5
FLOP_PER_MFMA = 2 * 32 * 32 * 16      # v_mfma_f32_32x32x16_bf16, per wave-level instruction


def main(outdir, stats):
    mf, fe, wr = (parse(f'{outdir}/pmc_bench_{t}.txt') for t in ('mfma', 'fetch', 'write'))
    dur, calls = {}, {}
    for line in open(stats):
        f = line.split(None, 6)
        if len(f) == 7 and f[0].isdigit():
            dur[f[6].strip()[:58]] = float(f[2])
            calls[f[6].strip()[:58]] = int(f[0])
    once = [v for k, v in calls.items() if 'embed_fwd' in k or 'head_fwd' in k]
    steps = max(1, once[0] if once else min(calls.values()))       # launches of a once-per-step kernel = steps in the trace
    print('# per-kernel roofline, 64 clips x 243 frames, bf16, one stream (tools/roofline_report.py; inputs: the kernel trace and the three PMC passes of')
    print('# tools/gpu_session.sh profiles).  flops = SQ_INSTS_MFMA x 32768; bytes = FETCH_SIZE x 2 + WRITE_SIZE; bound = min(2500 TF/s, flop/byte x 8 TB/s).')
    print(f'{"kernel":58s} {"/step":>5s} {"us":>7s} {"TF/s":>7s} {"TB/s":>6s} {"fl/B":>6s} {"bound":>5s} {"roof":>7s} {"frac":>6s} {"of 5.5TB/s":>10s}')
    rows = []
    for k, c in mf.items():
        key = k[:58]
        if key not in dur:
            continue
        us = dur[key]
        insts = c.get('SQ_INSTS_MFMA', (0, 0.0))[1]
        flops = insts * FLOP_PER_MFMA
        by = (fe.get(k, {}).get('FETCH_SIZE', (0, 0.0))[1] * 2 + wr.get(k, {}).get('WRITE_SIZE', (0, 0.0))[1]) * 1024
        tf, tb = flops / us / 1e6, by / us / 1e6
        ai = flops / by if by else 0.0
        roof_tf = min(PEAK_TF, ai * PEAK_TB) if flops else 0.0
        bound = 'mfma' if flops and ai * PEAK_TB >= PEAK_TF else 'hbm'
        frac = (tf / roof_tf) if bound == 'mfma' or (flops and roof_tf) else tb / PEAK_TB
        rows.append((calls[key] * us, key, calls[key] // steps, us, tf, tb, ai, bound, roof_tf if flops else PEAK_TB, frac, tb / ACH_TB))
    tot = sum(r[0] for r in rows)
    for t, key, n, us, tf, tb, ai, bound, roof, frac, fa in sorted(rows, reverse=True):
        if t < 0.002 * tot:
            continue
        roof_s = f'{roof:7.0f}' if bound == 'mfma' or tf > 0 else f'{roof:5.1f}TB'
        print(f'{key:58s} {n:5d} {us:7.1f} {tf:7.1f} {tb:6.2f} {ai:6.1f} {bound:>5s} {roof_s:>7s} {frac:6.1%} {fa:10.1%}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
