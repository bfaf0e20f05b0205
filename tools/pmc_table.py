"""Merge the three PMC passes of tools/gpu_session.sh profiles (gpurun_out/pmc_bench_{mfma,fetch,write}.txt) and the kernel-trace
summary into one per-kernel table (the format bench.py's `roofline.traffic` reads).

    python tools/pmc_table.py gpurun_out profiles/r01_kernel_stats_v6_singlestream.txt > profiles/r01_pmc_bench_v6.txt

Corrections as the micro-architecture guide prescribes: FETCH_SIZE and WRITE_SIZE are in KB; FETCH_SIZE is doubled on
gfx950 (128-byte requests counted as 64); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x 256 CU x GRBM_GUI_ACTIVE / 8 XCD)."""
import re
import sys


def parse(path):
    d = {}
    for line in open(path):
        m = re.match(r'(.{60}) (\S+)\s+n=\s*(\d+) avg=\s*([\d.]+)', line)
        if m:
            d.setdefault(m.group(1).strip(), {})[m.group(2)] = (int(m.group(3)), float(m.group(4)))
    return d


def main(outdir, stats):
    mf, fe, wr = (parse(f'{outdir}/pmc_bench_{t}.txt') for t in ('mfma', 'fetch', 'write'))
    dur = {}
    for line in open(stats):
        f = line.split(None, 6)
        if len(f) == 7 and f[0].isdigit():
            dur[f[6].strip()[:58]] = float(f[2])
    print('# PMC summary per kernel (one bench step + warm-up + instrumented step, MBX_DUAL_STREAM=0, 64 clips x 243 frames, bf16). averages per launch.')
    print('# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x 256 CU x GRBM_GUI_ACTIVE/8 XCD);  HBM bytes: FETCH_SIZE (KB; x2 on gfx950 for wide streams), WRITE_SIZE (KB)')
    print(f'{"kernel":58s} {"n":>4s} {"us":>8s} {"mfma_busy":>9s} {"lds_conf%":>9s} {"fetch_MB":>9s} {"fetchx2":>8s} {"write_MB":>9s} {"L2hit%":>7s}')
    rows = []
    for k in mf:
        c = mf[k]
        n = c['GRBM_GUI_ACTIVE'][0]
        gui = c['GRBM_GUI_ACTIVE'][1]
        busy = c.get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0.0))[1] / (4 * 256 * gui / 8) if gui else 0.0
        conf = c.get('SQ_LDS_BANK_CONFLICT', (0, 0.0))[1] / max(c.get('SQ_LDS_IDX_ACTIVE', (0, 1.0))[1], 1.0)
        fetch = fe.get(k, {}).get('FETCH_SIZE', (0, 0.0))[1] / 1024
        hit = fe.get(k, {}).get('TCC_HIT_sum', (0, 0.0))[1]
        miss = wr.get(k, {}).get('TCC_MISS_sum', (0, 0.0))[1]
        write = wr.get(k, {}).get('WRITE_SIZE', (0, 0.0))[1] / 1024
        us = next((v for kk, v in dur.items() if kk[:56] == k[:56]), 0.0)
        rows.append((n * (2 * fetch + write), f'{k[:58]:58s} {n:4d} {us:8.1f} {busy:9.1%} {conf:9.1%} {fetch:9.1f} {2 * fetch:8.1f} {write:9.1f} {hit / max(hit + miss, 1.0):7.1%}'))
    for _, r in sorted(rows, reverse=True):
        print(r)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
