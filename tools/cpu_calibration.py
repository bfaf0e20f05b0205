"""CPU-baseline calibration: the unmodified reference DSTformer and the torch port (oracle/torch_model.py: the reference's operator
mix as one plain torch function) timed back to back on the SAME host cores with bench.cpu_baseline's own code and budget.  bench.py reports
`cpu_baseline.kind = "port"` on hosts without a reference checkout (the GPU box); this script is where the ratio between
the two comes from -- run it where /root/reference (or $MOTIONBERT_REFERENCE) exists and commit the log under profiles/.

    python tools/cpu_calibration.py [--budget 20] [--repeat 5] > profiles/r04_cpu_calibration.txt
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--budget', type=float, default=20.0)
    ap.add_argument('--frames', type=int, default=243)
    ap.add_argument('--repeat', type=int, default=5)
    args = ap.parse_args()
    ref_dir = os.environ.get('MOTIONBERT_REFERENCE', '/root/reference')
    runs = []
    for rep in range(args.repeat):                                # alternate the two legs: shared hosts drift by tens of percent
        os.environ['MOTIONBERT_REFERENCE'] = ref_dir
        ref = bench.cpu_baseline(bench.FULL, args.frames, args.budget)
        os.environ['MOTIONBERT_REFERENCE'] = '/nonexistent'       # forces the port leg
        port = bench.cpu_baseline(bench.FULL, args.frames, args.budget)
        runs.append(dict(reference=ref, port=port,
                         port_over_reference=round(port['value'] / ref['value'], 3) if ref['kind'] == 'reference' else None))
    ratios = [r['port_over_reference'] for r in runs if r['port_over_reference']]
    mean = sum(ratios) / len(ratios) if ratios else None
    out = dict(runs=runs, port_over_reference_range=[min(ratios), max(ratios)] if ratios else None,
               port_over_reference_mean=round(mean, 3) if ratios else None,
               port_over_reference_spread=round((max(ratios) - min(ratios)) / 2, 3) if ratios else None,
               note=f'reference checkout: {ref_dir}; same process, same thread count, legs alternated {args.repeat}x')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
