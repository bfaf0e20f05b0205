"""profiles/r06_bf16_headroom.txt: for every frozen bf16 gate of tests/test_gpu_model.py::test_baseline_shape_fixture_fwd_bwd, the value
measured on the GPU / the gate, on both reference-minted fixtures.  Reads the report the GPU test run
leaves in gpurun_out/model_parity.json.
    python tools/headroom_report.py [gpurun_out/model_parity.json] > profiles/r06_bf16_headroom.txt"""
import json
import sys

path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/model_parity.json'
rep = json.load(open(path))
print('# bf16 gates of tests/test_gpu_model.py::test_baseline_shape_fixture_fwd_bwd: measured / gate = fraction used (1.0 = at the gate)')
print('# gate forms: out < min(2 ac, max(4e-2, ac)); dx and global gradient < min(2 ac, max(0.08, ac)); per tensor <= max(3 ac_t, 0.08 of the global norm)')
print(f"# {'fixture':34s} {'quantity':18s} {'measured':>10s} {'gate':>10s} {'used':>7s}")
for key in sorted(k for k in rep if k.startswith('headroom.')):
    h = rep[key]
    for q in ('out', 'dx', 'grad_global'):
        v, g = h[q]
        print(f"  {key[9:]:34s} {q:18s} {v:10.5f} {g:10.5f} {v / g:7.2f}")
    v, g, n = h['per_tensor_worst']
    print(f"  {key[9:]:34s} {'worst tensor':18s} {v:10.5f} {g:10.5f} {v / g:7.2f}   {n}")
    ac = rep.get('fixture.' + key[9:], {}).get('reference_autocast_bf16')
    if ac:
        print(f"  {'':34s} (the reference under autocast on this fixture: out {ac['out']:.4f}, global gradient {ac['grad_global']:.4f}, worst tensor {ac['worst_grad']:.4f})")
