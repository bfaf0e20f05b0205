"""Build libmbx.so (the gfx950 kernel library) in-tree with hipcc.

    python -m motionbert_amd.build            # incremental
    python -m motionbert_amd.build --force

hipcc cross-compiles for gfx950 without a GPU present.  The shared object is git-ignored but
travels with the working tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'csrc', '_obj')
LIB = os.path.join(HERE, 'libmbx.so')
SOURCES = ['elementwise.hip', 'gemm.hip', 'gemm_pipe.hip', 'mlp_fused.hip', 'gemm_rows.hip', 'gemm_rows_n.hip', 'attention.hip', 'train_step.hip', 'augment.hip', 'probe.hip']
HEADERS = [os.path.join(CSRC, 'mbx_common.h'), os.path.join(CSRC, 'gelu_fast.h'), os.path.join(CSRC, 'lds_stream.h'), os.path.join(os.path.dirname(HERE), 'include', 'mbx.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wno-unused-result']
# per-source additions.  gemm_rows_n.hip: the body of the trip loop (32 slots with their folded-after-unrolling switches) x 4 or 8 stages
# exceeds LLVM's default size limit for `#pragma unroll` in the 8-tile instantiation -- the loop then stays rolled, the token ring is
# indexed at run time and becomes a 272-byte stack object (tests/test_library_abi.py::test_device_code_policy would reject the scratch)
EXTRA_FLAGS = {'gemm_rows_n.hip': ['-mllvm', '-pragma-unroll-threshold=65536']}


def flags_for(src: str):
    return FLAGS + EXTRA_FLAGS.get(os.path.basename(src), [])


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (ROCm toolchain required to build libmbx.so)')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + HEADERS):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [hipcc] + flags_for(s) + ['-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=max(1, len(jobs))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s.replace('.hip', '.o')) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
