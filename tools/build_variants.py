"""Build diagnostic / A-B variants of the kernel library next to the product one.

    python tools/build_variants.py diag                      # -DMBX_DIAG: env switches (MBX_NT256_MASK, MBX_DBG, ...) are live
    python tools/build_variants.py name -DFOO=1 -DBAR=2      # any extra compile flags

Output: tools/variants/libmbx_<name>.so (git-ignored, travels to the GPU box with gpurun).  The measurement scripts pick a
variant with MBX_LIB=tools/variants/libmbx_<name>.so; the product library (motionbert_amd/libmbx.so) has no switches."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from motionbert_amd import build as B


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    if name == 'diag' and '-DMBX_DIAG' not in extra:
        extra = ['-DMBX_DIAG'] + extra
    out = os.path.join(ROOT, 'tools', 'variants')
    obj = os.path.join(out, '_obj_' + name)
    os.makedirs(obj, exist_ok=True)
    hipcc = B._hipcc()

    def cc(src):
        o = os.path.join(obj, src.replace('.hip', '.o'))
        subprocess.run([hipcc] + B.flags_for(src) + extra + ['-c', os.path.join(B.CSRC, src), '-o', o], check=True)
        return o
    with ThreadPoolExecutor(max_workers=len(B.SOURCES)) as ex:
        objs = list(ex.map(cc, B.SOURCES))
    lib = os.path.join(out, f'libmbx_{name}.so')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs, check=True)
    print(lib)


if __name__ == '__main__':
    main()
