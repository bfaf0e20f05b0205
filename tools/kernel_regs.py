"""Per-kernel register / LDS usage of the built libmbx.so (from the code objects' metadata notes).
    python tools/kernel_regs.py [pattern] [--lib path]"""
import os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'


def main():
    src = os.path.join(ROOT, 'motionbert_amd', 'libmbx.so')
    if '--lib' in sys.argv:
        k = sys.argv.index('--lib'); src = sys.argv[k + 1]; del sys.argv[k:k + 2]
    pat = sys.argv[1] if len(sys.argv) > 1 else ''
    tmp = tempfile.mkdtemp()
    lib = os.path.join(tmp, 'libmbx.so')
    shutil.copy(src, lib)
    subprocess.run([f'{LLVM}/llvm-objdump', '--offloading', lib], check=True, capture_output=True, cwd=tmp)
    rows = []
    for f in sorted(os.listdir(tmp)):
        if 'amdgcn' not in f:
            continue
        notes = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
        for blk in notes.split('- .agpr_count:')[1:]:
            g = lambda k: re.search(r'\.' + k + r':\s+(\S+)', blk)
            name = subprocess.run(['c++filt', g('name').group(1)], capture_output=True, text=True).stdout.strip()
            rows.append((name.split('(')[0][:70], int(g('vgpr_count').group(1)), int(blk.split()[0]), int(g('sgpr_count').group(1)),
                         int(g('group_segment_fixed_size').group(1)), int(g('private_segment_fixed_size').group(1))))
    print(f'{"kernel":70s} vgpr agpr sgpr  lds  scratch')
    for r in sorted(rows):
        if pat in r[0]:
            print(f'{r[0]:70s} {r[1]:4d} {r[2]:4d} {r[3]:4d} {r[4]:6d} {r[5]:4d}')
    shutil.rmtree(tmp)


if __name__ == '__main__':
    main()
