"""Disassembly of one kernel of the built libmbx.so, or a run-length summary of its interesting instructions.
    python tools/kernel_asm.py <kernel-name-substring> [--full] [--lib path]"""
import os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
KEYS = ('s_barrier', 's_waitcnt', 's_cbranch', 's_branch', 'v_mfma', 'ds_read_b64_tr', 'ds_read_b128', 'global_load_lds', 'global_load', 'global_store',
        'buffer_', 'ds_write', 'ds_read', 'scratch_', 's_endpgm')


def kernel_asm(name, lib=None):
    lib = lib or os.path.join(ROOT, 'motionbert_amd', 'libmbx.so')
    tmp = tempfile.mkdtemp()
    try:
        cp = os.path.join(tmp, 'lib.so')
        shutil.copy(lib, cp)
        subprocess.run([f'{LLVM}/llvm-objdump', '--offloading', cp], check=True, capture_output=True, cwd=tmp)
        for f in sorted(os.listdir(tmp)):
            if 'amdgcn' not in f:
                continue
            asm = subprocess.run([f'{LLVM}/llvm-objdump', '-d', '--mcpu=gfx950', os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            out, on = [], False
            for line in asm.splitlines():
                m = re.match(r'^[0-9a-f]+ <(.*)>:$', line)
                if m:
                    on = name in m.group(1)
                    if on and out:
                        break
                if on:
                    out.append(line)
            if out:
                return out
    finally:
        shutil.rmtree(tmp)
    return []


def main():
    name = sys.argv[1]
    lib = sys.argv[sys.argv.index('--lib') + 1] if '--lib' in sys.argv else None
    lines = kernel_asm(name, lib)
    if '--full' in sys.argv:
        print('\n'.join(lines))
        return
    last, n = None, 0
    for line in lines:
        t = line.split('//')[0].split()
        if not t:
            continue
        op = t[0]
        key = next((k for k in KEYS if op.startswith(k)), None)
        if key is None:
            continue
        tag = op if key in ('s_waitcnt', 's_cbranch', 's_branch') else key
        if key == 's_waitcnt':
            tag = ' '.join(t[:4])
        if tag == last:
            n += 1
        else:
            if last:
                print(f'{last} x{n}')
            last, n = tag, 1
    print(f'{last} x{n}')


if __name__ == '__main__':
    main()
