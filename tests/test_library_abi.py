"""The C ABI: libmbx.so builds for gfx950, loads without a GPU and exports every symbol that
include/mbx.h declares (no compute is launched here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from motionbert_amd import build, hip_ops
    if not os.path.exists(hip_ops.LIB_PATH):
        build.build(verbose=False)
    return hip_ops.load_library()


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'mbx.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mbx_[a-z0-9_]+)\s*\(', src)))


def test_header_and_binding_agree(lib):
    from motionbert_amd import hip_ops
    syms = declared_symbols()
    assert len(syms) >= 20
    assert sorted(hip_ops.SIGNATURES) == syms
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/mbx.h but not exported by libmbx.so'


def test_argument_errors_are_reported_not_crashed(lib):
    # shape validation happens before any launch, so this is safe without a GPU
    assert lib.mbx_version() >= 100
    rc = lib.mbx_gemm_nt(None, None, None, 0, None, None, None, None, None, 4, 8, 64, 1, None)
    assert rc != 0 and b'null' in lib.mbx_last_error()
    rc = lib.mbx_attn_fwd(1, 1, 1, 1, 4, 17, 8, 48, 0.1, 0, 1, None)
    assert rc != 0 and b'head dim' in lib.mbx_last_error()
    assert lib.mbx_gemm_tn_ws(4131, 1536, 512) > 0 and lib.mbx_layernorm_bwd_ws(512) > 0


def test_device_code_policy(lib, tmp_path):
    """Static checks on the gfx950 code objects of the built library:
    * no kernel touches scratch memory (`scratch_load/store`): a stack object or a register spill in an epilogue turns
      into HBM traffic -- round 2 found 64 B/lane of scratch in the fc1-GELU epilogue (+41 % HBM writes per launch);
    * every cross-lane reduction is VALU-only (DPP / permlane), no `ds_bpermute` / `ds_permute` / `ds_swizzle`: a house rule
      since round 1 (the LDS crossbar shares the LDS pipe with the kernels' tile traffic).  The round-1 suspicion of a
      hardware fault behind it was NOT reproduced by tools/probes/bpermute_stress.hip and is retracted in DESIGN.md;
    * the bf16 MFMA kernels are in the binary."""
    import shutil
    import subprocess
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not os.path.exists(objdump):
        pytest.skip('llvm-objdump not available')
    from motionbert_amd import hip_ops
    copy = tmp_path / 'libmbx.so'          # llvm-objdump writes the extracted bundles next to its input
    shutil.copy(hip_ops.LIB_PATH, copy)
    subprocess.run([objdump, '--offloading', str(copy)], check=True, capture_output=True, cwd=tmp_path)
    objs = [p for p in tmp_path.iterdir() if 'amdgcn' in p.name]
    assert objs, 'no device code objects found in libmbx.so'
    n_mfma = 0
    for o in objs:
        asm = subprocess.run([objdump, '-d', '--mcpu=gfx950', str(o)], check=True, capture_output=True, text=True).stdout
        assert 'scratch_load' not in asm and 'scratch_store' not in asm, f'{o.name}: a kernel uses scratch memory'
        assert 'ds_bpermute' not in asm and 'ds_permute' not in asm and 'ds_swizzle' not in asm, f'{o.name}: LDS-crossbar permute found'
        n_mfma += asm.count('v_mfma_f32_32x32x16_bf16')
    assert n_mfma > 100, 'expected the bf16 MFMA kernels in the device code'


def test_asm_prefetch_registers_are_left_alone(lib, tmp_path):
    """The N-resident row-owner kernels (csrc/gemm_rows_n.hip) fetch their token fragments with inline-asm loads four stages ahead of the
    MFMAs that consume them, ordered by the kernels' own counted waits.  To the compiler an asm output is valid where the statement
    stands: if register pressure made it copy, spill or reuse one of those registers before the data has landed, the copy would hold
    stale bits (round 5 found both failure modes in the epilogue of an earlier version; round 6 moved the last register prefetch of an
    epilogue input -- xhat -- to LDS-DMA, which has no register side).  Checked on the disassembly of both kernels: the first instruction
    that names a destination register of a token load after the load is an MFMA (never a copy, a spill or a VALU operation), and no
    load of this form exists besides the token loads (8 in the preamble + 16 per trip, two trips in the code)."""
    import re
    import shutil
    import subprocess
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not os.path.exists(objdump):
        pytest.skip('llvm-objdump not available')
    from motionbert_amd import hip_ops
    copy = tmp_path / 'libmbx.so'
    shutil.copy(hip_ops.LIB_PATH, copy)
    subprocess.run([objdump, '--offloading', str(copy)], check=True, capture_output=True, cwd=tmp_path)
    seen = 0
    for o in [p for p in tmp_path.iterdir() if 'amdgcn' in p.name]:
        asm = subprocess.run([objdump, '-d', '--mcpu=gfx950', str(o)], check=True, capture_output=True, text=True).stdout
        for kern in ('rows_n_lnbwd_kernel', 'rows_n_resid_ln_kernel'):
            m = re.search(r'^[0-9a-f]+ <_Z\d+' + kern + r'[^>]*>:\n(.*?)s_endpgm', asm, re.S | re.M)
            if not m:
                continue
            seen += 1
            lines = [l.split('//')[0] for l in m.group(1).split('\n')]
            loads = []      # (line, registers) of global_load_dwordx4 vdst, voff, s[..]: the SGPR-base form only the token loads use
            for n, l in enumerate(lines):
                mm = re.search(r'global_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[\d+:\d+\]', l)
                if mm:
                    loads.append((n, set(range(int(mm.group(1)), int(mm.group(2)) + 1))))
            # 8 in the preamble, 16 in the ordinary trip, 8 in the peeled last trip (its second half has no next trip to fetch for)
            assert len(loads) == 32, (kern, len(loads))
            for n, regs in loads:
                for k in range(n + 1, len(lines)):
                    used = set()
                    for mm in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', lines[k]):
                        used.update([int(mm.group(3))] if mm.group(3) else range(int(mm.group(1)), int(mm.group(2)) + 1))
                    if used & regs:
                        assert 'v_mfma' in lines[k], f'{kern}: "{lines[k].strip()}" touches the registers of the token load in line {n} before an MFMA read them'
                        break
    assert seen == 2, 'row-owner kernels not found in the device code'

