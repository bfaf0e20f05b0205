"""ctypes binding of libmbx.so: the kernel provider (`ops`) used by motionbert_amd.engine.

Thin by design: every method checks nothing but dtypes/contiguity, takes raw device pointers from
the torch tensors (PyTorch owns all memory), passes the current HIP stream and raises
RuntimeError(mbx_last_error()) on a non-zero return.  There is no CPU implementation and no
fallback: if libmbx.so cannot be loaded the import of this module's `get()` fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Dict, List, Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MBX_LIB', os.path.join(_HERE, 'libmbx.so'))   # MBX_LIB: A/B builds of the kernel library

MBX_F32, MBX_BF16, MBX_BF16_LO = 0, 1, 2
_DT = {torch.float32: MBX_F32, torch.bfloat16: MBX_BF16}

_vp, _i, _f, _sz, _i64p = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_void_p

# name -> (restype, argtypes); mirrors include/mbx.h one to one
SIGNATURES = {
    'mbx_last_error': (C.c_char_p, []),
    'mbx_version': (_i, []),
    'mbx_prep_weights': (_i, [_i64p, _i, _i, _i, _i, _vp]),
    'mbx_embed_fwd': (_i, [_vp] * 6 + [_i] * 5 + [_vp]),
    'mbx_embed_bwd_ws': (_sz, [_i] * 4),
    'mbx_embed_bwd': (_i, [_vp] * 8 + [_i] * 5 + [_vp, _vp]),
    'mbx_embed_bwd_pair': (_i, [_vp] * 9 + [_i] * 5 + [_vp, _vp]),
    'mbx_layernorm_fwd': (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'mbx_layernorm_bwd_ws': (_sz, [_i]),
    'mbx_layernorm_bwd': (_i, [_vp] * 11 + [_i, _i, _i, _vp, _vp]),
    'mbx_gemm_nt': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mbx_gemm_tn_ws': (_sz, [_i, _i, _i]),
    'mbx_gemm_tn': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'mbx_fold_norm_weights': (_i, [_i64p, _i, _i, _i, _vp]),
    'mbx_gemm_nt_dgelu_stats': (_i, [_vp] * 7 + [_i, _i, _i, _vp]),
    'mbx_gemm_nt_gelu_d': (_i, [_vp] * 5 + [_i, _i, _i, _vp]),
    'mbx_gemm_nt_mul': (_i, [_vp] * 4 + [_i, _i, _i, _vp]),
    'mbx_lnbwd_rowc': (_i, [_vp, _i, _vp, _vp, _i, _i, _vp]),
    'mbx_gemm_nt_lnbwd': (_i, [_vp] * 8 + [_i, _i, _i, _vp]),
    'mbx_gemm_nt_lnbwd_t': (_i, [_vp] * 8 + [_i, _i, _i, _vp]),
    'mbx_unfold_norm_grads_ws': (_sz, [_i, _i]),
    'mbx_unfold_norm_grads': (_i, [_vp] * 7 + [_i, _i, _vp, _vp]),
    'mbx_mlp_pack_bytes': (_sz, [_i, _i]),
    'mbx_mlp_pack_weights': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'mbx_proj_mlp_pack_bytes': (_sz, [_i, _i]),
    'mbx_proj_mlp_pack_weights': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    'mbx_proj_mlp_fused_fwd': (_i, [_vp] * 8 + [_f, _i, _i, _i, _vp]),
    'mbx_mlp_fused_fwd': (_i, [_vp, _i] + [_vp] * 7 + [_f, _vp, _vp, _i, _i, _i, _vp]),
    'mbx_rows_pack_bytes': (_sz, [_i, _i]),
    'mbx_rows_pack_nk': (_i, [_vp, _vp, _i, _i, _vp]),
    'mbx_rows_gemm_nk': (_i, [_vp] * 7 + [_i, _i, _i, _vp]),
    'mbx_rows_gemm_nk_ln': (_i, [_vp] * 4 + [_f, _vp, _i, _i, _i, _vp]),
    'mbx_gelu_fwd': (_i, [_vp, _vp, _sz, _i, _vp]),
    'mbx_split_bf16': (_i, [_vp, _vp, _vp, _sz, _vp]),
    'mbx_gemm_nt_x3p': (_i, [_vp] * 5 + [_i] + [_vp] * 4 + [_i, _i, _i, _vp]),
    'mbx_layernorm_bwd_planes': (_i, [_vp] * 12 + [_i, _i, _vp, _vp]),
    'mbx_layernorm_fwd_planes': (_i, [_vp] * 3 + [_f] + [_vp] * 4 + [_i, _i, _vp]),
    'mbx_gemm_nt_x3': (_i, [_vp] * 5 + [_i] + [_vp] * 5 + [_i, _i, _i, _vp]),
    'mbx_gemm_tn_x3_workspace': (_sz, [_i, _i, _i]),
    'mbx_gemm_tn_x3': (_i, [_vp] * 6 + [_i, _i, _i, _vp, _vp]),
    'mbx_attn_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    'mbx_attn_bwd_planes': (_i, [_vp] * 6 + [_i, _i, _i, _i, _i, _f, _i, _f, C.c_uint64, _vp]),
    'mbx_attn_bwd': (_i, [_vp] * 5 + [_i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    'mbx_attn_fwd_drop': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _f, C.c_uint64, _vp]),
    'mbx_attn_bwd_drop': (_i, [_vp] * 5 + [_i, _i, _i, _i, _i, _f, _i, _i, _f, C.c_uint64, _vp]),
    'mbx_attn_bwd_stats': (_i, [_vp] * 8 + [_i, _i, _i, _i, _i, _f, _i, _vp]),
    'mbx_fuse_fwd': (_i, [_vp] * 6 + [_i, _i, _vp]),
    'mbx_fuse_ln_fwd': (_i, [_vp] * 12 + [_f] + [_vp] * 2 + [_i] * 3 + [_vp]),
    'mbx_fuse_bwd_ws': (_sz, [_i]),
    'mbx_fuse_bwd': (_i, [_vp] * 11 + [_i, _i, _i, _vp, _vp]),
    'mbx_fuse_bwd_pair': (_i, [_vp] * 10 + [_i, _i, _vp, _vp]),
    'mbx_average': (_i, [_vp, _vp, _vp, _sz, _vp]),
    'mbx_average_bwd': (_i, [_vp] * 5 + [_sz, _i, _vp]),
    'mbx_head_fwd': (_i, [_vp] * 4 + [_i, _i, _i, _vp]),
    'mbx_head_bwd_ws': (_sz, [_i, _i]),
    'mbx_head_bwd': (_i, [_vp] * 6 + [_i, _i, _i, _i, _vp, _vp]),
    'mbx_tanh_bwd': (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    'mbx_pose_loss_ws': (_sz, [_i, _i]),
    'mbx_pose_loss': (_i, [_vp, _vp, _f, _f, _vp, _vp, _f, _i, _i, _i, _vp, _vp]),
    'mbx_loss_2d_weighted_ws': (_sz, [_i, _i]),
    'mbx_loss_2d_weighted': (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _f, _i, _i, _i, _vp, _vp]),
    'mbx_pool_rep_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, C.c_uint64, _vp]),
    'mbx_tanh_pool_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, C.c_uint64, _i, _vp]),
    'mbx_dropout': (_i, [_vp, _vp, _sz, _f, C.c_uint64, _i, _vp]),
    'mbx_residual_drop': (_i, [_vp, _vp, _sz, _i, _i, _f, C.c_uint64, _f, C.c_uint64, _vp]),
    'mbx_grad_drop': (_i, [_vp, _vp, _sz, _i, _i, _f, C.c_uint64, _f, C.c_uint64, _i, _vp]),
    'mbx_augment2d': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp] + [_f] * 8 + [_i, C.c_uint64, _vp]),
    'mbx_embed_fwd_tta': (_i, [_vp] * 7 + [_i] * 5 + [_vp]),
    'mbx_flip_average': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mbx_rows_n_pack_bytes': (_sz, [_i, _i]),
    'mbx_rows_lnbwd_t': (_i, [_vp] * 6 + [_i, _i, _i, _vp]),
    'mbx_rows_n_pack_many': (_i, [_i64p, _i, _i, _i, _vp]),
    'mbx_rows_resid_ln': (_i, [_vp] * 8 + [_f, _i, _i, _i, _vp]),
    'mbx_mfma_probe_ws': (_sz, [_i]),
    'mbx_mfma_probe': (_i, [_vp, _i, _i, C.c_uint, _vp, _vp]),
    'mbx_adamw_step': (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _f, _f, _f, _f, _i, _vp]),
}


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """dlopen libmbx.so and attach the prototypes.  Works without a GPU (symbols only)."""
    if not os.path.exists(path):
        raise RuntimeError(f'{path} not found: build it with `python -m motionbert_amd.build` '
                           '(hipcc --offload-arch=gfx950); there is no fallback implementation')
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype, fn.argtypes = res, args
    return lib


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class HipOps:
    """Kernel provider backed by libmbx.so.  Stateless apart from size caches."""

    multi_stream = True   # every call takes the current stream: safe to drive from several HIP streams
    grad_stream_t = True  # mbx_gemm_nt_lnbwd_t exists: the gradient residual stream may travel as bf16 inside a Block

    def __init__(self, lib: Optional[C.CDLL] = None):
        self.lib = lib or load_library()
        self._ws_cache: Dict[tuple, int] = {}
        self._desc_cache: Dict[tuple, dict] = {}
        self.weight_cache: Dict[int, tuple] = {}      # device index -> (key, prepared weights) of the last no-grad forward (engine.prepare_weights)
        self._lock = threading.Lock()

    # ------------------------------------------------------------------ plumbing
    def _ck(self, rc: int):
        if rc != 0:
            raise RuntimeError('libmbx: ' + self.lib.mbx_last_error().decode())

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def _ws(self, key: tuple, fn, *args, device=None) -> torch.Tensor:
        n = self._ws_cache.get(key)
        if n is None:
            n = int(fn(*args))
            self._ws_cache[key] = n
        return torch.empty(max(n, 16), dtype=torch.uint8, device=device)

    # ------------------------------------------------------------------ weights
    def prep_weights(self, P: Dict[str, torch.Tensor], names: List[str], tdtype, need_t: bool, x3: bool = False):
        """T-typed copies of every Linear weight: Wn[name] [N,K] and (for backward) Wt[name] [K,N].
        x3 (precision 'bf16x3'): every entry is a (hi, lo) pair of bf16 planes, w = hi + lo up to 2^-16 relative."""
        if x3:
            hi_n, hi_t = self._prep_weights(P, names, torch.bfloat16, need_t, False)
            lo_n, lo_t = self._prep_weights(P, names, torch.bfloat16, need_t, True)
            return ({n: (hi_n[n], lo_n[n]) for n in names}, {n: (hi_t[n], lo_t[n]) for n in names} if need_t else {})
        return self._prep_weights(P, names, tdtype, need_t, False)

    def _desc_room(self, kind: str, keep: int = 16):
        """Called under the lock before a descriptor-table entry of `kind` is added: once `keep` of that kind exist (models come and
        go, operands move) they are dropped -- only that kind, the dict is shared by the three packers."""
        stale = [k for k in self._desc_cache if k[0] == kind]
        if len(stale) >= keep:
            for k in stale:
                del self._desc_cache[k]

    def _prep_weights(self, P, names, tdtype, need_t, lo):
        ws = [P[n + '.weight'] for n in names]
        dev = ws[0].device
        dt = _DT[tdtype]
        key = ('prep', tuple((w.data_ptr(), tuple(w.shape)) for w in ws), dt, dev.index)
        with self._lock:
            ent = self._desc_cache.get(key)
            if ent is None:
                offs, rows, off = [], [], 0
                for w in ws:
                    N, K = w.shape
                    rows.append([w.data_ptr(), 0, 0, N, K])
                    offs.append(off)
                    off += N * K
                self._desc_room('prep')
                ent = dict(desc=torch.tensor(rows, dtype=torch.int64).to(dev),
                           offs=(torch.tensor(offs, dtype=torch.int64) * torch.empty(0, dtype=tdtype).element_size()).to(dev),
                           offs_host=offs, total=off, max_n=max(w.shape[0] for w in ws), max_k=max(w.shape[1] for w in ws))
                self._desc_cache[key] = ent
        for w in ws:
            if w.dtype != torch.float32 or not w.is_contiguous():
                raise RuntimeError('libmbx: parameters must be contiguous fp32')
        desc = ent['desc'].clone()
        make_n = tdtype != torch.float32          # fp32 mode reads the parameters in place
        flat_n = torch.empty(ent['total'], dtype=tdtype, device=dev) if make_n else None
        flat_t = torch.empty(ent['total'], dtype=tdtype, device=dev) if need_t else None
        if make_n:
            desc[:, 1] = ent['offs'] + flat_n.data_ptr()
        if need_t:
            desc[:, 2] = ent['offs'] + flat_t.data_ptr()
        if make_n or need_t:
            self._ck(self.lib.mbx_prep_weights(desc.data_ptr(), len(ws), ent['max_n'], ent['max_k'], MBX_BF16_LO if lo else dt, self._stream()))
        Wn, Wt = {}, {}
        for n, w, off in zip(names, ws, ent['offs_host']):
            N, K = w.shape
            Wn[n] = flat_n[off:off + N * K].view(N, K) if make_n else w.detach()
            if need_t:
                Wt[n] = flat_t[off:off + N * K].view(K, N)
        return Wn, Wt

    # ------------------------------------------------------------------ LayerNorm folding (bf16 path)
    @staticmethod
    def can_fold(tdtype, cfg) -> bool:
        """The folded-LayerNorm kernels exist for bf16 operands; every kernel-side shape constraint of the set is mirrored HERE, because
        the decision is taken before the forward (which then drops the fp32 sub-layer inputs) and a refusal in backward would come too
        late: GELU' row dots per 64-column block and the 64-wide k tiles of the LayerNorm-backward GEMM (hidden % 64, C % 64: K of the dX
        GEMMs is 3 C / hidden, N is C), attention row dots only in the bf16 kernels with head dim 32 / 64 (sequence length <= 256 is a limit
        of every attention kernel and raises in forward already)."""
        return tdtype == torch.bfloat16 and cfg.hidden % 64 == 0 and cfg.C % 64 == 0 and cfg.hd in (32, 64)

    def fold_norm_weights(self, P: Dict[str, torch.Tensor], pairs, need_t: bool, tdtype=torch.bfloat16):
        """For every (linear, norm) pair: Wn[linear] = bf16(W diag(gamma)) [N,K], Wt[linear] = its transpose [K,N] (need_t),
        Bf[linear] = b + W beta (fp32 [N]), Rs[linear] = row sums of the rounded folded weights (fp32 [N]).  One call, two launches."""
        if tdtype != torch.bfloat16:
            raise RuntimeError('libmbx: fold_norm_weights is a bf16 path')
        ws = [P[l + '.weight'] for l, _ in pairs]
        dev = ws[0].device
        key = ('fold', tuple((w.data_ptr(), tuple(w.shape), P[n + '.weight'].data_ptr(), P[n + '.bias'].data_ptr(),
                              0 if P.get(l + '.bias') is None else P[l + '.bias'].data_ptr()) for (l, n), w in zip(pairs, ws)), dev.index)
        with self._lock:
            ent = self._desc_cache.get(key)
            if ent is None:
                rows, offs, voffs, off, voff = [], [], [], 0, 0
                for (l, n), w in zip(pairs, ws):
                    N, K = w.shape
                    b = P.get(l + '.bias')
                    for t in (w, P[n + '.weight'], P[n + '.bias']) + ((b,) if b is not None else ()):
                        if t.dtype != torch.float32 or not t.is_contiguous():
                            raise RuntimeError('libmbx: parameters must be contiguous fp32')
                    rows.append([w.data_ptr(), 0 if b is None else b.data_ptr(), P[n + '.weight'].data_ptr(), P[n + '.bias'].data_ptr(),
                                 0, 0, 0, 0, N, K])
                    offs.append(off)
                    voffs.append(voff)
                    off += N * K
                    voff += N
                self._desc_room('fold')
                ent = dict(desc=torch.tensor(rows, dtype=torch.int64).to(dev), offs=(torch.tensor(offs, dtype=torch.int64) * 2).to(dev),
                           voffs=(torch.tensor(voffs, dtype=torch.int64) * 4).to(dev), offs_host=offs, voffs_host=voffs, total=off,
                           vtotal=voff, max_n=max(w.shape[0] for w in ws), max_k=max(w.shape[1] for w in ws))
                self._desc_cache[key] = ent
        desc = ent['desc'].clone()
        flat_n = torch.empty(ent['total'], dtype=torch.bfloat16, device=dev)
        flat_t = torch.empty(ent['total'], dtype=torch.bfloat16, device=dev) if need_t else None
        flat_b = torch.empty(ent['vtotal'], dtype=torch.float32, device=dev)
        flat_r = torch.empty(ent['vtotal'], dtype=torch.float32, device=dev)
        desc[:, 4] = ent['offs'] + flat_n.data_ptr()
        if need_t:
            desc[:, 5] = ent['offs'] + flat_t.data_ptr()
        desc[:, 6] = ent['voffs'] + flat_b.data_ptr()
        desc[:, 7] = ent['voffs'] + flat_r.data_ptr()
        self._ck(self.lib.mbx_fold_norm_weights(desc.data_ptr(), len(ws), ent['max_n'], ent['max_k'], self._stream()))
        Wn, Wt, Bf, Rs = {}, {}, {}, {}
        for (l, _), w, off, voff in zip(pairs, ws, ent['offs_host'], ent['voffs_host']):
            N, K = w.shape
            Wn[l] = flat_n[off:off + N * K].view(N, K)
            if need_t:
                Wt[l] = flat_t[off:off + N * K].view(K, N)
            Bf[l], Rs[l] = flat_b[voff:voff + N], flat_r[voff:voff + N]
        return Wn, Wt, Bf, Rs

    def gemm_nt_dgelu_stats(self, a_t, w_t, out_t, aux_t, bias_f, rsum, part):
        M, K = a_t.shape
        N = w_t.shape[0]
        self._ck(self.lib.mbx_gemm_nt_dgelu_stats(_p(a_t), _p(w_t), _p(out_t), _p(aux_t), _p(bias_f), _p(rsum), _p(part), M, N, K, self._stream()))

    # round 5: fc1 + GELU with the derivative saved for backward instead of the pre-activation, and the one-multiply backward epilogue
    @staticmethod
    def can_gelu_d(tdtype, cfg) -> bool:
        return tdtype == torch.bfloat16 and cfg.hidden >= 256 and cfg.hidden % 8 == 0 and cfg.C % 64 == 0

    def gemm_nt_gelu_d(self, a_t, w_t, bias, out_d, out_g):
        M, K = a_t.shape
        N = w_t.shape[0]
        self._ck(self.lib.mbx_gemm_nt_gelu_d(_p(a_t), _p(w_t), _p(bias), _p(out_d), _p(out_g), M, N, K, self._stream()))

    def gemm_nt_mul(self, a_t, w_t, aux_t, out_t):
        M, K = a_t.shape
        N = w_t.shape[0]
        self._ck(self.lib.mbx_gemm_nt_mul(_p(a_t), _p(w_t), _p(aux_t), _p(out_t), M, N, K, self._stream()))

    def attn_bwd_stats(self, qkv, o, do, lse, dqkv, bias_f, rsum, part, B, T, J, H, scale, mode):
        hd = o.shape[-1] // H
        self._ck(self.lib.mbx_attn_bwd_stats(_p(qkv), _p(o), _p(do), _p(lse), _p(dqkv), _p(bias_f), _p(rsum), _p(part), B, T, J, H, hd,
                                             float(scale), int(mode), self._stream()))

    def lnbwd_rowc(self, part, rstd, rowc, C):
        nb, M = part.shape[0], part.shape[1]          # block-major partial row dots [nb, M, 2]
        self._ck(self.lib.mbx_lnbwd_rowc(_p(part), nb, _p(rstd), _p(rowc), M, int(C), self._stream()))

    def gemm_nt_lnbwd(self, a_t, w_t, xhat, rowc, dres, extra, dx, dx_t):
        """dres fp32: dx fp32 (+ optional bf16 copy dx_t).  dres bf16 (the gradient stream inside a Block): dx and dx_t both optional."""
        M, K = a_t.shape
        N = w_t.shape[0]
        fn = self.lib.mbx_gemm_nt_lnbwd_t if dres.dtype == torch.bfloat16 else self.lib.mbx_gemm_nt_lnbwd
        self._ck(fn(_p(a_t), _p(w_t), _p(xhat), _p(rowc), _p(dres), _p(extra), _p(dx), _p(dx_t), M, N, K, self._stream()))

    # ------------------------------------------------------------------ raw-operand LayerNorm + fused MLP (bf16 no-grad path)
    @staticmethod
    def can_fuse_mlp(tdtype, cfg) -> bool:
        """The fused MLP forward (mbx_mlp_fused_fwd) and the raw-operand GEMMs exist for bf16, C in {256, 512}."""
        return tdtype == torch.bfloat16 and cfg.C in (256, 512) and cfg.hidden % 64 == 0 and cfg.hidden <= 1536

    def mlp_pack_weights(self, w1_t, w2_t):
        """fc1 [hidden, C] and fc2 [C, hidden] (bf16) -> the MFMA-fragment stream mbx_mlp_fused_fwd consumes."""
        hidden, Cc = w1_t.shape
        packed = torch.empty(int(self.lib.mbx_mlp_pack_bytes(Cc, hidden)), dtype=torch.uint8, device=w1_t.device)
        self._ck(self.lib.mbx_mlp_pack_weights(_p(w1_t), _p(w2_t), _p(packed), Cc, hidden, self._stream()))
        return packed

    def proj_mlp_pack_weights(self, wp_t, w1_t, w2_t):
        """attn.proj [C, C], fc1 [hidden, C] (folded) and fc2 [C, hidden] (bf16) -> the fragment stream of mbx_proj_mlp_fused_fwd."""
        hidden, Cc = w1_t.shape
        packed = torch.empty(int(self.lib.mbx_proj_mlp_pack_bytes(Cc, hidden)), dtype=torch.uint8, device=w1_t.device)
        self._ck(self.lib.mbx_proj_mlp_pack_weights(_p(wp_t), _p(w1_t), _p(w2_t), _p(packed), Cc, hidden, self._stream()))
        return packed

    def proj_mlp_fused_fwd(self, o_t, packed, bp, b1, b2, rsum, resid, y, eps):
        """y = y1 + fc2(gelu(fc1(LayerNorm(y1)))), y1 = resid + o . Wp^T + bp: attention proj + residual + the whole MLP sub-layer, one kernel."""
        M, Cc = resid.shape
        self._ck(self.lib.mbx_proj_mlp_fused_fwd(_p(o_t), _p(packed), _p(bp), _p(b1), _p(b2), _p(rsum), _p(resid), _p(y), float(eps), M, Cc,
                                                 b1.shape[0], self._stream()))

    def mlp_fused_fwd(self, a_t, raw_in, packed, b1, b2, rsum, resid, y, y_t, eps, mean, rstd):
        """a_t = None: the raw operand is made in the kernel from the fp32 rows of `resid`."""
        M, Cc = resid.shape
        self._ck(self.lib.mbx_mlp_fused_fwd(_p(a_t), int(bool(raw_in)), _p(packed), _p(b1), _p(b2), _p(rsum), _p(resid), _p(y), _p(y_t),
                                            float(eps), _p(mean), _p(rstd), M, Cc, b1.shape[0], self._stream()))

    # ------------------------------------------------------------------ row-owner GEMMs (gemm_rows.hip)
    def rows_pack_nk(self, w_t):
        """w [N, K] bf16 (K in {256, 512}) -> the MFMA-fragment stream of the K-resident row-owner GEMM."""
        N, K = w_t.shape
        packed = torch.empty(int(self.lib.mbx_rows_pack_bytes(N, K)), dtype=torch.uint8, device=w_t.device)
        self._ck(self.lib.mbx_rows_pack_nk(_p(w_t), _p(packed), N, K, self._stream()))
        return packed

    def rows_gemm_nk(self, a_t, packed, bias, out_t, rsum=None, mean=None, rstd=None):
        """out_t bf16 [M,N] = a . w^T + bias, or with (rsum, mean, rstd) the raw-operand LayerNorm form out = rstd (a . w^T - mean rsum) + bias."""
        M, K = a_t.shape
        self._ck(self.lib.mbx_rows_gemm_nk(_p(a_t), _p(packed), _p(bias), _p(rsum), _p(mean), _p(rstd), _p(out_t), M, out_t.shape[1], K,
                                           self._stream()))

    def rows_gemm_nk_ln(self, x, packed, bias, rsum, eps, out_t):
        """out_t bf16 [M,N] = Linear'(LayerNorm(x)) straight from the fp32 rows x [M,K]: operand bf16(x) and the row statistics are made
        in the kernel (no LayerNorm pass, no bf16 copy of the residual stream)."""
        M, K = x.shape
        self._ck(self.lib.mbx_rows_gemm_nk_ln(_p(x), _p(packed), _p(bias), _p(rsum), float(eps), _p(out_t), M, out_t.shape[1], K, self._stream()))

    # ------------------------------------------------------------------ N-resident row-owner GEMM + LayerNorm backward (round 5)
    @staticmethod
    def can_rows_lnbwd(tdtype, cfg) -> bool:
        """mbx_rows_lnbwd_t / mbx_rows_resid_ln exist for bf16, dim_feat 512 (contraction lengths C, 3 C, hidden: multiples of 256, >= 512) and
        -- round 6 -- dim_feat 256 (MotionBERT-Lite: multiples of 256, >= 256); anything else falls back to the tile kernels."""
        return (tdtype == torch.bfloat16 and cfg.C in (256, 512) and cfg.hidden % 256 == 0 and cfg.hidden >= (512 if cfg.C == 512 else 256))

    def rows_n_pack(self, w_t):
        """w bf16 [N, K], N = 512 or 256 (the operand of the dX GEMM) in the fragment order of mbx_rows_lnbwd_t / mbx_rows_resid_ln."""
        return self.rows_n_pack_many([w_t])[0]

    def rows_n_pack_many(self, ws):
        """rows_n_pack of every operand of the list in one launch; the results are views into one buffer.  The descriptor table lives on the
        device and is cached by the operands' offsets from the first one: no host-to-device copy in the steady state (nor inside a graph
        capture after a warm-up step); source and destination addresses are filled in on the device."""
        if not ws:
            return []
        dev = ws[0].device
        N = ws[0].shape[0]
        for w in ws:
            if w.shape[0] != N or N not in (256, 512) or w.shape[1] < N or w.shape[1] % 256 or w.dtype != torch.bfloat16 or not w.is_contiguous():
                raise RuntimeError(f'rows_n_pack_many: operand {tuple(w.shape)} {w.dtype} (bf16 [N, K], one N = 256 or 512 for all, K % 256 == 0, '
                                   f'K >= N, contiguous)')
        # The descriptor table is cached by the operands' offsets from the first one -- stable from step to step when they are views into
        # ONE flat buffer (prep_weights / fold_norm_weights make them so).  Operands of separate allocations get absolute addresses in
        # the key instead: still correct, just a cache entry per set of addresses.
        base = ws[0].data_ptr()
        store = ws[0].untyped_storage().data_ptr()
        if any(w.untyped_storage().data_ptr() != store for w in ws):
            base = 0
        key = ('rnpack', base == 0, N, tuple((w.data_ptr() - base, w.shape[1]) for w in ws), dev.index)
        with self._lock:
            ent = self._desc_cache.get(key)
            if ent is None:
                offs, off = [], 0
                for w in ws:
                    offs.append(off)
                    off += N * w.shape[1] * 2
                self._desc_room('rnpack')
                ent = dict(desc=torch.tensor([[w.data_ptr() - base, o, w.shape[1]] for w, o in zip(ws, offs)], dtype=torch.int64).to(dev), offs=offs, total=off,
                           max_k=max(w.shape[1] for w in ws))
                self._desc_cache[key] = ent
        flat = torch.empty(ent['total'], dtype=torch.uint8, device=dev)
        desc = ent['desc'].clone()
        desc[:, 0] += base
        desc[:, 1] += flat.data_ptr()
        self._ck(self.lib.mbx_rows_n_pack_many(desc.data_ptr(), len(ws), N, ent['max_k'], self._stream()))
        return [flat[o:o + N * w.shape[1] * 2] for w, o in zip(ws, ent['offs'])]

    def rows_lnbwd_t(self, dy_t, packed, xhat, rstd, dres_t, dx_t):
        """dx_t = T(dres_t + LayerNorm'(dy . w^T)) with both row means taken in the kernel (no row dots from the producers of dy)."""
        M, K = dy_t.shape
        self._ck(self.lib.mbx_rows_lnbwd_t(_p(dy_t), _p(packed), _p(xhat), _p(rstd), _p(dres_t), _p(dx_t), M, dx_t.shape[1], K, self._stream()))

    can_rows_resid_ln = can_rows_lnbwd      # same shape constraints (dim_feat 512 or 256; contraction lengths C and hidden)

    def rows_resid_ln(self, a_t, packed, bias, resid, y, xhat, mean, rstd, eps):
        """y = resid + a . w^T + bias (fp32) and the plain LayerNorm of its rows (xhat in the operand type, mean, rstd) in one kernel."""
        M, K = a_t.shape
        self._ck(self.lib.mbx_rows_resid_ln(_p(a_t), _p(packed), _p(bias), _p(resid), _p(y), _p(xhat), _p(mean), _p(rstd), float(eps), M, y.shape[1], K,
                                            self._stream()))

    # ------------------------------------------------------------------ measurement aid (bench.py, tools/clock_power.py)
    def mfma_probe(self, seconds: float = 0.25, wgs_per_cu: int = 1, device=None):
        """The bf16 MFMA rate the part sustains with nothing but MFMAs in the loop (random operands), run for about `seconds` so
        that the power management has settled: dict(tflops, clock_ghz, ms).  `clock_ghz` = shader cycles / real time inside the
        kernel (median over workgroups)."""
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        n_wg = cus * wgs_per_cu
        ws = torch.zeros(int(self.lib.mbx_mfma_probe_ws(n_wg)), dtype=torch.uint8, device=dev)
        flops = C.c_double(0.0)

        def run(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._ck(self.lib.mbx_mfma_probe(_p(ws), n_wg, iters, 1234, C.byref(flops), self._stream()))
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1)
        ms = run(20000)                                            # calibrate, then one long launch
        iters = max(20000, int(20000 * seconds * 1e3 / max(ms, 1e-3)))
        ms = run(iters)
        st = ws[n_wg * 1024:].view(torch.int64).view(n_wg, 2).cpu().double()
        clock = float((st[:, 0] / (st[:, 1] * 1e-8)).median()) / 1e9
        return dict(tflops=flops.value / ms / 1e9, clock_ghz=clock, ms=ms, iters=iters, wgs_per_cu=wgs_per_cu)

    def unfold_norm_grads(self, dw, db, w, gamma, beta, dgamma, dbeta):
        N, K = dw.shape
        ws = self._ws(('unf', N, K), self.lib.mbx_unfold_norm_grads_ws, N, K, device=dw.device)
        self._ck(self.lib.mbx_unfold_norm_grads(_p(dw), _p(db), _p(w), _p(gamma), _p(beta), _p(dgamma), _p(dbeta), N, K, _p(ws), self._stream()))

    # ------------------------------------------------------------------ embedding
    def embed_fwd(self, x, w, b, pos, temp, h, B, T, J):
        self._ck(self.lib.mbx_embed_fwd(_p(x), _p(w), _p(b), _p(pos), _p(temp), _p(h), B, T, J, x.shape[-1], h.shape[-1],
                                        self._stream()))

    def embed_bwd(self, dh, x, w, dw, db, dpos, dtemp, dx, B, T, J):
        Din, Cc = x.shape[-1], dh.shape[-1]
        ws = self._ws(('emb', T, J, Din, Cc), self.lib.mbx_embed_bwd_ws, T, J, Din, Cc, device=dh.device)
        dtemp.zero_()
        self._ck(self.lib.mbx_embed_bwd(_p(dh), _p(x), _p(w), _p(dw), _p(db), _p(dpos), _p(dtemp), _p(dx), B, T, J, Din, Cc,
                                        _p(ws), self._stream()))

    def embed_bwd_pair(self, dh_a, dh_b, x, w, dw, db, dpos, dtemp, dx, B, T, J):
        """embed_bwd with the incoming gradient as the sum of two bf16 tensors (the two Blocks' input gradients of level 0)."""
        Din, Cc = x.shape[-1], dh_a.shape[-1]
        ws = self._ws(('emb', T, J, Din, Cc), self.lib.mbx_embed_bwd_ws, T, J, Din, Cc, device=dh_a.device)
        dtemp.zero_()
        self._ck(self.lib.mbx_embed_bwd_pair(_p(dh_a), _p(dh_b), _p(x), _p(w), _p(dw), _p(db), _p(dpos), _p(dtemp), _p(dx), B, T, J, Din, Cc,
                                             _p(ws), self._stream()))

    # ------------------------------------------------------------------ layernorm
    layernorm_fwd_planes_ok = True      # the producers below accept a (hi, lo) pair of bf16 planes where a bf16x3 GEMM operand is due

    def layernorm_fwd(self, x, g, b, eps, y_t, mean, rstd):
        M, Cc = x.shape
        if isinstance(y_t, tuple):      # bf16x3: the operand planes straight from the kernel
            self._ck(self.lib.mbx_layernorm_fwd_planes(_p(x), _p(g), _p(b), float(eps), _p(y_t[0]), _p(y_t[1]), _p(mean), _p(rstd), M, Cc,
                                                       self._stream()))
            return
        self._ck(self.lib.mbx_layernorm_fwd(_p(x), _p(g), _p(b), float(eps), _p(y_t), _p(mean), _p(rstd), M, Cc,
                                            _DT[y_t.dtype], self._stream()))

    def layernorm_bwd(self, dy_t, x, mean, rstd, g, dres, extra, dx, dx_t, dg, db):
        M, Cc = x.shape
        ws = self._ws(('lnb', Cc), self.lib.mbx_layernorm_bwd_ws, Cc, device=x.device)
        if isinstance(dx_t, tuple):      # bf16x3: the T copy of dx as operand planes
            self._ck(self.lib.mbx_layernorm_bwd_planes(_p(dy_t), _p(x), _p(mean), _p(rstd), _p(g), _p(dres), _p(extra), _p(dx), _p(dx_t[0]),
                                                       _p(dx_t[1]), _p(dg), _p(db), M, Cc, _p(ws), self._stream()))
            return
        self._ck(self.lib.mbx_layernorm_bwd(_p(dy_t), _p(x), _p(mean), _p(rstd), _p(g), _p(dres), _p(extra), _p(dx), _p(dx_t),
                                            _p(dg), _p(db), M, Cc, _DT[dy_t.dtype], _p(ws), self._stream()))

    # ------------------------------------------------------------------ GEMMs
    def gelu_fwd(self, u, g):
        self._ck(self.lib.mbx_gelu_fwd(_p(u), _p(g), u.numel(), _DT[u.dtype], self._stream()))

    def split(self, t):
        """fp32 tensor -> (hi, lo) bf16 planes of the bf16x3 split (t = hi + lo up to 2^-16 relative)."""
        hi, lo = torch.empty_like(t, dtype=torch.bfloat16), torch.empty_like(t, dtype=torch.bfloat16)
        self._ck(self.lib.mbx_split_bf16(_p(t), _p(hi), _p(lo), t.numel(), self._stream()))
        return hi, lo

    def gemm_nt(self, a_t, w_t, bias, epi, out_t=None, out2_t=None, out_f=None, resid=None, aux_t=None):
        if isinstance(a_t, tuple):       # bf16x3: (hi, lo) operand planes, fp32 T-typed outputs
            (ah, al), (wh, wl) = a_t, w_t
            M, K = ah.shape
            N = wh.shape[0]
            if wh.shape[1] != K:
                raise RuntimeError(f'libmbx: gemm_nt_x3 operand mismatch {tuple(ah.shape)} x {tuple(wh.shape)}')
            planes = out2_t if isinstance(out2_t, tuple) else (out_t if isinstance(out_t, tuple) else None)
            if planes is not None:       # the last output leaves as the (hi, lo) planes of the operand split
                self._ck(self.lib.mbx_gemm_nt_x3p(_p(ah), _p(al), _p(wh), _p(wl), _p(bias), int(epi), _p(None if planes is out_t else out_t),
                                                  _p(planes[0]), _p(planes[1]), _p(aux_t), M, N, K, self._stream()))
                return
            self._ck(self.lib.mbx_gemm_nt_x3(_p(ah), _p(al), _p(wh), _p(wl), _p(bias), int(epi), _p(out_t), _p(out2_t), _p(out_f),
                                             _p(resid), _p(aux_t), M, N, K, self._stream()))
            return
        M, K = a_t.shape
        N = w_t.shape[0]
        if w_t.shape[1] != K or a_t.dtype != w_t.dtype:
            raise RuntimeError(f'libmbx: gemm_nt operand mismatch {tuple(a_t.shape)} {a_t.dtype} x {tuple(w_t.shape)} {w_t.dtype}')
        self._ck(self.lib.mbx_gemm_nt(_p(a_t), _p(w_t), _p(bias), int(epi), _p(out_t), _p(out2_t), _p(out_f), _p(resid),
                                      _p(aux_t), M, N, K, _DT[a_t.dtype], self._stream()))

    def gemm_tn(self, dy_t, a_t, dw, db):
        if isinstance(dy_t, tuple):      # bf16x3
            (yh, yl), (ah, al) = dy_t, a_t
            M, N = yh.shape
            K = ah.shape[1]
            ws = self._ws(('tnx3', M, N, K), self.lib.mbx_gemm_tn_x3_workspace, M, N, K, device=yh.device)
            self._ck(self.lib.mbx_gemm_tn_x3(_p(yh), _p(yl), _p(ah), _p(al), _p(dw), _p(db), M, N, K, _p(ws), self._stream()))
            return
        M, N = dy_t.shape
        K = a_t.shape[1]
        ws = self._ws(('tn', M, N, K), self.lib.mbx_gemm_tn_ws, M, N, K, device=dy_t.device)
        self._ck(self.lib.mbx_gemm_tn(_p(dy_t), _p(a_t), _p(dw), _p(db), M, N, K, _DT[dy_t.dtype], _p(ws), self._stream()))

    # ------------------------------------------------------------------ attention
    def attn_fwd(self, qkv, o, lse, B, T, J, H, scale, mode, drop=None):
        """drop = (p, seed): nn.Dropout(p) on the probabilities inside the kernel (counter-based mask, dropmask.py)."""
        hd = o.shape[-1] // H
        if drop is not None and drop[0] > 0:
            self._ck(self.lib.mbx_attn_fwd_drop(_p(qkv), _p(o), _p(lse), B, T, J, H, hd, float(scale), int(mode), _DT[qkv.dtype],
                                                float(drop[0]), int(drop[1]), self._stream()))
            return
        self._ck(self.lib.mbx_attn_fwd(_p(qkv), _p(o), _p(lse), B, T, J, H, hd, float(scale), int(mode), _DT[qkv.dtype],
                                       self._stream()))

    def attn_bwd(self, qkv, o, do, lse, dqkv, B, T, J, H, scale, mode, drop=None):
        hd = o.shape[-1] // H
        if isinstance(dqkv, tuple):      # bf16x3 (fp32 kernels): dq / dk / dv as operand planes
            p, seed = (float(drop[0]), int(drop[1])) if drop is not None and drop[0] > 0 else (0.0, 0)
            self._ck(self.lib.mbx_attn_bwd_planes(_p(qkv), _p(o), _p(do), _p(lse), _p(dqkv[0]), _p(dqkv[1]), B, T, J, H, hd, float(scale),
                                                  int(mode), p, seed, self._stream()))
            return
        if drop is not None and drop[0] > 0:
            self._ck(self.lib.mbx_attn_bwd_drop(_p(qkv), _p(o), _p(do), _p(lse), _p(dqkv), B, T, J, H, hd, float(scale), int(mode),
                                                _DT[qkv.dtype], float(drop[0]), int(drop[1]), self._stream()))
            return
        self._ck(self.lib.mbx_attn_bwd(_p(qkv), _p(o), _p(do), _p(lse), _p(dqkv), B, T, J, H, hd, float(scale), int(mode),
                                       _DT[qkv.dtype], self._stream()))

    # ------------------------------------------------------------------ fusion
    def fuse_fwd(self, x_st, x_ts, w, b, out, alpha):
        M, Cc = x_st.shape
        self._ck(self.lib.mbx_fuse_fwd(_p(x_st), _p(x_ts), _p(w), _p(b), _p(out), _p(alpha), M, Cc, self._stream()))

    def fuse_ln_fwd(self, x_st, x_ts, w, b, out, alpha, g1, b1, xn1, g2, b2, xn2, eps, mean, rstd):
        M, Cc = x_st.shape
        self._ck(self.lib.mbx_fuse_ln_fwd(_p(x_st), _p(x_ts), _p(w), _p(b), _p(out), _p(alpha), _p(g1), _p(b1), _p(xn1), _p(g2), _p(b2),
                                          _p(xn2), float(eps), _p(mean), _p(rstd), M, Cc, _DT[xn1.dtype], self._stream()))

    def fuse_bwd(self, dh, x_st, x_ts, alpha, w, d_st, d_ts, d_st_t, d_ts_t, dw, db):
        M, Cc = x_st.shape
        ws = self._ws(('fub', Cc), self.lib.mbx_fuse_bwd_ws, Cc, device=dh.device)
        self._ck(self.lib.mbx_fuse_bwd(_p(dh), _p(x_st), _p(x_ts), _p(alpha), _p(w), _p(d_st), _p(d_ts), _p(d_st_t), _p(d_ts_t),
                                       _p(dw), _p(db), M, Cc, _DT[d_st_t.dtype], _p(ws), self._stream()))

    def fuse_bwd_pair(self, dh_a, dh_b, x_st, x_ts, alpha, w, d_st_t, d_ts_t, dw, db):
        """fuse_bwd with the incoming gradient as the sum of two bf16 tensors (the two Blocks' input gradients of the level above)."""
        M, Cc = x_st.shape
        ws = self._ws(('fub', Cc), self.lib.mbx_fuse_bwd_ws, Cc, device=x_st.device)
        self._ck(self.lib.mbx_fuse_bwd_pair(_p(dh_a), _p(dh_b), _p(x_st), _p(x_ts), _p(alpha), _p(w), _p(d_st_t), _p(d_ts_t), _p(dw), _p(db), M, Cc,
                                            _p(ws), self._stream()))

    def average(self, x_st, x_ts, out):
        self._ck(self.lib.mbx_average(_p(x_st), _p(x_ts), _p(out), x_st.numel(), self._stream()))

    def average_bwd(self, dh, d_st, d_ts, d_st_t, d_ts_t):
        self._ck(self.lib.mbx_average_bwd(_p(dh), _p(d_st), _p(d_ts), _p(d_st_t), _p(d_ts_t), dh.numel(), _DT[d_st_t.dtype],
                                          self._stream()))

    # ------------------------------------------------------------------ tail
    def head_fwd(self, rep, w, b, out):
        M, R = rep.shape
        self._ck(self.lib.mbx_head_fwd(_p(rep), _p(w), _p(b), _p(out), M, R, w.shape[0], self._stream()))

    def head_bwd(self, dout, rep, w, dpre_t, dw, db):
        M, R = rep.shape
        D = w.shape[0]
        ws = self._ws(('hdb', R, D), self.lib.mbx_head_bwd_ws, R, D, device=rep.device)
        self._ck(self.lib.mbx_head_bwd(_p(dout), _p(rep), _p(w), _p(dpre_t), _p(dw), _p(db), M, R, D, _DT[dpre_t.dtype], _p(ws),
                                       self._stream()))

    def tanh_bwd(self, drep, rep, dpre_t):
        self._ck(self.lib.mbx_tanh_bwd(_p(drep), _p(rep), _p(dpre_t), rep.numel(), _DT[dpre_t.dtype], self._stream()))


    # ------------------------------------------------------------------ input stage
    def augment2d(self, x, y, noise, uniform_range, jitter_std, d2c, mask_ratio, mask_T_ratio, flags, seed):
        B, T, J, Cin = x.shape
        mean, std, weight = noise if noise is not None else (None, None, None)
        a, b, m, s = d2c
        self._ck(self.lib.mbx_augment2d(_p(x), _p(y), B, T, J, Cin, _p(mean), _p(std), _p(weight), float(uniform_range), float(jitter_std),
                                        float(a), float(b), float(m), float(s), float(mask_ratio), float(mask_T_ratio), int(flags),
                                        int(seed), self._stream()))

    def embed_fwd_tta(self, x, perm, w, b, pos, temp, h, B, T, J):
        self._ck(self.lib.mbx_embed_fwd_tta(_p(x), _p(perm), _p(w), _p(b), _p(pos), _p(temp), _p(h), B, T, J, x.shape[-1], h.shape[-1],
                                            self._stream()))

    def flip_average(self, out2, perm, out):
        B, T, J, D = out.shape
        self._ck(self.lib.mbx_flip_average(_p(out2), _p(perm), _p(out), B, T, J, D, self._stream()))

    # ------------------------------------------------------------------ dropout / drop-path
    def dropout(self, x, y, p, seed):
        self._ck(self.lib.mbx_dropout(_p(x), _p(y), x.numel(), float(p), int(seed), _DT[x.dtype], self._stream()))

    def residual_drop(self, y, x, rows_per_sample, p, seed, p_path, seed_path):
        rows, Cc = y.shape
        self._ck(self.lib.mbx_residual_drop(_p(y), _p(x), rows, Cc, int(rows_per_sample), float(p), int(seed), float(p_path),
                                            int(seed_path), self._stream()))

    def grad_drop(self, dy, dy_t, rows_per_sample, p, seed, p_path, seed_path):
        rows, Cc = dy.shape
        self._ck(self.lib.mbx_grad_drop(_p(dy), _p(dy_t), rows, Cc, int(rows_per_sample), float(p), int(seed), float(p_path),
                                        int(seed_path), _DT[dy_t.dtype], self._stream()))

    # ------------------------------------------------------------------ ActionNet pooling
    def pool_rep_fwd(self, rep, pooled, N, Mp, T, J, p=0.0, seed=0):
        self._ck(self.lib.mbx_pool_rep_fwd(_p(rep), _p(pooled), N, Mp, T, J, rep.shape[-1], float(p), int(seed), self._stream()))

    def tanh_pool_bwd(self, dpooled, rep, dpre_t, N, Mp, T, J, p=0.0, seed=0):
        self._ck(self.lib.mbx_tanh_pool_bwd(_p(dpooled), _p(rep), _p(dpre_t), N, Mp, T, J, rep.shape[-1], float(p), int(seed),
                                            _DT[dpre_t.dtype], self._stream()))

    # ------------------------------------------------------------------ training step
    def pose_loss(self, pred, gt, lambda_scale, lambda_velocity, losses, dpred, grad_scale=1.0):
        B, T, J, D = pred.shape
        if D != 3 or gt.shape != pred.shape:
            raise RuntimeError(f'libmbx: pose_loss needs pred, gt [B,T,J,3], got {tuple(pred.shape)} / {tuple(gt.shape)}')
        ws = self._ws(('pl', B, T), self.lib.mbx_pose_loss_ws, B, T, device=pred.device)
        self._ck(self.lib.mbx_pose_loss(_p(pred), _p(gt), float(lambda_scale), float(lambda_velocity), _p(losses), _p(dpred),
                                        float(grad_scale), B, T, J, _p(ws), self._stream()))

    def loss_2d_weighted(self, pred, target, conf, loss, dpred, grad_scale=1.0):
        """pred [B,T,J,3]; target [B,T,J,>=2] (x, y first) and conf [B,T,J] or [B,T,J,1] may be strided VIEWS of one [B,T,J,3]
        2D batch (target = batch, conf = batch[..., 2]): only their last-dimension element stride is passed down."""
        B, T, J, D = pred.shape
        if D != 3 or tuple(target.shape[:3]) != (B, T, J) or target.shape[-1] < 2 or conf.numel() != B * T * J:
            raise RuntimeError(f'libmbx: loss_2d_weighted needs pred [B,T,J,3], target [B,T,J,>=2], conf [B,T,J(,1)], got '
                               f'{tuple(pred.shape)} / {tuple(target.shape)} / {tuple(conf.shape)}')
        cf = conf.reshape(B, T, J) if conf.dim() == 4 else conf
        ts, cs = target.stride(2), cf.stride(2)
        if (target.stride(3) != 1 or target.stride(1) != J * ts or target.stride(0) != T * J * ts or
                cf.stride(1) != J * cs or cf.stride(0) != T * J * cs or target.dtype != torch.float32 or cf.dtype != torch.float32):
            raise RuntimeError('libmbx: loss_2d_weighted: target / conf must be fp32 and dense over (B, T, J) with a constant per-joint stride')
        ws = self._ws(('l2d', B, T), self.lib.mbx_loss_2d_weighted_ws, B, T, device=pred.device)
        self._ck(self.lib.mbx_loss_2d_weighted(_p(pred), _p(target), int(ts), _p(cf), int(cs), _p(loss), _p(dpred), float(grad_scale),
                                               B, T, J, _p(ws), self._stream()))

    def adamw_step(self, p, g, m, v, state, beta1, beta2, eps, weight_decay, tick=True):
        self._ck(self.lib.mbx_adamw_step(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(state), float(beta1), float(beta2), float(eps),
                                         float(weight_decay), int(bool(tick)), self._stream()))


_OPS: Optional[HipOps] = None
_OPS_LOCK = threading.Lock()


def get() -> HipOps:
    """Process-wide HipOps instance; raises if libmbx.so is missing (no fallback)."""
    global _OPS
    if _OPS is None:
        with _OPS_LOCK:
            if _OPS is None:
                _OPS = HipOps()
    return _OPS


def peek() -> Optional[HipOps]:
    """The process-wide instance if one has been created, else None (never loads the library: model.train() on a CPU box)."""
    return _OPS
