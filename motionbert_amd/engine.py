"""Host-side sequencing of the DSTformer hot path over the HIP kernel set.

This module owns no arithmetic.  It walks the dual-stream block structure of
the reference model (`lib/model/DSTformer.py:329-358` forward, `:239-249`
Block) and issues one call per kernel on an `ops` object whose methods map
1:1 onto the C-ABI entry points declared in `include/mbx.h`.  The product
`ops` is `motionbert_amd.hip_ops.HipOps` (ctypes -> libmbx.so -> gfx950
kernels); there is no CPU implementation in the package.

Data layout in HBM (all row-major, token index m = (b*T + t)*J + j):
  residual stream   [M, C]  fp32   one fresh buffer per sub-layer output
  GEMM operands     [M, K]  T      T = compute dtype (bf16 or fp32)
  qkv               [M, 3C] T      channel order [3][H][hd]  (DSTformer.py:143)
  attention out     [M, C]  T      heads concatenated head-major
  softmax stats     [M, H]  fp32   log-sum-exp per (token, head)
  LN stats          [M]     fp32   mean, rstd
The temporal attention reads its (b, j, h) sequences straight out of the
[B, T, J, 3C] qkv tensor with a row stride of J*3C elements: the three
permute->contiguous copies of `DSTformer.py:190-192` do not exist here.

What is saved for backward (per sub-layer).  Folded sequencing (bf16 default): the plain normalisation xhat (T), rstd, and
qkv / o / lse (attention) or the pre-/post-GELU hidden (MLP); the fp32 sub-layer input is NOT kept.  Plain sequencing (fp32-class
modes, dropout): the fp32 input, LN mean / rstd, the normalised T-typed GEMM input and the same.  Attention probabilities are never
stored: the backward kernels recompute them from q, k and lse.

No-grad sequencing (bf16, `rawln`): nothing is saved, and inside a Block no LayerNorm pass, no bf16 copy of the residual stream and
no hidden tensor exist.  Both consumers of a LayerNorm read the fp32 rows of the residual stream themselves, round them to the bf16
operand and take the row statistics from the same loads: the attention's qkv Linear as a row-owner GEMM that applies the LayerNorm
as row constants in its epilogue (mbx_rows_gemm_nk_ln), the MLP as ONE kernel (mbx_mlp_fused_fwd).  4 launches per sub-layer pair
(qkv, attention, proj + residual, MLP) instead of 7.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, ClassVar, Dict, List, Optional

import os

import torch

# GEMM epilogues (values mirror enum mbx_epilogue in include/mbx.h)
EPI_STORE = 0      # out_t  = acc + bias                      (T)
EPI_GELU = 1       # out_t  = acc + bias ; out2_t = gelu(out) (T, T)
EPI_RESID = 2      # out_f  = resid + acc + bias              (fp32)
EPI_TANH = 3       # out_f  = tanh(acc + bias)                (fp32)
EPI_DGELU = 4      # out_t  = acc * gelu'(aux_t)              (T)

MODE_SPATIAL = 0
MODE_TEMPORAL = 1


@dataclass(frozen=True)
class ModelCfg:
    dim_in: int
    dim_out: int
    C: int          # dim_feat
    R: int          # dim_rep
    depth: int
    H: int
    hidden: int
    J: int
    maxlen: int
    eps: float
    scale: float
    att_fuse: bool
    qkv_bias: bool
    drop: float = 0.0            # drop_rate: pos_drop, proj_drop, both MLP drops (DSTformer.py:77,104,278)
    attn_drop: float = 0.0       # attn_drop_rate: on the attention probabilities (DSTformer.py:96)
    dpr: tuple = ()              # DropPath rate per level, linspace(0, drop_path_rate, depth) (DSTformer.py:279)

    @property
    def hd(self) -> int:
        return self.C // self.H


@dataclass(frozen=True)
class Switches:
    """The A/B switches of the sequencing: one `MBX_*` environment variable each, every one ON by default except the two marked.
    They are read ONCE, when this module is imported (`SWITCHES`); a test or a tool that flips one afterwards calls
    `reload_switches()` (or patches `engine.SWITCHES` with `Switches.from_env({...})`).  What each one selects is described where
    `Engine.__init__` consumes it."""
    x3_planes: bool       # MBX_X3_PLANES      bf16x3: producers write the operand planes themselves (0: every operand through mbx_split_bf16)
    dual_stream: bool     # MBX_DUAL_STREAM    the ts Block of a level on a second HIP stream
    wgrad_stream: bool    # MBX_WGRAD_STREAM   (default OFF) weight-gradient GEMMs on a third stream
    fold_ln: bool         # MBX_FOLD_LN        LayerNorm folded into the Linear it feeds (bf16)
    fold_dx_first: bool   # MBX_FOLD_ORDER     (default OFF) the dX GEMM before the weight gradient
    grad_stream: bool     # MBX_GRAD_STREAM    gradient of the residual stream in the operand type between sub-layers
    rows_lnbwd: bool      # MBX_ROWS_LNBWD     row-owner LayerNorm-backward GEMM
    rows_resid_ln: bool   # MBX_ROWS_RESID_LN  row-owner residual GEMM + next LayerNorm
    block_grad_t: bool    # MBX_BLOCK_GRAD_T   ... and across Block boundaries
    gelu_d: bool          # MBX_GELU_D         fc1 saves gelu'(u) instead of u
    rawln: bool           # MBX_RAWLN          no-grad sequencing (raw-operand LayerNorm + fused MLP)
    proj_mlp: bool        # MBX_PROJ_MLP       no-grad: proj + residual inside the MLP kernel

    _ENV: ClassVar[tuple] = (('x3_planes', 'MBX_X3_PLANES', '1'), ('dual_stream', 'MBX_DUAL_STREAM', '1'), ('wgrad_stream', 'MBX_WGRAD_STREAM', '0'),
            ('fold_ln', 'MBX_FOLD_LN', '1'), ('fold_dx_first', 'MBX_FOLD_ORDER', '0'), ('grad_stream', 'MBX_GRAD_STREAM', '1'),
            ('rows_lnbwd', 'MBX_ROWS_LNBWD', '1'), ('rows_resid_ln', 'MBX_ROWS_RESID_LN', '1'), ('block_grad_t', 'MBX_BLOCK_GRAD_T', '1'),
            ('gelu_d', 'MBX_GELU_D', '1'), ('rawln', 'MBX_RAWLN', '1'), ('proj_mlp', 'MBX_PROJ_MLP', '1'))

    @classmethod
    def from_env(cls, env=None) -> 'Switches':
        env = os.environ if env is None else env
        return cls(**{field: env.get(var, default) == '1' for field, var, default in cls._ENV})


SWITCHES = Switches.from_env()


def reload_switches(env=None) -> Switches:
    """Re-read the MBX_* switches (tests and A/B tools that change the environment after import)."""
    global SWITCHES
    SWITCHES = Switches.from_env(env)
    return SWITCHES


# sub-layer order of the two Block flavours (DSTformer.py:240-249)
ORDER = {
    'st': (('attn', 'norm1_s', 'attn_s', MODE_SPATIAL), ('mlp', 'norm2_s', 'mlp_s', None),
           ('attn', 'norm1_t', 'attn_t', MODE_TEMPORAL), ('mlp', 'norm2_t', 'mlp_t', None)),
    'ts': (('attn', 'norm1_t', 'attn_t', MODE_TEMPORAL), ('mlp', 'norm2_t', 'mlp_t', None),
           ('attn', 'norm1_s', 'attn_s', MODE_SPATIAL), ('mlp', 'norm2_s', 'mlp_s', None)),
}


def linear_names(cfg: ModelCfg) -> List[str]:
    """Every nn.Linear whose weight runs through the MFMA GEMM (prefix without '.weight')."""
    names = []
    for stream in ('blocks_st', 'blocks_ts'):
        for i in range(cfg.depth):
            for a in ('attn_s', 'attn_t'):
                names += [f'{stream}.{i}.{a}.qkv', f'{stream}.{i}.{a}.proj']
            for m in ('mlp_s', 'mlp_t'):
                names += [f'{stream}.{i}.{m}.fc1', f'{stream}.{i}.{m}.fc2']
    names.append('pre_logits.fc')
    return names


def folded_pairs(cfg: ModelCfg) -> List[tuple]:
    """(linear, norm) pairs whose LayerNorm is folded into the Linear it feeds (bf16 path): every norm1 -> qkv and norm2 -> fc1
    of every Block (DSTformer.py:241-249).  The final `norm` -> pre_logits.fc keeps the plain kernels."""
    pairs = []
    for stream in ('blocks_st', 'blocks_ts'):
        for i in range(cfg.depth):
            for sfx in ('s', 't'):
                pairs.append((f'{stream}.{i}.attn_{sfx}.qkv', f'{stream}.{i}.norm1_{sfx}'))
                pairs.append((f'{stream}.{i}.mlp_{sfx}.fc1', f'{stream}.{i}.norm2_{sfx}'))
    return pairs


def grad_bucket(name: str, depth: int) -> int:
    """Bucket of a parameter in backward completion order: tail first, then the levels from the
    last to the first, the embedding last.  The flat gradient buffer is laid out bucket by bucket."""
    if name.startswith(('head.', 'pre_logits.', 'norm.')):
        return 0
    if name.startswith(('blocks_st.', 'blocks_ts.', 'ts_attn.')):
        return depth - int(name.split('.')[1])
    return depth + 1


class Engine:
    """One forward (and optionally backward) pass.  `P` maps reference state_dict
    names to fp32 device tensors; `Wn[name]` / `Wt[name]` are the T-typed [N,K]
    and transposed [K,N] copies produced by ops.prep_weights."""

    _side_streams: Dict[int, Any] = {}

    def __init__(self, ops, cfg: ModelCfg, P: Dict[str, torch.Tensor], tdtype: torch.dtype, x3: bool = False, drop_seed=None):
        self.ops, self.cfg, self.P, self.T = ops, cfg, P, tdtype
        sw = SWITCHES      # the A/B switches, read once at import (class Switches)
        # Dropout / DropPath: active only in training with a rate > 0 -- `drop_seed` is then the base seed of this
        # forward pass (None = everything off, the case of every shipped config).  Masks are counter-based (dropmask.py): the
        # element-wise ones and DropPath run as three small kernels around the fused path, the attention-probability dropout
        # inside the attention kernels (round 3: the mask is a hash of the element's index in the reference's attn tensor).
        self.drop_seed = drop_seed
        # low-memory mode (model.recompute): what only feeds GEMMs in backward -- the LayerNorm output of every sub-layer and the
        # MLP's post-activation -- is rebuilt there (one LayerNorm-forward / one GELU launch each) instead of being kept: 14 -> 12
        # bytes per residual element for an attention sub-layer, 14 -> 6 for an MLP; the full model then trains at 256 clips x 243
        # frames inside the 288 GB of one MI355X.
        self.recompute = False
        # precision 'bf16x3': T-typed tensors are fp32; a tensor that feeds a GEMM is split into (hi, lo) bf16 planes first
        # (`_mm`), and where it ONLY feeds GEMMs (LayerNorm output, GELU output) the planes are what is kept for backward
        self.x3 = x3
        self.x3_planes = sw.x3_planes and hasattr(ops, 'layernorm_fwd_planes_ok')      # A/B switch: 0 = every operand through mbx_split_bf16
        self.Wn: Dict[str, torch.Tensor] = {}
        self.Wt: Dict[str, torch.Tensor] = {}
        # The st and ts blocks of a level are independent (DSTformer.py:341-342 feeds both the same x): with
        # MBX_DUAL_STREAM=1 the ts block runs on a second HIP stream so that HBM-bound kernels of one stream
        # (LayerNorm, GEMM epilogues) overlap MFMA-bound kernels of the other.
        self.dual = sw.dual_stream and getattr(ops, 'multi_stream', False)
        # weight-gradient GEMMs feed nothing downstream in backward: MBX_WGRAD_STREAM=1 issues them on a third stream.  Off by
        # default: 137.0 -> 136.0 ms per step at 64 clips, but the operands stay alive until that stream catches up
        # (record_stream), which at 256 clips (231 GiB resident) sends the allocator into retries: 421 -> 37 clips/s.
        self.wgrad_async = sw.wgrad_stream
        # LayerNorm folding (round 3, bf16 path; include/mbx.h "LayerNorm folded into the Linear it feeds"): the LayerNorm kernels
        # write the plain normalisation xhat, the affine part lives in the qkv / fc1 weights, and the LayerNorm BACKWARD runs as the
        # epilogue of the dX GEMM from row dots the attention-backward / GELU' kernels emit -- 40 LayerNorm-backward launches and the
        # saved fp32 sub-layer inputs disappear.  Off with dropout (the masks sit between the producer and the row dots), in the
        # fp32-class modes (bf16 kernels only), or by request (model.fold_ln = False / MBX_FOLD_LN=0: the A/B switch).
        self.fold = (sw.fold_ln and not x3 and drop_seed is None and
                     bool(getattr(ops, 'can_fold', lambda *_: False)(tdtype, cfg)))
        self.fold_dx_first = sw.fold_dx_first
        # Gradient residual stream in the operand type BETWEEN the four sub-layers of a Block (round 4; fp32 at the Block boundaries,
        # fp32 arithmetic in the kernels): the folded LayerNorm-backward GEMM reads its dres as bf16 and writes ONLY the bf16 dx, which
        # is the stream and the next GEMMs' operand at once -- 4 instead of 12 bytes per element and launch.  Numerics:
        # tools/gradstream_numerics.py / profiles/r04_gradstream_numerics.txt (every gate unchanged).  MBX_GRAD_STREAM=0: the A/B switch.
        self.gstream_allowed = sw.grad_stream and bool(getattr(ops, 'grad_stream_t', False))
        self.gstream = False      # decided per backward (the folded sequencing only)
        self.Bf: Dict[str, torch.Tensor] = {}
        self.Rs: Dict[str, torch.Tensor] = {}
        # Round 5: inside a Block (bf16 gradient stream in and out, no second summand) the folded LayerNorm backward runs as the
        # epilogue of a row-owner GEMM that takes both row means from its own accumulators (mbx_rows_lnbwd_t): the producers of dY --
        # attention backward, the GELU' GEMM -- run WITHOUT their row dots there, and no row-constant launch is needed.  The first
        # sub-layer of a Block (fp32 out, the other block's gradient added) keeps the tile kernel.  MBX_ROWS_LNBWD=0: the A/B switch.
        self.rows_lnbwd = (sw.rows_lnbwd and bool(getattr(ops, 'can_rows_lnbwd', lambda *_: False)(tdtype, cfg)))
        self.Pn: Dict[str, torch.Tensor] = {}       # transposed folded weights in the fragment order of mbx_rows_lnbwd_t
        # Round 5: the FORWARD residual GEMM (proj / fc2) of a sub-layer that is followed by a LayerNorm runs on the same row-owner shape
        # and leaves the plain normalisation of its output rows with them (mbx_rows_resid_ln): three of the four LayerNorm launches of a
        # Block -- each a second read of the fp32 rows -- disappear.  Folded sequencing only (the kernel writes xhat, not gamma xhat +
        # beta), no dropout on the branch.  MBX_ROWS_RESID_LN=0: the A/B switch.
        self.rows_resid_ln = (sw.rows_resid_ln and bool(getattr(ops, 'can_rows_resid_ln', lambda *_: False)(tdtype, cfg)))
        self.Pf: Dict[str, torch.Tensor] = {}       # proj / fc2 weights of those sub-layers in the fragment order of the row-owner kernels
        # Round 5: the gradient of the residual stream stays in the operand type ACROSS Block boundaries too: the first
        # sub-layer of a Block then also takes the row-owner LayerNorm backward (no row dots, no row constants, bf16 out), and the fusion
        # backward of the level below reads the two Blocks' input gradients as a bf16 pair and adds them (same bytes as one fp32 tensor).
        # Numerics: four realisations on both reference-minted fixtures, every frozen gate (profiles/r05_boundary_numerics.txt).
        # The embedding backward reads level 0's pair the same way (mbx_embed_bwd_pair).  MBX_BLOCK_GRAD_T=0: the A/B switch.
        self.block_grad_t = sw.block_grad_t and self.rows_lnbwd and hasattr(ops, 'fuse_bwd_pair') and hasattr(ops, 'embed_bwd_pair')
        # Round 5: with the row means taken by that kernel the GELU' epilogue no longer has to produce the dot of
        # du with the pre-activation, so fc1's forward epilogue saves gelu'(u) -- taken from the fp32 accumulator -- INSTEAD of u (same
        # bytes) and the backward epilogue is one multiply (mbx_gemm_nt_gelu_d / mbx_gemm_nt_mul).  MBX_GELU_D=0: the A/B switch.
        self.gelu_d = (sw.gelu_d and self.rows_lnbwd and bool(getattr(ops, 'can_gelu_d', lambda *_: False)(tdtype, cfg)))
        # no-grad sequencing of a Block (decided per forward): raw-operand LayerNorm + fused MLP, see the module docstring
        self.rawln = False
        self.rawln_allowed = sw.rawln      # A/B switch: 0 = the training sequencing without saves
        self.Pk: Dict[str, torch.Tensor] = {}       # fc1 / fc2 of every MLP (with the proj in front of it) in the fragment order of the fused kernels
        self.Pk_proj: Dict[str, str] = {}           # MLP -> the proj Linear packed in front of it (derived from ORDER)
        self.proj_mlp = sw.proj_mlp and hasattr(ops, 'proj_mlp_fused_fwd')      # A/B switch: 0 = proj + residual as its own GEMM

    def _streams(self):
        """(main, side) streams for the dual-stream schedule, or (None, None)."""
        if not (self.dual and self.dev.type == 'cuda'):
            return None, None
        idx = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        side = Engine._side_streams.get(idx)
        if side is None:
            side = Engine._side_streams[idx] = torch.cuda.Stream(device=idx)
        return torch.cuda.current_stream(idx), side

    _w_streams: Dict[int, Any] = {}

    def _wstream(self):
        if not (self.wgrad_async and self.dev.type == 'cuda'):
            return None
        idx = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        ws = Engine._w_streams.get(idx)
        if ws is None:
            ws = Engine._w_streams[idx] = torch.cuda.Stream(device=idx)
        return ws

    def _tn(self, dy_t, a_t, dw, db):
        """dW / db GEMM, optionally on the weight-gradient stream (call it BEFORE the dX GEMM of the same dy)."""
        ws = self._wstream()
        if ws is None:
            return self.ops.gemm_tn(dy_t, a_t, dw, db)
        ws.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ws):
            self.ops.gemm_tn(dy_t, a_t, dw, db)
        for t in (dy_t if isinstance(dy_t, tuple) else (dy_t,)) + (a_t if isinstance(a_t, tuple) else (a_t,)):
            t.record_stream(ws)

    def _join_wgrads(self):
        ws = self._wstream()
        if ws is not None:
            torch.cuda.current_stream().wait_stream(ws)

    # ------------------------------------------------------------------ helpers
    def _f(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.dev)

    def _t(self, *shape):
        return torch.empty(shape, dtype=self.T, device=self.dev)

    def _bias(self, name):
        return self.P.get(name + '.bias')

    def _drops(self, pre: str, sub: int):
        """Active dropout of sub-layer `sub` of block `pre` ('blocks_st.3'): None, or (p, seed_branch, seed_act, p_path, seed_path,
        p_attn, seed_attn)."""
        if self.drop_seed is None:
            return None
        from .dropmask import site_seed
        cfg = self.cfg
        stream, level = (0 if pre.startswith('blocks_st') else 1), int(pre.split('.')[1])
        pp = float(cfg.dpr[level]) if cfg.dpr else 0.0
        if cfg.drop <= 0 and pp <= 0 and cfg.attn_drop <= 0:
            return None
        ss = lambda kind: site_seed(self.drop_seed, level, stream, sub, kind)
        return (cfg.drop, ss(1), ss(2), pp, ss(3), cfg.attn_drop, ss(0))

    def _xn(self, sv, pre, norm):
        """The normalised GEMM operand of a sub-layer in backward: the saved one, or (recompute mode) LayerNorm of the saved input."""
        if sv['xn'] is not None:
            return sv['xn']
        cfg, P = self.cfg, self.P
        xn, mean, rstd = self._op(self.M, cfg.C), self._f(self.M), self._f(self.M)
        self.ops.layernorm_fwd(sv['x'], P[f'{pre}.{norm}.weight'], P[f'{pre}.{norm}.bias'], cfg.eps, xn, mean, rstd)
        return self._mm(xn)

    def _mm(self, t):
        """GEMM operand form of a T-typed tensor: itself, or its (hi, lo) bf16 planes in bf16x3 mode (already planes: unchanged)."""
        return self.ops.split(t) if self.x3 and not isinstance(t, tuple) else t

    def _op(self, *shape):
        """A tensor that is ONLY read as a GEMM operand: T-typed, or -- bf16x3 -- the pair of bf16 planes its producer writes directly
        (no fp32 copy for mbx_split_bf16 to read again)."""
        if self.x3 and self.x3_planes:
            return (torch.empty(shape, dtype=torch.bfloat16, device=self.dev), torch.empty(shape, dtype=torch.bfloat16, device=self.dev))
        return self._t(*shape)

    def _weight_cache_key(self, need_grad: bool):
        """Key of the prepared-weight cache of the no-grad path: the parameters' storage and version counters (every
        in-place update -- optimizer.step(), FlatAdamW, load_state_dict -- bumps them) and everything that selects a format.  None =
        do not cache: a backward follows (training re-prepares every step anyway), or a hipGraph is being captured (the re-pack has
        to be part of the graph so that a replay picks up weights updated in place, graph.py).  Edits through `param.data` bypass the
        version counters (as they bypass autograd's own checks); what covers the ordinary train-then-eval loop is that every
        need_grad forward, `model.train()` and `load_state_dict()` drop the cache (prepare_weights, model.py).  An edit through
        `.data` in a pure inference process still needs `hip_ops.get().weight_cache.clear()`."""
        if need_grad or not hasattr(self.ops, 'weight_cache') or self.dev.type != 'cuda' or torch.cuda.is_current_stream_capturing():
            return None
        return (tuple((p.data_ptr(), p._version) for p in self.P.values()), self.T, self.fold, self.rawln, self.proj_mlp, self.x3, self.rows_resid_ln)

    def prepare_weights(self, need_grad: bool):
        """T-typed copies of every Linear weight (and their transposes when a backward follows); with LayerNorm folding the
        qkv / fc1 weights are W diag(gamma) and come with their folded bias and row sums.  Under no_grad the prepared set is kept
        on the kernel provider and reused as long as no parameter changed: an inference call then launches no fold / pack kernels
        (at depth 5 they were ~60 small launches per forward, as many as a B = 1 forward has kernels of its own)."""
        cfg, ops, P = self.cfg, self.ops, self.P
        if need_grad and hasattr(ops, 'weight_cache') and self.dev.type == 'cuda':
            # a training step follows: whatever it does to the parameters -- also through `p.data`, which bumps no version counter
            # (legacy optimizers, EMA code, a replayed graph) -- the next no-grad forward must not find this forward's prepared set
            # (ADVICE r5).  model.train() / load_state_dict() drop it as well (model.py).
            ops.weight_cache.pop(self.dev.index, None)
        key = self._weight_cache_key(need_grad)
        if key is not None:
            hit = ops.weight_cache.get(self.dev.index)
            # same storage + same versions is not enough: a NEW model built after the old one died gets the same addresses and the
            # same (small) version numbers from the allocator -- the entry must be about these very tensor objects
            if hit is not None and hit[0] == key and len(hit[1]) == len(P) and all(r() is p for r, p in zip(hit[1], P.values())):
                self.Wn, self.Wt, self.Bf, self.Rs, self.Pk, self.Pk_proj, self.Pf = hit[2]
                return
        self._prepare_weights(need_grad)
        if key is not None:
            import weakref
            ops.weight_cache[self.dev.index] = (key, [weakref.ref(p) for p in P.values()], (self.Wn, self.Wt, self.Bf, self.Rs, self.Pk, self.Pk_proj, self.Pf))

    def _prepare_weights(self, need_grad: bool):
        cfg, ops, P = self.cfg, self.ops, self.P
        if self.fold:
            pairs = folded_pairs(cfg)
            folded = {l for l, _ in pairs}
            self.Wn, self.Wt = ops.prep_weights(P, [n for n in linear_names(cfg) if n not in folded], self.T, need_grad)
            fn, ft, self.Bf, self.Rs = ops.fold_norm_weights(P, pairs, need_grad, self.T)
            self.Wn.update(fn)
            self.Wt.update(ft)
            if self.rows_resid_ln and not self.rawln:
                # proj / fc2 of the first three sub-layers of every Block (the fourth feeds the fusion, not a LayerNorm)
                lins = [f'{stream}.{i}.{m}.' + ('proj' if typ == 'attn' else 'fc2') for stream, kind in (('blocks_st', 'st'), ('blocks_ts', 'ts'))
                        for i in range(cfg.depth) for typ, _norm, m, _mode in ORDER[kind][:-1]]
                self.Pf = dict(zip(lins, ops.rows_n_pack_many([self.Wn[lin] for lin in lins])))
            if need_grad and self.rows_lnbwd and self.gstream_allowed:
                # (not for the first sub-layer of a Block whose input gradient leaves in fp32 with the other stream's gradient added --
                # MBX_BLOCK_GRAD_T=0: the tile kernel's epilogue)
                first = {f'{stream}.{i}.{ORDER[kind][0][2]}.qkv' for stream, kind in (('blocks_st', 'st'), ('blocks_ts', 'ts')) for i in range(cfg.depth)
                         if not (self.block_grad_t and cfg.att_fuse)}
                lins = [lin for lin, _ in pairs if lin not in first]
                self.Pn = dict(zip(lins, ops.rows_n_pack_many([ft[lin] for lin in lins])))
            if self.rawln:
                for stream, kind in (('blocks_st', 'st'), ('blocks_ts', 'ts')):
                    for i in range(cfg.depth):
                        order = ORDER[kind]
                        for sub, (typ, _norm, m, _mode) in enumerate(order):
                            if typ != 'mlp':
                                continue
                            pre = f'{stream}.{i}.{m}'
                            # the proj that runs in front of this MLP in the same kernel: the attention right before it in ORDER
                            a = order[sub - 1][2] if sub > 0 and order[sub - 1][0] == 'attn' else None
                            if self.proj_mlp and a is not None:
                                self.Pk_proj[pre] = f'{stream}.{i}.{a}.proj'
                                self.Pk[pre] = ops.proj_mlp_pack_weights(self.Wn[self.Pk_proj[pre]], self.Wn[pre + '.fc1'], self.Wn[pre + '.fc2'])
                            else:
                                self.Pk[pre] = ops.mlp_pack_weights(self.Wn[pre + '.fc1'], self.Wn[pre + '.fc2'])
                        for a in ('attn_s', 'attn_t'):
                            lin = f'{stream}.{i}.{a}.qkv'
                            self.Pk[lin] = ops.rows_pack_nk(self.Wn[lin])
        else:
            self.Wn, self.Wt = (ops.prep_weights(P, linear_names(cfg), self.T, need_grad, x3=True) if self.x3 else
                                ops.prep_weights(P, linear_names(cfg), self.T, need_grad))

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, return_rep, need_grad: bool, tta_perm=None):
        """`return_rep`: False (pose output), True (representation, DSTformer.py:360) or a tuple ('pool', persons, p, seed):
        the ActionNet pooling of model_action.py:15-24 fused onto the representation (mean over persons and frames of the
        dropped-out representation -> [B / persons, J, R])."""
        cfg, ops, P = self.cfg, self.ops, self.P
        pool = return_rep if isinstance(return_rep, tuple) else None
        return_rep = bool(return_rep)
        self.dev = x.device
        B, T, J, Din = x.shape
        if tta_perm is not None:      # flip test-time augmentation: the batch doubles, the second half is the flipped view
            B = 2 * B
        M, C = B * T * J, cfg.C
        self.B, self.Tlen, self.M = B, T, M
        self.rawln = (not need_grad and self.fold and self.drop_seed is None and self.rawln_allowed and
                      bool(getattr(ops, 'can_fuse_mlp', lambda *_: False)(self.T, cfg)))
        self.prepare_weights(need_grad)
        h = self._f(M, C)
        if tta_perm is not None:
            ops.embed_fwd_tta(x, tta_perm, P['joints_embed.weight'], P['joints_embed.bias'], P['pos_embed'], P['temp_embed'], h, B // 2, T, J)
        else:
            ops.embed_fwd(x, P['joints_embed.weight'], P['joints_embed.bias'], P['pos_embed'], P['temp_embed'], h, B, T, J)
        if self.drop_seed is not None and cfg.drop > 0:      # pos_drop (DSTformer.py:337)
            from .dropmask import site_seed
            ops.dropout(h, h, cfg.drop, site_seed(self.drop_seed, -1, 0, 0, 1))
        saved: Dict[str, Any] = dict(x=x, levels=[], return_rep=return_rep)
        main, side = self._streams()
        fuse_ln = cfg.att_fuse and hasattr(ops, 'fuse_ln_fwd')
        ln_st = ln_ts = ln_tail = None      # (xn, mean, rstd) of h already produced by the fusion kernel of the previous level
        if self.fold and not self.rawln:
            # folded: both Blocks of a level read the SAME plain normalisation of h (their gamma / beta live in their own weights) -- one
            # launch for level 0 too, as the fusion kernel provides one xhat for the later levels
            xn0, mean0, rstd0 = self._op(M, C), self._f(M), self._f(M)
            ops.layernorm_fwd(h, None, None, cfg.eps, xn0, mean0, rstd0)
            ln_st = ln_ts = (xn0, mean0, rstd0)
            del xn0, mean0, rstd0
        for i in range(cfg.depth):
            if side is not None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    x_ts, sv_ts = self._block_fwd(h, f'blocks_ts.{i}', 'ts', need_grad, ln_ts)
                x_st, sv_st = self._block_fwd(h, f'blocks_st.{i}', 'st', need_grad, ln_st)
                main.wait_stream(side)
            else:
                x_st, sv_st = self._block_fwd(h, f'blocks_st.{i}', 'st', need_grad, ln_st)
                x_ts, sv_ts = self._block_fwd(h, f'blocks_ts.{i}', 'ts', need_grad, ln_ts)
            ln_st = ln_ts = None      # consumed; the fusion kernel below may provide the next level's
            hn = self._f(M, C)
            if fuse_ln and not (self.rawln and i + 1 < cfg.depth):      # (no-grad: the next level's consumers read the fp32 rows themselves)
                # the fusion kernel also normalises its output for its consumers: the first LayerNorm of both blocks of the next
                # level (one mean / rstd for the two), or the final `norm` after the last level
                alpha, mean, rstd = self._f(M, 2), self._f(M), self._f(M)
                last = i + 1 == cfg.depth
                n1 = 'norm' if last else f"blocks_st.{i + 1}.{ORDER['st'][0][1]}"
                n2 = None if last else f"blocks_ts.{i + 1}.{ORDER['ts'][0][1]}"
                if self.fold and not last:      # ONE plain normalisation serves both blocks of the next level
                    xn1 = self._t(M, C)
                    ops.fuse_ln_fwd(x_st, x_ts, P[f'ts_attn.{i}.weight'], P[f'ts_attn.{i}.bias'], hn, alpha,
                                    None, None, xn1, None, None, None, cfg.eps, mean, rstd)
                    ln_st = ln_ts = (xn1, mean, rstd)
                else:
                    xn1, xn2 = self._t(M, C), (None if last else self._t(M, C))
                    ops.fuse_ln_fwd(x_st, x_ts, P[f'ts_attn.{i}.weight'], P[f'ts_attn.{i}.bias'], hn, alpha,
                                    P[n1 + '.weight'], P[n1 + '.bias'], xn1, P[n2 + '.weight'] if n2 else None, P[n2 + '.bias'] if n2 else None, xn2,
                                    cfg.eps, mean, rstd)
                    if last:
                        ln_tail = (xn1, mean, rstd)
                    else:
                        ln_st, ln_ts = (xn1, mean, rstd), (xn2, mean, rstd)
            elif cfg.att_fuse:
                alpha = self._f(M, 2)
                ops.fuse_fwd(x_st, x_ts, P[f'ts_attn.{i}.weight'], P[f'ts_attn.{i}.bias'], hn, alpha)
            else:
                alpha = None
                ops.average(x_st, x_ts, hn)
            if need_grad:
                saved['levels'].append(dict(st=sv_st, ts=sv_ts, x_st=x_st, x_ts=x_ts, alpha=alpha))
            h = hn
        if ln_tail is not None:
            xn, mean, rstd = ln_tail
        else:
            xn, mean, rstd = self._op(M, C), self._f(M), self._f(M)
            ops.layernorm_fwd(h, P['norm.weight'], P['norm.bias'], cfg.eps, xn, mean, rstd)
        xn = self._mm(xn)
        # the returned tensor is allocated in its final 4-D shape (the kernels see [M, .] views of it): autograd refuses
        # in-place writes into an output that is itself a view (callers do `out[:, :, 0, :] = 0`, train.py:76)
        rep4 = self._f(B, T, J, cfg.R)
        rep = rep4.view(M, cfg.R)
        ops.gemm_nt(xn, self.Wn['pre_logits.fc'], P['pre_logits.fc.bias'], EPI_TANH, out_f=rep)
        if need_grad:
            saved.update(h=h, xn=xn, mean=mean, rstd=rstd, rep=rep, pool=pool)
        if pool is not None:
            _, persons, p_drop, seed = pool
            if B % persons:
                raise RuntimeError(f'pooled representation: batch {B} is not a multiple of persons={persons}')
            out = self._f(B // persons, J, cfg.R)
            ops.pool_rep_fwd(rep, out, B // persons, persons, T, J, p_drop, seed)
        elif return_rep:
            out = rep4
        else:
            out = self._f(B, T, J, cfg.dim_out)
            ops.head_fwd(rep, P['head.weight'], P['head.bias'], out.view(M, cfg.dim_out))
            if tta_perm is not None:      # flip back + average (train.py:70-72)
                both, out = out, self._f(B // 2, T, J, cfg.dim_out)
                ops.flip_average(both, tta_perm, out)
        return out, saved

    def _block_fwd(self, x, pre, kind, need_grad, ln=None):
        """`ln` = (xn, mean, rstd) of x when its producer already normalised it (the fusion kernel of the previous level, for the
        first sub-layer of a Block); never anything else."""
        svs = []
        order = ORDER[kind]
        pend = None      # no-grad: an attention whose proj + residual has been left to the MLP kernel that follows: dict(o=, proj=)
        for sub, (typ, norm, mod, mode) in enumerate(order):
            nxt = order[sub + 1][1] if sub + 1 < len(order) else None      # the norm that reads this sub-layer's output
            if typ == 'attn':
                defer = self.rawln and self.proj_mlp and sub + 1 < len(order) and order[sub + 1][0] == 'mlp'
                x, sv, ln, pend = self._attn_fwd(x, pre, norm, mod, mode, need_grad, sub, ln, nxt, defer_proj=defer)
            else:
                x, sv, ln = self._mlp_fwd(x, pre, norm, mod, need_grad, sub, ln, nxt, pending_proj=pend)
                pend = None
            svs.append(sv)
        assert pend is None, 'an attention deferred its proj to an MLP that never came'
        return x, svs

    def _resid_gemm(self, a, lin, x, dm, pre, nxt):
        """y = x + a . W^T + b (fp32 residual stream): returns (y, ln) -- ln = (xhat, mean, rstd) of y for the next sub-layer when the
        row-owner kernel ran (it owns whole rows, so LayerNorm's statistics cost it nothing), else None."""
        cfg, ops, P = self.cfg, self.ops, self.P
        y = self._f(self.M, cfg.C)
        # (the kernel addresses its 2-KiB rows with 32-bit offsets: beyond 2^21 rows the tile kernel + LayerNorm take over)
        if self.fold and nxt is not None and lin in self.Pf and self.M < (1 << 21) and not (dm is not None and (dm[0] > 0 or dm[3] > 0)):
            xn, mean, rstd = self._op(self.M, cfg.C), self._f(self.M), self._f(self.M)
            ops.rows_resid_ln(a, self.Pf[lin], P[lin + '.bias'], x, y, xn, mean, rstd, cfg.eps)
            return y, (xn, mean, rstd)
        ops.gemm_nt(a, self.Wn[lin], P[lin + '.bias'], EPI_RESID, resid=x, out_f=y)
        if (not self.rawln) and dm is not None and (dm[0] > 0 or dm[3] > 0):      # proj_drop / MLP drop + DropPath on the branch (DSTformer.py:84,148-149,241-242)
            ops.residual_drop(y, x, cfg.J, dm[0], dm[1], dm[3], dm[4])
        return y, None

    def _attn_fwd(self, x, pre, norm, attn, mode, need_grad, sub=0, ln=None, nxt=None, defer_proj=False):
        cfg, ops, P = self.cfg, self.ops, self.P
        M, C = self.M, cfg.C
        qkv = self._t(M, 3 * C)
        lin = f'{pre}.{attn}.qkv'
        if self.rawln:      # no-grad: LayerNorm + qkv straight from the fp32 rows (operand, statistics and row constants made in the kernel)
            xn = mean = rstd = None
            ops.rows_gemm_nk_ln(x, self.Pk[lin], self.Bf[lin], self.Rs[lin], cfg.eps, qkv)
        else:
            if ln is not None:   # LayerNorm(x) came with x from its producer
                xn, mean, rstd = ln
            else:
                xn, mean, rstd = self._op(M, C), self._f(M), self._f(M)
                if self.fold:
                    ops.layernorm_fwd(x, None, None, cfg.eps, xn, mean, rstd)
                else:
                    ops.layernorm_fwd(x, P[f'{pre}.{norm}.weight'], P[f'{pre}.{norm}.bias'], cfg.eps, xn, mean, rstd)
            xn = self._mm(xn)
            ops.gemm_nt(xn, self.Wn[lin], self.Bf[lin] if self.fold else self._bias(lin), EPI_STORE, out_t=qkv)
        dm = self._drops(pre, sub)
        o, lse = self._t(M, C), self._f(M, cfg.H)
        if dm is not None and dm[5] > 0:      # attn_drop: the counter-based mask is applied to the probabilities inside the kernel
            ops.attn_fwd(qkv, o, lse, self.B, self.Tlen, cfg.J, cfg.H, cfg.scale, mode, drop=(dm[5], dm[6]))
        else:
            ops.attn_fwd(qkv, o, lse, self.B, self.Tlen, cfg.J, cfg.H, cfg.scale, mode)
        if defer_proj:      # no-grad: proj + residual run inside the MLP kernel that follows
            return x, None, None, dict(o=o, proj=f'{pre}.{attn}.proj')
        o_op = self._mm(o)      # (bf16x3: the operand planes are kept for the weight gradient too -- o itself stays fp32 for the attention backward)
        y, ln_y = self._resid_gemm(o_op, f'{pre}.{attn}.proj', x, dm, pre, nxt)
        if self.fold:      # backward needs xhat and rstd only: the fp32 sub-layer input is not kept
            sv = dict(x=None, mean=None, rstd=rstd, xn=xn, qkv=qkv, o=o, lse=lse, dm=dm) if need_grad else None
        else:
            sv = dict(x=x, mean=mean, rstd=rstd, xn=None if self.recompute else xn, qkv=qkv, o=o, lse=lse, dm=dm) if need_grad else None
        if sv is not None and self.x3 and self.x3_planes and not self.recompute:
            sv['o_op'] = o_op
        return y, sv, ln_y, None

    def _mlp_fwd(self, x, pre, norm, mlp, need_grad, sub=1, ln=None, nxt=None, pending_proj=None):
        cfg, ops, P = self.cfg, self.ops, self.P
        M, C = self.M, cfg.C
        if self.rawln:                # no-grad: the whole sub-layer is one kernel; the hidden never reaches HBM, the operand is bf16(x) made in the kernel
            lin = f'{pre}.{mlp}.fc1'
            y = self._f(M, C)
            if pending_proj is not None:      # x + proj(o) first, in the same kernel
                if self.Pk_proj.get(f'{pre}.{mlp}') != pending_proj['proj']:
                    raise RuntimeError(f"{pre}.{mlp}: packed with {self.Pk_proj.get(f'{pre}.{mlp}')} in front, asked to run {pending_proj['proj']}")
                ops.proj_mlp_fused_fwd(pending_proj['o'], self.Pk[f'{pre}.{mlp}'], P[pending_proj['proj'] + '.bias'], self.Bf[lin],
                                       P[f'{pre}.{mlp}.fc2.bias'], self.Rs[lin], x, y, cfg.eps)
            else:
                ops.mlp_fused_fwd(None, True, self.Pk[f'{pre}.{mlp}'], self.Bf[lin], P[f'{pre}.{mlp}.fc2.bias'], self.Rs[lin], x, y, None,
                                  cfg.eps, None, None)
            return y, None, None
        if ln is not None:            # LayerNorm(x) came with x from the residual GEMM of the previous sub-layer
            xn, mean, rstd = ln
        else:
            xn, mean, rstd = self._op(M, C), self._f(M), self._f(M)
            if self.fold:
                ops.layernorm_fwd(x, None, None, cfg.eps, xn, mean, rstd)
            else:
                ops.layernorm_fwd(x, P[f'{pre}.{norm}.weight'], P[f'{pre}.{norm}.bias'], cfg.eps, xn, mean, rstd)
        xn = self._mm(xn)
        dm = self._drops(pre, sub)
        mlp_drop = dm is not None and dm[0] > 0
        # u only feeds GELU' in backward; g is only fc2's operand (and the weight gradient's) unless the MLP dropout touches it first
        u, g = (self._t(M, cfg.hidden) if need_grad else None), (self._t(M, cfg.hidden) if mlp_drop else self._op(M, cfg.hidden))
        # the derivative instead of the pre-activation where backward will run the row-owner tail (every MLP of a Block does: never the
        # first sub-layer, so bf16 gradient in and out and no second summand)
        save_d = bool(need_grad and self.fold and self.gelu_d and self.gstream_allowed and not self.recompute and not mlp_drop
                      and f'{pre}.{mlp}.fc1' in self.Pn and self.M < (1 << 22))
        if save_d:
            ops.gemm_nt_gelu_d(xn, self.Wn[f'{pre}.{mlp}.fc1'], self.Bf[f'{pre}.{mlp}.fc1'], u, g)
        else:
            ops.gemm_nt(xn, self.Wn[f'{pre}.{mlp}.fc1'], self.Bf[f'{pre}.{mlp}.fc1'] if self.fold else P[f'{pre}.{mlp}.fc1.bias'],
                        EPI_GELU, out_t=u, out2_t=g)
        if dm is not None and dm[0] > 0:                      # MLP drop after the activation (DSTformer.py:82)
            ops.dropout(g, g, dm[0], dm[2])
        g = self._mm(g)
        y, ln_y = self._resid_gemm(g, f'{pre}.{mlp}.fc2', x, dm, pre, nxt)
        if self.fold:
            sv = dict(x=None, mean=None, rstd=rstd, xn=xn, u=None if save_d else u, d=u if save_d else None, g=None if self.recompute else g, dm=dm) if need_grad else None
        else:
            sv = dict(x=x, mean=mean, rstd=rstd, xn=None if self.recompute else xn, u=u, g=None if self.recompute else g, dm=dm) if need_grad else None
        return y, sv, ln_y

    # ----------------------------------------------------------------- backward
    def backward(self, saved, dout: torch.Tensor, grads: Dict[str, torch.Tensor], want_dx: bool, on_ready=None):
        """Fills `grads[name]` (fp32, pre-allocated, same shapes as the parameters) and
        returns d(input) or None.  `dout` is d(out) with the shape `forward` returned.
        `on_ready(bucket)` is called as soon as every gradient of a bucket has been enqueued
        (bucket 0 = tail, 1..depth = levels depth-1..0, depth+1 = embedding; see `grad_bucket`):
        the data-parallel wrapper starts that bucket's all-reduce while backward continues."""
        cfg, ops, P = self.cfg, self.ops, self.P
        M, C, R = self.M, cfg.C, cfg.R
        B, T, J = self.B, self.Tlen, cfg.J
        G = self.grads = grads
        self.gstream = self.fold and self.gstream_allowed
        dpre = self._t(M, R)
        if saved.get('pool') is not None:
            _, persons, p_drop, seed = saved['pool']
            ops.tanh_pool_bwd(dout.reshape(B // persons, J, R), saved['rep'], dpre, B // persons, persons, T, J, p_drop, seed)
        elif saved['return_rep']:
            ops.tanh_bwd(dout.reshape(M, R), saved['rep'], dpre)
        if saved['return_rep']:
            if 'head.weight' in G:
                G['head.weight'].zero_()
                G['head.bias'].zero_()
        else:
            ops.head_bwd(dout.reshape(M, cfg.dim_out), saved['rep'], P['head.weight'], dpre,
                         G['head.weight'], G['head.bias'])
        dxn = self._t(M, C)
        dpre = self._mm(dpre)
        self._tn(dpre, saved['xn'], G['pre_logits.fc.weight'], G['pre_logits.fc.bias'])
        ops.gemm_nt(dpre, self.Wt['pre_logits.fc'], None, EPI_STORE, out_t=dxn)
        dh = self._f(M, C)
        ops.layernorm_bwd(dxn, saved['h'], saved['mean'], saved['rstd'], P['norm.weight'],
                          None, None, dh, None, G['norm.weight'], G['norm.bias'])
        del dxn, dpre
        if on_ready is not None:
            self._join_wgrads()
            on_ready(0)
        pair = None      # the two Blocks' input gradients of the level above, T-typed (block_grad_t), instead of their fp32 sum dh
        for i in reversed(range(cfg.depth)):
            lv = saved['levels'][i]
            # (gradient stream in the operand type: the Blocks read the T-typed copies only, the fp32 ones are not even written)
            d_st, d_ts = (None, None) if (self.gstream and cfg.att_fuse) else (self._f(M, C), self._f(M, C))
            d_st_t, d_ts_t = self._t(M, C), self._t(M, C)
            if cfg.att_fuse:
                if pair is not None:
                    ops.fuse_bwd_pair(pair[0], pair[1], lv['x_st'], lv['x_ts'], lv['alpha'], P[f'ts_attn.{i}.weight'],
                                      d_st_t, d_ts_t, G[f'ts_attn.{i}.weight'], G[f'ts_attn.{i}.bias'])
                else:
                    ops.fuse_bwd(dh, lv['x_st'], lv['x_ts'], lv['alpha'], P[f'ts_attn.{i}.weight'],
                                 d_st, d_ts, d_st_t, d_ts_t, G[f'ts_attn.{i}.weight'], G[f'ts_attn.{i}.bias'])
            else:
                ops.average_bwd(dh, d_st, d_ts, d_st_t, d_ts_t)
            pair = dh = None
            # this level's Blocks hand their input gradients down T-typed when the row-owner tail can take their first sub-layer
            out_t = bool(self.block_grad_t and self.gstream and cfg.att_fuse and (i > 0 or self.drop_seed is None) and self.M < (1 << 22)
                         and all(f'{st}.{i}.{ORDER[kind][0][2]}.qkv' in self.Pn for st, kind in (('blocks_st', 'st'), ('blocks_ts', 'ts'))))
            main, side = self._streams()
            if out_t:
                if side is not None:
                    side.wait_stream(main)
                _, a_t = self._block_bwd(d_st, d_st_t, lv['st'], f'blocks_st.{i}', 'st', None, last_needs_t=True)
                if side is not None:
                    with torch.cuda.stream(side):
                        _, b_t = self._block_bwd(d_ts, d_ts_t, lv['ts'], f'blocks_ts.{i}', 'ts', None, last_needs_t=True)
                    main.wait_stream(side)
                    b_t.record_stream(main)
                else:
                    _, b_t = self._block_bwd(d_ts, d_ts_t, lv['ts'], f'blocks_ts.{i}', 'ts', None, last_needs_t=True)
                pair = (a_t, b_t)
                del a_t, b_t
            elif side is not None:
                # the st block runs on the main stream, the ts block on the side stream; the side stream's LAST kernel (the
                # LayerNorm backward at the block input) waits for the main block and adds its input gradient as `extra`
                # -- no separate d1 + d2 pass over the residual stream (1.6 GB per level)
                side.wait_stream(main)
                d1, _ = self._block_bwd(d_st, d_st_t, lv['st'], f'blocks_st.{i}', 'st', None, last_needs_t=False)

                def other(d1=d1, main=main, side=side):
                    side.wait_stream(main)
                    d1.record_stream(side)
                    return d1
                with torch.cuda.stream(side):
                    dh, _ = self._block_bwd(d_ts, d_ts_t, lv['ts'], f'blocks_ts.{i}', 'ts', other, last_needs_t=False)
                main.wait_stream(side)
                dh.record_stream(main)
                del d1
            else:
                d1, _ = self._block_bwd(d_st, d_st_t, lv['st'], f'blocks_st.{i}', 'st', None, last_needs_t=False)
                # the second stream's last LN-backward also adds the first stream's input gradient
                dh, _ = self._block_bwd(d_ts, d_ts_t, lv['ts'], f'blocks_ts.{i}', 'ts', d1, last_needs_t=False)
                del d1
            del d_st, d_st_t, d_ts, d_ts_t
            saved['levels'][i] = None  # release this level's activations
            if on_ready is not None:
                self._join_wgrads()
                on_ready(cfg.depth - i)
        if self.drop_seed is not None and cfg.drop > 0:
            from .dropmask import site_seed
            ops.dropout(dh, dh, cfg.drop, site_seed(self.drop_seed, -1, 0, 0, 1))
        dx = torch.empty_like(saved['x']) if want_dx else None
        if pair is not None:
            ops.embed_bwd_pair(pair[0], pair[1], saved['x'], P['joints_embed.weight'], G['joints_embed.weight'], G['joints_embed.bias'],
                               G['pos_embed'], G['temp_embed'], dx, B, T, J)
        else:
            ops.embed_bwd(dh, saved['x'], P['joints_embed.weight'], G['joints_embed.weight'], G['joints_embed.bias'],
                          G['pos_embed'], G['temp_embed'], dx, B, T, J)
        self._join_wgrads()
        if on_ready is not None:
            on_ready(cfg.depth + 1)
        return dx

    def _block_bwd(self, dy, dy_t, svs, pre, kind, extra_last, last_needs_t):
        order = ORDER[kind]
        for idx in reversed(range(4)):
            typ, norm, mod, mode = order[idx]
            extra = extra_last if idx == 0 else None
            need_t = True if idx > 0 else last_needs_t
            if typ == 'attn':
                dy, dy_t = self._attn_bwd(dy, dy_t, svs[idx], pre, norm, mod, mode, extra, need_t)
            else:
                dy, dy_t = self._mlp_bwd(dy, dy_t, svs[idx], pre, norm, mod, extra, need_t)
            svs[idx] = None
        return dy, dy_t

    def _attn_bwd(self, dy, dy_t, sv, pre, norm, attn, mode, extra, need_t):
        cfg, ops, P, G = self.cfg, self.ops, self.P, self.grads
        M, C = self.M, cfg.C
        do = self._t(M, C)
        dm = sv.get('dm')
        if dm is not None and (dm[0] > 0 or dm[3] > 0):      # gradient entering the dropped branch; the residual path keeps dy
            dy_t = self._t(M, C)
            ops.grad_drop(dy, dy_t, cfg.J, dm[0], dm[1], dm[3], dm[4])
            if self.x3:
                dy_t = self._mm(dy_t)
        elif self.x3 and not isinstance(dy_t, tuple):
            dy_t = self._mm(dy)          # bf16x3: the GEMM operand is the split of the fp32 gradient itself (unless its producer wrote the planes)
        self._tn(dy_t, sv['o_op'] if 'o_op' in sv else self._mm(sv['o']), G[f'{pre}.{attn}.proj.weight'], G[f'{pre}.{attn}.proj.bias'])
        ops.gemm_nt(dy_t, self.Wt[f'{pre}.{attn}.proj'], None, EPI_STORE, out_t=do)
        dqkv = self._t(M, 3 * C) if self.fold else self._op(M, 3 * C)     # (bf16x3: the attention backward writes the operand planes itself)
        if self.fold:
            lin = f'{pre}.{attn}.qkv'
            if self._rows_tail_ok(lin, dy_t, extra, need_t):      # the consumer takes its row means itself: plain attention backward
                ops.attn_bwd(sv['qkv'], sv['o'], do, sv['lse'], dqkv, self.B, self.Tlen, cfg.J, cfg.H, cfg.scale, mode)
                del do
                return self._fold_tail(dqkv, None, sv, lin, f'{pre}.{norm}', dy, dy_t, extra, need_t)
            part = self._f(2 * cfg.H, M, 2)       # block-major; per head: the q columns, the k + v columns
            ops.attn_bwd_stats(sv['qkv'], sv['o'], do, sv['lse'], dqkv, self.Bf[lin], self.Rs[lin], part, self.B, self.Tlen, cfg.J, cfg.H,
                               cfg.scale, mode)
            del do
            return self._fold_tail(dqkv, part, sv, lin, f'{pre}.{norm}', dy, dy_t, extra, need_t)
        if dm is not None and dm[5] > 0:
            ops.attn_bwd(sv['qkv'], sv['o'], do, sv['lse'], dqkv, self.B, self.Tlen, cfg.J, cfg.H, cfg.scale, mode, drop=(dm[5], dm[6]))
        else:
            ops.attn_bwd(sv['qkv'], sv['o'], do, sv['lse'], dqkv, self.B, self.Tlen, cfg.J, cfg.H, cfg.scale, mode)
        del do
        dxn = self._t(M, C)
        dqkv = self._mm(dqkv)
        self._tn(dqkv, self._xn(sv, pre, norm), G[f'{pre}.{attn}.qkv.weight'], G.get(f'{pre}.{attn}.qkv.bias'))
        ops.gemm_nt(dqkv, self.Wt[f'{pre}.{attn}.qkv'], None, EPI_STORE, out_t=dxn)
        del dqkv
        dx = self._f(M, C)
        dx_t = (self._t(M, C) if not self.x3 else (self._op(M, C) if self.x3_planes else None)) if need_t else None      # bf16x3: operand planes
        if callable(extra):      # dual-stream backward: the other block's input gradient, awaited only now
            extra = extra()
        ops.layernorm_bwd(dxn, sv['x'], sv['mean'], sv['rstd'], P[f'{pre}.{norm}.weight'], dy, extra,
                          dx, dx_t, G[f'{pre}.{norm}.weight'], G[f'{pre}.{norm}.bias'])
        return dx, dx_t

    def _rows_tail_ok(self, lin, dy_t, extra, need_t) -> bool:
        """The row-owner LayerNorm-backward GEMM serves a folded pair whose gradient arrives and leaves in the operand type only."""
        return bool(self.rows_lnbwd and self.gstream and need_t and dy_t is not None and extra is None and lin in self.Pn and self.M < (1 << 22))

    def _fold_tail(self, dY, part, sv, lin, norm, dy, dy_t, extra, need_t):
        """Folded (LayerNorm -> Linear) pair, backward from the Linear's output gradient dY and its row dots `part`: weight
        gradient, then dx = dy [+ extra] + LayerNorm'(dY . W') as the epilogue of the dX GEMM, then the parameter gradients of
        W, gamma and beta from the folded weight gradient.  With the gradient stream in the operand type (`gstream`) the incoming
        dy is read as dy_t and, inside a Block (need_t), only the T-typed dx is written: returns (None, dx_t)."""
        cfg, ops, P, G = self.cfg, self.ops, self.P, self.grads
        M, C = self.M, cfg.C
        rows = part is None                # (round 5) row means taken by the dX kernel itself: no row dots, no row constants
        if not rows:
            rowc = self._f(M, 4)
            ops.lnbwd_rowc(part, sv['rstd'], rowc, C)
        db = G.get(lin + '.bias')
        if db is None:                     # qkv_bias=False: the column sums of dY are still needed for d(beta)
            db = self._f(dY.shape[1])
        ws = self._wstream()
        stream = self.gstream and dy_t is not None
        dres = dy_t if stream else dy

        def dx_gemm(extra):
            if rows:
                dx_t = self._t(M, C)
                ops.rows_lnbwd_t(dY, self.Pn[lin], sv['xn'], sv['rstd'], dres, dx_t)
                return None, dx_t
            dx = None if (stream and need_t) else self._f(M, C)
            dx_t = self._t(M, C) if need_t else None
            if callable(extra):      # dual-stream backward: the other block's input gradient, awaited only now
                extra = extra()
            ops.gemm_nt_lnbwd(dY, self.Wt[lin], sv['xn'], rowc, dres, extra, dx, dx_t)
            return dx, dx_t
        if self.fold_dx_first:       # A/B: the dX GEMM (critical path) before the weight gradient
            out = dx_gemm(extra)
        if ws is None:
            ops.gemm_tn(dY, sv['xn'], G[lin + '.weight'], db)
            ops.unfold_norm_grads(G[lin + '.weight'], db, P[lin + '.weight'], P[norm + '.weight'], P[norm + '.bias'],
                                  G[norm + '.weight'], G[norm + '.bias'])
        else:
            ws.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(ws):
                ops.gemm_tn(dY, sv['xn'], G[lin + '.weight'], db)
                ops.unfold_norm_grads(G[lin + '.weight'], db, P[lin + '.weight'], P[norm + '.weight'], P[norm + '.bias'],
                                      G[norm + '.weight'], G[norm + '.bias'])
            for t in (dY, sv['xn'], db):
                t.record_stream(ws)
        return out if self.fold_dx_first else dx_gemm(extra)

    def _mlp_bwd(self, dy, dy_t, sv, pre, norm, mlp, extra, need_t):
        cfg, ops, P, G = self.cfg, self.ops, self.P, self.grads
        M, C = self.M, cfg.C
        du = self._t(M, cfg.hidden)
        dm = sv.get('dm')
        if dm is not None and (dm[0] > 0 or dm[3] > 0):
            dy_t = self._t(M, C)
            ops.grad_drop(dy, dy_t, cfg.J, dm[0], dm[1], dm[3], dm[4])
            if self.x3:
                dy_t = self._mm(dy_t)
        elif self.x3 and not isinstance(dy_t, tuple):
            dy_t = self._mm(dy)
        g = sv['g']
        if g is None:                                         # recompute mode: post-activation from the saved pre-activation
            g = self._t(M, cfg.hidden)
            ops.gelu_fwd(sv['u'], g)
            if dm is not None and dm[0] > 0:
                ops.dropout(g, g, dm[0], dm[2])
            g = self._mm(g)
        self._tn(dy_t, g, G[f'{pre}.{mlp}.fc2.weight'], G[f'{pre}.{mlp}.fc2.bias'])
        del g
        if self.fold:
            lin = f'{pre}.{mlp}.fc1'
            if sv.get('d') is not None:                           # forward saved gelu'(u): one multiply, then the row-owner tail
                if dy_t is None:
                    dy_t = dy.to(self.T)
                if not self._rows_tail_ok(lin, dy_t, extra, need_t):
                    raise RuntimeError(f'{lin}: the derivative was saved for the row-owner LayerNorm backward, which cannot run here')
                ops.gemm_nt_mul(dy_t, self.Wt[f'{pre}.{mlp}.fc2'], sv['d'], du)
                return self._fold_tail(du, None, sv, lin, f'{pre}.{norm}', dy, dy_t, extra, need_t)
            if self._rows_tail_ok(lin, dy_t, extra, need_t):      # the consumer takes its row means itself: plain GELU' epilogue
                ops.gemm_nt(dy_t, self.Wt[f'{pre}.{mlp}.fc2'], None, EPI_DGELU, out_t=du, aux_t=sv['u'])
                return self._fold_tail(du, None, sv, lin, f'{pre}.{norm}', dy, dy_t, extra, need_t)
            part = self._f(cfg.hidden // 64, M, 2)
            ops.gemm_nt_dgelu_stats(dy_t, self.Wt[f'{pre}.{mlp}.fc2'], du, sv['u'], self.Bf[lin], self.Rs[lin], part)
            return self._fold_tail(du, part, sv, lin, f'{pre}.{norm}', dy, dy_t, extra, need_t)
        if self.x3 and not (dm is not None and dm[0] > 0):
            du = self._op(M, cfg.hidden)                      # bf16x3: the GELU' epilogue writes the operand planes itself
        ops.gemm_nt(dy_t, self.Wt[f'{pre}.{mlp}.fc2'], None, EPI_DGELU, out_t=du, aux_t=sv['u'])
        if dm is not None and dm[0] > 0:                      # backward of the drop after the activation (commutes with GELU')
            ops.dropout(du, du, dm[0], dm[2])
        dxn = self._t(M, C)
        du = self._mm(du)
        self._tn(du, self._xn(sv, pre, norm), G[f'{pre}.{mlp}.fc1.weight'], G[f'{pre}.{mlp}.fc1.bias'])
        ops.gemm_nt(du, self.Wt[f'{pre}.{mlp}.fc1'], None, EPI_STORE, out_t=dxn)
        del du
        dx = self._f(M, C)
        dx_t = (self._t(M, C) if not self.x3 else (self._op(M, C) if self.x3_planes else None)) if need_t else None      # bf16x3: operand planes
        if callable(extra):      # dual-stream backward: the other block's input gradient, awaited only now
            extra = extra()
        ops.layernorm_bwd(dxn, sv['x'], sv['mean'], sv['rstd'], P[f'{pre}.{norm}.weight'], dy, extra,
                          dx, dx_t, G[f'{pre}.{norm}.weight'], G[f'{pre}.{norm}.bias'])
        return dx, dx_t
