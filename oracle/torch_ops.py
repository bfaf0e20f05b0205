"""TEST INFRASTRUCTURE ONLY: a plain-PyTorch fp32 restatement of every kernel of libmbx.so.

It implements the `ops` protocol of `motionbert_amd/engine.py` with stock torch ops so that
  * the host sequencing (which activations are saved, how gradients flow through the dual
    streams, how parameter gradients are laid out) can be verified on a machine without a GPU
    against the reference's autograd gradients (tests/test_host_logic.py),
  * every HIP kernel has a same-op fp32 torch reference on the GPU box (tests/test_gpu_kernels.py),
  * bench.py has a CPU port to time on the host cores (`cpu_baseline`, kind "port"): it issues the
    same ATen CPU kernels (addmm / bmm / layer-norm style ops) the reference model would.
Like the rest of oracle/ it is never imported by the package and is not a fallback: the product
path only knows `motionbert_amd.hip_ops.HipOps`.

Each method states the contract the corresponding HIP kernel must satisfy.
"""
import math

import torch
import torch.nn.functional as F

from motionbert_amd.engine import (EPI_DGELU, EPI_GELU, EPI_RESID, EPI_STORE, EPI_TANH, MODE_SPATIAL)


def _gelu_grad(u):
    return 0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)


class MockOps:
    def __init__(self):
        self.calls = []

    def _log(self, name):
        self.calls.append(name)

    # weights -------------------------------------------------------------
    @staticmethod
    def split(t):
        """bf16x3 operand split (mbx_split_bf16): t = hi + lo up to 2^-16 relative."""
        hi = t.to(torch.bfloat16)
        return hi, (t - hi.float()).to(torch.bfloat16)

    def prep_weights(self, P, names, tdtype, need_t, x3=False):
        self._log('prep_weights')
        self.last_need_t = bool(need_t)
        if x3:
            Wn = {n: self.split(P[n + '.weight'].detach().contiguous()) for n in names}
            Wt = {n: self.split(P[n + '.weight'].detach().t().contiguous()) for n in names} if need_t else {}
            return Wn, Wt
        Wn = {n: P[n + '.weight'].detach().to(tdtype).contiguous() for n in names}
        Wt = {n: P[n + '.weight'].detach().t().to(tdtype).contiguous() for n in names} if need_t else {}
        return Wn, Wt

    # LayerNorm folding (include/mbx.h "LayerNorm folded into the Linear it feeds") ---------------------------
    @staticmethod
    def can_fold(tdtype, cfg):
        return cfg.hidden % 64 == 0

    def fold_norm_weights(self, P, pairs, need_t, tdtype=torch.bfloat16):
        """Wn = T(W diag(gamma)), Wt = its transpose, Bf = b + W beta, Rs = row sums of the ROUNDED folded weights."""
        self._log('fold_norm_weights')
        Wn, Wt, Bf, Rs = {}, {}, {}, {}
        for lin, norm in pairs:
            W, g, bt, b = P[lin + '.weight'].detach(), P[norm + '.weight'].detach(), P[norm + '.bias'].detach(), P.get(lin + '.bias')
            Wf = (W * g[None, :]).to(tdtype)
            Wn[lin] = Wf.contiguous()
            if need_t:
                Wt[lin] = Wf.t().contiguous()
            Bf[lin] = W @ bt + (b.detach() if b is not None else 0.0)
            Rs[lin] = Wf.float().sum(1)
        return Wn, Wt, Bf, Rs

    @staticmethod
    def _stat_vec(v, dtype):
        """The bf16 kernels take their row dots as packed-bf16 products: rsum and b' enter rounded to bf16 there."""
        return v.to(torch.bfloat16).float() if dtype == torch.bfloat16 else v

    def gemm_nt_dgelu_stats(self, a_t, w_t, out_t, aux_t, bias_f, rsum, part):
        """EPI_DGELU + part[n // 64][m] = { sum du rsum, sum du (u - bias_f) } over each 64-column block of the ROUNDED du."""
        self._log(f'gemm_nt.{EPI_DGELU}.stats')
        acc = a_t.float() @ w_t.float().t()
        u = aux_t.float()
        out_t.copy_((acc * _gelu_grad(u)).to(out_t.dtype))
        d = out_t.float()
        M, N = d.shape
        rsum, bias_f = self._stat_vec(rsum, out_t.dtype), self._stat_vec(bias_f, out_t.dtype)
        part.copy_(torch.stack([(d * rsum).reshape(M, N // 64, 64).sum(-1), (d * (u - bias_f)).reshape(M, N // 64, 64).sum(-1)], -1).transpose(0, 1))

    fuse_gelu_d = True        # tests switch it off to exercise the GELU' epilogue from the saved pre-activation

    def can_gelu_d(self, tdtype, cfg) -> bool:
        return bool(self.fuse_gelu_d)

    def gemm_nt_gelu_d(self, a_t, w_t, bias, out_d, out_g):
        """mbx_gemm_nt_gelu_d: out_g = gelu(a . w^T + bias), out_d = gelu'(a . w^T + bias), both from the fp32 product."""
        self._log('gemm_nt.gelu_d')
        u = a_t.float() @ w_t.float().t() + (bias if bias is not None else 0.)
        out_g.copy_(F.gelu(u).to(out_g.dtype))
        out_d.copy_(_gelu_grad(u).to(out_d.dtype))

    def gemm_nt_mul(self, a_t, w_t, aux_t, out_t):
        """mbx_gemm_nt_mul: out = (a . w^T) * aux."""
        self._log('gemm_nt.mul')
        out_t.copy_(((a_t.float() @ w_t.float().t()) * aux_t.float()).to(out_t.dtype))

    def attn_bwd_stats(self, qkv, o, do, lse, dqkv, bias_f, rsum, part, B, T, J, H, scale, mode):
        """attn_bwd + part[2 h + role][m] = { sum d rsum, sum d (qkv - bias_f) } of the rounded dqkv over the head's q columns
        (role 0) and over its k and v columns (role 1)."""
        self.attn_bwd(qkv, o, do, lse, dqkv, B, T, J, H, scale, mode)
        self.calls[-1] += '.stats'
        M = dqkv.shape[0]
        rsum, bias_f = self._stat_vec(rsum, dqkv.dtype), self._stat_vec(bias_f, dqkv.dtype)
        d = dqkv.float().reshape(M, 3, H, -1)
        y = (qkv.float() - bias_f).reshape(M, 3, H, -1)
        t1, t2 = (d * rsum.reshape(1, 3, H, -1)).sum(3), (d * y).sum(3)            # [M, 3, H]
        p1 = torch.stack([t1[:, 0], t1[:, 1] + t1[:, 2]], -1)                       # [M, H, role]
        p2 = torch.stack([t2[:, 0], t2[:, 1] + t2[:, 2]], -1)
        part.copy_(torch.stack([p1, p2], -1).reshape(M, 2 * H, 2).transpose(0, 1))      # block-major [2H, M, 2]

    def lnbwd_rowc(self, part, rstd, rowc, C):
        """rowc[m] = {rstd, rstd c1, rstd c2, 0}, c = column-block sums of part / C."""
        self._log('lnbwd_rowc')
        c = part.sum(0) / C        # part is block-major [nb, M, 2]
        rowc.copy_(torch.stack([rstd, rstd * c[:, 0], rstd * c[:, 1], torch.zeros_like(rstd)], -1))

    def gemm_nt_lnbwd(self, a_t, w_t, xhat, rowc, dres, extra, dx, dx_t):
        """dx = dres [+ extra] + rowc.x (a . w^T) - rowc.y - xhat rowc.z;  dx_t = T copy."""
        self._log('gemm_nt.lnbwd' + ('' if dx is not None else '.stream'))
        acc = a_t.float() @ w_t.float().t()
        r = dres.float() + rowc[:, 0:1] * acc - rowc[:, 1:2] - xhat.float() * rowc[:, 2:3]      # dres: fp32, or T inside a Block (gradient stream)
        if extra is not None:
            r = r + extra
        if dx is not None:
            dx.copy_(r)
        if dx_t is not None:
            dx_t.copy_(r.to(dx_t.dtype))

    fuse_rows_lnbwd = True    # tests switch it off to exercise the row-dot sequencing inside a Block too

    def can_rows_lnbwd(self, tdtype, cfg):
        return bool(self.fuse_rows_lnbwd)

    def rows_n_pack(self, w_t):
        self._log('rows_n_pack')
        return w_t

    def rows_n_pack_many(self, ws):
        self._log('rows_n_pack_many')
        return list(ws)

    def rows_lnbwd_t(self, dy_t, packed, xhat, rstd, dres_t, dx_t):
        """mbx_rows_lnbwd_t: dx_t = T(dres_t + rstd (dxhat - mean dxhat - xhat mean(dxhat xhat))), dxhat = dy . w^T in fp32."""
        self._log('rows_lnbwd_t')
        acc = dy_t.float() @ packed.float().t()
        xh = xhat.float()
        c1 = acc.mean(-1, keepdim=True)
        c2 = (acc * xh).mean(-1, keepdim=True)
        dx_t.copy_((dres_t.float() + rstd[:, None] * (acc - c1 - xh * c2)).to(dx_t.dtype))

    fuse_rows_resid_ln = True     # tests switch it off to exercise the residual GEMM + stand-alone LayerNorm sequencing

    def can_rows_resid_ln(self, tdtype, cfg):
        return bool(self.fuse_rows_resid_ln)

    def rows_resid_ln(self, a_t, packed, bias, resid, y, xhat, mean, rstd, eps):
        """mbx_rows_resid_ln: y = resid + a . w^T + bias (fp32); xhat = T((y - mean) rstd), two-pass statistics of the fp32 rows of y."""
        self._log('rows_resid_ln')
        y.copy_(resid + a_t.float() @ packed.float().t() + bias)
        mu = y.mean(-1)
        rs = torch.rsqrt(((y - mu[:, None]) ** 2).mean(-1) + eps)
        mean.copy_(mu)
        rstd.copy_(rs)
        xhat.copy_(((y - mu[:, None]) * rs[:, None]).to(xhat.dtype))

    def unfold_norm_grads(self, dw, db, w, gamma, beta, dgamma, dbeta):
        """in place: dw <- gamma[k] dw + db[n] beta[k];  dgamma = sum_n w dw';  dbeta = sum_n w db."""
        self._log('unfold_norm_grads')
        dgamma.copy_((w * dw).sum(0))
        dbeta.copy_((w * db[:, None]).sum(0))
        dw.copy_(dw * gamma[None, :] + db[:, None] * beta[None, :])

    # embedding -----------------------------------------------------------
    def embed_fwd(self, x, w, b, pos, temp, h, B, T, J):
        self._log('embed_fwd')
        y = x.reshape(B, T, J, -1) @ w.t() + b + pos.reshape(1, 1, J, -1) + temp[:, :T]
        h.copy_(y.reshape(h.shape))

    def embed_bwd_pair(self, dh_a, dh_b, x, w, dw, db, dpos, dtemp, dx, B, T, J):
        """mbx_embed_bwd_pair: embed_bwd on dh = dh_a + dh_b (two T-typed tensors)."""
        self.embed_bwd(dh_a.float() + dh_b.float(), x, w, dw, db, dpos, dtemp, dx, B, T, J, _name='embed_bwd_pair')

    def embed_bwd(self, dh, x, w, dw, db, dpos, dtemp, dx, B, T, J, _name='embed_bwd'):
        self._log(_name)
        C = dh.shape[-1]
        d4 = dh.reshape(B, T, J, C)
        dw.copy_(dh.t() @ x.reshape(-1, x.shape[-1]))
        db.copy_(dh.sum(0))
        dpos.copy_(d4.sum((0, 1)).reshape(dpos.shape))
        dtemp.zero_()
        dtemp[0, :T, 0] = d4.sum((0, 2))
        if dx is not None:
            dx.copy_((dh @ w).reshape(dx.shape))

    # layernorm -----------------------------------------------------------
    def layernorm_fwd(self, x, g, b, eps, y_t, mean, rstd):
        self._log('layernorm_fwd')
        mu = x.mean(-1)
        var = ((x - mu[:, None]) ** 2).mean(-1)
        rs = torch.rsqrt(var + eps)
        mean.copy_(mu)
        rstd.copy_(rs)
        xhat = (x - mu[:, None]) * rs[:, None]
        y = xhat if g is None else xhat * g + b                             # g = b = None: plain normalisation
        if isinstance(y_t, tuple):      # bf16x3: the operand planes straight from the kernel
            self._log('planes')
            hi, lo = self.split(y)
            y_t[0].copy_(hi)
            y_t[1].copy_(lo)
        else:
            y_t.copy_(y.to(y_t.dtype))

    def layernorm_bwd(self, dy_t, x, mean, rstd, g, dres, extra, dx, dx_t, dg, db):
        """dx = [dres] + [extra] + LN'(dy); dx_t = T copy of dx; dg, db reduced over rows."""
        self._log('layernorm_bwd')
        dy = dy_t.float()
        xhat = (x - mean[:, None]) * rstd[:, None]
        dg.copy_((dy * xhat).sum(0))
        db.copy_(dy.sum(0))
        dxh = dy * g
        r = rstd[:, None] * (dxh - dxh.mean(-1, keepdim=True) - xhat * (dxh * xhat).mean(-1, keepdim=True))
        if dres is not None:
            r = r + dres
        if extra is not None:
            r = r + extra
        dx.copy_(r)
        if isinstance(dx_t, tuple):      # bf16x3: the T copy as operand planes
            self._log('planes')
            hi, lo = self.split(r)
            dx_t[0].copy_(hi)
            dx_t[1].copy_(lo)
        elif dx_t is not None:
            dx_t.copy_(r.to(dx_t.dtype))

    # GEMMs ----------------------------------------------------------------
    def gemm_nt(self, a_t, w_t, bias, epi, out_t=None, out2_t=None, out_f=None, resid=None, aux_t=None):
        """acc[M,N] = a_t[M,K] @ w_t[N,K]^T in fp32 accumulation, then the epilogue."""
        self._log(f'gemm_nt.{epi}')
        if isinstance(a_t, tuple):     # bf16x3: three bf16 products in fp32 accumulation (the lo.lo term is dropped)
            (ah, al), (wh, wl) = a_t, w_t
            acc = ah.float() @ wh.float().t() + ah.float() @ wl.float().t() + al.float() @ wh.float().t()
        else:
            acc = a_t.float() @ w_t.float().t()
        if bias is not None:
            acc = acc + bias
        def put(dst, v):      # a T-typed output, or (bf16x3) the pair of operand planes the epilogue writes instead of fp32
            if isinstance(dst, tuple):
                self._log('planes')
                hi, lo = self.split(v)
                dst[0].copy_(hi)
                dst[1].copy_(lo)
            else:
                dst.copy_(v.to(dst.dtype))
        if epi == EPI_STORE:
            put(out_t, acc)
        elif epi == EPI_GELU:
            if out_t is not None:
                out_t.copy_(acc.to(out_t.dtype))
            put(out2_t, F.gelu(acc))   # gelu of the fp32 value, as the kernel epilogue does
        elif epi == EPI_RESID:
            out_f.copy_(resid + acc)
        elif epi == EPI_TANH:
            out_f.copy_(torch.tanh(acc))
        elif epi == EPI_DGELU:
            put(out_t, acc * _gelu_grad(aux_t.float()))
        else:
            raise ValueError(epi)

    layernorm_fwd_planes_ok = True      # producers accept a (hi, lo) pair where a bf16x3 GEMM operand is due
    grad_stream_t = True      # gemm_nt_lnbwd takes a T-typed dres and may skip the fp32 dx (the gradient stream inside a Block)

    # LayerNorm as a raw operand + the fused MLP forward (include/mbx.h; the no-grad path) ---------------------
    fuse_mlp = True           # tests switch it off to exercise the unfused no-grad sequencing

    def can_fuse_mlp(self, tdtype, cfg):
        return bool(self.fuse_mlp) and cfg.hidden % 64 == 0

    def mlp_pack_weights(self, w1_t, w2_t):
        """The HIP library reorders the two weights into MFMA-fragment order; the restatement keeps them as they are."""
        self._log('mlp_pack_weights')
        return (w1_t, w2_t)

    def rows_pack_nk(self, w_t):
        self._log('rows_pack_nk')
        return w_t

    def rows_gemm_nk(self, a_t, packed, bias, out_t, rsum=None, mean=None, rstd=None):
        """mbx_rows_gemm_nk: out = a . w^T + bias, or the raw-operand LayerNorm form with (rsum, mean, rstd)."""
        self._log('rows_gemm_nk')
        acc = a_t.float() @ packed.float().t()
        if mean is not None:
            acc = rstd[:, None] * (acc - mean[:, None] * rsum)
        out_t.copy_((acc + (bias if bias is not None else 0.)).to(out_t.dtype))

    def rows_gemm_nk_ln(self, x, packed, bias, rsum, eps, out_t):
        """mbx_rows_gemm_nk_ln: Linear'(LayerNorm(x)) from the fp32 rows: operand T(x), statistics of the fp32 rows, row constants applied
        to the accumulators: out = rstd (T(x) . w'^T - mean rsum) + b'."""
        self._log('rows_gemm.ln')
        mu = x.mean(-1, keepdim=True)
        rs = torch.rsqrt(((x - mu) ** 2).mean(-1, keepdim=True) + eps)
        sh = x[:, :1]                  # the kernel rounds the row SHIFTED by its first element (LayerNorm does not see the shift)
        acc = (x - sh).to(packed.dtype).float() @ packed.float().t()
        out_t.copy_((rs * (acc - (mu - sh) * rsum) + bias).to(out_t.dtype))

    def proj_mlp_pack_weights(self, wp_t, w1_t, w2_t):
        self._log('proj_mlp_pack_weights')
        return (wp_t, w1_t, w2_t)

    def proj_mlp_fused_fwd(self, o_t, packed, bp, b1, b2, rsum, resid, y, eps):
        """mbx_proj_mlp_fused_fwd: y1 = resid + o . Wp^T + bp (fp32), then the fused MLP on y1 with its operand T(y1) and the statistics of
        the fp32 rows of y1."""
        self._log('proj_mlp_fused_fwd')
        wp_t, w1_t, w2_t = packed
        y1 = resid + o_t.float() @ wp_t.float().t() + bp
        self.calls.append('_inner')
        self.mlp_fused_fwd(None, True, (w1_t, w2_t), b1, b2, rsum, y1, y, None, eps, None, None)
        del self.calls[-2:]

    def mlp_fused_fwd(self, a_t, raw_in, packed, b1, b2, rsum, resid, y, y_t, eps, mean, rstd):
        """y = resid + fc2(gelu(fc1)), fc1 = a . W1^T + b1 (raw_in = 0: a is the normalised operand) or
        rstd_a (a . W1^T - mean_a rsum) + b1 with (mean_a, rstd_a) the statistics of the bf16 rows of a (raw_in = 1);
        a_t = None: the operand is T(resid - resid[:, 0]) and (mean_a, rstd_a) are the statistics of the same shifted fp32 rows;
        the hidden passes through the operand type once (the kernel packs gelu(.) to bf16 for the second MFMA);
        y_t = T(y); mean / rstd = LayerNorm statistics of the rows of y."""
        self._log('mlp_fused_fwd' + ('' if a_t is not None else '.from_x'))
        w1_t, w2_t = packed
        if a_t is None:           # the operand is T(resid - resid[:, 0]); the statistics are those of the fp32 rows (shifted alike)
            sh = resid[:, :1]
            a_t = (resid - sh).to(w1_t.dtype)
            mu = resid.mean(-1, keepdim=True) - sh
            rs = torch.rsqrt(((resid - sh - mu) ** 2).mean(-1, keepdim=True) + eps)
        elif raw_in:
            af = a_t.float()
            mu = af.mean(-1, keepdim=True)
            rs = torch.rsqrt(((af * af).mean(-1, keepdim=True) - mu * mu).clamp_min(0) + eps)
        af = a_t.float()
        acc = af @ w1_t.float().t()
        if raw_in:
            u = rs * (acc - mu * rsum) + b1
        else:
            u = acc + b1
        gact = F.gelu(u).to(a_t.dtype)
        out = resid + gact.float() @ w2_t.float().t() + b2
        y.copy_(out)
        if y_t is not None:
            y_t.copy_(out.to(y_t.dtype))
        if mean is not None:
            mu_y = out.mean(-1)
            mean.copy_(mu_y)
            rstd.copy_(torch.rsqrt(((out - mu_y[:, None]) ** 2).mean(-1) + eps))

    def gelu_fwd(self, u, g):
        self._log('gelu_fwd')
        g.copy_(F.gelu(u.float()).to(g.dtype))

    def gemm_tn(self, dy_t, a_t, dw, db):
        """dw[N,K] = dy_t[M,N]^T @ a_t[M,K]; db[N] = column sums of dy_t (both fp32)."""
        self._log('gemm_tn')
        if isinstance(dy_t, tuple):
            (yh, yl), (ah, al) = dy_t, a_t
            dw.copy_(yh.float().t() @ ah.float() + yh.float().t() @ al.float() + yl.float().t() @ ah.float())
            if db is not None:
                db.copy_(yh.float().sum(0) + yl.float().sum(0))
            return
        dw.copy_(dy_t.float().t() @ a_t.float())
        if db is not None:
            db.copy_(dy_t.float().sum(0))

    # ActionNet pooling (mbx_pool_rep_fwd / mbx_tanh_pool_bwd) ------------------------------------
    @staticmethod
    def _keep(idx, p, seed):
        """The counter-based dropout mask of train_step.hip (drop_keep): motionbert_amd/dropmask.py restates it in torch."""
        from motionbert_amd.dropmask import keep
        return keep(idx, p, seed)

    def _mask(self, rep, p, seed):
        if p <= 0:
            return torch.ones_like(rep)
        idx = torch.arange(rep.numel(), dtype=torch.int64, device=rep.device).reshape(rep.shape)
        return self._keep(idx, p, seed).to(rep.dtype) / (1.0 - p)

    def pool_rep_fwd(self, rep, pooled, N, Mp, T, J, p=0.0, seed=0):
        self._log('pool_rep_fwd')
        pooled.copy_((rep * self._mask(rep, p, seed)).reshape(N, Mp * T, J, -1).mean(1))

    def tanh_pool_bwd(self, dpooled, rep, dpre_t, N, Mp, T, J, p=0.0, seed=0):
        self._log('tanh_pool_bwd')
        d = dpooled.reshape(N, 1, J, -1).expand(N, Mp * T, J, dpooled.shape[-1]).reshape(rep.shape) / (Mp * T)
        dpre_t.copy_((d * self._mask(rep, p, seed) * (1 - rep * rep)).to(dpre_t.dtype))

    # dropout / drop-path (mbx_dropout, mbx_residual_drop, mbx_grad_drop) ---------------------------
    def _branch_mask(self, rows, Cc, rps, p, seed, p_path, seed_path, device):
        m = self._mask(torch.empty(rows, Cc, device=device), p, seed)
        if p_path > 0:
            idx = torch.arange(rows, dtype=torch.int64, device=device) // rps
            m = m * (self._keep(idx, p_path, seed_path).float() / (1.0 - p_path))[:, None]
        return m

    def dropout(self, x, y, p, seed):
        self._log('dropout')
        y.copy_((x.float() * self._mask(x.float(), p, seed)).to(y.dtype))

    def residual_drop(self, y, x, rows_per_sample, p, seed, p_path, seed_path):
        self._log('residual_drop')
        y.copy_(x + (y - x) * self._branch_mask(y.shape[0], y.shape[1], rows_per_sample, p, seed, p_path, seed_path, y.device))

    def grad_drop(self, dy, dy_t, rows_per_sample, p, seed, p_path, seed_path):
        self._log('grad_drop')
        dy_t.copy_((dy * self._branch_mask(dy.shape[0], dy.shape[1], rows_per_sample, p, seed, p_path, seed_path, dy.device)).to(dy_t.dtype))

    # attention -------------------------------------------------------------
    def _split(self, qkv, B, T, J, H):
        C = qkv.shape[-1] // 3
        q5 = qkv.float().reshape(B, T, J, 3, H, C // H)
        return q5[:, :, :, 0], q5[:, :, :, 1], q5[:, :, :, 2]

    @staticmethod
    def _attn_mask(p, mode, drop):
        """keep / (1 - p) on the probabilities, over the flat index of the REFERENCE's attn tensor: [B T, H, J, J] (spatial,
        DSTformer.py:180) is this [B,T,H,J,J] flattened; [B, H, J, T, T] (temporal, :194) is this [B,J,H,T,T] with J and H swapped."""
        from motionbert_amd.dropmask import mask_like
        if drop is None or drop[0] <= 0:
            return None
        if mode == MODE_SPATIAL:
            return mask_like(p, drop[0], drop[1])
        return mask_like(p.permute(0, 2, 1, 3, 4).contiguous(), drop[0], drop[1]).permute(0, 2, 1, 3, 4)

    def attn_fwd(self, qkv, o, lse, B, T, J, H, scale, mode, drop=None):
        self._log(f'attn_fwd.{mode}' + ('.drop' if drop is not None and drop[0] > 0 else ''))
        q, k, v = self._split(qkv, B, T, J, H)
        if mode == MODE_SPATIAL:
            s = torch.einsum('btihd,btjhd->bthij', q, k) * scale
            l = torch.logsumexp(s, -1)                      # [B,T,H,J]
            p = torch.exp(s - l[..., None])
            m = self._attn_mask(p, mode, drop)
            oo = torch.einsum('bthij,btjhd->btihd', p if m is None else p * m, v)
            lse.copy_(l.permute(0, 1, 3, 2).reshape(lse.shape))
        else:
            s = torch.einsum('bsjhd,btjhd->bjhst', q, k) * scale
            l = torch.logsumexp(s, -1)                      # [B,J,H,T]
            p = torch.exp(s - l[..., None])
            m = self._attn_mask(p, mode, drop)
            oo = torch.einsum('bjhst,btjhd->bsjhd', p if m is None else p * m, v)
            lse.copy_(l.permute(0, 3, 1, 2).reshape(lse.shape))
        o.copy_(oo.reshape(o.shape).to(o.dtype))

    def attn_bwd(self, qkv, o, do, lse, dqkv, B, T, J, H, scale, mode, drop=None):
        """with `drop`: dP = mask (dO V^T), dV from the dropped probabilities, delta = rowsum(dO o) as ever (o carries the mask)."""
        self._log(f'attn_bwd.{mode}' + ('.drop' if drop is not None and drop[0] > 0 else ''))
        q, k, v = self._split(qkv, B, T, J, H)
        hd = q.shape[-1]
        do5 = do.float().reshape(B, T, J, H, hd)
        o5 = o.float().reshape(B, T, J, H, hd)
        delta = (do5 * o5).sum(-1)                           # [B,T,J,H]
        l4 = lse.reshape(B, T, J, H)
        out = torch.zeros(B, T, J, 3, H, hd, device=qkv.device)
        if mode == MODE_SPATIAL:
            s = torch.einsum('btihd,btjhd->bthij', q, k) * scale
            p = torch.exp(s - l4.permute(0, 1, 3, 2)[..., None])
            m = self._attn_mask(p, mode, drop)
            dp = torch.einsum('btihd,btjhd->bthij', do5, v)
            pd = p
            if m is not None:
                dp, pd = dp * m, p * m
            ds = p * (dp - delta.permute(0, 1, 3, 2)[..., None]) * scale
            out[:, :, :, 0] = torch.einsum('bthij,btjhd->btihd', ds, k)
            out[:, :, :, 1] = torch.einsum('bthij,btihd->btjhd', ds, q)
            out[:, :, :, 2] = torch.einsum('bthij,btihd->btjhd', pd, do5)
        else:
            s = torch.einsum('bsjhd,btjhd->bjhst', q, k) * scale
            p = torch.exp(s - l4.permute(0, 2, 3, 1)[..., None])
            m = self._attn_mask(p, mode, drop)
            dp = torch.einsum('bsjhd,btjhd->bjhst', do5, v)
            pd = p
            if m is not None:
                dp, pd = dp * m, p * m
            ds = p * (dp - delta.permute(0, 2, 3, 1)[..., None]) * scale
            out[:, :, :, 0] = torch.einsum('bjhst,btjhd->bsjhd', ds, k)
            out[:, :, :, 1] = torch.einsum('bjhst,bsjhd->btjhd', ds, q)
            out[:, :, :, 2] = torch.einsum('bjhst,bsjhd->btjhd', pd, do5)
        if isinstance(dqkv, tuple):      # bf16x3: dq / dk / dv leave as the operand planes of the GEMMs that read them
            self._log('planes')
            hi, lo = self.split(out.reshape(dqkv[0].shape))
            dqkv[0].copy_(hi)
            dqkv[1].copy_(lo)
        else:
            dqkv.copy_(out.reshape(dqkv.shape).to(dqkv.dtype))

    # fusion ----------------------------------------------------------------
    def fuse_fwd(self, x_st, x_ts, w, b, out, alpha):
        self._log('fuse_fwd')
        a = torch.softmax(torch.cat([x_st, x_ts], -1) @ w.t() + b, -1)
        alpha.copy_(a)
        out.copy_(x_st * a[:, 0:1] + x_ts * a[:, 1:2])

    def fuse_ln_fwd(self, x_st, x_ts, w, b, out, alpha, g1, b1, xn1, g2, b2, xn2, eps, mean, rstd):
        self.fuse_fwd(x_st, x_ts, w, b, out, alpha)
        self.layernorm_fwd(out, g1, b1, eps, xn1, mean, rstd)
        if xn2 is not None:
            self.layernorm_fwd(out, g2, b2, eps, xn2, mean, rstd)
        del self.calls[-(3 if xn2 is not None else 2):]
        self._log('fuse_ln_fwd')

    def fuse_bwd_pair(self, dh_a, dh_b, x_st, x_ts, alpha, w, d_st_t, d_ts_t, dw, db):
        """mbx_fuse_bwd_pair: fuse_bwd on dh = dh_a + dh_b (two T-typed tensors), T-typed outputs only."""
        self.fuse_bwd(dh_a.float() + dh_b.float(), x_st, x_ts, alpha, w, None, None, d_st_t, d_ts_t, dw, db, _name='fuse_bwd_pair')

    def fuse_bwd(self, dh, x_st, x_ts, alpha, w, d_st, d_ts, d_st_t, d_ts_t, dw, db, _name='fuse_bwd'):
        self._log(_name)
        C = x_st.shape[-1]
        da = torch.stack([(dh * x_st).sum(-1), (dh * x_ts).sum(-1)], -1)
        dl = alpha * (da - (da * alpha).sum(-1, keepdim=True))
        dw.copy_(dl.t() @ torch.cat([x_st, x_ts], -1))
        db.copy_(dl.sum(0))
        dcat = dl @ w
        r_st, r_ts = dh * alpha[:, 0:1] + dcat[:, :C], dh * alpha[:, 1:2] + dcat[:, C:]
        if d_st is not None:          # (None with the gradient stream in the operand type: only the T-typed copies are written)
            d_st.copy_(r_st)
            d_ts.copy_(r_ts)
        d_st_t.copy_(r_st.to(d_st_t.dtype))
        d_ts_t.copy_(r_ts.to(d_ts_t.dtype))

    def average(self, x_st, x_ts, out):
        self._log('average')
        out.copy_((x_st + x_ts) * 0.5)

    def average_bwd(self, dh, d_st, d_ts, d_st_t, d_ts_t):
        self._log('average_bwd')
        d_st.copy_(dh * 0.5)
        d_ts.copy_(dh * 0.5)
        d_st_t.copy_(d_st.to(d_st_t.dtype))
        d_ts_t.copy_(d_ts.to(d_ts_t.dtype))

    # tail ------------------------------------------------------------------
    def head_fwd(self, rep, w, b, out):
        self._log('head_fwd')
        out.copy_(rep @ w.t() + b)

    def head_bwd(self, dout, rep, w, dpre_t, dw, db):
        """dpre = (dout @ w) * (1 - rep^2)  (tanh' folded in);  dw = dout^T rep;  db = sum dout."""
        self._log('head_bwd')
        dpre_t.copy_(((dout @ w) * (1 - rep * rep)).to(dpre_t.dtype))
        dw.copy_(dout.t() @ rep)
        db.copy_(dout.sum(0))

    def tanh_bwd(self, drep, rep, dpre_t):
        self._log('tanh_bwd')
        dpre_t.copy_((drep * (1 - rep * rep)).to(dpre_t.dtype))
