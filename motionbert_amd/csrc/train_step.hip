// the training step AROUND the backbone (train.py:174-206, 289, 360-362; lib/model/loss.py:56-142).
//   mbx_pose_loss  : loss_mpjpe + lambda_scale * n_mpjpe + lambda_velocity * loss_velocity and d(loss)/d(pred) in ONE pass
//                    over pred / gt [B,T,J,3] -- the loss is the root of backward, so its gradient is a closed form; no
//                    intermediate tensors, no host synchronisation (the reference reads 8 scalars back per step with .item()).
//   mbx_adamw_step : AdamW (decoupled weight decay, torch.optim.AdamW semantics) over ONE flat fp32 parameter / gradient /
//                    moment buffer: one launch for all 42.5 M parameters; step count and learning rate live on the device so
//                    that the whole training step can be replayed from a hipGraph.
#include "mbx_common.h"

// ---------------------------------------------------------------------------------------------------------------
// pose loss.  One wave per frame (b, t), lane j < J owns joint j; sums over the joints of a frame are wave reductions.
//   mpjpe    = mean_{b,t,j} |p - g|                                                        (loss.py:56-62)
//   n_mpjpe  = mean |s p - g|,  s = sum_j g.p / sum_j p.p per frame                        (loss.py:81-91)
//   velocity = mean_{b,t>=1,j} |(p_t - p_{t-1}) - (g_t - g_{t-1})|                          (loss.py:133-142)
// Gradients (|r| = 0 contributes 0, as torch.norm's backward does):
//   d mpjpe / dp_k   = r_k / |r_k| / n
//   d n_mpjpe / dp_k = [ s e_k + c (g_k b - 2 a p_k) / b^2 ] / n,   e_j = (s p_j - g_j) / |.|,  c = sum_j e_j.p_j
//   d vel / dp_t     = [ u_t - u_{t+1} ] / n_v,   u_t = d_t / |d_t|  (u_0 = u_T = 0)
// Per frame it writes 4 partial sums (already divided by the element counts): mpjpe, n_mpjpe, velocity, total.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float norm3(float x, float y, float z) { return sqrtf(fmaf(x, x, fmaf(y, y, z * z))); }

__global__ __launch_bounds__(256) void pose_loss_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                        float* __restrict__ part, float* __restrict__ dpred, float ls, float lv,
                                                        float gscale, int B, int T, int J) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frame = blockIdx.x * 4 + wave;
    if (frame >= B * T) return;                      // whole waves only: the reductions below need all 64 lanes
    const int t = frame % T;
    const bool on = lane < J;
    const size_t o = ((size_t)frame * J + (on ? lane : 0)) * 3;
    const float inv_n = 1.0f / ((float)B * T * J), inv_nv = T > 1 ? 1.0f / ((float)B * (T - 1) * J) : 0.f;
    float p[3] = {0.f, 0.f, 0.f}, g[3] = {0.f, 0.f, 0.f};
    if (on) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { p[c] = pred[o + c]; g[c] = gt[o + c]; }
    }
    // ---- mpjpe
    const float r0 = p[0] - g[0], r1 = p[1] - g[1], r2 = p[2] - g[2];
    const float n1 = on ? norm3(r0, r1, r2) : 0.f;
    const float i1 = n1 > 0.f ? 1.0f / n1 : 0.f;
    float grad[3] = {r0 * i1 * inv_n, r1 * i1 * inv_n, r2 * i1 * inv_n};
    // ---- n_mpjpe
    const float a = wave_sum(g[0] * p[0] + g[1] * p[1] + g[2] * p[2]);
    const float b = wave_sum(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    const float s = a / b;
    const float q0 = s * p[0] - g[0], q1 = s * p[1] - g[1], q2 = s * p[2] - g[2];
    const float n2 = on ? norm3(q0, q1, q2) : 0.f;
    const float i2 = n2 > 0.f ? 1.0f / n2 : 0.f;
    const float e0 = q0 * i2, e1 = q1 * i2, e2 = q2 * i2;
    const float c = wave_sum(e0 * p[0] + e1 * p[1] + e2 * p[2]);
    const float k1 = c / b, k2 = 2.0f * a * c / (b * b);          // c (g b - 2 a p) / b^2 = k1 g - k2 p
    grad[0] += ls * inv_n * (s * e0 + k1 * g[0] - k2 * p[0]);
    grad[1] += ls * inv_n * (s * e1 + k1 * g[1] - k2 * p[1]);
    grad[2] += ls * inv_n * (s * e2 + k1 * g[2] - k2 * p[2]);
    // ---- velocity
    float n3 = 0.f;
    if (on && T > 1) {
        const size_t fs = (size_t)J * 3;
        if (t >= 1) {
            const float d0 = (p[0] - pred[o - fs]) - (g[0] - gt[o - fs]), d1 = (p[1] - pred[o - fs + 1]) - (g[1] - gt[o - fs + 1]),
                        d2 = (p[2] - pred[o - fs + 2]) - (g[2] - gt[o - fs + 2]);
            n3 = norm3(d0, d1, d2);
            const float i3 = n3 > 0.f ? lv * inv_nv / n3 : 0.f;
            grad[0] += d0 * i3; grad[1] += d1 * i3; grad[2] += d2 * i3;
        }
        if (t + 1 < T) {
            const float d0 = (pred[o + fs] - p[0]) - (gt[o + fs] - g[0]), d1 = (pred[o + fs + 1] - p[1]) - (gt[o + fs + 1] - g[1]),
                        d2 = (pred[o + fs + 2] - p[2]) - (gt[o + fs + 2] - g[2]);
            const float nn = norm3(d0, d1, d2);
            const float i3 = nn > 0.f ? lv * inv_nv / nn : 0.f;
            grad[0] -= d0 * i3; grad[1] -= d1 * i3; grad[2] -= d2 * i3;
        }
    }
    if (on && dpred) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) dpred[o + cc] = gscale * grad[cc];
    }
    const float s1 = wave_sum(n1) * inv_n, s2 = wave_sum(n2) * inv_n, s3 = wave_sum(n3) * inv_nv;
    if (lane == 0) {
        float* pr = part + (size_t)frame * 4;
        pr[0] = s1; pr[1] = s2; pr[2] = s3; pr[3] = s1 + ls * s2 + lv * s3;
    }
}
extern "C" size_t mbx_pose_loss_ws(int B, int T) { return (size_t)B * T * 4 * sizeof(float) + 256; }
extern "C" int mbx_pose_loss(const float* pred, const float* gt, float lambda_scale, float lambda_velocity, float* losses,
                             float* dpred, float grad_scale, int B, int T, int J, void* ws, void* stream) {
    MBX_CHECK_ARG(pred && gt && losses && ws, "pose_loss: null pointer");
    MBX_CHECK_ARG(B > 0 && T > 0 && J > 0 && J <= 64, "pose_loss: bad shape B=%d T=%d J=%d (J <= 64)", B, T, J);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pose_loss_kernel, dim3((B * T + 3) / 4), dim3(256), 0, s, pred, gt, (float*)ws, dpred, lambda_scale,
                       lambda_velocity, grad_scale, B, T, J);
    MBX_LAUNCH_CHECK("pose_loss");
    return mbx_launch_colsum((const float*)ws, B * T, 4, 0, 4, losses, s);   // fixed order, no atomics: deterministic
}

// ---------------------------------------------------------------------------------------------------------------
// 2D re-projection loss of the pre-training's 2D branch (lib/model/loss.py:72-77, used at train.py:200-203):
//   loss = mean_{b,t,j} | (pred_xy - target_xy) * conf |          conf = detector confidence of the joint (train.py:164)
//   d loss / d pred_xy = conf^2 (pred_xy - target_xy) / | (pred_xy - target_xy) conf | / n,   d / d pred_z = 0
// (|.| = 0 contributes a zero gradient, as torch.norm's backward does).  pred [B,T,J,3]; target and conf are read with an
// element stride so that the [B,T,J,3] 2D batch itself serves as both (x, y in channels 0-1, confidence in channel 2:
// no deep copy of the confidence, train.py:164).  One wave per frame, lane j < J owns joint j; per-frame partial sums.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_2d_kernel(const float* __restrict__ pred, const float* __restrict__ target, int tstride,
                                                      const float* __restrict__ conf, int cstride, float* __restrict__ part,
                                                      float* __restrict__ dpred, float gscale, int nframes, int J) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frame = blockIdx.x * 4 + wave;
    if (frame >= nframes) return;                    // whole waves only
    const bool on = lane < J;
    const size_t tok = (size_t)frame * J + (on ? lane : 0);
    const float inv_n = 1.0f / ((float)nframes * J);
    float nrm = 0.f, g0 = 0.f, g1 = 0.f;
    if (on) {
        const float c = conf[tok * cstride];
        const float r0 = (pred[tok * 3] - target[tok * tstride]) * c, r1 = (pred[tok * 3 + 1] - target[tok * tstride + 1]) * c;
        nrm = sqrtf(fmaf(r0, r0, r1 * r1));
        const float inv = nrm > 0.f ? c * inv_n / nrm : 0.f;
        g0 = r0 * inv; g1 = r1 * inv;
    }
    if (on && dpred) { dpred[tok * 3] = gscale * g0; dpred[tok * 3 + 1] = gscale * g1; dpred[tok * 3 + 2] = 0.f; }
    const float s = wave_sum(nrm) * inv_n;
    if (lane == 0) part[frame] = s;
}
extern "C" size_t mbx_loss_2d_weighted_ws(int B, int T) { return (size_t)B * T * sizeof(float) + 256; }
extern "C" int mbx_loss_2d_weighted(const float* pred, const float* target, int target_stride, const float* conf, int conf_stride,
                                    float* loss, float* dpred, float grad_scale, int B, int T, int J, void* ws, void* stream) {
    MBX_CHECK_ARG(pred && target && conf && loss && ws, "loss_2d_weighted: null pointer");
    MBX_CHECK_ARG(B > 0 && T > 0 && J > 0 && J <= 64, "loss_2d_weighted: bad shape B=%d T=%d J=%d (J <= 64)", B, T, J);
    MBX_CHECK_ARG(target_stride >= 2 && conf_stride >= 1, "loss_2d_weighted: target stride %d (>= 2), conf stride %d (>= 1)", target_stride, conf_stride);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_2d_kernel, dim3((B * T + 3) / 4), dim3(256), 0, s, pred, target, target_stride, conf, conf_stride,
                       (float*)ws, dpred, grad_scale, B * T, J);
    MBX_LAUNCH_CHECK("loss_2d_weighted");
    return mbx_launch_colsum((const float*)ws, B * T, 1, 0, 1, loss, s);   // fixed order, no atomics: deterministic
}

// ---------------------------------------------------------------------------------------------------------------
// AdamW over a flat buffer (torch.optim.AdamW, amsgrad=False, maximize=False):
//   p *= 1 - lr wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr / (1-b1^t) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
// `state` = {step count t (float), learning rate} on the device: the tick kernel advances t, so a captured hipGraph replays
// the correct bias corrections, and a host-side LR schedule only has to rewrite state[1].
// ---------------------------------------------------------------------------------------------------------------
__global__ void adamw_tick_kernel(float* state) { state[0] += 1.0f; }
__device__ __forceinline__ void adamw_one(float& pv, float gv, float& mv, float& vv, float b1, float b2, float eps, float decay, float step,
                                          float rs2) {
    mv = fmaf(b1, mv, (1.0f - b1) * gv);
    vv = fmaf(b2, vv, (1.0f - b2) * gv * gv);
    pv = pv * decay - step * mv / (sqrtf(vv) * rs2 + eps);
}
// elements [0, head) and [head + 4 n4, n) one per thread (a range of the flat buffer need not start or end on a 16-byte
// boundary: frozen parameters are skipped range by range), the aligned middle with 16-byte accesses
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n, size_t head, size_t n4, const float* __restrict__ state,
                                                    float b1, float b2, float eps, float wd) {
    const float t = state[0], lr = state[1];
    const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
    const float step = lr / bc1, rs2 = rsqrtf(bc2), decay = 1.0f - lr * wd;
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (tid < 8) {      // at most 3 leading + 3 trailing elements
        const size_t i = tid < 4 ? tid : head + 4 * n4 + (tid - 4);
        if ((tid < 4 && i < head) || (tid >= 4 && i < n)) adamw_one(p[i], g[i], m[i], v[i], b1, b2, eps, decay, step, rs2);
    }
    float* pa = p + head; const float* ga = g + head; float* ma = m + head; float* va = v + head;
    for (size_t i = tid; i < n4; i += (size_t)gridDim.x * 256) {
        float pv[4], gv[4], mv[4], vv[4];
        load4<float>(pa + i * 4, pv); load4<float>(ga + i * 4, gv); load4<float>(ma + i * 4, mv); load4<float>(va + i * 4, vv);
#pragma unroll
        for (int e = 0; e < 4; ++e) adamw_one(pv[e], gv[e], mv[e], vv[e], b1, b2, eps, decay, step, rs2);
        store4<float>(pa + i * 4, pv); store4<float>(ma + i * 4, mv); store4<float>(va + i * 4, vv);
    }
}
extern "C" int mbx_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float* state, float beta1, float beta2,
                              float eps, float weight_decay, int tick, void* stream) {
    MBX_CHECK_ARG(p && g && m && v && state, "adamw_step: null pointer");
    const size_t mis = (size_t)((uintptr_t)p & 15);
    MBX_CHECK_ARG(mis % 4 == 0 && ((uintptr_t)g & 15) == mis && ((uintptr_t)m & 15) == mis && ((uintptr_t)v & 15) == mis,
                  "adamw_step: p, g, m, v must be 4-byte aligned and share their offset within a 16-byte line (same range of parallel flat buffers)");
    hipStream_t s = (hipStream_t)stream;
    if (tick) hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(1), 0, s, state);
    if (n) {
        size_t head = ((16 - mis) % 16) / 4;
        if (head > n) head = n;
        const size_t n4 = (n - head) / 4;
        const size_t want = (n4 + 255) / 256;
        const int grid = (int)(want < 256 * 16 ? (want ? want : 1) : 256 * 16);
        hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, s, p, g, m, v, n, head, n4, state, beta1, beta2, eps, weight_decay);
    }
    MBX_LAUNCH_CHECK("adamw_step");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// the ActionNet pooling on the representation (lib/model/model_action.py:15-24,62-70).
// ActionHeadClassification drops out the [N, Mp, T, J, R] representation element-wise, averages it over the T frames and
// over the Mp persons and feeds [N, J*R] to fc1.  Here the dropout mask, both means and -- in backward -- the tanh' of the
// backbone tail are one pass each over `rep`:
//   pooled[n, j, r] = 1/(Mp T) sum_{m,t} keep(idx) / (1-p) * rep[((n Mp + m) T + t) J + j, r]
//   dpre[tok, r]    = dpooled[n, j, r] / (Mp T) * keep(idx) / (1-p) * (1 - rep[tok, r]^2)
// so the [N, Mp, T, J, R] cotangent (541 MB at N = 32, Mp = 2, T = 243: an expand + a permute copy in the reference) is
// never materialised.  keep(idx) is a counter-based hash of (seed, element index): the same mask in both passes without
// storing it.  p = 0 (evaluation, or dropout_ratio 0) skips the hash.
// ---------------------------------------------------------------------------------------------------------------
// (drop_keep, the counter-based keep decision, lives in mbx_common.h: the attention kernels use it for the probability dropout)
// one block per (n, j); thread c4 owns 4 consecutive channels; loops over (m, t)
__global__ __launch_bounds__(128) void pool_rep_fwd_kernel(const float* __restrict__ rep, float* __restrict__ pooled, int Mp, int T,
                                                           int J, int R, float p, uint32_t seed_lo, uint32_t seed_hi) {
    const int n = blockIdx.x / J, j = blockIdx.x % J;
    const uint32_t thresh = p > 0.f ? (uint32_t)fminf(p * 4294967296.0f, 4294967295.0f) : 0u;
    const float keep_scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (int c = threadIdx.x * 4; c < R; c += 128 * 4) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int m = 0; m < Mp; ++m)
            for (int t = 0; t < T; ++t) {
                const size_t tok = ((size_t)(n * Mp + m) * T + t) * J + j;
                float v[4];
                const uint64_t base = (uint64_t)tok * R + c;        // multiple of 4: base + e never carries into the high half
                load4<float>(rep + base, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // (a multiplier, not `k ? v : 0`: hipcc 7.2 miscompiled the select -- it reused the destination register of
                    //  the in-flight 16-byte load as hash scratch and element 0 came out as 0; tools/probes/README.md)
                    const float km = (p > 0.f && !drop_keep(seed_lo, seed_hi, (uint32_t)base + e, (uint32_t)(base >> 32), thresh)) ? 0.f : 1.f;
                    a[e] = fmaf(km, v[e], a[e]);
                }
            }
        const float s = keep_scale / (float)(Mp * T);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] *= s;
        store4<float>(pooled + ((size_t)n * J + j) * R + c, a);
    }
}
extern "C" int mbx_pool_rep_fwd(const float* rep, float* pooled, int N, int Mp, int T, int J, int R, float p, uint64_t seed,
                                void* stream) {
    MBX_CHECK_ARG(rep && pooled, "pool_rep_fwd: null pointer");
    MBX_CHECK_ARG(N > 0 && Mp > 0 && T > 0 && J > 0 && R > 0 && R % 4 == 0 && p >= 0.f && p < 1.f, "pool_rep_fwd: bad arguments");
    hipLaunchKernelGGL(pool_rep_fwd_kernel, dim3(N * J), dim3(128), 0, (hipStream_t)stream, rep, pooled, Mp, T, J, R, p,
                       (uint32_t)seed, (uint32_t)(seed >> 32));
    MBX_LAUNCH_CHECK("pool_rep_fwd");
    return 0;
}
template <typename T_>
__global__ __launch_bounds__(256) void tanh_pool_bwd_kernel(const float* __restrict__ dpooled, const float* __restrict__ rep,
                                                            T_* __restrict__ dpre, size_t ntok, int Mp, int T, int J, int R, float p,
                                                            uint32_t seed_lo, uint32_t seed_hi) {
    const uint32_t thresh = p > 0.f ? (uint32_t)fminf(p * 4294967296.0f, 4294967295.0f) : 0u;
    const float s = (p > 0.f ? 1.0f / (1.0f - p) : 1.0f) / (float)(Mp * T);
    const int r4 = R >> 2;
    const size_t total = ntok * r4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t tok = i / r4;
        const int c = (int)(i % r4) * 4;
        const int j = (int)(tok % J);
        const int n = (int)(tok / ((size_t)J * T * Mp));
        float d[4], r[4];
        const uint64_t base = (uint64_t)tok * R + c;
        load4<float>(dpooled + ((size_t)n * J + j) * R + c, d);
        load4<float>(rep + base, r);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float km = (p > 0.f && !drop_keep(seed_lo, seed_hi, (uint32_t)base + e, (uint32_t)(base >> 32), thresh)) ? 0.f : s;
            d[e] = d[e] * km * (1.0f - r[e] * r[e]);
        }
        store4<T_>(dpre + base, d);
    }
}
extern "C" int mbx_tanh_pool_bwd(const float* dpooled, const float* rep, void* dpre_t, int N, int Mp, int T, int J, int R, float p,
                                 uint64_t seed, int dtype, void* stream) {
    MBX_CHECK_ARG(dpooled && rep && dpre_t, "tanh_pool_bwd: null pointer");
    MBX_CHECK_ARG(N > 0 && Mp > 0 && T > 0 && J > 0 && R > 0 && R % 4 == 0 && p >= 0.f && p < 1.f, "tanh_pool_bwd: bad arguments");
    const size_t ntok = (size_t)N * Mp * T * J;
    const size_t want = (ntok * (R / 4) + 255) / 256;
    const int grid = (int)(want < 4096 ? (want ? want : 1) : 4096);
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    if (dtype == MBX_BF16)
        hipLaunchKernelGGL(tanh_pool_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dpooled, rep, (bf16_t*)dpre_t, ntok, Mp, T, J, R, p, lo, hi);
    else if (dtype == MBX_F32)
        hipLaunchKernelGGL(tanh_pool_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dpooled, rep, (float*)dpre_t, ntok, Mp, T, J, R, p, lo, hi);
    else
        return mbx_set_error("tanh_pool_bwd: unknown dtype %d", dtype);
    MBX_LAUNCH_CHECK("tanh_pool_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Dropout / DropPath with p > 0 in training (DSTformer.py:77,96,104,278; drop.py:17-32).  No shipped
// config uses them (lib/utils/learning.py:83-85 does not forward the rates), so they are not fused into the GEMM
// epilogues: three small element-wise kernels apply them around the fused path, with the counter-based mask above
// (keep(seed, flat element index)), i.e. nothing is stored between forward and backward but the seeds.
//   mbx_dropout        y = x * keep / (1-p)                        pos_drop, MLP drop after GELU; and their backward (same call)
//   mbx_residual_drop  y <- x + (y - x) * keep_e/(1-p) * keep_path/(1-p_path)     proj_drop / MLP drop after fc2 + DropPath on the
//                      branch of a residual sub-layer whose fused epilogue already produced y = x + branch
//   mbx_grad_drop      dy_t = T(dy * the same two masks): the gradient entering that branch
// DropPath draws ONE mask value per leading index of the [B*T, J, C] tensor (drop.py:27: shape (x.shape[0], 1, 1)), i.e. per
// frame: rows_per_sample = J.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float keep_mul(float p, uint32_t thresh, uint32_t slo, uint32_t shi, uint64_t idx, float scale) {
    return (p > 0.f && !drop_keep(slo, shi, (uint32_t)idx, (uint32_t)(idx >> 32), thresh)) ? 0.f : scale;
}
__device__ __forceinline__ uint32_t p_thresh(float p) { return p > 0.f ? (uint32_t)fminf(p * 4294967296.0f, 4294967295.0f) : 0u; }

template <typename T_>
__global__ __launch_bounds__(256) void dropout_kernel(const T_* __restrict__ x, T_* __restrict__ y, size_t n4, float p, uint32_t slo,
                                                      uint32_t shi) {
    const uint32_t th = p_thresh(p);
    const float sc = 1.0f / (1.0f - p);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float v[4];
        load4<T_>(x + i * 4, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= keep_mul(p, th, slo, shi, i * 4 + e, sc);
        store4<T_>(y + i * 4, v);
    }
}
extern "C" int mbx_dropout(const void* x, void* y, size_t n, float p, uint64_t seed, int dtype, void* stream) {
    MBX_CHECK_ARG(x && y && n % 4 == 0 && p >= 0.f && p < 1.f, "dropout: bad arguments");
    if (n == 0) return 0;
    const size_t want = (n / 4 + 255) / 256;
    const int grid = (int)(want < 4096 ? want : 4096);
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    if (dtype == MBX_BF16)
        hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, n / 4, p, lo, hi);
    else if (dtype == MBX_F32)
        hipLaunchKernelGGL(dropout_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, n / 4, p, lo, hi);
    else
        return mbx_set_error("dropout: unknown dtype %d", dtype);
    MBX_LAUNCH_CHECK("dropout");
    return 0;
}
// MODE 0: y = x + (y - x) * m  (in place on y);  MODE 1: out_t = T(dy * m)
template <typename T_, int MODE>
__global__ __launch_bounds__(256) void branch_drop_kernel(float* __restrict__ y, const float* __restrict__ x, T_* __restrict__ out_t,
                                                          size_t rows, int C, int rps, float p, uint32_t slo, uint32_t shi,
                                                          float pp, uint32_t plo, uint32_t phi) {
    const uint32_t th = p_thresh(p), pth = p_thresh(pp);
    const float sc = 1.0f / (1.0f - p), psc = 1.0f / (1.0f - pp);
    const int c4 = C >> 2;
    const size_t total = rows * c4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / c4;
        const int c = (int)(i % c4) * 4;
        const float mp = keep_mul(pp, pth, plo, phi, row / rps, psc);
        float a[4], b[4];
        load4<float>(y + row * C + c, a);
        if (MODE == 0) load4<float>(x + row * C + c, b);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float m = mp * keep_mul(p, th, slo, shi, row * C + c + e, sc);
            a[e] = MODE == 0 ? fmaf(a[e] - b[e], m, b[e]) : a[e] * m;
        }
        if (MODE == 0) store4<float>(y + row * C + c, a); else store4<T_>(out_t + row * C + c, a);
    }
}
extern "C" int mbx_residual_drop(float* y, const float* x, size_t rows, int C, int rows_per_sample, float p, uint64_t seed,
                                 float p_path, uint64_t seed_path, void* stream) {
    MBX_CHECK_ARG(y && x && C > 0 && C % 4 == 0 && rows_per_sample > 0 && p >= 0.f && p < 1.f && p_path >= 0.f && p_path < 1.f,
                  "residual_drop: bad arguments");
    if (rows == 0) return 0;
    const size_t want = (rows * (C / 4) + 255) / 256;
    const int grid = (int)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL((branch_drop_kernel<float, 0>), dim3(grid), dim3(256), 0, (hipStream_t)stream, y, x, (float*)nullptr, rows, C,
                       rows_per_sample, p, (uint32_t)seed, (uint32_t)(seed >> 32), p_path, (uint32_t)seed_path, (uint32_t)(seed_path >> 32));
    MBX_LAUNCH_CHECK("residual_drop");
    return 0;
}
extern "C" int mbx_grad_drop(const float* dy, void* dy_t, size_t rows, int C, int rows_per_sample, float p, uint64_t seed, float p_path,
                             uint64_t seed_path, int dtype, void* stream) {
    MBX_CHECK_ARG(dy && dy_t && C > 0 && C % 4 == 0 && rows_per_sample > 0 && p >= 0.f && p < 1.f && p_path >= 0.f && p_path < 1.f,
                  "grad_drop: bad arguments");
    if (rows == 0) return 0;
    const size_t want = (rows * (C / 4) + 255) / 256;
    const int grid = (int)(want < 4096 ? want : 4096);
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32), plo = (uint32_t)seed_path, phi = (uint32_t)(seed_path >> 32);
    if (dtype == MBX_BF16)
        hipLaunchKernelGGL((branch_drop_kernel<bf16_t, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, const_cast<float*>(dy), (const float*)nullptr,
                           (bf16_t*)dy_t, rows, C, rows_per_sample, p, lo, hi, p_path, plo, phi);
    else if (dtype == MBX_F32)
        hipLaunchKernelGGL((branch_drop_kernel<float, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, const_cast<float*>(dy), (const float*)nullptr,
                           (float*)dy_t, rows, C, rows_per_sample, p, lo, hi, p_path, plo, phi);
    else
        return mbx_set_error("grad_drop: unknown dtype %d", dtype);
    MBX_LAUNCH_CHECK("grad_drop");
    return 0;
}
