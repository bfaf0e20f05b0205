// Bandwidth-bound kernels of the DSTformer hot path for gfx950: embedding, LayerNorm, adaptive
// fusion, head, weight preparation and the reduction finalizer.  All of them are "one wave64 per
// token row" or flat elementwise kernels with 16-byte vector accesses; per-token reductions use
// wave shuffles (no LDS), parameter-gradient reductions accumulate in registers across a
// grid-stride loop, reduce across the 4 waves of a block through LDS and leave one partial row
// per block that colsum_kernel folds deterministically (no atomics).
#include "mbx_common.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
int mbx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}
extern "C" const char* mbx_last_error(void) { return g_err; }

// ---- per-device launch properties (declared in mbx_common.h) --------------------------------------------------------------------
// table of (kernel address, device) -> largest dynamic-LDS size already granted.  Open addressing over 512 slots, entries are only
// ever added (there are < 100 kernel instantiations x a handful of devices); a slot is claimed by a compare-exchange on its key, its
// size only grows.  A lost race costs one redundant hipFuncSetAttribute, never a missing one.
#include <atomic>
static std::atomic<uintptr_t> g_lds_key[512];
static std::atomic<size_t> g_lds_size[512];
int mbx_set_dyn_lds(const void* kernel, size_t bytes, const char* who) {
    if (bytes > 160 * 1024) return mbx_set_error("%s: needs %zu bytes of LDS (> 160 KiB)", who, bytes);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const uintptr_t key = (reinterpret_cast<uintptr_t>(kernel) << 6) ^ (uintptr_t)(dev + 1);
    int slot = (int)((key * 0x9E3779B97F4A7C15ull) >> 55);      // 9 bits
    for (int probe = 0; probe < 512; ++probe, slot = (slot + 1) & 511) {
        uintptr_t k = g_lds_key[slot].load(std::memory_order_acquire);
        if (k == 0 && g_lds_key[slot].compare_exchange_strong(k, key, std::memory_order_acq_rel)) k = key;
        if (k != key) continue;
        if (g_lds_size[slot].load(std::memory_order_acquire) >= bytes) return 0;
        const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return mbx_set_error("%s: hipFuncSetAttribute(%zu): %s", who, bytes, hipGetErrorString(e));
        size_t cur = g_lds_size[slot].load(std::memory_order_relaxed);
        while (cur < bytes && !g_lds_size[slot].compare_exchange_weak(cur, bytes, std::memory_order_release)) { }
        return 0;
    }
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);      // table full: as before
    return e == hipSuccess ? 0 : mbx_set_error("%s: hipFuncSetAttribute(%zu): %s", who, bytes, hipGetErrorString(e));
}
static std::atomic<int> g_cus[64];
int mbx_cu_count() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int c = g_cus[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    g_cus[dev].store(c, std::memory_order_relaxed);
    return c;
}
extern "C" int mbx_version(void) { return 100; }

static inline int clamp_grid(size_t want, int cap) { return (int)(want < (size_t)cap ? (want ? want : 1) : (size_t)cap); }

// ------------------------------------------------------------------------------------------------
// colsum finalize: out[c] = sum_p part[p*stride + col0 + c], fixed summation order (deterministic).
// Vector kernel: a lane owns 4 consecutive columns (one 16-B load per partial row); a block is CL column lanes x
// PL = 256/CL part lanes, every part lane keeps four loads in flight, the part lanes are folded through LDS.
// CL = 64 for wide outputs (weight-gradient tiles: few partial rows, 10^5..10^6 columns), CL = 16 for narrow ones
// (LayerNorm dgamma/dbeta: 10^3 partial rows, 10^3 columns).  Columns >= split go to out1 (two outputs, one launch).
// The scalar kernel below remains for outputs whose width or offset is not a multiple of 4.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <int CL>
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ part, int nparts, size_t stride, int ncols,
                                                      float* __restrict__ out0, float* __restrict__ out1, int split) {
    constexpr int PLW = 64 / CL, PL = 4 * PLW;
    __shared__ float4 red[PL][CL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane % CL, pl = wave * PLW + lane / CL;
    const int c = (blockIdx.x * CL + cl) * 4;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    if (c < ncols) {
        const float* p = part + c;
        int i = pl;
        for (; i + 3 * PL < nparts; i += 4 * PL) {
            const float4 v0 = ld4(p + (size_t)i * stride), v1 = ld4(p + (size_t)(i + PL) * stride);
            const float4 v2 = ld4(p + (size_t)(i + 2 * PL) * stride), v3 = ld4(p + (size_t)(i + 3 * PL) * stride);
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; i < nparts; i += PL) a0 += ld4(p + (size_t)i * stride);
    }
    red[pl][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (pl == 0 && c < ncols) {
        float4 t = red[0][cl];
#pragma unroll
        for (int q = 1; q < PL; ++q) t += red[q][cl];
        float* o = c < split ? out0 + c : out1 + (c - split);
        if ((reinterpret_cast<uintptr_t>(o) & 15) == 0) {
            *reinterpret_cast<float4*>(o) = t;
        } else {
            o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
        }
    }
}
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ part, int nparts, int stride, int col0,
                                                     int ncols, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < ncols) {
        const float* p = part + col0 + c;
        int i = grp;
        for (; i + 12 < nparts; i += 16) {
            s0 += p[(size_t)i * stride];
            s1 += p[(size_t)(i + 4) * stride];
            s2 += p[(size_t)(i + 8) * stride];
            s3 += p[(size_t)(i + 12) * stride];
        }
        for (; i < nparts; i += 4) s0 += p[(size_t)i * stride];
    }
    red[grp][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && c < ncols) out[c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
// two outputs: columns [0, split) -> out0, [split, ncols) -> out1 (pass split = ncols, out1 = nullptr for one output)
static int launch_colsum2(const float* part, int nparts, int stride, int col0, int ncols, float* out0, float* out1, int split,
                          hipStream_t s) {
    const bool vec = ncols % 4 == 0 && col0 % 4 == 0 && stride % 4 == 0 && split % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(part + col0) & 15) == 0;
    if (!vec) {
        hipLaunchKernelGGL(colsum_kernel, dim3((split + 63) / 64), dim3(256), 0, s, part, nparts, stride, col0, split, out0);
        if (ncols > split)
            hipLaunchKernelGGL(colsum_kernel, dim3((ncols - split + 63) / 64), dim3(256), 0, s, part, nparts, stride, col0 + split,
                               ncols - split, out1);
    } else if (ncols / 4 >= 64 * 512) {
        hipLaunchKernelGGL(colsum4_kernel<64>, dim3((ncols / 4 + 63) / 64), dim3(256), 0, s, part + col0, nparts, (size_t)stride, ncols,
                           out0, out1, split);
    } else {
        hipLaunchKernelGGL(colsum4_kernel<16>, dim3((ncols / 4 + 15) / 16), dim3(256), 0, s, part + col0, nparts, (size_t)stride, ncols,
                           out0, out1, split);
    }
    MBX_LAUNCH_CHECK("colsum");
    return 0;
}
int mbx_launch_colsum(const float* part, int nparts, int stride, int col0, int ncols, float* out, hipStream_t s) {
    return launch_colsum2(part, nparts, stride, col0, ncols, out, nullptr, ncols, s);
}

// ------------------------------------------------------------------------------------------------
// weight preparation: fp32 [N,K] -> T [N,K] and T [K,N]
// ------------------------------------------------------------------------------------------------
// bf16 remainder of an fp32 value: lo = bf16(v - float(bf16(v)))  (the 'lo' plane of the bf16x3 split)
struct BfLo {};
template <> struct Cvt<BfLo> {
    static __device__ __forceinline__ bf16_t from_f(float v) { return f2bf(v - bf2f(f2bf(v))); }
};
template <typename T> struct PrepOut { typedef T type; };
template <> struct PrepOut<BfLo> { typedef bf16_t type; };
template <typename TT>
__global__ __launch_bounds__(256) void prep_weights_kernel(const int64_t* __restrict__ desc) {
    typedef typename PrepOut<TT>::type T;
    __shared__ float tile[32][33];
    const int64_t* d = desc + (size_t)blockIdx.y * 5;
    const float* src = reinterpret_cast<const float*>(d[0]);
    T* dst_n = reinterpret_cast<T*>(d[1]);
    T* dst_t = reinterpret_cast<T*>(d[2]);
    const int N = (int)d[3], K = (int)d[4];
    const int tk = (K + 31) / 32, tn = (N + 31) / 32;
    if ((int)blockIdx.x >= tk * tn) return;
    const int n0 = (blockIdx.x / tk) * 32, k0 = (blockIdx.x % tk) * 32;
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ry + 8 * i;
        float v = 0.f;
        if (n0 + r < N && k0 + cx < K) {
            v = src[(size_t)(n0 + r) * K + k0 + cx];
            if (dst_n) dst_n[(size_t)(n0 + r) * K + k0 + cx] = Cvt<TT>::from_f(v);
        }
        tile[r][cx] = v;
    }
    __syncthreads();
    if (dst_t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = ry + 8 * i;  // k index within the tile
            if (k0 + r < K && n0 + cx < N) dst_t[(size_t)(k0 + r) * N + n0 + cx] = Cvt<TT>::from_f(tile[cx][r]);
        }
    }
}
extern "C" int mbx_prep_weights(const int64_t* desc, int n_desc, int max_n, int max_k, int dtype, void* stream) {
    MBX_CHECK_ARG(desc && n_desc > 0 && max_n > 0 && max_k > 0, "prep_weights: bad arguments");
    dim3 grid(((max_n + 31) / 32) * ((max_k + 31) / 32), n_desc);
    if (dtype == MBX_BF16)
        hipLaunchKernelGGL(prep_weights_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, desc);
    else if (dtype == MBX_F32)
        hipLaunchKernelGGL(prep_weights_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, desc);
    else if (dtype == MBX_BF16_LO)
        hipLaunchKernelGGL(prep_weights_kernel<BfLo>, grid, dim3(256), 0, (hipStream_t)stream, desc);
    else
        return mbx_set_error("prep_weights: unknown dtype %d", dtype);
    MBX_LAUNCH_CHECK("prep_weights");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// GELU of a stored pre-activation: g = gelu(u).  Only used by the low-memory (recompute) mode of the engine, which keeps the
// pre-activation `u` of an MLP and rebuilds the post-activation in backward instead of saving both (nn.GELU, DSTformer.py:70).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ u, T* __restrict__ g, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float v[4];
        load4<T>(u + i * 4, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        store4<T>(g + i * 4, v);
    }
}
extern "C" int mbx_gelu_fwd(const void* u, void* g, size_t n, int dtype, void* stream) {
    MBX_CHECK_ARG(u && g && n % 4 == 0, "gelu_fwd: bad arguments");
    if (n == 0) return 0;
    const int grid = clamp_grid((n / 4 + 255) / 256, 256 * 16);
    if (dtype == MBX_BF16)
        hipLaunchKernelGGL(gelu_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)u, (bf16_t*)g, n / 4);
    else if (dtype == MBX_F32)
        hipLaunchKernelGGL(gelu_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)u, (float*)g, n / 4);
    else
        return mbx_set_error("gelu_fwd: unknown dtype %d", dtype);
    MBX_LAUNCH_CHECK("gelu_fwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// bf16x3 operand split: x (fp32) -> hi = bf16(x), lo = bf16(x - hi); x = hi + lo up to 2^-16 relative
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                                                        size_t n4) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n4; idx += (size_t)gridDim.x * 256) {
        float v[4], h[4], l[4];
        load4<float>(x + idx * 4, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = bf2f(f2bf(v[e])); l[e] = v[e] - h[e]; }
        store4<bf16_t>(hi + idx * 4, h);
        store4<bf16_t>(lo + idx * 4, l);
    }
}
extern "C" int mbx_split_bf16(const float* x, void* hi, void* lo, size_t n, void* stream) {
    MBX_CHECK_ARG(x && hi && lo, "split_bf16: null pointer");
    MBX_CHECK_ARG(n % 4 == 0, "split_bf16: n %% 4 != 0");
    if (n == 0) return 0;
    const int grid = clamp_grid((n / 4 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(split_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)hi, (bf16_t*)lo, n / 4);
    MBX_LAUNCH_CHECK("split_bf16");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// embedding forward: h[m, c] = sum_k x[m,k] w[c,k] + b[c] + pos[j,c] + temp[t,c]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, const float* __restrict__ pos,
                                                        const float* __restrict__ temp, float* __restrict__ h, int M,
                                                        int T, int J, int Din, int C) {
    const int c4n = C >> 2;
    const size_t total = (size_t)M * c4n;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int m = (int)(idx / c4n), c = (int)(idx % c4n) * 4;
        const int j = m % J, t = (m / J) % T;
        const float4 bb = *reinterpret_cast<const float4*>(b + c);
        const float4 pp = *reinterpret_cast<const float4*>(pos + (size_t)j * C + c);
        const float4 tt = *reinterpret_cast<const float4*>(temp + (size_t)t * C + c);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int k = 0; k < Din; ++k) {
            const float xv = x[(size_t)m * Din + k];
            a0 = fmaf(xv, w[(size_t)(c + 0) * Din + k], a0);
            a1 = fmaf(xv, w[(size_t)(c + 1) * Din + k], a1);
            a2 = fmaf(xv, w[(size_t)(c + 2) * Din + k], a2);
            a3 = fmaf(xv, w[(size_t)(c + 3) * Din + k], a3);
        }
        // same association as the reference: ((xW + b) + pos) + temp
        *reinterpret_cast<float4*>(h + (size_t)m * C + c) =
            make_float4(((a0 + bb.x) + pp.x) + tt.x, ((a1 + bb.y) + pp.y) + tt.y, ((a2 + bb.z) + pp.z) + tt.z,
                        ((a3 + bb.w) + pp.w) + tt.w);
    }
}
// Row-chunked form for the model sizes (C/4 a power of two in [16, 256], dim_in <= 4): a thread owns ONE channel quad for a
// contiguous chunk of tokens, so its weight rows and bias live in registers, (j, t) advance incrementally (no divisions in
// the loop) and four tokens are in flight per thread.  Same association of the sum as above.  0.43 -> 0.15 ms at 64 clips.
#define EMBF_MAXDIN 4
__global__ __launch_bounds__(256) void embed_fwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, const float* __restrict__ pos,
                                                             const float* __restrict__ temp, float* __restrict__ h, int M,
                                                             int T, int J, int Din, int C, int chunk) {
    const int c4n = C >> 2, rpp = 256 / c4n;                 // rows per pass of the block (<= 16 <= J checked on the host)
    const int c = ((int)threadIdx.x % c4n) * 4, rsub = (int)threadIdx.x / c4n;
    float wv[4][EMBF_MAXDIN];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < EMBF_MAXDIN; ++k) wv[e][k] = k < Din ? w[(size_t)(c + e) * Din + k] : 0.f;
    const float4 bb = *reinterpret_cast<const float4*>(b + c);
    const int end = min(M, ((int)blockIdx.x + 1) * chunk);
    int m = (int)blockIdx.x * chunk + rsub;
    int j = m % J, t = (m / J) % T;
    for (; m < end; m += 4 * rpp) {
        float xv[4][EMBF_MAXDIN];
        float4 pp[4], tt[4];
        int jj = j, tq = t;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int mu = min(m + u * rpp, M - 1);
#pragma unroll
            for (int k = 0; k < EMBF_MAXDIN; ++k) xv[u][k] = k < Din ? x[(size_t)mu * Din + k] : 0.f;
            pp[u] = *reinterpret_cast<const float4*>(pos + (size_t)jj * C + c);
            tt[u] = *reinterpret_cast<const float4*>(temp + (size_t)tq * C + c);
            jj += rpp;
            if (jj >= J) { jj -= J; tq = tq + 1 == T ? 0 : tq + 1; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < EMBF_MAXDIN; ++k)
                if (k < Din) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = fmaf(xv[u][k], wv[e][k], a[e]);
                }
            if (m + u * rpp < end)
                *reinterpret_cast<float4*>(h + (size_t)(m + u * rpp) * C + c) =
                    make_float4(((a[0] + bb.x) + pp[u].x) + tt[u].x, ((a[1] + bb.y) + pp[u].y) + tt[u].y,
                                ((a[2] + bb.z) + pp[u].z) + tt[u].z, ((a[3] + bb.w) + pp[u].w) + tt[u].w);
        }
        j = jj;
        t = tq;
    }
}
extern "C" int mbx_embed_fwd(const float* x, const float* w, const float* b, const float* pos, const float* temp,
                             float* h, int B, int T, int J, int Din, int C, void* stream) {
    MBX_CHECK_ARG(x && w && b && pos && temp && h, "embed_fwd: null pointer");
    MBX_CHECK_ARG(B > 0 && T > 0 && J > 0 && Din > 0 && C > 0 && C % 4 == 0, "embed_fwd: bad shape (C %% 4 != 0?)");
    const int M = B * T * J;
    const int c4n = C / 4;
    if (c4n >= 16 && c4n <= 256 && (c4n & (c4n - 1)) == 0 && Din <= EMBF_MAXDIN && 256 / c4n <= J) {
        const int rpp = 256 / c4n;
        const int blocks = clamp_grid((M + 4 * rpp - 1) / (4 * rpp), 256 * 8);
        int chunk = (M + blocks - 1) / blocks;
        chunk = ((chunk + 4 * rpp - 1) / (4 * rpp)) * (4 * rpp);        // whole passes per block
        const int grid = (M + chunk - 1) / chunk;
        hipLaunchKernelGGL(embed_fwd_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, b, pos, temp, h, M, T, J, Din, C, chunk);
    } else {
        const int grid = clamp_grid(((size_t)M * (C / 4) + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, b, pos, temp, h, M, T, J, Din, C);
    }
    MBX_LAUNCH_CHECK("embed_fwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// embedding backward.  One block per frame index t: a thread owns channel c and walks (j, b);
// dtemp[t,c] is final, dpos / dw / db leave one partial row per t (folded by colsum).
// partial row layout: [J*C | C*Din | C]
// ------------------------------------------------------------------------------------------------
#define EMB_MAXDIN 4
// 256 threads = (C/4 channel quads) x (256 / (C/4) slices of the clip range); 16-byte loads, eight clips in flight per
// thread; the slices are folded through LDS.  (The scalar one-clip-at-a-time walk of round 1 took 1.05 ms at 64 clips.)
// dh_a / dh_b (round 5): the incoming gradient as the sum of two bf16 tensors (the input gradients of the two Blocks of level 0, as
// their row-owner LayerNorm-backward kernels leave them) instead of one fp32 tensor
__device__ __forceinline__ float4 embed_dh4(const float* __restrict__ dh, const bf16_t* __restrict__ dh_a, const bf16_t* __restrict__ dh_b, size_t o) {
    if (dh_a == nullptr) return *reinterpret_cast<const float4*>(dh + o);
    float u[4], v[4];
    load4<bf16_t>(dh_a + o, u);
    load4<bf16_t>(dh_b + o, v);
    return make_float4(u[0] + v[0], u[1] + v[1], u[2] + v[2], u[3] + v[3]);
}
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                        float* __restrict__ dtemp, float* __restrict__ part, int B, int T,
                                                        int J, int Din, int C, const bf16_t* __restrict__ dh_a = nullptr,
                                                        const bf16_t* __restrict__ dh_b = nullptr) {
    __shared__ float4 red[256][1 + EMB_MAXDIN];
    const int t = blockIdx.x;
    const int stride = J * C + C * Din + C;
    float* prow = part + (size_t)t * stride;
    const int nq = C / 4, ns = max(256 / nq, 1);          // channel quads, clip slices (C % 4 == 0 checked on the host)
    for (int q0 = 0; q0 < nq; q0 += 256) {                // C <= 1024: one pass
        const int q = q0 + (int)threadIdx.x % min(nq, 256), sl = (int)threadIdx.x / min(nq, 256);
        const bool act = q < nq && sl < ns;
        const int c = q * 4;
        float4 at = make_float4(0.f, 0.f, 0.f, 0.f), aw[EMB_MAXDIN];
#pragma unroll
        for (int k = 0; k < EMB_MAXDIN; ++k) aw[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < J; ++j) {
            float4 ap = make_float4(0.f, 0.f, 0.f, 0.f);
            if (act) {
                for (int b0 = sl; b0 < B; b0 += 8 * ns) {
                    float4 g[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int b = min(b0 + u * ns, B - 1);
                        g[u] = embed_dh4(dh, dh_a, dh_b, (((size_t)b * T + t) * J + j) * C + c);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int b = b0 + u * ns;
                        const float on = b < B ? 1.f : 0.f;
                        const size_t m = ((size_t)min(b, B - 1) * T + t) * J + j;
                        ap.x = fmaf(on, g[u].x, ap.x); ap.y = fmaf(on, g[u].y, ap.y); ap.z = fmaf(on, g[u].z, ap.z); ap.w = fmaf(on, g[u].w, ap.w);
#pragma unroll
                        for (int k = 0; k < EMB_MAXDIN; ++k)
                            if (k < Din) {
                                const float xv = on * x[m * Din + k];
                                aw[k].x = fmaf(g[u].x, xv, aw[k].x); aw[k].y = fmaf(g[u].y, xv, aw[k].y);
                                aw[k].z = fmaf(g[u].z, xv, aw[k].z); aw[k].w = fmaf(g[u].w, xv, aw[k].w);
                            }
                    }
                }
            }
            // fold the clip slices of this joint (fixed order), slice 0 writes the dpos partial
            red[threadIdx.x][0] = ap;
            __syncthreads();
            if (act && sl == 0) {
                for (int s2 = 1; s2 < ns; ++s2) {
                    const float4 o = red[threadIdx.x + s2 * nq][0];
                    ap.x += o.x; ap.y += o.y; ap.z += o.z; ap.w += o.w;
                }
                *reinterpret_cast<float4*>(prow + (size_t)j * C + c) = ap;
                at.x += ap.x; at.y += ap.y; at.z += ap.z; at.w += ap.w;
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < EMB_MAXDIN; ++k) red[threadIdx.x][1 + k] = aw[k];
        __syncthreads();
        if (act && sl == 0) {
            *reinterpret_cast<float4*>(dtemp + (size_t)t * C + c) = at;
            *reinterpret_cast<float4*>(prow + (size_t)J * C + (size_t)C * Din + c) = at;
#pragma unroll
            for (int k = 0; k < EMB_MAXDIN; ++k)
                if (k < Din) {
                    float4 w4 = aw[k];
                    for (int s2 = 1; s2 < ns; ++s2) {
                        const float4 o = red[threadIdx.x + s2 * nq][1 + k];
                        w4.x += o.x; w4.y += o.y; w4.z += o.z; w4.w += o.w;
                    }
                    float* d = prow + (size_t)J * C + (size_t)c * Din + k;
                    d[0] = w4.x; d[Din] = w4.y; d[2 * Din] = w4.z; d[3 * Din] = w4.w;
                }
        }
        __syncthreads();
    }
}
// dx[m,k] = sum_c dh[m,c] w[c,k]  (one wave per token)
__global__ __launch_bounds__(256) void embed_bwd_dx_kernel(const float* __restrict__ dh, const float* __restrict__ w,
                                                           float* __restrict__ dx, int M, int Din, int C,
                                                           const bf16_t* __restrict__ dh_a = nullptr, const bf16_t* __restrict__ dh_b = nullptr) {
    const int lane = threadIdx.x & 63;
    for (int m = blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += gridDim.x * 4) {
        float a[EMB_MAXDIN] = {0.f, 0.f, 0.f, 0.f};
        for (int c = lane; c < C; c += 64) {
            const float g = dh_a == nullptr ? dh[(size_t)m * C + c] : bf2f(dh_a[(size_t)m * C + c]) + bf2f(dh_b[(size_t)m * C + c]);
#pragma unroll
            for (int k = 0; k < EMB_MAXDIN; ++k)
                if (k < Din) a[k] = fmaf(g, w[(size_t)c * Din + k], a[k]);
        }
#pragma unroll
        for (int k = 0; k < EMB_MAXDIN; ++k) {
            const float s = wave_sum(a[k]);
            if (lane == 0 && k < Din) dx[(size_t)m * Din + k] = s;
        }
    }
}
extern "C" size_t mbx_embed_bwd_ws(int T, int J, int Din, int C) { return (size_t)T * ((size_t)J * C + (size_t)C * Din + C) * sizeof(float); }
static int launch_embed_bwd(const float* dh, const bf16_t* dh_a, const bf16_t* dh_b, const float* x, const float* w, float* dw, float* db,
                            float* dpos, float* dtemp, float* dx, int B, int T, int J, int Din, int C, void* ws, void* stream) {
    MBX_CHECK_ARG((dh || (dh_a && dh_b)) && x && w && dw && db && dpos && dtemp && ws, "embed_bwd: null pointer");
    MBX_CHECK_ARG(Din <= EMB_MAXDIN, "embed_bwd: dim_in %d > %d unsupported", Din, EMB_MAXDIN);
    MBX_CHECK_ARG(B > 0 && T > 0 && J > 0 && C > 0 && C % 4 == 0, "embed_bwd: bad shape (C %% 4 != 0?)");
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)ws;
    const int stride = J * C + C * Din + C;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(T), dim3(256), 0, s, dh, x, dtemp, part, B, T, J, Din, C, dh_a, dh_b);
    MBX_LAUNCH_CHECK("embed_bwd");
    if (mbx_launch_colsum(part, T, stride, 0, J * C, dpos, s)) return 1;
    if (mbx_launch_colsum(part, T, stride, J * C, C * Din, dw, s)) return 1;
    if (mbx_launch_colsum(part, T, stride, J * C + C * Din, C, db, s)) return 1;
    if (dx) {
        const int M = B * T * J;
        hipLaunchKernelGGL(embed_bwd_dx_kernel, dim3(clamp_grid((M + 3) / 4, 2048)), dim3(256), 0, s, dh, w, dx, M, Din, C, dh_a, dh_b);
        MBX_LAUNCH_CHECK("embed_bwd_dx");
    }
    return 0;
}
extern "C" int mbx_embed_bwd(const float* dh, const float* x, const float* w, float* dw, float* db, float* dpos,
                             float* dtemp, float* dx, int B, int T, int J, int Din, int C, void* ws, void* stream) {
    MBX_CHECK_ARG(dh, "embed_bwd: null pointer");
    return launch_embed_bwd(dh, nullptr, nullptr, x, w, dw, db, dpos, dtemp, dx, B, T, J, Din, C, ws, stream);
}
// the same with dh = dh_a + dh_b, both bf16 [B*T*J, C]
extern "C" int mbx_embed_bwd_pair(const void* dh_a, const void* dh_b, const float* x, const float* w, float* dw, float* db, float* dpos,
                                  float* dtemp, float* dx, int B, int T, int J, int Din, int C, void* ws, void* stream) {
    MBX_CHECK_ARG(dh_a && dh_b, "embed_bwd_pair: null pointer");
    return launch_embed_bwd(nullptr, (const bf16_t*)dh_a, (const bf16_t*)dh_b, x, w, dw, db, dpos, dtemp, dx, B, T, J, Din, C, ws, stream);
}

// ------------------------------------------------------------------------------------------------
// row-per-wave helpers: a lane owns VPL float4 slots at channel c = (k*64 + lane)*4
// ------------------------------------------------------------------------------------------------
#define ROW_LOOP(k) _Pragma("unroll") for (int k = 0; k < VPL; ++k)
#define ROW_C(k) (((k) * 64 + lane) * 4)

static inline int vpl_for(int C) { return C <= 256 ? 1 : C <= 512 ? 2 : C <= 1024 ? 4 : C <= 2048 ? 8 : 0; }
#define DISPATCH_VPL(vpl, ...)           \
    switch (vpl) {                       \
        case 1: { constexpr int VPL = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int VPL = 2; __VA_ARGS__; } break; \
        case 4: { constexpr int VPL = 4; __VA_ARGS__; } break; \
        case 8: { constexpr int VPL = 8; __VA_ARGS__; } break; \
        default: return mbx_set_error("row kernel: channel count unsupported (need C %% 4 == 0 and C <= 2048)"); \
    }

// ------------------------------------------------------------------------------------------------
// LayerNorm forward / backward.  Both are HBM-bound row kernels (one wave per row).  To keep enough bytes in flight
// a wave works on TWO rows per iteration and issues every load of both rows before the first use; FULL (C == VPL*256,
// the model sizes 256/512/1024) drops the per-slot bounds tests that would otherwise split the load cluster.
// The grid is sized so that all workgroups are resident at once (<= 4 x 256 CUs).
// ------------------------------------------------------------------------------------------------
#define LN_BLOCKS 1024
template <typename T> struct Raw4;
template <> struct Raw4<float> {
    typedef float4 type;
    static __device__ __forceinline__ void unpack(const float4& r, float (&v)[4]) { v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w; }
};
template <> struct Raw4<bf16_t> {
    typedef uint2 type;
    static __device__ __forceinline__ void unpack(const uint2& r, float (&v)[4]) {
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
        v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    }
};
__device__ __forceinline__ void unpack4(const float4& r, float (&v)[4]) { v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w; }

template <int VPL, bool FULL>
__device__ __forceinline__ void ln_fwd_load(const float* __restrict__ x, int row, int C, int lane, float4 (&raw)[VPL]) {
    ROW_LOOP(k) {
        const int c = ROW_C(k);
        raw[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FULL || c < C) raw[k] = ld4(x + (size_t)row * C + c);
    }
}
template <typename T, int VPL, bool FULL>
__device__ __forceinline__ void ln_fwd_row(const float4 (&raw)[VPL], const float (&g)[VPL][4], const float (&bt)[VPL][4], float eps,
                                           float invC, T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                           int row, int C, int lane, bf16_t* __restrict__ y_lo = nullptr) {
    float v[VPL][4];
    float s = 0.f;
    ROW_LOOP(k) {
        unpack4(raw[k], v[k]);
        s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);     // slots past C hold zeros
    }
    const float mu = wave_sum(s) * invC;
    float q = 0.f;
    ROW_LOOP(k) {
        const int c = ROW_C(k);
        if (FULL || c < C) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[k][i] -= mu; q = fmaf(v[k][i], v[k][i], q); }
        }
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) * invC + eps);
    ROW_LOOP(k) {
        const int c = ROW_C(k);
        if (FULL || c < C) {
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = fmaf(v[k][i] * rs, g[k][i], bt[k][i]);
            if (sizeof(T) == 4 && y_lo)      // fp32-class mode: the GEMM operand planes of the bf16x3 split instead of fp32 (y = hi plane)
                store4_planes(reinterpret_cast<bf16_t*>(y) + (size_t)row * C + c, y_lo + (size_t)row * C + c, o);
            else
                store4<T>(y + (size_t)row * C + c, o);
        }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}
template <typename T, int VPL, bool FULL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, int C,
                                                     bf16_t* __restrict__ y_lo) {
    const int lane = threadIdx.x & 63;
    float g[VPL][4], bt[VPL][4];
    ROW_LOOP(k) {
        const int c = ROW_C(k);
#pragma unroll
        for (int i = 0; i < 4; ++i) { g[k][i] = 0.f; bt[k][i] = 0.f; }
        if (gamma == nullptr) {     // plain normalisation (the affine part lives in the consumer's folded weights)
#pragma unroll
            for (int i = 0; i < 4; ++i) g[k][i] = 1.f;
        } else if (FULL || c < C) { load4<float>(gamma + c, g[k]); load4<float>(beta + c, bt[k]); }
    }
    const float invC = 1.0f / (float)C;
    const int step = gridDim.x * 4;
    for (int row0 = blockIdx.x * 4 + (threadIdx.x >> 6); row0 < M; row0 += 2 * step) {
        const int row1 = row0 + step;
        float4 ra[VPL], rb[VPL];
        ln_fwd_load<VPL, FULL>(x, row0, C, lane, ra);
        if (row1 < M) ln_fwd_load<VPL, FULL>(x, row1, C, lane, rb);
        ln_fwd_row<T, VPL, FULL>(ra, g, bt, eps, invC, y, mean, rstd, row0, C, lane, y_lo);
        if (row1 < M) ln_fwd_row<T, VPL, FULL>(rb, g, bt, eps, invC, y, mean, rstd, row1, C, lane, y_lo);
    }
}
#define LN_FWD_LAUNCH(TT, FULLV)                                                                                             \
    DISPATCH_VPL(vpl, hipLaunchKernelGGL((ln_fwd_kernel<TT, VPL, FULLV>), dim3(grid), dim3(256), 0, s, x, gamma, beta, eps, (TT*)y, \
                                         mean, rstd, M, C, (bf16_t*)y_lo))
static int launch_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, void* y, void* y_lo, float* mean,
                                float* rstd, int M, int C, int dtype, void* stream);
extern "C" int mbx_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, void* y, float* mean,
                                 float* rstd, int M, int C, int dtype, void* stream) {
    return launch_layernorm_fwd(x, gamma, beta, eps, y, nullptr, mean, rstd, M, C, dtype, stream);
}
extern "C" int mbx_layernorm_fwd_planes(const float* x, const float* gamma, const float* beta, float eps, void* y_hi, void* y_lo,
                                        float* mean, float* rstd, int M, int C, void* stream) {
    MBX_CHECK_ARG(y_lo, "layernorm_fwd_planes: null pointer");
    return launch_layernorm_fwd(x, gamma, beta, eps, y_hi, y_lo, mean, rstd, M, C, MBX_F32, stream);
}
static int launch_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, void* y, void* y_lo, float* mean,
                                float* rstd, int M, int C, int dtype, void* stream) {
    MBX_CHECK_ARG(x && y && mean && rstd, "layernorm_fwd: null pointer");
    MBX_CHECK_ARG((gamma && beta) || (!gamma && !beta), "layernorm_fwd: gamma and beta come together (both NULL = plain normalisation)");
    MBX_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "layernorm_fwd: bad shape M=%d C=%d", M, C);
    MBX_CHECK_ARG(dtype == MBX_BF16 || dtype == MBX_F32, "layernorm_fwd: unknown dtype %d", dtype);
    const int grid = clamp_grid((M + 3) / 4, LN_BLOCKS);
    const int vpl = vpl_for(C);
    const bool full = vpl > 0 && C == vpl * 256;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MBX_BF16) {
        if (full) { LN_FWD_LAUNCH(bf16_t, true); } else { LN_FWD_LAUNCH(bf16_t, false); }
    } else {
        if (full) { LN_FWD_LAUNCH(float, true); } else { LN_FWD_LAUNCH(float, false); }
    }
    MBX_LAUNCH_CHECK("layernorm_fwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// block-level fold of per-wave register partials: wave w writes acc to lds[w][...], then the block
// sums the 4 waves and writes one partial row.  n = floats per wave.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_fold_store(const float* lds /*[4][n]*/, int n, float* __restrict__ dst) {
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = (lds[i] + lds[n + i]) + (lds[2 * n + i] + lds[3 * n + i]);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward (+ residual-gradient add, + T copy for the next GEMM)
// partial row layout: [dgamma C | dbeta C]
// ------------------------------------------------------------------------------------------------
template <typename T, int VPL, bool FULL, bool RES>
__device__ __forceinline__ void ln_bwd_load(const T* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ dres,
                                            const float* __restrict__ mean, const float* __restrict__ rstd, int row, int C, int lane,
                                            typename Raw4<T>::type (&rd)[VPL], float4 (&rx)[VPL], float4 (&rr)[VPL], float& mu,
                                            float& rs) {
    mu = mean[row];
    rs = rstd[row];
    ROW_LOOP(k) {
        const int c = ROW_C(k);
        if (FULL || c < C) {
            const size_t o = (size_t)row * C + c;
            rd[k] = *reinterpret_cast<const typename Raw4<T>::type*>(dy + o);
            rx[k] = ld4(x + o);
            if (RES) rr[k] = ld4(dres + o);
        }
    }
}
template <typename T, int VPL, bool FULL, bool RES>
__device__ __forceinline__ void ln_bwd_row(const typename Raw4<T>::type (&rd)[VPL], const float4 (&rx)[VPL], const float4 (&rr)[VPL],
                                           float mu, float rs, const float (&g)[VPL][4], float (&ag)[VPL][4], float (&ab)[VPL][4],
                                           float invC, const float* __restrict__ extra, float* __restrict__ dx, T* __restrict__ dx_t,
                                           int row, int C, int lane, bf16_t* __restrict__ dx_lo = nullptr) {
    float d[VPL][4], xh[VPL][4];
    float s1 = 0.f, s2 = 0.f;
    ROW_LOOP(k) {
        const int c = ROW_C(k);
        if (FULL || c < C) {
            Raw4<T>::unpack(rd[k], d[k]);
            unpack4(rx[k], xh[k]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xh[k][i] = (xh[k][i] - mu) * rs;
                ag[k][i] = fmaf(d[k][i], xh[k][i], ag[k][i]);
                ab[k][i] += d[k][i];
                d[k][i] *= g[k][i];
                s1 += d[k][i];
                s2 = fmaf(d[k][i], xh[k][i], s2);
            }
        }
    }
    s1 = wave_sum(s1) * invC;
    s2 = wave_sum(s2) * invC;
    ROW_LOOP(k) {
        const int c = ROW_C(k);
        if (FULL || c < C) {
            float r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = rs * (d[k][i] - s1 - xh[k][i] * s2);
            if (RES) {
                float t[4];
                unpack4(rr[k], t);
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] += t[i];
            }
            if (extra) {
                float t[4];
                load4<float>(extra + (size_t)row * C + c, t);
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] += t[i];
            }
            store4<float>(dx + (size_t)row * C + c, r);
            if (sizeof(T) == 4 && dx_lo)      // fp32-class mode: dx_t = the operand planes of the next sub-layer's split-operand GEMMs
                store4_planes(reinterpret_cast<bf16_t*>(dx_t) + (size_t)row * C + c, dx_lo + (size_t)row * C + c, r);
            else if (dx_t) store4<T>(dx_t + (size_t)row * C + c, r);
        }
    }
}
template <typename T, int VPL, bool FULL, bool RES>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres,
                                                     const float* __restrict__ extra, float* __restrict__ dx,
                                                     T* __restrict__ dx_t, float* __restrict__ part, int M, int C,
                                                     bf16_t* __restrict__ dx_lo) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[VPL][4], ag[VPL][4], ab[VPL][4];
    ROW_LOOP(k) {
        const int c = ROW_C(k);
#pragma unroll
        for (int i = 0; i < 4; ++i) { ag[k][i] = 0.f; ab[k][i] = 0.f; g[k][i] = 0.f; }
        if (FULL || c < C) load4<float>(gamma + c, g[k]);
    }
    const float invC = 1.0f / (float)C;
    const int step = gridDim.x * 4;
    for (int row0 = blockIdx.x * 4 + wave; row0 < M; row0 += 2 * step) {
        const int row1 = row0 + step;
        typename Raw4<T>::type da[VPL], db[VPL];
        float4 xa[VPL], xb[VPL], ra[VPL], rb[VPL];
        float mua, rsa, mub, rsb;
        ln_bwd_load<T, VPL, FULL, RES>(dy, x, dres, mean, rstd, row0, C, lane, da, xa, ra, mua, rsa);
        if (row1 < M) ln_bwd_load<T, VPL, FULL, RES>(dy, x, dres, mean, rstd, row1, C, lane, db, xb, rb, mub, rsb);
        ln_bwd_row<T, VPL, FULL, RES>(da, xa, ra, mua, rsa, g, ag, ab, invC, extra, dx, dx_t, row0, C, lane, dx_lo);
        if (row1 < M) ln_bwd_row<T, VPL, FULL, RES>(db, xb, rb, mub, rsb, g, ag, ab, invC, extra, dx, dx_t, row1, C, lane, dx_lo);
    }
    float* mine = lds + (size_t)wave * 2 * C;
    ROW_LOOP(k) {
        const int c = ROW_C(k);
        if (FULL || c < C) { store4<float>(mine + c, ag[k]); store4<float>(mine + C + c, ab[k]); }
    }
    __syncthreads();
    block_fold_store(lds, 2 * C, part + (size_t)blockIdx.x * 2 * C);
}
#define LN_BWD_LAUNCH(TT, FULLV, RESV)                                                                                        \
    DISPATCH_VPL(vpl, hipLaunchKernelGGL((ln_bwd_kernel<TT, VPL, FULLV, RESV>), dim3(grid), dim3(256), shm, s, (const TT*)dy, x, mean, \
                                         rstd, gamma, dres, extra, dx, (TT*)dx_t, part, M, C, (bf16_t*)dx_lo))
#define LN_BWD_LAUNCH_T(TT)                                                            \
    if (full && dres) { LN_BWD_LAUNCH(TT, true, true); }                               \
    else if (full) { LN_BWD_LAUNCH(TT, true, false); }                                 \
    else if (dres) { LN_BWD_LAUNCH(TT, false, true); }                                 \
    else { LN_BWD_LAUNCH(TT, false, false); }
extern "C" size_t mbx_layernorm_bwd_ws(int C) { return (size_t)LN_BLOCKS * 2 * C * sizeof(float); }
static int launch_layernorm_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                const float* dres, const float* extra, float* dx, void* dx_t, void* dx_lo, float* dgamma, float* dbeta,
                                int M, int C, int dtype, void* ws, void* stream);
extern "C" int mbx_layernorm_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                 const float* dres, const float* extra, float* dx, void* dx_t, float* dgamma, float* dbeta,
                                 int M, int C, int dtype, void* ws, void* stream) {
    return launch_layernorm_bwd(dy, x, mean, rstd, gamma, dres, extra, dx, dx_t, nullptr, dgamma, dbeta, M, C, dtype, ws, stream);
}
extern "C" int mbx_layernorm_bwd_planes(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                        const float* dres, const float* extra, float* dx, void* dx_hi, void* dx_lo, float* dgamma,
                                        float* dbeta, int M, int C, void* ws, void* stream) {
    MBX_CHECK_ARG(dx_hi && dx_lo, "layernorm_bwd_planes: null pointer");
    return launch_layernorm_bwd(dy, x, mean, rstd, gamma, dres, extra, dx, dx_hi, dx_lo, dgamma, dbeta, M, C, MBX_F32, ws, stream);
}
static int launch_layernorm_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                const float* dres, const float* extra, float* dx, void* dx_t, void* dx_lo, float* dgamma, float* dbeta,
                                int M, int C, int dtype, void* ws, void* stream) {
    MBX_CHECK_ARG(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && ws, "layernorm_bwd: null pointer");
    MBX_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "layernorm_bwd: bad shape M=%d C=%d", M, C);
    MBX_CHECK_ARG(dtype == MBX_BF16 || dtype == MBX_F32, "layernorm_bwd: unknown dtype %d", dtype);
    const int grid = clamp_grid((M + 3) / 4, LN_BLOCKS);
    const size_t shm = (size_t)4 * 2 * C * sizeof(float);
    const int vpl = vpl_for(C);
    const bool full = vpl > 0 && C == vpl * 256;
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)ws;
    if (dtype == MBX_BF16) {
        LN_BWD_LAUNCH_T(bf16_t)
    } else {
        LN_BWD_LAUNCH_T(float)
    }
    MBX_LAUNCH_CHECK("layernorm_bwd");
    return launch_colsum2(part, grid, 2 * C, 0, 2 * C, dgamma, dbeta, C, s);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm folding (round 3, bf16 path).  A LayerNorm that only feeds one Linear (every norm1 / norm2 of a Block:
// DSTformer.py:241-249 -> Attention.qkv :143 / MLP.fc1 :80) is split into the plain normalisation xhat = (x - mean) rstd,
// which the LayerNorm kernel writes as the GEMM operand, and its affine part, which moves into the Linear:
//     y = (xhat g + b_ln) W^T + b = xhat (W diag g)^T + (b + W b_ln) = xhat W'^T + b'.
// What that buys is in BACKWARD: with s[n] = sum_k W'[n,k],
//     c1[m] = mean_k dxhat[m,k]           = (1/C) sum_n dY[m,n] s[n]
//     c2[m] = mean_k dxhat[m,k] xhat[m,k] = (1/C) sum_n dY[m,n] (Y[m,n] - b'[n])        (Y - b' = xhat W'^T)
// are row dots over quantities the kernel that PRODUCES dY already holds (attention backward: dqkv and qkv; GELU' epilogue:
// du and u), so the LayerNorm backward  dx = dres + rstd (dxhat - c1 - xhat c2)  needs no row reduction of its own and runs
// as the epilogue of the dX GEMM (mbx_gemm_nt_lnbwd): the stand-alone LayerNorm-backward pass (16 B per residual element,
// 40 launches, 12 % of the round-2 step) and the bf16 round trip of d(xn) disappear.  The parameter gradients follow from the
// folded weight gradient dW' = dY^T xhat without touching the tokens:
//     dW[n,k] = g[k] dW'[n,k] + db'[n] b_ln[k],   dg[k] = sum_n W[n,k] dW'[n,k],   db_ln[k] = sum_n W[n,k] db'[n],   db = db'.
// ------------------------------------------------------------------------------------------------
// descriptor record (10 x int64): {w f32 [N,K], bias f32 [N] or 0, gamma [K], beta [K], dst_n bf16 [N,K], dst_t bf16 [K,N] or 0,
//                                  bias_f f32 [N], rsum f32 [N], N, K}
__global__ __launch_bounds__(256) void fold_weights_kernel(const int64_t* __restrict__ desc) {
    __shared__ float tile[32][33];
    const int64_t* d = desc + (size_t)blockIdx.y * 10;
    const float* src = reinterpret_cast<const float*>(d[0]);
    const float* gam = reinterpret_cast<const float*>(d[2]);
    bf16_t* dst_n = reinterpret_cast<bf16_t*>(d[4]);
    bf16_t* dst_t = reinterpret_cast<bf16_t*>(d[5]);
    const int N = (int)d[8], K = (int)d[9];
    const int tk = (K + 31) / 32, tn = (N + 31) / 32;
    if ((int)blockIdx.x >= tk * tn) return;
    const int n0 = (blockIdx.x / tk) * 32, k0 = (blockIdx.x % tk) * 32;
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const float gk = k0 + cx < K ? gam[k0 + cx] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ry + 8 * i;
        float v = 0.f;
        if (n0 + r < N && k0 + cx < K) {
            v = src[(size_t)(n0 + r) * K + k0 + cx] * gk;
            dst_n[(size_t)(n0 + r) * K + k0 + cx] = f2bf(v);
        }
        tile[r][cx] = v;
    }
    __syncthreads();
    if (dst_t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = ry + 8 * i;  // k index within the tile
            if (k0 + r < K && n0 + cx < N) dst_t[(size_t)(k0 + r) * N + n0 + cx] = f2bf(tile[cx][r]);
        }
    }
}
// one wave per output row n: bias_f[n] = bias[n] + sum_k w[n,k] beta[k];  rsum[n] = sum_k float(bf16(w[n,k] gamma[k]))
// (the ROUNDED folded weights: c1 must be the row mean of what the dX GEMM actually accumulates)
__global__ __launch_bounds__(256) void fold_rows_kernel(const int64_t* __restrict__ desc) {
    const int64_t* d = desc + (size_t)blockIdx.y * 10;
    const float* w = reinterpret_cast<const float*>(d[0]);
    const float* bias = reinterpret_cast<const float*>(d[1]);
    const float* gam = reinterpret_cast<const float*>(d[2]);
    const float* bet = reinterpret_cast<const float*>(d[3]);
    float* bias_f = reinterpret_cast<float*>(d[6]);
    float* rsum = reinterpret_cast<float*>(d[7]);
    const int N = (int)d[8], K = (int)d[9];
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float sb = 0.f, sr = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float wv = w[(size_t)n * K + k];
        sb = fmaf(wv, bet[k], sb);
        sr += bf2f(f2bf(wv * gam[k]));
    }
    sb = wave_sum(sb);
    sr = wave_sum(sr);
    if (lane == 0) { bias_f[n] = (bias ? bias[n] : 0.f) + sb; rsum[n] = sr; }
}
extern "C" int mbx_fold_norm_weights(const int64_t* desc, int n_desc, int max_n, int max_k, void* stream) {
    MBX_CHECK_ARG(desc && n_desc > 0 && max_n > 0 && max_k > 0, "fold_norm_weights: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(fold_weights_kernel, dim3(((max_n + 31) / 32) * ((max_k + 31) / 32), n_desc), dim3(256), 0, s, desc);
    MBX_LAUNCH_CHECK("fold_weights");
    hipLaunchKernelGGL(fold_rows_kernel, dim3((max_n + 3) / 4, n_desc), dim3(256), 0, s, desc);
    MBX_LAUNCH_CHECK("fold_rows");
    return 0;
}

// rowc[m] = {rstd, rstd c1, rstd c2, 0} from the producers' partial row dots part[nb][M][2] = {sum dY s, sum dY (Y - b')} over
// nb column blocks (2 x heads for the attention backward, 64-column blocks for the GELU' epilogue); block-major, so that both the
// producers' stores and these loads run along the tokens; fixed summation order.
__global__ __launch_bounds__(256) void lnbwd_rowc_kernel(const float* __restrict__ part, int nb, const float* __restrict__ rstd,
                                                         float4* __restrict__ rowc, int M, float invC) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float2* pr = reinterpret_cast<const float2*>(part) + m;
    float p1 = 0.f, p2 = 0.f, q1 = 0.f, q2 = 0.f;
    int b = 0;
    for (; b + 1 < nb; b += 2) {
        const float2 v0 = pr[(size_t)b * M], v1 = pr[(size_t)(b + 1) * M];
        p1 += v0.x; p2 += v0.y; q1 += v1.x; q2 += v1.y;
    }
    if (b < nb) { const float2 v0 = pr[(size_t)b * M]; p1 += v0.x; p2 += v0.y; }
    const float rs = rstd[m];
    rowc[m] = make_float4(rs, rs * (p1 + q1) * invC, rs * (p2 + q2) * invC, 0.f);
}
extern "C" int mbx_lnbwd_rowc(const float* part, int nb, const float* rstd, float* rowc, int M, int C, void* stream) {
    MBX_CHECK_ARG(part && rstd && rowc && nb > 0 && M > 0 && C > 0, "lnbwd_rowc: bad arguments");
    hipLaunchKernelGGL(lnbwd_rowc_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, nb, rstd,
                       reinterpret_cast<float4*>(rowc), M, 1.0f / (float)C);
    MBX_LAUNCH_CHECK("lnbwd_rowc");
    return 0;
}

// parameter gradients of a folded (LayerNorm -> Linear) pair from the folded weight gradient, in place:
//   dw[n,k] <- gamma[k] dw[n,k] + db[n] beta[k];   partial dgamma[k] = sum_n w[n,k] dw'[n,k];   partial dbeta[k] = sum_n w[n,k] db[n]
// block = 64 rows (n) x 64 columns (k); partial row layout [dgamma K | dbeta K] per 64-row block, folded by colsum.
__global__ __launch_bounds__(256) void unfold_norm_grads_kernel(float* __restrict__ dw, const float* __restrict__ db,
                                                                const float* __restrict__ w, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ part, int N, int K) {
    __shared__ float red[2][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + tx, nb0 = blockIdx.y * 64;
    float ag = 0.f, ab = 0.f;
    if (k < K) {
        const float g = gamma[k], bt = beta[k];
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int n = nb0 + ty + 4 * i;
            if (n < N) {
                const size_t o = (size_t)n * K + k;
                const float dwp = dw[o], wv = w[o], dbn = db[n];
                ag = fmaf(wv, dwp, ag);
                ab = fmaf(wv, dbn, ab);
                dw[o] = fmaf(g, dwp, dbn * bt);
            }
        }
    }
    red[0][ty][tx] = ag;
    red[1][ty][tx] = ab;
    __syncthreads();
    if (ty == 0 && k < K) {
        float* pr = part + (size_t)blockIdx.y * 2 * K;
        pr[k] = (red[0][0][tx] + red[0][1][tx]) + (red[0][2][tx] + red[0][3][tx]);
        pr[K + k] = (red[1][0][tx] + red[1][1][tx]) + (red[1][2][tx] + red[1][3][tx]);
    }
}
extern "C" size_t mbx_unfold_norm_grads_ws(int N, int K) { return (size_t)((N + 63) / 64) * 2 * K * sizeof(float); }
extern "C" int mbx_unfold_norm_grads(float* dw, const float* db, const float* w, const float* gamma, const float* beta,
                                     float* dgamma, float* dbeta, int N, int K, void* ws, void* stream) {
    MBX_CHECK_ARG(dw && db && w && gamma && beta && dgamma && dbeta && ws, "unfold_norm_grads: null pointer");
    MBX_CHECK_ARG(N > 0 && K > 0, "unfold_norm_grads: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (N + 63) / 64;
    hipLaunchKernelGGL(unfold_norm_grads_kernel, dim3((K + 63) / 64, nblk), dim3(256), 0, s, dw, db, w, gamma, beta, (float*)ws, N, K);
    MBX_LAUNCH_CHECK("unfold_norm_grads");
    return launch_colsum2((const float*)ws, nblk, 2 * K, 0, 2 * K, dgamma, dbeta, K, s);
}

// ------------------------------------------------------------------------------------------------
// adaptive fusion forward
// ------------------------------------------------------------------------------------------------
template <int VPL>
__global__ __launch_bounds__(256) void fuse_fwd_kernel(const float* __restrict__ x_st, const float* __restrict__ x_ts,
                                                       const float* __restrict__ w, const float* __restrict__ b,
                                                       float* __restrict__ out, float* __restrict__ alpha, int M, int C) {
    const int lane = threadIdx.x & 63;
    float w0s[VPL][4], w0t[VPL][4], w1s[VPL][4], w1t[VPL][4];
    ROW_LOOP(k) {
        const int c = ROW_C(k);
        if (c < C) {
            load4<float>(w + c, w0s[k]); load4<float>(w + C + c, w0t[k]);
            load4<float>(w + 2 * C + c, w1s[k]); load4<float>(w + 3 * C + c, w1t[k]);
        }
    }
    const float b0 = b[0], b1 = b[1];
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
        float a[VPL][4], t[VPL][4];
        float l0 = 0.f, l1 = 0.f;
        ROW_LOOP(k) {
            const int c = ROW_C(k);
            if (c < C) {
                load4<float>(x_st + (size_t)row * C + c, a[k]);
                load4<float>(x_ts + (size_t)row * C + c, t[k]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    l0 = fmaf(a[k][i], w0s[k][i], fmaf(t[k][i], w0t[k][i], l0));
                    l1 = fmaf(a[k][i], w1s[k][i], fmaf(t[k][i], w1t[k][i], l1));
                }
            }
        }
        l0 = wave_sum(l0) + b0;
        l1 = wave_sum(l1) + b1;
        const float mx = fmaxf(l0, l1);
        const float e0 = __expf(l0 - mx), e1 = __expf(l1 - mx);
        const float inv = 1.0f / (e0 + e1);
        const float a0 = e0 * inv, a1 = e1 * inv;
        ROW_LOOP(k) {
            const int c = ROW_C(k);
            if (c < C) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = fmaf(a[k][i], a0, t[k][i] * a1);
                store4<float>(out + (size_t)row * C + c, o);
            }
        }
        if (lane == 0) { alpha[(size_t)row * 2] = a0; alpha[(size_t)row * 2 + 1] = a1; }
    }
}
extern "C" int mbx_fuse_fwd(const float* x_st, const float* x_ts, const float* w, const float* b, float* out,
                            float* alpha, int M, int C, void* stream) {
    MBX_CHECK_ARG(x_st && x_ts && w && b && out && alpha, "fuse_fwd: null pointer");
    MBX_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "fuse_fwd: bad shape");
    const int grid = clamp_grid((M + 3) / 4, 256 * 8);
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_VPL(vpl_for(C), hipLaunchKernelGGL((fuse_fwd_kernel<VPL>), dim3(grid), dim3(256), 0, s, x_st, x_ts, w, b, out, alpha, M, C));
    MBX_LAUNCH_CHECK("fuse_fwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// adaptive fusion forward + the LayerNorms that read its output (DSTformer.py:343-349 followed by Block.norm1_s / norm1_t of
// the next level, :241 / :247, or by the final `norm`, :350): the fused row is still in registers, so its statistics and up
// to two affine-normalised T-typed copies leave with it -- the two consumers share ONE mean / rstd (same input row), and
// neither re-reads the 541 MB residual tensor.  Arithmetic identical to fuse_fwd_kernel followed by ln_fwd_row.
// ------------------------------------------------------------------------------------------------
template <typename T, int VPL>
__global__ __launch_bounds__(256) void fuse_ln_fwd_kernel(const float* __restrict__ x_st, const float* __restrict__ x_ts,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          float* __restrict__ out, float* __restrict__ alpha,
                                                          const float* __restrict__ g1, const float* __restrict__ b1, T* __restrict__ xn1,
                                                          const float* __restrict__ g2, const float* __restrict__ b2, T* __restrict__ xn2,
                                                          float eps, float* __restrict__ mean, float* __restrict__ rstd, int M, int C) {
    const int lane = threadIdx.x & 63;
    float w0s[VPL][4], w0t[VPL][4], w1s[VPL][4], w1t[VPL][4];
    ROW_LOOP(k) {
        const int c = ROW_C(k);
        if (c < C) {
            load4<float>(w + c, w0s[k]); load4<float>(w + C + c, w0t[k]);
            load4<float>(w + 2 * C + c, w1s[k]); load4<float>(w + 3 * C + c, w1t[k]);
        }
    }
    const float b0 = b[0], bb1 = b[1];
    const float invC = 1.0f / (float)C;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
        float a[VPL][4], t[VPL][4];
        float l0 = 0.f, l1 = 0.f;
        ROW_LOOP(k) {
            const int c = ROW_C(k);
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[k][i] = 0.f; t[k][i] = 0.f; }
            if (c < C) {
                load4<float>(x_st + (size_t)row * C + c, a[k]);
                load4<float>(x_ts + (size_t)row * C + c, t[k]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    l0 = fmaf(a[k][i], w0s[k][i], fmaf(t[k][i], w0t[k][i], l0));
                    l1 = fmaf(a[k][i], w1s[k][i], fmaf(t[k][i], w1t[k][i], l1));
                }
            }
        }
        l0 = wave_sum(l0) + b0;
        l1 = wave_sum(l1) + bb1;
        const float mx = fmaxf(l0, l1);
        const float e0 = __expf(l0 - mx), e1 = __expf(l1 - mx);
        const float inv = 1.0f / (e0 + e1);
        const float a0 = e0 * inv, a1 = e1 * inv;
        float s = 0.f;
        ROW_LOOP(k) {
            const int c = ROW_C(k);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[k][i] = fmaf(a[k][i], a0, t[k][i] * a1);     // the fused row (zeros past C)
            if (c < C) store4<float>(out + (size_t)row * C + c, a[k]);
            s += (a[k][0] + a[k][1]) + (a[k][2] + a[k][3]);
        }
        const float mu = wave_sum(s) * invC;
        float q = 0.f;
        ROW_LOOP(k) {
            const int c = ROW_C(k);
            if (c < C) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { a[k][i] -= mu; q = fmaf(a[k][i], a[k][i], q); }
            }
        }
        const float rs = 1.0f / sqrtf(wave_sum(q) * invC + eps);
        ROW_LOOP(k) {
            const int c = ROW_C(k);
            if (c < C) {
                float gg[4] = {1.f, 1.f, 1.f, 1.f}, bt[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
                if (g1) { load4<float>(g1 + c, gg); load4<float>(b1 + c, bt); }
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = fmaf(a[k][i] * rs, gg[i], bt[i]);
                store4<T>(xn1 + (size_t)row * C + c, o);
                if (xn2) {
                    load4<float>(g2 + c, gg); load4<float>(b2 + c, bt);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = fmaf(a[k][i] * rs, gg[i], bt[i]);
                    store4<T>(xn2 + (size_t)row * C + c, o);
                }
            }
        }
        if (lane == 0) { alpha[(size_t)row * 2] = a0; alpha[(size_t)row * 2 + 1] = a1; mean[row] = mu; rstd[row] = rs; }
    }
}
extern "C" int mbx_fuse_ln_fwd(const float* x_st, const float* x_ts, const float* w, const float* b, float* out, float* alpha,
                               const float* g1, const float* b1, void* xn1, const float* g2, const float* b2, void* xn2, float eps,
                               float* mean, float* rstd, int M, int C, int dtype, void* stream) {
    MBX_CHECK_ARG(x_st && x_ts && w && b && out && alpha && xn1 && mean && rstd, "fuse_ln_fwd: null pointer");
    MBX_CHECK_ARG((g1 && b1) || (!g1 && !b1 && !xn2), "fuse_ln_fwd: g1 / b1 come together (both NULL = one plain normalisation, no second output)");
    MBX_CHECK_ARG((g2 && b2 && xn2) || (!g2 && !b2 && !xn2), "fuse_ln_fwd: the second LayerNorm needs gamma, beta and an output (or none of them)");
    MBX_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "fuse_ln_fwd: bad shape");
    MBX_CHECK_ARG(dtype == MBX_BF16 || dtype == MBX_F32, "fuse_ln_fwd: unknown dtype %d", dtype);
    const int grid = clamp_grid((M + 3) / 4, 256 * 8);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MBX_BF16) {
        DISPATCH_VPL(vpl_for(C), hipLaunchKernelGGL((fuse_ln_fwd_kernel<bf16_t, VPL>), dim3(grid), dim3(256), 0, s, x_st, x_ts, w, b, out, alpha,
                                                    g1, b1, (bf16_t*)xn1, g2, b2, (bf16_t*)xn2, eps, mean, rstd, M, C));
    } else {
        DISPATCH_VPL(vpl_for(C), hipLaunchKernelGGL((fuse_ln_fwd_kernel<float, VPL>), dim3(grid), dim3(256), 0, s, x_st, x_ts, w, b, out, alpha,
                                                    g1, b1, (float*)xn1, g2, b2, (float*)xn2, eps, mean, rstd, M, C));
    }
    MBX_LAUNCH_CHECK("fuse_ln_fwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// adaptive fusion backward.  partial row layout: [dw 4C | db 2 | pad 2]
// ------------------------------------------------------------------------------------------------
#define FUSE_BWD_BLOCKS 2048
template <typename T, int VPL>
__global__ __launch_bounds__(256) void fuse_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ x_st,
                                                       const float* __restrict__ x_ts, const float* __restrict__ alpha,
                                                       const float* __restrict__ w, float* __restrict__ d_st,
                                                       float* __restrict__ d_ts, T* __restrict__ d_st_t,
                                                       T* __restrict__ d_ts_t, float* __restrict__ part, int M, int C,
                                                       const bf16_t* __restrict__ dh_a = nullptr, const bf16_t* __restrict__ dh_b = nullptr) {
    // dh_a / dh_b (round 5): the incoming gradient as the SUM of two bf16 tensors -- the input gradients of the two Blocks of the level
    // above, which leave their row-owner LayerNorm-backward kernels in the operand type -- instead of one fp32 tensor (same bytes)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = 4 * C + 4;
    float w0s[VPL][4], w0t[VPL][4], w1s[VPL][4], w1t[VPL][4];
    float g0s[VPL][4], g0t[VPL][4], g1s[VPL][4], g1t[VPL][4];
    float gb0 = 0.f, gb1 = 0.f;
    ROW_LOOP(k) {
        const int c = ROW_C(k);
#pragma unroll
        for (int i = 0; i < 4; ++i) { g0s[k][i] = g0t[k][i] = g1s[k][i] = g1t[k][i] = 0.f; w0s[k][i] = w0t[k][i] = w1s[k][i] = w1t[k][i] = 0.f; }
        if (c < C) {
            load4<float>(w + c, w0s[k]); load4<float>(w + C + c, w0t[k]);
            load4<float>(w + 2 * C + c, w1s[k]); load4<float>(w + 3 * C + c, w1t[k]);
        }
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float a0 = alpha[(size_t)row * 2], a1 = alpha[(size_t)row * 2 + 1];
        float d[VPL][4], a[VPL][4], t[VPL][4];
        float da0 = 0.f, da1 = 0.f;
        ROW_LOOP(k) {
            const int c = ROW_C(k);
            if (c < C) {
                if (dh_a != nullptr) {
                    float e[4];
                    load4<bf16_t>(dh_a + (size_t)row * C + c, d[k]);
                    load4<bf16_t>(dh_b + (size_t)row * C + c, e);
#pragma unroll
                    for (int i = 0; i < 4; ++i) d[k][i] += e[i];
                } else {
                    load4<float>(dh + (size_t)row * C + c, d[k]);
                }
                load4<float>(x_st + (size_t)row * C + c, a[k]);
                load4<float>(x_ts + (size_t)row * C + c, t[k]);
#pragma unroll
                for (int i = 0; i < 4; ++i) { da0 = fmaf(d[k][i], a[k][i], da0); da1 = fmaf(d[k][i], t[k][i], da1); }
            }
        }
        da0 = wave_sum(da0);
        da1 = wave_sum(da1);
        const float dot = da0 * a0 + da1 * a1;
        const float dl0 = a0 * (da0 - dot), dl1 = a1 * (da1 - dot);
        gb0 += dl0;
        gb1 += dl1;
        ROW_LOOP(k) {
            const int c = ROW_C(k);
            if (c < C) {
                float rs[4], rt[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rs[i] = fmaf(d[k][i], a0, fmaf(dl0, w0s[k][i], dl1 * w1s[k][i]));
                    rt[i] = fmaf(d[k][i], a1, fmaf(dl0, w0t[k][i], dl1 * w1t[k][i]));
                    g0s[k][i] = fmaf(dl0, a[k][i], g0s[k][i]);
                    g0t[k][i] = fmaf(dl0, t[k][i], g0t[k][i]);
                    g1s[k][i] = fmaf(dl1, a[k][i], g1s[k][i]);
                    g1t[k][i] = fmaf(dl1, t[k][i], g1t[k][i]);
                }
                if (d_st != nullptr) {      // (NULL with the gradient stream in the operand type: the Blocks read d_st_t / d_ts_t only)
                    store4<float>(d_st + (size_t)row * C + c, rs);
                    store4<float>(d_ts + (size_t)row * C + c, rt);
                }
                store4<T>(d_st_t + (size_t)row * C + c, rs);
                store4<T>(d_ts_t + (size_t)row * C + c, rt);
            }
        }
    }
    float* mine = lds + (size_t)wave * n;
    ROW_LOOP(k) {
        const int c = ROW_C(k);
        if (c < C) {
            store4<float>(mine + c, g0s[k]); store4<float>(mine + C + c, g0t[k]);
            store4<float>(mine + 2 * C + c, g1s[k]); store4<float>(mine + 3 * C + c, g1t[k]);
        }
    }
    if (lane == 0) { mine[4 * C] = gb0; mine[4 * C + 1] = gb1; mine[4 * C + 2] = 0.f; mine[4 * C + 3] = 0.f; }
    __syncthreads();
    block_fold_store(lds, n, part + (size_t)blockIdx.x * n);
}
extern "C" size_t mbx_fuse_bwd_ws(int C) { return (size_t)FUSE_BWD_BLOCKS * (4 * (size_t)C + 4) * sizeof(float); }
extern "C" int mbx_fuse_bwd(const float* dh, const float* x_st, const float* x_ts, const float* alpha, const float* w,
                            float* d_st, float* d_ts, void* d_st_t, void* d_ts_t, float* dw, float* db, int M, int C,
                            int dtype, void* ws, void* stream) {
    MBX_CHECK_ARG(dh && x_st && x_ts && alpha && w && d_st_t && d_ts_t && dw && db && ws, "fuse_bwd: null pointer");
    MBX_CHECK_ARG((d_st && d_ts) || (!d_st && !d_ts), "fuse_bwd: d_st and d_ts come together (both NULL: only the T-typed copies are written)");
    MBX_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "fuse_bwd: bad shape");
    const int grid = clamp_grid((M + 3) / 4, FUSE_BWD_BLOCKS);
    const int n = 4 * C + 4;
    const size_t shm = (size_t)4 * n * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)ws;
    if (dtype == MBX_BF16) {
        DISPATCH_VPL(vpl_for(C), hipLaunchKernelGGL((fuse_bwd_kernel<bf16_t, VPL>), dim3(grid), dim3(256), shm, s, dh, x_st, x_ts, alpha, w, d_st, d_ts, (bf16_t*)d_st_t, (bf16_t*)d_ts_t, part, M, C));
    } else if (dtype == MBX_F32) {
        DISPATCH_VPL(vpl_for(C), hipLaunchKernelGGL((fuse_bwd_kernel<float, VPL>), dim3(grid), dim3(256), shm, s, dh, x_st, x_ts, alpha, w, d_st, d_ts, (float*)d_st_t, (float*)d_ts_t, part, M, C));
    } else {
        return mbx_set_error("fuse_bwd: unknown dtype %d", dtype);
    }
    MBX_LAUNCH_CHECK("fuse_bwd");
    if (mbx_launch_colsum(part, grid, n, 0, 4 * C, dw, s)) return 1;
    if (mbx_launch_colsum(part, grid, n, 4 * C, 2, db, s)) return 1;
    return 0;
}

// the same with dh = dh_a + dh_b, both bf16 [M,C] (see the kernel); bf16 outputs only
extern "C" int mbx_fuse_bwd_pair(const void* dh_a, const void* dh_b, const float* x_st, const float* x_ts, const float* alpha, const float* w,
                                 void* d_st_t, void* d_ts_t, float* dw, float* db, int M, int C, void* ws, void* stream) {
    MBX_CHECK_ARG(dh_a && dh_b && x_st && x_ts && alpha && w && d_st_t && d_ts_t && dw && db && ws, "fuse_bwd_pair: null pointer");
    MBX_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "fuse_bwd_pair: bad shape");
    const int grid = clamp_grid((M + 3) / 4, FUSE_BWD_BLOCKS);
    const int n = 4 * C + 4;
    const size_t shm = (size_t)4 * n * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)ws;
    DISPATCH_VPL(vpl_for(C), hipLaunchKernelGGL((fuse_bwd_kernel<bf16_t, VPL>), dim3(grid), dim3(256), shm, s, (const float*)nullptr, x_st, x_ts, alpha, w,
                                                (float*)nullptr, (float*)nullptr, (bf16_t*)d_st_t, (bf16_t*)d_ts_t, part, M, C, (const bf16_t*)dh_a, (const bf16_t*)dh_b));
    MBX_LAUNCH_CHECK("fuse_bwd_pair");
    if (mbx_launch_colsum(part, grid, n, 0, 4 * C, dw, s)) return 1;
    if (mbx_launch_colsum(part, grid, n, 4 * C, 2, db, s)) return 1;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// att_fuse=False: plain average and its backward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void average_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4((u.x + v.x) * 0.5f, (u.y + v.y) * 0.5f, (u.z + v.z) * 0.5f, (u.w + v.w) * 0.5f);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void average_bwd_kernel(const float* __restrict__ dh, float* __restrict__ d_st,
                                                          float* __restrict__ d_ts, T* __restrict__ d_st_t,
                                                          T* __restrict__ d_ts_t, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float v[4];
        load4<float>(dh + i * 4, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= 0.5f;
        store4<float>(d_st + i * 4, v);
        store4<float>(d_ts + i * 4, v);
        store4<T>(d_st_t + i * 4, v);
        store4<T>(d_ts_t + i * 4, v);
    }
}
extern "C" int mbx_average(const float* x_st, const float* x_ts, float* out, size_t n, void* stream) {
    MBX_CHECK_ARG(x_st && x_ts && out && n % 4 == 0, "average: bad arguments");
    hipLaunchKernelGGL(average_kernel, dim3(clamp_grid((n / 4 + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, x_st, x_ts, out, n / 4);
    MBX_LAUNCH_CHECK("average");
    return 0;
}
extern "C" int mbx_average_bwd(const float* dh, float* d_st, float* d_ts, void* d_st_t, void* d_ts_t, size_t n,
                               int dtype, void* stream) {
    MBX_CHECK_ARG(dh && d_st && d_ts && d_st_t && d_ts_t && n % 4 == 0, "average_bwd: bad arguments");
    const int grid = clamp_grid((n / 4 + 255) / 256, 4096);
    if (dtype == MBX_BF16)
        hipLaunchKernelGGL(average_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dh, d_st, d_ts, (bf16_t*)d_st_t, (bf16_t*)d_ts_t, n / 4);
    else if (dtype == MBX_F32)
        hipLaunchKernelGGL(average_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dh, d_st, d_ts, (float*)d_st_t, (float*)d_ts_t, n / 4);
    else
        return mbx_set_error("average_bwd: unknown dtype %d", dtype);
    MBX_LAUNCH_CHECK("average_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// head: Linear(R -> Dout), Dout <= 8 (skinny: VALU dot products, wave reduction)
// ------------------------------------------------------------------------------------------------
#define HEAD_MAXD 8
template <int VPL>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ rep, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ out, int M, int R,
                                                       int Dout) {
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
        float v[VPL][4];
        ROW_LOOP(k) {
            const int c = ROW_C(k);
            if (c < R) load4<float>(rep + (size_t)row * R + c, v[k]);
        }
        float mine = 0.f;
        for (int i = 0; i < Dout; ++i) {
            float a = 0.f;
            ROW_LOOP(k) {
                const int c = ROW_C(k);
                if (c < R) {
                    float ww[4];
                    load4<float>(w + (size_t)i * R + c, ww);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a = fmaf(v[k][e], ww[e], a);
                }
            }
            a = wave_sum(a) + b[i];
            if (lane == i) mine = a;
        }
        if (lane < Dout) out[(size_t)row * Dout + lane] = mine;
    }
}
extern "C" int mbx_head_fwd(const float* rep, const float* w, const float* b, float* out, int M, int R, int Dout,
                            void* stream) {
    MBX_CHECK_ARG(rep && w && b && out, "head_fwd: null pointer");
    MBX_CHECK_ARG(M > 0 && R % 4 == 0 && Dout >= 1 && Dout <= HEAD_MAXD, "head_fwd: need R %% 4 == 0 and 1 <= dim_out <= %d", HEAD_MAXD);
    const int grid = clamp_grid((M + 3) / 4, 256 * 8);
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_VPL(vpl_for(R), hipLaunchKernelGGL((head_fwd_kernel<VPL>), dim3(grid), dim3(256), 0, s, rep, w, b, out, M, R, Dout));
    MBX_LAUNCH_CHECK("head_fwd");
    return 0;
}

// head backward: dpre = (dout . w) * (1 - rep^2); partial row layout: [dw Dout*R | db 8]
#define HEAD_BWD_BLOCKS 1024
template <typename T, int VPL>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ rep,
                                                       const float* __restrict__ w, T* __restrict__ dpre,
                                                       float* __restrict__ part, int M, int R, int Dout) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = Dout * R + 8;
    // zero this wave's LDS accumulator rows (dw accumulates in LDS-free registers per output i below)
    float* mine = lds + (size_t)wave * n;
    // Round 3: the rows of `rep` are read ONCE per group of four output channels (one pass for the model's dim_out = 3; the round-2
    // kernel walked them once per channel and once more for dpre: 2.1 GB instead of 0.8 GB per launch).  The first pass also writes
    // dpre; the weight-gradient rows of a group live in 4 x VPL x 4 registers.
    constexpr int DG = 4;
    for (int i0 = 0; i0 < Dout; i0 += DG) {
        float gw[DG][VPL][4], gbv[DG];
#pragma unroll
        for (int j = 0; j < DG; ++j) {
            gbv[j] = 0.f;
            ROW_LOOP(k) {
#pragma unroll
                for (int e = 0; e < 4; ++e) gw[j][k][e] = 0.f;
            }
        }
        for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
            float dvg[DG];
#pragma unroll
            for (int j = 0; j < DG; ++j) dvg[j] = i0 + j < Dout ? dout[(size_t)row * Dout + i0 + j] : 0.f;
#pragma unroll
            for (int j = 0; j < DG; ++j) gbv[j] += dvg[j];
            float dv[HEAD_MAXD];
            if (i0 == 0) {
#pragma unroll
                for (int q = 0; q < HEAD_MAXD; ++q) dv[q] = q < DG ? dvg[q < DG ? q : 0] : (q < Dout ? dout[(size_t)row * Dout + q] : 0.f);
            }
            ROW_LOOP(k) {
                const int c = ROW_C(k);
                if (c < R) {
                    float r[4];
                    load4<float>(rep + (size_t)row * R + c, r);
#pragma unroll
                    for (int j = 0; j < DG; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) gw[j][k][e] = fmaf(dvg[j], r[e], gw[j][k][e]);
                    if (i0 == 0) {      // dpre = (dout . w) (1 - rep^2): all Dout weights, once
                        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int q = 0; q < HEAD_MAXD; ++q) {
                            if (q < Dout) {
                                float ww[4];
                                load4<float>(w + (size_t)q * R + c, ww);
#pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] = fmaf(dv[q], ww[e], o[e]);
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] *= (1.0f - r[e] * r[e]);
                        store4<T>(dpre + (size_t)row * R + c, o);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < DG; ++j) {
            if (i0 + j < Dout) {
                ROW_LOOP(k) {
                    const int c = ROW_C(k);
                    if (c < R) store4<float>(mine + (size_t)(i0 + j) * R + c, gw[j][k]);
                }
                if (lane == 0) mine[Dout * R + i0 + j] = gbv[j];
            }
        }
    }
    if (lane == 0)
        for (int q = Dout; q < 8; ++q) mine[Dout * R + q] = 0.f;      // the unused tail of the [db 8] block
    __syncthreads();
    block_fold_store(lds, n, part + (size_t)blockIdx.x * n);
}
extern "C" size_t mbx_head_bwd_ws(int R, int Dout) { return (size_t)HEAD_BWD_BLOCKS * ((size_t)Dout * R + 8) * sizeof(float); }
extern "C" int mbx_head_bwd(const float* dout, const float* rep, const float* w, void* dpre_t, float* dw, float* db,
                            int M, int R, int Dout, int dtype, void* ws, void* stream) {
    MBX_CHECK_ARG(dout && rep && w && dpre_t && dw && db && ws, "head_bwd: null pointer");
    MBX_CHECK_ARG(M > 0 && R % 4 == 0 && Dout >= 1 && Dout <= HEAD_MAXD, "head_bwd: need R %% 4 == 0 and 1 <= dim_out <= %d", HEAD_MAXD);
    const int grid = clamp_grid((M + 3) / 4, HEAD_BWD_BLOCKS);
    const int n = Dout * R + 8;
    const size_t shm = (size_t)4 * n * sizeof(float);
    MBX_CHECK_ARG(shm <= 160 * 1024, "head_bwd: dim_out*dim_rep too large for the LDS fold");
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)ws;
    if (dtype == MBX_BF16) {
        DISPATCH_VPL(vpl_for(R), hipLaunchKernelGGL((head_bwd_kernel<bf16_t, VPL>), dim3(grid), dim3(256), shm, s, dout, rep, w, (bf16_t*)dpre_t, part, M, R, Dout));
    } else if (dtype == MBX_F32) {
        DISPATCH_VPL(vpl_for(R), hipLaunchKernelGGL((head_bwd_kernel<float, VPL>), dim3(grid), dim3(256), shm, s, dout, rep, w, (float*)dpre_t, part, M, R, Dout));
    } else {
        return mbx_set_error("head_bwd: unknown dtype %d", dtype);
    }
    MBX_LAUNCH_CHECK("head_bwd");
    if (mbx_launch_colsum(part, grid, n, 0, Dout * R, dw, s)) return 1;
    if (mbx_launch_colsum(part, grid, n, Dout * R, Dout, db, s)) return 1;
    return 0;
}

// tanh backward on the representation path: dpre = drep * (1 - rep^2)
template <typename T>
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ drep, const float* __restrict__ rep,
                                                       T* __restrict__ dpre, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float d[4], r[4];
        load4<float>(drep + i * 4, d);
        load4<float>(rep + i * 4, r);
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] *= (1.0f - r[k] * r[k]);
        store4<T>(dpre + i * 4, d);
    }
}
extern "C" int mbx_tanh_bwd(const float* drep, const float* rep, void* dpre_t, size_t n, int dtype, void* stream) {
    MBX_CHECK_ARG(drep && rep && dpre_t && n % 4 == 0, "tanh_bwd: bad arguments");
    const int grid = clamp_grid((n / 4 + 255) / 256, 4096);
    if (dtype == MBX_BF16)
        hipLaunchKernelGGL(tanh_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, drep, rep, (bf16_t*)dpre_t, n / 4);
    else if (dtype == MBX_F32)
        hipLaunchKernelGGL(tanh_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, drep, rep, (float*)dpre_t, n / 4);
    else
        return mbx_set_error("tanh_bwd: unknown dtype %d", dtype);
    MBX_LAUNCH_CHECK("tanh_bwd");
    return 0;
}
