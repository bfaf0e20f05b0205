// Pipelined bf16 MFMA GEMMs for gfx950 (the throughput path; gemm.hip keeps the fp32 parity kernels).
//
// Structure (both kernels): 512-thread workgroup = 8 waves (2 per SIMD), one workgroup per CU,
// a 3-stage LDS ring filled by LDS-DMA (global_load_lds_dwordx4: HBM/L2 -> LDS without touching
// VGPRs), two k-tiles in flight behind the one being multiplied, ONE raw s_barrier per k-tile and
// counted s_waitcnt vmcnt(N) (never 0 inside the loop) so that the DMA of tiles t+1, t+2 stays in
// flight across the barrier while the MFMAs of tile t run.
//
//   gemm_nt_pipe : acc[M,N] = A[M,K] . W[N,K]^T, 256 x 128 output tile, wave grid 4(M) x 2(N), each wave
//                  64 x 64 = 2 x 2 MFMA 32x32x16 tiles; stage = A [256][64] + W [128][64] bf16 = 48 KiB.
//                  LDS image: rows of 128 B, 16-byte chunk c of row r at physical chunk c ^ ((r >> 1) & 7)
//                  (conflict-free ds_read_b128 fragment reads).  LDS-DMA writes lane-linear, so the
//                  swizzle is applied to the per-lane SOURCE address (same involution on both sides).
//   gemm_tn_pipe : dW[N,K] = dY[M,N]^T . A[M,K] (contraction over the TOKEN dimension, which is the
//                  slow dimension of both operands).  The tiles are staged untransposed,
//                  [64 tokens][256 n] and [64 tokens][128 k], and the MFMA fragments are fetched with the
//                  gfx950 transpose read ds_read_b64_tr_b16: lane r of a 16-lane group supplies the
//                  address of 4 consecutive elements of token row (r >> 2), column block 4 (r & 3); the
//                  hardware returns to lane i the 4 consecutive TOKENS of column i (mapping measured on
//                  hardware with tools/probes/tr_probe.hip).  256 x 128 output tile, split over the
//                  token dimension, fp32 partial tiles folded by colsum (deterministic, no atomics).
#include "mbx_common.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

#define GLDS16(src, dst) __builtin_amdgcn_global_load_lds((gbl_void_t*)(src), (lds_void_t*)(dst), 16, 0, 0)
#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// erf-GELU for the bf16 path: Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7, far below bf16 resolution);
// erf(u / sqrt 2) and the Gaussian of GELU' share one exponential, exp(-u^2 / 2).
__device__ __forceinline__ void erf_parts(float u, float& erf_v, float& gauss) {
    const float x = fabsf(u) * 0.70710678118654752440f;
    const float t = __frcp_rn(fmaf(0.3275911f, x, 1.0f));
    gauss = __expf(-0.5f * u * u);
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    erf_v = copysignf(fmaf(-poly, gauss, 1.0f), u);
}
__device__ __forceinline__ float gelu_fast(float u) {
    float e, g;
    erf_parts(u, e, g);
    return 0.5f * u * (1.0f + e);
}
__device__ __forceinline__ float gelu_fast_grad(float u) {
    float e, g;
    erf_parts(u, e, g);
    return fmaf(u * g, 0.39894228040143267794f, 0.5f * (1.0f + e));
}

__device__ __forceinline__ int xcd_remap2(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ================================================================================================
// gemm_nt_pipe: 256 x 128 tile, BK = 32, 3-stage ring of 24 KiB -> 72 KiB per workgroup, TWO workgroups
// (16 waves, 4 per SIMD) per CU: while one workgroup drains its tile through the store-bound epilogue
// the other one keeps the MFMA pipe and the LDS-DMA stream busy.
// ================================================================================================
static constexpr int P_BM = 256, P_BN = 128, P_BK = 32, P_ROWB = 64;
static constexpr int P_A_BYTES = P_BM * P_ROWB, P_W_BYTES = P_BN * P_ROWB, P_STAGE = P_A_BYTES + P_W_BYTES;  // 24 KiB
static constexpr int P_NSTAGE = 3;

// rows of 64 B = four 16-byte chunks; chunk c of row r at physical chunk c ^ ((r >> 2) & 3)
__device__ __forceinline__ int sw_off(int row, int chunk) { return row * P_ROWB + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int EPI>
__global__ __launch_bounds__(512, 4) void gemm_nt_pipe_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                              const float* __restrict__ bias, bf16_t* __restrict__ out_t,
                                                              bf16_t* __restrict__ out2_t, float* __restrict__ out_f,
                                                              const float* __restrict__ resid, const bf16_t* __restrict__ aux,
                                                              int M, int N, int K, int ntn, int dbg, long long* trace) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 3 stages x 24 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = xcd_remap2(blockIdx.x, gridDim.x);
    const int n0 = (lid % ntn) * P_BN, m0 = (lid / ntn) * P_BM;
    const int wm = wave >> 1, wn = wave & 1;

    // LDS-DMA assignment: one instruction = 16 rows x 64 B.  wave w fills A rows [32 w, 32 w + 32) (2 instr)
    // and W rows [16 w, 16 w + 16) (1 instr).  lane -> (row = R0 + (lane >> 2), physical chunk p = lane & 3),
    // source chunk = p ^ ((row >> 2) & 3).
    const int lr = lane >> 2, lp = lane & 3;
    const bf16_t* srcA[2];
    const bf16_t* srcW;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 32 + i * 16 + lr;
        srcA[i] = A + (size_t)min(m0 + row, M - 1) * K + ((lp ^ ((row >> 2) & 3)) << 3);
    }
    {
        const int row = wave * 16 + lr;
        srcW = W + (size_t)min(n0 + row, N - 1) * K + ((lp ^ ((row >> 2) & 3)) << 3);
    }
    char* dstA = smem + wave * 32 * P_ROWB;               // + stage * P_STAGE + i * 1024
    char* dstW = smem + P_A_BYTES + wave * 16 * P_ROWB;   // + stage * P_STAGE

    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = K / P_BK;
#define NT_ISSUE(kt_, stage_)                                                          \
    do {                                                                               \
        const size_t ko_ = (size_t)(kt_) * P_BK;                                       \
        GLDS16(srcA[0] + ko_, dstA + (stage_) * P_STAGE);                              \
        GLDS16(srcA[1] + ko_, dstA + (stage_) * P_STAGE + 1024);                       \
        GLDS16(srcW + ko_, dstW + (stage_) * P_STAGE);                                 \
    } while (0)

    // dbg bits (diagnostics only, MBX_DBG env): 1 = skip MFMA block, 2 = skip LDS-DMA, 4 = skip epilogue
    if (!(dbg & 2)) {
        NT_ISSUE(0, 0);
        if (nk > 1) NT_ISSUE(1, 1);
    }
    const int i = lane & 31, g = lane >> 5;
    int stage = 0;
    // diagnostics: cycle stamps of one wave of one workgroup (MBX_TRACE_BUF), 4 per k-tile + 2 for the epilogue
    const bool tr_on = trace != nullptr && blockIdx.x == 4000 && tid == 0;
#ifdef MBX_TRACE
#define TSTAMP(slot_) do { if (tr_on) trace[slot_] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TSTAMP(slot_) do { (void)tr_on; } while (0)
#endif
    TSTAMP(0);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) WAIT_VMCNT(3); else WAIT_VMCNT(0);   // tile kt landed (this wave's share); kt+1 may fly
        TSTAMP(1 + kt * 4);
        __builtin_amdgcn_s_barrier();                          // everyone's share landed; stage (kt+2)%3 is free
        TSTAMP(2 + kt * 4);
        if (kt + 2 < nk && !(dbg & 2)) {
            const int st2 = stage >= 1 ? stage - 1 : 2;        // (kt + 2) % 3
            NT_ISSUE(kt + 2, st2);
        }
        TSTAMP(3 + kt * 4);
        const char* sA = smem + stage * P_STAGE;
        const char* sW = sA + P_A_BYTES;
        if (!(dbg & 1))
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8_t fw[2], fa[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fw[t] = *reinterpret_cast<const bf16x8_t*>(sW + sw_off(wn * 64 + t * 32 + i, 2 * s + g));
                fa[t] = *reinterpret_cast<const bf16x8_t*>(sA + sw_off(wm * 64 + t * 32 + i, 2 * s + g));
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[tn], fa[tm], acc[tn][tm], 0, 0, 0);
        }
        stage = stage == 2 ? 0 : stage + 1;
        TSTAMP(4 + kt * 4);
    }
#undef NT_ISSUE

    // ---- fused epilogue, coalesced ------------------------------------------------------------------
    // The accumulator layout (lane = row, 4-column quads) would make every global store touch 32 rows;
    // each wave transposes its 64 x 64 fp32 tile through LDS (the ring is idle now) in two 64 x 32 halves
    // and walks each half row-major: one instruction = 8 rows x 32 columns, 16-byte vectors per lane.
    constexpr int EROW = 32 * 4 + 16;                       // padded LDS row of the per-wave staging half-tile
    if (dbg & 4) return;
    __builtin_amdgcn_s_barrier();                           // all waves are done reading the last stage
    char* er = smem + wave * (64 * EROW);                   // 9 KiB per wave, 72 KiB per workgroup
    const int ec = (lane & 7) * 4, erow0 = lane >> 3;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(er + (tm * 32 + i) * EROW + (8 * q + 4 * g) * 4) =
                    make_float4(acc[tn][tm][4 * q], acc[tn][tm][4 * q + 1], acc[tn][tm][4 * q + 2], acc[tn][tm][4 * q + 3]);
        const int n = n0 + wn * 64 + tn * 32 + ec;
        float bb[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias && n < N) load4<float>(bias + n, bb);
#pragma unroll 4
        for (int p = 0; p < 8; ++p) {
            const int rl = p * 8 + erow0, m = m0 + wm * 64 + rl;
            const float4 t4 = *reinterpret_cast<const float4*>(er + rl * EROW + ec * 4);
            if (m < M && n < N) {
                float v[4] = {t4.x + bb[0], t4.y + bb[1], t4.z + bb[2], t4.w + bb[3]};
                const size_t o = (size_t)m * N + n;
                if (EPI == MBX_EPI_STORE) {
                    store4<bf16_t>(out_t + o, v);
                } else if (EPI == MBX_EPI_GELU) {
                    if (out_t) store4<bf16_t>(out_t + o, v);   // pre-activation is only needed for backward
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
                    store4<bf16_t>(out2_t + o, v);
                } else if (EPI == MBX_EPI_RESID) {
                    float r[4];
                    load4<float>(resid + o, r);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += r[e];
                    store4<float>(out_f + o, v);
                } else if (EPI == MBX_EPI_TANH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
                    store4<float>(out_f + o, v);
                } else if (EPI == MBX_EPI_DGELU) {
                    float u[4];
                    load4<bf16_t>(aux + o, u);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= gelu_fast_grad(u[e]);
                    store4<bf16_t>(out_t + o, v);
                }
            }
        }
    }
    TSTAMP(1 + nk * 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TSTAMP(2 + nk * 4);
#undef TSTAMP
}

// ================================================================================================
// gemm_nt_pipe256: 256 x 256 tile, 8 waves as 2 (M) x 4 (N), each wave 128 x 64 = 4 x 2 MFMA tiles
// (6 LDS fragment reads per 8 MFMAs instead of 4 per 4), BK = 32, 4-stage ring of 32 KiB (three k-tiles
// in flight), one workgroup per CU.  Versus the 256 x 128 kernel it moves 1/3 fewer bytes L2 -> LDS and
// 1/4 fewer bytes LDS -> registers per FLOP; both paths showed up as the limiters of that kernel's loop.
// ================================================================================================
static constexpr int Q_BM = 256, Q_BN = 256, Q_BK = 32;
static constexpr int Q_A_BYTES = Q_BM * P_ROWB, Q_W_BYTES = Q_BN * P_ROWB, Q_STAGE = Q_A_BYTES + Q_W_BYTES;  // 32 KiB
static constexpr int Q_NSTAGE = 4;

// coalesced epilogue shared by the 256 x 256 kernels.  A wave owns 128 rows x (32 NTN) columns of the tile as
// acc[tn][tm] (tn: 32-column block, tm: 32-row block; in the transposed MFMA orientation lane (i, g) holds
// C[tm*32 + i][tn*32 + 8q + 4g + e] in acc[tn][tm][4q + e]).  Each pass stages one 32-row x 64-column fp32 tile in the
// wave's private LDS area (8.5 KiB) and writes it out as complete 128/256-byte row segments.
template <int EPI, int NTN>
__device__ __forceinline__ void nt_epilogue(f32x16_t (&acc)[NTN][4], char* er, const float* __restrict__ bias,
                                            bf16_t* __restrict__ out_t, bf16_t* __restrict__ out2_t, float* __restrict__ out_f,
                                            const float* __restrict__ resid, const bf16_t* __restrict__ aux, int M, int N,
                                            int row_base, int col_base, int lane) {
    const int i = lane & 31, g = lane >> 5;
    constexpr int EROW = 64 * 4 + 16;
    const int ec = (lane & 15) * 4, erow0 = lane >> 4;
#pragma unroll
    for (int h = 0; h < NTN / 2; ++h) {
        const int n = col_base + h * 64 + ec;
        float bb[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias && n < N) load4<float>(bias + n, bb);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(er + i * EROW + (tn * 32 + 8 * q + 4 * g) * 4) =
                        make_float4(acc[2 * h + tn][tm][4 * q], acc[2 * h + tn][tm][4 * q + 1], acc[2 * h + tn][tm][4 * q + 2],
                                    acc[2 * h + tn][tm][4 * q + 3]);
#pragma unroll 4
            for (int p = 0; p < 8; ++p) {
                const int rl = p * 4 + erow0, m = row_base + tm * 32 + rl;
                const float4 t4 = *reinterpret_cast<const float4*>(er + rl * EROW + ec * 4);
                if (m < M && n < N) {
                    float v[4] = {t4.x + bb[0], t4.y + bb[1], t4.z + bb[2], t4.w + bb[3]};
                    const size_t o = (size_t)m * N + n;
                    if (EPI == MBX_EPI_STORE) {
                        store4<bf16_t>(out_t + o, v);
                    } else if (EPI == MBX_EPI_GELU) {
                        if (out_t) store4<bf16_t>(out_t + o, v);   // pre-activation is only needed for backward
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
                        store4<bf16_t>(out2_t + o, v);
                    } else if (EPI == MBX_EPI_RESID) {
                        float r[4];
                        load4<float>(resid + o, r);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += r[e];
                        store4<float>(out_f + o, v);
                    } else if (EPI == MBX_EPI_TANH) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
                        store4<float>(out_f + o, v);
                    } else if (EPI == MBX_EPI_DGELU) {
                        float u[4];
                        load4<bf16_t>(aux + o, u);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= gelu_fast_grad(u[e]);
                        store4<bf16_t>(out_t + o, v);
                    }
                }
            }
        }
    }
}
// 8-wave layout: wave (wm, wn) = (wave >> 2, wave & 3) owns rows [128 wm, +128) x cols [64 wn, +64)
template <int EPI>
__device__ __forceinline__ void nt256_epilogue(f32x16_t (&acc)[2][4], char* smem, const float* __restrict__ bias,
                                               bf16_t* __restrict__ out_t, bf16_t* __restrict__ out2_t, float* __restrict__ out_f,
                                               const float* __restrict__ resid, const bf16_t* __restrict__ aux, int M, int N, int m0,
                                               int n0, int wave, int lane) {
    __builtin_amdgcn_s_barrier();   // every wave is done with the k-loop's LDS stages
    nt_epilogue<EPI, 2>(acc, smem + wave * (32 * (64 * 4 + 16)), bias, out_t, out2_t, out_f, resid, aux, M, N,
                        m0 + (wave >> 2) * 128, n0 + (wave & 3) * 64, lane);
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_pipe256_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                                 const float* __restrict__ bias, bf16_t* __restrict__ out_t,
                                                                 bf16_t* __restrict__ out2_t, float* __restrict__ out_f,
                                                                 const float* __restrict__ resid, const bf16_t* __restrict__ aux,
                                                                 int M, int N, int K, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 4 stages x 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = xcd_remap2(blockIdx.x, gridDim.x);
    const int n0 = (lid % ntn) * Q_BN, m0 = (lid / ntn) * Q_BM;
    const int wm = wave >> 2, wn = wave & 3;   // wave tile: rows [128 wm, +128), cols [64 wn, +64)

    // LDS-DMA: one instruction = 16 rows x 64 B; wave w fills rows [32 w, 32 w + 32) of A and of W (2 + 2 instr)
    const int lr = lane >> 2, lp = lane & 3;
    const bf16_t* srcA[2];
    const bf16_t* srcW[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 32 + i * 16 + lr;
        const int sw = (lp ^ ((row >> 2) & 3)) << 3;
        srcA[i] = A + (size_t)min(m0 + row, M - 1) * K + sw;
        srcW[i] = W + (size_t)min(n0 + row, N - 1) * K + sw;
    }
    char* dstA = smem + wave * 32 * P_ROWB;
    char* dstW = smem + Q_A_BYTES + wave * 32 * P_ROWB;

    f32x16_t acc[2][4];   // [tn][tm]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = K / Q_BK;
#define Q_ISSUE(kt_, stage_)                                                           \
    do {                                                                               \
        const size_t ko_ = (size_t)(kt_) * Q_BK;                                       \
        GLDS16(srcA[0] + ko_, dstA + (stage_) * Q_STAGE);                              \
        GLDS16(srcA[1] + ko_, dstA + (stage_) * Q_STAGE + 1024);                       \
        GLDS16(srcW[0] + ko_, dstW + (stage_) * Q_STAGE);                              \
        GLDS16(srcW[1] + ko_, dstW + (stage_) * Q_STAGE + 1024);                       \
    } while (0)

    Q_ISSUE(0, 0);
    if (nk > 1) Q_ISSUE(1, 1);
    if (nk > 2) Q_ISSUE(2, 2);
    const int i = lane & 31, g = lane >> 5;
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = nk - 1 - kt;                 // tiles issued after kt that may stay in flight (<= 2)
        if (ahead >= 2) WAIT_VMCNT(8); else if (ahead == 1) WAIT_VMCNT(4); else WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();                  // tile kt landed for everyone; stage (kt+3)%4 is free
        if (kt + 3 < nk) Q_ISSUE(kt + 3, (stage + 3) & 3);
        const char* sA = smem + stage * Q_STAGE;
        const char* sW = sA + Q_A_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8_t fw[2], fa[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) fw[t] = *reinterpret_cast<const bf16x8_t*>(sW + sw_off(wn * 64 + t * 32 + i, 2 * s + g));
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t] = *reinterpret_cast<const bf16x8_t*>(sA + sw_off(wm * 128 + t * 32 + i, 2 * s + g));
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[tn], fa[tm], acc[tn][tm], 0, 0, 0);
        }
        stage = (stage + 1) & 3;
    }
#undef Q_ISSUE

    nt256_epilogue<EPI>(acc, smem, bias, out_t, out2_t, out_f, resid, aux, M, N, m0, n0, wave, lane);
}

// ================================================================================================
// gemm_nt_rs256: the same 256 x 256 tile / 8-wave layout, but the k-tiles travel HBM/L2 -> VGPR -> LDS
// (global_load_dwordx4 + ds_write_b128) instead of through the LDS-DMA path, whose L2 -> LDS rate beside a
// running MFMA loop was measured at ~20 B/clk/CU (tools/probes/mfma_dma.hip) and caps the DMA kernels' loop.
// Three k-tiles are in flight in registers (48 VGPRs), tile kt+1 is written to the other LDS stage after
// the MFMAs of tile kt, one barrier per k-tile, two LDS stages.
// ================================================================================================
static constexpr int S_SMEM = 8 * 32 * (64 * 4 + 16) > 2 * Q_STAGE ? 8 * 32 * (64 * 4 + 16) : 2 * Q_STAGE;   // epilogue staging needs 68 KiB
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_rs256_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                               const float* __restrict__ bias, bf16_t* __restrict__ out_t,
                                                               bf16_t* __restrict__ out2_t, float* __restrict__ out_f,
                                                               const float* __restrict__ resid, const bf16_t* __restrict__ aux,
                                                               int M, int N, int K, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = xcd_remap2(blockIdx.x, gridDim.x);
    const int n0 = (lid % ntn) * Q_BN, m0 = (lid / ntn) * Q_BM;
    const int wm = wave >> 2, wn = wave & 3;

    // staging assignment = the DMA kernels' LDS image: wave w owns rows [32 w, 32 w + 32) of A and of W, lane l holds the
    // 16-byte chunk that lands at byte 16 l of each 16-row block (source chunk un-swizzled on the global side)
    const int lr = lane >> 2, lp = lane & 3;
    const bf16_t* srcA[2];
    const bf16_t* srcW[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = wave * 32 + q * 16 + lr;
        const int sw = (lp ^ ((row >> 2) & 3)) << 3;
        srcA[q] = A + (size_t)min(m0 + row, M - 1) * K + sw;
        srcW[q] = W + (size_t)min(n0 + row, N - 1) * K + sw;
    }
    char* dst = smem + wave * 32 * P_ROWB + lane * 16;

    f32x16_t acc[2][4];   // [tn][tm]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = K / Q_BK;
    typedef unsigned u32x4v_t __attribute__((ext_vector_type(4)));
    u32x4v_t s0a, s0b, s0c, s0d, s1a, s1b, s1c, s1d, s2a, s2b, s2c, s2d;   // three register slots x (A0, A1, W0, W1)
    // the loads are issued through inline asm so that the compiler's waitcnt pass does not see them: it would wait for
    // ALL outstanding loads (vmcnt(0)) before the first ds_write of a slot, which serialises the prefetch; the counted
    // waits below (loads return in order, no stores in the loop) keep the two newest tiles in flight instead.
#define S_LD1(dst_, ptr_) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst_) : "v"(ptr_) : "memory")
#define S_LOAD(kt_, a_, b_, c_, d_)                                         \
    do {                                                                    \
        const size_t ko_ = (size_t)(kt_) * Q_BK;                            \
        S_LD1(a_, srcA[0] + ko_);                                           \
        S_LD1(b_, srcA[1] + ko_);                                           \
        S_LD1(c_, srcW[0] + ko_);                                           \
        S_LD1(d_, srcW[1] + ko_);                                           \
    } while (0)
#define S_WRITE(stage_, a_, b_, c_, d_)                                     \
    do {                                                                    \
        char* d0_ = dst + (stage_) * Q_STAGE;                               \
        *reinterpret_cast<u32x4v_t*>(d0_) = a_;                             \
        *reinterpret_cast<u32x4v_t*>(d0_ + 1024) = b_;                      \
        *reinterpret_cast<u32x4v_t*>(d0_ + Q_A_BYTES) = c_;                 \
        *reinterpret_cast<u32x4v_t*>(d0_ + Q_A_BYTES + 1024) = d_;          \
    } while (0)
    const int i = lane & 31, g = lane >> 5;
#define S_COMPUTE(stage_)                                                                                                    \
    do {                                                                                                                     \
        const char* sA = smem + (stage_) * Q_STAGE;                                                                          \
        const char* sW = sA + Q_A_BYTES;                                                                                     \
        _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                                   \
            bf16x8_t fw[2], fa[4];                                                                                           \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                    \
                fw[t] = *reinterpret_cast<const bf16x8_t*>(sW + sw_off(wn * 64 + t * 32 + i, 2 * s_ + g));                   \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                    \
                fa[t] = *reinterpret_cast<const bf16x8_t*>(sA + sw_off(wm * 128 + t * 32 + i, 2 * s_ + g));                  \
            _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                                                 \
                _Pragma("unroll") for (int tm = 0; tm < 4; ++tm)                                                             \
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[tn], fa[tm], acc[tn][tm], 0, 0, 0);             \
        }                                                                                                                    \
    } while (0)
    // one k-tile: tile kt_ is in LDS stage kt_ & 1; slot CUR (this tile's old registers) is free for tile kt_ + 3,
    // slot NXT holds tile kt_ + 1 and goes to the other stage after the MFMAs
#define S_ITER(kt_, ca_, cb_, cc_, cd_, na_, nb_, nc_, nd_)                                                \
    do {                                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        __builtin_amdgcn_s_barrier();                                                                      \
        if ((kt_) + 3 < nk) S_LOAD((kt_) + 3, ca_, cb_, cc_, cd_);                                         \
        S_COMPUTE((kt_) & 1);                                                                              \
        if ((kt_) + 1 < nk) {                                                                              \
            if ((kt_) + 3 < nk) WAIT_VMCNT(8); else if ((kt_) + 2 < nk) WAIT_VMCNT(4); else WAIT_VMCNT(0); \
            S_WRITE(((kt_) + 1) & 1, na_, nb_, nc_, nd_);                                                  \
        }                                                                                                  \
    } while (0)

    S_LOAD(0, s0a, s0b, s0c, s0d);
    if (nk > 1) S_LOAD(1, s1a, s1b, s1c, s1d);
    if (nk > 2) S_LOAD(2, s2a, s2b, s2c, s2d);
    if (nk > 2) WAIT_VMCNT(8); else if (nk > 1) WAIT_VMCNT(4); else WAIT_VMCNT(0);
    S_WRITE(0, s0a, s0b, s0c, s0d);
    for (int kt = 0; kt < nk; kt += 3) {
        S_ITER(kt, s0a, s0b, s0c, s0d, s1a, s1b, s1c, s1d);
        if (kt + 1 < nk) S_ITER(kt + 1, s1a, s1b, s1c, s1d, s2a, s2b, s2c, s2d);
        if (kt + 2 < nk) S_ITER(kt + 2, s2a, s2b, s2c, s2d, s0a, s0b, s0c, s0d);
    }
#undef S_ITER
#undef S_COMPUTE
#undef S_WRITE
#undef S_LOAD
#undef S_LD1
    nt256_epilogue<EPI>(acc, smem, bias, out_t, out2_t, out_f, resid, aux, M, N, m0, n0, wave, lane);
}

// ================================================================================================
// gemm_nt_w4: 256 x 256 tile, FOUR waves (one per SIMD, up to 512 registers each) as 2 (M) x 2 (N), each wave
// 128 x 128 = 4 x 4 MFMA tiles (256 accumulator registers).  Why: the 8-wave kernels read 96 KiB of fragments out
// of LDS per 32-deep k-tile and the DMA writes 32 KiB into it -- 1024 clk of the 128 B/clk LDS port against 1031 clk
// of MFMA work, i.e. the loop is LDS-bandwidth-bound (the register-staged variant above, which bypasses the DMA
// path, runs at the same speed).  With 128 x 128 per wave a k-tile needs 8 fragment reads per 16 MFMAs instead of
// 6 per 8: 64 + 32 = 96 KiB per k-tile, 75 % of the port at full MFMA rate.  One wave per SIMD has nobody to hide
// its LDS latency behind, so the fragments are software-pipelined by hand: the reads of k-step s+1 are issued
// before the 16 MFMAs of k-step s; the barrier that publishes the next k-tile sits between the two k-steps.
// LDS-DMA ring of 4 stages x 32 KiB as in gemm_nt_pipe256 (each wave issues 8 DMA instructions per k-tile).
// ================================================================================================
template <int EPI, int NST>
__global__ __launch_bounds__(256, 1) void gemm_nt_w4_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ out_t,
                                                            bf16_t* __restrict__ out2_t, float* __restrict__ out_f,
                                                            const float* __restrict__ resid, const bf16_t* __restrict__ aux,
                                                            int M, int N, int K, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // NST stages x 32 KiB (NST - 1 k-tiles in flight)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = xcd_remap2(blockIdx.x, gridDim.x);
    const int n0 = (lid % ntn) * Q_BN, m0 = (lid / ntn) * Q_BM;
    const int wm = wave >> 1, wn = wave & 1;   // wave tile: rows [128 wm, +128), cols [128 wn, +128)

    // LDS-DMA: one instruction = 16 rows x 64 B; wave w fills rows [64 w, 64 w + 64) of A and of W (4 + 4 instr)
    const int lr = lane >> 2, lp = lane & 3;
    const bf16_t* srcA[4];
    const bf16_t* srcW[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wave * 64 + q * 16 + lr;
        const int sw = (lp ^ ((row >> 2) & 3)) << 3;
        srcA[q] = A + (size_t)min(m0 + row, M - 1) * K + sw;
        srcW[q] = W + (size_t)min(n0 + row, N - 1) * K + sw;
    }
    char* dstA = smem + wave * 64 * P_ROWB;
    char* dstW = smem + Q_A_BYTES + wave * 64 * P_ROWB;

    f32x16_t acc[4][4];   // [tn][tm]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = K / Q_BK;
#define W4_ISSUE(kt_, stage_)                                                          \
    do {                                                                               \
        const size_t ko_ = (size_t)(kt_) * Q_BK;                                       \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                             \
            GLDS16(srcA[q_] + ko_, dstA + (stage_) * Q_STAGE + q_ * 1024);             \
            GLDS16(srcW[q_] + ko_, dstW + (stage_) * Q_STAGE + q_ * 1024);             \
        }                                                                              \
    } while (0)
    const int i = lane & 31, g = lane >> 5;
    // fragment offsets inside a stage (k-step s adds chunk 2 s: the swizzle only touches chunk bits 0..1 -> xor 2 s... see below)
    int offA[4], offW[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        offA[t] = sw_off(wm * 128 + t * 32 + i, g);
        offW[t] = Q_A_BYTES + sw_off(wn * 128 + t * 32 + i, g);
    }
    // chunk index 2 s + g = (2 s) ^ g for g in {0, 1}; sw_off(row, c) = row * 64 + ((c ^ key) << 4) with key = (row >> 2) & 3, and
    // (2 s + g) ^ key = (g ^ key) ^ (2 s)  ->  k-step 1 is the k-step-0 address with bit 5 (chunk bit 1) flipped
#define W4_READ(buf_, stage_, s_)                                                                                \
    do {                                                                                                         \
        const char* st_ = smem + (stage_) * Q_STAGE;                                                             \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                          \
            fa[buf_][t] = *reinterpret_cast<const bf16x8_t*>(st_ + (offA[t] ^ ((s_) << 5)));                    \
            fw[buf_][t] = *reinterpret_cast<const bf16x8_t*>(st_ + (offW[t] ^ ((s_) << 5)));                    \
        }                                                                                                        \
    } while (0)
#define W4_MMA(buf_)                                                                                             \
    do {                                                                                                         \
        _Pragma("unroll") for (int tn = 0; tn < 4; ++tn)                                                         \
            _Pragma("unroll") for (int tm = 0; tm < 4; ++tm)                                                     \
                acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[buf_][tn], fa[buf_][tm], acc[tn][tm], 0, 0, 0); \
    } while (0)

    // pin the instruction mix of the two halves of an iteration (the machine scheduler otherwise sinks the fragment
    // reads of the next k-step to the end of the MFMA block, where their latency is exposed: one wave per SIMD)
#define W4_SCHED_READS()                                                  \
    do {                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                \
        _Pragma("unroll") for (int z_ = 0; z_ < 8; ++z_) {                \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            \
        }                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                \
    } while (0)
#define W4_SCHED_READS_DMA()                                              \
    do {                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                \
        _Pragma("unroll") for (int z_ = 0; z_ < 8; ++z_) {                \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            \
        }                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                \
    } while (0)
    bf16x8_t fa[2][4], fw[2][4];
    // The loop body is branch-free (one basic block: the compiler's lgkmcnt bookkeeping stays exact): past the end of K
    // the DMA re-fetches the last k-tile into a stage nobody reads any more, so every wait is "one newer group in flight".
    W4_ISSUE(0, 0);
    W4_ISSUE(min(1, nk - 1), 1);
    W4_ISSUE(min(2, nk - 1), 2);
    if (NST == 5) { W4_ISSUE(min(3, nk - 1), 3); WAIT_VMCNT(24); } else { WAIT_VMCNT(16); }
    __builtin_amdgcn_s_barrier();                      // tile 0 landed for everyone
    W4_READ(0, 0, 0);
    int stage = 0;
    for (int kt = 0; kt < nk - 1; ++kt) {
        const int nstage = stage + 1 == NST ? 0 : stage + 1;
        W4_READ(1, stage, 1);                          // fragments of k-step 1, in flight behind the MFMAs of k-step 0
        W4_MMA(0);
        W4_SCHED_READS();                              // 2 MFMAs, then one fragment read per MFMA, then 6 MFMAs of slack
        if (NST == 5) WAIT_VMCNT(16); else WAIT_VMCNT(8);   // own DMA of tile kt+1 landed (NST - 3 newer groups may still fly)
        __builtin_amdgcn_s_barrier();                  // ... for everyone; everyone is past tile kt-1 -> its stage is free
        W4_ISSUE(min(kt + NST - 1, nk - 1), (stage + NST - 1) % NST);
        W4_READ(0, nstage, 0);                         // first fragments of the next tile, behind the MFMAs of k-step 1
        W4_MMA(1);
        W4_SCHED_READS_DMA();                          // same, with one DMA issue beside every fragment read
        stage = nstage;
    }
    W4_READ(1, stage, 1);
    W4_MMA(0);
    W4_MMA(1);
#undef W4_MMA
#undef W4_SCHED_READS
#undef W4_SCHED_READS_DMA
#undef W4_READ
#undef W4_ISSUE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // incl. the re-fetched tail tiles: the epilogue reuses the stages
    __builtin_amdgcn_s_barrier();   // every wave is done with the k-loop's LDS stages
    nt_epilogue<EPI, 4>(acc, smem + wave * (32 * (64 * 4 + 16)), bias, out_t, out2_t, out_f, resid, aux, M, N, m0 + wm * 128,
                        n0 + wn * 128, lane);
}

// ================================================================================================
// gemm_nt_persist: persistent 256 x 256 kernel for the store-only epilogues (bf16 outputs: STORE, GELU).
//
// Measured on this chip (tools/probes/mfma_dma.hip, MBX_DBG ablations, MBX_TRACE stamps):
//   * LDS-DMA out of L2 sustains ~20 B/clk/CU beside a running MFMA loop -> the k-loop of a 256 x 256 tile
//     tops out near 1.3 PFLOP/s, a 256 x 128 tile near 1.0;
//   * the epilogue of a tile is HBM-write-bound (~3.9 TB/s chip-wide) and a CU's vector-memory pipe is in
//     order: while a workgroup bursts its stores, the LDS-DMA loads of every workgroup on that CU queue
//     behind them, so "loop + epilogue" times ADD (0.50 + 0.21 ms for the QKV GEMM) whatever the occupancy.
// Hence: one persistent workgroup per CU walks its tiles; the LDS-DMA ring runs across tile boundaries
// (no pipeline refill per tile), and a finished tile is kept as packed bf16 in 64 VGPRs and TRICKLED out
// during the next tile's first 16 k-iterations (through a small LDS staging tile so that every store
// instruction writes complete 128-byte lines), so the store queue never backs up.  The stores are issued
// BEFORE the iteration's LDS-DMA group: vmcnt is in order on gfx950 and counts stores, so "at most the two
// newest DMA groups in flight" is still vmcnt(8) however many stores were actually issued.
// ================================================================================================
typedef unsigned u32x2v_t __attribute__((ext_vector_type(2)));
static constexpr int R_STAGE_OFF = Q_NSTAGE * Q_STAGE;                 // 8 waves x [16][64] bf16 staging tiles (18 KiB)
static constexpr int R_BIAS_OFF = R_STAGE_OFF + 8 * 16 * (64 * 2 + 16);  // bias behind it (<= 1536 floats)

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_persist_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                                 const float* __restrict__ bias, bf16_t* __restrict__ out_t,
                                                                 bf16_t* __restrict__ out2_t, int M, int N, int K, int ntn,
                                                                 int ntiles, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 4 stages x 32 KiB + bias
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int i = lane & 31, g = lane >> 5;
    const int lr = lane >> 2, lp = lane & 3;
    float* sbias = reinterpret_cast<float*>(smem + R_BIAS_OFF);
    for (int c = tid; c < N; c += 512) sbias[c] = bias ? bias[c] : 0.f;

    const int nk = K / Q_BK;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles v = blockIdx.x + j * gridDim.x
    const int f_total = my_tiles * nk;

    // ---- LDS-DMA issue state (runs three k-tiles ahead of the compute state, across tile boundaries) ----
    const bf16_t* srcA[2];
    const bf16_t* srcW[2];
    int is_j = 0, is_kt = 0, is_stage = 0;
    char* dstA = smem + wave * 32 * P_ROWB;
    char* dstW = smem + Q_A_BYTES + wave * 32 * P_ROWB;
#define R_SETUP(j_)                                                                    \
    do {                                                                               \
        const int lid_ = xcd_remap2((int)blockIdx.x + (j_) * (int)gridDim.x, ntiles);  \
        const int n0_ = (lid_ % ntn) * Q_BN, m0_ = (lid_ / ntn) * Q_BM;                \
        _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                             \
            const int row_ = wave * 32 + q_ * 16 + lr;                                 \
            const int sw_ = (lp ^ ((row_ >> 2) & 3)) << 3;                             \
            srcA[q_] = A + (size_t)min(m0_ + row_, M - 1) * K + sw_;                   \
            srcW[q_] = W + (size_t)min(n0_ + row_, N - 1) * K + sw_;                   \
        }                                                                              \
    } while (0)
#define R_ISSUE()                                                                      \
    do {                                                                               \
        const size_t ko_ = (size_t)is_kt * Q_BK;                                       \
        GLDS16(srcA[0] + ko_, dstA + is_stage * Q_STAGE);                              \
        GLDS16(srcA[1] + ko_, dstA + is_stage * Q_STAGE + 1024);                       \
        GLDS16(srcW[0] + ko_, dstW + is_stage * Q_STAGE);                              \
        GLDS16(srcW[1] + ko_, dstW + is_stage * Q_STAGE + 1024);                       \
        is_stage = (is_stage + 1) & 3;                                                 \
        if (++is_kt == nk) { is_kt = 0; ++is_j; if (is_j < my_tiles) R_SETUP(is_j); }  \
    } while (0)

    if (my_tiles > 0) R_SETUP(0);
    int issued = 0;
    for (; issued < 3 && issued < f_total; ++issued) R_ISSUE();

    f32x16_t acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    uint32_t pend[2][4][8];          // previous tile, bf16 pairs: pend[tn][tm][2 q + h] = columns 8 q + 4 g + 2 h, +1
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 8; ++r) pend[a][b][r] = 0u;
    bool pend_valid = false;
    int pm0 = 0, pn0 = 0;

    // ---- trickled, COALESCED stores of the pending tile -------------------------------------------------
    // (a first version stored the accumulator layout directly, 16-byte pieces of 32 rows per instruction:
    //  PMC showed 1.29 GB written for 0.81 GB of output plus 0.2 GB of read-for-ownership fetches, because
    //  partially written lines were evicted from L2 -- and it was slower than the burst epilogue.)
    // The pending tile (packed bf16 in `pend`) goes out in 8 half-blocks of 16 rows per wave: in k-iteration
    // 2h the 32 lanes owning those rows drop their 8 quads into a wave-private [16][64] staging tile in LDS,
    // in k-iteration 2h+1 every lane reads 16 contiguous bytes back and stores them: one instruction covers
    // 8 complete 128-byte lines.
    constexpr int SROW = 64 * 2 + 16;                                  // staging row: 64 bf16 + pad
    char* stg = smem + R_STAGE_OFF + wave * (16 * SROW);
#define R_STAGE_WRITE(H_)                                                                                  \
    do {                                                                                                   \
        if (pend_valid && ((i >> 4) == ((H_) & 1))) {                                                      \
            _Pragma("unroll") for (int tn_ = 0; tn_ < 2; ++tn_)                                            \
                _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                           \
                    *reinterpret_cast<uint2*>(stg + (i & 15) * SROW + (tn_ * 32 + 8 * q_ + 4 * g) * 2) =   \
                        make_uint2(pend[tn_][(H_) >> 1][2 * q_], pend[tn_][(H_) >> 1][2 * q_ + 1]);        \
        }                                                                                                  \
    } while (0)
#define R_STAGE_STORE(H_)                                                                                  \
    do {                                                                                                   \
        _Pragma("unroll") for (int ps_ = 0; ps_ < 2; ++ps_) {                                              \
            const int rl_ = ps_ * 8 + (lane >> 3);                                                         \
            const uint4 v_ = *reinterpret_cast<const uint4*>(stg + rl_ * SROW + (lane & 7) * 16);          \
            const int m_ = pm0 + wm * 128 + ((H_) >> 1) * 32 + ((H_) & 1) * 16 + rl_;                      \
            const int n_ = pn0 + wn * 64 + (lane & 7) * 8;                                                 \
            if (pend_valid && m_ < M && n_ < N) {                                                          \
                const size_t o_ = (size_t)m_ * N + n_;                                                     \
                *reinterpret_cast<uint4*>(out_t + o_) = v_;                                                \
                if (EPI == MBX_EPI_GELU) {                                                                 \
                    const uint32_t w_[4] = {v_.x, v_.y, v_.z, v_.w};                                       \
                    uint32_t g_[4];                                                                        \
                    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_)                                       \
                        g_[e_] = pack_bf2(gelu_fast(__uint_as_float(w_[e_] << 16)), gelu_fast(__uint_as_float(w_[e_] & 0xffff0000u))); \
                    *reinterpret_cast<uint4*>(out2_t + o_) = make_uint4(g_[0], g_[1], g_[2], g_[3]);       \
                }                                                                                          \
            }                                                                                              \
        }                                                                                                  \
    } while (0)

    // one k-iteration; KI (0..15): even KI stages half-block KI/2 of the pending tile, odd KI stores it
#define R_ITER(KI, TRICKLE)                                                                                \
    do {                                                                                                   \
        const int ahead_ = f_total - 1 - f;                                                                \
        if (ahead_ >= 2) WAIT_VMCNT(8); else if (ahead_ == 1) WAIT_VMCNT(4); else WAIT_VMCNT(0);           \
        __builtin_amdgcn_s_barrier();                                                                      \
        if (TRICKLE) { if (((KI) & 1) == 0) R_STAGE_WRITE((KI) >> 1); else R_STAGE_STORE((KI) >> 1); }     \
        if (issued < f_total) { R_ISSUE(); ++issued; }                                                     \
        const char* sA_ = smem + stage * Q_STAGE;                                                          \
        const char* sW_ = sA_ + Q_A_BYTES;                                                                 \
        _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                 \
            bf16x8_t fw_[2], fa_[4];                                                                       \
            _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_)                                               \
                fw_[t_] = *reinterpret_cast<const bf16x8_t*>(sW_ + sw_off(wn * 64 + t_ * 32 + i, 2 * s_ + g));   \
            _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_)                                               \
                fa_[t_] = *reinterpret_cast<const bf16x8_t*>(sA_ + sw_off(wm * 128 + t_ * 32 + i, 2 * s_ + g));  \
            _Pragma("unroll") for (int tn_ = 0; tn_ < 2; ++tn_)                                            \
                _Pragma("unroll") for (int tm_ = 0; tm_ < 4; ++tm_)                                        \
                    acc[tn_][tm_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw_[tn_], fa_[tm_], acc[tn_][tm_], 0, 0, 0);   \
        }                                                                                                  \
        stage = (stage + 1) & 3;                                                                           \
        ++f;                                                                                               \
    } while (0)

    __syncthreads();   // bias staged
    int stage = 0, f = 0;
    for (int j = 0; j < my_tiles; ++j) {
        // first 16 k-iterations: compute this tile, trickle the previous one
        R_ITER(0, true);  R_ITER(1, true);  R_ITER(2, true);  R_ITER(3, true);
        R_ITER(4, true);  R_ITER(5, true);  R_ITER(6, true);  R_ITER(7, true);
        R_ITER(8, true);  R_ITER(9, true);  R_ITER(10, true); R_ITER(11, true);
        R_ITER(12, true); R_ITER(13, true); R_ITER(14, true); R_ITER(15, true);
        for (int kt = 16; kt < nk; ++kt) R_ITER(0, false);
        // tile finished: accumulators (+ bias) -> packed bf16 pending registers
        const int lid = xcd_remap2((int)blockIdx.x + j * (int)gridDim.x, ntiles);
        pn0 = (lid % ntn) * Q_BN;
        pm0 = (lid / ntn) * Q_BM;
        pend_valid = true;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nc = min(pn0 + wn * 64 + tn * 32 + 8 * q + 4 * g, N - 4);
                const float4 b4 = *reinterpret_cast<const float4*>(sbias + nc);
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) {
                    pend[tn][tm][2 * q] = pack_bf2(acc[tn][tm][4 * q] + b4.x, acc[tn][tm][4 * q + 1] + b4.y);
                    pend[tn][tm][2 * q + 1] = pack_bf2(acc[tn][tm][4 * q + 2] + b4.z, acc[tn][tm][4 * q + 3] + b4.w);
                }
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    }
    // last tile: nothing left to hide behind -> the same staged stores as one burst
    R_STAGE_WRITE(0); R_STAGE_STORE(0); R_STAGE_WRITE(1); R_STAGE_STORE(1);
    R_STAGE_WRITE(2); R_STAGE_STORE(2); R_STAGE_WRITE(3); R_STAGE_STORE(3);
    R_STAGE_WRITE(4); R_STAGE_STORE(4); R_STAGE_WRITE(5); R_STAGE_STORE(5);
    R_STAGE_WRITE(6); R_STAGE_STORE(6); R_STAGE_WRITE(7); R_STAGE_STORE(7);
#undef R_STAGE_WRITE
#undef R_STAGE_STORE
#undef R_ITER
#undef R_ISSUE
#undef R_SETUP
}

template <typename K>
static int set_lds_attr(K kernel, size_t bytes, const char* who) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return mbx_set_error("%s: hipFuncSetAttribute(%zu): %s", who, bytes, hipGetErrorString(e));
    return 0;
}

static int launch_nt256(const void* a, const void* w, const float* bias, int epi, void* out_t, void* out2_t, float* out_f,
                        const float* resid, const void* aux, int M, int N, int K, hipStream_t s) {
    const int ntn = (N + Q_BN - 1) / Q_BN, ntm = (M + Q_BM - 1) / Q_BM;
    dim3 grid((unsigned)ntn * ntm), block(512);
    static const int w4 = [] { const char* e = getenv("MBX_NT_W4"); return e ? atoi(e) : 0; }();
    if (w4) {
        const size_t shm_w4 = (size_t)(w4 == 5 ? 5 : 4) * Q_STAGE;
#define MBX_W_CASE(E)                                                                                             \
    case E:                                                                                                       \
        if (w4 == 5) {                                                                                            \
            if (set_lds_attr(gemm_nt_w4_kernel<E, 5>, shm_w4, "gemm_nt_w4")) return 1;                            \
            hipLaunchKernelGGL((gemm_nt_w4_kernel<E, 5>), grid, dim3(256), shm_w4, s, (const bf16_t*)a, (const bf16_t*)w, bias, \
                               (bf16_t*)out_t, (bf16_t*)out2_t, out_f, resid, (const bf16_t*)aux, M, N, K, ntn);  \
        } else {                                                                                                  \
            if (set_lds_attr(gemm_nt_w4_kernel<E, 4>, shm_w4, "gemm_nt_w4")) return 1;                            \
            hipLaunchKernelGGL((gemm_nt_w4_kernel<E, 4>), grid, dim3(256), shm_w4, s, (const bf16_t*)a, (const bf16_t*)w, bias, \
                               (bf16_t*)out_t, (bf16_t*)out2_t, out_f, resid, (const bf16_t*)aux, M, N, K, ntn);  \
        }                                                                                                         \
        break;
        switch (epi) {
            MBX_W_CASE(MBX_EPI_STORE)
            MBX_W_CASE(MBX_EPI_GELU)
            MBX_W_CASE(MBX_EPI_RESID)
            MBX_W_CASE(MBX_EPI_TANH)
            MBX_W_CASE(MBX_EPI_DGELU)
            default: return mbx_set_error("gemm_nt: unknown epilogue %d", epi);
        }
#undef MBX_W_CASE
        MBX_LAUNCH_CHECK("gemm_nt_w4");
        return 0;
    }
    static const int rs = [] { const char* e = getenv("MBX_NT_RS"); return e ? atoi(e) : 0; }();
    if (rs) {
        const size_t shm_rs = S_SMEM;
#define MBX_S_CASE(E)                                                                                               \
    case E:                                                                                                         \
        if (set_lds_attr(gemm_nt_rs256_kernel<E>, shm_rs, "gemm_nt_rs256")) return 1;                               \
        hipLaunchKernelGGL((gemm_nt_rs256_kernel<E>), grid, block, shm_rs, s, (const bf16_t*)a, (const bf16_t*)w, bias, \
                           (bf16_t*)out_t, (bf16_t*)out2_t, out_f, resid, (const bf16_t*)aux, M, N, K, ntn);        \
        break;
        switch (epi) {
            MBX_S_CASE(MBX_EPI_STORE)
            MBX_S_CASE(MBX_EPI_GELU)
            MBX_S_CASE(MBX_EPI_RESID)
            MBX_S_CASE(MBX_EPI_TANH)
            MBX_S_CASE(MBX_EPI_DGELU)
            default: return mbx_set_error("gemm_nt: unknown epilogue %d", epi);
        }
#undef MBX_S_CASE
        MBX_LAUNCH_CHECK("gemm_nt_rs256");
        return 0;
    }
    const size_t shm = Q_NSTAGE * Q_STAGE;
#define MBX_Q_CASE(E)                                                                                                 \
    case E:                                                                                                           \
        if (set_lds_attr(gemm_nt_pipe256_kernel<E>, shm, "gemm_nt_pipe256")) return 1;                                \
        hipLaunchKernelGGL((gemm_nt_pipe256_kernel<E>), grid, block, shm, s, (const bf16_t*)a, (const bf16_t*)w, bias, \
                           (bf16_t*)out_t, (bf16_t*)out2_t, out_f, resid, (const bf16_t*)aux, M, N, K, ntn);          \
        break;
    switch (epi) {
        MBX_Q_CASE(MBX_EPI_STORE)
        MBX_Q_CASE(MBX_EPI_GELU)
        MBX_Q_CASE(MBX_EPI_RESID)
        MBX_Q_CASE(MBX_EPI_TANH)
        MBX_Q_CASE(MBX_EPI_DGELU)
        default: return mbx_set_error("gemm_nt: unknown epilogue %d", epi);
    }
#undef MBX_Q_CASE
    MBX_LAUNCH_CHECK("gemm_nt_pipe256");
    return 0;
}

int mbx_launch_gemm_nt_pipe(const void* a, const void* w, const float* bias, int epi, void* out_t, void* out2_t, float* out_f,
                            const float* resid, const void* aux, int M, int N, int K, hipStream_t s) {
    static const int use256 = [] { const char* e = getenv("MBX_NT256"); return e ? atoi(e) : 1; }();
    // measured (tools/gemm_bench.py, M = 264k): the 256 x 256 kernel wins 8-11 % where the epilogue only stores
    // (qkv, fc1, dX GEMMs); with a second HBM stream in the epilogue (residual / GELU' input) two smaller
    // workgroups per CU are faster.
    static const int persist = [] { const char* e = getenv("MBX_NT_PERSIST"); return e ? atoi(e) : 0; }();   // experiment, off: see DESIGN.md
    if (persist && out_t && (epi == MBX_EPI_STORE || epi == MBX_EPI_GELU) && N >= 256 && N <= 1536 && N % 8 == 0 && K % (16 * Q_BK) == 0) {
        const int ntn_p = (N + Q_BN - 1) / Q_BN, ntiles = ntn_p * ((M + Q_BM - 1) / Q_BM);
        const int grid_p = ntiles < 256 ? ntiles : 256;          // one persistent workgroup per CU
        const size_t shm_p = R_BIAS_OFF + 1536 * sizeof(float);
        static const int dbg_p = [] { const char* e = getenv("MBX_DBG"); return e ? atoi(e) : 0; }();
        if (epi == MBX_EPI_STORE) {
            if (set_lds_attr(gemm_nt_persist_kernel<MBX_EPI_STORE>, shm_p, "gemm_nt_persist")) return 1;
            hipLaunchKernelGGL((gemm_nt_persist_kernel<MBX_EPI_STORE>), dim3(grid_p), dim3(512), shm_p, s, (const bf16_t*)a,
                               (const bf16_t*)w, bias, (bf16_t*)out_t, (bf16_t*)out2_t, M, N, K, ntn_p, ntiles, dbg_p);
        } else {
            if (set_lds_attr(gemm_nt_persist_kernel<MBX_EPI_GELU>, shm_p, "gemm_nt_persist")) return 1;
            hipLaunchKernelGGL((gemm_nt_persist_kernel<MBX_EPI_GELU>), dim3(grid_p), dim3(512), shm_p, s, (const bf16_t*)a,
                               (const bf16_t*)w, bias, (bf16_t*)out_t, (bf16_t*)out2_t, M, N, K, ntn_p, ntiles, dbg_p);
        }
        MBX_LAUNCH_CHECK("gemm_nt_persist");
        return 0;
    }
    const bool light_epi = epi == MBX_EPI_STORE || epi == MBX_EPI_GELU || epi == MBX_EPI_TANH;
    if (use256 && light_epi && N >= 256) return launch_nt256(a, w, bias, epi, out_t, out2_t, out_f, resid, aux, M, N, K, s);
    const int ntn = (N + P_BN - 1) / P_BN, ntm = (M + P_BM - 1) / P_BM;
    dim3 grid((unsigned)ntn * ntm), block(512);
    const size_t shm = P_NSTAGE * P_STAGE;
    static const int dbg = [] { const char* e = getenv("MBX_DBG"); return e ? atoi(e) : 0; }();
    static long long* const trace = [] { const char* e = getenv("MBX_TRACE_BUF"); return e ? (long long*)strtoull(e, nullptr, 0) : (long long*)nullptr; }();
#define MBX_NTP_CASE(E)                                                                                               \
    case E:                                                                                                           \
        if (set_lds_attr(gemm_nt_pipe_kernel<E>, shm, "gemm_nt_pipe")) return 1;                                      \
        hipLaunchKernelGGL((gemm_nt_pipe_kernel<E>), grid, block, shm, s, (const bf16_t*)a, (const bf16_t*)w, bias,   \
                           (bf16_t*)out_t, (bf16_t*)out2_t, out_f, resid, (const bf16_t*)aux, M, N, K, ntn, dbg, trace);          \
        break;
    switch (epi) {
        MBX_NTP_CASE(MBX_EPI_STORE)
        MBX_NTP_CASE(MBX_EPI_GELU)
        MBX_NTP_CASE(MBX_EPI_RESID)
        MBX_NTP_CASE(MBX_EPI_TANH)
        MBX_NTP_CASE(MBX_EPI_DGELU)
        default: return mbx_set_error("gemm_nt: unknown epilogue %d", epi);
    }
#undef MBX_NTP_CASE
    MBX_LAUNCH_CHECK("gemm_nt_pipe");
    return 0;
}

// ================================================================================================
// gemm_tn_pipe : dW[N,K] = dY[M,N]^T . A[M,K]
// ================================================================================================
// stage: dY tile [64 tokens][256 n] (512 B rows, 32 KiB) + A tile [64 tokens][128 k] (256 B rows, 16 KiB).
// 64-byte chunk c of token row r sits at physical chunk c ^ (r & 3) (XOR on the low two bits of the
// chunk index): the four token rows touched by one transpose read then fall into four different
// 64-byte bank quarters.
static constexpr int T_BN = 256, T_BK = 128, T_BMS = 64;
static constexpr int T_Y_BYTES = T_BMS * T_BN * 2, T_A_BYTES = T_BMS * T_BK * 2, T_STAGE = T_Y_BYTES + T_A_BYTES;  // 48 KiB

// byte offset of (token row r, element column c) in a tile with ROWB bytes per row
template <int ROWB>
__device__ __forceinline__ int tr_off(int r, int c) {
    const int byte = c * 2, c64 = byte >> 6;
    return r * ROWB + (((c64 & ~3) | ((c64 ^ r) & 3)) << 6) + (byte & 63);
}

// MFMA operand fragment (8 tokens of one column per lane) for columns [col0, col0 + 32) and tokens
// [tok0 + 8 g, tok0 + 8 g + 8):  two transpose reads of 4 tokens each.
typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s_t lds_v4s_t;
template <int ROWB>
__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int tok0, int col0, int lane) {
    const int g = lane >> 5, r16 = lane & 15, nb = col0 + 16 * ((lane >> 4) & 1) + 4 * (r16 & 3);
    const int t = tok0 + 8 * g + (r16 >> 2);
    union { v4s_t h[2]; bf16x8_t v; } f;
    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(tile + tr_off<ROWB>(t, nb)));
    f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(tile + tr_off<ROWB>(t + 4, nb)));
    return f.v;
}

__global__ __launch_bounds__(512, 2) void gemm_tn_pipe_kernel(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ A,
                                                              float* __restrict__ part_w, float* __restrict__ part_b, int M,
                                                              int N, int K, int ntk, int ntiles, int nsplits,
                                                              int chunks_per_split, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 3 stages x 48 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware order: workgroup ids round-robin over the 8 XCDs (private L2s).  XCD x owns the token
    // splits x, x+8, ... and walks (split, tile) in order, so the ntn*ntk tiles that stream the SAME token
    // range are resident on one XCD at about the same time and share its L2.
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int split = (idx / ntiles) * 8 + xcd, tile = idx % ntiles;
    if (split >= nsplits) return;
    const int n0 = (tile / ntk) * T_BN, k0 = (tile % ntk) * T_BK;
    const int wr = wave >> 1, wc = wave & 1;  // wave grid 4 (n) x 2 (k), 64 x 64 each
    const int nchunks = (M + T_BMS - 1) / T_BMS;
    const int c_beg = split * chunks_per_split, c_end = min(nchunks, c_beg + chunks_per_split);

    // LDS-DMA: dY tile = 32 instructions of 1 KiB (2 token rows each); A tile = 16 instructions (4 rows each).
    // wave w: dY rows [8 w, 8 w + 8) (4 instr), A rows [8 w, 8 w + 8) (2 instr).
    // dY instr i: rows 8w + 2i + (lane >> 5), physical 16-byte chunk p = lane & 31 (row has 32 chunks)
    // A  instr i: rows 8w + 4i + (lane >> 4), physical chunk p = lane & 15 (row has 16 chunks)
    // physical chunk p holds logical chunk p ^ ((row & 3) << 2)   [XOR on the 64-byte-chunk bits]
    int yrow[4], arow[2];
    size_t ysrc[4], asrc[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        yrow[i] = wave * 8 + 2 * i + (lane >> 5);
        const int p = lane & 31, lc = p ^ ((yrow[i] & 3) << 2);
        const int col = min(n0 + lc * 8, N - 8);  // columns past N are never stored; clamp keeps the read in bounds
        ysrc[i] = (size_t)col;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        arow[i] = wave * 8 + 4 * i + (lane >> 4);
        const int p = lane & 15, lc = p ^ ((arow[i] & 3) << 2);
        const int col = min(k0 + lc * 8, K - 8);
        asrc[i] = (size_t)col;
    }
    char* dstY = smem + wave * 8 * 512;
    char* dstA = smem + T_Y_BYTES + wave * 8 * 256;

#define TN_ISSUE(chunk_, stage_)                                                                    \
    do {                                                                                            \
        const int mb_ = (chunk_) * T_BMS;                                                           \
        char* dy_ = dstY + (stage_) * T_STAGE;                                                      \
        char* da_ = dstA + (stage_) * T_STAGE;                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                            \
            GLDS16(dY + (size_t)min(mb_ + yrow[i_], M - 1) * N + ysrc[i_], dy_ + i_ * 1024);        \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                            \
            GLDS16(A + (size_t)min(mb_ + arow[i_], M - 1) * K + asrc[i_], da_ + i_ * 1024);         \
    } while (0)

    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // bias gradient = column sums of dY = dY^T . 1: the k-tile-0 workgroups multiply their dY fragments with an
    // all-ones B operand (one extra MFMA per fragment in the waves of wave-column 0) -- no extra pass over dY.
    const bool want_db = (part_b != nullptr) && (k0 == 0) && (wc == 0);
    f32x16_t accb[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[a][r] = 0.f;
    union { uint32_t u[4]; bf16x8_t v; } ones;
    ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = 0x3f803f80u;

    const int nc = c_end - c_beg;
    if (!(dbg & 2)) {
        if (nc > 0) TN_ISSUE(c_beg, 0);
        if (nc > 1) TN_ISSUE(c_beg + 1, 1);
    }
    int stage = 0;
    for (int c = 0; c < nc; ++c) {
        if (c + 1 < nc) WAIT_VMCNT(6); else WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();
        if (c + 2 < nc && !(dbg & 2)) {
            const int st2 = stage >= 1 ? stage - 1 : 2;
            TN_ISSUE(c_beg + c + 2, st2);
        }
        const char* sY = smem + stage * T_STAGE;
        const char* sA = sY + T_Y_BYTES;
        // tokens past M in the last chunk: rows were clamped to M-1 (duplicates) -> zero their contribution
        const int valid = M - (c_beg + c) * T_BMS;   // tokens of this chunk that exist (>= 64 for all but the last)
        if (!(dbg & 1))
#pragma unroll
        for (int s = 0; s < 4; ++s) {                // 16 tokens per MFMA k-step
            bf16x8_t fy[2], fa[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fy[t] = tr_frag<512>(sY, 16 * s, wr * 64 + t * 32, lane);
                fa[t] = tr_frag<256>(sA, 16 * s, wc * 64 + t * 32, lane);
            }
            if (valid < T_BMS) {
                // lane holds tokens 16 s + 8 g + e (e < 8) in elements e: zero the ones >= valid (one operand suffices)
                const int tb = 16 * s + 8 * (lane >> 5);
                union { bf16x8_t v; uint16_t h[8]; } z;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    z.v = fy[t];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (tb + e >= valid) z.h[e] = 0;
                    fy[t] = z.v;
                }
            }
#pragma unroll
            for (int tr = 0; tr < 2; ++tr)
#pragma unroll
                for (int tc = 0; tc < 2; ++tc)
                    acc[tr][tc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[tr], fa[tc], acc[tr][tc], 0, 0, 0);
            if (want_db) {
#pragma unroll
                for (int tr = 0; tr < 2; ++tr) accb[tr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[tr], ones.v, accb[tr], 0, 0, 0);
            }
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
#undef TN_ISSUE
    if (want_db && (lane & 31) == 0) {   // every column of accb holds the same sums: column 0 of each half writes them
        const int g2 = lane >> 5;
#pragma unroll
        for (int tr = 0; tr < 2; ++tr)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wr * 64 + tr * 32 + (r & 3) + 8 * (r >> 2) + 4 * g2;
                if (n < N) part_b[(size_t)split * N + n] = accb[tr][r];
            }
    }

    float* pw = part_w + (size_t)split * N * K;
    const int i = lane & 31, g = lane >> 5;
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc) {
            const int k = k0 + wc * 64 + tc * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wr * 64 + tr * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (k < K && n < N) pw[(size_t)n * K + k] = acc[tr][tc][r];
            }
        }
}

// ================================================================================================
// gemm_tn_pipe256: 256 (n) x 256 (k) output tile, stage = dY [32 tokens][256 n] + A [32 tokens][256 k] (32 KiB),
// 4-stage ring, waves 2 (n) x 4 (k) with 128 x 64 wave tiles: 12 transpose reads per 8 MFMAs instead of 8 per 4
// (the 256 x 128 kernel saturates the LDS: 64 ds_read_b64_tr_b16 per wave per 16 MFMAs) and 1/3 less LDS-DMA.
// ================================================================================================
static constexpr int U_BN = 256, U_BK = 256, U_BMS = 32;
static constexpr int U_TILE = U_BMS * 512, U_STAGE = 2 * U_TILE;   // 16 KiB per operand tile, 32 KiB per stage

__global__ __launch_bounds__(512, 2) void gemm_tn_pipe256_kernel(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ A,
                                                                 float* __restrict__ part_w, float* __restrict__ part_b, int M,
                                                                 int N, int K, int ntk, int ntiles, int nsplits,
                                                                 int chunks_per_split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 4 stages x 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int split = (idx / ntiles) * 8 + xcd, tile = idx % ntiles;
    if (split >= nsplits) return;
    const int n0 = (tile / ntk) * U_BN, k0 = (tile % ntk) * U_BK;
    const int wr = wave >> 2, wc = wave & 3;   // wave tile: n rows [128 wr, +128), k cols [64 wc, +64)
    const int nchunks = (M + U_BMS - 1) / U_BMS;
    const int c_beg = split * chunks_per_split, c_end = min(nchunks, c_beg + chunks_per_split);
    const int nc = c_end - c_beg;

    // LDS-DMA: each operand tile = 16 instructions of 1 KiB (2 token rows of 512 B); wave w: rows 4 w + 2 i + (lane >> 5)
    int rowv[2];
    size_t ycol[2], acol[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        rowv[i] = wave * 4 + 2 * i + (lane >> 5);
        const int lc = (lane & 31) ^ ((rowv[i] & 3) << 2);
        ycol[i] = (size_t)min(n0 + lc * 8, N - 8);
        acol[i] = (size_t)min(k0 + lc * 8, K - 8);
    }
    char* dstY = smem + wave * 4 * 512;
    char* dstA = dstY + U_TILE;
#define U_ISSUE(chunk_, stage_)                                                                      \
    do {                                                                                             \
        const int mb_ = (chunk_) * U_BMS;                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                           \
            const size_t r_ = (size_t)min(mb_ + rowv[i_], M - 1);                                    \
            GLDS16(dY + r_ * N + ycol[i_], dstY + (stage_) * U_STAGE + i_ * 1024);                   \
            GLDS16(A + r_ * K + acol[i_], dstA + (stage_) * U_STAGE + i_ * 1024);                    \
        }                                                                                            \
    } while (0)

    f32x16_t acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const bool want_db = (part_b != nullptr) && (k0 == 0) && (wc == 0);
    f32x16_t accb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[a][r] = 0.f;
    union { uint32_t u[4]; bf16x8_t v; } ones;
    ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = 0x3f803f80u;

    if (nc > 0) U_ISSUE(c_beg, 0);
    if (nc > 1) U_ISSUE(c_beg + 1, 1);
    if (nc > 2) U_ISSUE(c_beg + 2, 2);
    int stage = 0;
    for (int c = 0; c < nc; ++c) {
        const int ahead = nc - 1 - c;
        if (ahead >= 2) WAIT_VMCNT(8); else if (ahead == 1) WAIT_VMCNT(4); else WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();
        if (c + 3 < nc) U_ISSUE(c_beg + c + 3, (stage + 3) & 3);
        const char* sY = smem + stage * U_STAGE;
        const char* sA = sY + U_TILE;
        const int valid = M - (c_beg + c) * U_BMS;   // tokens of this chunk that exist
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8_t fy[4], fa[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) fy[t] = tr_frag<512>(sY, 16 * s, wr * 128 + t * 32, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t) fa[t] = tr_frag<512>(sA, 16 * s, wc * 64 + t * 32, lane);
            if (valid < U_BMS) {
                const int tb = 16 * s + 8 * (lane >> 5);
                union { bf16x8_t v; uint16_t h[8]; } z;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    z.v = fy[t];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (tb + e >= valid) z.h[e] = 0;
                    fy[t] = z.v;
                }
            }
#pragma unroll
            for (int tr = 0; tr < 4; ++tr)
#pragma unroll
                for (int tc = 0; tc < 2; ++tc)
                    acc[tr][tc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[tr], fa[tc], acc[tr][tc], 0, 0, 0);
            if (want_db) {
#pragma unroll
                for (int tr = 0; tr < 4; ++tr) accb[tr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[tr], ones.v, accb[tr], 0, 0, 0);
            }
        }
        stage = (stage + 1) & 3;
    }
#undef U_ISSUE
    const int i = lane & 31, g = lane >> 5;
    if (want_db && i == 0) {
#pragma unroll
        for (int tr = 0; tr < 4; ++tr)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wr * 128 + tr * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (n < N) part_b[(size_t)split * N + n] = accb[tr][r];
            }
    }
    float* pw = part_w + (size_t)split * N * K;
#pragma unroll
    for (int tr = 0; tr < 4; ++tr)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc) {
            const int k = k0 + wc * 64 + tc * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wr * 128 + tr * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (k < K && n < N) pw[(size_t)n * K + k] = acc[tr][tc][r];
            }
        }
}

static bool tn_use256(int N, int K) {
    static const int en = [] { const char* e = getenv("MBX_TN256"); return e ? atoi(e) : 1; }();
    return en && N >= 256 && K >= 256;
}
static int tnp_splits(int M, int N, int K) {
    const bool big = tn_use256(N, K);
    const int tiles = big ? ((N + U_BN - 1) / U_BN) * ((K + U_BK - 1) / U_BK) : ((N + T_BN - 1) / T_BN) * ((K + T_BK - 1) / T_BK);
    const int nchunks = big ? (M + U_BMS - 1) / U_BMS : (M + T_BMS - 1) / T_BMS;
    // one workgroup per CU: the launch should be an exact number of 256-workgroup waves (measured: 288 blocks
    // cost 1.10 ms where 768 cost 0.77 ms on the QKV weight gradient) and a multiple of the 8 XCDs
    int s = ((512 / tiles + 7) / 8) * 8;
    for (int w = 1; w <= 4; ++w)
        if ((256 * w) % tiles == 0 && ((256 * w) / tiles) % 8 == 0 && (256 * w) / tiles <= 128) { s = (256 * w) / tiles; break; }
    if (s > 128) s = 128;
    if (s > nchunks) s = nchunks;
    if (s < 1) s = 1;
    return s;
}
size_t mbx_gemm_tn_pipe_ws(int M, int N, int K) {
    const size_t sp = tnp_splits(M, N, K);
    return (sp * N * K + sp * N) * sizeof(float) + 256;
}
int mbx_launch_gemm_tn_pipe(const void* dy, const void* a, float* dw, float* db, int M, int N, int K, void* ws, hipStream_t s) {
    const bool big = tn_use256(N, K);
    const int ntn = big ? (N + U_BN - 1) / U_BN : (N + T_BN - 1) / T_BN, ntk = big ? (K + U_BK - 1) / U_BK : (K + T_BK - 1) / T_BK;
    const int splits = tnp_splits(M, N, K);
    const int nchunks = big ? (M + U_BMS - 1) / U_BMS : (M + T_BMS - 1) / T_BMS;
    const int cps = (nchunks + splits - 1) / splits;
    float* part_w = splits == 1 ? dw : (float*)ws;
    float* part_b = db ? (splits == 1 ? db : (float*)ws + (size_t)splits * N * K) : nullptr;
    if (big) {
        const size_t shm256 = 4 * U_STAGE;
        if (set_lds_attr(gemm_tn_pipe256_kernel, shm256, "gemm_tn_pipe256")) return 1;
        const int ntiles256 = ntn * ntk, groups256 = (splits + 7) / 8;
        hipLaunchKernelGGL(gemm_tn_pipe256_kernel, dim3(8 * groups256 * ntiles256), dim3(512), shm256, s, (const bf16_t*)dy,
                           (const bf16_t*)a, part_w, part_b, M, N, K, ntk, ntiles256, splits, cps);
        MBX_LAUNCH_CHECK("gemm_tn_pipe256");
        if (splits > 1) {
            if (mbx_launch_colsum(part_w, splits, N * K, 0, N * K, dw, s)) return 1;
            if (db && mbx_launch_colsum(part_b, splits, N, 0, N, db, s)) return 1;
        }
        return 0;
    }
    const size_t shm = 3 * T_STAGE;
    static const int dbg = [] { const char* e = getenv("MBX_DBG"); return e ? atoi(e) : 0; }();
    if (set_lds_attr(gemm_tn_pipe_kernel, shm, "gemm_tn_pipe")) return 1;
    const int ntiles = ntn * ntk, groups = (splits + 7) / 8;
    hipLaunchKernelGGL(gemm_tn_pipe_kernel, dim3(8 * groups * ntiles), dim3(512), shm, s, (const bf16_t*)dy, (const bf16_t*)a, part_w,
                       part_b, M, N, K, ntk, ntiles, splits, cps, dbg);
    MBX_LAUNCH_CHECK("gemm_tn_pipe");
    if (splits > 1) {
        if (mbx_launch_colsum(part_w, splits, N * K, 0, N * K, dw, s)) return 1;
        if (db && mbx_launch_colsum(part_b, splits, N, 0, N, db, s)) return 1;
    }
    return 0;
}
