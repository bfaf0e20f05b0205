// Pipelined bf16 MFMA GEMMs for gfx950 (the throughput path; gemm.hip keeps the fp32 parity kernels, mlp_fused.hip the fused MLP).
//
// What ships, and which launches of a training step (64 clips) run on it:
//   gemm_nt_pp256_kernel<EPI, X3>   acc[M,N] = A[M,K] . W[N,K]^T, 256 x 256 tile, 8 waves as 2 (M) x 4 (N) of 128 x 64 (4 x 2 MFMA
//                                   32x32x16 tiles), BK = 32, 4-stage 32 KiB LDS ring filled by LDS-DMA, ONE workgroup per CU,
//                                   PING-PONG schedule (waves 4-7 one phase behind waves 0-3: while one wave of a SIMD reads its
//                                   fragments the other issues its 16 MFMAs).  Epilogues: STORE (qkv, every plain dX GEMM),
//                                   GELU (fc1), TANH (pre_logits), DGELU (+ the row dots of the folded LayerNorm backward).
//                                   X3 = the split-operand fp32-class mode (three bf16 passes over hi / lo planes).
//   gemm_nt_pipe_kernel<EPI>        the same product on a 256 x 128 tile, 8 waves as 4 x 2 of 64 x 64, 3-stage 24 KiB ring, TWO
//                                   workgroups per CU; carries the epilogues that stream fp32: RESID (proj, fc2: + residual),
//                                   LNBWD / LNBWD_T (LayerNorm backward as the epilogue of the dX GEMM; _T: the gradient residual
//                                   stream arrives and leaves as bf16).
//                                   (Round 5: the three epilogues that never had a caller in the default path are gone from the
//                                   source and the ABI -- RESID_LN, the residual GEMM whose last column tile normalises its row block
//                                   (round 3: bit-identical, time-neutral, profiles/r03_resid_ln.txt), RESID_T / STORE_LN, the bf16
//                                   interface of the first raw-operand LayerNorm (round 4; the row-owner kernels of gemm_rows.hip
//                                   read the fp32 rows instead).  Their measurements stay in DESIGN.md; the code is in git history.)
//   gemm_tn_pipe256_kernel<X3>      dW[N,K] = dY[M,N]^T . A[M,K] (contraction over the TOKEN dimension, the slow dimension of both
//                                   operands): tiles staged untransposed, fragments fetched with the gfx950 transpose read
//                                   ds_read_b64_tr_b16 (inline asm: see tr_frag_a), 256 x 256 output tile, token dimension split
//                                   over workgroups, fp32 partial tiles folded by a deterministic column sum; the same ping-pong
//                                   schedule; bias gradient as packed-bf16 dots spread evenly over every wave of every workgroup.
//   gemm_tn_pipe_kernel             the 256 x 128 weight-gradient kernel for N or K < 256 (MotionBERT-Lite's C = 256 shapes use
//                                   the 256 x 256 one as well; this serves the small tails).
// LDS image of every NT stage: rows of 64 B = four 16-byte chunks, chunk c of row r at physical chunk c ^ ((r >> 2) & 3)
// (conflict-free ds_read_b128 fragment reads).  LDS-DMA (global_load_lds_dwordx4) writes lane-linear, so the swizzle is applied to
// the per-lane SOURCE address -- the same involution on both sides.  One raw s_barrier per phase and counted s_waitcnt vmcnt(N)
// (never 0 inside a loop) keep two or three k-tiles in flight across the barriers.  Tile -> workgroup orders are XCD-aware
// (xcd_remap2: the column tiles of a row block run on CUs that share an L2).
// The product build has no switches: constants below that read like knobs are closed A/B experiments (numbers in DESIGN.md and
// profiles/r0*_*.txt); ablation switches (`dbg`), cycle stamps and the round-1 lockstep loop exist in -DMBX_DIAG builds only
// (tools/build_variants.py diag).
#include "mbx_common.h"
#include "gelu_fast.h"
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

#define GLDS16(src, dst) __builtin_amdgcn_global_load_lds((gbl_void_t*)(src), (lds_void_t*)(dst), 16, 0, 0)
#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// 16-byte output store of the GEMM epilogues.  POLICY 1 = non-temporal (`global_store_dwordx4 ... nt`): used by the bf16 store /
// GELU epilogues of the 256x256 kernel, whose output stream (0.5-1.0 GB per launch, read next by another kernel from HBM anyway)
// otherwise competes for the L2 lines the neighbouring tiles share their A / W panels through; measured -8 % on the qkv and
// fc1 launches (0.559 -> 0.516 ms, 0.572 -> 0.525 ms), -1 % on the step; nothing or a loss on the residual / LayerNorm-backward /
// GELU' epilogues, which keep plain stores (profiles/r03_store_policy.txt).
typedef uint32_t epi_u32x4_t __attribute__((ext_vector_type(4)));
template <int POLICY>
__device__ __forceinline__ void epi_store16(void* p, const uint4& v) {
    if (POLICY == 1) {
        const epi_u32x4_t d = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(d, reinterpret_cast<epi_u32x4_t*>(p));
    } else {
        *reinterpret_cast<uint4*>(p) = v;
    }
}
// 16-byte streaming input of an epilogue (residual, dres, xhat, u): POLICY 1 = non-temporal load
typedef float epi_f32x4_t __attribute__((ext_vector_type(4)));
template <int POLICY>
__device__ __forceinline__ float4 epi_load_f4(const float* p) {
    if (POLICY == 1) {
        const epi_f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const epi_f32x4_t*>(p));
        return make_float4(v[0], v[1], v[2], v[3]);
    }
    return *reinterpret_cast<const float4*>(p);
}
template <int POLICY>
__device__ __forceinline__ uint4 epi_load_u4(const bf16_t* p) {
    if (POLICY == 1) {
        const epi_u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const epi_u32x4_t*>(p));
        return make_uint4(v[0], v[1], v[2], v[3]);
    }
    return *reinterpret_cast<const uint4*>(p);
}
// measured (profiles/r03_store_policy.txt, session W): non-temporal loads of the residual -4 % on proj (0.343 -> 0.328 ms), nothing
// on fc2; of dres / xhat in the LayerNorm-backward epilogue +0-4 %; of u in the GELU' epilogue nothing
constexpr int MBX_LD_RES = 1;
constexpr int MBX_LD_LNB = 0;
constexpr int MBX_LD_DG = 0;
constexpr int MBX_ST_PP = 1;   // bf16 store / GELU epilogue of the 256x256 kernel
constexpr int MBX_ST_DG = 0;   // GELU' epilogue
constexpr int MBX_ST_RES = 0;   // residual epilogue (fp32 stream)
constexpr int MBX_ST_LNB = 0;   // LayerNorm-backward epilogue

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
// (Round 3, measured and dropped: BK = 64 for this kernel -- 128-byte row segments, i.e. whole cache lines per LDS-DMA request, half
// the barriers, 3 x 48 KiB stages and therefore ONE workgroup per CU.  Correct; the DMA-only loop gains 16 % (0.436 -> 0.364 ms at
// K = 1536), the kernels nothing: lnb_qkv 0.811 -> 0.812 ms, fc2 0.513 -> 0.550 ms, proj 0.347 -> 0.366 ms, step 130.7 -> 132.3 ms.
// profiles/r03_ntp_ablation.txt)

// rows of 64 B = four 16-byte chunks; chunk c of row r at physical chunk c ^ ((r >> 2) & 3)
__device__ __forceinline__ int sw_off(int row, int chunk) { return row * P_ROWB + ((chunk ^ ((row >> 2) & 3)) << 4); }

// what the LayerNorm-backward epilogue with the bf16 gradient stream needs beside the common arguments
struct NtLnTail {
    const bf16_t* dres_t;   // MBX_EPI_LNBWD_T: the incoming gradient of the residual stream as bf16 (instead of `resid` fp32)
};

// (Round 3, measured and dropped: a ninth PRODUCER wave per workgroup that issues all 24 LDS-DMA instructions of a k-tile, so
// that no compute wave stalls on the vector-memory path in front of its MFMAs.  Correct, and 5-24 % slower on every shape
// (lnb_qkv 0.777 -> 0.822 ms, proj 0.341 -> 0.424 ms): one wave issues the 24 instructions more slowly than eight waves issue
// three each -- DMA-only loop 0.436 -> 0.490 ms -- and at 96 VGPRs the residual epilogue spills.  profiles/r03_ntp_ablation.txt)
template <int EPI>
__global__ __launch_bounds__(512, 4) void gemm_nt_pipe_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                              const float* __restrict__ bias, bf16_t* __restrict__ out_t,
                                                              bf16_t* __restrict__ out2_t, float* __restrict__ out_f,
                                                              const float* __restrict__ resid, const bf16_t* __restrict__ aux,
                                                              int M, int N, int K, int ntn, const float4* __restrict__ rowc,
                                                              const float* __restrict__ extra, const NtLnTail ln
#ifdef MBX_DIAG
                                                              , int dbg, long long* trace
#endif
                                                              ) {
#ifndef MBX_DIAG
    constexpr int dbg = 0;                 // ablation switches and cycle stamps exist in diagnostic builds only
    constexpr long long* trace = nullptr;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 3 stages x 24 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = xcd_remap2(blockIdx.x, gridDim.x);
    const int n0 = (lid % ntn) * P_BN, m0 = (lid / ntn) * P_BM;
    const int nk = K / P_BK;
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
        srcA[i] = A + (size_t)min(((dbg & 8) ? 0 : m0) + row, M - 1) * K + ((lp ^ ((row >> 2) & 3)) << 3);   // dbg 8: every tile reads row block 0 (L2-resident A)
    }
    {
        const int row = wave * 16 + lr;
        srcW = W + (size_t)min(n0 + row, N - 1) * K + ((lp ^ ((row >> 2) & 3)) << 3);
        // dbg 64 (timing probe only, results are wrong): read W as if it were packed k-tile-major, [N/16][K/32][16][32] -- one
        // LDS-DMA instruction = one contiguous KiB (eight whole lines) instead of sixteen half lines
        if (dbg & 64) srcW = W + (size_t)(min(n0 / 16 + wave, N / 16 - 1)) * (K / P_BK) * 512 + lane * 8;
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

#define NT_ISSUE(kt_, stage_)                                                          \
    do {                                                                               \
        const size_t ko_ = (size_t)(kt_) * P_BK;                                       \
        GLDS16(srcA[0] + ko_, dstA + (stage_) * P_STAGE);                              \
        GLDS16(srcA[1] + ko_, dstA + (stage_) * P_STAGE + 1024);                       \
        GLDS16(srcW + ((dbg & 64) ? (size_t)(kt_) * 512 : ko_), dstW + (stage_) * P_STAGE); \
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
    if constexpr (EPI == MBX_EPI_LNBWD || EPI == MBX_EPI_LNBWD_T) {
        // LayerNorm backward as the epilogue of the dX GEMM ("LayerNorm folding", elementwise.hip): acc = d(xhat),
        //   dx = dres [+ extra] + rstd acc - rstd c1 - xhat rstd c2,   rowc[m] = {rstd, rstd c1, rstd c2, -}.
        // Staging: 32 rows x 64 columns fp32 per pass (pitch 272 B, 8.5 KiB of the wave's 9 KiB), walked with eight lanes per row,
        // eight columns per lane: one instruction = 8 rows x 256 B (fp32 in / out) or 8 rows x 128 B (bf16 xhat in, dx_t out).
        constexpr int EP = 64 * 4 + 16;
        const int rr8 = lane >> 3, cc = (lane & 7) * 8;
        const int n = n0 + wn * 64 + cc, nc = min(n, N - 8);
        // the eight passes (two 32-row halves x four) are one pipeline: the inputs of pass P + LNB_D - 1 are requested before pass P
        // is computed (clamped addresses; invalid lanes never store), across the staging of the second half as well
constexpr int MBX_LNB_DEPTH = 2;
        constexpr int LNB_D = MBX_LNB_DEPTH;
        const int mb0 = m0 + wm * 64;
        // LNBWD_T (round 4): the gradient of the residual stream travels BETWEEN the sub-layers of a Block as bf16 -- dres arrives as
        // eight bf16 (one 16-byte load instead of two), and out_f may be NULL (the bf16 output is then the stream AND the next GEMM's
        // operand); fp32 accumulation and arithmetic are unchanged.  tools/gradstream_numerics.py: every gate of the bf16 path holds.
        constexpr bool DRES_T = EPI == MBX_EPI_LNBWD_T;
        float4 dy0[LNB_D], dy1[LNB_D], rc[LNB_D];
        uint4 xh[LNB_D];
#define LNB_LOAD(P_)                                                                            \
        do {                                                                                    \
            const size_t mo_ = (size_t)min(mb0 + (P_) * 8 + rr8, M - 1);                        \
            rc[(P_) % LNB_D] = rowc[mo_];                                                       \
            if (DRES_T) {                                                                       \
                const uint4 t_ = epi_load_u4<MBX_LD_LNB>(ln.dres_t + mo_ * N + nc);             \
                dy0[(P_) % LNB_D] = make_float4(__uint_as_float(t_.x << 16), __uint_as_float(t_.x & 0xffff0000u), __uint_as_float(t_.y << 16), __uint_as_float(t_.y & 0xffff0000u)); \
                dy1[(P_) % LNB_D] = make_float4(__uint_as_float(t_.z << 16), __uint_as_float(t_.z & 0xffff0000u), __uint_as_float(t_.w << 16), __uint_as_float(t_.w & 0xffff0000u)); \
            } else {                                                                            \
                dy0[(P_) % LNB_D] = epi_load_f4<MBX_LD_LNB>(resid + mo_ * N + nc);              \
                dy1[(P_) % LNB_D] = epi_load_f4<MBX_LD_LNB>(resid + mo_ * N + nc + 4);          \
            }                                                                                   \
            xh[(P_) % LNB_D] = epi_load_u4<MBX_LD_LNB>(aux + mo_ * N + nc);                                 \
        } while (0)
#pragma unroll
        for (int P = 0; P < LNB_D - 1; ++P) LNB_LOAD(P);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const int mb = mb0 + tm * 32;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(er + i * EP + (tn * 32 + 8 * q + 4 * g) * 4) =
                        make_float4(acc[tn][tm][4 * q], acc[tn][tm][4 * q + 1], acc[tn][tm][4 * q + 2], acc[tn][tm][4 * q + 3]);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (tm * 4 + p + LNB_D - 1 < 8) LNB_LOAD(tm * 4 + p + LNB_D - 1);
                const int rl = p * 8 + rr8, m = mb + rl, sl = (tm * 4 + p) % LNB_D;
                const float4 a0 = *reinterpret_cast<const float4*>(er + rl * EP + cc * 4);
                const float4 a1 = *reinterpret_cast<const float4*>(er + rl * EP + cc * 4 + 16);
                if (m < M && n < N) {
                    const size_t o = (size_t)m * N + n;
                    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    const float dv[8] = {dy0[sl].x, dy0[sl].y, dy0[sl].z, dy0[sl].w, dy1[sl].x, dy1[sl].y, dy1[sl].z, dy1[sl].w};
                    const uint32_t xw[4] = {xh[sl].x, xh[sl].y, xh[sl].z, xh[sl].w};
                    const float rs = rc[sl].x, k1 = rc[sl].y, k2 = rc[sl].z;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x0 = __uint_as_float(xw[e] << 16), x1 = __uint_as_float(xw[e] & 0xffff0000u);
                        v[2 * e] = dv[2 * e] + fmaf(rs, av[2 * e], -fmaf(x0, k2, k1));
                        v[2 * e + 1] = dv[2 * e + 1] + fmaf(rs, av[2 * e + 1], -fmaf(x1, k2, k1));
                    }
                    if (extra) {
                        const float4 e0 = *reinterpret_cast<const float4*>(extra + o), e1 = *reinterpret_cast<const float4*>(extra + o + 4);
                        v[0] += e0.x; v[1] += e0.y; v[2] += e0.z; v[3] += e0.w;
                        v[4] += e1.x; v[5] += e1.y; v[6] += e1.z; v[7] += e1.w;
                    }
                    if (!DRES_T || out_f) {
                        epi_store16<MBX_ST_LNB>(out_f + o, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
                        epi_store16<MBX_ST_LNB>(out_f + o + 4, make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])));
                    }
                    if (out_t)
                        epi_store16<MBX_ST_LNB>(out_t + o, make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])));
                }
            }
#undef LNB_LOAD
        }
        TSTAMP(1 + nk * 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TSTAMP(2 + nk * 4);
        return;
    }
    const int ec = (lane & 7) * 4, erow0 = lane >> 3;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int n = n0 + wn * 64 + tn * 32 + ec;
        // RESID: all eight residual loads of this half are issued BEFORE the accumulators are staged, so their HBM latency
        // runs under the LDS round trip (clamped addresses, unconditional: out-of-range lanes never store)
        float4 rr[8];
        if (EPI == MBX_EPI_RESID) {
#pragma unroll
            for (int p = 0; p < 8; ++p)
                rr[p] = epi_load_f4<MBX_LD_RES>(resid + (size_t)min(m0 + wm * 64 + p * 8 + erow0, M - 1) * N + min(n, N - 4));
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(er + (tm * 32 + i) * EROW + (8 * q + 4 * g) * 4) =
                    make_float4(acc[tn][tm][4 * q], acc[tn][tm][4 * q + 1], acc[tn][tm][4 * q + 2], acc[tn][tm][4 * q + 3]);
        float bb[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias && n < N) load4<float>(bias + n, bb);
#pragma unroll
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
                    v[0] += rr[p].x; v[1] += rr[p].y; v[2] += rr[p].z; v[3] += rr[p].w;
                    epi_store16<MBX_ST_RES>(out_f + o, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
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
#ifndef MBX_NT_PP_DEFAULT
#define MBX_NT_PP_DEFAULT 1
#endif
#ifndef MBX_NT256_DEFAULT_MASK
#define MBX_NT256_DEFAULT_MASK ((1 << MBX_EPI_STORE) | (1 << MBX_EPI_GELU) | (1 << MBX_EPI_TANH) | (1 << MBX_EPI_DGELU))
#endif

// coalesced epilogue shared by the 256 x 256 kernels.  A wave owns 128 rows x (32 NTN) columns of the tile as
// acc[tn][tm] (tn: 32-column block, tm: 32-row block; in the transposed MFMA orientation lane (i, g) holds
// C[tm*32 + i][tn*32 + 8q + 4g + e] in acc[tn][tm][4q + e]).  Each pass stages one 32-row x 64-column fp32 tile in the
// wave's private LDS area (8.5 KiB) and writes it out as complete 128/256-byte row segments.
template <int EPI, int NTN, typename TO = bf16_t>
__device__ __forceinline__ void nt_epilogue(f32x16_t (&acc)[NTN][4], char* er, const float* __restrict__ bias,
                                            TO* __restrict__ out_t, TO* __restrict__ out2_t, float* __restrict__ out_f,
                                            const float* __restrict__ resid, const TO* __restrict__ aux, int M, int N,
                                            int row_base, int col_base, int lane, bf16_t* __restrict__ pl_hi = nullptr,
                                            bf16_t* __restrict__ pl_lo = nullptr) {
    // pl_hi / pl_lo (fp32-class mode only): the LAST output of the epilogue (STORE: out, GELU: the activation, DGELU: the gradient)
    // leaves as the two bf16 planes of the bf16x3 operand split instead of fp32 -- its only reader is a split-operand GEMM
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
            // RESID: the eight residual loads of this 32-row block are issued BEFORE the accumulators are staged, so their
            // HBM latency runs under the LDS round trip (clamped addresses, unconditional: out-of-range lanes never store)
            float4 rr[8];
            if (EPI == MBX_EPI_RESID) {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int m = min(row_base + tm * 32 + p * 4 + erow0, M - 1);
                    rr[p] = *reinterpret_cast<const float4*>(resid + (size_t)m * N + min(n, N - 4));
                }
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(er + i * EROW + (tn * 32 + 8 * q + 4 * g) * 4) =
                        make_float4(acc[2 * h + tn][tm][4 * q], acc[2 * h + tn][tm][4 * q + 1], acc[2 * h + tn][tm][4 * q + 2],
                                    acc[2 * h + tn][tm][4 * q + 3]);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int rl = p * 4 + erow0, m = row_base + tm * 32 + rl;
                const float4 t4 = *reinterpret_cast<const float4*>(er + rl * EROW + ec * 4);
                if (m < M && n < N) {
                    float v[4] = {t4.x + bb[0], t4.y + bb[1], t4.z + bb[2], t4.w + bb[3]};
                    const size_t o = (size_t)m * N + n;
                    if (EPI == MBX_EPI_STORE) {
                        if (sizeof(TO) == 4 && pl_hi) store4_planes(pl_hi + o, pl_lo + o, v);
                        else store4<TO>(out_t + o, v);
                    } else if (EPI == MBX_EPI_GELU) {
                        if (out_t) store4<TO>(out_t + o, v);   // pre-activation is only needed for backward
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = sizeof(TO) == 4 ? gelu_erf(v[e]) : gelu_fast(v[e]);
                        if (sizeof(TO) == 4 && pl_hi) store4_planes(pl_hi + o, pl_lo + o, v);
                        else store4<TO>(out2_t + o, v);
                    } else if (EPI == MBX_EPI_RESID) {
                        v[0] += rr[p].x; v[1] += rr[p].y; v[2] += rr[p].z; v[3] += rr[p].w;
                        store4<float>(out_f + o, v);
                    } else if (EPI == MBX_EPI_TANH) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
                        store4<float>(out_f + o, v);
                    } else if (EPI == MBX_EPI_DGELU) {
                        float u[4];
                        load4<TO>(aux + o, u);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= sizeof(TO) == 4 ? gelu_erf_grad(u[e]) : gelu_fast_grad(u[e]);
                        if (sizeof(TO) == 4 && pl_hi) store4_planes(pl_hi + o, pl_lo + o, v);
                        else store4<TO>(out_t + o, v);
                    }
                }
            }
        }
    }
}
// bf16-output epilogues without a second input stream (STORE, GELU): bias and GELU are applied in the accumulator
// layout, the tile is converted to bf16 BEFORE it is staged (half the LDS bytes of the fp32 staging above) and walked
// row-major with ONE 16-byte store per lane (8 bf16): one instruction = 8 rows x 128 contiguous bytes.  The store tail of
// such an epilogue is store-ISSUE bound, not bandwidth bound (guide T21): half the store instructions for the same bytes.
// Staging tile: 32 rows x 64 bf16, row pitch 144 B (16-byte aligned rows for ds_read_b128); GELU stages two tiles.
static constexpr int EB_PITCH = 64 * 2 + 16, EB_TILE = 32 * EB_PITCH;   // 4608 B
template <int EPI, int NTN, bool FULL>
__device__ __forceinline__ void nt_epilogue_bf16(f32x16_t (&acc)[NTN][4], char* er, const float* __restrict__ bias,
                                                 bf16_t* __restrict__ out_t, bf16_t* __restrict__ out2_t, int M, int N,
                                                 int row_base, int col_base, int lane) {
    static_assert(EPI == MBX_EPI_STORE || EPI == MBX_EPI_GELU || EPI == MBX_EPI_GELU_D, "bf16 staging: STORE / GELU / GELU_D only");
    const int i = lane & 31, g = lane >> 5;
    const int rr = lane >> 3, cc = (lane & 7) * 8;
    char* er2 = er + EB_TILE;
#pragma unroll
    for (int h = 0; h < NTN / 2; ++h) {
        float bb[2][4][4];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nb = col_base + h * 64 + tn * 32 + 8 * q + 4 * g;
                bb[tn][q][0] = bb[tn][q][1] = bb[tn][q][2] = bb[tn][q][3] = 0.f;
                if (bias && (FULL || nb < N)) load4<float>(bias + nb, bb[tn][q]);
            }
        const int n = col_base + h * 64 + cc;
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = acc[2 * h + tn][tm][4 * q + e] + bb[tn][q][e];
                    const int off = i * EB_PITCH + (tn * 32 + 8 * q + 4 * g) * 2;
                    if (EPI == MBX_EPI_STORE) {
                        *reinterpret_cast<uint2*>(er + off) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                    } else {
                        mbx_f32x2_t g0, g1;
                        if (EPI == MBX_EPI_GELU_D) {       // the derivative instead of the pre-activation (both from one erf and one Gaussian)
                            mbx_f32x2_t d0, d1;
                            gelu_fast_both2(mbx_f32x2_t{v[0], v[1]}, g0, d0);
                            gelu_fast_both2(mbx_f32x2_t{v[2], v[3]}, g1, d1);
                            *reinterpret_cast<uint2*>(er + off) = make_uint2(pack_bf2(d0[0], d0[1]), pack_bf2(d1[0], d1[1]));
                        } else {
                            if (out_t) *reinterpret_cast<uint2*>(er + off) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                            g0 = gelu_fast2(mbx_f32x2_t{v[0], v[1]}); g1 = gelu_fast2(mbx_f32x2_t{v[2], v[3]});
                        }
                        *reinterpret_cast<uint2*>(er2 + off) = make_uint2(pack_bf2(g0[0], g0[1]), pack_bf2(g1[0], g1[1]));
                    }
                }
            uint4 t1[4], t2[4];   // all reads first (unpredicated), then the predicated stores: no wait per store
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                // unconditional (with out_t == nullptr the bytes are stale and never stored): a conditionally assigned array
                // becomes a stack object -- 64 B of scratch per lane, +41 % HBM writes on the fc1 launches (profiles/r02_pmc_bench.txt)
                t1[p] = *reinterpret_cast<const uint4*>(er + (p * 8 + rr) * EB_PITCH + cc * 2);
                if (EPI != MBX_EPI_STORE) t2[p] = *reinterpret_cast<const uint4*>(er2 + (p * 8 + rr) * EB_PITCH + cc * 2);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int m = row_base + tm * 32 + p * 8 + rr;
                if (FULL || (m < M && n < N)) {     // FULL: the wave's whole 128 x 64 NTN/2 block is inside the matrix (no branches)
                    const size_t o = (size_t)m * N + n;
                    if (EPI == MBX_EPI_STORE) {
                        epi_store16<MBX_ST_PP>(out_t + o, t1[p]);
                    } else {
                        if (out_t) epi_store16<MBX_ST_PP>(out_t + o, t1[p]);
                        epi_store16<MBX_ST_PP>(out2_t + o, t2[p]);
                    }
                }
            }
        }
    }
}
// GELU' epilogue (second input stream `aux`): fp32 staging as in nt_epilogue, but each lane owns 8 columns: 16-byte
// aux load, 16-byte store, one instruction = 8 rows x 128 B.
// Round 3: with `st_part` the epilogue also leaves, per row and 64-column block, the two row dots the folded LayerNorm backward
// needs (see "LayerNorm folding" in elementwise.hip): part[n / 64][m] = { sum_n du s[n], sum_n du (u - b'[n]) } over the
// block's columns, du = the bf16-ROUNDED output (what the dX GEMM will read), s and b' rounded to bf16 (packed-bf16 dot
// products; see fill_stat_vec in attention.hip for the error budget).  Eight lanes share a row: three DPP steps.
// MUL: `aux` holds the derivative itself (saved by the MBX_EPI_GELU_D epilogue of fc1's forward): one multiply, no erf, no Gaussian.
template <int NTN, bool MUL = false>
__device__ __forceinline__ void nt_epilogue_dgelu(f32x16_t (&acc)[NTN][4], char* er, bf16_t* __restrict__ out_t,
                                                  const bf16_t* __restrict__ aux, int M, int N, int row_base, int col_base,
                                                  int lane, const float* __restrict__ st_bias = nullptr,
                                                  const float* __restrict__ st_rsum = nullptr, float* __restrict__ st_part = nullptr) {
    const int i = lane & 31, g = lane >> 5;
    constexpr int EROW = 64 * 4 + 16;
    const int rr = lane >> 3, cc = (lane & 7) * 8;
#pragma unroll
    for (int h = 0; h < NTN / 2; ++h) {
        const int n = col_base + h * 64 + cc;
        uint32_t svp[4] = {0u, 0u, 0u, 0u}, nbp[4] = {0u, 0u, 0u, 0u};   // bf16 pairs of rsum and of -b' for the lane's eight columns
        if (st_part) {
            const float4 s0 = *reinterpret_cast<const float4*>(st_rsum + min(n, N - 8)), s1 = *reinterpret_cast<const float4*>(st_rsum + min(n, N - 8) + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(st_bias + min(n, N - 8)), b1 = *reinterpret_cast<const float4*>(st_bias + min(n, N - 8) + 4);
            svp[0] = pack_bf2(s0.x, s0.y); svp[1] = pack_bf2(s0.z, s0.w); svp[2] = pack_bf2(s1.x, s1.y); svp[3] = pack_bf2(s1.z, s1.w);
            nbp[0] = pack_bf2(-b0.x, -b0.y); nbp[1] = pack_bf2(-b0.z, -b0.w); nbp[2] = pack_bf2(-b1.x, -b1.y); nbp[3] = pack_bf2(-b1.z, -b1.w);
        }
        // the aux (pre-activation) stream runs one 32-row block ahead of the staging: its HBM latency is paid under the LDS
        // round trip of the previous block (clamped addresses, unconditional: out-of-range lanes never store)
        const bf16_t* auxc = aux + min(n, N - 8);
        uint4 ua[2][4];
#pragma unroll
        for (int p = 0; p < 4; ++p) ua[0][p] = epi_load_u4<MBX_LD_DG>(auxc + (size_t)min(row_base + p * 8 + rr, M - 1) * N);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            if (tm + 1 < 4) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    ua[(tm + 1) & 1][p] = epi_load_u4<MBX_LD_DG>(auxc + (size_t)min(row_base + (tm + 1) * 32 + p * 8 + rr, M - 1) * N);
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(er + i * EROW + (tn * 32 + 8 * q + 4 * g) * 4) =
                        make_float4(acc[2 * h + tn][tm][4 * q], acc[2 * h + tn][tm][4 * q + 1], acc[2 * h + tn][tm][4 * q + 2],
                                    acc[2 * h + tn][tm][4 * q + 3]);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int rl = p * 8 + rr, m = row_base + tm * 32 + rl;
                const float4 a0 = *reinterpret_cast<const float4*>(er + rl * EROW + cc * 4);
                const float4 a1 = *reinterpret_cast<const float4*>(er + rl * EROW + cc * 4 + 16);
                float q1 = 0.f, q2 = 0.f;
                const bool ok = m < M && n < N;
                if (ok) {
                    const size_t o = (size_t)m * N + n;
                    const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    const uint32_t uw[4] = {ua[tm & 1][p].x, ua[tm & 1][p].y, ua[tm & 1][p].z, ua[tm & 1][p].w};
                    uint32_t r[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u0 = __uint_as_float(uw[e] << 16), u1 = __uint_as_float(uw[e] & 0xffff0000u);
                        const mbx_f32x2_t dv_ = mbx_f32x2_t{v[2 * e], v[2 * e + 1]} * (MUL ? mbx_f32x2_t{u0, u1} : gelu_fast_grad2(mbx_f32x2_t{u0, u1}));
                        r[e] = pack_bf2(dv_[0], dv_[1]);
                        if (st_part) {      // packed-bf16 dots (v_dot2c_f32_bf16): du . rsum and du . (u - b')
                            q1 = dot2_bf16(r[e], svp[e], q1);
                            q2 = dot2_bf16(r[e], uw[e], dot2_bf16(r[e], nbp[e], q2));
                        }
                    }
                    epi_store16<MBX_ST_DG>(out_t + o, make_uint4(r[0], r[1], r[2], r[3]));
                }
                if (st_part) {     // wave-uniform; every lane takes part in the cross-lane steps
                    q1 += dpp_mov<0xB1>(q1, q1); q2 += dpp_mov<0xB1>(q2, q2);      // quad_perm [1,0,3,2]
                    q1 += dpp_mov<0x4E>(q1, q1); q2 += dpp_mov<0x4E>(q2, q2);      // quad_perm [2,3,0,1]
                    q1 += dpp_mov<0x141>(q1, q1); q2 += dpp_mov<0x141>(q2, q2);    // row_half_mirror: the other quad of the 8-lane group
                    if (ok && (lane & 7) == 0)
                        *reinterpret_cast<float2*>(st_part + ((size_t)(n >> 6) * M + m) * 2) = make_float2(q1, q2);   // [block][row]: 8 rows = 64 contiguous bytes
                }
            }
        }
    }
}
// 8-wave layout: wave (wm, wn) = (wave >> 2, wave & 3) owns rows [128 wm, +128) x cols [64 wn, +64)
static constexpr int Q_EPI_WAVE_BYTES = 2 * EB_TILE > 32 * (64 * 4 + 16) ? 2 * EB_TILE : 32 * (64 * 4 + 16);   // 9216 B per wave
template <int EPI, typename TO = bf16_t>
__device__ __forceinline__ void nt256_epilogue(f32x16_t (&acc)[2][4], char* smem, const float* __restrict__ bias,
                                               TO* __restrict__ out_t, TO* __restrict__ out2_t, float* __restrict__ out_f,
                                               const float* __restrict__ resid, const TO* __restrict__ aux, int M, int N, int m0,
                                               int n0, int wave, int lane, const float* __restrict__ st_bias = nullptr,
                                               const float* __restrict__ st_rsum = nullptr, float* __restrict__ st_part = nullptr,
                                               bf16_t* __restrict__ pl_hi = nullptr, bf16_t* __restrict__ pl_lo = nullptr) {
    __builtin_amdgcn_s_barrier();   // every wave is done with the k-loop's LDS stages
    char* er = smem + wave * Q_EPI_WAVE_BYTES;
    const int row_base = m0 + (wave >> 2) * 128, col_base = n0 + (wave & 3) * 64;
    if constexpr (sizeof(TO) == 4) {      // fp32-class mode (bf16x3): every T-typed tensor is fp32
        nt_epilogue<EPI, 2, TO>(acc, er, bias, out_t, out2_t, out_f, resid, aux, M, N, row_base, col_base, lane, pl_hi, pl_lo);
    } else if constexpr (EPI == MBX_EPI_STORE || EPI == MBX_EPI_GELU || EPI == MBX_EPI_GELU_D) {
        if (row_base + 128 <= M && col_base + 64 <= N)
            nt_epilogue_bf16<EPI, 2, true>(acc, er, bias, out_t, out2_t, M, N, row_base, col_base, lane);
        else
            nt_epilogue_bf16<EPI, 2, false>(acc, er, bias, out_t, out2_t, M, N, row_base, col_base, lane);
    }
    else if constexpr (EPI == MBX_EPI_DGELU)
        nt_epilogue_dgelu<2>(acc, er, out_t, aux, M, N, row_base, col_base, lane, st_bias, st_rsum, st_part);
    else if constexpr (EPI == MBX_EPI_MULAUX)
        nt_epilogue_dgelu<2, true>(acc, er, out_t, aux, M, N, row_base, col_base, lane);
    else
        nt_epilogue<EPI, 2, TO>(acc, er, bias, out_t, out2_t, out_f, resid, aux, M, N, row_base, col_base, lane);
}

#ifdef MBX_DIAG   // the lockstep loop of round 1: kept as the A/B baseline of diagnostic builds only (MBX_NT_PP=0)
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

#endif

// ================================================================================================
// gemm_nt_pp256: the 256 x 256 tile / 8-wave layout / 4-stage LDS-DMA ring of gemm_nt_pipe256 with a PING-PONG schedule.
// A 512-thread workgroup puts two waves on every SIMD (waves w and w + 4).  In gemm_nt_pipe256 the two run in lockstep:
// after every barrier both read their fragments from the LDS at the same time (matrix pipe idle), then both queue MFMAs.
// Here a k-tile is two phases, R = {12 ds_read_b128 of the wave's fragments for the whole k-tile} and M = {16 MFMAs from
// those registers, with the 4 LDS-DMA issues of tile kt+3 spread between them}, one s_barrier after each phase, and the
// waves 4-7 ("trailing group") run ONE phase behind waves 0-3: while one wave of a SIMD is in M the other is in R, so the
// matrix pipe of the SIMD always has exactly one wave feeding it.
//   phase 2kt   : leading R(kt)   | trailing M(kt-1)
//   phase 2kt+1 : leading M(kt)   | trailing R(kt)
// Ring safety (stage of tile kt+3 == stage of tile kt-1): the last reader of tile kt-1 is trailing R(kt-1) = phase 2kt-1,
// which ends with lgkmcnt(0) before its barrier; the earliest writer is leading M(kt) = phase 2kt+1.  Landing: tile kt is
// first read in phase 2kt, so every wave confirms its own share (counted vmcnt) before the barrier that ends phase 2kt-1:
// leading waves at the end of M(kt-1), trailing waves at the end of R(kt-1).
// ================================================================================================
// X3 = split-operand fp32-class mode (precision 'bf16x3'): A = A_hi + A_lo, W = W_hi + W_lo (bf16 planes, lo = the bf16
// of the remainder), acc = A_hi W_hi^T + A_hi W_lo^T + A_lo W_hi^T on the same accumulators -- three passes over k as ONE
// stream of 3 K/32 k-tiles through the ring (the A_lo W_lo^T term, 2^-16 relative, is dropped); products of bf16 are exact
// in the fp32 MFMA accumulation, so the result is fp32-class (~1e-6) at a third of the bf16 rate.  T-typed tensors are fp32.
template <int EPI, bool X3 = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp256_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                               const bf16_t* __restrict__ A_lo, const bf16_t* __restrict__ W_lo,
                                                               const float* __restrict__ bias, typename std::conditional<X3, float, bf16_t>::type* __restrict__ out_t,
                                                               typename std::conditional<X3, float, bf16_t>::type* __restrict__ out2_t, float* __restrict__ out_f,
                                                               const float* __restrict__ resid, const typename std::conditional<X3, float, bf16_t>::type* __restrict__ aux,
                                                               int M, int N, int K, int ntn,
                                                               const float* __restrict__ st_bias, const float* __restrict__ st_rsum,
                                                               float* __restrict__ st_part, bf16_t* __restrict__ pl_hi,
                                                               bf16_t* __restrict__ pl_lo
#ifdef MBX_DIAG
                                                               , long long* trace      // cycle stamps: diagnostic builds only
#endif
                                                               ) {
    typedef typename std::conditional<X3, float, bf16_t>::type TO;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 4 stages x 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = xcd_remap2(blockIdx.x, gridDim.x);
    const int n0 = (lid % ntn) * Q_BN, m0 = (lid / ntn) * Q_BM;
    const int wm = wave >> 2, wn = wave & 3;   // wave tile: rows [128 wm, +128), cols [64 wn, +64)
    const bool trailing = wave >= 4;            // wave-uniform (readfirstlane above)
#ifdef MBX_DIAG
    if (trace != nullptr && blockIdx.x == 3000 && (tid == 0 || tid == 256)) trace[(tid == 256 ? 2048 : 0) + 1000] = (long long)__builtin_readcyclecounter();
#endif

    const int lr = lane >> 2, lp = lane & 3;
    const bf16_t* srcA[2];
    const bf16_t* srcW[2];
    const bf16_t* srcAl[2];   // X3: the lo planes (same offsets)
    const bf16_t* srcWl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 32 + i * 16 + lr;
        const int sw = (lp ^ ((row >> 2) & 3)) << 3;
        const size_t oa = (size_t)min(m0 + row, M - 1) * K + sw, ow = (size_t)min(n0 + row, N - 1) * K + sw;
        srcA[i] = A + oa;
        srcW[i] = W + ow;
        srcAl[i] = X3 ? A_lo + oa : nullptr;
        srcWl[i] = X3 ? W_lo + ow : nullptr;
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

    const int nk1 = K / Q_BK;                 // k-tiles of one pass
    const int nk = X3 ? 3 * nk1 : nk1;        // k-tiles of the whole stream: passes (A_hi,W_hi), (A_hi,W_lo), (A_lo,W_hi)
    // stream position kt_ -> (pass, k offset); branch-free selects (v_cndmask on the pointers)
#define PP_ISSUE1(kt_, stage_, j_)                                                                                  \
    do {                                                                                                            \
        const int p1_ = X3 && (kt_) >= nk1, p2_ = X3 && (kt_) >= 2 * nk1;                                           \
        const size_t ko_ = (size_t)((kt_) - (p1_ ? nk1 : 0) - (p2_ ? nk1 : 0)) * Q_BK;                              \
        if ((j_) == 0) GLDS16((p2_ ? srcAl[0] : srcA[0]) + ko_, dstA + (stage_) * Q_STAGE);                         \
        else if ((j_) == 1) GLDS16((p2_ ? srcAl[1] : srcA[1]) + ko_, dstA + (stage_) * Q_STAGE + 1024);             \
        else if ((j_) == 2) GLDS16(((p1_ && !p2_) ? srcWl[0] : srcW[0]) + ko_, dstW + (stage_) * Q_STAGE);          \
        else GLDS16(((p1_ && !p2_) ? srcWl[1] : srcW[1]) + ko_, dstW + (stage_) * Q_STAGE + 1024);                  \
    } while (0)
#define PP_ISSUE(kt_, stage_) do { PP_ISSUE1(kt_, stage_, 0); PP_ISSUE1(kt_, stage_, 1); PP_ISSUE1(kt_, stage_, 2); PP_ISSUE1(kt_, stage_, 3); } while (0)

    PP_ISSUE(0, 0);
    if (nk > 1) PP_ISSUE(1, 1);
    if (nk > 2) PP_ISSUE(2, 2);
    // tile 0 is read in phase 0: everybody confirms its share now
    if (nk > 2) WAIT_VMCNT(8); else if (nk > 1) WAIT_VMCNT(4); else WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    if (trailing) __builtin_amdgcn_s_barrier();   // the trailing group sits out phase 0

    const int i = lane & 31, g = lane >> 5;
    int stage = 0;
#ifdef MBX_DIAG
    // cycle stamps of the first wave of each group of ONE workgroup (tools/pp_trace.py): 4 per k-tile
    const bool tr_on = trace != nullptr && blockIdx.x == 3000 && (tid == 0 || tid == 256);
    long long* const tr = trace + (tid == 256 ? 2048 : 0);
#define PSTAMP(slot_) do { if (tr_on) tr[slot_] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define PSTAMP(slot_) do { } while (0)
#endif
    PSTAMP(0);
    bf16x8_t fw[2][2], fa[2][4];
    // phase R: the wave's fragments of one whole k-tile -> registers
#define PP_READ(stage_)                                                                                              \
    do {                                                                                                             \
        const char* sA_ = smem + (stage_) * Q_STAGE;                                                                 \
        const char* sW_ = sA_ + Q_A_BYTES;                                                                           \
        _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                           \
            _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_)                                                         \
                fw[s_][t_] = *reinterpret_cast<const bf16x8_t*>(sW_ + sw_off(wn * 64 + t_ * 32 + i, 2 * s_ + g));    \
            _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_)                                                         \
                fa[s_][t_] = *reinterpret_cast<const bf16x8_t*>(sA_ + sw_off(wm * 128 + t_ * 32 + i, 2 * s_ + g));   \
        }                                                                                                            \
    } while (0)
    // phase M: 16 MFMAs; with DMA_, the four LDS-DMA pieces of tile kt+3 go between the groups of four.  The phase must
    // stay BRANCH-FREE: one wave alone feeds the SIMD's matrix pipe here, and every taken branch between two MFMAs
    // leaves the pipe idle while the instruction buffer refills (measured: 16 MFMAs took 908 cycles instead of 530
    // with a few wave-uniform branches between the groups).
#define PP_MMA(DMA_, kt3_, st3_)                                                                                     \
    do {                                                                                                             \
        _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                             \
            _Pragma("unroll") for (int tn_ = 0; tn_ < 2; ++tn_) {                                                    \
                _Pragma("unroll") for (int tm_ = 0; tm_ < 4; ++tm_)                                                  \
                    acc[tn_][tm_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[s_][tn_], fa[s_][tm_], acc[tn_][tm_], 0, 0, 0); \
                if (DMA_) PP_ISSUE1(kt3_, st3_, 2 * s_ + tn_);                                                       \
            }                                                                                                        \
    } while (0)
#define PP_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

    int kt = 0;
    // steady state: tiles kt+1, kt+2 are in flight at the end of R(kt) -> vmcnt(4) confirms this wave's share of tile kt+1
    // (first read in phase 2kt+2; both groups confirm it by the end of their R(kt), i.e. by phase 2kt+1 at the latest)
    for (; kt + 3 < nk; ++kt) {
        PP_READ(stage);
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        PSTAMP(1 + 4 * kt);
        PP_BARRIER();
        PSTAMP(2 + 4 * kt);
        PP_MMA(true, kt + 3, (stage + 3) & 3);
        PSTAMP(3 + 4 * kt);
        PP_BARRIER();
        PSTAMP(4 + 4 * kt);
        stage = (stage + 1) & 3;
    }
    // the last three k-tiles: nothing left to issue
    for (; kt < nk; ++kt) {
        PP_READ(stage);
        if (nk - 1 - kt >= 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        PSTAMP(1 + 4 * kt);
        PP_BARRIER();
        PSTAMP(2 + 4 * kt);
        PP_MMA(false, 0, 0);
        PSTAMP(3 + 4 * kt);
        PP_BARRIER();
        PSTAMP(4 + 4 * kt);
        stage = (stage + 1) & 3;
    }
#undef PP_BARRIER
#undef PP_MMA
#undef PP_READ
#undef PP_ISSUE
#undef PP_ISSUE1
#undef PSTAMP
    if (!trailing) __builtin_amdgcn_s_barrier();   // pairs with the trailing group's last phase

    nt256_epilogue<EPI, TO>(acc, smem, bias, out_t, out2_t, out_f, resid, aux, M, N, m0, n0, wave, lane, st_bias, st_rsum, st_part, pl_hi, pl_lo);
#ifdef MBX_DIAG
    if (trace != nullptr && blockIdx.x == 3000 && (tid == 0 || tid == 256)) {
        long long* const tr2 = trace + (tid == 256 ? 2048 : 0);
        tr2[1001] = (long long)__builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tr2[1002] = (long long)__builtin_readcyclecounter();
    }
#endif
}

template <typename K>
static int set_lds_attr(K kernel, size_t bytes, const char* who) {
    return mbx_set_dyn_lds(reinterpret_cast<const void*>(kernel), bytes, who);
}

static int launch_nt256(const void* a, const void* w, const float* bias, int epi, void* out_t, void* out2_t, float* out_f,
                        const float* resid, const void* aux, int M, int N, int K, hipStream_t s, const float* st_bias = nullptr,
                        const float* st_rsum = nullptr, float* st_part = nullptr) {
    const int ntn = (N + Q_BN - 1) / Q_BN, ntm = (M + Q_BM - 1) / Q_BM;
    dim3 grid((unsigned)ntn * ntm), block(512);
    const size_t shm = Q_NSTAGE * Q_STAGE;
#ifdef MBX_DIAG
    static const int pp = mbx_env_int("MBX_NT_PP", MBX_NT_PP_DEFAULT);
#define MBX_Q_LOCKSTEP(E)                                                                                             \
        if (!pp) {                                                                                                    \
            if (st_part != nullptr)                                                                                   \
                return mbx_set_error("gemm_nt: the lockstep loop (MBX_NT_PP=0, diagnostic builds) has no row-dot epilogue"); \
            if (set_lds_attr(gemm_nt_pipe256_kernel<E>, shm, "gemm_nt_pipe256")) return 1;                            \
            hipLaunchKernelGGL((gemm_nt_pipe256_kernel<E>), grid, block, shm, s, (const bf16_t*)a, (const bf16_t*)w, bias, \
                               (bf16_t*)out_t, (bf16_t*)out2_t, out_f, resid, (const bf16_t*)aux, M, N, K, ntn);      \
            break;                                                                                                    \
        }
    static long long* const pptrace = [] { const char* e = getenv("MBX_TRACE_BUF"); return e ? (long long*)strtoull(e, nullptr, 0) : (long long*)nullptr; }();
#define MBX_Q_TRACE_ARG , pptrace
#else
#define MBX_Q_LOCKSTEP(E)
#define MBX_Q_TRACE_ARG
#endif
#define MBX_Q_CASE(E)                                                                                                 \
    case E:                                                                                                           \
        MBX_Q_LOCKSTEP(E)                                                                                             \
        if (set_lds_attr(gemm_nt_pp256_kernel<E>, shm, "gemm_nt_pp256")) return 1;                                    \
        hipLaunchKernelGGL((gemm_nt_pp256_kernel<E>), grid, block, shm, s, (const bf16_t*)a, (const bf16_t*)w,        \
                           (const bf16_t*)nullptr, (const bf16_t*)nullptr, bias,                                      \
                           (bf16_t*)out_t, (bf16_t*)out2_t, out_f, resid, (const bf16_t*)aux, M, N, K, ntn,          \
                           st_bias, st_rsum, st_part, (bf16_t*)nullptr, (bf16_t*)nullptr MBX_Q_TRACE_ARG); \
        break;
    switch (epi) {
        MBX_Q_CASE(MBX_EPI_STORE)
        MBX_Q_CASE(MBX_EPI_GELU)
        MBX_Q_CASE(MBX_EPI_RESID)
        MBX_Q_CASE(MBX_EPI_TANH)
        MBX_Q_CASE(MBX_EPI_DGELU)
        MBX_Q_CASE(MBX_EPI_GELU_D)
        MBX_Q_CASE(MBX_EPI_MULAUX)
        default: return mbx_set_error("gemm_nt: unknown epilogue %d", epi);
    }
#undef MBX_Q_CASE
#undef MBX_Q_LOCKSTEP
#undef MBX_Q_TRACE_ARG
    MBX_LAUNCH_CHECK("gemm_nt_pp256");
    return 0;
}

// fp32-class split-operand GEMM (precision 'bf16x3'): always the 256 x 256 ping-pong kernel; T-typed tensors are fp32
int mbx_launch_gemm_nt_x3(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias, int epi,
                          float* out_t, float* out2_t, float* out_f, const float* resid, const float* aux, int M, int N, int K,
                          hipStream_t s, void* pl_hi, void* pl_lo) {
    const int ntn = (N + Q_BN - 1) / Q_BN, ntm = (M + Q_BM - 1) / Q_BM;
    dim3 grid((unsigned)ntn * ntm), block(512);
    const size_t shm = Q_NSTAGE * Q_STAGE;
#ifdef MBX_DIAG
#define MBX_X3_TRACE_ARG , (long long*)nullptr
#else
#define MBX_X3_TRACE_ARG
#endif
#define MBX_X3_CASE(E)                                                                                                \
    case E:                                                                                                           \
        if (set_lds_attr(gemm_nt_pp256_kernel<E, true>, shm, "gemm_nt_x3")) return 1;                                 \
        hipLaunchKernelGGL((gemm_nt_pp256_kernel<E, true>), grid, block, shm, s, (const bf16_t*)a_hi, (const bf16_t*)w_hi, \
                           (const bf16_t*)a_lo, (const bf16_t*)w_lo, bias, out_t, out2_t, out_f, resid, aux, M, N, K, ntn, \
                           (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (bf16_t*)pl_hi, (bf16_t*)pl_lo \
                           MBX_X3_TRACE_ARG);                                                                        \
        break;
    switch (epi) {
        MBX_X3_CASE(MBX_EPI_STORE)
        MBX_X3_CASE(MBX_EPI_GELU)
        MBX_X3_CASE(MBX_EPI_RESID)
        MBX_X3_CASE(MBX_EPI_TANH)
        MBX_X3_CASE(MBX_EPI_DGELU)
        default: return mbx_set_error("gemm_nt_x3: unknown epilogue %d", epi);
    }
#undef MBX_X3_CASE
    MBX_LAUNCH_CHECK("gemm_nt_x3");
    return 0;
}

int mbx_launch_gemm_nt_pipe(const void* a, const void* w, const float* bias, int epi, void* out_t, void* out2_t, float* out_f,
                            const float* resid, const void* aux, int M, int N, int K, hipStream_t s) {
    // Which epilogues run on the 256 x 256 kernel (one workgroup per CU) rather than on the 256 x 128 kernel (two per CU):
    // bit e = epilogue e.  Measured (tools/gemm_bench.py, M = 264k, profiles/): see the table in DESIGN.md.
    static const int mask256 = mbx_env_int("MBX_NT256_MASK", MBX_NT256_DEFAULT_MASK);
    if (((mask256 >> epi) & 1) && N >= 256) return launch_nt256(a, w, bias, epi, out_t, out2_t, out_f, resid, aux, M, N, K, s);
    const int ntn = (N + P_BN - 1) / P_BN, ntm = (M + P_BM - 1) / P_BM;
    dim3 grid((unsigned)ntn * ntm), block(512);
    const size_t shm = P_NSTAGE * P_STAGE + (size_t)mbx_env_int("MBX_NTP_LDS_PAD", 0) * 1024;   // diagnostics: padding -> one workgroup per CU
#ifdef MBX_DIAG
    static const int dbg = mbx_env_int("MBX_DBG", 0);
    static long long* const trace = [] { const char* e = getenv("MBX_TRACE_BUF"); return e ? (long long*)strtoull(e, nullptr, 0) : (long long*)nullptr; }();
#define MBX_NTP_DIAG_ARGS , dbg, trace
#else
#define MBX_NTP_DIAG_ARGS
#endif
#define MBX_NTP_CASE(E)                                                                                               \
    case E:                                                                                                           \
        if (set_lds_attr(gemm_nt_pipe_kernel<E>, shm, "gemm_nt_pipe")) return 1;                                      \
        hipLaunchKernelGGL((gemm_nt_pipe_kernel<E>), grid, block, shm, s, (const bf16_t*)a, (const bf16_t*)w, bias,   \
                           (bf16_t*)out_t, (bf16_t*)out2_t, out_f, resid, (const bf16_t*)aux, M, N, K, ntn,           \
                           (const float4*)nullptr, (const float*)nullptr, NtLnTail{} MBX_NTP_DIAG_ARGS);              \
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

// ---- round 3: the two GEMM entries of the folded LayerNorm backward (see "LayerNorm folding" in elementwise.hip) --------------
// du = (dy . W2) * gelu'(u)  (the DGELU epilogue) + the row dots part[N/64][M][2] of du with rsum and (u - bias_f)
extern "C" int mbx_gemm_nt_dgelu_stats(const void* a, const void* w, void* out_t, const void* aux_t, const float* bias_f,
                                       const float* rsum, float* part, int M, int N, int K, void* stream) {
    MBX_CHECK_ARG(a && w && out_t && aux_t && bias_f && rsum && part, "gemm_nt_dgelu_stats: null pointer");
    MBX_CHECK_ARG(M > 0 && N > 0 && N % 64 == 0 && K > 0 && K % 64 == 0, "gemm_nt_dgelu_stats: bad shape M=%d N=%d K=%d (N %% 64, K %% 64)", M, N, K);
    return launch_nt256(a, w, nullptr, MBX_EPI_DGELU, out_t, nullptr, nullptr, nullptr, aux_t, M, N, K, (hipStream_t)stream, bias_f, rsum, part);
}

// ---- round 5: the activation's derivative saved instead of the pre-activation ---------------------------
// fc1 + GELU: out_g = gelu(a . w^T + bias), out_d = gelu'(a . w^T + bias) taken from the fp32 accumulator; the backward epilogue is then
// ONE multiply (mbx_gemm_nt_mul).  Possible since the row-owner LayerNorm-backward GEMM takes its row means itself: the GELU' epilogue no
// longer has to produce the dot of du with the pre-activation.
extern "C" int mbx_gemm_nt_gelu_d(const void* a, const void* w, const float* bias, void* out_d, void* out_g, int M, int N, int K, void* stream) {
    MBX_CHECK_ARG(a && w && out_d && out_g, "gemm_nt_gelu_d: null pointer");
    MBX_CHECK_ARG(M > 0 && N >= 256 && N % 8 == 0 && K >= 64 && K % 64 == 0, "gemm_nt_gelu_d: bad shape M=%d N=%d (>= 256, %% 8) K=%d (%% 64)", M, N, K);
    return launch_nt256(a, w, bias, MBX_EPI_GELU_D, out_d, out_g, nullptr, nullptr, nullptr, M, N, K, (hipStream_t)stream);
}
// out = (a . w^T) * aux  (aux = the saved derivative)
extern "C" int mbx_gemm_nt_mul(const void* a, const void* w, const void* aux, void* out, int M, int N, int K, void* stream) {
    MBX_CHECK_ARG(a && w && aux && out, "gemm_nt_mul: null pointer");
    MBX_CHECK_ARG(M > 0 && N >= 256 && N % 8 == 0 && K >= 64 && K % 64 == 0, "gemm_nt_mul: bad shape M=%d N=%d (>= 256, %% 8) K=%d (%% 64)", M, N, K);
    return launch_nt256(a, w, nullptr, MBX_EPI_MULAUX, out, nullptr, nullptr, nullptr, aux, M, N, K, (hipStream_t)stream);
}
// dx = dres [+ extra] + rstd (dy . Wt' - c1 - xhat c2)  with rowc[m] = {rstd, rstd c1, rstd c2, -};  dx_t = bf16 copy (or NULL)
extern "C" int mbx_gemm_nt_lnbwd(const void* a, const void* w, const void* xhat, const float* rowc, const float* dres,
                                 const float* extra, float* dx, void* dx_t, int M, int N, int K, void* stream) {
    MBX_CHECK_ARG(a && w && xhat && rowc && dres && dx, "gemm_nt_lnbwd: null pointer");
    MBX_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0 && K > 0 && K % 64 == 0, "gemm_nt_lnbwd: bad shape M=%d N=%d K=%d (N %% 8, K %% 64)", M, N, K);
    MBX_CHECK_ARG((reinterpret_cast<uintptr_t>(rowc) & 15) == 0, "gemm_nt_lnbwd: rowc must be 16-byte aligned");
    const int ntn = (N + P_BN - 1) / P_BN, ntm = (M + P_BM - 1) / P_BM;
    const size_t shm = P_NSTAGE * P_STAGE + (size_t)mbx_env_int("MBX_NTP_LDS_PAD", 0) * 1024;
    hipStream_t s = (hipStream_t)stream;
#ifdef MBX_DIAG
    static const int dbg = mbx_env_int("MBX_DBG", 0);
    static long long* const trace = [] { const char* e = getenv("MBX_TRACE_BUF"); return e ? (long long*)strtoull(e, nullptr, 0) : (long long*)nullptr; }();
#endif
    if (set_lds_attr(gemm_nt_pipe_kernel<MBX_EPI_LNBWD>, shm, "gemm_nt_lnbwd")) return 1;
    hipLaunchKernelGGL((gemm_nt_pipe_kernel<MBX_EPI_LNBWD>), dim3((unsigned)ntn * ntm), dim3(512), shm, s, (const bf16_t*)a, (const bf16_t*)w,
                       (const float*)nullptr, (bf16_t*)dx_t, (bf16_t*)nullptr, dx, dres, (const bf16_t*)xhat, M, N, K, ntn,
                       reinterpret_cast<const float4*>(rowc), extra, NtLnTail{}
#ifdef MBX_DIAG
                       , dbg, trace
#endif
                       );
    MBX_LAUNCH_CHECK("gemm_nt_lnbwd");
    return 0;
}

// mbx_gemm_nt_lnbwd with the gradient residual stream in bf16: dres_t bf16 [M,N]; dx f32 or NULL; dx_t bf16 or NULL (at least one)
extern "C" int mbx_gemm_nt_lnbwd_t(const void* a, const void* w, const void* xhat, const float* rowc, const void* dres_t,
                                   const float* extra, float* dx, void* dx_t, int M, int N, int K, void* stream) {
    MBX_CHECK_ARG(a && w && xhat && rowc && dres_t && (dx || dx_t), "gemm_nt_lnbwd_t: null pointer");
    MBX_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0 && K > 0 && K % 64 == 0, "gemm_nt_lnbwd_t: bad shape M=%d N=%d K=%d (N %% 8, K %% 64)", M, N, K);
    MBX_CHECK_ARG((reinterpret_cast<uintptr_t>(rowc) & 15) == 0, "gemm_nt_lnbwd_t: rowc must be 16-byte aligned");
    const int ntn = (N + P_BN - 1) / P_BN, ntm = (M + P_BM - 1) / P_BM;
    const size_t shm = P_NSTAGE * P_STAGE;
    if (set_lds_attr(gemm_nt_pipe_kernel<MBX_EPI_LNBWD_T>, shm, "gemm_nt_lnbwd_t")) return 1;
    NtLnTail ln{};
    ln.dres_t = (const bf16_t*)dres_t;
    hipLaunchKernelGGL((gemm_nt_pipe_kernel<MBX_EPI_LNBWD_T>), dim3((unsigned)ntn * ntm), dim3(512), shm, (hipStream_t)stream, (const bf16_t*)a,
                       (const bf16_t*)w, (const float*)nullptr, (bf16_t*)dx_t, (bf16_t*)nullptr, dx, (const float*)nullptr, (const bf16_t*)xhat,
                       M, N, K, ntn, reinterpret_cast<const float4*>(rowc), extra, ln
#ifdef MBX_DIAG
                       , 0, (long long*)nullptr
#endif
                       );
    MBX_LAUNCH_CHECK("gemm_nt_lnbwd_t");
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

// The same fragment through INLINE ASM.  Why: the compiler cannot see that the explicit `s_waitcnt vmcnt(N)` + s_barrier
// pairs of the ring protocol order the LDS-DMA writes before the fragment reads, and for the transpose-read builtin (unlike
// a plain LDS load) its waitcnt pass puts `s_waitcnt vmcnt(0)` in front of the first read of every chunk -- i.e. it drains
// the whole LDS-DMA ring once per chunk (found in round 3 by reading the ISA of the shipped gemm_tn_pipe256 loop: the
// "three chunks in flight" were one).  An asm read is invisible to that pass; in exchange the kernel owns the lgkmcnt
// bookkeeping: results may only be used behind an explicit s_waitcnt, and TR_TIE makes that a data dependence.
template <int ROWB>
__device__ __forceinline__ bf16x8_t tr_frag_a(const char* tile, int tok0, int col0, int lane) {
    const int g = lane >> 5, r16 = lane & 15, nb = col0 + 16 * ((lane >> 4) & 1) + 4 * (r16 & 3);
    const int t = tok0 + 8 * g + (r16 >> 2);
    const uint32_t a = (uint32_t)(uintptr_t)(lds_void_t*)(tile + tr_off<ROWB>(t, nb));   // rows t and t + 4 share the swizzle
    union { v4s_t h[2]; bf16x8_t v; } f;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.h[0]) : "v"(a) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.h[1]) : "v"(a), "n"(4 * ROWB) : "memory");
    return f.v;
}
// `s_waitcnt <what>` as a data dependence of the six fragments of one register set (see tr_frag_a)
#define TR_WAIT6(what_, f0, f1, f2, f3, f4, f5) \
    asm volatile("s_waitcnt " what_ : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : : "memory")

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
// A/B knobs of the weight-gradient GEMM's operand traffic (tools/build_variants.py; the product build uses the defaults):
//   MBX_TN_AHEAD  chunks in flight behind the one being multiplied (3; 2 = a smaller L2 footprint per workgroup)
//   MBX_TN_ORDER  0: the k tiles of an n panel are consecutive workgroups; 1: the n tiles of a k panel
//   MBX_TN_AUX    cache-policy bits of the LDS-DMA loads (0 default, 2 = nt)
constexpr int MBX_TN_AHEAD = 3;
constexpr int MBX_TN_ORDER = 0;
constexpr int MBX_TN_AUX = 0;
#define GLDS16_TN(src, dst) __builtin_amdgcn_global_load_lds((gbl_void_t*)(src), (lds_void_t*)(dst), 16, 0, MBX_TN_AUX)
static constexpr int U_TILE = U_BMS * 512, U_STAGE = 2 * U_TILE;   // 16 KiB per operand tile, 32 KiB per stage

// X3 (precision 'bf16x3'): dW = dY_hi^T A_hi + dY_hi^T A_lo + dY_lo^T A_hi, the three passes laid end to end as ONE token
// stream of 3 * ceil(M / 32) chunks (pass p: tokens of (dY_hi, A_hi), (dY_hi, A_lo), (dY_lo, A_hi)); db sums passes 0 and 2.
#ifdef MBX_TN_TRACE
__device__ long long* g_tn_trace;
#endif
template <bool X3>
__global__ __launch_bounds__(512, 2) void gemm_tn_pipe256_kernel(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ A,
                                                                 const bf16_t* __restrict__ dY_lo, const bf16_t* __restrict__ A_lo,
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
    const int nchunks1 = (M + U_BMS - 1) / U_BMS;          // chunks of one pass
    const int nchunks = X3 ? 3 * nchunks1 : nchunks1;
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
        const int p1_ = X3 && (chunk_) >= nchunks1, p2_ = X3 && (chunk_) >= 2 * nchunks1;            \
        const int mb_ = ((chunk_) - (p1_ ? nchunks1 : 0) - (p2_ ? nchunks1 : 0)) * U_BMS;            \
        const bf16_t* ys_ = p2_ ? dY_lo : dY;                                                        \
        const bf16_t* as_ = (p1_ && !p2_) ? A_lo : A;                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                           \
            const size_t r_ = (size_t)min(mb_ + rowv[i_], M - 1);                                    \
            GLDS16_TN(ys_ + r_ * N + ycol[i_], dstY + (stage_) * U_STAGE + i_ * 1024);               \
            GLDS16_TN(as_ + r_ * K + acol[i_], dstA + (stage_) * U_STAGE + i_ * 1024);               \
        }                                                                                            \
    } while (0)

    f32x16_t acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // bias gradient = column sums of dY (a lane of a transposed fragment holds 8 tokens of ONE column).  Round 3 finding: in
    // round 2 wave column 0 of the k-tile-0 workgroups summed all four fragments of its wave row on the VALU (~130 operations in
    // the read phase) -- and the kernel WITHOUT a bias gradient measured 18-22 % faster (dW_qkv 0.634 -> 0.518 ms, -DMBX_TN_NODB):
    // what costs is not the work but its imbalance.  The slow waves hold their workgroup back at both barriers of every chunk, and
    // the slow workgroup falls behind the k tiles that stream the same dY panel, so the panel is fetched from HBM again instead
    // of found in L2.  Now the work is EVEN: the ntk k-tile workgroups of an n panel and the four waves of a wave row hold the
    // same dY fragments, so wave (wr, wc) of k tile kt sums fragment wc of ONE 16-token step (ntk = 2: step kt; ntk = 4: half of the
    // dwords of step kt & 1; ntk = 1: both steps) as packed-bf16 dots with ones -- 4 (8) VALU operations per chunk in every wave of
    // every workgroup -- into its own partial slot; the slots are folded with the token splits.  (Also measured: the same dots in
    // the MFMA phase 0.602 ms -- v_dot2c beside MFMAs is expensive --, two extra ones-MFMAs per wave 0.640 ms.)
    const int nslot = part_b == nullptr ? 0 : (ntk == 2 || ntk == 4) ? ntk : 1;       // partial slots per token split
    const int kt_ = k0 / U_BK;
    const bool want_db = nslot > 0 && (nslot > 1 || kt_ == 0);
    const int db_s = nslot == 1 ? 0 : (kt_ & 1);                                       // this workgroup's 16-token step (nslot = 1: both)
    uint32_t one_e[4];                                                                 // bf16 {1, 1} on this workgroup's dwords, 0 elsewhere
#pragma unroll
    for (int e = 0; e < 4; ++e) one_e[e] = (nslot == 4 && (e >> 1) != (kt_ >> 1)) ? 0u : 0x3f803f80u;
    float bsum = 0.f;

    // Ping-pong schedule (see gemm_nt_pp256): waves 4-7 run one phase behind waves 0-3; phase R = the 12 transposed
    // fragments of a 32-token chunk -> registers (24 ds_read_b64_tr_b16), phase M = its 16 (+4 for the bias gradient) MFMAs
    // with the LDS-DMA of chunk c+3 issued first; one barrier after each phase; the M phase is branch-free.
    const bool trailing = wave >= 4;
    if (nc > 0) U_ISSUE(c_beg, 0);
    if (nc > 1) U_ISSUE(c_beg + 1, 1);
    if (nc > 2) U_ISSUE(c_beg + 2, 2);
    if (nc > 2) WAIT_VMCNT(8); else if (nc > 1) WAIT_VMCNT(4); else WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    if (trailing) __builtin_amdgcn_s_barrier();
    bf16x8_t fy[2][4], fa[2][2];
#define U_READ(stage_, vc_)                                                                                          \
    do {                                                                                                             \
        const char* sY_ = smem + (stage_) * U_STAGE;                                                                 \
        const char* sA_ = sY_ + U_TILE;                                                                              \
        _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                           \
            _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) fy[s_][t_] = tr_frag_a<512>(sY_, 16 * s_, wr * 128 + t_ * 32, lane); \
            _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) fa[s_][t_] = tr_frag_a<512>(sA_, 16 * s_, wc * 64 + t_ * 32, lane);  \
        }                                                                                                            \
    } while (0)
    // behind the explicit lgkmcnt(0): zero the tokens past M (X3: per pass)
#define U_FIX(vc_)                                                                                                   \
    do {                                                                                                             \
        const int p1_ = X3 && (vc_) >= nchunks1, p2_ = X3 && (vc_) >= 2 * nchunks1;                                  \
        const int valid_ = M - ((vc_) - (p1_ ? nchunks1 : 0) - (p2_ ? nchunks1 : 0)) * U_BMS;                        \
        if (valid_ < U_BMS) {   /* last chunk of a pass: rows were clamped to M-1 (duplicates) -> zero their contribution */ \
            _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                       \
                const int tb_ = 16 * s_ + 8 * (lane >> 5);                                                           \
                union { bf16x8_t v; uint16_t h[8]; } z_;                                                             \
                _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) {                                                   \
                    z_.v = fy[s_][t_];                                                                               \
                    _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) if (tb_ + e_ >= valid_) z_.h[e_] = 0;           \
                    fy[s_][t_] = z_.v;                                                                               \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)
#define U_MMA()                                                                                                      \
    do {                                                                                                             \
        _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                             \
            _Pragma("unroll") for (int tr_ = 0; tr_ < 4; ++tr_)                                                      \
                _Pragma("unroll") for (int tc_ = 0; tc_ < 2; ++tc_)                                                  \
                    acc[tr_][tc_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[s_][tr_], fa[s_][tc_], acc[tr_][tc_], 0, 0, 0); \
    } while (0)
#define U_DB1(s_, t_)                                                                                                \
    do {                                                                                                             \
        union { bf16x8_t v; uint32_t u[4]; } z_;                                                                     \
        z_.v = fy[s_][t_];                                                                                           \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) bsum = dot2_bf16(z_.u[e_], one_e[e_], bsum);                \
    } while (0)
    /* this wave's share of the bias gradient: one wave-uniform jump per chunk -- fragment wc of step db_s (and of the other step */
    /* too when there is one slot); X3: pass 1 streams dY_hi a second time and is skipped                                        */
#define U_DB_SHARE(vc_)                                                                                              \
    do {                                                                                                             \
        if (want_db && !(X3 && (vc_) >= nchunks1 && (vc_) < 2 * nchunks1)) {                                         \
            switch (db_s * 4 + wc) {                                                                                 \
                case 0: U_DB1(0, 0); if (nslot == 1) U_DB1(1, 0); break;                                             \
                case 1: U_DB1(0, 1); if (nslot == 1) U_DB1(1, 1); break;                                             \
                case 2: U_DB1(0, 2); if (nslot == 1) U_DB1(1, 2); break;                                             \
                case 3: U_DB1(0, 3); if (nslot == 1) U_DB1(1, 3); break;                                             \
                case 4: U_DB1(1, 0); break;                                                                          \
                case 5: U_DB1(1, 1); break;                                                                          \
                case 6: U_DB1(1, 2); break;                                                                          \
                default: U_DB1(1, 3); break;                                                                         \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)
#define U_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
    int stage = 0;
#ifdef MBX_TN_TRACE     // diagnostic builds (tools/tn_trace.py): cycles of wave 0 / wave 4 of every workgroup in the four parts of a chunk, summed
    long long tt[4] = {0, 0, 0, 0}, t0 = (long long)__builtin_readcyclecounter();
#define TN_TS(k_) do { const long long t1_ = (long long)__builtin_readcyclecounter(); tt[k_] += t1_ - t0; t0 = t1_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TN_TS(k_) do { } while (0)
#endif
    for (int c = 0; c < nc; ++c) {
        const int vc = c_beg + c;
        U_READ(stage, vc);
        // own share of chunk c+1 landed (c+2 may fly); the fragments have arrived (asm reads: nothing may touch them earlier).
        // (Round 6: the vmcnt wait stands alone, ONE statement carries the data dependence of the first six fragments.  With the two
        // forms "vmcnt(4) lgkmcnt(0)" / "vmcnt(0) lgkmcnt(0)" in an if / else the compiler gave the statement's tied operands other
        // registers than the reads had written and copied the six fragments over -- 12 v_mov_b64 per chunk at the end of the read
        // phase, the phase that paces the kernel; in the tail branch it placed the copies in FRONT of the wait, i.e. it copied registers
        // with their reads in flight, harmless only because issuing the other twelve reads takes longer than the first twelve to land.)
        if (nc - 1 - c >= 2) WAIT_VMCNT(4); else WAIT_VMCNT(0);
        TR_WAIT6("lgkmcnt(0)", fy[0][0], fy[0][1], fy[0][2], fy[0][3], fa[0][0], fa[0][1]);
        TR_WAIT6("lgkmcnt(0)", fy[1][0], fy[1][1], fy[1][2], fy[1][3], fa[1][0], fa[1][1]);
        U_FIX(vc);
        U_DB_SHARE(vc);      // at the end of the read phase (behind the MFMAs of the other phase it measured the same: 0.566 vs 0.563 ms)
        TN_TS(0);
        U_BARRIER();
        TN_TS(1);
        if (c + MBX_TN_AHEAD < nc) U_ISSUE(vc + MBX_TN_AHEAD, (stage + MBX_TN_AHEAD) & 3);
        U_MMA();
        TN_TS(2);
        U_BARRIER();
        TN_TS(3);
        stage = (stage + 1) & 3;
    }
#ifdef MBX_TN_TRACE
    if (g_tn_trace != nullptr && (tid == 0 || tid == 256)) {
        long long* const tr = g_tn_trace + ((size_t)blockIdx.x * 2 + (tid == 256)) * 5;
        tr[0] = tt[0]; tr[1] = tt[1]; tr[2] = tt[2]; tr[3] = tt[3]; tr[4] = nc;
    }
#endif
#undef TN_TS
    if (!trailing) __builtin_amdgcn_s_barrier();
#undef U_BARRIER
#undef U_DB_SHARE
#undef U_DB1
#undef U_MMA
#undef U_FIX
#undef U_READ
#undef U_ISSUE
    const int i = lane & 31, g = lane >> 5;
    if (want_db) {     // lanes l and l + 32 hold the two token halves of column l of the wave's fragment (columns 32 wc .. + 31 of its row block)
        const float tot = wave_halves<WaveAdd>(bsum);
        const int n = n0 + wr * 128 + wc * 32 + i;
        if (g == 0 && n < N) part_b[((size_t)split * nslot + (nslot > 1 ? kt_ : 0)) * N + n] = tot;
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
    static const int en = mbx_env_int("MBX_TN256", 1);
    return en && N >= 256 && K >= 256;
}
static int tnp_splits(int M, int N, int K, bool x3 = false) {
    const bool big = tn_use256(N, K);
    const int tiles = big ? ((N + U_BN - 1) / U_BN) * ((K + U_BK - 1) / U_BK) : ((N + T_BN - 1) / T_BN) * ((K + T_BK - 1) / T_BK);
    const int nchunks = (x3 ? 3 : 1) * (big ? (M + U_BMS - 1) / U_BMS : (M + T_BMS - 1) / T_BMS);
    // one workgroup per CU: the launch should be an exact number of 256-workgroup waves (measured: 288 blocks
    // cost 1.10 ms where 768 cost 0.77 ms on the QKV weight gradient) and a multiple of the 8 XCDs
    int s = 0;
    for (int w = 1; w <= 4; ++w)
        if ((256 * w) % tiles == 0 && ((256 * w) / tiles) % 8 == 0 && (256 * w) / tiles <= 128) { s = (256 * w) / tiles; break; }
    if (s == 0 && big) {
        // (round 6) no split count <= 128 gives whole rounds: 1 or 3 output tiles -- MotionBERT-Lite's dW of proj [256, 256] and of qkv
        // [768, 256], which ran on 128 (half a round) and 384 workgroups (a round and a half).  Pick the multiple of 8 up to 256 that
        // minimises  rounds x chunks per split x ~1 us  +  the partials' round trip (written here, read by the column sum) at ~4 TB/s
        double best = 1e30;
        for (int c = 8; c <= 256; c += 8) {
            const int rounds = (tiles * c + 255) / 256, cps = (nchunks + c - 1) / c;
            const double cost = (double)rounds * cps + (double)c * N * K * 8.0 / 4e6;
            if (cost < best) { best = cost; s = c; }
        }
    }
    if (s == 0) { s = ((512 / tiles + 7) / 8) * 8; if (s > 128) s = 128; }
    if (s > nchunks) s = nchunks;
    if (s < 1) s = 1;
    return s;
}
// partial bias-gradient rows per token split (the k tiles of an n panel share the work: see gemm_tn_pipe256_kernel)
static int tn_db_slots(int N, int K) {
    if (!tn_use256(N, K)) return 1;
    const int ntk = (K + U_BK - 1) / U_BK;
    return (ntk == 2 || ntk == 4) ? ntk : 1;
}
size_t mbx_gemm_tn_pipe_ws(int M, int N, int K) {
    const size_t sp = tnp_splits(M, N, K);
    return (sp * N * K + sp * 4 * N) * sizeof(float) + 256;
}
int mbx_launch_gemm_tn_pipe(const void* dy, const void* a, float* dw, float* db, int M, int N, int K, void* ws, hipStream_t s) {
    const bool big = tn_use256(N, K);
    const int ntn = big ? (N + U_BN - 1) / U_BN : (N + T_BN - 1) / T_BN, ntk = big ? (K + U_BK - 1) / U_BK : (K + T_BK - 1) / T_BK;
    const int splits = tnp_splits(M, N, K);
    const int nchunks = big ? (M + U_BMS - 1) / U_BMS : (M + T_BMS - 1) / T_BMS;
    const int cps = (nchunks + splits - 1) / splits;
    const int slots = tn_db_slots(N, K);
    float* part_w = splits == 1 ? dw : (float*)ws;
    float* part_b = db ? (splits * slots == 1 ? db : (float*)ws + (size_t)splits * N * K) : nullptr;
    if (big) {
        const size_t shm256 = 4 * U_STAGE;
        const int ntiles256 = ntn * ntk, groups256 = (splits + 7) / 8;
        if (set_lds_attr(gemm_tn_pipe256_kernel<false>, shm256, "gemm_tn_pipe256")) return 1;
#ifdef MBX_TN_TRACE
        {
            static long long* const tb = [] { const char* e = getenv("MBX_TRACE_BUF"); return e ? (long long*)strtoull(e, nullptr, 0) : (long long*)nullptr; }();
            (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_tn_trace), &tb, sizeof(tb), 0, hipMemcpyHostToDevice, s);
        }
#endif
        hipLaunchKernelGGL(gemm_tn_pipe256_kernel<false>, dim3(8 * groups256 * ntiles256), dim3(512), shm256, s, (const bf16_t*)dy,
                           (const bf16_t*)a, (const bf16_t*)nullptr, (const bf16_t*)nullptr, part_w, part_b, M, N, K, ntk, ntiles256,
                           splits, cps);
        MBX_LAUNCH_CHECK("gemm_tn_pipe256");
        if (splits > 1 && mbx_launch_colsum(part_w, splits, N * K, 0, N * K, dw, s)) return 1;
        if (db && splits * slots > 1 && mbx_launch_colsum(part_b, splits * slots, N, 0, N, db, s)) return 1;
        return 0;
    }
    const size_t shm = 3 * T_STAGE;
    static const int dbg = mbx_env_int("MBX_DBG", 0);
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

// fp32-class split-operand weight gradient (precision 'bf16x3'); N, K >= 256 and multiples of 256 are not required (clamped
// columns as in the bf16 kernel) but the 256 x 256 kernel is always used.
size_t mbx_gemm_tn_x3_ws(int M, int N, int K) {
    const size_t sp = tnp_splits(M, N, K, true);
    return (sp * N * K + sp * 4 * N) * sizeof(float) + 256;
}
int mbx_launch_gemm_tn_x3(const void* dy_hi, const void* dy_lo, const void* a_hi, const void* a_lo, float* dw, float* db, int M, int N,
                          int K, void* ws, hipStream_t s) {
    const int ntn = (N + U_BN - 1) / U_BN, ntk = (K + U_BK - 1) / U_BK;
    const int splits = tnp_splits(M, N, K, true);
    const int nchunks = 3 * ((M + U_BMS - 1) / U_BMS);
    const int cps = (nchunks + splits - 1) / splits;
    const int ntk_ = (K + U_BK - 1) / U_BK, slots = (ntk_ == 2 || ntk_ == 4) ? ntk_ : 1;
    float* part_w = splits == 1 ? dw : (float*)ws;
    float* part_b = db ? (splits * slots == 1 ? db : (float*)ws + (size_t)splits * N * K) : nullptr;
    const size_t shm256 = 4 * U_STAGE;
    if (set_lds_attr(gemm_tn_pipe256_kernel<true>, shm256, "gemm_tn_x3")) return 1;
    const int ntiles256 = ntn * ntk, groups256 = (splits + 7) / 8;
    hipLaunchKernelGGL(gemm_tn_pipe256_kernel<true>, dim3(8 * groups256 * ntiles256), dim3(512), shm256, s, (const bf16_t*)dy_hi,
                       (const bf16_t*)a_hi, (const bf16_t*)dy_lo, (const bf16_t*)a_lo, part_w, part_b, M, N, K, ntk, ntiles256, splits, cps);
    MBX_LAUNCH_CHECK("gemm_tn_x3");
    if (splits > 1 && mbx_launch_colsum(part_w, splits, N * K, 0, N * K, dw, s)) return 1;
    if (db && splits * slots > 1 && mbx_launch_colsum(part_b, splits * slots, N, 0, N, db, s)) return 1;
    return 0;
}
