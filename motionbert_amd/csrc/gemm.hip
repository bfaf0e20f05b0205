// MFMA GEMMs of the DSTformer hot path for gfx950 (94 % of the model's FLOPs).
//
//   gemm_nt :  acc[M,N] = A[M,K] . W[N,K]^T      every nn.Linear forward and every dX GEMM
//   gemm_tn :  dW[N,K]  = dY[M,N]^T . A[M,K]     every weight gradient (contraction over the tokens)
//
// Both run the same inner product on a 128 x 128 output tile per 256-thread workgroup (4 waves in a
// 2 x 2 grid, each wave 64 x 64 = 2 x 2 MFMA tiles of 32 x 32):
//   bf16 : v_mfma_f32_32x32x16_bf16, 16-byte LDS fragment reads (8 bf16 along k per lane)
//   fp32 : v_mfma_f32_32x32x2_f32,   8-byte LDS fragment reads (exact fp32, the 1e-3 parity mode)
// LDS holds two operand tiles of [128 rows][128 bytes] per stage (k contiguous in a row), double
// buffered (64 KiB -> 2 workgroups per CU).  A row is eight 16-byte chunks; chunk c of row r lives
// at physical chunk  c ^ ((r >> 1) & 7)  so that the 16-lane groups of ds_read_b128 hit 16 distinct
// 16-byte slots of the 256-byte bank row (conflict-free fragment reads) and the 8-lane groups of
// ds_write_b128 stay conflict-free as well.
// The MFMA is issued "transposed" (A-operand = the N-side tile, B-operand = the M-side tile) so
// that a lane ends up holding 4 consecutive output columns of one output row: the fused epilogue
// (bias / erf-GELU / residual add / tanh / GELU') then works on 16-byte (fp32) or 8-byte (bf16)
// vectors straight from the accumulator registers.
// Workgroup -> tile order is XCD-aware: the 8 XCDs have private L2s and the dispatcher round-robins
// consecutive workgroup ids over them, so ids are remapped to give each XCD a contiguous run of
// tiles; consecutive tiles walk N first, i.e. they re-use the same A rows out of that XCD's L2.
// gemm_tn transposes its operands on the way into LDS (8x8 bf16 / 4x4 fp32 register transposes, the
// k-major token dimension becomes the contiguous one), splits the token dimension over workgroups
// and leaves fp32 partial tiles that a deterministic column-sum folds (no atomics); the bias
// gradient (column sums of dY) rides along in the staging registers.
#include "mbx_common.h"
#include "lds_stream.h"
#include <stdlib.h>

template <typename T> struct GemmT;
template <> struct GemmT<bf16_t> { static constexpr int BK = 64, EPC = 8; };
template <> struct GemmT<float>  { static constexpr int BK = 32, EPC = 4; };

static constexpr int G_BM = 128, G_BN = 128, G_ROWB = 128, G_TILEB = 128 * 128;  // bytes per operand tile

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * G_ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

// bijective remap of the dispatch id so that each XCD (id % 8) owns a contiguous range of tiles
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- one BK-deep step of the 64x64 wave tile out of LDS ------------------------------------------
// sA: tile whose rows become accumulator ROWS (MFMA A operand), sB: rows become accumulator COLUMNS.
template <typename T>
__device__ __forceinline__ void wave_mma(const char* sA, const char* sB, int rowA0, int rowB0, int lane,
                                         f32x16_t (&acc)[2][2]);
template <>
__device__ __forceinline__ void wave_mma<bf16_t>(const char* sA, const char* sB, int rowA0, int rowB0, int lane,
                                                 f32x16_t (&acc)[2][2]) {
    const int i = lane & 31, g = lane >> 5;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        bf16x8_t fa[2], fb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            fa[t] = *reinterpret_cast<const bf16x8_t*>(sA + lds_off(rowA0 + t * 32 + i, 2 * s + g));
            fb[t] = *reinterpret_cast<const bf16x8_t*>(sB + lds_off(rowB0 + t * 32 + i, 2 * s + g));
        }
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
                acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ta], fb[tb], acc[ta][tb], 0, 0, 0);
    }
}
template <>
__device__ __forceinline__ void wave_mma<float>(const char* sA, const char* sB, int rowA0, int rowB0, int lane,
                                                f32x16_t (&acc)[2][2]) {
    const int i = lane & 31, g = lane >> 5;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float2 fa[2], fb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            fa[t] = *reinterpret_cast<const float2*>(sA + lds_off(rowA0 + t * 32 + i, c) + g * 8);
            fb[t] = *reinterpret_cast<const float2*>(sB + lds_off(rowB0 + t * 32 + i, c) + g * 8);
        }
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ta].x, fb[tb].x, acc[ta][tb], 0, 0, 0);
                acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ta].y, fb[tb].y, acc[ta][tb], 0, 0, 0);
            }
    }
}

// ================================================================================================
// gemm_nt
// ================================================================================================
template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                         const float* __restrict__ bias, T* __restrict__ out_t,
                                                         T* __restrict__ out2_t, float* __restrict__ out_f,
                                                         const float* __restrict__ resid, const T* __restrict__ aux,
                                                         int M, int N, int K, int ntn) {
    constexpr int BK = GemmT<T>::BK, EPC = GemmT<T>::EPC;
    __shared__ __attribute__((aligned(16))) char smem[4 * G_TILEB];  // [stage][A|W]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (lid % ntn) * G_BN, m0 = (lid / ntn) * G_BM;
    const int wm = wave >> 1, wn = wave & 1;

    // global -> register staging: 4 x 16 B per operand per thread; thread owns chunk `ch` of rows r0 + 32 i
    const int r0 = tid >> 3, ch = tid & 7;
    const int soff0 = lds_off(r0, ch);  // rows r0 + 32 i share the swizzle term: offset = soff0 + i * 32 * 128
    const T* pa[4];
    const T* pw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pa[i] = A + (size_t)min(m0 + r0 + 32 * i, M - 1) * K + ch * EPC;
        pw[i] = W + (size_t)min(n0 + r0 + 32 * i, N - 1) * K + ch * EPC;
    }
    uint4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
#define NT_GLOAD(kt)                                                                   \
    do {                                                                               \
        const size_t ko_ = (size_t)(kt) * BK;                                          \
        ra0 = *reinterpret_cast<const uint4*>(pa[0] + ko_);                            \
        ra1 = *reinterpret_cast<const uint4*>(pa[1] + ko_);                            \
        ra2 = *reinterpret_cast<const uint4*>(pa[2] + ko_);                            \
        ra3 = *reinterpret_cast<const uint4*>(pa[3] + ko_);                            \
        rw0 = *reinterpret_cast<const uint4*>(pw[0] + ko_);                            \
        rw1 = *reinterpret_cast<const uint4*>(pw[1] + ko_);                            \
        rw2 = *reinterpret_cast<const uint4*>(pw[2] + ko_);                            \
        rw3 = *reinterpret_cast<const uint4*>(pw[3] + ko_);                            \
    } while (0)
#define NT_SSTORE(stage)                                                               \
    do {                                                                               \
        char* sa_ = smem + (stage) * 2 * G_TILEB + soff0;                              \
        *reinterpret_cast<uint4*>(sa_) = ra0;                                          \
        *reinterpret_cast<uint4*>(sa_ + 32 * G_ROWB) = ra1;                            \
        *reinterpret_cast<uint4*>(sa_ + 64 * G_ROWB) = ra2;                            \
        *reinterpret_cast<uint4*>(sa_ + 96 * G_ROWB) = ra3;                            \
        *reinterpret_cast<uint4*>(sa_ + G_TILEB) = rw0;                                \
        *reinterpret_cast<uint4*>(sa_ + G_TILEB + 32 * G_ROWB) = rw1;                  \
        *reinterpret_cast<uint4*>(sa_ + G_TILEB + 64 * G_ROWB) = rw2;                  \
        *reinterpret_cast<uint4*>(sa_ + G_TILEB + 96 * G_ROWB) = rw3;                  \
    } while (0)

    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = K / BK;
    NT_GLOAD(0);
    NT_SSTORE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) NT_GLOAD(kt + 1);
        const char* sa = smem + cur * 2 * G_TILEB;
        // accumulator rows <- W tile rows (n), accumulator columns <- A tile rows (m)
        wave_mma<T>(sa + G_TILEB, sa, wn * 64, wm * 64, lane, acc);
        if (kt + 1 < nk) NT_SSTORE(cur ^ 1);
        __syncthreads();
    }
#undef NT_GLOAD
#undef NT_SSTORE

    // ---- fused epilogue: lane owns row m, columns n .. n+3 per register quad ----------------------
    const int i = lane & 31, g = lane >> 5;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const int m = m0 + wm * 64 + tm * 32 + i;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + tn * 32 + 8 * q + 4 * g;
                if (m < M && n < N) {
                float v[4] = {acc[tn][tm][4 * q], acc[tn][tm][4 * q + 1], acc[tn][tm][4 * q + 2], acc[tn][tm][4 * q + 3]};
                if (bias) {
                    float bb[4];
                    load4<float>(bias + n, bb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bb[e];
                }
                const size_t o = (size_t)m * N + n;
                if (EPI == MBX_EPI_STORE) {
                    store4<T>(out_t + o, v);
                } else if (EPI == MBX_EPI_GELU) {
                    if (out_t) store4<T>(out_t + o, v);   // pre-activation is only needed for backward
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                    store4<T>(out2_t + o, v);
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
                    load4<T>(aux + o, u);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(u[e]);
                    store4<T>(out_t + o, v);
                }
                }
            }
        }
    }
}

template <typename T>
static int launch_gemm_nt(const void* a, const void* w, const float* bias, int epi, void* out_t, void* out2_t,
                          float* out_f, const float* resid, const void* aux, int M, int N, int K, hipStream_t s) {
    const int ntn = (N + G_BN - 1) / G_BN, ntm = (M + G_BM - 1) / G_BM;
    dim3 grid((unsigned)ntn * ntm), block(256);
#define MBX_NT_CASE(E)                                                                                              \
    case E:                                                                                                         \
        hipLaunchKernelGGL((gemm_nt_kernel<T, E>), grid, block, 0, s, (const T*)a, (const T*)w, bias, (T*)out_t,    \
                           (T*)out2_t, out_f, resid, (const T*)aux, M, N, K, ntn);                                  \
        break;
    switch (epi) {
        MBX_NT_CASE(MBX_EPI_STORE)
        MBX_NT_CASE(MBX_EPI_GELU)
        MBX_NT_CASE(MBX_EPI_RESID)
        MBX_NT_CASE(MBX_EPI_TANH)
        MBX_NT_CASE(MBX_EPI_DGELU)
        default: return mbx_set_error("gemm_nt: unknown epilogue %d", epi);
    }
#undef MBX_NT_CASE
    MBX_LAUNCH_CHECK("gemm_nt");
    return 0;
}

extern "C" int mbx_gemm_nt(const void* a, const void* w, const float* bias, int epilogue, void* out_t, void* out2_t,
                           float* out_f, const float* resid, const void* aux_t, int M, int N, int K, int dtype,
                           void* stream) {
    MBX_CHECK_ARG(a && w, "gemm_nt: null operand");
    MBX_CHECK_ARG(M > 0 && N > 0 && K > 0 && N % 8 == 0, "gemm_nt: bad shape M=%d N=%d K=%d (N %% 8 != 0?)", M, N, K);
    MBX_CHECK_ARG((size_t)M * (size_t)(N > K ? N : K) < ((size_t)1 << 40), "gemm_nt: operand too large");
    switch (epilogue) {
        case MBX_EPI_STORE: MBX_CHECK_ARG(out_t, "gemm_nt: STORE needs out_t"); break;
        case MBX_EPI_GELU: MBX_CHECK_ARG(out2_t, "gemm_nt: GELU needs out2_t (out_t, the pre-activation, is optional)"); break;
        case MBX_EPI_RESID: MBX_CHECK_ARG(out_f && resid, "gemm_nt: RESID needs out_f and resid"); break;
        case MBX_EPI_TANH: MBX_CHECK_ARG(out_f, "gemm_nt: TANH needs out_f"); break;
        case MBX_EPI_DGELU: MBX_CHECK_ARG(out_t && aux_t, "gemm_nt: DGELU needs out_t and aux_t"); break;
        default: return mbx_set_error("gemm_nt: unknown epilogue %d", epilogue);
    }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MBX_BF16) {
        MBX_CHECK_ARG(K % GemmT<bf16_t>::BK == 0, "gemm_nt(bf16): K=%d must be a multiple of 64", K);
        if (!mbx_use_v1_gemm())
            return mbx_launch_gemm_nt_pipe(a, w, bias, epilogue, out_t, out2_t, out_f, resid, aux_t, M, N, K, s);
        return launch_gemm_nt<bf16_t>(a, w, bias, epilogue, out_t, out2_t, out_f, resid, aux_t, M, N, K, s);
    }
    if (dtype == MBX_F32) {
        MBX_CHECK_ARG(K % GemmT<float>::BK == 0, "gemm_nt(f32): K=%d must be a multiple of 32", K);
        return launch_gemm_nt<float>(a, w, bias, epilogue, out_t, out2_t, out_f, resid, aux_t, M, N, K, s);
    }
    return mbx_set_error("gemm_nt: unknown dtype %d", dtype);
}

// ================================================================================================
// gemm_tn : dW[N,K] = dY[M,N]^T . A[M,K]
// ================================================================================================
// register transposes of one 16-byte-wide block: RB rows (tokens) x RB columns -> RB rows of RB tokens
template <typename T> struct TBlock;
template <> struct TBlock<bf16_t> {
    static constexpr int RB = 8;
    static __device__ __forceinline__ void transpose(const uint4 (&r)[8], uint4 (&o)[8]) {
        const uint32_t* rw = reinterpret_cast<const uint32_t*>(r);  // rw[row*4 + d]: columns 2d (lo), 2d+1 (hi)
        uint32_t* ow = reinterpret_cast<uint32_t*>(o);              // ow[col*4 + d]: rows 2d (lo), 2d+1 (hi)
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t lo = rw[(2 * d) * 4 + (c >> 1)], hi = rw[(2 * d + 1) * 4 + (c >> 1)];
                ow[c * 4 + d] = (c & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
            }
    }
    static __device__ __forceinline__ void colsum(const uint4 (&r)[8], float (&s)[8]) {
        const uint32_t* rw = reinterpret_cast<const uint32_t*>(r);
#pragma unroll
        for (int row = 0; row < 8; ++row)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                s[2 * d] += __uint_as_float(rw[row * 4 + d] << 16);
                s[2 * d + 1] += __uint_as_float(rw[row * 4 + d] & 0xffff0000u);
            }
    }
};
template <> struct TBlock<float> {
    static constexpr int RB = 4;
    static __device__ __forceinline__ void transpose(const uint4 (&r)[4], uint4 (&o)[4]) {
        o[0] = make_uint4(r[0].x, r[1].x, r[2].x, r[3].x);
        o[1] = make_uint4(r[0].y, r[1].y, r[2].y, r[3].y);
        o[2] = make_uint4(r[0].z, r[1].z, r[2].z, r[3].z);
        o[3] = make_uint4(r[0].w, r[1].w, r[2].w, r[3].w);
    }
    static __device__ __forceinline__ void colsum(const uint4 (&r)[4], float (&s)[4]) {
#pragma unroll
        for (int row = 0; row < 4; ++row) {
            s[0] += __uint_as_float(r[row].x); s[1] += __uint_as_float(r[row].y);
            s[2] += __uint_as_float(r[row].z); s[3] += __uint_as_float(r[row].w);
        }
    }
};

// part_w: [splits][N*K] fp32 partial tiles (or dW itself when splits == 1); part_b: [splits][N] or NULL
template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const T* __restrict__ dY, const T* __restrict__ A,
                                                         float* __restrict__ part_w, float* __restrict__ part_b, int M,
                                                         int N, int K, int ntk, int chunks_per_split) {
    constexpr int RB = TBlock<T>::RB;
    constexpr int BMS = 8 * RB;              // tokens per stage: 64 (bf16) / 32 (fp32) = 128 bytes per LDS row
    constexpr int NBLK = 8 * (128 / RB);     // RB x RB blocks per operand tile
    constexpr int ITERS = 2 * NBLK / 256;    // blocks per thread per stage: 1 (bf16) / 2 (fp32)
    __shared__ __attribute__((aligned(16))) char smem[4 * G_TILEB];  // [stage][dY^T | A^T]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x, split = blockIdx.y;
    const int n0 = (tile / ntk) * 128, k0 = (tile % ntk) * 128;
    const int wr = wave >> 1, wc = wave & 1;
    const bool want_db = (part_b != nullptr) && (k0 == 0);
    const int nchunks = (M + BMS - 1) / BMS;
    const int c_beg = split * chunks_per_split, c_end = min(nchunks, c_beg + chunks_per_split);

    uint4 reg[ITERS][RB];
    float bsum[RB];
#pragma unroll
    for (int e = 0; e < RB; ++e) bsum[e] = 0.f;

    auto gload = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int blk = tid + 256 * it, op = blk / NBLK, b = blk % NBLK, mb = b & 7, nb = b >> 3;
            const T* P = op ? A : dY;
            const int ld = op ? K : N, col = (op ? k0 : n0) + nb * RB;
            const int mrow = chunk * BMS + mb * RB;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                if (mrow + i < M && col < ld)
                    reg[it][i] = *reinterpret_cast<const uint4*>(P + (size_t)(mrow + i) * ld + col);
                else
                    reg[it][i] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    auto sstore = [&](int stage) {
        char* base = smem + stage * 2 * G_TILEB;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int blk = tid + 256 * it, op = blk / NBLK, b = blk % NBLK, mb = b & 7, nb = b >> 3;
            if (want_db && op == 0) TBlock<T>::colsum(reg[it], bsum);
            uint4 o[RB];
            TBlock<T>::transpose(reg[it], o);
            char* dst = base + op * G_TILEB;
#pragma unroll
            for (int j = 0; j < RB; ++j) *reinterpret_cast<uint4*>(dst + lds_off(nb * RB + j, mb)) = o[j];
        }
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (c_beg < c_end) {
        gload(c_beg);
        sstore(0);
    }
    __syncthreads();
    for (int c = c_beg; c < c_end; ++c) {
        const int cur = (c - c_beg) & 1;
        if (c + 1 < c_end) gload(c + 1);
        const char* base = smem + cur * 2 * G_TILEB;
        // accumulator rows <- dY^T rows (n), accumulator columns <- A^T rows (k)
        wave_mma<T>(base, base + G_TILEB, wr * 64, wc * 64, lane, acc);
        if (c + 1 < c_end) sstore(cur ^ 1);
        __syncthreads();
    }

    // store the partial tile: lane column = k (coalesced along k), register rows = n
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

    // bias gradient: fold the per-thread column sums of the dY blocks (threads with op == 0)
    if (want_db) {
        float* red = reinterpret_cast<float*>(smem);  // [8 mb][128 n]
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int blk = tid + 256 * it, op = blk / NBLK, b = blk % NBLK, mb = b & 7, nb = b >> 3;
            if (op == 0) {
                // with ITERS == 2 (fp32) the first iteration is always op 0 and holds the whole sum
#pragma unroll
                for (int e = 0; e < RB; ++e) red[mb * 128 + nb * RB + e] = bsum[e];
            }
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < N) {
            float s = 0.f;
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) s += red[mb * 128 + tid];
            part_b[(size_t)split * N + n0 + tid] = s;
        }
    }
}

// ---- fp32 weight gradient, round 4: no transposes at all.  v_mfma_f32_32x32x2_f32 takes ONE float per lane and operand: lane
// (i, kk) holds dY[t + kk][n_i] (A operand) and A[t + kk][k_i] (B operand) -- 32 consecutive floats of a token row for the lanes of a
// half wave, which is how the rows lie in HBM.  So the token panels go to LDS row-major exactly as they are (LDS-DMA, one
// instruction = two token rows x 512 bytes, whole lines), an operand is a `ds_read2_b32` (both 32-column tiles of the wave at once,
// conflict-free: 32 consecutive dwords per half wave), and the loop is 4 MFMAs of 64 cycles per two LDS instructions.  The
// round-1 kernel above (register-staged, 4 x 4 register transposes on the way into LDS) ran at 57 TFLOP/s with 40 % LDS bank
// conflicts and 2.3x its operand bytes in fetches; it stays for shapes that are not multiples of 128.
//   tile 128 (n) x 128 (k), 4 waves 2 x 2, 16 tokens per stage (2 x 8 KiB), 4-stage ring = 64 KiB: two workgroups per CU.
static constexpr int TF_BT = 16, TF_STAGE = 2 * TF_BT * 512, TF_NST = 4;
__global__ __launch_bounds__(256, 2) void gemm_tn_f32_kernel(const float* __restrict__ dY, const float* __restrict__ A,
                                                             float* __restrict__ part_w, float* __restrict__ part_b, int M, int N,
                                                             int K, int ntk, int chunks_per_split, int nsplits) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kk = lane >> 5;
    // Workgroup -> (token split, output tile): consecutive dispatch ids go round-robin over the 8 XCDs (private L2s), so ALL tiles of a
    // token split are given to ONE XCD (split = xcd + 8 m): they read the same two token panels at about the same time, and each
    // panel leaves HBM once per launch instead of once per XCD that holds one of its tiles (measured before: 2.3x the operand bytes).
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, ntiles = ntk * (N / 128);
    const int tile = idx % ntiles, split = xcd + 8 * (idx / ntiles);
    if (split >= nsplits) return;
    const int n0 = (tile / ntk) * 128, k0 = (tile % ntk) * 128;
    const int wr = wave >> 1, wc = wave & 1;
    const bool want_db = (part_b != nullptr) && (k0 == 0) && (wc == 0);
    const int nchunks = (M + TF_BT - 1) / TF_BT;
    const int c_beg = split * chunks_per_split, c_end = min(nchunks, c_beg + chunks_per_split);
    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum[2] = {0.f, 0.f};
    if (c_beg < c_end) {
        // this wave's four DMA instructions of a stage: instruction j = wave + 4 u: operand j >> 3 (0 dY, 1 A), token pair j & 7;
        // lane l moves the 16-byte piece (l & 31) of token row (l >> 5) of the pair.  Tokens past M repeat row M - 1 (masked at use).
        const unsigned ldsb = (unsigned)(uintptr_t)(const lds_void_t*)smem;
        auto issue = [&](int chunk) {
            const int cc = min(chunk, c_end - 1);                 // past the end: a harmless re-read, so that the wait counts stay constant
            const unsigned st = ldsb + ((chunk - c_beg) & (TF_NST - 1)) * TF_STAGE;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = wave + 4 * u, op = j >> 3, rp = j & 7;
                const int tok = min(cc * TF_BT + 2 * rp + kk, M - 1);
                const float* src = op ? A + (size_t)tok * K + k0 + 4 * i : dY + (size_t)tok * N + n0 + 4 * i;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(st + op * (TF_BT * 512) + rp * 1024) : "memory");
            }
        };
        issue(c_beg); issue(c_beg + 1); issue(c_beg + 2);
        // operand reads: dword (wr 64 + i) of row (2 p + kk) of the dY panel and + 32 dwords; the same with wc in the A panel
        const unsigned ra = ldsb + kk * 512 + (wr * 64 + i) * 4, rb = ldsb + TF_BT * 512 + kk * 512 + (wc * 64 + i) * 4;
        typedef __attribute__((address_space(3))) const float lds_f32_t;
        for (int c = c_beg; c < c_end; ++c) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // stage c has landed (this wave's pieces; the two stages behind it may be in flight)
            __builtin_amdgcn_s_barrier();                         // ... everyone's; and everyone has left stage c - 1, whose slot is refilled now
            issue(c + 3);
            const unsigned so = ((c - c_beg) & (TF_NST - 1)) * TF_STAGE;
            const bool ragged = (c + 1) * TF_BT > M;              // wave-uniform: only the very last chunk
#pragma unroll
            for (int p = 0; p < TF_BT / 2; ++p) {
                float a0 = *reinterpret_cast<lds_f32_t*>(ra + so + p * 1024), a1 = *reinterpret_cast<lds_f32_t*>(ra + so + p * 1024 + 128);
                const float b0 = *reinterpret_cast<lds_f32_t*>(rb + so + p * 1024), b1 = *reinterpret_cast<lds_f32_t*>(rb + so + p * 1024 + 128);
                if (ragged && c * TF_BT + 2 * p + kk >= M) { a0 = 0.f; a1 = 0.f; }
                if (want_db) { bsum[0] += a0; bsum[1] += a1; }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the re-read tail stages have landed: nothing arrives in LDS after the workgroup ends
    }
    // the partial tile: lane column = k (coalesced along k), register rows = n
    float* pw = part_w + (size_t)split * N * K;
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc) {
            const int k = k0 + wc * 64 + tc * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wr * 64 + tr * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                pw[(size_t)n * K + k] = acc[tr][tc][r];
            }
        }
    if (want_db) {     // column sums of dY: this lane's tokens (parity kk) + the other half wave's
#pragma unroll
        for (int tr = 0; tr < 2; ++tr) {
            const float sum = wave_halves<WaveAdd>(bsum[tr]);
            if (kk == 0) part_b[(size_t)split * N + n0 + wr * 64 + tr * 32 + i] = sum;
        }
    }
}

static int tn_splits(int M, int N, int K, int bms) {
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    const int nchunks = (M + bms - 1) / bms;
    int s = (1024 + tiles - 1) / tiles;
    if (s >= 8) {      // a multiple of 8 (the fp32 kernel gives whole splits to XCDs), preferably one that fills whole rounds of 512 workgroups
        s = (s + 7) / 8 * 8;
        for (int t = s; t <= 64; t += 8)
            if ((tiles * t) % 512 == 0) { s = t; break; }
    }
    if (s > 64) s = 64;
    if (s > nchunks) s = nchunks;
    if (s < 1) s = 1;
    return s;
}
bool mbx_use_v1_gemm() {
    static const bool v1 = mbx_env_int("MBX_GEMM_V1", 0) == 1;
    return v1;
}
extern "C" size_t mbx_gemm_tn_ws(int M, int N, int K) {
    const int s = tn_splits(M, N, K, 32);  // upper bound over both dtypes
    const size_t v1 = ((size_t)s * N * K + (size_t)s * N) * sizeof(float) + 256;
    const size_t v2 = mbx_gemm_tn_pipe_ws(M, N, K);
    return v1 > v2 ? v1 : v2;
}
static int launch_gemm_tn_f32(const void* dy, const void* a, float* dw, float* db, int M, int N, int K, void* ws, hipStream_t s) {
    const int ntn = N / 128, ntk = K / 128;
    const int nchunks = (M + TF_BT - 1) / TF_BT;
    const int splits = tn_splits(M, N, K, 32);                      // the split count the workspace was sized for
    const int cps = (nchunks + splits - 1) / splits;
    float* part_w = splits == 1 ? dw : (float*)ws;
    float* part_b = db ? (splits == 1 ? db : (float*)ws + (size_t)splits * N * K) : nullptr;
    if (mbx_set_dyn_lds(reinterpret_cast<const void*>(gemm_tn_f32_kernel), TF_NST * TF_STAGE, "gemm_tn")) return 1;
    hipLaunchKernelGGL(gemm_tn_f32_kernel, dim3(8 * ntn * ntk * ((splits + 7) / 8)), dim3(256), TF_NST * TF_STAGE, s, (const float*)dy,
                       (const float*)a, part_w, part_b, M, N, K, ntk, cps, splits);
    MBX_LAUNCH_CHECK("gemm_tn_f32");
    if (splits > 1) {
        if (mbx_launch_colsum(part_w, splits, N * K, 0, N * K, dw, s)) return 1;
        if (db && mbx_launch_colsum(part_b, splits, N, 0, N, db, s)) return 1;
    }
    return 0;
}
template <typename T>
static int launch_gemm_tn(const void* dy, const void* a, float* dw, float* db, int M, int N, int K, void* ws, hipStream_t s) {
    constexpr int BMS = 8 * TBlock<T>::RB;
    const int ntn = (N + 127) / 128, ntk = (K + 127) / 128;
    const int splits = tn_splits(M, N, K, BMS);
    const int nchunks = (M + BMS - 1) / BMS;
    const int cps = (nchunks + splits - 1) / splits;
    float* part_w = splits == 1 ? dw : (float*)ws;
    float* part_b = db ? (splits == 1 ? db : (float*)ws + (size_t)splits * N * K) : nullptr;
    hipLaunchKernelGGL((gemm_tn_kernel<T>), dim3(ntn * ntk, splits), dim3(256), 0, s, (const T*)dy, (const T*)a, part_w,
                       part_b, M, N, K, ntk, cps);
    MBX_LAUNCH_CHECK("gemm_tn");
    if (splits > 1) {
        if (mbx_launch_colsum(part_w, splits, N * K, 0, N * K, dw, s)) return 1;
        if (db && mbx_launch_colsum(part_b, splits, N, 0, N, db, s)) return 1;
    }
    return 0;
}
extern "C" int mbx_gemm_tn(const void* dy, const void* a, float* dw, float* db, int M, int N, int K, int dtype, void* ws,
                           void* stream) {
    MBX_CHECK_ARG(dy && a && dw && ws, "gemm_tn: null pointer");
    MBX_CHECK_ARG(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0, "gemm_tn: bad shape M=%d N=%d K=%d (N, K %% 8)", M, N, K);
    MBX_CHECK_ARG((size_t)N * K < ((size_t)1 << 31), "gemm_tn: output too large");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MBX_BF16) {
        if (!mbx_use_v1_gemm()) return mbx_launch_gemm_tn_pipe(dy, a, dw, db, M, N, K, ws, s);
        return launch_gemm_tn<bf16_t>(dy, a, dw, db, M, N, K, ws, s);
    }
    if (dtype == MBX_F32)
        return (N % 128 == 0 && K % 128 == 0 && M >= 4 * TF_BT) ? launch_gemm_tn_f32(dy, a, dw, db, M, N, K, ws, s)
                                                               : launch_gemm_tn<float>(dy, a, dw, db, M, N, K, ws, s);
    return mbx_set_error("gemm_tn: unknown dtype %d", dtype);
}

// ================================================================================================
// fp32-class split-operand GEMMs (precision 'bf16x3'): operands arrive as bf16 hi / lo planes (mbx_split_bf16,
// mbx_prep_weights with MBX_BF16_LO), every T-typed output / auxiliary tensor is fp32.
// ================================================================================================
extern "C" int mbx_gemm_nt_x3(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias,
                              int epilogue, float* out_t, float* out2_t, float* out_f, const float* resid, const float* aux_t,
                              int M, int N, int K, void* stream) {
    MBX_CHECK_ARG(a_hi && a_lo && w_hi && w_lo, "gemm_nt_x3: null operand");
    MBX_CHECK_ARG(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 64 == 0, "gemm_nt_x3: bad shape M=%d N=%d K=%d (N %% 8, K %% 64)", M, N, K);
    switch (epilogue) {
        case MBX_EPI_STORE: MBX_CHECK_ARG(out_t, "gemm_nt_x3: STORE needs out_t"); break;
        case MBX_EPI_GELU: MBX_CHECK_ARG(out2_t, "gemm_nt_x3: GELU needs out2_t"); break;
        case MBX_EPI_RESID: MBX_CHECK_ARG(out_f && resid, "gemm_nt_x3: RESID needs out_f and resid"); break;
        case MBX_EPI_TANH: MBX_CHECK_ARG(out_f, "gemm_nt_x3: TANH needs out_f"); break;
        case MBX_EPI_DGELU: MBX_CHECK_ARG(out_t && aux_t, "gemm_nt_x3: DGELU needs out_t and aux_t"); break;
        default: return mbx_set_error("gemm_nt_x3: unknown epilogue %d", epilogue);
    }
    return mbx_launch_gemm_nt_x3(a_hi, a_lo, w_hi, w_lo, bias, epilogue, out_t, out2_t, out_f, resid, aux_t, M, N, K, (hipStream_t)stream);
}
extern "C" int mbx_gemm_nt_x3p(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias, int epilogue,
                               float* out_t, void* pl_hi, void* pl_lo, const float* aux_t, int M, int N, int K, void* stream) {
    MBX_CHECK_ARG(a_hi && a_lo && w_hi && w_lo && pl_hi && pl_lo, "gemm_nt_x3p: null operand / plane");
    MBX_CHECK_ARG(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 64 == 0, "gemm_nt_x3p: bad shape M=%d N=%d K=%d (N %% 8, K %% 64)", M, N, K);
    MBX_CHECK_ARG(epilogue == MBX_EPI_STORE || epilogue == MBX_EPI_GELU || epilogue == MBX_EPI_DGELU, "gemm_nt_x3p: epilogue %d has no plane output", epilogue);
    MBX_CHECK_ARG(epilogue != MBX_EPI_DGELU || aux_t, "gemm_nt_x3p: DGELU needs aux_t");
    return mbx_launch_gemm_nt_x3(a_hi, a_lo, w_hi, w_lo, bias, epilogue, out_t, nullptr, nullptr, nullptr, aux_t, M, N, K, (hipStream_t)stream,
                                 pl_hi, pl_lo);
}
extern "C" size_t mbx_gemm_tn_x3_workspace(int M, int N, int K) { return mbx_gemm_tn_x3_ws(M, N, K); }
extern "C" int mbx_gemm_tn_x3(const void* dy_hi, const void* dy_lo, const void* a_hi, const void* a_lo, float* dw, float* db, int M,
                              int N, int K, void* ws, void* stream) {
    MBX_CHECK_ARG(dy_hi && dy_lo && a_hi && a_lo && dw && ws, "gemm_tn_x3: null pointer");
    MBX_CHECK_ARG(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0, "gemm_tn_x3: bad shape M=%d N=%d K=%d (N, K %% 8)", M, N, K);
    MBX_CHECK_ARG((size_t)N * K < ((size_t)1 << 31), "gemm_tn_x3: output too large");
    return mbx_launch_gemm_tn_x3(dy_hi, dy_lo, a_hi, a_lo, dw, db, M, N, K, ws, (hipStream_t)stream);
}
