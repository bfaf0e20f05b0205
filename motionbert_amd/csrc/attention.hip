// Spatial (17 joints) and temporal (<= 243 frames) multi-head attention of the DSTformer for gfx950
// (reference: lib/model/DSTformer.py:178-186 forward_spatial, :188-200 forward_temporal).
//
// One "problem" is one (frame, head) [spatial, L = J keys] or one (clip, joint, head) [temporal,
// L = T keys].  Sequence element s of a problem is token tok0 + s*tstep of the packed qkv tensor
// [M, 3C] (channel order [3][H][hd]); the temporal problems therefore read their rows with a stride
// of J*3C elements straight from the [B,T,J,3C] layout -- no permute/contiguous copies.
//
// Everything is MFMA 32x32 (bf16: 32x32x16, fp32: 32x32x2) in the "transposed" orientation: the
// query (or, in the dK/dV kernel, the key) index is the LANE, so each lane owns whole softmax rows:
//     S^T[key][q]  = sum_d K[key][d]  Q[q][d]      A operand: K rows from LDS,  B operand: Q from registers
//     O^T[d][q]    = sum_k V^T[d][k]  P^T[k][q]    A operand: V^T rows from LDS, B operand: P, already in
//                                                  registers in exactly the layout the MFMA wants
// so the softmax needs no LDS and only ONE cross-lane exchange (lanes l and l+32 share a query):
// row max and row sum are in-register reductions followed by a single v_permlane32_swap (wave_halves).
// The score matrix [B,H,J,T,T] that the reference materialises (32 MB per clip) never exists;
// backward recomputes the probabilities from q, k and the saved log-sum-exp.
//
// LDS tiles (per problem): row-major [KP][hd] tiles padded by 16 B per row (conflict-free 16-byte
// fragment reads).  Products that contract over the SEQUENCE index (P.V, dS^T.Q, ...) need that index
// contiguous per lane although it is the row index of the tile: bf16 fetches those operands with the
// gfx950 transpose read ds_read_b64_tr_b16 (lane r of a 16-lane group addresses 4 d-columns of row
// r >> 2; the hardware hands lane i the 4 consecutive rows of column i), fp32 operands are single floats
// per lane and are read directly.  No transposed copies exist in LDS.
// Short sequences (L <= 32: every spatial problem, temporal with T <= 32) run one problem per wave
// with wave-private LDS, four problems per 256-thread workgroup; longer ones share the tiles across the
// eight waves of a workgroup, one 32-row query (key) block per wave.  Kernel map:
//   attn_fwd_kernel<T, HD, SHARED>        forward, both forms
//   attn_bwd_small_kernel<T, HD>          backward, L <= 32: dQ, dK, dV from tiles staged once per wave
//   attn_bwd_fused_kernel<HD>             backward, bf16, 32 < L <= 256: ONE workgroup of 16 waves per problem, Q / K / V / dO
//                                         staged once, dQ role (8 waves) and dK/dV role (8 waves) concurrently
//   attn_bwd_dq_kernel / attn_bwd_dkv_kernel   backward, fp32 (tiles twice as large: the four of them do not fit one CU's
//                                         LDS), and the A/B baseline of -DMBX_ATTN_BWD_TWO_KERNELS builds
// Outputs never leave a kernel in the accumulator layout (lane = row, 4 consecutive d per register quad: one store
// instruction would touch 32 rows x 16 bytes).  They are staged through LDS tiles that are dead by then and copied out with
// eight lanes per 128-byte row segment.
#include "mbx_common.h"

// threads per workgroup: long sequences (K/V or Q/dO shared in LDS) use 8 waves -- one 32-row block each for T <= 256 --
// so that the two workgroups a CU holds (72 KiB of LDS each) give every SIMD four waves to hide latency behind
template <bool SHARED> struct AttnBlock { static constexpr int THREADS = SHARED ? 512 : 256, WAVES = THREADS / 64; };
// the dK/dV kernel needs ~170 VGPRs (two 32x64 accumulators per lane besides the score fragments): eight waves per
// workgroup would not raise its occupancy, and four waves with eight key blocks each measured faster
constexpr int MBX_ATTN_KV_THREADS = 256;
template <bool SHARED> struct AttnBlockKV { static constexpr int THREADS = SHARED ? MBX_ATTN_KV_THREADS : 256, WAVES = THREADS / 64; };

// ------------------------------------------------------------------------------------------------
// per-type helpers
// ------------------------------------------------------------------------------------------------
template <typename T> struct AT;
template <> struct AT<bf16_t> { static constexpr int EPC = 8, RB = 8, SZ = 2; };
template <> struct AT<float>  { static constexpr int EPC = 4, RB = 4, SZ = 4; };

// B operand of the "rows" products: the d-vector of ONE sequence element, held by the two lanes
// (g = 0, 1) that own it.   bf16: v[s] = 8 bf16 at d = 16 s + 8 g;   fp32: v[c] = 2 floats at d = 4 c + 2 g
template <typename T, int HD> struct BReg;
template <int HD> struct BReg<bf16_t, HD> {
    uint4 v[HD / 16];
    __device__ __forceinline__ void load(const bf16_t* row, int g, bool valid) {
#pragma unroll
        for (int s = 0; s < HD / 16; ++s)
            v[s] = valid ? *reinterpret_cast<const uint4*>(row + 16 * s + 8 * g) : make_uint4(0u, 0u, 0u, 0u);
    }
    // sum_d a[d]*b[d] over this lane's half of the d range
    static __device__ __forceinline__ float dot(const BReg& a, const BReg& b) {
        float acc = 0.f;
        const uint32_t* x = reinterpret_cast<const uint32_t*>(a.v);
        const uint32_t* y = reinterpret_cast<const uint32_t*>(b.v);
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) {
            acc = fmaf(__uint_as_float(x[i] << 16), __uint_as_float(y[i] << 16), acc);
            acc = fmaf(__uint_as_float(x[i] & 0xffff0000u), __uint_as_float(y[i] & 0xffff0000u), acc);
        }
        return acc;
    }
};
template <int HD> struct BReg<float, HD> {
    float2 v[HD / 4];
    __device__ __forceinline__ void load(const float* row, int g, bool valid) {
#pragma unroll
        for (int c = 0; c < HD / 4; ++c)
            v[c] = valid ? *reinterpret_cast<const float2*>(row + 4 * c + 2 * g) : make_float2(0.f, 0.f);
    }
    static __device__ __forceinline__ float dot(const BReg& a, const BReg& b) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) acc = fmaf(a.v[c].x, b.v[c].x, fmaf(a.v[c].y, b.v[c].y, acc));
        return acc;
    }
};

// acc[i][lane] += sum_d tile[row0 + i][d] * B[lane][d]     (tile row-major in LDS, `stride` bytes per row)
template <typename T, int HD> struct MmaRows;
template <int HD> struct MmaRows<bf16_t, HD> {
    static __device__ __forceinline__ void run(const char* tile, int stride, int row0, const BReg<bf16_t, HD>& b, int lane,
                                               f32x16_t& acc) {
        const char* p = tile + (size_t)(row0 + (lane & 31)) * stride + (lane >> 5) * 16;
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) {
            const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(p + s * 32);
            const bf16x8_t bb = *reinterpret_cast<const bf16x8_t*>(&b.v[s]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc, 0, 0, 0);
        }
    }
};
template <int HD> struct MmaRows<float, HD> {
    static __device__ __forceinline__ void run(const char* tile, int stride, int row0, const BReg<float, HD>& b, int lane,
                                               f32x16_t& acc) {
        const char* p = tile + (size_t)(row0 + (lane & 31)) * stride + (lane >> 5) * 8;
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) {
            const float2 a = *reinterpret_cast<const float2*>(p + c * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.v[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.v[c].y, acc, 0, 0, 0);
        }
    }
};

// acc[i][lane] += sum_{e in fragment f} X[e][d0 + i] * P[e][lane]
// P is a 32x32 accumulator fragment (16 registers): register r of lane (., g) belongs to sequence
// element  e(f, r, g) = 32 f + (r & 3) + 8 (r >> 2) + 4 g.
//   bf16: `tile` is the TRANSPOSED tile [d][e] (e contiguous): two 8-byte reads give the 8 elements of a k-step
//   fp32: `tile` is the ROW-MAJOR tile [e][d]: scalar reads, consecutive lanes -> consecutive d
template <typename T> struct MmaCols;
typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s_t lds_v4s_t;
template <> struct MmaCols<bf16_t> {
    static __device__ __forceinline__ void run(const char* tile, int stride, int d0, int f, const f32x16_t& p, int lane,
                                               f32x16_t& acc) {
        const int g = lane >> 5, r16 = lane & 15;
        // this lane's address: row base + (r16 >> 2), d-column block d0 + 16*((lane>>4)&1) + 4*(r16&3)
        const char* a0 = tile + (size_t)(32 * f + 4 * g + (r16 >> 2)) * stride + (d0 + 16 * ((lane >> 4) & 1) + 4 * (r16 & 3)) * 2;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            union { uint32_t u[4]; bf16x8_t v; } pb;
            union { v4s_t h[2]; bf16x8_t v; } a;
#pragma unroll
            for (int e = 0; e < 4; ++e) pb.u[e] = pack_bf2(p[8 * t + 2 * e], p[8 * t + 2 * e + 1]);
            a.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(a0 + (size_t)(16 * t) * stride));       // rows base .. base+3
            a.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(a0 + (size_t)(16 * t + 8) * stride));   // rows base+8 .. base+11
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, pb.v, acc, 0, 0, 0);
        }
    }
};
template <> struct MmaCols<float> {
    static __device__ __forceinline__ void run(const char* tile, int stride, int d0, int f, const f32x16_t& p, int lane,
                                               f32x16_t& acc) {
        const int g = lane >> 5;
        const char* col = tile + (size_t)(d0 + (lane & 31)) * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = 32 * f + (r & 3) + 8 * (r >> 2) + 4 * g;
            const float a = *reinterpret_cast<const float*>(col + (size_t)e * stride);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, p[r], acc, 0, 0, 0);
        }
    }
};

// ---- LDS tile fills (cooperative over `gsize` threads, this thread = gtid) -------------------------
// Two row-major tiles at once: dstX[row][0..HD) = srcX[row * rstrideX + 0..HD) for row < L, zero for
// L <= row < KP.  All global loads of both tiles are issued before the first LDS store (the rows are
// kilobytes apart, so each load is a separate HBM/L2 round trip: they must overlap, not serialize).
// NPT = chunks per thread and tile: CH covers KP <= gsize; callers with KP <= gsize / 2 (32 rows per 64 lanes, <= 256 rows per 512
// threads) pass CH / 2 and keep half of the staging registers
template <typename T, int HD, int NPT = HD / AT<T>::EPC>
__device__ __forceinline__ void fill_two(char* dst0, const T* src0, size_t rs0, char* dst1, const T* src1, size_t rs1, int stride,
                                         int L, int KP, int gtid, int gsize, uint4 (&v0)[NPT], uint4 (&v1)[NPT]) {
    constexpr int CH = HD / AT<T>::EPC;   // 16-byte chunks per row
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int idx = gtid + i * gsize, row = idx / CH, ch = idx % CH;
        v0[i] = make_uint4(0u, 0u, 0u, 0u);
        v1[i] = make_uint4(0u, 0u, 0u, 0u);
        if (row < L) {
            v0[i] = *reinterpret_cast<const uint4*>(src0 + (size_t)row * rs0 + ch * AT<T>::EPC);
            v1[i] = *reinterpret_cast<const uint4*>(src1 + (size_t)row * rs1 + ch * AT<T>::EPC);
        }
    }
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int idx = gtid + i * gsize, row = idx / CH, ch = idx % CH;
        if (row < KP) {
            *reinterpret_cast<uint4*>(dst0 + (size_t)row * stride + ch * 16) = v0[i];
            *reinterpret_cast<uint4*>(dst1 + (size_t)row * stride + ch * 16) = v1[i];
        }
    }
}
template <typename T, int HD, int NPT = HD / AT<T>::EPC>
__device__ __forceinline__ void fill_two(char* dst0, const T* src0, size_t rs0, char* dst1, const T* src1, size_t rs1, int stride,
                                         int L, int KP, int gtid, int gsize) {
    uint4 v0[NPT], v1[NPT];
    fill_two<T, HD, NPT>(dst0, src0, rs0, dst1, src1, rs1, stride, L, KP, gtid, gsize, v0, v1);
}
// store an accumulator pair/quad set: lane owns sequence element `row`, registers own d
template <typename T, int HD>
__device__ __forceinline__ void store_rowfrag(T* row, const f32x16_t (&acc)[HD / 32], float mul, int g) {
#pragma unroll
    for (int df = 0; df < HD / 32; ++df)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v[4] = {acc[df][4 * q] * mul, acc[df][4 * q + 1] * mul, acc[df][4 * q + 2] * mul, acc[df][4 * q + 3] * mul};
            store4<T>(row + df * 32 + 8 * q + 4 * g, v);
        }
}

// the same as the two bf16 planes of the bf16x3 operand split (fp32-class mode: dq / dk / dv are only read by split-operand GEMMs)
template <int HD>
__device__ __forceinline__ void store_rowfrag_planes(bf16_t* hi, bf16_t* lo, const f32x16_t (&acc)[HD / 32], int g) {
#pragma unroll
    for (int df = 0; df < HD / 32; ++df)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v[4] = {acc[df][4 * q], acc[df][4 * q + 1], acc[df][4 * q + 2], acc[df][4 * q + 3]};
            store4_planes(hi + df * 32 + 8 * q + 4 * g, lo + df * 32 + 8 * q + 4 * g, v);
        }
}

struct Prob {
    size_t tok0;
    size_t dbase;     // flat index of this problem's probability element (0, 0) in the reference's attn tensor (dropout masks)
    int tstep, L, h;
};
__device__ __forceinline__ Prob decode_prob(int prob, int mode, int Tn, int J, int H) {
    Prob p;
    p.h = prob % H;
    const int rest = prob / H;
    if (mode == MBX_ATTN_SPATIAL) {
        p.tok0 = (size_t)rest * J; p.tstep = 1; p.L = J;
        p.dbase = ((size_t)rest * H + p.h) * J * J;                       // attn [B T, H, J, J]       (DSTformer.py:180)
    } else {
        const int j = rest % J, b = rest / J;
        p.tok0 = (size_t)b * Tn * J + j; p.tstep = J; p.L = Tn;
        p.dbase = (((size_t)b * H + p.h) * J + j) * Tn * Tn;              // attn [B, H, J, T, T]      (DSTformer.py:194)
    }
    return p;
}
// Dropout on the attention probabilities (nn.Dropout(attn_drop), DSTformer.py:96,182,196): multiplier keep / (1 - p) of element
// (query qi, key ki) -- the counter-based mask of dropmask.py over the flat index of the reference's attn tensor.  The softmax
// statistics (row max, row sum, lse) are those of the UNdropped probabilities; in backward dP = mask (dO V^T) and
// delta = rowsum(dO O) as without dropout (O already carries the mask).
__device__ __forceinline__ float drop_mul(const MbxDrop& dr, size_t dbase, int qi, int ki, int L) {
    const size_t idx = dbase + (size_t)qi * L + ki;
    return drop_keep(dr.seed_lo, dr.seed_hi, (uint32_t)idx, (uint32_t)(idx >> 32), dr.thresh) ? dr.scale : 0.f;
}

template <typename T> __host__ __device__ constexpr int rm_stride(int HD) { return HD * AT<T>::SZ + 16; }   // row-major tile

// ================================================================================================
// forward: flash-style walk over 32-key fragments with a running row max / row sum (the whole row
// is at most 8 fragments, but keeping only one fragment of scores live keeps the wave at ~100 VGPRs)
// ================================================================================================
template <typename T, int HD, bool SHARED, bool DROP = false>
__global__ __launch_bounds__(AttnBlock<SHARED>::THREADS, (sizeof(T) == 2 && !DROP) ? 4 : 1) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o, float* __restrict__ lse,
                                                       int Tn, int J, int H, float scale, int mode, int nprob, int KP, MbxDrop dr) {
    constexpr bool IS_BF = sizeof(T) == 2;
    constexpr int KSTR = rm_stride<T>(HD);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5;
    const int C = H * HD, C3 = 3 * C;
    constexpr int VSTR = KSTR;
    const int KBYTES = KP * KSTR;
    const int VBYTES = KP * VSTR;
    int prob = SHARED ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    const bool pvalid = prob < nprob;
    prob = min(prob, nprob - 1);
    const Prob P = decode_prob(prob, mode, Tn, J, H);
    const int gtid = SHARED ? tid : lane, gsize = SHARED ? AttnBlock<SHARED>::THREADS : 64;
    char* kt = smem + (SHARED ? 0 : wave * (KBYTES + VBYTES));
    char* vt = kt + KBYTES;
    const size_t rstride = (size_t)P.tstep * C3;
    const T* base = qkv + P.tok0 * C3 + (size_t)P.h * HD;
    const int nfr = (P.L + 31) / 32;
    // L <= 256 (check_attn_args): at most eight 32-query blocks -- one per wave of a shared workgroup, one in all for a
    // wave-private problem
    const int qb = SHARED ? wave : 0;
    const int q = qb * 32 + (lane & 31);
    // this lane's query row is requested BEFORE the K / V fill (round 3): one memory round trip for the prologue instead of two
constexpr int MBX_ATTN_Q_EARLY = 1;
    BReg<T, HD> qreg;
    if (MBX_ATTN_Q_EARLY && qb < nfr)
        qreg.load(qkv + (P.tok0 + (size_t)min(q, P.L - 1) * P.tstep) * C3 + (size_t)P.h * HD, g, pvalid && q < P.L);
    fill_two<T, HD, HD / AT<T>::EPC / 2>(kt, base + C, rstride, vt, base + 2 * C, rstride, KSTR, P.L, KP, gtid, gsize);
    __syncthreads();

    const float c2 = scale * 1.44269504088896341f;
    f32x16_t oacc[HD / 32];
#pragma unroll
    for (int df = 0; df < HD / 32; ++df)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[df][r] = 0.f;
    float l = 1.f;
    if (qb < nfr) {
        const bool qvalid = pvalid && q < P.L;
        const size_t tok = P.tok0 + (size_t)min(q, P.L - 1) * P.tstep;
        if (!MBX_ATTN_Q_EARLY) qreg.load(qkv + tok * C3 + (size_t)P.h * HD, g, qvalid);
        // softmax in base 2: p = 2^(s*c2 - m2) with c2 = scale*log2(e) -- one fma + one v_exp_f32 per score; the max is
        // taken over the raw scores (scale > 0) and only the tail fragment masks keys past L
        float m2 = -INFINITY;
        l = 0.f;                        // l: this lane's half of the row sum (lanes l, l^32 share a query)
        for (int f = 0; f < nfr; ++f) {
            f32x16_t s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            MmaRows<T, HD>::run(kt, KSTR, 32 * f, qreg, lane, s);
            if (32 * f + 32 > P.L) {   // wave-uniform: the last, partial fragment
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * f + (r & 3) + 8 * (r >> 2) + 4 * g;
                    s[r] = key < P.L ? s[r] : -INFINITY;
                }
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = wave_halves<WaveMax>(mx);   // every fragment f < nfr holds >= 1 valid key: finite
            const float mn2 = fmaxf(m2, mx * c2);
            const float corr = __builtin_amdgcn_exp2f(m2 - mn2);        // first fragment: 2^(-inf) = 0
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -mn2));
                ps += s[r];
            }
            l = fmaf(l, corr, ps);
            m2 = mn2;
            if (DROP) {       // the row sum above is the undropped one; what multiplies V carries the mask
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] *= drop_mul(dr, P.dbase, q, 32 * f + (r & 3) + 8 * (r >> 2) + 4 * g, P.L);
            }
#pragma unroll
            for (int df = 0; df < HD / 32; ++df) {
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[df][r] *= corr;
                MmaCols<T>::run(vt, VSTR, df * 32, f, s, lane, oacc[df]);
            }
        }
        l = wave_halves<WaveAdd>(l);
        if (qvalid && g == 0) lse[tok * H + P.h] = (m2 + __builtin_amdgcn_logf(l)) * 0.69314718055994531f;   // natural-log units
    }
    // ---- output: staged through the (now dead) K tile so that eight lanes write one whole row segment per instruction; the
    // accumulator layout stored directly touches 32 rows x 16 bytes per instruction ----
    if (SHARED) __syncthreads();            // every wave is done with K and V (wave-private tiles: in-order LDS, no barrier)
    if (qb < nfr) store_rowfrag<T, HD>(reinterpret_cast<T*>(kt + (size_t)q * KSTR), oacc, 1.0f / l, g);
    if (SHARED) __syncthreads();
    if (pvalid) {
        constexpr int CH = HD * (int)sizeof(T) / 16;
        T* ob = o + P.tok0 * C + (size_t)P.h * HD;
        const size_t ostride = (size_t)P.tstep * C;
        for (int idx = gtid; idx < P.L * CH; idx += gsize) {
            const int r = idx / CH, ch = idx % CH;
            const uint4 a = *reinterpret_cast<const uint4*>(kt + r * KSTR + ch * 16);
            *reinterpret_cast<uint4*>(ob + (size_t)r * ostride + ch * (16 / (int)sizeof(T))) = a;
        }
    }
}

// ================================================================================================
// backward, part 1: dQ   (lane = query)
// ================================================================================================
template <typename T, int HD, bool SHARED, bool DROP = false>
__global__ __launch_bounds__(AttnBlock<SHARED>::THREADS, (SHARED && sizeof(T) == 2 && !DROP) ? 4 : 1) void attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ o,
                                                          const T* __restrict__ d_o, const float* __restrict__ lse,
                                                          T* __restrict__ dqkv, int Tn, int J, int H, float scale, int mode,
                                                          int nprob, int KP, MbxDrop dr, bf16_t* __restrict__ dq_lo) {
    constexpr bool IS_BF = sizeof(T) == 2;
    constexpr int RSTR = rm_stride<T>(HD);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5;
    const int C = H * HD, C3 = 3 * C;
    const int per_prob = 2 * KP * RSTR;
    int prob = SHARED ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    const bool pvalid = prob < nprob;
    prob = min(prob, nprob - 1);
    const Prob P = decode_prob(prob, mode, Tn, J, H);
    const int gtid = SHARED ? tid : lane, gsize = SHARED ? AttnBlock<SHARED>::THREADS : 64;
    char* kt = smem + (SHARED ? 0 : wave * per_prob);
    char* vt = kt + KP * RSTR;
    const size_t rstride = (size_t)P.tstep * C3;
    const T* base = qkv + P.tok0 * C3 + (size_t)P.h * HD;
    fill_two<T, HD>(kt, base + C, rstride, vt, base + 2 * C, rstride, RSTR, P.L, KP, gtid, gsize);
    __syncthreads();

    const int nfr = (P.L + 31) / 32;
    for (int qb = SHARED ? wave : 0; qb < nfr; qb += SHARED ? AttnBlock<SHARED>::WAVES : 1) {
        const int q = qb * 32 + (lane & 31);
        const bool qvalid = pvalid && q < P.L;
        const size_t tok = P.tok0 + (size_t)min(q, P.L - 1) * P.tstep;
        BReg<T, HD> qreg, doreg, oreg;
        qreg.load(qkv + tok * C3 + (size_t)P.h * HD, g, qvalid);
        doreg.load(d_o + tok * C + (size_t)P.h * HD, g, qvalid);
        oreg.load(o + tok * C + (size_t)P.h * HD, g, qvalid);
        float delta = BReg<T, HD>::dot(doreg, oreg);
        delta = wave_halves<WaveAdd>(delta);
        // p = exp(s*scale - lse) = 2^(s*c2 - lse*log2 e): one fma + one v_exp_f32.  No masks: an invalid query lane is
        // never stored, and a padded key has a zero K row (dS * K = 0) and a zero V row (dP = 0)
        const float lq2 = qvalid ? lse[tok * H + P.h] * 1.44269504088896341f : 0.f;
        const float c2 = scale * 1.44269504088896341f;

        f32x16_t dq[HD / 32];
#pragma unroll
        for (int df = 0; df < HD / 32; ++df)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[df][r] = 0.f;
        for (int f = 0; f < nfr; ++f) {
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            MmaRows<T, HD>::run(kt, RSTR, 32 * f, qreg, lane, s);
            MmaRows<T, HD>::run(vt, RSTR, 32 * f, doreg, lane, dp);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lq2));
                const float dpr = DROP ? dp[r] * drop_mul(dr, P.dbase, q, 32 * f + (r & 3) + 8 * (r >> 2) + 4 * g, P.L) : dp[r];
                s[r] = p * (dpr - delta) * scale;  // dS
            }
#pragma unroll
            for (int df = 0; df < HD / 32; ++df) MmaCols<T>::run(kt, RSTR, df * 32, f, s, lane, dq[df]);
        }
        if (qvalid) {
            const size_t off = tok * C3 + (size_t)P.h * HD;
            if (sizeof(T) == 4 && dq_lo) store_rowfrag_planes<HD>(reinterpret_cast<bf16_t*>(dqkv) + off, dq_lo + off, dq, g);   // dqkv = the hi plane
            else store_rowfrag<T, HD>(dqkv + off, dq, 1.0f, g);
        }
    }
}

// ================================================================================================
// backward, part 2: dK, dV   (lane = key)
// ================================================================================================
template <typename T, int HD, bool SHARED, bool DROP = false>
__global__ __launch_bounds__(AttnBlockKV<SHARED>::THREADS, (SHARED && !DROP) ? 2 : 1) void attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ o,
                                                           const T* __restrict__ d_o, const float* __restrict__ lse,
                                                           T* __restrict__ dqkv, int Tn, int J, int H, float scale, int mode,
                                                           int nprob, int KP, MbxDrop dr, bf16_t* __restrict__ dq_lo) {
    constexpr bool IS_BF = sizeof(T) == 2;
    constexpr int RSTR = rm_stride<T>(HD);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5;
    const int C = H * HD, C3 = 3 * C;
    const int per_prob = 2 * KP * RSTR + 2 * KP * 4;
    int prob = SHARED ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    const bool pvalid = prob < nprob;
    prob = min(prob, nprob - 1);
    const Prob P = decode_prob(prob, mode, Tn, J, H);
    const int gtid = SHARED ? tid : lane, gsize = SHARED ? AttnBlockKV<SHARED>::THREADS : 64;
    char* qt = smem + (SHARED ? 0 : wave * per_prob);   // Q   [KP][hd]
    char* dot_ = qt + KP * RSTR;                          // dO  [KP][hd]
    float* lse_s = reinterpret_cast<float*>(dot_ + KP * RSTR);
    float* del_s = lse_s + KP;
    const size_t rstride = (size_t)P.tstep * C3, ostride = (size_t)P.tstep * C;
    const T* qbase = qkv + P.tok0 * C3 + (size_t)P.h * HD;
    const T* dobase = d_o + P.tok0 * C + (size_t)P.h * HD;
    const T* obase = o + P.tok0 * C + (size_t)P.h * HD;
    fill_two<T, HD>(qt, qbase, rstride, dot_, dobase, ostride, RSTR, P.L, KP, gtid, gsize);
    // per-query statistics: lse and delta = sum_d dO*O.  KP <= gsize: one row per thread, all 2*HD/4 (8) vector
    // loads of the row in flight together.
    {
        const int q = gtid;
        if (q < KP) {
            float l = 0.f, dl = 0.f;
            if (q < P.L) {
                l = lse[(P.tok0 + (size_t)q * P.tstep) * H + P.h];
                const T* a = dobase + (size_t)q * ostride;
                const T* b = obase + (size_t)q * ostride;
                float x[HD / 4][4], y[HD / 4][4];
#pragma unroll
                for (int d = 0; d < HD / 4; ++d) { load4<T>(a + 4 * d, x[d]); load4<T>(b + 4 * d, y[d]); }
#pragma unroll
                for (int d = 0; d < HD / 4; ++d)
                    dl = fmaf(x[d][0], y[d][0], fmaf(x[d][1], y[d][1], fmaf(x[d][2], y[d][2], fmaf(x[d][3], y[d][3], dl))));
            }
            lse_s[q] = l * 1.44269504088896341f;   // base-2 units: p = 2^(s*c2 - lse2)
            del_s[q] = dl;
        }
    }
    __syncthreads();

    const int nfr = (P.L + 31) / 32;
    const float c2 = scale * 1.44269504088896341f;
    for (int kb = SHARED ? wave : 0; kb < nfr; kb += SHARED ? AttnBlockKV<SHARED>::WAVES : 1) {
        const int key = kb * 32 + (lane & 31);
        const bool kvalid = pvalid && key < P.L;
        const size_t tok = P.tok0 + (size_t)min(key, P.L - 1) * P.tstep;
        BReg<T, HD> kreg, vreg;
        kreg.load(qkv + tok * C3 + C + (size_t)P.h * HD, g, kvalid);
        vreg.load(qkv + tok * C3 + 2 * C + (size_t)P.h * HD, g, kvalid);
        f32x16_t dk[HD / 32], dv[HD / 32];
#pragma unroll
        for (int df = 0; df < HD / 32; ++df)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[df][r] = 0.f; dv[df][r] = 0.f; }
        for (int f = 0; f < nfr; ++f) {
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            MmaRows<T, HD>::run(qt, RSTR, 32 * f, kreg, lane, s);      // s[r]  <-> (query e(f,r,g), key = lane)
            MmaRows<T, HD>::run(dot_, RSTR, 32 * f, vreg, lane, dp);   // dP
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int q0 = 32 * f + 8 * qd + 4 * g;
                const float4 l4 = *reinterpret_cast<const float4*>(lse_s + q0);
                const float4 d4 = *reinterpret_cast<const float4*>(del_s + q0);
                const float la[4] = {l4.x, l4.y, l4.z, l4.w}, da[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * qd + e;
                    // no masks: padded queries have zero Q and dO rows (dS^T Q = 0, P^T dO = 0), invalid key lanes are not stored
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -la[e]));
                    const float km = DROP ? drop_mul(dr, P.dbase, q0 + e, key, P.L) : 1.f;
                    dp[r] = p * (dp[r] * km - da[e]) * scale;  // dS
                    s[r] = p * km;                             // P (dropped: what multiplied V in forward)
                }
            }
#pragma unroll
            for (int df = 0; df < HD / 32; ++df) {
                MmaCols<T>::run(dot_, RSTR, df * 32, f, s, lane, dv[df]);   // dV^T += dO^T P
                MmaCols<T>::run(qt, RSTR, df * 32, f, dp, lane, dk[df]);    // dK^T += Q^T dS
            }
        }
        if (kvalid) {
            const size_t off = tok * C3 + C + (size_t)P.h * HD;
            if (sizeof(T) == 4 && dq_lo) {
                store_rowfrag_planes<HD>(reinterpret_cast<bf16_t*>(dqkv) + off, dq_lo + off, dk, g);
                store_rowfrag_planes<HD>(reinterpret_cast<bf16_t*>(dqkv) + off + C, dq_lo + off + C, dv, g);
            } else {
                store_rowfrag<T, HD>(dqkv + off, dk, 1.0f, g);
                store_rowfrag<T, HD>(dqkv + off + C, dv, 1.0f, g);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Row dots for the folded LayerNorm backward (round 3; "LayerNorm folding" in elementwise.hip).  With `st_part` the bf16 backward
// kernels also leave, per (token, head), part[2 h + role][m] = { sum d rsum, sum d (y - b') } with role 0 = the head's q columns
// (d = dq, y = q) and role 1 = its k and v columns, d = the bf16-ROUNDED gradient being stored.  They are taken where the
// gradient rows are staged for the copy-out: the lane that writes a row fragment of dq / dk / dv into the LDS tile of q / k / v
// first reads the original values it is about to overwrite (same row, same columns, 8 bytes at a time) -- no extra pass, no
// global re-read (a first version re-read q, k, v in the copy-out loop: +21 % / +28 % on the two kernels), and as packed-bf16
// dot products (v_dot2c_f32_bf16: six VALU operations per four columns; the unpack / fma form cost +12 % / +26 %).  rsum / -b' of
// the head's 3 hd columns sit in LDS as bf16 pairs (`vec`); every lane of a half-wave reads the same address (broadcast).
// ------------------------------------------------------------------------------------------------
// All LDS reads of a 32-column half (originals and vectors) are issued before its first write: interleaved, every iteration
// waits a full LDS round trip on its own reads (the writes may alias as far as the compiler can tell) -- measured +14 % on the
// one-wave kernel.  One half at a time keeps the sixteen-wave kernel inside its 128 VGPRs.
template <int HD>
__device__ __forceinline__ void store_rowfrag_dot(bf16_t* __restrict__ row, const f32x16_t (&acc)[HD / 32], int g,
                                                  const uint4* __restrict__ vec, float& p1, float& p2) {
#pragma unroll
    for (int df = 0; df < HD / 32; ++df) {
        uint2 o[4];      // the original q / k / v values about to be overwritten
        uint4 vv[4];     // {rsum pair, rsum pair, -b' pair, -b' pair} of the same 4 columns
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int d0 = df * 32 + 8 * q + 4 * g;
            o[q] = *reinterpret_cast<const uint2*>(row + d0);
            vv[q] = vec[d0 >> 2];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int d0 = df * 32 + 8 * q + 4 * g;
            const uint32_t lo = pack_bf2(acc[df][4 * q], acc[df][4 * q + 1]), hi = pack_bf2(acc[df][4 * q + 2], acc[df][4 * q + 3]);
            p1 = dot2_bf16(lo, vv[q].x, dot2_bf16(hi, vv[q].y, p1));
            p2 = dot2_bf16(lo, o[q].x, dot2_bf16(hi, o[q].y, dot2_bf16(lo, vv[q].z, dot2_bf16(hi, vv[q].w, p2))));
            *reinterpret_cast<uint2*>(row + d0) = make_uint2(lo, hi);
        }
    }
}
// vec[j * HD / 4 + d / 4] = bf16 pairs {rsum[d], rsum[d+1]}, {rsum[d+2], rsum[d+3]}, {-b'[d], -b'[d+1]}, {-b'[d+2], -b'[d+3]} of the
// head's columns in tensor j = q, k, v (global column j C + h HD + d).  Rounding the two vectors to bf16 moves c1 / c2 by
// ~1e-3 / sqrt(C) of a gradient element (independent errors over the 3 hd columns): far below the bf16 noise of the data.
template <int HD>
__device__ __forceinline__ void fill_stat_vec(uint4* vec, const float* __restrict__ rsum, const float* __restrict__ bias, int C, int h,
                                              int gtid, int gsize) {
    for (int idx = gtid; idx < 3 * HD / 4; idx += gsize) {
        const int j = idx / (HD / 4), d = (idx % (HD / 4)) * 4;
        const float4 r = *reinterpret_cast<const float4*>(rsum + j * C + h * HD + d);
        const float4 b = *reinterpret_cast<const float4*>(bias + j * C + h * HD + d);
        vec[idx] = make_uint4(pack_bf2(r.x, r.y), pack_bf2(r.z, r.w), pack_bf2(-b.x, -b.y), pack_bf2(-b.z, -b.w));
    }
}

// ================================================================================================
// backward for short sequences (L <= 32: every spatial problem), dQ, dK and dV in ONE kernel.
// One problem per wave, four per workgroup; Q, K, V and dO are staged once in wave-private LDS tiles and
// serve both orientations (lane = query for dQ, lane = key for dK/dV), so qkv/dO/O are read from HBM once
// and dqkv is written once: 2.2 GB per launch at 64 clips instead of 3.5 GB for the two-kernel form.
// ================================================================================================
template <typename T, int HD, bool DROP = false>
__global__ __launch_bounds__(256) void attn_bwd_small_kernel(const T* __restrict__ qkv, const T* __restrict__ o,
                                                             const T* __restrict__ d_o, const float* __restrict__ lse,
                                                             T* __restrict__ dqkv, int Tn, int J, int H, float scale, int mode,
                                                             int nprob, const float* __restrict__ st_bias,
                                                             const float* __restrict__ st_rsum, float* __restrict__ st_part, MbxDrop dr,
                                                             bf16_t* __restrict__ dq_lo) {
    constexpr int KP = 32;
    constexpr int RSTR = rm_stride<T>(HD);
    constexpr int TILE = KP * RSTR;
    constexpr int PER = 4 * TILE + 2 * KP * 4;      // four tiles, lse / delta (74.75 KB per workgroup at hd = 64: two workgroups per CU)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, i = lane & 31;
    const int C = H * HD, C3 = 3 * C;
    int prob = (int)blockIdx.x * 4 + wave;
    const bool pvalid = prob < nprob;
    prob = min(prob, nprob - 1);
    const Prob P = decode_prob(prob, mode, Tn, J, H);
    char* qt = smem + wave * PER;
    char* kt = qt + TILE;
    char* vt = kt + TILE;
    char* dot_ = vt + TILE;
    float* lse_s = reinterpret_cast<float*>(dot_ + TILE);
    float* del_s = lse_s + KP;
    uint4* vec = reinterpret_cast<uint4*>(dot_);      // row-dot vectors (stats variant): they move into the dO tile once it is dead
    const bool stats = sizeof(T) == 2 && st_part != nullptr;      // wave-uniform
    const size_t rstride = (size_t)P.tstep * C3, ostride = (size_t)P.tstep * C;
    const T* qbase = qkv + P.tok0 * C3 + (size_t)P.h * HD;
    const T* dobase = d_o + P.tok0 * C + (size_t)P.h * HD;
    const T* obase = o + P.tok0 * C + (size_t)P.h * HD;
constexpr int MBX_ATTN_SMALL_EARLY = 1;
    constexpr bool EARLY = MBX_ATTN_SMALL_EARLY && sizeof(T) == 2;
    uint4 vec_reg = make_uint4(0u, 0u, 0u, 0u);      // this lane's entry of the row-dot vectors (3 hd / 4 <= 48 entries), parked in registers
    if constexpr (EARLY) {
        // ONE memory round trip for the whole prologue: lse and the chunks of all four tiles are requested before the first LDS
        // store.  Before: two fills and the statistics, three round trips in a row, on a kernel with eight waves per CU.
        constexpr int CHB = HD / 8, NPT = KP * CHB / 64;      // 16-byte chunks per row; chunks per lane and tile (4 at hd = 64)
        float l = 0.f;
        if (lane < P.L) l = lse[(P.tok0 + (size_t)lane * P.tstep) * H + P.h];
        uint4 tq[NPT], tk[NPT], tv[NPT], td[NPT];
#pragma unroll
        for (int c = 0; c < NPT; ++c) {
            const int idx = lane + c * 64, row = idx / CHB, ch = idx % CHB;
            tq[c] = tk[c] = tv[c] = td[c] = make_uint4(0u, 0u, 0u, 0u);
            if (row < P.L) {
                const T* r3 = qbase + (size_t)row * rstride + ch * 8;
                tq[c] = *reinterpret_cast<const uint4*>(r3);
                tk[c] = *reinterpret_cast<const uint4*>(r3 + C);
                tv[c] = *reinterpret_cast<const uint4*>(r3 + 2 * C);
                td[c] = *reinterpret_cast<const uint4*>(dobase + (size_t)row * ostride + ch * 8);
            }
        }
        if (stats && lane < 3 * HD / 4) {
            const int j = lane / (HD / 4), d = (lane % (HD / 4)) * 4;
            const float4 r = *reinterpret_cast<const float4*>(st_rsum + j * C + P.h * HD + d);
            const float4 b = *reinterpret_cast<const float4*>(st_bias + j * C + P.h * HD + d);
            vec_reg = make_uint4(pack_bf2(r.x, r.y), pack_bf2(r.z, r.w), pack_bf2(-b.x, -b.y), pack_bf2(-b.z, -b.w));
        }
#pragma unroll
        for (int c = 0; c < NPT; ++c) {
            const int idx = lane + c * 64, off = (idx / CHB) * RSTR + (idx % CHB) * 16;
            *reinterpret_cast<uint4*>(qt + off) = tq[c];
            *reinterpret_cast<uint4*>(kt + off) = tk[c];
            *reinterpret_cast<uint4*>(vt + off) = tv[c];
            *reinterpret_cast<uint4*>(dot_ + off) = td[c];
        }
        if (lane < KP) lse_s[lane] = l;
    } else {
        fill_two<T, HD, HD / AT<T>::EPC / 2>(qt, qbase, rstride, kt, qbase + C, rstride, RSTR, P.L, KP, lane, 64);
        fill_two<T, HD, HD / AT<T>::EPC / 2>(vt, qbase + 2 * C, rstride, dot_, dobase, ostride, RSTR, P.L, KP, lane, 64);
        if (stats && lane < 3 * HD / 4) {
            const int j = lane / (HD / 4), d = (lane % (HD / 4)) * 4;
            const float4 r = *reinterpret_cast<const float4*>(st_rsum + j * C + P.h * HD + d);
            const float4 b = *reinterpret_cast<const float4*>(st_bias + j * C + P.h * HD + d);
            vec_reg = make_uint4(pack_bf2(r.x, r.y), pack_bf2(r.z, r.w), pack_bf2(-b.x, -b.y), pack_bf2(-b.z, -b.w));
        }
        if (lane < KP) lse_s[lane] = lane < P.L ? lse[(P.tok0 + (size_t)lane * P.tstep) * H + P.h] : 0.f;
    }
    __syncthreads();
    const bool rvalid = pvalid && i < P.L;          // this lane's sequence element (query in pass 1, key in pass 2)
    const size_t tok = P.tok0 + (size_t)min(i, P.L - 1) * P.tstep;

    // ---- pass 1, lane = query: dQ ------------------------------------------------------------------
    f32x16_t dq[HD / 32];
    {
        BReg<T, HD> qreg, doreg;
        qreg.load(reinterpret_cast<const T*>(qt + (size_t)i * RSTR), g, true);
        doreg.load(reinterpret_cast<const T*>(dot_ + (size_t)i * RSTR), g, true);
        const float lq = lse_s[i];
        f32x16_t sf, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sf[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int df = 0; df < HD / 32; ++df)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[df][r] = 0.f;
        MmaRows<T, HD>::run(kt, RSTR, 0, qreg, lane, sf);
        MmaRows<T, HD>::run(vt, RSTR, 0, doreg, lane, dp);
        float dl = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * g;
            const float pr = (rvalid && key < P.L) ? __expf(sf[r] * scale - lq) : 0.f;
            const float dpr = DROP ? dp[r] * drop_mul(dr, P.dbase, min(i, P.L - 1), min(key, P.L - 1), P.L) : dp[r];
            sf[r] = pr;
            dp[r] = dpr;
            dl = fmaf(pr, dpr, dl);
        }
        // delta = rowsum(P o dP) -- the same number as dO . O (O = P V, dP = dO V^T), taken from the fragments this lane and its
        // partner (lane ^ 32: the other half of the keys) already hold: O is not read at all (round 3: -270 MB per launch)
        const float delta = wave_halves<WaveAdd>(dl);
        if (g == 0) del_s[i] = delta;      // for pass 2 (lane = key): same wave, LDS operations execute in order
#pragma unroll
        for (int r = 0; r < 16; ++r) sf[r] = sf[r] * (dp[r] - delta) * scale;
#pragma unroll
        for (int df = 0; df < HD / 32; ++df) MmaCols<T>::run(kt, RSTR, df * 32, 0, sf, lane, dq[df]);
    }
    // ---- pass 2, lane = key: dK, dV ----------------------------------------------------------------
    {
        BReg<T, HD> kreg, vreg;
        kreg.load(reinterpret_cast<const T*>(kt + (size_t)i * RSTR), g, true);
        vreg.load(reinterpret_cast<const T*>(vt + (size_t)i * RSTR), g, true);
        // the K tile is dead from here on (wave-private, LDS operations of one wave execute in order): it stages dQ -- unless
        // the row dots are wanted: then dQ waits in registers and every gradient is staged over ITS OWN original at the end
        if (!stats) store_rowfrag<T, HD>(reinterpret_cast<T*>(kt + (size_t)i * RSTR), dq, 1.0f, g);
        f32x16_t sf, dp, dk[HD / 32], dv[HD / 32];
#pragma unroll
        for (int r = 0; r < 16; ++r) { sf[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int df = 0; df < HD / 32; ++df)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[df][r] = 0.f; dv[df][r] = 0.f; }
        MmaRows<T, HD>::run(qt, RSTR, 0, kreg, lane, sf);       // sf[r] <-> (query e(r, g), key = lane)
        MmaRows<T, HD>::run(dot_, RSTR, 0, vreg, lane, dp);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int q0 = 8 * qd + 4 * g;
            const float4 l4 = *reinterpret_cast<const float4*>(lse_s + q0);
            const float4 d4 = *reinterpret_cast<const float4*>(del_s + q0);
            const float la[4] = {l4.x, l4.y, l4.z, l4.w}, da[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * qd + e;
                const float pr = (rvalid && q0 + e < P.L) ? __expf(sf[r] * scale - la[e]) : 0.f;
                const float km = DROP ? drop_mul(dr, P.dbase, min(q0 + e, P.L - 1), min(i, P.L - 1), P.L) : 1.f;
                dp[r] = pr * (dp[r] * km - da[e]) * scale;
                sf[r] = pr * km;
            }
        }
#pragma unroll
        for (int df = 0; df < HD / 32; ++df) {
            MmaCols<T>::run(dot_, RSTR, df * 32, 0, sf, lane, dv[df]);
            MmaCols<T>::run(qt, RSTR, df * 32, 0, dp, lane, dk[df]);
        }
        if constexpr (sizeof(T) == 2) {
            if (stats) {
                // all four tiles are dead as operands; q, k, v are still intact: dq -> Q tile, dk -> K tile, dv -> V tile, each
                // lane taking the two dots of its row fragment with the values it overwrites (vectors: into the dead dO tile first;
                // wave-private LDS executes in order, no barrier)
                if (lane < 3 * HD / 4) vec[lane] = vec_reg;
                float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
                store_rowfrag_dot<HD>(reinterpret_cast<bf16_t*>(qt + (size_t)i * RSTR), dq, g, vec, a1, a2);
                store_rowfrag_dot<HD>(reinterpret_cast<bf16_t*>(kt + (size_t)i * RSTR), dk, g, vec + HD / 4, b1, b2);
                store_rowfrag_dot<HD>(reinterpret_cast<bf16_t*>(vt + (size_t)i * RSTR), dv, g, vec + 2 * (HD / 4), b1, b2);
                a1 = wave_halves<WaveAdd>(a1); a2 = wave_halves<WaveAdd>(a2);      // the two column halves of a row: lanes i, i + 32
                b1 = wave_halves<WaveAdd>(b1); b2 = wave_halves<WaveAdd>(b2);
                if (rvalid && g == 0) {     // part[2 h + role][token]: the joints of a frame are consecutive tokens
                    const size_t Mtot = (size_t)(nprob / H) * P.L;
                    *reinterpret_cast<float2*>(st_part + ((size_t)(2 * P.h) * Mtot + tok) * 2) = make_float2(a1, a2);
                    *reinterpret_cast<float2*>(st_part + ((size_t)(2 * P.h + 1) * Mtot + tok) * 2) = make_float2(b1, b2);
                }
            }
        }
        if (!stats) {
            // V tile (dead since vreg was loaded) stages dK, the Q tile (dead after the last product above) stages dV
            store_rowfrag<T, HD>(reinterpret_cast<T*>(vt + (size_t)i * RSTR), dk, 1.0f, g);
            store_rowfrag<T, HD>(reinterpret_cast<T*>(qt + (size_t)i * RSTR), dv, 1.0f, g);
        }
    }
    // ---- copy out: eight lanes per 16-byte-chunked row, whole row segments per store instruction (the accumulator layout
    // stored directly touches 32 rows x 16 bytes per instruction) ----
    if (pvalid) {
        constexpr int CH = HD * (int)sizeof(T) / 16;
        T* ob = dqkv + P.tok0 * C3 + (size_t)P.h * HD;
        const char* sa = stats ? qt : kt;      // where dq, dk, dv were staged (see above)
        const char* sb = stats ? kt : vt;
        const char* sc = stats ? vt : qt;
        for (int idx = lane; idx < P.L * CH; idx += 64) {
            const int r = idx / CH, ch = idx % CH, off = r * RSTR + ch * 16;
            const uint4 a = *reinterpret_cast<const uint4*>(sa + off);
            const uint4 b = *reinterpret_cast<const uint4*>(sb + off);
            const uint4 c = *reinterpret_cast<const uint4*>(sc + off);
            if (sizeof(T) == 4 && dq_lo) {      // fp32-class mode: the operand planes of the bf16x3 split (dqkv = the hi plane), four floats per chunk
                const size_t eo = P.tok0 * C3 + (size_t)P.h * HD + (size_t)r * rstride + ch * 4;
                bf16_t* const hi = reinterpret_cast<bf16_t*>(dqkv);
                const float va[4] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)};
                const float vb[4] = {__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)};
                const float vc[4] = {__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z), __uint_as_float(c.w)};
                store4_planes(hi + eo, dq_lo + eo, va);
                store4_planes(hi + eo + C, dq_lo + eo + C, vb);
                store4_planes(hi + eo + 2 * C, dq_lo + eo + 2 * C, vc);
                continue;
            }
            T* dst = ob + (size_t)r * rstride + ch * (16 / (int)sizeof(T));
            *reinterpret_cast<uint4*>(dst) = a;
            *reinterpret_cast<uint4*>(dst + C) = b;
            *reinterpret_cast<uint4*>(dst + 2 * C) = c;
        }
    }
}

// ================================================================================================
// backward for long bf16 sequences (temporal attention, 32 < L <= 256): dQ, dK and dV in ONE kernel.
// One workgroup of SIXTEEN waves per problem.  Q, K, V and dO are staged ONCE as row-major LDS tiles (4 x 36 KiB at
// L = 243, hd = 64) together with the per-query statistics (lse, delta = dO.O), so qkv, dO and O are read from HBM
// once and dqkv is written once: 2.2 GB per launch at 64 clips instead of 3.5 GB for the dQ + dK/dV kernel pair.
//   waves 0-7  (lane = query): dQ of query block `wave`      -- S, dP, dS.K    with K, V as A tiles, q / dO rows in registers
//   waves 8-15 (lane = key):   dK, dV of key block `wave-8`  -- S^T, dP^T, P^T.dO, dS^T.Q with Q, dO as A tiles; the k / v
//              rows are read from the LDS tiles as MFMA B operands at every use (not held in registers): with two 32x64
//              accumulators per lane that keeps the role inside the 128 VGPRs of a four-waves-per-SIMD kernel.
// Both roles run concurrently on every SIMD (two waves of each), so the MFMA chain of one hides the exp / pack VALU work
// of the other.  No masks, as in the two-kernel form: padded rows of all four tiles are zero and invalid lanes never store.
// ================================================================================================
__device__ __forceinline__ bf16_t* qt_row(char* tile, int row, int stride) { return reinterpret_cast<bf16_t*>(tile + (size_t)row * stride); }
template <int HD>
__device__ __forceinline__ void mma_rows_ldsb(const char* tile, int stride, int row0, const char* brow, int lane, f32x16_t& acc) {
    const char* p = tile + (size_t)(row0 + (lane & 31)) * stride + (lane >> 5) * 16;
#pragma unroll
    for (int s = 0; s < HD / 16; ++s) {
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(p + s * 32);
        const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(brow + s * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
}

struct Raw4b {     // four bf16 in a uint2 -> fp32
    static __device__ __forceinline__ void unpack(const uint2& r, float (&v)[4]) {
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
        v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    }
};
#ifndef MBX_ATTN_DBG
#define MBX_ATTN_DBG 0      // ablation bits of diagnostic builds (timing only): 1 no compute loops, 2 no copy-out stores, 4 no tile / statistics loads, 8 no exp2 (p = 1)
#endif
// Diagnostic builds only (-DMBX_ATTN_TRACE, tools/attn_trace.py): 8 int64 per workgroup (thread 0) -- s_memrealtime (100 MHz) at entry,
// loads of the fill issued, tiles + statistics in LDS (first barrier passed), compute done (second barrier), gradients staged (third
// barrier), copy-out stores issued, stores acknowledged; the hardware id.  The buffer address comes from MBX_TRACE_BUF.
#ifdef MBX_ATTN_TRACE
__device__ long long* g_attn_trace;
#define AT_TS(slot_) do { if (threadIdx.x == 0) ats[slot_] = (long long)wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define AT_TS(slot_) do { } while (0)
#endif
template <int HD, bool STATS>
__global__ __launch_bounds__(1024, 1) void attn_bwd_fused_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                                 const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                                 bf16_t* __restrict__ dqkv, int Tn, int J, int H, float scale,
                                                                 int mode, int nprob, int KP, const float* __restrict__ st_bias,
                                                                 const float* __restrict__ st_rsum, float* __restrict__ st_part) {
    typedef bf16_t T;
    constexpr int RSTR = rm_stride<T>(HD), CH = HD / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef MBX_ATTN_TRACE
    long long ats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    AT_TS(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 5;
    const int C = H * HD, C3 = 3 * C;
    const Prob P = decode_prob((int)blockIdx.x, mode, Tn, J, H);
    const int TB = KP * RSTR;
    char* qt = smem;
    char* kt = qt + TB;
    char* vt = kt + TB;
    char* dot_ = vt + TB;
    float* lse_s = reinterpret_cast<float*>(dot_ + TB);
    float* del_s = lse_s + KP;
    uint4* vec = reinterpret_cast<uint4*>(del_s + KP);     // behind the statistics: the row-dot vectors (stats variant; KP % 32 == 0 keeps it 16-byte aligned)
    const size_t rstride = (size_t)P.tstep * C3, ostride = (size_t)P.tstep * C;
    const T* qbase = qkv + P.tok0 * C3 + (size_t)P.h * HD;
    const T* dobase = d_o + P.tok0 * C + (size_t)P.h * HD;
    const T* obase = o + P.tok0 * C + (size_t)P.h * HD;
    if (STATS) fill_stat_vec<HD>(vec, st_rsum, st_bias, C, P.h, tid, 1024);

    // ---- stage the four tiles: 16-byte chunks, all loads of a pass in flight before the first LDS store.  The per-query statistics
    // come out of the same loads (round 6): delta = dO . O needs the dO chunk the thread fetches for the tile anyway, so it fetches the
    // matching chunk of O beside it and the CH lanes of a row reduce their 8-element dots on the VALU; lane ch = 0 also fetches lse.
    // (Rounds 2-5 read dO a second time, four lanes per row: 62 instead of 31 KiB of statistics traffic per problem, on a fill that
    // runs at the CU's ~24 KB/us -- profiles/r06_attn_trace.txt.)  KP CH <= 2048: one pass, two chunks per thread. ----
    for (int i0 = tid; i0 < KP * CH; i0 += 2048) {
        uint4 v[2][4], vo[2];
        float sl[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = i0 + u * 1024, row = idx / CH, ch = idx % CH;
#pragma unroll
            for (int t = 0; t < 4; ++t) v[u][t] = make_uint4(0u, 0u, 0u, 0u);
            vo[u] = make_uint4(0u, 0u, 0u, 0u);
            if (idx < KP * CH && row < P.L && !(MBX_ATTN_DBG & 4)) {
                const T* r3 = qbase + (size_t)row * rstride + ch * 8;
                v[u][0] = *reinterpret_cast<const uint4*>(r3);
                v[u][1] = *reinterpret_cast<const uint4*>(r3 + C);
                v[u][2] = *reinterpret_cast<const uint4*>(r3 + 2 * C);
                v[u][3] = *reinterpret_cast<const uint4*>(dobase + (size_t)row * ostride + ch * 8);
                vo[u] = *reinterpret_cast<const uint4*>(obase + (size_t)row * ostride + ch * 8);
                if (ch == 0) sl[u] = lse[(P.tok0 + (size_t)row * P.tstep) * H + P.h];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = i0 + u * 1024, row = idx / CH, ch = idx % CH;
            if (idx < KP * CH) {
                const int off = row * RSTR + ch * 16;
                *reinterpret_cast<uint4*>(qt + off) = v[u][0];
                *reinterpret_cast<uint4*>(kt + off) = v[u][1];
                *reinterpret_cast<uint4*>(vt + off) = v[u][2];
                *reinterpret_cast<uint4*>(dot_ + off) = v[u][3];
            }
        }
        AT_TS(1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {      // (every lane takes part in the DPP steps: rows past KP carry zeros)
            const int idx = i0 + u * 1024, row = idx / CH, ch = idx % CH;
            float x[4], y[4], dl = 0.f;
            Raw4b::unpack(make_uint2(v[u][3].x, v[u][3].y), x); Raw4b::unpack(make_uint2(vo[u].x, vo[u].y), y);
            dl = fmaf(x[0], y[0], fmaf(x[1], y[1], fmaf(x[2], y[2], fmaf(x[3], y[3], dl))));
            Raw4b::unpack(make_uint2(v[u][3].z, v[u][3].w), x); Raw4b::unpack(make_uint2(vo[u].z, vo[u].w), y);
            dl = fmaf(x[0], y[0], fmaf(x[1], y[1], fmaf(x[2], y[2], fmaf(x[3], y[3], dl))));
            dl += dpp_mov<0xB1>(dl, dl);                      // quad_perm [1,0,3,2]
            dl += dpp_mov<0x4E>(dl, dl);                      // quad_perm [2,3,0,1]
            if (CH == 8) dl += dpp_mov<0x141>(dl, dl);        // row_half_mirror: the other quad of the row's eight lanes
            if (ch == 0 && idx < KP * CH) {
                lse_s[row] = sl[u] * 1.44269504088896341f;    // base-2 units: p = 2^(s*c2 - lse2)
                del_s[row] = dl;
            }
        }
    }
    __syncthreads();
    AT_TS(2);

    const int nfr = (P.L + 31) / 32;
    const float c2 = scale * 1.44269504088896341f;
    // accumulators of both roles share registers: dQ role acc[0 .. HD/32), dK/dV role dK = acc[0 .. HD/32), dV = acc[HD/32 .. 2 HD/32)
    f32x16_t acc[2 * (HD / 32)];
#pragma unroll
    for (int a = 0; a < 2 * (HD / 32); ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
#ifndef MBX_ATTN_SWAP
#define MBX_ATTN_SWAP 1
#endif
    // (round 6, measured null: s_setprio 1 around the S / dP clusters, and around all four MFMA clusters of an iteration -- 0.8638 ->
    // 0.8643 / 0.8743 ms; the re-cut "dQ then dV | dK" with every operand row in registers: 0.937 ms -- profiles/r06_attn_trace.txt)
    // which waves take which role: the SIMD issues oldest wave first, and the dK + dV role is the long one (16 MFMAs per 32 x 32 block
    // against 12, its k / v operand rows re-read from LDS) -- it goes to the OLDER waves (A/B: -DMBX_ATTN_SWAP=0 is round 2's assignment)
    const bool qrole = MBX_ATTN_SWAP ? wave >= 8 : wave < 8;
    const int blk = wave & 7, row = blk * 32 + (lane & 31);     // this lane's query (dQ role) or key (dK + dV role)
    if (blk < nfr && !(MBX_ATTN_DBG & 1)) {                       // L <= 256 (check_attn_args): at most eight 32-row blocks, one wave of each role per block
        if (qrole) {
            // -------------------------------------------------------------- dQ   (lane = query)
            BReg<T, HD> qreg, doreg;
            qreg.load(reinterpret_cast<const T*>(qt + (size_t)row * RSTR), g, true);
            doreg.load(reinterpret_cast<const T*>(dot_ + (size_t)row * RSTR), g, true);
            const float delta = del_s[row], lq2 = lse_s[row];
            for (int f = 0; f < nfr; ++f) {
                f32x16_t s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
                MmaRows<T, HD>::run(kt, RSTR, 32 * f, qreg, lane, s);
                MmaRows<T, HD>::run(vt, RSTR, 32 * f, doreg, lane, dp);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lq2));
                    s[r] = p * (dp[r] - delta);          // dS / scale: the softmax scale goes onto the finished accumulator, once
                }
#pragma unroll
                for (int df = 0; df < HD / 32; ++df) MmaCols<T>::run(kt, RSTR, df * 32, f, s, lane, acc[df]);
            }
        } else {
            // -------------------------------------------------------------- dK, dV   (lane = key)
            int boff = row * RSTR + g * 16;
            for (int f = 0; f < nfr; ++f) {
                asm volatile("" : "+v"(boff));      // the k / v operand reads stay inside the loop (hoisted they cost 32 VGPRs)
                f32x16_t s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
                mma_rows_ldsb<HD>(qt, RSTR, 32 * f, kt + boff, lane, s);       // s[r]  <-> (query e(f,r,g), key = lane)
                mma_rows_ldsb<HD>(dot_, RSTR, 32 * f, vt + boff, lane, dp);    // dP
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int q0 = 32 * f + 8 * qd + 4 * g;
                    const float4 l4 = *reinterpret_cast<const float4*>(lse_s + q0);
                    const float4 d4 = *reinterpret_cast<const float4*>(del_s + q0);
                    const float la[4] = {l4.x, l4.y, l4.z, l4.w}, da[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * qd + e;
                        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -la[e]));
                        dp[r] = p * (dp[r] - da[e]);           // dS / scale
                        s[r] = p;                              // P
                    }
                }
#pragma unroll
                for (int df = 0; df < HD / 32; ++df) {
                    MmaCols<T>::run(dot_, RSTR, df * 32, f, s, lane, acc[HD / 32 + df]);   // dV^T += dO^T P
                    MmaCols<T>::run(qt, RSTR, df * 32, f, dp, lane, acc[df]);              // dK^T += Q^T dS
                }
            }
        }
        // dQ (one role) and dK (the other) both sit in acc[0 .. HD/32): the softmax scale their dS carries is applied here, in fp32, once per
        // accumulator element instead of once per dS element per key block (round 6: 16 multiplies less per wave and iteration)
#pragma unroll
        for (int a = 0; a < HD / 32; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] *= scale;
    }
    AT_TS(3);
#ifdef MBX_ATTN_TRACE
    if (g_attn_trace != nullptr && (threadIdx.x & 63) == 0) g_attn_trace[(size_t)gridDim.x * 9 + (size_t)blockIdx.x * 16 + (threadIdx.x >> 6)] = (long long)wall_clock64();
#endif
    __syncthreads();                       // every wave is done reading the tiles: they become the output staging area
    AT_TS(4);
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));         // indices for the epilogue are re-derived here, not carried through the loops (128-VGPR budget)
    const int g2 = (tid2 >> 5) & 1, row2 = blk * 32 + (tid2 & 31);
    const bool qrole2 = MBX_ATTN_SWAP ? wave >= 8 : wave < 8;
    if (blk < nfr) {
        if (STATS) {        // each gradient row fragment goes over its own original, whose row dots it takes first
            float p1 = 0.f, p2 = 0.f;
            if (qrole2) {
                store_rowfrag_dot<HD>(qt_row(qt, row2, RSTR), reinterpret_cast<const f32x16_t (&)[HD / 32]>(acc[0]), g2, vec, p1, p2);
            } else {
                store_rowfrag_dot<HD>(qt_row(kt, row2, RSTR), reinterpret_cast<const f32x16_t (&)[HD / 32]>(acc[0]), g2, vec + HD / 4, p1, p2);
                store_rowfrag_dot<HD>(qt_row(vt, row2, RSTR), reinterpret_cast<const f32x16_t (&)[HD / 32]>(acc[HD / 32]), g2, vec + 2 * (HD / 4), p1, p2);
            }
            p1 = wave_halves<WaveAdd>(p1);
            p2 = wave_halves<WaveAdd>(p2);
            if (g2 == 0 && row2 < P.L) {
                const size_t Mtot = (size_t)(nprob / H) * P.L;
                *reinterpret_cast<float2*>(st_part + ((size_t)(2 * P.h + (qrole2 ? 0 : 1)) * Mtot + P.tok0 + (size_t)row2 * P.tstep) * 2) = make_float2(p1, p2);
            }
        } else if (qrole2) {
            store_rowfrag<T, HD>(reinterpret_cast<T*>(qt + (size_t)row2 * RSTR), reinterpret_cast<const f32x16_t (&)[HD / 32]>(acc[0]), 1.0f, g2);
        } else {
            store_rowfrag<T, HD>(reinterpret_cast<T*>(kt + (size_t)row2 * RSTR), reinterpret_cast<const f32x16_t (&)[HD / 32]>(acc[0]), 1.0f, g2);
            store_rowfrag<T, HD>(reinterpret_cast<T*>(vt + (size_t)row2 * RSTR), reinterpret_cast<const f32x16_t (&)[HD / 32]>(acc[HD / 32]), 1.0f, g2);
        }
    }
    __syncthreads();
    AT_TS(5);
    // ---- copy out: eight lanes write one whole 128-byte row segment of dq / dk / dv per instruction (stored straight from the
    // accumulator layout an instruction covers 32 rows x 16 bytes: measured 0.61 ms of pure memory time per launch) ----
    T* obase3 = dqkv + P.tok0 * C3 + (size_t)P.h * HD;
    for (int i0 = tid2; i0 < KP * CH; i0 += 2048) {
        uint4 v[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = min(i0 + u * 1024, KP * CH - 1), off = (idx / CH) * RSTR + (idx % CH) * 16;
            v[u][0] = *reinterpret_cast<const uint4*>(qt + off);
            v[u][1] = *reinterpret_cast<const uint4*>(kt + off);
            v[u][2] = *reinterpret_cast<const uint4*>(vt + off);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = i0 + u * 1024, r = idx / CH, ch = idx % CH;
            if (idx < KP * CH && r < P.L && !(MBX_ATTN_DBG & 2)) {
                T* r3 = obase3 + (size_t)r * rstride + ch * 8;
                *reinterpret_cast<uint4*>(r3) = v[u][0];
                *reinterpret_cast<uint4*>(r3 + C) = v[u][1];
                *reinterpret_cast<uint4*>(r3 + 2 * C) = v[u][2];
            }
        }
    }
#ifdef MBX_ATTN_TRACE
    AT_TS(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AT_TS(7);
    if (g_attn_trace != nullptr && threadIdx.x == 0) {
        long long* const tr = g_attn_trace + (size_t)blockIdx.x * 9;
#pragma unroll
        for (int k = 0; k < 8; ++k) tr[k] = ats[k];
        tr[8] = (long long)__builtin_amdgcn_s_getreg(63492) | ((long long)__builtin_amdgcn_s_getreg(63508) << 32);
    }
#endif
}

// ================================================================================================
// host side
// ================================================================================================
static int check_attn_args(const char* who, int B, int T, int J, int H, int hd, int mode, int dtype) {
    MBX_CHECK_ARG(B > 0 && T > 0 && J > 0 && H > 0, "%s: bad shape", who);
    MBX_CHECK_ARG(hd == 32 || hd == 64, "%s: head dim %d unsupported (32 or 64)", who, hd);
    MBX_CHECK_ARG(mode == MBX_ATTN_SPATIAL || mode == MBX_ATTN_TEMPORAL, "%s: unknown mode %d", who, mode);
    MBX_CHECK_ARG(dtype == MBX_BF16 || dtype == MBX_F32, "%s: unknown dtype %d", who, dtype);
    const int L = mode == MBX_ATTN_SPATIAL ? J : T;
    MBX_CHECK_ARG(L <= 256, "%s: sequence length %d > 256 unsupported", who, L);
    return 0;
}

template <typename K>
static int set_lds(K kernel, size_t bytes, const char* who) {
    if (bytes > 160 * 1024) return mbx_set_error("%s: needs %zu bytes of LDS (> 160 KiB)", who, bytes);
    return bytes > 64 * 1024 ? mbx_set_dyn_lds(reinterpret_cast<const void*>(kernel), bytes, who) : 0;
}

// p in [0, 1): 0 = no dropout (the kernels without the mask code)
static int make_drop(const char* who, float p, uint64_t seed, MbxDrop& dr) {
    MBX_CHECK_ARG(p >= 0.f && p < 1.f, "%s: dropout rate %g outside [0, 1)", who, (double)p);
    dr.seed_lo = (uint32_t)seed;
    dr.seed_hi = (uint32_t)(seed >> 32);
    dr.thresh = p > 0.f ? (uint32_t)fminf(p * 4294967296.0f, 4294967295.0f) : 0u;
    dr.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    return 0;
}

template <typename T, int HD, bool SHARED, bool DROP>
static int launch_fwd(const void* qkv, void* o, float* lse, int Tn, int J, int H, float scale, int mode, int nprob, int KP, hipStream_t s,
                      const MbxDrop& dr) {
    constexpr bool IS_BF = sizeof(T) == 2;
    const size_t per = (size_t)2 * KP * rm_stride<T>(HD);
    const size_t shm = SHARED ? per : 4 * per;
    auto kern = attn_fwd_kernel<T, HD, SHARED, DROP>;
    if (set_lds(kern, shm, "attn_fwd")) return 1;
    const int grid = SHARED ? nprob : (nprob + 3) / 4;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(AttnBlock<SHARED>::THREADS), shm, s, (const T*)qkv, (T*)o, lse, Tn, J, H, scale, mode, nprob, KP, dr);
    MBX_LAUNCH_CHECK("attn_fwd");
    return 0;
}

static int attn_fwd_impl(const void* qkv, void* o, float* lse, int B, int T, int J, int H, int hd, float scale, int mode,
                         int dtype, void* stream, const MbxDrop& dr) {
    MBX_CHECK_ARG(qkv && o && lse, "attn_fwd: null pointer");
    MBX_CHECK_ARG(scale > 0.f, "attn_fwd: scale must be positive (the running max is taken over the raw scores), got %g", (double)scale);
    if (check_attn_args("attn_fwd", B, T, J, H, hd, mode, dtype)) return 1;
    const int L = mode == MBX_ATTN_SPATIAL ? J : T;
    const int nprob = mode == MBX_ATTN_SPATIAL ? B * T * H : B * J * H;
    const int KP = ((L + 31) / 32) * 32;
    const bool shared = KP > 32;
    hipStream_t s = (hipStream_t)stream;
#define MBX_FWD2(TT, HDV, DR)                                                                        \
    (shared ? launch_fwd<TT, HDV, true, DR>(qkv, o, lse, T, J, H, scale, mode, nprob, KP, s, dr)       \
            : launch_fwd<TT, HDV, false, DR>(qkv, o, lse, T, J, H, scale, mode, nprob, KP, s, dr))
#define MBX_FWD(TT, HDV) (dr.thresh ? MBX_FWD2(TT, HDV, true) : MBX_FWD2(TT, HDV, false))
    if (dtype == MBX_BF16) return hd == 64 ? MBX_FWD(bf16_t, 64) : MBX_FWD(bf16_t, 32);
    return hd == 64 ? MBX_FWD(float, 64) : MBX_FWD(float, 32);
#undef MBX_FWD
#undef MBX_FWD2
}
extern "C" int mbx_attn_fwd(const void* qkv, void* o, float* lse, int B, int T, int J, int H, int hd, float scale, int mode,
                            int dtype, void* stream) {
    MbxDrop dr;
    make_drop("attn_fwd", 0.f, 0, dr);
    return attn_fwd_impl(qkv, o, lse, B, T, J, H, hd, scale, mode, dtype, stream, dr);
}
// mbx_attn_fwd with nn.Dropout(p) on the probabilities (DSTformer.py:96,182,196): o = (mask P / (1 - p)) V, lse of the undropped
// softmax; the mask is dropmask.keep(flat index in the reference's attn tensor, p, seed)
extern "C" int mbx_attn_fwd_drop(const void* qkv, void* o, float* lse, int B, int T, int J, int H, int hd, float scale, int mode,
                                 int dtype, float p, uint64_t seed, void* stream) {
    MbxDrop dr;
    if (make_drop("attn_fwd_drop", p, seed, dr)) return 1;
    return attn_fwd_impl(qkv, o, lse, B, T, J, H, hd, scale, mode, dtype, stream, dr);
}

template <typename T, int HD, bool SHARED, bool DROP>
static int launch_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int Tn, int J, int H,
                      float scale, int mode, int nprob, int KP, hipStream_t s, const MbxDrop& dr, void* dq_lo) {
    constexpr bool IS_BF = sizeof(T) == 2;
    const int RSTR = rm_stride<T>(HD);
    const size_t per_dq = (size_t)2 * KP * RSTR;
    const size_t per_dkv = (size_t)2 * KP * RSTR + 2 * KP * 4;
    const int grid = SHARED ? nprob : (nprob + 3) / 4;
    auto k1 = attn_bwd_dq_kernel<T, HD, SHARED, DROP>;
    auto k2 = attn_bwd_dkv_kernel<T, HD, SHARED, DROP>;
    const size_t shm1 = SHARED ? per_dq : 4 * per_dq, shm2 = SHARED ? per_dkv : 4 * per_dkv;
    if (set_lds(k1, shm1, "attn_bwd_dq") || set_lds(k2, shm2, "attn_bwd_dkv")) return 1;
    hipLaunchKernelGGL(k1, dim3(grid), dim3(AttnBlock<SHARED>::THREADS), shm1, s, (const T*)qkv, (const T*)o, (const T*)d_o, lse, (T*)dqkv, Tn, J, H, scale, mode, nprob, KP, dr, (bf16_t*)dq_lo);
    MBX_LAUNCH_CHECK("attn_bwd_dq");
    hipLaunchKernelGGL(k2, dim3(grid), dim3(AttnBlockKV<SHARED>::THREADS), shm2, s, (const T*)qkv, (const T*)o, (const T*)d_o, lse, (T*)dqkv, Tn, J, H, scale, mode, nprob, KP, dr, (bf16_t*)dq_lo);
    MBX_LAUNCH_CHECK("attn_bwd_dkv");
    return 0;
}

#ifdef MBX_ATTN_TRACE
#define MBX_ATTN_TRACE_SET(s_) do { static long long* const tb = [] { const char* e = getenv("MBX_TRACE_BUF"); return e ? (long long*)strtoull(e, nullptr, 0) : (long long*)nullptr; }(); \
                                    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_attn_trace), &tb, sizeof(tb), 0, hipMemcpyHostToDevice, (s_)); } while (0)
#else
#define MBX_ATTN_TRACE_SET(s_) do { } while (0)
#endif
static int attn_bwd_impl(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int B, int T, int J,
                         int H, int hd, float scale, int mode, int dtype, void* stream, const float* st_bias, const float* st_rsum,
                         float* st_part, const MbxDrop& dr, void* dq_lo = nullptr) {
    MBX_CHECK_ARG(qkv && o && d_o && lse && dqkv, "attn_bwd: null pointer");
    MBX_CHECK_ARG(!dq_lo || dtype == MBX_F32, "attn_bwd: operand planes are an output form of the fp32 kernels");
    MBX_CHECK_ARG(!(st_part && dr.thresh), "attn_bwd: the row dots of the folded LayerNorm backward and dropout do not combine");
    if (check_attn_args("attn_bwd", B, T, J, H, hd, mode, dtype)) return 1;
    const int L = mode == MBX_ATTN_SPATIAL ? J : T;
    const int nprob = mode == MBX_ATTN_SPATIAL ? B * T * H : B * J * H;
    const int KP = ((L + 31) / 32) * 32;
    const bool shared = KP > 32;
    hipStream_t s = (hipStream_t)stream;
#ifndef MBX_ATTN_BWD_TWO_KERNELS      // A/B builds only (tools/build_variants.py): force the dQ + dK/dV kernel pair
    if (shared && dtype == MBX_BF16 && !dr.thresh) {   // dropout: the dQ + dK/dV pair below (this kernel sits at its 128-VGPR budget: the mask hash spills)
        const size_t shm = (size_t)4 * KP * rm_stride<bf16_t>(hd) + 2 * KP * 4 + 3 * hd * 4;      // + the row-dot vectors (stats variant)
        if (shm <= 160 * 1024) {
#define MBX_BWD_FUSED(HDV)                                                                                            \
    do {                                                                                                              \
        auto k = st_part ? attn_bwd_fused_kernel<HDV, true> : attn_bwd_fused_kernel<HDV, false>;                     \
        if (set_lds(k, shm, "attn_bwd_fused")) return 1;                                                              \
        MBX_ATTN_TRACE_SET(s);                                                                                        \
        hipLaunchKernelGGL(k, dim3(nprob), dim3(1024), shm, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)d_o, lse, \
                           (bf16_t*)dqkv, T, J, H, scale, mode, nprob, KP, st_bias, st_rsum, st_part);                \
        MBX_LAUNCH_CHECK("attn_bwd_fused");                                                                           \
        return 0;                                                                                                     \
    } while (0)
            if (hd == 64) MBX_BWD_FUSED(64); else MBX_BWD_FUSED(32);
#undef MBX_BWD_FUSED
        }
    }
#endif
    if (!shared) {
#define MBX_BWD_SMALL2(TT, HDV, DR)                                                                                   \
    do {                                                                                                              \
        const size_t shm = (size_t)4 * (4 * 32 * rm_stride<TT>(HDV) + 2 * 32 * 4);                                    \
        auto k = attn_bwd_small_kernel<TT, HDV, DR>;                                                                  \
        if (set_lds(k, shm, "attn_bwd_small")) return 1;                                                              \
        hipLaunchKernelGGL(k, dim3((nprob + 3) / 4), dim3(256), shm, s, (const TT*)qkv, (const TT*)o, (const TT*)d_o, lse, \
                           (TT*)dqkv, T, J, H, scale, mode, nprob, st_bias, st_rsum, st_part, dr, (bf16_t*)dq_lo);   \
        MBX_LAUNCH_CHECK("attn_bwd_small");                                                                           \
        return 0;                                                                                                     \
    } while (0)
#define MBX_BWD_SMALL(TT, HDV) do { if (dr.thresh) MBX_BWD_SMALL2(TT, HDV, true); else MBX_BWD_SMALL2(TT, HDV, false); } while (0)
        if (dtype == MBX_BF16) { if (hd == 64) MBX_BWD_SMALL(bf16_t, 64); else MBX_BWD_SMALL(bf16_t, 32); }
        else { if (hd == 64) MBX_BWD_SMALL(float, 64); else MBX_BWD_SMALL(float, 32); }
#undef MBX_BWD_SMALL
#undef MBX_BWD_SMALL2
    }
    MBX_CHECK_ARG(!st_part, "attn_bwd_stats: this shape runs the two-kernel backward, which has no row-dot output");
#define MBX_BWD2(TT, HDV, DR)                                                                                         \
    (shared ? launch_bwd<TT, HDV, true, DR>(qkv, o, d_o, lse, dqkv, T, J, H, scale, mode, nprob, KP, s, dr, dq_lo)    \
            : launch_bwd<TT, HDV, false, DR>(qkv, o, d_o, lse, dqkv, T, J, H, scale, mode, nprob, KP, s, dr, dq_lo))
#define MBX_BWD(TT, HDV) (dr.thresh ? MBX_BWD2(TT, HDV, true) : MBX_BWD2(TT, HDV, false))
    if (dtype == MBX_BF16) return hd == 64 ? MBX_BWD(bf16_t, 64) : MBX_BWD(bf16_t, 32);
    return hd == 64 ? MBX_BWD(float, 64) : MBX_BWD(float, 32);
#undef MBX_BWD
#undef MBX_BWD2
}

extern "C" int mbx_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int B, int T, int J,
                            int H, int hd, float scale, int mode, int dtype, void* stream) {
    MbxDrop dr;
    make_drop("attn_bwd", 0.f, 0, dr);
    return attn_bwd_impl(qkv, o, d_o, lse, dqkv, B, T, J, H, hd, scale, mode, dtype, stream, nullptr, nullptr, nullptr, dr);
}
// fp32 kernels with dq / dk / dv written as the operand planes of the bf16x3 split (dqkv_hi, dqkv_lo bf16 [M, 3C]); p = 0: no dropout
extern "C" int mbx_attn_bwd_planes(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv_hi, void* dqkv_lo, int B,
                                   int T, int J, int H, int hd, float scale, int mode, float p, uint64_t seed, void* stream) {
    MBX_CHECK_ARG(dqkv_lo, "attn_bwd_planes: null pointer");
    MbxDrop dr;
    if (make_drop("attn_bwd_planes", p, seed, dr)) return 1;
    return attn_bwd_impl(qkv, o, d_o, lse, dqkv_hi, B, T, J, H, hd, scale, mode, MBX_F32, stream, nullptr, nullptr, nullptr, dr, dqkv_lo);
}
// backward of mbx_attn_fwd_drop (same p and seed): dP = mask (dO V^T) / (1 - p), dV from the dropped probabilities
extern "C" int mbx_attn_bwd_drop(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int B, int T, int J,
                                 int H, int hd, float scale, int mode, int dtype, float p, uint64_t seed, void* stream) {
    MbxDrop dr;
    if (make_drop("attn_bwd_drop", p, seed, dr)) return 1;
    return attn_bwd_impl(qkv, o, d_o, lse, dqkv, B, T, J, H, hd, scale, mode, dtype, stream, nullptr, nullptr, nullptr, dr);
}
// mbx_attn_bwd (bf16) + part[2H][M][2] = { sum dqkv rsum, sum dqkv (qkv - bias_f) } per (head, q | k+v columns, token); rsum, bias_f [3C] f32
extern "C" int mbx_attn_bwd_stats(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, const float* bias_f,
                                  const float* rsum, float* part, int B, int T, int J, int H, int hd, float scale, int mode,
                                  void* stream) {
    MBX_CHECK_ARG(bias_f && rsum && part, "attn_bwd_stats: null pointer");
    MBX_CHECK_ARG((reinterpret_cast<uintptr_t>(part) & 7) == 0, "attn_bwd_stats: part must be 8-byte aligned");
    MbxDrop dr;
    make_drop("attn_bwd_stats", 0.f, 0, dr);
    return attn_bwd_impl(qkv, o, d_o, lse, dqkv, B, T, J, H, hd, scale, mode, MBX_BF16, stream, bias_f, rsum, part, dr);
}
