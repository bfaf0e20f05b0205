// "Row owner" NT GEMMs: out[M, N] = epilogue(a[M, K] . w[N, K]^T) for the Linear layers of a Block (reference
// lib/model/DSTformer.py:97 qkv, :103 proj, :69-70 fc1 / fc2 and their input gradients), in the form the fused MLP kernel
// (mlp_fused.hip) proved: a workgroup = 4 waves x 32 token rows, every wave multiplies ALL weight fragments with its own rows, the
// weights arrive as a pre-packed sequence of 1-KiB MFMA fragments through an LDS ring filled by LDS-DMA (whole-KiB requests, one
// `ds_read_b128 base + imm` per MFMA, no address arithmetic, no bank conflicts), the MFMAs are asm statements in a hand-pipelined
// slot sequence.  The 256 x 256 / 256 x 128 tile kernels of gemm_pipe.hip stage BOTH operands through LDS as row-major 64-byte
// pieces (half-line DMA requests: the request rate of the vector memory path bounds them, DESIGN.md "what the 256x128 kernel is
// actually bound by"); here the token operand never touches LDS.
//
//   K-resident form (rows_nk_kernel, K in {256, 512}: qkv, fc1, dX of fc2, proj):
//     the wave's 32 x K slice of `a` sits in registers as K/16 operand fragments (K/4 VGPRs), loaded once, straight from global
//     memory in fragment layout (lane (i, g): 16 bytes of row i at k = 16 s + 8 g);  N is walked in chunks of 64 output columns:
//     tile tn = 0 (32 columns) over all k-steps, then tn = 1; the epilogue of a finished tile (bias / row constants, bf16, two
//     16-byte stores per lane) runs in the MFMA shadow of the next one.  ~230 registers, 64 KiB ring + 8 N bytes of LDS:
//     TWO workgroups per CU, so one's prologue (32 loads per lane) hides under the other's MFMAs.
//     The packer permutes the weight rows inside a chunk so that lane (i, g) of the accumulators ends up with the 32 CONSECUTIVE
//     output columns 64 c + 32 g + [0, 32) of token i: a lane's stores are 16-byte pieces of one 64-byte run, lanes i and i + 32
//     complete a 128-byte line.
#include "mbx_common.h"
#include "lds_stream.h"

#ifndef MBX_ROWS_DBG
#define MBX_ROWS_DBG 0      // ablation bits of diagnostic builds (timing only): 1 no epilogue, 2 no LDS-DMA, 4 no fragment reads, 16 no MFMAs
#endif
static constexpr int R_BM = 128;               // token rows per workgroup
static constexpr int R_SL = 8;                 // MFMA slots (= weight fragments) per ring stage
static constexpr int R_STAGE = R_SL * 1024;
static constexpr int R_NRS = 8;                // ring stages
static constexpr int R_RING = R_NRS * R_STAGE; // 64 KiB
static constexpr int R_CH = 64;                // output columns per chunk
static constexpr int R_TB = 4 * 1024;          // one 8-row x 128-byte store image per wave

// ---- packed weight stream, K-resident form: chunk c (output columns [64 c, 64 c + 64)) = 2 K/16 fragments, fragment (tn, s) at
// index tn K/16 + s; lane (i, g) of a fragment owns bytes [16 l, 16 l + 16) = w[row(c, tn, i)][16 s + 8 g + t], t = 0..7, with
// row(c, tn, i) = 64 c + 32 tn + 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3): accumulator register 4 q + e of lane (., g) is tile row
// 8 q + 4 g + e, so that lane holds the 16 consecutive columns 64 c + 32 tn + 16 g + 4 q + e.
__global__ __launch_bounds__(256) void rows_pack_nk_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int N, int K) {
    const int frag = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int KS = K / 16, per_chunk = 2 * KS;
    if (frag >= N / R_CH * per_chunk) return;
    const int c = frag / per_chunk, f = frag % per_chunk, tn = f / KS, s = f % KS;
    const int i = lane & 31, g = lane >> 5;
    const int row = R_CH * c + 32 * tn + 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3);
    *reinterpret_cast<uint4*>(out + (size_t)frag * 512 + lane * 8) = *reinterpret_cast<const uint4*>(w + (size_t)row * K + 16 * s + 8 * g);
}

enum { ROWS_EPI_STORE = 0, ROWS_EPI_STORE_LN = 1 };

// s_waitcnt vmcnt(n) with n known only after unrolling (an asm "n" operand must be a constant expression at parse time)
__device__ __forceinline__ void rows_vmwait(int n) {
    switch (n) {
#define RW_(v_) case v_: asm volatile("s_waitcnt vmcnt(" #v_ ")" ::: "memory"); break;
        RW_(0) RW_(1) RW_(2) RW_(3) RW_(4) RW_(5) RW_(6) RW_(7) RW_(8) RW_(9) RW_(10) RW_(11) RW_(12) RW_(13) RW_(14) RW_(15) RW_(16)
#undef RW_
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// Vector memory operations a wave issues between the second DMA piece of stage q - 6 (slot 7 of that stage) and the wait in slot 4 of
// stage q, q at position p of a tile of 4 stages (K = 512): the ten DMA pieces of stages q - 5 .. q - 1 (slots 5 and 7 of every stage)
// and the stores inside that window (DMA before store within a slot): the two stores of the epilogue that runs under this tile
// (tile slots 9 and 11, `own`) and the two of the one that ran under the tile before (`prev`).
__host__ __device__ constexpr int rows_vm_window(int p, bool own, bool prev) {
    int n = 10;
    const int lo = 2 * (8 * (p - 6) + 7), hi = 2 * (8 * p + 4);
    for (int ts = 9; ts <= 11; ts += 2) {
        if (own && 2 * ts + 1 > lo && 2 * ts + 1 < hi) ++n;
        if (prev && 2 * (ts - 32) + 1 > lo && 2 * (ts - 32) + 1 < hi) ++n;
    }
    return n;
}

// Diagnostic builds only (-DMBX_ROWS_TRACE, tools/rows_trace.py): 8 int64 per workgroup -- s_memrealtime (100 MHz) at entry, after the
// prologue (operand in registers, first stages landed), after chunk 0, at half of the chunks, after the loop, after the last epilogue
// with its stores acknowledged; the hardware id; the shader cycles of the whole workgroup.
#ifdef MBX_ROWS_TRACE
__device__ long long* g_rows_trace;
#define RS_STAMP(slot_) do { tsr[slot_] = (long long)wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)      // (kept in scalar registers until the end)
#else
#define RS_STAMP(slot_) do { } while (0)
#endif
template <int K, int EPI, bool FROMX>
__global__ __launch_bounds__(256, 2) void rows_nk_kernel(const void* __restrict__ a, const char* __restrict__ wpk,
                                                         const float* __restrict__ bias, const float* __restrict__ rsum,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         bf16_t* __restrict__ out, float eps, int M, int N, int nfull, int parts) {
    constexpr int KS = K / 16;             // k-steps = operand fragments per wave = slots per 32-column tile
    constexpr int SPC = 2 * KS / R_SL;     // stages per chunk (K = 512: 8, K = 256: 4)
    constexpr int PF = 4;                  // weight fragments in flight ahead of the MFMA that consumes them
    extern __shared__ __attribute__((aligned(16))) char smem[];   // ring 64 KiB | 4 x 1 KiB store images | bias [N] | rsum [N]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
#ifdef MBX_ROWS_TRACE
    long long tsr[6] = {0, 0, 0, 0, 0, 0};
    const long long cyc0 = (long long)__builtin_readcyclecounter();
#endif
    RS_STAMP(0);
    // Work units: the first `nfull` workgroups (whole rounds of the chip: 2 per CU) take one 128-row tile each and all of N; the row
    // tiles of the last, partial round are cut into `parts` column ranges, one workgroup each, so that the round that would leave
    // most CUs idle (64 clips: 4.03 rounds of 512 tiles) shrinks to a fraction of a tile time.
    int tile = blockIdx.x, c0 = 0, c1 = N / R_CH;
    if ((int)blockIdx.x >= nfull) {
        const int r = blockIdx.x - nfull, per = c1 / parts;
        tile = nfull + r / parts;
        c0 = (r % parts) * per;
        c1 = c0 + per;
    }
    const int mw = tile * R_BM + 32 * wave;
    // rows past M repeat row M - 1: they compute, and store, exactly that row's values (benign, and every wave issues the same
    // number of vector memory instructions -- the counted waits below depend on it)
    const int row = min(mw + i, M - 1);
    char* const ring = smem;
    float* const bs = reinterpret_cast<float*>(smem + R_RING + R_TB);
    float* const rs = bs + N;
    const int q0 = c0 * SPC, NS = c1 * SPC;
    auto seq_off = [&](int qq) -> int { return min(q0 + qq, NS - 1) * R_STAGE; };   // past the end: a harmless re-read of the last stage
    // piece d (0, 1) of a stage = its fragments 4 d .. 4 d + 3, one per wave
    const unsigned wvo = wave * 1024 + lane * 16;
    const unsigned dl = (unsigned)(uintptr_t)(const lds_void_t*)ring + wave * 1024;
#define RS_ISSUE1(src_, slot_, d_) glds16_s((src_) + (d_) * 4096, wvo, dl + (slot_) * R_STAGE + (d_) * 4096)
    f32x16_t acc[2];
    u32x4_t fb[8];
    const unsigned fr = (unsigned)(uintptr_t)(const lds_void_t*)ring + lane * 16;
    const unsigned bl = (unsigned)(uintptr_t)(const lds_void_t*)smem + R_RING + R_TB + 64 * g;    // this lane's bias run of tile 0 of chunk 0

    // ---- the first seven stages of the weight stream; stage q + 7 is requested in stage q, behind its barrier
#pragma unroll
    for (int st = 0; st < R_NRS - 1; ++st) {
        const char* const sp = wpk + seq_off(st);
        RS_ISSUE1(sp, st, 0);
        RS_ISSUE1(sp, st, 1);
    }
    for (int k = tid; k < N; k += 256) {
        bs[k] = bias != nullptr ? bias[k] : 0.f;
        if (EPI == ROWS_EPI_STORE_LN) rs[k] = rsum[k];
    }
    u32x4_t X[KS];
    float ln_rs = 1.f, ln_k = 0.f;         // raw-operand LayerNorm: out = rstd acc + (b' - rstd mean rsum)
    if constexpr (FROMX) {
        // the operand is bf16(x) rounded HERE from the fp32 rows of the residual stream, and the LayerNorm statistics are taken from
        // the same loads: lane (i, g) sees half of row i (shifted sums; the halves are joined below by Chan's formula)
        // The operand is bf16(x - x[row][0]): LayerNorm does not see a shift of its row, and rounding the SHIFTED values keeps the
        // rounding error proportional to the spread of the row instead of to its magnitude (a row with a large common offset would
        // otherwise lose mean^2 / var of its precision).  The shift is the row's first element -- the s = 0 value of lane
        // (i, 0), handed to lane (i, 1) by one permlane swap -- so both halves, and the row constants below, use the same one.
        // (Round 5, measured and dropped: these 2 K/16 loads per lane as asm statements in batches of eight with counted waits, 16 KiB
        // per wave in flight instead of the ~4 loads the compiler keeps in flight here -- the prologue took the same 27 us of the tile's
        // 128 (profiles/r05_rows_trace.txt): it is bound by the CU's miss bandwidth, which the other workgroup's weight stream shares.)
        const float* const ap = reinterpret_cast<const float*>(a) + (size_t)row * K + 8 * g;
        float xsh = 0.f, xs1 = 0.f, xs2 = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 lo = *reinterpret_cast<const float4*>(ap + 16 * s), hi = *reinterpret_cast<const float4*>(ap + 16 * s + 4);
            if (s == 0) xsh = wave_lower_half(lo.x);
            const float d[8] = {lo.x - xsh, lo.y - xsh, lo.z - xsh, lo.w - xsh, hi.x - xsh, hi.y - xsh, hi.z - xsh, hi.w - xsh};
            xs1 += ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
#pragma unroll
            for (int e = 0; e < 8; ++e) xs2 = fmaf(d[e], d[e], xs2);
            X[s] = u32x4_t{pack_bf2(d[0], d[1]), pack_bf2(d[2], d[3]), pack_bf2(d[4], d[5]), pack_bf2(d[6], d[7])};
        }
        constexpr float nh = (float)(K / 2);
        const float mean_h = xsh + xs1 / nh, m2_h = xs2 - xs1 * xs1 / nh;
        const float mean_o = wave_halves<WaveAdd>(mean_h) - mean_h;
        const float m2_both = wave_halves<WaveAdd>(m2_h);
        const float delta = mean_o - mean_h;
        const float var = fmaxf((m2_both + delta * delta * (nh * 0.5f)) / (float)K, 0.f);
        ln_rs = 1.0f / sqrtf(var + eps);
        ln_k = -ln_rs * (0.5f * (mean_h + mean_o) - xsh);      // the mean of the shifted row
    } else {
        const bf16_t* const ap = reinterpret_cast<const bf16_t*>(a) + (size_t)row * K + 8 * g;
#pragma unroll
        for (int s = 0; s < KS; ++s) X[s] = *reinterpret_cast<const u32x4_t*>(ap + 16 * s);
        if (EPI == ROWS_EPI_STORE_LN) { ln_rs = rstd[row]; ln_k = -ln_rs * mean[row]; }
    }
    // ONE drain for everything issued so far (the first DMA pieces, the biases, X), as a builtin: hipcc books its own loads as
    // landed here -- otherwise it puts a counted vmcnt in front of the first use of every X[s], and such a wait, blind to the DMA
    // statements issued in between, drains the weight stream.
    __builtin_amdgcn_s_waitcnt(0x0070);                         // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();                               // the first stages and the biases are in LDS
    RS_STAMP(1);
#pragma unroll
    for (int k = 0; k < PF; ++k) fb[k] = lds_read16(fr, k * 1024);

    // ---- epilogue of a finished tile (16 accumulator registers = the lane's 16 consecutive columns 32 tn + 16 g + 4 qq + e): four
    // micro-steps (bias / row constants, bf16) make eight packed dwords; then the wave's 32 rows x 64 bytes leave through its 1-KiB
    // LDS image, sixteen rows per round: the 32 lanes that own the rows write their two 16-byte pieces (piece p of row r at
    // p ^ (r >> 1)), all lanes read one piece back row-major, and ONE store instruction writes sixteen 64-byte half lines (the other
    // half of a line follows 32 slots later from the chunk's second tile and meets the first in L2).
    int ce = c0;                            // chunk of the tile being packed and stored
    uint32_t pkt[8];
    u32x4_t ebb = {0, 0, 0, 0}, err = {0, 0, 0, 0};             // bias / rsum of the next micro-step, read one slot ahead of their use
    u32x4_t rv = {0, 0, 0, 0};
    const int mbase = min(mw, M - 1), mlast = M - 1 - mbase;    // (wave-uniform) first row of the wave, clamped; last valid row offset
    char* const obase = reinterpret_cast<char*>(out + (size_t)mbase * N);
    const unsigned tbw = (unsigned)(uintptr_t)(const lds_void_t*)smem + R_RING + wave * 1024 + (i & 15) * 64;
    const unsigned tbr = (unsigned)(uintptr_t)(const lds_void_t*)smem + R_RING + wave * 1024 + (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 3) & 3)) << 4);
#define RS_ELOAD(tn_, qq_)                                                                                           \
    do {                                                                                                             \
        ebb = lds_read16(bl, (ce * R_CH + 32 * (tn_) + 4 * (qq_)) * 4);                                              \
        if (EPI == ROWS_EPI_STORE_LN) err = lds_read16(bl, (N + ce * R_CH + 32 * (tn_) + 4 * (qq_)) * 4);            \
    } while (0)
#define RS_ESTEP(tn_, qq_)                                                                                           \
    do {                                                                                                             \
        const u32x4_t bb_ = ebb, rr_ = err;                                                                          \
        if ((qq_) < 3) RS_ELOAD(tn_, (qq_) + 1);                                                                     \
        float v_[4];                                                                                                 \
        if (EPI == ROWS_EPI_STORE_LN) {                                                                              \
            _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_)                                                         \
                v_[e_] = fmaf(ln_rs, acc[tn_][4 * (qq_) + e_], fmaf(ln_k, __uint_as_float(rr_[e_]), __uint_as_float(bb_[e_]))); \
        } else {                                                                                                     \
            _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) v_[e_] = acc[tn_][4 * (qq_) + e_] + __uint_as_float(bb_[e_]); \
        }                                                                                                            \
        pkt[2 * (qq_)] = pack_bf2(v_[0], v_[1]);                                                                     \
        pkt[2 * (qq_) + 1] = pack_bf2(v_[2], v_[3]);                                                                 \
    } while (0)
    // round r_ (rows 16 r .. 16 r + 15): write piece j_ of the owners' two, read back row-major, store
#define RS_TW1(r_, j_)                                                                                               \
    do {                                                                                                             \
        if ((i >> 4) == (r_))                                                                                        \
            *reinterpret_cast<__attribute__((address_space(3))) u32x4_t*>(tbw + (((2 * g + (j_)) ^ ((i >> 1) & 3)) << 4)) = \
                u32x4_t{pkt[4 * (j_)], pkt[4 * (j_) + 1], pkt[4 * (j_) + 2], pkt[4 * (j_) + 3]};                     \
    } while (0)
// (the compiler barriers: other LANES wrote the image -- without them the read is merged with the previous round's, "the thread's own view")
#define RS_TR(r_) do { asm volatile("" ::: "memory"); rv = lds_read16(tbr, 0); asm volatile("" ::: "memory"); } while (0)
#define RS_TS(tn_, r_)                                                                                               \
    do {   /* wave-uniform base + a 32-bit lane offset: no 64-bit row pointers for the register allocator to keep around */ \
        const unsigned off_ = ((unsigned)min(16 * (r_) + (lane >> 2), mlast) * (unsigned)N + 8u * (lane & 3)) * 2u;  \
        if (!(MBX_ROWS_DBG & 64)) *reinterpret_cast<u32x4_t*>(obase + (size_t)(ce * R_CH + 32 * (tn_)) * 2 + off_) = rv; \
    } while (0)
#define MFMA_PAD1(a_) asm volatile("s_nop 7" : "+v"(a_))
    // Hook of tile slot ts_ while tile tn_ is multiplied: the tile before (accumulators 1 - tn_) is packed in slots 1..4 (constants
    // read one slot ahead), its rounds follow: slots 5, 6 write round 0 | 7 read 0, write | 8 write round 1 | 9 read 1, store 0 | 11 store 1
    // (LDS executes a wave's instructions in order: a read sees its round's writes and precedes the next round's).
#define RS_HOOK(tn_, ts_)                                                                                            \
    do {                                                                                                             \
        if (!(MBX_ROWS_DBG & 1)) {                                                                                   \
            if ((ts_) == 0) RS_ELOAD(1 - (tn_), 0);                                                                  \
            if ((ts_) == 1) MFMA_PAD1(acc[1 - (tn_)]);                                                               \
            if ((ts_) >= 1 && (ts_) <= 4) RS_ESTEP(1 - (tn_), (ts_) - 1);                                            \
            if (!(MBX_ROWS_DBG & 128)) {                                                                             \
                if ((ts_) == 9) RS_TS(1 - (tn_), 0);                                                                 \
                if ((ts_) == 11) RS_TS(1 - (tn_), 1);                                                                \
                if ((ts_) == 7) RS_TR(0);                                                                            \
                if ((ts_) == 9) RS_TR(1);                                                                            \
                if ((ts_) == 5) RS_TW1(0, 0);                                                                        \
                if ((ts_) == 6) RS_TW1(0, 1);                                                                        \
                if ((ts_) == 7) RS_TW1(1, 0);                                                                        \
                if ((ts_) == 8) RS_TW1(1, 1);                                                                        \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)

    // One tile = KS slots = KS / 8 stages.  Slot k of a stage: read the fragment of slot k + PF (from slot 4 on a fragment of the NEXT
    // stage: the barrier sits in front of slot 4), one MFMA, in slots 5 and 7 the two LDS-DMA pieces of stage q + 7 (into the ring
    // slot of stage q - 1, which everyone has left once the barrier of stage q is passed), the hook.  The wait in front of the barrier
    // ("stage q + 1 has landed") leaves the vector memory operations issued since that stage's pieces in flight: ten DMA pieces --
    // six stages of the stream -- and the stores among them, which therefore have ~45 slots to be acknowledged.  (K = 256: the window
    // spans three tiles; the stores are not counted, i.e. the wait is a few operations stronger than necessary.)
    int q = 0;
#define RS_TILE(tn_, HASPREV_, PREVST_)                                                                              \
    _Pragma("unroll") for (int p_ = 0; p_ < KS / R_SL; ++p_) {                                                       \
        unsigned st_ = fr + (q & 7) * R_STAGE, sn_ = fr + ((q + 1) & 7) * R_STAGE;                                   \
        asm volatile("" : "+v"(st_), "+v"(sn_));                                                                     \
        const char* const n7_ = wpk + seq_off(q + 7);                                                                \
        const int l7_ = (q + 7) & 7;                                                                                 \
        _Pragma("unroll") for (int k_ = 0; k_ < R_SL; ++k_) {                                                        \
            const int ts_ = R_SL * p_ + k_;                                                                          \
            if (k_ == 4) {                                                                                           \
                constexpr bool st_on_ = KS == 32 && !(MBX_ROWS_DBG & (1 | 64 | 128));                                \
                rows_vmwait((MBX_ROWS_DBG & 2) ? 0 : rows_vm_window(p_, st_on_ && (HASPREV_), st_on_ && (PREVST_))); \
                __builtin_amdgcn_sched_barrier(0);                                                                   \
                __builtin_amdgcn_s_barrier();                                                                        \
                __builtin_amdgcn_sched_barrier(0);                                                                   \
            }                                                                                                        \
            if (!(MBX_ROWS_DBG & 4))                                                                                 \
                fb[(k_ + PF) & 7] = k_ + PF < R_SL ? lds_read16(st_, (k_ + PF) * 1024) : lds_read16(sn_, (k_ + PF - R_SL) * 1024); \
            if (!(MBX_ROWS_DBG & 16) || ts_ == 0) {                                                                  \
                if (ts_ == 0) MFMA_FC1_Z(acc[tn_], fb[k_ & 7], X[0]);                                                \
                else MFMA_FC1(acc[tn_], fb[k_ & 7], X[ts_]);                                                         \
            }                                                                                                        \
            if ((k_ == 5 || k_ == 7) && !(MBX_ROWS_DBG & 2)) RS_ISSUE1(n7_, l7_, (k_ - 5) >> 1);                     \
            if (HASPREV_) RS_HOOK(tn_, ts_);                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
        ++q;                                                                                                         \
    }

    RS_TILE(0, false, false)                // chunk 0: tile 0 has no predecessor, the epilogue under tile 1 is the first
    RS_TILE(1, true, false)
    RS_STAMP(2);
    for (int c = c0 + 1; c < c1; ++c) {
#ifdef MBX_ROWS_TRACE
        if (c == (c0 + c1) / 2) RS_STAMP(3);
#endif
        ce = c - 1;
        RS_TILE(0, true, true)              // tile 1 of the previous chunk leaves
        ce = c;
        RS_TILE(1, true, true)              // tile 0 of this chunk
    }
    RS_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the re-read tail stages have landed (nothing may arrive in LDS after the workgroup ends)
    if (MBX_ROWS_DBG & 1) return;
    asm volatile("s_nop 15" : "+v"(acc[1]));
    ce = c1 - 1;
    RS_ELOAD(1, 0);
    RS_ESTEP(1, 0); RS_ESTEP(1, 1); RS_ESTEP(1, 2); RS_ESTEP(1, 3);
#pragma unroll
    for (int r = 0; r < 2; ++r) { RS_TW1(r, 0); RS_TW1(r, 1); RS_TR(r); RS_TS(1, r); }
#ifdef MBX_ROWS_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RS_STAMP(5);
    if (g_rows_trace != nullptr && threadIdx.x == 0) {
        long long* const tr = g_rows_trace + (size_t)blockIdx.x * 8;
#pragma unroll
        for (int k = 0; k < 6; ++k) tr[k] = tsr[k];
        tr[6] = (long long)__builtin_amdgcn_s_getreg(63492) | ((long long)__builtin_amdgcn_s_getreg(63508) << 32);
        tr[7] = (long long)__builtin_readcyclecounter() - cyc0;
    }
#endif
}

// ---- C ABI -------------------------------------------------------------------------------------------------------------------
extern "C" size_t mbx_rows_pack_bytes(int N, int K) { return (size_t)N * K * sizeof(bf16_t); }

extern "C" int mbx_rows_pack_nk(const void* w, void* packed, int N, int K, void* stream) {
    MBX_CHECK_ARG(w && packed, "rows_pack_nk: null pointer");
    MBX_CHECK_ARG((K == 256 || K == 512) && N > 0 && N % R_CH == 0, "rows_pack_nk: K=%d (256 or 512), N=%d (%% 64)", K, N);
    const int nfrag = N / R_CH * 2 * (K / 16);
    hipLaunchKernelGGL(rows_pack_nk_kernel, dim3((nfrag + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, (bf16_t*)packed, N, K);
    MBX_LAUNCH_CHECK("rows_pack_nk");
    return 0;
}

template <int K, int EPI, bool FROMX>
static int launch_rows_nk(const void* a, const void* packed, const float* bias, const float* rsum, const float* mean, const float* rstd,
                          void* out, float eps, int M, int N, hipStream_t s) {
    const size_t shm = R_RING + R_TB + (size_t)2 * N * sizeof(float);
    if (mbx_set_dyn_lds(reinterpret_cast<const void*>(rows_nk_kernel<K, EPI, FROMX>), shm, "rows_gemm_nk")) return 1;
    // whole rounds of the chip (two workgroups per CU) as whole tiles; the tiles of the last, partial round in column ranges
    const int slots = 2 * mbx_cu_count();
    const int tiles = (M + R_BM - 1) / R_BM, nch = N / R_CH;
    int nfull = tiles / slots * slots, parts = 1;
    for (int pp = 8; pp > 1; --pp)
        if (nch % pp == 0 && (tiles - nfull) * pp <= slots) { parts = pp; break; }
    if (parts == 1) nfull = tiles;
#ifdef MBX_ROWS_TRACE
    {
        static long long* const tb = [] { const char* e = getenv("MBX_TRACE_BUF"); return e ? (long long*)strtoull(e, nullptr, 0) : (long long*)nullptr; }();
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_rows_trace), &tb, sizeof(tb), 0, hipMemcpyHostToDevice, s);
    }
#endif
    hipLaunchKernelGGL((rows_nk_kernel<K, EPI, FROMX>), dim3(nfull + (tiles - nfull) * parts), dim3(256), shm, s, a, (const char*)packed, bias,
                       rsum, mean, rstd, (bf16_t*)out, eps, M, N, nfull, parts);
    MBX_LAUNCH_CHECK("rows_gemm_nk");
    return 0;
}

extern "C" int mbx_rows_gemm_nk(const void* a, const void* packed, const float* bias, const float* rsum, const float* mean,
                                const float* rstd, void* out, int M, int N, int K, void* stream) {
    MBX_CHECK_ARG(a && packed && out, "rows_gemm_nk: null pointer");
    MBX_CHECK_ARG(M > 0 && (K == 256 || K == 512) && N >= R_CH && N % R_CH == 0 && N <= 4096, "rows_gemm_nk: bad shape M=%d N=%d (%% 64) K=%d (256 or 512)", M, N, K);
    const bool ln = mean != nullptr;
    MBX_CHECK_ARG(!ln || (rstd && rsum), "rows_gemm_nk: the raw-operand form needs mean, rstd and rsum");
    hipStream_t s = (hipStream_t)stream;
    if (K == 512) return ln ? launch_rows_nk<512, ROWS_EPI_STORE_LN, false>(a, packed, bias, rsum, mean, rstd, out, 0.f, M, N, s)
                            : launch_rows_nk<512, ROWS_EPI_STORE, false>(a, packed, bias, rsum, mean, rstd, out, 0.f, M, N, s);
    return ln ? launch_rows_nk<256, ROWS_EPI_STORE_LN, false>(a, packed, bias, rsum, mean, rstd, out, 0.f, M, N, s)
              : launch_rows_nk<256, ROWS_EPI_STORE, false>(a, packed, bias, rsum, mean, rstd, out, 0.f, M, N, s);
}

extern "C" int mbx_rows_gemm_nk_ln(const float* x, const void* packed, const float* bias, const float* rsum, float eps, void* out,
                                   int M, int N, int K, void* stream) {
    MBX_CHECK_ARG(x && packed && out && rsum, "rows_gemm_nk_ln: null pointer");
    MBX_CHECK_ARG(M > 0 && (K == 256 || K == 512) && N >= R_CH && N % R_CH == 0 && N <= 4096, "rows_gemm_nk_ln: bad shape M=%d N=%d (%% 64) K=%d (256 or 512)", M, N, K);
    hipStream_t s = (hipStream_t)stream;
    if (K == 512) return launch_rows_nk<512, ROWS_EPI_STORE_LN, true>(x, packed, bias, rsum, nullptr, nullptr, out, eps, M, N, s);
    return launch_rows_nk<256, ROWS_EPI_STORE_LN, true>(x, packed, bias, rsum, nullptr, nullptr, out, eps, M, N, s);
}
