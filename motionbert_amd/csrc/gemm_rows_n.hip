// "N-resident" row-owner GEMMs (round 5): one loop, two epilogues that need COMPLETE rows of a 512-column output --
//   rows_n_lnbwd_kernel     mbx_rows_lnbwd_t    dX of a folded (LayerNorm -> Linear) pair + the LayerNorm backward        (this header)
//   rows_n_resid_ln_kernel  mbx_rows_resid_ln   proj / fc2 + residual + the NEXT LayerNorm's statistics and xhat           (further down)
//   rows_n_pack_many_kernel mbx_rows_n_pack_many the weight operand of both as a stream of 1-KiB MFMA fragments
//
// The LayerNorm-backward GEMM: the input gradient of a folded
// (LayerNorm -> Linear) pair INSIDE a Block (reference lib/model/DSTformer.py:241-249: norm1 -> attn.qkv :143, norm2 -> mlp.fc1 :80;
// the backward of these lines is autograd's in the reference),
//     dxhat = dY . W'                                   [M, 512] <- [M, K] x [K, 512],  K = 1536 (qkv) or 1024 (fc1)
//     dx    = dres + rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat))      (LayerNorm backward, gamma folded into W')
// with dres and dx travelling as bf16 (the gradient residual stream of include/mbx.h "mbx_gemm_nt_lnbwd_t").
//
// Why another GEMM shape.  The 256 x 128 tile kernel (gemm_pipe.hip) runs this product in four column tiles per row block, so the two
// row means of LayerNorm's backward had to come from the PRODUCERS of dY as row dots (round 3), and its N = 512 / long-K loop is the
// one shape the vendor library beats by 20 % (profiles/r04_gemm_vs_blas.txt).  Here a workgroup = 4 waves owns 128 COMPLETE rows:
// wave w keeps its 32 x 512 slice of dxhat in 256 accumulator registers (the fc2 half of the fused MLP, mlp_fused.hip), the weights
// stream as packed 1-KiB MFMA fragments through the 4 x 32 KiB LDS ring, the wave's token fragments (16 bytes per lane and k-step:
// row i, columns 16 s + 8 g of dY) are plain global loads issued four stages ahead, and the epilogue takes both row means from the
// accumulators themselves -- no row dots from the attention-backward / GELU' kernels, no row-constant launch.
// Skeleton measured first (tools/probes/mlp_shape_probe.hip, profiles/r05_mlp_shape_probe.txt): the bare product 0.40 ms at K = 1536 and
// 0.27 ms at K = 1024 (the tile kernel: 0.51 / 0.38 plain, 0.61 / 0.41 with this epilogue; the vendor's plain product 0.40 / 0.31).
//
// vmcnt discipline (loads and LDS-DMA share one in-order counter): stage q issues, in slot order, the weight pieces 2..7 of stage q + 3
// (slots 3..23), the two token fragments of stage q + 4 (slots 25, 29) and pieces 0, 1 of stage q + 4 (slots 27, 31).  At the barrier of
// stage q (slot 27) "stage q + 1 has landed" leaves the 21 younger operations in flight: (q-2: T P T P) + (q-1: 6 P, T, P, T, P) +
// (q: 6 P, T); everything older -- in particular the tokens of stage q + 1, issued in stage q - 3 -- has landed with it, so the token
// registers need no wait of their own.  The preamble issues the first eight tokens BEFORE the ring instead of interleaved: the first two
// stages of a trip wait for 17 / 19 instead of 21 (a slightly stronger wait on later trips, no branch).
// Round 6 -- the tail of the stream.  From slot 27 of stage 4 of the LAST trip on there is no stage q + 4 / q + 3 left to request.  Round 5
// re-read the last stage in those 26 slots (so that the counts stay constant); now they request the first 26 KiB per wave of the
// EPILOGUE's input -- xhat here, resid in the kernel further down -- straight into the ring slots the loop has left for good (slot s is
// free once the barrier of the stage that read it last has been passed, exactly as for a refill).  Same position in the in-order counter,
// so no counted wait changed.  It also took an in-order hazard out of the loop: round 5 fetched xhat with 32 register loads spread over
// the last trip, first-touch HBM loads that sat in the counter in FRONT of the weight pieces every barrier waits for -- the last trip
// ran 4 us slower than the others (profiles/r06_rn_trace.txt).
#include "mbx_common.h"
#include "lds_stream.h"

#ifndef MBX_RN_DBG
#define MBX_RN_DBG 0        // ablation bits of diagnostic builds (timing only, results wrong): 1 no epilogue, 2 no loop (prologue + epilogue only)
#endif
// Diagnostic builds only (-DMBX_RN_TRACE, tools/rn_trace.py): 12 int64 per workgroup of the LayerNorm-backward kernel (wave 0, lane 0) --
// s_memrealtime (100 MHz) at entry, first stage landed, end of the loop, after the drain + barrier, after xhat -> LDS, after pass 1,
// after each quarter of pass 2 (stores issued), stores acknowledged.  The buffer address comes from the environment variable MBX_TRACE_BUF.
#ifdef MBX_RN_TRACE
__device__ long long* g_rn_trace;
#define RN_TS(slot_) do { tsr[slot_] = (long long)wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)      // (kept in scalar registers until the end)
#else
#define RN_TS(slot_) do { } while (0)
#endif
static constexpr int RN_BM = 128;              // token rows per workgroup (4 waves x 32)
static constexpr int RN_STAGE = 32 * 1024;     // one ring stage = 32 fragments = KPS k-steps x NT column tiles
static constexpr int RN_RING = 4 * RN_STAGE;
// Geometry of the N = 32 NT output columns a wave owns completely.  NT = 16: dim_feat 512, 256 accumulator registers, a stage = 2 k-steps;
// NT = 8 (round 6): dim_feat 256 -- MotionBERT-Lite, configs/pretrain/MB_lite.yaml:18-24 -- 128 accumulator registers, a stage = 4 k-steps.
// A trip of the loop is always 16 k-steps (the token ring's indices are then static): 8 or 4 stages.
// The token fragments of a stage must be OLDER in the in-order counter than the last weight piece of that stage (requested in slot 23, two
// stages ahead): then the barrier's counted wait for the pieces covers them and the token registers need no wait of their own.  NT = 16:
// requested LA = 4 stages ahead, in slots 25 and 29; NT = 8: a stage has four k-steps, the ring of 16 holds four stages, so LA = 3 and the
// slots 1, 5, 9, 13 -- in FRONT of slot 23.  (Requested two stages ahead in the slots 17..29, as first built, the tokens are younger than
// the piece the barrier waits for: 4131 rows passed, 264,384 rows came back with a handful of wrong ones.)
template <int NT> struct RnGeo {
    static constexpr int N = 32 * NT, KPS = 32 / NT, TRIP = 16 / KPS, NQ = NT / 4;
    static constexpr int LA = NT == 16 ? 4 : 3, TS0 = NT == 16 ? 25 : 1;      // token lookahead in stages; slot of a stage's first token request
};
// Counted wait at the barrier of stage u of a trip (slot 27): "stage q + 1 has landed" leaves the operations issued behind its last piece
// (slot 23 of stage q - 2) in flight -- every stage issues 8 pieces (P) and, unless the stage it fetches for lies behind the LAST trip, KPS
// token fragments (T) in the slots ts0, ts0 + 4, ....  The first two stages of EVERY trip use the counts of the first trip, whose
// predecessors are the preamble's tokens, 24 P + 2 P (a slightly stronger wait on later trips, no branch).
__host__ __device__ constexpr int rn_cnt(int kps, int trip, int la, int ts0, bool last, int u) {
    int t_lo = 0, t_hi = 0;                        // the token requests of a stage in front of slot 27 / behind slot 23
    for (int j = 0; j < kps; ++j) { t_lo += ts0 + 4 * j < 27; t_hi += ts0 + 4 * j > 23; }
    const bool t2 = !last || u - 2 + la < trip, t1 = !last || u - 1 + la < trip, t0 = !last || u + la < trip;
    if (u == 0) return 8 + 2 + 6 + (t0 ? t_lo : 0);
    if (u == 1) return 2 + (8 + (t1 ? kps : 0)) + 6 + (t0 ? t_lo : 0);
    return (2 + (t2 ? t_hi : 0)) + (8 + (t1 ? kps : 0)) + (6 + (t0 ? t_lo : 0));
}
static_assert(rn_cnt(2, 8, 4, 25, false, 2) == 21 && rn_cnt(2, 8, 4, 25, true, 0) == 17 && rn_cnt(2, 8, 4, 25, true, 1) == 19 &&
              rn_cnt(2, 8, 4, 25, true, 3) == 21 && rn_cnt(2, 8, 4, 25, true, 4) == 20 && rn_cnt(2, 8, 4, 25, true, 5) == 18 &&
              rn_cnt(2, 8, 4, 25, true, 7) == 16, "the counts of round 5, by hand");
static_assert(rn_cnt(4, 4, 3, 1, false, 0) == 20 && rn_cnt(4, 4, 3, 1, false, 1) == 24 && rn_cnt(4, 4, 3, 1, false, 3) == 24 &&
              rn_cnt(4, 4, 3, 1, true, 0) == 20 && rn_cnt(4, 4, 3, 1, true, 1) == 20 && rn_cnt(4, 4, 3, 1, true, 2) == 16 &&
              rn_cnt(4, 4, 3, 1, true, 3) == 16, "the counts of the 8-tile geometry, by hand");

// packed stream: fragment (kk, nt) at index kk * NT + nt; lane (i, g) owns bytes [16 l, 16 l + 16) = w[32 nt + i][16 kk + 8 g + t],
// t = 0..7, w = the [32 NT, K] row-major operand of mbx_gemm_nt (for dX: the transposed folded weight W'^T)
// One launch packs many operands: record r of `desc` = {src bf16 [32 NT, K], dst, K} (3 x int64); blockIdx.y = record
__global__ __launch_bounds__(256) void rows_n_pack_many_kernel(const int64_t* __restrict__ desc, int NT) {
    const int64_t* d = desc + (size_t)blockIdx.y * 3;
    const bf16_t* w = reinterpret_cast<const bf16_t*>(d[0]);
    bf16_t* out = reinterpret_cast<bf16_t*>(d[1]);
    const int K = (int)d[2];
    const int frag = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (frag >= (K / 16) * NT) return;
    const int kk = frag / NT, nt = frag % NT, i = lane & 31, g = lane >> 5;
    *reinterpret_cast<uint4*>(out + (size_t)frag * 512 + lane * 8) = *reinterpret_cast<const uint4*>(w + (size_t)(32 * nt + i) * K + 16 * kk + 8 * g);
}

// LDS-DMA with the LDS address as (wave-uniform register + compile-time constant): one scalar register for all ring positions instead of
// one per position (the peeled last trip doubles their number, and an "s" operand that has been spilled to a VGPR does not assemble)
__device__ __forceinline__ void rn_glds(const char* base, unsigned voff, unsigned lds_base, int slot_piece) {
    switch (slot_piece) {      // ring slot * 8 + piece (an asm "n" operand must be a constant at parse time: the switch folds after unrolling)
#define RG_(p_) case p_: asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_base), "n"(p_ * 4096) : "memory", "scc"); break;
        RG_(0) RG_(1) RG_(2) RG_(3) RG_(4) RG_(5) RG_(6) RG_(7) RG_(8) RG_(9) RG_(10) RG_(11) RG_(12) RG_(13) RG_(14) RG_(15)
        RG_(16) RG_(17) RG_(18) RG_(19) RG_(20) RG_(21) RG_(22) RG_(23) RG_(24) RG_(25) RG_(26) RG_(27) RG_(28) RG_(29) RG_(30) RG_(31)
#undef RG_
        default: break;
    }
}
template <int N> __device__ __forceinline__ void rn_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
// the same with a count that is known only after unrolling (an asm "n" operand must be a constant at parse time: the switch folds)
__device__ __forceinline__ void rn_vmwait_n(int n) {
    switch (n) {
#define RW_(v_) case v_: asm volatile("s_waitcnt vmcnt(" #v_ ")" ::: "memory"); break;
        RW_(8) RW_(10) RW_(16) RW_(17) RW_(18) RW_(19) RW_(20) RW_(21) RW_(22) RW_(23) RW_(24) RW_(25) RW_(26) RW_(32)
#undef RW_
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;      // (a count the schedule does not produce: the strongest wait)
    }
}

template <int NT>
__global__ __launch_bounds__(256, 1) void rows_n_lnbwd_kernel(const bf16_t* __restrict__ dy, const char* __restrict__ wpk,
                                                             const bf16_t* __restrict__ xhat, const float* __restrict__ rstd,
                                                             const bf16_t* __restrict__ dres_t, bf16_t* __restrict__ dx_t, int M, int K) {
    constexpr int PF = 5;
    constexpr int RN_N = RnGeo<NT>::N, KPS = RnGeo<NT>::KPS, TRIP = RnGeo<NT>::TRIP, NQ = RnGeo<NT>::NQ, LA = RnGeo<NT>::LA, TS0 = RnGeo<NT>::TS0;
    extern __shared__ __attribute__((aligned(16))) char ring[];
#ifdef MBX_RN_TRACE
    long long tsr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    RN_TS(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
    const int mw = blockIdx.x * RN_BM + 32 * wave;
    const unsigned fr = (unsigned)(uintptr_t)(const lds_void_t*)ring + lane * 16;
    const unsigned wvo = wave * 1024 + lane * 16;
    const unsigned dl = (unsigned)(uintptr_t)(const lds_void_t*)ring + wave * 1024;
    const int nstages = K / (16 * KPS);
    // stage q of the stream (past the end: a harmless re-read of the last stage into a slot nobody reads again)
#define RN_ISSUE(q_, j_) glds16_s(wpk + (size_t)min((q_), nstages - 1) * RN_STAGE + (j_) * 4096, wvo, dl + ((q_) & 3) * RN_STAGE + (j_) * 4096)
    // token fragments: lane (i, g) = 16 bytes of row i at k = 16 s + 8 g; rows past M repeat row M - 1 (their results are never stored)
    // (round 6: a wave-uniform 64-bit base that moves from trip to trip + ONE 32-bit lane offset, instead of two 64-bit lane pointers --
    // three registers for a loop that has none to spare; the C entries check M K 2 < 2^32)
    const char* ap = reinterpret_cast<const char*>(dy);
    const unsigned aoff = ((unsigned)min(mw + i, M - 1) * (unsigned)K + 8u * g) * 2u;
    // xhat, the epilogue's first input, is requested DURING the loop, by LDS-DMA (round 6; round 5 loaded it into 128 registers in the
    // last trip and wrote them to LDS after the loop: 2 us per tile in situ, profiles/r06_rn_trace.txt).  Piece n = 8 j + r4 = rows
    // 4 r4 .. 4 r4 + 3 of quarter j (128 columns = 256 bytes per row); lane (xr, xp) fetches the 16-byte piece xp ^ (row & 15) of row
    // 4 r4 + xr, so that piece p of row r lands in slot p ^ (r & 15) of its row.  Pieces 26..31 (quarter 3, r4 = 2..7) go into the wave's
    // fifth buffer HERE -- the oldest vector memory operations of the kernel, so no counted wait below changes; pieces 0..25 take the 26
    // slots of the last trip in which the stream has nothing left to fetch (RN_TRIP) and land in the wave's KiB of ring slot n >> 3,
    // piece n & 7 -- each freed by the barrier in front of the slot that refills it.
    const int xr = lane >> 4, xp = lane & 15;
    unsigned xoff[8];                              // byte offset of (row 4 r4 + xr, piece xp ^ (row & 15) of quarter 0) in xhat / dres / dx
#pragma unroll
    for (int r4 = 0; r4 < 8; ++r4) xoff[r4] = (unsigned)min(mw + 4 * r4 + xr, M - 1) * (RN_N * 2) + ((xp ^ ((4 * r4 + xr) & 15)) << 4);
    if constexpr (8 * NQ > 26) {                   // (NT = 8: xhat is 16 KiB per wave, all of it fits the 26 slots)
        const unsigned bdl0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const lds_void_t*)ring + RN_RING + wave * 8192);
#pragma unroll
        for (int r4 = 2; r4 < 8; ++r4) glds16_s(reinterpret_cast<const char*>(xhat) + 3 * 256, xoff[r4], bdl0 + r4 * 1024);
    }
#define RN_DP_OK(n_) ((n_) < 8 * NQ)               /* (past xhat's last piece the slot re-reads the last stage, as in round 5) */
#define RN_DPIECE(n_, sp_) rn_glds(reinterpret_cast<const char*>(xhat) + ((n_) >> 3) * 256, xoff[(n_) & 7], dlu, sp_)
    u32x4_t tok[16];                               // fragment s lives in tok[s & 15]
// (the byte offset of the k-step goes into the scalar base -- two SALU instructions per load: inside a template an asm "n" operand is checked
// at instantiation and must be a constant expression there, which an index of an unrolled loop is not)
#define RN_TOK(dst_, off_) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst_) : "v"(aoff), "s"(ap + (off_)) : "memory")
#define RN_TOKN(dst_, off_) RN_TOK(dst_, (off_) + 512)      /* the tokens of the next trip's first half */
#pragma unroll
    for (int ks = 0; ks < KPS * LA; ++ks) RN_TOK(tok[ks], 32 * ks);        // the k-steps of the first LA stages
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) RN_ISSUE(q, j);
    RN_ISSUE(3, 0);
    RN_ISSUE(3, 1);
    f32x16_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        asm volatile("" : "+a"(acc[t]));            // zeroed in the accumulator file while the first loads are in flight
    }
    rn_vmwait<18>();                               // stage 0 (and, older, the first tokens) has landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    RN_TS(1);
    u32x4_t fb[8];
#pragma unroll
    for (int k = 0; k < PF; ++k) fb[k] = lds_read16(fr, k * 1024);
    // ---- the product: trips of TRIP stages = 16 k-steps (the token ring's indices are then static); K % 256 == 0.
    // Slot k of a stage: column tile k % NT, k-step k / NT of the stage's KPS.  The token fragments of k-step ks = KPS (u + LA) + j -- LA
    // stages ahead, RnGeo -- are requested in slot TS0 + 4 j of stage u: from this trip's rows while ks < 16, from the next trip's beyond
    // (not in the LAST trip: a register an asm load writes but nobody reads is dead to the compiler, which hands it to something else --
    // and the load lands in it later).
    const unsigned dlu = __builtin_amdgcn_readfirstlane(dl);      // (wave-uniform by construction; said again for the "s" operands below)
#define RN_KS(u_, k_) (KPS * ((u_) + LA) + ((k_) - TS0) / 4)      /* the k-step whose tokens slot k of stage u requests */
#define RN_TRIP(LAST_, DP_)                                                                                                 \
    {                                                                                                                \
        _Pragma("unroll") for (int u = 0; u < TRIP; ++u) {                                                           \
            const int q = q0 + u;                                                                                    \
            unsigned st = fr + (u & 3) * RN_STAGE, sn = fr + ((u + 1) & 3) * RN_STAGE;                               \
            asm volatile("" : "+v"(st), "+v"(sn));                                                                   \
            const char* const n3 = wpk + (size_t)min(q + 3, nstages - 1) * RN_STAGE;                                 \
            const char* const n4 = wpk + (size_t)min(q + 4, nstages - 1) * RN_STAGE;                                 \
            _Pragma("unroll") for (int k = 0; k < 32; ++k) {                                                         \
                if (k == 32 - PF) {                                                                                  \
                    rn_vmwait_n(rn_cnt(KPS, TRIP, LA, TS0, LAST_, u));                                               \
                    __builtin_amdgcn_sched_barrier(0);                                                               \
                    __builtin_amdgcn_s_barrier();                                                                    \
                    __builtin_amdgcn_sched_barrier(0);                                                               \
                }                                                                                                    \
                fb[(k + PF) & 7] = k + PF < 32 ? lds_read16(st, (k + PF) * 1024) : lds_read16(sn, (k + PF - 32) * 1024); \
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[k % NT]) : "v"(fb[k & 7]), "v"(tok[(KPS * u + k / NT) & 15])); \
                if ((k & 3) == 3) {                                                                                  \
                    /* LAST trip, from slot 27 of stage TRIP - 4 on: the stream has no stage q + 4 / q + 3 left, and the ring slot     \
                       such a piece would go to is never read again (it used to be a re-read of the last stage, so that the wait     \
                       counts stay constant).  DP_: piece n = 0..25 of the EPILOGUE's input goes there instead -- same position in   \
                       the in-order counter, so no counted wait changes -- into the wave's KiB of ring slot n >> 3, piece n & 7 */     \
                    if (LAST_ && DP_ && k < 24 && u >= TRIP - 3 && RN_DP_OK(8 * (u - (TRIP - 3)) + (k >> 2) + 2))                    \
                        RN_DPIECE(8 * (u - (TRIP - 3)) + (k >> 2) + 2, ((u + 3) & 3) * 8 + (k >> 2) + 2);                            \
                    else if (LAST_ && DP_ && k >= 24 && u >= TRIP - 4 && RN_DP_OK(8 * (u - (TRIP - 4)) + (k >> 2) - 6))              \
                        RN_DPIECE(8 * (u - (TRIP - 4)) + (k >> 2) - 6, ((u + 4) & 3) * 8 + (k >> 2) - 6);                            \
                    else if (k < 24) rn_glds(n3 + ((k >> 2) + 2) * 4096, wvo, dlu, ((u + 3) & 3) * 8 + (k >> 2) + 2);                \
                    else rn_glds(n4 + ((k >> 2) - 6) * 4096, wvo, dlu, ((u + 4) & 3) * 8 + (k >> 2) - 6);                             \
                }                                                                                                    \
                if ((k & 3) == 1 && k >= TS0 && k < TS0 + 4 * KPS) {                                                 \
                    if (u + LA < TRIP) RN_TOK(tok[RN_KS(u, k) & 15], 32 * (RN_KS(u, k) & 15));                       \
                    else if (!LAST_) RN_TOKN(tok[RN_KS(u, k) & 15], 32 * (RN_KS(u, k) & 15));                        \
                }                                                                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                                   \
            }                                                                                                        \
        }                                                                                                            \
        ap += 512;                                                                                                   \
    }
    int q0 = 0;
    if (!(MBX_RN_DBG & 2)) {
        for (; q0 + TRIP < nstages; q0 += TRIP) RN_TRIP(false, false)
        RN_TRIP(true, true)
    }
    RN_TS(2);
    // (no drain of the vector memory counter here: the pieces of xhat requested in the last stages are first-touch loads that are still
    // on their way; pass 1 below waits for them quarter by quarter, by count)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // every wave is done with the ring: 32 KiB of it per wave are buffers now
    RN_TS(3);
    if (MBX_RN_DBG & 1) return;
    // ONE pad for all sixteen tiles (the wait states between the last MFMA and the first non-MFMA reader of its result)
    if constexpr (NT == 16)
        asm volatile("s_nop 15" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]),
                     "+a"(acc[8]), "+a"(acc[9]), "+a"(acc[10]), "+a"(acc[11]), "+a"(acc[12]), "+a"(acc[13]), "+a"(acc[14]), "+a"(acc[15]));
    else
        asm volatile("s_nop 15" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]));
    // ---- epilogue.  xhat is on chip: piece n = 8 j + r4 in the wave's KiB of (ring slot n >> 3, piece n & 7) for n < 26 and in KiB r4 of
    // the wave's fifth buffer for n >= 26; inside a KiB row r takes 256 bytes, its 16-byte piece p sits at slot p ^ (r & 15), and a lane
    // reads its accumulator positions (row i, columns 32 ntl + 8 qq + 4 g + e of the quarter) as 8-byte halves of pieces.  The wave's other
    // 8 KiB -- ring slot 3, pieces 2..7, and KiBs 0, 1 of the fifth buffer -- take one quarter of dres at a time, by LDS-DMA, in the same
    // image (an earlier version fetched dres with asm loads into registers: under this epilogue's register pressure the compiler gave
    // all eight loads of a quarter ONE destination and copied it out before the data had landed -- an asm output is "defined" where the
    // statement stands).
    //   pass 1: the two row means from the accumulators and xhat;
    //   pass 2, per quarter: dx written over the xhat it was made from, read back row-major, stored as whole 256-byte row segments;
    //   the next quarter of dres is requested as soon as this one has been read, and lands while this one's dx is read out and stored.
    // (Measured and dropped in round 6, profiles/r06_rows_n_ablation.txt: dres requested during the loop as well / a whole quarter of
    // arithmetic ahead -- the epilogue is not waiting for its loads but for its own LDS round trips and ~2700 VALU instructions at one
    // wave per SIMD.)
    // (lane-derived values are re-derived from an opaque copy of the thread index: otherwise they are carried through the loop)
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, i_e = lane_e & 31, g_e = lane_e >> 5, xr_e = lane_e >> 4, xp_e = lane_e & 15;
    char* const rw = ring + wave * 1024;                              // the wave's KiB of (ring slot s, piece p): rw + s * 32768 + p * 4096
    char* const bd = ring + RN_RING + wave * 8192;
    const unsigned rwl = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const lds_void_t*)ring + wave * 1024);
    const unsigned bdl = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const lds_void_t*)ring + RN_RING + wave * 8192);
    // KiB (j, r4) of xhat / KiB c of the dres image, for compile-time indices: generic pointer, and LDS address for the DMA
#define RN_XCH(j_, r4_) (8 * (j_) + (r4_) < 26 ? rw + (j_) * 32768 + (r4_) * 4096 : bd + (r4_) * 1024)
#define RN_STG_L(c_) ((c_) < 2 ? bdl + (c_) * 1024 : rwl + 3 * 32768 + (c_) * 4096)
    // ... and for the lane's own row i (KiB i >> 2, row i & 3 of it) in the accumulator layout
    const int ck = i_e >> 2, inr = (i_e & 3) * 256 + 8 * g_e, sx = (i_e & 15) << 4;
    char* const xc_i = rw + ck * 4096 + inr;                          // + j * 32768 for quarters 0..2
    char* const xc3_i = (ck < 2 ? rw + 3 * 32768 + ck * 4096 : bd + ck * 1024) + inr;      // (quarter 3: NT = 16 only)
    const char* const dx_ = (ck < 2 ? bd + ck * 1024 : rw + 3 * 32768 + ck * 4096) + inr;
    // DMA instruction r4 of a quarter of dres: the same rows and pieces as of xhat
    unsigned doff[8];
#pragma unroll
    for (int r4 = 0; r4 < 8; ++r4) {
        const int rl = 4 * r4 + xr_e;
        doff[r4] = (unsigned)min(mw + rl, M - 1) * (RN_N * 2) + ((xp_e ^ (rl & 15)) << 4);
    }
#define RN_DDMA(j_) _Pragma("unroll") for (int r4_ = 0; r4_ < 8; ++r4_)                                              \
        glds16_s(reinterpret_cast<const char*>(dres_t) + (j_) * 256, doff[r4_], RN_STG_L(r4_))
    RN_DDMA(0);
    const float rs = rstd[min(mw + i_e, M - 1)];
    float c1 = 0.f, c2 = 0.f;
    RN_TS(4);
    // (the four reads of tile nt + 1 are issued in front of the arithmetic of tile nt.  One wave per SIMD means nothing else covers a
    // ds_read -> use pair: as written in round 5 every tile of both passes waited out its own LDS round trip, 228 s_waitcnt in 2700 VALU
    // instructions -- profiles/r06_rows_n_ablation.txt)
#define RN_XRD(nt_, qq_) (*reinterpret_cast<const uint2*>(((nt_) < 12 ? xc_i + ((nt_) >> 2) * 32768 : xc3_i) + (((((nt_) & 3) * 4 + (qq_)) << 4) ^ sx)))      /* (NT = 8: nt < 8) */
    uint2 xq[2][4];
    // Counted waits for xhat, in issue order  ... | n0 .. n7 | n8 .. n15 | n16 .. n23 | n24 n25 | D0 (8) [| rstd]: quarter j of xhat (pieces
    // 8 j .. 8 j + 7; 26..31 are the oldest operations of the kernel; NT = 8: the slots 16..25 re-read the last stage) has landed when at
    // most 26 / 18 / 10 / 8 younger operations are in flight.  (The compiler's load of rstd sits behind D0 in program order; if it is
    // there the waits are one stronger than needed.)
    rn_vmwait<26>();
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) xq[0][qq] = RN_XRD(0, qq);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if (nt == 3) rn_vmwait<18>(); else if (nt == 7) rn_vmwait<10>(); else if (nt == 11) rn_vmwait<8>();
        if (nt + 1 < NT) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) xq[(nt + 1) & 1][qq] = RN_XRD(nt + 1, qq);
        }
        const f32x16_t t = acc[nt];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const uint2 xv = xq[nt & 1][qq];
            const float x0 = __uint_as_float(xv.x << 16), x1 = __uint_as_float(xv.x & 0xffff0000u);
            const float x2 = __uint_as_float(xv.y << 16), x3 = __uint_as_float(xv.y & 0xffff0000u);
            c1 += (t[4 * qq] + t[4 * qq + 1]) + (t[4 * qq + 2] + t[4 * qq + 3]);
            c2 = fmaf(t[4 * qq], x0, fmaf(t[4 * qq + 1], x1, fmaf(t[4 * qq + 2], x2, fmaf(t[4 * qq + 3], x3, c2))));
        }
        __builtin_amdgcn_sched_barrier(0);                            // one tile's 16 registers at a time out of the accumulator file
    }
    c1 = wave_halves<WaveAdd>(c1) * (1.0f / (float)RN_N);             // the two halves of row i: lanes i and i + 32
    c2 = wave_halves<WaveAdd>(c2) * (1.0f / (float)RN_N);
    const float k1 = -rs * c1, k2 = -rs * c2;                         // dx = dres + rs t + k1 + k2 xhat
    RN_TS(5);
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        // vector memory operations younger than the DMA of quarter j: the stores of quarter j - 1 (8)
        if (j == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#define RN_DRD(ntl_, qq_) (*reinterpret_cast<const uint2*>(dx_ + ((((ntl_) * 4 + (qq_)) << 4) ^ sx)))
        uint2 xp2[2][4], dp2[2][4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) { xp2[0][qq] = RN_XRD(4 * j, qq); dp2[0][qq] = RN_DRD(0, qq); }
#pragma unroll
        for (int ntl = 0; ntl < 4; ++ntl) {
            if (ntl + 1 < 4) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) { xp2[(ntl + 1) & 1][qq] = RN_XRD(4 * j + ntl + 1, qq); dp2[(ntl + 1) & 1][qq] = RN_DRD(ntl + 1, qq); }
            }
            const f32x16_t t = acc[4 * j + ntl];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                uint2* const p = reinterpret_cast<uint2*>((j < 3 ? xc_i + j * 32768 : xc3_i) + (((ntl * 4 + qq) << 4) ^ sx));
                const uint2 xv = xp2[ntl & 1][qq];
                const uint2 dv = dp2[ntl & 1][qq];
                const float d0 = __uint_as_float(dv.x << 16), d1 = __uint_as_float(dv.x & 0xffff0000u);
                const float d2 = __uint_as_float(dv.y << 16), d3 = __uint_as_float(dv.y & 0xffff0000u);
                const float x0 = __uint_as_float(xv.x << 16), x1 = __uint_as_float(xv.x & 0xffff0000u);
                const float x2 = __uint_as_float(xv.y << 16), x3 = __uint_as_float(xv.y & 0xffff0000u);
                const float o0 = fmaf(rs, t[4 * qq], fmaf(k2, x0, d0 + k1)), o1 = fmaf(rs, t[4 * qq + 1], fmaf(k2, x1, d1 + k1));
                const float o2 = fmaf(rs, t[4 * qq + 2], fmaf(k2, x2, d2 + k1)), o3 = fmaf(rs, t[4 * qq + 3], fmaf(k2, x3, d3 + k1));
                *p = make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));  // over the xhat value it was made from: same lane, same place
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // dres has been read (its image is free), dx is in place
        if (j + 1 < NQ) { if (j == 0) { RN_DDMA(1); } else if (j == 1) { RN_DDMA(2); } else if (j == 2) { RN_DDMA(3); } }
#pragma unroll
        for (int r4 = 0; r4 < 8; ++r4) {
            const int rl = 4 * r4 + xr_e;
            const uint4 v = *reinterpret_cast<const uint4*>(RN_XCH(j, r4) + lane_e * 16);
            // rows past M were computed from row M - 1's inputs (every load is clamped) and are stored onto row M - 1: identical bytes,
            // and every wave issues the same number of vector memory instructions -- the counted waits above depend on it
            *reinterpret_cast<uint4*>(dx_t + (size_t)min(mw + rl, M - 1) * RN_N + j * 128 + ((xp_e ^ (rl & 15)) << 3)) = v;
        }
        RN_TS(6 + j);
    }
#ifdef MBX_RN_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RN_TS(10);
    if (g_rn_trace != nullptr && threadIdx.x == 0) {
        long long* const tr = g_rn_trace + (size_t)blockIdx.x * 12;
#pragma unroll
        for (int k = 0; k < 11; ++k) tr[k] = tsr[k];
        tr[11] = (long long)__builtin_amdgcn_s_getreg(63492) | ((long long)__builtin_amdgcn_s_getreg(63508) << 32);
    }
#endif
}

// ---- the same product as the FORWARD residual GEMM of a sub-layer that is followed by a LayerNorm (round 5):
//     y    = resid + a . w^T + bias                      fp32 [M, 512]   (proj / fc2 + residual, DSTformer.py:241-249)
//     xhat = (y - mean(y)) rstd(y)                       bf16 [M, 512]   (the next sub-layer's LayerNorm, plain normalisation: gamma and
//                                                                         beta live in the folded weights of the Linear it feeds)
// A wave owns whole rows, so the statistics of LayerNorm come from its own registers (two passes over the 256 accumulators, the
// arithmetic of ln_fwd_row in elementwise.hip) and the stand-alone LayerNorm launch -- which re-reads y from HBM -- disappears.
// The tile kernel could not do this without a reduction across its four column tiles (round 3 / 4: built, slower, removed).
// Epilogue: the wave's 32 KiB of the ring are two 16-KiB buffers for one 128-column
// quarter of resid each (fp32, by LDS-DMA, double-buffered: a row takes 512 bytes, its 16-byte piece p sits at slot p ^ row); y is
// written over the resid it was made from, read back row-major and stored as 512-byte row segments while the accumulators keep y;
// then mean and rstd, then xhat through the fifth 8-KiB buffer exactly as dx leaves the LayerNorm-backward kernel above.
// (Measured and dropped: the first two quarters of resid requested into registers during the last trip of the loop, as the xhat
// prefetch of the kernel above -- 0.380 -> 0.381 ms at K = 512, 0.465 -> 0.490 at K = 1024: across the CUs this kernel is bound by
// HBM bandwidth (mixed reads and writes at ~4.2 TB/s of algorithmic bytes), not by the latency of one tile's loads.)
template <int NT>
__global__ __launch_bounds__(256, 1) void rows_n_resid_ln_kernel(const bf16_t* __restrict__ a, const char* __restrict__ wpk,
                                                                const float* __restrict__ bias, const float* __restrict__ resid,
                                                                float* __restrict__ y, bf16_t* __restrict__ xhat_o,
                                                                float* __restrict__ mean_o, float* __restrict__ rstd_o, float eps, int M, int K) {
    constexpr int PF = 5;
    constexpr int RN_N = RnGeo<NT>::N, KPS = RnGeo<NT>::KPS, TRIP = RnGeo<NT>::TRIP, NQ = RnGeo<NT>::NQ, LA = RnGeo<NT>::LA, TS0 = RnGeo<NT>::TS0;
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
    const int mw = blockIdx.x * RN_BM + 32 * wave;
    const unsigned fr = (unsigned)(uintptr_t)(const lds_void_t*)ring + lane * 16;
    const unsigned wvo = wave * 1024 + lane * 16;
    const unsigned dl = (unsigned)(uintptr_t)(const lds_void_t*)ring + wave * 1024;
    const unsigned dlu = __builtin_amdgcn_readfirstlane(dl);
    const int nstages = K / (16 * KPS);
    const char* ap = reinterpret_cast<const char*>(a);
    const unsigned aoff = ((unsigned)min(mw + i, M - 1) * (unsigned)K + 8u * g) * 2u;
    // the wave's copy of the bias (8 floats per lane), requested first -- the oldest vector memory operations of the kernel, so no
    // counted wait below changes -- and parked in the wave's fifth buffer after the loop (the wait there carries the dependence)
    u32x4_t bq0, bq1;
    {
        const float* const bp = bias + (lane & (RN_N / 8 - 1)) * 8;      // (NT = 8: 256 floats, the upper half wave repeats the lower)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bq0) : "v"(bp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(bq1) : "v"(bp) : "memory");
    }
    u32x4_t tok[16];
#pragma unroll
    for (int ks = 0; ks < KPS * LA; ++ks) RN_TOK(tok[ks], 32 * ks);        // the k-steps of the first LA stages
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) RN_ISSUE(q, j);
    RN_ISSUE(3, 0);
    RN_ISSUE(3, 1);
    f32x16_t acc[NT];                              // acc[nt][4 qq + e] = out[row i][32 nt + 8 qq + 4 g + e]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        asm volatile("" : "+a"(acc[t]));
    }
    rn_vmwait<18>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    u32x4_t fb[8];
#pragma unroll
    for (int k = 0; k < PF; ++k) fb[k] = lds_read16(fr, k * 1024);
    // (round 6) the 26 slots of the last trip in which the stream has nothing left to fetch request the first 26 KiB of the epilogue's
    // input: KiB m = 16 b + d = rows 2 d, 2 d + 1 of quarter b of resid (fp32, 128 columns = 512 bytes per row; lane l fetches the piece
    // that belongs in slot l & 31 of row 2 d + (l >> 5): piece (l & 31) ^ row), into the wave's KiB of ring slot m >> 3, piece m & 7 --
    // all of quarter 0 and ten sixteenths of quarter 1.  (They used to re-read the last stage: a fifth of the stream's requests at K = 512.)
    unsigned doff[16];                             // byte offset of (row 2 d + g, piece i ^ row of quarter 0) in resid and y
#pragma unroll
    for (int d = 0; d < 16; ++d) doff[d] = (unsigned)min(mw + 2 * d + g, M - 1) * (RN_N * 4) + ((i ^ (2 * d + g)) << 4);
#undef RN_DPIECE
#undef RN_DP_OK
#define RN_DP_OK(n_) true
#define RN_DPIECE(n_, sp_) rn_glds(reinterpret_cast<const char*>(resid) + ((n_) >> 4) * 512, doff[(n_) & 15], dlu, sp_)
    int q0 = 0;
    if (!(MBX_RN_DBG & 2)) {
        for (; q0 + TRIP < nstages; q0 += TRIP) RN_TRIP(false, false)
        RN_TRIP(true, true)
    }
    // (no drain of the vector memory counter: the KiBs of resid requested in the last stages are still on their way; counted wait below.
    // The bias registers -- the oldest loads of the kernel, landed since the first counted wait of the loop -- take their data dependence here)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq0), "+v"(bq1) : : "memory");
    __builtin_amdgcn_s_barrier();                                    // every wave is done with the ring
    if (MBX_RN_DBG & 1) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) MFMA_PAD_A(acc[t]);

    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, i_e = lane_e & 31, g_e = lane_e >> 5, xr_e = lane_e >> 4, xp_e = lane_e & 15;
    char* const rw = ring + wave * 1024;                              // the wave's KiB of (ring slot s, piece p): rw + s * 32768 + p * 4096
    char* const bx = ring + RN_RING + wave * 8192;
    const unsigned rwl = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const lds_void_t*)ring + wave * 1024);
    // Two images of one quarter of resid each (double-buffered against the stores): KiB d (rows 2 d, 2 d + 1: a row takes 512 bytes, its
    // 16-byte piece p sits at slot p ^ row) of image b is the wave's KiB m = 16 b + d of the ring, i.e. of (ring slot m >> 3, piece m & 7).
    // DMA / read-out instruction d moves that KiB; the same byte offset (doff) addresses resid and y.
#define RN_RCH(b_, d_) (rw + ((16 * (b_) + (d_)) >> 3) * 32768 + ((16 * (b_) + (d_)) & 7) * 4096)
#define RN_RCH_L(b_, d_) (rwl + ((16 * (b_) + (d_)) >> 3) * 32768 + ((16 * (b_) + (d_)) & 7) * 4096)
#define RN_RDMA(j_, b_, d0_) _Pragma("unroll") for (int d_ = (d0_); d_ < 16; ++d_)                                   \
        glds16_s(reinterpret_cast<const char*>(resid) + (j_) * 512, doff_e[d_], RN_RCH_L(b_, d_))
    unsigned doff_e[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        const int rl = 2 * d + g_e;
        doff_e[d] = (unsigned)min(mw + rl, M - 1) * (RN_N * 4) + ((i_e ^ rl) << 4);
    }
    RN_RDMA(1, 1, 10);                                                // what the last trip could not fit of quarter 1: KiBs 26..31 of the ring
    // quarter 0 (KiBs 0..15 of the last trip's 26) has landed when at most the 10 KiBs behind it and these 6 are in flight
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    *reinterpret_cast<u32x4_t*>(bx + lane_e * 32) = bq0;              // bias[8 lane .. 8 lane + 7]: read below by every lane that holds those columns
    *reinterpret_cast<u32x4_t*>(bx + lane_e * 32 + 16) = bq1;
    // the lane's own row i in the accumulator layout: KiB i >> 1 of an image, row i & 1 of it
    char* const rb_i = rw + (i_e >> 4) * 32768 + ((i_e >> 1) & 7) * 4096 + (i_e & 1) * 512;      // + b * 65536
    const int sx = i_e << 4;
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        // vector memory operations younger than the last DMA of quarter j, in issue order  R1' (6) | S0 (16) R2 (16) | S1 R3 | S2 | S3
        // (NT = 8: R1' | S0 | S1): the stores of quarter j - 1 and, if there is one, the DMA of quarter j + 1.  Quarter 0: waited for above
        if (j >= 1) rn_vmwait_n(j + 1 < NQ ? 32 : 16);
        char* const rbj = rb_i + (j & 1) * 65536;
#define RN_RRD(ntl_, qq_) (*reinterpret_cast<const float4*>(rbj + (((8 * (ntl_) + 2 * (qq_) + g_e) << 4) ^ sx)))
#define RN_BRD(ntl_, qq_) (*reinterpret_cast<const float4*>(bx + (32 * (4 * j + (ntl_)) + 8 * (qq_) + 4 * g_e) * 4))
        // (the reads of tile ntl + 1 are issued in front of the arithmetic of tile ntl: one wave per SIMD, nothing else covers the round trip)
        float4 rq[2][4], bq[2][4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) { rq[0][qq] = RN_RRD(0, qq); bq[0][qq] = RN_BRD(0, qq); }
#pragma unroll
        for (int ntl = 0; ntl < 4; ++ntl) {
            if (ntl + 1 < 4) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) { rq[(ntl + 1) & 1][qq] = RN_RRD(ntl + 1, qq); bq[(ntl + 1) & 1][qq] = RN_BRD(ntl + 1, qq); }
            }
            f32x16_t t = acc[4 * j + ntl];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                float4* const p = reinterpret_cast<float4*>(rbj + (((8 * ntl + 2 * qq + g_e) << 4) ^ sx));
                const float4 r = rq[ntl & 1][qq], b = bq[ntl & 1][qq];
                t[4 * qq] += r.x + b.x; t[4 * qq + 1] += r.y + b.y; t[4 * qq + 2] += r.z + b.z; t[4 * qq + 3] += r.w + b.w;
                *p = make_float4(t[4 * qq], t[4 * qq + 1], t[4 * qq + 2], t[4 * qq + 3]);      // y over the resid it was made from
                s1 += (t[4 * qq] + t[4 * qq + 1]) + (t[4 * qq + 2] + t[4 * qq + 3]);
            }
            acc[4 * j + ntl] = t;                                     // the accumulators keep y for the statistics and xhat
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const float4 v = *reinterpret_cast<const float4*>(RN_RCH(j & 1, d) + lane_e * 16);
            // (rows past M carry row M - 1's values -- every load is clamped -- and are stored onto row M - 1: identical bytes, and every
            // wave issues the same number of vector memory instructions, which the counted waits depend on)
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(y) + doff_e[d] + j * 512) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the image has been read out
        if (j + 2 < NQ) { if (j == 0) { RN_RDMA(2, 0, 0); } else if (j == 1) { RN_RDMA(3, 1, 0); } }
    }
    // LayerNorm statistics of the 512 values of row i (lanes i and i + 32 hold its halves): two passes, as ln_fwd_row
    const float mu = wave_halves<WaveAdd>(s1) * (1.0f / (float)RN_N);
    float s2 = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const f32x16_t t = acc[nt];
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float c = t[e] - mu; s2 = fmaf(c, c, s2); }
        __builtin_amdgcn_sched_barrier(0);
    }
    const float rs = 1.0f / sqrtf(wave_halves<WaveAdd>(s2) * (1.0f / (float)RN_N) + eps);
    // xhat leaves through the wave's 32 KiB of the ring, which y has left (round 6; round 5 sent the four quarters through the 8-KiB fifth
    // buffer one after the other, two LDS round trips per quarter): quarter j, rows 4 r4 .. 4 r4 + 3 (256 bytes per row, piece p of row r at
    // slot p ^ (r & 15)) in the wave's KiB of (ring slot j, piece r4); written in the accumulator layout, read back row-major, stored as
    // whole 256-byte row segments -- all writes, one wait, all reads
    char* const xw_i = rw + (i_e >> 2) * 4096 + (i_e & 3) * 256 + 8 * g_e;
    const int sxh = (i_e & 15) << 4;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
#pragma unroll
        for (int ntl = 0; ntl < 4; ++ntl) {
            const f32x16_t t = acc[4 * j + ntl];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
                *reinterpret_cast<uint2*>(xw_i + j * 32768 + (((ntl * 4 + qq) << 4) ^ sxh)) =
                    make_uint2(pack_bf2((t[4 * qq] - mu) * rs, (t[4 * qq + 1] - mu) * rs), pack_bf2((t[4 * qq + 2] - mu) * rs, (t[4 * qq + 3] - mu) * rs));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
#pragma unroll
        for (int r4 = 0; r4 < 8; ++r4) {
            const int rl = 4 * r4 + xr_e;
            const uint4 v = *reinterpret_cast<const uint4*>(rw + j * 32768 + r4 * 4096 + lane_e * 16);
            *reinterpret_cast<uint4*>(xhat_o + (size_t)min(mw + rl, M - 1) * RN_N + j * 128 + ((xp_e ^ (rl & 15)) << 3)) = v;
        }
    }
    if (g_e == 0 && mw + i_e < M) {
        mean_o[mw + i_e] = mu;
        rstd_o[mw + i_e] = rs;
    }
}

// ---- C ABI -------------------------------------------------------------------------------------------------------------------
extern "C" size_t mbx_rows_n_pack_bytes(int N, int K) { return (size_t)N * K * sizeof(bf16_t); }

extern "C" int mbx_rows_n_pack_many(const int64_t* desc, int n_desc, int N, int max_k, void* stream) {
    MBX_CHECK_ARG(desc && n_desc > 0, "rows_n_pack_many: bad arguments");
    MBX_CHECK_ARG((N == 512 || N == 256) && max_k >= 256 && max_k % 256 == 0, "rows_n_pack_many: N=%d (256 or 512), max_k=%d (%% 256)", N, max_k);
    hipLaunchKernelGGL(rows_n_pack_many_kernel, dim3(((max_k / 16) * (N / 32) + 3) / 4, n_desc), dim3(256), 0, (hipStream_t)stream, desc, N / 32);
    MBX_LAUNCH_CHECK("rows_n_pack_many");
    return 0;
}

// shapes both kernels take: N = 512 (16 column tiles per wave; K >= 512: one ordinary trip in front of the peeled last one, as tested since
// round 5) or N = 256 (8 tiles; K >= 256: MotionBERT-Lite's proj contracts over 256 channels, a single -- the peeled -- trip)
#define RN_CHECK_SHAPE(who_)                                                                                         \
    MBX_CHECK_ARG(M > 0 && ((N == 512 && K >= 512) || (N == 256 && K >= 256)) && K % 256 == 0,                      \
                  who_ ": bad shape M=%d N=%d (512 with K >= 512, or 256 with K >= 256) K=%d (%% 256)", M, N, K)

extern "C" int mbx_rows_lnbwd_t(const void* dy, const void* packed, const void* xhat, const float* rstd, const void* dres_t, void* dx_t,
                                int M, int N, int K, void* stream) {
    MBX_CHECK_ARG(dy && packed && xhat && rstd && dres_t && dx_t, "rows_lnbwd_t: null pointer");
    RN_CHECK_SHAPE("rows_lnbwd_t");
    MBX_CHECK_ARG((size_t)M * N * 2 < ((size_t)1 << 32), "rows_lnbwd_t: M=%d rows of %d bytes exceed the 32-bit row offsets of the kernel", M, N * 2);
    MBX_CHECK_ARG((size_t)M * K * 2 < ((size_t)1 << 32), "rows_lnbwd_t: dy of M=%d x K=%d exceeds the 32-bit lane offsets of the kernel", M, K);
    MBX_CHECK_ARG(dx_t != dres_t && dx_t != xhat && dx_t != dy, "rows_lnbwd_t: dx_t aliases an input (rows past M re-read row M - 1 after it was stored)");
    const void* kern = N == 512 ? reinterpret_cast<const void*>(rows_n_lnbwd_kernel<16>) : reinterpret_cast<const void*>(rows_n_lnbwd_kernel<8>);
    if (mbx_set_dyn_lds(kern, RN_RING + 4 * 8192, "rows_lnbwd_t")) return 1;
#ifdef MBX_RN_TRACE
    {
        static long long* const tb = [] { const char* e = getenv("MBX_TRACE_BUF"); return e ? (long long*)strtoull(e, nullptr, 0) : (long long*)nullptr; }();
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_rn_trace), &tb, sizeof(tb), 0, hipMemcpyHostToDevice, (hipStream_t)stream);
    }
#endif
    if (N == 512)
        hipLaunchKernelGGL(rows_n_lnbwd_kernel<16>, dim3((M + RN_BM - 1) / RN_BM), dim3(256), RN_RING + 4 * 8192, (hipStream_t)stream, (const bf16_t*)dy,
                           (const char*)packed, (const bf16_t*)xhat, rstd, (const bf16_t*)dres_t, (bf16_t*)dx_t, M, K);
    else
        hipLaunchKernelGGL(rows_n_lnbwd_kernel<8>, dim3((M + RN_BM - 1) / RN_BM), dim3(256), RN_RING + 4 * 8192, (hipStream_t)stream, (const bf16_t*)dy,
                           (const char*)packed, (const bf16_t*)xhat, rstd, (const bf16_t*)dres_t, (bf16_t*)dx_t, M, K);
    MBX_LAUNCH_CHECK("rows_lnbwd_t");
    return 0;
}

extern "C" int mbx_rows_resid_ln(const void* a, const void* packed, const float* bias, const float* resid, float* y, void* xhat,
                                 float* mean, float* rstd, float eps, int M, int N, int K, void* stream) {
    MBX_CHECK_ARG(a && packed && bias && resid && y && xhat && mean && rstd, "rows_resid_ln: null pointer");
    RN_CHECK_SHAPE("rows_resid_ln");
    MBX_CHECK_ARG((size_t)M * N * 4 < ((size_t)1 << 32), "rows_resid_ln: M=%d rows of %d bytes exceed the 32-bit row offsets of the kernel", M, N * 4);
    MBX_CHECK_ARG((size_t)M * K * 2 < ((size_t)1 << 32), "rows_resid_ln: a of M=%d x K=%d exceeds the 32-bit lane offsets of the kernel", M, K);
    MBX_CHECK_ARG((const void*)y != (const void*)resid && xhat != a, "rows_resid_ln: y aliases resid (or xhat aliases a): rows past M re-read row M - 1 after it was stored");
    const void* kern = N == 512 ? reinterpret_cast<const void*>(rows_n_resid_ln_kernel<16>) : reinterpret_cast<const void*>(rows_n_resid_ln_kernel<8>);
    if (mbx_set_dyn_lds(kern, RN_RING + 4 * 8192, "rows_resid_ln")) return 1;
    if (N == 512)
        hipLaunchKernelGGL(rows_n_resid_ln_kernel<16>, dim3((M + RN_BM - 1) / RN_BM), dim3(256), RN_RING + 4 * 8192, (hipStream_t)stream, (const bf16_t*)a,
                           (const char*)packed, bias, resid, y, (bf16_t*)xhat, mean, rstd, eps, M, K);
    else
        hipLaunchKernelGGL(rows_n_resid_ln_kernel<8>, dim3((M + RN_BM - 1) / RN_BM), dim3(256), RN_RING + 4 * 8192, (hipStream_t)stream, (const bf16_t*)a,
                           (const char*)packed, bias, resid, y, (bf16_t*)xhat, mean, rstd, eps, M, K);
    MBX_LAUNCH_CHECK("rows_resid_ln");
    return 0;
}
