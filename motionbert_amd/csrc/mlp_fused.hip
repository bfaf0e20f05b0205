// Fused MLP sub-layer, forward only (the no-grad / inference path):
//     y = x + fc2(gelu(fc1(LayerNorm(x))))  and the statistics of LayerNorm(y) for the sub-layer that reads y next
// (reference lib/model/DSTformer.py:79-85 MLP.forward inside Block.forward :241-249) as ONE kernel: the hidden [M, hidden]
// tensor (pre- and post-activation) never exists in HBM, and the next LayerNorm costs no pass over y.
// Per MLP the unfused kernel set moves  xhat 1 + (g 2 + 2) + x 2 + y 2 (+ the next LayerNorm: 2 + 1)  bf16-row units of HBM
// traffic, this kernel  operand 1 + x 2 + y 2 + bf16(y) 1.
//
// LayerNorm as a "raw operand" (the folded form of include/mbx.h taken one step further): Linear(LayerNorm(y)) =
// rstd (y . W'^T - mean rsum) + b' with W' = W diag(gamma), b' = b + W beta, rsum[n] = sum_k W'[n,k].  So a PRODUCER of the
// residual stream only has to leave bf16(y) and the row statistics (mean, rstd) -- no second pass over the row -- and the
// CONSUMER GEMM applies the row constants in its epilogue.  This kernel is both: with `mean_in / rstd_in` its fc1 takes the raw
// bf16 rows of its producer (otherwise xhat from a LayerNorm kernel), and it always can emit bf16(y) + (mean, rstd) of its output.
//
// Structure ("row owner"): a workgroup = 4 waves, ONE per SIMD (the whole 512-entry register file per lane), 128 token rows;
// wave w owns rows [32 w, 32 w + 32) COMPLETELY:
//   * X  = its 32 x C slice of xhat as MFMA operand fragments, C/4 registers, loaded once through LDS;
//   * per chunk of 64 hidden columns:  acc1[32 x 64] = X . W1c^T (2 x C/16 MFMAs 32x32x16), bias + GELU on the accumulator
//     registers, and -- this is what keeps the hidden on chip -- the packed bf16 result IS the token operand of the second GEMM:
//     in the transposed MFMA orientation used throughout this library (D = W-fragment x token-fragment, lane = token) lane (i, g)
//     ends up holding hidden columns {8q + 4g + e} of token i, i.e. eight of the sixteen k-slots of two K = 16 steps, as long as
//     W2's fragments list the hidden index in the same permuted order -- which the weight packer guarantees;
//   * acc2[32 x C] += G . W2c^T (C/32 x 4 MFMAs), C/2 accumulator registers that live across all chunks;
//     -- they START from residual + bias (loaded in the prologue), so the MFMAs leave y itself;
//   * epilogue: write y (fp32) and bf16(y), mean / rstd over the wave's complete rows.
// Register file (C = 512): acc2 256 + X 128 = 384 registers in the ACCUMULATOR half of the unified file (MFMA reads X from there
// as its B operand), 128 VGPRs for acc1 (32), two generations of the packed hidden (32), eight weight fragments in flight (32) and
// the GELU arithmetic.  hipcc cannot be talked into that split (its allocation of the plain-builtin version spills and shuffles
// 2000+ v_accvgpr moves per tile), so the MFMAs are inline asm with explicit register classes (cdna_hip_programming.md 5.7);
// everything else is compiler code.  What the asm statements own: the wait states between an MFMA chain and the first non-MFMA
// reader of its result (the NOP statements below), and issue order (volatile statements keep source order, so the fragment reads
// are software-pipelined by hand, PF fragments ahead).
// Every wave multiplies ALL weight fragments with its own rows, so the weights stream through a 4 x 32 KiB LDS ring filled by
// LDS-DMA; the packer (mlp_pack_kernel) lays them out as the exact sequence of 1-KiB MFMA fragments (lane-linear: lane l's
// 16 bytes at 16 l) the loop consumes, so one DMA instruction moves one contiguous KiB and a fragment read is
// `ds_read_b128 base + 16 lane + imm` -- conflict-free by construction, no address arithmetic.
// Pipeline (GELU must overlap MFMAs of another chunk -- with one wave per SIMD nothing else covers it):
//     A(0) gelu(0) | A(1) [B(0) || gelu(1)] | A(2) [B(1) || gelu(2)] | ... | A(n-1) [B(n-2) || gelu(n-1)] | B(n-1)
// A(c) = fc1 of chunk c, B(c) = fc2 of chunk c; the weight stream is consumed in exactly this order, 32 fragments per stage,
// ONE barrier per stage, placed at 3/4 of the stage so that the next stage's first fragments are read under this stage's last MFMAs.
#include "mbx_common.h"
#include "gelu_fast.h"

#include "lds_stream.h"

// Ablation switches of diagnostic builds (tools/build_variants.py name -DMBX_MLP_DBG=bits; results are wrong, timing only):
// 1 no GELU micro-steps beside fc2, 2 no LDS-DMA in the loop, 4 no fragment reads, 8 no epilogue, 16 no MFMAs, 32 no barriers,
// 64 no stages at all (prologue + epilogue only), 128 GELU micro-steps spread over all four stages of a chunk (dummy source)
#ifndef MBX_MLP_DBG
#define MBX_MLP_DBG 0
#endif
// A/B: cache policy of the tile's one-touch activation traffic.  bit 0: non-temporal stores of y in the epilogue; bit 1: non-temporal
// LDS-DMA loads of o and the residual rows in the prologue.  (The weight stream, 2.5 MiB that all 32 CUs of an XCD re-read every
// tile, competes with ~20 MiB of activations per tile period for the 4 MiB L2.)
// Measured in three sessions (profiles/r05_mlp_variants.txt): both bits -1.6 / -2.7 / -2.2 %, either bit alone within noise -> 3.
#ifndef MBX_MLP_NT
#define MBX_MLP_NT 3
#endif
#define MF_GLDS_ACT(src_, dst_) do { if (MBX_MLP_NT & 2) glds16_nt((src_), (dst_)); else GLDS16((src_), (dst_)); } while (0)
static constexpr int F_BM = 128;               // token rows per workgroup of the throughput shape (4 waves x 32)
static constexpr int F_STAGE = 32 * 1024;      // one ring stage = 32 fragments of 1 KiB
static constexpr int F_RING = 4 * F_STAGE;     // 128 KiB
static constexpr int F_CH = 64;                // hidden columns per chunk

// ---- packed weight stream ---------------------------------------------------------------------------------------------------
// chunk c (hidden columns [64 c, 64 c + 64)) = C/4 fragments: first the fc1 part, fragment (s, tn) at index 2 s + tn
// (s = k-step of 16 input channels, tn = 32-column half of the chunk), then the fc2 part, fragment (kk, nt) at index
// kk C/32 + nt (kk = one of the chunk's four K = 16 steps, nt = 32-column tile of the output).  Inside a fragment lane
// l = (i, g) = (l & 31, l >> 5) owns bytes [16 l, 16 l + 16):
//   fc1 (s, tn):  W1[64 c + 32 tn + i][16 s + 8 g + t],                                   t = 0..7
//   fc2 (kk, nt): W2[32 nt + i][64 c + 32 (kk >> 1) + 16 (kk & 1) + 8 (t >> 2) + 4 g + (t & 3)]
// -- the hidden index that lane (., g) of the fc1 accumulator holds in registers 4 q + e, q = 2 (kk & 1) + (t >> 2), e = t & 3.
__global__ __launch_bounds__(256) void mlp_pack_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2,
                                                       bf16_t* __restrict__ out, int C, int hidden) {
    const int frag = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int per_chunk = C / 4, fa = C / 8;                  // fragments per chunk (fc1: C/16 k-steps x 2, fc2: 4 x C/32), of which fc1
    if (frag >= hidden / F_CH * per_chunk) return;
    const int c = frag / per_chunk, f = frag % per_chunk;
    const int i = lane & 31, g = lane >> 5;
    uint4 v;
    if (f < fa) {
        const int s = f >> 1, tn = f & 1;
        v = *reinterpret_cast<const uint4*>(w1 + (size_t)(F_CH * c + 32 * tn + i) * C + 16 * s + 8 * g);
    } else {
        const int nt2 = C / 32, kk = (f - fa) / nt2, nt = (f - fa) % nt2;
        const bf16_t* p = w2 + (size_t)(32 * nt + i) * hidden + F_CH * c + 32 * (kk >> 1) + 16 * (kk & 1) + 4 * g;
        const uint2 lo = *reinterpret_cast<const uint2*>(p), hi = *reinterpret_cast<const uint2*>(p + 8);
        v = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    *reinterpret_cast<uint4*>(out + (size_t)frag * 512 + lane * 8) = v;
}

// proj part of the stream (the kernel's PROJ form: attention proj + residual in front of the MLP): fragment (kk, nt) at index
// kk C/32 + nt, lane (i, g): Wp[32 nt + i][16 kk + 8 g + t] -- the fc1 fragment format, k-step-major so that one token fragment
// serves the C/32 output tiles of a k-step in a row.
__global__ __launch_bounds__(256) void mlp_pack_proj_kernel(const bf16_t* __restrict__ wp, bf16_t* __restrict__ out, int C) {
    const int frag = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nt2 = C / 32;
    if (frag >= (C / 16) * nt2) return;
    const int kk = frag / nt2, nt = frag % nt2, i = lane & 31, g = lane >> 5;
    *reinterpret_cast<uint4*>(out + (size_t)frag * 512 + lane * 8) = *reinterpret_cast<const uint4*>(wp + (size_t)(32 * nt + i) * C + 16 * kk + 8 * g);
}

// XOR applied to the 16-byte piece index of the X image so that the fragment reads (32 rows, one piece each) are conflict-free
// for ds_read_b128's 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}: rows r = i & 7 of row group rb = i >> 3
__device__ __forceinline__ int x_swz(int r, int rb) { return ((r >> 1) & 3) | (((rb >> 1) & 1) << 2); }

// two bias floats from the LDS copy (read as a dword pair through the same kind of access as the fragment reads)
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 ge_bias_ld(const char* p) {
    const u32x2_t v = *reinterpret_cast<const u32x2_t*>(p);
    return make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
}

// PROJ (round 4, last step of the no-grad path): the attention's proj + residual runs in front of the MLP in the same kernel --
//     y1 = resid + xh . Wp^T + bp   (xh = the attention output o, bf16)      y = y1 + fc2(gelu(fc1(LayerNorm(y1))))
// The fc2 accumulators (which start from resid + bp) take the proj product first, C/16 k-steps x C/32 tiles with the token
// fragments of o where X will live; then every lane turns its accumulator registers into the fc1 operand bf16(y1) in registers
// (the accumulator layout holds columns 8 q + 4 g + e of a token, an operand fragment 8 consecutive ones: one v_permlane32_swap
// per register pair) and takes the LayerNorm statistics of y1 from the same values.  y1 never exists in HBM (- 8 bytes per element),
// one launch less per Block half.
// Diagnostic builds only (tools/build_variants.py name -DMBX_MLP_TRACE; tools/mlp_trace.py): 24 int64 per workgroup (wave 0):
// s_memrealtime ticks (100 MHz) in slots 0..14 (phases) and 16..20 (the four stages of chunk 8), the hardware id in slot 15, the
// shader-cycle counter at entry / exit in slots 22 / 23.  The buffer address comes from the environment variable MBX_TRACE_BUF.
#ifdef MBX_MLP_TRACE
__device__ long long* g_mlp_trace;
#define MF_TS(slot_) do { if (tr_on) tr[slot_] = (long long)wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
// (the stage stamps inside the chunk loop stay in scalar registers and are written out at the end: nothing for the loop to carry in VGPRs)
#if MBX_MLP_TRACE + 0 >= 2      // (a second trace build: these five stamps cost the register allocator 20 bytes of scratch in the PROJ form)
#define MF_TSC(slot_) do { if (c == 8) tsc[(slot_) - 16] = (long long)wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MF_TSC(slot_) do { } while (0)
#endif
#else
#define MF_TS(slot_) do { } while (0)
#define MF_TSC(slot_) do { } while (0)
#endif

// (Round 5, measured and dropped: 1 or 2 waves per workgroup -- 32- / 64-row tiles -- for small M, so that one clip (M = 4131,
// infer_wild.py:66-88) occupies 130 CUs instead of 33.  Parity-green, bit-identical rows, and SLOWER: 0.092 ms per launch against
// 0.087 (the forward of one clip 2.09 against 1.52 ms): a launch of less than one round takes ONE tile's critical path whatever the
// number of workgroups, and a lone wave issues four times the LDS-DMA pieces.  profiles/r05_mlp_variants.txt)
template <int C, bool PROJ = false>
__global__ __launch_bounds__(256, 1) void mlp_fused_kernel(const bf16_t* __restrict__ xh, const char* __restrict__ wpk,
                                                           const float* __restrict__ b1, const float* __restrict__ b2,
                                                           const float* __restrict__ rsum, int raw_in, const float* resid, float* y,
                                                           bf16_t* __restrict__ yb_out, float eps, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out, int M, int hidden, const float* __restrict__ bp) {
    constexpr int NT2 = C / 32;            // 32-column tiles of the output row
    constexpr int KS = C / 16;             // fc1 k-steps
    constexpr int S2 = C / 256;            // stages per fc1 part and per fc2 part of a chunk
    constexpr int KKS = 32 / NT2;          // fc2 k-steps per stage
    constexpr int CHB = 2 * S2 * F_STAGE;  // bytes of one chunk of the stream
    constexpr int NH = C / 256;            // 256-column halves of the output row (epilogue)
    constexpr int GSTEPS = 2 / S2;         // GELU micro-steps per fc2 slot (64 per chunk over S2 * 32 slots)
    constexpr int XL = C == 512 ? 4 : 0;   // the last XL fragments of X are kept in LDS and visit registers only around their use (the
                                           // second fc1 stage): the GELU stages, where every VGPR is spoken for, do not carry them
#ifndef MBX_MLP_PF
#define MBX_MLP_PF 5
#endif
    constexpr int PF = MBX_MLP_PF;         // weight fragments in flight ahead of the MFMA that consumes them (<= 7)
    extern __shared__ __attribute__((aligned(16))) char smem[];   // ring 128 KiB | b1 [hidden] | rsum [hidden] | b2 [C] | XL KiB per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
    const int m0 = blockIdx.x * F_BM, mw = m0 + 32 * wave;
    char* const ring = smem;
    float* const b1s = reinterpret_cast<float*>(smem + F_RING);
    float* const rss = b1s + hidden;
    float* const b2s = rss + hidden;
    char* const xsp = reinterpret_cast<char*>(b2s + C) + wave * (XL * 1024) + lane * 16;
    constexpr int NPS = PROJ ? KS * NT2 / 32 : 0;   // stages of the proj part of the stream
    const int nch = hidden / F_CH, NS = nch * 2 * S2;
    float* const bps = reinterpret_cast<float*>(smem + F_RING + (size_t)(2 * hidden + C) * 4 + (C == 512 ? 4 * 4096 : 0));   // PROJ: bp [C] behind everything else
#ifdef MBX_MLP_TRACE
    const bool tr_on = g_mlp_trace != nullptr && tid == 0;
    long long* const tr = g_mlp_trace + (size_t)blockIdx.x * 24;
    if (tr_on) { tr[15] = (long long)__builtin_amdgcn_s_getreg(63492) | ((long long)__builtin_amdgcn_s_getreg(63508) << 32); tr[22] = (long long)__builtin_readcyclecounter(); }
#endif
    long long tsc[5] = {0, 0, 0, 0, 0};
    MF_TS(0);

    // (Measured and dropped, round 4: a start stagger of the first workgroup of every CU by k/8 of a tile period, to run the memory
    // phases of some CUs under the compute phases of the others -- 101.2 us per tile round with and 102.7-103.3 without, i.e. null:
    // the CUs are not phase-locked.  Prologue-only and epilogue-only builds run at 4.8 / 5.4 TB/s = ~11 B/clk per CU, the per-CU
    // miss-bandwidth limit, and that time ADDS to the loop's: profiles/r04_mlp_fused_ablation.txt.)
    for (int k = tid; k < hidden; k += 256) { b1s[k] = b1[k]; rss[k] = raw_in ? rsum[k] : 0.f; }
    for (int k = tid; k < C; k += 256) { b2s[k] = b2[k]; if (PROJ) bps[k] = bp[k] + b2[k]; }

    // ---- X, the token operand of fc1, and the row constants of the raw-operand LayerNorm: fc1 = rstd acc + (b' - rstd mean rsum).
    //   xh given, raw_in = 0: xh is the normalised operand; constants (1, 0): fc1 = 1 acc + (b' + 0 rsum).
    //   xh given, raw_in = 1: xh = bf16 rows of the residual stream; (mean, rstd) are taken from the very values the MFMAs multiply
    //                         (lane (i, g) holds half of row i: packed-bf16 dots with ones / with itself, halves joined across lane ^ 32).
    //   xh == NULL:           the operand is bf16(resid) made HERE from the fp32 rows that are loaded for the accumulators anyway
    //                         (no second input stream, 256 instead of 384 KiB of prologue loads per tile), statistics in fp32.
    const bool from_x = xh == nullptr;             // wave-uniform (a kernel argument)
    u32x4_t X[KS];
    float ln_rs = 1.f, ln_k = 0.f;
    f32x16_t acc2[NT2];
    float xsh = 0.f, xs1 = 0.f, xs2 = 0.f;         // from_x: shifted sums over this lane's half row
#ifndef MBX_MLP_PRO_PIPE
#define MBX_MLP_PRO_PIPE 1
#endif
#if MBX_MLP_PRO_PIPE
    // ---- (round 5) the tile's inputs arrive as a PIPELINE of 16-KiB jobs through two buffers (the halves of the wave's 32 KiB of the
    // ring, which is idle until the weight stream starts): o in column halves (-> X fragments), then the residual rows in quarters of
    // 128 columns (-> four accumulator tiles each, which START from residual + bias so that the epilogue only writes; from_x: also
    // the operand fragments and the row statistics).  Job k + 2 is requested as soon as job k has been read out of its buffer, so one
    // or two jobs (64-128 KiB per CU) are in flight while the lanes move the previous one into registers.  Round 4 ran three 32-KiB
    // phases strictly one after the other -- request, drain, read: 17.9 us per tile in situ (profiles/r05_mlp_trace_before.txt).
    //   o job h:        instruction jj = 8 rows x 128 B: row group rb = jj & 3, column segment 4 h + (jj >> 2); image as in round 4
    //   residual job q: instruction r2 = rows 2 r2, 2 r2 + 1 x 512 B (columns [128 q, 128 q + 128)); 16-byte piece p of row r at slot
    //                   p ^ (r & 15) (applied to the source address): conflict-free in the accumulator layout -- lane (i, g): row i,
    //                   piece 8 ntl + 2 qq + g -- and in the operand-fragment layout -- row i, pieces 4 sl + 2 g, + 1.
    {
        constexpr int NO = KS / 16, NQ = C / 128;  // o jobs (xh given), residual jobs
        char* const bufA = ring + wave * 32768;
        char* const bufB = bufA + 16384;
        const int xr = lane >> 3, xp = lane & 7;
        auto issue_o = [&](int h, char* buf) {
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int rb = jj & 3, cs = 4 * h + (jj >> 2);
                const int row = min(mw + 8 * rb + xr, M - 1);
                MF_GLDS_ACT(xh + (size_t)row * C + cs * 64 + ((xp ^ x_swz(xr, rb)) << 3), buf + jj * 1024);
            }
        };
        auto issue_q = [&](int q, char* buf) {
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) {
                const int rl = 2 * r2 + (lane >> 5), p = lane & 31;
                MF_GLDS_ACT(resid + (size_t)min(mw + rl, M - 1) * C + q * 128 + ((p ^ (rl & 15)) << 2), buf + r2 * 1024);
            }
        };
        auto read_o = [&](int h, const char* buf) {
#pragma unroll
            for (int sl = 0; sl < 16; ++sl) {
                const int rb = i >> 3, r = i & 7, p = 2 * (sl & 3) + g;
                X[16 * h + sl] = *reinterpret_cast<const u32x4_t*>(buf + ((sl >> 2) * 4 + rb) * 1024 + r * 128 + ((p ^ x_swz(r, rb)) << 4));
            }
        };
        auto read_q = [&](int q, const char* buf) {
#pragma unroll
            for (int ntl = 0; ntl < 4; ++ntl) {
                const int nt = 4 * q + ntl;
                f32x16_t t;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float4 bb = *reinterpret_cast<const float4*>((PROJ ? bps : b2s) + nt * 32 + 8 * qq + 4 * g);   // (PROJ: bp + b2)
                    const float4 xv = *reinterpret_cast<const float4*>(buf + i * 512 + (((ntl * 8 + 2 * qq + g) ^ (i & 15)) << 4));
                    t[4 * qq] = xv.x + bb.x; t[4 * qq + 1] = xv.y + bb.y; t[4 * qq + 2] = xv.z + bb.z; t[4 * qq + 3] = xv.w + bb.w;
                    if (from_x) {
                        if (q == 0 && ntl == 0 && qq == 0) xsh = wave_lower_half(xv.x);      // x[row][0] for both half rows (see below)
                        const float d0 = xv.x - xsh, d1 = xv.y - xsh, d2 = xv.z - xsh, d3 = xv.w - xsh;
                        xs1 += (d0 + d1) + (d2 + d3);
                        xs2 = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, fmaf(d3, d3, xs2))));
                    }
                }
                acc2[nt] = t;
                asm volatile("s_nop 1" : "+a"(acc2[nt]));             // a whole tile at a time into accumulator registers, where it stays
            }
            if (from_x) {
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) {                      // k-steps 8 q + sl: eight consecutive channels of row i per lane
                    const float4 lo = *reinterpret_cast<const float4*>(buf + i * 512 + (((4 * sl + 2 * g) ^ (i & 15)) << 4));
                    const float4 hi = *reinterpret_cast<const float4*>(buf + i * 512 + (((4 * sl + 2 * g + 1) ^ (i & 15)) << 4));
                    // operand = bf16(x - x[row][0]): LayerNorm does not see the shift, and the rounding error then scales with the spread
                    // of the row, not with its magnitude (gemm_rows.hip, FROMX)
                    X[8 * q + sl] = u32x4_t{pack_bf2(lo.x - xsh, lo.y - xsh), pack_bf2(lo.z - xsh, lo.w - xsh), pack_bf2(hi.x - xsh, hi.y - xsh),
                                            pack_bf2(hi.z - xsh, hi.w - xsh)};
                }
            }
        };
        // job list: [o 0 .. NO - 1 (xh given)] + [residual 0 .. NQ - 1]; job k lives in buffer k & 1
#define MF_PRO_ISSUE(k_, no_) do { if ((k_) < (no_)) issue_o((k_), ((k_) & 1) ? bufB : bufA); else issue_q((k_) - (no_), ((k_) & 1) ? bufB : bufA); } while (0)
#define MF_PRO_JOBS(no_)                                                                                             \
        do {                                                                                                         \
            constexpr int NJ_ = (no_) + NQ;                                                                          \
            MF_PRO_ISSUE(0, no_);                                                                                    \
            MF_PRO_ISSUE(1, no_);                                                                                    \
            _Pragma("unroll") for (int k_ = 0; k_ < NJ_; ++k_) {                                                     \
                if (k_ + 1 < NJ_) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                  \
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
                if (k_ == (no_)) {        /* the first residual job reads the biases: they are in LDS behind this barrier */ \
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
                    __builtin_amdgcn_s_barrier();                                                                    \
                }                                                                                                    \
                if (k_ < (no_)) read_o(k_, (k_ & 1) ? bufB : bufA); else read_q(k_ - (no_), (k_ & 1) ? bufB : bufA); \
                if (k_ + 2 < NJ_) {                                                                                  \
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* the buffer has been read out */      \
                    MF_PRO_ISSUE(k_ + 2, no_);                                                                       \
                }                                                                                                    \
            }                                                                                                        \
        } while (0)
        if (!from_x) MF_PRO_JOBS(NO);
        else MF_PRO_JOBS(0);
        MF_TS(4);
        if (!from_x && raw_in && !PROJ) {          // wave-uniform (PROJ: the operand of fc1 and its statistics come from the proj product)
            float sa = 0.f, sb = 0.f, qa = 0.f, qb = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                sa = dot2_bf16(X[s][0], 0x3f803f80u, sa); qa = dot2_bf16(X[s][0], X[s][0], qa);
                sb = dot2_bf16(X[s][1], 0x3f803f80u, sb); qb = dot2_bf16(X[s][1], X[s][1], qb);
                sa = dot2_bf16(X[s][2], 0x3f803f80u, sa); qa = dot2_bf16(X[s][2], X[s][2], qa);
                sb = dot2_bf16(X[s][3], 0x3f803f80u, sb); qb = dot2_bf16(X[s][3], X[s][3], qb);
            }
            const float st = wave_halves<WaveAdd>(sa + sb), qt = wave_halves<WaveAdd>(qa + qb);
            const float mu = st * (1.0f / (float)C);
            ln_rs = 1.0f / sqrtf(fmaxf(qt * (1.0f / (float)C) - mu * mu, 0.f) + eps);
            ln_k = -ln_rs * mu;
        }
    }
#else
    if (!from_x) {
        // the wave's 32 rows of xh -> LDS (wave-private image, whole 128-byte lines per DMA) -> operand fragments
        char* const ximg = ring + wave * (64 * C);
        const int xr = lane >> 3, xp = lane & 7;
#pragma unroll 4
        for (int j = 0; j < KS; ++j) {             // one instruction = 8 rows x 128 B: row group rb = j & 3, column segment j >> 2
            const int rb = j & 3, cs = j >> 2;
            const int row = min(mw + 8 * rb + xr, M - 1);
            MF_GLDS_ACT(xh + (size_t)row * C + cs * 64 + ((xp ^ x_swz(xr, rb)) << 3), ximg + j * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int rb = i >> 3, r = i & 7, p = 2 * (s & 3) + g;
            X[s] = *reinterpret_cast<const u32x4_t*>(ximg + ((s >> 2) * 4 + rb) * 1024 + r * 128 + ((p ^ x_swz(r, rb)) << 4));
        }
        if (raw_in && !PROJ) {                     // wave-uniform (PROJ: the operand of fc1 and its statistics come from the proj product)
            float sa = 0.f, sb = 0.f, qa = 0.f, qb = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                sa = dot2_bf16(X[s][0], 0x3f803f80u, sa); qa = dot2_bf16(X[s][0], X[s][0], qa);
                sb = dot2_bf16(X[s][1], 0x3f803f80u, sb); qb = dot2_bf16(X[s][1], X[s][1], qb);
                sa = dot2_bf16(X[s][2], 0x3f803f80u, sa); qa = dot2_bf16(X[s][2], X[s][2], qa);
                sb = dot2_bf16(X[s][3], 0x3f803f80u, sb); qb = dot2_bf16(X[s][3], X[s][3], qb);
            }
            const float st = wave_halves<WaveAdd>(sa + sb), qt = wave_halves<WaveAdd>(qa + qb);
            const float mu = st * (1.0f / (float)C);
            ln_rs = 1.0f / sqrtf(fmaxf(qt * (1.0f / (float)C) - mu * mu, 0.f) + eps);
            ln_k = -ln_rs * mu;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MF_TS(1);
    __builtin_amdgcn_s_barrier();                  // the biases are in LDS
    MF_TS(2);

    // ---- the fc2 accumulators start from residual + bias: y = x + b2 + G . W2^T is then what the MFMAs leave, and the epilogue
    // only writes.  Per 256-column half the wave's 32 residual rows arrive in its LDS image by LDS-DMA (one instruction = one row's
    // KiB; 16-byte piece p of row r at p ^ (r & 15), applied to the source address: conflict-free in the accumulator layout -- lane
    // (i, g): row i, piece 8 ntl + 2 qq + g --, in the operand-fragment layout -- row i, pieces 4 s + 2 g, + 1 -- and row-major), and
    // each lane takes its accumulator registers from it; from_x: also its operand fragments and its half of the row's statistics.
    char* const er = ring + wave * 32768;
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the image's previous readers (X fragments / first half) are done
#pragma unroll 4
        for (int r = 0; r < 32; ++r)
            MF_GLDS_ACT(resid + (size_t)min(mw + r, M - 1) * C + hh * 256 + ((lane ^ (r & 15)) << 2), er + r * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int ntl = 0; ntl < 8; ++ntl) {
            const int nt = hh * 8 + ntl;
            f32x16_t t;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const float4 bb = *reinterpret_cast<const float4*>((PROJ ? bps : b2s) + nt * 32 + 8 * qq + 4 * g);   // (PROJ: bp + b2)
                const float4 xv = *reinterpret_cast<const float4*>(er + i * 1024 + (((ntl * 8 + 2 * qq + g) ^ (i & 15)) << 4));
                t[4 * qq] = xv.x + bb.x; t[4 * qq + 1] = xv.y + bb.y; t[4 * qq + 2] = xv.z + bb.z; t[4 * qq + 3] = xv.w + bb.w;
                if (from_x) {
                    if (hh == 0 && ntl == 0 && qq == 0) xsh = wave_lower_half(xv.x);      // x[row][0] for both half rows (see below)
                    const float d0 = xv.x - xsh, d1 = xv.y - xsh, d2 = xv.z - xsh, d3 = xv.w - xsh;
                    xs1 += (d0 + d1) + (d2 + d3);
                    xs2 = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, fmaf(d3, d3, xs2))));
                }
            }
            acc2[nt] = t;
            asm volatile("s_nop 1" : "+a"(acc2[nt]));             // a whole tile at a time into accumulator registers, where it stays
        }
        MF_TS(3 + hh);
        if (from_x) {
#pragma unroll
            for (int sl = 0; sl < 16; ++sl) {                     // k-steps 16 hh + sl: eight consecutive channels of row i per lane
                const float4 lo = *reinterpret_cast<const float4*>(er + i * 1024 + (((4 * sl + 2 * g) ^ (i & 15)) << 4));
                const float4 hi = *reinterpret_cast<const float4*>(er + i * 1024 + (((4 * sl + 2 * g + 1) ^ (i & 15)) << 4));
                // operand = bf16(x - x[row][0]): LayerNorm does not see the shift, and the rounding error then scales with the spread of
                // the row, not with its magnitude (gemm_rows.hip, FROMX)
                X[hh * 16 + sl] = u32x4_t{pack_bf2(lo.x - xsh, lo.y - xsh), pack_bf2(lo.z - xsh, lo.w - xsh), pack_bf2(hi.x - xsh, hi.y - xsh),
                                          pack_bf2(hi.z - xsh, hi.w - xsh)};
            }
        }
    }
#endif
    if (from_x) {                                  // the row statistics: this lane's half and the partner lane's (lane ^ 32), Chan's formula
        constexpr float nh = (float)(C / 2);
        const float mean_h = xsh + xs1 / nh, m2_h = xs2 - xs1 * xs1 / nh;
        const float mean_o = wave_halves<WaveAdd>(mean_h) - mean_h;
        const float m2_both = wave_halves<WaveAdd>(m2_h);
        const float delta = mean_o - mean_h;
        const float var = fmaxf((m2_both + delta * delta * (nh * 0.5f)) / (float)C, 0.f);
        ln_rs = 1.0f / sqrtf(var + eps);
        ln_k = -ln_rs * (0.5f * (mean_h + mean_o) - xsh);      // the mean of the shifted row
    }
    if constexpr (!PROJ) {                         // (PROJ: X is made after the proj product, below)
#pragma unroll
        for (int s = 0; s < XL; ++s) *reinterpret_cast<u32x4_t*>(xsp + s * 1024) = X[KS - XL + s];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // every wave is done with its image: the ring is free
    MF_TS(5);

    // ---- weight stream -------------------------------------------------------------------------------------------------------
    // stage sequence q -> byte offset in the packed stream:  A(0) | A(1) B(0) | A(2) B(1) | ... | A(n-1) B(n-2) | B(n-1)
    // (past the end: a harmless re-read of the last stage into a slot nobody reads again, so that the wait counts stay constant)
    // (PROJ: the NPS stages of the proj product come first, in order)
    auto seq_off = [&](int qq) -> int {
        if (PROJ && qq < NPS) return qq * F_STAGE;
        const int qc = min(qq - NPS, NS - 1), k = qc - S2, grp = k / (2 * S2), r = k % (2 * S2);
        const int mid = (r < S2 ? grp + 1 : grp) * CHB + r * F_STAGE;
        const int last = (nch - 1) * CHB + (S2 + r) * F_STAGE;
        return NPS * F_STAGE + (qc < S2 ? qc * F_STAGE : (grp >= nch - 1 ? last : mid));
    };
    // piece d of a stage = its fragments 4 d .. 4 d + 3, one per wave: this lane's 16 bytes sit at stage + 4096 d + 1024 wave + 16 lane
    const unsigned wvo = wave * 1024 + lane * 16;
    const unsigned dl = (unsigned)(uintptr_t)(const lds_void_t*)ring + wave * 1024;
#define MF_ISSUE1(src_, slot_, d_) glds16_s((src_) + (d_) * 4096, wvo, dl + (slot_) * F_STAGE + (d_) * 4096)
    // stage q + 1 has landed (this wave's pieces: counted wait; everyone's: barrier) and everyone is done with stage q - 1
#define MF_SYNC(n_)                                                                                                  \
    do {                                                                                                             \
        asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory");                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)

    f32x16_t acc1[2];
    // packed hidden: the fc2 of chunk c - 1 reads G[0..3] in order while the GELU of chunk c produces its four fragments in order;
    // fragment n + 1 of the new generation goes into the registers of fragment n of the old one (dead by then), fragment 0 into G[4]
    u32x4_t G[5];
    u32x4_t fb[8];                                              // weight fragments: slot k of a stage lives in fb[k & 7] (PF + 1 of them alive)
    u32x4_t xl[XL > 0 ? XL : 1];                                // the LDS-resident fragments of X, in registers during the second fc1 stage only
    const unsigned fr = (unsigned)(uintptr_t)(const lds_void_t*)ring + lane * 16;   // fragment f of ring slot s: fr + s * F_STAGE + f * 1024

    // Ring schedule (round 5, MBX_MLP_RING_DEEP = 1): the barrier of stage q sits in front of slot 32 - PF = 27 -- the first slot that reads
    // a fragment of stage q + 1, and the first at which EVERY wave has issued its last read of stage q's buffer (slot 26) -- so that
    // buffer is refilled right away with stage q + 4 (pieces 0, 1 in slots 27 and 31, pieces 2..7 in slots 3..23 of the next stage):
    // three stages (96 KiB) of the stream are in flight and the last-issued piece of a stage has two stage times to land.  Round 4 put
    // the barrier at slot 24 and refilled the buffer of stage q - 1 there, which had been idle for 29 slots by then: one and a half
    // stages in flight, ONE stage time for the last piece -- and a stage took as long as that piece's latency (0.76 us in situ for
    // 32 MFMAs = 0.5 us, with or without the GELU beside them: profiles/r05_mlp_trace_stages.txt).
#ifndef MBX_MLP_RING_DEEP
#define MBX_MLP_RING_DEEP 1
#endif
    constexpr int RD = MBX_MLP_RING_DEEP ? 1 : 0;               // extra stages of run-ahead
    constexpr int SYNC_SLOT = RD ? 32 - PF : 24;
    int q = 0;                                                  // stage sequence number
    {
        const char* const s0 = wpk + seq_off(0);
        const char* const s1 = wpk + seq_off(1);
        const char* const s2 = wpk + seq_off(2);
        const char* const s3 = wpk + seq_off(3);
#pragma unroll
        for (int d = 0; d < 8; ++d) MF_ISSUE1(s0, 0, d);
#pragma unroll
        for (int d = 0; d < 8; ++d) MF_ISSUE1(s1, 1, d);
        if (RD) {
#pragma unroll
            for (int d = 0; d < 8; ++d) MF_ISSUE1(s2, 2, d);
            MF_ISSUE1(s3, 3, 0);
            MF_ISSUE1(s3, 3, 1);
        } else {
            MF_ISSUE1(s2, 2, 0);
            MF_ISSUE1(s2, 2, 1);
        }
    }
    if (RD) MF_SYNC(18); else MF_SYNC(10);                      // stage 0 is in LDS (younger: stages 1 [, 2] and two pieces)
    MF_TS(6);
#pragma unroll
    for (int k = 0; k < PF; ++k) fb[k] = lds_read16(fr, k * 1024);

    // One stage = 32 slots.  Slot k: read the fragment of slot k + PF (from slot 32 - PF on, a fragment of the NEXT stage: the
    // barrier sits in front of that slot), one MFMA, every fourth slot one LDS-DMA piece (slots 3..23: pieces 2..7 of stage q + 3,
    // slots 27, 31: pieces 0, 1 of stage q + 4), then HOOK_ (GELU micro-steps); sched_barrier(0) pins the slot.
#define MF_STAGE(MMA_, HOOK_)                                                                                        \
    if (!(MBX_MLP_DBG & 64)) do {                                                                                    \
        unsigned st_ = fr + (q & 3) * F_STAGE, sn_ = fr + ((q + 1) & 3) * F_STAGE;                                   \
        asm volatile("" : "+v"(st_), "+v"(sn_));                                                                     \
        const char* const n2_ = wpk + seq_off(q + 2 + RD);                                                           \
        const char* const n3_ = wpk + seq_off(q + 3 + RD);                                                           \
        const int l2_ = (q + 2 + RD) & 3, l3_ = (q + 3 + RD) & 3;                                                    \
        _Pragma("unroll") for (int k_ = 0; k_ < 32; ++k_) {                                                          \
            if (k_ == SYNC_SLOT && !(MBX_MLP_DBG & 32)) { if (RD) MF_SYNC(16); else MF_SYNC(8); }                    \
            if (!(MBX_MLP_DBG & 4))                                                                                  \
                fb[(k_ + PF) & 7] = k_ + PF < 32 ? lds_read16(st_, (k_ + PF) * 1024) : lds_read16(sn_, (k_ + PF - 32) * 1024); \
            if (!(MBX_MLP_DBG & 16) || (k_ & 15) == 0) MMA_(k_, fb[k_ & 7]);                                         \
            if ((k_ & 3) == 3 && !(MBX_MLP_DBG & 2)) { if (k_ < 24) MF_ISSUE1(n2_, l2_, (k_ >> 2) + 2); else MF_ISSUE1(n3_, l3_, (k_ >> 2) - 6); } \
            if (!(MBX_MLP_DBG & 1)) HOOK_(k_);                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
        ++q;                                                                                                         \
    } while (0)

    // ---- GELU of acc1 (+ bias) -> the next generation of G, as 64 micro-steps of ~7 VALU operations: pair j = 0..15 (two accumulator registers), four
    // steps per pair (A&S 7.1.28 as in gelu_fast2: polynomial / polynomial + reciprocal / r^16 / combine + pack).  Pair j: column
    // half tn = j >> 3, register quad qq = (j >> 1) & 3, registers 4 qq + 2 (j & 1) + {0, 1}.
    mbx_f32x2_t ge_u, ge_a, ge_d;
    float2 ge_b = make_float2(0.f, 0.f), ge_r = make_float2(0.f, 0.f);   // bias / rsum of the next pair, read right after the current pair took its own
    int gc = 0;                                                 // chunk whose GELU runs (bias index)
#define GE_BIAS(j_) ge_bias_ld(smem + F_RING + 16 * g + (gc * F_CH + ((j_) >> 3) * 32 + 8 * (((j_) >> 1) & 3) + 2 * ((j_) & 1)) * 4)
#define GE_RSUM(j_) ge_bias_ld(smem + F_RING + 4 * hidden + 16 * g + (gc * F_CH + ((j_) >> 3) * 32 + 8 * (((j_) >> 1) & 3) + 2 * ((j_) & 1)) * 4)
#define GE_LOAD(j_) do { ge_b = GE_BIAS(j_); ge_r = GE_RSUM(j_); } while (0)
// (timing probe MBX_MLP_DBG & 128: the first eight GELU pairs run beside the fc1 stages on a dummy source -- what would spreading the
// GELU over all four stages of a chunk buy?)
#define GE_SRC(j_, tn_, r_) (((MBX_MLP_DBG & 128) && (j_) < 8) ? ge_dummy : acc1[tn_][r_])
    float ge_dummy = ln_rs;
// (scalar since round 5: measured 1-2 % faster than the packed form in two sessions -- packed fp32 VALU beside MFMAs costs more than its
// two scalar halves, MI355X_MICROARCH.md "price of one filler beside MFMAs"; MBX_MLP_GELU_SCALAR=0 is the A/B switch)
#ifndef MBX_MLP_GELU_SCALAR
#define MBX_MLP_GELU_SCALAR 1
#endif
#if MBX_MLP_GELU_SCALAR      // A/B form: the same arithmetic on scalar fp32 operations (two interleaved chains, no packed instructions)
    float gu0, gu1, ga0, ga1, gd0, gd1;
#define GE_STEP(ms_)                                                                                                 \
    do {                                                                                                             \
        const int j_ = (ms_) >> 2, st_ = (ms_) & 3, tn_ = j_ >> 3, qq_ = (j_ >> 1) & 3, r0_ = 4 * qq_ + 2 * (j_ & 1);  /* fold after unrolling */ \
        if (st_ == 0) {                                                                                              \
            gu0 = fmaf(ln_rs, acc1[tn_][r0_], fmaf(ln_k, ge_r.x, ge_b.x)); gu1 = fmaf(ln_rs, acc1[tn_][r0_ + 1], fmaf(ln_k, ge_r.y, ge_b.y)); \
            if (j_ + 1 < 16) GE_LOAD(j_ + 1);                                                                        \
            ga0 = fabsf(gu0); ga1 = fabsf(gu1);                                                                      \
            gd0 = fmaf(ga0, 5.382975e-06f, 4.8890636e-05f); gd1 = fmaf(ga1, 5.382975e-06f, 4.8890636e-05f);          \
            gd0 = fmaf(gd0, ga0, 3.8003575e-05f); gd1 = fmaf(gd1, ga1, 3.8003575e-05f);                              \
            gd0 = fmaf(gd0, ga0, 3.2776264e-03f); gd1 = fmaf(gd1, ga1, 3.2776264e-03f);                              \
        } else if (st_ == 1) {                                                                                       \
            gd0 = fmaf(gd0, ga0, 2.1141006e-02f); gd1 = fmaf(gd1, ga1, 2.1141006e-02f);                              \
            gd0 = fmaf(gd0, ga0, 4.9867347e-02f); gd1 = fmaf(gd1, ga1, 4.9867347e-02f);                              \
            gd0 = fmaf(gd0, ga0, 1.0f); gd1 = fmaf(gd1, ga1, 1.0f);                                                  \
            gd0 = __builtin_amdgcn_rcpf(gd0); gd1 = __builtin_amdgcn_rcpf(gd1);                                      \
        } else if (st_ == 2) {                                                                                       \
            gd0 *= gd0; gd1 *= gd1; gd0 *= gd0; gd1 *= gd1; gd0 *= gd0; gd1 *= gd1; gd0 *= gd0; gd1 *= gd1;          \
        } else {                                                                                                     \
            const float o0_ = ((gu0 + ga0) - ga0 * gd0) * 0.5f, o1_ = ((gu1 + ga1) - ga1 * gd1) * 0.5f;              \
            G[(tn_ * 2 + (qq_ >> 1) + 4) % 5][2 * (qq_ & 1) + (j_ & 1)] = pack_bf2(o0_, o1_);                        \
        }                                                                                                            \
    } while (0)
#else
#define GE_STEP(ms_)                                                                                                 \
    do {                                                                                                             \
        const int j_ = (ms_) >> 2, st_ = (ms_) & 3, tn_ = j_ >> 3, qq_ = (j_ >> 1) & 3, r0_ = 4 * qq_ + 2 * (j_ & 1);  /* fold after unrolling */ \
        if (st_ == 0) {                                                                                              \
            ge_u = mbx_f32x2_t{fmaf(ln_rs, GE_SRC(j_, tn_, r0_), fmaf(ln_k, ge_r.x, ge_b.x)), fmaf(ln_rs, GE_SRC(j_, tn_, r0_ + 1), fmaf(ln_k, ge_r.y, ge_b.y))}; \
            if (j_ + 1 < 16) GE_LOAD(j_ + 1);                     /* the next pair's constants: four slots ahead of their use */ \
            ge_a = mbx_f32x2_t{fabsf(ge_u[0]), fabsf(ge_u[1])};                                                      \
            ge_d = ge_a * 5.382975e-06f + 4.8890636e-05f;                                                            \
            ge_d = ge_d * ge_a + 3.8003575e-05f;                                                                     \
            ge_d = ge_d * ge_a + 3.2776264e-03f;                                                                     \
        } else if (st_ == 1) {                                                                                       \
            ge_d = ge_d * ge_a + 2.1141006e-02f;                                                                     \
            ge_d = ge_d * ge_a + 4.9867347e-02f;                                                                     \
            ge_d = ge_d * ge_a + 1.0f;                                                                               \
            ge_d = mbx_f32x2_t{__builtin_amdgcn_rcpf(ge_d[0]), __builtin_amdgcn_rcpf(ge_d[1])};                      \
        } else if (st_ == 2) {                                                                                       \
            ge_d = ge_d * ge_d; ge_d = ge_d * ge_d; ge_d = ge_d * ge_d; ge_d = ge_d * ge_d;                          \
        } else {                                                                                                     \
            const mbx_f32x2_t o_ = ((ge_u + ge_a) - ge_a * ge_d) * 0.5f;                                             \
            G[(tn_ * 2 + (qq_ >> 1) + 4) % 5][2 * (qq_ & 1) + (j_ & 1)] = pack_bf2(o_[0], o_[1]);                             \
        }                                                                                                            \
    } while (0)
#endif

    // fc1 slot k of the stage's half h: k-step s = 16 h + (k >> 1), column half tn = k & 1
#define MMA_A0(k_, w_) do { if ((k_) < 2) MFMA_FC1_Z(acc1[(k_) & 1], w_, X[(k_) >> 1]); else MFMA_FC1(acc1[(k_) & 1], w_, X[(k_) >> 1]); } while (0)
#define MMA_A1(k_, w_) do { if (16 + ((k_) >> 1) >= KS - XL) MFMA_FC1(acc1[(k_) & 1], w_, xl[(XL > 0 ? 16 + ((k_) >> 1) - (KS - XL) : 0)]); \
                            else MFMA_FC1(acc1[(k_) & 1], w_, X[16 + ((k_) >> 1)]); } while (0)
#define HOOK_A1(k_) do { if (XL > 0 && (k_) >= 16 && (k_) < 16 + 2 * XL && !((k_) & 1)) xl[((k_) - 16) >> 1] = *reinterpret_cast<const u32x4_t*>(xsp + (((k_) - 16) >> 1) * 1024); } while (0)
#define HOOK_NONE(k_) do { } while (0)
#define HOOK_PA0(k_) do { if ((MBX_MLP_DBG & 128) && !((k_) & 1)) GE_STEP((k_) >> 1); } while (0)
#define HOOK_PA1(k_) do { HOOK_A1(k_); if ((MBX_MLP_DBG & 128) && !((k_) & 1)) GE_STEP(16 + ((k_) >> 1)); } while (0)
    // fc2 slot k of the stage's part hb: k-step kk = hb KKS + k / NT2, output tile nt = k % NT2
#define MMA_B0(k_, w_) MFMA_FC2(acc2[(k_) % NT2], w_, G[(k_) / NT2])
#define MMA_B1(k_, w_) MFMA_FC2(acc2[(k_) % NT2], w_, G[KKS + (k_) / NT2])
    // GELU micro-steps beside fc2 slot k of part hb (C = 512: one per slot; C = 256: two); the first slot also pads the fc1 chain
#define HOOK_G0(k_) do { if ((k_) == 0) MFMA_PAD_V(acc1[0], acc1[1]); if (MBX_MLP_DBG & 128) { if (!((k_) & 1)) GE_STEP(32 + ((k_) >> 1)); } \
                         else if (GSTEPS == 2) { GE_STEP(2 * (k_)); GE_STEP(2 * (k_) + 1); } else GE_STEP(k_); } while (0)
#define HOOK_G1(k_) do { if (MBX_MLP_DBG & 128) { if (!((k_) & 1)) GE_STEP(48 + ((k_) >> 1)); } else GE_STEP(32 + (k_)); } while (0)

#define MF_PART_A()                                                                                                  \
    do {                                                                                                             \
        MF_STAGE(MMA_A0, HOOK_PA0);                                                                                  \
        if constexpr (S2 == 2) MF_STAGE(MMA_A1, HOOK_PA1);                                                           \
    } while (0)

    if constexpr (PROJ) {
        // ---- proj: acc2 (= resid + bp) += o . Wp^T.  One stage = 32 / NT2 k-steps x NT2 output tiles.
#ifndef MBX_MLP_PROJ_UNROLL
#define MBX_MLP_PROJ_UNROLL 1
#endif
#if MBX_MLP_PROJ_UNROLL
        // (round 5) every stage written out: the token fragments are then named registers.  The rolled loop of round 4 had to pick
        // them out of X by a chain of compares per stage (X cannot be indexed by a run-time stage counter) and paid a taken branch
        // between two MFMA bursts: 1.47 us per stage in situ where the weight stream allows 0.6 (profiles/r05_mlp_trace_before.txt).
#define MMA_P(k_, w_) MFMA_FC2(acc2[(k_) % NT2], w_, X[KKS * p + (k_) / NT2])
#pragma unroll
        for (int p = 0; p < NPS; ++p) MF_STAGE(MMA_P, HOOK_NONE);
#else
        u32x4_t xk[KKS];
#define MMA_P(k_, w_) MFMA_FC2(acc2[(k_) % NT2], w_, xk[(k_) / NT2])
        for (int p = 0; p < NPS; ++p) {
#pragma unroll
            for (int pp = 0; pp < NPS; ++pp)
                if (p == pp) {
#pragma unroll
                    for (int j = 0; j < KKS; ++j) xk[j] = X[KKS * pp + j];
                }
            MF_STAGE(MMA_P, HOOK_NONE);
        }
#endif
        MF_TS(7);
        // ---- y1 sits in the accumulators: operand fragments and LayerNorm statistics from them.  Lane (i, g) holds columns
        // 32 nt + 8 q + 4 g + e of token i in acc2[nt][4 q + e]; fragment s = 2 nt + h wants its columns 16 s + 8 g + [0, 8): quads
        // q = 2 h + g of BOTH half waves.  v_permlane32_swap(a, b) exchanges a's upper half with b's lower half: from
        // (a, b) = registers e of quads (2 h, 2 h + 1) every lane gets its own quad's value of lane (i, 0) in a and of lane (i, 1) in b.
#pragma unroll
        for (int t = 0; t < NT2; ++t) MFMA_PAD_A(acc2[t]);
        float s1 = 0.f, s2 = 0.f;
        // the operand is bf16(y1 - y1[row][0]) (LayerNorm does not see the shift; the rounding error then scales with the spread of
        // the row, not with its magnitude -- gemm_rows.hip, FROMX): the shift of BOTH half rows is lane (i, 0)'s first value
        float sh;
        {
            float v0 = acc2[0][0];
            asm volatile("s_nop 1" : "+v"(v0));
            sh = wave_lower_half(v0 - b2s[4 * g]);
        }
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            f32x16_t t = acc2[nt];                                 // y1 + b2 (the accumulators started from resid + bp + b2, so that the
#pragma unroll                                                     // epilogue stays a pure read of them): take b2 (and the shift) off again
            for (int qq = 0; qq < 4; ++qq) {
                const float4 bb = *reinterpret_cast<const float4*>(b2s + nt * 32 + 8 * qq + 4 * g);
                t[4 * qq] -= bb.x + sh; t[4 * qq + 1] -= bb.y + sh; t[4 * qq + 2] -= bb.z + sh; t[4 * qq + 3] -= bb.w + sh;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { s1 += t[r]; s2 = fmaf(t[r], t[r], s2); }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float a[4], b[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] = t[8 * h + e]; b[e] = t[8 * h + 4 + e]; }
                // four swaps behind ONE hazard pad (the VALU writes above -> permlane read; swap -> swap of other registers needs none)
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7"
                             : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]), "+v"(a[3]), "+v"(b[3]));
                X[2 * nt + h] = u32x4_t{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
                asm volatile("" : "+v"(X[2 * nt + h]));      // packed HERE (otherwise the conversions sink to the fragment's first use and the floats spill)
            }
            __builtin_amdgcn_sched_barrier(0);                    // one tile's 16 registers at a time out of the accumulator file
        }
        {
            constexpr float nh = (float)(C / 2);
            const float mean_h = sh + s1 / nh, m2_h = s2 - s1 * s1 / nh;
            const float mean_o = wave_halves<WaveAdd>(mean_h) - mean_h;
            const float m2_both = wave_halves<WaveAdd>(m2_h);
            const float delta = mean_o - mean_h;
            const float var = fmaxf((m2_both + delta * delta * (nh * 0.5f)) / (float)C, 0.f);
            ln_rs = 1.0f / sqrtf(var + eps);
            ln_k = -ln_rs * (0.5f * (mean_h + mean_o) - sh);      // the mean of the shifted row
        }
#pragma unroll
        for (int s = 0; s < XL; ++s) *reinterpret_cast<u32x4_t*>(xsp + s * 1024) = X[KS - XL + s];
    }
    MF_TS(8);
    // A(0), gelu(0): the only GELU that overlaps nothing
    MF_PART_A();
    MFMA_PAD_V(acc1[0], acc1[1]);
    GE_LOAD(0);
#define GE_ALL4(b_) GE_STEP((b_)); GE_STEP((b_) + 1); GE_STEP((b_) + 2); GE_STEP((b_) + 3)
#define GE_ALL16(b_) GE_ALL4(b_); GE_ALL4((b_) + 4); GE_ALL4((b_) + 8); GE_ALL4((b_) + 12)
    GE_ALL16(0); GE_ALL16(16); GE_ALL16(32); GE_ALL16(48);
#define G_ROTATE() do { asm volatile("s_nop 7" : "+v"(G[0]), "+v"(G[1]), "+v"(G[2]), "+v"(G[3]), "+v"(G[4]));                       \
                        G[3] = G[2]; G[2] = G[1]; G[1] = G[0]; G[0] = G[4];                                                         \
                        asm volatile("s_nop 3" : "+v"(G[0]), "+v"(G[1]), "+v"(G[2]), "+v"(G[3])); } while (0)
    G_ROTATE();
    MF_TS(9);
#ifdef MBX_MLP_ALIGN_PAD      // A/B: shift the chunk loop's code by 4 bytes (MI355X_MICROARCH.md: hand-placed streams can be sensitive to their 8-byte phase)
    asm volatile("s_nop 0");
#endif
#define MF_CHUNK()                                                                                                   \
    do {                                                                                                             \
        MF_TSC(16);                        /* (trace builds: the four stages of chunk 8, stamped one by one) */      \
        MF_STAGE(MMA_A0, HOOK_PA0);        /* A(c) */                                                                \
        MF_TSC(17);                                                                                                  \
        if constexpr (S2 == 2) MF_STAGE(MMA_A1, HOOK_PA1);                                                           \
        MF_TSC(18);                                                                                                  \
        gc = c;                                                                                                      \
        GE_LOAD(0);                                                                                                  \
        MF_STAGE(MMA_B0, HOOK_G0);         /* B(c - 1) || gelu(c) */                                                 \
        MF_TSC(19);                                                                                                  \
        if constexpr (S2 == 2) MF_STAGE(MMA_B1, HOOK_G1);                                                            \
        MF_TSC(20);                                                                                                  \
        G_ROTATE();                                                                                                  \
    } while (0)
    int c = 1;
#ifdef MBX_MLP_CHUNK_UNROLL   // A/B: three chunks per loop trip -- a taken branch between two MFMA bursts costs ~0.2 us (stage A0 1.04 us, A1 0.72)
    for (; c + 2 < nch; ++c) { MF_CHUNK(); ++c; MF_CHUNK(); ++c; MF_CHUNK(); }
#endif
    for (; c < nch; ++c) MF_CHUNK();
    MF_TS(10);
    MF_STAGE(MMA_B0, HOOK_NONE);           // B(n - 1)
    if constexpr (S2 == 2) MF_STAGE(MMA_B1, HOOK_NONE);
    MF_TS(11);

    // ---- epilogue: the accumulators hold y.  Statistics per lane (= half a row) as shifted sums, the halves joined by Chan's
    // formula; then per 256-column half the accumulators go through the wave's LDS image and leave row-major: one instruction =
    // one row's KiB of y / 512 B of bf16(y).  The accumulators are only READ (arithmetic on them would make new 16-register values).
    if (MBX_MLP_DBG & 8) return;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the re-read tail stages have landed, the last prefetches returned
    __builtin_amdgcn_s_barrier();                                 // every wave is done with the ring
    MF_TS(12);
#pragma unroll
    for (int t = 0; t < NT2; ++t) MFMA_PAD_A(acc2[t]);
    // (lane-derived values are re-derived from an opaque copy of the thread index: otherwise the compiler carries a dozen of them
    // through the main loop, where every VGPR is spoken for)
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, i_e = lane_e & 31, g_e = lane_e >> 5;
    const int rows = min(32, M - mw);
    // every accumulator tile is read ONCE: into the image, and (when the statistics are wanted) into the shifted sums on the way
    const bool want_stats = mean_out != nullptr;
    float sh = 0.f, s1 = 0.f, s2 = 0.f;
    char* const eo = ring + wave * 32768;
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (second half: the row-major reads of the first one are done)
#pragma unroll
        for (int ntl = 0; ntl < 8; ++ntl) {
            const int nt = hh * 8 + ntl;
            const f32x16_t t = acc2[nt];
            if (hh == 0 && ntl == 0) sh = t[0];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                *reinterpret_cast<float4*>(eo + i_e * 1024 + (((ntl * 8 + 2 * qq + g_e) ^ (i_e & 15)) << 4)) =
                    make_float4(t[4 * qq], t[4 * qq + 1], t[4 * qq + 2], t[4 * qq + 3]);
                if (want_stats) {                                 // wave-uniform
                    const float d0 = t[4 * qq] - sh, d1 = t[4 * qq + 1] - sh, d2 = t[4 * qq + 2] - sh, d3 = t[4 * qq + 3] - sh;
                    s1 += (d0 + d1) + (d2 + d3);
                    s2 = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, fmaf(d3, d3, s2))));
                }
            }
            __builtin_amdgcn_sched_barrier(0);                    // one tile's 16 registers at a time out of the accumulator file
        }
        float* const yrow = y + (size_t)mw * C + hh * 256;           // wave-uniform bases + a 32-bit lane offset
        bf16_t* const brow = yb_out + (size_t)mw * C + hh * 256;
#pragma unroll 4
        for (int r = 0; r < rows; ++r) {
            const float4 t = *reinterpret_cast<const float4*>(eo + r * 1024 + ((lane_e ^ (r & 15)) << 4));
            if (MBX_MLP_NT & 1) {
                typedef float nt_f4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(nt_f4{t.x, t.y, t.z, t.w}, reinterpret_cast<nt_f4*>(yrow + (size_t)r * C + lane_e * 4));
            } else {
                *reinterpret_cast<float4*>(yrow + (size_t)r * C + lane_e * 4) = t;
            }
            if (yb_out != nullptr)
                *reinterpret_cast<uint2*>(brow + (size_t)r * C + lane_e * 4) = make_uint2(pack_bf2(t.x, t.y), pack_bf2(t.z, t.w));
        }
    }
#ifdef MBX_MLP_TRACE
    MF_TS(13);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MF_TS(14);
    if (tr_on) { tr[23] = (long long)__builtin_readcyclecounter(); tr[16] = tsc[0]; tr[17] = tsc[1]; tr[18] = tsc[2]; tr[19] = tsc[3]; tr[20] = tsc[4]; }
#endif
    if (want_stats) {
        constexpr float nh = (float)(C / 2);                          // values per lane
        const float mean_h = sh + s1 / nh, m2_h = s2 - s1 * s1 / nh;  // this half row: mean, sum of squared deviations
        const float mean_o = wave_halves<WaveAdd>(mean_h) - mean_h;   // the partner lane's half (lane ^ 32)
        const float m2_both = wave_halves<WaveAdd>(m2_h);
        const float delta = mean_o - mean_h;
        const float var = fmaxf((m2_both + delta * delta * (nh * 0.5f)) / (float)C, 0.f);
        if (g_e == 0 && i_e < rows) {
            mean_out[mw + i_e] = 0.5f * (mean_h + mean_o);
            rstd_out[mw + i_e] = 1.0f / sqrtf(var + eps);
        }
    }
}

// ---- C ABI -------------------------------------------------------------------------------------------------------------------
extern "C" size_t mbx_mlp_pack_bytes(int C, int hidden) { return (size_t)2 * C * hidden * sizeof(bf16_t); }

extern "C" int mbx_mlp_pack_weights(const void* w1, const void* w2, void* packed, int C, int hidden, void* stream) {
    MBX_CHECK_ARG(w1 && w2 && packed, "mlp_pack_weights: null pointer");
    MBX_CHECK_ARG((C == 256 || C == 512) && hidden > 0 && hidden % F_CH == 0, "mlp_pack_weights: C=%d (256 or 512), hidden=%d (%% 64)", C, hidden);
    const int nfrag = hidden / F_CH * (C / 4);
    hipLaunchKernelGGL(mlp_pack_kernel, dim3((nfrag + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w1, (const bf16_t*)w2,
                       (bf16_t*)packed, C, hidden);
    MBX_LAUNCH_CHECK("mlp_pack_weights");
    return 0;
}

template <int C, bool PROJ>
static int launch_mlp_fused(const void* a, const void* packed, const float* b1, const float* b2, const float* rsum, int raw_in,
                            const float* resid, float* y, void* yb, float eps, float* mean, float* rstd, int M, int hidden, hipStream_t s,
                            const float* bp = nullptr) {
    const size_t shm = F_RING + (size_t)(2 * hidden + C) * sizeof(float) + (C == 512 ? 4 * 4096 : 0) + (PROJ ? C * sizeof(float) : 0);
    if (mbx_set_dyn_lds(reinterpret_cast<const void*>(mlp_fused_kernel<C, PROJ>), shm, "mlp_fused_fwd")) return 1;
#ifdef MBX_MLP_TRACE
    {
        static long long* const tb = [] { const char* e = getenv("MBX_TRACE_BUF"); return e ? (long long*)strtoull(e, nullptr, 0) : (long long*)nullptr; }();
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_mlp_trace), &tb, sizeof(tb), 0, hipMemcpyHostToDevice, s);
    }
#endif
    hipLaunchKernelGGL((mlp_fused_kernel<C, PROJ>), dim3((M + F_BM - 1) / F_BM), dim3(256), shm, s, (const bf16_t*)a, (const char*)packed, b1, b2,
                       rsum, raw_in, resid, y, (bf16_t*)yb, eps, mean, rstd, M, hidden, bp);
    MBX_LAUNCH_CHECK("mlp_fused_fwd");
    return 0;
}

extern "C" int mbx_mlp_fused_fwd(const void* a, int raw_in, const void* packed, const float* b1, const float* b2, const float* rsum,
                                 const float* resid, float* y, void* y_t, float eps, float* mean, float* rstd, int M, int C,
                                 int hidden, void* stream) {
    MBX_CHECK_ARG(packed && b1 && b2 && resid && y, "mlp_fused_fwd: null pointer");
    MBX_CHECK_ARG(a || raw_in, "mlp_fused_fwd: a = NULL (the operand is taken from resid) is a raw operand: raw_in must be 1");
    MBX_CHECK_ARG(M > 0 && (C == 256 || C == 512) && hidden >= F_CH && hidden % F_CH == 0 && hidden <= 1536,
                  "mlp_fused_fwd: bad shape M=%d C=%d (256 or 512) hidden=%d (%% 64, 64..1536)", M, C, hidden);
    MBX_CHECK_ARG(!raw_in || rsum, "mlp_fused_fwd: a raw operand needs rsum (row sums of the folded fc1 weights)");
    MBX_CHECK_ARG((mean && rstd) || (!mean && !rstd), "mlp_fused_fwd: mean and rstd come together");
    hipStream_t s = (hipStream_t)stream;
    if (C == 512) return launch_mlp_fused<512, false>(a, packed, b1, b2, rsum, raw_in, resid, y, y_t, eps, mean, rstd, M, hidden, s);
    return launch_mlp_fused<256, false>(a, packed, b1, b2, rsum, raw_in, resid, y, y_t, eps, mean, rstd, M, hidden, s);
}

// ---- attention proj + residual + the MLP sub-layer in one kernel (the PROJ form above) --------------------------------------------
extern "C" size_t mbx_proj_mlp_pack_bytes(int C, int hidden) { return (size_t)C * C * sizeof(bf16_t) + mbx_mlp_pack_bytes(C, hidden); }

extern "C" int mbx_proj_mlp_pack_weights(const void* wp, const void* w1, const void* w2, void* packed, int C, int hidden, void* stream) {
    MBX_CHECK_ARG(wp && w1 && w2 && packed, "proj_mlp_pack_weights: null pointer");
    MBX_CHECK_ARG((C == 256 || C == 512) && hidden > 0 && hidden % F_CH == 0, "proj_mlp_pack_weights: C=%d (256 or 512), hidden=%d (%% 64)", C, hidden);
    const int nfp = (C / 16) * (C / 32);
    hipLaunchKernelGGL(mlp_pack_proj_kernel, dim3((nfp + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)wp, (bf16_t*)packed, C);
    MBX_LAUNCH_CHECK("proj_mlp_pack_weights");
    return mbx_mlp_pack_weights(w1, w2, (char*)packed + (size_t)C * C * sizeof(bf16_t), C, hidden, stream);
}

extern "C" int mbx_proj_mlp_fused_fwd(const void* o, const void* packed, const float* bp, const float* b1, const float* b2,
                                      const float* rsum, const float* resid, float* y, float eps, int M, int C, int hidden, void* stream) {
    MBX_CHECK_ARG(o && packed && bp && b1 && b2 && rsum && resid && y, "proj_mlp_fused_fwd: null pointer");
    MBX_CHECK_ARG(M > 0 && (C == 256 || C == 512) && hidden >= F_CH && hidden % F_CH == 0 && hidden <= 1536,
                  "proj_mlp_fused_fwd: bad shape M=%d C=%d (256 or 512) hidden=%d (%% 64, 64..1536)", M, C, hidden);
    hipStream_t s = (hipStream_t)stream;
    if (C == 512) return launch_mlp_fused<512, true>(o, packed, b1, b2, rsum, 1, resid, y, nullptr, eps, nullptr, nullptr, M, hidden, s, bp);
    return launch_mlp_fused<256, true>(o, packed, b1, b2, rsum, 1, resid, y, nullptr, eps, nullptr, nullptr, M, hidden, s, bp);
}
