// Measurement aid, not part of the model: the bf16 MFMA rate this part SUSTAINS under its power cap, with nothing but
// v_mfma_f32_32x32x16_bf16 in the loop -- the ceiling every `roofline.frac` of bench.py is also quoted against (VERDICT r4 item 7:
// "settle the ceiling with evidence").  The datasheet peak (2.5 PFLOP/s) assumes 2.4 GHz; under a matrix load the package sits at its
// power limit and clocks lower, and how much lower depends on the DATA (zero operands toggle nothing and run ~19 % faster,
// MI355X_MICROARCH.md "DVFS give-back") -- so the operands here are pseudo-random bf16 values of the magnitude the model's GEMMs see.
// One workgroup = 4 waves, one per SIMD; `wgs_per_cu` workgroups per CU; every wave issues `iters` x 16 independent-accumulator MFMAs
// back to back (four accumulators, so no MFMA waits for its predecessor).  The kernel also returns the shader cycles and the 100 MHz
// real-time ticks it ran for, i.e. the effective clock, per workgroup.
#include "mbx_common.h"

__device__ __forceinline__ uint32_t probe_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// two bf16 values in [-2, 2) with random mantissas (exponent field 0x3f / 0x40 region): sign | 0x3f80..0x407f
__device__ __forceinline__ uint32_t probe_bf2(uint32_t h) {
    const uint32_t lo = (h & 0x8000u) | 0x3f00u | (h & 0xffu) | ((h >> 3) & 0x100u);
    const uint32_t hi = ((h >> 16) & 0x8000u) | 0x3f00u | ((h >> 16) & 0xffu) | ((h >> 19) & 0x100u);
    return lo | (hi << 16);
}

__global__ __launch_bounds__(256, 1) void mfma_probe_kernel(float* __restrict__ sink, long long* __restrict__ stamps, int iters, uint32_t seed) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t id = (blockIdx.x * 256u + threadIdx.x) * 16u + seed;
    u32x4 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[k][e] = probe_bf2(probe_hash(id + 4 * k + e));
            b[k][e] = probe_bf2(probe_hash(id + 0x9e3779b9u + 4 * k + e)) & 0xbfffbfffu;   // |b| < 1: the accumulators stay finite-ish
        }
    }
    f32x16_t acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    const long long c0 = (long long)__builtin_readcyclecounter(), r0 = (long long)wall_clock64();
    // 16 iterations = 256 MFMAs of straight-line code per loop trip: a taken branch between MFMAs stalls the matrix pipe for a few
    // hundred cycles (tools/probes/mfma_issue.hip), so it has to be rare
#pragma unroll 1
    for (int it = 0; it < iters; it += 16) {
#pragma unroll
        for (int r = 0; r < 64; ++r) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[(k + r) & 3]), __builtin_bit_cast(bf16x8_t, b[(k + (r >> 2)) & 3]), acc[k], 0, 0, 0);
        }
    }
    const long long c1 = (long long)__builtin_readcyclecounter(), r1 = (long long)wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[k][e];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && stamps != nullptr) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

extern "C" size_t mbx_mfma_probe_ws(int n_wg) { return (size_t)n_wg * (256 * sizeof(float) + 2 * sizeof(long long)); }

extern "C" int mbx_mfma_probe(void* ws, int n_wg, int iters, unsigned seed, double* flops, void* stream) {
    iters = (iters + 15) / 16 * 16;
    MBX_CHECK_ARG(ws && n_wg > 0 && iters > 0, "mfma_probe: bad arguments (ws=%p n_wg=%d iters=%d)", ws, n_wg, iters);
    float* const sink = reinterpret_cast<float*>(ws);
    long long* const stamps = reinterpret_cast<long long*>(reinterpret_cast<char*>(ws) + (size_t)n_wg * 256 * sizeof(float));
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, sink, stamps, iters, (uint32_t)seed);
    MBX_LAUNCH_CHECK("mfma_probe");
    if (flops) *flops = (double)n_wg * 4.0 * (double)iters * 16.0 * (2.0 * 32 * 32 * 16);
    return 0;
}
