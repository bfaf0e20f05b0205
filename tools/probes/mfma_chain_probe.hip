// Round 5: what does it cost to accumulate consecutive v_mfma_f32_32x32x16_bf16 into the SAME accumulator when other instructions
// sit between them?  The row-owner GEMM (gemm_rows.hip) runs one 32-column tile = 32 MFMAs as ONE dependent chain (so that the
// previous tile's epilogue can run under it), with a fragment read and epilogue / LDS-DMA work between the links; the fused MLP
// alternates two (fc1) or sixteen (fc2) accumulators.  MI355X_MICROARCH.md: "one EXTRA issue slot between two MFMAs on the SAME
// accumulator: +43 cycles for the first extra state".  Here: NACC accumulators used round-robin, one ds_read_b128 per MFMA (PF ahead),
// FILL VALU operations per slot, 1 or 2 workgroups of 4 waves per CU (one or two waves per SIMD).  Output: cycles per MFMA per SIMD.
//     hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_chain_probe.hip -o tools/probes/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ u32x4_t lds_read16(unsigned base, int imm) {
    return *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t*>(base + imm);
}
template <int NACC, int FILL>
__global__ __launch_bounds__(256, 2) void chain_kernel(float* __restrict__ sink, long long* __restrict__ cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char buf[32 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32 * 1024 / 4; i += 256) reinterpret_cast<uint32_t*>(buf)[i] = 0x3c003c00u ^ ((i * 2654435761u) & 0x83ff83ffu);
    __syncthreads();
    const unsigned fr = (unsigned)(uintptr_t)(const lds_void_t*)buf + lane * 16;
    u32x4_t X[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) X[k][e] = 0x3f803f80u ^ ((uint32_t)(tid * 2654435761u + k * 40503u + e * 9973u) & 0x007f007fu);
    f32x16_t acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    float fv[4] = {1.0f + lane * 1e-3f, 0.5f, 0.25f, 0.125f};
    constexpr int PF = 4;
    u32x4_t fb[8];
#pragma unroll
    for (int k = 0; k < PF; ++k) fb[k] = lds_read16(fr, k * 1024);
    const long long c0 = (long long)__builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        unsigned st = fr;
        asm volatile("" : "+v"(st));
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            fb[(k + PF) & 7] = lds_read16(st, ((k + PF) & 31) * 1024);
            acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[k & 7]), __builtin_bit_cast(bf16x8_t, X[k & 3]), acc[k % NACC], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < FILL; ++f) fv[f & 3] = fmaf(fv[f & 3], fv[(f + 1) & 3], 0.001f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long c1 = (long long)__builtin_readcyclecounter();
    float s = fv[0] + fv[1] + fv[2] + fv[3];
#pragma unroll
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][15];
    sink[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = c1 - c0;
}
template <int NACC, int FILL>
static void run(float* sink, long long* cyc, int wgs_per_cu) {
    const int iters = 2000, blocks = 256 * wgs_per_cu;
    hipLaunchKernelGGL((chain_kernel<NACC, FILL>), dim3(blocks), dim3(256), 0, 0, sink, cyc, iters);
    hipLaunchKernelGGL((chain_kernel<NACC, FILL>), dim3(blocks), dim3(256), 0, 0, sink, cyc, iters);
    hipDeviceSynchronize();
    static long long h[512];
    hipMemcpy(h, cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0;
    for (int b = 0; b < blocks; ++b) m += (double)h[b];
    m /= blocks;
    // a wave issues 32 * iters MFMAs in m cycles; with w waves per SIMD the SIMD issues w times as many in the same time
    printf("accumulators %2d  VALU/slot %d  waves/SIMD %d: %6.1f cycles per MFMA and wave = %5.1f per MFMA and SIMD (32 = the matrix pipe)\n", NACC, FILL, wgs_per_cu,
           m / (32.0 * iters), m / (32.0 * iters) / wgs_per_cu);
}
int main() {
    float* sink;
    long long* cyc;
    hipMalloc(&sink, 512 * 256 * sizeof(float));
    hipMalloc(&cyc, 512 * sizeof(long long));
    for (int w = 1; w <= 2; ++w) {
        run<1, 0>(sink, cyc, w); run<2, 0>(sink, cyc, w); run<4, 0>(sink, cyc, w);
        run<1, 2>(sink, cyc, w); run<2, 2>(sink, cyc, w); run<4, 2>(sink, cyc, w);
        run<1, 5>(sink, cyc, w); run<2, 5>(sink, cyc, w);
    }
    return 0;
}
