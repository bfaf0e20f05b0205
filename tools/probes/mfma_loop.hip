// Micro-benchmark: how fast can a workgroup feed v_mfma_f32_32x32x16_bf16 from LDS with the fragment
// pattern of the GEMM kernels?  Variants: wave tile (2x2 or 4x2 MFMA tiles), barrier per step or not,
// s_setprio around the MFMA cluster, explicit fragment double-buffering.  Prints TFLOP/s per variant.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__device__ __forceinline__ int sw_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int TM, int TN, bool BARRIER, bool PRIO, bool DBUF>
__global__ __launch_bounds__(512) void loop_kernel(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 64 KiB of operand data
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, g = lane >> 5;
    for (int k = tid; k < 16384; k += 512) reinterpret_cast<uint32_t*>(smem)[k] = 0x3c003c00u + (k & 7);
    __syncthreads();
    f32x16_t acc[TN][TM];
    for (int a = 0; a < TN; ++a) for (int b = 0; b < TM; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const char* sA = smem + (wave & 1) * 8192;
    const char* sW = smem + 32768 + (wave >> 1) * 4096;
    bf16x8_t fa[2][TM], fw[2][TN];
    auto rd = [&](int buf, int s, int it) {
        const char* a = sA + ((it & 3) * 16384 % 32768);
        const char* w = sW + ((it & 1) * 16384);
        for (int t = 0; t < TM; ++t) fa[buf][t] = *reinterpret_cast<const bf16x8_t*>(a + sw_off((t * 32 + i) & 127, 2 * s + g));
        for (int t = 0; t < TN; ++t) fw[buf][t] = *reinterpret_cast<const bf16x8_t*>(w + sw_off((t * 32 + i) & 63, 2 * s + g));
    };
    if (DBUF) rd(0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        if (BARRIER) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int cur = DBUF ? (s & 1) : 0;
            if (DBUF) rd(cur ^ 1, (s + 1) & 1, it + (s == 1)); else rd(0, s, it);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[cur][a], fa[cur][b], acc[a][b], 0, 0, 0);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < TN; ++a) for (int b = 0; b < TM; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 512 + tid] = s;
}

template <int TM, int TN, bool BARRIER, bool PRIO, bool DBUF>
void run(const char* name, int blocks_per_cu, float* d_out) {
    const int iters = 2000, grid = 256 * blocks_per_cu;
    auto k = loop_kernel<TM, TN, BARRIER, PRIO, DBUF>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 65536, 0, d_out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 65536, 0, d_out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 8 * iters * 2 * TM * TN * 32.0 * 32 * 16 * 2;
    printf("%-44s blocks/CU=%d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 256 * 2 * 512 * 4);
    run<2, 2, false, false, false>("2x2 tiles, no barrier", 1, d);
    run<2, 2, false, false, false>("2x2 tiles, no barrier", 2, d);
    run<2, 2, true, false, false>("2x2 tiles, barrier/32k", 1, d);
    run<2, 2, true, false, false>("2x2 tiles, barrier/32k", 2, d);
    run<2, 2, true, true, false>("2x2 tiles, barrier, setprio", 2, d);
    run<2, 2, true, false, true>("2x2 tiles, barrier, frag dbuf", 2, d);
    run<4, 2, false, false, false>("4x2 tiles, no barrier", 1, d);
    run<4, 2, true, false, false>("4x2 tiles, barrier/32k", 1, d);
    run<4, 2, true, true, false>("4x2 tiles, barrier, setprio", 1, d);
    run<4, 2, true, false, true>("4x2 tiles, barrier, frag dbuf", 1, d);
    run<4, 2, true, true, true>("4x2 tiles, barrier, setprio, frag dbuf", 1, d);
    return 0;
}
