// Micro-benchmark: MFMA-from-LDS + LDS-DMA (L2-resident source) + a streaming store stream to HBM.
// Question: can output stores overlap the DMA/MFMA loop at all, and does the store flavour matter?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define GLDS16(src, dst) __builtin_amdgcn_global_load_lds((gbl_void_t*)(src), (lds_void_t*)(dst), 16, 0, 0)
__device__ __forceinline__ int sw_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// STORE_MODE: 0 none, 1 plain 16-B coalesced (1 KiB per wave-instruction), 2 non-temporal, 3 row-scattered 8 B (32 rows)
template <int NDMA, bool DOMFMA, int STORE_MODE, int STORE_EVERY, bool HBMMIX = false>
__global__ __launch_bounds__(512) void k(float* out, const char* src, size_t span, char* dst, int iters, const char* hbm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 31, g = lane >> 5;
    for (int kk = tid; kk < 16384; kk += 512) reinterpret_cast<uint32_t*>(smem)[kk] = 0x3c003c00u + (kk & 7);
    __syncthreads();
    f32x16_t acc[2][4];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const char* sA = smem + (wave & 1) * 8192;
    const char* sW = smem + 32768 + (wave >> 1) * 4096;
    char* land = smem + 65536 + wave * 8192;
    const char* gp = src + ((size_t)blockIdx.x * 8 + wave) * 65536 % span + (size_t)(lane >> 2) * 1024 + (lane & 3) * 16;
    // each wave owns a private 1 MiB-per-launch output stream in HBM
    char* dp = dst + ((size_t)blockIdx.x * 8 + wave) * ((size_t)iters / STORE_EVERY + 1) * 1024;
    for (int it = 0; it < iters; ++it) {
        if (NDMA > 0 && it >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // conservative: counts DMA groups only
        __builtin_amdgcn_s_barrier();
        if (NDMA > 0) {
            const char* p = gp + ((size_t)it * 64) % 512 + ((size_t)(it >> 3) * 16 * 1024) % (span / 4);
            // HBMMIX: every 6th k-iteration the first two instructions stream never-seen data (the A operand's first touch)
            const bool miss = HBMMIX && (it % 6) == 0;
            const char* hp = hbm + (((size_t)blockIdx.x * 8 + wave) * (size_t)(iters / 6 + 1) + (size_t)(it / 6)) * 2048 + lane * 16;
#pragma unroll
            for (int d = 0; d < NDMA; ++d)
                GLDS16((miss && d < 2) ? hp + d * 1024 : p + (size_t)d * 16 * 1024, land + ((it & 1) * NDMA + d) * 1024);
        }
        if (DOMFMA) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8_t fa[4], fw[2];
            const char* a = sA + ((it & 1) * 16384);
            const char* w = sW + ((it & 1) * 16384);
            for (int t = 0; t < 4; ++t) fa[t] = *reinterpret_cast<const bf16x8_t*>(a + sw_off((t * 32 + i) & 127, 2 * s + g));
            for (int t = 0; t < 2; ++t) fw[t] = *reinterpret_cast<const bf16x8_t*>(w + sw_off((t * 32 + i) & 63, 2 * s + g));
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a2][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[a2], fa[b], acc[a2][b], 0, 0, 0);
        }
        }
        if (STORE_MODE != 0 && (it % STORE_EVERY) == 0) {
            const u32x4_t v = {(unsigned)it, (unsigned)lane, 3u, 4u};
            char* q = dp + (size_t)(it / STORE_EVERY) * 1024;
            if (STORE_MODE == 1) *reinterpret_cast<u32x4_t*>(q + lane * 16) = v;
            else if (STORE_MODE == 2) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(q + lane * 16));
            else {   // 32 rows x 16 B: emulate an accumulator-layout store into a 3072-byte-pitch matrix
                char* r = dst + ((size_t)blockIdx.x * 8 + wave) * 98304 % ((size_t)1 << 30) + (size_t)(lane & 31) * 3072 + (lane >> 5) * 8 + (size_t)((it / STORE_EVERY) % 192) * 16;
                *reinterpret_cast<uint2*>(r) = make_uint2((unsigned)it, (unsigned)lane);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 512 + tid] = s + land[tid];
}
template <int NDMA, bool DOMFMA, int STORE_MODE, int STORE_EVERY, bool HBMMIX = false>
void run(const char* name, float* d_out, const char* src, size_t span, char* dst) {
    const int iters = 4096, grid = 256;
    auto kern = k<NDMA, DOMFMA, STORE_MODE, STORE_EVERY, HBMMIX>;
    const char* hbm = dst + ((size_t)5 << 30);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 131072, 0, d_out, src, span, dst, 16, hbm);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 131072, 0, d_out, src, span, dst, iters, hbm);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = DOMFMA ? (double)grid * 8 * iters * 2 * 8 * 32.0 * 32 * 16 * 2 : 0;
    const double bytes = (double)grid * 8 * iters * NDMA * 1024;
    const double sbytes = STORE_MODE == 0 ? 0 : (double)grid * 8 * (iters / STORE_EVERY) * (STORE_MODE == 3 ? 512 : 1024);
    printf("%-52s %8.3f ms  %7.1f TF/s  DMA %6.2f TB/s  stores %5.2f TB/s\n", name, ms, flops / ms / 1e9, bytes / ms / 1e9, sbytes / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 256 * 2 * 512 * 4);
    char* src; hipMalloc(&src, (size_t)64 << 20); hipMemset(src, 1, (size_t)64 << 20);
    char* dst; hipMalloc(&dst, ((size_t)10 << 30)); 
    const size_t span = (size_t)2 << 20;
    run<4, true, 0, 1>("mfma + dma (L2 src), no stores", d, src, span, dst);
    run<4, true, 0, 1, true>("mfma + dma (1/6 of half the DMA from HBM), no stores", d, src, span, dst);
    run<4, true, 1, 2, true>("  + plain stores every 2nd iter", d, src, span, dst);
    run<4, true, 1, 1, true>("  + plain stores every iter", d, src, span, dst);
    run<4, true, 3, 1, true>("  + row-scattered 8 B stores each iter", d, src, span, dst);
    run<4, true, 2, 1, true>("  + NT stores every iter", d, src, span, dst);
    return 0;
}
