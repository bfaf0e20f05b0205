// Throughput of ds_read_b64_tr_b16 (and ds_read_b128 for comparison) per CU as a function of the number of waves per SIMD that
// issue it, with the weight-gradient GEMM's access pattern (tile [32 tokens][256 columns] bf16, 64-byte-chunk XOR swizzle, the
// fragment addresses of gemm_pipe.hip's tr_frag).  Every wave issues bursts of READS reads + s_waitcnt lgkmcnt(0), ITERS times.
// Question (round 3): is the ~44 cycles per read seen by ONE wave per SIMD a per-wave limit (more waves -> more bytes per clock)
// or is the LDS itself that slow for this instruction?      hipcc --offload-arch=gfx950 -O3 tr_rate.hip -o tr_rate && ./tr_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 2000
template <int ROWB> __device__ __forceinline__ int tr_off(int r, int c) {
    const int byte = c * 2, c64 = byte >> 6;
    return r * ROWB + (((c64 & ~3) | ((c64 ^ r) & 3)) << 6) + (byte & 63);
}
template <int MODE, int READS>
__global__ void rate(uint32_t* out, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 32 * 512 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(lds)[i] = i * 2654435761u;
    __syncthreads();
    const int g = lane >> 5, r16 = lane & 15;
    uint32_t addr[READS];
#pragma unroll
    for (int k = 0; k < READS; ++k) {
        if (MODE == 0) {      // transposing 8-byte read: fragment k of the wave's tile (columns 32 (k % 8) + ..., token half k / 8)
            const int nb = 32 * ((k + wave) % 8) + 16 * ((lane >> 4) & 1) + 4 * (r16 & 3), t = 16 * ((k / 8) & 1) + 8 * g + (r16 >> 2);
            addr[k] = (uint32_t)(uintptr_t)lds + tr_off<512>(t, nb);
        } else {              // 16-byte read, NT-GEMM style: row = lane & 31, chunk by lane half (conflict-free swizzle not modelled: linear rows of 64 B)
            addr[k] = (uint32_t)(uintptr_t)lds + (((lane & 31) + 32 * ((k + wave) % 8)) * 64 + g * 16 + 32 * ((k / 8) & 1)) % (32 * 512);
        }
    }
    uint32_t acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    for (int it = 0; it < ITERS; ++it) {
        v2u a[READS];
        v4u b[READS];
#pragma unroll
        for (int k = 0; k < READS; ++k) {
            if (MODE == 0) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a[k]) : "v"(addr[k]) : "memory");
            else asm volatile("ds_read_b128 %0, %1" : "=v"(b[k]) : "v"(addr[k]) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < READS; ++k) acc ^= MODE == 0 ? (a[k].x ^ a[k].y) : (b[k].x ^ b[k].w);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE, int READS> static void run(const char* name, int waves_per_simd) {
    uint32_t* out; long long* cyc;
    const int threads = 256 * waves_per_simd, blocks = 256;
    hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto k = rate<MODE, READS>;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 32 * 512, 0, out, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 32 * 512, 0, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < blocks; ++i) c += (double)h[i]; c /= blocks;
    const double reads = (double)ITERS * READS * 4 * waves_per_simd;   // wave-instructions per CU
    const double bytes = reads * 64 * (MODE == 0 ? 8 : 16);
    printf("%-22s waves/SIMD %d reads/burst %2d: %8.0f counter ticks, %.3f ms -> %6.1f B per us per CU x1e-3 = %6.1f GB/s per CU, %5.1f ns per wave-read per wave\n",
           name, waves_per_simd, READS, c, ms, bytes / (ms * 1e3) / 1e3, bytes / (ms * 1e-3) / 1e9, ms * 1e6 / (ITERS * READS));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 2, 4}) run<0, 24>("ds_read_b64_tr_b16", w);
    for (int w : {1, 2, 4}) run<0, 12>("ds_read_b64_tr_b16", w);
    for (int w : {1, 2, 4}) run<1, 12>("ds_read_b128", w);
    return 0;
}
