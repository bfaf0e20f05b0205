// Empirical probe of ds_read_b64_tr_b16 on gfx950: fills LDS with element indices and dumps what
// each lane receives for a few per-lane address patterns.  Output: text table on stdout.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint16_t* out, int pattern) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    uint32_t addr;
    if (pattern == 0) addr = 0;                                   // uniform
    else if (pattern == 1) addr = l * 8;                          // lane-linear 8 B
    else if (pattern == 2) addr = (l & 15) * 128 + (l >> 4) * 8;  // 16 rows of 64 elements, 4-element column blocks
    else addr = (l & 15) * 2 + (l >> 4) * 128;                    // guide formula: (l&15) + (l>>4)*64 elements
    uint32_t base = (uint32_t)(uintptr_t)lds;  // LDS offset
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr) : "memory");
    out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int p = 0; p < 4; ++p) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d\n", p);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
