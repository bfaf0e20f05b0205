// Probe: what do the DPP controls and the gfx950 permlane swaps actually move?  Prints, for v = lane id, the source
// lane each step delivers, and the result of the full butterfly.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../motionbert_amd/csrc/mbx_common.h"
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    const float v = (float)lane;
    out[0 * 64 + lane] = dpp_mov<0xB1>(v, v);
    out[1 * 64 + lane] = dpp_mov<0x4E>(v, v);
    out[2 * 64 + lane] = dpp_mov<0x141>(v, v);
    out[3 * 64 + lane] = dpp_mov<0x140>(v, v);
    const float w = 100.f + lane;
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, w), false, false);
    out[4 * 64 + lane] = __builtin_bit_cast(float, r[0]);
    out[5 * 64 + lane] = __builtin_bit_cast(float, r[1]);
    r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, w), false, false);
    out[6 * 64 + lane] = __builtin_bit_cast(float, r[0]);
    out[7 * 64 + lane] = __builtin_bit_cast(float, r[1]);
    out[8 * 64 + lane] = wave_sum(v);                       // expect 2016 everywhere
    out[9 * 64 + lane] = wave_max((float)((lane * 37) % 64)); // expect 63 everywhere
    out[11 * 64 + lane] = dpp_mov<0x142, 0xa>(-1.f, v);
    out[12 * 64 + lane] = dpp_mov<0x143, 0xc>(-1.f, v);
    out[13 * 64 + lane] = wave_halves<WaveAdd>(v);   // expect lane + (lane ^ 32)
    out[10 * 64 + lane] = wave_sum((float)(1 << (lane % 20)) * (lane < 20 ? 1.f : 0.f));  // expect 2^20 - 1
}
int main() {
    float* d; hipMalloc(&d, 14 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[14 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[14] = {"quad[1,0,3,2]", "quad[2,3,0,1]", "row_half_mirror", "row_mirror", "swap16.r0", "swap16.r1", "swap32.r0", "swap32.r1", "wave_sum(lane)", "wave_max", "wave_sum(bits)", "row_bcast15.a", "row_bcast31.c", "wave_halves"};
    for (int r = 0; r < 14; ++r) { printf("%-16s", names[r]); for (int l = 0; l < 64; ++l) printf(" %g", h[r * 64 + l]); printf("\n"); }
    return 0;
}
