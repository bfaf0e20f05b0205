// Shared device/host helpers for the gfx950 kernel set of libmbx.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mbx.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits (the C ABI passes void*; T is chosen by `dtype`)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA bf16 A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define MBX_WAVE 64

// ---------------------------------------------------------------- error plumbing (thread-local)
int mbx_set_error(const char* fmt, ...);
#define MBX_CHECK_ARG(cond, ...)                    \
    do {                                            \
        if (!(cond)) return mbx_set_error(__VA_ARGS__); \
    } while (0)
#define MBX_LAUNCH_CHECK(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) return mbx_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------- per-device launch properties (elementwise.hip)
// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device) instead of once per launch (host time on the B = 1 latency
// path), and the CU count of the CURRENT device: both remembered in lock-free per-device tables -- launches come from the
// main thread, the autograd engine's thread and DataParallel's replica threads, possibly for different devices.
int mbx_set_dyn_lds(const void* kernel, size_t bytes, const char* who);
int mbx_cu_count();

// ---------------------------------------------------------------- diagnostics switches
// The product library reads NO environment variables: every kernel choice is fixed at build time.  A/B variants and
// ablation switches exist only in -DMBX_DIAG builds (tools/build_variants.py -> tools/variants/libmbx_*.so, loaded by the
// measurement scripts through MBX_LIB), where mbx_env_int() consults the environment.
#ifdef MBX_DIAG
#include <stdlib.h>
static inline int mbx_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
#define mbx_env_int(name, dflt) (dflt)
#endif

// ---------------------------------------------------------------- scalar conversions
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round-to-nearest-even: the compiler lowers these conversions to v_cvt_pk_bf16_f32 on gfx950
typedef __bf16 mbx_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float mbx_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const mbx_f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, mbx_bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }
// c + a.lo * b.lo + a.hi * b.hi on two packed bf16 pairs (v_dot2c_f32_bf16): row dots over bf16 data without unpacking
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(mbx_bf16x2_t, a), __builtin_bit_cast(mbx_bf16x2_t, b), c, false);
}

template <typename T> struct Cvt;
template <> struct Cvt<float> {
    static __device__ __forceinline__ float to_f(float v) { return v; }
    static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct Cvt<bf16_t> {
    static __device__ __forceinline__ float to_f(bf16_t v) { return bf2f(v); }
    static __device__ __forceinline__ bf16_t from_f(float v) { return f2bf(v); }
};

// 4 consecutive T elements <-> float[4] (16-B / 8-B vector access; pointers must be aligned to 4 elements)
template <typename T> __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
}
// the bf16x3 operand split of four fp32 values, written where a producer would otherwise leave fp32 for mbx_split_bf16 to read again:
// hi = bf16(v), lo = bf16(v - hi) (the arithmetic of split_bf16_kernel, bit for bit)
__device__ __forceinline__ void store4_planes(bf16_t* hi, bf16_t* lo, const float (&v)[4]) {
    float h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { h[e] = bf2f(f2bf(v[e])); l[e] = v[e] - h[e]; }
    store4<bf16_t>(hi, h);
    store4<bf16_t>(lo, l);
}

// ---------------------------------------------------------------- wave-level reductions (64 lanes)
// VALU-only reduction: four DPP butterfly steps inside each 16-lane row (quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror), then row_bcast:15 / row_bcast:31 carry the row sums upward so that lane 63 holds the total, which
// v_readlane broadcasts.  No ds_bpermute: measured on MI355X, ds_bpermute-based reductions returned a wrong sum for
// about one wave per 10^6 when a second stream kept other kernels resident on the same CUs (tools/determinism_check.py).
// Every lane of the wave must be active.
template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ float dpp_mov(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                                 ROW_MASK, 0xf, false));
}
struct WaveAdd {
    static __device__ __forceinline__ float op(float a, float b) { return a + b; }
    static __device__ __forceinline__ float identity(float) { return 0.f; }
};
struct WaveMax {
    static __device__ __forceinline__ float op(float a, float b) { return fmaxf(a, b); }
    static __device__ __forceinline__ float identity(float v) { return v; }
};
template <typename Op> __device__ __forceinline__ float wave_reduce(float v) {
    v = Op::op(v, dpp_mov<0xB1>(v, v));    // quad_perm [1,0,3,2]
    v = Op::op(v, dpp_mov<0x4E>(v, v));    // quad_perm [2,3,0,1]
    v = Op::op(v, dpp_mov<0x141>(v, v));   // row_half_mirror
    v = Op::op(v, dpp_mov<0x140>(v, v));   // row_mirror: every lane now holds its row's result
    v = Op::op(v, dpp_mov<0x142, 0xa>(Op::identity(v), v));   // row_bcast:15 into rows 1 and 3
    v = Op::op(v, dpp_mov<0x143, 0xc>(Op::identity(v), v));   // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// op(v[lane], v[lane ^ 32]) on every lane: v_permlane32_swap exchanges the upper half of its first operand with the
// lower half of the second, so {v, v} becomes {lo, lo} and {hi, hi}.  Inline asm because the clang builtin of ROCm 7.2
// returns the first register for both results (tools/probes/wave_reduce.hip); the s_nop covers the VALU-write hazard.
template <typename Op> __device__ __forceinline__ float wave_halves(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return Op::op(a, b);
}
// v[lane & 31] on every lane: the lower half wave's value, broadcast to the upper half (one v_permlane32_swap)
__device__ __forceinline__ float wave_lower_half(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a;
}
__device__ __forceinline__ float wave_sum(float v) { return wave_reduce<WaveAdd>(v); }
__device__ __forceinline__ float wave_max(float v) { return wave_reduce<WaveMax>(v); }

// ---------------------------------------------------------------- counter-based dropout masks (restated in motionbert_amd/dropmask.py)
// element index = 64 bit, passed as its two halves (the 4 elements a thread owns differ only in the low two bits)
__device__ __forceinline__ bool drop_keep(uint32_t seed_lo, uint32_t seed_hi, uint32_t idx_lo, uint32_t idx_hi, uint32_t thresh) {
    // two rounds of a 32-bit multiply-xorshift mix over (idx, seed); keep <=> the 32-bit hash >= p * 2^32
    uint32_t h = idx_lo * 0x9E3779B1u ^ seed_lo;
    h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13;
    h += idx_hi * 0xC2B2AE3Du + seed_hi;
    h ^= h >> 16; h *= 0x27D4EB2Fu; h ^= h >> 15;
    return h >= thresh;
}
// attention-probability dropout (nn.Dropout on the softmax output, DSTformer.py:96,182,196): thresh = 0 -> off
struct MbxDrop {
    uint32_t seed_lo, seed_hi, thresh;
    float scale;      // 1 / (1 - p)
};

// ---------------------------------------------------------------- activation math (fp32)
__device__ __forceinline__ float gelu_erf(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float u) {
    return 0.5f * (1.0f + erff(u * 0.70710678118654752440f)) + u * __expf(-0.5f * u * u) * 0.39894228040143267794f;
}

// ---------------------------------------------------------------- column-sum finalize (shared by all reductions)
// out[c] = sum_{p < nparts} part[p * stride + col0 + c],  c < ncols
int mbx_launch_colsum(const float* part, int nparts, int stride, int col0, int ncols, float* out, hipStream_t s);

// ---------------------------------------------------------------- pipelined bf16 GEMMs (gemm_pipe.hip)
int mbx_launch_gemm_nt_pipe(const void* a, const void* w, const float* bias, int epi, void* out_t, void* out2_t, float* out_f,
                            const float* resid, const void* aux, int M, int N, int K, hipStream_t s);
size_t mbx_gemm_tn_pipe_ws(int M, int N, int K);
int mbx_launch_gemm_tn_pipe(const void* dy, const void* a, float* dw, float* db, int M, int N, int K, void* ws, hipStream_t s);
int mbx_launch_gemm_nt_x3(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias, int epi,
                          float* out_t, float* out2_t, float* out_f, const float* resid, const float* aux, int M, int N, int K,
                          hipStream_t s, void* pl_hi = nullptr, void* pl_lo = nullptr);
size_t mbx_gemm_tn_x3_ws(int M, int N, int K);
int mbx_launch_gemm_tn_x3(const void* dy_hi, const void* dy_lo, const void* a_hi, const void* a_lo, float* dw, float* db, int M, int N,
                          int K, void* ws, hipStream_t s);
bool mbx_use_v1_gemm();   // MBX_GEMM_V1=1 selects the simple double-buffered kernels of gemm.hip (A/B testing)
