// Fast erf-GELU / GELU' for the bf16 kernels (nn.GELU, reference lib/model/DSTformer.py:70,79-85): shared by the GEMM epilogues
// (gemm_pipe.hip) and the fused MLP forward (mlp_fused.hip).  tests/test_kernel_constants.py reads the literals of this file.
#pragma once
#include "mbx_common.h"

// erf-GELU for the bf16 path: Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7, far below bf16 resolution);
// erf(u / sqrt 2) and the Gaussian of GELU' share one exponential, exp(-u^2 / 2).
__device__ __forceinline__ void erf_parts(float u, float& erf_v, float& gauss) {
    const float x = fabsf(u) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));   // v_rcp_f32 (1 ulp); __frcp_rn expands to a 10-instruction IEEE division
    gauss = __expf(-0.5f * u * u);
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    erf_v = copysignf(fmaf(-poly, gauss, 1.0f), u);
}
__device__ __forceinline__ float gelu_fast(float u) {
    float e, g;
    erf_parts(u, e, g);
    return 0.5f * u * (1.0f + e);
}
// GELU of two values at once for the forward epilogue of the 256x256 kernel, which is VALU-bound (~80 issue cycles per element with
// gelu_fast: two quarter-rate transcendentals and six unpacked operations).  Abramowitz-Stegun 7.1.28,
//     erf(x) = 1 - (1 + a1 x + ... + a6 x^6)^-16,  |error| <= 3e-7,
// needs no exponential, and every other operation is a packed-fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32):
//     gelu(u) = u Phi(u) = (u + |u| erf(|u| / sqrt 2)) / 2 = ((u + a) - a r^16) / 2,   a = |u|,  r = 1 / D(a),
// with 2^(-k/2) folded into the coefficients of D.  ~54 issue cycles per element; fp32 evaluation error 7e-7 absolute.
__device__ __forceinline__ mbx_f32x2_t gelu_fast2(mbx_f32x2_t u) {
    const mbx_f32x2_t a = {fabsf(u[0]), fabsf(u[1])};
    mbx_f32x2_t d = a * 5.382975e-06f + 4.8890636e-05f;      // a6 / 8, a5 / 2^2.5
    d = d * a + 3.8003575e-05f;                               // a4 / 4
    d = d * a + 3.2776264e-03f;                               // a3 / 2^1.5
    d = d * a + 2.1141006e-02f;                               // a2 / 2
    d = d * a + 4.9867347e-02f;                               // a1 / sqrt 2
    d = d * a + 1.0f;
    mbx_f32x2_t rr = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    rr = rr * rr; rr = rr * rr; rr = rr * rr; rr = rr * rr;   // r^16
    return ((u + a) - a * rr) * 0.5f;
}
// GELU'(u) = Phi(u) + u phi(u) of two values at once (the GELU' epilogue): the 7.1.26 form of gelu_fast_grad with every
// non-transcendental operation packed; erf(|u| / sqrt 2) gets its sign back with v_bfi.
__device__ __forceinline__ mbx_f32x2_t gelu_fast_grad2(mbx_f32x2_t u) {
    const mbx_f32x2_t a = {fabsf(u[0]), fabsf(u[1])};
    const mbx_f32x2_t den = a * 0.23164189f + 1.0f;          // 0.3275911 / sqrt 2
    const mbx_f32x2_t t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    const mbx_f32x2_t w = (u * u) * -0.72134752044448170368f;   // -u^2 / 2 in base-2 units
    const mbx_f32x2_t gauss = {__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])};
    mbx_f32x2_t poly = t * 1.061405429f + -1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t + -0.284496736f;
    poly = poly * t + 0.254829592f;
    poly = poly * t;
    const mbx_f32x2_t e = 1.0f - poly * gauss;               // erf(|u| / sqrt 2)
    const mbx_f32x2_t se = {copysignf(e[0], u[0]), copysignf(e[1], u[1])};
    return (u * gauss) * 0.39894228040143267794f + (se * 0.5f + 0.5f);
}
// GELU and GELU' of two values at once (the fc1 epilogue that saves the derivative instead of the pre-activation): both from the
// 7.1.26 parts of gelu_fast_grad2 -- gelu = u (1 + erf) / 2, gelu' = (1 + erf) / 2 + u phi(u).
__device__ __forceinline__ void gelu_fast_both2(mbx_f32x2_t u, mbx_f32x2_t& gl, mbx_f32x2_t& gr) {
    const mbx_f32x2_t a = {fabsf(u[0]), fabsf(u[1])};
    const mbx_f32x2_t den = a * 0.23164189f + 1.0f;
    const mbx_f32x2_t t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    const mbx_f32x2_t w = (u * u) * -0.72134752044448170368f;
    const mbx_f32x2_t gauss = {__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])};
    mbx_f32x2_t poly = t * 1.061405429f + -1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t + -0.284496736f;
    poly = poly * t + 0.254829592f;
    poly = poly * t;
    const mbx_f32x2_t e = 1.0f - poly * gauss;
    const mbx_f32x2_t se = {copysignf(e[0], u[0]), copysignf(e[1], u[1])};
    const mbx_f32x2_t phi = se * 0.5f + 0.5f;                 // Phi(u)
    gl = u * phi;
    gr = (u * gauss) * 0.39894228040143267794f + phi;
}
__device__ __forceinline__ float gelu_fast_grad(float u) {
    float e, g;
    erf_parts(u, e, g);
    return fmaf(u * g, 0.39894228040143267794f, 0.5f * (1.0f + e));
}
