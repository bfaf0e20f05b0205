// Pieces shared by the "row owner" kernels (mlp_fused.hip, gemm_rows.hip): the LDS-DMA weight stream as asm statements, fragment
// reads with one address register per stage, and the MFMA statements with explicit register classes.
#pragma once
#include "mbx_common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
// LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes -> 1 KiB of LDS at the wave-uniform address in M0) as an asm statement.
// Through the builtin hipcc books every DMA as an LDS event of a second kind: it then waits lgkmcnt(0) instead of a counted
// lgkmcnt(N) in front of every MFMA (the fragment prefetch would be drained seven slots out of eight) and vmcnt(0) in front of LDS
// reads it cannot tell apart from the DMA's destination.  As asm the DMA is invisible to that pass; the kernel owns its vmcnt
// (the counted waits of MF_SYNC) -- cdna_hip_programming.md 5.7: M0 is written and restored inside the statement.
__device__ __forceinline__ void glds16(const void* src, const void* lds_dst) {
    unsigned keep;
    const unsigned dst = (unsigned)(uintptr_t)(const lds_void_t*)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
#define GLDS16(src, dst) glds16((src), (dst))
// the same with the non-temporal policy: for one-touch activation rows, so that they do not push the weight stream (which every CU of
// the XCD re-reads) out of the 4 MiB L2
__device__ __forceinline__ void glds16_nt(const void* src, const void* lds_dst) {
    unsigned keep;
    const unsigned dst = (unsigned)(uintptr_t)(const lds_void_t*)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
// The weight stream's form: wave-uniform 64-bit base in SGPRs + this lane's constant 32-bit byte offset in a VGPR (no per-piece
// vector address arithmetic), M0 written but not restored -- nothing the compiler emits in this kernel reads M0 (gfx9+ LDS
// instructions do not), and every DMA statement sets it itself.
__device__ __forceinline__ void glds16_s(const char* base, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
// 16 bytes from LDS at a byte address kept in a register the compiler cannot fold a constant into (so the fragment reads of a stage
// are `ds_read_b128 v, base offset:imm`: one address register per stage instead of one v_add per read)
typedef __attribute__((address_space(3))) const uint32_t lds_u32_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));   // one MFMA operand fragment (8 bf16)
__device__ __forceinline__ u32x4_t lds_read16(unsigned base, int imm) {
    return *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t*>(base + imm);
}

// MFMA statements.  D = A(weights fragment, VGPR) x B(token fragment) + C.  fc1: accumulator in VGPRs (the GELU reads it), token
// operand X in accumulator registers; fc2: accumulator in accumulator registers, token operand = the packed hidden (VGPR).
// MFMA_FC1_Z starts a chain from zero (no register initialisation, hence no write -> MFMA wait state to own).
#define MFMA_FC1_Z(acc_, w_, x_) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc_) : "v"(w_), "v"(x_))
#define MFMA_FC1(acc_, w_, x_) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc_) : "v"(w_), "v"(x_))
#define MFMA_FC2(acc_, w_, g_) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc_) : "v"(w_), "v"(g_))
// wait states between the last MFMA that writes a register and its first non-MFMA reader / writer (8-pass XDL: 12; 16 taken)
#define MFMA_PAD_V(a_, b_) asm volatile("s_nop 15" : "+v"(a_), "+v"(b_))
#define MFMA_PAD_A(a_) asm volatile("s_nop 15" : "+a"(a_))

