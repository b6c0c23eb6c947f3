// lane_reduce.h -- wave-wide sums without LDS traffic (DPP row rotations + v_permlane{16,32}_swap) and two-instruction
// complex multiply-accumulates, shared by the panel tridiagonalisation (tridiag_panel.hip) and the block-reflector
// back-transform (wy_apply.hip).
#pragma once
#include "common.h"

namespace admmnet {

// sum over the 16 lanes of a DPP row (every lane gets the sum)
__device__ __forceinline__ float pn_row16_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xF, 0xF, false));
    return x;
}
// sum over the four 16-lane rows of the wave, lane by lane (x of lanes l, l ^ 16, l ^ 32, l ^ 48): every lane gets it
// (inline asm on purpose: the compiler's __builtin_amdgcn_permlane16/32_swap returned wrong sums here -- with identical
//  operands it folds the two results into one, and the hidden-copy workaround still failed the n = 257 test)
__device__ __forceinline__ float pn_group_sum(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    x = a + b;
    a = x;
    b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
// the same for two values at once (re / im): the volatile asm blocks of two separate calls cannot overlap, one block
// with both swaps can
__device__ __forceinline__ void pn_group_sum2(float &x, float &y) {
    float a = x, b = x, c = y, d = y;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    x = a + b;
    y = c + d;
    a = x;
    b = x;
    c = y;
    d = y;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    x = a + b;
    y = c + d;
}
typedef float v2f __attribute__((ext_vector_type(2)));
// Complex multiply-accumulate in TWO packed instructions (the scalar form takes four): v_pk_fma_f32 with op_sel picking
// the real / imaginary half of each operand pair and neg_lo / neg_hi the sign -- the compiler folds the broadcast of the
// first product but materialises the swapped, negated operand of the second (v_xor + v_mov), hence the asm.  The skinny
// phases are bound by the VALU issue of ONE wave per SIMD (a wave64 instruction holds the 16-lane SIMD for 4 cycles).
__device__ __forceinline__ v2f pk_cfma(v2f acc, v2f a, v2f b) {        // acc + a b
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "+v"(acc)
        : "v"(a), "v"(b));
    return acc;
}
__device__ __forceinline__ v2f pk_cfma_conj(v2f acc, v2f a, v2f b) {   // acc + conj(a) b
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]"
        : "+v"(acc)
        : "v"(a), "v"(b));
    return acc;
}
__device__ __forceinline__ v2f pk2(float2 a) { return v2f{a.x, a.y}; }
}  // namespace admmnet
