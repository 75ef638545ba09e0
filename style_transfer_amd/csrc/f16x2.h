// fp32-class products on the fp16 matrix cores (gfx950): the two-piece operand split and the
// three-product step shared by the fp16-split convolution (conv_h2.hip), Gram and SYMM kernels.
//
// a = hi + lo with hi = fp16(s a), lo = fp16(s a - hi) (round to nearest even; the residual is
// exact in fp32): 2 x 11 significand bits.  s is a power of two chosen from the operand's own
// maximum (max |a| s in [2^13, 2^14)), so nothing can overflow and the scaling is exact.  A product
// is hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_f16 -- every fp16 x fp16 product is exact in the
// fp32 accumulator; lo lo, 2^-22 of the product, is dropped.  Against the three-piece bf16 form
// (bf16x3.h: six products, 5.5 vector instructions per element for the split) this is three
// products and two instructions per element: hi and lo are ONE v_fma_mix each.
// tools/f16x2_numerics.py has the error model (2e-7 .. 6e-7 of max against float64 on the
// convolution, Gram and SYMM shapes -- what an fp32 MFMA chain has on the same data).
#pragma once

#include <hip/hip_runtime.h>

namespace stx {

typedef _Float16 f16x8h __attribute__((ext_vector_type(8)));
typedef float f32x16h __attribute__((ext_vector_type(16)));

// power of two s (as an exponent) with amax * 2^s in [2^13, 2^14); amax given as float bits
__host__ __device__ inline int h2_scale_exp(unsigned amax_bits) {
    const int e = (int)((amax_bits >> 23) & 0xffu);      // biased exponent of the maximum
    int s = 13 - (e - 127);
    return s < -126 ? -126 : s > 127 ? 127 : s;          // (zero blobs: any scale does)
}
__device__ __forceinline__ float pow2f(int e) {          // 2^e, e clamped to the normal range
    const int b = e + 127;
    return __builtin_bit_cast(float, (unsigned)(b < 1 ? 1 : b > 254 ? 254 : b) << 23);
}

// x[0..7] -> the fragments hi, lo of s x (element j of a fragment = piece of x[j]).  One asm block
// per four values: a partially written register (op_sel destination) is read two instructions
// after its last write at the earliest.
__device__ __forceinline__ void split2_f16(const float (&x)[8], float s, f16x8h &hi, f16x8h &lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
        asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
            "v_fma_mixhi_f16 %0, %5, %8, 0\n\t"
            "v_fma_mixlo_f16 %1, %6, %8, 0\n\t"
            "v_fma_mixhi_f16 %1, %7, %8, 0\n\t"
            "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %2, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixlo_f16 %3, %6, %8, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %3, %7, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(h[2 * q]), "=&v"(h[2 * q + 1]), "=&v"(l[2 * q]), "=&v"(l[2 * q + 1])
            : "v"(x[4 * q]), "v"(x[4 * q + 1]), "v"(x[4 * q + 2]), "v"(x[4 * q + 3]), "v"(s));
    typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
    hi = __builtin_bit_cast(f16x8h, (u32x4s){h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(f16x8h, (u32x4s){l[0], l[1], l[2], l[3]});
}

// four values -> two dwords of hi pieces and two of lo pieces (the SYMM kernel's D staging)
__device__ __forceinline__ void split2_f16_quad(float x0, float x1, float x2, float x3, float s, unsigned (&h)[2],
                                                unsigned (&l)[2]) {
    asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
        "v_fma_mixhi_f16 %0, %5, %8, 0\n\t"
        "v_fma_mixlo_f16 %1, %6, %8, 0\n\t"
        "v_fma_mixhi_f16 %1, %7, %8, 0\n\t"
        "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %2, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %6, %8, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %7, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h[0]), "=&v"(h[1]), "=&v"(l[0]), "=&v"(l[1])
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(s));
}

// acc += a b^T over the 16 k of a step: three products, smallest first
__device__ __forceinline__ f32x16h mfma_split3(f16x8h a_hi, f16x8h a_lo, f16x8h b_hi, f16x8h b_lo, f32x16h acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc, 0, 0, 0);
    return acc;
}

// the largest of the kAmaxSlots words a producer left (float bits of max |x|), wave-uniform
__device__ __forceinline__ unsigned amax_of_slots(const unsigned *slots, int n) {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < n; ++i) m = max(m, slots[i]);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}

}  // namespace stx
