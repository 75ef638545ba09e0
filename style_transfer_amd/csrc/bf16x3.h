// fp32-class products on the bf16 matrix cores (gfx950): operand split and the six-product step
// shared by the Gram and SYMM kernels.
//
// x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (round to nearest
// even; the residuals are exact in fp32): 3 x 8 significand bits cover fp32's 24.  Of the nine
// piece products the six down to 2^-16 relative are kept -- x1y1, x1y2, x2y1, x1y3, x3y1, x2y2 --
// what is dropped is below 2^-24 |x||y|, the rounding of one fp32 product.  Every bf16 x bf16
// product is exact in fp32, so the only roundings are the accumulator's, the same class as the
// fp32 MFMA's (tools/bf16x3_numerics.py: 1.5e-7 .. 3.6e-7 of max against float64 on the Gram /
// SYMM shapes, the fp32 chain itself 1.5e-7 .. 3.1e-7).  v_mfma_f32_32x32x16_bf16 retires 16 k
// per 32 cycles where v_mfma_f32_32x32x2_f32 needs 8 x 64: six of them are 2.7x less matrix time
// than the fp32 form, and -- unlike the fp32 MFMA, which runs at and on the vector rate -- bf16
// MFMAs leave the vector pipe to the split itself (5.5 instructions per element).
#pragma once

#include <hip/hip_runtime.h>

namespace stx {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16b __attribute__((ext_vector_type(16)));

// x[0..7] -> three bf16x8 fragments (element j of a fragment = piece of x[j])
__device__ __forceinline__ void split3_bf16(const float (&x)[8], bf16x8 &p1, bf16x8 &p2, bf16x8 &p3) {
    unsigned q1[4], q2[4], q3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x[2 * j], b = x[2 * j + 1];
        unsigned u1, u2, u3;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u1) : "v"(a), "v"(b));
        const float ra = a - __builtin_bit_cast(float, u1 << 16);
        const float rb = b - __builtin_bit_cast(float, u1 & 0xffff0000u);
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u2) : "v"(ra), "v"(rb));
        const float sa = ra - __builtin_bit_cast(float, u2 << 16);
        const float sb = rb - __builtin_bit_cast(float, u2 & 0xffff0000u);
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u3) : "v"(sa), "v"(sb));
        q1[j] = u1, q2[j] = u2, q3[j] = u3;
    }
    typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
    p1 = __builtin_bit_cast(bf16x8, (u32x4s){q1[0], q1[1], q1[2], q1[3]});
    p2 = __builtin_bit_cast(bf16x8, (u32x4s){q2[0], q2[1], q2[2], q2[3]});
    p3 = __builtin_bit_cast(bf16x8, (u32x4s){q3[0], q3[1], q3[2], q3[3]});
}

// one value -> its three pieces as raw bf16 bit patterns (host-free scalar form of the above)
__device__ __forceinline__ void split3_bf16_scalar(float x, unsigned short &s1, unsigned short &s2,
                                                   unsigned short &s3) {
    unsigned u1, u2, u3;
    asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(u1) : "v"(x));
    const float r = x - __builtin_bit_cast(float, u1 << 16);
    asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(u2) : "v"(r));
    const float t = r - __builtin_bit_cast(float, u2 << 16);
    asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(u3) : "v"(t));
    s1 = (unsigned short)(u1 & 0xffffu), s2 = (unsigned short)(u2 & 0xffffu), s3 = (unsigned short)(u3 & 0xffffu);
}

// acc += a b^T over the 16 k of a step: six products, smallest first
__device__ __forceinline__ f32x16b mfma_split6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16b acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

}  // namespace stx
