// oss_mfma.h -- v_mfma_f32_32x32x16_{bf16,f16} wrapper shared by the GEMM-shaped kernels (oss_conv1x1.hip,
// oss_proj.hip).  Operand maps (cdna_hip_programming.md section 3): lane l holds A[i = l & 31][k = 8 (l >> 5) .. +8]
// and B[k = 8 (l >> 5) .. +8][j = l & 31]; D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31], r in [0, 16).
#pragma once
#include "oss_device.h"

namespace oss {

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
    static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a),
                                                       __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b), c, 0, 0, 0);
    }
};
template <> struct Mfma<f16_t> {
    static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) _Float16, a),
                                                      __builtin_bit_cast(__attribute__((ext_vector_type(8))) _Float16, b), c, 0, 0, 0);
    }
};

template <typename T> __device__ __forceinline__ short to_bits(float v) { return (short)from_f32<T>(v).v; }

// 8 consecutive fp32 -> one 8 x 16-bit MFMA operand (v_cvt_pk_* on gfx950)
template <typename T> __device__ __forceinline__ s16x8 cvt8(const f32x4 &lo, const f32x4 &hi) {
    u32x4 q = {pack2<T>(lo.x, lo.y), pack2<T>(lo.z, lo.w), pack2<T>(hi.x, hi.y), pack2<T>(hi.z, hi.w)};
    return __builtin_bit_cast(s16x8, q);
}

}  // namespace oss
