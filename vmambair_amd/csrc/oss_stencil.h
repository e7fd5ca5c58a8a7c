// oss_stencil.h -- 3x3-stencil access helpers shared by the depth-wise convolutions (oss_dwconv.hip) and the thin dense 3x3
// convolutions (oss_conv3x3_thin.hip): 8 consecutive pixels of one image row per lane (16-bit I/O; float since round 4).
#pragma once
#include "oss_device.h"

namespace oss {

// One 16-byte access per lane and image row, and the two halo pixels of a lane's 8-pixel group come from the
// neighbouring lanes by DPP (wave_shr / wave_shl) instead of two more (2-byte) loads: 3 load instructions per 8 outputs
// against 9 per 4 in the kernel above, which is what bounds it (2-byte-per-lane global accesses move 128 B per wave
// instruction on gfx950).  Needs W % 8 == 0 with W / 8 (lanes per image row) dividing 64, so that a row's groups never
// straddle a wave, and 16-byte aligned planes.
// (round 4) Raw8<T>: the 8 pixels of a lane as they lie in memory -- one 16-byte word for the 16-bit types, two for float (the fused
// forms took 16-bit I/O only through round 3; the reference trains in fp32).  Loads return Raw8 without touching the value, so that
// several rows can be in flight before the first is unpacked.
template <typename T> struct Raw8 { u32x4 q[sizeof(T) == 4 ? 2 : 1]; };
template <typename T>
__device__ __forceinline__ Raw8<T> load8_raw(const T *p) {
    Raw8<T> r;
    r.q[0] = *reinterpret_cast<const u32x4 *>(p);
    if constexpr (sizeof(T) == 4) r.q[1] = *reinterpret_cast<const u32x4 *>(p + 4);
    return r;
}
template <typename T>
__device__ __forceinline__ void store8_raw(T *p, const Raw8<T> &r) {
    *reinterpret_cast<u32x4 *>(p) = r.q[0];
    if constexpr (sizeof(T) == 4) *reinterpret_cast<u32x4 *>(p + 4) = r.q[1];
}
template <typename T>
__device__ __forceinline__ Raw8<T> zero8() {
    Raw8<T> r;
    r.q[0] = u32x4{0u, 0u, 0u, 0u};
    if constexpr (sizeof(T) == 4) r.q[1] = u32x4{0u, 0u, 0u, 0u};
    return r;
}
template <typename T>
__device__ __forceinline__ void unpack8(const Raw8<T> &r, float (&v)[8]) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = __uint_as_float(r.q[0][j]); v[4 + j] = __uint_as_float(r.q[1][j]); }
    } else {
        unpack2<T>(r.q[0].x, v[0], v[1]); unpack2<T>(r.q[0].y, v[2], v[3]); unpack2<T>(r.q[0].z, v[4], v[5]); unpack2<T>(r.q[0].w, v[6], v[7]);
    }
}
template <typename T>
__device__ __forceinline__ Raw8<T> pack8(const float (&v)[8]) {
    Raw8<T> r;
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { r.q[0][j] = __float_as_uint(v[j]); r.q[1][j] = __float_as_uint(v[4 + j]); }
    } else {
        r.q[0] = u32x4{pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7])};
    }
    return r;
}
template <typename T>
__device__ __forceinline__ void load8(const T *p, float (&v)[8]) { unpack8<T>(load8_raw<T>(p), v); }
template <typename T>
__device__ __forceinline__ void store8(T *p, const float (&v)[8]) { store8_raw<T>(p, pack8<T>(v)); }
// v[0..9] = pixels w0-1 .. w0+8 of image row h + dy (zeros outside the image); every lane of the wave must call it.
// EDGE (round 4): rows whose W / 8 lane groups do not tile a wave (W = 160: 20 groups; the RealSR tiles, the Deraining tree's
// progressive patch sizes 160 / 192 / 320 / 384) straddle waves, and DPP cannot reach across one: the first lane of a wave
// that is not the first group of its row (and the last lane that is not the last group) fetches its one halo pixel from memory
// -- two one-lane loads per wave and row instead of falling back to the 4-pixel kernel (three times the load instructions).
template <typename T, bool EDGE = false>
__device__ __forceinline__ void row10(const T *plane, int h, int dy, int H, int W, int w0, bool first, bool last, float (&v)[10]) {
    const int hh = h + dy;
    const bool ok = hh >= 0 && hh < H;
    float m[8];
    const T *rowp = plane + (int64_t)(ok ? hh : h) * W + w0;
    Raw8<T> q = load8_raw<T>(rowp);
    if (!ok) q = zero8<T>();   // a row outside the image: selects on the packed words instead of eight on the values
    unpack8<T>(q, m);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j + 1] = m[j];
    v[0] = shift_from_prev_lane(v[8], 0.f, first);
    v[9] = shift_from_next_lane(v[1], 0.f, last);
    if constexpr (EDGE) {
        const int lane = threadIdx.x & 63;
        if (lane == 0 && !first) v[0] = ok ? to_f32(rowp[-1]) : 0.f;
        if (lane == 63 && !last) v[9] = ok ? to_f32(rowp[8]) : 0.f;
    }
}

// row10 in two halves, for kernels that put several rows' loads in flight before the first one is used (non-EDGE widths only):
// row10_issue returns the raw bytes of image row h + dy (of row h when that row is outside the image) -- no use of the loaded
// value, so no wait; row10_finish turns them into v[0..9] exactly as row10 does (zeros for a row outside the image, halo by DPP).
template <typename T>
__device__ __forceinline__ Raw8<T> row10_issue(const T *plane, int h, int dy, int H, int W, int w0) {
    const int hh = h + dy;
    return load8_raw<T>(plane + (int64_t)((hh >= 0 && hh < H) ? hh : h) * W + w0);
}
template <typename T>
__device__ __forceinline__ void row10_finish(Raw8<T> q, int h, int dy, int H, bool first, bool last, float (&v)[10]) {
    const int hh = h + dy;
    if (!(hh >= 0 && hh < H)) q = zero8<T>();
    float m[8];
    unpack8<T>(q, m);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j + 1] = m[j];
    v[0] = shift_from_prev_lane(v[8], 0.f, first);
    v[9] = shift_from_next_lane(v[1], 0.f, last);
}

// lanes per image row do not tile a wave -> the EDGE instantiations (rows straddle waves)
static inline bool stencil_edge(int W) { return (64 % (W / 8)) != 0; }

// Phi(a) and phi(a) of the exact (erf) gelu from ONE exponential: erf(|z|) = 1 - (a1 t + ... + a5 t^5) exp(-z^2),
// t = 1 / (1 + p |z|)  (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 -- fp32 round-off level, four orders of magnitude under the
// rounding of the 16-bit tensors these kernels read and write), and with z = a / sqrt(2) the exponential is the one of
// phi(a) = exp(-a^2 / 2) / sqrt(2 pi).  The library erff costs ~35 instructions behind a divergent branch; this is 12.
__device__ __forceinline__ void gelu_parts(float a, float &cdf, float &pdf) {
    const float e = exp2_hw(-0.5f * a * a * kLog2e);
    const float z = fabsf(a) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.f));
    float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    const float half_tail = 0.5f * poly * t * e;          // (1 - erf|z|) / 2
    cdf = a >= 0.f ? 1.f - half_tail : half_tail;
    pdf = 0.3989422804014327f * e;
}


}  // namespace oss
