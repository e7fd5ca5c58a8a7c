// oss_layernorm.hip -- per-pixel LayerNorm over the channel axis of an NCHW tensor, NCHW in / NCHW out
// (norm1, norm2, out_norm, channel_norm of the OSS block).  The reference does it as
// rearrange 'b c h w -> b (h w) c' -> mean / var(unbiased=False) -> (x-mu)/sqrt(var+1e-5)*w+b ->
// rearrange back (SRGAN/VmambaIR/archs/MambaSISR6_arch.py:144-195): two permute copies and ~6
// elementwise/reduction kernels per call, three calls per block.  Here: one kernel forward, one kernel
// (+ a tiny finishing kernel for dweight/dbias) backward, no permutes.  HBM-bound.
//
// Workgroup = 64 consecutive pixels of one image x ALL channels; NW = 4 / 8 / 16 waves, wave w owns the
// channels c = w, w + NW, w + 2 NW, ... and lane l the pixel l of the tile, so every load / store
// instruction of a wave moves 64 consecutive pixels of one channel plane (coalesced) and the channel
// axis is spread over the waves (the 64x64 training patch has only 4096 pixels per image: a
// thread-per-pixel layout leaves half the chip idle and serialises 3 C strided loads per thread).
// A wave keeps its <= 24 channel values in registers (CPW = 24; C <= 384), so x is read from HBM
// once; wider C streams (re-reads hit L2).  Per-pixel sums cross the waves through LDS in a fixed
// order (bit-reproducible).  Two-pass variance (no E[x^2]-mu^2 cancellation).
// Optional fused epilogue: y *= silu(gate) (SS2D_1: y1 * act(z), MambaSISR6_arch.py:488-493).
#include "oss_device.h"
#include "oss_host.h"

namespace oss {

__device__ __forceinline__ float silu_f(float z) { return z * __builtin_amdgcn_rcpf(1.f + exp2_hw(-z * kLog2e)); }

constexpr int kLnMaxWaves = 16;

// sum over the workgroup's waves of one value per pixel (lane); fixed order
__device__ __forceinline__ float ln_cross_wave_sum(float *red /*[kLnMaxWaves][64]*/, float v, int wave, int lane, int nw) {
    red[wave * 64 + lane] = v;
    __syncthreads();
    float t = 0.f;
    for (int k = 0; k < nw; ++k) t += red[k * 64 + lane];
    return t;
}

// x: (B, C, P) with element strides (xsb, xsc), pixels contiguous.  y/gate: contiguous (B, C, P).
// CPW > 0: channels per wave held in registers (needs C <= CPW * NW); CPW == 0: streaming.
template <typename TX, typename TY, bool GATE, int CPW>
__global__ void __launch_bounds__(1024)
oss_ln_nchw_fwd_kernel(const TX *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                       const TY *__restrict__ gate, TY *__restrict__ y, float *__restrict__ mean_out,
                       float *__restrict__ rstd_out, int C, int P, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, float eps) {
    __shared__ float red[2][kLnMaxWaves * 64];
    constexpr int NI = CPW > 0 ? CPW : 1;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int p = blockIdx.x * 64 + lane;
    const bool ok = p < P;
    const int pc = ok ? p : P - 1;
    const TX *xp = x + b * xsb + pc;
    const TY *gp = GATE ? gate + b * gsb + pc : nullptr;
    const bool with_bias = bias != nullptr;
    float xv[NI], zv[NI];
    float s = 0.f;
    if constexpr (CPW > 0) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) {  // clamped index: all loads issue back to back
            const int c = wave + i * nw, cc = c < C ? c : C - 1;
            xv[i] = to_f32(xp[cc * xsc]);
            if constexpr (GATE) zv[i] = to_f32(gp[cc * gsc]);
        }
#pragma unroll
        for (int i = 0; i < CPW; ++i) s += (wave + i * nw < C) ? xv[i] : 0.f;
    } else {
        for (int c = wave; c < C; c += nw) s += to_f32(xp[c * xsc]);
    }
    const float mu = ln_cross_wave_sum(red[0], s, wave, lane, nw) / (float)C;
    float v = 0.f;
    if constexpr (CPW > 0) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) { const float d = (wave + i * nw < C) ? xv[i] - mu : 0.f; v = __builtin_fmaf(d, d, v); }
    } else {
        for (int c = wave; c < C; c += nw) { const float d = to_f32(xp[c * xsc]) - mu; v = __builtin_fmaf(d, d, v); }
    }
    const float rstd = 1.0f / sqrtf(ln_cross_wave_sum(red[1], v, wave, lane, nw) / (float)C + eps);
    if (wave == 0 && ok) { mean_out[(size_t)b * P + p] = mu; rstd_out[(size_t)b * P + p] = rstd; }
    TY *yp = y + (size_t)b * C * P + pc;
    auto emit = [&](int c, float xval, float zval) {
        // BiasFree (MambaSISR6_arch.py:160-164) divides x (not x - mu) by sigma
        float o = with_bias ? (xval - mu) * rstd * w[c] + bias[c] : xval * rstd * w[c];
        if constexpr (GATE) o *= silu_f(zval);
        if (ok) yp[(size_t)c * P] = from_f32<TY>(o);
    };
    if constexpr (CPW > 0) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) { const int c = wave + i * nw; if (c < C) emit(c, xv[i], GATE ? zv[i] : 0.f); }
    } else {
        for (int c = wave; c < C; c += nw) emit(c, to_f32(xp[c * xsc]), GATE ? to_f32(gp[c * gsc]) : 0.f);
    }
}

// dx, and per-workgroup partials of dweight / dbias (and dgate when GATE).
//   WithBias: xhat = (x-mu) rstd;  g = dy w;  dx = rstd (g - mean(g) - xhat mean(g xhat))
//   BiasFree: y = x rstd w, rstd = (var+eps)^-1/2 with var around mu:
//             dx = rstd g - (x - mu) rstd^3 mean(g x)
// A channel belongs to one wave, so its dweight / dbias partial is a plain 64-lane sum, stored straight
// to part[blk][2][C].
template <typename TX, typename TY, bool GATE, int CPW>
__global__ void __launch_bounds__(1024)
oss_ln_nchw_bwd_kernel(const TX *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                       const TY *__restrict__ gate, const TY *__restrict__ dy, const float *__restrict__ mean_in,
                       const float *__restrict__ rstd_in, TX *__restrict__ dx, TY *__restrict__ dgate,
                       float *__restrict__ part /*[nblk][2][C]*/, int C, int P, int64_t xsb, int64_t xsc, int64_t gsb,
                       int64_t gsc) {
    __shared__ float red[2][kLnMaxWaves * 64];
    constexpr int NI = CPW > 0 ? CPW : 1;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int p = blockIdx.x * 64 + lane;
    const bool ok = p < P;
    const int pc = ok ? p : P - 1;
    const TX *xp = x + b * xsb + pc;
    const TY *gyp = dy + (size_t)b * C * P + pc;
    const TY *gp = GATE ? gate + b * gsb + pc : nullptr;
    const bool with_bias = bias != nullptr;
    const float mu = mean_in[(size_t)b * P + pc], rstd = rstd_in[(size_t)b * P + pc];
    const float okf = ok ? 1.f : 0.f;
    float *pw_out = part + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
    float xv[NI], gv[NI], zv[NI];  // x, dy, gate
    float s1 = 0.f, s2 = 0.f;
    auto first = [&](int c, float xval, float gy, float zval) {
        const float xh = with_bias ? (xval - mu) * rstd : xval * rstd;
        float g = gy * okf;
        if constexpr (GATE) g *= silu_f(zval);
        const float pw = segment_sum_to_last<64>(g * xh);
        const float pb = segment_sum_to_last<64>(g);
        if (lane == 63) { pw_out[c] = pw; pw_out[C + c] = pb; }
        const float gw = g * w[c];
        s1 += gw;
        s2 = __builtin_fmaf(gw, with_bias ? xh : xval, s2);
    };
    if constexpr (CPW > 0) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            const int c = wave + i * nw, cc = c < C ? c : C - 1;
            xv[i] = to_f32(xp[cc * xsc]);
            gv[i] = to_f32(gyp[(size_t)cc * P]);
            if constexpr (GATE) zv[i] = to_f32(gp[cc * gsc]);
        }
#pragma unroll
        for (int i = 0; i < CPW; ++i) { const int c = wave + i * nw; if (c < C) first(c, xv[i], gv[i], GATE ? zv[i] : 0.f); }
    } else {
        for (int c = wave; c < C; c += nw)
            first(c, to_f32(xp[c * xsc]), to_f32(gyp[(size_t)c * P]), GATE ? to_f32(gp[c * gsc]) : 0.f);
    }
    const float m1 = ln_cross_wave_sum(red[0], s1, wave, lane, nw) / (float)C;
    const float m2 = ln_cross_wave_sum(red[1], s2, wave, lane, nw) / (float)C;
    TX *dxp = dx + (size_t)b * C * P + pc;
    TY *dgp = GATE ? dgate + (size_t)b * C * P + pc : nullptr;
    auto second = [&](int c, float xval, float gy, float zval) {
        const float xh = (xval - mu) * rstd;
        float g = gy;
        if constexpr (GATE) {
            const float sg = __builtin_amdgcn_rcpf(1.f + exp2_hw(-zval * kLog2e));  // sigmoid(z)
            const float sl = zval * sg;
            const float o = with_bias ? xh * w[c] + bias[c] : xval * rstd * w[c];  // LN output before the gate
            if (ok) dgp[(size_t)c * P] = from_f32<TY>(gy * o * (sg + sl * (1.f - sg)));
            g *= sl;
        }
        const float gw = g * w[c];
        const float d = with_bias ? rstd * (gw - m1 - xh * m2) : rstd * gw - xh * rstd * rstd * m2;
        if (ok) dxp[(size_t)c * P] = from_f32<TX>(d);
    };
    if constexpr (CPW > 0) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) { const int c = wave + i * nw; if (c < C) second(c, xv[i], gv[i], GATE ? zv[i] : 0.f); }
    } else {
        for (int c = wave; c < C; c += nw)
            second(c, to_f32(xp[c * xsc]), to_f32(gyp[(size_t)c * P]), GATE ? to_f32(gp[c * gsc]) : 0.f);
    }
}

__global__ void __launch_bounds__(256)
oss_ln_nchw_bwd_finish(const float *__restrict__ part, float *__restrict__ dw, float *__restrict__ db, int nblk, int C) {
    // 16 outputs x 16 slices of the partial list per workgroup (2C / 16 workgroups: enough of them to pull the
    // partials at bandwidth); slices combined in a fixed order
    __shared__ float red[16][17];
    const int col = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + col;
    float s = 0.f;
    if (i < 2 * C)
        for (int k = slice; k < nblk; k += 16) s += part[(size_t)k * 2 * C + i];
    red[slice][col] = s;
    __syncthreads();
    if (slice == 0 && i < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][col];
        if (i < C) dw[i] = t;
        else if (db) db[i - C] = t;
    }
}

constexpr int kLnCPW = 24;

// waves per workgroup: the fewest of 4 / 8 / 16 that keep a wave's channels in registers
static int ln_waves(int C) { return C <= 4 * kLnCPW ? 4 : (C <= 8 * kLnCPW ? 8 : 16); }
static bool ln_cached(int C) { return C <= kLnMaxWaves * kLnCPW; }

size_t ln_nchw_bwd_partial_floats(int B, int C, int P) { return (size_t)((P + 63) / 64) * B * 2 * C; }

template <typename TX, typename TY>
static int ln_fwd_t(const void *x, const float *w, const float *bias, const void *gate, void *y, float *mean, float *rstd,
                    int B, int C, int P, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, float eps, hipStream_t s) {
    dim3 grid((P + 63) / 64, B), block(64 * ln_waves(C));
    const TX *xp = reinterpret_cast<const TX *>(x);
    const TY *gp = reinterpret_cast<const TY *>(gate);
    TY *yp = reinterpret_cast<TY *>(y);
    if (ln_cached(C)) {
        if (gate) hipLaunchKernelGGL((oss_ln_nchw_fwd_kernel<TX, TY, true, kLnCPW>), grid, block, 0, s, xp, w, bias, gp, yp, mean, rstd, C, P, xsb, xsc, gsb, gsc, eps);
        else      hipLaunchKernelGGL((oss_ln_nchw_fwd_kernel<TX, TY, false, kLnCPW>), grid, block, 0, s, xp, w, bias, gp, yp, mean, rstd, C, P, xsb, xsc, gsb, gsc, eps);
    } else {
        if (gate) hipLaunchKernelGGL((oss_ln_nchw_fwd_kernel<TX, TY, true, 0>), grid, block, 0, s, xp, w, bias, gp, yp, mean, rstd, C, P, xsb, xsc, gsb, gsc, eps);
        else      hipLaunchKernelGGL((oss_ln_nchw_fwd_kernel<TX, TY, false, 0>), grid, block, 0, s, xp, w, bias, gp, yp, mean, rstd, C, P, xsb, xsc, gsb, gsc, eps);
    }
    return (int)hipGetLastError();
}

template <typename TX, typename TY>
static int ln_bwd_t(const void *x, const float *w, const float *bias, const void *gate, const void *dy, const float *mean,
                    const float *rstd, void *dx, void *dgate, float *dw, float *db, float *part, int B, int C, int P,
                    int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, hipStream_t s) {
    dim3 grid((P + 63) / 64, B), block(64 * ln_waves(C));
    const int nblk = grid.x * grid.y;
    const TX *xp = reinterpret_cast<const TX *>(x);
    const TY *gp = reinterpret_cast<const TY *>(gate);
    const TY *dyp = reinterpret_cast<const TY *>(dy);
    TX *dxp = reinterpret_cast<TX *>(dx);
    TY *dgp = reinterpret_cast<TY *>(dgate);
    if (ln_cached(C)) {
        if (gate) hipLaunchKernelGGL((oss_ln_nchw_bwd_kernel<TX, TY, true, kLnCPW>), grid, block, 0, s, xp, w, bias, gp, dyp, mean, rstd, dxp, dgp, part, C, P, xsb, xsc, gsb, gsc);
        else      hipLaunchKernelGGL((oss_ln_nchw_bwd_kernel<TX, TY, false, kLnCPW>), grid, block, 0, s, xp, w, bias, gp, dyp, mean, rstd, dxp, dgp, part, C, P, xsb, xsc, gsb, gsc);
    } else {
        if (gate) hipLaunchKernelGGL((oss_ln_nchw_bwd_kernel<TX, TY, true, 0>), grid, block, 0, s, xp, w, bias, gp, dyp, mean, rstd, dxp, dgp, part, C, P, xsb, xsc, gsb, gsc);
        else      hipLaunchKernelGGL((oss_ln_nchw_bwd_kernel<TX, TY, false, 0>), grid, block, 0, s, xp, w, bias, gp, dyp, mean, rstd, dxp, dgp, part, C, P, xsb, xsc, gsb, gsc);
    }
    hipLaunchKernelGGL(oss_ln_nchw_bwd_finish, dim3((2 * C + 15) / 16), dim3(256), 0, s, part, dw, db, nblk, C);
    return (int)hipGetLastError();
}

#define OSS_LN_DISPATCH(FN, ...)                                                        \
    switch ((int)xt * 3 + (int)yt) {                                                    \
        case 0: return FN<float, float>(__VA_ARGS__);                                   \
        case 1: return FN<float, f16_t>(__VA_ARGS__);                                   \
        case 2: return FN<float, bf16_t>(__VA_ARGS__);                                  \
        case 3: return FN<f16_t, float>(__VA_ARGS__);                                   \
        case 4: return FN<f16_t, f16_t>(__VA_ARGS__);                                   \
        case 6: return FN<bf16_t, float>(__VA_ARGS__);                                  \
        case 8: return FN<bf16_t, bf16_t>(__VA_ARGS__);                                 \
        default: return OSS_ERR_SHAPE;                                                  \
    }

int ln_nchw_fwd(oss_dtype xt, oss_dtype yt, const void *x, const float *w, const float *bias, const void *gate, void *y,
                float *mean, float *rstd, int B, int C, int P, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, float eps,
                hipStream_t s) {
    OSS_LN_DISPATCH(ln_fwd_t, x, w, bias, gate, y, mean, rstd, B, C, P, xsb, xsc, gsb, gsc, eps, s)
}

int ln_nchw_bwd(oss_dtype xt, oss_dtype yt, const void *x, const float *w, const float *bias, const void *gate,
                const void *dy, const float *mean, const float *rstd, void *dx, void *dgate, float *dw, float *db,
                float *part, int B, int C, int P, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, hipStream_t s) {
    OSS_LN_DISPATCH(ln_bwd_t, x, w, bias, gate, dy, mean, rstd, dx, dgate, dw, db, part, B, C, P, xsb, xsc, gsb, gsc, s)
}

}  // namespace oss
