// oss_layernorm.hip -- per-pixel LayerNorm over the channel axis of an NCHW tensor, NCHW in / NCHW out
// (norm1, norm2, out_norm, channel_norm of the OSS block).  The reference does it as
// rearrange 'b c h w -> b (h w) c' -> mean / var(unbiased=False) -> (x-mu)/sqrt(var+1e-5)*w+b ->
// rearrange back (SRGAN/VmambaIR/archs/MambaSISR6_arch.py:144-195): two permute copies and ~6
// elementwise/reduction kernels per call, three calls per block.  Here: one kernel forward, one kernel
// (+ a tiny finishing kernel for dweight/dbias) backward, no permutes.  HBM-bound.
//
// One thread = one pixel; the channel loop strides by the plane size, so a wave reads 64 consecutive
// pixels of one channel per instruction (coalesced).  Two-pass variance (no E[x^2]-mu^2 cancellation).
// Optional fused epilogue: y *= silu(gate) (SS2D_1: y1 * act(z), MambaSISR6_arch.py:488-493).
#include "oss_device.h"
#include "oss_host.h"

namespace oss {

__device__ __forceinline__ float silu_f(float z) { return z * __builtin_amdgcn_rcpf(1.f + exp2_hw(-z * kLog2e)); }

// x: (B, C, P) with element strides (xsb, xsc), pixels contiguous.  y/gate: contiguous (B, C, P).
template <typename TX, typename TY, bool WITH_BIAS, bool GATE>
__global__ void __launch_bounds__(256)
oss_ln_nchw_fwd_kernel(const TX *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                       const TY *__restrict__ gate, TY *__restrict__ y, float *__restrict__ mean_out,
                       float *__restrict__ rstd_out, int C, int P, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, float eps) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const TX *xp = x + b * xsb + p;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += to_f32(xp[c * xsc]);
    const float mu = s / (float)C;
    float v = 0.f;
    for (int c = 0; c < C; ++c) { const float d = to_f32(xp[c * xsc]) - mu; v = __builtin_fmaf(d, d, v); }
    const float rstd = 1.0f / sqrtf(v / (float)C + eps);
    mean_out[(size_t)b * P + p] = mu;
    rstd_out[(size_t)b * P + p] = rstd;
    TY *yp = y + (size_t)b * C * P + p;
    const TY *gp = GATE ? gate + b * gsb + p : nullptr;
    for (int c = 0; c < C; ++c) {
        const float xv = to_f32(xp[c * xsc]);
        // BiasFree (MambaSISR6_arch.py:160-164) divides x (not x - mu) by sigma
        float o = WITH_BIAS ? (xv - mu) * rstd * w[c] + bias[c] : xv * rstd * w[c];
        if constexpr (GATE) o *= silu_f(to_f32(gp[c * gsc]));
        yp[(size_t)c * P] = from_f32<TY>(o);
    }
}

// dx, and per-workgroup partials of dweight / dbias (and dgate when GATE).
//   WITH_BIAS: xhat = (x-mu) rstd;  g = dy w;  dx = rstd (g - mean(g) - xhat mean(g xhat))
//   BiasFree : y = x rstd w, rstd = (var+eps)^-1/2 with var around mu:
//              dx = rstd g - (x - mu) rstd^3 mean(g x)
template <typename TX, typename TY, bool WITH_BIAS, bool GATE>
__global__ void __launch_bounds__(256)
oss_ln_nchw_bwd_kernel(const TX *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                       const TY *__restrict__ gate, const TY *__restrict__ dy, const float *__restrict__ mean_in,
                       const float *__restrict__ rstd_in, TX *__restrict__ dx, TY *__restrict__ dgate,
                       float *__restrict__ part /*[nblk][2][C]*/, int C, int P, int64_t xsb, int64_t xsc, int64_t gsb,
                       int64_t gsc) {
    extern __shared__ float red[];  // [4 waves][2][C]
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool ok = p < P;
    const int pc = ok ? p : P - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const TX *xp = x + b * xsb + pc;
    const TY *gyp = dy + (size_t)b * C * P + pc;
    const TY *gp = GATE ? gate + b * gsb + pc : nullptr;
    const float mu = mean_in[(size_t)b * P + pc], rstd = rstd_in[(size_t)b * P + pc];
    const float okf = ok ? 1.f : 0.f;
    float s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < C; ++c) {
        const float xv = to_f32(xp[c * xsc]);
        const float xh = WITH_BIAS ? (xv - mu) * rstd : xv * rstd;
        float g = to_f32(gyp[(size_t)c * P]) * okf;
        if constexpr (GATE) {
            const float z = to_f32(gp[c * gsc]);
            g *= silu_f(z);
        }
        // parameter gradients: sum over the wave's 64 pixels, then over the workgroup's 4 waves
        const float pw = segment_sum_to_last<64>(g * xh);
        const float pb = segment_sum_to_last<64>(g);
        if (lane == 63) { red[(wave * 2) * C + c] = pw; red[(wave * 2 + 1) * C + c] = pb; }
        const float gw = g * w[c];
        s1 += gw;
        s2 = __builtin_fmaf(gw, WITH_BIAS ? xh : xv, s2);
    }
    const float m1 = s1 / (float)C, m2 = s2 / (float)C;
    TX *dxp = dx + (size_t)b * C * P + pc;
    TY *dgp = GATE ? dgate + (size_t)b * C * P + pc : nullptr;
    for (int c = 0; c < C; ++c) {
        const float xv = to_f32(xp[c * xsc]);
        const float xh = (xv - mu) * rstd;
        const float gy = to_f32(gyp[(size_t)c * P]);
        float g = gy;
        if constexpr (GATE) {
            const float z = to_f32(gp[c * gsc]);
            const float sg = __builtin_amdgcn_rcpf(1.f + exp2_hw(-z * kLog2e));  // sigmoid(z)
            const float sl = z * sg;
            const float o = WITH_BIAS ? xh * w[c] + bias[c] : xv * rstd * w[c];  // LN output before the gate
            if (ok) dgp[(size_t)c * P] = from_f32<TY>(gy * o * (sg + sl * (1.f - sg)));
            g *= sl;
        }
        const float gw = g * w[c];
        const float d = WITH_BIAS ? rstd * (gw - m1 - xh * m2) : rstd * gw - xh * rstd * rstd * m2;
        if (ok) dxp[(size_t)c * P] = from_f32<TX>(d);
    }
    __syncthreads();
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i - which * C;
        part[(size_t)blk * 2 * C + i] = ((red[(0 * 2 + which) * C + c] + red[(1 * 2 + which) * C + c]) +
                                         red[(2 * 2 + which) * C + c]) + red[(3 * 2 + which) * C + c];
    }
}

__global__ void __launch_bounds__(256)
oss_ln_nchw_bwd_finish(const float *__restrict__ part, float *__restrict__ dw, float *__restrict__ db, int nblk, int C) {
    // 64 outputs x 4 slices of the partial list per workgroup; slices combined in a fixed order
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    float s = 0.f;
    if (i < 2 * C)
        for (int k = slice; k < nblk; k += 4) s += part[(size_t)k * 2 * C + i];
    red[slice][col] = s;
    __syncthreads();
    if (slice == 0 && i < 2 * C) {
        const float t = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        if (i < C) dw[i] = t;
        else if (db) db[i - C] = t;
    }
}

template <typename TX, typename TY>
static int ln_fwd_t(const void *x, const float *w, const float *bias, const void *gate, void *y, float *mean, float *rstd,
                    int B, int C, int P, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, float eps, hipStream_t s) {
    dim3 grid((P + 255) / 256, B);
    const TX *xp = reinterpret_cast<const TX *>(x);
    const TY *gp = reinterpret_cast<const TY *>(gate);
    TY *yp = reinterpret_cast<TY *>(y);
    if (bias) {
        if (gate) hipLaunchKernelGGL((oss_ln_nchw_fwd_kernel<TX, TY, true, true>), grid, dim3(256), 0, s, xp, w, bias, gp, yp, mean, rstd, C, P, xsb, xsc, gsb, gsc, eps);
        else      hipLaunchKernelGGL((oss_ln_nchw_fwd_kernel<TX, TY, true, false>), grid, dim3(256), 0, s, xp, w, bias, gp, yp, mean, rstd, C, P, xsb, xsc, gsb, gsc, eps);
    } else {
        if (gate) hipLaunchKernelGGL((oss_ln_nchw_fwd_kernel<TX, TY, false, true>), grid, dim3(256), 0, s, xp, w, bias, gp, yp, mean, rstd, C, P, xsb, xsc, gsb, gsc, eps);
        else      hipLaunchKernelGGL((oss_ln_nchw_fwd_kernel<TX, TY, false, false>), grid, dim3(256), 0, s, xp, w, bias, gp, yp, mean, rstd, C, P, xsb, xsc, gsb, gsc, eps);
    }
    return (int)hipGetLastError();
}

template <typename TX, typename TY>
static int ln_bwd_t(const void *x, const float *w, const float *bias, const void *gate, const void *dy, const float *mean,
                    const float *rstd, void *dx, void *dgate, float *dw, float *db, float *part, int B, int C, int P,
                    int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, hipStream_t s) {
    dim3 grid((P + 255) / 256, B);
    const int nblk = grid.x * grid.y;
    const size_t smem = sizeof(float) * 8 * (size_t)C;
    const TX *xp = reinterpret_cast<const TX *>(x);
    const TY *gp = reinterpret_cast<const TY *>(gate);
    const TY *dyp = reinterpret_cast<const TY *>(dy);
    TX *dxp = reinterpret_cast<TX *>(dx);
    TY *dgp = reinterpret_cast<TY *>(dgate);
    if (bias) {
        if (gate) hipLaunchKernelGGL((oss_ln_nchw_bwd_kernel<TX, TY, true, true>), grid, dim3(256), smem, s, xp, w, bias, gp, dyp, mean, rstd, dxp, dgp, part, C, P, xsb, xsc, gsb, gsc);
        else      hipLaunchKernelGGL((oss_ln_nchw_bwd_kernel<TX, TY, true, false>), grid, dim3(256), smem, s, xp, w, bias, gp, dyp, mean, rstd, dxp, dgp, part, C, P, xsb, xsc, gsb, gsc);
    } else {
        if (gate) hipLaunchKernelGGL((oss_ln_nchw_bwd_kernel<TX, TY, false, true>), grid, dim3(256), smem, s, xp, w, bias, gp, dyp, mean, rstd, dxp, dgp, part, C, P, xsb, xsc, gsb, gsc);
        else      hipLaunchKernelGGL((oss_ln_nchw_bwd_kernel<TX, TY, false, false>), grid, dim3(256), smem, s, xp, w, bias, gp, dyp, mean, rstd, dxp, dgp, part, C, P, xsb, xsc, gsb, gsc);
    }
    hipLaunchKernelGGL(oss_ln_nchw_bwd_finish, dim3((2 * C + 63) / 64), dim3(256), 0, s, part, dw, db, nblk, C);
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
