// oss_conv3x3_thin.hip -- dense 3x3 convolutions (stride 1, zero padding 1) with a THIN side: at most 4 channels in or out.
//
// The UNets around the OSS blocks open with conv(3 -> 48) (OverlapPatchEmbed, SRGAN/VmambaIR/archs/MambaSISR6_arch.py:520-528)
// and close with conv(96 -> 3) at the full output resolution (the x4 tail's last layer, archs/common.py:45-60 +
// MambaSISR6_arch.py:598-602; Mamber32.output, mamber32_arch.py:608).  These are not GEMM-shaped: with 3 channels on one side
// an implicit-GEMM tile is 90 % padding, and the vendor kernels that ran them were the most expensive convolutions of the step
// -- conv_last at 256 x 256, batch 8: forward 375 us, weight gradient 354 us, input gradient 74 us for 100 MB of activations
// (profiles/r03_rocprof_bench_steady_state.txt: igemm_fwd ..bt256x32x8, igemm_wrw ..bt64x32x32, igemm_bwd ..bt256x32x8), plus
// NCHW <-> NHWC transposes around each.  Here they are what they are, HBM-bound stencils on NCHW, written like the depth-wise
// kernels (oss_dwconv.hip / oss_stencil.h): a lane owns 8 consecutive pixels of an image row, one 16-byte load per row, the
// halo pixels come from the neighbouring lanes by DPP, fp32 master weights arrive as scalar loads, fp32 accumulation.
//
//   reduce  (C -> F, F <= 4):  y[f] = bias[f] + sum_c sum_t W[f][c][t] x[c][p + t]      conv_last forward
//            the four waves of a workgroup split the C channels of the same 512 pixels and are summed through LDS (fixed order)
//   expand  (F -> C):          y[c] = bias[c] + sum_f sum_t W[c][f][t] x[f][p + t]      patch_embed forward; conv_last input
//            gradient (x = dy, taps mirrored, W read transposed); the F input planes' rows stay in registers for all C outputs
//   wgrad:   S[c][f][t] = sum_p few[f][p] many[c][p + t]   one workgroup per (many-plane c, batch) -> partials, summed over batch
//            by the deferred finishing launch; conv_last: few = dy, many = x; patch_embed: few = x, many = dy, taps mirrored
// 16-bit I/O (bf16 / fp16), W % 8 == 0, 16-byte aligned planes; anything else stays with the caller's fallback.
#include <initializer_list>
#include "oss_device.h"
#include "oss_host.h"
#include "oss_stencil.h"

namespace oss {

struct ThinW {            // where weight element (thin index f, wide index c, tap t) lives: w[f * sf + c * sc + (flip ? 8 - t : t)]
    int64_t sf, sc;
    int flip;
};

// ---- reduce: C -> F ---------------------------------------------------------------------------------------------------------
template <typename T, int F, bool EDGE>
__global__ void __launch_bounds__(256)
oss_conv3x3_reduce_kernel(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias, T *__restrict__ y,
                          int C, int H, int W, int64_t xsb, int64_t xsc, int64_t ysb, int64_t ysc, ThinW ws) {
    __shared__ float red[3][F * 8][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: the weights stay scalar loads
    const int b = blockIdx.y;
    const int lpr = W >> 3, ngroups = lpr * H;
    const int g = blockIdx.x * 64 + lane;
    const bool live = g < ngroups;
    const int gc = live ? g : ngroups - 1;      // dead lanes shadow the last group: every lane takes part in the halo exchange
    const int h = gc / lpr, cg = gc - h * lpr, w0 = cg << 3;
    const bool first = cg == 0, last = cg == lpr - 1;
    float acc[F][8];
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[f][j] = 0.f;
    for (int c = wave; c < C; c += 4) {
        const T *xp = x + b * xsb + c * xsc;
        float v[3][10];
#pragma unroll
        for (int r = 0; r < 3; ++r) row10<T, EDGE>(xp, h, r - 1, H, W, w0, first, last, v[r]);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float *k = w + f * ws.sf + c * ws.sc;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float k0 = k[ws.flip ? 8 - r * 3 : r * 3], k1 = k[ws.flip ? 7 - r * 3 : r * 3 + 1], k2 = k[ws.flip ? 6 - r * 3 : r * 3 + 2];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[f][j] = __builtin_fmaf(k0, v[r][j], __builtin_fmaf(k1, v[r][j + 1], __builtin_fmaf(k2, v[r][j + 2], acc[f][j])));
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[wave - 1][f * 8 + j][lane] = acc[f][j];
    }
    __syncthreads();
    if (wave == 0 && live) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            float o[8];
            const float bv = bias ? bias[f] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = ((acc[f][j] + red[0][f * 8 + j][lane]) + (red[1][f * 8 + j][lane] + red[2][f * 8 + j][lane])) + bv;
            store8<T>(y + b * ysb + f * ysc + (int64_t)h * W + w0, o);
        }
    }
}

// ---- expand: F -> C ---------------------------------------------------------------------------------------------------------
// grid (groups of 256 lanes, batch, slices of the C outputs)
template <typename T, int F, bool EDGE>
__global__ void __launch_bounds__(256)
oss_conv3x3_expand_kernel(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias, T *__restrict__ y,
                          int C, int H, int W, int64_t xsb, int64_t xsc, int64_t ysb, int64_t ysc, ThinW ws) {
    const int b = blockIdx.y;
    const int per = (C + gridDim.z - 1) / gridDim.z, c_begin = blockIdx.z * per, c_end = min(C, c_begin + per);
    const int lpr = W >> 3, ngroups = lpr * H;
    const int g = blockIdx.x * 256 + threadIdx.x;
    const bool live = g < ngroups;
    const int gc = live ? g : ngroups - 1;
    const int h = gc / lpr, cg = gc - h * lpr, w0 = cg << 3;
    const bool first = cg == 0, last = cg == lpr - 1;
    float v[F][3][10];
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int r = 0; r < 3; ++r) row10<T, EDGE>(x + b * xsb + f * xsc, h, r - 1, H, W, w0, first, last, v[f][r]);
    if (!live) return;   // the halo exchange is over: dead lanes have nothing to store
    for (int c = c_begin; c < c_end; ++c) {
        float o[8];
        const float bv = bias ? bias[c] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = bv;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float *k = w + f * ws.sf + c * ws.sc;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float k0 = k[ws.flip ? 8 - r * 3 : r * 3], k1 = k[ws.flip ? 7 - r * 3 : r * 3 + 1], k2 = k[ws.flip ? 6 - r * 3 : r * 3 + 2];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    o[j] = __builtin_fmaf(k0, v[f][r][j], __builtin_fmaf(k1, v[f][r][j + 1], __builtin_fmaf(k2, v[f][r][j + 2], o[j])));
            }
        }
        store8<T>(y + b * ysb + c * ysc + (int64_t)h * W + w0, o);
    }
}

// ---- weight gradient ----------------------------------------------------------------------------------------------------------
// S[c][f][t] = sum over the plane of few[f][p] * many[c][p + off(t)]; one workgroup per (c, batch); partial layout chosen by the
// host: part[b * pstride + f * psf + c * psc + (pflip ? 8 - t : t)], bias partial (BIAS, workgroups with c == 0 only)
// part[b * pstride + pbias + f] = sum of few[f].
struct ThinP {
    int64_t pstride, psf, psc, pbias;
    int pflip;
};
template <typename T, int F, bool EDGE, bool BIAS>
__global__ void __launch_bounds__(256)
oss_conv3x3_thin_wgrad_kernel(const T *__restrict__ few, const T *__restrict__ many, float *__restrict__ part, int C, int H, int W,
                              int64_t fsb, int64_t fsc, int64_t msb, int64_t msc, ThinP pp) {
    constexpr int NA = F * 9 + (BIAS ? F : 0);
    const int c = blockIdx.x, b = blockIdx.y;
    const T *mp = many + b * msb + c * msc;
    const T *fp = few + b * fsb;
    float acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0.f;
    const int lpr = W >> 3, ngroups = lpr * H;
    for (int g0 = 0; g0 < ngroups; g0 += 256) {   // uniform trip count: every lane takes part in the halo exchange
        const int g = g0 + threadIdx.x;
        const bool live = g < ngroups;
        const int gc = live ? g : ngroups - 1;
        const int h = gc / lpr, cg = gc - h * lpr, w0 = cg << 3;
        const bool first = cg == 0, last = cg == lpr - 1;
        float fv[F][8];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            u32x4 q = *reinterpret_cast<const u32x4 *>(fp + f * fsc + (int64_t)h * W + w0);
            if (!live) q = u32x4{0u, 0u, 0u, 0u};   // a lane that shadows the last group adds nothing
            unpack2<T>(q.x, fv[f][0], fv[f][1]); unpack2<T>(q.y, fv[f][2], fv[f][3]);
            unpack2<T>(q.z, fv[f][4], fv[f][5]); unpack2<T>(q.w, fv[f][6], fv[f][7]);
        }
        if constexpr (BIAS) {
            if (c == 0) {
#pragma unroll
                for (int f = 0; f < F; ++f)
                    acc[F * 9 + f] += ((fv[f][0] + fv[f][1]) + (fv[f][2] + fv[f][3])) + ((fv[f][4] + fv[f][5]) + (fv[f][6] + fv[f][7]));
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float v[10];
            row10<T, EDGE>(mp, h, r - 1, H, W, w0, first, last, v);
#pragma unroll
            for (int f = 0; f < F; ++f)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    float a = acc[f * 9 + r * 3 + cc];
#pragma unroll
                    for (int j = 0; j < 8; ++j) a = __builtin_fmaf(fv[f][j], v[j + cc], a);
                    acc[f * 9 + r * 3 + cc] = a;
                }
        }
    }
    __shared__ float red[4][NA];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const float s = segment_sum_to_last<64>(acc[i]);
        if (lane == 63) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NA) {
        const int i = threadIdx.x;
        const float s = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
        if (i < F * 9) {
            const int f = i / 9, t = i - f * 9;
            part[b * pp.pstride + f * pp.psf + c * pp.psc + (pp.pflip ? 8 - t : t)] = s;
        } else if (c == 0) {
            part[b * pp.pstride + pp.pbias + (i - F * 9)] = s;
        }
    }
}

// partial vectors summed over batch in batch order (the eager path; inside a training step the sum joins the deferred finishing
// launch, oss_flush_finishes)
__global__ void __launch_bounds__(256)
oss_conv3x3_thin_finish(const float *__restrict__ part, float *__restrict__ dw, float *__restrict__ db, int K, size_t pvec, size_t nw,
                        size_t nout) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nout) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += part[(size_t)k * pvec + i];
    if (i < nw) dw[i] = s; else db[i - nw] = s;
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
static bool thin_aligned(std::initializer_list<const void *> ptrs, std::initializer_list<int64_t> strides) {
    for (const void *p : ptrs)
        if (reinterpret_cast<uintptr_t>(p) & 15u) return false;
    for (int64_t st : strides)
        if (st % 8 != 0) return false;
    return true;
}

int conv3x3_thin_ok(oss_dtype io, int Cin, int Cout, int H, int W) {
    if (io != OSS_F16 && io != OSS_BF16) return 0;
    if (Cin < 1 || Cout < 1 || H < 1 || W < 8 || W % 8 != 0 || W > 4096) return 0;
    return (Cin <= 4 || Cout <= 4) ? 1 : 0;
}

#define OSS_THIN_F(F_, CALL)                     \
    switch (F_) {                                \
        case 1: { constexpr int FF = 1; CALL; } break; \
        case 2: { constexpr int FF = 2; CALL; } break; \
        case 3: { constexpr int FF = 3; CALL; } break; \
        default: { constexpr int FF = 4; CALL; } break; \
    }

template <typename T>
static int thin_reduce(const void *x, const float *w, const float *bias, void *y, int B, int C, int F, int H, int W, int64_t xsb,
                       int64_t xsc, int64_t ysb, int64_t ysc, ThinW ws, hipStream_t s) {
    const dim3 grid(((W / 8) * H + 63) / 64, B);
    const bool edge = stencil_edge(W);
    OSS_THIN_F(F, {
        if (edge)
            hipLaunchKernelGGL((oss_conv3x3_reduce_kernel<T, FF, true>), grid, dim3(256), 0, s, reinterpret_cast<const T *>(x), w, bias,
                               reinterpret_cast<T *>(y), C, H, W, xsb, xsc, ysb, ysc, ws);
        else
            hipLaunchKernelGGL((oss_conv3x3_reduce_kernel<T, FF, false>), grid, dim3(256), 0, s, reinterpret_cast<const T *>(x), w, bias,
                               reinterpret_cast<T *>(y), C, H, W, xsb, xsc, ysb, ysc, ws);
    });
    return (int)hipGetLastError();
}

template <typename T>
static int thin_expand(const void *x, const float *w, const float *bias, void *y, int B, int C, int F, int H, int W, int64_t xsb,
                       int64_t xsc, int64_t ysb, int64_t ysc, ThinW ws, hipStream_t s) {
    const int gx = ((W / 8) * H + 255) / 256;
    // slices of the C outputs so that the launch has a few thousand waves (each slice re-reads the F input planes' rows)
    int gz = 1;
    while (gz < 8 && (long)gx * B * gz * 4 < 4096 && C / (gz * 2) >= 8) gz *= 2;
    const dim3 grid(gx, B, gz);
    const bool edge = stencil_edge(W);
    OSS_THIN_F(F, {
        if (edge)
            hipLaunchKernelGGL((oss_conv3x3_expand_kernel<T, FF, true>), grid, dim3(256), 0, s, reinterpret_cast<const T *>(x), w, bias,
                               reinterpret_cast<T *>(y), C, H, W, xsb, xsc, ysb, ysc, ws);
        else
            hipLaunchKernelGGL((oss_conv3x3_expand_kernel<T, FF, false>), grid, dim3(256), 0, s, reinterpret_cast<const T *>(x), w, bias,
                               reinterpret_cast<T *>(y), C, H, W, xsb, xsc, ysb, ysc, ws);
    });
    return (int)hipGetLastError();
}

int conv3x3_thin_fwd(oss_dtype io, const void *x, const float *w, const float *bias, void *y, int B, int Cin, int Cout, int H, int W,
                     int64_t xsb, int64_t xsc, int64_t ysb, int64_t ysc, hipStream_t s) {
    if (!conv3x3_thin_ok(io, Cin, Cout, H, W) || !thin_aligned({x, y}, {xsb, xsc, ysb, ysc})) return OSS_ERR_SHAPE;
    if (B <= 0 || B > 65535) return OSS_ERR_SHAPE;
    // weight (Cout, Cin, 3, 3): element (co, ci, t) at co * Cin * 9 + ci * 9 + t
    if (Cout <= 4) {   // reduce: thin index f = co, wide index c = ci
        const ThinW ws{(int64_t)Cin * 9, 9, 0};
        return io == OSS_F16 ? thin_reduce<f16_t>(x, w, bias, y, B, Cin, Cout, H, W, xsb, xsc, ysb, ysc, ws, s)
                             : thin_reduce<bf16_t>(x, w, bias, y, B, Cin, Cout, H, W, xsb, xsc, ysb, ysc, ws, s);
    }
    const ThinW ws{9, (int64_t)Cin * 9, 0};   // expand: f = ci, c = co
    return io == OSS_F16 ? thin_expand<f16_t>(x, w, bias, y, B, Cout, Cin, H, W, xsb, xsc, ysb, ysc, ws, s)
                         : thin_expand<bf16_t>(x, w, bias, y, B, Cout, Cin, H, W, xsb, xsc, ysb, ysc, ws, s);
}

// dx[ci] = sum_co sum_t W[co][ci][8 - t] dy[co][p + t]   (the transposed convolution: taps mirrored, weight read transposed)
int conv3x3_thin_dgrad(oss_dtype io, const void *dy, const float *w, void *dx, int B, int Cin, int Cout, int H, int W, int64_t gsb,
                       int64_t gsc, int64_t dsb, int64_t dsc, hipStream_t s) {
    if (!conv3x3_thin_ok(io, Cin, Cout, H, W) || !thin_aligned({dy, dx}, {gsb, gsc, dsb, dsc})) return OSS_ERR_SHAPE;
    if (B <= 0 || B > 65535) return OSS_ERR_SHAPE;
    if (Cout <= 4) {   // expand dy (F = Cout planes) to the Cin planes of dx: f = co, c = ci
        const ThinW ws{(int64_t)Cin * 9, 9, 1};
        return io == OSS_F16 ? thin_expand<f16_t>(dy, w, nullptr, dx, B, Cin, Cout, H, W, gsb, gsc, dsb, dsc, ws, s)
                             : thin_expand<bf16_t>(dy, w, nullptr, dx, B, Cin, Cout, H, W, gsb, gsc, dsb, dsc, ws, s);
    }
    const ThinW ws{9, (int64_t)Cin * 9, 1};   // reduce the Cout planes of dy to the F = Cin planes of dx: f = ci, c = co
    return io == OSS_F16 ? thin_reduce<f16_t>(dy, w, nullptr, dx, B, Cout, Cin, H, W, gsb, gsc, dsb, dsc, ws, s)
                         : thin_reduce<bf16_t>(dy, w, nullptr, dx, B, Cout, Cin, H, W, gsb, gsc, dsb, dsc, ws, s);
}

size_t conv3x3_thin_wgrad_partial_floats(int B, int Cin, int Cout) { return (size_t)B * ((size_t)Cin * Cout * 9 + Cout); }

template <typename T>
static int thin_wgrad(const void *few, const void *many, float *part, int B, int C, int F, int H, int W, int64_t fsb, int64_t fsc,
                      int64_t msb, int64_t msc, ThinP pp, bool with_bias, hipStream_t s) {
    const dim3 grid(C, B);
    const bool edge = stencil_edge(W);
#define OSS_THIN_WG(E_, B_)                                                                                                    \
    hipLaunchKernelGGL((oss_conv3x3_thin_wgrad_kernel<T, FF, E_, B_>), grid, dim3(256), 0, s, reinterpret_cast<const T *>(few), \
                       reinterpret_cast<const T *>(many), part, C, H, W, fsb, fsc, msb, msc, pp)
    OSS_THIN_F(F, {
        if (edge) { if (with_bias) OSS_THIN_WG(true, true); else OSS_THIN_WG(true, false); }
        else { if (with_bias) OSS_THIN_WG(false, true); else OSS_THIN_WG(false, false); }
    });
#undef OSS_THIN_WG
    return (int)hipGetLastError();
}

// dw[co][ci][t] = sum_{b,p} dy[co][p] x[ci][p + t],  db[co] = sum dy[co]
int conv3x3_thin_wgrad(oss_dtype io, const void *x, const void *dy, float *dw, float *db, float *part, int B, int Cin, int Cout, int H,
                       int W, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, hipStream_t s) {
    if (!conv3x3_thin_ok(io, Cin, Cout, H, W) || !thin_aligned({x, dy}, {xsb, xsc, gsb, gsc})) return OSS_ERR_SHAPE;
    if (B <= 0 || B > 65535 || !dw || !part) return OSS_ERR_NULL;
    const size_t nw = (size_t)Cin * Cout * 9, pvec = nw + Cout;
    int rc;
    if (Cout <= 4) {
        // few = dy (f = co), many = x (c = ci): S[c][f][t] is dw[f][c][t]
        const ThinP pp{(int64_t)pvec, (int64_t)Cin * 9, 9, (int64_t)nw, 0};
        rc = io == OSS_F16 ? thin_wgrad<f16_t>(dy, x, part, B, Cin, Cout, H, W, gsb, gsc, xsb, xsc, pp, db != nullptr, s)
                           : thin_wgrad<bf16_t>(dy, x, part, B, Cin, Cout, H, W, gsb, gsc, xsb, xsc, pp, db != nullptr, s);
    } else {
        // few = x (f = ci), many = dy (c = co): sum_p x[f][p] dy[c][p + t] = sum_q dy[c][q] x[f][q - t] = dw[c][f][8 - t];
        // the bias gradient sums the MANY side here, which this kernel does not do: the caller's reduction (db == NULL only)
        if (db) return OSS_ERR_SHAPE;
        const ThinP pp{(int64_t)pvec, 9, (int64_t)Cin * 9, (int64_t)nw, 1};
        rc = io == OSS_F16 ? thin_wgrad<f16_t>(x, dy, part, B, Cout, Cin, H, W, xsb, xsc, gsb, gsc, pp, false, s)
                           : thin_wgrad<bf16_t>(x, dy, part, B, Cout, Cin, H, W, xsb, xsc, gsb, gsc, pp, false, s);
    }
    if (rc != 0) return rc;
    const size_t nout = nw + (db ? Cout : 0);
    if (defer_finish())
        defer_sum(part, B, pvec, nout, dw, nw, db);
    else
        hipLaunchKernelGGL(oss_conv3x3_thin_finish, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, s, part, dw, db, B, pvec, nw, nout);
    return (int)hipGetLastError();
}

}  // namespace oss
