// oss_ffn.hip -- the gate of the EFFN (FeedForward of the OSS block, SRGAN/VmambaIR/archs/MambaSISR6_arch.py:213-217):
//   x1, x2 = dwconv(project_in(x)).chunk(2, dim=1);  out = gelu(x1) * x2          (exact, erf-based gelu)
// forward: one pass (read both halves, write out); backward: one pass that writes the gradient of BOTH halves into
// the (B, 2 Hd, H, W) buffer the depth-wise conv's backward reads -- instead of gelu, mul forward and gelu_backward,
// two muls and a cat backward.  fp32 math, one rounding.  HBM-bound: 16-byte accesses, 8 elements per lane.
#include "oss_device.h"
#include "oss_host.h"

namespace oss {

__device__ __forceinline__ float gelu_f(float x, float &cdf) {
    cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
    return x * cdf;
}

// h: (B, 2, n) with batch stride hsb (elements), halves contiguous; out: (B, n) contiguous
template <typename T>
__global__ void __launch_bounds__(256)
oss_gelu_gate_fwd_kernel(const T *__restrict__ h, T *__restrict__ out, size_t n, int64_t hsb) {
    const int b = blockIdx.y;
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const T *h1 = h + b * hsb + i, *h2 = h1 + n;
    T *o = out + (size_t)b * n + i;
    const int valid = (int)min((size_t)8, n - i);
    const bool vec = valid == 8 && ((reinterpret_cast<uintptr_t>(h1) | reinterpret_cast<uintptr_t>(h2) | reinterpret_cast<uintptr_t>(o)) & 15u) == 0;
    float a[8], g[8];
    load_items<8>(h1, valid, vec, a);
    load_items<8>(h2, valid, vec, g);
#pragma unroll
    for (int k = 0; k < 8; ++k) { float cdf; a[k] = gelu_f(a[k], cdf) * g[k]; }
    store_items<8>(o, valid, vec, a);
}

// dh[b, 0] = dout * x2 * gelu'(x1);  dh[b, 1] = dout * gelu(x1);  dh: (B, 2, n) contiguous
template <typename T>
__global__ void __launch_bounds__(256)
oss_gelu_gate_bwd_kernel(const T *__restrict__ h, const T *__restrict__ dout, T *__restrict__ dh, size_t n, int64_t hsb, int64_t gsb) {
    const int b = blockIdx.y;
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const T *h1 = h + b * hsb + i, *h2 = h1 + n, *gp = dout + b * gsb + i;
    T *d1 = dh + (size_t)b * 2 * n + i, *d2 = d1 + n;
    const int valid = (int)min((size_t)8, n - i);
    const bool vec = valid == 8 && ((reinterpret_cast<uintptr_t>(h1) | reinterpret_cast<uintptr_t>(h2) | reinterpret_cast<uintptr_t>(gp) |
                                     reinterpret_cast<uintptr_t>(d1) | reinterpret_cast<uintptr_t>(d2)) & 15u) == 0;
    float a[8], g[8], dy[8], o1[8], o2[8];
    load_items<8>(h1, valid, vec, a);
    load_items<8>(h2, valid, vec, g);
    load_items<8>(gp, valid, vec, dy);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float cdf;
        const float ge = gelu_f(a[k], cdf);
        const float pdf = 0.3989422804014327f * exp2_hw(-0.5f * a[k] * a[k] * kLog2e);
        o1[k] = dy[k] * g[k] * __builtin_fmaf(a[k], pdf, cdf);
        o2[k] = dy[k] * ge;
    }
    store_items<8>(d1, valid, vec, o1);
    store_items<8>(d2, valid, vec, o2);
}

int gelu_gate_fwd(oss_dtype io, const void *h, void *out, int B, size_t n, int64_t hsb, hipStream_t s) {
    if (B > 65535) return OSS_ERR_SHAPE;
    dim3 grid((unsigned)((n + 2047) / 2048), B);
    switch (io) {
        case OSS_F32: hipLaunchKernelGGL(oss_gelu_gate_fwd_kernel<float>, grid, dim3(256), 0, s, reinterpret_cast<const float *>(h), reinterpret_cast<float *>(out), n, hsb); break;
        case OSS_F16: hipLaunchKernelGGL(oss_gelu_gate_fwd_kernel<f16_t>, grid, dim3(256), 0, s, reinterpret_cast<const f16_t *>(h), reinterpret_cast<f16_t *>(out), n, hsb); break;
        case OSS_BF16: hipLaunchKernelGGL(oss_gelu_gate_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, reinterpret_cast<const bf16_t *>(h), reinterpret_cast<bf16_t *>(out), n, hsb); break;
        default: return OSS_ERR_SHAPE;
    }
    return (int)hipGetLastError();
}

int gelu_gate_bwd(oss_dtype io, const void *h, const void *dout, void *dh, int B, size_t n, int64_t hsb, int64_t gsb, hipStream_t s) {
    if (B > 65535) return OSS_ERR_SHAPE;
    dim3 grid((unsigned)((n + 2047) / 2048), B);
    switch (io) {
        case OSS_F32: hipLaunchKernelGGL(oss_gelu_gate_bwd_kernel<float>, grid, dim3(256), 0, s, reinterpret_cast<const float *>(h), reinterpret_cast<const float *>(dout), reinterpret_cast<float *>(dh), n, hsb, gsb); break;
        case OSS_F16: hipLaunchKernelGGL(oss_gelu_gate_bwd_kernel<f16_t>, grid, dim3(256), 0, s, reinterpret_cast<const f16_t *>(h), reinterpret_cast<const f16_t *>(dout), reinterpret_cast<f16_t *>(dh), n, hsb, gsb); break;
        case OSS_BF16: hipLaunchKernelGGL(oss_gelu_gate_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, reinterpret_cast<const bf16_t *>(h), reinterpret_cast<const bf16_t *>(dout), reinterpret_cast<bf16_t *>(dh), n, hsb, gsb); break;
        default: return OSS_ERR_SHAPE;
    }
    return (int)hipGetLastError();
}

}  // namespace oss
