// oss_merge.hip -- cross-merge of the four spatial scan directions (reference:
// SRGAN/VmambaIR/archs/MambaSISR6_arch.py:427-430) in one pass:
//   y[b,d,h,w] = ((o0[b,d,hW+w] + o2[b,d,hW+w]) + o1[b,d,wH+h]) + o3[b,d,wH+h]        (fp32)
// with o_k the omni scan's un-flipped outputs (B, 4, D, L): directions 0/2 in row-major, 1/3 in
// column-major order.  The association order is the reference's ((y0 + flip y2) + T y1) + T flip y3,
// so the result is bit-identical to the torch expression.  The two column-major planes go through a
// padded 32x32 LDS tile so that both the reads and the writes are coalesced.  HBM-bound: 4 reads +
// 1 write per element.
#include "oss_device.h"
#include "oss_host.h"

namespace oss {

template <typename T>
__global__ void __launch_bounds__(256)
oss_merge4_kernel(const T *__restrict__ out, float *__restrict__ y, int D, int H, int W) {
    __shared__ float t1[32][33], t3[32][33];
    const int plane = blockIdx.z;  // b * D + d
    const int b = plane / D, d = plane - b * D;
    const int L = H * W;
    const T *o0 = out + ((size_t)(b * 4 + 0) * D + d) * L;
    const T *o1 = out + ((size_t)(b * 4 + 1) * D + d) * L;
    const T *o2 = out + ((size_t)(b * 4 + 2) * D + d) * L;
    const T *o3 = out + ((size_t)(b * 4 + 3) * D + d) * L;
    const int h0 = blockIdx.y * 32, w0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    // column-major planes: element (h, w) lives at w*H + h; read with h fastest.  All sixteen loads of a thread are issued
    // first, with clamped addresses (a load under `ok ? :` / `if` is its own round trip: branch + s_waitcnt vmcnt(0))
    float a1[4], a3[4], a0[4], a2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int w = min(w0 + ty + 8 * r, W - 1), h = min(h0 + tx, H - 1);
        a1[r] = to_f32(o1[(size_t)w * H + h]);
        a3[r] = to_f32(o3[(size_t)w * H + h]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int h = min(h0 + ty + 8 * r, H - 1), w = min(w0 + tx, W - 1);
        const size_t i = (size_t)h * W + w;
        a0[r] = to_f32(o0[i]);
        a2[r] = to_f32(o2[i]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        t1[ty + 8 * r][tx] = a1[r];
        t3[ty + 8 * r][tx] = a3[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int hl = ty + 8 * r;  // local h
        const int h = h0 + hl, w = w0 + tx;
        if (h < H && w < W) {
            const size_t i = (size_t)h * W + w;
            const float a = a0[r] + a2[r];
            const float c = a + t1[tx][hl];
            y[(size_t)plane * L + i] = c + t3[tx][hl];
        }
    }
}

int merge4(oss_dtype io, const void *out, float *y, int B, int D, int H, int W, hipStream_t s) {
    dim3 grid((W + 31) / 32, (H + 31) / 32, B * D);
    switch (io) {
        case OSS_F32: hipLaunchKernelGGL(oss_merge4_kernel<float>, grid, dim3(256), 0, s, reinterpret_cast<const float *>(out), y, D, H, W); break;
        case OSS_F16: hipLaunchKernelGGL(oss_merge4_kernel<f16_t>, grid, dim3(256), 0, s, reinterpret_cast<const f16_t *>(out), y, D, H, W); break;
        case OSS_BF16: hipLaunchKernelGGL(oss_merge4_kernel<bf16_t>, grid, dim3(256), 0, s, reinterpret_cast<const bf16_t *>(out), y, D, H, W); break;
        default: return OSS_ERR_SHAPE;
    }
    return (int)hipGetLastError();
}

}  // namespace oss
