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

// 16-bit input, even H and W >= 64: 64 x 64 tiles, two adjacent elements per lane on the contiguous axis of every access
// (4-byte loads, 8-byte stores); same order of additions as the kernel above, hence the same bits.
template <typename T>
__global__ void __launch_bounds__(256)
oss_merge4_pair_kernel(const T *__restrict__ out, float *__restrict__ y, int D, int H, int W) {
    __shared__ float t1[64][65], t3[64][65];
    const int plane = blockIdx.z;  // b * D + d
    const int b = plane / D, d = plane - b * D;
    const int L = H * W;
    const T *o0 = out + ((size_t)(b * 4 + 0) * D + d) * L;
    const T *o1 = out + ((size_t)(b * 4 + 1) * D + d) * L;
    const T *o2 = out + ((size_t)(b * 4 + 2) * D + d) * L;
    const T *o3 = out + ((size_t)(b * 4 + 3) * D + d) * L;
    const int h0 = blockIdx.y * 64, w0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 pairs x 8
    float a1[8][2], a3[8][2], a0[8][2], a2[8][2];
    const int h2 = h0 + 2 * tx, h2c = min(h2, H - 2);
    const int w = w0 + 2 * tx, wc = min(w, W - 2);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const size_t i = (size_t)min(w0 + ty + 8 * r, W - 1) * H + h2c;   // column-major planes: h fastest
        load_v<T, 2>(o1 + i, a1[r]);
        load_v<T, 2>(o3 + i, a3[r]);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const size_t i = (size_t)min(h0 + ty + 8 * r, H - 1) * W + wc;
        load_v<T, 2>(o0 + i, a0[r]);
        load_v<T, 2>(o2 + i, a2[r]);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        t1[2 * tx][ty + 8 * r] = a1[r][0]; t1[2 * tx + 1][ty + 8 * r] = a1[r][1];
        t3[2 * tx][ty + 8 * r] = a3[r][0]; t3[2 * tx + 1][ty + 8 * r] = a3[r][1];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int hl = ty + 8 * r, h = h0 + hl;
        if (h < H && w < W) {
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float a = a0[r][e] + a2[r][e];
                const float c = a + t1[hl][2 * tx + e];
                v[e] = c + t3[hl][2 * tx + e];
            }
            store_v<float, 2>(y + (size_t)plane * L + (size_t)h * W + w, v);
        }
    }
}

int merge4(oss_dtype io, const void *out, float *y, int B, int D, int H, int W, hipStream_t s) {
    if (io != OSS_F32 && H >= 64 && W >= 64 && H % 2 == 0 && W % 2 == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 3u) == 0 && (reinterpret_cast<uintptr_t>(y) & 7u) == 0) {
        dim3 grid64((W + 63) / 64, (H + 63) / 64, B * D);
        if (io == OSS_F16) hipLaunchKernelGGL(oss_merge4_pair_kernel<f16_t>, grid64, dim3(256), 0, s, reinterpret_cast<const f16_t *>(out), y, D, H, W);
        else               hipLaunchKernelGGL(oss_merge4_pair_kernel<bf16_t>, grid64, dim3(256), 0, s, reinterpret_cast<const bf16_t *>(out), y, D, H, W);
        return (int)hipGetLastError();
    }
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
