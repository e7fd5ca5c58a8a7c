// oss_conv1x1_f32.hip -- the GEMM-shaped products of the OSS block at the REFERENCE's own precision (fp32 I/O, no autocast):
// 1x1 convolutions (in_conv / out_conv / project_in / project_out, SRGAN/VmambaIR/archs/MambaSISR6_arch.py:205,211,281,329) and
// the weight gradients of the x_proj / dt_proj products (:406-411) on v_mfma_f32_32x32x2_f32 -- true fp32 multiplies and adds on
// the matrix cores, no reduced-precision detour.
//
// Rounds 1-3 sent fp32 activations to the vendor convolution and to torch.einsum: 1x1 weight gradients through NCHW -> NHWC
// transposes and igemm_wrw (10.5 ms of the fp32 step), the projection weight gradients through four batched library GEMMs with
// permute copies around them (11 ms) -- profiles/r04_rocprof_bench_fp32_steady_state.txt.  On NCHW the fp32 MFMA needs no
// layout work at all: its operands are ONE value per lane (A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31]), so
//   * forward / input gradient  Y[m][p] = sum_k W(m, k) X[k][p]:  a lane loads 16 bytes = 4 consecutive pixels of channel
//     k0 + (l >> 5) and feeds them to FOUR MFMAs (one per pixel of the quad: the instruction does not care which pixel is
//     "column j"), 32 lanes x 16 B = 512 contiguous bytes per channel row; results leave as 16-byte stores;
//   * weight gradient  dW[m][n] = sum_p dY[m][p] X[n][p]:  the contraction runs over pixels, both operands are rows of an NCHW
//     tensor: a lane loads 4 consecutive pixels of its row and the four MFMAs each contract one of them.
// fp32 MFMA is 1/16 of the bf16 rate (256 flop / clock / CU), so these kernels are bound by the matrix pipe, not by their loads.
#include <initializer_list>
#include "oss_device.h"
#include "oss_host.h"
#include "oss_mfma.h"

namespace oss {

__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// ---- forward / input gradient -------------------------------------------------------------------------------------------------
// Y[b][m][p] = sum_k w[m * wsm + k * wsk] X[b][k][p] (+ bias[m]) (+ res[b][m][p]);  one wave = (32 MT) rows x 128 pixels.
// grid (ceil(P / 128), ceil(M / (32 MT)), B), 64 threads.  P % 4 == 0, 16-byte aligned rows (host-checked).
template <int MT>
__global__ void __launch_bounds__(64, 2)   // <= 256 registers (accumulators included): two waves per SIMD keep the matrix pipe fed across the loads
oss_conv1x1_f32_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ y,
                       int M, int K, int P, int64_t xsb, int64_t xsk, int64_t ysb, int64_t ysm, int64_t wsm, int64_t wsk,
                       const float *__restrict__ res) {
    const int lane = threadIdx.x, col = lane & 31, kg = lane >> 5;
    const int p0 = blockIdx.x * 128 + 4 * col;
    const int m0 = blockIdx.y * 32 * MT;
    const int b = blockIdx.z;
    const bool pok = p0 < P;
    const float *xb = x + b * xsb + (pok ? p0 : 0);
    f32x16 acc[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][q][r] = 0.f;
    bool mok[MT];
    const float *wr[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = m0 + 32 * t + col;
        mok[t] = m < M;
        wr[t] = w + (int64_t)(mok[t] ? m : 0) * wsm;
    }
    constexpr int U = 2;   // k-steps whose loads are issued together
    for (int k0 = 0; k0 < K; k0 += 2 * U) {
        f32x4 xv[U];
        float a[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 2 * u + kg;
            const bool kok = k < K;
            const int kc = kok ? k : 0;
            xv[u] = *reinterpret_cast<const f32x4 *>(xb + (int64_t)kc * xsk);
            if (!(kok && pok)) xv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const float v = wr[t][(int64_t)kc * wsk];
                a[u][t] = (kok && mok[t]) ? v : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[t][q] = mfma_f32(a[u][t], xv[u][q], acc[t][q]);
    }
    if (!pok) return;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (m < M) {
                const float bv = bias ? bias[m] : 0.f;
                f32x4 o = {acc[t][0][r] + bv, acc[t][1][r] + bv, acc[t][2][r] + bv, acc[t][3][r] + bv};
                if (res) {
                    const f32x4 rv = *reinterpret_cast<const f32x4 *>(res + b * ysb + (int64_t)m * ysm + p0);
                    o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
                }
                *reinterpret_cast<f32x4 *>(y + b * ysb + (int64_t)m * ysm + p0) = o;
            }
        }
}

int conv1x1_f32_ok(int M, int K, int P, int64_t xsb, int64_t xsk, const void *x, const void *y, const void *res) {
    if (M < 1 || K < 1 || P < 4 || P % 4 != 0) return 0;
    if (xsb % 4 != 0 || xsk % 4 != 0) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res)) & 15u) return 0;
    return 1;
}

int conv1x1_f32(const float *x, const float *w, const float *bias, float *y, int B, int M, int K, int P, int64_t xsb, int64_t xsk,
                int64_t wsm, int64_t wsk, hipStream_t s, const float *res) {
    if (!conv1x1_f32_ok(M, K, P, xsb, xsk, x, y, res)) return OSS_ERR_SHAPE;
    if (B <= 0 || B > 65535) return OSS_ERR_SHAPE;
    const int64_t ysb = (int64_t)M * P, ysm = P;
    // 64-row tiles when that still gives every SIMD a wave, else 32-row tiles (twice the waves, the activations read twice as often)
    const long waves64 = (long)((P + 127) / 128) * ((M + 63) / 64) * B;
    if (waves64 >= 1024 && M > 32) {
        const dim3 grid((P + 127) / 128, (M + 63) / 64, B);
        hipLaunchKernelGGL(oss_conv1x1_f32_kernel<2>, grid, dim3(64), 0, s, x, w, bias, y, M, K, P, xsb, xsk, ysb, ysm, wsm, wsk, res);
    } else {
        const dim3 grid((P + 127) / 128, (M + 31) / 32, B);
        hipLaunchKernelGGL(oss_conv1x1_f32_kernel<1>, grid, dim3(64), 0, s, x, w, bias, y, M, K, P, xsb, xsk, ysb, ysm, wsm, wsk, res);
    }
    return (int)hipGetLastError();
}

// ---- weight gradient (and every other "rows x rows over pixels" product) ------------------------------------------------------
// part[(b * slabs + slab)][g][m][n] = sum over the slab's pixels of A[b][g][m][p] Bm[b][g % GB][n][p]
//   A row (b, g, m):  a + b * asb + g * asg + m * asm;   Bm row (b, g % GB, n):  bm + b * bsb + (g % GB) * bsg + n * bsn
// one wave = a (32 TM) x (32 TN) tile of one (b, g, slab); grid (slabs, B * G, ceil(tiles / 4)), 256 threads (4 tiles).
constexpr int kF32WgradSlab = 512;   // pixels per partial product
template <int TM, int TN>
__global__ void __launch_bounds__(256)
oss_rows_f32_wgrad_kernel(const float *__restrict__ a, const float *__restrict__ bm, float *__restrict__ part, int M, int N, int P, int G,
                          int GB, int64_t asb, int64_t asg, int64_t asm_, int64_t bsb, int64_t bsg, int64_t bsn) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, kg = lane >> 5;
    const int b = blockIdx.y / G, g = blockIdx.y - b * G, slab = blockIdx.x;
    const int mt = (M + 32 * TM - 1) / (32 * TM), nt = (N + 32 * TN - 1) / (32 * TN);
    const int tile = blockIdx.z * 4 + wave;
    if (tile >= mt * nt) return;
    const int m0 = (tile / nt) * 32 * TM, n0 = (tile % nt) * 32 * TN;
    const int pbeg = slab * kF32WgradSlab, pend = min(P, pbeg + kF32WgradSlab);
    const float *ab = a + b * asb + g * asg, *bb = bm + b * bsb + (g % GB) * bsg;
    const float *ar[TM], *br[TN];
    bool aok[TM], bok[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + 32 * i + col;
        aok[i] = m < M;
        ar[i] = ab + (int64_t)(aok[i] ? m : 0) * asm_;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + 32 * j + col;
        bok[j] = n < N;
        br[j] = bb + (int64_t)(bok[j] ? n : 0) * bsn;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int U = 2;   // 8-pixel steps whose loads are issued together
    for (int p = pbeg; p < pend; p += 8 * U) {
        f32x4 av[U][TM], bv[U][TN];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pp = p + 8 * u;
            const bool ok = pp + 4 * kg < pend;   // P % 4 == 0: a lane's quad is inside or outside as a whole
            const int pc = ok ? pp + 4 * kg : pbeg;   // (a quad past the end reads the slab's first one and is zeroed)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                av[u][i] = *reinterpret_cast<const f32x4 *>(ar[i] + pc);
                if (!(ok && aok[i])) av[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bv[u][j] = *reinterpret_cast<const f32x4 *>(br[j] + pc);
                if (!(ok && bok[j])) bv[u][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[i][j] = mfma_f32(av[u][i][q], bv[u][j][q], acc[i][j]);
    }
    float *pb = part + ((size_t)(b * gridDim.x + slab) * G + g) * M * N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + 32 * j + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (m < M && n < N) pb[(size_t)m * N + n] = acc[i][j][r];
            }
        }
}

__global__ void __launch_bounds__(256)
oss_rows_f32_wgrad_finish(const float *__restrict__ part, float *__restrict__ out, int K, size_t pvec) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pvec) return;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= K; k += 8) {   // eight loads in flight, added in index order
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(k + j) * pvec + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; k < K; ++k) s += part[(size_t)k * pvec + i];
    out[i] = s;
}

int rows_f32_wgrad_slabs(int P) { return (P + kF32WgradSlab - 1) / kF32WgradSlab; }
size_t rows_f32_wgrad_partial_floats(int B, int G, int M, int N, int P) { return (size_t)B * rows_f32_wgrad_slabs(P) * G * M * N; }

int rows_f32_wgrad_ok(int M, int N, int P, const void *a, const void *bm, std::initializer_list<int64_t> strides) {
    if (M < 1 || N < 1 || P < 4 || P % 4 != 0) return 0;
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(bm)) & 15u) return 0;
    for (int64_t st : strides)
        if (st % 4 != 0) return 0;
    return 1;
}

int rows_f32_wgrad(const float *a, const float *bm, float *out, float *part, int B, int G, int GB, int M, int N, int P, int64_t asb,
                   int64_t asg, int64_t asm_, int64_t bsb, int64_t bsg, int64_t bsn, hipStream_t s) {
    if (!rows_f32_wgrad_ok(M, N, P, a, bm, {asb, asg, asm_, bsb, bsg, bsn})) return OSS_ERR_SHAPE;
    if (B <= 0 || G <= 0 || GB <= 0 || (size_t)B * G > 65535) return OSS_ERR_SHAPE;
    const int slabs = rows_f32_wgrad_slabs(P);
    // 64 x 32 tiles; 32 x 32 when the wider tile would leave most SIMDs without a wave
    const int t21 = ((M + 63) / 64) * ((N + 31) / 32), t11 = ((M + 31) / 32) * ((N + 31) / 32);
    if ((long)t21 * slabs * B * G >= 768 && M > 32) {
        const dim3 grid(slabs, B * G, (t21 + 3) / 4);
        hipLaunchKernelGGL((oss_rows_f32_wgrad_kernel<2, 1>), grid, dim3(256), 0, s, a, bm, part, M, N, P, G, GB, asb, asg, asm_, bsb, bsg, bsn);
    } else {
        const dim3 grid(slabs, B * G, (t11 + 3) / 4);
        hipLaunchKernelGGL((oss_rows_f32_wgrad_kernel<1, 1>), grid, dim3(256), 0, s, a, bm, part, M, N, P, G, GB, asb, asg, asm_, bsb, bsg, bsn);
    }
    const size_t pvec = (size_t)G * M * N;
    if (defer_finish())
        defer_sum(part, slabs * B, pvec, pvec, out, pvec, nullptr);
    else
        hipLaunchKernelGGL(oss_rows_f32_wgrad_finish, dim3((unsigned)((pvec + 255) / 256)), dim3(256), 0, s, part, out, slabs * B, pvec);
    return (int)hipGetLastError();
}

}  // namespace oss
