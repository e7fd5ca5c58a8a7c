// oss_conv1x1_wgk.hip -- workgroup-level 1x1 convolution for WIDE contractions (192 < K <= 512, M <= 128 output rows): the two
// GEMMs of the EFFN that the kernel of oss_conv1x1_wg.hip cannot take -- project_out (hidden = int(2.66 dim) = 255 -> dim, with the
// block's skip connection) and the input gradient of project_in (2 hidden = 510 -> dim) -- FeedForward, SRGAN/VmambaIR/archs/
// MambaSISR6_arch.py:201-218.  (round 6)
//
// They ran on the wave-level kernels of oss_conv1x1.hip (oss_conv1x1_pairw / _pairk: 14.7 / 19.3 us per launch against 4.9 / 6.5 us
// of memory time at the headline shapes, 1.8 ms per step): every wave builds its own copy of the activation operand from 4-byte loads
// that walk the channel axis.  Here, as in oss_conv1x1_wg.hip, a workgroup copies its activation tile into LDS as it lies in memory
// and reads the MFMA fragments back with ds_read_b64_tr_b16 -- but the K axis is walked in CHUNKS of 128 channels (the whole
// K x 128-pixel tile of K = 510 would be 163 KB), and the weights of a chunk are staged in LDS too, ONCE per
// workgroup (each wave owns one 32-row tile of the output over all 128 pixels, so all waves read the same activation tile and
// disjoint weight rows):
//   * plain form   (y = W x,  W (M, K) row-major, K odd: rows are not 16-byte aligned): the chunk's (M x 128) block is read with
//     lane-consecutive 4-byte loads (a wave = 256 contiguous bytes of one row), narrowed in pairs and written as a [m][k] image;
//     a lane's A fragment (8 consecutive k of its row) is one ds_read_b128;
//   * transposed form (dx = W^T dy, W (K, M) row-major): the chunk's 128 rows are ONE contiguous block, copied with 16-byte loads
//     into a [k][m] image and read back with the same transpose-read as the activations (m in the place of the pixel).
// Channels past K (the tail of the last chunk) are zero in BOTH images.  The next chunk's global loads are issued before the
// current chunk's MFMAs and land in registers; a barrier after the MFMAs frees the (single) pair of LDS images, the registers are
// written into them, a second barrier publishes them: 80 KB of LDS.  Results leave through a wave-private LDS tile (aliasing the
// operand images) as 16-byte stores, with bias and the skip connection added in fp32 and rounded once.
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "oss_device.h"
#include "oss_host.h"
#include "oss_mfma.h"

namespace oss {

typedef short wgk_s16x4 __attribute__((ext_vector_type(4)));

constexpr int kWgkPT = 128;            // pixels per workgroup
constexpr int kWgkKC = 128;            // channels per chunk
constexpr int kWgkPX = kWgkPT + 32;    // row pitch of the activation image and of the [k][m] weight image (see oss_conv1x1_wg.hip)
constexpr int kWgkMK = kWgkKC + 8;     // row pitch of the [m][k] weight image: 272 bytes, 16-byte aligned rows
constexpr int kWgkXB = kWgkKC * kWgkPX;                 // elements per activation buffer
constexpr int kWgkWB = 128 * kWgkPX;                    // elements per weight buffer (covers both images: 128 x 160 >= 128 x 136)

template <typename T, bool WT, bool RES>
__global__ void __launch_bounds__(256)
oss_conv1x1_wgk_kernel(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias, T *__restrict__ y, int M,
                       int K, int P, int64_t xsb, int64_t xsk, const T *__restrict__ res) {
    constexpr int PT = kWgkPT, KC = kWgkKC, PX = kWgkPX, MK = kWgkMK, NCT = PT / 32;
    constexpr int PITCH = PT + 8;                                   // output staging tile
    using OT = typename std::conditional<RES, float, T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char wgk_smem[];
    T *xs = reinterpret_cast<T *>(wgk_smem);                        // [KC][PX]
    T *ws = xs + kWgkXB;                                            // [128][PX] (plain: [m][MK]; transposed: [k][PX])
    OT *os = reinterpret_cast<OT *>(wgk_smem);                      // [4 waves][32][PITCH], after the K loop (aliases the operands)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, p0 = blockIdx.x * PT;
    const T *xb = x + b * xsb + p0;
    const int col = lane & 31, kg = lane >> 5;
    const int nch = (K + KC - 1) / KC;
    const int mt_total = (M + 31) >> 5;
    // ---- staging registers of one chunk ---------------------------------------------------------------------------------
    constexpr int XN = KC * (PT / 8) / 256;                         // 16-byte pieces of the activation chunk per thread (8)
    constexpr int WNV = KC * 128 / 4 / 256;                         // transposed form: 16-byte pieces of [128 k][<= 128 m] per thread (16)
    constexpr int WNP = 128 * KC / 2 / 256;                         // plain form: k-pairs of [<= 128 m][128 k] per thread (32)
    u32x4 xr[XN];
    f32x4 wv[WT ? WNV : 1];
    float wp[WT ? 1 : WNP][2];
    const int Mq = (M + 3) >> 2;                                    // transposed form: 16-byte pieces per weight row
    auto issue = [&](int c) {
        const int k0 = c * KC;
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int idx = tid + i * 256, r = idx >> 4, pc = idx & 15;     // 16 pieces per 128-pixel row
            const int k = min(k0 + r, K - 1);
            xr[i] = *reinterpret_cast<const u32x4 *>(xb + (size_t)k * xsk + 8 * pc);
        }
        if constexpr (WT) {
#pragma unroll
            for (int i = 0; i < WNV; ++i) {
                const int idx = tid + i * 256;
                const int r = idx / Mq, q = idx - r * Mq;                   // row k0 + r, columns 4 q .. 4 q + 3
                const int k = min(k0 + min(r, KC - 1), K - 1);
                wv[i] = *reinterpret_cast<const f32x4 *>(w + (size_t)k * M + 4 * min(q, Mq - 1));
            }
        } else {
#pragma unroll
            for (int i = 0; i < WNP; ++i) {
                const int idx = tid + i * 256, m = idx >> 6, kp = idx & 63; // row m, k = k0 + 2 kp, + 1: lanes walk k (256 contiguous bytes)
                const float *row = w + (size_t)min(m, M - 1) * K;
                wp[i][0] = row[min(k0 + 2 * kp, K - 1)];
                wp[i][1] = row[min(k0 + 2 * kp + 1, K - 1)];
            }
        }
    };
    auto commit = [&](int c) {
        const int k0 = c * KC;
        T *xd = xs;
        T *wd = ws;
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int idx = tid + i * 256, r = idx >> 4, pc = idx & 15;
            const u32x4 v = (k0 + r < K) ? xr[i] : u32x4{0u, 0u, 0u, 0u};   // channels past K: zero
            *reinterpret_cast<u32x4 *>(xd + r * PX + 8 * pc) = v;
        }
        if constexpr (WT) {
#pragma unroll
            for (int i = 0; i < WNV; ++i) {
                const int idx = tid + i * 256;
                const int r = idx / Mq, q = idx - r * Mq;
                if (r < KC) {
                    const bool ok = k0 + r < K;
                    const f32x4 v = wv[i];
                    const u32x2 o = ok ? u32x2{pack2<T>(v.x, v.y), pack2<T>(v.z, v.w)} : u32x2{0u, 0u};
                    *reinterpret_cast<u32x2 *>(wd + r * PX + 4 * q) = o;    // [k][m]; columns past M are never read (clamped rows)
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < WNP; ++i) {
                const int idx = tid + i * 256, m = idx >> 6, kp = idx & 63;
                const float a = (k0 + 2 * kp < K) ? wp[i][0] : 0.f, bq = (k0 + 2 * kp + 1 < K) ? wp[i][1] : 0.f;
                if (m < M) *reinterpret_cast<uint32_t *>(wd + m * MK + 2 * kp) = pack2<T>(a, bq);   // [m][k]
            }
        }
    };
    // ---- K loop ----------------------------------------------------------------------------------------------------------
    const int i16 = lane & 15, g = lane >> 4;
    const int tr_off = (8 * (g >> 1) + (i16 >> 2)) * PX + 16 * (g & 1) + 4 * (i16 & 3);   // see oss_conv1x1_wg.hip
    const bool mine = wave < mt_total;                              // this wave's 32-row tile of the output
    const int m0 = wave * 32;
    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    const float bl = (bias && mine) ? bias[min(m0 + col, M - 1)] : 0.f;
    issue(0);
    commit(0);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) issue(c + 1);                              // in flight during this chunk's MFMAs
        if (mine) {
            const T *xc = xs;
            const T *wc = ws;
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                s16x8 af;
                if constexpr (WT) {
                    const T *ap = wc + ks * 16 * PX + m0 + tr_off;
                    const wgk_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wgk_s16x4 *)(ap));
                    const wgk_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wgk_s16x4 *)(ap + 4 * PX));
                    af = s16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                } else {
                    const u32x4 q = *reinterpret_cast<const u32x4 *>(wc + (m0 + col) * MK + ks * 16 + kg * 8);
                    af = __builtin_bit_cast(s16x8, q);
                }
                if (m0 + col >= M) af = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const T *bp = xc + ks * 16 * PX + ct * 32 + tr_off;
                    const wgk_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wgk_s16x4 *)(bp));
                    const wgk_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wgk_s16x4 *)(bp + 4 * PX));
                    const s16x8 bf = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    acc[ct] = Mfma<T>::run(af, bf, acc[ct]);
                }
            }
        }
        __syncthreads();                                            // every wave is done reading the images of chunk c
        if (c + 1 < nch) {
            commit(c + 1);
            __syncthreads();
        }
    }
    // ---- epilogue: results (+ bias) -> the wave's LDS tile [row][pixel] -> 16-byte stores (+ the residual, rounded once) --------
    if (!mine) return;
    OT *ow = os + wave * 32 * PITCH;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
        const float bv = __int_as_float(__builtin_amdgcn_ds_bpermute(row << 2, __float_as_int(bl)));
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            if constexpr (RES) ow[row * PITCH + ct * 32 + col] = acc[ct][r] + bv;
            else               ow[row * PITCH + ct * 32 + col] = from_f32<T>(acc[ct][r] + bv);
        }
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int CPR = PT / 8;
#pragma unroll
    for (int q = 0; q < 32 * CPR / 64; ++q) {
        const int chunk = q * 64 + lane, row = chunk / CPR, pc = chunk - row * CPR;
        if (m0 + row < M) {
            const size_t o = ((size_t)b * M + m0 + row) * P + p0 + 8 * pc;
            if constexpr (RES) {
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(ow + row * PITCH + 8 * pc);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(ow + row * PITCH + 8 * pc + 4);
                const u32x4 rq = *reinterpret_cast<const u32x4 *>(res + o);
                float r8[8];
                unpack2<T>(rq.x, r8[0], r8[1]); unpack2<T>(rq.y, r8[2], r8[3]); unpack2<T>(rq.z, r8[4], r8[5]); unpack2<T>(rq.w, r8[6], r8[7]);
                *reinterpret_cast<u32x4 *>(y + o) = u32x4{pack2<T>(v0.x + r8[0], v0.y + r8[1]), pack2<T>(v0.z + r8[2], v0.w + r8[3]),
                                                          pack2<T>(v1.x + r8[4], v1.y + r8[5]), pack2<T>(v1.z + r8[6], v1.w + r8[7])};
            } else {
                *reinterpret_cast<u32x4 *>(y + o) = *reinterpret_cast<const u32x4 *>(ow + row * PITCH + 8 * pc);
            }
        }
    }
}

static size_t wgk_lds_bytes(bool res) {
    const size_t operands = (size_t)(kWgkXB + kWgkWB) * 2;                           // one activation + one weight image, 16-bit
    const size_t staging = (size_t)4 * 32 * (kWgkPT + 8) * (res ? 4 : 2);
    return operands > staging ? operands : staging;
}

// VMAMBAIR_CONV1X1_WGK=0: the wide-K 1x1 convolutions stay on the wave-level kernels (A-B timing)
static std::atomic<int> g_wgk{-1};
void conv1x1_set_wgk(int on) { g_wgk.store(on ? 1 : 0); }
static bool wgk_on() {
    int v = g_wgk.load();
    if (v < 0) {
        const char *e = std::getenv("VMAMBAIR_CONV1X1_WGK");
        v = (e && e[0] == '0') ? 0 : 1;
        g_wgk.store(v);
    }
    return v != 0;
}

// 1 when the K-chunked workgroup-level kernel takes the shape (wt: the transposed-weight form)
int conv1x1_wgk_ok(oss_dtype io, int M, int K, int P, int64_t xsb, int64_t xsk, const void *x, const void *y, const float *w,
                   const void *res, int wt) {
    if (!wgk_on()) return 0;
    if (io != OSS_BF16 && io != OSS_F16) return 0;
    if (K <= 192 || K > 512 || M < 1 || M > 128 || P % kWgkPT != 0) return 0;
    if (wt && M % 4 != 0) return 0;                                                  // 16-byte pieces of a weight row
    if (xsb % 8 != 0 || xsk % 8 != 0) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res)) & 15u) return 0;
    if (wt && (reinterpret_cast<uintptr_t>(w) & 15u)) return 0;
    return wgk_lds_bytes(res != nullptr) <= kMaxLdsBytes ? 1 : 0;
}

template <typename T>
static int wgk_launch(const T *x, const float *w, const float *bias, T *y, int B, int M, int K, int P, int64_t xsb, int64_t xsk, int wt,
                      const T *res, hipStream_t s) {
    const size_t smem = wgk_lds_bytes(res != nullptr);
    dim3 grid(P / kWgkPT, B);
#define OSS_WGK(WT_, RES_)                                                                                           \
    do {                                                                                                             \
        static LdsGate gate;                                                                                         \
        auto kern = oss_conv1x1_wgk_kernel<T, WT_, RES_>;                                                            \
        if (const int e = gate.ensure(reinterpret_cast<const void *>(kern), smem)) return e;                         \
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, x, w, bias, y, M, K, P, xsb, xsk, res);                   \
    } while (0)
    if (wt) { if (res) OSS_WGK(true, true); else OSS_WGK(true, false); }
    else    { if (res) OSS_WGK(false, true); else OSS_WGK(false, false); }
#undef OSS_WGK
    return (int)hipGetLastError();
}

int conv1x1_wgk(oss_dtype io, const void *x, const float *w, const float *bias, void *y, int B, int M, int K, int P, int64_t xsb,
                int64_t xsk, int wt, hipStream_t s, const void *res) {
    if (!conv1x1_wgk_ok(io, M, K, P, xsb, xsk, x, y, w, res, wt)) return OSS_ERR_SHAPE;
    if (io == OSS_BF16)
        return wgk_launch<bf16_t>(reinterpret_cast<const bf16_t *>(x), w, bias, reinterpret_cast<bf16_t *>(y), B, M, K, P, xsb, xsk, wt,
                                  reinterpret_cast<const bf16_t *>(res), s);
    return wgk_launch<f16_t>(reinterpret_cast<const f16_t *>(x), w, bias, reinterpret_cast<f16_t *>(y), B, M, K, P, xsb, xsk, wt,
                             reinterpret_cast<const f16_t *>(res), s);
}

}  // namespace oss
