// oss_conv1x1_wg.hip -- workgroup-level forward / input-gradient form of the 1x1 convolutions of the OSS block
// (in_conv, out_conv, project_in and their siblings, SRGAN/VmambaIR/archs/MambaSISR6_arch.py:205,211,281,329) for K % 16 == 0,
// K <= 192, H W % 128 == 0, 16-bit I/O, incl. bias and the fused skip connection.
//
// The wave-level kernels of oss_conv1x1.hip build the MFMA activation operand (8 input channels of ONE pixel per lane) from
// 4-byte loads that walk the channel axis of an NCHW tensor -- one load per channel and pixel pair, every wave its own copy
// -- and store their results 4 bytes per lane; at the headline shapes they run as chains of latencies at one wave per SIMD
// (DESIGN.md section 9).  Here a workgroup
//   1. copies its activation tile x[b, 0..K, p0..p0+128) into LDS AS IT LIES IN MEMORY ([channel][pixel], 16-byte accesses on
//      both sides: each element crosses the CU's load path once per workgroup instead of once per wave and row-tile split),
//   2. reads the operand fragments back with ds_read_b64_tr_b16 -- the LDS transpose-read of gfx950: a 16-lane group reads a
//      [4 channels][16 pixels] block, 8 bytes per lane, and every lane receives the four channels of ONE pixel -- so the
//      [pixel][8 channels] fragment the MFMA wants comes out of a [channel][pixel] image without any strided access,
//   3. gives every wave whole 32-row tiles of the output over all 128 pixels (four MFMA column tiles per weight fragment),
//   4. sends the results through a wave-private LDS tile and out as 16-byte stores along the pixel axis.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "oss_device.h"
#include "oss_host.h"
#include "oss_mfma.h"

namespace oss {

typedef short s16x4 __attribute__((ext_vector_type(4)));

// Copy of a [K][PT] activation tile (16-byte chunks, rows xsk apart) into LDS rows of PITCH elements, 256 threads.
// (round 4) Written as `for (idx = tid; idx < K * CPR; idx += 256) lds[..] = global[..]` this was a loop of load -> s_waitcnt vmcnt(0)
// -> ds_write round trips, one per chunk and thread (6 at K = 96, 12 at K = 192), at the head of a kernel that runs ONE workgroup
// per CU -- nothing else on the CU to hide them behind.  Now the loads of up to GROUP chunks are issued back to back before the
// first LDS write waits for its data.
template <typename T, int K, int PT, int PITCH, int GROUP = 6>
__device__ __forceinline__ void copy_tile_to_lds(const T *__restrict__ src, int64_t xsk, T *__restrict__ dst, int tid) {
    constexpr int CPR = PT / 8, TOTAL = K * CPR, NCP = (TOTAL + 255) / 256;
#pragma unroll
    for (int i0 = 0; i0 < NCP; i0 += GROUP) {
        u32x4 q[GROUP];
#pragma unroll
        for (int i = 0; i < GROUP; ++i) {
            if (i0 + i < NCP) {
                const int idx = min(tid + (i0 + i) * 256, TOTAL - 1);
                const int c = idx / CPR, pc = idx - c * CPR;
                q[i] = *reinterpret_cast<const u32x4 *>(src + c * xsk + 8 * pc);
            }
        }
#pragma unroll
        for (int i = 0; i < GROUP; ++i) {
            if (i0 + i < NCP) {
                const int idx = tid + (i0 + i) * 256;
                const int c = idx / CPR, pc = idx - c * CPR;
                if (TOTAL % 256 == 0 || idx < TOTAL) *reinterpret_cast<u32x4 *>(dst + c * PITCH + 8 * pc) = q[i];
            }
        }
    }
}

// W(m, k) = w[m * ws_m + k * ws_k] as in oss_conv1x1.hip: forward ws_m = K, ws_k = 1; input gradient ws_m = 1, ws_k = M.
// PT: pixels per workgroup (128; 64 when 128-pixel tiles would leave the chip with fewer than two workgroups per CU).
// RES: y = W x + bias + res (the block's skip connection); the results then wait in LDS as fp32 so that the sum is rounded once.
// PF: the next row tile's weight fragments are requested before the current tile's MFMAs (KS <= 6: two tiles' raw fragments fit).
// LN: the activation tile is first LayerNorm-ed over its channels, per pixel, inside LDS (SS2D_1 / EFFN read norm1(x) / norm2(x):
// MambaSISR6_arch.py:514-516, LayerNorm :144-195) -- the normalised tile is also written out (the weight gradient's operand) with
// mean / rstd (the LayerNorm backward's), so the separate LayerNorm launch and its round trip through memory are gone.
template <typename T>
struct WgLnArgs {
    const float *w, *b;   // LayerNorm weight, bias (NULL: the BiasFree form, x * rstd * w)
    T *n;                 // (B, K, P) normalised activations
    float *mean, *rstd;   // (B, P)
    float eps;
};

template <typename T, int KS, bool WT, int PT, bool RES, bool PF, bool LN>
__global__ void __launch_bounds__(256)
oss_conv1x1_wg_kernel(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias, T *__restrict__ y,
                      int M, int P, int64_t xsb, int64_t xsk, const T *__restrict__ res, WgLnArgs<T> ln) {
    // LDS row pitches in elements (16-byte aligned rows).  (round 5) Two pitches: profiles/r04_pmc_lds_conflicts_per_kernel.txt had
    // 44-47 % of this kernel's LDS cycles as bank-conflict cycles; the banking rules are per instruction (MI355X_MICROARCH.md, LDS).
    // (a) The activation tile: ds_read_b64_tr_b16 is serviced in two 32-lane halves over 64 banks; a half = two 16-lane groups
    // (pixel blocks 32 bytes apart), each 4 channel rows x 32 bytes.  With rows PT + 8 elements (272 bytes) apart the eight pieces
    // sat at bank bytes 0,16,32,48 | 32,48,64,80: three deep.  Rows 64 bytes past a bank sweep (PT + 32 elements = 320 bytes at
    // PT = 128, 192 at PT = 64) put them at 0,64,128,192 | 32,96,160,224: one sweep, no overlap.
    // (b) The output staging tile (PT = 128): PT + 16, with 4-byte writes and a rotated read-back (below).
    constexpr int K = 16 * KS, NCT = PT / 32;
#ifdef OSS_EXP_WG_OLD_PITCH   // (OSS_EXP_*: A-B timing builds only, tools/build_experiment.sh)
    constexpr int PX = PT + 8;
#else
    constexpr int PX = PT + 32;
#endif
#ifdef OSS_EXP_WG_OLD_EPI
    constexpr bool kPairEpi = false;
#else
    constexpr bool kPairEpi = !RES && PT == 128;
#endif
    constexpr int PITCH = kPairEpi ? PT + 16 : PT + 8;   // output staging tile
    using OT = typename std::conditional<RES, float, T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
    T *xs = reinterpret_cast<T *>(wg_smem);                    // [K][PX]
    OT *os = reinterpret_cast<OT *>(xs + K * PX);              // [4 waves][32][PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, p0 = blockIdx.x * PT;
    const T *xb = x + b * xsb + p0;
    const int col = lane & 31, kg = lane >> 5;
    const int mt_total = (M + 31) >> 5;
    struct WRaw { f32x4 lo[KS], hi[KS]; float bl; };
    auto issue = [&](int mt, WRaw &r) {   // weight fragments + the bias value of row m0 + col: one group of loads
        const int mrow = min(mt * 32 + col, M - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = ks * 16 + kg * 8;
#ifdef OSS_EXP_WG_HALF_W   // TIMING ONLY (wrong results): half of the weight bytes -- what 16-bit master copies would cost to load
            if constexpr (!WT) {
                r.lo[ks] = *reinterpret_cast<const f32x4 *>(w + (size_t)mrow * K + k0);
                r.hi[ks] = r.lo[ks];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    r.lo[ks][e] = w[(size_t)(k0 + e) * M + mrow];
                    r.hi[ks][e] = r.lo[ks][e];
                }
            }
#else
            if constexpr (!WT) {
                r.lo[ks] = *reinterpret_cast<const f32x4 *>(w + (size_t)mrow * K + k0);
                r.hi[ks] = *reinterpret_cast<const f32x4 *>(w + (size_t)mrow * K + k0 + 4);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    r.lo[ks][e] = w[(size_t)(k0 + e) * M + mrow];
                    r.hi[ks][e] = w[(size_t)(k0 + 4 + e) * M + mrow];
                }
            }
#endif
        }
        r.bl = bias ? bias[mrow] : 0.f;
    };
    WRaw cur;
    // row tiles of this workgroup: (wave + 4 j) * gridDim.z + blockIdx.z -- the launch may split a problem's row tiles over
    // gridDim.z workgroups (each stages the activation tile again) when the pixel tiles alone would not fill the chip
    const int nz = gridDim.z, zi = blockIdx.z;
    if (wave * nz + zi < mt_total) issue(wave * nz + zi, cur);   // in flight across the activation copy
    // (round 4) LayerNorm weight / bias of all K channels: requested here with everything else, parked in LDS behind the output staging
    // area.  They used to be read where they are used, `if (c < K) { wc = ln.w[c]; ...` -- a branch, two loads and s_waitcnt vmcnt(0)
    // per channel of the thread: 12 dependent round trips at K = 96 in a kernel that runs one workgroup per CU.
    float *lnw_s = reinterpret_cast<float *>(os + 4 * 32 * PITCH), *lnb_s = lnw_s + K;
    float lnw_r = 0.f, lnb_r = 0.f;
    if constexpr (LN) {
        const int c = min(tid, K - 1);
        lnw_r = ln.w[c];
        lnb_r = ln.b ? ln.b[c] : 0.f;
    }
    // 1. the activation tile, 16 bytes per lane: lane -> (channel, 8-pixel chunk), a row's chunks on consecutive lanes
    constexpr int CPR = PT / 8;   // chunks per row
    copy_tile_to_lds<T, K, PT, PX>(xb, xsk, xs, tid);
    if constexpr (LN) {
        if (tid < K) { lnw_s[tid] = lnw_r; lnb_s[tid] = lnb_r; }
    }
    __syncthreads();
    if constexpr (LN) {
        // a thread owns FOUR adjacent pixels (one 8-byte LDS access per channel) and every NPARTth .. channel: two-pass mean /
        // variance in fp32 as oss_ln_nchw_fwd_kernel, the partial sums of a pixel combined through LDS in a fixed order
        constexpr int QPT = PT / 4, NPART = 256 / QPT, CPT = (K + NPART - 1) / NPART;   // 128 px: 32 quads x 8 parts
        float *red = reinterpret_cast<float *>(os);          // [NPART][PT], the output staging area is still unused
        const int quad = tid % QPT, part = tid / QPT, px = 4 * quad;
        float v[CPT][4];
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = part + i * NPART;
            if (c < K) {
                const u32x2 q = *reinterpret_cast<const u32x2 *>(xs + c * PX + px);
                unpack2<T>(q.x, v[i][0], v[i][1]); unpack2<T>(q.y, v[i][2], v[i][3]);
            } else {
                v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
            }
        }
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < CPT; ++i)
#pragma unroll
            for (int u = 0; u < 4; ++u) sum[u] += v[i][u];
        *reinterpret_cast<f32x4 *>(red + part * PT + px) = f32x4{sum[0], sum[1], sum[2], sum[3]};
        __syncthreads();
        float mu[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NPART; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(red + q * PT + px);
            mu[0] += t.x; mu[1] += t.y; mu[2] += t.z; mu[3] += t.w;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) mu[u] /= (float)K;
        __syncthreads();
        float sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            if (part + i * NPART < K) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { const float d = v[i][u] - mu[u]; sq[u] = __builtin_fmaf(d, d, sq[u]); }
            }
        }
        *reinterpret_cast<f32x4 *>(red + part * PT + px) = f32x4{sq[0], sq[1], sq[2], sq[3]};
        __syncthreads();
        float rstd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NPART; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(red + q * PT + px);
            rstd[0] += t.x; rstd[1] += t.y; rstd[2] += t.z; rstd[3] += t.w;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) rstd[u] = 1.0f / sqrtf(rstd[u] / (float)K + ln.eps);
        if (part == 0 && zi == 0) {
            *reinterpret_cast<f32x4 *>(ln.mean + (size_t)b * P + p0 + px) = f32x4{mu[0], mu[1], mu[2], mu[3]};
            *reinterpret_cast<f32x4 *>(ln.rstd + (size_t)b * P + p0 + px) = f32x4{rstd[0], rstd[1], rstd[2], rstd[3]};
        }
        const bool with_bias = ln.b != nullptr;
        float mu_c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) mu_c[u] = with_bias ? mu[u] : 0.f;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = part + i * NPART;
            if (c < K) {
                const float wc = lnw_s[c], bc = lnb_s[c];
                float o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) o[u] = (v[i][u] - mu_c[u]) * rstd[u] * wc + bc;   // BiasFree: mu_c = 0, bc = 0 (x * rstd * w)
                *reinterpret_cast<u32x2 *>(xs + c * PX + px) = u32x2{pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3])};
            }
        }
        __syncthreads();   // (also: red[] has been read by everyone before the output staging reuses it)
        T *nb = ln.n + (size_t)b * K * P + p0;
        for (int idx = tid; zi == 0 && idx < K * CPR; idx += 256) {
            const int c = idx / CPR, pc = idx - c * CPR;
            *reinterpret_cast<u32x4 *>(nb + (size_t)c * P + 8 * pc) = *reinterpret_cast<const u32x4 *>(xs + c * PX + 8 * pc);
        }
    }
    const int i16 = lane & 15, g = lane >> 4;
    // transpose-read address of this lane inside a [4][16] block: row i16 / 4, pixels 4 (i16 % 4) .. + 3; the lane then HOLDS
    // pixel i16 of the block.  Blocks of the 16-lane groups: pixels 16 (g & 1) + .. of the column tile, channels 8 (g >> 1) + ..
    const int tr_off = (8 * (g >> 1) + (i16 >> 2)) * PX + 16 * (g & 1) + 4 * (i16 & 3);
    OT *ow = os + wave * 32 * PITCH;
    for (int mt = wave * nz + zi; mt < mt_total; mt += 4 * nz) {
        const int m0 = mt * 32;
        WRaw nxt;
        if constexpr (PF) { if (mt + 4 * nz < mt_total) issue(mt + 4 * nz, nxt); }
        f32x16 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const s16x8 af = m0 + col < M ? cvt8<T>(cur.lo[ks], cur.hi[ks]) : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const T *bp = xs + ks * 16 * PX + ct * 32 + tr_off;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(bp));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(bp + 4 * PX));
                const s16x8 bf = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                acc[ct] = Mfma<T>::run(af, bf, acc[ct]);
            }
        }
        // 4. results (+ bias) -> the wave's LDS tile [row][pixel] -> 16-byte stores (+ the residual, rounded once)
        if constexpr (!kPairEpi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
                const float bv = __int_as_float(__builtin_amdgcn_ds_bpermute(row << 2, __float_as_int(cur.bl)));
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    if constexpr (RES) ow[row * PITCH + ct * 32 + col] = acc[ct][r] + bv;
                    else               ow[row * PITCH + ct * 32 + col] = from_f32<T>(acc[ct][r] + bv);
                }
            }
        } else {
            // (round 5) 16-bit results: one 4-byte write per lane and register PAIR instead of two 2-byte writes (two lanes per bank
            // word each).  Registers r and r + 2 hold rows A and A + 2 of the same pixel column; neighbouring lanes swap one value
            // (DPP quad_perm [1,0,3,2]) so that the even lane owns pixels (col, col + 1) of row A and the odd lane pixels
            // (col - 1, col) of row A + 2.  ds_write_b32 is serviced in 32-lane halves over 32 banks: rows are 72 words apart, so the
            // even lanes of a half cover words w .. w+15 and the odd lanes w+144 .. = w+16 .. w+31 (mod 32): 32 lanes, 32 banks.
            const bool odd = lane & 1;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = (rr & 1) + 4 * (rr >> 1);          // 0,1, 4,5, 8,9, 12,13: paired with r + 2
                constexpr int kPairStep = 2;
                const int rowA = (r & 3) + 8 * (r >> 2) + 4 * kg, rowB = rowA + 2;
                const float bvA = __int_as_float(__builtin_amdgcn_ds_bpermute(rowA << 2, __float_as_int(cur.bl)));
                const float bvB = __int_as_float(__builtin_amdgcn_ds_bpermute(rowB << 2, __float_as_int(cur.bl)));
                T *dst = ow + (odd ? rowB : rowA) * PITCH + (col & ~1);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const float va = acc[ct][r] + bvA, vb = acc[ct][r + kPairStep] + bvB;
                    const float give = odd ? va : vb;
                    const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(give), 0xB1, 0xf, 0xf, false));
                    *reinterpret_cast<uint32_t *>(dst + ct * 32) = odd ? pack2<T>(recv, vb) : pack2<T>(va, recv);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 32 * CPR / 64; ++q) {
            const int chunk = q * 64 + lane, row = chunk / CPR;
            // ds_read_b128 is serviced in four fixed 16-lane groups that are conflict-free on 1024 contiguous bytes; rows 288 bytes
            // apart are shifted by two 16-byte slots per row, so the lane takes the chunk two slots back per row: same banks as if
            // the rows were contiguous (the global stores stay whole 16-byte pieces of a row, in a rotated order)
            const int pc = kPairEpi ? ((chunk - row * CPR) - 2 * row) & (CPR - 1) : chunk - row * CPR;
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
        __builtin_amdgcn_wave_barrier();
        if constexpr (PF) { cur = nxt; } else { if (mt + 4 * nz < mt_total) issue(mt + 4 * nz, cur); }
    }
}

#ifdef OSS_EXP_WG_OLD_PITCH
constexpr int kWgXPad = 8;
#else
constexpr int kWgXPad = 32;
#endif
// (the staging tile is sized for its widest pitch, PT + 16)
static size_t wg_lds_bytes(int K, int pt, bool res) { return (size_t)K * 2 * (pt + kWgXPad) + (size_t)4 * 32 * (res ? 4 : 2) * (pt + 16) + 2 * (size_t)K * sizeof(float); }   // + the LayerNorm weight / bias image
// pixels per workgroup (64 | 128) and the row-tile split (gridDim.z): 0 = by shape, else forced (A-B timing).  Measured
// (tools/conv_wg_test.py, and inside the SR and Deraining steps): K = 96 / 192 want 128 pixels when that still gives a
// workgroup per CU, K <= 48 wants 64; a launch with fewer workgroups than CUs (Deraining levels 1.. at batch 4) loses to the
// wave-level kernels unless its row tiles are split over more workgroups.
static std::atomic<int> g_wg_pixels{-1};
void conv1x1_wg_set_pixels(int pt) { g_wg_pixels.store(pt == 64 || pt == 128 ? pt : 0); }
static void wg_shape(int B, int M, int K, int P, int &pt, int &nz) {
    int f = g_wg_pixels.load();
    if (f < 0) {
        const char *e = std::getenv("VMAMBAIR_CONV1X1_WG_PIXELS");
        const int v = e ? std::atoi(e) : 0;
        f = (v == 64 || v == 128) ? v : 0;
        g_wg_pixels.store(f);
    }
    const long t128 = (long)B * (P / 128);
    pt = f ? f : ((K <= 48 && t128 < 1024) || t128 < 256 ? 64 : 128);
    const long wgs = (long)B * (P / pt);
    const int groups = ((M + 31) / 32 + 3) / 4;           // row tiles in units of a workgroup's four waves
    static const int target = [] { const char *e = std::getenv("VMAMBAIR_CONV1X1_WG_TARGET"); return e ? std::atoi(e) : 256; }();
    nz = (int)std::min<long>(groups, (target + wgs - 1) / wgs);
    if (nz < 1) nz = 1;
}

static constexpr bool wg_ks_built(int ks) { return ks == 1 || ks == 2 || ks == 3 || ks == 4 || ks == 6 || ks == 8 || ks == 12; }
// 1 when the workgroup-level kernel takes the shape
int conv1x1_wg_ok(oss_dtype io, int M, int K, int P, int64_t xsb, int64_t xsk, const void *x, const void *y, const float *w,
                  const void *res) {
    // The fused skip connection (RES) is written and parity-green, but inside the training step it measured SLOWER than the
    // wave-level kernel (10.6 vs 9.4 us at 96 -> 96: the residual arrives cold from HBM behind the fp32 LDS round trip), so
    // those calls stay where they were and the RES instantiations are not built.
    if (res) return 0;
    if (io != OSS_BF16 && io != OSS_F16) return 0;
    if (K % 16 != 0 || K < 16 || K > 192 || P % 128 != 0 || M < 1) return 0;
    // built for the widths a power-of-two-times-{1,3} base width gives (K = 16, 32, 48, 64, 96, 128, 192: dim 16 / 32 / 48 / 64 and
    // their doublings); K = 80, 112, 144, 160, 176 stay on the wave-level kernels (round 5: 80 instantiations nobody launched)
    if (!wg_ks_built(K / 16)) return 0;
    if (xsb % 8 != 0 || xsk % 8 != 0) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(res)) & 15u)
        return 0;
    return wg_lds_bytes(K, 128, res != nullptr) <= kMaxLdsBytes ? 1 : 0;
}

template <typename T, bool WT, bool LN = false>
static int wg_launch(const T *x, const float *w, const float *bias, T *y, int B, int M, int K, int P, int64_t xsb, int64_t xsk,
                     const T *res, hipStream_t s, WgLnArgs<T> ln = WgLnArgs<T>{}) {
    int pt, nz;
    wg_shape(B, M, K, P, pt, nz);
    const size_t smem = wg_lds_bytes(K, pt, res != nullptr);
    dim3 grid(P / pt, B, nz);
#define OSS_WG3(KS_, PT_, RES_)                                                                                      \
    do {                                                                                                             \
        static LdsGate gate;                                                                                         \
        auto kern = oss_conv1x1_wg_kernel<T, KS_, WT, PT_, RES_, (KS_ <= 6), LN>;                                        \
        if (const int e = gate.ensure(reinterpret_cast<const void *>(kern), smem)) return e;                         \
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, x, w, bias, y, M, P, xsb, xsk, res, ln);                      \
    } while (0)
#define OSS_WG(KS_)                                                                                                  \
    do {                                                                                                             \
        if (pt == 128) OSS_WG3(KS_, 128, false); else OSS_WG3(KS_, 64, false);                                       \
    } while (0)
    switch (K / 16) {
        case 1: OSS_WG(1); break;
        case 2: OSS_WG(2); break;
        case 3: OSS_WG(3); break;
        case 4: OSS_WG(4); break;
        case 6: OSS_WG(6); break;
        case 8: OSS_WG(8); break;
        case 12: OSS_WG(12); break;
        default: return OSS_ERR_SHAPE;
    }
#undef OSS_WG
#undef OSS_WG3
    return (int)hipGetLastError();
}

// wt = 0: y = W x + bias [+ res] with w (M, K) row-major; wt = 1: W(m, k) = w[k * M + m] (the input gradient of a (K, M) weight)
int conv1x1_wg(oss_dtype io, const void *x, const float *w, const float *bias, void *y, int B, int M, int K, int P, int64_t xsb,
               int64_t xsk, int wt, hipStream_t s, const void *res) {
    if (!conv1x1_wg_ok(io, M, K, P, xsb, xsk, x, y, w, res)) return OSS_ERR_SHAPE;
    if (io == OSS_BF16)
        return wt ? wg_launch<bf16_t, true>(reinterpret_cast<const bf16_t *>(x), w, bias, reinterpret_cast<bf16_t *>(y), B, M, K, P, xsb, xsk,
                                            reinterpret_cast<const bf16_t *>(res), s)
                  : wg_launch<bf16_t, false>(reinterpret_cast<const bf16_t *>(x), w, bias, reinterpret_cast<bf16_t *>(y), B, M, K, P, xsb, xsk,
                                             reinterpret_cast<const bf16_t *>(res), s);
    return wt ? wg_launch<f16_t, true>(reinterpret_cast<const f16_t *>(x), w, bias, reinterpret_cast<f16_t *>(y), B, M, K, P, xsb, xsk,
                                       reinterpret_cast<const f16_t *>(res), s)
              : wg_launch<f16_t, false>(reinterpret_cast<const f16_t *>(x), w, bias, reinterpret_cast<f16_t *>(y), B, M, K, P, xsb, xsk,
                                        reinterpret_cast<const f16_t *>(res), s);
}

// ---- input gradient of a 1x1 convolution + the backward of the LayerNorm in front of it, one launch ---------------------------
// (in_conv after norm1: dn = W^T dy never leaves the CU.)  M = dim <= 128 output rows, so after the MFMA loop the workgroup holds
// dn for ALL channels of its pixels in LDS (the output staging tiles, rounded to T as the separate kernels hand it over); the
// LayerNorm backward of oss_layernorm.hip is a per-pixel function of (dn, x, mean, rstd) over the channels plus the skip
// connection's gradient:
//   s1 = sum_c g w,  s2 = sum_c g w xh;   dx = rstd (g w - s1 / C - xh s2 / C) + skip     (WithBias; BiasFree as in the LN kernel)
// and per-channel partial sums of g xh (d weight) and g (d bias) over the workgroup's pixels, finished in a fixed order later.
template <typename T>
struct WgLnBwdArgs {
    const T *x;             // (B, M, P) LayerNorm input
    const float *w;         // LayerNorm weight
    const float *mean, *rstd;
    const T *skip;          // (B, M, P) gradient arriving over the skip connection, or NULL
    T *dx;                  // (B, M, P)
    float *part;            // [workgroup][2 M]: partial d weight, d bias
    int with_bias;
};

// The LayerNorm-backward stage shared by the two kernels below: dn for all M channels of the workgroup's PT pixels in `os`
// ([4 tiles][32][PITCH]), the LayerNorm input tile in `xt` and the skip gradient in `st` ([M][PITCH] each), `red` = [2][NPART][PT] floats.
template <typename T, int PT>
__device__ __forceinline__ void wg_lnbwd_stage(const WgLnBwdArgs<T> &a, const T *os, const T *xt, const T *st, float *red, int M, int P,
                                               int b, int p0, int tid, int lane) {
    constexpr int PITCH = PT + 8;
    // the LayerNorm backward: a thread owns four adjacent pixels and the channels part, part + NPART, ...
    constexpr int QPT = PT / 4, NPART = 256 / QPT, CMAX = (128 + NPART - 1) / NPART;
    const int quad = tid % QPT, part = tid / QPT, px = 4 * quad;
    const bool with_bias = a.with_bias != 0, has_skip = a.skip != nullptr;
    float mu[4], rs[4];
    {
        const f32x4 m4 = *reinterpret_cast<const f32x4 *>(a.mean + (size_t)b * P + p0 + px);
        const f32x4 r4 = *reinterpret_cast<const f32x4 *>(a.rstd + (size_t)b * P + p0 + px);
        mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w;
        rs[0] = r4.x; rs[1] = r4.y; rs[2] = r4.z; rs[3] = r4.w;
    }
    float gv[CMAX][4], xv[CMAX][4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    // (round 4) the LayerNorm weights of this thread's channels, all requested here (clamped index, no branch).  Read where they are
    // used -- `if (c < M) { wc = a.w[c]; ...` in both loops below -- each was a branch + load + s_waitcnt vmcnt(0): 2 x 16 dependent
    // round trips in a kernel that runs one or two workgroups per CU.
    float wcs[CMAX];
#pragma unroll
    for (int i = 0; i < CMAX; ++i) wcs[i] = a.w[min(part + i * NPART, M - 1)];
    float *pw = a.part + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 * M;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) {
        const int c = part + i * NPART;
        float aw = 0.f, ab = 0.f;
        if (c < M) {
            const u32x2 gq = *reinterpret_cast<const u32x2 *>(os + ((c >> 5) * 32 + (c & 31)) * PITCH + px);
            const u32x2 xq2 = *reinterpret_cast<const u32x2 *>(xt + c * PITCH + px);
            unpack2<T>(gq.x, gv[i][0], gv[i][1]); unpack2<T>(gq.y, gv[i][2], gv[i][3]);
            unpack2<T>(xq2.x, xv[i][0], xv[i][1]); unpack2<T>(xq2.y, xv[i][2], xv[i][3]);
            const float wc = wcs[i];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float xh = with_bias ? (xv[i][u] - mu[u]) * rs[u] : xv[i][u] * rs[u];
                const float gg = gv[i][u];
                aw = __builtin_fmaf(gg, xh, aw);
                ab += gg;
                const float gw = gg * wc;
                s1[u] += gw;
                s2[u] = __builtin_fmaf(gw, with_bias ? xh : xv[i][u], s2[u]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) { gv[i][u] = 0.f; xv[i][u] = 0.f; }
        }
        // d weight / d bias partials of channel c over the workgroup's pixels: the 32 or 16 quads of this part are adjacent lanes
        const float tw = segment_sum_to_last<QPT>(aw), tb = segment_sum_to_last<QPT>(ab);
        if ((lane & (QPT - 1)) == QPT - 1 && c < M) { pw[c] = tw; pw[M + c] = tb; }
    }
    *reinterpret_cast<f32x4 *>(red + part * PT + px) = f32x4{s1[0], s1[1], s1[2], s1[3]};
    *reinterpret_cast<f32x4 *>(red + (NPART + part) * PT + px) = f32x4{s2[0], s2[1], s2[2], s2[3]};
    __syncthreads();
    float m1[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NPART; ++q) {
        const f32x4 t1 = *reinterpret_cast<const f32x4 *>(red + q * PT + px), t2 = *reinterpret_cast<const f32x4 *>(red + (NPART + q) * PT + px);
        m1[0] += t1.x; m1[1] += t1.y; m1[2] += t1.z; m1[3] += t1.w;
        m2[0] += t2.x; m2[1] += t2.y; m2[2] += t2.z; m2[3] += t2.w;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { m1[u] /= (float)M; m2[u] /= (float)M; }
    T *dxb = a.dx + (size_t)b * M * P + p0;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) {
        const int c = part + i * NPART;
        if (c < M) {
            const float wc = wcs[i];
            float sk[4] = {0.f, 0.f, 0.f, 0.f};
            if (has_skip) {
                const u32x2 s2q = *reinterpret_cast<const u32x2 *>(st + c * PITCH + px);
                unpack2<T>(s2q.x, sk[0], sk[1]); unpack2<T>(s2q.y, sk[2], sk[3]);
            }
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float xh = (xv[i][u] - mu[u]) * rs[u];
                const float gw = gv[i][u] * wc;
                d[u] = (with_bias ? rs[u] * (gw - m1[u] - xh * m2[u]) : rs[u] * gw - xh * rs[u] * rs[u] * m2[u]) + sk[u];
            }
            *reinterpret_cast<u32x2 *>(dxb + (size_t)c * P + px) = u32x2{pack2<T>(d[0], d[1]), pack2<T>(d[2], d[3])};
        }
    }
}

template <typename T, int KS, int PT>
__global__ void __launch_bounds__(256)
oss_conv1x1_dgrad_lnbwd_kernel(const T *__restrict__ dy, const float *__restrict__ w, int M, int P, int64_t xsb, int64_t xsk,
                               WgLnBwdArgs<T> a) {
    constexpr int K = 16 * KS, PITCH = PT + 8, NCT = PT / 32, CPR = PT / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
    T *xs = reinterpret_cast<T *>(wg_smem);                    // [K][PITCH] the dy tile; later x tile | skip tile (2 M <= K rows)
    T *os = xs + K * PITCH;                                    // [4 waves][32][PITCH]: dn, all row tiles
    float *red = reinterpret_cast<float *>(os + 4 * 32 * PITCH);   // [2][8][PT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, p0 = blockIdx.x * PT;
    const T *xb = dy + b * xsb + p0;
    const int col = lane & 31, kg = lane >> 5;
    const int mt_total = (M + 31) >> 5;      // <= 4: one row tile per wave
    // this wave's weight fragments (input gradient: W(m, k) = w[k * M + m])
    f32x4 wlo[KS], whi[KS];
    {
        const int mrow = min(wave * 32 + col, M - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = ks * 16 + kg * 8;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wlo[ks][e] = w[(size_t)(k0 + e) * M + mrow];
                whi[ks][e] = w[(size_t)(k0 + 4 + e) * M + mrow];
            }
        }
    }
    copy_tile_to_lds<T, K, PT, PITCH>(xb, xsk, xs, tid);
    // the LayerNorm input and the skip gradient of the tile: requested now, parked in LDS after the MFMAs
    constexpr int NLD = (128 * CPR + 255) / 256;   // 16-byte chunks per thread for <= 128 rows
    u32x4 xq[NLD], sq[NLD];
    const T *xin = a.x + (size_t)b * M * P + p0, *sin = a.skip ? a.skip + (size_t)b * M * P + p0 : xin;
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int idx = min(q * 256 + tid, M * CPR - 1), c = idx / CPR, pc = idx - c * CPR;
        xq[q] = *reinterpret_cast<const u32x4 *>(xin + (size_t)c * P + 8 * pc);
        sq[q] = *reinterpret_cast<const u32x4 *>(sin + (size_t)c * P + 8 * pc);
    }
    __syncthreads();
    const int i16 = lane & 15, g = lane >> 4;
    const int tr_off = (8 * (g >> 1) + (i16 >> 2)) * PITCH + 16 * (g & 1) + 4 * (i16 & 3);
    if (wave < mt_total) {
        const int m0 = wave * 32;
        f32x16 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const s16x8 af = m0 + col < M ? cvt8<T>(wlo[ks], whi[ks]) : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const T *bp = xs + ks * 16 * PITCH + ct * 32 + tr_off;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(bp));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(bp + 4 * PITCH));
                const s16x8 bf = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                acc[ct] = Mfma<T>::run(af, bf, acc[ct]);
            }
        }
        T *ow = os + wave * 32 * PITCH;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) ow[row * PITCH + ct * 32 + col] = from_f32<T>(acc[ct][r]);
        }
    }
    __syncthreads();   // dn complete in os; the dy tile in xs is no longer needed
    T *xt = xs, *st = xs + M * PITCH;   // (K >= 2 M rows: checked by the host)
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int idx = q * 256 + tid;
        if (idx < M * CPR) {
            const int c = idx / CPR, pc = idx - c * CPR;
            *reinterpret_cast<u32x4 *>(xt + c * PITCH + 8 * pc) = xq[q];
            *reinterpret_cast<u32x4 *>(st + c * PITCH + 8 * pc) = sq[q];
        }
    }
    __syncthreads();
    wg_lnbwd_stage<T, PT>(a, os, xt, st, red, M, P, b, p0, tid, lane);
}

__global__ void __launch_bounds__(256)
oss_wg_lnbwd_finish(const float *__restrict__ part, float *__restrict__ dw, float *__restrict__ db, int nblk, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * C) return;
    float sum = 0.f;
    for (int k = 0; k < nblk; ++k) sum += part[(size_t)k * 2 * C + i];
    if (i < C) dw[i] = sum;
    else if (db) db[i - C] = sum;
}

size_t conv1x1_dgrad_lnbwd_partial_floats(int B, int M, int P) { return (size_t)B * (P / 64) * 2 * M; }

int conv1x1_dgrad_lnbwd_ok(oss_dtype io, int M, int K, int P, int B) {
    if (io != OSS_BF16 && io != OSS_F16) return 0;
    if (K % 16 != 0 || K < 2 * M || K > 192 || M < 1 || M > 128 || P % 128 != 0) return 0;
    const int ks = K / 16;
    // (An any-K form for project_in after norm2 -- K = 2 hidden = 510, 64-pixel workgroups, weights fetched in chunks of 8 k-steps --
    // was built, parity-green, and measured slower inside the step: 219.2 against 221.7 images/s with the two separate kernels.)
    return (ks == 2 || ks == 3 || ks == 4 || ks == 6 || ks == 8 || ks == 12) ? 1 : 0;
}

template <typename T>
static int lnbwd_launch(const T *dy, const float *w, int B, int M, int K, int P, int64_t xsb, int64_t xsk, WgLnBwdArgs<T> a, float *dlw,
                        float *dlb, hipStream_t s) {
    const int pt = (long)B * (P / 128) >= 256 ? 128 : 64;
    const size_t smem = sizeof(uint16_t) * ((size_t)K + 4 * 32) * (pt + 8) + sizeof(float) * 2 * 8 * 128;
    dim3 grid(P / pt, B);
#define OSS_LNB(KS_, PT_)                                                                                   \
    do {                                                                                                    \
        static LdsGate gate;                                                                                \
        auto kern = oss_conv1x1_dgrad_lnbwd_kernel<T, KS_, PT_>;                                            \
        if (const int e = gate.ensure(reinterpret_cast<const void *>(kern), smem)) return e;                \
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, dy, w, M, P, xsb, xsk, a);                       \
    } while (0)
#define OSS_LNB2(KS_) do { if (pt == 128) OSS_LNB(KS_, 128); else OSS_LNB(KS_, 64); } while (0)
    switch (K / 16) {
        case 2: OSS_LNB2(2); break;
        case 3: OSS_LNB2(3); break;
        case 4: OSS_LNB2(4); break;
        case 6: OSS_LNB2(6); break;
        case 8: OSS_LNB2(8); break;
        case 12: OSS_LNB2(12); break;
        default: return OSS_ERR_SHAPE;
    }
#undef OSS_LNB2
#undef OSS_LNB
    const int nblk = (int)(grid.x * grid.y);
    if (defer_finish())
        defer_sum(a.part, nblk, (size_t)2 * M, (size_t)(dlb ? 2 : 1) * M, dlw, (size_t)M, dlb);
    else
        hipLaunchKernelGGL(oss_wg_lnbwd_finish, dim3((2 * M + 255) / 256), dim3(256), 0, s, a.part, dlw, dlb, nblk, M);
    return (int)hipGetLastError();
}

// dx = LayerNorm_backward(W^T dy; x, mean, rstd) + skip, d ln weight / bias; dy (B, K, P), w (K, M) row-major
int conv1x1_dgrad_lnbwd(oss_dtype io, const void *dy, const float *w, const void *x, const float *ln_w, int with_bias, const float *mean,
                        const float *rstd, const void *skip, void *dx, float *dlw, float *dlb, float *part, int B, int M, int K, int P,
                        int64_t xsb, int64_t xsk, hipStream_t s) {
    if (!conv1x1_dgrad_lnbwd_ok(io, M, K, P, B) || xsb % 8 != 0 || xsk % 8 != 0) return OSS_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(skip) | reinterpret_cast<uintptr_t>(dx) |
         reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd)) & 15u)
        return OSS_ERR_SHAPE;
    if (io == OSS_BF16) {
        WgLnBwdArgs<bf16_t> a{reinterpret_cast<const bf16_t *>(x), ln_w, mean, rstd, reinterpret_cast<const bf16_t *>(skip),
                              reinterpret_cast<bf16_t *>(dx), part, with_bias};
        return lnbwd_launch<bf16_t>(reinterpret_cast<const bf16_t *>(dy), w, B, M, K, P, xsb, xsk, a, dlw, dlb, s);
    }
    WgLnBwdArgs<f16_t> a{reinterpret_cast<const f16_t *>(x), ln_w, mean, rstd, reinterpret_cast<const f16_t *>(skip),
                         reinterpret_cast<f16_t *>(dx), part, with_bias};
    return lnbwd_launch<f16_t>(reinterpret_cast<const f16_t *>(dy), w, B, M, K, P, xsb, xsk, a, dlw, dlb, s);
}

// n = LayerNorm(x) (ln_w, ln_b or NULL), mean, rstd written out; y = W n + bias.  Same shape rules as conv1x1_wg (forward only).
int ln_conv1x1_wg(oss_dtype io, const void *x, const float *ln_w, const float *ln_b, float eps, void *n, float *mean, float *rstd,
                  const float *w, const float *bias, void *y, int B, int M, int K, int P, int64_t xsb, int64_t xsk, hipStream_t s) {
    if (!conv1x1_wg_ok(io, M, K, P, xsb, xsk, x, y, w, nullptr) || (reinterpret_cast<uintptr_t>(n) & 15u)) return OSS_ERR_SHAPE;
    if (io == OSS_BF16) {
        WgLnArgs<bf16_t> a{ln_w, ln_b, reinterpret_cast<bf16_t *>(n), mean, rstd, eps};
        return wg_launch<bf16_t, false, true>(reinterpret_cast<const bf16_t *>(x), w, bias, reinterpret_cast<bf16_t *>(y), B, M, K, P, xsb,
                                              xsk, nullptr, s, a);
    }
    WgLnArgs<f16_t> a{ln_w, ln_b, reinterpret_cast<f16_t *>(n), mean, rstd, eps};
    return wg_launch<f16_t, false, true>(reinterpret_cast<const f16_t *>(x), w, bias, reinterpret_cast<f16_t *>(y), B, M, K, P, xsb, xsk,
                                         nullptr, s, a);
}

}  // namespace oss
