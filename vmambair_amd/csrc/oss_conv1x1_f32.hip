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
// Out-of-range operands are loaded from a clamped (valid) address and zeroed with a bit mask: a `cond ? load : 0` makes hipcc branch
// around the load and end the branch with s_waitcnt vmcnt(0) -- every load its own memory round trip (checked in the ISA:
// `L j L w0` per k-step; DESIGN.md 4.4 rule 6).
// ... and masking the loaded value right away would make the round's MFMAs wait for the NEXT round's loads; instead an operand
// that must read as zero is loaded FROM a zero quad: the predicate goes into the address (two selects), nothing touches the value.
// (The select is done on integers and the result is read through an explicit global-address-space pointer: selecting between a
// kernel-argument pointer and the address of a __device__ variable as C++ pointers yields a FLAT pointer, flat loads count in
// both wait counters, and the compiler then waits vmcnt(0) before every MFMA round.)
__device__ const float g_zero_quad[4] __attribute__((aligned(16))) = {0.f, 0.f, 0.f, 0.f};
__device__ const float g_one_quad[4] __attribute__((aligned(16))) = {1.f, 1.f, 1.f, 1.f};   // the virtual all-ones row of the bias column
typedef const f32x4 __attribute__((address_space(1))) *GlobalQuadPtr;
typedef const float __attribute__((address_space(1))) *GlobalFloatPtr;
__device__ __forceinline__ uintptr_t addr_or_zero(const float *p, bool ok) {
    return ok ? reinterpret_cast<uintptr_t>(p) : reinterpret_cast<uintptr_t>(&g_zero_quad[0]);
}
__device__ __forceinline__ f32x4 quad_or_zero(const float *p, bool ok) { return *reinterpret_cast<GlobalQuadPtr>(addr_or_zero(p, ok)); }
// ... or four ones (`one`, when ok): the address is chosen before the load, nothing is selected after it
__device__ __forceinline__ f32x4 quad_one_or_zero(const float *p, bool ok, bool one) {
    const uintptr_t a = ok ? (one ? reinterpret_cast<uintptr_t>(&g_one_quad[0]) : reinterpret_cast<uintptr_t>(p)) : reinterpret_cast<uintptr_t>(&g_zero_quad[0]);
    return *reinterpret_cast<GlobalQuadPtr>(a);
}
__device__ __forceinline__ float float_or_zero(const float *p, bool ok) { return *reinterpret_cast<GlobalFloatPtr>(addr_or_zero(p, ok)); }

// ---- forward / input gradient -------------------------------------------------------------------------------------------------
// Y[b][g][m][p] = sum_k W[g](m, k) X[b][g % GX][k][p] (+ bias[m]) (+ res[b][g][m][p]) (+ the old Y: accumulate)
//   W[g](m, k) = w[g * wsg + m * wsm + k * wsk];  X row k of (b, g % GX): x + b * xsb + (g % GX) * xsg + k * xsk;  likewise y, res.
// G = 1 is the 1x1 convolution (transposed weight strides: its input gradient); G = 4, GX = 2 the x_proj product of the four scan
// directions (directions k and k + 2 read the same flattening, MambaSISR6_arch.py:401-408), G = GX = 4 the dt_proj product.
// One wave = (32 MT) rows x 128 pixels; grid (ceil(P / 128), ceil(M / (32 MT)), B * G), 64 threads.  P % 4 == 0, 16-byte aligned rows.
struct F32Gemm {
    const float *x, *w, *bias, *res;
    float *y;
    int M, K, P, G, GX, accumulate;
    int64_t xsb, xsg, xsk, wsg, wsm, wsk, ysb, ysg, ysm, rsb, rsg, rsm;
};
// SPLIT: a call with few tiles (a 1x1 convolution at batch 8 is 768 tiles for 1024 SIMDs) leaves every wave alone on its SIMD with
// nothing to cover its loads -- measured 57 us per launch against ~5 us of matrix time (profiles/r04_ab_fp32_kernels.txt).  The
// four waves of a 256-thread workgroup then take a quarter of K each (four times the waves, a quarter of the dependent
// load -> MFMA rounds per wave) and wave 0 adds the four accumulator sets through LDS in wave order (deterministic).
template <int MT, bool SPLIT>
__global__ void __launch_bounds__(SPLIT ? 256 : 64, 2)   // <= 256 registers (accumulators included)
oss_conv1x1_f32_kernel(const F32Gemm a_) {
    const F32Gemm &a = a_;
    const int lane = threadIdx.x & 63, col = lane & 31, kg = lane >> 5;
    const int wave = SPLIT ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
    const int p0 = blockIdx.x * 128 + 4 * col;
    const int m0 = blockIdx.y * 32 * MT;
    const int b = blockIdx.z / a.G, g = blockIdx.z - b * a.G;
    const int M = a.M;
    // this wave's channels [kbeg, K): an even number per wave, so that every k-step of two belongs to one wave
    const int kq = SPLIT ? (((a.K + 3) / 4 + 1) & ~1) : a.K;
    const int kbeg = SPLIT ? wave * kq : 0, K = SPLIT ? min(a.K, kbeg + kq) : a.K;
    const bool pok = p0 < a.P;
    const float *xb = a.x + b * a.xsb + (g % a.GX) * a.xsg + (pok ? p0 : 0);
    const float *wb = a.w + g * a.wsg;
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
        wr[t] = wb + (int64_t)(mok[t] ? m : 0) * a.wsm;
    }
    // Rounds of U k-steps: the loads of round r + 1 are issued BEFORE the MFMAs of round r (two register sets), so that a wave
    // alone on its SIMD still keeps the matrix pipe busy across the ~1.5 us a load round trip takes under load (the first version
    // issued a round's loads, waited, then ran its MFMAs: 57 us per 1x1 convolution against ~15 us of matrix time).
    constexpr int U = MT == 1 ? 4 : 2;
    auto load_round = [&](int k0, f32x4 (&xv)[U], float (&av)[U][MT]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 2 * u + kg;
            const bool kok = k < K;
            const int kc = kok ? k : 0;
            xv[u] = quad_or_zero(xb + (int64_t)kc * a.xsk, kok && pok);
#pragma unroll
            for (int t = 0; t < MT; ++t) av[u][t] = float_or_zero(wr[t] + (int64_t)kc * a.wsk, kok && mok[t]);
        }
    };
    auto mfma_round = [&](const f32x4 (&xv)[U], const float (&av)[U][MT]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[t][q] = mfma_f32(av[u][t], xv[u][q], acc[t][q]);
    };
    f32x4 xa[U], xc[U];
    float aa[U][MT], ac[U][MT];
    // The prefetches are UNCONDITIONAL (a round past K reads the zero quad): behind a branch the compiler cannot count on them
    // being in flight and falls back to waiting for (nearly) everything before the round's first MFMA (checked in the ISA).
    load_round(kbeg, xa, aa);
    for (int k0 = kbeg; k0 < K; k0 += 4 * U) {
        load_round(k0 + 2 * U, xc, ac);
        __builtin_amdgcn_sched_barrier(0);
        mfma_round(xa, aa);
        __builtin_amdgcn_sched_barrier(0);
        load_round(k0 + 4 * U, xa, aa);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 2 * U < K) mfma_round(xc, ac);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (SPLIT) {
        extern __shared__ __attribute__((aligned(16))) float red[];   // [3 waves][MT * 16 quads][64 lanes][4]
        if (wave > 0) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *reinterpret_cast<f32x4 *>(red + ((size_t)((wave - 1) * MT * 16 + t * 16 + r) * 64 + lane) * 4) =
                        f32x4{acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll 1   // one wave's set at a time: the compiler otherwise hoists all 192 quads into registers (spills)
        for (int w2 = 0; w2 < 3; ++w2)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(red + ((size_t)(w2 * MT * 16 + t * 16 + r) * 64 + lane) * 4);
                    acc[t][0][r] += v.x; acc[t][1][r] += v.y; acc[t][2][r] += v.z; acc[t][3][r] += v.w;
                }
    }
    if (!pok) return;
    float *yb = a.y + b * a.ysb + g * a.ysg + p0;
    const float *rb = a.res ? a.res + b * a.rsb + g * a.rsg + p0 : nullptr;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        // what the epilogue adds (skip connection, the previous pass of an accumulating call) is fetched as ONE group of loads per
        // row tile -- a load per row behind its own branch was sixteen serial round trips
        f32x4 add[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) add[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (each group: all sixteen loads first, into registers of their own, then the sums -- written as one loop the compiler kept
        // load r, s_waitcnt vmcnt(0), use r, load r + 1, ... : the ISA had two runs of sixteen `global_load ; s_waitcnt vmcnt(0)` pairs)
        if (a.bias) {
            float bvs[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kg;
                bvs[r] = float_or_zero(a.bias + m, m < M);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) add[r] = f32x4{bvs[r], bvs[r], bvs[r], bvs[r]};
        }
        if (rb) {   // eight rows at a time: sixteen 16-byte temporaries next to MT = 2's accumulators do not fit the register file
#pragma unroll
            for (int h = 0; h < 16; h += 8) {
                f32x4 rvs[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int m = m0 + 32 * t + ((h + r) & 3) + 8 * ((h + r) >> 2) + 4 * kg;
                    rvs[r] = quad_or_zero(rb + (int64_t)m * a.rsm, m < M);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 8; ++r) { add[h + r].x += rvs[r].x; add[h + r].y += rvs[r].y; add[h + r].z += rvs[r].z; add[h + r].w += rvs[r].w; }
            }
        }
        if (a.accumulate) {
#pragma unroll
            for (int h = 0; h < 16; h += 8) {
                f32x4 ovs[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int m = m0 + 32 * t + ((h + r) & 3) + 8 * ((h + r) >> 2) + 4 * kg;
                    ovs[r] = quad_or_zero(yb + (int64_t)m * a.ysm, m < M);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 8; ++r) { add[h + r].x += ovs[r].x; add[h + r].y += ovs[r].y; add[h + r].z += ovs[r].z; add[h + r].w += ovs[r].w; }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kg;
            const f32x4 o = {acc[t][0][r] + add[r].x, acc[t][1][r] + add[r].y, acc[t][2][r] + add[r].z, acc[t][3][r] + add[r].w};
            if (m < M) *reinterpret_cast<f32x4 *>(yb + (int64_t)m * a.ysm) = o;
        }
    }
}

static bool f32_aligned(std::initializer_list<const void *> ptrs, std::initializer_list<int64_t> strides) {
    for (const void *p : ptrs)
        if (reinterpret_cast<uintptr_t>(p) & 15u) return false;
    for (int64_t st : strides)
        if (st % 4 != 0) return false;
    return true;
}

int gemm_f32(const F32Gemm &a, int B, hipStream_t s) {
    if (a.M < 1 || a.K < 1 || a.P < 4 || a.P % 4 != 0 || a.G < 1 || a.GX < 1) return OSS_ERR_SHAPE;
    if (!f32_aligned({a.x, a.y, a.res}, {a.xsb, a.xsg, a.xsk, a.ysb, a.ysg, a.ysm, a.rsb, a.rsg, a.rsm})) return OSS_ERR_SHAPE;
    if (B <= 0 || (long)B * a.G > 65535) return OSS_ERR_SHAPE;
    const long px = (a.P + 127) / 128, bg = (long)B * a.G;
    const long waves64 = px * ((a.M + 63) / 64) * bg, waves32 = px * ((a.M + 31) / 32) * bg;
    if (waves64 >= 3072 && a.M > 32) {          // plenty of tiles: 64-row tiles, one wave each
        const dim3 grid(px, (a.M + 63) / 64, bg);
        hipLaunchKernelGGL((oss_conv1x1_f32_kernel<2, false>), grid, dim3(64), 0, s, a);
    } else if (waves32 >= 3072 || a.K < 32) {   // 32-row tiles, one wave each (a short K has nothing to split)
        const dim3 grid(px, (a.M + 31) / 32, bg);
        hipLaunchKernelGGL((oss_conv1x1_f32_kernel<1, false>), grid, dim3(64), 0, s, a);
    } else {                                    // few tiles: four waves per tile, a quarter of K each
        const dim3 grid(px, (a.M + 31) / 32, bg);
        static LdsGate gate;
        const size_t smem = sizeof(float) * 3 * 16 * 64 * 4;
        auto kern = oss_conv1x1_f32_kernel<1, true>;
        if (const int e = gate.ensure(reinterpret_cast<const void *>(kern), smem)) return e;
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, a);
    }
    return (int)hipGetLastError();
}

int conv1x1_f32(const float *x, const float *w, const float *bias, float *y, int B, int M, int K, int P, int64_t xsb, int64_t xsk,
                int64_t wsm, int64_t wsk, hipStream_t s, const float *res) {
    F32Gemm a{};
    a.x = x; a.w = w; a.bias = bias; a.res = res; a.y = y;
    a.M = M; a.K = K; a.P = P; a.G = 1; a.GX = 1; a.accumulate = 0;
    a.xsb = xsb; a.xsg = 0; a.xsk = xsk; a.wsg = 0; a.wsm = wsm; a.wsk = wsk;
    a.ysb = (int64_t)M * P; a.ysg = 0; a.ysm = P; a.rsb = a.ysb; a.rsg = 0; a.rsm = P;
    return gemm_f32(a, B, s);
}

// x_proj / dt_proj of the four scan directions at fp32 I/O (MambaSISR6_arch.py:406-411) -- the products oss_proj_fwd / oss_proj_dgrad
// run on the vector ALU for float tensors in rounds 1-3 (7.8 ms of the fp32 step, profiles/r04_rocprof_bench_fp32_steady_state.txt)
int proj_f32_ok(int B, int D, int C, int R, int L, std::initializer_list<const void *> ptrs) {
    if (L < 4 || L % 4 != 0 || B <= 0 || 4L * B > 65535 || D < 1 || R < 1 || C <= R) return 0;
    for (const void *p : ptrs)
        if (reinterpret_cast<uintptr_t>(p) & 15u) return 0;
    return 1;
}
// xdbl[b, k] (C rows) = Wx[k] (C x D) x2[b, k % 2] (D rows);  dts[b, k] (D rows) = Wdt[k] (D x R) xdbl[b, k][:R]
int proj_fwd_f32(const float *x2, const float *Wx, const float *Wdt, float *xdbl, float *dts, int B, int D, int C, int R, int L,
                 hipStream_t s) {
    const int64_t l = L;
    F32Gemm a{};
    a.x = x2; a.w = Wx; a.y = xdbl; a.M = C; a.K = D; a.P = L; a.G = 4; a.GX = 2;
    a.xsb = 2 * D * l; a.xsg = D * l; a.xsk = l; a.wsg = (int64_t)C * D; a.wsm = D; a.wsk = 1;
    a.ysb = 4 * C * l; a.ysg = C * l; a.ysm = l;
    int e = gemm_f32(a, B, s);
    if (e || !dts) return e;
    F32Gemm d{};
    d.x = xdbl; d.w = Wdt; d.y = dts; d.M = D; d.K = R; d.P = L; d.G = 4; d.GX = 4;
    d.xsb = 4 * C * l; d.xsg = C * l; d.xsk = l; d.wsg = (int64_t)D * R; d.wsm = R; d.wsk = 1;
    d.ysb = 4 * D * l; d.ysg = D * l; d.ysm = l;
    return gemm_f32(d, B, s);
}
// dxdbl[b, k][:R] = Wdt[k]^T ddts[b, k];  dx2[b, j] = Wx[j]^T dxdbl[b, j] + Wx[j + 2]^T dxdbl[b, j + 2] (+ du[b, j] + du[b, j + 2])
int proj_dgrad_f32(const float *ddts, float *dxdbl, const float *du, const float *Wx, const float *Wdt, float *dx2, int B, int D, int C,
                   int R, int L, hipStream_t s) {
    const int64_t l = L;
    F32Gemm t{};
    t.x = ddts; t.w = Wdt; t.y = dxdbl; t.M = R; t.K = D; t.P = L; t.G = 4; t.GX = 4;
    t.xsb = 4 * D * l; t.xsg = D * l; t.xsk = l; t.wsg = (int64_t)D * R; t.wsm = 1; t.wsk = R;
    t.ysb = 4 * C * l; t.ysg = C * l; t.ysm = l;
    int e = gemm_f32(t, B, s);
    if (e) return e;
    for (int kk = 0; kk < 2; ++kk) {   // directions j (kk = 0) and j + 2 (kk = 1) of flattening j, the second pass accumulates
        F32Gemm a{};
        a.x = dxdbl + kk * 2 * C * l; a.w = Wx + kk * 2 * (int64_t)C * D; a.y = dx2;
        a.res = du ? du + kk * 2 * D * l : nullptr;
        a.M = D; a.K = C; a.P = L; a.G = 2; a.GX = 2; a.accumulate = kk;
        a.xsb = 4 * C * l; a.xsg = C * l; a.xsk = l; a.wsg = (int64_t)C * D; a.wsm = 1; a.wsk = D;
        a.ysb = 2 * D * l; a.ysg = D * l; a.ysm = l; a.rsb = 4 * D * l; a.rsg = D * l; a.rsm = l;
        e = gemm_f32(a, B, s);
        if (e) return e;
    }
    return 0;
}

// ---- weight gradient (and every other "rows x rows over pixels" product) ------------------------------------------------------
// part[(b * slabs + slab)][g][m][n] = sum over the slab's pixels of A[b][g][m][p] Bm[b][g % GB][n][p]
//   A row (b, g, m):  a + b * asb + g * asg + m * asm;   Bm row (b, g % GB, n):  bm + b * bsb + (g % GB) * bsg + n * bsn
// one wave = a (32 TM) x (32 TN) tile of one (b, g, slab); grid (slabs, B * G, ceil(tiles / 4)), 256 threads (4 tiles).
constexpr int kF32WgradSlab = 512;   // pixels per partial product
template <int TM, int TN>
__global__ void __launch_bounds__(256)
oss_rows_f32_wgrad_kernel(const float *__restrict__ a, const float *__restrict__ bm, float *__restrict__ part, int M, int N, int P, int G,
                          int GB, int64_t asb, int64_t asg, int64_t asm_, int64_t bsb, int64_t bsg, int64_t bsn,
                          int NB /* N, or N + 1 (G == 1): a virtual all-ones row of Bm whose column of the product is sum_p A = the bias
                          gradient of a 1x1 convolution -- it used to be a torch sum over dy of its own, 100 launches per fp32 step */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, kg = lane >> 5;
    const int b = blockIdx.y / G, g = blockIdx.y - b * G, slab = blockIdx.x;
    const int mt = (M + 32 * TM - 1) / (32 * TM), nt = (NB + 32 * TN - 1) / (32 * TN);
    const int tile = blockIdx.z * 4 + wave;
    if (tile >= mt * nt) return;
    const int m0 = (tile / nt) * 32 * TM, n0 = (tile % nt) * 32 * TN;
    const int pbeg = slab * kF32WgradSlab, pend = min(P, pbeg + kF32WgradSlab);
    const float *ab = a + b * asb + g * asg, *bb = bm + b * bsb + (g % GB) * bsg;
    const float *ar[TM], *br[TN];
    bool aok[TM], bok[TN], bone[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + 32 * i + col;
        aok[i] = m < M;
        ar[i] = ab + (int64_t)(aok[i] ? m : 0) * asm_;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + 32 * j + col;
        bone[j] = n == N && NB > N;
        bok[j] = n < N || bone[j];
        br[j] = bb + (int64_t)(n < N ? n : 0) * bsn;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // rounds of U 8-pixel steps, the next round's loads in flight during the current round's MFMAs (two register sets)
    constexpr int U = (TM + TN <= 2) ? 4 : 2;
    auto load_round = [&](int p, f32x4 (&av)[U][TM], f32x4 (&bv)[U][TN]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pp = p + 8 * u;
            const bool ok = pp + 4 * kg < pend;   // P % 4 == 0: a lane's quad is inside or outside as a whole
            const int pc = ok ? pp + 4 * kg : pbeg;   // (a quad past the end reads the slab's first one and is zeroed)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                av[u][i] = quad_or_zero(ar[i] + pc, ok && aok[i]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bv[u][j] = quad_one_or_zero(br[j] + pc, ok && bok[j], bone[j]);
            }
        }
    };
    auto mfma_round = [&](const f32x4 (&av)[U][TM], const f32x4 (&bv)[U][TN]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[i][j] = mfma_f32(av[u][i][q], bv[u][j][q], acc[i][j]);
    };
    f32x4 a0[U][TM], b0[U][TN], a1[U][TM], b1[U][TN];
    load_round(pbeg, a0, b0);   // (prefetches unconditional, rounds past the slab read the zero quad: see the GEMM kernel)
    for (int p = pbeg; p < pend; p += 16 * U) {
        load_round(p + 8 * U, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_round(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        load_round(p + 16 * U, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (p + 8 * U < pend) mfma_round(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    const size_t pvec = (size_t)G * M * N + (NB > N ? M : 0);   // one partial vector per (batch, slab): [G][M][N], then the M bias sums
    float *pb = part + (size_t)(b * gridDim.x + slab) * pvec + (size_t)g * M * N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + 32 * j + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (m < M && n < N) pb[(size_t)m * N + n] = acc[i][j][r];
                else if (m < M && n == N && NB > N) pb[(size_t)M * N + m] = acc[i][j][r];   // (G == 1)
            }
        }
}

__global__ void __launch_bounds__(256)
oss_rows_f32_wgrad_finish(const float *__restrict__ part, float *__restrict__ out, int K, size_t pvec, size_t nw, float *__restrict__ db) {
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
    if (i < nw) out[i] = s;
    else db[i - nw] = s;   // the bias column
}

// (round 4, measured and removed: these products as ONE grouped launch, as the 16-bit path runs them -- shared descriptor, 32 x 32 tiles.
// 2 launches of 3.6 ms = 7.2 ms per step against 7.3 ms for the 302 separate launches, 148.6 against 150.5 images/s: the operands are
// cold by then (11.8 GB held) and this kernel reads them 16 bytes per row and load instruction, 32 cache lines each -- the separate
// launches find them in the L2 / MALL of the pass that produced them.  A grouped form would need the LDS-staged operand tiles of the
// 16-bit kernel first.  profiles/r04_fp32_grouped_wgrad_no_gain.txt)
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
                   int64_t asg, int64_t asm_, int64_t bsb, int64_t bsg, int64_t bsn, hipStream_t s, float *db) {
    if (!rows_f32_wgrad_ok(M, N, P, a, bm, {asb, asg, asm_, bsb, bsg, bsn})) return OSS_ERR_SHAPE;
    if (B <= 0 || G <= 0 || GB <= 0 || (size_t)B * G > 65535 || (db && G != 1)) return OSS_ERR_SHAPE;
    const int NB = N + (db ? 1 : 0);
    const int slabs = rows_f32_wgrad_slabs(P);
    // 64 x 32 tiles; 32 x 32 when the wider tile would leave most SIMDs without a wave
    const int t21 = ((M + 63) / 64) * ((NB + 31) / 32), t11 = ((M + 31) / 32) * ((NB + 31) / 32);
    if ((long)t21 * slabs * B * G >= 1024 && M > 32) {
        const dim3 grid(slabs, B * G, (t21 + 3) / 4);
        hipLaunchKernelGGL((oss_rows_f32_wgrad_kernel<2, 1>), grid, dim3(256), 0, s, a, bm, part, M, N, P, G, GB, asb, asg, asm_, bsb, bsg, bsn, NB);
    } else {
        const dim3 grid(slabs, B * G, (t11 + 3) / 4);
        hipLaunchKernelGGL((oss_rows_f32_wgrad_kernel<1, 1>), grid, dim3(256), 0, s, a, bm, part, M, N, P, G, GB, asb, asg, asm_, bsb, bsg, bsn, NB);
    }
    const size_t nw = (size_t)G * M * N, pvec = nw + (db ? M : 0);
    if (defer_finish())
        defer_sum(part, slabs * B, pvec, pvec, out, nw, db);
    else
        hipLaunchKernelGGL(oss_rows_f32_wgrad_finish, dim3((unsigned)((pvec + 255) / 256)), dim3(256), 0, s, part, out, slabs * B, pvec, nw, db);
    return (int)hipGetLastError();
}

}  // namespace oss
