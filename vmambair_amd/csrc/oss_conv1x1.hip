// oss_conv1x1.hip -- the dense 1x1 projections of the OSS block on the CDNA4 matrix cores:
// in_conv / out_conv of SS2D_1 and project_in / project_out of the EFFN
// (SRGAN/VmambaIR/archs/MambaSISR6_arch.py:205,211,281,329), directly on NCHW tensors.
//
// These are the only GEMM-shaped ops of the block.  The vendor conv path spends 4 launches forward and
// ~8 backward per projection on NCHW<->NHWC transposes, weight casts and bias tensor-ops around one
// implicit-GEMM kernel (profiles/r01_rocprof_bench_v8_summary.txt), which is what bounds a 64x64
// training step.  Here: ONE launch forward (fp32 master weights converted to bf16 in the loader, bias
// fused), one for the input gradient (same kernel, weights read transposed), one split-K launch + a
// finishing launch for the weight gradient.
//
// MFMA: v_mfma_f32_32x32x16_{bf16,f16}, one 32 x 32 output tile per wave-instruction group, fp32
// accumulation.  Operand maps (cdna_hip_programming.md section 3): lane l holds A[i = l & 31][k = 8 (l >> 5) .. +8] and
// B[k = 8 (l >> 5) .. +8][j = l & 31]; D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31], r in [0,16).
//   forward / dgrad: rows = output channels, cols = 32 consecutive pixels, k = input channels.  The
//     activation operand needs 8 channels of ONE pixel per lane while NCHW stores pixels contiguously:
//     8 two-byte loads per lane, each coalesced over the 32 pixel lanes (64-byte segments).
//   wgrad: k = pixels, so both operands are 16-byte contiguous loads (dy rows and x rows).
#include <atomic>
#include <cstdlib>
#include "oss_device.h"
#include "oss_global_ptr.h"
#include "oss_host.h"
#include "oss_mfma.h"

namespace oss {

// y[b, m, p] = sum_k W(m, k) x[b, k, p] (+ bias[m]);  W(m, k) = w[m * ws_m + k * ws_k] (fp32 master weights).
//   forward: M = Cout, K = Cin, ws_m = Cin, ws_k = 1;  dgrad: M = Cin, K = Cout, ws_m = 1, ws_k = Cin.
// x: (B, K, P) with strides (xsb, xsk), pixels contiguous; y: (B, M, P) contiguous.
// One wave = one 32 (rows) x 32 (pixels) output tile over the full K: grid (pixel blocks of 4 waves,
// batch, row tiles), so that small images (64 pixels at the UNet's level 4) still spread over the chip.
template <typename T>
__global__ void __launch_bounds__(256)
oss_conv1x1_kernel(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias, T *__restrict__ y,
                   int M, int K, int P, int64_t xsb, int64_t xsk, int64_t ws_m, int64_t ws_k, const T *__restrict__ res) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int m0 = blockIdx.z * 32;
    const int p0 = (blockIdx.x * 4 + wave) * 32;
    if (p0 >= P) return;
    const int col = lane & 31, kg = lane >> 5;
    const int p = p0 + col;
    const bool pok = p < P;
    const T *xb = x + b * xsb + (pok ? p : 0);
    T *yb = y + (size_t)b * M * P;
    const int mrow = m0 + col;  // A-operand row of this lane
    const bool mok = mrow < M;
    const float *wrow = w + (mok ? mrow : 0) * ws_m;
    const bool wvec = (ws_k == 1) && ((reinterpret_cast<uintptr_t>(wrow) & 15u) == 0) && (K % 8 == 0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int ksteps = (K + 15) >> 4;

    for (int ks = 0; ks < ksteps; ++ks) {
        const int k0 = ks * 16 + kg * 8;
        s16x8 af, bf;
        if (wvec && k0 + 8 <= K) {
            const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wrow + k0);
            const f32x4 w1 = *reinterpret_cast<const f32x4 *>(wrow + k0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                af[e] = mok ? to_bits<T>(w0[e]) : (short)0;
                af[4 + e] = mok ? to_bits<T>(w1[e]) : (short)0;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) bf[e] = pok ? (short)xb[(k0 + e) * xsk].v : (short)0;
        } else {
            // clamped addresses + selects instead of predicated loads: no branches per element
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e;
                const bool kok = k < K;
                const int kc = kok ? k : K - 1;
                const float wv = wrow[kc * ws_k];
                const short xv = (short)xb[kc * xsk].v;
                af[e] = (mok && kok) ? to_bits<T>(wv) : (short)0;
                bf[e] = (pok && kok) ? xv : (short)0;
            }
        }
        acc = Mfma<T>::run(af, bf, acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        if (row < M && pok) {
            float v = acc[r] + (bias ? bias[row] : 0.f);
            if (res) v += to_f32(res[(size_t)b * M * P + (size_t)row * P + p]);  // fused residual: y = W x + bias + res
            yb[(size_t)row * P + p] = from_f32<T>(v);
        }
    }
}

// K <= 16 * KS form: the activation fragments of the wave's 32 pixels are loaded ONCE for the whole K
// into registers and reused for `mt_per_wave` row tiles (the 2-byte strided activation loads are the
// expensive part; the weights come out of L1/L2).  WT = false: W(m, k) = w[m * K + k], two 16-byte
// loads per k-step (needs K % 8 == 0); WT = true (input gradient): W(m, k) = w[k * M + m].
template <typename T, int KS, bool WT>
__global__ void __launch_bounds__(256)
oss_conv1x1_reuse_kernel(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                         T *__restrict__ y, int M, int K, int P, int64_t xsb, int xsk, int mt_per_wave, const T *__restrict__ res) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int p0 = (blockIdx.x * 4 + wave) * 32;
    if (p0 >= P) return;
    const int col = lane & 31, kg = lane >> 5;
    const int p = p0 + col;
    const bool pok = p < P;
    const T *xb = x + b * xsb + (pok ? p : 0);
    T *yb = y + (size_t)b * M * P;
    const int ksteps = (K + 15) >> 4;
    s16x8 bfr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ks * 16 + kg * 8 + e;
            const bool kok = k < K;
#ifdef OSS_EXP_CONV_NOLOAD  // (timing experiments only: tools/build_experiment.sh)
            const short xv = (short)(lane * 3 + k);
#else
            const short xv = (short)xb[(kok ? k : K - 1) * xsk].v;
#endif
            bfr[ks][e] = (pok && kok) ? xv : (short)0;
        }
    }
    const int mt_total = (M + 31) >> 5;
    const int mt_end = min(mt_total, (int)(blockIdx.z + 1) * mt_per_wave);
    for (int mt = blockIdx.z * mt_per_wave; mt < mt_end; ++mt) {
        const int m0 = mt * 32;
        const int mrow = m0 + col;
        const bool mok = mrow < M;
        const int mc = mok ? mrow : 0;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks < ksteps) {
                const int k0 = ks * 16 + kg * 8;
                s16x8 af;
                if constexpr (!WT) {
                    const bool kok = k0 + 8 <= K;
                    const float *wp = w + mc * K + (kok ? k0 : 0);
#ifdef OSS_EXP_CONV_NOW
                    const f32x4 w0 = {(float)lane, 1.f, 2.f, (float)mt}, w1 = {3.f, (float)ks, 4.f, 5.f};
#else
                    const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wp);
                    const f32x4 w1 = *reinterpret_cast<const f32x4 *>(wp + 4);
#endif
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        af[e] = (mok && kok) ? to_bits<T>(w0[e]) : (short)0;
                        af[4 + e] = (mok && kok) ? to_bits<T>(w1[e]) : (short)0;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = k0 + e;
                        const bool kok = k < K;
                        const float wv = w[(kok ? k : K - 1) * M + mc];
                        af[e] = (mok && kok) ? to_bits<T>(wv) : (short)0;
                    }
                }
                acc = Mfma<T>::run(af, bfr[ks], acc);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
#ifdef OSS_EXP_CONV_NOSTORE
            if (row < M && pok && acc[r] == 123456.789f) {
#else
            if (row < M && pok) {
#endif
                float v = acc[r] + (bias ? bias[row] : 0.f);
                if (res) v += to_f32(res[(size_t)b * M * P + (size_t)row * P + p]);  // fused residual: y = W x + bias + res
                yb[(size_t)row * P + p] = from_f32<T>(v);
            }
        }
    }
}

// Weight fragment of one MFMA k-step: the 8 consecutive k of row `mrow` starting at k0, narrowed to T; zero outside the
// (M x K) matrix.  Addresses are always valid, so the loads carry no condition.  WT: W(m, k) = w[k * M + m] (input gradient),
// else w[m * K + k]; WVEC: two 16-byte loads (K % 8 == 0, aligned).
template <typename T, bool WT, bool WVEC>
__device__ __forceinline__ s16x8 load_wfrag(const float *__restrict__ w, int mrow, int M, int K, int k0) {
    const bool mok = mrow < M;
    const int mc = mok ? mrow : 0;
    if constexpr (!WT && WVEC) {
        const bool kok = k0 + 8 <= K;
        const float *wp = w + mc * K + (kok ? k0 : 0);
#ifdef OSS_EXP_HALF_W   // TIMING ONLY (wrong results): half of the weight bytes -- what 16-bit weight copies would cost to load
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wp), w1 = w0;
#else
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wp), w1 = *reinterpret_cast<const f32x4 *>(wp + 4);
#endif
        const s16x8 f = cvt8<T>(w0, w1);
        return (mok && kok) ? f : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    } else {
        float wv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            const bool kok = k < K;
            const int kc = kok ? k : K - 1;
#ifdef OSS_EXP_HALF_W
            const int kh = kc & ~1;
            const float w1 = WT ? w[kh * M + mc] : w[mc * K + kh];
#else
            const float w1 = WT ? w[kc * M + mc] : w[mc * K + kc];
#endif
            wv[e] = (mok && kok) ? w1 : 0.f;
        }
        return cvt8<T>(f32x4{wv[0], wv[1], wv[2], wv[3]}, f32x4{wv[4], wv[5], wv[6], wv[7]});
    }
}

// Epilogue of the pixel-pair kernels: the 16 accumulator rows of a lane (+ bias, + residual) -> one 4-byte store per row.
// The bias values and the residual words of ALL rows are fetched first (clamped addresses, no per-row condition) and the
// stores follow back to back.  The first version loaded bias[row] and res[row] inside `if (row < M && pok)` row by row:
// hipcc then ends every row with s_waitcnt vmcnt(0), which on gfx9 also waits for the previous row's STORE -- 16
// serialised memory round trips, ~2/3 of the kernel's time (profiles/r02_conv1x1_epilogue.txt).
template <typename T>
__device__ __forceinline__ void pair_epilogue(const f32x16 &acca, const f32x16 &accb, const float *__restrict__ bias,
                                              const T *__restrict__ resp /* this lane's pixel pair, row 0; or NULL */,
                                              T *__restrict__ yp /* same for the output */, int m0, int kg, int M, int P, bool pok) {
    const int psw = P >> 1;
    float bv[16];
    uint32_t rw[16];
    if (bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = bias[min(m0 + (r & 3) + 8 * (r >> 2) + 4 * kg, M - 1)];
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = 0.f;
    }
    if (resp) {
        const uint32_t *rp = reinterpret_cast<const uint32_t *>(resp);
#pragma unroll
        for (int r = 0; r < 16; ++r) rw[r] = rp[(size_t)min(m0 + (r & 3) + 8 * (r >> 2) + 4 * kg, M - 1) * psw];
    }
    uint32_t *yw = reinterpret_cast<uint32_t *>(yp);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        float va = acca[r] + bv[r], vb = accb[r] + bv[r];
        if (resp) {
            float r0, r1;
            unpack2<T>(rw[r], r0, r1);
            va += r0;
            vb += r1;
        }
        if (row < M && pok) yw[(size_t)row * psw] = pack2<T>(va, vb);
    }
}

// Raw (fp32) weight fragments of one row tile, all KS k-steps: in flight as a group, narrowed when they are needed.
template <int KS>
struct WRaw { f32x4 lo[KS], hi[KS]; };

// issue the loads of row `mrow`'s fragments (addresses clamped into the matrix: no condition on any load)
template <int KS, bool WT, bool WVEC>
__device__ __forceinline__ void wraw_issue(WRaw<KS> &r, const float *__restrict__ w, int mrow, int M, int K, int kg) {
#ifdef OSS_EXP_CONV_NOW   // timing experiment: weight fragments without memory traffic
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { r.lo[ks] = f32x4{1.f, 2.f, (float)mrow, (float)kg}; r.hi[ks] = r.lo[ks]; }
    return;
#endif
    const int mc = mrow < M ? mrow : 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k0 = ks * 16 + kg * 8;
        if constexpr (!WT && WVEC) {
            const float *wp = w + mc * K + (k0 + 8 <= K ? k0 : 0);
            r.lo[ks] = *reinterpret_cast<const f32x4 *>(wp);
            r.hi[ks] = *reinterpret_cast<const f32x4 *>(wp + 4);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e, kc = k < K ? k : K - 1;
                const float v = WT ? w[kc * M + mc] : w[mc * K + kc];
                if (e < 4) r.lo[ks][e] = v; else r.hi[ks][e - 4] = v;
            }
        }
    }
}

// ... and narrow them to T, zero outside the (M x K) matrix
template <typename T, int KS, bool WT, bool WVEC>
__device__ __forceinline__ void wraw_narrow(const WRaw<KS> &r, s16x8 (&af)[KS], int mrow, int M, int K, int kg) {
    const bool mok = mrow < M;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k0 = ks * 16 + kg * 8;
        if constexpr (!WT && WVEC) {
            const s16x8 f = cvt8<T>(r.lo[ks], r.hi[ks]);
            af[ks] = (mok && k0 + 8 <= K) ? f : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        } else {
            f32x4 lo = r.lo[ks], hi = r.hi[ks];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo[e] = (mok && k0 + e < K) ? lo[e] : 0.f;
                hi[e] = (mok && k0 + 4 + e < K) ? hi[e] : 0.f;
            }
            af[ks] = cvt8<T>(lo, hi);
        }
    }
}

// Forward weights (W(m, k) = w[m * K + k], K % 8 == 0) through wave-private LDS.  The MFMA fragment of a lane is 8 consecutive
// k of ITS OWN row: loaded straight from memory, one 16-byte load instruction touches 32 rows = 64 cache lines and the
// texture addresser spends a cycle on each (the weight loads were 6 of the 18.6 us of in_conv at d = 96 -- experiment build
// CONV_NOW, profiles/r02_conv1x1_pipeline.txt).  A row tile is ONE contiguous block of 32 K floats: lane l fetches the 16-byte
// chunks l, l + 64, ... (fully coalesced), narrows them and parks them as [32][16 KS + 8] T (padded rows: conflict-free
// 16-byte reads); the fragments are read back from there.  No barrier: LDS operations of one wave execute in order.
template <int KS>
__device__ __forceinline__ void wflat_issue(WRaw<KS> &r, const float *__restrict__ w, int m0, int M, int K, int lane) {
    const int lim = min(M - m0, 32) * K - 4;          // last chunk inside the matrix
    const float *base = w + (size_t)m0 * K;
#pragma unroll
    for (int i = 0; i < 2 * KS; ++i) {
        const f32x4 q = *reinterpret_cast<const f32x4 *>(base + min(256 * i + 4 * lane, lim));
        if (i < KS) r.lo[i] = q; else r.hi[i - KS] = q;
    }
}
template <typename T, int KS>
__device__ __forceinline__ void wflat_frags(const WRaw<KS> &r, T *__restrict__ myl, s16x8 (&af)[KS], int m0, int M, int K,
                                            int col, int kg, int row0, int k0, int sr, int sk) {
    constexpr int RS = 16 * KS + 8;
    const int nl = K >> 3;            // chunks per lane: 32 K floats / 4 / 64
    int rr = row0, kk = k0;
#pragma unroll
    for (int i = 0; i < 2 * KS; ++i) {
        const f32x4 q = i < KS ? r.lo[i] : r.hi[i < KS ? 0 : i - KS];
        if (i < nl) *reinterpret_cast<u32x2 *>(myl + rr * RS + kk) = u32x2{pack2<T>(q.x, q.y), pack2<T>(q.z, q.w)};
        kk += sk;
        rr += sr;
        if (kk >= K) { kk -= K; ++rr; }
    }
    __builtin_amdgcn_wave_barrier();
    const bool mok = m0 + col < M;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const u32x4 f = *reinterpret_cast<const u32x4 *>(myl + col * RS + ks * 16 + kg * 8);
        af[ks] = (mok && ks * 16 + kg * 8 + 8 <= K) ? __builtin_bit_cast(s16x8, f) : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    __builtin_amdgcn_wave_barrier();
}

// Pixel-pair form of the kernel above (P even, 4-byte aligned rows): one wave = 64 pixels as TWO MFMA column
// tiles that interleave -- tile A holds the even pixels p0 + 2c, tile B the odd ones p0 + 2c + 1 (c = lane & 31).
// One 4-byte load per lane and channel feeds both tiles (128-byte row segments instead of 64), one weight
// fragment feeds two MFMAs, and the epilogue packs the two tiles' results into one 4-byte store per lane and
// row (the 2-byte stores of the single-tile form were ~40% of its time: profiles/r01_conv_ablation.txt).
// wvec: weights readable as two 16-byte loads per fragment (K % 8 == 0, aligned); else element-wise.
// Whole K in registers (K <= 16 KS); a wave produces `mt_per_wave` row tiles one after the other.
//
// Round 2: the kernel is a chain of memory round trips (~1 us each under load), not a stream, so the order of the loads is
// the design.  (1) A tile's operands -- weight fragments, one bias value per lane (redistributed with ds_bpermute), the 16
// residual words -- are issued as ONE group with clamped addresses and no per-load condition; (2) PF: the NEXT tile's group
// is issued before the current tile's MFMAs and stores, so a wave with several tiles pays the latency once; (3) the first
// group is issued before the activation loads.  The first version fetched weights k-step by k-step and bias / residual row
// by row, each behind its own s_waitcnt vmcnt(0) (which on gfx9 also waits for the previous row's store): 12 row tiles on
// one wave took 36 us (profiles/r02_conv1x1_pipeline.txt).
template <typename T, int KS, bool WT, bool WVEC, bool RES, bool PF>
__global__ void __launch_bounds__(256)
oss_conv1x1_pair_kernel(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                        T *__restrict__ y, int M, int K, int P, int64_t xsb, int xsk, int mt_per_wave, const T *__restrict__ res) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int p0 = (blockIdx.x * 4 + wave) * 64;
    if (p0 >= P) return;
    const int col = lane & 31, kg = lane >> 5;
    const int p = p0 + 2 * col;  // even pixel of this lane's pair
    const bool pok = p < P;      // P is even: the odd pixel is in range too
    const int pc = pok ? p : 0;
    const uint32_t *xw = reinterpret_cast<const uint32_t *>(x + b * xsb + pc);
    const int xsw = xsk >> 1;    // row stride in 4-byte words
    const int psw = P >> 1;
    uint32_t *yw = reinterpret_cast<uint32_t *>(y + (size_t)b * M * P + pc);
    const uint32_t *rp = RES ? reinterpret_cast<const uint32_t *>(res + (size_t)b * M * P + pc) : nullptr;
    const float *bp = bias ? bias : w;   // always a readable address: the bias load carries no branch
    const int ksteps = (K + 15) >> 4;
    const int mt_total = (M + 31) >> 5;
    const int mt_begin = blockIdx.z * mt_per_wave, mt_end = min(mt_total, mt_begin + mt_per_wave);
    if (mt_begin >= mt_end) return;

    constexpr bool WLDS = !WT && WVEC;   // forward weights through LDS (see wflat_issue)
    __shared__ __attribute__((aligned(16))) T wlds[WLDS ? 4 * 32 * (16 * KS + 8) : 8];
    T *myl = wlds + (WLDS ? wave * 32 * (16 * KS + 8) : 0);
    const int wf_row0 = WLDS ? (4 * lane) / K : 0, wf_k0 = WLDS ? (4 * lane) % K : 0, wf_sr = WLDS ? 256 / K : 0, wf_sk = WLDS ? 256 % K : 0;
    WRaw<KS> wr;
    float bl;
    uint32_t rw[16];
    auto issue = [&](int mt) {   // one tile's operand group
        const int m0 = mt * 32;
        if constexpr (WLDS) wflat_issue<KS>(wr, w, m0, M, K, lane);
        else                wraw_issue<KS, WT, WVEC>(wr, w, m0 + col, M, K, kg);
        bl = bp[min(m0 + col, M - 1)];
        if constexpr (RES) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rw[r] = rp[(size_t)min(m0 + (r & 3) + 8 * (r >> 2) + 4 * kg, M - 1) * psw];
        }
    };
    issue(mt_begin);

    s16x8 bfa[KS], bfb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ks * 16 + kg * 8 + e;
            const bool kok = k < K;
            // always-valid address, value masked afterwards: a load under a per-lane condition costs a branch each
            const uint32_t raw = xw[(kok ? k : K - 1) * xsw];
            const uint32_t v = (pok && kok) ? raw : 0u;
            bfa[ks][e] = (short)(v & 0xffffu);
            bfb[ks][e] = (short)(v >> 16);
        }
    }

    for (int mt = mt_begin; mt < mt_end; ++mt) {
        const int m0 = mt * 32;
        s16x8 af[KS];
        if constexpr (WLDS) wflat_frags<T, KS>(wr, myl, af, m0, M, K, col, kg, wf_row0, wf_k0, wf_sr, wf_sk);
        else                wraw_narrow<T, KS, WT, WVEC>(wr, af, m0 + col, M, K, kg);
        const float bcur = bias ? bl : 0.f;
        uint32_t rcur[16];
        if constexpr (RES) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rcur[r] = rw[r];
        }
        if constexpr (PF) issue(min(mt + 1, mt_total - 1));   // unconditional: the last one is wasted, a branch would cost a wait
        f32x16 acca, accb;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acca[r] = 0.f; accb[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks < ksteps) {
                acca = Mfma<T>::run(af[ks], bfa[ks], acca);
                accb = Mfma<T>::run(af[ks], bfb[ks], accb);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rin = (r & 3) + 8 * (r >> 2) + 4 * kg, row = m0 + rin;
            const float bv = __int_as_float(__builtin_amdgcn_ds_bpermute(rin << 2, __float_as_int(bcur)));
            float va = acca[r] + bv, vb = accb[r] + bv;
            if constexpr (RES) {
                float r0, r1;
                unpack2<T>(rcur[r], r0, r1);
                va += r0;
                vb += r1;
            }
            if (row < M && pok) yw[(size_t)row * psw] = pack2<T>(va, vb);
        }
        if constexpr (!PF) {
            if (mt + 1 < mt_end) issue(mt + 1);
        }
    }
}

// Forward 1x1 convolution with 192 < K <= 512 and ANY K (the EFFN's project_out: K = hidden width 255 / 127 x 2), one row tile
// per workgroup.  W(m, k) = w[m * K + k] with odd K has no 16-byte-aligned rows, so the K-chunked kernel below fetched its
// weight fragments element by element: 4-byte loads whose 32 lanes-of-a-row touch 32 different rows -- 64 cache lines per load
// instruction, 64 such instructions per chunk and tile, the texture addresser busy half of the kernel (TA_BUSY 31 K of ~63 K
// cycles at 255 -> 96, 26 us for a 23 MB problem).  But the 32 rows of a tile are ONE contiguous, 16-byte-aligned block of 32 K
// floats (m0 is a multiple of 32): the workgroup copies it with coalesced 16-byte loads, narrows it and scatters it into an LDS
// image [32][K8 + 8] of T once; all four waves then read their fragments with ds_read_b128.
// WT (input gradient, W(m, k) = w[k * M + m], M % 4 == 0): the tile is 32 contiguous floats of every row k; a 16-byte chunk holds
// 4 consecutive m of one k and is scattered down a column of the image.
template <typename T, bool WT>
__global__ void __launch_bounds__(256)
oss_conv1x1_pairw_kernel(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                         T *__restrict__ y, int M, int K, int P, int64_t xsb, int xsk, const T *__restrict__ res) {
    constexpr int KS = 8, CH = KS * 16, KMAX = 512, NI = 32 * KMAX / 4 / 256;   // 16 chunks of 16 bytes per thread at most
    __shared__ __attribute__((aligned(16))) T wl[32 * (KMAX + 8)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int p0 = (blockIdx.x * 4 + wave) * 64;
    const bool active = p0 < P;      // (no early return: the staging below has a workgroup barrier)
    const int col = lane & 31, kg = lane >> 5;
    const int p = p0 + 2 * col;
    const bool pok = active && p < P;
    const uint32_t *xw = reinterpret_cast<const uint32_t *>(x + b * xsb + (pok ? p : 0));
    const int xsw = xsk >> 1;
    const int m0 = blockIdx.z * 32;
    const int RS = ((K + 7) & ~7) + 8;

    auto load_x = [&](int kc, s16x8 (&fa)[KS], s16x8 (&fb)[KS]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kc + ks * 16 + kg * 8 + e;
                const bool kok = k < K;
                const uint32_t raw = xw[(kok ? k : K - 1) * xsw];
                const uint32_t v = (pok && kok) ? raw : 0u;
                fa[ks][e] = (short)(v & 0xffffu);
                fb[ks][e] = (short)(v >> 16);
            }
        }
    };
    s16x8 fa[KS], fb[KS];
    load_x(0, fa, fb);   // in flight across the staging

    if constexpr (WT) {   // ---- the tile's weights -> LDS (transposing)
        const int kr = tid >> 3, m4 = (tid & 7) * 4;          // 32 rows k per pass, 8 chunks of 4 columns m each
        const int mcl = min(m0 + m4, M - 4);                   // (M % 4 == 0: a chunk is inside the matrix or wholly outside)
        f32x4 q[NI];
#pragma unroll
#ifdef OSS_EXP_HALF_W
        for (int i = 0; i < NI; ++i) q[i] = *reinterpret_cast<const f32x4 *>(w + (size_t)min(kr + 32 * (i & ~1), K - 1) * M + mcl);
#else
        for (int i = 0; i < NI; ++i) q[i] = *reinterpret_cast<const f32x4 *>(w + (size_t)min(kr + 32 * i, K - 1) * M + mcl);
#endif
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int k = kr + 32 * i;
            if (k < K) {
#pragma unroll
                for (int e = 0; e < 4; ++e) wl[(m4 + e) * RS + k] = from_f32<T>(q[i][e]);
            }
        }
    } else {   // ---- the tile's weights -> LDS
        // the tile's rows are ONE contiguous block of `have` floats, copied in quads.  When `have` is not a multiple of 4 (odd K with
        // an odd number of rows in a ragged last tile: K = 193, M = 33) the last quad starts past `lim` and is read `d` elements
        // EARLIER (clamped address): its values then sit `d` slots further up in the register.  (Round 6: until then they were
        // stored unshifted and the last 1 - 3 weights of the matrix's last row were wrong -- found by tools/conv_wave_check.py.)
        const int have = min(M - m0, 32) * K, total = 32 * K, lim = max(have - 4, 0);
        const float *base = w + (size_t)m0 * K;
        f32x4 q[NI];
#pragma unroll
#ifdef OSS_EXP_HALF_W
        for (int i = 0; i < NI; ++i) q[i] = *reinterpret_cast<const f32x4 *>(base + min(4 * ((i & ~1) * 256 + tid), lim));
#else
        for (int i = 0; i < NI; ++i) q[i] = *reinterpret_cast<const f32x4 *>(base + min(4 * (i * 256 + tid), lim));
#endif
        int r = (4 * tid) / K, k = (4 * tid) - r * K;
        const int sr = 1024 / K, sk = 1024 - sr * K;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int off = 4 * (i * 256 + tid);
            if (off < total) {
                const int d = max(off - lim, 0);          // 0 except for the block's last, partial quad (1 .. 3)
                float v4[4] = {q[i][0], q[i][1], q[i][2], q[i][3]};
                if (d == 1) { v4[0] = v4[1]; v4[1] = v4[2]; v4[2] = v4[3]; }
                else if (d == 2) { v4[0] = v4[2]; v4[1] = v4[3]; }
                else if (d >= 3) { v4[0] = v4[3]; }
                int rr = r, kk = k;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (rr < 32 && off + e < have) wl[rr * RS + kk] = from_f32<T>(v4[e]);
                    if (++kk == K) { kk = 0; ++rr; }
                }
            }
            k += sk;
            r += sr;
            if (k >= K) { k -= K; ++r; }
        }
    }
    __syncthreads();
    if (!active) return;

    f32x16 acca, accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acca[r] = 0.f; accb[r] = 0.f; }
    const bool mok = m0 + col < M;
    for (int kc = 0; kc < K; kc += CH) {
        if (kc) load_x(kc, fa, fb);
        s16x8 af[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = kc + ks * 16 + kg * 8;
            // (a fragment wholly past K may lie beyond the row's pad: clamp the address, the mask below zeroes it)
            const u32x4 f = *reinterpret_cast<const u32x4 *>(wl + col * RS + min(k0, RS - 8));
            s16x8 v = __builtin_bit_cast(s16x8, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (mok && k0 + e < K) ? v[e] : (short)0;
            af[ks] = v;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (kc + ks * 16 < K) {
                acca = Mfma<T>::run(af[ks], fa[ks], acca);
                accb = Mfma<T>::run(af[ks], fb[ks], accb);
            }
        }
    }
    T *yb = y + (size_t)b * M * P;
    pair_epilogue<T>(acca, accb, bias, res ? res + (size_t)b * M * P + (pok ? p : 0) : nullptr, yb + p, m0, kg, M, P, pok);
}

// (round 4, measured and removed: the K walk software-pipelined for MT = 1 -- chunks of 4 k-steps, the next chunk's 32 activation loads
// requested behind the current chunk's weight fragments and in flight during its 8 MFMAs: 24.4 us per launch against 19.3 for the
// form below at project_in's input gradient (K = 510 / 254), 233.5 against 235.4 images/s; twice the weight-fragment round trips and a
// drain of the prefetch at the loop head cost more than the overlap bought.  tools/r4_call33.sh)
// K is walked in chunks of KS k-steps (KS * 16 channels) whose activation fragments live in registers; up to MT
// output row tiles per workgroup are accumulated across the chunks, so any K and M are covered by one kernel
// (grid.z splits M into groups of <= MT tiles).
template <typename T, int KS, int MT, bool WT, bool WVEC>
__global__ void __launch_bounds__(256)
oss_conv1x1_pairk_kernel(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                        T *__restrict__ y, int M, int K, int P, int64_t xsb, int xsk, const T *__restrict__ res) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int p0 = (blockIdx.x * 4 + wave) * 64;
    if (p0 >= P) return;
    const int col = lane & 31, kg = lane >> 5;
    const int p = p0 + 2 * col;  // even pixel of this lane's pair
    const bool pok = p < P;      // P is even: the odd pixel is in range too
    const uint32_t *xw = reinterpret_cast<const uint32_t *>(x + b * xsb + (pok ? p : 0));
    const int xsw = xsk >> 1;    // row stride in 4-byte words
    T *yb = y + (size_t)b * M * P;
    const int mt0 = blockIdx.z * MT;
    f32x16 acca[MT], accb[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acca[t][r] = 0.f; accb[t][r] = 0.f; }

    for (int kc = 0; kc < K; kc += KS * 16) {
        s16x8 bfa[KS], bfb[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kc + ks * 16 + kg * 8 + e;
                const bool kok = k < K;
                const uint32_t raw = xw[(kok ? k : K - 1) * xsw];
                const uint32_t v = (pok && kok) ? raw : 0u;
                bfa[ks][e] = (short)(v & 0xffffu);
                bfb[ks][e] = (short)(v >> 16);
            }
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            s16x8 af[KS];   // the tile's weight fragments of this chunk, fetched together (rows / columns outside the matrix: zeros)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) af[ks] = load_wfrag<T, WT, WVEC>(w, (mt0 + t) * 32 + col, M, K, kc + ks * 16 + kg * 8);
            if ((mt0 + t) * 32 < M) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (kc + ks * 16 < K) {
                        acca[t] = Mfma<T>::run(af[ks], bfa[ks], acca[t]);
                        accb[t] = Mfma<T>::run(af[ks], bfb[ks], accb[t]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
        pair_epilogue<T>(acca[t], accb[t], bias, res ? res + (size_t)b * M * P + (pok ? p : 0) : nullptr, yb + p, (mt0 + t) * 32, kg, M, P, pok);
}

// partial[slab][g][m][n] = sum over the slab's pixels of dy_g[b, m, p] x_g[b, n, p];  slabs = B * ceil(P / SLAB).
// G independent problems per launch (group g: dy + g gsg, x + g xsg); row m of dy sits at
// (m / Mh) gs_hi + (m % Mh) gsm, so that one problem can take its rows from two places (the two scan
// directions that share a flattening, oss_proj.hip).
constexpr int kWgradSlab = 512;   // pixels per partial product
// The body is shared by the one-problem launch and the grouped launch (oss_conv1x1_wgrad_grouped_kernel): `slab` of `nslabs`,
// `by` = batch * G + group, `bz` = group of four 32 x 32 tiles -- the launch coordinates of the one-problem grid.
template <typename T>
__device__ __forceinline__ void
wgrad_body(const T *__restrict__ dy, const T *__restrict__ x, float *__restrict__ part, int M, int N, int P,
           int64_t gsb, int64_t gsm, int64_t xsb, int64_t xsn, int G, int64_t gsg, int64_t xsg, int Mh,
           int64_t gs_hi, int NB /* N, or N + 1: a virtual all-ones row of x whose column of dW is dbias */,
           int slab, int nslabs, int by, int bz, int span /* pixels per partial product: a multiple of kWgradSlab */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = by / G, g = by - b * G;
    const int pbeg = slab * span, pend = min(P, pbeg + span);
    const int col = lane & 31, kg = lane >> 5;
    const int mt = (M + 31) >> 5, nt = (NB + 31) >> 5;
    const T *gb = dy + b * gsb + g * gsg;
    const T *xb = x + b * xsb + g * xsg;
    // one partial vector per (batch, slab): [G * M rows in DESTINATION order][N], then (NB > N) the M dbias sums
    const size_t pvec = (size_t)G * M * N + (NB > N ? M : 0);
    float *pb = part + (size_t)(b * nslabs + slab) * pvec;
    const short kOne = (short)from_f32<T>(1.0f).v;  // 1.0 in the I/O type
    const s16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0}, ones8 = {kOne, kOne, kOne, kOne, kOne, kOne, kOne, kOne};
    const bool aligned = (((reinterpret_cast<uintptr_t>(gb) | reinterpret_cast<uintptr_t>(xb)) & 15u) == 0) &&
                         (gsm % 8 == 0) && (xsn % 8 == 0) && (pbeg % 8 == 0) && (gs_hi % 8 == 0);
    {
        const int tile = bz * 4 + wave;  // one 32 x 32 tile of dW per wave
        if (tile >= mt * nt) return;
        const int m0 = (tile / nt) * 32, n0 = (tile % nt) * 32;
        const int mrow = m0 + col, nrow = n0 + col;
        const bool mok = mrow < M, nok = nrow < N, one = nrow == N && NB > N;
        const int mr = mok ? mrow : 0;
        const T *ga = gb + (mr / Mh) * gs_hi + (mr % Mh) * gsm;
        const T *xa = xb + (nok ? nrow : 0) * xsn;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;

        int pk = pbeg;
#ifndef OSS_EXP_WGRAD_DIRECT
        for (; aligned && pend - pk >= kWgradSlab; pk += kWgradSlab) {   // full kWgradSlab pieces of the span, one after the other
            // A full slab, operands through LDS.  The MFMA fragment of a lane is 8 consecutive pixels of ITS OWN row (row =
            // lane & 31): loaded straight from memory, one 16-byte load instruction touches 32 rows -- 64 separate cache
            // lines of which it uses 16 bytes each -- and the texture addresser, which takes a cycle per line, was busy for
            // the whole kernel (TA_BUSY 41.5 K of 43 K cycles, profiles/r02_pmc_wgrad_ta_bound.txt).  Here a load instruction
            // covers 8 rows x 128 contiguous bytes (lane = (row % 8, 16-byte chunk)), the wave parks the 32 x 64-pixel pieces
            // of both operands in its own LDS region (rows padded to 144 bytes: conflict-free 16-byte reads) and reads the
            // fragments back.  Wave-private, so no barrier: LDS operations of one wave execute in order.
            constexpr int PIECES = kWgradSlab / 64, RS = 72;   // row stride in elements (64 + 8 pad)
            __shared__ __attribute__((aligned(16))) T stage[4][2][32 * RS];
            T *sa = stage[wave][0], *sb = stage[wave][1];
            const int r8 = lane >> 3, q = lane & 7;
            const T *pa[4], *px[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int mr2 = min(m0 + r8 + 8 * j, M - 1), nr2 = min(n0 + r8 + 8 * j, N - 1);
                pa[j] = gb + (mr2 / Mh) * gs_hi + (mr2 % Mh) * gsm + pk + q * 8;
                px[j] = xb + nr2 * xsn + pk + q * 8;
            }
            // a window of WIN pieces in flight: piece pc + WIN is requested as soon as piece pc has left its registers
            constexpr int WIN = PIECES < 4 ? PIECES : 4;
            u32x4 qa[WIN][4], qb[WIN][4];
#pragma unroll
            for (int pc = 0; pc < WIN; ++pc)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    qa[pc][j] = *reinterpret_cast<const u32x4 *>(pa[j] + pc * 64);
                    qb[pc][j] = *reinterpret_cast<const u32x4 *>(px[j] + pc * 64);
                }
#pragma unroll
            for (int pc = 0; pc < PIECES; ++pc) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    *reinterpret_cast<u32x4 *>(sa + (r8 + 8 * j) * RS + q * 8) = qa[pc % WIN][j];
                    *reinterpret_cast<u32x4 *>(sb + (r8 + 8 * j) * RS + q * 8) = qb[pc % WIN][j];
                }
                if (pc + WIN < PIECES) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        qa[pc % WIN][j] = *reinterpret_cast<const u32x4 *>(pa[j] + (pc + WIN) * 64);
                        qb[pc % WIN][j] = *reinterpret_cast<const u32x4 *>(px[j] + (pc + WIN) * 64);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                u32x4 fa[4], fb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    fa[u] = *reinterpret_cast<const u32x4 *>(sa + col * RS + u * 16 + kg * 8);
                    fb[u] = *reinterpret_cast<const u32x4 *>(sb + col * RS + u * 16 + kg * 8);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const s16x8 af = mok ? __builtin_bit_cast(s16x8, fa[u]) : zero8;
                    const s16x8 bf = nok ? __builtin_bit_cast(s16x8, fb[u]) : (one ? ones8 : zero8);
                    acc = Mfma<T>::run(af, bf, acc);
                }
            }
            __builtin_amdgcn_wave_barrier();   // the next piece's LDS writes stay behind this piece's reads
        }
#else
        if (aligned && pend - pk == kWgradSlab) {
            // (timing experiment) a full slab: ALL its operand loads (2 x 16 bytes per k-step and lane) are issued before the first MFMA.
            // A k-step is one MFMA (64 cycles) but a load is a ~1 us round trip: walked 64 pixels at a time (8 loads,
            // wait, 4 MFMAs) the kernel was a chain of 8 round trips at 1.5 waves per SIMD.
            constexpr int IT = kWgradSlab / 16;
            u32x4 qa[IT], qb[IT];
#pragma unroll
            for (int u = 0; u < IT; ++u) {
#ifdef OSS_EXP_WGRAD_NOLOAD   // timing experiment: operands without memory traffic
                qa[u] = u32x4{(uint32_t)lane, (uint32_t)u, 1u, 2u};
                qb[u] = u32x4{(uint32_t)lane, (uint32_t)u, 3u, 4u};
#else
                qa[u] = *reinterpret_cast<const u32x4 *>(ga + pk + u * 16 + kg * 8);
                qb[u] = *reinterpret_cast<const u32x4 *>(xa + pk + u * 16 + kg * 8);
#endif
            }
#pragma unroll
            for (int u = 0; u < IT; ++u) {
                const s16x8 af = mok ? __builtin_bit_cast(s16x8, qa[u]) : zero8;
                const s16x8 bf = nok ? __builtin_bit_cast(s16x8, qb[u]) : (one ? ones8 : zero8);
                acc = Mfma<T>::run(af, bf, acc);
            }
            pk = pend;
        } else
#endif
        if (aligned) {
            // 4 k-steps (64 pixels) per iteration: all eight 16-byte loads are issued before the first MFMA needs them
            for (; pk + 64 <= pend; pk += 64) {
                u32x4 qa[4], qb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    qa[u] = *reinterpret_cast<const u32x4 *>(ga + pk + u * 16 + kg * 8);
                    qb[u] = *reinterpret_cast<const u32x4 *>(xa + pk + u * 16 + kg * 8);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const s16x8 af = mok ? __builtin_bit_cast(s16x8, qa[u]) : zero8;
                    const s16x8 bf = nok ? __builtin_bit_cast(s16x8, qb[u]) : (one ? ones8 : zero8);
                    acc = Mfma<T>::run(af, bf, acc);
                }
            }
        }
        for (; pk < pend; pk += 16) {
            const int k0 = pk + kg * 8;
            s16x8 af, bf;
            if (aligned && k0 + 8 <= pend) {
                const u32x4 qa = *reinterpret_cast<const u32x4 *>(ga + k0);
                const u32x4 qb = *reinterpret_cast<const u32x4 *>(xa + k0);
                af = mok ? __builtin_bit_cast(s16x8, qa) : zero8;
                bf = nok ? __builtin_bit_cast(s16x8, qb) : (one ? ones8 : zero8);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool kok = (k0 + e) < pend;
                    const int kc = kok ? k0 + e : pend - 1;
                    const short av = (short)ga[kc].v, bv = (short)xa[kc].v;
                    af[e] = (mok && kok) ? av : (short)0;
                    bf[e] = kok ? (nok ? bv : (one ? kOne : (short)0)) : (short)0;
                }
            }
            acc = Mfma<T>::run(af, bf, acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            const int cn = n0 + col;
#ifdef OSS_EXP_WGRAD_NOSTORE   // timing experiment: results not written
            if (row < M && cn < NB && acc[r] == 123456.789f) {
#else
            if (row < M && cn < NB) {
#endif
                const size_t drow = ((size_t)(row / Mh) * G + g) * Mh + row % Mh;  // destination row of (group g, row)
                if (cn < N) pb[drow * N + cn] = acc[r];
                else pb[(size_t)G * M * N + row] = acc[r];   // dbias (only ever with G == 1)
            }
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
oss_conv1x1_wgrad_kernel(const T *__restrict__ dy, const T *__restrict__ x, float *__restrict__ part, int M, int N, int P,
                         int64_t gsb, int64_t gsm, int64_t xsb, int64_t xsn, int G, int64_t gsg, int64_t xsg, int Mh,
                         int64_t gs_hi, int NB) {
    wgrad_body<T>(dy, x, part, M, N, P, gsb, gsm, xsb, xsn, G, gsg, xsg, Mh, gs_hi, NB, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y,
                  (int)blockIdx.z, kWgradSlab);
}

// Grouped launch: ALL weight-gradient products of a backward pass as one kernel.  Nobody needs a weight gradient before the
// optimizer, yet each one-problem launch is a single under-filled round of <= ~200 workgroups whose 10 us are latency, not
// work (302 launches = 3.2 ms of a 40 ms step, profiles/r03_rocprof_bench_steady_state_v1.txt).  With oss_set_defer_wgrad(1)
// conv1x1_wgrad() only records its problem; oss_flush_wgrads() copies the descriptor table to the device and runs this
// kernel over every recorded problem's workgroups back to back (block -> problem through a 16-bit table), so the chip is
// full for the whole launch and the operands stream at memory speed.  Same per-problem arithmetic; with a span of one 512-pixel
// piece per partial (oss_conv1x1_wgrad_set_span(1)) also the same partial layout and finishing sums, i.e. bit-identical weight
// gradients -- the default span of 4 pieces adds the same terms in another order (equal to fp32 round-off, run-to-run stable).
struct WgradDesc {
    const void *dy, *x;
    float *part;
    int64_t gsb, gsm, xsb, xsn, gsg, xsg, gs_hi;
    int M, N, P, G, Mh, NB, slabs, bgs /* batch * G */;
    unsigned first_block;
    int io;
    int span, reserved_;   // pixels per partial product
};
template <typename T>
__global__ void __launch_bounds__(256)
oss_conv1x1_wgrad_grouped_kernel(const WgradDesc *__restrict__ descs, const uint16_t *__restrict__ block_problem) {
    const WgradDesc d = descs[block_problem[blockIdx.x]];
    const unsigned local = blockIdx.x - d.first_block;
    const int slab = (int)(local % (unsigned)d.slabs);
    const unsigned r = local / (unsigned)d.slabs;
    const int by = (int)(r % (unsigned)d.bgs), bz = (int)(r / (unsigned)d.bgs);
    // (round 4, measured and removed: the tile groups bz of one (slab, batch) on consecutive slots of ONE XCD so that they meet in its
    // L2 -- 233.2 images/s without against 232.5 / 232.8 with, FETCH_SIZE 5.75 GB either way: the launch's 5.9 GB are the operands
    // themselves, held since the forward / backward pass, read once at 4 TB/s; profiles/r04_pmc_grouped_wgrad.txt)
    const WgradDesc *dt = descs + block_problem[blockIdx.x];
    wgrad_body<T>(table_ptr<const T>(&dt->dy), table_ptr<const T>(&dt->x), table_ptr<float>(&dt->part), d.M, d.N, d.P, d.gsb, d.gsm, d.xsb,
                  d.xsn, d.G, d.gsg, d.xsg, d.Mh, d.gs_hi, d.NB, slab, d.slabs, by, bz, d.span);
}

// out[j] = sum over the nslab partial vectors (fixed order), j < nw -> dw[j], else db[j - nw].
// 64 outputs x 4 slices of the slab list per workgroup: slice s adds slabs s, s + 4, ... (4 loads in flight),
// the four slice sums are combined in a fixed order.
__global__ void __launch_bounds__(256)
oss_conv1x1_wgrad_finish(const float *__restrict__ part, float *__restrict__ dw, float *__restrict__ db, int nslab, size_t pvec,
                         size_t nw) {
    __shared__ float red[4][64];
    const int colx = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + colx;
    float s = 0.f;
    if (i < pvec) {
        const float *pp = part + i;
        int k = slice;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (; k + 12 < nslab; k += 16) {
            s0 += pp[(size_t)k * pvec];
            s1 += pp[(size_t)(k + 4) * pvec];
            s2 += pp[(size_t)(k + 8) * pvec];
            s3 += pp[(size_t)(k + 12) * pvec];
        }
        for (; k < nslab; k += 4) s0 += pp[(size_t)k * pvec];
        s = (s0 + s1) + (s2 + s3);
    }
    red[slice][colx] = s;
    __syncthreads();
    if (slice == 0 && i < pvec) {
        const float t = (red[0][colx] + red[1][colx]) + (red[2][colx] + red[3][colx]);
        if (i < nw) dw[i] = t;
        else db[i - nw] = t;
    }
}

// how many waves a launch of the whole-K kernels aims for (row tiles are dealt to more workgroups until it is reached);
// VMAMBAIR_CONV1X1_WAVES overrides for A-B timing
static long conv1x1_target_waves() {
    static const long v = [] {
        const char *e = getenv("VMAMBAIR_CONV1X1_WAVES");
        const long t = e ? atol(e) : 0;
        return t > 0 ? t : 1024L;   // 1024 / 2048 / 4096: 177.0 / 176.2 / 175.0 images/s (profiles/r02_conv1x1_pipeline.txt)
    }();
    return v;
}

template <typename T>
static void conv1x1_launch(const T *x, const float *w, const float *bias, T *y, int B, int M, int K, int P, int64_t xsb,
                           int64_t xsk, int64_t ws_m, int64_t ws_k, hipStream_t s, const T *res) {
    const int pblocks = (P + 127) / 128, mt = (M + 31) / 32;
    const bool wt = (ws_m == 1 && ws_k == M);          // input gradient: weights read transposed
    const bool plain = (ws_k == 1 && ws_m == K);
    // pixel-pair form: even P, even row strides, 4-byte aligned bases
    const bool pair_ok = P % 2 == 0 && xsk % 2 == 0 && xsb % 2 == 0 && xsk < (1 << 24) && (size_t)M * K < (1u << 30) &&
                         (wt || plain) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 3u) == 0;
    if (pair_ok) {
        const bool wvec = plain && K % 8 == 0 && (reinterpret_cast<uintptr_t>(w) & 15u) == 0;
        const int pb = (P + 255) / 256;
        const long waves_p = (long)B * ((P + 63) / 64);
        const int xk = (int)xsk;
        if (plain && !wvec && K > 16 * 3 && K <= 512 && (reinterpret_cast<uintptr_t>(w) & 15u) == 0) {
            // forward with rows that are not 16-byte aligned (K = 127, the EFFN hidden width at dim 48): weight tile through LDS
            hipLaunchKernelGGL((oss_conv1x1_pairw_kernel<T, false>), dim3(pb, B, mt), dim3(256), 0, s, x, w, bias, y, M, K, P, xsb, xk, res);
            return;
        }
        if (K <= 16 * 12) {
            // whole K in registers: enough workgroups to fill the chip twice, otherwise as few activation re-loads as possible
            int split = (int)((conv1x1_target_waves() + waves_p - 1) / waves_p);
            if (split < 1) split = 1;
            if (split > mt) split = mt;
            const int per = (mt + split - 1) / split;
            dim3 grid(pb, B, (mt + per - 1) / per);
#define OSS_PAIR2(KS_, WT_, WV_, PF_) do { if (res) hipLaunchKernelGGL((oss_conv1x1_pair_kernel<T, KS_, WT_, WV_, true, PF_>), grid, dim3(256), 0, s, x, w, bias, y, M, K, P, xsb, xk, per, res); \
                                           else     hipLaunchKernelGGL((oss_conv1x1_pair_kernel<T, KS_, WT_, WV_, false, PF_>), grid, dim3(256), 0, s, x, w, bias, y, M, K, P, xsb, xk, per, res); } while (0)
#define OSS_PAIR(KS_, PF_) do { if (wt) OSS_PAIR2(KS_, true, false, PF_); else if (wvec) OSS_PAIR2(KS_, false, true, PF_); else OSS_PAIR2(KS_, false, false, PF_); } while (0)
            // PF (next tile's operands in flight during the current tile): the raw fp32 fragments of two tiles fit up to 6 k-steps
            if (K <= 16 * 3)      OSS_PAIR(3, true);
            else if (K <= 16 * 6) OSS_PAIR(6, true);
            else if (K <= 16 * 8) OSS_PAIR(8, false);    // K = 127 (EFFN hidden width at dim 48): no duplicate k-steps
            else                  OSS_PAIR(12, false);
#undef OSS_PAIR2
#undef OSS_PAIR
#undef OSS_PAIR1
        } else {
            // K in chunks of 128: up to 3 row tiles accumulate per workgroup (fewest re-loads that still fill the chip)
            int per = 3;
            while (per > 1 && waves_p * ((mt + per - 1) / per) < 4096) --per;
            if (per > mt) per = mt;
            // forward: weights through LDS.  (The same for the input gradient -- template WT, parity-green -- measured SLOWER,
            // 188.0 against 191.0 images/s: its element-wise loads are coalesced already (32 consecutive m per k) and the
            // transposing LDS scatter plus the barrier cost more than they save.  Kept for A-B runs: VMAMBAIR_CONV1X1_WT_LDS=1.)
            static const bool wt_lds = getenv("VMAMBAIR_CONV1X1_WT_LDS") != nullptr;
            if (per == 1 && K <= 512 && (reinterpret_cast<uintptr_t>(w) & 15u) == 0 && (plain || (wt && wt_lds && M % 4 == 0))) {
                if (plain) hipLaunchKernelGGL((oss_conv1x1_pairw_kernel<T, false>), dim3(pb, B, mt), dim3(256), 0, s, x, w, bias, y, M, K, P, xsb, xk, res);
                else       hipLaunchKernelGGL((oss_conv1x1_pairw_kernel<T, true>), dim3(pb, B, mt), dim3(256), 0, s, x, w, bias, y, M, K, P, xsb, xk, res);
                return;
            }
            dim3 grid(pb, B, (mt + per - 1) / per);
#define OSS_PAIR1(MT_, WT_, WV_) hipLaunchKernelGGL((oss_conv1x1_pairk_kernel<T, 8, MT_, WT_, WV_>), grid, dim3(256), 0, s, x, w, bias, y, M, K, P, xsb, xk, res)
#define OSS_PAIR(MT_) do { if (wt) OSS_PAIR1(MT_, true, false); else if (wvec) OSS_PAIR1(MT_, false, true); else OSS_PAIR1(MT_, false, false); } while (0)
            if (per == 1) OSS_PAIR(1); else if (per == 2) OSS_PAIR(2); else OSS_PAIR(3);
#undef OSS_PAIR
#undef OSS_PAIR1
        }
        return;
    }
    const bool reuse_ok = K <= 16 * 12 && xsk < (1 << 24) && (size_t)M * K < (1u << 30) &&
                          (wt || (plain && K % 8 == 0 && (reinterpret_cast<uintptr_t>(w) & 15u) == 0));
    if (reuse_ok) {
        // enough waves to fill 256 CUs x 4 SIMDs twice, otherwise as few activation re-loads as possible
        const long waves_p = (long)B * ((P + 31) / 32);
        int split = (int)((2048 + waves_p - 1) / waves_p);
        if (split < 1) split = 1;
        if (split > mt) split = mt;
        const int per = (mt + split - 1) / split;
        dim3 grid(pblocks, B, (mt + per - 1) / per);
        const int xk = (int)xsk;
        if (K <= 16 * 6) {
            if (wt) hipLaunchKernelGGL((oss_conv1x1_reuse_kernel<T, 6, true>), grid, dim3(256), 0, s, x, w, bias, y, M, K, P, xsb, xk, per, res);
            else    hipLaunchKernelGGL((oss_conv1x1_reuse_kernel<T, 6, false>), grid, dim3(256), 0, s, x, w, bias, y, M, K, P, xsb, xk, per, res);
        } else {
            if (wt) hipLaunchKernelGGL((oss_conv1x1_reuse_kernel<T, 12, true>), grid, dim3(256), 0, s, x, w, bias, y, M, K, P, xsb, xk, per, res);
            else    hipLaunchKernelGGL((oss_conv1x1_reuse_kernel<T, 12, false>), grid, dim3(256), 0, s, x, w, bias, y, M, K, P, xsb, xk, per, res);
        }
    } else {
        dim3 grid(pblocks, B, mt);
        hipLaunchKernelGGL(oss_conv1x1_kernel<T>, grid, dim3(256), 0, s, x, w, bias, y, M, K, P, xsb, xsk, ws_m, ws_k, res);
    }
}

// 0 = never the workgroup-level kernel of oss_conv1x1_wg.hip (A-B timing).  Initial value from VMAMBAIR_CONV1X1_WG (default on).
static std::atomic<int> g_conv_wg{-1};
static bool conv_wg_on() {
    int m = g_conv_wg.load();
    if (m < 0) {
        const char *e = std::getenv("VMAMBAIR_CONV1X1_WG");
        m = (e && std::atoi(e) == 0) ? 0 : 1;
        g_conv_wg.store(m);
    }
    return m != 0;
}

void conv1x1_set_wg(int on) { g_conv_wg.store(on ? 1 : 0); }

int conv1x1(oss_dtype io, const void *x, const float *w, const float *bias, void *y, int B, int M, int K, int P, int64_t xsb,
            int64_t xsk, int64_t ws_m, int64_t ws_k, hipStream_t s, const void *res) {
    {
        const bool wt = (ws_m == 1 && ws_k == M), plain = (ws_k == 1 && ws_m == K);
        if ((wt || plain) && conv_wg_on() && conv1x1_wg_ok(io, M, K, P, xsb, xsk, x, y, w, res))
            return conv1x1_wg(io, x, w, bias, y, B, M, K, P, xsb, xsk, wt ? 1 : 0, s, res);
    }
    switch (io) {
        case OSS_BF16:
            conv1x1_launch<bf16_t>(reinterpret_cast<const bf16_t *>(x), w, bias, reinterpret_cast<bf16_t *>(y), B, M, K, P, xsb,
                                   xsk, ws_m, ws_k, s, reinterpret_cast<const bf16_t *>(res));
            break;
        case OSS_F16:
            conv1x1_launch<f16_t>(reinterpret_cast<const f16_t *>(x), w, bias, reinterpret_cast<f16_t *>(y), B, M, K, P, xsb, xsk,
                                  ws_m, ws_k, s, reinterpret_cast<const f16_t *>(res));
            break;
        default:   // fp32 I/O: true fp32 on the matrix cores (oss_conv1x1_f32.hip), no reduced-precision detour
            return conv1x1_f32(reinterpret_cast<const float *>(x), w, bias, reinterpret_cast<float *>(y), B, M, K, P, xsb, xsk, ws_m, ws_k,
                               s, reinterpret_cast<const float *>(res));
    }
    return (int)hipGetLastError();
}

// The same partial product with TM x TN MFMA tiles (32 TM rows of dW x 32 TN columns) per wave: a wave then reads its 32 TM
// rows of dy and 32 TN rows of x once for TM TN tiles, so the launch re-reads dy ceil(NB / 32 TN) times and x
// ceil(M / 32 TM) times instead of ceil(NB / 32) and ceil(M / 32).  The 1 x 1 kernel above moves 22 MB through L2 for the
// 9.4 MB of a (96 x 48, 8 x 4096 pixels) problem and runs at about the speed that traffic allows; 2 x 2 moves 12.6 MB.
// OPT-IN (VMAMBAIR_WGRAD_TILE = 12 | 21 | 22) until measured on the box; identical partial layout and finishing.
template <typename T, int TM, int TN>
__global__ void __launch_bounds__(256)
oss_conv1x1_wgrad_tiles_kernel(const T *__restrict__ dy, const T *__restrict__ x, float *__restrict__ part, int M, int N, int P,
                               int64_t gsb, int64_t gsm, int64_t xsb, int64_t xsn, int G, int64_t gsg, int64_t xsg, int Mh,
                               int64_t gs_hi, int NB) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y / G, g = blockIdx.y - b * G, slab = blockIdx.x;
    const int pbeg = slab * kWgradSlab, pend = min(P, pbeg + kWgradSlab);
    const int col = lane & 31, kg = lane >> 5;
    const int mt = (M + 32 * TM - 1) / (32 * TM), nt = (NB + 32 * TN - 1) / (32 * TN);
    const T *gb = dy + b * gsb + g * gsg;
    const T *xb = x + b * xsb + g * xsg;
    const size_t pvec = (size_t)G * M * N + (NB > N ? M : 0);
    float *pb = part + (size_t)(b * gridDim.x + slab) * pvec;
    const short kOne = (short)from_f32<T>(1.0f).v;
    const s16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0}, ones8 = {kOne, kOne, kOne, kOne, kOne, kOne, kOne, kOne};
    const bool aligned = (((reinterpret_cast<uintptr_t>(gb) | reinterpret_cast<uintptr_t>(xb)) & 15u) == 0) &&
                         (gsm % 8 == 0) && (xsn % 8 == 0) && (pbeg % 8 == 0) && (gs_hi % 8 == 0);
    const int tile = blockIdx.z * 4 + wave;  // one (32 TM) x (32 TN) tile of dW per wave
    if (tile >= mt * nt) return;
    const int m0 = (tile / nt) * 32 * TM, n0 = (tile % nt) * 32 * TN;
    const T *ga[TM];
    const T *xa[TN];
    bool mok[TM], nok[TN], one[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mrow = m0 + 32 * i + col;
        mok[i] = mrow < M;
        const int mr = mok[i] ? mrow : 0;
        ga[i] = gb + (mr / Mh) * gs_hi + (mr % Mh) * gsm;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nrow = n0 + 32 * j + col;
        nok[j] = nrow < N;
        one[j] = nrow == N && NB > N;
        xa[j] = xb + (nok[j] ? nrow : 0) * xsn;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // operand fragments of one 16-pixel k-step (8 pixels per lane half) for every row tile / column tile
    auto frag = [&](const T *row, int k0, bool ok, bool ones, bool fast) -> s16x8 {
        if (fast) {
            const u32x4 q = *reinterpret_cast<const u32x4 *>(row + k0);
            return ok ? __builtin_bit_cast(s16x8, q) : (ones ? ones8 : zero8);
        }
        s16x8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool kok = (k0 + e) < pend;
            const int kc = kok ? k0 + e : pend - 1;
            const short v = (short)row[kc].v;
            f[e] = kok ? (ok ? v : (ones ? kOne : (short)0)) : (short)0;
        }
        return f;
    };
    int pk = pbeg;
    if (aligned && pend - pbeg == kWgradSlab) {
        // a full slab: every operand load first, then the MFMAs (see oss_conv1x1_wgrad_kernel)
        constexpr int IT = kWgradSlab / 16;
        u32x4 qa[IT][TM], qb[IT][TN];
#pragma unroll
        for (int u = 0; u < IT; ++u) {
#pragma unroll
            for (int i = 0; i < TM; ++i) qa[u][i] = *reinterpret_cast<const u32x4 *>(ga[i] + pbeg + u * 16 + kg * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) qb[u][j] = *reinterpret_cast<const u32x4 *>(xa[j] + pbeg + u * 16 + kg * 8);
        }
#pragma unroll
        for (int u = 0; u < IT; ++u)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const s16x8 af = mok[i] ? __builtin_bit_cast(s16x8, qa[u][i]) : zero8;
                    const s16x8 bf = nok[j] ? __builtin_bit_cast(s16x8, qb[u][j]) : (one[j] ? ones8 : zero8);
                    acc[i][j] = Mfma<T>::run(af, bf, acc[i][j]);
                }
        pk = pend;
    } else if (aligned) {
        for (; pk + 32 <= pend; pk += 32) {   // 2 k-steps per iteration: 2 (TM + TN) 16-byte loads in flight
            s16x8 af[2][TM], bf[2][TN];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[u][i] = frag(ga[i], pk + u * 16 + kg * 8, mok[i], false, true);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[u][j] = frag(xa[j], pk + u * 16 + kg * 8, nok[j], one[j], true);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = Mfma<T>::run(af[u][i], bf[u][j], acc[i][j]);
        }
    }
    for (; pk < pend; pk += 16) {
        const int k0 = pk + kg * 8;
        const bool fast = aligned && k0 + 8 <= pend;
        s16x8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = frag(ga[i], k0, mok[i], false, fast);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = frag(xa[j], k0, nok[j], one[j], fast);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = Mfma<T>::run(af[i], bf[j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kg;
                const int cn = n0 + 32 * j + col;
                if (row < M && cn < NB) {
                    const size_t drow = ((size_t)(row / Mh) * G + g) * Mh + row % Mh;
                    if (cn < N) pb[drow * N + cn] = acc[i][j][r];
                    else pb[(size_t)G * M * N + row] = acc[i][j][r];
                }
            }
}

// 0 = the 1 x 1 kernel (default); 12, 21, 22 = TM x TN tiles per wave.  Initial value from VMAMBAIR_WGRAD_TILE,
// changed at run time by oss_conv1x1_wgrad_set_tile (tests, A-B timing)
static std::atomic<int> g_wgrad_tile{-1};
static int wgrad_tile_mode() {
    int m = g_wgrad_tile.load();
    if (m < 0) {
        const char *e = std::getenv("VMAMBAIR_WGRAD_TILE");
        const int v = e ? std::atoi(e) : 0;
        m = (v == 12 || v == 21 || v == 22) ? v : 0;
        g_wgrad_tile.store(m);
    }
    return m;
}
void conv1x1_wgrad_set_tile(int mode) { g_wgrad_tile.store((mode == 12 || mode == 21 || mode == 22) ? mode : 0); }

template <typename T>
static void wgrad_tiles_launch(int mode, const T *dy, const T *x, float *part, int B, int M, int N, int P, int64_t gsb, int64_t gsm,
                               int64_t xsb, int64_t xsn, int G, int64_t gsg, int64_t xsg, int Mh, int64_t gs_hi, int NB,
                               int slabs, hipStream_t s) {
    const int tm = mode / 10, tn = mode % 10;
    const int tiles = ((M + 32 * tm - 1) / (32 * tm)) * ((NB + 32 * tn - 1) / (32 * tn));
    dim3 grid(slabs, B * G, (tiles + 3) / 4);
#define OSS_WG(TM_, TN_) hipLaunchKernelGGL((oss_conv1x1_wgrad_tiles_kernel<T, TM_, TN_>), grid, dim3(256), 0, s, dy, x, part, M, N, P, \
                                            gsb, gsm, xsb, xsn, G, gsg, xsg, Mh, gs_hi, NB)
    if (mode == 12) OSS_WG(1, 2); else if (mode == 21) OSS_WG(2, 1); else OSS_WG(2, 2);
#undef OSS_WG
}

int conv1x1_wgrad_slabs(int P) { return (P + kWgradSlab - 1) / kWgradSlab; }

// Pixels per partial product of the GROUPED launch, in units of kWgradSlab.  One-problem launches keep 1 (they need every
// workgroup they can get); in the grouped launch thousands of workgroups are queued anyway and every partial vector is
// M x N floats written once and read once by the finishing sum -- at 512 pixels the headline step moved 0.9 GB of partials
// each way (the finishing launch ran 222 us at memory speed).  Initial value from VMAMBAIR_WGRAD_SPAN.
static std::atomic<int> g_wgrad_span{-1};
static int wgrad_span_mult() {
    int m = g_wgrad_span.load();
    if (m < 0) {
        const char *e = std::getenv("VMAMBAIR_WGRAD_SPAN");
        m = e ? std::atoi(e) : 4;
        m = m < 1 ? 1 : (m > 64 ? 64 : m);
        g_wgrad_span.store(m);
    }
    return m;
}
void conv1x1_wgrad_set_span(int mult) { g_wgrad_span.store(mult < 1 ? 1 : (mult > 64 ? 64 : mult)); }

int conv1x1_wgrad(oss_dtype io, const void *dy, const void *x, float *dw, float *part, int B, int M, int N, int P,
                  int64_t gsb, int64_t gsm, int64_t xsb, int64_t xsn, hipStream_t s, int G, int64_t gsg, int64_t xsg, int Mh,
                  int64_t gs_hi, float *db) {
    const int NB = N + (db ? 1 : 0);
    if (G < 1 || (size_t)B * G > 65535) return OSS_ERR_SHAPE;
    if (io == OSS_F32) {   // plain 1x1 weight gradient only (the projection products call rows_f32_wgrad themselves)
        if (G != 1 || (Mh > 0 && Mh != M)) return OSS_ERR_SHAPE;
        return rows_f32_wgrad(reinterpret_cast<const float *>(dy), reinterpret_cast<const float *>(x), dw, part, B, 1, 1, M, N, P, gsb, 0,
                              gsm, xsb, 0, xsn, s, db);
    }
    if (Mh <= 0 || Mh > M) Mh = M;
    int slabs = conv1x1_wgrad_slabs(P);
    const int tiles = ((M + 31) / 32) * ((NB + 31) / 32);
    dim3 grid(slabs, B * G, (tiles + 3) / 4);
    const int tmode = wgrad_tile_mode();
    if (defer_wgrad() && !tmode && (io == OSS_BF16 || io == OSS_F16) && (size_t)slabs * B * G * ((tiles + 3) / 4) < (1u << 24)) {
        // a recorded product's partials do not exist until oss_flush_wgrads: its finishing sum must be deferred too (it would
        // otherwise run right below, on a buffer nobody has written yet)
        if (!defer_finish()) return OSS_ERR_WORKSPACE;
        WgradDesc d;
        d.span = kWgradSlab * wgrad_span_mult(); d.reserved_ = 0;
        slabs = (P + d.span - 1) / d.span;   // never more than conv1x1_wgrad_slabs(P): the caller's partial buffer is large enough
        d.dy = dy; d.x = x; d.part = part;
        d.gsb = gsb; d.gsm = gsm; d.xsb = xsb; d.xsn = xsn; d.gsg = gsg; d.xsg = xsg; d.gs_hi = gs_hi;
        d.M = M; d.N = N; d.P = P; d.G = G; d.Mh = Mh; d.NB = NB; d.slabs = slabs; d.bgs = B * G;
        d.first_block = 0; d.io = (int)io;
        defer_wgrad_push(&d, sizeof(d), (unsigned)(slabs * B * G * ((tiles + 3) / 4)));
    } else
    if (tmode && (io == OSS_BF16 || io == OSS_F16)) {
        if (io == OSS_BF16)
            wgrad_tiles_launch<bf16_t>(tmode, reinterpret_cast<const bf16_t *>(dy), reinterpret_cast<const bf16_t *>(x), part, B, M, N, P,
                                       gsb, gsm, xsb, xsn, G, gsg, xsg, Mh, gs_hi, NB, slabs, s);
        else
            wgrad_tiles_launch<f16_t>(tmode, reinterpret_cast<const f16_t *>(dy), reinterpret_cast<const f16_t *>(x), part, B, M, N, P,
                                      gsb, gsm, xsb, xsn, G, gsg, xsg, Mh, gs_hi, NB, slabs, s);
    } else
    switch (io) {
        case OSS_BF16:
            hipLaunchKernelGGL(oss_conv1x1_wgrad_kernel<bf16_t>, grid, dim3(256), 0, s, reinterpret_cast<const bf16_t *>(dy),
                               reinterpret_cast<const bf16_t *>(x), part, M, N, P, gsb, gsm, xsb, xsn, G, gsg, xsg, Mh, gs_hi, NB);
            break;
        case OSS_F16:
            hipLaunchKernelGGL(oss_conv1x1_wgrad_kernel<f16_t>, grid, dim3(256), 0, s, reinterpret_cast<const f16_t *>(dy),
                               reinterpret_cast<const f16_t *>(x), part, M, N, P, gsb, gsm, xsb, xsn, G, gsg, xsg, Mh, gs_hi, NB);
            break;
        default: return OSS_ERR_SHAPE;
    }
    if (db && G != 1) return OSS_ERR_SHAPE;
    const size_t nw = (size_t)G * M * N, pvec = nw + (db ? M : 0);
    if (defer_finish())
        defer_sum(part, slabs * B, pvec, pvec, dw, nw, db);
    else
        hipLaunchKernelGGL(oss_conv1x1_wgrad_finish, dim3((unsigned)((pvec + 63) / 64)), dim3(256), 0, s, part, dw, db, slabs * B,
                           pvec, nw);
    return (int)hipGetLastError();
}

// ---- grouped weight gradients: flush (oss_capi.hip keeps the recorded descriptors as raw bytes + block counts) -----------------
size_t wgrad_desc_bytes() { return sizeof(WgradDesc); }
// descs: n descriptors in host memory with first_block filled; d_descs / d_map: their device copies (already queued on s)
int wgrad_grouped_launch(int io, const void *d_descs, const void *d_map, unsigned total_blocks, hipStream_t s) {
    if (total_blocks == 0) return 0;
    if (io == OSS_BF16)
        hipLaunchKernelGGL(oss_conv1x1_wgrad_grouped_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, s,
                           reinterpret_cast<const WgradDesc *>(d_descs), reinterpret_cast<const uint16_t *>(d_map));
    else if (io == OSS_F16)
        hipLaunchKernelGGL(oss_conv1x1_wgrad_grouped_kernel<f16_t>, dim3(total_blocks), dim3(256), 0, s,
                           reinterpret_cast<const WgradDesc *>(d_descs), reinterpret_cast<const uint16_t *>(d_map));
    else
        return OSS_ERR_SHAPE;
    return (int)hipGetLastError();
}
void wgrad_desc_set_first_block(void *desc, unsigned first) { reinterpret_cast<WgradDesc *>(desc)->first_block = first; }
int wgrad_desc_io(const void *desc) { return reinterpret_cast<const WgradDesc *>(desc)->io; }

}  // namespace oss
