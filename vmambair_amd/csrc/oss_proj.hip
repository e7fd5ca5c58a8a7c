// oss_proj.hip -- the two in-block projections of the spatial branch of SS2D_1, omni form:
//   x_dbl[b,k,c,l] = sum_d x_proj_weight[k,c,d] xs[b,k,d,l]          (c < R + 2N)
//   dts  [b,k,d,l] = sum_r dt_projs_weight[k,d,r] x_dbl[b,k,r,l]     (the first R rows of x_dbl)
// (SRGAN/VmambaIR/archs/MambaSISR6_arch.py:406-411: two einsums, a split and a .contiguous()), their
// input gradient, and the two time-ordered flattenings the omni scan reads (cross_scan2 / cross_merge2,
// SURVEY.md Appendix B k = 0, 1).  Directions k and k + 2 see the same activations (x2[:, k % 2], the
// scan kernels walk k >= 2 backwards), so one workgroup = 64 time steps of one flattening j and
// produces the rows of BOTH directions j and j + 2 from one read of x2.
//
// HBM-bound byte work with ~70-110 multiply-adds per element: fp32 FMAs on the vector ALU, weights as
// scalar (SGPR) operands straight from the scalar cache -- lane = time step, so every load / store of a
// wave is 64 consecutive elements of one row, and no operand ever crosses lanes.  Output rows
// (forward) / input rows (gradient) are dealt to the waves of the workgroup; nothing is reshaped into
// a GEMM.  Weights are the fp32 master parameters; activations / gradients are the I/O type T.
#include "oss_device.h"
#include "oss_host.h"
#include "oss_mfma.h"

namespace oss {

constexpr int kProjMaxWaves = 16;

// cooperative copy of `rows` x `cols` floats (cols % 4 == 0, 16-byte aligned rows) from a row-major global
// table (row r at src + row_off(r)) into LDS rows of `ld` floats
template <typename RowOff>
__device__ __forceinline__ void stage_rows(float *dst, int ld, const float *src, RowOff row_off, int rows, int cols) {
    const int c4 = cols >> 2;
    for (int i = threadIdx.x; i < rows * c4; i += blockDim.x) {
        const int r = i / c4, c = (i - r * c4) << 2;
        *reinterpret_cast<f32x4 *>(dst + r * ld + c) = *reinterpret_cast<const f32x4 *>(src + row_off(r) + c);
    }
}

// ---------------------------------------------------------------------------------------------
// forward: x2 (B, 2, D, L) -> xdbl (B, 4, C, L), dts (B, 4, D, L);  grid (ceil(L/64), 2, B).
// The 2C x D weight block of the workgroup's two directions goes through LDS in slabs of DC columns
// (it is far larger than the scalar cache, and every wave of the workgroup walks all of it); a wave reads
// its rows' weights as uniform 16-byte LDS loads.  Dynamic LDS: 2C * DC floats.
// ---------------------------------------------------------------------------------------------
template <typename T, int NQ, int RMAX>
__global__ void __launch_bounds__(1024)
oss_proj_fwd_kernel(const T *__restrict__ x2, const float *__restrict__ Wx, const float *__restrict__ Wdt,
                    T *__restrict__ xdbl, T *__restrict__ dts, int D, int C, int R, int L, int DC) {
    extern __shared__ float wl[];        // [2C][DC]
    __shared__ float zl[2 * RMAX * 64];  // the dt rows of both directions, [kk][r][lane]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int j = blockIdx.y, b = blockIdx.z;
    const int p = blockIdx.x * 64 + lane;
    const bool ok = p < L;
    const int pc = ok ? p : L - 1;
    const T *xb = x2 + ((size_t)(b * 2 + j) * D) * L + pc;
    const int rows2 = 2 * C;
    auto wrow = [&](int q) { const int kk = q >= C ? 1 : 0; return (size_t)((j + 2 * kk) * C + (q - kk * C)) * D; };

    float acc[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) acc[i] = 0.f;
    // wave w owns rows q = w, w + nw, ... of the 2C rows (q < C: direction j, else direction j + 2);
    // rows past the end are clamped (computed and dropped) so the FMA loop has no branches
    int qoff[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) qoff[i] = min(wave + i * nw, rows2 - 1) * DC;

    const bool vec = (D % 4 == 0) && (DC % 8 == 0);
    for (int s0 = 0; s0 < D; s0 += DC) {
        const int cols = min(DC, D - s0);
        __syncthreads();  // the previous slab is no longer read
        if (vec && cols % 4 == 0) {
            stage_rows(wl, DC, Wx + s0, wrow, rows2, cols);
        } else {
            for (int i = threadIdx.x; i < rows2 * cols; i += blockDim.x) {
                const int r = i / cols, c = i - r * cols;
                wl[r * DC + c] = Wx[wrow(r) + s0 + c];
            }
        }
        __syncthreads();
        int dl = 0;
        T xraw[8];  // the next chunk of 8 activations, loaded one iteration ahead
        if (cols >= 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) xraw[e] = xb[(size_t)(s0 + e) * L];
        }
        for (; dl + 8 <= cols; dl += 8) {
            float xv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = to_f32(xraw[e]);
            if (dl + 16 <= cols) {
#pragma unroll
                for (int e = 0; e < 8; ++e) xraw[e] = xb[(size_t)(s0 + dl + 8 + e) * L];
            }
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const f32x4 wa = *reinterpret_cast<const f32x4 *>(wl + qoff[i] + dl);
                const f32x4 wb = *reinterpret_cast<const f32x4 *>(wl + qoff[i] + dl + 4);
                acc[i] = __builtin_fmaf(wa.x, xv[0], acc[i]);
                acc[i] = __builtin_fmaf(wa.y, xv[1], acc[i]);
                acc[i] = __builtin_fmaf(wa.z, xv[2], acc[i]);
                acc[i] = __builtin_fmaf(wa.w, xv[3], acc[i]);
                acc[i] = __builtin_fmaf(wb.x, xv[4], acc[i]);
                acc[i] = __builtin_fmaf(wb.y, xv[5], acc[i]);
                acc[i] = __builtin_fmaf(wb.z, xv[6], acc[i]);
                acc[i] = __builtin_fmaf(wb.w, xv[7], acc[i]);
            }
        }
        for (; dl < cols; ++dl) {
            const float xv = to_f32(xb[(size_t)(s0 + dl) * L]);
#pragma unroll
            for (int i = 0; i < NQ; ++i) acc[i] = __builtin_fmaf(wl[qoff[i] + dl], xv, acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = wave + i * nw;
        if (q < rows2) {
            const int kk = q >= C ? 1 : 0, c = q - kk * C, k = j + 2 * kk;
            const T tv = from_f32<T>(acc[i]);
            if (ok) xdbl[((size_t)(b * 4 + k) * C + c) * L + p] = tv;
            if (c < R) zl[(kk * RMAX + c) * 64 + lane] = to_f32(tv);  // dts is computed from the ROUNDED x_dbl
        }
    }
    __syncthreads();
    float zr[2][RMAX];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int r = 0; r < RMAX; ++r) zr[kk][r] = r < R ? zl[(kk * RMAX + r) * 64 + lane] : 0.f;
    for (int d = wave; d < D; d += nw) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int k = j + 2 * kk;
            const float *wr = Wdt + ((size_t)k * D + d) * R;
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < RMAX; ++r)
                if (r < R) s = __builtin_fmaf(wr[r], zr[kk][r], s);
            if (ok) dts[((size_t)(b * 4 + k) * D + d) * L + p] = from_f32<T>(s);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// input gradient.  ddts (B, 4, D, L); dxdbl (B, 4, C, L) with the B / C rows (c >= R) already holding
// dB / dC of the scan backward -- this kernel fills the dt rows:
//   dxdbl[b,k,r,l] = sum_d dt_w[k,d,r] ddts[b,k,d,l]
//   dx2[b,j,d,l]   = sum_{k in {j, j+2}} ( sum_c x_w[k,c,d] dxdbl[b,k,c,l] + du[b,k,d,l] )
// du (B, 4, D, L) = the scan's own gradient w.r.t. u (NULL: none).  Dynamic LDS: proj_dgrad_lds_bytes().
// ---------------------------------------------------------------------------------------------
template <typename T, int DS, int RMAX, int MAXT>
__global__ void __launch_bounds__(MAXT)
oss_proj_dgrad_kernel(const T *__restrict__ ddts, T *__restrict__ dxdbl, const T *__restrict__ du,
                      const float *__restrict__ Wx, const float *__restrict__ Wdt, T *__restrict__ dx2, int D, int C, int R,
                      int L, int slice, int QC) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int j = blockIdx.y, b = blockIdx.z;
    const int p = blockIdx.x * 64 + lane;
    const bool ok = p < L;
    const int pc = ok ? p : L - 1;
    float *v = sm;                          // [2C][64]: dxdbl rows of both directions, this lane's time step
    float *red = sm + (size_t)2 * C * 64;   // [nw][R][64]; dead after phase A, then the weight slab lives here

    // A: dt rows.  Wave w sums over d = w, w + nw, ...; the waves are combined through LDS in a fixed order.
    float pa[2][RMAX];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int r = 0; r < RMAX; ++r) pa[kk][r] = 0.f;
    for (int d = wave; d < D; d += nw) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int k = j + 2 * kk;
            const float g = to_f32(ddts[((size_t)(b * 4 + k) * D + d) * L + pc]);
            const float *wr = Wdt + ((size_t)k * D + d) * R;
#pragma unroll
            for (int r = 0; r < RMAX; ++r)
                if (r < R) pa[kk][r] = __builtin_fmaf(wr[r], g, pa[kk][r]);
        }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = j + 2 * kk;
#pragma unroll
        for (int r = 0; r < RMAX; ++r)
            if (r < R) red[(wave * R + r) * 64 + lane] = pa[kk][r];
        __syncthreads();
        for (int r = wave; r < R; r += nw) {
            float s = 0.f;
            for (int w2 = 0; w2 < nw; ++w2) s += red[(w2 * R + r) * 64 + lane];
            const T tv = from_f32<T>(s);
            v[(kk * C + r) * 64 + lane] = to_f32(tv);
            if (ok) dxdbl[((size_t)(b * 4 + k) * C + r) * L + p] = tv;
        }
        __syncthreads();
    }
    // B: the dB / dC rows
    for (int q = wave; q < 2 * C; q += nw) {
        const int kk = q >= C ? 1 : 0, c = q - kk * C;
        if (c >= R) v[q * 64 + lane] = to_f32(dxdbl[((size_t)(b * 4 + j + 2 * kk) * C + c) * L + pc]);
    }
    __syncthreads();
    // C: wave w owns the rows d in [w * slice, (w + 1) * slice) of dx2.  The weight rows go through LDS in slabs
    // of QC rows x D (aliasing the dead reduction buffer), read back as uniform 16-byte loads.
    const int dbeg = wave * slice;
    float acc[DS];
#pragma unroll
    for (int i = 0; i < DS; ++i) acc[i] = 0.f;
    auto wrow = [&](int q) { const int kk = q >= C ? 1 : 0; return (size_t)((j + 2 * kk) * C + (q - kk * C)) * D; };
    float *wl = v + (size_t)2 * C * 64;  // [QC][D]
    const bool full = dbeg + DS <= D && slice == DS && (D % 4 == 0);
    for (int q0 = 0; q0 < 2 * C; q0 += QC) {
        const int nq = min(QC, 2 * C - q0);
        __syncthreads();
        if (D % 4 == 0) {
            stage_rows(wl, D, Wx, [&](int r) { return wrow(q0 + r); }, nq, D);
        } else {
            for (int i = threadIdx.x; i < nq * D; i += blockDim.x) {
                const int r = i / D, c = i - r * D;
                wl[r * D + c] = Wx[wrow(q0 + r) + c];
            }
        }
        __syncthreads();
        if (full) {
            for (int q = 0; q < nq; ++q) {
                const float val = v[(q0 + q) * 64 + lane];
                const float *wr = wl + q * D + dbeg;
#pragma unroll
                for (int i = 0; i < DS; i += 4) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wr + i);
                    acc[i] = __builtin_fmaf(w4.x, val, acc[i]);
                    acc[i + 1] = __builtin_fmaf(w4.y, val, acc[i + 1]);
                    acc[i + 2] = __builtin_fmaf(w4.z, val, acc[i + 2]);
                    acc[i + 3] = __builtin_fmaf(w4.w, val, acc[i + 3]);
                }
            }
        } else if (dbeg < D) {
            for (int q = 0; q < nq; ++q) {
                const float val = v[(q0 + q) * 64 + lane];
                const float *wr = wl + q * D;
#pragma unroll
                for (int i = 0; i < DS; ++i)
                    if (i < slice && dbeg + i < D) acc[i] = __builtin_fmaf(wr[dbeg + i], val, acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < DS; ++i) {
        const int d = dbeg + i;
        if (i < slice && d < D && ok) {
            float s = acc[i];
            if (du) s += to_f32(du[((size_t)(b * 4 + j) * D + d) * L + p]) + to_f32(du[((size_t)(b * 4 + j + 2) * D + d) * L + p]);
            dx2[((size_t)(b * 2 + j) * D + d) * L + p] = from_f32<T>(s);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 16-bit I/O: the same two products on the matrix cores (v_mfma_f32_32x32x16), one wave = 32 time steps.
//   forward : xdbl[b, k(q), c(q), p] = sum_d Wx[q, d] x2[b, j, d, p]       rows q of the 2C (directions j, j + 2)
//   gradient: dx2[b, j, d, p] = sum_q Wx[q, d] dxdbl[b, k(q), c(q), p] + du[b, j, d, p] + du[b, j + 2, d, p]
// The activation operand needs 8 rows of ONE time step per lane while rows are stored time-contiguous:
// 8 two-byte loads per lane and k-step, each coalesced over the 32 lanes of a half wave.  They are loaded
// once per wave for the whole contraction (KS k-steps, in registers) and reused for every 32-row tile of
// the output; the fp32 master weights come out of L1/L2 and are narrowed on the fly (v_cvt_pk).
// grid (ceil(L / 128), 2 * splits, B): 4 waves = 4 x 32 time steps; `per` output tiles per workgroup.
// ---------------------------------------------------------------------------------------------
// One wave = 64 time steps as two interleaved MFMA column tiles (even steps p0 + 2c / odd steps p0 + 2c + 1):
// one 4-byte load per lane and row feeds both, results leave as 4-byte stores (L even; see oss_conv1x1.hip).
template <typename T, int KS>
__global__ void __launch_bounds__(256)
oss_proj_fwd_mfma_kernel(const T *__restrict__ x2, const float *__restrict__ Wx, T *__restrict__ xdbl, int D, int C, int L,
                         int per) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.y & 1, split = blockIdx.y >> 1, b = blockIdx.z;
    const int p0 = (blockIdx.x * 4 + wave) * 64;
    if (p0 >= L) return;
    const int col = lane & 31, kg = lane >> 5;
    const int p = p0 + 2 * col;
    const bool pok = p < L;
    const uint32_t *xw = reinterpret_cast<const uint32_t *>(x2 + ((size_t)(b * 2 + j) * D) * L + (pok ? p : 0));
    const int lw = L >> 1;
    const int ksteps = (D + 15) >> 4;
    s16x8 bfa[KS], bfb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ks * 16 + kg * 8 + e;
            const bool kok = k < D;
            uint32_t v;
            if constexpr (KS <= 12) {   // always-valid address, masked afterwards (no branch per load)
                const uint32_t raw = xw[(size_t)(kok ? k : D - 1) * lw];
                v = (pok && kok) ? raw : 0u;
            } else {                    // the deep-level instantiations (hundreds of loads): branchy form, compiles 5x faster
                v = (pok && kok) ? xw[(size_t)(kok ? k : D - 1) * lw] : 0u;
            }
            bfa[ks][e] = (short)(v & 0xffffu);
            bfb[ks][e] = (short)(v >> 16);
        }
    }
    const int M = 2 * C, mt_total = (M + 31) >> 5;
    const int mt_end = min(mt_total, (split + 1) * per);
    const bool wvec = (D % 8 == 0) && ((reinterpret_cast<uintptr_t>(Wx) & 15u) == 0);
    for (int mt = split * per; mt < mt_end; ++mt) {
        const int q = mt * 32 + col;  // A-operand row of this lane
        const bool qok = q < M;
        const int qc = qok ? q : 0, kk = qc >= C ? 1 : 0;
        const float *wrow = Wx + ((size_t)((j + 2 * kk) * C + (qc - kk * C))) * D;
        // the tile's weight fragments in groups of GS k-steps -- ONE batch of loads (clamped addresses), then the MFMAs:
        // fetched inside the `ks < ksteps` branch every k-step waited for its own loads (a round trip per k-step)
        constexpr int GS = KS < 12 ? KS : 12;
        f32x16 acca, accb;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acca[r] = 0.f; accb[r] = 0.f; }
#pragma unroll
        for (int g0 = 0; g0 < KS; g0 += GS) {
            s16x8 af[GS];
            if (wvec) {
                f32x4 w0[GS], w1[GS];
#pragma unroll
                for (int u = 0; u < GS; ++u) {
                    const int k0 = (g0 + u) * 16 + kg * 8;
                    const float *wp = wrow + (k0 + 8 <= D ? k0 : 0);
                    w0[u] = *reinterpret_cast<const f32x4 *>(wp);
                    w1[u] = *reinterpret_cast<const f32x4 *>(wp + 4);
                }
#pragma unroll
                for (int u = 0; u < GS; ++u) {
                    const s16x8 f = cvt8<T>(w0[u], w1[u]);
                    af[u] = (qok && (g0 + u) * 16 + kg * 8 + 8 <= D) ? f : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
                }
            } else {
#pragma unroll
                for (int u = 0; u < GS; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = (g0 + u) * 16 + kg * 8 + e;
                        const float wv = wrow[k < D ? k : 0];
                        af[u][e] = (qok && k < D) ? to_bits<T>(wv) : (short)0;
                    }
            }
#pragma unroll
            for (int u = 0; u < GS; ++u) {
                if (g0 + u < ksteps) {
                    acca = Mfma<T>::run(af[u], bfa[g0 + u], acca);
                    accb = Mfma<T>::run(af[u], bfb[g0 + u], accb);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (row < M && pok) {
                const int k2 = row >= C ? 1 : 0;
                *reinterpret_cast<uint32_t *>(xdbl + ((size_t)(b * 4 + j + 2 * k2) * C + (row - k2 * C)) * L + p) =
                    pack2<T>(acca[r], accb[r]);
            }
        }
    }
}

template <typename T, int KS>
__global__ void __launch_bounds__(256)
oss_proj_dgrad_mfma_kernel(const T *__restrict__ dxdbl, const T *__restrict__ du, const float *__restrict__ Wx,
                           T *__restrict__ dx2, int D, int C, int L, int per) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.y & 1, split = blockIdx.y >> 1, b = blockIdx.z;
    const int p0 = (blockIdx.x * 4 + wave) * 64;
    if (p0 >= L) return;
    const int col = lane & 31, kg = lane >> 5;
    const int p = p0 + 2 * col;
    const bool pok = p < L;
    const int K = 2 * C, ksteps = (K + 15) >> 4;
    // 32-bit offsets from per-batch / per-flattening bases (one address VGPR per load)
    const uint32_t *zw = reinterpret_cast<const uint32_t *>(dxdbl + ((size_t)(b * 4 + j) * C) * L + (pok ? p : 0));
    const float *wb = Wx + (size_t)j * C * D;              // row q: q D (+ C D for q >= C)
    const uint32_t lw = (uint32_t)L >> 1, CLw = (uint32_t)C * lw, CD = (uint32_t)C * D;
    s16x8 bfa[KS], bfb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ks * 16 + kg * 8 + e;
            const bool kok = k < K;
            const uint32_t q = kok ? k : 0;
            const uint32_t raw = zw[q * lw + (q >= (uint32_t)C ? CLw : 0u)];  // row q (+ C L: direction j + 2); always valid
            const uint32_t v = (pok && kok) ? raw : 0u;
            bfa[ks][e] = (short)(v & 0xffffu);
            bfb[ks][e] = (short)(v >> 16);
        }
    }
    const int mt_total = (D + 31) >> 5;
    const int mt_end = min(mt_total, (split + 1) * per);
#pragma unroll 1
    for (int mt = split * per; mt < mt_end; ++mt) {
        const int d = mt * 32 + col;  // A-operand row (a row of dx2) of this lane
        const bool dok = d < D;
        const uint32_t dc = dok ? d : 0;
        // operand loads of the tile in two groups, no per-load condition (see oss_conv1x1.hip: fetched k-step by k-step and,
        // in the epilogue, row by row inside `if (row < D && pok)`, every one of them was its own memory round trip):
        // (1) the du words of the 16 output rows, needed last, issued first; (2) all weight fragments
        uint32_t ua[16], ub[16];
        if (du) {
            const uint32_t *du0 = reinterpret_cast<const uint32_t *>(du + ((size_t)(b * 4 + j) * D) * L + (pok ? p : 0));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t o = (size_t)min(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg, D - 1) * lw;
                ua[r] = du0[o];
                ub[r] = du0[(size_t)D * lw * 2 + o];  // direction j + 2: 2 D rows further
            }
        }
        f32x16 acca, accb;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acca[r] = 0.f; accb[r] = 0.f; }
        constexpr int GS = KS < 8 ? KS : 8;   // k-steps whose weight loads form one batch
#pragma unroll
        for (int g0 = 0; g0 < KS; g0 += GS) {
            float wv[GS][8];
#pragma unroll
            for (int u = 0; u < GS; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {  // W^T: 32 consecutive d per k -> 128-byte rows
                    const int k = (g0 + u) * 16 + kg * 8 + e;
                    const uint32_t q = k < K ? k : 0;
                    wv[u][e] = wb[q * (uint32_t)D + (q >= (uint32_t)C ? CD : 0u) + dc];
                }
#pragma unroll
            for (int u = 0; u < GS; ++u) {
                if (g0 + u < ksteps) {
                    float m[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) m[e] = (dok && (g0 + u) * 16 + kg * 8 + e < K) ? wv[u][e] : 0.f;
                    const s16x8 af = cvt8<T>(f32x4{m[0], m[1], m[2], m[3]}, f32x4{m[4], m[5], m[6], m[7]});
                    acca = Mfma<T>::run(af, bfa[g0 + u], acca);
                    accb = Mfma<T>::run(af, bfb[g0 + u], accb);
                }
            }
        }
        uint32_t *ob = reinterpret_cast<uint32_t *>(dx2 + ((size_t)(b * 2 + j) * D) * L + (pok ? p : 0));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            float va = acca[r], vb = accb[r];
            if (du) {
                float a0, a1, b0, b1;
                unpack2<T>(ua[r], a0, a1);
                unpack2<T>(ub[r], b0, b1);
                va += a0 + b0;
                vb += a1 + b1;
            }
            if (row < D && pok) ob[(size_t)row * lw] = pack2<T>(va, vb);
        }
    }
}

// value of lane `src` (a compile-time index after unrolling) in every lane: v_readlane_b32, no memory traffic
__device__ __forceinline__ float lane_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// dts[b, k, d, p] = sum_r Wdt[k, d, r] xdbl[b, k, r, p]   (dt_proj on the ROUNDED dt rows; write-bound: the
// (B, 4D, L) result is the largest tensor of the block).  grid (ceil(L / (64 V)), 4, B); lane = V consecutive time
// steps; wave w owns d = w, w + nw, ...
template <typename T, int RMAX, int V>
__global__ void __launch_bounds__(1024)
oss_dt_fwd_kernel(const T *__restrict__ xdbl, const float *__restrict__ Wdt, T *__restrict__ dts, int D, int C, int R, int L) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int k = blockIdx.y, b = blockIdx.z;
    const int p = (blockIdx.x * 64 + lane) * V;
    const bool ok = p < L;  // V == 2 only with L even: the pair is in range as a whole
    const int pc = ok ? p : 0;
    float zr[RMAX][V];
    // clamped row index, value masked afterwards: `if (r < R) load` is a wave-uniform branch per load, and hipcc closes
    // each with s_waitcnt vmcnt(0) -- R serialised round trips before the first output (rule 2)
#pragma unroll
    for (int r = 0; r < RMAX; ++r) load_v<T, V>(xdbl + ((size_t)(b * 4 + k) * C + min(r, R - 1)) * L + pc, zr[r]);
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
#pragma unroll
        for (int i = 0; i < V; ++i) zr[r][i] = r < R ? zr[r][i] : 0.f;
    // a channel's R weights: ONE vector load (lane r holds w[r], zero past R), the next channel's row in flight during the
    // current one; the values come out with v_readlane.  (R scalar loads per channel, each waited for, made the R = 24
    // instantiation latency-bound: 27 us for a 13 MB problem.)
    const int rl = min(lane, R - 1);
    auto wrow = [&](int d) { return lane < R ? Wdt[((size_t)k * D + min(d, D - 1)) * R + rl] : 0.f; };
    float wnext = wrow(wave);
    for (int d = wave; d < D; d += nw) {
        const float wl = wnext;
        wnext = wrow(d + nw);
        float s[V];
#pragma unroll
        for (int i = 0; i < V; ++i) s[i] = 0.f;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const float wv = lane_bcast(wl, r);
#pragma unroll
            for (int i = 0; i < V; ++i) s[i] = __builtin_fmaf(wv, zr[r][i], s[i]);
        }
        if (ok) store_v<T, V>(dts + ((size_t)(b * 4 + k) * D + d) * L + p, s);
    }
}

// dxdbl[b, k, r, p] = sum_d Wdt[k, d, r] ddts[b, k, d, p]  (r < R; the other rows of dxdbl are not touched).
// Wave w sums d = w, w + nw, ...; the waves are combined through LDS in a fixed order.  LDS: nw * R * 64 V floats.
template <typename T, int RMAX, int V>
__global__ void __launch_bounds__(1024)
oss_dt_dgrad_kernel(const T *__restrict__ ddts, const float *__restrict__ Wdt, T *__restrict__ dxdbl, int D, int C, int R, int L) {
    extern __shared__ float red[];  // [nw][R][64 V]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int k = blockIdx.y, b = blockIdx.z;
    const int p = (blockIdx.x * 64 + lane) * V;
    const bool ok = p < L;
    const int pc = ok ? p : 0;
    float pa[RMAX][V];
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
#pragma unroll
        for (int i = 0; i < V; ++i) pa[r][i] = 0.f;
    // four rows of ddts in flight per pass (one load, one wait per row was a round trip per row); rows past D: clamped
    // address, zero weight
    constexpr int DU = 4;
    for (int d0 = wave; d0 < D; d0 += DU * nw) {
        float g[DU][V];
#pragma unroll
        for (int u = 0; u < DU; ++u) load_v<T, V>(ddts + ((size_t)(b * 4 + k) * D + min(d0 + u * nw, D - 1)) * L + pc, g[u]);
        float wl[DU];   // the rows' weight vectors (lane r holds w[r]; zero past R and for rows past D), see oss_dt_fwd_kernel
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int d = d0 + u * nw;
            wl[u] = (lane < R && d < D) ? Wdt[((size_t)k * D + min(d, D - 1)) * R + min(lane, R - 1)] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < DU; ++u) {
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
                const float wv = lane_bcast(wl[u], r);
#pragma unroll
                for (int i = 0; i < V; ++i) pa[r][i] = __builtin_fmaf(wv, g[u][i], pa[r][i]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
        if (r < R) {
#pragma unroll
            for (int i = 0; i < V; ++i) red[((wave * R + r) * V + i) * 64 + lane] = pa[r][i];
        }
    __syncthreads();
    for (int r = wave; r < R; r += nw) {
        float s[V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
            s[i] = 0.f;
            for (int w2 = 0; w2 < nw; ++w2) s[i] += red[((w2 * R + r) * V + i) * 64 + lane];
        }
        if (ok) store_v<T, V>(dxdbl + ((size_t)(b * 4 + k) * C + r) * L + p, s);
    }
}

// ---------------------------------------------------------------------------------------------
// cross_scan2: x (B, D, H, W) of type TI (contiguous planes, strides (xsb, xsc)) ->
//   x2[b,0,d, h W + w] = x[b,d,h,w]   (row-major flattening,    direction 0 and, walked backwards, 2)
//   x2[b,1,d, w H + h] = x[b,d,h,w]   (column-major flattening, direction 1 and 3)
// One 32 x 32 tile of a plane per workgroup, transposed through LDS: reads and writes coalesced.
// cross_merge2 is its adjoint: dx[b,d,h,w] = g2[b,0,d,hW+w] + g2[b,1,d,wH+h] (sum in fp32, one rounding).
// ---------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void __launch_bounds__(256)
oss_cross_scan2_kernel(const TI *__restrict__ x, TO *__restrict__ x2, int D, int H, int W, int64_t xsb, int64_t xsc) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int tiles_w = (W + 31) >> 5;
    const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
    const int d = blockIdx.y, b = blockIdx.z;
    const TI *xp = x + b * xsb + d * xsc;
    const size_t L = (size_t)H * W;
    TO *o0 = x2 + ((size_t)(b * 2 + 0) * D + d) * L;
    TO *o1 = x2 + ((size_t)(b * 2 + 1) * D + d) * L;
    float val[4];   // the four loads first (clamped addresses): load / store pairs under `if` were four serial round trips
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int h = min(th * 32 + ty + 8 * r, H - 1), w = min(tw * 32 + tx, W - 1);
        val[r] = to_f32(xp[(size_t)h * W + w]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int h = th * 32 + ty + 8 * r, w = tw * 32 + tx;
        if (h < H && w < W) {
            o0[(size_t)h * W + w] = from_f32<TO>(val[r]);
            tile[ty + 8 * r][tx] = val[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int w = tw * 32 + ty + 8 * r, h = th * 32 + tx;
        if (h < H && w < W) o1[(size_t)w * H + h] = from_f32<TO>(tile[tx][ty + 8 * r]);
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
oss_cross_merge2_kernel(const T *__restrict__ g2, T *__restrict__ dx, int D, int H, int W) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int tiles_w = (W + 31) >> 5;
    const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
    const int d = blockIdx.y, b = blockIdx.z;
    const size_t L = (size_t)H * W;
    const T *g0 = g2 + ((size_t)(b * 2 + 0) * D + d) * L;
    const T *g1 = g2 + ((size_t)(b * 2 + 1) * D + d) * L;
    T *o = dx + ((size_t)b * D + d) * L;
    float v1[4], v0[4];   // all eight loads first (clamped addresses), see oss_cross_scan2_kernel
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int w = min(tw * 32 + ty + 8 * r, W - 1), h = min(th * 32 + tx, H - 1);
        v1[r] = to_f32(g1[(size_t)w * H + h]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int h = min(th * 32 + ty + 8 * r, H - 1), w = min(tw * 32 + tx, W - 1);
        v0[r] = to_f32(g0[(size_t)h * W + w]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[tx][ty + 8 * r] = v1[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int h = th * 32 + ty + 8 * r, w = tw * 32 + tx;
        if (h < H && w < W) o[(size_t)h * W + w] = from_f32<T>(v0[r] + tile[ty + 8 * r][tx]);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int g_proj_force_valu = 0;  // tests: run 16-bit I/O through the vector-ALU kernels too
static int proj_waves(int D) { return D <= 192 ? 4 : (D <= 384 ? 8 : 16); }
// input gradient: 24-row slices (24 accumulators + 6 uniform 16-byte weight loads in flight) while 16 waves allow it
static int proj_dgrad_waves(int D) { return D <= 96 ? 4 : (D <= 192 ? 8 : 16); }

// weight rows per LDS slab of the input-gradient kernel (<= 48 KiB)
static int proj_dgrad_qc(int D, int C) { return max(1, min(2 * C, 12288 / D)); }
size_t proj_dgrad_lds_bytes(int D, int C, int R) {
    const size_t red = (size_t)proj_dgrad_waves(D) * R * 64, slab = (size_t)proj_dgrad_qc(D, C) * D;
    return sizeof(float) * (2 * (size_t)C * 64 + (red > slab ? red : slab));
}
// columns per LDS slab of the forward kernel: D split into equal parts of <= 128, multiple of 8
static int proj_fwd_dc(int D) {
    const int parts = (D + 127) / 128;
    return (((D + parts - 1) / parts) + 7) & ~7;
}



template <typename T>
static int proj_fwd_t(const void *x2, const float *Wx, const float *Wdt, void *xdbl, void *dts, int B, int D, int C, int R,
                      int L, hipStream_t s) {
    const int nw = proj_waves(D);
    const int nq = (2 * C + nw - 1) / nw;
    if (nq > 32 || R > 32 || B > 65535) return OSS_ERR_SHAPE;
    dim3 grid((L + 63) / 64, 2, B), block(64 * nw);
    const T *xp = reinterpret_cast<const T *>(x2);
    T *zp = reinterpret_cast<T *>(xdbl), *dp = reinterpret_cast<T *>(dts);
    const int dc = proj_fwd_dc(D);
    const size_t smem = sizeof(float) * 2 * (size_t)C * dc;
#define OSS_PROJ_FWD(NQ_, RM_)                                                                              \
    do {                                                                                                    \
        auto kern = oss_proj_fwd_kernel<T, NQ_, RM_>;                                                       \
        static LdsGate gate;                                                                                \
        if (const int e = gate.ensure(reinterpret_cast<const void *>(kern), smem)) return e;                \
        hipLaunchKernelGGL(kern, grid, block, smem, s, xp, Wx, Wdt, zp, dp, D, C, R, L, dc);               \
    } while (0)
    if (R <= 8) {
        if (nq <= 12) OSS_PROJ_FWD(12, 8); else if (nq <= 20) OSS_PROJ_FWD(20, 8); else OSS_PROJ_FWD(32, 8);
    } else {
        if (nq <= 12) OSS_PROJ_FWD(12, 32); else if (nq <= 20) OSS_PROJ_FWD(20, 32); else OSS_PROJ_FWD(32, 32);
    }
#undef OSS_PROJ_FWD
    return (int)hipGetLastError();
}

template <typename T, int DS, int RMAX, int MAXT>
static int proj_dgrad_launch(const T *ddts, T *dxdbl, const T *du, const float *Wx, const float *Wdt, T *dx2, int B, int D, int C,
                             int R, int L, int nw, int slice, hipStream_t s) {
    const size_t smem = proj_dgrad_lds_bytes(D, C, R);
    auto kern = oss_proj_dgrad_kernel<T, DS, RMAX, MAXT>;
    static LdsGate gate;
    if (const int e = gate.ensure(reinterpret_cast<const void *>(kern), smem)) return e;
    dim3 grid((L + 63) / 64, 2, B), block(64 * nw);
    hipLaunchKernelGGL(kern, grid, block, smem, s, ddts, dxdbl, du, Wx, Wdt, dx2, D, C, R, L, slice, proj_dgrad_qc(D, C));
    return (int)hipGetLastError();
}

template <typename T>
static int proj_dgrad_t(const void *ddts, void *dxdbl, const void *du, const float *Wx, const float *Wdt, void *dx2, int B, int D,
                        int C, int R, int L, hipStream_t s) {
    const int nw = proj_dgrad_waves(D);
    const int slice = (((D + nw - 1) / nw) + 7) & ~7;
    if (slice > 64 || R > 32 || B > 65535) return OSS_ERR_SHAPE;
    const T *gp = reinterpret_cast<const T *>(ddts), *up = reinterpret_cast<const T *>(du);
    T *zp = reinterpret_cast<T *>(dxdbl), *xp = reinterpret_cast<T *>(dx2);
#define OSS_PROJ_DG(DS_, RM_, MT_) return proj_dgrad_launch<T, DS_, RM_, MT_>(gp, zp, up, Wx, Wdt, xp, B, D, C, R, L, nw, slice, s)
    if (R <= 8) {
        if (nw == 4) OSS_PROJ_DG(24, 8, 256);
        if (nw == 8) OSS_PROJ_DG(24, 8, 512);
        if (slice <= 24) OSS_PROJ_DG(24, 8, 1024); else if (slice <= 48) OSS_PROJ_DG(48, 8, 1024); else OSS_PROJ_DG(64, 8, 1024);
    } else {
        if (nw == 4) OSS_PROJ_DG(24, 32, 256);
        if (nw == 8) OSS_PROJ_DG(24, 32, 512);
        if (slice <= 24) OSS_PROJ_DG(24, 32, 1024); else if (slice <= 48) OSS_PROJ_DG(48, 32, 1024); else OSS_PROJ_DG(64, 32, 1024);
    }
#undef OSS_PROJ_DG
}

// output tiles per workgroup: as few activation re-loads as possible once the chip is full (2 waves per SIMD)
static int mfma_tiles_per_wg(int B, int L, int mt) {
    const long waves = 2L * B * ((L + 63) / 64);
    long split = (4096 + waves - 1) / waves;
    if (split < 1) split = 1;
    if (split > mt) split = mt;
    return (int)((mt + split - 1) / split);
}

static int dt_waves(int B, int L, int D) {
    // enough waves to fill the chip at the deep levels (few time steps), 4 per workgroup where there are plenty
    const long wgs = 4L * B * ((L + 127) / 128);
    int nw = 4;
    while (nw < 16 && wgs * nw < 16384 && nw * 8 <= D) nw *= 2;
    return nw;
}

template <typename T>
static int proj_fwd_mfma_t(const void *x2, const float *Wx, const float *Wdt, void *xdbl, void *dts, int B, int D, int C, int R,
                           int L, hipStream_t s) {
    const T *xp = reinterpret_cast<const T *>(x2);
    T *zp = reinterpret_cast<T *>(xdbl), *dp = reinterpret_cast<T *>(dts);
    const int mt = (2 * C + 31) / 32, per = mfma_tiles_per_wg(B, L, mt), splits = (mt + per - 1) / per;
    dim3 grid((L + 255) / 256, 2 * splits, B);
    const int ks = (D + 15) / 16;
    if (ks <= 3)       hipLaunchKernelGGL((oss_proj_fwd_mfma_kernel<T, 3>), grid, dim3(256), 0, s, xp, Wx, zp, D, C, L, per);   // d_inner 48: no duplicate k-steps
    else if (ks <= 6)  hipLaunchKernelGGL((oss_proj_fwd_mfma_kernel<T, 6>), grid, dim3(256), 0, s, xp, Wx, zp, D, C, L, per);
    else if (ks <= 12) hipLaunchKernelGGL((oss_proj_fwd_mfma_kernel<T, 12>), grid, dim3(256), 0, s, xp, Wx, zp, D, C, L, per);
    else if (ks <= 24) hipLaunchKernelGGL((oss_proj_fwd_mfma_kernel<T, 24>), grid, dim3(256), 0, s, xp, Wx, zp, D, C, L, per);
    else               hipLaunchKernelGGL((oss_proj_fwd_mfma_kernel<T, 48>), grid, dim3(256), 0, s, xp, Wx, zp, D, C, L, per);
    if (!dts) return (int)hipGetLastError();   // delta is evaluated inside the scan (oss_scan_fwd_params.dt_weight)
    // eight time steps per lane (16-byte accesses: dts, the (B, 4D, L) result, is what this kernel moves) where rows allow it
    const bool wide = R <= 8 && L % 8 == 0 && L >= 2048 && ((reinterpret_cast<uintptr_t>(zp) | reinterpret_cast<uintptr_t>(dp)) & 15u) == 0;
    if (wide) {
        const int nw = dt_waves(B, L / 4, D);
        dim3 g8((L + 511) / 512, 4, B);
        hipLaunchKernelGGL((oss_dt_fwd_kernel<T, 8, 8>), g8, dim3(64 * nw), 0, s, zp, Wdt, dp, D, C, R, L);
        return (int)hipGetLastError();
    }
    const int nw = dt_waves(B, L, D);
    dim3 g2((L + 127) / 128, 4, B);  // L is even here: two time steps per lane
    if (R <= 8) hipLaunchKernelGGL((oss_dt_fwd_kernel<T, 8, 2>), g2, dim3(64 * nw), 0, s, zp, Wdt, dp, D, C, R, L);
    else        hipLaunchKernelGGL((oss_dt_fwd_kernel<T, 32, 2>), g2, dim3(64 * nw), 0, s, zp, Wdt, dp, D, C, R, L);
    return (int)hipGetLastError();
}

template <typename T>
static int proj_dgrad_mfma_t(const void *ddts, void *dxdbl, const void *du, const float *Wx, const float *Wdt, void *dx2, int B,
                             int D, int C, int R, int L, hipStream_t s) {
    const T *gp = reinterpret_cast<const T *>(ddts), *up = reinterpret_cast<const T *>(du);
    T *zp = reinterpret_cast<T *>(dxdbl), *xp = reinterpret_cast<T *>(dx2);
    int nw = dt_waves(B, L, D);
    while (nw > 1 && sizeof(float) * 128 * (size_t)nw * R > 48 * 1024) nw >>= 1;  // cross-wave reduction buffer <= 48 KiB
    dim3 g1((L + 127) / 128, 4, B);
    const size_t smem = sizeof(float) * 128 * (size_t)nw * R;
    if (!ddts) {}   // the scan backward already filled the dt rows of dxdbl (oss_scan_bwd_params.ddt)
    else if (R <= 8 && L % 8 == 0 && L >= 2048 && ((reinterpret_cast<uintptr_t>(gp) | reinterpret_cast<uintptr_t>(zp)) & 15u) == 0) {
        // eight time steps per lane (16-byte loads of ddts)
        int nw8 = dt_waves(B, L / 4, D);
        while (nw8 > 1 && sizeof(float) * 512 * (size_t)nw8 * R > 48 * 1024) nw8 >>= 1;
        const size_t smem8 = sizeof(float) * 512 * (size_t)nw8 * R;
        hipLaunchKernelGGL((oss_dt_dgrad_kernel<T, 8, 8>), dim3((L + 511) / 512, 4, B), dim3(64 * nw8), smem8, s, gp, Wdt, zp, D, C, R, L);
    }
    else if (R <= 8) hipLaunchKernelGGL((oss_dt_dgrad_kernel<T, 8, 2>), g1, dim3(64 * nw), smem, s, gp, Wdt, zp, D, C, R, L);
    else             hipLaunchKernelGGL((oss_dt_dgrad_kernel<T, 32, 2>), g1, dim3(64 * nw), smem, s, gp, Wdt, zp, D, C, R, L);
    const int mt = (D + 31) / 32, per = mfma_tiles_per_wg(B, L, mt), splits = (mt + per - 1) / per;
    dim3 grid((L + 255) / 256, 2 * splits, B);
    const int ks = (2 * C + 15) / 16;
    if (ks <= 5)      hipLaunchKernelGGL((oss_proj_dgrad_mfma_kernel<T, 5>), grid, dim3(256), 0, s, zp, up, Wx, xp, D, C, L, per);
    else if (ks <= 8) hipLaunchKernelGGL((oss_proj_dgrad_mfma_kernel<T, 8>), grid, dim3(256), 0, s, zp, up, Wx, xp, D, C, L, per);
    else              hipLaunchKernelGGL((oss_proj_dgrad_mfma_kernel<T, 16>), grid, dim3(256), 0, s, zp, up, Wx, xp, D, C, L, per);
    return (int)hipGetLastError();
}

// 16-bit I/O and shapes the register-resident operands cover -> matrix-core path
bool proj_mfma_ok(oss_dtype io, int B, int D, int C, int R, int L) {
    return io != OSS_F32 && L % 2 == 0 && D <= 16 * 48 && 2 * C <= 16 * 16 && R <= 32 && B <= 65535 && g_proj_force_valu == 0;
}

int proj_fwd(oss_dtype io, const void *x2, const float *Wx, const float *Wdt, void *xdbl, void *dts, int B, int D, int C, int R,
             int L, hipStream_t s) {
    if (!dts && !proj_mfma_ok(io, B, D, C, R, L)) return OSS_ERR_NULL;   // dts may be omitted on the matrix-core path only
    if (proj_mfma_ok(io, B, D, C, R, L))
        return io == OSS_BF16 ? proj_fwd_mfma_t<bf16_t>(x2, Wx, Wdt, xdbl, dts, B, D, C, R, L, s)
                              : proj_fwd_mfma_t<f16_t>(x2, Wx, Wdt, xdbl, dts, B, D, C, R, L, s);
    if (io == OSS_F32 && dts && !g_proj_force_valu && proj_f32_ok(B, D, C, R, L, {x2, xdbl, dts}))   // (round 4) fp32 matrix cores
        return proj_fwd_f32(reinterpret_cast<const float *>(x2), Wx, Wdt, reinterpret_cast<float *>(xdbl), reinterpret_cast<float *>(dts),
                            B, D, C, R, L, s);
    switch (io) {
        case OSS_F32: return proj_fwd_t<float>(x2, Wx, Wdt, xdbl, dts, B, D, C, R, L, s);
        case OSS_F16: return proj_fwd_t<f16_t>(x2, Wx, Wdt, xdbl, dts, B, D, C, R, L, s);
        case OSS_BF16: return proj_fwd_t<bf16_t>(x2, Wx, Wdt, xdbl, dts, B, D, C, R, L, s);
    }
    return OSS_ERR_SHAPE;
}

int proj_dgrad(oss_dtype io, const void *ddts, void *dxdbl, const void *du, const float *Wx, const float *Wdt, void *dx2, int B,
               int D, int C, int R, int L, hipStream_t s) {
    if (!ddts && !proj_mfma_ok(io, B, D, C, R, L)) return OSS_ERR_NULL;
    if (proj_mfma_ok(io, B, D, C, R, L))
        return io == OSS_BF16 ? proj_dgrad_mfma_t<bf16_t>(ddts, dxdbl, du, Wx, Wdt, dx2, B, D, C, R, L, s)
                              : proj_dgrad_mfma_t<f16_t>(ddts, dxdbl, du, Wx, Wdt, dx2, B, D, C, R, L, s);
    if (io == OSS_F32 && ddts && !g_proj_force_valu && proj_f32_ok(B, D, C, R, L, {ddts, dxdbl, du, dx2}))
        return proj_dgrad_f32(reinterpret_cast<const float *>(ddts), reinterpret_cast<float *>(dxdbl), reinterpret_cast<const float *>(du),
                              Wx, Wdt, reinterpret_cast<float *>(dx2), B, D, C, R, L, s);
    switch (io) {
        case OSS_F32: return proj_dgrad_t<float>(ddts, dxdbl, du, Wx, Wdt, dx2, B, D, C, R, L, s);
        case OSS_F16: return proj_dgrad_t<f16_t>(ddts, dxdbl, du, Wx, Wdt, dx2, B, D, C, R, L, s);
        case OSS_BF16: return proj_dgrad_t<bf16_t>(ddts, dxdbl, du, Wx, Wdt, dx2, B, D, C, R, L, s);
    }
    return OSS_ERR_SHAPE;
}

void proj_force_valu(int on) { g_proj_force_valu = on ? 1 : 0; }

// 16-bit output, even H and W >= 64: the same two kernels on 64 x 64 tiles with TWO adjacent elements per lane on the contiguous
// axis of every access (4-byte accesses; a wave's 2-byte accesses move 128 bytes per instruction -- DESIGN.md 4.4 rule 1).
// The transposed side pairs two consecutive h: lane hx reads tile rows 2 hx and 2 hx + 1 (row stride 65 floats: conflict-free).
template <typename TI, typename TO>
__global__ void __launch_bounds__(256)
oss_cross_scan2_pair_kernel(const TI *__restrict__ x, TO *__restrict__ x2, int D, int H, int W, int64_t xsb, int64_t xsc) {
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 pairs x 8
    const int tiles_w = (W + 63) >> 6;
    const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
    const int d = blockIdx.y, b = blockIdx.z;
    const TI *xp = x + b * xsb + d * xsc;
    const size_t L = (size_t)H * W;
    TO *o0 = x2 + ((size_t)(b * 2 + 0) * D + d) * L;
    TO *o1 = x2 + ((size_t)(b * 2 + 1) * D + d) * L;
    float val[8][2];
    const int w = tw * 64 + 2 * tx, wc = min(w, W - 2);
#pragma unroll
    for (int r = 0; r < 8; ++r) load_v<TI, 2>(xp + (size_t)min(th * 64 + ty + 8 * r, H - 1) * W + wc, val[r]);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int h = th * 64 + ty + 8 * r;
        tile[ty + 8 * r][2 * tx] = val[r][0];
        tile[ty + 8 * r][2 * tx + 1] = val[r][1];
        if (h < H && w < W) store_v<TO, 2>(o0 + (size_t)h * W + w, val[r]);
    }
    __syncthreads();
    const int h2 = th * 64 + 2 * tx;   // this lane's pair of rows
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int wl = ty + 8 * r, wg = tw * 64 + wl;
        const float v[2] = {tile[2 * tx][wl], tile[2 * tx + 1][wl]};
        if (wg < W && h2 < H) store_v<TO, 2>(o1 + (size_t)wg * H + h2, v);
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
oss_cross_merge2_pair_kernel(const T *__restrict__ g2, T *__restrict__ dx, int D, int H, int W) {
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int tiles_w = (W + 63) >> 6;
    const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
    const int d = blockIdx.y, b = blockIdx.z;
    const size_t L = (size_t)H * W;
    const T *g0 = g2 + ((size_t)(b * 2 + 0) * D + d) * L;
    const T *g1 = g2 + ((size_t)(b * 2 + 1) * D + d) * L;
    T *o = dx + ((size_t)b * D + d) * L;
    float v1[8][2], v0[8][2];
    const int h2 = th * 64 + 2 * tx, h2c = min(h2, H - 2);
    const int w = tw * 64 + 2 * tx, wc = min(w, W - 2);
#pragma unroll
    for (int r = 0; r < 8; ++r) load_v<T, 2>(g1 + (size_t)min(tw * 64 + ty + 8 * r, W - 1) * H + h2c, v1[r]);
#pragma unroll
    for (int r = 0; r < 8; ++r) load_v<T, 2>(g0 + (size_t)min(th * 64 + ty + 8 * r, H - 1) * W + wc, v0[r]);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        tile[2 * tx][ty + 8 * r] = v1[r][0];
        tile[2 * tx + 1][ty + 8 * r] = v1[r][1];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int hl = ty + 8 * r, h = th * 64 + hl;
        const float v[2] = {v0[r][0] + tile[hl][2 * tx], v0[r][1] + tile[hl][2 * tx + 1]};
        if (h < H && w < W) store_v<T, 2>(o + (size_t)h * W + w, v);
    }
}

template <typename TI, typename TO>
static int cross_scan2_t(const void *x, void *x2, int B, int D, int H, int W, int64_t xsb, int64_t xsc, hipStream_t s) {
    if constexpr (sizeof(TO) == 2) {
        const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(x2);
        if (H >= 64 && W >= 64 && H % 2 == 0 && W % 2 == 0 && xsb % 2 == 0 && xsc % 2 == 0 && (al & (sizeof(TI) == 4 ? 7u : 3u)) == 0) {
            dim3 grid64(((H + 63) / 64) * ((W + 63) / 64), D, B);
            hipLaunchKernelGGL((oss_cross_scan2_pair_kernel<TI, TO>), grid64, dim3(256), 0, s, reinterpret_cast<const TI *>(x),
                               reinterpret_cast<TO *>(x2), D, H, W, xsb, xsc);
            return (int)hipGetLastError();
        }
    }
    dim3 grid(((H + 31) / 32) * ((W + 31) / 32), D, B);
    hipLaunchKernelGGL((oss_cross_scan2_kernel<TI, TO>), grid, dim3(256), 0, s, reinterpret_cast<const TI *>(x),
                       reinterpret_cast<TO *>(x2), D, H, W, xsb, xsc);
    return (int)hipGetLastError();
}

int cross_scan2(oss_dtype it, oss_dtype ot, const void *x, void *x2, int B, int D, int H, int W, int64_t xsb, int64_t xsc,
                hipStream_t s) {
    if (B > 65535 || D > 65535) return OSS_ERR_SHAPE;
    switch ((int)it * 3 + (int)ot) {
        case 0: return cross_scan2_t<float, float>(x, x2, B, D, H, W, xsb, xsc, s);
        case 1: return cross_scan2_t<float, f16_t>(x, x2, B, D, H, W, xsb, xsc, s);
        case 2: return cross_scan2_t<float, bf16_t>(x, x2, B, D, H, W, xsb, xsc, s);
        case 4: return cross_scan2_t<f16_t, f16_t>(x, x2, B, D, H, W, xsb, xsc, s);
        case 8: return cross_scan2_t<bf16_t, bf16_t>(x, x2, B, D, H, W, xsb, xsc, s);
        default: return OSS_ERR_SHAPE;
    }
}

int cross_merge2(oss_dtype io, const void *g2, void *dx, int B, int D, int H, int W, hipStream_t s) {
    if (B > 65535 || D > 65535) return OSS_ERR_SHAPE;
    if (io != OSS_F32 && H >= 64 && W >= 64 && H % 2 == 0 && W % 2 == 0 &&
        ((reinterpret_cast<uintptr_t>(g2) | reinterpret_cast<uintptr_t>(dx)) & 3u) == 0) {
        dim3 grid64(((H + 63) / 64) * ((W + 63) / 64), D, B);
        if (io == OSS_F16) hipLaunchKernelGGL(oss_cross_merge2_pair_kernel<f16_t>, grid64, dim3(256), 0, s, reinterpret_cast<const f16_t *>(g2), reinterpret_cast<f16_t *>(dx), D, H, W);
        else               hipLaunchKernelGGL(oss_cross_merge2_pair_kernel<bf16_t>, grid64, dim3(256), 0, s, reinterpret_cast<const bf16_t *>(g2), reinterpret_cast<bf16_t *>(dx), D, H, W);
        return (int)hipGetLastError();
    }
    dim3 grid(((H + 31) / 32) * ((W + 31) / 32), D, B);
    switch (io) {
        case OSS_F32:
            hipLaunchKernelGGL(oss_cross_merge2_kernel<float>, grid, dim3(256), 0, s, reinterpret_cast<const float *>(g2),
                               reinterpret_cast<float *>(dx), D, H, W);
            break;
        case OSS_F16:
            hipLaunchKernelGGL(oss_cross_merge2_kernel<f16_t>, grid, dim3(256), 0, s, reinterpret_cast<const f16_t *>(g2),
                               reinterpret_cast<f16_t *>(dx), D, H, W);
            break;
        case OSS_BF16:
            hipLaunchKernelGGL(oss_cross_merge2_kernel<bf16_t>, grid, dim3(256), 0, s, reinterpret_cast<const bf16_t *>(g2),
                               reinterpret_cast<bf16_t *>(dx), D, H, W);
            break;
        default: return OSS_ERR_SHAPE;
    }
    return (int)hipGetLastError();
}

}  // namespace oss
