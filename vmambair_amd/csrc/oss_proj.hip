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

namespace oss {

constexpr int kProjMaxWaves = 16;

// acc_r += sum_e W[off_r + e] * x[e] for four rows r of a wave-uniform fp32 table and eight lane values x:
// the 4 x 8 weights go through the scalar cache into SGPRs and are consumed as SGPR FMA operands.  Written
// as one asm block because the compiler's scheduler otherwise hoists every scalar load of the unrolled
// loop above the FMAs and spills the SGPRs to VGPR lanes.  off_r: BYTE offsets from `base`.
__device__ __forceinline__ void fma_rows4x8(float &a0, float &a1, float &a2, float &a3, const float (&x)[8], const float *base,
                                            uint32_t off0, uint32_t off1, uint32_t off2, uint32_t off3) {
    asm volatile(
        "s_load_dwordx8 s[36:43], %12, %13\n\t"
        "s_load_dwordx8 s[44:51], %12, %14\n\t"
        "s_load_dwordx8 s[52:59], %12, %15\n\t"
        "s_load_dwordx8 s[60:67], %12, %16\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_fmac_f32 %0, s36, %4\n\t"
        "v_fmac_f32 %1, s44, %4\n\t"
        "v_fmac_f32 %2, s52, %4\n\t"
        "v_fmac_f32 %3, s60, %4\n\t"
        "v_fmac_f32 %0, s37, %5\n\t"
        "v_fmac_f32 %1, s45, %5\n\t"
        "v_fmac_f32 %2, s53, %5\n\t"
        "v_fmac_f32 %3, s61, %5\n\t"
        "v_fmac_f32 %0, s38, %6\n\t"
        "v_fmac_f32 %1, s46, %6\n\t"
        "v_fmac_f32 %2, s54, %6\n\t"
        "v_fmac_f32 %3, s62, %6\n\t"
        "v_fmac_f32 %0, s39, %7\n\t"
        "v_fmac_f32 %1, s47, %7\n\t"
        "v_fmac_f32 %2, s55, %7\n\t"
        "v_fmac_f32 %3, s63, %7\n\t"
        "v_fmac_f32 %0, s40, %8\n\t"
        "v_fmac_f32 %1, s48, %8\n\t"
        "v_fmac_f32 %2, s56, %8\n\t"
        "v_fmac_f32 %3, s64, %8\n\t"
        "v_fmac_f32 %0, s41, %9\n\t"
        "v_fmac_f32 %1, s49, %9\n\t"
        "v_fmac_f32 %2, s57, %9\n\t"
        "v_fmac_f32 %3, s65, %9\n\t"
        "v_fmac_f32 %0, s42, %10\n\t"
        "v_fmac_f32 %1, s50, %10\n\t"
        "v_fmac_f32 %2, s58, %10\n\t"
        "v_fmac_f32 %3, s66, %10\n\t"
        "v_fmac_f32 %0, s43, %11\n\t"
        "v_fmac_f32 %1, s51, %11\n\t"
        "v_fmac_f32 %2, s59, %11\n\t"
        "v_fmac_f32 %3, s67, %11\n\t"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "s"(base), "s"(off0), "s"(off1),
          "s"(off2), "s"(off3)
        : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67");
}

// ---------------------------------------------------------------------------------------------
// forward: x2 (B, 2, D, L) -> xdbl (B, 4, C, L), dts (B, 4, D, L);  grid (ceil(L/64), 2, B)
// ---------------------------------------------------------------------------------------------
template <typename T, int NQ, int RMAX>
__global__ void __launch_bounds__(1024)
oss_proj_fwd_kernel(const T *__restrict__ x2, const float *__restrict__ Wx, const float *__restrict__ Wdt,
                    T *__restrict__ xdbl, T *__restrict__ dts, int D, int C, int R, int L) {
    __shared__ float zl[2 * RMAX * 64];  // the dt rows of both directions, [kk][r][lane]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int j = blockIdx.y, b = blockIdx.z;
    const int p = blockIdx.x * 64 + lane;
    const bool ok = p < L;
    const int pc = ok ? p : L - 1;
    const T *xb = x2 + ((size_t)(b * 2 + j) * D) * L + pc;
    const int rows2 = 2 * C;

    float acc[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) acc[i] = 0.f;
    // wave w owns rows q = w, w + nw, ... of the 2C rows (q < C: direction j, else direction j + 2);
    // rows past the end are clamped (computed and dropped) so the FMA loop has no branches
    uint32_t woff[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = min(wave + i * nw, rows2 - 1);
        const int kk = q >= C ? 1 : 0, c = q - kk * C;
        woff[i] = (uint32_t)(((j + 2 * kk) * C + c) * D);
    }
    int d0 = 0;
    T xraw[8];  // the next chunk of 8 activations, loaded one iteration ahead
    if (D >= 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) xraw[e] = xb[(size_t)e * L];
    }
    for (; d0 + 8 <= D; d0 += 8) {
        float xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = to_f32(xraw[e]);
        if (d0 + 16 <= D) {
#pragma unroll
            for (int e = 0; e < 8; ++e) xraw[e] = xb[(size_t)(d0 + 8 + e) * L];
        }
#pragma unroll
        for (int i = 0; i < NQ; i += 4)
            fma_rows4x8(acc[i], acc[i + 1], acc[i + 2], acc[i + 3], xv, Wx, (woff[i] + d0) * 4u, (woff[i + 1] + d0) * 4u,
                        (woff[i + 2] + d0) * 4u, (woff[i + 3] + d0) * 4u);
    }
    for (; d0 < D; ++d0) {
        const float xv = to_f32(xb[(size_t)d0 * L]);
#pragma unroll
        for (int i = 0; i < NQ; ++i) acc[i] = __builtin_fmaf(Wx[(size_t)woff[i] + d0], xv, acc[i]);
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
// du (B, 4, D, L) = the scan's own gradient w.r.t. u (NULL: none).  Dynamic LDS: (nw R + 2C) * 64 floats.
// ---------------------------------------------------------------------------------------------
template <typename T, int DS, int RMAX>
__global__ void __launch_bounds__(1024)
oss_proj_dgrad_kernel(const T *__restrict__ ddts, T *__restrict__ dxdbl, const T *__restrict__ du,
                      const float *__restrict__ Wx, const float *__restrict__ Wdt, T *__restrict__ dx2, int D, int C, int R,
                      int L, int slice) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int j = blockIdx.y, b = blockIdx.z;
    const int p = blockIdx.x * 64 + lane;
    const bool ok = p < L;
    const int pc = ok ? p : L - 1;
    float *red = sm;                       // [nw][R][64]
    float *v = sm + (size_t)nw * R * 64;   // [2C][64]: dxdbl rows of both directions, this lane's time step

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
    // C: wave w owns the rows d in [w * slice, (w + 1) * slice) of dx2
    const int dbeg = wave * slice;
    float acc[DS];
#pragma unroll
    for (int i = 0; i < DS; ++i) acc[i] = 0.f;
    if (dbeg + DS <= D) {  // full slice: unconditional wide scalar loads
        for (int q = 0; q < 2 * C; ++q) {
            const int kk = q >= C ? 1 : 0, c = q - kk * C;
            const float val = v[q * 64 + lane];
            const float *wr = Wx + ((size_t)((j + 2 * kk) * C + c)) * D + dbeg;
#pragma unroll
            for (int i = 0; i < DS; ++i) acc[i] = __builtin_fmaf(wr[i], val, acc[i]);
        }
    } else if (dbeg < D) {
        for (int q = 0; q < 2 * C; ++q) {
            const int kk = q >= C ? 1 : 0, c = q - kk * C;
            const float val = v[q * 64 + lane];
            const float *wr = Wx + ((size_t)((j + 2 * kk) * C + c)) * D;
#pragma unroll
            for (int i = 0; i < DS; ++i)
                if (i < slice && dbeg + i < D) acc[i] = __builtin_fmaf(wr[dbeg + i], val, acc[i]);
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
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int h = th * 32 + ty + 8 * r, w = tw * 32 + tx;
        if (h < H && w < W) {
            const float val = to_f32(xp[(size_t)h * W + w]);
            o0[(size_t)h * W + w] = from_f32<TO>(val);
            tile[ty + 8 * r][tx] = val;
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
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int w = tw * 32 + ty + 8 * r, h = th * 32 + tx;
        if (h < H && w < W) tile[tx][ty + 8 * r] = to_f32(g1[(size_t)w * H + h]);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int h = th * 32 + ty + 8 * r, w = tw * 32 + tx;
        if (h < H && w < W) o[(size_t)h * W + w] = from_f32<T>(to_f32(g0[(size_t)h * W + w]) + tile[ty + 8 * r][tx]);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int proj_waves(int D) { return D <= 192 ? 4 : (D <= 384 ? 8 : 16); }

size_t proj_dgrad_lds_bytes(int D, int C, int R) { return sizeof(float) * 64 * ((size_t)proj_waves(D) * R + 2 * (size_t)C); }

template <typename K>
static int enable_lds(K kern, size_t bytes) {
    if (bytes <= 48 * 1024) return 0;
    if (bytes > 160 * 1024) return OSS_ERR_SHAPE;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
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
#define OSS_PROJ_FWD(NQ_, RM_) hipLaunchKernelGGL((oss_proj_fwd_kernel<T, NQ_, RM_>), grid, block, 0, s, xp, Wx, Wdt, zp, dp, D, C, R, L)
    if (R <= 8) {
        if (nq <= 12) OSS_PROJ_FWD(12, 8); else if (nq <= 20) OSS_PROJ_FWD(20, 8); else OSS_PROJ_FWD(32, 8);
    } else {
        if (nq <= 12) OSS_PROJ_FWD(12, 32); else if (nq <= 20) OSS_PROJ_FWD(20, 32); else OSS_PROJ_FWD(32, 32);
    }
#undef OSS_PROJ_FWD
    return (int)hipGetLastError();
}

template <typename T, int DS, int RMAX>
static int proj_dgrad_launch(const T *ddts, T *dxdbl, const T *du, const float *Wx, const float *Wdt, T *dx2, int B, int D, int C,
                             int R, int L, int nw, int slice, hipStream_t s) {
    const size_t smem = proj_dgrad_lds_bytes(D, C, R);
    auto kern = oss_proj_dgrad_kernel<T, DS, RMAX>;
    static size_t enabled = 48 * 1024;
    if (smem > enabled) {
        const int e = enable_lds(kern, smem);
        if (e) return e;
        enabled = smem;
    }
    dim3 grid((L + 63) / 64, 2, B), block(64 * nw);
    hipLaunchKernelGGL(kern, grid, block, smem, s, ddts, dxdbl, du, Wx, Wdt, dx2, D, C, R, L, slice);
    return (int)hipGetLastError();
}

template <typename T>
static int proj_dgrad_t(const void *ddts, void *dxdbl, const void *du, const float *Wx, const float *Wdt, void *dx2, int B, int D,
                        int C, int R, int L, hipStream_t s) {
    const int nw = proj_waves(D);
    const int slice = (((D + nw - 1) / nw) + 7) & ~7;
    if (slice > 64 || R > 32 || B > 65535) return OSS_ERR_SHAPE;
    const T *gp = reinterpret_cast<const T *>(ddts), *up = reinterpret_cast<const T *>(du);
    T *zp = reinterpret_cast<T *>(dxdbl), *xp = reinterpret_cast<T *>(dx2);
#define OSS_PROJ_DG(DS_, RM_) return proj_dgrad_launch<T, DS_, RM_>(gp, zp, up, Wx, Wdt, xp, B, D, C, R, L, nw, slice, s)
    if (R <= 8) {
        if (slice <= 24) OSS_PROJ_DG(24, 8); else if (slice <= 48) OSS_PROJ_DG(48, 8); else OSS_PROJ_DG(64, 8);
    } else {
        if (slice <= 24) OSS_PROJ_DG(24, 32); else if (slice <= 48) OSS_PROJ_DG(48, 32); else OSS_PROJ_DG(64, 32);
    }
#undef OSS_PROJ_DG
}

int proj_fwd(oss_dtype io, const void *x2, const float *Wx, const float *Wdt, void *xdbl, void *dts, int B, int D, int C, int R,
             int L, hipStream_t s) {
    switch (io) {
        case OSS_F32: return proj_fwd_t<float>(x2, Wx, Wdt, xdbl, dts, B, D, C, R, L, s);
        case OSS_F16: return proj_fwd_t<f16_t>(x2, Wx, Wdt, xdbl, dts, B, D, C, R, L, s);
        case OSS_BF16: return proj_fwd_t<bf16_t>(x2, Wx, Wdt, xdbl, dts, B, D, C, R, L, s);
    }
    return OSS_ERR_SHAPE;
}

int proj_dgrad(oss_dtype io, const void *ddts, void *dxdbl, const void *du, const float *Wx, const float *Wdt, void *dx2, int B,
               int D, int C, int R, int L, hipStream_t s) {
    switch (io) {
        case OSS_F32: return proj_dgrad_t<float>(ddts, dxdbl, du, Wx, Wdt, dx2, B, D, C, R, L, s);
        case OSS_F16: return proj_dgrad_t<f16_t>(ddts, dxdbl, du, Wx, Wdt, dx2, B, D, C, R, L, s);
        case OSS_BF16: return proj_dgrad_t<bf16_t>(ddts, dxdbl, du, Wx, Wdt, dx2, B, D, C, R, L, s);
    }
    return OSS_ERR_SHAPE;
}

template <typename TI, typename TO>
static int cross_scan2_t(const void *x, void *x2, int B, int D, int H, int W, int64_t xsb, int64_t xsc, hipStream_t s) {
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
