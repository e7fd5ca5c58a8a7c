// oss_dwconv.hip -- depth-wise 3x3 convolution (stride 1, zero padding 1) of the OSS block for gfx950:
// SS2D_1.conv2d (SRGAN/VmambaIR/archs/MambaSISR6_arch.py:286-294, groups = channels, bias) and the
// EFFN dwconv (:209, groups = channels, no bias).  HBM-bound stencil: every plane is read once and
// written once; the 3x3 neighbourhood comes out of L1/L2 (a 64x64 plane is 8-16 KB).
//
// MIOpen runs these grouped convolutions in bf16 through generic grouped-conv / naive kernels
// (profiles/r01_rocprof_bench_eager_last_step.txt: 390 ms of a 640 ms step in one
// grouped_conv_bwd_weight kernel); the three kernels below replace forward, input-gradient (same
// stencil with the taps mirrored) and weight/bias-gradient.
#include <initializer_list>
#include "oss_device.h"
#include "oss_host.h"
#include "oss_stencil.h"

namespace oss {

__device__ __forceinline__ float sigmoid_f32(float z) { return __builtin_amdgcn_rcpf(1.f + exp2_hw(-z * kLog2e)); }
__device__ __forceinline__ float silu_f32(float z) { return z * sigmoid_f32(z); }
// d silu(z) / dz
__device__ __forceinline__ float dsilu_f32(float z) { const float s = sigmoid_f32(z); return s * __builtin_fmaf(z, 1.f - s, 1.f); }

template <typename T> struct Vec4;  // 4 consecutive elements
template <> struct Vec4<float> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[4]) {
        f32x4 q = *reinterpret_cast<const f32x4 *>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    }
    static __device__ __forceinline__ void store(float *p, const float (&v)[4]) {
        *reinterpret_cast<f32x4 *>(p) = f32x4{v[0], v[1], v[2], v[3]};
    }
};
template <typename T> struct Vec4 {
    static __device__ __forceinline__ void load(const T *p, float (&v)[4]) {
        u32x2 q = *reinterpret_cast<const u32x2 *>(p);
        unpack2<T>(q.x, v[0], v[1]); unpack2<T>(q.y, v[2], v[3]);
    }
    static __device__ __forceinline__ void store(T *p, const float (&v)[4]) {
        *reinterpret_cast<u32x2 *>(p) = u32x2{pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
    }
};

// y[b,c] = stencil(x[b,c]; taps of channel c (mirrored when flip)) + bias[c]
// x: planes contiguous (H*W), batch / channel strides in elements; y likewise.
template <typename T, bool VEC>
__global__ void __launch_bounds__(256)
oss_dwconv3x3_kernel(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                     T *__restrict__ y, int C, int H, int W, int64_t xsb, int64_t xsc, int64_t ysb, int64_t ysc,
                     int flip, T *__restrict__ pre /* contiguous (B, C, H, W) or NULL: receives the conv output before the activation */,
                     int act /* y = silu(conv) */) {
    const int c = blockIdx.y, b = blockIdx.z;
    const T *xp = x + b * xsb + c * xsc;
    T *yp = y + b * ysb + c * ysc;
    T *pp = pre ? pre + ((size_t)b * C + c) * H * W : nullptr;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = w[c * 9 + (flip ? 8 - i : i)];
    const float bv = bias ? bias[c] : 0.f;
    if constexpr (VEC) {
        const int groups_per_row = W >> 2;
        const int g = blockIdx.x * 256 + threadIdx.x;
        if (g >= groups_per_row * H) return;
        const int h = g / groups_per_row, w0 = (g - h * groups_per_row) << 2;
        float acc[4] = {bv, bv, bv, bv};
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int hh = h + dy;
            if (hh < 0 || hh >= H) continue;
            const T *row = xp + (int64_t)hh * W;
            float v[6];
            float m[4];
            Vec4<T>::load(row + w0, m);
            v[1] = m[0]; v[2] = m[1]; v[3] = m[2]; v[4] = m[3];
            v[0] = (w0 > 0) ? to_f32(row[w0 - 1]) : 0.f;
            v[5] = (w0 + 4 < W) ? to_f32(row[w0 + 4]) : 0.f;
            const float k0 = k[(dy + 1) * 3], k1 = k[(dy + 1) * 3 + 1], k2 = k[(dy + 1) * 3 + 2];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_fmaf(k0, v[j], __builtin_fmaf(k1, v[j + 1], __builtin_fmaf(k2, v[j + 2], acc[j])));
        }
        if (pp) Vec4<T>::store(pp + (int64_t)h * W + w0, acc);   // kept for the backward
        if (act) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = silu_f32(acc[j]);
        }
        Vec4<T>::store(yp + (int64_t)h * W + w0, acc);
    } else {
        const int p = blockIdx.x * 256 + threadIdx.x;
        if (p >= H * W) return;
        const int h = p / W, ww = p - h * W;
        float acc = bv;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int hh = h + dy;
            if (hh < 0 || hh >= H) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int wc = ww + dx;
                if (wc < 0 || wc >= W) continue;
                acc = __builtin_fmaf(k[(dy + 1) * 3 + dx + 1], to_f32(xp[(int64_t)hh * W + wc]), acc);
            }
        }
        if (pp) pp[p] = from_f32<T>(acc);
        if (act) acc = silu_f32(acc);
        yp[p] = from_f32<T>(acc);
    }
}

// ---- 16-bit I/O, 8 pixels per lane: load8 / store8 / row10 of oss_stencil.h ------------------------------------------
template <typename T, bool EDGE = false>
__global__ void __launch_bounds__(256)
oss_dwconv3x3_wide_kernel(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                          T *__restrict__ y, int C, int H, int W, int64_t xsb, int64_t xsc, int64_t ysb, int64_t ysc,
                          int flip, T *__restrict__ pre, int act,
                          T *__restrict__ yt = nullptr /* (flat2 form) the same plane once more, column-major: element (h, w) at
                          w * H + h, planes at yt + b * ysb + c * ysc -- the second flattening that SS2D_1's scans walk
                          (cross_scan_2d, MambaSISR6_arch.py:399-404), written here instead of by a transpose launch of its own.
                          Needs 256 % (W / 8) == 0, 256 / (W / 8) % 8 == 0 and H % 8 == 0 (flat2_ok) */) {
    __shared__ __attribute__((aligned(16))) T tile[2048];   // flat2 form: the workgroup's 256 groups of 8 pixels, row-major
    const int c = blockIdx.y, b = blockIdx.z;
    const T *xp = x + b * xsb + c * xsc;
    T *yp = y + b * ysb + c * ysc;
    T *pp = pre ? pre + ((size_t)b * C + c) * H * W : nullptr;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = w[c * 9 + (flip ? 8 - i : i)];
    const float bv = bias ? bias[c] : 0.f;
    const int lpr = W >> 3, ngroups = lpr * H;
    const int g = blockIdx.x * 256 + threadIdx.x;
    const bool live = g < ngroups;
    const int gc = live ? g : ngroups - 1;     // dead lanes of the last workgroup shadow the last group (no stores)
    const int h = gc / lpr, cg = gc - h * lpr, w0 = cg << 3;
    const bool first = cg == 0, last = cg == lpr - 1;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bv;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        float v[10];
        row10<T, EDGE>(xp, h, dy, H, W, w0, first, last, v);
        const float k0 = k[(dy + 1) * 3], k1 = k[(dy + 1) * 3 + 1], k2 = k[(dy + 1) * 3 + 2];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc[j] = __builtin_fmaf(k0, v[j], __builtin_fmaf(k1, v[j + 1], __builtin_fmaf(k2, v[j + 2], acc[j])));
    }
    if (!live && !yt) return;
    if (live && pp) store8<T>(pp + (int64_t)h * W + w0, acc);   // kept for the backward of the unfused path
    if (act) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = silu_f32(acc[j]);
    }
    if (live) store8<T>(yp + (int64_t)h * W + w0, acc);
    if (yt) {
        // the workgroup's R = 256 / lpr rows through LDS: written as computed (lane = 8 pixels of a row), read back with lane = (column,
        // 8 rows) -- consecutive lanes read consecutive 2-byte columns of a row (no bank conflicts) and store 16 bytes of a column
        if (live) store8<T>(tile + threadIdx.x * 8, acc);
        __syncthreads();
        const int rows = 256 / lpr, hb = blockIdx.x * rows;
        const int wc = threadIdx.x % W, hq = threadIdx.x / W;   // W * rows / 8 = 256 octets
        const int h0 = hb + hq * 8;
        if (h0 < H) {   // H % 8 == 0: an octet is inside the plane as a whole
            float col8[8];   // (exact: the tile holds values already rounded to T)
#pragma unroll
            for (int j = 0; j < 8; ++j) col8[j] = to_f32(tile[(hq * 8 + j) * W + wc]);
            store8<T>(yt + b * ysb + c * ysc + (int64_t)wc * H + h0, col8);
        }
    }
}

// weight / bias gradient partials of one (channel, batch) plane, same access scheme
template <typename T, bool EDGE = false>
__global__ void __launch_bounds__(256)
oss_dwconv3x3_wgrad_wide_kernel(const T *__restrict__ x, const T *__restrict__ dy, float *__restrict__ part /*[B][C][10]*/,
                                int C, int H, int W, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc,
                                const T *__restrict__ pre, T *__restrict__ dpre) {
    const int c = blockIdx.x, b = blockIdx.y;
    const T *pp = pre ? pre + ((size_t)b * C + c) * H * W : nullptr;
    T *qp = pre ? dpre + ((size_t)b * C + c) * H * W : nullptr;
    float acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.f;
    const T *xp = x + b * xsb + c * xsc;
    const T *gp = dy + b * gsb + c * gsc;
    const int lpr = W >> 3, ngroups = lpr * H;
    for (int g0 = 0; g0 < ngroups; g0 += 256) {   // uniform trip count: every lane takes part in the DPP halo exchange
        const int g = g0 + threadIdx.x;
        const bool live = g < ngroups;
        const int gc = live ? g : ngroups - 1;
        const int h = gc / lpr, cg = gc - h * lpr, w0 = cg << 3;
        const bool first = cg == 0, last = cg == lpr - 1;
        float gv[8];
        load8<T>(gp + (int64_t)h * W + w0, gv);
        if (pp) {
            float pv[8];
            load8<T>(pp + (int64_t)h * W + w0, pv);
#pragma unroll
            for (int j = 0; j < 8; ++j) gv[j] *= dsilu_f32(pv[j]);
            if (live) store8<T>(qp + (int64_t)h * W + w0, gv);
#pragma unroll
            for (int j = 0; j < 8; ++j) gv[j] = to_f32(from_f32<T>(gv[j]));  // the input-gradient pass sees the rounded value
        }
        if (!live) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gv[j] = 0.f;
        }
        acc[9] += ((gv[0] + gv[1]) + (gv[2] + gv[3])) + ((gv[4] + gv[5]) + (gv[6] + gv[7]));
#pragma unroll
        for (int dyy = -1; dyy <= 1; ++dyy) {
            float v[10];
            row10<T, EDGE>(xp, h, dyy, H, W, w0, first, last, v);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                float a = acc[(dyy + 1) * 3 + dx];
#pragma unroll
                for (int j = 0; j < 8; ++j) a = __builtin_fmaf(gv[j], v[j + dx], a);
                acc[(dyy + 1) * 3 + dx] = a;
            }
        }
    }
    __shared__ float red[4][10];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const float s = segment_sum_to_last<64>(acc[i]);
        if (lane == 63) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 10)
        part[(size_t)b * C * 10 + (threadIdx.x < 9 ? (size_t)c * 9 + threadIdx.x : (size_t)9 * C + c)] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// ---- the convolution together with what follows it, without materialising the convolution --------------------------
// SS2D_1: x = silu(conv2d(x)) (MambaSISR6_arch.py:486); EFFN: x1, x2 = dwconv(t).chunk(2); gelu(x1) * x2 (:213-217).
// The convolution output is never stored: the forward writes only what the next layer reads, and the backward recomputes it
// from the rows of the input it needs anyway for the weight gradient.  One workgroup of the backward owns the whole (H, W)
// plane of its channel (pair): pass 1 forms the gradient that reaches the convolution (rounded to the I/O type, as the
// separate kernels hand it over), leaves it in LDS and accumulates the weight / bias gradient partials; pass 2 convolves
// the LDS planes with the mirrored taps -- three launches (activation backward, weight gradient, input gradient) and
// four plane round trips through HBM become one launch that reads x, dy and writes dx.
enum { kDwSilu = 0, kDwGate = 1 };
// one element as a 2- or 4-byte load left it in a register -> fp32
template <typename T> __device__ __forceinline__ float raw_item_f32(uint32_t r) {
    if constexpr (sizeof(T) == 4) return __uint_as_float(r);
    else return to_f32(T{(uint16_t)r});
}

// (gelu_parts: oss_stencil.h -- shared with the fused EFFN forward, oss_effn.hip)
template <typename T>
__device__ __forceinline__ void conv_rows(const float (&v)[3][10], const float *__restrict__ k, float bv, float (&acc)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bv;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc[j] = __builtin_fmaf(k[r * 3], v[r][j], __builtin_fmaf(k[r * 3 + 1], v[r][j + 1], __builtin_fmaf(k[r * 3 + 2], v[r][j + 2], acc[j])));
}

// out[b, c] = gelu(conv(t[b, c])) * conv(t[b, c + Hd]);  grid (groups of 256 lanes, Hd, B)
// (round 4, measured and removed: two vertically adjacent groups per lane -- 8 row loads per two outputs instead of 12, the headline
// launch 4 waves per SIMD in one round instead of 7.9 at 6 resident: 13.3 us per launch either way, 236.8 / 235.9 against 236.7 / 236.8
// images/s; tools/r4_call20.sh)
template <typename T, bool EDGE = false>
__global__ void __launch_bounds__(256)
oss_dwgate_fwd_kernel(const T *__restrict__ t, const float *__restrict__ w, const float *__restrict__ bias, T *__restrict__ out,
                      int Hd, int H, int W, int64_t tsb, int64_t tsc, int64_t osb, int64_t osc) {
    const int c = blockIdx.y, b = blockIdx.z;
    const T *p1 = t + b * tsb + c * tsc, *p2 = p1 + Hd * tsc;
    T *op = out + b * osb + c * osc;
    const float *k1 = w + c * 9, *k2 = w + (c + Hd) * 9;
    const float b1 = bias ? bias[c] : 0.f, b2 = bias ? bias[c + Hd] : 0.f;
    const int lpr = W >> 3, ngroups = lpr * H;
    const int g = blockIdx.x * 256 + threadIdx.x;
    const bool live = g < ngroups;
    const int gc = live ? g : ngroups - 1;
    const int h = gc / lpr, cg = gc - h * lpr, w0 = cg << 3;
    const bool first = cg == 0, last = cg == lpr - 1;
    float v[3][10], x1[8], x2[8];
#pragma unroll
    for (int r = 0; r < 3; ++r) row10<T, EDGE>(p1, h, r - 1, H, W, w0, first, last, v[r]);
    conv_rows<T>(v, k1, b1, x1);
#pragma unroll
    for (int r = 0; r < 3; ++r) row10<T, EDGE>(p2, h, r - 1, H, W, w0, first, last, v[r]);
    conv_rows<T>(v, k2, b2, x2);
    if (!live) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float cdf, pdf;
        gelu_parts(x1[j], cdf, pdf);
        x1[j] = x1[j] * cdf * x2[j];
    }
    store8<T>(op + (int64_t)h * W + w0, x1);
}

// MODE kDwSilu: one channel per workgroup, grid (C, B); dy is the gradient of silu(conv(x) + bias).
// MODE kDwGate: channels c and c + Hd, grid (Hd, B); dy (B, Hd, H, W) is the gradient of gelu(x1) * x2.
// dynamic LDS: NCH planes of (H + 2) rows of W elements of T (rows 0 and H + 1 stay zero: the padding of pass 2)
// KEEP (round 4; planes of <= 512 lane groups, i.e. <= 4096 pixels, non-EDGE widths): the kernel was a chain of dependent round trips --
// pass 1 loaded one channel's three rows, waited, convolved, then the partner channel's; pass 2 loaded x AGAIN, group by group, each
// load waited for on the spot (PMC at the headline gate shape: 26 % of the waves' cycles issuing vector instructions, 38 % parked).
// Now every load of a pass-1 group (dy, 3 rows x NCH channels) is in flight before the first is used, and the centre rows stay in
// registers (16 bytes per channel and group, at most 2 groups per lane) for pass 2, which then reads LDS only.
template <typename T, int MODE, bool EDGE = false, bool KEEP = false, bool MERGE = false /* KEEP form: dyt is there (a template
parameter, not `if (dyt)`: behind a branch the compiler converts the loaded values where they were loaded, i.e. waits for them there) */>
__global__ void __launch_bounds__(256, 4)   // 4 waves per SIMD = 4 workgroups per CU: the headline's 1016 gate workgroups in one round
oss_dwconv3x3_bwd_fused_kernel(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                               const T *__restrict__ dy, T *__restrict__ dx, float *__restrict__ part /*[B][C][10]*/,
                               int C, int H, int W, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, int64_t dsb, int64_t dsc,
                               const T *__restrict__ dyt = nullptr /* (flat2 form, kDwSilu) the gradient of the column-major flattening,
                               same strides: dy(h, w) += dyt[w * H + h], the sum rounded to T -- the adjoint of cross_scan_2d's two
                               forward flattenings (MambaSISR6_arch.py:399-404) read here instead of merged by a launch of its own */) {
    static_assert(!(KEEP && EDGE), "KEEP: rows whose lane groups tile a wave");
    constexpr int NCH = MODE == kDwGate ? 2 : 1;
    constexpr int NKEEP = KEEP ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char dw_smem[];
    T *sg = reinterpret_cast<T *>(dw_smem);
    const int c0 = blockIdx.x, b = blockIdx.y;
    const int cstep = C / NCH;                       // kDwGate: the partner channel is c0 + Hd
    const size_t plane = (size_t)(H + 2) * W;
    const T *gp = dy + b * gsb + c0 * gsc;
    const int lpr = W >> 3, ngroups = lpr * H;
    Raw8<T> xkeep[NKEEP][NCH];   // KEEP: the centre row of x of the lane's pass-1 groups (pass 2 walks the same groups)
    // zero the two padding rows of every LDS plane
    for (int i = threadIdx.x; i < NCH * 2 * lpr; i += 256) {
        const int ch = i / (2 * lpr), r = i - ch * 2 * lpr, row = r < lpr ? 0 : H + 1, col = (r < lpr ? r : r - lpr) << 3;
        store8_raw<T>(sg + ch * plane + (size_t)row * W + col, zero8<T>());
    }
    // pass 1: the gradient that reaches the convolution, rounded to T, into the LDS planes
    struct Raw { Raw8<T> g, rq[NCH][3]; uint32_t tv[MERGE ? 8 : 1]; };   // as loaded, one register each: a conversion (or packing two of them) at issue time would be a wait
    auto coords = [&](int g0, bool &live, int &h, int &w0, bool &first, bool &last) {
        const int g = g0 + threadIdx.x;
        live = g < ngroups;
        const int gc = live ? g : ngroups - 1;
        h = gc / lpr;
        const int cg = gc - h * lpr;
        w0 = cg << 3; first = cg == 0; last = cg == lpr - 1;
    };
    auto issue = [&](int g0, Raw &r) {   // KEEP: every load of a pass-1 group, nothing waited for
        bool live, first, last; int h, w0;
        coords(g0, live, h, w0, first, last);
        r.g = load8_raw<T>(gp + (int64_t)h * W + w0);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int k = 0; k < 3; ++k) r.rq[ch][k] = row10_issue<T>(x + b * xsb + (c0 + ch * cstep) * xsc, h, k - 1, H, W, w0);
        if constexpr (MERGE) {   // 8 two-byte loads: the 8 lanes that share a column group read 16 consecutive bytes of each column
            const T *tp = dyt + b * gsb + c0 * gsc + (int64_t)w0 * H + h;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (sizeof(T) == 4) r.tv[j] = __float_as_uint(tp[(int64_t)j * H]);
                else                          r.tv[j] = tp[(int64_t)j * H].v;
            }
        }
    };
    auto finish = [&](bool live, int h, int w0, const float (&gv)[8], const float (&pre)[NCH][8]) {
        float gq[NCH][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (MODE == kDwSilu) {
                gq[0][j] = gv[j] * dsilu_f32(pre[0][j]);
            } else {
                const float a = pre[0][j];
                float cdf, pdf;
                gelu_parts(a, cdf, pdf);
                gq[0][j] = gv[j] * pre[1][j] * __builtin_fmaf(a, pdf, cdf);
                gq[1][j] = gv[j] * a * cdf;
            }
        }
        if (live) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) store8<T>(sg + ch * plane + (size_t)(h + 1) * W + w0, gq[ch]);
        }
    };
    auto process = [&](int g0, const Raw &r, Raw8<T> (&keep)[NCH]) {
        bool live, first, last; int h, w0;
        coords(g0, live, h, w0, first, last);
        float gv[8], pre[NCH][8];
        unpack8<T>(r.g, gv);
        if constexpr (MERGE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gv[j] = to_f32(from_f32<T>(gv[j] + raw_item_f32<T>(r.tv[j])));   // as the merge launch stored it
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = c0 + ch * cstep;
            float v[3][10];
#pragma unroll
            for (int k = 0; k < 3; ++k) row10_finish<T>(r.rq[ch][k], h, k - 1, H, first, last, v[k]);
            conv_rows<T>(v, w + c * 9, bias ? bias[c] : 0.f, pre[ch]);
            keep[ch] = r.rq[ch][1];
        }
        finish(live, h, w0, gv, pre);
    };
    auto pass1 = [&](int g0) {   // the loads of one channel at a time, each waited for where it is used
        bool live, first, last; int h, w0;
        coords(g0, live, h, w0, first, last);
        float gv[8], pre[NCH][8];
        load8<T>(gp + (int64_t)h * W + w0, gv);
        if constexpr (MODE == kDwSilu) {
            if (dyt) {
                const T *tp = dyt + b * gsb + c0 * gsc + (int64_t)w0 * H + h;
                float tv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) tv[j] = to_f32(tp[(int64_t)j * H]);
#pragma unroll
                for (int j = 0; j < 8; ++j) gv[j] = to_f32(from_f32<T>(gv[j] + tv[j]));   // as the merge launch stored it
            }
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = c0 + ch * cstep;
            const T *xp = x + b * xsb + c * xsc;
            float v[3][10];
#pragma unroll
            for (int k = 0; k < 3; ++k) row10<T, EDGE>(xp, h, k - 1, H, W, w0, first, last, v[k]);
            conv_rows<T>(v, w + c * 9, bias ? bias[c] : 0.f, pre[ch]);
            __builtin_amdgcn_sched_barrier(0);   // one channel's rows at a time in registers
        }
        finish(live, h, w0, gv, pre);
    };
    if constexpr (KEEP) {
        const bool two = ngroups > 256;
        Raw r0, r1;
        issue(0, r0);
        if constexpr (MODE == kDwSilu) {   // one channel: both groups' loads fit the registers of four waves per SIMD
            if (two) issue(256, r1);
            __builtin_amdgcn_sched_barrier(0);   // everything above is in flight before anything below waits
            process(0, r0, xkeep[0]);
            if (two) process(256, r1, xkeep[1]);
        } else {
            __builtin_amdgcn_sched_barrier(0);
            process(0, r0, xkeep[0]);
            if (two) {
                issue(256, r1);
                __builtin_amdgcn_sched_barrier(0);
                process(256, r1, xkeep[1]);
            }
        }
    } else if constexpr (!EDGE && sizeof(T) == 2 && MODE == kDwGate) {
        // planes of more than 512 groups (the Deraining tree's 128 x 128 level): the group's seven loads in flight together, as in the
        // KEEP form; x is loaded again in pass 2
        for (int g0 = 0; g0 < ngroups; g0 += 256) {   // uniform trip count: every lane takes part in the DPP halo exchange
            Raw r;
            issue(g0, r);
            __builtin_amdgcn_sched_barrier(0);
            process(g0, r, xkeep[0]);
        }
    } else {
        for (int g0 = 0; g0 < ngroups; g0 += 256) pass1(g0);   // uniform trip count: every lane takes part in the DPP halo exchange
    }
    __syncthreads();   // the LDS planes are complete
    // pass 2: with q = the gradient rows h - 1 .. h + 1 out of LDS (row h of the image is LDS row h + 1; both gradients below see
    // the rounded values),  dx[p] = sum_t k[8 - t] q[p + t]  (the mirrored stencil)  and, from the same registers,
    // dw[8 - t] += x[p] q[p + t],  db += q[p]
    // (one channel at a time, all of its groups: the second channel's rows, addresses and sums would not fit the 128 registers
    // of four waves per SIMD next to the first one's)
    __shared__ float red[4][NCH * 10];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c = c0 + ch * cstep;
        const T *sp = sg + ch * plane;
        const T *xp = x + b * xsb + c * xsc;
        T *dxp = dx + b * dsb + c * dsc;
        float acc[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) acc[i] = 0.f;
        auto pass2 = [&](int g0, const Raw8<T> &kept) {
            const int g = g0 + threadIdx.x;
            const bool live = g < ngroups;
            const int gc = live ? g : ngroups - 1;
            const int h = gc / lpr, cg = gc - h * lpr, w0 = cg << 3;
            const bool first = cg == 0, last = cg == lpr - 1;
            float xc[8], o[8];
            {
                Raw8<T> q;
                if constexpr (KEEP) q = kept;
                else q = load8_raw<T>(xp + (int64_t)h * W + w0);
                if (!live) q = zero8<T>();   // a lane that shadows the last group adds nothing to the sums
                unpack8<T>(q, xc);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float m[8], vv[10];
                load8<T>(sp + (size_t)(h + r) * W + w0, m);
#pragma unroll
                for (int j = 0; j < 8; ++j) vv[j + 1] = m[j];
                vv[0] = shift_from_prev_lane(vv[8], 0.f, first);
                vv[9] = shift_from_next_lane(vv[1], 0.f, last);
                if constexpr (EDGE) {   // a row's groups straddle waves: the halo pixel across the wave edge comes from the LDS plane
                    const T *rp = sp + (size_t)(h + r) * W + w0;
                    if (lane == 0 && !first) vv[0] = to_f32(rp[-1]);
                    if (lane == 63 && !last) vv[9] = to_f32(rp[8]);
                }
                const float k0 = w[c * 9 + 8 - r * 3], k1 = w[c * 9 + 7 - r * 3], k2 = w[c * 9 + 6 - r * 3];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    o[j] = __builtin_fmaf(k0, vv[j], __builtin_fmaf(k1, vv[j + 1], __builtin_fmaf(k2, vv[j + 2], o[j])));
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    float a = acc[8 - (r * 3 + cc)];
#pragma unroll
                    for (int j = 0; j < 8; ++j) a = __builtin_fmaf(xc[j], vv[j + cc], a);
                    acc[8 - (r * 3 + cc)] = a;
                }
                if (r == 1) {
                    const float sq = ((vv[1] + vv[2]) + (vv[3] + vv[4])) + ((vv[5] + vv[6]) + (vv[7] + vv[8]));
                    acc[9] += live ? sq : 0.f;
                }
            }
            if (live) store8<T>(dxp + (int64_t)h * W + w0, o);
        };
        if constexpr (KEEP) {
            pass2(0, xkeep[0][ch]);
            if (ngroups > 256) pass2(256, xkeep[1][ch]);
        } else {
            for (int g0 = 0; g0 < ngroups; g0 += 256) pass2(g0, xkeep[0][0]);
        }
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const float s = segment_sum_to_last<64>(acc[i]);
            if (lane == 63) red[wave][ch * 10 + i] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < NCH * 10) {
        const int ch = threadIdx.x / 10, i = threadIdx.x - ch * 10, c = c0 + ch * cstep;
        part[(size_t)b * C * 10 + (i < 9 ? (size_t)c * 9 + i : (size_t)9 * C + c)] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    }
}

// the 8-pixel kernels apply when a row's W / 8 lane groups tile a wave and every plane / row start is 16-byte aligned
template <typename T>
static bool wide_ok(int W, std::initializer_list<const void *> ptrs, std::initializer_list<int64_t> strides) {
    if (W % 8 != 0 || W > 512) return false;   // (64 % (W / 8)) != 0: the EDGE instantiations; float I/O since round 4
    for (const void *p : ptrs)
        if (reinterpret_cast<uintptr_t>(p) & 15u) return false;
    for (int64_t st : strides)
        if (st % 8 != 0) return false;
    return true;
}

// dw[c][ky][kx] = sum_{b,h,w} dy[b,c,h,w] x[b,c,h+ky-1,w+kx-1];  db[c] = sum dy.
// Stage 1: one workgroup per (channel, batch) plane writes 10 partial sums; stage 2 adds the batch
// partials in batch order (deterministic, no atomics).
template <typename T, bool VEC>
__global__ void __launch_bounds__(256)
oss_dwconv3x3_wgrad_kernel(const T *__restrict__ x, const T *__restrict__ dy, float *__restrict__ part /*[B][C][10]*/,
                           int C, int H, int W, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc,
                           const T *__restrict__ pre, T *__restrict__ dpre /* fused silu: the gradient that reaches the conv is
                           dy * silu'(pre); it is also written to dpre (contiguous) for the input-gradient pass */) {
    const int c = blockIdx.x, b = blockIdx.y;
    const T *pp = pre ? pre + ((size_t)b * C + c) * H * W : nullptr;
    T *qp = pre ? dpre + ((size_t)b * C + c) * H * W : nullptr;
    float acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.f;
    const T *xp = x + b * xsb + c * xsc;
    const T *gp = dy + b * gsb + c * gsc;
    if constexpr (VEC) {
        const int gpr = W >> 2, ngroups = gpr * H;
        for (int g = threadIdx.x; g < ngroups; g += 256) {
            const int h = g / gpr, w0 = (g - h * gpr) << 2;
            float gv[4];
            Vec4<T>::load(gp + (int64_t)h * W + w0, gv);
            if (pp) {
                float pv[4];
                Vec4<T>::load(pp + (int64_t)h * W + w0, pv);
#pragma unroll
                for (int j = 0; j < 4; ++j) gv[j] *= dsilu_f32(pv[j]);
                Vec4<T>::store(qp + (int64_t)h * W + w0, gv);
#pragma unroll
                for (int j = 0; j < 4; ++j) gv[j] = to_f32(from_f32<T>(gv[j]));  // the input-gradient pass sees the rounded value
            }
            acc[9] += (gv[0] + gv[1]) + (gv[2] + gv[3]);
#pragma unroll
            for (int dyy = -1; dyy <= 1; ++dyy) {
                const int hh = h + dyy;
                if (hh < 0 || hh >= H) continue;
                const T *row = xp + (int64_t)hh * W;
                float v[6], m[4];
                Vec4<T>::load(row + w0, m);
                v[1] = m[0]; v[2] = m[1]; v[3] = m[2]; v[4] = m[3];
                v[0] = (w0 > 0) ? to_f32(row[w0 - 1]) : 0.f;
                v[5] = (w0 + 4 < W) ? to_f32(row[w0 + 4]) : 0.f;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    float a = acc[(dyy + 1) * 3 + dx];
#pragma unroll
                    for (int j = 0; j < 4; ++j) a = __builtin_fmaf(gv[j], v[j + dx], a);
                    acc[(dyy + 1) * 3 + dx] = a;
                }
            }
        }
    } else {
        const int HW = H * W;
        for (int p = threadIdx.x; p < HW; p += 256) {
            const int h = p / W, ww = p - h * W;
            float g = to_f32(gp[p]);
            if (pp) {
                const T gq = from_f32<T>(g * dsilu_f32(to_f32(pp[p])));
                qp[p] = gq;
                g = to_f32(gq);
            }
            acc[9] += g;
#pragma unroll
            for (int dyy = -1; dyy <= 1; ++dyy) {
                const int hh = h + dyy;
                if (hh < 0 || hh >= H) continue;
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int wc = ww + dx;
                    if (wc < 0 || wc >= W) continue;
                    acc[(dyy + 1) * 3 + dx + 1] = __builtin_fmaf(g, to_f32(xp[(int64_t)hh * W + wc]), acc[(dyy + 1) * 3 + dx + 1]);
                }
            }
        }
    }
    __shared__ float red[4][10];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const float s = segment_sum_to_last<64>(acc[i]);
        if (lane == 63) red[wave][i] = s;
    }
    __syncthreads();
    // one partial vector per batch element: [C][9] tap sums, then the C bias sums
    if (threadIdx.x < 10)
        part[(size_t)b * C * 10 + (threadIdx.x < 9 ? (size_t)c * 9 + threadIdx.x : (size_t)9 * C + c)] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ void __launch_bounds__(256)
oss_dwconv3x3_wgrad_finish(const float *__restrict__ part, float *__restrict__ dw, float *__restrict__ db, int B, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // over C*10
    if (i >= C * 10) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += part[(size_t)b * C * 10 + i];
    if (i < 9 * C) dw[i] = s;
    else if (db) db[i - 9 * C] = s;
}

template <typename T>
static int dwconv_launch(const void *x, const float *w, const float *bias, void *y, int B, int C, int H, int W,
                         int64_t xsb, int64_t xsc, int64_t ysb, int64_t ysc, int flip, hipStream_t s, void *pre, int act) {
    const T *xp = reinterpret_cast<const T *>(x);
    T *yp = reinterpret_cast<T *>(y);
    T *prp = reinterpret_cast<T *>(pre);
    {
        if (wide_ok<T>(W, {xp, yp, prp}, {xsb, xsc, ysb, ysc})) {
            dim3 grid(((W / 8) * H + 255) / 256, C, B);
            if ((64 % (W / 8)) != 0)
                hipLaunchKernelGGL((oss_dwconv3x3_wide_kernel<T, true>), grid, dim3(256), 0, s, xp, w, bias, yp, C, H, W, xsb, xsc, ysb, ysc, flip, prp, act);
            else
                hipLaunchKernelGGL((oss_dwconv3x3_wide_kernel<T, false>), grid, dim3(256), 0, s, xp, w, bias, yp, C, H, W, xsb, xsc, ysb, ysc, flip, prp, act);
            return (int)hipGetLastError();
        }
    }
    const uintptr_t amask = sizeof(T) == 4 ? 15u : 7u;
    const bool vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(xp) | reinterpret_cast<uintptr_t>(yp) | reinterpret_cast<uintptr_t>(prp)) & amask) == 0 &&
                     (xsb % 4 == 0) && (xsc % 4 == 0) && (ysb % 4 == 0) && (ysc % 4 == 0);
    if (vec) {
        const int groups = (W / 4) * H;
        dim3 grid((groups + 255) / 256, C, B);
        hipLaunchKernelGGL((oss_dwconv3x3_kernel<T, true>), grid, dim3(256), 0, s, xp, w, bias, yp, C, H, W, xsb, xsc, ysb, ysc, flip, prp, act);
    } else {
        dim3 grid((H * W + 255) / 256, C, B);
        hipLaunchKernelGGL((oss_dwconv3x3_kernel<T, false>), grid, dim3(256), 0, s, xp, w, bias, yp, C, H, W, xsb, xsc, ysb, ysc, flip, prp, act);
    }
    return (int)hipGetLastError();
}

int dwconv3x3(oss_dtype io, const void *x, const float *w, const float *bias, void *y, int B, int C, int H, int W,
              int64_t xsb, int64_t xsc, int64_t ysb, int64_t ysc, int flip, hipStream_t s, void *pre, int act) {
    switch (io) {
        case OSS_F32: return dwconv_launch<float>(x, w, bias, y, B, C, H, W, xsb, xsc, ysb, ysc, flip, s, pre, act);
        case OSS_F16: return dwconv_launch<f16_t>(x, w, bias, y, B, C, H, W, xsb, xsc, ysb, ysc, flip, s, pre, act);
        case OSS_BF16: return dwconv_launch<bf16_t>(x, w, bias, y, B, C, H, W, xsb, xsc, ysb, ysc, flip, s, pre, act);
    }
    return OSS_ERR_SHAPE;
}

// ---- fused forms: host side ------------------------------------------------------------------------------------------
static size_t fused_lds_bytes(int nch, int H, int W, size_t esize) { return (size_t)nch * (H + 2) * W * esize; }

// shapes the fused kernels take: 16-bit I/O, rows of W / 8 lane groups (EDGE instantiations when they do not tile a wave), the
// channel (pair)'s planes in LDS
static bool dw_edge(int W) { return (64 % (W / 8)) != 0; }
int dwconv3x3_fused_ok(oss_dtype io, int H, int W, int nch) {
    if (io != OSS_F16 && io != OSS_BF16 && io != OSS_F32) return 0;   // (float I/O: round 4 -- the reference's own training precision)
    if (nch < 1 || nch > 2 || H <= 0 || W <= 0 || W % 8 != 0 || W > 512) return 0;
    return fused_lds_bytes(nch, H, W, io == OSS_F32 ? 4 : 2) + 4 * 20 * sizeof(float) <= kMaxLdsBytes ? 1 : 0;
}

static bool aligned16(std::initializer_list<const void *> ptrs, std::initializer_list<int64_t> strides) {
    for (const void *p : ptrs)
        if (reinterpret_cast<uintptr_t>(p) & 15u) return false;
    for (int64_t st : strides)
        if (st % 8 != 0) return false;
    return true;
}

template <typename T>
static int dwgate_fwd_launch(const void *t, const float *w, const float *bias, void *out, int B, int Hd, int H, int W,
                             int64_t tsb, int64_t tsc, int64_t osb, int64_t osc, hipStream_t s) {
    if (!aligned16({t, out}, {tsb, tsc, osb, osc})) return OSS_ERR_SHAPE;
    dim3 grid(((W / 8) * H + 255) / 256, Hd, B);
    if (dw_edge(W))
        hipLaunchKernelGGL((oss_dwgate_fwd_kernel<T, true>), grid, dim3(256), 0, s, reinterpret_cast<const T *>(t), w, bias,
                           reinterpret_cast<T *>(out), Hd, H, W, tsb, tsc, osb, osc);
    else
        hipLaunchKernelGGL((oss_dwgate_fwd_kernel<T, false>), grid, dim3(256), 0, s, reinterpret_cast<const T *>(t), w, bias,
                           reinterpret_cast<T *>(out), Hd, H, W, tsb, tsc, osb, osc);
    return (int)hipGetLastError();
}

// the forward alone streams (three rows per lane group, nothing in LDS): any plane of rows of W / 8 <= 64 lane groups -- the inference
// path of planes too large for the backward's LDS-resident form (RealSR at 272 x 272 tiles and untiled 512 x 512)
int dwgate_fwd_ok(oss_dtype io, int H, int W) {
    if (io != OSS_F16 && io != OSS_BF16 && io != OSS_F32) return 0;
    return (H > 0 && W > 0 && W % 8 == 0 && W <= 512) ? 1 : 0;
}

int dwgate_fwd(oss_dtype io, const void *t, const float *w, const float *bias, void *out, int B, int Hd, int H, int W,
               int64_t tsb, int64_t tsc, int64_t osb, int64_t osc, hipStream_t s) {
    if (!dwgate_fwd_ok(io, H, W)) return OSS_ERR_SHAPE;
    if (io == OSS_F32) return dwgate_fwd_launch<float>(t, w, bias, out, B, Hd, H, W, tsb, tsc, osb, osc, s);
    return io == OSS_F16 ? dwgate_fwd_launch<f16_t>(t, w, bias, out, B, Hd, H, W, tsb, tsc, osb, osc, s)
                         : dwgate_fwd_launch<bf16_t>(t, w, bias, out, B, Hd, H, W, tsb, tsc, osb, osc, s);
}

template <typename T, int MODE>
static int bwd_fused_launch(const void *x, const float *w, const float *bias, const void *dy, void *dx, float *dw, float *db,
                            float *part, int B, int C, int H, int W, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc,
                            int64_t dsb, int64_t dsc, hipStream_t s, const void *dyt = nullptr) {
    constexpr int NCH = MODE == kDwGate ? 2 : 1;
    if (!aligned16({x, dy, dx}, {xsb, xsc, gsb, gsc, dsb, dsc})) return OSS_ERR_SHAPE;
    static LdsGate gate, gate_e, gate_k;
    const size_t smem = fused_lds_bytes(NCH, H, W, sizeof(T));
    const bool edge = dw_edge(W);
    static const bool keep_ok = [] { const char *e = getenv("VMAMBAIR_DW_KEEP"); return !(e && e[0] == '0'); }();   // A-B timing
    // <= 2 groups per lane: the centre rows of x stay in registers (16-bit I/O: the float form's raw rows are twice the registers and
    // end up in scratch memory -- it keeps the row-by-row loads)
    bool keep = false;
    auto kern = edge ? oss_dwconv3x3_bwd_fused_kernel<T, MODE, true, false> : oss_dwconv3x3_bwd_fused_kernel<T, MODE, false, false>;
    LdsGate *g = edge ? &gate_e : &gate;
    if constexpr (sizeof(T) == 2) {
        keep = keep_ok && !edge && (W / 8) * H <= 512;
        if (keep) { kern = oss_dwconv3x3_bwd_fused_kernel<T, MODE, false, true>; g = &gate_k; }
        if constexpr (MODE == kDwSilu) {
            static LdsGate gate_m;
            if (keep && dyt) { kern = oss_dwconv3x3_bwd_fused_kernel<T, MODE, false, true, true>; g = &gate_m; }
        }
    }
    if (const int e = g->ensure(reinterpret_cast<const void *>(kern), smem + 4 * NCH * 10 * sizeof(float))) return e;
    hipLaunchKernelGGL(kern, dim3(C / NCH, B), dim3(256), smem, s, reinterpret_cast<const T *>(x), w, bias,
                       reinterpret_cast<const T *>(dy), reinterpret_cast<T *>(dx), part, C, H, W, xsb, xsc, gsb, gsc, dsb, dsc,
                       reinterpret_cast<const T *>(dyt));
    if (defer_finish())
        defer_sum(part, B, (size_t)C * 10, (size_t)C * (db ? 10 : 9), dw, (size_t)C * 9, db);
    else
        hipLaunchKernelGGL(oss_dwconv3x3_wgrad_finish, dim3((C * 10 + 255) / 256), dim3(256), 0, s, part, dw, db, B, C);
    return (int)hipGetLastError();
}

// ---- flat2 forms: the depth-wise convolution of SS2D_1 together with cross_scan_2d's two forward flattenings -----------------------
// forward: x2 (B, 2, C, H * W) = [silu(conv(x)) row-major | the same column-major]; backward: dy = g2[:, 0] + transpose(g2[:, 1])
int dwconv3x3_flat2_ok(oss_dtype io, int H, int W) {
    if (!dwconv3x3_fused_ok(io, H, W, 1) || H % 8 != 0) return 0;
    const int lpr = W / 8;
    return (256 % lpr == 0 && (256 / lpr) % 8 == 0) ? 1 : 0;
}

template <typename T>
static int flat2_fwd_launch(const void *x, const float *w, const float *bias, void *x2, int B, int C, int H, int W, int64_t xsb,
                            int64_t xsc, hipStream_t s) {
    const int64_t L = (int64_t)H * W;
    const T *xp = reinterpret_cast<const T *>(x);
    T *yp = reinterpret_cast<T *>(x2);
    if (!wide_ok<T>(W, {xp, yp}, {xsb, xsc, L})) return OSS_ERR_SHAPE;
    dim3 grid(((W / 8) * H + 255) / 256, C, B);
    hipLaunchKernelGGL((oss_dwconv3x3_wide_kernel<T, false>), grid, dim3(256), 0, s, xp, w, bias, yp, C, H, W, xsb, xsc, 2 * C * L, L, 0,
                       (T *)nullptr, 1, yp + C * L);
    return (int)hipGetLastError();
}

int dwconv3x3_silu_flat2_fwd(oss_dtype io, const void *x, const float *w, const float *bias, void *x2, int B, int C, int H, int W,
                             int64_t xsb, int64_t xsc, hipStream_t s) {
    if (!dwconv3x3_flat2_ok(io, H, W)) return OSS_ERR_SHAPE;
    if (io == OSS_F32) return flat2_fwd_launch<float>(x, w, bias, x2, B, C, H, W, xsb, xsc, s);
    return io == OSS_F16 ? flat2_fwd_launch<f16_t>(x, w, bias, x2, B, C, H, W, xsb, xsc, s)
                         : flat2_fwd_launch<bf16_t>(x, w, bias, x2, B, C, H, W, xsb, xsc, s);
}

int dwconv3x3_silu_flat2_bwd(oss_dtype io, const void *x, const float *w, const float *bias, const void *g2, void *dx, float *dw,
                             float *db, float *part, int B, int C, int H, int W, int64_t xsb, int64_t xsc, int64_t dsb, int64_t dsc,
                             hipStream_t s) {
    if (!dwconv3x3_flat2_ok(io, H, W)) return OSS_ERR_SHAPE;
    const int64_t L = (int64_t)H * W;
    const char *gt = reinterpret_cast<const char *>(g2) + (size_t)C * L * (io == OSS_F32 ? 4 : 2);
    if (io == OSS_F32) return bwd_fused_launch<float, kDwSilu>(x, w, bias, g2, dx, dw, db, part, B, C, H, W, xsb, xsc, 2 * C * L, L, dsb, dsc, s, gt);
    return io == OSS_F16 ? bwd_fused_launch<f16_t, kDwSilu>(x, w, bias, g2, dx, dw, db, part, B, C, H, W, xsb, xsc, 2 * C * L, L, dsb, dsc, s, gt)
                         : bwd_fused_launch<bf16_t, kDwSilu>(x, w, bias, g2, dx, dw, db, part, B, C, H, W, xsb, xsc, 2 * C * L, L, dsb, dsc, s, gt);
}

int dwconv3x3_bwd_fused(oss_dtype io, int mode, const void *x, const float *w, const float *bias, const void *dy, void *dx,
                        float *dw, float *db, float *part, int B, int C, int H, int W, int64_t xsb, int64_t xsc, int64_t gsb,
                        int64_t gsc, int64_t dsb, int64_t dsc, hipStream_t s) {
    const int nch = mode == kDwGate ? 2 : 1;
    if (!dwconv3x3_fused_ok(io, H, W, nch) || C % nch != 0) return OSS_ERR_SHAPE;
    if (io == OSS_F32)
        return mode == kDwGate ? bwd_fused_launch<float, kDwGate>(x, w, bias, dy, dx, dw, db, part, B, C, H, W, xsb, xsc, gsb, gsc, dsb, dsc, s)
                               : bwd_fused_launch<float, kDwSilu>(x, w, bias, dy, dx, dw, db, part, B, C, H, W, xsb, xsc, gsb, gsc, dsb, dsc, s);
    if (mode == kDwGate)
        return io == OSS_F16 ? bwd_fused_launch<f16_t, kDwGate>(x, w, bias, dy, dx, dw, db, part, B, C, H, W, xsb, xsc, gsb, gsc, dsb, dsc, s)
                             : bwd_fused_launch<bf16_t, kDwGate>(x, w, bias, dy, dx, dw, db, part, B, C, H, W, xsb, xsc, gsb, gsc, dsb, dsc, s);
    return io == OSS_F16 ? bwd_fused_launch<f16_t, kDwSilu>(x, w, bias, dy, dx, dw, db, part, B, C, H, W, xsb, xsc, gsb, gsc, dsb, dsc, s)
                         : bwd_fused_launch<bf16_t, kDwSilu>(x, w, bias, dy, dx, dw, db, part, B, C, H, W, xsb, xsc, gsb, gsc, dsb, dsc, s);
}

template <typename T>
static int wgrad_launch(const void *x, const void *dy, float *dw, float *db, float *part, int B, int C, int H, int W,
                        int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, hipStream_t s, const void *pre, void *dpre) {
    const T *xp = reinterpret_cast<const T *>(x);
    const T *gp = reinterpret_cast<const T *>(dy);
    const T *prp = reinterpret_cast<const T *>(pre);
    T *dpp = reinterpret_cast<T *>(dpre);
    const uintptr_t amask = sizeof(T) == 4 ? 15u : 7u;
    const bool vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(xp) | reinterpret_cast<uintptr_t>(gp) | reinterpret_cast<uintptr_t>(prp) |
                                         reinterpret_cast<uintptr_t>(dpp)) & amask) == 0 &&
                     (xsb % 4 == 0) && (xsc % 4 == 0) && (gsb % 4 == 0) && (gsc % 4 == 0);
    bool wide = false;
    wide = wide_ok<T>(W, {xp, gp, prp, dpp}, {xsb, xsc, gsb, gsc});
    {
        if (wide)
        {
            if ((64 % (W / 8)) != 0)
                hipLaunchKernelGGL((oss_dwconv3x3_wgrad_wide_kernel<T, true>), dim3(C, B), dim3(256), 0, s, xp, gp, part, C, H, W, xsb, xsc, gsb, gsc, prp, dpp);
            else
                hipLaunchKernelGGL((oss_dwconv3x3_wgrad_wide_kernel<T, false>), dim3(C, B), dim3(256), 0, s, xp, gp, part, C, H, W, xsb, xsc, gsb, gsc, prp, dpp);
        }
    }
    if (wide) {
    } else if (vec)
        hipLaunchKernelGGL((oss_dwconv3x3_wgrad_kernel<T, true>), dim3(C, B), dim3(256), 0, s, xp, gp, part, C, H, W, xsb, xsc, gsb, gsc, prp, dpp);
    else
        hipLaunchKernelGGL((oss_dwconv3x3_wgrad_kernel<T, false>), dim3(C, B), dim3(256), 0, s, xp, gp, part, C, H, W, xsb, xsc, gsb, gsc, prp, dpp);
    if (defer_finish())
        defer_sum(part, B, (size_t)C * 10, (size_t)C * (db ? 10 : 9), dw, (size_t)C * 9, db);
    else
        hipLaunchKernelGGL(oss_dwconv3x3_wgrad_finish, dim3((C * 10 + 255) / 256), dim3(256), 0, s, part, dw, db, B, C);
    return (int)hipGetLastError();
}

int dwconv3x3_wgrad(oss_dtype io, const void *x, const void *dy, float *dw, float *db, float *part, int B, int C, int H,
                    int W, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, hipStream_t s, const void *pre, void *dpre) {
    if ((pre != nullptr) != (dpre != nullptr)) return OSS_ERR_NULL;
    switch (io) {
        case OSS_F32: return wgrad_launch<float>(x, dy, dw, db, part, B, C, H, W, xsb, xsc, gsb, gsc, s, pre, dpre);
        case OSS_F16: return wgrad_launch<f16_t>(x, dy, dw, db, part, B, C, H, W, xsb, xsc, gsb, gsc, s, pre, dpre);
        case OSS_BF16: return wgrad_launch<bf16_t>(x, dy, dw, db, part, B, C, H, W, xsb, xsc, gsb, gsc, s, pre, dpre);
    }
    return OSS_ERR_SHAPE;
}

}  // namespace oss
