// oss_effn.hip -- the second half of an OSS block, forward only, as ONE launch (inference: nothing is kept for a backward):
//   out = x + project_out( gelu(x1) * x2 ),   x1, x2 = dwconv3x3( project_in( norm2(x) ) ).chunk(2, dim=1)
//   (SRGAN/VmambaIR/archs/MambaSISR6_arch.py: FeedForward :201-218, the block's `x + ffn(norm2(x))` :513-516, LayerNorm :144-195;
//    the RealSR and Deraining trees carry the same module.)
// The launch-per-layer chain moves the 2h-channel tensor t = project_in(..) through memory twice and the h-channel gate output
// twice -- at the untiled RealSR plane (dim 96, 512 x 512, fp16) 100 + 93 + 112 us for 50 MB of block input and 50 MB of block
// output (profiles/r06_rocprof_realsr_untiled_steady_state.txt).  Here a workgroup owns a TH x TW pixel tile of one image:
//   0. the tile of x with a one-pixel halo goes into LDS ([channel][halo pixel], 8-byte pieces, zeros outside the image) and is
//      LayerNorm-ed there over the channels (two-pass fp32 statistics as oss_ln_nchw_fwd_kernel; pixels outside the image stay 0:
//      the depth-wise convolution pads t with zeros, and without a project_in bias t = W 0 = 0 there);
//   then, per chunk of 16 channel PAIRS (c, c + h) -- the two operands of the gate --
//   1. GEMM 1 (MFMA 32x32x16): one 32-row tile = the 16 x1 rows and the 16 x2 rows of W_in, over all halo pixels; the activation
//      operand comes out of the [channel][pixel] LDS image by ds_read_b64_tr_b16 as in oss_conv1x1_wg.hip; t (rounded to the I/O
//      type, as the chain stores it) goes to LDS;
//   2. every thread convolves ONE 8-pixel row piece of one pair (both planes, fp32 taps and accumulation) and gates it: the
//      h-channel operand of project_out, rounded to the I/O type, in LDS;
//   3. GEMM 2: a wave owns 32 output pixels and all D output rows, one k-step of 16 per chunk, accumulators in registers over all
//      chunks;
//   4. the sums leave through LDS as fp32, meet the skip connection (x, 16 bytes per lane along the image row) and are rounded once.
// The halo costs GEMM 1 (TH + 2)(TW + 8) / (TH TW) = 1.9 x the MFMAs of the plain product -- on a pipe that is a quarter busy; the
// depth-wise taps and the gate (the VALU work that bounds the kernel) are computed for output pixels only.
// 16-bit weights: the caller hands W_in and W_out already rounded to the I/O type (what the chain's kernels do per fragment with
// v_cvt_pk at every use; in inference the weights are constants and the rounding is done once), W_out rows padded with zeros to a
// multiple of 16 columns.
#include <initializer_list>
#include "oss_device.h"
#include "oss_host.h"
#include "oss_mfma.h"
#include "oss_stencil.h"

namespace oss {

typedef short s16x4e __attribute__((ext_vector_type(4)));

struct EffnArgs {
    const void *x;           // (B, D, H, W), strides xsb / xsc, rows contiguous
    void *out;               // same layout, strides osb / osc
    const float *ln_w, *ln_b;   // norm2 (ln_b NULL: BiasFree)
    const void *w_in;        // (2 h, D) I/O type
    const float *w_dw;       // (2 h, 9)
    const void *w_out;       // (D, HP) I/O type, HP = h rounded up to 16, zero padded
    int h, HP, H, W;
    int64_t xsb, xsc, osb, osc;
    float eps;
    int tiles_x;
};

template <typename T, int KS, int TH, int TW>
struct EffnGeo {
    static constexpr int D = 16 * KS;
    static constexpr int SP = TW + 8;              // stored columns of a halo row: image columns w0 - 4 .. w0 + TW + 3
    static constexpr int HR = TH + 2;
    static constexpr int NQ = HR * SP;             // stored halo pixels (flat index q = hr * SP + sc)
    static constexpr int N1T = (NQ + 31) / 32;     // GEMM 1 column tiles
    static constexpr int XP = ((N1T * 32 + 127) / 128) * 128 + 32;   // xs row pitch: 64 bytes past a 256-byte bank sweep (oss_conv1x1_wg.hip)
    static constexpr int TP = N1T * 32 + 8;        // ts row pitch
    static constexpr int N2 = TH * TW;             // output pixels
    static constexpr int N2T = N2 / 32;
    static constexpr int GP = ((N2 + 127) / 128) * 128 + 32;         // gs row pitch
    static constexpr int OP = N2 + 4;              // output staging pitch (floats)
    static constexpr int MT2 = (D + 31) / 32;
    static constexpr size_t xs_bytes = (size_t)D * XP * sizeof(T);
    static constexpr size_t ts_bytes = (size_t)32 * TP * sizeof(T);
    static constexpr size_t gs_bytes = (size_t)16 * GP * sizeof(T);
    static constexpr size_t ln_bytes = (size_t)2 * D * sizeof(float);
    static constexpr size_t lds_bytes = xs_bytes + ts_bytes + gs_bytes + ln_bytes;
    static_assert(N2T == 4, "a wave owns one 32-pixel column tile of GEMM 2");
    static_assert(16 * TH * (TW / 8) == 256, "one (pair, row, 8-pixel piece) per thread and chunk");
    static_assert((size_t)D * OP * sizeof(float) <= xs_bytes, "the output staging tile reuses the activation image");
    static_assert((size_t)4 * NQ * sizeof(float) <= ts_bytes, "the LayerNorm partial sums reuse the t tile");
    static_assert(NQ % 4 == 0 && NQ / 4 <= 64, "LayerNorm: quads of pixels x 4 channel parts on 256 threads");
};

template <typename T, int KS, int TH, int TW>
__global__ void __launch_bounds__(256, 2)
oss_effn_fwd_kernel(EffnArgs a) {
    using G = EffnGeo<T, KS, TH, TW>;
    constexpr int D = G::D, SP = G::SP, HR = G::HR, NQ = G::NQ, N1T = G::N1T, XP = G::XP, TP = G::TP, GP = G::GP, OP = G::OP, MT2 = G::MT2;
    extern __shared__ __attribute__((aligned(16))) unsigned char effn_smem[];
    T *xs = reinterpret_cast<T *>(effn_smem);                                   // [D][XP]
    T *ts = reinterpret_cast<T *>(effn_smem + G::xs_bytes);                     // [32][TP]: rows 0..15 = x1 of the chunk's pairs, 16..31 = x2
    T *gs = reinterpret_cast<T *>(effn_smem + G::xs_bytes + G::ts_bytes);       // [16][GP]
    float *lnw_s = reinterpret_cast<float *>(effn_smem + G::xs_bytes + G::ts_bytes + G::gs_bytes), *lnb_s = lnw_s + D;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int h0 = ty * TH, w0 = tx * TW, H = a.H, W = a.W, h = a.h;
    const T *xb = reinterpret_cast<const T *>(a.x) + b * a.xsb;
    const T *w_in = reinterpret_cast<const T *>(a.w_in), *w_out = reinterpret_cast<const T *>(a.w_out);
    const int col = lane & 31, kg = lane >> 5;

    // weight fragments: W_in's (GEMM 1, used at the top of a chunk) are requested during the previous chunk's step 2; the depth-wise
    // taps and W_out's fragments (steps 2 and 3) at the top of their own chunk, behind GEMM 1.  (All three one chunk ahead in two
    // register sets: 108 VGPRs, and the D = 96 bf16 instantiation spilled.)
    u32x4 a1[KS], a2[MT2];
    float k1[9], k2[9];
    const int pr = tid >> 4, orow = (tid & 15) / (TW / 8), seg = (tid & 15) % (TW / 8);   // this thread's piece in step 2
    auto issue_a1 = [&](int c0) {
        const int cr = c0 + (col & 15);
        const int ch = min(col < 16 ? cr : h + cr, 2 * h - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a1[ks] = *reinterpret_cast<const u32x4 *>(w_in + (size_t)ch * D + ks * 16 + kg * 8);
    };
    auto issue_rest = [&](int c0) {
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) {
            const int dm = min(mt * 32 + col, D - 1);
            a2[mt] = *reinterpret_cast<const u32x4 *>(w_out + (size_t)dm * a.HP + c0 + kg * 8);
        }
        const int cp = min(c0 + pr, h - 1);
#pragma unroll
        for (int i = 0; i < 9; ++i) { k1[i] = a.w_dw[(size_t)cp * 9 + i]; k2[i] = a.w_dw[(size_t)(h + cp) * 9 + i]; }
    };
    issue_a1(0);
    if (tid < D) { lnw_s[tid] = a.ln_w[tid]; lnb_s[tid] = a.ln_b ? a.ln_b[tid] : 0.f; }

    // 0a. the halo tile of x: 8-byte pieces (4 pixels), (channel, halo row, piece) on consecutive threads
    {
        constexpr int PPR = SP / 4, TOTAL = D * HR * PPR, NIT = (TOTAL + 255) / 256, GROUP = 8;
#pragma unroll
        for (int i0 = 0; i0 < NIT; i0 += GROUP) {
            u32x2 q[GROUP];
#pragma unroll
            for (int i = 0; i < GROUP; ++i) {
                if (i0 + i < NIT) {
                    const int idx = min(tid + (i0 + i) * 256, TOTAL - 1);
                    const int d = idx / (HR * PPR), r = idx - d * (HR * PPR), hr = r / PPR, pc = r - hr * PPR;
                    const int ih = h0 - 1 + hr, iw = w0 - 4 + 4 * pc;
                    const bool in = ih >= 0 && ih < H && iw >= 0 && iw < W;
                    const T *src = xb + d * a.xsc + (int64_t)(in ? ih : 0) * W + (in ? iw : 0);
                    q[i] = *reinterpret_cast<const u32x2 *>(src);
                    if (!in) q[i] = u32x2{0u, 0u};
                }
            }
#pragma unroll
            for (int i = 0; i < GROUP; ++i) {
                if (i0 + i < NIT) {
                    const int idx = tid + (i0 + i) * 256;
                    if (TOTAL % 256 == 0 || idx < TOTAL) {
                        const int d = idx / (HR * PPR), r = idx - d * (HR * PPR), hr = r / PPR, pc = r - hr * PPR;
                        *reinterpret_cast<u32x2 *>(xs + d * XP + hr * SP + 4 * pc) = q[i];
                    }
                }
            }
        }
    }
    __syncthreads();
    // 0b. LayerNorm over the channels, per stored pixel: a thread owns FOUR adjacent pixels and every 4th channel
    {
        constexpr int QPT = NQ / 4, NPART = 4, CPT = D / NPART;
        static_assert(D % NPART == 0, "channels per part");
        float *red = reinterpret_cast<float *>(ts);   // [NPART][NQ]
        const bool act = tid < QPT * NPART;
        const int quad = act ? tid % QPT : 0, part = act ? tid / QPT : 0, px = 4 * quad;
        const int hr = px / SP, sc = px - hr * SP, ih = h0 - 1 + hr, iw = w0 - 4 + sc;
        const bool in = ih >= 0 && ih < H && iw >= 0 && iw < W;   // (W % 4 == 0: a quad is inside or outside as a whole)
        // (the pixel's values are read from LDS again in each of the three passes: 24 x 4 of them held in registers over the passes
        // pushed the D = 96 instantiation into scratch memory)
        auto ld4 = [&](int i, float (&v)[4]) {
            const u32x2 q = *reinterpret_cast<const u32x2 *>(xs + (part + i * NPART) * XP + px);
            unpack2<T>(q.x, v[0], v[1]); unpack2<T>(q.y, v[2], v[3]);
        };
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int i = 0; i < CPT; ++i) {
            float v[4];
            ld4(i, v);
#pragma unroll
            for (int u = 0; u < 4; ++u) sum[u] += v[u];
        }
        if (act) *reinterpret_cast<f32x4 *>(red + part * NQ + px) = f32x4{sum[0], sum[1], sum[2], sum[3]};
        __syncthreads();
        float mu[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NPART; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(red + q * NQ + px);
            mu[0] += t.x; mu[1] += t.y; mu[2] += t.z; mu[3] += t.w;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) mu[u] /= (float)D;
        __syncthreads();
        float sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int i = 0; i < CPT; ++i) {
            float v[4];
            ld4(i, v);
#pragma unroll
            for (int u = 0; u < 4; ++u) { const float d = v[u] - mu[u]; sq[u] = __builtin_fmaf(d, d, sq[u]); }
        }
        if (act) *reinterpret_cast<f32x4 *>(red + part * NQ + px) = f32x4{sq[0], sq[1], sq[2], sq[3]};
        __syncthreads();
        float rstd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NPART; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(red + q * NQ + px);
            rstd[0] += t.x; rstd[1] += t.y; rstd[2] += t.z; rstd[3] += t.w;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) rstd[u] = in ? 1.0f / sqrtf(rstd[u] / (float)D + a.eps) : 0.f;
        const bool with_bias = a.ln_b != nullptr;
        float mu_c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) mu_c[u] = with_bias ? mu[u] : 0.f;
        if (act) {
#pragma unroll 4
            for (int i = 0; i < CPT; ++i) {
                const int c = part + i * NPART;
                const float wc = lnw_s[c], bc = in ? lnb_s[c] : 0.f;
                float v[4], o[4];
                ld4(i, v);
#pragma unroll
                for (int u = 0; u < 4; ++u) o[u] = (v[u] - mu_c[u]) * rstd[u] * wc + bc;   // outside the image: rstd = bc = 0 -> 0
                *reinterpret_cast<u32x2 *>(xs + c * XP + px) = u32x2{pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3])};
            }
        }
        __syncthreads();   // (also: red[] = ts has been read by everyone)
    }

    const int i16 = lane & 15, g = lane >> 4;
    const int tr1 = (8 * (g >> 1) + (i16 >> 2)) * XP + 16 * (g & 1) + 4 * (i16 & 3);
    const int tr2 = (8 * (g >> 1) + (i16 >> 2)) * GP + 16 * (g & 1) + 4 * (i16 & 3) + 32 * wave;
    constexpr int NPW = (N1T + 3) / 4;   // GEMM 1 column tiles per wave
    f32x16 acc2[MT2];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[mt][r] = 0.f;

    for (int c0 = 0; c0 < h; c0 += 16) {
        issue_rest(c0);
        // 1. t[32 rows][halo pixels] = W_in rows of the chunk x the normalised tile
        {
            const bool rv = c0 + (col & 15) < h;
            s16x8 af[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) af[ks] = rv ? __builtin_bit_cast(s16x8, a1[ks]) : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int jj = 0; jj < NPW; ++jj) {
                const int j = wave + 4 * jj;
                if (j < N1T) {   // (wave-uniform)
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const T *bp = xs + ks * 16 * XP + 32 * j + tr1;
                        const s16x4e lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4e *)(bp));
                        const s16x4e hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4e *)(bp + 4 * XP));
                        const s16x8 bf = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                        acc = Mfma<T>::run(af[ks], bf, acc);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
                        ts[row * TP + 32 * j + col] = from_f32<T>(acc[r]);
                    }
                }
            }
        }
        __syncthreads();
        if (c0 + 16 < h) issue_a1(c0 + 16);
        // 2. depth-wise 3x3 of both planes of the thread's pair on its 8-pixel row piece, then the gate
        {
            float x12[2][8];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const T *tp = ts + (pl * 16 + pr) * TP;
                const float *k = pl ? k2 : k1;
#pragma unroll
                for (int j = 0; j < 8; ++j) x12[pl][j] = 0.f;
#pragma unroll
                for (int dr = 0; dr < 3; ++dr) {
                    const int q = (orow + dr) * SP + 4 + 8 * seg;
                    const u32x2 qa = *reinterpret_cast<const u32x2 *>(tp + q - 4), qb = *reinterpret_cast<const u32x2 *>(tp + q),
                                qc = *reinterpret_cast<const u32x2 *>(tp + q + 4), qd = *reinterpret_cast<const u32x2 *>(tp + q + 8);
                    float v[10], dump;
                    unpack2<T>(qa.y, dump, v[0]);
                    unpack2<T>(qb.x, v[1], v[2]); unpack2<T>(qb.y, v[3], v[4]);
                    unpack2<T>(qc.x, v[5], v[6]); unpack2<T>(qc.y, v[7], v[8]);
                    unpack2<T>(qd.x, v[9], dump);
                    const float k0 = k[dr * 3], k1 = k[dr * 3 + 1], k2 = k[dr * 3 + 2];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        x12[pl][j] = __builtin_fmaf(k0, v[j], __builtin_fmaf(k1, v[j + 1], __builtin_fmaf(k2, v[j + 2], x12[pl][j])));
                }
            }
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float cdf, pdf;
                gelu_parts(x12[0][j], cdf, pdf);
                o[j] = x12[0][j] * cdf * x12[1][j];
            }
            *reinterpret_cast<u32x4 *>(gs + pr * GP + orow * TW + 8 * seg) =
                u32x4{pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]), pack2<T>(o[4], o[5]), pack2<T>(o[6], o[7])};
        }
        __syncthreads();
        // 3. out[D rows][the wave's 32 pixels] += W_out[:, chunk] x gate
        {
            const T *bp = gs + tr2;
            const s16x4e lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4e *)(bp));
            const s16x4e hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4e *)(bp + 4 * GP));
            const s16x8 bf = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
                const s16x8 af = mt * 32 + col < D ? __builtin_bit_cast(s16x8, a2[mt]) : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
                acc2[mt] = Mfma<T>::run(af, bf, acc2[mt]);
            }
        }
    }
    __syncthreads();   // everyone is done with xs (GEMM 1 of the last chunk) before it becomes the output staging tile
    // 4. fp32 sums -> LDS -> + x -> one rounding -> 16-byte stores along the image rows
    float *os = reinterpret_cast<float *>(effn_smem);   // [D][OP]
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dm = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (dm < D) os[dm * OP + 32 * wave + col] = acc2[mt][r];
        }
    __syncthreads();
    {
        constexpr int PPR = TW / 8, TOTAL = D * TH * PPR;
        T *ob = reinterpret_cast<T *>(a.out) + b * a.osb;
        for (int idx = tid; idx < TOTAL; idx += 256) {
            const int dm = idx / (TH * PPR), r = idx - dm * (TH * PPR), orw = r / PPR, sg = r - orw * PPR;
            const int ih = h0 + orw, iw = w0 + 8 * sg;
            if (ih < H && iw < W) {
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(os + dm * OP + orw * TW + 8 * sg);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(os + dm * OP + orw * TW + 8 * sg + 4);
                const u32x4 rq = *reinterpret_cast<const u32x4 *>(xb + dm * a.xsc + (int64_t)ih * W + iw);
                float r8[8];
                unpack2<T>(rq.x, r8[0], r8[1]); unpack2<T>(rq.y, r8[2], r8[3]); unpack2<T>(rq.z, r8[4], r8[5]); unpack2<T>(rq.w, r8[6], r8[7]);
                *reinterpret_cast<u32x4 *>(ob + dm * a.osc + (int64_t)ih * W + iw) =
                    u32x4{pack2<T>(v0.x + r8[0], v0.y + r8[1]), pack2<T>(v0.z + r8[2], v0.w + r8[3]),
                          pack2<T>(v1.x + r8[4], v1.y + r8[5]), pack2<T>(v1.z + r8[6], v1.w + r8[7])};
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
static constexpr bool effn_ks_built(int ks) { return ks == 2 || ks == 3 || ks == 4 || ks == 6; }

int effn_fwd_ok(oss_dtype io, int D, int hidden, int H, int W) {
    if (io != OSS_F16 && io != OSS_BF16) return 0;
    if (D % 16 != 0 || !effn_ks_built(D / 16) || hidden < 1 || hidden > 16384) return 0;
    return (H > 0 && W > 0 && W % 8 == 0) ? 1 : 0;
}

template <typename T, int KS>
static int effn_launch(const EffnArgs &a0, int B, hipStream_t s) {
    constexpr int TH = 8, TW = 16;
    using G = EffnGeo<T, KS, TH, TW>;
    EffnArgs a = a0;
    a.tiles_x = (a.W + TW - 1) / TW;
    const int tiles_y = (a.H + TH - 1) / TH;
    static LdsGate gate;
    auto kern = oss_effn_fwd_kernel<T, KS, TH, TW>;
    if (const int e = gate.ensure(reinterpret_cast<const void *>(kern), G::lds_bytes)) return e;
    hipLaunchKernelGGL(kern, dim3(a.tiles_x * tiles_y, B), dim3(256), G::lds_bytes, s, a);
    return (int)hipGetLastError();
}

int effn_fwd(oss_dtype io, const void *x, const float *ln_w, const float *ln_b, const void *w_in, const float *w_dw, const void *w_out,
             void *out, int B, int D, int hidden, int H, int W, int64_t xsb, int64_t xsc, int64_t osb, int64_t osc, float eps,
             hipStream_t s) {
    if (!effn_fwd_ok(io, D, hidden, H, W)) return OSS_ERR_SHAPE;
    if (B <= 0 || B > 65535) return OSS_ERR_SHAPE;
    for (const void *p : {x, (const void *)out, w_in, w_out})
        if (reinterpret_cast<uintptr_t>(p) & 15u) return OSS_ERR_SHAPE;
    for (int64_t st : {xsb, xsc, osb, osc})
        if (st % 8 != 0) return OSS_ERR_SHAPE;
    if ((long)((W + 15) / 16) * ((H + 7) / 8) > 2147483647L) return OSS_ERR_SHAPE;
    EffnArgs a{x, out, ln_w, ln_b, w_in, w_dw, w_out, hidden, (hidden + 15) / 16 * 16, H, W, xsb, xsc, osb, osc, eps, 0};
#define OSS_EFFN(KS_)                                                                         \
    return io == OSS_F16 ? effn_launch<f16_t, KS_>(a, B, s) : effn_launch<bf16_t, KS_>(a, B, s)
    switch (D / 16) {
        case 2: OSS_EFFN(2);
        case 3: OSS_EFFN(3);
        case 4: OSS_EFFN(4);
        case 6: OSS_EFFN(6);
        default: return OSS_ERR_SHAPE;
    }
#undef OSS_EFFN
}

}  // namespace oss
