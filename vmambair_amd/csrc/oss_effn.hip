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
// v_cvt_pk at every use; in inference the weights are constants and the rounding is done once) and padded with zeros to HP = h rounded
// up to 16: W_in and the depth-wise taps as [x1 half: HP rows | x2 half: HP rows], W_out rows as HP columns -- the loop has no masks.
#include <cstdlib>
#include <initializer_list>
#include <type_traits>
#include "oss_device.h"
#include "oss_host.h"
#include "oss_mfma.h"
#include "oss_stencil.h"

namespace oss {

typedef short s16x4e __attribute__((ext_vector_type(4)));

// two fp32 -> one word of two I/O elements, one instruction for both types (oss_device.h's generic pack2 is two v_cvt_f16_f32 and an
// OR for fp16: 48 of this kernel's ~550 VALU instructions per chunk)
template <typename T> __device__ __forceinline__ uint32_t pk2(float lo, float hi) { return pack2<T>(lo, hi); }
template <> __device__ __forceinline__ uint32_t pk2<f16_t>(float lo, float hi) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 h = __builtin_convertvector(f32x2{lo, hi}, f16x2);   // v_cvt_pk_f16_f32 (gfx950), round to nearest even
    uint32_t u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}

struct EffnArgs {
    const void *x;           // (B, D, H, W), strides xsb / xsc, rows contiguous
    void *out;               // same layout, strides osb / osc
    const float *ln_w, *ln_b;   // norm2 (ln_b NULL: BiasFree)
    const void *w_in;        // (2 HP, D) I/O type: rows 0 .. h - 1 = the x1 half, HP .. HP + h - 1 = the x2 half, the rest zero
    const float *w_dw;       // (2 HP, 9), same row order
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
    // a WAVE owns RW = TH / 4 output rows end to end: its halo is RW + 2 stored rows = WQ stored pixels starting at q = wave * RW * SP
    static constexpr int RW = TH / 4;
    static constexpr int WQ = (RW + 2) * SP;
    static constexpr int N1W = WQ / 32;            // GEMM 1 column tiles of a wave (the waves' halos overlap: 4 N1W tiles against NQ / 32)
    // xs: rows of XP = 256 pixel slots, row c rotated by 32 (c & 3) slots: ds_read_b64_tr_b16 is serviced in 32-lane halves over 64
    // banks, a half = two 16-lane groups x four channel rows x 32 bytes -- with the rotation the eight pieces lie at bank bytes
    // 0,64,128,192 | 32,96,160,224 (the rule of oss_conv1x1_wg.hip, whose pitch of 32 slots past a bank sweep does not fit here)
    static constexpr int XP = 256;
    static_assert(NQ <= XP, "a row's stored pixels fit its slots");
    // ts: wave-private [32 channels][TP]; 8-byte writes (16-lane groups = 16 channels, banks (a / 4) mod 32): rows 50 words apart put
    // the 16 pieces on 32 different banks; the 8-byte reads of step 2 are two deep at every pitch (tools: brute force in DESIGN.md)
    static constexpr int TP = WQ + 4;
    static constexpr int N2 = TH * TW;             // output pixels
    static constexpr int N2T = N2 / 32;
    static constexpr int GP = 32;                  // gs row pitch (wave-private [16][32]: rows 64 bytes apart = the transpose-read's bank rule)
    static constexpr int OP = N2 + 4;              // output staging pitch (floats)
    static constexpr int MT2 = (D + 31) / 32;
    static constexpr size_t xs_bytes = (size_t)D * XP * sizeof(T);
    static constexpr size_t ts_bytes = (size_t)4 * 32 * TP * sizeof(T);
    static constexpr size_t gs_bytes = (size_t)4 * 16 * GP * sizeof(T);
    static constexpr size_t lds_bytes = xs_bytes + ts_bytes + gs_bytes;
    static_assert((size_t)2 * D * sizeof(float) <= gs_bytes, "the LayerNorm weight / bias image lies in the gate tiles until the main loop");
    static_assert(N2T == 4 && RW * TW == 32, "a wave owns one 32-pixel column tile of GEMM 2 = its own output rows");
    static_assert(WQ % 32 == 0, "a wave's halo pixels are whole MFMA column tiles");
    static_assert(16 * RW * (TW / 8) == 64, "one (pair, row, 8-pixel piece) per lane and chunk");
    static_assert((size_t)D * OP * sizeof(float) <= xs_bytes + ts_bytes, "the output staging tile reuses the activation image and the t tiles");
    static_assert((size_t)4 * NQ * sizeof(float) <= ts_bytes, "the LayerNorm partial sums reuse the t tiles");
    static_assert(NQ % 4 == 0 && NQ / 4 <= 64, "LayerNorm: quads of pixels x 4 channel parts on 256 threads");
};

// (the narrow forms fit three workgroups per CU in LDS: registers for three waves per SIMD)
template <typename T, int KS, int TH, int TW>
__global__ void __launch_bounds__(256, (KS > 3 || std::is_same<T, bf16_t>::value) ? 2 : 3)   // (bf16 unpacking: 26 registers more)
oss_effn_fwd_kernel(EffnArgs a) {
    using G = EffnGeo<T, KS, TH, TW>;
    constexpr int D = G::D, SP = G::SP, HR = G::HR, NQ = G::NQ, N1W = G::N1W, RW = G::RW, XP = G::XP, TP = G::TP, GP = G::GP, OP = G::OP, MT2 = G::MT2;
    extern __shared__ __attribute__((aligned(16))) unsigned char effn_smem[];
    T *xs = reinterpret_cast<T *>(effn_smem);                                   // [D][XP]
    T *ts_all = reinterpret_cast<T *>(effn_smem + G::xs_bytes);                 // [4 waves][32][TP]: rows 0..15 = x1 of the chunk's pairs, 16..31 = x2
    T *gs_all = reinterpret_cast<T *>(effn_smem + G::xs_bytes + G::ts_bytes);   // [4 waves][16][GP]
    float *lnw_s = reinterpret_cast<float *>(gs_all), *lnb_s = lnw_s + D;        // (prologue only)
    auto xoff = [](int c, int q) { return c * XP + ((q + 32 * (c & 3)) & (XP - 1)); };   // element (channel c, stored pixel q) of xs
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int h0 = ty * TH, w0 = tx * TW, H = a.H, W = a.W, h = a.h;
    const T *xb = reinterpret_cast<const T *>(a.x) + b * a.xsb;
    const T *w_in = reinterpret_cast<const T *>(a.w_in), *w_out = reinterpret_cast<const T *>(a.w_out);
    const int col = lane & 31, kg = lane >> 5;

    // weight fragments: W_in's (GEMM 1, used at the top of a chunk) are requested during the previous chunk's step 2; the depth-wise
    // taps and W_out's fragments (steps 2 and 3) at the top of their own chunk, behind GEMM 1.  (All three one chunk ahead in two
    // register sets: 108 VGPRs, and the D = 96 bf16 instantiation spilled.)
    u32x4 a1[KS];
    struct Rest { u32x4 a2[MT2]; float k1[9], k2[9]; };
    const int pr = lane >> 2, orow = (lane & 3) / (TW / 8), seg = (lane & 3) % (TW / 8);   // this lane's piece in step 2 (orow: of the wave's RW rows)
    auto issue_a1 = [&](int c0) {
        const int ch = c0 + (col & 15) + (col < 16 ? 0 : a.HP);   // (zero rows behind each half: no masks in the loop)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a1[ks] = *reinterpret_cast<const u32x4 *>(w_in + (size_t)ch * D + ks * 16 + kg * 8);
    };
    auto issue_rest = [&](int c0, Rest &f) {
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) {
            const int dm = min(mt * 32 + col, D - 1);
            f.a2[mt] = *reinterpret_cast<const u32x4 *>(w_out + (size_t)dm * a.HP + c0 + kg * 8);
        }
        const int cp = c0 + pr;
#pragma unroll
        for (int i = 0; i < 9; ++i) { f.k1[i] = a.w_dw[(size_t)cp * 9 + i]; f.k2[i] = a.w_dw[(size_t)(a.HP + cp) * 9 + i]; }
    };
    Rest cur;
    issue_a1(0);
    issue_rest(0, cur);
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
                        *reinterpret_cast<u32x2 *>(xs + xoff(d, hr * SP + 4 * pc)) = q[i];
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
        float *red = reinterpret_cast<float *>(ts_all);   // [NPART][NQ]
        const bool act = tid < QPT * NPART;
        const int quad = act ? tid % QPT : 0, part = act ? tid / QPT : 0, px = 4 * quad;
        const int hr = px / SP, sc = px - hr * SP, ih = h0 - 1 + hr, iw = w0 - 4 + sc;
        const bool in = ih >= 0 && ih < H && iw >= 0 && iw < W;   // (W % 4 == 0: a quad is inside or outside as a whole)
        // (the pixel's values are read from LDS again in each of the three passes: 24 x 4 of them held in registers over the passes
        // pushed the D = 96 instantiation into scratch memory)
        auto ld4 = [&](int i, float (&v)[4]) {
            const u32x2 q = *reinterpret_cast<const u32x2 *>(xs + xoff(part + i * NPART, px));
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
                *reinterpret_cast<u32x2 *>(xs + xoff(c, px)) = u32x2{pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3])};
            }
        }
        __syncthreads();   // (also: red[] = the t tiles has been read by everyone)
    }

    // From here to the epilogue the waves do not meet: wave w computes t on ITS halo (stored rows RW w .. RW w + RW + 1 = column tiles
    // of GEMM 1 starting at pixel q0), convolves and gates ITS RW output rows, and multiplies them into ITS 32 output pixels.  The
    // neighbouring waves' halos overlap by two stored rows (GEMM 1 does 4 N1W column tiles instead of NQ / 32 -- MFMAs on a pipe
    // that is a tenth busy) and in exchange the two workgroup barriers per chunk of the first version are gone: with two waves per
    // SIMD, every barrier left the SIMD idle whenever both of its waves were waiting (profiles/r06_pmc_sq_effn_v1_barriers.txt:
    // VALU busy 27 % of the wave cycles, 36 % waiting, 27 % neither).  LDS operations of one wave execute in order.
    T *ts = ts_all + wave * 32 * TP, *gs = gs_all + wave * 16 * GP;
    const int q0 = wave * RW * SP;
    const int i16 = lane & 15, g = lane >> 4;
    const int tr1_row = (8 * (g >> 1) + (i16 >> 2)) * XP;
    int tr1_px[N1W];   // the lane's four pixels of column tile j, rotated like its channel row (rows r and r + 4 k share r & 3)
#pragma unroll
    for (int j = 0; j < N1W; ++j) tr1_px[j] = (q0 + 32 * j + 16 * (g & 1) + 4 * (i16 & 3) + 32 * (i16 >> 2)) & (XP - 1);
    const int tr2 = (8 * (g >> 1) + (i16 >> 2)) * GP + 16 * (g & 1) + 4 * (i16 & 3);
    f32x16 acc2[MT2];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[mt][r] = 0.f;

    for (int c0 = 0; c0 < h; c0 += 16) {
        const bool more = c0 + 16 < h;
        // 1. t[32 rows][the wave's halo pixels] = W_in rows of the chunk x the normalised tile
        {
            // pixels are the ROWS of this product (activations = operand A) and the chunk's 32 channels its columns: a lane then holds four
            // consecutive pixels of ONE channel per register quad and t goes to LDS as 8-byte pieces (as [channel][pixel] rows with
            // 2-byte writes the stores alone were 190 LDS cycles per wave and chunk, two lanes per bank word)
            s16x8 wf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf[ks] = __builtin_bit_cast(s16x8, a1[ks]);
            // the wave's N1W column tiles side by side (N1W accumulators): one tile after the other on ONE accumulator was a chain of
            // dependent MFMAs, each behind the LDS latency of its own two transpose-reads (lgkmcnt(1) before every MFMA in the ISA)
            f32x16 acc[N1W];
#pragma unroll
            for (int j = 0; j < N1W; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            auto rd = [&](int ks, int j) -> s16x8 {
                const T *bp = xs + ks * 16 * XP + tr1_row + tr1_px[j];
                const s16x4e lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4e *)(bp));
                const s16x4e hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4e *)(bp + 4 * XP));
                return s16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            };
            s16x8 xf[N1W];
#pragma unroll
            for (int j = 0; j < N1W; ++j) xf[j] = rd(0, j);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s16x8 xn[N1W];
                if (ks + 1 < KS) {
#pragma unroll
                    for (int j = 0; j < N1W; ++j) xn[j] = rd(ks + 1, j);
                }
#pragma unroll
                for (int j = 0; j < N1W; ++j) acc[j] = Mfma<T>::run(xf[j], wf[ks], acc[j]);
                if (ks + 1 < KS) {
#pragma unroll
                    for (int j = 0; j < N1W; ++j) xf[j] = xn[j];
                }
            }
#pragma unroll
            for (int j = 0; j < N1W; ++j)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)   // rows (= pixels) 8 rg + 4 kg .. + 3 of the tile, column (= channel) col
                    *reinterpret_cast<u32x2 *>(ts + col * TP + 32 * j + 8 * rg + 4 * kg) =
                        u32x2{pk2<T>(acc[j][4 * rg], acc[j][4 * rg + 1]), pk2<T>(acc[j][4 * rg + 2], acc[j][4 * rg + 3])};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (more) issue_a1(c0 + 16);
        // 2. depth-wise 3x3 of both planes of the lane's pair on its 8-pixel row piece, then the gate
        {
            // every LDS read of the piece first (both planes, three rows, four 8-byte pieces each), then the arithmetic
            u32x2 raw[2][3][4];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int dr = 0; dr < 3; ++dr) {
                    const T *tp = ts + (pl * 16 + pr) * TP + (orow + dr) * SP + 8 * seg;
#pragma unroll
                    for (int i = 0; i < 4; ++i) raw[pl][dr][i] = *reinterpret_cast<const u32x2 *>(tp + 4 * i);
                }
            float x12[2][8];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const float *k = pl ? cur.k2 : cur.k1;
#pragma unroll
                for (int j = 0; j < 8; ++j) x12[pl][j] = 0.f;
#pragma unroll
                for (int dr = 0; dr < 3; ++dr) {
                    float v[10], dump;
                    unpack2<T>(raw[pl][dr][0].y, dump, v[0]);
                    unpack2<T>(raw[pl][dr][1].x, v[1], v[2]); unpack2<T>(raw[pl][dr][1].y, v[3], v[4]);
                    unpack2<T>(raw[pl][dr][2].x, v[5], v[6]); unpack2<T>(raw[pl][dr][2].y, v[7], v[8]);
                    unpack2<T>(raw[pl][dr][3].x, v[9], dump);
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
                u32x4{pk2<T>(o[0], o[1]), pk2<T>(o[2], o[3]), pk2<T>(o[4], o[5]), pk2<T>(o[6], o[7])};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 3. out[D rows][the wave's 32 pixels] += W_out[:, chunk] x gate
        {
            const T *bp = gs + tr2;
            const s16x4e lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4e *)(bp));
            const s16x4e hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4e *)(bp + 4 * GP));
            const s16x8 bf = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
                const s16x8 af = (D % 32 == 0 || mt * 32 + col < D) ? __builtin_bit_cast(s16x8, cur.a2[mt]) : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
                acc2[mt] = Mfma<T>::run(af, bf, acc2[mt]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the next chunk's step 1 / 2 overwrite ts / gs
        if (more) issue_rest(c0 + 16, cur);   // (a second register set for them, requested a chunk ahead: no gain, A-B in DESIGN.md)
    }
    __syncthreads();   // everyone is done with xs and the t tiles before they become the output staging tile
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
                    u32x4{pk2<T>(v0.x + r8[0], v0.y + r8[1]), pk2<T>(v0.z + r8[2], v0.w + r8[3]),
                          pk2<T>(v1.x + r8[4], v1.y + r8[5]), pk2<T>(v1.z + r8[6], v1.w + r8[7])};
            }
        }
    }
}

// The weights as the kernel above reads them, from the module's fp32 parameters, ONE launch (so that a captured inference graph can
// redo it on every replay and never serves stale copies): w_in (2 HP, D) and w_out (D, HP) rounded to the I/O type, w_dw (2 HP, 9)
// float, both halves' padding rows and w_out's padding columns zero.
template <typename T>
__global__ void __launch_bounds__(256)
oss_effn_round_weights_kernel(const float *__restrict__ pin /*(2 h, D)*/, const float *__restrict__ pdw /*(2 h, 9)*/,
                              const float *__restrict__ pout /*(D, h)*/, T *__restrict__ w_in, float *__restrict__ w_dw,
                              T *__restrict__ w_out, int D, int h, int HP) {
    const int n1 = 2 * HP * D, n2 = 2 * HP * 9, n3 = D * HP;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n1) {
        const int r = i / D, c = i - r * D, half = r / HP, rr = r - half * HP;
        w_in[i] = from_f32<T>(rr < h ? pin[(size_t)(half * h + rr) * D + c] : 0.f);
    } else if (i < n1 + n2) {
        const int k = i - n1, r = k / 9, t = k - r * 9, half = r / HP, rr = r - half * HP;
        w_dw[k] = rr < h ? pdw[(size_t)(half * h + rr) * 9 + t] : 0.f;
    } else if (i < n1 + n2 + n3) {
        const int k = i - n1 - n2, r = k / HP, c = k - r * HP;
        w_out[k] = from_f32<T>(c < h ? pout[(size_t)r * h + c] : 0.f);
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
int effn_round_weights(oss_dtype io, const float *pin, const float *pdw, const float *pout, void *w_in, float *w_dw, void *w_out, int D,
                       int hidden, hipStream_t s) {
    if (io != OSS_F16 && io != OSS_BF16) return OSS_ERR_SHAPE;
    if (D < 1 || hidden < 1 || hidden > 16384 || D > 4096) return OSS_ERR_SHAPE;
    const int HP = (hidden + 15) / 16 * 16;
    const long total = 2L * HP * D + 2L * HP * 9 + (long)D * HP;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (io == OSS_F16)
        hipLaunchKernelGGL(oss_effn_round_weights_kernel<f16_t>, grid, dim3(256), 0, s, pin, pdw, pout, reinterpret_cast<f16_t *>(w_in), w_dw,
                           reinterpret_cast<f16_t *>(w_out), D, hidden, HP);
    else
        hipLaunchKernelGGL(oss_effn_round_weights_kernel<bf16_t>, grid, dim3(256), 0, s, pin, pdw, pout, reinterpret_cast<bf16_t *>(w_in), w_dw,
                           reinterpret_cast<bf16_t *>(w_out), D, hidden, HP);
    return (int)hipGetLastError();
}

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
