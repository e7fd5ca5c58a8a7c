// oss_scan_bwd_pair.h -- the selective-scan backward walking TWO states per pass in packed fp32
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two fp32 lanes-worth per VALU issue slot).
//
// Same algorithm, row ownership, workspace and finishing kernel as oss_scan_bwd_kernel (oss_scan_bwd.hip; reference:
// cus/selective_scan_bwd_kernel.cuh:66-273).  That kernel measured VALU-issue bound (~270 vector instructions per
// (wave, state, 512-step chunk), 69 % of the issue peak at 252 us for u:(8,384,4096)).  The recurrences of two states
// n, n+1 of the same row are independent and share delta, u, dout -- so they ride in the two halves of 64-bit
// register pairs and every multiply / fma of the state passes, and every add of the cross-row dB/dC reduction,
// handles both.  For that the LDS images interleave the two states:
//   * B / C tiles  [pair][item/2][lane][4] = {n:t, n+1:t, n:t+1, n+1:t+1}: one ds_read_b128 = two steps of both states;
//   * dB / dC slabs [row][B|C][lane][I*2]  (16-byte chunks XOR-swizzled by lane/4 so that the 64-byte lane stride of the
//     writes does not alias banks), summed over the rows with one ds_read_b64 + v_pk_add_f32 per row.
// An odd state count gets a padding state with A = B = C = 0 (all its terms are exact zeros, nothing is stored).
// Included by oss_scan_bwd.hip only (needs BwdWs).
#pragma once
#include "oss_device.h"

namespace oss {

typedef f32x2 f2;
__device__ __forceinline__ f2 splat2(float x) { return f2{x, x}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 exp2_2(f2 x) { return f2{exp2_hw(x.x), exp2_hw(x.y)}; }

// element (pair q, lane p, item i, state s) of a pair tile -> q*2*TC + (i/2)*256 + p*4 + (i%2)*2 + s   (TC = 64 I)
template <int I>
__device__ __forceinline__ int pair_off(int q, int p, int i) {
    return q * (128 * I) + (i >> 1) * 256 + p * 4 + (i & 1) * 2;
}

// four consecutive memory elements m0 .. m0+3 of a row of length L as fp32 (0 outside the row)
template <typename T>
__device__ __forceinline__ void load4_row(const T *row, int m0, int L, bool inside, float (&v)[4]) {
    const T *p = row + m0;
    constexpr uintptr_t amask = (sizeof(T) == 4) ? 15u : 7u;
    if (inside && (reinterpret_cast<uintptr_t>(p) & amask) == 0) {
        if constexpr (sizeof(T) == 4) {
            const f32x4 q = *reinterpret_cast<const f32x4 *>(p);
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
        } else {
            const u32x2 q = *reinterpret_cast<const u32x2 *>(p);
            unpack2<T>(q.x, v[0], v[1]);
            unpack2<T>(q.y, v[2], v[3]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + j;
            v[j] = (m >= 0 && m < L) ? to_f32(row[m]) : 0.f;
        }
    }
}

// Stage states n0 .. n0+nb-1 (rows of gB / gC, already offset to n0) x TC scan positions from t0 as pair tiles.
// `rev`: scan position s reads memory L-1-s.  A missing partner state (nb odd) is staged as zeros.
template <typename T, int I, int NT>
__device__ __forceinline__ void stage_bc_pairs(float *sB, float *sC, const T *gB, const T *gC, int64_t strideB,
                                               int64_t strideC, int nb, int t0, int L, bool rev, int tid) {
    constexpr int TC = 64 * I;
    constexpr int Q = TC / 4;  // 4-position groups per state row
    const bool inside = (t0 + TC <= L);
    const int npairs = (nb + 1) >> 1;
    for (int idx = tid; idx < npairs * Q; idx += NT) {
        const int q = idx / Q, k = idx - q * Q;
        const int s0 = t0 + 4 * k;                 // first scan position of the group
        const int m0 = rev ? (L - 4 - s0) : s0;    // lowest memory index of the group
        float b[2][4], c[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int n = 2 * q + s;
            if (n < nb) {
                load4_row<T>(gB + n * strideB, m0, L, inside, b[s]);
                load4_row<T>(gC + n * strideC, m0, L, inside, c[s]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { b[s][j] = 0.f; c[s][j] = 0.f; }
            }
        }
        if (rev) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float t;
                t = b[s][0]; b[s][0] = b[s][3]; b[s][3] = t;
                t = b[s][1]; b[s][1] = b[s][2]; b[s][2] = t;
                t = c[s][0]; c[s][0] = c[s][3]; c[s][3] = t;
                t = c[s][1]; c[s][1] = c[s][2]; c[s][2] = t;
            }
        }
        const int pos = (4 * k) / I, i0 = (4 * k) % I;
        const int off = pair_off<I>(q, pos, i0);
        *reinterpret_cast<f32x4 *>(sB + off) = f32x4{b[0][0], b[1][0], b[0][1], b[1][1]};
        *reinterpret_cast<f32x4 *>(sB + off + 256) = f32x4{b[0][2], b[1][2], b[0][3], b[1][3]};
        *reinterpret_cast<f32x4 *>(sC + off) = f32x4{c[0][0], c[1][0], c[0][1], c[1][1]};
        *reinterpret_cast<f32x4 *>(sC + off + 256) = f32x4{c[0][2], c[1][2], c[0][3], c[1][3]};
    }
}

// The 64-lane inclusive scan of segment_scan<64> for two independent (P, h) pairs, interleaved: each pair's
// instructions sit in the other's DPP wait states, so no s_nop between the steps.
#define OSS_DPP_STEP2(CTRL)                                  \
    "v_fmac_f32_dpp %0, %0, %1 " CTRL " bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %2, %2, %3 " CTRL " bank_mask:0xf\n\t" \
    "v_mul_f32_dpp %1, %1, %1 " CTRL " bank_mask:0xf\n\t"  \
    "v_mul_f32_dpp %3, %3, %3 " CTRL " bank_mask:0xf\n\t"
__device__ __forceinline__ void segment_scan2_64(f2 &P, f2 &h) {
    float hx = h.x, hy = h.y, px = P.x, py = P.y;
    asm volatile("s_nop 1\n\t" OSS_DPP_STEP2("row_shr:1 row_mask:0xf") OSS_DPP_STEP2("row_shr:2 row_mask:0xf")
                 OSS_DPP_STEP2("row_shr:4 row_mask:0xf") OSS_DPP_STEP2("row_shr:8 row_mask:0xf")
                 OSS_DPP_STEP2("row_bcast:15 row_mask:0xa") OSS_DPP_STEP2("row_bcast:31 row_mask:0xc")
                 : "+v"(hx), "+v"(px), "+v"(hy), "+v"(py));
    h = f2{hx, hy};
    P = f2{px, py};
}
#undef OSS_DPP_STEP2
__device__ __forceinline__ f2 mirror2_64(f2 v, int lane) {
    return f2{segment_mirror<64>(v.x, lane), segment_mirror<64>(v.y, lane)};
}
__device__ __forceinline__ f2 shift_prev2(f2 v, f2 fill, bool seg_first) {
    return f2{shift_from_prev_lane(v.x, fill.x, seg_first), shift_from_prev_lane(v.y, fill.y, seg_first)};
}
__device__ __forceinline__ f2 shift_next2(f2 v, f2 fill, bool seg_last) {
    return f2{shift_from_next_lane(v.x, fill.x, seg_last), shift_from_next_lane(v.y, fill.y, seg_last)};
}

// one wave = one row (64 lanes x I items = a chunk of TC steps); WAVES rows per workgroup; NBB states staged at once
template <typename T, int I, int WAVES, int NBB, int MINW>
__global__ void __launch_bounds__(WAVES * 64, MINW)
oss_scan_bwd_pair_kernel(const oss_scan_bwd_params p, const BwdWs ws) {
    constexpr int ROWS = WAVES;
    constexpr int TC = 64 * I;
    constexpr int NT = WAVES * 64;
    constexpr int CH = I / 2;           // 16-byte chunks of a lane's slab block (two items x two states each)
    static_assert(TC % kScanChunk == 0, "");
    static_assert(I % 4 == 0 && (CH & (CH - 1)) == 0 && NBB % 2 == 0, "");

    const oss_scan_fwd_params &f = p.f;
    const int L = f.seqlen, N = f.dstate, G = f.n_groups;
    const int Np = (N + 1) & ~1;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sB = smem;                          // [NBB/2] pair tiles of TC*2
    float *sC = sB + NBB * TC;
    float *slab = sC + NBB * TC;               // [ROWS][2][TC*2]  dB / dC terms of the current pair, per row
    float *sA2 = slab + ROWS * 4 * TC;         // [ROWS][Np]  A * log2(e)
    float *sdhc = sA2 + ROWS * Np;             // [ROWS][Np]  dh at the first step of the later chunk
    float *sdA = sdhc + ROWS * Np;             // [ROWS][Np]  dA partial of the row
    float *sdln = sdA + ROWS * Np;             // [ROWS]      delta of the first step of the later chunk

    const int tid = threadIdx.x, lane = tid & 63, wrow = tid >> 6;
    const int pos = lane;
    const bool seg_first = (pos == 0), seg_last = (pos == 63);

    const int rows_per_group = f.dim / G;
    const int tiles_per_group = ws.tiles;
    int bid = blockIdx.x;
    const int tile = bid % tiles_per_group; bid /= tiles_per_group;
    const int g = bid % G;
    const int b = bid / G;
    const int row_in_group = tile * ROWS + wrow;
    const bool row_valid = row_in_group < rows_per_group;
    const int d = g * rows_per_group + (row_valid ? row_in_group : 0);
    const bool rev = g >= f.rev_group_start;
    const int d_u = f.u_row_mod > 0 ? d % f.u_row_mod : d;

    const T *u_row = reinterpret_cast<const T *>(f.u) + b * f.u_batch_stride + d_u * f.u_d_stride;
    const T *dt_row = reinterpret_cast<const T *>(f.delta) + b * f.delta_batch_stride + d * f.delta_d_stride;
    const int d_g = p.dout_row_mod > 0 ? d % p.dout_row_mod : d;
    const T *g_row = reinterpret_cast<const T *>(p.dout) + b * p.dout_batch_stride + d_g * p.dout_d_stride;
    T *du_row = reinterpret_cast<T *>(p.du) + b * p.du_batch_stride + d * p.du_d_stride;
    T *dd_row = reinterpret_cast<T *>(p.ddelta) + b * p.ddelta_batch_stride + d * p.ddelta_d_stride;
    const T *gB = reinterpret_cast<const T *>(f.B) + b * f.B_batch_stride + g * f.B_group_stride;
    const T *gC = reinterpret_cast<const T *>(f.C) + b * f.C_batch_stride + g * f.C_group_stride;
    const float Dd = f.D ? f.D[d] : 0.f;
    const float bias = f.delta_bias ? f.delta_bias[d] : 0.f;
    const int n_xchunks = (L + kScanChunk - 1) / kScanChunk;
    const float *x_row = f.x ? f.x + ((size_t)b * f.dim + d) * n_xchunks * 2 * N : nullptr;
    float *ws_bc = ws.bc + ((size_t)(b * G + g) * tiles_per_group + tile) * 2 * N * L;

    for (int idx = tid; idx < ROWS * Np; idx += NT) {
        const int r = idx / Np, n = idx - r * Np;
        const int rg = tile * ROWS + r;
        const int dd = g * rows_per_group + (rg < rows_per_group ? rg : 0);
        float a2 = 0.f;   // the padding state of an odd N: a = exp2(0) = 1, B = C = 0
        if (n < N) {
            const float av = f.A[dd * f.A_d_stride + n];
            a2 = (f.a_log_form ? -__expf(av) : av) * kLog2e;
        }
        sA2[idx] = a2;
        sdhc[idx] = 0.f;
        sdA[idx] = 0.f;
    }
    if (tid < ROWS) sdln[tid] = 0.f;

    float dD_acc = 0.f, db_acc = 0.f;
    const int n_chunks = (L + TC - 1) / TC;
    for (int c = n_chunks - 1; c >= 0; --c) {
        const int t0 = c * TC;
        const int tl = t0 + pos * I;
        const int valid = max(0, min(I, L - tl));
        // (u and the softplus derivative are needed again only by the per-element outputs below: they are re-read /
        // re-derived there instead of living in 2 I registers across the state loop -- the kernel sits at the VGPR limit)
        float dl[I], gg[I], w[I];
        f2 Q[I], dd[I];
        load_items_dir<I>(dt_row, tl, valid, L, rev, dl);
        load_items_dir<I>(g_row, tl, valid, L, rev, gg);
        float S = 0.f;
        {
            float uu[I];
            load_items_dir<I>(u_row, tl, valid, L, rev, uu);
#pragma unroll
            for (int i = 0; i < I; ++i) {
                const float raw = dl[i] + bias;
                float e;
                const float x = f.delta_softplus ? softplus_thr(raw, e) : raw;
                dl[i] = (i < valid) ? x : 0.f;
                if (!row_valid) gg[i] = 0.f;   // a row slot past the end of the group must not contribute to dB/dC
                w[i] = row_valid ? dl[i] * uu[i] : 0.f;
                Q[i] = f2{0.f, 0.f};
                dd[i] = f2{0.f, 0.f};
                S += dl[i];
            }
        }
        const int xi = t0 / kScanChunk - 1;  // saved forward state entering this chunk (bwd_kernel.cuh:184)
        __syncthreads();  // sdln/sdhc written by the previous iteration (or the init) are visible
        const float dln_c = sdln[wrow];  // delta of step t0+TC (first step of the later chunk), 0 past the end
        const float dln_lane = shift_from_next_lane(dl[0], dln_c, seg_last);
        const float Sshift = S - dl[0] + dln_lane;  // sum of delta over steps tl+1 .. tl+I

        for (int n0 = 0; n0 < N; n0 += NBB) {
            const int nb = min(NBB, N - n0);
            __syncthreads();
            stage_bc_pairs<T, I, NT>(sB, sC, gB + (int64_t)n0 * f.B_dstate_stride, gC + (int64_t)n0 * f.C_dstate_stride,
                                     f.B_dstate_stride, f.C_dstate_stride, nb, t0, L, rev, tid);
            __syncthreads();
            for (int q = 0; 2 * q < nb; ++q) {
                const int n = n0 + 2 * q;          // states n and n+1 (n+1 == N: the padding state)
                const f2 A2 = *reinterpret_cast<const f2 *>(sA2 + wrow * Np + n);
                f2 hc = f2{0.f, 0.f};
                if (xi >= 0) {
                    hc.x = x_row[(size_t)xi * 2 * N + 2 * n + 1];
                    if (n + 1 < N) hc.y = x_row[(size_t)xi * 2 * N + 2 * n + 3];
                }
                const f2 dhc = *reinterpret_cast<const f2 *>(sdhc + wrow * Np + n);
                const float *tb = sB + pair_off<I>(q, pos, 0);
                const float *tc = sC + pair_off<I>(q, pos, 0);
                f2 a[I], hh[I];
                // ---- forward recompute: local recurrence (hh holds b_t w_t until the state pass)
                f2 h = f2{0.f, 0.f};
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(tb + k * 256);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int i = 2 * k + j;
                        a[i] = exp2_2(splat2(dl[i]) * A2);
                        hh[i] = f2{b4[2 * j], b4[2 * j + 1]} * splat2(w[i]);
                        h = (i == 0) ? hh[0] : fma2(a[i], h, hh[i]);
                    }
                }
                f2 P = exp2_2(splat2(S) * A2);
                segment_scan2_64(P, h);
                const f2 hfull = fma2(P, hc, h);
                const f2 hin = shift_prev2(hfull, hc, seg_first);
                // ---- forward states h_t in place
                {
                    f2 hp = hin;
#pragma unroll
                    for (int i = 0; i < I; ++i) {
                        hp = fma2(a[i], hp, hh[i]);
                        hh[i] = hp;
                    }
                }
                // ---- reverse recurrence: element (a_{t+1}, C_t g_t)   (bwd_kernel.cuh:170-193)
                const f2 a_edge = exp2_2(splat2(dln_c) * A2);
                const f2 a_nl = shift_next2(a[0], a_edge, seg_last);
                f2 dloc = f2{0.f, 0.f};
#pragma unroll
                for (int k = CH - 1; k >= 0; --k) {
                    const f32x4 c4 = *reinterpret_cast<const f32x4 *>(tc + k * 256);
#pragma unroll
                    for (int j = 1; j >= 0; --j) {
                        const int i = 2 * k + j;
                        const f2 an = (i == I - 1) ? a_nl : a[(i + 1) % I];
                        const f2 cg = f2{c4[2 * j], c4[2 * j + 1]} * splat2(gg[i]);
                        dloc = (i == I - 1) ? cg : fma2(an, dloc, cg);
                    }
                }
                f2 Pm = mirror2_64(exp2_2(splat2(Sshift) * A2), lane);
                f2 dm = mirror2_64(dloc, lane);
                segment_scan2_64(Pm, dm);
                const f2 dfull_m = fma2(Pm, dhc, dm);                    // dh at the first step of the mirrored lane
                const f2 dex_m = shift_prev2(dfull_m, dhc, seg_first);
                f2 dh = mirror2_64(dex_m, lane);                         // dh entering this lane from the right
                if (seg_last) *reinterpret_cast<f2 *>(sdhc + wrow * Np + n) = dfull_m;   // mirrored-last = first lane in time
                // ---- reverse pass with gradients (bwd_kernel.cuh:196-206); p_t = a_t h_{t-1}.  The dB / dC terms of the
                // pair go straight into this row's slab (two items x two states = one 16-byte chunk each), so they
                // never occupy registers beyond one chunk; hence the barrier up front
                __syncthreads();  // the previous pair's slab sums have been read
                float *sb = slab + (size_t)(wrow * 2) * (2 * TC) + pos * (2 * I);
                float *sc = sb + 2 * TC;
                const int swz = (pos >> 2) & (CH - 1);
                f2 dA_acc = f2{0.f, 0.f};
#pragma unroll
                for (int k = CH - 1; k >= 0; --k) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(tb + k * 256);
                    const f32x4 c4 = *reinterpret_cast<const f32x4 *>(tc + k * 256);
                    f2 vB[2], vC[2];
#pragma unroll
                    for (int j = 1; j >= 0; --j) {
                        const int i = 2 * k + j;
                        const f2 an = (i == I - 1) ? a_nl : a[(i + 1) % I];
                        dh = fma2(an, dh, f2{c4[2 * j], c4[2 * j + 1]} * splat2(gg[i]));
                        Q[i] = fma2(dh, f2{b4[2 * j], b4[2 * j + 1]}, Q[i]);
                        const f2 hprev = (i == 0) ? hin : hh[(i + I - 1) % I];
                        const f2 r = dh * (a[i] * hprev);
                        dd[i] = fma2(A2, r, dd[i]);
                        dA_acc = fma2(splat2(dl[i]), r, dA_acc);
                        // rows past the end of the group carry u = dout = 0, so they contribute exact zeros
                        vB[j] = dh * splat2(w[i]);
                        vC[j] = splat2(gg[i]) * hh[i];
                    }
                    const int pk = (k ^ swz) * 4;
                    *reinterpret_cast<f32x4 *>(sb + pk) = f32x4{vB[0].x, vB[0].y, vB[1].x, vB[1].y};
                    *reinterpret_cast<f32x4 *>(sc + pk) = f32x4{vC[0].x, vC[0].y, vC[1].x, vC[1].y};
                }
                const f2 dA_sum = f2{segment_sum_to_last<64>(dA_acc.x), segment_sum_to_last<64>(dA_acc.y)};
                if (seg_last) {
                    f2 *pa = reinterpret_cast<f2 *>(sdA + wrow * Np + n);
                    *pa = *pa + dA_sum;
                }
                // ---- cross-row reduction of dB/dC of the pair through the slabs
                __syncthreads();
                for (int j = tid; j < 2 * TC; j += NT) {   // one (B|C, time step) per thread, both states
                    const int bc = j >= TC ? 1 : 0;
                    const int ts = j - bc * TC;
                    const int pp = ts / I, i = ts - pp * I;
                    const int off = bc * (2 * TC) + pp * (2 * I) + (((i >> 1) ^ ((pp >> 2) & (CH - 1))) << 2) + (i & 1) * 2;
                    f2 acc = f2{0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < ROWS; ++r)   // fixed order: deterministic
                        acc = acc + *reinterpret_cast<const f2 *>(slab + (size_t)r * (4 * TC) + off);
                    const int t = t0 + ts;  // scan position; mirrored groups store at L-1-t
                    if (t < L) {
                        const int tm = rev ? (L - 1 - t) : t;
                        float *o = ws_bc + (size_t)(bc * N + n) * L + tm;
                        o[0] = acc.x;
                        if (n + 1 < N) o[L] = acc.y;
                    }
                }
            }
        }
        // ---- per-element outputs (bwd_kernel.cuh:151,200-203,228-245)
        float du[I], dv[I];
        {
            float uu[I], raw[I];
            load_items_dir<I>(u_row, tl, valid, L, rev, uu);
            load_items_dir<I>(dt_row, tl, valid, L, rev, raw);
#pragma unroll
            for (int i = 0; i < I; ++i) {
                float sg = 1.f;
                if (f.delta_softplus) {   // d softplus = sigmoid(raw) for raw <= 20, 1 above (bwd_kernel.cuh:228-241)
                    const float r = raw[i] + bias;
                    const float e = exp2_hw(r * kLog2e);
                    sg = (r <= 20.f) ? e * __builtin_amdgcn_rcpf(1.f + e) : 1.f;
                }
                if (i >= valid) sg = 0.f;
                const float ui = row_valid ? uu[i] : 0.f;
                const float Qi = Q[i].x + Q[i].y;
                const float ddi = dd[i].x + dd[i].y;
                du[i] = __builtin_fmaf(Qi, dl[i], Dd * gg[i]);
                const float ddel = __builtin_fmaf(Qi, ui, ddi * kLn2);
                dv[i] = ddel * sg;
                dD_acc = __builtin_fmaf(gg[i], ui, dD_acc);
                db_acc += dv[i];
            }
        }
        if (row_valid) {
            store_items_dir<I>(du_row, tl, valid, L, rev, du);
            store_items_dir<I>(dd_row, tl, valid, L, rev, dv);
        }
        __syncthreads();  // every lane has read sdln for this chunk
        if (seg_first) sdln[wrow] = dl[0];
    }
    // ---- per-row partials over the sequence
    const float dD_sum = segment_sum_to_last<64>(dD_acc);
    const float db_sum = segment_sum_to_last<64>(db_acc);
    if (seg_last && row_valid) {
        if (ws.dD) ws.dD[(size_t)b * f.dim + d] = dD_sum;
        if (ws.db) ws.db[(size_t)b * f.dim + d] = db_sum;
        for (int n = 0; n < N; ++n) ws.dA[((size_t)b * f.dim + d) * N + n] = sdA[wrow * Np + n];
    }
}

}  // namespace oss
