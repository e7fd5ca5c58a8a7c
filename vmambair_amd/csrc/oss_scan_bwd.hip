// oss_scan_bwd.hip -- selective-scan backward for gfx950 (replaces the reference's
// selective_scan_bwd_kernel, cus/selective_scan_bwd_kernel.cuh:66-273, and the host-side
// zero-fills / casts around it, cus/selective_scan.cpp:319-327,347).  Arithmetic per SURVEY.md
// Appendix A.
//
// Same row ownership as the forward (LPR lanes of one wave own a row for the whole sequence, I
// items per lane, B/C of the (batch, group) staged once per chunk in LDS), chunks walked from the
// last to the first:
//   * forward states of the chunk are recomputed from the state saved in x every kScanChunk steps
//     (no carry needed in this direction);
//   * the reverse recurrence dh_t = C_t g_t + a_{t+1} dh_{t+1} is the same monoid scanned over the
//     mirrored lane order (ds_bpermute mirror + the forward DPP scan), its carry and the delta of
//     the chunk's first step (a_{t+1} across the chunk edge is exp2(A * delta_first), so no
//     per-state edge value is kept) live in LDS;
//   * dB/dC -- sums over the rows of a group -- are reduced across the workgroup's rows through LDS
//     slabs: every wave stores its row's I values per lane for the current state, one barrier,
//     then wave w sums time slice w over all rows in a fixed order and writes the workgroup's
//     partial with plain coalesced stores; a finishing kernel adds the partials of a group in tile
//     order and casts to the I/O type.  (LDS float atomics were measured at ~200 clocks per wave
//     instruction on gfx950 -- profiles/r01_ubench.txt -- and made the first version 20x slower;
//     device-scope global float atomics cannot stay in the per-XCD L2s.)  No atomics anywhere, no
//     zero-filled outputs, bit-reproducible.
//   * dA, dD, ddelta_bias are per-(batch,row) partials reduced over batch by the finishing kernel.
#include "oss_device.h"
#include "oss_host.h"

namespace oss {

// workspace layout (floats):
//   [0, nBC)                 dB/dC (+ d dt-factor) partials  [batch][group][tile][2 N + RP][L]
//   [nBC, nBC + batch*dim*N) dA partials     [batch][dim][N]
//   then dD partials [batch][dim], then ddelta_bias partials [batch][dim], then dt-weight partials [batch][dim][kMaxDtRank]
// RP = rows of the gradient of the dt factor z (fused-delta form: dt_rank), else 0.
struct BwdWs {
    float *bc, *dA, *dD, *db, *dW;
    int tiles, rp;
};
__host__ __device__ inline size_t ws_bc_floats(int batch, int G, int tiles, int N, int L, int RP = 0) {
    return (size_t)batch * G * tiles * (2 * N + RP) * L;
}

// SPS: states walked together between two barriers (2: two independent dependency chains per wave for the
// scheduler to interleave and half the barriers, at twice the slab LDS and ~2x the registers -- for grids
// that cannot fill the chip with waves anyway)
// MINW: waves per SIMD the register allocation must leave room for (4: <= 128 VGPRs, 2: <= 256)
template <typename T, int LPR, int I, int WAVES, int NBB, int SPS, int MINW>
__global__ void __launch_bounds__(WAVES * 64, MINW)
oss_scan_bwd_kernel(const oss_scan_bwd_params p, const BwdWs ws) {
    constexpr int RPW = 64 / LPR;
    constexpr int ROWS = WAVES * RPW;
    constexpr int TC = LPR * I;
    constexpr int NT = WAVES * 64;
    static_assert(TC % kScanChunk == 0, "");
    static_assert(LPR == 64, "slab reduction: one row per wave");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sB = smem;                    // [NBB][TC]  tile_off layout
    float *sC = sB + NBB * TC;           // [NBB][TC]
    float *slab = sC + NBB * TC;         // [SPS][ROWS][2][TC]  per-row dB / dC terms of the current state(s)
    float *sA2 = slab + SPS * ROWS * 2 * TC;           // [N][ROWS]
    float *sdhc = sA2 + (size_t)p.f.dstate * ROWS;     // [N][ROWS]  dh of the first step of the later chunk
    float *sdA = sdhc + (size_t)p.f.dstate * ROWS;     // [N][ROWS]  dA partial of the row
    float *sdln = sdA + (size_t)p.f.dstate * ROWS;     // [ROWS]     delta of the first step of the later chunk

    const oss_scan_fwd_params &f = p.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pos = lane & (LPR - 1);
    const int wrow = wave * RPW + lane / LPR;
    const bool seg_first = (pos == 0), seg_last = (pos == LPR - 1);

    const int L = f.seqlen, N = f.dstate, G = f.n_groups;
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

    for (int idx = tid; idx < N * ROWS; idx += NT) {
        const int n = idx / ROWS, r = idx - n * ROWS;
        const int rg = tile * ROWS + r;
        const int dd = g * rows_per_group + (rg < rows_per_group ? rg : 0);
        const float av = f.A[dd * f.A_d_stride + n];
        sA2[idx] = (f.a_log_form ? -__expf(av) : av) * kLog2e;
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
        float uu[I], dl[I], gg[I], sig[I], w[I], Q[I], dd[I];
        load_items_dir<I>(u_row, tl, valid, L, rev, uu);
        load_items_dir<I>(dt_row, tl, valid, L, rev, dl);
        load_items_dir<I>(g_row, tl, valid, L, rev, gg);
        if (!row_valid) {  // a row slot past the end of the group must not contribute to dB/dC
#pragma unroll
            for (int i = 0; i < I; ++i) { uu[i] = 0.f; gg[i] = 0.f; }
        }
        float S = 0.f;
#pragma unroll
        for (int i = 0; i < I; ++i) {
            const float raw = dl[i] + bias;
            float x = raw, s = 1.f;
            if (f.delta_softplus) {
                float e;
                x = softplus_thr(raw, e);
                // d softplus = sigmoid(raw) for raw <= 20, 1 above (bwd_kernel.cuh:228-241)
                s = (raw <= 20.f) ? e * __builtin_amdgcn_rcpf(1.f + e) : 1.f;
            }
            const bool ok = i < valid;
            dl[i] = ok ? x : 0.f;
            sig[i] = ok ? s : 0.f;
            w[i] = dl[i] * uu[i];
            Q[i] = 0.f;
            dd[i] = 0.f;
            S += dl[i];
        }
        const int xi = t0 / kScanChunk - 1;  // saved forward state entering this chunk (bwd_kernel.cuh:184)
        __syncthreads();  // sdln/sdhc written by the previous iteration (or the init) are visible
        const float dln_c = sdln[wrow];  // delta of step t0+TC (first step of the later chunk), 0 past the end
        const float dln_lane = shift_from_next_lane(dl[0], dln_c, seg_last);
        const float Sshift = S - dl[0] + dln_lane;  // sum of delta over steps tl+1 .. tl+I

        for (int n0 = 0; n0 < N; n0 += NBB) {
            const int nb = min(NBB, N - n0);
            __syncthreads();
            stage_bc_tiles<T, LPR, I, NT>(sB, sC, gB + (int64_t)n0 * f.B_dstate_stride,
                                          gC + (int64_t)n0 * f.C_dstate_stride, f.B_dstate_stride,
                                          f.C_dstate_stride, nb, t0, L, rev, tid);
            __syncthreads();
            // everything of one state up to its dB / dC terms (vB, vC); accumulates Q, dd, sdA, stores sdhc
            auto state_pass = [&](int nn, float (&vB)[I], float (&vC)[I]) {
                const int n = n0 + nn;
                const float A2 = sA2[n * ROWS + wrow];
                // issued here, consumed after the first pass and the scan: the latency is covered (a per-chunk
                // prefetch through LDS measured 10 % slower, profiles/r01_sweep_v3_bwd.txt)
                const float hc = (xi >= 0) ? x_row[(size_t)xi * 2 * N + 2 * n + 1] : 0.f;
                const float dhc = sdhc[n * ROWS + wrow];
                float a[I], hh[I];
                // ---- forward recompute: local recurrence (hh holds b_t until the state pass)
                float h = 0.f;
                const float *tb = sB + tile_off<LPR, I>(nn, pos, 0);
                const float *tc = sC + tile_off<LPR, I>(nn, pos, 0);
#pragma unroll
                for (int k = 0; k < I / 4; ++k) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(tb + k * (LPR * 4));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = 4 * k + j;
                        a[i] = exp2_hw(dl[i] * A2);
                        hh[i] = b4[j] * w[i];
                        h = (i == 0) ? hh[0] : __builtin_fmaf(a[i], h, hh[i]);
                    }
                }
                float P = exp2_hw(S * A2);
                segment_scan<LPR>(P, h);
                const float hfull = __builtin_fmaf(P, hc, h);
                const float hin = shift_from_prev_lane(hfull, hc, seg_first);
                // ---- forward states h_t in place
                {
                    float hp = hin;
#pragma unroll
                    for (int i = 0; i < I; ++i) {
                        hp = __builtin_fmaf(a[i], hp, hh[i]);
                        hh[i] = hp;
                    }
                }
                // ---- reverse recurrence: element (a_{t+1}, C_t g_t)   (bwd_kernel.cuh:170-193)
                const float a_edge = exp2_hw(dln_c * A2);
                const float a_nl = shift_from_next_lane(a[0], a_edge, seg_last);
                float dloc = 0.f;
#pragma unroll
                for (int k = I / 4 - 1; k >= 0; --k) {
                    const f32x4 c4 = *reinterpret_cast<const f32x4 *>(tc + k * (LPR * 4));
#pragma unroll
                    for (int j = 3; j >= 0; --j) {
                        const int i = 4 * k + j;
                        const float an = (i == I - 1) ? a_nl : a[(i + 1) % I];
                        const float cg = c4[j] * gg[i];
                        dloc = (i == I - 1) ? cg : __builtin_fmaf(an, dloc, cg);
                    }
                }
                float Pm = segment_mirror<LPR>(exp2_hw(Sshift * A2), lane);
                float dm = segment_mirror<LPR>(dloc, lane);
                segment_scan<LPR>(Pm, dm);
                const float dfull_m = __builtin_fmaf(Pm, dhc, dm);       // dh at the first step of mirrored lane
                const float dex_m = shift_from_prev_lane(dfull_m, dhc, seg_first);
                float dh = segment_mirror<LPR>(dex_m, lane);              // dh entering this lane from the right
                if (seg_last) sdhc[n * ROWS + wrow] = dfull_m;            // mirrored-last = first lane in time
                // ---- reverse pass with gradients (bwd_kernel.cuh:196-206); p_t = a_t h_{t-1}
                float dA_acc = 0.f;
#pragma unroll
                for (int k = I / 4 - 1; k >= 0; --k) {
#ifdef OSS_EXP_NO_REREAD  // (timing experiments only: tools/build_experiment.sh) sensitivity to the B/C tile reads
                    const f32x4 b4 = f32x4{dl[4 * k], dl[4 * k + 1], dl[4 * k + 2], dl[4 * k + 3]};
                    const f32x4 c4 = f32x4{w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]};
#else
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(tb + k * (LPR * 4));
                    const f32x4 c4 = *reinterpret_cast<const f32x4 *>(tc + k * (LPR * 4));
#endif
#pragma unroll
                    for (int j = 3; j >= 0; --j) {
                        const int i = 4 * k + j;
                        const float an = (i == I - 1) ? a_nl : a[(i + 1) % I];
                        dh = __builtin_fmaf(an, dh, c4[j] * gg[i]);
                        Q[i] = __builtin_fmaf(dh, b4[j], Q[i]);
                        const float hprev = (i == 0) ? hin : hh[(i + I - 1) % I];
                        const float r = dh * (a[i] * hprev);
                        dd[i] = __builtin_fmaf(A2, r, dd[i]);
                        dA_acc = __builtin_fmaf(dl[i], r, dA_acc);
                        // rows past the end of the group carry u = dout = 0, so they contribute exact zeros
                        vB[i] = dh * w[i];
                        vC[i] = gg[i] * hh[i];
                    }
                }
                const float dA_sum = segment_sum_to_last<LPR>(dA_acc);
                if (seg_last) sdA[n * ROWS + wrow] += dA_sum;
            };
            for (int nn = 0; nn < nb; nn += SPS) {
                float vB[SPS][I], vC[SPS][I];
                const bool two = SPS == 2 && nn + 1 < nb;
                state_pass(nn, vB[0], vC[0]);
                if constexpr (SPS == 2) {
                    if (two) state_pass(nn + 1, vB[1], vC[1]);
                }
#ifndef OSS_EXP_NO_REDUCE  // (timing experiments only: tools/build_experiment.sh)
                // ---- cross-row reduction of dB/dC for these states through the slabs
                __syncthreads();  // the previous states' slice sums have been read
#pragma unroll
                for (int q = 0; q < SPS; ++q) {
                    if (q == 0 || two) {
                        float *sb = slab + ((q * ROWS + wrow) * 2) * TC + pos * I;
                        float *sc = sb + TC;
#pragma unroll
                        for (int k = 0; k < I / 4; ++k) {
                            *reinterpret_cast<f32x4 *>(sb + 4 * k) = f32x4{vB[q][4 * k], vB[q][4 * k + 1], vB[q][4 * k + 2], vB[q][4 * k + 3]};
                            *reinterpret_cast<f32x4 *>(sc + 4 * k) = f32x4{vC[q][4 * k], vC[q][4 * k + 1], vC[q][4 * k + 2], vC[q][4 * k + 3]};
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < SPS; ++q) {
                    if (q == 0 || two) {
                        const int n = n0 + nn + q;
                        for (int ts = tid; ts < TC; ts += NT) {  // one time step per thread (TC / NT passes)
                            float accb = 0.f, accc = 0.f;
#pragma unroll
                            for (int r = 0; r < ROWS; ++r) {  // fixed order: deterministic
                                accb += slab[((q * ROWS + r) * 2) * TC + ts];
                                accc += slab[((q * ROWS + r) * 2 + 1) * TC + ts];
                            }
                            const int t = t0 + ts;  // scan position; mirrored groups store at L-1-t
                            if (t < L) {
                                const int tm = rev ? (L - 1 - t) : t;
                                ws_bc[(size_t)n * L + tm] = accb;
                                ws_bc[(size_t)(N + n) * L + tm] = accc;
                            }
                        }
                    }
                }
#else
                if (vB[0][0] + vC[0][I - 1] == 12345.678f) ws_bc[0] = vB[0][1];  // keep the values alive
#endif
            }
        }
        // ---- per-element outputs (bwd_kernel.cuh:151,200-203,228-245)
        float du[I], dv[I];
#pragma unroll
        for (int i = 0; i < I; ++i) {
            du[i] = __builtin_fmaf(Q[i], dl[i], Dd * gg[i]);
            const float ddel = __builtin_fmaf(Q[i], uu[i], dd[i] * kLn2);
            dv[i] = ddel * sig[i];
            dD_acc = __builtin_fmaf(gg[i], uu[i], dD_acc);
            db_acc += dv[i];
        }
        if (row_valid) {
            store_items_dir<I>(du_row, tl, valid, L, rev, du);
            store_items_dir<I>(dd_row, tl, valid, L, rev, dv);
        }
        __syncthreads();  // every lane has read sdln for this chunk
        if (seg_first) sdln[wrow] = dl[0];
    }
    // ---- per-row partials over the sequence
    const float dD_sum = segment_sum_to_last<LPR>(dD_acc);
    const float db_sum = segment_sum_to_last<LPR>(db_acc);
    if (seg_last && row_valid) {
        if (ws.dD) ws.dD[(size_t)b * f.dim + d] = dD_sum;
        if (ws.db) ws.db[(size_t)b * f.dim + d] = db_sum;
        for (int n = 0; n < N; ++n) ws.dA[((size_t)b * f.dim + d) * N + n] = sdA[n * ROWS + wrow];
    }
}

// sum of `count` partials `stride` floats apart, in index order, eight loads in flight (time-segmented launches leave
// batch x segments of them per output: one dependent load per iteration made this slab the finishing kernel's critical path)
__device__ __forceinline__ float sum_strided(const float *p, int count, size_t stride) {
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= count; k += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(size_t)(k + j) * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; k < count; ++k) s += p[(size_t)k * stride];
    return s;
}

// dA, dD, ddelta_bias: sum over batch (x time segments) in index order (device function: runs in the trailing blocks
// of the finishing launch)
__device__ __forceinline__ void finish_w(int i, const float *ws_dA, const float *ws_dD, const float *ws_db, float *dA,
                                         float *dD, float *db, int batch, int dim, int N, const float *A_log,
                                         int64_t A_d_stride, const float *ws_dW, float *dW, int R) {
    const int nA = dim * N;
    if (dW && i >= nA + dim) {   // dt-weight gradient of the fused-delta form: sum over batch in batch order
        const int j = i - nA - dim;
        if (j < dim * R) {
            const int d = j / R, r = j - d * R;
            dW[j] = sum_strided(ws_dW + (size_t)d * kMaxDtRank + r, batch, (size_t)dim * kMaxDtRank);
        }
        return;
    }
    if (i < nA) {
        float s = sum_strided(ws_dA + i, batch, (size_t)nA);
        if (A_log) s *= -__expf(A_log[(i / N) * A_d_stride + (i % N)]);  // d/dA_log of A = -exp(A_log)
        dA[i] = s;
    } else if (i < nA + dim) {
        const int d = i - nA;
        if (dD) dD[d] = sum_strided(ws_dD + d, batch, (size_t)dim);
        if (db) db[d] = sum_strided(ws_db + d, batch, (size_t)dim);
    }
}

// ONE finishing launch: blocks [0, nblk_bc) add the row-tile partials of dB / dC (and of the dt-factor gradient in the
// fused-delta form) in tile order and cast to the I/O type; the remaining blocks reduce the weight-gradient partials over batch.
struct FinishArgs {
    const float *ws_bc;
    void *dB, *dC, *dZ;
    int tiles, N, RP, R, G;
    size_t L, total;            // total = batch * G * (2 N + R) * L outputs
    unsigned nblk_bc;
    const float *ws_dA, *ws_dD, *ws_db, *ws_dW;
    float *dA, *dD, *db, *dW;
    int batch, dim;
    int wbatch;                 // weight-gradient partials per row: batch * time segments
    const float *A_log;
    int64_t A_d_stride;
    size_t out_group_stride;    // (batch, group) block stride of dB and of dC
    int64_t dz_batch_stride, dz_group_stride, dz_rank_stride;
    // (round 6) the dt-factor gradient in the same launch (oss_scan_bwd_params.finish_dt_weight): extra z-slabs of the grid run
    //   dZ[b, g, r, t] = sum over the rows d of group g of dtw[d, r] * ddelta[b, d, t]
    // -- what oss_dt_dgrad_kernel (oss_proj.hip) did as the NEXT launch of the block's backward -- next to the partial-row sums
    const void *ddelta;
    int64_t dd_batch_stride, dd_d_stride;
    const float *dtw;
    int dtR;                    // 0 = off
    unsigned dt_blocks;         // B * G * ceil(L / 256)
};
// grid = (ceil(L / (256 V)), 2 N + R output rows, batch * G + 1): no index arithmetic beyond one multiply-add per pointer; the
// last z-slab runs the weight-gradient sums (grid-stride).  V = 4: every lane adds four consecutive time steps with 16-byte
// loads of the partial rows (a wave's 4-byte loads move 256 bytes per instruction -- the V = 1 form spent its time issuing
// loads, 0.033 ms for 134 MB at u:(8,384,4096)); needs L % 4 == 0 and 8-byte aligned outputs, else V = 1.
// the dt-factor gradient of one (batch, group, 256-step slice): lane = 4 consecutive steps (8-byte loads for the 16-bit types), the
// block's four waves split the group's rows (wave w: rows w, w + 4, ...; twelve in flight per pass) and are added through LDS in
// wave order (32 KB).  A row's dt_rank weights ride in the lanes (lane r = w[r]; v_readlane), as in oss_dt_dgrad_kernel.
// (First version: one wave walked all rows of its 256 steps -- 96 dependent loads on 512 waves made these workgroups the long pole
// of the launch, 40 us against 24 + 12 for the two separate launches.)  Needs L % 4 == 0.
constexpr int kFinishDtSteps = 256;
template <typename T>
__device__ __forceinline__ void finish_dt_body(const FinishArgs &a, unsigned blk) {
    constexpr int DU = 12, RM = 8, NW = 4;
    __shared__ float red[NW * RM * kFinishDtSteps];
    const unsigned per = (unsigned)((a.L + kFinishDtSteps - 1) / kFinishDtSteps);
    const unsigned bg = blk / per, sl = blk - bg * per;
    const unsigned b = bg / (unsigned)a.G, g = bg - b * (unsigned)a.G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t t = (size_t)sl * kFinishDtSteps + (size_t)lane * 4;
    const bool ok = t < a.L;
    const size_t tc = ok ? t : 0;
    const int rows = a.dim / a.G, R = a.dtR;
    const T *src = reinterpret_cast<const T *>(a.ddelta) + (size_t)b * a.dd_batch_stride + (size_t)g * rows * a.dd_d_stride + tc;
    const float *wg = a.dtw + (size_t)g * rows * R;
    float acc[RM][4];
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[r][i] = 0.f;
    const int rl = min(lane, R - 1);
    for (int d0 = wave; d0 < rows; d0 += DU * NW) {
        float gq[DU][4], wl[DU];
#pragma unroll
        for (int u = 0; u < DU; ++u) {   // clamped rows, masked through a zero weight: the loads stay one group
            const int dd = d0 + u * NW, d = min(dd, rows - 1);
            const T *q = src + (size_t)d * a.dd_d_stride;
            if constexpr (sizeof(T) == 4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(q);
                gq[u][0] = v.x; gq[u][1] = v.y; gq[u][2] = v.z; gq[u][3] = v.w;
            } else {
                const u32x2 v = *reinterpret_cast<const u32x2 *>(q);
                unpack2<T>(v.x, gq[u][0], gq[u][1]);
                unpack2<T>(v.y, gq[u][2], gq[u][3]);
            }
            wl[u] = (lane < R && dd < rows) ? wg[(size_t)d * R + rl] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < DU; ++u)
#pragma unroll
            for (int r = 0; r < RM; ++r) {
                const float wv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wl[u]), r));
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[r][i] = __builtin_fmaf(wv, gq[u][i], acc[r][i]);
            }
    }
#pragma unroll
    for (int r = 0; r < RM; ++r)
        *reinterpret_cast<f32x4 *>(red + (wave * RM + r) * kFinishDtSteps + lane * 4) = f32x4{acc[r][0], acc[r][1], acc[r][2], acc[r][3]};
    __syncthreads();
    T *dst = reinterpret_cast<T *>(a.dZ) + (size_t)b * a.dz_batch_stride + (size_t)g * a.dz_group_stride + t;
    for (int r = wave; r < R; r += NW) {
        f32x4 sum = *reinterpret_cast<const f32x4 *>(red + r * kFinishDtSteps + lane * 4);
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(red + (w2 * RM + r) * kFinishDtSteps + lane * 4);
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        if (ok) {
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<f32x4 *>(dst + (size_t)r * a.dz_rank_stride) = sum;
            } else {
                *reinterpret_cast<u32x2 *>(dst + (size_t)r * a.dz_rank_stride) = u32x2{pack2<T>(sum.x, sum.y), pack2<T>(sum.z, sum.w)};
            }
        }
    }
}

// PB: the partial rows are bf16 (oss_scan_bwd_v2.h: kPartialsBf16 -- the round-2 kernels at bf16 I/O), same element layout
template <typename T, int V, bool PB = false>
__global__ void __launch_bounds__(256)
oss_scan_bwd_finish(const FinishArgs a) {
    const size_t n_bg = (size_t)a.batch * a.G;
    // z-slabs [0, dt_slabs): the dt-factor gradient, one 256-step slice per workgroup -- FIRST, so that these (longer) workgroups
    // start with the launch and the partial-row sums fill in around them; then batch * G slabs of partial-row sums; the last
    // slab reduces the weight-gradient partials
    const unsigned dt_slabs = a.dt_blocks ? (a.dt_blocks + gridDim.x * gridDim.y - 1) / (gridDim.x * gridDim.y) : 0;
    if (blockIdx.z < dt_slabs) {
        const unsigned blk = (unsigned)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (blk < a.dt_blocks) finish_dt_body<T>(a, blk);
        return;
    }
    const size_t bz = blockIdx.z - dt_slabs;
    if (bz >= n_bg) {
        const int total_w = a.dim * a.N + a.dim + a.dim * a.R;
        const int stride = (int)(gridDim.x * gridDim.y) * 256;
        for (int i = (int)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + (int)threadIdx.x; i < total_w; i += stride)
            finish_w(i, a.ws_dA, a.ws_dD, a.ws_db, a.dA, a.dD, a.db, a.wbatch, a.dim, a.N, a.A_log, a.A_d_stride, a.ws_dW, a.dW, a.R);
        return;
    }
    const size_t t = ((size_t)blockIdx.x * 256 + threadIdx.x) * V;
    if (t >= a.L) return;
    const size_t row = blockIdx.y, bg = bz;
    const size_t pt = (2 * (size_t)a.N + a.RP) * a.L;   // partial floats per (b, g, tile)
    using PT = typename std::conditional<PB, bf16_t, float>::type;
    const PT *base = reinterpret_cast<const PT *>(a.ws_bc) + bg * a.tiles * pt + row * a.L + t;
    T *dst;
    if (row < (size_t)a.N) dst = reinterpret_cast<T *>(a.dB) + bg * a.out_group_stride + row * a.L + t;
    else if (row < 2 * (size_t)a.N) dst = reinterpret_cast<T *>(a.dC) + bg * a.out_group_stride + (row - a.N) * a.L + t;
    else {
        const size_t b = bg / a.G, g = bg - b * a.G;
        dst = reinterpret_cast<T *>(a.dZ) + b * a.dz_batch_stride + g * a.dz_group_stride + (row - 2 * a.N) * a.dz_rank_stride + t;
    }
    if constexpr (V == 4) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < a.tiles; k0 += 8) {   // eight 16-byte loads in flight, added in tile order
            f32x4 v8[8];
            if constexpr (PB) {
                u32x2 r8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) r8[k] = *reinterpret_cast<const u32x2 *>(base + (size_t)min(k0 + k, a.tiles - 1) * pt);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float q0, q1, q2, q3;
                    unpack2<bf16_t>(r8[k].x, q0, q1);
                    unpack2<bf16_t>(r8[k].y, q2, q3);
                    v8[k] = f32x4{q0, q1, q2, q3};
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int kk = min(k0 + k, a.tiles - 1);   // clamped address, masked below: the loads stay one group
                    v8[k] = *reinterpret_cast<const f32x4 *>(base + (size_t)kk * pt);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k0 + k < a.tiles) { s.x += v8[k].x; s.y += v8[k].y; s.z += v8[k].z; s.w += v8[k].w; }
        }
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<f32x4 *>(dst) = s;
        } else {
            *reinterpret_cast<u32x2 *>(dst) = u32x2{pack2<T>(s.x, s.y), pack2<T>(s.z, s.w)};
        }
    } else {
        float s = 0.f;
        for (int k0 = 0; k0 < a.tiles; k0 += 8) {   // eight loads in flight, added in tile order (same sum as a rolled loop)
            float v8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v8[k] = (k0 + k < a.tiles) ? to_f32(base[(size_t)(k0 + k) * pt]) : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v8[k];
        }
        *dst = from_f32<T>(s);
    }
}

}  // namespace oss
#include "oss_scan_bwd_v2.h"
namespace oss {

static thread_local LaunchTimer *g_finish_timer = nullptr;   // set by scan_bwd_dispatch around the launchers

// workspace carving shared by the launchers; -> OSS_OK or OSS_ERR_WORKSPACE.  n_seg > 1 (time-segmented launch): one
// weight-gradient partial per (batch, segment, row) and, behind them, the reverse-carry pairs (*carry).
static int carve_ws(const oss_scan_bwd_params &p, int rows_per_wg, BwdWs &ws, float *&wdD, float *&wdb, int n_seg = 1,
                    float **carry = nullptr, int n_cseg = 0) {
    const oss_scan_fwd_params &f = p.f;
    const int rows_per_group = f.dim / f.n_groups;
    const int tiles = (rows_per_group + rows_per_wg - 1) / rows_per_wg;
    const int rp = f.dt_weight ? f.dt_rank : 0;
    const size_t n_bc = ws_bc_floats(f.batch, f.n_groups, tiles, f.dstate, f.seqlen, rp);
    const size_t wb = (size_t)f.batch * n_seg;
    const size_t n_w = wb * f.dim * (f.dstate + 2 + (rp ? kMaxDtRank : 0));
    const size_t n_carry = n_seg > 1 ? scan_carry_bytes(f.batch, f.dim, f.dstate, std::max(n_seg, n_cseg)) / sizeof(float) : 0;
    const size_t need = sizeof(float) * (n_bc + n_w + n_carry);
    if (!p.workspace || p.workspace_bytes < need) return OSS_ERR_WORKSPACE;
    ws.bc = reinterpret_cast<float *>(p.workspace);
    ws.dA = ws.bc + n_bc;
    ws.dD = ws.dA + wb * f.dim * f.dstate;
    ws.db = ws.dD + wb * f.dim;
    ws.dW = rp ? ws.db + wb * f.dim : nullptr;
    if (carry) *carry = ws.bc + n_bc + n_w;
    ws.tiles = tiles;
    ws.rp = rp;
    wdD = ws.dD; wdb = ws.db;
    if (!p.dD) ws.dD = nullptr;
    if (!p.ddelta_bias) ws.db = nullptr;
    return OSS_OK;
}

template <typename T, bool PB = false>
static int launch_finish(const oss_scan_bwd_params &p, const BwdWs &ws, float *wdD, float *wdb, hipStream_t stream, int n_seg = 1) {
    const oss_scan_fwd_params &f = p.f;
    FinishArgs a;
    a.wbatch = f.batch * n_seg;
    a.ws_bc = ws.bc; a.dB = p.dB; a.dC = p.dC; a.dZ = p.ddt;
    a.tiles = ws.tiles; a.N = f.dstate; a.RP = ws.rp; a.R = ws.rp ? f.dt_rank : 0; a.G = f.n_groups;
    a.L = (size_t)f.seqlen;
    a.total = (size_t)f.batch * f.n_groups * (2 * (size_t)f.dstate + a.R) * f.seqlen;
    a.nblk_bc = (unsigned)((a.total + 255) / 256);
    a.ws_dA = ws.dA; a.ws_dD = wdD; a.ws_db = wdb; a.ws_dW = ws.dW;
    a.dA = p.dA; a.dD = p.dD; a.db = p.ddelta_bias; a.dW = ws.rp ? p.ddt_weight : nullptr;
    a.batch = f.batch; a.dim = f.dim;
    a.A_log = f.a_log_form ? f.A : nullptr; a.A_d_stride = f.A_d_stride;
    a.out_group_stride = p.dBC_group_stride > 0 ? (size_t)p.dBC_group_stride : (size_t)f.dstate * f.seqlen;
    a.dz_batch_stride = p.ddt_batch_stride; a.dz_group_stride = p.ddt_group_stride; a.dz_rank_stride = p.ddt_rank_stride;
    // the dt-factor gradient in this launch (finish_dt_weight; never together with the fused-delta form, which has its own rows)
    a.ddelta = p.ddelta; a.dd_batch_stride = p.ddelta_batch_stride; a.dd_d_stride = p.ddelta_d_stride;
    a.dtw = nullptr; a.dtR = 0; a.dt_blocks = 0;
    const bool want_dt = p.finish_dt_weight != nullptr;
    if (want_dt) {
        if (f.dt_weight || !p.ddt || !p.ddelta || !scan_finish_dt_ok(f.seqlen, p.finish_dt_rank)) return OSS_ERR_SHAPE;
        const auto al8 = [](const void *q, size_t m) { return (reinterpret_cast<uintptr_t>(q) & (m - 1)) == 0; };
        const size_t va = sizeof(T) == 4 ? 16 : 8;
        if (!al8(p.ddelta, va) || !al8(p.ddt, va) || p.ddelta_batch_stride % 4 || p.ddelta_d_stride % 4 || p.ddt_batch_stride % 4 ||
            p.ddt_group_stride % 4 || p.ddt_rank_stride % 4)
            return OSS_ERR_SHAPE;
        a.dtw = p.finish_dt_weight; a.dtR = p.finish_dt_rank; a.dZ = p.ddt;
        a.dt_blocks = (unsigned)((size_t)f.batch * f.n_groups * ((f.seqlen + kFinishDtSteps - 1) / kFinishDtSteps));
    }
    if ((size_t)f.batch * f.n_groups + 1 > 65535 || 2 * f.dstate + a.R > 65535) return OSS_ERR_SHAPE;
    // four time steps per lane when every partial row and every output row starts 16 / 8-byte aligned
    const auto al = [](const void *q, size_t m) { return (reinterpret_cast<uintptr_t>(q) & (m - 1)) == 0; };
    const size_t oa = sizeof(T) == 4 ? 16 : 8;
    bool vec = f.seqlen % 4 == 0 && a.out_group_stride % 4 == 0 && al(ws.bc, 16) && al(p.dB, oa) && al(p.dC, oa);
    if (a.R) vec = vec && al(p.ddt, oa) && a.dz_batch_stride % 4 == 0 && a.dz_group_stride % 4 == 0 && a.dz_rank_stride % 4 == 0;
    const int V = vec ? 4 : 1;
    const unsigned gx = (unsigned)((f.seqlen + 256 * V - 1) / (256 * V)), gy = (unsigned)(2 * f.dstate + a.R);
    const unsigned dt_slabs = a.dt_blocks ? (a.dt_blocks + gx * gy - 1) / (gx * gy) : 0;
    if ((size_t)f.batch * f.n_groups + 1 + dt_slabs > 65535) return OSS_ERR_SHAPE;
    const dim3 grid(gx, gy, (unsigned)(f.batch * f.n_groups + 1) + dt_slabs);
    if (g_finish_timer) g_finish_timer->begin(stream);
    if (vec) hipLaunchKernelGGL((oss_scan_bwd_finish<T, 4, PB>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((oss_scan_bwd_finish<T, 1, PB>), grid, dim3(256), 0, stream, a);
    if (g_finish_timer) g_finish_timer->end(stream);
    return (int)hipGetLastError();
}

template <typename K, typename... Extra>
static int launch_main(K kern, size_t smem, LdsGate &gate, unsigned nblocks, int nthreads,
                       const oss_scan_bwd_params &p, const BwdWs &ws, hipStream_t stream, LaunchTimer *timer, Extra... extra) {
    if (const int e = gate.ensure(reinterpret_cast<const void *>(kern), smem)) return e;
    if (timer) timer->begin(stream);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(nthreads), smem, stream, p, ws, extra...);
    if (timer) timer->end(stream);
    return (int)hipGetLastError();
}

template <typename T, int LPR, int I, int WAVES, int NBB, int SPS, int MINW>
static int launch_bwd(const oss_scan_bwd_params &p, hipStream_t stream, LaunchTimer *timer) {
    constexpr int ROWS = WAVES * (64 / LPR);
    constexpr int TC = LPR * I;
    const oss_scan_fwd_params &f = p.f;
    BwdWs ws;
    float *wdD, *wdb;
    int rc = carve_ws(p, ROWS, ws, wdD, wdb);
    if (rc != OSS_OK) return rc;
    const size_t smem = sizeof(float) * (2 * (size_t)NBB * TC + 2 * (size_t)SPS * ROWS * TC + 3 * (size_t)f.dstate * ROWS + ROWS);
    if constexpr (!(I == 4 && WAVES == 4)) {
        // large dstate: the per-row state tables no longer fit next to the tiles and slabs -> the small-shape variant.  The
        // workspace was carved for THIS variant's row tiles, so it is carved again there (the query sizes for 4-row tiles)
        if (smem > kMaxLdsBytes) return launch_bwd<T, 64, 4, 4, 16, 1, 3>(p, stream, timer);
    }
    static LdsGate gate;
    rc = launch_main(oss_scan_bwd_kernel<T, LPR, I, WAVES, NBB, SPS, MINW>, smem, gate,
                     (unsigned)(f.batch * f.n_groups * ws.tiles), WAVES * 64, p, ws, stream, timer);
    if (rc != OSS_OK) return rc;
    return launch_finish<T>(p, ws, wdD, wdb, stream);
}

#ifndef OSS_CARRY_WAVES
#define OSS_CARRY_WAVES 0   // 0 = the main kernel's row tile
#endif
constexpr int kCarryWaves = OSS_CARRY_WAVES;

// round-2 kernel (oss_scan_bwd_v2.h): lane-resident per-state scalars, register-prefetched tiles, one barrier per state
template <typename T, int WAVES, int NBB, int MINW, bool FD = false, bool PB = false>
static int launch_bwd2(const oss_scan_bwd_params &p, int seg_req, hipStream_t stream, LaunchTimer *timer) {
    if constexpr (!FD) {
        if (p.f.dt_weight) {
            if constexpr (kBuildFusedDt) return launch_bwd2<T, WAVES, NBB, MINW, true>(p, seg_req, stream, timer);
            else return OSS_ERR_SHAPE;   // this library was built without OSS_WITH_FUSED_DT
        }
    }
    constexpr int TC = 512;
    const oss_scan_fwd_params &f = p.f;
    const int rows_per_group = f.dim / f.n_groups;
    const int tiles = (rows_per_group + WAVES - 1) / WAVES;
    if constexpr (!PB && kPartialsBf16Ok<T, FD>) {
        // bf16 row-tile partials (oss_scan_bwd_v2.h: kV2Bf16Partials): only when THIS call asks for them (tune_partials == 2)
        const bool lane_states = kBuildLaneStates && f.hs != nullptr;
        if (p.tune_partials == 2 && !lane_states)
            return launch_bwd2<T, WAVES, NBB, MINW, false, true>(p, seg_req, stream, timer);
    }
    const unsigned wgs = (unsigned)(f.batch * f.n_groups * tiles);
    const int n_chunks = (f.seqlen + TC - 1) / TC;
    int n_seg = FD ? 1 : scan_pick_segments(wgs, n_chunks, seg_req, 0.3);
    int cps = n_chunks;
    if (n_seg > 1) { cps = (n_chunks + n_seg - 1) / n_seg; n_seg = (n_chunks + cps - 1) / cps; }
    // carry segments per main segment (BwdSeg; oss_host.h: scan_carry_split)
    int csub = scan_carry_split((long)wgs, n_seg, cps, n_chunks, f.tune_carry_split);
    int ccps = (cps + csub - 1) / csub, n_cseg = n_seg > 1 ? n_seg * csub : 0;
    BwdWs ws;
    float *wdD, *wdb, *carry = nullptr;
    int rc = carve_ws(p, WAVES, ws, wdD, wdb, n_seg, &carry, n_cseg);
    if (rc != OSS_OK && n_seg > 1 && csub > 1) {   // room for the main segments' carry pairs but not the finer pieces' 
        csub = 1; ccps = cps; n_cseg = n_seg;
        rc = carve_ws(p, WAVES, ws, wdD, wdb, n_seg, &carry, n_cseg);
    }
    if (rc != OSS_OK && n_seg > 1) {   // a caller that sized the workspace for the unsegmented form
        n_seg = 1; cps = n_chunks;
        rc = carve_ws(p, WAVES, ws, wdD, wdb);
    }
    if (rc != OSS_OK) return rc;
    g_last_bwd_segments.store(n_seg);
    // lane states saved by the forward pass (f.hs): the kernels that load them instead of re-running the forward recurrence
    const bool hs = kBuildLaneStates && !FD && !PB && f.hs != nullptr;
    // two tile buffers, two slab buffers (+ dt weights | + two buffers of this wave's lane states)
    const bool slab_q = kV2SlabQ && !FD && !(hs && WAVES > 8);   // oss_scan_bwd2_kernel: SQ
    const size_t smem = sizeof(float) * (4 * (size_t)NBB * TC + 4 * (size_t)WAVES * (slab_q ? kSlabA : TC) +
                                         (FD ? WAVES * kMaxDtRank : 0) + (hs ? 2 * (size_t)WAVES * NBB * 64 : 0));
    g_last_bwd_lane_states.store(hs ? 1 : 0);
    if constexpr (!FD) {
        if (n_seg > 1) {
            const BwdSeg sg{carry, n_seg, cps, csub, ccps, n_cseg};
            // the carry pass is per ROW (nothing is summed over rows), so its row tile is free: 4-row workgroups (three times the
            // workgroups, several per CU) measured SLOWER than the main kernel's tile -- u:(4,192,16384) 0.355 against 0.340 ms for
            // the whole call -- because every workgroup stages the group's C rows again
            constexpr int CW = kCarryWaves > 0 ? kCarryWaves : WAVES;
            const int ctiles = (rows_per_group + CW - 1) / CW;
            auto kc = oss_scan_bwd_carry_kernel<T, CW>;
            if (timer) { timer->segmented(); timer->begin(stream); }
            if (g_finish_timer) g_finish_timer->segmented();
            hipLaunchKernelGGL(kc, dim3((unsigned)(f.batch * f.n_groups * ctiles * (n_cseg - csub))), dim3(CW * 64), sizeof(float) * 2 * kNB * TC,
                               stream, p, sg, ctiles);
            static LdsGate gate_s, gate_sh;
            bool launched = false;
            if constexpr (kBuildLaneStates && !PB) {
                if (hs) {
                    auto km = oss_scan_bwd2_kernel<T, WAVES, NBB, MINW, false, true, true>;
                    if (const int e = gate_sh.ensure(reinterpret_cast<const void *>(km), smem)) return e;
                    hipLaunchKernelGGL(km, dim3(wgs * (unsigned)n_seg), dim3(WAVES * 64), smem, stream, p, ws, sg);
                    launched = true;
                }
            }
            if (!launched) {
                auto km = oss_scan_bwd2_kernel<T, WAVES, NBB, MINW, false, true, false, PB>;
                if (const int e = gate_s.ensure(reinterpret_cast<const void *>(km), smem)) return e;
                hipLaunchKernelGGL(km, dim3(wgs * (unsigned)n_seg), dim3(WAVES * 64), smem, stream, p, ws, sg);
            }
            if (timer) timer->end(stream);
            rc = (int)hipGetLastError();
            if (rc != OSS_OK) return rc;
            return launch_finish<T, PB>(p, ws, wdD, wdb, stream, n_seg);
        }
        if constexpr (kBuildLaneStates && !PB) {
            if (hs) {
                static LdsGate gate_h;
                rc = launch_main(oss_scan_bwd2_kernel<T, WAVES, NBB, MINW, false, false, true>, smem, gate_h, wgs, WAVES * 64, p, ws,
                                 stream, timer, BwdSeg{nullptr, 1, n_chunks, 1, n_chunks, 1});
                if (rc != OSS_OK) return rc;
                return launch_finish<T, false>(p, ws, wdD, wdb, stream);
            }
        }
    }
    static LdsGate gate;
    rc = launch_main(oss_scan_bwd2_kernel<T, WAVES, NBB, MINW, FD, false, false, PB>, smem, gate, wgs, WAVES * 64, p, ws, stream, timer,
                     BwdSeg{nullptr, 1, n_chunks, 1, n_chunks, 1});
    if (rc != OSS_OK) return rc;
    return launch_finish<T, PB>(p, ws, wdD, wdb, stream);
}

// variant table (numbers kept from rounds 1-3; 0 and 2..9 -- the other round-1 row tiles and the packed two-states-per-pass
// kernels, all measured slower than the round-2 kernel -- were removed in round 4 and now mean variant 1):
//   1: round-1 kernel, 64 lanes x 4 items, 4 rows/WG, 16 states per LDS tile (TC 256, 40 KiB LDS): short sequences, few rows
//      per group, dstate > 64 (the round-2 kernel keeps one lane per state) and the fallback when a launch would not fit LDS
//  10: round-2 kernel (oss_scan_bwd_v2.h), 12 rows/WG (128 KiB LDS)   11: 8 rows/WG   12: 6 rows/WG   13: 4 rows/WG
static const int kBwdRows[] = {4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 12, 8, 6, 4};
int scan_bwd_rows_per_wg(int variant) { return kBwdRows[(variant < 0 || variant > 13) ? 1 : variant]; }

template <typename T>
static int scan_bwd_dispatch_(const oss_scan_bwd_params &p, int variant, int seg_req, hipStream_t stream, LaunchTimer *timer);
template <typename T>
int scan_bwd_dispatch(const oss_scan_bwd_params &p, int variant, int seg_req, hipStream_t stream, LaunchTimer *timer,
                      LaunchTimer *finish_timer) {
    g_finish_timer = finish_timer;
    g_last_bwd_segments.store(1);
    const int rc = scan_bwd_dispatch_<T>(p, variant, seg_req, stream, timer);
    g_finish_timer = nullptr;
    return rc;
}
template <typename T>
static int scan_bwd_dispatch_(const oss_scan_bwd_params &p, int variant, int seg_req, hipStream_t stream, LaunchTimer *timer) {
    if (p.f.dt_weight) {   // delta computed inside the scan: round-2 kernels only
        if (p.f.dstate > 64 || p.f.dt_rank < 1 || p.f.dt_rank > kMaxDtRank || !p.ddt || !p.ddt_weight) return OSS_ERR_SHAPE;
        if (variant < 10) variant = 13;
    }
    if (variant >= 10 && p.f.dstate > 64) variant = 1;   // the round-2 kernel keeps one lane per state
    switch (variant) {
        case 10: return launch_bwd2<T, 12, 4, 3>(p, seg_req, stream, timer);
        case 11: return launch_bwd2<T, 8, 4, 2>(p, seg_req, stream, timer);
        case 12: return launch_bwd2<T, 6, 4, 2>(p, seg_req, stream, timer);
        case 13: return launch_bwd2<T, 4, 4, 2>(p, seg_req, stream, timer);   // 4-row workgroups for calls with few rows (one wave per SIMD, no spills)
        default: return launch_bwd<T, 64, 4, 4, 16, 1, 3>(p, stream, timer);
    }
}

template int scan_bwd_dispatch<float>(const oss_scan_bwd_params &, int, int, hipStream_t, LaunchTimer *, LaunchTimer *);
template int scan_bwd_dispatch<bf16_t>(const oss_scan_bwd_params &, int, int, hipStream_t, LaunchTimer *, LaunchTimer *);
template int scan_bwd_dispatch<f16_t>(const oss_scan_bwd_params &, int, int, hipStream_t, LaunchTimer *, LaunchTimer *);

}  // namespace oss
