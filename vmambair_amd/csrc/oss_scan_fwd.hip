// oss_scan_fwd.hip -- selective-scan forward for gfx950 (replaces the reference's
// selective_scan_fwd_kernel, cus/selective_scan_fwd_kernel.cuh:61-172; arithmetic per SURVEY.md
// Appendix A).
//
// Decomposition (MI355X-first, not the reference's block-scan-per-state):
//   * one workgroup = one (batch b, group g, tile of ROWS rows of that group).  The B/C rows of
//     (b, g) are shared by every row of the group, so they are staged ONCE per chunk into LDS as
//     fp32 and re-read by all ROWS rows (the reference re-loads them from global per row).
//   * one row (b, d) is owned by LPR consecutive lanes of ONE wave for the whole sequence; each
//     lane owns I consecutive time steps of the current chunk (TC = LPR*I steps per chunk).  The
//     carry between chunks never leaves the wave, so there is no block-level scan and no barrier
//     per state: the only barriers are the two around the LDS tile refill.
//   * per state: thread-local sequential recurrence over the I items (registers), a DPP
//     (row_shr / row_bcast) inclusive scan of the per-lane (prod a, h) pairs over the LPR lanes,
//     then a second sequential pass that re-runs the recurrence from the exact incoming state and
//     accumulates y += C*h.  One v_exp_f32 per (element, state); the per-lane product of a's is
//     exp2(A * sum(delta)) (one extra exp per lane and state) instead of I multiplies.
#include "oss_device.h"
#include "oss_host.h"

namespace oss {

// FD: delta evaluated here from the rank-R factor (include/vmambair_oss.h: dt_weight) -- its own instantiation, so that the
// plain form keeps the register allocation it was tuned with
//
// SEG: time-segmented launches for calls whose (batch, group, row tile) grid leaves CUs idle (RealSR tiles at batch 1,
// the Deraining tree's level 0).  The reference walks the chunks of a row sequentially inside one block
// (cus/selective_scan_fwd_kernel.cuh:101-102,147-158) and so does SEG = 0 here.  With SEG the sequence is cut into
// n_seg segments of cps chunks, one workgroup per (batch, group, row tile, segment), two launches:
//   SEG = 1 (local pass, segments 0 .. n_seg-2): the recurrence from a ZERO state over the segment, nothing but the
//            B side of the step (no C, no y, no stores) -> carry[b, d, s, n] = (prod of a over the segment, h at its end);
//   SEG = 2 (real pass, every segment): the state entering segment s is the fold of the pairs of segments 0 .. s-1 in
//            the scan monoid (selective_scan_common.h:89-96) -- s fused multiply-adds per (row, state) -- then the plain
//            step over the segment's chunks.
// Same arithmetic per step as the unsegmented kernel; the state entering a segment is associated differently
// (segment-local partial states folded instead of one running chain), i.e. equal up to fp32 round-off.
struct FwdSeg {
    float *carry;   // [batch][dim][n_cseg][dstate][2] floats
    int n_seg, cps; // segments per row, chunks (of TC steps) per segment
    // (round 5) the local pass has its own, finer segmentation (as the backward's carry pass, oss_scan_bwd_v2.h: BwdSeg): csub local
    // pieces of ccps = ceil(cps / csub) chunks per segment (the last one shorter), n_cseg = n_seg * csub slots per row; the real pass
    // folds seg * csub pairs
    int csub, ccps, n_cseg;
};
template <typename T, int LPR, int I, int WAVES, bool FD = false, int SEG = 0>
__global__ void __launch_bounds__(WAVES * 64)
oss_scan_fwd_kernel(const oss_scan_fwd_params p, const FwdSeg sg) {
    constexpr int RPW = 64 / LPR;      // rows per wave
    constexpr int ROWS = WAVES * RPW;  // rows per workgroup
    constexpr int TC = LPR * I;        // time steps per chunk
    constexpr int NT = WAVES * 64;
    static_assert(TC % kScanChunk == 0, "chunk must be a multiple of the x granularity");
    static_assert(I % 4 == 0, "");

    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef OSS_EXP_FWD_NO_TILE_PAD   // (OSS_EXP_*: A-B timing builds only, tools/build_experiment.sh)
    constexpr bool kPad = false;
#else
    constexpr bool kPad = true;       // bank-conflict-free staging writes (oss_device.h: tile_off_pad)
#endif
    constexpr int TR = kPad ? kTileRowPad<LPR, I> : TC;   // floats per state row of a tile
    constexpr int PL = kPad ? LPR * 4 + kTilePlanePad<I> : LPR * 4;   // floats between the quads of a lane
    float *sB = smem;                 // [kNB][TR]
    float *sC = smem + kNB * TR;      // [kNB][TR]
    float *carH = smem + 2 * kNB * TR;               // [dstate][ROWS]   h entering the chunk
    float *carP = carH + (size_t)p.dstate * ROWS;    // [dstate][ROWS]   prod of a since t = 0
    float *sA2 = carP + (size_t)p.dstate * ROWS;     // [dstate][ROWS]   A * log2(e)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pos = lane & (LPR - 1);
    const int wrow = wave * RPW + lane / LPR;  // row slot inside the workgroup
    const bool seg_first = (pos == 0), seg_last = (pos == LPR - 1);

    const int L = p.seqlen, N = p.dstate;
    const int rows_per_group = p.dim / p.n_groups;
    const int tiles_per_group = (rows_per_group + ROWS - 1) / ROWS;
    // workgroup -> (batch, group, [segment,] row tile).  The row tiles of one (batch, group, segment) stage the same B / C rows;
    // consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so when the number of such sets divides
    // by 8 the tiles of a set get ids with the same residue mod 8 and share ONE L2's copy of B / C instead of pulling eight
    // (as the backward does, oss_scan_bwd_v2.h; round 3 measured FETCH = 2.5 x the input bytes for this kernel at
    // u:(8,384,4096): tile = id % 8 dealt a group's eight tiles to the eight XCDs).  Speed only: nothing depends on placement.
    int bid = blockIdx.x, tile;
    const int ns = SEG == 0 ? 1 : (SEG == 1 ? (sg.n_seg - 1) * sg.csub : sg.n_seg);   // the last segment has no successor: no local pass
#ifdef OSS_EXP_FWD_NO_XCD   // (OSS_EXP_*: A-B timing builds only, tools/build_experiment.sh)
    constexpr bool kXcdOrder = false;
#else
    constexpr bool kXcdOrder = true;
#endif
    // (round 5) Generalised from "the number of sets divides by 8" to "the number of WORKGROUPS divides by 8": XCD x (ids with residue x)
    // takes the x-th eighth of the (set, tile) pairs in set-major order, so it works on whole sets except at the two ends of its
    // run.  u:(1,384,25600) in 9 segments is 36 sets x 8 tiles: the old rule did not apply and every XCD fetched every set's B / C
    // (FETCH 87 MB for 36 MB of inputs, profiles/r05_pmc_FETCH_SIZE_realsr.txt).
    const int total = p.batch * p.n_groups * ns * tiles_per_group;
    if (kXcdOrder && total % 8 == 0) {
        const int i = (bid & 7) * (total >> 3) + (bid >> 3);
        tile = i % tiles_per_group;
        bid = i / tiles_per_group;
    } else {
        tile = bid % tiles_per_group; bid /= tiles_per_group;
    }
    int seg = 0;
    if constexpr (SEG != 0) { seg = bid % ns; bid /= ns; }
    const int g = bid % p.n_groups;
    const int b = bid / p.n_groups;
    const int row_in_group = tile * ROWS + wrow;
    const bool row_valid = row_in_group < rows_per_group;
    const int d = g * rows_per_group + (row_valid ? row_in_group : 0);
    const bool rev = g >= p.rev_group_start;                       // time-mirrored direction
    const int d_u = p.u_row_mod > 0 ? d % p.u_row_mod : d;          // directions sharing one copy of u

    const T *u_row = reinterpret_cast<const T *>(p.u) + b * p.u_batch_stride + d_u * p.u_d_stride;
    constexpr bool fused_dt = FD;   // delta = dt_weight[d, :] . z[b, g, :, t] evaluated here
    const T *dt_row = reinterpret_cast<const T *>(p.delta) + b * p.delta_batch_stride +
                      (fused_dt ? g * p.dt_group_stride : d * p.delta_d_stride);
    const float *dt_w = fused_dt ? p.dt_weight + (size_t)d * p.dt_rank : nullptr;
    T *out_row = reinterpret_cast<T *>(p.out) + b * p.out_batch_stride + d * p.out_d_stride;
    const T *gB = reinterpret_cast<const T *>(p.B) + b * p.B_batch_stride + g * p.B_group_stride;
    const T *gC = reinterpret_cast<const T *>(p.C) + b * p.C_batch_stride + g * p.C_group_stride;
    const float Dd = p.D ? p.D[d] : 0.f;
    const float bias = p.delta_bias ? p.delta_bias[d] : 0.f;
    const int n_xchunks = (L + kScanChunk - 1) / kScanChunk;
    float *x_row = p.x + ((size_t)b * p.dim + d) * n_xchunks * 2 * N;
    const size_t hs_stride = lane_state_stride(L);
    float *hs_line = (kBuildLaneStates && SEG != 1 && p.hs) ? p.hs + ((size_t)b * p.dim + d) * N * hs_stride : nullptr;

    // carries start at (h, P) = (0, 1); A is pre-scaled once (fwd_kernel.cuh:125-127)
    for (int idx = tid; idx < N * ROWS; idx += NT) {
        const int n = idx / ROWS, r = idx - n * ROWS;
        const int rg = tile * ROWS + r;
        const int dd = g * rows_per_group + (rg < rows_per_group ? rg : 0);
        float h0 = 0.f, P0 = 1.f;
        if constexpr (SEG == 2) {   // fold the local pairs of the earlier segments, in time order
            const float2 *cr = reinterpret_cast<const float2 *>(sg.carry) + (((size_t)b * p.dim + dd) * sg.n_cseg) * N + n;
            for (int j = 0; j < seg * sg.csub; ++j) {
                const float2 pr = cr[(size_t)j * N];
                h0 = __builtin_fmaf(pr.x, h0, pr.y);
                P0 *= pr.x;
            }
        }
        carH[idx] = h0;
        carP[idx] = P0;
        const float av = p.A[dd * p.A_d_stride + n];
        sA2[idx] = (p.a_log_form ? -__expf(av) : av) * kLog2e;
    }

    const int n_chunks = (L + TC - 1) / TC;
    // SEG = 1: `seg` counts local pieces, csub per main segment: piece j of main segment s = chunks s cps + j ccps .. (the last piece of
    // a segment is shorter when ccps does not divide cps)
    int c_begin = 0, c_end = n_chunks;
    if constexpr (SEG == 1) {
        const int s_ = seg / sg.csub, j_ = seg - s_ * sg.csub;
        c_begin = s_ * sg.cps + j_ * sg.ccps;
        c_end = min(min(c_begin + sg.ccps, (s_ + 1) * sg.cps), n_chunks);
    } else if constexpr (SEG == 2) {
        c_begin = seg * sg.cps;
        c_end = min(n_chunks, c_begin + sg.cps);
    }
    for (int c = c_begin; c < c_end; ++c) {
        const int t0 = c * TC;
        const int tl = t0 + pos * I;  // first time step of this lane
        const int valid = max(0, min(I, L - tl));
        float dl[I], w[I], y[I];
        {
            float uu[I];
            if constexpr (FD) {   // all rank rows and u in flight together, then the projection
                DtRows<T, I> zr;
                RawItems<T, I> ru;
                if (dt_rows_fast_ok<I>(dt_row, p.dt_rank_stride, tl, valid, L, rev) && raw_fast_ok<I>(u_row, tl, valid, L, rev)) {
                    dt_rows_load<I>(zr, dt_row, p.dt_rank_stride, dt_w, p.dt_rank, tl, valid, L, rev, true);
                    ru = load_raw_fast<I>(u_row, tl, L, rev);
                } else {
                    dt_rows_load<I>(zr, dt_row, p.dt_rank_stride, dt_w, p.dt_rank, tl, valid, L, rev, false);
                    ru = load_raw_slow<I>(u_row, tl, valid, L, rev);
                }
                dt_rows_apply<I>(zr, p.dt_rank, rev, dl);
                unpack_raw_dir<I>(ru, rev, uu);
            } else {
                load_items_dir<I>(u_row, tl, valid, L, rev, uu);
                load_items_dir<I>(dt_row, tl, valid, L, rev, dl);
            }
#pragma unroll
            for (int i = 0; i < I; ++i) {
                float x = dl[i] + bias;
                if (p.delta_softplus) { float e; x = softplus_thr(x, e); }
                x = (i < valid) ? x : 0.f;  // identity element beyond the end (fwd_kernel.cuh:138-142)
                dl[i] = x;
                w[i] = x * uu[i];
                y[i] = Dd * uu[i];
            }
        }
        float S = 0.f;
#pragma unroll
        for (int i = 0; i < I; ++i) S += dl[i];

        // does this lane end on an x boundary (or hold the last element)?
        const int tend = tl + I;  // one past this lane's last step
        const bool writes_x = row_valid && valid > 0 && ((tend % kScanChunk) == 0 || tend >= L);
        const int xc = (tl / kScanChunk);

        for (int n0 = 0; n0 < N; n0 += kNB) {
            const int nb = min(kNB, N - n0);
            __syncthreads();  // everyone is done with the previous tile (and the carry init)
            stage_bc_tiles<T, LPR, I, NT, SEG != 1, kPad>(sB, sC, gB + (int64_t)n0 * p.B_dstate_stride,
                                          gC + (int64_t)n0 * p.C_dstate_stride, p.B_dstate_stride,
                                          p.C_dstate_stride, nb, t0, L, rev, tid);
            __syncthreads();
            for (int nn = 0; nn < nb; ++nn) {
                const int n = n0 + nn;
                const float A2 = sA2[n * ROWS + wrow];
                const float hc = carH[n * ROWS + wrow];
                const float Pc = carP[n * ROWS + wrow];
                float a[I], bb[I];
                float h = 0.f;
                const float *tb = sB + nn * TR + pos * 4;
#pragma unroll
                for (int k = 0; k < I / 4; ++k) {
                    const f32x4 bv = *reinterpret_cast<const f32x4 *>(tb + k * PL);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = 4 * k + j;
                        a[i] = exp2_hw(dl[i] * A2);
                        bb[i] = bv[j] * w[i];
                        h = (i == 0) ? bb[0] : __builtin_fmaf(a[i], h, bb[i]);
                    }
                }
                float P = exp2_hw(S * A2);
                segment_scan<LPR>(P, h);
                // fold in the state entering the chunk; hfull = state after this lane's last step
                const float hfull = __builtin_fmaf(P, hc, h);
                const float Pfull = P * Pc;
                float hin = shift_from_prev_lane(hfull, hc, seg_first);
                if (seg_last) { carH[n * ROWS + wrow] = hfull; carP[n * ROWS + wrow] = Pfull; }
                if constexpr (SEG != 1) {
                    if (writes_x) {
                        float2 st = make_float2(Pfull, hfull);
                        *reinterpret_cast<float2 *>(x_row + (size_t)xc * 2 * N + 2 * n) = st;
                    }
                    const float *tc = sC + nn * TR + pos * 4;
                    h = hin;
                    float h_mid = 0.f;   // I = 16: the state after this lane's 8th step
#pragma unroll
                    for (int k = 0; k < I / 4; ++k) {
                        const f32x4 cv = *reinterpret_cast<const f32x4 *>(tc + k * PL);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int i = 4 * k + j;
                            h = __builtin_fmaf(a[i], h, bb[i]);
                            if (I == 16 && i == 7) h_mid = h;
                            y[i] = __builtin_fmaf(cv[j], h, y[i]);
                        }
                    }
                    // lane states for the backward (include/vmambair_oss.h: hs): the state ENTERING every 8-step block whose
                    // first step lies inside the sequence -- a by-product of this pass
                    if (hs_line && row_valid && valid > 0) {
                        float *hl = hs_line + (size_t)n * hs_stride + (tl >> 3);
                        if constexpr (I == 16) {
                            if (valid > 8) *reinterpret_cast<float2 *>(hl) = make_float2(hin, h_mid);
                            else hl[0] = hin;
                        } else if constexpr (I == 8) {
                            hl[0] = hin;
                        } else {   // I = 4: every other lane starts a block
                            if ((tl & 7) == 0) hl[0] = hin;
                        }
                    }
                }
            }
        }
        if constexpr (SEG != 1) {
            if (row_valid) store_items_dir<I>(out_row, tl, valid, L, rev, y);
        }
    }
    if constexpr (SEG == 1) {   // the segment's pair: what the later segments fold
        __syncthreads();
        for (int idx = tid; idx < N * ROWS; idx += NT) {
            const int n = idx / ROWS, r = idx - n * ROWS;
            const int rg = tile * ROWS + r;
            if (rg < rows_per_group) {
                const int dd = g * rows_per_group + rg;
                float2 *cr = reinterpret_cast<float2 *>(sg.carry) + (((size_t)b * p.dim + dd) * sg.n_cseg + seg) * N + n;
                *cr = make_float2(carP[idx], carH[idx]);
            }
        }
    }
}

template <typename T, int LPR, int I, int WAVES, bool FD = false>
static int launch_fwd(const oss_scan_fwd_params &p, int seg_req, hipStream_t stream) {
    if constexpr (!FD) {
        if (p.dt_weight) {
            if constexpr (kBuildFusedDt) return launch_fwd<T, LPR, I, WAVES, true>(p, seg_req, stream);
            else return OSS_ERR_SHAPE;   // this library was built without OSS_WITH_FUSED_DT
        }
    }
    constexpr int ROWS = WAVES * (64 / LPR);
    constexpr int TC = LPR * I;
    const int rows_per_group = p.dim / p.n_groups;
    const int tiles = (rows_per_group + ROWS - 1) / ROWS;
#ifdef OSS_EXP_FWD_NO_TILE_PAD
    constexpr int TR = TC;
#else
    constexpr int TR = kTileRowPad<LPR, I>;
#endif
    const size_t smem = sizeof(float) * (2 * (size_t)kNB * TR + 3 * (size_t)p.dstate * ROWS);
    if constexpr (!(LPR == 64 && I == 4 && WAVES == 4)) {
        // large dstate: tiles + per-row carries no longer fit 160 KiB -> the small-shape variant (4 rows, 256-step chunks)
        if (smem > kMaxLdsBytes) return launch_fwd<T, 64, 4, 4, FD>(p, seg_req, stream);
    }
    const unsigned wgs = (unsigned)(p.batch * p.n_groups * tiles);
    const int n_chunks = (p.seqlen + TC - 1) / TC;
    int n_seg = FD ? 1 : scan_pick_segments(wgs, n_chunks, seg_req, 0.55);
    // the carry slots are per LOCAL-pass piece.  A caller's workspace that holds the main segments' slots but not the finer
    // pieces' keeps the segments and drops the finer split; only one that does not hold even those goes unsegmented
    if (n_seg > 1 && !p.workspace) n_seg = 1;
    g_last_fwd_segments.store(1);
    if constexpr (!FD) {
        if (n_seg > 1) {
            FwdSeg sg;
            sg.carry = reinterpret_cast<float *>(p.workspace);
            sg.cps = (n_chunks + n_seg - 1) / n_seg;
            sg.n_seg = (n_chunks + sg.cps - 1) / sg.cps;
            sg.csub = scan_carry_split((long)wgs, sg.n_seg, sg.cps, n_chunks, p.tune_carry_split);
            if (p.workspace_bytes < scan_carry_bytes(p.batch, p.dim, p.dstate, sg.n_seg * sg.csub)) sg.csub = 1;
            sg.ccps = (sg.cps + sg.csub - 1) / sg.csub;
            sg.n_cseg = sg.n_seg * sg.csub;
            if (p.workspace_bytes < scan_carry_bytes(p.batch, p.dim, p.dstate, sg.n_cseg)) sg.n_seg = 1;
          if (sg.n_seg > 1) {
            g_last_fwd_segments.store(sg.n_seg);
            auto k1 = oss_scan_fwd_kernel<T, LPR, I, WAVES, false, 1>;
            auto k2 = oss_scan_fwd_kernel<T, LPR, I, WAVES, false, 2>;
            static LdsGate gate1, gate2;
            if (const int e = gate1.ensure(reinterpret_cast<const void *>(k1), smem)) return e;
            if (const int e = gate2.ensure(reinterpret_cast<const void *>(k2), smem)) return e;
            hipLaunchKernelGGL(k1, dim3(wgs * (unsigned)((sg.n_seg - 1) * sg.csub)), dim3(WAVES * 64), smem, stream, p, sg);
            hipLaunchKernelGGL(k2, dim3(wgs * (unsigned)sg.n_seg), dim3(WAVES * 64), smem, stream, p, sg);
            return (int)hipGetLastError();
          }
        }
    }
    auto kern = oss_scan_fwd_kernel<T, LPR, I, WAVES, FD, 0>;
    static LdsGate gate;
    if (const int e = gate.ensure(reinterpret_cast<const void *>(kern), smem)) return e;
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(WAVES * 64), smem, stream, p, FwdSeg{nullptr, 1, n_chunks, 1, n_chunks, 1});
    return (int)hipGetLastError();
}

// variant table: (lanes per row, items per lane, waves per workgroup)
//   0: 64 x 8  x 8   (TC  512,  8 rows/WG)   many waves for small batches
//   1: 32 x 16 x 8   (TC  512, 16 rows/WG)
//   2: 16 x 16 x 4   (TC  256, 16 rows/WG)   fewest scan steps, needs many rows
//   3: 64 x 16 x 8   (TC 1024,  8 rows/WG)
//   4: 64 x 4  x 4   (TC  256,  4 rows/WG)   short sequences / few rows per group
//   6: 64 x 16 x 12  (TC 1024, 12 rows/WG): row counts that give <= 256 such workgroups
//   (5 = 64 x 8 x 12 and 7 = 64 x 16 x 6 were never picked by scan_fwd_pick_variant: removed in round 4, the numbers now mean 4)
template <typename T>
int scan_fwd_dispatch(const oss_scan_fwd_params &p, int variant, int seg_req, hipStream_t stream) {
    switch (variant) {
        case 0: return launch_fwd<T, 64, 8, 8>(p, seg_req, stream);
        case 1: return launch_fwd<T, 32, 16, 8>(p, seg_req, stream);
        case 2: return launch_fwd<T, 16, 16, 4>(p, seg_req, stream);
        case 3: return launch_fwd<T, 64, 16, 8>(p, seg_req, stream);
        case 6: return launch_fwd<T, 64, 16, 12>(p, seg_req, stream);   // 12 rows per workgroup: one workgroup per CU at 3072 rows
        default: return launch_fwd<T, 64, 4, 4>(p, seg_req, stream);
    }
}

template int scan_fwd_dispatch<float>(const oss_scan_fwd_params &, int, int, hipStream_t);
template int scan_fwd_dispatch<bf16_t>(const oss_scan_fwd_params &, int, int, hipStream_t);
template int scan_fwd_dispatch<f16_t>(const oss_scan_fwd_params &, int, int, hipStream_t);

}  // namespace oss
