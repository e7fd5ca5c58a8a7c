// oss_scan_bwd_v2.h -- round-2 selective-scan backward: the same algorithm, row ownership, workspace and finishing
// kernel as oss_scan_bwd_kernel (oss_scan_bwd.hip; reference cus/selective_scan_bwd_kernel.cuh:66-273), restructured
// around what the round-1 counters and ISA showed (profiles/r01_pmc_sq_scan.txt): that kernel
// issues ~22 vector instructions per (element, state) but a wave spends half its time parked -- every state iteration
// exposed five LDS round trips (three wave-uniform scalars read back from LDS, B/C tiles read right before use), one
// global load (the saved forward state) and two workgroup barriers, with only 3 waves per SIMD to cover them.
//
//   * wave-uniform per-state scalars live in LANES of four registers (lane n = state n: A*log2e, the forward state
//     entering the chunk, the reverse carry dh, the dA accumulator) and are fetched with v_readlane_b32 (an SGPR
//     broadcast, no memory access); the saved forward states of a chunk are ONE global load per chunk (lane n loads
//     state n);
//   * B/C tiles of state n+1 are fetched from LDS into a second register set while state n computes (two sets
//     alternate, so no copies), and stay in registers for all passes of a state (4 b128 reads per state instead of 8);
//   * the cross-row dB/dC slabs are double-buffered: ONE barrier per state instead of two; the slab sum uses 16-byte
//     reads, four waves per state, rotating over the workgroup so that the work is even across SIMDs;
//   * u and softplus'(delta) are not held across the state loop (they are re-read / recomputed once per chunk), which
//     pays for the second tile set in registers;
//   * the B/C tiles of the NEXT batch of states (or of the next chunk's first batch) are fetched from global memory into a
//     few registers before the current batch's state loop and written to a second LDS tile buffer after it: the global
//     latency and the two barriers of a synchronous refill (35 us of 268 at u:(8,384,4096), profiles/r02_scan_bwd_v2_
//     experiments.txt) disappear behind the state loop.
// Needs dstate <= 64 (one lane per state); larger dstate takes the round-1 kernel.
// Included by oss_scan_bwd.hip only (needs BwdWs).
#pragma once
#include "oss_device.h"

namespace oss {

__device__ __forceinline__ float lane_get(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// lane `lane` of vec := a wave-uniform value (a compare + select: v_writelane_b32 needs the value in an SGPR and the
// lane select in M0 on this ISA family, and is not exposed as a builtin by this compiler)
__device__ __forceinline__ float lane_set(float vec, int my_lane, int lane, float uniform_val) {
    return (my_lane == lane) ? uniform_val : vec;
}

template <int I>
__device__ __forceinline__ void read_tile(const float *t, float (&v)[I]) {
#pragma unroll
    for (int k = 0; k < I / 4; ++k) {
        const f32x4 q = *reinterpret_cast<const f32x4 *>(t + k * 256);   // tile_off<64, I>: quads of a lane are 64*4 floats apart
        v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
    }
}

// one 4-position group of a B or C row, as loaded (fp32: 4 words, 16-bit types: 2 words)
template <typename T> struct RawQuad { uint32_t w[sizeof(T)]; };

// memory elements m0 .. m0+3 of a row (zero outside [0, L)); `fast`: all four in range and the address is 8/16-byte aligned
template <typename T>
__device__ __forceinline__ RawQuad<T> load_quad(const T *row, int m0, int L, bool fast) {
    RawQuad<T> r;
    const T *p = row + m0;
    if (fast) {
        if constexpr (sizeof(T) == 4) {
            const u32x4 q = *reinterpret_cast<const u32x4 *>(p);
            r.w[0] = q.x; r.w[1] = q.y; r.w[2] = q.z; r.w[3] = q.w;
        } else {
            const u32x2 q = *reinterpret_cast<const u32x2 *>(p);
            r.w[0] = q.x; r.w[1] = q.y;
        }
    } else {
        uint32_t e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + j;
            const bool ok = m >= 0 && m < L;
            if constexpr (sizeof(T) == 4) e[j] = ok ? __float_as_uint(row[m]) : 0u;
            else e[j] = ok ? (uint32_t)row[m].v : 0u;
        }
        if constexpr (sizeof(T) == 4) { r.w[0] = e[0]; r.w[1] = e[1]; r.w[2] = e[2]; r.w[3] = e[3]; }
        else { r.w[0] = e[0] | (e[1] << 16); r.w[1] = e[2] | (e[3] << 16); }
    }
    return r;
}
template <typename T>
__device__ __forceinline__ f32x4 quad_to_f32(const RawQuad<T> &r, bool rev) {
    float v[4];
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(r.w[j]);
    } else {
        unpack2<T>(r.w[0], v[0], v[1]);
        unpack2<T>(r.w[1], v[2], v[3]);
    }
    return rev ? f32x4{v[3], v[2], v[1], v[0]} : f32x4{v[0], v[1], v[2], v[3]};
}

// FD: delta computed inside the scan from the rank-R factor z (include/vmambair_oss.h: dt_weight); the kernel then emits the
// gradient of z (summed over the rows of the group: slab rounds like dB / dC, two rank rows per round) and the per-row
// gradient of dt_weight instead of ddelta.
#ifdef OSS_EXP_V2_GROUP_PRO
constexpr bool kV2GroupPro = true;
#else
constexpr bool kV2GroupPro = false;
#endif
#ifdef OSS_EXP_V2_GROUP_EPI
constexpr bool kV2GroupEpi = true;
#else
constexpr bool kV2GroupEpi = false;
#endif
#ifdef OSS_EXP_V2_HCV_EARLY
constexpr bool kV2HcvEarly = true;
#else
constexpr bool kV2HcvEarly = false;
#endif
// round 4: (a) the slab in a bank-conflict-free image (SlabQ below) and (b) the chunk-edge factor exp2(A delta_first) of all
// states as ONE v_exp_f32 per chunk (lane n = state n) instead of one per state pass.  OSS_EXP_V2_OLD_SLAB / _OLD_AEDGE
// restore the round-2 forms for A-B timing (tools/build_experiment.sh).
#ifdef OSS_EXP_V2_OLD_SLAB
constexpr bool kV2SlabQ = false;
#else
constexpr bool kV2SlabQ = true;
#endif
#ifdef OSS_EXP_V2_OLD_AEDGE
constexpr bool kV2EdgeLanes = false;
#else
constexpr bool kV2EdgeLanes = true;
#endif
// round 6: the workgroup's dB / dC row-tile partials -- written here, re-read by oss_scan_bwd_finish -- are
// 268 of the call's 417 MB of HBM traffic at u:(8,384,4096) bf16 (algorithmic: 142).  OPT-IN per call (oss_scan_bwd_params.
// tune_partials = 2, bf16 I/O only): the partials (each already the fp32 sum of the tile's <= 12 rows) go out as bf16 -- half the
// partial bytes both ways, traffic 2.9 x -> 2.0 x algorithmic, finishing kernel 18 -> 10 us, headline step +1.7 %
// (profiles/r06_ab_bf16_partials.txt).  NOT the default, because it is not free numerically: every partial is rounded to 8 bits of
// mantissa BEFORE the tiles are summed, an absolute error of 2^-9 x |partial| that survives when the tiles cancel -- at the
// reference's own test grid (randn inputs, |dB| up to 83) one element in 131 072 left the reference's bf16 tolerance (atol 5e-2,
// test_selective_scan.py:400,490-502) with only TWO tiles.  The default keeps the reference's fp32 accumulation
// (cus/selective_scan_bwd_kernel.cuh:208-221) to the letter.  OSS_EXP_V2_F32_PARTIALS compiles the bf16 form out.
#ifdef OSS_EXP_V2_F32_PARTIALS
constexpr bool kV2Bf16Partials = false;
#else
constexpr bool kV2Bf16Partials = true;
#endif
template <typename T, bool FD> constexpr bool kPartialsBf16Ok = kV2Bf16Partials && !FD && std::is_same<T, bf16_t>::value;
// SlabQ: one row's dB (or dC) terms of one state, 512 scan positions = 64 lanes x 2 quads.  The round-2 image was time order
// (lane p wrote its quads at floats 8p and 8p + 4: a 32-byte lane stride, so a 16-lane phase of a ds_write_b128 covered only
// half of the 64 banks, two-way conflicts on every slab write -- SQ_LDS_BANK_CONFLICT 31 % of the LDS cycles,
// profiles/r03_pmc_sq_scan.txt).  Now quad k of lane p sits at float k * kSlabK + 4 p: consecutive lanes write consecutive
// 16-byte pieces; kSlabK = 256 + 32 puts the second quads half a bank sweep away from the first ones, so that the summing
// lanes -- lane s reads quad (s & 1) of position base + (s >> 1), i.e. scan positions base * 8 + 4 s: the partial rows still
// leave as fully coalesced 16-byte stores -- are conflict-free too.
constexpr int kSlabK = 288;            // floats between the two quads of a lane
constexpr int kSlabA = 2 * kSlabK;     // floats per (row, dB | dC) array
// SEG: time-segmented launch (workgroup = (batch, group, row tile, SEGMENT of cps chunks)) for calls whose row-tile grid
// leaves CUs idle.  The reference walks a row's chunks last to first inside one block (cus/selective_scan_bwd_kernel.cuh:
// 120-125,184); here a segment starts from (a) the forward state saved in x -- free, x holds one every 256 steps -- and
// (b) the reverse carry dh entering from the later segments, folded from the per-segment pairs that
// oss_scan_bwd_carry_kernel (below) leaves in sg.carry.  Per-(batch, row) partials of dA / dD / dbias get one slot per
// segment (summed by the finishing kernel in segment order).
struct BwdSeg {
    float *carry;   // [batch][dim][n_cseg][dstate][2]: (prod of a_{t+1} over the carry segment, dh at its first step from a zero carry)
    int n_seg, cps; // segments per row, 512-step chunks per segment
    // (round 5) the carry pass has its own, finer segmentation: csub carry pieces of ccps = ceil(cps / csub) chunks per main segment
    // (the last one shorter), n_cseg = n_seg * csub slots per row.  A 2-segment launch at u:(4,384,4096) ran its carry pass on 128 workgroups x 4 chunks (half the
    // CUs idle, 33 us next to the main kernel's 101); with csub = 2 it is 256 workgroups x 2 chunks, and the main kernel folds
    // two pairs instead of one.  The C rows are staged per chunk, so finer TIME segments re-stage nothing (finer ROW tiles did).
    int csub, ccps, n_cseg;
};
// HS: the forward pass left the state entering every 8-step block in f.hs (include/vmambair_oss.h: lane states).  A lane's
// 8 steps then start from a LOADED state: the local forward recurrence, the product of a over the lane and one of the two lane
// scans per state drop out of the state pass (28 of its 165 vector instructions); the loads go global -> LDS directly
// (global_load_lds_dword: no registers) one staging batch ahead, into a wave-private region next to the B / C tiles.
template <typename T, int WAVES, int NBB, int MINW, bool FD, bool SEG = false, bool HS = false, bool PB = false>
__global__ void __launch_bounds__(WAVES * 64, MINW)
oss_scan_bwd2_kernel(const oss_scan_bwd_params p, const BwdWs ws, const BwdSeg sg) {
    static_assert(!PB || (kPartialsBf16Ok<T, FD> && !HS), "bf16 partial rows: bf16 I/O, not the fused-delta / lane-state forms");
    static_assert(!(FD && SEG), "the fused-delta form is not segmented");
    static_assert(!(FD && HS), "the fused-delta form recomputes the forward states");
    constexpr int LPR = 64, I = 8;
    constexpr int ROWS = WAVES;
    constexpr int TC = LPR * I;
    constexpr int NT = WAVES * 64;
    // slab sums: 12-wave workgroups spread one state's sum over 8 waves (half waves take half of the rows each); smaller
    // workgroups use 4 waves with whole-row sums (all 8 waves summing every state measured slower: 0.185 vs 0.172 ms)
    constexpr bool SPLIT = WAVES > 8;
    constexpr int NTASK = SPLIT ? 8 : 4;         // (dB | dC) x (quarters | halves) of the chunk
    constexpr int RW = WAVES < NTASK ? WAVES : NTASK;
    constexpr int HR = SPLIT ? ROWS / 2 : ROWS;  // rows summed by one lane
    static_assert(ROWS % 2 == 0, "");
    constexpr int Q = TC / 4;                    // 4-position groups per tile row
    constexpr int QPT = (NBB * Q + NT - 1) / NT; // groups of one tile batch per thread
    static_assert(TC % kScanChunk == 0 && NBB % 2 == 0 && TC == 512, "");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sT = smem;                          // [2 buffers][B | C][NBB][TC]  tile_off layout
    // the fused-delta form also parks its ddelta rows in the slab, in time order.  (r5) So does the lane-state form of the 12-wave
    // workgroup: 32 KB of tiles + 110.6 KB of SlabQ slabs + 24.6 KB of lane states is 167 KB, over the 160 KB of a workgroup -- since
    // round 4's SlabQ image every `hs` call of variant 10 returned OSS_ERR_SHAPE (an opt-in build feature whose tests skip on the
    // shipped library: nobody saw it).  The time-order slab (98.3 KB) fits.
    constexpr bool SQ = kV2SlabQ && !FD && !(HS && WAVES > 8);
    constexpr int SA = SQ ? kSlabA : TC;       // floats per (row, dB | dC) array of the slab
    float *slab = smem + 2 * 2 * NBB * TC;     // [2][ROWS][2][SA]  per-row dB / dC terms of one state; two buffers
    float *sW = slab + 2 * ROWS * 2 * SA;      // FD: [ROWS][kMaxDtRank] dt weights of the workgroup's rows (zero-padded)
    float *sH = sW;                            // HS: [2 buffers][WAVES][NBB][64] lane states of this wave's row (never with FD)

    const oss_scan_fwd_params &f = p.f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the slab-sum rotation stays in scalar code
    const int pos = lane;
    const int wrow = wave;
    const bool seg_first = (pos == 0), seg_last = (pos == LPR - 1);

    const int L = f.seqlen, N = f.dstate, G = f.n_groups;
    const int rows_per_group = f.dim / G;
    const int tiles_per_group = ws.tiles;
    // workgroup -> (batch, group, row tile).  The row tiles of one (batch, group) read the same B / C rows; consecutive
    // workgroup ids go round-robin over the 8 XCDs (each with its own L2), so when the (batch, group) count divides by 8 the
    // tiles of a group are given ids with the same residue mod 8: they then share one L2's copy of B / C instead of fetching
    // it eight times (speed only -- nothing depends on where a workgroup runs).
    int bid = blockIdx.x, tile, bg;
    const int n_seg = SEG ? sg.n_seg : 1;
    const int total = f.batch * G * n_seg * tiles_per_group;   // (round 5: any workgroup count that divides by 8, see oss_scan_fwd.hip)
    if (total % 8 == 0) {
        const int i = (bid & 7) * (total >> 3) + (bid >> 3);
        bg = i / tiles_per_group;
        tile = i % tiles_per_group;
    } else {
        tile = bid % tiles_per_group;
        bg = bid / tiles_per_group;
    }
    int seg = 0;
    if constexpr (SEG) { seg = bg % n_seg; bg /= n_seg; }   // the tiles of one (batch, group, segment) share B / C
    const int g = bg % G;
    const int b = bg / G;
    const int row_in_group = tile * ROWS + wrow;
    const bool row_valid = row_in_group < rows_per_group;
    const int d = g * rows_per_group + (row_valid ? row_in_group : 0);
    const bool rev = g >= f.rev_group_start;
    const int d_u = f.u_row_mod > 0 ? d % f.u_row_mod : d;

    const T *u_row = reinterpret_cast<const T *>(f.u) + b * f.u_batch_stride + d_u * f.u_d_stride;
    const T *dt_row = reinterpret_cast<const T *>(f.delta) + b * f.delta_batch_stride +
                      (FD ? g * f.dt_group_stride : d * f.delta_d_stride);   // FD: rank row 0 of the (batch, group) block of z
    const float *dt_w = FD ? f.dt_weight + (size_t)d * f.dt_rank : nullptr;
    const int R = FD ? f.dt_rank : 0;
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
    // PB: partial rows as bf16 (same element layout, half the bytes)
    using PT = typename std::conditional<PB, bf16_t, float>::type;
    PT *ws_bc = reinterpret_cast<PT *>(ws.bc) + ((size_t)(b * G + g) * tiles_per_group + tile) * (2 * N + ws.rp) * L;
    const bool ws_vec = (L % 4) == 0;   // 16-byte (bf16 partials: 8-byte) stores of the partial rows

    // ---- tile staging, split in two: global -> registers (issue), registers -> LDS (commit)
    constexpr uintptr_t amask = (sizeof(T) == 4) ? 15u : 7u;
    RawQuad<T> pb[QPT], pc[QPT];
    const size_t hs_stride = lane_state_stride(L);
    const float *hs_row = HS ? f.hs + ((size_t)b * f.dim + d) * N * hs_stride + lane : nullptr;
    auto stage_issue = [&](int t0s, int n0s, int buf) {
        const int nbs = min(NBB, N - n0s);
        const bool fullchunk = (t0s + TC <= L);
        if constexpr (HS) {   // this wave's lane states of the batch: global -> LDS, complete by the time stage_commit has waited
            float *dst = sH + (size_t)((buf * WAVES + wave) * NBB) * 64;
            const float *src = hs_row + (size_t)n0s * hs_stride + (t0s >> 3);
#pragma unroll
            for (int k = 0; k < NBB; ++k)
                if (k < nbs) __builtin_amdgcn_global_load_lds(src + (size_t)k * hs_stride, dst + k * 64, 4, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            const int idx = tid + j * NT;
            const int n = idx / Q, k = idx - n * Q;
            if (idx < nbs * Q) {
                const int s0 = t0s + 4 * k;                       // first scan position of the group
                const int m0 = rev ? (L - 4 - s0) : s0;           // lowest memory index of the group
                const T *rb = gB + (int64_t)(n0s + n) * f.B_dstate_stride;
                const T *rc = gC + (int64_t)(n0s + n) * f.C_dstate_stride;
                const bool fast = fullchunk &&
                                  (((reinterpret_cast<uintptr_t>(rb + m0) | reinterpret_cast<uintptr_t>(rc + m0)) & amask) == 0);
                pb[j] = load_quad<T>(rb, m0, L, fast);
                pc[j] = load_quad<T>(rc, m0, L, fast);
            }
        }
    };
    auto stage_commit = [&](int buf, int n0s) {
        const int nbs = min(NBB, N - n0s);
        if constexpr (HS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the lane-state loads into LDS have landed
        float *dB_ = sT + (size_t)buf * 2 * NBB * TC, *dC_ = dB_ + NBB * TC;
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            const int idx = tid + j * NT;
            const int n = idx / Q, k = idx - n * Q;
            if (idx < nbs * Q) {
                const int off = tile_off<LPR, I>(n, (4 * k) / I, (4 * k) % I);
                *reinterpret_cast<f32x4 *>(dB_ + off) = quad_to_f32<T>(pb[j], rev);
                *reinterpret_cast<f32x4 *>(dC_ + off) = quad_to_f32<T>(pc[j], rev);
            }
        }
    };

    // lane n of these registers belongs to state n of this wave's row
    float A2v = 0.f, dhcv = 0.f, dAv = 0.f, hcv = 0.f;
    if (lane < N) {
        const float av = f.A[d * f.A_d_stride + lane];
        A2v = (f.a_log_form ? -__expf(av) : av) * kLog2e;
    }
    float dln_c = 0.f;  // delta of the first step of the later chunk (wave-uniform); 0 past the end
    float aev = 1.f;    // lane n: exp2(A_n * dln_c), a_{t+1} across the chunk edge (kV2EdgeLanes)

    float dD_acc = 0.f, db_acc = 0.f;
    float dWv = 0.f;   // FD: lane r = gradient of dt_weight[d, r]
    const int n_chunks = (L + TC - 1) / TC;
    const int c_begin = SEG ? seg * sg.cps : 0;
    const int c_end = SEG ? min(n_chunks, c_begin + sg.cps) : n_chunks;
    if constexpr (SEG) {
        // dh entering from the later segments: fold their pairs, last segment first
        if (lane < N) {
            const float2 *cr = reinterpret_cast<const float2 *>(sg.carry) + (((size_t)b * f.dim + d) * sg.n_cseg) * N + lane;
            float dh = 0.f;
            for (int j = sg.n_cseg - 1; j >= (seg + 1) * sg.csub; --j) {   // the carry segments behind this segment's last step
                const float2 pr = cr[(size_t)j * N];
                dh = __builtin_fmaf(pr.x, dh, pr.y);
            }
            dhcv = dh;
        }
        const int t1 = c_end * TC;   // first step of the next segment
        if (t1 < L) {
            float x = to_f32(dt_row[rev ? (L - 1 - t1) : t1]) + bias;
            if (f.delta_softplus) { float e; x = softplus_thr(x, e); }
            dln_c = x;
        }
    }
    int par = 0;   // slab buffer of the next state
    int tbuf = 0;  // tile buffer of the current batch
    int rot = 0;   // first wave of the current state's slab sum (advances by RW per state, mod WAVES)
    if constexpr (FD) {
        if (tid < ROWS * kMaxDtRank) {
            const int rr = tid / kMaxDtRank, r = tid - rr * kMaxDtRank;
            const bool ok = tile * ROWS + rr < rows_per_group && r < R;
            sW[tid] = ok ? f.dt_weight[(size_t)(g * rows_per_group + tile * ROWS + rr) * R + r] : 0.f;
        }
    }
    // tiles of the very first batch: synchronous
    stage_issue((c_end - 1) * TC, 0, 0);
    stage_commit(0, 0);
    __syncthreads();
    for (int c = c_end - 1; c >= c_begin; --c) {
        const int t0 = c * TC;
        const int tl = t0 + pos * I;
        const int valid = max(0, min(I, L - tl));
        const bool chunk_full = (t0 + TC <= L);
        float dl[I], gg[I], w[I], Q_[I], dd[I];
        // FD: softplus'(delta) of the chunk is kept from here to the per-element outputs (16-bit I/O: packed pairs in the I/O
        // type -- the materialised-delta form rounds delta itself to that type), instead of evaluating the projection twice
        uint32_t sigp[FD ? (sizeof(T) == 4 ? I : I / 2) : 1];
        {
            float uu[I];
            if constexpr (FD) {   // every load of the chunk in flight before the first use
                DtRows<T, I> zr;
                RawItems<T, I> ru, rg;
                if (dt_rows_fast_ok<I>(dt_row, f.dt_rank_stride, tl, valid, L, rev) && raw_fast_ok<I>(u_row, tl, valid, L, rev) &&
                    raw_fast_ok<I>(g_row, tl, valid, L, rev)) {
                    dt_rows_load<I>(zr, dt_row, f.dt_rank_stride, dt_w, R, tl, valid, L, rev, true);
                    ru = load_raw_fast<I>(u_row, tl, L, rev);
                    rg = load_raw_fast<I>(g_row, tl, L, rev);
                } else {
                    dt_rows_load<I>(zr, dt_row, f.dt_rank_stride, dt_w, R, tl, valid, L, rev, false);
                    ru = load_raw_slow<I>(u_row, tl, valid, L, rev);
                    rg = load_raw_slow<I>(g_row, tl, valid, L, rev);
                }
                dt_rows_apply<I>(zr, R, rev, dl);
                unpack_raw_dir<I>(ru, rev, uu);
                unpack_raw_dir<I>(rg, rev, gg);
            } else if constexpr (!kV2GroupPro) {
                load_items_dir<I>(u_row, tl, valid, L, rev, uu);
                load_items_dir<I>(dt_row, tl, valid, L, rev, dl);
                load_items_dir<I>(g_row, tl, valid, L, rev, gg);
            } else {
                // u, delta, dout of the chunk and the saved forward state as ONE group of loads behind one fast / slow decision:
                // three load_items_dir calls each end in their own s_waitcnt vmcnt(0) (a branch per call), i.e. three
                // serial round trips per chunk and wave before any arithmetic (and a fourth for the state)
                RawItems<T, I> ru, rd, rg;
                if (raw_fast_ok<I>(u_row, tl, valid, L, rev) && raw_fast_ok<I>(dt_row, tl, valid, L, rev) &&
                    raw_fast_ok<I>(g_row, tl, valid, L, rev)) {
                    ru = load_raw_fast<I>(u_row, tl, L, rev);
                    rd = load_raw_fast<I>(dt_row, tl, L, rev);
                    rg = load_raw_fast<I>(g_row, tl, L, rev);
                } else {
                    ru = load_raw_slow<I>(u_row, tl, valid, L, rev);
                    rd = load_raw_slow<I>(dt_row, tl, valid, L, rev);
                    rg = load_raw_slow<I>(g_row, tl, valid, L, rev);
                }
                if constexpr (kV2HcvEarly)
                    hcv = (t0 >= kScanChunk && lane < N) ? x_row[(size_t)(t0 / kScanChunk - 1) * 2 * N + 2 * lane + 1] : 0.f;
                unpack_raw_dir<I>(ru, rev, uu);
                unpack_raw_dir<I>(rd, rev, dl);
                unpack_raw_dir<I>(rg, rev, gg);
            }
            if (!row_valid) {  // a row slot past the end of the group must not contribute to dB/dC
#pragma unroll
                for (int i = 0; i < I; ++i) { uu[i] = 0.f; gg[i] = 0.f; }
            }
            float sg[FD ? I : 1];
#pragma unroll
            for (int i = 0; i < I; ++i) {
                float x = dl[i] + bias;
                if (f.delta_softplus) {
                    const float raw_ = x;
                    float e;
                    x = softplus_thr(raw_, e);
                    if constexpr (FD) sg[i] = (i < valid) ? ((raw_ <= 20.f) ? e * __builtin_amdgcn_rcpf(1.f + e) : 1.f) : 0.f;   // bwd_kernel.cuh:228-241
                } else if constexpr (FD) {
                    sg[i] = (i < valid) ? 1.f : 0.f;
                }
                dl[i] = (i < valid) ? x : 0.f;
                w[i] = dl[i] * uu[i];
                Q_[i] = 0.f;
                dd[i] = 0.f;
            }
            if constexpr (FD) {
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int i = 0; i < I; ++i) sigp[i] = __float_as_uint(sg[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < I / 2; ++i) sigp[i] = pack2<T>(sg[2 * i], sg[2 * i + 1]);
                }
            }
        }
        float S = 0.f;
#pragma unroll
        for (int i = 0; i < I; ++i) S += dl[i];
        if constexpr (!HS && (FD || !(kV2GroupPro && kV2HcvEarly))) {   // saved forward state entering this chunk (bwd_kernel.cuh:184); otherwise fetched with the chunk's rows
            const int xi = t0 / kScanChunk - 1;
            hcv = (xi >= 0 && lane < N) ? x_row[(size_t)xi * 2 * N + 2 * lane + 1] : 0.f;
        }
        const float dln_lane = shift_from_next_lane(dl[0], dln_c, seg_last);
        const float Sshift = S - dl[0] + dln_lane;  // sum of delta over steps tl+1 .. tl+I
        if constexpr (kV2EdgeLanes) aev = exp2_hw(dln_c * A2v);
        // lane-dependent parts of the slab-sum addresses: a half wave = 32 groups of 4 scan positions x one half of the rows
        const int sl = SPLIT ? (lane & 31) : lane;
        const float *sum_src = slab + (SQ ? (sl & 1) * kSlabK + 4 * (sl >> 1) : 4 * sl) + (SPLIT ? (lane >> 5) * (HR * 2 * SA) : 0);
        PT *sum_dst = ws_bc + (rev ? (L - 4 - t0 - 4 * sl) : (t0 + 4 * sl));

        // everything of one state (B/C tiles already in registers); leaves its dB / dC terms in the slab buffer `par`
        auto state_pass = [&](int n, float (&bt)[I], float (&ct)[I], float hin_saved) {
            const float A2 = lane_get(A2v, n), dhc = lane_get(dhcv, n);
            float a[I], hh[I];
            float hin;
            if constexpr (HS) {
                // ---- forward states: the state entering this lane's steps was saved by the forward pass
#pragma unroll
                for (int i = 0; i < I; ++i) {
                    a[i] = exp2_hw(dl[i] * A2);
                    hh[i] = bt[i] * w[i];
                }
                hin = (valid > 0) ? hin_saved : 0.f;   // entries past the end of the sequence are never written
            } else {
                // ---- forward recompute: local recurrence (hh holds b_t until the state pass)
                const float hc = lane_get(hcv, n);
                float h = 0.f;
#pragma unroll
                for (int i = 0; i < I; ++i) {
                    a[i] = exp2_hw(dl[i] * A2);
                    hh[i] = bt[i] * w[i];
                    h = (i == 0) ? hh[0] : __builtin_fmaf(a[i], h, hh[i]);
                }
                float P = exp2_hw(S * A2);
                segment_scan<LPR>(P, h);
                const float hfull = __builtin_fmaf(P, hc, h);
                hin = shift_from_prev_lane(hfull, hc, seg_first);
            }
            {
                float hp = hin;
#pragma unroll
                for (int i = 0; i < I; ++i) {
                    hp = __builtin_fmaf(a[i], hp, hh[i]);
                    hh[i] = hp;
                }
            }
            // ---- reverse recurrence: element (a_{t+1}, C_t g_t)   (bwd_kernel.cuh:170-193)
            const float a_edge = kV2EdgeLanes ? lane_get(aev, n) : exp2_hw(dln_c * A2);
            const float a_nl = shift_from_next_lane(a[0], a_edge, seg_last);
            float dloc = 0.f;
#pragma unroll
            for (int i = I - 1; i >= 0; --i) {
                const float an = (i == I - 1) ? a_nl : a[(i + 1) % I];
                ct[i] *= gg[i];   // C_t g_t, used twice
                dloc = (i == I - 1) ? ct[i] : __builtin_fmaf(an, dloc, ct[i]);
            }
            float Pm = segment_mirror<LPR>(exp2_hw(Sshift * A2), lane);
            float dm = segment_mirror<LPR>(dloc, lane);
            segment_scan<LPR>(Pm, dm);
            const float dfull_m = __builtin_fmaf(Pm, dhc, dm);       // dh at the first step of mirrored lane
            const float dex_m = shift_from_prev_lane(dfull_m, dhc, seg_first);
            float dh = segment_mirror<LPR>(dex_m, lane);              // dh entering this lane from the right
            dhcv = lane_set(dhcv, lane, n, lane_get(dfull_m, 63));    // mirrored-last lane = first lane in time
            // ---- reverse pass with gradients (bwd_kernel.cuh:196-206); p_t = a_t h_{t-1}
            float *sb = slab + ((par * ROWS + wrow) * 2) * SA + (SQ ? pos * 4 : pos * I);
            float dA_acc = 0.f;
#pragma unroll
            for (int k = I / 4 - 1; k >= 0; --k) {
                float vB[4], vC[4];
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const int i = 4 * k + j;
                    const float an = (i == I - 1) ? a_nl : a[(i + 1) % I];
                    dh = __builtin_fmaf(an, dh, ct[i]);
                    Q_[i] = __builtin_fmaf(dh, bt[i], Q_[i]);
                    const float hprev = (i == 0) ? hin : hh[(i + I - 1) % I];
                    const float r = dh * (a[i] * hprev);
                    dd[i] = __builtin_fmaf(A2, r, dd[i]);
                    dA_acc = __builtin_fmaf(dl[i], r, dA_acc);
                    // rows past the end of the group carry u = dout = 0, so they contribute exact zeros
                    vB[j] = dh * w[i];
                    vC[j] = gg[i] * hh[i];
                }
#ifndef OSS_EXP_V2_NOSLAB   // (OSS_EXP_*: timing experiments only, tools/build_experiment.sh -- results are wrong)
                *reinterpret_cast<f32x4 *>(sb + (SQ ? kSlabK : 4) * k) = f32x4{vB[0], vB[1], vB[2], vB[3]};
                *reinterpret_cast<f32x4 *>(sb + SA + (SQ ? kSlabK : 4) * k) = f32x4{vC[0], vC[1], vC[2], vC[3]};
#else
                if (vB[0] + vC[3] == 12345.678f) sb[0] = vB[1] + vC[2];
#endif
            }
            const float dA_sum = segment_sum_to_last<LPR>(dA_acc) + lane_get(dAv, n);
            dAv = lane_set(dAv, lane, n, lane_get(dA_sum, 63));
        };
        // sum the slabs of state n over the workgroup's rows (fixed order) and write the workgroup's partial.  Eight tasks
        // (dB | dC) x (128-position quarter of the chunk), one per wave rot .. rot+7 (wrapping); inside a task the two half
        // waves sum the lower / upper half of the rows for the same 32 position groups (16-byte reads, conflict-free) and
        // are combined with v_permlane32_swap -- the latency of this sum sits on the critical path of every state (all
        // waves wait for the summing ones at the next barrier), so it is spread as thin as the lanes allow.
        auto slab_sum = [&](int prow0, int prow1, int buf) {   // partial rows of the slab's first / second array
#ifndef OSS_EXP_V2_NOSUM
            int rw = wave - rot;
            rw += (rw < 0) ? WAVES : 0;
            if (rw < RW) {
                for (int task = rw; task < NTASK; task += RW) {
                    const int arr = task / (NTASK / 2), off = (task % (NTASK / 2)) * (TC / (NTASK / 2));
                    // SlabQ: the task's positions start at off / 8, two quads per position -> off / 2 floats into the image
                    const float *src = sum_src + (size_t)buf * ROWS * 2 * SA + arr * SA + (SQ ? off / 2 : off);
                    f32x4 acc = *reinterpret_cast<const f32x4 *>(src);
#pragma unroll
                    for (int r = 1; r < HR; ++r) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + r * 2 * SA);
                        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                    }
                    if constexpr (SPLIT) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {   // lower-rows sum + upper-rows sum, in every lane
                            const int bits = __float_as_int(acc[k]);
                            const auto sw = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
                            acc[k] = __int_as_float(sw[0]) + __int_as_float(sw[1]);
                        }
                    }
                    PT *dst = sum_dst + (size_t)(arr ? prow1 : prow0) * L + (rev ? -off : off);
                    if (!SPLIT || lane < 32) {
                        if (chunk_full && ws_vec) {
                            if constexpr (PB) {
                                *reinterpret_cast<u32x2 *>(dst) = rev ? u32x2{pack2<bf16_t>(acc.w, acc.z), pack2<bf16_t>(acc.y, acc.x)}
                                                                      : u32x2{pack2<bf16_t>(acc.x, acc.y), pack2<bf16_t>(acc.z, acc.w)};
                            } else {
                                *reinterpret_cast<f32x4 *>(dst) = rev ? f32x4{acc.w, acc.z, acc.y, acc.x} : acc;
                            }
                        } else {
                            const int t = t0 + off + 4 * sl;   // scan position of acc.x; mirrored groups store at L-1-t
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (t + j < L) {
                                    if constexpr (PB) dst[rev ? (3 - j) : j] = from_f32<bf16_t>(acc[j]);
                                    else dst[rev ? (3 - j) : j] = acc[j];
                                }
                        }
                    }
                }
            }
#endif
            rot += RW;
            rot -= (rot >= WAVES) ? WAVES : 0;
        };

        for (int n0 = 0; n0 < N; n0 += NBB) {
            const int nb = min(NBB, N - n0);
            // the tiles this batch needs are in buffer tbuf (committed and fenced by the previous batch's last barrier);
            // fetch the next batch's -- of this chunk, or the first of the next chunk -- from global memory now
            const bool more_here = n0 + NBB < N;
            const bool have_next = more_here || c > c_begin;
            const int nt0 = more_here ? t0 : t0 - TC, nn0 = more_here ? n0 + NBB : 0;
            if (have_next) stage_issue(nt0, nn0, tbuf ^ 1);
            const float *tb = sT + (size_t)tbuf * 2 * NBB * TC + pos * 4, *tc = tb + NBB * TC;
            float b0[I], c0[I], b1[I], c1[I];
            float h0s = 0.f, h1s = 0.f;   // HS: the saved state entering this lane's steps, per state of the register double set
            const float *th = HS ? sH + (size_t)((tbuf * WAVES + wave) * NBB) * 64 + lane : nullptr;
            read_tile<I>(tb, b0);
            read_tile<I>(tc, c0);
            if constexpr (HS) h0s = th[0];
            // every LDS read issued so far has landed before the loop starts: lgkmcnt counts in order, so without this the
            // compiler must assume, at the top of every iteration, that the tiles of the CURRENT state may still be in
            // flight behind the prefetch of the next one -- and waits for the prefetch (checked in the ISA)
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
            for (int nn = 0; nn < nb; nn += 2) {
                const bool has1 = nn + 1 < nb;
                if (has1) {
                    read_tile<I>(tb + (nn + 1) * TC, b1); read_tile<I>(tc + (nn + 1) * TC, c1);
                    if constexpr (HS) h1s = th[(nn + 1) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                state_pass(n0 + nn, b0, c0, h0s);
                if (!has1 && have_next) stage_commit(tbuf ^ 1, nn0);   // before the batch's last barrier, which then fences it
#ifndef OSS_EXP_V2_NOBAR
                __syncthreads();                       // state n's slabs are complete; buffer par^1 is free again
#endif
                slab_sum(n0 + nn, N + n0 + nn, par);
                par ^= 1;
                if (has1) {
                    const bool last = nn + 2 >= nb;
                    if (!last) {
                        read_tile<I>(tb + (nn + 2) * TC, b0); read_tile<I>(tc + (nn + 2) * TC, c0);
                        if constexpr (HS) h0s = th[(nn + 2) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    state_pass(n0 + nn + 1, b1, c1, h1s);
                    if (last && have_next) stage_commit(tbuf ^ 1, nn0);
#ifndef OSS_EXP_V2_NOBAR
                    __syncthreads();
#endif
                    slab_sum(n0 + nn + 1, N + n0 + nn + 1, par);
                    par ^= 1;
                }
            }
            tbuf ^= 1;
        }
        // ---- per-element outputs (bwd_kernel.cuh:151,200-203,228-245); u (and softplus' unless FD) re-derived here
        {
            float uu[I], sg[I], du[I], dv[I];
            DtRows<T, FD ? I : 4> zr;   // FD: the rank rows again (for the dt_weight gradient), in flight together with u
            if constexpr (FD) {
                RawItems<T, I> ru;
                if (dt_rows_fast_ok<I>(dt_row, f.dt_rank_stride, tl, valid, L, rev) && raw_fast_ok<I>(u_row, tl, valid, L, rev)) {
                    dt_rows_load<I>(zr, dt_row, f.dt_rank_stride, dt_w, R, tl, valid, L, rev, true);
                    ru = load_raw_fast<I>(u_row, tl, L, rev);
                } else {
                    dt_rows_load<I>(zr, dt_row, f.dt_rank_stride, dt_w, R, tl, valid, L, rev, false);
                    ru = load_raw_slow<I>(u_row, tl, valid, L, rev);
                }
                unpack_raw_dir<I>(ru, rev, uu);
            }
            if constexpr (FD) {
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int i = 0; i < I; ++i) sg[i] = __uint_as_float(sigp[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < I / 2; ++i) unpack2<T>(sigp[i], sg[2 * i], sg[2 * i + 1]);
                }
            } else {
                float raw[I];
                if constexpr (!kV2GroupEpi) {
                    load_items_dir<I>(u_row, tl, valid, L, rev, uu);
                    load_items_dir<I>(dt_row, tl, valid, L, rev, raw);
                } else {   // u and delta again (not kept across the state loop: registers), as one group of loads
                    RawItems<T, I> ru, rd;
                    if (raw_fast_ok<I>(u_row, tl, valid, L, rev) && raw_fast_ok<I>(dt_row, tl, valid, L, rev)) {
                        ru = load_raw_fast<I>(u_row, tl, L, rev);
                        rd = load_raw_fast<I>(dt_row, tl, L, rev);
                    } else {
                        ru = load_raw_slow<I>(u_row, tl, valid, L, rev);
                        rd = load_raw_slow<I>(dt_row, tl, valid, L, rev);
                    }
                    unpack_raw_dir<I>(ru, rev, uu);
                    unpack_raw_dir<I>(rd, rev, raw);
                }
#pragma unroll
                for (int i = 0; i < I; ++i) {
                    float s = 1.f;
                    if (f.delta_softplus) {
                        const float r_ = raw[i] + bias;
                        const float e = exp2_hw(r_ * kLog2e);
                        // d softplus = sigmoid(raw) for raw <= 20, 1 above (bwd_kernel.cuh:228-241)
                        s = (r_ <= 20.f) ? e * __builtin_amdgcn_rcpf(1.f + e) : 1.f;
                    }
                    sg[i] = (i < valid) ? s : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < I; ++i) {
                const float ui = row_valid ? uu[i] : 0.f;
                du[i] = __builtin_fmaf(Q_[i], dl[i], Dd * gg[i]);
                const float ddel = __builtin_fmaf(Q_[i], ui, dd[i] * kLn2);
                dv[i] = ddel * sg[i];
                dD_acc = __builtin_fmaf(gg[i], ui, dD_acc);
                db_acc += dv[i];
            }
            if (row_valid) {
                store_items_dir<I>(du_row, tl, valid, L, rev, du);
                if constexpr (!FD) store_items_dir<I>(dd_row, tl, valid, L, rev, dv);
            }
            if constexpr (FD) {
                // backward of the archs' dt einsum (MambaSISR6_arch.py:411) on the chunk:
                //   dw[d, r]  = sum over t of ddelta[d, t] * z[r, t]                  per row: lane r of dWv
                //   dz[r, t]  = sum over the group's rows of w[d, r] * ddelta[d, t]   ONE slab round: every wave leaves its row of
                //               ddelta in the slab, eight waves form the rank-R combinations for one scan position per lane
                if (!row_valid) {
#pragma unroll
                    for (int i = 0; i < I; ++i) dv[i] = 0.f;
                }
#pragma unroll
                for (int r = 0; r < kMaxDtRank; ++r) {
                    if (r < R) {
                        float zz[I], acc = 0.f;
                        unpack_raw_dir<I>(zr.z[r], rev, zz);
#pragma unroll
                        for (int i = 0; i < I; ++i) acc = __builtin_fmaf(dv[i], zz[i], acc);
                        const float tot = segment_sum_to_last<LPR>(acc) + lane_get(dWv, r);
                        dWv = lane_set(dWv, lane, r, lane_get(tot, 63));
                    }
                }
                float *sb = slab + ((par * ROWS + wrow) * 2) * TC + pos * I;
#pragma unroll
                for (int k = 0; k < I / 4; ++k)
                    *reinterpret_cast<f32x4 *>(sb + 4 * k) = f32x4{dv[4 * k], dv[4 * k + 1], dv[4 * k + 2], dv[4 * k + 3]};
                __syncthreads();
                {   // waves rot .. rot+7: lane = one scan position; acc[r] = sum over rows of w[row, r] * ddelta[row, t]
                    constexpr int RZ = WAVES < 8 ? WAVES : 8;
                    int rw = wave - rot;
                    rw += (rw < 0) ? WAVES : 0;
                    if (rw < RZ) {
                        for (int e = rw * 64 + lane; e < TC; e += RZ * 64) {
                            const float *src = slab + (size_t)par * ROWS * 2 * TC + e;
                            float acc[kMaxDtRank];
#pragma unroll
                            for (int r = 0; r < kMaxDtRank; ++r) acc[r] = 0.f;
#pragma unroll
                            for (int rr = 0; rr < ROWS; ++rr) {
                                const float v = src[rr * 2 * TC];
                                const f32x4 w0 = *reinterpret_cast<const f32x4 *>(sW + rr * kMaxDtRank);       // LDS broadcast reads
                                const f32x4 w1 = *reinterpret_cast<const f32x4 *>(sW + rr * kMaxDtRank + 4);   // (zeros past the rank)
                                acc[0] = __builtin_fmaf(w0.x, v, acc[0]); acc[1] = __builtin_fmaf(w0.y, v, acc[1]);
                                acc[2] = __builtin_fmaf(w0.z, v, acc[2]); acc[3] = __builtin_fmaf(w0.w, v, acc[3]);
                                if (R > 4) {
                                    acc[4] = __builtin_fmaf(w1.x, v, acc[4]); acc[5] = __builtin_fmaf(w1.y, v, acc[5]);
                                    acc[6] = __builtin_fmaf(w1.z, v, acc[6]); acc[7] = __builtin_fmaf(w1.w, v, acc[7]);
                                }
                            }
                            const int t = t0 + e;
                            if (t < L) {
                                float *dst = ws_bc + (size_t)(2 * N) * L + (rev ? (L - 1 - t) : t);
#pragma unroll
                                for (int r = 0; r < kMaxDtRank; ++r)
                                    if (r < R) dst[(size_t)r * L] = acc[r];
                            }
                        }
                    }
                    rot += RZ;
                    rot -= (rot >= WAVES) ? WAVES : 0;
                }
                par ^= 1;
            }
        }
        dln_c = lane_get(dl[0], 0);
    }
    // ---- per-row partials over the sequence
    const float dD_sum = segment_sum_to_last<LPR>(dD_acc);
    const float db_sum = segment_sum_to_last<LPR>(db_acc);
    const size_t slot = (size_t)(b * n_seg + seg) * f.dim + d;   // one partial per (batch, segment, row)
    if (row_valid) {
        if (seg_last) {
            if (ws.dD) ws.dD[slot] = dD_sum;
            if (ws.db) ws.db[slot] = db_sum;
        }
        if (lane < N) ws.dA[slot * N + lane] = dAv;
    }
    if constexpr (FD) {
        if (row_valid && lane < R) ws.dW[slot * kMaxDtRank + lane] = dWv;
    }
}

// Reverse-carry pass of the time-segmented backward (launched before oss_scan_bwd2_kernel<..., SEG = true>): for every
// segment s >= 1 of every row the pair
//     ( prod_{t0 < t <= t1} a_t ,  dh_{t0} evaluated with dh_{t1} = 0 )
// of the reverse recurrence dh_t = C_t g_t + a_{t+1} dh_{t+1} (cus/selective_scan_bwd_kernel.cuh:170-193) over the
// segment's steps [t0, t1).  Only the recurrence itself: delta (+ softplus), dout, C -- 4 vector instructions and one
// v_exp_f32 per (element, state) against the main kernel's ~28.  Same lane / chunk decomposition as the main kernel
// (row = one wave, 64 lanes x 8 steps).
// (round 4) The pass was 0.36 of a main pass in time for a fifth of its instructions: per chunk it loaded delta / dout, waited,
// staged the C tile synchronously between two barriers, and updated the running product of a with two v_readlane + a
// select per state.  Now (as in the main kernel) the NEXT chunk's delta / dout / C-tile loads are issued before the current
// chunk's state loop and committed to a second LDS tile buffer after it (one barrier per tile batch), and the product of a
// over a chunk is exp2(A_n * (sum of delta over the chunk, shifted by one step)) -- ONE v_exp_f32 per chunk with lane n =
// state n, after one wave reduction of the per-lane sums, instead of a chain through every state's scan.
template <typename T, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
oss_scan_bwd_carry_kernel(const oss_scan_bwd_params p, const BwdSeg sg, int tiles_per_group) {
    constexpr int LPR = 64, I = 8, TC = LPR * I, NT = WAVES * 64, ROWS = WAVES;
    constexpr int Q = TC / 4;                          // 4-position groups per tile row
    constexpr int QPT = (kNB * Q + NT - 1) / NT;       // groups of one tile batch per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2 buffers][kNB][TC] C tiles, tile_off image
    const oss_scan_fwd_params &f = p.f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool seg_last = (lane == LPR - 1);
    const int L = f.seqlen, N = f.dstate, G = f.n_groups;
    const int rows_per_group = f.dim / G;
    int bid = blockIdx.x;
    const int tile = bid % tiles_per_group; bid /= tiles_per_group;
    const int n_cs = sg.n_cseg - sg.csub;   // the carry segments of main segment 0 have no predecessor to hand a carry to
    const int seg = sg.csub + bid % n_cs; bid /= n_cs;
    const int g = bid % G;
    const int b = bid / G;
    const int row_in_group = tile * ROWS + wave;
    const bool row_valid = row_in_group < rows_per_group;
    const int d = g * rows_per_group + (row_valid ? row_in_group : 0);
    const bool rev = g >= f.rev_group_start;
    const T *dt_row = reinterpret_cast<const T *>(f.delta) + b * f.delta_batch_stride + d * f.delta_d_stride;
    const int d_g = p.dout_row_mod > 0 ? d % p.dout_row_mod : d;
    const T *g_row = reinterpret_cast<const T *>(p.dout) + b * p.dout_batch_stride + d_g * p.dout_d_stride;
    const T *gC = reinterpret_cast<const T *>(f.C) + b * f.C_batch_stride + g * f.C_group_stride;
    const float bias = f.delta_bias ? f.delta_bias[d] : 0.f;

    float A2v = 0.f, dhcv = 0.f, Pv = 1.f;   // lane n = state n (dstate <= 64)
    if (lane < N) {
        const float av = f.A[d * f.A_d_stride + lane];
        A2v = (f.a_log_form ? -__expf(av) : av) * kLog2e;
    }
    const int n_chunks = (L + TC - 1) / TC;
    // carry piece `seg` = piece j of main segment s: chunks s cps + j ccps .., clipped to the main segment and to the sequence (a short
    // last main segment may leave its last pieces empty: they keep the identity pair (1, 0))
    const int ms_ = seg / sg.csub, mj_ = seg - ms_ * sg.csub;
    const int c_begin = ms_ * sg.cps + mj_ * sg.ccps, c_end = min(min(c_begin + sg.ccps, (ms_ + 1) * sg.cps), n_chunks);
    float dln_c = 0.f;
    {
        const int t1 = c_end * TC;
        if (t1 < L) {
            float x = to_f32(dt_row[rev ? (L - 1 - t1) : t1]) + bias;
            if (f.delta_softplus) { float e; x = softplus_thr(x, e); }
            dln_c = x;
        }
    }
    // ---- staging: global -> registers (issue), registers -> LDS (commit)
    constexpr uintptr_t amask = (sizeof(T) == 4) ? 15u : 7u;
    RawQuad<T> pc[QPT];
    auto stage_issue = [&](int t0s, int n0s) {
        const int nbs = min(kNB, N - n0s);
        const bool fullchunk = (t0s + TC <= L);
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            const int idx = tid + j * NT;
            const int n = idx / Q, k = idx - n * Q;
            if (idx < nbs * Q) {
                const int s0 = t0s + 4 * k;
                const int m0 = rev ? (L - 4 - s0) : s0;
                const T *rc = gC + (int64_t)(n0s + n) * f.C_dstate_stride;
                const bool fast = fullchunk && ((reinterpret_cast<uintptr_t>(rc + m0) & amask) == 0);
                pc[j] = load_quad<T>(rc, m0, L, fast);
            }
        }
    };
    auto stage_commit = [&](int buf, int n0s) {
        const int nbs = min(kNB, N - n0s);
        float *dC_ = smem + (size_t)buf * kNB * TC;
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            const int idx = tid + j * NT;
            const int n = idx / Q, k = idx - n * Q;
            if (idx < nbs * Q) *reinterpret_cast<f32x4 *>(dC_ + tile_off<LPR, I>(n, (4 * k) / I, (4 * k) % I)) = quad_to_f32<T>(pc[j], rev);
        }
    };
    // delta / dout of a chunk as raw items: issued one chunk ahead
    RawItems<T, I> rd, rg;
    auto rows_issue = [&](int t0s) {
        const int tl = t0s + lane * I;
        const int valid = max(0, min(I, L - tl));
        if (raw_fast_ok<I>(dt_row, tl, valid, L, rev) && raw_fast_ok<I>(g_row, tl, valid, L, rev)) {
            rd = load_raw_fast<I>(dt_row, tl, L, rev);
            rg = load_raw_fast<I>(g_row, tl, L, rev);
        } else {
            rd = load_raw_slow<I>(dt_row, tl, valid, L, rev);
            rg = load_raw_slow<I>(g_row, tl, valid, L, rev);
        }
    };
    int tbuf = 0;
    rows_issue((c_end - 1) * TC);
    stage_issue((c_end - 1) * TC, 0);
    stage_commit(0, 0);
    __syncthreads();
    for (int c = c_end - 1; c >= c_begin; --c) {
        const int t0 = c * TC;
        const int tl = t0 + lane * I;
        const int valid = max(0, min(I, L - tl));
        float dl[I], gg[I];
        unpack_raw_dir<I>(rd, rev, dl);
        unpack_raw_dir<I>(rg, rev, gg);
        if (c > c_begin) rows_issue(t0 - TC);   // the next (earlier) chunk's rows: in flight during this chunk's state loops
        float S = 0.f;
#pragma unroll
        for (int i = 0; i < I; ++i) {
            float x = dl[i] + bias;
            if (f.delta_softplus) { float e; x = softplus_thr(x, e); }
            dl[i] = (i < valid) ? x : 0.f;
            S += dl[i];
        }
        const float dln_lane = shift_from_next_lane(dl[0], dln_c, seg_last);
        const float Sshift = S - dl[0] + dln_lane;   // sum of delta over steps tl+1 .. tl+I
        // product of a_{t+1} over the chunk = exp2(A * sum over the wave of Sshift): one reduction + one v_exp_f32 per chunk
        const float Stot = lane_get(segment_sum_to_last<LPR>(Sshift), 63);
        Pv *= exp2_hw(Stot * A2v);
        for (int n0 = 0; n0 < N; n0 += kNB) {
            const int nb = min(kNB, N - n0);
            const bool more_here = n0 + kNB < N;
            const bool have_next = more_here || c > c_begin;
            if (have_next) stage_issue(more_here ? t0 : t0 - TC, more_here ? n0 + kNB : 0);
            const float *tc = smem + (size_t)tbuf * kNB * TC + lane * 4;
            for (int nn = 0; nn < nb; ++nn) {
                const int n = n0 + nn;
                const float A2 = lane_get(A2v, n), dhc = lane_get(dhcv, n);
                float ct[I];
                read_tile<I>(tc + nn * TC, ct);
                float a[I];
#pragma unroll
                for (int i = 0; i < I; ++i) a[i] = exp2_hw(dl[i] * A2);
                const float a_nl = shift_from_next_lane(a[0], exp2_hw(dln_c * A2), seg_last);
                float dloc = 0.f;
#pragma unroll
                for (int i = I - 1; i >= 0; --i) {
                    const float an = (i == I - 1) ? a_nl : a[(i + 1) % I];
                    const float cg = ct[i] * gg[i];
                    dloc = (i == I - 1) ? cg : __builtin_fmaf(an, dloc, cg);
                }
                float Pm = segment_mirror<LPR>(exp2_hw(Sshift * A2), lane);
                float dm = segment_mirror<LPR>(dloc, lane);
                segment_scan<LPR>(Pm, dm);
                const float dfull_m = __builtin_fmaf(Pm, dhc, dm);   // mirrored-last lane: dh at the chunk's first step
                dhcv = lane_set(dhcv, lane, n, lane_get(dfull_m, 63));
            }
            if (have_next) stage_commit(tbuf ^ 1, more_here ? n0 + kNB : 0);
            __syncthreads();   // the next batch's tile is complete, this one has been read by everyone
            tbuf ^= 1;
        }
        dln_c = lane_get(dl[0], 0);
    }
    if (row_valid && lane < N) {
        float2 *cr = reinterpret_cast<float2 *>(sg.carry) + (((size_t)b * f.dim + d) * sg.n_cseg + seg) * N + lane;
        *cr = make_float2(Pv, dhcv);
    }
}

}  // namespace oss
