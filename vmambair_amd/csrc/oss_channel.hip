// oss_channel.hip -- the channel branch of SS2D_1 (the two channel-direction scans of the Omni Selective Scan,
// SRGAN/VmambaIR/archs/MambaSISR6_arch.py:438-483,495-496; RealSR rank-R form MambaRealSR11_arch.py:758-817):
//   pooled[b, l] = mean over pixels of y2[b, l]                (l runs over the d_inner channels: L = d_inner)
//   seq[i, l]    = cin_w[i] pooled[l] + cin_b[i]               (the 1 -> dc_inner lift; identity for RealSR)
//   z[k, c, l]   = sum_i xc_proj[k, c, i] seq[i, l];  dts[k, i, l] = sum_r dtc_w[k, i, r] z[k, r, l]
//   y[k, i, :]   = selective scan of row (k, i) over l (k = 1: from l = L-1 down to 0), A = -exp(Ac_logs), D = Dsc
//   yc[l]        = sum_i cout_w[i] (y[0, i, l] + y[1, i, l]) + cout_b;   c = LayerNorm_l(yc)
//   out          = y2 * c + y2   ("mul_add")   or   y2 + c   ("add")
// Everything between the pooling and the gate is a few thousand numbers per image.  The reference (and any
// op-by-op port) spends ~15 launches forward and ~20 backward on it; here: one workgroup per image does the
// whole thing in one launch per direction of autograd, plus row reductions / row-affine passes over the
// (B, d, H, W) tensor at both ends.  fp32 throughout (the RealSR tree forces fp32 here even under AMP).
//
// Scan layout inside the workgroup: lane = (row i of the direction) * 16 + state n; sums over the 16 states are DPP row
// reductions, sums over the rows of a direction cross the 16-lane rows with ds_bpermute.  A direction's L steps are cut
// into S = NT / 128 time segments, one WAVE per (direction, segment): a single wave on a SIMD issues one vector
// instruction per ~5.6 cycles (profiles/r02_ubench_issue_rates.txt), which -- not the dependent FMA -- is what the
// one-wave-per-direction scan of round 1 ran at (~290 cycles per step).  Every segment first runs the bare recurrence
// from a zero state (a, B u and one FMA per step) to get its (product of a, end state) pair, the pairs are chained
// through LDS (the scan monoid of oss_device.h, S - 1 FMAs), then every segment runs the full step (outputs, saved
// states / gradients) from its true incoming state.  The states h[row, l, n] of every step are kept in HBM
// for the backward recurrence (it needs h_{t-1}; 8 x 768 x 16 floats per image at most).
#include "oss_device.h"
#include "oss_host.h"

namespace oss {

constexpr int kChN = 16;  // dc_state of every reference config
constexpr int kChU = 8;   // scan steps whose operands are fetched together
constexpr int kChNT = 512;  // threads per workgroup: 8 waves = 2 directions x 4 time segments
constexpr int kChRed = 8;   // floats of the block-sum scratch

template <int NT>
__device__ __forceinline__ float block_sum(float v, float *red /*[kChRed]*/, int tid) {
    static_assert(NT == 256 || NT == 512, "block_sum: 4 or 8 waves");
    const float w = segment_sum_to_last<64>(v);
    __syncthreads();  // red may still be read from a previous call
    if ((tid & 63) == 63) red[tid >> 6] = w;
    __syncthreads();
    float t = (red[0] + red[1]) + (red[2] + red[3]);
    if constexpr (NT == 512) t += (red[4] + red[5]) + (red[6] + red[7]);
    return t;
}

// sum over the rows i of a direction (lanes n, 16 + n, 32 + n, 48 + n); every lane gets the total
__device__ __forceinline__ float sum_over_rows(float v, int lane) {
    v += __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, __float_as_int(v)));
    v += __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(v)));
    return v;
}

// use_lds: the projections z and dts also live in LDS for this kernel's own reads (they are written to HBM for the
// backward either way); without it every phase pays a round trip through L2
// (a template parameter, not a runtime pointer select: generic-address loads would tie the LDS and vector-memory
// wait counters together inside the serial scan loop)
// (round 4, measured and reverted: every parameter of the phases below requested in one batch at the top of the kernel, with clamped
// indices and stand-in pointers instead of `cond ? p[i] : 0` -- most of those loads are wave-uniform and were scalar loads already;
// 16.7 us per launch against 15.4 as it stands, profiles/r04_serial_loads.txt)
template <bool use_lds, int NT>
__global__ void __launch_bounds__(NT)
oss_chan_fwd_kernel(oss_chan_params p) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, tid = threadIdx.x, L = p.L, dc = p.dc, Cc = p.Cc, Rc = p.Rc;
#ifdef OSS_EXP_CHAN_TIMING  // (timing experiments only: tools/build_experiment.sh) phase stamps in shader cycles -> zt[0..7]
    long long ts_[8]; int ns_ = 0;
#define OSS_STAMP() do { ts_[ns_++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define OSS_STAMP() do { } while (0)
#endif
    OSS_STAMP();
    const bool lift = p.cin_w != nullptr;
    float *seq = sm;                 // [dc][L]
    float *ybuf = seq + dc * L;      // [2 dc][L]
    float *ycs = ybuf + 2 * dc * L;  // [L]
    float *red = ycs + L;            // [kChRed]
    float *ypart = red + kChRed;     // [4][2 dc][L]: the state groups' partial sums of y
    // the pooled descriptor from the LayerNorm forward's per-workgroup output sums.  Few tiles (the training patches: 8 ... 32): one thread
    // per channel adds them in order.  Many tiles (an untiled 512 x 512 RealSR plane leaves 1024): that was 96 of the 512 threads busy with
    // 8 loads in flight each, 128 dependent rounds = most of a 61 us launch -- then NS threads per channel add contiguous runs of the
    // tiles, and the NS runs are added in order (a fixed order either way: reruns are bit-identical).  (round 6)
    const bool pool_runs = p.pool_part && p.n_part >= 64;   // (uniform)
    if (pool_runs) {
        const int ns = max(1, min(8, NT / L)), per = (p.n_part + ns - 1) / ns;
        float *run = ypart;   // [ns][L] of its [4][2 dc][L]: free until the scans
        for (int idx = tid; idx < ns * L; idx += NT) {
            const int sub = idx / L, l = idx - sub * L;
            const int k0 = sub * per, k1 = min(p.n_part, k0 + per);
            const float *pp = p.pool_part + (size_t)b * p.n_part * L + l;
            float sum = 0.f;
            int k = k0;
            for (; k + 16 <= k1; k += 16) {
                float v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = pp[(size_t)(k + q) * L];
#pragma unroll
                for (int q = 0; q < 16; ++q) sum += v[q];
            }
            for (; k < k1; ++k) sum += pp[(size_t)k * L];
            run[idx] = sum;
        }
        __syncthreads();
        for (int l = tid; l < L; l += NT) {
            float sum = run[l];
            for (int sub = 1; sub < ns; ++sub) sum += run[sub * L + l];
            ycs[l] = sum * p.pool_scale;
        }
        __syncthreads();   // ycs holds the pooled values
    }
    for (int l = tid; l < L; l += NT) {
        float pl;
        if (pool_runs) {
            pl = ycs[l];
            const_cast<float *>(p.pooled)[(size_t)b * L + l] = pl;   // kept for the backward
        } else if (p.pool_part) {
            const float *pp = p.pool_part + (size_t)b * p.n_part * L + l;
            float sum = 0.f;
            int k = 0;
            for (; k + 8 <= p.n_part; k += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = pp[(size_t)(k + q) * L];
#pragma unroll
                for (int q = 0; q < 8; ++q) sum += v[q];
            }
            for (; k < p.n_part; ++k) sum += pp[(size_t)k * L];
            pl = sum * p.pool_scale;
            const_cast<float *>(p.pooled)[(size_t)b * L + l] = pl;   // kept for the backward
        } else {
            pl = p.pooled[(size_t)b * L + l];
        }
        for (int i = 0; i < dc; ++i) seq[i * L + l] = lift ? __builtin_fmaf(p.cin_w[i], pl, p.cin_b[i]) : pl;
    }
    __syncthreads();
    OSS_STAMP();
    float *zg = p.zt + (size_t)b * 2 * L * Cc;  // [k][l][c]
    float *dg = p.dts + (size_t)b * 2 * dc * L;
    float *zb, *db, *dlsF = nullptr;   // dlsF [2 dc][L]: softplus(dts + bias), computed once per (row, l) for the 16 state lanes
    if constexpr (use_lds) { zb = ypart + 4 * 2 * dc * L; db = zb + 2 * L * Cc; dlsF = db + 2 * dc * L; } else { zb = zg; db = dg; }
    // z[k][l][c] = sum_i Wxc[k][c][i] seq[i][l]: a thread keeps the dc weights of its column (k, c) in registers and walks l
    // (one output per iteration with its index arithmetic and dc dependent L2 loads measured 13 of the kernel's 32 us at
    // L = 96, profiles/r01_chan_phase_timing.txt)
    {
        const int ncol = 2 * Cc;
        const int tpc = ncol >= NT ? 1 : NT / ncol;          // threads per column
        for (int col = tid % (ncol < NT ? ncol : NT); col < ncol; col += NT) {
            const int sub = ncol < NT ? tid / ncol : 0;
            if (sub >= tpc) break;
            float wv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wv[i] = i < dc ? p.Wxc[col * dc + i] : 0.f;
            const int k = col / Cc, c = col - k * Cc;
#pragma unroll 4
            for (int l = sub; l < L; l += tpc) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < dc) s = __builtin_fmaf(wv[i], seq[i * L + l], s);
                zb[(k * L + l) * Cc + c] = s;
                if (use_lds) zg[(k * L + l) * Cc + c] = s;
            }
        }
    }
    __syncthreads();
    OSS_STAMP();
    // dts[row][l] = sum_r Wdtc[row][r] z[k][l][r]: NT / (2 dc) threads per row, the row's first 8 weights in registers
    {
        const int nrow = 2 * dc, tpr = NT / nrow;      // dc <= 4: >= 32 threads per row
        const int row = tid / tpr, sub = tid - row * tpr;
        if (row < nrow) {
            const int k = row / dc;
            const float brow = p.dt_bias[row];
            float wr[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) wr[r] = r < Rc ? p.Wdtc[row * Rc + r] : 0.f;
#pragma unroll 2
            for (int l = sub; l < L; l += tpr) {
                const float *zr = zb + (k * L + l) * Cc;
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (r < Rc) s = __builtin_fmaf(wr[r], zr[r], s);
                for (int r = 8; r < Rc; ++r) s = __builtin_fmaf(p.Wdtc[row * Rc + r], zr[r], s);
                db[row * L + l] = s;
                if constexpr (use_lds) {
                    dg[row * L + l] = s;
                    float e;
                    dlsF[row * L + l] = softplus_thr(s + brow, e);
                }
            }
        }
    }
    __syncthreads();
    OSS_STAMP();
    // ---- the two channel-direction scans, time across the lanes (round 3) -------------------------------------------------
    // wave = (direction k, group of NPG states); lane p of a 128-step chunk owns steps 2p, 2p + 1 of the direction's walk.
    // Per (row, state): the two local steps, one 64-lane scan of the recurrence monoid (oss_device.h: segment_scan), the chunk
    // carry -- instead of a lane per (row, state) walking the steps one after the other (round 2: four 24-step segments per
    // direction, two passes each, every step paying cross-lane sums over the states and its own LDS round trips: 5.3 us of the
    // 13 us kernel at L = 96, 36.6 of 56.5 at L = 384).  Sums over the states of a group stay in the lane; the four groups'
    // partial sums of y go to LDS and are added in group order afterwards (fixed order: reruns are bit-identical).
    static_assert(NT == 512, "wave = (direction, one of four state groups)");
    constexpr int GPD = NT / 128, NPG = kChN / GPD;   // state groups per direction, states per group
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    {
        const int k = wave / GPD, ng = wave - k * GPD, n0 = ng * NPG;
        float A2[4][NPG], carry[4][NPG], Dv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = k * dc + min(i, dc - 1);
            Dv[i] = p.Dsc[row];
#pragma unroll
            for (int nn = 0; nn < NPG; ++nn) { A2[i][nn] = -__expf(p.A_logs[row * kChN + n0 + nn]) * kLog2e; carry[i][nn] = 0.f; }
        }
        const int nchunk = (L + 127) >> 7;
        for (int c = 0; c < nchunk; ++c) {
            const int t0 = c * 128 + 2 * lane;
            const bool v0 = t0 < L, v1 = t0 + 1 < L;
            const int tc0 = min(t0, L - 1), tc1 = min(t0 + 1, L - 1);
            const int l0 = k ? L - 1 - tc0 : tc0, l1 = k ? L - 1 - tc1 : tc1;
            const float *z0 = zb + (k * L + l0) * Cc + Rc + n0, *z1 = zb + (k * L + l1) * Cc + Rc + n0;
            float B0[NPG], B1[NPG], C0[NPG], C1[NPG];
#pragma unroll
            for (int nn = 0; nn < NPG; ++nn) { B0[nn] = z0[nn]; B1[nn] = z1[nn]; C0[nn] = z0[kChN + nn]; C1[nn] = z1[kChN + nn]; }
            float yv0[4], yv1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                yv0[i] = 0.f; yv1[i] = 0.f;
                if (i < dc) {   // (uniform)
                    const int row = k * dc + i;
                    float dl0, dl1;
                    if constexpr (use_lds) { dl0 = dlsF[row * L + l0]; dl1 = dlsF[row * L + l1]; }
                    else {
                        float e;
                        const float bias = p.dt_bias[row];
                        dl0 = softplus_thr(db[row * L + l0] + bias, e);
                        dl1 = softplus_thr(db[row * L + l1] + bias, e);
                    }
                    dl0 = v0 ? dl0 : 0.f;   // a step past the end: a = 1, B u delta = 0 -- the identity
                    dl1 = v1 ? dl1 : 0.f;
                    const float u0 = seq[i * L + l0], u1 = seq[i * L + l1];
                    float hv0[NPG], hv1[NPG], a0[NPG], a1[NPG], b0[NPG], b1[NPG], P[NPG], h[NPG];
#pragma unroll
                    for (int nn = 0; nn < NPG; ++nn) {
                        a0[nn] = exp2_hw(dl0 * A2[i][nn]); a1[nn] = exp2_hw(dl1 * A2[i][nn]);
                        b0[nn] = dl0 * B0[nn] * u0; b1[nn] = dl1 * B1[nn] * u1;
                        P[nn] = a0[nn] * a1[nn]; h[nn] = __builtin_fmaf(a1[nn], b0[nn], b1[nn]);   // the lane's two steps from a zero state
                    }
                    segment_scan4_64(P, h);
#pragma unroll
                    for (int nn = 0; nn < NPG; ++nn) {
                        const float hp = shift_from_prev_lane(h[nn], 0.f, false), Pp = shift_from_prev_lane(P[nn], 1.f, false);
                        const float hin = __builtin_fmaf(Pp, carry[i][nn], hp);  // the state entering the lane's steps
                        const float h0 = __builtin_fmaf(a0[nn], hin, b0[nn]), h1 = __builtin_fmaf(a1[nn], h0, b1[nn]);
                        hv0[nn] = h0; hv1[nn] = h1;
                        yv0[i] = __builtin_fmaf(C0[nn], h0, yv0[i]);
                        yv1[i] = __builtin_fmaf(C1[nn], h1, yv1[i]);
                        carry[i][nn] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h1), 63));
                    }
                    float *hr = p.hs + (((size_t)b * 2 * dc + row) * L) * kChN + n0;
                    static_assert(NPG == 4, "one 16-byte store per (row, step)");
                    if (v0) *reinterpret_cast<f32x4 *>(hr + (size_t)l0 * kChN) = f32x4{hv0[0], hv0[1], hv0[2], hv0[3]};
                    if (v1) *reinterpret_cast<f32x4 *>(hr + (size_t)l1 * kChN) = f32x4{hv1[0], hv1[1], hv1[2], hv1[3]};
                    if (ng == 0) { yv0[i] = __builtin_fmaf(Dv[i], u0, yv0[i]); yv1[i] = __builtin_fmaf(Dv[i], u1, yv1[i]); }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // the group's partial sums of y (group 0: + D u)
                if (i < dc) {
                    float *yr = ypart + (ng * 2 * dc + k * dc + i) * L;
                    if (v0) yr[l0] = yv0[i];
                    if (v1) yr[l1] = yv1[i];
                }
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < 2 * dc * L; idx += NT)   // y = D u + the groups' sums over their states, in group order
        ybuf[idx] = ((ypart[idx] + ypart[2 * dc * L + idx]) + ypart[2 * 2 * dc * L + idx]) + ypart[3 * 2 * dc * L + idx];
    __syncthreads();
    OSS_STAMP();
    float part = 0.f;
    for (int l = tid; l < L; l += NT) {
        float s = lift ? p.cout_b[0] : 0.f;
        for (int i = 0; i < dc; ++i) s = __builtin_fmaf(lift ? p.cout_w[i] : 1.f, ybuf[i * L + l] + ybuf[(dc + i) * L + l], s);
        ycs[l] = s;
        part += s;
    }
    for (int idx = tid; idx < 2 * dc * L; idx += NT) p.y[(size_t)b * 2 * dc * L + idx] = ybuf[idx];
    const float mu = block_sum<NT>(part, red, tid) / (float)L;
    float q = 0.f;
    for (int l = tid; l < L; l += NT) { const float d = ycs[l] - mu; q = __builtin_fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(block_sum<NT>(q, red, tid) / (float)L + 1e-5f);
    for (int l = tid; l < L; l += NT) {
        p.yc[(size_t)b * L + l] = ycs[l];
        p.c[(size_t)b * L + l] = __builtin_fmaf((ycs[l] - mu) * rstd, p.cn_w[l], p.cn_b[l]);
    }
    if (tid == 0) { p.stat[b * 2] = mu; p.stat[b * 2 + 1] = rstd; }
#ifdef OSS_EXP_CHAN_TIMING
    OSS_STAMP();
    __syncthreads();
    if (b == 0 && tid == 0)
        for (int q = 1; q < ns_; ++q) zg[q - 1] = (float)(ts_[q] - ts_[q - 1]);
#endif
#undef OSS_STAMP
}

// gradient slots of one image in gpart (B, NP); oss_chan_grad_floats() = NP
struct ChanSlots {
    int cnw, cnb, coutw, coutb, dA, dD, dbias, wdtc, wxc, cinw, cinb, total;
    __host__ __device__ ChanSlots(int L, int dc, int Rc, int Cc) {
        cnw = 0; cnb = L; coutw = 2 * L; coutb = coutw + dc; dA = coutb + 1; dD = dA + 2 * dc * kChN; dbias = dD + 2 * dc;
        wdtc = dbias + 2 * dc; wxc = wdtc + 2 * dc * Rc; cinw = wxc + 2 * Cc * dc; cinb = cinw + dc; total = cinb + dc;
    }
};

template <bool use_lds /* the three scratch arrays live in LDS instead of HBM */, int NT>
__global__ void __launch_bounds__(NT)
oss_chan_bwd_kernel(oss_chan_params p, const float *__restrict__ gc /*(B, L): grad of c*/, float *__restrict__ dpool,
                    float *__restrict__ gpart, float *__restrict__ dzt, float *__restrict__ ddts, float *__restrict__ dug,
                    int stage_zdt /* the dt columns of z fit in LDS next to everything else */) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, tid = threadIdx.x, L = p.L, dc = p.dc, Cc = p.Cc, Rc = p.Rc;
#ifdef OSS_EXP_CHAN_TIMING  // phase stamps -> gpart[b = 0][0..] (the other images write zeros there)
    long long ts_[10]; int ns_ = 0;
#define OSS_STAMP() do { ts_[ns_++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define OSS_STAMP() do { } while (0)
#endif
    OSS_STAMP();
    const bool lift = p.cin_w != nullptr;
    const ChanSlots sl(L, dc, Rc, Cc);
    // the scan phase's layout (see there): wave = (direction, group of NPG states), lane = two steps of a 128-step chunk.  The
    // operands it reads from HBM -- B, C and the saved states, written by the forward pass milliseconds ago -- are requested
    // HERE for the first chunk, and for chunk c + 1 while chunk c is computed: one memory latency under the LayerNorm phase
    // instead of one in front of every chunk.
    static_assert(NT == 512, "wave = (direction, one of four state groups)");
    constexpr int GPD = NT / 128, NPG = kChN / GPD;
    static_assert(NPG == 4, "16-byte accesses of a group's states");
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int sk = wave / GPD, sng = wave - sk * GPD, sn0 = sng * NPG;
    struct ScanOps { float B0[NPG], B1[NPG], C0[NPG], C1[NPG]; f32x4 hq0[4], hq1[4], hq2[4]; };
    auto fetch_ops = [&](int c, ScanOps &o) {
        const int u0 = c * 128 + 2 * lane;
        const int c0 = min(u0, L - 1), c1 = min(u0 + 1, L - 1), c2 = min(u0 + 2, L - 1);
        const int l0 = sk ? c0 : L - 1 - c0, l1 = sk ? c1 : L - 1 - c1, l2 = sk ? c2 : L - 1 - c2;
        const float *zsrc = p.zt + (size_t)b * 2 * L * Cc;
        const float *z0 = zsrc + (sk * L + l0) * Cc + Rc + sn0, *z1 = zsrc + (sk * L + l1) * Cc + Rc + sn0;
#pragma unroll
        for (int nn = 0; nn < NPG; ++nn) { o.B0[nn] = z0[nn]; o.B1[nn] = z1[nn]; o.C0[nn] = z0[kChN + nn]; o.C1[nn] = z1[kChN + nn]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float *hr = p.hs + (((size_t)b * 2 * dc + sk * dc + min(i, dc - 1)) * L) * kChN + sn0;
            o.hq0[i] = *reinterpret_cast<const f32x4 *>(hr + (size_t)l0 * kChN);
            o.hq1[i] = *reinterpret_cast<const f32x4 *>(hr + (size_t)l1 * kChN);
            o.hq2[i] = *reinterpret_cast<const f32x4 *>(hr + (size_t)l2 * kChN);
        }
    };
    ScanOps cur;
    fetch_ops(0, cur);
    float *seq = sm;              // [dc][L]
    float *dys = seq + dc * L;    // [L]   grad of yc
    float *dsq = dys + L;         // [dc][L] grad of seq
    float *red = dsq + dc * L;    // [kChRed]
    float *dpl = red + kChRed;    // [4][2 dc][L]: the state groups' partial sums of d delta (before softplus') ...
    float *dpu = dpl + 4 * 2 * dc * L;   // ... and of du
    float *sWx = dpu + 4 * 2 * dc * L;   // [2][Cc][dc]  xc_proj weights (read 2 Cc times per output of the dseq pass)
    float *zdt = sWx + 2 * Cc * dc;   // [2][L][Rc]  the dt columns of z (operands of the dtc_projs weight gradient)
    float *lds_end = zdt + (stage_zdt ? 2 * L * Rc : 0);
    float *gp = gpart + (size_t)b * sl.total;
    for (int idx = tid; idx < 2 * Cc * dc; idx += NT) sWx[idx] = p.Wxc[idx];
    if (stage_zdt) {
        const float *zsrc = p.zt + (size_t)b * 2 * L * Cc;
        for (int kl = tid; kl < 2 * L; kl += NT)
            for (int r = 0; r < Rc; ++r) zdt[kl * Rc + r] = zsrc[kl * Cc + r];
    }
    const float *db = p.dts + (size_t)b * 2 * dc * L;
    float *dzb, *ddb, *dub, *dlsL = nullptr, *sgsL = nullptr;
    if constexpr (use_lds) {
        dzb = lds_end; ddb = dzb + 2 * L * Cc; dub = ddb + 2 * dc * L;
        dlsL = dub + 2 * dc * L; sgsL = dlsL + 2 * dc * L;
        // softplus(dts + bias) and its derivative once per (row, l) instead of once per state lane inside the serial scan
#pragma unroll 4
        for (int idx = tid; idx < 2 * dc * L; idx += NT) {
            const float x = db[idx] + p.dt_bias[idx / L];
            float e;
            dlsL[idx] = softplus_thr(x, e);
            sgsL[idx] = (x <= 20.f) ? e * __builtin_amdgcn_rcpf(1.f + e) : 1.f;
        }
    } else {
        dzb = dzt + (size_t)b * 2 * L * Cc; ddb = ddts + (size_t)b * 2 * dc * L; dub = dug + (size_t)b * 2 * dc * L;
    }
    const float mu = p.stat[b * 2], rstd = p.stat[b * 2 + 1];
    float s1 = 0.f, s2 = 0.f;
    for (int l = tid; l < L; l += NT) {
        const float pl = p.pooled[(size_t)b * L + l];
        for (int i = 0; i < dc; ++i) seq[i * L + l] = lift ? __builtin_fmaf(p.cin_w[i], pl, p.cin_b[i]) : pl;
        const float g0 = gc[(size_t)b * L + l], xh = (p.yc[(size_t)b * L + l] - mu) * rstd, g = g0 * p.cn_w[l];
        gp[sl.cnw + l] = g0 * xh;
        gp[sl.cnb + l] = g0;
        s1 += g;
        s2 = __builtin_fmaf(g, xh, s2);
    }
    const float m1 = block_sum<NT>(s1, red, tid) / (float)L;
    const float m2 = block_sum<NT>(s2, red, tid) / (float)L;
    for (int l = tid; l < L; l += NT) {
        const float xh = (p.yc[(size_t)b * L + l] - mu) * rstd, g = gc[(size_t)b * L + l] * p.cn_w[l];
        dys[l] = rstd * (g - m1 - xh * m2);
    }
    __syncthreads();
    OSS_STAMP();
    const float *yb = p.y + (size_t)b * 2 * dc * L;
    const float *zb = p.zt + (size_t)b * 2 * L * Cc;
    // ---- both reverse recurrences, time across the lanes (round 3; the forward kernel's layout walked backwards) -----------
    // wave = (direction k, group of NPG states); lane p of a 128-step chunk owns the steps tau = 2p, 2p + 1 counted from the
    // END of the direction's walk (t = L - 1 - tau), so  dh_t = dy_t C_t + a_{t+1} dh_{t+1}  is a forward scan in tau with
    // the pair (a_{t+1}, dy_t C_t): one 64-lane scan per (row, state) and chunk instead of a serial walk with cross-lane sums
    // per step (round 2: 15 of the kernel's 27 us at L = 96, 61 of 103 at L = 384).  Sums over the rows of a direction (dB,
    // dC) and over a group's states stay in the lane; the four groups' partial sums of d delta / du go to LDS and a pass with
    // one wave per row adds them in group order and finishes them (softplus', D, the row sums): fixed order, bit-identical reruns.
    {
        const int k = sk, ng = sng, n0 = sn0;
        float Av[4][NPG], carry[4][NPG], carry_a[4][NPG], accA[4][NPG], cwv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = k * dc + min(i, dc - 1);
            cwv[i] = i < dc ? (lift ? p.cout_w[i] : 1.f) : 0.f;
#pragma unroll
            for (int nn = 0; nn < NPG; ++nn) {
                Av[i][nn] = -__expf(p.A_logs[row * kChN + n0 + nn]);
                carry[i][nn] = 0.f; carry_a[i][nn] = 1.f; accA[i][nn] = 0.f;
            }
        }
        const int nchunk = (L + 127) >> 7;
        for (int c = 0; c < nchunk; ++c) {
            const int u0 = c * 128 + 2 * lane;                         // tau of the lane's first step
            const bool v0 = u0 < L, v1 = u0 + 1 < L, v2 = u0 + 2 < L;  // (tau + 2: the step before the lane's second one)
            const int c0 = min(u0, L - 1), c1 = min(u0 + 1, L - 1), c2 = min(u0 + 2, L - 1);
            // position along the channel axis of step tau: direction 0 walks l upwards (t = l), direction 1 downwards
            const int l0 = k ? c0 : L - 1 - c0, l1 = k ? c1 : L - 1 - c1, l2 = k ? c2 : L - 1 - c2;
            ScanOps nxt;
            if (c + 1 < nchunk) fetch_ops(c + 1, nxt);   // in flight while this chunk is computed
            float (&B0)[NPG] = cur.B0, (&B1)[NPG] = cur.B1, (&C0)[NPG] = cur.C0, (&C1)[NPG] = cur.C1;
            f32x4 (&hq0)[4] = cur.hq0, (&hq1)[4] = cur.hq1, (&hq2)[4] = cur.hq2;
            const float dys0 = v0 ? dys[l0] : 0.f, dys1 = v1 ? dys[l1] : 0.f;
            float dB0[NPG], dB1[NPG], dC0[NPG], dC1[NPG], ddl0[4], ddl1[4], du0[4], du1[4];
#pragma unroll
            for (int nn = 0; nn < NPG; ++nn) { dB0[nn] = 0.f; dB1[nn] = 0.f; dC0[nn] = 0.f; dC1[nn] = 0.f; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ddl0[i] = 0.f; ddl1[i] = 0.f; du0[i] = 0.f; du1[i] = 0.f;
                if (i < dc) {   // (uniform)
                    const int row = k * dc + i;
                    float dl0, dl1;
                    if constexpr (use_lds) { dl0 = dlsL[row * L + l0]; dl1 = dlsL[row * L + l1]; }
                    else {
                        float e;
                        const float bias = p.dt_bias[row];
                        dl0 = softplus_thr(db[row * L + l0] + bias, e);
                        dl1 = softplus_thr(db[row * L + l1] + bias, e);
                    }
                    dl0 = v0 ? dl0 : 0.f;   // a step past the end: a = 1, nothing flows
                    dl1 = v1 ? dl1 : 0.f;
                    const float us0 = seq[i * L + l0], us1 = seq[i * L + l1];
                    const float dy0 = cwv[i] * dys0, dy1 = cwv[i] * dys1;
                    float a0[NPG], a1[NPG], al0[NPG], g0[NPG], g1[NPG], P[NPG], h[NPG];
#pragma unroll
                    for (int nn = 0; nn < NPG; ++nn) {
                        a0[nn] = exp2_hw(dl0 * Av[i][nn] * kLog2e); a1[nn] = exp2_hw(dl1 * Av[i][nn] * kLog2e);
                        // the factor in front of the incoming dh at step tau is a of step tau - 1 (= t + 1)
                        al0[nn] = shift_from_prev_lane(a1[nn], carry_a[i][nn], false);
                        g0[nn] = dy0 * C0[nn]; g1[nn] = dy1 * C1[nn];
                        P[nn] = al0[nn] * a0[nn]; h[nn] = __builtin_fmaf(a0[nn], g0[nn], g1[nn]);
                    }
                    segment_scan4_64(P, h);
#pragma unroll
                    for (int nn = 0; nn < NPG; ++nn) {
                        const float A = Av[i][nn];
                        const float hp = shift_from_prev_lane(h[nn], 0.f, false), Pp = shift_from_prev_lane(P[nn], 1.f, false);
                        const float din = __builtin_fmaf(Pp, carry[i][nn], hp);   // a_{t+1} dh_{t+1} reaching the lane's first step
                        const float dh0 = __builtin_fmaf(al0[nn], din, g0[nn]), dh1 = __builtin_fmaf(a0[nn], dh0, g1[nn]);
                        carry[i][nn] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dh1), 63));
                        carry_a[i][nn] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a1[nn]), 63));
                        const float h0 = hq0[i][nn], h1 = hq1[i][nn];
                        const float hp0 = v1 ? h1 : 0.f, hp1 = v2 ? hq2[i][nn] : 0.f;   // the state BEFORE the step (0 before t = 0)
                        dC0[nn] = __builtin_fmaf(dy0, h0, dC0[nn]);
                        dC1[nn] = __builtin_fmaf(dy1, h1, dC1[nn]);
                        dB0[nn] = __builtin_fmaf(dh0 * dl0, us0, dB0[nn]);
                        dB1[nn] = __builtin_fmaf(dh1 * dl1, us1, dB1[nn]);
                        ddl0[i] = __builtin_fmaf(dh0, __builtin_fmaf(A * a0[nn], hp0, B0[nn] * us0), ddl0[i]);
                        ddl1[i] = __builtin_fmaf(dh1, __builtin_fmaf(A * a1[nn], hp1, B1[nn] * us1), ddl1[i]);
                        du0[i] = __builtin_fmaf(dh0 * dl0, B0[nn], du0[i]);
                        du1[i] = __builtin_fmaf(dh1 * dl1, B1[nn], du1[i]);
                        accA[i][nn] = __builtin_fmaf(dh0 * dl0 * a0[nn], hp0, __builtin_fmaf(dh1 * dl1 * a1[nn], hp1, accA[i][nn]));
                    }
                }
            }
            {   // dB, dC of the group's states (already summed over the direction's rows)
                float *dz0 = dzb + (k * L + l0) * Cc + Rc + n0, *dz1 = dzb + (k * L + l1) * Cc + Rc + n0;
#pragma unroll
                for (int nn = 0; nn < NPG; ++nn) {
                    if (v0) { dz0[nn] = dB0[nn]; dz0[kChN + nn] = dC0[nn]; }
                    if (v1) { dz1[nn] = dB1[nn]; dz1[kChN + nn] = dC1[nn]; }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // the group's partial sums of d delta (before softplus') and du
                if (i < dc) {
                    const int o = (ng * 2 * dc + k * dc + i) * L;
                    if (v0) { dpl[o + l0] = ddl0[i]; dpu[o + l0] = du0[i]; }
                    if (v1) { dpl[o + l1] = ddl1[i]; dpu[o + l1] = du1[i]; }
                }
            }
            if (c + 1 < nchunk) cur = nxt;
        }
        // dA_log per (row, state): the sum over the steps
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < dc) {
#pragma unroll
                for (int nn = 0; nn < NPG; ++nn) {
                    const float tA = segment_sum_to_last<64>(accA[i][nn]);
                    if (lane == 63) gp[sl.dA + (k * dc + i) * kChN + n0 + nn] = tA * Av[i][nn];   // d/dA_log: A = -exp(A_log)
                }
            }
        }
        __syncthreads();
        // wave = row: the groups' partial sums in group order, through softplus', plus the D path; dD and d dt_bias of the row
        if (wave < 2 * dc) {
            const int row = wave, i = row % dc;
            const float Dr = p.Dsc[row], cw = lift ? p.cout_w[i] : 1.f, bias = p.dt_bias[row];
            float sB = 0.f, sD = 0.f;
            for (int l = lane; l < L; l += 64) {
                const int o = row * L + l, st = 2 * dc * L;
                const float tl = ((dpl[o] + dpl[st + o]) + dpl[2 * st + o]) + dpl[3 * st + o];
                const float tu = ((dpu[o] + dpu[st + o]) + dpu[2 * st + o]) + dpu[3 * st + o];
                float sg;
                if constexpr (use_lds) { sg = sgsL[o]; }
                else {
                    const float x = db[o] + bias, e = exp2_hw(x * kLog2e);
                    sg = (x <= 20.f) ? e * __builtin_amdgcn_rcpf(1.f + e) : 1.f;
                }
                const float dy = cw * dys[l], ddt = tl * sg;
                ddb[o] = ddt;
                dub[o] = __builtin_fmaf(dy, Dr, tu);
                sB += ddt;
                sD = __builtin_fmaf(dy, seq[i * L + l], sD);
            }
            const float tB = segment_sum_to_last<64>(sB), tD = segment_sum_to_last<64>(sD);
            if (lane == 63) { gp[sl.dbias + row] = tB; gp[sl.dD + row] = tD; }
        }
        // gradients of conv_cout (sums over l of dyc * (y0 + y1), and of dyc): wave o does output o; slot dc = the bias
        if (wave <= dc) {
            const int o = wave;
            float a = 0.f;
            if (lift) {
                for (int l = lane; l < L; l += 64)
                    a = o < dc ? __builtin_fmaf(dys[l], yb[o * L + l] + yb[(dc + o) * L + l], a) : a + dys[l];
            }
            const float t = segment_sum_to_last<64>(a);
            if (lane == 63) gp[sl.coutw + o] = t;
        }
    }
    __syncthreads();
    OSS_STAMP();
    // dt rows of dz
    {   // a thread keeps the dc weights of its column (k, r) in registers and walks l
        const int ncol = 2 * Rc;
        const int tpc = ncol >= NT ? 1 : NT / ncol;
        for (int col = tid % (ncol < NT ? ncol : NT); col < ncol; col += NT) {
            const int sub = ncol < NT ? tid / ncol : 0;
            if (sub >= tpc) break;
            const int k = col / Rc, r = col - k * Rc;
            float wv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wv[i] = i < dc ? p.Wdtc[(k * dc + i) * Rc + r] : 0.f;
#pragma unroll 4
            for (int l = sub; l < L; l += tpc) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < dc) s = __builtin_fmaf(wv[i], ddb[(k * dc + i) * L + l], s);
                dzb[(k * L + l) * Cc + r] = s;
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < dc * L; idx += NT) {
        const int i = idx / L, l = idx - i * L;
        float s = dub[i * L + l] + dub[(dc + i) * L + l];
        for (int k = 0; k < 2; ++k) {
            const float *dz = dzb + (k * L + l) * Cc;
            const float *wx = sWx + k * Cc * dc + i;
#pragma unroll 8
            for (int c = 0; c < Cc; ++c) s = __builtin_fmaf(wx[c * dc], dz[c], s);
        }
        dsq[idx] = s;
    }
    __syncthreads();
    OSS_STAMP();
    for (int l = tid; l < L; l += NT) {
        float s = 0.f;
        for (int i = 0; i < dc; ++i) s = lift ? __builtin_fmaf(p.cin_w[i], dsq[i * L + l], s) : s + dsq[i * L + l];
        dpool[(size_t)b * L + l] = s;
    }
    OSS_STAMP();
    // parameter gradients that are sums over l: one output per thread
    const int n_wdtc = 2 * dc * Rc, n_wxc = 2 * Cc * dc;
    for (int o = tid; o < n_wdtc + n_wxc + 2 * dc; o += NT) {
        float s = 0.f;
        if (o < n_wdtc) {
            const int row = o / Rc, r = o - row * Rc, k = row / dc;
            if (stage_zdt) {
#pragma unroll 8
                for (int l = 0; l < L; ++l) s = __builtin_fmaf(ddb[row * L + l], zdt[(k * L + l) * Rc + r], s);
            } else {
#pragma unroll 8
                for (int l = 0; l < L; ++l) s = __builtin_fmaf(ddb[row * L + l], zb[(k * L + l) * Cc + r], s);
            }
            gp[sl.wdtc + o] = s;
        } else if (o < n_wdtc + n_wxc) {
            const int q = o - n_wdtc, i = q % dc, kc = q / dc, k = kc / Cc, c = kc - k * Cc;
#pragma unroll 8
            for (int l = 0; l < L; ++l) s = __builtin_fmaf(dzb[(k * L + l) * Cc + c], seq[i * L + l], s);
            gp[sl.wxc + q] = s;
        } else {
            const int q = o - n_wdtc - n_wxc, i = q % dc;
            if (lift) {
                if (q < dc) { for (int l = 0; l < L; ++l) s = __builtin_fmaf(dsq[i * L + l], p.pooled[(size_t)b * L + l], s); }
                else        { for (int l = 0; l < L; ++l) s += dsq[i * L + l]; }
            }
            gp[sl.cinw + q] = s;
        }
    }
#ifdef OSS_EXP_CHAN_TIMING
    OSS_STAMP();
    __syncthreads();
    if (tid == 0)
        for (int q = 1; q < ns_; ++q) gp[q - 1] = b == 0 ? (float)(ts_[q] - ts_[q - 1]) : 0.f;
#endif
#undef OSS_STAMP
}

// gsum[j] = sum_b gpart[b][j] in batch order
__global__ void __launch_bounds__(256)
oss_chan_grad_finish(const float *__restrict__ gpart, float *__restrict__ gsum, int B, int NP) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= NP) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += gpart[(size_t)b * NP + j];
    gsum[j] = s;
}

// out[row] = alpha * sum_p a[row, p] * (bmul ? bmul[row, p] : 1);  rows = batch * channels, planes contiguous
template <typename T>
__global__ void __launch_bounds__(256)
oss_rowsum_kernel(const T *__restrict__ a, const T *__restrict__ bmul, float *__restrict__ out, int C, int P, int64_t asb,
                  int64_t asc, int64_t bsb, int64_t bsc, float alpha) {
    __shared__ float red[4];
    const int row = blockIdx.x, b = row / C, c = row - b * C, tid = threadIdx.x;
    const T *ap = a + b * asb + c * asc;
    const T *bp = bmul ? bmul + b * bsb + c * bsc : nullptr;
    float s = 0.f;
    const bool vec = (P % 8 == 0) && ((reinterpret_cast<uintptr_t>(ap) | reinterpret_cast<uintptr_t>(bp)) & 15u) == 0;
    if (vec) {
        // two 2048-element pieces per trip, all four loads requested before the first product (round 4: one piece per trip with the
        // second operand behind `if (bp)` was load -> wait -> load -> wait, twice over at P = 4096)
        const T *bq = bp ? bp : ap;   // no second operand: any readable address, the value is not used
        for (int p = tid * 8; p < P; p += 4096) {
            const bool two = p + 2048 < P;
            const int p2 = two ? p + 2048 : p;
            float a0[8], b0[8], a1[8], b1[8];
            load_items<8>(ap + p, 8, true, a0);
            load_items<8>(bq + p, 8, true, b0);
            load_items<8>(ap + p2, 8, true, a1);
            load_items<8>(bq + p2, 8, true, b1);
#pragma unroll
            for (int i = 0; i < 8; ++i) s = bp ? __builtin_fmaf(a0[i], b0[i], s) : s + a0[i];
            if (two) {
#pragma unroll
                for (int i = 0; i < 8; ++i) s = bp ? __builtin_fmaf(a1[i], b1[i], s) : s + a1[i];
            }
        }
    } else {
        for (int p = tid; p < P; p += 256) s = bp ? __builtin_fmaf(to_f32(ap[p]), to_f32(bp[p]), s) : s + to_f32(ap[p]);
    }
    const float t = block_sum<256>(s, red, tid);
    if (tid == 0) out[row] = t * alpha;
}

// y[row, p] = x[row, p] * (mul ? 1 + mul[row] : 1) + (add ? alpha * add[row] : 0)
template <typename T>
__global__ void __launch_bounds__(256)
oss_row_affine_kernel(const T *__restrict__ x, const float *__restrict__ mul, const float *__restrict__ add, T *__restrict__ y,
                      int C, int P, int64_t xsb, int64_t xsc, float alpha) {
    const int row = blockIdx.y, b = row / C, c = row - b * C;
    const T *xp = x + b * xsb + c * xsc;
    T *yp = y + (size_t)row * P;
    // (round 4) the two per-row scalars and the row's data: three independent loads, requested together (`mul ? 1 + mul[row] : 1` was
    // a branch + load + wait of its own, twice, in front of the data load: three dependent round trips in a 5 us kernel)
    const float *mp = mul ? mul + row : reinterpret_cast<const float *>(x), *ap = add ? add + row : mp;   // stand-ins: readable
    const float mv = *mp, av = *ap;
    const int p = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (p >= P) return;
    const bool vec = (P % 8 == 0) && ((reinterpret_cast<uintptr_t>(xp) | reinterpret_cast<uintptr_t>(yp)) & 15u) == 0;
    float v[8];
    const int valid = min(8, P - p);
    load_items<8>(xp + p, valid, vec, v);
    const float sc = mul ? 1.f + mv : 1.f, sh = add ? alpha * av : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], sc, sh);
    store_items<8>(yp + p, valid, vec, v);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
size_t chan_grad_floats(int L, int dc, int Rc, int Cc) { return (size_t)ChanSlots(L, dc, Rc, Cc).total; }

constexpr size_t kChanLdsMax = 160 * 1024;   // the whole LDS of a CU: one workgroup per image, a handful of images
static int chan_enable_lds(const void *kern, size_t bytes) {
    if (bytes <= 48 * 1024) return 0;
    return (int)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChanLdsMax);
}

static int chan_check(const oss_chan_params &p) {
    if (p.B <= 0 || p.L <= 0 || p.dc < 1 || p.dc > 4 || p.Rc < 1 || p.Cc != p.Rc + 2 * kChN) return OSS_ERR_SHAPE;
    if (!p.pooled || !p.Wxc || !p.Wdtc || !p.dt_bias || !p.A_logs || !p.Dsc || !p.cn_w || !p.cn_b || !p.zt || !p.dts || !p.hs ||
        !p.y || !p.yc || !p.stat || !p.c)
        return OSS_ERR_NULL;
    if ((p.cin_w != nullptr) != (p.cout_w != nullptr)) return OSS_ERR_NULL;
    if (p.cin_w && (!p.cin_b || !p.cout_b)) return OSS_ERR_NULL;
    return 0;
}

int chan_fwd(const oss_chan_params &p, hipStream_t s) {
    if (int e = chan_check(p)) return e;
    size_t smem = sizeof(float) * ((size_t)(3 * p.dc + 1) * p.L + kChRed + 4 * 2 * (size_t)p.dc * p.L);
    if (smem > kChanLdsMax) return OSS_ERR_SHAPE;
    const size_t extra = sizeof(float) * (2 * (size_t)p.L * p.Cc + 4 * (size_t)p.dc * p.L);
    const int use_lds = smem + extra <= kChanLdsMax ? 1 : 0;
    if (use_lds) smem += extra;
    if (use_lds) {
        if (int e = chan_enable_lds(reinterpret_cast<const void *>(oss_chan_fwd_kernel<true, kChNT>), smem)) return e;
        hipLaunchKernelGGL((oss_chan_fwd_kernel<true, kChNT>), dim3(p.B), dim3(kChNT), smem, s, p);
    } else {
        if (int e = chan_enable_lds(reinterpret_cast<const void *>(oss_chan_fwd_kernel<false, kChNT>), smem)) return e;
        hipLaunchKernelGGL((oss_chan_fwd_kernel<false, kChNT>), dim3(p.B), dim3(kChNT), smem, s, p);
    }
    return (int)hipGetLastError();
}

int chan_bwd(const oss_chan_params &p, const float *gc, float *dpool, float *gsum, float *scratch, hipStream_t s) {
    if (int e = chan_check(p)) return e;
    if (!gc || !dpool || !gsum || !scratch) return OSS_ERR_NULL;
    size_t smem = sizeof(float) * ((size_t)(2 * p.dc + 1) * p.L + kChRed + 2 * 4 * 2 * (size_t)p.dc * p.L + 2 * (size_t)p.Cc * p.dc);
    if (smem > kChanLdsMax) return OSS_ERR_SHAPE;
    const size_t zdt_bytes = sizeof(float) * 2 * (size_t)p.L * p.Rc;
    const int stage_zdt = smem + zdt_bytes <= kChanLdsMax ? 1 : 0;
    if (stage_zdt) smem += zdt_bytes;
    const size_t extra = sizeof(float) * (2 * (size_t)p.L * p.Cc + 8 * (size_t)p.dc * p.L);
    const int use_lds = smem + extra <= kChanLdsMax ? 1 : 0;
    if (use_lds) smem += extra;
    if (use_lds) {
        if (int e = chan_enable_lds(reinterpret_cast<const void *>(oss_chan_bwd_kernel<true, kChNT>), smem)) return e;
    } else {
        if (int e = chan_enable_lds(reinterpret_cast<const void *>(oss_chan_bwd_kernel<false, kChNT>), smem)) return e;
    }
    const size_t np = chan_grad_floats(p.L, p.dc, p.Rc, p.Cc);
    float *gpart = scratch;
    float *dzt = gpart + (size_t)p.B * np;
    float *ddts = dzt + (size_t)p.B * 2 * p.L * p.Cc;
    float *dug = ddts + (size_t)p.B * 2 * p.dc * p.L;
    if (use_lds) hipLaunchKernelGGL((oss_chan_bwd_kernel<true, kChNT>), dim3(p.B), dim3(kChNT), smem, s, p, gc, dpool, gpart, dzt, ddts, dug, stage_zdt);
    else         hipLaunchKernelGGL((oss_chan_bwd_kernel<false, kChNT>), dim3(p.B), dim3(kChNT), smem, s, p, gc, dpool, gpart, dzt, ddts, dug, stage_zdt);
    if (defer_finish())
        defer_sum(gpart, p.B, np, np, gsum, np, nullptr);
    else
        hipLaunchKernelGGL(oss_chan_grad_finish, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, gpart, gsum, p.B, (int)np);
    return (int)hipGetLastError();
}

size_t chan_bwd_scratch_floats(int B, int L, int dc, int Rc, int Cc) {
    return (size_t)B * (chan_grad_floats(L, dc, Rc, Cc) + 2 * (size_t)L * Cc + 4 * (size_t)dc * L);
}

int rowsum(oss_dtype io, const void *a, const void *bmul, float *out, int B, int C, int P, int64_t asb, int64_t asc, int64_t bsb,
           int64_t bsc, float alpha, hipStream_t s) {
    dim3 grid(B * C);
    switch (io) {
        case OSS_F32: hipLaunchKernelGGL(oss_rowsum_kernel<float>, grid, dim3(256), 0, s, reinterpret_cast<const float *>(a), reinterpret_cast<const float *>(bmul), out, C, P, asb, asc, bsb, bsc, alpha); break;
        case OSS_F16: hipLaunchKernelGGL(oss_rowsum_kernel<f16_t>, grid, dim3(256), 0, s, reinterpret_cast<const f16_t *>(a), reinterpret_cast<const f16_t *>(bmul), out, C, P, asb, asc, bsb, bsc, alpha); break;
        case OSS_BF16: hipLaunchKernelGGL(oss_rowsum_kernel<bf16_t>, grid, dim3(256), 0, s, reinterpret_cast<const bf16_t *>(a), reinterpret_cast<const bf16_t *>(bmul), out, C, P, asb, asc, bsb, bsc, alpha); break;
        default: return OSS_ERR_SHAPE;
    }
    return (int)hipGetLastError();
}

int row_affine(oss_dtype io, const void *x, const float *mul, const float *add, void *y, int B, int C, int P, int64_t xsb,
               int64_t xsc, float alpha, hipStream_t s) {
    if ((long)B * C > 65535) return OSS_ERR_SHAPE;
    dim3 grid((P + 2047) / 2048, B * C);
    switch (io) {
        case OSS_F32: hipLaunchKernelGGL(oss_row_affine_kernel<float>, grid, dim3(256), 0, s, reinterpret_cast<const float *>(x), mul, add, reinterpret_cast<float *>(y), C, P, xsb, xsc, alpha); break;
        case OSS_F16: hipLaunchKernelGGL(oss_row_affine_kernel<f16_t>, grid, dim3(256), 0, s, reinterpret_cast<const f16_t *>(x), mul, add, reinterpret_cast<f16_t *>(y), C, P, xsb, xsc, alpha); break;
        case OSS_BF16: hipLaunchKernelGGL(oss_row_affine_kernel<bf16_t>, grid, dim3(256), 0, s, reinterpret_cast<const bf16_t *>(x), mul, add, reinterpret_cast<bf16_t *>(y), C, P, xsb, xsc, alpha); break;
        default: return OSS_ERR_SHAPE;
    }
    return (int)hipGetLastError();
}

}  // namespace oss
