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
    float *segP = red + kChRed;      // [NT]: product of a over a wave's time segment, per lane
    float *segH = segP + NT;         // [NT]: the segment's end state from a zero start
    for (int l = tid; l < L; l += NT) {
        const float pl = p.pooled[(size_t)b * L + l];
        for (int i = 0; i < dc; ++i) seq[i * L + l] = lift ? __builtin_fmaf(p.cin_w[i], pl, p.cin_b[i]) : pl;
    }
    __syncthreads();
    OSS_STAMP();
    float *zg = p.zt + (size_t)b * 2 * L * Cc;  // [k][l][c]
    float *dg = p.dts + (size_t)b * 2 * dc * L;
    float *zb, *db, *dlsF = nullptr;   // dlsF [2 dc][L]: softplus(dts + bias), computed once per (row, l) for the 16 state lanes
    if constexpr (use_lds) { zb = segH + NT; db = zb + 2 * L * Cc; dlsF = db + 2 * dc * L; } else { zb = zg; db = dg; }
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
    constexpr int S = NT / 128;   // time segments per direction
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    {
        const int k = wave / S, seg = wave - k * S, i = lane >> 4, n = lane & 15;
        const bool act = i < dc;
        const int row = k * dc + (act ? i : 0);
        const float A2 = -__expf(p.A_logs[row * kChN + n]) * kLog2e;
        const float Dv = p.Dsc[row], bias = p.dt_bias[row];
        const float *dr = (use_lds ? dlsF : db) + row * L, *ur = seq + (act ? i : 0) * L;
        float *hr = p.hs + ((size_t)b * 2 * dc + row) * L * kChN;
        const int Ls = ((L + S - 1) / S + kChU - 1) / kChU * kChU;   // steps per segment, whole fetch chunks
        const int tb = min(seg * Ls, L), te = min(tb + Ls, L);
        auto delta_at = [&](int l) {
            if constexpr (use_lds) { return dr[l]; } else { float e; return softplus_thr(dr[l] + bias, e); }
        };
        // pass A: the recurrence alone from a zero state -> (product of a, end state) of this segment.  A step past the
        // segment's end takes delta = 0: a = 1, B u delta = 0, i.e. the identity.
        float hl = 0.f, sdl = 0.f;
        for (int t0 = tb; t0 < te; t0 += kChU) {
            float dlv[kChU], buv[kChU];
#pragma unroll
            for (int j = 0; j < kChU; ++j) {
                const int t = min(t0 + j, te - 1), l = k ? L - 1 - t : t;
                const float dl = t0 + j < te ? delta_at(l) : 0.f;
                dlv[j] = dl;
                buv[j] = dl * zb[(k * L + l) * Cc + Rc + n] * ur[l];
            }
#pragma unroll
            for (int j = 0; j < kChU; ++j) {
                hl = __builtin_fmaf(exp2_hw(dlv[j] * A2), hl, buv[j]);
                sdl += dlv[j];
            }
        }
        segP[tid] = exp2_hw(sdl * A2);
        segH[tid] = hl;
        __syncthreads();
        float h = 0.f;   // the state entering this segment: segments 0 .. seg-1 of the direction, in order
        for (int q = 0; q < seg; ++q) h = __builtin_fmaf(segP[(k * S + q) * 64 + lane], h, segH[(k * S + q) * 64 + lane]);
        // pass B, per chunk of kChU steps: (1) fetch + everything that does not depend on the recurrence (exp, B u),
        // (2) the recurrence itself -- one dependent FMA per step, (3) the outputs (independent reductions)
        for (int t0 = tb; t0 < te; t0 += kChU) {
            float av[kChU], bu[kChU], us[kChU], Cs[kChU], hv[kChU];
#pragma unroll
            for (int j = 0; j < kChU; ++j) {
                const int t = min(t0 + j, te - 1), l = k ? L - 1 - t : t;
                const float *zr = zb + (k * L + l) * Cc + Rc;
                const float dl = delta_at(l);
                us[j] = ur[l];
                av[j] = exp2_hw(dl * A2);
                bu[j] = dl * zr[n] * us[j];
                Cs[j] = zr[kChN + n];
            }
#pragma unroll
            for (int j = 0; j < kChU; ++j) {
                h = __builtin_fmaf(av[j], h, bu[j]);   // steps past the end repeat the last one; their results are dropped
                hv[j] = h;
            }
#pragma unroll
            for (int j = 0; j < kChU; ++j) {
                const int t = t0 + j;
                if (t < te) {
                    const int l = k ? L - 1 - t : t;
                    if (act) hr[l * kChN + n] = hv[j];
                    const float tot = segment_sum_to_last<16>(Cs[j] * hv[j]);
                    if (n == 15 && act) ybuf[row * L + l] = __builtin_fmaf(Dv, us[j], tot);
                }
            }
        }
    }
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
    float *seq = sm;              // [dc][L]
    float *dys = seq + dc * L;    // [L]   grad of yc
    float *dsq = dys + L;         // [dc][L] grad of seq
    float *red = dsq + dc * L;    // [kChRed], then [5][NT] scratch of the time segments (pairs to chain, partial sums)
    float *sWx = red + kChRed + 5 * NT;   // [2][Cc][dc]  xc_proj weights (read 2 Cc times per output of the dseq pass)
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
    constexpr int S = NT / 128;   // time segments per direction (see the header comment)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    {
        const int k = wave / S, seg = wave - k * S, i = lane >> 4, n = lane & 15;
        const bool act = i < dc;
        const int row = k * dc + (act ? i : 0);
        const float A = -__expf(p.A_logs[row * kChN + n]), A2 = A * kLog2e;
        const float Dv = p.Dsc[row], bias = p.dt_bias[row];
        const float cw = act ? (lift ? p.cout_w[i] : 1.f) : 0.f;
        const float *dr = (use_lds ? dlsL : db) + row * L, *ur = seq + (act ? i : 0) * L;
        const float *sgr = use_lds ? sgsL + row * L : nullptr;
        const float *hr = p.hs + ((size_t)b * 2 * dc + row) * L * kChN;
        const int Ls = ((L + S - 1) / S + kChU - 1) / kChU * kChU;
        const int tb = min(seg * Ls, L), te = min(tb + Ls, L);   // this wave walks t = te-1 ... tb
        float dA = 0.f, dD = 0.f, dbs = 0.f;
        // raw operands of the steps t0, t0 - 1, ...: fetched one chunk ahead of the arithmetic
        auto fetch = [&](int t0, float (&xs)[kChU], float (&Bs)[kChU], float (&Cs)[kChU], float (&hv)[kChU + 1]) {
#pragma unroll
            for (int j = 0; j < kChU; ++j) {
                const int t = max(t0 - j, 0), l = k ? L - 1 - t : t;
                const float *zr = zb + (k * L + l) * Cc + Rc;
                xs[j] = dr[l];
                Bs[j] = zr[n];
                Cs[j] = zr[kChN + n];
                hv[j] = hr[l * kChN + n];
            }
            const int t = t0 - kChU;  // the state before the chunk's last step
            const int tc = max(t, 0), l = k ? L - 1 - tc : tc;
            hv[kChU] = t >= 0 ? hr[l * kChN + n] : 0.f;
        };
        float xs[kChU], Bs[kChU], Cs[kChU], hv[kChU + 1];
        if (te > tb) fetch(te - 1, xs, Bs, Cs, hv);   // in flight across pass A
        // pass A: the reverse recurrence alone from a zero carry -> (product of a, carry leaving the segment at tb)
        float cl = 0.f, sdl = 0.f;
        for (int t0 = te - 1; t0 >= tb; t0 -= kChU) {
            float dlv[kChU], gv[kChU];
#pragma unroll
            for (int j = 0; j < kChU; ++j) {
                const int t = max(t0 - j, tb), l = k ? L - 1 - t : t;
                const bool on = t0 - j >= tb;
                float dl;
                if constexpr (use_lds) { dl = dr[l]; } else { float e; dl = softplus_thr(dr[l] + bias, e); }
                dlv[j] = on ? dl : 0.f;
                gv[j] = on ? cw * dys[l] * zb[(k * L + l) * Cc + Rc + kChN + n] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < kChU; ++j) {   // off steps: a = 1, g = 0 -- the identity
                cl = exp2_hw(dlv[j] * A2) * (gv[j] + cl);
                sdl += dlv[j];
            }
        }
        float *segP = red + kChRed, *segC = segP + NT;
        segP[tid] = exp2_hw(sdl * A2);
        segC[tid] = cl;
        __syncthreads();
        float carry = 0.f;   // a_{t+1} dh_{t+1} entering this segment from the later ones, latest first
        for (int q = S - 1; q > seg; --q) carry = __builtin_fmaf(segP[(k * S + q) * 64 + lane], carry, segC[(k * S + q) * 64 + lane]);
        for (int t0 = te - 1; t0 >= tb; t0 -= kChU) {
            float nxs[kChU], nBs[kChU], nCs[kChU], nhv[kChU + 1];
            if (t0 - kChU >= tb) fetch(t0 - kChU, nxs, nBs, nCs, nhv);
            // (1) recurrence-independent terms, (2) the reverse recurrence (two dependent FMAs per step),
            // (3) the gradients of the step (independent of each other)
            float us[kChU], dyv[kChU], dls[kChU], sg[kChU], av[kChU], dhv[kChU];
#pragma unroll
            for (int j = 0; j < kChU; ++j) {
                const int t = max(t0 - j, 0), l = k ? L - 1 - t : t;
                if constexpr (use_lds) {
                    dls[j] = xs[j];
                    sg[j] = sgr[l];
                } else {
                    const float x = xs[j] + bias;
                    float e;
                    dls[j] = softplus_thr(x, e);
                    sg[j] = (x <= 20.f) ? e * __builtin_amdgcn_rcpf(1.f + e) : 1.f;
                }
                av[j] = exp2_hw(dls[j] * A2);
                us[j] = ur[l];
                dyv[j] = t0 - j >= tb ? cw * dys[l] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < kChU; ++j) {
                const bool on = t0 - j >= tb;
                const float dh = __builtin_fmaf(dyv[j], Cs[j], carry);
                dhv[j] = dh;
                carry = on ? av[j] * dh : carry;
            }
#pragma unroll
            for (int j = 0; j < kChU; ++j) {
                const int t = t0 - j;
                if (t >= tb) {
                    const int l = k ? L - 1 - t : t;
                    const float dl = dls[j], u = us[j], Bv = Bs[j], a = av[j], dh = dhv[j];
                    const float h = hv[j], hp = t > 0 ? hv[j + 1] : 0.f;
                    const float dCs = sum_over_rows(dyv[j] * h, lane);
                    const float dBs = sum_over_rows(dh * dl * u, lane);
                    if (i == 0) {
                        float *dz = dzb + (k * L + l) * Cc + Rc;
                        dz[n] = dBs;
                        dz[kChN + n] = dCs;
                    }
                    const float ddl = segment_sum_to_last<16>(dh * __builtin_fmaf(A * a, hp, Bv * u));
                    const float du = segment_sum_to_last<16>(dh * dl * Bv);
                    dA = __builtin_fmaf(dh * dl * a, hp, dA);
                    if (n == 15 && act) {
                        const float ddt = ddl * sg[j];
                        ddb[row * L + l] = ddt;
                        dub[row * L + l] = __builtin_fmaf(dyv[j], Dv, du);
                        dbs += ddt;
                        dD = __builtin_fmaf(dyv[j], u, dD);
                    }
                }
            }
            if (t0 - kChU >= tb) {
#pragma unroll
                for (int j = 0; j < kChU; ++j) { xs[j] = nxs[j]; Bs[j] = nBs[j]; Cs[j] = nCs[j]; hv[j] = nhv[j]; }
                hv[kChU] = nhv[kChU];
            }
        }
        // the segments' partial sums of dA / dD / dbias, combined in segment order by the wave of segment 0
        float *accA = segC + NT, *accD = accA + NT, *accB = accD + NT;
        accA[tid] = dA; accD[tid] = dD; accB[tid] = dbs;
        __syncthreads();
        if (seg == 0 && act) {
            float tA = 0.f, tD = 0.f, tB = 0.f;
            for (int q = 0; q < S; ++q) {
                const int o = (k * S + q) * 64 + lane;
                tA += accA[o]; tD += accD[o]; tB += accB[o];
            }
            gp[sl.dA + row * kChN + n] = tA * A;  // d/dA_log: A = -exp(A_log)
            if (n == 15) { gp[sl.dD + row] = tD; gp[sl.dbias + row] = tB; }
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
        for (int p = tid * 8; p < P; p += 2048) {
            float av[8], bv[8];
            load_items<8>(ap + p, 8, true, av);
            if (bp) load_items<8>(bp + p, 8, true, bv);
#pragma unroll
            for (int i = 0; i < 8; ++i) s = bp ? __builtin_fmaf(av[i], bv[i], s) : s + av[i];
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
    const float sc = mul ? 1.f + mul[row] : 1.f, sh = add ? alpha * add[row] : 0.f;
    const int p = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (p >= P) return;
    const bool vec = (P % 8 == 0) && ((reinterpret_cast<uintptr_t>(xp) | reinterpret_cast<uintptr_t>(yp)) & 15u) == 0;
    float v[8];
    const int valid = min(8, P - p);
    load_items<8>(xp + p, valid, vec, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], sc, sh);
    store_items<8>(yp + p, valid, vec, v);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
size_t chan_grad_floats(int L, int dc, int Rc, int Cc) { return (size_t)ChanSlots(L, dc, Rc, Cc).total; }

constexpr size_t kChanLdsMax = 96 * 1024;
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
    size_t smem = sizeof(float) * ((size_t)(3 * p.dc + 1) * p.L + kChRed + 2 * kChNT);
    if (smem > 48 * 1024) return OSS_ERR_SHAPE;
    const size_t extra = sizeof(float) * (2 * (size_t)p.L * p.Cc + 4 * (size_t)p.dc * p.L);
    const int use_lds = smem + extra <= kChanLdsMax ? 1 : 0;
    if (use_lds) smem += extra;
    if (use_lds) {
        if (int e = chan_enable_lds(reinterpret_cast<const void *>(oss_chan_fwd_kernel<true, kChNT>), smem)) return e;
        hipLaunchKernelGGL((oss_chan_fwd_kernel<true, kChNT>), dim3(p.B), dim3(kChNT), smem, s, p);
    } else {
        hipLaunchKernelGGL((oss_chan_fwd_kernel<false, kChNT>), dim3(p.B), dim3(kChNT), smem, s, p);
    }
    return (int)hipGetLastError();
}

int chan_bwd(const oss_chan_params &p, const float *gc, float *dpool, float *gsum, float *scratch, hipStream_t s) {
    if (int e = chan_check(p)) return e;
    if (!gc || !dpool || !gsum || !scratch) return OSS_ERR_NULL;
    size_t smem = sizeof(float) * ((size_t)(2 * p.dc + 1) * p.L + kChRed + 5 * kChNT + 2 * (size_t)p.Cc * p.dc);
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
