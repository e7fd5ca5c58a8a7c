// oss_layernorm.hip -- per-pixel LayerNorm over the channel axis of an NCHW tensor, NCHW in / NCHW out
// (norm1, norm2, out_norm, channel_norm of the OSS block).  The reference does it as
// rearrange 'b c h w -> b (h w) c' -> mean / var(unbiased=False) -> (x-mu)/sqrt(var+1e-5)*w+b ->
// rearrange back (SRGAN/VmambaIR/archs/MambaSISR6_arch.py:144-195): two permute copies and ~6
// elementwise/reduction kernels per call, three calls per block.  Here: one kernel forward, one kernel
// (+ a tiny finishing kernel for dweight/dbias) backward, no permutes.  HBM-bound.
//
// Workgroup = 64 consecutive pixels of one image x ALL channels; NW = 4 / 8 / 16 waves, wave w owns the
// channels c = w, w + NW, w + 2 NW, ... and lane l the pixel l of the tile, so every load / store
// instruction of a wave moves 64 consecutive pixels of one channel plane (coalesced) and the channel
// axis is spread over the waves (the 64x64 training patch has only 4096 pixels per image: a
// thread-per-pixel layout leaves half the chip idle and serialises 3 C strided loads per thread).
// A wave keeps its <= 24 channel values in registers (CPW = 24; C <= 384), so x is read from HBM
// once; wider C streams (re-reads hit L2).  Per-pixel sums cross the waves through LDS in a fixed
// order (bit-reproducible).  Two-pass variance (no E[x^2]-mu^2 cancellation).
// Optional fused epilogue: y *= silu(gate) (SS2D_1: y1 * act(z), MambaSISR6_arch.py:488-493).
#include "oss_device.h"
#include "oss_host.h"

namespace oss {

__device__ __forceinline__ float silu_f(float z) { return z * __builtin_amdgcn_rcpf(1.f + exp2_hw(-z * kLog2e)); }

constexpr int kLnMaxWaves = 16;

// value of lane `src` (compile-time after unrolling) in every lane
__device__ __forceinline__ float ln_lane_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// sum over the workgroup's waves of V values per lane; fixed order.  red: [kLnMaxWaves][V][64]
template <int V>
__device__ __forceinline__ void ln_cross_wave_sum(float *red, float (&v)[V], int wave, int lane, int nw) {
#pragma unroll
    for (int i = 0; i < V; ++i) red[(wave * V + i) * 64 + lane] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < V; ++i) {
        float t = 0.f;
        for (int k = 0; k < nw; ++k) t += red[(k * V + i) * 64 + lane];
        v[i] = t;
    }
}

// x: (B, C, P) with element strides (xsb, xsc), pixels contiguous.  y/gate: contiguous (B, C, P).
// CPW > 0: channels per wave held in registers (needs C <= CPW * NW); CPW == 0: streaming.
// V: consecutive pixels per lane (2: 4-/8-byte accesses; needs even P and even strides); MAXT: 64 * waves bound.
template <typename TX, typename TY, bool GATE, int CPW, int V, int MAXT>
__global__ void __launch_bounds__(MAXT)
oss_ln_nchw_fwd_kernel(const TX *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                       const TY *__restrict__ gate, TY *__restrict__ y, float *__restrict__ mean_out,
                       float *__restrict__ rstd_out, int C, int P, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, float eps,
                       float *__restrict__ pool_part /* [B][gridDim.x][C] or NULL: per channel, the sum of the workgroup's OUTPUT values
                       (as stored, i.e. rounded to TY) -- SS2D_1's channel branch starts from mean_hw of this tensor (MambaSISR6_arch.py:
                       438-441), and a channel belongs to one wave here, so the pooling pass over y comes down to a 64-lane sum */) {
    __shared__ float red[2][kLnMaxWaves * 64 * V];
    constexpr int NI = CPW > 0 ? CPW : 1;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int p = (blockIdx.x * 64 + lane) * V;
    const bool ok = p < P;  // V == 2 only with P even: a pair is in range as a whole
    const int pc = ok ? p : 0;
    const TX *xp = x + b * xsb + pc;
    const TY *gp = GATE ? gate + b * gsb + pc : nullptr;
    const bool with_bias = bias != nullptr;
    float xv[NI][V], zv[NI][V];
    float s[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s[i] = 0.f;
    if constexpr (CPW > 0) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) {  // clamped index: all loads issue back to back
            const int c = wave + i * nw, cc = c < C ? c : C - 1;
            load_v<TX, V>(xp + cc * xsc, xv[i]);
            if constexpr (GATE) load_v<TY, V>(gp + cc * gsc, zv[i]);
        }
#pragma unroll
        for (int i = 0; i < CPW; ++i)
#pragma unroll
            for (int u = 0; u < V; ++u) s[u] += (wave + i * nw < C) ? xv[i][u] : 0.f;
    } else {
        for (int c = wave; c < C; c += nw) {
            float t[V];
            load_v<TX, V>(xp + c * xsc, t);
#pragma unroll
            for (int u = 0; u < V; ++u) s[u] += t[u];
        }
    }
    ln_cross_wave_sum<V>(red[0], s, wave, lane, nw);
    float mu[V], q[V], rstd[V];
#pragma unroll
    for (int u = 0; u < V; ++u) { mu[u] = s[u] / (float)C; q[u] = 0.f; }
    if constexpr (CPW > 0) {
#pragma unroll
        for (int i = 0; i < CPW; ++i)
#pragma unroll
            for (int u = 0; u < V; ++u) {
                const float d = (wave + i * nw < C) ? xv[i][u] - mu[u] : 0.f;
                q[u] = __builtin_fmaf(d, d, q[u]);
            }
    } else {
        for (int c = wave; c < C; c += nw) {
            float t[V];
            load_v<TX, V>(xp + c * xsc, t);
#pragma unroll
            for (int u = 0; u < V; ++u) { const float d = t[u] - mu[u]; q[u] = __builtin_fmaf(d, d, q[u]); }
        }
    }
    ln_cross_wave_sum<V>(red[1], q, wave, lane, nw);
#pragma unroll
    for (int u = 0; u < V; ++u) rstd[u] = 1.0f / sqrtf(q[u] / (float)C + eps);
    if (wave == 0 && ok) {
        store_v<float, V>(mean_out + (size_t)b * P + p, mu);
        store_v<float, V>(rstd_out + (size_t)b * P + p, rstd);
    }
    TY *yp = y + (size_t)b * C * P + pc;
    float wvec = 0.f, bvec = 0.f;   // the wave's weights / biases: lane i = channel slot i (see the backward kernel)
    if constexpr (CPW > 0) {
        const int cs = min(wave + min(lane, CPW - 1) * nw, C - 1);
        wvec = w[cs];
        bvec = with_bias ? bias[cs] : 0.f;
    }
    auto emit = [&](int c, float wc, float bc, const float (&xval)[V], const float (&zval)[V]) {
        float o[V];
#pragma unroll
        for (int u = 0; u < V; ++u) {
            // BiasFree (MambaSISR6_arch.py:160-164) divides x (not x - mu) by sigma
            o[u] = with_bias ? (xval[u] - mu[u]) * rstd[u] * wc + bc : xval[u] * rstd[u] * wc;
            if constexpr (GATE) o[u] *= silu_f(zval[u]);
        }
        if (ok) store_v<TY, V>(yp + (size_t)c * P, o);
        if (pool_part) {   // (uniform)
            float t = 0.f;
#pragma unroll
            for (int u = 0; u < V; ++u) t += ok ? to_f32(from_f32<TY>(o[u])) : 0.f;
            t = segment_sum_to_last<64>(t);
            if (lane == 63) pool_part[((size_t)b * gridDim.x + blockIdx.x) * C + c] = t;
        }
    };
    if constexpr (CPW > 0) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) { const int c = wave + i * nw; if (c < C) emit(c, ln_lane_bcast(wvec, i), ln_lane_bcast(bvec, i), xv[i], zv[i]); }
    } else {
        for (int c = wave; c < C; c += nw) {
            float t[V], z[V];
            load_v<TX, V>(xp + c * xsc, t);
#pragma unroll
            for (int u = 0; u < V; ++u) z[u] = 0.f;
            if constexpr (GATE) load_v<TY, V>(gp + c * gsc, z);
            emit(c, w[c], with_bias ? bias[c] : 0.f, t, z);
        }
    }
}

// dx, and per-workgroup partials of dweight / dbias (and dgate when GATE).
//   WithBias: xhat = (x-mu) rstd;  g = dy w;  dx = rstd (g - mean(g) - xhat mean(g xhat))
//   BiasFree: y = x rstd w, rstd = (var+eps)^-1/2 with var around mu:
//             dx = rstd g - (x - mu) rstd^3 mean(g x)
// A channel belongs to one wave, so its dweight / dbias partial is a plain 64-lane sum, stored straight
// to part[blk][2][C].
template <typename TX, typename TY, bool GATE, int CPW, int V, int MAXT>
__global__ void __launch_bounds__(MAXT)
oss_ln_nchw_bwd_kernel(const TX *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                       const TY *__restrict__ gate, const TY *__restrict__ dy, const float *__restrict__ mean_in,
                       const float *__restrict__ rstd_in, TX *__restrict__ dx, TY *__restrict__ dgate,
                       float *__restrict__ part /*[nblk][2][C]*/, int C, int P, int64_t xsb, int64_t xsc, int64_t gsb,
                       int64_t gsc, const TX *__restrict__ res /* (B, C, P) or NULL: added to dx (gradient of a skip connection) */,
                       int64_t dgsb /* batch stride of dgate (channel stride P): lets it land in one half of a wider buffer */,
                       const float *__restrict__ dy_mul, const float *__restrict__ dy_add, float add_scale /* dy_add != NULL: the
                       gradient that enters is dy * (1 + dy_mul[b, c]) + add_scale * dy_add[b, c] (dy_mul NULL: dy + ...), both
                       (B, C) -- the backward of SS2D_1's channel gate (y2 * c + y2 with c a function of mean_hw(y2)) folded into
                       this kernel's load instead of a pass of its own */) {
    __shared__ float red[2][kLnMaxWaves * 64 * V];
    // per-channel dweight / dbias partials of this workgroup, staged here and written out in one coalesced pass: stored
    // to global memory channel by channel inside the first pass, every store dragged an s_waitcnt vmcnt behind it that
    // also waited for the store of the channel before (gfx9 counts stores in vmcnt)
    __shared__ float pws[CPW > 0 ? 2 * kLnMaxWaves * CPW : 1];
    constexpr int NI = CPW > 0 ? CPW : 1;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int p = (blockIdx.x * 64 + lane) * V;
    const bool ok = p < P;
    const int pc = ok ? p : 0;
    const TX *xp = x + b * xsb + pc;
    // the skip connection's gradient (added to dx).  Loaded without a branch: with res == NULL the loads re-read x at the very
    // addresses the pass has just fetched (cache hits, values discarded) -- never dx, the output other lanes are writing
    // (ADVICE r2).  A wave-uniform `if (res)` around the loads makes hipcc treat every later use of the loaded registers as
    // "maybe still in flight": s_waitcnt vmcnt(0) before each -- which then waits for the previous STORE.
    const bool has_res = res != nullptr;
    const TX *rsp = has_res ? res + (size_t)b * C * P + pc : xp;
    const int64_t rsc = has_res ? (int64_t)P : xsc;
    const TY *gyp = dy + (size_t)b * C * P + pc;
    const TY *gp = GATE ? gate + b * gsb + pc : nullptr;
    const bool with_bias = bias != nullptr;
    float mu[V], rstd[V];
    load_v<float, V>(mean_in + (size_t)b * P + pc, mu);
    load_v<float, V>(rstd_in + (size_t)b * P + pc, rstd);
    const float okf = ok ? 1.f : 0.f;
    float *pw_out = part + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
    float xv[NI][V], gv[NI][V], zv[NI][V], rv[NI][V];  // x, dy, gate, res
    float s1[V], s2[V];
#pragma unroll
    for (int u = 0; u < V; ++u) { s1[u] = 0.f; s2[u] = 0.f; }
    // the wave's weights and biases as ONE vector load each (lane i = channel slot i): a scalar load per channel, waited for
    // inside the pass, was a round trip per channel
    float wvec = 0.f, bvec = 0.f, mvec = 1.f, avec = 0.f;
    const bool affine = dy_add != nullptr, scaled = dy_mul != nullptr;
    const float *mulp = scaled ? dy_mul + (size_t)b * C : w, *addp = affine ? dy_add + (size_t)b * C : w;   // always readable
    if constexpr (CPW > 0) {
        const int cs = min(wave + min(lane, CPW - 1) * nw, C - 1);
        wvec = w[cs];
        bvec = with_bias ? bias[cs] : 0.f;
        const float mr = mulp[cs], ar = addp[cs];
        mvec = scaled ? 1.f + mr : 1.f;
        avec = affine ? ar * add_scale : 0.f;
    }
    auto first = [&](int c, float wc, const float (&xval)[V], const float (&gy)[V], const float (&zval)[V]) {
        float aw = 0.f, ab = 0.f;
#pragma unroll
        for (int u = 0; u < V; ++u) {
            const float xh = with_bias ? (xval[u] - mu[u]) * rstd[u] : xval[u] * rstd[u];
            float g = gy[u] * okf;
            if constexpr (GATE) g *= silu_f(zval[u]);
            aw = __builtin_fmaf(g, xh, aw);
            ab += g;
            const float gw = g * wc;
            s1[u] += gw;
            s2[u] = __builtin_fmaf(gw, with_bias ? xh : xval[u], s2[u]);
        }
        const float pw = segment_sum_to_last<64>(aw);
        const float pb = segment_sum_to_last<64>(ab);
        if (lane == 63) {
            if constexpr (CPW > 0) { pws[c] = pw; pws[C + c] = pb; }
            else                   { pw_out[c] = pw; pw_out[C + c] = pb; }
        }
    };
    if constexpr (CPW > 0) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            const int c = wave + i * nw, cc = c < C ? c : C - 1;
            load_v<TX, V>(xp + cc * xsc, xv[i]);
            load_v<TY, V>(gyp + (size_t)cc * P, gv[i]);
            if constexpr (GATE) load_v<TY, V>(gp + cc * gsc, zv[i]);
        }
        // needed last, in flight with everything else.  (The gated form -- out_norm, which has no skip connection in any of
        // the nets -- keeps a branch: unconditional dummy loads would double its traffic.)
        if (!GATE || has_res) {
#pragma unroll
            for (int i = 0; i < CPW; ++i) {
                const int c = wave + i * nw, cc = c < C ? c : C - 1;
                load_v<TX, V>(rsp + cc * rsc, rv[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < CPW; ++i)
#pragma unroll
                for (int u = 0; u < V; ++u) rv[i][u] = 0.f;
        }
        if (affine) {
#pragma unroll
            for (int i = 0; i < CPW; ++i) {
                const float mi = ln_lane_bcast(mvec, i), ai = ln_lane_bcast(avec, i);
#pragma unroll
                for (int u = 0; u < V; ++u) gv[i][u] = __builtin_fmaf(gv[i][u], mi, ai);
            }
        }
#pragma unroll
        for (int i = 0; i < CPW; ++i) { const int c = wave + i * nw; if (c < C) first(c, ln_lane_bcast(wvec, i), xv[i], gv[i], zv[i]); }
    } else {
        for (int c = wave; c < C; c += nw) {
            float t[V], g[V], z[V];
            load_v<TX, V>(xp + c * xsc, t);
            load_v<TY, V>(gyp + (size_t)c * P, g);
            if (affine) {
                const float mi = scaled ? 1.f + mulp[c] : 1.f, ai = addp[c] * add_scale;
#pragma unroll
                for (int u = 0; u < V; ++u) g[u] = __builtin_fmaf(g[u], mi, ai);
            }
#pragma unroll
            for (int u = 0; u < V; ++u) z[u] = 0.f;
            if constexpr (GATE) load_v<TY, V>(gp + c * gsc, z);
            first(c, w[c], t, g, z);
        }
    }
    ln_cross_wave_sum<V>(red[0], s1, wave, lane, nw);
    ln_cross_wave_sum<V>(red[1], s2, wave, lane, nw);
    if constexpr (CPW > 0) {   // (the barriers above ordered the pws writes)
        for (int t = threadIdx.x; t < 2 * C; t += blockDim.x) pw_out[t] = pws[t];
    }
    float m1[V], m2[V];
#pragma unroll
    for (int u = 0; u < V; ++u) { m1[u] = s1[u] / (float)C; m2[u] = s2[u] / (float)C; }
    TX *dxp = dx + (size_t)b * C * P + pc;
    TY *dgp = GATE ? dgate + (size_t)b * dgsb + pc : nullptr;
    auto second = [&](int c, float wc, float bc, const float (&xval)[V], const float (&gy)[V], const float (&zval)[V], const float (&rval)[V]) {
        float d[V], dg[V];
#pragma unroll
        for (int u = 0; u < V; ++u) {
            const float xh = (xval[u] - mu[u]) * rstd[u];
            float g = gy[u];
            if constexpr (GATE) {
                const float sg = __builtin_amdgcn_rcpf(1.f + exp2_hw(-zval[u] * kLog2e));  // sigmoid(z)
                const float sl = zval[u] * sg;
                const float o = with_bias ? xh * wc + bc : xval[u] * rstd[u] * wc;  // LN output before the gate
                dg[u] = gy[u] * o * (sg + sl * (1.f - sg));
                g *= sl;
            }
            const float gw = g * wc;
            d[u] = with_bias ? rstd[u] * (gw - m1[u] - xh * m2[u]) : rstd[u] * gw - xh * rstd[u] * rstd[u] * m2[u];
        }
#pragma unroll
        for (int u = 0; u < V; ++u) d[u] += has_res ? rval[u] : 0.f;
        if (ok) {
            store_v<TX, V>(dxp + (size_t)c * P, d);
            if constexpr (GATE) store_v<TY, V>(dgp + (size_t)c * P, dg);
        }
    };
    if constexpr (CPW > 0) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) { const int c = wave + i * nw; if (c < C) second(c, ln_lane_bcast(wvec, i), ln_lane_bcast(bvec, i), xv[i], gv[i], zv[i], rv[i]); }
    } else {
        for (int c = wave; c < C; c += nw) {
            float t[V], g[V], z[V], r[V];
            load_v<TX, V>(xp + c * xsc, t);
            load_v<TY, V>(gyp + (size_t)c * P, g);
            if (affine) {
                const float mi = scaled ? 1.f + mulp[c] : 1.f, ai = addp[c] * add_scale;
#pragma unroll
                for (int u = 0; u < V; ++u) g[u] = __builtin_fmaf(g[u], mi, ai);
            }
#pragma unroll
            for (int u = 0; u < V; ++u) { z[u] = 0.f; r[u] = 0.f; }
            if constexpr (GATE) load_v<TY, V>(gp + c * gsc, z);
            load_v<TX, V>(rsp + c * rsc, r);
            second(c, w[c], with_bias ? bias[c] : 0.f, t, g, z, r);
        }
    }
}

__global__ void __launch_bounds__(256)
oss_ln_nchw_bwd_finish(const float *__restrict__ part, float *__restrict__ dw, float *__restrict__ db, int nblk, int C) {
    // 16 outputs x 16 slices of the partial list per workgroup (2C / 16 workgroups: enough of them to pull the
    // partials at bandwidth); slices combined in a fixed order
    __shared__ float red[16][17];
    const int col = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + col;
    float s = 0.f;
    if (i < 2 * C)
        for (int k = slice; k < nblk; k += 16) s += part[(size_t)k * 2 * C + i];
    red[slice][col] = s;
    __syncthreads();
    if (slice == 0 && i < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][col];
        if (i < C) dw[i] = t;
        else if (db) db[i - C] = t;
    }
}

constexpr int kLnCPW = 24;

// waves per workgroup: the fewest of 4 / 8 / 16 that keep a wave's channels in registers
static int ln_waves(int C) { return C <= 4 * kLnCPW ? 4 : (C <= 8 * kLnCPW ? 8 : 16); }
// ... and 8 instead of 4 when 4 would leave the chip short of waves (a 64x64 patch at batch 8 is 256 tiles of 128 pixels:
// one 4-wave workgroup per CU, i.e. one wave per SIMD and nothing to hide the load latency behind)
static int ln_waves(int B, int C, int P) {
    int nw = ln_waves(C);
    if (nw == 4 && C >= 48 && (long)B * ((P + 127) / 128) * 4 < 4096) nw = 8;
    return nw;
}
static bool ln_cached(int C) { return C <= kLnMaxWaves * kLnCPW; }
// two pixels per lane: needs pair-aligned rows, and the register room of <= 8 waves per workgroup
static bool ln_pairs(int C, int P, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, const void *a, const void *b2,
                     const void *c2, const void *d2) {
    const uintptr_t al = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b2) | reinterpret_cast<uintptr_t>(c2) |
                         reinterpret_cast<uintptr_t>(d2);
    return ln_cached(C) && ln_waves(C) <= 8 && P % 2 == 0 && xsb % 2 == 0 && xsc % 2 == 0 && gsb % 2 == 0 && gsc % 2 == 0 &&
           (al & 7u) == 0;
}
static int ln_tile(bool pairs) { return pairs ? 128 : 64; }

size_t ln_nchw_bwd_partial_floats(int B, int C, int P) { return (size_t)((P + 63) / 64) * B * 2 * C; }

// (kLnCPW / 2 channel slots per wave when they suffice -- C = 96 on 8 waves, C = 48 on 4: a slot past C re-loads channel
// C - 1, so the full-width instantiation spent half of its load instructions, and of the vmcnt range, on duplicates)
#define OSS_LN_LAUNCH(KERN, GATE_, ...)                                                                         \
    do {                                                                                                        \
        const bool half = C <= nw * (kLnCPW / 2);                                                               \
        if (!ln_cached(C))   hipLaunchKernelGGL((KERN<TX, TY, GATE_, 0, 1, 1024>), grid, block, 0, s, __VA_ARGS__);      \
        else if (!pairs)     hipLaunchKernelGGL((KERN<TX, TY, GATE_, kLnCPW, 1, 1024>), grid, block, 0, s, __VA_ARGS__); \
        else if (nw == 4 && half) hipLaunchKernelGGL((KERN<TX, TY, GATE_, kLnCPW / 2, 2, 256>), grid, block, 0, s, __VA_ARGS__);  \
        else if (nw == 4)    hipLaunchKernelGGL((KERN<TX, TY, GATE_, kLnCPW, 2, 256>), grid, block, 0, s, __VA_ARGS__);  \
        else if (half)       hipLaunchKernelGGL((KERN<TX, TY, GATE_, kLnCPW / 2, 2, 512>), grid, block, 0, s, __VA_ARGS__);  \
        else                 hipLaunchKernelGGL((KERN<TX, TY, GATE_, kLnCPW, 2, 512>), grid, block, 0, s, __VA_ARGS__);  \
    } while (0)

template <typename TX, typename TY>
static int ln_fwd_t(const void *x, const float *w, const float *bias, const void *gate, void *y, float *mean, float *rstd,
                    int B, int C, int P, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, float eps, hipStream_t s, float *pool_part) {
    const int nw = ln_waves(B, C, P);
    const bool pairs = ln_pairs(C, P, xsb, xsc, gsb, gsc, x, gate, y, mean) && (reinterpret_cast<uintptr_t>(rstd) & 7u) == 0;
    const int tile = ln_tile(pairs);
    dim3 grid((P + tile - 1) / tile, B), block(64 * nw);
    const TX *xp = reinterpret_cast<const TX *>(x);
    const TY *gp = reinterpret_cast<const TY *>(gate);
    TY *yp = reinterpret_cast<TY *>(y);
    if (pool_part && !pairs) return OSS_ERR_SHAPE;   // the caller sized pool_part for 128-pixel workgroups (ln_nchw_fwd_pool_tiles)
    if (gate) OSS_LN_LAUNCH(oss_ln_nchw_fwd_kernel, true, xp, w, bias, gp, yp, mean, rstd, C, P, xsb, xsc, gsb, gsc, eps, pool_part);
    else      OSS_LN_LAUNCH(oss_ln_nchw_fwd_kernel, false, xp, w, bias, gp, yp, mean, rstd, C, P, xsb, xsc, gsb, gsc, eps, pool_part);
    return (int)hipGetLastError();
}

template <typename TX, typename TY>
static int ln_bwd_t(const void *x, const float *w, const float *bias, const void *gate, const void *dy, const float *mean,
                    const float *rstd, void *dx, void *dgate, float *dw, float *db, float *part, int B, int C, int P,
                    int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, hipStream_t s, const void *res, int64_t dgsb,
                    const float *dy_mul, const float *dy_add, float add_scale) {
    if (dy_mul != nullptr && dy_add == nullptr) return OSS_ERR_NULL;
    if (dgsb <= 0) dgsb = (int64_t)C * P;
    const int nw = ln_waves(B, C, P);
    const bool pairs = ln_pairs(C, P, xsb, xsc, gsb, gsc, x, gate, dy, dx) && dgsb % 2 == 0 &&
                       ((reinterpret_cast<uintptr_t>(dgate) | reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd) |
                         reinterpret_cast<uintptr_t>(res)) & 7u) == 0;
    const TX *rp = reinterpret_cast<const TX *>(res);
    const int tile = ln_tile(pairs);
    dim3 grid((P + tile - 1) / tile, B), block(64 * nw);
    const int nblk = grid.x * grid.y;
    const TX *xp = reinterpret_cast<const TX *>(x);
    const TY *gp = reinterpret_cast<const TY *>(gate);
    const TY *dyp = reinterpret_cast<const TY *>(dy);
    TX *dxp = reinterpret_cast<TX *>(dx);
    TY *dgp = reinterpret_cast<TY *>(dgate);
    if (gate) OSS_LN_LAUNCH(oss_ln_nchw_bwd_kernel, true, xp, w, bias, gp, dyp, mean, rstd, dxp, dgp, part, C, P, xsb, xsc, gsb, gsc, rp, dgsb, dy_mul, dy_add, add_scale);
    else      OSS_LN_LAUNCH(oss_ln_nchw_bwd_kernel, false, xp, w, bias, gp, dyp, mean, rstd, dxp, dgp, part, C, P, xsb, xsc, gsb, gsc, rp, dgsb, dy_mul, dy_add, add_scale);
    if (defer_finish())
        defer_sum(part, nblk, (size_t)2 * C, (size_t)(db ? 2 : 1) * C, dw, (size_t)C, db);
    else
        hipLaunchKernelGGL(oss_ln_nchw_bwd_finish, dim3((2 * C + 15) / 16), dim3(256), 0, s, part, dw, db, nblk, C);
    return (int)hipGetLastError();
}

#define OSS_LN_DISPATCH(FN, ...)                                                        \
    switch ((int)xt * 3 + (int)yt) {                                                    \
        case 0: return FN<float, float>(__VA_ARGS__);                                   \
        case 1: return FN<float, f16_t>(__VA_ARGS__);                                   \
        case 2: return FN<float, bf16_t>(__VA_ARGS__);                                  \
        case 3: return FN<f16_t, float>(__VA_ARGS__);                                   \
        case 4: return FN<f16_t, f16_t>(__VA_ARGS__);                                   \
        case 6: return FN<bf16_t, float>(__VA_ARGS__);                                  \
        case 8: return FN<bf16_t, bf16_t>(__VA_ARGS__);                                 \
        default: return OSS_ERR_SHAPE;                                                  \
    }

int ln_nchw_fwd(oss_dtype xt, oss_dtype yt, const void *x, const float *w, const float *bias, const void *gate, void *y,
                float *mean, float *rstd, int B, int C, int P, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, float eps,
                hipStream_t s, float *pool_part) {
    OSS_LN_DISPATCH(ln_fwd_t, x, w, bias, gate, y, mean, rstd, B, C, P, xsb, xsc, gsb, gsc, eps, s, pool_part)
}

// workgroups per image of the forward when its output sums are wanted (0: the shape does not take the 128-pixel form)
int ln_nchw_fwd_pool_tiles(int C, int P, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc) {
    const void *a = reinterpret_cast<const void *>(uintptr_t(256));
    return ln_pairs(C, P, xsb, xsc, gsb, gsc, a, a, a, a) ? (P + 127) / 128 : 0;
}

int ln_nchw_bwd(oss_dtype xt, oss_dtype yt, const void *x, const float *w, const float *bias, const void *gate,
                const void *dy, const float *mean, const float *rstd, void *dx, void *dgate, float *dw, float *db,
                float *part, int B, int C, int P, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, hipStream_t s, const void *res,
                int64_t dgsb, const float *dy_mul, const float *dy_add, float add_scale) {
    OSS_LN_DISPATCH(ln_bwd_t, x, w, bias, gate, dy, mean, rstd, dx, dgate, dw, db, part, B, C, P, xsb, xsc, gsb, gsc, s, res, dgsb, dy_mul, dy_add, add_scale)
}

}  // namespace oss
