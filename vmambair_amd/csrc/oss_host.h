// oss_host.h -- host-side declarations shared by the kernel translation units and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/vmambair_oss.h"

#include <atomic>
#include <initializer_list>

namespace oss {
struct bf16_t;
struct f16_t;

// Dynamic LDS above the 48 KiB default has to be enabled per kernel AND per device (hipFuncAttributeMaxDynamicSharedMemorySize).
// One gate per kernel instantiation: what has been enabled so far, one lock-free slot per device, so that a process driving
// several GPUs (or several host threads) never launches with a stale assumption.
constexpr size_t kMaxLdsBytes = 160 * 1024;   // gfx950: 160 KiB per workgroup
struct LdsGate {
    std::atomic<size_t> enabled[16] = {};
    int ensure(const void *kern, size_t bytes) {
        if (bytes <= 48 * 1024) return 0;
        if (bytes > kMaxLdsBytes) return OSS_ERR_SHAPE;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<size_t> &slot = enabled[dev & 15];
        if (bytes <= slot.load(std::memory_order_acquire)) return 0;
        const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
        size_t cur = slot.load(std::memory_order_relaxed);
        while (cur < bytes && !slot.compare_exchange_weak(cur, bytes, std::memory_order_release)) {}
        return 0;
    }
};

// Runtime-selected scan forms that rounds 4-5 kept behind build flags and round 6 ships in every library (an
// opt-in build that no driver run compiles rots unseen): delta computed inside the scans (`FD`; DESIGN.md 4.3:
// parity-green against the oracle, measured slower, chosen per call by oss_scan_fwd_params.dt_weight) and lane states saved by the
// forward pass for the backward (`HS`, DESIGN.md section 9: chosen per call by oss_scan_fwd_params.hs).  oss_scan_features()
// reports both bits; the constants stay so that a size-constrained build can still drop the instantiations.
#ifdef OSS_WITHOUT_FUSED_DT
constexpr bool kBuildFusedDt = false;
#else
constexpr bool kBuildFusedDt = true;
#endif
#ifdef OSS_WITHOUT_LANE_STATES
constexpr bool kBuildLaneStates = false;
#else
constexpr bool kBuildLaneStates = true;
#endif

// Time-segmented launches (oss_scan_fwd.hip: FwdSeg, oss_scan_bwd_v2.h: BwdSeg).  seg_req: -1 = heuristic, 0 / 1 = never,
// n > 1 = n segments (clamped to the number of chunks).
constexpr int kMaxSegments = 64;
int scan_pick_segments(long wgs, int n_chunks, int seg_req, double ovh);
inline size_t scan_carry_bytes(int batch, int dim, int dstate, int n_seg) {
    return sizeof(float) * 2 * (size_t)batch * dim * dstate * n_seg;
}
extern std::atomic<int> g_last_fwd_segments, g_last_bwd_segments, g_last_bwd_lane_states;
// pieces of the first (local / carry) launch per main segment: the largest piece count c <= forced (forced > 0), or with
// wgs * (n_seg - 1) * c <= 512 (heuristic), whose n_seg * c carry slots fit what the workspace queries allow.  A main segment
// of cps chunks is cut into c pieces of ceil(cps / c) chunks, the last one shorter (oss_scan_set_carry_split)
int scan_carry_split(long wgs, int n_seg, int cps, int n_chunks, int per_call = 0);
// the dt-factor gradient inside the backward's finishing launch (oss_scan_bwd_params.finish_dt_weight): whole 8 / 16-byte quads
inline bool scan_finish_dt_ok(int seqlen, int rank) { return seqlen > 0 && seqlen % 4 == 0 && rank >= 1 && rank <= 8; }
template <typename T> int scan_fwd_dispatch(const oss_scan_fwd_params &p, int variant, int seg_req, hipStream_t stream);
// one timer brackets the MAIN backward kernel, a second one the finishing kernel (oss_prof_* buckets 1 and 2)
struct LaunchTimer {
    virtual void begin(hipStream_t) = 0;
    virtual void end(hipStream_t) = 0;
    virtual void segmented() {}   // the launch about to be timed is a time-segmented one (its own profiler bucket)
    virtual ~LaunchTimer() = default;
};
template <typename T>
int scan_bwd_dispatch(const oss_scan_bwd_params &p, int variant, int seg_req, hipStream_t stream, LaunchTimer *timer,
                      LaunchTimer *finish_timer = nullptr);

// number of row tiles the backward splits a group into for `variant` (workspace sizing)
int scan_bwd_rows_per_wg(int variant);
int scan_bwd_pick_variant(int batch, int dim, int seqlen, int dstate, int n_groups);
int dwconv3x3(oss_dtype io, const void *x, const float *w, const float *bias, void *y, int B, int C, int H, int W,
              int64_t xsb, int64_t xsc, int64_t ysb, int64_t ysc, int flip, hipStream_t s, void *pre = nullptr, int act = 0);
// the convolution fused with what follows it (oss_dwconv.hip): mode 0 = silu (SS2D_1), 1 = gelu gate of the EFFN
int dwconv3x3_fused_ok(oss_dtype io, int H, int W, int nch);
int effn_fwd_ok(oss_dtype io, int D, int hidden, int H, int W);
int effn_round_weights(oss_dtype io, const float *pin, const float *pdw, const float *pout, void *w_in, float *w_dw, void *w_out, int D,
                       int hidden, hipStream_t s);
int effn_fwd(oss_dtype io, const void *x, const float *ln_w, const float *ln_b, const void *w_in, const float *w_dw, const void *w_out,
             void *out, int B, int D, int hidden, int H, int W, int64_t xsb, int64_t xsc, int64_t osb, int64_t osc, float eps,
             hipStream_t s);
int dwgate_fwd_ok(oss_dtype io, int H, int W);
int dwgate_fwd(oss_dtype io, const void *t, const float *w, const float *bias, void *out, int B, int Hd, int H, int W,
               int64_t tsb, int64_t tsc, int64_t osb, int64_t osc, hipStream_t s);
int dwconv3x3_bwd_fused(oss_dtype io, int mode, const void *x, const float *w, const float *bias, const void *dy, void *dx,
                        float *dw, float *db, float *part, int B, int C, int H, int W, int64_t xsb, int64_t xsc, int64_t gsb,
                        int64_t gsc, int64_t dsb, int64_t dsc, hipStream_t s);
// the silu form together with cross_scan_2d's two flattenings / their adjoint (x2, g2: (B, 2, C, H * W) contiguous)
int dwconv3x3_flat2_ok(oss_dtype io, int H, int W);
int dwconv3x3_silu_flat2_fwd(oss_dtype io, const void *x, const float *w, const float *bias, void *x2, int B, int C, int H, int W,
                             int64_t xsb, int64_t xsc, hipStream_t s);
int dwconv3x3_silu_flat2_bwd(oss_dtype io, const void *x, const float *w, const float *bias, const void *g2, void *dx, float *dw,
                             float *db, float *part, int B, int C, int H, int W, int64_t xsb, int64_t xsc, int64_t dsb, int64_t dsc,
                             hipStream_t s);
int dwconv3x3_wgrad(oss_dtype io, const void *x, const void *dy, float *dw, float *db, float *part, int B, int C, int H,
                    int W, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, hipStream_t s, const void *pre = nullptr,
                    void *dpre = nullptr);
// fp32 I/O on v_mfma_f32_32x32x2_f32 (oss_conv1x1_f32.hip)
int conv1x1_f32(const float *x, const float *w, const float *bias, float *y, int B, int M, int K, int P, int64_t xsb, int64_t xsk,
                int64_t wsm, int64_t wsk, hipStream_t s, const float *res);
int proj_f32_ok(int B, int D, int C, int R, int L, std::initializer_list<const void *> ptrs);
int proj_fwd_f32(const float *x2, const float *Wx, const float *Wdt, float *xdbl, float *dts, int B, int D, int C, int R, int L,
                 hipStream_t s);
int proj_dgrad_f32(const float *ddts, float *dxdbl, const float *du, const float *Wx, const float *Wdt, float *dx2, int B, int D, int C,
                   int R, int L, hipStream_t s);
size_t rows_f32_wgrad_partial_floats(int B, int G, int M, int N, int P);
int rows_f32_wgrad(const float *a, const float *bm, float *out, float *part, int B, int G, int GB, int M, int N, int P, int64_t asb,
                   int64_t asg, int64_t asm_, int64_t bsb, int64_t bsg, int64_t bsn, hipStream_t s, float *db = nullptr);
// thin dense 3x3 convolutions (oss_conv3x3_thin.hip): <= 4 channels in or out
int conv3x3_thin_ok(oss_dtype io, int Cin, int Cout, int H, int W);
int conv3x3_thin_fwd(oss_dtype io, const void *x, const float *w, const float *bias, void *y, int B, int Cin, int Cout, int H, int W,
                     int64_t xsb, int64_t xsc, int64_t ysb, int64_t ysc, hipStream_t s);
int conv3x3_thin_dgrad(oss_dtype io, const void *dy, const float *w, void *dx, int B, int Cin, int Cout, int H, int W, int64_t gsb,
                       int64_t gsc, int64_t dsb, int64_t dsc, hipStream_t s);
size_t conv3x3_thin_wgrad_partial_floats(int B, int Cin, int Cout);
int conv3x3_thin_wgrad(oss_dtype io, const void *x, const void *dy, float *dw, float *db, float *part, int B, int Cin, int Cout, int H,
                       int W, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, hipStream_t s);
int ln_nchw_fwd(oss_dtype xt, oss_dtype yt, const void *x, const float *w, const float *bias, const void *gate, void *y,
                float *mean, float *rstd, int B, int C, int P, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, float eps,
                hipStream_t s, float *pool_part = nullptr);
int ln_nchw_fwd_pool_tiles(int C, int P, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc);
size_t ln_nchw_bwd_partial_floats(int B, int C, int P);
int ln_nchw_bwd(oss_dtype xt, oss_dtype yt, const void *x, const float *w, const float *bias, const void *gate,
                const void *dy, const float *mean, const float *rstd, void *dx, void *dgate, float *dw, float *db,
                float *part, int B, int C, int P, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, hipStream_t s,
                const void *res = nullptr, int64_t dgsb = 0, const float *dy_mul = nullptr, const float *dy_add = nullptr,
                float add_scale = 1.f);
int merge4(oss_dtype io, const void *out, float *y, int B, int D, int H, int W, hipStream_t s);
// workgroup-level 1x1 convolution (oss_conv1x1_wg.hip)
int ln_conv1x1_wg(oss_dtype io, const void *x, const float *ln_w, const float *ln_b, float eps, void *n, float *mean, float *rstd,
                  const float *w, const float *bias, void *y, int B, int M, int K, int P, int64_t xsb, int64_t xsk, hipStream_t s);
size_t conv1x1_dgrad_lnbwd_partial_floats(int B, int M, int P);
int conv1x1_dgrad_lnbwd_ok(oss_dtype io, int M, int K, int P, int B);
int conv1x1_dgrad_lnbwd(oss_dtype io, const void *dy, const float *w, const void *x, const float *ln_w, int with_bias, const float *mean,
                        const float *rstd, const void *skip, void *dx, float *dlw, float *dlb, float *part, int B, int M, int K, int P,
                        int64_t xsb, int64_t xsk, hipStream_t s);
void conv1x1_set_wg(int on);
void conv1x1_wg_set_pixels(int pt);
int conv1x1_wg_ok(oss_dtype io, int M, int K, int P, int64_t xsb, int64_t xsk, const void *x, const void *y, const float *w,
                  const void *res = nullptr);
int conv1x1_wg(oss_dtype io, const void *x, const float *w, const float *bias, void *y, int B, int M, int K, int P, int64_t xsb,
               int64_t xsk, int wt, hipStream_t s, const void *res = nullptr);
int conv1x1(oss_dtype io, const void *x, const float *w, const float *bias, void *y, int B, int M, int K, int P, int64_t xsb,
            int64_t xsk, int64_t ws_m, int64_t ws_k, hipStream_t s, const void *res = nullptr);
int conv1x1_wgrad_slabs(int P);
void conv1x1_wgrad_set_tile(int mode);
void conv1x1_wgrad_set_span(int mult);
int conv1x1_wgrad(oss_dtype io, const void *dy, const void *x, float *dw, float *part, int B, int M, int N, int P,
                  int64_t gsb, int64_t gsm, int64_t xsb, int64_t xsn, hipStream_t s, int G = 1, int64_t gsg = 0, int64_t xsg = 0,
                  int Mh = 0, int64_t gs_hi = 0, float *db = nullptr);
int proj_fwd(oss_dtype io, const void *x2, const float *Wx, const float *Wdt, void *xdbl, void *dts, int B, int D, int C, int R,
             int L, hipStream_t s);
int proj_dgrad(oss_dtype io, const void *ddts, void *dxdbl, const void *du, const float *Wx, const float *Wdt, void *dx2, int B,
               int D, int C, int R, int L, hipStream_t s);
void proj_force_valu(int on);
bool proj_mfma_ok(oss_dtype io, int B, int D, int C, int R, int L);
int cross_scan2(oss_dtype it, oss_dtype ot, const void *x, void *x2, int B, int D, int H, int W, int64_t xsb, int64_t xsc,
                hipStream_t s);
int cross_merge2(oss_dtype io, const void *g2, void *dx, int B, int D, int H, int W, hipStream_t s);
size_t chan_grad_floats(int L, int dc, int Rc, int Cc);
size_t chan_bwd_scratch_floats(int B, int L, int dc, int Rc, int Cc);
int chan_fwd(const oss_chan_params &p, hipStream_t s);
int chan_bwd(const oss_chan_params &p, const float *gc, float *dpool, float *gsum, float *scratch, hipStream_t s);
int rowsum(oss_dtype io, const void *a, const void *bmul, float *out, int B, int C, int P, int64_t asb, int64_t asc, int64_t bsb,
           int64_t bsc, float alpha, hipStream_t s);
int row_affine(oss_dtype io, const void *x, const float *mul, const float *add, void *y, int B, int C, int P, int64_t xsb,
               int64_t xsc, float alpha, hipStream_t s);
int gelu_gate_fwd(oss_dtype io, const void *h, void *out, int B, size_t n, int64_t hsb, hipStream_t s);
int gelu_gate_bwd(oss_dtype io, const void *h, const void *dout, void *dh, int B, size_t n, int64_t hsb, int64_t gsb, hipStream_t s);
// oss_set_defer_finish(1): the launchers do not run their finishing kernels; they register the reduction instead
// (out[j] = sum_{k<K} src[k * stride + j], j < n0 -> dst0[j], else dst1[j - n0]) and oss_flush_finishes runs them all
// oss_set_defer_wgrad(1): conv1x1_wgrad() records its problem instead of launching; oss_flush_wgrads runs them all as ONE
// grouped launch (oss_conv1x1.hip: oss_conv1x1_wgrad_grouped_kernel)
bool defer_wgrad();
void defer_wgrad_push(const void *desc, size_t bytes, unsigned blocks);
size_t wgrad_desc_bytes();
int wgrad_grouped_launch(int io, const void *d_descs, const void *d_map, unsigned total_blocks, hipStream_t s);
void wgrad_desc_set_first_block(void *desc, unsigned first);
int wgrad_desc_io(const void *desc);
bool defer_finish();
void defer_sum(const float *src, int K, size_t stride, size_t V, float *dst0, size_t n0, float *dst1);
int sum_partials_multi(const oss_sum_chunk *chunks, int n_chunks, hipStream_t s);
int adam_ema_step(const oss_adam_chunk *chunks, int n_chunks, float *state, float lr, float beta1, float beta2, float eps,
                  float ema_decay, hipStream_t s, float weight_decay = 0.f, const float *grad_scale = nullptr);
int scan_fwd_pick_variant(int batch, int dim, int seqlen, int dstate, int n_groups, int elem_bytes);
}  // namespace oss
