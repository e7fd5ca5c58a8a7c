// oss_capi.hip -- the extern "C" surface declared in include/vmambair_oss.h.
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <vector>
#include <cstring>
#include <algorithm>
#include "oss_device.h"
#include "oss_host.h"

namespace oss {

static std::atomic<int> g_force_fwd{-1}, g_force_bwd{-1};
static std::atomic<int> g_last_fwd{-1}, g_last_bwd{-1};
static std::atomic<int> g_force_fwd_seg{-1}, g_force_bwd_seg{-1};
std::atomic<int> g_last_fwd_segments{1}, g_last_bwd_segments{1}, g_last_bwd_lane_states{0};

// Segments of a launch that has `wgs` workgroups (one per CU at a time for the wide variants) over `n_chunks` chunks when
// it is not cut in time.  Cost model in units of one chunk of one workgroup: rounds over the 256 CUs x chunks per segment x
// (1 + ovh), where ovh is the extra sweep a segmented launch pays (forward: the B side of the step again, ~0.55 of a pass;
// backward: the reverse recurrence alone, ~0.3).  Launches that already put a workgroup on every CU are left alone, and a
// larger segment count has to win by 5 % to be taken.
int scan_pick_segments(long wgs, int n_chunks, int seg_req, double ovh) {
    if (seg_req == 0 || seg_req == 1 || n_chunks < 2 || wgs <= 0) return 1;
    if (seg_req > 1) return std::min(std::min(seg_req, n_chunks), kMaxSegments);
    if (wgs >= 256) return 1;
    int best = 1;
    double best_cost = (double)n_chunks;
    for (int s = 2; s <= std::min(n_chunks, 16); ++s) {
        const int cps = (n_chunks + s - 1) / s;
        if ((n_chunks + cps - 1) / cps != s) continue;   // not a segment count this chunk count can have
        const double rounds = (double)((wgs * s + 255) / 256);
        const double cost = rounds * cps * (1.0 + ovh);
        if (cost < 0.95 * best_cost) { best = s; best_cost = cost; }
    }
    return best;
}

static std::atomic<int> g_carry_split{0};
int scan_carry_split(long wgs, int n_seg, int cps, int n_chunks, int per_call) {
    if (n_seg <= 1) return 1;
    const int forced = per_call > 0 ? per_call : g_carry_split.load();   // the call's own field wins over the test-only global
    const int max_slots = std::min(n_chunks, kMaxSegments);   // what the workspace queries size the carry area for
    int csub = 1;
    for (int c = 2; c <= cps; ++c) {
        const int ccps = (cps + c - 1) / c, pieces = (cps + ccps - 1) / ccps;   // pieces of ccps chunks, the last one shorter
        if (n_seg * pieces > max_slots) break;
        if (forced > 0 ? pieces <= forced : wgs * (n_seg - 1) * pieces <= 512) csub = pieces;
    }
    return csub;
}

// ---- variant heuristics -----------------------------------------------------------------------
// The forward/backward kernels are VALU-bound, so the cheapest variant in instructions per
// (element, state) wins as long as the launch still fills 256 CUs x 4 SIMDs with >= 2 waves.
static std::atomic<int> g_defer_finish{0};
static std::mutex g_defer_mu;
static std::vector<oss_sum_chunk> g_defer_chunks;
bool defer_finish() { return g_defer_finish.load() != 0; }
void defer_sum(const float *src, int K, size_t stride, size_t V, float *dst0, size_t n0, float *dst1) {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    for (size_t j = 0; j < V;) {   // chunks of <= 1024 outputs that do not straddle the dst0 / dst1 boundary
        const size_t lim = j < n0 ? n0 : V;
        const size_t n = std::min<size_t>(1024, lim - j);
        oss_sum_chunk c;
        c.src = src;
        c.dst = j < n0 ? dst0 + j : dst1 + (j - n0);
        c.stride = (int64_t)stride;
        c.j0 = (int)j; c.n = (int)n; c.K = K; c.reserved_ = 0;
        g_defer_chunks.push_back(c);
        j += n;
    }
}
// ---- deferred weight gradients (grouped launch) -------------------------------------------------------------------------
static std::atomic<int> g_defer_wgrad{0};
static std::vector<unsigned char> g_wgrad_descs;   // recorded descriptors, wgrad_desc_bytes() each
static std::vector<unsigned> g_wgrad_blocks;       // workgroups of each
bool defer_wgrad() { return g_defer_wgrad.load() != 0; }
void defer_wgrad_push(const void *desc, size_t bytes, unsigned blocks) {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    const unsigned char *p = reinterpret_cast<const unsigned char *>(desc);
    g_wgrad_descs.insert(g_wgrad_descs.end(), p, p + bytes);
    g_wgrad_blocks.push_back(blocks);
}

int scan_fwd_pick_variant(int batch, int dim, int seqlen, int dstate, int n_groups, int elem_bytes) {
    (void)dstate; (void)elem_bytes;
    const int rows_per_group = dim / n_groups;
    const long rows = (long)batch * dim;
    if (seqlen <= 256 || rows_per_group < 8) return (rows_per_group >= 16 && rows >= 4096) ? 2 : 4;
    // 16 lanes per row (4 rows per wave): fewest scan steps, but only rows/4 waves
    if (rows_per_group % 16 == 0 && rows / 4 >= 2048) return 2;
    if (rows_per_group % 16 == 0 && rows / 2 >= 2048) return 1;
    // long sequences whose 12-row tiles leave CUs idle (RealSR tiles at batch 1: u:(1,384,25600) is 32 workgroups; Deraining
    // level 0 at batch 4: 64): the widest workgroup, cut into time segments by scan_pick_segments until the CUs are covered
    // (u:(1,384,25600) f16 0.129 ms in 9 segments against 0.354 ms for one 8-row workgroup per tile; u:(4,192,16384) bf16
    // 0.139 in 4 segments against 0.222 -- profiles/r03_segment_sweep.txt)
    // (u:(1,768,6400) f16: 0.070 ms in 4 segments against 0.093.  At exactly 128 such workgroups -- u:(8,192,4096), u:(4,384,4096) --
    // variant 3 stayed ahead until round 5; with the local pass cut finer than the main launch (oss_scan_set_carry_split) two
    // segments of the 12-row form win there too: u:(4,384,4096) bf16 0.059 against 0.064 ms, profiles/r05_sweep_batch4_scan_variants_and_segments.txt)
    if (seqlen >= 4096 && rows_per_group % 12 == 0 && (long)batch * n_groups * (rows_per_group / 12) <= 128) return 6;
    // <= one 8-row workgroup per CU and a long sequence: 16 items per lane (half the chunk hand-overs; u:(8,192,4096)
    // bf16 0.065 ms against 0.077, profiles/r01_sweep_v4_bwd_variants.txt)
    if (seqlen >= 1024 && (long)batch * n_groups * ((rows_per_group + 7) / 8) <= 256) return 3;
    // 12-row workgroups when they give <= one workgroup per CU (u:(8,384,4096) bf16: 0.081 ms against 0.113)
    if (seqlen >= 1024 && rows_per_group >= 12 && (long)batch * n_groups * ((rows_per_group + 11) / 12) <= 256) return 6;
    return 0;
}

int scan_bwd_pick_variant(int batch, int dim, int seqlen, int dstate, int n_groups) {
    const int rows_per_group = dim / n_groups;
    // short sequences, few rows per group, dstate > 64 (the round-2 kernel keeps one lane per state): the round-1 kernel
    if (seqlen <= 256 || rows_per_group < 8 || dstate > 64) return 1;
    const long wgs = (long)batch * n_groups * ((rows_per_group + 7) / 8);
    // round-2 kernel (oss_scan_bwd_v2.h) from here on: u:(8,192,4096) 0.172 ms against 0.203 for the round-1 kernel,
    // u:(8,384,4096) 0.235 against 0.284 (profiles/r02_scan_bwd_v2_experiments.txt)
    // long sequences whose 12-row tiles leave CUs idle: 12-row workgroups cut into time segments (scan_pick_segments) instead
    // of ever smaller row tiles -- a third of the dB / dC partial tiles of the 4-row form and three waves per SIMD
    // (u:(4,192,16384) bf16: 0.328 + 0.029 ms in 4 segments against 0.590 + 0.087 for variant 13; u:(1,384,25600) f16:
    // 0.293 + 0.022 in 9 segments against 1.316 -- profiles/r03_segment_sweep.txt)
    // (u:(8,192,4096) bf16, 128 such workgroups: 0.166 + 0.015 ms in 2 segments against 0.168 + 0.020 for variant 11)
    if (seqlen >= 2048 && rows_per_group % 12 == 0 && (long)batch * n_groups * (rows_per_group / 12) < 192) return 10;
    // very few rows (Deraining level 0 at batch 4: 96 8-row workgroups for 256 CUs): 4-row workgroups spread the same waves over
    // twice the CUs, one wave per SIMD instead of two (u:(4,192,16384): 0.585 ms against 0.684)
    if (wgs <= 128 && rows_per_group % 4 == 0) return 13;
    if (wgs <= 256) return 11;
    // more rows per workgroup: fewer dB / dC partial tiles and an even load (u:(8,384,4096) bf16: 0.271 ms against
    // 0.355; u:(32,384,4096): 1.07 against 1.14)
    return rows_per_group >= 12 ? 10 : 11;
}

// ---- per-launch event timing (oss_prof_*) -----------------------------------------------------
constexpr int kProfVariants = 32;   // 0..15: the kernel variants; 16 + v: time-segmented launches of variant v
struct ProfBucket {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double ms = 0.0, bytes = 0.0, own = 0.0;
    long long launches = 0;
};
static std::mutex g_prof_mu;
static std::atomic<int> g_prof_on{0};
static ProfBucket g_prof[3][kProfVariants][3];   // forward kernel, backward main kernel, backward finishing kernel
static std::vector<hipEvent_t> g_event_pool;

static hipEvent_t prof_event() {
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}

struct ProfTimer : LaunchTimer {  // begin/end bracket exactly one kernel launch
    bool on; int which, variant, io; double bytes, own; hipEvent_t e0{}, e1{};
    ProfTimer(int which_, int variant_, int io_, double bytes_, double own_ = 0.0)
        : on(g_prof_on.load() != 0 && variant_ >= 0 && variant_ < 16), which(which_), variant(variant_),
          io(io_), bytes(bytes_), own(own_) {}
    void segmented() override { if (variant < 16) variant += 16; }
    void begin(hipStream_t s) override {
        if (!on) return;
        {
            std::lock_guard<std::mutex> lk(g_prof_mu);
            e0 = prof_event(); e1 = prof_event();
        }
        (void)hipEventRecord(e0, s);
    }
    void end(hipStream_t s) override {
        if (!on) return;
        (void)hipEventRecord(e1, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        ProfBucket &b = g_prof[which][variant][io];
        b.pending.emplace_back(e0, e1);
        b.bytes += bytes;
        b.own += own;
        b.launches += 1;
    }
};


// ---- algorithmic bytes per kernel FAMILY of the block's non-scan launches (round 6: bench.py `roofline.non_scan`) ------------
// Every entry point below adds the bytes its launch must move at minimum -- each operand read once, each result written once,
// fp32 master weights included, scratch partials NOT included -- to its family's counter while counting is on.  bench.py reads
// the counters after ONE eager training step and divides by the family's kernel time from a kernel trace of the same step.
enum Fam { FAM_CONV1X1 = 0, FAM_WGRAD, FAM_DWCONV, FAM_PROJ, FAM_LN, FAM_CHAN, FAM_MERGE, FAM_GATE, FAM_CONV3X3, FAM_FINISH, FAM_OPTIM, FAM_N };
static const char *const kFamName[FAM_N] = {"conv1x1 forward / input gradient (+ fused LayerNorm)", "1x1 / projection weight gradients",
                                            "depth-wise 3x3 (+ silu / gelu gate / flattenings)", "x_proj / dt_proj forward + input gradient",
                                            "LayerNorm (+ silu gate, pooled sums)", "channel branch (+ rowsum / row_affine)",
                                            "cross-scan / cross-merge / merge4", "gelu gate (unfused)", "thin 3x3 convolutions",
                                            "deferred partial-sum finishing", "Adam(W) + EMA"};
// substrings of the kernel names a family launches (matched by bench.py against the kernel trace)
static const char *const kFamKernels[FAM_N] = {"oss_conv1x1_pair_kernel|oss_conv1x1_pairk_kernel|oss_conv1x1_pairw_kernel|oss_conv1x1_wg_kernel|"
                                               "oss_conv1x1_dgrad_lnbwd_kernel|oss_conv1x1_f32_kernel",
                                               "oss_conv1x1_wgrad|oss_rows_f32_wgrad|oss_proj_wgrad",
                                               "oss_dwconv3x3|oss_dwgate|oss_effn",
                                               "oss_proj_fwd|oss_proj_dgrad|oss_dt_fwd|oss_dt_dgrad|oss_proj_valu",
                                               "oss_ln_nchw",
                                               "oss_chan_|oss_rowsum|oss_row_affine",
                                               "oss_merge4|oss_cross_scan2|oss_cross_merge2",
                                               "oss_gelu_gate",
                                               "oss_conv3x3_",
                                               "oss_sum_partials",
                                               "oss_adam"};
static std::atomic<int> g_fam_on{0};
static double g_fam_bytes[FAM_N];
static long long g_fam_calls[FAM_N];
static inline int esz(oss_dtype io) { return io == OSS_F32 ? 4 : 2; }
static inline void fam_count(int fam, double bytes) {
    if (!g_fam_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_fam_bytes[fam] += bytes;
    g_fam_calls[fam] += 1;
}

// bytes a launch must move at minimum
static double fwd_alg_bytes(const oss_scan_fwd_params &p, int s) {
    const double BKL = (double)p.batch * p.dim * p.seqlen, BGNL = (double)p.batch * p.n_groups * p.dstate * p.seqlen;
    const double xb = 4.0 * p.batch * p.dim * oss_scan_num_chunks(p.seqlen) * 2 * p.dstate;
    return s * (3.0 * BKL + 2.0 * BGNL) + 4.0 * ((double)p.dim * p.dstate + 2.0 * p.dim) + xb;
}
static double bwd_alg_bytes(const oss_scan_fwd_params &p, int s) {
    const double BKL = (double)p.batch * p.dim * p.seqlen, BGNL = (double)p.batch * p.n_groups * p.dstate * p.seqlen;
    const double xb = 4.0 * p.batch * p.dim * oss_scan_num_chunks(p.seqlen) * 2 * p.dstate;
    return s * (5.0 * BKL + 4.0 * BGNL) + 4.0 * (2.0 * p.dim * p.dstate + 4.0 * p.dim) + xb;
}

// the kernel's OWN algorithmic bytes: in the omni form directions k and k + K/2 share the rows of u (u_row_mod) and of dout
// (dout_row_mod), so those tensors are counted once (SURVEY.md 8d "fused-op accounting": both figures are reported)
static double fwd_own_bytes(const oss_scan_fwd_params &p, int s) {
    const double full = fwd_alg_bytes(p, s);
    if (p.u_row_mod <= 0) return full;
    return full - (double)s * p.batch * (p.dim - p.u_row_mod) * p.seqlen;
}
static double bwd_own_bytes(const oss_scan_bwd_params &q, int s) {
    const oss_scan_fwd_params &p = q.f;
    double v = bwd_alg_bytes(p, s);
    if (p.u_row_mod > 0) v -= (double)s * p.batch * (p.dim - p.u_row_mod) * p.seqlen;
    if (q.dout_row_mod > 0) v -= (double)s * p.batch * (p.dim - q.dout_row_mod) * p.seqlen;
    return v;
}

static int check_fwd(const oss_scan_fwd_params *p) {
    if (!p || !p->u || !p->delta || !p->A || !p->B || !p->C) return OSS_ERR_NULL;
    if (p->dt_weight && (!kBuildFusedDt || p->dt_rank < 1 || p->dt_rank > kMaxDtRank)) return OSS_ERR_SHAPE;
    if (p->batch < 0 || p->dim <= 0 || p->seqlen < 0 || p->dstate <= 0 || p->n_groups <= 0) return OSS_ERR_SHAPE;
    if (p->dim % p->n_groups != 0) return OSS_ERR_SHAPE;  // selective_scan.cpp:190
    if (p->dstate > OSS_MAX_DSTATE) return OSS_ERR_DSTATE;  // selective_scan.cpp:191
    if (p->rev_group_start < 0 || p->u_row_mod < 0 || (p->u_row_mod > 0 && p->dim % p->u_row_mod != 0)) return OSS_ERR_SHAPE;
    return OSS_OK;
}

}  // namespace oss

using namespace oss;

extern "C" {

int oss_scan_chunk(void) { return kScanChunk; }
int oss_scan_num_chunks(int seqlen) { return seqlen <= 0 ? 0 : (seqlen + kScanChunk - 1) / kScanChunk; }

int oss_scan_fwd(const oss_scan_fwd_params *p, oss_dtype io, oss_stream_t stream) {
    int rc = check_fwd(p);
    if (rc != OSS_OK) return rc;
    if (!p->out || !p->x) return OSS_ERR_NULL;
    if (p->batch == 0 || p->seqlen == 0) return OSS_OK;
    const int eb = io == OSS_F32 ? 4 : 2;
    // per-call fields (oss_scan_fwd_params.tune_*) win over the process-global test overrides; 0 / -1 = heuristic
    int v = p->tune_variant > 0 ? p->tune_variant - 1 : g_force_fwd.load();
    if (v < 0) v = scan_fwd_pick_variant(p->batch, p->dim, p->seqlen, p->dstate, p->n_groups, eb);
    g_last_fwd.store(v);
    g_last_fwd_segments.store(1);
    const int sr = p->tune_segments > 0 ? p->tune_segments : g_force_fwd_seg.load();
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfTimer prof(0, v, (int)io, fwd_alg_bytes(*p, eb), fwd_own_bytes(*p, eb));
    prof.begin(s);
    switch (io) {
        case OSS_F32: rc = scan_fwd_dispatch<float>(*p, v, sr, s); break;
        case OSS_F16: rc = scan_fwd_dispatch<f16_t>(*p, v, sr, s); break;
        case OSS_BF16: rc = scan_fwd_dispatch<bf16_t>(*p, v, sr, s); break;
        default: rc = OSS_ERR_SHAPE;
    }
    if (g_last_fwd_segments.load() > 1) prof.segmented();   // local pass + real pass: timed as one call, in its own bucket
    prof.end(s);
    return rc;
}

static int bwd_variant_for(int batch, int dim, int seqlen, int dstate, int n_groups, int per_call = 0) {
    int v = per_call > 0 ? per_call - 1 : g_force_bwd.load();
    if (v < 0) v = scan_bwd_pick_variant(batch, dim, seqlen, dstate, n_groups);
    return v;
}

size_t oss_scan_bwd_workspace_bytes(int batch, int dim, int seqlen, int dstate, int n_groups) {
    if (batch <= 0 || dim <= 0 || seqlen <= 0 || dstate <= 0 || n_groups <= 0 || dim % n_groups) return 0;
    // sized for the variant with the fewest rows per workgroup so that any variant fits
    const int rows_per_group = dim / n_groups;
    const int rows = scan_bwd_rows_per_wg(1);
    const size_t tiles = (size_t)(rows_per_group + rows - 1) / rows;
    // + kMaxDtRank partial rows per tile and kMaxDtRank dt-weight partials per (batch, row): the fused-delta form;
    // time-segmented launches: one weight-gradient partial per (batch, segment, row) + the reverse-carry pairs
    const size_t segs = (size_t)std::min(kMaxSegments, std::max(1, (seqlen + 511) / 512));
    const size_t floats = (size_t)batch * n_groups * tiles * (2 * dstate + kMaxDtRank) * seqlen +
                          (size_t)batch * segs * dim * (dstate + 2 + kMaxDtRank) +
                          (segs > 1 ? 2 * (size_t)batch * dim * dstate * segs : 0);
    return floats * sizeof(float);
}

size_t oss_scan_lane_state_floats(int batch, int dim, int seqlen, int dstate) {
    if (!kBuildLaneStates || batch <= 0 || dim <= 0 || seqlen <= 0 || dstate <= 0) return 0;
    return (size_t)batch * dim * dstate * lane_state_stride(seqlen);
}

size_t oss_scan_fwd_workspace_bytes(int batch, int dim, int seqlen, int dstate, int n_groups) {
    if (batch <= 0 || dim <= 0 || seqlen <= 0 || dstate <= 0 || n_groups <= 0 || dim % n_groups) return 0;
    const int segs = std::min(kMaxSegments, (seqlen + kScanChunk - 1) / kScanChunk);   // the shortest chunk any variant has
    return segs > 1 ? scan_carry_bytes(batch, dim, dstate, segs) : 0;
}

int oss_scan_bwd_finish_dt_ok(int seqlen, int dt_rank) { return scan_finish_dt_ok(seqlen, dt_rank) ? 1 : 0; }

int oss_scan_bwd(const oss_scan_bwd_params *p, oss_dtype io, oss_stream_t stream) {
    if (!p) return OSS_ERR_NULL;
    int rc = check_fwd(&p->f);
    if (rc != OSS_OK) return rc;
    if (!p->dout || !p->du || !p->dA || !p->dB || !p->dC) return OSS_ERR_NULL;
    if (!p->f.dt_weight && !p->ddelta) return OSS_ERR_NULL;
    if (p->f.dt_weight && (!p->ddt || !p->ddt_weight)) return OSS_ERR_NULL;
    const oss_scan_fwd_params &f = p->f;
    if (f.batch == 0 || f.seqlen == 0) return OSS_OK;
    if (!f.x && oss_scan_num_chunks(f.seqlen) > 1) return OSS_ERR_NULL;  // selective_scan.cpp:310
    if (p->dout_row_mod < 0 || (p->dout_row_mod > 0 && f.dim % p->dout_row_mod != 0)) return OSS_ERR_SHAPE;
    const int v = bwd_variant_for(f.batch, f.dim, f.seqlen, f.dstate, f.n_groups, p->tune_variant);
    g_last_bwd.store(v);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int eb = io == OSS_F32 ? 4 : 2;
    ProfTimer prof(1, v, (int)io, bwd_alg_bytes(f, eb), bwd_own_bytes(*p, eb));
    ProfTimer fprof(2, v, (int)io, 0.0);   // the finishing kernel of the same call
    const int sr = p->tune_segments > 0 ? p->tune_segments : g_force_bwd_seg.load();
    g_last_bwd_lane_states.store(0);
    switch (io) {
        case OSS_F32: return scan_bwd_dispatch<float>(*p, v, sr, s, &prof, &fprof);
        case OSS_F16: return scan_bwd_dispatch<f16_t>(*p, v, sr, s, &prof, &fprof);
        case OSS_BF16: return scan_bwd_dispatch<bf16_t>(*p, v, sr, s, &prof, &fprof);
    }
    return OSS_ERR_SHAPE;
}

int oss_dwconv3x3_fwd(oss_dtype io, const void *x, const float *weight, const float *bias, void *y, void *pre_silu, int batch,
                      int channels, int height, int width, int64_t xsb, int64_t xsc, int64_t ysb, int64_t ysc, int flip,
                      oss_stream_t stream) {
    fam_count(FAM_DWCONV, (double)batch * channels * height * width * esz(io) * (pre_silu ? 3.0 : 2.0) + 40.0 * channels);
    if (!x || !weight || !y) return OSS_ERR_NULL;
    if (batch <= 0 || channels <= 0 || height <= 0 || width <= 0 || channels > 65535 || batch > 65535) return OSS_ERR_SHAPE;
    return dwconv3x3(io, x, weight, bias, y, batch, channels, height, width, xsb, xsc, ysb, ysc, flip,
                     reinterpret_cast<hipStream_t>(stream), pre_silu, pre_silu != nullptr);
}

int oss_dwconv3x3_fused_ok(oss_dtype io, int height, int width, int channels_per_workgroup) {
    return dwconv3x3_fused_ok(io, height, width, channels_per_workgroup);
}

int oss_dwconv3x3_silu_fwd(oss_dtype io, const void *x, const float *weight, const float *bias, void *y, int batch, int channels,
                           int height, int width, int64_t xsb, int64_t xsc, int64_t ysb, int64_t ysc, oss_stream_t stream) {
    fam_count(FAM_DWCONV, (double)batch * channels * height * width * esz(io) * 2.0 + 40.0 * channels);
    if (!x || !weight || !y) return OSS_ERR_NULL;
    if (batch <= 0 || channels <= 0 || height <= 0 || width <= 0 || channels > 65535 || batch > 65535) return OSS_ERR_SHAPE;
    return dwconv3x3(io, x, weight, bias, y, batch, channels, height, width, xsb, xsc, ysb, ysc, 0,
                     reinterpret_cast<hipStream_t>(stream), nullptr, 1);
}

int oss_dwconv3x3_silu_bwd(oss_dtype io, const void *x, const float *weight, const float *bias, const void *dy, void *dx,
                           float *dweight, float *dbias, float *partials, int batch, int channels, int height, int width,
                           int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, int64_t dsb, int64_t dsc, oss_stream_t stream) {
    fam_count(FAM_DWCONV, (double)batch * channels * height * width * esz(io) * 3.0 + 80.0 * channels);
    if (!x || !weight || !dy || !dx || !dweight || !partials) return OSS_ERR_NULL;
    if (batch <= 0 || channels <= 0 || height <= 0 || width <= 0 || channels > 65535 || batch > 65535) return OSS_ERR_SHAPE;
    return dwconv3x3_bwd_fused(io, 0, x, weight, bias, dy, dx, dweight, dbias, partials, batch, channels, height, width, xsb, xsc,
                               gsb, gsc, dsb, dsc, reinterpret_cast<hipStream_t>(stream));
}

int oss_dwconv3x3_flat2_ok(oss_dtype io, int height, int width) { return dwconv3x3_flat2_ok(io, height, width); }

int oss_dwconv3x3_silu_flat2_fwd(oss_dtype io, const void *x, const float *weight, const float *bias, void *x2, int batch, int channels,
                                 int height, int width, int64_t xsb, int64_t xsc, oss_stream_t stream) {
    fam_count(FAM_DWCONV, (double)batch * channels * height * width * esz(io) * 3.0 + 40.0 * channels);
    if (!x || !weight || !x2) return OSS_ERR_NULL;
    if (batch <= 0 || channels <= 0 || height <= 0 || width <= 0 || channels > 65535 || batch > 65535) return OSS_ERR_SHAPE;
    return dwconv3x3_silu_flat2_fwd(io, x, weight, bias, x2, batch, channels, height, width, xsb, xsc, reinterpret_cast<hipStream_t>(stream));
}

int oss_dwconv3x3_silu_flat2_bwd(oss_dtype io, const void *x, const float *weight, const float *bias, const void *g2, void *dx,
                                 float *dweight, float *dbias, float *partials, int batch, int channels, int height, int width,
                                 int64_t xsb, int64_t xsc, int64_t dsb, int64_t dsc, oss_stream_t stream) {
    fam_count(FAM_DWCONV, (double)batch * channels * height * width * esz(io) * 4.0 + 80.0 * channels);
    if (!x || !weight || !g2 || !dx || !dweight || !partials) return OSS_ERR_NULL;
    if (batch <= 0 || channels <= 0 || height <= 0 || width <= 0 || channels > 65535 || batch > 65535) return OSS_ERR_SHAPE;
    return dwconv3x3_silu_flat2_bwd(io, x, weight, bias, g2, dx, dweight, dbias, partials, batch, channels, height, width, xsb, xsc, dsb,
                                    dsc, reinterpret_cast<hipStream_t>(stream));
}

int oss_effn_fwd_ok(oss_dtype io, int channels, int hidden, int height, int width) { return effn_fwd_ok(io, channels, hidden, height, width); }

int oss_effn_round_weights(oss_dtype io, const float *project_in, const float *dwconv, const float *project_out, void *w_in, float *w_dw,
                           void *w_out, int channels, int hidden, oss_stream_t stream) {
    if (!project_in || !dwconv || !project_out || !w_in || !w_dw || !w_out) return OSS_ERR_NULL;
    return effn_round_weights(io, project_in, dwconv, project_out, w_in, w_dw, w_out, channels, hidden, reinterpret_cast<hipStream_t>(stream));
}

int oss_effn_fwd(oss_dtype io, const void *x, const float *norm_weight, const float *norm_bias, const void *w_in, const float *w_dw,
                 const void *w_out, void *out, int batch, int channels, int hidden, int height, int width, int64_t xsb, int64_t xsc,
                 int64_t osb, int64_t osc, float eps, oss_stream_t stream) {
    fam_count(FAM_DWCONV, (double)batch * channels * height * width * esz(io) * 3.0 + (double)hidden * (3.0 * channels * esz(io) + 72.0));
    if (!x || !norm_weight || !w_in || !w_dw || !w_out || !out) return OSS_ERR_NULL;
    if (batch <= 0 || channels <= 0 || hidden <= 0 || height <= 0 || width <= 0) return OSS_ERR_SHAPE;
    return effn_fwd(io, x, norm_weight, norm_bias, w_in, w_dw, w_out, out, batch, channels, hidden, height, width, xsb, xsc, osb, osc, eps,
                    reinterpret_cast<hipStream_t>(stream));
}

int oss_dwgate_fwd_ok(oss_dtype io, int height, int width) { return dwgate_fwd_ok(io, height, width); }

int oss_dwgate_fwd(oss_dtype io, const void *t, const float *weight, const float *bias, void *out, int batch, int hidden,
                   int height, int width, int64_t tsb, int64_t tsc, int64_t osb, int64_t osc, oss_stream_t stream) {
    fam_count(FAM_DWCONV, (double)batch * hidden * height * width * esz(io) * 3.0 + 80.0 * hidden);
    if (!t || !weight || !out) return OSS_ERR_NULL;
    if (batch <= 0 || hidden <= 0 || height <= 0 || width <= 0 || hidden > 32767 || batch > 65535) return OSS_ERR_SHAPE;
    return dwgate_fwd(io, t, weight, bias, out, batch, hidden, height, width, tsb, tsc, osb, osc, reinterpret_cast<hipStream_t>(stream));
}

int oss_dwgate_bwd(oss_dtype io, const void *t, const float *weight, const float *bias, const void *dout, void *dt,
                   float *dweight, float *dbias, float *partials, int batch, int hidden, int height, int width, int64_t tsb,
                   int64_t tsc, int64_t gsb, int64_t gsc, int64_t dsb, int64_t dsc, oss_stream_t stream) {
    fam_count(FAM_DWCONV, (double)batch * hidden * height * width * esz(io) * 5.0 + 160.0 * hidden);
    if (!t || !weight || !dout || !dt || !dweight || !partials) return OSS_ERR_NULL;
    if (batch <= 0 || hidden <= 0 || height <= 0 || width <= 0 || hidden > 32767 || batch > 65535) return OSS_ERR_SHAPE;
    return dwconv3x3_bwd_fused(io, 1, t, weight, bias, dout, dt, dweight, dbias, partials, batch, 2 * hidden, height, width, tsb,
                               tsc, gsb, gsc, dsb, dsc, reinterpret_cast<hipStream_t>(stream));
}

int oss_dwconv3x3_wgrad(oss_dtype io, const void *x, const void *dy, float *dweight, float *dbias, float *partials,
                        const void *pre_silu, void *dpre, int batch, int channels, int height, int width, int64_t xsb,
                        int64_t xsc, int64_t gsb, int64_t gsc, oss_stream_t stream) {
    fam_count(FAM_DWCONV, (double)batch * channels * height * width * esz(io) * (pre_silu ? 4.0 : 2.0));
    if (!x || !dy || !dweight || !partials) return OSS_ERR_NULL;
    if (batch <= 0 || channels <= 0 || height <= 0 || width <= 0 || batch > 65535) return OSS_ERR_SHAPE;
    return dwconv3x3_wgrad(io, x, dy, dweight, dbias, partials, batch, channels, height, width, xsb, xsc, gsb, gsc,
                           reinterpret_cast<hipStream_t>(stream), pre_silu, dpre);
}

int oss_conv1x1_fwd(oss_dtype io, const void *x, const float *weight, const float *bias, const void *residual, void *y,
                    int batch, int cout, int cin, int pixels, int64_t xsb, int64_t xsc, oss_stream_t stream) {
    fam_count(FAM_CONV1X1, (double)batch * pixels * esz(io) * ((double)cin + cout + (residual ? cout : 0)) + 4.0 * cin * cout);
    if (!x || !weight || !y) return OSS_ERR_NULL;
    if (batch <= 0 || cout <= 0 || cin <= 0 || pixels <= 0 || batch > 65535 || cout > 32 * 65535) return OSS_ERR_SHAPE;
    return conv1x1(io, x, weight, bias, y, batch, cout, cin, pixels, xsb, xsc, cin, 1, reinterpret_cast<hipStream_t>(stream),
                   residual);
}

int oss_conv1x1_dgrad(oss_dtype io, const void *dy, const float *weight, void *dx, int batch, int cout, int cin, int pixels,
                      int64_t gsb, int64_t gsc, oss_stream_t stream) {
    fam_count(FAM_CONV1X1, (double)batch * pixels * esz(io) * ((double)cin + cout) + 4.0 * cin * cout);
    if (!dy || !weight || !dx) return OSS_ERR_NULL;
    if (batch <= 0 || cout <= 0 || cin <= 0 || pixels <= 0 || batch > 65535) return OSS_ERR_SHAPE;
    // dx[ci] = sum_co W[co][ci] dy[co]: the same GEMM with the weights read transposed
    return conv1x1(io, dy, weight, nullptr, dx, batch, cin, cout, pixels, gsb, gsc, 1, cin, reinterpret_cast<hipStream_t>(stream));
}

size_t oss_conv1x1_wgrad_partial_floats(int batch, int cout, int cin, int pixels) {
    if (batch <= 0 || cout <= 0 || cin <= 0 || pixels <= 0) return 0;
    return (size_t)batch * conv1x1_wgrad_slabs(pixels) * cout * (cin + 1);  // + 1: the dbias column
}

int oss_conv1x1_wgrad(oss_dtype io, const void *dy, const void *x, float *dweight, float *dbias, float *partials, int batch,
                      int cout, int cin, int pixels, int64_t gsb, int64_t gsc, int64_t xsb, int64_t xsc, oss_stream_t stream) {
    fam_count(FAM_WGRAD, (double)batch * pixels * esz(io) * ((double)cin + cout) + 4.0 * cin * cout);
    if (!dy || !x || !dweight || !partials) return OSS_ERR_NULL;
    if (batch <= 0 || cout <= 0 || cin <= 0 || pixels <= 0 || batch > 65535) return OSS_ERR_SHAPE;
    return conv1x1_wgrad(io, dy, x, dweight, partials, batch, cout, cin, pixels, gsb, gsc, xsb, xsc,
                         reinterpret_cast<hipStream_t>(stream), 1, 0, 0, 0, 0, dbias);
}

size_t oss_proj_wgrad_partial_floats(int batch, int D, int C, int R, int seqlen) {
    if (batch <= 0 || D <= 0 || C <= 0 || R <= 0 || seqlen <= 0) return 0;
    // the two weight gradients keep separate scratch regions (their finishing may be deferred)
    const size_t slabs = (size_t)batch * conv1x1_wgrad_slabs(seqlen);
    const size_t a = slabs * 2 * (2 * (size_t)C) * D, b = slabs * 4 * (size_t)D * R;
    return a + b;
}

int oss_proj_fwd(oss_dtype io, const void *x2, const float *x_proj_weight, const float *dt_projs_weight, void *xdbl, void *dts,
                 int batch, int D, int C, int R, int seqlen, oss_stream_t stream) {
    fam_count(FAM_PROJ, (double)batch * seqlen * esz(io) * (2.0 * D + 4.0 * C + (dts ? 4.0 * R + 4.0 * D : 0.0)) + 16.0 * C * D + 16.0 * D * R);
    if (!x2 || !x_proj_weight || !dt_projs_weight || !xdbl) return OSS_ERR_NULL;   // dts == NULL: fused-delta form
    if (batch <= 0 || D <= 0 || seqlen <= 0 || R <= 0 || C <= R) return OSS_ERR_SHAPE;
    return proj_fwd(io, x2, x_proj_weight, dt_projs_weight, xdbl, dts, batch, D, C, R, seqlen, reinterpret_cast<hipStream_t>(stream));
}

int oss_proj_dgrad(oss_dtype io, const void *ddts, void *dxdbl, const void *du, const float *x_proj_weight,
                   const float *dt_projs_weight, void *dx2, int batch, int D, int C, int R, int seqlen, oss_stream_t stream) {
    fam_count(FAM_PROJ, (double)batch * seqlen * esz(io) * ((ddts ? 4.0 * D + 4.0 * R : 0.0) + 4.0 * C + (du ? 4.0 * D : 0.0) + 2.0 * D) + 16.0 * C * D);
    if (!dxdbl || !x_proj_weight || !dt_projs_weight || !dx2) return OSS_ERR_NULL;   // ddts == NULL: fused-delta form
    if (batch <= 0 || D <= 0 || seqlen <= 0 || R <= 0 || C <= R) return OSS_ERR_SHAPE;
    return proj_dgrad(io, ddts, dxdbl, du, x_proj_weight, dt_projs_weight, dx2, batch, D, C, R, seqlen,
                      reinterpret_cast<hipStream_t>(stream));
}

void oss_proj_set_path(int force_vector_alu) { proj_force_valu(force_vector_alu); }
int oss_proj_rows_optional_ok(oss_dtype io, int batch, int D, int C, int R, int seqlen) {
    return proj_mfma_ok(io, batch, D, C, R, seqlen) ? 1 : 0;
}
int oss_scan_fused_dt_ok(oss_dtype io, int batch, int D, int C, int R, int dstate, int seqlen) {
    return kBuildFusedDt && proj_mfma_ok(io, batch, D, C, R, seqlen) && R >= 1 && R <= kMaxDtRank && dstate <= 64 && seqlen >= 512;
}
int oss_conv1x1_wg(oss_dtype io, const void *x, const float *weight, const float *bias, void *y, int batch, int cout, int cin,
                   int pixels, int64_t xsb, int64_t xsc, int transposed_weight, oss_stream_t stream) {
    fam_count(FAM_CONV1X1, (double)batch * pixels * esz(io) * ((double)cin + cout) + 4.0 * cin * cout);
    if (!x || !weight || !y) return OSS_ERR_NULL;
    if (batch <= 0 || batch > 65535) return OSS_ERR_SHAPE;
    return conv1x1_wg(io, x, weight, bias, y, batch, cout, cin, pixels, xsb, xsc, transposed_weight, reinterpret_cast<hipStream_t>(stream));
}

int oss_ln_conv1x1_ok(oss_dtype io, int cout, int cin, int pixels) {
    static const float dummy = 0.f;   // shape / dtype rules only: an aligned stand-in for the pointers
    const void *a = reinterpret_cast<const void *>(uintptr_t(256));
    (void)dummy;
    return conv1x1_wg_ok(io, cout, cin, pixels, (int64_t)cin * pixels, pixels, a, a, reinterpret_cast<const float *>(a), nullptr);
}

int oss_ln_conv1x1_fwd(oss_dtype io, const void *x, const float *ln_weight, const float *ln_bias, float eps, void *n, float *mean,
                       float *rstd, const float *weight, const float *bias, void *y, int batch, int cout, int cin, int pixels,
                       int64_t xsb, int64_t xsc, oss_stream_t stream) {
    fam_count(FAM_CONV1X1, (double)batch * pixels * (esz(io) * (2.0 * cin + cout) + 8.0) + 4.0 * cin * cout);
    if (!x || !ln_weight || !n || !mean || !rstd || !weight || !y) return OSS_ERR_NULL;
    if (batch <= 0 || batch > 65535) return OSS_ERR_SHAPE;
    return ln_conv1x1_wg(io, x, ln_weight, ln_bias, eps, n, mean, rstd, weight, bias, y, batch, cout, cin, pixels, xsb, xsc,
                         reinterpret_cast<hipStream_t>(stream));
}

int oss_conv1x1_dgrad_ln_bwd_ok(oss_dtype io, int cout, int cin, int pixels, int batch) {
    return conv1x1_dgrad_lnbwd_ok(io, cin, cout, pixels, batch);
}
size_t oss_conv1x1_dgrad_ln_bwd_partial_floats(int batch, int cin, int pixels) {
    if (batch <= 0 || cin <= 0 || pixels <= 0) return 0;
    return conv1x1_dgrad_lnbwd_partial_floats(batch, cin, pixels);
}
int oss_conv1x1_dgrad_ln_bwd(oss_dtype io, const void *dy, const float *weight, const void *x, const float *ln_weight, int ln_has_bias,
                             const float *mean, const float *rstd, const void *skip_grad, void *dx, float *dln_weight, float *dln_bias,
                             float *partials, int batch, int cout, int cin, int pixels, int64_t gsb, int64_t gsc, oss_stream_t stream) {
    fam_count(FAM_CONV1X1, (double)batch * pixels * (esz(io) * ((double)cout + 2.0 * cin + (skip_grad ? cin : 0)) + 8.0) + 4.0 * cin * cout);
    if (!dy || !weight || !x || !ln_weight || !mean || !rstd || !dx || !dln_weight || !partials) return OSS_ERR_NULL;
    if (batch <= 0 || batch > 65535) return OSS_ERR_SHAPE;
    return conv1x1_dgrad_lnbwd(io, dy, weight, x, ln_weight, ln_has_bias, mean, rstd, skip_grad, dx, dln_weight, dln_bias, partials, batch,
                               cin, cout, pixels, gsb, gsc, reinterpret_cast<hipStream_t>(stream));
}

void oss_conv1x1_set_wg(int on, int pixels) { conv1x1_set_wg(on); conv1x1_wg_set_pixels(pixels); }
void oss_conv1x1_wgrad_set_tile(int mode) { conv1x1_wgrad_set_tile(mode); }
void oss_conv1x1_wgrad_set_span(int mult) { conv1x1_wgrad_set_span(mult); }

int oss_proj_wgrad(oss_dtype io, const void *x2, const void *xdbl, const void *dxdbl, const void *ddts, float *dx_proj_weight,
                   float *ddt_projs_weight, float *partials, int batch, int D, int C, int R, int seqlen, oss_stream_t stream) {
    fam_count(FAM_WGRAD, (double)batch * seqlen * esz(io) * (2.0 * D + 8.0 * C + (ddts ? 4.0 * D : 0.0)));
    if (!x2 || !xdbl || !dxdbl || !dx_proj_weight || !partials) return OSS_ERR_NULL;
    if (ddts && !ddt_projs_weight) return OSS_ERR_NULL;   // ddts == NULL: the scan backward produced ddt_projs_weight itself
    if (batch <= 0 || D <= 0 || seqlen <= 0 || R <= 0 || C <= R) return OSS_ERR_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t L = seqlen;
    if (io == OSS_F32) {   // fp32 I/O: both products on the fp32 matrix-core kernel (oss_conv1x1_f32.hip), same scratch regions
        // dx_proj_weight[k][c][d] = sum_{b,l} dxdbl[b, k, c, l] x2[b, k % 2, d, l]
        int e = rows_f32_wgrad(reinterpret_cast<const float *>(dxdbl), reinterpret_cast<const float *>(x2), dx_proj_weight, partials, batch,
                               4, 2, C, D, seqlen, 4 * C * L, C * L, L, 2 * D * L, D * L, L, s);
        if (e || !ddts) return e;
        // ddt_projs_weight[k][d][r] = sum_{b,l} ddts[b, k, d, l] xdbl[b, k, r, l]
        float *part2 = partials + (size_t)batch * conv1x1_wgrad_slabs(seqlen) * 2 * (2 * (size_t)C) * D;
        return rows_f32_wgrad(reinterpret_cast<const float *>(ddts), reinterpret_cast<const float *>(xdbl), ddt_projs_weight, part2, batch, 4,
                              4, D, R, seqlen, 4 * D * L, D * L, L, 4 * C * L, C * L, L, s);
    }
    // x_proj_weight: one problem per flattening j; its 2C rows are the rows of directions j and j + 2 of dxdbl
    int e = conv1x1_wgrad(io, dxdbl, x2, dx_proj_weight, partials, batch, 2 * C, D, seqlen, 4 * C * L, L, 2 * D * L, L, s,
                          /*G*/ 2, /*gsg*/ C * L, /*xsg*/ D * L, /*Mh*/ C, /*gs_hi*/ 2 * C * L);
    if (e || !ddts) return e;
    // dt_projs_weight: one problem per direction k: ddts[:, k] (D rows) x the dt rows of xdbl[:, k] (R rows)
    float *part2 = partials + (size_t)batch * conv1x1_wgrad_slabs(seqlen) * 2 * (2 * (size_t)C) * D;
    return conv1x1_wgrad(io, ddts, xdbl, ddt_projs_weight, part2, batch, D, R, seqlen, 4 * D * L, L, 4 * C * L, L, s,
                         /*G*/ 4, /*gsg*/ D * L, /*xsg*/ C * L, /*Mh*/ D, 0);
}

int oss_cross_scan2(oss_dtype in_type, oss_dtype out_type, const void *x, void *x2, int batch, int D, int height, int width,
                    int64_t x_batch_stride, int64_t x_channel_stride, oss_stream_t stream) {
    fam_count(FAM_MERGE, (double)batch * D * height * width * (esz(in_type) + 2.0 * esz(out_type)));
    if (!x || !x2) return OSS_ERR_NULL;
    if (batch <= 0 || D <= 0 || height <= 0 || width <= 0) return OSS_ERR_SHAPE;
    return cross_scan2(in_type, out_type, x, x2, batch, D, height, width, x_batch_stride, x_channel_stride,
                       reinterpret_cast<hipStream_t>(stream));
}

int oss_cross_merge2(oss_dtype io, const void *g2, void *dx, int batch, int D, int height, int width, oss_stream_t stream) {
    fam_count(FAM_MERGE, (double)batch * D * height * width * esz(io) * 3.0);
    if (!g2 || !dx) return OSS_ERR_NULL;
    if (batch <= 0 || D <= 0 || height <= 0 || width <= 0) return OSS_ERR_SHAPE;
    return cross_merge2(io, g2, dx, batch, D, height, width, reinterpret_cast<hipStream_t>(stream));
}

int oss_chan_fwd(const oss_chan_params *p, oss_stream_t stream) {
    fam_count(FAM_CHAN, p ? 4.0 * p->B * ((double)p->L * (2.0 * p->Cc + 2.0 * p->dc * 18.0 + 3.0) + (p->pool_part ? (double)p->n_part * p->L : p->L)) : 0.0);
    if (!p) return OSS_ERR_NULL;
    return chan_fwd(*p, reinterpret_cast<hipStream_t>(stream));
}

size_t oss_chan_grad_floats(int L, int dc, int Rc, int Cc) { return chan_grad_floats(L, dc, Rc, Cc); }
size_t oss_chan_bwd_scratch_floats(int B, int L, int dc, int Rc, int Cc) { return chan_bwd_scratch_floats(B, L, dc, Rc, Cc); }

int oss_chan_bwd(const oss_chan_params *p, const float *gc, float *dpooled, float *grads, float *scratch, oss_stream_t stream) {
    fam_count(FAM_CHAN, p ? 4.0 * p->B * ((double)p->L * (2.0 * p->Cc + 2.0 * p->dc * 18.0 + 5.0)) : 0.0);
    if (!p) return OSS_ERR_NULL;
    return chan_bwd(*p, gc, dpooled, grads, scratch, reinterpret_cast<hipStream_t>(stream));
}

int oss_rowsum(oss_dtype io, const void *a, const void *bmul, float *out, int batch, int channels, int pixels, int64_t asb,
               int64_t asc, int64_t bsb, int64_t bsc, float alpha, oss_stream_t stream) {
    fam_count(FAM_CHAN, (double)batch * channels * pixels * esz(io) * (bmul ? 2.0 : 1.0));
    if (!a || !out) return OSS_ERR_NULL;
    if (batch <= 0 || channels <= 0 || pixels <= 0) return OSS_ERR_SHAPE;
    return rowsum(io, a, bmul, out, batch, channels, pixels, asb, asc, bsb, bsc, alpha, reinterpret_cast<hipStream_t>(stream));
}

int oss_row_affine(oss_dtype io, const void *x, const float *mul, const float *add, void *y, int batch, int channels, int pixels,
                   int64_t xsb, int64_t xsc, float alpha, oss_stream_t stream) {
    fam_count(FAM_CHAN, (double)batch * channels * pixels * esz(io) * 2.0);
    if (!x || !y) return OSS_ERR_NULL;
    if (batch <= 0 || channels <= 0 || pixels <= 0) return OSS_ERR_SHAPE;
    return row_affine(io, x, mul, add, y, batch, channels, pixels, xsb, xsc, alpha, reinterpret_cast<hipStream_t>(stream));
}

int oss_gelu_gate_fwd(oss_dtype io, const void *h, void *out, int batch, size_t half_elems, int64_t h_batch_stride,
                      oss_stream_t stream) {
    fam_count(FAM_GATE, (double)batch * half_elems * esz(io) * 3.0);
    if (!h || !out) return OSS_ERR_NULL;
    if (batch <= 0 || half_elems == 0) return OSS_ERR_SHAPE;
    return gelu_gate_fwd(io, h, out, batch, half_elems, h_batch_stride, reinterpret_cast<hipStream_t>(stream));
}

int oss_gelu_gate_bwd(oss_dtype io, const void *h, const void *dout, void *dh, int batch, size_t half_elems,
                      int64_t h_batch_stride, int64_t dout_batch_stride, oss_stream_t stream) {
    fam_count(FAM_GATE, (double)batch * half_elems * esz(io) * 5.0);
    if (!h || !dout || !dh) return OSS_ERR_NULL;
    if (batch <= 0 || half_elems == 0) return OSS_ERR_SHAPE;
    return gelu_gate_bwd(io, h, dout, dh, batch, half_elems, h_batch_stride, dout_batch_stride, reinterpret_cast<hipStream_t>(stream));
}

int oss_conv3x3_thin_ok(oss_dtype io, int cin, int cout, int height, int width) { return conv3x3_thin_ok(io, cin, cout, height, width); }
int oss_conv3x3_thin_fwd(oss_dtype io, const void *x, const float *weight, const float *bias, void *y, int batch, int cin, int cout,
                         int height, int width, int64_t xsb, int64_t xsc, int64_t ysb, int64_t ysc, oss_stream_t stream) {
    fam_count(FAM_CONV3X3, (double)batch * height * width * esz(io) * ((double)cin + cout));
    if (!x || !weight || !y) return OSS_ERR_NULL;
    return conv3x3_thin_fwd(io, x, weight, bias, y, batch, cin, cout, height, width, xsb, xsc, ysb, ysc, reinterpret_cast<hipStream_t>(stream));
}
int oss_conv3x3_thin_dgrad(oss_dtype io, const void *dy, const float *weight, void *dx, int batch, int cin, int cout, int height,
                           int width, int64_t gsb, int64_t gsc, int64_t dsb, int64_t dsc, oss_stream_t stream) {
    fam_count(FAM_CONV3X3, (double)batch * height * width * esz(io) * ((double)cin + cout));
    if (!dy || !weight || !dx) return OSS_ERR_NULL;
    return conv3x3_thin_dgrad(io, dy, weight, dx, batch, cin, cout, height, width, gsb, gsc, dsb, dsc, reinterpret_cast<hipStream_t>(stream));
}
size_t oss_conv3x3_thin_wgrad_partial_floats(int batch, int cin, int cout) {
    return (batch <= 0 || cin <= 0 || cout <= 0) ? 0 : conv3x3_thin_wgrad_partial_floats(batch, cin, cout);
}
int oss_conv3x3_thin_wgrad(oss_dtype io, const void *x, const void *dy, float *dweight, float *dbias, float *partial, int batch, int cin,
                           int cout, int height, int width, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc, oss_stream_t stream) {
    fam_count(FAM_CONV3X3, (double)batch * height * width * esz(io) * ((double)cin + cout));
    if (!x || !dy || !dweight || !partial) return OSS_ERR_NULL;
    return conv3x3_thin_wgrad(io, x, dy, dweight, dbias, partial, batch, cin, cout, height, width, xsb, xsc, gsb, gsc,
                              reinterpret_cast<hipStream_t>(stream));
}

void oss_set_defer_wgrad(int on) {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    g_defer_wgrad.store(on ? 1 : 0);
    g_wgrad_descs.clear();
    g_wgrad_blocks.clear();
}
size_t oss_deferred_wgrads(void) {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    return g_wgrad_blocks.size();
}
size_t oss_deferred_wgrad_table_bytes(void) {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    size_t blocks = 0;
    for (unsigned b : g_wgrad_blocks) blocks += b;
    // descriptors (padded to 256 bytes) + one 16-bit problem index per workgroup
    return ((g_wgrad_descs.size() + 255) & ~(size_t)255) + 2 * blocks;
}
// max_products: flush only the FIRST that many recorded products (recording order = the backward's order, last layers first) and
// keep the rest recorded -- the bucketed gradient exchange (train_graph.py) finishes the gradients of the layers whose backward
// ran first, hands them to the all-reduce and flushes the next group while that runs.  0 = all.
int oss_flush_wgrads_n(void *host_table, void *device_table, size_t capacity_bytes, size_t max_products, oss_stream_t stream) {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    const size_t n_all = g_wgrad_blocks.size();
    const size_t n = max_products > 0 ? std::min(n_all, max_products) : n_all;
    if (n == 0) return 0;
    if (!host_table || !device_table) return OSS_ERR_NULL;
    if (n > 65535) return OSS_ERR_SHAPE;
    const size_t db = wgrad_desc_bytes();
    const size_t desc_bytes = (n * db + 255) & ~(size_t)255;
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) total += g_wgrad_blocks[i];
    if (desc_bytes + 2 * total > capacity_bytes) return OSS_ERR_WORKSPACE;
    if (total > 0x7fffffffu) return OSS_ERR_SHAPE;
    const int io = wgrad_desc_io(g_wgrad_descs.data());
    unsigned first = 0;
    uint16_t *map = reinterpret_cast<uint16_t *>(reinterpret_cast<unsigned char *>(host_table) + desc_bytes);
    for (size_t i = 0; i < n; ++i) {
        if (wgrad_desc_io(g_wgrad_descs.data() + i * db) != io) return OSS_ERR_SHAPE;   // one I/O type per flush
        wgrad_desc_set_first_block(g_wgrad_descs.data() + i * db, first);
        for (unsigned k = 0; k < g_wgrad_blocks[i]; ++k) map[first + k] = (uint16_t)i;
        first += g_wgrad_blocks[i];
    }
    std::memcpy(host_table, g_wgrad_descs.data(), n * db);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const hipError_t e = hipMemcpyAsync(device_table, host_table, desc_bytes + 2 * total, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return (int)e;
    g_wgrad_descs.erase(g_wgrad_descs.begin(), g_wgrad_descs.begin() + n * db);
    g_wgrad_blocks.erase(g_wgrad_blocks.begin(), g_wgrad_blocks.begin() + n);
    return wgrad_grouped_launch(io, device_table, reinterpret_cast<unsigned char *>(device_table) + desc_bytes, (unsigned)total, s);
}
int oss_flush_wgrads(void *host_table, void *device_table, size_t capacity_bytes, oss_stream_t stream) {
    return oss_flush_wgrads_n(host_table, device_table, capacity_bytes, 0, stream);
}

void oss_set_defer_finish(int on) {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    g_defer_finish.store(on ? 1 : 0);
    g_defer_chunks.clear();
}
size_t oss_deferred_chunks(void) {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    return g_defer_chunks.size();
}
// max_chunks: only the FIRST that many registered chunks (registration order = the backward's order); 0 = all
int oss_flush_finishes_n(void *host_table, void *device_table, size_t capacity_chunks, size_t max_chunks, oss_stream_t stream) {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    const size_t n_all = g_defer_chunks.size();
    const size_t n = max_chunks > 0 ? std::min(n_all, max_chunks) : n_all;
    if (n == 0) return 0;
    if (!host_table || !device_table) return OSS_ERR_NULL;
    if (n > capacity_chunks) return OSS_ERR_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (g_fam_on.load(std::memory_order_relaxed)) {
        double by = 0.0;
        for (size_t i = 0; i < n; ++i) by += 4.0 * g_defer_chunks[i].n * (g_defer_chunks[i].K + 1.0);
        fam_count(FAM_FINISH, by);
    }
    std::memcpy(host_table, g_defer_chunks.data(), n * sizeof(oss_sum_chunk));
    hipError_t e = hipMemcpyAsync(device_table, host_table, n * sizeof(oss_sum_chunk), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return (int)e;
    g_defer_chunks.erase(g_defer_chunks.begin(), g_defer_chunks.begin() + n);
    return sum_partials_multi(reinterpret_cast<const oss_sum_chunk *>(device_table), (int)n, s);
}
int oss_flush_finishes(void *host_table, void *device_table, size_t capacity_chunks, oss_stream_t stream) {
    return oss_flush_finishes_n(host_table, device_table, capacity_chunks, 0, stream);
}

int oss_adam_ema_step(const oss_adam_chunk *chunks, int n_chunks, float *state, float lr, float beta1, float beta2, float eps,
                      float ema_decay, oss_stream_t stream) {
    fam_count(FAM_OPTIM, 36.0 * n_chunks * 2048.0);
    if (!chunks || !state) return OSS_ERR_NULL;
    if (n_chunks <= 0) return OSS_ERR_SHAPE;
    return adam_ema_step(chunks, n_chunks, state, lr, beta1, beta2, eps, ema_decay, reinterpret_cast<hipStream_t>(stream));
}

int oss_adamw_ema_step(const oss_adam_chunk *chunks, int n_chunks, float *state, float lr, float beta1, float beta2, float eps,
                       float weight_decay, float ema_decay, const float *grad_scale, oss_stream_t stream) {
    fam_count(FAM_OPTIM, 36.0 * n_chunks * 2048.0);
    if (!chunks || !state) return OSS_ERR_NULL;
    if (n_chunks <= 0 || weight_decay < 0.f) return OSS_ERR_SHAPE;
    return adam_ema_step(chunks, n_chunks, state, lr, beta1, beta2, eps, ema_decay, reinterpret_cast<hipStream_t>(stream),
                         weight_decay, grad_scale);
}

int oss_merge4(oss_dtype io, const void *out, float *y, int batch, int D, int height, int width, oss_stream_t stream) {
    fam_count(FAM_MERGE, (double)batch * D * height * width * (4.0 * esz(io) + 4.0));
    if (!out || !y) return OSS_ERR_NULL;
    if (batch <= 0 || D <= 0 || height <= 0 || width <= 0 || (long)batch * D > 65535) return OSS_ERR_SHAPE;
    return merge4(io, out, y, batch, D, height, width, reinterpret_cast<hipStream_t>(stream));
}

int oss_ln_nchw_fwd(oss_dtype xt, oss_dtype yt, const void *x, const float *weight, const float *bias, const void *gate,
                    void *y, float *mean, float *rstd, int batch, int channels, int pixels, int64_t xsb, int64_t xsc,
                    int64_t gsb, int64_t gsc, float eps, oss_stream_t stream) {
    fam_count(FAM_LN, (double)batch * channels * pixels * (esz(xt) + esz(yt) * (gate ? 2.0 : 1.0)) + 8.0 * batch * pixels);
    if (!x || !weight || !y || !mean || !rstd) return OSS_ERR_NULL;
    if (batch <= 0 || channels <= 0 || pixels <= 0 || batch > 65535) return OSS_ERR_SHAPE;
    return ln_nchw_fwd(xt, yt, x, weight, bias, gate, y, mean, rstd, batch, channels, pixels, xsb, xsc, gsb, gsc, eps,
                       reinterpret_cast<hipStream_t>(stream));
}

int oss_ln_nchw_fwd_pool_tiles(int channels, int pixels, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc) {
    if (channels <= 0 || pixels <= 0) return 0;
    return ln_nchw_fwd_pool_tiles(channels, pixels, xsb, xsc, gsb, gsc);
}

int oss_ln_nchw_fwd_pool(oss_dtype xt, oss_dtype yt, const void *x, const float *weight, const float *bias, const void *gate, void *y,
                         float *mean, float *rstd, float *pool_part, int batch, int channels, int pixels, int64_t xsb, int64_t xsc,
                         int64_t gsb, int64_t gsc, float eps, oss_stream_t stream) {
    fam_count(FAM_LN, (double)batch * channels * pixels * (esz(xt) + esz(yt) * (gate ? 2.0 : 1.0)) + 8.0 * batch * pixels);
    if (!x || !weight || !y || !mean || !rstd || !pool_part) return OSS_ERR_NULL;
    if (batch <= 0 || channels <= 0 || pixels <= 0 || batch > 65535 || channels > 4096) return OSS_ERR_SHAPE;
    return ln_nchw_fwd(xt, yt, x, weight, bias, gate, y, mean, rstd, batch, channels, pixels, xsb, xsc, gsb, gsc, eps,
                       reinterpret_cast<hipStream_t>(stream), pool_part);
}

int oss_ln_nchw_bwd(oss_dtype xt, oss_dtype yt, const void *x, const float *weight, const float *bias, const void *gate,
                    const void *dy, const float *mean, const float *rstd, void *dx, void *dgate, float *dweight,
                    float *dbias, float *partials, const void *skip_grad, int batch, int channels, int pixels, int64_t xsb,
                    int64_t xsc, int64_t gsb, int64_t gsc, int64_t dgate_batch_stride, oss_stream_t stream) {
    fam_count(FAM_LN, (double)batch * channels * pixels * (2.0 * esz(xt) + esz(yt) * (1.0 + (gate ? 2.0 : 0.0) + (skip_grad ? 1.0 : 0.0))) + 8.0 * batch * pixels);
    if (!x || !weight || !dy || !mean || !rstd || !dx || !dweight || !partials) return OSS_ERR_NULL;
    if (gate && !dgate) return OSS_ERR_NULL;
    if (batch <= 0 || channels <= 0 || pixels <= 0 || batch > 65535 || channels > 4096) return OSS_ERR_SHAPE;
    return ln_nchw_bwd(xt, yt, x, weight, bias, gate, dy, mean, rstd, dx, dgate, dweight, dbias, partials, batch, channels,
                       pixels, xsb, xsc, gsb, gsc, reinterpret_cast<hipStream_t>(stream), skip_grad, dgate_batch_stride);
}

int oss_ln_nchw_bwd_affine(oss_dtype xt, oss_dtype yt, const void *x, const float *weight, const float *bias, const void *gate,
                           const void *dy, const float *dy_mul, const float *dy_add, float add_scale, const float *mean,
                           const float *rstd, void *dx,
                           void *dgate, float *dweight, float *dbias, float *partials, const void *skip_grad, int batch,
                           int channels, int pixels, int64_t xsb, int64_t xsc, int64_t gsb, int64_t gsc,
                           int64_t dgate_batch_stride, oss_stream_t stream) {
    fam_count(FAM_LN, (double)batch * channels * pixels * (2.0 * esz(xt) + esz(yt) * (1.0 + (gate ? 2.0 : 0.0) + (skip_grad ? 1.0 : 0.0))) + 8.0 * batch * pixels);
    if (!x || !weight || !dy || !mean || !rstd || !dx || !dweight || !partials || !dy_add) return OSS_ERR_NULL;
    if (gate && !dgate) return OSS_ERR_NULL;
    if (batch <= 0 || channels <= 0 || pixels <= 0 || batch > 65535 || channels > 4096) return OSS_ERR_SHAPE;
    return ln_nchw_bwd(xt, yt, x, weight, bias, gate, dy, mean, rstd, dx, dgate, dweight, dbias, partials, batch, channels,
                       pixels, xsb, xsc, gsb, gsc, reinterpret_cast<hipStream_t>(stream), skip_grad, dgate_batch_stride, dy_mul,
                       dy_add, add_scale);
}

size_t oss_ln_nchw_bwd_partial_floats(int batch, int channels, int pixels) {
    if (batch <= 0 || channels <= 0 || pixels <= 0) return 0;
    return ln_nchw_bwd_partial_floats(batch, channels, pixels);
}

void oss_prof_enable(int on) { g_prof_on.store(on ? 1 : 0); }

void oss_prof_family_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (on) for (int f = 0; f < FAM_N; ++f) { g_fam_bytes[f] = 0.0; g_fam_calls[f] = 0; }
    g_fam_on.store(on ? 1 : 0);
}
int oss_prof_family_count(void) { return FAM_N; }
int oss_prof_family(int family, const char **name, const char **kernel_patterns, double *algorithmic_bytes, long long *calls) {
    if (family < 0 || family >= FAM_N) return OSS_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (name) *name = kFamName[family];
    if (kernel_patterns) *kernel_patterns = kFamKernels[family];
    if (algorithmic_bytes) *algorithmic_bytes = g_fam_bytes[family];
    if (calls) *calls = g_fam_calls[family];
    return OSS_OK;
}

void oss_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &k : g_prof) for (auto &v : k) for (auto &b : v) {
        for (auto &pr : b.pending) { (void)hipEventSynchronize(pr.second); g_event_pool.push_back(pr.first); g_event_pool.push_back(pr.second); }
        b.pending.clear(); b.ms = 0.0; b.bytes = 0.0; b.own = 0.0; b.launches = 0;
    }
}

int oss_prof_collect2(int which, int variant, oss_dtype io, double *total_ms, long long *launches, double *algorithmic_bytes,
                      double *own_bytes) {
    const int rc = oss_prof_collect(which, variant, io, total_ms, launches, algorithmic_bytes);
    if (rc == OSS_OK && own_bytes) {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        *own_bytes = g_prof[which][variant][(int)io].own;
    }
    return rc;
}

int oss_prof_collect(int which, int variant, oss_dtype io, double *total_ms, long long *launches, double *algorithmic_bytes) {
    if (which < 0 || which > 2 || variant < 0 || variant >= kProfVariants || (int)io < 0 || (int)io > 2) return OSS_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfBucket &b = g_prof[which][variant][(int)io];
    for (auto &pr : b.pending) {
        (void)hipEventSynchronize(pr.second);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) b.ms += ms;
        g_event_pool.push_back(pr.first); g_event_pool.push_back(pr.second);
    }
    b.pending.clear();
    if (total_ms) *total_ms = b.ms;
    if (launches) *launches = b.launches;
    if (algorithmic_bytes) *algorithmic_bytes = b.bytes;
    return OSS_OK;
}

void oss_scan_set_variant(int fwd_variant, int bwd_variant) {
    g_force_fwd.store(fwd_variant);
    g_force_bwd.store(bwd_variant);
}
int oss_scan_last_variant(int which) { return which == 0 ? g_last_fwd.load() : g_last_bwd.load(); }
void oss_scan_set_segments(int fwd_segments, int bwd_segments) {
    g_force_fwd_seg.store(fwd_segments);
    g_force_bwd_seg.store(bwd_segments);
}
int oss_scan_last_segments(int which) { return which == 0 ? g_last_fwd_segments.load() : g_last_bwd_segments.load(); }
void oss_scan_set_carry_split(int split) { g_carry_split.store(split < 0 ? 0 : split); }
int oss_scan_last_lane_states(void) { return g_last_bwd_lane_states.load(); }
int oss_scan_features(void) { return (kBuildFusedDt ? OSS_FEATURE_FUSED_DT : 0) | (kBuildLaneStates ? OSS_FEATURE_LANE_STATES : 0); }

// copy kernels of oss_hbm_copy.  Default (mode 2): one 16-byte element per lane and a grid as large as the buffer -- the
// dispatcher streams short workgroups faster than any loop keeps loads in flight: 6.12 TB/s on 1 GiB, against 5.58 for
// contiguous 32 KiB pieces per workgroup with eight loads in flight (mode 0), 5.55 for one such piece per workgroup (mode 3),
// 5.23 for sixteen loads in flight (mode 4) and 4.75 for the round-2 kernel (mode 1: grid-stride, nontemporal) --
// profiles/r03_copy_modes.txt; the guide's measured float4 copy is 6.29.  VMAMBAIR_COPY_MODE selects (A-B runs).
__global__ void __launch_bounds__(256) oss_copy_pieces_kernel(const f32x4 *src, f32x4 *dst, size_t n_pieces) {
    constexpr size_t PIECE = 2048;   // 16-byte elements per piece
    for (size_t pc = blockIdx.x; pc < n_pieces; pc += gridDim.x) {
        const f32x4 *s = src + pc * PIECE + threadIdx.x;
        f32x4 *d = dst + pc * PIECE + threadIdx.x;
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = s[k * 256];
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k * 256] = v[k];
    }
}
// mode 3: one 32 KiB piece per workgroup, grid = number of pieces (no loop: the dispatcher streams workgroups); mode 4: sixteen
// loads in flight per lane (64 KiB pieces)
__global__ void __launch_bounds__(256) oss_copy_piece_per_wg_kernel(const f32x4 *src, f32x4 *dst) {
    const f32x4 *s = src + (size_t)blockIdx.x * 2048 + threadIdx.x;
    f32x4 *d = dst + (size_t)blockIdx.x * 2048 + threadIdx.x;
    f32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = s[k * 256];
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k * 256] = v[k];
}
__global__ void __launch_bounds__(256) oss_copy_pieces16_kernel(const f32x4 *src, f32x4 *dst, size_t n_pieces) {
    constexpr size_t PIECE = 4096;
    for (size_t pc = blockIdx.x; pc < n_pieces; pc += gridDim.x) {
        const f32x4 *s = src + pc * PIECE + threadIdx.x;
        f32x4 *d = dst + pc * PIECE + threadIdx.x;
        f32x4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = s[k * 256];
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k * 256] = v[k];
    }
}
__global__ void __launch_bounds__(256) oss_copy_kernel(const f32x4 *src, f32x4 *dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + 7 * stride < n; i += 8 * stride) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
        for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v[k], dst + i + k * stride);
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) oss_copy_flat_kernel(const f32x4 *src, f32x4 *dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

int oss_hbm_copy(const void *src, void *dst, size_t n_bytes, oss_stream_t stream) {
    if (!src || !dst) return OSS_ERR_NULL;
    const size_t n = n_bytes / 16;
    if (n == 0) return OSS_OK;
    static const int mode = [] { const char *e = std::getenv("VMAMBAIR_COPY_MODE"); return e ? std::atoi(e) : 2; }();
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const f32x4 *sp = reinterpret_cast<const f32x4 *>(src);
    f32x4 *dp = reinterpret_cast<f32x4 *>(dst);
    if (mode == 1) {
        size_t blocks = (n + 255) / 256;
        if (blocks > 256 * 8) blocks = 256 * 8;
        hipLaunchKernelGGL(oss_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, s, sp, dp, n);
    } else if (mode == 2 && (n + 255) / 256 <= 0x7fffffffu) {
        hipLaunchKernelGGL(oss_copy_flat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sp, dp, n);
    } else if (mode == 3 && n % 2048 == 0) {
        hipLaunchKernelGGL(oss_copy_piece_per_wg_kernel, dim3((unsigned)(n / 2048)), dim3(256), 0, s, sp, dp);
    } else if (mode == 4 && n % 4096 == 0) {
        hipLaunchKernelGGL(oss_copy_pieces16_kernel, dim3((unsigned)std::min<size_t>(n / 4096, 256 * 4)), dim3(256), 0, s, sp, dp, n / 4096);
    } else {
        const size_t pieces = n / 2048, tail = n - pieces * 2048;
        if (pieces) {
            const unsigned blocks = (unsigned)std::min<size_t>(pieces, 256 * 8);
            hipLaunchKernelGGL(oss_copy_pieces_kernel, dim3(blocks), dim3(256), 0, s, sp, dp, pieces);
        }
        if (tail)
            hipLaunchKernelGGL(oss_copy_flat_kernel, dim3((unsigned)((tail + 255) / 256)), dim3(256), 0, s, sp + pieces * 2048,
                               dp + pieces * 2048, tail);
    }
    return (int)hipGetLastError();
}

// a one-lane kernel whose only purpose is its NAME in a kernel trace: tools/prof_summary.py keeps the launches between
// marker 1 and marker 2 (bench.py puts them around the timed region), so that a summary holds steady-state graph replays only
__global__ void oss_prof_marker_begin() {}
__global__ void oss_prof_marker_end() {}
int oss_prof_marker(int which, oss_stream_t stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (which == 1) hipLaunchKernelGGL(oss_prof_marker_begin, dim3(1), dim3(1), 0, s);
    else hipLaunchKernelGGL(oss_prof_marker_end, dim3(1), dim3(1), 0, s);
    return (int)hipGetLastError();
}

#ifndef OSS_SCAN_BUILD_ID
#define OSS_SCAN_BUILD_ID "unknown"
#endif
const char *oss_scan_build_id(void) { return OSS_SCAN_BUILD_ID; }

const char *oss_version(void) { return "vmambair_oss 0.4 (gfx950)"; }

int oss_abi_version(void) { return OSS_ABI_VERSION; }
size_t oss_abi_struct_bytes(int which) {
    switch (which) {
        case 0: return sizeof(oss_scan_fwd_params);
        case 1: return sizeof(oss_scan_bwd_params);
        case 2: return sizeof(oss_chan_params);
        default: return 0;
    }
}

}  // extern "C"
