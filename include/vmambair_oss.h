/*
 * vmambair_oss.h -- C ABI of the MI355X (gfx950) Omni-Selective-Scan core.
 *
 * This is the drop-in boundary for the one native module the reference's archs import:
 *     import selective_scan_cuda_core          (reference: SRGAN/VmambaIR/archs/MambaSISR6_arch.py:22,
 *                                               Deraining/basicsr/models/archs/mamber32_arch.py:18,
 *                                               RealSR/VmambaIR/archs/MambaRealSR11_arch.py:24,32)
 * whose two entry points are pybind11 functions
 *     fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows) -> [out, x]
 *     bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows)
 *            -> [du, ddelta, dA, dB, dC, dD, ddelta_bias]
 * (reference: Mamba/kernels/selective_scan/csrc/selective_scan/cus/selective_scan.cpp:157-164,
 *  241-250,351-354).  Behind those, the reference fills a parameter struct of sizes, element
 * strides and raw pointers (selective_scan.h:26-90, SSMParamsBase / SSMParamsBwd) and launches
 * one kernel per I/O dtype on a caller-supplied stream
 * (cus/selective_scan_core_fwd.cu:6-8, cus/selective_scan_core_bwd.cu:6-8).
 *
 * The functions below are what a binding for that module links against: plain pointers, sizes,
 * element strides and a HIP stream -- no torch types.  The binding layer (ours:
 * vmambair_amd/_capi.py + selective_scan_cuda_core.py; a maintainer's: INTEGRATION.md) does the
 * shape/dtype checks and allocates outputs exactly as cus/selective_scan.cpp:165-220,256-327.
 *
 * All device pointers must live on the device that is current when the call is made.  Calls are
 * asynchronous on `stream`, never synchronise the host, and are re-entrant.
 * Return value: 0 on success, a negative OSS_ERR_* for a rejected argument, or a positive
 * hipError_t if the launch failed.
 */
#ifndef VMAMBAIR_OSS_H
#define VMAMBAIR_OSS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSS_OK 0
#define OSS_ERR_NULL (-1)        /* a required pointer is NULL                               */
#define OSS_ERR_SHAPE (-2)       /* batch/dim/seqlen/dstate/n_groups invalid (dim % groups)   */
#define OSS_ERR_DSTATE (-3)      /* dstate > OSS_MAX_DSTATE (reference: selective_scan.cpp:191) */
#define OSS_ERR_WORKSPACE (-4)   /* workspace missing or too small                           */

#define OSS_MAX_DSTATE 256

/* I/O element type of u, delta, B, C, out, dout, du, ddelta (the reference's input_t);
 * A, D, delta_bias, x and every weight gradient are always float (weight_t). */
typedef enum { OSS_F32 = 0, OSS_F16 = 1, OSS_BF16 = 2 } oss_dtype;

/* Opaque stream handle: a hipStream_t (NULL = the default stream). */
typedef void *oss_stream_t;

/* Mirrors SSMParamsBase (selective_scan.h:26-66).  Strides are in ELEMENTS; the last (time)
 * dimension of u, delta, B, C, out must be contiguous (selective_scan.cpp:183-199). */
typedef struct {
    int batch, dim, seqlen, dstate, n_groups;
    int delta_softplus;
    /* Omni-scan direction handling without materialising the four flattenings:
     *   rev_group_start : rows of groups g >= rev_group_start are scanned from t = L-1 down to 0, i.e.
     *                     every time-indexed tensor of those rows/groups (u, delta, B, C, out, dout,
     *                     du, ddelta, dB, dC) is read and written mirrored in time.  n_groups (or
     *                     any value >= n_groups) = none: the reference's plain call.
     *   u_row_mod       : when > 0, row d reads u[:, d % u_row_mod, :] (du still has `dim` rows), so
     *                     that directions k and k+2 share one copy of the activations.  0 = off. */
    int rev_group_start, u_row_mod;
    /* a_log_form != 0: the `A` pointer holds A_log and the kernels use A = -exp(A_log) (the archs'
     * `As = -torch.exp(self.A_logs.float())`, MambaSISR6_arch.py:415); bwd then returns dA_log. */
    int a_log_form, reserved0_;
    int64_t u_batch_stride, u_d_stride;
    int64_t delta_batch_stride, delta_d_stride;
    int64_t out_batch_stride, out_d_stride;
    int64_t A_d_stride;                                  /* A is (dim, dstate), dstate stride 1 */
    int64_t B_batch_stride, B_group_stride, B_dstate_stride;
    int64_t C_batch_stride, C_group_stride, C_dstate_stride;
    const void *u, *delta;  /* (batch, dim, seqlen) io dtype                                   */
    const float *A;         /* (dim, dstate)                                                   */
    const void *B, *C;      /* (batch, n_groups, dstate, seqlen) io dtype                      */
    const float *D;         /* (dim) or NULL                                                   */
    const float *delta_bias;/* (dim) or NULL                                                   */
    void *out;              /* (batch, dim, seqlen) io dtype                                   */
    float *x;               /* (batch, dim, oss_scan_num_chunks(seqlen), 2*dstate) contiguous  */
    /* Delta computed INSIDE the scan (the archs' `dts = einsum(dts, dt_projs_weight)` followed by
     * the scan, SRGAN/VmambaIR/archs/MambaSISR6_arch.py:409-424): with dt_weight != NULL the `delta` pointer holds the
     * rank-dt_rank factor z: (batch, n_groups, >= dt_rank rows, seqlen) io dtype -- the first rows of x_dbl -- with element
     * strides (delta_batch_stride, dt_group_stride, dt_rank_stride), time contiguous (mirrored like B / C for groups >=
     * rev_group_start), and the kernels evaluate
     *     delta[b, d, t] = sum_r dt_weight[d * dt_rank + r] * z[b, group(d), r, t]        (fp32, r ascending)
     * so the (batch, dim, seqlen) delta tensor never exists.  delta_d_stride is ignored then.  dt_rank <= 8 (every
     * reference config where the fused form is used: D <= 128). */
    const float *dt_weight; /* (dim, dt_rank) float contiguous, or NULL = `delta` is the (batch, dim, seqlen) tensor */
    int dt_rank, reserved1_;
    int64_t dt_group_stride, dt_rank_stride;
    /* Optional scratch of oss_scan_fwd_workspace_bytes() bytes (no init needed) for oss_scan_fwd.  With it, calls whose
     * (batch, group, row tile) grid cannot fill the GPU (batch-1 inference tiles, few-row levels) are cut into time
     * segments that run as separate workgroups (two launches: segment-local pass, then the real pass from the folded
     * carries; oss_scan_set_segments).  NULL / too small = every row is walked by one workgroup, as the reference does
     * (cus/selective_scan_fwd_kernel.cuh:101-102).  Ignored by oss_scan_bwd (which has its own workspace). */
    void *workspace;
    size_t workspace_bytes;
    /* Optional lane states (round 3; oss_scan_features() & OSS_FEATURE_LANE_STATES -- in every library since round 6; a
     * library built without it ignores the field): oss_scan_lane_state_floats() floats, layout [batch][dim][dstate][L8] with
     * L8 = round_up(ceil(seqlen / 8), 64): entry k of a (batch, row, state) line is the state h ENTERING scan steps 8k .. 8k+7
     * (h after step 8k - 1; 0 for k = 0).  oss_scan_fwd writes them when hs != NULL (a by-product of its second pass);
     * oss_scan_bwd given the same buffer in f.hs reads them instead of re-running the forward recurrence and one lane scan per
     * state (round-2 kernels, dstate <= 64, not the fused-delta form) -- same gradients to fp32 round-off.  NULL = recompute
     * from `x`, as the reference's backward does (cus/selective_scan_bwd_kernel.cuh:184-186).  The torch layers carry the
     * buffer as a third tensor next to (out, x): selective_scan_fwd(..., want_hs) -> [out, x, hs], selective_scan_bwd(..., hs). */
    float *hs;
    /* Per-call launch tuning (round 6; ABI 7).  0 everywhere = the library's heuristics, which is what every product caller
     * passes (memset the struct).  These fields are the thread- and stream-safe way to force a launch shape; the process-global
     * setters further down (oss_scan_set_variant / _segments / _carry_split) exist for the test-suite and A-B timing scripts only
     * and are overridden by a non-zero field.
     *   tune_variant     : v + 1 forces kernel variant v of THIS call's direction (oss_scan_fwd: forward variants 0..7; in
     *                      oss_scan_bwd_params.f the field is ignored -- the backward's own tune_variant is used)
     *   tune_segments    : 1 = never cut the call in time, n > 1 = n time segments per row (clamped to the chunk count)
     *   tune_carry_split : n >= 1 = pieces per main segment of the local / reverse-carry pass (see oss_scan_set_carry_split);
     *                      oss_scan_bwd reads it from f.tune_carry_split */
    int tune_variant, tune_segments, tune_carry_split, reserved2_;
} oss_scan_fwd_params;

/* Mirrors SSMParamsBwd (selective_scan.h:68-90). */
typedef struct {
    oss_scan_fwd_params f;  /* f.out unused; f.x = the tensor fwd returned (required when
                               oss_scan_num_chunks(seqlen) > 1, selective_scan.cpp:310)          */
    int64_t dout_batch_stride, dout_d_stride;
    int64_t du_batch_stride, du_d_stride;
    int64_t ddelta_batch_stride, ddelta_d_stride;
    const void *dout;       /* (batch, dim, seqlen) io dtype                                   */
    void *du, *ddelta;      /* (batch, dim, seqlen) io dtype                                   */
    float *dA;              /* (dim, dstate) contiguous, overwritten                           */
    void *dB, *dC;          /* (batch, n_groups, dstate, seqlen) io dtype, overwritten (the cast of
                               selective_scan.cpp:347 is fused); contiguous unless
                               dBC_group_stride is set                                          */
    float *dD;              /* (dim) or NULL, overwritten                                      */
    float *ddelta_bias;     /* (dim) or NULL, overwritten                                      */
    void *workspace;        /* oss_scan_bwd_workspace_bytes() bytes of scratch, no init needed */
    size_t workspace_bytes;
    int dout_row_mod;       /* > 0: row d reads dout[:, d % dout_row_mod, :] (the merge hands the same
                               gradient to directions k and k + 2); 0 = off                        */
    int reserved_;
    int64_t dBC_group_stride; /* element stride between (batch, group) blocks of dB and of dC; the batch
                                 stride is n_groups times it.  0 = dstate * seqlen (contiguous).  Lets
                                 the caller have dB / dC written into the rows of a wider buffer.  */
    /* f.dt_weight != NULL (delta computed inside the scan): ddelta is not written; instead
     *   ddt[b, g, r, t]  = sum over the rows d of group g of dt_weight[d, r] * ddelta[b, d, t]   (io dtype; the dt rows of the
     *                      gradient of x_dbl -- with dBC_group_stride the kernel fills ALL rows of that gradient)
     *   ddt_weight[d, r] = sum over b, t of ddelta[b, d, t] * z[b, g, r, t]                       (float, (dim, dt_rank))
     * Only the round-2 kernels do this: dstate <= 64, 16-bit or float I/O, dt_rank <= 8. */
    void *ddt;
    float *ddt_weight;
    int64_t ddt_batch_stride, ddt_group_stride, ddt_rank_stride;
    /* per-call launch tuning of the backward (see oss_scan_fwd_params.tune_*): v + 1 forces backward variant v (1, 10..13);
     * 1 = never segment, n > 1 = n time segments.  0 = heuristic. */
    int tune_variant, tune_segments;
    /* row-tile partials of dB / dC (scratch between the main and the finishing kernel): 0 / 1 = fp32 (default: the reference's fp32
     * accumulation, cus/selective_scan_bwd_kernel.cuh:208-221, to the letter); 2 = bf16 partials at bf16 I/O (round-2 kernels, not the
     * fused-delta / lane-state forms; ignored elsewhere): half the scratch traffic, but every tile's partial is rounded to 8 bits of
     * mantissa before the tiles are summed -- an absolute error of 2^-9 x |partial| that can leave the reference's bf16 tolerance
     * when the tiles cancel.  An opt-in for callers who accept that (measured +1.7 % on the headline step). */
    int tune_partials, reserved3_;
    /* (round 6) The dt-factor gradient in the backward's finishing launch.  With finish_dt_weight != NULL (and f.dt_weight == NULL:
     * delta materialised, ddelta written as usual) the finishing kernel -- which already adds the row-tile partials of dB / dC --
     * also evaluates the adjoint of the archs' `dts = einsum("b k r l, k d r -> b k d l", dts, dt_projs_weight)`
     * (SRGAN/VmambaIR/archs/MambaSISR6_arch.py:411):
     *     ddt[b, g, r, t] = sum over the rows d of group g of finish_dt_weight[d * finish_dt_rank + r] * ddelta[b, d, t]
     * into `ddt` (element strides ddt_batch_stride / ddt_group_stride / ddt_rank_stride; normally the first rows of the gradient
     * of x_dbl that dBC_group_stride already points dB / dC into), in extra workgroups of the SAME launch: the two are independent
     * and memory-bound, so the block's backward has one launch less (oss_dt_dgrad_kernel of oss_proj_dgrad, which is then called
     * with ddts == NULL).  oss_scan_bwd_finish_dt_ok() says whether a shape qualifies (rank <= 8, seqlen % 4 == 0; 8 / 16-byte
     * aligned ddelta / ddt rows). */
    const float *finish_dt_weight;  /* (dim, finish_dt_rank) float contiguous, or NULL */
    int finish_dt_rank, reserved4_;
} oss_scan_bwd_params;

/* Time steps between two saved states in `x` (the reference's is 2048,
 * selective_scan.cpp:217; `x` is opaque to every caller, only our bwd reads it). */
int oss_scan_chunk(void);
int oss_scan_num_chunks(int seqlen);

/* Replaces selective_scan_fwd_cuda<1, input_t, float> (cus/selective_scan_fwd_kernel.cuh:174-207). */
size_t oss_scan_fwd_workspace_bytes(int batch, int dim, int seqlen, int dstate, int n_groups);   /* 0 when seqlen <= 256 */
size_t oss_scan_lane_state_floats(int batch, int dim, int seqlen, int dstate);                    /* size of `hs` */
int oss_scan_fwd(const oss_scan_fwd_params *p, oss_dtype io, oss_stream_t stream);

/* Replaces selective_scan_bwd_cuda<1, input_t, float> (cus/selective_scan_bwd_kernel.cuh:275-310)
 * plus the zero-fills and casts around it (cus/selective_scan.cpp:319-327,347). */
/* (the workspace size covers the fused-delta form with dt_rank <= 8 as well) */
size_t oss_scan_bwd_workspace_bytes(int batch, int dim, int seqlen, int dstate, int n_groups);
int oss_scan_bwd(const oss_scan_bwd_params *p, oss_dtype io, oss_stream_t stream);

/* 1 when oss_scan_bwd_params.finish_dt_weight may be used at this length / rank (see the struct) */
int oss_scan_bwd_finish_dt_ok(int seqlen, int dt_rank);

/* 1 when the fused-delta form (oss_scan_fwd_params.dt_weight) covers this SS2D_1 shape: 16-bit I/O on the matrix-core
 * projection kernels (which may then be called with dts / ddts == NULL), dt_rank <= 8, dstate <= 64, seqlen >= 512
 * (shorter sequences take the small-shape kernels, which read delta). */
int oss_scan_fused_dt_ok(oss_dtype io, int batch, int D, int C, int R, int dstate, int seqlen);

/* ---- PROCESS-GLOBAL tuning overrides: test-suite / A-B timing scripts ONLY ------------------------------------------------
 * The three setters below mutate process-wide state that every later call from every thread and stream reads; they are not a
 * product interface.  An integrator who needs a particular launch shape sets the per-call fields
 * oss_scan_fwd_params.tune_* / oss_scan_bwd_params.tune_* instead, which win over these.
 * Kernel-variant override: -1 = heuristic (default).  Forward variants 0..7, backward variants 0..13
 * (tables in oss_scan_fwd.hip / oss_scan_bwd.hip; 8 and 9 = two states per pass in packed fp32, oss_scan_bwd_pair.h; 10..13 =
 * the round-2 kernel, oss_scan_bwd_v2.h).  An
 * unknown number falls back to the small-shape variant. */
void oss_scan_set_variant(int fwd_variant, int bwd_variant);
int oss_scan_last_variant(int which /* 0 fwd, 1 bwd */);
/* Time segments per row of the next launches (tuning / tests): -1 = heuristic (segments only when the launch would leave
 * most CUs without a workgroup), 0 or 1 = never, n > 1 = n segments (clamped to the number of chunks of the kernel variant;
 * the backward segments only its round-2 kernels, variants 10..13, dstate <= 64, and never the fused-delta form).
 * oss_scan_last_segments: what the last call used (1 = unsegmented). */
void oss_scan_set_segments(int fwd_segments, int bwd_segments);
int oss_scan_last_segments(int which /* 0 fwd, 1 bwd */);
/* (round 5) A time-segmented call runs a cheap first launch -- the forward's segment-local pass, the backward's reverse-carry
 * pass -- whose per-segment (product, state) pairs the main launch folds.  That first launch has its own, FINER segmentation: each main
 * segment's cps chunks are cut into pieces of ceil(cps / split) chunks (ANY split: the last piece is shorter or empty and an empty
 * piece keeps the identity pair), one workgroup and one pair slot per piece, so that it fills the CUs the main launch's segment
 * count was chosen for.  0 = heuristic (default: up to 512 workgroups), 1 = as coarse as the main launch (rounds 2-4).  Results
 * differ from split = 1 only in how the carries are associated (fp32 round-off).  The pair slots live in the caller's workspace:
 * size it with oss_scan_fwd_workspace_bytes / oss_scan_bwd_workspace_bytes (they cover min(chunks, 64) slots per row and state);
 * a workspace that holds the main segments' slots but not the finer pieces' makes the launch fall back to split = 1, and one
 * that does not even hold those to the unsegmented launch. */
void oss_scan_set_carry_split(int split);
/* 1 when the last oss_scan_bwd call ran the kernels that load the forward pass's lane states (f.hs), else 0 */
int oss_scan_last_lane_states(void);

/* Depth-wise 3x3 convolution, stride 1, zero padding 1, of the OSS block: SS2D_1.conv2d
 * (SRGAN/VmambaIR/archs/MambaSISR6_arch.py:286-294) and the EFFN dwconv (:209); both are
 * nn.Conv2d(C, C, 3, padding=1, groups=C).  x, y, dy, dx: (batch, C, H, W) io dtype with contiguous
 * H*W planes and element strides (batch, channel); weight (C, 9) and bias (C) float (bias may be
 * NULL).  oss_dwconv3x3_fwd computes y = conv(x) + bias; with flip = 1 it applies the mirrored taps,
 * i.e. the input gradient dx = conv_transpose(dy).  oss_dwconv3x3_wgrad overwrites dweight (C, 9)
 * and dbias (C, or NULL) with the sums over batch and pixels; `partials` is batch*C*10 floats of
 * scratch (no init needed).
 * Fused activation (SS2D_1: x = act(conv2d(x)), MambaSISR6_arch.py:486): with pre_silu != NULL (contiguous (batch, C, H, W))
 * the forward stores the convolution there and y = silu(convolution); the weight-gradient call given the same pre_silu
 * uses dy * silu'(pre_silu) as the gradient reaching the convolution and also writes it to dpre (contiguous), which is then
 * the input of the flip = 1 call that produces dx. */
int oss_dwconv3x3_fwd(oss_dtype io, const void *x, const float *weight, const float *bias, void *y, void *pre_silu, int batch,
                      int channels, int height, int width, int64_t x_batch_stride, int64_t x_channel_stride,
                      int64_t y_batch_stride, int64_t y_channel_stride, int flip, oss_stream_t stream);
int oss_dwconv3x3_wgrad(oss_dtype io, const void *x, const void *dy, float *dweight, float *dbias, float *partials,
                        const void *pre_silu, void *dpre, int batch, int channels, int height, int width,
                        int64_t x_batch_stride, int64_t x_channel_stride, int64_t dy_batch_stride, int64_t dy_channel_stride,
                        oss_stream_t stream);

/* The depth-wise convolution fused with the element-wise step that follows it, without ever storing the convolution
 * (16-bit io, and OSS_F32 since ABI 6 / round 4; oss_dwconv3x3_fused_ok(io, H, W, 1 | 2) says whether a shape qualifies: width % 8 == 0,
 * and 1 (silu) or 2 (gate) planes of (H + 2) x W elements fit the 160 KiB LDS of a workgroup; pointers 16-byte aligned and strides
 * multiples of 8 elements, else OSS_ERR_SHAPE).
 *   oss_dwconv3x3_silu_fwd: y = silu(conv(x) + bias)                         (SS2D_1: x = act(conv2d(x)), MambaSISR6_arch.py:486)
 *   oss_dwconv3x3_silu_bwd: dx, dweight, dbias of it from x and dy -- the convolution is recomputed from the rows of x the
 *                           weight gradient reads anyway, dy * silu' stays in LDS between the two passes of ONE launch
 *   oss_dwgate_fwd:         out (batch, hidden, H, W) = gelu(x1) * x2 with x1, x2 = the two channel halves of
 *                           conv(t) + bias, t (batch, 2 hidden, H, W)         (FeedForward.forward, MambaSISR6_arch.py:213-217)
 *                           The forward streams (nothing LDS-resident) and takes every plane oss_dwgate_fwd_ok accepts: width a multiple
 *                           of 8 and <= 512, any height -- the inference path of planes the backward's form cannot hold
 *   oss_dwgate_bwd:         dt, dweight (2 hidden, 9), dbias of it from t and dout, one launch
 * partials: batch * channels * 10 floats of scratch (channels = 2 hidden for the gate); dbias may be NULL.  The gradients that
 * reach the convolution are rounded to the io type before both uses, as the separate kernels hand them over. */
int oss_dwconv3x3_fused_ok(oss_dtype io, int height, int width, int channels_per_workgroup);
int oss_dwconv3x3_silu_fwd(oss_dtype io, const void *x, const float *weight, const float *bias, void *y, int batch, int channels,
                           int height, int width, int64_t x_batch_stride, int64_t x_channel_stride, int64_t y_batch_stride,
                           int64_t y_channel_stride, oss_stream_t stream);
int oss_dwconv3x3_silu_bwd(oss_dtype io, const void *x, const float *weight, const float *bias, const void *dy, void *dx,
                           float *dweight, float *dbias, float *partials, int batch, int channels, int height, int width,
                           int64_t x_batch_stride, int64_t x_channel_stride, int64_t dy_batch_stride, int64_t dy_channel_stride,
                           int64_t dx_batch_stride, int64_t dx_channel_stride, oss_stream_t stream);
int oss_dwgate_fwd_ok(oss_dtype io, int height, int width);

/* The whole second half of an OSS block, forward only (inference), as ONE launch:
 *   out = x + project_out( gelu(x1) * x2 ),  x1, x2 = dwconv3x3( project_in( norm2(x) ) ).chunk(2)
 * (FeedForward.forward, SRGAN/VmambaIR/archs/MambaSISR6_arch.py:201-218; the block's `x = x + self.ffn(self.norm2(x))`, :513-516;
 * LayerNorm :144-195 -- norm_bias NULL = the BiasFree form.)  The 2 hidden-channel and hidden-channel intermediates never reach memory;
 * they are rounded to the io type where the launch-per-layer chain (oss_conv1x1_fwd with norm, oss_dwgate_fwd, oss_conv1x1_fwd with
 * residual) stores them, so the two forms agree to the rounding of the last accumulation.  No biases on the three convolutions (no
 * reference config has them: `bias: False` in every options file).
 *   x, out: (batch, channels, H, W) of the io type (OSS_F16 / OSS_BF16), rows contiguous, 16-byte aligned, strides multiples of 8
 * With HP = hidden rounded up to a multiple of 16 and every padding row / column ZERO (the kernel's loop has no masks):
 *   w_in:   (2 HP, channels) of the io type -- project_in.weight rounded once by the caller (in inference a constant): rows
 *           0 .. hidden - 1 = its first half (x1), rows HP .. HP + hidden - 1 = its second half (x2)
 *   w_dw:   (2 HP, 9) float -- dwconv.weight in the same row order
 *   w_out:  (channels, HP) of the io type -- project_out.weight
 * oss_effn_fwd_ok: channels in {32, 48, 64, 96}, width a multiple of 8; anything else (and every training call) stays on the chain. */
int oss_effn_fwd_ok(oss_dtype io, int channels, int hidden, int height, int width);
/* w_in / w_dw / w_out of oss_effn_fwd from the module's float parameters (project_in.weight (2 hidden, channels), dwconv.weight
 * (2 hidden, 9), project_out.weight (channels, hidden)) in ONE launch: an inference graph captures it in front of the forward and
 * so never replays stale copies after the parameters changed in place. */
int oss_effn_round_weights(oss_dtype io, const float *project_in, const float *dwconv, const float *project_out, void *w_in, float *w_dw,
                           void *w_out, int channels, int hidden, oss_stream_t stream);
int oss_effn_fwd(oss_dtype io, const void *x, const float *norm_weight, const float *norm_bias, const void *w_in, const float *w_dw,
                 const void *w_out, void *out, int batch, int channels, int hidden, int height, int width, int64_t x_batch_stride,
                 int64_t x_channel_stride, int64_t out_batch_stride, int64_t out_channel_stride, float eps, oss_stream_t stream);
int oss_dwgate_fwd(oss_dtype io, const void *t, const float *weight, const float *bias, void *out, int batch, int hidden,
                   int height, int width, int64_t t_batch_stride, int64_t t_channel_stride, int64_t out_batch_stride,
                   int64_t out_channel_stride, oss_stream_t stream);
int oss_dwgate_bwd(oss_dtype io, const void *t, const float *weight, const float *bias, const void *dout, void *dt,
                   float *dweight, float *dbias, float *partials, int batch, int hidden, int height, int width,
                   int64_t t_batch_stride, int64_t t_channel_stride, int64_t dout_batch_stride, int64_t dout_channel_stride,
                   int64_t dt_batch_stride, int64_t dt_channel_stride, oss_stream_t stream);

/* The same convolution + silu together with cross_scan_2d's two forward flattenings (MambaSISR6_arch.py:399-404, 486): the scans of
 * SS2D_1 read silu(conv2d(x)) row-major AND column-major, and their input gradient comes back as one tensor per flattening.
 *   oss_dwconv3x3_silu_flat2_fwd: x2 (batch, 2, channels, H * W) contiguous = [silu(conv(x) + bias) | the same planes transposed,
 *                                 element (h, w) at w * H + h] -- what oss_dwconv3x3_silu_fwd + oss_cross_scan2 produce, bit for bit
 *   oss_dwconv3x3_silu_flat2_bwd: as oss_dwconv3x3_silu_bwd with dy(h, w) = g2[:, 0](h, w) + g2[:, 1](w, h) rounded to the io type --
 *                                 what oss_cross_merge2 + oss_dwconv3x3_silu_bwd produce, bit for bit; g2 (batch, 2, channels, H * W)
 * oss_dwconv3x3_flat2_ok(io, H, W): fused_ok(io, H, W, 1), H % 8 == 0, W / 8 a power of two <= 32; else OSS_ERR_SHAPE. */
int oss_dwconv3x3_flat2_ok(oss_dtype io, int height, int width);
int oss_dwconv3x3_silu_flat2_fwd(oss_dtype io, const void *x, const float *weight, const float *bias, void *x2, int batch, int channels,
                                 int height, int width, int64_t x_batch_stride, int64_t x_channel_stride, oss_stream_t stream);
int oss_dwconv3x3_silu_flat2_bwd(oss_dtype io, const void *x, const float *weight, const float *bias, const void *g2, void *dx,
                                 float *dweight, float *dbias, float *partials, int batch, int channels, int height, int width,
                                 int64_t x_batch_stride, int64_t x_channel_stride, int64_t dx_batch_stride, int64_t dx_channel_stride,
                                 oss_stream_t stream);

/* 1x1 convolutions of the OSS block (in_conv / out_conv / project_in / project_out,
 * MambaSISR6_arch.py:205,211,281,329) as MFMA GEMMs on NCHW tensors; io = OSS_BF16 or OSS_F16 (fp32
 * I/O is rejected with OSS_ERR_SHAPE: it stays on the vendor conv).  weight: float (Cout, Cin)
 * contiguous (master weights, converted in the loader); x / dy: (batch, C, pixels) io dtype, pixels
 * contiguous, element strides (batch, channel); y / dx contiguous.
 *   fwd  : y  = W x + bias (+ residual, contiguous like y: the skip connection of the block, MambaSISR6_arch.py:515-516)
 *   dgrad: dx = W^T dy
 *   wgrad: dweight (Cout, Cin) float = sum_{b,p} dy x^T and, when dbias != NULL, dbias (Cout) float =
 *          sum_{b,p} dy (an all-ones row appended to x inside the kernel); `partials` =
 *          oss_conv1x1_wgrad_partial_floats() floats of scratch. */
int oss_conv1x1_fwd(oss_dtype io, const void *x, const float *weight, const float *bias, const void *residual, void *y,
                    int batch, int cout, int cin, int pixels, int64_t x_batch_stride, int64_t x_channel_stride,
                    oss_stream_t stream);
int oss_conv1x1_dgrad(oss_dtype io, const void *dy, const float *weight, void *dx, int batch, int cout, int cin, int pixels,
                      int64_t dy_batch_stride, int64_t dy_channel_stride, oss_stream_t stream);
size_t oss_conv1x1_wgrad_partial_floats(int batch, int cout, int cin, int pixels);
/* Tuning override of oss_conv1x1_wgrad / oss_proj_wgrad: MFMA tiles per wave, 0 = 1 x 1 (default), 12 / 21 / 22 = rows x columns
 * of 32 x 32 tiles (fewer re-reads of the operands, fewer waves).  Same results up to the summation order; initial value from
 * the environment variable VMAMBAIR_WGRAD_TILE. */
/* Workgroup-level form of oss_conv1x1_fwd / oss_conv1x1_dgrad without residual (oss_conv1x1_wg.hip): cin % 16 == 0, cin <= 192,
 * pixels % 128 == 0, 16-byte aligned tensors; anything else returns OSS_ERR_SHAPE (callers then use the wave-level kernels).
 * transposed_weight = 1: weight is (cin, cout) row-major, i.e. the call is the input gradient of a (cin, cout) convolution. */
int oss_conv1x1_wg(oss_dtype io, const void *x, const float *weight, const float *bias, void *y, int batch, int cout, int cin,
                   int pixels, int64_t x_batch_stride, int64_t x_channel_stride, int transposed_weight, oss_stream_t stream);
/* LayerNorm over channels followed by a 1x1 convolution as ONE launch (norm1 -> in_conv, norm2 -> project_in of the OSS block,
 * MambaSISR6_arch.py:514-516 with :144-195, :205, :281): n = LN(x) (weight, bias or NULL for the BiasFree form, eps), written out
 * together with mean / rstd (batch, pixels) for the backward passes (oss_ln_nchw_bwd, oss_conv1x1_wgrad take them as before), and
 * y = W n + bias.  Shapes of oss_conv1x1_wg: cin % 16 == 0, cin <= 192, pixels % 128 == 0, 16-bit io, 16-byte aligned tensors;
 * oss_ln_conv1x1_ok says whether (io, cout, cin, pixels) qualifies. */
int oss_ln_conv1x1_ok(oss_dtype io, int cout, int cin, int pixels);
int oss_ln_conv1x1_fwd(oss_dtype io, const void *x, const float *ln_weight, const float *ln_bias, float eps, void *n, float *mean,
                       float *rstd, const float *weight, const float *bias, void *y, int batch, int cout, int cin, int pixels,
                       int64_t x_batch_stride, int64_t x_channel_stride, oss_stream_t stream);
/* The backward of that pair's activation path as ONE launch: dn = W^T dy (oss_conv1x1_dgrad) never leaves the CU, the LayerNorm
 * backward (oss_ln_nchw_bwd: x, mean, rstd, + skip_grad, d weight / d bias through `partials` =
 * oss_conv1x1_dgrad_ln_bwd_partial_floats(batch, cin, pixels) floats and the usual (deferred) finishing sum) runs on it in LDS.
 * Shapes: cin <= 128, cout % 16 == 0, 2 cin <= cout <= 192, pixels % 128 == 0 (oss_conv1x1_dgrad_ln_bwd_ok); x, skip_grad, dx contiguous
 * (batch, cin, pixels); dy (batch, cout, pixels) with element strides.  The weight gradient stays oss_conv1x1_wgrad on n. */
int oss_conv1x1_dgrad_ln_bwd_ok(oss_dtype io, int cout, int cin, int pixels, int batch);
size_t oss_conv1x1_dgrad_ln_bwd_partial_floats(int batch, int cin, int pixels);
int oss_conv1x1_dgrad_ln_bwd(oss_dtype io, const void *dy, const float *weight, const void *x, const float *ln_weight, int ln_has_bias,
                             const float *mean, const float *rstd, const void *skip_grad, void *dx, float *dln_weight, float *dln_bias,
                             float *partials, int batch, int cout, int cin, int pixels, int64_t dy_batch_stride,
                             int64_t dy_channel_stride, oss_stream_t stream);
/* A-B switches of the dispatch inside oss_conv1x1_fwd / _dgrad: on = 0 never takes the workgroup-level kernel (env
 * VMAMBAIR_CONV1X1_WG=0); pixels = 64 | 128 forces its tile width, 0 = by grid size */
void oss_conv1x1_set_wg(int on, int pixels);
void oss_conv1x1_wgrad_set_tile(int mode);
/* pixels per partial product of the GROUPED weight-gradient launch (oss_flush_wgrads), in units of 512: default 4 (env
 * VMAMBAIR_WGRAD_SPAN); 1 reproduces the one-problem launches bit for bit, larger values write fewer partial vectors */
void oss_conv1x1_wgrad_set_span(int mult);
int oss_conv1x1_wgrad(oss_dtype io, const void *dy, const void *x, float *dweight, float *dbias, float *partials, int batch,
                      int cout, int cin, int pixels, int64_t dy_batch_stride, int64_t dy_channel_stride,
                      int64_t x_batch_stride, int64_t x_channel_stride, oss_stream_t stream);

/* The two in-block projections of the spatial branch (MambaSISR6_arch.py:406-411), omni form, and the
 * flattenings around them.  Layouts (all contiguous, io dtype): x2 (batch, 2, D, L) = the row-major
 * and column-major flattenings of the (batch, D, H, W) activations (k = 0, 1;
 * directions 2, 3 are the same rows walked backwards by the scan); xdbl (batch, 4, C, L) with
 * C = R + 2 dstate rows per direction: R dt rows, then B, then C of the scan; dts (batch, 4, D, L).
 * Weights are float: x_proj_weight (4, C, D), dt_projs_weight (4, D, R).
 *   oss_cross_scan2 : x (batch, D, H, W) of in_type (contiguous planes, element strides batch / channel)
 *                     -> x2 of out_type        (replaces the stack / transpose / flip of :395-404)
 *   oss_cross_merge2: dx (batch, D, H, W) = g2[:, 0] + transpose(g2[:, 1])   (its adjoint)
 *   oss_proj_fwd    : xdbl[b,k,c,:] = sum_d x_proj_weight[k,c,d] x2[b,k%2,d,:], rounded to io;
 *                     dts[b,k,d,:]  = sum_r dt_projs_weight[k,d,r] xdbl[b,k,r,:]
 *   oss_proj_dgrad  : dxdbl holds dB / dC of the scan backward in its rows c >= R on entry; the dt rows
 *                     are filled with dt_projs_weight^T ddts, then
 *                     dx2[b,j] = sum_{k in {j,j+2}} (x_proj_weight[k]^T dxdbl[b,k] + du[b,k]) (du may be NULL)
 *   oss_proj_wgrad  : both weight gradients (float, overwritten) as split-K MFMA products; io =
 *                     OSS_BF16 / OSS_F16 only (OSS_ERR_SHAPE for float: callers use a vendor GEMM);
 *                     partials = oss_proj_wgrad_partial_floats() floats of scratch. */
int oss_cross_scan2(oss_dtype in_type, oss_dtype out_type, const void *x, void *x2, int batch, int D, int height, int width,
                    int64_t x_batch_stride, int64_t x_channel_stride, oss_stream_t stream);
int oss_cross_merge2(oss_dtype io, const void *g2, void *dx, int batch, int D, int height, int width, oss_stream_t stream);
int oss_proj_fwd(oss_dtype io, const void *x2, const float *x_proj_weight, const float *dt_projs_weight, void *xdbl, void *dts,
                 int batch, int D, int C, int R, int seqlen, oss_stream_t stream);
int oss_proj_dgrad(oss_dtype io, const void *ddts, void *dxdbl, const void *du, const float *x_proj_weight,
                   const float *dt_projs_weight, void *dx2, int batch, int D, int C, int R, int seqlen, oss_stream_t stream);
size_t oss_proj_wgrad_partial_floats(int batch, int D, int C, int R, int seqlen);
/* 16-bit I/O runs oss_proj_fwd / _dgrad on the matrix cores (MFMA); float I/O, or force_vector_alu != 0
 * (tests, A-B timing), on the vector-ALU kernels.  Same results up to the summation order. */
void oss_proj_set_path(int force_vector_alu);
/* 1 when oss_proj_fwd may be called with dts == NULL / oss_proj_dgrad with ddts == NULL at this shape (the matrix-core projection
 * kernels: 16-bit I/O, even seqlen, D <= 768, 2 C <= 256, R <= 32, and the vector-ALU path not forced) -- i.e. when the dt rows of
 * the gradient of x_dbl may come from somewhere else (oss_scan_bwd_params.finish_dt_weight, the fused-delta form) */
int oss_proj_rows_optional_ok(oss_dtype io, int batch, int D, int C, int R, int seqlen);
int oss_proj_wgrad(oss_dtype io, const void *x2, const void *xdbl, const void *dxdbl, const void *ddts, float *dx_proj_weight,
                   float *ddt_projs_weight, float *partials, int batch, int D, int C, int R, int seqlen, oss_stream_t stream);

/* Channel branch of SS2D_1 (MambaSISR6_arch.py:438-483; RealSR form MambaRealSR11_arch.py:758-817) between the
 * pooled descriptor and the gate vector c, one launch per direction of autograd (oss_channel.hip).  All float.
 * L = d_inner (the scan runs over the channels), dc = dc_inner (1 for the RealSR form, then cin_* / cout_* are
 * NULL and seq = pooled), Rc = the channel dt rank, Cc = Rc + 32 (dc_state is 16 in every reference config). */
typedef struct {
    int B, L, dc, Rc, Cc, reserved_;
    const float *pooled;            /* (B, L): mean of y2 over the pixels                                   */
    const float *cin_w, *cin_b;     /* conv_cin (dc) weight / bias, or NULL                                  */
    const float *Wxc;               /* xc_proj_weight (2, Cc, dc)                                            */
    const float *Wdtc;              /* dtc_projs_weight (2, dc, Rc)                                          */
    const float *dt_bias;           /* dtc_projs_bias (2 dc)                                                 */
    const float *A_logs;            /* Ac_logs (2 dc, 16)                                                    */
    const float *Dsc;               /* (2 dc)                                                                */
    const float *cout_w, *cout_b;   /* conv_cout (dc) weight / (1) bias, or NULL                             */
    const float *cn_w, *cn_b;       /* channel_norm weight / bias (L)                                        */
    float *zt;                      /* (B, 2, L, Cc)   saved: projections, c fastest                         */
    float *dts;                     /* (B, 2 dc, L)    saved: dt_proj output before bias / softplus          */
    float *hs;                      /* (B, 2 dc, L, 16) saved: the scan state after every step               */
    float *y;                       /* (B, 2 dc, L)    saved: scan outputs (direction 1 stored un-flipped)   */
    float *yc;                      /* (B, L)          saved: LayerNorm input                                */
    float *stat;                    /* (B, 2)          saved: LayerNorm mean, rstd                           */
    float *c;                       /* (B, L)          OUT: the gate vector                                  */
    /* oss_chan_fwd only.  pool_part != NULL: `pooled` is an OUTPUT -- pooled[b, l] = pool_scale * sum_k pool_part[b, k, l] over the
     * n_part per-workgroup output sums oss_ln_nchw_fwd_pool left (summed in k order) -- instead of an input.                 */
    const float *pool_part;         /* (B, n_part, L) or NULL                                                */
    int n_part;
    float pool_scale;               /* 1 / (H W)                                                             */
} oss_chan_params;
int oss_chan_fwd(const oss_chan_params *p, oss_stream_t stream);
/* gc: gradient of c (B, L).  OUT dpooled (B, L); grads: oss_chan_grad_floats() floats = the parameter gradients
 * summed over the batch in batch order, laid out [cn_w L | cn_b L | cout_w dc | cout_b 1 | Ac_logs 2 dc 16 |
 * Dsc 2 dc | dt_bias 2 dc | Wdtc 2 dc Rc | Wxc 2 Cc dc | cin_w dc | cin_b dc]; scratch:
 * oss_chan_bwd_scratch_floats() floats. */
size_t oss_chan_grad_floats(int L, int dc, int Rc, int Cc);
size_t oss_chan_bwd_scratch_floats(int B, int L, int dc, int Rc, int Cc);
int oss_chan_bwd(const oss_chan_params *p, const float *gc, float *dpooled, float *grads, float *scratch, oss_stream_t stream);
/* out[b, c] = alpha * sum_p a[b, c, p] * (bmul ? bmul[b, c, p] : 1)  (pooling, and the gate's gradient);
 * y[b, c, p] = x[b, c, p] * (mul ? 1 + mul[b, c] : 1) + (add ? alpha * add[b, c] : 0)  (the gate and its adjoint).
 * a, bmul, x: io dtype, contiguous planes, element strides (batch, channel); y contiguous. */
int oss_rowsum(oss_dtype io, const void *a, const void *bmul, float *out, int batch, int channels, int pixels,
               int64_t a_batch_stride, int64_t a_channel_stride, int64_t b_batch_stride, int64_t b_channel_stride, float alpha,
               oss_stream_t stream);
int oss_row_affine(oss_dtype io, const void *x, const float *mul, const float *add, void *y, int batch, int channels,
                   int pixels, int64_t x_batch_stride, int64_t x_channel_stride, float alpha, oss_stream_t stream);

/* Gate of the EFFN (MambaSISR6_arch.py:213-217): h (batch, 2, half_elems) = the two channel halves of
 * dwconv(project_in(x)) (halves contiguous, element batch stride given); out (batch, half_elems) = gelu(h[:, 0]) * h[:, 1]
 * with the exact (erf) gelu; bwd writes the gradient of both halves, dh (batch, 2, half_elems) contiguous. */
int oss_gelu_gate_fwd(oss_dtype io, const void *h, void *out, int batch, size_t half_elems, int64_t h_batch_stride,
                      oss_stream_t stream);
int oss_gelu_gate_bwd(oss_dtype io, const void *h, const void *dout, void *dh, int batch, size_t half_elems,
                      int64_t h_batch_stride, int64_t dout_batch_stride, oss_stream_t stream);

/* Deferred finishing.  Every two-stage reduction of the backward kernels (split-K slabs of oss_conv1x1_wgrad /
 * oss_proj_wgrad, per-workgroup partials of oss_ln_nchw_bwd, oss_dwconv3x3_wgrad, oss_chan_bwd) normally ends with a
 * small finishing launch that adds the partial vectors in a fixed order (~12 such launches per OSS block).  After
 * oss_set_defer_finish(1) those launches are not issued: each call registers its reduction with the library instead
 * (scratch buffers and gradient outputs must then stay alive and unread), and oss_flush_finishes runs ALL of them as
 * one launch: it writes the chunk table (oss_deferred_chunks() entries of sizeof(oss_sum_chunk) bytes, one per <= 1024
 * outputs) into host_table (pinned; must stay alive if the call is captured into a hipGraph), copies it to
 * device_table on the stream and launches the summation there.  Same fixed summation order on every run.
 * oss_set_defer_finish(0/1) also drops whatever was registered and not flushed. */
typedef struct {
    const void *src;   /* partial vector 0 */
    void *dst;         /* output of element j0 */
    int64_t stride;    /* floats between consecutive partial vectors */
    int j0, n, K, reserved_;
} oss_sum_chunk;
void oss_set_defer_finish(int on);
size_t oss_deferred_chunks(void);
int oss_flush_finishes(void *host_table, void *device_table, size_t capacity_chunks, oss_stream_t stream);
/* (round 6) only the FIRST max_chunks registered chunks (registration order = the order of the backward); 0 = all.  The rest stays
 * registered.  With oss_flush_wgrads_n: the gradients of the layers whose backward ran first can be finished -- and handed to the
 * data-parallel all-reduce -- while the remaining groups are still being flushed (vmambair_amd/train_graph.py: grad_buckets). */
int oss_flush_finishes_n(void *host_table, void *device_table, size_t capacity_chunks, size_t max_chunks, oss_stream_t stream);

/* Deferred weight gradients.  After oss_set_defer_wgrad(1), oss_conv1x1_wgrad and the x_proj / dt_proj products of oss_proj_wgrad
 * (16-bit I/O) do not launch: each call records its problem (operands, partial buffer and outputs must stay alive and unread)
 * and -- with oss_set_defer_finish(1) -- its finishing sum; oss_flush_wgrads then runs EVERY recorded product as one grouped
 * launch (the descriptor table, oss_deferred_wgrad_table_bytes() bytes, goes through host_table -- pinned, kept alive when
 * the call is captured into a hipGraph -- to device_table).  Call it before oss_flush_finishes.  With
 * oss_conv1x1_wgrad_set_span(1) the results are bit-identical to the one-launch-per-product form (same tiles, same partial
 * layout, same summation order); the default span (4: a workgroup walks four 512-pixel pieces into one accumulator, a quarter
 * of the partial vectors) changes the order of the fp32 additions -- equal to round-off, bit-identical from run to run.
 * A product recorded while oss_set_defer_finish is off is rejected (OSS_ERR_WORKSPACE): its finishing sum would read partials
 * that do not exist before the flush.
 * oss_set_defer_wgrad(0/1) also drops whatever was recorded and not flushed. */
void oss_set_defer_wgrad(int on);
size_t oss_deferred_wgrads(void);
size_t oss_deferred_wgrad_table_bytes(void);
int oss_flush_wgrads(void *host_table, void *device_table, size_t capacity_bytes, oss_stream_t stream);
/* only the FIRST max_products recorded products (recording order = the order of the backward: last layers first); 0 = all */
int oss_flush_wgrads_n(void *host_table, void *device_table, size_t capacity_bytes, size_t max_products, oss_stream_t stream);

/* Adam + EMA of the training step (MambaSISR_model.py:120-147: torch.optim.Adam without amsgrad / weight decay,
 * then ema = decay * ema + (1 - decay) * param) as one elementwise launch over a chunk table in device memory:
 * one entry per <= OSS_ADAM_CHUNK consecutive elements of one parameter tensor (all float; ema may be NULL).
 * state: 4 floats in device memory {step count, 1 - beta1^t, 1 - beta2^t, learning rate} (the fourth is read only with
 * lr < 0; a caller that always passes lr >= 0 may hand over 3); the call advances the step
 * count first (so the launch pair can be replayed inside a hipGraph).  lr >= 0: the learning rate of this call (baked into
 * a captured launch); lr < 0: the kernel reads state[3], which the host may rewrite between replays -- the reference
 * trainings change the rate during a run (MultiStepLR, SRGAN/options/MambaSISR15_x4.yml:84-87;
 * CosineAnnealingRestartCyclicLR every iteration, Deraining/Deraining/Options/Deraining_mamber33.yml:81-85). */
#define OSS_ADAM_CHUNK 2048
typedef struct {
    void *param;
    const void *grad;
    void *exp_avg, *exp_avg_sq, *ema;
    int n, reserved_;
} oss_adam_chunk;
int oss_adam_ema_step(const oss_adam_chunk *chunks, int n_chunks, float *state, float lr, float beta1, float beta2, float eps,
                      float ema_decay, oss_stream_t stream);
/* The Deraining tree's step (Deraining/basicsr/models/image_restoration_model.py:121-167): torch.optim.AdamW (decoupled
 * decay: param *= 1 - lr * weight_decay before the Adam update) after clip_grad_norm_(params, 0.01).  grad_scale (device
 * memory, one float, or NULL) multiplies every gradient inside the launch: the caller computes
 * min(1, max_norm / (total_norm + 1e-6)) on the device, so nothing is read back and the step stays graph-capturable.
 * ema_decay as above (entries with ema == NULL skip it: the Deraining YAML sets no EMA). */
int oss_adamw_ema_step(const oss_adam_chunk *chunks, int n_chunks, float *state, float lr, float beta1, float beta2, float eps,
                       float weight_decay, float ema_decay, const float *grad_scale, oss_stream_t stream);

/* Cross-merge of the four spatial directions (MambaSISR6_arch.py:427-430) on the omni scan's
 * un-flipped outputs: out (batch, 4, D, H*W) io dtype contiguous (directions 0/2 row-major, 1/3
 * column-major) -> y (batch, D, H, W) float = ((o0 + o2) + T o1) + T o3, the reference's association
 * order.  batch * D must be < 65536. */
int oss_merge4(oss_dtype io, const void *out, float *y, int batch, int D, int height, int width, oss_stream_t stream);

/* Per-pixel LayerNorm over the channel axis, NCHW in / NCHW out (LayerNorm of the OSS block,
 * SRGAN/VmambaIR/archs/MambaSISR6_arch.py:144-195: biased variance, eps inside the sqrt; bias == NULL
 * selects the BiasFree form x / sqrt(var + eps) * w).  x: (batch, C, pixels) of type x_type with element
 * strides (batch, channel), pixels contiguous; y (and dgate) contiguous (batch, C, pixels) of y_type;
 * gate of y_type with its own (batch, channel) strides; weight / bias / dweight / dbias float (C).
 * gate != NULL fuses y = LN(x) * silu(gate)
 * (SS2D_1: y1 * act(z), :488-493).  mean / rstd: (batch, pixels) float, written by fwd, read by bwd.
 * bwd: partials = oss_ln_nchw_bwd_partial_floats(batch, channels, pixels) floats of scratch
 * (per-workgroup dweight / dbias sums, combined in a fixed order by a finishing kernel).  skip_grad (x_type,
 * contiguous like dx, or NULL) is added to dx: the gradient arriving over the block's skip connection.
 * dgate_batch_stride: element stride between the images of dgate (0 = channels * pixels): SS2D_1 splits one tensor into x | z
 * (MambaSISR6_arch.py:487), and the gradient of the gate z is written straight into its half of that tensor's gradient. */
size_t oss_ln_nchw_bwd_partial_floats(int batch, int channels, int pixels);
int oss_ln_nchw_fwd(oss_dtype x_type, oss_dtype y_type, const void *x, const float *weight, const float *bias,
                    const void *gate, void *y, float *mean, float *rstd, int batch, int channels, int pixels,
                    int64_t x_batch_stride, int64_t x_channel_stride, int64_t gate_batch_stride,
                    int64_t gate_channel_stride, float eps, oss_stream_t stream);
/* oss_ln_nchw_fwd that also leaves, per workgroup of 128 pixels and channel, the sum of its OUTPUT values (as stored) in pool_part
 * (batch, oss_ln_nchw_fwd_pool_tiles(...), channels): the pooled descriptor of SS2D_1's channel branch without a pass over y
 * (oss_chan_params.pool_part).  _pool_tiles returns 0 for shapes that do not take the 128-pixel form (odd pixels / strides). */
int oss_ln_nchw_fwd_pool_tiles(int channels, int pixels, int64_t x_batch_stride, int64_t x_channel_stride, int64_t gate_batch_stride,
                               int64_t gate_channel_stride);
int oss_ln_nchw_fwd_pool(oss_dtype x_type, oss_dtype y_type, const void *x, const float *weight, const float *bias,
                         const void *gate, void *y, float *mean, float *rstd, float *pool_part, int batch, int channels, int pixels,
                         int64_t x_batch_stride, int64_t x_channel_stride, int64_t gate_batch_stride,
                         int64_t gate_channel_stride, float eps, oss_stream_t stream);
int oss_ln_nchw_bwd(oss_dtype x_type, oss_dtype y_type, const void *x, const float *weight, const float *bias,
                    const void *gate, const void *dy, const float *mean, const float *rstd, void *dx, void *dgate,
                    float *dweight, float *dbias, float *partials, const void *skip_grad, int batch, int channels, int pixels,
                    int64_t x_batch_stride, int64_t x_channel_stride, int64_t gate_batch_stride,
                    int64_t gate_channel_stride, int64_t dgate_batch_stride, oss_stream_t stream);
/* The same with the incoming gradient taken as dy * (1 + dy_mul[b, c]) + add_scale * dy_add[b, c] (both (batch, channels) float;
 * dy_mul NULL: no factor): the backward of SS2D_1's channel gate  out = y2 * c + y2 | y2 + c,  c = f(mean_hw(y2))
 * (MambaSISR6_arch.py:495-496) is  d y2 = g * (1 + c) | g  +  d pooled / (H W), an affine map per (image, channel) of the gradient g
 * that reaches `out` -- applied here on load instead of in a pass of its own (oss_row_affine) that writes d y2 only for this
 * kernel to read it back. */
int oss_ln_nchw_bwd_affine(oss_dtype x_type, oss_dtype y_type, const void *x, const float *weight, const float *bias,
                           const void *gate, const void *dy, const float *dy_mul, const float *dy_add, float add_scale,
                           const float *mean, const float *rstd, void *dx, void *dgate, float *dweight, float *dbias, float *partials,
                           const void *skip_grad, int batch, int channels, int pixels, int64_t x_batch_stride,
                           int64_t x_channel_stride, int64_t gate_batch_stride, int64_t gate_channel_stride,
                           int64_t dgate_batch_stride, oss_stream_t stream);

/* Optional per-launch timing of the two scan kernels (bench.py's roofline leg): when enabled every
 * main forward / backward kernel launch is bracketed by HIP events recorded on the launch stream.
 * oss_prof_collect synchronises those events and returns, for one bucket (which: 0 = forward
 * kernel, 1 = backward main kernel, 2 = backward finishing kernel; variant; io dtype), the summed kernel time, the number of launches
 * and the summed ALGORITHMIC bytes (SURVEY.md section 8d formulas).
 * Returns 0, or OSS_ERR_SHAPE for a bucket out of range. */
void oss_prof_enable(int on);
void oss_prof_reset(void);
/* variant 16 + v: the time-segmented launches of kernel variant v (carry / local pass included in the call's time) */
int oss_prof_collect(int which, int variant, oss_dtype io, double *total_ms, long long *launches,
                     double *algorithmic_bytes);
/* ... plus the kernel's OWN algorithmic bytes (tensors the omni form shares between directions counted once); which = 2
 * selects the backward's finishing kernel (time only). */
int oss_prof_collect2(int which, int variant, oss_dtype io, double *total_ms, long long *launches, double *algorithmic_bytes,
                      double *own_bytes);

/* HBM copy-kernel (float4 read+write) used by bench.py to measure the achievable bandwidth in
 * the same run as the scan kernels; copies n_bytes (multiple of 16) from src to dst. */
int oss_hbm_copy(const void *src, void *dst, size_t n_bytes, oss_stream_t stream);

/* Launches an empty kernel named oss_prof_marker_begin (which = 1) / oss_prof_marker_end (2) on `stream`: bench.py brackets
 * its timed region with them so that tools/prof_summary.py can cut a rocprofv3 kernel trace down to the steady-state
 * replays (no warm-up, capture or vendor solver search). */
int oss_prof_marker(int which, oss_stream_t stream);

/* Hash of the sources the scan kernels are compiled from (oss_scan_*.hip/.h, oss_device.h), fixed at build time.  PMC
 * counter records under profiles/ carry the id of the build they were measured on; bench.py reports counters only when it
 * matches the loaded library ("stale" otherwise). */
const char *oss_scan_build_id(void);

const char *oss_version(void);

/* Dense 3x3 convolutions (stride 1, zero padding 1, NCHW) with a THIN side -- at most 4 channels in or out: the layers the UNets
 * open and close with (OverlapPatchEmbed conv(3 -> 48), SRGAN/VmambaIR/archs/MambaSISR6_arch.py:520-528; the x4 tail's last
 * conv(96 -> 3) at the output resolution, archs/common.py:45-60; Mamber32.output, Deraining/basicsr/models/archs/
 * mamber32_arch.py:608).  HBM-bound stencils, not GEMMs: 16-bit I/O, fp32 master weights (cout, cin, 3, 3) and bias, fp32
 * accumulation, 8 pixels per lane.  oss_conv3x3_thin_ok says whether a shape qualifies (bf16 / fp16, width % 8 == 0,
 * min(cin, cout) <= 4); pointers 16-byte aligned, plane strides (elements) multiples of 8, planes contiguous.
 *   fwd:   y = conv(x) + bias
 *   dgrad: dx = the transposed convolution of dy
 *   wgrad: dweight (cout, cin, 3, 3) and -- cout <= 4 only -- dbias, through `partial`
 *          (oss_conv3x3_thin_wgrad_partial_floats floats, no init): one partial vector per image, added in batch order by the
 *          deferred finishing launch (oss_set_defer_finish) or right away. */
int oss_conv3x3_thin_ok(oss_dtype io, int cin, int cout, int height, int width);
int oss_conv3x3_thin_fwd(oss_dtype io, const void *x, const float *weight, const float *bias, void *y, int batch, int cin, int cout,
                         int height, int width, int64_t x_batch_stride, int64_t x_channel_stride, int64_t y_batch_stride,
                         int64_t y_channel_stride, oss_stream_t stream);
int oss_conv3x3_thin_dgrad(oss_dtype io, const void *dy, const float *weight, void *dx, int batch, int cin, int cout, int height,
                           int width, int64_t dy_batch_stride, int64_t dy_channel_stride, int64_t dx_batch_stride,
                           int64_t dx_channel_stride, oss_stream_t stream);
size_t oss_conv3x3_thin_wgrad_partial_floats(int batch, int cin, int cout);
int oss_conv3x3_thin_wgrad(oss_dtype io, const void *x, const void *dy, float *dweight, float *dbias, float *partial, int batch, int cin,
                           int cout, int height, int width, int64_t x_batch_stride, int64_t x_channel_stride, int64_t dy_batch_stride,
                           int64_t dy_channel_stride, oss_stream_t stream);

/* Algorithmic bytes of the block's NON-scan launches, by kernel family (round 6; bench.py `roofline.non_scan`).  While counting
 * is on, every entry point of this header that launches a kernel adds the bytes its launch must move at minimum (each operand
 * read once, each result written once, fp32 master weights included, scratch partials not) to its family's counter -- host-side
 * bookkeeping only, so it also works while a hipGraph is being captured.  oss_prof_family_enable(1) clears the counters.
 * oss_prof_family: name, the '|'-separated substrings of the family's kernel names (for matching a kernel trace), bytes and
 * entry-point calls since counting was switched on.  The scan kernels have their own event profiler above (oss_prof_collect2). */
void oss_prof_family_enable(int on);
int oss_prof_family_count(void);
int oss_prof_family(int family, const char **name, const char **kernel_patterns, double *algorithmic_bytes, long long *calls);

/* Runtime-selected scan forms the loaded library contains.  Since round 6 every build has both (rounds 4-5 kept them behind
 * build flags): the fused-delta form of SURVEY.md 8f row 1 (chosen per call by oss_scan_fwd_params.dt_weight != NULL) and the
 * lane states (chosen per call by oss_scan_fwd_params.hs != NULL).  Both are parity-green against the oracle and measured
 * slower inside the training step, so the host layers leave them off unless asked.  A build that
 * compiled one out (-DOSS_WITHOUT_FUSED_DT / -DOSS_WITHOUT_LANE_STATES) rejects dt_weight with OSS_ERR_SHAPE, answers 0 from
 * oss_scan_fused_dt_ok() / oss_scan_lane_state_floats() and ignores `hs`. */
#define OSS_FEATURE_FUSED_DT 1
#define OSS_FEATURE_LANE_STATES 2
int oss_scan_features(void);

/* ABI guard.  oss_scan_fwd_params / oss_scan_bwd_params / oss_chan_params cross the boundary BY POINTER, so a binding layer
 * compiled against another revision of this header would hand the kernels garbage pointers.  OSS_ABI_VERSION is bumped with
 * every change of a struct or of an entry point's argument list; oss_abi_struct_bytes(which) is the library's own sizeof
 * (which: 0 = oss_scan_fwd_params, 1 = oss_scan_bwd_params, 2 = oss_chan_params; 0 for anything else).  Every binding layer
 * in this tree (vmambair_amd/_capi.py, csrc_host/oss_torch_host.cpp through vmambair_amd/_host.py) compares both with its
 * own compile-time values when it loads and refuses to run on a mismatch. */
#define OSS_ABI_VERSION 7
int oss_abi_version(void);
size_t oss_abi_struct_bytes(int which);

#ifdef __cplusplus
}
#endif
#endif /* VMAMBAIR_OSS_H */
