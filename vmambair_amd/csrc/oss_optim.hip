// oss_optim.hip -- Adam + EMA of the training step as ONE launch over all parameter tensors.
// The reference's optimize_parameters (SRGAN/VmambaIR/models/MambaSISR_model.py:120-147: optimizer_g.step() with
// torch.optim.Adam, then model_ema(decay)) touches ~1450 parameter tensors, most of them tiny; torch's fused
// multi-tensor Adam + the foreach EMA take ~130 launches and ~2.5 ms per step for 60 MB of state.  Here a chunk
// table (one entry per <= 2048 elements of one tensor: pointers to param / grad / exp_avg / exp_avg_sq / ema)
// built once by the host drives a single elementwise kernel; the step counter and the two bias corrections live
// in device memory so that the launch can be replayed inside a hipGraph.  Same arithmetic as torch.optim.Adam
// (no amsgrad, eps outside the sqrt of the bias-corrected second moment); optional decoupled weight decay = torch.optim.AdamW
// and a gradient scale read from device memory = clip_grad_norm_ (the Deraining step: image_restoration_model.py:121-167).
#include "oss_device.h"
#include "oss_global_ptr.h"
#include "oss_host.h"

namespace oss {

// state[0] = step count (float), state[1] = 1 - beta1^t, state[2] = 1 - beta2^t, state[3] = learning rate (read when the
// host passes lr < 0: both reference trainings change it during the run -- MultiStepLR at 50k / 70k iterations,
// SRGAN/options/MambaSISR15_x4.yml:84-87; CosineAnnealingRestartCyclicLR stepped every iteration,
// Deraining/Deraining/Options/Deraining_mamber33.yml:81-85 through update_learning_rate, Deraining/basicsr/models/base_model.py:
// 183-193 -- and a value baked into a captured launch could not follow)
__global__ void oss_adam_tick_kernel(float *state, float beta1, float beta2) {
    const float t = state[0] + 1.f;
    state[0] = t;
    state[1] = 1.f - powf(beta1, t);
    state[2] = 1.f - powf(beta2, t);
}

__global__ void __launch_bounds__(256)
oss_adam_ema_kernel(const oss_adam_chunk *__restrict__ chunks, const float *__restrict__ state, float lr_arg, float beta1,
                    float beta2, float eps, float ema_decay, float weight_decay, const float *__restrict__ grad_scale) {
    const oss_adam_chunk c = chunks[blockIdx.x];
    const float lr = lr_arg < 0.f ? state[3] : lr_arg;
    const float decay_keep = 1.f - lr * weight_decay;
    const float step_size = lr / state[1], inv_bc2_sqrt = rsqrtf(state[2]);
    // gs: gradient-clipping coefficient (clip_grad_norm_: grads *= min(1, max_norm / (total_norm + 1e-6))) read from
    // device memory so that the launch can sit in a hipGraph; decay_keep = 1 - lr * weight_decay (AdamW, decoupled)
    const float gs = grad_scale ? *grad_scale : 1.f;
    // pointers out of the chunk table: told to be global (oss_global_ptr.h), else every access below is a FLAT instruction
    const oss_adam_chunk *ct = chunks + blockIdx.x;
    float *p = table_ptr<float>(&ct->param), *m = table_ptr<float>(&ct->exp_avg);
    float *v = table_ptr<float>(&ct->exp_avg_sq), *e = table_ptr<float>(&ct->ema);
    const float *g = table_ptr<const float>(&ct->grad);
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(e)) & 15u) == 0;
    for (int i = threadIdx.x * 4; i < c.n; i += 1024) {
        const int valid = min(4, c.n - i);
        float pv[4], gv[4], mv[4], vv[4], ev[4];
        const bool vk = vec && valid == 4;
        load_items<4>(p + i, valid, vk, pv);
        load_items<4>(g + i, valid, vk, gv);
        load_items<4>(m + i, valid, vk, mv);
        load_items<4>(v + i, valid, vk, vv);
        if (e) load_items<4>(e + i, valid, vk, ev);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            gv[k] *= gs;
            pv[k] *= decay_keep;                                              // torch.optim.AdamW: param.mul_(1 - lr * wd) first
            mv[k] = __builtin_fmaf(beta1, mv[k], (1.f - beta1) * gv[k]);      // exp_avg.lerp_(grad, 1 - beta1)
            vv[k] = __builtin_fmaf(beta2, vv[k], (1.f - beta2) * gv[k] * gv[k]);
            const float denom = sqrtf(vv[k]) * inv_bc2_sqrt + eps;
            pv[k] -= step_size * (mv[k] / denom);
            if (e) ev[k] = __builtin_fmaf(ema_decay, ev[k], (1.f - ema_decay) * pv[k]);
        }
        store_items<4>(p + i, valid, vk, pv);
        store_items<4>(m + i, valid, vk, mv);
        store_items<4>(v + i, valid, vk, vv);
        if (e) store_items<4>(e + i, valid, vk, ev);
    }
}

// One launch for ALL deferred partial-sum reductions of a backward pass (weight-gradient split-K slabs, per-workgroup
// LayerNorm / depth-wise-conv / channel-branch partials): chunk c = up to 1024 consecutive outputs of one reduction, four per
// lane (16-byte loads when source and destination allow), out[j] = sum_{k < K} src[k * stride + j] in a FIXED order -- the k range
// in 4 interleaved slices of 4 interleaved partial sums each, combined pairwise (the same on every run; the separate
// finishing kernels each have their own fixed order).
// (Round 2 gave every 64 outputs a workgroup of their own: 190 K workgroups of two loads per lane, 228 us per step.)
__device__ __forceinline__ float sum_slices(const float (&a)[4][4]) {
    float r[4];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) r[sl] = (a[sl][0] + a[sl][1]) + (a[sl][2] + a[sl][3]);
    return (r[0] + r[1]) + (r[2] + r[3]);
}

__global__ void __launch_bounds__(256)
oss_sum_partials_kernel(const oss_sum_chunk *__restrict__ chunks) {
    const oss_sum_chunk c = chunks[blockIdx.x];
    const int j = threadIdx.x * 4;
    if (j >= c.n) return;
    const float *pp = table_ptr<const float>(&chunks[blockIdx.x].src) + c.j0 + j;   // table pointers: global (oss_global_ptr.h)
    float *dst = table_ptr<float>(&chunks[blockIdx.x].dst) + j;
    const size_t st = (size_t)c.stride;
    const bool vec = j + 4 <= c.n && ((reinterpret_cast<uintptr_t>(pp) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0 && (st & 3) == 0;
    // acc[slice][i] takes k = slice + 4 i + 16 m, i.e. element [q & 3][q >> 2] of a group of 16 rows; the register arrays
    // are only ever indexed by unrolled constants (a run-time index would move them to scratch memory), so the tail of the
    // k range is a group of wave-uniformly predicated loads that add zeros
    if (vec) {
        f32x4 acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[a][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < c.K; k += 16) {
            f32x4 v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q)
                v[q] = k + q < c.K ? *reinterpret_cast<const f32x4 *>(pp + (size_t)(k + q) * st) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q & 3][q >> 2] += v[q];
        }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a[4][4];
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
#pragma unroll
                for (int i = 0; i < 4; ++i) a[sl][i] = acc[sl][i][e];
            o[e] = sum_slices(a);
        }
        *reinterpret_cast<f32x4 *>(dst) = f32x4{o[0], o[1], o[2], o[3]};
    } else {
        const int ne = min(4, c.n - j);
        float a[4][4][4];   // [element][slice][i]
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
#pragma unroll
                for (int i = 0; i < 4; ++i) a[e][sl][i] = 0.f;
        for (int k = 0; k < c.K; k += 16) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    a[e][q & 3][q >> 2] += (k + q < c.K && e < ne) ? pp[(size_t)(k + q) * st + e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < ne) dst[e] = sum_slices(a[e]);
    }
}

int sum_partials_multi(const oss_sum_chunk *chunks, int n_chunks, hipStream_t s) {
    hipLaunchKernelGGL(oss_sum_partials_kernel, dim3(n_chunks), dim3(256), 0, s, chunks);
    return (int)hipGetLastError();
}

int adam_ema_step(const oss_adam_chunk *chunks, int n_chunks, float *state, float lr, float beta1, float beta2, float eps,
                  float ema_decay, hipStream_t s, float weight_decay, const float *grad_scale) {
    hipLaunchKernelGGL(oss_adam_tick_kernel, dim3(1), dim3(1), 0, s, state, beta1, beta2);
    hipLaunchKernelGGL(oss_adam_ema_kernel, dim3(n_chunks), dim3(256), 0, s, chunks, state, lr, beta1, beta2, eps, ema_decay,
                       weight_decay, grad_scale);
    return (int)hipGetLastError();
}

}  // namespace oss
