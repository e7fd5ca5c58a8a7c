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
    float *p = reinterpret_cast<float *>(c.param), *m = reinterpret_cast<float *>(c.exp_avg);
    float *v = reinterpret_cast<float *>(c.exp_avg_sq), *e = reinterpret_cast<float *>(c.ema);
    const float *g = reinterpret_cast<const float *>(c.grad);
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
// LayerNorm / depth-wise-conv / channel-branch partials): chunk c = 64 consecutive outputs of one reduction,
// out[j] = sum_{k < K} src[k * stride + j] in a fixed order (4 slices of the k range, combined in a fixed order).
__global__ void __launch_bounds__(256)
oss_sum_partials_kernel(const oss_sum_chunk *__restrict__ chunks) {
    __shared__ float red[4][64];
    const oss_sum_chunk c = chunks[blockIdx.x];
    const int colx = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const float *src = reinterpret_cast<const float *>(c.src);
    float s = 0.f;
    if (colx < c.n) {
        const float *pp = src + c.j0 + colx;
        const size_t st = (size_t)c.stride;
        int k = slice;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (; k + 12 < c.K; k += 16) {
            s0 += pp[(size_t)k * st];
            s1 += pp[(size_t)(k + 4) * st];
            s2 += pp[(size_t)(k + 8) * st];
            s3 += pp[(size_t)(k + 12) * st];
        }
        for (; k < c.K; k += 4) s0 += pp[(size_t)k * st];
        s = (s0 + s1) + (s2 + s3);
    }
    red[slice][colx] = s;
    __syncthreads();
    if (slice == 0 && colx < c.n) reinterpret_cast<float *>(c.dst)[colx] = (red[0][colx] + red[1][colx]) + (red[2][colx] + red[3][colx]);
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
