// tools/ubench.hip -- gfx950 micro-benchmarks that price the design choices of the scan kernels:
// VALU issue rates (v_fma_f32, v_pk_fma_f32, v_exp_f32, DPP-source VOP2), LDS float-atomic rate and
// the achievable HBM copy bandwidth.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ITERS = 4096;

__global__ void __launch_bounds__(256) k_fma(float *out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
        x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

// plain v_fma_f32 pinned by inline asm: the C loop of k_fma above is turned into v_pk_fma_f32 by the SLP vectorizer
// (checked in the ISA), so its "lane-instr/s" is really the packed rate -- round 1 read it as the scalar rate
__global__ void __launch_bounds__(256) k_fma_asm(float *out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
// one dependent chain per wave (what a recurrence looks like): issue-to-issue latency of v_fma_f32
__global__ void __launch_bounds__(256) k_fma_chain(float *out, float a, float b) {
    float x0 = threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     : "+v"(x0) : "v"(a), "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0;
}
__global__ void __launch_bounds__(256) k_pkfma_chain(float *out, float a, float b) {
    f32x2 A = {a, a}, B = {b, b};
    f32x2 x0 = {(float)threadIdx.x, 1.f};
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                     "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                     : "+v"(x0) : "v"(A), "v"(B));
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0.x + x0.y;
}

__global__ void __launch_bounds__(256) k_pkfma(float *out, float a, float b) {
    f32x2 A = {a, a}, B = {b, b};
    f32x2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(A), "v"(B));
    }
    f32x2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

__global__ void __launch_bounds__(256) k_exp(float *out, float a) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + .1f, x2 = x0 + .2f, x3 = x0 + .3f, x4 = x0 + .4f, x5 = x0 + .5f, x6 = x0 + .6f, x7 = x0 + .7f;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                     "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + a;
}

// 1 exp : 3 fma interleaved (the mix of the scan's first pass)
__global__ void __launch_bounds__(256) k_mix(float *out, float a, float b) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + .1f, y0 = x0, y1 = x1, y2 = x0, y3 = x1, y4 = x0, y5 = x1;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n"
                     "v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5) : "v"(a), "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + y0 + y1 + y2 + y3 + y4 + y5;
}

__global__ void __launch_bounds__(256) k_dpp(float *out, float a) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_mul_f32_dpp %0, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_mul_f32_dpp %2, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_mul_f32_dpp %4, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %5, %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_mul_f32_dpp %6, %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %7, %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void __launch_bounds__(256) k_ldsadd(float *out, float a) {
    __shared__ float acc[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) acc[i] = 0.f;
    __syncthreads();
    float *p = acc + threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __hip_atomic_fetch_add(p + j * 256, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x];
}

__global__ void __launch_bounds__(256) k_ldsread(float *out) {
    __shared__ f32x4 buf[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    f32x4 s = {0, 0, 0, 0};
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(threadIdx.x * 16)), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            s += v;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

__global__ void __launch_bounds__(256) k_copy(const f32x4 *src, f32x4 *dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) dst[i] = src[i];
}

template <typename F> static float time_ms(F f, int reps = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    const int blocks = prop.multiProcessorCount * 8;  // 8 x 256-thread blocks per CU = 8 waves/SIMD
    float *out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    const double lanes = (double)blocks * 256;
    struct { const char *name; double ops; float ms; } r[8];
    float ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); });
    printf("v_fma_f32        : %8.2f T lane-instr/s  (%.1f TFLOP/s)\n", lanes * ITERS * 8 / ms / 1e9, lanes * ITERS * 8 * 2 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); });
    printf("v_pk_fma_f32     : %8.2f T lane-instr/s  (%.1f TFLOP/s)\n", lanes * ITERS * 8 / ms / 1e9, lanes * ITERS * 8 * 4 / ms / 1e9);
    for (int wps : {8, 3, 1}) {   // waves per SIMD: 8 (full), 3 (the scan backward's occupancy), 1
        const int nb = prop.multiProcessorCount * wps;
        const double ln = (double)nb * 256;
        const double simd_cycles = 2.4e9 * prop.multiProcessorCount * 4;
        float m1 = time_ms([&] { hipLaunchKernelGGL(k_fma_asm, dim3(nb), dim3(256), 0, 0, out, 1.0001f, 0.5f); });
        float m2 = time_ms([&] { hipLaunchKernelGGL(k_pkfma, dim3(nb), dim3(256), 0, 0, out, 1.0001f, 0.5f); });
        float m3 = time_ms([&] { hipLaunchKernelGGL(k_fma_chain, dim3(nb), dim3(256), 0, 0, out, 1.0001f, 0.5f); });
        float m4 = time_ms([&] { hipLaunchKernelGGL(k_pkfma_chain, dim3(nb), dim3(256), 0, 0, out, 1.0001f, 0.5f); });
        float m5 = time_ms([&] { hipLaunchKernelGGL(k_exp, dim3(nb), dim3(256), 0, 0, out, 0.f); });
        auto cyc = [&](float ms_) { return simd_cycles * (ms_ * 1e-3) / (ln / 64 * ITERS * 8); };   // SIMD cycles (at 2.4 GHz) per wave instruction
        printf("%d waves/SIMD: v_fma_f32 (asm) %6.2f T lane-instr/s = %.2f cyc/instr | v_pk_fma_f32 %6.2f T = %.2f cyc | dependent chain: fma %.2f cyc, pk_fma %.2f cyc | v_exp_f32 %.2f cyc\n",
               wps, ln * ITERS * 8 / m1 / 1e9, cyc(m1), ln * ITERS * 8 / m2 / 1e9, cyc(m2), cyc(m3), cyc(m4), cyc(m5));
    }
    ms = time_ms([&] { hipLaunchKernelGGL(k_exp, dim3(blocks), dim3(256), 0, 0, out, 0.f); });
    printf("v_exp_f32        : %8.2f T lane-instr/s\n", lanes * ITERS * 8 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); });
    printf("1 exp : 3 fma    : %8.2f T lane-instr/s\n", lanes * ITERS * 8 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_dpp, dim3(blocks), dim3(256), 0, 0, out, 1.0001f); });
    printf("v_mul_f32_dpp    : %8.2f T lane-instr/s\n", lanes * ITERS * 8 / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_ldsadd, dim3(blocks), dim3(256), 0, 0, out, 1.0f); });
    printf("ds_add_f32       : %8.2f T lane-ops/s    (%.2f wave-instr/clk/CU at 2.4 GHz)\n", lanes * ITERS * 8 / ms / 1e9,
           lanes * ITERS * 8 / 64 / (ms * 1e-3) / prop.multiProcessorCount / 2.4e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_ldsread, dim3(blocks), dim3(256), 0, 0, out); });
    printf("ds_read_b128     : %8.2f TB/s\n", lanes * ITERS * 8 * 16 / ms / 1e9);
    for (size_t mb : {256, 1024, 4096}) {
        const size_t bytes = mb << 20;
        f32x4 *src, *dst; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes));
        CK(hipMemset(src, 1, bytes));
        ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(prop.multiProcessorCount * 8), dim3(256), 0, 0, src, dst, bytes / 16); });
        printf("copy %5zu MiB   : %8.2f GB/s read+write\n", mb, 2.0 * bytes / ms / 1e6);
        CK(hipFree(src)); CK(hipFree(dst));
    }
    return 0;
}
