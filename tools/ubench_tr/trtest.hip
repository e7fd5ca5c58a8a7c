// semantics probe of ds_read_b64_tr_b16 (gfx950): prints, per lane, the four 16-bit values returned when lane l reads at
// element offset 4 l of an LDS array holding lds[i] = i.   hipcc --offload-arch=gfx950 trtest.hip -o trtest && ./trtest
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short *out, int stride_elems) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    // lane i of a 16-lane group: row i / 4 of a [4][16] block with row stride `stride_elems`, columns 4 (i % 4) .. + 3
    const int off = g * 1024 + (i >> 2) * stride_elems + (i & 3) * 4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + off));
    out[l * 4 + 0] = v.x; out[l * 4 + 1] = v.y; out[l * 4 + 2] = v.z; out[l * 4 + 3] = v.w;
}
int main() {
    short *d; hipMalloc(&d, 64 * 4 * sizeof(short));
    for (int stride : {16, 136}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
        short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("row stride %d elements\n", stride);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d%s", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3], (l & 1) ? "\n" : "   |   ");
    }
    return 0;
}
