// Cycles per ds_read_b64_tr_b16 wave-instruction as a function of the LDS row stride (bytes) of a
// natural [pixel][channel] bf16 image, with the lane->address map of conv_wgrad_tr.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void k(long* out, int stride, int pixstep, int nwaves_active) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((int*)smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, la = lane & 15, kq = lane >> 4;
    const int prow = 4 * kq + (la >> 2), cq = 4 * (la & 3);
    int acc = 0;
    long t0 = clock64();
    for (int it = 0; it < 256; ++it) {
        const int base = ((it & 7) * 16) * stride * pixstep;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int o0 = base + prow * pixstep * stride + (16 * m + cq) * 2;
            s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + (o0 & 0xfff8)));
            acc += v[0] + v[1] + v[2] + v[3];
        }
    }
    long t1 = clock64();
    if (lane == 0) out[threadIdx.x >> 6] = t1 - t0;
    if (acc == 12345678) out[63] = acc;
}
int main() {
    long* d; hipMalloc(&d, 64 * 8);
    long h[64];
    int strides[] = {32, 64, 96, 112, 128, 160, 192, 224, 288, 416};
    for (int nw : {1, 4, 8}) {
        for (int ps : {1, 2}) for (int st : strides) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64 * nw), 65536, 0, d, st, ps, nw);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("waves %d pixstep %d stride %4d : %.1f cycles per tr-read per wave (wave0)\n", nw, ps, st, (double)h[0] / (256 * 3));
        }
    }
    return 0;
}
