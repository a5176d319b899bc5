// LDS throughput of the access pattern of conv_wgrad_dma.hip: 8 waves of a workgroup issue ds_read_b64_tr_b16 (or plain
// ds_read_b64 / ds_read_b128) on [32 px][64 ch] sub-tiles with the 64-byte-half swizzle.  Prints wave-cycles per
// instruction per CU.   hipcc --offload-arch=gfx950 -O3 tr16_rate.hip -o tr16_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
template <int MODE>
__global__ void k(long* out, int iters) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    for (int i = threadIdx.x; i < 40960 / 4; i += blockDim.x) ((int*)smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int g = lane >> 4, i16 = lane & 15, sw = (i16 >> 3) & 1;
    const int lane_off = (8 * (g >> 1) + (i16 >> 2)) * 128 + (4 * (g & 1) + (i16 & 3)) * 8;
    const int col0 = lane_off + ((0 ^ sw) << 6), col1 = lane_off + ((1 ^ sw) << 6);
    const char* base = smem + (wid % 4) * 4096;
    int acc = 0;
    __syncthreads();
    const long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int off = ((u & 1) ? col1 : col0) + ((u >> 1) & 1) * 2048 + ((u >> 2) & 1) * 512 + ((it & 3) * 4096);
            if (MODE == 0) {
                s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off));
                acc += v[0] + v[3];
            } else if (MODE == 1) {
                s16x4 v = *(const s16x4*)(base + off);
                acc += v[0] + v[3];
            } else {
                i32x4 v = *(const i32x4*)(base + (off & ~15));
                acc += v[0] + v[3];
            }
        }
    }
    const long t1 = clock64();
    if (lane == 0) out[wid] = t1 - t0;
    if (acc == 12345678) out[63] = acc;
}
int main() {
    long* d; hipMalloc(&d, 64 * 8);
    long h[64];
    const int iters = 2048;
    for (int mode = 0; mode < 3; ++mode)
        for (int waves : {1, 4, 8, 16}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 65536, 0, d, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 65536, 0, d, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * waves), 65536, 0, d, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
            long mx = 0;
            for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
            const double per = (double)mx / (iters * 8.0 * waves);
            printf("mode %d (%s) waves %2d: %.2f clk per wave-instruction per CU  (%.1f B/clk)\n", mode,
                   mode == 0 ? "tr_b64" : (mode == 1 ? "b64" : "b128"), waves, per, (mode == 2 ? 1024.0 : 512.0) / per);
        }
    return 0;
}
