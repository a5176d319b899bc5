// Operand layout / scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3, OCP) operands and unit scales.
// Hypothesis: lane l holds 32 consecutive k of row/col (l & 31), k = 32 * (l >> 5) + [0, 32); C/D as every 32x32 MFMA.
//   hipcc --offload-arch=gfx950 -O2 mx_fp8_probe.hip -o mx_fp8_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k(const uint8_t* A, const uint8_t* B, float* C, int scale_a, int scale_b) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    const int row = lane & 31, k0 = 32 * (lane >> 5);
    for (int i = 0; i < 8; ++i) {
        a[i] = *(const int*)(A + row * 64 + k0 + 4 * i);     // A[row][k]
        b[i] = *(const int*)(B + row * 64 + k0 + 4 * i);     // B stored as [col][k]
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), n = lane & 31;
        C[m * 32 + n] = c[r];
    }
}
static float fp8_to_f(uint8_t v) {   // OCP e4m3fn
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 0) f = ldexpf((float)m, -9);
    else if (e == 15 && m == 7) f = NAN;
    else f = ldexpf(1.f + m / 8.f, e - 7);
    return s ? -f : f;
}
int main() {
    uint8_t hA[32 * 64], hB[32 * 64];
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) {
        hA[i] = (uint8_t)((rand() % 2 ? 0x80 : 0) | ((rand() % 6 + 4) << 3) | (rand() % 8));   // exponents 4..9: values 1/8 .. 7.5
        hB[i] = (uint8_t)((rand() % 2 ? 0x80 : 0) | ((rand() % 6 + 4) << 3) | (rand() % 8));
    }
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, 32 * 32 * 4);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    for (int sc : {0x7f, 0x7f7f7f7f, 0x80, 0}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, sc, sc);
        float hC[32 * 32];
        hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0, ratio = 0;
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n) {
                double ref = 0;
                for (int kk = 0; kk < 64; ++kk) ref += (double)fp8_to_f(hA[m * 64 + kk]) * fp8_to_f(hB[n * 64 + kk]);
                maxerr = fmax(maxerr, fabs(ref - hC[m * 32 + n]));
                maxref = fmax(maxref, fabs(ref));
                if (m == 3 && n == 5) ratio = hC[m * 32 + n] / ref;
            }
        printf("scale 0x%x: max |err| %.4g (max |ref| %.4g), C[3][5]/ref = %.4g\n", sc, maxerr, maxref, ratio);
    }
    return 0;
}
