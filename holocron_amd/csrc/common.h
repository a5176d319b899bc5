// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
// Everything here is wave64 / MFMA / LDS specific; there is no other backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HC_OK 0
#define HC_ERR_ARG 1
#define HC_ERR_LAUNCH 2

typedef unsigned short bf16_t;  // raw bf16 bits in HBM
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define HC_OOB 0xFFFFFFF0u  // voffset that is out of range for every buffer descriptor we build

__device__ __forceinline__ float bf16_to_f32(unsigned short b) {
    return __builtin_bit_cast(float, (unsigned int)b << 16);
}
// fp32 -> bf16, round-to-nearest-even (same rule as torch's float -> bfloat16), on the gfx950
// hardware converter: ONE v_cvt_pk_bf16_f32 per two values (a software RNE costs ~5 VALU per value,
// which made every epilogue and every elementwise pass VALU-bound).
typedef __attribute__((ext_vector_type(2))) __bf16 hc_bf16x2;
typedef __attribute__((ext_vector_type(2))) float hc_f32x2;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    const hc_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, hc_bf16x2));
}
__device__ __forceinline__ unsigned short f32_to_bf16(float f) { return (unsigned short)(pack_bf16x2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf16lo(unsigned int w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16hi(unsigned int w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned int voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
}

// LDS tile [rows][BK] bf16 with a 16-byte-chunk XOR swizzle so that the ds_read_b128 fragment
// reads (lane = row, fixed chunk) and the ds_write_b128 staging writes (lane = chunk within a
// row) are both bank-conflict free.  Derivation in DESIGN.md "LDS image".
template <int BK>
__device__ __forceinline__ int lds_off(int row, int chunk) {
    constexpr int NC = BK / 8;   // 16-byte chunks per row
    constexpr int RP = 16 / NC;  // rows per 256-byte bank row
    return row * (BK * 2) + ((chunk ^ ((row / RP) % NC)) << 4);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

static inline int hc_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? HC_OK : HC_ERR_LAUNCH;
}
