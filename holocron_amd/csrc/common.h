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
// Direct-to-LDS DMA of 16 bytes per lane (buffer_load_dwordx4 ... lds) issued through inline assembly.  The compiler's
// wait-count insertion treats the builtin as an LDS store that later LDS reads and barriers must wait for with
// vmcnt(0), which defeats multi-stage pipelines; with the asm form the kernels own the ordering (explicit
// `s_waitcnt vmcnt(n)` + barrier).  lds_off: wave-uniform LDS byte address (lane i lands at +16 i).
typedef __attribute__((address_space(3))) void hc_lds_void;
__device__ __forceinline__ void hc_dma16(const u32x4 rsrc, unsigned lds_off, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_off), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ u32x4 hc_raw_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    u32x4 r;
    r[0] = (unsigned)a;
    r[1] = (unsigned)(a >> 32) & 0xffffu;
    r[2] = bytes;
    r[3] = 0x00020000u;
    return r;
}
template <int N>
__device__ __forceinline__ void hc_wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ unsigned hc_lds_addr(const void* p) { return (unsigned)(size_t)(hc_lds_void*)p; }

__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned int voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned int voff, unsigned int soff) {   // + scalar offset
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
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

// Zero-fill as a kernel launch.  hipMemsetAsync nodes of a captured stream did not stay ordered against the kernels around
// them when the graph was replayed (a workspace zeroed, accumulated into with atomics and reduced several times per step came
// back with stale contents from the second replay on), so everything that may run under hipGraph capture clears its scratch
// with this kernel instead.
static __global__ void hc_zero_kernel(unsigned char* __restrict__ p, size_t bytes) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    if ((((size_t)p | bytes) & 15) == 0) {
        u32x4* q = reinterpret_cast<u32x4*>(p);
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (size_t i = tid; i < bytes / 16; i += nth) q[i] = z;
    } else if ((((size_t)p | bytes) & 3) == 0) {
        unsigned int* q = reinterpret_cast<unsigned int*>(p);
        for (size_t i = tid; i < bytes / 4; i += nth) q[i] = 0u;
    } else {
        for (size_t i = tid; i < bytes; i += nth) p[i] = 0;
    }
}
static inline hipError_t hc_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    size_t blocks = (bytes / 16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(hc_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned char*)p, bytes);
    return hipGetLastError();
}

// Number of replicas of every per-channel statistics accumulator ([replicas][k][C], workgroup b adds into replica b % replicas).
// Default HC_STAT_REPLICAS; hc_set_deterministic(1) raises it above the largest grid so that every workgroup owns its slot and the
// finalize kernels add the slots in a fixed order (bit-reproducible statistics).  Defined in rep_bn.hip.
extern "C" int hc_get_stat_replicas(void);
extern "C" int hc_get_deterministic(void);

static inline int hc_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? HC_OK : HC_ERR_LAUNCH;
}
