// L2 / MALL / HBM -> LDS fill-rate probe (round 6): how many bytes per clock can ONE CU ingest through `buffer_load_dwordx4 ... lds`
// (1 KiB per wave instruction) and through plain `buffer_load_dwordx4` into registers, as a function of waves per CU, loads in
// flight per wave and where the data lives.  Every MFMA kernel of this repo is fed this way; the number bounds their tiles.
//   hipcc --offload-arch=gfx950 -O3 -o fill_probe scripts/probes/fill_probe.hip && ./fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ u32x4 raw_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    u32x4 r;
    r[0] = (unsigned)a;
    r[1] = (unsigned)(a >> 32) & 0xffffu;
    r[2] = bytes;
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ void dma16(const u32x4 rsrc, unsigned lds_off, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_off), "v"(voff), "s"(rsrc) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// share: 0 = every workgroup its own region, 1 = the workgroups of an XCD (blockIdx % 8) share one region (weights-like)
// Row mode: one DMA instruction fetches 1024 / rowb rows of `rowb` contiguous bytes each, `pitch` bytes apart (a k-slice of an NHWC
// tensor: rowb = 2 BK bytes of a pixel whose channels span `pitch` bytes) - what the conv kernels' staging really issues.
template <int DEPTH>
__global__ __launch_bounds__(1024) void fill_rows_kernel(const char* src, unsigned region, int share, int iters, int rowb, int pitch,
                                                         unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t rbase = (size_t)(share ? (blockIdx.x & 7) : blockIdx.x) * region;
    const u32x4 rs = raw_rsrc(src + rbase, region);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem + (unsigned)wid * DEPTH * 1024u;
    const int lpr = rowb / 16, rows = 64 / lpr;              // lanes per row, rows per instruction
    unsigned off = ((unsigned)(wid * rows + lane / lpr) * (unsigned)pitch + (unsigned)(lane % lpr) * 16u) % region;
    const unsigned stride = (unsigned)(nw * rows) * (unsigned)pitch;
    for (int i = 0; i < iters; ++i) {
        wait_vm<DEPTH - 1>();
        dma16(rs, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(i % DEPTH) * 1024u), off);
        off += stride;
        if (off >= region) off -= region;
    }
    wait_vm<0>();
    if (*reinterpret_cast<unsigned*>(smem + wid * DEPTH * 1024 + lane * 4) == 0x12345678u) sink[0] = 1;
}

template <int DEPTH, bool TO_LDS>
__global__ __launch_bounds__(1024) void fill_kernel(const char* src, unsigned region, int share, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t rbase = (size_t)(share ? (blockIdx.x & 7) : blockIdx.x) * region;
    const u32x4 rs = raw_rsrc(src + rbase, region);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src + rbase), 0, region, 0x00020000);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem + (unsigned)wid * DEPTH * 1024u;
    unsigned off = ((unsigned)wid * 1024u + (unsigned)lane * 16u) % region;
    const unsigned stride = (unsigned)nw * 1024u;
    u32x4 acc = {0, 0, 0, 0};
    if (TO_LDS) {
        for (int i = 0; i < iters; ++i) {
            wait_vm<DEPTH - 1>();
            dma16(rs, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(i % DEPTH) * 1024u), off);
            off += stride;
            if (off >= region) off -= region;
        }
        wait_vm<0>();
        acc[0] = *reinterpret_cast<unsigned*>(smem + wid * DEPTH * 1024 + lane * 4);
    } else {
        u32x4 v[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) v[k] = u32x4{0, 0, 0, 0};
        for (int i = 0; i < iters; i += DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                acc ^= v[k];
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsb, off, 0, 0);
                off += stride;
                if (off >= region) off -= region;
            }
        }
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) acc ^= v[k];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int DEPTH, bool TO_LDS>
double run(const char* src, unsigned region, int share, int waves, int wgs, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned* sink; hipMalloc(&sink, 4);
    auto k = fill_kernel<DEPTH, TO_LDS>;
    const int smem = waves * DEPTH * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(64 * waves), smem, 0, src, region, share, iters, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(64 * waves), smem, 0, src, region, share, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(sink);
    return (double)wgs * waves * iters * 1024.0 / (ms * 1e-3);
}

double run_rows(const char* src, unsigned region, int share, int waves, int wgs, int iters, int rowb, int pitch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned* sink; hipMalloc(&sink, 4);
    auto k = fill_rows_kernel<4>;
    const int smem = waves * 4 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(64 * waves), smem, 0, src, region, share, iters, rowb, pitch, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(64 * waves), smem, 0, src, region, share, iters, rowb, pitch, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(sink);
    return (double)wgs * waves * iters * 1024.0 / (ms * 1e-3);
}

int main(int argc, char** argv) {
    if (argc > 1) {          // ./fill_probe rows: the row-strided table only
        const size_t total = (size_t)1 << 30;
        char* src; hipMalloc(&src, total);
        hipMemset(src, 1, total);
        printf("%-34s %6s %6s %6s %10s %12s %10s\n", "source (L2-resident)", "rowB", "pitch", "waves", "TB/s chip", "GB/s per CU", "B/clk/CU");
        for (int share = 1; share >= 0; --share)
            for (int rowb : {1024, 256, 128, 64, 32})
                for (int pitch : {0, 256, 512, 2560}) {
                    const int pt = pitch == 0 ? rowb : pitch;
                    if (pt < rowb) continue;
                    for (int waves : {4, 8}) {
                        const unsigned region = share ? (1u << 20) : (256u << 10);
                        const double bps = run_rows(src, region, share, waves, 256, 4096, rowb, pt);
                        printf("%-34s %6d %6d %6d %10.2f %12.1f %10.1f\n", share ? "1 MB shared per XCD" : "256 KB private per WG (64 MB: MALL)", rowb, pt, waves,
                               bps / 1e12, bps / 256 / 1e9, bps / 256 / 2.4e9);
                    }
                }
        hipFree(src);
        return 0;
    }
    const size_t total = (size_t)3 << 30;
    char* src; hipMalloc(&src, total);
    hipMemset(src, 1, total);
    struct Case { const char* name; unsigned region; int share; };
    const Case cases[] = {{"L2 shared 1 MB per XCD", 1u << 20, 1}, {"L2 private 64 KB per WG", 64u << 10, 0},
                          {"MALL private 256 KB per WG (64 MB)", 256u << 10, 0}, {"HBM private 8 MB per WG (2 GB)", 8u << 20, 0}};
    printf("%-40s %6s %6s %6s %10s %12s %10s\n", "source", "path", "waves", "depth", "TB/s chip", "GB/s per CU", "B/clk/CU");
    for (const Case& c : cases) {
        for (int lds = 1; lds >= 0; --lds) {
            for (int waves : {4, 8, 16}) {
                for (int depth : {2, 4, 8}) {
                    if (waves * depth * 1024 > 150 * 1024) continue;
                    const int iters = c.region >= (8u << 20) ? 2048 : 4096;
                    double bps = 0;
                    if (lds) bps = depth == 2 ? run<2, true>(src, c.region, c.share, waves, 256, iters) : depth == 4 ? run<4, true>(src, c.region, c.share, waves, 256, iters) : run<8, true>(src, c.region, c.share, waves, 256, iters);
                    else bps = depth == 2 ? run<2, false>(src, c.region, c.share, waves, 256, iters) : depth == 4 ? run<4, false>(src, c.region, c.share, waves, 256, iters) : run<8, false>(src, c.region, c.share, waves, 256, iters);
                    printf("%-40s %6s %6d %6d %10.2f %12.1f %10.1f\n", c.name, lds ? "lds" : "vgpr", waves, depth, bps / 1e12, bps / 256 / 1e9, bps / 256 / 2.4e9);
                }
            }
        }
    }
    // two workgroups per CU (the classic conv form): 512 workgroups of 4 waves
    for (const Case& c : cases) {
        const double bps = run<4, true>(src, c.region, c.share, 4, 512, 4096);
        printf("%-40s %6s %6s %6d %10.2f %12.1f %10.1f\n", c.name, "lds", "2x4", 4, bps / 1e12, bps / 256 / 1e9, bps / 256 / 2.4e9);
    }
    hipFree(src);
    return 0;
}
