// Streaming 1 x 1 convolution for NARROW inputs (at most 128 channels per pixel) - the expansion convolutions of the inverted
// bottlenecks (holocron/models/classification/rexnet.py:97-103: conv_sequence(in_channels, in_channels * t, kernel_size=1)) and the data
// gradients of their projection convolutions.
//
// The gather-conv (conv_gather.hip) gives such a layer one workgroup per 128 pixels x 64 / 96 / 192 channels: a single k-loop pass
// whose DMA wait, MFMAs and staged stores do not overlap, and the input tile is fetched once per channel tile.  Measured per layer
// (scripts/bench_pointwise.py): 1.2-2.2 TB/s on launches that write 5-12 bytes for every byte they read.
// Here the WEIGHTS are stationary: a wave keeps the bf16 MFMA fragments of 64 output channels x all input channels in registers
// (<= 64 VGPRs) for its whole life and walks 32-pixel tiles; per tile it loads the pixel fragments straight from global memory (16
// bytes per lane and k-step, all issued before the first MFMA), runs 2 x K / 16 MFMAs, swaps accumulator halves between the lane
// pairs so that every lane owns 16 consecutive channels of one pixel (two 16-byte stores), and adds the BatchNorm statistics into
// per-lane registers that are reduced once, at the end.  No LDS, no barriers: latency is hidden by 8-12 resident waves per CU, each
// with its own loads in flight.  The four waves of a workgroup take neighbouring channel groups of the SAME pixel tiles (the input
// tile comes from L1 after the first wave).
#include <cstdlib>
#include "common.h"
#include "../../include/holocron_hip.h"

namespace {

typedef unsigned int u32;

// KS = k-steps of 16 input channels held in registers (srcC <= 16 KS); STATS: per-channel sum / sum of squares of the fp32 results
template <int KS, bool STATS>
__global__ __launch_bounds__(256) void conv_pw_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
                                                      float* __restrict__ stats, const long M, const int Cin, const int Cout,
                                                      const int ngroups, const long ntasks, const int reps) {
    constexpr int OPITCH = 144;                       // 128 B of a pixel's 64 channels + one 16-byte pad chunk
    __shared__ __attribute__((aligned(16))) char stage_all[4 * 32 * OPITCH];
    char* stg = stage_all + (threadIdx.x >> 6) * (32 * OPITCH);
    const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;              // a multiple of ngroups: a wave keeps its channel group
    const int g = (int)(wave0 % ngroups);
    const int ch0 = g * 64;
    const int ks = Cin / 16;
    // weight fragments: A operand, row = output channel ch0 + 32 b + lr, k = 16 s + 8 lh + [0, 8)
    u32x4 wf[2][KS];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int co = ch0 + 32 * b + lr;
            const bool ok = co < Cout && s < ks;
            const u32x4 v = *reinterpret_cast<const u32x4*>(w + (ok ? (size_t)co * Cin + 16 * s + 8 * lh : 0));
            wf[b][s] = u32x4{ok ? v[0] : 0u, ok ? v[1] : 0u, ok ? v[2] : 0u, ok ? v[3] : 0u};
        }
    float s1[2][16], s2[2][16];
    if (STATS) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) s1[b][e] = s2[b][e] = 0.f;
    }
    // The pixel fragments of the NEXT tile are requested before the stores of the current one: memory operations retire in order, so a
    // load issued behind the stores would wait for them to reach memory (4-5 us per tile measured with that order - every tile paid a
    // load AND a store round trip).
    u32x4 xf[KS];
    bool pok;
    long p;
    auto request = [&](long task) __attribute__((always_inline)) {
        const long tile = task / ngroups;                 // (task % ngroups == g for every task of this wave)
        p = tile * 32 + lr;
        pok = task < ntasks && p < M;
#pragma unroll
        for (int s = 0; s < KS; ++s)
            xf[s] = *reinterpret_cast<const u32x4*>(x + ((pok && s < ks) ? (size_t)p * Cin + 16 * s + 8 * lh : 0));
    };
    request(wave0);
    for (long task = wave0; task < ntasks; task += nwaves) {
        const long pcur = p;
        const bool pokc = pok;
        f32x16 acc[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[b][q] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bool ok = pokc && s < ks;
            const u32x4 z = {ok ? xf[s][0] : 0u, ok ? xf[s][1] : 0u, ok ? xf[s][2] : 0u, ok ? xf[s][3] : 0u};
            const bf16x8 bx = __builtin_bit_cast(bf16x8, z);
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[b][s]), bx, acc[b], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        request(task + nwaves);                           // clamped to a valid address past the end
        __builtin_amdgcn_sched_barrier(0);
        // acc[b][q]: channel ch0 + 32 b + (q & 3) + 8 (q >> 2) + 4 lh of pixel p.  The lane pair (lane, lane ^ 32) swaps halves so that
        // lh = 0 ends with channels [0, 16) and lh = 1 with [16, 32) of the block, in order
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float send[8], recv[8], o[16];
#pragma unroll
            for (int e = 0; e < 8; ++e) send[e] = lh ? acc[b][e] : acc[b][8 + e];
#pragma unroll
            for (int e = 0; e < 8; ++e) recv[e] = __shfl_xor(send[e], 32, 64);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = lh ? recv[e] : acc[b][e];
                o[4 + e] = lh ? acc[b][8 + e] : recv[e];
                o[8 + e] = lh ? recv[4 + e] : acc[b][4 + e];
                o[12 + e] = lh ? acc[b][12 + e] : recv[4 + e];
            }
            if (STATS) {
#pragma unroll
                for (int e = 0; e < 16; ++e) { s1[b][e] += o[e]; s2[b][e] += o[e] * o[e]; }   // rows past M multiplied zeros
            }
            // this lane's 16 channels of block b go to the wave's LDS tile [32 pixels][64 channels] (pitch 144 B) ...
            u32x4 lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]); hi[e] = pack_bf16x2(o[8 + 2 * e], o[8 + 2 * e + 1]); }
            char* sp = stg + lr * OPITCH + 64 * b + 32 * lh;
            *reinterpret_cast<u32x4*>(sp) = lo;
            *reinterpret_cast<u32x4*>(sp + 16) = hi;
        }
        // ... and leave it as ROW-CONTIGUOUS 16-byte stores: eight consecutive lanes write the 128 contiguous bytes of one pixel's 64
        // channels, a wave instruction covers 8 whole cache lines.  (Stored straight from the accumulators a lane's two 16-byte pieces
        // sat 32 bytes apart and every store instruction touched 32 lines in quarters: the launches ran at the store-issue rate,
        // 1.3-3.4 TB/s on tensors that are 5-12 x wider than their input.)  One wave, in-order LDS queue: no barrier, only the wait.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {
            const long ptile = pcur - lr;                          // first pixel of the tile
            const int chunk = lane & 7, cch = ch0 + chunk * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int px = 8 * j + (lane >> 3);
                const u32x4 v = *reinterpret_cast<const u32x4*>(stg + px * OPITCH + chunk * 16);
                if (task < ntasks && ptile + px < M && cch < Cout) *reinterpret_cast<u32x4*>(y + (size_t)(ptile + px) * Cout + cch) = v;
            }
        }
        asm volatile("" ::: "memory");
    }
    if (STATS) {
        // sum over the 32 pixel lanes of each half wave, then one atomic per (channel, k) into this wave's replica
        float* rep = stats + (size_t)(wave0 % reps) * 2 * Cout;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float a = s1[b][e], c = s2[b][e];
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
                const int co = ch0 + 32 * b + 16 * lh + e;
                if (lr == 0 && co < Cout) {
                    atomicAdd(rep + co, a);
                    atomicAdd(rep + Cout + co, c);
                }
            }
        }
    }
}

template <int KS>
void pw_launch(const hc_conv_desc& d, long M, int ngroups, long ntasks, unsigned grid, hipStream_t st) {
    const bf16_t* x = reinterpret_cast<const bf16_t*>(d.src0);
    const bf16_t* w = reinterpret_cast<const bf16_t*>(d.wpk);
    bf16_t* y = reinterpret_cast<bf16_t*>(d.dst);
    if (d.stats != nullptr)
        hipLaunchKernelGGL((conv_pw_kernel<KS, true>), dim3(grid), dim3(256), 0, st, x, w, y, d.stats, M, d.srcC, d.Cout, ngroups, ntasks,
                           hc_get_stat_replicas());
    else
        hipLaunchKernelGGL((conv_pw_kernel<KS, false>), dim3(grid), dim3(256), 0, st, x, w, y, (float*)nullptr, M, d.srcC, d.Cout, ngroups,
                           ntasks, hc_get_stat_replicas());
}

}  // namespace

extern "C" {

int hc_conv_pointwise_supported(const hc_conv_desc* dp) {
    if (dp == nullptr) return 0;
    const hc_conv_desc& d = *dp;
    if (d.src0 == nullptr || d.wpk == nullptr || d.dst == nullptr) return 0;
    if (d.nclass != 1 || d.T != 1 || d.src1 != nullptr || d.resid != nullptr || d.bias != nullptr || d.act != 0 || d.pix_scale != nullptr ||
        d.ch_mult != nullptr || d.co_split != 0)
        return 0;
    const hc_conv_class& c = d.cls[0];
    if (c.ntaps != 1 || c.tap[0] != 0 || c.ostep != 1 || c.istep != 1 || c.oy0 != 0 || c.ox0 != 0) return 0;
    if (c.OHg != d.OH || c.OWg != d.OW || d.IH != d.OH || d.IW != d.OW) return 0;
    if (d.srcC <= 0 || d.srcC > 128 || (d.srcC % 16) != 0 || d.Cout <= 0 || (d.Cout % 16) != 0) return 0;
    if (d.stats != nullptr && hc_get_deterministic()) return 0;      // the statistics atomics of several waves share a replica
    if (((reinterpret_cast<unsigned long long>(d.src0) | reinterpret_cast<unsigned long long>(d.wpk) |
          reinterpret_cast<unsigned long long>(d.dst)) & 15ull) != 0)
        return 0;
    return 1;
}

int hc_conv_pointwise(const hc_conv_desc* dp, hc_stream_t stream) {
    if (!hc_conv_pointwise_supported(dp)) return HC_ERR_ARG;
    const hc_conv_desc& d = *dp;
    const long M = (long)d.N * d.OH * d.OW;
    if (M == 0) return HC_OK;
    const int ngroups = (d.Cout + 63) / 64;
    const long tiles = (M + 31) / 32, ntasks = tiles * ngroups;
    // four workgroups (16 waves) per CU where the registers allow; the wave count is a multiple of the channel groups so that a wave
    // keeps its group
    long waves = 256L * 4 * 4;
    if (waves > ntasks) waves = ntasks;
    waves = (waves + 4L * ngroups - 1) / (4L * ngroups) * (4L * ngroups);
    const unsigned grid = (unsigned)(waves / 4);
    hipStream_t st = (hipStream_t)stream;
    const int ks = d.srcC / 16;
    if (ks <= 1) pw_launch<1>(d, M, ngroups, ntasks, grid, st);
    else if (ks <= 2) pw_launch<2>(d, M, ngroups, ntasks, grid, st);
    else if (ks <= 4) pw_launch<4>(d, M, ngroups, ntasks, grid, st);
    else pw_launch<8>(d, M, ngroups, ntasks, grid, st);
    return hc_launch_status();
}

}  // extern "C"
