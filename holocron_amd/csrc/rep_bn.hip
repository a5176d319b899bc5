// HBM-bound passes around the MFMA convs: RepBlock training-mode BatchNorm fusion
// (reference: holocron/models/classification/repvgg.py:71-73 -> three nn.BatchNorm2d, a python
// sum() and one shared ReLU), global average pool, layout changes and weight packing.
//
// All activation tensors are NHWC bf16; a thread owns one 16-byte chunk (8 channels) per
// iteration and the launch is sized so that a thread's channel group never changes, which
// keeps the per-channel coefficients in registers.
#include "common.h"
#include "../../include/holocron_hip.h"
#include <stdlib.h>

namespace {

constexpr int EW_THREADS = 256;

struct EwGrid {
    int blocks;
};
// number of blocks such that (blocks*256) % (C/8) == 0 and every thread gets >= ~8 chunks
inline int ew_blocks(long nchunks, int cg, int cpt_default = 8) {
    int a = cg, b = EW_THREADS;
    while (b) { int t = a % b; a = b; b = t; }
    const int unit = cg / a;  // blocks must be a multiple of this
    // chunks per thread: 8 for pure streaming passes; 16 for passes that end in a per-workgroup
    // reduction (their tail costs as much as streaming a few MB, measured with scripts/bench_ew.py)
    const int cpt = cpt_default;
    long want = nchunks / (EW_THREADS * cpt);
    // mid-size tensors: fill the chip (at least ~4 workgroups per CU) as long as every thread still has 2 chunks
    constexpr long fl = 1024;                    // 512 / 2048 measured inside the noise (round 4)
    long floor_blocks = nchunks / (EW_THREADS * 2);
    if (floor_blocks > fl) floor_blocks = fl;
    if (want < floor_blocks) want = floor_blocks;
    if (want > 2048) want = 2048;
    if (want < 1) want = 1;
    long k = (want + unit - 1) / unit;
    return (int)(k * unit);
}

__device__ __forceinline__ void unpack8(const u32x4 v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bf16lo(v[i]); f[2 * i + 1] = bf16hi(v[i]); }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
    return v;
}
__device__ __forceinline__ void load8f(const float* p, float (&f)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
}

// Streaming accesses of the elementwise passes.  NT: tensors far larger than the 256 MB last-level cache (the 112x112 and 56x56
// stages at batch 256: 77-308 MB each, 4-7 of them per pass) are read and written with the non-temporal hint - nothing of them
// survives until its next use anyway, and not allocating them sped those passes up by 8-12 %; on the small stages the same hint
// costs 3-18 % (their operands ARE still in cache from the producing kernel), so the host picks per launch.
template <bool NT>
__device__ __forceinline__ u32x4 ld16(const u32x4* p) {
    return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT>
__device__ __forceinline__ void st16(u32x4* p, const u32x4 v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
// single-branch passes (bn_act_apply / bn_act_bwd_apply): streaming form from 200 MB on.  Step-level A/B, same box, off / 200 / 48 MB:
// rexnet1_0x 18.36-18.43 / 18.18-18.23 / 18.20-18.24 ms, yolov4 28.65-28.67 / 28.60-28.68 / 28.71-28.75 ms (its 95-190 MB tensors are still in
// the last-level cache when the pass reads them: the hint costs there)
static inline bool ba_streaming(long nchunks) { return nchunks * 16 >= 200L * 1000000L; }
static inline bool ew_streaming(long nchunks) {      // one tensor >= 48 MB
    constexpr long thr = 48;                     // 32 / 96 measured inside the noise (round 4)
    return thr >= 0 && nchunks * 16 >= thr * 1000000L;
}

// Per-workgroup reduction of per-thread partial sums v[S][8] (8 channels of the thread's channel
// group) WITHOUT LDS atomics (contended ds_add_f32 costs ~8 us per workgroup here): partials go to
// LDS [thread][S*8 (+1 pad)], then each output (sum k, channel c) adds the ~256/cg threads that own
// channel group c/8 and issues ONE global atomic into the replica slab `dst` ([S][C]).
template <int S>
__device__ __forceinline__ void block_reduce_flush(const float (&v)[S][8], int cg, int C, float* __restrict__ dst,
                                                   float* __restrict__ lds, int nsum) {
    constexpr int STR = S * 8 + 1;
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < S; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) lds[tid * STR + k * 8 + e] = v[k][e];
    __syncthreads();
    const int blockbase = (int)(((long)blockIdx.x * EW_THREADS) % cg);   // channel group of thread 0
    for (int o = tid; o < nsum * C; o += EW_THREADS) {
        const int k = o / C, c = o - k * C;
        const int g = c >> 3, e = c & 7;
        int first = g - blockbase;
        if (first < 0) first += cg;
        float sum = 0.f;
        for (int t = first; t < EW_THREADS; t += cg) sum += lds[t * STR + k * 8 + e];
        atomicAdd(dst + (size_t)k * C + c, sum);
    }
}

// ---------------------------------------------------------------- forward BN finalize
// A workgroup finishes 16 channels: thread (rg, ch) = (tid >> 4, tid & 15) adds the replicas rg, rg + 16, ... of channel ch - 16
// consecutive channels per replica row are one 64-byte run, where the former one-wave-per-channel sweep touched 64 cache lines per
// load instruction (10 us per launch in the round-2 trace for 590 KB of statistics) - the 16 partial sums meet in LDS in a fixed
// order and the first 16 threads write the coefficients.  All loads of a thread are independent (one memory round trip); the order
// of the additions is fixed, so the deterministic mode's single-writer slots still give bit-identical steps.
constexpr int FIN_CH = 16, FIN_RG = 16;
__global__ __launch_bounds__(256) void rep_bn_finalize_kernel(const hc_rep_bn_desc d, const int reps) {
    __shared__ float sm[FIN_RG][6][FIN_CH + 1];
    const int ch = threadIdx.x & (FIN_CH - 1), rg = threadIdx.x >> 4;
    const int c = blockIdx.x * FIN_CH + ch;
    const bool cin = c < d.C;
    const float cnt = (float)d.count;
    // channels >= c_valid are layout padding: zero affine, no parameters / running statistics behind them
    const bool chan_ok = cin && (d.c_valid <= 0 || c < d.c_valid);
    bool on[3];
    float gam[3], bet[3], rm[3], rv[3], s1[3], s2[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        on[b] = d.gamma[b] != nullptr && chan_ok;
        s1[b] = s2[b] = 0.f;
        // unconditional loads (a dead branch / channel reads branch 0's live address and is ignored): requested here, used after the
        // replica sweep - no branch, no wait in front of the sweep
        const int cs = cin ? c : d.C - 1;
        const float* dummy = d.gamma[0] + cs;
        gam[b] = *(on[b] ? d.gamma[b] + c : dummy);
        bet[b] = *(on[b] ? d.beta[b] + c : dummy);
        rm[b] = *((on[b] && d.running_mean[b] != nullptr) ? d.running_mean[b] + c : dummy);
        rv[b] = *((on[b] && d.running_mean[b] != nullptr) ? d.running_var[b] + c : dummy);
    }
    if (d.training) {
        // branch-free loads: a condition per load makes the compiler branch around it and drain vmcnt behind every pair - 24 serial
        // round trips (14 us measured).  A dead branch / channel reads a live one's address instead and is zeroed afterwards.
        const float* sp[3];
        const int cc = cin ? c : d.C - 1;
#pragma unroll
        for (int b = 0; b < 3; ++b) sp[b] = (d.stats[b] != nullptr ? d.stats[b] : d.stats[0]) + cc;
        const size_t row = (size_t)d.C;
        // eight replicas per round: 48 loads requested, THEN added (left alone the compiler pairs every load with its add behind a
        // vmcnt(0..3) - a handful of loads in flight); rounds past the end re-read replica rg and add zero
        for (int r0 = rg; r0 < reps; r0 += FIN_RG * 8) {
            float v[8][6];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + u * FIN_RG;
                const size_t rr = (size_t)(r < reps ? r : rg);
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    v[u][2 * b] = sp[b][(2 * rr) * row];
                    v[u][2 * b + 1] = sp[b][(2 * rr + 1) * row];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float keep = (r0 + u * FIN_RG) < reps ? 1.f : 0.f;
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    s1[b] += keep * v[u][2 * b];
                    s2[b] += keep * v[u][2 * b + 1];
                }
            }
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            sm[rg][2 * b][ch] = on[b] ? s1[b] : 0.f;
            sm[rg][2 * b + 1][ch] = on[b] ? s2[b] : 0.f;
        }
    }
    __syncthreads();
    if (rg != 0 || !cin) return;
    float shift = 0.f;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        float a = 0.f, mean = 0.f, invstd = 0.f;
        if (on[b]) {
            if (d.training) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int q = 0; q < FIN_RG; ++q) {
                    t1 += sm[q][2 * b][ch];
                    t2 += sm[q][2 * b + 1][ch];
                }
                mean = t1 / cnt;
                float var = t2 / cnt - mean * mean;
                var = var > 0.f ? var : 0.f;
                invstd = rsqrtf(var + d.eps);
                if (d.running_mean[b] != nullptr) {
                    const float unb = d.count > 1 ? var * (cnt / (cnt - 1.f)) : var;
                    d.running_mean[b][c] = (1.f - d.momentum) * rm[b] + d.momentum * mean;
                    d.running_var[b][c] = (1.f - d.momentum) * rv[b] + d.momentum * unb;
                }
                if (c == 0 && d.num_batches_tracked[b] != nullptr) d.num_batches_tracked[b][0] += 1;
            } else {
                mean = rm[b];
                invstd = rsqrtf(rv[b] + d.eps);
            }
            a = gam[b] * invstd;
            shift += bet[b] - a * mean;
        }
        d.coef[b * d.C + c] = a;
        if (d.save != nullptr) {
            d.save[(2 * b) * d.C + c] = mean;
            d.save[(2 * b + 1) * d.C + c] = invstd;
        }
    }
    d.coef[3 * d.C + c] = shift;
}

// ---------------------------------------------------------------- forward apply
// pre-activation of the fused block output.  ONE explicit fma chain used by the forward apply and by the backward kernels that
// recompute the ReLU mask from it (hc_rep_bwd_*_z): the same inputs give the same bits, so (z > 0) there IS (out > 0) here
// (out = bf16(relu(z)) > 0 exactly when z > 0: a positive fp32 never rounds to a bf16 zero above 2^-134).
template <bool HAS_ID>
__device__ __forceinline__ float rep_preact(float a3, float f3, float a1, float f1, float a0, float f0, float sh) {
    float z = __builtin_fmaf(a1, f1, sh);
    if (HAS_ID) z = __builtin_fmaf(a0, f0, z);
    return __builtin_fmaf(a3, f3, z);
}
template <bool HAS_ID, bool STATS, bool NT>
__global__ __launch_bounds__(EW_THREADS) void rep_apply_kernel(const u32x4* __restrict__ y3, const u32x4* __restrict__ y1,
                                                               const u32x4* __restrict__ x, const float* __restrict__ coef,
                                                               u32x4* __restrict__ out, float* __restrict__ out_stats,
                                                               long nchunks, int C, int act, const int reps) {
    extern __shared__ float sred[];  // [2][C] when STATS
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * EW_THREADS + threadIdx.x;
    const long stride = (long)gridDim.x * EW_THREADS;
    const int c0 = (int)(gtid % cg) * 8;
    float a3[8], a1[8], a0[8], sh[8];
    load8f(coef + c0, a3);
    load8f(coef + C + c0, a1);
    if (HAS_ID) load8f(coef + 2 * C + c0, a0);
    load8f(coef + 3 * C + c0, sh);
    float sv[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sv[0][i] = sv[1][i] = 0.f;
    for (long q = gtid; q < nchunks; q += stride) {
        float f3[8], f1[8], f0[8], o[8];
        unpack8(ld16<NT>(y3 + q), f3);
        unpack8(ld16<NT>(y1 + q), f1);
        if (HAS_ID) unpack8(ld16<NT>(x + q), f0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float z = rep_preact<HAS_ID>(a3[i], f3[i], a1[i], f1[i], HAS_ID ? a0[i] : 0.f, HAS_ID ? f0[i] : 0.f, sh[i]);
            if (act == 1) z = z > 0.f ? z : 0.f;
            o[i] = z;
        }
        const u32x4 pk = pack8(o);
        st16<NT>(out + q, pk);
        if (STATS) {  // statistics of what the next block will actually read (bf16-rounded)
            float r[8];
            unpack8(pk, r);
#pragma unroll
            for (int i = 0; i < 8; ++i) { sv[0][i] += r[i]; sv[1][i] += r[i] * r[i]; }
        }
    }
    if (STATS) block_reduce_flush<2>(sv, cg, C, out_stats + (size_t)(blockIdx.x % reps) * 2 * C, sred, 2);
}

// per-channel sum / sum of squares of an NHWC bf16 tensor (identity-branch BN statistics when the
// producer did not emit them)
__global__ __launch_bounds__(EW_THREADS) void channel_stats_kernel(const u32x4* __restrict__ x, float* __restrict__ stats,
                                                                   long nchunks, int C, const int reps) {
    extern __shared__ float sred[];  // [2][C]
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * EW_THREADS + threadIdx.x;
    const long stride = (long)gridDim.x * EW_THREADS;
    const int c0 = (int)(gtid % cg) * 8;
    float sv[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sv[0][i] = sv[1][i] = 0.f;
    for (long q = gtid; q < nchunks; q += stride) {
        float f[8];
        unpack8(x[q], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) { sv[0][i] += f[i]; sv[1][i] += f[i] * f[i]; }
    }
    (void)c0;
    block_reduce_flush<2>(sv, cg, C, stats + (size_t)(blockIdx.x % reps) * 2 * C, sred, 2);
}

// ---------------------------------------------------------------- backward reduce
// ZMASK: the ReLU mask is recomputed from the pre-activation (coef = the forward's [4][C] affine; act 0 = no activation) instead of
// read from `out` - one tensor less per pass (2 of the 13 tensor passes of a block's BatchNorm backward)
template <bool HAS_ID, bool ZMASK, bool NT>
__global__ __launch_bounds__(EW_THREADS) void rep_bwd_reduce_kernel(const u32x4* __restrict__ g, const u32x4* __restrict__ out,
                                                                    const u32x4* __restrict__ y3, const u32x4* __restrict__ y1,
                                                                    const u32x4* __restrict__ x, const float* __restrict__ coef,
                                                                    const int act, float* __restrict__ red,
                                                                    long nchunks, int C, const int reps) {
    extern __shared__ float sred[];  // [4][C]
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * EW_THREADS + threadIdx.x;
    const long stride = (long)gridDim.x * EW_THREADS;
    const int c0 = (int)(gtid % cg) * 8;
    float a3[8], a1[8], a0[8], sh[8];
    if (ZMASK) {
        load8f(coef + c0, a3);
        load8f(coef + C + c0, a1);
        if (HAS_ID) load8f(coef + 2 * C + c0, a0);
        load8f(coef + 3 * C + c0, sh);
    }
    float sv[4][8];   // dz, dz*y3, dz*y1, dz*x
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) sv[k][i] = 0.f;
    for (long q = gtid; q < nchunks; q += stride) {
        float fg[8], fo[8], f3[8], f1[8], f0[8];
        unpack8(ld16<NT>(g + q), fg);
        if (!ZMASK) unpack8(ld16<NT>(out + q), fo);
        unpack8(ld16<NT>(y3 + q), f3);
        unpack8(ld16<NT>(y1 + q), f1);
        if (HAS_ID) unpack8(ld16<NT>(x + q), f0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (ZMASK) fo[i] = act == 1 ? rep_preact<HAS_ID>(a3[i], f3[i], a1[i], f1[i], HAS_ID ? a0[i] : 0.f, HAS_ID ? f0[i] : 0.f, sh[i]) : 1.f;
            const float dz = fo[i] > 0.f ? fg[i] : 0.f;
            sv[0][i] += dz;
            sv[1][i] += dz * f3[i];
            sv[2][i] += dz * f1[i];
            if (HAS_ID) sv[3][i] += dz * f0[i];
        }
    }
    (void)c0;
    block_reduce_flush<4>(sv, cg, C, red + (size_t)(blockIdx.x % reps) * 4 * C, sred, HAS_ID ? 4 : 3);
}

__global__ __launch_bounds__(256) void rep_bn_bwd_finalize_kernel(const hc_rep_bn_bwd_desc d, const int reps) {
    // same shape as the forward finalize: 16 channels per workgroup, 16 replica groups, coalesced replica rows, fixed-order LDS combine
    __shared__ float sm[FIN_RG][4][FIN_CH + 1];
    const int ch = threadIdx.x & (FIN_CH - 1), rg = threadIdx.x >> 4;
    const int c = blockIdx.x * FIN_CH + ch;
    const bool cin = c < d.C;
    const float cnt = (float)d.count;
    const int nb = d.has_identity ? 3 : 2;
    bool on[3];
    float gam[3], mean_[3], inv_[3], dg0[3], db0[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        on[b] = cin && b < nb && d.gamma[b] != nullptr && (d.c_valid <= 0 || c < d.c_valid);
        gam[b] = mean_[b] = inv_[b] = dg0[b] = db0[b] = 0.f;
        if (on[b] && rg == 0) {
            gam[b] = d.gamma[b][c];
            mean_[b] = d.save[(2 * b) * d.C + c];
            inv_[b] = d.save[(2 * b + 1) * d.C + c];
            if (d.accumulate) {
                if (d.dgamma[b] != nullptr) dg0[b] = d.dgamma[b][c];
                if (d.dbeta[b] != nullptr) db0[b] = d.dbeta[b][c];
            }
        }
    }
    float rsum[4] = {0.f, 0.f, 0.f, 0.f};
    {
        const float* rp = d.red + (cin ? c : d.C - 1);
        const size_t row = (size_t)d.C;
        for (int r0 = rg; r0 < reps; r0 += FIN_RG * 8) {     // 32 loads requested, then added (see the forward finalize)
            float v[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + u * FIN_RG;
                const size_t rr = (size_t)(r < reps ? r : rg);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[u][k] = rp[(rr * 4 + k) * row];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float keep = ((r0 + u * FIN_RG) < reps && cin) ? 1.f : 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) rsum[k] += keep * v[u][k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) sm[rg][k][ch] = rsum[k];
    __syncthreads();
    if (rg != 0 || !cin) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < FIN_RG; ++q) t += sm[q][k][ch];
        rsum[k] = t;
    }
    const float sdz = rsum[0];
    for (int b = 0; b < 3; ++b) {
        float A = 0.f, B = 0.f, Cc = 0.f;
        if (on[b]) {
            const float mean = mean_[b], invstd = inv_[b];
            const float sdzy = rsum[b + 1];
            const float dgamma = invstd * (sdzy - mean * sdz);
            const float a = gam[b] * invstd;
            A = a;
            if (!d.frozen) {            // batch statistics: the two centring terms of BatchNorm's backward
                B = -a * invstd * dgamma / cnt;
                Cc = -a * sdz / cnt - B * mean;
            }
            if (d.dgamma[b] != nullptr) d.dgamma[b][c] = d.accumulate ? dg0[b] + dgamma : dgamma;
            if (d.dbeta[b] != nullptr) d.dbeta[b][c] = d.accumulate ? db0[b] + sdz : sdz;
        }
        d.bcoef[(3 * b) * d.C + c] = A;
        d.bcoef[(3 * b + 1) * d.C + c] = B;
        d.bcoef[(3 * b + 2) * d.C + c] = Cc;
    }
}

template <bool HAS_ID, bool ZMASK, bool NT>
__global__ __launch_bounds__(EW_THREADS) void rep_bwd_apply_kernel(const u32x4* __restrict__ g, const u32x4* __restrict__ out,
                                                                   const u32x4* __restrict__ y3, const u32x4* __restrict__ y1,
                                                                   const u32x4* __restrict__ x, const float* __restrict__ coef,
                                                                   const int act, const float* __restrict__ bc,
                                                                   u32x4* __restrict__ dy3, u32x4* __restrict__ dy1,
                                                                   u32x4* __restrict__ dxid, long nchunks, int C) {
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * EW_THREADS + threadIdx.x;
    const long stride = (long)gridDim.x * EW_THREADS;
    const int c0 = (int)(gtid % cg) * 8;
    float a3[8], a1[8], a0[8], sh[8];
    if (ZMASK) {
        load8f(coef + c0, a3);
        load8f(coef + C + c0, a1);
        if (HAS_ID) load8f(coef + 2 * C + c0, a0);
        load8f(coef + 3 * C + c0, sh);
    }
    float A3[8], B3[8], C3[8], A1[8], B1[8], C1[8], A0[8], B0[8], C0[8];
    load8f(bc + 0 * C + c0, A3); load8f(bc + 1 * C + c0, B3); load8f(bc + 2 * C + c0, C3);
    load8f(bc + 3 * C + c0, A1); load8f(bc + 4 * C + c0, B1); load8f(bc + 5 * C + c0, C1);
    if (HAS_ID) { load8f(bc + 6 * C + c0, A0); load8f(bc + 7 * C + c0, B0); load8f(bc + 8 * C + c0, C0); }
    for (long q = gtid; q < nchunks; q += stride) {
        float fg[8], fo[8], f3[8], f1[8], f0[8], o3[8], o1[8], o0[8];
        unpack8(ld16<NT>(g + q), fg);
        if (!ZMASK) unpack8(ld16<NT>(out + q), fo);
        unpack8(ld16<NT>(y3 + q), f3);
        unpack8(ld16<NT>(y1 + q), f1);
        if (HAS_ID) unpack8(ld16<NT>(x + q), f0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (ZMASK) fo[i] = act == 1 ? rep_preact<HAS_ID>(a3[i], f3[i], a1[i], f1[i], HAS_ID ? a0[i] : 0.f, HAS_ID ? f0[i] : 0.f, sh[i]) : 1.f;
            const float dz = fo[i] > 0.f ? fg[i] : 0.f;
            o3[i] = __builtin_fmaf(A3[i], dz, __builtin_fmaf(B3[i], f3[i], C3[i]));   // explicit chains: every instantiation rounds alike
            o1[i] = __builtin_fmaf(A1[i], dz, __builtin_fmaf(B1[i], f1[i], C1[i]));
            if (HAS_ID) o0[i] = __builtin_fmaf(A0[i], dz, __builtin_fmaf(B0[i], f0[i], C0[i]));
        }
        st16<NT>(dy3 + q, pack8(o3));
        st16<NT>(dy1 + q, pack8(o1));
        if (HAS_ID) st16<NT>(dxid + q, pack8(o0));
    }
}

// ---------------------------------------------------------------- generic conv -> BN -> activation (+ residual)
// The building block of conv_sequence() (holocron/models/utils.py:61-84): y = conv(x) comes from the
// gather-conv with the statistics epilogue, rep_bn_finalize (single branch) gives (a, shift);
// out = act(a*y + shift) [+ res].  Backward recomputes z = a*y + shift instead of storing it.
__device__ __forceinline__ float act_fwd(float v, int act, float slope) {
    switch (act) {
        case 1: return v > 0.f ? v : 0.f;
        case 2: { const float t = fminf(fmaxf(v + 2.f, 0.f), 2.f); return 0.5f * v * t; }
        case 3: return v > 0.f ? v : slope * v;
        case 4: {   // mish = x tanh(softplus x) = x n / (n + 2), n = e^x (e^x + 2): one v_exp_f32 and one v_rcp_f32
            const float e = __expf(fminf(v, 20.f)), n = e * (e + 2.f);
            return v * n * __builtin_amdgcn_rcpf(n + 2.f);
        }
        case 5: return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
        case 6: return fminf(fmaxf(v, 0.f), 6.f);
        default: return v;
    }
}
__device__ __forceinline__ float act_bwd(float v, int act, float slope) {   // d act / d v
    switch (act) {
        case 1: return v > 0.f ? 1.f : 0.f;
        case 2: { const float t = v + 2.f; float d = 0.5f * fminf(fmaxf(t, 0.f), 2.f); if (t >= 0.f && t <= 2.f) d += 0.5f * v; return d; }
        case 3: return v > 0.f ? 1.f : slope;
        case 4: {   // d/dx [x t], t = n / (n + 2) = 1 - 2 r, r = 1 / (n + 2), n = e (e + 2) = u + e, u = e (e + 1): t' = 4 u r^2, so
            // mish' = 1 + r (4 x u r - 2): eleven instructions instead of fifteen - these passes are VALU-bound on YOLOv4
            const float e = __expf(fminf(v, 20.f)), u = __builtin_fmaf(e, e, e);
            const float r = __builtin_amdgcn_rcpf((u + e) + 2.f);
            return __builtin_fmaf(r, __builtin_fmaf(v * u, 4.f * r, -2.f), 1.f);
        }
        case 5: { const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-v)); return sg * (1.f + v * (1.f - sg)); }
        case 6: return (v > 0.f && v < 6.f) ? 1.f : 0.f;
        default: return 1.f;
    }
}

// DropBlock after the activation (conv_sequence order conv -> BN -> act -> DropBlock, models/utils.py:75-84) rides
// in the same pass: keep [npix] fp32 0/1 and count[1] = sum(keep) come from hc_dropblock_mask.
__device__ __forceinline__ float drop_scale(const float* __restrict__ count, long npix) {
    const float cnt = count[0];
    return cnt > 0.f ? (float)npix / cnt : 1.f;
}

// The output (apply) and the incoming gradient (backward) may live inside a wider concat buffer: `ld8` is their
// channels-per-pixel / 8 (the pointer already includes the channel offset).
// U pixels of a thread's channel group per iteration, all their loads requested before the first is used, and (NT) the non-temporal
// hint for tensors far larger than the last-level cache - what the RepBlock passes have had since round 2.  With one pixel per iteration
// a thread had one or two 16-byte loads in flight and the passes ran at 4.2-4.4 TB/s on the 257-617 MB tensors of ReXNet (round 5).
// ACT: the activation code as a COMPILE-TIME constant (round 6).  With the runtime switch the compiler kept a chain of scalar compares
// and branches around every one of a thread's eight elements (108-114 s_cbranch per kernel body, two v_exp_f32 per element in the
// text): the small, cache-resident tensors of YOLOv4 (280 launches of 10-19 us per step) were issue-bound on that chain.
template <bool HAS_RES, bool NT, int ACT>
__global__ __launch_bounds__(EW_THREADS) void bn_act_apply_kernel(const u32x4* __restrict__ y, const float* __restrict__ coef,
                                                                  const u32x4* __restrict__ res, int res_cg,
                                                                  const float* __restrict__ keep, const float* __restrict__ count,
                                                                  u32x4* __restrict__ out, int out_ld8, long npix, int C, int act,
                                                                  float slope, const float* __restrict__ keep2,
                                                                  const float* __restrict__ count2) {
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * EW_THREADS + threadIdx.x;
    const long stride = (long)gridDim.x * EW_THREADS;   // multiple of cg
    const int cgi = (int)(gtid % cg);
    const int c0 = cgi * 8;
    const long pstep = stride / cg;
    float a[8], sh[8];
    load8f(coef + c0, a);
    load8f(coef + 3 * C + c0, sh);
    const float dsc = keep != nullptr ? drop_scale(count, npix) : 1.f;
    // a second DropBlock BEHIND the residual add (DarkNet's ResBlock: dropblock(x + conv(x)), darknetv3.py:59-61) rides along too
    const float dsc2 = keep2 != nullptr ? drop_scale(count2, npix) : 1.f;
    // the residual may cover only the first res_cg channel groups (ReXBlock: out[:, :Cin] += x, rexnet.py:141)
    const bool has_r = HAS_RES && cgi < res_cg;
    constexpr int BA_U = NT ? 4 : 1;                 // streaming tensors: four pixels per iteration; cache-resident ones: the one-pixel loop
    for (long p0 = gtid / cg; p0 < npix; p0 += BA_U * pstep) {
        u32x4 vy[BA_U], vr[BA_U];
        float kp[BA_U], kp2[BA_U];
#pragma unroll
        for (int u = 0; u < BA_U; ++u) {
            const long p = p0 + u * pstep;
            const bool ok = p < npix;
            const long pc = ok ? p : p0;
            vy[u] = ld16<NT>(y + pc * cg + cgi);
            if (has_r) vr[u] = ld16<NT>(res + pc * res_cg + cgi);
            kp[u] = keep != nullptr ? keep[pc] * dsc : 1.f;
            kp2[u] = keep2 != nullptr ? keep2[pc] * dsc2 : 1.f;
        }
#pragma unroll
        for (int u = 0; u < BA_U; ++u) {
            const long p = p0 + u * pstep;
            if (p >= npix) break;
            float fy[8], fr[8], o[8];
            unpack8(vy[u], fy);
            if (has_r) unpack8(vr[u], fr);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float z = act_fwd(a[i] * fy[i] + sh[i], ACT, slope);
                if (keep != nullptr) z *= kp[u];
                if (has_r) z += fr[i];
                if (keep2 != nullptr) z *= kp2[u];
                o[i] = z;
            }
            st16<NT>(out + p * out_ld8 + cgi, pack8(o));
        }
    }
}

template <int ACT>
__global__ __launch_bounds__(EW_THREADS) void bn_act_bwd_reduce_kernel(const u32x4* __restrict__ g, int g_ld8,
                                                                       const u32x4* __restrict__ y, const float* __restrict__ coef,
                                                                       const float* __restrict__ keep, const float* __restrict__ count,
                                                                       float* __restrict__ red, long npix, int C, int act,
                                                                       float slope, const int reps, const float* __restrict__ keep2,
                                                                       const float* __restrict__ count2) {
    extern __shared__ float sred[];
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * EW_THREADS + threadIdx.x;
    const long stride = (long)gridDim.x * EW_THREADS;
    const int cgi = (int)(gtid % cg);
    const int c0 = cgi * 8;
    const long pstep = stride / cg;
    float a[8], sh[8];
    load8f(coef + c0, a);
    load8f(coef + 3 * C + c0, sh);
    const float dsc = keep != nullptr ? drop_scale(count, npix) : 1.f;
    const float dsc2 = keep2 != nullptr ? drop_scale(count2, npix) : 1.f;
    float sv[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sv[0][i] = sv[1][i] = 0.f;
    // (one pixel per iteration: the four-pixel form of the apply kernels measured SLOWER here on every size, 33 -> 46 us at 96 MB)
    for (long p = gtid / cg; p < npix; p += pstep) {
        float fg[8], fy[8];
        unpack8(g[p * g_ld8 + cgi], fg);
        unpack8(y[p * cg + cgi], fy);
        const float kp = (keep != nullptr ? keep[p] * dsc : 1.f) * (keep2 != nullptr ? keep2[p] * dsc2 : 1.f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float dz = fg[i] * kp * act_bwd(a[i] * fy[i] + sh[i], ACT, slope);
            sv[0][i] += dz;
            sv[1][i] += dz * fy[i];
        }
    }
    // same slab layout as the RepBlock reduce ([4][C] per replica): sum dz at k=0, sum dz*y at k=1
    block_reduce_flush<2>(sv, cg, C, red + (size_t)(blockIdx.x % reps) * 4 * C, sred, 2);
}

template <bool NT, int ACT>
__global__ __launch_bounds__(EW_THREADS) void bn_act_bwd_apply_kernel(const u32x4* __restrict__ g, int g_ld8,
                                                                      const u32x4* __restrict__ y, const float* __restrict__ coef,
                                                                      const float* __restrict__ bc, const float* __restrict__ keep,
                                                                      const float* __restrict__ count, u32x4* __restrict__ dy,
                                                                      long npix, int C, int act, float slope,
                                                                      const float* __restrict__ keep2, const float* __restrict__ count2,
                                                                      u32x4* __restrict__ gres) {
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * EW_THREADS + threadIdx.x;
    const long stride = (long)gridDim.x * EW_THREADS;
    const int cgi = (int)(gtid % cg);
    const int c0 = cgi * 8;
    const long pstep = stride / cg;
    float a[8], sh[8], A[8], B[8], Cc[8];
    load8f(coef + c0, a);
    load8f(coef + 3 * C + c0, sh);
    load8f(bc + c0, A);
    load8f(bc + C + c0, B);
    load8f(bc + 2 * C + c0, Cc);
    const float dsc = keep != nullptr ? drop_scale(count, npix) : 1.f;
    const float dsc2 = keep2 != nullptr ? drop_scale(count2, npix) : 1.f;
    constexpr int BA_U = NT ? 4 : 1;
    for (long p0 = gtid / cg; p0 < npix; p0 += BA_U * pstep) {
        u32x4 vg[BA_U], vy[BA_U];
        float kp[BA_U], kp2[BA_U];
#pragma unroll
        for (int u = 0; u < BA_U; ++u) {
            const long p = p0 + u * pstep;
            const long pc = p < npix ? p : p0;
            vg[u] = ld16<NT>(g + pc * g_ld8 + cgi);
            vy[u] = ld16<NT>(y + pc * cg + cgi);
            kp[u] = keep != nullptr ? keep[pc] * dsc : 1.f;
            kp2[u] = keep2 != nullptr ? keep2[pc] * dsc2 : 1.f;
        }
#pragma unroll
        for (int u = 0; u < BA_U; ++u) {
            const long p = p0 + u * pstep;
            if (p >= npix) break;
            const long q = p * cg + cgi;
            float fg[8], fy[8], o[8];
            unpack8(vg[u], fg);
            unpack8(vy[u], fy);
            if (keep2 != nullptr) {      // gradient through the DropBlock behind the residual add: what the residual input receives
#pragma unroll
                for (int i = 0; i < 8; ++i) fg[i] *= kp2[u];
                if (gres != nullptr) st16<NT>(gres + q, pack8(fg));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float dz = fg[i] * kp[u] * act_bwd(a[i] * fy[i] + sh[i], ACT, slope);
                o[i] = A[i] * dz + B[i] * fy[i] + Cc[i];
            }
            st16<NT>(dy + q, pack8(o));
        }
    }
}

// ---------------------------------------------------------------- global average pool
// y[n][c] += mean over an HW slice: grid (slices, N); R = 256 / cg threads share a channel group and are combined in LDS,
// one atomicAdd per (block, channel) into the zeroed output (the squeeze-excite pools of ReXNet read 100+ MB each)
__global__ __launch_bounds__(256) void gap_fwd_kernel(const u32x4* __restrict__ x, float* __restrict__ y, int N, int HW, int C,
                                                      int rows_per_block) {
    __shared__ float sh[256 * 9];
    const int cg = C / 8;
    const int n = blockIdx.y;
    const int h0 = blockIdx.x * rows_per_block;
    int h1 = h0 + rows_per_block;
    if (h1 > HW) h1 = HW;
    const float inv = 1.f / (float)HW;
    const u32x4* p = x + (long)n * HW * cg;
    if (cg <= 256) {
        const int R = 256 / cg;                   // threads per channel group
        const int g = threadIdx.x % cg, r0 = threadIdx.x / cg;
        float s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = 0.f;
        if (r0 < R) {
            for (int h = h0 + r0; h < h1; h += R) {
                float f[8];
                unpack8(p[(long)h * cg + g], f);
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] += f[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) sh[threadIdx.x * 9 + i] = s[i];
        __syncthreads();
        for (int o = threadIdx.x; o < C; o += 256) {
            const int gg = o >> 3, e = o & 7;
            float acc = 0.f;
            for (int r = 0; r < R; ++r) acc += sh[(r * cg + gg) * 9 + e];
            atomicAdd(y + (long)n * C + o, acc * inv);
        }
    } else {
        for (int g = threadIdx.x; g < cg; g += 256) {
            float s[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] = 0.f;
            for (int h = h0; h < h1; ++h) {
                float f[8];
                unpack8(p[(long)h * cg + g], f);
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] += f[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) atomicAdd(y + (long)n * C + g * 8 + i, s[i] * inv);
        }
    }
}
__global__ void gap_bwd_kernel(const float* __restrict__ dy, u32x4* __restrict__ dx, int N, int HW, int C) {
    const int cg = C / 8;
    const long total = (long)N * HW * cg;
    const float inv = 1.f / (float)HW;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const int g = (int)(q % cg);
        const int n = (int)(q / ((long)HW * cg));
        float f[8];
        load8f(dy + (long)n * C + g * 8, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] *= inv;
        dx[q] = pack8(f);
    }
}

// ---------------------------------------------------------------- layout / packing
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int N, int C, int H, int W, int Cpad) {
    const long total = (long)N * H * W * Cpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const long p = i / Cpad;
        const int w = (int)(p % W);
        const long r = p / W;
        const int h = (int)(r % H);
        const int n = (int)(r / H);
        y[i] = c < C ? f32_to_bf16(x[(((long)n * C + c) * H + h) * W + w]) : (bf16_t)0;
    }
}
__global__ void nhwc_to_nchw_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W, int Cpad) {
    const long total = (long)N * C * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        long r = i / W;
        const int h = (int)(r % H);
        r /= H;
        const int c = (int)(r % C);
        const int n = (int)(r / C);
        y[i] = bf16_to_f32(x[(((long)n * H + h) * W + w) * Cpad + c]);
    }
}
// mode 0: wpk[co][tap0 + kh*KW+kw][ci]            (forward)
// mode 1: wpk[ci][tap0 + flipped(kh,kw)][co]      (data gradient)
// mode 2: wpk[co][0][(kh*KW+kw)*Cin + ci], K padded to T*? (im2col order; T = padded K)
// modes 3 / 4: the row-unit image of conv_rows.hip (forward / data gradient), C = Cout = Cin, a multiple of 48:
//   wpk[step = tap * C/32 + kc/32][row r(rc)][j(kc % 32)], rc = the conv's OUTPUT channel (mode 3: co, mode 4: ci), kc its input
//   channel, tap = tap0 + kh*KW+kw (mode 4: spatially flipped).  r() puts the 16 rows of an MFMA A fragment next to each other
//   (a lane group g of the D tile ends up with channels 8 g .. 8 g + 7 and 32 + 4 g .. 32 + 4 g + 3 of the wave's 48), j() puts a lane's eight k values (4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3) into one 16-byte piece.
__device__ __forceinline__ long rows_image_index(int rc, int kc, int tap, int Cc) {
    // channel c of a wave's 48 -> (fragment f, MFMA row 4 g + i): lane group g of the D tile then holds channels 8 g .. 8 g + 7
    // (fragments 0, 1) and 32 + 4 g .. 32 + 4 g + 3 (fragment 2): one 16-byte and one 8-byte piece per pixel, contiguous over g
    const int w = rc / 48, c = rc - 48 * w;
    const int f = c < 32 ? (c >> 2) & 1 : 2, gq = c < 32 ? c >> 3 : (c - 32) >> 2, i = c & 3;
    const int r = 48 * w + 16 * f + 4 * gq + i;
    const int t = kc >> 5, kk = kc & 31, hi = kk >> 4, g = (kk & 15) >> 2, e = kk & 3;
    const int j = 8 * g + 4 * hi + e;
    return ((long)(tap * ((Cc + 31) >> 5) + t) * Cc + r) * 32 + j;       // a tap's channels are padded to whole k32 steps (C = 48: two)
}
// modes 5 / 6: the fragment images of conv_s2.hip (stride-2 forward), [co / 16][step][64 lanes][8] bf16 - one coalesced 1 KB load per
// MFMA A fragment, lane = 16 g + (co % 16).
//   mode 5 (NHWC layers, Cin % 8 == 0): the K stream is 16-byte pieces, piece q = tap0 + tap * (Cin / 8) + ci / 8 (tap0 = 0 for the
//     3x3 kernel, 9 Cin / 8 for the 1x1); k32 step q / 4 - ld (ld = first step of THIS image), lane group q % 4, element ci % 8.
//   mode 6 (stem, Cin = 3): a kernel row kh is 16 k slots (slot 4 kw + ci), step kh / 2, lane group 2 (kh % 2) + slot / 8; the 1x1
//     passes tap0 = 4 (it sits at the centre tap) and has a one-step image.
// Entries that no source element maps to (the other conv's pieces in a shared step, padding slots) must be zero: allocate zeroed.
__device__ __forceinline__ long s2_image_index(int mode, int co, int ci, int t, int Cin, int tap0, int T, int ld, int Cout = 0, int KK = 9) {
    int s, g, e;
    if (mode == 7) {
        // data gradient (conv_s2.hip s2_dgrad_kernel): rows = ci, K = co pieces, taps in output-parity order:
        // (1,1) W1 | (1,0) (1,2) | (0,1) (2,1) | (0,0) (0,2) (2,0) (2,2)
        const int pos = KK == 1 ? 1 : (t == 4 ? 0 : t == 3 ? 2 : t == 5 ? 3 : t == 1 ? 4 : t == 7 ? 5 : t == 0 ? 6 : t == 2 ? 7 : t == 6 ? 8 : 9);
        const int q = pos * (Cout >> 3) + (co >> 3);
        return ((((long)(ci >> 4) * T + (q >> 2)) * 64) + (q & 3) * 16 + (ci & 15)) * 8 + (co & 7);
    }
    if (mode == 5) {
        const int q = tap0 + t * (Cin >> 3) + (ci >> 3);
        s = (q >> 2) - ld; g = q & 3; e = ci & 7;
    } else {
        const int ta = tap0 + t, kh = ta / 3, kw = ta - 3 * kh, slot = 4 * kw + ci;
        s = (kh >> 1) - ld; g = 2 * (kh & 1) + (slot >> 3); e = slot & 7;
    }
    return ((((long)(co >> 4) * T + s) * 64) + g * 16 + (co & 15)) * 8 + e;
}
__global__ void pack_weight_kernel(const float* __restrict__ w, bf16_t* __restrict__ wpk, int Cout, int Cin, int KH, int KW,
                                   int mode, int tap0, int T, int ld) {
    const long total = (long)Cout * Cin * KH * KW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int kw = (int)(i % KW);
        long r = i / KW;
        const int kh = (int)(r % KH);
        r /= KH;
        const int ci = (int)(r % Cin);
        const int co = (int)(r / Cin);
        const bf16_t v = f32_to_bf16(w[i]);
        if (mode == 0) {
            wpk[((long)co * T + tap0 + kh * KW + kw) * Cin + ci] = v;
        } else if (mode == 1) {
            const int t = (KH - 1 - kh) * KW + (KW - 1 - kw);
            wpk[((long)ci * T + tap0 + t) * Cout + co] = v;
        } else if (mode == 3) {
            wpk[rows_image_index(co, ci, tap0 + kh * KW + kw, Cout)] = v;
        } else if (mode == 4) {
            wpk[rows_image_index(ci, co, tap0 + (KH - 1 - kh) * KW + (KW - 1 - kw), Cout)] = v;
        } else if (mode >= 5) {
            wpk[s2_image_index(mode, co, ci, kh * KW + kw, Cin, tap0, T, ld, Cout, KH * KW)] = v;
        } else {
            wpk[(long)co * T + tap0 + (kh * KW + kw) * Cin + ci] = v;
        }
    }
}
// all conv weights of a model in one launch: blockIdx.y = item.  The fp32 OIHW source is read in runs along its fastest axes
// (ci, tap) and the bf16 destination is written in runs along ITS fastest axis (ci for the forward image, co for the
// data-gradient image) through an LDS tile: a thread-per-output-element walk reads the data-gradient image with a stride of
// Cin * KK floats (one cache line per lane), which made this kernel 20x slower than its traffic.
constexpr int PK_BATCH = 5;       // 16-byte source loads a thread keeps in flight
constexpr int PK_MAXKK = 9;
constexpr int PK_TILE = 16 * 64 * PK_MAXKK;      // floats: 16 x 64 (co x ci or ci x co) x taps
// mode 0: tiles of 16 co x 64 ci, written [co][tap][ci] (rows of ld >= Cin; the padding is left untouched)
// mode 1: tiles of 64 co x 16 ci, written [ci][flipped tap][co]
// KKC: compile-time tap count (1 | 9; 0 = run time) - the index arithmetic is all divisions, which must be by constants
template <int MODE, int KKC>
__device__ __forceinline__ void pack_tiles(const hc_pack_item& it, bf16_t* __restrict__ wpk, float* __restrict__ tile, const int tl) {
    const int KK = KKC ? KKC : it.KH * it.KW;
    constexpr int TCO = MODE == 0 ? 16 : 64, TCI = MODE == 0 ? 64 : 16;
    const int nco = (it.Cout + TCO - 1) / TCO, nci = (it.Cin + TCI - 1) / TCI;
    const int run = TCI * KK;                       // contiguous source floats per co row of a tile
    const int lrun = run + 1;                       // LDS row stride (odd: the transposing reads spread over the banks)
    {
        (void)nco;
        const int co0 = (tl / nci) * TCO, ci0 = (tl % nci) * TCI;
        const int cw = min(TCI, it.Cin - ci0) * KK;   // valid floats of a row
        __syncthreads();
        // whole tiles of 16-byte-aligned rows: 16-byte loads, ALL of a thread's (nine for a 3x3 tile) issued in batches of five before their LDS stores.
        // The dword form (six in flight per thread) ran the RepVGG-A0 pack at 1.5 TB/s: 296 MB in 198 us, 2 % of the step.
        const bool vin = KKC != 0 && cw == run && co0 + TCO <= it.Cout && ((it.Cin * KK) & 3) == 0 && ((reinterpret_cast<size_t>(it.w) & 15) == 0);
        if (vin) {
            constexpr int RUN4 = KKC ? TCI * KKC / 4 : 1, NV = (TCO * RUN4 + 255) / 256;
#pragma unroll 1
            for (int k0 = 0; k0 < NV; k0 += PK_BATCH) {           // batches: all nine at once cost 168 registers and a workgroup per CU
                f32x4 v[PK_BATCH];
#pragma unroll
                for (int k = 0; k < PK_BATCH; ++k) {
                    const int e = threadIdx.x + 256 * (k0 + k);
                    const int r = e / RUN4, c = 4 * (e - r * RUN4);
                    if (e < TCO * RUN4)
                        v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(it.w + ((long)(co0 + r) * it.Cin + ci0) * KK + c));
                }
#pragma unroll
                for (int k = 0; k < PK_BATCH; ++k) {
                    const int e = threadIdx.x + 256 * (k0 + k);
                    const int r = e / RUN4, c = 4 * (e - r * RUN4);
                    if (e < TCO * RUN4) {
                        float* d = tile + r * lrun + c;
                        d[0] = v[k][0]; d[1] = v[k][1]; d[2] = v[k][2]; d[3] = v[k][3];
                    }
                }
            }
        } else {
#pragma unroll 6
            for (int e = threadIdx.x; e < TCO * run; e += 256) {      // unrolled: six independent loads in flight per thread
                const int r = e / run, c = e - r * run;
                if (co0 + r < it.Cout && c < cw) tile[r * lrun + c] = it.w[((long)(co0 + r) * it.Cin + ci0) * KK + c];
            }
        }
        __syncthreads();
        // 16-byte stores: a thread gathers EIGHT consecutive destination elements from the tile (the one-element form issued a 2-byte
        // store per lane, 128 bytes per wave instruction, and the 1280 x 1280 x 3 x 3 images alone are 59 MB of them: 0.2 ms per step)
        const int ldd = it.ld > 0 ? it.ld : (MODE == 0 ? it.Cin : it.Cout);
        const bool wide = (ldd & 7) == 0 && ((reinterpret_cast<size_t>(wpk) & 15) == 0) &&
                          (MODE == 0 ? (ci0 + TCI <= it.Cin && co0 + TCO <= it.Cout) : (co0 + TCO <= it.Cout && ci0 + TCI <= it.Cin));
        if (wide) {
            constexpr int INNER = MODE == 0 ? TCI : TCO;            // destination-contiguous axis of the tile
            constexpr int OUTER = MODE == 0 ? TCO : TCI;
#pragma unroll 2
            for (int e = threadIdx.x; e < OUTER * KK * (INNER / 8); e += 256) {
                const int c8 = e % (INNER / 8), q = e / (INNER / 8), t = q % KK, o = q / KK;
                float f[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    f[k] = MODE == 0 ? tile[o * lrun + (8 * c8 + k) * KK + t] : tile[(8 * c8 + k) * lrun + o * KK + (KK - 1 - t)];
                const u32x4 pk = pack8(f);
                const long dst = MODE == 0 ? ((long)(co0 + o) * it.T + it.tap0 + t) * ldd + ci0 + 8 * c8
                                           : ((long)(ci0 + o) * it.T + it.tap0 + t) * ldd + co0 + 8 * c8;
                *reinterpret_cast<u32x4*>(wpk + dst) = pk;
            }
        } else if (MODE == 0) {
            const int ld = it.ld > 0 ? it.ld : it.Cin;
#pragma unroll 4
            for (int e = threadIdx.x; e < TCO * KK * TCI; e += 256) {
                const int ci = e % TCI, q = e / TCI, t = q % KK, r = q / KK;
                if (co0 + r < it.Cout && ci0 + ci < it.Cin)
                    wpk[((long)(co0 + r) * it.T + it.tap0 + t) * ld + ci0 + ci] = f32_to_bf16(tile[r * lrun + ci * KK + t]);
            }
        } else {
            const int ld = it.ld > 0 ? it.ld : it.Cout;
#pragma unroll 4
            for (int e = threadIdx.x; e < TCI * KK * TCO; e += 256) {
                const int r = e % TCO, q = e / TCO, tf = q % KK, ci = q / KK;
                if (co0 + r < it.Cout && ci0 + ci < it.Cin)
                    wpk[((long)(ci0 + ci) * it.T + it.tap0 + tf) * ld + co0 + r] = f32_to_bf16(tile[r * lrun + ci * KK + (KK - 1 - tf)]);
            }
        }
    }
}

// modes 3 / 4 (row-unit images of conv_rows.hip) through an LDS tile: one unit = 48 row channels x one k32 block x all taps.  The fp32
// source is read in 16-byte pieces along its contiguous runs (mode 3: a co row's 32 ci x taps; mode 4: a co row's 48 ci x taps), kept as
// bf16 in LDS (27 KB), and every tap's 48 x 32 block of the image leaves as 3 KB of contiguous 16-byte stores (the lane's eight k values
// of one MFMA fragment row).  The element walk wrote 2 bytes per lane at scattered addresses: the fourteen 192-channel blocks of
// RepVGG-A0 alone kept the pack at 1.3 TB/s.  Pieces past the channel count (C = 48: the second k block is half empty) are written as
// zeros, which is what the image holds there anyway.
template <int MODE, int KK, int nk>                                // nk: valid k channels of the unit's k32 block (32; 16 in the last block when C % 32 = 16)
__device__ __forceinline__ void pack_rows_unit(const hc_pack_item& it, bf16_t* __restrict__ wpk, float* __restrict__ tile_f, const int w48,
                                               const int kblk) {
    bf16_t* tile = reinterpret_cast<bf16_t*>(tile_f);
    const int C = it.Cout, KB = (C + 31) >> 5;
    constexpr int NR = MODE == 3 ? 48 : 32;                        // LDS rows: mode 3 one per row channel, mode 4 one per k channel
    constexpr int LS = (MODE == 3 ? 32 : 48) * KK + 2;             // LDS row stride in elements (an odd number of dwords)
    static_assert(NR * LS * 2 <= (PK_TILE + 64) * 4, "tile buffer too small");
    constexpr int rows = MODE == 3 ? 48 : nk;
    constexpr int run4 = (MODE == 3 ? nk : 48) * KK / 4;               // 16-byte pieces per source run
    __syncthreads();
    constexpr int NV = (rows * run4 + 255) / 256;
#pragma unroll 1
    for (int k0 = 0; k0 < NV; k0 += PK_BATCH) {
        f32x4 v[PK_BATCH];
#pragma unroll
        for (int k = 0; k < PK_BATCH; ++k) {
            const int e = threadIdx.x + 256 * (k0 + k);
            const int r = e / run4, c4 = e - r * run4;
            if (r < rows) {
                const long src = MODE == 3 ? ((long)(48 * w48 + r) * C + 32 * kblk) * KK : ((long)(32 * kblk + r) * C + 48 * w48) * KK;
                v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(it.w + src) + c4);
            }
        }
#pragma unroll
        for (int k = 0; k < PK_BATCH; ++k) {
            const int e = threadIdx.x + 256 * (k0 + k);
            const int r = e / run4, c4 = e - r * run4;
            if (r < rows) {
                bf16_t* d = tile + r * LS + 4 * c4;
                d[0] = f32_to_bf16(v[k][0]); d[1] = f32_to_bf16(v[k][1]); d[2] = f32_to_bf16(v[k][2]); d[3] = f32_to_bf16(v[k][3]);
            }
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < KK * 48 * 4; o += 256) {
        const int g = o & 3, q = o >> 2, rp = q % 48, t = q / 48;
        const int f = rp >> 4, gq = (rp & 15) >> 2, i = rp & 3;
        const int c = f < 2 ? 8 * gq + 4 * f + i : 32 + 4 * gq + i;            // row channel whose image row is 48 w + rp
        unsigned short h[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int kk = 16 * (x >> 2) + 4 * g + (x & 3);
            const bf16_t b = MODE == 3 ? tile[c * LS + kk * KK + t] : tile[kk * LS + c * KK + t];
            h[x] = kk < nk ? __builtin_bit_cast(unsigned short, b) : (unsigned short)0;
        }
        u32x4 pk;
#pragma unroll
        for (int x = 0; x < 4; ++x) pk[x] = (unsigned)h[2 * x] | ((unsigned)h[2 * x + 1] << 16);
        const int tap = it.tap0 + (MODE == 3 ? t : KK - 1 - t);
        *reinterpret_cast<u32x4*>(wpk + (((long)tap * KB + kblk) * C + 48 * w48 + rp) * 32 + 8 * g) = pk;
    }
}
// Work units of an item: its LDS tiles (modes 0 / 1, kernels up to 3x3) or chunks of PK_CHUNK source elements (element-wise modes).
// The grid is ONE dimension of a few thousand workgroups that walk the flat unit list (the former (tiles, items) grid launched
// 512 x 108 workgroups of which 50 000 had nothing to do; removing them did not change the 175 us of the RepVGG-A0 pack, whose time
// is in the two 59 MB passes over the 1280 x 1280 x 3 x 3 tensor).
constexpr int PK_CHUNK = 8192;
__device__ __forceinline__ bool pack_rows_tiled(const hc_pack_item& it) {
    const int KK = it.KH * it.KW;
    return (it.mode == 3 || it.mode == 4) && (KK == 9 || KK == 1) && it.Cout == it.Cin && it.Cout % 48 == 0 &&
           ((reinterpret_cast<size_t>(it.w) | reinterpret_cast<size_t>(it.dst)) & 15) == 0;
}
__device__ __forceinline__ int pack_units(const hc_pack_item& it) {
    const int KK = it.KH * it.KW;
    if (pack_rows_tiled(it)) return (it.Cout / 48) * ((it.Cout + 31) >> 5);
    if (it.mode >= 2 || KK > PK_MAXKK) return (int)(((long)it.Cout * it.Cin * KK + PK_CHUNK - 1) / PK_CHUNK);
    return it.mode == 0 ? ((it.Cout + 15) / 16) * ((it.Cin + 63) / 64) : ((it.Cout + 63) / 64) * ((it.Cin + 15) / 16);
}
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const hc_pack_item* __restrict__ items, const int nitems) {
    __shared__ float tile[PK_TILE + 64];
    extern __shared__ int pref[];                    // [nitems + 1] exclusive prefix of the unit counts
    for (int i = threadIdx.x; i < nitems; i += 256) pref[i + 1] = pack_units(items[i]);
    __syncthreads();
    if (threadIdx.x == 0) {
        pref[0] = 0;
        for (int i = 0; i < nitems; ++i) pref[i + 1] += pref[i];
    }
    __syncthreads();
    const int total_units = pref[nitems];
    for (int flat = blockIdx.x; flat < total_units; flat += gridDim.x) {
        int lo = 0, hi = nitems;                     // largest item index with pref[item] <= flat
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pref[mid] <= flat) lo = mid; else hi = mid;
        }
        const hc_pack_item it = items[lo];
        const int tl = flat - pref[lo];
        const int KK = it.KH * it.KW;
        bf16_t* wpk = reinterpret_cast<bf16_t*>(it.dst);
        const long total = (long)it.Cout * it.Cin * KK;
        const long o0 = (long)tl * PK_CHUNK, o1 = o0 + PK_CHUNK < total ? o0 + PK_CHUNK : total;
        if (pack_rows_tiled(it)) {
            const int KB = (it.Cout + 31) >> 5, w48 = tl / KB, kblk = tl - w48 * KB;
            const bool full = it.Cout - 32 * kblk >= 32;
            if (it.mode == 3) {
                if (KK == 9) { if (full) pack_rows_unit<3, 9, 32>(it, wpk, tile, w48, kblk); else pack_rows_unit<3, 9, 16>(it, wpk, tile, w48, kblk); }
                else { if (full) pack_rows_unit<3, 1, 32>(it, wpk, tile, w48, kblk); else pack_rows_unit<3, 1, 16>(it, wpk, tile, w48, kblk); }
            } else {
                if (KK == 9) { if (full) pack_rows_unit<4, 9, 32>(it, wpk, tile, w48, kblk); else pack_rows_unit<4, 9, 16>(it, wpk, tile, w48, kblk); }
                else { if (full) pack_rows_unit<4, 1, 32>(it, wpk, tile, w48, kblk); else pack_rows_unit<4, 1, 16>(it, wpk, tile, w48, kblk); }
            }
        } else if (it.mode >= 3) {                   // row-unit images: walk the source (16-byte runs on both sides)
            for (long o = o0 + threadIdx.x; o < o1; o += 256) {
                const int t = (int)(o % KK);
                const long r = o / KK;
                const int ci = (int)(r % it.Cin), co = (int)(r / it.Cin);
                const bf16_t v = f32_to_bf16(it.w[o]);
                if (it.mode == 3) wpk[rows_image_index(co, ci, it.tap0 + t, it.Cout)] = v;
                else if (it.mode == 4) wpk[rows_image_index(ci, co, it.tap0 + KK - 1 - t, it.Cout)] = v;
                else wpk[s2_image_index(it.mode, co, ci, t, it.Cin, it.tap0, it.T, it.ld, it.Cout, KK)] = v;
            }
        } else if (it.mode == 2 || KK > PK_MAXKK) {  // im2col order / large kernels: element-wise walk
            for (long o = o0 + threadIdx.x; o < o1; o += 256) {
                int co, ci, t;
                long dst;
                if (it.mode == 0) {
                    ci = (int)(o % it.Cin);
                    const long r = o / it.Cin;
                    t = (int)(r % KK);
                    co = (int)(r / KK);
                    dst = ((long)co * it.T + it.tap0 + t) * (it.ld > 0 ? it.ld : it.Cin) + ci;
                } else if (it.mode == 1) {
                    co = (int)(o % it.Cout);
                    const long r = o / it.Cout;
                    const int tf = (int)(r % KK);
                    ci = (int)(r / KK);
                    t = KK - 1 - tf;
                    dst = ((long)ci * it.T + it.tap0 + tf) * (it.ld > 0 ? it.ld : it.Cout) + co;
                } else {                     // [co][tap0 + tap*Cin + ci]
                    ci = (int)(o % it.Cin);
                    const long r = o / it.Cin;
                    t = (int)(r % KK);
                    co = (int)(r / KK);
                    dst = (long)co * it.T + it.tap0 + t * it.Cin + ci;
                }
                wpk[dst] = f32_to_bf16(it.w[((long)co * it.Cin + ci) * KK + t]);
            }
        } else if (it.mode == 0) {
            if (KK == 9) pack_tiles<0, 9>(it, wpk, tile, tl);
            else if (KK == 1) pack_tiles<0, 1>(it, wpk, tile, tl);
            else pack_tiles<0, 0>(it, wpk, tile, tl);
        } else {
            if (KK == 9) pack_tiles<1, 9>(it, wpk, tile, tl);
            else if (KK == 1) pack_tiles<1, 1>(it, wpk, tile, tl);
            else pack_tiles<1, 0>(it, wpk, tile, tl);
        }
    }
}

// im2col for tiny Cin (stem): x NCHW fp32 -> col [N][OH][OW][Kpad] bf16, k = (kh*KW+kw)*Cin+ci
// CINC / KSC: compile-time channel count and (square) kernel size, 0 = run time.
// Grid: y walks the output rows (n, oy), x the (ox, 8-wide k chunk) items of one row - the former flat index cost three 64-bit
// div / mod pairs per item (a few hundred VALU instructions for eight 4-byte gathers: the YOLOv4 stem ran at 2 TB/s of its own bytes);
// here a row costs one 32-bit division per thread and an item one more, offsets inside an image are 32-bit.
template <int CINC, int KSC, typename Store>
__device__ __forceinline__ void im2col_rows(const float* __restrict__ x, int N, int Cin_, int H, int W, int OH, int OW, int KH_, int KW_,
                                            int stride, int pad, int Kpad, Store store) {
    const int Cin = CINC ? CINC : Cin_, KH = KSC ? KSC : KH_, KW = KSC ? KSC : KW_;
    const unsigned kchunks = (unsigned)Kpad / 8u, rowq = (unsigned)OW * kchunks;
    const int K = Cin * KH * KW;
    const unsigned rows = (unsigned)N * (unsigned)OH;
    for (unsigned row = blockIdx.y; row < rows; row += gridDim.y) {
        const unsigned n = row / (unsigned)OH;
        const int oy = (int)(row - n * (unsigned)OH);
        const float* __restrict__ xn = x + (size_t)n * Cin * H * W;
        for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < rowq; t += gridDim.x * blockDim.x) {
            const unsigned oxu = t / kchunks;
            const int kc = (int)(t - oxu * kchunks), ox = (int)oxu;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kc * 8 + e;
                float v = 0.f;
                if (k < K) {
                    const int ci = k % Cin, tp = k / Cin;
                    const int kh = tp / KW, kw = tp % KW;
                    const int iy = oy * stride + kh - pad, ix = ox * stride + kw - pad;
                    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = xn[(ci * H + iy) * W + ix];
                }
                f[e] = v;
            }
            store((size_t)row * rowq + t, f);
        }
    }
}

template <int CINC, int KSC>
__global__ __launch_bounds__(256) void im2col_small_kernel(const float* __restrict__ x, bf16_t* __restrict__ col, int N, int Cin, int H, int W,
                                                           int OH, int OW, int KH, int KW, int stride, int pad, int Kpad) {
    im2col_rows<CINC, KSC>(x, N, Cin, H, W, OH, OW, KH, KW, stride, pad, Kpad,
                           [col](size_t q, const float (&f)[8]) { reinterpret_cast<u32x4*>(col)[q] = pack8(f); });
}
// same gather, quantised straight to OCP e4m3 bytes (fp8 inference stem): col [N][OH][OW][Kpad] uint8 = fp8(x * inv_scale)
template <int CINC, int KSC>
__global__ __launch_bounds__(256) void im2col_small_fp8_kernel(const float* __restrict__ x, unsigned char* __restrict__ col, int N, int Cin,
                                                               int H, int W, int OH, int OW, int KH, int KW, int stride, int pad, int Kpad,
                                                               float inv_scale) {
    im2col_rows<CINC, KSC>(x, N, Cin, H, W, OH, OW, KH, KW, stride, pad, Kpad, [col, inv_scale](size_t q, const float (&f)[8]) {
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = fminf(fmaxf(f[e] * inv_scale, -448.f), 448.f);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(g[0], g[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(g[2], g[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(g[4], g[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(g[6], g[7], hi, true);
        u32x2 pk;
        pk[0] = (unsigned)lo;
        pk[1] = (unsigned)hi;
        reinterpret_cast<u32x2*>(col)[q] = pk;
    });
}
// The 3-channel 3 x 3 pad-1 stem (every model's first layer) with the input staged through LDS: a workgroup owns an output row, loads the
// 3 channels x 3 input rows it needs with coalesced 16-byte loads (zero halo columns / rows written in place), and the eight 4-byte
// gathers of an item become LDS reads - the global-memory side of the kernel is then two coalesced streams.  Same values, same
// layout as im2col_rows (bit-identical column tensor).  LDS: 9 rows of W + 2 floats.
// The assembly of an item is VALU-bound if the (ci, kh, kw) of each of its eight k are derived arithmetically (~200 instructions per
// item, four clocks each per wave: 3-5 CU clocks per item measured, whatever the grid): a table of LDS byte offsets per k, built once
// per workgroup, leaves one add and two LDS reads per element; the padding k >= 27 point into a tenth, all-zero tile row, so there
// is no select either.  (A persistent grid with the next row's loads prefetched into registers measured the same as one row per
// workgroup - the latency chain was never the limit - and is not kept.)
// LDS: 10 rows of W + 2 floats, then Kpad ints.
template <typename Store>
__device__ __forceinline__ void im2col3_lds_rows(const float* __restrict__ x, int N, int H, int W, int OH, int OW, int stride, int Kpad,
                                                 float* __restrict__ tile, Store store) {
    const unsigned kchunks = (unsigned)Kpad / 8u, rowq = (unsigned)OW * kchunks;
    const unsigned rows = (unsigned)N * (unsigned)OH;
    const int LW = W + 2, w4 = W >> 2;
    int* __restrict__ offs = reinterpret_cast<int*>(tile + 10 * LW);
    for (int k = threadIdx.x; k < Kpad; k += blockDim.x) {
        const int ci = k % 3, tp = k / 3;
        const int kh = tp / 3, kw = tp - kh * 3;
        offs[k] = (k < 27 ? (ci * 3 + kh) * LW + kw : 9 * LW) * 4;
    }
    for (int i = threadIdx.x; i < LW; i += blockDim.x) tile[9 * LW + i] = 0.f;
    const bool p2 = (kchunks & (kchunks - 1u)) == 0u;
    const int sh = __ffs((int)kchunks) - 1;
    for (unsigned row = blockIdx.x; row < rows; row += gridDim.x) {
        const unsigned n = row / (unsigned)OH;
        const int oy = (int)(row - n * (unsigned)OH);
        const float* __restrict__ xn = x + (size_t)n * 3 * H * W;
        __syncthreads();                         // the previous row's readers are done with the tile
        if ((W & 3) == 0) {
            for (int i = threadIdx.x; i < 9 * w4; i += blockDim.x) {
                const int r = i / w4, c4 = i - r * w4;          // r = ci * 3 + kh
                const int ci = r / 3, kh = r - ci * 3;
                const int iy = oy * stride + kh - 1;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)iy < (unsigned)H) v = *reinterpret_cast<const f32x4*>(xn + (ci * H + iy) * W + c4 * 4);
                float* d = tile + r * LW + 1 + c4 * 4;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            }
        } else {
            for (int i = threadIdx.x; i < 9 * W; i += blockDim.x) {
                const int r = i / W, c = i - r * W;
                const int ci = r / 3, kh = r - ci * 3;
                const int iy = oy * stride + kh - 1;
                tile[r * LW + 1 + c] = (unsigned)iy < (unsigned)H ? xn[(ci * H + iy) * W + c] : 0.f;
            }
        }
        if (threadIdx.x < 18) tile[(threadIdx.x >> 1) * LW + ((threadIdx.x & 1) ? W + 1 : 0)] = 0.f;     // halo columns
        __syncthreads();
        const char* tb = reinterpret_cast<const char*>(tile);
        for (unsigned t = threadIdx.x; t < rowq; t += blockDim.x) {
            const unsigned oxu = p2 ? (t >> sh) : (t / kchunks);
            const int kc = (int)(t - oxu * kchunks);
            const int bbyte = (int)oxu * stride * 4;             // column ix = ox * stride + kw - 1 sits at index ix + 1
            const int* ok = offs + kc * 8;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = *reinterpret_cast<const float*>(tb + bbyte + ok[e]);
            store((size_t)row * rowq + t, f);
        }
    }
}

__global__ __launch_bounds__(256) void im2col3_lds_kernel(const float* __restrict__ x, bf16_t* __restrict__ col, int N, int H, int W, int OH,
                                                          int OW, int stride, int Kpad) {
    extern __shared__ float im2col_tile[];
    im2col3_lds_rows(x, N, H, W, OH, OW, stride, Kpad, im2col_tile,
                     [col](size_t q, const float (&f)[8]) { reinterpret_cast<u32x4*>(col)[q] = pack8(f); });
}
__global__ __launch_bounds__(256) void im2col3_lds_fp8_kernel(const float* __restrict__ x, unsigned char* __restrict__ col, int N, int H, int W,
                                                              int OH, int OW, int stride, int Kpad, float inv_scale) {
    extern __shared__ float im2col_tile[];
    im2col3_lds_rows(x, N, H, W, OH, OW, stride, Kpad, im2col_tile, [col, inv_scale](size_t q, const float (&f)[8]) {
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = fminf(fmaxf(f[e] * inv_scale, -448.f), 448.f);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(g[0], g[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(g[2], g[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(g[4], g[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(g[6], g[7], hi, true);
        u32x2 pk;
        pk[0] = (unsigned)lo;
        pk[1] = (unsigned)hi;
        reinterpret_cast<u32x2*>(col)[q] = pk;
    });
}
static inline size_t im2col3_lds_bytes(int W, int Kpad) { return ((size_t)10 * (W + 2) + Kpad) * sizeof(float); }
// the LDS form applies to the 3-channel 3 x 3 pad-1 stem whose staged rows fit 64 KB; HC_IM2COL_LDS=0 keeps the gather form (A/B)
static inline bool im2col3_lds_ok(int Cin, int KH, int KW, int pad, int W, int OW, int stride, int Kpad) {
    static const int on = [] { const char* e = getenv("HC_IM2COL_LDS"); return e == nullptr ? 1 : atoi(e); }();
    return on && Cin == 3 && KH == 3 && KW == 3 && pad == 1 && stride >= 1 && Kpad <= 256 && im2col3_lds_bytes(W, Kpad) <= 64 * 1024
           && (long)(OW - 1) * stride + 1 <= W;
}
// grid of the row walk: x covers one output row's items, y the rows (grid-stride beyond 65 535)
static inline dim3 im2col_grid(int N, int OH, int OW, int Kpad) {
    const long rowq = (long)OW * (Kpad / 8), rows = (long)N * OH;
    return dim3((unsigned)((rowq + 255) / 256), (unsigned)(rows > 65535 ? 65535 : (rows < 1 ? 1 : rows)));
}
// dwcol fp32 [Cout][Kpad] (k = (kh*KW+kw)*Cin+ci) -> dw OIHW
__global__ void unpack_im2col_grad_kernel(const float* __restrict__ dwcol, float* __restrict__ dw, int Cout, int Cin, int KH, int KW,
                                          int Kpad, int beta) {
    const int total = Cout * Cin * KH * KW;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int kw = i % KW;
    int r = i / KW;
    const int kh = r % KH;
    r /= KH;
    const int ci = r % Cin;
    const int co = r / Cin;
    const float v = dwcol[(long)co * Kpad + (kh * KW + kw) * Cin + ci];
    dw[i] = beta ? dw[i] + v : v;
}

inline int grid_for(long total, int threads = 256, int cap = 4096) {
    long b = (total + threads - 1) / threads;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

// ---- run-to-run determinism switch (include/holocron_hip.h) ----
static int g_hc_deterministic = 0;
static int g_hc_stat_replicas = HC_STAT_REPLICAS;

extern "C" {
int hc_get_stat_replicas(void) { return g_hc_stat_replicas; }
int hc_get_deterministic(void) { return g_hc_deterministic; }
int hc_set_deterministic(int on) {
    g_hc_deterministic = on ? 1 : 0;
    g_hc_stat_replicas = on ? HC_STAT_REPLICAS_DETERMINISTIC : HC_STAT_REPLICAS;
    return HC_OK;
}

int hc_rep_bn_finalize(const hc_rep_bn_desc* d, hc_stream_t stream) {
    if (d == nullptr || d->coef == nullptr || d->C <= 0) return HC_ERR_ARG;
    hipLaunchKernelGGL(rep_bn_finalize_kernel, dim3((d->C + FIN_CH - 1) / FIN_CH), dim3(256), 0, (hipStream_t)stream, *d, hc_get_stat_replicas());
    return hc_launch_status();
}

int hc_rep_apply(const void* y3, const void* y1, const void* x, const float* coef, void* out, float* out_stats, int64_t npix,
                 int32_t C, int32_t act, hc_stream_t stream) {
    if (y3 == nullptr || y1 == nullptr || coef == nullptr || out == nullptr || (C % 8) != 0) return HC_ERR_ARG;
    const long nchunks = (long)npix * (C / 8);
    const int blocks = ew_blocks(nchunks, C / 8, out_stats ? 16 : 8);
    hipStream_t st = (hipStream_t)stream;
    const size_t sm = out_stats ? EW_THREADS * 17 * sizeof(float) : 0;
#define HC_LAUNCH_APPLY1(ID, ST, NTM)                                                                                    \
    hipLaunchKernelGGL((rep_apply_kernel<ID, ST, NTM>), dim3(blocks), dim3(EW_THREADS), sm, st, (const u32x4*)y3,        \
                       (const u32x4*)y1, (const u32x4*)x, coef, (u32x4*)out, out_stats, nchunks, C, act, hc_get_stat_replicas())
#define HC_LAUNCH_APPLY(ID, ST)                                   \
    do {                                                          \
        if (ew_streaming(nchunks)) HC_LAUNCH_APPLY1(ID, ST, true); \
        else HC_LAUNCH_APPLY1(ID, ST, false);                     \
    } while (0)
    if (x != nullptr) {
        if (out_stats) HC_LAUNCH_APPLY(true, true); else HC_LAUNCH_APPLY(true, false);
    } else {
        if (out_stats) HC_LAUNCH_APPLY(false, true); else HC_LAUNCH_APPLY(false, false);
    }
#undef HC_LAUNCH_APPLY
#undef HC_LAUNCH_APPLY1
    return hc_launch_status();
}

int hc_channel_stats(const void* x, float* stats, int64_t npix, int32_t C, hc_stream_t stream) {
    if (x == nullptr || stats == nullptr || (C % 8) != 0) return HC_ERR_ARG;
    const long nchunks = (long)npix * (C / 8);
    const int blocks = ew_blocks(nchunks, C / 8, 16);
    hipLaunchKernelGGL(channel_stats_kernel, dim3(blocks), dim3(EW_THREADS), EW_THREADS * 17 * sizeof(float), (hipStream_t)stream,
                       (const u32x4*)x, stats, nchunks, C, hc_get_stat_replicas());
    return hc_launch_status();
}

static int rep_bwd_reduce_launch(const void* g, const void* out, const float* coef, int act, const void* y3, const void* y1,
                                 const void* x, float* red, int64_t npix, int32_t C, hc_stream_t stream) {
    if (g == nullptr || (out == nullptr && coef == nullptr) || y3 == nullptr || y1 == nullptr || red == nullptr || (C % 8) != 0)
        return HC_ERR_ARG;
    const long nchunks = (long)npix * (C / 8);
    const int blocks = ew_blocks(nchunks, C / 8, 16);
    hipStream_t st = (hipStream_t)stream;
    const size_t sm = EW_THREADS * 33 * sizeof(float);
#define HC_RBR1(ID, ZM, NTM)                                                                                                             \
    hipLaunchKernelGGL((rep_bwd_reduce_kernel<ID, ZM, NTM>), dim3(blocks), dim3(EW_THREADS), sm, st, (const u32x4*)g,                   \
                       (const u32x4*)out, (const u32x4*)y3, (const u32x4*)y1, (const u32x4*)x, coef, act, red, nchunks, C,             \
                       hc_get_stat_replicas())
#define HC_RBR(ID, ZM)                \
    do {                              \
        if (ew_streaming(nchunks)) HC_RBR1(ID, ZM, true); \
        else HC_RBR1(ID, ZM, false);  \
    } while (0)
    if (coef != nullptr) {
        if (x != nullptr) HC_RBR(true, true);
        else HC_RBR(false, true);
    } else {
        if (x != nullptr) HC_RBR(true, false);
        else HC_RBR(false, false);
    }
#undef HC_RBR
#undef HC_RBR1
    return hc_launch_status();
}
int hc_rep_bwd_reduce(const void* g, const void* out, const void* y3, const void* y1, const void* x, float* red, int64_t npix,
                      int32_t C, hc_stream_t stream) {
    if (out == nullptr) return HC_ERR_ARG;
    return rep_bwd_reduce_launch(g, out, nullptr, 1, y3, y1, x, red, npix, C, stream);
}
int hc_rep_bwd_reduce_z(const void* g, const float* coef, int32_t act, const void* y3, const void* y1, const void* x, float* red,
                        int64_t npix, int32_t C, hc_stream_t stream) {
    if (coef == nullptr || (act != 0 && act != 1)) return HC_ERR_ARG;
    return rep_bwd_reduce_launch(g, nullptr, coef, act, y3, y1, x, red, npix, C, stream);
}

int hc_rep_bn_bwd_finalize(const hc_rep_bn_bwd_desc* d, hc_stream_t stream) {
    if (d == nullptr || d->red == nullptr || d->save == nullptr || d->bcoef == nullptr) return HC_ERR_ARG;
    hipLaunchKernelGGL(rep_bn_bwd_finalize_kernel, dim3((d->C + FIN_CH - 1) / FIN_CH), dim3(256), 0, (hipStream_t)stream, *d, hc_get_stat_replicas());
    return hc_launch_status();
}

static int rep_bwd_apply_launch(const void* g, const void* out, const float* coef, int act, const void* y3, const void* y1,
                                const void* x, const float* bcoef, void* dy3, void* dy1, void* dxid, int64_t npix, int32_t C,
                                hc_stream_t stream) {
    if (g == nullptr || (out == nullptr && coef == nullptr) || y3 == nullptr || y1 == nullptr || bcoef == nullptr || dy3 == nullptr ||
        dy1 == nullptr || (C % 8) != 0)
        return HC_ERR_ARG;
    if ((x == nullptr) != (dxid == nullptr)) return HC_ERR_ARG;
    const long nchunks = (long)npix * (C / 8);
    const int blocks = ew_blocks(nchunks, C / 8);
    hipStream_t st = (hipStream_t)stream;
#define HC_RBA1(ID, ZM, NTM)                                                                                                         \
    hipLaunchKernelGGL((rep_bwd_apply_kernel<ID, ZM, NTM>), dim3(blocks), dim3(EW_THREADS), 0, st, (const u32x4*)g, (const u32x4*)out, \
                       (const u32x4*)y3, (const u32x4*)y1, (const u32x4*)x, coef, act, bcoef, (u32x4*)dy3, (u32x4*)dy1,             \
                       (u32x4*)dxid, nchunks, C)
#define HC_RBA(ID, ZM)                \
    do {                              \
        if (ew_streaming(nchunks)) HC_RBA1(ID, ZM, true); \
        else HC_RBA1(ID, ZM, false);  \
    } while (0)
    if (coef != nullptr) {
        if (x != nullptr) HC_RBA(true, true);
        else HC_RBA(false, true);
    } else {
        if (x != nullptr) HC_RBA(true, false);
        else HC_RBA(false, false);
    }
#undef HC_RBA
#undef HC_RBA1
    return hc_launch_status();
}
int hc_rep_bwd_apply(const void* g, const void* out, const void* y3, const void* y1, const void* x, const float* bcoef, void* dy3,
                     void* dy1, void* dxid, int64_t npix, int32_t C, hc_stream_t stream) {
    if (out == nullptr) return HC_ERR_ARG;
    return rep_bwd_apply_launch(g, out, nullptr, 1, y3, y1, x, bcoef, dy3, dy1, dxid, npix, C, stream);
}
int hc_rep_bwd_apply_z(const void* g, const float* coef, int32_t act, const void* y3, const void* y1, const void* x,
                       const float* bcoef, void* dy3, void* dy1, void* dxid, int64_t npix, int32_t C, hc_stream_t stream) {
    if (coef == nullptr || (act != 0 && act != 1)) return HC_ERR_ARG;
    return rep_bwd_apply_launch(g, nullptr, coef, act, y3, y1, x, bcoef, dy3, dy1, dxid, npix, C, stream);
}

int hc_bn_act_apply(const void* y, const float* coef, const void* res, int32_t res_C, const float* keep, const float* count,
                    void* out, int32_t out_ld, int64_t npix, int32_t C, int32_t act, float slope, hc_stream_t stream) {
    return hc_bn_act_apply_post(y, coef, res, res_C, keep, count, nullptr, nullptr, out, out_ld, npix, C, act, slope, stream);
}
int hc_bn_act_apply_post(const void* y, const float* coef, const void* res, int32_t res_C, const float* keep, const float* count,
                         const float* keep2, const float* count2, void* out, int32_t out_ld, int64_t npix, int32_t C, int32_t act,
                         float slope, hc_stream_t stream) {
    if (y == nullptr || coef == nullptr || out == nullptr || (C % 8) != 0 || (out_ld % 8) != 0 || out_ld < C) return HC_ERR_ARG;
    if (res != nullptr && (res_C <= 0 || res_C > C || (res_C % 8) != 0)) return HC_ERR_ARG;
    if ((keep == nullptr) != (count == nullptr) || (keep2 == nullptr) != (count2 == nullptr)) return HC_ERR_ARG;
    const long nchunks = (long)npix * (C / 8);
    const int blocks = ew_blocks(nchunks, C / 8);
    hipStream_t st = (hipStream_t)stream;
    if (act < 0 || act > 6) return HC_ERR_ARG;
#define HC_BAA(RS, NTM, A)                                                                                                       \
    hipLaunchKernelGGL((bn_act_apply_kernel<RS, NTM, A>), dim3(blocks), dim3(EW_THREADS), 0, st, (const u32x4*)y, coef, (const u32x4*)res, \
                       RS ? res_C / 8 : 0, keep, count, (u32x4*)out, out_ld / 8, (long)npix, C, act, slope, keep2, count2)
#define HC_BAA_ACT(RS, NTM)                                                                                                      \
    switch (act) {                                                                                                               \
        case 0: HC_BAA(RS, NTM, 0); break; case 1: HC_BAA(RS, NTM, 1); break; case 2: HC_BAA(RS, NTM, 2); break;                  \
        case 3: HC_BAA(RS, NTM, 3); break; case 4: HC_BAA(RS, NTM, 4); break; case 5: HC_BAA(RS, NTM, 5); break;                  \
        default: HC_BAA(RS, NTM, 6); break;                                                                                       \
    }
    const bool nt = ba_streaming(nchunks);
    if (res != nullptr) { if (nt) HC_BAA_ACT(true, true) else HC_BAA_ACT(true, false) }
    else { if (nt) HC_BAA_ACT(false, true) else HC_BAA_ACT(false, false) }
#undef HC_BAA_ACT
#undef HC_BAA
    return hc_launch_status();
}
int hc_bn_act_bwd_reduce(const void* g, int32_t g_ld, const void* y, const float* coef, const float* keep, const float* count,
                         float* red, int64_t npix, int32_t C, int32_t act, float slope, hc_stream_t stream) {
    return hc_bn_act_bwd_reduce_post(g, g_ld, y, coef, keep, count, nullptr, nullptr, red, npix, C, act, slope, stream);
}
int hc_bn_act_bwd_reduce_post(const void* g, int32_t g_ld, const void* y, const float* coef, const float* keep, const float* count,
                              const float* keep2, const float* count2, float* red, int64_t npix, int32_t C, int32_t act, float slope,
                              hc_stream_t stream) {
    if (g == nullptr || y == nullptr || coef == nullptr || red == nullptr || (C % 8) != 0 || (g_ld % 8) != 0 || g_ld < C) return HC_ERR_ARG;
    if ((keep == nullptr) != (count == nullptr) || (keep2 == nullptr) != (count2 == nullptr)) return HC_ERR_ARG;
    const long nchunks = (long)npix * (C / 8);
    const int blocks = ew_blocks(nchunks, C / 8, 16);
    if (act < 0 || act > 6) return HC_ERR_ARG;
#define HC_BAR(A)                                                                                                                  \
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<A>, dim3(blocks), dim3(EW_THREADS), EW_THREADS * 17 * sizeof(float), (hipStream_t)stream, \
                       (const u32x4*)g, g_ld / 8, (const u32x4*)y, coef, keep, count, red, (long)npix, C, act, slope, hc_get_stat_replicas(), \
                       keep2, count2)
    switch (act) {
        case 0: HC_BAR(0); break; case 1: HC_BAR(1); break; case 2: HC_BAR(2); break; case 3: HC_BAR(3); break;
        case 4: HC_BAR(4); break; case 5: HC_BAR(5); break; default: HC_BAR(6); break;
    }
#undef HC_BAR
    return hc_launch_status();
}
int hc_bn_act_bwd_apply(const void* g, int32_t g_ld, const void* y, const float* coef, const float* bcoef, const float* keep,
                        const float* count, void* dy, int64_t npix, int32_t C, int32_t act, float slope, hc_stream_t stream) {
    return hc_bn_act_bwd_apply_post(g, g_ld, y, coef, bcoef, keep, count, nullptr, nullptr, nullptr, dy, npix, C, act, slope, stream);
}
int hc_bn_act_bwd_apply_post(const void* g, int32_t g_ld, const void* y, const float* coef, const float* bcoef, const float* keep,
                             const float* count, const float* keep2, const float* count2, void* gres, void* dy, int64_t npix,
                             int32_t C, int32_t act, float slope, hc_stream_t stream) {
    if (g == nullptr || y == nullptr || coef == nullptr || bcoef == nullptr || dy == nullptr || (C % 8) != 0 || (g_ld % 8) != 0 ||
        g_ld < C)
        return HC_ERR_ARG;
    if ((keep == nullptr) != (count == nullptr) || (keep2 == nullptr) != (count2 == nullptr)) return HC_ERR_ARG;
    if (gres != nullptr && keep2 == nullptr) return HC_ERR_ARG;
    const long nchunks = (long)npix * (C / 8);
    const int blocks = ew_blocks(nchunks, C / 8);
    if (act < 0 || act > 6) return HC_ERR_ARG;
#define HC_BAB(NTM, A)                                                                                                       \
    hipLaunchKernelGGL((bn_act_bwd_apply_kernel<NTM, A>), dim3(blocks), dim3(EW_THREADS), 0, (hipStream_t)stream, (const u32x4*)g, g_ld / 8, \
                       (const u32x4*)y, coef, bcoef, keep, count, (u32x4*)dy, (long)npix, C, act, slope, keep2, count2, (u32x4*)gres)
#define HC_BAB_ACT(NTM)                                                                                                      \
    switch (act) {                                                                                                           \
        case 0: HC_BAB(NTM, 0); break; case 1: HC_BAB(NTM, 1); break; case 2: HC_BAB(NTM, 2); break; case 3: HC_BAB(NTM, 3); break; \
        case 4: HC_BAB(NTM, 4); break; case 5: HC_BAB(NTM, 5); break; default: HC_BAB(NTM, 6); break;                         \
    }
    if (ba_streaming(nchunks)) HC_BAB_ACT(true) else HC_BAB_ACT(false)
#undef HC_BAB_ACT
#undef HC_BAB
    return hc_launch_status();
}
int hc_gap_fwd(const void* x, float* y, int32_t N, int32_t HW, int32_t C, hc_stream_t stream) {
    if (x == nullptr || y == nullptr || (C % 8) != 0) return HC_ERR_ARG;
    if (N <= 0 || HW <= 0 || N > 65535) return (N == 0 || HW == 0) ? HC_OK : HC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hc_zero_async(y, sizeof(float) * (size_t)N * C, st) != hipSuccess) return HC_ERR_LAUNCH;
    const int cg = C / 8;
    const int R = cg <= 256 ? 256 / cg : 1;
    // enough workgroups to fill the chip, at least 4 rows per thread
    int rows = R * 4;
    while ((long)N * ((HW + rows - 1) / rows) > 8192 && rows < HW) rows *= 2;
    if (hc_get_deterministic()) rows = HW;      // one workgroup per image: a single add per output, no order to depend on
    hipLaunchKernelGGL(gap_fwd_kernel, dim3((HW + rows - 1) / rows, N), dim3(256), 0, st, (const u32x4*)x, y, N, HW, C, rows);
    return hc_launch_status();
}
int hc_gap_bwd(const float* dy, void* dx, int32_t N, int32_t HW, int32_t C, hc_stream_t stream) {
    if (dy == nullptr || dx == nullptr || (C % 8) != 0) return HC_ERR_ARG;
    const long total = (long)N * HW * (C / 8);
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, (u32x4*)dx, N, HW, C);
    return hc_launch_status();
}

int hc_nchw_to_nhwc_bf16(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cpad, hc_stream_t stream) {
    if (x == nullptr || y == nullptr || Cpad < C) return HC_ERR_ARG;
    const long total = (long)N * H * W * Cpad;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, N,
                       C, H, W, Cpad);
    return hc_launch_status();
}
int hc_nhwc_bf16_to_nchw(const void* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cpad, hc_stream_t stream) {
    if (x == nullptr || y == nullptr || Cpad < C) return HC_ERR_ARG;
    const long total = (long)N * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       y, N, C, H, W, Cpad);
    return hc_launch_status();
}
int hc_pack_conv_weight(const float* w, void* wpk, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, int32_t mode, int32_t tap0,
                        int32_t T, hc_stream_t stream) {
    if (w == nullptr || wpk == nullptr || mode < 0 || mode > 4) return HC_ERR_ARG;     // modes 5 / 6 need `ld`: hc_pack_conv_weights_multi
    if (mode >= 3 && (Cout != Cin || Cout % 48 != 0)) return HC_ERR_ARG;
    const long total = (long)Cout * Cin * KH * KW;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)wpk, Cout, Cin, KH,
                       KW, mode, tap0, T, 0);
    return hc_launch_status();
}
int hc_pack_conv_weights_multi(const hc_pack_item* items, int32_t nitems, int64_t max_elems, hc_stream_t stream) {
    if (items == nullptr || nitems < 0) return HC_ERR_ARG;
    if (nitems == 0) return HC_OK;
    // a flat list of work units (LDS tiles / element chunks) walked by a fixed-size grid; max_elems bounds the units of one item
    long units = (long)nitems * ((max_elems + 1023) / 1024);
    int bx = units > 4096 ? 4096 : (int)units;
    if (bx < 1) bx = 1;
    if (nitems > 8000) return HC_ERR_ARG;            // the prefix array lives in LDS
    hipLaunchKernelGGL(pack_weight_multi_kernel, dim3(bx), dim3(256), (nitems + 1) * sizeof(int), (hipStream_t)stream, items, nitems);
    return hc_launch_status();
}
int hc_im2col_small(const float* x, void* col, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t OH, int32_t OW, int32_t KH,
                    int32_t KW, int32_t stride, int32_t pad, int32_t Kpad, hc_stream_t stream) {
    if (x == nullptr || col == nullptr || (Kpad % 8) != 0 || Cin * KH * KW > Kpad) return HC_ERR_ARG;
    if ((long)Cin * H * W > 2147483647L) return HC_ERR_ARG;          // offsets inside an image are 32-bit
    if (N <= 0 || OH <= 0 || OW <= 0) return HC_OK;
    const dim3 grid = im2col_grid(N, OH, OW, Kpad);
    if (im2col3_lds_ok(Cin, KH, KW, pad, W, OW, stride, Kpad)) {
        const long rows = (long)N * OH;
        hipLaunchKernelGGL(im2col3_lds_kernel, dim3((unsigned)(rows > 65535 ? 65535 : rows)), dim3(256), im2col3_lds_bytes(W, Kpad), (hipStream_t)stream, x,
                           (bf16_t*)col, N, H, W, OH, OW, stride, Kpad);
        return hc_launch_status();
    }
    if (Cin == 3 && KH == 3 && KW == 3)
        hipLaunchKernelGGL((im2col_small_kernel<3, 3>), grid, dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)col, N, Cin, H, W, OH, OW, KH, KW,
                           stride, pad, Kpad);
    else
        hipLaunchKernelGGL((im2col_small_kernel<0, 0>), grid, dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)col, N, Cin, H, W, OH, OW, KH, KW,
                           stride, pad, Kpad);
    return hc_launch_status();
}
int hc_im2col_small_fp8(const float* x, void* col, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t OH, int32_t OW, int32_t KH,
                        int32_t KW, int32_t stride, int32_t pad, int32_t Kpad, float inv_scale, hc_stream_t stream) {
    if (x == nullptr || col == nullptr || (Kpad % 8) != 0 || Cin * KH * KW > Kpad) return HC_ERR_ARG;
    if ((long)Cin * H * W > 2147483647L) return HC_ERR_ARG;
    if (N <= 0 || OH <= 0 || OW <= 0) return HC_OK;
    const dim3 grid = im2col_grid(N, OH, OW, Kpad);
    if (im2col3_lds_ok(Cin, KH, KW, pad, W, OW, stride, Kpad)) {
        const long rows = (long)N * OH;
        hipLaunchKernelGGL(im2col3_lds_fp8_kernel, dim3((unsigned)(rows > 65535 ? 65535 : rows)), dim3(256), im2col3_lds_bytes(W, Kpad), (hipStream_t)stream, x,
                           (unsigned char*)col, N, H, W, OH, OW, stride, Kpad, inv_scale);
        return hc_launch_status();
    }
    if (Cin == 3 && KH == 3 && KW == 3)
        hipLaunchKernelGGL((im2col_small_fp8_kernel<3, 3>), grid, dim3(256), 0, (hipStream_t)stream, x, (unsigned char*)col, N, Cin, H, W, OH,
                           OW, KH, KW, stride, pad, Kpad, inv_scale);
    else
        hipLaunchKernelGGL((im2col_small_fp8_kernel<0, 0>), grid, dim3(256), 0, (hipStream_t)stream, x, (unsigned char*)col, N, Cin, H, W, OH,
                           OW, KH, KW, stride, pad, Kpad, inv_scale);
    return hc_launch_status();
}
int hc_unpack_im2col_grad(const float* dwcol, float* dw, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, int32_t Kpad,
                          int32_t beta, hc_stream_t stream) {
    if (dwcol == nullptr || dw == nullptr) return HC_ERR_ARG;
    const int total = Cout * Cin * KH * KW;
    hipLaunchKernelGGL(unpack_im2col_grad_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, dwcol, dw, Cout, Cin,
                       KH, KW, Kpad, beta);
    return hc_launch_status();
}

const char* hc_version(void) { return "holocron_hip 0.1 (gfx950)"; }

}  // extern "C"
