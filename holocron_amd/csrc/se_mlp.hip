// The squeeze-excite MLP of ReXNet (holocron/models/classification/rexnet.py:38-66: 1x1 conv C -> C/r without bias, BatchNorm2d with
// batch statistics over the N pooled vectors, activation, 1x1 conv C/r -> C with bias; the sigmoid belongs to the gate kernel) as
// two forward and four backward launches on the pooled [N][Cp] vectors.
//
// Through the generic units this MLP was ~19 launches per block and step (cast copies, two gather-convs on 256 "pixels", BatchNorm
// finalize / apply passes, weight-gradient launches with their split reductions, column sums for the bias): 1.9 ms of kernel time per
// rexnet1_0x step for 46 MFLOP per GEMM at most - latency, not work.  Here every GEMM is one launch of the same wave-level kernel:
//   forward    FC1    h1[n][r] = sum_c pooled[n][c] W1[r][c]  + partial sums of h1, h1^2 per 32-row tile (no atomics: fixed order)
//              FC2    statistics -> (a, shift); logits[n][c] = b2[c] + sum_r act(a h1 + shift)[n][r] W2[c][r]  (bf16, pad channels 0)
//   backward   DH     dh[n][r] = sum_c dl[n][c] W2[c][r];  g = dh act'(.)  + partial sums of g, g xhat
//              DW2    dW2[c][r] = sum_n dl[n][c] h[n][r], db2[c] = sum_n dl[n][c]
//              DPOOL  dh1 = a (g - mean(g) - xhat mean(g xhat));  dpool[n][c] = sum_r dh1[n][r] W1[r][c]   (fp32, pad channels 0)
//              DW1    dW1[r][c] = sum_n dh1[n][r] pooled[n][c];  dgamma = sum g xhat, dbeta = sum g
// A workgroup owns a 32 x 32 output tile; its waves split the k-steps (16 wide, v_mfma_f32_32x32x16_bf16) and every wave loads ITS
// operand fragments straight from global memory into registers - all loads of a wave are issued before the first MFMA, so a launch
// is one memory round trip, a handful of MFMAs, one LDS reduction over the waves (fixed order) and the epilogue: a first version
// with fp32 FMAs out of LDS tiles filled by load - store loops took 20-85 us per launch (one round trip per loop iteration).
// Operands are rounded to bf16 like every other convolution of the framework (the generic units rounded the pooled vectors and the
// weights too); accumulation, BatchNorm statistics and every stored intermediate are fp32.
#include "common.h"
#include "../../include/holocron_hip.h"

namespace {

constexpr int SE_RMAX = 128;   // reduced channels supported
constexpr int SE_TM = 32;      // rows per statistics tile = MFMA tile edge

enum { FC1 = 0, FC2 = 1, DH = 2, DW2 = 3, DPOOL = 4, DW1 = 5 };

__device__ __forceinline__ float se_act(float y, int act) {
    if (act == 1) return y > 0.f ? y : 0.f;
    if (act == 6) return fminf(fmaxf(y, 0.f), 6.f);
    return y;
}
__device__ __forceinline__ float se_dact(float y, int act) {
    if (act == 1) return y > 0.f ? 1.f : 0.f;
    if (act == 6) return (y > 0.f && y < 6.f) ? 1.f : 0.f;
    return 1.f;
}
__device__ __forceinline__ bf16x8 se_pack8(const float (&v)[8]) {
    const u32x4 p = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    return __builtin_bit_cast(bf16x8, p);
}

// per-channel vectors of the BatchNorm in the middle, in LDS: a = gamma rstd, b = beta - mean a, mean, rstd, mg = mean(g), mgx = mean(g xhat)
struct SeCoef { float a[SE_RMAX], b[SE_RMAX], mean[SE_RMAX], rstd[SE_RMAX], mg[SE_RMAX], mgx[SE_RMAX]; };

// FROM_PART: forward (FC2) - statistics from the partial sums of h1, h1^2; the `writer` workgroup stores them and updates the running
// statistics.  Otherwise: from the saved statistics; WITH_G adds the backward means from part2 (and dgamma / dbeta by the writer).
// sum over the row tiles of part[t][0 | 1][r], in tile order; the loads of 8 tiles are issued together (one round trip at batch 256)
__device__ __forceinline__ void se_tile_sums(const float* __restrict__ part, int tiles, int R, int r, float& s1, float& s2) {
    s1 = 0.f;
    s2 = 0.f;
    for (int t0 = 0; t0 < tiles; t0 += 8) {
        float p1[8], p2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = (t0 + u < tiles) ? t0 + u : 0;
            p1[u] = part[((size_t)t * 2 + 0) * R + r];
            p2[u] = part[((size_t)t * 2 + 1) * R + r];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s1 += (t0 + u < tiles) ? p1[u] : 0.f;
            s2 += (t0 + u < tiles) ? p2[u] : 0.f;
        }
    }
}

template <bool FROM_PART, bool WITH_G>
__device__ __forceinline__ void se_coefs(const hc_se_mlp_desc& d, SeCoef& co, const bool writer) {
    const int N = d.N, R = d.R, tiles = (N + SE_TM - 1) / SE_TM;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        float mean, rstd;
        const float gamma = d.gamma[r], beta = d.beta[r];
        if (FROM_PART) {
            float s1, s2;
            se_tile_sums(d.part, tiles, R, r, s1, s2);
            mean = s1 / (float)N;
            const float var = fmaxf(s2 / (float)N - mean * mean, 0.f);
            rstd = rsqrtf(var + d.eps);
            if (writer) {
                d.stat[r] = mean;
                d.stat[R + r] = rstd;
                if (d.running_mean != nullptr) {
                    const float unb = N > 1 ? var * (float)N / (float)(N - 1) : var;
                    d.running_mean[r] = (1.f - d.momentum) * d.running_mean[r] + d.momentum * mean;
                    d.running_var[r] = (1.f - d.momentum) * d.running_var[r] + d.momentum * unb;
                }
            }
        } else {
            mean = d.stat[r];
            rstd = d.stat[R + r];
        }
        if (WITH_G) {
            float s1, s2;
            se_tile_sums(d.part2, tiles, R, r, s1, s2);
            co.mg[r] = s1 / (float)N;
            co.mgx[r] = s2 / (float)N;
            if (writer && d.dgamma != nullptr) {
                d.dgamma[r] = s2;
                d.dbeta[r] = s1;
            }
        }
        const float a = gamma * rstd;
        co.a[r] = a;
        co.b[r] = beta - mean * a;
        co.mean[r] = mean;
        co.rstd[r] = rstd;
    }
    if (FROM_PART && writer && threadIdx.x == 0 && d.num_batches_tracked != nullptr) d.num_batches_tracked[0] += 1;
}

// ---- operand fragments: 8 consecutive k of one row.
// Two phases, and BRANCH-FREE on purpose.  se_raw_*: address arithmetic and unconditional loads from clamped (always valid) indices,
// nothing that touches a loaded value; se_fin_*: selects, BatchNorm / activation arithmetic, rounding to bf16.  The kernel puts a
// scheduling barrier between the two: all loads of a wave are in flight before the first value is used.  (With a branch per guarded
// element the wait-count pass put s_waitcnt vmcnt(0) at every join; with the selects next to the loads the scheduler sank every
// k-step's loads in front of its MFMA - one memory round trip per k-step either way.)
typedef unsigned int u32;
__device__ __forceinline__ u32 se_ld32(const float* __restrict__ p, size_t idx, bool ok) {
    return reinterpret_cast<const u32*>(p)[ok ? idx : 0];
}
__device__ __forceinline__ float se_f(u32 bits) { return __builtin_bit_cast(float, bits); }

// `on`: this k-step exists (wave-uniform); a k-step past the end loads from clamped addresses and contributes zeros
template <int KIND>
__device__ __forceinline__ void se_raw_a(const hc_se_mlp_desc& d, int i, int k0, bool on, u32 (&r)[16]) {
    const int N = d.N, C = d.C, Cp = d.Cp, R = d.R;
    if (KIND == FC1) {               // pooled[n = i][c = k0 ..]: 32 contiguous, aligned bytes (Cp >= ceil16(C), pad channels are zeros)
        const bool ok = on && i < N;
        const float* p = d.pooled + (ok ? (size_t)i * Cp + k0 : 0);
        const u32x4 lo = *reinterpret_cast<const u32x4*>(p);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { r[e] = lo[e]; r[4 + e] = hi[e]; }
    } else if (KIND == DH) {         // dl[n = i][c = k0 ..] bf16: one 16-byte load
        const bool ok = on && i < N;
        const u32x4 p = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(d.dl) + (ok ? (size_t)i * Cp + k0 : 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = p[e];
    } else if (KIND == FC2 || KIND == DPOOL) {   // h1 (and g) [n = i][r = k0 ..]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = on && i < N && k0 + e < R;
            const size_t idx = (size_t)i * R + k0 + e;
            r[e] = se_ld32(d.h1, idx, ok);
            if (KIND == DPOOL) r[8 + e] = se_ld32(d.g, idx, ok);
        }
    } else if (KIND == DW2) {        // dl[n = k0 ..][c = i]
        const bf16_t* p = reinterpret_cast<const bf16_t*>(d.dl);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = on && i < C && k0 + e < N;
            r[e] = p[ok ? (size_t)(k0 + e) * Cp + i : 0];
        }
    } else {                         // DW1: h1, g [n = k0 ..][r = i]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = on && i < R && k0 + e < N;
            const size_t idx = (size_t)(k0 + e) * R + i;
            r[e] = se_ld32(d.h1, idx, ok);
            r[8 + e] = se_ld32(d.g, idx, ok);
        }
    }
}
template <int KIND>
__device__ __forceinline__ void se_raw_b(const hc_se_mlp_desc& d, int j, int k0, bool on, u32 (&r)[8]) {
    const int N = d.N, C = d.C, Cp = d.Cp, R = d.R;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (KIND == FC1) r[e] = se_ld32(d.w1, (size_t)j * C + k0 + e, on && j < R && k0 + e < C);              // W1[r = j][c = k0 ..]
        else if (KIND == DH) r[e] = se_ld32(d.w2, (size_t)(k0 + e) * R + j, on && j < R && k0 + e < C);        // W2[c = k0 ..][r = j]
        else if (KIND == FC2) r[e] = se_ld32(d.w2, (size_t)j * R + k0 + e, on && j < C && k0 + e < R);         // W2[c = j][r = k0 ..]
        else if (KIND == DPOOL) r[e] = se_ld32(d.w1, (size_t)(k0 + e) * C + j, on && j < C && k0 + e < R);     // W1[r = k0 ..][c = j]
        else if (KIND == DW2) r[e] = se_ld32(d.h1, (size_t)(k0 + e) * R + j, on && j < R && k0 + e < N);       // h1[n = k0 ..][r = j]
        else r[e] = se_ld32(d.pooled, (size_t)(k0 + e) * Cp + j, on && j < C && k0 + e < N);                   // pooled[n = k0 ..][c = j]
    }
}

// hidden activation / its gradient from the raw h1 (and g) bits of element (n, r); channels past R read coefficient slot 0 and are
// deselected
__device__ __forceinline__ float se_h(const SeCoef& co, int act, u32 h1, int r, bool ok) {
    const int rr = ok ? r : 0;
    const float y = co.a[rr] * se_f(h1) + co.b[rr];
    return ok ? se_act(y, act) : 0.f;
}
__device__ __forceinline__ float se_dh1(const SeCoef& co, u32 h1, u32 g, int r, bool ok) {
    const int rr = ok ? r : 0;
    const float xh = (se_f(h1) - co.mean[rr]) * co.rstd[rr];
    const float v = co.a[rr] * (se_f(g) - co.mg[rr] - xh * co.mgx[rr]);
    return ok ? v : 0.f;
}

template <int KIND>
__device__ __forceinline__ bf16x8 se_fin_a(const hc_se_mlp_desc& d, const SeCoef& co, int i, int k0, bool on, const u32 (&r)[16], float& bsum) {
    const int N = d.N, C = d.C, R = d.R;
    float v[8];
    if (KIND == DH) {
        const bool ok = on && i < N;
        const u32x4 z = {ok ? r[0] : 0u, ok ? r[1] : 0u, ok ? r[2] : 0u, ok ? r[3] : 0u};
        return __builtin_bit_cast(bf16x8, z);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (KIND == FC1) v[e] = (on && i < N) ? se_f(r[e]) : 0.f;
        else if (KIND == FC2) v[e] = se_h(co, d.act, r[e], k0 + e, on && i < N && k0 + e < R);
        else if (KIND == DPOOL) v[e] = se_dh1(co, r[e], r[8 + e], k0 + e, on && i < N && k0 + e < R);
        else if (KIND == DW2) { v[e] = (on && i < C && k0 + e < N) ? bf16_to_f32((bf16_t)r[e]) : 0.f; bsum += v[e]; }
        else v[e] = se_dh1(co, r[e], r[8 + e], i, on && i < R && k0 + e < N);
    }
    return se_pack8(v);
}
template <int KIND>
__device__ __forceinline__ bf16x8 se_fin_b(const hc_se_mlp_desc& d, const SeCoef& co, int j, int k0, bool on, const u32 (&r)[8]) {
    const int N = d.N, C = d.C, R = d.R;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (KIND == FC1 || KIND == DH) v[e] = (on && j < R && k0 + e < C) ? se_f(r[e]) : 0.f;
        else if (KIND == FC2 || KIND == DPOOL) v[e] = (on && j < C && k0 + e < R) ? se_f(r[e]) : 0.f;
        else if (KIND == DW2) v[e] = se_h(co, d.act, r[e], j, on && j < R && k0 + e < N);
        else v[e] = (on && j < C && k0 + e < N) ? se_f(r[e]) : 0.f;
    }
    return se_pack8(v);
}

// D[i][j] = sum_k A[i][k] B[j][k] on a 32 x 32 tile; NW waves split the k-steps, U k-steps per wave and round.
// Accumulator layout of v_mfma_f32_32x32x16_bf16 (as in conv_gather.hip): acc[q] = D[(q & 3) + 8 (q >> 2) + 4 (lane >> 5)][lane & 31];
// operand lane = (row lane & 31, k = 8 (lane >> 5) + [0, 8)).
template <int KIND, int NW, int U>
__global__ __launch_bounds__(64 * NW) void se_gemm_kernel(const hc_se_mlp_desc d) {
    __shared__ SeCoef co;
    __shared__ float red[NW][16][64];
    __shared__ float redb[NW][64];
    const int N = d.N, C = d.C, Cp = d.Cp, R = d.R;
    const int I = (KIND == DW2) ? C : (KIND == DW1 ? R : N);
    const int J = (KIND == FC1 || KIND == DH || KIND == DW2) ? R : ((KIND == DW1) ? C : Cp);
    const int K = (KIND == FC1 || KIND == DH) ? C : ((KIND == FC2 || KIND == DPOOL) ? R : N);
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    const bool first_wg = blockIdx.x == 0 && blockIdx.y == 0;
    if (KIND == FC2) se_coefs<true, false>(d, co, first_wg);
    if (KIND == DW2) se_coefs<false, false>(d, co, false);
    if (KIND == DPOOL) se_coefs<false, true>(d, co, false);
    if (KIND == DW1) se_coefs<false, true>(d, co, first_wg);
    if (KIND != FC1 && KIND != DH) __syncthreads();

    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float bsum = 0.f;                          // DW2: sum over this lane's n of dl[n][c] (the bias gradient)
    const int nk = (K + 15) / 16;
    for (int base = 0; base < nk; base += NW * U) {
        u32 ra[U][16], rb[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ks = base + u * NW + wave;            // wave-uniform; a k-step past the end multiplies zeros (no branches)
            const int k0 = ks * 16 + 8 * lh;
            se_raw_a<KIND>(d, i0 + lr, k0, ks < nk, ra[u]);
            se_raw_b<KIND>(d, j0 + lr, k0, ks < nk, rb[u]);
        }
        __builtin_amdgcn_sched_barrier(0);                  // every load above is issued before any value below is touched
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ks = base + u * NW + wave;
            const int k0 = ks * 16 + 8 * lh;
            const bf16x8 a = se_fin_a<KIND>(d, co, i0 + lr, k0, ks < nk, ra[u], bsum);
            const bf16x8 b = se_fin_b<KIND>(d, co, j0 + lr, k0, ks < nk, rb[u]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    }
    // the waves' partial tiles, added in wave order by wave 0
#pragma unroll
    for (int q = 0; q < 16; ++q) red[wave][q][lane] = acc[q];
    if (KIND == DW2) redb[wave][lane] = bsum;
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        float s = red[0][q][lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += red[w][q][lane];
        acc[q] = s;
    }
    const int j = j0 + lr;
    if (KIND == FC1 || KIND == DH) {           // rows n, columns r: store + per-tile column sums
        float mean = 0.f, rstd = 0.f, a = 0.f, sh = 0.f;
        if (KIND == DH && j < R) {
            mean = d.stat[j];
            rstd = d.stat[R + j];
            a = d.gamma[j] * rstd;
            sh = d.beta[j] - mean * a;
        }
        float s1 = 0.f, s2 = 0.f;
        float h1v[16];
        if (KIND == DH) {                      // the 16 h1 values of this lane's column in one round trip (clamped indices)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int n = i0 + (q & 3) + 8 * (q >> 2) + 4 * lh;
                h1v[q] = d.h1[(n < N && j < R) ? (size_t)n * R + j : 0];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int n = i0 + (q & 3) + 8 * (q >> 2) + 4 * lh;
            if (n >= N || j >= R) continue;
            if (KIND == FC1) {
                d.h1[(size_t)n * R + j] = acc[q];
                s1 += acc[q];
                s2 += acc[q] * acc[q];
            } else {
                const float h1 = h1v[q];
                const float g = acc[q] * se_dact(a * h1 + sh, d.act);
                d.g[(size_t)n * R + j] = g;
                s1 += g;
                s2 += g * ((h1 - mean) * rstd);
            }
        }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (lh == 0 && j < R) {
            float* part = KIND == FC1 ? d.part : d.part2;
            part[((size_t)blockIdx.x * 2 + 0) * R + j] = s1;
            part[((size_t)blockIdx.x * 2 + 1) * R + j] = s2;
        }
    } else if (KIND == FC2 || KIND == DPOOL) { // rows n, columns c
        if (j >= Cp) return;
        const float bias = (KIND == FC2 && d.b2 != nullptr && j < C) ? d.b2[j] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int n = i0 + (q & 3) + 8 * (q >> 2) + 4 * lh;
            if (n >= N) continue;
            const float v = j < C ? acc[q] + bias : 0.f;
            if (KIND == FC2) reinterpret_cast<bf16_t*>(d.logits)[(size_t)n * Cp + j] = f32_to_bf16(v);
            else d.dpool[(size_t)n * Cp + j] = v;
        }
    } else if (KIND == DW2) {                  // rows c, columns r
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = i0 + (q & 3) + 8 * (q >> 2) + 4 * lh;
            if (c < C && j < R) d.dw2[(size_t)c * R + j] = acc[q];
        }
        if (blockIdx.y == 0 && d.db2 != nullptr) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += redb[w][lane];
            s += __shfl_xor(s, 32, 64);        // the two k halves of a row
            if (lh == 0 && i0 + lr < C) d.db2[i0 + lr] = s;
        }
    } else {                                   // DW1: rows r, columns c
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int r = i0 + (q & 3) + 8 * (q >> 2) + 4 * lh;
            if (r < R && j < C) d.dw1[(size_t)r * C + j] = acc[q];
        }
    }
}

bool se_desc_ok(const hc_se_mlp_desc& d) {
    return d.N > 0 && d.C > 0 && d.Cp >= (d.C + 15) / 16 * 16 && (d.Cp % 8) == 0 && d.R > 0 && d.R <= SE_RMAX && (d.act == 0 || d.act == 1 || d.act == 6) &&
           d.pooled != nullptr && d.w1 != nullptr && d.gamma != nullptr && d.beta != nullptr && d.w2 != nullptr && d.h1 != nullptr &&
           d.stat != nullptr && (reinterpret_cast<unsigned long long>(d.pooled) & 15ull) == 0;
}

template <int KIND, int NW, int U>
void se_launch(const hc_se_mlp_desc& d, int I, int J, hipStream_t st) {
    hipLaunchKernelGGL((se_gemm_kernel<KIND, NW, U>), dim3((I + 31) / 32, (J + 31) / 32), dim3(64 * NW), 0, st, d);
}

}  // namespace

extern "C" {

int64_t hc_se_mlp_part_floats(int32_t N, int32_t R) { return (int64_t)((N + SE_TM - 1) / SE_TM) * 2 * R; }

int hc_se_mlp_fwd(const hc_se_mlp_desc* dp, hc_stream_t stream) {
    if (dp == nullptr) return HC_ERR_ARG;
    const hc_se_mlp_desc& d = *dp;
    if (!se_desc_ok(d) || d.part == nullptr || d.logits == nullptr) return HC_ERR_ARG;
    if ((d.running_mean == nullptr) != (d.running_var == nullptr)) return HC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nk = (d.C + 15) / 16;              // K = C: k-steps per wave and round sized to the layer (unused slots load and multiply zeros)
    if (nk <= 24) se_launch<FC1, 8, 3>(d, d.N, d.R, st);
    else if (nk <= 40) se_launch<FC1, 8, 5>(d, d.N, d.R, st);
    else se_launch<FC1, 8, 10>(d, d.N, d.R, st);  // up to 1280 channels in one round
    se_launch<FC2, 4, 2>(d, d.N, d.Cp, st);     // K = R <= 128: 8 k-steps
    return hc_launch_status();
}

int hc_se_mlp_bwd(const hc_se_mlp_desc* dp, hc_stream_t stream) {
    if (dp == nullptr) return HC_ERR_ARG;
    const hc_se_mlp_desc& d = *dp;
    if (!se_desc_ok(d) || d.dl == nullptr || d.g == nullptr || d.part2 == nullptr || d.dpool == nullptr || d.dw1 == nullptr ||
        d.dw2 == nullptr || (d.dgamma == nullptr) != (d.dbeta == nullptr) || (reinterpret_cast<unsigned long long>(d.dl) & 15ull) != 0)
        return HC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if ((d.C + 15) / 16 <= 24) se_launch<DH, 8, 3>(d, d.N, d.R, st);
    else se_launch<DH, 8, 5>(d, d.N, d.R, st);   // (10 k-steps per wave spill: 64-bit addresses of the strided W2 loads)
    se_launch<DW2, 8, 2>(d, d.C, d.R, st);      // K = N: 16 k-steps at batch 256
    se_launch<DPOOL, 4, 2>(d, d.N, d.Cp, st);
    se_launch<DW1, 8, 2>(d, d.R, d.C, st);
    return hc_launch_status();
}

}  // extern "C"
