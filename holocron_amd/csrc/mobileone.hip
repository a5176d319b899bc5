// MobileOne over-parameterised blocks in training form (reference: holocron/models/classification/mobileone.py:31-176).
// A block output is  act( sum_b BN_b(y_b) )  where the y_b are the outputs of B parallel linear branches of the same input
// (K depthwise 3x3 + one depthwise 1x1 + the input itself, or K dense 1x1 + the input itself), every branch with its own
// BatchNorm.  The branch outputs are produced by the convolution kernels (statistics in their epilogues); this file holds the
// "multi-source BatchNorm sum" passes around them and the multi-plane depthwise data gradient:
//   msbn_finalize      statistics -> per-branch scale / shift (+ running statistics)                     [B x C threads]
//   msbn_apply         out = act(sum_b scale_b y_b + shift)            reads B sources, writes 1          HBM-bound
//   msbn_bwd_reduce    sum gz, sum gz y_b  (gz = g act'(out))          reads B + 2 sources                HBM-bound
//   msbn_bwd_finalize  d gamma_b, d beta_b, the three coefficients of  dy_b = k1 gz + k2 y_b + k3
//   msbn_bwd_apply     writes the B branch gradients                   reads B + 2, writes B              HBM-bound
//   dwrep_dgrad        dx = sum_b dwconv3x3^T(dy_b, w_b) + extra       reads P planes, writes 1           HBM-bound
// A thread owns 8 channels (one 16-byte chunk) of a pixel and keeps the per-channel coefficients of every branch in
// registers; launches keep (threads % channel groups) == 0 so that the channel group of a thread never changes.
#include "common.h"
#include "../../include/holocron_hip.h"

namespace {

constexpr int MB_THREADS = 256;
constexpr int MAXB = HC_MSBN_MAX_BRANCHES;

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
__device__ __forceinline__ void load8f(const float* __restrict__ p, float (&f)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = a[e]; f[4 + e] = b[e]; }
}

// blocks such that (blocks * 256) % cg == 0
inline int mb_blocks(long items, int cg, int per_thread) {
    int a = cg, b = MB_THREADS;
    while (b) { int t = a % b; a = b; b = t; }
    const int unit = cg / a;
    long want = items / ((long)MB_THREADS * per_thread);
    long floor_blocks = items / MB_THREADS;
    if (floor_blocks > 1024) floor_blocks = 1024;
    if (want < floor_blocks) want = floor_blocks;
    if (want > 4096) want = 4096;
    if (want < 1) want = 1;
    const long k = (want + unit - 1) / unit;
    return (int)(k * unit);
}

// per-workgroup reduction of v[S][8] over the threads that share a channel group, then one atomic per (k, c).  The rows go
// through LDS CH at a time (256 x (8 CH + 1) floats): a 9- or 7-row reduction in one piece would take 58-75 KB of LDS and
// leave two workgroups per CU on these HBM-bound kernels.  lds: MB_THREADS * (8 * CH + 1) floats.
template <int S, int CH = (S < 3 ? S : 3)>
__device__ __forceinline__ void block_reduce_flush(const float (&v)[S][8], int cg, int C, float* __restrict__ dst, float* __restrict__ lds) {
    constexpr int STR = CH * 8 + 1;
    const int tid = threadIdx.x;
    const int blockbase = (int)(((long)blockIdx.x * MB_THREADS) % cg);
#pragma unroll
    for (int k0 = 0; k0 < S; k0 += CH) {
        if (k0) __syncthreads();
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k0 + k < S) {
#pragma unroll
                for (int e = 0; e < 8; ++e) lds[tid * STR + k * 8 + e] = v[k0 + k][e];
            }
        __syncthreads();
        const int rows = (S - k0) < CH ? (S - k0) : CH;
        for (int o = tid; o < rows * C; o += MB_THREADS) {
            const int k = o / C, c = o - k * C;
            const int g = c >> 3, e = c & 7;
            int first = g - blockbase;
            if (first < 0) first += cg;
            float sum = 0.f;
            for (int t = first; t < MB_THREADS; t += cg) sum += lds[t * STR + k * 8 + e];
            atomicAdd(dst + (size_t)(k0 + k) * C + c, sum);
        }
    }
}

// ---------------------------------------------------------------- statistics -> affine
// 8 lanes per (branch, channel): each sums 16 of the reps partial sums, a butterfly combines them
constexpr int FIN_SUB = 8;
__device__ __forceinline__ float sub_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}

__global__ __launch_bounds__(256) void msbn_finalize_kernel(const hc_msbn_desc d, const int reps) {
    const int i = (blockIdx.x * 256 + threadIdx.x) / FIN_SUB;
    const int sub = threadIdx.x & (FIN_SUB - 1);
    const bool live = i < d.B * d.C;
    const int b = live ? i / d.C : 0, c = live ? i - b * d.C : 0;
    const hc_msbn_branch br = d.br[b];
    const bool valid = live && c < d.c_valid;
    float s = 0.f, q = 0.f;
    if (valid && d.training) {
        for (int r = sub; r < reps; r += FIN_SUB) {
            s += br.stats[((size_t)r * 2 + 0) * br.stats_ld + c];
            q += br.stats[((size_t)r * 2 + 1) * br.stats_ld + c];
        }
    }
    s = sub_sum(s);
    q = sub_sum(q);
    if (!live || sub != 0) return;
    float scale = 0.f, shift = 0.f, mean = 0.f, rstd = 0.f;
    if (valid) {
        float var;
        if (d.training) {
            const float n = (float)d.count;
            mean = s / n;
            var = fmaxf(q / n - mean * mean, 0.f);
            if (br.running_mean != nullptr) {
                const float m = br.momentum;
                br.running_mean[c] = (1.f - m) * br.running_mean[c] + m * mean;
                br.running_var[c] = (1.f - m) * br.running_var[c] + m * var * (n / fmaxf(n - 1.f, 1.f));
            }
        } else {
            mean = br.running_mean[c];
            var = br.running_var[c];
        }
        rstd = 1.f / sqrtf(var + br.eps);
        scale = br.gamma[c] * rstd;
        shift = br.beta[c] - mean * scale;
    }
    d.coef[((size_t)b * 2 + 0) * d.C + c] = scale;
    d.coef[((size_t)b * 2 + 1) * d.C + c] = shift;
    d.save[((size_t)b * 2 + 0) * d.C + c] = mean;
    d.save[((size_t)b * 2 + 1) * d.C + c] = rstd;
    if (c == 0 && d.training && br.num_batches_tracked != nullptr) *br.num_batches_tracked += 1;
}

// red: [reps][B + 1][C]; k = 0: sum gz, k = 1 + b: sum gz * y_b
__global__ __launch_bounds__(256) void msbn_bwd_finalize_kernel(const hc_msbn_desc d, const int reps) {
    const int i = (blockIdx.x * 256 + threadIdx.x) / FIN_SUB;
    const int sub = threadIdx.x & (FIN_SUB - 1);
    const bool live = i < d.B * d.C;
    const int b = live ? i / d.C : 0, c = live ? i - b * d.C : 0;
    const hc_msbn_branch br = d.br[b];
    const bool valid = live && c < d.c_valid;
    float sg = 0.f, sgy = 0.f;
    if (valid) {
        for (int r = sub; r < reps; r += FIN_SUB) {
            sg += d.red[((size_t)r * (d.B + 1) + 0) * d.C + c];
            sgy += d.red[((size_t)r * (d.B + 1) + 1 + b) * d.C + c];
        }
    }
    sg = sub_sum(sg);
    sgy = sub_sum(sgy);
    if (!live || sub != 0) return;
    float k1 = 0.f, k2 = 0.f, k3 = 0.f;
    if (valid) {
        const float mean = d.save[((size_t)b * 2 + 0) * d.C + c], rstd = d.save[((size_t)b * 2 + 1) * d.C + c];
        const float dgam = rstd * (sgy - mean * sg);
        const float A = br.gamma[c] * rstd;
        k1 = A;
        if (d.training) {
            const float n = (float)d.count;
            const float t = rstd * dgam / n;
            k2 = -A * t;
            k3 = -A * sg / n + A * t * mean;
        }
        if (br.dgamma != nullptr) br.dgamma[c] = d.accumulate ? br.dgamma[c] + dgam : dgam;
        if (br.dbeta != nullptr) br.dbeta[c] = d.accumulate ? br.dbeta[c] + sg : sg;
    }
    d.bcoef[((size_t)b * 3 + 0) * d.C + c] = k1;
    d.bcoef[((size_t)b * 3 + 1) * d.C + c] = k2;
    d.bcoef[((size_t)b * 3 + 2) * d.C + c] = k3;
}

// ---------------------------------------------------------------- elementwise passes
struct Srcs {
    const u32x4* y[MAXB];
    int ld8[MAXB];
    u32x4* dy[MAXB];
    int dld8[MAXB];
};

template <int B, bool STATS>
__global__ __launch_bounds__(MB_THREADS) void msbn_apply_kernel(const Srcs s, const float* __restrict__ coef, u32x4* __restrict__ out,
                                                                float* __restrict__ out_stats, long npix, int C, int act, const int reps) {
    extern __shared__ float sred[];
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * MB_THREADS + threadIdx.x;
    const long pstep = (long)gridDim.x * MB_THREADS / cg;
    const int cgi = (int)(gtid % cg);
    const int c0 = cgi * 8;
    float a[B][8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[e] = 0.f;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        float t[8];
        load8f(coef + ((size_t)b * 2 + 0) * C + c0, a[b]);
        load8f(coef + ((size_t)b * 2 + 1) * C + c0, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) sh[e] += t[e];
    }
    float sv[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sv[0][e] = sv[1][e] = 0.f;
    for (long p = gtid / cg; p < npix; p += pstep) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = sh[e];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            float f[8];
            unpack8(s.y[b][p * s.ld8[b] + cgi], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += a[b][e] * f[e];
        }
        if (act == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaxf(o[e], 0.f);
        }
        const u32x4 pk = pack8(o);
        out[p * cg + cgi] = pk;
        if (STATS) {   // statistics of the tensor the consumer reads (rounded values)
            float r[8];
            unpack8(pk, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) { sv[0][e] += r[e]; sv[1][e] += r[e] * r[e]; }
        }
    }
    if (STATS) block_reduce_flush<2>(sv, cg, C, out_stats + (size_t)(blockIdx.x % reps) * 2 * C, sred);
}

template <int B>
__global__ __launch_bounds__(MB_THREADS) void msbn_bwd_reduce_kernel(const Srcs s, const u32x4* __restrict__ g, int g_ld8,
                                                                     const u32x4* __restrict__ out, float* __restrict__ red, long npix,
                                                                     int C, int act, const int reps) {
    extern __shared__ float sred[];
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * MB_THREADS + threadIdx.x;
    const long pstep = (long)gridDim.x * MB_THREADS / cg;
    const int cgi = (int)(gtid % cg);
    float sv[B + 1][8];
#pragma unroll
    for (int k = 0; k <= B; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) sv[k][e] = 0.f;
    for (long p = gtid / cg; p < npix; p += pstep) {
        float gz[8];
        unpack8(g[p * g_ld8 + cgi], gz);
        if (act == 1) {
            float o[8];
            unpack8(out[p * cg + cgi], o);
#pragma unroll
            for (int e = 0; e < 8; ++e) gz[e] = o[e] > 0.f ? gz[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) sv[0][e] += gz[e];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            float f[8];
            unpack8(s.y[b][p * s.ld8[b] + cgi], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) sv[1 + b][e] += gz[e] * f[e];
        }
    }
    block_reduce_flush<B + 1>(sv, cg, C, red + (size_t)(blockIdx.x % reps) * (B + 1) * C, sred);
}

template <int B>
__global__ __launch_bounds__(MB_THREADS) void msbn_bwd_apply_kernel(const Srcs s, const u32x4* __restrict__ g, int g_ld8,
                                                                    const u32x4* __restrict__ out, const float* __restrict__ bc,
                                                                    long npix, int C, int act) {
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * MB_THREADS + threadIdx.x;
    const long pstep = (long)gridDim.x * MB_THREADS / cg;
    const int cgi = (int)(gtid % cg);
    const int c0 = cgi * 8;
    float k1[B][8], k2[B][8], k3[B][8];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        load8f(bc + ((size_t)b * 3 + 0) * C + c0, k1[b]);
        load8f(bc + ((size_t)b * 3 + 1) * C + c0, k2[b]);
        load8f(bc + ((size_t)b * 3 + 2) * C + c0, k3[b]);
    }
    for (long p = gtid / cg; p < npix; p += pstep) {
        float gz[8];
        unpack8(g[p * g_ld8 + cgi], gz);
        if (act == 1) {
            float o[8];
            unpack8(out[p * cg + cgi], o);
#pragma unroll
            for (int e = 0; e < 8; ++e) gz[e] = o[e] > 0.f ? gz[e] : 0.f;
        }
#pragma unroll
        for (int b = 0; b < B; ++b) {
            if (s.dy[b] == nullptr) continue;
            float f[8], o[8];
            unpack8(s.y[b][p * s.ld8[b] + cgi], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = k1[b][e] * gz[e] + k2[b][e] * f[e] + k3[b][e];
            s.dy[b][p * s.dld8[b] + cgi] = pack8(o);
        }
    }
}

// ---------------------------------------------------------------- multi-plane depthwise data gradient
struct Planes {
    const u32x4* dy[MAXB];
    const float* w[MAXB];     // forward tap-major fp32 [9][C]
};

// stride 1: dx[n][h][w] = sum_b sum_t w_b[t] dy_b[n][h + 1 - kh][w + 1 - kw]  (+ extra);  strips of TW pixels along W
template <int TW>
__global__ __launch_bounds__(MB_THREADS) void dwrep_dgrad_s1_kernel(const Planes pl, int P, const u32x4* __restrict__ extra,
                                                                    u32x4* __restrict__ dx, int N, int H, int W, int C) {
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * MB_THREADS + threadIdx.x;
    const long nthreads = (long)gridDim.x * MB_THREADS;
    const int cgi = (int)(gtid % cg);
    const int strips_w = (W + TW - 1) / TW;
    const long nstrips = (long)N * H * strips_w;
    for (long s = gtid / cg; s < nstrips; s += nthreads / cg) {
        const unsigned su = (unsigned)s, ru = su / (unsigned)strips_w;   // 32-bit: the launcher bounds the strip count
        const int sw = (int)(su - ru * (unsigned)strips_w);
        const unsigned nu = ru / (unsigned)H;
        const int h = (int)(ru - nu * (unsigned)H);
        const long n = nu;
        const int w0 = sw * TW;
        float acc[TW][8];
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            if (extra != nullptr && w0 + j < W) {
                unpack8(extra[((n * H + h) * (long)W + w0 + j) * cg + cgi], acc[j]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
            }
        }
        for (int b = 0; b < P; ++b) {
            const u32x4* __restrict__ src = pl.dy[b];
            const float* __restrict__ wb = pl.w[b] + cgi * 8;
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {       // source row h + dh - 1 pairs with kernel row kh = 2 - dh
                const int sh = h + dh - 1;
                if (sh < 0 || sh >= H) continue;
                const u32x4* row = src + ((n * H + sh) * (long)W) * cg + cgi;
                float wr[3][8];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) load8f(wb + (size_t)((2 - dh) * 3 + kw) * C, wr[kw]);
#pragma unroll
                for (int c = 0; c < TW + 2; ++c) {
                    const int sc = w0 + c - 1;
                    if (sc < 0 || sc >= W) continue;
                    float f[8];
                    unpack8(row[(long)sc * cg], f);
#pragma unroll
                    for (int j = 0; j < TW; ++j) {
                        const int dw = c - j;           // source column = (w0 + j) + dw - 1, kernel column kw = 2 - dw
                        if (dw < 0 || dw > 2) continue;
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[j][e] += wr[2 - dw][e] * f[e];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TW; ++j)
            if (w0 + j < W) dx[((n * H + h) * (long)W + w0 + j) * cg + cgi] = pack8(acc[j]);
    }
}

// stride 2: dx[n][h][w] = sum_b sum over taps with (h + 1 - kh), (w + 1 - kw) even of w_b[kh][kw] dy_b[n][(h+1-kh)/2][(w+1-kw)/2]
__global__ __launch_bounds__(MB_THREADS) void dwrep_dgrad_s2_kernel(const Planes pl, int P, u32x4* __restrict__ dx, int N, int H, int W,
                                                                    int OH, int OW, int C) {
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * MB_THREADS + threadIdx.x;
    const long nthreads = (long)gridDim.x * MB_THREADS;
    const int cgi = (int)(gtid % cg);
    const long npix = (long)N * H * W;
    for (long p = gtid / cg; p < npix; p += nthreads / cg) {
        const unsigned pu = (unsigned)p, qu = pu / (unsigned)W;   // 32-bit: the launcher bounds the pixel count
        const int iw = (int)(pu - qu * (unsigned)W);
        const unsigned nu = qu / (unsigned)H;
        const int ih = (int)(qu - nu * (unsigned)H);
        const long n = nu;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int t = ih + 1 - kh;
            if (t < 0 || (t & 1)) continue;
            const int oh = t >> 1;
            if (oh >= OH) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int u = iw + 1 - kw;
                if (u < 0 || (u & 1)) continue;
                const int ow = u >> 1;
                if (ow >= OW) continue;
                const long q = ((n * OH + oh) * (long)OW + ow) * cg + cgi;
                for (int b = 0; b < P; ++b) {
                    float f[8], wr[8];
                    unpack8(pl.dy[b][q], f);
                    load8f(pl.w[b] + (size_t)(kh * 3 + kw) * C + cgi * 8, wr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += wr[e] * f[e];
                }
            }
        }
        dx[p * cg + cgi] = pack8(acc);
    }
}

bool fill_srcs(const hc_msbn_io* io, Srcs& s, bool need_dy) {
    if (io == nullptr || io->B < 1 || io->B > MAXB || io->C <= 0 || (io->C % 8) != 0 || io->npix < 0) return false;
    for (int b = 0; b < MAXB; ++b) {
        s.y[b] = nullptr; s.ld8[b] = 0; s.dy[b] = nullptr; s.dld8[b] = 0;
    }
    for (int b = 0; b < io->B; ++b) {
        if (io->y[b] == nullptr || io->ld[b] < io->C || (io->ld[b] % 8) != 0) return false;
        s.y[b] = (const u32x4*)io->y[b];
        s.ld8[b] = io->ld[b] / 8;
        if (need_dy && io->dy[b] != nullptr) {
            if (io->dld[b] < io->C || (io->dld[b] % 8) != 0) return false;
            s.dy[b] = (u32x4*)io->dy[b];
            s.dld8[b] = io->dld[b] / 8;
        }
    }
    return true;
}

#define MSBN_DISPATCH(B_, CALL)        \
    switch (B_) {                      \
        case 1: { constexpr int BB = 1; CALL; } break; \
        case 2: { constexpr int BB = 2; CALL; } break; \
        case 3: { constexpr int BB = 3; CALL; } break; \
        case 4: { constexpr int BB = 4; CALL; } break; \
        case 5: { constexpr int BB = 5; CALL; } break; \
        default: { constexpr int BB = 6; CALL; } break; \
    }

bool desc_ok(const hc_msbn_desc* d) {
    if (d == nullptr || d->B < 1 || d->B > MAXB || d->C <= 0 || d->c_valid < 0 || d->c_valid > d->C || d->count <= 0) return false;
    return true;
}

}  // namespace

extern "C" {

int hc_msbn_finalize(const hc_msbn_desc* d, hc_stream_t stream) {
    if (!desc_ok(d) || d->coef == nullptr || d->save == nullptr) return HC_ERR_ARG;
    for (int b = 0; b < d->B; ++b) {
        const hc_msbn_branch& br = d->br[b];
        if (br.gamma == nullptr || br.beta == nullptr) return HC_ERR_ARG;
        if (d->training ? (br.stats == nullptr || br.stats_ld < d->c_valid) : (br.running_mean == nullptr || br.running_var == nullptr))
            return HC_ERR_ARG;
        if ((br.running_mean == nullptr) != (br.running_var == nullptr)) return HC_ERR_ARG;
    }
    const int n = d->B * d->C * FIN_SUB;
    hipLaunchKernelGGL(msbn_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *d, hc_get_stat_replicas());
    return hc_launch_status();
}

int hc_msbn_bwd_finalize(const hc_msbn_desc* d, hc_stream_t stream) {
    if (!desc_ok(d) || d->red == nullptr || d->save == nullptr || d->bcoef == nullptr) return HC_ERR_ARG;
    for (int b = 0; b < d->B; ++b)
        if (d->br[b].gamma == nullptr) return HC_ERR_ARG;
    const int n = d->B * d->C * FIN_SUB;
    hipLaunchKernelGGL(msbn_bwd_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *d, hc_get_stat_replicas());
    return hc_launch_status();
}

int hc_msbn_apply(const hc_msbn_io* io, const float* coef, void* out, float* out_stats, int32_t act, hc_stream_t stream) {
    Srcs s;
    if (!fill_srcs(io, s, false) || coef == nullptr || out == nullptr || (act != 0 && act != 1)) return HC_ERR_ARG;
    if (io->npix == 0) return HC_OK;
    const int cg = io->C / 8;
    const int blocks = mb_blocks((long)io->npix * cg, cg, out_stats != nullptr ? 8 : 2);
    hipStream_t st = (hipStream_t)stream;
    const long npix = (long)io->npix;
    const int C = io->C;
    if (out_stats != nullptr) {
        const size_t lds = (size_t)MB_THREADS * 17 * sizeof(float);
        MSBN_DISPATCH(io->B, hipLaunchKernelGGL((msbn_apply_kernel<BB, true>), dim3(blocks), dim3(MB_THREADS), lds, st, s, coef, (u32x4*)out,
                                                out_stats, npix, C, act, hc_get_stat_replicas()));
    } else {
        MSBN_DISPATCH(io->B, hipLaunchKernelGGL((msbn_apply_kernel<BB, false>), dim3(blocks), dim3(MB_THREADS), 0, st, s, coef, (u32x4*)out,
                                                out_stats, npix, C, act, hc_get_stat_replicas()));
    }
    return hc_launch_status();
}

int hc_msbn_bwd_reduce(const hc_msbn_io* io, const void* g, int32_t g_ld, const void* out, float* red, int32_t act,
                       hc_stream_t stream) {
    Srcs s;
    if (!fill_srcs(io, s, false) || g == nullptr || red == nullptr || g_ld < io->C || (g_ld % 8) != 0 || (act != 0 && act != 1) ||
        (act == 1 && out == nullptr))
        return HC_ERR_ARG;
    if (io->npix == 0) return HC_OK;
    const int cg = io->C / 8;
    const int blocks = mb_blocks((long)io->npix * cg, cg, 16);
    hipStream_t st = (hipStream_t)stream;
    const long npix = (long)io->npix;
    const int C = io->C;
    MSBN_DISPATCH(io->B, hipLaunchKernelGGL((msbn_bwd_reduce_kernel<BB>), dim3(blocks), dim3(MB_THREADS),
                                            (size_t)MB_THREADS * 25 * sizeof(float), st, s, (const u32x4*)g, g_ld / 8,
                                            (const u32x4*)out, red, npix, C, act, hc_get_stat_replicas()));
    return hc_launch_status();
}

int hc_msbn_bwd_apply(const hc_msbn_io* io, const void* g, int32_t g_ld, const void* out, const float* bcoef, int32_t act,
                      hc_stream_t stream) {
    Srcs s;
    if (!fill_srcs(io, s, true) || g == nullptr || bcoef == nullptr || g_ld < io->C || (g_ld % 8) != 0 || (act != 0 && act != 1) ||
        (act == 1 && out == nullptr))
        return HC_ERR_ARG;
    if (io->npix == 0) return HC_OK;
    const int cg = io->C / 8;
    const int blocks = mb_blocks((long)io->npix * cg, cg, 2);
    hipStream_t st = (hipStream_t)stream;
    const long npix = (long)io->npix;
    const int C = io->C;
    MSBN_DISPATCH(io->B, hipLaunchKernelGGL((msbn_bwd_apply_kernel<BB>), dim3(blocks), dim3(MB_THREADS), 0, st, s, (const u32x4*)g,
                                            g_ld / 8, (const u32x4*)out, bcoef, npix, C, act));
    return hc_launch_status();
}

int hc_dwrep_dgrad(const void* const* dy, const float* const* wpk, int32_t nplanes, const void* extra, void* dx, int32_t N, int32_t H,
                   int32_t W, int32_t C, int32_t stride, hc_stream_t stream) {
    if (dy == nullptr || wpk == nullptr || dx == nullptr || nplanes < 1 || nplanes > MAXB || C <= 0 || (C % 8) != 0 ||
        (stride != 1 && stride != 2) || (stride == 2 && extra != nullptr) || (long)N * H * W >= (1L << 31))
        return HC_ERR_ARG;
    Planes pl;
    for (int b = 0; b < MAXB; ++b) { pl.dy[b] = nullptr; pl.w[b] = nullptr; }
    for (int b = 0; b < nplanes; ++b) {
        if (dy[b] == nullptr || wpk[b] == nullptr) return HC_ERR_ARG;
        pl.dy[b] = (const u32x4*)dy[b];
        pl.w[b] = wpk[b];
    }
    if ((long)N * H * W == 0) return HC_OK;
    const int cg = C / 8;
    hipStream_t st = (hipStream_t)stream;
    if (stride == 1) {
        constexpr int TW = 4;
        const long items = (long)N * H * ((W + TW - 1) / TW) * cg;
        hipLaunchKernelGGL((dwrep_dgrad_s1_kernel<TW>), dim3(mb_blocks(items, cg, 2)), dim3(MB_THREADS), 0, st, pl, nplanes,
                           (const u32x4*)extra, (u32x4*)dx, N, H, W, C);
    } else {
        const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
        hipLaunchKernelGGL(dwrep_dgrad_s2_kernel, dim3(mb_blocks((long)N * H * W * cg, cg, 4)), dim3(MB_THREADS), 0, st, pl, nplanes,
                           (u32x4*)dx, N, H, W, OH, OW, C);
    }
    return hc_launch_status();
}

}  // extern "C"
