// Data-movement passes of the CSP / PAN / SPP stacks on NHWC bf16 activations (reference:
// holocron/models/classification/darknetv4.py:112-115 chunk+cat, holocron/models/detection/yolov4.py:134-139
// upsample+cat, holocron/nn/modules/downsample.py:154-167 SPP).  All are pure HBM traffic: a thread moves
// 16-byte chunks (8 channels); source and destination carry their own channels-per-pixel ("ld") and
// channel offset so that a concat is written in place by its producers instead of by an extra pass.
#include <cstdlib>
#include "common.h"
#include "../../include/holocron_hip.h"

namespace {

inline int grid_for(long total, int threads = 256, int cap = 16384) {
    long b = (total + threads - 1) / threads;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
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

// dst[p][dc0 + c] = src[p][sc0 + c], c < C ; everything in units of 8 channels
// IDX = unsigned when every index of the launch fits 32 bits (the launcher checks): the 64-bit division per 16-byte element was a
// hundred VALU instructions for one load and one store
template <typename IDX>
__global__ void nhwc_copy_kernel(const u32x4* __restrict__ src, int sld8, int sc8, u32x4* __restrict__ dst, int dld8, int dc8,
                                 long npix, int c8) {
    const IDX total = (IDX)npix * (IDX)c8;
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
        const IDX p = i / (IDX)c8;
        const IDX c = i - p * (IDX)c8;
        dst[p * (IDX)dld8 + (IDX)dc8 + c] = src[p * (IDX)sld8 + (IDX)sc8 + c];
    }
}

// nearest x2: dst[n][2h+a][2w+b] = src[n][h][w]
__global__ void upsample2x_fwd_kernel(const u32x4* __restrict__ src, int sld8, int sc8, u32x4* __restrict__ dst, int dld8, int dc8,
                                      int N, int H, int W, int c8) {
    const long total = (long)N * 2 * H * 2 * W * c8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / c8;
        const int c = (int)(i - p * c8);
        const int ow = (int)(p % (2 * W));
        const int oh = (int)((p / (2 * W)) % (2 * H));
        const long n = p / ((long)4 * W * H);
        const long sp = (n * H + (oh >> 1)) * W + (ow >> 1);
        dst[p * dld8 + dc8 + c] = src[sp * sld8 + sc8 + c];
    }
}
// dsrc[n][h][w] = sum of the four dst gradients (fp32 accumulation, one bf16 rounding)
__global__ void upsample2x_bwd_kernel(const u32x4* __restrict__ g, int gld8, int gc8, u32x4* __restrict__ dx, int xld8, int xc8,
                                      int N, int H, int W, int c8) {
    const long total = (long)N * H * W * c8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / c8;
        const int c = (int)(i - p * c8);
        const int w = (int)(p % W);
        const int h = (int)((p / W) % H);
        const long n = p / ((long)W * H);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const long gp = (n * 2 * H + 2 * h + a) * (2 * W) + 2 * w + b;
                float f[8];
                unpack8(g[gp * gld8 + gc8 + c], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += f[e];
            }
        dx[p * xld8 + xc8 + c] = pack8(acc);
    }
}

// SPP with windows 5/9/13 (stride 1, pad k/2, -inf padding): out[p] = [x | max5 | max9 | max13] and the argmax of
// each window as a byte (dy+6)*13 + (dx+6).  Ties keep the FIRST maximum in row-major window order, the rule of
// torch's max_pool2d, so that the gradient lands on the same element.
__global__ __launch_bounds__(256) void spp_fwd_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ out, u32x2* __restrict__ idx,
                                                      int N, int H, int W, int c8) {
    const long total = (long)N * H * W * c8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / c8;
        const int c = (int)(i - p * c8);
        const int w = (int)(p % W);
        const int h = (int)((p / W) % H);
        const long n = p / ((long)W * H);
        float m[3][8];
        unsigned int am[3][8];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) { m[k][e] = -INFINITY; am[k][e] = 6 * 13 + 6; }
        for (int dy = -6; dy <= 6; ++dy) {
            const int yy = h + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -6; dx <= 6; ++dx) {
                const int xx = w + dx;
                if (xx < 0 || xx >= W) continue;
                float f[8];
                unpack8(x[((n * H + yy) * W + xx) * c8 + c], f);
                const unsigned int code = (unsigned int)((dy + 6) * 13 + (dx + 6));
                const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
                const int r = ady > adx ? ady : adx;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (f[e] > m[2][e]) { m[2][e] = f[e]; am[2][e] = code; }
                    if (r <= 4 && f[e] > m[1][e]) { m[1][e] = f[e]; am[1][e] = code; }
                    if (r <= 2 && f[e] > m[0][e]) { m[0][e] = f[e]; am[0][e] = code; }
                }
            }
        }
        const long ob = p * 4 * c8 + c;
        out[ob] = x[p * c8 + c];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            out[ob + (k + 1) * c8] = pack8(m[k]);
            u32x2 pk;
            pk[0] = am[k][0] | (am[k][1] << 8) | (am[k][2] << 16) | (am[k][3] << 24);
            pk[1] = am[k][4] | (am[k][5] << 8) | (am[k][6] << 16) | (am[k][7] << 24);
            idx[((long)k * N * H * W + p) * c8 + c] = pk;
        }
    }
}
// dx[p] = g[p][x part] + sum over the windows containing p of g[q][pool k] where argmax_k(q) == p
__global__ __launch_bounds__(256) void spp_bwd_kernel(const u32x4* __restrict__ g, const u32x2* __restrict__ idx, u32x4* __restrict__ dx,
                                                      int N, int H, int W, int c8) {
    const long total = (long)N * H * W * c8;
    const long npix = (long)N * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / c8;
        const int c = (int)(i - p * c8);
        const int w = (int)(p % W);
        const int h = (int)((p / W) % H);
        const long n = p / ((long)W * H);
        float acc[8];
        unpack8(g[p * 4 * c8 + c], acc);
        for (int dy = -6; dy <= 6; ++dy) {
            const int qy = h + dy;   // output position q whose window may contain p
            if (qy < 0 || qy >= H) continue;
            for (int dx_ = -6; dx_ <= 6; ++dx_) {
                const int qx = w + dx_;
                if (qx < 0 || qx >= W) continue;
                const long q = (n * H + qy) * W + qx;
                // p seen from q has offset (-dy, -dx_)
                const unsigned int code = (unsigned int)((6 - dy) * 13 + (6 - dx_));
                const int ady = dy < 0 ? -dy : dy, adx = dx_ < 0 ? -dx_ : dx_;
                const int r = ady > adx ? ady : adx;
                const int k0 = r <= 2 ? 0 : (r <= 4 ? 1 : 2);
                for (int k = k0; k < 3; ++k) {
                    const u32x2 pk = idx[((long)k * npix + q) * c8 + c];
                    // any byte equal to `code`?  The exact zero-byte test on pk ^ code-in-every-byte: ten operations for the 99 % of
                    // (q, k) that miss instead of eight shift-mask-compare triples (the walk is VALU-bound: 231 us per launch before)
                    const unsigned int x0 = pk[0] ^ (code * 0x01010101u), x1 = pk[1] ^ (code * 0x01010101u);
                    if ((((x0 - 0x01010101u) & ~x0) | ((x1 - 0x01010101u) & ~x1)) & 0x80808080u) {} else continue;
                    bool hit[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) hit[e] = ((pk[e >> 2] >> (8 * (e & 3))) & 0xffu) == code;
                    float f[8];
                    unpack8(g[q * 4 * c8 + (k + 1) * c8 + c], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += hit[e] ? f[e] : 0.f;
                }
            }
        }
        dx[p * c8 + c] = pack8(acc);
    }
}

// ---------------------------------------------------------------- SPP forward, LDS-tiled with integer keys (round 6)
// Both window walks are VALU-bound, not memory-bound: 169 positions x 8 channels x (compare + two selects) x 1-3 pools is ~8 k lane
// operations per (pixel, channel group), 3 G per launch of YOLOv4's 16 x 512 x 19 x 19 map = 86 us of the chip's 64 lanes per CU and
// clock however the operands arrive (142 us measured from global memory, 162 us from an LDS tile with the same loop).  This kernel
// changes the arithmetic: a workgroup stages one image x SPP_G channel groups in LDS as 32-bit KEYS, (order-preserving map of the
// bf16 value) << 16, with a 6-pixel halo of the smallest key; in the window loop a position costs one OR (key | 255 - code) and one
// integer MAX per pool: the largest key is the largest value and, among equal values, the smallest code = the first maximum in
// row-major window order, torch's rule.  (-0 is keyed as +0: they compare equal in float, and a tie between them must go to the
// first, not to +0; the stored maximum is then +0 where the walk may store -0 - equal as numbers.  NaNs sort above +inf.)
constexpr int SPP_G = 2;
__device__ __forceinline__ unsigned int spp_key(unsigned int b) {        // b: bf16 bits; monotone in the value
    if ((b & 0x7fffu) == 0u) b = 0u;
    return ((b & 0x8000u) ? (~b & 0xffffu) : (b | 0x8000u)) << 16;
}
__device__ __forceinline__ unsigned int spp_unkey(unsigned int k) {      // back to bf16 bits
    const unsigned int b = k >> 16;
    return (b & 0x8000u) ? (b & 0x7fffu) : (~b & 0xffffu);
}
__global__ __launch_bounds__(256) void spp_fwd_tile_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ out, u32x2* __restrict__ idx,
                                                           int N, int H, int W, int c8) {
    extern __shared__ __attribute__((aligned(16))) char spp_smem[];
    u32x4* tile = reinterpret_cast<u32x4*>(spp_smem);          // [H + 12][W + 12][SPP_G][2]: eight keys per (pixel, channel group)
    const int TW = W + 12, TH = H + 12;
    const int n = blockIdx.y, cg0 = blockIdx.x * SPP_G;
    const int ng = (c8 - cg0) < SPP_G ? (c8 - cg0) : SPP_G;
    const int tid = threadIdx.x;
    const unsigned int kmin = spp_key(0xff80u);                 // -inf
    for (int i = tid; i < TH * TW * SPP_G; i += 256) {
        const int g = i % SPP_G, t = i / SPP_G, tx = t % TW, ty = t / TW;
        const int yy = ty - 6, xx = tx - 6;
        u32x4 lo = {kmin, kmin, kmin, kmin}, hi = lo;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W && g < ng) {
            const u32x4 v = x[((long)(n * H + yy) * W + xx) * c8 + cg0 + g];
            lo = u32x4{spp_key(v[0] & 0xffffu), spp_key(v[0] >> 16), spp_key(v[1] & 0xffffu), spp_key(v[1] >> 16)};
            hi = u32x4{spp_key(v[2] & 0xffffu), spp_key(v[2] >> 16), spp_key(v[3] & 0xffffu), spp_key(v[3] >> 16)};
        }
        tile[2 * i] = lo;
        tile[2 * i + 1] = hi;
    }
    __syncthreads();
    const long npix = (long)N * H * W;
    for (int i = tid; i < H * W * SPP_G; i += 256) {
        const int g = i % SPP_G, pp = i / SPP_G, w = pp % W, h = pp / W;
        if (g >= ng) continue;
        unsigned int m[3][8];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) m[k][e] = 0u;
        const u32x4* ctr = tile + (((h + 6) * TW + (w + 6)) * SPP_G + g) * 2;
#pragma unroll
        for (int dy = -6; dy <= 6; ++dy) {
            const u32x4* row = ctr + dy * TW * SPP_G * 2;
#pragma unroll
            for (int dx = -6; dx <= 6; ++dx) {
                const u32x4 lo = row[dx * SPP_G * 2], hi = row[dx * SPP_G * 2 + 1];
                const unsigned int tag = 255u - (unsigned int)((dy + 6) * 13 + (dx + 6));
                const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
                const int r = ady > adx ? ady : adx;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned int key = (e < 4 ? lo[e & 3] : hi[e & 3]) | tag;
                    m[2][e] = key > m[2][e] ? key : m[2][e];
                    if (r <= 4) m[1][e] = key > m[1][e] ? key : m[1][e];
                    if (r <= 2) m[0][e] = key > m[0][e] ? key : m[0][e];
                }
            }
        }
        const long p = (long)(n * H + h) * W + w;
        const int c = cg0 + g;
        const long ob = p * 4 * c8 + c;
        out[ob] = x[p * c8 + c];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = spp_unkey(m[k][2 * q]) | (spp_unkey(m[k][2 * q + 1]) << 16);
            out[ob + (k + 1) * c8] = o;
            u32x2 pk;
            pk[0] = (255u - (m[k][0] & 0xffu)) | ((255u - (m[k][1] & 0xffu)) << 8) | ((255u - (m[k][2] & 0xffu)) << 16) | ((255u - (m[k][3] & 0xffu)) << 24);
            pk[1] = (255u - (m[k][4] & 0xffu)) | ((255u - (m[k][5] & 0xffu)) << 8) | ((255u - (m[k][6] & 0xffu)) << 16) | ((255u - (m[k][7] & 0xffu)) << 24);
            idx[((long)k * npix + p) * c8 + c] = pk;
        }
    }
}
// (The backward walk was built on the same tile too - argmax bytes and the pooled gradient rows in LDS, the exact any-byte-equal test in
// front of the per-channel one, a row of codes fetched ahead of its tests: 279-327 us per launch against 237 us for spp_bwd_kernel, whose
// 1 444 small workgroups hide the walk's dependent chains better than one 512-thread workgroup per CU does.  Not kept.)
// backward as a SCATTER (not in deterministic mode): a workgroup owns one image x SPP_G channel groups and an fp32 tile of their dx in LDS;
// every output position q adds its three pooled gradients to the pixels its argmax bytes name (24 `ds_add_f32` per (q, channel group)
// instead of a 275-position search per pixel), then the tile plus the pass-through part of g is written out.  The order of the fp32
// additions into one pixel is not fixed (sums of at most 275 terms, rounded to bf16 afterwards); hc_set_deterministic(1) keeps the walk.
__global__ __launch_bounds__(256) void spp_bwd_scatter_kernel(const u32x4* __restrict__ g, const u32x2* __restrict__ idx, u32x4* __restrict__ dx,
                                                              int N, int H, int W, int c8) {
    extern __shared__ __attribute__((aligned(16))) char spp_smem[];
    float* acc = reinterpret_cast<float*>(spp_smem);           // [H * W][SPP_G][8]
    const int n = blockIdx.y, cg0 = blockIdx.x * SPP_G;
    const int ng = (c8 - cg0) < SPP_G ? (c8 - cg0) : SPP_G;
    const int tid = threadIdx.x, HW = H * W;
    const long npix = (long)N * HW;
    for (int i = tid; i < HW * SPP_G * 8; i += 256) acc[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < HW * SPP_G * 3; i += 256) {
        const int k = i % 3, j = i / 3, gi = j % SPP_G, qq = j / SPP_G;
        if (gi >= ng) continue;
        const int qx = qq % W, qy = qq / W;
        const long q = (long)n * HW + qq;
        const u32x2 pk = idx[((long)k * npix + q) * c8 + cg0 + gi];
        float f[8];
        unpack8(g[q * 4 * c8 + (k + 1) * c8 + cg0 + gi], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int code = (int)((pk[e >> 2] >> (8 * (e & 3))) & 0xffu);
            const int py = qy + code / 13 - 6, px = qx + code % 13 - 6;      // always inside the map (the forward never picks padding)
            atomicAdd(acc + ((py * W + px) * SPP_G + gi) * 8 + e, f[e]);
        }
    }
    __syncthreads();
    for (int i = tid; i < HW * SPP_G; i += 256) {
        const int gi = i % SPP_G, pp = i / SPP_G;
        if (gi >= ng) continue;
        const long p = (long)n * HW + pp;
        float a[8];
        unpack8(g[p * 4 * c8 + cg0 + gi], a);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += acc[(pp * SPP_G + gi) * 8 + e];
        dx[p * c8 + cg0 + gi] = pack8(a);
    }
}
static bool spp_scatter_ok(int H, int W) {
    const char* e = getenv("HC_SPP_TILE");
    const int on = e == nullptr ? 1 : atoi(e);
    return on && !hc_get_deterministic() && (long)H * W * SPP_G * 32 <= 96 * 1024;
}
static bool spp_tile_ok(int H, int W) {
    const char* e = getenv("HC_SPP_TILE");           // read per call: the test flips it inside one process
    const int on = e == nullptr ? 1 : atoi(e);
    return on && (long)(H + 12) * (W + 12) * SPP_G * 32 <= 96 * 1024;
}

// ---------------------------------------------------------------- SlimConv2d gate + fold (nn/modules/conv.py:352-364)
// x NHWC bf16 [N][HW][ld] with C real channels, l bf16 [N][lld] gate logits, s = sigmoid(l), h = C / 2:
//   top[j] = x[j] s[j] + x[j+h] s[j+h] ;  bot[j] = x[j] s[C-1-j] + x[j+h] s[C-1-j-h]      (w.flip(dims=(1,)))
// outputs NHWC bf16 with old channels per pixel (pad channels are written as zeros).  Scalar 2-byte accesses: the
// fold pairs channel j with j + C/2, which is not 16-byte friendly for the small widths this layer is used at.
__global__ void slim_fold_fwd_kernel(const bf16_t* __restrict__ x, int ld, const bf16_t* __restrict__ l, int lld,
                                     bf16_t* __restrict__ top, bf16_t* __restrict__ bot, int old, long N, long HW, int C) {
    const int h = C / 2;
    const long total = N * HW * old;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % old);
        const long p = i / old;
        const long n = p / HW;
        float t = 0.f, b = 0.f;
        if (j < h) {
            const bf16_t* px = x + p * ld;
            const bf16_t* pl = l + n * lld;
            const float x0 = bf16_to_f32(px[j]), x1 = bf16_to_f32(px[j + h]);
            const float s0 = 1.f / (1.f + __expf(-bf16_to_f32(pl[j]))), s1 = 1.f / (1.f + __expf(-bf16_to_f32(pl[j + h])));
            const float f0 = 1.f / (1.f + __expf(-bf16_to_f32(pl[C - 1 - j]))), f1 = 1.f / (1.f + __expf(-bf16_to_f32(pl[C - 1 - j - h])));
            t = x0 * s0 + x1 * s1;
            b = x0 * f0 + x1 * f1;
        }
        top[i] = f32_to_bf16(t);
        bot[i] = f32_to_bf16(b);
    }
}
// ds[n][c] = sum_hw gtop[c mod h] x[c] + gbot[(C-1-c) mod h] x[C-1-c] ; dl = ds * s (1 - s).  one block per (n, c)
__global__ __launch_bounds__(256) void slim_fold_bwd_gate_kernel(const bf16_t* __restrict__ x, int ld, const bf16_t* __restrict__ l,
                                                                 int lld, const bf16_t* __restrict__ gtop,
                                                                 const bf16_t* __restrict__ gbot, int old, bf16_t* __restrict__ dl,
                                                                 long HW, int C) {
    const int c = blockIdx.x;
    const long n = blockIdx.y;
    const int h = C / 2;
    const int cf = C - 1 - c;
    float acc = 0.f;
    for (long q = threadIdx.x; q < HW; q += blockDim.x) {
        const long p = n * HW + q;
        acc += bf16_to_f32(gtop[p * old + (c % h)]) * bf16_to_f32(x[p * ld + c]) +
               bf16_to_f32(gbot[p * old + (cf % h)]) * bf16_to_f32(x[p * ld + cf]);
    }
    acc = wave_sum(acc);
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float s = 1.f / (1.f + __expf(-bf16_to_f32(l[n * lld + c])));
        dl[n * lld + c] = f32_to_bf16((sh[0] + sh[1] + sh[2] + sh[3]) * s * (1.f - s));
    }
}
// dx[c] = gtop[c mod h] s[c] + gbot[c mod h] s[C-1-c] + dpool[n][c] / HW   (pad channels: 0)
__global__ void slim_fold_bwd_apply_kernel(const bf16_t* __restrict__ l, int lld, const bf16_t* __restrict__ gtop,
                                           const bf16_t* __restrict__ gbot, int old, const float* __restrict__ dpool, int dld,
                                           bf16_t* __restrict__ dx, int ld, long N, long HW, int C) {
    const int h = C / 2;
    const long total = N * HW * ld;
    const float inv = 1.f / (float)HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ld);
        const long p = i / ld;
        const long n = p / HW;
        float v = 0.f;
        if (c < C) {
            const bf16_t* pl = l + n * lld;
            const float s = 1.f / (1.f + __expf(-bf16_to_f32(pl[c]))), f = 1.f / (1.f + __expf(-bf16_to_f32(pl[C - 1 - c])));
            v = bf16_to_f32(gtop[p * old + (c % h)]) * s + bf16_to_f32(gbot[p * old + (c % h)]) * f + dpool[n * dld + c] * inv;
        }
        dx[i] = f32_to_bf16(v);
    }
}

// ---------------------------------------------------------------- fp8 (OCP e4m3) helpers of the inference path
__global__ void quantize_fp8_kernel(const bf16_t* __restrict__ src, int sld, unsigned char* __restrict__ dst, int dld, long npix, int C,
                                    float inv_scale) {
    const int q4 = dld / 4;   // 4 output bytes per thread
    const long total = npix * q4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / q4;
        const int c = (int)(i - p * q4) * 4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = (c + e < C) ? bf16_to_f32(src[p * sld + c + e]) * inv_scale : 0.f;
            v[e] = fminf(fmaxf(t, -448.f), 448.f);
        }
        int pk = 0;
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], pk, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
        *reinterpret_cast<int*>(dst + p * dld + c) = pk;
    }
}
__device__ __forceinline__ float fp8_e4m3_to_f32(unsigned int b) {   // OCP e4m3fn byte -> fp32 (no NaN handling needed here)
    const unsigned int s = (b & 0x80u) << 24, e = (b >> 3) & 15u, m = b & 7u;
    if (e == 0) return __builtin_bit_cast(float, s) + (s ? -1.f : 1.f) * (float)m * 0.001953125f;   // m * 2^-9
    return __builtin_bit_cast(float, s | ((e + 120u) << 23) | (m << 20));
}
__global__ __launch_bounds__(256) void gap_fp8_kernel(const unsigned char* __restrict__ x, float* __restrict__ y, int HW, int ld, int C,
                                                      float scale) {
    const long n = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const unsigned char* p = x + n * HW * ld + c;
    float s = 0.f;
    for (int h = 0; h < HW; ++h) s += fp8_e4m3_to_f32(p[(long)h * ld]);
    y[n * C + c] = s * scale / (float)HW;
}

// ---------------------------------------------------------------- nn.MaxPool2d(2) of the DarkNet-19 / 24 bodies (darknet.py:83, darknetv2.py:94)
// out[n][oh][ow] = max over the 2 x 2 window (floor mode: a trailing odd row / column is dropped); idx: 2 bits per channel
// (8 channels -> one uint16 per 16-byte chunk), the first maximum in row-major order like torch's max_pool2d
__global__ void maxpool2_fwd_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ out, unsigned short* __restrict__ idx, int N, int H,
                                    int W, int OH, int OW, int c8) {
    const long total = (long)N * OH * OW * c8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / c8;
        const int c = (int)(i - p * c8);
        const int ow = (int)(p % OW);
        const int oh = (int)((p / OW) % OH);
        const long n = p / ((long)OW * OH);
        float best[8];
        unsigned code = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long sp = (n * H + 2 * oh + (k >> 1)) * W + 2 * ow + (k & 1);
            const u32x4 v = x[sp * c8 + c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = bf16lo(v[e]), hi = bf16hi(v[e]);
                if (k == 0 || lo > best[2 * e]) { best[2 * e] = lo; code = (code & ~(3u << (4 * e))) | ((unsigned)k << (4 * e)); }
                if (k == 0 || hi > best[2 * e + 1]) { best[2 * e + 1] = hi; code = (code & ~(3u << (4 * e + 2))) | ((unsigned)k << (4 * e + 2)); }
            }
        }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(best[2 * e], best[2 * e + 1]);
        out[i] = o;
        idx[i] = (unsigned short)code;
    }
}
__global__ void maxpool2_bwd_kernel(const u32x4* __restrict__ g, const unsigned short* __restrict__ idx, u32x4* __restrict__ dx, int N, int H,
                                    int W, int OH, int OW, int c8) {
    const long total = (long)N * H * W * c8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / c8;
        const int c = (int)(i - p * c8);
        const int w = (int)(p % W);
        const int h = (int)((p / W) % H);
        const long n = p / ((long)W * H);
        u32x4 o = {0u, 0u, 0u, 0u};
        if ((h >> 1) < OH && (w >> 1) < OW) {
            const long q = ((n * OH + (h >> 1)) * OW + (w >> 1)) * c8 + c;
            const u32x4 gv = g[q];
            const unsigned code = idx[q], k = (unsigned)((h & 1) * 2 + (w & 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned lo = ((code >> (4 * e)) & 3u) == k ? (gv[e] & 0xffffu) : 0u;
                const unsigned hi = ((code >> (4 * e + 2)) & 3u) == k ? (gv[e] & 0xffff0000u) : 0u;
                o[e] = lo | hi;
            }
        }
        dx[i] = o;
    }
}

// ---------------------------------------------------------------- concat_downsample2d (holocron/nn/functional.py:116-136)
// dst[n][oh][ow][dc0 + (a*s + b)*C + c] = src[n][oh*s + a][ow*s + b][c]   (and its gradient: the inverse scatter)
template <bool BWD>
__global__ void space_to_depth_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int dld8, int dc8, int N, int OH, int OW, int c8,
                                      int s) {
    const long total = (long)N * OH * OW * s * s * c8;
    const int H = OH * s, W = OW * s;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c8);
        long r = i / c8;
        const int ab = (int)(r % (s * s));
        r /= s * s;
        const int ow = (int)(r % OW);
        const int oh = (int)((r / OW) % OH);
        const long n = r / ((long)OW * OH);
        const long big = ((n * H + oh * s + ab / s) * W + ow * s + ab % s) * c8 + c;               // dense [N][H][W][C]
        const long small = ((n * OH + oh) * OW + ow) * (long)dld8 + dc8 + (long)ab * c8 + c;         // [N][OH][OW][ld]
        if (BWD) dst[big] = src[small];
        else dst[small] = src[big];
    }
}

// dy = g * (out > 0 ? 1 : slope): gradient through ReLU / LeakyReLU from the stored OUTPUT (same sign as the pre-activation)
__global__ void leaky_bwd_kernel(const u32x4* __restrict__ g, int gld8, const u32x4* __restrict__ out, u32x4* __restrict__ dy, long npix,
                                 int c8, float slope) {
    const long total = npix * c8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / c8;
        const int c = (int)(i - p * c8);
        const u32x4 gv = g[p * gld8 + c], ov = out[i];
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = bf16lo(gv[e]) * (bf16lo(ov[e]) > 0.f ? 1.f : slope);
            const float b = bf16hi(gv[e]) * (bf16hi(ov[e]) > 0.f ? 1.f : slope);
            o[e] = pack_bf16x2(a, b);
        }
        dy[i] = o;
    }
}

}  // namespace

extern "C" {

int hc_nhwc_copy(const void* src, int32_t src_ld, int32_t src_c0, void* dst, int32_t dst_ld, int32_t dst_c0, int64_t npix, int32_t C,
                 hc_stream_t stream) {
    if (src == nullptr || dst == nullptr || npix < 0) return HC_ERR_ARG;
    if ((src_ld | src_c0 | dst_ld | dst_c0 | C) & 7) return HC_ERR_ARG;
    if (C <= 0 || src_c0 + C > src_ld || dst_c0 + C > dst_ld) return HC_ERR_ARG;
    if (npix == 0) return HC_OK;
    // 32-bit indices when the element count (plus one grid stride) and both buffers' extents fit
    const long ext = (long)npix * ((src_ld > dst_ld ? src_ld : dst_ld) / 8) + 16384L * 256;
    if (ext < 4294967295L)
        hipLaunchKernelGGL(nhwc_copy_kernel<unsigned>, dim3(grid_for(npix * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src,
                           src_ld / 8, src_c0 / 8, (u32x4*)dst, dst_ld / 8, dst_c0 / 8, (long)npix, C / 8);
    else
        hipLaunchKernelGGL(nhwc_copy_kernel<long>, dim3(grid_for(npix * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src,
                           src_ld / 8, src_c0 / 8, (u32x4*)dst, dst_ld / 8, dst_c0 / 8, (long)npix, C / 8);
    return hc_launch_status();
}

int hc_upsample2x_fwd(const void* src, int32_t src_ld, int32_t src_c0, void* dst, int32_t dst_ld, int32_t dst_c0, int32_t N, int32_t H,
                      int32_t W, int32_t C, hc_stream_t stream) {
    if (src == nullptr || dst == nullptr) return HC_ERR_ARG;
    if ((src_ld | src_c0 | dst_ld | dst_c0 | C) & 7) return HC_ERR_ARG;
    if (C <= 0 || src_c0 + C > src_ld || dst_c0 + C > dst_ld) return HC_ERR_ARG;
    if ((long)N * H * W == 0) return HC_OK;
    hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(grid_for((long)N * 4 * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const u32x4*)src, src_ld / 8, src_c0 / 8, (u32x4*)dst, dst_ld / 8, dst_c0 / 8, N, H, W, C / 8);
    return hc_launch_status();
}
int hc_upsample2x_bwd(const void* g, int32_t g_ld, int32_t g_c0, void* dx, int32_t dx_ld, int32_t dx_c0, int32_t N, int32_t H, int32_t W,
                      int32_t C, hc_stream_t stream) {
    if (g == nullptr || dx == nullptr) return HC_ERR_ARG;
    if ((g_ld | g_c0 | dx_ld | dx_c0 | C) & 7) return HC_ERR_ARG;
    if (C <= 0 || g_c0 + C > g_ld || dx_c0 + C > dx_ld) return HC_ERR_ARG;
    if ((long)N * H * W == 0) return HC_OK;
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for((long)N * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const u32x4*)g, g_ld / 8, g_c0 / 8, (u32x4*)dx, dx_ld / 8, dx_c0 / 8, N, H, W, C / 8);
    return hc_launch_status();
}

int hc_spp_fwd(const void* x, void* out, void* idx, int32_t N, int32_t H, int32_t W, int32_t C, hc_stream_t stream) {
    if (x == nullptr || out == nullptr || idx == nullptr || C <= 0 || (C & 7)) return HC_ERR_ARG;
    if ((long)N * H * W == 0) return HC_OK;
    if (spp_tile_ok(H, W) && N <= 65535) {
        const int smem = (H + 12) * (W + 12) * SPP_G * 32;
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spp_fwd_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            attr = true;
        }
        hipLaunchKernelGGL(spp_fwd_tile_kernel, dim3((C / 8 + SPP_G - 1) / SPP_G, N), dim3(256), smem, (hipStream_t)stream, (const u32x4*)x,
                           (u32x4*)out, (u32x2*)idx, N, H, W, C / 8);
        return hc_launch_status();
    }
    hipLaunchKernelGGL(spp_fwd_kernel, dim3(grid_for((long)N * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const u32x4*)x, (u32x4*)out, (u32x2*)idx, N, H, W, C / 8);
    return hc_launch_status();
}
int hc_spp_bwd(const void* g, const void* idx, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, hc_stream_t stream) {
    if (g == nullptr || dx == nullptr || idx == nullptr || C <= 0 || (C & 7)) return HC_ERR_ARG;
    if ((long)N * H * W == 0) return HC_OK;
    if (spp_scatter_ok(H, W) && N <= 65535) {
        const int smem = H * W * SPP_G * 32;
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spp_bwd_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            attr = true;
        }
        hipLaunchKernelGGL(spp_bwd_scatter_kernel, dim3((C / 8 + SPP_G - 1) / SPP_G, N), dim3(256), smem, (hipStream_t)stream, (const u32x4*)g,
                           (const u32x2*)idx, (u32x4*)dx, N, H, W, C / 8);
        return hc_launch_status();
    }
    hipLaunchKernelGGL(spp_bwd_kernel, dim3(grid_for((long)N * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const u32x4*)g, (const u32x2*)idx, (u32x4*)dx, N, H, W, C / 8);
    return hc_launch_status();
}

int hc_slim_fold_fwd(const void* x, int32_t x_ld, const void* gate_logits, int32_t l_ld, void* top, void* bot, int32_t out_ld, int64_t N,
                     int64_t HW, int32_t C, hc_stream_t stream) {
    if (x == nullptr || gate_logits == nullptr || top == nullptr || bot == nullptr || C < 2 || (C & 1) || x_ld < C || l_ld < C ||
        out_ld < C / 2)
        return HC_ERR_ARG;
    const long total = (long)N * HW * out_ld;
    if (total == 0) return HC_OK;
    hipLaunchKernelGGL(slim_fold_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, x_ld,
                       (const bf16_t*)gate_logits, l_ld, (bf16_t*)top, (bf16_t*)bot, out_ld, (long)N, (long)HW, C);
    return hc_launch_status();
}
int hc_slim_fold_bwd_gate(const void* x, int32_t x_ld, const void* gate_logits, int32_t l_ld, const void* gtop, const void* gbot,
                          int32_t out_ld, void* dlogits, int64_t N, int64_t HW, int32_t C, hc_stream_t stream) {
    if (x == nullptr || gate_logits == nullptr || gtop == nullptr || gbot == nullptr || dlogits == nullptr || C < 2 || (C & 1) ||
        x_ld < C || l_ld < C || out_ld < C / 2 || N > 65535)
        return HC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if ((long)N * l_ld == 0) return HC_OK;
    if (hc_zero_async(dlogits, (size_t)N * l_ld * 2, st) != hipSuccess) return HC_ERR_LAUNCH;
    hipLaunchKernelGGL(slim_fold_bwd_gate_kernel, dim3(C, (unsigned)N), dim3(256), 0, st, (const bf16_t*)x, x_ld,
                       (const bf16_t*)gate_logits, l_ld, (const bf16_t*)gtop, (const bf16_t*)gbot, out_ld, (bf16_t*)dlogits, (long)HW, C);
    return hc_launch_status();
}
int hc_slim_fold_bwd_apply(const void* gate_logits, int32_t l_ld, const void* gtop, const void* gbot, int32_t out_ld, const float* dpool,
                           int32_t dpool_ld, void* dx, int32_t x_ld, int64_t N, int64_t HW, int32_t C, hc_stream_t stream) {
    if (gate_logits == nullptr || gtop == nullptr || gbot == nullptr || dpool == nullptr || dx == nullptr || C < 2 || (C & 1) ||
        x_ld < C || l_ld < C || out_ld < C / 2 || dpool_ld < C)
        return HC_ERR_ARG;
    const long total = (long)N * HW * x_ld;
    if (total == 0) return HC_OK;
    hipLaunchKernelGGL(slim_fold_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)gate_logits, l_ld, (const bf16_t*)gtop, (const bf16_t*)gbot, out_ld, dpool, dpool_ld, (bf16_t*)dx,
                       x_ld, (long)N, (long)HW, C);
    return hc_launch_status();
}

int hc_quantize_fp8(const void* src_bf16, int32_t src_ld, void* dst_fp8, int32_t dst_ld, int64_t npix, int32_t C, float inv_scale,
                    hc_stream_t stream) {
    if (src_bf16 == nullptr || dst_fp8 == nullptr || C <= 0 || src_ld < C || dst_ld < C || (dst_ld % 4) != 0) return HC_ERR_ARG;
    if (npix == 0) return HC_OK;
    hipLaunchKernelGGL(quantize_fp8_kernel, dim3(grid_for((long)npix * (dst_ld / 4))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src_bf16, src_ld, (unsigned char*)dst_fp8, dst_ld, (long)npix, C, inv_scale);
    return hc_launch_status();
}
int hc_gap_fp8(const void* x_fp8, float* y, int32_t N, int32_t HW, int32_t ld, int32_t C, float scale, hc_stream_t stream) {
    if (x_fp8 == nullptr || y == nullptr || C <= 0 || ld < C || N < 0 || N > 65535) return HC_ERR_ARG;
    if (N == 0 || HW == 0) return HC_OK;
    hipLaunchKernelGGL(gap_fp8_kernel, dim3((C + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)x_fp8, y, HW, ld,
                       C, scale);
    return hc_launch_status();
}

int hc_maxpool2_fwd(const void* x, void* out, void* idx, int32_t N, int32_t H, int32_t W, int32_t C, hc_stream_t stream) {
    if (x == nullptr || out == nullptr || idx == nullptr || C <= 0 || (C & 7) || H < 0 || W < 0) return HC_ERR_ARG;
    const int OH = H / 2, OW = W / 2;
    if ((long)N * OH * OW == 0) return HC_OK;
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for((long)N * OH * OW * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x,
                       (u32x4*)out, (unsigned short*)idx, N, H, W, OH, OW, C / 8);
    return hc_launch_status();
}
int hc_maxpool2_bwd(const void* g, const void* idx, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, hc_stream_t stream) {
    if (g == nullptr || dx == nullptr || idx == nullptr || C <= 0 || (C & 7) || H < 0 || W < 0) return HC_ERR_ARG;
    if ((long)N * H * W == 0) return HC_OK;
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for((long)N * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const u32x4*)g,
                       (const unsigned short*)idx, (u32x4*)dx, N, H, W, H / 2, W / 2, C / 8);
    return hc_launch_status();
}
int hc_space_to_depth(const void* src, void* dst, int32_t ld, int32_t c0, int32_t N, int32_t OH, int32_t OW, int32_t C, int32_t scale,
                      int32_t backward, hc_stream_t stream) {
    if (src == nullptr || dst == nullptr || C <= 0 || ((C | ld | c0) & 7) || scale < 1 || c0 + scale * scale * C > ld) return HC_ERR_ARG;
    const long total = (long)N * OH * OW * scale * scale * (C / 8);
    if (total == 0) return HC_OK;
    if (backward)
        hipLaunchKernelGGL((space_to_depth_kernel<true>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst,
                           ld / 8, c0 / 8, N, OH, OW, C / 8, scale);
    else
        hipLaunchKernelGGL((space_to_depth_kernel<false>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst,
                           ld / 8, c0 / 8, N, OH, OW, C / 8, scale);
    return hc_launch_status();
}
int hc_leaky_bwd(const void* g, int32_t g_ld, const void* out, void* dy, int64_t npix, int32_t C, float slope, hc_stream_t stream) {
    if (g == nullptr || out == nullptr || dy == nullptr || C <= 0 || ((C | g_ld) & 7) || g_ld < C || npix < 0) return HC_ERR_ARG;
    if (npix == 0) return HC_OK;
    hipLaunchKernelGGL(leaky_bwd_kernel, dim3(grid_for((long)npix * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const u32x4*)g, g_ld / 8,
                       (const u32x4*)out, (u32x4*)dy, (long)npix, C / 8, slope);
    return hc_launch_status();
}

}  // extern "C"
