// Depthwise 3x3 convolution on NHWC bf16 (reference call sites: ReXBlock's groups=C conv,
// holocron/models/classification/rexnet.py:111-124, and FReLU, holocron/nn/modules/activation.py:58-82).
// 9 MAC per output element: pure HBM traffic, so everything here is about moving each activation once with
// 16-byte accesses.  A thread owns 8 channels (one 16-byte chunk) and walks strips of TW output pixels along W;
// the 3 x (TW*stride + 2) input window of a strip is loaded once and reused by the TW outputs.  The launch keeps a
// thread's channel group fixed, so the 72 weights stay in registers and the BatchNorm statistics (sum, sum of
// squares of the fp32 results, like the MFMA conv epilogue) are register running sums flushed once per workgroup.
#include <cstdlib>
#include "common.h"
#include "../../include/holocron_hip.h"

namespace {

constexpr int DW_THREADS = 256;

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

// blocks such that (blocks * 256) % cg == 0 (a thread's channel group never changes)
inline int dw_blocks(long items, int cg, int per_thread) {
    int a = cg, b = DW_THREADS;
    while (b) { int t = a % b; a = b; b = t; }
    const int unit = cg / a;
    long want = items / ((long)DW_THREADS * per_thread);
    long floor_blocks = items / DW_THREADS;      // mid-size tensors: fill the chip before batching work per thread
    if (floor_blocks > 1024) floor_blocks = 1024;
    if (want < floor_blocks) want = floor_blocks;
    if (want > 4096) want = 4096;
    if (want < 1) want = 1;
    const long k = (want + unit - 1) / unit;
    return (int)(k * unit);
}

// per-workgroup reduction of v[S][8] over the threads that share a channel group, then one atomic per (k, c).  The rows go
// through LDS CH at a time (256 x (8 CH + 1) floats): a 9- or 7-row reduction in one piece would take 58-75 KB of LDS and
// leave two workgroups per CU on these HBM-bound kernels.  lds: DW_THREADS * (8 * CH + 1) floats.
template <int S, int CH = (S < 3 ? S : 3)>
__device__ __forceinline__ void block_reduce_flush(const float (&v)[S][8], int cg, int C, float* __restrict__ dst, float* __restrict__ lds) {
    constexpr int STR = CH * 8 + 1;
    const int tid = threadIdx.x;
    const int blockbase = (int)(((long)blockIdx.x * DW_THREADS) % cg);
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
        for (int o = tid; o < rows * C; o += DW_THREADS) {
            const int k = o / C, c = o - k * C;
            const int g = c >> 3, e = c & 7;
            int first = g - blockbase;
            if (first < 0) first += cg;
            float sum = 0.f;
            for (int t = first; t < DW_THREADS; t += cg) sum += lds[t * STR + k * 8 + e];
            atomicAdd(dst + (size_t)(k0 + k) * C + c, sum);
        }
    }
}

// ---------------------------------------------------------------- forward (and stride-1 data gradient with flipped taps)
// w: fp32 [9][C] (tap-major).  y[n][oh][ow][c] = sum_t w[t][c] * x[n][oh*S + kh - 1][ow*S + kw - 1][c]
template <int STRIDE, int TW>
__global__ __launch_bounds__(DW_THREADS) void dw3x3_fwd_kernel(const u32x4* __restrict__ x, const float* __restrict__ w,
                                                               u32x4* __restrict__ y, float* __restrict__ stats, int N, int H, int W,
                                                               int OH, int OW, int C, const int reps) {
    extern __shared__ float sred[];
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * DW_THREADS + threadIdx.x;
    const long nthreads = (long)gridDim.x * DW_THREADS;
    const int cgi = (int)(gtid % cg);
    const int strips_w = (OW + TW - 1) / TW;
    const long nstrips = (long)N * OH * strips_w;
    float wr[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(w + (size_t)t * C + cgi * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(w + (size_t)t * C + cgi * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { wr[t][e] = a[e]; wr[t][4 + e] = b[e]; }
    }
    float sv[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sv[0][e] = sv[1][e] = 0.f;
    constexpr int IWN = (TW - 1) * STRIDE + 3;   // input columns a strip touches
    for (long s = gtid / cg; s < nstrips; s += nthreads / cg) {
        const unsigned su = (unsigned)s, ru = su / (unsigned)strips_w;   // 32-bit: the launcher bounds the strip count
        const int sw = (int)(su - ru * (unsigned)strips_w);
        const unsigned nu = ru / (unsigned)OH;
        const int oh = (int)(ru - nu * (unsigned)OH);
        const long n = nu;
        const int ow0 = sw * TW;
        float acc[TW][8];
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * STRIDE + kh - 1;
            if (ih < 0 || ih >= H) continue;
            const u32x4* row = x + ((n * H + ih) * (long)W) * cg + cgi;
#pragma unroll
            for (int c = 0; c < IWN; ++c) {
                const int iw = ow0 * STRIDE + c - 1;
                if (iw < 0 || iw >= W) continue;
                float f[8];
                unpack8(row[(long)iw * cg], f);
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int kw = c - j * STRIDE;
                    if (kw < 0 || kw > 2) continue;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[j][e] += wr[kh * 3 + kw][e] * f[e];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int ow = ow0 + j;
            if (ow >= OW) break;
            y[((n * OH + oh) * (long)OW + ow) * cg + cgi] = pack8(acc[j]);
            if (stats != nullptr) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { sv[0][e] += acc[j][e]; sv[1][e] += acc[j][e] * acc[j][e]; }
            }
        }
    }
    if (stats != nullptr) block_reduce_flush<2>(sv, cg, C, stats + (size_t)(blockIdx.x % reps) * 2 * C, sred);
}

// ---------------------------------------------------------------- forward, stride 1, LDS-TILED (round 4)
// The strip kernel above loads the 3 x (TW + 2) window of every TW outputs from global memory: 4.5 16-byte loads per output, two
// thirds of them rows another wave (usually another CU) has just fetched - L2 hits, but each one a texture-path request, and that
// request rate held the kernel at 1.85 TB/s.  Here a workgroup owns a 64-channel slice (blockIdx.y) and walks 8 x 32 output tiles:
// the 10 x 34 input window of a tile goes HBM / L2 -> LDS by DMA once (1.33 fetched bytes per output byte instead of 4.5, zero halo =
// out-of-range offsets), and the 3 x 10 window of a thread's 8-pixel strip is 30 ds_read_b128 of it.  Two workgroups per CU overlap one
// tile's DMA wait with the other's arithmetic.  Same tap order per output as the strip kernel: bit-identical results.
// Tile shapes: 8 rows x 32 columns (NSTRIP = 4) for wide maps, 16 x 16 (NSTRIP = 2) for the 14 x 14 / 16 x 16 maps - 256 threads =
// 8 channel groups x NSTRIP strips of 8 pixels x TH rows either way
// GROUPS_ = channel groups (of 8 channels) per slice: 8 (128 bytes per pixel), or 4 for 32-channel layers (8 x 64 tiles: the same
// 256 threads, no idle lanes)
template <int NSTRIP_, int GROUPS_ = 8>
struct DwTile {
    static constexpr int NSTRIP = NSTRIP_, GROUPS = GROUPS_, SW = 8, TW = SW * NSTRIP, TH = 256 / (GROUPS * NSTRIP);
    static constexpr int PPP = 64 / GROUPS;                // pixels per 1 KB DMA piece
    static constexpr int PITCH_PX = (TW + 2 + PPP - 1) / PPP * PPP, PX_BYTES = 16 * GROUPS, ROW_BYTES = PITCH_PX * PX_BYTES, ROWS = TH + 2;
    static constexpr int WIN_BYTES = ROWS * ROW_BYTES;     // 51 200 B (8 x 32) / 55 296 B (16 x 16) / 51 200 B (8 x 64, 4 groups)
    static constexpr int PIECES_ROW = PITCH_PX / PPP, PIECES = ROWS * PIECES_ROW;
};
using DwThin = DwTile<8, 4>;
// the tiled stride-2 kernels take output maps of at least 12 columns (the round-3 strip kernels below that)
static int dw_s2_on() {
    static const int on = [] { const char* t = getenv("HC_DW_TILE"); return (t != nullptr && atoi(t) == 0) ? 0 : 1; }();
    return on;
}
static int dw_s2_minw() {
    constexpr int v = 12;
    return v;
}
// 4-group slices for layers of 32 channels or fewer (HC_DW_TILE_THIN >= 1, the default).  HC_DW_TILE_THIN=2 also splits 72 .. 96
// channels into 4-group slices: measured SLOWER (96@56 stride 1: 3.62 -> 2.83 TB/s; 96@112 stride 2 no better than the strip kernel) -
// three workgroups then fetch 64-byte thirds of every 192-byte pixel at different times, and every 128-byte line is fetched twice
inline int dw_blocks_exact(long items, int cg, int per_thread) {      // no fill-the-chip floor: `per_thread` items per thread
    int a = cg, b = DW_THREADS;
    while (b) { int t = a % b; a = b; b = t; }
    const int unit = cg / a;
    long want = (items + (long)DW_THREADS * per_thread - 1) / ((long)DW_THREADS * per_thread);
    if (want < 1) want = 1;
    return (int)(((want + unit - 1) / unit) * unit);
}
static int dw_row7() {        // HC_DW_ROW7=n: whole rows of the 5..7-pixel-wide maps, n rows per thread (0: strips of four, A/B)
    static const int on = [] { const char* e = getenv("HC_DW_ROW7"); return e == nullptr ? 2 : atoi(e); }();
    return on;
}
static bool dw_thin(int cg) {
    constexpr int mode = 1;
    return (mode >= 1 && cg <= 4) || (mode >= 2 && cg >= 9 && cg <= 12);
}
template <int NSTRIP_, int GROUPS_ = 8>
__global__ __launch_bounds__(DW_THREADS, 2) void dw3x3_fwd_tile_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                                       u32x4* __restrict__ y, float* __restrict__ stats, int N, int H,
                                                                       int W, int C, const int reps, int tiles_x, int tiles_y) {
    using G = DwTile<NSTRIP_, GROUPS_>;
    constexpr int TH = G::TH, SW = G::SW, NSTRIP = G::NSTRIP, TW = G::TW, ROW_BYTES = G::ROW_BYTES, PX_BYTES = G::PX_BYTES, PIECES = G::PIECES,
                  PIECES_ROW = G::PIECES_ROW, GR = G::GROUPS, PPP = G::PPP;
    (void)TH;
    extern __shared__ __attribute__((aligned(1024))) char dsm[];
    const int cg = C / 8, slice = blockIdx.y;
    const int gs = min(GR, cg - slice * GR);               // channel groups of this slice
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gl = tid & (GR - 1), sidx = (tid / GR) & (NSTRIP - 1), r = tid / (GR * NSTRIP);
    const bool live = gl < gs;
    const int cgi = slice * GR + (live ? gl : 0);
    float wr[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(w + (size_t)t * C + cgi * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(w + (size_t)t * C + cgi * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { wr[t][e] = a[e]; wr[t][4 + e] = b[e]; }
    }
    float sv[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sv[0][e] = sv[1][e] = 0.f;
    u32x4 rs;
    {
        const unsigned long long v = reinterpret_cast<unsigned long long>(x);
        rs[0] = __builtin_amdgcn_readfirstlane((unsigned)v);
        rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
        rs[2] = __builtin_amdgcn_readfirstlane((unsigned)((long)N * H * W * C * 2));
        rs[3] = 0x00020000u;
    }
    const unsigned lds0 = hc_lds_addr(dsm);
    const int ntiles = N * tiles_y * tiles_x;
    // DMA lane constants: lane -> (pixel of the 8-pixel piece, channel group)
    const int dpx = lane / GR, dgr = lane & (GR - 1);
    const unsigned choff = (unsigned)((slice * GR * 8 + dgr * 8) * 2);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / (tiles_y * tiles_x), rem = tile - n * (tiles_y * tiles_x);
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        __syncthreads();                                   // everybody is done reading the previous window
        for (int q = wid; q < PIECES; q += 4) {
            const int wr_ = q / PIECES_ROW, px = (q - wr_ * PIECES_ROW) * PPP + dpx;
            const int iy = oy0 - 1 + wr_, ix = ox0 - 1 + px;
            const bool ok = dgr < gs && px < TW + 2 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const unsigned off = (unsigned)(((n * H + iy) * W + ix) * C * 2) + choff;
            hc_dma16(rs, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(q * 1024)), ok ? off : HC_OOB);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int oy = oy0 + r;
        if (live && oy < H) {
            float acc[SW][8];
#pragma unroll
            for (int j = 0; j < SW; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
            const char* base = dsm + r * ROW_BYTES + (sidx * SW) * PX_BYTES + gl * 16;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                for (int c = 0; c < SW + 2; ++c) {
                    float f[8];
                    unpack8(*reinterpret_cast<const u32x4*>(base + kh * ROW_BYTES + c * PX_BYTES), f);
#pragma unroll
                    for (int j = 0; j < SW; ++j) {
                        const int kw = c - j;
                        if (kw < 0 || kw > 2) continue;
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[j][e] += wr[kh * 3 + kw][e] * f[e];
                    }
                }
            }
            u32x4* orow = y + ((long)(n * H + oy) * W) * cg + cgi;
#pragma unroll
            for (int j = 0; j < SW; ++j) {
                const int ox = ox0 + sidx * SW + j;
                if (ox >= W) break;
                orow[(long)ox * cg] = pack8(acc[j]);
                if (stats != nullptr) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { sv[0][e] += acc[j][e]; sv[1][e] += acc[j][e] * acc[j][e]; }
                }
            }
        }
    }
    if (stats != nullptr) {         // the 32 threads of a channel group are combined in LDS (fixed order), one atomic per (k, channel)
        __syncthreads();
        float* red = reinterpret_cast<float*>(dsm);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[tid * 17 + k * 8 + e] = live ? sv[k][e] : 0.f;
        __syncthreads();
        float* rep = stats + (size_t)((blockIdx.x + blockIdx.y * gridDim.x) % reps) * 2 * C;
        for (int o = tid; o < 2 * GR * 8; o += DW_THREADS) {
            const int k = o / (GR * 8), c = o - k * (GR * 8), g = c >> 3, e = c & 7;
            if (g >= gs) continue;
            float sum = 0.f;
            for (int t = g; t < DW_THREADS; t += GR) sum += red[t * 17 + k * 8 + e];
            atomicAdd(rep + (size_t)k * C + slice * GR * 8 + c, sum);
        }
    }
}

// ---------------------------------------------------------------- stride-2 data gradient
// dx[n][h][w][c] = sum over taps with (h + 1 - kh) and (w + 1 - kw) even of w[kh][kw][c] * dy[n][(h+1-kh)/2][(w+1-kw)/2][c]
__global__ __launch_bounds__(DW_THREADS) void dw3x3_dgrad_s2_kernel(const u32x4* __restrict__ dy, const float* __restrict__ w,
                                                                    u32x4* __restrict__ dx, int N, int H, int W, int OH, int OW, int C) {
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * DW_THREADS + threadIdx.x;
    const long nthreads = (long)gridDim.x * DW_THREADS;
    const int cgi = (int)(gtid % cg);
    float wr[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(w + (size_t)t * C + cgi * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(w + (size_t)t * C + cgi * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { wr[t][e] = a[e]; wr[t][4 + e] = b[e]; }
    }
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
                float f[8];
                unpack8(dy[((n * OH + oh) * (long)OW + ow) * cg + cgi], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += wr[kh * 3 + kw][e] * f[e];
            }
        }
        dx[p * cg + cgi] = pack8(acc);
    }
}

// ---------------------------------------------------------------- stride-2 data gradient, 2 x 2 blocks (round 4)
// The kernel above decodes one dx pixel per thread and loads the 1, 2 or 4 gradient pixels its parity class touches: 2.25 loads and
// three integer divisions per 16 bytes written.  Here a thread owns two dx rows (2a, 2a + 1) over 2 SW columns: the 2 x (SW + 1)
// gradient pixels dy[a .. a + 1][b0 .. b0 + SW] feed all 4 SW outputs (0.625 loads per output at SW = 4), one index decode per strip.
// Same taps in the same order per output as the kernel above (out-of-range taps add +0): bit-identical.
template <int SW>
__global__ __launch_bounds__(DW_THREADS) void dw3x3_dgrad_s2_blk_kernel(const u32x4* __restrict__ dy, const float* __restrict__ w,
                                                                        u32x4* __restrict__ dx, int N, int H, int W, int OH, int OW, int C) {
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * DW_THREADS + threadIdx.x;
    const long nthreads = (long)gridDim.x * DW_THREADS;
    const int cgi = (int)(gtid % cg);
    float wr[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(w + (size_t)t * C + cgi * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(w + (size_t)t * C + cgi * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { wr[t][e] = a[e]; wr[t][4 + e] = b[e]; }
    }
    const int strips_w = (OW + SW - 1) / SW;
    const long nstrips = (long)N * OH * strips_w;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    for (long s = gtid / cg; s < nstrips; s += nthreads / cg) {
        const unsigned su = (unsigned)s, ru = su / (unsigned)strips_w;
        const int sw = (int)(su - ru * (unsigned)strips_w);
        const unsigned nu = ru / (unsigned)OH;
        const int a = (int)(ru - nu * (unsigned)OH);
        const long n = nu;
        const int b0 = sw * SW;
        u32x4 gp[2][SW + 1];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const bool rok = a + rr < OH;
            const u32x4* grow = dy + ((n * OH + (rok ? a + rr : a)) * (long)OW) * cg + cgi;
#pragma unroll
            for (int j = 0; j <= SW; ++j) gp[rr][j] = (rok && b0 + j < OW) ? grow[(long)(b0 + j) * cg] : zero4;
        }
        float g[2][SW + 1][8];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int j = 0; j <= SW; ++j) unpack8(gp[rr][j], g[rr][j]);
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int ih = 2 * a + pr;
            if (ih >= H) continue;
            u32x4* orow = dx + ((n * H + ih) * (long)W) * cg + cgi;
#pragma unroll
            for (int jj = 0; jj < 2 * SW; ++jj) {
                const int iw = 2 * b0 + jj;
                if (iw >= W) break;
                const int pc = jj & 1, jb = jj >> 1;
                float acc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int t = pr + 1 - kh;               // ih + 1 - kh = 2 a + t
                    if (t < 0 || (t & 1)) continue;
                    const int rr = t >> 1;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int u = pc + 1 - kw;
                        if (u < 0 || (u & 1)) continue;
                        const int cj = jb + (u >> 1);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] += wr[kh * 3 + kw][e] * g[rr][cj][e];
                    }
                }
                orow[(long)iw * cg] = pack8(acc);
            }
        }
    }
}

// ---------------------------------------------------------------- forward / weight gradient, stride 2, LDS-TILED (round 4)
// A workgroup owns a slice of GROUPS channel groups and walks 8 x TW output tiles (TW = 16 at 8 groups, 32 at 4 groups; 4 outputs per
// thread): the 17 x (2 TW + 1) input window goes to LDS by DMA as one linear run of pixels (no row pitch: 72 704 / 71 680 bytes, two
// workgroups per CU), 1.1 fetched input bytes per input byte used, and a thread's 3 x 9 window is 27 ds_read_b128.  Tap order per output
// as dw3x3_fwd_kernel<2, 2>: bit-identical forward.
template <int GROUPS_>
struct DwTileS2 {
    static constexpr int GROUPS = GROUPS_, SW = 4, NSTRIP = 32 / GROUPS, TH = 8, TW = SW * NSTRIP;
    static constexpr int WINW = 2 * TW + 1, WINH = 2 * TH + 1, WINPX = WINW * WINH, PPP = 64 / GROUPS, PX_BYTES = 16 * GROUPS;
    static constexpr int PIECES = (WINPX + PPP - 1) / PPP, WIN_BYTES = PIECES * 1024;
};
using DwS2Wide = DwTileS2<8>;
using DwS2Thin = DwTileS2<4>;

template <int GROUPS_>
__global__ __launch_bounds__(DW_THREADS, 2) void dw3x3_fwd_s2_tile_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                                          u32x4* __restrict__ y, float* __restrict__ stats, int N, int H,
                                                                          int W, int OH, int OW, int C, const int reps, int tiles_x,
                                                                          int tiles_y) {
    using G = DwTileS2<GROUPS_>;
    constexpr int TH = G::TH, SW = G::SW, NSTRIP = G::NSTRIP, TW = G::TW, WINW = G::WINW, WINPX = G::WINPX, PX_BYTES = G::PX_BYTES,
                  PIECES = G::PIECES, GR = G::GROUPS, PPP = G::PPP;
    extern __shared__ __attribute__((aligned(1024))) char dsm[];
    const int cg = C / 8, slice = blockIdx.y;
    const int gs = min(GR, cg - slice * GR);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gl = tid & (GR - 1), sidx = (tid / GR) & (NSTRIP - 1), r = tid / (GR * NSTRIP);
    const bool live = gl < gs;
    const int cgi = slice * GR + (live ? gl : 0);
    float wr[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(w + (size_t)t * C + cgi * 8);
        const f32x4 b = *reinterpret_cast<const f32x4*>(w + (size_t)t * C + cgi * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { wr[t][e] = a[e]; wr[t][4 + e] = b[e]; }
    }
    float sv[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sv[0][e] = sv[1][e] = 0.f;
    u32x4 rs;
    {
        const unsigned long long v = reinterpret_cast<unsigned long long>(x);
        rs[0] = __builtin_amdgcn_readfirstlane((unsigned)v);
        rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
        rs[2] = __builtin_amdgcn_readfirstlane((unsigned)((long)N * H * W * C * 2));
        rs[3] = 0x00020000u;
    }
    const unsigned lds0 = hc_lds_addr(dsm);
    const int ntiles = N * tiles_y * tiles_x;
    const int dpx = lane / GR, dgr = lane & (GR - 1);
    const unsigned choff = (unsigned)((slice * GR * 8 + dgr * 8) * 2);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / (tiles_y * tiles_x), rem = tile - n * (tiles_y * tiles_x);
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        __syncthreads();
        for (int q = wid; q < PIECES; q += 4) {
            const int idx = q * PPP + dpx;
            const int wr_ = idx / WINW, px = idx - wr_ * WINW;
            const int iy = 2 * oy0 - 1 + wr_, ix = 2 * ox0 - 1 + px;
            const bool ok = dgr < gs && idx < WINPX && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const unsigned off = (unsigned)(((n * H + iy) * W + ix) * C * 2) + choff;
            hc_dma16(rs, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(q * 1024)), ok ? off : HC_OOB);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int oy = oy0 + r;
        if (live && oy < OH) {
            float acc[SW][8];
#pragma unroll
            for (int j = 0; j < SW; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
            const char* base = dsm + ((2 * r) * WINW + 2 * sidx * SW) * PX_BYTES + gl * 16;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                for (int c = 0; c < 2 * SW + 1; ++c) {
                    float f[8];
                    unpack8(*reinterpret_cast<const u32x4*>(base + (kh * WINW + c) * PX_BYTES), f);
#pragma unroll
                    for (int j = 0; j < SW; ++j) {
                        const int kw = c - 2 * j;
                        if (kw < 0 || kw > 2) continue;
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[j][e] += wr[kh * 3 + kw][e] * f[e];
                    }
                }
            }
            u32x4* orow = y + ((long)(n * OH + oy) * OW) * cg + cgi;
#pragma unroll
            for (int j = 0; j < SW; ++j) {
                const int ox = ox0 + sidx * SW + j;
                if (ox >= OW) break;
                orow[(long)ox * cg] = pack8(acc[j]);
                if (stats != nullptr) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { sv[0][e] += acc[j][e]; sv[1][e] += acc[j][e] * acc[j][e]; }
                }
            }
        }
    }
    if (stats != nullptr) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(dsm);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[tid * 17 + k * 8 + e] = live ? sv[k][e] : 0.f;
        __syncthreads();
        float* rep = stats + (size_t)((blockIdx.x + blockIdx.y * gridDim.x) % reps) * 2 * C;
        for (int o = tid; o < 2 * GR * 8; o += DW_THREADS) {
            const int k = o / (GR * 8), c = o - k * (GR * 8), g = c >> 3, e = c & 7;
            if (g >= gs) continue;
            float sum = 0.f;
            for (int t = g; t < DW_THREADS; t += GR) sum += red[t * 17 + k * 8 + e];
            atomicAdd(rep + (size_t)k * C + slice * GR * 8 + c, sum);
        }
    }
}

template <int GROUPS_>
__global__ __launch_bounds__(DW_THREADS, 2) void dw3x3_wgrad_s2_tile_kernel(const void* __restrict__ x, const u32x4* __restrict__ dy,
                                                                            float* __restrict__ dw, int N, int H, int W, int OH, int OW,
                                                                            int C, const int reps, int tiles_x, int tiles_y) {
    using G = DwTileS2<GROUPS_>;
    constexpr int TH = G::TH, SW = G::SW, NSTRIP = G::NSTRIP, TW = G::TW, WINW = G::WINW, WINPX = G::WINPX, PX_BYTES = G::PX_BYTES,
                  PIECES = G::PIECES, GR = G::GROUPS, PPP = G::PPP;
    extern __shared__ __attribute__((aligned(1024))) char dsm[];
    const int cg = C / 8, slice = blockIdx.y;
    const int gs = min(GR, cg - slice * GR);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gl = tid & (GR - 1), sidx = (tid / GR) & (NSTRIP - 1), r = tid / (GR * NSTRIP);
    const bool live = gl < gs;
    const int cgi = slice * GR + (live ? gl : 0);
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    u32x4 rs;
    {
        const unsigned long long v = reinterpret_cast<unsigned long long>(x);
        rs[0] = __builtin_amdgcn_readfirstlane((unsigned)v);
        rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
        rs[2] = __builtin_amdgcn_readfirstlane((unsigned)((long)N * H * W * C * 2));
        rs[3] = 0x00020000u;
    }
    const unsigned lds0 = hc_lds_addr(dsm);
    const int ntiles = N * tiles_y * tiles_x;
    const int dpx = lane / GR, dgr = lane & (GR - 1);
    const unsigned choff = (unsigned)((slice * GR * 8 + dgr * 8) * 2);
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / (tiles_y * tiles_x), rem = tile - n * (tiles_y * tiles_x);
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        __syncthreads();
        for (int q = wid; q < PIECES; q += 4) {
            const int idx = q * PPP + dpx;
            const int wr_ = idx / WINW, px = idx - wr_ * WINW;
            const int iy = 2 * oy0 - 1 + wr_, ix = 2 * ox0 - 1 + px;
            const bool ok = dgr < gs && idx < WINPX && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const unsigned off = (unsigned)(((n * H + iy) * W + ix) * C * 2) + choff;
            hc_dma16(rs, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(q * 1024)), ok ? off : HC_OOB);
        }
        const int oy = oy0 + r;
        u32x4 gp[SW];
        const u32x4* grow = dy + ((long)(n * OH + (oy < OH ? oy : 0)) * OW) * cg + cgi;
#pragma unroll
        for (int j = 0; j < SW; ++j) {
            const int ox = ox0 + sidx * SW + j;
            gp[j] = (live && oy < OH && ox < OW) ? grow[(long)ox * cg] : zero4;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (live && oy < OH) {
            const char* base = dsm + ((2 * r) * WINW + 2 * sidx * SW) * PX_BYTES + gl * 16;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                for (int c = 0; c < 2 * SW + 1; ++c) {
                    float f[8];
                    unpack8(*reinterpret_cast<const u32x4*>(base + (kh * WINW + c) * PX_BYTES), f);
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int j2 = c - kw;
                        if (j2 < 0 || (j2 & 1) || (j2 >> 1) >= SW) continue;
                        float g[8];
                        unpack8(gp[j2 >> 1], g);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[kh * 3 + kw][e] += g[e] * f[e];
                    }
                }
            }
        }
    }
    float* red = reinterpret_cast<float*>(dsm);
    float* rep = dw + (size_t)((blockIdx.x + blockIdx.y * gridDim.x) % reps) * 9 * C;
#pragma unroll
    for (int k0 = 0; k0 < 9; k0 += 3) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[tid * 25 + k * 8 + e] = live ? acc[k0 + k][e] : 0.f;
        __syncthreads();
        for (int o = tid; o < 3 * GR * 8; o += DW_THREADS) {
            const int k = o / (GR * 8), c = o - k * (GR * 8), g = c >> 3, e = c & 7;
            if (g >= gs) continue;
            float sum = 0.f;
            for (int t = g; t < DW_THREADS; t += GR) sum += red[t * 25 + k * 8 + e];
            atomicAdd(rep + (size_t)(k0 + k) * C + slice * GR * 8 + c, sum);
        }
    }
}

// ---------------------------------------------------------------- weight gradient
// dw[replica][t][c] += sum over (n, oh, ow) of dy * x(tap t)
// a thread owns 8 channels and walks strips of TW output pixels of a row: the (TW - 1) S + 3 input columns of each of the
// three rows are loaded once and feed every (pixel, kw) pair they belong to - 5.5 16-byte loads per pixel instead of 10
template <int STRIDE, int TW>
__global__ __launch_bounds__(DW_THREADS) void dw3x3_wgrad_kernel(const u32x4* __restrict__ x, const u32x4* __restrict__ dy,
                                                                 float* __restrict__ dw, int N, int H, int W, int OH, int OW, int C, const int reps) {
    extern __shared__ float sred[];
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * DW_THREADS + threadIdx.x;
    const long nthreads = (long)gridDim.x * DW_THREADS;
    const int cgi = (int)(gtid % cg);
    const int strips_w = (OW + TW - 1) / TW;
    const long nstrips = (long)N * OH * strips_w;
    constexpr int IWN = (TW - 1) * STRIDE + 3;
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    for (long s = gtid / cg; s < nstrips; s += nthreads / cg) {
        const unsigned su = (unsigned)s, ru = su / (unsigned)strips_w;   // 32-bit: the launcher bounds the strip count
        const int sw = (int)(su - ru * (unsigned)strips_w);
        const unsigned nu = ru / (unsigned)OH;
        const int oh = (int)(ru - nu * (unsigned)OH);
        const long n = nu;
        const int ow0 = sw * TW;
        float g[TW][8];
        const u32x4* grow = dy + ((n * OH + oh) * (long)OW + ow0) * cg + cgi;
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            if (ow0 + j < OW) {
                unpack8(grow[(long)j * cg], g[j]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) g[j][e] = 0.f;
            }
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * STRIDE + kh - 1;
            if (ih < 0 || ih >= H) continue;
            const u32x4* row = x + ((n * H + ih) * (long)W) * cg + cgi;
#pragma unroll
            for (int c = 0; c < IWN; ++c) {
                const int iw = ow0 * STRIDE + c - 1;
                if (iw < 0 || iw >= W) continue;
                float f[8];
                unpack8(row[(long)iw * cg], f);
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int kw = c - j * STRIDE;
                    if (kw < 0 || kw > 2) continue;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[kh * 3 + kw][e] += g[j][e] * f[e];
                }
            }
        }
    }
    block_reduce_flush<9>(acc, cg, C, dw + (size_t)(blockIdx.x % reps) * 9 * C, sred);
}

// ---------------------------------------------------------------- weight gradient, stride 1, LDS-TILED (round 4)
// Same tiling as dw3x3_fwd_tile_kernel: the 10 x 34 x 64-channel window of x goes to LDS by DMA once per 8 x 32 tile, a thread multiplies
// the 8 gradient pixels of its strip (read straight from global memory, 16 bytes each, 128-byte runs over the channel groups) with
// the 3 x 10 window around them and keeps its 72 running sums for the whole kernel; one LDS reduction over the 32 threads of a channel
// group and one atomic per (tap, channel) at the end, into the replica slab the strip kernel writes too.
template <int NSTRIP_, int GROUPS_ = 8>
__global__ __launch_bounds__(DW_THREADS, 2) void dw3x3_wgrad_tile_kernel(const void* __restrict__ x, const u32x4* __restrict__ dy,
                                                                         float* __restrict__ dw, int N, int H, int W, int C, const int reps,
                                                                         int tiles_x, int tiles_y) {
    using G = DwTile<NSTRIP_, GROUPS_>;
    constexpr int TH = G::TH, SW = G::SW, NSTRIP = G::NSTRIP, TW = G::TW, ROW_BYTES = G::ROW_BYTES, PX_BYTES = G::PX_BYTES, PIECES = G::PIECES,
                  PIECES_ROW = G::PIECES_ROW, GR = G::GROUPS, PPP = G::PPP;
    (void)TH;
    extern __shared__ __attribute__((aligned(1024))) char dsm[];
    const int cg = C / 8, slice = blockIdx.y;
    const int gs = min(GR, cg - slice * GR);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gl = tid & (GR - 1), sidx = (tid / GR) & (NSTRIP - 1), r = tid / (GR * NSTRIP);
    const bool live = gl < gs;
    const int cgi = slice * GR + (live ? gl : 0);
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    u32x4 rs;
    {
        const unsigned long long v = reinterpret_cast<unsigned long long>(x);
        rs[0] = __builtin_amdgcn_readfirstlane((unsigned)v);
        rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
        rs[2] = __builtin_amdgcn_readfirstlane((unsigned)((long)N * H * W * C * 2));
        rs[3] = 0x00020000u;
    }
    const unsigned lds0 = hc_lds_addr(dsm);
    const int ntiles = N * tiles_y * tiles_x;
    const int dpx = lane / GR, dgr = lane & (GR - 1);
    const unsigned choff = (unsigned)((slice * GR * 8 + dgr * 8) * 2);
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / (tiles_y * tiles_x), rem = tile - n * (tiles_y * tiles_x);
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        __syncthreads();
        for (int q = wid; q < PIECES; q += 4) {
            const int wr_ = q / PIECES_ROW, px = (q - wr_ * PIECES_ROW) * PPP + dpx;
            const int iy = oy0 - 1 + wr_, ix = ox0 - 1 + px;
            const bool ok = dgr < gs && px < TW + 2 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const unsigned off = (unsigned)(((n * H + iy) * W + ix) * C * 2) + choff;
            hc_dma16(rs, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(q * 1024)), ok ? off : HC_OOB);
        }
        // the gradient pixels of this thread's strip, straight from global memory (in flight beside the window DMA)
        const int oy = oy0 + r;
        u32x4 gp[SW];
        const u32x4* grow = dy + ((long)(n * H + (oy < H ? oy : 0)) * W) * cg + cgi;
#pragma unroll
        for (int j = 0; j < SW; ++j) {
            const int ox = ox0 + sidx * SW + j;
            gp[j] = (live && oy < H && ox < W) ? grow[(long)ox * cg] : zero4;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (live && oy < H) {
            const char* base = dsm + r * ROW_BYTES + (sidx * SW) * PX_BYTES + gl * 16;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                for (int c = 0; c < SW + 2; ++c) {
                    float f[8];
                    unpack8(*reinterpret_cast<const u32x4*>(base + kh * ROW_BYTES + c * PX_BYTES), f);
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int j = c - kw;
                        if (j < 0 || j >= SW) continue;
                        float g[8];
                        unpack8(gp[j], g);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[kh * 3 + kw][e] += g[e] * f[e];
                    }
                }
            }
        }
    }
    // 9 x 8 sums per thread -> per (tap, channel) over the 32 threads of a channel group, three taps at a time through LDS
    float* red = reinterpret_cast<float*>(dsm);
    float* rep = dw + (size_t)((blockIdx.x + blockIdx.y * gridDim.x) % reps) * 9 * C;
#pragma unroll
    for (int k0 = 0; k0 < 9; k0 += 3) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[tid * 25 + k * 8 + e] = live ? acc[k0 + k][e] : 0.f;
        __syncthreads();
        for (int o = tid; o < 3 * GR * 8; o += DW_THREADS) {
            const int k = o / (GR * 8), c = o - k * (GR * 8), g = c >> 3, e = c & 7;
            if (g >= gs) continue;
            float sum = 0.f;
            for (int t = g; t < DW_THREADS; t += GR) sum += red[t * 25 + k * 8 + e];
            atomicAdd(rep + (size_t)(k0 + k) * C + slice * GR * 8 + c, sum);
        }
    }
}

// dw OIHW fp32 [C][1][3][3] = sum over replicas of slab [R][9][Cpad]; 8 lanes per element (c fastest: coalesced slab reads),
// each summing 16 replicas
__global__ __launch_bounds__(256) void dw3x3_wgrad_finish_kernel(const float* __restrict__ slab, float* __restrict__ dw, int C, int Cpad,
                                                                 int accumulate, const int reps) {
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    const bool live = i < C * 9;
    const int t = live ? i / C : 0, c = live ? i - t * C : 0;
    float s = 0.f;
    if (live)
        for (int r = sub; r < reps; r += 8) s += slab[((size_t)r * 9 + t) * Cpad + c];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    if (live && sub == 0) dw[c * 9 + t] = accumulate ? dw[c * 9 + t] + s : s;
}
// w OIHW fp32 [C][1][3][3] -> tap-major fp32 [9][Cpad] (zero padded), optionally flipped (stride-1 data gradient)
__global__ void dw3x3_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int C, int Cpad, int flip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cpad * 9) return;
    const int t = i / Cpad, c = i - t * Cpad;
    out[i] = c < C ? w[c * 9 + (flip ? 8 - t : t)] : 0.f;
}

// every depthwise kernel of a model in ONE launch (the per-step repack after the optimizer moved the weights): blockIdx.y = item
__global__ void dw3x3_pack_multi_kernel(const hc_dwpack_item* __restrict__ items) {
    const hc_dwpack_item it = items[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= it.Cpad * 9) return;
    const int t = i / it.Cpad, c = i - t * it.Cpad;
    it.out[i] = c < it.C ? it.w[c * 9 + (it.flip ? 8 - t : t)] : 0.f;
}

// elementwise max of two NHWC bf16 tensors (FReLU: max(x, bn(conv(x))), activation.py:79-82) and its gradient split
__global__ void max_fwd_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ o, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float fa[8], fb[8], fo[8];
        unpack8(a[i], fa);
        unpack8(b[i], fb);
#pragma unroll
        for (int e = 0; e < 8; ++e) fo[e] = fmaxf(fa[e], fb[e]);
        o[i] = pack8(fo);
    }
}
__global__ void max_bwd_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b, const u32x4* __restrict__ g,
                               u32x4* __restrict__ da, u32x4* __restrict__ db, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float fa[8], fb[8], fg[8], oa[8], ob[8];
        unpack8(a[i], fa);
        unpack8(b[i], fb);
        unpack8(g[i], fg);
#pragma unroll
        for (int e = 0; e < 8; ++e) {   // torch.max(a, b): ties split the gradient evenly
            const float wa = fa[e] > fb[e] ? 1.f : (fa[e] == fb[e] ? 0.5f : 0.f);
            oa[e] = fg[e] * wa;
            ob[e] = fg[e] * (1.f - wa);
        }
        da[i] = pack8(oa);
        db[i] = pack8(ob);
    }
}

// ---------------------------------------------------------------- squeeze-excite gate (rexnet.py:63-66) + ReLU6
// out = relu6(z * sigmoid(l[n][c])) ; z NHWC bf16 [N][HW][C], l bf16 [N][C] (gate logits).  act: 0 none, 6 relu6.
// IDX = unsigned when the element count plus one grid stride fits 32 bits (the launchers check): two 64-bit div / mod per 16-byte
// item were ~200 of the ~300 VALU instructions of these passes
template <typename IDX>
__global__ __launch_bounds__(DW_THREADS) void se_scale_fwd_kernel(const u32x4* __restrict__ z, const u32x4* __restrict__ l,
                                                                  u32x4* __restrict__ out, long N, long HW, int C, int act) {
    const int cg = C / 8;
    const IDX total = (IDX)(N * HW * cg), per_img = (IDX)(HW * cg);
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
        const IDX n = i / per_img;
        const int c = (int)((i - n * per_img) % (IDX)cg);
        float fz[8], fl[8], o[8];
        unpack8(z[i], fz);
        unpack8(l[n * cg + c], fl);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-fl[e]));
            float v = fz[e] * sg;
            if (act == 6) v = fminf(fmaxf(v, 0.f), 6.f);
            o[e] = v;
        }
        out[i] = pack8(o);
    }
}
// dgate[n][c] = sum_hw g * mask * z (fp32) ; grid (blocks_per_image, N): a thread keeps its channel group
__global__ __launch_bounds__(DW_THREADS) void se_scale_bwd_reduce_kernel(const u32x4* __restrict__ g, const u32x4* __restrict__ z,
                                                                         const u32x4* __restrict__ l, float* __restrict__ dgate,
                                                                         long HW, int C, int act) {
    extern __shared__ float sred[];
    const int cg = C / 8;
    const long n = blockIdx.y;
    const long gtid = (long)blockIdx.x * DW_THREADS + threadIdx.x;
    const long nthreads = (long)gridDim.x * DW_THREADS;
    const int cgi = (int)(gtid % cg);
    float fl[8], sg[8];
    unpack8(l[n * cg + cgi], fl);
#pragma unroll
    for (int e = 0; e < 8; ++e) sg[e] = __builtin_amdgcn_rcpf(1.f + __expf(-fl[e]));
    float sv[1][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sv[0][e] = 0.f;
    for (long p = gtid / cg; p < HW; p += nthreads / cg) {
        const long q = (n * HW + p) * cg + cgi;
        float fg[8], fz[8];
        unpack8(g[q], fg);
        unpack8(z[q], fz);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = fz[e] * sg[e];
            const float m = (act == 6) ? ((v > 0.f && v < 6.f) ? 1.f : 0.f) : 1.f;
            sv[0][e] += fg[e] * m * fz[e];
        }
    }
    block_reduce_flush<1>(sv, cg, C, dgate + n * C, sred);
}
// dl[n][c] = dgate * s (1 - s) (bf16)  — tiny
__global__ void se_gate_grad_kernel(const float* __restrict__ dgate, const bf16_t* __restrict__ l, bf16_t* __restrict__ dl, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = 1.f / (1.f + __expf(-bf16_to_f32(l[i])));
    dl[i] = f32_to_bf16(dgate[i] * s * (1.f - s));
}
// dz = g * mask * s + dpool[n][c] / HW   (dpool: gradient of the global average pool input, fp32 [N][C])
template <typename IDX>
__global__ __launch_bounds__(DW_THREADS) void se_scale_bwd_apply_kernel(const u32x4* __restrict__ g, const u32x4* __restrict__ z,
                                                                        const u32x4* __restrict__ l, const float* __restrict__ dpool,
                                                                        u32x4* __restrict__ dz, long N, long HW, int C, int act) {
    const int cg = C / 8;
    const IDX total = (IDX)(N * HW * cg), per_img = (IDX)(HW * cg);
    const float inv = 1.f / (float)HW;
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
        const IDX n = i / per_img;
        const int c = (int)((i - n * per_img) % (IDX)cg);
        float fg[8], fz[8], fl[8], o[8];
        unpack8(g[i], fg);
        unpack8(z[i], fz);
        unpack8(l[n * cg + c], fl);
        const float* dp = dpool + n * C + c * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-fl[e]));
            const float v = fz[e] * sg;
            const float m = (act == 6) ? ((v > 0.f && v < 6.f) ? 1.f : 0.f) : 1.f;
            o[e] = fg[e] * m * sg + dp[e] * inv;
        }
        dz[i] = pack8(o);
    }
}

// ---------------------------------------------------------------- NormConv2d helpers (functional.py:322-413)
// one wave per output pixel: lanes stride over (tap, 16-byte channel chunk)
__global__ __launch_bounds__(DW_THREADS) void patch_stats_kernel(const u32x4* __restrict__ x, int ld8, float* __restrict__ mean,
                                                                 float* __restrict__ rstd, int N, int H, int W, int OH, int OW, int C,
                                                                 int KH, int KW, int stride, int pad, float eps) {
    const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long npix = (long)N * OH * OW;
    if (pix >= npix) return;
    const int ow = (int)(pix % OW), oh = (int)((pix / OW) % OH);
    const long n = pix / ((long)OW * OH);
    const int cg = (C + 7) / 8;
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < KH * KW * cg; i += 64) {
        const int t = i / cg, c = i - t * cg;
        const int ih = oh * stride + t / KW - pad, iw = ow * stride + t % KW - pad;
        if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
        float f[8];
        unpack8(x[((n * H + ih) * (long)W + iw) * ld8 + c], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (c * 8 + e < C) { s1 += f[e]; s2 += f[e] * f[e]; }
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        const float k = (float)(C * KH * KW);
        const float m = s1 / k;
        float var = s2 / k - m * m;
        var = var > 0.f ? var : 0.f;
        mean[pix] = m;
        rstd[pix] = rsqrtf(var + eps);
    }
}
__global__ __launch_bounds__(DW_THREADS) void normconv_bwd_scale_kernel(const u32x4* __restrict__ g, const float* __restrict__ mean,
                                                                        const float* __restrict__ rstd, u32x4* __restrict__ gs,
                                                                        float* __restrict__ red, long npix, int C, const int reps) {
    extern __shared__ float sred[];
    const int cg = C / 8;
    const long gtid = (long)blockIdx.x * DW_THREADS + threadIdx.x;
    const long nthreads = (long)gridDim.x * DW_THREADS;
    const int cgi = (int)(gtid % cg);
    float sv[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sv[0][e] = sv[1][e] = 0.f;
    for (long p = gtid / cg; p < npix; p += nthreads / cg) {
        float f[8], o[8];
        unpack8(g[p * cg + cgi], f);
        const float r = rstd[p], rm = r * mean[p];
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[e] = f[e] * r; sv[0][e] += f[e]; sv[1][e] += f[e] * rm; }
        gs[p * cg + cgi] = pack8(o);
    }
    block_reduce_flush<2>(sv, cg, C, red + (size_t)(blockIdx.x % reps) * 2 * C, sred);
}

}  // namespace

extern "C" {

int hc_dw3x3_pack(const float* w, float* out, int32_t C, int32_t Cpad, int32_t flip, hc_stream_t stream) {
    if (w == nullptr || out == nullptr || C <= 0 || Cpad < C || (Cpad % 8) != 0) return HC_ERR_ARG;
    hipLaunchKernelGGL(dw3x3_pack_kernel, dim3((Cpad * 9 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, out, C, Cpad, flip);
    return hc_launch_status();
}

int hc_dw3x3_pack_multi(const hc_dwpack_item* items, int32_t nitems, int32_t max_cpad, hc_stream_t stream) {
    if (nitems < 0 || max_cpad < 0) return HC_ERR_ARG;
    if (nitems == 0 || max_cpad == 0) return HC_OK;
    if (items == nullptr || nitems > 65535) return HC_ERR_ARG;
    hipLaunchKernelGGL(dw3x3_pack_multi_kernel, dim3((max_cpad * 9 + 255) / 256, nitems), dim3(256), 0, (hipStream_t)stream, items);
    return hc_launch_status();
}

int hc_dw3x3_fwd(const void* x, const float* wpk, void* y, float* stats, int32_t N, int32_t H, int32_t W, int32_t C, int32_t stride,
                 hc_stream_t stream) {
    if ((long)N * H * W >= (1L << 31)) return HC_ERR_ARG;   // the kernels decode pixel indices in 32 bits
    if (x == nullptr || wpk == nullptr || y == nullptr || C <= 0 || (C % 8) != 0 || (stride != 1 && stride != 2)) return HC_ERR_ARG;
    const int OH = (H + 2 - 3) / stride + 1, OW = (W + 2 - 3) / stride + 1;
    if ((long)N * OH * OW == 0) return HC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int cg = C / 8;
    const size_t lds = stats != nullptr ? (size_t)DW_THREADS * 17 * sizeof(float) : 0;
    // HC_DW_TILE=0: the strip kernel for every stride-1 launch (A/B); the LDS-tiled kernel takes maps of at least HC_DW_TILE_MINW
    // (default 12) pixels and 32-bit byte offsets
    static const int tile_on = [] { const char* e = getenv("HC_DW_TILE"); return e == nullptr ? 1 : atoi(e); }();
    constexpr int tile_minw = 12;
    constexpr int tile_minc = 32;
    if (stride == 1 && tile_on && W >= tile_minw && H >= 8 && C >= tile_minc && (double)N * H * W * C * 2.0 < 4294967000.0) {
        const bool narrow = W <= 16;                       // 16 x 16 tiles for the 14 x 14 / 16 x 16 maps, 8 x 32 otherwise
        const bool thin = !narrow && dw_thin(cg);              // 32 channels or fewer: 4-group slices, 8 x 64 tiles
        const int tw = narrow ? DwTile<2>::TW : (thin ? DwThin::TW : DwTile<4>::TW), th = narrow ? DwTile<2>::TH : DwTile<4>::TH;
        const int tiles_x = (W + tw - 1) / tw, tiles_y = (H + th - 1) / th;
        const int nslices = thin ? (cg + 3) / 4 : (cg + 7) / 8;
        const long ntiles = (long)N * tiles_x * tiles_y;
        long gx = (2 * 256 + nslices - 1) / nslices;       // two resident workgroups per CU over all slices
        if (gx > ntiles) gx = ntiles;
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dw3x3_fwd_tile_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, DwTile<4>::WIN_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dw3x3_fwd_tile_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, DwTile<2>::WIN_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dw3x3_fwd_tile_kernel<8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, DwThin::WIN_BYTES);
            attr = true;
        }
        if (stats != nullptr && hc_get_deterministic() && gx * nslices > hc_get_stat_replicas()) return HC_ERR_ARG;
        if (narrow)
            hipLaunchKernelGGL(dw3x3_fwd_tile_kernel<2>, dim3((unsigned)gx, nslices), dim3(DW_THREADS), DwTile<2>::WIN_BYTES, st, x, wpk, (u32x4*)y,
                               stats, N, H, W, C, hc_get_stat_replicas(), tiles_x, tiles_y);
        else if (thin)
            hipLaunchKernelGGL((dw3x3_fwd_tile_kernel<8, 4>), dim3((unsigned)gx, nslices), dim3(DW_THREADS), DwThin::WIN_BYTES, st, x, wpk,
                               (u32x4*)y, stats, N, H, W, C, hc_get_stat_replicas(), tiles_x, tiles_y);
        else
            hipLaunchKernelGGL(dw3x3_fwd_tile_kernel<4>, dim3((unsigned)gx, nslices), dim3(DW_THREADS), DwTile<4>::WIN_BYTES, st, x, wpk, (u32x4*)y,
                               stats, N, H, W, C, hc_get_stat_replicas(), tiles_x, tiles_y);
    } else if (stride == 1 && OW > 4 && OW <= 7 && dw_row7()) {
        // 7 x 7 maps (ReXNet's last stage): one thread per output ROW - 3 x 7 loads for 7 outputs where strips of 4 + 3 take 3 x (6 + 5)
        constexpr int TW = 7;
        const long items = (long)N * OH * cg;
        hipLaunchKernelGGL((dw3x3_fwd_kernel<1, TW>), dim3(dw_blocks_exact(items, cg, dw_row7())), dim3(DW_THREADS), lds, st, (const u32x4*)x, wpk,
                           (u32x4*)y, stats, N, H, W, OH, OW, C, hc_get_stat_replicas());
    } else if (stride == 1) {
        constexpr int TW = 4;
        const long items = (long)N * OH * ((OW + TW - 1) / TW) * cg;
        hipLaunchKernelGGL((dw3x3_fwd_kernel<1, TW>), dim3(dw_blocks(items, cg, 2)), dim3(DW_THREADS), lds, st, (const u32x4*)x, wpk,
                           (u32x4*)y, stats, N, H, W, OH, OW, C, hc_get_stat_replicas());
    } else if (dw_s2_on() && OW >= dw_s2_minw() && OH >= 8 && C >= tile_minc && (double)N * H * W * C * 2.0 < 4294967000.0) {
        const bool thin = dw_thin(cg);
        const int tw = thin ? DwS2Thin::TW : DwS2Wide::TW, th = DwS2Wide::TH;
        const int tiles_x = (OW + tw - 1) / tw, tiles_y = (OH + th - 1) / th;
        const int nslices = thin ? (cg + 3) / 4 : (cg + 7) / 8;
        const long ntiles = (long)N * tiles_x * tiles_y;
        long gx = (2 * 256 + nslices - 1) / nslices;
        if (gx > ntiles) gx = ntiles;
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dw3x3_fwd_s2_tile_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, DwS2Wide::WIN_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dw3x3_fwd_s2_tile_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, DwS2Thin::WIN_BYTES);
            attr = true;
        }
        if (stats != nullptr && hc_get_deterministic() && gx * nslices > hc_get_stat_replicas()) return HC_ERR_ARG;
        if (thin)
            hipLaunchKernelGGL(dw3x3_fwd_s2_tile_kernel<4>, dim3((unsigned)gx, nslices), dim3(DW_THREADS), DwS2Thin::WIN_BYTES, st, x, wpk,
                               (u32x4*)y, stats, N, H, W, OH, OW, C, hc_get_stat_replicas(), tiles_x, tiles_y);
        else
            hipLaunchKernelGGL(dw3x3_fwd_s2_tile_kernel<8>, dim3((unsigned)gx, nslices), dim3(DW_THREADS), DwS2Wide::WIN_BYTES, st, x, wpk,
                               (u32x4*)y, stats, N, H, W, OH, OW, C, hc_get_stat_replicas(), tiles_x, tiles_y);
    } else {
        constexpr int TW = 2;
        const long items = (long)N * OH * ((OW + TW - 1) / TW) * cg;
        hipLaunchKernelGGL((dw3x3_fwd_kernel<2, TW>), dim3(dw_blocks(items, cg, 2)), dim3(DW_THREADS), lds, st, (const u32x4*)x, wpk,
                           (u32x4*)y, stats, N, H, W, OH, OW, C, hc_get_stat_replicas());
    }
    return hc_launch_status();
}

int hc_dw3x3_dgrad(const void* dy, const float* wpk, const float* wpk_flipped, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                   int32_t stride, hc_stream_t stream) {
    if (dy == nullptr || dx == nullptr || C <= 0 || (C % 8) != 0 || (stride != 1 && stride != 2)) return HC_ERR_ARG;
    if ((long)N * H * W >= (1L << 31)) return HC_ERR_ARG;
    if ((long)N * H * W == 0) return HC_OK;
    if (stride == 1) {   // correlation of dy with the flipped taps
        if (wpk_flipped == nullptr) return HC_ERR_ARG;
        return hc_dw3x3_fwd(dy, wpk_flipped, dx, nullptr, N, H, W, C, 1, stream);
    }
    if (wpk == nullptr) return HC_ERR_ARG;
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const int cg = C / 8;
    if (dw_s2_on() && OW >= 4) {
        constexpr int sw2 = 4;
        if (sw2 == 2) {
            const long items = (long)N * OH * ((OW + 1) / 2) * cg;
            hipLaunchKernelGGL(dw3x3_dgrad_s2_blk_kernel<2>, dim3(dw_blocks(items, cg, 4)), dim3(DW_THREADS), 0, (hipStream_t)stream,
                               (const u32x4*)dy, wpk, (u32x4*)dx, N, H, W, OH, OW, C);
        } else {
            const long items = (long)N * OH * ((OW + 3) / 4) * cg;
            hipLaunchKernelGGL(dw3x3_dgrad_s2_blk_kernel<4>, dim3(dw_blocks(items, cg, 4)), dim3(DW_THREADS), 0, (hipStream_t)stream,
                               (const u32x4*)dy, wpk, (u32x4*)dx, N, H, W, OH, OW, C);
        }
        return hc_launch_status();
    }
    hipLaunchKernelGGL(dw3x3_dgrad_s2_kernel, dim3(dw_blocks((long)N * H * W * cg, cg, 4)), dim3(DW_THREADS), 0, (hipStream_t)stream,
                       (const u32x4*)dy, wpk, (u32x4*)dx, N, H, W, OH, OW, C);
    return hc_launch_status();
}

int64_t hc_dw3x3_wgrad_ws_bytes(int32_t C) { return (int64_t)hc_get_stat_replicas() * 9 * C * sizeof(float); }

int hc_dw3x3_wgrad(const void* x, const void* dy, void* ws, float* dw, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Creal,
                   int32_t stride, int32_t accumulate, hc_stream_t stream) {
    if (x == nullptr || dy == nullptr || ws == nullptr || dw == nullptr || C <= 0 || (C % 8) != 0 || Creal <= 0 || Creal > C ||
        (stride != 1 && stride != 2) || (long)N * H * W >= (1L << 31))
        return HC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hc_zero_async(ws, (size_t)hc_dw3x3_wgrad_ws_bytes(C), st) != hipSuccess) return HC_ERR_LAUNCH;
    const int OH = (H + 2 - 3) / stride + 1, OW = (W + 2 - 3) / stride + 1;
    const int cg = C / 8;
    if ((long)N * OH * OW > 0) {
        const size_t lds = (size_t)DW_THREADS * 25 * sizeof(float);
        static const int tile_on = [] { const char* e = getenv("HC_DW_TILE"); return e == nullptr ? 1 : atoi(e); }();
        constexpr int tile_minw = 12;
        constexpr int tile_minc = 32;
        if (stride == 1 && tile_on && W >= tile_minw && H >= 8 && C >= tile_minc && (double)N * H * W * C * 2.0 < 4294967000.0 &&
            !hc_get_deterministic()) {
            const bool narrow = W <= 16;
            const bool thin = !narrow && dw_thin(cg);
            const int tw = narrow ? DwTile<2>::TW : (thin ? DwThin::TW : DwTile<4>::TW), th = narrow ? DwTile<2>::TH : DwTile<4>::TH;
            const int tiles_x = (W + tw - 1) / tw, tiles_y = (H + th - 1) / th;
            const int nslices = thin ? (cg + 3) / 4 : (cg + 7) / 8;
            const long ntiles = (long)N * tiles_x * tiles_y;
            long gx = (2 * 256 + nslices - 1) / nslices;
            if (gx > ntiles) gx = ntiles;
            static bool attr = false;
            if (!attr) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dw3x3_wgrad_tile_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, DwTile<4>::WIN_BYTES);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dw3x3_wgrad_tile_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, DwTile<2>::WIN_BYTES);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dw3x3_wgrad_tile_kernel<8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, DwThin::WIN_BYTES);
                attr = true;
            }
            if (narrow)
                hipLaunchKernelGGL(dw3x3_wgrad_tile_kernel<2>, dim3((unsigned)gx, nslices), dim3(DW_THREADS), DwTile<2>::WIN_BYTES, st, x,
                                   (const u32x4*)dy, (float*)ws, N, H, W, C, hc_get_stat_replicas(), tiles_x, tiles_y);
            else if (thin)
                hipLaunchKernelGGL((dw3x3_wgrad_tile_kernel<8, 4>), dim3((unsigned)gx, nslices), dim3(DW_THREADS), DwThin::WIN_BYTES, st, x,
                                   (const u32x4*)dy, (float*)ws, N, H, W, C, hc_get_stat_replicas(), tiles_x, tiles_y);
            else
                hipLaunchKernelGGL(dw3x3_wgrad_tile_kernel<4>, dim3((unsigned)gx, nslices), dim3(DW_THREADS), DwTile<4>::WIN_BYTES, st, x,
                                   (const u32x4*)dy, (float*)ws, N, H, W, C, hc_get_stat_replicas(), tiles_x, tiles_y);
        } else if (stride == 1 && OW > 4 && OW <= 7 && dw_row7()) {
            constexpr int TW = 7;
            const long items = (long)N * OH * cg;
            // (rows per thread, rexnet1_0x bs 256 one box: 2 -> 16.62 ms per step, 4 -> 16.67, 7 -> 16.82, 14 -> 17.15; strips of four: 16.72)
            hipLaunchKernelGGL((dw3x3_wgrad_kernel<1, TW>), dim3(dw_blocks_exact(items, cg, dw_row7())), dim3(DW_THREADS), lds, st, (const u32x4*)x,
                               (const u32x4*)dy, (float*)ws, N, H, W, OH, OW, C, hc_get_stat_replicas());
        } else if (stride == 1) {
            constexpr int TW = 4;
            const long items = (long)N * OH * ((OW + TW - 1) / TW) * cg;
            hipLaunchKernelGGL((dw3x3_wgrad_kernel<1, TW>), dim3(dw_blocks(items, cg, 4)), dim3(DW_THREADS), lds, st, (const u32x4*)x,
                               (const u32x4*)dy, (float*)ws, N, H, W, OH, OW, C, hc_get_stat_replicas());
        } else if (dw_s2_on() && OW >= dw_s2_minw() && OH >= 8 && C >= tile_minc && (double)N * H * W * C * 2.0 < 4294967000.0 &&
                   !hc_get_deterministic()) {
            const bool thin = dw_thin(cg);
            const int tw = thin ? DwS2Thin::TW : DwS2Wide::TW, th = DwS2Wide::TH;
            const int tiles_x = (OW + tw - 1) / tw, tiles_y = (OH + th - 1) / th;
            const int nslices = thin ? (cg + 3) / 4 : (cg + 7) / 8;
            const long ntiles = (long)N * tiles_x * tiles_y;
            long gx = (2 * 256 + nslices - 1) / nslices;
            if (gx > ntiles) gx = ntiles;
            static bool attr = false;
            if (!attr) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dw3x3_wgrad_s2_tile_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, DwS2Wide::WIN_BYTES);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dw3x3_wgrad_s2_tile_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, DwS2Thin::WIN_BYTES);
                attr = true;
            }
            if (thin)
                hipLaunchKernelGGL(dw3x3_wgrad_s2_tile_kernel<4>, dim3((unsigned)gx, nslices), dim3(DW_THREADS), DwS2Thin::WIN_BYTES, st, x,
                                   (const u32x4*)dy, (float*)ws, N, H, W, OH, OW, C, hc_get_stat_replicas(), tiles_x, tiles_y);
            else
                hipLaunchKernelGGL(dw3x3_wgrad_s2_tile_kernel<8>, dim3((unsigned)gx, nslices), dim3(DW_THREADS), DwS2Wide::WIN_BYTES, st, x,
                                   (const u32x4*)dy, (float*)ws, N, H, W, OH, OW, C, hc_get_stat_replicas(), tiles_x, tiles_y);
        } else {
            constexpr int TW = 2;
            const long items = (long)N * OH * ((OW + TW - 1) / TW) * cg;
            hipLaunchKernelGGL((dw3x3_wgrad_kernel<2, TW>), dim3(dw_blocks(items, cg, 8)), dim3(DW_THREADS), lds, st, (const u32x4*)x,
                               (const u32x4*)dy, (float*)ws, N, H, W, OH, OW, C, hc_get_stat_replicas());
        }
    }
    hipLaunchKernelGGL(dw3x3_wgrad_finish_kernel, dim3((Creal * 9 * 8 + 255) / 256), dim3(256), 0, st, (const float*)ws, dw, Creal, C,
                       accumulate, hc_get_stat_replicas());
    return hc_launch_status();
}

int hc_max_fwd(const void* a, const void* b, void* out, int64_t nelem, hc_stream_t stream) {
    if (a == nullptr || b == nullptr || out == nullptr || nelem < 0 || (nelem % 8) != 0) return HC_ERR_ARG;
    if (nelem == 0) return HC_OK;
    long blocks = (nelem / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(max_fwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)a, (const u32x4*)b, (u32x4*)out,
                       (long)(nelem / 8));
    return hc_launch_status();
}
int hc_max_bwd(const void* a, const void* b, const void* g, void* da, void* db, int64_t nelem, hc_stream_t stream) {
    if (a == nullptr || b == nullptr || g == nullptr || da == nullptr || db == nullptr || nelem < 0 || (nelem % 8) != 0) return HC_ERR_ARG;
    if (nelem == 0) return HC_OK;
    long blocks = (nelem / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(max_bwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)a, (const u32x4*)b,
                       (const u32x4*)g, (u32x4*)da, (u32x4*)db, (long)(nelem / 8));
    return hc_launch_status();
}

int hc_se_scale_fwd(const void* z, const void* gate_logits, void* out, int64_t N, int64_t HW, int32_t C, int32_t act, hc_stream_t stream) {
    if (z == nullptr || gate_logits == nullptr || out == nullptr || C <= 0 || (C % 8) != 0 || (act != 0 && act != 6)) return HC_ERR_ARG;
    const long total = (long)N * HW * (C / 8);
    if (total == 0) return HC_OK;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (total + 8192L * DW_THREADS < 4294967295L)
        hipLaunchKernelGGL(se_scale_fwd_kernel<unsigned>, dim3((int)blocks), dim3(DW_THREADS), 0, (hipStream_t)stream, (const u32x4*)z,
                           (const u32x4*)gate_logits, (u32x4*)out, (long)N, (long)HW, C, act);
    else
        hipLaunchKernelGGL(se_scale_fwd_kernel<long>, dim3((int)blocks), dim3(DW_THREADS), 0, (hipStream_t)stream, (const u32x4*)z,
                           (const u32x4*)gate_logits, (u32x4*)out, (long)N, (long)HW, C, act);
    return hc_launch_status();
}
int hc_se_scale_bwd_gate(const void* g, const void* z, const void* gate_logits, float* dgate, void* dlogits, int64_t N, int64_t HW,
                         int32_t C, int32_t act, hc_stream_t stream) {
    if (g == nullptr || z == nullptr || gate_logits == nullptr || dgate == nullptr || dlogits == nullptr || C <= 0 || (C % 8) != 0 ||
        (act != 0 && act != 6) || N > 65535)
        return HC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if ((long)N * C == 0) return HC_OK;
    if (hc_zero_async(dgate, sizeof(float) * (size_t)N * C, st) != hipSuccess) return HC_ERR_LAUNCH;
    const int cg = C / 8;
    if (HW > 0) {
        int bx = dw_blocks((long)HW * cg, cg, 8);
        hipLaunchKernelGGL(se_scale_bwd_reduce_kernel, dim3(bx, (unsigned)N), dim3(DW_THREADS), DW_THREADS * 9 * sizeof(float), st,
                           (const u32x4*)g, (const u32x4*)z, (const u32x4*)gate_logits, dgate, (long)HW, C, act);
    }
    const long n = (long)N * C;
    hipLaunchKernelGGL(se_gate_grad_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, st, (const float*)dgate,
                       (const bf16_t*)gate_logits, (bf16_t*)dlogits, n);
    return hc_launch_status();
}
int hc_se_scale_bwd_apply(const void* g, const void* z, const void* gate_logits, const float* dpool, void* dz, int64_t N, int64_t HW,
                          int32_t C, int32_t act, hc_stream_t stream) {
    if (g == nullptr || z == nullptr || gate_logits == nullptr || dpool == nullptr || dz == nullptr || C <= 0 || (C % 8) != 0 ||
        (act != 0 && act != 6))
        return HC_ERR_ARG;
    const long total = (long)N * HW * (C / 8);
    if (total == 0) return HC_OK;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (total + 8192L * DW_THREADS < 4294967295L)
        hipLaunchKernelGGL(se_scale_bwd_apply_kernel<unsigned>, dim3((int)blocks), dim3(DW_THREADS), 0, (hipStream_t)stream, (const u32x4*)g,
                           (const u32x4*)z, (const u32x4*)gate_logits, dpool, (u32x4*)dz, (long)N, (long)HW, C, act);
    else
        hipLaunchKernelGGL(se_scale_bwd_apply_kernel<long>, dim3((int)blocks), dim3(DW_THREADS), 0, (hipStream_t)stream, (const u32x4*)g,
                           (const u32x4*)z, (const u32x4*)gate_logits, dpool, (u32x4*)dz, (long)N, (long)HW, C, act);
    return hc_launch_status();
}

int hc_patch_stats(const void* x, int32_t x_ld, float* mean, float* rstd, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH,
                   int32_t KW, int32_t stride, int32_t pad, float eps, hc_stream_t stream) {
    if (x == nullptr || mean == nullptr || rstd == nullptr || C <= 0 || x_ld < C || (x_ld % 8) != 0 || KH < 1 || KW < 1 || stride < 1 ||
        pad < 0)
        return HC_ERR_ARG;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    const long npix = (long)N * OH * OW;
    if (npix <= 0) return HC_OK;
    hipLaunchKernelGGL(patch_stats_kernel, dim3((unsigned)((npix + 3) / 4)), dim3(DW_THREADS), 0, (hipStream_t)stream, (const u32x4*)x,
                       x_ld / 8, mean, rstd, N, H, W, OH, OW, C, KH, KW, stride, pad, eps);
    return hc_launch_status();
}
int hc_normconv_bwd_scale(const void* g, const float* mean, const float* rstd, void* gs, float* red, int64_t npix, int32_t C,
                          hc_stream_t stream) {
    if (g == nullptr || mean == nullptr || rstd == nullptr || gs == nullptr || red == nullptr || C <= 0 || (C % 8) != 0) return HC_ERR_ARG;
    if (npix == 0) return HC_OK;
    const int cg = C / 8;
    hipLaunchKernelGGL(normconv_bwd_scale_kernel, dim3(dw_blocks((long)npix * cg, cg, 8)), dim3(DW_THREADS),
                       DW_THREADS * 17 * sizeof(float), (hipStream_t)stream, (const u32x4*)g, mean, rstd, (u32x4*)gs, red, (long)npix, C, hc_get_stat_replicas());
    return hc_launch_status();
}

}  // extern "C"
