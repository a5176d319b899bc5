// Weight gradient of the conv stacks on CDNA4 MFMA.
//
//   dW[co][ci][kh][kw] = sum_m dy[m][co] * x[pix(m) + tap][ci]         (m over N*OH*OW)
//
// Replaces aten::convolution_backward(weight) behind nn.Conv2d (conv_sequence,
// holocron/models/utils.py:73).  GEMM view per tap: D[ci][co] with the reduction over output
// pixels.  Both operands are "K-major" in NHWC memory (channels contiguous, pixels strided),
// but MFMA wants 8 consecutive k per lane, so the staging pass transposes 8(pixel)x8(channel)
// bf16 blocks in registers (32 v_perm_b32) before the ds_write_b128.  Split-K over pixel
// ranges writes fp32 slabs [split][co][tap][ci]; a reduce kernel sums the slabs and emits the
// reference OIHW layout.
#include "common.h"
#include "../../include/holocron_hip.h"

// small-channel specialisation (conv_wgrad_tr.hip)
int wgrad_tr_nsplit(const hc_wgrad_desc& d);
int wgrad_dma_nsplit(const hc_wgrad_desc& d);      // conv_wgrad_dma.hip
int wgrad_dma_launch(const hc_wgrad_desc& d, hipStream_t st, int* nsplit_out);
int wgrad_tr_launch(const hc_wgrad_desc& d, hipStream_t st, int* nsplit_out);

namespace {

constexpr int BKP = 64;  // pixels per k-step

// swizzle valid for both the 8-lane staging writes (rows 8 apart) and the fragment reads
__device__ __forceinline__ int lds_off_t(int row, int chunk) {
    const int f = (((row >> 1) ^ (row >> 4)) & 1) | ((((row >> 2) ^ (row >> 5)) & 1) << 1) | (((row >> 3) & 1) << 2);
    return row * (BKP * 2) + ((chunk ^ f) << 4);
}

struct WgradArgs {
    hc_wgrad_desc d;
    int nsplit, steps_per_split, total_steps, n_ci_tiles;
};

// load an 8(pixel) x 8(channel) block: r[e] = 8 channels of pixel e
// transpose -> o[c] = 8 pixels of channel c
__device__ __forceinline__ void transpose8x8(const u32x4 (&r)[8], u32x4 (&o)[8]) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[2 * w][j] = __builtin_amdgcn_perm(r[2 * j + 1][w], r[2 * j][w], 0x05040100u);
            o[2 * w + 1][j] = __builtin_amdgcn_perm(r[2 * j + 1][w], r[2 * j][w], 0x07060302u);
        }
    }
}

template <int MR, int NR>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
    constexpr int WM = 2, WN = 2, NT = 256;
    constexpr int BA = 32 * MR * WM;  // ci tile (A rows)
    constexpr int BB = 32 * NR * WN;  // co tile (B cols)
    constexpr int ABLK = BA;          // number of 8x8 blocks in the A tile: (BA/8)*8
    constexpr int BBLK = BB;
    constexpr int AI = (ABLK + NT - 1) / NT, BI = (BBLK + NT - 1) / NT;
    constexpr int ABYTES = BA * BKP * 2;
    constexpr int STAGE = (BA + BB) * BKP * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const hc_wgrad_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int tb = (tid + 128) & 255;  // dy blocks are staged by the other half of the workgroup
    const int tapi = blockIdx.x / a.n_ci_tiles;
    const int cibase = (blockIdx.x % a.n_ci_tiles) * BA;
    const int cobase = blockIdx.y * BB;
    const int split = blockIdx.z;
    const int kh = tapi / d.KW, kw = tapi % d.KW;
    const int M = d.N * d.OH * d.OW;
    const int s_begin = split * a.steps_per_split;
    int s_end = s_begin + a.steps_per_split;
    if (s_end > a.total_steps) s_end = a.total_steps;

    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(d.x, (unsigned)d.N * d.IH * d.IW * d.Cin * 2u);
    const __amdgpu_buffer_rsrc_t rsy = make_rsrc(d.dy, (unsigned)M * d.Cout * 2u);

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 areg[AI][8], breg[BI][8];

    auto load_tiles = [&](int step) {
        const int kb = step * BKP;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int b = tid + i * NT;
            const int cg = b % (BA / 8), po = b / (BA / 8);
            const int ci = cibase + cg * 8;
            int m = kb + po * 8;
            const bool live = (b < ABLK) && (ci < d.Cin);
            int n = m / (d.OH * d.OW);
            int rem = m - n * (d.OH * d.OW);
            int oy = rem / d.OW, ox = rem - oy * d.OW;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int iy = oy * d.stride + kh - d.pad, ix = ox * d.stride + kw - d.pad;
                const bool ok = live && (m < M) && ((unsigned)iy < (unsigned)d.IH) && ((unsigned)ix < (unsigned)d.IW);
                const unsigned voff = ok ? (unsigned)((n * d.IH + iy) * d.IW + ix) * (unsigned)d.Cin * 2u + ci * 2u : HC_OOB;
                areg[i][e] = buf_load16(rsx, voff);
                ++m;
                if (++ox == d.OW) { ox = 0; if (++oy == d.OH) { oy = 0; ++n; } }
            }
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int b = tb + i * NT;
            const int cg = b % (BB / 8), po = b / (BB / 8);
            const int co = cobase + cg * 8;
            const int m0 = kb + po * 8;
            const bool live = (b < BBLK) && (co < d.Cout);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int m = m0 + e;
                const unsigned voff = (live && m < M) ? (unsigned)m * (unsigned)d.Cout * 2u + co * 2u : HC_OOB;
                breg[i][e] = buf_load16(rsy, voff);
            }
        }
    };
    auto store_tiles = [&](int stage) {
        char* sa = smem + stage * STAGE;
        char* sb = sa + ABYTES;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int b = tid + i * NT;
            if (AI * NT != ABLK && b >= ABLK) continue;
            const int cg = b % (BA / 8), po = b / (BA / 8);
            u32x4 o[8];
            transpose8x8(areg[i], o);
#pragma unroll
            for (int c = 0; c < 8; ++c) *reinterpret_cast<u32x4*>(sa + lds_off_t(cg * 8 + c, po)) = o[c];
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int b = tb + i * NT;
            if (BI * NT != BBLK && b >= BBLK) continue;
            const int cg = b % (BB / 8), po = b / (BB / 8);
            u32x4 o[8];
            transpose8x8(breg[i], o);
#pragma unroll
            for (int c = 0; c < 8; ++c) *reinterpret_cast<u32x4*>(sb + lds_off_t(cg * 8 + c, po)) = o[c];
        }
    };
    auto compute = [&](int stage) {
        const char* sa = smem + stage * STAGE;
        const char* sb = sa + ABYTES;
        const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < BKP / 16; ++kk) {
            bf16x8 fa[MR], fb[NR];
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
                fa[mr] = *reinterpret_cast<const bf16x8*>(sa + lds_off_t((wm * MR + mr) * 32 + lr, kk * 2 + lh));
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
                fb[nr] = *reinterpret_cast<const bf16x8*>(sb + lds_off_t((wn * NR + nr) * 32 + lr, kk * 2 + lh));
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mr], fb[nr], acc[mr][nr], 0, 0, 0);
        }
    };

    if (s_begin < s_end) {
        load_tiles(s_begin);
        store_tiles(0);
    }
    __syncthreads();
    for (int s = s_begin; s < s_end; ++s) {
        const bool more = (s + 1 < s_end);
        const int st = (s - s_begin) & 1;
        if (more) load_tiles(s + 1);
        compute(st);
        if (more) store_tiles(st ^ 1);
        __syncthreads();
    }

    // slab[split][co][tap][ci]: lane = co column, accumulator quads = 4 consecutive ci
    float* ws = reinterpret_cast<float*>(d.ws);
    const int T = d.KH * d.KW;
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int co = cobase + (wn * NR + nr) * 32 + lr;
        if (co >= d.Cout) continue;
        float* row = ws + (((long)split * d.Cout + co) * T + tapi) * d.Cin;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ci = cibase + (wm * MR + mr) * 32 + 8 * q + 4 * lh;
                if (ci >= d.Cin) continue;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[mr][nr][4 * q + e];
                *reinterpret_cast<f32x4*>(row + ci) = v;
            }
        }
    }
}

// dw[co][ci][t] = beta*dw + sum_split ws[split][co][t][ci].  One workgroup per (co, 64-wide ci tile, chunk of splits):
// thread (t, c) sums its element over the chunk's splits with reads coalesced along ci, the tile is turned
// through LDS and written as one contiguous run of 64*T floats of the OIHW gradient.  Few-channel layers have few
// (co, ci tile) pairs and hundreds of splits: blockIdx.z spreads the splits over more workgroups, which then combine with
// atomics (dw zeroed by the launcher when beta == 0).
// civ: valid input channels of the destination (<= Cin): dw is [gridDim.x][civ][T], slab columns ci >= civ are padding and dropped
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int Cout, int T, int Cin,
                                    int beta, int chunk, int civ) {
    extern __shared__ float sm[];  // [T][64]
    const int co = blockIdx.x, ci0 = blockIdx.y * 64;
    const int t = threadIdx.x / 64, c = threadIdx.x % 64;
    const long slab = (long)Cout * T * Cin;
    const int k0 = blockIdx.z * chunk, k1 = min(nsplit, k0 + chunk);
    float s = 0.f;
    if (ci0 + c < Cin) {
        const float* p = ws + ((long)co * T + t) * Cin + ci0 + c;
        int k = k0;
        for (; k + 16 <= k1; k += 16) {   // 16 independent loads in flight (the launch is a chain of nsplit / 16 memory round trips)
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p[(long)(k + u) * slab];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; k + 4 <= k1; k += 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p[(long)(k + u) * slab];
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u];
        }
        for (; k < k1; ++k) s += p[(long)k * slab];
    }
    sm[t * 64 + c] = s;
    __syncthreads();
    const int j = threadIdx.x;          // element j of the [64][T] output run
    const int cl = j / T, tt = j - cl * T;
    if (ci0 + cl < civ) {
        float* o = dw + ((long)co * civ + ci0 + cl) * T + tt;
        const float v = sm[tt * 64 + cl];
        if (gridDim.z > 1) atomicAdd(o, v);
        else *o = beta ? *o + v : v;
    }
}

// Grouped form: blockIdx.z = job; job j's slabs start at ws + j * nsplit * slab and its gradient is dws.p[j] (= or +=, fixed order)
struct WgradGroupPtrs { float* p[HC_WGRAD_MAX_JOBS]; };
__global__ void wgrad_reduce_group_kernel(const float* __restrict__ ws, const WgradGroupPtrs dws, int nsplit, int Cout, int T, int Cin,
                                          int beta) {
    extern __shared__ float sm[];  // [T][64]
    const int co = blockIdx.x, ci0 = blockIdx.y * 64, job = blockIdx.z;
    const int t = threadIdx.x / 64, c = threadIdx.x % 64;
    const long slab = (long)Cout * T * Cin;
    float s = 0.f;
    if (ci0 + c < Cin) {
        const float* p = ws + (long)job * nsplit * slab + ((long)co * T + t) * Cin + ci0 + c;
        int k = 0;
        for (; k + 8 <= nsplit; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long)(k + u) * slab];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < nsplit; ++k) s += p[(long)k * slab];
    }
    sm[t * 64 + c] = s;
    __syncthreads();
    const int j = threadIdx.x;
    const int cl = j / T, tt = j - cl * T;
    if (ci0 + cl < Cin) {
        float* o = dws.p[job] + ((long)co * Cin + ci0 + cl) * T + tt;
        const float v = sm[tt * 64 + cl];
        *o = beta ? *o + v : v;
    }
}

// the split range goes over several workgroups when the (co, ci tile) grid alone cannot fill the chip
static inline int launch_wgrad_reduce(const hc_wgrad_desc& d, int nsplit, hipStream_t st) {
    const int T = d.KH * d.KW;
    const int cov = d.co_valid > 0 ? d.co_valid : d.Cout, civ = d.ci_valid > 0 ? d.ci_valid : d.Cin;
    if (cov > d.Cout || civ > d.Cin) return HC_ERR_ARG;
    const int base = cov * ((civ + 63) / 64);
    int nz = 1;
    if (base < 512 && nsplit >= 64 && !hc_get_deterministic()) {   // (the z-slices combine with atomics)
        nz = (512 + base - 1) / base;
        if (nz > nsplit / 16) nz = nsplit / 16;
        if (nz < 1) nz = 1;
    }
    const int chunk = (nsplit + nz - 1) / nz;
    nz = (nsplit + chunk - 1) / chunk;
    if (nz > 1 && !d.beta) {
        if (hc_zero_async(d.dw, sizeof(float) * (size_t)cov * civ * T, st) != hipSuccess) return HC_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cov, (civ + 63) / 64, nz), dim3(64 * T), 64 * T * sizeof(float), st,
                       reinterpret_cast<const float*>(d.ws), d.dw, nsplit, d.Cout, T, d.Cin, d.beta, chunk, civ);
    return hc_launch_status();
}

// plan_only: size the split count (workspace query) without launching
template <int MR, int NR>
int launch_wgrad(const hc_wgrad_desc& d, hipStream_t st, bool plan_only, int* nsplit_out) {
    constexpr int BA = 64 * MR, BB = 64 * NR;
    constexpr int smem = 2 * (BA + BB) * BKP * 2;
    WgradArgs a;
    a.d = d;
    const int M = d.N * d.OH * d.OW;
    const int T = d.KH * d.KW;
    a.total_steps = (M + BKP - 1) / BKP;
    a.n_ci_tiles = (d.Cin + BA - 1) / BA;
    const int n_co_tiles = (d.Cout + BB - 1) / BB;
    const int tiles = a.n_ci_tiles * T * n_co_tiles;
    auto kern = wgrad_kernel<MR, NR>;
    static int occ = 0;
    if (occ == 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 256, smem) != hipSuccess || n < 1) n = 1;
        occ = n;
    }
    // one resident round of workgroups (a partial second round would cost a full round of time)
    int nsplit = (256 * occ) / tiles;
    if (nsplit > a.total_steps) nsplit = a.total_steps;
    if (nsplit < 1) nsplit = 1;
    a.steps_per_split = (a.total_steps + nsplit - 1) / nsplit;
    nsplit = (a.total_steps + a.steps_per_split - 1) / a.steps_per_split;
    a.nsplit = nsplit;
    *nsplit_out = nsplit;
    if (plan_only) return HC_OK;
    dim3 grid(a.n_ci_tiles * T, n_co_tiles, nsplit);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, a);
    return launch_wgrad_reduce(d, nsplit, st);
}

// tile shape (MR, NR) -> ci tile 64*MR, co tile 64*NR: least padded work, then fewest tiles
struct TileSel { int mr, nr; };
TileSel select_tiles(const hc_wgrad_desc& d) {
    static const TileSel opts[] = {{1, 1}, {2, 2}, {3, 1}, {3, 2}};
    TileSel best = opts[1];
    double best_cost = 1e30;
    for (const TileSel& o : opts) {
        const int ba = 64 * o.mr, bb = 64 * o.nr;
        const double padded = (double)((d.Cin + ba - 1) / ba * ba) * ((d.Cout + bb - 1) / bb * bb);
        // small tiles re-read more operand bytes per flop: penalise by 1/ba + 1/bb
        const double cost = padded * (1.0 + 24.0 * (1.0 / ba + 1.0 / bb));
        if (cost < best_cost) { best_cost = cost; best = o; }
    }
    return best;
}

int generic_dispatch(const hc_wgrad_desc& d, hipStream_t st, bool plan_only, int* ns) {
    const TileSel t = select_tiles(d);
    if (t.mr == 1) return launch_wgrad<1, 1>(d, st, plan_only, ns);
    if (t.mr == 2) return launch_wgrad<2, 2>(d, st, plan_only, ns);
    if (t.nr == 1) return launch_wgrad<3, 1>(d, st, plan_only, ns);
    return launch_wgrad<3, 2>(d, st, plan_only, ns);
}

}  // namespace

// ---- grouped launch of same-shaped layers (conv_wgrad_dma.hip) --------------------------------------------------------------
int wgrad_dma_group_nsplit(const hc_wgrad_desc& d, int njobs);
int wgrad_dma_group_launch(const hc_wgrad_desc& d, const void* const* xs, const void* const* dys, int njobs, hipStream_t st, int* nsplit_out);

static bool group_template(const hc_wgrad_group_desc* g, hc_wgrad_desc& d) {
    if (g == nullptr || g->njobs < 1 || g->njobs > HC_WGRAD_MAX_JOBS) return false;
    d = hc_wgrad_desc{};
    d.x = g->x[0]; d.dy = g->dy[0]; d.dw = g->dw[0]; d.ws = g->ws;
    d.N = g->N; d.IH = g->IH; d.IW = g->IW; d.Cin = g->Cin; d.OH = g->OH; d.OW = g->OW; d.Cout = g->Cout;
    d.KH = g->KH; d.KW = g->KW; d.stride = g->stride; d.pad = g->pad; d.beta = g->beta;
    if ((d.Cin % 8) != 0 || (d.Cout % 8) != 0 || d.stride < 1 || d.KH * d.KW > 16) return false;
    if ((double)d.N * d.IH * d.IW * d.Cin * 2.0 >= 4294967280.0 || (double)d.N * d.OH * d.OW * d.Cout * 2.0 >= 4294967280.0) return false;
    return true;
}

extern "C" int hc_conv_wgrad_group_supported(const hc_wgrad_group_desc* g) {
    hc_wgrad_desc d;
    if (!group_template(g, d)) return 0;
    return wgrad_dma_group_nsplit(d, g->njobs) > 0 ? 1 : 0;
}

extern "C" int64_t hc_conv_wgrad_group_ws_bytes(const hc_wgrad_group_desc* g) {
    hc_wgrad_desc d;
    if (!group_template(g, d)) return -1;
    const int ns = wgrad_dma_group_nsplit(d, g->njobs);
    if (ns <= 0) return -1;
    return (int64_t)g->njobs * ns * d.Cout * d.KH * d.KW * d.Cin * 4;
}

extern "C" int hc_conv_wgrad_group(const hc_wgrad_group_desc* g, hc_stream_t stream) {
    hc_wgrad_desc d;
    if (!group_template(g, d) || g->ws == nullptr) return HC_ERR_ARG;
    for (int j = 0; j < g->njobs; ++j)
        if (g->x[j] == nullptr || g->dy[j] == nullptr || g->dw[j] == nullptr) return HC_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int ns = 0;
    const int rc = wgrad_dma_group_launch(d, g->x, g->dy, g->njobs, st, &ns);
    if (rc < 0) return HC_ERR_ARG;            // shape outside the DMA kernel's plan: the caller launches the layers one by one
    if (rc != HC_OK) return rc;
    const int T = d.KH * d.KW;
    WgradGroupPtrs pp;
    for (int j = 0; j < HC_WGRAD_MAX_JOBS; ++j) pp.p[j] = j < g->njobs ? g->dw[j] : nullptr;
    hipLaunchKernelGGL(wgrad_reduce_group_kernel, dim3(d.Cout, (d.Cin + 63) / 64, g->njobs), dim3(64 * T), 64 * T * sizeof(float), st,
                       reinterpret_cast<const float*>(g->ws), pp, ns, d.Cout, T, d.Cin, d.beta);
    return hc_launch_status();
}

extern "C" int64_t hc_conv_wgrad_ws_bytes(const hc_wgrad_desc* d) {
    if (d == nullptr) return -1;
    int ns = wgrad_dma_nsplit(*d);
    if (ns == 0) ns = wgrad_tr_nsplit(*d);
    if (ns == 0 && generic_dispatch(*d, nullptr, true, &ns) != HC_OK) return -1;
    return (int64_t)ns * d->Cout * d->KH * d->KW * d->Cin * 4;
}

extern "C" int hc_conv_wgrad(const hc_wgrad_desc* dp, hc_stream_t stream) {
    if (dp == nullptr) return HC_ERR_ARG;
    const hc_wgrad_desc& d = *dp;
    if (d.x == nullptr || d.dy == nullptr || d.dw == nullptr || d.ws == nullptr) return HC_ERR_ARG;
    if ((d.Cin % 8) != 0 || (d.Cout % 8) != 0 || d.stride < 1 || d.KH * d.KW > 16) return HC_ERR_ARG;
    if ((double)d.N * d.IH * d.IW * d.Cin * 2.0 >= 4294967280.0) return HC_ERR_ARG;
    if ((double)d.N * d.OH * d.OW * d.Cout * 2.0 >= 4294967280.0) return HC_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int ns = 0;
    int rc = wgrad_dma_launch(d, st, &ns);
    if (rc < 0) rc = wgrad_tr_launch(d, st, &ns);
    if (rc >= 0) {
        if (rc != HC_OK) return rc;
        return launch_wgrad_reduce(d, ns, st);
    }
    return generic_dispatch(d, st, false, &ns);
}
