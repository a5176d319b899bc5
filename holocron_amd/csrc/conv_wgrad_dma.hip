// Weight gradient for wide layers: k-pipelined MFMA GEMM over output pixels whose operands travel
// HBM/L2 -> LDS by DMA (buffer_load ... lds, no staging registers) in their NATURAL NHWC layout and reach the matrix
// cores through the gfx950 transposing LDS read.
//
//   dW[co][kh][kw][ci] = sum_m dy[m][co] * x[pix(m) + (kh, kw)][ci],     m = (n, oh, ow) flattened
//
// GEMM view: M = co, N = ci (one accumulator set per tap of the group), K = m.  A workgroup owns a (64*WM co) x
// (64*WN ci) x TG-tap tile (TG = 3: one kernel row of a 3x3; TG = 1: 1x1) and a contiguous range of 32-pixel k-steps
// (split-K across workgroups, fp32 slabs reduced by wgrad_reduce_kernel).  Per k-step it stages
//   WM      sub-tiles [32 pixels][64 co] of dy and
//   TG * WN sub-tiles [32 pixels][64 ci] of x, one per tap: the shifted pixel of tap (kh, kw) is a different source
//           address per pixel (zero padding = out-of-range buffer offset), computed once per step by 32 lanes into a
//           small LDS table,
// each sub-tile = 4 KB = 4 wave-wide DMA instructions.  Both operands are pixel-major in memory while MFMA wants
// k (= pixels) contiguous per lane: ds_read_b64_tr_b16 does that transposition on the way out of LDS (semantics probed
// in scripts/probes/tr16_probe.hip).  The DMA destination is lane-linear, so the bank-conflict swizzle (the 64-byte
// half of a pixel row is XORed with bit 1 of the pixel index: the four pixels a 32-lane read phase touches land on
// four distinct 64-byte bank slots) is applied on the SOURCE address.  Three stages are in flight: the loads of
// step s+2 are issued before the MFMAs of step s, one barrier per step, `s_waitcnt vmcnt(n)` leaves the newest
// stage pending.
//
// MFMA: v_mfma_f32_32x32x16_bf16, wave tile 64 co x 64 ci x TG taps (2 x 2 x TG accumulators of 16 registers).
#include "common.h"
#include "../../include/holocron_hip.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) short wd_s16x4;
typedef __attribute__((address_space(3))) wd_s16x4 wd_lds_s16x4;
typedef __attribute__((address_space(3))) void wd_lds_void;

namespace wdm {

struct Args {
    hc_wgrad_desc d;
    int M;                       // output pixels
    int total_steps, steps_per_split;
    int n_co_tiles, n_ci_tiles, n_tg;
};

constexpr int NS = 3;            // pipeline stages
constexpr int NTAB = 4;          // pixel-table slots
constexpr int SUB = 4096;        // bytes of a [32 px][64 ch] sub-tile

__device__ __forceinline__ bf16x8 tr_pair(const char* base, int off) {
    const wd_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wd_lds_s16x4*)(base + off));
    const wd_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wd_lds_s16x4*)(base + off + 512));   // pixels +4
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int WM, int WN, int TG>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_dma_kernel(const Args a) {
    constexpr int NW = WM * WN;
    constexpr int SUBA = WM, SUBB = TG * WN, NSUB = SUBA + SUBB;
    constexpr int STAGE = NSUB * SUB;
    constexpr int NI = NSUB * 4;                     // DMA instructions per stage
    constexpr int CEILI = (NI + NW - 1) / NW;        // per wave (padded with zero-fill dummies: constant vmcnt)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* dummy = smem + NS * STAGE;                 // 1 KB sink of the padding instructions
    int* table = reinterpret_cast<int*>(smem + NS * STAGE + 1024);   // [NTAB][32][4]: xoff, vmask, dyoff, -

    const hc_wgrad_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid % WM, wn = wid / WM;
    int b = blockIdx.x;
    const int tg = b % a.n_tg;  b /= a.n_tg;
    const int cot = b % a.n_co_tiles;
    const int cit = b / a.n_co_tiles;
    const int co0 = cot * 64 * WM, ci0 = cit * 64 * WN;
    const int split = blockIdx.y;
    const int T = d.KH * d.KW;
    const int kh = (TG == 1) ? 0 : tg;               // 3x3: one kernel row per group; 1x1: the only tap
    const int tap0 = tg * TG;

    const int s_begin = split * a.steps_per_split;
    int s_end = s_begin + a.steps_per_split;
    if (s_end > a.total_steps) s_end = a.total_steps;
    const int nsteps = s_end - s_begin;

    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(d.x, (unsigned)d.N * d.IH * d.IW * d.Cin * 2u);
    const __amdgpu_buffer_rsrc_t rsy = make_rsrc(d.dy, (unsigned)d.N * d.OH * d.OW * d.Cout * 2u);

    // ---- pixel table of one k-step (lanes 0..31 of wave 0) ---------------------------------------------
    auto make_table = [&](int step) {   // step relative to s_begin; steps past the range are all-invalid
        const int j = lane;
        const int m = (s_begin + step) * 32 + j;
        int xoff = 0, vmask = 0;
        unsigned dyoff = HC_OOB;
        if (step < nsteps && m < a.M) {
            const int hw = d.OH * d.OW;
            const int n = m / hw;
            const int r = m - n * hw;
            const int oh = r / d.OW, ow = r - oh * d.OW;
            const int ih = oh * d.stride + kh - d.pad, iw0 = ow * d.stride - d.pad;
            if ((unsigned)ih < (unsigned)d.IH) {
#pragma unroll
                for (int t = 0; t < TG; ++t)
                    if ((unsigned)(iw0 + t) < (unsigned)d.IW) vmask |= 1 << t;
            }
            xoff = ((n * d.IH + ih) * d.IW + iw0) * d.Cin * 2;     // may wrap below zero for iw0 = -1: fine
            dyoff = (unsigned)m * (unsigned)d.Cout * 2u;
        }
        int* e = table + ((step % NTAB) * 32 + j) * 4;
        e[0] = xoff;
        e[1] = vmask;
        e[2] = (int)dyoff;
    };

    // ---- DMA of one stage --------------------------------------------------------------------------
    const int prow = lane >> 3;                       // pixel row inside a 1 KB DMA slab (8 rows of 128 B)
    const int pchunk = lane & 7;                      // physical 16-byte chunk inside the row
    auto issue = [&](int step) {
        char* st = smem + (step % NS) * STAGE;
        const int* tb = table + (step % NTAB) * 32 * 4;
#pragma unroll
        for (int ii = 0; ii < CEILI; ++ii) {
            const int i = wid + ii * NW;              // wave-uniform
            if (i < NI) {
                const int sub = i >> 2, q = i & 3;
                const int p = 8 * q + prow;
                const int lc = pchunk ^ (((p >> 1) & 1) << 2);     // source-side swizzle
                const int* e = tb + p * 4;
                unsigned voff;
                if (sub < SUBA) {
                    const int ch = co0 + sub * 64 + lc * 8;
                    const unsigned dyo = (unsigned)e[2];
                    voff = (dyo != HC_OOB && ch < d.Cout) ? dyo + (unsigned)ch * 2u : HC_OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsy, (wd_lds_void*)(st + sub * SUB + q * 1024), 16, voff, 0, 0, 0);
                } else {
                    const int sb = sub - SUBA;
                    const int t = sb / WN, w = sb - t * WN;
                    const int ch = ci0 + w * 64 + lc * 8;
                    const bool ok = ((e[1] >> t) & 1) && ch < d.Cin;
                    voff = ok ? (unsigned)e[0] + (unsigned)((t * d.Cin + ch) * 2) : HC_OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (wd_lds_void*)(st + sub * SUB + q * 1024), 16, voff, 0, 0, 0);
                }
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (wd_lds_void*)dummy, 16, HC_OOB, 0, 0, 0);
            }
        }
    };

    // ---- fragment addressing -----------------------------------------------------------------------
    const int g = lane >> 4, i16 = lane & 15;
    const int sw = (i16 >> 3) & 1;                                   // swizzle bit of this lane's pixels
    const int lane_off = (8 * (g >> 1) + (i16 >> 2)) * 128 + (4 * (g & 1) + (i16 & 3)) * 8;
    const int col0 = lane_off + ((0 ^ sw) << 6), col1 = lane_off + ((1 ^ sw) << 6);   // channel blocks 0 / 1 of a sub-tile

    f32x16 acc[TG][2][2];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][mb][nb][r] = 0.f;

    auto compute = [&](int step) {
        const char* st = smem + (step % NS) * STAGE;
        const char* sa = st + wm * SUB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ko = ks * 2048;                                // 16 pixels * 128 B
            const bf16x8 fa0 = tr_pair(sa, col0 + ko), fa1 = tr_pair(sa, col1 + ko);
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const char* sb = st + (SUBA + t * WN + wn) * SUB;
                const bf16x8 fb0 = tr_pair(sb, col0 + ko), fb1 = tr_pair(sb, col1 + ko);
                acc[t][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc[t][0][0], 0, 0, 0);
                acc[t][0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc[t][0][1], 0, 0, 0);
                acc[t][1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc[t][1][0], 0, 0, 0);
                acc[t][1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc[t][1][1], 0, 0, 0);
            }
        }
    };

    // ---- pipeline ------------------------------------------------------------------------------------
    if (wid == 0 && lane < 32) { make_table(0); make_table(1); make_table(2); }
    __syncthreads();
    issue(0);
    issue(1);
    for (int s = 0; s < nsteps; ++s) {
        // everything but the newest stage (CEILI instructions of this wave) has landed
        if (CEILI == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (CEILI == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (CEILI == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (CEILI == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (CEILI == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (CEILI == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (CEILI == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();                 // stage s visible to all, compute(s-1) finished, table(s+2) visible
        issue(s + 2);                    // into the buffer compute(s-1) just released
        if (wid == 0 && lane < 32) make_table(s + 3);
        compute(s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the zero-fill tail before the workgroup retires

    // ---- slab[split][co][tap][ci] ----------------------------------------------------------------------
    float* ws = reinterpret_cast<float*>(d.ws);
    const int ln = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int ci = ci0 + wn * 64 + nb * 32 + ln;
                if (ci >= d.Cin) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (co < d.Cout) ws[(((long)split * d.Cout + co) * T + tap0 + t) * d.Cin + ci] = acc[t][mb][nb][r];
                }
            }
}

struct Plan {
    bool ok;
    Args a;
    int WM, WN, TG, nsplit;
};

inline Plan make_plan(const hc_wgrad_desc& d) {
    Plan pl{};
    pl.ok = false;
    static const int enable = getenv("HC_WDMA") ? atoi(getenv("HC_WDMA")) : 1;
    if (!enable) return pl;
    const int T = d.KH * d.KW;
    if (!((d.KH == 3 && d.KW == 3) || T == 1)) return pl;
    if (d.Cin % 64 || d.Cout % 64 || d.stride < 1) return pl;
    static const int min_c = getenv("HC_WDMA_MINC") ? atoi(getenv("HC_WDMA_MINC")) : 128;
    if (d.Cin < min_c || d.Cout < min_c) return pl;     // narrower layers: the row-staged tr kernel reads x once, not per tap
    // measured (scripts/bench_layers.py): a 192-wide co tile (3 waves x 1) loses to the row-staged kernel on 192 x 192
    if (d.Cout % 256 != 0 && d.Cout % 192 == 0 && d.Cin <= 192) return pl;
    int WM, WN;
    if (d.Cout % 256 == 0) WM = 4;
    else if (d.Cout % 192 == 0) WM = 3;
    else if (d.Cout % 128 == 0) WM = 2;
    else return pl;
    WN = (d.Cin % 128 == 0) ? 2 : 1;     // at most 8 waves: 192 accumulator registers per wave for a kernel row
    pl.WM = WM;
    pl.WN = WN;
    pl.TG = T == 1 ? 1 : 3;
    Args& a = pl.a;
    a.d = d;
    a.M = d.N * d.OH * d.OW;
    a.total_steps = (a.M + 31) / 32;
    a.n_co_tiles = d.Cout / (64 * WM);
    a.n_ci_tiles = d.Cin / (64 * WN);
    a.n_tg = T / pl.TG;
    pl.ok = true;
    return pl;
}

template <int WM, int WN, int TG>
int launch(Plan& pl, hipStream_t st, bool do_launch) {
    constexpr int NSUB = WM + TG * WN;
    constexpr int smem = NS * NSUB * SUB + 1024 + NTAB * 32 * 16;
    auto kern = wgrad_dma_kernel<WM, WN, TG>;
    static int occ = 0;
    if (occ == 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 64 * WM * WN, smem) != hipSuccess || n < 1) n = 1;
        occ = n;
    }
    Args& a = pl.a;
    const int tiles = a.n_co_tiles * a.n_ci_tiles * a.n_tg;
    int nsplit = (256 * occ) / tiles;     // one resident round of workgroups
    if (nsplit < 1) nsplit = 1;
    if (nsplit > a.total_steps) nsplit = a.total_steps;
    a.steps_per_split = (a.total_steps + nsplit - 1) / nsplit;
    pl.nsplit = (a.total_steps + a.steps_per_split - 1) / a.steps_per_split;
    if (!do_launch) return HC_OK;
    hipLaunchKernelGGL(kern, dim3(tiles, pl.nsplit), dim3(64 * WM * WN), smem, st, a);
    return hc_launch_status();
}

inline int dispatch(Plan& pl, hipStream_t st, bool do_launch) {
#define WDM_CASE(M, N, G) \
    if (pl.WM == M && pl.WN == N && pl.TG == G) return launch<M, N, G>(pl, st, do_launch);
    WDM_CASE(4, 2, 3) WDM_CASE(4, 2, 1) WDM_CASE(3, 2, 3) WDM_CASE(3, 2, 1) WDM_CASE(2, 2, 3) WDM_CASE(2, 2, 1)
    WDM_CASE(4, 1, 3) WDM_CASE(4, 1, 1) WDM_CASE(3, 1, 3) WDM_CASE(3, 1, 1) WDM_CASE(2, 1, 3) WDM_CASE(2, 1, 1)
#undef WDM_CASE
    return -1;
}

}  // namespace wdm

int wgrad_dma_nsplit(const hc_wgrad_desc& d) {
    wdm::Plan pl = wdm::make_plan(d);
    if (!pl.ok) return 0;
    if (wdm::dispatch(pl, nullptr, false) != HC_OK) return 0;
    return pl.nsplit;
}

int wgrad_dma_launch(const hc_wgrad_desc& d, hipStream_t st, int* nsplit_out) {
    wdm::Plan pl = wdm::make_plan(d);
    if (!pl.ok) return -1;
    const int rc = wdm::dispatch(pl, st, true);
    *nsplit_out = pl.nsplit;
    return rc;
}
