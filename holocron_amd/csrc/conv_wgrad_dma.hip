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
    const void* xs[HC_WGRAD_MAX_JOBS];    // per-job operands of a grouped launch (same-shaped layers): job = blockIdx.y / nsplit
    const void* dys[HC_WGRAD_MAX_JOBS];
    int njobs, nsplit;
    int M;                       // output pixels
    int total_steps, steps_per_split;
    int n_co_tiles, n_ci_tiles, n_tg;
    int adv_n, adv_oh, adv_ow;   // 32 pixels expressed in (images, rows, columns)
};

constexpr int NS = 3;            // pipeline stages
constexpr int NTAB = 4;          // pixel-table slots
constexpr int SUB = 4096;        // bytes of a [32 px][64 ch] sub-tile

__device__ __forceinline__ bf16x8 tr_pair(const char* base, int off) {
    const wd_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wd_lds_s16x4*)(base + off));
    const wd_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wd_lds_s16x4*)(base + off + 512));   // pixels +4
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// buffer_load ... lds issued through inline assembly: the compiler's wait-count insertion treats the builtin as a store
// to LDS that every later LDS read (and every barrier) must wait for with vmcnt(0), which serialises the pipeline.
// The explicit `s_waitcnt vmcnt(n)` + barrier in the main loop is the only ordering these loads need.
__device__ __forceinline__ void dma16(const u32x4 rsrc, unsigned lds_off, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_off), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ u32x4 raw_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    u32x4 r;
    r[0] = (unsigned)a;
    r[1] = (unsigned)(a >> 32) & 0xffffu;
    r[2] = bytes;
    r[3] = 0x00020000u;
    return r;
}

template <int WM, int WN, int TG>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_dma_kernel(const Args a) {
    constexpr int NW = WM * WN;
    constexpr int SUBA = WM, SUBB = TG * WN, NSUB = SUBA + SUBB;
    constexpr int STAGE = NSUB * SUB;
    constexpr int NI = NSUB * 4;                     // DMA instructions per stage
    constexpr int CEILI = (NI + NW - 1) / NW;        // per wave (padded with zero-fill dummies: constant vmcnt)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    int* table = reinterpret_cast<int*>(smem + NS * STAGE + 1024);   // [NTAB][32][4]: xoff, vmask, dyoff, -
    const unsigned lds0 = (unsigned)(size_t)(wd_lds_void*)smem;      // LDS byte address of the staging area
    const unsigned dummy_off = lds0 + NS * STAGE;                    // 1 KB sink of the padding instructions

    const hc_wgrad_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid % WM, wn = wid / WM;
    int b = blockIdx.x;
    const int tg = b % a.n_tg;  b /= a.n_tg;
    const int cot = b % a.n_co_tiles;
    const int cit = b / a.n_co_tiles;
    const int co0 = cot * 64 * WM, ci0 = cit * 64 * WN;
    const int jb = (int)blockIdx.y / a.nsplit;            // wave-uniform: which of the group's layers
    const int split = (int)blockIdx.y - jb * a.nsplit;
    const int T = d.KH * d.KW;
    const int kh = (TG == 1) ? 0 : tg;               // 3x3: one kernel row per group; 1x1: the only tap
    const int tap0 = tg * TG;

    const int s_begin = split * a.steps_per_split;
    int s_end = s_begin + a.steps_per_split;
    if (s_end > a.total_steps) s_end = a.total_steps;
    const int nsteps = s_end - s_begin;

    const u32x4 rsx = raw_rsrc(a.xs[jb], (unsigned)d.N * d.IH * d.IW * d.Cin * 2u);
    const u32x4 rsy = raw_rsrc(a.dys[jb], (unsigned)d.N * d.OH * d.OW * d.Cout * 2u);

    // ---- pixel table of one k-step: lanes 0..31 of wave 0 walk the pixels 32 at a time (no divisions in the loop) ----
    int tn = 0, toh = 0, tow = 0, tstep = 0;         // (n, oh, ow) of this lane's pixel at table step `tstep`
    {
        const int m = s_begin * 32 + (lane & 31);
        const int hw = d.OH * d.OW;
        tn = m / hw;
        const int r = m - tn * hw;
        toh = r / d.OW;
        tow = r - toh * d.OW;
    }
    auto make_table = [&]() {                        // writes table(tstep), then advances by 32 pixels
        const int m = (s_begin + tstep) * 32 + lane;
        int xoff = 0, vmask = 0;
        unsigned dyoff = HC_OOB;
        if (tstep < nsteps && m < a.M) {
            const int ih = toh * d.stride + kh - d.pad, iw0 = tow * d.stride - d.pad;
            if ((unsigned)ih < (unsigned)d.IH) {
#pragma unroll
                for (int t = 0; t < TG; ++t)
                    if ((unsigned)(iw0 + t) < (unsigned)d.IW) vmask |= 1 << t;
            }
            xoff = ((tn * d.IH + ih) * d.IW + iw0) * d.Cin * 2;     // may wrap below zero for iw0 = -1: fine
            dyoff = (unsigned)m * (unsigned)d.Cout * 2u;
        }
        int* e = table + ((tstep % NTAB) * 32 + lane) * 4;
        e[0] = xoff;
        e[1] = vmask;
        e[2] = (int)dyoff;
        ++tstep;
        tow += a.adv_ow;  toh += a.adv_oh;  tn += a.adv_n;          // 32 = adv_n * OH*OW + adv_oh * OW + adv_ow
        if (tow >= d.OW) { tow -= d.OW; ++toh; }
        if (toh >= d.OH) { toh -= d.OH; ++tn; }
    };

    // ---- DMA of one stage: everything that does not depend on the step is a per-lane constant ----------------
    const int prow = lane >> 3;                       // pixel row inside a 1 KB DMA slab (8 rows of 128 B)
    const int pchunk = lane & 7;                      // physical 16-byte chunk inside the row
    int i_tab[CEILI];                                 // table index (ints) of the pixel this lane stages
    unsigned i_cof[CEILI];                            // channel (+ tap) byte offset, HC_OOB when the channel is out of range
#pragma unroll
    for (int ii = 0; ii < CEILI; ++ii) {
        const int i = wid + ii * NW;
        const int sub = i >> 2, q = i & 3;
        const int p = 8 * q + prow;
        const int lc = pchunk ^ (((p >> 1) & 1) << 2);             // source-side swizzle
        i_tab[ii] = p * 4;
        if (sub < SUBA) {
            const int ch = co0 + sub * 64 + lc * 8;
            i_cof[ii] = ch < d.Cout ? (unsigned)ch * 2u : HC_OOB;
        } else {
            const int sb = sub - SUBA;
            const int t = sb / WN, w = sb - t * WN;
            const int ch = ci0 + w * 64 + lc * 8;
            i_cof[ii] = ch < d.Cin ? (unsigned)((t * d.Cin + ch) * 2) : HC_OOB;
        }
    }
    auto issue = [&](int step) {
        const unsigned st = lds0 + (unsigned)(step % NS) * STAGE;
        const int* tb = table + (step % NTAB) * 32 * 4;
#pragma unroll
        for (int ii = 0; ii < CEILI; ++ii) {
            const int i = wid + ii * NW;              // wave-uniform
            if (i < NI) {
                const int sub = i >> 2, q = i & 3;
                const int e0 = tb[i_tab[ii]], e1 = tb[i_tab[ii] + 1], e2 = tb[i_tab[ii] + 2];
                const unsigned dst = __builtin_amdgcn_readfirstlane(st + (unsigned)(sub * SUB + q * 1024));
                if (sub < SUBA) {
                    const bool ok = ((unsigned)e2 != HC_OOB) & (i_cof[ii] != HC_OOB);
                    dma16(rsy, dst, ok ? (unsigned)e2 + i_cof[ii] : HC_OOB);
                } else {
                    const int t = (sub - SUBA) / WN;
                    const bool ok = (((e1 >> t) & 1) != 0) & (i_cof[ii] != HC_OOB);
                    dma16(rsx, dst, ok ? (unsigned)e0 + i_cof[ii] : HC_OOB);
                }
            } else {
                dma16(rsx, __builtin_amdgcn_readfirstlane(dummy_off), HC_OOB);
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
    if (wid == 0 && lane < 32) { make_table(); make_table(); make_table(); }
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
        if (wid == 0 && lane < 32) make_table();     // table(s+3)
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
                    if (co < d.Cout) ws[(((long)(jb * a.nsplit + split) * d.Cout + co) * T + tap0 + t) * d.Cin + ci] = acc[t][mb][nb][r];
                }
            }
}

struct Plan {
    bool ok;
    Args a;
    int WM, WN, TG, nsplit;
};

inline Plan make_plan(const hc_wgrad_desc& d, const void* const* xs = nullptr, const void* const* dys = nullptr, int njobs = 1) {
    Plan pl{};
    pl.ok = false;
    if (njobs < 1 || njobs > HC_WGRAD_MAX_JOBS) return pl;
    static const int enable = getenv("HC_WDMA") ? atoi(getenv("HC_WDMA")) : 1;
    if (!enable) return pl;
    const int T = d.KH * d.KW;
    if (!((d.KH == 3 && d.KW == 3) || T == 1)) return pl;
    if (d.Cin % 64 || d.Cout % 64 || d.stride < 1) return pl;
    constexpr int min_c = 128;
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
    a.njobs = njobs;
    for (int j = 0; j < HC_WGRAD_MAX_JOBS; ++j) {
        a.xs[j] = j < njobs ? (xs != nullptr ? xs[j] : d.x) : nullptr;
        a.dys[j] = j < njobs ? (dys != nullptr ? dys[j] : d.dy) : nullptr;
    }
    a.M = d.N * d.OH * d.OW;
    a.total_steps = (a.M + 31) / 32;
    a.n_co_tiles = d.Cout / (64 * WM);
    a.n_ci_tiles = d.Cin / (64 * WN);
    a.n_tg = T / pl.TG;
    const int hw = d.OH * d.OW;
    a.adv_n = 32 / hw;
    a.adv_oh = (32 - a.adv_n * hw) / d.OW;
    a.adv_ow = 32 - a.adv_n * hw - a.adv_oh * d.OW;
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
    // Split-K count.  Default: one resident round of workgroups (n = 256 occ / tiles).  For RepVGG-A0's 1280 x 1280 x 3 x 3 layer that
    // is n = 1: 150 tiles of 392 steps on 256 CUs, 41 % of the chip idle for 531 us.  HC_WDMA_NSPLIT=0 picks n by a cost model instead
    // (rounds(tiles n) x steps per split + n slabs of Cout T Cin 8 bytes at ~4 TB/s against ~55 ns per MFMA of a step): n = 3 there,
    // 450 workgroups in two rounds of 131 steps.  Measured (same box, round 3): the weight-gradient family 2.56 -> 2.48 ms per step when
    // the step runs on ONE stream - and the headline step, whose weight gradients run on a second stream, 10.687 -> 10.72 ms: the CUs
    // the one-round launch leaves free are where the main stream's BatchNorm passes and data gradients run meanwhile.  So the rule
    // stays; the model is what a single-stream caller wants (HC_WDMA_NSPLIT=n > 0 forces n).
    const int slots = 256 * occ;
    static const int forced = getenv("HC_WDMA_NSPLIT") ? atoi(getenv("HC_WDMA_NSPLIT")) : -1;
    // a GROUP of same-shaped layers fills the chip with its tiles x jobs: the split-K factor - and with it the fp32 slab traffic,
    // which for YOLOv4's mid layers (3-6 tiles, 40-85 splits of a 0.6-2.4 MB slab) is several times the layer's own bytes - drops by
    // the group size
    int nsplit = slots / (tiles * a.njobs);
    if (forced == 0) {
        const double step_us = 0.055 * (2 * TG * 4);
        const double slab_steps = ((double)a.d.Cout * a.d.KH * a.d.KW * a.d.Cin * 8.0 / 4.0e6) / step_us;
        double best = 1e30;
        for (int n = 1; n <= 64 && n <= a.total_steps; ++n) {
            const int rounds = (tiles * n + slots - 1) / slots, per = (a.total_steps + n - 1) / n;
            const double cost = (double)rounds * per + slab_steps * n;
            if (cost < best) { best = cost; nsplit = n; }
        }
    } else if (forced > 0) {
        nsplit = forced;
    }
    if (nsplit < 1) nsplit = 1;
    if (nsplit > a.total_steps) nsplit = a.total_steps;
    a.steps_per_split = (a.total_steps + nsplit - 1) / nsplit;
    pl.nsplit = (a.total_steps + a.steps_per_split - 1) / a.steps_per_split;
    a.nsplit = pl.nsplit;
    if (!do_launch) return HC_OK;
    hipLaunchKernelGGL(kern, dim3(tiles, pl.nsplit * a.njobs), dim3(64 * WM * WN), smem, st, a);
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

// grouped form: njobs same-shaped layers in one launch; slab (job, split) at ws + (job * nsplit + split) * Cout T Cin floats
int wgrad_dma_group_nsplit(const hc_wgrad_desc& d, int njobs) {
    wdm::Plan pl = wdm::make_plan(d, nullptr, nullptr, njobs);
    if (!pl.ok) return 0;
    if (wdm::dispatch(pl, nullptr, false) != HC_OK) return 0;
    return pl.nsplit;
}

int wgrad_dma_group_launch(const hc_wgrad_desc& d, const void* const* xs, const void* const* dys, int njobs, hipStream_t st, int* nsplit_out) {
    wdm::Plan pl = wdm::make_plan(d, xs, dys, njobs);
    if (!pl.ok) return -1;
    const int rc = wdm::dispatch(pl, st, true);
    *nsplit_out = pl.nsplit;
    return rc;
}

int wgrad_dma_launch(const hc_wgrad_desc& d, hipStream_t st, int* nsplit_out) {
    wdm::Plan pl = wdm::make_plan(d);
    if (!pl.ok) return -1;
    const int rc = wdm::dispatch(pl, st, true);
    *nsplit_out = pl.nsplit;
    return rc;
}
