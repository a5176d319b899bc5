// Implicit-GEMM gather convolution on CDNA4 MFMA (v_mfma_f32_32x32x16_bf16).
//
// One kernel serves the forward conv and the data gradient of Holocron's conv stacks
// (reference: nn.Conv2d built by conv_sequence, holocron/models/utils.py:73; RepBlock,
// holocron/models/classification/repvgg.py:71-73).  GEMM view, per parity class:
//     D[co][pix] = sum_{tap, k} Wpk[co][wt(tap)][k] * SRC_{src(tap)}[pix shifted by tap][k]
// A operand = packed weights (rows = output channels), B operand = gathered NHWC pixels
// (cols = output pixels), so every lane ends up with 4 consecutive output channels of one
// pixel per accumulator quad -> 8-byte NHWC stores.  Zero padding and ragged tiles come from
// buffer-descriptor range checks (out-of-range voffset reads 0), not from branches.
#include <cstdlib>
#include "common.h"
#include "../../include/holocron_hip.h"

namespace {



__device__ __forceinline__ float apply_act(float v, int act, float slope = 0.1f) {
    switch (act) {
        case 1: return v > 0.f ? v : 0.f;
        case 2: { float t = fminf(fmaxf(v + 2.f, 0.f), 2.f); return 0.5f * v * t; }
        case 3: return v > 0.f ? v : slope * v;
        case 4: { const float e = __expf(fminf(v, 20.f)), n = e * (e + 2.f); return v * n * __builtin_amdgcn_rcpf(n + 2.f); }
        case 5: return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
        case 6: return fminf(fmaxf(v, 0.f), 6.f);
        default: return v;
    }
}

// FP8: inference-only variant on OCP e4m3 bytes (config C5, reparametrised RepVGG).  A k-step of 64 fp8 channels has the
// byte geometry of a 32-element bf16 k-step, so staging, swizzle and DMA are shared (the host passes srcC / 2); the
// fragments are 32 bytes per lane (row = lane & 31, k = 32 * (lane >> 5) + [0, 32), probed in scripts/probes/mx_fp8_probe.hip)
// and one v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (E8M0 0x7f) replaces two bf16 MFMAs at twice the rate;
// the epilogue applies the per-channel dequant * requant factor and bias, ReLU, and stores fp8.
typedef __attribute__((ext_vector_type(8))) int hc_i32x8;
// NS / MINB: (2, 2) = the classic form - double buffer, DMA of step s + 1 behind the MFMAs of step s, two co-resident 4-wave
// workgroups per CU hide each other's barriers.  (4, 1) = the BIG-TILE form for layers with thousands of channels (RepVGG's
// 1280-channel blocks): ONE workgroup per CU on a 256 x 256 tile, half the LDS-DMA and L1 bytes per flop of the 128 x 128 tile.
// With nobody else on the CU the DMA runs THREE k32 steps ahead (asm-issued buffer_load ... lds + a counted s_waitcnt vmcnt: the
// compiler's own wait insertion would drain the queue at every barrier), its pieces are issued one at a time inside the MFMA stream,
// and the fragment reads of a step are skewed across the barrier.  Eight waves of 128 x 64 (two per SIMD, 128 accumulators each)
// beat four waves of 128 x 128 (one per SIMD, 256 accumulators): see the table at the dispatch in hc_conv_gather.
template <int MR, int NR, int WM, int WN, int BK, bool FP8, int NS = 2, int MINB = 2>
__global__ __launch_bounds__(64 * WM * WN, MINB) void conv_gather_kernel(const hc_conv_desc d, const int reps, const int flags) {
    static_assert(!FP8 || BK == 32, "fp8: 64 one-byte channels per k-step");
    // BIG = the deep-pipeline loop (asm-issued DMA NS - 1 steps ahead, skewed fragment schedule, pruned epilogue): the big tiles with one
    // workgroup per CU (MINB = 1) and, since round 6, the classic 128 x 64 / 128 x 128 tiles with two (NS = 3 / 4, MINB = 2) for
    // launches with long k-loops and few tiles - see `launch_deep`
    constexpr bool BIG = NS > 2;
    static_assert(!BIG || !FP8, "the deep-pipeline form is bf16 only");
    constexpr int NT = 64 * WM * WN;
    constexpr int BC = 32 * MR * WM;  // output-channel tile (A rows)
    constexpr int BP = 32 * NR * WN;  // output-pixel tile (B cols)
    constexpr int NC = BK / 8;        // 16-byte chunks per tile row
    constexpr int WBYTES = BC * BK * 2;
    constexpr int STAGE = (BC + BP) * BK * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // XCD-aware tile order: workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest), so a plain
    // (pixel tile, channel tile) grid puts every channel tile that is in flight on every XCD and the weight tiles thrash the
    // 4 MB L2s.  Give XCD k the k-th contiguous run of the channel-tile-major list instead: the workgroups that share an L2 share
    // one weight tile and walk neighbouring pixel tiles.  flags & 2: the parity classes of a stride-2 data gradient (grid z) are the
    // FASTEST index of that list - the four classes of a pixel tile read the same source pixels, and with z slowest a source tensor too
    // large for the caches is streamed four times (64@304 -> 32@608: 4 x 189 MB of reads for 378 MB of stores, 340 -> 303 us).  Only
    // then: the classes use different weight taps, and on the smaller maps (128@152 -> 64@304 and below) mixing them costs 7-12 %.
    int bx, by, bz = blockIdx.z;
    if (gridDim.z > 1 && (flags & 2) != 0) {
        const int gx = gridDim.x, ncls = gridDim.z;
        const int T = gx * gridDim.y * ncls, L = (blockIdx.z * gridDim.y + blockIdx.y) * gx + blockIdx.x;
        const int q = T >> 3, r = T & 7, xcd = L & 7, j = L >> 3;
        int tile = xcd * q + (xcd < r ? xcd : r) + j;
        bz = tile % ncls;
        tile /= ncls;
        by = tile / gx;
        bx = tile - by * gx;
    } else {      // per class plane (the classes have 1 / 2 / 2 / 4 taps: one run of a z-major list per XCD would leave the XCDs unevenly loaded)
        const int gx = gridDim.x, T = gx * gridDim.y, L = blockIdx.y * gx + blockIdx.x;
        const int q = T >> 3, r = T & 7, xcd = L & 7, j = L >> 3;
        const int tile = xcd * q + (xcd < r ? xcd : r) + j;
        by = tile / gx;
        bx = tile - by * gx;
    }
    const hc_conv_class& cl = d.cls[bz];
    const int OHg = cl.OHg, OWg = cl.OWg;
    const int M = d.N * OHg * OWg;
    // the class grid IS the output grid (every stride-1 forward conv and data gradient): output pixel index = m, no (n, i, j) decode - two
    // integer divisions per store of the epilogue's staged loop otherwise, ~1 us per workgroup whose whole life may be 10 us (1 x 1 layers).
    // Same box, two pairs (with the pure-1x1 prologue below): YOLOv4 26.17 -> 25.60 ms, rexnet1_0x 17.25 -> 16.89 ms
    const bool lin_out = cl.ostep == 1 && cl.oy0 == 0 && cl.ox0 == 0 && OHg == d.OH && OWg == d.OW;
    const int pbase = bx * BP;
    if (pbase >= M) return;
    const int cbase = by * BC;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int IH = d.IH, IW = d.IW, srcC = d.srcC, T = d.T, Cout = d.Cout;

    const unsigned src_bytes = (unsigned)d.N * IH * IW * srcC * 2u;
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(d.wpk, (unsigned)Cout * T * srcC * 2u);

    // ---- staging: direct-to-LDS DMA (buffer_load ... lds), no VGPR round trip, no ds_write ------
    // One wave instruction fills 1 KiB of LDS linearly (lane i -> +16 i bytes) = RPI tile rows.  The
    // XOR chunk swizzle of lds_off<BK>() therefore moves to the SOURCE side: the lane that owns
    // physical chunk pc of row r fetches logical chunk pc ^ f(r) of that row from global memory.
    constexpr int NW = WM * WN;
    constexpr int RPI = 512 / BK;            // tile rows per DMA instruction
    constexpr int WQ = BC / RPI, XQ = BP / RPI;   // DMA instructions per tile
    constexpr int WJ = (WQ + NW - 1) / NW, XJ = (XQ + NW - 1) / NW;
    static_assert(BC % RPI == 0 && BP % RPI == 0, "tile rows must be a multiple of the DMA granule");
    const int lrow = lane / NC, lpc = lane % NC;
    // Per staged row: byte offset of (tap offset (0,0), this lane's logical chunk) and a bitmask of
    // the taps whose source pixel is inside the image -> the per-step address is one add + select.
    unsigned x_base[XJ], x_vmask[XJ];
    // a pure 1 x 1 (one tap at offset (0, 0), source grid = class grid): source pixel = m, always inside the image - no decode
    const bool lin_in = cl.ntaps == 1 && (cl.tap[0] & 0xffff) == 0 && cl.istep == 1 && IH == OHg && IW == OWg;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
        const int row = (wid + j * NW) * RPI + lrow;
        const int m = pbase + row;
        const bool ok = (wid + j * NW < XQ) && (m < M);
        const int mm = ok ? m : 0;
        if (lin_in) {
            x_base[j] = (unsigned)mm * (unsigned)srcC * 2u + (unsigned)((lpc ^ ((row / (16 / NC)) % NC)) * 16);
            x_vmask[j] = ok ? 1u : 0u;
            continue;
        }
        const int n = mm / (OHg * OWg);
        const int rem = mm - n * (OHg * OWg);
        const int oi = rem / OWg, oj = rem - oi * OWg;
        const int iy0 = oi * cl.istep, ix0 = oj * cl.istep;
        x_base[j] = (unsigned)((n * IH + iy0) * IW + ix0) * (unsigned)srcC * 2u + (unsigned)((lpc ^ ((row / (16 / NC)) % NC)) * 16);
        unsigned vm = 0;
        for (int t = 0; t < cl.ntaps; ++t) {
            const int tp = cl.tap[t];
            const int iy = iy0 + (int)(signed char)(tp & 0xff), ix = ix0 + (int)(signed char)((tp >> 8) & 0xff);
            if (ok && ((unsigned)iy < (unsigned)IH) && ((unsigned)ix < (unsigned)IW)) vm |= 1u << t;
        }
        x_vmask[j] = vm;
    }
    unsigned w_off[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int row = (wid + j * NW) * RPI + lrow;
        const int co = cbase + row;
        const unsigned kc = (unsigned)((lpc ^ ((row / (16 / NC)) % NC)) * 16);
        w_off[j] = (wid + j * NW < WQ && co < Cout) ? (unsigned)co * T * srcC * 2u + kc : HC_OOB;
    }

    f32x16 acc[MR][NR];
#pragma unroll
    for (int a = 0; a < MR; ++a)
#pragma unroll
        for (int b = 0; b < NR; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int kcb = srcC / BK;
    const int S = cl.ntaps * kcb;
    typedef __attribute__((address_space(3))) void lds_void;

    auto issue = [&](int stage, int tap, int ck) {
        char* sw = smem + stage * STAGE;
        char* sx = sw + WBYTES;
        const int tp = cl.tap[tap];
        const int dy = (int)(signed char)(tp & 0xff), dx = (int)(signed char)((tp >> 8) & 0xff);
        const int sidx = (tp >> 16) & 0xff, wt = (tp >> 24) & 0xff;
        const __amdgpu_buffer_rsrc_t rsx = make_rsrc(sidx ? d.src1 : d.src0, src_bytes);
        // wave-uniform part of the address: tap shift + channel slice (may be "negative": unsigned wrap is fine)
        const unsigned tofs = (unsigned)((dy * IW + dx) * srcC * 2 + ck * BK * 2);
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            if (XQ % NW == 0 || wid + j * NW < XQ) {   // wave-uniform
                const unsigned voff = ((x_vmask[j] >> tap) & 1u) ? x_base[j] + tofs : HC_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_void*)(sx + (wid + j * NW) * 1024), 16, voff, 0, 0, 0);
            }
        }
        const unsigned wk = (unsigned)(wt * srcC + ck * BK) * 2u;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            if (WQ % NW == 0 || wid + j * NW < WQ) {
                const unsigned voff = (w_off[j] == HC_OOB) ? HC_OOB : w_off[j] + wk;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void*)(sw + (wid + j * NW) * 1024), 16, voff, 0, 0, 0);
            }
        }
    };
    // Fragment addresses: the swizzle term depends only on (lane, kk) because every fragment starts
    // at a multiple of 32 rows, so all reads of a k-step are base + compile-time immediates.
    int frag_off[BK / 16];
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) frag_off[kk] = lds_off<BK>(lane & 31, kk * 2 + (lane >> 5));
    const int a_row0 = wm * MR * 32 * BK * 2, b_row0 = WBYTES + wn * NR * 32 * BK * 2;
    const int f8_off0 = lds_off<BK>(lane & 31, 2 * (lane >> 5)), f8_off1 = lds_off<BK>(lane & 31, 2 * (lane >> 5) + 1);
    auto compute = [&](int stage) {
        const char* st = smem + stage * STAGE;
        if (FP8) {
            hc_i32x8 a[MR], b[NR];
            const char* pa = st + a_row0;
            const char* pb = st + b_row0;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
                const u32x4 lo = *reinterpret_cast<const u32x4*>(pa + mr * 32 * BK * 2 + f8_off0);
                const u32x4 hi = *reinterpret_cast<const u32x4*>(pa + mr * 32 * BK * 2 + f8_off1);
                a[mr] = hc_i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            }
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const u32x4 lo = *reinterpret_cast<const u32x4*>(pb + nr * 32 * BK * 2 + f8_off0);
                const u32x4 hi = *reinterpret_cast<const u32x4*>(pb + nr * 32 * BK * 2 + f8_off1);
                b[nr] = hc_i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            }
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    acc[mr][nr] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[mr], b[nr], acc[mr][nr], 0, 0, 0, 0x7f, 0, 0x7f);
            return;
        }
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 a[MR], b[NR];
            const char* pa = st + a_row0 + frag_off[kk];
            const char* pb = st + b_row0 + frag_off[kk];
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) a[mr] = *reinterpret_cast<const bf16x8*>(pa + mr * 32 * BK * 2);
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) b[nr] = *reinterpret_cast<const bf16x8*>(pb + nr * 32 * BK * 2);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mr], b[nr], acc[mr][nr], 0, 0, 0);
        }
    };

    if constexpr (!BIG) {
    // ---- main loop: the DMA of step s+1 is in flight during the MFMAs of step s; one barrier/step ----
    if (S > 0) issue(0, 0, 0);   // a parity class may have no taps (1x1 stride-2 dgrad): result is just resid
    int tap = 0, ck = 0;
    for (int s = 0; s < S; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of stage s has landed
        __syncthreads();                                     // ... everybody's has, and compute(s-1) is over
        if (s + 1 < S) {
            // taps inner, channel blocks outer: consecutive steps read the same pixels shifted by one tap, so the x rows are
            // still in L2 (one step of the co-resident workgroups apart) instead of a whole channel sweep apart (+6 % on the
            // 1280-channel layers, neutral elsewhere)
            if (++tap == cl.ntaps) { tap = 0; ++ck; }
            issue((s + 1) & 1, tap, ck);
        }
        compute(s & 1);
    }
    __syncthreads();
    } else {
    // ---- deep pipeline: stage s + NS - 1 is issued at step s (into the buffer step s - 1 just left), so NS - 2 whole steps of
    // DMA stay in flight across every barrier.  A wave issues PER = XJ + WJ instructions per stage, all unconditional: the counted
    // wait `vmcnt((NS - 2) PER)` is exactly "my share of stage s has landed".
    // every wave issues the same number of DMA instructions per stage (the counted wait needs that): the pixel rows must divide
    // evenly over the waves; weight rows that do not (BC = 96, 192: 6 / 12 pieces over 8 waves) are padded with zero-fill pieces
    // into a 1 KB scratch slot per wave behind the stages
    static_assert(XQ % NW == 0, "the pixel tile must deal whole DMA pieces to every wave");
    constexpr int PER = XJ + WJ;
    static_assert((NS - 2) * PER <= 63, "vmcnt immediate");
    const unsigned lds0 = hc_lds_addr(smem);
    constexpr bool WRAG = WQ % NW != 0;
    const unsigned scratch = lds0 + (unsigned)(NS * STAGE);
    const int widu = __builtin_amdgcn_readfirstlane(wid);
    auto uniform = [](const u32x4 r) __attribute__((always_inline)) {
        u32x4 o;
        o[0] = __builtin_amdgcn_readfirstlane(r[0]); o[1] = __builtin_amdgcn_readfirstlane(r[1]);
        o[2] = __builtin_amdgcn_readfirstlane(r[2]); o[3] = __builtin_amdgcn_readfirstlane(r[3]);
        return o;
    };
    const u32x4 qw = uniform(hc_raw_rsrc(d.wpk, (unsigned)Cout * T * srcC * 2u));
    const u32x4 qx0 = uniform(hc_raw_rsrc(d.src0, src_bytes));
    const u32x4 qx1 = uniform(hc_raw_rsrc(d.src1 != nullptr ? d.src1 : d.src0, src_bytes));
    auto issue_deep = [&](int stage, int tap_, int ck_) __attribute__((always_inline)) {
        const unsigned sw = lds0 + (unsigned)(stage * STAGE), sx = sw + WBYTES;
        const int tp = cl.tap[tap_];
        const int dy = (int)(signed char)(tp & 0xff), dx = (int)(signed char)((tp >> 8) & 0xff);
        const int sidx = (tp >> 16) & 0xff, wt = (tp >> 24) & 0xff;
        const unsigned tofs = (unsigned)((dy * IW + dx) * srcC * 2 + ck_ * BK * 2);
        if (sidx) {
#pragma unroll
            for (int j = 0; j < XJ; ++j)
                hc_dma16(qx1, __builtin_amdgcn_readfirstlane(sx + (unsigned)((widu + j * NW) * 1024)),
                         ((x_vmask[j] >> tap_) & 1u) ? x_base[j] + tofs : HC_OOB);
        } else {
#pragma unroll
            for (int j = 0; j < XJ; ++j)
                hc_dma16(qx0, __builtin_amdgcn_readfirstlane(sx + (unsigned)((widu + j * NW) * 1024)),
                         ((x_vmask[j] >> tap_) & 1u) ? x_base[j] + tofs : HC_OOB);
        }
        const unsigned wk = (unsigned)(wt * srcC + ck_ * BK) * 2u;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const bool real = !WRAG || widu + j * NW < WQ;     // wave-uniform
            hc_dma16(qw, __builtin_amdgcn_readfirstlane(real ? sw + (unsigned)((widu + j * NW) * 1024) : scratch + (unsigned)(widu * 1024)),
                     (w_off[j] == HC_OOB) ? HC_OOB : w_off[j] + wk);
        }
    };
    constexpr int KK = BK / 16;
    // Fragment schedule: with one wave per SIMD nobody else covers an LDS round trip, so the halves of a stage are skewed across the
    // barrier - the reads of (s, first half) fly under the MFMAs of (s - 1, second half), those of (s, second half) under the
    // MFMAs of (s, first half).  sched_barrier keeps the compiler from sinking every read next to its first use.
    bf16x8 fa0[MR], fb0[NR], fa1[MR], fb1[NR];
    auto read_half = [&](int stage, int kk, bf16x8 (&fa)[MR], bf16x8 (&fb)[NR]) __attribute__((always_inline)) {
        const char* st = smem + stage * STAGE;
        const char* pa = st + a_row0 + frag_off[kk];
        const char* pb = st + b_row0 + frag_off[kk];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) fa[mr] = *reinterpret_cast<const bf16x8*>(pa + mr * 32 * BK * 2);
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) fb[nr] = *reinterpret_cast<const bf16x8*>(pb + nr * 32 * BK * 2);
    };
    // The DMA of stage s + NS - 1 is not issued as a burst behind the barrier (where nothing covers it: an LDS-DMA instruction costs
    // the issuing wave 100-185 cycles next to other memory instructions, 8 of them per stage = most of a stage's 1031 MFMA cycles)
    // but ONE PIECE PER ROW OF FOUR MFMAS, inside the matrix stream, where it costs what fits between two MFMAs.
    // Branch-free on purpose: a stage past the end is issued all the same with out-of-range offsets (zero fill into a buffer nobody
    // reads - the vmcnt bookkeeping stays one constant), and the source descriptor of the tap is picked once per step; with scalar
    // branches around every piece the compiler's wait-count pass gave up on the LDS queue and put lgkmcnt(0) in front of the first
    // MFMA of every step, which serialises exactly the round trip the skew is there to hide.
    struct Pend { u32x4 qx; unsigned sw, sx, tofs, wk, live; int tap; };
    auto plan = [&](int stage, int tap_, int ck_, bool on, int tp) __attribute__((always_inline)) {   // tp = cl.tap[tap_], loaded a step ago
        Pend c;
        c.live = on ? 1u : 0u;
        c.sw = lds0 + (unsigned)(stage * STAGE);
        c.sx = c.sw + WBYTES;
        const int dy = (int)(signed char)(tp & 0xff), dx = (int)(signed char)((tp >> 8) & 0xff);
        const bool second = ((tp >> 16) & 0xff) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) c.qx[i] = second ? qx1[i] : qx0[i];
        c.tap = tap_;
        c.tofs = (unsigned)((dy * IW + dx) * srcC * 2 + ck_ * BK * 2);
        c.wk = (unsigned)(((tp >> 24) & 0xff) * srcC + ck_ * BK) * 2u;
        return c;
    };
    auto piece = [&](const Pend& c, const int q) __attribute__((always_inline)) {      // q: compile-time piece index, x rows first
        if (q < XJ) {
            const unsigned voff = (((x_vmask[q] >> c.tap) & c.live) != 0u) ? x_base[q] + c.tofs : HC_OOB;
            hc_dma16(c.qx, __builtin_amdgcn_readfirstlane(c.sx + (unsigned)((widu + q * NW) * 1024)), voff);
        } else {
            const int j = q - XJ;
            const unsigned voff = (w_off[j] == HC_OOB || c.live == 0u) ? HC_OOB : w_off[j] + c.wk;
            const bool real = !WRAG || widu + j * NW < WQ;     // wave-uniform; the padding pieces zero-fill the wave's scratch slot
            hc_dma16(qw, __builtin_amdgcn_readfirstlane(real ? c.sw + (unsigned)((widu + j * NW) * 1024) : scratch + (unsigned)(widu * 1024)), voff);
        }
    };
    // phase f of a step: MR rows of NR MFMAs on (fa, fb); the PER pieces of a stage are dealt over the KK phases (pieces
    // [f PER / KK, (f + 1) PER / KK) go to phase f) and piece k of a phase's n goes behind row k MR / n
    auto mma_phase = [&](const bf16x8 (&fa)[MR], const bf16x8 (&fb)[NR], const Pend& c, const int f, const bool mul) __attribute__((always_inline)) {
        const int p0 = (f * PER) / KK, p1 = ((f + 1) * PER) / KK, np = p1 - p0;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            if (mul) {
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mr], fb[nr], acc[mr][nr], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < PER; ++k)
                if (k < np && (k * MR) / (np > 0 ? np : 1) == mr) piece(c, p0 + k);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto mma_half = [&](const bf16x8 (&fa)[MR], const bf16x8 (&fb)[NR]) __attribute__((always_inline)) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
                acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mr], fb[nr], acc[mr][nr], 0, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < MR; ++i) fa1[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NR; ++i) fb1[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    int tap = 0, ck = 0;
    for (int p = 0; p < NS - 1; ++p) {                       // S >= NS - 1 (hc_conv_gather dispatches this form for S >= 128 only)
        issue_deep(p, tap, ck);
        if (++tap == cl.ntaps) { tap = 0; ++ck; }
    }
    // tap words live in the lanes of one VGPR: a scalar load inside the loop would put an SMEM op on the lgkm counter, and with one
    // outstanding (they return out of order) every LDS wait of the step degrades to lgkmcnt(0)
    const int tapv = cl.tap[lane < HC_MAX_TAPS ? lane : 0];
    int tpw = __builtin_amdgcn_readlane(tapv, tap);
#pragma unroll 1
    for (int s = 0; s < S; ++s) {
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): the fragments read a phase ago are in (free by now) - said with
                                                             // the builtin so that the compiler's own wait insertion knows it too
        hc_wait_vmcnt<(NS - 2) * PER>();                     // NS - 2 stages stay in flight (real or, past the end, zero-fill dummies)
        __syncthreads();                                     // lgkmcnt(0) + s_barrier on gfx950: the DMA queue is left alone
        const bool more = s + NS - 1 < S;
        const Pend c = plan((s + NS - 1) % NS, tap, ck, more, tpw);
        if (more && ++tap == cl.ntaps) { tap = 0; ++ck; }
        tpw = __builtin_amdgcn_readlane(tapv, tap);
        const int stg = s % NS;
        read_half(stg, 0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        mma_phase(fa1, fb1, c, 0, true);                     // (s - 1, last k16): read before the barrier, multiplied behind it (zeros at s = 0)
#pragma unroll
        for (int kk = 1; kk < KK; ++kk) {
            if (kk & 1) {
                read_half(stg, kk, fa1, fb1);
                __builtin_amdgcn_sched_barrier(0);
                mma_phase(fa0, fb0, c, kk, true);
            } else {
                read_half(stg, kk, fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                mma_phase(fa1, fb1, c, kk, true);
            }
        }
    }
    if (S > 0) mma_half(fa1, fb1);
    hc_wait_vmcnt<0>();
    __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------------------
    const int lr = lane & 31, lh = lane >> 5;
    // per-channel batch statistics of the fp32 result (training-mode BatchNorm), before rounding.
    // Contention matters more than instruction count here: a 112x112 layer has 25k workgroups and
    // only 2*Cout distinct addresses, so (1) the waves of a workgroup are combined in LDS first and
    // (2) the global atomics are spread over HC_STAT_REPLICAS copies that bn_finalize sums.
    if (d.stats != nullptr) {
        // [WN][2][BC], the staging tiles are dead now: one plane per pixel wave, summed in a fixed order below (no LDS
        // atomics - a workgroup's contribution must not depend on which of its waves finishes first)
        float* sred = reinterpret_cast<float*>(smem) + wn * 2 * BC;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            float s1[16], s2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    const float v = acc[mr][nr][r];
                    a1 += v;
                    a2 += v * v;
                }
                s1[r] = a1;
                s2[r] = a2;
            }
            // butterfly over the 32 pixel-lanes of each half wave; 16 values -> 1 per lane pair
#pragma unroll
            for (int w = 8, o = 16; w >= 1; w >>= 1, o >>= 1) {
                const bool up = (lane & o) != 0;
#pragma unroll
                for (int i = 0; i < w; ++i) {
                    const float k1 = up ? s1[i + w] : s1[i], g1 = up ? s1[i] : s1[i + w];
                    const float k2 = up ? s2[i + w] : s2[i], g2 = up ? s2[i] : s2[i + w];
                    s1[i] = k1 + __shfl_xor(g1, o);
                    s2[i] = k2 + __shfl_xor(g2, o);
                }
            }
            s1[0] += __shfl_xor(s1[0], 1);
            s2[0] += __shfl_xor(s2[0], 1);
            if ((lane & 1) == 0) {
                const int r = 8 * ((lane >> 4) & 1) + 4 * ((lane >> 3) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 1) & 1);
                const int cl_ = (wm * MR + mr) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                sred[cl_] = s1[0];
                sred[BC + cl_] = s2[0];
            }
        }
        __syncthreads();
        float* rep = d.stats + (size_t)(bx % reps) * 2 * Cout;
        for (int i = tid; i < 2 * BC; i += NT) {
            const int which = i / BC, c = i - which * BC;
            if (cbase + c < Cout) {
                const float* pl = reinterpret_cast<const float*>(smem) + i;
                float v = pl[0];
#pragma unroll
                for (int w = 1; w < WN; ++w) v += pl[w * 2 * BC];
                const int cg_ = cbase + c;
                if (!BIG && d.co_split > 0) {      // stacked convolutions: each has its own statistics array
                    const int C1 = d.co_split, C2 = Cout - d.co_split;
                    if (cg_ < C1) atomicAdd(d.stats + (size_t)(bx % reps) * 2 * C1 + which * C1 + cg_, v);
                    else atomicAdd(d.stats2 + (size_t)(bx % reps) * 2 * C2 + which * C2 + (cg_ - C1), v);
                } else {
                    atomicAdd(rep + which * Cout + cbase + c, v);
                }
            }
        }
    }

    if (FP8) {
        // out = fp8(act(acc * ch_mult[co] + bias[co])): 4 consecutive channels of a pixel = one 32-bit store
        unsigned char* dst8 = reinterpret_cast<unsigned char*>(d.dst);
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int m = pbase + (wn * NR + nr) * 32 + lr;
            if (m >= M) continue;
            const int n = m / (OHg * OWg);
            const int rem = m - n * (OHg * OWg);
            const int oi = rem / OWg, oj = rem - oi * OWg;
            const long pix = ((long)n * d.OH + (oi * cl.ostep + cl.oy0)) * d.OW + (oj * cl.ostep + cl.ox0);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = cbase + (wm * MR + mr) * 32 + 8 * q + 4 * lh;
                    if (co >= Cout) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[mr][nr][4 * q + e] * d.ch_mult[co + e] + (d.bias != nullptr ? d.bias[co + e] : 0.f);
                        if (d.act != 0) t = apply_act(t, d.act);
                        v[e] = fminf(fmaxf(t, -448.f), 448.f);       // e4m3fn has no inf: saturate
                    }
                    int pk = 0;
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], pk, false);
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
                    *reinterpret_cast<int*>(dst8 + pix * Cout + co) = pk;
                }
            }
        }
        return;
    }
    bf16_t* dst = reinterpret_cast<bf16_t*>(d.dst);
    const bf16_t* resid = reinterpret_cast<const bf16_t*>(d.resid);
    // Coalesced stores: a lane of the D tile holds 4 channels (8 bytes) of ONE pixel and its neighbours hold other pixels, so
    // direct stores touch 32 cache lines per instruction.  When whole 16-byte channel chunks exist (Cout % 8 == 0) the tile goes
    // through LDS instead ([pixel][channel] bf16, pitch BC * 2 + 8 bytes: conflict-free 8-byte writes) and leaves as 16-byte
    // stores whose consecutive lanes walk a pixel's channels - full rows of the NHWC destination.
    constexpr int OPITCH = BC * 2 + 8;
    const bool staged = (flags & 1) != 0 && (Cout & 7) == 0 && (d.co_split & 7) == 0;
    const bool res_after = d.resid_after_act != 0;
    const float aslope = d.act_slope != 0.f ? d.act_slope : 0.1f;
    char* ost = smem;
    // per-channel scale / shift of the epilogue (inference BatchNorm, bias): ONE coalesced read per workgroup into an LDS table behind
    // the output staging area.  Read from global memory inside the store loop they were eight dependent loads per accumulator quad
    // that the compiler cannot hoist past the stores: ~20 us per launch, more than the BatchNorm launches the fused epilogue replaces.
    constexpr int COEF_OFF = (BP * OPITCH + 15) / 16 * 16;
    float* ctab = reinterpret_cast<float*>(smem + COEF_OFF);
    const bool has_coef = d.ch_scale != nullptr || d.bias != nullptr;
    if (has_coef) {
        for (int i = tid; i < BC; i += NT) {
            const bool in = cbase + i < Cout;
            ctab[i] = (in && d.ch_scale != nullptr) ? d.ch_scale[cbase + i] : 1.f;
            ctab[BC + i] = (in && d.bias != nullptr) ? d.bias[cbase + i] : 0.f;
        }
    }
    if (staged || has_coef) __syncthreads();        // the statistics scratch / the last staging tiles are dead
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int m = pbase + (wn * NR + nr) * 32 + lr;
        if (m >= M) continue;
        long pix = m;
        if (!lin_out) {
            const int n = m / (OHg * OWg);
            const int rem = m - n * (OHg * OWg);
            const int oi = rem / OWg, oj = rem - oi * OWg;
            pix = ((long)n * d.OH + (oi * cl.ostep + cl.oy0)) * d.OW + (oj * cl.ostep + cl.ox0);
        }
        const long pofs = pix * Cout;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = cbase + (wm * MR + mr) * 32 + 8 * q + 4 * lh;
                if (co >= Cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[mr][nr][4 * q + e];
                if constexpr (!BIG) {           // (the big-tile form is only dispatched without these: hc_conv_gather)
                if (d.pix_scale != nullptr) {   // NormConv2d: rstd_p * (sum_k W p_k - mean_p * sum_k W)  (functional.py:345-349)
                    const float ps = d.pix_scale[pix], pm = d.pix_shift[pix];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < Cout) v[e] = ps * (v[e] - pm * d.ch_coef[co + e]);
                }
                }
                if (has_coef) {                     // inference BatchNorm (scale, shift) / bias, from the LDS table
                    const f32x4 cs = *reinterpret_cast<const f32x4*>(ctab + (co - cbase));
                    const f32x4 cb = *reinterpret_cast<const f32x4*>(ctab + BC + (co - cbase));
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * cs[e] + cb[e];
                }
                if (staged) {
                    float r4[4] = {0.f, 0.f, 0.f, 0.f};
                    if (resid != nullptr) {
                        const u32x2 rv = *reinterpret_cast<const u32x2*>(resid + pofs + co);
                        r4[0] = bf16lo(rv[0]); r4[1] = bf16hi(rv[0]); r4[2] = bf16lo(rv[1]); r4[3] = bf16hi(rv[1]);
                    }
                    if (d.act != 0) {              // (training units: no activation here, the residual order does not matter)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = apply_act(res_after ? v[e] : v[e] + r4[e], d.act, aslope) + (res_after ? r4[e] : 0.f);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += r4[e];
                    }
                    u32x2 o;
                    o[0] = pack_bf16x2(v[0], v[1]);
                    o[1] = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(ost + ((wn * NR + nr) * 32 + lr) * OPITCH + (co - cbase) * 2) = o;
                } else if constexpr (BIG) {          // big tile: staged stores only (hc_conv_gather checks)
                } else if (d.co_split > 0) {         // stacked convolutions: two destinations with their own channel counts
                    const bool second = co >= d.co_split;
                    bf16_t* dp = second ? reinterpret_cast<bf16_t*>(d.dst2) + pix * (Cout - d.co_split) + (co - d.co_split)
                                        : dst + pix * d.co_split + co;
                    u32x2 o;
                    o[0] = pack_bf16x2(v[0], v[1]);
                    o[1] = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(dp) = o;
                } else if ((Cout & 3) == 0) {
                    float r4[4] = {0.f, 0.f, 0.f, 0.f};
                    if (resid != nullptr) {
                        const u32x2 rv = *reinterpret_cast<const u32x2*>(resid + pofs + co);
                        r4[0] = bf16lo(rv[0]); r4[1] = bf16hi(rv[0]); r4[2] = bf16lo(rv[1]); r4[3] = bf16hi(rv[1]);
                    }
                    if (d.act != 0) {              // (training units: no activation here, the residual order does not matter)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = apply_act(res_after ? v[e] : v[e] + r4[e], d.act, aslope) + (res_after ? r4[e] : 0.f);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += r4[e];
                    }
                    u32x2 o;
                    o[0] = pack_bf16x2(v[0], v[1]);
                    o[1] = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(dst + pofs + co) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (co + e >= Cout) continue;
                        float t = v[e];
                        if (resid != nullptr && !res_after) t += bf16_to_f32(resid[pofs + co + e]);
                        if (d.act != 0) t = apply_act(t, d.act, aslope);
                        if (resid != nullptr && res_after) t += bf16_to_f32(resid[pofs + co + e]);
                        dst[pofs + co + e] = f32_to_bf16(t);
                    }
                }
            }
        }
    }
    if (staged) {
        __syncthreads();
        constexpr int CPR = BC / 8;      // 16-byte chunks per staged pixel row
        for (int i = tid; i < BP * CPR; i += NT) {
            const int pl = i / CPR, ch = i - pl * CPR;
            const int m = pbase + pl, co = cbase + ch * 8;
            if (m >= M || co >= Cout) continue;
            long pix = m;
            if (!lin_out) {
                const int n = m / (OHg * OWg);
                const int rem = m - n * (OHg * OWg);
                const int oi = rem / OWg, oj = rem - oi * OWg;
                pix = ((long)n * d.OH + (oi * cl.ostep + cl.oy0)) * d.OW + (oj * cl.ostep + cl.ox0);
            }
            const u32x2 lo = *reinterpret_cast<const u32x2*>(ost + pl * OPITCH + ch * 16);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(ost + pl * OPITCH + ch * 16 + 8);
            const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
            if (!BIG && d.co_split > 0) {
                if (co >= d.co_split) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(d.dst2) + pix * (Cout - d.co_split) + (co - d.co_split)) = v;
                else *reinterpret_cast<u32x4*>(dst + pix * d.co_split + co) = v;
            } else {
                *reinterpret_cast<u32x4*>(dst + pix * Cout + co) = v;
            }
        }
    }
}

template <int MR, int NR, int WM, int WN, int BK, bool FP8 = false, int NS = 2, int MINB = 2>
int launch_cfg(const hc_conv_desc& d, hipStream_t st) {
    constexpr int BC = 32 * MR * WM, BP = 32 * NR * WN;
    // k-loop stages (+ the big-tile form's scratch slots for padding weight pieces) / output staging
    constexpr int smem_k = NS * (BC + BP) * BK * 2 + ((NS > 2 && (BC * BK / 512) % (WM * WN) != 0) ? WM * WN * 1024 : 0), smem_o = BP * (BC * 2 + 8);
    static_assert(smem_k <= 160 * 1024 && smem_o <= 160 * 1024, "LDS budget");
    constexpr int smem_c = (smem_o + 15) / 16 * 16 + 2 * BC * 4;          // output staging + the epilogue's scale / shift table
    static_assert(smem_c <= 160 * 1024, "LDS budget");
    constexpr int smem = smem_k > smem_c ? smem_k : smem_c;
    // bit 0: staged (16-byte coalesced) epilogue stores; bit 1: parity classes fastest in the tile order (sources of at least
    // HC_CONV_CLSFAST MB, default 128; 0 = never)
    static const double clsfast_mb = [] { const char* e = getenv("HC_CONV_CLSFAST"); return e == nullptr ? 128.0 : atof(e); }();
    const int flags = 1 | ((d.nclass > 1 && clsfast_mb > 0 && (double)d.N * d.IH * d.IW * d.srcC * 2.0 >= clsfast_mb * 1e6) ? 2 : 0);
    int maxM = 0;
    for (int c = 0; c < d.nclass; ++c) {
        const int m = d.N * d.cls[c].OHg * d.cls[c].OWg;
        if (m > maxM) maxM = m;
    }
    if (maxM == 0) return HC_OK;
    dim3 grid((maxM + BP - 1) / BP, (d.Cout + BC - 1) / BC, d.nclass);
    auto kern = conv_gather_kernel<MR, NR, WM, WN, BK, FP8, NS, MINB>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    if (d.stats != nullptr && hc_get_deterministic() && (int)grid.x > hc_get_stat_replicas()) return HC_ERR_ARG;
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), smem, st, d, hc_get_stat_replicas(), FP8 ? 0 : flags);
    return hc_launch_status();
}

// Measured and dropped (round 3): a 256 x 256 tile with EIGHT waves (launch_cfg<4, 2, 2, 4>, 128 accumulator registers per lane) under
// this kernel's old __launch_bounds__(512, 2): the bound capped the register file at 128 per lane, the kernel spilled 576 bytes per
// lane and ran at 208 TFLOP/s against 855 (1779 us against 433 us forward on the 1280-channel layers).  The big tile that works is
// the four-wave one above (NS = 4, MINB = 1: 512 registers per lane, 256 of them accumulators).
template <int BK>
int launch_bk(const hc_conv_desc& d, hipStream_t st) {
    const int C = d.Cout;
    // channel tile: smallest padding waste, prefer the widest tile on ties
    if (C <= 64) return launch_cfg<1, 2, 2, 2, BK>(d, st);
    if (C <= 96) return launch_cfg<3, 1, 1, 4, BK>(d, st);
    // ... unless the launch would leave most of the chip idle: a layer with few pixels (YOLOv4's 19 x 19 and 38 x 38 maps at batch 16:
    // 46 / 181 pixel tiles) has fewer 128-channel tiles than the 512 workgroup slots of the chip - 64-channel tiles double the
    // workgroups (below `fill` = 400 tiles).  YOLOv4 608^2 batch 16, same box: 29.19 ms per step without
    // the rule, 29.08 / 28.68 / 28.75 with n = 256 / 400 / 600
    static const int fill = [] { const char* e = getenv("HC_CONV_FILL"); return e == nullptr ? 400 : atoi(e); }();
    if (fill > 0 && C % 64 == 0) {
        long maxM = 0;
        for (int c = 0; c < d.nclass; ++c) {
            const long m = (long)d.N * d.cls[c].OHg * d.cls[c].OWg;
            if (m > maxM) maxM = m;
        }
        const long tiles = ((maxM + 127) / 128) * ((C + 127) / 128) * d.nclass;
        if (tiles < fill) return launch_cfg<1, 2, 2, 2, BK>(d, st);
    }
    const int w128 = ((C + 127) / 128) * 128 - C, w192 = ((C + 191) / 192) * 192 - C;
    if (w192 <= w128) return launch_cfg<3, 2, 2, 2, BK>(d, st);
    return launch_cfg<2, 2, 2, 2, BK>(d, st);
}

// SHORT k-loops (1 x 1 convolutions over at most 256 channels: 2-8 k32 steps): a workgroup's life is its prologue, two to eight
// steps and the epilogue (statistics, staged stores), none of which overlap inside one workgroup.  Four co-resident workgroups per CU
// (k32 stages: 24-32 KB of LDS each, at most 128 registers) cover one another's latencies instead of two.
int launch_short(const hc_conv_desc& d, hipStream_t st) {
    const int C = d.Cout;
    // (k64 steps - 128-byte rows, which the L2 serves at 53 B/clk/CU against 28 for the 64-byte rows of a k32 step,
    // scripts/probes/fill_probe.hip - fit only three workgroups per CU and measured SLOWER: YOLOv4 26.26 -> 26.84 ms, rexnet1_0x
    // 17.87 -> 18.00 ms, same box, two pairs.  Occupancy is what covers these launches, not the row width.)
    // <= 32 output channels: a 32-channel x 256-pixel tile (the 64-channel tile multiplies 32 rows of zero-filled weights and idles half
    // of the staged store loop).  HC_CONV_C32=0: the 64-channel tile (A/B)
    static const int c32 = [] { const char* e = getenv("HC_CONV_C32"); return e == nullptr ? 1 : atoi(e); }();
    // (k64 stages - 128-byte rows - for the multi-tap launches of this form, per shape: 64@304 -> 32 3 x 3 175 -> 236 us, 64@152 3 x 3 69 -> 82 us,
    // 128@76 3 x 3 47 -> 50 / 68 us with 64- / 128-channel tiles: worse on every shape, profiles/r06_dispatch_by_shape.txt)
    if (c32 && C <= 32) return launch_cfg<1, 2, 1, 4, 32, false, 2, 4>(d, st);
    if (C <= 64) return launch_cfg<1, 2, 2, 2, 32, false, 2, 4>(d, st);
    if (C % 128 != 0 && C % 64 == 0) return launch_cfg<1, 2, 2, 2, 32, false, 2, 4>(d, st);
    return launch_cfg<2, 2, 2, 2, 32, false, 2, 4>(d, st);
}

// LONG k-loops on FEW tiles (YOLOv4's 3 x 3 layers over 256-1024 channels on 19 x 19 / 38 x 38 maps at batch 16: 72-288 k32 steps, 46-181
// pixel tiles): the classic form keeps ONE step of DMA in flight, and with one or two workgroups on a CU a step then costs an L2 / Infinity
// Cache round trip (~2000 clocks per k64 step measured on 1024@19 -> 512: 135 us, 0.16 of peak) however small the tile's MFMA work is
// (256 clocks).  Same tiles, same two workgroups per CU, but the big-tile form's loop: stages issued NS - 1 steps ahead with counted
// waits, pieces inside the MFMA stream.
// Per-shape table (profiles/r06_deep_by_shape.txt, us per launch, YOLOv4 batch 16): 1024@19 -> 512 3 x 3 135 -> 92, 512@19 -> 512 3 x 3 73 -> 53,
// 256@38 -> 512@19 46 -> 35, 2048@19 -> 512 1 x 1 46 -> 37 - the launches of at most 512 such workgroups.  With more (256@38 -> 256 3 x 3:
// 724) the 74 KB of three k64 stages make two rounds of what the classic form (49 KB: three per CU) runs in one: 48 -> 53-57 us in this
// form, as in every variant tried for them (128 x 128 tiles with four k32 stages, two per CU: 53; 128 x 64 with four k32 stages, three per
// CU: 57; 128 x 128 with three k64 stages, one per CU: 77) - those launches keep the classic form (-1).
int launch_deep(const hc_conv_desc& d, hipStream_t st) {
    const int C = d.Cout;
    long maxM = 0;
    for (int c = 0; c < d.nclass; ++c) {
        const long m = (long)d.N * d.cls[c].OHg * d.cls[c].OWg;
        if (m > maxM) maxM = m;
    }
    const long wg64 = ((maxM + 127) / 128) * ((C + 63) / 64) * d.nclass;
    if ((C <= 64 || C % 64 == 0) && wg64 <= 512) return launch_cfg<1, 2, 2, 2, 64, false, 3, 2>(d, st);
    return -1;
}

// fp8 inference: the descriptor arrives with srcC in channels (= bytes); the kernel sees it in 2-byte units
int launch_fp8(const hc_conv_desc& din, hipStream_t st) {
    hc_conv_desc d = din;
    d.srcC = din.srcC / 2;
    const int C = d.Cout;
    if (C <= 64) return launch_cfg<1, 2, 2, 2, 32, true>(d, st);
    if (C <= 96) return launch_cfg<3, 1, 1, 4, 32, true>(d, st);
    const int w128 = ((C + 127) / 128) * 128 - C, w192 = ((C + 191) / 192) * 192 - C;
    if (w192 <= w128) return launch_cfg<3, 2, 2, 2, 32, true>(d, st);
    return launch_cfg<2, 2, 2, 2, 32, true>(d, st);
}

}  // namespace

extern "C" int hc_conv_gather(const hc_conv_desc* dp, hc_stream_t stream) {
    if (dp == nullptr) return HC_ERR_ARG;
    const hc_conv_desc& d = *dp;
    if (d.src0 == nullptr || d.wpk == nullptr || d.dst == nullptr) return HC_ERR_ARG;
    if (d.nclass < 1 || d.nclass > 4 || d.srcC <= 0 || (d.srcC % 16) != 0 || d.Cout <= 0) return HC_ERR_ARG;
    if ((double)d.N * d.IH * d.IW * d.srcC * 2.0 >= 4294967280.0) return HC_ERR_ARG;
    if ((double)d.Cout * d.T * d.srcC * 2.0 >= 4294967280.0) return HC_ERR_ARG;
    for (int c = 0; c < d.nclass; ++c) {
        if (d.cls[c].ntaps < 0 || d.cls[c].ntaps > HC_MAX_TAPS) return HC_ERR_ARG;
        for (int t = 0; t < d.cls[c].ntaps; ++t)
            if (((d.cls[c].tap[t] >> 16) & 0xff) != 0 && d.src1 == nullptr) return HC_ERR_ARG;
    }
    if (d.co_split != 0) {        // stacked convolutions
        if (d.co_split < 0 || d.co_split >= d.Cout || (d.co_split % 4) != 0 || (d.Cout % 4) != 0 || d.dst2 == nullptr) return HC_ERR_ARG;
        if ((d.stats == nullptr) != (d.stats2 == nullptr)) return HC_ERR_ARG;
        if (d.resid != nullptr || d.bias != nullptr || d.act != 0 || d.pix_scale != nullptr || d.ch_mult != nullptr) return HC_ERR_ARG;
    }
    if (d.ch_scale != nullptr && (d.bias == nullptr || d.co_split != 0 || d.pix_scale != nullptr || d.ch_mult != nullptr)) return HC_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // narrow 1 x 1 convolutions: the weight-stationary streaming kernel (conv_pointwise.hip).  HC_CONV_PW=0 off; 1: launches WITHOUT
    // statistics whose output is at least twice as wide as the input (the data gradients of the projections: store-heavy, 1.4-2.7 TB/s
    // in the gather form); 3: those with statistics too (the expansions forward); 2: everything the kernel supports; 4: = 1 without the
    // narrow-source rule below (A/B)
    static const int pw = [] { const char* e = getenv("HC_CONV_PW"); return e == nullptr ? 1 : atoi(e); }();
    // ... and of the launches WITH statistics (the expansions forward) those whose source has 16 or 48 channels on a large map: the gather
    // form stages them in k16 steps (32-byte rows: 14 B/clk/CU of fill).  Per shape (profiles/r06_dispatch_by_shape.txt): 16@112 -> 192
    // 687 -> 389 us, 48@56 -> 192 210 -> 125, 48@56 -> 256 156 -> 133, 48@28 -> 512 95 -> 76; every other expansion loses in this kernel
    // (32@56 -> 192 97 -> 115 from the short-loop form, 80..128@14 -> 448..768 +50-100 %), which is why HC_CONV_PW=3 measured neutral
    auto narrow_stats = [&]() {
        return d.stats != nullptr && d.nclass == 1 && d.srcC % 32 != 0 && d.srcC <= 64 && (long)d.N * d.cls[0].OHg * d.cls[0].OWg >= 150000;
    };
    if (pw > 0 && hc_conv_pointwise_supported(dp) &&
        (pw == 2 || (d.Cout >= 2 * d.srcC && (pw == 3 || d.stats == nullptr || (pw == 1 && narrow_stats())))))
        return hc_conv_pointwise(dp, stream);
    if (d.ch_mult != nullptr) {   // fp8 inference path
        if ((d.srcC % 64) != 0 || (d.Cout % 4) != 0 || d.stats != nullptr || d.resid != nullptr || d.pix_scale != nullptr) return HC_ERR_ARG;
        if ((double)d.N * d.IH * d.IW * d.srcC >= 4294967280.0) return HC_ERR_ARG;
        return launch_fp8(d, st);
    }
    // SHORT-LOOP form for multi-tap launches whose longest parity class has at most 36 k32 steps (taps x channels <= 1152: the 3 x 3
    // layers over <= 128 channels, the stride-2 data gradients of the 304^2 / 608^2 maps): YOLOv4 26.36 -> 26.10 ms and 26.3 -> 25.75 ms
    // on two boxes (taps x channels <= 640: 26.14; <= 2304: 25.9; <= 4608: 26.35 = none); headline and rexnet1_0x unchanged.  It is
    // asked BEFORE the big tile: on 128@76 x 76 (3 x 3, batch 16) the big tile was dispatched at 61 / 53 us forward / data gradient
    // where this form takes 48 / 45 us (per-shape table, profiles/r06_dispatch_by_shape.txt); 3 x 3 over 256 channels loses in this
    // form (256@38: 49 -> 66 us).  HC_CONV_SHORT_FIRST=0 restores the old order.
    static const int short_gate = [] { const char* e = getenv("HC_CONV_SHORT"); return e == nullptr ? 1024 : atoi(e); }();
    static const int short_taps = [] { const char* e = getenv("HC_CONV_SHORT_TAPS"); return e == nullptr ? 1152 : atoi(e); }();
    static const int short_first = [] { const char* e = getenv("HC_CONV_SHORT_FIRST"); return e == nullptr ? 1 : atoi(e); }();
    static const double short_work_mt = [] { const char* e = getenv("HC_CONV_SHORT_WORK_MT"); return e == nullptr ? 11.0e6 : atof(e); }();
    auto short_multitap = [&]() {
        if (!(short_gate && d.srcC % 32 == 0 && d.co_split == 0 && d.pix_scale == nullptr && d.Cout % 8 == 0)) return false;
        int mt = 0;
        double work = 0.0;
        for (int c = 0; c < d.nclass; ++c) {
            mt = d.cls[c].ntaps > mt ? d.cls[c].ntaps : mt;
            work += (double)d.N * d.cls[c].OHg * d.cls[c].OWg * d.Cout;
        }
        // ... and up to twice that loop length when the launch is large (>= 11 M outputs, as for the 1 x 1 rule below): 3 x 3 over 256
        // channels, 256@38 -> 512: 85 -> 79 us, 256@76 -> 128: 83 -> 74 us (both from the big tile); 256@38 -> 256 stays (49 -> 66 us)
        return mt > 1 && (mt * d.srcC <= short_taps || (mt * d.srcC <= 2 * short_taps && work >= short_work_mt));
    };
    if (short_first && short_multitap()) return launch_short(d, st);
    // Big-tile form (256 x 256, see the kernel): one workgroup per CU, so it only pays when the tile count fills whole rounds of the
    // 256 CUs (RepVGG-A0's 1280-channel layers at batch 256: 49 x 5 = 245 tiles).  HC_CONV_BIG=0 sends these layers back to the
    // 128 x 128 tile (same-box A/B), =2 picks the four-wave form.  Measured on those layers (scripts/bench_block1280.py, us per
    // launch: 3x3 192 -> 1280 / 3x3 1280 -> 1280 / 1x1 1280 -> 1280 / data gradient 1280 -> 1280):
    //     128 x 128, 2 workgroups per CU (round 2)            95 / 382 / 68 / 437     MFMA busy 0.46
    //     256 x 256, 4 waves of 128 x 128, DMA as a burst    108 / 405 / 82 / 456     0.42   (64-byte rows, 3 steps ahead)
    //     ... DMA pieces inside the MFMA stream              112 / 360 / 86 / 403     0.48
    //     ... 128-byte rows (BK = 64), 1 step ahead          101 / 377 / 78 / 432     0.44
    //     256 x 256, 8 waves of 128 x 64 (default)            78 / 337 / 61 / 384     0.53
    //     ... 128-byte rows (BK = 64), 1 step ahead           80 / 349 / 63 / 398
    // The same pipelined loop on the 128 x 128 tile with two workgroups per CU: YOLOv4 530.5 -> 523 img/s, RepVGG-A0 neutral - there the
    // co-resident workgroup already covers the burst, and the compiler's schedule of 4 reads + 4 MFMAs needs no skew.  Not kept.
    // Knock-outs of the four-wave form: MFMAs + barriers alone 235 us (0.68 busy at 2.32 GHz), + fragment reads 295, + DMA 326, all
    // three 373 at 2.05 GHz - with one wave per SIMD every LDS / DMA instruction is issue time the matrix pipe waits out, and the
    // chip clocks down 12 % under the full mix.  Two waves per SIMD issue under each other's MFMAs; halving the bytes per flop is
    // what the big tile adds on top of that.
    // The big-tile FAMILY (round 4): the same eight-wave pipelined kernel (a wave = 128 channels x 64 pixels) on 256 x 256 tiles for
    // wide layers - a ragged last channel tile is allowed while at most 15 % of the tile rows are padding - and on 128 x 512 tiles for
    // 97-128 output channels.  One workgroup per CU: it pays when the tiles fill the 256 CUs in whole rounds or in many rounds, so the
    // predicate is the EFFICIENCY of the launch, (useful channels / tile channels) x (useful pixels / tile pixels) x (tiles / CU
    // slots of the rounds they take), against `big_eff` = 0.70.  Measured per shape against the 128 x 128 form
    // (scripts/bench_bigtile.py, forward + statistics, us):
    //     128 @ 76 x 76 batch 16 (eff 0.71)   54.0 -> 51.8      256 -> 512 @ 38 x 38 batch 16 (0.71)   83.4 -> 76.8
    //     128 @ 28 x 28 batch 256 (0.77)     104.3 -> 98.6      256 @ 14 x 14 batch 256 (0.77)         88.6 -> 73.6
    //     1280 @ 7 x 7 batch 256 (0.96)      383.9 -> 337.6     256 @ 38 x 38 batch 16 (0.35)          48.0 -> 73.8 (not dispatched)
    // Tiles with 96-channel waves (96 x 512, 192 x 256, 384 x 128) were built and measured too: 0.82-0.97 of the 128 x 128 form
    // at the same efficiencies (6 MFMAs per 5 fragment reads instead of 8 per 6) - not kept.
    // HC_CONV_BIG=0 switches the family off (A/B).  (The four-wave 256 x 256 form of round 3 - 0.42-0.48 of the matrix pipe against
    // 0.53 - is retired: git show 067617d:holocron_amd/csrc/conv_gather.hip.)
    static const int big = [] { const char* e = getenv("HC_CONV_BIG"); return e == nullptr ? 1 : atoi(e); }();
    constexpr double big_eff = 0.70;
    if (big && d.nclass == 1 && d.co_split == 0 && d.pix_scale == nullptr && d.srcC % 32 == 0 && d.Cout % 8 == 0) {
        const long M = (long)d.N * d.cls[0].OHg * d.cls[0].OWg;
        const int S = d.cls[0].ntaps * (d.srcC / 32);
        constexpr int staged = 1;
        const int C = d.Cout;
        const int bc = (C > 96 && C <= 128) ? 128 : 256, bp = bc == 128 ? 512 : 256;
        const long ct = (C + bc - 1) / bc, pt = (M + bp - 1) / bp, tiles = ct * pt, rounds = (tiles + 255) / 256;
        const double ceff = (double)C / (double)(ct * bc);
        const double eff = ceff * ((double)M / (double)(pt * bp)) * ((double)tiles / (double)(rounds * 256));
        if (staged && S >= 16 && ceff >= 0.85 && eff >= big_eff) {
            // (128-byte rows - BK = 64 - leave room for two stages only, one step ahead: YOLOv4 26.81 / 26.83 against 26.78 / 26.88 ms,
            // the headline 10.33 against 10.35 ms on one box: no difference, as in round 3.  Three stages of 128-byte rows need a
            // 256 x 128 tile whose waves read 1.25 fragments per MFMA instead of 0.75.)
            if (bc == 128) return launch_cfg<4, 2, 1, 8, 32, false, 4, 1>(d, st);
            return launch_cfg<4, 2, 2, 4, 32, false, 4, 1>(d, st);
        }
    }
    // HC_CONV_SHORT=n: 1 x 1 convolutions over at most n channels MAY take the short-loop form (0 = never).  Which do is decided by
    // the per-shape tables of one YOLOv4 and one rexnet1_0x step, each on one box (profiles/r06_short_form_by_shape.txt):
    //   <= 128 channels: always (31 -> 24 us on 128@76 -> 128, 222 -> 150 us on 64@304 -> 128, 196 -> 97 us on 32@56 -> 192);
    //   129-256 channels: when the launch has >= 11 M output elements (HC_CONV_SHORT_WORK) - wins on YOLOv4's 76 x 76 maps, on
    //     256@38 -> 512 (34 -> 26 us) and on rexnet's 160 / 192@7 -> 896..1088 (50 -> 28 us), LOSES on 256@38 -> 256 (18 -> 26 us);
    //   above: only with <= 128 output channels (rexnet's 14 x 14 projections: 51 -> 44 us on 704 -> 128); 512 / 1024 -> 256..1024 on
    //     19 x 19 / 38 x 38 maps lose 30-40 % (1024@19 -> 512: 28 -> 40 us): 16-32 k32 steps are a loop the classic form's 128-byte rows
    //     feed better, and those grids have too few tiles for four workgroups per CU to matter.
    // HC_CONV_SHORT_ALL=1: every 1 x 1 launch up to n channels (the rule before this table)
    static const int short_on = [] { const char* e = getenv("HC_CONV_SHORT"); return e == nullptr ? 1024 : atoi(e); }();
    static const double short_work = [] { const char* e = getenv("HC_CONV_SHORT_WORK"); return e == nullptr ? 11.0e6 : atof(e); }();
    static const bool short_all = [] { const char* e = getenv("HC_CONV_SHORT_ALL"); return e != nullptr && atoi(e) != 0; }();
    if (short_on && d.nclass == 1 && d.cls[0].ntaps == 1 && d.srcC % 32 == 0 && d.srcC <= short_on && d.co_split == 0 && d.pix_scale == nullptr &&
        d.Cout % 8 == 0) {
        const double work = (double)d.N * d.cls[0].OHg * d.cls[0].OWg * d.Cout;
        if (short_all || d.srcC <= 128 || (d.srcC <= 256 ? work >= short_work : d.Cout <= 128)) return launch_short(d, st);
    }
    if (!short_first && short_multitap()) return launch_short(d, st);
    // HC_CONV_DEEP=0 switches the deep-pipeline loop on the classic tile off (A/B); it is asked for launches with at least 16 k32 steps in
    // every parity class (HC_CONV_DEEP_S; 32: 10 109 against 10 015 us of gather-conv per YOLOv4 step)
    static const int deep = [] { const char* e = getenv("HC_CONV_DEEP"); return e == nullptr ? 1 : atoi(e); }();
    static const int deep_s = [] { const char* e = getenv("HC_CONV_DEEP_S"); return e == nullptr ? 16 : atoi(e); }();
    if (deep && d.co_split == 0 && d.pix_scale == nullptr && d.srcC % 64 == 0 && d.Cout % 8 == 0) {
        int ms = 1 << 30;
        for (int c = 0; c < d.nclass; ++c) {
            const int sc = d.cls[c].ntaps * (d.srcC / 32);
            ms = sc < ms ? sc : ms;
        }
        if (ms >= deep_s) {
            const int rc = launch_deep(d, st);
            if (rc != -1) return rc;
        }
    }
    // `bk_cap` caps the k-step (a 192-channel tile with 64-channel k-steps stages 2 x 40 KB - two workgroups need the
    // whole 160 KB of LDS)
    constexpr int bk_cap = 64;                      // k-step cap (32 measured 3 % slower on the ReXNet 1 x 1 layers, round 4)
    if (d.srcC % 64 == 0 && bk_cap >= 64) return launch_bk<64>(d, st);
    if (d.srcC % 32 == 0 && bk_cap >= 32) return launch_bk<32>(d, st);
    return launch_bk<16>(d, st);
}
