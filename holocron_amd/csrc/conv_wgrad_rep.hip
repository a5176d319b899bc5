// Both weight gradients of a RepVGG block - dW3 (3x3, pad 1) and dW1 (1x1, pad 0), same stride - in ONE launch, from ONE
// staging of the block input, for up to 16 blocks of the same shape at a time.
//
//   dW3[co][ci][kh][kw] = sum_m dy3[m][co] * x[pix(m) + (kh-1, kw-1)][ci]
//   dW1[co][ci]         = sum_m dy1[m][co] * x[pix(m)][ci]                       m = (n, oy, ox),  pix(m) = (n, s*oy, s*ox)
//
// Replaces the two aten::convolution_backward(weight) calls behind the two nn.Conv2d of a RepBlock
// (holocron/models/classification/repvgg.py:57-60 via models/utils.py:73).
//
// Why one kernel.  The separate kernels (conv_wgrad_tr.hip) launch 3x3 and 1x1 apart (x staged twice), stage through
// registers (no load / MFMA overlap with one workgroup per CU), and need a ~40-way split-K on the 14 x 14 layers whose fp32
// slabs double the HBM traffic of the layer.  Here:
//   * a workgroup owns a (16*MR ci) x (16*NR co) tile of ALL TEN taps.  Its four wave ROLES split the taps, not the tile, 3 + 3 + 2 + 2:
//     kernel row 0, kernel row 1, taps (2,0) (2,1), and tap (2,2) together with the 1x1 against dy1 (the 1x1 input pixel is tap
//     (1,1)).  A wave's taps share their dy fragments, and nobody re-reads a neighbour's x fragments: 2*(3*MR + NR) transposing
//     LDS reads per 3*MR*NR MFMAs instead of 2*(9*MR + NR) per 9*MR*NR/WAVES.  Eight-wave variants (two ci halves or two pixel
//     halves) rotate the roles of the second group by two so that every SIMD carries one 3-tap and one 2-tap wave.
//   * the 96 x 96 tile (MR = NR = 6; the 96- and 192-wide stages, round 5) deals differently: EIGHT waves = two tap groups of five
//     (taps 0-4 against dy3 | taps 5-8 against dy3 + the 1x1 against dy1) x two co halves x two ci halves, 5 x 3 x 3 MFMAs and 180
//     accumulator registers per wave and k32 step (`run5`).  Twice the accumulators per CU halve the L2 -> LDS ingest per MFMA, which
//     bounded the 96 x 48 tile (its DMA alone took 504 of its 744 us on the fourteen 192 @ 14 x 14 blocks; here 295 of 670).
//   * operands stream HBM/L2 -> LDS by DMA (buffer_load ... lds) in their natural NHWC layout, one step (R output rows)
//     ahead of the MFMAs, no staging registers.  The batch is walked as ONE tall image: image n owns virtual input rows
//     [n*PI, (n+1)*PI), PI = s*(OH+1), row 0 = the zero halo row (out-of-range DMA -> zeros), and virtual output rows
//     [n*(OH+1), ...) whose last one is a gap row with dy = 0.  x rows live in a ring of LDS row slots (slot = V mod NSLOT):
//     every input row is fetched once per workgroup, whatever the kernel height, and steps never see an image boundary.
//   * up to 16 blocks of one shape (the 14 identical 192-channel blocks of repvgg_a0) share a launch: blockIdx ->
//     (block, tile, split).  With 14 x 8 tiles the pixel range is split 2 ways instead of 41, the slab traffic drops from
//     ~100% of the layer's input bytes to ~8%, and one reduce launch serves all 28 gradients.
//   * MFMA operands come out of LDS through ds_read_b64_tr_b16 exactly as in conv_wgrad_tr.hip (semantics probed in
//     scripts/probes/tr16_probe.hip); v_mfma_f32_16x16x32_bf16, A = x (rows = ci), B = dy (cols = co), k = 32 pixels.
//
// Output: fp32 slabs [job][split][co][10][ci] -> wrep_reduce_kernel sums the splits in a FIXED order (no atomics: the result
// is bit-reproducible) and writes the reference layouts OIHW [co][ci][3][3] and [co][ci][1][1].
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "../../include/holocron_hip.h"

typedef __attribute__((ext_vector_type(4))) short wr_s16x4;
typedef __attribute__((address_space(3))) wr_s16x4 wr_lds_s16x4;

namespace wrep {

struct Args {
    const void* x[HC_WREP_MAX_JOBS];
    const void* dy3[HC_WREP_MAX_JOBS];
    const void* dy1[HC_WREP_MAX_JOBS];
    float* ws;
    int N, IH, IW, Cin, OH, OW, Cout, s;
    int njobs, nsplit, steps_per_split, total_steps;
    int n_ci, n_co;             // tiles
    int R, P, P32;              // output rows per step, pixels per step, padded to 32
    int PO, PI;                 // virtual rows per image: output (OH + 1), input (s * PO)
    int XW, SX, XJ, ROWB;       // staged pixels per x row, LDS pixel stride, DMA instructions / bytes per row
    int NSLOT, PF;              // ring slots, steps in flight
    int SD, DJ, DHALF, DSLOT;   // dy: LDS pixel stride, DMA instructions per tensor, bytes per tensor / per stage
    int off_dy, off_tab, off_sink;
    int nd;                     // DMA instructions per wave and step (uniform: padded with zero-fill dummies when PF > 1)
    int dbg;                    // experiment knob HC_WREP_DBG: 1 = no MFMA phase, 2 = no DMA, 4 = no LDS fragment reads
};

__device__ __forceinline__ wr_s16x4 tr_read(const char* lds_base, int off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((wr_lds_s16x4*)(lds_base + off));
}
__device__ __forceinline__ bf16x8 tr_pair(const char* lds_base, int o0, int o1) {
    const wr_s16x4 lo = tr_read(lds_base, o0), hi = tr_read(lds_base, o1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__device__ __forceinline__ void wait_vm(int n) {   // s_waitcnt vmcnt(n) for a wave-uniform runtime n
    switch (n) {
#define WR_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        WR_W(0) WR_W(1) WR_W(2) WR_W(3) WR_W(4) WR_W(5) WR_W(6) WR_W(7) WR_W(8) WR_W(9) WR_W(10) WR_W(11) WR_W(12) WR_W(13)
        WR_W(14) WR_W(15) WR_W(16) WR_W(17) WR_W(18) WR_W(19) WR_W(20) WR_W(21) WR_W(22) WR_W(23) WR_W(24) WR_W(25) WR_W(26)
        WR_W(27) WR_W(28) WR_W(29) WR_W(30) WR_W(31) WR_W(32)
#undef WR_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

constexpr int XJW = 4;   // x DMA instructions per row and wave (rows up to 12 KB)
constexpr int DJW = 4;   // dy DMA instructions per tensor and wave (stages up to 16 KB per tensor)

// HV = 2: eight waves, the ci tile split over two groups of four waves (two waves per SIMD at <= 256 registers: while one waits for
// LDS or issues DMA the other feeds the matrix pipe - with one wave per SIMD nothing overlaps unless the instruction stream says so)
// PV = 2 (the 48 x 48 tile, whose ci blocks do not split evenly): eight waves, the two groups take alternate halves of the 32-pixel
// k-steps of every step and write their own partial slab (the reduction kernel adds twice as many) - same reason as HV = 2.
// BIGR (MR = NR = 6: the 96 ci x 96 co x 10-tap tile, round 5): EIGHT waves, two per SIMD, 180 accumulator registers each.  Roles are
// (tap group of five) x (co half) x (ci half): group 0 = taps (0,0) (0,1) (0,2) (1,0) (1,1) against dy3, group 1 = (1,2) (2,0) (2,1) (2,2)
// against dy3 + the 1x1 against dy1.  Twice the accumulators per CU of the 96 x 48 tile (which uses 42 % of the register file for them):
// 18 KB of operands per k32 step for 360 MFMAs instead of 12.3 KB for 180 - the L2 -> LDS fill, not the matrix pipe, bounds the smaller
// tile.  The first form - four waves at one per SIMD, 360 accumulators each - compiled to 584 v_accvgpr_write + 124
// v_accvgpr_read per 90 MFMAs (the accumulators do not fit the 256 AGPRs and the allocator shuttles them) and ran at 26 % of the matrix rate.
template <int MR, int NR, int HV, int PV = 1>
__global__ __launch_bounds__(256 * HV * PV, (PV == 1 && (HV == 2 || MR * NR <= 9)) ? 2 : 1) void wrep_kernel(const Args a) {
    constexpr int NW = 4 * HV * PV, NT_ = 256 * HV * PV, MH = MR / HV;
    constexpr bool BIGR = MR == 6 && NR == 6;
    constexpr int NRW = BIGR ? NR / 2 : NR;                // co blocks of one wave
    static_assert(!BIGR || (HV == 2 && PV == 1), "the 96 x 96 tile runs on eight waves");
    static_assert(HV == 1 || PV == 1, "one kind of wave-group split at a time");
    static_assert(MR % HV == 0, "ci tile must split evenly over the wave groups");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = (wid + ((NW == 8 && (wid & 4)) ? 2 : 0)) & 3;   // which taps (see `run`); the second wave group is rotated by two
    const int half = HV == 2 ? wid >> 2 : 0;               // which MH ci blocks of the tile
    const int pz = PV == 2 ? wid >> 2 : 0;                 // which k-steps of a step

    // ---- blockIdx -> (job, split, tile): the tiles of one (job, split) - same pixels, different channels - share an XCD's L2
    const int NT = a.n_ci * a.n_co;
    const int xcd = blockIdx.x & 7, rr_ = blockIdx.x >> 3;
    const int tile = rr_ % NT, u = (rr_ / NT) * 8 + xcd;
    if (u >= a.njobs * a.nsplit) return;
    const int job = u / a.nsplit, split = u - job * a.nsplit;
    const int cit = tile / a.n_co, cot = tile - cit * a.n_co;
    const int ci0 = cit * 16 * MR, co0 = cot * 16 * NR;

    const int s = a.s;
    const int step0 = split * a.steps_per_split;
    int step1 = step0 + a.steps_per_split;
    if (step1 > a.total_steps) step1 = a.total_steps;
    const int nsteps = step1 - step0;
    const int Ua = step0 * a.R, Ub = step1 * a.R;          // virtual output rows of this split

    // buffer descriptors in SGPRs: the pointers are indexed by `job` (a quotient, i.e. computed on the vector ALU) - make the
    // uniformity explicit or the inline-asm DMA may be handed a VGPR descriptor
    auto uniform_rsrc = [](const void* p, unsigned bytes) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        u32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((unsigned)v);
        r[1] = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
        r[2] = __builtin_amdgcn_readfirstlane(bytes);
        r[3] = 0x00020000u;
        return r;
    };
    const u32x4 rsx = uniform_rsrc(a.x[job], (unsigned)a.N * a.IH * a.IW * a.Cin * 2u);
    const u32x4 rs3 = uniform_rsrc(a.dy3[job], (unsigned)a.N * a.OH * a.OW * a.Cout * 2u);
    const u32x4 rs1 = uniform_rsrc(a.dy1[job], (unsigned)a.N * a.OH * a.OW * a.Cout * 2u);
    const unsigned lds0 = hc_lds_addr(smem);

    // ---- pixel table (step-invariant): {s * row, s * col * SX} of step pixel p; padding pixels alias pixel 0 (their dy is 0)
    int* tab = reinterpret_cast<int*>(smem + a.off_tab);
    for (int p = tid; p < a.P32; p += NT_) {
        int r = 0, c = 0;
        if (p < a.P) { r = p / a.OW; c = p - r * a.OW; }
        tab[2 * p] = r * s;
        tab[2 * p + 1] = c * s * a.SX;
    }

    // ---- DMA lane constants ----------------------------------------------------------------------------------------
    // x row: LDS slot q = 64 j + lane of the row  ->  staged pixel q / S16 (pixel 0 = the left halo column), 16-byte piece q % S16
    const int S16 = a.SX >> 4, CI16 = 2 * MR;
    unsigned xsrc[XJW];
#pragma unroll
    for (int jj = 0; jj < XJW; ++jj) {
        const int j = wid + NW * jj;
        const int q = 64 * j + lane;
        const int px = q / S16, sub = q - px * S16;
        const int ix = px - 1;
        const bool ok = (j < a.XJ) && (px < a.XW) && (sub < CI16) && (ix >= 0) && (ix < a.IW);
        xsrc[jj] = ok ? (unsigned)(ix * a.Cin + ci0) * 2u + (unsigned)sub * 16u : HC_OOB;
    }
    // dy stage: slot q = 64 j + lane -> step pixel q / SD16 (row r, column c of the step), piece q % SD16.  In memory the rows of a
    // step are contiguous except that the virtual gap row of an image does not exist: byte offset = step base + dyo - (rows past
    // the gap ? one row : 0)
    const int SD16 = a.SD >> 4, CO16 = 2 * NR;
    const unsigned dyrow = (unsigned)a.OW * (unsigned)a.Cout * 2u;
    int dyr[DJW];                                          // row of the step, -1: padding lane (always zero-filled)
    unsigned dyo[DJW];
#pragma unroll
    for (int jj = 0; jj < DJW; ++jj) {
        const int j = wid + NW * jj;
        const int q = 64 * j + lane;
        const int p = q / SD16, sub = q - p * SD16;
        const bool ok = (j < a.DJ) && (p < a.P) && (sub < CO16);
        const int r = p / a.OW, c = p - r * a.OW;
        dyr[jj] = ok ? r : -1;
        dyo[jj] = (unsigned)r * dyrow + (unsigned)(c * a.Cout + co0) * 2u + (unsigned)sub * 16u;
    }
    const int ndw = (a.DJ - wid + NW - 1) / NW;            // dy DMA instructions of this wave per tensor
    const int nxw = (a.XJ - wid + NW - 1) / NW;            // x DMA instructions of this wave per row

    // ---- running positions (wave-uniform) ----------------------------------------------------------------------
    const int V0 = s * Ua;                                 // first virtual input row to fetch
    int xn = V0 / a.PI, xv = V0 - xn * a.PI;               // ... = row xv of image xn (xv = 0: zero halo, 1..IH: real)
    int xslot = V0 % a.NSLOT;
    int dU = Ua;                                           // first virtual output row of the next step to fetch
    int dn = dU / a.PO, doy = dU - dn * a.PO;
    int dstage = 0;

    auto issue_x_rows = [&](int nrows) {
        for (int k = 0; k < nrows; ++k) {
            const bool real = (xv >= 1) && (xv <= a.IH) && (xn < a.N);
            const unsigned rowbase = real ? (unsigned)((xn * a.IH + (xv - 1)) * a.IW) * (unsigned)a.Cin * 2u : HC_OOB;
            const unsigned dst = lds0 + (unsigned)xslot * (unsigned)a.ROWB + (unsigned)wid * 1024u;
#pragma unroll
            for (int jj = 0; jj < XJW; ++jj)
                if (jj < nxw)      // HC_OOB + anything stays out of range: unsigned saturation is not needed, the range check is on the sum
                    hc_dma16(rsx, __builtin_amdgcn_readfirstlane(dst + (unsigned)(jj * NW) * 1024u),
                             (real && xsrc[jj] != HC_OOB) ? rowbase + xsrc[jj] : HC_OOB);
            if (++xv == a.PI) { xv = 0; ++xn; }
            if (++xslot == a.NSLOT) xslot = 0;
        }
    };
    auto issue_dy = [&]() {
        const unsigned dst = lds0 + (unsigned)a.off_dy + (unsigned)dstage * (unsigned)a.DSLOT + (unsigned)wid * 1024u;
        // rows [0, rgap) of the step belong to image dn, row rgap is its gap row, later rows to image dn + 1; rows >= rend are past
        // the end of the split / of the batch
        const int rgap = a.OH - doy;                       // may be negative (the step starts on a gap row: doy == OH) or >= R
        int rend = Ub - dU;
        const int rbatch = (a.N - dn) * a.PO - doy;
        if (rbatch < rend) rend = rbatch;
        const unsigned base = (unsigned)((dn * a.OH + doy) * a.OW) * (unsigned)a.Cout * 2u;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
#pragma unroll
            for (int jj = 0; jj < DJW; ++jj)
                if (jj < ndw) {
                    const int r = dyr[jj];
                    const bool ok = (r >= 0) && (r < rend) && (r != rgap);
                    const unsigned off = base + dyo[jj] - (r > rgap ? dyrow : 0u);
                    hc_dma16(which == 0 ? rs3 : rs1,
                             __builtin_amdgcn_readfirstlane(dst + (unsigned)which * (unsigned)a.DHALF + (unsigned)(jj * NW) * 1024u),
                             ok ? off : HC_OOB);
                }
        }
        dU += a.R;
        doy += a.R;
        if (doy >= a.PO) { doy -= a.PO; ++dn; }
        if (++dstage > a.PF) dstage = 0;
    };

    // ---- fragment lane constants -------------------------------------------------------------------------------------
    const int la = lane & 15, kq = lane >> 4;
    const int prow = 4 * kq + (la >> 2);                   // k-slot -> pixel map of conv_wgrad_tr.hip (same for both operands)
    const int cq2 = 8 * (la & 3);                          // byte offset of this lane's channel quad in a 16-channel block
    const int SX1 = a.SX, SX2 = 2 * a.SX;
    const int SD16B = 16 * a.SD;
    const int nk = (a.dbg & 1) ? 0 : (a.P32 >> 5);
    const int RS = a.R * s;
    const int ndma = RS * nxw + 2 * ndw;                   // DMA instructions of this wave per step

    // ---- prologue: the first window (rows of step 0 incl. halo) and PF steps of dy; each later step adds R*s rows ----
    issue_x_rows((a.R - 1) * s + 3);
    issue_dy();
    if (a.PF > 1) {
        issue_x_rows(RS);
        issue_dy();
    }

    // The step loop, once per wave ROLE.  The ten taps are dealt 3 + 3 + 2 + 2: role 0 = kernel row 0, role 1 = kernel row 1,
    // role 2 = taps (2,0), (2,1), role 3 = tap (2,2) against dy3 plus the 1x1 against dy1 - and the second wave group takes the roles
    // rotated by two, so that every SIMD hosts one 3-tap and one 2-tap wave (45 MFMAs per k-step each; with a 3 + 3 + 3 + 1 deal the
    // two 1-tap waves shared one SIMD and the other three carried 54).  One instantiation per role instead of a branch inside the
    // loop: with one accumulator array live across `if (role)` the register allocator copied the accumulators at every join
    // (438 v_accvgpr_mov per k-step, 5 VALU per MFMA in the PMC counters).
    auto run = [&](auto kh_c, auto kw0_c, auto n3_c, auto has1_c) {
        constexpr int KH = decltype(kh_c)::value, KW0 = decltype(kw0_c)::value, N3 = decltype(n3_c)::value, HAS1 = decltype(has1_c)::value;
        constexpr int TAPS = N3 + HAS1;
        f32x4 acc[TAPS][MH][NR];
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int m = 0; m < MH; ++m)
#pragma unroll
                for (int q = 0; q < NR; ++q) acc[t][m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int kwoff = 32 * MH * half;                  // first ci block of this wave
        int base = (s * Ua) % a.NSLOT;                     // ring slot of the first input row of the current step
        int cstage = 0;
        for (int t = 0; t < nsteps; ++t) {
            wait_vm(a.PF > 1 ? ndma : 0);                  // everything but the newest step in flight has landed
            __syncthreads();                               // step t visible to all waves; compute(t-1) finished everywhere
            if (!(a.dbg & 2)) {
                issue_x_rows(RS);                          // step t + PF (past the end of the split: harmless real / zero rows)
                issue_dy();
            }
            const char* dyb = smem + a.off_dy + cstage * a.DSLOT;
            for (int g = (pz * nk) / PV; g < ((pz + 1) * nk) / PV; ++g) {
                const int p0 = 32 * g + prow;
                const int r0 = tab[2 * p0], c0 = tab[2 * p0 + 1], r1 = tab[2 * p0 + 32], c1 = tab[2 * p0 + 33];
                const int d0 = p0 * a.SD + cq2;
                if (N3 > 0) {                              // taps (KH, KW0 .. KW0 + N3 - 1) against dy3
                    int s0 = base + r0 + KH, s1 = base + r1 + KH;
                    if (s0 >= a.NSLOT) s0 -= a.NSLOT;
                    if (s1 >= a.NSLOT) s1 -= a.NSLOT;
                    const int xa0 = s0 * a.ROWB + c0 + cq2 + kwoff, xa1 = s1 * a.ROWB + c1 + cq2 + kwoff;
                    bf16x8 fb[NR];
#pragma unroll
                    for (int q = 0; q < NR; ++q) fb[q] = tr_pair(dyb, d0 + 32 * q, d0 + SD16B + 32 * q);
#pragma unroll
                    for (int kw = 0; kw < N3; ++kw) {
                        const int ko = (KW0 + kw) == 0 ? 0 : ((KW0 + kw) == 1 ? SX1 : SX2);
#pragma unroll
                        for (int m = 0; m < MH; ++m) {
                            const bf16x8 fa = tr_pair(smem, xa0 + ko + 32 * m, xa1 + ko + 32 * m);
#pragma unroll
                            for (int q = 0; q < NR; ++q)
                                acc[kw][m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[q], acc[kw][m][q], 0, 0, 0);
                        }
                    }
                }
                if (HAS1) {                                // the 1x1: centre row, centre column, against dy1
                    int s0 = base + r0 + 1, s1 = base + r1 + 1;
                    if (s0 >= a.NSLOT) s0 -= a.NSLOT;
                    if (s1 >= a.NSLOT) s1 -= a.NSLOT;
                    const int xa0 = s0 * a.ROWB + c0 + cq2 + kwoff + SX1, xa1 = s1 * a.ROWB + c1 + cq2 + kwoff + SX1;
                    bf16x8 fb[NR];
#pragma unroll
                    for (int q = 0; q < NR; ++q) fb[q] = tr_pair(dyb + a.DHALF, d0 + 32 * q, d0 + SD16B + 32 * q);
#pragma unroll
                    for (int m = 0; m < MH; ++m) {
                        const bf16x8 fa = tr_pair(smem, xa0 + 32 * m, xa1 + 32 * m);
#pragma unroll
                        for (int q = 0; q < NR; ++q)
                            acc[N3][m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[q], acc[N3][m][q], 0, 0, 0);
                    }
                }
            }
            base += RS;
            if (base >= a.NSLOT) base -= a.NSLOT;
            if (++cstage > a.PF) cstage = 0;
        }
        wait_vm(0);                                        // drain the run-ahead DMA before the workgroup retires

        // ---- slab[job][split][co][10][ci]: lane = co column, the 4 accumulator values = 4 consecutive ci -------------
        float* ws = a.ws + (size_t)((job * a.nsplit + split) * PV + pz) * (size_t)a.Cout * 10u * (size_t)a.Cin;
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int co = co0 + 16 * q + la;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int tap = t < N3 ? 3 * KH + KW0 + t : 9;
                float* row = ws + ((size_t)co * 10 + tap) * a.Cin + ci0 + 16 * MH * half + 4 * kq;
#pragma unroll
                for (int m = 0; m < MH; ++m) *reinterpret_cast<f32x4*>(row + 16 * m) = acc[t][m][q];
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    // ---- BIGR: five taps x six ci blocks x three co blocks per wave ----------------------------------------------------
    auto run5 = [&](auto tg_c) {
        constexpr int TG = decltype(tg_c)::value;
        const int ch = (wid >> 1) & 1;                      // co half of this wave (ci half: `half` = wid >> 2)
        const int kwoff = 32 * MH * half;
        f32x4 acc[5][MH][NRW];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int m = 0; m < MH; ++m)
#pragma unroll
                for (int q = 0; q < NRW; ++q) acc[j][m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        int base = (s * Ua) % a.NSLOT;
        int cstage = 0;
        const int cohalf = ch * NRW * 32;                   // byte offset of the wave's co blocks inside a staged dy pixel
        for (int t = 0; t < nsteps; ++t) {
            wait_vm(a.PF > 1 ? ndma : 0);
            __syncthreads();
            if (!(a.dbg & 2)) {
                issue_x_rows(RS);
                issue_dy();
            }
            const char* dyb = smem + a.off_dy + cstage * a.DSLOT;
            for (int g = 0; g < nk; ++g) {
                const int p0 = 32 * g + prow;
                const int r0 = tab[2 * p0], c0 = tab[2 * p0 + 1], r1 = tab[2 * p0 + 32], c1 = tab[2 * p0 + 33];
                const int d0 = p0 * a.SD + cq2 + cohalf;
                bf16x8 fb[NRW];                             // dy3 fragments; group 1 re-loads them from dy1 in front of its last tap (the 1x1)
#pragma unroll
                for (int q = 0; q < NRW; ++q) fb[q] = tr_pair(dyb, d0 + 32 * q, d0 + SD16B + 32 * q);
                // x addresses of the (up to three) kernel rows this group touches
                int xr0[3], xr1[3];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    int s0 = base + r0 + kh, s1 = base + r1 + kh;
                    if (s0 >= a.NSLOT) s0 -= a.NSLOT;
                    if (s1 >= a.NSLOT) s1 -= a.NSLOT;
                    xr0[kh] = s0 * a.ROWB + c0 + cq2 + kwoff;
                    xr1[kh] = s1 * a.ROWB + c1 + cq2 + kwoff;
                }
                // the 5 x MH (tap, ci block) groups as ONE software pipeline: the x fragment of group i + 1 is requested before the three
                // MFMAs of group i (left to the compiler every group was read -> wait -> 3 MFMAs: ~180 cycles per 48 of matrix work, and the
                // two waves of a SIMD together kept its pipe 53 % busy)
                auto xfrag = [&](const int i) __attribute__((always_inline)) {
                    const int j = i / MH, m = i - j * MH;
                    const bool is1 = TG == 1 && j == 4;
                    const int tap = is1 ? 4 : (TG == 0 ? 0 : 5) + j;
                    const int kh = tap / 3, kw = tap - 3 * kh;
                    const int ko = kw == 0 ? 0 : (kw == 1 ? SX1 : SX2);
                    return tr_pair(smem, xr0[kh] + ko + 32 * m, xr1[kh] + ko + 32 * m);
                };
                bf16x8 fa0 = xfrag(0), fa1;
#pragma unroll
                for (int i = 0; i < 5 * MH; ++i) {
                    const int j = i / MH, m = i - j * MH;
                    if (TG == 1 && j == 4 && m == 0) {          // the 1x1 (centre tap against dy1): swap the dy fragments
#pragma unroll
                        for (int q = 0; q < NRW; ++q) fb[q] = tr_pair(dyb + a.DHALF, d0 + 32 * q, d0 + SD16B + 32 * q);
                    }
                    if (i + 1 < 5 * MH) {
                        if (i & 1) fa0 = xfrag(i + 1); else fa1 = xfrag(i + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < NRW; ++q)
                        acc[j][m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((i & 1) ? fa1 : fa0, fb[q], acc[j][m][q], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            base += RS;
            if (base >= a.NSLOT) base -= a.NSLOT;
            if (++cstage > a.PF) cstage = 0;
        }
        wait_vm(0);
        float* ws = a.ws + (size_t)(job * a.nsplit + split) * (size_t)a.Cout * 10u * (size_t)a.Cin;
#pragma unroll
        for (int q = 0; q < NRW; ++q) {
            const int co = co0 + 16 * (ch * NRW + q) + la;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int tap = (TG == 1 && j == 4) ? 9 : (TG == 0 ? j : 5 + j);
                float* row = ws + ((size_t)co * 10 + tap) * a.Cin + ci0 + 16 * MH * half + 4 * kq;
#pragma unroll
                for (int m = 0; m < MH; ++m) *reinterpret_cast<f32x4*>(row + 16 * m) = acc[j][m][q];
            }
        }
    };
    if (BIGR) {
        if ((wid & 1) == 0) run5(I0{}); else run5(I1{});
        return;
    }
    if (role == 0) run(I0{}, I0{}, I3{}, I0{});
    else if (role == 1) run(I1{}, I0{}, I3{}, I0{});
    else if (role == 2) run(I2{}, I0{}, I2{}, I0{});
    else run(I2{}, I2{}, I1{}, I1{});
}

// dw3[co][ci][t] (t < 9), dw1[co][ci] = (accumulate ? old : 0) + sum_split slab[job][split][co][t][ci], splits added in a fixed
// order.  One workgroup per (co, 64 / SG consecutive ci, job): thread (sg, t, c) sums the splits sg, sg + SG, ... with reads
// contiguous along ci; the SG partial sums are combined in order through LDS and the tile is written as contiguous runs of
// the two OIHW gradients.
template <int SG>
__global__ __launch_bounds__(640) void wrep_reduce_kernel(const float* __restrict__ ws, hc_rep_wgrad_desc d, int nsplit) {
    constexpr int CT = 64 / SG;
    __shared__ float sm[SG][10][CT + 1];
    const int co = blockIdx.x, ci0 = blockIdx.y * CT, job = blockIdx.z;
    const int tid = threadIdx.x;
    const int c = tid % CT, t = (tid / CT) % 10, sg = tid / (CT * 10);
    const size_t slab = (size_t)d.Cout * 10u * (size_t)d.Cin;
    float sum = 0.f;
    if (ci0 + c < d.Cin) {
        const float* p = ws + (size_t)job * nsplit * slab + ((size_t)co * 10 + t) * d.Cin + ci0 + c;
        int k = sg;
        for (; k + 3 * SG < nsplit; k += 4 * SG) {      // four independent loads in flight
            const float v0 = p[(size_t)k * slab], v1 = p[(size_t)(k + SG) * slab], v2 = p[(size_t)(k + 2 * SG) * slab],
                        v3 = p[(size_t)(k + 3 * SG) * slab];
            sum += v0; sum += v1; sum += v2; sum += v3;
        }
        for (; k < nsplit; k += SG) sum += p[(size_t)k * slab];
    }
    sm[sg][t][c] = sum;
    __syncthreads();
    // 9 * CT contiguous floats of dw3 (element j: ci = j / 9, tap = j % 9), then CT of dw1
    for (int j = tid; j < 10 * CT; j += 640) {
        const bool is3 = j < 9 * CT;
        const int cl = is3 ? j / 9 : j - 9 * CT, tt = is3 ? j - cl * 9 : 9;
        if (ci0 + cl >= d.Cin) continue;
        float v = sm[0][tt][cl];
#pragma unroll
        for (int g = 1; g < SG; ++g) v += sm[g][tt][cl];
        float* o = is3 ? d.dw3[job] + ((size_t)co * d.Cin + ci0 + cl) * 9 + tt : d.dw1[job] + (size_t)co * d.Cin + ci0 + cl;
        *o = d.accumulate ? *o + v : v;
    }
}

// LDS pixel stride >= bytes, 16-byte aligned, such that 8 consecutive step pixels (input pixel step = conv stride) land on
// 8 distinct 32-byte slots of the 256-byte bank row (tr_b64 reads; same rule as conv_wgrad_tr.hip)
inline int round_stride(int bytes, int stride) {
    int sx = (bytes + 15) / 16 * 16;
    for (;; sx += 16) {
        bool ok = true;
        unsigned seen = 0;
        for (int i = 0; i < 8 && ok; ++i) {
            const int pos = (i * stride * sx) % 256;
            if (pos % 32) { ok = false; break; }
            const unsigned bit = 1u << (pos / 32);
            if (seen & bit) ok = false;
            seen |= bit;
        }
        if (ok) return sx;
    }
}

struct Plan {
    bool ok;
    Args a;
    int MR, NR, HV, PV, smem, grid;
};

inline Plan make_plan(const hc_rep_wgrad_desc& d) {
    Plan pl{};
    pl.ok = false;
    static const int enable = getenv("HC_WREP") ? atoi(getenv("HC_WREP")) : 1;
    if (!enable) return pl;
    if (d.njobs < 1 || d.njobs > HC_WREP_MAX_JOBS || d.N < 1) return pl;
    const int s = d.stride;
    if (s != 1 && s != 2) return pl;
    if (d.OH != (d.IH + 2 - 3) / s + 1 || d.OW != (d.IW + 2 - 3) / s + 1) return pl;
    if (d.OH != (d.IH - 1) / s + 1 || d.OW != (d.IW - 1) / s + 1) return pl;     // the 1x1 branch's output size
    if ((double)d.N * d.IH * d.IW * d.Cin * 2.0 >= 4294967040.0 || (double)d.N * d.OH * d.OW * d.Cout * 2.0 >= 4294967040.0) return pl;
    int MR = 0, NR = 0;
    constexpr int tile_env = 0;
    // 96 x 96 where both widths allow (RepVGG-A0's 96- and 192-wide stages): half the x re-reads of the 96 x 48 tile per output; same-box
    // step 9.91-10.00 ms against 10.00-10.07 with the 96 x 48 tile everywhere (profiles/r05_wgrad_rep_big_tile.txt)
    if (d.Cin % 96 == 0 && d.Cout % 96 == 0 && tile_env != 33) { MR = 6; NR = 6; }
    else if (d.Cin % 96 == 0 && d.Cout % 48 == 0 && tile_env != 33) { MR = 6; NR = 3; }
    else if (d.Cin % 64 == 0 && d.Cout % 64 == 0) { MR = 4; NR = 4; }
    else if (d.Cin % 48 == 0 && d.Cout % 48 == 0) { MR = 3; NR = 3; }
    else return pl;
    constexpr int max_c = 512;
    if (d.Cin > max_c || d.Cout > max_c) return pl;       // wide layers: the k-pipelined DMA kernel (conv_wgrad_dma.hip)
    Args& a = pl.a;
    for (int j = 0; j < d.njobs; ++j) { a.x[j] = d.x[j]; a.dy3[j] = d.dy3[j]; a.dy1[j] = d.dy1[j]; }
    a.ws = reinterpret_cast<float*>(d.ws);
    a.N = d.N; a.IH = d.IH; a.IW = d.IW; a.Cin = d.Cin; a.OH = d.OH; a.OW = d.OW; a.Cout = d.Cout; a.s = s;
    a.njobs = d.njobs;
    a.n_ci = d.Cin / (16 * MR);
    a.n_co = d.Cout / (16 * NR);
    a.PO = d.OH + 1;
    a.PI = s * a.PO;
    a.XW = (d.OW - 1) * s + 3;
    a.SX = round_stride(16 * MR * 2, s);
    a.XJ = (a.XW * (a.SX / 16) + 63) / 64;
    a.ROWB = a.XJ * 1024;
    a.SD = round_stride(16 * NR * 2, 1);
    if (a.XJ > 4 * XJW) return pl;
    // LDS budget of a workgroup: all of it for the big tiles (one workgroup per CU: 512 registers per lane); the 48 x 48 tile fits
    // two waves per SIMD, and two / three co-resident workgroups overlap one's DMA issue and barriers with the other's MFMAs
    constexpr int lds_env = 0;
    const int LDS_MAX = lds_env > 0 ? lds_env : 160 * 1024 - 512;
    // rows per step: best fill of the 32-pixel k-steps, then the longest step; deepest prefetch that fits
    double best = -1.0;
    int bestR = 0, bestPF = 0;
    for (int R = 1; R <= a.PO && R * d.OW <= 128; ++R) {
        const int P = R * d.OW, P32 = (P + 31) / 32 * 32;
        const int DJ = (P32 * (a.SD / 16) + 63) / 64;
        if (DJ > (MR % 2 == 0 ? 8 : 4) * DJW) continue;       // eight waves issue the DMA of the even-MR tiles (HV = 2)
        for (int PF = 2; PF >= 1; --PF) {
            const int NSLOT = (R - 1) * s + 3 + PF * R * s;
            const long bytes = (long)NSLOT * a.ROWB + (long)(PF + 1) * 2 * DJ * 1024 + P32 * 8 + 1024;
            if (bytes > LDS_MAX) continue;
            // PF = 2 only pays while a step is short (HBM latency not covered by one step of MFMAs)
            // the 96 x 96 tile: the longest well-filled step wins whatever the depth (its fill floor is 295 us of a 670 us launch; measured
            // on 192@14: R / PF = 4 / 2 686-702 us, 4 / 1 687-696, 5 / 1 763, 6 / 1 669, 3 / 2 867, 2 / 2 786)
            const bool bigr = MR == 6 && NR == 6;
            const double score = bigr ? (double)P / P32 + 1e-3 * P
                                      : (double)P / P32 + 1e-3 * P + (PF == 2 && P32 <= 64 ? 0.05 : 0.0) - (PF == 2 && P32 > 64 ? 0.5 : 0.0);
            if (score > best) { best = score; bestR = R; bestPF = PF; }
        }
    }
    if (bestR == 0) return pl;
    a.R = bestR;
    a.PF = bestPF;
    a.P = a.R * d.OW;
    a.P32 = (a.P + 31) / 32 * 32;
    a.DJ = (a.P32 * (a.SD / 16) + 63) / 64;
    a.DHALF = a.DJ * 1024;
    a.DSLOT = 2 * a.DHALF;
    a.NSLOT = (a.R - 1) * s + 3 + a.PF * a.R * s;
    a.off_dy = a.NSLOT * a.ROWB;
    a.off_tab = a.off_dy + (a.PF + 1) * a.DSLOT;
    a.off_sink = (a.off_tab + a.P32 * 8 + 1023) / 1024 * 1024;
    pl.smem = a.off_sink + 1024;
    a.nd = a.R * s * ((a.XJ + 3) / 4) + 2 * ((a.DJ + 3) / 4);
    static const int dbg_env = getenv("HC_WREP_DBG") ? atoi(getenv("HC_WREP_DBG")) : 0;
    a.dbg = dbg_env;
    if (a.PF > 1 && a.nd > 32) return pl;
    const int UT = d.N * a.PO;
    a.total_steps = (UT + a.R - 1) / a.R;
    const int NT = a.n_ci * a.n_co;
    int per_cu = (160 * 1024) / pl.smem;                  // co-resident workgroups: LDS, and registers (2 waves per SIMD at most)
    if (per_cu > ((MR * NR <= 9) ? 2 : 1)) per_cu = (MR * NR <= 9) ? 2 : 1;
    if (per_cu < 1) per_cu = 1;
    int nsplit = 256 * per_cu / (d.njobs * NT);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > a.total_steps) nsplit = a.total_steps;
    a.steps_per_split = (a.total_steps + nsplit - 1) / nsplit;
    a.nsplit = (a.total_steps + a.steps_per_split - 1) / a.steps_per_split;
    pl.grid = ((d.njobs * a.nsplit + 7) / 8) * 8 * NT;
    pl.MR = MR;
    pl.NR = NR;
    pl.HV = (MR % 2 == 0) ? 2 : 1;
    // the 48 x 48 tile: when the LDS footprint leaves room for one workgroup per CU only, run it with eight waves (pixel split)
    constexpr int pv_env = 0;
    pl.PV = (pl.HV == 1 && MR == 3 && (a.P32 >> 5) >= 2 && (pv_env == 2 || (pv_env == 0 && per_cu == 1))) ? 2 : 1;
    pl.ok = true;
    return pl;
}

template <int MR, int NR, int HV, int PV = 1>
int launch(const Plan& pl, hipStream_t st) {
    auto kern = wrep_kernel<MR, NR, HV, PV>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(256 * HV * PV), pl.smem, st, pl.a);
    return hc_launch_status();
}

}  // namespace wrep

extern "C" int hc_rep_wgrad_supported(const hc_rep_wgrad_desc* d) {
    if (d == nullptr) return 0;
    return wrep::make_plan(*d).ok ? 1 : 0;
}

extern "C" int64_t hc_rep_wgrad_ws_bytes(const hc_rep_wgrad_desc* d) {
    if (d == nullptr) return -1;
    const wrep::Plan pl = wrep::make_plan(*d);
    if (!pl.ok) return -1;
    return (int64_t)d->njobs * pl.a.nsplit * pl.PV * d->Cout * 10 * d->Cin * 4;
}

extern "C" int hc_rep_wgrad_plan(const hc_rep_wgrad_desc* d, int32_t* out8) {
    if (d == nullptr || out8 == nullptr) return HC_ERR_ARG;
    const wrep::Plan pl = wrep::make_plan(*d);
    if (!pl.ok) return HC_ERR_ARG;
    out8[0] = pl.MR; out8[1] = pl.NR; out8[2] = pl.a.R; out8[3] = pl.a.PF; out8[4] = pl.a.nsplit; out8[5] = pl.grid;
    out8[6] = pl.smem; out8[7] = pl.a.NSLOT;
    return HC_OK;
}

extern "C" int hc_rep_wgrad(const hc_rep_wgrad_desc* dp, hc_stream_t stream) {
    if (dp == nullptr) return HC_ERR_ARG;
    const hc_rep_wgrad_desc& d = *dp;
    if (d.ws == nullptr) return HC_ERR_ARG;
    for (int j = 0; j < d.njobs && j < HC_WREP_MAX_JOBS; ++j)
        if (d.x[j] == nullptr || d.dy3[j] == nullptr || d.dy1[j] == nullptr || d.dw3[j] == nullptr || d.dw1[j] == nullptr) return HC_ERR_ARG;
    const wrep::Plan pl = wrep::make_plan(d);
    if (!pl.ok) return HC_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc;
    if (pl.MR == 6 && pl.NR == 6) rc = wrep::launch<6, 6, 2>(pl, st);
    else if (pl.MR == 6) rc = wrep::launch<6, 3, 2>(pl, st);          // even MR: always the eight-wave (HV = 2) form
    else if (pl.MR == 4) rc = wrep::launch<4, 4, 2>(pl, st);
    else rc = pl.PV == 2 ? wrep::launch<3, 3, 1, 2>(pl, st) : wrep::launch<3, 3, 1>(pl, st);
    if (rc != HC_OK) return rc;
    const int ns = pl.a.nsplit * pl.PV;
    if (ns <= 32) {
        hipLaunchKernelGGL(wrep::wrep_reduce_kernel<1>, dim3(d.Cout, (d.Cin + 63) / 64, d.njobs), dim3(640), 0, st, pl.a.ws, d, ns);
    } else if (ns <= 128) {
        hipLaunchKernelGGL(wrep::wrep_reduce_kernel<4>, dim3(d.Cout, (d.Cin + 15) / 16, d.njobs), dim3(640), 0, st, pl.a.ws, d, ns);
    } else {
        hipLaunchKernelGGL(wrep::wrep_reduce_kernel<8>, dim3(d.Cout, (d.Cin + 7) / 8, d.njobs), dim3(640), 0, st, pl.a.ws, d, ns);
    }
    return hc_launch_status();
}
