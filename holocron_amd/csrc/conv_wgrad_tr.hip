// Weight gradient with the operands staged ONCE per pixel chunk in their natural NHWC layout.
//
//   dW[co][ci][kh][kw] = sum_m dy[m][co] * x[pix(m) + tap][ci]
//
// The generic kernel (conv_wgrad.hip) re-gathers x once per tap and transposes in registers; on
// the 48/96-channel 112^2..28^2 layers of RepVGG that is 9x the L2 traffic and VALU-bound.  Here a
// workgroup stages a chunk of R output rows of dy and the matching input rows of x (with halo) in
// LDS *once*, in their natural NHWC layout, and computes every tap of the group from that one
// image: the shifted operand of tap (kh,kw) is just a different LDS address, and the
// pixel-major -> k-contiguous transposition MFMA needs comes for free from the gfx950
// transposing LDS read (ds_read_b64_tr_b16; semantics probed on hardware, see
// scripts/probes/tr16_probe.hip: inside each 16-lane group lane i receives element (i&3) of the
// 8-byte rows addressed by lanes 4j + (i>>2), j = 0..3).
//
// MFMA: v_mfma_f32_16x16x32_bf16, A = x (rows = ci), B = dy (cols = co), k = 32 output pixels.
// Output: fp32 slabs [split][co][tap][ci], reduced by wgrad_reduce_kernel (conv_wgrad.hip).
#include "common.h"
#include <stdlib.h>
#include "../../include/holocron_hip.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

namespace wtr {

struct Args {
    hc_wgrad_desc d;
    int R, P, P32;             // output rows per chunk, pixels per chunk (R * CW), padded to 32
    int CW, nseg;              // output columns per chunk and column segments per row (wide images: OW > 128)
    int XR, XW, SX, SD;        // staged x region rows/cols, LDS row strides (bytes) of x and dy
    int chunks_per_img, nchunks, chunks_per_split;
    int n_ci_tiles, n_co_tiles, n_tg, BCO;
    int off_dy, off_tab, off_zero;  // LDS byte offsets (x tile at 0)
};

__device__ __forceinline__ s16x4 tr_read(const char* lds_base, int off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_base + off));
}

// registers left for the prefetch ring after the accumulators (4*MR*NR*TG) and fragments
constexpr int xmax_for(int mr, int nr, int tg) {
    const int v = (200 - 4 * mr * nr * tg - 16 * nr) / 4;
    return v > 16 ? 16 : (v < 2 ? 2 : v);
}

// SEG: the image is wider than a chunk and is cut into column segments (a separate instantiation so that the
// full-row kernels keep their register budget)
template <int MR, int NR, int TG, bool SEG>
__global__ __launch_bounds__(512) void wgrad_tr_kernel(const Args a) {
    constexpr int BCI = 16 * MR;
    constexpr int DMAX = 4 * NR;                 // dy passes: P32 (<=128) / (32/NR pixels per pass)
    constexpr int XMAX = xmax_for(MR, NR, TG);   // x (row, pass) items per thread; the plan guarantees the bound
    constexpr int NCI = BCI / 8;  // 16-byte chunks per staged x pixel
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const hc_wgrad_desc& d = a.d;
    const int NT = blockDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int b = blockIdx.x;
    const int tg = b % a.n_tg;  b /= a.n_tg;
    const int cot = b % a.n_co_tiles;
    const int cit = b / a.n_co_tiles;
    const int ci0 = cit * BCI, co0 = cot * a.BCO;
    const int split = blockIdx.y;
    const int T = d.KH * d.KW;
    // taps [t0, t0+ntaps) of this group: the whole kernel, one kernel row, or a single tap
    const int t0 = tg * TG;
    // TG always divides T (9|3|1 for 3x3, 1 for 1x1): every group is full, so the tap loop is
    // straight-line code and the compiler can run the LDS reads ahead of the MFMAs
    const int kh0 = t0 / d.KW;
    const int s = d.stride;

    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(d.x, (unsigned)d.N * d.IH * d.IW * d.Cin * 2u);
    const __amdgpu_buffer_rsrc_t rsy = make_rsrc(d.dy, (unsigned)d.N * d.OH * d.OW * d.Cout * 2u);

    char* sx = smem;
    char* sdy = smem + a.off_dy;
    int* tab = reinterpret_cast<int*>(smem + a.off_tab);

    // pixel table: LDS offset of the tap-(0,0) input pixel of chunk pixel p (chunk-invariant)
    for (int p = tid; p < a.P32; p += NT) {
        int off = 0;
        if (p < a.P) {
            const int r = p / a.CW, c = p - r * a.CW;
            off = ((r * s) * a.XW + c * s) * a.SX;
        }
        tab[p] = off;
    }
    if (tid < 4) reinterpret_cast<int*>(smem + a.off_zero)[tid] = 0;

    f32x4 acc[TG][MR][NR];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int q = 0; q < NR; ++q) acc[t][m][q] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment lane constants
    const int la = lane & 15, kq = lane >> 4;
    // k-slot -> pixel map (identical for both operands, so any permutation is legal): read r of a
    // lane covers pixels 16r + 4kq + (la>>2), which makes each 32-lane LDS group touch 8
    // CONSECUTIVE pixels -> 8 distinct 32-byte bank slots with the strides of round_stride()
    const int prow = 4 * kq + (la >> 2);
    const int cq = 4 * (la & 3);                         // channel quad within a 16-channel block
    int dy_coff[NR];                                     // byte offset of this lane's channel quad in a dy row
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int c = 16 * (wid * NR + q) + cq;
        dy_coff[q] = (c < a.BCO && co0 + c < d.Cout) ? c * 2 : -1;
    }
    // x channel offsets: the ci tile (16*MR) always divides Cin, so every lane's quad is valid and
    // a tr-read address is ONE add: table offset (per pixel group) + [tap offset + channel offset]
    const int x_c0 = cq * 2;

    const int c_begin = split * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    if (c_end > a.nchunks) c_end = a.nchunks;

    int toff[TG];   // LDS byte offset of every tap of the group inside the staged x image (uniform)
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        const int kh = (t0 + t) / d.KW, kw = (t0 + t) - kh * d.KW;
        toff[t] = ((kh - kh0) * a.XW + kw) * a.SX;
    }
    // staging roles (chunk-invariant)
    const int NCO8 = a.BCO / 8;                 // 16-byte chunks per dy pixel; NT % NCO8 == 0 always
    const int dcc = tid % NCO8, dp = tid / NCO8, dpp = NT / NCO8;
    const int xpp = NT / NCI;                   // x pixels per pass
    const int xcc = tid % NCI, xp = tid / NCI;
    const bool xstager = tid < xpp * NCI;
    const int xrp = (a.XW + xpp - 1) / xpp;     // passes per staged x row
    const int xitems = a.XR * xrp;              // <= XMAX
    u32x4 sd[DMAX], sxr[XMAX];

    // issue every global load of chunk `ch` into registers (nothing waits here)
    auto issue = [&](int ch) {
        const int n = ch / a.chunks_per_img;
        const int rem = ch - n * a.chunks_per_img;
        const int rc = SEG ? rem / a.nseg : rem;
        const int ox0 = SEG ? (rem - rc * a.nseg) * a.CW : 0;
        const int oy0 = rc * a.R;
        const int rows_left = d.OH - oy0, cols_left = d.OW - ox0;
        const int rvalid = rows_left < a.R ? rows_left : a.R;
        const int cvalid = cols_left < a.CW ? cols_left : a.CW;
        const unsigned gbase = (unsigned)((n * d.OH + oy0) * d.OW + ox0) * (unsigned)d.Cout * 2u + (unsigned)co0 * 2u +
                               (unsigned)dcc * 16u;
        const bool dcok = co0 + dcc * 8 < d.Cout;
        if (!SEG) {                  // full rows: the chunk's pixels are contiguous in dy
            const int pvalid = rvalid * d.OW;
#pragma unroll
            for (int i = 0; i < DMAX; ++i) {
                const int p = i * dpp + dp;
                sd[i] = buf_load16(rsy, (dcok && p < pvalid) ? gbase + (unsigned)p * (unsigned)d.Cout * 2u : HC_OOB);
            }
        } else {                     // column segment: (row, col) of every slot (wide images only; costs a division per slot)
#pragma unroll
            for (int i = 0; i < DMAX; ++i) {
                const int p = i * dpp + dp;
                const int r = p / a.CW, c = p - r * a.CW;
                const bool ok = dcok && (r < rvalid) && (c < cvalid);
                sd[i] = buf_load16(rsy, ok ? gbase + (unsigned)(r * d.OW + c) * (unsigned)d.Cout * 2u : HC_OOB);
            }
        }
        const int iy_base = oy0 * s - d.pad + kh0;
        const bool xcok = xstager && (ci0 + xcc * 8 < d.Cin);
        const unsigned nbase = (unsigned)(n * d.IH * d.IW) * (unsigned)d.Cin * 2u + (unsigned)ci0 * 2u + (unsigned)xcc * 16u;
        int xr = 0, ps = 0;
#pragma unroll
        for (int j = 0; j < XMAX; ++j) {
            const int iy = iy_base + xr, ix = ox0 * s + ps * xpp + xp - d.pad;
            const bool ok = xcok && (j < xitems) && ((unsigned)iy < (unsigned)d.IH) && ((unsigned)ix < (unsigned)d.IW);
            sxr[j] = buf_load16(rsx, ok ? nbase + (unsigned)(iy * d.IW + ix) * (unsigned)d.Cin * 2u : HC_OOB);
            if (++ps == xrp) { ps = 0; ++xr; }
        }
    };
    // registers -> LDS (natural NHWC rows)
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < DMAX; ++i) {
            const int p = i * dpp + dp;
            if (p < a.P32) *reinterpret_cast<u32x4*>(sdy + p * a.SD + dcc * 16) = sd[i];
        }
        int xr = 0, ps = 0;
#pragma unroll
        for (int j = 0; j < XMAX; ++j) {
            const int xc = ps * xpp + xp;
            if (xstager && j < xitems && xc < a.XW)
                *reinterpret_cast<u32x4*>(sx + (xr * a.XW + xc) * a.SX + xcc * 16) = sxr[j];
            if (++ps == xrp) { ps = 0; ++xr; }
        }
    };

    // PREFETCH: keep the next chunk's loads in flight during the MFMA phase (costs DMAX+XMAX
    // 4-register slots that stay live across the MFMAs)
    constexpr bool PREFETCH = (4 * MR * NR * TG + 4 * (DMAX + XMAX)) <= 150;
    if (PREFETCH && c_begin < c_end) issue(c_begin);
    for (int ch = c_begin; ch < c_end; ++ch) {
        if (!PREFETCH) issue(ch);   // all loads of the chunk in flight at once, then the LDS stores
        __syncthreads();   // previous chunk fully consumed (also orders the table writes)
        commit();
        __syncthreads();
        if (PREFETCH && ch + 1 < c_end) issue(ch + 1);
        // ---- MFMA over the chunk: 32 pixels per step, every tap from the same staged image -------
        for (int g = 0; g < a.P32; g += 32) {
            const int p0 = g + prow;
            const int xb0 = tab[p0] + x_c0, xb1 = tab[p0 + 16] + x_c0;
            bf16x8 fb[NR];
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int o0 = dy_coff[q] >= 0 ? a.off_dy + p0 * a.SD + dy_coff[q] : a.off_zero;
                const int o1 = dy_coff[q] >= 0 ? a.off_dy + (p0 + 16) * a.SD + dy_coff[q] : a.off_zero;
                const s16x4 lo = tr_read(smem, o0), hi = tr_read(smem, o1);
                fb[q] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int t = 0; t < TG; ++t) {
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const int tm = toff[t] + 32 * m;   // wave-uniform
                    const s16x4 lo = tr_read(smem, xb0 + tm), hi = tr_read(smem, xb1 + tm);
                    const bf16x8 fa = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                    for (int q = 0; q < NR; ++q)
                        acc[t][m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[q], acc[t][m][q], 0, 0, 0);
                }
            }
        }
    }

    // ---- slab[split][co][tap][ci]: lane = co column, the 4 accumulator values = 4 consecutive ci ----
    float* ws = reinterpret_cast<float*>(d.ws);
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int cl = 16 * (wid * NR + q) + la;
        const int co = co0 + cl;
        if (cl < a.BCO && co < d.Cout) {
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const int tap = t0 + t;
                float* row = ws + (((long)split * d.Cout + co) * T + tap) * d.Cin;
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const int ci = ci0 + 16 * m + 4 * kq;
                    if (ci < d.Cin) *reinterpret_cast<f32x4*>(row + ci) = acc[t][m][q];
                }
            }
        }
    }
}

inline int round_stride(int bytes, int stride) {
    // LDS row stride >= bytes, 16-byte aligned, such that 8 consecutive staged pixels (pixel step =
    // conv stride) land on 8 distinct 32-byte slots of the 256-byte bank row (tr_b64 reads)
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
    int MR, NR, TG, WN, smem, nsplit;
};

inline Plan make_plan_seg(const hc_wgrad_desc& d, int NR, int nseg) {
    Plan pl{};
    pl.ok = false;
    const int T = d.KH * d.KW;
    if (d.Cin % 16 || d.Cout % 16 || d.stride < 1 || d.stride > 2) return pl;
    if (!(T == 9 && d.KH == 3) && T != 1) return pl;
    // measured on MI355X (scripts/bench_layers.py): beyond 192x256 channels the k-pipelined generic
    // kernel is faster (1280-wide layers re-stage dy once per ci tile here)
    constexpr int max_cin = 192;
    constexpr int max_cout = 256;
    if (d.Cin > max_cin || d.Cout > max_cout) return pl;
    // images wider than a chunk (128 pixels) are cut into column segments, each staged with its own halo
    const int CW = (d.OW + nseg - 1) / nseg;
    // ci tile = 16*MR: the largest supported MR that divides Cin/16
    const int c16 = d.Cin / 16;
    int MR = 0;
    for (int m : {8, 6, 4, 3, 2, 1})       // (1: 16-channel inputs - ReXNet's first expansion 16 -> 96 @ 112 ran the generic kernel at 1.5 TB/s)
        if (c16 % m == 0) { MR = m; break; }
    if (MR == 0) return pl;
    int TG = T;
    if (T == 9 && MR * NR > 4) TG = 3;       // accumulators: TG*MR*NR*4 registers per lane
    if (MR * NR * TG > 36) TG = 1;
    Args& a = pl.a;
    a.d = d;
    a.n_tg = T / TG;
    a.n_ci_tiles = c16 / MR;
    const int cob = (d.Cout + 16 * NR - 1) / (16 * NR);  // co blocks of one wave
    int WN = cob;
    a.n_co_tiles = 1;
    const int max_wn = (MR * NR * TG >= 36) ? 6 : 8;
    while (WN > max_wn) {
        a.n_co_tiles += 1;
        WN = (cob + a.n_co_tiles - 1) / a.n_co_tiles;
    }
    a.BCO = 16 * NR * WN;
    const int nkh = (TG >= T) ? d.KH : 1;   // kernel rows spanned by one tap group
    a.SD = round_stride(a.BCO * 2, 1);
    a.SX = round_stride(16 * MR * 2, d.stride);
    const int xpp = 64 * WN / (2 * MR);      // staged x pixels per pass of the workgroup
    const int xmax = xmax_for(MR, NR, TG);
    a.CW = CW;
    a.nseg = nseg;
    int R = 112 / CW;
    if (R < 1) R = 1;
    if (R > d.OH) R = d.OH;
    for (;; --R) {
        a.R = R;
        a.P = R * CW;
        a.P32 = (a.P + 31) / 32 * 32;
        a.XR = (R - 1) * d.stride + nkh;
        a.XW = (CW - 1) * d.stride + d.KW;
        const int xbytes = a.XR * a.XW * a.SX;
        a.off_dy = (xbytes + 255) / 256 * 256;
        a.off_tab = a.off_dy + (a.P32 * a.SD + 255) / 256 * 256;
        a.off_zero = a.off_tab + a.P32 * 4;
        pl.smem = a.off_zero + 16;
        const int xitems = a.XR * ((a.XW + xpp - 1) / xpp);
        if (pl.smem <= 78 * 1024 && xitems <= xmax && a.P32 <= 128) break;
        if (R == 1) return pl;               // does not fit the prefetch ring: caller falls back
    }
    a.chunks_per_img = ((d.OH + a.R - 1) / a.R) * nseg;
    a.nchunks = d.N * a.chunks_per_img;
    pl.MR = MR;
    pl.NR = NR;
    pl.TG = TG;
    pl.WN = WN;
    pl.ok = true;
    return pl;
}

// fewest column segments whose halo-extended x rows fit the staging registers / LDS
inline Plan make_plan_nr(const hc_wgrad_desc& d, int NR) {
    const int seg0 = (d.OW + 127) / 128;
    for (int nseg = seg0; nseg <= seg0 + 6 && (nseg == seg0 || (d.OW + nseg - 1) / nseg >= 16); ++nseg) {
        const Plan pl = make_plan_seg(d, NR, nseg);
        if (pl.ok) return pl;
    }
    Plan none{};
    none.ok = false;
    return none;
}

// One resident round: the number of workgroups equals what the chip holds at once (a second,
// partial round of workgroups would double the kernel time), each split a contiguous chunk range.
inline void size_splits(Plan& pl, int blocks_per_cu) {
    Args& a = pl.a;
    const int tiles = a.n_ci_tiles * a.n_co_tiles * a.n_tg;
    int resident = 256 * (blocks_per_cu < 1 ? 1 : blocks_per_cu);
    int nsplit = resident / tiles;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > a.nchunks) nsplit = a.nchunks;
    a.chunks_per_split = (a.nchunks + nsplit - 1) / nsplit;
    pl.nsplit = (a.nchunks + a.chunks_per_split - 1) / a.chunks_per_split;
}

inline Plan make_plan(const hc_wgrad_desc& d) {
    if (d.Cout > 128) {
        const Plan p2 = make_plan_nr(d, 2);
        if (p2.ok && p2.a.P >= 64) return p2;   // two co blocks per wave only if the chunk stays long
    }
    return make_plan_nr(d, 1);
}

// launch == false: only size the plan (workspace query)
template <int MR, int NR, int TG, bool SEG>
int launch(Plan& pl, hipStream_t st, bool do_launch) {
    auto kern = wgrad_tr_kernel<MR, NR, TG, SEG>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    static int occ_cache[9][161];   // [waves][smem KiB] -> workgroups per CU (0 = unknown)
    const int kib = (pl.smem + 1023) / 1024;
    int& occ = occ_cache[pl.WN][kib];
    if (occ == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 64 * pl.WN, pl.smem) != hipSuccess || n < 1) n = 1;
        occ = n;
    }
    size_splits(pl, occ);
    if (!do_launch) return HC_OK;
    dim3 grid(pl.a.n_ci_tiles * pl.a.n_co_tiles * pl.a.n_tg, pl.nsplit);
    hipLaunchKernelGGL(kern, grid, dim3(64 * pl.WN), pl.smem, st, pl.a);
    return hc_launch_status();
}

}  // namespace wtr

// entry points used by conv_wgrad.hip's dispatcher
static int dispatch(wtr::Plan& pl, hipStream_t st, bool do_launch) {
    const int T = pl.a.d.KH * pl.a.d.KW;
#define WTR_CASE(M, N, G) \
    if (pl.MR == M && pl.NR == N && pl.TG == G) \
        return pl.a.nseg > 1 ? wtr::launch<M, N, G, true>(pl, st, do_launch) : wtr::launch<M, N, G, false>(pl, st, do_launch);
    if (T == 1) {
        WTR_CASE(1, 1, 1) WTR_CASE(1, 2, 1)
        WTR_CASE(2, 1, 1) WTR_CASE(3, 1, 1) WTR_CASE(4, 1, 1) WTR_CASE(6, 1, 1) WTR_CASE(8, 1, 1)
        WTR_CASE(2, 2, 1) WTR_CASE(3, 2, 1) WTR_CASE(4, 2, 1) WTR_CASE(6, 2, 1) WTR_CASE(8, 2, 1)
    } else {
        WTR_CASE(1, 1, 9) WTR_CASE(1, 2, 9)
        WTR_CASE(2, 1, 9) WTR_CASE(3, 1, 9) WTR_CASE(4, 1, 9) WTR_CASE(6, 1, 3) WTR_CASE(8, 1, 3)
        WTR_CASE(2, 2, 9) WTR_CASE(3, 2, 3) WTR_CASE(4, 2, 3) WTR_CASE(6, 2, 3) WTR_CASE(8, 2, 1)
    }
#undef WTR_CASE
    return -1;
}

int wgrad_tr_nsplit(const hc_wgrad_desc& d) {
    wtr::Plan pl = wtr::make_plan(d);
    if (!pl.ok) return 0;
    if (dispatch(pl, nullptr, false) != HC_OK) return 0;
    return pl.nsplit;
}

int wgrad_tr_launch(const hc_wgrad_desc& d, hipStream_t st, int* nsplit_out) {
    wtr::Plan pl = wtr::make_plan(d);
    if (!pl.ok) return -1;
    const int rc = dispatch(pl, st, true);
    *nsplit_out = pl.nsplit;
    return rc;
}
