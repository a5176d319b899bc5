// Weight gradient for SMALL channel counts (Cin <= 128): HBM-bound layers with huge pixel counts.
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
#include "../../include/holocron_hip.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

namespace wtr {

struct Args {
    hc_wgrad_desc d;
    int R, P, P32;             // output rows per chunk, valid pixels, padded to 32
    int XR, XW, SX, SD;        // staged x region rows/cols, LDS row strides (bytes) of x and dy
    int chunks_per_img, nchunks, chunks_per_split;
    int n_ci_tiles, n_co_tiles, n_tg, BCO;
    int off_dy, off_tab, off_zero;  // LDS byte offsets (x tile at 0)
};

__device__ __forceinline__ s16x4 tr_read(const char* lds_base, int off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_base + off));
}

template <int MR, int TG>
__global__ __launch_bounds__(512) void wgrad_tr_kernel(const Args a) {
    constexpr int BCI = 16 * MR;
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
    // taps of this group: all of them (n_tg == 1) or one kernel row
    const int kh0 = (a.n_tg == 1) ? 0 : tg;
    const int nkh = (a.n_tg == 1) ? d.KH : 1;
    const int ntaps = nkh * d.KW;
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
            const int r = p / d.OW, c = p - r * d.OW;
            off = ((r * s) * a.XW + c * s) * a.SX;
        }
        tab[p] = off;
    }
    if (tid < 4) reinterpret_cast<int*>(smem + a.off_zero)[tid] = 0;

    f32x4 acc[TG][MR];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment lane constants
    const int la = lane & 15, kq = lane >> 4;
    // k-slot -> pixel map (identical for both operands, so any permutation is legal): read r of a
    // lane covers pixels 16r + 4kq + (la>>2), which makes each 32-lane LDS group touch 8
    // CONSECUTIVE pixels -> 8 distinct 32-byte bank slots with the strides of round_stride()
    const int prow = 4 * kq + (la >> 2);
    const int cq = 4 * (la & 3);                         // channel quad within a 16-channel block
    const int dy_c = 16 * wid + cq;                      // channel within the co tile
    const bool dy_ok = (dy_c < a.BCO) && (co0 + dy_c < d.Cout);
    int x_coff[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) x_coff[m] = (ci0 + 16 * m + cq < d.Cin) ? (16 * m + cq) * 2 : -1;

    const int c_begin = split * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    if (c_end > a.nchunks) c_end = a.nchunks;

    const int NCO8 = a.BCO / 8;
    for (int ch = c_begin; ch < c_end; ++ch) {
        const int n = ch / a.chunks_per_img;
        const int oy0 = (ch - n * a.chunks_per_img) * a.R;
        __syncthreads();  // previous chunk fully consumed (also orders the table writes)
        // ---- stage dy: P32 pixels x BCO channels, rows past the image / chunk are zero --------
        {
            const int rows_left = d.OH - oy0;
            const int pvalid = (rows_left < a.R ? rows_left : a.R) * d.OW;
            const unsigned gbase = (unsigned)((n * d.OH + oy0) * d.OW) * (unsigned)d.Cout * 2u + (unsigned)co0 * 2u;
            const int items = a.P32 * NCO8;
            for (int i = tid; i < items; i += NT) {
                const int p = i / NCO8, cc = i - p * NCO8;
                const bool ok = (p < pvalid) && (co0 + cc * 8 < d.Cout);
                const unsigned voff = ok ? gbase + (unsigned)p * (unsigned)d.Cout * 2u + cc * 16u : HC_OOB;
                *reinterpret_cast<u32x4*>(sdy + p * a.SD + cc * 16) = buf_load16(rsy, voff);
            }
        }
        // ---- stage x: XR rows x XW cols x BCI channels (zero outside the image) ----------------
        {
            const int iy_base = oy0 * s - d.pad + kh0;
            const int rowitems = a.XW * NCI;
            for (int xr = 0; xr < a.XR; ++xr) {
                const int iy = iy_base + xr;
                const bool rok = (unsigned)iy < (unsigned)d.IH;
                const unsigned rbase = (unsigned)((n * d.IH + (rok ? iy : 0)) * d.IW) * (unsigned)d.Cin * 2u + (unsigned)ci0 * 2u;
                char* lrow = sx + xr * a.XW * a.SX;
                for (int i = tid; i < rowitems; i += NT) {
                    const int xc = i / NCI, cc = i - xc * NCI;
                    const int ix = xc - d.pad;
                    const bool ok = rok && ((unsigned)ix < (unsigned)d.IW) && (ci0 + cc * 8 < d.Cin);
                    const unsigned voff = ok ? rbase + (unsigned)ix * (unsigned)d.Cin * 2u + cc * 16u : HC_OOB;
                    *reinterpret_cast<u32x4*>(lrow + xc * a.SX + cc * 16) = buf_load16(rsx, voff);
                }
            }
        }
        __syncthreads();
        // ---- MFMA over the chunk: 32 pixels per step, every tap from the same staged image -------
        for (int g = 0; g < a.P32; g += 32) {
            const int p0 = g + prow;
            const int xo0 = tab[p0], xo1 = tab[p0 + 16];
            bf16x8 fb;
            {
                const int o0 = dy_ok ? a.off_dy + p0 * a.SD + dy_c * 2 : a.off_zero;
                const int o1 = dy_ok ? a.off_dy + (p0 + 16) * a.SD + dy_c * 2 : a.off_zero;
                const s16x4 lo = tr_read(smem, o0), hi = tr_read(smem, o1);
                fb = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                if (t < ntaps) {
                    const int khl = t / d.KW, kw = t - khl * d.KW;
                    const int toff = (khl * a.XW + kw) * a.SX;
#pragma unroll
                    for (int m = 0; m < MR; ++m) {
                        const int o0 = x_coff[m] >= 0 ? xo0 + toff + x_coff[m] : a.off_zero;
                        const int o1 = x_coff[m] >= 0 ? xo1 + toff + x_coff[m] : a.off_zero;
                        const s16x4 lo = tr_read(smem, o0), hi = tr_read(smem, o1);
                        const bf16x8 fa = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                        acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[t][m], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- slab[split][co][tap][ci]: lane = co column, the 4 accumulator values = 4 consecutive ci ----
    float* ws = reinterpret_cast<float*>(d.ws);
    const int co = co0 + 16 * wid + la;
    if (16 * wid + la < a.BCO && co < d.Cout) {
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            if (t < ntaps) {
                const int tap = kh0 * d.KW + t;
                float* row = ws + (((long)split * d.Cout + co) * T + tap) * d.Cin;
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const int ci = ci0 + 16 * m + 4 * kq;
                    if (ci < d.Cin) *reinterpret_cast<f32x4*>(row + ci) = acc[t][m];
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
    int MR, TG, WN, smem, nsplit;
};

inline Plan make_plan(const hc_wgrad_desc& d) {
    Plan pl{};
    pl.ok = false;
    const int T = d.KH * d.KW;
    if (d.Cin % 16 || d.Cout % 16 || d.Cin > 128 || d.stride < 1 || d.stride > 2) return pl;
    if (!(T == 9 && d.KH == 3) && T != 1) return pl;
    const int MR = d.Cin / 16;
    if (MR != 2 && MR != 3 && MR != 4 && MR != 6 && MR != 8) return pl;
    int TG = T;
    if (T == 9 && MR > 4) TG = 3;           // accumulators: TG*MR*4 registers per lane
    Args& a = pl.a;
    a.d = d;
    a.n_tg = T / TG;
    int WN = d.Cout / 16;
    a.n_co_tiles = 1;
    while (WN > 8) {                          // at most 8 waves (co tile <= 128)
        a.n_co_tiles += 1;
        WN = (d.Cout / 16 + a.n_co_tiles - 1) / a.n_co_tiles;
    }
    a.BCO = 16 * WN;
    a.n_ci_tiles = 1;
    const int nkh = (a.n_tg == 1) ? d.KH : 1;
    a.SD = round_stride(a.BCO * 2, 1);
    a.SX = round_stride(d.Cin * 2, d.stride);
    int R = 112 / d.OW;
    if (R < 1) R = 1;
    if (R > d.OH) R = d.OH;
    for (;; --R) {
        a.R = R;
        a.P = R * d.OW;
        a.P32 = (a.P + 31) / 32 * 32;
        a.XR = (R - 1) * d.stride + nkh;
        a.XW = (d.OW - 1) * d.stride + d.KW;
        const int xbytes = a.XR * a.XW * a.SX;
        a.off_dy = (xbytes + 255) / 256 * 256;
        a.off_tab = a.off_dy + (a.P32 * a.SD + 255) / 256 * 256;
        a.off_zero = a.off_tab + a.P32 * 4;
        pl.smem = a.off_zero + 16;
        if (pl.smem <= 72 * 1024 || R == 1) break;
    }
    if (pl.smem > 150 * 1024) return pl;
    a.chunks_per_img = (d.OH + a.R - 1) / a.R;
    a.nchunks = d.N * a.chunks_per_img;
    const int tiles = a.n_ci_tiles * a.n_co_tiles * a.n_tg;
    int nsplit = (640 + tiles - 1) / tiles;
    if (nsplit > a.nchunks) nsplit = a.nchunks;
    a.chunks_per_split = (a.nchunks + nsplit - 1) / nsplit;
    pl.nsplit = (a.nchunks + a.chunks_per_split - 1) / a.chunks_per_split;
    pl.MR = MR;
    pl.TG = TG;
    pl.WN = WN;
    pl.ok = true;
    return pl;
}

template <int MR, int TG>
int launch(const Plan& pl, hipStream_t st) {
    auto kern = wgrad_tr_kernel<MR, TG>;
    static int attr_smem = 0;
    if (pl.smem > attr_smem) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_smem = 160 * 1024;
    }
    dim3 grid(pl.a.n_ci_tiles * pl.a.n_co_tiles * pl.a.n_tg, pl.nsplit);
    hipLaunchKernelGGL(kern, grid, dim3(64 * pl.WN), pl.smem, st, pl.a);
    return hc_launch_status();
}

}  // namespace wtr

// entry points used by conv_wgrad.hip's dispatcher
int wgrad_tr_nsplit(const hc_wgrad_desc& d) {
    const wtr::Plan pl = wtr::make_plan(d);
    return pl.ok ? pl.nsplit : 0;
}

int wgrad_tr_launch(const hc_wgrad_desc& d, hipStream_t st, int* nsplit_out) {
    const wtr::Plan pl = wtr::make_plan(d);
    if (!pl.ok) return -1;
    *nsplit_out = pl.nsplit;
    const int T = d.KH * d.KW;
#define WTR_CASE(M, G) \
    if (pl.MR == M && pl.TG == G) return wtr::launch<M, G>(pl, st);
    if (T == 1) {
        WTR_CASE(2, 1) WTR_CASE(3, 1) WTR_CASE(4, 1) WTR_CASE(6, 1) WTR_CASE(8, 1)
    } else {
        WTR_CASE(2, 9) WTR_CASE(3, 9) WTR_CASE(4, 9) WTR_CASE(6, 3) WTR_CASE(8, 3)
    }
#undef WTR_CASE
    return -1;
}
