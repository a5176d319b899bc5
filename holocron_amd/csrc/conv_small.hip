// Stride-1 3x3 (+ fused 1x1) convolution for SMALL channel counts (C <= 48) on huge images — the
// HBM-bound 48-channel 112^2 / 56^2 RepVGG layers (reference: RepBlock.forward,
// holocron/models/classification/repvgg.py:71-73, and its data gradient).
//
// The generic gather-conv fetches, per 128-pixel tile, the whole weight matrix (41 KB) and nine
// shifted copies of the input from L2: ~150 KB for 12 KB of output.  Here a persistent workgroup
// keeps the packed weights in LDS for its whole lifetime and walks over output-row tiles; for each
// tile the R+2 input rows (with halo, natural NHWC) are DMA'd once into a double-buffered LDS
// window and all nine taps are read from it at shifted addresses.  The 1x1 branch is the centre
// tap with its own weights and accumulator, so x is read ONCE for both branches.
//
//   mode 0 (forward) : out3 = W3 (*) A            out1 = W1 . A            (+ BN statistics of both)
//   mode 1 (dgrad)   : out3 = W3 (*) A + W1 . B + resid      (A = dy3, B = dy1, weights flipped by the packer)
//
// MFMA v_mfma_f32_32x32x16_bf16, D[co][pix]: A operand = weights (rows = out channels, 2 x 32), B =
// pixels (4 waves x 32).  LDS strides are == 112 (mod 256) so that the 16-lane groups of
// ds_read_b128 hit 16 distinct 16-byte bank slots (rows r*112 mod 256 are all different for the
// group row sets {0-3,12-15,20-27} / {4-11,16-19,28-31}).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "../../include/holocron_hip.h"

namespace csm {

#ifdef HC_CSM_TRACE
__device__ unsigned long long g_trace[16];   // cycle sums per phase of one wave (scripts/probes/csm_trace.py; experiments only)
#define CSM_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); if (trace_on) tr[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define CSM_T(i) do { } while (0)
#endif

constexpr int SX = 112;        // bytes per staged pixel (C <= 48 -> <= 96 used)
constexpr int SXO = 144;       // bytes per pixel of the output staging tile (Cout <= 64 -> <= 128 used)
constexpr int NT = 256, NW = 4;
constexpr int MAXJ = 12;       // DMA instructions per wave per window (A) — bound checked on the host

struct Args {
    hc_conv_small_desc d;
    int R, P, XWp, ntiles, tiles_per_img;
    int ws3, ws1;              // LDS row strides (bytes) of the resident weights
    int off_w1, off_win, win_bytes, off_win2, win2_bytes;   // LDS map (w3 at 0); windows are double buffered
    int off_stat;              // fp32 [2 tensors][2][64] running BN sums of this workgroup
    int off_stage;             // output staging tile: [2 tensors][128 px][144 B]
    int nja, njb;              // DMA instructions per wave for window A / B
    int qa, qb;                // DMA instructions per window (the last wave pass may be partial)
    int wrows;                 // weight rows kept in LDS (Cout rounded up to 16; MFMA rows beyond read finite junk)
    int dbg;                   // HC_CSM_DBG knock-outs (timing experiments only): 1 no MFMA/LDS reads, 2 no stats, 4 no stores, 8 no DMA
    int reps;                  // statistics replicas
};

template <int KC>   // KC = C / 16 (1, 2, 3)
__global__ __launch_bounds__(NT, 1) void conv_small_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    const hc_conv_small_desc& d = a.d;
    constexpr int C = 16 * KC;
    constexpr int NCC = C / 8;                 // 16-byte chunks per pixel
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int H = d.H, W = d.W, Cout = d.Cout;
    const bool dgrad = (d.mode & 1) == 1;
    const unsigned img_bytes = (unsigned)d.N * H * W * C * 2u;
    const __amdgpu_buffer_rsrc_t rsa = make_rsrc(d.srcA, img_bytes);
    const __amdgpu_buffer_rsrc_t rsb = make_rsrc(dgrad ? d.srcB : d.srcA, img_bytes);

    // ---- weights live in REGISTERS for the whole workgroup lifetime (one wave per SIMD: 512 VGPRs) ------
    // A fragment of (row block m, tap t, k16 chunk kc): lane (row = lane&31, k half = lane>>5) holds the 8
    // bf16 W[m*32 + row][t][kc*16 + 8*half ..].  Rows >= Cout are zero.
    // Wave (mb, ph): output-channel block mb = wid & 1 (32 rows) x pixel half ph = wid >> 1 (64 pixels).
    const int mb = wid & 1, ph = wid >> 1;
    bf16x8 wreg3[9 * KC], wreg1[KC];
    {
        const bf16_t* w3 = reinterpret_cast<const bf16_t*>(d.w3);
        const bf16_t* w1 = reinterpret_cast<const bf16_t*>(d.w1);
        const int row = mb * 32 + (lane & 31), lh_ = lane >> 5;
        const bool ok = row < Cout;
#pragma unroll
        for (int i = 0; i < 9 * KC; ++i) {
            u32x4 v = u32x4{0, 0, 0, 0};
            if (ok) v = *reinterpret_cast<const u32x4*>(w3 + (long)row * d.w3_rstride + i * 16 + lh_ * 8);
            wreg3[i] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int i = 0; i < KC; ++i) {
            u32x4 v = u32x4{0, 0, 0, 0};
            if (ok) v = *reinterpret_cast<const u32x4*>(w1 + (long)row * d.w1_rstride + i * 16 + lh_ * 8);
            wreg1[i] = __builtin_bit_cast(bf16x8, v);
        }
    }

    float* sstat = reinterpret_cast<float*>(smem + a.off_stat);
    sstat[tid] = 0.f;   // 256 threads == 2*2*64 entries

    // ---- DMA bookkeeping (tile invariant): window A = (R+2) x (W+2) pixels, window B = R x W pixels ----
    // LDS position of (instr q, lane) = q*1024 + 16*lane -> pixel = pos/112, chunk = (pos%112)/16
    int a_rel[MAXJ], a_wr[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        const int q16 = (wid + j * NW) * 64 + lane;      // 16-byte slot index inside the window
        const int pix = q16 / 7, cc = q16 - pix * 7;
        const int wr = pix / a.XWp, wc = pix - wr * a.XWp;
        const bool ok = (wid + j * NW < a.qa) && (cc < NCC) && (wr < a.R + 2) && (wc >= 1) && (wc <= W);
        a_rel[j] = ok ? ((wr * W + (wc - 1)) * C + cc * 8) * 2 : -1;
        a_wr[j] = wr;
    }
    int b_rel[MAXJ / 2];
#pragma unroll
    for (int j = 0; j < MAXJ / 2; ++j) {
        const int q16 = (wid + j * NW) * 64 + lane;
        const int pix = q16 / 7, cc = q16 - pix * 7;
        const bool ok = dgrad && (wid + j * NW < a.qb) && (cc < NCC) && (pix < a.P);
        b_rel[j] = ok ? (pix * C + cc * 8) * 2 : -1;
    }

    auto issue = [&](int tile, int buf) {
        const int n = tile / a.tiles_per_img;
        const int oy0 = (tile - n * a.tiles_per_img) * a.R;
        const int rowbase = ((n * H + oy0 - 1) * W) * C * 2;        // byte offset of window row 0 (may be "negative")
        char* wa = smem + a.off_win + buf * a.win_bytes;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            if (wid + j * NW < a.qa) {                               // wave-uniform
                const int iy = oy0 - 1 + a_wr[j];
                const bool ok = (a_rel[j] >= 0) && ((unsigned)iy < (unsigned)H);
                const unsigned voff = ok ? (unsigned)(rowbase + a_rel[j]) : HC_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_void*)(wa + (wid + j * NW) * 1024), 16, voff, 0, 0, 0);
            }
        }
        if (dgrad) {
            char* wb = smem + a.off_win2 + buf * a.win2_bytes;
            const int rows_left = H - oy0;
            const int pvalid = (rows_left < a.R ? rows_left : a.R) * W;
            const int base2 = ((n * H + oy0) * W) * C * 2;
#pragma unroll
            for (int j = 0; j < MAXJ / 2; ++j) {
                if (wid + j * NW < a.qb) {
                    const int q16 = (wid + j * NW) * 64 + lane;
                    const bool ok = (b_rel[j] >= 0) && (q16 / 7 < pvalid);
                    const unsigned voff = ok ? (unsigned)(base2 + b_rel[j]) : HC_OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_void*)(wb + (wid + j * NW) * 1024), 16, voff, 0, 0, 0);
                }
            }
        }
    };

    // ---- per-lane compute constants: two 32-pixel blocks per wave -----------------------------------
    const int lr = lane & 31, lh = lane >> 5;
    int xoff[2], xoff2[2];
    bool pin[2];
    int prr[2];
#pragma unroll
    for (int nr = 0; nr < 2; ++nr) {
        const int p = ph * 64 + nr * 32 + lr;                 // pixel of this lane inside the tile
        pin[nr] = p < a.P;
        const int pr = pin[nr] ? p / W : 0, pc = pin[nr] ? p - pr * W : 0;
        prr[nr] = pr;
        xoff[nr] = (pr * a.XWp + pc) * SX + lh * 16;          // window A offset of tap (0,0), k-half lh
        xoff2[nr] = (pin[nr] ? p : 0) * SX + lh * 16;         // window B (no halo)
    }

    float* stats3 = d.stats3;
    float* stats1 = d.stats1;
    bf16_t* out3 = reinterpret_cast<bf16_t*>(d.out3);
    bf16_t* out1 = reinterpret_cast<bf16_t*>(d.out1);
    const bf16_t* resid = reinterpret_cast<const bf16_t*>(d.resid);

    float rs1[2][16], rs2[2][16];   // running BN sums [tensor][accumulator register] of this wave's row block
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) { rs1[w][r] = 0.f; rs2[w][r] = 0.f; }

    int tile = blockIdx.x;
    if (tile < a.ntiles) issue(tile, 0);
    int buf = 0;
    for (; tile < a.ntiles; tile += gridDim.x, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // window `buf` landed everywhere; previous tile fully consumed
        const int nxt = tile + gridDim.x;
        if (nxt < a.ntiles && !(a.dbg & 8)) issue(nxt, buf ^ 1);

        const char* wa = smem + a.off_win + buf * a.win_bytes;
        f32x16 acc3[2], acc1[2];   // [pixel block]
#pragma unroll
        for (int nr = 0; nr < 2; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc3[nr][r] = 0.f; acc1[nr][r] = 0.f; }
        if (!(a.dbg & 1))
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            // batch the LDS reads of a whole kernel row (3 taps x KC chunks x 2 pixel blocks) ahead of its MFMAs
            bf16x8 bfr[3][KC][2];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                    for (int nr = 0; nr < 2; ++nr)
                        bfr[kw][kc][nr] = *reinterpret_cast<const bf16x8*>(wa + xoff[nr] + (kh * a.XWp + kw) * SX + kc * 32);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                    for (int nr = 0; nr < 2; ++nr) {
                        acc3[nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg3[(kh * 3 + kw) * KC + kc], bfr[kw][kc][nr], acc3[nr], 0, 0, 0);
                        if (kh == 1 && kw == 1 && !dgrad)   // the 1x1 branch shares the centre-tap pixels
                            acc1[nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg1[kc], bfr[kw][kc][nr], acc1[nr], 0, 0, 0);
                    }
        }
        if (dgrad) {   // + W1^T . dy1 (second source, no halo) into the same accumulator
            const char* wb = smem + a.off_win2 + buf * a.win2_bytes;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
                for (int nr = 0; nr < 2; ++nr) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(wb + xoff2[nr] + kc * 32);
                    acc3[nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg1[kc], b, acc3[nr], 0, 0, 0);
                }
            }
        }

        // ---- epilogue: running statistics (forward) and NHWC stores -------------------------------------
        const int n = tile / a.tiles_per_img;
        const int oy0 = (tile - n * a.tiles_per_img) * a.R;
        const int rows_left = H - oy0;
        const int pvalid = (rows_left < a.R ? rows_left : a.R) * W;     // valid pixels of this tile
        bool live[2];
#pragma unroll
        for (int nr = 0; nr < 2; ++nr) live[nr] = pin[nr] && (oy0 + prr[nr] < H);
        if (stats3 != nullptr && !(a.dbg & 2)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a0_ = live[0] ? acc3[0][r] : 0.f, a1_ = live[1] ? acc3[1][r] : 0.f;
                const float b0_ = live[0] ? acc1[0][r] : 0.f, b1_ = live[1] ? acc1[1][r] : 0.f;
                rs1[0][r] += a0_ + a1_; rs2[0][r] += a0_ * a0_ + a1_ * a1_;
                rs1[1][r] += b0_ + b1_; rs2[1][r] += b0_ * b0_ + b1_ * b1_;
            }
        }
        // Stores: the tile's pixels x Cout channels are ONE contiguous run of NHWC memory.  Stage the tile in
        // LDS ([tensor][pixel][112 B]) and write it out as 16-byte chunks, thread-linear = fully coalesced.
        if (!(a.dbg & 4)) {
            char* stg = smem + a.off_stage;
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                if (which == 1 && out1 == nullptr) break;
#pragma unroll
                for (int nr = 0; nr < 2; ++nr) {
                    char* row = stg + which * (128 * SXO) + (ph * 64 + nr * 32 + lr) * SXO;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int co = mb * 32 + 8 * q + 4 * lh;
                        if (co < Cout) {
                            u32x2 o;
                            if (which == 0) {
                                o[0] = pack_bf16x2(acc3[nr][4 * q], acc3[nr][4 * q + 1]);
                                o[1] = pack_bf16x2(acc3[nr][4 * q + 2], acc3[nr][4 * q + 3]);
                            } else {
                                o[0] = pack_bf16x2(acc1[nr][4 * q], acc1[nr][4 * q + 1]);
                                o[1] = pack_bf16x2(acc1[nr][4 * q + 2], acc1[nr][4 * q + 3]);
                            }
                            *reinterpret_cast<u32x2*>(row + co * 2) = o;
                        }
                    }
                }
            }
            __syncthreads();
            const int cpp = Cout / 8;                                   // 16-byte chunks per pixel
            const int nchunks = pvalid * cpp;
            const long gbase = ((long)(n * H + oy0) * W) * Cout;       // first element of the tile
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                if (which == 1 && out1 == nullptr) break;
                const char* st_ = stg + which * (128 * SXO);
                bf16_t* outp = (which == 0 ? out3 : out1) + gbase;
                for (int c = tid; c < nchunks; c += NT) {
                    const int px = c / cpp, part = c - px * cpp;
                    u32x4 v = *reinterpret_cast<const u32x4*>(st_ + px * SXO + part * 16);
                    if (which == 0 && resid != nullptr) {
                        const u32x4 rv = *reinterpret_cast<const u32x4*>(resid + gbase + (long)c * 8);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[e] = pack_bf16x2(bf16lo(v[e]) + bf16lo(rv[e]), bf16hi(v[e]) + bf16hi(rv[e]));
                    }
                    *reinterpret_cast<u32x4*>(outp + (long)c * 8) = v;
                }
            }
        }
    }
    if (stats3 != nullptr) {   // butterfly over the pixel lanes ONCE, combine the waves in LDS, one flush
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            float s1[16], s2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { s1[r] = rs1[which][r]; s2[r] = rs2[which][r]; }
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
                const int co = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                atomicAdd(&sstat[which * 128 + co], s1[0]);
                atomicAdd(&sstat[which * 128 + 64 + co], s2[0]);
            }
        }
        __syncthreads();
        const int which = tid >> 7, kind = (tid >> 6) & 1, co = tid & 63;
        if (co < Cout) {
            float* st = (which == 0 ? stats3 : stats1) + (size_t)(blockIdx.x % a.reps) * 2 * Cout;
            atomicAdd(st + kind * Cout + co, sstat[tid]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Software-pipelined form of the kernel above.  With one wave per SIMD nothing overlaps by itself: a knock-out
// timing of the 48-channel 112^2 layer (HC_CSM_DBG) showed the phases of a tile simply adding up - loop skeleton
// 66 us, DMA issue 78, LDS reads + MFMAs 268, statistics 60, staging + stores 172 of 591 us.  Here the epilogue of
// tile i-1 (statistics, bf16 packing, staging, coalesced stores) is issued inside the MFMA stream of tile i: the
// accumulators of the previous tile stay in registers (P3 / P1), the statistics and staging writes sit in the same
// basic block as the first kernel row's MFMAs, the stores are issued right after the staging barrier and drain behind
// the remaining two kernel rows.  Stores / residual loads are buffer instructions predicated by an out-of-range
// offset (no divergent loop), their chunk -> (pixel, part) split is hoisted out of the tile loop, the DMA goes through
// inline asm (common.h hc_dma16) and the barriers are raw s_barrier + explicit waits, so that neither the next
// window's DMA nor the draining stores are waited for in the middle of a tile.
template <int KC, bool DGRAD, bool STATS, int NJA, int NJB>
__global__ __launch_bounds__(NT, 1) void conv_small_pipe_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const hc_conv_small_desc& d = a.d;
    constexpr int C = 16 * KC;
    constexpr int NCC = C / 8;
    constexpr int NOUT = DGRAD ? 1 : 2;        // tensors written
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction: keep it in an SGPR
    const int H = d.H, W = d.W, Cout = d.Cout;
    const unsigned one_img = (unsigned)H * W * C * 2u;             // bytes of one input image
    const unsigned out_bytes = (unsigned)d.N * H * W * Cout * 2u;
    const __amdgpu_buffer_rsrc_t rso3 = make_rsrc(d.out3, out_bytes);
    const __amdgpu_buffer_rsrc_t rso1 = make_rsrc(DGRAD ? d.out3 : d.out1, out_bytes);
    const __amdgpu_buffer_rsrc_t rsr = make_rsrc(d.resid, (DGRAD && d.resid != nullptr) ? out_bytes : 0u);   // null: reads 0
    const unsigned smem0 = hc_lds_addr(smem);

    const int mb = wid & 1, ph = wid >> 1;
    bf16x8 wreg3[9 * KC], wreg1[KC];
    {
        const bf16_t* w3 = reinterpret_cast<const bf16_t*>(d.w3);
        const bf16_t* w1 = reinterpret_cast<const bf16_t*>(d.w1);
        const int row = mb * 32 + (lane & 31), lh_ = lane >> 5;
        const bool ok = row < Cout;
#pragma unroll
        for (int i = 0; i < 9 * KC; ++i) {
            u32x4 v = u32x4{0, 0, 0, 0};
            if (ok) v = *reinterpret_cast<const u32x4*>(w3 + (long)row * d.w3_rstride + i * 16 + lh_ * 8);
            wreg3[i] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int i = 0; i < KC; ++i) {
            u32x4 v = u32x4{0, 0, 0, 0};
            if (ok) v = *reinterpret_cast<const u32x4*>(w1 + (long)row * d.w1_rstride + i * 16 + lh_ * 8);
            wreg1[i] = __builtin_bit_cast(bf16x8, v);
        }
    }
    float* sstat = reinterpret_cast<float*>(smem + a.off_stat);
    sstat[tid] = 0.f;

    // ---- DMA bookkeeping.  The source descriptor is rebuilt PER IMAGE (base = image n, range = one image), so that the halo
    // row above the first and below the last image row is out of range by itself (a negative row offset wraps) and the DMA
    // writes the zero padding; per instruction that leaves one v_add: lanes that carry no pixel (halo columns, slack
    // behind the window) add BIG instead of their offset and are out of range for every row.
    constexpr unsigned BIG = 0x40000000u;      // host: one image < BIG - one row
    unsigned a_rel[NJA > 0 ? NJA : 1];
#pragma unroll
    for (int j = 0; j < NJA; ++j) {
        const int q16 = (wid + j * NW) * 64 + lane;
        const int pix = q16 / 7, cc = q16 - pix * 7;
        const int wr = pix / a.XWp, wc = pix - wr * a.XWp;
        const bool ok = (wid + j * NW < a.qa) && (cc < NCC) && (wr < a.R + 2) && (wc >= 1) && (wc <= W);
        a_rel[j] = ok ? (unsigned)(((wr * W + (wc - 1)) * C + cc * 8) * 2) : BIG;
    }
    unsigned b_rel[NJB > 0 ? NJB : 1];
#pragma unroll
    for (int j = 0; j < NJB; ++j) {
        const int q16 = (wid + j * NW) * 64 + lane;
        const int pix = q16 / 7, cc = q16 - pix * 7;
        const bool ok = DGRAD && (wid + j * NW < a.qb) && (cc < NCC) && (pix < a.P);
        b_rel[j] = ok ? (unsigned)((pix * C + cc * 8) * 2) : BIG;
    }
    const unsigned dummy = smem0 + a.off_stat + 1024;
    auto issue = [&](int tile, int buf) {
        const int n = tile / a.tiles_per_img;
        const int oy0 = (tile - n * a.tiles_per_img) * a.R;
        const u32x4 rsa = hc_raw_rsrc(reinterpret_cast<const char*>(d.srcA) + (size_t)n * one_img, one_img);
        const unsigned rowrel = (unsigned)((oy0 - 1) * W * C * 2);          // "negative" for the first tile of an image
        const unsigned wa = smem0 + a.off_win + buf * a.win_bytes;
        // branch-free: instruction slots beyond the window (wave-uniform) are aimed at a 1 KB dummy area (their lanes are
        // all BIG: the DMA writes zeros there); NJA / NJB bound the unrolled count
#pragma unroll
        for (int j = 0; j < NJA; ++j)
            hc_dma16(rsa, (wid + j * NW < a.qa) ? wa + (wid + j * NW) * 1024 : dummy, rowrel + a_rel[j]);
        if (DGRAD) {
            const u32x4 rsb = hc_raw_rsrc(reinterpret_cast<const char*>(d.srcB) + (size_t)n * one_img, one_img);
            const unsigned wb = smem0 + a.off_win2 + buf * a.win2_bytes;
            const unsigned base2 = (unsigned)(oy0 * W * C * 2);               // rows past the image end are out of range
#pragma unroll
            for (int j = 0; j < NJB; ++j)
                hc_dma16(rsb, (wid + j * NW < a.qb) ? wb + (wid + j * NW) * 1024 : dummy, base2 + b_rel[j]);
        }
    };

    const int lr = lane & 31, lh = lane >> 5;
    int xoff[2], xoff2[2], prr[2];
    bool pin[2];
#pragma unroll
    for (int nr = 0; nr < 2; ++nr) {
        const int p = ph * 64 + nr * 32 + lr;
        pin[nr] = p < a.P;
        const int pr = pin[nr] ? p / W : 0, pc = pin[nr] ? p - pr * W : 0;
        prr[nr] = pr;
        xoff[nr] = (pr * a.XWp + pc) * SX + lh * 16;
        xoff2[nr] = (pin[nr] ? p : 0) * SX + lh * 16;
    }
    // coalesced stores: chunk c = tid + 256 k of the tile -> staging offset (tile invariant), global offset 16 c
    const int cpp = Cout / 8;
    int st_lds[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = tid + k * NT;
        const int px = c / cpp, part = c - px * cpp;
        st_lds[k] = (px < 128 ? px : 0) * SXO + part * 16;
    }
    char* stg = smem + a.off_stage;

    // running BN sums [tensor][pair of accumulator registers]: kept as float2 so that the updates are v_pk_mul / v_pk_add /
    // v_pk_fma_f32 - the epilogue of tile i-1 shares the VALU with the address work of tile i, and with one wave per SIMD
    // the VALU, not the MFMA pipe, was the busier unit
    typedef float f2_t __attribute__((ext_vector_type(2)));
    f2_t rs1[2][8], rs2[2][8];
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int r = 0; r < 8; ++r) { rs1[w][r] = f2_t{0.f, 0.f}; rs2[w][r] = f2_t{0.f, 0.f}; }

    // ---- pieces of a tile -------------------------------------------------------------------------------
    // MFMA stream of a tile as 9 (+1) chunks: chunk t = tap (kh, kw) = KC x 2 fragments (k16 chunk x pixel block), chunk 9 = the
    // second source of the data gradient.  Fragments are fetched from LDS TWO chunks ahead of their MFMAs (the scheduler
    // otherwise sinks every ds_read next to its use and each MFMA pair eats a full LDS latency: 5.7k clocks per tile for
    // 1.9k clocks of MFMA); sched_group_barrier pins the read / MFMA interleave.  [T0, T1) is the chunk range of one
    // basic block (the tile is split by the staging barrier).
    constexpr int NCH = DGRAD ? 10 : 9;
    auto load_chunk = [&](const int t, const char* wa, const char* wb, bf16x8 (&f)[KC][2]) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int nr = 0; nr < 2; ++nr)
                f[kc][nr] = (t < 9) ? *reinterpret_cast<const bf16x8*>(wa + xoff[nr] + ((t / 3) * a.XWp + (t % 3)) * SX + kc * 32)
                                    : *reinterpret_cast<const bf16x8*>(wb + xoff2[nr] + kc * 32);
    };
    auto mfma_chunk = [&](const int t, const bf16x8 (&f)[KC][2], f32x16 (&acc3)[2], f32x16 (&acc1)[2]) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int nr = 0; nr < 2; ++nr) {
                acc3[nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t < 9 ? wreg3[t * KC + kc] : wreg1[kc], f[kc][nr], acc3[nr], 0, 0, 0);
                if (t == 4 && !DGRAD)   // the 1x1 branch shares the centre-tap pixels
                    acc1[nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg1[kc], f[kc][nr], acc1[nr], 0, 0, 0);
            }
    };
#define CSM_STEP(T)                                                                          \
    if ((T) >= T0 && (T) < T1) {                                                             \
        if ((T) + 2 < T1) load_chunk((T) + 2, wa, wb, fr[((T) + 2) % 3]);                    \
        mfma_chunk((T), fr[(T) % 3], acc3, acc1);                                            \
    }
#define CSM_PIN(T)                                                                           \
    if ((T) >= T0 && (T) < T1) {                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < KC * 2; ++i_) {                              \
            if ((T) + 2 < T1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             \
            __builtin_amdgcn_sched_group_barrier(0x008, ((T) == 4 && !DGRAD) ? 2 : 1, 0);    \
        }                                                                                    \
    }
    auto mfma_range = [&](auto t0c, auto t1c, const char* wa, const char* wb, f32x16 (&acc3)[2], f32x16 (&acc1)[2]) {
        constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value;
        bf16x8 fr[3][KC][2];
        load_chunk(T0, wa, wb, fr[T0 % 3]);
        if (T0 + 1 < T1) load_chunk(T0 + 1, wa, wb, fr[(T0 + 1) % 3]);
        CSM_STEP(0) CSM_STEP(1) CSM_STEP(2) CSM_STEP(3) CSM_STEP(4) CSM_STEP(5) CSM_STEP(6) CSM_STEP(7) CSM_STEP(8) CSM_STEP(9)
        __builtin_amdgcn_sched_group_barrier(0x100, (T0 + 1 < T1 ? 2 : 1) * KC * 2, 0);
        CSM_PIN(0) CSM_PIN(1) CSM_PIN(2) CSM_PIN(3) CSM_PIN(4) CSM_PIN(5) CSM_PIN(6) CSM_PIN(7) CSM_PIN(8) CSM_PIN(9)
    };
#undef CSM_STEP
#undef CSM_PIN
    using i0_t = std::integral_constant<int, 0>;
    using i3_t = std::integral_constant<int, 3>;
    using iN_t = std::integral_constant<int, NCH>;
    auto stage = [&](const int ptile, const f32x16 (&P3)[2], const f32x16 (&P1)[2]) {   // statistics + bf16 staging of a finished tile
        if (STATS) {
            const int n = ptile / a.tiles_per_img;
            const int oy0 = (ptile - n * a.tiles_per_img) * a.R;
            const float m0 = (pin[0] && (oy0 + prr[0] < H)) ? 1.f : 0.f, m1 = (pin[1] && (oy0 + prr[1] < H)) ? 1.f : 0.f;
            const f2_t m0v = {m0, m0}, m1v = {m1, m1};
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const f2_t a0_ = f2_t{P3[0][2 * r], P3[0][2 * r + 1]} * m0v, a1_ = f2_t{P3[1][2 * r], P3[1][2 * r + 1]} * m1v;
                const f2_t b0_ = f2_t{P1[0][2 * r], P1[0][2 * r + 1]} * m0v, b1_ = f2_t{P1[1][2 * r], P1[1][2 * r + 1]} * m1v;
                rs1[0][r] += a0_ + a1_; rs2[0][r] += a0_ * a0_ + a1_ * a1_;
                rs1[1][r] += b0_ + b1_; rs2[1][r] += b0_ * b0_ + b1_ * b1_;
            }
        }
#pragma unroll
        for (int which = 0; which < NOUT; ++which)
#pragma unroll
            for (int nr = 0; nr < 2; ++nr) {
                char* row = stg + which * (128 * SXO) + (ph * 64 + nr * 32 + lr) * SXO;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // rows >= Cout are zero-weight rows: their (zero) results land in the slack of the 144-byte staging
                    // row (64 channels = 128 bytes fit), which the store loop never reads - no branch in this block
                    const int co = mb * 32 + 8 * q + 4 * lh;
                    u32x2 o;
                    if (which == 0) {
                        o[0] = pack_bf16x2(P3[nr][4 * q], P3[nr][4 * q + 1]);
                        o[1] = pack_bf16x2(P3[nr][4 * q + 2], P3[nr][4 * q + 3]);
                    } else {
                        o[0] = pack_bf16x2(P1[nr][4 * q], P1[nr][4 * q + 1]);
                        o[1] = pack_bf16x2(P1[nr][4 * q + 2], P1[nr][4 * q + 3]);
                    }
                    *reinterpret_cast<u32x2*>(row + co * 2) = o;
                }
            }
    };
    // addresses of a finished tile's 16-byte chunks (out of range past the tile) and, for the data gradient, its residual:
    // loaded at the top of the next tile, a kernel row of MFMAs ahead of the adds that use it
    auto store_prep = [&](const int ptile, unsigned (&voff)[4], u32x4 (&rv)[4]) {
        const int n = ptile / a.tiles_per_img;
        const int oy0 = (ptile - n * a.tiles_per_img) * a.R;
        const int rows_left = H - oy0;
        const int nchunks = (rows_left < a.R ? rows_left : a.R) * W * cpp;
        const unsigned gbase = (unsigned)(((n * H + oy0) * W) * Cout) * 2u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = tid + k * NT;
            voff[k] = c < nchunks ? gbase + (unsigned)c * 16u : HC_OOB;
            if (DGRAD) rv[k] = buf_load16(rsr, voff[k]);
        }
    };
    auto store = [&](const unsigned (&voff)[4], const u32x4 (&rv)[4]) {   // staging -> HBM, 16 bytes per lane, thread-linear
#pragma unroll
        for (int which = 0; which < NOUT; ++which)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u32x4 v = *reinterpret_cast<const u32x4*>(stg + which * (128 * SXO) + st_lds[k]);
                if (DGRAD) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = pack_bf16x2(bf16lo(v[e]) + bf16lo(rv[k][e]), bf16hi(v[e]) + bf16hi(rv[k][e]));
                }
                __builtin_amdgcn_raw_buffer_store_b128(v, which == 0 ? rso3 : rso1, voff[k], 0, 0);
            }
    };
    auto lds_barrier = [&]() {     // LDS traffic of this wave done, then the workgroup barrier; outstanding DMA / stores are NOT waited for
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- tile loop: two accumulator sets whose roles (being computed / being written out) swap every tile --------
    const int G = gridDim.x;
    auto zero = [&](f32x16 (&x3)[2], f32x16 (&x1)[2]) {
#pragma unroll
        for (int nr = 0; nr < 2; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) { x3[nr][r] = 0.f; x1[nr][r] = 0.f; }
    };
#ifdef HC_CSM_TRACE
    const bool trace_on = blockIdx.x == 8 && tid == 0;
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    auto body = [&](const int ntile, const int buf, const int ptile, f32x16 (&cur3)[2], f32x16 (&cur1)[2],
                    const f32x16 (&prv3)[2], const f32x16 (&prv1)[2]) {
        CSM_T(0);
        hc_wait_vmcnt<0>();                    // this tile's window (issued one tile ago) and the previous tile's stores
        CSM_T(1);
        lds_barrier();                         // ... everywhere; all waves are done with the other window and the staging tile
        CSM_T(2);
        if (ntile < a.ntiles) issue(ntile, buf ^ 1);
        CSM_T(3);
        const char* wa = smem + a.off_win + buf * a.win_bytes;
        const char* wb = smem + a.off_win2 + buf * a.win2_bytes;
        zero(cur3, cur1);
        unsigned voff[4];
        u32x4 rv[4];
        store_prep(ptile, voff, rv);
        mfma_range(i0_t{}, i3_t{}, wa, wb, cur3, cur1);   // first kernel row: same basic block as the previous tile's statistics and staging writes
        stage(ptile, prv3, prv1);
        CSM_T(4);
        lds_barrier();                         // staging tile complete
        CSM_T(5);
        store(voff, rv);                       // stores drain behind the remaining MFMAs ...
        __builtin_amdgcn_sched_barrier(0);     // ... so they must be ISSUED before them (the scheduler sank them to the loop end)
        CSM_T(6);
        mfma_range(i3_t{}, iN_t{}, wa, wb, cur3, cur1);
        CSM_T(7);
    };
    // (Swapping two accumulator sets between "being computed" and "being written out" instead of copying 64 registers per
    // tile was tried: the doubled loop body spills ~75 VGPRs.)
    // Tile order.  A window is R + 2 input rows, so a row is wanted by up to three tiles; handed out round-robin
    // (tile = block + k * grid) those are three workgroups on three different XCDs, i.e. three L2s, and the input came in
    // from the memory side three times (PMC: 920 MB fetched for a 308 MB tensor).  Workgroups are dispatched to XCD
    // (block % 8), so instead the grid/8 workgroups of one XCD take grid/8 CONSECUTIVE tiles at every step: neighbouring
    // rows meet in one L2 at the same time and only the rows at the ends of such a run are fetched twice.
    const int per = G >> 3;
    const bool xcd_order = (G & 7) == 0;
    // (k * 8 + x) * per + s  ==  k * G + (x * per + s): the two orders differ in the first tile only - no select inside the
    // loop (a branch there gave the loop a separate latch block, and the compiler sank the statistics update into it, behind
    // the last MFMA)
    const int tile0 = xcd_order ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    auto tile_of = [&](const int k) { return tile0 + k * G; };
    f32x16 P3[2], P1[2], Q3[2], Q1[2];
    int k = 0;
    int tile = tile_of(0);                     // < ntiles: the grid never exceeds the tile count
    issue(tile, 0);
    {                                          // first tile: nothing to overlap with yet
        hc_wait_vmcnt<0>();
        lds_barrier();
        if (tile_of(1) < a.ntiles) issue(tile_of(1), 1);
        zero(P3, P1);
        mfma_range(i0_t{}, iN_t{}, smem + a.off_win, smem + a.off_win2, P3, P1);
    }
    int ptile = tile, buf = 1;
    for (k = 1, tile = tile_of(1); tile < a.ntiles; ++k, tile = tile_of(k), buf ^= 1) {
        body(tile_of(k + 1), buf, ptile, Q3, Q1, P3, P1);
#pragma unroll
        for (int nr = 0; nr < 2; ++nr) { P3[nr] = Q3[nr]; P1[nr] = Q1[nr]; }
        ptile = tile;
    }
    lds_barrier();                             // every wave is past its last reads of the staging tile
    stage(ptile, P3, P1);
    {
        unsigned voff[4];
        u32x4 rv[4];
        store_prep(ptile, voff, rv);
        lds_barrier();
        store(voff, rv);
    }

#ifdef HC_CSM_TRACE
    if (trace_on) for (int q = 0; q < 8; ++q) g_trace[q] = tr[q];
#endif
    if (STATS) {
        float* stats3 = d.stats3;
        float* stats1 = d.stats1;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            float s1[16], s2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { s1[r] = rs1[which][r >> 1][r & 1]; s2[r] = rs2[which][r >> 1][r & 1]; }
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
                const int co = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                atomicAdd(&sstat[which * 128 + co], s1[0]);
                atomicAdd(&sstat[which * 128 + 64 + co], s2[0]);
            }
        }
        __syncthreads();
        const int which = tid >> 7, kind = (tid >> 6) & 1, co = tid & 63;
        if (co < Cout) {
            float* st = (which == 0 ? stats3 : stats1) + (size_t)(blockIdx.x % a.reps) * 2 * Cout;
            atomicAdd(st + kind * Cout + co, sstat[tid]);
        }
    }
}

inline int round112(int bytes) {   // smallest stride >= bytes that is == 112 (mod 256)
    int s = 112;
    while (s < bytes) s += 256;
    return s;
}

inline bool make_args(const hc_conv_small_desc& d, Args& a, int& smem) {
    if (d.C % 16 || d.C > 48 || d.Cout % 8 || d.Cout > 64 || d.W > 128 || d.W < 8 || d.H < 1) return false;
    a.d = d;
    int R = 128 / d.W;
    if (R < 1) R = 1;
    if (R > d.H) R = d.H;
    a.R = R;
    a.P = R * d.W;
    if (a.P > 128) return false;
    a.XWp = d.W + 2;
    a.tiles_per_img = (d.H + R - 1) / R;
    a.ntiles = d.N * a.tiles_per_img;
    a.ws3 = round112(9 * d.C * 2);
    a.ws1 = round112(d.C * 2);
    a.wrows = 0;
    a.off_w1 = 0;
    a.off_stage = 0;
    a.off_win = (2 * 128 * SXO + 1023) / 1024 * 1024;
    const int winA = (R + 2) * a.XWp * SX;
    const int qa = (winA + 1023) / 1024;             // DMA instructions for window A
    a.qa = qa;
    a.nja = (qa + NW - 1) / NW;
    a.win_bytes = qa * 1024;
    a.off_win2 = a.off_win + 2 * a.win_bytes;
    a.njb = 0;
    a.qb = 0;
    a.win2_bytes = 0;
    if ((d.mode & 1) == 1) {
        const int winB = a.P * SX;
        const int qb = (winB + 1023) / 1024;
        a.qb = qb;
        a.njb = (qb + NW - 1) / NW;
        a.win2_bytes = qb * 1024;
    }
    a.off_stat = a.off_win2 + 2 * a.win2_bytes;
    smem = a.off_stat + 2048;   // + 1 KB dummy DMA target of the pipelined kernel
    // rows 32..63 of the second MFMA row block may lie beyond wrows: they must still be inside the allocation
    if (a.nja > MAXJ || a.njb > MAXJ / 2 || smem > 160 * 1024 || (d.Cout % 8) != 0) return false;
    return true;
}

template <int KC>
void launch(const Args& a, int grid, int smem, hipStream_t st) {
    auto kern = conv_small_kernel<KC>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), smem, st, a);
}

template <int KC, bool DGRAD, bool STATS, int NJA, int NJB>
void launch_pipe(const Args& a, int grid, int smem, hipStream_t st) {
    auto kern = conv_small_pipe_kernel<KC, DGRAD, STATS, NJA, NJB>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), smem, st, a);
}
template <int KC>
void launch_pipe_mode(const Args& a, int grid, int smem, hipStream_t st) {
    // the unrolled DMA counts are compile-time: a bucket that covers the window (wide images: 10 + 4; the rest: MAXJ)
    const bool narrow = a.nja <= 10 && a.njb <= 4;
    if ((a.d.mode & 1) == 1) {
        if (narrow) launch_pipe<KC, true, false, 10, 4>(a, grid, smem, st);
        else launch_pipe<KC, true, false, MAXJ, MAXJ / 2>(a, grid, smem, st);
    } else if (a.d.stats3 != nullptr) {
        if (narrow) launch_pipe<KC, false, true, 10, 0>(a, grid, smem, st);
        else launch_pipe<KC, false, true, MAXJ, 0>(a, grid, smem, st);
    } else {
        if (narrow) launch_pipe<KC, false, false, 10, 0>(a, grid, smem, st);
        else launch_pipe<KC, false, false, MAXJ, 0>(a, grid, smem, st);
    }
}

}  // namespace csm

// (the image-resident 8-wave kernel for 64 .. 256 channels of round 2, conv_resident.hip, is retired: the row-unit kernel below
// replaced it on every shape it served - git show 1f81691:holocron_amd/csrc/conv_resident.hip)
// row-unit kernel for 192 @ 14x14 and 96 @ 28x28 (conv_rows.hip); HC_CONV_ROWS=0 disables it
bool hc_conv_rows_supported(const hc_conv_small_desc& d);
int hc_conv_rows_launch(const hc_conv_small_desc& d, hipStream_t st);
// streaming row-unit kernel for 48 channels @ 112 / 56 (conv_rows48.hip); HC_CONV_ROWS48=0 disables it
bool hc_conv_rows48_supported(const hc_conv_small_desc& d);
int hc_conv_rows48_launch(const hc_conv_small_desc& d, hipStream_t st);
extern "C" int hc_conv_small(const hc_conv_small_desc* dp, hc_stream_t stream) {
    if (dp == nullptr) return HC_ERR_ARG;
    const hc_conv_small_desc& d = *dp;
    if (d.mode & HC_CONV_SMALL_ROWS_IMAGE)
        return d.C >= 64 ? hc_conv_rows_launch(d, reinterpret_cast<hipStream_t>(stream))
                         : hc_conv_rows48_launch(d, reinterpret_cast<hipStream_t>(stream));
    if (d.C >= 64) return HC_ERR_ARG;          // 64+ channels: the row-unit image (HC_CONV_SMALL_ROWS_IMAGE) or the gather-conv
    if (d.srcA == nullptr || d.w3 == nullptr || d.w1 == nullptr || d.out3 == nullptr) return HC_ERR_ARG;
    if ((d.mode & 1) == 1 && d.srcB == nullptr) return HC_ERR_ARG;
    if ((d.mode & 1) == 0 && d.out1 == nullptr) return HC_ERR_ARG;
    if ((double)d.N * d.H * d.W * d.C * 2.0 >= 2147483000.0) return HC_ERR_ARG;
    csm::Args a;
    int smem = 0;
    if (!csm::make_args(d, a, smem)) return HC_ERR_ARG;
    { const char* e = getenv("HC_CSM_DBG"); a.dbg = e ? atoi(e) : 0; }
    a.reps = hc_get_stat_replicas();
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int grid = a.ntiles < 256 ? a.ntiles : 256;      // one persistent workgroup per CU
    // test / experiment knobs: HC_CONV_SMALL_GRID caps the grid (several tiles per workgroup on small inputs),
    // HC_CONV_SMALL_PIPE=0 selects the non-pipelined kernel
    const char* e_grid = getenv("HC_CONV_SMALL_GRID");      // read per call: tests flip them inside one process
    const char* e_pipe = getenv("HC_CONV_SMALL_PIPE");
    const int grid_cap = e_grid ? atoi(e_grid) : 0;
    const bool pipe = e_pipe == nullptr || atoi(e_pipe) != 0;
    if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
    const bool pipe_ok = pipe && a.dbg == 0 && (double)d.N * d.H * d.W * d.Cout * 2.0 < 2147483000.0 &&
                         (double)d.H * d.W * d.C * 2.0 < 1.0e9 &&
                         (d.stats3 == nullptr) == (d.stats1 == nullptr);
    if (pipe_ok) {
        if (d.C == 16) csm::launch_pipe_mode<1>(a, grid, smem, st);
        else if (d.C == 32) csm::launch_pipe_mode<2>(a, grid, smem, st);
        else csm::launch_pipe_mode<3>(a, grid, smem, st);
        return hc_launch_status();
    }
    if (d.C == 16) csm::launch<1>(a, grid, smem, st);
    else if (d.C == 32) csm::launch<2>(a, grid, smem, st);
    else csm::launch<3>(a, grid, smem, st);
    return hc_launch_status();
}

#ifdef HC_CSM_TRACE
extern "C" int hc_conv_small_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(csm::g_trace), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 2;
}
#endif

extern "C" int hc_conv_small_supported(const hc_conv_small_desc* dp) {
    if (dp == nullptr) return 0;
    if (dp->mode & HC_CONV_SMALL_ROWS_IMAGE) return (hc_conv_rows_supported(*dp) || hc_conv_rows48_supported(*dp)) ? 1 : 0;
    if (dp->C >= 64) return 0;
    csm::Args a;
    int smem = 0;
    return csm::make_args(*dp, a, smem) ? 1 : 0;
}
