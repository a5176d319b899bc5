// Stride-2 RepBlock forward for the HBM-bound front of RepVGG (reference: RepBlock.forward, holocron/models/classification/
// repvgg.py:57-60,71-73 - a 3x3 / pad 1 and a 1x1 / pad 0 conv of the same input with stride 2, each followed by a training-mode
// BatchNorm whose batch statistics need per-channel sum / sum of squares of the fp32 conv results):
//
//   stem            3 @ 224 x 224 (NCHW fp32, the image batch) -> 48 @ 112 x 112      693 MB of HBM traffic per launch at batch 256
//   48 @ 112 x 112  -> 48 @ 56 x 56                                                   462 MB
//   48 @ 56 x 56    -> 96 @ 28 x 28                                                   154 MB
//
// All three have 13 - 120 FLOP per byte: the roof is HBM, and what the gather-conv spent on them (an explicit
// im2col column tensor for the stem - 411 MB written and read back -, 16-channel k-steps with 32-byte gathers, four separate
// launches) was 2.3 - 3.8 x the algorithmic bytes on the read side.  Here a workgroup owns R output rows of
// one image:
//   * the 2R + 1 input rows it needs go into LDS ONCE (LDS DMA for the NHWC bf16 layers; fp32 planes -> bf16 [pixel][4] for the
//     stem), zero halo through out-of-range DMA offsets;
//   * the 3x3 and the 1x1 kernel are ONE K stream of 16-byte pieces (9 taps x Cin/8 pieces, then the Cin/8 pieces of the 1x1 at
//     the centre tap): k32 step s = pieces 4s .. 4s+3, lane group g of v_mfma_f32_16x16x32_bf16 takes piece 4s + g of its pixel.
//     3x3 rows multiply steps [0, S3), 1x1 rows steps [S1B, S); the step that straddles the boundary is used by both with zero
//     weights on the other's pieces - 94 % of the issued MACs are real at 48 channels;
//   * a wave owns 16 (or 32) output channels of BOTH convs and keeps their weights in registers for its whole life (the image of
//     hc_pack_conv_weight modes 5 / 6 is one coalesced 1 KB load per fragment), pixel fragments are 16-byte LDS reads that are
//     bank-conflict free at stride 2 because a window slot is an ODD number of 16-byte units (Cin/8 + 1 pad);
//   * D[co][pixel]: a lane holds 4 consecutive channels of one pixel -> 8-byte stores, the waves of a workgroup fill whole pixels;
//     BatchNorm sums are running register sums, folded over the 16 pixel lanes by DPP row rotations and flushed once per wave.
// Several small workgroups per CU (window 16 - 63 KB) overlap one's window wait with the others' stores: no persistent loop.
#include <cstdlib>
#include "common.h"
#include "../../include/holocron_hip.h"

namespace cs2 {

__device__ __forceinline__ float row16_sum(float v) {      // every lane of a DPP row ends up with the row total
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

// XCD-aware unit order: workgroups are dealt to the 8 XCDs round-robin in dispatch order; give XCD k the k-th contiguous run of
// the (image, row block) list so that the row blocks that share halo rows meet in one L2
__device__ __forceinline__ int xcd_tile(int L, int T) {
    const int q = T >> 3, r = T & 7, xcd = L & 7, j = L >> 3;
    return xcd * q + (xcd < r ? xcd : r) + j;
}

struct Args {
    hc_conv_s2_desc d;
    int reps;
    int dbg;           // timing knock-outs (HC_CONV_S2_DBG; results are wrong): 1 no output stores, 4 no window loads
};

// ------------------------------------------------------------------------------------------------------------------ NHWC bf16 layers
template <int CIN, int COUT, int WIN, int R, int CTW>
struct Geo {
    static constexpr int PT = CIN / 8;                   // 16-byte pieces per tap
    static constexpr int NP3 = 9 * PT, NP = 10 * PT;     // pieces of the 3x3 part / of the whole K stream
    static constexpr int S = (NP + 3) / 4;               // k32 steps
    static constexpr int S3 = (NP3 + 3) / 4;             // steps [0, S3) carry 3x3 pieces
    static constexpr int S1B = NP3 / 4;                  // steps [S1B, S) carry 1x1 pieces
    static constexpr int S1 = S - S1B;
    static constexpr int PSC = PT + 1, PS = PSC * 16;    // window slot: the pixel's channels + one pad chunk
    static constexpr int WS = WIN + 1;                   // slots per window row: left halo + WIN pixels
    static constexpr int ROWS = 2 * R + 1;
    static constexpr int NSLOT = ROWS * WS;
    static constexpr int NDMA = (NSLOT * PS + 1023) / 1024;
    static constexpr int WINB = NDMA * 1024;
    static constexpr int WOUT = WIN / 2, NPIX = R * WOUT, NFRAG = (NPIX + 15) / 16;
    static constexpr int NW = COUT / (16 * CTW), NT = 64 * NW;
    static constexpr int SMEM = WINB + 256;              // slack: the discarded columns of a ragged last fragment read past the window
    static_assert(CIN % 8 == 0 && (PSC & 1) == 1, "a window slot must be an odd number of 16-byte units");
    static_assert(COUT % (16 * CTW) == 0 && NT <= 1024, "waves");
    static_assert(NP % 4 == 0, "the K stream must end on a k32 step");
    static_assert(SMEM <= 160 * 1024, "LDS budget");
};

// Persistent workgroups (the one-row-block-per-workgroup form of round 3 was retired in round 5: 0.14 ms per step slower): two workgroups per CU walk the (image, row block) list of their XCD; the
// weights are loaded ONCE per workgroup instead of once per row block (48 KB of L2 reads against a 38 - 63 KB window), the window
// of block i + 1 is DMA'd into the second LDS buffer while block i is multiplied (one barrier per block), two pixel fragments run
// as independent accumulator chains, and the BatchNorm sums are flushed once per workgroup.
// Measured and dropped: a loader / consumer split (a fourth wave that only issues DMA, LDS counters instead of barriers, so that no
// consumer ever waits on vmcnt - which on CDNA4 also counts its output stores): 136 us against 147 us on the 48 @ 112 layer, nothing
// on the step; one loader wave per CU with three buffers: 192 us.  With the stores knocked out the kernel still takes 111 - 117 us
// for 308 MB: the window DMA (halo rows re-read per block, one zero-filled pad chunk in seven lanes) is what bounds it.
template <int CIN, int COUT, int WIN, int R, int CTW>
__global__ __launch_bounds__(64 * (COUT / (16 * CTW))) void s2_fwd_persist_kernel(const Args a, const int ntiles) {
    using G = Geo<CIN, COUT, WIN, R, CTW>;
    constexpr int PT = G::PT, S = G::S, S3 = G::S3, S1B = G::S1B, S1 = G::S1, PS = G::PS, WS = G::WS, WOUT = G::WOUT;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const hc_conv_s2_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, g = lane >> 4;
    const int H = d.H, HO = H / 2, RB = HO / R;
    const unsigned lds0 = hc_lds_addr(smem);
    // XCD k owns the k-th contiguous eighth of the tile list; its workgroups (local index j of GX) take tiles j, j + GX, ...
    const int xcd = blockIdx.x & 7, jloc = blockIdx.x >> 3, GX = gridDim.x >> 3;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int t_begin = xcd * q8 + (xcd < r8 ? xcd : r8), t_count = q8 + (xcd < r8 ? 1 : 0);
    const u32x4 rs = hc_raw_rsrc(d.x, (unsigned)d.N * H * WIN * CIN * 2u);

    auto stage = [&](int tl, int buf) __attribute__((always_inline)) {
        const int tile = t_begin + tl;
        const int n = tile / RB, r0 = (tile - n * RB) * R;
        const unsigned img = (unsigned)n * (unsigned)(H * WIN * CIN * 2);
        for (int j = wid; j < G::NDMA; j += G::NW) {
            const int J = j * 64 + lane;
            const int slot = J / G::PSC, c = J - slot * G::PSC;
            const int r = slot / WS, x = slot - r * WS;
            const int ih = 2 * r0 - 1 + r;
            const bool ok = c < PT && slot < G::NSLOT && x >= 1 && ih >= 0;
            const unsigned off = img + (unsigned)((ih * WIN + x - 1) * CIN * 2 + c * 16);
            hc_dma16(rs, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * G::WINB + j * 1024)), ok ? off : HC_OOB);
        }
    };
    if (jloc < t_count) stage(jloc, 0);

    const __amdgpu_buffer_rsrc_t rw3 = make_rsrc(d.w3img, (unsigned)(COUT / 16 * S3 * 1024));
    const __amdgpu_buffer_rsrc_t rw1 = make_rsrc(d.w1img, (unsigned)(COUT / 16 * S1 * 1024));
    u32x4 a3[CTW][S3], a1[CTW][S1];
#pragma unroll
    for (int t = 0; t < CTW; ++t) {
        const int ct = wid * CTW + t;
#pragma unroll
        for (int s = 0; s < S3; ++s) a3[t][s] = buf_load16(rw3, (unsigned)(lane * 16), (unsigned)((ct * S3 + s) * 1024));
#pragma unroll
        for (int s = 0; s < S1; ++s) a1[t][s] = buf_load16(rw1, (unsigned)(lane * 16), (unsigned)((ct * S1 + s) * 1024));
    }
    int boff[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int q = 4 * s + g;
        const int tap = q < G::NP3 ? q / PT : 4;
        const int c = q < G::NP3 ? q - tap * PT : q - G::NP3;
        boff[s] = ((tap / 3) * WS + tap % 3) * PS + c * 16;
    }
    float st3[CTW][2][4], st1[CTW][2][4];
#pragma unroll
    for (int t = 0; t < CTW; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) st3[t][k][e] = st1[t][k][e] = 0.f;

    bf16_t* y3 = reinterpret_cast<bf16_t*>(d.y3);
    bf16_t* y1 = reinterpret_cast<bf16_t*>(d.y1);
    constexpr int NF2 = (G::NFRAG + 1) / 2;
    int it = 0;
#pragma unroll 1
    for (int tl = jloc; tl < t_count; tl += GX, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of block `tl` have landed (and its earlier stores are out)
        __syncthreads();                                       // ... everybody's; and everybody is done reading the other buffer
        if (tl + GX < t_count) stage(tl + GX, (it + 1) & 1);
        const int tile = t_begin + tl;
        const int n = tile / RB, r0 = (tile - n * RB) * R;
        const size_t obase = ((size_t)n * HO + r0) * WOUT * COUT;
        const char* wbase = smem + (it & 1) * G::WINB;
#pragma unroll 1
        for (int f2 = 0; f2 < NF2; ++f2) {
            int p[2];
            bool ok[2];
            const char* pb[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                p[u] = (2 * f2 + u) * 16 + px;
                ok[u] = p[u] < G::NPIX;
                const int pc = ok[u] ? p[u] : G::NPIX - 1;
                const int orow = pc / WOUT, ox = pc - orow * WOUT;
                pb[u] = wbase + (2 * orow * WS + 2 * ox) * PS;
            }
            f32x4 acc3[2][CTW], acc1[2][CTW];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < CTW; ++t) acc3[u][t] = acc1[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < S; ++s) {
                bf16x8 b[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) b[u] = *reinterpret_cast<const bf16x8*>(pb[u] + boff[s]);
#pragma unroll
                for (int t = 0; t < CTW; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (s < S3) acc3[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a3[t][s]), b[u], acc3[u][t], 0, 0, 0);
                        if (s >= S1B) acc1[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1[t][s - S1B]), b[u], acc1[u][t], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < CTW; ++t) {
                    const int co = 16 * (wid * CTW + t) + 4 * g;
                    if (d.stats3 != nullptr) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v3 = ok[u] ? acc3[u][t][e] : 0.f, v1 = ok[u] ? acc1[u][t][e] : 0.f;
                            st3[t][0][e] += v3; st3[t][1][e] += v3 * v3;
                            st1[t][0][e] += v1; st1[t][1][e] += v1 * v1;
                        }
                    }
                    if (ok[u] && !(a.dbg & 1)) {
                        const size_t o = obase + (size_t)p[u] * COUT + co;
                        *reinterpret_cast<u32x2*>(y3 + o) = u32x2{pack_bf16x2(acc3[u][t][0], acc3[u][t][1]), pack_bf16x2(acc3[u][t][2], acc3[u][t][3])};
                        *reinterpret_cast<u32x2*>(y1 + o) = u32x2{pack_bf16x2(acc1[u][t][0], acc1[u][t][1]), pack_bf16x2(acc1[u][t][2], acc1[u][t][3])};
                    }
                }
        }
    }
    if (d.stats3 != nullptr) {
        const size_t slot = (size_t)((blockIdx.x * G::NW + wid) % a.reps) * 2 * COUT;
#pragma unroll
        for (int t = 0; t < CTW; ++t) {
            const int co = 16 * (wid * CTW + t) + 4 * g;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float m3 = 0.f, m1 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s3 = row16_sum(st3[t][k][e]), s1 = row16_sum(st1[t][k][e]);
                    m3 = px == e ? s3 : m3;
                    m1 = px == e ? s1 : m1;
                }
                if (px < 4) {
                    atomicAdd(d.stats3 + slot + k * COUT + co + px, m3);
                    atomicAdd(d.stats1 + slot + k * COUT + co + px, m1);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ stem (3 channels)
// Window slot = one input pixel as {c0, c1, c2, 0} bf16 (8 bytes); a window row = [left halo | 224 pixels | one zero slot].  A kernel
// row of the 3x3 is 16 k slots: the three pixels (2 ox - 1 .. 2 ox + 1) x 4 and four slots with zero weights, i.e. two 16-byte
// reads at byte 16 ox and 16 ox + 16 of the row (aligned, conflict free).  k32 step 0 = kernel rows 0 and 1 (lane groups 0, 1 | 2, 3),
// step 1 = kernel row 2 (the other half of its K reads row 2 again against zero weights); the 1x1 lives in step 0 (row 1, pixel 2 ox).
template <int R>
struct StemGeo {
    static constexpr int WIN = 224, WOUT = 112, COUT = 48;
    static constexpr int WSB = (WIN + 2) * 8;            // bytes per window row (226 slots)
    static constexpr int ROWS = 2 * R + 1;
    static constexpr int NFRAG = R * (WOUT / 16);
    static constexpr int NT = 256;
    static constexpr int OPITCH = 104;                   // staged output pixel: 96 bytes + 8 (26 dwords: the 8-byte writes of 16 pixels hit 32 banks once)
    static constexpr int OST = 2 * 16 * OPITCH;          // per wave: one fragment of both outputs
    static constexpr int WINB = (ROWS * WSB + 63) / 64 * 64;
    static constexpr int SMEM = WINB + (NT / 64) * OST + 64;
};

template <int R>
__global__ __launch_bounds__(256) void s2_stem_kernel(const Args a) {
    using G = StemGeo<R>;
    constexpr int WIN = G::WIN, WOUT = G::WOUT, COUT = G::COUT, WSB = G::WSB;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const hc_conv_s2_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, g = lane >> 4;
    const int H = d.H, HO = H / 2, RB = HO / R;
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int n = tile / RB, r0 = (tile - n * RB) * R;

    // ---- weights: all three 16-channel tiles of both convs (pack mode 6): 3 x (2 + 1) fragments
    const __amdgpu_buffer_rsrc_t rw3 = make_rsrc(d.w3img, 3u * 2u * 1024u);
    const __amdgpu_buffer_rsrc_t rw1 = make_rsrc(d.w1img, 3u * 1024u);
    u32x4 a3[3][2], a1[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        a3[t][0] = buf_load16(rw3, (unsigned)(lane * 16), (unsigned)((t * 2 + 0) * 1024));
        a3[t][1] = buf_load16(rw3, (unsigned)(lane * 16), (unsigned)((t * 2 + 1) * 1024));
        a1[t] = buf_load16(rw1, (unsigned)(lane * 16), (unsigned)(t * 1024));
    }

    // ---- window: fp32 NCHW planes -> bf16 {c0, c1, c2, 0} slots.  One item = 4 consecutive pixels of a row (three float4 loads)
    {
        const float* xin = reinterpret_cast<const float*>(d.x) + (size_t)n * 3 * H * WIN;
        constexpr int IPR = WIN / 4, NITEM = G::ROWS * IPR;
        constexpr int NIT = (NITEM + G::NT - 1) / G::NT;
        f32x4 v[NIT][3];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * G::NT + tid;
            const int r = idx / IPR, c4 = idx - r * IPR;
            const int ih = 2 * r0 - 1 + r;
            const bool ok = idx < NITEM && ih >= 0 && !(a.dbg & 4);
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
                v[it][ci] = ok ? *reinterpret_cast<const f32x4*>(xin + ((size_t)ci * H + ih) * WIN + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * G::NT + tid;
            const int r = idx / IPR, c4 = idx - r * IPR;
            if (idx < NITEM) {
                char* wp = smem + r * WSB + (1 + 4 * c4) * 8;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<u32x2*>(wp + 8 * k) = u32x2{pack_bf16x2(v[it][0][k], v[it][1][k]), pack_bf16x2(v[it][2][k], 0.f)};
            }
        }
        if (tid < G::ROWS * 2) {                       // left halo slot and the zero slot behind the last pixel
            const int r = tid >> 1;
            *reinterpret_cast<u32x2*>(smem + r * WSB + ((tid & 1) ? (WIN + 1) * 8 : 0)) = u32x2{0u, 0u};
        }
    }
    float st3[3][2][4], st1[3][2][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) st3[t][k][e] = st1[t][k][e] = 0.f;
    __syncthreads();

    bf16_t* y3 = reinterpret_cast<bf16_t*>(d.y3);
    bf16_t* y1 = reinterpret_cast<bf16_t*>(d.y1);
    const size_t obase = ((size_t)n * HO + r0) * WOUT * COUT;
#pragma unroll 1
    for (int f = wid; f < G::NFRAG; f += G::NT / 64) {
        const int orow = f / (WOUT / 16), ox = (f - orow * (WOUT / 16)) * 16 + px;
        const char* p0 = smem + (2 * orow + (g >> 1)) * WSB + 16 * ox + 16 * (g & 1);
        const char* p1 = smem + (2 * orow + 2) * WSB + 16 * ox + 16 * (g & 1);
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(p0);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(p1);
        // outputs: staged per wave as [tensor][16 pixels][96 B], then three fully coalesced 16-byte store instructions (a fragment's
        // 16 pixels x 48 channels are 1536 consecutive bytes of each output) with the non-temporal hint: written once, re-read much later
        char* ost = smem + G::WINB + wid * G::OST + px * G::OPITCH + g * 8;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            f32x4 c3 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = f32x4{0.f, 0.f, 0.f, 0.f};
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a3[t][0]), b0, c3, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a3[t][1]), b1, c3, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1[t]), b0, c1, 0, 0, 0);
            if (d.stats3 != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st3[t][0][e] += c3[e]; st3[t][1][e] += c3[e] * c3[e];
                    st1[t][0][e] += c1[e]; st1[t][1][e] += c1[e] * c1[e];
                }
            }
            *reinterpret_cast<u32x2*>(ost + 32 * t) = u32x2{pack_bf16x2(c3[0], c3[1]), pack_bf16x2(c3[2], c3[3])};
            *reinterpret_cast<u32x2*>(ost + 16 * G::OPITCH + 32 * t) = u32x2{pack_bf16x2(c1[0], c1[1]), pack_bf16x2(c1[2], c1[3])};
        }
        if (!(a.dbg & 1)) {
            const size_t fbase = obase + ((size_t)orow * WOUT + (f - orow * (WOUT / 16)) * 16) * COUT;     // first element of the fragment
            const char* rd = smem + G::WINB + wid * G::OST;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = i * 64 + lane, tz = q / 96, r = q - tz * 96, pp = r / 6, c = r - pp * 6;
                const char* src = rd + tz * 16 * G::OPITCH + pp * G::OPITCH + c * 16;
                const u32x2 lo = *reinterpret_cast<const u32x2*>(src), hi = *reinterpret_cast<const u32x2*>(src + 8);
                bf16_t* dst = (tz ? y1 : y3) + fbase + pp * COUT + c * 8;
                __builtin_nontemporal_store(u32x4{lo[0], lo[1], hi[0], hi[1]}, reinterpret_cast<u32x4*>(dst));
            }
        }
    }
    if (d.stats3 != nullptr) {
        const size_t slot = (size_t)((blockIdx.x * (G::NT / 64) + wid) % a.reps) * 2 * COUT;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float m3 = 0.f, m1 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s3 = row16_sum(st3[t][k][e]), s1 = row16_sum(st1[t][k][e]);
                    m3 = px == e ? s3 : m3;
                    m1 = px == e ? s1 : m1;
                }
                if (px < 4) {
                    atomicAdd(d.stats3 + slot + k * COUT + 16 * t + 4 * g + px, m3);
                    atomicAdd(d.stats1 + slot + k * COUT + 16 * t + 4 * g + px, m1);
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------ data gradient
// dx[iy][ix][ci] of the same block: sum over the taps whose stride-2 footprint hits (iy, ix).  By output parity (py, px) = (iy & 1,
// ix & 1), with a = iy >> 1, b = ix >> 1:
//   (0, 0): dy3[a][b] . W3[1][1]  +  dy1[a][b] . W1                                      2 taps
//   (0, 1): dy3[a][b + 1] . W3[1][0]  +  dy3[a][b] . W3[1][2]                            2 taps
//   (1, 0): dy3[a + 1][b] . W3[0][1]  +  dy3[a][b] . W3[2][1]                            2 taps
//   (1, 1): dy3[a + 1][b + 1] . W3[0][0] + dy3[a + 1][b] . W3[0][2] + dy3[a][b + 1] . W3[2][0] + dy3[a][b] . W3[2][2]      4 taps
// One K stream of 10 taps x Cout / 8 pieces in exactly this order (pack mode 7): every class is a whole number of k32 steps, its
// pixels are CONSECUTIVE dy pixels (stride 1 in LDS, conflict free for odd slot sizes), and a workgroup that owns R rows of dy
// (+ one halo row / column, zero outside) produces 2 R complete rows of dx.  A wave owns 16 input channels, weights in registers.
template <int CO, int CI, int WDX, int R>
struct DGeo {
    static constexpr int PT = CO / 8;
    static constexpr int S = 10 * PT / 4;                // k32 steps of the whole stream
    static constexpr int PSC = PT + 1, PS = PSC * 16;
    static constexpr int WO = WDX / 2, WS = WO + 1;      // dy slots per window row: WO pixels + right halo
    static constexpr int N3 = (R + 1) * WS, N1 = R * WS; // slots of the dy3 / dy1 windows
    static constexpr int D3 = (N3 * PS + 1023) / 1024, D1 = (N1 * PS + 1023) / 1024;
    static constexpr int OFF1 = D3 * 1024;               // dy1 window behind the dy3 window
    static constexpr int NPIX = R * WO, NFRAG = (NPIX + 15) / 16;
    static constexpr int NW = CI / 16, NT = 64 * NW;
    static constexpr int SMEM = (D3 + D1) * 1024 + 256;
    static_assert((PSC & 1) == 1 && (10 * PT) % 4 == 0 && (2 * PT) % 4 == 0, "K stream");
};

template <int CO, int CI, int WDX, int R>
__global__ __launch_bounds__(64 * (CI / 16)) void s2_dgrad_kernel(const hc_conv_s2_dgrad_desc d, const int ntiles, const int dbg) {
    // persistent like s2_fwd_persist_kernel: weights once per workgroup, the dy windows of block i + 1 in flight under block i
    using G = DGeo<CO, CI, WDX, R>;
    constexpr int PT = G::PT, S = G::S, PS = G::PS, WS = G::WS, WO = G::WO;
    constexpr int BUF = (G::D3 + G::D1) * 1024;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, g = lane >> 4;
    const int H = d.H, HO = H / 2, RB = HO / R;
    const unsigned lds0 = hc_lds_addr(smem);
    const int xcd = blockIdx.x & 7, jloc = blockIdx.x >> 3, GX = gridDim.x >> 3;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int t_begin = xcd * q8 + (xcd < r8 ? xcd : r8), t_count = q8 + (xcd < r8 ? 1 : 0);
    const unsigned bytes = (unsigned)d.N * HO * WO * CO * 2u;
    const u32x4 rs3 = hc_raw_rsrc(d.dy3, bytes), rs1 = hc_raw_rsrc(d.dy1, bytes);

    auto stage = [&](int tl, int buf) __attribute__((always_inline)) {
        const int tile = t_begin + tl;
        const int n = tile / RB, r0 = (tile - n * RB) * R;
        const unsigned img = (unsigned)n * (unsigned)(HO * WO * CO * 2);
        for (int j = wid; j < G::D3 + G::D1; j += G::NW) {
            const bool first = j < G::D3;
            const int J = (first ? j : j - G::D3) * 64 + lane;
            const int slot = J / G::PSC, c = J - slot * G::PSC;
            const int r = slot / WS, x = slot - r * WS;
            const bool ok = c < PT && x < WO && r0 + r < HO && slot < (first ? G::N3 : G::N1);
            const unsigned off = img + (unsigned)(((r0 + r) * WO + x) * CO * 2 + c * 16);
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * BUF + j * 1024));
            if (dbg & 4) continue;
            if (first) hc_dma16(rs3, dst, ok ? off : HC_OOB);
            else hc_dma16(rs1, dst, ok ? off : HC_OOB);
        }
    };
    if (jloc < t_count) stage(jloc, 0);

    const __amdgpu_buffer_rsrc_t rw = make_rsrc(d.wimg, (unsigned)(CI / 16 * S * 1024));
    u32x4 aw[S];
#pragma unroll
    for (int s = 0; s < S; ++s) aw[s] = buf_load16(rw, (unsigned)(lane * 16), (unsigned)((wid * S + s) * 1024));
    // byte offset of piece 4 s + g relative to the dy3 slot of (a, b): stream position pos -> (tensor, da, db)
    int boff[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int q = 4 * s + g, pos = q / PT, c = q - pos * PT;
        // pos:      0      1(dy1)  2      3      4      5      6      7      8      9
        const int da = (pos == 4 || pos == 6 || pos == 7) ? 1 : 0;
        const int db = (pos == 2 || pos == 6 || pos == 8) ? 1 : 0;
        boff[s] = (da * WS + db) * PS + c * 16 + (pos == 1 ? G::OFF1 : 0);
    }
    bf16_t* dx = reinterpret_cast<bf16_t*>(d.dx);
    constexpr int CS1 = 2 * PT / 4, CS2 = 4 * PT / 4, CS3 = 6 * PT / 4;     // first step of the parity classes 1, 2, 3
    constexpr int NF2 = (G::NFRAG + 1) / 2;
    int it = 0;
#pragma unroll 1
    for (int tl = jloc; tl < t_count; tl += GX, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tl + GX < t_count) stage(tl + GX, (it + 1) & 1);
        const int tile = t_begin + tl;
        const int n = tile / RB, r0 = (tile - n * RB) * R;
        const size_t obase = ((size_t)n * H + 2 * r0) * WDX * CI + 16 * wid + 4 * g;
        const char* wbase = smem + (it & 1) * BUF;
#pragma unroll 1
        for (int f2 = 0; f2 < NF2; ++f2) {
            int a_[2], b_[2];
            bool ok[2];
            const char* pb[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int p = (2 * f2 + u) * 16 + px;
                ok[u] = p < G::NPIX;
                const int pc = ok[u] ? p : G::NPIX - 1;
                a_[u] = pc / WO;
                b_[u] = pc - a_[u] * WO;
                pb[u] = wbase + (a_[u] * WS + b_[u]) * PS;
            }
            f32x4 acc[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[u][k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int k = s < CS1 ? 0 : s < CS2 ? 1 : s < CS3 ? 2 : 3;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bf16x8 bv = *reinterpret_cast<const bf16x8*>(pb[u] + boff[s]);
                    acc[u][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aw[s]), bv, acc[u][k], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (ok[u] && !(dbg & 1)) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const size_t o = obase + ((size_t)(2 * a_[u] + (k >> 1)) * WDX + 2 * b_[u] + (k & 1)) * CI;
                        *reinterpret_cast<u32x2*>(dx + o) = u32x2{pack_bf16x2(acc[u][k][0], acc[u][k][1]), pack_bf16x2(acc[u][k][2], acc[u][k][3])};
                    }
                }
                if ((dbg & 1) && acc[u][0][0] + acc[u][1][0] + acc[u][2][0] + acc[u][3][0] == 123.456f) dx[obase] = 0;
            }
        }
    }
}

template <int CO, int CI, int WDX, int R>
int launch_dgrad(const hc_conv_s2_dgrad_desc& d, hipStream_t st) {
    using G = DGeo<CO, CI, WDX, R>;
    auto kern = s2_dgrad_kernel<CO, CI, WDX, R>;
    constexpr int smem = 2 * (G::D3 + G::D1) * 1024 + 256;
    static_assert(smem <= 80 * 1024, "two workgroups per CU");
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        once = true;
    }
    static const int dbg = getenv("HC_CONV_S2_DBG") ? atoi(getenv("HC_CONV_S2_DBG")) : 0;
    const int ntiles = d.N * (d.H / 2 / R);
    int grid = 512;
    if (grid > ntiles) grid = (ntiles + 7) / 8 * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(G::NT), smem, st, d, ntiles, dbg);
    return hc_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------ stem weight gradient
// dW3[co][ci][kh][kw] = sum_p dy3[p][co] x[ci][2 oy + kh - 1][2 ox + kw - 1],  dW1[co][ci] = sum_p dy1[p][co] x[ci][2 oy][2 ox]
// straight from the fp32 image batch and the two NHWC gradients - no column tensor (the im2col path wrote 411 MB and read it twice).
// GEMM view: M = 48 output channels (x 2 tensors), N = 16 k-slots per kernel row (slot 4 kw + ci, the forward kernel's layout),
// K = output pixels.  Both operands have the pixel as their SLOW index in memory, which is what ds_read_b64_tr_b16 is for (lane a of
// a 16-lane group addresses k-row a >> 2, 8-byte column chunk a & 3; lane i receives column i - scripts/probes/tr16_probe.hip):
//   A = dy^T : LDS image [pixel][48] exactly as in HBM (96-byte pitch: 8 consecutive pixels fall into 8 distinct 32-byte bank slots),
//   B = window: a k-row is an output pixel, its four column chunks the input pixels 2 ox - 1 .. 2 ox + 2 of kernel row kh (the 4th
//       is the unused k-slot quad 12 .. 15).
// Persistent workgroups walk (image, row pair) tiles and keep 12 accumulator tiles (3 co x 3 kh for dW3, 3 co for dW1) per wave; every
// wave writes one fp32 slab, a second kernel adds the slabs in a fixed order (bit-reproducible) into the OIHW gradients.
struct SwArgs {
    const float* x;
    const void* dy3;
    const void* dy1;
    float* ws;
    int N, H, ntiles;
};
constexpr int SW_R = 2, SW_WIN = 224, SW_WOUT = 112, SW_CO = 48;
constexpr int SW_WSB = (SW_WIN + 2) * 8;                  // window row bytes (as in the forward kernel)
constexpr int SW_ROWS = 2 * SW_R + 1;
constexpr int SW_WINB = (SW_ROWS * SW_WSB + 1023) / 1024 * 1024;
constexpr int SW_NPIX = SW_R * SW_WOUT;                   // 224 pixels = 7 k32 steps
constexpr int SW_DYB = SW_NPIX * SW_CO * 2;               // bytes of one gradient tile (21504 = 21 KB: 21 DMA pieces)
constexpr int SW_DYP = (SW_DYB + 1023) / 1024;
constexpr int SW_SMEM = SW_WINB + 2 * SW_DYP * 1024 + 64;     // >= 4 * SW_SLAB * 4 = 49152: the end-of-kernel wave reduction reuses it
constexpr int SW_SLAB = 12 * 256;                         // floats per wave slab: 12 tiles of 16 x 16

__device__ __forceinline__ bf16x8 sw_tr_pair(const char* p0, const char* p1) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0), hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(256) void s2_stem_wgrad_kernel(const SwArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int la = lane & 15, kq = lane >> 4;
    const int H = a.H, HO = H / 2, RB = HO / SW_R;
    const unsigned lds0 = hc_lds_addr(smem);
    const unsigned dybytes = (unsigned)a.N * HO * SW_WOUT * SW_CO * 2u;
    const u32x4 rs3 = hc_raw_rsrc(a.dy3, dybytes), rs1 = hc_raw_rsrc(a.dy1, dybytes);
    f32x4 acc3[3][3], acc1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        acc1[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) acc3[c][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    char* sdy3 = smem + SW_WINB;
    char* sdy1 = sdy3 + SW_DYP * 1024;
#pragma unroll 1
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int n = tile / RB, r0 = (tile - n * RB) * SW_R;
        __syncthreads();                                  // everybody is done with the previous tile's LDS image
        // gradient tiles: R output rows are contiguous in memory -> linear DMA (the rounded tail reads out of range -> zeros)
        {
            const unsigned base = (unsigned)((n * HO + r0) * SW_WOUT) * (unsigned)(SW_CO * 2);
            for (int j = wid; j < 2 * SW_DYP; j += 4) {
                const bool first = j < SW_DYP;
                const int jj = first ? j : j - SW_DYP;
                const unsigned off = (unsigned)(jj * 1024 + lane * 16);
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(SW_WINB + j * 1024));
                if (first) hc_dma16(rs3, dst, off < (unsigned)SW_DYB ? base + off : HC_OOB);
                else hc_dma16(rs1, dst, off < (unsigned)SW_DYB ? base + off : HC_OOB);
            }
        }
        // input window: fp32 planes -> {c0, c1, c2, 0} bf16 slots (slot x = input column x - 1)
        {
            const float* xin = a.x + (size_t)n * 3 * H * SW_WIN;
            constexpr int IPR = SW_WIN / 4, NITEM = SW_ROWS * IPR;
            constexpr int NIT = (NITEM + 255) / 256;
            f32x4 v[NIT][3];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = it * 256 + tid;
                const int r = idx / IPR, c4 = idx - r * IPR;
                const int ih = 2 * r0 - 1 + r;
                const bool ok = idx < NITEM && ih >= 0;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci)
                    v[it][ci] = ok ? *reinterpret_cast<const f32x4*>(xin + ((size_t)ci * H + ih) * SW_WIN + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = it * 256 + tid;
                const int r = idx / IPR, c4 = idx - r * IPR;
                if (idx < NITEM) {
                    char* wp = smem + r * SW_WSB + (1 + 4 * c4) * 8;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        *reinterpret_cast<u32x2*>(wp + 8 * k) = u32x2{pack_bf16x2(v[it][0][k], v[it][1][k]), pack_bf16x2(v[it][2][k], 0.f)};
                }
            }
            if (tid < SW_ROWS * 2) {
                const int r = tid >> 1;
                *reinterpret_cast<u32x2*>(smem + r * SW_WSB + ((tid & 1) ? (SW_WIN + 1) * 8 : 0)) = u32x2{0u, 0u};
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // k32 steps of the tile, dealt to the four waves
        for (int gstep = wid; gstep < SW_NPIX / 32; gstep += 4) {
            const char* pa[2];
            const char* pbw[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int p = 32 * gstep + 16 * h + 4 * kq + (la >> 2);       // this lane's k-row (output pixel of the tile)
                const int ar = p / SW_WOUT, ox = p - ar * SW_WOUT;
                pa[h] = sdy3 + p * (SW_CO * 2) + 8 * (la & 3);
                pbw[h] = smem + (2 * ar) * SW_WSB + (2 * ox + (la & 3)) * 8;
            }
            bf16x8 b[3];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) b[kh] = sw_tr_pair(pbw[0] + kh * SW_WSB, pbw[1] + kh * SW_WSB);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const bf16x8 a3 = sw_tr_pair(pa[0] + 32 * c, pa[1] + 32 * c);
                const bf16x8 a1 = sw_tr_pair(pa[0] + (sdy1 - sdy3) + 32 * c, pa[1] + (sdy1 - sdy3) + 32 * c);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) acc3[c][kh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b[kh], acc3[c][kh], 0, 0, 0);
                acc1[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b[1], acc1[c], 0, 0, 0);
            }
        }
    }
    // slab of this workgroup: [12 tiles][row = co % 16][col = k slot]; lane (la = column, kq) holds rows 4 kq .. 4 kq + 3.  The four
    // waves are added through LDS in wave order (the tiles' LDS image is dead by now)
    __syncthreads();
    float* part = reinterpret_cast<float*>(smem) + wid * SW_SLAB;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[((c * 3 + kh) * 16 + 4 * kq + r) * 16 + la] = acc3[c][kh][r];
#pragma unroll
        for (int r = 0; r < 4; ++r) part[((9 + c) * 16 + 4 * kq + r) * 16 + la] = acc1[c][r];
    }
    __syncthreads();
    float* slab = a.ws + (size_t)blockIdx.x * SW_SLAB;
    const float* all = reinterpret_cast<const float*>(smem);
    for (int e = tid; e < SW_SLAB; e += 256) slab[e] = ((all[e] + all[SW_SLAB + e]) + all[2 * SW_SLAB + e]) + all[3 * SW_SLAB + e];
}

// dw3[co][ci][kh][kw] / dw1[co][ci] = (accumulate ? old : 0) + sum over the workgroup slabs.  One workgroup per 16 slab elements:
// thread (e, part) adds the slabs part, part + 16, ... (four loads in flight), the 16 partial sums are combined in a fixed order
__global__ __launch_bounds__(256) void s2_stem_wgrad_reduce_kernel(const float* __restrict__ ws, int nslab, float* __restrict__ dw3,
                                                                   float* __restrict__ dw1, int accumulate) {
    __shared__ float sm[16][17];
    const int el = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;                   // element of a slab
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = part;
    for (; k + 48 < nslab; k += 64) {
        s0 += ws[(size_t)k * SW_SLAB + e];
        s1 += ws[(size_t)(k + 16) * SW_SLAB + e];
        s2 += ws[(size_t)(k + 32) * SW_SLAB + e];
        s3 += ws[(size_t)(k + 48) * SW_SLAB + e];
    }
    for (; k < nslab; k += 16) s0 += ws[(size_t)k * SW_SLAB + e];
    sm[part][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (part != 0) return;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += sm[q][el];
    const int tile = e >> 8, row = (e >> 4) & 15, col = e & 15;
    const int kw = col >> 2, ci = col & 3;
    if (kw > 2 || ci > 2) return;
    if (tile < 9) {
        const int c = tile / 3, kh = tile - 3 * c, co = 16 * c + row;
        float* o = dw3 + ((co * 3 + ci) * 3 + kh) * 3 + kw;
        *o = accumulate ? *o + v : v;
    } else if (kw == 1) {
        const int co = 16 * (tile - 9) + row;
        float* o = dw1 + co * 3 + ci;
        *o = accumulate ? *o + v : v;
    }
}

// ------------------------------------------------------------------------------------------------------------------ stem, fused with its BatchNorm
// The stem block (repvgg.py:71-73 with in_channels = 3) is the one place where the conv is cheaper to REDO than to store: its input
// is 0.6 MB per image (fp32 planes), each of its two outputs 1.2 MB per image in bf16, and the BatchNorm passes around it (apply,
// backward reduce, backward apply) moved y3 / y1 / dy3 / dy1 eleven times per step - 3.4 GB of the step's 19 GB of BatchNorm traffic at
// batch 256.  Here y3 and y1 never exist in HBM; every pass recomputes the two convs from the image window with the SAME two / one
// MFMAs per fragment as s2_stem_kernel (same operands in the same k slots: the fp32 results are bit-identical from pass to pass, so
// the ReLU mask of the backward IS the forward's):
//   mode 0  statistics  image -> per-channel sum / sum of squares of the fp32 conv results                    154 MB read
//   mode 1  apply       image -> out = act(a3 c3 + a1 c1 + shift), NHWC bf16 (+ the statistics of `out`)      154 MB read, 308 MB written
//   mode 2  backward    image, g -> three small matrices from which the WHOLE backward of the block follows    462 MB read
// against 770 + 924 (conv, apply) + 924 + 1540 + 770 (reduce, apply, weight gradient) = 4.9 GB for the unfused sequence.  BatchNorm
// normalises the fp32 conv results here (the unfused path normalises their bf16 roundings): one rounding less per branch.
//
// Backward in ONE pass.  With dz = g [z > 0] (z recomputed), BatchNorm's backward gives dy_b = A_b dz + B_b c_b + C_b per branch b
// (A, B, C per channel from the sums sum dz, sum dz c_b: rep_bn_bwd_finalize_kernel), and the conv weight gradient is
// dW3[co][k] = sum_p dy3[p][co] xw[p][k] over the 27 window entries xw[p][k] of output pixel p.  Because c3 = W3 . xw is LINEAR in
// the window and the window is only 27 wide, every sum over pixels factors through
//   G [co][k]  = sum_p dz[p][co] xw[p][k]        48 x 27   (the weight gradient of dz; its centre-tap columns are the 1x1 conv's)
//   XX[k'][k]  = sum_p xw[p][k'] xw[p][k]        27 x 27   (Gram matrix of the windows)
//   and the window's pad slot {c0, c1, c2, PAD} set to 1 (its weights are zero, the convs do not see it): column (centre tap, pad) of
//   G is sum_p dz[p][co] and of XX is sum_p xw[p][k']
// namely  sum dz c3 = <W3[co], G[co]>,  dW3 = A3 G + B3 (W3 XX) + C3 sum xw,  likewise for the 1x1 conv on the centre-tap block.  The
// pass therefore needs no second sweep for the weight gradients (the two-sweep form read image + g twice: reduce, then a weight-gradient
// launch that re-formed dy3 / dy1) and no per-pixel products in the VALU (which bounded that form: 108 VALU operations per 16-pixel
// fragment at 4 cycles each against 9 MFMAs): dz is masked in place in the LDS gradient tile, G takes 9 and XX 6 MFMAs per 32 pixels -
// XX's two operands are the SAME registers (the B-operand layout of the window IS the A-operand layout of its transpose).
// stem_bwd_finalize_kernel turns (G, XX) into dgamma, dbeta, dW3, dW1 in fp32; nothing is rounded to bf16 on the way (the unfused path
// rounds dy3 / dy1 as MFMA operands), so the result is closer to the fp32 reference, not further.
// Persistent workgroups walk (image, row block) tiles in XCD-contiguous runs with the next tile's loads in flight.
struct SfArgs {
    const float* x;
    const void* w3img;
    const void* w1img;
    const float* coef;       // [4][48]: a3, a1, (unused), shift - hc_rep_bn_finalize's output          modes 1, 2
    const void* g;           // NHWC bf16 [N][112][112][48]                                              mode 2
    void* out;               // NHWC bf16                                                                mode 1
    float* acc0;             // mode 0: stats3 [reps][2][48]; mode 1: statistics of `out` [reps][2][48] or null; mode 2: slabs
    float* acc1;             // mode 0: stats1
    int N, H, ntiles, act, reps;
    int dbg;                 // HC_STEM_DBG timing knock-outs (results are wrong): 1 no G / XX stage, 4 no gradient DMA, 8 no conv stage
};

constexpr int SF_NT = 15;                                  // mode 2 accumulator tiles per wave: 9 of G (3 co x 3 kh), 6 of XX (kh' <= kh)
constexpr int SF_SLAB = SF_NT * 256;                       // floats per workgroup slab

template <int MODE>
struct SfGeo {
    static constexpr int R = MODE < 2 ? 4 : 2;                 // output rows per tile
    static constexpr int WSB = (224 + 2) * 8, ROWS = 2 * R + 1;
    static constexpr int WINB = (ROWS * WSB + 1023) / 1024 * 1024;
    static constexpr int NPIX = R * 112, NFRAG = NPIX / 16;
    static constexpr int GB = MODE == 2 ? (NPIX * 96 + 1023) / 1024 * 1024 : 0;       // gradient tile (masked in place: dz)
    static constexpr int GP = GB / 1024;
    static constexpr int NG = MODE == 2 ? 2 : 0;               // two gradient buffers: the next tile's is DMA'd while this one is worked on
    static constexpr int OPITCH = 104, OST = 16 * OPITCH;      // mode 1: one staged fragment per wave
    static constexpr int USED = WINB + NG * GB + (MODE == 1 ? 4 * OST : 0) + 64;
    static constexpr int SMEM = (MODE == 2 && USED < 4 * SF_SLAB * 4) ? 4 * SF_SLAB * 4 : USED;      // the end-of-kernel wave reduction reuses the tiles' LDS
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void stem_fused_kernel(const SfArgs a) {
    using G = SfGeo<MODE>;
    constexpr int R = G::R, WSB = G::WSB, ROWS = G::ROWS, CO = 48, WOUT = 112, WIN = 224;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, g = lane >> 4;
    const int H = a.H, HO = H / 2, RB = HO / R;
    char* const sgb = smem + G::WINB;             // gradient tiles [pixel][48] bf16

    // weights of all three 16-channel tiles of both convs (pack mode 6), kept for the workgroup's life
    const __amdgpu_buffer_rsrc_t rw3 = make_rsrc(a.w3img, 3u * 2u * 1024u);
    const __amdgpu_buffer_rsrc_t rw1 = make_rsrc(a.w1img, 3u * 1024u);
    u32x4 a3[3][2], a1[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        a3[t][0] = buf_load16(rw3, (unsigned)(lane * 16), (unsigned)((t * 2 + 0) * 1024));
        a3[t][1] = buf_load16(rw3, (unsigned)(lane * 16), (unsigned)((t * 2 + 1) * 1024));
        a1[t] = buf_load16(rw1, (unsigned)(lane * 16), (unsigned)(t * 1024));
    }
    // per-lane coefficients of its 12 channels (16 t + 4 g + e)
    f32x4 ka3[3], ka1[3], ksh[3];
    if (MODE >= 1) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            ka3[t] = *reinterpret_cast<const f32x4*>(a.coef + 16 * t + 4 * g);
            ka1[t] = *reinterpret_cast<const f32x4*>(a.coef + CO + 16 * t + 4 * g);
            ksh[t] = *reinterpret_cast<const f32x4*>(a.coef + 3 * CO + 16 * t + 4 * g);
        }
    }
    // The weights / coefficients must have LANDED before the first prefetch is issued: the compiler's wait-count pass otherwise finds
    // them behind the prefetch loads in the in-order vmcnt queue and puts `s_waitcnt vmcnt(0)` in front of the fragment loop's MFMAs -
    // every tile then waits for the NEXT tile's loads (seen in the ISA of the first prefetching version)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        asm volatile("" : "+v"(a3[t][0]), "+v"(a3[t][1]), "+v"(a1[t]));
        if (MODE >= 1) asm volatile("" : "+v"(ka3[t]), "+v"(ka1[t]), "+v"(ksh[t]));
    }
    // running sums: mode 0 {sum c3, sum c3^2, sum c1, sum c1^2}, mode 1 {sum out, sum out^2} of the bf16-rounded output (the identity
    // BatchNorm of the next block normalises exactly those values)
    constexpr int NK = MODE == 0 ? 4 : (MODE == 1 ? 2 : 1);
    f32x4 sm[NK][3];
    if (MODE <= 1) {
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int t = 0; t < 3; ++t) sm[k][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 accg[3][3], accx[6];                    // mode 2: G[co tile][kh] (16 co x 16 k slots each), XX[(kh', kh)] for kh' <= kh
    if (MODE == 2) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) accg[c][k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 6; ++i) accx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned lds0 = hc_lds_addr(smem);
    const u32x4 rsg = hc_raw_rsrc(a.g, MODE == 2 ? (unsigned)a.N * HO * WOUT * CO * 2u : 0u);

    // ---- tile loop.  The NEXT tile's operands are requested before this tile is worked on: the image window into registers (it is
    // converted to bf16 slots on the way into LDS, after this tile's last read of the window), the gradient tile by DMA into the other
    // gradient buffer.  Single-buffered, a workgroup spent most of a tile's time waiting for its loads.
    constexpr int IPR = WIN / 4, NITEM = ROWS * IPR, NIT = (NITEM + 255) / 256;
    f32x4 v[NIT][3];
    bool vin[NIT];                                // the item's row lies inside the image (its pad slot is 1 in mode 2)
    auto request = [&](const int L, const int gbuf) {
        const int tile = xcd_tile(L, a.ntiles);
        const int n = tile / RB, r0 = (tile - n * RB) * R;
        const float* xin = a.x + (size_t)n * 3 * H * WIN;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 256 + tid;
            const int r = idx / IPR, c4 = idx - r * IPR;
            const int ih = 2 * r0 - 1 + r;
            const bool ok = idx < NITEM && ih >= 0;
            vin[it] = ok;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
                v[it][ci] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xin + ((size_t)ci * H + ih) * WIN + 4 * c4))
                               : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (MODE == 2 && !(a.dbg & 4)) {                  // gradient tile: R output rows are contiguous in memory -> linear DMA
            const unsigned base = (unsigned)((n * HO + r0) * WOUT) * (unsigned)(CO * 2);
            for (int j = wid; j < G::GP; j += 4) {
                const unsigned off = (unsigned)(j * 1024 + lane * 16);
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(G::WINB + gbuf * G::GB + j * 1024));
                hc_dma16(rsg, dst, off < (unsigned)(G::NPIX * 96) ? base + off : HC_OOB);
            }
        }
    };
    int gcur = 0;
    if ((int)blockIdx.x < a.ntiles) request(blockIdx.x, 0);
#pragma unroll 1
    for (int L = blockIdx.x; L < a.ntiles; L += gridDim.x) {
        const int tile = xcd_tile(L, a.ntiles);
        const int n = tile / RB, r0 = (tile - n * RB) * R;
        char* const sg = sgb + gcur * G::GB;              // this tile's gradient (masked in place: dz)
        // window: fp32 planes -> {c0, c1, c2, pad} bf16 slots (slot x = input column x - 1), as in s2_stem_kernel; pad = 0, or 1 inside
        // the image in mode 2.  Nobody reads the window any more: the previous tile ended on a barrier
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 256 + tid;
            const int r = idx / IPR, c4 = idx - r * IPR;
            const float pad = (MODE == 2 && vin[it]) ? 1.f : 0.f;
            if (idx < NITEM) {
                char* wp = smem + r * WSB + (1 + 4 * c4) * 8;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<u32x2*>(wp + 8 * k) = u32x2{pack_bf16x2(v[it][0][k], v[it][1][k]), pack_bf16x2(v[it][2][k], pad)};
            }
        }
        if (tid < ROWS * 2) {                             // left halo slot and the zero slot behind the last pixel
            const int r = tid >> 1;
            *reinterpret_cast<u32x2*>(smem + r * WSB + ((tid & 1) ? (WIN + 1) * 8 : 0)) = u32x2{0u, 0u};
        }
        if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this tile's gradient has landed
        __syncthreads();
        if (L + (int)gridDim.x < a.ntiles) request(L + gridDim.x, gcur ^ 1);

        // ---- the convs of this tile, 16 pixels x 48 channels x 2 per fragment
#pragma unroll 1
        for (int f = wid; f < ((a.dbg & 8) ? 0 : G::NFRAG); f += 4) {
            const int orow = f / (WOUT / 16), fx = f - orow * (WOUT / 16), ox = fx * 16 + px;
            const char* p0 = smem + (2 * orow + (g >> 1)) * WSB + 16 * ox + 16 * (g & 1);
            const char* p1 = smem + (2 * orow + 2) * WSB + 16 * ox + 16 * (g & 1);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(p0);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(p1);
            const int pt = orow * WOUT + ox;              // this lane's pixel of the tile
            char* ost = smem + G::WINB + wid * G::OST + px * G::OPITCH + g * 8;       // mode 1
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                f32x4 c3 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = f32x4{0.f, 0.f, 0.f, 0.f};
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a3[t][0]), b0, c3, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a3[t][1]), b1, c3, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1[t]), b0, c1, 0, 0, 0);
                if (MODE == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sm[0][t][e] += c3[e]; sm[1][t][e] += c3[e] * c3[e];
                        sm[2][t][e] += c1[e]; sm[3][t][e] += c1[e] * c1[e];
                    }
                    continue;
                }
                // pre-activation: the fma chain of rep_preact<false> (csrc/rep_bn.hip), so that every pass rounds alike
                float z[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = __builtin_fmaf(ka3[t][e], c3[e], __builtin_fmaf(ka1[t][e], c1[e], ksh[t][e]));
                if (MODE == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) z[e] = (a.act == 1 && !(z[e] > 0.f)) ? 0.f : z[e];
                    const u32x2 pk = u32x2{pack_bf16x2(z[0], z[1]), pack_bf16x2(z[2], z[3])};
                    *reinterpret_cast<u32x2*>(ost + 32 * t) = pk;
                    if (a.acc0 != nullptr) {
                        const float r[4] = {bf16lo(pk[0]), bf16hi(pk[0]), bf16lo(pk[1]), bf16hi(pk[1])};
#pragma unroll
                        for (int e = 0; e < 4; ++e) { sm[0][t][e] += r[e]; sm[1][t][e] += r[e] * r[e]; }
                    }
                    continue;
                }
                // mode 2: dz = g [z > 0], masked in place (two bf16 per word: nothing is re-rounded)
                if (a.act == 1) {
                    char* gp = sg + pt * (CO * 2) + 32 * t + 8 * g;
                    const u32x2 gw = *reinterpret_cast<const u32x2*>(gp);
                    const unsigned m0 = (z[0] > 0.f ? 0x0000ffffu : 0u) | (z[1] > 0.f ? 0xffff0000u : 0u);
                    const unsigned m1 = (z[2] > 0.f ? 0x0000ffffu : 0u) | (z[3] > 0.f ? 0xffff0000u : 0u);
                    *reinterpret_cast<u32x2*>(gp) = u32x2{gw[0] & m0, gw[1] & m1};
                }
            }
            if (MODE == 1) {
                // the fragment's 16 pixels x 48 channels are 1536 consecutive bytes of `out`: 96 coalesced 16-byte stores
                bf16_t* out = reinterpret_cast<bf16_t*>(a.out);
                const size_t fbase = (((size_t)n * HO + r0 + orow) * WOUT + fx * 16) * CO;
                const char* rd = smem + G::WINB + wid * G::OST;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int q = i * 64 + lane, pp = q / 6, c = q - pp * 6;
                    if (q < 96) {
                        const char* src = rd + pp * G::OPITCH + c * 16;
                        const u32x2 lo = *reinterpret_cast<const u32x2*>(src), hi = *reinterpret_cast<const u32x2*>(src + 8);
                        __builtin_nontemporal_store(u32x4{lo[0], lo[1], hi[0], hi[1]}, reinterpret_cast<u32x4*>(out + fbase + pp * CO + c * 8));
                    }
                }
            }
        }
        if (MODE == 2) {
            __syncthreads();                              // dz of the whole tile is in LDS
            // G and XX over the tile's pixels, k32 steps dealt to the four waves: A = dz^T (transposing reads of the [pixel][48] tile,
            // as in s2_stem_wgrad_kernel), B = the window rows; XX multiplies the window registers with themselves
            const int la = lane & 15, kq = lane >> 4;
            for (int gstep = wid; gstep < ((a.dbg & 1) ? 0 : G::NPIX / 32); gstep += 4) {
                const char* pa[2];
                const char* pbw[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int p = 32 * gstep + 16 * h + 4 * kq + (la >> 2);       // this lane's k-row (output pixel of the tile)
                    const int ar = p / WOUT, ox = p - ar * WOUT;
                    pa[h] = sg + p * (CO * 2) + 8 * (la & 3);
                    pbw[h] = smem + (2 * ar) * WSB + (2 * ox + (la & 3)) * 8;
                }
                bf16x8 b[3];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) b[kh] = sw_tr_pair(pbw[0] + kh * WSB, pbw[1] + kh * WSB);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const bf16x8 fz = sw_tr_pair(pa[0] + 32 * c, pa[1] + 32 * c);
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) accg[c][kh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fz, b[kh], accg[c][kh], 0, 0, 0);
                }
                accx[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0], b[0], accx[0], 0, 0, 0);
                accx[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0], b[1], accx[1], 0, 0, 0);
                accx[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0], b[2], accx[2], 0, 0, 0);
                accx[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1], b[1], accx[3], 0, 0, 0);
                accx[4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1], b[2], accx[4], 0, 0, 0);
                accx[5] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[2], b[2], accx[5], 0, 0, 0);
            }
        }
        __syncthreads();                                  // everybody is done with this tile's window and gradient buffer
        gcur ^= 1;
    }

    if (MODE == 0 || (MODE == 1 && a.acc0 != nullptr)) {
        // fold the 16 pixel lanes of a DPP row, lanes px < 4 of every row own channel 16 t + 4 g + px; one slot per wave
        const size_t slot = (size_t)((blockIdx.x * 4 + wid) % a.reps);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                float m = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s = row16_sum(sm[k][t][e]);
                    m = px == e ? s : m;
                }
                if (px < 4) {
                    const int ch = 16 * t + 4 * g + px;
                    if (MODE == 0) atomicAdd((k < 2 ? a.acc0 : a.acc1) + slot * 2 * CO + (k & 1) * CO + ch, m);
                    else atomicAdd(a.acc0 + slot * 2 * CO + k * CO + ch, m);
                }
            }
    }
    if (MODE == 2) {
        // slab of this workgroup: [15 tiles][row][col], D layout: lane (la = column, kq) holds rows 4 kq .. 4 kq + 3.  Tiles 0 .. 8:
        // G[co tile c][kh] (row = co % 16, col = k slot 4 kw + ci | pad), tiles 9 .. 14: XX (row = slot of kernel row kh', col = slot
        // of kernel row kh) for (kh', kh) = (0,0) (0,1) (0,2) (1,1) (1,2) (2,2).  The four waves are added through LDS in wave order.
        const int la = lane & 15, kq = lane >> 4;
        __syncthreads();
        float* part = reinterpret_cast<float*>(smem) + wid * SF_SLAB;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[((c * 3 + kh) * 16 + 4 * kq + r) * 16 + la] = accg[c][kh][r];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[((9 + i) * 16 + 4 * kq + r) * 16 + la] = accx[i][r];
        __syncthreads();
        float* slab = a.acc0 + (size_t)blockIdx.x * SF_SLAB;
        const float* all = reinterpret_cast<const float*>(smem);
        for (int e = tid; e < SF_SLAB; e += 256) slab[e] = ((all[e] + all[SF_SLAB + e]) + all[2 * SF_SLAB + e]) + all[3 * SF_SLAB + e];
    }
}

// S[e] = sum over the workgroup slabs, in a fixed order (one workgroup per 16 elements, as s2_stem_wgrad_reduce_kernel)
__global__ __launch_bounds__(256) void stem_bwd_slab_sum_kernel(const float* __restrict__ ws, int nslab, float* __restrict__ S) {
    __shared__ float sm[16][17];
    const int el = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = part;
    for (; k + 48 < nslab; k += 64) {
        s0 += ws[(size_t)k * SF_SLAB + e];
        s1 += ws[(size_t)(k + 16) * SF_SLAB + e];
        s2 += ws[(size_t)(k + 32) * SF_SLAB + e];
        s3 += ws[(size_t)(k + 48) * SF_SLAB + e];
    }
    for (; k < nslab; k += 16) s0 += ws[(size_t)k * SF_SLAB + e];
    sm[part][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (part != 0) return;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += sm[q][el];
    S[e] = v;
}

// (G, XX) -> everything the block's backward returns.  One workgroup; thread = (co, j) with j = one of the 27 3x3 weights, or a
// per-channel job.  Weights enter as the bf16 roundings the MFMAs multiplied with (RNE of the fp32 master, like the packer).
struct SfFin {
    const float* S;          // [15][16][16]
    const float* w3;         // fp32 master [48][3][3][3]
    const float* w1;         // [48][3]
    const float* save;       // [6][48]: mean3, invstd3, mean1, invstd1, ...
    const float* gamma3;
    const float* gamma1;
    float* dgamma3; float* dbeta3; float* dgamma1; float* dbeta1;
    float* dw3; float* dw1;
    float count;
    int frozen, accumulate;
};
__device__ __forceinline__ float sf_bf16r(float f) { return bf16_to_f32(f32_to_bf16(f)); }
__device__ __forceinline__ float sf_xx(const float* S, int kh1, int s1, int kh2, int s2) {      // XX[(kh1, s1)][(kh2, s2)] (symmetric; tiles kh1 <= kh2)
    if (kh1 > kh2) { const int t = kh1; kh1 = kh2; kh2 = t; const int u = s1; s1 = s2; s2 = u; }
    const int tile = 9 + (kh1 == 0 ? kh2 : (kh1 == 1 ? 2 + kh2 : 5));
    return S[(tile * 16 + s1) * 16 + s2];
}
__global__ __launch_bounds__(256) void stem_bwd_finalize_kernel(const SfFin f) {
    __shared__ float sA[2][48], sB[2][48], sC[2][48];
    // the sums and the (bf16-rounded) weights once into LDS: every output below is a 27-term dot product over them, and out of global
    // memory each term was a dependent L2 round trip (16 us for 1.4 k outputs at the very end of the backward pass)
    __shared__ float sS[SF_NT * 256], sw3[48 * 27], sw1[48 * 3];
    const int tid = threadIdx.x;
    for (int i = tid; i < SF_NT * 256 / 4; i += 256) reinterpret_cast<f32x4*>(sS)[i] = reinterpret_cast<const f32x4*>(f.S)[i];
    for (int i = tid; i < 48 * 27; i += 256) sw3[i] = sf_bf16r(f.w3[i]);
    if (tid < 48 * 3) sw1[tid] = sf_bf16r(f.w1[tid]);
    __syncthreads();
    const float* S = sS;
    auto gcol = [&](int co, int kh, int slot) { return S[(((co >> 4) * 3 + kh) * 16 + (co & 15)) * 16 + slot]; };
    if (tid < 96) {
        const int b = tid / 48, co = tid - 48 * b;              // branch 0: 3x3, 1: 1x1
        const float sdz = gcol(co, 1, 7);                        // centre tap, pad slot: sum of dz
        float sdzy = 0.f;
        if (b == 0) {
            for (int ci = 0; ci < 3; ++ci)
                for (int kh = 0; kh < 3; ++kh)
                    for (int kw = 0; kw < 3; ++kw) sdzy += sw3[((co * 3 + ci) * 3 + kh) * 3 + kw] * gcol(co, kh, 4 * kw + ci);
        } else {
            for (int ci = 0; ci < 3; ++ci) sdzy += sw1[co * 3 + ci] * gcol(co, 1, 4 + ci);
        }
        const float mean = f.save[(2 * b) * 48 + co], invstd = f.save[(2 * b + 1) * 48 + co];
        const float gam = (b == 0 ? f.gamma3 : f.gamma1)[co];
        const float dgamma = invstd * (sdzy - mean * sdz);       // the expressions of rep_bn_bwd_finalize_kernel
        const float av = gam * invstd;
        float B = 0.f, Cc = 0.f;
        if (!f.frozen) {
            B = -av * invstd * dgamma / f.count;
            Cc = -av * sdz / f.count - B * mean;
        }
        sA[b][co] = av; sB[b][co] = B; sC[b][co] = Cc;
        float* dg = b == 0 ? f.dgamma3 : f.dgamma1;
        float* db = b == 0 ? f.dbeta3 : f.dbeta1;
        if (dg != nullptr) dg[co] = f.accumulate ? dg[co] + dgamma : dgamma;
        if (db != nullptr) db[co] = f.accumulate ? db[co] + sdz : sdz;
    }
    __syncthreads();
    for (int o = tid; o < 48 * 27 + 48 * 3; o += 256) {
        if (o < 48 * 27) {
            const int co = o / 27, j = o - 27 * co, ci = j / 9, kh = (j - 9 * ci) / 3, kw = j - 9 * ci - 3 * kh, slot = 4 * kw + ci;
            float wxx = 0.f;                                     // sum_k' W3[co][k'] XX[k'][k]
            for (int ci2 = 0; ci2 < 3; ++ci2)
                for (int kh2 = 0; kh2 < 3; ++kh2)
                    for (int kw2 = 0; kw2 < 3; ++kw2)
                        wxx += sw3[((co * 3 + ci2) * 3 + kh2) * 3 + kw2] * sf_xx(S, kh2, 4 * kw2 + ci2, kh, slot);
            const float v = sA[0][co] * gcol(co, kh, slot) + sB[0][co] * wxx + sC[0][co] * sf_xx(S, kh, slot, 1, 7);
            f.dw3[o] = f.accumulate ? f.dw3[o] + v : v;
        } else {
            const int q = o - 48 * 27, co = q / 3, ci = q - 3 * co, slot = 4 + ci;
            float wxx = 0.f;
            for (int ci2 = 0; ci2 < 3; ++ci2) wxx += sw1[co * 3 + ci2] * sf_xx(S, 1, 4 + ci2, 1, slot);
            const float v = sA[1][co] * gcol(co, 1, slot) + sB[1][co] * wxx + sC[1][co] * sf_xx(S, 1, slot, 1, 7);
            f.dw1[q] = f.accumulate ? f.dw1[q] + v : v;
        }
    }
}

template <typename K>
void set_smem(K kern, int smem) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
}

template <int CIN, int COUT, int WIN, int R, int CTW>
int launch_persist(const Args& a, hipStream_t st, int wg_per_cu) {
    using G = Geo<CIN, COUT, WIN, R, CTW>;
    auto kern = s2_fwd_persist_kernel<CIN, COUT, WIN, R, CTW>;
    constexpr int smem = 2 * G::WINB + 256;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static bool once = false;
    if (!once) { set_smem(kern, smem); once = true; }
    const int ntiles = a.d.N * (a.d.H / 2 / R);
    int grid = 256 * wg_per_cu;
    if (grid > ntiles) grid = (ntiles + 7) / 8 * 8;
    if (a.d.stats3 != nullptr && hc_get_deterministic() && grid * G::NW > a.reps) return HC_ERR_ARG;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(G::NT), smem, st, a, ntiles);
    return hc_launch_status();
}

template <int R>
int launch_stem(const Args& a, hipStream_t st) {
    using G = StemGeo<R>;
    auto kern = s2_stem_kernel<R>;
    static bool once = false;
    if (!once) { set_smem(kern, G::SMEM); once = true; }
    const int grid = a.d.N * (a.d.H / 2 / R);
    if (a.d.stats3 != nullptr && hc_get_deterministic() && grid * (G::NT / 64) > a.reps) return HC_ERR_ARG;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(G::NT), G::SMEM, st, a);
    return hc_launch_status();
}

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e != nullptr ? atoi(e) : dflt;
}

}  // namespace cs2

extern "C" int hc_conv_s2_supported(const hc_conv_s2_desc* dp) {
    static const bool on = cs2::env_int("HC_CONV_S2", 1) != 0;
    if (!on || dp == nullptr) return 0;
    const hc_conv_s2_desc& d = *dp;
    if (d.N < 1 || d.H != d.W) return 0;
    if ((double)d.N * d.H * d.W * (d.x_nchw_f32 ? 12.0 : d.Cin * 2.0) >= 4294967000.0) return 0;
    if ((double)d.N * (d.H / 2) * (d.W / 2) * d.Cout * 2.0 >= 4294967000.0) return 0;
    if (d.x_nchw_f32) return d.Cin == 3 && d.Cout == 48 && d.W == 224 && d.H % 16 == 0;
    if (d.Cin == 48 && d.Cout == 48 && d.W == 112) return d.H % 8 == 0;
    if (d.Cin == 48 && d.Cout == 96 && d.W == 56) return d.H % 8 == 0;
    return 0;
}

extern "C" int hc_conv_s2_dgrad_supported(const hc_conv_s2_dgrad_desc* dp) {
    static const bool on = cs2::env_int("HC_CONV_S2", 1) != 0;
    if (!on || dp == nullptr) return 0;
    const hc_conv_s2_dgrad_desc& d = *dp;
    if (d.N < 1 || d.H != d.W || d.H % 8 != 0) return 0;
    if ((double)d.N * d.H * d.W * d.Cin * 2.0 >= 4294967000.0 || (double)d.N * (d.H / 2) * (d.W / 2) * d.Cout * 2.0 >= 4294967000.0) return 0;
    return (d.Cin == 48 && d.Cout == 48 && d.W == 112) || (d.Cin == 48 && d.Cout == 96 && d.W == 56);
}

extern "C" int hc_conv_s2_dgrad(const hc_conv_s2_dgrad_desc* dp, hc_stream_t stream) {
    if (!hc_conv_s2_dgrad_supported(dp)) return HC_ERR_ARG;
    const hc_conv_s2_dgrad_desc& d = *dp;
    if (d.dy3 == nullptr || d.dy1 == nullptr || d.wimg == nullptr || d.dx == nullptr) return HC_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d.Cout == 48) return cs2::launch_dgrad<48, 48, 112, 2>(d, st);
    return cs2::launch_dgrad<96, 48, 56, 2>(d, st);
}

constexpr int HC_S2_STEM_WGRAD_GRID = 768;        // three 52 KB workgroups per CU

extern "C" int64_t hc_conv_s2_stem_wgrad_ws_bytes(void) { return (int64_t)HC_S2_STEM_WGRAD_GRID * cs2::SW_SLAB * 4; }

extern "C" int hc_conv_s2_stem_wgrad(const float* x, const void* dy3, const void* dy1, float* dw3, float* dw1, void* ws, int32_t N,
                                     int32_t H, int32_t W, int32_t accumulate, hc_stream_t stream) {
    static const bool on = cs2::env_int("HC_CONV_S2", 1) != 0;
    if (!on) return HC_ERR_ARG;
    if (x == nullptr || dy3 == nullptr || dy1 == nullptr || dw3 == nullptr || dw1 == nullptr || ws == nullptr) return HC_ERR_ARG;
    if (N < 1 || W != 224 || H != 224 || (double)N * H * W * 12.0 >= 4294967000.0) return HC_ERR_ARG;
    cs2::SwArgs a;
    a.x = x; a.dy3 = dy3; a.dy1 = dy1; a.ws = reinterpret_cast<float*>(ws);
    a.N = N; a.H = H; a.ntiles = N * (H / 2 / cs2::SW_R);
    int grid = HC_S2_STEM_WGRAD_GRID;
    if (grid > a.ntiles) grid = a.ntiles;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static bool once = false;
    if (!once) { cs2::set_smem(cs2::s2_stem_wgrad_kernel, cs2::SW_SMEM); once = true; }
    hipLaunchKernelGGL(cs2::s2_stem_wgrad_kernel, dim3(grid), dim3(256), cs2::SW_SMEM, st, a);
    hipLaunchKernelGGL(cs2::s2_stem_wgrad_reduce_kernel, dim3(cs2::SW_SLAB / 16), dim3(256), 0, st, a.ws, grid, dw3, dw1, accumulate);
    return hc_launch_status();
}

extern "C" int hc_conv_s2_fwd(const hc_conv_s2_desc* dp, hc_stream_t stream) {
    if (!hc_conv_s2_supported(dp)) return HC_ERR_ARG;
    const hc_conv_s2_desc& d = *dp;
    if (d.x == nullptr || d.w3img == nullptr || d.w1img == nullptr || d.y3 == nullptr || d.y1 == nullptr) return HC_ERR_ARG;
    if ((d.stats3 == nullptr) != (d.stats1 == nullptr)) return HC_ERR_ARG;
    cs2::Args a;
    a.d = d;
    a.reps = hc_get_stat_replicas();
    a.dbg = cs2::env_int("HC_CONV_S2_DBG", 0);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int rsel = cs2::env_int("HC_CONV_S2_R", 0);       // 0 = default rows per workgroup, 1 = the smaller variant (read per call: tests flip it)
    if (d.x_nchw_f32) return rsel ? cs2::launch_stem<8>(a, st) : cs2::launch_stem<4>(a, st);
    if (d.Cout == 48) return rsel ? cs2::launch_persist<48, 48, 112, 2, 1>(a, st, 1) : cs2::launch_persist<48, 48, 112, 1, 1>(a, st, 2);
    return rsel ? cs2::launch_persist<48, 96, 56, 4, 2>(a, st, 1) : cs2::launch_persist<48, 96, 56, 2, 2>(a, st, 2);
}

// ---- the stem block fused with its BatchNorm passes (stem_fused_kernel): y3 / y1 are never stored
namespace cs2 {
constexpr int SF_MAX_GRID = 1024;

template <int MODE>
int launch_stem_fused(SfArgs& a, hipStream_t st) {
    using G = SfGeo<MODE>;
    auto kern = stem_fused_kernel<MODE>;
    // persistent workgroups = what is RESIDENT (two or three per CU, by registers and LDS): one more per CU would start when the first
    // finishes and leave most of the chip idle behind it
    static int resident = 0;
    if (resident == 0) {
        set_smem(kern, G::SMEM);
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, G::SMEM) != hipSuccess || per_cu < 1) per_cu = 2;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
        resident = per_cu * prop.multiProcessorCount;
        if (resident > SF_MAX_GRID) resident = SF_MAX_GRID;
        resident = resident / 8 * 8;
    }
    a.ntiles = a.N * (a.H / 2 / G::R);
    int grid = resident;
    if (grid > a.ntiles) grid = a.ntiles;
    a.reps = hc_get_stat_replicas();
    a.dbg = env_int("HC_STEM_DBG", 0);
    if ((MODE == 0 || (MODE == 1 && a.acc0 != nullptr)) && hc_get_deterministic() && grid * 4 > a.reps) return -1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), G::SMEM, st, a);
    return grid;
}

static bool sf_fill(SfArgs& a, const hc_stem_desc* dp) {
    if (!hc_stem_fused_supported(dp) || dp->x == nullptr || dp->w3img == nullptr || dp->w1img == nullptr) return false;
    a = SfArgs{};
    a.x = dp->x; a.w3img = dp->w3img; a.w1img = dp->w1img; a.N = dp->N; a.H = dp->H;
    return true;
}
}  // namespace cs2

extern "C" int hc_stem_fused_supported(const hc_stem_desc* dp) {
    static const bool on = cs2::env_int("HC_CONV_S2", 1) != 0 && cs2::env_int("HC_STEM_FUSED", 1) != 0;
    if (!on || dp == nullptr) return 0;
    if (dp->N < 1 || dp->W != 224 || dp->H < 16 || dp->H % 16 != 0) return 0;
    // 32-bit buffer descriptors / byte offsets: the fp32 image (12 B per pixel) AND, for the apply / backward passes, the bf16 output and
    // gradient of 48 channels per output pixel (2 x the image's bytes: the tighter bound - ADVICE r5)
    return (double)dp->N * dp->H * dp->W * 12.0 < 4294967000.0 &&
           (double)dp->N * (dp->H / 2) * (dp->W / 2) * 48.0 * 2.0 < 4294967000.0;
}

extern "C" int hc_stem_stats(const hc_stem_desc* dp, float* stats3, float* stats1, hc_stream_t stream) {
    cs2::SfArgs a;
    if (!cs2::sf_fill(a, dp) || stats3 == nullptr || stats1 == nullptr) return HC_ERR_ARG;
    a.acc0 = stats3; a.acc1 = stats1;
    if (cs2::launch_stem_fused<0>(a, reinterpret_cast<hipStream_t>(stream)) < 0) return HC_ERR_ARG;
    return hc_launch_status();
}

extern "C" int hc_stem_apply(const hc_stem_desc* dp, const float* coef, int32_t act, void* out, float* out_stats, hc_stream_t stream) {
    cs2::SfArgs a;
    if (!cs2::sf_fill(a, dp) || coef == nullptr || out == nullptr) return HC_ERR_ARG;
    a.coef = coef; a.act = act; a.out = out; a.acc0 = out_stats;
    if (cs2::launch_stem_fused<1>(a, reinterpret_cast<hipStream_t>(stream)) < 0) return HC_ERR_ARG;
    return hc_launch_status();
}

extern "C" int64_t hc_stem_bwd_ws_bytes(void) { return ((int64_t)cs2::SF_MAX_GRID + 1) * cs2::SF_SLAB * 4; }

extern "C" int hc_stem_bwd(const hc_stem_desc* dp, const hc_stem_bwd_desc* bp, hc_stream_t stream) {
    cs2::SfArgs a;
    if (!cs2::sf_fill(a, dp) || bp == nullptr) return HC_ERR_ARG;
    const hc_stem_bwd_desc& b = *bp;
    if (b.coef == nullptr || b.g == nullptr || b.save == nullptr || b.gamma3 == nullptr || b.gamma1 == nullptr || b.w3 == nullptr ||
        b.w1 == nullptr || b.dw3 == nullptr || b.dw1 == nullptr || b.ws == nullptr)
        return HC_ERR_ARG;
    a.coef = b.coef; a.act = b.act; a.g = b.g; a.acc0 = reinterpret_cast<float*>(b.ws);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = cs2::launch_stem_fused<2>(a, st);
    if (grid < 0) return HC_ERR_ARG;
    float* S = a.acc0 + (size_t)cs2::SF_MAX_GRID * cs2::SF_SLAB;
    hipLaunchKernelGGL(cs2::stem_bwd_slab_sum_kernel, dim3(cs2::SF_SLAB / 16), dim3(256), 0, st, a.acc0, grid, S);
    cs2::SfFin f;
    f.S = S; f.w3 = b.w3; f.w1 = b.w1; f.save = b.save; f.gamma3 = b.gamma3; f.gamma1 = b.gamma1;
    f.dgamma3 = b.dgamma3; f.dbeta3 = b.dbeta3; f.dgamma1 = b.dgamma1; f.dbeta1 = b.dbeta1; f.dw3 = b.dw3; f.dw1 = b.dw1;
    f.count = (float)((double)dp->N * (dp->H / 2) * (dp->W / 2)); f.frozen = b.frozen; f.accumulate = b.accumulate;
    hipLaunchKernelGGL(cs2::stem_bwd_finalize_kernel, dim3(1), dim3(256), 0, st, f);
    return hc_launch_status();
}
