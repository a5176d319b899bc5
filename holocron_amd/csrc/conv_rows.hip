// Stride-1 3x3 (+ 1x1) convolution of a RepBlock for the mid-size stages - 192 channels @ 14x14 and 96 channels @ 28x28 of RepVGG-A0
// (reference: RepBlock.forward, holocron/models/classification/repvgg.py:71-73, and its data gradient) - built around ROW UNITS.
// Same contract as the other hc_conv_small kernels (hc_conv_small_desc):
//   mode 0 (forward) : out3 = W3 (*) A, out1 = W1 . A (+ BN statistics of both)
//   mode 1 (dgrad)   : out3 = W3 (*) A + W1 . B + resid      (A = dy3, B = dy1, weights flipped by the packer)
//
// Why another kernel: the image-resident kernel (conv_resident.hip) runs one 8-wave workgroup per image in lockstep - stage the
// window, 30 barrier-separated k-steps, store - and a knock-out timing (HC_CRS_DBG) put 21 of its 59 us into the exposed window
// load and store phases and the k-loop at 38 us against 24 us of matrix-pipe time; 23 % of its MFMA columns are padding (14-wide rows
// in 16-wide halves of 32-pixel tiles, 7 of 8 tiles used).  Here
//   * the unit of work is 7 output rows of one image (two units per 14x14 image, four per 28x28 image); its 9-row input window
//     (zero halo, natural NHWC + one 16-byte pad chunk per pixel) is DMA'd into LDS.  A workgroup owns an image; its eight waves
//     form two TEAMS of four, each team walks every other unit with its own window (two waves per SIMD: one team's LDS / VMEM issue
//     runs under the other team's MFMAs) and synchronises only with itself (LDS arrival counter, no workgroup barrier);
//   * the MFMA is v_mfma_f32_16x16x32_bf16 with D[co][pixel]: one 16-pixel ROW SEGMENT per B fragment (14/16 resp. 28/32 columns
//     valid = 87.5 %, no idle tiles), 3 x 16 output channels per wave -> 21 MFMAs per k32 step against 3 weight loads + 14 ds_read_b64;
//   * every wave owns its 48 output channels, so nobody shares its weights: they go straight from L2 into registers, two k32 steps
//     ahead, as ONE fully coalesced 1 KB buffer_load_dwordx4 per fragment out of a weight image laid out for exactly that
//     (hc_pack_conv_weight modes 3 / 4).  No LDS ring, no weight barrier: the k-loop has no barrier at all (the window is read-only
//     between the two barriers of a unit).  Measured on the way: private LDS rings fed by buffer_load ... lds with one step of
//     look-ahead ran at DMA latency per step (41 us per image even with 3 workgroups on the chip), and 8-byte loads from the
//     [channel][tap][C] image touch 16 cache lines per quarter wave, which made the texture addresser the bottleneck (95 us);
//   * pixel fragments are read one step ahead into a second register set; a sched_barrier keeps the requests of the coming steps in
//     front of the current step's MFMAs (left alone, the scheduler sinks every read next to its first use).
// LDS reads are 8-byte pieces: the k index of a lane's 8 values is (4 g .. 4 g + 3, 16 + 4 g .. 16 + 4 g + 3) of the 32-channel block for A
// and B alike, i.e. two pieces 32 bytes apart; with a pixel pitch of 2 x odd 8-byte units (400 B for 192 channels, 208 B for 96) the 16
// pixels x 2 k-groups of a half wave fall into 32 distinct 8-byte bank pairs.  (ds_read_b128 services lanes {0-3,12-15,20-27}
// together, which mixes two k-groups in one bank cycle and cannot be made conflict-free for this fragment shape.)
// Output channels are permuted inside a wave's 48 so that lane group g ends up with channels 8 g .. 8 g + 7 and 32 + 4 g .. 32 + 4 g + 3
// of its pixel: one 16-byte and one 8-byte store, 64 + 32 contiguous bytes per pixel over the four groups, instead of three 8-byte
// pieces 32 bytes apart.
// MI355X, batch 256 (scripts/check_rows.py, profiles/r02_final_conv_rows_shapes.txt): 192 @ 14x14 forward + statistics 40 us
// (920 TFLOP/s; image-resident kernel 62.7), data gradient 44 us (62.7); 96 @ 28x28 46 / 56 us (gather-conv: 3 launches, 195 us).
#include <type_traits>
#include "common.h"
#include "../../include/holocron_hip.h"

namespace crw {

constexpr int NT = 512, UR = 7;

struct Args {
    hc_conv_small_desc d;
    int reps;          // statistics replicas
    int delay;         // start-up delay of the second team (s_sleep rounds): de-phases the two teams of a workgroup
};

// W > 0: the map width is a compile-time constant (the two RepVGG-A0 stages at 224 x 224: their codegen is what rounds 2-3 tuned).
// W == 0: the FAMILY form - any width up to 16 SEGW pixels (SEGW = 1 for 192 channels, 2 for 96) and any height (ragged last unit):
// the window keeps the pitch of the widest map, the width is a kernel argument.  RepVGG-A0 at any input resolution stays on this
// kernel (192 @ 12 x 12 / 16 x 16, 96 @ 24 x 24 / 32 x 32, ...).
template <int C, int W>
struct Geo {
    static constexpr int CK = C / 32;                      // k32 steps per tap
    static constexpr int PSC = C / 8 + 1;                  // 16-byte chunks per window slot (one pad chunk)
    static constexpr int PS = PSC * 16;                    // bytes per window slot
    static constexpr int SEGW = W > 0 ? (W + 15) / 16 : (C >= 192 ? 1 : 2);   // 16-pixel segments per row = pixel waves of a team
    static constexpr int WMAX = W > 0 ? W : 16 * SEGW;
    static constexpr int WW = WMAX + 1;                    // slots per window row: left halo + W pixels (the right halo is the next row's left)
    static constexpr int NSLOT = (UR + 2) * WW + 1;
    static constexpr int NDMA = (NSLOT * PS + 1023) / 1024;   // 1 KB DMA instructions per window
    static constexpr int WIN = NDMA * 1024;
    static constexpr int CWN = 4 / SEGW;                   // channel waves of a team
    static constexpr int SMEM = 2 * WIN + 64 + 1024;       // one window per team + the two team counters + slack: the fragment reads of
                                                           // discarded columns run up to a few hundred bytes past the second window
    static constexpr int S3 = 9 * CK, S1 = CK, S = S3 + S1;
    static_assert(C == 48 * CWN, "one wave = 48 output channels");
    static_assert(CK % 3 == 0, "three rotating weight register sets per tap");
    static_assert((PS / 8) % 4 == 2, "pixel pitch must be 2 x odd 8-byte units");
    static_assert(SMEM <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ void dma16s(const u32x4 rsrc, unsigned lds_off, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_off), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ u32x4 uniform_rsrc(const void* p, unsigned bytes) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)v);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ bf16x8 frag2(const char* lo_p, const char* hi_p) {       // two 8-byte pieces
    const u32x2 lo = *reinterpret_cast<const u32x2*>(lo_p), hi = *reinterpret_cast<const u32x2*>(hi_p);
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}
// sum over the 16 lanes of a DPP row (rotations by 8, 4, 2, 1: every lane ends up with the row total), no LDS traffic
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

// DBG (timing knock-outs, HC_CRW_DBG; results are wrong): 1 no MFMA, 2 no fragment reads, 4 no weight DMA, 8 no window staging,
// 16 no epilogue
template <int C, int W, int MODE, int DBG>
__global__ __launch_bounds__(NT, 1) void conv_rows_kernel(const Args a) {
    using G = Geo<C, W>;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const hc_conv_small_desc& d = a.d;
    constexpr int CK = G::CK, PS = G::PS, WW = G::WW, S3 = G::S3, S1 = G::S1, S = G::S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wid >> 2, w4 = wid & 3;
    const int cw = w4 % G::CWN, pw = w4 / G::CWN;
    const int H = d.H, UPI = W > 0 ? H / UR : (H + UR - 1) / UR;
    const int Wr = W > 0 ? W : d.W;                        // map width: compile-time for the tuned instantiations
    const int n = blockIdx.x;
    const unsigned lds0 = hc_lds_addr(smem);
    const int px = lane & 15, g = lane >> 4;
    const int col = 16 * pw + px;
    const bool col_ok = col < Wr;
    const int region = team * G::WIN;
    // The two teams synchronise only among themselves (an LDS arrival counter per team instead of s_barrier) and the second one starts
    // late: one team's window wait / re-staging / store drain then runs under the other team's MFMAs instead of next to its own
    int* cnt = reinterpret_cast<int*>(smem + 2 * G::WIN) + team * 8;
    int gen = 0;
    if (tid < 16) reinterpret_cast<int*>(smem + 2 * G::WIN)[tid] = 0;
    __syncthreads();
    auto team_sync = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        gen += 4;
        if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gen) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    if (team == 1)
        for (int i = 0; i < a.delay; ++i) __builtin_amdgcn_s_sleep(64);

    const unsigned act_bytes = (unsigned)d.N * H * Wr * C * 2u;
    const u32x4 rsA = uniform_rsrc(d.srcA, act_bytes);
    const u32x4 rsB = uniform_rsrc(MODE == 1 ? d.srcB : d.srcA, act_bytes);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(d.w3, (unsigned)(S * C * 64));

    // ---- weights: straight from L2 into registers, two steps ahead (every wave owns its 48 output channels, nothing to share through
    // LDS).  The row-unit image (hc_pack_conv_weight modes 3 / 4) is [step][row][32]: the A fragment f of wave cw at step s is the 1 KB
    // at (s C + 48 cw + 16 f) * 64, lane (m = lane & 15, g) takes the 16 bytes at row m, piece g - one fully coalesced
    // buffer_load_dwordx4 per fragment.  (Loading from the [channel][tap][C] image instead costs two 8-byte loads per fragment that
    // touch 16 different cache lines per quarter wave: the texture addresser then takes longer than the MFMAs.)
    const unsigned wl = (unsigned)((48 * cw + (lane & 15)) * 64 + g * 16);
    u32x4 af[3][3];                                          // [step % 3][fragment]
    // Forward: the 1x1 goes FIRST (sequence steps [0, S1) = image steps [S3, S)), so that its output stores drain under the nine taps of
    // the 3x3 instead of behind the kernel's last MFMA (one exposed store burst per unit instead of two); the data gradient keeps the
    // image order (its 1x1 source is staged into the window after the last tap).
    auto load_a = [&](int buf, int s) __attribute__((always_inline)) {   // weight fragments of unit-local sequence step s
        if (DBG & 4) return;
        const int ps = MODE == 0 ? (s < S1 ? S3 + s : s - S1) : s;
#pragma unroll
        for (int f = 0; f < 3; ++f) af[buf][f] = buf_load16(rsw, wl, (unsigned)(ps * (C * 64) + f * 1024));
    };

    // ---- window DMA by the four waves of a team: slot (r, x) = r WW + x of a unit holds input pixel (row0 - 1 + r, x - 1); chunk PSC - 1
    // of a slot, the halo and the rounded tail are zero-filled through out-of-range offsets.  j0 .. j1: which 1 KB pieces.
    auto stage_window = [&](const u32x4 rs, int row0, int j0, int j1) __attribute__((always_inline)) {
        if (DBG & 8) return;
        const unsigned img = (unsigned)n * (unsigned)(H * Wr * C * 2);
        for (int j = j0 + w4; j < j1; j += 4) {
            const int J = j * 64 + lane;
            const int slot = J / G::PSC, c = J - slot * G::PSC;
            const int r = slot / WW, x = slot - r * WW;
            const int ih = row0 - 1 + r;
            const bool ok = c < G::PSC - 1 && slot < G::NSLOT && x >= 1 && (W > 0 || x <= Wr) && ih >= 0 && ih < H;
            const unsigned off = img + (unsigned)((ih * Wr + x - 1) * C * 2 + c * 16);
            hc_dma16(rs, lds0 + (unsigned)(region + j * 1024), ok ? off : HC_OOB);
        }
    };

    // ---- B fragments: pixel px of segment i (unit row i) at tap (dr, dc): slot (i + 1 + dr, col + 1 + dc), k-group g -> + 8 g (+ 32 for
    // the second piece)
    const int vb = region + (WW + col + 1) * PS + g * 8;
    bf16x8 bfr[2][UR];
    auto load_b = [&](int buf, int bofs) __attribute__((always_inline)) {
        if (DBG & 2) return;
        const char* sb = smem + vb + bofs;
#pragma unroll
        for (int i = 0; i < UR; ++i) bfr[buf][i] = frag2(sb + i * WW * PS, sb + i * WW * PS + 32);
    };

    f32x4 acc[3][UR];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int i = 0; i < UR; ++i) acc[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto mfma_step = [&](int ab, int bb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < UR; ++i)
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                if (DBG & 1) acc[f][i][0] += __builtin_bit_cast(float, af[ab][f][0]) * (float)bfr[bb][i][0];
                else acc[f][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[ab][f]), bfr[bb][i], acc[f][i], 0, 0, 0);
            }
    };
    // one k32 step (unit-local index s; weight registers s % 3, pixel registers s & 1): request the weights of step s + 2, read the
    // pixel fragments of step s + 1 (window offset bnext), multiply step s
    auto step = [&](int a3, int par, int s, int s_end, int bnext, bool prefetch) __attribute__((always_inline)) {
        if (s + 2 < s_end) load_a((a3 + 2) % 3, s + 2);
        if (prefetch) load_b(par ^ 1, bnext);
        __builtin_amdgcn_sched_barrier(0);                  // keep the requests of the coming steps IN FRONT of this step's MFMAs
        mfma_step(a3, par);
        __builtin_amdgcn_sched_barrier(0);
    };
    // the CK steps of one tap (CK % 3 == 0: the weight register set of a step is t % 3); P0 = parity of its first step
    auto tap_block = [&](auto p0, int s0, int s_end, int tofs, int tofs_next, bool last_prefetch) __attribute__((always_inline)) {
        constexpr int P0 = decltype(p0)::value;
#pragma unroll
        for (int t = 0; t < CK; ++t) {
            if (t + 1 < CK) step(t % 3, (P0 + t) & 1, s0 + t, s_end, tofs + (t + 1) * 64, true);
            else step(t % 3, (P0 + t) & 1, s0 + t, s_end, tofs_next, last_prefetch);
        }
    };

    // ---- epilogue of one unit and one output: lane (px, g) holds channels 48 cw + 12 g + 4 f + e of pixel (row0 + i, col)
    const int cbase = 48 * cw + 8 * g;       // first channel of this lane's 16-byte piece (see rows_image_index, rep_bn.hip)
    auto epilogue = [&](void* outp, float* stats, const void* residp, int row0) __attribute__((always_inline)) {
        if (DBG & 16) {
            float t = 0.f;
#pragma unroll
            for (int f = 0; f < 3; ++f)
#pragma unroll
                for (int i = 0; i < UR; ++i) t += acc[f][i][0];
            if (t == 123.456f) reinterpret_cast<float*>(outp)[tid] = t;
            return;
        }
        const size_t img = (size_t)n * H * Wr * C;
        __builtin_amdgcn_sched_barrier(0);     // the residual loads below stay below: hoisted into the last k-steps they spill the accumulators
        float st[2][12];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 12; ++e) st[k][e] = 0.f;
#pragma unroll
        for (int i = 0; i < UR; ++i) {
            const bool ok = col_ok && (W > 0 || row0 + i < H);          // family form: the last unit of an image may be ragged
            const int rr = W > 0 ? row0 + i : (ok ? row0 + i : 0);          // (compile-time form: the tuned instantiations' own address)
            const size_t e0 = img + (size_t)(rr * Wr + (col_ok ? col : 0)) * C + cbase;
            bf16_t* op = reinterpret_cast<bf16_t*>(outp) + e0;
            const bf16_t* rp = residp != nullptr ? reinterpret_cast<const bf16_t*>(residp) + e0 : nullptr;
            // channel pieces of this lane: 8 g + 4 f (f = 0, 1) and 32 + 4 g (f = 2).  Forward: one 16-byte + one 8-byte store; the data
            // gradient (residual loads, a staging loop and both sources in the same scheduling region) keeps three 8-byte accesses:
            // the 16-byte form there costs the register allocator 120 spilled accumulator registers
            u32x2 pk[3];
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                f32x4 v = acc[f][i];
                const int cofs = f < 2 ? 4 * f : 32 - 4 * g;
                if (stats != nullptr) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = ok ? v[e] : 0.f;
                        st[0][f * 4 + e] += x;
                        st[1][f * 4 + e] += x * x;
                    }
                }
                if (MODE == 1 && rp != nullptr && ok) {
                    const u32x2 rv = *reinterpret_cast<const u32x2*>(rp + cofs);
                    v[0] += bf16lo(rv[0]); v[1] += bf16hi(rv[0]); v[2] += bf16lo(rv[1]); v[3] += bf16hi(rv[1]);
                }
                pk[f] = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                if (MODE == 1 && ok) *reinterpret_cast<u32x2*>(op + cofs) = pk[f];
            }
            if (MODE == 0 && ok) {
                *reinterpret_cast<u32x4*>(op) = u32x4{pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                *reinterpret_cast<u32x2*>(op + 32 - 4 * g) = pk[2];
            }
        }
        // BN statistics of the fp32 results: fold the 16 pixel lanes of every k-group (DPP rotations: every lane gets the total), let
        // lane px keep channel px of its group's 12, and add 48 consecutive channels per instruction into this WAVE's replica slot (one
        // writer per slot and address when the replicas outnumber the waves of the grid: bit-reproducible)
        if (stats != nullptr) {
            float* rep = stats + (size_t)((blockIdx.x * 8 + wid) % a.reps) * 2 * C + cbase;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float mine = 0.f;
#pragma unroll
                for (int e = 0; e < 12; ++e) {
                    const float x = row16_sum(st[k][e]);
                    mine = px == e ? x : mine;
                }
                if (px < 12) atomicAdd(rep + k * C + (px < 8 ? px : 24 - 4 * g + px), mine);    // slot e -> channel 8 g + e | 32 + 4 g + e - 8
            }
        }
    };

    // ---- the image: the units of this team, one after the other (both teams walk the same number of units)
    for (int u = team; u < UPI; u += 2) {
        const int row0 = u * UR;
        if (u >= 2) team_sync();                            // every wave of the team is done reading the previous unit's window
        stage_window(rsA, row0, 0, G::NDMA);
        team_sync();                                        // the window has landed
        load_a(0, 0);
        load_a(1, 1);
        zero_acc();
        // taps in pairs so that the register-set / ring-stage parity of every step is a compile-time constant for odd CK too
        auto tap_ofs = [&](int tap) __attribute__((always_inline)) { return tap < 9 ? ((tap / 3 - 1) * WW + (tap % 3 - 1)) * PS : 0; };
        if (MODE == 0) {
            // 1x1 (centre tap) first, its epilogue (statistics + stores) is issued in front of the 3x3's 9 CK steps
            load_b(0, 0);
            tap_block(std::integral_constant<int, 0>{}, 0, S, 0, tap_ofs(0), true);
            epilogue(d.out1, d.stats1, nullptr, row0);
            zero_acc();
#pragma nounroll
            for (int tap = 0; tap < 8; tap += 2) {
                const int o0 = tap_ofs(tap), o1 = tap_ofs(tap + 1), o2 = tap_ofs(tap + 2);
                tap_block(std::integral_constant<int, (CK & 1)>{}, (tap + 1) * CK, S, o0, o1, true);
                tap_block(std::integral_constant<int, 0>{}, (tap + 2) * CK, S, o1, o2, true);
            }
            tap_block(std::integral_constant<int, (CK & 1)>{}, 9 * CK, S, tap_ofs(8), 0, false);
            epilogue(d.out3, d.stats3, nullptr, row0);
        } else {
            load_b(0, (-WW - 1) * PS);
            // the weight slices of a unit are one linear sequence of S3 + S1 steps (w1 directly follows w3's taps in the dgrad packing);
            // the ring runs two steps ahead over the phase boundary
#pragma nounroll
            for (int tap = 0; tap < 8; tap += 2) {
                const int o0 = tap_ofs(tap), o1 = tap_ofs(tap + 1), o2 = tap_ofs(tap + 2);
                tap_block(std::integral_constant<int, 0>{}, tap * CK, S, o0, o1, true);
                tap_block(std::integral_constant<int, (CK & 1)>{}, (tap + 1) * CK, S, o1, o2, true);
            }
            tap_block(std::integral_constant<int, 0>{}, 8 * CK, S, tap_ofs(8), 0, false);   // the 1x1 source is staged after the last tap: nothing to prefetch
            // second source through the same window: only the unit's own rows (slots WW .. 8 WW) are read by the centre tap
            team_sync();
            stage_window(rsB, row0, (WW * PS) / 1024, (8 * WW * PS + PS + 1023) / 1024);
            team_sync();
            load_b(S3 & 1, 0);
            tap_block(std::integral_constant<int, (S3 & 1)>{}, S3, S, 0, 0, false);
            epilogue(d.out3, nullptr, d.resid, row0);
        }
    }
}

template <int C, int W>
bool shape_ok(const hc_conv_small_desc& d) {
    if (W == 0)        // the family form: any width one team's pixel waves cover, any height
        return d.C == C && d.Cout == C && d.W >= 1 && d.W <= Geo<C, 0>::WMAX && d.H >= 1 && d.N >= 1 &&
               (double)d.N * d.H * d.W * C * 2.0 < 2147483000.0;
    return d.C == C && d.Cout == C && d.W == W && d.H >= 2 * UR && d.H % (2 * UR) == 0 && d.H <= 56 && d.N >= 1 &&
           (double)d.N * d.H * d.W * C * 2.0 < 2147483000.0;
}

template <int C, int W, int MODE, int DBG>
void launch1(const Args& a, hipStream_t st) {
    auto kern = conv_rows_kernel<C, W, MODE, DBG>;
    constexpr int smem = Geo<C, W>::SMEM;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.d.N), dim3(NT), smem, st, a);
}

template <int C, int W>
void launch(const Args& a, hipStream_t st) {
    static const int dbg = getenv("HC_CRW_DBG") ? atoi(getenv("HC_CRW_DBG")) : 0;
    const bool dg = (a.d.mode & 1) == 1;
#define CRW_CASE(k) case k: if (dg) launch1<C, W, 1, k>(a, st); else launch1<C, W, 0, k>(a, st); break;
    switch (dbg) {
        CRW_CASE(1) CRW_CASE(2) CRW_CASE(3) CRW_CASE(4) CRW_CASE(7) CRW_CASE(8) CRW_CASE(16) CRW_CASE(24) CRW_CASE(31)
        default: if (dg) launch1<C, W, 1, 0>(a, st); else launch1<C, W, 0, 0>(a, st); break;
    }
#undef CRW_CASE
}

}  // namespace crw

// called by hc_conv_small / hc_conv_small_supported (conv_small.hip); HC_CONV_ROWS=0 routes these shapes to the image-resident kernel
// (192 @ 14) or the gather-conv (96 @ 28) again
bool hc_conv_rows_supported(const hc_conv_small_desc& d) {
    static const bool on = [] { const char* e = getenv("HC_CONV_ROWS"); return e == nullptr || atoi(e) != 0; }();
    if (!on || (d.mode & HC_CONV_SMALL_ROWS_IMAGE) == 0 || (d.mode & ~(HC_CONV_SMALL_ROWS_IMAGE | 1)) != 0) return false;
    // a predicate, not a shape list: 192 channels up to 16 pixels wide, 96 channels up to 32, any height (HC_CONV_ROWS_ANY=0: only the
    // two tuned 224 x 224 stages, as in rounds 2-3)
    constexpr bool any = true;
    if (crw::shape_ok<192, 14>(d) || crw::shape_ok<96, 28>(d)) return true;
    return any && (crw::shape_ok<192, 0>(d) || crw::shape_ok<96, 0>(d));
}
int hc_conv_rows_launch(const hc_conv_small_desc& d, hipStream_t st) {
    if (!hc_conv_rows_supported(d)) return HC_ERR_ARG;
    const bool dg = (d.mode & 1) == 1;
    if (d.srcA == nullptr || d.w3 == nullptr || d.out3 == nullptr) return HC_ERR_ARG;
    if (!dg && d.out1 == nullptr) return HC_ERR_ARG;
    if (dg && d.srcB == nullptr) return HC_ERR_ARG;
    if (!dg && (d.stats3 == nullptr) != (d.stats1 == nullptr)) return HC_ERR_ARG;
    crw::Args a;
    a.d = d;
    a.reps = hc_get_stat_replicas();
    constexpr int delay = 0;                         // de-phasing the two teams measured nothing (rounds 3-4)
    a.delay = delay;
    if (crw::shape_ok<192, 14>(d)) crw::launch<192, 14>(a, st);
    else if (crw::shape_ok<96, 28>(d)) crw::launch<96, 28>(a, st);
    else if (d.C == 192) { if (dg) crw::launch1<192, 0, 1, 0>(a, st); else crw::launch1<192, 0, 0, 0>(a, st); }
    else { if (dg) crw::launch1<96, 0, 1, 0>(a, st); else crw::launch1<96, 0, 0, 0>(a, st); }
    return hc_launch_status();
}
