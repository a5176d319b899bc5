// Streaming row-unit convolution for the 48-channel stages of RepVGG-A0 (48 @ 112x112 and 48 @ 56x56, stride 1): fused 3x3 + 1x1
// forward (+ statistics) and data gradient (+ residual) of a RepBlock (reference: RepBlock.forward,
// holocron/models/classification/repvgg.py:71-73).  Same contract and weight image as conv_rows.hip (hc_conv_small_desc with
// HC_CONV_SMALL_ROWS_IMAGE; hc_pack_conv_weight modes 3 / 4 with the 48 input channels of a tap padded to two k32 steps).
//
// These layers are HBM-bound (0.9-1.2 GB per launch against 148 GFLOP): what counts is bytes in flight, not MFMA rate.  The kernel
// keeps the row-unit dataflow of conv_rows.hip - a unit of output rows, its zero-haloed input window DMA'd into LDS, 16-pixel row
// segments as MFMA columns, weights straight from L2 into registers - and changes what made that kernel a lockstep machine:
//   * persistent workgroups (one per CU) whose two 4-wave TEAMS walk their own sequence of units and synchronise only among
//     themselves (an LDS counter, not s_barrier): while one team waits for its window the other multiplies, so a CU always has a
//     window's worth of loads (64-76 KB) and a unit's stores in flight; the second team starts half a unit late on purpose;
//   * a unit is 4 rows of a 112-wide map (wave = one row = 7 segments) or 8 rows of a 56-wide map (wave = two rows = 8 segments);
//   * K = 48 per tap is two k32 steps, the second one half empty: its second 8-byte piece re-reads the first one against zero weights
//     (never garbage: 0 x Inf would be NaN).
#include <type_traits>
#include "common.h"
#include "../../include/holocron_hip.h"

namespace crq {

constexpr int NT = 512, C = 48, CK = 2, PSC = C / 8 + 1, PS = PSC * 16, S3 = 9 * CK, S1 = CK, S = S3 + S1;
static_assert((PS / 8) % 4 == 2, "pixel pitch must be 2 x odd 8-byte units");

struct Args {
    hc_conv_small_desc d;
    int reps;          // statistics replicas
    int nunits;        // N * H / UR
    int delay;         // start-up delay of the second team (s_sleep rounds)
};

template <int W>
struct Geo {
    static constexpr int RW = W > 64 ? 1 : 2;              // rows per wave
    static constexpr int SEGW = (W + 15) / 16;             // 16-pixel segments per row
    static constexpr int NF = RW * SEGW;                   // fragment columns per wave
    static constexpr int UR = 4 * RW;                      // rows per unit (four waves)
    static constexpr int WW = W + 1;
    static constexpr int NSLOT = (UR + 2) * WW + 1;
    static constexpr int NDMA = (NSLOT * PS + 1023) / 1024;
    static constexpr int WIN = NDMA * 1024;
    static constexpr int SMEM = 2 * WIN + 64 + 1024;       // one window per team + the two team counters + slack: the fragment reads of
                                                           // discarded columns run up to a few hundred bytes past the second window
    static_assert(SMEM <= 160 * 1024, "LDS budget");
};

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
__device__ __forceinline__ float row16_sum(float v) {      // sum over the 16 lanes of a DPP row, every lane gets it
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

// DBG (timing knock-outs, HC_CRQ_DBG; results are wrong): 1 no MFMA, 2 no fragment reads, 4 no weight loads, 8 no window staging,
// 16 no epilogue
template <int W, int MODE, int DBG>
__global__ __launch_bounds__(NT, 1) void conv_rows48_kernel(const Args a) {
    using G = Geo<W>;
    constexpr int WW = G::WW, NF = G::NF, UR = G::UR, RW = G::RW, SEGW = G::SEGW;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const hc_conv_small_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wid >> 2, pw = wid & 3;
    const int H = d.H, UPI = H / UR;
    const unsigned lds0 = hc_lds_addr(smem);
    const int px = lane & 15, g = lane >> 4;
    const int region = team * G::WIN;
    int* cnt = reinterpret_cast<int*>(smem + 2 * G::WIN) + team * 8;       // this team's arrival counter
    int gen = 0;

    if (tid < 16) reinterpret_cast<int*>(smem + 2 * G::WIN)[tid] = 0;
    __syncthreads();

    // four waves, one counter: arrive (after this wave's LDS reads / DMA have completed), then wait for the other three
    auto team_sync = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        gen += 4;
        if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gen) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };

    const unsigned act_bytes = (unsigned)d.N * H * W * C * 2u;
    const u32x4 rsA = uniform_rsrc(d.srcA, act_bytes);
    const u32x4 rsB = uniform_rsrc(MODE == 1 ? d.srcB : d.srcA, act_bytes);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(d.w3, (unsigned)(S * C * 64));

    // weights: [step][48 rows][32] image, fragment f of step s = the 1 KB at (48 s + 16 f) * 64; every wave of a team loads the same
    const unsigned wl = (unsigned)((lane & 15) * 64 + g * 16);
    u32x4 af[3][3];
    auto load_a = [&](int buf, int s) __attribute__((always_inline)) {
        if (DBG & 4) return;
#pragma unroll
        for (int f = 0; f < 3; ++f) af[buf][f] = buf_load16(rsw, wl, (unsigned)(s * (C * 64) + f * 1024));
    };

    // window DMA by the four waves of a team (layout of conv_rows.hip); j0 .. j1: which 1 KB pieces
    auto stage_window = [&](const u32x4 rs, int n, int row0, int j0, int j1) __attribute__((always_inline)) {
        if (DBG & 8) return;
        const unsigned img = (unsigned)n * (unsigned)(H * W * C * 2);
        for (int j = j0 + pw; j < j1; j += 4) {
            const int J = j * 64 + lane;
            const int slot = J / PSC, c = J - slot * PSC;
            const int r = slot / WW, x = slot - r * WW;
            const int ih = row0 - 1 + r;
            const bool ok = c < PSC - 1 && slot < G::NSLOT && x >= 1 && ih >= 0 && ih < H;
            const unsigned off = img + (unsigned)((ih * W + x - 1) * C * 2 + c * 16);
            hc_dma16(rs, lds0 + (unsigned)(region + j * 1024), ok ? off : HC_OOB);
        }
    };

    // B fragment column i = (row rw of this wave, segment sc): pixel (pw RW + rw, 16 sc + px); tap (dr, dc) shifts the slot
    const int vb = region + ((pw * RW + 1) * WW + px + 1) * PS + g * 8;
    bf16x8 bfr[2][NF];
    auto load_b = [&](int buf, int bofs, int second) __attribute__((always_inline)) {     // second: byte distance of the second piece
        if (DBG & 2) return;
        const char* sb = smem + vb + bofs;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int o = ((i / SEGW) * WW + 16 * (i % SEGW)) * PS;
            bfr[buf][i] = frag2(sb + o, sb + o + second);
        }
    };

    f32x4 acc[3][NF];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int i = 0; i < NF; ++i) acc[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto mfma_step = [&](int ab, int bb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                if (DBG & 1) acc[f][i][0] += __builtin_bit_cast(float, af[ab][f][0]) * (float)bfr[bb][i][0];
                else acc[f][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[ab][f]), bfr[bb][i], acc[f][i], 0, 0, 0);
            }
    };
    // step s (weights s % 3, pixels s & 1): request the weights of s + 2, read the pixels of s + 1, multiply s.
    // Step s = (tap, kk): window offset of the tap + 64 kk; kk = 1 holds channels 32..47 only -> second piece = first piece
    auto step = [&](int a3, int par, int s, int bnext, int second_next, bool prefetch) __attribute__((always_inline)) {
        if (s + 2 < S) load_a((a3 + 2) % 3, s + 2);
        if (prefetch) load_b(par ^ 1, bnext, second_next);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(a3, par);
        __builtin_amdgcn_sched_barrier(0);
    };

    const int cbase = 8 * g;                 // first channel of this lane's 16-byte piece (see rows_image_index, rep_bn.hip)
    auto epilogue = [&](void* outp, float* stats, const void* residp, int n, int row0) __attribute__((always_inline)) {
        if (DBG & 16) {
            float t = 0.f;
#pragma unroll
            for (int f = 0; f < 3; ++f)
#pragma unroll
                for (int i = 0; i < NF; ++i) t += acc[f][i][0];
            if (t == 123.456f) reinterpret_cast<float*>(outp)[tid] = t;
            return;
        }
        const size_t img = (size_t)n * H * W * C;
        __builtin_amdgcn_sched_barrier(0);     // the residual loads below stay below: hoisted into the last k-steps they spill the accumulators
        float st[2][12];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 12; ++e) st[k][e] = 0.f;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int row = row0 + pw * RW + i / SEGW, col = 16 * (i % SEGW) + px;
            const bool ok = col < W;
            const size_t e0 = img + (size_t)(row * W + (ok ? col : 0)) * C + cbase;
            bf16_t* op = reinterpret_cast<bf16_t*>(outp) + e0;
            const bf16_t* rp = residp != nullptr ? reinterpret_cast<const bf16_t*>(residp) + e0 : nullptr;
            // residual: the same two pieces (16 + 8 bytes) as the stores
            u32x4 r16 = {0u, 0u, 0u, 0u};
            u32x2 r8 = {0u, 0u};
            if (rp != nullptr && ok) {
                r16 = *reinterpret_cast<const u32x4*>(rp);
                r8 = *reinterpret_cast<const u32x2*>(rp + 32 - 4 * g);
            }
            u32x2 pk[3];
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                f32x4 v = acc[f][i];
                if (stats != nullptr) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = ok ? v[e] : 0.f;
                        st[0][f * 4 + e] += x;
                        st[1][f * 4 + e] += x * x;
                    }
                }
                const unsigned ra = f == 0 ? r16[0] : (f == 1 ? r16[2] : r8[0]), rb = f == 0 ? r16[1] : (f == 1 ? r16[3] : r8[1]);
                v[0] += bf16lo(ra); v[1] += bf16hi(ra); v[2] += bf16lo(rb); v[3] += bf16hi(rb);
                pk[f] = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
            if (ok) {         // channels 8 g .. 8 g + 7: 16 bytes (64 contiguous bytes over the four lane groups), 32 + 4 g ..: 8 bytes
                *reinterpret_cast<u32x4*>(op) = u32x4{pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                *reinterpret_cast<u32x2*>(op + 32 - 4 * g) = pk[2];
            }
        }
        if (stats != nullptr) {      // one slot per wave (conv_rows.hip); a wave's units add in program order
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

    // ---- the units of this team: u = 2 b + team, + 2 * gridDim.x, ...  (the two teams of a workgroup take adjacent units: their
    // windows share two rows in L2)
    if (team == 1)
        for (int i = 0; i < a.delay; ++i) __builtin_amdgcn_s_sleep(64);
    bool first = true;
    for (int u = 2 * blockIdx.x + team; u < a.nunits; u += 2 * gridDim.x) {
        const int n = u / UPI, row0 = (u - n * UPI) * UR;
        if (!first) team_sync();                            // every wave of the team is done reading the previous window
        first = false;
        stage_window(rsA, n, row0, 0, G::NDMA);
        load_a(0, 0);
        load_a(1, 1);
        team_sync();                                        // the window has landed
        zero_acc();
        load_b(0, (-WW - 1) * PS, 32);
        // 3x3: one kernel row (3 taps x 2 k-steps = 6 steps: a multiple of the 3 weight sets and of the 2 pixel sets) per iteration
#pragma nounroll
        for (int kh = 0; kh < 3; ++kh) {
            const int rofs = (kh - 1) * WW * PS;
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int kw = t >> 1, kk = t & 1;
                // next step: (kw, 1) after (kw, 0); (kw + 1, 0) after (kw, 1); first step of the next kernel row / of the 1x1 after t = 5
                int bnext, second;
                bool pf = true;
                if (kk == 0) { bnext = rofs + (kw - 1) * PS + 64; second = 0; }
                else if (t < 5) { bnext = rofs + kw * PS; second = 32; }
                else {
                    bnext = kh < 2 ? kh * WW * PS - PS : 0;
                    second = 32;
                    pf = kh < 2 || MODE == 0;               // dgrad: the 1x1 source is staged after the last tap
                }
                step(t % 3, kk, 6 * kh + t, bnext, second, pf);
            }
        }
        if (MODE == 0) {
            epilogue(d.out3, d.stats3, nullptr, n, row0);
            zero_acc();
        } else {
            team_sync();
            stage_window(rsB, n, row0, (WW * PS) / 1024, ((UR + 1) * WW * PS + PS + 1023) / 1024);
            team_sync();
            load_b(0, 0, 32);
        }
        step(0, 0, S3, 64, 0, true);
        step(1, 1, S3 + 1, 0, 0, false);
        if (MODE == 0) epilogue(d.out1, d.stats1, nullptr, n, row0);
        else epilogue(d.out3, nullptr, d.resid, n, row0);
    }
}

template <int W>
bool shape_ok(const hc_conv_small_desc& d) {
    return d.C == C && d.Cout == C && d.W == W && d.H >= Geo<W>::UR && d.H % Geo<W>::UR == 0 && d.N >= 1 &&
           (double)d.N * d.H * d.W * C * 2.0 < 2147483000.0;
}

template <int W, int MODE, int DBG>
void launch1(const Args& a, hipStream_t st) {
    auto kern = conv_rows48_kernel<W, MODE, DBG>;
    constexpr int smem = Geo<W>::SMEM;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    int grid = (a.nunits + 1) / 2;
    if (grid > 256) grid = 256;                             // one persistent workgroup per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), smem, st, a);
}

template <int W>
void launch(Args& a, hipStream_t st) {
    static const int dbg = getenv("HC_CRQ_DBG") ? atoi(getenv("HC_CRQ_DBG")) : 0;
    constexpr int delay = 2;
    a.nunits = a.d.N * (a.d.H / Geo<W>::UR);
    a.delay = delay;
    const bool dg = (a.d.mode & 1) == 1;
#define CRQ_CASE(k) case k: if (dg) launch1<W, 1, k>(a, st); else launch1<W, 0, k>(a, st); break;
    switch (dbg) {
        CRQ_CASE(1) CRQ_CASE(2) CRQ_CASE(4) CRQ_CASE(7) CRQ_CASE(8) CRQ_CASE(16) CRQ_CASE(24) CRQ_CASE(31)
        default: if (dg) launch1<W, 1, 0>(a, st); else launch1<W, 0, 0>(a, st); break;
    }
#undef CRQ_CASE
}

}  // namespace crq

// called by hc_conv_small / hc_conv_small_supported (conv_small.hip).  HC_CONV_ROWS48: 0 leaves these shapes to the persistent
// small-channel kernel, 1 (default) takes the forward only, 2 the data gradient too.  Measured at batch 256 (scripts/check_rows.py):
// forward + statistics 288 / 73 us (112 / 56 wide) against 351 / 92 us, data gradient 305 / 99 us against 325 / 97 us - and the
// whole step does not move with the data gradient on this kernel, so it stays where it was.
bool hc_conv_rows48_supported(const hc_conv_small_desc& d) {
    static const int on = [] { const char* e = getenv("HC_CONV_ROWS48"); return e == nullptr ? 1 : atoi(e); }();
    if (on <= 0 || (d.mode & HC_CONV_SMALL_ROWS_IMAGE) == 0 || (d.mode & ~(HC_CONV_SMALL_ROWS_IMAGE | 1)) != 0) return false;
    if ((d.mode & 1) && on < 2) return false;
    return crq::shape_ok<112>(d) || crq::shape_ok<56>(d);
}
int hc_conv_rows48_launch(const hc_conv_small_desc& d, hipStream_t st) {
    if (!hc_conv_rows48_supported(d)) return HC_ERR_ARG;
    const bool dg = (d.mode & 1) == 1;
    if (d.srcA == nullptr || d.w3 == nullptr || d.out3 == nullptr) return HC_ERR_ARG;
    if (!dg && d.out1 == nullptr) return HC_ERR_ARG;
    if (dg && d.srcB == nullptr) return HC_ERR_ARG;
    if (!dg && (d.stats3 == nullptr) != (d.stats1 == nullptr)) return HC_ERR_ARG;
    crq::Args a;
    a.d = d;
    a.reps = hc_get_stat_replicas();
    if (d.W == 112) crq::launch<112>(a, st);
    else crq::launch<56>(a, st);
    return hc_launch_status();
}
