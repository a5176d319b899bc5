// Pointwise / small-reduction kernels of the path: HardMish, pairwise box ops, greedy NMS,
// focal loss, label-smoothed cross entropy.  Compiled with -ffp-contract=off so that the fp32
// expression trees round exactly like the reference's separate torch kernels (needed for the
// bit-exact NMS / box-index contract, SURVEY.md §7 hard part 5).
#include "common.h"
#include "../../include/holocron_hip.h"

namespace {

inline int grid_for(long total, int threads = 256, int cap = 8192) {
    long b = (total + threads - 1) / threads;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// ---------------------------------------------------------------- hard_mish (functional.py:30-41)
__global__ void hard_mish_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float t = fminf(fmaxf(v + 2.f, 0.f), 2.f);
        y[i] = 0.5f * v * t;  // (0.5 * x) * clamp(x + 2, 0, 2)
    }
}
__global__ void hard_mish_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        // d/dx [0.5 x clamp(x+2,0,2)]: 0 for x<-2 ; x+1 for -2<=x<=0 ; 1 for x>0 (torch clamp passes grad on bounds)
        const float t = v + 2.f;
        float d = 0.5f * fminf(fmaxf(t, 0.f), 2.f);
        if (t >= 0.f && t <= 2.f) d += 0.5f * v;
        dx[i] = dy[i] * d;
    }
}

// ---------------------------------------------------------------- pairwise boxes (ops/boxes.py)
struct Box { float x1, y1, x2, y2; };
__device__ __forceinline__ float box_area(const Box b) { return (b.x2 - b.x1) * (b.y2 - b.y1); }
__device__ __forceinline__ float pair_iou(const Box a, const Box b, float* union_out) {
    const float area1 = box_area(a), area2 = box_area(b);
    const float ltx = fmaxf(a.x1, b.x1), lty = fmaxf(a.y1, b.y1);
    const float rbx = fminf(a.x2, b.x2), rby = fminf(a.y2, b.y2);
    const float w = fmaxf(rbx - ltx, 0.f), h = fmaxf(rby - lty, 0.f);
    const float inter = w * h;
    const float uni = (area1 + area2) - inter;
    if (union_out) *union_out = uni;
    return inter / uni;
}
__device__ __forceinline__ float pair_penalty(const Box a, const Box b) {
    // ops/boxes.py:79-103
    float cx = fmaxf(a.x2, b.x2) - fminf(a.x1, b.x1);
    float cy = fmaxf(a.y2, b.y2) - fminf(a.y1, b.y1);
    const float c2 = cx * cx + cy * cy;
    float dx = (a.x1 + a.x2) - (b.x1 + b.x2);
    float dy = (a.y1 + a.y2) - (b.y1 + b.y2);
    const float cd2 = (dx * dx + dy * dy) / 4.f;
    return cd2 / c2;
}
__global__ void box_pairwise_kernel(const float* __restrict__ b1, const float* __restrict__ b2, float* __restrict__ out, int M,
                                    int N, int kind) {
    const long total = (long)M * N;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int i = (int)(t / N), j = (int)(t % N);
        const Box a = {b1[4 * i], b1[4 * i + 1], b1[4 * i + 2], b1[4 * i + 3]};
        const Box b = {b2[4 * j], b2[4 * j + 1], b2[4 * j + 2], b2[4 * j + 3]};
        float r;
        if (kind == 0) {
            r = pair_iou(a, b, nullptr);
        } else if (kind == 1) {
            float uni;
            const float iou = pair_iou(a, b, &uni);
            const float w = fmaxf(fmaxf(a.x2, b.x2) - fminf(a.x1, b.x1), 0.f);
            const float h = fmaxf(fmaxf(a.y2, b.y2) - fminf(a.y1, b.y1), 0.f);
            const float area = w * h;
            r = iou - (area - uni) / area;
        } else if (kind == 2 || kind == 3) {
            // ciou_loss == diou_loss numerically: the alpha*v term is added to a temporary copy
            // in the reference (ops/boxes.py:208-209)
            r = (1.f - pair_iou(a, b, nullptr)) + pair_penalty(a, b);
        } else if (kind == 4) {
            r = pair_penalty(a, b);
        } else {
            const float v = atanf((a.x2 - a.x1) / (a.y2 - a.y1)) - atanf((b.x2 - b.x1) / (b.y2 - b.y1));
            r = (v * v) * (float)(4.0 / (3.141592653589793 * 3.141592653589793));
        }
        out[t] = r;
    }
}

// Gradient of box_pairwise_kernel: db1[i] += sum_j g[i][j] d r_ij / d b1[i], db2[j] += sum_i ... (the caller zeroes db1 / db2).
// Sub-gradient conventions are torch autograd's for the expressions of ops/boxes.py: binary max / min split a tie evenly,
// clamp(min=0) passes the gradient where the argument is >= 0.
__device__ __forceinline__ void max_adj(float x, float y, float g, float& gx, float& gy) {
    if (x > y) gx += g; else if (x < y) gy += g; else { gx += 0.5f * g; gy += 0.5f * g; }
}
__device__ __forceinline__ void min_adj(float x, float y, float g, float& gx, float& gy) {
    if (x < y) gx += g; else if (x > y) gy += g; else { gx += 0.5f * g; gy += 0.5f * g; }
}
__global__ void box_pairwise_bwd_kernel(const float* __restrict__ b1, const float* __restrict__ b2, const float* __restrict__ gout,
                                        float* __restrict__ db1, float* __restrict__ db2, int M, int N, int kind) {
    const long total = (long)M * N;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int i = (int)(t / N), j = (int)(t % N);
        const float g = gout[t];
        if (g == 0.f) continue;
        const Box a = {b1[4 * i], b1[4 * i + 1], b1[4 * i + 2], b1[4 * i + 3]};
        const Box b = {b2[4 * j], b2[4 * j + 1], b2[4 * j + 2], b2[4 * j + 3]};
        float ga[4] = {0.f, 0.f, 0.f, 0.f}, gb[4] = {0.f, 0.f, 0.f, 0.f};   // adjoints of (x1, y1, x2, y2)
        if (kind == 5) {   // 4/pi^2 (atan(w1/h1) - atan(w2/h2))^2
            const float w1 = a.x2 - a.x1, h1 = a.y2 - a.y1, w2 = b.x2 - b.x1, h2 = b.y2 - b.y1;
            const float q1 = w1 / h1, q2 = w2 / h2;
            const float v = atanf(q1) - atanf(q2);
            const float gv = g * 2.f * v * (float)(4.0 / (3.141592653589793 * 3.141592653589793));
            const float gq1 = gv / (1.f + q1 * q1), gq2 = -gv / (1.f + q2 * q2);
            const float gw1 = gq1 / h1, gh1 = -gq1 * w1 / (h1 * h1), gw2 = gq2 / h2, gh2 = -gq2 * w2 / (h2 * h2);
            ga[0] = -gw1; ga[2] = gw1; ga[1] = -gh1; ga[3] = gh1;
            gb[0] = -gw2; gb[2] = gw2; gb[1] = -gh2; gb[3] = gh2;
        } else {
            float g_iou = 0.f, g_pen = 0.f, g_uni = 0.f, g_C = 0.f;
            // forward pieces
            const float area1 = box_area(a), area2 = box_area(b);
            const float ltx = fmaxf(a.x1, b.x1), lty = fmaxf(a.y1, b.y1), rbx = fminf(a.x2, b.x2), rby = fminf(a.y2, b.y2);
            const float wr = rbx - ltx, hr = rby - lty;
            const float w = fmaxf(wr, 0.f), h = fmaxf(hr, 0.f);
            const float inter = w * h, uni = (area1 + area2) - inter;
            const float cwr = fmaxf(a.x2, b.x2) - fminf(a.x1, b.x1), chr_ = fmaxf(a.y2, b.y2) - fminf(a.y1, b.y1);
            if (kind == 0) g_iou = g;
            else if (kind == 1) {
                const float cw = fmaxf(cwr, 0.f), ch = fmaxf(chr_, 0.f), Carea = cw * ch;
                g_iou = g;
                g_uni = g / Carea;                        // giou = iou - 1 + uni / C
                g_C = -g * uni / (Carea * Carea);
                const float g_cw = cwr >= 0.f ? g_C * ch : 0.f, g_ch = chr_ >= 0.f ? g_C * cw : 0.f;
                max_adj(a.x2, b.x2, g_cw, ga[2], gb[2]);
                min_adj(a.x1, b.x1, -g_cw, ga[0], gb[0]);
                max_adj(a.y2, b.y2, g_ch, ga[3], gb[3]);
                min_adj(a.y1, b.y1, -g_ch, ga[1], gb[1]);
            } else if (kind == 2 || kind == 3) { g_iou = -g; g_pen = g; }
            else g_pen = g;
            if (g_pen != 0.f) {   // pen = cd2 / c2 (ops/boxes.py:79-103; no clamp on the enclosing box there)
                const float c2 = cwr * cwr + chr_ * chr_;
                const float dx = (a.x1 + a.x2) - (b.x1 + b.x2), dy = (a.y1 + a.y2) - (b.y1 + b.y2);
                const float cd2 = (dx * dx + dy * dy) / 4.f;
                const float g_cd2 = g_pen / c2, g_c2 = -g_pen * cd2 / (c2 * c2);
                const float g_dx = g_cd2 * dx / 2.f, g_dy = g_cd2 * dy / 2.f;
                ga[0] += g_dx; ga[2] += g_dx; gb[0] -= g_dx; gb[2] -= g_dx;
                ga[1] += g_dy; ga[3] += g_dy; gb[1] -= g_dy; gb[3] -= g_dy;
                const float g_cx = g_c2 * 2.f * cwr, g_cy = g_c2 * 2.f * chr_;
                max_adj(a.x2, b.x2, g_cx, ga[2], gb[2]);
                min_adj(a.x1, b.x1, -g_cx, ga[0], gb[0]);
                max_adj(a.y2, b.y2, g_cy, ga[3], gb[3]);
                min_adj(a.y1, b.y1, -g_cy, ga[1], gb[1]);
            }
            // iou = inter / uni
            float g_inter = g_iou / uni;
            g_uni += -g_iou * inter / (uni * uni);
            const float g_area1 = g_uni, g_area2 = g_uni;
            g_inter -= g_uni;
            const float g_wr = wr >= 0.f ? g_inter * h : 0.f, g_hr = hr >= 0.f ? g_inter * w : 0.f;
            min_adj(a.x2, b.x2, g_wr, ga[2], gb[2]);
            max_adj(a.x1, b.x1, -g_wr, ga[0], gb[0]);
            min_adj(a.y2, b.y2, g_hr, ga[3], gb[3]);
            max_adj(a.y1, b.y1, -g_hr, ga[1], gb[1]);
            const float aw = a.x2 - a.x1, ah = a.y2 - a.y1, bw = b.x2 - b.x1, bh = b.y2 - b.y1;
            ga[2] += g_area1 * ah; ga[0] -= g_area1 * ah; ga[3] += g_area1 * aw; ga[1] -= g_area1 * aw;
            gb[2] += g_area2 * bh; gb[0] -= g_area2 * bh; gb[3] += g_area2 * bw; gb[1] -= g_area2 * bw;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (ga[k] != 0.f) atomicAdd(db1 + 4 * i + k, ga[k]);
            if (gb[k] != 0.f) atomicAdd(db2 + 4 * j + k, gb[k]);
        }
    }
}

// ---------------------------------------------------------------- NMS (torchvision.ops.nms semantics)
// pass 1: mask[i][w] bit b set  <=>  j = 64*w + b > i  and  IoU(i, j) > thr.  A workgroup = four waves = four column blocks of one
// row block (a wave per 64 x 64 tile, its column boxes in its own LDS slice): a quarter of the workgroups of the one-wave form, whose
// launch was dominated by dispatching 3.5 M workgroups most of which (below the diagonal, past a small problem's end) exit at once.
__device__ __forceinline__ void nms_mask_body(const float* __restrict__ boxes, int n, float thr, unsigned long long* __restrict__ mask,
                                              int nw, const int rb, const int cb0) {
    __shared__ float sb[4][64 * 4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cb = cb0 + wv;
    const bool active = cb >= rb && cb < nw;             // only j > i matters
    const int jn = active ? min(64, n - cb * 64) : 0;
    if (lane < jn) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sb[wv][lane * 4 + e] = boxes[(long)(cb * 64 + lane) * 4 + e];
    }
    __syncthreads();
    const int i = rb * 64 + lane;
    if (!active || i >= n) return;
    const float x1 = boxes[(long)i * 4], y1 = boxes[(long)i * 4 + 1], x2 = boxes[(long)i * 4 + 2], y2 = boxes[(long)i * 4 + 3];
    const float iarea = (x2 - x1) * (y2 - y1);
    unsigned long long bits = 0;
    const int start = (rb == cb) ? lane + 1 : 0;
    for (int b = start; b < jn; ++b) {
        const float bx1 = sb[wv][b * 4], by1 = sb[wv][b * 4 + 1], bx2 = sb[wv][b * 4 + 2], by2 = sb[wv][b * 4 + 3];
        const float xx1 = fmaxf(x1, bx1), yy1 = fmaxf(y1, by1);
        const float xx2 = fminf(x2, bx2), yy2 = fminf(y2, by2);
        const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
        const float inter = w * h;
        const float barea = (bx2 - bx1) * (by2 - by1);
        const float ovr = inter / ((iarea + barea) - inter);
        if (ovr > thr) bits |= 1ull << b;
    }
    mask[(long)i * nw + cb] = bits;
}
__global__ __launch_bounds__(256) void nms_mask_kernel(const float* __restrict__ boxes, int n, float thr,
                                                        unsigned long long* __restrict__ mask, int nw) {
    if ((int)blockIdx.x * 4 + 3 < (int)blockIdx.y) return;      // the whole workgroup is below the diagonal
    nms_mask_body(boxes, n, thr, mask, nw, blockIdx.y, blockIdx.x * 4);
}
// Batched form: blockIdx.z = problem; problem p owns boxes [off[p], off[p + 1]) and the mask words at ws + ws_off[p]
__global__ __launch_bounds__(256) void nms_mask_batched_kernel(const float* __restrict__ boxes, const int* __restrict__ off, float thr,
                                                                unsigned long long* __restrict__ ws, const long* __restrict__ ws_off) {
    const int p = blockIdx.z;
    const int o = off[p], n = off[p + 1] - o;
    const int nw = (n + 63) / 64;
    if ((int)blockIdx.x * 4 >= nw || (int)blockIdx.y >= nw || (int)blockIdx.x * 4 + 3 < (int)blockIdx.y) return;
    nms_mask_body(boxes + (long)o * 4, n, thr, ws + ws_off[p], nw, blockIdx.y, blockIdx.x * 4);
}
// pass 2: one workgroup of sixteen waves walks the sorted boxes 64 at a time, in a two-stage pipeline (round 6; the serial form of
// rounds 1-5 - one lane group looping over the kept rows with dependent loads, everyone else waiting - was 3.2 of the 13 ms of a
// YOLOv4 608^2 eval pass):
//   wave 0, iteration b:      resolves the dependency chain of block b on its diagonal word (only boxes still alive are visited:
//                             find-first-set + v_readlane of the row, all scalar), appends the kept indices (one lane per kept box,
//                             prefix popcount), publishes the block's kept list, and ORs the kept rows' word b + 1 into `removed`
//                             itself (one row per lane, butterfly OR) - the one word it needs at iteration b + 1;
//   waves 1-15, iteration b:  OR the rows kept in block b - 1 into the words >= b + 1 of `removed` (four row groups x 240 word slots,
//                             up to four independent 8-byte loads per thread in flight, ds_or_b64 into LDS).
// One barrier per iteration; every decision is a boolean function of the mask, so the result is the serial scan's bit for bit.
__device__ __forceinline__ void nms_scan_body(const unsigned long long* __restrict__ mask, int n, int nw, int* __restrict__ keep,
                                              int* __restrict__ nkeep) {
    extern __shared__ unsigned long long removed[];  // nw words
    __shared__ unsigned char klist[2][64];
    __shared__ int kcnt[2];
    const int tid = threadIdx.x;
    for (int w = tid; w < nw; w += 1024) removed[w] = 0ull;
    if (tid < 2) kcnt[tid] = 0;
    __syncthreads();
    int count = 0;                                   // wave 0: kept so far (uniform)
    for (int blk = 0; blk < nw; ++blk) {             // (the rows kept in the last block have no later words to mark)
        if (tid < 64) {
            if (blk < nw) {
                const int lane = tid;
                const int base = blk * 64;
                const int cnt = min(64, n - base);
                const unsigned long long diag = (lane < cnt) ? mask[(long)(base + lane) * nw + blk] : 0ull;
                const unsigned lo = (unsigned)diag, hi = (unsigned)(diag >> 32);
                const unsigned long long valid = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
                unsigned long long cand = ~removed[blk] & valid, kept = 0ull;
                while (cand != 0ull) {
                    const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)cand) - 1);
                    kept |= 1ull << b;
                    const unsigned long long row = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, b) << 32) |
                                                   (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)lo, b);
                    cand &= ~(row | (1ull << b));          // row b holds only boxes behind b
                }
                const int kc = __popcll(kept);
                if ((kept >> lane) & 1ull) {
                    const int pos = __popcll(kept & ((1ull << lane) - 1ull));
                    keep[count + pos] = base + lane;
                    klist[blk & 1][pos] = (unsigned char)lane;
                }
                if (lane == 0) kcnt[blk & 1] = kc;
                count += kc;
                // the word the next iteration starts from: rows kept in THIS block, word blk + 1
                if (blk + 1 < nw) {
                    unsigned long long v = ((kept >> lane) & 1ull) ? mask[(long)(base + lane) * nw + blk + 1] : 0ull;
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) v |= __shfl_xor(v, o);
                    if (lane == 0 && v != 0ull) atomicOr(&removed[blk + 1], v);
                }
            }
        } else if (blk >= 1) {
            const int pb = blk - 1;                  // rows kept in block pb -> words >= pb + 2 = blk + 1
            const int kc = kcnt[pb & 1];
            const int t = tid - 64, slot = t % 240, rg = t / 240;       // 960 threads = 4 row groups x 240 word slots
            const long rbase = (long)pb * 64;
            const unsigned char* kl = klist[pb & 1];
            for (int w = blk + 1 + slot; w < nw && kc > 0; w += 240) {
                unsigned long long acc = 0ull;
                int j = rg;
                for (; j + 12 < kc; j += 16) {
                    const unsigned long long a0 = mask[(rbase + kl[j]) * nw + w], a1 = mask[(rbase + kl[j + 4]) * nw + w],
                                             a2 = mask[(rbase + kl[j + 8]) * nw + w], a3 = mask[(rbase + kl[j + 12]) * nw + w];
                    acc |= (a0 | a1) | (a2 | a3);
                }
                for (; j < kc; j += 4) acc |= mask[(rbase + kl[j]) * nw + w];
                if (acc != 0ull) atomicOr(&removed[w], acc);
            }
        }
        __syncthreads();
    }
    if (tid == 0) nkeep[0] = count;
}
__global__ __launch_bounds__(1024) void nms_scan_kernel(const unsigned long long* __restrict__ mask, int n, int nw,
                                                         int* __restrict__ keep, int* __restrict__ nkeep) {
    nms_scan_body(mask, n, nw, keep, nkeep);
}
// Batched form: one workgroup per problem (they run side by side on different CUs); kept indices are local to the problem and
// land at keep + off[p]
__global__ __launch_bounds__(1024) void nms_scan_batched_kernel(const unsigned long long* __restrict__ ws, const long* __restrict__ ws_off,
                                                                 const int* __restrict__ off, int* __restrict__ keep,
                                                                 int* __restrict__ nkeep) {
    const int p = blockIdx.x;
    const int o = off[p], n = off[p + 1] - o;
    if (n <= 0) {
        if (threadIdx.x == 0) nkeep[p] = 0;
        return;
    }
    nms_scan_body(ws + ws_off[p], n, (n + 63) / 64, keep + o, nkeep + p);
}

// ---------------------------------------------------------------- focal loss (functional.py:59-113)
__global__ void focal_fwd_kernel(const float* __restrict__ x, const long* __restrict__ target, const float* __restrict__ weight,
                                 float* __restrict__ loss_el, uint8_t* __restrict__ valid, int N, int K, long S, int ignore_index,
                                 float gamma) {
    const long total = (long)N * S;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long n = t / S, s = t % S;
        const float* px = x + n * K * S + s;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, px[(long)k * S]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(px[(long)k * S] - mx);
        const long tg = target[t];
        float logpt = (px[tg * S] - mx) - logf(se);
        const float pt = expf(logpt);
        if (weight != nullptr) logpt = weight[tg] * logpt;
        loss_el[t] = -1.f * powf(1.f - pt, gamma) * logpt;
        valid[t] = !(ignore_index >= 0 && ignore_index < K && tg == ignore_index);
    }
}
__global__ void focal_bwd_kernel(const float* __restrict__ x, const long* __restrict__ target, const float* __restrict__ weight,
                                 const float* __restrict__ dloss, float* __restrict__ dx, int N, int K, long S, float gamma) {
    const long total = (long)N * S;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long n = t / S, s = t % S;
        const float* px = x + n * K * S + s;
        float* pdx = dx + n * K * S + s;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, px[(long)k * S]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(px[(long)k * S] - mx);
        const long tg = target[t];
        const float lse = logf(se);
        const float logpt = (px[tg * S] - mx) - lse;
        const float pt = expf(logpt);
        const float w = weight != nullptr ? weight[tg] : 1.f;
        const float om = 1.f - pt;
        // loss = -(1-pt)^g * w * logpt ; d/dlogpt = -w [ (1-pt)^g - g (1-pt)^(g-1) pt logpt ]
        float dl;
        if (gamma == 0.f) dl = -w;
        else dl = -w * (powf(om, gamma) - gamma * powf(om, gamma - 1.f) * pt * logpt);
        const float gup = dloss[t] * dl;
        for (int k = 0; k < K; ++k) {
            const float pk = expf((px[(long)k * S] - mx) - lse);
            pdx[(long)k * S] = gup * ((k == tg ? 1.f : 0.f) - pk);
        }
    }
}

// ---------------------------------------------------------------- label-smoothed cross entropy
__global__ void ce_fwd_bwd_kernel(const float* __restrict__ logits, const long* __restrict__ target, float* __restrict__ loss_el,
                                  float* __restrict__ dlogits, int N, int K, float ls) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* px = logits + (long)n * K;
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, px[k]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(px[k] - mx);
    const float lse = logf(se) + mx;
    const long tg = target[n];
    float sum_logp = 0.f;
    for (int k = 0; k < K; ++k) sum_logp += px[k] - lse;
    const float nll = -(((tg >= 0 && tg < K) ? px[tg] : NAN) - lse);   // class index out of range: NaN loss, no out-of-bounds read
    const float smooth = -sum_logp / (float)K;
    loss_el[n] = (1.f - ls) * nll + ls * smooth;
    if (dlogits != nullptr) {
        const float invN = 1.f / (float)N;
        for (int k = 0; k < K; ++k) {
            const float p = expf(px[k] - lse);
            dlogits[(long)n * K + k] = (p - (1.f - ls) * (k == tg ? 1.f : 0.f) - ls / (float)K) * invN;
        }
    }
}


// Mean label-smoothed cross entropy as the training loop calls it (nn.CrossEntropyLoss(label_smoothing=ls), mean reduction,
// references/classification/train.py:194): forward = ONE single-workgroup launch that writes the scalar loss and the number of valid
// rows (target != ignore_index); the partial sums are combined in a fixed order, so the loss is bit-reproducible.  torch's own
// composition is ~25 launches (log_softmax, nll_loss, sum, neg, mul, div, masked_fill, ...).  aux = {valid rows}
__global__ __launch_bounds__(1024) void ce_mean_fwd_kernel(const float* __restrict__ logits, const long* __restrict__ target,
                                                           float* __restrict__ loss, float* __restrict__ aux, int N, int K, float ls,
                                                           long ignore_index) {
    __shared__ float s_nll[1024], s_sm[1024], s_cnt[1024];
    float nll = 0.f, sm = 0.f, cnt = 0.f;
    for (int n = threadIdx.x; n < N; n += 1024) {
        const long tg = target[n];
        if (tg == ignore_index) continue;
        const float* px = logits + (long)n * K;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, px[k]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(px[k] - mx);
        const float lse = logf(se) + mx;
        float sum_logp = 0.f;
        for (int k = 0; k < K; ++k) sum_logp += px[k] - lse;
        nll += -(((tg >= 0 && tg < K) ? px[tg] : NAN) - lse);   // class index out of range (torch asserts): NaN loss, no out-of-bounds read
        sm += -sum_logp;
        cnt += 1.f;
    }
    s_nll[threadIdx.x] = nll; s_sm[threadIdx.x] = sm; s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s_nll[threadIdx.x] += s_nll[threadIdx.x + w];
            s_sm[threadIdx.x] += s_sm[threadIdx.x + w];
            s_cnt[threadIdx.x] += s_cnt[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float c = s_cnt[0];
        // torch: (1 - ls) * mean(nll) + ls * mean(sum_k -logp_k) / K ; no valid row -> nan (0 / 0), like torch
        loss[0] = (1.f - ls) * (s_nll[0] / c) + ls * ((s_sm[0] / c) / (float)K);
        aux[0] = c;
    }
}
// ---- the same criterion for wide class counts (K > 64: 1000-class heads).  The thread-per-row kernels above walk a row serially
// with a 4 K-byte stride between lanes: 398 us forward + 324 us backward at 256 x 1000.  Here a WAVE owns a row (coalesced 16-byte
// loads, two passes: max + sum of logits, then the exponential sum), writes the row's log-sum-exp and its two partial sums to aux,
// and a single-workgroup kernel adds the rows in a fixed order.  aux = {valid rows, lse[N], {nll_n, smooth_n}[N]}.
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(256) void ce_rows_fwd_kernel(const float* __restrict__ logits, const long* __restrict__ target,
                                                          float* __restrict__ aux, int N, int K, long ignore_index) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float* lse_out = aux + 1;
    float* part = aux + 1 + N;
    const long tg = target[n];
    if (tg == ignore_index) {
        if (lane == 0) { lse_out[n] = 0.f; part[2 * n] = 0.f; part[2 * n + 1] = 0.f; }
        return;
    }
    const float* px = logits + (long)n * K;
    float mx = -INFINITY, sx = 0.f, se = 0.f;
    if ((K & 3) == 0 && (reinterpret_cast<unsigned long long>(logits) & 15ull) == 0) {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(px);
        for (int i = lane; i < K / 4; i += 64) {
            const f32x4 v = p4[i];
            mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
            sx += (v[0] + v[1]) + (v[2] + v[3]);
        }
        mx = wave_max(mx);
        for (int i = lane; i < K / 4; i += 64) {
            const f32x4 v = p4[i];
            se += (expf(v[0] - mx) + expf(v[1] - mx)) + (expf(v[2] - mx) + expf(v[3] - mx));
        }
    } else {
        for (int k = lane; k < K; k += 64) { const float v = px[k]; mx = fmaxf(mx, v); sx += v; }
        mx = wave_max(mx);
        for (int k = lane; k < K; k += 64) se += expf(px[k] - mx);
    }
    sx = wave_sum(sx);
    se = wave_sum(se);
    if (lane == 0) {
        const float lse = logf(se) + mx;
        lse_out[n] = lse;
        part[2 * n] = -(((tg >= 0 && tg < K) ? px[tg] : NAN) - lse);   // out-of-range class index: NaN loss, no out-of-bounds read
        part[2 * n + 1] = -(sx - (float)K * lse);          // - sum_k log p_k
    }
}
__global__ __launch_bounds__(1024) void ce_rows_finish_kernel(const long* __restrict__ target, float* __restrict__ loss,
                                                              float* __restrict__ aux, int N, int K, float ls, long ignore_index) {
    __shared__ float s_nll[1024], s_sm[1024], s_cnt[1024];
    const float* part = aux + 1 + N;
    float nll = 0.f, sm = 0.f, cnt = 0.f;
    for (int n = threadIdx.x; n < N; n += 1024) {
        if (target[n] == ignore_index) continue;
        nll += part[2 * n];
        sm += part[2 * n + 1];
        cnt += 1.f;
    }
    s_nll[threadIdx.x] = nll; s_sm[threadIdx.x] = sm; s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s_nll[threadIdx.x] += s_nll[threadIdx.x + w];
            s_sm[threadIdx.x] += s_sm[threadIdx.x + w];
            s_cnt[threadIdx.x] += s_cnt[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float c = s_cnt[0];
        loss[0] = (1.f - ls) * (s_nll[0] / c) + ls * ((s_sm[0] / c) / (float)K);
        aux[0] = c;
    }
}
__global__ __launch_bounds__(256) void ce_rows_bwd_kernel(const float* __restrict__ logits, const long* __restrict__ target,
                                                          const float* __restrict__ dloss, const float* __restrict__ aux,
                                                          float* __restrict__ dlogits, int N, int K, float ls, long ignore_index) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* px = logits + (long)n * K;
    float* pd = dlogits + (long)n * K;
    const long tg = target[n];
    const bool dead = tg == ignore_index;
    const float lse = aux[1 + n], gs = dead ? 0.f : dloss[0] / aux[0], on = (1.f - ls) * gs, sm = ls / (float)K * gs;
    if ((K & 3) == 0 && ((reinterpret_cast<unsigned long long>(logits) | reinterpret_cast<unsigned long long>(dlogits)) & 15ull) == 0) {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(px);
        f32x4* d4 = reinterpret_cast<f32x4*>(pd);
        for (int i = lane; i < K / 4; i += 64) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if (!dead) {
                const f32x4 v = p4[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = expf(v[e] - lse) * gs - ((long)(4 * i + e) == tg ? on : 0.f) - sm;
            }
            d4[i] = o;
        }
    } else {
        for (int k = lane; k < K; k += 64) pd[k] = dead ? 0.f : expf(px[k] - lse) * gs - ((long)k == tg ? on : 0.f) - sm;
    }
}

// dlogits[n][k] = dloss * (softmax_k - (1 - ls) [k == target] - ls / K) / valid rows ; rows with the ignored target get zeros
__global__ void ce_mean_bwd_kernel(const float* __restrict__ logits, const long* __restrict__ target, const float* __restrict__ dloss,
                                   const float* __restrict__ aux, float* __restrict__ dlogits, int N, int K, float ls, long ignore_index) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* px = logits + (long)n * K;
    float* pd = dlogits + (long)n * K;
    const long tg = target[n];
    if (tg == ignore_index) {
        for (int k = 0; k < K; ++k) pd[k] = 0.f;
        return;
    }
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, px[k]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(px[k] - mx);
    const float lse = logf(se) + mx;
    const float gs = dloss[0] / aux[0];
    for (int k = 0; k < K; ++k) {
        const float p = expf(px[k] - lse);
        pd[k] = (p - (1.f - ls) * (k == tg ? 1.f : 0.f) - ls / (float)K) * gs;
    }
}

// ---------------------------------------------------------------- input side of the training step
// Mixup (holocron/utils/data/collate.py:39-64): out[i] = lam x[i] + (1 - lam) x[perm[i]] over rows of D elements
template <typename T>
__global__ void mixup_kernel(const T* __restrict__ x, const long* __restrict__ perm, T* __restrict__ out, long D, float lam);
template <>
__global__ void mixup_kernel<float>(const float* __restrict__ x, const long* __restrict__ perm, float* __restrict__ out, long D, float lam) {
    const long i = blockIdx.y, j = perm[i];
    const float* a = x + i * D;
    const float* b = x + j * D;
    float* o = out + i * D;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < D; k += (long)gridDim.x * blockDim.x) o[k] = lam * a[k] + (1.f - lam) * b[k];
}
template <>
__global__ void mixup_kernel<bf16_t>(const bf16_t* __restrict__ x, const long* __restrict__ perm, bf16_t* __restrict__ out, long D, float lam) {
    const long i = blockIdx.y, j = perm[i];
    const bf16_t* a = x + i * D;
    const bf16_t* b = x + j * D;
    bf16_t* o = out + i * D;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < D; k += (long)gridDim.x * blockDim.x)
        o[k] = f32_to_bf16(lam * bf16_to_f32(a[k]) + (1.f - lam) * bf16_to_f32(b[k]));
}
// class indices -> mixed one-hot rows: out[i][c] = lam [t[i] == c] + (1 - lam) [t[perm[i]] == c]
__global__ void mixup_onehot_kernel(const long* __restrict__ t, const long* __restrict__ perm, float* __restrict__ out, long N, int C, float lam) {
    const long total = N * C;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const long i = q / C;
        const int c = (int)(q - i * C);
        out[q] = (t[i] == c ? lam : 0.f) + (t[perm[i]] == c ? 1.f - lam : 0.f);
    }
}
// top-1 / top-k hits of a batch accumulated on the device (trainer/classification.py:60-66 without the per-batch .item()):
// one wave per row; rank of the target = number of classes that beat it (ties broken towards the lower index)
__global__ __launch_bounds__(256) void topk_hits_kernel(const float* __restrict__ logits, const long* __restrict__ target, long N, int C, int k,
                                                        float* __restrict__ counters) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* l = logits + row * C;
    const long t = target[row];
    float beat = 0.f;
    if (t >= 0 && t < C) {
        const float tv = l[t];
        for (int c = lane; c < C; c += 64) {
            const float v = l[c];
            if (v > tv || (v == tv && c < t)) beat += 1.f;
        }
    } else {
        beat = lane == 0 ? (float)C : 0.f;
    }
    beat = wave_sum(beat);
    if (lane == 0) {
        if (beat < 1.f) atomicAdd(counters + 0, 1.f);
        if (k > 1 && beat < (float)k) atomicAdd(counters + 1, 1.f);
        atomicAdd(counters + 2, 1.f);
    }
}

}  // namespace

extern "C" {

int hc_hard_mish_fwd(const float* x, float* y, int64_t n, hc_stream_t stream) {
    if (x == nullptr || y == nullptr || n < 0) return HC_ERR_ARG;
    if (n == 0) return HC_OK;
    hipLaunchKernelGGL(hard_mish_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n);
    return hc_launch_status();
}
int hc_hard_mish_bwd(const float* x, const float* dy, float* dx, int64_t n, hc_stream_t stream) {
    if (x == nullptr || dy == nullptr || dx == nullptr || n < 0) return HC_ERR_ARG;
    if (n == 0) return HC_OK;
    hipLaunchKernelGGL(hard_mish_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, (long)n);
    return hc_launch_status();
}

int hc_box_pairwise(const float* b1, const float* b2, float* out, int32_t M, int32_t N, int32_t kind, hc_stream_t stream) {
    if (M < 0 || N < 0 || kind < 0 || kind > 5) return HC_ERR_ARG;
    if (M == 0 || N == 0) return HC_OK;
    if (b1 == nullptr || b2 == nullptr || out == nullptr) return HC_ERR_ARG;
    hipLaunchKernelGGL(box_pairwise_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, (hipStream_t)stream, b1, b2, out, M, N, kind);
    return hc_launch_status();
}
int hc_box_pairwise_bwd(const float* b1, const float* b2, const float* g, float* db1, float* db2, int32_t M, int32_t N, int32_t kind,
                        hc_stream_t stream) {
    if (M < 0 || N < 0 || kind < 0 || kind > 5) return HC_ERR_ARG;
    if (M == 0 || N == 0) return HC_OK;
    hipLaunchKernelGGL(box_pairwise_bwd_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, (hipStream_t)stream, b1, b2, g, db1, db2, M, N,
                       kind);
    return hc_launch_status();
}

int64_t hc_nms_ws_bytes(int32_t n) {
    if (n <= 0) return 0;
    const int64_t nw = (n + 63) / 64;
    return (int64_t)n * nw * 8;
}
int hc_nms_sorted(const float* boxes, int32_t n, float iou_thr, void* ws, int32_t* keep, int32_t* nkeep, hc_stream_t stream) {
    if (n < 0 || nkeep == nullptr) return HC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (hc_zero_async(nkeep, sizeof(int32_t), st) != hipSuccess) return HC_ERR_LAUNCH;
        return HC_OK;
    }
    if (boxes == nullptr || ws == nullptr || keep == nullptr) return HC_ERR_ARG;
    const int nw = (n + 63) / 64;
    if ((size_t)nw * 8 > 60000) return HC_ERR_ARG;  // bitmap must fit LDS (n <= 480000)
    hipLaunchKernelGGL(nms_mask_kernel, dim3((nw + 3) / 4, nw), dim3(256), 0, st, boxes, n, iou_thr, (unsigned long long*)ws, nw);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(1024), nw * 8, st, (const unsigned long long*)ws, n, nw, keep, nkeep);
    return hc_launch_status();
}

int hc_nms_sorted_batched(const float* boxes, const int32_t* off, int32_t nprob, int32_t nmax, float iou_thr, void* ws,
                          const int64_t* ws_off, int32_t* keep, int32_t* nkeep, hc_stream_t stream) {
    if (nprob < 0 || nmax < 0 || nkeep == nullptr) return HC_ERR_ARG;
    if (nprob == 0) return HC_OK;
    if (off == nullptr) return HC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (nmax == 0) return hc_zero_async(nkeep, sizeof(int32_t) * (size_t)nprob, st) == hipSuccess ? HC_OK : HC_ERR_LAUNCH;
    if (boxes == nullptr || ws == nullptr || ws_off == nullptr || keep == nullptr || nprob > 65535) return HC_ERR_ARG;
    const int nw = (nmax + 63) / 64;
    if ((size_t)nw * 8 > 60000 || nw > 65535) return HC_ERR_ARG;
    hipLaunchKernelGGL(nms_mask_batched_kernel, dim3((nw + 3) / 4, nw, nprob), dim3(256), 0, st, boxes, off, iou_thr, (unsigned long long*)ws,
                       (const long*)ws_off);
    hipLaunchKernelGGL(nms_scan_batched_kernel, dim3(nprob), dim3(1024), nw * 8, st, (const unsigned long long*)ws, (const long*)ws_off,
                       off, keep, nkeep);
    return hc_launch_status();
}

int hc_focal_loss_fwd(const float* x, const int64_t* target, const float* weight, float* loss_el, uint8_t* valid, int32_t N,
                      int32_t K, int64_t S, int32_t ignore_index, float gamma, hc_stream_t stream) {
    if (x == nullptr || target == nullptr || loss_el == nullptr || valid == nullptr || K <= 0) return HC_ERR_ARG;
    if ((long)N * S == 0) return HC_OK;
    hipLaunchKernelGGL(focal_fwd_kernel, dim3(grid_for((long)N * S)), dim3(256), 0, (hipStream_t)stream, x, (const long*)target,
                       weight, loss_el, valid, N, K, (long)S, ignore_index, gamma);
    return hc_launch_status();
}
int hc_focal_loss_bwd(const float* x, const int64_t* target, const float* weight, const float* dloss_el, float* dx, int32_t N,
                      int32_t K, int64_t S, float gamma, hc_stream_t stream) {
    if (x == nullptr || target == nullptr || dloss_el == nullptr || dx == nullptr || K <= 0) return HC_ERR_ARG;
    if ((long)N * S == 0) return HC_OK;
    hipLaunchKernelGGL(focal_bwd_kernel, dim3(grid_for((long)N * S)), dim3(256), 0, (hipStream_t)stream, x, (const long*)target,
                       weight, dloss_el, dx, N, K, (long)S, gamma);
    return hc_launch_status();
}

int hc_ce_mean_fwd(const float* logits, const int64_t* target, float* loss, float* aux, int32_t N, int32_t K, float label_smoothing,
                   int64_t ignore_index, hc_stream_t stream) {
    if (logits == nullptr || target == nullptr || loss == nullptr || aux == nullptr || N <= 0 || K <= 0) return HC_ERR_ARG;
    if (K > 64) {
        hipLaunchKernelGGL(ce_rows_fwd_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, (const long*)target, aux, N, K,
                           (long)ignore_index);
        hipLaunchKernelGGL(ce_rows_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const long*)target, loss, aux, N, K,
                           label_smoothing, (long)ignore_index);
        return hc_launch_status();
    }
    hipLaunchKernelGGL(ce_mean_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, (const long*)target, loss, aux, N, K,
                       label_smoothing, (long)ignore_index);
    return hc_launch_status();
}
int64_t hc_ce_mean_aux_floats(int32_t N) { return 1 + 3 * (int64_t)(N > 0 ? N : 0); }
int hc_ce_mean_bwd(const float* logits, const int64_t* target, const float* dloss, const float* aux, float* dlogits, int32_t N,
                   int32_t K, float label_smoothing, int64_t ignore_index, hc_stream_t stream) {
    if (logits == nullptr || target == nullptr || dloss == nullptr || aux == nullptr || dlogits == nullptr || N <= 0 || K <= 0)
        return HC_ERR_ARG;
    if (K > 64) {
        hipLaunchKernelGGL(ce_rows_bwd_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, (const long*)target, dloss, aux,
                           dlogits, N, K, label_smoothing, (long)ignore_index);
        return hc_launch_status();
    }
    hipLaunchKernelGGL(ce_mean_bwd_kernel, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, logits, (const long*)target, dloss,
                       aux, dlogits, N, K, label_smoothing, (long)ignore_index);
    return hc_launch_status();
}

int hc_ce_fwd_bwd(const float* logits, const int64_t* target, float* loss_el, float* dlogits, int32_t N, int32_t K,
                  float label_smoothing, hc_stream_t stream) {
    if (logits == nullptr || target == nullptr || loss_el == nullptr || N <= 0 || K <= 0) return HC_ERR_ARG;
    hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, logits, (const long*)target,
                       loss_el, dlogits, N, K, label_smoothing);
    return hc_launch_status();
}

int hc_mixup(const void* x, const int64_t* perm, void* out, int64_t N, int64_t D, int32_t dtype, float lam, hc_stream_t stream) {
    if (x == nullptr || perm == nullptr || out == nullptr || N < 0 || D < 0 || N > 65535 || (dtype != 0 && dtype != 1)) return HC_ERR_ARG;
    if (N == 0 || D == 0) return HC_OK;
    long bx = (D + 1023) / 1024;
    if (bx > 1024) bx = 1024;
    if (dtype == 0)
        hipLaunchKernelGGL(mixup_kernel<float>, dim3((unsigned)bx, (unsigned)N), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const long*)perm,
                           (float*)out, (long)D, lam);
    else
        hipLaunchKernelGGL(mixup_kernel<bf16_t>, dim3((unsigned)bx, (unsigned)N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (const long*)perm, (bf16_t*)out, (long)D, lam);
    return hc_launch_status();
}
int hc_mixup_onehot(const int64_t* target, const int64_t* perm, float* out, int64_t N, int32_t C, float lam, hc_stream_t stream) {
    if (target == nullptr || perm == nullptr || out == nullptr || N < 0 || C <= 0) return HC_ERR_ARG;
    if (N == 0) return HC_OK;
    hipLaunchKernelGGL(mixup_onehot_kernel, dim3(grid_for(N * C)), dim3(256), 0, (hipStream_t)stream, (const long*)target, (const long*)perm, out,
                       (long)N, C, lam);
    return hc_launch_status();
}
int hc_topk_hits(const float* logits, const int64_t* target, int64_t N, int32_t C, int32_t k, float* counters, hc_stream_t stream) {
    if (logits == nullptr || target == nullptr || counters == nullptr || N < 0 || C <= 0 || k < 1) return HC_ERR_ARG;
    if (N == 0) return HC_OK;
    hipLaunchKernelGGL(topk_hits_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, (const long*)target, (long)N, C, k,
                       counters);
    return hc_launch_status();
}

}  // extern "C"
