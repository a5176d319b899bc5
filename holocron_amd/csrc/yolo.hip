// YOLOv4 detection layer: box decoding, target assignment, the four-part loss with its gradient, and the
// candidate filter in front of NMS (reference: holocron/models/detection/yolov4.py:269-420).
//
// One thread owns one (image, cell, anchor) predictor = 5 + num_classes logits.  The logits are read in place
// from whatever layout the head conv left them in, described by element strides (sn, sc, sp) for image, channel
// and pixel: NHWC bf16 with a padded channel count (our conv path) or NCHW fp32 (a plain tensor handed to the
// layer).  Channel index = anchor * (5 + nc) + k, as the reference's reshape(b, A, 5 + nc, h, w) implies.
// Compiled with -ffp-contract=off: the decode follows the reference's operation order.
#include "common.h"
#include "../../include/holocron_hip.h"

namespace {

struct Logits {
    const void* p;
    int bf16;
    long sn, sc, sp;
    __device__ __forceinline__ float get(long n, long pix, int ch) const {
        const long off = n * sn + pix * sp + (long)ch * sc;
        return bf16 ? bf16_to_f32(((const bf16_t*)p)[off]) : ((const float*)p)[off];
    }
};
struct Grads {
    void* p;
    int bf16;
    long sn, sc, sp;
    __device__ __forceinline__ void put(long n, long pix, int ch, float v) const {
        const long off = n * sn + pix * sp + (long)ch * sc;
        if (bf16) ((bf16_t*)p)[off] = f32_to_bf16(v);
        else ((float*)p)[off] = v;
    }
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

struct Box { float x1, y1, x2, y2; };

// yolov4.py:279-293.  raw_wh = exp(t_wh) * anchor before the clamp (needed by the gradient)
__device__ __forceinline__ Box decode_box(float tx, float ty, float tw, float th, int cx, int cy, int W, int H, float aw, float ah,
                                          float scale_xy, float* sx, float* sy, float* rw, float* rh) {
    const float sgx = sigmoidf_(tx), sgy = sigmoidf_(ty);
    float bx = scale_xy * sgx - 0.5f * (scale_xy - 1.f);
    float by = scale_xy * sgy - 0.5f * (scale_xy - 1.f);
    bx = (bx + (float)cx) / (float)W;
    by = (by + (float)cy) / (float)H;
    const float w0 = expf(tw) * aw, h0 = expf(th) * ah;
    const float bw = fminf(fmaxf(w0, 0.f), 2.f), bh = fminf(fmaxf(h0, 0.f), 2.f);
    Box b;
    b.x1 = bx - 0.5f * bw;
    b.y1 = by - 0.5f * bh;
    b.x2 = b.x1 + bw;
    b.y2 = b.y1 + bh;
    if (sx) { *sx = sgx; *sy = sgy; *rw = w0; *rh = h0; }
    return b;
}

__device__ __forceinline__ float iou_of(const Box a, const Box b) {
    const float area1 = (a.x2 - a.x1) * (a.y2 - a.y1), area2 = (b.x2 - b.x1) * (b.y2 - b.y1);
    const float w = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), 0.f);
    const float h = fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1), 0.f);
    const float inter = w * h;
    return inter / ((area1 + area2) - inter);
}
__device__ __forceinline__ float penalty_of(const Box a, const Box b) {   // ops/boxes.py:69-103
    const float ex = fmaxf(a.x2, b.x2) - fminf(a.x1, b.x1), ey = fmaxf(a.y2, b.y2) - fminf(a.y1, b.y1);
    const float c2 = ex * ex + ey * ey;
    const float sx = (a.x1 + a.x2) - (b.x1 + b.x2), sy = (a.y1 + a.y2) - (b.y1 + b.y2);
    return ((sx * sx + sy * sy) / 4.f) / c2;
}
// torch.max / torch.min split the gradient evenly on ties
__device__ __forceinline__ float gt_w(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }

// d iou / d a (a = predicted box), accumulated as g * d
__device__ __forceinline__ void iou_grad(const Box a, const Box b, float g, float (&d)[4]) {
    const float aw = a.x2 - a.x1, ah = a.y2 - a.y1;
    const float area1 = aw * ah, area2 = (b.x2 - b.x1) * (b.y2 - b.y1);
    const float w0 = fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), h0 = fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1);
    const float w = fmaxf(w0, 0.f), h = fmaxf(h0, 0.f);
    const float inter = w * h;
    const float uni = (area1 + area2) - inter;
    // iou = inter / uni ; d = (dinter * uni - inter * duni) / uni^2 ; duni = darea1 - dinter
    const float gi = g * (1.f / uni + inter / (uni * uni));   // coefficient of d inter
    const float ga = -g * inter / (uni * uni);                // coefficient of d area1
    const float mw = w0 >= 0.f ? 1.f : 0.f, mh = h0 >= 0.f ? 1.f : 0.f;   // clamp(min=0) passes the gradient at 0
    const float dw = gi * h * mw, dh = gi * w * mh;           // d / d w0, d / d h0
    d[0] += -dw * gt_w(a.x1, b.x1) - ga * ah;
    d[1] += -dh * gt_w(a.y1, b.y1) - ga * aw;
    d[2] += dw * gt_w(b.x2, a.x2) + ga * ah;
    d[3] += dh * gt_w(b.y2, a.y2) + ga * aw;
}
__device__ __forceinline__ void penalty_grad(const Box a, const Box b, float g, float (&d)[4]) {
    const float ex = fmaxf(a.x2, b.x2) - fminf(a.x1, b.x1), ey = fmaxf(a.y2, b.y2) - fminf(a.y1, b.y1);
    const float c2 = ex * ex + ey * ey;
    const float sx = (a.x1 + a.x2) - (b.x1 + b.x2), sy = (a.y1 + a.y2) - (b.y1 + b.y2);
    const float cd2 = (sx * sx + sy * sy) / 4.f;
    const float gn = g / c2;                 // coefficient of d cd2
    const float gc = -g * cd2 / (c2 * c2);   // coefficient of d c2
    const float gex = gc * 2.f * ex, gey = gc * 2.f * ey;
    d[0] += gn * sx * 0.5f - gex * gt_w(b.x1, a.x1);
    d[1] += gn * sy * 0.5f - gey * gt_w(b.y1, a.y1);
    d[2] += gn * sx * 0.5f + gex * gt_w(a.x2, b.x2);
    d[3] += gn * sy * 0.5f + gey * gt_w(a.y2, b.y2);
}

// ---------------------------------------------------------------- decode (+ eval-time scores)
__global__ void yolo_decode_kernel(Logits x, int N, int H, int W, int A, int nc, const float* __restrict__ anchors, float scale_xy,
                                   float* __restrict__ boxes, float* __restrict__ obj, float* __restrict__ score,
                                   long* __restrict__ label, int clamp01) {
    const long total = (long)N * H * W * A;
    const int D = 5 + nc;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int a = (int)(t % A);
        const long pixg = t / A;
        const long pix = pixg % ((long)H * W);
        const long n = pixg / ((long)H * W);
        const int cx = (int)(pix % W), cy = (int)(pix / W);
        const int c0 = a * D;
        Box b = decode_box(x.get(n, pix, c0), x.get(n, pix, c0 + 1), x.get(n, pix, c0 + 2), x.get(n, pix, c0 + 3), cx, cy, W, H,
                           anchors[2 * a], anchors[2 * a + 1], scale_xy, nullptr, nullptr, nullptr, nullptr);
        if (clamp01) {
            b.x1 = fminf(fmaxf(b.x1, 0.f), 1.f); b.y1 = fminf(fmaxf(b.y1, 0.f), 1.f);
            b.x2 = fminf(fmaxf(b.x2, 0.f), 1.f); b.y2 = fminf(fmaxf(b.y2, 0.f), 1.f);
        }
        boxes[4 * t] = b.x1; boxes[4 * t + 1] = b.y1; boxes[4 * t + 2] = b.x2; boxes[4 * t + 3] = b.y2;
        if (obj != nullptr) {
            const float so = sigmoidf_(x.get(n, pix, c0 + 4));
            obj[t] = so;
            if (score != nullptr) {
                // (sigmoid(scores)).max(-1): first maximum on ties
                float best = -1.f;
                int bi = 0;
                for (int k = 0; k < nc; ++k) {
                    const float s = sigmoidf_(x.get(n, pix, c0 + 5 + k));
                    if (s > best) { best = s; bi = k; }
                }
                score[t] = best * so;
                label[t] = bi;
            }
        }
    }
}

// ---------------------------------------------------------------- target assignment (yolov4.py:350-373)
__global__ void yolo_assign_kernel(const float* __restrict__ gt, const int* __restrict__ gt_img, int G, const float* __restrict__ anchors,
                                   int H, int W, int A, uint8_t* __restrict__ obj_mask, uint8_t* __restrict__ cell_gt) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const float x1 = gt[4 * g], y1 = gt[4 * g + 1], x2 = gt[4 * g + 2], y2 = gt[4 * g + 3];
    // centres: mean of (x1, x2) and (y1, y2), scaled by the grid and truncated (.to(long))
    const long cx = (long)(((x1 + x2) / 2.f) * (float)W), cy = (long)(((y1 + y2) / 2.f) * (float)H);
    if (cx < 0 || cx >= W || cy < 0 || cy >= H) return;   // the reference would raise an index error here
    const float gw = x2 - x1, gh = y2 - y1;
    // box_iou(cat(-wh, wh), cat(-anchor, anchor)).argmax(1): boxes centred on the origin
    const float area1 = (gw - (-gw)) * (gh - (-gh));
    float best = -INFINITY;
    int ba = 0;
    for (int a = 0; a < A; ++a) {
        const float aw = anchors[2 * a], ah = anchors[2 * a + 1];
        const float area2 = (aw - (-aw)) * (ah - (-ah));
        const float w = fmaxf(fminf(gw, aw) - fmaxf(-gw, -aw), 0.f), h = fmaxf(fminf(gh, ah) - fmaxf(-gh, -ah), 0.f);
        const float inter = w * h;
        const float iou = inter / ((area1 + area2) - inter);
        if (iou > best) { best = iou; ba = a; }
    }
    const long cell = ((long)gt_img[g] * H + cy) * W + cx;
    obj_mask[cell * A + ba] = 1;
    cell_gt[cell] = 1;
}

// ---------------------------------------------------------------- losses (yolov4.py:375-420)
// sums[0] += (sigmoid(o) - iou_max)^2 over assigned predictors, sums[1] += sigmoid(o)^2 over predictors of cells
// without a ground-truth centre, sums[2] += min_k (1 - iou_k + penalty_k), sums[3] += mean_c BCE(logit_c, onehot)
template <bool BWD>
__global__ __launch_bounds__(256) void yolo_loss_kernel(Logits x, Grads dx, int N, int H, int W, int A, int nc,
                                                        const float* __restrict__ anchors, float scale_xy,
                                                        const float* __restrict__ gt, const long* __restrict__ gt_label,
                                                        const int* __restrict__ gt_off, const uint8_t* __restrict__ obj_mask,
                                                        const uint8_t* __restrict__ cell_gt, float* __restrict__ sums,
                                                        const float* __restrict__ gcoef) {
    const long total = (long)N * H * W * A;
    const int D = 5 + nc;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float gc[4] = {0.f, 0.f, 0.f, 0.f};
    if (BWD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) gc[i] = gcoef[i];
    }
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int a = (int)(t % A);
        const long pixg = t / A;
        const long pix = pixg % ((long)H * W);
        const long n = pixg / ((long)H * W);
        const int c0 = a * D;
        const float so = sigmoidf_(x.get(n, pix, c0 + 4));
        float go = 0.f;   // d / d objectness logit
        if (!cell_gt[pixg]) {
            acc[1] += so * so;
            if (BWD) go += gc[1] * 2.f * so * so * (1.f - so);
        }
        const bool is_obj = obj_mask[t] != 0;
        const int k0 = gt_off[n], k1 = gt_off[n + 1];
        if (is_obj && k1 > k0) {
            const int cx = (int)(pix % W), cy = (int)(pix / W);
            float sx, sy, rw, rh;
            const Box b = decode_box(x.get(n, pix, c0), x.get(n, pix, c0 + 1), x.get(n, pix, c0 + 2), x.get(n, pix, c0 + 3), cx, cy,
                                     W, H, anchors[2 * a], anchors[2 * a + 1], scale_xy, &sx, &sy, &rw, &rh);
            float iou_max = -INFINITY, loss_min = INFINITY;
            int kmax = k0, kmin = k0;
            for (int k = k0; k < k1; ++k) {
                const Box gb = {gt[4 * k], gt[4 * k + 1], gt[4 * k + 2], gt[4 * k + 3]};
                const float iou = iou_of(b, gb);
                const float l = (1.f - iou) + penalty_of(b, gb);
                if (iou > iou_max) { iou_max = iou; kmax = k; }
                if (l < loss_min) { loss_min = l; kmin = k; }
            }
            const float diff = so - iou_max;
            acc[0] += diff * diff;
            acc[2] += loss_min;
            const int lab = (int)gt_label[kmax];
            float bce = 0.f;
            // the class logits sixteen at a time, all loads first: the few assigned predictors of a launch (one lane of a wave here and
            // there) each walked nc DEPENDENT global loads - 80 round trips, which WAS the kernel's duration (66 / 85 us per launch for
            // 0.3 M predictors).  Same additions in the same order.
            for (int cb = 0; cb < nc; cb += 16) {
                float vv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) vv[u] = x.get(n, pix, c0 + 5 + (cb + u < nc ? cb + u : nc - 1));
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int c = cb + u;
                    if (c >= nc) break;
                    const float v = vv[u];
                    const float tgt = c == lab ? 1.f : 0.f;
                    // softplus(-|v|) with the hardware exp / log: the library log1pf(expf()) pair is ~250 instructions, and 80 of them in
                    // the one active lane of a wave WAS the forward launch (70-85 us; the backward, which does not evaluate it, 43 us)
                    const float tq = __expf(-fabsf(v));
                    bce += fmaxf(v, 0.f) - v * tgt + (tq < 1e-4f ? tq : __logf(1.f + tq));
                    if (BWD) dx.put(n, pix, c0 + 5 + c, gc[3] * (sigmoidf_(v) - tgt) / (float)nc);
                }
            }
            acc[3] += bce / (float)nc;
            if (BWD) {
                go += gc[0] * 2.f * diff * so * (1.f - so);
                float d[4] = {0.f, 0.f, 0.f, 0.f};
                // the objectness target is the (differentiable) IoU itself: d/d iou of (so - iou)^2
                const Box gmax = {gt[4 * kmax], gt[4 * kmax + 1], gt[4 * kmax + 2], gt[4 * kmax + 3]};
                iou_grad(b, gmax, gc[0] * (-2.f * diff), d);
                const Box gmin = {gt[4 * kmin], gt[4 * kmin + 1], gt[4 * kmin + 2], gt[4 * kmin + 3]};
                iou_grad(b, gmin, -gc[2], d);
                penalty_grad(b, gmin, gc[2], d);
                // (x1, y1, x2, y2) -> (bx, by, bw, bh) -> logits
                const float gbx = d[0] + d[2], gby = d[1] + d[3];
                const float gbw = 0.5f * (d[2] - d[0]), gbh = 0.5f * (d[3] - d[1]);
                dx.put(n, pix, c0, gbx * scale_xy * sx * (1.f - sx) / (float)W);
                dx.put(n, pix, c0 + 1, gby * scale_xy * sy * (1.f - sy) / (float)H);
                dx.put(n, pix, c0 + 2, (rw >= 0.f && rw <= 2.f) ? gbw * rw : 0.f);
                dx.put(n, pix, c0 + 3, (rh >= 0.f && rh <= 2.f) ? gbh * rh : 0.f);
            }
        }
        // (predictors without an assigned box: their box and class gradients are zero - dlogits ARRIVES zeroed, hc_yolo_loss_bwd's
        // contract; writing 4 + nc scattered 2-byte zeros per predictor here was the kernel: 143 us per launch on the 76 x 76 map)
        if (BWD) dx.put(n, pix, c0 + 4, go);
    }
    if (!BWD) {
        __shared__ float sh[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = wave_sum(acc[i]);
            if ((threadIdx.x & 63) == 0) sh[i][threadIdx.x >> 6] = v;
        }
        __syncthreads();
        if (threadIdx.x < 4) atomicAdd(&sums[threadIdx.x], sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3]);
    }
}


// ================================================================ YOLOv1 / YOLOv2 (holocron/models/detection/yolo.py:48-215)
// Already formatted predictions (fp32, contiguous): boxes [N][H][W][A][4] (xc, yc, w, h), objectness [N][H][W][A], class
// probabilities [N][H][W][As][nc] with As = 1 (YOLOv1: one distribution per cell) or A.  cell_rel != 0: the centre is relative to
// the cell (YOLOv1.to_isoboxes, yolo.py:140-163), else it is absolute (YOLOv2.to_isoboxes, yolov2.py:157-173).
__device__ __forceinline__ Box isobox(const float* __restrict__ b, int cx, int cy, int W, int H, int cell_rel) {
    const float X = cell_rel ? (b[0] + (float)cx) / (float)W : b[0];
    const float Y = cell_rel ? (b[1] + (float)cy) / (float)H : b[1];
    const float hw = b[2] / 2.f, hh = b[3] / 2.f;
    return Box{X - hw, Y - hh, X + hw, Y + hh};
}

// one thread per ground-truth box: the responsible anchor of its cell (highest IoU, first on ties) -> assign[g] = (cell * A + a),
// iou_out[g]; mark[n][cell][a] = 1 (yolo.py:97-103)
__global__ void yolo1_assign_kernel(const float* __restrict__ pb, const float* __restrict__ gt, const int* __restrict__ gt_img, int G, int H,
                                    int W, int A, int cell_rel, int* __restrict__ assign, float* __restrict__ iou_out,
                                    unsigned char* __restrict__ mark) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const Box gb = {gt[4 * g], gt[4 * g + 1], gt[4 * g + 2], gt[4 * g + 3]};
    const int n = gt_img[g];
    int cx = (int)(((gb.x1 + gb.x2) / 2.f) * (float)W), cy = (int)(((gb.y1 + gb.y2) / 2.f) * (float)H);
    cx = cx < 0 ? 0 : (cx >= W ? W - 1 : cx);
    cy = cy < 0 ? 0 : (cy >= H ? H - 1 : cy);
    const long cell = ((long)n * H + cy) * W + cx;
    float best = -1.f;
    int abest = 0;
    for (int a = 0; a < A; ++a) {
        const float v = iou_of(gb, isobox(pb + (cell * A + a) * 4, cx, cy, W, H, cell_rel));
        if (v > best) { best = v; abest = a; }
    }
    assign[g] = (int)(cell * A + abest);
    iou_out[g] = best;
    mark[cell * A + abest] = 1;
}

// ground-truth terms, one thread per box (yolo.py:104-121).  sums: obj, noobj, bbox, clf.  BWD: gradients (scaled by
// gc[0..3]) are accumulated with atomics (several boxes may share a cell / anchor); the buffers are zeroed by the caller.
template <bool BWD>
__global__ void yolo1_gt_kernel(const float* __restrict__ pb, const float* __restrict__ po, const float* __restrict__ ps,
                                const float* __restrict__ gt, const long* __restrict__ gt_label, const int* __restrict__ gt_img,
                                const int* __restrict__ gt_off, int G, int H, int W, int A, int As, int nc, int cell_rel,
                                const int* __restrict__ assign, const float* __restrict__ gc, float* __restrict__ sums,
                                float* __restrict__ dpb, float* __restrict__ dpo, float* __restrict__ dps) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const Box gb = {gt[4 * g], gt[4 * g + 1], gt[4 * g + 2], gt[4 * g + 3]};
    const int n = gt_img[g];
    const int slot = assign[g];                     // cell * A + a
    const long cell = slot / A;
    const int cx = (int)(cell % W), cy = (int)((cell / W) % H);
    const float* b = pb + (long)slot * 4;
    const Box pbx = isobox(b, cx, cy, W, H, cell_rel);
    const float iou = iou_of(gb, pbx);
    const float o = po[slot];
    // classification: (one_hot - p)^2 over the As distributions of the cell
    const long lab = gt_label[g];
    float clf = 0.f;
    for (int sidx = 0; sidx < As; ++sidx)
        for (int c = 0; c < nc; ++c) {
            const long q = (cell * As + sidx) * nc + c;
            const float d = (c == lab ? 1.f : 0.f) - ps[q];
            clf += d * d;
            if (BWD) atomicAdd(dps + q, gc[3] * (-2.f * d));
        }
    // box centre against the prediction's centre; sqrt(w), sqrt(h) of EVERY box of the image against the prediction's
    // (yolo.py:115-120: `gt_wh.sqrt()` is not indexed by the box)
    const float X = (pbx.x1 + pbx.x2) / 2.f, Y = (pbx.y1 + pbx.y2) / 2.f;
    const float gx = (gb.x1 + gb.x2) / 2.f, gy = (gb.y1 + gb.y2) / 2.f;
    float bbox = (gx - X) * (gx - X) + (gy - Y) * (gy - Y);
    const float sw = sqrtf(b[2]), sh = sqrtf(b[3]);
    float dsw = 0.f, dsh = 0.f;
    for (int k = gt_off[n]; k < gt_off[n + 1]; ++k) {
        const float a = sqrtf(gt[4 * k + 2] - gt[4 * k]) - sw, c = sqrtf(gt[4 * k + 3] - gt[4 * k + 1]) - sh;
        bbox += a * a + c * c;
        dsw += -2.f * a;
        dsh += -2.f * c;
    }
    const float diff = iou - o;
    if (!BWD) {
        atomicAdd(sums + 0, diff * diff);
        atomicAdd(sums + 2, bbox);
        atomicAdd(sums + 3, clf);
    } else {
        atomicAdd(dpo + slot, gc[0] * (-2.f * diff));
        float d[4] = {0.f, 0.f, 0.f, 0.f};
        iou_grad(pbx, gb, gc[0] * 2.f * diff, d);             // the objectness target is the differentiable IoU
        float dX = d[0] + d[2], dY = d[1] + d[3];
        float dw = 0.5f * (d[2] - d[0]), dh = 0.5f * (d[3] - d[1]);
        dX += gc[2] * (-2.f * (gx - X));
        dY += gc[2] * (-2.f * (gy - Y));
        dw += gc[2] * dsw * (0.5f / sw);
        dh += gc[2] * dsh * (0.5f / sh);
        atomicAdd(dpb + (long)slot * 4 + 0, cell_rel ? dX / (float)W : dX);
        atomicAdd(dpb + (long)slot * 4 + 1, cell_rel ? dY / (float)H : dY);
        atomicAdd(dpb + (long)slot * 4 + 2, dw);
        atomicAdd(dpb + (long)slot * 4 + 3, dh);
    }
}

// no-object term over every (cell, anchor) without a box, optionally skipping predictions whose best IoU with a box of the
// image reaches 0.5 (yolo.py:123-128)
template <bool BWD>
__global__ __launch_bounds__(256) void yolo1_noobj_kernel(const float* __restrict__ pb, const float* __restrict__ po,
                                                          const float* __restrict__ gt, const int* __restrict__ gt_off, int N, int H, int W,
                                                          int A, int cell_rel, int ignore_high_iou,
                                                          const unsigned char* __restrict__ mark, const float* __restrict__ gc,
                                                          float* __restrict__ sums, float* __restrict__ dpo) {
    const long total = (long)N * H * W * A;
    float acc = 0.f;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        if (mark[t]) continue;
        const long cell = t / A;
        const int cx = (int)(cell % W), cy = (int)((cell / W) % H);
        const int n = (int)(cell / ((long)W * H));
        if (ignore_high_iou) {
            const Box bx = isobox(pb + t * 4, cx, cy, W, H, cell_rel);
            float best = -1.f;
            for (int k = gt_off[n]; k < gt_off[n + 1]; ++k)
                best = fmaxf(best, iou_of(bx, Box{gt[4 * k], gt[4 * k + 1], gt[4 * k + 2], gt[4 * k + 3]}));
            if (best >= 0.5f) continue;
        }
        const float o = po[t];
        if (BWD) atomicAdd(dpo + t, gc[1] * 2.f * o);
        else acc += o * o;
    }
    if (!BWD) {
        __shared__ float sh[4];
        const float v = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(sums + 1, sh[0] + sh[1] + sh[2] + sh[3]);
    }
}

// to_isoboxes (+ clamp) and the eval-time score of post_process (yolo.py:165-215): score = max_c p_c * objectness
__global__ void yolo1_decode_kernel(const float* __restrict__ pb, const float* __restrict__ po, const float* __restrict__ ps, long total,
                                    int H, int W, int A, int nc, int cell_rel, int clamp01, float* __restrict__ boxes,
                                    float* __restrict__ score, long* __restrict__ label) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long cell = t / A;
        const int cx = (int)(cell % W), cy = (int)((cell / W) % H);
        Box b = isobox(pb + t * 4, cx, cy, W, H, cell_rel);
        if (clamp01) {
            b.x1 = fminf(fmaxf(b.x1, 0.f), 1.f); b.y1 = fminf(fmaxf(b.y1, 0.f), 1.f);
            b.x2 = fminf(fmaxf(b.x2, 0.f), 1.f); b.y2 = fminf(fmaxf(b.y2, 0.f), 1.f);
        }
        boxes[4 * t] = b.x1; boxes[4 * t + 1] = b.y1; boxes[4 * t + 2] = b.x2; boxes[4 * t + 3] = b.y2;
        if (score != nullptr) {
            float best = ps[t * nc];
            long bi = 0;
            for (int c = 1; c < nc; ++c) {
                const float v = ps[t * nc + c];
                if (v > best) { best = v; bi = c; }
            }
            score[t] = best * po[t];
            label[t] = bi;
        }
    }
}

inline int grid_for(long total, int threads = 256, int cap = 4096) {
    long b = (total + threads - 1) / threads;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// _format_outputs of YOLOv1 (yolo.py:314-334) and YOLOv2 (yolov2.py:175-200): raw head output -> boxes / objectness / class
// distribution, one thread per predictor (n, i, j, a).  The two layouts differ only in where a predictor's logits sit, so the host
// hands over element strides: box / objectness logit k of predictor (n, i, j, a) at n*sn + i*si + j*sj + a*sa + k*sk, class logit c
// at cls0 + n*sn + i*si + j*sj + a*ca + c*cc.  YOLOv1 (v2 == 0): four sigmoids, the class softmax is shared by the anchors of a
// cell (As = 1, ca = 0: anchor 0's thread owns it).  YOLOv2: sigmoid + cell offset over the grid size for the centre, anchor * exp
// for the size.
struct FmtLayout { long sn, si, sj, sa, sk, cls0, ca, cc; };

__global__ void yolo_format_fwd_kernel(const float* __restrict__ x, FmtLayout L, long total, int H, int W, int A, int As, int nc, int v2,
                                       const float* __restrict__ anchors, float* __restrict__ boxes, float* __restrict__ obj,
                                       float* __restrict__ scores) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int a = (int)(t % A);
        const long cell = t / A;
        const int j = (int)(cell % W), i = (int)((cell / W) % H);
        const long n = cell / ((long)W * H);
        const long pos = n * L.sn + i * L.si + j * L.sj;
        const float* xb = x + pos + a * L.sa;
        const float s0 = sigmoidf_(xb[0]), s1 = sigmoidf_(xb[L.sk]);
        float b0, b1, b2, b3;
        if (v2) {
            b0 = (s0 + (float)j) / (float)W;
            b1 = (s1 + (float)i) / (float)H;
            b2 = anchors[2 * a] * expf(xb[2 * L.sk]);
            b3 = anchors[2 * a + 1] * expf(xb[3 * L.sk]);
        } else {
            b0 = s0; b1 = s1;
            b2 = sigmoidf_(xb[2 * L.sk]);
            b3 = sigmoidf_(xb[3 * L.sk]);
        }
        boxes[4 * t] = b0; boxes[4 * t + 1] = b1; boxes[4 * t + 2] = b2; boxes[4 * t + 3] = b3;
        obj[t] = sigmoidf_(xb[4 * L.sk]);
        if (a < As) {
            const float* xc = x + L.cls0 + pos + a * L.ca;
            float* ps = scores + (cell * As + a) * nc;
            float mx = xc[0];
            for (int c = 1; c < nc; ++c) mx = fmaxf(mx, xc[c * L.cc]);
            float sum = 0.f;
            for (int c = 0; c < nc; ++c) {
                const float e = expf(xc[c * L.cc] - mx);
                ps[c] = e;
                sum += e;
            }
            for (int c = 0; c < nc; ++c) ps[c] = ps[c] / sum;
        }
    }
}

// gradient of the above: every logit belongs to exactly one predictor thread, so dx (layout Ld, same shape as x) is written once
// everywhere and needs no zero fill; a missing cotangent (nullptr) counts as zero.  p is the forward's class distribution.
__global__ void yolo_format_bwd_kernel(const float* __restrict__ x, FmtLayout L, FmtLayout Ld, long total, int H, int W, int A, int As,
                                       int nc, int v2, const float* __restrict__ anchors, const float* __restrict__ p,
                                       const float* __restrict__ gb, const float* __restrict__ go, const float* __restrict__ gs,
                                       float* __restrict__ dx) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int a = (int)(t % A);
        const long cell = t / A;
        const int j = (int)(cell % W), i = (int)((cell / W) % H);
        const long n = cell / ((long)W * H);
        const float* xb = x + n * L.sn + i * L.si + j * L.sj + a * L.sa;
        const long dpos = n * Ld.sn + i * Ld.si + j * Ld.sj;
        float* db = dx + dpos + a * Ld.sa;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        if (gb != nullptr) { g[0] = gb[4 * t]; g[1] = gb[4 * t + 1]; g[2] = gb[4 * t + 2]; g[3] = gb[4 * t + 3]; }
        const float s0 = sigmoidf_(xb[0]), s1 = sigmoidf_(xb[L.sk]);
        if (v2) {
            db[0] = g[0] / (float)W * (s0 * (1.f - s0));
            db[Ld.sk] = g[1] / (float)H * (s1 * (1.f - s1));
            db[2 * Ld.sk] = g[2] * (anchors[2 * a] * expf(xb[2 * L.sk]));
            db[3 * Ld.sk] = g[3] * (anchors[2 * a + 1] * expf(xb[3 * L.sk]));
        } else {
            const float s2 = sigmoidf_(xb[2 * L.sk]), s3 = sigmoidf_(xb[3 * L.sk]);
            db[0] = g[0] * (s0 * (1.f - s0));
            db[Ld.sk] = g[1] * (s1 * (1.f - s1));
            db[2 * Ld.sk] = g[2] * (s2 * (1.f - s2));
            db[3 * Ld.sk] = g[3] * (s3 * (1.f - s3));
        }
        const float s4 = sigmoidf_(xb[4 * L.sk]);
        db[4 * Ld.sk] = (go != nullptr ? go[t] : 0.f) * (s4 * (1.f - s4));
        if (a < As) {
            float* dc = dx + Ld.cls0 + dpos + a * Ld.ca;
            const long so = (cell * As + a) * nc;
            if (gs == nullptr) {
                for (int c = 0; c < nc; ++c) dc[c * Ld.cc] = 0.f;
            } else {
                float dot = 0.f;
                for (int c = 0; c < nc; ++c) dot += gs[so + c] * p[so + c];
                for (int c = 0; c < nc; ++c) dc[c * Ld.cc] = p[so + c] * (gs[so + c] - dot);
            }
        }
    }
}

}  // namespace

extern "C" {

int hc_yolo_decode(const void* logits, int32_t dtype, int64_t sn, int64_t sc, int64_t sp, int32_t N, int32_t H, int32_t W, int32_t A,
                   int32_t num_classes, const float* anchors, float scale_xy, float* boxes, float* obj, float* score, int64_t* label,
                   int32_t clamp01, hc_stream_t stream) {
    if (logits == nullptr || anchors == nullptr || boxes == nullptr || A <= 0 || num_classes < 0 || (dtype != 0 && dtype != 1))
        return HC_ERR_ARG;
    if ((score == nullptr) != (label == nullptr) || (score != nullptr && obj == nullptr)) return HC_ERR_ARG;
    const long total = (long)N * H * W * A;
    if (total == 0) return HC_OK;
    const Logits x = {logits, dtype, (long)sn, (long)sc, (long)sp};
    hipLaunchKernelGGL(yolo_decode_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, N, H, W, A, num_classes,
                       anchors, scale_xy, boxes, obj, score, (long*)label, clamp01);
    return hc_launch_status();
}

int hc_yolo_assign(const float* gt_boxes, const int32_t* gt_img, int32_t G, const float* anchors, int32_t N, int32_t H, int32_t W,
                   int32_t A, uint8_t* obj_mask, uint8_t* cell_gt, hc_stream_t stream) {
    if (obj_mask == nullptr || cell_gt == nullptr || anchors == nullptr || G < 0 || A <= 0) return HC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t cells = (size_t)N * H * W;
    if (hc_zero_async(obj_mask, cells * A, st) != hipSuccess) return HC_ERR_LAUNCH;
    if (hc_zero_async(cell_gt, cells, st) != hipSuccess) return HC_ERR_LAUNCH;
    if (G == 0) return HC_OK;
    if (gt_boxes == nullptr || gt_img == nullptr) return HC_ERR_ARG;
    hipLaunchKernelGGL(yolo_assign_kernel, dim3((G + 127) / 128), dim3(128), 0, st, gt_boxes, gt_img, G, anchors, H, W, A, obj_mask,
                       cell_gt);
    return hc_launch_status();
}

int hc_yolo_loss_fwd(const void* logits, int32_t dtype, int64_t sn, int64_t sc, int64_t sp, int32_t N, int32_t H, int32_t W, int32_t A,
                     int32_t num_classes, const float* anchors, float scale_xy, const float* gt_boxes, const int64_t* gt_labels,
                     const int32_t* gt_off, const uint8_t* obj_mask, const uint8_t* cell_gt, float* sums, hc_stream_t stream) {
    if (logits == nullptr || anchors == nullptr || gt_off == nullptr || obj_mask == nullptr || cell_gt == nullptr || sums == nullptr ||
        (dtype != 0 && dtype != 1) || A <= 0)
        return HC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hc_zero_async(sums, 4 * sizeof(float), st) != hipSuccess) return HC_ERR_LAUNCH;
    const long total = (long)N * H * W * A;
    if (total == 0) return HC_OK;
    const Logits x = {logits, dtype, (long)sn, (long)sc, (long)sp};
    const Grads none = {nullptr, 0, 0, 0, 0};
    hipLaunchKernelGGL((yolo_loss_kernel<false>), dim3(grid_for(total, 256, 1024)), dim3(256), 0, st, x, none, N, H, W, A, num_classes,
                       anchors, scale_xy, gt_boxes, (const long*)gt_labels, gt_off, obj_mask, cell_gt, sums, (const float*)nullptr);
    return hc_launch_status();
}

int hc_yolo_loss_bwd(const void* logits, int32_t dtype, int64_t sn, int64_t sc, int64_t sp, int32_t N, int32_t H, int32_t W, int32_t A,
                     int32_t num_classes, const float* anchors, float scale_xy, const float* gt_boxes, const int64_t* gt_labels,
                     const int32_t* gt_off, const uint8_t* obj_mask, const uint8_t* cell_gt, const float* gcoef, void* dlogits,
                     hc_stream_t stream) {
    if (logits == nullptr || anchors == nullptr || gt_off == nullptr || obj_mask == nullptr || cell_gt == nullptr || gcoef == nullptr ||
        dlogits == nullptr || (dtype != 0 && dtype != 1) || A <= 0)
        return HC_ERR_ARG;
    const long total = (long)N * H * W * A;
    if (total == 0) return HC_OK;
    const Logits x = {logits, dtype, (long)sn, (long)sc, (long)sp};
    const Grads dx = {dlogits, dtype, (long)sn, (long)sc, (long)sp};
    hipLaunchKernelGGL((yolo_loss_kernel<true>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, dx, N, H, W, A,
                       num_classes, anchors, scale_xy, gt_boxes, (const long*)gt_labels, gt_off, obj_mask, cell_gt, (float*)nullptr,
                       gcoef);
    return hc_launch_status();
}

int hc_yolo1_loss_fwd(const float* pred_boxes, const float* pred_o, const float* pred_scores, int32_t N, int32_t H, int32_t W, int32_t A,
                      int32_t As, int32_t nc, int32_t cell_rel, int32_t ignore_high_iou, const float* gt_boxes, const int64_t* gt_labels,
                      const int32_t* gt_img, const int32_t* gt_off, int32_t G, int32_t* assign, uint8_t* mark, float* sums,
                      hc_stream_t stream) {
    if (pred_boxes == nullptr || pred_o == nullptr || pred_scores == nullptr || gt_off == nullptr || mark == nullptr || sums == nullptr ||
        N < 0 || H <= 0 || W <= 0 || A <= 0 || (As != 1 && As != A) || nc <= 0 || G < 0)
        return HC_ERR_ARG;
    if (G > 0 && (gt_boxes == nullptr || gt_labels == nullptr || gt_img == nullptr || assign == nullptr)) return HC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)N * H * W * A;
    if (hc_zero_async(mark, (size_t)total, st) != hipSuccess) return HC_ERR_LAUNCH;
    if (hc_zero_async(sums, 4 * sizeof(float), st) != hipSuccess) return HC_ERR_LAUNCH;
    if (G > 0) {
        // the IoU of the assignment is recomputed by the loss kernel; iou_out reuses the tail of `assign` ([2 * G] ints)
        hipLaunchKernelGGL(yolo1_assign_kernel, dim3((G + 63) / 64), dim3(64), 0, st, pred_boxes, gt_boxes, (const int*)gt_img, G, H, W, A,
                           cell_rel, (int*)assign, reinterpret_cast<float*>(assign + G), (unsigned char*)mark);
        hipLaunchKernelGGL((yolo1_gt_kernel<false>), dim3((G + 63) / 64), dim3(64), 0, st, pred_boxes, pred_o, pred_scores, gt_boxes,
                           (const long*)gt_labels, (const int*)gt_img, (const int*)gt_off, G, H, W, A, As, nc, cell_rel, (const int*)assign,
                           (const float*)nullptr, sums, (float*)nullptr, (float*)nullptr, (float*)nullptr);
    }
    if (total > 0)
        hipLaunchKernelGGL((yolo1_noobj_kernel<false>), dim3(grid_for(total)), dim3(256), 0, st, pred_boxes, pred_o, gt_boxes,
                           (const int*)gt_off, N, H, W, A, cell_rel, ignore_high_iou, (const unsigned char*)mark, (const float*)nullptr, sums,
                           (float*)nullptr);
    return hc_launch_status();
}

int hc_yolo1_loss_bwd(const float* pred_boxes, const float* pred_o, const float* pred_scores, int32_t N, int32_t H, int32_t W, int32_t A,
                      int32_t As, int32_t nc, int32_t cell_rel, int32_t ignore_high_iou, const float* gt_boxes, const int64_t* gt_labels,
                      const int32_t* gt_img, const int32_t* gt_off, int32_t G, const int32_t* assign, const uint8_t* mark,
                      const float* grad_sums, float* d_boxes, float* d_o, float* d_scores, hc_stream_t stream) {
    if (pred_boxes == nullptr || pred_o == nullptr || pred_scores == nullptr || gt_off == nullptr || mark == nullptr ||
        grad_sums == nullptr || d_boxes == nullptr || d_o == nullptr || d_scores == nullptr || N < 0 || H <= 0 || W <= 0 || A <= 0 ||
        (As != 1 && As != A) || nc <= 0 || G < 0)
        return HC_ERR_ARG;
    if (G > 0 && (gt_boxes == nullptr || gt_labels == nullptr || gt_img == nullptr || assign == nullptr)) return HC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)N * H * W * A;
    if (hc_zero_async(d_boxes, sizeof(float) * 4 * (size_t)total, st) != hipSuccess) return HC_ERR_LAUNCH;
    if (hc_zero_async(d_o, sizeof(float) * (size_t)total, st) != hipSuccess) return HC_ERR_LAUNCH;
    if (hc_zero_async(d_scores, sizeof(float) * (size_t)N * H * W * As * nc, st) != hipSuccess) return HC_ERR_LAUNCH;
    if (G > 0)
        hipLaunchKernelGGL((yolo1_gt_kernel<true>), dim3((G + 63) / 64), dim3(64), 0, st, pred_boxes, pred_o, pred_scores, gt_boxes,
                           (const long*)gt_labels, (const int*)gt_img, (const int*)gt_off, G, H, W, A, As, nc, cell_rel, (const int*)assign,
                           grad_sums, (float*)nullptr, d_boxes, d_o, d_scores);
    if (total > 0)
        hipLaunchKernelGGL((yolo1_noobj_kernel<true>), dim3(grid_for(total)), dim3(256), 0, st, pred_boxes, pred_o, gt_boxes,
                           (const int*)gt_off, N, H, W, A, cell_rel, ignore_high_iou, (const unsigned char*)mark, grad_sums, (float*)nullptr,
                           d_o);
    return hc_launch_status();
}

int hc_yolo1_decode(const float* b_coords, const float* b_o, const float* b_scores, int32_t N, int32_t H, int32_t W, int32_t A, int32_t nc,
                    int32_t cell_rel, int32_t clamp01, float* boxes, float* score, int64_t* label, hc_stream_t stream) {
    if (b_coords == nullptr || boxes == nullptr || N < 0 || H <= 0 || W <= 0 || A <= 0) return HC_ERR_ARG;
    if ((score == nullptr) != (label == nullptr)) return HC_ERR_ARG;
    if (score != nullptr && (b_o == nullptr || b_scores == nullptr || nc <= 0)) return HC_ERR_ARG;
    const long total = (long)N * H * W * A;
    if (total == 0) return HC_OK;
    hipLaunchKernelGGL(yolo1_decode_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, b_coords, b_o, b_scores, total, H, W, A,
                       nc, cell_rel, clamp01, boxes, score, (long*)label);
    return hc_launch_status();
}

static inline FmtLayout fmt_layout(const int64_t* s) { return FmtLayout{s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]}; }

int hc_yolo_format_fwd(const float* x, const int64_t* layout, int32_t N, int32_t H, int32_t W, int32_t A, int32_t As, int32_t nc,
                       int32_t v2, const float* anchors, float* boxes, float* obj, float* scores, hc_stream_t stream) {
    if (x == nullptr || layout == nullptr || boxes == nullptr || obj == nullptr || scores == nullptr) return HC_ERR_ARG;
    if (N < 0 || H <= 0 || W <= 0 || A <= 0 || nc <= 0 || (As != 1 && As != A) || (v2 != 0 && anchors == nullptr)) return HC_ERR_ARG;
    const long total = (long)N * H * W * A;
    if (total == 0) return HC_OK;
    hipLaunchKernelGGL(yolo_format_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, fmt_layout(layout), total, H, W,
                       A, As, nc, v2, anchors, boxes, obj, scores);
    return hc_launch_status();
}

int hc_yolo_format_bwd(const float* x, const int64_t* layout, const int64_t* dx_layout, int32_t N, int32_t H, int32_t W, int32_t A,
                       int32_t As, int32_t nc, int32_t v2, const float* anchors, const float* scores, const float* g_boxes,
                       const float* g_obj, const float* g_scores, float* dx, hc_stream_t stream) {
    if (x == nullptr || layout == nullptr || dx_layout == nullptr || dx == nullptr) return HC_ERR_ARG;
    if (N < 0 || H <= 0 || W <= 0 || A <= 0 || nc <= 0 || (As != 1 && As != A) || (v2 != 0 && anchors == nullptr)) return HC_ERR_ARG;
    if (g_scores != nullptr && scores == nullptr) return HC_ERR_ARG;
    const long total = (long)N * H * W * A;
    if (total == 0) return HC_OK;
    hipLaunchKernelGGL(yolo_format_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, fmt_layout(layout),
                       fmt_layout(dx_layout), total, H, W, A, As, nc, v2, anchors, scores, g_boxes, g_obj, g_scores, dx);
    return hc_launch_status();
}

}  // extern "C"
