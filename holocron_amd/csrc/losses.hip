// Poly-1 loss, Dice loss partial sums and DropBlock (reference: holocron/nn/functional.py:465-613).
// All three are HBM/latency-bound pointwise or small-reduction work on fp32 NCHW-style [N][K][S] tensors.
#include "common.h"
#include "../../include/holocron_hip.h"

namespace {

inline int grid_for(long total, int threads = 256, int cap = 8192) {
    long b = (total + threads - 1) / threads;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// ---------------------------------------------------------------- poly-1 loss (functional.py:540-613)
// hard labels: one thread per (n, s) position; loss = -logpt + eps (1 - exp(logpt)), times weight[target]
__global__ void poly_hard_fwd_kernel(const float* __restrict__ x, const long* __restrict__ target, const float* __restrict__ weight,
                                     float* __restrict__ loss_el, uint8_t* __restrict__ valid, int N, int K, long S,
                                     int ignore_index, float eps) {
    const long total = (long)N * S;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long n = t / S, s = t % S;
        const float* px = x + n * K * S + s;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, px[(long)k * S]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(px[(long)k * S] - mx);
        const long tg = target[t];
        const float logpt = (px[tg * S] - mx) - logf(se);
        float l = -1.f * logpt + eps * (1.f - expf(logpt));
        if (weight != nullptr) l = weight[tg] * l;
        loss_el[t] = l;
        valid[t] = !(ignore_index >= 0 && ignore_index < K && tg == ignore_index);
    }
}
__global__ void poly_hard_bwd_kernel(const float* __restrict__ x, const long* __restrict__ target, const float* __restrict__ weight,
                                     const float* __restrict__ dloss, float* __restrict__ dx, int N, int K, long S, float eps) {
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
        const float pt = expf((px[tg * S] - mx) - lse);
        const float w = weight != nullptr ? weight[tg] : 1.f;
        const float gup = dloss[t] * w * (-1.f - eps * pt);  // dL/dlogpt
        for (int k = 0; k < K; ++k) {
            const float pk = expf((px[(long)k * S] - mx) - lse);
            pdx[(long)k * S] = gup * ((k == tg ? 1.f : 0.f) - pk);
        }
    }
}
// soft labels: l_k = logp_k * t_k ; per-position output = sum over kept classes of w_k (-l_k + eps (1 - exp l_k))
__global__ void poly_soft_fwd_kernel(const float* __restrict__ x, const float* __restrict__ target, const float* __restrict__ weight,
                                     float* __restrict__ loss_pos, int N, int K, long S, int ignore_index, float eps) {
    const long total = (long)N * S;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long n = t / S, s = t % S;
        const float* px = x + n * K * S + s;
        const float* pt = target + n * K * S + s;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, px[(long)k * S]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(px[(long)k * S] - mx);
        const float lse = logf(se);
        float acc = 0.f;
        for (int k = 0; k < K; ++k) {
            if (k == ignore_index) continue;  // only reachable for 0 <= ignore_index < K
            const float l = ((px[(long)k * S] - mx) - lse) * pt[(long)k * S];
            float v = -1.f * l + eps * (1.f - expf(l));
            if (weight != nullptr) v = weight[k] * v;
            acc += v;
        }
        loss_pos[t] = acc;
    }
}
__global__ void poly_soft_bwd_kernel(const float* __restrict__ x, const float* __restrict__ target, const float* __restrict__ weight,
                                     const float* __restrict__ dloss, float* __restrict__ dx, int N, int K, long S,
                                     int ignore_index, float eps) {
    const long total = (long)N * S;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long n = t / S, s = t % S;
        const float* px = x + n * K * S + s;
        const float* pt = target + n * K * S + s;
        float* pdx = dx + n * K * S + s;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, px[(long)k * S]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += expf(px[(long)k * S] - mx);
        const float lse = logf(se);
        const float gup = dloss[t];
        // g_k = dL/dlogp_k ; dx_j = g_j - p_j * sum_k g_k
        float gsum = 0.f;
        for (int k = 0; k < K; ++k) {
            if (k == ignore_index) continue;
            const float tk = pt[(long)k * S];
            const float l = ((px[(long)k * S] - mx) - lse) * tk;
            const float w = weight != nullptr ? weight[k] : 1.f;
            gsum += gup * w * (-1.f - eps * expf(l)) * tk;
        }
        for (int k = 0; k < K; ++k) {
            const float logp = (px[(long)k * S] - mx) - lse;
            float g = 0.f;
            if (k != ignore_index) {
                const float tk = pt[(long)k * S];
                const float w = weight != nullptr ? weight[k] : 1.f;
                g = gup * w * (-1.f - eps * expf(logp * tk)) * tk;
            }
            pdx[(long)k * S] = g - expf(logp) * gsum;
        }
    }
}

// ---------------------------------------------------------------- dice partial sums (functional.py:503-537)
// sums[0][k] = sum x*t ; sums[1][k] = sum x ; sums[2][k] = sum t over (n, s).  grid (blocks, K)
__global__ __launch_bounds__(256) void dice_sums_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                        float* __restrict__ sums, int N, int K, long S) {
    const int k = blockIdx.y;
    const long total = (long)N * S;
    float a = 0.f, b = 0.f, c = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / S, s = i % S;
        const long off = (n * K + k) * S + s;
        const float xv = x[off], tv = t[off];
        a += xv * tv;
        b += xv;
        c += tv;
    }
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    __shared__ float sh[3][4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = a; sh[1][w] = b; sh[2][w] = c; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float v = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
        atomicAdd(&sums[threadIdx.x * K + k], v);
    }
}
// dx = d(sum x*t)_k * t + d(sum x)_k
__global__ void dice_bwd_kernel(const float* __restrict__ t, const float* __restrict__ dsums, float* __restrict__ dx, int N, int K,
                                long S) {
    const long total = (long)N * K * S;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)((i / S) % K);
        dx[i] = dsums[k] * t[i] + dsums[K + k];
    }
}

// ---------------------------------------------------------------- DropBlock (functional.py:465-500)
// keep[n,h,w] = 1 - max over the bs x bs window (stride 1, pad bs/2) of (noise <= gamma); count += sum keep
__global__ __launch_bounds__(256) void dropblock_mask_kernel(const float* __restrict__ noise, float* __restrict__ keep,
                                                             float* __restrict__ count, int N, int H, int W, int bs, float gamma) {
    const long total = (long)N * H * W;
    const int r = bs / 2;
    float local = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W), h = (int)((i / W) % H);
        const long n = i / ((long)W * H);
        const float* pn = noise + n * H * W;
        bool drop = false;
        for (int dy = -r; dy <= r; ++dy) {
            const int yy = h + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -r; dx <= r; ++dx) {
                const int xx = w + dx;
                if (xx < 0 || xx >= W) continue;
                drop |= pn[yy * W + xx] <= gamma;
            }
        }
        const float kv = drop ? 0.f : 1.f;
        keep[i] = kv;
        local += kv;
    }
    local = wave_sum(local);
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(count, sh[0] + sh[1] + sh[2] + sh[3]);
}
// y = x * keep[n,h,w] * (count > 0 ? numel / count : 1).  layout 0: [N][C][HW] (NCHW), 1: [N][HW][C] (NHWC)
template <bool BF16>
__global__ void dropblock_apply_kernel(const void* __restrict__ xv, const float* __restrict__ keep, const float* __restrict__ count,
                                       void* __restrict__ yv, long N, int Cc, long HW, int nhwc) {
    const long total = N * Cc * HW;
    const float cnt = count[0];
    const float scale = cnt > 0.f ? (float)(N * HW) / cnt : 1.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long pos;
        if (nhwc) pos = i / Cc;
        else pos = (i / (HW * Cc)) * HW + (i % HW);
        const float m = keep[pos];
        // x *= mask ; x *= numel / one_count (two roundings, like the reference's two in-place multiplies)
        if (BF16) {
            float v = bf16_to_f32(((const bf16_t*)xv)[i]) * m;
            v = bf16_to_f32(f32_to_bf16(v));
            ((bf16_t*)yv)[i] = f32_to_bf16(v * scale);
        } else {
            ((float*)yv)[i] = (((const float*)xv)[i] * m) * scale;
        }
    }
}

// all DropBlock masks of a training step in ONE launch.  noise / keep are arenas, item.off the layer's first element in both;
// counts[layer] += sum(keep).  A workgroup walks 32 x 32 output tiles: the (32 + 2r)^2 halo of block centres goes to LDS once, the
// r-dilation is separable (row OR, then column OR) - 2 (2r + 1) LDS reads per pixel instead of (2r + 1)^2 global loads (the 608^2 map
// of YOLOv4's stem alone is 5.9 M pixels x 49).
// The tiles of ALL layers form one flat list walked by a persistent grid (per-layer tile counts scanned into LDS by every workgroup,
// a tile's layer found by bisection): the former (tiles of the largest layer) x (layers) grid launched 512 workgroups for each of
// YOLOv4's 123 layers, 55 of which have 16 tiles - 196 us per step, most of it workgroups with nothing to do.  The counts are sums
// of zeros and ones (exact in fp32 whatever the order): bit-identical to the per-layer kernel.
constexpr int DB_T = 32, DB_RMAX = 6;      // tile edge, largest supported radius (block_size <= 13)
constexpr int DB_MAXI = 256;               // layers per launch (the host side chunks longer lists)
__global__ __launch_bounds__(256) void dropblock_mask_batched_kernel(const hc_drop_item* __restrict__ items, int nitems,
                                                                     const float* __restrict__ noise, float* __restrict__ keep,
                                                                     float* __restrict__ counts) {
    __shared__ unsigned char cen[(DB_T + 2 * DB_RMAX) * (DB_T + 2 * DB_RMAX)];
    __shared__ unsigned char rowor[(DB_T + 2 * DB_RMAX) * DB_T];
    __shared__ int pre[DB_MAXI + 1];
    __shared__ hc_drop_item its[DB_MAXI];
    {   // pre[i] = tiles of the layers before i: thread i holds layer i, one workgroup-wide scan through LDS
        int mine = 0;
        if ((int)threadIdx.x < nitems) {
            const hc_drop_item it = items[threadIdx.x];
            its[threadIdx.x] = it;
            mine = it.N * ((it.H + DB_T - 1) / DB_T) * ((it.W + DB_T - 1) / DB_T);
        }
        pre[threadIdx.x + 1 <= DB_MAXI ? threadIdx.x + 1 : DB_MAXI] = mine;
        if (threadIdx.x == 0) pre[0] = 0;
        __syncthreads();
        for (int d = 1; d < DB_MAXI; d <<= 1) {            // Hillis-Steele over pre[1..256]
            const int idx = threadIdx.x + 1;
            const int add = idx - d >= 1 ? pre[idx - d] : 0;
            __syncthreads();
            pre[idx] += add;
            __syncthreads();
        }
    }
    // a workgroup owns a CONTIGUOUS run of tiles, so it stays inside one layer for most of it and adds to that layer's count once
    // (same-address atomics serialise in the L2: one per wave and tile made the launch 1.1 ms)
    __shared__ float sh[4];
    const int total = pre[nitems];
    const int chunk = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int f0 = (int)blockIdx.x * chunk, f1 = f0 + chunk < total ? f0 + chunk : total;
    int cur = -1;
    float local = 0.f;
    auto flush = [&]() {                                     // uniform over the workgroup
        if (cur < 0) return;
        const float w = wave_sum(local);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float v = sh[0] + sh[1] + sh[2] + sh[3];
            if (v != 0.f) atomicAdd(counts + cur, v);
        }
        local = 0.f;
    };
    for (int flat = f0; flat < f1; ++flat) {
        int lo = 0, hi = nitems;                             // largest lo with pre[lo] <= flat (uniform over the workgroup)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= flat) lo = mid; else hi = mid;
        }
        if (lo != cur) {
            flush();
            cur = lo;
        }
        const hc_drop_item it = its[lo];
        const int tile = flat - pre[lo];
        const int H = it.H, W = it.W, r = it.block_size / 2;
        const int E = DB_T + 2 * r;                                  // halo tile edge
        const unsigned einv = (1u << 20) / (unsigned)E + 1u;         // i / E for i < E * E <= 1936 as a multiply (exact there)
        const int th = (H + DB_T - 1) / DB_T, tw = (W + DB_T - 1) / DB_T;
        const int tx = tile % tw, ty = (tile / tw) % th;
        const long n = tile / (tw * th);
        const int y0 = ty * DB_T - r, x0 = tx * DB_T - r;
        const float* pn = noise + it.off + n * H * W;
        float* pk = keep + it.off;
        __syncthreads();
        for (int i = threadIdx.x; i < E * E; i += 256) {
            const int iy = (int)(((unsigned)i * einv) >> 20);
            const int yy = y0 + iy, xx = x0 + (i - iy * E);
            cen[i] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W && pn[yy * W + xx] <= it.gamma) ? 1 : 0;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < E * DB_T; i += 256) {       // row OR over 2r + 1 columns
            const int row = i / DB_T, col = i % DB_T;
            unsigned char v = 0;
            for (int d = 0; d <= 2 * r; ++d) v |= cen[row * E + col + d];
            rowor[i] = v;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < DB_T * DB_T; i += 256) {    // column OR over 2r + 1 rows
            const int row = i / DB_T, col = i % DB_T;
            const int y = ty * DB_T + row, x = tx * DB_T + col;
            if (y >= H || x >= W) continue;
            unsigned char v = 0;
            for (int d = 0; d <= 2 * r; ++d) v |= rowor[(row + d) * DB_T + col];
            const float kv = v ? 0.f : 1.f;
            pk[(n * H + y) * W + x] = kv;
            local += kv;
        }
    }
    flush();
}

}  // namespace

extern "C" {

int hc_poly_loss_hard_fwd(const float* x, const int64_t* target, const float* weight, float* loss_el, uint8_t* valid, int32_t N,
                          int32_t K, int64_t S, int32_t ignore_index, float eps, hc_stream_t stream) {
    if (x == nullptr || target == nullptr || loss_el == nullptr || valid == nullptr || K <= 0) return HC_ERR_ARG;
    if ((long)N * S == 0) return HC_OK;
    hipLaunchKernelGGL(poly_hard_fwd_kernel, dim3(grid_for((long)N * S)), dim3(256), 0, (hipStream_t)stream, x,
                       (const long*)target, weight, loss_el, valid, N, K, (long)S, ignore_index, eps);
    return hc_launch_status();
}
int hc_poly_loss_hard_bwd(const float* x, const int64_t* target, const float* weight, const float* dloss_el, float* dx, int32_t N,
                          int32_t K, int64_t S, float eps, hc_stream_t stream) {
    if (x == nullptr || target == nullptr || dloss_el == nullptr || dx == nullptr || K <= 0) return HC_ERR_ARG;
    if ((long)N * S == 0) return HC_OK;
    hipLaunchKernelGGL(poly_hard_bwd_kernel, dim3(grid_for((long)N * S)), dim3(256), 0, (hipStream_t)stream, x,
                       (const long*)target, weight, dloss_el, dx, N, K, (long)S, eps);
    return hc_launch_status();
}
int hc_poly_loss_soft_fwd(const float* x, const float* target, const float* weight, float* loss_pos, int32_t N, int32_t K,
                          int64_t S, int32_t ignore_index, float eps, hc_stream_t stream) {
    if (x == nullptr || target == nullptr || loss_pos == nullptr || K <= 0) return HC_ERR_ARG;
    if ((long)N * S == 0) return HC_OK;
    const int ign = (ignore_index >= 0 && ignore_index < K) ? ignore_index : -1;
    hipLaunchKernelGGL(poly_soft_fwd_kernel, dim3(grid_for((long)N * S)), dim3(256), 0, (hipStream_t)stream, x, target, weight,
                       loss_pos, N, K, (long)S, ign, eps);
    return hc_launch_status();
}
int hc_poly_loss_soft_bwd(const float* x, const float* target, const float* weight, const float* dloss_pos, float* dx, int32_t N,
                          int32_t K, int64_t S, int32_t ignore_index, float eps, hc_stream_t stream) {
    if (x == nullptr || target == nullptr || dloss_pos == nullptr || dx == nullptr || K <= 0) return HC_ERR_ARG;
    if ((long)N * S == 0) return HC_OK;
    const int ign = (ignore_index >= 0 && ignore_index < K) ? ignore_index : -1;
    hipLaunchKernelGGL(poly_soft_bwd_kernel, dim3(grid_for((long)N * S)), dim3(256), 0, (hipStream_t)stream, x, target, weight,
                       dloss_pos, dx, N, K, (long)S, ign, eps);
    return hc_launch_status();
}

int hc_dice_sums(const float* x, const float* target, float* sums, int32_t N, int32_t K, int64_t S, hc_stream_t stream) {
    if (x == nullptr || target == nullptr || sums == nullptr || K <= 0 || N < 0 || S < 0) return HC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hc_zero_async(sums, sizeof(float) * 3 * K, st) != hipSuccess) return HC_ERR_LAUNCH;
    if ((long)N * S == 0) return HC_OK;
    int bx = grid_for((long)N * S, 256, 1024);
    if ((long)bx * K > 16384) bx = (int)(16384 / K > 0 ? 16384 / K : 1);
    hipLaunchKernelGGL(dice_sums_kernel, dim3(bx, K), dim3(256), 0, st, x, target, sums, N, K, (long)S);
    return hc_launch_status();
}
int hc_dice_bwd(const float* target, const float* dsums, float* dx, int32_t N, int32_t K, int64_t S, hc_stream_t stream) {
    if (target == nullptr || dsums == nullptr || dx == nullptr || K <= 0) return HC_ERR_ARG;
    if ((long)N * S == 0) return HC_OK;
    hipLaunchKernelGGL(dice_bwd_kernel, dim3(grid_for((long)N * K * S)), dim3(256), 0, (hipStream_t)stream, target, dsums, dx, N, K,
                       (long)S);
    return hc_launch_status();
}

int hc_dropblock_mask(const float* noise, float* keep, float* count, int32_t N, int32_t H, int32_t W, int32_t block_size,
                      float gamma, hc_stream_t stream) {
    if (noise == nullptr || keep == nullptr || count == nullptr || block_size < 1 || (block_size & 1) == 0) return HC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hc_zero_async(count, sizeof(float), st) != hipSuccess) return HC_ERR_LAUNCH;
    if ((long)N * H * W == 0) return HC_OK;
    hipLaunchKernelGGL(dropblock_mask_kernel, dim3(grid_for((long)N * H * W, 256, 2048)), dim3(256), 0, st, noise, keep, count, N,
                       H, W, block_size, gamma);
    return hc_launch_status();
}
int hc_dropblock_apply(const void* x, const float* keep, const float* count, void* y, int64_t N, int32_t C, int64_t HW,
                       int32_t dtype, int32_t nhwc, hc_stream_t stream) {
    if (x == nullptr || keep == nullptr || count == nullptr || y == nullptr || (dtype != 0 && dtype != 1)) return HC_ERR_ARG;
    const long total = (long)N * C * HW;
    if (total == 0) return HC_OK;
    if (dtype == 0)
        hipLaunchKernelGGL(dropblock_apply_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, keep, count, y,
                           (long)N, C, (long)HW, nhwc);
    else
        hipLaunchKernelGGL(dropblock_apply_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, keep, count, y,
                           (long)N, C, (long)HW, nhwc);
    return hc_launch_status();
}

int hc_dropblock_mask_batched(const hc_drop_item* items, int32_t nitems, int64_t max_pixels, const float* noise, float* keep,
                              float* counts, hc_stream_t stream) {
    if (nitems < 0 || (nitems > 0 && (items == nullptr || noise == nullptr || keep == nullptr || counts == nullptr))) return HC_ERR_ARG;
    if (nitems == 0) return HC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hc_zero_async(counts, sizeof(float) * (size_t)nitems, st) != hipSuccess) return HC_ERR_LAUNCH;
    (void)max_pixels;                                     // (sized the former per-layer grid)
    for (int c = 0; c < nitems; c += DB_MAXI) {
        const int nc = nitems - c < DB_MAXI ? nitems - c : DB_MAXI;
        hipLaunchKernelGGL(dropblock_mask_batched_kernel, dim3(2048), dim3(256), 0, st, items + c, nc, noise, keep, counts + c);
    }
    return hc_launch_status();
}

}  // extern "C"
