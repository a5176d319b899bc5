// Multi-tensor optimizer steps: one launch for all parameters instead of ~10 torch kernels per
// tensor (reference: holocron/optim/adabelief.py:121-167, holocron/optim/lars.py:90-135).
// HBM-bound: AdaBelief moves 28 B/param (theta r/w, g r, m r/w, s r/w).
#include "common.h"
#include "../../include/holocron_hip.h"

namespace {

__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__global__ __launch_bounds__(256) void adabelief_kernel(const hc_mt_chunk* __restrict__ chunks,
                                                         const hc_adabelief_group* __restrict__ groups) {
    const hc_mt_chunk ck = chunks[blockIdx.x];
    const hc_adabelief_group gr = groups[ck.group];
    // bias corrections in double like the python reference (adabelief.py:146-147)
    const double bc1 = 1.0 - pow(gr.beta1, (double)gr.step);
    const double bc2 = 1.0 - pow(gr.beta2, (double)gr.step);
    const float beta1 = (float)gr.beta1, beta2 = (float)gr.beta2;
    const float omb1 = (float)(1.0 - gr.beta1), omb2 = (float)(1.0 - gr.beta2);
    const float sqrt_bc2 = (float)sqrt(bc2);
    const float eps = (float)gr.eps;
    const float neg_step = (float)(-(gr.lr / bc1));
    const float wd = (float)gr.weight_decay;
    const bool ams = gr.amsgrad != 0 && ck.smax != nullptr;

    auto upd = [&](float& p, float g, float& m, float& s, float* smax) {
        if (wd != 0.f) g = g + wd * p;          // grad.add(param, alpha=weight_decay)
        m = m * beta1 + omb1 * g;               // exp_avg.mul_(beta1).add_(grad, alpha=1-beta1)
        const float r = g - m;                  // grad_residual
        s = s * beta2 + omb2 * (r * r);         // exp_avg_sq.mul_(beta2).addcmul_(r, r, value=1-beta2)
        float den;
        if (smax != nullptr) {
            *smax = fmaxf(*smax, s);
            den = sqrtf(*smax) / sqrt_bc2 + eps;
        } else {
            den = sqrtf(s) / sqrt_bc2 + eps;
        }
        p = p + neg_step * (m / den);           // param.addcdiv_(exp_avg, denom, value=-step_size)
    };

    const int n = ck.n;
    const bool vec = !ams && aligned16(ck.p) && aligned16(ck.g) && aligned16(ck.m) && aligned16(ck.s);
    if (vec) {
        const int n4 = n >> 2;
        f32x4* p4 = reinterpret_cast<f32x4*>(ck.p);
        const f32x4* g4 = reinterpret_cast<const f32x4*>(ck.g);
        f32x4* m4 = reinterpret_cast<f32x4*>(ck.m);
        f32x4* s4 = reinterpret_cast<f32x4*>(ck.s);
        for (int i = threadIdx.x; i < n4; i += 256) {
            f32x4 p = p4[i], g = g4[i], m = m4[i], s = s4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[e], me = m[e], se = s[e];
                upd(pe, g[e], me, se, nullptr);
                p[e] = pe; m[e] = me; s[e] = se;
            }
            p4[i] = p; m4[i] = m; s4[i] = s;
        }
        for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) upd(ck.p[i], ck.g[i], ck.m[i], ck.s[i], nullptr);
    } else {
        for (int i = threadIdx.x; i < n; i += 256) upd(ck.p[i], ck.g[i], ck.m[i], ck.s[i], ams ? ck.smax + i : nullptr);
    }
}

__global__ void adabelief_advance_kernel(hc_adabelief_group* __restrict__ groups, int ngroups) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < ngroups) groups[g].step += 1;
}

__global__ __launch_bounds__(256) void lars_norm_kernel(const hc_mt_chunk* __restrict__ chunks, float* __restrict__ norms) {
    const hc_mt_chunk ck = chunks[blockIdx.x];
    float sp = 0.f, sg = 0.f;
    for (int i = threadIdx.x; i < ck.n; i += 256) {
        const float p = ck.p[i], g = ck.g[i];
        sp += p * p;
        sg += g * g;
    }
    sp = wave_sum(sp);
    sg = wave_sum(sg);
    __shared__ float sh[8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sh[w] = sp; sh[4 + w] = sg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(norms + 2 * ck.tensor, sh[0] + sh[1] + sh[2] + sh[3]);
        atomicAdd(norms + 2 * ck.tensor + 1, sh[4] + sh[5] + sh[6] + sh[7]);
    }
}

__global__ __launch_bounds__(256) void lars_update_kernel(const hc_mt_chunk* __restrict__ chunks,
                                                          const hc_lars_group* __restrict__ groups,
                                                          const float* __restrict__ norms) {
    const hc_mt_chunk ck = chunks[blockIdx.x];
    const hc_lars_group gr = groups[ck.group];
    const float wd = (float)gr.weight_decay, mu = (float)gr.momentum, omd = (float)(1.0 - gr.dampening);
    const float p_norm = sqrtf(norms[2 * ck.tensor]);
    float denom = sqrtf(norms[2 * ck.tensor + 1]);
    if (wd != 0.f) denom = denom + wd * p_norm;  // denom.add_(p_norm, alpha=weight_decay)
    const float local_lr = (p_norm == 0.f || denom == 0.f) ? 1.f : p_norm / denom;
    const float neg = (float)(-(gr.lr * (double)local_lr));
    const bool init = (ck.flags & 1) != 0;
    for (int i = threadIdx.x; i < ck.n; i += 256) {
        float p = ck.p[i], g = ck.g[i];
        if (wd != 0.f) { g = g + wd * p; ck.g[i] = g; }  // d_p.add_(p.data, alpha=weight_decay), in place (Q4)
        float d = g;
        if (mu != 0.f) {
            float buf;
            if (init) buf = g;                              // momentum_buffer = clone(d_p)
            else buf = ck.m[i] * mu + omd * g;              // buf.mul_(momentum).add_(d_p, alpha=1-dampening)
            ck.m[i] = buf;
            d = gr.nesterov ? g + mu * buf : buf;
        }
        ck.p[i] = p + neg * d;
    }
}

// ---------------------------------------------------------------- AdEMAMix (holocron/optim/ademamix.py:158-200)
// chunk fields: m = exp_avg, s = exp_avg_sq, smax = exp_avg_slow
__global__ __launch_bounds__(256) void ademamix_kernel(const hc_mt_chunk* __restrict__ chunks, const hc_adamx_group* __restrict__ groups) {
    const hc_mt_chunk ck = chunks[blockIdx.x];
    const hc_adamx_group gr = groups[ck.group];
    const double bc1 = 1.0 - pow(gr.beta1, (double)gr.step), bc2 = 1.0 - pow(gr.beta2, (double)gr.step);
    const float b1 = (float)gr.beta1, b2 = (float)gr.beta2, b3 = (float)gr.beta3;
    const float o1 = (float)(1.0 - gr.beta1), o2 = (float)(1.0 - gr.beta2), o3 = (float)(1.0 - gr.beta3);
    const float sqrt_bc2 = (float)sqrt(bc2), fbc1 = (float)bc1, eps = (float)gr.eps, alpha = (float)gr.alpha;
    const float neg_lr = (float)(-gr.lr), wd = (float)gr.weight_decay;
    for (int i = threadIdx.x; i < ck.n; i += 256) {
        float p = ck.p[i], g = ck.g[i];
        if (wd != 0.f) g = g + wd * p;
        const float m1 = ck.m[i] * b1 + o1 * g;
        const float nu = ck.s[i] * b2 + o2 * (g * g);
        const float m2 = ck.smax[i] * b3 + o3 * g;
        ck.m[i] = m1; ck.s[i] = nu; ck.smax[i] = m2;
        const float den = sqrtf(nu) / sqrt_bc2 + eps;
        ck.p[i] = p + neg_lr * ((m1 / fbc1 + alpha * m2) / den);
    }
}

// ---------------------------------------------------------------- AdamP (holocron/optim/adamp.py:142-200)
// pass 1: moments (in place) and the four per-tensor sums the projection test needs: p.g, p.p, g.g, p.pt
__device__ __forceinline__ float adamp_pt(float m, float s, float fbc1, float sqrt_bc2, float eps) {
    return (m / fbc1) / (sqrtf(s) / sqrt_bc2 + eps);
}
__global__ __launch_bounds__(256) void adamp_moments_kernel(const hc_mt_chunk* __restrict__ chunks, const hc_adamx_group* __restrict__ groups,
                                                            float* __restrict__ sums) {
    const hc_mt_chunk ck = chunks[blockIdx.x];
    const hc_adamx_group gr = groups[ck.group];
    const double bc1 = 1.0 - pow(gr.beta1, (double)gr.step), bc2 = 1.0 - pow(gr.beta2, (double)gr.step);
    const float b1 = (float)gr.beta1, b2 = (float)gr.beta2, o1 = (float)(1.0 - gr.beta1), o2 = (float)(1.0 - gr.beta2);
    const float sqrt_bc2 = (float)sqrt(bc2), fbc1 = (float)bc1, eps = (float)gr.eps, wd = (float)gr.weight_decay;
    const bool ams = gr.amsgrad != 0 && ck.smax != nullptr;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < ck.n; i += 256) {
        const float p = ck.p[i];
        float g = ck.g[i];
        if (wd != 0.f) g = g + wd * p;
        const float m = ck.m[i] * b1 + o1 * g;
        float s = ck.s[i] * b2 + o2 * (g * g);
        ck.m[i] = m; ck.s[i] = s;
        if (ams) { s = fmaxf(ck.smax[i], s); ck.smax[i] = s; }
        const float pt = adamp_pt(m, s, fbc1, sqrt_bc2, eps);
        acc[0] += p * g; acc[1] += p * p; acc[2] += g * g; acc[3] += p * pt;
    }
    __shared__ float sh[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float v = wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(sums + 4 * ck.tensor + threadIdx.x, sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}
// pass 2: pt recomputed from the updated moments; projection onto the tangent space of p when
// cos(p, g) < delta / sqrt(numel) (the reference's host-side `if`, adamp.py:196, decided on the device)
__global__ __launch_bounds__(256) void adamp_update_kernel(const hc_mt_chunk* __restrict__ chunks, const hc_adamx_group* __restrict__ groups,
                                                           const float* __restrict__ sums, const int* __restrict__ numel) {
    const hc_mt_chunk ck = chunks[blockIdx.x];
    const hc_adamx_group gr = groups[ck.group];
    const double bc1 = 1.0 - pow(gr.beta1, (double)gr.step), bc2 = 1.0 - pow(gr.beta2, (double)gr.step);
    const float sqrt_bc2 = (float)sqrt(bc2), fbc1 = (float)bc1, eps = (float)gr.eps, neg_lr = (float)(-gr.lr);
    const bool ams = gr.amsgrad != 0 && ck.smax != nullptr;
    const float pg = sums[4 * ck.tensor], pp = sums[4 * ck.tensor + 1], gg = sums[4 * ck.tensor + 2], ppt = sums[4 * ck.tensor + 3];
    // F.cosine_similarity(x1, x2, eps=1e-8): x1.x2 / sqrt(max(|x1|^2 |x2|^2, eps^2))
    const float cosv = pg / sqrtf(fmaxf(pp * gg, 1e-16f));
    const bool project = cosv < (float)gr.delta / sqrtf((float)numel[ck.tensor]);
    const float pn = sqrtf(pp) + eps;                 // param.norm().add_(eps)
    const float coef = project ? (ppt / pn) / pn : 0.f;   // pt -= (sum(p/pn * pt)) * p/pn
    for (int i = threadIdx.x; i < ck.n; i += 256) {
        const float p = ck.p[i];
        const float s = ams ? ck.smax[i] : ck.s[i];
        float pt = adamp_pt(ck.m[i], s, fbc1, sqrt_bc2, eps);
        pt = pt - coef * p;
        ck.p[i] = p + neg_lr * pt;
    }
}

}  // namespace

extern "C" {

int hc_adabelief_step(const hc_mt_chunk* chunks, int32_t nchunks, hc_adabelief_group* groups, int32_t ngroups,
                      int32_t advance, hc_stream_t stream) {
    if (chunks == nullptr || groups == nullptr || nchunks < 0 || ngroups < 1) return HC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    // the step counters live on the device so that a step replayed from a captured hipGraph keeps
    // counting without any host write (a host-updated counter would race with queued replays)
    if (advance) hipLaunchKernelGGL(adabelief_advance_kernel, dim3((ngroups + 63) / 64), dim3(64), 0, st, groups, ngroups);
    if (nchunks > 0) hipLaunchKernelGGL(adabelief_kernel, dim3(nchunks), dim3(256), 0, st, chunks, groups);
    return hc_launch_status();
}

int hc_lars_step(const hc_mt_chunk* chunks, int32_t nchunks, const hc_lars_group* groups, float* norms, int32_t ntensors,
                 hc_stream_t stream) {
    if (chunks == nullptr || groups == nullptr || norms == nullptr || nchunks < 0) return HC_ERR_ARG;
    if (nchunks == 0) return HC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hc_zero_async(norms, sizeof(float) * 2 * ntensors, st) != hipSuccess) return HC_ERR_LAUNCH;
    hipLaunchKernelGGL(lars_norm_kernel, dim3(nchunks), dim3(256), 0, st, chunks, norms);
    hipLaunchKernelGGL(lars_update_kernel, dim3(nchunks), dim3(256), 0, st, chunks, groups, norms);
    return hc_launch_status();
}

int hc_ademamix_step(const hc_mt_chunk* chunks, int32_t nchunks, const hc_adamx_group* groups, hc_stream_t stream) {
    if (chunks == nullptr || groups == nullptr || nchunks < 0) return HC_ERR_ARG;
    if (nchunks == 0) return HC_OK;
    hipLaunchKernelGGL(ademamix_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, chunks, groups);
    return hc_launch_status();
}
int hc_adamp_step(const hc_mt_chunk* chunks, int32_t nchunks, const hc_adamx_group* groups, float* sums, const int32_t* numel,
                  int32_t ntensors, hc_stream_t stream) {
    if (chunks == nullptr || groups == nullptr || sums == nullptr || numel == nullptr || nchunks < 0 || ntensors < 0) return HC_ERR_ARG;
    if (nchunks == 0) return HC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hc_zero_async(sums, sizeof(float) * 4 * (size_t)ntensors, st) != hipSuccess) return HC_ERR_LAUNCH;
    hipLaunchKernelGGL(adamp_moments_kernel, dim3(nchunks), dim3(256), 0, st, chunks, groups, sums);
    hipLaunchKernelGGL(adamp_update_kernel, dim3(nchunks), dim3(256), 0, st, chunks, groups, sums, (const int*)numel);
    return hc_launch_status();
}

}  // extern "C"
