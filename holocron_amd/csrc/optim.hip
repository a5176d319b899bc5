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

// block-wide sum of K per-thread partials -> atomicAdd into dst[0..K)
template <int K>
__device__ __forceinline__ void block_sums_to(const float (&acc)[K], float* __restrict__ dst) {
    __shared__ float sh[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float v = wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < K) atomicAdd(dst + threadIdx.x, sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

// ---------------------------------------------------------------- LAMB / RaLars (holocron/optim/lamb.py:84-137, ralars.py:66-140)
// update direction from the (already updated) moments; mode 0: LAMB (no bias correction, lamb.py:121-123), 1: rectified
// adaptive momentum (ralars.py:108-115), 2: adaptive momentum without rectification (:117-121), 3: plain momentum (:123-124)
struct LambCoef {
    float b1, b2, o1, o2, eps, wd, bc1, bc2, rect;
    int mode;
};
__device__ __forceinline__ LambCoef lamb_coef(const hc_lamb_group& gr) {
    LambCoef c;
    c.b1 = (float)gr.beta1; c.b2 = (float)gr.beta2; c.o1 = (float)(1.0 - gr.beta1); c.o2 = (float)(1.0 - gr.beta2);
    c.eps = (float)gr.eps; c.wd = (float)gr.weight_decay; c.rect = (float)gr.rect; c.mode = gr.mode;
    c.bc1 = (float)(1.0 - pow(gr.beta1, (double)gr.step));
    c.bc2 = (float)(1.0 - pow(gr.beta2, (double)gr.step));
    return c;
}
__device__ __forceinline__ float lamb_dir(const LambCoef& c, float p, float m, float v) {
    float u;
    if (c.mode == 0) u = m / (sqrtf(v) + c.eps);
    else if (c.mode == 1) u = c.rect * ((m / c.bc1) / (sqrtf(v / c.bc2) + c.eps));
    else if (c.mode == 2) u = (m / c.bc1) / (sqrtf(v / c.bc2) + c.eps);
    else u = m / c.bc1;
    if (c.wd != 0.f) u = u + c.wd * p;
    return u;
}
// pass 1: moments in place, per-tensor sum p^2 and sum u^2
__global__ __launch_bounds__(256) void lamb_moments_kernel(const hc_mt_chunk* __restrict__ chunks, const hc_lamb_group* __restrict__ groups,
                                                           float* __restrict__ norms) {
    const hc_mt_chunk ck = chunks[blockIdx.x];
    const LambCoef c = lamb_coef(groups[ck.group]);
    float acc[2] = {0.f, 0.f};
    for (int i = threadIdx.x; i < ck.n; i += 256) {
        const float p = ck.p[i], g = ck.g[i];
        const float m = ck.m[i] * c.b1 + c.o1 * g;
        const float v = ck.s[i] * c.b2 + c.o2 * (g * g);
        ck.m[i] = m; ck.s[i] = v;
        const float u = lamb_dir(c, p, m, v);
        acc[0] += p * p; acc[1] += u * u;
    }
    block_sums_to<2>(acc, norms + 2 * ck.tensor);
}
// pass 2: local_lr = 1 if phi(|p|) == 0 or |u| == 0 else phi(|p|) / |u|;  p -= lr * local_lr * u
__global__ __launch_bounds__(256) void lamb_update_kernel(const hc_mt_chunk* __restrict__ chunks, const hc_lamb_group* __restrict__ groups,
                                                          const float* __restrict__ norms, float* __restrict__ local_lr) {
    const hc_mt_chunk ck = chunks[blockIdx.x];
    const hc_lamb_group gr = groups[ck.group];
    const LambCoef c = lamb_coef(gr);
    const float pn = sqrtf(norms[2 * ck.tensor]), un = sqrtf(norms[2 * ck.tensor + 1]);
    const float phi = fminf(fmaxf(pn, (float)gr.clip_lo), (float)gr.clip_hi);
    const float loc = (phi == 0.f || un == 0.f) ? 1.f : phi / un;
    if (threadIdx.x == 0) local_lr[ck.tensor] = loc;      // every chunk of the tensor writes the same value
    const float neg = (float)(-gr.lr) * loc;
    for (int i = threadIdx.x; i < ck.n; i += 256) {
        const float p = ck.p[i];
        ck.p[i] = p + neg * lamb_dir(c, p, ck.m[i], ck.s[i]);
    }
}

// ---------------------------------------------------------------- TAdam (holocron/optim/tadam.py:157-212)
// pass 1: per-tensor sum of (g - m)^2 / (v + eps) over the OLD moments (g includes the weight decay term)
__global__ __launch_bounds__(256) void tadam_sum_kernel(const hc_mt_chunk* __restrict__ chunks, const hc_adamx_group* __restrict__ groups,
                                                        float* __restrict__ sums) {
    const hc_mt_chunk ck = chunks[blockIdx.x];
    const hc_adamx_group gr = groups[ck.group];
    const float eps = (float)gr.eps, wd = (float)gr.weight_decay;
    float acc[1] = {0.f};
    for (int i = threadIdx.x; i < ck.n; i += 256) {
        float g = ck.g[i];
        if (wd != 0.f) g = g + wd * ck.p[i];
        const float d = g - ck.m[i];
        acc[0] += (d * d) / (ck.s[i] + eps);
    }
    block_sums_to<1>(acc, sums + ck.tensor);
}
// per tensor: w_t = (dof + numel) / (sum + dof); wts[t] = (W_t, w_t) for the update pass; W_t <- W_t (2 beta1 - 1) / beta1 + w_t
__global__ void tadam_scalar_kernel(const float* __restrict__ sums, const float* __restrict__ dof, const int* __restrict__ numel,
                                    float* const* __restrict__ W, const int* __restrict__ tgroup,
                                    const hc_adamx_group* __restrict__ groups, float* __restrict__ wts, int ntensors) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntensors) return;
    const float b1 = (float)groups[tgroup[t]].beta1;
    const float w = (dof[t] + (float)numel[t]) / (sums[t] + dof[t]);
    const float Wold = *W[t];
    wts[2 * t] = Wold;
    wts[2 * t + 1] = w;
    *W[t] = Wold * ((2.f * b1 - 1.f) / b1) + w;
}
// pass 2: m = m W/(W+w) + w g/(W+w); v = beta2 v + (1 - beta2) g^2; p -= lr/bc1 * m / (sqrt(v [max])/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void tadam_update_kernel(const hc_mt_chunk* __restrict__ chunks, const hc_adamx_group* __restrict__ groups,
                                                           const float* __restrict__ wts) {
    const hc_mt_chunk ck = chunks[blockIdx.x];
    const hc_adamx_group gr = groups[ck.group];
    const double bc1 = 1.0 - pow(gr.beta1, (double)gr.step), bc2 = 1.0 - pow(gr.beta2, (double)gr.step);
    const float b2 = (float)gr.beta2, o2 = (float)(1.0 - gr.beta2), eps = (float)gr.eps, wd = (float)gr.weight_decay;
    const float sqrt_bc2 = (float)sqrt(bc2), neg_step = (float)(-(gr.lr / bc1));
    const bool ams = gr.amsgrad != 0 && ck.smax != nullptr;
    const float W = wts[2 * ck.tensor], w = wts[2 * ck.tensor + 1];
    const float km = W / (W + w), kg = w / (W + w);
    for (int i = threadIdx.x; i < ck.n; i += 256) {
        const float p = ck.p[i];
        float g = ck.g[i];
        if (wd != 0.f) g = g + wd * p;
        const float m = ck.m[i] * km + kg * g;
        float v = ck.s[i] * b2 + o2 * (g * g);
        ck.m[i] = m; ck.s[i] = v;
        if (ams) { v = fmaxf(ck.smax[i], v); ck.smax[i] = v; }
        ck.p[i] = p + neg_step * (m / (sqrtf(v) / sqrt_bc2 + eps));
    }
}

// ---------------------------------------------------------------- Adan (holocron/optim/adan.py:146-199)
// chunk.m = exp_avg, chunk.s = exp_avg_sq (EMA of the gradient difference), chunk.smax = exp_avg_delta;
// extra[chunk].m = max_exp_avg_delta (amsgrad), extra[chunk].s = prev_grad (read only: the reference never writes it)
__global__ __launch_bounds__(256) void adan_kernel(const hc_mt_chunk* __restrict__ chunks, const hc_mt_chunk* __restrict__ extra,
                                                   const hc_adamx_group* __restrict__ groups) {
    const hc_mt_chunk ck = chunks[blockIdx.x], ex = extra[blockIdx.x];
    const hc_adamx_group gr = groups[ck.group];
    const float bc1 = (float)(1.0 - pow(gr.beta1, (double)gr.step)), bc2 = (float)(1.0 - pow(gr.beta2, (double)gr.step));
    const float sqrt_bc3 = (float)sqrt(1.0 - pow(gr.beta3, (double)gr.step));
    const float b1 = (float)gr.beta1, b2 = (float)gr.beta2, b3 = (float)gr.beta3;
    const float o1 = (float)(1.0 - gr.beta1), o2 = (float)(1.0 - gr.beta2), o3 = (float)(1.0 - gr.beta3);
    const float eps = (float)gr.eps, wd = (float)gr.weight_decay, neg_lr = (float)(-gr.lr);
    const float shrink = (float)(1.0 + gr.weight_decay * gr.lr);
    const bool ams = gr.amsgrad != 0 && ex.m != nullptr;
    for (int i = threadIdx.x; i < ck.n; i += 256) {
        float p = ck.p[i];
        float g = ck.g[i];
        if (wd != 0.f) g = g + wd * p;
        const float m = ck.m[i] * b1 + o1 * g;
        const float dg = g - (ex.s != nullptr ? ex.s[i] : 0.f);
        const float v = ck.s[i] * b2 + o2 * dg;
        const float t = g + b2 * dg;
        float n = ck.smax[i] * b3 + o3 * (t * t);
        ck.m[i] = m; ck.s[i] = v; ck.smax[i] = n;
        if (ams) { n = fmaxf(ex.m[i], n); ex.m[i] = n; }
        const float den = sqrtf(n) / sqrt_bc3 + eps;
        p = p + neg_lr * ((m / bc1 + b2 * v / bc2) / den);
        if (wd != 0.f) p = p / shrink;
        ck.p[i] = p;
    }
}

// ---------------------------------------------------------------- Lookahead / Scout synchronisation (holocron/optim/wrapper.py:121-134)
// chunk.p = fast weights, chunk.m = slow weights: slow += rate (fast - slow) [rate > 0]; fast = slow
__global__ __launch_bounds__(256) void lookahead_sync_kernel(const hc_mt_chunk* __restrict__ chunks, float rate) {
    const hc_mt_chunk ck = chunks[blockIdx.x];
    for (int i = threadIdx.x; i < ck.n; i += 256) {
        float sl = ck.m[i];
        if (rate > 0.f) {
            const float d = ck.p[i] - sl;
            sl = sl + rate * d;
            ck.m[i] = sl;
        }
        ck.p[i] = sl;
    }
}

// ---------------------------------------------------------------- gradient-bucket pack / unpack (holocron_amd/parallel.py)
// dst[i] = scale * src[i] over up to HC_MULTI_COPY_MAX tensor pieces in ONE launch, fp32 <-> fp32 / bf16.  The piece table travels
// BY VALUE in the kernel arguments: nothing has to be uploaded (or kept alive) for a launch captured in a hipGraph, and the eager
// step pays no host-to-device copy per bucket.  Pure streaming: 16 bytes per lane on the fp32 side when both pointers allow it.
template <typename TS, typename TD>
__device__ __forceinline__ void multi_copy_piece(const TS* __restrict__ src, TD* __restrict__ dst, const long n, const float scale) {
    constexpr bool SB = sizeof(TS) == 2, DB = sizeof(TD) == 2;
    auto ld = [&](long i) { if constexpr (SB) return bf16_to_f32(src[i]); else return src[i]; };
    auto st = [&](long i, float v) { if constexpr (DB) dst[i] = f32_to_bf16(v); else dst[i] = v; };
    const long t = (long)blockIdx.x * 256 + threadIdx.x, nt = (long)gridDim.x * 256;
    const bool vec = (reinterpret_cast<uintptr_t>(src) % (4 * sizeof(TS)) == 0) && (reinterpret_cast<uintptr_t>(dst) % (4 * sizeof(TD)) == 0);
    long done = 0;
    if (vec) {
        const long n4 = n >> 2;
        for (long q = t; q < n4; q += nt) {
            float v[4];
            if constexpr (SB) {
                const u32x2 w = reinterpret_cast<const u32x2*>(src)[q];
                v[0] = bf16lo(w[0]); v[1] = bf16hi(w[0]); v[2] = bf16lo(w[1]); v[3] = bf16hi(w[1]);
            } else {
                const f32x4 w = reinterpret_cast<const f32x4*>(src)[q];
                v[0] = w[0]; v[1] = w[1]; v[2] = w[2]; v[3] = w[3];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= scale;
            if constexpr (DB) {
                u32x2 w;
                w[0] = pack_bf16x2(v[0], v[1]); w[1] = pack_bf16x2(v[2], v[3]);
                reinterpret_cast<u32x2*>(dst)[q] = w;
            } else {
                const f32x4 w = {v[0], v[1], v[2], v[3]};
                reinterpret_cast<f32x4*>(dst)[q] = w;
            }
        }
        done = n4 << 2;
    }
    for (long i = done + t; i < n; i += nt) st(i, ld(i) * scale);
}

__global__ __launch_bounds__(256) void multi_copy_kernel(const hc_multi_copy_desc d) {
    const int it = blockIdx.y;
    const long n = d.n[it];
    if (d.src_bf16) {
        if (d.dst_bf16) multi_copy_piece(static_cast<const bf16_t*>(d.src[it]), static_cast<bf16_t*>(d.dst[it]), n, d.scale);
        else multi_copy_piece(static_cast<const bf16_t*>(d.src[it]), static_cast<float*>(d.dst[it]), n, d.scale);
    } else {
        if (d.dst_bf16) multi_copy_piece(static_cast<const float*>(d.src[it]), static_cast<bf16_t*>(d.dst[it]), n, d.scale);
        else multi_copy_piece(static_cast<const float*>(d.src[it]), static_cast<float*>(d.dst[it]), n, d.scale);
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

int hc_multi_copy(const hc_multi_copy_desc* d, hc_stream_t stream) {
    if (d == nullptr || d->nitems < 0 || d->nitems > HC_MULTI_COPY_MAX) return HC_ERR_ARG;
    if (d->nitems == 0) return HC_OK;
    long nmax = 0;
    for (int i = 0; i < d->nitems; ++i) {
        if (d->n[i] < 0 || (d->n[i] > 0 && (d->src[i] == nullptr || d->dst[i] == nullptr))) return HC_ERR_ARG;
        nmax = d->n[i] > nmax ? d->n[i] : nmax;
    }
    if (nmax == 0) return HC_OK;
    // pieces are at most a few hundred thousand elements (the host splits larger tensors): 16 workgroups sweep the largest one in
    // a handful of rounds, and a launch of 64 such pieces is 1024 workgroups - four per CU
    long gx = (nmax + 256 * 16 - 1) / (256 * 16);
    if (gx > 16) gx = 16;
    hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)gx, (unsigned)d->nitems), dim3(256), 0, (hipStream_t)stream, *d);
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

int hc_lamb_step(const hc_mt_chunk* chunks, int32_t nchunks, const hc_lamb_group* groups, float* norms, float* local_lr,
                 int32_t ntensors, hc_stream_t stream) {
    if (chunks == nullptr || groups == nullptr || norms == nullptr || local_lr == nullptr || nchunks < 0 || ntensors < 0) return HC_ERR_ARG;
    if (nchunks == 0) return HC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hc_zero_async(norms, sizeof(float) * 2 * (size_t)ntensors, st) != hipSuccess) return HC_ERR_LAUNCH;
    hipLaunchKernelGGL(lamb_moments_kernel, dim3(nchunks), dim3(256), 0, st, chunks, groups, norms);
    hipLaunchKernelGGL(lamb_update_kernel, dim3(nchunks), dim3(256), 0, st, chunks, groups, (const float*)norms, local_lr);
    return hc_launch_status();
}
int hc_tadam_step(const hc_mt_chunk* chunks, int32_t nchunks, const hc_adamx_group* groups, float* scratch, const float* dof,
                  const int32_t* numel, const int32_t* tensor_group, float* const* W_t, int32_t ntensors, hc_stream_t stream) {
    if (chunks == nullptr || groups == nullptr || scratch == nullptr || dof == nullptr || numel == nullptr || tensor_group == nullptr ||
        W_t == nullptr || nchunks < 0 || ntensors < 0)
        return HC_ERR_ARG;
    if (nchunks == 0) return HC_OK;
    hipStream_t st = (hipStream_t)stream;
    float* sums = scratch;                       // [ntensors]
    float* wts = scratch + ntensors;             // [ntensors][2]
    if (hc_zero_async(sums, sizeof(float) * (size_t)ntensors, st) != hipSuccess) return HC_ERR_LAUNCH;
    hipLaunchKernelGGL(tadam_sum_kernel, dim3(nchunks), dim3(256), 0, st, chunks, groups, sums);
    hipLaunchKernelGGL(tadam_scalar_kernel, dim3((ntensors + 63) / 64), dim3(64), 0, st, (const float*)sums, dof, (const int*)numel, W_t,
                       (const int*)tensor_group, groups, wts, ntensors);
    hipLaunchKernelGGL(tadam_update_kernel, dim3(nchunks), dim3(256), 0, st, chunks, groups, (const float*)wts);
    return hc_launch_status();
}
int hc_adan_step(const hc_mt_chunk* chunks, const hc_mt_chunk* extra, int32_t nchunks, const hc_adamx_group* groups, hc_stream_t stream) {
    if (chunks == nullptr || extra == nullptr || groups == nullptr || nchunks < 0) return HC_ERR_ARG;
    if (nchunks == 0) return HC_OK;
    hipLaunchKernelGGL(adan_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, chunks, extra, groups);
    return hc_launch_status();
}
int hc_lookahead_sync(const hc_mt_chunk* chunks, int32_t nchunks, float sync_rate, hc_stream_t stream) {
    if (chunks == nullptr || nchunks < 0 || !(sync_rate >= 0.f && sync_rate <= 1.f)) return HC_ERR_ARG;
    if (nchunks == 0) return HC_OK;
    hipLaunchKernelGGL(lookahead_sync_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, chunks, sync_rate);
    return hc_launch_status();
}

}  // extern "C"
