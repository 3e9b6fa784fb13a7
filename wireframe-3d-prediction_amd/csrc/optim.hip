// Row f-2: the step tail of the reference's training loop (train.py:96,141-142) —
//     torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)   then   Adam(lr, weight_decay).step()
// as TWO launches over all parameter tensors at once instead of ~6 foreach kernels x 80 tensors:
//
//   gradsq_kernel   per 16 K-element chunk of every gradient: sum of squares -> partials[chunk]
//   clip_adam_kernel  every workgroup folds the partials (a few thousand floats, L2-resident) into the global norm and the
//                   clip coefficient min(1, max_norm / (norm + 1e-6)) — no third launch, no host read — then applies to
//                   its chunk:  g *= coef (in place, like clip_grad_norm_);  g += wd * p;  m = lerp(m, g, 1 - b1);
//                   v = b2 * v + (1 - b2) * g * g;  p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps)      (torch's Adam, :142)
//                   Tensors with p == NULL are clip-only: the reference's lazily created point_pool_proj is in
//                   model.parameters() (so in the norm and scaled) but not in the optimizer.
//
// HBM-bound: reads p, g, m, v and writes p, g, m, v once: 32 B per parameter (0.99 GB for the 31 M parameters).
// Tensor tables travel as kernel arguments (<= 80 tensors per launch), so per-step gradient reallocation costs nothing.
#include <math.h>

#include "wf3d_common.h"

namespace {

constexpr int OPT_CHUNK = 16384;          // elements per workgroup
constexpr int OPT_MAXT = 80;              // tensors per launch (kernel-argument budget: 80 x 40 B + 81 x 4 B < 4 KB)

struct SqArgs {
    const float* g[OPT_MAXT];
    long n[OPT_MAXT];
    int blk_begin[OPT_MAXT + 1];
    int ntensors;
};

struct AdamArgs {
    float* p[OPT_MAXT];
    float* g[OPT_MAXT];
    float* m[OPT_MAXT];
    float* v[OPT_MAXT];
    long n[OPT_MAXT];
    int blk_begin[OPT_MAXT + 1];
    int ntensors;
};

template <typename A>
__device__ __forceinline__ int find_tensor(const A& a, int blk) {
    int lo = 0, hi = a.ntensors - 1;                 // last t with blk_begin[t] <= blk
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.blk_begin[mid] <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ float block_sum(float v, float* s_red) {
    v = wf3d_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w];       // every thread, same order
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void gradsq_kernel(const SqArgs a, float* __restrict__ partials, int part_off) {
    __shared__ float s_red[4];
    const int t = find_tensor(a, blockIdx.x);
    const long base = (long)(blockIdx.x - a.blk_begin[t]) * OPT_CHUNK;
    const long n = a.n[t];
    const float* g = a.g[t];
    const long end = base + OPT_CHUNK < n ? base + OPT_CHUNK : n;
    float s = 0.f;
    if ((((uintptr_t)g) & 15) == 0) {
        for (long i = base + threadIdx.x * 4; i < end; i += 1024) {
            if (i + 4 <= end) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(g + i);
                s += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
            } else {
                for (long j = i; j < end; ++j) s += g[j] * g[j];
            }
        }
    } else {
        for (long i = base + threadIdx.x; i < end; i += 256) s += g[i] * g[i];
    }
    s = block_sum(s, s_red);
    if (threadIdx.x == 0) partials[part_off + blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void clip_adam_kernel(const AdamArgs a, const float* __restrict__ partials, int npart,
                                                         float max_norm, float lr, float omb1, float beta2, float omb2,
                                                         float eps, float wd, float bc1, float bc2_sqrt,
                                                         float* __restrict__ norm_out) {
    __shared__ float s_red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < npart; i += 256) s += partials[i];
    const float norm = sqrtf(block_sum(s, s_red));
    float coef = max_norm / (norm + 1e-6f);                      // clip_grad_norm_: clamp(max_norm / (total_norm + 1e-6), max=1)
    // max_norm <= 0: no clipping.  A NaN norm must poison every gradient as clip_grad_norm_ does (torch.clamp keeps the
    // NaN): fminf(NaN, 1) would return 1 and let a step with a non-finite gradient through unscaled.
    coef = max_norm > 0.f ? (norm != norm ? norm : fminf(coef, 1.0f)) : 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = norm;
    const int t = find_tensor(a, blockIdx.x);
    const long base = (long)(blockIdx.x - a.blk_begin[t]) * OPT_CHUNK;
    const long n = a.n[t];
    const long end = base + OPT_CHUNK < n ? base + OPT_CHUNK : n;
    float* p = a.p[t];
    float* g = a.g[t];
    float* m = a.m[t];
    float* v = a.v[t];
    const float step = lr / bc1;      // omb1 = 1 - beta1, omb2 = 1 - beta2 come rounded from double (1.0f - 0.999f is off by 5e-5)
    auto upd = [&](float& pp, float& gg, float& mm, float& vv) {
        gg *= coef;                                   // the clipped gradient is what stays in .grad
        const float ge = fmaf(wd, pp, gg);            // L2 weight decay folded into the gradient (torch Adam)
        mm = fmaf(omb1, ge - mm, mm);                 // exp_avg.lerp_(grad, 1 - beta1)
        vv = fmaf(omb2 * ge, ge, beta2 * vv);         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        pp -= step * (mm / (sqrtf(vv) / bc2_sqrt + eps));
    };
    const bool vec = (((uintptr_t)g | (uintptr_t)p | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    if (p == nullptr) {                               // clip-only tensor
        for (long i = base + threadIdx.x; i < end; i += 256) g[i] *= coef;
        return;
    }
    if (vec) {
        for (long i = base + threadIdx.x * 4; i < end; i += 1024) {
            if (i + 4 <= end) {
                f32x4 pp = *reinterpret_cast<f32x4*>(p + i), gg = *reinterpret_cast<f32x4*>(g + i);
                f32x4 mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a0 = pp[e], a1 = gg[e], a2 = mm[e], a3 = vv[e];
                    upd(a0, a1, a2, a3);
                    pp[e] = a0; gg[e] = a1; mm[e] = a2; vv[e] = a3;
                }
                *reinterpret_cast<f32x4*>(p + i) = pp; *reinterpret_cast<f32x4*>(g + i) = gg;
                *reinterpret_cast<f32x4*>(m + i) = mm; *reinterpret_cast<f32x4*>(v + i) = vv;
            } else {
                for (long j = i; j < end; ++j) upd(p[j], g[j], m[j], v[j]);
            }
        }
    } else {
        for (long i = base + threadIdx.x; i < end; i += 256) upd(p[i], g[i], m[i], v[i]);
    }
}

}  // namespace

extern "C" size_t wf3d_clip_adam_ws_floats(const long* numel, int ntensors) {
    size_t blocks = 0;
    for (int i = 0; i < ntensors; ++i) blocks += (size_t)wf3d_cdiv(numel[i] > 0 ? numel[i] : 1, OPT_CHUNK);
    return blocks + 1;
}

extern "C" int wf3d_clip_adam_step(float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                   const long* numel, int ntensors, double max_norm, double lr, double beta1, double beta2,
                                   double eps, double weight_decay, int step, float* ws, size_t ws_floats, float* total_norm,
                                   void* stream) {
    WF3D_CHECK(ntensors >= 0 && params && grads && exp_avg && exp_avg_sq && numel, WF3D_ERR_ARG, "wf3d_clip_adam_step: null table");
    if (ntensors == 0) return WF3D_OK;
    WF3D_CHECK(step >= 1 && lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1., WF3D_ERR_ARG, "wf3d_clip_adam_step: bad hyper-parameters");
    WF3D_CHECK(ws && ws_floats >= wf3d_clip_adam_ws_floats(numel, ntensors), WF3D_ERR_WS, "wf3d_clip_adam_step: workspace too small");
    for (int i = 0; i < ntensors; ++i) {
        WF3D_CHECK(grads[i] && numel[i] > 0, WF3D_ERR_ARG, "wf3d_clip_adam_step: tensor %d has no gradient", i);
        WF3D_CHECK(!params[i] || (exp_avg[i] && exp_avg_sq[i]), WF3D_ERR_ARG, "wf3d_clip_adam_step: tensor %d lacks optimizer state", i);
    }
    hipStream_t st = (hipStream_t)stream;
    int npart = 0;
    for (int t0 = 0; t0 < ntensors; t0 += OPT_MAXT) {
        const int nt = ntensors - t0 < OPT_MAXT ? ntensors - t0 : OPT_MAXT;
        SqArgs a;
        a.ntensors = nt;
        int blk = 0;
        for (int i = 0; i < nt; ++i) {
            a.g[i] = grads[t0 + i]; a.n[i] = numel[t0 + i]; a.blk_begin[i] = blk;
            blk += wf3d_cdiv(numel[t0 + i], OPT_CHUNK);
        }
        for (int i = nt; i <= OPT_MAXT; ++i) a.blk_begin[i] = blk;
        hipLaunchKernelGGL(gradsq_kernel, dim3(blk), dim3(256), 0, st, a, ws, npart);
        WF3D_LAUNCH_CHECK();
        npart += blk;
    }
    // hyper-parameters arrive as doubles (Python floats) and every derived scalar is formed in double before the one
    // rounding to fp32, as torch does on the host: 1 - 0.999 must become 0.001f, not 1.0f - 0.999f
    const float bc1 = (float)(1.0 - pow(beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    for (int t0 = 0; t0 < ntensors; t0 += OPT_MAXT) {
        const int nt = ntensors - t0 < OPT_MAXT ? ntensors - t0 : OPT_MAXT;
        AdamArgs a;
        a.ntensors = nt;
        int blk = 0;
        for (int i = 0; i < nt; ++i) {
            a.p[i] = params[t0 + i]; a.g[i] = grads[t0 + i]; a.m[i] = exp_avg[t0 + i]; a.v[i] = exp_avg_sq[t0 + i];
            a.n[i] = numel[t0 + i]; a.blk_begin[i] = blk;
            blk += wf3d_cdiv(numel[t0 + i], OPT_CHUNK);
        }
        for (int i = nt; i <= OPT_MAXT; ++i) a.blk_begin[i] = blk;
        hipLaunchKernelGGL(clip_adam_kernel, dim3(blk), dim3(256), 0, st, a, ws, npart, (float)max_norm, (float)lr, (float)(1.0 - beta1),
                           (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay, bc1, bc2_sqrt, t0 == 0 ? total_norm : nullptr);
        WF3D_LAUNCH_CHECK();
    }
    return WF3D_OK;
}
