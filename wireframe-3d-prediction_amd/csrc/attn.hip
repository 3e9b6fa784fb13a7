// Multi-head self-attention over the (ragged) vertex sets of the edge head.
//
// Reference: nn.MultiheadAttention(embed 512, 8 heads, batch_first) called per
// sample with batch 1 (EdgePredictor.py:41-46,109-111; torch
// multi_head_attention_forward: q scaled by 1/sqrt(hd) before QK^T, softmax,
// dropout on the weights, PV).  Here all samples run in ONE launch: workgroup
// (head, sample) stages that head's K and V rows of the sample in LDS
// (up to 256 rows x 64 dims at a time: 2 x 68 KB of the CU's 160 KB; longer samples in chunks) and each wave64 owns
// query rows.  Rows are padded to hd+4 floats so that the ds_read_b128 of 16
// different key rows land on 16 distinct 16-B slots.
//
// Vertex rows are COMPACT: sample s owns rows voff[s] .. voff[s+1]-1 of every
// [Rv, *] matrix, so ragged batches cost nothing and no key mask is needed.
#include <stdlib.h>

#include "wf3d_common.h"

namespace {

constexpr int ATT_WAVES = 4;

// head_dim 64 runs on the matrix cores (attn_mfma.hip); WF3D_ATTN_MFMA=0 forces the VALU kernels below
bool use_mfma(int E, int heads) {
    static const int on = [] { const char* e = getenv("WF3D_ATTN_MFMA"); return e ? atoi(e) : 1; }();
    return on && E / heads == 64;
}

struct AttnParams {
    const float* qkv;      // [Rv, 3E]
    float* ctx;            // fwd out [Rv, E]
    float* lse;            // [Rv, heads]
    const float* dctx;     // bwd in  [Rv, E]
    float* dqkv;           // bwd out [Rv, 3E]
    const int32_t* voff;   // [S+1]
    int E, heads, hd, vmax, ch;          // ch = rows of K / V (or Q / dctx) staged at a time = min(vmax, 256)
    float scale;
    uint32_t seed, thresh; float dscale;
};

__device__ __forceinline__ float dot_lds(const float* a, const float* b, int hd) {
    float s = 0.f;
    for (int d = 0; d < hd; d += 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + d);
        const f32x4 y = *reinterpret_cast<const f32x4*>(b + d);
        s += x[0] * y[0]; s += x[1] * y[1]; s += x[2] * y[2]; s += x[3] * y[3];
    }
    return s;
}

// stage rows [r0, r0+n) of column block `col0..col0+hd` of src[*, ld] into dst[n][hd+4]
__device__ __forceinline__ void stage_rows(float* dst, const float* __restrict__ src, int ld, int r0, int n, int col0,
                                           int hd, float mul) {
    const int per = hd / 4;
    for (int idx = threadIdx.x; idx < n * per; idx += blockDim.x) {
        const int r = idx / per, c = (idx % per) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)(r0 + r) * ld + col0 + c);
        v *= mul;
        *reinterpret_cast<f32x4*>(dst + r * (hd + 4) + c) = v;
    }
}

// More than p.ch (= min(vmax, 256)) vertices in a sample: the keys are walked in chunks of p.ch rows staged in LDS, each
// chunk's own softmax (its maximum, its sum, its dropout-masked normalised weights, its partial output) is merged into
// the running (lse, ctx) rows in global memory:  lse' = log(e^lse + e^lse_c),  ctx' = e^(lse - lse') ctx + e^(lse_c - lse') ctx_c.
// With one chunk (every BASELINE config: vmax <= 256) nothing is merged and the arithmetic is the single-pass one.
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int h = blockIdx.x, s = blockIdx.y;
    const int r0 = p.voff[s], n = p.voff[s + 1] - r0;
    if (n <= 0) return;
    const int hd = p.hd, ldr = hd + 4, E3 = 3 * p.E;
    float* Ks = sm;
    float* Vs = Ks + p.ch * ldr;
    float* qb = Vs + p.ch * ldr;                  // [waves][hd]
    float* pb = qb + ATT_WAVES * hd;              // [waves][ch]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* q = qb + wave * hd;
    float* pw = pb + wave * p.ch;
    const int iters = (n + ATT_WAVES - 1) / ATT_WAVES;
    for (int kc = 0; kc < n; kc += p.ch) {
        const int cn = min(p.ch, n - kc);
        __syncthreads();
        stage_rows(Ks, p.qkv, E3, r0 + kc, cn, p.E + h * hd, hd, 1.0f);
        stage_rows(Vs, p.qkv, E3, r0 + kc, cn, 2 * p.E + h * hd, hd, 1.0f);
        __syncthreads();
        for (int it = 0; it < iters; ++it) {
            const int i = it * ATT_WAVES + wave;
            const bool act = i < n;
            if (act)
                for (int d = lane; d < hd; d += 64) q[d] = p.qkv[(size_t)(r0 + i) * E3 + h * hd + d] * p.scale;
            __syncthreads();
            float sc[4], m = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = lane + 64 * t;
                sc[t] = (act && j < cn) ? dot_lds(q, Ks + j * ldr, hd) : -INFINITY;
                m = fmaxf(m, sc[t]);
            }
            m = wf3d_wave_max(m);
            float l = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                sc[t] = (act && lane + 64 * t < cn) ? expf(sc[t] - m) : 0.f;
                l += sc[t];
            }
            l = wf3d_wave_sum(l);
            const float inv = act ? 1.0f / l : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = lane + 64 * t;
                if (j < p.ch) {
                    float pv = sc[t] * inv;
                    if (p.thresh && act && j < cn)
                        pv = wf3d_keep(p.seed, (uint32_t)((r0 + i) * p.heads + h), (uint32_t)(kc + j), p.thresh) ? pv * p.dscale : 0.f;
                    pw[j] = pv;
                }
            }
            const float lse_c = m + logf(l);
            float wa = 0.f, wb = 1.0f, lse_new = lse_c;              // weights of the running row / of this chunk
            if (act && kc > 0) {
                const float lse_o = p.lse[(size_t)(r0 + i) * p.heads + h];
                const float mx = fmaxf(lse_o, lse_c);
                lse_new = mx + logf(expf(lse_o - mx) + expf(lse_c - mx));
                wa = expf(lse_o - lse_new); wb = expf(lse_c - lse_new);
            }
            __syncthreads();
            if (act) {
                if (lane == 0) p.lse[(size_t)(r0 + i) * p.heads + h] = lse_new;
                for (int d = lane; d < hd; d += 64) {
                    float o = 0.f;
                    for (int j = 0; j < cn; ++j) o += pw[j] * Vs[j * ldr + d];
                    float* dst = p.ctx + (size_t)(r0 + i) * p.E + h * hd + d;
                    *dst = kc > 0 ? wa * *dst + wb * o : o;
                }
            }
            __syncthreads();
        }
    }
}

// Backward, two sweeps with the same LDS footprint as forward:
//   sweep 1 (K, V chunks staged; wave per query i):  P_i = exp(s_i - lse_i), dP_i = dctx_i·V^T,
//            delta_i = dctx_i·ctx_i (= sum_j P~_ij dP~_ij: the flash form, valid with dropout and across chunks),
//            dS_i = P_i∘(dP_i - delta_i),  dQ_i = scale * dS_i·K
//   sweep 2 (scaled Q, dctx chunks staged; wave per key j): recompute P_:j, dS_:j over queries,
//            dK_j = dS_:j^T·(scale Q),  dV_j = P~_:j^T·dctx          
// No atomics: every dq/dk/dv element is produced by exactly one wave; with more than p.ch vertices the partial sums of
// the chunks are accumulated by that wave in global memory, in chunk order.
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int h = blockIdx.x, s = blockIdx.y;
    const int r0 = p.voff[s], n = p.voff[s + 1] - r0;
    if (n <= 0) return;
    const int hd = p.hd, ldr = hd + 4, E3 = 3 * p.E;
    float* Xa = sm;                               // sweep 1: K      sweep 2: scaled Q
    float* Xb = Xa + p.ch * ldr;                  // sweep 1: V      sweep 2: dctx
    float* qb = Xb + p.ch * ldr;                  // [waves][hd]
    float* ob = qb + ATT_WAVES * hd;              // [waves][hd]
    float* pb = ob + ATT_WAVES * hd;              // [waves][ch]  dS
    float* pb2 = pb + ATT_WAVES * p.ch;           // [waves][ch]  P~ (sweep 2)
    float* dl = pb2 + ATT_WAVES * p.ch;           // [ch] delta_i of the staged queries (sweep 2)
    float* ls = dl + p.ch;                        // [ch] lse_i of the staged queries (sweep 2)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int iters = (n + ATT_WAVES - 1) / ATT_WAVES;
    float* q = qb + wave * hd;
    float* o = ob + wave * hd;
    float* pw = pb + wave * p.ch;
    float* pw2 = pb2 + wave * p.ch;

    for (int kc = 0; kc < n; kc += p.ch) {
        const int cn = min(p.ch, n - kc);
        __syncthreads();
        stage_rows(Xa, p.qkv, E3, r0 + kc, cn, p.E + h * hd, hd, 1.0f);
        stage_rows(Xb, p.qkv, E3, r0 + kc, cn, 2 * p.E + h * hd, hd, 1.0f);
        __syncthreads();
        for (int it = 0; it < iters; ++it) {
            const int i = it * ATT_WAVES + wave;
            const bool act = i < n;
            float dsum = 0.f;
            if (act)
                for (int d = lane; d < hd; d += 64) {
                    q[d] = p.qkv[(size_t)(r0 + i) * E3 + h * hd + d] * p.scale;
                    const float g = p.dctx[(size_t)(r0 + i) * p.E + h * hd + d];
                    o[d] = g;
                    dsum += g * p.ctx[(size_t)(r0 + i) * p.E + h * hd + d];
                }
            dsum = wf3d_wave_sum(dsum);
            __syncthreads();
            const float lse_i = act ? p.lse[(size_t)(r0 + i) * p.heads + h] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = lane + 64 * t;
                float ds = 0.f;
                if (act && j < cn) {
                    const float pr = expf(dot_lds(q, Xa + j * ldr, hd) - lse_i);
                    float g = dot_lds(o, Xb + j * ldr, hd);
                    if (p.thresh)
                        g = wf3d_keep(p.seed, (uint32_t)((r0 + i) * p.heads + h), (uint32_t)(kc + j), p.thresh) ? g * p.dscale : 0.f;
                    ds = pr * (g - dsum);
                }
                if (j < p.ch) pw[j] = ds;
            }
            __syncthreads();
            if (act) {
                for (int d = lane; d < hd; d += 64) {
                    float a = 0.f;
                    for (int j = 0; j < cn; ++j) a += pw[j] * Xa[j * ldr + d];
                    float* dst = p.dqkv + (size_t)(r0 + i) * E3 + h * hd + d;
                    *dst = kc > 0 ? *dst + a * p.scale : a * p.scale;
                }
            }
            __syncthreads();
        }
    }
    // sweep 2
    for (int qc = 0; qc < n; qc += p.ch) {
        const int cn = min(p.ch, n - qc);
        __syncthreads();
        stage_rows(Xa, p.qkv, E3, r0 + qc, cn, h * hd, hd, p.scale);
        stage_rows(Xb, p.dctx, p.E, r0 + qc, cn, h * hd, hd, 1.0f);
        for (int i = wave; i < cn; i += ATT_WAVES) {          // delta and lse of the staged queries
            float dsum = 0.f;
            for (int d = lane; d < hd; d += 64)
                dsum += p.dctx[(size_t)(r0 + qc + i) * p.E + h * hd + d] * p.ctx[(size_t)(r0 + qc + i) * p.E + h * hd + d];
            dsum = wf3d_wave_sum(dsum);
            if (lane == 0) { dl[i] = dsum; ls[i] = p.lse[(size_t)(r0 + qc + i) * p.heads + h]; }
        }
        __syncthreads();
        for (int it = 0; it < iters; ++it) {
            const int j = it * ATT_WAVES + wave;
            const bool act = j < n;
            if (act)
                for (int d = lane; d < hd; d += 64) {
                    q[d] = p.qkv[(size_t)(r0 + j) * E3 + p.E + h * hd + d];         // k_j
                    o[d] = p.qkv[(size_t)(r0 + j) * E3 + 2 * p.E + h * hd + d];     // v_j
                }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int i = lane + 64 * t;
                float ds = 0.f, pt = 0.f;
                if (act && i < cn) {
                    const float pr = expf(dot_lds(Xa + i * ldr, q, hd) - ls[i]);
                    float g = dot_lds(Xb + i * ldr, o, hd);
                    float mk = 1.0f;
                    if (p.thresh)
                        mk = wf3d_keep(p.seed, (uint32_t)((r0 + qc + i) * p.heads + h), (uint32_t)j, p.thresh) ? p.dscale : 0.f;
                    ds = pr * (g * mk - dl[i]);
                    pt = pr * mk;
                }
                if (i < p.ch) { pw[i] = ds; pw2[i] = pt; }
            }
            __syncthreads();
            if (act) {
                for (int d = lane; d < hd; d += 64) {
                    float ak = 0.f, av = 0.f;
                    for (int i = 0; i < cn; ++i) {
                        ak += pw[i] * Xa[i * ldr + d];
                        av += pw2[i] * Xb[i * ldr + d];
                    }
                    float* dk = p.dqkv + (size_t)(r0 + j) * E3 + p.E + h * hd + d;
                    float* dv = p.dqkv + (size_t)(r0 + j) * E3 + 2 * p.E + h * hd + d;
                    *dk = qc > 0 ? *dk + ak : ak;
                    *dv = qc > 0 ? *dv + av : av;
                }
            }
            __syncthreads();
        }
    }
}

int attn_chunk(int vmax) { return vmax < 256 ? vmax : 256; }

size_t attn_lds_bytes(int vmax, int hd, bool bwd) {
    const size_t ch = (size_t)attn_chunk(vmax);
    size_t f = 2 * ch * (hd + 4) + (size_t)ATT_WAVES * hd + (size_t)ATT_WAVES * ch;
    if (bwd) f += (size_t)ATT_WAVES * hd + (size_t)ATT_WAVES * ch + 2 * ch;
    return f * sizeof(float);
}

int attn_check(const char* who, int S, int E, int heads, int vmax, bool bwd) {
    WF3D_CHECK(S >= 0 && E > 0 && heads > 0 && E % heads == 0, WF3D_ERR_ARG, "%s: bad dims", who);
    const int hd = E / heads;
    WF3D_CHECK(hd % 4 == 0, WF3D_ERR_UNSUPPORTED, "%s: head_dim %d must be a multiple of 4", who, hd);
    WF3D_CHECK(vmax >= 0, WF3D_ERR_ARG, "%s: bad vmax %d", who, vmax);
    WF3D_CHECK(attn_lds_bytes(vmax, hd, bwd) <= 160 * 1024, WF3D_ERR_UNSUPPORTED,
               "%s: 256-row chunks of head_dim %d do not fit the 160 KiB LDS", who, hd);
    WF3D_CHECK(S <= 65535, WF3D_ERR_UNSUPPORTED, "%s: more than 65535 samples", who);
    return WF3D_OK;
}

}  // namespace

extern "C" int wf3d_attn_fwd(const float* qkv, const int32_t* voff, int S, int vmax, int E, int heads, float drop_p,
                             uint32_t drop_seed, float* ctx, float* lse, void* stream) {
    int rc = attn_check("wf3d_attn_fwd", S, E, heads, vmax, false);
    if (rc) return rc;
    if (S == 0 || vmax == 0) return WF3D_OK;
    WF3D_CHECK(qkv && voff && ctx && lse, WF3D_ERR_ARG, "wf3d_attn_fwd: null pointer");
    WF3D_CHECK(drop_p >= 0.f && drop_p < 1.f, WF3D_ERR_ARG, "wf3d_attn_fwd: bad drop_p");
    if (use_mfma(E, heads) && vmax <= 256) return wf3d_attn_fwd_mfma(qkv, voff, S, vmax, E, heads, drop_p, drop_seed, ctx, lse, stream);
    AttnParams p{};
    p.qkv = qkv; p.ctx = ctx; p.lse = lse; p.voff = voff;
    p.E = E; p.heads = heads; p.hd = E / heads; p.vmax = vmax; p.ch = attn_chunk(vmax);
    p.scale = 1.0f / sqrtf((float)p.hd);
    if (drop_p > 0.f) { p.seed = drop_seed; p.thresh = (uint32_t)((double)drop_p * 4294967296.0); p.dscale = 1.0f / (1.0f - drop_p); }
    const size_t lds = attn_lds_bytes(vmax, p.hd, false);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        WF3D_CHECK(e == hipSuccess, WF3D_ERR_LAUNCH, "wf3d_attn_fwd: cannot raise dynamic LDS to %zu", lds);
    }
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(heads, S), dim3(256), lds, (hipStream_t)stream, p);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_attn_bwd(const float* qkv, const float* dctx, const float* ctx, const float* lse,
                             const int32_t* voff, int S, int vmax, int E, int heads, float drop_p, uint32_t drop_seed,
                             float* dqkv, void* stream) {
    int rc = attn_check("wf3d_attn_bwd", S, E, heads, vmax, true);
    if (rc) return rc;
    if (S == 0 || vmax == 0) return WF3D_OK;
    WF3D_CHECK(qkv && dctx && ctx && lse && voff && dqkv, WF3D_ERR_ARG, "wf3d_attn_bwd: null pointer");
    WF3D_CHECK(drop_p >= 0.f && drop_p < 1.f, WF3D_ERR_ARG, "wf3d_attn_bwd: bad drop_p");
    if (use_mfma(E, heads) && vmax <= 256)
        return wf3d_attn_bwd_mfma(qkv, dctx, ctx, lse, voff, S, vmax, E, heads, drop_p, drop_seed, dqkv, stream);
    AttnParams p{};
    p.qkv = qkv; p.dctx = dctx; p.ctx = (float*)ctx; p.lse = (float*)lse; p.dqkv = dqkv; p.voff = voff;
    p.E = E; p.heads = heads; p.hd = E / heads; p.vmax = vmax; p.ch = attn_chunk(vmax);
    p.scale = 1.0f / sqrtf((float)p.hd);
    if (drop_p > 0.f) { p.seed = drop_seed; p.thresh = (uint32_t)((double)drop_p * 4294967296.0); p.dscale = 1.0f / (1.0f - drop_p); }
    const size_t lds = attn_lds_bytes(vmax, p.hd, true);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        WF3D_CHECK(e == hipSuccess, WF3D_ERR_LAUNCH, "wf3d_attn_bwd: cannot raise dynamic LDS to %zu", lds);
    }
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(heads, S), dim3(256), lds, (hipStream_t)stream, p);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}
