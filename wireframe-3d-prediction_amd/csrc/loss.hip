// Row f-1: the device side of losses/WireframeLoss.py — the caller of backward.
//
// The reference builds one [V, V] Hungarian cost matrix per sample with torch ops and a
// `.cpu().numpy()` sync PER SAMPLE (WireframeLoss.py:130-236), then evaluates SmoothL1 / BCE with
// a dozen more launches.  Here: one kernel writes all B cost matrices (one device->host copy for
// the whole batch; the assignment itself stays scipy on the host, as in the reference), and one
// kernel evaluates the three loss terms AND their gradients w.r.t. the model outputs — which are
// exactly the cotangents entering the hot path's backward.
#include "wf3d_common.h"

namespace {

// cost[b,p,t] = sum_k |v[b,p,k] - tv[b,t,k]| + |e[b,p] - 1|   for t < count[b]   (real targets)
//             = e[b,p]                                         for t >= count[b]  (dummy columns)
__global__ __launch_bounds__(256) void loss_cost_kernel(const float* __restrict__ verts, long vs_b, long vs_v,
                                                         const float* __restrict__ exist,
                                                         const float* __restrict__ tverts, int Vt,
                                                         const int64_t* __restrict__ counts, int V,
                                                         float* __restrict__ cost) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= V * V) return;
    const int p = idx / V, t = idx % V;
    const int cnt = (int)counts[b];
    const float e = exist[(size_t)b * V + p];
    float c = e;
    if (t < cnt) {
        const float* v = verts + (size_t)b * vs_b + (size_t)p * vs_v;
        const float* w = tverts + ((size_t)b * Vt + t) * 3;
        c = (fabsf(v[0] - w[0]) + fabsf(v[1] - w[1]) + fabsf(v[2] - w[2])) + fabsf(e - 1.0f);
    }
    cost[(size_t)b * V * V + idx] = c;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wf3d_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// nn.BCELoss element: -(t*max(log p, -100) + (1-t)*max(log(1-p), -100)); grad (p-t)/max(p(1-p), 1e-12)
__device__ __forceinline__ float bce(float p, float t, float& g) {
    g = (p - t) / fmaxf(p * (1.0f - p), 1e-12f);
    return -(t * fmaxf(logf(p), -100.0f) + (1.0f - t) * fmaxf(log1pf(-p), -100.0f));
}

// One workgroup per sample: partial sums of the three terms + all gradients.
__global__ __launch_bounds__(256) void loss_terms_kernel(const float* __restrict__ verts, long vs_b, long vs_v,
                                                          const float* __restrict__ exist,
                                                          const float* __restrict__ edge, int Ep,
                                                          const float* __restrict__ tverts, int Vt,
                                                          const float* __restrict__ texist,
                                                          const float* __restrict__ tlabel, int Et, int min_e,
                                                          const int32_t* __restrict__ m_pred,
                                                          const int32_t* __restrict__ m_tgt,
                                                          const int32_t* __restrict__ m_off, int V, int B,
                                                          float gv, float ge, float gd,      // weight / normaliser
                                                          float* __restrict__ dverts, float* __restrict__ dexist,
                                                          float* __restrict__ dedge, float* __restrict__ part) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    // vertex term: SmoothL1 (beta = 1) over the Hungarian-matched pairs
    for (int i = tid; i < V * 3; i += 256) dverts[(size_t)b * V * 3 + i] = 0.f;
    __syncthreads();
    float sv = 0.f;
    for (int m = m_off[b] + tid; m < m_off[b + 1]; m += 256) {
        const int p = m_pred[m], t = m_tgt[m];
        const float* v = verts + (size_t)b * vs_b + (size_t)p * vs_v;
        const float* w = tverts + ((size_t)b * Vt + t) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = v[k] - w[k], a = fabsf(d);
            sv += a < 1.0f ? 0.5f * d * d : a - 0.5f;
            dverts[((size_t)b * V + p) * 3 + k] = gv * (a < 1.0f ? d : (d > 0.f ? 1.0f : -1.0f));   // p unique per match
        }
    }
    // existence term
    float se = 0.f;
    for (int i = tid; i < V; i += 256) {
        float g;
        se += bce(exist[(size_t)b * V + i], texist[(size_t)b * V + i], g);
        dexist[(size_t)b * V + i] = ge * g;
    }
    // edge term over the common width min(Ep, Et)
    float sd = 0.f;
    for (int i = tid; i < Ep; i += 256) {
        float g = 0.f;
        if (i < min_e) sd += bce(edge[(size_t)b * Ep + i], tlabel[(size_t)b * Et + i], g);
        dedge[(size_t)b * Ep + i] = gd * g;
    }
    sv = block_sum(sv, red); se = block_sum(se, red); sd = block_sum(sd, red);
    if (tid == 0) { part[b * 3] = sv; part[b * 3 + 1] = se; part[b * 3 + 2] = sd; }
}

__global__ void loss_final_kernel(const float* __restrict__ part, int B, float nv, float ne, float nd, float wv, float we,
                                  float wd, float* __restrict__ out) {
    float sv = 0.f, se = 0.f, sd = 0.f;
    for (int b = 0; b < B; ++b) { sv += part[b * 3]; se += part[b * 3 + 1]; sd += part[b * 3 + 2]; }
    const float lv = nv > 0.f ? sv / nv : 0.f, le = ne > 0.f ? se / ne : 0.f, ld = nd > 0.f ? sd / nd : 0.f;
    out[0] = lv; out[1] = le; out[2] = ld; out[3] = wv * lv + we * le + wd * ld;
}

}  // namespace

extern "C" int wf3d_loss_cost_matrix(const float* verts, long sample_stride, long vertex_stride, const float* exist,
                                     const float* tverts, int Vt, const int64_t* counts, int B, int V, float* cost,
                                     void* stream) {
    WF3D_CHECK(B >= 0 && V > 0 && Vt > 0, WF3D_ERR_ARG, "wf3d_loss_cost_matrix: bad dims");
    if (B == 0) return WF3D_OK;
    WF3D_CHECK(verts && exist && tverts && counts && cost, WF3D_ERR_ARG, "wf3d_loss_cost_matrix: null pointer");
    WF3D_CHECK(B <= 65535, WF3D_ERR_UNSUPPORTED, "wf3d_loss_cost_matrix: B > 65535");
    hipLaunchKernelGGL(loss_cost_kernel, dim3(wf3d_cdiv((long)V * V, 256), B), dim3(256), 0, (hipStream_t)stream, verts,
                       sample_stride, vertex_stride, exist, tverts, Vt, counts, V, cost);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_loss_terms(const float* verts, long sample_stride, long vertex_stride, const float* exist,
                               const float* edge, int Ep, const float* tverts, int Vt, const float* texist,
                               const float* tlabel, int Et, const int32_t* m_pred, const int32_t* m_tgt,
                               const int32_t* m_off, int n_match, int B, int V, float w_vertex, float w_exist,
                               float w_edge, float* dverts, float* dexist, float* dedge, float* losses, void* ws,
                               size_t ws_bytes, void* stream) {
    WF3D_CHECK(B > 0 && V > 0 && Ep >= 0 && Et >= 0 && n_match >= 0, WF3D_ERR_ARG, "wf3d_loss_terms: bad dims");
    WF3D_CHECK(verts && exist && tverts && texist && m_off && dverts && dexist && losses, WF3D_ERR_ARG, "wf3d_loss_terms: null pointer");
    WF3D_CHECK(Ep == 0 || (edge && dedge), WF3D_ERR_ARG, "wf3d_loss_terms: null edge tensors");
    WF3D_CHECK(ws && ws_bytes >= (size_t)B * 3 * sizeof(float), WF3D_ERR_WS, "wf3d_loss_terms: workspace too small");
    const int min_e = (Ep > 0 && Et > 0 && tlabel) ? (Ep < Et ? Ep : Et) : 0;
    const float nv = 3.0f * (float)n_match, ne = (float)B * V, nd = (float)B * min_e;
    const float gv = n_match ? w_vertex / nv : 0.f, ge = w_exist / ne, gd = min_e ? w_edge / nd : 0.f;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_terms_kernel, dim3(B), dim3(256), 0, st, verts, sample_stride, vertex_stride, exist, edge, Ep,
                       tverts, Vt, texist, tlabel, Et, min_e, m_pred, m_tgt, m_off, V, B, gv, ge, gd, dverts, dexist,
                       dedge, (float*)ws);
    WF3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(1), 0, st, (const float*)ws, B, nv, ne, nd, w_vertex, w_exist,
                       w_edge, losses);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}
