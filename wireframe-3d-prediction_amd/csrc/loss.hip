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


// ---------------------------------------------------------------------------
// Linear sum assignment on the device: shortest-augmenting-path (Jonker-Volgenant, the
// algorithm behind scipy.optimize.linear_sum_assignment) on a square [V, V] cost matrix, one
// wave64 per sample, lanes = columns (V <= 256: up to 4 columns per lane), dual variables and
// path costs in fp64 like scipy.  Removes the loss's only device->host sync: with it the
// whole train step is asynchronous.  Output col4row[b, p] = column assigned to prediction p.
// Ties: a lane-order rule (unassigned column first, then lowest index); the dummy columns of
// the wireframe cost are identical, so only assignments to real targets are meaningful and
// those are unique whenever the optimum is.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void lsa_kernel(const float* __restrict__ cost, int V, int32_t* __restrict__ col4row_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
    double* u = reinterpret_cast<double*>(lsm);            // [V] row duals
    double* v = u + V;                                     // [V] column duals
    double* spc = v + V;                                   // [V] shortest path costs
    int* path = reinterpret_cast<int*>(spc + V);           // [V]
    int* row4col = path + V;                               // [V]
    int* col4row = row4col + V;                            // [V]
    int* srlist = col4row + V;                             // [V]
    unsigned char* in_sc = reinterpret_cast<unsigned char*>(srlist + V);   // [V]
    const int lane = threadIdx.x, b = blockIdx.x;
    const float* C = cost + (size_t)b * V * V;
    for (int j = lane; j < V; j += 64) { u[j] = 0.0; v[j] = 0.0; row4col[j] = -1; col4row[j] = -1; }
    __syncthreads();
    for (int cur = 0; cur < V; ++cur) {
        for (int j = lane; j < V; j += 64) { spc[j] = INFINITY; in_sc[j] = 0; }
        __syncthreads();
        int i = cur, sink = -1, nsr = 0;
        double min_val = 0.0;
        while (sink < 0) {
            if (lane == 0) srlist[nsr] = i;
            ++nsr;
            const double ui = u[i];
            double best = INFINITY;
            int bestj = 0x7fffffff, best_free = 0;
            for (int j = lane; j < V; j += 64) {
                if (in_sc[j]) continue;
                const double r = min_val + (double)C[(size_t)i * V + j] - ui - v[j];
                double cur_spc = spc[j];
                if (r < cur_spc) { cur_spc = r; spc[j] = r; path[j] = i; }
                const int fr = row4col[j] < 0;
                if (cur_spc < best || (cur_spc == best && (fr > best_free || (fr == best_free && j < bestj)))) {
                    best = cur_spc; bestj = j; best_free = fr;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o, 64);
                const int oj = __shfl_xor(bestj, o, 64), of = __shfl_xor(best_free, o, 64);
                if (ob < best || (ob == best && (of > best_free || (of == best_free && oj < bestj)))) {
                    best = ob; bestj = oj; best_free = of;
                }
            }
            min_val = best;
            if (lane == 0) in_sc[bestj] = 1;
            __syncthreads();
            const int r4c = row4col[bestj];
            if (r4c < 0) sink = bestj; else i = r4c;
        }
        // dual update
        for (int k = lane; k < nsr; k += 64) {
            const int r = srlist[k];
            u[r] += (r == cur) ? min_val : min_val - spc[col4row[r]];
        }
        for (int j = lane; j < V; j += 64)
            if (in_sc[j]) v[j] -= min_val - spc[j];
        __syncthreads();
        // augment along the path (sequential, wave-uniform)
        if (lane == 0) {
            int j = sink;
            while (true) {
                const int r = path[j];
                row4col[j] = r;
                const int prev = col4row[r];
                col4row[r] = j;
                j = prev;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }
    for (int p = lane; p < V; p += 64) col4row_out[(size_t)b * V + p] = col4row[p];
}

// V <= 64: the whole solver state lives in registers — lane j owns column j (v_j, shortest-path cost, predecessor,
// row4col_j, visited flag) and row j (u_j, col4row_j); a uniform row/column index is read with v_readlane, the
// arg-min is one fp64 wave minimum plus two ballots, and the cost matrix sits in LDS (<= 16 KB) instead of being
// re-read from L2 once per path step.  Same algorithm, same tie rule as lsa_kernel.
__device__ __forceinline__ double lane_read_f64(double x, int lane_uniform) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, lane_uniform);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), lane_uniform);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// RECT (counts given): the wireframe cost has V - count identical dummy columns (cost e_p for prediction p), so the
// square problem equals the rectangular one "assign each of the count real targets to a distinct prediction, cost
// c[p,t] - e_p" (+ a constant): rows = targets, lane = prediction, only `count` augmentations instead of V (the
// dataset's counts are 4..38 of 64).  Unmatched predictions get the dummy columns count, count+1, ... in lane order.
template <bool RECT>
__global__ __launch_bounds__(64) void lsa64_kernel(const float* __restrict__ cost, const int64_t* __restrict__ counts, int V,
                                                    int32_t* __restrict__ col4row_out) {
    __shared__ float Cs[64 * 64];
    const int lane = threadIdx.x, b = blockIdx.x;
    const float* C = cost + (size_t)b * V * V;
    const int nrow = RECT ? min((int)counts[b], V) : V;
    const bool col_ok = lane < V;
    double ep = 0.0;
    if (RECT) {
        for (int idx = lane; idx < nrow * V; idx += 64) Cs[idx] = C[(size_t)(idx % V) * V + idx / V];      // Cs[t][p] = cost[p][t]
        if (col_ok && nrow < V) ep = (double)C[(size_t)lane * V + V - 1];
    } else {
        for (int idx = lane; idx < V * V; idx += 64) Cs[idx] = C[idx];
    }
    __syncthreads();
    double u = 0.0, v = 0.0;
    int row4col = -1, col4row = -1;
    for (int cur = 0; cur < nrow; ++cur) {
        double spc = INFINITY;
        bool in_sc = !col_ok;
        int path = -1;
        int i = cur, sink = -1;
        double min_val = 0.0;
        unsigned long long sr_mask = 0ull;
        while (sink < 0) {
            sr_mask |= 1ull << i;
            const double ui = lane_read_f64(u, i);
            if (!in_sc) {
                const double r = min_val + ((double)Cs[i * V + lane] - ep) - ui - v;
                if (r < spc) { spc = r; path = i; }
            }
            double m = in_sc ? INFINITY : spc;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmin(m, __shfl_xor(m, o, 64));
            const unsigned long long cand = __ballot(!in_sc && spc == m);
            const unsigned long long fr = cand & __ballot(row4col < 0);
            const int pick = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(fr ? fr : cand));
            min_val = m;
            if (lane == pick) in_sc = true;
            const int r4c = __builtin_amdgcn_readlane(row4col, pick);
            if (r4c < 0) sink = pick; else i = __builtin_amdgcn_readfirstlane(r4c);
        }
        // dual update: rows in SR (lane = row), columns in SC (lane = column)
        {
            const int c = col4row < 0 ? 0 : col4row;
            const double spc_c = __shfl(spc, c, 64);
            if ((sr_mask >> lane) & 1ull) u += (lane == cur) ? min_val : min_val - spc_c;
            if (in_sc && col_ok) v -= min_val - spc;
        }
        // augment along the path (wave-uniform walk)
        int j = sink;
        while (true) {
            const int r = __builtin_amdgcn_readlane(path, j);
            if (lane == j) row4col = r;
            const int prev = __builtin_amdgcn_readlane(col4row, r);
            if (lane == r) col4row = j;
            if (r == cur) break;
            j = __builtin_amdgcn_readfirstlane(prev);
        }
    }
    if (RECT) {
        // lane = prediction: its target, or the next free dummy column
        const unsigned long long un = __ballot(col_ok && row4col < 0);
        const int rank = __builtin_popcountll(un & ((1ull << lane) - 1ull));
        if (col_ok) col4row_out[(size_t)b * V + lane] = row4col >= 0 ? row4col : nrow + rank;
    } else if (col_ok) {
        col4row_out[(size_t)b * V + lane] = col4row;
    }
}

// loss terms from the device assignment: prediction p of sample b is matched iff col4row[b,p] < count[b]
__global__ __launch_bounds__(256) void loss_terms_dev_kernel(const float* __restrict__ verts, long vs_b, long vs_v,
                                                              const float* __restrict__ exist,
                                                              const float* __restrict__ edge, int Ep,
                                                              const float* __restrict__ tverts, int Vt,
                                                              const float* __restrict__ texist,
                                                              const float* __restrict__ tlabel, int Et, int min_e,
                                                              const int32_t* __restrict__ col4row,
                                                              const int64_t* __restrict__ counts, int V, int B,
                                                              float wv, float ge, float gd,
                                                              float* __restrict__ dverts, float* __restrict__ dexist,
                                                              float* __restrict__ dedge, float* __restrict__ part) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    long nm = 0;
    for (int s = 0; s < B; ++s) nm += min((long)counts[s], (long)V);        // total matches = sum of real targets
    const float gv = nm > 0 ? wv / (3.0f * (float)nm) : 0.f;
    const int cnt = (int)counts[b];
    float sv = 0.f;
    for (int p = tid; p < V; p += 256) {
        const int t = col4row[(size_t)b * V + p];
        const bool m = t < cnt;
        const float* vp = verts + (size_t)b * vs_b + (size_t)p * vs_v;
        const float* w = tverts + ((size_t)b * Vt + (m ? t : 0)) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float g = 0.f;
            if (m) {
                const float d = vp[k] - w[k], a = fabsf(d);
                sv += a < 1.0f ? 0.5f * d * d : a - 0.5f;
                g = gv * (a < 1.0f ? d : (d > 0.f ? 1.0f : -1.0f));
            }
            dverts[((size_t)b * V + p) * 3 + k] = g;
        }
    }
    float se = 0.f;
    for (int i = tid; i < V; i += 256) {
        float g;
        se += bce(exist[(size_t)b * V + i], texist[(size_t)b * V + i], g);
        dexist[(size_t)b * V + i] = ge * g;
    }
    float sd = 0.f;
    for (int i = tid; i < Ep; i += 256) {
        float g = 0.f;
        if (i < min_e) sd += bce(edge[(size_t)b * Ep + i], tlabel[(size_t)b * Et + i], g);
        dedge[(size_t)b * Ep + i] = gd * g;
    }
    sv = block_sum(sv, red); se = block_sum(se, red); sd = block_sum(sd, red);
    if (tid == 0) { part[b * 3] = sv; part[b * 3 + 1] = se; part[b * 3 + 2] = sd; }
}

__global__ void loss_final_dev_kernel(const float* __restrict__ part, const int64_t* __restrict__ counts, int B, int V,
                                      float ne, float nd, float wv, float we, float wd, float* __restrict__ out) {
    float sv = 0.f, se = 0.f, sd = 0.f;
    long nm = 0;
    for (int b = 0; b < B; ++b) { sv += part[b * 3]; se += part[b * 3 + 1]; sd += part[b * 3 + 2]; nm += min((long)counts[b], (long)V); }
    const float lv = nm > 0 ? sv / (3.0f * (float)nm) : 0.f, le = ne > 0.f ? se / ne : 0.f, ld = nd > 0.f ? sd / nd : 0.f;
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

extern "C" int wf3d_loss_assign(const float* cost, int B, int V, int32_t* col4row, void* stream) {
    WF3D_CHECK(B >= 0 && V > 0, WF3D_ERR_ARG, "wf3d_loss_assign: bad dims");
    WF3D_CHECK(V <= 1024, WF3D_ERR_UNSUPPORTED, "wf3d_loss_assign: V > 1024");
    if (B == 0) return WF3D_OK;
    WF3D_CHECK(cost && col4row, WF3D_ERR_ARG, "wf3d_loss_assign: null pointer");
    if (V <= 64) {
        hipLaunchKernelGGL(lsa64_kernel<false>, dim3(B), dim3(64), 0, (hipStream_t)stream, cost, (const int64_t*)nullptr, V, col4row);
    } else {
        const size_t lds = (size_t)V * (3 * sizeof(double) + 4 * sizeof(int) + 1) + 16;
        hipLaunchKernelGGL(lsa_kernel, dim3(B), dim3(64), lds, (hipStream_t)stream, cost, V, col4row);
    }
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_loss_terms_assigned(const float* verts, long sample_stride, long vertex_stride, const float* exist,
                                        const float* edge, int Ep, const float* tverts, int Vt, const float* texist,
                                        const float* tlabel, int Et, const int32_t* col4row, const int64_t* counts,
                                        int B, int V, float w_vertex, float w_exist, float w_edge, float* dverts,
                                        float* dexist, float* dedge, float* losses, void* ws, size_t ws_bytes,
                                        void* stream) {
    WF3D_CHECK(B > 0 && V > 0 && Ep >= 0 && Et >= 0, WF3D_ERR_ARG, "wf3d_loss_terms_assigned: bad dims");
    WF3D_CHECK(verts && exist && tverts && texist && col4row && counts && dverts && dexist && losses, WF3D_ERR_ARG,
               "wf3d_loss_terms_assigned: null pointer");
    WF3D_CHECK(Ep == 0 || (edge && dedge), WF3D_ERR_ARG, "wf3d_loss_terms_assigned: null edge tensors");
    WF3D_CHECK(ws && ws_bytes >= (size_t)B * 3 * sizeof(float), WF3D_ERR_WS, "wf3d_loss_terms_assigned: workspace too small");
    const int min_e = (Ep > 0 && Et > 0 && tlabel) ? (Ep < Et ? Ep : Et) : 0;
    const float ne = (float)B * V, nd = (float)B * min_e;
    const float ge = w_exist / ne, gd = min_e ? w_edge / nd : 0.f;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_terms_dev_kernel, dim3(B), dim3(256), 0, st, verts, sample_stride, vertex_stride, exist, edge,
                       Ep, tverts, Vt, texist, tlabel, Et, min_e, col4row, counts, V, B, w_vertex, ge, gd, dverts, dexist,
                       dedge, (float*)ws);
    WF3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_final_dev_kernel, dim3(1), dim3(1), 0, st, (const float*)ws, counts, B, V, ne, nd, w_vertex,
                       w_exist, w_edge, losses);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_loss_assign_counts(const float* cost, const int64_t* counts, int B, int V, int32_t* col4row, void* stream) {
    WF3D_CHECK(B >= 0 && V > 0, WF3D_ERR_ARG, "wf3d_loss_assign_counts: bad dims");
    if (B == 0) return WF3D_OK;
    WF3D_CHECK(cost && counts && col4row, WF3D_ERR_ARG, "wf3d_loss_assign_counts: null pointer");
    if (V > 64) return wf3d_loss_assign(cost, B, V, col4row, stream);       // general kernel: square problem
    hipLaunchKernelGGL(lsa64_kernel<true>, dim3(B), dim3(64), 0, (hipStream_t)stream, cost, counts, V, col4row);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

// ---------------------------------------------------------------------------
// Training-step meter (train.py:145-157): the reference reads total_loss.item() twice and copies sample 0's vertices to
// the host EVERY step to track loss history, best loss and a monitoring RMSE.  Here one small launch per step folds
// those into a device-resident record that the loop reads back whenever it logs (one copy per k steps, no per-step sync).
//   state[0] = steps recorded, [1] = best total loss, [2] = best vertex RMSE, [3..7] = last total / vertex / existence /
//   edge loss and RMSE, [8 .. 8 + capacity) = ring of the last `capacity` total losses (step t at 8 + t % capacity).
//   RMSE = sqrt(mean((pred - target)^2)) over the first counts[0] vertices x 3 coordinates of sample 0 (train.py:148-150).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void meter_update_kernel(const float* __restrict__ total, const float* __restrict__ lv,
                                                            const float* __restrict__ le, const float* __restrict__ ld,
                                                            const float* __restrict__ pred, long pstride,
                                                            const float* __restrict__ tgt, long tstride,
                                                            const int64_t* __restrict__ count0, int max_v,
                                                            float* __restrict__ state, int capacity) {
    __shared__ float red[4];
    long c = count0 ? count0[0] : (long)max_v;
    c = c < 0 ? 0 : (c > max_v ? max_v : c);
    float s = 0.f;
    for (long i = threadIdx.x; i < c * 3; i += 256) {
        const float d = pred[(i / 3) * pstride + i % 3] - tgt[(i / 3) * tstride + i % 3];
        s += d * d;
    }
    s = wf3d_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float sq = (red[0] + red[1]) + (red[2] + red[3]);
        const float rmse = c > 0 ? sqrtf(sq / (float)(c * 3)) : 0.f;
        const float t = total[0];
        const long step = (long)state[0];
        state[1] = step == 0 ? t : fminf(state[1], t);
        state[2] = step == 0 ? rmse : fminf(state[2], rmse);
        state[3] = t; state[4] = lv ? lv[0] : 0.f; state[5] = le ? le[0] : 0.f; state[6] = ld ? ld[0] : 0.f; state[7] = rmse;
        if (capacity > 0) state[8 + step % capacity] = t;
        state[0] = (float)(step + 1);
    }
}

extern "C" int wf3d_meter_update(const float* total, const float* vertex_loss, const float* existence_loss, const float* edge_loss,
                                 const float* pred_vertices, long pred_stride, const float* target_vertices, long target_stride,
                                 const int64_t* count0, int max_v, float* state, int capacity, void* stream) {
    WF3D_CHECK(max_v >= 0 && capacity >= 0, WF3D_ERR_ARG, "wf3d_meter_update: bad dims");
    WF3D_CHECK(total && state && (max_v == 0 || (pred_vertices && target_vertices)), WF3D_ERR_ARG, "wf3d_meter_update: null pointer");
    hipLaunchKernelGGL(meter_update_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, total, vertex_loss, existence_loss, edge_loss,
                       pred_vertices, pred_stride, target_vertices, target_stride, count0, max_v, state, capacity);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}
