// Edge-head data movement and the split first layer of the edge-pair MLP.
//
// Reference (EdgePredictor.py:117-137) gathers f[i], f[j], c[i], c[j], |c_i-c_j|
// into an [E, 1031] matrix (134.6 MB per sample at V=256) and multiplies it by
// edge_mlp.0.weight [512, 1031].  Splitting that weight by the concat's column
// blocks, W = [Wa | Wb | Wc | Wd | wd], gives
//     pre[(i,j)] = (F Wa^T + C Wc^T + b)[i] + (F Wb^T + C Wd^T)[j] + |c_i-c_j| * wd
// i.e. two V-row GEMMs (wf3d_gemm) and the E-row *combine* below; the concat is
// never materialised (validated to 2.5e-7 in §A.4).
//
// Rows are COMPACT over the batch: sample s owns vertex rows voff[s]..voff[s+1]-1
// and edge rows eoff[s]..eoff[s+1]-1 (its V_s(V_s-1)/2 pairs in the reference's
// lexicographic i<j order, EdgePredictor.py:83-86).
#include "wf3d_common.h"

namespace {

// ---- vertex rows: gather counts[s] leading vertices of each sample ------------
__global__ __launch_bounds__(256) void gather_verts_kernel(const float* __restrict__ verts, long s_stride, long v_stride,
                                                            const int32_t* __restrict__ voff,
                                                            const int32_t* __restrict__ vsample, int Rv, int vd,
                                                            float* __restrict__ cv) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Rv * vd) return;
    const int r = idx / vd, k = idx % vd;
    const int s = vsample[r];
    cv[idx] = verts[(long)s * s_stride + (long)(r - voff[s]) * v_stride + k];
}

__global__ __launch_bounds__(256) void scatter_dverts_kernel(const float* __restrict__ dcv, const int32_t* __restrict__ voff,
                                                              int B, int V, int vd, float* __restrict__ dverts) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * V * vd) return;
    const int k = idx % vd, v = (idx / vd) % V, s = idx / (vd * V);
    const int n = voff[s + 1] - voff[s];
    dverts[idx] = v < n ? dcv[(size_t)(voff[s] + v) * vd + k] : 0.f;
}

// ---- pair combine forward: one wave per edge row --------------------------------
// pre[e,:] = Pa[i,:] + Pb[j,:] + delta*wd ; also LayerNorm stats of the row (the
// wave already holds it) and delta.
// Each wave walks EPW consecutive edge rows: the per-column constants (distance weight, gamma, beta) are loaded once,
// and consecutive edges share their first vertex i (lexicographic order), so the Pa row stays in registers until i
// changes.
template <int NS>
__global__ __launch_bounds__(256) void pair_fwd_kernel(const float* __restrict__ Pa, const float* __restrict__ Pb,
                                                        const float* __restrict__ cv, const float* __restrict__ wd,
                                                        int wd_stride, const int32_t* __restrict__ voff,
                                                        const int32_t* __restrict__ eoff,
                                                        const int32_t* __restrict__ esample, int Re, int H, float eps,
                                                        float* __restrict__ pre, float* __restrict__ mu,
                                                        float* __restrict__ rs, float* __restrict__ delta,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int act, uint32_t seed, uint32_t thresh, float dscale,
                                                        float* __restrict__ h_sx8, int vd) {
    constexpr int EPW = 8;
    const int lane = threadIdx.x & 63;
    const int e0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * EPW;
    if (e0 >= Re) return;
    const int e1 = min(Re, e0 + EPW);
    f32x4 wdv[NS], g4[NS], b4[NS], pa[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const int c = lane * 4 + 256 * t;
        wdv[t] = g4[t] = b4[t] = pa[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < H) {
#pragma unroll
            for (int k = 0; k < 4; ++k) wdv[t][k] = wd[(size_t)(c + k) * wd_stride];
            if (h_sx8) { g4[t] = *reinterpret_cast<const f32x4*>(gamma + c); b4[t] = *reinterpret_cast<const f32x4*>(beta + c); }
        }
    }
    int s = esample[e0], vbase = voff[s], v = voff[s + 1] - vbase, eend = eoff[s + 1];
    int i, j;
    edge_ij(e0 - eoff[s], v, i, j);
    int pa_row = -1;
    for (int e = e0; e < e1; ++e) {
        if (e == eend) {                       // first edge of the next sample that has edges
            s = esample[e]; vbase = voff[s]; v = voff[s + 1] - vbase; eend = eoff[s + 1];
            i = 0; j = 1;
        }
        const int ri = vbase + i, rj = vbase + j;
        float dl2 = 0.f;                                   // |c_i - c_j|^2 over the vd coordinates (vd = 3: x, y, z in this order)
        for (int k = 0; k < vd; ++k) { const float d = cv[ri * vd + k] - cv[rj * vd + k]; dl2 = fmaf(d, d, dl2); }
        const float dl = sqrtf(dl2);
        f32x4 val[NS];
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int c = lane * 4 + 256 * t;
            val[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (c < H) {
                if (ri != pa_row) pa[t] = *reinterpret_cast<const f32x4*>(Pa + (size_t)ri * H + c);
                const f32x4 b = *reinterpret_cast<const f32x4*>(Pb + (size_t)rj * H + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    val[t][k] = pa[t][k] + b[k] + dl * wdv[t][k];
                    sum += val[t][k];
                }
                if (pre) *reinterpret_cast<f32x4*>(pre + (size_t)e * H + c) = val[t];
            }
        }
        pa_row = ri;
        const float mean = wf3d_wave_sum(sum) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int c = lane * 4 + 256 * t;
            if (c < H) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float d = val[t][k] - mean; q += d * d; }
            }
        }
        const float var = wf3d_wave_sum(q) / (float)H;
        const float rstd = 1.0f / sqrtf(var + eps);
        if (lane == 0) {
            mu[e] = mean;
            rs[e] = rstd;
            delta[e] = dl;
        }
        // h = drop(act(LN(pre))) as the sx8 operand of the next Linear, from the row the wave still holds
        // (what wf3d_ln_prep would produce from a second read of pre; same dropout counter: row e, column c).
        if (h_sx8) {
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                const int c = lane * 4 + 256 * t;
                if (c < H) {
                    float o[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        o[k] = wf3d_act_rt(act, (val[t][k] - mean) * rstd * g4[t][k] + b4[t][k]);
                        if (thresh) o[k] = wf3d_keep(seed, (uint32_t)e, (uint32_t)(c + k), thresh) ? o[k] * dscale : 0.f;
                    }
                    // sx8 group = 8 columns = lanes (2m, 2m+1): even lane stores the 8 high parts, odd lane the 8 low parts
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    bf16x4 hi, lo;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { hi[k] = (__bf16)o[k]; lo[k] = (__bf16)(o[k] - (float)hi[k]); }
                    const uint2 mine_hi = __builtin_bit_cast(uint2, hi), mine_lo = __builtin_bit_cast(uint2, lo);
                    const bool odd = lane & 1;
                    uint2 send = odd ? mine_hi : mine_lo, got;
                    got.x = __shfl_xor((int)send.x, 1, 64);
                    got.y = __shfl_xor((int)send.y, 1, 64);
                    const uint4 w = odd ? make_uint4(got.x, got.y, mine_lo.x, mine_lo.y) : make_uint4(mine_hi.x, mine_hi.y, got.x, got.y);
                    *reinterpret_cast<uint4*>(h_sx8 + (size_t)e * H + (c & ~7) + (odd ? 4 : 0)) = w;
                }
            }
        }
        if (++j == v) { ++i; j = i + 1; }
    }
}

// ---- pair combine backward: one workgroup per vertex row --------------------------
// dPa[v] = sum over edges with i = v of dpre[e]; dPb[v] = sum over edges with j = v;
// dcv[v] += sum over incident edges of (dpre[e]·wd) * (c_v - c_other)/delta_e
// (segmented reductions: no atomics)
template <int NS>
__global__ __launch_bounds__(256) void pair_bwd_kernel(const float* __restrict__ dpre, const float* __restrict__ delta,
                                                        const float* __restrict__ cv, const float* __restrict__ wd,
                                                        int wd_stride, const int32_t* __restrict__ voff,
                                                        const int32_t* __restrict__ eoff,
                                                        const int32_t* __restrict__ vsample, int H,
                                                        float* __restrict__ dPa, float* __restrict__ dPb,
                                                        float* __restrict__ dcv, const float* __restrict__ wcoord,
                                                        int wcoord_stride, int vd) {
    constexpr int MAXD = 8;                                          // coordinates per vertex (host-checked)
    extern __shared__ __attribute__((aligned(16))) float red[];     // [2][H] + [4][MAXD] + [4][MAXD]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x;
    const int s = vsample[r];
    const int v0 = voff[s], nv = voff[s + 1] - v0, me = r - v0;
    const int e0 = eoff[s];
    f32x4 wv[NS], aa[NS], ab[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const int c = lane * 4 + 256 * t;
        aa[t] = ab[t] = wv[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < H)
#pragma unroll
            for (int k = 0; k < 4; ++k) wv[t][k] = wd[(size_t)(c + k) * wd_stride];
    }
    float cme[MAXD], gc[MAXD];
#pragma unroll
    for (int k = 0; k < MAXD; ++k) { cme[k] = k < vd ? cv[r * vd + k] : 0.f; gc[k] = 0.f; }
    // incident edges: other = 0..nv-1 except me; 4 waves interleave
    for (int other = wave; other < nv; other += 4) {
        if (other == me) continue;
        const bool as_i = other > me;                       // (me, other) with me as i
        const int e = e0 + (as_i ? edge_offset(me, nv) + (other - me - 1) : edge_offset(other, nv) + (me - other - 1));
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int c = lane * 4 + 256 * t;
            if (c < H) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(dpre + (size_t)e * H + c);
                if (as_i) aa[t] += g; else ab[t] += g;
#pragma unroll
                for (int k = 0; k < 4; ++k) dot += g[k] * wv[t][k];
            }
        }
        dot = wf3d_wave_sum(dot);
        const float w = dot / delta[e];
        const int ro = v0 + other;
#pragma unroll
        for (int k = 0; k < MAXD; ++k) if (k < vd) gc[k] += w * (cme[k] - cv[ro * vd + k]);
    }
    float* ra = red;
    float* rb = red + H;
    float* rc = red + 2 * H;                    // [4 waves][MAXD]
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                const int c = lane * 4 + 256 * t;
                if (c < H) {
                    f32x4 x = aa[t], y = ab[t];
                    if (w) { x += *reinterpret_cast<f32x4*>(ra + c); y += *reinterpret_cast<f32x4*>(rb + c); }
                    *reinterpret_cast<f32x4*>(ra + c) = x;
                    *reinterpret_cast<f32x4*>(rb + c) = y;
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < MAXD; ++k) rc[w * MAXD + k] = gc[k];
            }
        }
        __syncthreads();
    }
    // the coordinate columns of the first edge Linear: pre also held c_i . Wc^T + c_j . Wd^T, so this vertex's coordinates
    // receive dPa[r] . Wc + dPb[r] . Wd — two [rows, H] x [H, 3] products that would otherwise be separate launches
    float qc[MAXD];
#pragma unroll
    for (int k = 0; k < MAXD; ++k) qc[k] = 0.f;
    for (int c = threadIdx.x; c < H; c += 256) {
        const float a = ra[c], b = rb[c];
        dPa[(size_t)r * H + c] = a;
        dPb[(size_t)r * H + c] = b;
        if (wcoord) {
            const float* w = wcoord + (size_t)c * wcoord_stride;          // [Wc[c, 0..vd) | Wd[c, 0..vd)]
#pragma unroll
            for (int k = 0; k < MAXD; ++k) if (k < vd) qc[k] += a * w[k] + b * w[vd + k];
        }
    }
    float* rq = red + 2 * H + 4 * MAXD;
    if (wcoord) {
#pragma unroll
        for (int k = 0; k < MAXD; ++k) qc[k] = wf3d_wave_sum(qc[k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < MAXD; ++k) rq[wave * MAXD + k] = qc[k];
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < vd) {
        const int k = threadIdx.x;
        float v = (rc[k] + rc[MAXD + k]) + (rc[2 * MAXD + k] + rc[3 * MAXD + k]);
        if (wcoord) v += (rq[k] + rq[MAXD + k]) + (rq[2 * MAXD + k] + rq[3 * MAXD + k]);
        dcv[r * vd + k] = v;
    }
}

// ---- sigmoid into the padded [B, max_E] output; the padding (exactly 0.0) is written here too ----
__global__ __launch_bounds__(256) void edge_prob_fwd_kernel(const float* __restrict__ logit, const int32_t* __restrict__ eoff,
                                                             int B, int max_e, float* __restrict__ probs) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * max_e) return;
    const int s = (int)(idx / max_e), j = (int)(idx - (long)s * max_e);
    const int e0 = eoff[s], ne = eoff[s + 1] - e0;
    probs[idx] = j < ne ? 1.0f / (1.0f + expf(-logit[e0 + j])) : 0.0f;
}
__global__ __launch_bounds__(256) void edge_prob_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ dprobs,
                                                             const int32_t* __restrict__ eoff,
                                                             const int32_t* __restrict__ esample, int Re, int max_e,
                                                             float* __restrict__ dlogit) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Re) return;
    const int s = esample[e];
    const size_t o = (size_t)s * max_e + (e - eoff[s]);
    const float p = probs[o];
    dlogit[e] = dprobs[o] * p * (1.0f - p);
}

// ---- vertex head tail: existence sigmoid, count, and its backward -----------------
__global__ __launch_bounds__(256) void vertex_finalize_fwd_kernel(const float* __restrict__ o, int V, int vd,
                                                                   float* __restrict__ exist, int64_t* __restrict__ counts) {
    __shared__ int cnt;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int mine = 0;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float p = 1.0f / (1.0f + expf(-o[((size_t)b * V + v) * vd + 3]));
        exist[(size_t)b * V + v] = p;
        mine += p > 0.5f ? 1 : 0;
    }
    if (mine) atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) counts[b] = cnt;
}
__global__ __launch_bounds__(256) void vertex_finalize_bwd_kernel(const float* __restrict__ exist, const float* __restrict__ dexist,
                                                                   const float* __restrict__ d_o_in, int in_dim, int BV, int vd,
                                                                   float* __restrict__ d_o) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= BV * vd) return;
    const int k = idx % vd, bv = idx / vd;
    float g = (d_o_in && k < in_dim) ? d_o_in[bv * in_dim + k] : 0.f;
    if (k == 3 && dexist) { const float p = exist[bv]; g += dexist[bv] * p * (1.0f - p); }
    d_o[idx] = g;
}

}  // namespace

extern "C" int wf3d_edge_gather_verts(const float* verts, long sample_stride, long vertex_stride, const int32_t* voff,
                                      const int32_t* vsample, int Rv, int vd, float* cv, void* stream) {
    WF3D_CHECK(Rv >= 0 && vd >= 1 && vd <= 8, WF3D_ERR_ARG, "wf3d_edge_gather_verts: bad Rv, or vertex_dim not in 1..8");
    if (Rv == 0) return WF3D_OK;
    WF3D_CHECK(verts && voff && vsample && cv, WF3D_ERR_ARG, "wf3d_edge_gather_verts: null pointer");
    hipLaunchKernelGGL(gather_verts_kernel, dim3(wf3d_cdiv((long)Rv * vd, 256)), dim3(256), 0, (hipStream_t)stream, verts,
                       sample_stride, vertex_stride, voff, vsample, Rv, vd, cv);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_edge_scatter_dverts(const float* dcv, const int32_t* voff, int B, int V, int vd, float* dverts, void* stream) {
    WF3D_CHECK(B >= 0 && V >= 0 && vd >= 1 && vd <= 8, WF3D_ERR_ARG, "wf3d_edge_scatter_dverts: bad dims");
    if (B * V == 0) return WF3D_OK;
    WF3D_CHECK(voff && dverts, WF3D_ERR_ARG, "wf3d_edge_scatter_dverts: null pointer");
    hipLaunchKernelGGL(scatter_dverts_kernel, dim3(wf3d_cdiv((long)B * V * vd, 256)), dim3(256), 0, (hipStream_t)stream,
                       dcv, voff, B, V, vd, dverts);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

static int pair_fwd_impl(const float* Pa, const float* Pb, const float* cv, const float* wdelta, int wdelta_stride,
                         const int32_t* voff, const int32_t* eoff, const int32_t* esample, int Re, int H, float eps,
                         float* pre, float* mu, float* rs, float* delta, const float* gamma, const float* beta, int act,
                         float drop_p, uint32_t drop_seed, void* h_sx8, int vd, void* stream);

extern "C" int wf3d_edge_pair_fwd(const float* Pa, const float* Pb, const float* cv, const float* wdelta,
                                  int wdelta_stride, const int32_t* voff, const int32_t* eoff, const int32_t* esample,
                                  int Re, int H, int vd, float eps, float* pre, float* mu, float* rs, float* delta,
                                  void* stream) {
    return pair_fwd_impl(Pa, Pb, cv, wdelta, wdelta_stride, voff, eoff, esample, Re, H, eps, pre, mu, rs, delta, nullptr,
                         nullptr, 0, 0.f, 0u, nullptr, vd, stream);
}

extern "C" int wf3d_edge_pair_fwd_ln(const float* Pa, const float* Pb, const float* cv, const float* wdelta,
                                     int wdelta_stride, const int32_t* voff, const int32_t* eoff, const int32_t* esample,
                                     int Re, int H, int vd, float eps, float* pre, float* mu, float* rs, float* delta,
                                     const float* gamma, const float* beta, int act, float drop_p, uint32_t drop_seed,
                                     void* h_sx8, void* stream) {
    WF3D_CHECK(gamma && beta && h_sx8, WF3D_ERR_ARG, "wf3d_edge_pair_fwd_ln: null pointer");
    WF3D_CHECK(H % 8 == 0 && ((uintptr_t)h_sx8 % 16 == 0) && ((uintptr_t)gamma % 16 == 0) && ((uintptr_t)beta % 16 == 0),
               WF3D_ERR_UNSUPPORTED, "wf3d_edge_pair_fwd_ln: needs hidden %% 8 == 0 and 16-byte aligned pointers");
    WF3D_CHECK(act >= 0 && act <= 2 && drop_p >= 0.f && drop_p < 1.f, WF3D_ERR_ARG, "wf3d_edge_pair_fwd_ln: bad act/drop");
    return pair_fwd_impl(Pa, Pb, cv, wdelta, wdelta_stride, voff, eoff, esample, Re, H, eps, pre, mu, rs, delta, gamma, beta,
                         act, drop_p, drop_seed, h_sx8, vd, stream);
}

static int pair_fwd_impl(const float* Pa, const float* Pb, const float* cv, const float* wdelta, int wdelta_stride,
                         const int32_t* voff, const int32_t* eoff, const int32_t* esample, int Re, int H, float eps,
                         float* pre, float* mu, float* rs, float* delta, const float* gamma, const float* beta, int act,
                         float drop_p, uint32_t drop_seed, void* h_sx8, int vd, void* stream) {
    const uint32_t thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    const float dscale = 1.0f / (1.0f - drop_p);
    WF3D_CHECK(Re >= 0 && H > 0 && vd >= 1 && vd <= 8, WF3D_ERR_ARG, "wf3d_edge_pair_fwd: bad dims (vertex_dim must be 1..8)");
    WF3D_CHECK(H % 4 == 0 && H <= 2048, WF3D_ERR_UNSUPPORTED, "wf3d_edge_pair_fwd: hidden %d must be a multiple of 4, <= 2048", H);
    if (Re == 0) return WF3D_OK;
    WF3D_CHECK(Pa && Pb && cv && wdelta && voff && eoff && esample && (pre || h_sx8) && mu && rs && delta, WF3D_ERR_ARG,
               "wf3d_edge_pair_fwd: null pointer");
    const int ns = wf3d_cdiv(H, 256);
    hipStream_t st = (hipStream_t)stream;
#define WF3D_PF(NS_)                                                                                                 \
    hipLaunchKernelGGL((pair_fwd_kernel<NS_>), dim3(wf3d_cdiv(Re, 4 * 8)), dim3(256), 0, st, Pa, Pb, cv, wdelta,           \
                       wdelta_stride, voff, eoff, esample, Re, H, eps, pre, mu, rs, delta, gamma, beta, act, drop_seed,    \
                       thresh, dscale, (float*)h_sx8, vd)
    if (ns <= 1) WF3D_PF(1); else if (ns <= 2) WF3D_PF(2); else if (ns <= 4) WF3D_PF(4); else WF3D_PF(8);
#undef WF3D_PF
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_edge_pair_bwd(const float* dpre, const float* delta, const float* cv, const float* wdelta,
                                  int wdelta_stride, const int32_t* voff, const int32_t* eoff, const int32_t* vsample,
                                  int Rv, int H, int vd, float* dPa, float* dPb, float* dcv, const float* wcoord,
                                  int wcoord_stride, void* stream) {
    WF3D_CHECK(Rv >= 0 && H > 0 && vd >= 1 && vd <= 8, WF3D_ERR_ARG, "wf3d_edge_pair_bwd: bad dims (vertex_dim must be 1..8)");
    WF3D_CHECK(!wcoord || wcoord_stride >= 2 * vd, WF3D_ERR_ARG, "wf3d_edge_pair_bwd: wcoord rows hold [Wc | Wd] = 2 x vertex_dim floats");
    WF3D_CHECK(H % 4 == 0 && H <= 2048, WF3D_ERR_UNSUPPORTED, "wf3d_edge_pair_bwd: hidden %d must be a multiple of 4, <= 2048", H);
    if (Rv == 0) return WF3D_OK;
    WF3D_CHECK(dpre && delta && cv && wdelta && voff && eoff && vsample && dPa && dPb && dcv, WF3D_ERR_ARG,
               "wf3d_edge_pair_bwd: null pointer");
    const int ns = wf3d_cdiv(H, 256);
    const size_t lds = ((size_t)2 * H + 64) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define WF3D_PB(NS_)                                                                                                 \
    hipLaunchKernelGGL((pair_bwd_kernel<NS_>), dim3(Rv), dim3(256), lds, st, dpre, delta, cv, wdelta, wdelta_stride,   \
                       voff, eoff, vsample, H, dPa, dPb, dcv, wcoord, wcoord_stride, vd)
    if (ns <= 1) WF3D_PB(1); else if (ns <= 2) WF3D_PB(2); else if (ns <= 4) WF3D_PB(4); else WF3D_PB(8);
#undef WF3D_PB
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_edge_prob_fwd(const float* logit, const int32_t* eoff, int B, int max_e, float* probs, void* stream) {
    WF3D_CHECK(B >= 0 && max_e >= 0, WF3D_ERR_ARG, "wf3d_edge_prob_fwd: bad dims");
    if (B == 0 || max_e == 0) return WF3D_OK;
    WF3D_CHECK(logit && eoff && probs, WF3D_ERR_ARG, "wf3d_edge_prob_fwd: null pointer");
    hipLaunchKernelGGL(edge_prob_fwd_kernel, dim3(wf3d_cdiv((long)B * max_e, 256)), dim3(256), 0, (hipStream_t)stream, logit, eoff,
                       B, max_e, probs);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_edge_prob_bwd(const float* probs, const float* dprobs, const int32_t* eoff, const int32_t* esample,
                                  int Re, int max_e, float* dlogit, void* stream) {
    WF3D_CHECK(Re >= 0 && max_e >= 0, WF3D_ERR_ARG, "wf3d_edge_prob_bwd: bad dims");
    if (Re == 0) return WF3D_OK;
    WF3D_CHECK(probs && dprobs && eoff && esample && dlogit, WF3D_ERR_ARG, "wf3d_edge_prob_bwd: null pointer");
    hipLaunchKernelGGL(edge_prob_bwd_kernel, dim3(wf3d_cdiv(Re, 256)), dim3(256), 0, (hipStream_t)stream, probs, dprobs,
                       eoff, esample, Re, max_e, dlogit);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_vertex_finalize_fwd(const float* o, int B, int V, int vertex_dim, float* exist, int64_t* counts,
                                        void* stream) {
    WF3D_CHECK(B >= 0 && V > 0 && vertex_dim >= 4, WF3D_ERR_ARG, "wf3d_vertex_finalize_fwd: bad dims (vertex_dim must be >= 4)");
    if (B == 0) return WF3D_OK;
    WF3D_CHECK(o && exist && counts, WF3D_ERR_ARG, "wf3d_vertex_finalize_fwd: null pointer");
    hipLaunchKernelGGL(vertex_finalize_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, o, V, vertex_dim, exist, counts);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_vertex_finalize_bwd(const float* exist, const float* dexist, const float* d_o_in, int in_dim, int B, int V,
                                        int vertex_dim, float* d_o, void* stream) {
    WF3D_CHECK(B >= 0 && V > 0 && vertex_dim >= 4 && in_dim >= 1 && in_dim <= vertex_dim, WF3D_ERR_ARG, "wf3d_vertex_finalize_bwd: bad dims");
    if (B == 0) return WF3D_OK;
    WF3D_CHECK(exist && d_o, WF3D_ERR_ARG, "wf3d_vertex_finalize_bwd: null pointer");
    hipLaunchKernelGGL(vertex_finalize_bwd_kernel, dim3(wf3d_cdiv((long)B * V * vertex_dim, 256)), dim3(256), 0,
                       (hipStream_t)stream, exist, dexist, d_o_in, in_dim, B * V, vertex_dim, d_o);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}
