// Fused 4-way pooling over point_features [B, N, C] (HBM-bound, one pass).
//
// The reference reads point_features four times (masked mean + masked max in
// PointNetEncoder.py:103-111, unmasked mean + max in VertexPredictor.py:86-88);
// here one streaming pass yields all four reductions plus the two arg-max index
// vectors the backward scatter needs.
//
// pool4_partial_v4_kernel (C % 64 == 0): a wave reads 4 point rows x 64 channels per load instruction, 16 B per lane
// (lane = row-in-group x float4 column), four loads in flight per lane; each lane folds its rows in registers, the
// 4 row groups of a wave are folded by a WAVEFRONT SHUFFLE tree (xor 16, 32), the 4 waves of the workgroup by an
// LDS tree; the N axis is split over blockIdx.y so that thousands of workgroups fill the chip, and a tiny second
// kernel folds the splits.  Every fold keeps torch.max's first-max rule (equal values: the smaller n wins).
// pool4_partial_kernel (any C): thread = channel, 4-byte loads — the fallback.
#include <stdlib.h>

#include "wf3d_common.h"

namespace {

struct PoolPart {           // per (b, split, c)
    float msum, usum, mmax, umax;
    int arg_m, arg_u;
};

__global__ __launch_bounds__(256) void point_valid_kernel(const float* __restrict__ x, int M, int in_dim,
                                                           float* __restrict__ valid) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float s = 0.f;
    for (int k = 0; k < in_dim; ++k) s += fabsf(x[(size_t)m * in_dim + k]);
    valid[m] = s > 1e-9f ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(256) void pool4_partial_kernel(const float* __restrict__ pf, const float* __restrict__ valid,
                                                             int N, int C, int nsplit, int npb,
                                                             PoolPart* __restrict__ part, float* __restrict__ cnt_part) {
    const int b = blockIdx.z, sp = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int n0 = sp * npb, n1 = min(N, n0 + npb);
    float msum = 0.f, usum = 0.f, mmax = -INFINITY, umax = -INFINITY, cnt = 0.f;
    int am = -1, au = -1;
    if (c < C) {
        const float* p = pf + ((size_t)b * N + n0) * C + c;
        const float* vp = valid + (size_t)b * N;
        for (int n = n0; n < n1; ++n, p += C) {
            const float v = *p;
            const float ok = vp[n];
            usum += v;
            if (v > umax || au < 0) { umax = v; au = n; }
            if (ok != 0.f) {
                msum += v; cnt += 1.f;
                if (v > mmax || am < 0) { mmax = v; am = n; }
            }
        }
        PoolPart o{msum, usum, mmax, umax, am, au};
        part[((size_t)b * nsplit + sp) * C + c] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt_part[b * nsplit + sp] = cnt;
}

struct Pool4 {                 // running state of 4 channels
    f32x4 msum, usum, mmax, umax;
    int am[4], au[4];
};

// fold `o` into `a`; on equal maxima the smaller point index wins (torch.max returns the first maximum)
__device__ __forceinline__ void pool4_merge(Pool4& a, const Pool4& o) {
    a.msum += o.msum;
    a.usum += o.usum;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (o.au[e] >= 0 && (a.au[e] < 0 || o.umax[e] > a.umax[e] || (o.umax[e] == a.umax[e] && o.au[e] < a.au[e]))) {
            a.umax[e] = o.umax[e]; a.au[e] = o.au[e];
        }
        if (o.am[e] >= 0 && (a.am[e] < 0 || o.mmax[e] > a.mmax[e] || (o.mmax[e] == a.mmax[e] && o.am[e] < a.am[e]))) {
            a.mmax[e] = o.mmax[e]; a.am[e] = o.am[e];
        }
    }
}

__device__ __forceinline__ Pool4 pool4_shfl_xor(const Pool4& a, int mask) {
    Pool4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o.msum[e] = __shfl_xor(a.msum[e], mask, 64); o.usum[e] = __shfl_xor(a.usum[e], mask, 64);
        o.mmax[e] = __shfl_xor(a.mmax[e], mask, 64); o.umax[e] = __shfl_xor(a.umax[e], mask, 64);
        o.am[e] = __shfl_xor(a.am[e], mask, 64);     o.au[e] = __shfl_xor(a.au[e], mask, 64);
    }
    return o;
}

__global__ __launch_bounds__(256) void pool4_partial_v4_kernel(const float* __restrict__ pf, const float* __restrict__ valid,
                                                                int N, int C, int nsplit, int npb,
                                                                PoolPart* __restrict__ part, float* __restrict__ cnt_part) {
    __shared__ Pool4 s_p[3][16];
    __shared__ float s_cnt[4];
    const int b = blockIdx.z, sp = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rsub = lane >> 4, c = blockIdx.x * 64 + (lane & 15) * 4;
    const int n0 = sp * npb, n1 = min(N, n0 + npb);
    Pool4 a;
    a.msum = a.usum = f32x4{0.f, 0.f, 0.f, 0.f};
    a.mmax = a.umax = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int e = 0; e < 4; ++e) a.am[e] = a.au[e] = -1;
    float cnt = 0.f;
    const float* base = pf + (size_t)b * N * C + c;
    const float* vp = valid + (size_t)b * N;
    constexpr int U = 4;                                   // 4 independent 16-B loads per lane in flight
    for (int n = n0 + wave * 4 + rsub; n < n1; n += 16 * U) {
        f32x4 v[U];
        float ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int nn = n + 16 * u;
            if (nn < n1) { v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (size_t)nn * C)); ok[u] = vp[nn]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int nn = n + 16 * u;
            if (nn < n1) {
                a.usum += v[u];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[u][e] > a.umax[e] || a.au[e] < 0) { a.umax[e] = v[u][e]; a.au[e] = nn; }
                if (ok[u] != 0.f) {
                    a.msum += v[u];
                    cnt += 1.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (v[u][e] > a.mmax[e] || a.am[e] < 0) { a.mmax[e] = v[u][e]; a.am[e] = nn; }
                }
            }
        }
    }
    // wavefront shuffle tree over the wave's 4 row groups
    pool4_merge(a, pool4_shfl_xor(a, 16));
    pool4_merge(a, pool4_shfl_xor(a, 32));
    cnt += __shfl_xor(cnt, 16, 64);
    cnt += __shfl_xor(cnt, 32, 64);
    // LDS tree over the 4 waves
    if (wave > 0 && lane < 16) s_p[wave - 1][lane] = a;
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    if (wave == 0 && lane < 16) {
        pool4_merge(a, s_p[0][lane]);
        pool4_merge(a, s_p[1][lane]);
        pool4_merge(a, s_p[2][lane]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            part[((size_t)b * nsplit + sp) * C + c + e] = PoolPart{a.msum[e], a.usum[e], a.mmax[e], a.umax[e], a.am[e], a.au[e]};
        if (blockIdx.x == 0 && lane == 0) cnt_part[b * nsplit + sp] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
}

__global__ __launch_bounds__(256) void pool4_final_kernel(const PoolPart* __restrict__ part,
                                                           const float* __restrict__ cnt_part, int N, int C,
                                                           int nsplit, int ldo, float* __restrict__ nvalid_o,
                                                           float* __restrict__ mmax_o,
                                                           float* __restrict__ mavg_o, float* __restrict__ umean_o,
                                                           float* __restrict__ umax_o, int32_t* __restrict__ arg_m_o,
                                                           int32_t* __restrict__ arg_u_o, float* __restrict__ cnt_o) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    float cnt = 0.f;
    for (int s = 0; s < nsplit; ++s) cnt += cnt_part[b * nsplit + s];
    const float cl = fmaxf(cnt, 1.0f);                      // clamp(min=1), PointNetEncoder.py:86
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        cnt_o[b] = cl;
        if (nvalid_o) nvalid_o[b] = cnt;
    }
    if (c >= C) return;
    float msum = 0.f, usum = 0.f, mmax = -INFINITY, umax = -INFINITY;
    int am = -1, au = -1;
    for (int s = 0; s < nsplit; ++s) {
        const PoolPart q = part[((size_t)b * nsplit + s) * C + c];
        msum += q.msum; usum += q.usum;
        if (q.arg_u >= 0 && (q.umax > umax || au < 0)) { umax = q.umax; au = q.arg_u; }
        if (q.arg_m >= 0 && (q.mmax > mmax || am < 0)) { mmax = q.mmax; am = q.arg_m; }
    }
    const size_t o = (size_t)b * C + c, ov = (size_t)b * ldo + c;
    // where(isfinite(max), max, 0): a cloud with no valid point pools to 0 and passes no gradient
    const bool fin = am >= 0 && isfinite(mmax);
    mmax_o[ov] = fin ? mmax : 0.f;
    arg_m_o[o] = fin ? am : -1;
    mavg_o[ov] = msum / cl;
    umean_o[ov] = usum / (float)N;
    umax_o[ov] = umax;
    arg_u_o[o] = au;
}

// dpf written 16 B per lane: a thread owns 4 channels of one cloud (its 4 pooled cotangents and
// 2 arg-max indices live in registers) and walks the points of its split; 256/(C/4) points of a
// split are written per pass, each a contiguous C*4-byte row.
// SX8: dpf is written as the next GEMMs' split operand (include/wf3d.h, sx8): the two threads that own a group of 8
// channels swap halves, one stores the group's 8 high parts, the other its 8 low parts (wf3d_store_sx8_pair).
template <bool SX8>
__global__ __launch_bounds__(256) void pool4_bwd_kernel(const float* __restrict__ valid, const float* __restrict__ cnt,
                                                         const int32_t* __restrict__ arg_m, const int32_t* __restrict__ arg_u,
                                                         const float* __restrict__ dmmax, const float* __restrict__ dmavg,
                                                         const float* __restrict__ dumean, const float* __restrict__ dumax,
                                                         const float* __restrict__ dpf_direct, int N, int C,
                                                         int npb, int ldm, int ldu, float* __restrict__ dpf) {
    const int b = blockIdx.z;
    const int tpr = C / 4 < 256 ? C / 4 : 256;             // threads per point row
    const int rpb = 256 / tpr;                             // point rows per pass
    const int rsub = threadIdx.x / tpr;
    const int c = (blockIdx.x * tpr + threadIdx.x % tpr) * 4;
    if (c >= C || rsub >= rpb) return;
    const int n0 = blockIdx.y * npb, n1 = min(N, n0 + npb);
    const size_t o = (size_t)b * C + c, om = (size_t)b * ldm + c, ou = (size_t)b * ldu + c;
    const float inv_cnt = 1.0f / cnt[b], inv_n = 1.0f / (float)N;
    f32x4 g_avg = {0.f, 0.f, 0.f, 0.f}, g_mean = g_avg, g_mm = g_avg, g_um = g_avg;
    if (dmavg) g_avg = *reinterpret_cast<const f32x4*>(dmavg + om) * inv_cnt;
    if (dumean) g_mean = *reinterpret_cast<const f32x4*>(dumean + ou) * inv_n;
    if (dmmax) g_mm = *reinterpret_cast<const f32x4*>(dmmax + om);
    if (dumax) g_um = *reinterpret_cast<const f32x4*>(dumax + ou);
    int am[4], au[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { am[j] = arg_m[o + j]; au[j] = arg_u[o + j]; }
    for (int n = n0 + rsub; n < n1; n += rpb) {
        const size_t idx = ((size_t)b * N + n) * C + c;
        const float v = valid[(size_t)b * N + n];
        f32x4 g = g_mean + g_avg * v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (n == am[j]) g[j] += g_mm[j];
            if (n == au[j]) g[j] += g_um[j];
        }
        if (dpf_direct) g += *reinterpret_cast<const f32x4*>(dpf_direct + idx);
        if (SX8) {
            // lanes (2m, 2m+1) hold one sx8 group: one exchange, then 16 contiguous bytes per lane (wf3d_store_sx8_pair)
            const float gv[4] = {g[0], g[1], g[2], g[3]};
            wf3d_store_sx8_pair(dpf + idx, gv, (c & 4) != 0);
        } else {
            *reinterpret_cast<f32x4*>(dpf + idx) = g;
        }
    }
}

// scalar fallback for channel counts that are not a multiple of 4
__global__ __launch_bounds__(256) void pool4_bwd_scalar_kernel(const float* __restrict__ valid, const float* __restrict__ cnt,
                                                                const int32_t* __restrict__ arg_m, const int32_t* __restrict__ arg_u,
                                                                const float* __restrict__ dmmax, const float* __restrict__ dmavg,
                                                                const float* __restrict__ dumean, const float* __restrict__ dumax,
                                                                const float* __restrict__ dpf_direct, int N, int C,
                                                                int npb, int ldm, int ldu, float* __restrict__ dpf) {
    const int b = blockIdx.z;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int n0 = blockIdx.y * npb, n1 = min(N, n0 + npb);
    const size_t o = (size_t)b * C + c, om = (size_t)b * ldm + c, ou = (size_t)b * ldu + c;
    const float g_avg = dmavg ? dmavg[om] / cnt[b] : 0.f;
    const float g_mean = dumean ? dumean[ou] / (float)N : 0.f;
    const float g_mm = dmmax ? dmmax[om] : 0.f;
    const float g_um = dumax ? dumax[ou] : 0.f;
    const int am = arg_m[o], au = arg_u[o];
    for (int n = n0; n < n1; ++n) {
        const size_t idx = ((size_t)b * N + n) * C + c;
        float g = g_mean + valid[(size_t)b * N + n] * g_avg;
        if (n == am) g += g_mm;
        if (n == au) g += g_um;
        if (dpf_direct) g += dpf_direct[idx];
        dpf[idx] = g;
    }
}

// Column sums over all B*N rows of what pool4_bwd writes, from the [B, C] cotangents alone (the bias gradient of the
// Linear in front of the pool):  sum_b  dmavg*nvalid/cnt + dumean + [arg_m >= 0]*dmmax + [arg_u >= 0]*dumax.
__global__ __launch_bounds__(256) void pool4_bwd_bias_kernel(const float* __restrict__ cnt, const float* __restrict__ nvalid,
                                                              const int32_t* __restrict__ arg_m, const int32_t* __restrict__ arg_u,
                                                              const float* __restrict__ dmmax, const float* __restrict__ dmavg,
                                                              const float* __restrict__ dumean, const float* __restrict__ dumax,
                                                              int B, int C, int ldm, int ldu, float* __restrict__ out) {
    // 32 columns x 8 cloud lanes per workgroup (the clouds' loads are independent), folded in LDS in lane order
    __shared__ float s_p[8][32];
    const int cl = threadIdx.x & 31, bg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float acc = 0.f;
    if (c < C)
        for (int b = bg; b < B; b += 8) {
            const size_t o = (size_t)b * C + c, om = (size_t)b * ldm + c, ou = (size_t)b * ldu + c;
            float v = 0.f;
            if (dmavg) v += dmavg[om] * (nvalid[b] / cnt[b]);
            if (dumean) v += dumean[ou];
            if (dmmax && arg_m[o] >= 0) v += dmmax[om];
            if (dumax && arg_u[o] >= 0) v += dumax[ou];
            acc += v;
        }
    s_p[bg][cl] = acc;
    __syncthreads();
    if (bg == 0 && c < C) {
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) a += s_p[g][cl];
        out[c] = a;
    }
}

bool pool_v4(const float* pf, int C) { return C % 64 == 0 && ((uintptr_t)pf % 16) == 0; }

int pool_nsplit(int B, int N, int C) {
    // v4 kernel: C/64 workgroups per (cloud, split), >= 64 rows per wave; fallback: C/256 per (cloud, split)
    const int cb = C % 64 == 0 ? C / 64 : wf3d_cdiv(C, 256);
    static const int tgt = [] { const char* e = getenv("WF3D_POOL_WGS"); return e ? atoi(e) : 2048; }();   // sweep 1024..16384 in situ: 64..71 us, flat (scripts/micro/pool_sweep.sh)
    int ns = (C % 64 == 0 ? tgt : 2048) / (B * cb > 0 ? B * cb : 1);
    const int cap = wf3d_cdiv(N, C % 64 == 0 ? 256 : 32);
    if (ns > cap) ns = cap;
    if (ns < 1) ns = 1;
    return ns;
}

}  // namespace

extern "C" int wf3d_point_valid(const float* x, int M, int in_dim, float* valid, void* stream) {
    WF3D_CHECK(M >= 0 && in_dim > 0, WF3D_ERR_ARG, "wf3d_point_valid: bad dims");
    if (M == 0) return WF3D_OK;
    WF3D_CHECK(x && valid, WF3D_ERR_ARG, "wf3d_point_valid: null pointer");
    hipLaunchKernelGGL(point_valid_kernel, dim3(wf3d_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, x, M, in_dim, valid);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" size_t wf3d_pool4_ws_bytes(int B, int N, int C) {
    if (B <= 0 || N <= 0 || C <= 0) return 0;
    const int ns = pool_nsplit(B, N, C);
    size_t bytes = (size_t)B * ns * C * sizeof(PoolPart);
    bytes = (bytes + 255) & ~(size_t)255;
    return bytes + (size_t)B * ns * sizeof(float);
}

extern "C" int wf3d_pool4_fwd(const float* pf, const float* valid, int B, int N, int C, float* mmax, float* mavg,
                              float* umean, float* umax, int ldo, int32_t* arg_m, int32_t* arg_u, float* cnt,
                              float* nvalid, void* ws, size_t ws_bytes, void* stream) {
    WF3D_CHECK(ldo >= C, WF3D_ERR_ARG, "wf3d_pool4_fwd: ldo=%d < C=%d", ldo, C);
    WF3D_CHECK(B > 0 && N > 0 && C > 0, WF3D_ERR_ARG, "wf3d_pool4_fwd: bad dims B=%d N=%d C=%d", B, N, C);
    WF3D_CHECK(B <= 65535, WF3D_ERR_UNSUPPORTED, "wf3d_pool4_fwd: B > 65535");
    WF3D_CHECK(pf && valid && mmax && mavg && umean && umax && arg_m && arg_u && cnt, WF3D_ERR_ARG, "wf3d_pool4_fwd: null pointer");
    WF3D_CHECK(ws && ws_bytes >= wf3d_pool4_ws_bytes(B, N, C), WF3D_ERR_WS, "wf3d_pool4_fwd: workspace too small");
    const int ns = pool_nsplit(B, N, C);
    const int npb = wf3d_cdiv(N, ns);
    PoolPart* part = (PoolPart*)ws;
    size_t off = ((size_t)B * ns * C * sizeof(PoolPart) + 255) & ~(size_t)255;
    float* cnt_part = (float*)((char*)ws + off);
    hipStream_t st = (hipStream_t)stream;
    if (pool_v4(pf, C))
        hipLaunchKernelGGL(pool4_partial_v4_kernel, dim3(C / 64, ns, B), dim3(256), 0, st, pf, valid, N, C, ns, npb, part, cnt_part);
    else
        hipLaunchKernelGGL(pool4_partial_kernel, dim3(wf3d_cdiv(C, 256), ns, B), dim3(256), 0, st, pf, valid, N, C, ns, npb, part, cnt_part);
    WF3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(pool4_final_kernel, dim3(wf3d_cdiv(C, 256), B), dim3(256), 0, st, part, cnt_part, N, C, ns, ldo, nvalid,
                       mmax, mavg, umean, umax, arg_m, arg_u, cnt);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

static int pool4_bwd_impl(const float* valid, const float* cnt, const int32_t* arg_m, const int32_t* arg_u,
                          const float* dmmax, const float* dmavg, int ldm, const float* dumean, const float* dumax, int ldu,
                          const float* dpf_direct, int B, int N, int C, float* dpf, bool sx8, void* stream);

extern "C" int wf3d_pool4_bwd(const float* valid, const float* cnt, const int32_t* arg_m, const int32_t* arg_u,
                              const float* dmmax, const float* dmavg, int ldm, const float* dumean, const float* dumax,
                              int ldu, const float* dpf_direct, int B, int N, int C, float* dpf, void* stream) {
    return pool4_bwd_impl(valid, cnt, arg_m, arg_u, dmmax, dmavg, ldm, dumean, dumax, ldu, dpf_direct, B, N, C, dpf, false, stream);
}

extern "C" int wf3d_pool4_bwd_sx8(const float* valid, const float* cnt, const int32_t* arg_m, const int32_t* arg_u,
                                  const float* dmmax, const float* dmavg, int ldm, const float* dumean, const float* dumax,
                                  int ldu, const float* dpf_direct, int B, int N, int C, float* dpf_sx8, void* stream) {
    return pool4_bwd_impl(valid, cnt, arg_m, arg_u, dmmax, dmavg, ldm, dumean, dumax, ldu, dpf_direct, B, N, C, dpf_sx8, true, stream);
}

extern "C" int wf3d_pool4_bwd_bias(const float* cnt, const float* nvalid, const int32_t* arg_m, const int32_t* arg_u,
                                   const float* dmmax, const float* dmavg, int ldm, const float* dumean, const float* dumax,
                                   int ldu, int B, int C, float* dbias, void* stream) {
    WF3D_CHECK(B > 0 && C > 0 && cnt && nvalid && arg_m && arg_u && dbias, WF3D_ERR_ARG, "wf3d_pool4_bwd_bias: bad arguments");
    hipLaunchKernelGGL(pool4_bwd_bias_kernel, dim3(wf3d_cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, cnt, nvalid, arg_m,
                       arg_u, dmmax, dmavg, dumean, dumax, B, C, ldm, ldu, dbias);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

static int pool4_bwd_impl(const float* valid, const float* cnt, const int32_t* arg_m, const int32_t* arg_u,
                          const float* dmmax, const float* dmavg, int ldm, const float* dumean, const float* dumax, int ldu,
                          const float* dpf_direct, int B, int N, int C, float* dpf, bool sx8, void* stream) {
    WF3D_CHECK(B > 0 && N > 0 && C > 0 && B <= 65535, WF3D_ERR_ARG, "wf3d_pool4_bwd: bad dims");
    WF3D_CHECK(valid && cnt && arg_m && arg_u && dpf, WF3D_ERR_ARG, "wf3d_pool4_bwd: null pointer");
    const auto al = [](const float* p, int ld) { return !p || ((uintptr_t)p % 16 == 0 && ld % 4 == 0); };
    const bool vec = C % 4 == 0 && ((uintptr_t)dpf % 16 == 0) && (!dpf_direct || (uintptr_t)dpf_direct % 16 == 0) &&
                     al(dmmax, ldm) && al(dmavg, ldm) && al(dumean, ldu) && al(dumax, ldu);
    // (any C / 4: the kernel idles the threads beyond rows-per-pass x C / 4 of a workgroup — PointNetEncoder(output_dim=768)
    // used to be refused here, in the middle of a backward pass)
    WF3D_CHECK(!sx8 || (vec && C % 8 == 0), WF3D_ERR_UNSUPPORTED, "wf3d_pool4_bwd_sx8: needs C %% 8 == 0 and 16-B aligned tensors (C=%d)", C);
    if (vec) {
        int ns = 4096 / (B * wf3d_cdiv(C / 4, 256));
        const int cap = wf3d_cdiv(N, 16);
        ns = ns > cap ? cap : (ns < 1 ? 1 : ns);
        const int npb = wf3d_cdiv(N, ns);
        if (sx8) hipLaunchKernelGGL(pool4_bwd_kernel<true>, dim3(wf3d_cdiv(C / 4, 256), ns, B), dim3(256), 0, (hipStream_t)stream,
                                    valid, cnt, arg_m, arg_u, dmmax, dmavg, dumean, dumax, dpf_direct, N, C, npb, ldm, ldu, dpf);
        else     hipLaunchKernelGGL(pool4_bwd_kernel<false>, dim3(wf3d_cdiv(C / 4, 256), ns, B), dim3(256), 0, (hipStream_t)stream,
                                    valid, cnt, arg_m, arg_u, dmmax, dmavg, dumean, dumax, dpf_direct, N, C, npb, ldm, ldu, dpf);
    } else {
        const int ns = pool_nsplit(B, N, C);
        const int npb = wf3d_cdiv(N, ns);
        hipLaunchKernelGGL(pool4_bwd_scalar_kernel, dim3(wf3d_cdiv(C, 256), ns, B), dim3(256), 0, (hipStream_t)stream,
                           valid, cnt, arg_m, arg_u, dmmax, dmavg, dumean, dumax, dpf_direct, N, C, npb, ldm, ldu, dpf);
    }
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}
