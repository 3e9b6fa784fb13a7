// Linear layers over a HANDFUL OF ROWS (M <= 32: one row per cloud of the batch) — the feature-fusion MLP
// (models/PointNetEncoder.py:57-65,116) and the whole vertex head (models/VertexPredictor.py:94-117) — forward
// and backward.  With M = B rows these layers are pure weight streaming (93 MB of fp32 weights forward, the same
// again for dgrad, and 93 MB of weight gradients written): the job is to keep every CU pulling 16-B/lane loads and
// to spend as few launches as possible, not to feed the MFMA pipe.  The generic GEMM path took ~150 launches of
// 5-15 us for them (GEMM + split-K reduce + row statistics + LayerNorm apply + column sums, each its own kernel).
//
// Three kernels, all on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2_f32: an fp32 FMA chain):
//
//  skinny_fwd_kernel   Y = pro(X)·W^T + b (+ addend): one workgroup of 16 waves per 16 output columns, the waves
//                      split K (each lane streams 32 contiguous bytes of its weight row per step, 64 KB of weight
//                      loads in flight per CU), LDS reduction, bias / residual epilogue.  pro(X) =
//                      act(LayerNorm(X)) + addend is applied on load; the LayerNorm statistics of X are MERGED
//                      from per-16-column (mean, M2) partials that the producer of X wrote in its epilogue
//                      (Chan's parallel variance), so no statistics pass and no normalised copy ever exists.
//                      Up to 4 problems per launch (the three Linears that read the same vector e).
//  skinny_bwd_kernel   per Linear, in one launch: wgrad dW = dz^T·pro(X) (64 x 256 tiles, 32x32x2 MFMA over the M
//                      rows, dW streamed out), bias gradient, and dgrad partial slabs dX_s = dz[:, chunk s]·W[chunk s, :]
//                      (each lane 16 B of a weight row: the 4 values feed 4 MFMAs whose output columns interleave).
//                      dz is rebuilt on load from G = dh*act'*gamma, the saved pre-activation and the row sums
//                      (sum G, sum G*xhat), merged from per-64-column partials: the LayerNorm backward never runs
//                      as a kernel of its own.
//  skinny_reduce_kernel  sums the dgrad slabs in a fixed order (deterministic, no atomics), adds up to three slab sets
//                      (a vector with several consumers), and applies the elementwise half of the NEXT LayerNorm /
//                      activation backward: writes G, dgamma, dbeta and the per-64-column row-sum partials.
#include "wf3d_common.h"

namespace {

constexpr int SK_MAXP = 4;      // problems per grouped launch (mirrors WF3D_SKINNY_MAX_PROBLEMS)
constexpr int SK_FWD_WAVES = 16;
constexpr int SK_FWD_MAXK = 4096;  // LayerNorm prologue: affine vectors staged in LDS (32 KB); wider inputs read them from L2

// LayerNorm value with a FIXED operation order: forward prologue, wgrad prologue and the mask in the reduce kernel
// must take the same ReLU decision for the same element, so no site may be contracted differently by the compiler.
__device__ __forceinline__ float sk_xhat(float x, float mu, float rs) { return __fmul_rn(__fsub_rn(x, mu), rs); }
__device__ __forceinline__ float sk_ln(float x, float mu, float rs, float g, float b) { return __fmaf_rn(sk_xhat(x, mu, rs), g, b); }

struct FwdParams {
    wf3d_skinny_fwd_t p[SK_MAXP];
    int blk_begin[SK_MAXP + 1];
    int nprob, M;
    float eps;
};

template <int MT>
__global__ __launch_bounds__(1024) void skinny_fwd_kernel(const FwdParams P) {
    __shared__ float s_mu[32], s_rs[32];
    __shared__ float s_red[SK_FWD_WAVES][MT * 4][64];
    __shared__ __attribute__((aligned(16))) float s_gb[2][SK_FWD_MAXK];   // gamma, beta of the whole reduction range
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < SK_MAXP; ++i)
        if (i < P.nprob && (int)blockIdx.x >= P.blk_begin[i]) pi = i;
    const wf3d_skinny_fwd_t& q = P.p[pi];
    const int blk = blockIdx.x - P.blk_begin[pi];
    const int M = P.M, N = q.N, K = q.K;
    const bool ln = q.gamma != nullptr;

    const int c16 = lane & 15, q4 = lane >> 4;
    const int n0 = blk * 16;
    const int n = min(n0 + c16, N - 1);
    const float* wrow = q.W + (size_t)n * q.ldw;
    int xr[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) xr[t] = min(16 * t + c16, M - 1);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // software pipeline: the raw loads of step i+1 (weights, activations, affine) are issued before step i's
    // LayerNorm math and MFMAs, so every wave keeps two steps of 16-B loads in flight
    struct Raw { f32x4 w0, w1, x0[MT], x1[MT]; };                  // the two streams that come from HBM / L2
    auto load_raw = [&](int ch, Raw& r) {
        const int k = ch * 32 + 8 * q4;
        const bool v0 = k < K, v1 = k + 4 < K;                   // K % 4 == 0; beyond K everything is 0
        r.w0 = v0 ? *reinterpret_cast<const f32x4*>(wrow + k) : zero4;
        r.w1 = v1 ? *reinterpret_cast<const f32x4*>(wrow + k + 4) : zero4;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const float* xp = q.X + (size_t)xr[t] * q.ldx + k;
            r.x0[t] = v0 ? *reinterpret_cast<const f32x4*>(xp) : zero4;
            r.x1[t] = v1 ? *reinterpret_cast<const f32x4*>(xp + 4) : zero4;
        }
    };
    Raw cur;                                                     // first step's loads fly under the statistics merge
    if (wave * 32 < K) load_raw(wave, cur);
    const bool gb_lds = ln && K <= SK_FWD_MAXK;
    if (gb_lds)
        for (int k = tid * 4; k < K; k += 4096) {
            *reinterpret_cast<f32x4*>(&s_gb[0][k]) = *reinterpret_cast<const f32x4*>(q.gamma + k);
            *reinterpret_cast<f32x4*>(&s_gb[1][k]) = *reinterpret_cast<const f32x4*>(q.beta + k);
        }
    if (ln) {
        if (q.in_part) {
            // merge the producer's per-16-column (mean, M2) partials: 32 threads per row
            const int row = tid >> 5, j = tid & 31, nb = q.in_nblk;
            float sm = 0.f;
            if (row < M)
                for (int p = j; p < nb; p += 32) sm += q.in_part[((size_t)p * M + row) * 2];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
            const float mean = sm / (float)nb;
            float m2 = 0.f;
            if (row < M)
                for (int p = j; p < nb; p += 32) {
                    const float d = q.in_part[((size_t)p * M + row) * 2] - mean;
                    m2 += q.in_part[((size_t)p * M + row) * 2 + 1] + 16.0f * d * d;
                }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) m2 += __shfl_xor(m2, o, 64);
            if (j == 0 && row < M) {
                const float rs = 1.0f / sqrtf(m2 / (16.0f * (float)nb) + P.eps);
                s_mu[row] = mean;
                s_rs[row] = rs;
                if (blk == 0 && q.mu_out) { q.mu_out[row] = mean; q.rs_out[row] = rs; }
            }
        } else if (tid < M) {
            s_mu[tid] = q.mu[tid];
            s_rs[tid] = q.rs[tid];
        }
        __syncthreads();
    }

    f32x4 acc[MT];
    float rmu[MT], rrs[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        rmu[t] = ln ? s_mu[xr[t]] : 0.f;
        rrs[t] = ln ? s_rs[xr[t]] : 1.f;
    }
    for (int ch = wave; ch * 32 < K; ch += SK_FWD_WAVES) {
        Raw nxt;
        const bool more = (ch + SK_FWD_WAVES) * 32 < K;
        if (more) load_raw(ch + SK_FWD_WAVES, nxt);
        const int k = ch * 32 + 8 * q4;
        const bool v0 = k < K, v1 = k + 4 < K;
        f32x4 g0 = zero4, g1 = zero4, b0 = zero4, b1 = zero4;
        if (gb_lds) {
            if (v0) { g0 = *reinterpret_cast<const f32x4*>(&s_gb[0][k]); b0 = *reinterpret_cast<const f32x4*>(&s_gb[1][k]); }
            if (v1) { g1 = *reinterpret_cast<const f32x4*>(&s_gb[0][k + 4]); b1 = *reinterpret_cast<const f32x4*>(&s_gb[1][k + 4]); }
        } else if (ln) {
            if (v0) { g0 = *reinterpret_cast<const f32x4*>(q.gamma + k); b0 = *reinterpret_cast<const f32x4*>(q.beta + k); }
            if (v1) { g1 = *reinterpret_cast<const f32x4*>(q.gamma + k + 4); b1 = *reinterpret_cast<const f32x4*>(q.beta + k + 4); }
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            f32x4 x0 = cur.x0[t], x1 = cur.x1[t];
            if (ln) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x0[e] = wf3d_act_rt(q.act, sk_ln(x0[e], rmu[t], rrs[t], g0[e], b0[e]));
                    x1[e] = wf3d_act_rt(q.act, sk_ln(x1[e], rmu[t], rrs[t], g1[e], b1[e]));
                }
            }
            if (q.in_addend) {
                const float* ap = q.in_addend + (size_t)xr[t] * q.ld_in_addend + k;
                if (v0) x0 += *reinterpret_cast<const f32x4*>(ap);
                if (v1) x1 += *reinterpret_cast<const f32x4*>(ap + 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[e], cur.w0[e], acc[t], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[e], cur.w1[e], acc[t], 0, 0, 0);
        }
        if (more) cur = nxt;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_red[wave][t * 4 + r][lane] = acc[t][r];
    __syncthreads();
    if (tid < MT * 256) {
        const int v = tid >> 6;                       // t * 4 + reg
        float y = 0.f;
#pragma unroll
        for (int w = 0; w < SK_FWD_WAVES; ++w) y += s_red[w][v][lane];
        const int row = 16 * (v >> 2) + 4 * q4 + (v & 3);
        const int col = n0 + c16;
        const bool ok = row < M && col < N;
        if (q.bias && col < N) y += q.bias[col];
        if (q.out_addend && ok) y += q.out_addend[(size_t)row * q.ld_out_addend + col];
        if (ok) q.Y[(size_t)row * q.ldy + col] = y;
        if (q.stat_part) {                            // N % 16 == 0 guaranteed by the host
            float s = y;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mb = s * 0.0625f;
            const float d = y - mb;
            float m2 = d * d;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) m2 += __shfl_xor(m2, o, 64);
            if (c16 == 0 && row < M) {
                q.stat_part[((size_t)blk * M + row) * 2] = mb;
                q.stat_part[((size_t)blk * M + row) * 2 + 1] = m2;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward: wgrad tiles + dgrad slabs of up to SK_MAXP Linears per launch
// ---------------------------------------------------------------------------------------------------------
struct BwdParams {
    wf3d_skinny_bwd_t p[SK_MAXP];
    int blk_wgrad[SK_MAXP];      // first workgroup of the problem's wgrad tiles
    int blk_dgrad[SK_MAXP];      // first workgroup of its dgrad tiles
    int blk_end[SK_MAXP];
    int nprob, M;
};

__device__ __forceinline__ bool aligned16d(const void* p) { return ((uintptr_t)p & 15) == 0; }

constexpr int DZ_PITCH = 33;     // s_dz[n][m]: conflict-free for both operand read patterns

// dz[m][n0 + nl] for nl < ncount into s_dz[nl][m] (zero for n >= N, m >= M)
__device__ __forceinline__ void sk_load_dz(const wf3d_skinny_bwd_t& q, int M, int n0, int ncount, float (*s_dz)[DZ_PITCH],
                                           float (*s_c)[2]) {
    const int tid = threadIdx.x;
    const bool ln = q.z != nullptr;
    if (ln) {
        {   // 8 threads per row, each sums every 8th partial of both row sums; xor-shuffles stay inside the 8 lanes
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const int row = tid >> 3, j = tid & 7;
            f32x2 a = {0.f, 0.f};
            if (row < M)
                for (int p = j; p < q.rowpart_nblk; p += 8) a += *reinterpret_cast<const f32x2*>(q.rowpart + ((size_t)p * M + row) * 2);
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) { a[0] += __shfl_xor(a[0], o, 64); a[1] += __shfl_xor(a[1], o, 64); }
            if (j == 0) { s_c[row][0] = a[0] / (float)q.N; s_c[row][1] = a[1] / (float)q.N; }
        }
        __syncthreads();
    }
    if (q.N % 4 == 0 && q.lddy % 4 == 0 && (!ln || q.ldz % 4 == 0) && aligned16d(q.dY) && (!ln || aligned16d(q.z))) {
        // 16 B per lane, all of a thread's loads independent: thread = (4 columns, row m0 + rstep*i)
        const int ncq = ncount >> 2;                       // float4 per row: 16 (wgrad tile) or <= 32 (dgrad chunk)
        const int n4 = tid % ncq, m0 = tid / ncq, rstep = 256 / ncq;
        const int n = n0 + 4 * n4;
        if (m0 < rstep) {
#pragma unroll 4
            for (int m = m0; m < 32; m += rstep) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (m < M && n < q.N) {
                    v = *reinterpret_cast<const f32x4*>(q.dY + (size_t)m * q.lddy + n);
                    if (ln) {
                        const f32x4 zz = *reinterpret_cast<const f32x4*>(q.z + (size_t)m * q.ldz + n);
                        const float rs = q.rs[m], mu = q.mu[m], c1 = s_c[m][0], c2 = s_c[m][1];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = rs * (v[e] - c1 - sk_xhat(zz[e], mu, rs) * c2);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) s_dz[4 * n4 + e][m] = v[e];
            }
        }
        return;
    }
    for (int idx = tid; idx < 32 * ncount; idx += 256) {
        const int m = idx / ncount, nl = idx - m * ncount, n = n0 + nl;
        float v = 0.f;
        if (m < M && n < q.N) {
            const float g = q.dY[(size_t)m * q.lddy + n];
            if (ln) {
                const float rs = q.rs[m];
                const float xh = sk_xhat(q.z[(size_t)m * q.ldz + n], q.mu[m], rs);
                v = rs * (g - s_c[m][0] - xh * s_c[m][1]);
            } else {
                v = g;
            }
        }
        s_dz[nl][m] = v;
    }
}

__global__ __launch_bounds__(256) void skinny_bwd_kernel(const BwdParams P) {
    __shared__ float s_dz[128][DZ_PITCH];
    __shared__ __attribute__((aligned(16))) float s_a[32][256];
    __shared__ float s_c[32][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < SK_MAXP; ++i)
        if (i < P.nprob && (int)blockIdx.x >= P.blk_wgrad[i]) pi = i;
    const wf3d_skinny_bwd_t& q = P.p[pi];
    const int M = P.M, N = q.N, K = q.K;
    const int c = lane & 31, h = lane >> 5;

    if ((int)blockIdx.x < P.blk_dgrad[pi]) {
        // ---------------- wgrad tile: dW[n0:n0+64, k0:k0+256] = dz^T · pro(X), db ----------------
        const int t = blockIdx.x - P.blk_wgrad[pi];
        const int ktiles = (K + 255) / 256;
        const int nt = t / ktiles, kt = t - nt * ktiles;
        const int n0 = nt * 64, k0 = kt * 256;
        sk_load_dz(q, M, n0, 64, s_dz, s_c);
        const bool ln = q.xgamma != nullptr;
        const bool vec = q.ldx % 4 == 0 && aligned16d(q.X) && (!q.xadd || (q.ldxadd % 4 == 0 && aligned16d(q.xadd))) &&
                         (!ln || (aligned16d(q.xgamma) && aligned16d(q.xbeta)));
        if (vec) {
            // thread = 4 columns x 8 rows: the affine pair is loaded once, the 8 row loads are independent
            const int k4 = tid & 63, m0 = tid >> 6, k = k0 + 4 * k4;
            const bool kv = k < K;                                  // K % 4 == 0
            f32x4 g = {0.f, 0.f, 0.f, 0.f}, b = g;
            if (ln && kv) { g = *reinterpret_cast<const f32x4*>(q.xgamma + k); b = *reinterpret_cast<const f32x4*>(q.xbeta + k); }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m0 + 4 * i;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (m < M && kv) {
                    v = *reinterpret_cast<const f32x4*>(q.X + (size_t)m * q.ldx + k);
                    if (ln) {
                        const float mu = q.xmu[m], rs = q.xrs[m];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = wf3d_act_rt(q.xact, sk_ln(v[e], mu, rs, g[e], b[e]));
                    }
                    if (q.xadd) v += *reinterpret_cast<const f32x4*>(q.xadd + (size_t)m * q.ldxadd + k);
                }
                *reinterpret_cast<f32x4*>(&s_a[m][4 * k4]) = v;
            }
        } else
        for (int idx = tid; idx < 32 * 256; idx += 256) {
            const int m = idx >> 8, kk = idx & 255, k = k0 + kk;
            float v = 0.f;
            if (m < M && k < K) {
                v = q.X[(size_t)m * q.ldx + k];
                if (ln) v = wf3d_act_rt(q.xact, sk_ln(v, q.xmu[m], q.xrs[m], q.xgamma[k], q.xbeta[k]));
                if (q.xadd) v += q.xadd[(size_t)m * q.ldxadd + k];
            }
            s_a[m][kk] = v;
        }
        __syncthreads();
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int kb = 64 * wave;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const int m = 2 * s + h;
            const float a0 = s_dz[c][m], a1 = s_dz[32 + c][m];
            const float b0 = s_a[m][kb + c], b1 = s_a[m][kb + 32 + c];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = k0 + kb + 32 * j + c;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nn = n0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (nn < N && k < K) q.dW[(size_t)nn * q.lddw + k] = acc[i][j][r];
                }
            }
        if (kt == 0 && q.db && tid < 64 && n0 + tid < N) {
            float s = 0.f;
            for (int m = 0; m < M; ++m) s += s_dz[tid][m];
            q.db[n0 + tid] = s;
        }
        return;
    }
    // ---------------- dgrad tile: slab[s][:, k-range] = dz[:, chunk s] · W[chunk s, k-range] ----------------
    {
        const int t = blockIdx.x - P.blk_dgrad[pi];
        const int kranges = (K + 511) / 512;
        const int s = t / kranges, kr = t - s * kranges;
        const int nc = q.nc, n0 = s * nc;
        sk_load_dz(q, M, n0, nc, s_dz, s_c);
        __syncthreads();
        const int kq = kr * 512 + 128 * wave + 4 * c;
        const bool kval = kq < K;                                  // K % 4 == 0
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const float* wp = q.W + (size_t)(n0 + h) * q.ldw + kq;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        constexpr int U = 8;                                       // 8 x 1 KB of weight rows per wave per batch, two batches in flight
        const int nj = nc / 2;
        auto load_w = [&](int j0, f32x4* w) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u;
                w[u] = (kval && j < nj && n0 + 2 * j + h < N) ? *reinterpret_cast<const f32x4*>(wp + (size_t)(2 * j) * q.ldw) : zero4;
            }
        };
        f32x4 wc[U], wn[U];
        load_w(0, wc);
        for (int j0 = 0; j0 < nj; j0 += U) {
            if (j0 + U < nj) load_w(j0 + U, wn);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u;
                const float a = j < nj ? s_dz[2 * j + h][c] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wc[u][i], acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) wc[u] = wn[u];
        }
        if (kval) {
            float* out = q.slabs + (size_t)s * M * K + kq;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < M) *reinterpret_cast<f32x4*>(out + (size_t)m * K) = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// slab reduction + elementwise half of the LayerNorm / activation backward of the stage that produced this vector
// ---------------------------------------------------------------------------------------------------------
// One workgroup of 16 waves per 64 columns; lane = column.  Phase 1: wave w sums the slabs s = w, w+16, ... of every
// row into LDS partials (independent loads: latency-tolerant), phase 2: the 16 partials are folded in wave order
// (deterministic), the LayerNorm half applied, row sums taken per wave, column sums folded over the waves.
__global__ __launch_bounds__(1024) void skinny_reduce_kernel(const wf3d_skinny_red_t q, int M) {
    __shared__ float s_v[16][32][64];          // [wave][row][col] 128 KB
    __shared__ float s_g[16][64], s_b[16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = blockIdx.x * 64 + lane, K = q.K;
    const bool cv = col < K;
    const bool ln = q.z != nullptr;
    // eight rows at a time (sixteen ran 2.5x slower): the eight loads of a slab are independent and go out back to back (one row at a time is a
    // chain of M x slabs/16 dependent round trips)
    for (int m0 = 0; m0 < M; m0 += 8) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (cv) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
                for (int s = wave; s < q.nslab[i]; s += 16) {
                    const float* base = q.slabs[i] + ((size_t)s * M + m0) * K + col;
                    float t[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) t[j] = m0 + j < M ? base[(size_t)j * K] : 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += t[j];
                }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (m0 + j < M) s_v[wave][m0 + j][lane] = v[j];
    }
    __syncthreads();
    const float gam = (ln && cv) ? q.gamma[col] : 0.f, bet = (ln && cv) ? q.beta[col] : 0.f;
    float dg = 0.f, dbt = 0.f;
    for (int m = wave; m < M; m += 16) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += s_v[w][m][lane];
        if (cv) {
            if (q.extra) v += q.extra[(size_t)m * q.ldextra + col];
            if (q.dh) q.dh[(size_t)m * q.lddh + col] = v;
        }
        if (ln) {
            float G = 0.f, Gx = 0.f;
            if (cv) {
                const float mu = q.mu[m], rs = q.rs[m];
                const float xh = sk_xhat(q.z[(size_t)m * q.ldz + col], mu, rs);
                const float y = __fmaf_rn(xh, gam, bet);
                const float d = v * wf3d_act_grad_rt(q.act, y);
                dg += d * xh;
                dbt += d;
                G = d * gam;
                Gx = G * xh;
                q.G[(size_t)m * q.ldg + col] = G;
            }
            const float sG = wf3d_wave_sum(G), sGx = wf3d_wave_sum(Gx);
            if (lane == 0) {
                q.rowpart[((size_t)blockIdx.x * M + m) * 2] = sG;
                q.rowpart[((size_t)blockIdx.x * M + m) * 2 + 1] = sGx;
            }
        }
    }
    if (ln) {
        s_g[wave][lane] = dg;
        s_b[wave][lane] = dbt;
        __syncthreads();
        if (wave == 0 && cv) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) { a += s_g[w][lane]; b += s_b[w][lane]; }
            q.dgamma[col] = a;
            q.dbeta[col] = b;
        }
    }
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int wf3d_skinny_ok(int M, int K) { return M >= 1 && M <= WF3D_SKINNY_MAX_ROWS && K >= 4 && K % 4 == 0; }

extern "C" int wf3d_skinny_fwd(const wf3d_skinny_fwd_t* probs, int nprob, int M, float eps, void* stream) {
    WF3D_CHECK(probs && nprob >= 1 && nprob <= SK_MAXP, WF3D_ERR_ARG, "wf3d_skinny_fwd: 1..%d problems per launch", SK_MAXP);
    WF3D_CHECK(M >= 1 && M <= WF3D_SKINNY_MAX_ROWS, WF3D_ERR_UNSUPPORTED, "wf3d_skinny_fwd: M=%d rows (max %d)", M, WF3D_SKINNY_MAX_ROWS);
    FwdParams P;
    P.nprob = nprob; P.M = M; P.eps = eps;
    int blk = 0;
    for (int i = 0; i < nprob; ++i) {
        const wf3d_skinny_fwd_t& q = probs[i];
        WF3D_CHECK(q.X && q.W && q.Y && q.N >= 1 && q.K >= 4 && q.K % 4 == 0, WF3D_ERR_ARG, "wf3d_skinny_fwd[%d]: bad problem (N=%d K=%d)", i, q.N, q.K);
        WF3D_CHECK(q.ldx % 4 == 0 && q.ldw % 4 == 0 && aligned16(q.X) && aligned16(q.W), WF3D_ERR_UNSUPPORTED,
                   "wf3d_skinny_fwd[%d]: X / W rows must be 16-byte aligned", i);
        if (q.gamma) {
            WF3D_CHECK(q.beta && aligned16(q.gamma) && aligned16(q.beta), WF3D_ERR_ARG, "wf3d_skinny_fwd[%d]: gamma/beta", i);
            WF3D_CHECK((q.in_part && q.in_nblk * 16 == q.K) || (q.mu && q.rs), WF3D_ERR_ARG,
                       "wf3d_skinny_fwd[%d]: LayerNorm prologue needs statistics partials covering K or (mu, rs)", i);
        }
        if (q.in_addend) WF3D_CHECK(q.ld_in_addend % 4 == 0 && aligned16(q.in_addend), WF3D_ERR_UNSUPPORTED, "wf3d_skinny_fwd[%d]: in_addend alignment", i);
        WF3D_CHECK(!q.stat_part || q.N % 16 == 0, WF3D_ERR_UNSUPPORTED, "wf3d_skinny_fwd[%d]: statistics partials need N %% 16 == 0", i);
        P.p[i] = q;
        P.blk_begin[i] = blk;
        blk += wf3d_cdiv(q.N, 16);
    }
    for (int i = nprob; i <= SK_MAXP; ++i) P.blk_begin[i] = blk;
    if (M <= 16) hipLaunchKernelGGL(skinny_fwd_kernel<1>, dim3(blk), dim3(1024), 0, (hipStream_t)stream, P);
    else         hipLaunchKernelGGL(skinny_fwd_kernel<2>, dim3(blk), dim3(1024), 0, (hipStream_t)stream, P);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

static int sk_nslab(const wf3d_skinny_bwd_t& q) { return wf3d_cdiv(q.N, q.nc); }

extern "C" size_t wf3d_skinny_slab_floats(int M, int N, int K, int nc) { return (size_t)wf3d_cdiv(N, nc) * M * K; }

extern "C" int wf3d_skinny_bwd(const wf3d_skinny_bwd_t* probs, int nprob, int M, void* stream) {
    WF3D_CHECK(probs && nprob >= 1 && nprob <= SK_MAXP, WF3D_ERR_ARG, "wf3d_skinny_bwd: 1..%d problems per launch", SK_MAXP);
    WF3D_CHECK(M >= 1 && M <= WF3D_SKINNY_MAX_ROWS, WF3D_ERR_UNSUPPORTED, "wf3d_skinny_bwd: M=%d rows (max %d)", M, WF3D_SKINNY_MAX_ROWS);
    BwdParams P;
    P.nprob = nprob; P.M = M;
    int blk = 0;
    for (int i = 0; i < nprob; ++i) {
        const wf3d_skinny_bwd_t& q = probs[i];
        WF3D_CHECK(q.dY && q.W && q.N >= 1 && q.K >= 4 && q.K % 4 == 0, WF3D_ERR_ARG, "wf3d_skinny_bwd[%d]: bad problem", i);
        WF3D_CHECK(!q.z || (q.mu && q.rs && q.rowpart && q.rowpart_nblk >= 1), WF3D_ERR_ARG, "wf3d_skinny_bwd[%d]: LayerNorm form needs mu, rs, rowpart", i);
        WF3D_CHECK(!q.dW || q.X, WF3D_ERR_ARG, "wf3d_skinny_bwd[%d]: wgrad needs X", i);
        WF3D_CHECK(!q.xgamma || (q.xbeta && q.xmu && q.xrs), WF3D_ERR_ARG, "wf3d_skinny_bwd[%d]: X prologue needs mu, rs, gamma, beta", i);
        if (q.slabs) {
            WF3D_CHECK(q.nc >= 2 && q.nc <= 128 && q.nc % 2 == 0, WF3D_ERR_ARG, "wf3d_skinny_bwd[%d]: nc must be even, 2..128", i);
            WF3D_CHECK(q.ldw % 4 == 0 && aligned16(q.W) && aligned16(q.slabs), WF3D_ERR_UNSUPPORTED, "wf3d_skinny_bwd[%d]: W / slab alignment", i);
        }
        P.p[i] = q;
        P.blk_wgrad[i] = blk;
        if (q.dW) blk += wf3d_cdiv(q.N, 64) * wf3d_cdiv(q.K, 256);
        P.blk_dgrad[i] = blk;
        if (q.slabs) blk += sk_nslab(q) * wf3d_cdiv(q.K, 512);
        P.blk_end[i] = blk;
    }
    for (int i = nprob; i < SK_MAXP; ++i) P.blk_wgrad[i] = P.blk_dgrad[i] = P.blk_end[i] = blk;
    if (blk == 0) return WF3D_OK;
    hipLaunchKernelGGL(skinny_bwd_kernel, dim3(blk), dim3(256), 0, (hipStream_t)stream, P);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_skinny_reduce(const wf3d_skinny_red_t* prob, int M, void* stream) {
    WF3D_CHECK(prob && M >= 1 && M <= WF3D_SKINNY_MAX_ROWS, WF3D_ERR_ARG, "wf3d_skinny_reduce: bad arguments");
    const wf3d_skinny_red_t& q = *prob;
    WF3D_CHECK(q.K >= 1, WF3D_ERR_ARG, "wf3d_skinny_reduce: K");
    for (int i = 0; i < 3; ++i) WF3D_CHECK(q.nslab[i] == 0 || q.slabs[i], WF3D_ERR_ARG, "wf3d_skinny_reduce: slab set %d", i);
    WF3D_CHECK(!q.z || (q.mu && q.rs && q.gamma && q.beta && q.G && q.dgamma && q.dbeta && q.rowpart), WF3D_ERR_ARG,
               "wf3d_skinny_reduce: LayerNorm epilogue needs mu, rs, gamma, beta, G, dgamma, dbeta, rowpart");
    hipLaunchKernelGGL(skinny_reduce_kernel, dim3(wf3d_cdiv(q.K, 64)), dim3(1024), 0, (hipStream_t)stream, q, M);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}
