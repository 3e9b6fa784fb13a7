// MFMA self-attention for head_dim = 64 (the reference's 512/8), exact-fp32 matrix cores
// (v_mfma_f32_32x32x2_f32), one workgroup per (head, sample), K/V (or Q/dO) rows of the sample
// resident in LDS, each wave64 owning 32-row query (or key) blocks.
//
// Orientation trick: the score tile is computed TRANSPOSED, S^T = K·Q^T, so that the softmax axis
// (keys) runs over a lane's accumulator registers (+ one exchange with lane^32) and never across
// lanes, and the probability tile is already the B operand of the next contraction
// O^T = V^T·P^T, which sums over the accumulator's ROW index (cdna_hip_programming.md §3 "An
// accumulator tile as the next MFMA's operand"): MFMA step e pairs k = row(e, half 0) with
// row(e, half 1), i.e. B = acc register e as it stands; A = V[row(e, half)][d] is one conflict-free
// ds_read_b32.  The reduction index d of QK^T is assigned as d = 32*(lane>>5) + step so a lane's
// operand is 32 contiguous floats of its row.
//
// Backward = two sweeps with the same LDS footprint (no atomics): sweep 1 per query block
// (S^T, dP^T = V·dO^T, dS^T, dQ^T = K^T·dS^T), sweep 2 per key block with the roles of rows and
// columns swapped (S = Q·K^T, dP = dO·V^T, dK^T = Q^T·dS, dV^T = dO^T·P).  delta_i = dO_i·O_i.
#include "wf3d_common.h"

namespace {

constexpr int HD = 64;
constexpr int LDR = HD + 4;                 // padded LDS row (floats): 16 lanes of a b128 group hit 16 slots

struct MAttn {
    const float* qkv; const float* ctx_in; float* ctx; float* lse; const float* dctx; float* dqkv;
    const int32_t* voff;
    int E, heads, vpad;                     // vpad = rows reserved per operand in LDS (multiple of 32)
    float scale;
    uint32_t seed, thresh; float dscale;
};

__device__ __forceinline__ int rowof(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// stage rows [r0, r0+n) x 64 columns (col0..) of src[*, ld] into dst[npad][LDR], zero rows >= n
__device__ __forceinline__ void stage64(float* dst, const float* __restrict__ src, int ld, int r0, int n, int npad,
                                        int col0, float mul) {
    for (int idx = threadIdx.x; idx < npad * 16; idx += blockDim.x) {
        const int r = idx >> 4, c = (idx & 15) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < n) v = *reinterpret_cast<const f32x4*>(src + (size_t)(r0 + r) * ld + col0 + c) * mul;
        *reinterpret_cast<f32x4*>(dst + r * LDR + c) = v;
    }
}

// 32 contiguous floats (this lane's half of a 64-wide row) from global, zero when !ok
__device__ __forceinline__ void load_half_row(float (&f)[32], const float* __restrict__ p, bool ok, float mul) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4*>(p + 4 * k);
#pragma unroll
        for (int j = 0; j < 4; ++j) f[4 * k + j] = v[j] * mul;
    }
}

// acc[32x32] += A·B with A rows read from LDS (row = arow, this lane's half), B = 32 register values
__device__ __forceinline__ f32x16 mm_rows_regs(const float* lds_row_half, const float (&b)[32]) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(lds_row_half + 4 * k);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[4 * k + j], acc, 0, 0, 0);
    }
    return acc;
}

__global__ __launch_bounds__(256, 1) void attn_fwd_mfma_kernel(const MAttn p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int head = blockIdx.x, s = blockIdx.y;
    const int r0 = p.voff[s], n = p.voff[s + 1] - r0;
    if (n <= 0) return;
    const int npad = (n + 31) & ~31, nt = npad >> 5;
    const int E3 = 3 * p.E;
    float* Ks = sm;
    float* Vs = sm + p.vpad * LDR;
    stage64(Ks, p.qkv, E3, r0, n, npad, p.E + head * HD, 1.0f);
    stage64(Vs, p.qkv, E3, r0, n, npad, 2 * p.E + head * HD, 1.0f);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    for (int qb = wave; qb < nt; qb += 4) {
        const int i = qb * 32 + l31;                       // this lane's query
        const bool iok = i < n;
        float qf[32];
        load_half_row(qf, p.qkv + (size_t)(r0 + i) * E3 + head * HD + 32 * h, iok, p.scale);
        f32x16 st[8];                                      // S^T tiles: rows = keys, column = this lane's query
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) {
            if (jt < nt) {
                st[jt] = mm_rows_regs(Ks + (jt * 32 + l31) * LDR + 32 * h, qf);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (jt * 32 + rowof(e, h) >= n) st[jt][e] = -INFINITY;
                    m = fmaxf(m, st[jt][e]);
                }
            }
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) {
            if (jt < nt) {
#pragma unroll
                for (int e = 0; e < 16; ++e) { st[jt][e] = expf(st[jt][e] - m); l += st[jt][e]; }
            }
        }
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        if (iok && h == 0) p.lse[(size_t)(r0 + i) * p.heads + head] = m + logf(l);
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) {
            if (jt < nt) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int j = jt * 32 + rowof(e, h);
                    float pv = st[jt][e] * inv;
                    if (p.thresh) pv = wf3d_keep(p.seed, (uint32_t)((r0 + i) * p.heads + head), (uint32_t)j, p.thresh) ? pv * p.dscale : 0.f;
                    const float* vr = Vs + j * LDR + l31;
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[0], pv, o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32], pv, o[1], 0, 0, 0);
                }
            }
        }
        if (iok) {
            float* out = p.ctx + (size_t)(r0 + i) * p.E + head * HD;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(out + dt * 32 + 8 * g + 4 * h) = v;
                }
        }
    }
}

__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(const MAttn p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int head = blockIdx.x, s = blockIdx.y;
    const int r0 = p.voff[s], n = p.voff[s + 1] - r0;
    if (n <= 0) return;
    const int npad = (n + 31) & ~31, nt = npad >> 5;
    const int E3 = 3 * p.E;
    float* Xa = sm;                                        // sweep 1: K        sweep 2: scaled Q
    float* Xb = sm + p.vpad * LDR;                         // sweep 1: V        sweep 2: dO
    float* dl = Xb + p.vpad * LDR;                         // [vpad] delta_i
    float* ls = dl + p.vpad;                               // [vpad] lse_i (+inf for padded queries)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;

    stage64(Xa, p.qkv, E3, r0, n, npad, p.E + head * HD, 1.0f);
    stage64(Xb, p.qkv, E3, r0, n, npad, 2 * p.E + head * HD, 1.0f);
    for (int i = threadIdx.x; i < npad; i += blockDim.x)
        ls[i] = i < n ? p.lse[(size_t)(r0 + i) * p.heads + head] : INFINITY;
    __syncthreads();
    // ---- sweep 1: per query block -> dQ, delta
    for (int qb = wave; qb < nt; qb += 4) {
        const int i = qb * 32 + l31;
        const bool iok = i < n;
        float qf[32], df[32];
        load_half_row(qf, p.qkv + (size_t)(r0 + i) * E3 + head * HD + 32 * h, iok, p.scale);
        load_half_row(df, p.dctx + (size_t)(r0 + i) * p.E + head * HD + 32 * h, iok, 1.0f);
        float delta = 0.f;
        {
            const float* op = p.ctx_in + (size_t)(r0 + i) * p.E + head * HD + 32 * h;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (iok) v = *reinterpret_cast<const f32x4*>(op + 4 * k);
#pragma unroll
                for (int j = 0; j < 4; ++j) delta += v[j] * df[4 * k + j];
            }
            delta += __shfl_xor(delta, 32, 64);
        }
        if (h == 0) dl[i] = delta;
        const float lse_i = iok ? ls[i] : INFINITY;
        f32x16 dq[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;
        for (int jt = 0; jt < nt; ++jt) {
            const f32x16 st = mm_rows_regs(Xa + (jt * 32 + l31) * LDR + 32 * h, qf);
            const f32x16 dp = mm_rows_regs(Xb + (jt * 32 + l31) * LDR + 32 * h, df);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int j = jt * 32 + rowof(e, h);
                const float pr = j < n ? expf(st[e] - lse_i) : 0.f;
                float g = dp[e];
                if (p.thresh) g = wf3d_keep(p.seed, (uint32_t)((r0 + i) * p.heads + head), (uint32_t)j, p.thresh) ? g * p.dscale : 0.f;
                const float ds = pr * (g - delta);
                const float* kr = Xa + j * LDR + l31;
                dq[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[0], ds, dq[0], 0, 0, 0);
                dq[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[32], ds, dq[1], 0, 0, 0);
            }
        }
        if (iok) {
            float* out = p.dqkv + (size_t)(r0 + i) * E3 + head * HD;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {dq[dt][4 * g], dq[dt][4 * g + 1], dq[dt][4 * g + 2], dq[dt][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(out + dt * 32 + 8 * g + 4 * h) = v * p.scale;
                }
        }
    }
    __syncthreads();
    // ---- sweep 2: per key block -> dK, dV   (Xa = scaled Q, Xb = dO)
    stage64(Xa, p.qkv, E3, r0, n, npad, head * HD, p.scale);
    stage64(Xb, p.dctx, p.E, r0, n, npad, head * HD, 1.0f);
    __syncthreads();
    for (int kb = wave; kb < nt; kb += 4) {
        const int j = kb * 32 + l31;                       // this lane's key
        const bool jok = j < n;
        float kf[32], vf[32];
        load_half_row(kf, p.qkv + (size_t)(r0 + j) * E3 + p.E + head * HD + 32 * h, jok, 1.0f);
        load_half_row(vf, p.qkv + (size_t)(r0 + j) * E3 + 2 * p.E + head * HD + 32 * h, jok, 1.0f);
        f32x16 dk[2], dv[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { dk[dt][e] = 0.f; dv[dt][e] = 0.f; }
        for (int it = 0; it < nt; ++it) {
            const f32x16 sc = mm_rows_regs(Xa + (it * 32 + l31) * LDR + 32 * h, kf);     // S[i][j], rows = queries
            const f32x16 dp = mm_rows_regs(Xb + (it * 32 + l31) * LDR + 32 * h, vf);     // dP[i][j]
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = it * 32 + rowof(e, h);
                const float pr = expf(sc[e] - ls[i]);                                    // 0 for padded queries
                float mk = 1.0f;
                if (p.thresh) mk = wf3d_keep(p.seed, (uint32_t)((r0 + i) * p.heads + head), (uint32_t)j, p.thresh) ? p.dscale : 0.f;
                const float ds = pr * (dp[e] * mk - dl[i]);
                const float pt = pr * mk;
                const float* qr = Xa + i * LDR + l31;
                const float* gr = Xb + i * LDR + l31;
                dk[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qr[0], ds, dk[0], 0, 0, 0);
                dk[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qr[32], ds, dk[1], 0, 0, 0);
                dv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(gr[0], pt, dv[0], 0, 0, 0);
                dv[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(gr[32], pt, dv[1], 0, 0, 0);
            }
        }
        if (jok) {
            float* ok_ = p.dqkv + (size_t)(r0 + j) * E3 + p.E + head * HD;
            float* ov_ = p.dqkv + (size_t)(r0 + j) * E3 + 2 * p.E + head * HD;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 a = {dk[dt][4 * g], dk[dt][4 * g + 1], dk[dt][4 * g + 2], dk[dt][4 * g + 3]};
                    f32x4 b = {dv[dt][4 * g], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(ok_ + dt * 32 + 8 * g + 4 * h) = a;
                    *reinterpret_cast<f32x4*>(ov_ + dt * 32 + 8 * g + 4 * h) = b;
                }
        }
    }
}

}  // namespace

// internal entry points (dispatched from wf3d_attn_fwd / wf3d_attn_bwd when head_dim == 64)
extern "C" int wf3d_attn_fwd_mfma(const float* qkv, const int32_t* voff, int S, int vmax, int E, int heads, float drop_p,
                                  uint32_t drop_seed, float* ctx, float* lse, void* stream) {
    MAttn p{};
    p.qkv = qkv; p.ctx = ctx; p.lse = lse; p.voff = voff;
    p.E = E; p.heads = heads; p.vpad = (vmax + 31) & ~31;
    p.scale = 1.0f / sqrtf((float)HD);
    if (drop_p > 0.f) { p.seed = drop_seed; p.thresh = (uint32_t)((double)drop_p * 4294967296.0); p.dscale = 1.0f / (1.0f - drop_p); }
    const size_t lds = (size_t)2 * p.vpad * LDR * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        WF3D_CHECK(e == hipSuccess, WF3D_ERR_LAUNCH, "wf3d_attn_fwd: cannot raise dynamic LDS to %zu", lds);
    }
    hipLaunchKernelGGL(attn_fwd_mfma_kernel, dim3(heads, S), dim3(256), lds, (hipStream_t)stream, p);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_attn_bwd_mfma(const float* qkv, const float* dctx, const float* ctx, const float* lse,
                                  const int32_t* voff, int S, int vmax, int E, int heads, float drop_p,
                                  uint32_t drop_seed, float* dqkv, void* stream) {
    MAttn p{};
    p.qkv = qkv; p.dctx = dctx; p.ctx_in = ctx; p.lse = (float*)lse; p.dqkv = dqkv; p.voff = voff;
    p.E = E; p.heads = heads; p.vpad = (vmax + 31) & ~31;
    p.scale = 1.0f / sqrtf((float)HD);
    if (drop_p > 0.f) { p.seed = drop_seed; p.thresh = (uint32_t)((double)drop_p * 4294967296.0); p.dscale = 1.0f / (1.0f - drop_p); }
    const size_t lds = ((size_t)2 * p.vpad * LDR + 2 * p.vpad) * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        WF3D_CHECK(e == hipSuccess, WF3D_ERR_LAUNCH, "wf3d_attn_bwd: cannot raise dynamic LDS to %zu", lds);
    }
    hipLaunchKernelGGL(attn_bwd_mfma_kernel, dim3(heads, S), dim3(256), lds, (hipStream_t)stream, p);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}
