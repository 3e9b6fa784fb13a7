// Row-wise LayerNorm pieces and column reductions (HBM-bound kernels).
//
// Layout: activations are row-major [R, D]; one wave64 owns one row at a time
// and reads it as 16-B-per-lane coalesced float4 (lane l, slot i -> columns
// 4l + 256i ...), so every wave-instruction moves 1 KiB contiguous.
#include "wf3d_common.h"

namespace {

__device__ __forceinline__ bool vec_ok(const void* p, int ld, int D) {
    return ((uintptr_t)p % 16 == 0) && (ld % 4 == 0) && (D % 4 == 0);
}

// ---- row_stats: mu, rstd per row (two-pass, second pass served by L1/L2) -----
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ z, int R, int D, int ld,
                                                         float eps, float* __restrict__ mu, float* __restrict__ rs) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* p = z + (size_t)row * ld;
    const bool vec = vec_ok(z, ld, D);
    float s = 0.f;
    if (vec) {
        for (int c = lane * 4; c < D; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
    } else {
        for (int c = lane; c < D; c += 64) s += p[c];
    }
    const float mean = wf3d_wave_sum(s) / (float)D;
    float q = 0.f;
    if (vec) {
        for (int c = lane * 4; c < D; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[j] - mean; q += d * d; }
        }
    } else {
        for (int c = lane; c < D; c += 64) { const float d = p[c] - mean; q += d * d; }
    }
    const float var = wf3d_wave_sum(q) / (float)D;
    if (lane == 0) {
        mu[row] = mean;
        rs[row] = 1.0f / sqrtf(var + eps);
    }
}

// ---- ln_act_apply: out = drop(act(LN(z))) + addend ---------------------------
__global__ __launch_bounds__(256) void ln_act_apply_kernel(const float* __restrict__ z, int R, int D,
                                                            const float* __restrict__ mu, const float* __restrict__ rs,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int act, const float* __restrict__ addend, uint32_t seed,
                                                            uint32_t thresh, float scale, float* __restrict__ out) {
    const size_t total = (size_t)R * D;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / D), c = (int)(idx % D);
        float y = z[idx];
        if (mu) y = (y - mu[r]) * rs[r];
        if (gamma) y = y * gamma[c] + beta[c];
        y = wf3d_act_rt(act, y);
        if (thresh) y = wf3d_keep(seed, (uint32_t)r, (uint32_t)c, thresh) ? y * scale : 0.f;
        if (addend) y += addend[idx];
        out[idx] = y;
    }
}

// ---- ln_act_bwd ---------------------------------------------------------------
// h = drop(act(y)), y = xhat*gamma + beta, xhat = (z - mu) * rs.
//   g      = dh * dropmask * act'(y)
//   dgamma = sum_r g*xhat ; dbeta = sum_r g
//   dz     = rs * (g*gamma - mean_c(g*gamma) - xhat * mean_c(g*gamma*xhat))
//   dbias  = sum_r dz                                  
// Each wave keeps its row (z, dh) in registers (NS float4 slots per lane), so
// z and dh are read from HBM exactly once; column partials live in registers
// across the block's rows and are combined through LDS, then per-block
// partials [nblk][3][D] are summed by colsum_finalize (deterministic, no atomics).
// A row is shared by WPR waves of the workgroup (D/WPR columns each, NS float4 slots per lane)
// so that the per-lane state stays small (<= ~100 VGPRs -> >= 4 waves per SIMD): with one wave
// per 2048-wide row the kernel sat at 256 VGPRs / 1 wave per SIMD and 2.3 TB/s.  Row sums cross
// the waves through a double-buffered LDS slot and one barrier per row group.
// KX > 0 (first Linear of the per-point MLP, in_features K <= KX): its weight gradient
// dW[c, k] = sum_r dz[r, c] * x[r, k] is K more weighted column sums of the dz this kernel already holds in
// registers, so they are accumulated here ([3 + KX][D] partials) and dz itself — which nothing else needs, the
// input cloud takes no gradient — is never written.
// PAIR (first edge layer, KX = 1 with x0 = |c_i - c_j| per edge row): the pre-activation row is not read but REBUILT,
// z[e, :] = Pa[i, :] + Pb[j, :] + |c_i - c_j| * w_dist — the expression pair_fwd_kernel evaluated — from the two per-vertex
// tables (V x H per sample: L2-resident) and the row's (i, j), so that `pre` (2 KB per edge row: 2.1 GB at
// max_vertices = 256) need not be written by the forward pass nor read here.
// Columns per wave slice when WPR waves share a row: whole sx8 groups (multiples of 8) whenever the row is made of them —
// the two lanes that exchange halves of a group (even lane: columns 0..3, odd lane: 4..7) must sit in the same slice with
// the even lane first; a slice starting at a column = 4 (mod 8) paired every lane with the wrong neighbour and the sx8
// copy of dz was garbage for such widths (D = 264 with two waves per row: found by a round-3 test, no model width hits it).
__host__ __device__ __forceinline__ int slice_cols(int D, int wpr) {
    return D % 8 == 0 ? (D / 8 + wpr - 1) / wpr * 8 : (D / 4 + wpr - 1) / wpr * 4;
}

struct PairSrc {
    const float* Pa; const float* Pb; const float* wd; int wd_stride;
    const int32_t* voff; const int32_t* eoff; const int32_t* esample;
};

template <int NS, int WPR, int KX = 0, bool PAIR = false>
__global__ __launch_bounds__(256) void ln_act_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ z,
                                                          int R, int D, const float* __restrict__ mu,
                                                          const float* __restrict__ rs, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int act, uint32_t seed,
                                                          uint32_t thresh, float scale, float* __restrict__ dz,
                                                          float* __restrict__ dz_sx8, float* __restrict__ part,
                                                          const float* __restrict__ x0 = nullptr, int ldx = 0, int kx = 0,
                                                          const PairSrc ps = PairSrc{}) {
    constexpr int RPI = 4 / WPR;                       // rows per workgroup iteration
    __shared__ float xsum[2][4][2];                    // [parity][wave][s1, s2]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wslice = wave % WPR, wrow = wave / WPR;
    const int Dw = slice_cols(D, WPR);                 // columns per wave slice
    const int cbeg = wslice * Dw, cend = min(D, cbeg + Dw);
    const bool has_ln = mu != nullptr;
    f32x4 gam[NS], bet[NS];
    f32x4 a_dg[NS], a_db[NS], a_dbias[NS];
    f32x4 wdv[PAIR ? NS : 1];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int c = cbeg + lane * 4 + 256 * i;
        if (PAIR) {
            wdv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (c < cend)
#pragma unroll
                for (int k = 0; k < 4; ++k) wdv[i][k] = ps.wd[(size_t)(c + k) * ps.wd_stride];
        }
        gam[i] = (f32x4){1.f, 1.f, 1.f, 1.f};
        bet[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (gamma && c < cend) {
            gam[i] = *reinterpret_cast<const f32x4*>(gamma + c);
            bet[i] = *reinterpret_cast<const f32x4*>(beta + c);
        }
        a_dg[i] = a_db[i] = a_dbias[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 a_dw[KX > 0 ? KX : 1][NS];
#pragma unroll
    for (int k = 0; k < (KX > 0 ? KX : 1); ++k)
#pragma unroll
        for (int i = 0; i < NS; ++i) a_dw[k][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float invD = 1.0f / (float)D;
    const int iters = (R + gridDim.x * RPI - 1) / (gridDim.x * RPI);
    // the row of iteration it + 1 is loaded while row it is reduced and written: one barrier per row otherwise
    // leaves a workgroup with nothing in flight between its rows
    f32x4 zn[NS], dn[NS];
    float mn = 0.f, rn = 1.f, xkn[KX > 0 ? KX : 1];
    // PAIR: a workgroup takes a CONTIGUOUS block of edge rows, so that a wave steps from its row to its next one
    // (RPI rows further) by counting j / i up — no lookups, and Pa[i] stays in registers while i does.  (With the
    // grid-strided rows of the general form every row cost three dependent lookups before its two gathers could
    // start: measured 0.8 ms slower than reading a stored pre at max_vertices = 256.)
    auto row_of = [&](int it) { return PAIR ? (blockIdx.x * iters + it) * RPI + wrow : (it * gridDim.x + blockIdx.x) * RPI + wrow; };
    int ps_s = 0, ps_vbase = 0, ps_v = 2, ps_eend = 0, ps_i = 0, ps_j = 1, pa_row = -1;
    f32x4 pa[PAIR ? NS : 1];
    if (PAIR) {
        const int r0 = min(row_of(0), R - 1);
        ps_s = ps.esample[r0]; ps_vbase = ps.voff[ps_s]; ps_v = ps.voff[ps_s + 1] - ps_vbase; ps_eend = ps.eoff[ps_s + 1];
        edge_ij(r0 - ps.eoff[ps_s], ps_v, ps_i, ps_j);
    }
    auto load_row = [&](int it) {
        // the row index is the same for every lane of a wave: say so, and the per-row scalars (mu, rs, the K input values)
        // come through the scalar cache in one or two s_load instead of 2 + K vector loads of one address each
        const int row = __builtin_amdgcn_readfirstlane(row_of(it));
        const bool live = it < iters && row < R;
        mn = (has_ln && live) ? mu[row] : 0.f;
        rn = (has_ln && live) ? rs[row] : 1.f;
        if (KX > 0) {
#pragma unroll
            for (int k = 0; k < KX; ++k) xkn[k] = (live && k < kx) ? x0[(size_t)row * ldx + k] : 0.f;
        }
        const int ri = ps_vbase + ps_i, rj = ps_vbase + ps_j;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int c = cbeg + lane * 4 + 256 * i;
            zn[i] = dn[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (live && c < cend) {
                if (PAIR) {
                    if (ri != pa_row) pa[i] = *reinterpret_cast<const f32x4*>(ps.Pa + (size_t)ri * D + c);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(ps.Pb + (size_t)rj * D + c);
#pragma unroll
                    for (int k = 0; k < 4; ++k) zn[i][k] = pa[i][k] + b[k] + xkn[0] * wdv[i][k];      // as pair_fwd_kernel formed it
                } else {
                    zn[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(z + (size_t)row * D + c));     // last use of z
                }
                dn[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dh + (size_t)row * D + c));
            }
        }
        if (PAIR) {
            pa_row = live ? ri : pa_row;
            // step to this wave's next row: RPI edges further in the lexicographic (i, j) order, across samples
            int e = row;
#pragma unroll
            for (int k = 0; k < RPI; ++k) {
                ++e;
                if (e >= R) break;
                if (e == ps_eend) {                     // first edge of the next sample that has edges
                    ps_s = ps.esample[e]; ps_vbase = ps.voff[ps_s]; ps_v = ps.voff[ps_s + 1] - ps_vbase; ps_eend = ps.eoff[ps_s + 1];
                    ps_i = 0; ps_j = 1;
                } else if (++ps_j == ps_v) { ++ps_i; ps_j = ps_i + 1; }
            }
        }
    };
    load_row(0);
    for (int it = 0; it < iters; ++it) {
        const int row = row_of(it);
        const bool live = row < R;
        const float m = mn, r = rn;
        float xk[KX > 0 ? KX : 1];
        if (KX > 0) {
#pragma unroll
            for (int k = 0; k < KX; ++k) xk[k] = xkn[k];
        }
        f32x4 zc[NS], dc[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) { zc[i] = zn[i]; dc[i] = dn[i]; }
        load_row(it + 1);
        f32x4 xh[NS], gg[NS];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int c = cbeg + lane * 4 + 256 * i;
            xh[i] = gg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (live && c < cend) {
                const f32x4 zv = zc[i];
                const f32x4 dv = dc[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = (zv[j] - m) * r;
                    const float y = x * gam[i][j] + bet[i][j];
                    float d = dv[j];
                    if (thresh) d = wf3d_keep(seed, (uint32_t)row, (uint32_t)(c + j), thresh) ? d * scale : 0.f;
                    const float g = d * wf3d_act_grad_rt(act, y);
                    xh[i][j] = x; gg[i][j] = g;
                    const float gy = g * gam[i][j];
                    s1 += gy; s2 += gy * x;
                }
            }
        }
        float c1 = 0.f, c2 = 0.f;
        if (has_ln) {
            s1 = wf3d_wave_sum(s1); s2 = wf3d_wave_sum(s2);
            if (WPR > 1) {
                const int par = it & 1;
                if (lane == 0) { xsum[par][wave][0] = s1; xsum[par][wave][1] = s2; }
                __syncthreads();
                s1 = 0.f; s2 = 0.f;
#pragma unroll
                for (int w = 0; w < WPR; ++w) { s1 += xsum[par][wrow * WPR + w][0]; s2 += xsum[par][wrow * WPR + w][1]; }
            }
            c1 = s1 * invD; c2 = s2 * invD;
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int c = cbeg + lane * 4 + 256 * i;
            if (live && c < cend) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float g = gg[i][j], x = xh[i][j];
                    o[j] = has_ln ? r * (g * gam[i][j] - c1 - x * c2) : g;
                    a_dg[i][j] += g * x; a_db[i][j] += g; a_dbias[i][j] += o[j];
                }
                if (KX > 0) {
#pragma unroll
                    for (int k = 0; k < KX; ++k) a_dw[k][i] += o * xk[k];
                }
                if (dz) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(dz + (size_t)row * D + c));
                if (dz_sx8) {
                    // sx8 group = 8 columns = lanes (2m, 2m+1): even lane stores the 8 high parts,
                    // odd lane the 8 low parts, after swapping the halves they do not own
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    bf16x4 hi, lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { hi[j] = (__bf16)o[j]; lo[j] = (__bf16)(o[j] - (float)hi[j]); }
                    const uint2 mine_hi = __builtin_bit_cast(uint2, hi), mine_lo = __builtin_bit_cast(uint2, lo);
                    const bool odd = lane & 1;
                    uint2 send = odd ? mine_hi : mine_lo, got;
                    got.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.x, 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]
                    got.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.y, 0xB1, 0xF, 0xF, true);
                    uint4 w = odd ? make_uint4(got.x, got.y, mine_lo.x, mine_lo.y)
                                  : make_uint4(mine_hi.x, mine_hi.y, got.x, got.y);
                    float* g = dz_sx8 + (size_t)row * D + (c & ~7) + (odd ? 4 : 0);
                    // streaming: non-temporal loads and stores together are worth 6-8 % on a copy of this shape
                    // (scripts/micro/copy_patterns.hip)
                    __builtin_nontemporal_store(f32x4{__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w)},
                                                reinterpret_cast<f32x4*>(g));
                }
            }
        }
    }
    // column partials: waves of the same slice (different rows) combine through LDS, then one
    // partial row [3][D] per workgroup
    extern __shared__ __attribute__((aligned(16))) float red[];   // [3 + KX][D]
#pragma unroll
    for (int k = 0; k < 3 + KX; ++k) {
        for (int w = 0; w < RPI; ++w) {
            if (wrow == w) {
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int c = cbeg + lane * 4 + 256 * i;
                    if (c < cend) {
                        f32x4 v = k == 0 ? a_dg[i] : (k == 1 ? a_db[i] : (k == 2 ? a_dbias[i] : a_dw[k >= 3 ? k - 3 : 0][i]));
                        f32x4* dst = reinterpret_cast<f32x4*>(red + k * D + c);
                        if (w) v += *dst;
                        *dst = v;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int idx = threadIdx.x; idx < (3 + KX) * D; idx += 256) part[(size_t)blockIdx.x * (3 + KX) * D + idx] = red[idx];
}

// out[c] = sum_b part[b*stride + c].  32 columns per workgroup, the partial rows split 8 ways
// over the thread groups (coalesced 128-B reads, 4 loads in flight per thread), combined through LDS.
// tK > 0: column c = k * tD + d is stored at out[d * tK + k] (the [K][D] weight-gradient partials of the first
// layer land in the weight's own [D][K] layout).
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, int nblk, size_t stride,
                                                               int D, float* __restrict__ out, int tD = 0, int tK = 0) {
    __shared__ float red[8][32];
    const int col = threadIdx.x & 31, sub = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < D) {
        int b = sub;
        for (; b + 24 < nblk; b += 32) {
            s0 += part[(size_t)b * stride + c];
            s1 += part[(size_t)(b + 8) * stride + c];
            s2 += part[(size_t)(b + 16) * stride + c];
            s3 += part[(size_t)(b + 24) * stride + c];
        }
        for (; b < nblk; b += 8) s0 += part[(size_t)b * stride + c];
    }
    red[sub][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sub == 0 && c < D) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][col];
        out[tK > 0 ? (size_t)(c % tD) * tK + c / tD : (size_t)c] = t;
    }
}

// The same sum for column counts / strides that are multiples of 4: 64 columns per workgroup as float4, the partial
// rows split 16 ways, 4 loads (64 B) in flight per thread — twice the bytes in flight of the scalar form and 256-B
// segments per row instead of 128-B ones (the [1024][3 x 2048] partials of an encoder layer: 25 MB).
__global__ __launch_bounds__(256) void colsum_finalize4_kernel(const float* __restrict__ part, int nblk, size_t stride,
                                                                int D, float* __restrict__ out) {
    __shared__ f32x4 red[16][16];
    const int c4 = threadIdx.x & 15, sub = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + c4 * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (c < D) {
        const float* p = part + c;
        int b = sub;
        for (; b + 48 < nblk; b += 64) {
            s0 += *reinterpret_cast<const f32x4*>(p + (size_t)b * stride);
            s1 += *reinterpret_cast<const f32x4*>(p + (size_t)(b + 16) * stride);
            s2 += *reinterpret_cast<const f32x4*>(p + (size_t)(b + 32) * stride);
            s3 += *reinterpret_cast<const f32x4*>(p + (size_t)(b + 48) * stride);
        }
        for (; b < nblk; b += 16) s0 += *reinterpret_cast<const f32x4*>(p + (size_t)b * stride);
    }
    red[sub][c4] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sub == 0 && c < D) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][c4];
        *reinterpret_cast<f32x4*>(out + c) = t;
    }
}

void launch_finalize(const float* part, int nblk, size_t stride, int D, float* out, hipStream_t st, int tD = 0, int tK = 0) {
    if (tK == 0 && D % 4 == 0 && stride % 4 == 0 && ((uintptr_t)part % 16 == 0) && ((uintptr_t)out % 16 == 0) && nblk >= 64)
        hipLaunchKernelGGL(colsum_finalize4_kernel, dim3(wf3d_cdiv(D, 64)), dim3(256), 0, st, part, nblk, stride, D, out);
    else
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3(wf3d_cdiv(D, 32)), dim3(256), 0, st, part, nblk, stride, D, out, tD, tK);
}

// partial column sums: block (bx, by) sums rows [by*rpb, ...) of columns bx*256..
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int R, int D, int ld,
                                                              const float* __restrict__ w, int act, int rpb,
                                                              float* __restrict__ part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= D) return;
    const int r0 = blockIdx.y * rpb, r1 = min(R, r0 + rpb);
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += wf3d_act_rt(act, x[(size_t)r * ld + c]) * (w ? w[r] : 1.0f);
    part[(size_t)blockIdx.y * D + c] = s;
}

// ---- one-output Linear on act(z): the logits layer edge_mlp[9..10] (EdgePredictor.py:66-67) -------------
// A 1-wide GEMM wastes 127/128 of an MFMA tile; it is a row dot product.  Lanes: LPR = D/8 lanes per row
// (8 consecutive columns each), 64/LPR rows per wave.
template <int LPR>
__global__ __launch_bounds__(256) void rowdot_act_kernel(const float* __restrict__ z, int R, int D,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          int act, float* __restrict__ out) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, rsub = lane / LPR;
    const int row = (blockIdx.x * 4 + wave) * RPW + rsub;
    const int c = sub * 8;
    float s = 0.f;
    if (row < R) {
        const float* p = z + (size_t)row * D + c;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
        const f32x4 wa = *reinterpret_cast<const f32x4*>(w + c), wb = *reinterpret_cast<const f32x4*>(w + c + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += wf3d_act_rt(act, a[j]) * wa[j] + wf3d_act_rt(act, b[j]) * wb[j];
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (sub == 0 && row < R) out[row] = s + (bias ? bias[0] : 0.f);
}

// backward of the same layer fused with the activation backward of the layer below:
//   dz[r,c] = t_r * w[c] * act'(z[r,c])   (t = d logit),   dW[c] = sum_r t_r * act(z[r,c]),   dbz[c] = sum_r dz[r,c]
// dz goes out as the sx8 operand of the next dgrad / wgrad (and / or fp32); partial column sums per workgroup.
template <int LPR>
__global__ __launch_bounds__(256) void rowdot_act_bwd_kernel(const float* __restrict__ z, const float* __restrict__ t,
                                                              int R, int D, const float* __restrict__ w, int act,
                                                              float* __restrict__ dz, float* __restrict__ dz_sx8,
                                                              float* __restrict__ part) {
    constexpr int RPW = 64 / LPR;
    __shared__ float red[4 * RPW][2][LPR * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, rsub = lane / LPR;
    const int c = sub * 8;
    float wv[8], adw[8], adb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { wv[j] = w[c + j]; adw[j] = 0.f; adb[j] = 0.f; }
    for (int row = (blockIdx.x * 4 + wave) * RPW + rsub; row < R; row += gridDim.x * 4 * RPW) {
        const float tr = t[row];
        const float* p = z + (size_t)row * D + c;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float zv = j < 4 ? a[j] : b[j - 4];
            o[j] = tr * wv[j] * wf3d_act_grad_rt(act, zv);
            adw[j] += tr * wf3d_act_rt(act, zv);
            adb[j] += o[j];
        }
        if (dz) {
            float* q = dz + (size_t)row * D + c;
            *reinterpret_cast<f32x4*>(q) = f32x4{o[0], o[1], o[2], o[3]};
            *reinterpret_cast<f32x4*>(q + 4) = f32x4{o[4], o[5], o[6], o[7]};
        }
        if (dz_sx8) {
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            bf16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) { hi[j] = (__bf16)o[j]; lo[j] = (__bf16)(o[j] - (float)hi[j]); }
            float* q = dz_sx8 + (size_t)row * D + c;
            *reinterpret_cast<f32x4*>(q) = __builtin_bit_cast(f32x4, hi);
            *reinterpret_cast<f32x4*>(q + 4) = __builtin_bit_cast(f32x4, lo);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[wave * RPW + rsub][0][c + j] = adw[j]; red[wave * RPW + rsub][1][c + j] = adb[j]; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * D; idx += 256) {
        const int k = idx / D, col = idx % D;
        float sacc = 0.f;
#pragma unroll
        for (int g = 0; g < 4 * RPW; ++g) sacc += red[g][k][col];        // fixed order: deterministic
        part[(size_t)blockIdx.x * 2 * D + idx] = sacc;
    }
}

int rowdot_nblk(int R, int rpw) {
    int n = wf3d_cdiv(R, 4 * rpw * 8);
    return n > 1024 ? 1024 : (n < 1 ? 1 : n);
}
bool rowdot_ok(int D) { return D >= 8 && D <= 512 && D % 8 == 0 && ((D / 8) & (D / 8 - 1)) == 0; }

int bwd_nblk(int R) {
    int n = wf3d_cdiv(R, 4);
    return n > 1024 ? 1024 : (n < 1 ? 1 : n);
}
int colsum_nrb(int R) {
    // 8 rows per block while that stays under the cap: the per-vertex bias gradients (2,048 rows) were 64 workgroups
    // walking 64 rows each, 18 us for 4 MB
    int n = wf3d_cdiv(R, 8);
    return n > 2048 ? 2048 : (n < 1 ? 1 : n);
}

}  // namespace

extern "C" int wf3d_row_stats(const float* z, int R, int D, int ld, float eps, float* mu, float* rs, void* stream) {
    WF3D_CHECK(R >= 0 && D > 0 && ld >= D, WF3D_ERR_ARG, "wf3d_row_stats: bad dims R=%d D=%d ld=%d", R, D, ld);
    if (R == 0) return WF3D_OK;
    WF3D_CHECK(z && mu && rs, WF3D_ERR_ARG, "wf3d_row_stats: null pointer");
    hipLaunchKernelGGL(row_stats_kernel, dim3(wf3d_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, z, R, D, ld, eps, mu, rs);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_ln_act_apply(const float* z, int R, int D, const float* mu, const float* rs,
                                 const float* gamma, const float* beta, int act, const float* addend,
                                 float drop_p, uint32_t drop_seed, float* out, void* stream) {
    WF3D_CHECK(R >= 0 && D > 0, WF3D_ERR_ARG, "wf3d_ln_act_apply: bad dims");
    if (R == 0) return WF3D_OK;
    WF3D_CHECK(z && out, WF3D_ERR_ARG, "wf3d_ln_act_apply: null pointer");
    WF3D_CHECK(!mu || rs, WF3D_ERR_ARG, "wf3d_ln_act_apply: mu without rs");
    WF3D_CHECK(!gamma || beta, WF3D_ERR_ARG, "wf3d_ln_act_apply: gamma without beta");
    WF3D_CHECK(act >= 0 && act <= 2 && drop_p >= 0.f && drop_p < 1.f, WF3D_ERR_ARG, "wf3d_ln_act_apply: bad act/drop");
    const uint32_t thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    const float scale = 1.0f / (1.0f - drop_p);
    const size_t total = (size_t)R * D;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ln_act_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z, R, D, mu, rs, gamma,
                       beta, act, addend, drop_seed, thresh, scale, out);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" size_t wf3d_ln_act_bwd_ws_bytes(int R, int D) {
    if (R <= 0 || D <= 0) return 0;
    return (size_t)bwd_nblk(R) * 3 * D * sizeof(float);
}

extern "C" int wf3d_ln_act_bwd(const float* dh, const float* z, int R, int D, const float* mu, const float* rs,
                               const float* gamma, const float* beta, int act, float drop_p, uint32_t drop_seed,
                               float* dz, void* dz_sx8, float* dgamma, float* dbeta, float* dbias, void* ws,
                               size_t ws_bytes, void* stream) {
    WF3D_CHECK(R >= 0 && D > 0, WF3D_ERR_ARG, "wf3d_ln_act_bwd: bad dims");
    WF3D_CHECK(!dz_sx8 || (D % 8 == 0 && (uintptr_t)dz_sx8 % 16 == 0), WF3D_ERR_UNSUPPORTED,
               "wf3d_ln_act_bwd: the sx8 output needs D %% 8 == 0 and 16-byte alignment");
    WF3D_CHECK(D % 4 == 0 && D <= 4096, WF3D_ERR_UNSUPPORTED, "wf3d_ln_act_bwd: D=%d must be a multiple of 4, <= 4096", D);
    WF3D_CHECK(act >= 0 && act <= 2 && drop_p >= 0.f && drop_p < 1.f, WF3D_ERR_ARG, "wf3d_ln_act_bwd: bad act/drop");
    WF3D_CHECK(!mu || rs, WF3D_ERR_ARG, "wf3d_ln_act_bwd: mu without rs");
    WF3D_CHECK(!gamma || beta, WF3D_ERR_ARG, "wf3d_ln_act_bwd: gamma without beta");
    hipStream_t st = (hipStream_t)stream;
    if (R == 0) {
        if (dgamma) (void)hipMemsetAsync(dgamma, 0, D * sizeof(float), st);
        if (dbeta) (void)hipMemsetAsync(dbeta, 0, D * sizeof(float), st);
        if (dbias) (void)hipMemsetAsync(dbias, 0, D * sizeof(float), st);
        return WF3D_OK;
    }
    WF3D_CHECK(dh && z && (dz || dz_sx8), WF3D_ERR_ARG, "wf3d_ln_act_bwd: null pointer");
    WF3D_CHECK(((uintptr_t)dh % 16 == 0) && ((uintptr_t)z % 16 == 0) && ((uintptr_t)dz % 16 == 0) &&
               (!gamma || ((uintptr_t)gamma % 16 == 0 && (uintptr_t)beta % 16 == 0)),
               WF3D_ERR_ARG, "wf3d_ln_act_bwd: pointers must be 16-byte aligned");
    const int nblk = bwd_nblk(R);
    WF3D_CHECK(ws && ws_bytes >= (size_t)nblk * 3 * D * sizeof(float), WF3D_ERR_WS, "wf3d_ln_act_bwd: workspace too small");
    const uint32_t thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    const float scale = 1.0f / (1.0f - drop_p);
    float* part = (float*)ws;
    const size_t lds = (size_t)3 * D * sizeof(float);
    // waves per row: keep <= 2 float4 slots per lane (NS) wherever possible
    const int wpr = D > 512 ? 4 : (D > 256 ? 2 : 1);
    const int ns = wf3d_cdiv(slice_cols(D, wpr), 256);
#define WF3D_BWD(NS_, WPR_)                                                                                       \
    hipLaunchKernelGGL((ln_act_bwd_kernel<NS_, WPR_>), dim3(nblk), dim3(256), lds, st, dh, z, R, D, mu, rs, gamma, \
                       beta, act, drop_seed, thresh, scale, dz, (float*)dz_sx8, part)
    if (wpr == 4) { if (ns <= 1) WF3D_BWD(1, 4); else if (ns <= 2) WF3D_BWD(2, 4); else WF3D_BWD(4, 4); }
    else if (wpr == 2) WF3D_BWD(1, 2);
    else WF3D_BWD(1, 1);
#undef WF3D_BWD
    WF3D_LAUNCH_CHECK();
    float* outs[3] = {dgamma, dbeta, dbias};
    if (dgamma && dbeta == dgamma + D && dbias == dbeta + D) {
        // the three outputs are one contiguous [3][D] buffer: a single finalize pass
        launch_finalize(part, nblk, (size_t)3 * D, 3 * D, dgamma, st);
        WF3D_LAUNCH_CHECK();
        return WF3D_OK;
    }
    for (int k = 0; k < 3; ++k) {
        if (!outs[k]) continue;
        launch_finalize(part + (size_t)k * D, nblk, (size_t)3 * D, D, outs[k], st);
        WF3D_LAUNCH_CHECK();
    }
    return WF3D_OK;
}

extern "C" size_t wf3d_ln_act_bwd_first_ws_bytes(int R, int D) {
    if (R <= 0 || D <= 0) return 0;
    return (size_t)bwd_nblk(R) * (3 + 8) * D * sizeof(float);
}

extern "C" int wf3d_ln_act_bwd_first(const float* dh, const float* z, const float* x, int R, int D, int K, int ldx,
                                     const float* mu, const float* rs, const float* gamma, const float* beta, int act,
                                     float* dgamma, float* dbeta, float* dbias, float* dW, void* ws, size_t ws_bytes,
                                     void* stream) {
    WF3D_CHECK(R > 0 && D > 0 && K > 0 && K <= 8 && ldx >= K, WF3D_ERR_UNSUPPORTED, "wf3d_ln_act_bwd_first: needs R > 0, 1 <= K <= 8");
    WF3D_CHECK(D % 4 == 0 && D <= 1024, WF3D_ERR_UNSUPPORTED, "wf3d_ln_act_bwd_first: D=%d must be a multiple of 4, <= 1024", D);
    WF3D_CHECK(act >= 0 && act <= 2, WF3D_ERR_ARG, "wf3d_ln_act_bwd_first: bad act");
    WF3D_CHECK(dh && z && x && mu && rs && gamma && beta && dgamma && dbeta && dbias && dW, WF3D_ERR_ARG,
               "wf3d_ln_act_bwd_first: null pointer");
    WF3D_CHECK(dbeta == dgamma + D && dbias == dbeta + D, WF3D_ERR_ARG, "wf3d_ln_act_bwd_first: dgamma/dbeta/dbias must be one [3][D] buffer");
    WF3D_CHECK(((uintptr_t)dh % 16 == 0) && ((uintptr_t)z % 16 == 0) && ((uintptr_t)gamma % 16 == 0) && ((uintptr_t)beta % 16 == 0),
               WF3D_ERR_ARG, "wf3d_ln_act_bwd_first: pointers must be 16-byte aligned");
    const int nblk = bwd_nblk(R);
    WF3D_CHECK(ws && ws_bytes >= (size_t)nblk * 11 * D * sizeof(float), WF3D_ERR_WS, "wf3d_ln_act_bwd_first: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    const size_t lds = (size_t)11 * D * sizeof(float);
    const int wpr = D > 512 ? 4 : (D > 256 ? 2 : 1);
#define WF3D_BWD1(WPR_)                                                                                                  \
    hipLaunchKernelGGL((ln_act_bwd_kernel<1, WPR_, 8>), dim3(nblk), dim3(256), lds, st, dh, z, R, D, mu, rs, gamma, beta, \
                       act, 0u, 0u, 1.0f, (float*)nullptr, (float*)nullptr, part, x, ldx, K)
    // (one wave per 512-wide row with two slots, <2, 1, 8>: 184 VGPRs, 187 us against 173)
    if (wpr == 4) WF3D_BWD1(4); else if (wpr == 2) WF3D_BWD1(2); else WF3D_BWD1(1);
#undef WF3D_BWD1
    WF3D_LAUNCH_CHECK();
    launch_finalize(part, nblk, (size_t)11 * D, 3 * D, dgamma, st);
    WF3D_LAUNCH_CHECK();
    launch_finalize(part + 3 * D, nblk, (size_t)11 * D, D * K, dW, st, D, K);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" size_t wf3d_ln_act_bwd_wsum_ws_bytes(int R, int D) {
    if (R <= 0 || D <= 0) return 0;
    return (size_t)bwd_nblk(R) * 4 * D * sizeof(float);
}

// LayerNorm / activation backward that also returns one weighted column sum of the dz it produces:
// wsum[c] = sum_r dz[r, c] * wrow[r]   (first edge layer: the gradient of the distance column of its weight,
// wrow = |c_i - c_j| per edge row — EdgePredictor.py:130-137 — without a second pass over dz).
extern "C" int wf3d_ln_act_bwd_wsum(const float* dh, const float* z, const float* wrow, int R, int D, const float* mu,
                                    const float* rs, const float* gamma, const float* beta, int act, float drop_p,
                                    uint32_t drop_seed, float* dz, void* dz_sx8, float* dgamma, float* dbeta, float* wsum,
                                    void* ws, size_t ws_bytes, void* stream) {
    WF3D_CHECK(R > 0 && D > 0 && D % 4 == 0 && D <= 4096, WF3D_ERR_UNSUPPORTED, "wf3d_ln_act_bwd_wsum: bad dims R=%d D=%d", R, D);
    WF3D_CHECK(act >= 0 && act <= 2 && drop_p >= 0.f && drop_p < 1.f, WF3D_ERR_ARG, "wf3d_ln_act_bwd_wsum: bad act/drop");
    WF3D_CHECK(dh && z && wrow && mu && rs && gamma && beta && (dz || dz_sx8) && dgamma && dbeta && wsum && dbeta == dgamma + D,
               WF3D_ERR_ARG, "wf3d_ln_act_bwd_wsum: null pointer, or dgamma / dbeta not one [2][D] buffer");
    WF3D_CHECK(!dz_sx8 || (D % 8 == 0 && (uintptr_t)dz_sx8 % 16 == 0), WF3D_ERR_UNSUPPORTED, "wf3d_ln_act_bwd_wsum: sx8 output needs D %% 8 == 0");
    WF3D_CHECK(((uintptr_t)dh % 16 == 0) && ((uintptr_t)z % 16 == 0) && ((uintptr_t)dz % 16 == 0) && ((uintptr_t)gamma % 16 == 0) &&
               ((uintptr_t)beta % 16 == 0), WF3D_ERR_ARG, "wf3d_ln_act_bwd_wsum: pointers must be 16-byte aligned");
    const int nblk = bwd_nblk(R);
    WF3D_CHECK(ws && ws_bytes >= (size_t)nblk * 4 * D * sizeof(float), WF3D_ERR_WS, "wf3d_ln_act_bwd_wsum: workspace too small");
    const uint32_t thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    const float scale = 1.0f / (1.0f - drop_p);
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    const size_t lds = (size_t)4 * D * sizeof(float);
    const int wpr = D > 512 ? 4 : (D > 256 ? 2 : 1);
    const int ns = wf3d_cdiv(slice_cols(D, wpr), 256);
#define WF3D_BWDW(NS_, WPR_)                                                                                             \
    hipLaunchKernelGGL((ln_act_bwd_kernel<NS_, WPR_, 1>), dim3(nblk), dim3(256), lds, st, dh, z, R, D, mu, rs, gamma, beta, \
                       act, drop_seed, thresh, scale, dz, (float*)dz_sx8, part, wrow, 1, 1)
    if (wpr == 4) { if (ns <= 1) WF3D_BWDW(1, 4); else if (ns <= 2) WF3D_BWDW(2, 4); else WF3D_BWDW(4, 4); }
    else if (wpr == 2) WF3D_BWDW(1, 2);
    else WF3D_BWDW(1, 1);
#undef WF3D_BWDW
    WF3D_LAUNCH_CHECK();
    launch_finalize(part, nblk, (size_t)4 * D, 2 * D, dgamma, st);
    WF3D_LAUNCH_CHECK();
    launch_finalize(part + 3 * D, nblk, (size_t)4 * D, D, wsum, st);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

// The same pass for the first edge layer with its pre-activation rebuilt from the per-vertex tables (PairSrc above):
// dh [Re, H] -> dz (may alias dh) + dgamma / dbeta + wsum, `pre` never read.  EdgePredictor.py:122-137 backward.
extern "C" int wf3d_edge_pair_ln_bwd(const float* dh, const float* Pa, const float* Pb, const float* delta, const float* wdelta,
                                     int wdelta_stride, const int32_t* voff, const int32_t* eoff, const int32_t* esample, int Re,
                                     int H, const float* mu, const float* rs, const float* gamma, const float* beta, int act,
                                     float drop_p, uint32_t drop_seed, float* dz, float* dgamma, float* dbeta, float* wsum,
                                     void* ws, size_t ws_bytes, void* stream) {
    WF3D_CHECK(Re > 0 && H > 0 && H % 4 == 0 && H <= 1024, WF3D_ERR_UNSUPPORTED, "wf3d_edge_pair_ln_bwd: bad dims Re=%d H=%d", Re, H);
    WF3D_CHECK(act >= 0 && act <= 2 && drop_p >= 0.f && drop_p < 1.f, WF3D_ERR_ARG, "wf3d_edge_pair_ln_bwd: bad act/drop");
    WF3D_CHECK(dh && Pa && Pb && delta && wdelta && voff && eoff && esample && mu && rs && gamma && beta && dz && dgamma && dbeta &&
               wsum && dbeta == dgamma + H, WF3D_ERR_ARG, "wf3d_edge_pair_ln_bwd: null pointer, or dgamma / dbeta not one [2][H] buffer");
    WF3D_CHECK(((uintptr_t)dh % 16 == 0) && ((uintptr_t)Pa % 16 == 0) && ((uintptr_t)Pb % 16 == 0) && ((uintptr_t)dz % 16 == 0) &&
               ((uintptr_t)gamma % 16 == 0) && ((uintptr_t)beta % 16 == 0), WF3D_ERR_ARG, "wf3d_edge_pair_ln_bwd: pointers must be 16-byte aligned");
    const int nblk = bwd_nblk(Re);
    WF3D_CHECK(ws && ws_bytes >= (size_t)nblk * 4 * H * sizeof(float), WF3D_ERR_WS, "wf3d_edge_pair_ln_bwd: workspace too small");
    const uint32_t thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    const float scale = 1.0f / (1.0f - drop_p);
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    const size_t lds = (size_t)4 * H * sizeof(float);
    const PairSrc ps{Pa, Pb, wdelta, wdelta_stride, voff, eoff, esample};
    const int wpr = H > 512 ? 4 : (H > 256 ? 2 : 1);
#define WF3D_BWDP(NS_, WPR_)                                                                                                \
    hipLaunchKernelGGL((ln_act_bwd_kernel<NS_, WPR_, 1, true>), dim3(nblk), dim3(256), lds, st, dh, nullptr, Re, H, mu, rs, gamma, \
                       beta, act, drop_seed, thresh, scale, dz, nullptr, part, delta, 1, 1, ps)
    if (wpr == 4) WF3D_BWDP(1, 4); else if (wpr == 2) WF3D_BWDP(1, 2); else WF3D_BWDP(1, 1);
#undef WF3D_BWDP
    WF3D_LAUNCH_CHECK();
    launch_finalize(part, nblk, (size_t)4 * H, 2 * H, dgamma, st);
    WF3D_LAUNCH_CHECK();
    launch_finalize(part + 3 * H, nblk, (size_t)4 * H, H, wsum, st);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" size_t wf3d_edge_pair_ln_bwd_ws_bytes(int Re, int H) { return wf3d_ln_act_bwd_wsum_ws_bytes(Re, H); }

extern "C" int wf3d_rowdot_act_ok(int D) { return rowdot_ok(D) ? 1 : 0; }

extern "C" int wf3d_rowdot_act(const float* z, int R, int D, const float* w, const float* bias, int act, float* out,
                               void* stream) {
    WF3D_CHECK(R >= 0 && rowdot_ok(D), WF3D_ERR_UNSUPPORTED, "wf3d_rowdot_act: D=%d must be 8 * 2^k, <= 512", D);
    WF3D_CHECK(act >= 0 && act <= 2, WF3D_ERR_ARG, "wf3d_rowdot_act: bad act");
    if (R == 0) return WF3D_OK;
    WF3D_CHECK(z && w && out && ((uintptr_t)z % 16 == 0) && ((uintptr_t)w % 16 == 0), WF3D_ERR_ARG, "wf3d_rowdot_act: null or misaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    const int lpr = D / 8;
#define WF3D_RD(L_) hipLaunchKernelGGL((rowdot_act_kernel<L_>), dim3(wf3d_cdiv(R, 4 * (64 / L_))), dim3(256), 0, st, z, R, D, w, bias, act, out)
    switch (lpr) { case 1: WF3D_RD(1); break; case 2: WF3D_RD(2); break; case 4: WF3D_RD(4); break; case 8: WF3D_RD(8); break;
                   case 16: WF3D_RD(16); break; case 32: WF3D_RD(32); break; default: WF3D_RD(64); break; }
#undef WF3D_RD
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" size_t wf3d_rowdot_act_bwd_ws_bytes(int R, int D) {
    if (R <= 0 || !rowdot_ok(D)) return 0;
    return (size_t)rowdot_nblk(R, 64 / (D / 8)) * 2 * D * sizeof(float);
}

extern "C" int wf3d_rowdot_act_bwd(const float* z, const float* dlogit, int R, int D, const float* w, int act, float* dz,
                                   void* dz_sx8, float* dw, float* dbias_z, void* ws, size_t ws_bytes, void* stream) {
    WF3D_CHECK(R > 0 && rowdot_ok(D), WF3D_ERR_UNSUPPORTED, "wf3d_rowdot_act_bwd: D=%d must be 8 * 2^k, <= 512", D);
    WF3D_CHECK(act >= 0 && act <= 2, WF3D_ERR_ARG, "wf3d_rowdot_act_bwd: bad act");
    WF3D_CHECK(z && dlogit && w && (dz || dz_sx8) && dw && dbias_z && dbias_z == dw + D, WF3D_ERR_ARG,
               "wf3d_rowdot_act_bwd: null pointer, or dw / dbias_z not one [2][D] buffer");
    WF3D_CHECK(((uintptr_t)z % 16 == 0) && ((uintptr_t)dz % 16 == 0) && ((uintptr_t)dz_sx8 % 16 == 0), WF3D_ERR_ARG,
               "wf3d_rowdot_act_bwd: pointers must be 16-byte aligned");
    const int lpr = D / 8, nblk = rowdot_nblk(R, 64 / lpr);
    WF3D_CHECK(ws && ws_bytes >= (size_t)nblk * 2 * D * sizeof(float), WF3D_ERR_WS, "wf3d_rowdot_act_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
#define WF3D_RB(L_) hipLaunchKernelGGL((rowdot_act_bwd_kernel<L_>), dim3(nblk), dim3(256), 0, st, z, dlogit, R, D, w, act, dz, (float*)dz_sx8, part)
    switch (lpr) { case 1: WF3D_RB(1); break; case 2: WF3D_RB(2); break; case 4: WF3D_RB(4); break; case 8: WF3D_RB(8); break;
                   case 16: WF3D_RB(16); break; case 32: WF3D_RB(32); break; default: WF3D_RB(64); break; }
#undef WF3D_RB
    WF3D_LAUNCH_CHECK();
    launch_finalize(part, nblk, (size_t)2 * D, 2 * D, dw, st);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" size_t wf3d_colsum_ws_bytes(int R, int D) {
    if (R <= 0 || D <= 0) return 0;
    return (size_t)colsum_nrb(R) * D * sizeof(float);
}

extern "C" int wf3d_colsum(const float* x, int R, int D, int ld, const float* w, int act, float* out, void* ws,
                           size_t ws_bytes, void* stream) {
    WF3D_CHECK(R >= 0 && D > 0 && ld >= D && out && act >= 0 && act <= 2, WF3D_ERR_ARG, "wf3d_colsum: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (R == 0) { (void)hipMemsetAsync(out, 0, D * sizeof(float), st); return WF3D_OK; }
    WF3D_CHECK(x, WF3D_ERR_ARG, "wf3d_colsum: null x");
    const int nrb = colsum_nrb(R);
    WF3D_CHECK(ws && ws_bytes >= (size_t)nrb * D * sizeof(float), WF3D_ERR_WS, "wf3d_colsum: workspace too small");
    const int rpb = wf3d_cdiv(R, nrb);
    float* part = (float*)ws;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(wf3d_cdiv(D, 256), nrb), dim3(256), 0, st, x, R, D, ld, w, act, rpb, part);
    WF3D_LAUNCH_CHECK();
    launch_finalize(part, wf3d_cdiv(R, rpb), (size_t)D, D, out, st);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}
