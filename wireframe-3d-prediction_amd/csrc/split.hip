// Producers of the "sx8" split-precision operand format consumed by wf3d_gemm_split.
//
// sx8: a logical fp32 matrix [R, C] (C % 8 == 0) stored with the SAME bytes and row
// pitch as fp32, but every 32-byte group of 8 consecutive columns holds
//     [ 8 x bf16 high parts | 8 x bf16 low parts ],   v ~= hi + lo,
// hi = bf16_rne(v), lo = bf16_rne(v - hi)  (|v - hi - lo| <= 2^-17 |v|).  A staged
// row slice then delivers each lane's v_mfma_f32_32x32x16_bf16 fragment (8
// consecutive k) as one 16-byte chunk, with no conversion inside the GEMM.
#include <stdlib.h>

#include "wf3d_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void store_sx8(float* dst, const float (&v)[8]) {
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = (__bf16)v[j];
        lo[j] = (__bf16)(v[j] - (float)hi[j]);
    }
    *reinterpret_cast<f32x4*>(dst) = __builtin_bit_cast(f32x4, hi);
    *reinterpret_cast<f32x4*>(dst + 4) = __builtin_bit_cast(f32x4, lo);
}

// out_sx8[r, c] = in[r*rs + c*cs]   (cs == 1: plain copy-convert; rs == 1: transpose)
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ in, long rs, long cs, int R, int C,
                                                          float* __restrict__ out) {
    const long groups = (long)R * (C / 8);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < groups; idx += (long)gridDim.x * 256) {
        const int r = (int)(idx / (C / 8)), g = (int)(idx % (C / 8));
        const float* p = in + (long)r * rs + (long)g * 8 * cs;
        float v[8];
        if (cs == 1 && (((uintptr_t)p) % 16 == 0)) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p[(long)j * cs];
        }
        store_sx8(out + (long)r * C + (long)g * 8, v);
    }
}

// Up to 8 independent split_rows jobs in ONE launch: the weight matrices of a stage, each in both orientations (W for the
// forward GEMM, W^T for its dgrad), are split at the top of the forward instead of by 8 launches of ~7 us.
struct SplitJobs {
    const float* in[8];
    float* out[8];
    long rs[8], cs[8];
    int R[8], C[8], blk_begin[9];
    int njobs;
};
__global__ __launch_bounds__(256) void split_rows_multi_kernel(const SplitJobs J) {
    int j = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (i < J.njobs && (int)blockIdx.x >= J.blk_begin[i]) j = i;
    const int nblk = J.blk_begin[j + 1] - J.blk_begin[j], blk = blockIdx.x - J.blk_begin[j];
    const float* in = J.in[j];
    float* out = J.out[j];
    const long rs = J.rs[j], cs = J.cs[j];
    const int C = J.C[j];
    if (rs == 1 && cs != 1 && J.R[j] % 64 == 0 && C % 64 == 0) {
        // transposed view of a row-major matrix (the W^T operand of a dgrad): 64 x 64 tiles through LDS, so the reads
        // run along the input's rows (the element-wise walk below reads 4 bytes per 64-byte sector: 45 us for the
        // encoder's 5.2 M weights)
        __shared__ float tile[64][65];
        const int tr = J.R[j] / 64, ntile = tr * (C / 64);
        for (int t = blk; t < ntile; t += nblk) {
            const int r0 = (t % tr) * 64, c0 = (t / tr) * 64;
            const int rr = threadIdx.x & 63, cq = threadIdx.x >> 6;
#pragma unroll
            for (int i = 0; i < 16; ++i) tile[cq + 4 * i][rr] = in[(long)(r0 + rr) + (long)(c0 + cq + 4 * i) * cs];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = threadIdx.x + 256 * i, g = idx & 7, ro = idx >> 3;
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = tile[8 * g + k][ro];
                store_sx8(out + (long)(r0 + ro) * C + c0 + 8 * g, v);
            }
            __syncthreads();
        }
        return;
    }
    const long groups = (long)J.R[j] * (C / 8);
    for (long idx = (long)blk * 256 + threadIdx.x; idx < groups; idx += (long)nblk * 256) {
        const int r = (int)(idx / (C / 8)), g = (int)(idx % (C / 8));
        const float* p = in + (long)r * rs + (long)g * 8 * cs;
        float v[8];
        if (cs == 1 && (((uintptr_t)p) % 16 == 0)) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = a[k]; v[4 + k] = b[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p[(long)k * cs];
        }
        store_sx8(out + (long)r * C + (long)g * 8, v);
    }
}

// LayerNorm statistics + h = act(LN(z)) written in sx8: LPR lanes per row (64: one wave per row; 32: two rows per wave,
// for D <= 256 where 8 columns per lane would leave half of the wave idle), the row lives in registers (NS slots of 8
// columns per lane), z is read from HBM once.
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// FULL: D == NS * 8 * LPR exactly — every column test folds away and the row loop is straight-line code
template <int NS, int LPR = 64, bool FULL = false>
__global__ __launch_bounds__(256) void ln_prep_kernel(const float* __restrict__ z, int R, int D,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int act, float eps, uint32_t seed, uint32_t thresh, float dscale,
                                                       float* __restrict__ mu, float* __restrict__ rs,
                                                       float* __restrict__ h_sx8) {
    // persistent: a lane group walks rows first, first + step, ... with its slice of gamma / beta in registers and
    // the next row's z already requested while this row is reduced, normalised and stored.
    // Columns of a lane: two chunks of four per slot, CH columns apart — lane l owns [s*SPAN + 4l, +4) and
    // [s*SPAN + CH + 4l, +4), so that every load and every store of the wave covers consecutive 16-byte pieces
    // (SPAN = 8 * LPR columns per slot, CH = 4 * LPR).  Lanes 2m, 2m+1 share an sx8 group (store_sx8_pair).
    constexpr int RPB = 256 / LPR, SPAN = 8 * LPR, CH = 4 * LPR;
    const int lane = threadIdx.x & (LPR - 1);
    const bool odd = lane & 1;
    const int rstep = gridDim.x * RPB;
    float gm[NS][8], bt[NS][8];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { gm[i][j] = 1.f; bt[i][j] = 0.f; }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int c = SPAN * i + CH * hf + lane * 4;
            if (FULL || c < D) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c), b0 = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) { gm[i][4 * hf + j] = g0[j]; bt[i][4 * hf + j] = b0[j]; }
            }
        }
    }
    f32x4 na[NS], nb[NS];
    auto load_row = [&](int row) {
        const float* p = z + (size_t)min(row, R - 1) * D;          // clamped: a row >= R is loaded but never used
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int c = SPAN * i + lane * 4;
            na[i] = nb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (FULL || c < D) na[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + c));
            if (FULL || c + CH < D) nb[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + c + CH));
        }
    };
    int row = blockIdx.x * RPB + threadIdx.x / LPR;
    load_row(row);
    // whole rows (aligned lane groups) leave together: the shuffles below stay inside a row
    for (; row < R; row += rstep) {
        float v[NS][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[i][j] = na[i][j]; v[i][4 + j] = nb[i][j]; s += na[i][j] + nb[i][j]; }
        load_row(row + rstep);
        const float mean = row_sum<LPR>(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                if (FULL || SPAN * i + CH * hf + lane * 4 < D)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float d = v[i][4 * hf + j] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(row_sum<LPR>(q) / (float)D + eps);
        if (lane == 0) { mu[row] = mean; rs[row] = rstd; }
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int c = SPAN * i + CH * hf + lane * 4;
                if (FULL || c < D) {               // D % 8 == 0: both lanes of a pair are in or out together
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = wf3d_act_rt(act, (v[i][4 * hf + j] - mean) * rstd * gm[i][4 * hf + j] + bt[i][4 * hf + j]);
                    if (thresh) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            o[j] = wf3d_keep(seed, (uint32_t)row, (uint32_t)(c + j), thresh) ? o[j] * dscale : 0.f;
                    }
                    wf3d_store_sx8_pair(h_sx8 + (size_t)row * D + c, o, odd);
                }
            }
    }
}


// First Linear of the per-point MLP fused with its LayerNorm / activation / split
// (PointNetEncoder.py:35-45 with in_features = input_dim = 8): z = x·W^T + b is a pure
// 2 KB-per-row WRITE (K <= 8: 8 FMAs per output), so instead of a GEMM that writes z and an
// ln_prep pass that reads it back, one wave per row computes the row in registers from its
// slice of W (kept in registers across rows), and writes z (kept for backward), (mu, rstd)
// and h = act(LN(z)) in sx8.  Column layout per lane as in ln_prep_kernel (lane-contiguous 16-B pieces, lane pairs
// share an sx8 group; z and h leave through non-temporal stores).
// FAST: K == 8, D a multiple of 512, 16-byte aligned x / W rows and ReLU — two float4 loads per row, no per-element branches.
template <int NS, bool FAST>
__global__ __launch_bounds__(256) void first_layer_kernel(const float* __restrict__ x, int ldx, int K,
                                                           const float* __restrict__ W, int ldw,
                                                           const float* __restrict__ bias, int R, int D,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int act, float eps, float* __restrict__ z,
                                                           float* __restrict__ mu, float* __restrict__ rs,
                                                           float* __restrict__ h_sx8) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool odd = lane & 1;
    // columns of a lane, slot i: [512 i + 4 lane, +4) (values 0..3) and [512 i + 256 + 4 lane, +4) (values 4..7)
    auto col = [&](int i, int j) { return 512 * i + 256 * (j >> 2) + lane * 4 + (j & 3); };
    // W slice, bias, gamma, beta of the lane's columns stay in registers for all its rows (a load inside the row loop makes
    // the compiler wait for EVERYTHING in flight, the row's own stores included, once per row).
    // FAST also means D == 512 NS, K == 8 and 16-byte aligned W rows: the prologue is 16 + 6 vector loads, no tests.
    float w[NS][8][8], b[NS][8], gm[NS][8], bt[NS][8];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if (FAST) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int c = col(i, 4 * hf);
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + c), b4 = *reinterpret_cast<const f32x4*>(beta + c);
                f32x4 bb = {0.f, 0.f, 0.f, 0.f};
                if (bias) bb = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    gm[i][4 * hf + j] = g4[j]; bt[i][4 * hf + j] = b4[j]; b[i][4 * hf + j] = bb[j];
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(W + (size_t)(c + j) * ldw), w1 = *reinterpret_cast<const f32x4*>(W + (size_t)(c + j) * ldw + 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { w[i][4 * hf + j][k] = w0[k]; w[i][4 * hf + j][4 + k] = w1[k]; }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = col(i, j);
                b[i][j] = (c < D && bias) ? bias[c] : 0.f;
                gm[i][j] = c < D ? gamma[c] : 1.f;
                bt[i][j] = c < D ? beta[c] : 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) w[i][j][k] = (c < D && k < K) ? W[(size_t)c * ldw + k] : 0.f;
            }
        }
    }
    // the next row's x is requested before this row is computed and stored (one row ~ a memory round trip otherwise)
    auto load_x = [&](int row, float (&xr)[8]) {
        if (FAST) {
            const float* px = x + (size_t)min(row, R - 1) * ldx;           // clamped: the value of a row >= R is never used
            const f32x4 a = *reinterpret_cast<const f32x4*>(px), c4 = *reinterpret_cast<const f32x4*>(px + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { xr[k] = a[k]; xr[4 + k] = c4[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) xr[k] = (row < R && k < K) ? x[(size_t)row * ldx + k] : 0.f;
        }
    };
    float xn[8];
    load_x(blockIdx.x * 4 + wave, xn);
    for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
        float xr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xr[k] = xn[k];
        load_x(row + gridDim.x * 4, xn);
        float v[NS][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) a = fmaf(xr[k], w[i][j][k], a);
                v[i][j] = (FAST || col(i, j) < D) ? a + b[i][j] : 0.f;
                s += v[i][j];
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                if (FAST || col(i, 4 * hf) < D)
                    __builtin_nontemporal_store(f32x4{v[i][4 * hf], v[i][4 * hf + 1], v[i][4 * hf + 2], v[i][4 * hf + 3]},
                                                reinterpret_cast<f32x4*>(z + (size_t)row * D + col(i, 4 * hf)));
        }
        const float mean = wf3d_wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (FAST || col(i, j) < D) { const float d = v[i][j] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wf3d_wave_sum(q) / (float)D + eps);
        if (lane == 0) { mu[row] = mean; rs[row] = rstd; }
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int c = col(i, 4 * hf);
                if (FAST || c < D) {               // D % 8 == 0: both lanes of a pair are in or out together
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float y0 = (v[i][4 * hf + j] - mean) * rstd * gm[i][4 * hf + j] + bt[i][4 * hf + j];
                        o[j] = FAST ? fmaxf(y0, 0.f) : wf3d_act_rt(act, y0);
                    }
                    wf3d_store_sx8_pair(h_sx8 + (size_t)row * D + c, o, odd);
                }
            }
    }
}

// out_sx8[c, r] = split( pro(in[r, c]) ): tiled transpose through LDS so that both the
// fp32 reads (256 B per 16 lanes) and the sx8 writes (128 B per 4 lanes) are coalesced.
// pro = optional act(LN-affine(.)) with per-row (mu, rs) and per-column (gamma, beta):
// this is how the wgrad operands h^T and dz^T (k = point index contiguous) are produced.
__global__ __launch_bounds__(256) void split_transpose_kernel(const float* __restrict__ in, int R, int C, int ld,
                                                               const float* __restrict__ mu, const float* __restrict__ rs,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int act,
                                                               uint32_t seed, uint32_t thresh, float dscale,
                                                               int in_sx8, float* __restrict__ out) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int t = threadIdx.x;
    if (in_sx8) {
        // input already split (e.g. dz written by ln_act_bwd): rebuild v = hi + lo (exact in fp32)
        const int g = t & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rr = (t >> 3) + 32 * i;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (r0 + rr < R && c0 + g * 8 < C) {
                const float* p = in + (size_t)(r0 + rr) * ld + c0 + g * 8;
                const bf16x8 hi = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(p));
                const bf16x8 lo = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(p + 4));
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (float)hi[j] + (float)lo[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) tile[rr][g * 8 + j] = v[j];
        }
    } else {
        const int cc = (t & 15) * 4;
        f32x4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (gamma && c0 + cc < C) { g = *reinterpret_cast<const f32x4*>(gamma + c0 + cc); b = *reinterpret_cast<const f32x4*>(beta + c0 + cc); }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = (t >> 4) + 16 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r0 + rr < R && c0 + cc < C) {
                v = *reinterpret_cast<const f32x4*>(in + (size_t)(r0 + rr) * ld + c0 + cc);
                if (mu) {
                    const float m = mu[r0 + rr], s = rs[r0 + rr];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (v[j] - m) * s;
                }
                if (gamma) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] * g[j] + b[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = wf3d_act_rt(act, v[j]);
                if (thresh) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        v[j] = wf3d_keep(seed, (uint32_t)(r0 + rr), (uint32_t)(c0 + cc + j), thresh) ? v[j] * dscale : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[rr][cc + j] = v[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = t >> 2, g8 = (t & 3) + 4 * j;          // output row c0+c, group of 8 source rows
        if (c0 + c < C && r0 + g8 * 8 < R) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = tile[g8 * 8 + k][c];
            store_sx8(out + (size_t)(c0 + c) * R + r0 + g8 * 8, v);
        }
    }
}

}  // namespace

extern "C" int wf3d_split_rows(const float* in, long row_stride, long col_stride, int R, int C, void* out_sx8,
                               void* stream) {
    WF3D_CHECK(R >= 0 && C > 0 && C % 8 == 0, WF3D_ERR_UNSUPPORTED, "wf3d_split_rows: C=%d must be a positive multiple of 8", C);
    if (R == 0) return WF3D_OK;
    WF3D_CHECK(in && out_sx8 && ((uintptr_t)out_sx8 % 16 == 0), WF3D_ERR_ARG, "wf3d_split_rows: null or misaligned pointer");
    const long groups = (long)R * (C / 8);
    long blocks = (groups + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_rows_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, in, row_stride,
                       col_stride, R, C, (float*)out_sx8);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_split_rows_multi(const float* const* in, const long* row_stride, const long* col_stride, const int* R,
                                    const int* C, void* const* out_sx8, int njobs, void* stream) {
    WF3D_CHECK(njobs >= 1 && njobs <= 8 && in && row_stride && col_stride && R && C && out_sx8, WF3D_ERR_ARG,
               "wf3d_split_rows_multi: 1..8 jobs");
    SplitJobs J;
    J.njobs = njobs;
    int blk = 0;
    for (int i = 0; i < njobs; ++i) {
        WF3D_CHECK(in[i] && out_sx8[i] && R[i] > 0 && C[i] > 0 && C[i] % 8 == 0, WF3D_ERR_ARG, "wf3d_split_rows_multi: job %d (C must be a multiple of 8)", i);
        WF3D_CHECK(((uintptr_t)out_sx8[i] % 16) == 0, WF3D_ERR_ARG, "wf3d_split_rows_multi: job %d output must be 16-byte aligned", i);
        J.in[i] = in[i]; J.out[i] = (float*)out_sx8[i]; J.rs[i] = row_stride[i]; J.cs[i] = col_stride[i]; J.R[i] = R[i]; J.C[i] = C[i];
        J.blk_begin[i] = blk;
        long b = ((long)R[i] * (C[i] / 8) + 255) / 256;
        blk += (int)(b > 2048 ? 2048 : b);
    }
    for (int i = njobs; i <= 8; ++i) J.blk_begin[i] = blk;
    hipLaunchKernelGGL(split_rows_multi_kernel, dim3(blk), dim3(256), 0, (hipStream_t)stream, J);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_ln_prep(const float* z, int R, int D, const float* gamma, const float* beta, int act, float eps,
                            float drop_p, uint32_t drop_seed, float* mu, float* rs, void* h_sx8, void* stream) {
    WF3D_CHECK(drop_p >= 0.f && drop_p < 1.f, WF3D_ERR_ARG, "wf3d_ln_prep: bad drop_p");
    const uint32_t thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    const float dscale = 1.0f / (1.0f - drop_p);
    WF3D_CHECK(R >= 0 && D > 0, WF3D_ERR_ARG, "wf3d_ln_prep: bad dims");
    WF3D_CHECK(D % 8 == 0 && D <= 4096, WF3D_ERR_UNSUPPORTED, "wf3d_ln_prep: D=%d must be a multiple of 8, <= 4096", D);
    WF3D_CHECK(act >= 0 && act <= 2, WF3D_ERR_ARG, "wf3d_ln_prep: bad act");
    if (R == 0) return WF3D_OK;
    WF3D_CHECK(z && gamma && beta && mu && rs && h_sx8, WF3D_ERR_ARG, "wf3d_ln_prep: null pointer");
    WF3D_CHECK(((uintptr_t)z % 16 == 0) && ((uintptr_t)h_sx8 % 16 == 0) && ((uintptr_t)gamma % 16 == 0) &&
               ((uintptr_t)beta % 16 == 0), WF3D_ERR_ARG, "wf3d_ln_prep: pointers must be 16-byte aligned");
    const int ns = wf3d_cdiv(D, 512);
    hipStream_t st = (hipStream_t)stream;
    // persistent grid: a few workgroups per CU (WF3D_LNPREP_WGS overrides, for sweeps)
    static const int lp_wgs = [] { const char* e = getenv("WF3D_LNPREP_WGS"); return e ? atoi(e) : 4096; }();
    const int rpb = D <= 256 ? 8 : 4;
    int blocks = wf3d_cdiv(R, rpb);
    if (blocks > lp_wgs) blocks = lp_wgs;
#define WF3D_LP(NS_)                                                                                              \
    if (D == NS_ * 512)                                                                                           \
        hipLaunchKernelGGL((ln_prep_kernel<NS_, 64, true>), dim3(blocks), dim3(256), 0, st, z, R, D, gamma, beta, act, eps, \
                           drop_seed, thresh, dscale, mu, rs, (float*)h_sx8);                                      \
    else                                                                                                          \
        hipLaunchKernelGGL((ln_prep_kernel<NS_>), dim3(blocks), dim3(256), 0, st, z, R, D, gamma, beta, act, eps, \
                           drop_seed, thresh, dscale, mu, rs, (float*)h_sx8)
    if (D <= 256)      // two rows per wave
        hipLaunchKernelGGL((ln_prep_kernel<1, 32>), dim3(blocks), dim3(256), 0, st, z, R, D, gamma, beta, act, eps,
                           drop_seed, thresh, dscale, mu, rs, (float*)h_sx8);
    else if (ns <= 1) { WF3D_LP(1); } else if (ns <= 2) { WF3D_LP(2); } else if (ns <= 4) { WF3D_LP(4); } else { WF3D_LP(8); }
#undef WF3D_LP
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_first_layer_fwd(const float* x, int R, int K, int ldx, const float* W, int ldw, const float* bias,
                                    int D, const float* gamma, const float* beta, int act, float eps, float* z,
                                    float* mu, float* rs, void* h_sx8, void* stream) {
    WF3D_CHECK(R >= 0 && K > 0 && K <= 8 && ldx >= K && ldw >= K, WF3D_ERR_UNSUPPORTED,
               "wf3d_first_layer_fwd: needs 1 <= K <= 8 (K=%d)", K);
    WF3D_CHECK(D > 0 && D % 8 == 0 && D <= 1024, WF3D_ERR_UNSUPPORTED, "wf3d_first_layer_fwd: D=%d must be a multiple of 8, <= 1024", D);
    WF3D_CHECK(act >= 0 && act <= 2, WF3D_ERR_ARG, "wf3d_first_layer_fwd: bad act");
    if (R == 0) return WF3D_OK;
    WF3D_CHECK(x && W && gamma && beta && z && mu && rs && h_sx8, WF3D_ERR_ARG, "wf3d_first_layer_fwd: null pointer");
    WF3D_CHECK(((uintptr_t)z % 16 == 0) && ((uintptr_t)h_sx8 % 16 == 0) && ((uintptr_t)gamma % 16 == 0) &&
               ((uintptr_t)beta % 16 == 0), WF3D_ERR_ARG, "wf3d_first_layer_fwd: pointers must be 16-byte aligned");
    // each wave keeps its slice of W in registers for all its rows; 4096 workgroups measured best of 512..32768 with the
    // vector prologue (118 / 124 / 112 / 106 / 129 us at 512 / 1024 / 2048 / 4096 / 8192; 512 was best while the prologue
    // was 72 single loads)
    static const int fl_blocks = [] { const char* e = getenv("WF3D_FL_BLOCKS"); return e ? atoi(e) : 4096; }();
    int blocks = wf3d_cdiv(R, 4);
    blocks = blocks > fl_blocks ? fl_blocks : (blocks < 1 ? 1 : blocks);
    hipStream_t st = (hipStream_t)stream;
    const bool fast = K == 8 && ldx % 4 == 0 && ((uintptr_t)x % 16 == 0) && act == WF3D_ACT_RELU && D % 512 == 0 && ldw % 4 == 0 &&
                      ((uintptr_t)W % 16 == 0) && (!bias || (uintptr_t)bias % 16 == 0);
#define WF3D_FL(NS_, F_)                                                                                               \
    hipLaunchKernelGGL((first_layer_kernel<NS_, F_>), dim3(blocks), dim3(256), 0, st, x, ldx, K, W, ldw, bias, R, D, gamma, \
                       beta, act, eps, z, mu, rs, (float*)h_sx8)
    if (D <= 512) { if (fast) WF3D_FL(1, true); else WF3D_FL(1, false); }
    else          { if (fast) WF3D_FL(2, true); else WF3D_FL(2, false); }
#undef WF3D_FL
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_split_transpose(const float* in, int R, int C, int ld, const float* mu, const float* rs,
                                    const float* gamma, const float* beta, int act, float drop_p, uint32_t drop_seed,
                                    int in_sx8, void* out_sx8, void* stream) {
    WF3D_CHECK(drop_p >= 0.f && drop_p < 1.f && (!in_sx8 || drop_p == 0.f), WF3D_ERR_ARG, "wf3d_split_transpose: bad drop_p");
    const uint32_t thresh = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    const float dscale = 1.0f / (1.0f - drop_p);
    WF3D_CHECK(R > 0 && C > 0 && ld >= C, WF3D_ERR_ARG, "wf3d_split_transpose: bad dims");
    WF3D_CHECK(!in_sx8 || (C % 8 == 0 && ld % 8 == 0 && !mu && !gamma && act == 0), WF3D_ERR_UNSUPPORTED,
               "wf3d_split_transpose: an sx8 input needs C %% 8 == 0 and takes no prologue");
    WF3D_CHECK(R % 8 == 0 && C % 4 == 0 && ld % 4 == 0, WF3D_ERR_UNSUPPORTED,
               "wf3d_split_transpose: R %% 8, C %% 4 and ld %% 4 must be 0 (R=%d C=%d ld=%d)", R, C, ld);
    WF3D_CHECK(in && out_sx8 && ((uintptr_t)in % 16 == 0) && ((uintptr_t)out_sx8 % 16 == 0), WF3D_ERR_ARG,
               "wf3d_split_transpose: null or misaligned pointer");
    WF3D_CHECK(!mu || rs, WF3D_ERR_ARG, "wf3d_split_transpose: mu without rs");
    WF3D_CHECK(!gamma || (beta && (uintptr_t)gamma % 16 == 0 && (uintptr_t)beta % 16 == 0), WF3D_ERR_ARG,
               "wf3d_split_transpose: gamma/beta must both be given, 16-byte aligned");
    WF3D_CHECK(act >= 0 && act <= 2, WF3D_ERR_ARG, "wf3d_split_transpose: bad act");
    WF3D_CHECK(wf3d_cdiv(R, 64) <= 65535, WF3D_ERR_UNSUPPORTED, "wf3d_split_transpose: too many rows");
    hipLaunchKernelGGL(split_transpose_kernel, dim3(wf3d_cdiv(C, 64), wf3d_cdiv(R, 64)), dim3(256), 0,
                       (hipStream_t)stream, in, R, C, ld, mu, rs, gamma, beta, act, drop_seed, thresh, dscale, in_sx8,
                       (float*)out_sx8);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}
