// Shared device/host helpers for the wf3d HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/wf3d.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WF3D_WAVE 64

// Internal cross-translation-unit helpers are not part of the C ABI: hidden, so that libwf3d.so exports exactly what
// include/wf3d.h declares (tests/test_capi_symbols.py checks both directions).
#define WF3D_INTERNAL __attribute__((visibility("hidden")))

// ---- error plumbing (host) -------------------------------------------------
WF3D_INTERNAL void wf3d_set_error(const char* fmt, ...);
#define WF3D_CHECK(cond, code, ...)                \
    do {                                           \
        if (!(cond)) {                             \
            wf3d_set_error(__VA_ARGS__);           \
            return (code);                         \
        }                                          \
    } while (0)
#define WF3D_LAUNCH_CHECK()                                                   \
    do {                                                                      \
        hipError_t e_ = hipGetLastError();                                    \
        if (e_ != hipSuccess) {                                               \
            wf3d_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,     \
                           hipGetErrorString(e_));                            \
            return WF3D_ERR_LAUNCH;                                           \
        }                                                                     \
    } while (0)

// internal (not part of include/wf3d.h): LDS-DMA variant of the split GEMM, gemm_split.hip
extern "C" WF3D_INTERNAL int wf3d_gemm_split_dma_ok(int M, int N, int K, int lda, int ldb);
extern "C" WF3D_INTERNAL size_t wf3d_gemm_split_dma_ws_bytes(int M, int N, int K);
extern "C" WF3D_INTERNAL int wf3d_gemm_split_dma(const void* A_sx8, const void* B_sx8, float* C, const float* bias, int M, int N,
                                   int K, int lda, int ldb, int ldc, int accumulate, void* ws, size_t ws_bytes,
                                   void* stream);

// internal: MFMA attention for head_dim 64 (attn_mfma.hip)
extern "C" WF3D_INTERNAL int wf3d_attn_fwd_mfma(const float* qkv, const int32_t* voff, int S, int vmax, int E, int heads, float drop_p,
                                  uint32_t drop_seed, float* ctx, float* lse, void* stream);
extern "C" WF3D_INTERNAL int wf3d_attn_bwd_mfma(const float* qkv, const float* dctx, const float* ctx, const float* lse,
                                  const int32_t* voff, int S, int vmax, int E, int heads, float drop_p,
                                  uint32_t drop_seed, float* dqkv, void* stream);

static inline int wf3d_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- device helpers --------------------------------------------------------
#ifdef __HIPCC__
// erf in fp32 to ~1 ulp, branch-free: both polynomial pieces (N. Juffa's minimax coefficients: |x| <= 0.9277 odd
// polynomial in x; beyond, 1 - exp(-|x| * P(|x|))) are evaluated and selected — 21 VALU operations against the ~42 of
// the library erff, whose two paths a wave with mixed arguments both walks anyway.  The edge head evaluates GELU or its
// derivative on every element of its [edges, 512 / 256 / 128] activations: at max_vertices = 256 the pair kernel and
// the LayerNorm backward of the first edge layer are bound by these instructions, not by HBM.
__device__ __forceinline__ float wf3d_erf(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    r = copysignf(1.0f - __expf(r), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    q = fmaf(q, a, a);
    return t > 0.927734375f ? r : q;
}

// activations used on the path: ReLU (encoder / vertex head), erf-GELU (edge head)
template <int ACT>
__device__ __forceinline__ float wf3d_act(float y) {
    if (ACT == WF3D_ACT_RELU) return fmaxf(y, 0.0f);
    if (ACT == WF3D_ACT_GELU) return 0.5f * y * (1.0f + wf3d_erf(y * 0.70710678118654752440f));
    return y;
}
// d act(y) / dy
template <int ACT>
__device__ __forceinline__ float wf3d_act_grad(float y) {
    if (ACT == WF3D_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
    if (ACT == WF3D_ACT_GELU) {
        const float cdf = 0.5f * (1.0f + wf3d_erf(y * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * __expf(-0.5f * y * y);
        return cdf + y * pdf;
    }
    return 1.0f;
}
__device__ __forceinline__ float wf3d_act_rt(int act, float y) {
    return act == WF3D_ACT_RELU ? wf3d_act<WF3D_ACT_RELU>(y)
         : act == WF3D_ACT_GELU ? wf3d_act<WF3D_ACT_GELU>(y) : y;
}
__device__ __forceinline__ float wf3d_act_grad_rt(int act, float y) {
    return act == WF3D_ACT_RELU ? wf3d_act_grad<WF3D_ACT_RELU>(y)
         : act == WF3D_ACT_GELU ? wf3d_act_grad<WF3D_ACT_GELU>(y) : 1.0f;
}

// Counter-based dropout keep-mask: the same (seed,row,col) gives the same bit
// in the forward prologue and in the backward kernels, so masks are never stored.
__device__ __forceinline__ bool wf3d_keep(uint32_t seed, uint32_t row, uint32_t col, uint32_t thresh) {
    uint32_t h = seed ^ (row * 0x9E3779B1u) ^ (col * 0x85EBCA77u + 0x27D4EB2Fu);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h >= thresh;
}

// edge rows of one sample: all pairs i < j of its v vertices in lexicographic order (EdgePredictor.py:83-86);
// edge index e -> (i, j), and the index of pair (i, i + 1)
__device__ __forceinline__ void edge_ij(int e, int v, int& i, int& j) {
    const float b = (float)(2 * v - 1);
    int ii = (int)floorf((b - sqrtf(fmaxf(b * b - 8.0f * (float)e, 0.f))) * 0.5f);
    ii = max(0, min(ii, v - 2));
    // offset(i) = i*(2v-i-1)/2 ; fix up float rounding
    while (ii + 1 <= v - 2 && ((ii + 1) * (2 * v - ii - 2)) / 2 <= e) ++ii;
    while (ii > 0 && (ii * (2 * v - ii - 1)) / 2 > e) --ii;
    i = ii;
    j = e - (ii * (2 * v - ii - 1)) / 2 + ii + 1;
}
__device__ __forceinline__ int edge_offset(int i, int v) { return (i * (2 * v - i - 1)) / 2; }

__device__ __forceinline__ float wf3d_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wf3d_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// An sx8 group (8 columns: 16 B of bf16 high parts, 16 B of low parts; csrc/split.hip) written by a PAIR of lanes (lane-contiguous row passes): the even lane holds columns 0..3 of the group,
// the odd lane columns 4..7, each at `dst` = the address of its own four fp32 columns.  After one exchange the even lane
// owns the eight high parts (the group's first 16 bytes) and the odd lane the eight low parts (the second 16): every
// lane writes 16 bytes right where it read 16 — consecutive lanes, consecutive bytes — instead of two stores 32 B apart.
// (Measured on a plain copy of 131072 x 2048 floats, scripts/micro/copy_patterns.hip: 5.0 -> 5.4 TB/s for the store
// pattern alone, 5.7 with non-temporal loads and stores, which the 32-B-apart stores cannot use: 3.9 TB/s.)
__device__ __forceinline__ void wf3d_store_sx8_pair(float* dst, const float (&v)[4], bool odd) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    bf16x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = (__bf16)v[j];
        lo[j] = (__bf16)(v[j] - (float)hi[j]);
    }
    const u32x2 H = __builtin_bit_cast(u32x2, hi), L = __builtin_bit_cast(u32x2, lo);
    const unsigned s0 = odd ? H[0] : L[0], s1 = odd ? H[1] : L[1];
    // neighbour exchange inside each quad: DPP quad_perm [1, 0, 3, 2] (a VALU move, no trip through the LDS crossbar)
    const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, true);
    const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, true);
    const u32x4 o = odd ? u32x4{r0, r1, L[0], L[1]} : u32x4{H[0], H[1], r0, r1};
    __builtin_nontemporal_store(__builtin_bit_cast(f32x4, o), reinterpret_cast<f32x4*>(dst));
}


#endif
