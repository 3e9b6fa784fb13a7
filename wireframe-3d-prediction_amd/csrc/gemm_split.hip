// bf16x3 split-precision GEMM, LDS-DMA staged:  C[M,N] (+)= A·B^T (+ bias),
// A = sx8[M,K], B = sx8[N,K]  (sx8: see split.hip / include/wf3d.h).
//
// Operands need no transformation on their way in, so tiles go global -> LDS with
// `global_load_lds_dwordx4` (no VGPR staging, no ds_write): each wave-instruction
// lands 8 rows x 128 B = 1 KiB contiguously.  The LDS image is therefore unpadded
// [128 rows][128 B]; bank conflicts are removed by an XOR swizzle of the 16-B chunk
// index, chunk' = chunk ^ ((row >> 1) & 7), applied on the per-lane SOURCE address
// of the DMA and again on the fragment read (both-sides rule, cdna_hip_programming.md
// §5.4 rule 21): the 16 lanes of every ds_read_b128 group then hit 16 distinct slots.
// One 16-B chunk is exactly one MFMA fragment (8 consecutive k of one bf16 plane).
//
// Every product costs three bf16 MFMAs, acc += al*bh + ah*bl + ah*bh (fp32 accumulate).  Kernels in this file:
//   gemm_split_x16p_kernel / gemm_split_x16_kernel<false>   forward + dgrad: 256x256x32 tile, v_mfma_f32_16x16x32_bf16,
//                                                            3 A + 2 B LDS stages (persistent / plain launch)
//   gemm_split_x16_kernel<true>                              wgrad on reduction-major operands (ds_read_b64_tr_b16)
//   gemm_split_dma3_kernel / gemm_split_tn_kernel            256x128 tile, 32x32x16 MFMA: problems with < 512 tiles /
//                                                            wgrad outputs that are not multiples of 256
//   split_reduce_kernel                                      split-K slab fold
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "wf3d_common.h"

#ifndef WF3D_STAMP
#define WF3D_STAMP 0       // diagnostic build: the persistent kernel records s_memtime at every slice start (scripts/stamp_gemm.py)
#endif


namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int SBK = 32;                 // floats (= 128 B = 32 split elements) per row slice
constexpr int STILE = 128 * SBK;        // floats per operand tile (16 KB)

struct SplitParams {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc;
    int accumulate;
    int ksplit, kt_per_split;
    float* slab;
    int nbm, nbn;
    unsigned* ctl;     // persistent kernel: 8 tile-claim counters (64 B apart, zeroed per launch) + one mailbox word per workgroup
    int zmap;          // x16 kernel, wgrad: 1 = 1-D grid of tiles x ksplit blocks, XCD k owns the split-K ranges z = k (mod 8)
};

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The same instruction issued from inline asm.  The compiler models the builtin as a FLAT access
// that may touch LDS and from then on turns every LDS wait into `s_waitcnt lgkmcnt(0)`, which
// drains the fragment reads just issued for LATER MFMA groups; hidden in asm, the ds_read
// bookkeeping stays exact (counted lgkmcnt) and the DMA is tracked by our own vmcnt waits.
__device__ __forceinline__ void dma16_asm(const float* src, float* lds_wave_base) {
    const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds_wave_base;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(src), "s"(lds) : "memory");      // m0 is reserved: the compiler never keeps a value in it across statements
}

// ---------------------------------------------------------------------------
// Deep-pipelined variant: 256x128 tile, 8 waves (4 x 2, 64x64 each), THREE LDS
// stages (144 KB, one workgroup per CU, two waves per SIMD).  The DMA of slice
// t+2 is issued before the MFMAs of slice t and stays in flight across the
// barrier: a counted `s_waitcnt vmcnt(6)` (the 6 youngest = slice t+2's pieces)
// retires slice t+1 only, and a raw s_barrier publishes it (never __syncthreads,
// whose fence would drain vmcnt to 0 — cdna_hip_programming.md "Pipelining
// across barriers").  The 2-stage kernel above stalls every slice for the DMA
// latency (issue -> landed ~1-2k cycles vs 768 MFMA cycles per slice).
// ---------------------------------------------------------------------------
constexpr int T3_A = 256 * SBK, T3_B = 128 * SBK, T3_STAGE = T3_A + T3_B;

struct Frag { f32x4 ah[2], al[2], bh[2], bl[2]; };

__device__ __forceinline__ void load_frag(Frag& f, const float* As, const float* Bs, int wm, int wn, int l31, int h,
                                          int fsw, int s2) {
    const int phi = ((2 * (2 * s2 + h)) ^ fsw) * 4, plo = phi ^ 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float* pr = As + ((wm * 2 + i) * 32 + l31) * SBK;
        f.ah[i] = *reinterpret_cast<const f32x4*>(pr + phi);
        f.al[i] = *reinterpret_cast<const f32x4*>(pr + plo);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float* pr = Bs + ((wn * 2 + j) * 32 + l31) * SBK;
        f.bh[j] = *reinterpret_cast<const f32x4*>(pr + phi);
        f.bl[j] = *reinterpret_cast<const f32x4*>(pr + plo);
    }
}

// 12 MFMAs of one k16 step; DMA pieces P0, P0+1, P0+2 of the slice two ahead are issued one
// at a time after the 3rd, 6th and 9th MFMA (pieces 0..3 = A rows, 4..5 = B rows).
template <int P0>
__device__ __forceinline__ void mma12(f32x16 (&acc)[2][2], const Frag& f, bool ahead, const float* const (&asrc)[4],
                                      const float* const (&bsrc)[2], int kn, float* dA, float* dB) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.al[i]), __builtin_bit_cast(bf16x8, f.bh[j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.ah[i]), __builtin_bit_cast(bf16x8, f.bl[j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.ah[i]), __builtin_bit_cast(bf16x8, f.bh[j]), acc[i][j], 0, 0, 0);
            if (i * 2 + j < 3) {
                constexpr int dummy = 0; (void)dummy;
                const int piece = P0 + i * 2 + j;
                __builtin_amdgcn_sched_barrier(0);
                if (ahead) {
                    if (piece < 4) dma16(asrc[piece < 4 ? piece : 0] + kn, dA + piece * 8 * SBK);
                    else           dma16(bsrc[piece >= 4 ? piece - 4 : 0] + kn, dB + (piece - 4) * 8 * SBK);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}


__global__ __launch_bounds__(512, 2) void gemm_split_dma3_kernel(const SplitParams p) {
    __shared__ __attribute__((aligned(16))) float smem[3 * T3_STAGE];      // 147,456 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    const int nwg = p.nbm * p.nbn;
    const int bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int m0 = (vid / p.nbn) * 256, n0 = (vid % p.nbn) * 128;
    const int ktotal = p.K / SBK;
    const int kt0 = blockIdx.z * p.kt_per_split;
    const int kt1 = min(ktotal, kt0 + p.kt_per_split);

    const float* asrc[4];
    const float* bsrc[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (wave * 4 + q) * 8 + (lane >> 3);
        asrc[q] = p.A + (size_t)min(m0 + row, p.M - 1) * p.lda + ((lane & 7) ^ ((row >> 1) & 7)) * 4;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (wave * 2 + q) * 8 + (lane >> 3);
        bsrc[q] = p.B + (size_t)min(n0 + row, p.N - 1) * p.ldb + ((lane & 7) ^ ((row >> 1) & 7)) * 4;
    }
    auto issue = [&](int kt, int stage) {
        float* As = smem + stage * T3_STAGE + wave * 4 * 8 * SBK;
        float* Bs = smem + stage * T3_STAGE + T3_A + wave * 2 * 8 * SBK;
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16(asrc[q] + kt * SBK, As + q * 8 * SBK);
#pragma unroll
        for (int q = 0; q < 2; ++q) dma16(bsrc[q] + kt * SBK, Bs + q * 8 * SBK);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int fsw = (l31 >> 1) & 7;
    if (kt0 < kt1) issue(kt0, 0);
    if (kt0 + 1 < kt1) {
        issue(kt0 + 1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // Measured alternatives that LOST on this structure (same shapes, same process): a half-slice
    // stagger of waves 4-7 against their SIMD partners (-11 %: duplicated bodies, +70 VGPRs) and
    // fetching both k16 steps' fragments up front (-4 %).  Bunching the 6 DMA pieces at the loop
    // top instead of spreading them between MFMA groups: -7 %.
    Frag X, Y;
    int stage = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool ahead = kt + 2 < kt1;
        const int nstage = stage == 0 ? 2 : stage - 1;                   // (stage + 2) % 3
        float* dA = smem + nstage * T3_STAGE + wave * 4 * 8 * SBK;
        float* dB = smem + nstage * T3_STAGE + T3_A + wave * 2 * 8 * SBK;
        const int kn = (kt + 2) * SBK;
        const float* As = smem + stage * T3_STAGE;
        const float* Bs = As + T3_A;
        load_frag(X, As, Bs, wm, wn, l31, h, fsw, 0);
        mma12<0>(acc, X, ahead, asrc, bsrc, kn, dA, dB);
        load_frag(Y, As, Bs, wm, wn, l31, h, fsw, 1);
        mma12<3>(acc, Y, ahead, asrc, bsrc, kn, dA, dB);
        // retire slice kt+1 (all but the 6 youngest DMA pieces), make sure this wave's LDS reads
        // of slice kt are done (lgkmcnt) before anyone may overwrite the stage, then publish.
        if (ahead) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else       asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage = stage == 2 ? 0 : stage + 1;
    }

    const bool split = p.ksplit > 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + (wn * 2 + j) * 32 + l31;
            if (col >= p.N) continue;
            const float bv = (!split && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + (wm * 2 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row >= p.M) continue;
                float v = acc[i][j][e];
                if (split) {
                    p.slab[((size_t)blockIdx.z * p.M + row) * p.N + col] = v;
                } else {
                    v += bv;
                    float* c = p.C + (size_t)row * p.ldc + col;
                    if (p.accumulate) v += *c;
                    *c = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// TN form (wgrad): C[Mo,No] = A^T·B with A = sx8[K,Mo], B = sx8[K,No] — both operands
// reduction-major exactly as the forward/backward passes leave them (h and dz), so no
// transposed copies are materialised.  Tiles [32 k-rows][cols] are DMA'd row by row and
// the MFMA fragments (8 consecutive k of one column) are gathered by the hardware
// transposing read ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block of
// 16-bit elements and each lane receives one column of it.  An sx8 row keeps the 8 high
// (or low) parts of 8 consecutive columns in one 16-B chunk, which is exactly the 4-column
// 8-B pieces that instruction addresses.  Swizzle: chunk' = chunk ^ ((k&1) | (k&2)<<2)
// puts the 4 rows x 4 chunks a half-wave reads on 16 distinct 16-B slots (conflict-free).
// Same pipeline as gemm_split_dma3_kernel (256x128 tile, 8 waves, 3 stages, counted vmcnt).
// ---------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 tr_frag(const char* base, int off, int row_bytes) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off + 4 * row_bytes));
    return __builtin_bit_cast(f32x4, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int P0>
__device__ __forceinline__ void mma12_tn(f32x16 (&acc)[2][2], const Frag& f, bool ahead, const float* const (&asrc)[4],
                                         const float* const (&bsrc)[2], size_t ka, size_t kb, float* dA, float* dB) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.al[i]), __builtin_bit_cast(bf16x8, f.bh[j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.ah[i]), __builtin_bit_cast(bf16x8, f.bl[j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.ah[i]), __builtin_bit_cast(bf16x8, f.bh[j]), acc[i][j], 0, 0, 0);
            if (i * 2 + j < 3) {
                const int piece = P0 + i * 2 + j;
                __builtin_amdgcn_sched_barrier(0);
                if (ahead) {
                    if (piece < 4) dma16(asrc[piece < 4 ? piece : 0] + ka, dA + piece * 256);
                    else           dma16(bsrc[piece >= 4 ? piece - 4 : 0] + kb, dB + (piece - 4) * 256);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

__global__ __launch_bounds__(512, 2) void gemm_split_tn_kernel(const SplitParams p) {
    __shared__ __attribute__((aligned(16))) float smem[3 * T3_STAGE];      // A slice 32x256, B slice 32x128 floats
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    const int nwg = p.nbm * p.nbn;
    const int bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int m0 = (vid / p.nbn) * 256, n0 = (vid % p.nbn) * 128;      // p.M = Mo, p.N = No (full tiles only)
    const int ktotal = p.K / SBK;
    const int kt0 = blockIdx.z * p.kt_per_split;
    const int kt1 = min(ktotal, kt0 + p.kt_per_split);

    // DMA: A piece = one k-row of the tile (1 KB, 64 chunks); B piece = two k-rows (2 x 512 B)
    const float* asrc[4];
    const float* bsrc[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int kr = wave * 4 + q;
        const int f = (kr & 1) | ((kr & 2) << 2);
        asrc[q] = p.A + (size_t)kr * p.lda + m0 + (lane ^ f) * 4;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int kr = (wave * 2 + q) * 2 + (lane >> 5);
        const int f = (kr & 1) | ((kr & 2) << 2);
        bsrc[q] = p.B + (size_t)kr * p.ldb + n0 + ((lane & 31) ^ f) * 4;
    }
    const size_t astep = (size_t)SBK * p.lda, bstep = (size_t)SBK * p.ldb;   // floats per 32-row slice

    // fragment byte offsets inside a stage (transposing reads): lane = 16*gi + 4*q + pp
    const int gi = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int fq = (tq & 1) | ((tq & 2) << 2);
    int aoff[2], boff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c0 = (wm * 2 + i) * 32 + 16 * (gi & 1) + 4 * tp;                 // column inside the 256-wide A tile
        aoff[i] = (8 * (gi >> 1) + tq) * 1024 + (((2 * (c0 >> 3)) ^ fq) * 16) + (c0 & 7) * 2;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c0 = (wn * 2 + j) * 32 + 16 * (gi & 1) + 4 * tp;                 // column inside the 128-wide B tile
        boff[j] = (8 * (gi >> 1) + tq) * 512 + (((2 * (c0 >> 3)) ^ fq) * 16) + (c0 & 7) * 2;
    }
    auto load_frag_tn = [&](Frag& f, const char* As, const char* Bs, int s2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f.ah[i] = tr_frag(As, aoff[i] + s2 * 16 * 1024, 1024);
            f.al[i] = tr_frag(As, (aoff[i] ^ 16) + s2 * 16 * 1024, 1024);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f.bh[j] = tr_frag(Bs, boff[j] + s2 * 16 * 512, 512);
            f.bl[j] = tr_frag(Bs, (boff[j] ^ 16) + s2 * 16 * 512, 512);
        }
    };
    auto issue = [&](int kt, int stage) {
        float* As = smem + stage * T3_STAGE + wave * 4 * 256;
        float* Bs = smem + stage * T3_STAGE + T3_A + wave * 2 * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16(asrc[q] + kt * astep, As + q * 256);
#pragma unroll
        for (int q = 0; q < 2; ++q) dma16(bsrc[q] + kt * bstep, Bs + q * 256);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (kt0 < kt1) issue(kt0, 0);
    if (kt0 + 1 < kt1) {
        issue(kt0 + 1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    Frag X, Y;
    int stage = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool ahead = kt + 2 < kt1;
        const int nstage = stage == 0 ? 2 : stage - 1;
        float* dA = smem + nstage * T3_STAGE + wave * 4 * 256;
        float* dB = smem + nstage * T3_STAGE + T3_A + wave * 2 * 256;
        const size_t ka = (size_t)(kt + 2) * astep, kb = (size_t)(kt + 2) * bstep;
        const char* As = reinterpret_cast<const char*>(smem + stage * T3_STAGE);
        const char* Bs = As + T3_A * 4;
        load_frag_tn(X, As, Bs, 0);
        mma12_tn<0>(acc, X, ahead, asrc, bsrc, ka, kb, dA, dB);
        load_frag_tn(Y, As, Bs, 1);
        mma12_tn<3>(acc, Y, ahead, asrc, bsrc, ka, kb, dA, dB);
        if (ahead) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else       asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage = stage == 2 ? 0 : stage + 1;
    }

    const bool split = p.ksplit > 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + (wn * 2 + j) * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + (wm * 2 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                float v = acc[i][j][e];
                if (split) {
                    p.slab[((size_t)blockIdx.z * p.M + row) * p.N + col] = v;
                } else {
                    float* c = p.C + (size_t)row * p.ldc + col;
                    if (p.accumulate) v += *c;
                    *c = v;
                }
            }
        }
    }
}

// tile constants of the 256x256 kernels below (x16 / x16p): one operand slice = 256 rows x 32 floats = 32 KB
constexpr int T4_A = 256 * SBK, T4_STAGE = 2 * T4_A;

// ---------------------------------------------------------------------------
// 256x256x32 tile on v_mfma_f32_16x16x32_bf16.  Same LDS image, DMA and two stages as the
// 256x256 kernel above, different MFMA shape: in an MFMA-dense loop on real data this chip is
// clock-limited by power (scripts/micro/mfma_peak.hip, register-only, random operands: the
// 32x32x16 loop holds 1.89 PF, the 16x16x32 loop 2.06-2.21 PF at equal cycles per FLOP), so the
// shape that costs less energy per FLOP wins.  LDS bytes per FLOP are unchanged (128x64 per
// wave).  Each wave: 8 x 4 accumulator tiles of 16x16; per k32 slice 8 B reads up front, the
// A reads rolling one row-tile (12 MFMAs) ahead.  The MFMA takes the B fragment as its first
// operand, so a lane ends up with 4 consecutive columns of one output row: 16-B stores.
// Lane (r = lane & 15, g = lane >> 4) reads row r, chunks 2g (hi) / 2g+1 (lo); the swizzle that
// makes ds_read_b128's lane groups conflict-free for this pattern is swz16() below.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int swz16(int row) {
    const int q = (row >> 1) & 7;
    return q ^ ((((q >> 1) ^ (q >> 2)) & 1) << 1);
}
// TN form: chunk swizzle per k-row of the [32 k][256 cols] image.  A half-wave of a
// ds_read_b64_tr_b16 touches k-rows {t, 8+t : t = 0..3} (+4 for the second read) and two
// chunks that differ in bit 1, so f uses bits 0, 2, 3: 16 distinct 16-B slots per half-wave.
__device__ __forceinline__ int swz_tn16(int k) { return (k & 1) | ((k & 2) << 1) | (k & 8); }

__device__ __forceinline__ f32x4 tr_frag16(const char* base, int off) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off + 4 * 1024));
    return __builtin_bit_cast(f32x4, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// TN = false:  C[M,N] (+)= A·B^T,  A = sx8[M,K], B = sx8[N,K]   (forward, dgrad)
// TN = true :  C[M,N] (+)= A^T·B,  A = sx8[K,M], B = sx8[K,N]   (wgrad, operands as stored; M, N % 256 == 0):
//              the LDS image is [32 k-rows][256 cols] per operand and the fragments come out of
//              ds_read_b64_tr_b16: the 16 lanes of k-group g read the 4 x 16 block of k-rows 8g..8g+3
//              (then 8g+4..8g+7) and each receives its column's 4 k-values — exactly the 16x16x32 operand.
template <bool TN>
__global__ __launch_bounds__(512, 2) void gemm_split_x16_kernel(const SplitParams p) {
    // All 160 KB of LDS: THREE stages for A (the operand streamed from HBM: its slice t+2 is in flight while t
    // is multiplied) and two for B (weights / the other streamed operand: slice t+1).  Measured with in-kernel
    // stamps (round 1, instrumentation since removed) on the two-stage forward/dgrad kernel: 9-35 % of every slice was spent waiting for
    // the A pieces issued one slice earlier — HBM read latency under the kernel's own C-store traffic is more
    // than one slice (wgrad, which stores almost nothing: 2 %).
    __shared__ __attribute__((aligned(16))) float smem[5 * T4_A];          // 163,840 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                 // 2 x 4 waves: rows wm*128, cols wn*64
    const int r16 = lane & 15, g = lane >> 4;

    const int nwg = p.nbm * p.nbn;
    const int bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    int vid, zi;
    if (p.zmap) {
        // wgrad: every tile of one split-K range z reads the same rows of both operands.  Workgroups are dealt to the
        // XCDs round-robin, so give XCD k the ranges z = k, k + 8, ... with ALL their tiles: each operand slice is then
        // fetched into one L2 once and shared there (A by the nbn tiles of its row, B by the nbm tiles of its column)
        // instead of every XCD streaming the whole B operand (measured: 3.06 GB fetched per launch for 1.2 GB of operands).
        const int q = bid >> 3;
        vid = q % nwg;
        zi = (q / nwg) * 8 + xcd;
    } else {
        vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
        zi = blockIdx.z;
    }
    const int m0 = (vid / p.nbn) * 256, n0 = (vid % p.nbn) * 256;
    const int ktotal = p.K / SBK;
    const int kt0 = zi * p.kt_per_split;
    const int kt1 = min(ktotal, kt0 + p.kt_per_split);

    // DMA sources: 4 A pieces + 4 B pieces of 1 KB per wave and slice.  NT: a piece = 8 tile rows x 128 B;
    // TN: a piece = one k-row of the tile (256 cols x 4 B).  Either way piece q lands at wave*1024 + q*256 floats.
    const float* asrc[4];
    const float* bsrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (TN) {
            const int kr = wave * 4 + q;
            const int chunk = (lane ^ swz_tn16(kr)) * 4;
            asrc[q] = p.A + (size_t)kr * p.lda + m0 + chunk;
            bsrc[q] = p.B + (size_t)kr * p.ldb + n0 + chunk;
        } else {
            const int row = (wave * 4 + q) * 8 + (lane >> 3);
            const int chunk = ((lane & 7) ^ swz16(row)) * 4;
            asrc[q] = p.A + (size_t)min(m0 + row, p.M - 1) * p.lda + chunk;
            bsrc[q] = p.B + (size_t)min(n0 + row, p.N - 1) * p.ldb + chunk;
        }
    }
    const size_t astep = TN ? (size_t)SBK * p.lda : SBK, bstep = TN ? (size_t)SBK * p.ldb : SBK;   // floats per slice

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addressing.  NT: float offsets of the hi / lo chunk of row r16 (tile i adds 16 rows).
    const int fs = swz16(r16);
    const int c_hi = ((2 * g) ^ fs) * 4, c_lo = ((2 * g + 1) ^ fs) * 4;
    const int a_row = (wm * 128 + r16) * SBK, b_row = (wn * 64 + r16) * SBK;
    // TN: byte offsets of the hi chunk piece lane (g, tq, tp) addresses for tile i / j (lo = offset ^ 16)
    const int tq = r16 >> 2, tp = r16 & 3;
    const int ftn = swz_tn16(8 * g + tq);
    int a_tn[8], b_tn[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a_tn[i] = (8 * g + tq) * 1024 + ((2 * (wm * 16 + 2 * i + (tp >> 1))) ^ ftn) * 16 + (tp & 1) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) b_tn[j] = (8 * g + tq) * 1024 + ((2 * (wn * 8 + 2 * j + (tp >> 1))) ^ ftn) * 16 + (tp & 1) * 8;

    float* const smemB = smem + 3 * T4_A;
    {   // prologue: A(kt0), B(kt0), then A(kt0+1) — the wait below leaves exactly the last four in flight
        float* dA = smem + wave * 4 * 8 * SBK;
        const size_t k1 = (size_t)min(kt0 + 1, kt1 - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16_asm(asrc[q] + kt0 * astep, dA + q * 8 * SBK);
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16_asm(bsrc[q] + kt0 * bstep, smemB + wave * 4 * 8 * SBK + q * 8 * SBK);
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16_asm(asrc[q] + k1 * astep, dA + T4_A + q * 8 * SBK);
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    auto rdA = [&](const float* As, int i, bool lo) -> f32x4 {
        if (TN) return tr_frag16(reinterpret_cast<const char*>(As), lo ? (a_tn[i] ^ 16) : a_tn[i]);
        return *reinterpret_cast<const f32x4*>(As + a_row + i * 16 * SBK + (lo ? c_lo : c_hi));
    };
    auto rdB = [&](const float* Bs, int j, bool lo) -> f32x4 {
        if (TN) return tr_frag16(reinterpret_cast<const char*>(Bs), lo ? (b_tn[j] ^ 16) : b_tn[j]);
        return *reinterpret_cast<const f32x4*>(Bs + b_row + j * 16 * SBK + (lo ? c_lo : c_hi));
    };
    // Same software pipeline across the barrier as the persistent kernel (x16p_slice): the wait + barrier that publish
    // slice t+1 sit in front of row-tile 7 of slice t, and slice t+1's first fragments (B-hi into the second register
    // set, A row-tile 0 into pair 0, B-lo behind row-tile 7's B-lo group) are fetched under its twelve MFMAs.
    f32x4 bhA[4], bhB[4], bl[4], pah[2], pal[2];
    pal[0] = rdA(smem, 0, true);
#pragma unroll
    for (int j = 0; j < 4; ++j) bhA[j] = rdB(smemB, j, false);
    pah[0] = rdA(smem, 0, false);
#pragma unroll
    for (int j = 0; j < 4; ++j) bl[j] = rdB(smemB, j, true);
    __builtin_amdgcn_sched_barrier(0);

    int stage = 0, astage = 0;                              // B ring of 2, A ring of 3
    auto slice = [&](int kt, f32x4 (&bhc)[4], f32x4 (&bhn)[4]) {
        const int astage1 = astage == 2 ? 0 : astage + 1, astage2 = astage == 0 ? 2 : astage - 1;
        float* dA = smem + astage2 * T4_A + wave * 4 * 8 * SBK;            // A(kt+2)
        float* dB = smemB + (stage ^ 1) * T4_A + wave * 4 * 8 * SBK;       // B(kt+1)
        // branch-free: past the end the DMA re-fetches the last slice into stages nobody reads again
        const size_t ka = (size_t)min(kt + 2, kt1 - 1) * astep, kb = (size_t)min(kt + 1, kt1 - 1) * bstep;
        const float* As = smem + astage * T4_A;
        const float* Asn = smem + astage1 * T4_A;
        const float* Bsn = smemB + (stage ^ 1) * T4_A;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            pah[(i + 1) & 1] = rdA(As, i + 1, false);
            pal[(i + 1) & 1] = rdA(As, i + 1, true);
            __builtin_amdgcn_sched_barrier(0);
            if (i < 4) {          // B(kt+1) first, A(kt+2) last: the wait below skips exactly the 4 youngest
                if (i < 2) { dma16_asm(bsrc[2 * i] + kb, dB + (2 * i) * 8 * SBK);         dma16_asm(bsrc[2 * i + 1] + kb, dB + (2 * i + 1) * 8 * SBK); }
                else       { dma16_asm(asrc[2 * i - 4] + ka, dA + (2 * i - 4) * 8 * SBK); dma16_asm(asrc[2 * i - 3] + ka, dA + (2 * i - 3) * 8 * SBK); }
            }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 vah = __builtin_bit_cast(bf16x8, pah[i & 1]), val = __builtin_bit_cast(bf16x8, pal[i & 1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bhc[j]), val, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bl[j]), vah, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bhc[j]), vah, acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // B(kt+1) and A(kt+1) landed (all but the 4 youngest pieces = A(kt+2)); this wave's reads of the stages done
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        pal[0] = rdA(Asn, 0, true);
#pragma unroll
        for (int j = 0; j < 4; ++j) bhn[j] = rdB(Bsn, j, false);
        pah[0] = rdA(Asn, 0, false);
        __builtin_amdgcn_sched_barrier(0);
        {   // row-tile 7 (pair 1): B-lo group first, which frees the B-lo registers for the next slice's
            const bf16x8 vah = __builtin_bit_cast(bf16x8, pah[1]), val = __builtin_bit_cast(bf16x8, pal[1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[7][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bl[j]), vah, acc[7][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) bl[j] = rdB(Bsn, j, true);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[7][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bhc[j]), val, acc[7][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[7][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bhc[j]), vah, acc[7][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        stage ^= 1;
        astage = astage1;
    };
    const int npair = (kt1 - kt0) >> 1;
    int kt = kt0;
    for (int s = 0; s < npair; ++s) {
        slice(kt, bhA, bhB);
        slice(kt + 1, bhB, bhA);
        kt += 2;
    }
    if (kt < kt1) slice(kt, bhA, bhB);           // odd count: the tail's fragment reads fetch a stale stage, harmlessly
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // no LDS-DMA may outlive the workgroup

    const bool split = p.ksplit > 1;
    const bool vec = split ? (p.N % 4 == 0) : (p.ldc % 4 == 0 && ((uintptr_t)p.C % 16 == 0));
    // lane holds C[row = .. + r16][4 consecutive columns] of each (i, j) tile.  Row-tile outer, column-tile
    // inner: the two 64-B halves of every 128-B line of C leave in back-to-back stores.
    const int colw = n0 + wn * 64 + g * 4;
    f32x4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!split && p.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[j][e] = colw + j * 16 + e < p.N ? p.bias[colw + j * 16 + e] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = m0 + wm * 128 + i * 16 + r16;
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = colw + j * 16;
            if (col >= p.N) continue;
            f32x4 v = acc[i][j];
            float* c = split ? p.slab + ((size_t)zi * p.M + row) * p.N + col : p.C + (size_t)row * p.ldc + col;
            if (!split) v += bv[j];
            if (vec && col + 3 < p.N) {
                if (!split && p.accumulate) v += *reinterpret_cast<const f32x4*>(c);
                *reinterpret_cast<f32x4*>(c) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (col + e >= p.N) break;
                    float o = v[e];
                    if (!split && p.accumulate) o += c[e];
                    c[e] = o;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Persistent form of the forward / dgrad kernel above (full tiles, no split-K, no accumulate, 16-B aligned C): one
// workgroup per CU walks its XCD's tiles slot, slot + S, ... and treats their k-slices as ONE stream — the last slices
// of a tile already DMA the first A / B slices of the next, so a tile starts without the cold-miss prologue and without
// a workgroup launch.
//
// Tile boundary (round 3).  The first version stored the finished tile in a burst between two slices, and with a bias
// the compiler had put a bias load + `s_waitcnt vmcnt(0)` in front of every one of the 32 stores per lane (each store
// waited for the one before it to be acknowledged).  Now:
//   * the bias is the INITIAL value of the accumulators: the 16 values a lane needs for the next tile are fetched by
//     four hidden (inline-asm) loads at the top of the current tile's last slice — older than that slice's DMA pieces,
//     so the slice-end `vmcnt(4)` retires them — and enter the first MFMA of every accumulator as its C operand;
//   * the stores of tile t are issued INSIDE the first slice of tile t + 1, row-tile by row-tile, each group of four
//     right before the MFMAs that overwrite those accumulators.  That slice ends with `vmcnt(32)`: in the in-order
//     queue everything up to this slice's last B piece has landed (B(s+1), A(s+1)), while 28 of the 32 stores and the
//     A(s+2) pieces may still be in flight — the stores get a whole further slice before the next `vmcnt(4)` asks for
//     them, instead of standing between the pipeline and its next slice.
// ---------------------------------------------------------------------------
#if WF3D_STAMP
constexpr int STAMP_PER_WG = 1024;
__device__ unsigned long long g_stamps[256 * STAMP_PER_WG];
#endif

__device__ __forceinline__ f32x4 gload16_asm(const float* src) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(src) : "memory");    // completion: the caller's counted vmcnt
    return v;
}

// LDS-DMA piece with a wave-uniform base (SGPR pair) and a 32-bit per-lane byte offset: half the address registers of
// the flat form and no 64-bit add per piece.  `s_nop 3`: five wait states between a VALU-written SGPR (readfirstlane)
// and the VMEM instruction that reads it as its base, should the compiler have formed the base that way.
__device__ __forceinline__ void dma16_sbase(const float* sbase, unsigned voff_bytes, float* lds_wave_base) {
    const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds_wave_base;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff_bytes), "s"(sbase), "s"(lds) : "memory");
}

// One k32 slice of the 256x256 tile, software-pipelined ACROSS the barrier (round 3).  In-kernel stamps showed every
// steady-state slice taking 3,900 cycles for 3,072 cycles of MFMA: after the slice-end barrier all eight waves issue
// their first fragment reads at once and both waves of every SIMD wait for LDS.  Here the wait + barrier that publish
// slice s+1 sit in front of the LAST row-tile of slice s: by then every LDS read of slice s has been issued (the
// A fragments roll one row-tile ahead) and is drained by the same wait, so the stages of slice s are free and
// slice s+1's are valid — and the first fragments of slice s+1 (its four B-hi fragments into a second register set,
// A row-tile 0 into the rolling pair) are fetched while row-tile 7's twelve MFMAs run.  Row-tile 7 runs its B-lo group
// first, which frees the B-lo registers for slice s+1's B-lo reads behind it.  One barrier per slice, as before.
//   bhc / bhn: B-hi fragments of this / the next slice (two named sets, the caller alternates them).
template <bool BIAS>
__device__ __forceinline__ void x16p_slice(f32x4 (&acc)[8][4], f32x4 (&bhc)[4], f32x4 (&bhn)[4], f32x4 (&bl)[4], f32x4 (&pah)[2],
                                           f32x4 (&pal)[2], const float* As, const float* Bs, const float* Asn, const float* Bsn,
                                           int a_row, int b_row, int c_hi, int c_lo, const float* Ap, const float* Bp,
                                           const unsigned (&aoff)[4], const unsigned (&boff)[4], float* dA, float* dB,
                                           f32x4 (&init)[4]) {
    // A fragments alternate between two register pairs by row-tile parity (8 row-tiles: the pair that row-tile 0 of
    // the NEXT slice lands in is again pair 0 — no copies at the loop edge, so no wait for them either)
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        pah[(i + 1) & 1] = *reinterpret_cast<const f32x4*>(As + a_row + (i + 1) * 16 * SBK + c_hi);
        pal[(i + 1) & 1] = *reinterpret_cast<const f32x4*>(As + a_row + (i + 1) * 16 * SBK + c_lo);
        __builtin_amdgcn_sched_barrier(0);
        if (i < 4) {      // B(s+1) first, A(s+2) last: the wait below skips exactly the 4 youngest pieces
            if (i < 2) { dma16_sbase(Bp, boff[2 * i], dB + (2 * i) * 8 * SBK);     dma16_sbase(Bp, boff[2 * i + 1], dB + (2 * i + 1) * 8 * SBK); }
            else       { dma16_sbase(Ap, aoff[2 * i - 4], dA + (2 * i - 4) * 8 * SBK); dma16_sbase(Ap, aoff[2 * i - 3], dA + (2 * i - 3) * 8 * SBK); }
        }
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 vah = __builtin_bit_cast(bf16x8, pah[i & 1]), val = __builtin_bit_cast(bf16x8, pal[i & 1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bhc[j]), val, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bl[j]), vah, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bhc[j]), vah, acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // B(s+1) and A(s+1) landed (all but the 4 youngest pieces = A(s+2)); every read of this slice's stages is done.
    // With a bias the wait names the init registers: their first reader (the tile epilogue) is ordered behind it.
    if (BIAS) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" : "+v"(init[0]), "+v"(init[1]), "+v"(init[2]), "+v"(init[3]) :: "memory");
    else      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // first fragments of slice s+1, in the order its MFMAs will ask for them
    pal[0] = *reinterpret_cast<const f32x4*>(Asn + a_row + c_lo);
#pragma unroll
    for (int j = 0; j < 4; ++j) bhn[j] = *reinterpret_cast<const f32x4*>(Bsn + b_row + j * 16 * SBK + c_hi);
    pah[0] = *reinterpret_cast<const f32x4*>(Asn + a_row + c_hi);
    __builtin_amdgcn_sched_barrier(0);
    {   // row-tile 7 (pair 1): B-lo group first
        const bf16x8 vah = __builtin_bit_cast(bf16x8, pah[1]), val = __builtin_bit_cast(bf16x8, pal[1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[7][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bl[j]), vah, acc[7][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) bl[j] = *reinterpret_cast<const f32x4*>(Bsn + b_row + j * 16 * SBK + c_lo);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[7][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bhc[j]), val, acc[7][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[7][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bhc[j]), vah, acc[7][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool BIAS>
__global__ __launch_bounds__(512, 2) void gemm_split_x16p_kernel(const SplitParams p) {
    __shared__ __attribute__((aligned(16))) float smem[5 * T4_A];          // 163,840 B: A x 3 stages, B x 2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r16 = lane & 15, g = lane >> 4;

    // Tiles are CLAIMED, not dealt (round 3): the workgroups that share a blockIdx % 8 label (one XCD under round-robin
    // placement: speed only) draw the tiles of that label's contiguous range from an atomic counter, one tile ahead of
    // the one they compute.  A workgroup whose CU another stream's kernel holds (a collective) is dispatched late,
    // finds the range drawn and exits: k busy CUs cost k/256 of the launch instead of a whole extra round (a fixed
    // tile list per workgroup did: +45-60 %, scripts/bench_contention.py).  Which workgroup computes a tile has no
    // influence on its value.  The draw is made by one lane; its result reaches the other seven waves through a
    // mailbox word in global memory (all 160 KB of LDS are operand stages; the very first draw, before any DMA, goes
    // through LDS): draw in slice 0 of a tile, publish in slice 1, read in slice 2 — each memory operation is older
    // than its slice's DMA pieces, so the counted waits below are unchanged — first needed in slice ktotal - 2.
    const int nwg = p.nbm * p.nbn;
    const int bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int xbase = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const unsigned xcnt = xcd < r8 ? q8 + 1 : q8;
    unsigned* const ctr = p.ctl + xcd * 16;
    unsigned* const mbox = p.ctl + 128 + bid;
    if (tid == 0) reinterpret_cast<unsigned*>(smem)[0] = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned c0 = __builtin_amdgcn_readfirstlane(reinterpret_cast<const unsigned*>(smem)[0]);
    __syncthreads();
    if (c0 >= xcnt) return;
    int cur = xbase + (int)c0;                                             // tile being computed
    const int ktotal = p.K / SBK;                                          // even, >= 8 (host-checked)

    // per-lane BYTE offsets of the 4 A and 4 B pieces inside a tile (full tiles only: no row clamping; 256 rows x
    // pitch x 4 B < 2^32 is host-checked)
    unsigned aoff[4], boff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (wave * 4 + q) * 8 + (lane >> 3);
        const int chunk = ((lane & 7) ^ swz16(row)) * 4;
        aoff[q] = ((unsigned)row * (unsigned)p.lda + chunk) * 4u;
        boff[q] = ((unsigned)row * (unsigned)p.ldb + chunk) * 4u;
    }
    auto tile_m0 = [&](int id) { return (id / p.nbn) * 256; };
    auto tile_n0 = [&](int id) { return (id % p.nbn) * 256; };
    // lane (r16, g) of wave (wm, wn) holds C[m0 + wm*128 + i*16 + r16][n0 + wn*64 + j*16 + g*4 .. +3]
    const unsigned crow16 = 16u * (unsigned)p.ldc;
    const int lane_col = wn * 64 + g * 4;
    const unsigned coff = (unsigned)(wm * 128 + r16) * (unsigned)p.ldc + lane_col;

    f32x4 acc[8][4], init[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        init[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (BIAS) init[j] = *reinterpret_cast<const f32x4*>(p.bias + tile_n0(cur) + lane_col + j * 16);     // before any DMA: a plain, compiler-counted load
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = init[j];
    const int fs = swz16(r16);
    const int c_hi = ((2 * g) ^ fs) * 4, c_lo = ((2 * g + 1) ^ fs) * 4;
    const int a_row = (wm * 128 + r16) * SBK, b_row = (wn * 64 + r16) * SBK;
    float* const smemB = smem + 3 * T4_A;

    const float* Acur = p.A + (size_t)tile_m0(cur) * p.lda;                // scalar bases of the current / next tile
    const float* Bcur = p.B + (size_t)tile_n0(cur) * p.ldb;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc[0][0]) :: "memory");       // the bias loads are done before the counted DMA stream starts
    {   // prologue of the whole stream: A(0), B(0), A(1) of the first tile
        float* dA = smem + wave * 4 * 8 * SBK;
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16_sbase(Acur, aoff[q], dA + q * 8 * SBK);
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16_sbase(Bcur, boff[q], smemB + wave * 4 * 8 * SBK + q * 8 * SBK);
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16_sbase(Acur + SBK, aoff[q], dA + T4_A + q * 8 * SBK);
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // fragments of the very first slice
    f32x4 bhA[4], bhB[4], bl[4], pah[2], pal[2];
    pal[0] = *reinterpret_cast<const f32x4*>(smem + a_row + c_lo);
#pragma unroll
    for (int j = 0; j < 4; ++j) bhA[j] = *reinterpret_cast<const f32x4*>(smemB + b_row + j * 16 * SBK + c_hi);
    pah[0] = *reinterpret_cast<const f32x4*>(smem + a_row + c_hi);
#pragma unroll
    for (int j = 0; j < 4; ++j) bl[j] = *reinterpret_cast<const f32x4*>(smemB + b_row + j * 16 * SBK + c_lo);
    __builtin_amdgcn_sched_barrier(0);

    // ONE loop over the slices of all tiles this workgroup draws, two slices per trip (the two B-hi register sets)
    int stage = 0, astage = 0, kt = 0, nxt = cur;
    bool has_next = false;                                                 // known from slice 3 of a tile on
    unsigned drawn = 0;
    const float* Anext = Acur;
    const float* Bnext = Bcur;
    // C of a finished tile (wave-uniform base cb); the accumulators restart from the next tile's bias (or zero)
    auto store_tile = [&](float* cb) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float* c = cb + (coff + i * crow16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<f32x4*>(c + j * 16) = acc[i][j];
                acc[i][j] = BIAS ? init[j] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    int trip = 0;            // (slice counter of the diagnostic stamps)
    for (;; ++trip) {
        if (kt == ktotal) {
            // ---- tile `cur` is complete (checked at the TOP of the trip, so that nothing but the loop edge follows a
            // slice's tail: the compiler sinks the tail's fragment reads into whatever block comes next) ----
            store_tile(p.C + (size_t)tile_m0(cur) * p.ldc + tile_n0(cur));
            if (!has_next) break;
            kt = 0;
            cur = nxt;
            Acur = Anext; Bcur = Bnext;
            has_next = false;
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // The draw for the tile after `cur`: three hidden (inline-asm) memory operations, each issued at the top of
            // its slice — older than that slice's DMA pieces, so the slice's own vmcnt(4) retires it — and consumed
            // one slice later behind an empty asm that names the register (no extra wait anywhere; the compiler's
            // builtins for the same operations wait vmcnt(0) on the spot, draining this wave's pipeline once per tile).
            if (half == 0 && kt == 0) {
                if (tid == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(drawn) : "v"(ctr), "v"(1u) : "memory");
            }
            if (half == 1 && kt == 1) {
                asm volatile("" : "+v"(drawn));
                if (tid == 0) asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(mbox), "v"(drawn) : "memory");
            }
            if (half == 0 && kt == 2) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(drawn) : "v"(mbox) : "memory");
            if (half == 1 && kt == 3) {
                asm volatile("" : "+v"(drawn));
                const unsigned d = __builtin_amdgcn_readfirstlane(drawn);
                has_next = d < xcnt;
                nxt = has_next ? xbase + (int)d : cur;
                Anext = p.A + (size_t)tile_m0(nxt) * p.lda;
                Bnext = p.B + (size_t)tile_n0(nxt) * p.ldb;
            }
            const int astage1 = astage == 2 ? 0 : astage + 1, astage2 = astage == 0 ? 2 : astage - 1;
            float* dA = smem + astage2 * T4_A + wave * 4 * 8 * SBK;            // A(s+2)
            float* dB = smemB + (stage ^ 1) * T4_A + wave * 4 * 8 * SBK;       // B(s+1)
            // where slices s+1 / s+2 of the stream live: this tile, the next one, or (at the very end) a harmless re-fetch
            const bool b_here = kt + 1 < ktotal, a_here = kt + 2 < ktotal;
            const float* Bsrc = b_here ? Bcur : Bnext;
            const float* Asrc = a_here ? Acur : Anext;
            const int kb_i = b_here ? kt + 1 : (has_next ? 0 : ktotal - 1);
            const int ka_i = a_here ? kt + 2 : (has_next ? kt + 2 - ktotal : ktotal - 1);
            const float* Bp = Bsrc + (size_t)kb_i * SBK;
            const float* Ap = Asrc + (size_t)ka_i * SBK;
            const float* As = smem + astage * T4_A;
            const float* Bs = smemB + stage * T4_A;
            const float* Asn = smem + astage1 * T4_A;
            const float* Bsn = smemB + (stage ^ 1) * T4_A;
#if WF3D_STAMP
            if (tid == 0 && 2 * trip + half < STAMP_PER_WG) g_stamps[bid * STAMP_PER_WG + 2 * trip + half] = __builtin_amdgcn_s_memtime();    // older than this slice's DMA pieces
#endif
            if (BIAS && half == 1 && has_next && kt == ktotal - 1) {
                // the next tile's bias, requested before this slice's DMA pieces: the slice's vmcnt(4) retires it
                const float* bsrc = p.bias + tile_n0(nxt) + lane_col;
#pragma unroll
                for (int j = 0; j < 4; ++j) init[j] = gload16_asm(bsrc + j * 16);
            }
            if (half == 0) x16p_slice<BIAS>(acc, bhA, bhB, bl, pah, pal, As, Bs, Asn, Bsn, a_row, b_row, c_hi, c_lo, Ap, Bp, aoff, boff, dA, dB, init);
            else           x16p_slice<BIAS>(acc, bhB, bhA, bl, pah, pal, As, Bs, Asn, Bsn, a_row, b_row, c_hi, c_lo, Ap, Bp, aoff, boff, dA, dB, init);
            stage ^= 1;
            astage = astage1;
            ++kt;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // no LDS-DMA may outlive the workgroup
}

__global__ __launch_bounds__(256) void split_reduce_kernel(const SplitParams p) {
    const size_t total = (size_t)p.M * p.N;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int row = (int)(idx / p.N), col = (int)(idx % p.N);
        float v = 0.f;
        for (int z = 0; z < p.ksplit; ++z) v += p.slab[(size_t)z * total + idx];
        if (p.bias) v += p.bias[col];
        float* c = p.C + (size_t)row * p.ldc + col;
        if (p.accumulate) v += *c;
        *c = v;
    }
}

// The same fold for many slabs (the few-tile wgrads split K up to 256 ways): a serial walk over the slabs is one
// dependent 4-byte load per slab per thread (126 slabs of 128 KB took 55 us).  Here a wave owns 16 consecutive
// outputs; lane = (slab lane 0..15, float4 0..3) walks slabs sl, sl + 16, ... four loads in flight, and the 16
// partial sums meet in a fixed shuffle tree — the result does not depend on timing.
__global__ __launch_bounds__(256) void split_reduce_wide_kernel(const SplitParams p) {
    const size_t total = (size_t)p.M * p.N;
    const int lane = threadIdx.x & 63, sl = lane >> 2, q = lane & 3;
    const size_t idx = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + q * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool live = idx < total;
    if (live) {
        const float* base = p.slab + idx;
        for (int z = sl; z < p.ksplit; z += 64) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(base + (size_t)z * total);
            const f32x4 v1 = z + 16 < p.ksplit ? *reinterpret_cast<const f32x4*>(base + (size_t)(z + 16) * total) : zero;
            const f32x4 v2 = z + 32 < p.ksplit ? *reinterpret_cast<const f32x4*>(base + (size_t)(z + 32) * total) : zero;
            const f32x4 v3 = z + 48 < p.ksplit ? *reinterpret_cast<const f32x4*>(base + (size_t)(z + 48) * total) : zero;
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
    }
#pragma unroll
    for (int o = 4; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
    if (live && sl == 0) {
        const int row = (int)(idx / p.N), col = (int)(idx % p.N);          // N % 4 == 0: the float4 stays inside its row
        float* c = p.C + (size_t)row * p.ldc + col;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[e];
            if (p.bias) v += p.bias[col + e];
            if (p.accumulate) v += c[e];
            c[e] = v;
        }
    }
}

void launch_split_reduce(const SplitParams& p, hipStream_t st) {
    const size_t total = (size_t)p.M * p.N;
    if (p.ksplit > 16 && p.N % 4 == 0 && ((uintptr_t)p.slab % 16 == 0)) {
        hipLaunchKernelGGL(split_reduce_wide_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, p);
        return;
    }
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(split_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
}

// Kernel choice.  WF3D_SPLIT_DMA forces one: 2 = 128x128 2-stage, 3 = 256x128 3-stage,
// 4 = 256x256 2-stage (32x32x16 MFMA), 5 = 256x256 4-stage k16 register-pipelined, 6 = 256x256
// 2-stage on 16x16x32 MFMA, 7 = 128x128 two-workgroups-per-CU on 16x16x32 (measured 20 % slower than 6).  Default (unset): 6 when the output has >= 512 such tiles (the tall
// forward / dgrad GEMMs), else 3 (few tiles, long split-K reductions).  Measured on the encoder
// shapes, same process: 6 is 11-14 % faster than 4; 5 equals 4 with its DMA issued early (+5 %).
int cu_count() {
    // per device (a process may drive several); WF3D_RESERVED_CUS=r: the persistent kernel launches on r CUs fewer
    // (rounded to whole XCD-octets).  Since its tiles are claimed, not dealt, a CU held by another stream's kernel no
    // longer costs a round — the switch remains for experiments.
    static std::atomic<int> cache[16];
    static const int reserved = [] { const char* e = getenv("WF3D_RESERVED_CUS"); return e ? atoi(e) : 0; }();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    const bool cached = dev >= 0 && dev < 16;
    int v = cached ? cache[dev].load(std::memory_order_relaxed) : 0;
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (reserved > 0 && v - reserved >= 8) v = (v - reserved) / 8 * 8;
        if (cached) cache[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// 6 = 256x256 tile on v_mfma_f32_16x16x32_bf16 (gemm_split_x16_kernel / its persistent form): >= 192 tiles (3/4 of the CUs) and N a
//     multiple of 256 (a 128-wide output wastes half of every tile: 380 vs 420 us at 1,044,480 x 128 x 256);
// 3 = 256x128 tile, three LDS stages, 32x32x16 MFMA (gemm_split_dma3_kernel): the smaller ones.
// K is split only below 192 tiles: at 252 tiles (the edge MLP at cfg2) two K ranges + slabs + a combine launch cost more than
// the idle quarter-wave they fill (scripts/bench_edge_gemms.py: 62 -> 48, 52 -> 26, 36 -> 25, 86 -> 61 us for its four GEMMs).
// WF3D_SPLIT_DMA=3|6 forces one of them (tests/test_variants_gpu.py); the other round-1 variants live in
// scripts/ablation/gemm_split_variants.hip.txt.
int split_variant(int M, int N) {
    static const int forced = [] { const char* e = getenv("WF3D_SPLIT_DMA"); return e ? atoi(e) : 0; }();
    if (forced == 3 || forced == 6) return forced;
    static const int min_tiles = [] { const char* e = getenv("WF3D_X16_MIN_TILES"); return e ? atoi(e) : 192; }();
    return (long)wf3d_cdiv(M, 256) * wf3d_cdiv(N, 256) >= min_tiles && N % 256 == 0 ? 6 : 3;
}

void plan(int M, int N, int K, int& ksplit, int& kt_per) {
    const int v = split_variant(M, N);
    const int bm = 256, bn = v == 6 ? 256 : 128;
    const long tiles = (long)wf3d_cdiv(M, bm) * wf3d_cdiv(N, bn);
    const int ktotal = K / SBK;
    ksplit = 1; kt_per = ktotal;
    static const int nosplit = [] { const char* e = getenv("WF3D_SPLIT_NOSPLIT_TILES"); return e ? atoi(e) : 192; }();
    if (tiles >= nosplit || ktotal < 8) return;
    int want = (int)((512 + tiles - 1) / tiles);
    int ks = want < ktotal / 4 ? want : ktotal / 4;
    if (ks > 64) ks = 64;
    if (ks < 2) return;
    kt_per = wf3d_cdiv(ktotal, ks);
    ksplit = wf3d_cdiv(ktotal, kt_per);
}

}  // namespace

// Returns 1 when the DMA kernel can take the problem (K a multiple of the 32-wide slice,
// 16-B aligned rows); the register-staged kernel in gemm.hip handles everything else.
extern "C" int wf3d_gemm_split_dma_ok(int M, int N, int K, int lda, int ldb) {
    return M > 0 && N > 0 && K >= SBK && K % SBK == 0 && lda % 4 == 0 && ldb % 4 == 0;
}

#if WF3D_STAMP
extern "C" int wf3d_debug_stamps(unsigned long long* dst, size_t n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

static std::atomic<int> g_gemm_cus{0};           // wf3d_set_option("gemm_cus", n): workgroups of the persistent kernel (0 = one per CU)
constexpr size_t CTL_BYTES = 2048;        // persistent kernel: 8 claim counters 64 B apart (512 B, zeroed per launch) + 256+ mailbox words

extern "C" size_t wf3d_gemm_split_dma_ws_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    int ks, per;
    plan(M, N, K, ks, per);
    return ks > 1 ? (size_t)ks * M * N * sizeof(float) : CTL_BYTES;
}

extern "C" int wf3d_gemm_split_dma(const void* A_sx8, const void* B_sx8, float* C, const float* bias, int M, int N,
                                   int K, int lda, int ldb, int ldc, int accumulate, void* ws, size_t ws_bytes,
                                   void* stream) {
    WF3D_CHECK(wf3d_gemm_split_dma_ok(M, N, K, lda, ldb), WF3D_ERR_UNSUPPORTED, "wf3d_gemm_split_dma: shape not supported");
    WF3D_CHECK(A_sx8 && B_sx8 && C, WF3D_ERR_ARG, "wf3d_gemm_split_dma: null operand");
    WF3D_CHECK(lda >= K && ldb >= K && ldc >= N, WF3D_ERR_ARG, "wf3d_gemm_split_dma: leading dimension too small");
    WF3D_CHECK(((uintptr_t)A_sx8 % 16 == 0) && ((uintptr_t)B_sx8 % 16 == 0), WF3D_ERR_ARG, "wf3d_gemm_split_dma: misaligned operand");
    SplitParams p{};
    p.A = (const float*)A_sx8; p.B = (const float*)B_sx8; p.C = C; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.accumulate = accumulate;
    const int variant = split_variant(M, N);
    p.nbm = wf3d_cdiv(M, 256); p.nbn = wf3d_cdiv(N, variant == 6 ? 256 : 128);
    plan(M, N, K, p.ksplit, p.kt_per_split);
    const size_t need = p.ksplit > 1 ? (size_t)p.ksplit * M * N * sizeof(float) : 0;
    if (need && (ws == nullptr || ws_bytes < need)) { p.ksplit = 1; p.kt_per_split = K / SBK; }
    p.slab = p.ksplit > 1 ? (float*)ws : nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (variant == 6) {
        static const int persist_on = [] { const char* e = getenv("WF3D_SPLIT_PERSIST"); return e ? atoi(e) : 1; }();
        int cus = cu_count();
        { const int lim = g_gemm_cus.load(std::memory_order_relaxed); if (lim >= 8 && lim < cus) cus = lim / 8 * 8; }
        // persistent form: full tiles, one pass over K, plain (non-accumulating) 16-B stores
        if (persist_on && p.ksplit == 1 && !accumulate && M % 256 == 0 && N % 256 == 0 && K / SBK >= 8 && K % (2 * SBK) == 0 &&
            ws != nullptr && ws_bytes >= CTL_BYTES && (uintptr_t)ws % 64 == 0 && cus <= 384 && cus >= 8 && cus % 8 == 0 &&
            p.nbm * p.nbn >= 2 * cus && ldc % 4 == 0 && (long)lda * 1024 < (1L << 32) && (long)ldb * 1024 < (1L << 32) && (long)ldc * 1024 < (1L << 32) && ((uintptr_t)C % 16 == 0) && (!bias || (uintptr_t)bias % 16 == 0)) {
            p.ctl = (unsigned*)ws;
            if (hipMemsetAsync(ws, 0, 512, st) != hipSuccess) { wf3d_set_error("wf3d_gemm_split_dma: hipMemsetAsync failed"); return WF3D_ERR_LAUNCH; }
            if (bias) hipLaunchKernelGGL(gemm_split_x16p_kernel<true>, dim3(cus, 1, 1), dim3(512), 0, st, p);
            else      hipLaunchKernelGGL(gemm_split_x16p_kernel<false>, dim3(cus, 1, 1), dim3(512), 0, st, p);
        } else
            hipLaunchKernelGGL(gemm_split_x16_kernel<false>, dim3(p.nbm * p.nbn, 1, p.ksplit), dim3(512), 0, st, p);
    } else {
        hipLaunchKernelGGL(gemm_split_dma3_kernel, dim3(p.nbm * p.nbn, 1, p.ksplit), dim3(512), 0, st, p);
    }
    WF3D_LAUNCH_CHECK();
    if (p.ksplit > 1) {
        launch_split_reduce(p, st);
        WF3D_LAUNCH_CHECK();
    }
    return WF3D_OK;
}

// C[Mo,No] (+)= A^T·B, A = sx8[K,Mo], B = sx8[K,No]  (Linear wgrad: A = dY, B = X, both as stored)
extern "C" int wf3d_gemm_split_tn_ok(int Mo, int No, int K, int lda, int ldb) {
    return Mo > 0 && No > 0 && K >= SBK && Mo % 256 == 0 && No % 128 == 0 && K % SBK == 0 && lda % 4 == 0 && ldb % 4 == 0;
}

namespace {
std::atomic<int> g_tn_rounds{[] { const char* e = getenv("WF3D_TN_ROUNDS"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v > 8 ? 8 : v; }()};
}  // namespace

// Run-time switches (process-wide).  "tn_rounds" = workgroups per CU the 16+-tile wgrad launches are cut into (1..8).
extern "C" int wf3d_set_option(const char* name, int value) {
    WF3D_CHECK(name != nullptr, WF3D_ERR_ARG, "wf3d_set_option: null name");
    if (strcmp(name, "tn_rounds") == 0) {
        WF3D_CHECK(value >= 1 && value <= 8, WF3D_ERR_ARG, "wf3d_set_option: tn_rounds must be 1..8 (got %d)", value);
        g_tn_rounds.store(value, std::memory_order_relaxed);
        return WF3D_OK;
    }
    if (strcmp(name, "gemm_cus") == 0) {
        WF3D_CHECK(value == 0 || (value >= 8 && value <= 1024), WF3D_ERR_ARG, "wf3d_set_option: gemm_cus must be 0 (all) or 8..1024 (got %d)", value);
        g_gemm_cus.store(value, std::memory_order_relaxed);
        return WF3D_OK;
    }
    wf3d_set_error("wf3d_set_option: unknown option '%s'", name);
    return WF3D_ERR_ARG;
}

namespace {
// Wgrad kernel choice: 256x256 tiles on the 16x16x32 MFMA when the output allows (WF3D_TN16=0 forces
// the 256x128 32x32x16 kernel).
bool tn16(int Mo, int No) {
    static const int off = [] { const char* e = getenv("WF3D_TN16"); return e && atoi(e) == 0; }();
    return !off && Mo % 256 == 0 && No % 256 == 0;
}
void plan_tn(int Mo, int No, int K, int& ksplit, int& kt_per) {
    const bool big = tn16(Mo, No);
    const long tiles = (long)(Mo / 256) * (No / (big ? 256 : 128));
    const int ktotal = K / SBK;
    ksplit = 1; kt_per = ktotal;
    if (tiles >= 256 || ktotal < 8) return;
    // one workgroup per CU either way; the big tile runs one full wave of 256 workgroups.  WF3D_TN_ROUNDS=f deals f
    // workgroups per CU (K ranges f times shorter, f times the slabs): a CU held by another stream's kernel then costs
    // the launch 1/f of its time instead of all of it, at the price of the extra slab traffic.
    // Measured (scripts/bench_contention.py, scripts/ab_gemm.sh): the 32-tile wgrads (2048 x 1024) pay +2.5 % / +4 % for
    // f = 2 / 4 on an idle chip and lose 15 % / 0 % instead of 67 % beside a kernel that holds CUs; the 8-tile ones
    // (already 32 ranges of 2 MB slabs) pay +9 % / +28 %, so f applies from 16 tiles up.  wf3d_set_option("tn_rounds", f);
    // wf3d.dist turns it on (2) when gradients are reduced beside the backward pass.
    const int rounds = tiles >= 16 ? g_tn_rounds.load(std::memory_order_relaxed) : 1;
    int want = (int)(((big ? 256 : 512) * rounds + tiles - 1) / tiles);
    // One or two output tiles (the edge MLP's 128 x 256 and 256 x 512 weights over ~10^5 - 10^6 edge rows) are a streaming
    // reduction over K: they get a K range per CU (up to 256 slabs) as long as a range keeps >= 8 slices; more tiles
    // keep the 64-range cap that bounds the slab traffic (with 64 ranges the 128 x 256 wgrad of cfg5 ran on 64 CUs
    // at 2.0 TB/s, the 256 x 512 one on half the chip).
    const bool few = tiles <= 4;
    int ks = want < ktotal / (few ? 8 : 4) ? want : ktotal / (few ? 8 : 4);
    static const int few_cap = [] { const char* e = getenv("WF3D_TN_FEW_CAP"); return e ? atoi(e) : 256; }();
    const int cap = few ? few_cap : 64 * rounds;
    if (ks > cap) ks = cap;
    if (few && ks >= 8) ks &= ~7;                 // multiples of 8: the XCD-mapped launch order applies
    if (ks < 2) return;
    kt_per = wf3d_cdiv(ktotal, ks);
    ksplit = wf3d_cdiv(ktotal, kt_per);
}
}  // namespace

extern "C" size_t wf3d_gemm_split_tn_ws_bytes(int Mo, int No, int K) {
    if (Mo <= 0 || No <= 0 || K <= 0) return 0;
    int ks, per;
    plan_tn(Mo, No, K, ks, per);
    return ks > 1 ? (size_t)ks * Mo * No * sizeof(float) : 0;
}

extern "C" int wf3d_gemm_split_tn(const void* A_sx8, const void* B_sx8, float* C, int Mo, int No, int K, int lda,
                                  int ldb, int ldc, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    WF3D_CHECK(wf3d_gemm_split_tn_ok(Mo, No, K, lda, ldb), WF3D_ERR_UNSUPPORTED,
               "wf3d_gemm_split_tn: needs Mo %% 256 == 0, No %% 128 == 0, K %% 32 == 0 (Mo=%d No=%d K=%d)", Mo, No, K);
    WF3D_CHECK(A_sx8 && B_sx8 && C, WF3D_ERR_ARG, "wf3d_gemm_split_tn: null operand");
    WF3D_CHECK(lda >= Mo && ldb >= No && ldc >= No, WF3D_ERR_ARG, "wf3d_gemm_split_tn: leading dimension too small");
    WF3D_CHECK(((uintptr_t)A_sx8 % 16 == 0) && ((uintptr_t)B_sx8 % 16 == 0), WF3D_ERR_ARG, "wf3d_gemm_split_tn: misaligned operand");
    SplitParams p{};
    p.A = (const float*)A_sx8; p.B = (const float*)B_sx8; p.C = C; p.bias = nullptr;
    p.M = Mo; p.N = No; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.accumulate = accumulate;
    const bool big = tn16(Mo, No);
    p.nbm = Mo / 256; p.nbn = No / (big ? 256 : 128);
    plan_tn(Mo, No, K, p.ksplit, p.kt_per_split);
    const size_t need = p.ksplit > 1 ? (size_t)p.ksplit * Mo * No * sizeof(float) : 0;
    if (need && (ws == nullptr || ws_bytes < need)) { p.ksplit = 1; p.kt_per_split = K / SBK; }
    p.slab = p.ksplit > 1 ? (float*)ws : nullptr;
    hipStream_t st = (hipStream_t)stream;
    static const int zmap_off = [] { const char* e = getenv("WF3D_TN_ZMAP"); return e && atoi(e) == 0; }();
    if (big && !zmap_off && p.ksplit >= 8 && p.ksplit % 8 == 0) {
        p.zmap = 1;
        hipLaunchKernelGGL(gemm_split_x16_kernel<true>, dim3(p.nbm * p.nbn * p.ksplit, 1, 1), dim3(512), 0, st, p);
    } else if (big) hipLaunchKernelGGL(gemm_split_x16_kernel<true>, dim3(p.nbm * p.nbn, 1, p.ksplit), dim3(512), 0, st, p);
    else     hipLaunchKernelGGL(gemm_split_tn_kernel, dim3(p.nbm * p.nbn, 1, p.ksplit), dim3(512), 0, st, p);
    WF3D_LAUNCH_CHECK();
    if (p.ksplit > 1) {
        launch_split_reduce(p, st);
        WF3D_LAUNCH_CHECK();
    }
    return WF3D_OK;
}
