// fp32 MFMA GEMM for gfx950 with fused LayerNorm/activation prologue.
//
//   C[M,N] = pro(A)·B (+ bias[n]) (+ addend[m,n]) (+ C)
//
// One workgroup = 256 threads = 4 wave64; each wave owns TM x TN tiles of
// 32x32 built from v_mfma_f32_32x32x2_f32 (exact f32 FMA chain, 64 FLOP/clk/SIMD,
// MI355X_MICROARCH.md "Matrix cores").  K is walked in BK=32 slices that are
// staged global -> registers -> LDS with a two-stage LDS ring: the global loads
// of slices t+1 .. t+R (R = 2 or 3 register sets) are in flight under the MFMAs
// of slice t; slice t+1 is written to the other LDS stage after them
// (async-STAGE split, cdna_hip_programming.md T14), one barrier per slice.
// With wf3d_gemm_t.x3 the same kernel multiplies in bf16x3: the staging pass
// splits each fp32 value into bf16 (hi, lo) and the loop issues three
// v_mfma_f32_32x32x16_bf16 per product.
//
// LDS images (per operand, chosen by its global layout so that global reads are
// always 16-B coalesced and no transposition is ever needed):
//   KC  k-contiguous operand  [rows][36]   -> one ds_read_b128 per 4 MFMAs;
//       the 36-float row stride puts the 16 lanes of every ds_read_b128 lane
//       group on 16 distinct 16-B slots (conflict-free, MI355X_MICROARCH.md §LDS)
//   RC  row-contiguous operand [32][rows]  -> four ds_read_b32 (lanes read
//       consecutive rows: conflict-free)
// Both images deliver k = 8*ks + 4*(lane>>5) + j to MFMA j of k-step ks, so any
// A image pairs with any B image.
//
// The LayerNorm-apply + ReLU/GELU (+dropout) of the *previous* layer is fused
// into the staging pass of the activation operand, so normalised activations
// are never written to HBM: only pre-LN z and per-row (mu, rstd) exist.
#include <stdlib.h>

#include <type_traits>

#include "wf3d_common.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = 36;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct GemmParams {
    const float* A; const float* B; float* C;
    const float* bias; const float* addend;
    int M, N, K, lda, ldb, ldc, ld_addend;
    const float* pmu; const float* prs; const float* pgam; const float* pbet;
    int has_ln, has_affine;
    uint32_t drop_seed, drop_thresh; float drop_scale;
    int accumulate;
    int ksplit, kt_per_split;
    float* slab;
    int vecA, vecB, vecP;
    int nbm, nbn;
    const float* lr_u; const float* lr_v; int lr_k, ld_lr_u, ld_lr_v;    // epilogue: C += U[M, lr_k] . V[N, lr_k]^T
};

// ---- staging: global -> registers ------------------------------------------
template <int ROWS, bool KC>
__device__ __forceinline__ void load_tile(const float* __restrict__ base, int ld, int row0, int nrows,
                                          int k0, int kend, int vec, int tid, f32x4 (&v)[ROWS / 32]) {
    constexpr int NV = ROWS / 32;
    if (KC) {
        const int k = k0 + (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int grow = row0 + (tid >> 3) + 32 * i;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (grow < nrows && k < kend) {
                const float* p = base + (size_t)grow * ld + k;
                if (vec && k + 3 < kend) {
                    t = *reinterpret_cast<const f32x4*>(p);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (k + j < kend) t[j] = p[j];
                }
            }
            v[i] = t;
        }
    } else {
        constexpr int TPR = ROWS / 4;       // threads per k-row
        constexpr int KPP = 256 / TPR;      // k-rows per pass
        const int grow = row0 + (tid % TPR) * 4;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int k = k0 + tid / TPR + KPP * i;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (k < kend && grow < nrows) {
                const float* p = base + (size_t)k * ld + grow;
                if (vec && grow + 3 < nrows) {
                    t = *reinterpret_cast<const f32x4*>(p);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (grow + j < nrows) t[j] = p[j];
                }
            }
            v[i] = t;
        }
    }
}

// ---- prologue: v' = drop(act(LN-affine(v))) on in-bounds elements, 0 elsewhere
template <int ACT>
__device__ __forceinline__ float pro_one(float v, float mu, float rs, float g, float b, const GemmParams& p,
                                         uint32_t grow, uint32_t gcol, bool inb) {
    float y = p.has_ln ? (v - mu) * rs : v;
    if (p.has_affine) y = y * g + b;
    y = wf3d_act<ACT>(y);
    if (p.drop_thresh) y = wf3d_keep(p.drop_seed, grow, gcol, p.drop_thresh) ? y * p.drop_scale : 0.f;
    return inb ? y : 0.f;
}

// ---- staging: registers -> LDS ----------------------------------------------
template <int ROWS, bool KC>
__device__ __forceinline__ void store_tile(float* lds, int tid, const f32x4 (&v)[ROWS / 32]) {
    constexpr int NV = ROWS / 32;
    if (KC) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            *reinterpret_cast<f32x4*>(&lds[((tid >> 3) + 32 * i) * LDK + (tid & 7) * 4]) = v[i];
    } else {
        constexpr int TPR = ROWS / 4;
        constexpr int KPP = 256 / TPR;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            *reinterpret_cast<f32x4*>(&lds[(tid / TPR + KPP * i) * ROWS + (tid % TPR) * 4]) = v[i];
    }
}

// ---- staging, X3 mode: split fp32 registers into bf16 (hi, lo) pairs and write the k-contiguous "sx8" image
// [rows][36 floats], each 32-B group = [8 x hi | 8 x lo] of 8 consecutive k — the image the bf16x3 MFMA loop reads.
// A k-contiguous operand gives each thread 4 consecutive k of one row (two 8-B writes).  A row-contiguous operand
// gives 4 consecutive rows at one k: it keeps its orientation — a bf16 plane [32 k][ROWS] of high parts followed by
// one of low parts, k-row pitch ROWS*2 + 16 bytes (the four k-rows of a transposing read then sit on distinct
// banks) — and is transposed by the fragment read (x3_tr_frag), not here.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int x3_ks(int rows) { return rows * 2 + 16; }       // bytes per k-row of a plane; 2 planes fit rows * LDK floats
template <int ROWS, bool KC>
__device__ __forceinline__ void store_tile_x3(float* lds, int tid, const f32x4 (&v)[ROWS / 32]) {
    constexpr int NV = ROWS / 32;
    if (KC) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            bf16x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hi[j] = (__bf16)v[i][j];
                lo[j] = (__bf16)(v[i][j] - (float)hi[j]);
            }
            float* g = &lds[((tid >> 3) + 32 * i) * LDK + ((tid & 7) >> 1) * 8 + (tid & 1) * 2];
            *reinterpret_cast<bf16x4*>(g) = hi;
            *reinterpret_cast<bf16x4*>(g + 4) = lo;
        }
    } else {
        constexpr int TPR = ROWS / 4;
        constexpr int KPP = 256 / TPR;
        constexpr int KS = x3_ks(ROWS);
        char* lc = reinterpret_cast<char*>(lds);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            bf16x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hi[j] = (__bf16)v[i][j];
                lo[j] = (__bf16)(v[i][j] - (float)hi[j]);
            }
            char* g = lc + (tid / TPR + KPP * i) * KS + (tid % TPR) * 8;
            *reinterpret_cast<bf16x4*>(g) = hi;
            *reinterpret_cast<bf16x4*>(g + BK * KS) = lo;
        }
    }
}

// X3 fragment of a row-contiguous operand: the 16 lanes of a group read a 4 k x 16 rows block of the plane with
// ds_read_b64_tr_b16 (lane L addresses k-row L >> 2, rows 4 (L & 3) .. + 3) and each receives its own row's 4
// k-values; two reads give the 8 consecutive k of the 32x32x16 operand.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 x3_tr_frag(const char* base, int off, int ks) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off + 4 * ks));
    return __builtin_bit_cast(f32x4, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// ---- fragment read: 4 consecutive-k values of one 32-row tile for this lane --
template <int ROWS, bool KC>
__device__ __forceinline__ f32x4 read_frag(const float* lds, int row, int ks, int h) {
    if (KC) {
        return *reinterpret_cast<const f32x4*>(&lds[row * LDK + ks * 8 + 4 * h]);
    } else {
        f32x4 r;
        const float* p = &lds[(ks * 8 + 4 * h) * ROWS + row];
        r[0] = p[0]; r[1] = p[ROWS]; r[2] = p[2 * ROWS]; r[3] = p[3 * ROWS];
        return r;
    }
}

// branch-free staging loads for tiles that are completely inside the matrices
template <int ROWS, bool KC>
__device__ __forceinline__ void load_tile_fast(const float* __restrict__ p0, int ld, int k0, f32x4 (&v)[ROWS / 32]) {
    constexpr int NV = ROWS / 32;
    if (KC) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const f32x4*>(p0 + (size_t)(32 * i) * ld + k0);
    } else {
        constexpr int KPP = 256 / (ROWS / 4);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const f32x4*>(p0 + (size_t)(k0 + KPP * i) * ld);
    }
}

// compile-time unrolled steps u = U .. N-1 (register sets are indexed by u and must stay in registers); a step
// returning false ends the sequence
template <int U, int N, class F>
__device__ __forceinline__ void static_steps(F&& f) {
    if constexpr (U < N) {
        if (!f(std::integral_constant<int, U>{})) return;
        static_steps<U + 1, N>(f);
    }
}

template <int WAVES_M, int WAVES_N, int TM, int TN, bool A_KC, bool B_KC, int ACT, bool PRO, bool FAST, bool SPLIT, bool X3, int R>
__device__ __forceinline__ void gemm_body(const GemmParams& p, float* smem, int m0, int n0, int kt0, int kt1) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int A_SZ = (A_KC || X3) ? BM * LDK : BK * BM;
    constexpr int B_SZ = (B_KC || X3) ? BN * LDK : BK * BN;
    constexpr int NVA = BM / 32, NVB = BN / 32;
    constexpr bool PRO_A = PRO && A_KC;     // NT: activation operand is A [M,K]
    constexpr bool PRO_B = PRO && !A_KC;    // TN: activation operand is B [K,N]
    constexpr int TPRB = BN / 4, KPPB = 256 / TPRB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int h = lane >> 5, l31 = lane & 31;
    const int l15 = lane & 15, hr = (lane >> 4) & 1;        // X3 transposing reads: lane within its 16-group, row half

    // per-thread staging base pointers (FAST path)
    const float* pa0 = A_KC ? p.A + (size_t)(m0 + (tid >> 3)) * p.lda + (tid & 7) * 4
                            : p.A + (size_t)(tid / (BM / 4)) * p.lda + m0 + (tid % (BM / 4)) * 4;
    const float* pb0 = B_KC ? p.B + (size_t)(n0 + (tid >> 3)) * p.ldb + (tid & 7) * 4
                            : p.B + (size_t)(tid / TPRB) * p.ldb + n0 + (tid % TPRB) * 4;

    // loop-invariant prologue parameters
    float mu_a[NVA], rs_a[NVA];
    f32x4 gam_b = {1.f, 1.f, 1.f, 1.f}, bet_b = {0.f, 0.f, 0.f, 0.f};
    if (PRO_A) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int grow = m0 + (tid >> 3) + 32 * i;
            const bool ok = p.has_ln && grow < p.M;
            mu_a[i] = ok ? p.pmu[grow] : 0.f;
            rs_a[i] = ok ? p.prs[grow] : 1.f;
        }
    }
    if (PRO_B && p.has_affine) {
        const int gcol = n0 + (tid % TPRB) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (gcol + j < p.N) { gam_b[j] = p.pgam[gcol + j]; bet_b[j] = p.pbet[gcol + j]; }
    }

    // R register sets: the global loads of slices t+1 .. t+R are in flight while slice t is multiplied (the launches
    // this kernel serves run one or two workgroups per CU, so a slice's load latency — about 1.3 us measured,
    // whatever the MFMA shape — is hidden by depth, not by occupancy).  LDS keeps two stages.
    struct Regs {
        f32x4 ra[NVA], rb[NVB];
        f32x4 pg, pbv;                  // PRO_A: gamma/beta of this slice's k columns
        float mu_b[NVB], rs_b[NVB];     // PRO_B: stats of this slice's k rows
    };
    Regs S[R];
    auto fetch = [&](int kt, Regs& r) {
        const int k0 = kt * BK;
        if (FAST) {
            load_tile_fast<BM, A_KC>(pa0, p.lda, k0, r.ra);
            load_tile_fast<BN, B_KC>(pb0, p.ldb, k0, r.rb);
        } else {
            load_tile<BM, A_KC>(p.A, p.lda, m0, p.M, k0, p.K, p.vecA, tid, r.ra);
            load_tile<BN, B_KC>(p.B, p.ldb, n0, p.N, k0, p.K, p.vecB, tid, r.rb);
        }
        if (PRO_A) {
            r.pg = f32x4{1.f, 1.f, 1.f, 1.f};
            r.pbv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.has_affine) {
                const int k = k0 + (tid & 7) * 4;
                if (FAST) {
                    r.pg = *reinterpret_cast<const f32x4*>(p.pgam + k);
                    r.pbv = *reinterpret_cast<const f32x4*>(p.pbet + k);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        r.pg[j] = k + j < p.K ? p.pgam[k + j] : 1.f;
                        r.pbv[j] = k + j < p.K ? p.pbet[k + j] : 0.f;
                    }
                }
            }
        }
        if (PRO_B) {
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                const int k = k0 + tid / TPRB + KPPB * i;          // = row of the activation matrix
                const bool ok = p.has_ln && (FAST || k < p.K);
                r.mu_b[i] = ok ? p.pmu[k] : 0.f;
                r.rs_b[i] = ok ? p.prs[k] : 1.f;
            }
        }
    };
    auto commit = [&](int kt, Regs& r, int stage) {
        float* As = smem + stage * (A_SZ + B_SZ);
        float* Bs = As + A_SZ;
        if (PRO_A) {
            const int k = kt * BK + (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                const int grow = m0 + (tid >> 3) + 32 * i;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    r.ra[i][j] = pro_one<ACT>(r.ra[i][j], mu_a[i], rs_a[i], r.pg[j], r.pbv[j], p, grow, k + j,
                                              FAST || (grow < p.M && k + j < p.K));
            }
        }
        if (PRO_B) {
            const int gcol = n0 + (tid % TPRB) * 4;
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                const int k = kt * BK + tid / TPRB + KPPB * i;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    r.rb[i][j] = pro_one<ACT>(r.rb[i][j], r.mu_b[i], r.rs_b[i], gam_b[j], bet_b[j], p, k, gcol + j,
                                              FAST || (k < p.K && gcol + j < p.N));
            }
        }
        if (X3) {
            store_tile_x3<BM, A_KC>(As, tid, r.ra);
            store_tile_x3<BN, B_KC>(Bs, tid, r.rb);
        } else {
            store_tile<BM, A_KC>(As, tid, r.ra);
            store_tile<BN, B_KC>(Bs, tid, r.rb);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto multiply = [&](int stage) {
        const float* As = smem + stage * (A_SZ + B_SZ);
        const float* Bs = As + A_SZ;
        if (SPLIT || X3) {
            // bf16x3 split precision: each 32-B group of a staged row holds [8 x hi | 8 x lo] bf16 of 8
            // consecutive k; a*b ~= ah*bh + ah*bl + al*bh (fp32 accumulate), v_mfma_f32_32x32x16_bf16.
#pragma unroll
            for (int s2 = 0; s2 < BK / 16; ++s2) {
                f32x4 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (X3 && !A_KC) {
                        constexpr int KS = x3_ks(BM);
                        const int off = (16 * s2 + 8 * h + (l15 >> 2)) * KS + ((wm * TM + i) * 32 + 16 * hr + (l15 & 3) * 4) * 2;
                        ah[i] = x3_tr_frag(reinterpret_cast<const char*>(As), off, KS);
                        al[i] = x3_tr_frag(reinterpret_cast<const char*>(As), off + BK * KS, KS);
                    } else {
                        const float* pr = As + ((wm * TM + i) * 32 + l31) * LDK + (2 * s2 + h) * 8;
                        ah[i] = *reinterpret_cast<const f32x4*>(pr);
                        al[i] = *reinterpret_cast<const f32x4*>(pr + 4);
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (X3 && !B_KC) {
                        constexpr int KS = x3_ks(BN);
                        const int off = (16 * s2 + 8 * h + (l15 >> 2)) * KS + ((wn * TN + j) * 32 + 16 * hr + (l15 & 3) * 4) * 2;
                        bh[j] = x3_tr_frag(reinterpret_cast<const char*>(Bs), off, KS);
                        bl[j] = x3_tr_frag(reinterpret_cast<const char*>(Bs), off + BK * KS, KS);
                    } else {
                        const float* pr = Bs + ((wn * TN + j) * 32 + l31) * LDK + (2 * s2 + h) * 8;
                        bh[j] = *reinterpret_cast<const f32x4*>(pr);
                        bl[j] = *reinterpret_cast<const f32x4*>(pr + 4);
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl[j]), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < BK / 8; ++ks) {
                f32x4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = read_frag<BM, A_KC>(As, (wm * TM + i) * 32 + l31, ks, h);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = read_frag<BN, B_KC>(Bs, (wn * TN + j) * 32 + l31, ks, h);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
            }
        }
    };

    const int nk = kt1 - kt0;
    if (nk > 0) {
        static_steps<0, R>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if (u < nk) fetch(kt0 + u, S[u]);
            return true;
        });
        commit(kt0, S[0], 0);
    }
    __syncthreads();

    if (nk <= R) {
        // short reduction (the K = 3 coordinate products, split-K tails): every slice is already in flight
        static_steps<0, R>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if (u >= nk) return false;
            multiply(u & 1);
            if (u + 1 < nk) commit(kt0 + u + 1, S[(u + 1) % R], (u + 1) & 1);
            __syncthreads();
            return true;
        });
    } else {
        // past the end the fetch is clamped to the last slice (a harmless re-read that is never committed): the
        // loop body stays branch-free around the loads, so the compiler's vmcnt waits stay counted
        for (int s0 = 0; s0 < nk; s0 += R) {
            static_steps<0, R>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const int cur = s0 + u;
                if (cur >= nk) return false;
                fetch(min(kt0 + cur + R, kt1 - 1), S[u]);       // set u was committed one slice ago
                multiply(cur & 1);
                if (cur + 1 < nk) commit(kt0 + cur + 1, S[(u + 1) % R], (cur + 1) & 1);
                __syncthreads();
                return true;
            });
        }
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const bool split = p.ksplit > 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + l31;
            if (col >= p.N) continue;
            const float bv = (!split && p.bias) ? p.bias[col] : 0.f;
            float lv[4] = {0.f, 0.f, 0.f, 0.f};             // this column's row of V (low-rank epilogue term, lr_k <= 4)
            if (!split && p.lr_k > 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) lv[k] = k < p.lr_k ? p.lr_v[(size_t)col * p.ld_lr_v + k] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row >= p.M) continue;
                float v = acc[i][j][e];
                if (split) {
                    p.slab[((size_t)blockIdx.z * p.M + row) * p.N + col] = v;
                } else {
                    v += bv;
                    if (p.lr_k > 0) {
                        const float* u = p.lr_u + (size_t)row * p.ld_lr_u;
                        float t = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) t += k < p.lr_k ? u[k] * lv[k] : 0.f;
                        v += t;
                    }
                    if (p.addend) v += p.addend[(size_t)row * p.ld_addend + col];
                    float* c = p.C + (size_t)row * p.ldc + col;
                    if (p.accumulate) v += *c;
                    *c = v;
                }
            }
        }
    }
}

template <int WAVES_M, int WAVES_N, int TM, int TN, bool A_KC, bool B_KC, int ACT, bool PRO, bool SPLIT = false, bool X3 = false>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int A_SZ = (A_KC || X3) ? BM * LDK : BK * BM;
    constexpr int B_SZ = (B_KC || X3) ? BN * LDK : BK * BN;
    // register sets of slices in flight (gemm_body): what fits 256 VGPRs without spilling next to the accumulators
    constexpr int R = TM * TN == 1 ? 3 : (PRO ? 1 : 2);
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_SZ + B_SZ)];

    // XCD-aware tile order: blocks bid, bid+8, ... share an XCD (round-robin
    // dispatch); give each XCD a contiguous run of tiles, column tiles fastest,
    // so the A row-panel of a tile row is fetched into ONE L2 (bijective form,
    // cdna_hip_programming.md T1).
    const int nwg = p.nbm * p.nbn;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int m0 = (vid / p.nbn) * BM, n0 = (vid % p.nbn) * BN;

    const int ktotal = (p.K + BK - 1) / BK;
    const int kt0 = blockIdx.z * p.kt_per_split;
    const int kt1 = min(ktotal, kt0 + p.kt_per_split);

    // FAST: every staged tile of this workgroup lies fully inside A and B and is
    // 16-B loadable -> branch-free staging (block-uniform choice)
    const bool fast = p.vecA && p.vecB && p.vecP && (m0 + BM <= p.M) && (n0 + BN <= p.N) && (kt1 * BK <= p.K);
    if (fast) gemm_body<WAVES_M, WAVES_N, TM, TN, A_KC, B_KC, ACT, PRO, true, SPLIT, X3, R>(p, smem, m0, n0, kt0, kt1);
    else      gemm_body<WAVES_M, WAVES_N, TM, TN, A_KC, B_KC, ACT, PRO, false, SPLIT, X3, R>(p, smem, m0, n0, kt0, kt1);
}

// split-K combine: deterministic slab sum + epilogue terms
__global__ __launch_bounds__(256) void gemm_splitk_reduce(const GemmParams p) {
    const size_t total = (size_t)p.M * p.N;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int row = (int)(idx / p.N), col = (int)(idx % p.N);
        float v = 0.f;
        for (int z = 0; z < p.ksplit; ++z) v += p.slab[(size_t)z * total + idx];
        if (p.bias) v += p.bias[col];
        for (int k = 0; k < p.lr_k; ++k) v += p.lr_u[(size_t)row * p.ld_lr_u + k] * p.lr_v[(size_t)col * p.ld_lr_v + k];
        if (p.addend) v += p.addend[(size_t)row * p.ld_addend + col];
        float* c = p.C + (size_t)row * p.ldc + col;
        if (p.accumulate) v += *c;
        *c = v;
    }
}

// The same combine for many slabs (few output tiles, up to 256 K ranges): a wave owns 16 consecutive outputs, lane =
// (slab lane 0..15, float4 0..3) walks slabs sl, sl + 16, ... with four loads in flight, the 16 partial sums meet in
// a fixed shuffle tree (deterministic).
__global__ __launch_bounds__(256) void gemm_splitk_reduce_wide(const GemmParams p) {
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
            for (int k = 0; k < p.lr_k; ++k) v += p.lr_u[(size_t)row * p.ld_lr_u + k] * p.lr_v[(size_t)(col + e) * p.ld_lr_v + k];
            if (p.addend) v += p.addend[(size_t)row * p.ld_addend + col + e];
            if (p.accumulate) v += c[e];
            c[e] = v;
        }
    }
}

struct SplitPlan { int ksplit, kt_per; };

SplitPlan plan_split(int M, int N, int K, int bm, int bn) {
    const long tiles = (long)wf3d_cdiv(M, bm) * wf3d_cdiv(N, bn);
    const int ktotal = wf3d_cdiv(K, BK);
    SplitPlan s{1, ktotal};
    if (tiles >= 256 || ktotal < 8) return s;
    int want = (int)((512 + tiles - 1) / tiles);
    int ks = want < ktotal / 4 ? want : ktotal / 4;
    // Few tiles and a very long reduction (first-layer wgrad: 512 x 8 outputs over K = B*N rows) stream
    // their operands from HBM: they need >= 2 workgroups per CU in flight, slabs stay tiny.
    const int cap = tiles <= 8 ? 256 : 64;
    if (ks > cap) ks = cap;
    if (ks < 2) return s;
    s.kt_per = wf3d_cdiv(ktotal, ks);
    s.ksplit = wf3d_cdiv(ktotal, s.kt_per);      // no empty split
    return s;
}

inline bool small_m(int M) { return M <= 64; }
// A few thousand rows (the edge head's per-vertex Linears, M = sum of vertex counts): 128 x 128 tiles give fewer than
// one workgroup per CU and then need split-K slabs plus a reduce launch; 64 x 64 tiles (one MFMA tile per wave) fill
// the chip directly (the 2048 x 1536 x 512 in-projection: 36 -> 21 us in x3, 57 -> 36 us in fp32).  Only when the
// reduction is short enough that the halved operand reuse does not matter.
inline bool mid_tile(int M, int N, int K) {
    const long t128 = (long)wf3d_cdiv(M, 128) * wf3d_cdiv(N, 128), t64 = (long)wf3d_cdiv(M, 64) * wf3d_cdiv(N, 64);
    static const int lim = [] { const char* e = getenv("WF3D_MID_T128"); return e ? atoi(e) : 256; }();
    return M > 64 && t128 < lim && t64 >= 128 && K <= 2048;
}

template <int WM, int WN, int TM, int TN, bool AKC, bool BKC, int ACT, bool PRO, bool X3 = false>
void launch(const GemmParams& p, hipStream_t st) {
    dim3 grid(p.nbm * p.nbn, 1, p.ksplit);
    hipLaunchKernelGGL((gemm_kernel<WM, WN, TM, TN, AKC, BKC, ACT, PRO, false, X3>), grid, dim3(256), 0, st, p);
}

// kind: 1 = 32 x 128, 2 = 64 x 64, 0 = 128 x 128.  x3 (bf16x3 arithmetic on fp32 operands, split while staging) exists
// for the two larger tiles only: the 32-row tile serves launches that are weight-streaming bound.
template <bool AKC, bool BKC, int ACT, bool PRO>
void launch_tile(const GemmParams& p, int kind, bool x3, hipStream_t st) {
    if (kind == 1)      launch<1, 4, 1, 1, AKC, BKC, ACT, PRO>(p, st);
    else if (kind == 2) { if (x3) launch<2, 2, 1, 1, AKC, BKC, ACT, PRO, true>(p, st); else launch<2, 2, 1, 1, AKC, BKC, ACT, PRO>(p, st); }
    else                { if (x3) launch<2, 2, 2, 2, AKC, BKC, ACT, PRO, true>(p, st); else launch<2, 2, 2, 2, AKC, BKC, ACT, PRO>(p, st); }
}

template <int ACT, bool PRO>
void launch_tn(const GemmParams& p, int kind, bool x3, hipStream_t st) {
    if (kind == 2) { if (x3) launch<2, 2, 1, 1, false, false, ACT, PRO, true>(p, st); else launch<2, 2, 1, 1, false, false, ACT, PRO>(p, st); }
    else           { if (x3) launch<2, 2, 2, 2, false, false, ACT, PRO, true>(p, st); else launch<2, 2, 2, 2, false, false, ACT, PRO>(p, st); }
}

// wgrad of a per-vertex Linear (a [512, 512]-class output over a few thousand rows): 128 x 128 tiles leave 16 of them,
// each cut into 16 K ranges of 4 slices; 64 x 64 tiles give 64, cut 8 ways: 22 -> 16 us at [512, 512] x 2048 rows,
// 38 -> 29 us at [1536, 512] (scripts/bench_gemm_mid.py wgrad).  Long reductions over the larger outputs keep the
// big tile (77 vs 81 us at [1536, 512] x 8192).  WF3D_TN_MID=0 switches it off.
inline bool tn_mid(int M, int N, int K) {
    static const int on = [] { const char* e = getenv("WF3D_TN_MID"); return e ? atoi(e) : 1; }();
    const long t128 = (long)wf3d_cdiv(M, 128) * wf3d_cdiv(N, 128);
    return on && M > 64 && N > 64 && t128 < 64 && (K <= 4096 || t128 <= 16);
}

}  // namespace

extern "C" size_t wf3d_gemm_ws_bytes(int M, int N, int K, int layout) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const bool small = layout != WF3D_TN && small_m(M);
    const bool mid = layout != WF3D_TN ? (!small && mid_tile(M, N, K)) : tn_mid(M, N, K);
    SplitPlan s = plan_split(M, N, K, small ? 32 : (mid ? 64 : 128), mid ? 64 : 128);
    return s.ksplit > 1 ? (size_t)s.ksplit * M * N * sizeof(float) : 0;
}

extern "C" int wf3d_gemm(const wf3d_gemm_t* d, void* stream) {
    WF3D_CHECK(d != nullptr, WF3D_ERR_ARG, "wf3d_gemm: null descriptor");
    WF3D_CHECK(d->M >= 0 && d->N >= 0 && d->K >= 0, WF3D_ERR_ARG, "wf3d_gemm: negative dims");
    if (d->M == 0 || d->N == 0) return WF3D_OK;
    WF3D_CHECK(d->A && d->B && d->C, WF3D_ERR_ARG, "wf3d_gemm: null operand");
    WF3D_CHECK(d->layout >= WF3D_NT && d->layout <= WF3D_TN, WF3D_ERR_ARG, "wf3d_gemm: bad layout %d", d->layout);
    WF3D_CHECK(d->K > 0, WF3D_ERR_ARG, "wf3d_gemm: K must be > 0");
    const bool a_kc = d->layout != WF3D_TN, b_kc = d->layout == WF3D_NT;
    WF3D_CHECK(d->lda >= (a_kc ? d->K : d->M), WF3D_ERR_ARG, "wf3d_gemm: lda %d too small", d->lda);
    WF3D_CHECK(d->ldb >= (b_kc ? d->K : d->N), WF3D_ERR_ARG, "wf3d_gemm: ldb %d too small", d->ldb);
    WF3D_CHECK(d->ldc >= d->N, WF3D_ERR_ARG, "wf3d_gemm: ldc %d < N %d", d->ldc, d->N);
    WF3D_CHECK(!d->addend || d->ld_addend >= d->N, WF3D_ERR_ARG, "wf3d_gemm: ld_addend too small");
    WF3D_CHECK(!(d->pro_enable && d->layout == WF3D_NN), WF3D_ERR_UNSUPPORTED, "wf3d_gemm: no prologue for NN layout");
    WF3D_CHECK(!d->pro_enable || (d->pro_act >= 0 && d->pro_act <= 2), WF3D_ERR_ARG, "wf3d_gemm: bad pro_act");
    WF3D_CHECK(!d->pro_enable || !d->pro_mu || d->pro_rs, WF3D_ERR_ARG, "wf3d_gemm: pro_mu without pro_rs");
    WF3D_CHECK(!d->pro_enable || !d->pro_gamma || d->pro_beta, WF3D_ERR_ARG, "wf3d_gemm: pro_gamma without pro_beta");
    WF3D_CHECK(d->drop_p >= 0.f && d->drop_p < 1.f, WF3D_ERR_ARG, "wf3d_gemm: drop_p out of range");

    GemmParams p{};
    p.A = d->A; p.B = d->B; p.C = d->C; p.bias = d->bias; p.addend = d->addend;
    p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
    p.ld_addend = d->ld_addend;
    WF3D_CHECK(d->lr_k >= 0 && d->lr_k <= 4, WF3D_ERR_ARG, "wf3d_gemm: lr_k must be 0..4");
    WF3D_CHECK(d->lr_k == 0 || (d->lr_u && d->lr_v && d->ld_lr_u >= d->lr_k && d->ld_lr_v >= d->lr_k), WF3D_ERR_ARG,
               "wf3d_gemm: low-rank term needs U, V and leading dimensions >= lr_k");
    p.lr_u = d->lr_u; p.lr_v = d->lr_v; p.lr_k = d->lr_k; p.ld_lr_u = d->ld_lr_u; p.ld_lr_v = d->ld_lr_v;
    p.pmu = d->pro_mu; p.prs = d->pro_rs; p.pgam = d->pro_gamma; p.pbet = d->pro_beta;
    p.has_ln = d->pro_enable && d->pro_mu != nullptr;
    p.has_affine = d->pro_enable && d->pro_gamma != nullptr;
    if (d->pro_enable && d->drop_p > 0.f) {
        p.drop_seed = d->drop_seed;
        p.drop_thresh = (uint32_t)((double)d->drop_p * 4294967296.0);
        p.drop_scale = 1.0f / (1.0f - d->drop_p);
    }
    p.accumulate = d->accumulate;
    p.vecA = ((uintptr_t)d->A % 16 == 0) && (d->lda % 4 == 0);
    p.vecB = ((uintptr_t)d->B % 16 == 0) && (d->ldb % 4 == 0);
    p.vecP = !p.has_affine || (((uintptr_t)d->pro_gamma % 16 == 0) && ((uintptr_t)d->pro_beta % 16 == 0));

    const bool small = d->layout != WF3D_TN && small_m(d->M);
    const bool mid = d->layout != WF3D_TN ? (!small && mid_tile(d->M, d->N, d->K)) : tn_mid(d->M, d->N, d->K);
    const int kind = small ? 1 : (mid ? 2 : 0);
    const int bm = small ? 32 : (mid ? 64 : 128), bn = mid ? 64 : 128;
    p.nbm = wf3d_cdiv(d->M, bm);
    p.nbn = wf3d_cdiv(d->N, bn);
    SplitPlan s = plan_split(d->M, d->N, d->K, bm, bn);
    const size_t need = s.ksplit > 1 ? (size_t)s.ksplit * d->M * d->N * sizeof(float) : 0;
    if (need && (d->ws == nullptr || d->ws_bytes < need)) { s.ksplit = 1; s.kt_per = wf3d_cdiv(d->K, BK); }
    p.ksplit = s.ksplit; p.kt_per_split = s.kt_per;
    p.slab = s.ksplit > 1 ? (float*)d->ws : nullptr;

    hipStream_t st = (hipStream_t)stream;
    const int act = d->pro_enable ? d->pro_act : 0;
    const bool pro = d->pro_enable != 0;
    const bool x3 = d->x3 != 0;
    if (d->layout == WF3D_NT) {
        if (!pro)                         launch_tile<true, true, 0, false>(p, kind, x3, st);
        else if (act == WF3D_ACT_RELU)    launch_tile<true, true, WF3D_ACT_RELU, true>(p, kind, x3, st);
        else if (act == WF3D_ACT_GELU)    launch_tile<true, true, WF3D_ACT_GELU, true>(p, kind, x3, st);
        else                              launch_tile<true, true, WF3D_ACT_NONE, true>(p, kind, x3, st);
    } else if (d->layout == WF3D_NN) {
        launch_tile<true, false, 0, false>(p, kind, x3, st);
    } else {
        if (!pro)                         launch_tn<0, false>(p, kind, x3, st);
        else if (act == WF3D_ACT_RELU)    launch_tn<WF3D_ACT_RELU, true>(p, kind, x3, st);
        else if (act == WF3D_ACT_GELU)    launch_tn<WF3D_ACT_GELU, true>(p, kind, x3, st);
        else                              launch_tn<WF3D_ACT_NONE, true>(p, kind, x3, st);
    }
    WF3D_LAUNCH_CHECK();
    if (p.ksplit > 1) {
        const size_t total = (size_t)p.M * p.N;
        if (p.ksplit > 16 && p.N % 4 == 0 && ((uintptr_t)p.slab % 16 == 0)) {
            hipLaunchKernelGGL(gemm_splitk_reduce_wide, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, p);
        } else {
            int blocks = (int)((total + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(gemm_splitk_reduce, dim3(blocks), dim3(256), 0, st, p);
        }
        WF3D_LAUNCH_CHECK();
    }
    return WF3D_OK;
}

// ---------------------------------------------------------------------------
// bf16x3 split-precision GEMM (NT form): C[M,N] = A·B^T (+bias) with A, B in the
// "sx8" split format produced by wf3d_split_rows / wf3d_ln_prep: logical [R, K]
// fp32 values stored as [R][K/8][2][8] bf16 = 8 high parts then 8 low parts per
// 8 consecutive k (same bytes and row pitch as the fp32 tensor).
// ---------------------------------------------------------------------------
extern "C" size_t wf3d_gemm_split_ws_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    SplitPlan s = plan_split(M, N, K, 128, 128);
    const size_t a = s.ksplit > 1 ? (size_t)s.ksplit * M * N * sizeof(float) : 0;
    const size_t b = wf3d_gemm_split_dma_ws_bytes(M, N, K);
    return a > b ? a : b;
}

extern "C" int wf3d_gemm_split(const void* A_sx8, const void* B_sx8, float* C, const float* bias, int M, int N, int K,
                               int lda, int ldb, int ldc, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    WF3D_CHECK(M >= 0 && N >= 0 && K > 0, WF3D_ERR_ARG, "wf3d_gemm_split: bad dims");
    if (M == 0 || N == 0) return WF3D_OK;
    WF3D_CHECK(A_sx8 && B_sx8 && C, WF3D_ERR_ARG, "wf3d_gemm_split: null operand");
    WF3D_CHECK(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, WF3D_ERR_UNSUPPORTED,
               "wf3d_gemm_split: K, lda, ldb must be multiples of 8 (sx8 groups)");
    WF3D_CHECK(lda >= K && ldb >= K && ldc >= N, WF3D_ERR_ARG, "wf3d_gemm_split: leading dimension too small");
    WF3D_CHECK(((uintptr_t)A_sx8 % 16 == 0) && ((uintptr_t)B_sx8 % 16 == 0), WF3D_ERR_ARG, "wf3d_gemm_split: operands must be 16-byte aligned");
    {
        // LDS-DMA staged kernel for every shape it supports (WF3D_SPLIT_DMA=0 forces the register-staged one)
        static const int use_dma = [] { const char* e = getenv("WF3D_SPLIT_DMA"); return e ? atoi(e) : 1; }();
        if (use_dma && wf3d_gemm_split_dma_ok(M, N, K, lda, ldb))
            return wf3d_gemm_split_dma(A_sx8, B_sx8, C, bias, M, N, K, lda, ldb, ldc, accumulate, ws, ws_bytes, stream);
    }
    GemmParams p{};
    p.A = (const float*)A_sx8; p.B = (const float*)B_sx8; p.C = C; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.accumulate = accumulate;
    p.vecA = p.vecB = p.vecP = 1;
    p.nbm = wf3d_cdiv(M, 128); p.nbn = wf3d_cdiv(N, 128);
    SplitPlan sp = plan_split(M, N, K, 128, 128);
    const size_t need = sp.ksplit > 1 ? (size_t)sp.ksplit * M * N * sizeof(float) : 0;
    if (need && (ws == nullptr || ws_bytes < need)) { sp.ksplit = 1; sp.kt_per = wf3d_cdiv(K, BK); }
    p.ksplit = sp.ksplit; p.kt_per_split = sp.kt_per;
    p.slab = sp.ksplit > 1 ? (float*)ws : nullptr;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(p.nbm * p.nbn, 1, p.ksplit);
    hipLaunchKernelGGL((gemm_kernel<2, 2, 2, 2, true, true, 0, false, true>), grid, dim3(256), 0, st, p);
    WF3D_LAUNCH_CHECK();
    if (p.ksplit > 1) {
        const size_t total = (size_t)M * N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(gemm_splitk_reduce, dim3(blocks), dim3(256), 0, st, p);
        WF3D_LAUNCH_CHECK();
    }
    return WF3D_OK;
}
