// Row f-4: what follows the hot path at evaluation time — evaluate.py:74-110 and the Hausdorff line
// distance of eval/ap_calculator.py:8-36.
//
//   edge_endpoints_kernel   every candidate edge (i < j) of every sample in one launch: keep = prob > threshold (:79),
//                           end points gathered from the predicted vertices with the HIGHER-z vertex first (:88-89:
//                           flip of an ascending argsort over z; on equal z the second vertex comes first) -> the
//                           [B, max_e, 2, 3] / [B, max_e] arrays the host compacts after ONE device->host copy (the
//                           reference copies and loops per sample).
//   hausdorff_lines_kernel  [N, M] symmetric Hausdorff distance between N predicted and M labelled segments, each sampled
//                           at S <= 32 points: one thread per pair, the S x S point distances stay in registers (the
//                           reference materialises a (20 N) x (20 M) cdist matrix: 310 MB at N = 2016, M = 48).  float64,
//                           sample points a + w_k * (b - a) with numpy's linspace weights.
#include "wf3d_common.h"

namespace {

__device__ __forceinline__ int eo(int i, int nv) { return i * nv - (i * (i + 1)) / 2; }     // edges before row i

__global__ __launch_bounds__(256) void edge_endpoints_kernel(const float* __restrict__ verts, long sstride, long vstride,
                                                              const int32_t* __restrict__ counts, const float* __restrict__ probs,
                                                              int max_e, int V, float thr, float* __restrict__ ev,
                                                              unsigned char* __restrict__ keep) {
    const int s = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int nv = counts[s];
    if (idx >= V * V) return;
    const int i = idx / V, j = idx - i * V;
    if (!(i < j && j < nv)) return;
    const int e = eo(i, nv) + (j - i - 1);
    if (e >= max_e) return;
    const float* vi = verts + s * sstride + i * vstride;
    const float* vj = verts + s * sstride + j * vstride;
    const bool i_first = vi[2] > vj[2];
    const float* a = i_first ? vi : vj;
    const float* b = i_first ? vj : vi;
    float* o = ev + ((size_t)s * max_e + e) * 6;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = b[0]; o[4] = b[1]; o[5] = b[2];
    keep[(size_t)s * max_e + e] = probs[(size_t)s * max_e + e] > thr ? 1 : 0;
}

template <int SMAX>
__global__ __launch_bounds__(256) void hausdorff_lines_kernel(const double* __restrict__ pa, const double* __restrict__ pd,
                                                               const double* __restrict__ ta, const double* __restrict__ td,
                                                               int N, int M, int S, double* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * M) return;
    const int n = (int)(idx / M), m = (int)(idx - (long)n * M);
    const double step = S > 1 ? 1.0 / (double)(S - 1) : 0.0;
    const double ax = pa[n * 3], ay = pa[n * 3 + 1], az = pa[n * 3 + 2], dx = pd[n * 3], dy = pd[n * 3 + 1], dz = pd[n * 3 + 2];
    const double bx = ta[m * 3], by = ta[m * 3 + 1], bz = ta[m * 3 + 2], ex = td[m * 3], ey = td[m * 3 + 1], ez = td[m * 3 + 2];
    double tmin[SMAX];
#pragma unroll
    for (int j = 0; j < SMAX; ++j) tmin[j] = INFINITY;
    double h_pt = 0.0;
    for (int i = 0; i < S; ++i) {
        const double wi = i == S - 1 && S > 1 ? 1.0 : (double)i * step;
        const double px = ax + wi * dx, py = ay + wi * dy, pz = az + wi * dz;
        double rmin = INFINITY;
#pragma unroll
        for (int j = 0; j < SMAX; ++j) {
            if (j < S) {
                const double wj = j == S - 1 && S > 1 ? 1.0 : (double)j * step;
                const double qx = bx + wj * ex - px, qy = by + wj * ey - py, qz = bz + wj * ez - pz;
                const double d = sqrt(qx * qx + qy * qy + qz * qz);
                rmin = fmin(rmin, d);
                tmin[j] = fmin(tmin[j], d);
            }
        }
        h_pt = fmax(h_pt, rmin);
    }
    double h_tp = 0.0;
#pragma unroll
    for (int j = 0; j < SMAX; ++j)
        if (j < S) h_tp = fmax(h_tp, tmin[j]);
    out[idx] = fmax(h_pt, h_tp);
}

}  // namespace

extern "C" int wf3d_edge_endpoints(const float* verts, long sample_stride, long vertex_stride, const int32_t* counts,
                                   const float* probs, int B, int V, int max_e, float threshold, float* edge_vertices,
                                   unsigned char* keep, void* stream) {
    WF3D_CHECK(B >= 0 && V >= 0 && max_e >= 0, WF3D_ERR_ARG, "wf3d_edge_endpoints: bad dims");
    if (B == 0 || V < 2 || max_e == 0) return WF3D_OK;
    WF3D_CHECK(verts && counts && probs && edge_vertices && keep, WF3D_ERR_ARG, "wf3d_edge_endpoints: null pointer");
    WF3D_CHECK(B <= 65535, WF3D_ERR_UNSUPPORTED, "wf3d_edge_endpoints: B > 65535");
    hipLaunchKernelGGL(edge_endpoints_kernel, dim3(wf3d_cdiv((long)V * V, 256), B), dim3(256), 0, (hipStream_t)stream, verts,
                       sample_stride, vertex_stride, counts, probs, max_e, V, threshold, edge_vertices, keep);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_hausdorff_lines(const double* p_start, const double* p_diff, const double* t_start, const double* t_diff,
                                    int N, int M, int sample_points, double* out, void* stream) {
    WF3D_CHECK(N >= 0 && M >= 0 && sample_points >= 1 && sample_points <= 32, WF3D_ERR_UNSUPPORTED,
               "wf3d_hausdorff_lines: 1..32 sample points per segment (got %d)", sample_points);
    if (N == 0 || M == 0) return WF3D_OK;
    WF3D_CHECK(p_start && p_diff && t_start && t_diff && out, WF3D_ERR_ARG, "wf3d_hausdorff_lines: null pointer");
    hipLaunchKernelGGL(hausdorff_lines_kernel<32>, dim3(wf3d_cdiv((long)N * M, 256)), dim3(256), 0, (hipStream_t)stream, p_start,
                       p_diff, t_start, t_diff, N, M, sample_points, out);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}
