// Row f-3: the input pipeline in front of the hot path, datasets/building3d.py:95-190 of the reference —
// per sample: colour / 256, centroid + max-norm normalisation of xyz, random sampling of num_points rows, flip / z-rotation
// augmentation — as two kernels on clouds that stay resident in HBM.
//
//   cloud_normalize_kernel   once per cloud (when it is cached): raw float64 rows [n, C] (UTM coordinates ~6.5e6 m: fp32 cannot
//                            hold them) -> float64 normalised rows, centroid[3], max_distance; fp64 arithmetic in the
//                            reference's order (:118-121), one workgroup per cloud.
//   cloud_sample_kernel      per batch: out[b, p, :] = fp32(augment(norm[first[cloud[b]] + choice[b, p], :])) — the
//                            reference's gather (:127-128), flips and rotation about z (:130-145) on float64 values, rounded
//                            to fp32 once at the end like its final astype(np.float32) (:158).  The random choices and
//                            augmentation parameters come from the host, drawn from numpy in the reference's order, so a
//                            seeded run reproduces the reference's batches.
// Plus the host-side text parser for .xyz files (np.loadtxt replacement): wf3d_parse_floats.
#include <locale.h>
#include <stdio.h>
#include <stdlib.h>

#include "wf3d_common.h"

namespace {

__device__ __forceinline__ double block_reduce_d(double v, double* s, bool take_max) {
    const int tid = threadIdx.x;
    s[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s[tid] = take_max ? fmax(s[tid], s[tid + o]) : s[tid] + s[tid + o];
        __syncthreads();
    }
    const double r = s[0];
    __syncthreads();
    return r;
}

// one workgroup per cloud; `first` = row offsets of the packed clouds
__global__ __launch_bounds__(256) void cloud_normalize_kernel(const double* __restrict__ raw, const long* __restrict__ first,
                                                               int C, int color_lo, int color_hi, int normalize,
                                                               double* __restrict__ out, double* __restrict__ centroid,
                                                               double* __restrict__ max_distance) {
    __shared__ double s[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const long r0 = first[b], n = first[b + 1] - r0;
    const double* in = raw + r0 * C;
    double* o = out + r0 * C;
    double cx = 0, cy = 0, cz = 0, md = 1.0;
    if (normalize) {
        double sx = 0, sy = 0, sz = 0;
        for (long i = tid; i < n; i += 256) { sx += in[i * C]; sy += in[i * C + 1]; sz += in[i * C + 2]; }
        cx = block_reduce_d(sx, s, false) / (double)n;
        cy = block_reduce_d(sy, s, false) / (double)n;
        cz = block_reduce_d(sz, s, false) / (double)n;
        double m = 0;
        for (long i = tid; i < n; i += 256) {
            const double x = in[i * C] - cx, y = in[i * C + 1] - cy, z = in[i * C + 2] - cz;
            m = fmax(m, sqrt(x * x + y * y + z * z));
        }
        md = block_reduce_d(m, s, true);
        if (tid == 0) { centroid[b * 3] = cx; centroid[b * 3 + 1] = cy; centroid[b * 3 + 2] = cz; max_distance[b] = md; }
    }
    for (long i = tid; i < n; i += 256) {
        for (int k = 0; k < C; ++k) {
            double v = in[i * C + k];
            if (k >= color_lo && k < color_hi) v = v / 256.0;                       // building3d.py:105,110
            if (normalize && k < 3) v = (v - (k == 0 ? cx : (k == 1 ? cy : cz))) / md;
            o[i * C + k] = v;
        }
    }
}

__global__ __launch_bounds__(256) void cloud_sample_kernel(const double* __restrict__ norm, const long* __restrict__ first,
                                                            const int* __restrict__ cloud, const int* __restrict__ choice,
                                                            const double* __restrict__ aug, int P, int C,
                                                            float* __restrict__ out) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const double* row = norm + (first[cloud[b]] + choice[(long)b * P + p]) * C;
    const double fx = aug[b * 4], fy = aug[b * 4 + 1], c = aug[b * 4 + 2], s = aug[b * 4 + 3];
    const double x = fx * row[0], y = fy * row[1];
    float* o = out + ((long)b * P + p) * C;
    o[0] = (float)(x * c - y * s);               // [x y z] . rotz(t)^T  (building3d.py:144)
    o[1] = (float)(x * s + y * c);
    for (int k = 2; k < C; ++k) o[k] = (float)row[k];
}

}  // namespace

extern "C" int wf3d_cloud_normalize(const double* raw, const long* first, int nclouds, int C, int color_lo, int color_hi,
                                    int normalize, double* out, double* centroid, double* max_distance, void* stream) {
    WF3D_CHECK(nclouds >= 0 && C >= 3 && C <= 16, WF3D_ERR_ARG, "wf3d_cloud_normalize: bad dims");
    if (nclouds == 0) return WF3D_OK;
    WF3D_CHECK(raw && first && out && (!normalize || (centroid && max_distance)), WF3D_ERR_ARG, "wf3d_cloud_normalize: null pointer");
    hipLaunchKernelGGL(cloud_normalize_kernel, dim3(nclouds), dim3(256), 0, (hipStream_t)stream, raw, first, C, color_lo, color_hi,
                       normalize, out, centroid, max_distance);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

extern "C" int wf3d_cloud_sample(const double* norm, const long* first, const int* cloud, const int* choice, const double* aug,
                                 int B, int P, int C, float* out, void* stream) {
    WF3D_CHECK(B >= 0 && P >= 0 && C >= 3 && C <= 16, WF3D_ERR_ARG, "wf3d_cloud_sample: bad dims");
    if (B == 0 || P == 0) return WF3D_OK;
    WF3D_CHECK(norm && first && cloud && choice && aug && out, WF3D_ERR_ARG, "wf3d_cloud_sample: null pointer");
    WF3D_CHECK(B <= 65535, WF3D_ERR_UNSUPPORTED, "wf3d_cloud_sample: B > 65535");
    hipLaunchKernelGGL(cloud_sample_kernel, dim3(wf3d_cdiv(P, 256), B), dim3(256), 0, (hipStream_t)stream, norm, first, cloud, choice,
                       aug, P, C, out);
    WF3D_LAUNCH_CHECK();
    return WF3D_OK;
}

// Host: the numbers of a whitespace-separated text table.  `#` starts a comment (to the end of the line); numbers are
// parsed in the C locale whatever the process locale is.  out[0..max_vals) receives the values in file order; returns
// how many the file holds (> max_vals: call again with a larger buffer), -1 if the file cannot be read, -2 on a token
// that is not a number, -3 (wf3d_parse_table only) if two non-empty rows hold different numbers of values.
static long parse_numbers(const char* path, double* out, long max_vals, long* ncols) {
    static locale_t c_loc = newlocale(LC_ALL_MASK, "C", (locale_t)0);
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return -1; }
    const long sz = ftell(f);
    if (sz < 0 || fseek(f, 0, SEEK_SET) != 0) { fclose(f); return -1; }
    char* buf = (char*)malloc((size_t)sz + 1);
    if (!buf) { fclose(f); return -1; }
    const size_t got = fread(buf, 1, (size_t)sz, f);
    fclose(f);
    if (got != (size_t)sz) { free(buf); return -1; }
    buf[got] = 0;
    long n = 0, in_row = 0, cols = 0;
    bool ragged = false;
    char* p = buf;
    auto end_row = [&]() {
        if (in_row) {
            if (cols == 0) cols = in_row; else if (in_row != cols) ragged = true;
            in_row = 0;
        }
    };
    for (;;) {
        while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
        if (*p == '#') { while (*p && *p != '\n') ++p; }
        if (*p == '\n') { end_row(); ++p; continue; }
        if (!*p) break;
        char* end;
        const double v = c_loc ? strtod_l(p, &end, c_loc) : strtod(p, &end);
        if (end == p) { free(buf); return -2; }
        if (n < max_vals && out) out[n] = v;
        ++n; ++in_row;
        p = end;
    }
    end_row();
    free(buf);
    if (ncols) { *ncols = cols; if (ragged) return -3; }
    return n;
}

extern "C" long wf3d_parse_floats(const char* path, double* out, long max_vals) { return parse_numbers(path, out, max_vals, nullptr); }

// Same, and *ncols = the number of values per row (np.loadtxt's second dimension).
extern "C" long wf3d_parse_table(const char* path, double* out, long max_vals, long* ncols) {
    if (!ncols) return -1;
    return parse_numbers(path, out, max_vals, ncols);
}
