// MPEG-7 region shape (angular radial transform) per segment.
// Replaces MPEG7Descriptors.mpeg7_region_shape / art_transform / cart2radial
// (scripts/mpeg7Descriptors.py:144-186, 233-275).
//
// What the reference computes for segment s (quirks kept):
//   * the image is the WHOLE frame, not the bounding-box crop (the crop at :271-272 is never
//     used): value 1 on the segment, 0 on every other segment, and the grey level on pixels
//     outside every segment (label -1), because segment_generator only zeroes other segments;
//   * cart2radial (:156-185): polar[y_a][x_a] = floor(mean of image[y0:y1, x0:x1]) with
//     x0 = max(int(px), 0), x1 = min(int(px) + 2, W) and PYTHON SLICE semantics, so a negative
//     x1 counts from the end of the row (a sample point left of the frame averages almost the
//     whole row), an empty slice leaves 0;
//   * ART (:239-254): F[n][m] = | sum_y e^{i m th_y} sum_x polar[y][x] * x * R_n(x/W) | * c,
//     R_0 = 1, R_n = 2 cos(pi n rho), normalised by F[0][0].
//
// Device work: a summed-area table of the segment's value image (row scan, column scan) makes
// every window mean O(1) whatever its size; one workgroup per polar row folds the row into the
// three radial sums; one wave per segment finishes the 36 angular sums.  The sample positions
// use cos / sin tables computed on the host (numpy), so int(px) is decided on the same bits as
// in the reference.  Integer parts are exact; the float64 sums differ from numpy's only in
// summation order.
#include "common.h"
#include "../../include/imgan.h"
#include <math.h>

namespace {

// value image of segment s: 1 on the segment, grey outside every segment, else 0
__device__ __forceinline__ int seg_value(int label, int s, int grey) {
    return label == s ? 1 : (label < 0 ? grey : 0);
}

// sat: [nchunk][H+1][W+1] int32, row 0 / column 0 are zero.  One workgroup per (segment, row).
__global__ __launch_bounds__(256) void shape_rowscan_kernel(const uint8_t* __restrict__ grey,
                                                            const int32_t* __restrict__ labels,
                                                            int H, int W, int seg0,
                                                            int32_t* __restrict__ sat) {
    __shared__ int part[256];
    __shared__ int carry;
    const int y = blockIdx.x, c = blockIdx.y, s = seg0 + c;
    int32_t* row = sat + ((size_t)c * (H + 1) + (y + 1)) * (W + 1);
    if (threadIdx.x == 0) {
        carry = 0;
        row[0] = 0;
    }
    if (y == 0)                                          // the zero row on top
        for (int x = threadIdx.x; x <= W; x += 256) sat[(size_t)c * (H + 1) * (W + 1) + x] = 0;
    __syncthreads();
    for (int base = 0; base < W; base += 256) {
        const int x = base + threadIdx.x;
        int v = 0;
        if (x < W) v = seg_value(labels[(size_t)y * W + x], s, grey[(size_t)y * W + x]);
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (x < W) row[x + 1] = carry + part[threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 255) carry += part[255];
        __syncthreads();
    }
}

// one thread per (segment, column): running sum down the rows (coalesced across columns)
__global__ __launch_bounds__(256) void shape_colscan_kernel(int H, int W, int nchunk,
                                                            int32_t* __restrict__ sat) {
    const int x = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (x > W || c >= nchunk) return;
    int32_t* col = sat + (size_t)c * (H + 1) * (W + 1) + x;
    int acc = 0;
    for (int y = 1; y <= H; ++y) {
        acc += col[(size_t)y * (W + 1)];
        col[(size_t)y * (W + 1)] = acc;
    }
}

// image[a0:a1] with a0 = max(int(p), 0), a1 = min(int(p) + 2, size) under Python slice rules
__device__ __forceinline__ void py_slice(double p, int size, int* lo, int* hi) {
    // int() truncates toward zero; the sample points stay within a few image diagonals
    const int t = (int)p;
    int a0 = t >= 0 ? t : 0;
    int a1 = t + 2 <= size ? t + 2 : size;
    if (a1 < 0) a1 = a1 + size > 0 ? a1 + size : 0;      // negative stop counts from the end
    if (a0 > size) a0 = size;
    *lo = a0;
    *hi = a1 > a0 ? a1 : a0;
}

// One workgroup per (polar row y_a, segment): polar values of the row (optional output) and the
// radial sums G_n[y_a] = sum_x polar * x * R_n(x / W), n = 0, 1, 2.
__global__ __launch_bounds__(256) void shape_polar_kernel(const int32_t* __restrict__ sat, int H,
                                                          int W, int seg0,
                                                          const double* __restrict__ cos_tab,
                                                          const double* __restrict__ sin_tab,
                                                          double r_scale, uint8_t* __restrict__ polar,
                                                          double* __restrict__ radial) {
    __shared__ double red[3][256];
    const int ya = blockIdx.x, c = blockIdx.y;
    const int32_t* I = sat + (size_t)c * (H + 1) * (W + 1);
    const double ca = cos_tab[ya], sa = sin_tab[ya];
    const double half_w = W / 2.0, half_h = H / 2.0, inv_w = 1.0 / (double)W;
    double g0 = 0.0, g1 = 0.0, g2 = 0.0;
    for (int xa = threadIdx.x; xa < W; xa += 256) {
        const double radius = xa * r_scale;                       // :159
        const double px = radius * ca + half_w;                   // :161
        const double py = radius * sa + half_h;                   // :162
        int x0, x1, y0, y1;
        py_slice(px, W, &x0, &x1);
        py_slice(py, H, &y0, &y1);
        const int cnt = (x1 - x0) * (y1 - y0);
        int val = 0;
        if (cnt > 0) {
            const int sum = I[(size_t)y1 * (W + 1) + x1] - I[(size_t)y0 * (W + 1) + x1] -
                            I[(size_t)y1 * (W + 1) + x0] + I[(size_t)y0 * (W + 1) + x0];
            val = sum / cnt;                                       // :179, integer division
        }
        if (polar) polar[((size_t)(seg0 + c) * H + ya) * W + xa] = (uint8_t)val;
        if (val) {
            const double rho = inv_w * xa;                         // :241
            const double t = (double)val * (double)xa;
            g0 += t;
            g1 += t * (2.0 * cos(M_PI * 1.0 * rho));               // :248
            g2 += t * (2.0 * cos(M_PI * 2.0 * rho));
        }
    }
    red[0][threadIdx.x] = g0;
    red[1][threadIdx.x] = g1;
    red[2][threadIdx.x] = g2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off)
            for (int n = 0; n < 3; ++n) red[n][threadIdx.x] += red[n][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x < 3) radial[((size_t)(seg0 + c) * H + ya) * 3 + threadIdx.x] = red[threadIdx.x][0];
}

// One workgroup (64 lanes, 36 used) per segment: F[n][m] = |sum_y e^{i m th_y} G_n[y]| * scale,
// then the normalisation by F[0][0] (:255-256).
__global__ __launch_bounds__(64) void shape_art_kernel(const double* __restrict__ radial, int H,
                                                       int W, double* __restrict__ art) {
    __shared__ double f[36];
    const int s = blockIdx.x, t = threadIdx.x;
    const double ang_scale = (2.0 * M_PI) / (double)H, r_scale = 1.0 / (double)W;   // :236-237
    if (t < 36) {
        const int n = t / 12, m = t % 12;
        double re = 0.0, im = 0.0;
        const double* G = radial + (size_t)s * H * 3 + n;
        for (int y = 0; y < H; ++y) {
            double sn, cs;
            sincos((double)m * (ang_scale * (double)y), &sn, &cs);
            const double g = G[(size_t)y * 3];
            re += cs * g;
            im += sn * g;
        }
        // a_m carries 1/(2 pi) (:244); r_scale once per pixel (:251) and once per row sum (:252)
        f[t] = sqrt(re * re + im * im) * (1.0 / (2.0 * M_PI)) * r_scale * r_scale * ang_scale;
    }
    __syncthreads();
    if (t < 36) art[(size_t)s * 36 + t] = f[0] != 0.0 ? f[t] / f[0] : f[t];
}

}  // namespace

extern "C" size_t imgan_region_shape_scratch_bytes(int H, int W, int nseg) {
    if (H < 1 || W < 1 || nseg < 1) return 0;
    // one summed-area table per segment of a chunk (at most 64 segments at a time) + radial sums
    const size_t per_seg = (size_t)(H + 1) * (W + 1) * 4;
    const size_t chunk = nseg < 64 ? nseg : 64;
    return imgan::align_up(chunk * per_seg, 256) + imgan::align_up((size_t)nseg * H * 3 * 8, 256);
}

extern "C" int imgan_region_shape(const uint8_t* grey, const int32_t* labels, int H, int W,
                                  int nseg, const double* cos_tab, const double* sin_tab,
                                  double r_scale, uint8_t* polar_out, double* art_out,
                                  void* scratch, imgan_stream_t stream) {
    IMGAN_REQUIRE(grey && labels && cos_tab && sin_tab && art_out && scratch,
                  "imgan_region_shape: null pointer");
    IMGAN_REQUIRE(H >= 1 && W >= 1 && nseg >= 1 && H <= 65535, "imgan_region_shape: bad shape");
    IMGAN_REQUIRE((long long)H * W * 255 < 0x7fffffffLL,
                  "imgan_region_shape: image too large for the 32-bit summed-area table");
    IMGAN_REQUIRE(isfinite(r_scale) && r_scale > 0.0, "imgan_region_shape: bad radial scale");
    hipStream_t st = (hipStream_t)stream;
    const size_t per_seg = (size_t)(H + 1) * (W + 1) * 4;
    const int chunk = nseg < 64 ? nseg : 64;
    int32_t* sat = (int32_t*)scratch;
    double* radial = (double*)((char*)scratch + imgan::align_up((size_t)chunk * per_seg, 256));
    for (int s0 = 0; s0 < nseg; s0 += chunk) {
        const int nc = nseg - s0 < chunk ? nseg - s0 : chunk;
        hipLaunchKernelGGL(shape_rowscan_kernel, dim3(H, nc), dim3(256), 0, st, grey, labels, H, W,
                           s0, sat);
        hipLaunchKernelGGL(shape_colscan_kernel, dim3((unsigned)imgan::cdiv(W + 1, 256), nc),
                           dim3(256), 0, st, H, W, nc, sat);
        hipLaunchKernelGGL(shape_polar_kernel, dim3(H, nc), dim3(256), 0, st, sat, H, W, s0, cos_tab,
                           sin_tab, r_scale, polar_out, radial);
    }
    hipLaunchKernelGGL(shape_art_kernel, dim3(nseg), dim3(64), 0, st, radial, H, W, art_out);
    return imgan::check_launch("imgan_region_shape");
}
