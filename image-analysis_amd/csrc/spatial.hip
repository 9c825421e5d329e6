// Spatial relations between segments: label histograms of the eight compass sectors around a
// segment's enlarged bounding box.  Replaces the pixel loops of
// SpatialExtractor.find_spatial / ratio_indexing (scripts/spatialextract.py:25-38, 71-171).
//
// The reference builds, per segment and sector, a Python list of (y, x) points column by
// column: for column x the rows [lo(x), hi(x)) where each bound is a constant, int(a*x + b), or
// that value clamped against a constant (e.g. :75-79).  The host layer passes these column
// rules ("pieces": a sector is one or two column ranges) with the float64 line coefficients
// computed exactly as the reference does; the kernel evaluates int(a*x + b) in float64 with
// the same operation order (the library is built with -ffp-contract=off) and counts the
// labels of the rows in between (ratio_indexing :30-36: count of each other segment inside
// the sector).  One thread = one column of one piece; labels are coherent along a column, so
// runs are aggregated in registers and each run costs one integer atomic.
//
// HBM traffic: the enlarged boxes are read once per sector that covers a pixel, 4 B/px.
#include "common.h"
#include "../../include/imgan.h"

namespace {

// pieces_i: [np][8] = slot, sector, xa, xb, lo_kind, hi_kind, lo_c, hi_c
// pieces_d: [np][4] = lo_a, lo_b, hi_a, hi_b
// kind: 0 constant c; 1 int(a*x+b); 2 max(int(a*x+b), c); 3 min(int(a*x+b), c)
__device__ __forceinline__ int bound_at(int kind, double a, double b, int c, int x) {
    if (kind == 0) return c;
    const double v = a * (double)x + b;              // float64, multiply then add (no fma)
    // int() of the reference truncates toward zero; rows are clamped to the image below, so
    // the saturation only keeps the conversion defined for absurd coefficients
    const int t = v >= 2147483647.0 ? 2147483647 : (v <= -2147483648.0 ? (int)0x80000000 : (int)v);
    if (kind == 2) return t > c ? t : c;
    if (kind == 3) return t < c ? t : c;
    return t;
}

__global__ __launch_bounds__(256) void spatial_sector_kernel(const int32_t* __restrict__ labels,
                                                             int H, int W, int nlab,
                                                             const int32_t* __restrict__ pieces_i,
                                                             const double* __restrict__ pieces_d,
                                                             int32_t* __restrict__ counts,
                                                             int32_t* __restrict__ oob) {
    const int p = blockIdx.x;
    const int32_t* pi = pieces_i + (size_t)p * 8;
    const double* pd = pieces_d + (size_t)p * 4;
    const int x = pi[2] + blockIdx.y * 256 + threadIdx.x;
    if (x >= pi[3]) return;
    const int lo = bound_at(pi[4], pd[0], pd[1], pi[6], x);
    const int hi = bound_at(pi[5], pd[2], pd[3], pi[7], x);
    if (lo >= hi) return;
    if (x < 0 || x >= W || lo < 0 || hi > H) {       // the reference would index out of the image
        atomicExch(oob, 1);
        return;
    }
    int32_t* dst = counts + ((size_t)pi[0] * 8 + pi[1]) * nlab;
    int cur = -1, run = 0;
    for (int y = lo; y < hi; ++y) {
        const int l = labels[(size_t)y * W + x];
        if (l == cur) {
            ++run;
        } else {
            if (run && cur >= 0 && cur < nlab) atomicAdd(&dst[cur], run);
            cur = l;
            run = 1;
        }
    }
    if (run && cur >= 0 && cur < nlab) atomicAdd(&dst[cur], run);
}

}  // namespace

extern "C" int imgan_spatial_sectors(const int32_t* labels, int H, int W, int nlab,
                                     const int32_t* pieces_i, const double* pieces_d, int npieces,
                                     int max_cols, int32_t* counts, int32_t* oob_flag,
                                     imgan_stream_t stream) {
    IMGAN_REQUIRE(labels && counts && oob_flag, "imgan_spatial_sectors: null pointer");
    IMGAN_REQUIRE(H >= 1 && W >= 1 && nlab >= 1, "imgan_spatial_sectors: bad shape");
    IMGAN_REQUIRE((long long)H * W < 0x7fffffffLL, "imgan_spatial_sectors: image too large");
    if (npieces == 0 || max_cols <= 0) return 0;
    IMGAN_REQUIRE(pieces_i && pieces_d && npieces > 0, "imgan_spatial_sectors: bad piece list");
    dim3 grid((unsigned)npieces, (unsigned)imgan::cdiv(max_cols, 256));
    IMGAN_REQUIRE(grid.y <= 65535, "imgan_spatial_sectors: piece too wide");
    hipLaunchKernelGGL(spatial_sector_kernel, grid, dim3(256), 0, (hipStream_t)stream, labels, H, W,
                       nlab, pieces_i, pieces_d, counts, oob_flag);
    return imgan::check_launch("spatial_sector_kernel");
}
