// Connected components of equal-valued, 8-connected pixels by label-propagation union-find.
// Replaces the recursive flood fill of mpeg7Descriptors.py:40-74 as used at :118-128
// (spatial coherence of a dominant colour).  True 8-connectivity: the reference clamps its
// neighbour window at 255 (:52,:54) and is only correct for images up to 255 x 255.
//
// parent[p] starts as p; every pixel unites itself with its W, NW, N, NE neighbours of equal
// value (atomicMin hooks the larger root under the smaller one); a final pass flattens, so
// roots[p] is the smallest raster index of p's component.  Output is independent of the
// order in which the unions happen.
#include "common.h"
#include "../../include/imgan.h"
#include "cc_device.h"

namespace {

__global__ __launch_bounds__(256) void cc_init_kernel(const int32_t* __restrict__ values,
                                                      int32_t* __restrict__ parent,
                                                      long long npix) {
    const int b = blockIdx.y;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix;
         p += (long long)gridDim.x * 256)
        parent[(size_t)b * npix + p] = values[(size_t)b * npix + p] >= 0 ? (int32_t)p : -1;
}

__global__ __launch_bounds__(256) void cc_merge_kernel(const int32_t* __restrict__ values,
                                                       int32_t* __restrict__ parent, int H,
                                                       int W) {
    const int b = blockIdx.y;
    const long long npix = (long long)H * W;
    const int32_t* V = values + (size_t)b * npix;
    int32_t* P = parent + (size_t)b * npix;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix;
         p += (long long)gridDim.x * 256) {
        int v = V[p];
        if (v < 0) continue;
        int y = (int)(p / W), x = (int)(p - (long long)y * W);
        if (x > 0 && V[p - 1] == v) imgan::cc_union(P, (int)p, (int)p - 1);
        if (y > 0) {
            if (V[p - W] == v) imgan::cc_union(P, (int)p, (int)(p - W));
            if (x > 0 && V[p - W - 1] == v) imgan::cc_union(P, (int)p, (int)(p - W - 1));
            if (x + 1 < W && V[p - W + 1] == v) imgan::cc_union(P, (int)p, (int)(p - W + 1));
        }
    }
}

__global__ __launch_bounds__(256) void cc_flatten_kernel(int32_t* __restrict__ parent,
                                                         long long npix) {
    const int b = blockIdx.y;
    int32_t* P = parent + (size_t)b * npix;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix;
         p += (long long)gridDim.x * 256)
        if (P[p] >= 0) P[p] = imgan::cc_find(P, (int)p);
}

}  // namespace

extern "C" int imgan_cc_label(const int32_t* values, int32_t* roots, int B, int H, int W,
                              imgan_stream_t stream) {
    IMGAN_REQUIRE(values && roots, "imgan_cc_label: null pointer");
    IMGAN_REQUIRE(values != roots, "imgan_cc_label: in-place operation is not supported");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && B <= 65535, "imgan_cc_label: bad shape");
    IMGAN_REQUIRE((long long)H * W < 0x7fffffffLL, "imgan_cc_label: image too large for int32 ids");
    long long npix = (long long)H * W;
    dim3 grid((unsigned)std::min<long long>(imgan::cdiv(npix, 256), 16384), (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cc_init_kernel, grid, dim3(256), 0, st, values, roots, npix);
    hipLaunchKernelGGL(cc_merge_kernel, grid, dim3(256), 0, st, values, roots, H, W);
    hipLaunchKernelGGL(cc_flatten_kernel, grid, dim3(256), 0, st, roots, npix);
    return imgan::check_launch("imgan_cc_label");
}
