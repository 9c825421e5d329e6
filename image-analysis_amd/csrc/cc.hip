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
#include <algorithm>

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

// ---- connectivity enforcement of a superpixel label map (named by the task; scripts/slic.py has
// no such step, so the rule is this library's own, see imgan.h).  Three light passes after the
// component labelling: sizes by one atomic per pixel on the component's root, the largest
// component per label by a 64-bit atomic max of (size, first raster pixel reversed), then the
// new label per root and a gather.
namespace {

__global__ __launch_bounds__(256) void conn_size_kernel(const int32_t* __restrict__ roots,
                                                        int32_t* __restrict__ size, long long npix) {
    const size_t base = (size_t)blockIdx.y * npix;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        const int r = roots[base + p];
        if (r >= 0) atomicAdd(&size[base + r], 1);
    }
}

__device__ __forceinline__ unsigned long long conn_key(int size, int root) {
    return ((unsigned long long)(uint32_t)size << 32) | (unsigned long long)(0xffffffffu - (uint32_t)root);
}

__global__ __launch_bounds__(256) void conn_best_kernel(const int32_t* __restrict__ labels,
                                                        const int32_t* __restrict__ roots,
                                                        const int32_t* __restrict__ size,
                                                        unsigned long long* __restrict__ best,
                                                        long long npix, int nlabels) {
    const size_t base = (size_t)blockIdx.y * npix;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        if (roots[base + p] != (int)p) continue;             // p is its component's first raster pixel
        const int l = labels[base + p];
        if (l >= 0 && l < nlabels)
            atomicMax(&best[(size_t)blockIdx.y * nlabels + l], conn_key(size[base + p], (int)p));
    }
}

// size[root] := the label the component ends up with
__global__ __launch_bounds__(256) void conn_decide_kernel(const int32_t* __restrict__ labels,
                                                          const int32_t* __restrict__ roots,
                                                          int32_t* __restrict__ size,
                                                          const unsigned long long* __restrict__ best,
                                                          long long npix, int W, int nlabels) {
    const size_t base = (size_t)blockIdx.y * npix;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        if (roots[base + p] != (int)p) continue;
        const int l = labels[base + p];
        int out = l;
        const bool keeps = l >= nlabels || best[(size_t)blockIdx.y * nlabels + l] == conn_key(size[base + p], (int)p);
        if (!keeps) {
            const long long nb = (p % W > 0) ? p - 1 : p - W;   // left of the first pixel, else above it
            const int nl = nb >= 0 ? labels[base + nb] : -1;
            if (nl >= 0) out = nl;
        }
        size[base + p] = out;
    }
}

__global__ __launch_bounds__(256) void conn_apply_kernel(const int32_t* __restrict__ labels,
                                                         const int32_t* __restrict__ roots,
                                                         const int32_t* __restrict__ newlab,
                                                         int32_t* __restrict__ out, long long npix) {
    const size_t base = (size_t)blockIdx.y * npix;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        const int r = roots[base + p];
        out[base + p] = r >= 0 ? newlab[base + r] : labels[base + p];
    }
}

}  // namespace

extern "C" size_t imgan_enforce_connectivity_scratch_bytes(int B, int H, int W, int nlabels) {
    if (B < 1 || H < 1 || W < 1 || nlabels < 1) return 0;
    return (size_t)B * H * W * 4 * 2 + (size_t)B * nlabels * 8;
}

extern "C" int imgan_enforce_connectivity(const int32_t* labels, int32_t* out, int B, int H, int W,
                                          int nlabels, void* scratch, imgan_stream_t stream) {
    IMGAN_REQUIRE(labels && out && scratch && labels != out, "imgan_enforce_connectivity: bad pointers");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && nlabels >= 1 && B <= 65535,
                  "imgan_enforce_connectivity: bad shape");
    IMGAN_REQUIRE((long long)H * W < 0x7fffffffLL, "imgan_enforce_connectivity: image too large for int32 ids");
    const long long npix = (long long)H * W;
    int32_t* roots = (int32_t*)scratch;
    int32_t* size = roots + (size_t)B * npix;
    unsigned long long* best = (unsigned long long*)(size + (size_t)B * npix);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = imgan_cc_label(labels, roots, B, H, W, stream)) return rc;
    IMGAN_HIP(hipMemsetAsync(size, 0, (size_t)B * npix * 4 + (size_t)B * nlabels * 8, st));
    dim3 grid((unsigned)std::min<long long>(imgan::cdiv(npix, 256), 16384), (unsigned)B);
    hipLaunchKernelGGL(conn_size_kernel, grid, dim3(256), 0, st, roots, size, npix);
    hipLaunchKernelGGL(conn_best_kernel, grid, dim3(256), 0, st, labels, roots, size, best, npix, nlabels);
    hipLaunchKernelGGL(conn_decide_kernel, grid, dim3(256), 0, st, labels, roots, size, best, npix, W, nlabels);
    hipLaunchKernelGGL(conn_apply_kernel, grid, dim3(256), 0, st, labels, roots, size, out, npix);
    return imgan::check_launch("imgan_enforce_connectivity");
}
