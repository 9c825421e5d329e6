// cv2.resize(img, (w, h)) with the default INTER_LINEAR for 8-bit, 3-channel images, as
// optigen_utils.prepare_image (:19-24) and basegen.py:28-30 call it: OpenCV's fixed-point
// bilinear filter (11-bit weights, the row pass in int, the column pass with its >> 4, >> 16,
// + 2, >> 2 chain), and the exact 2 x 2 reduction that OpenCV routes to INTER_AREA.  Restated
// from the published algorithm (imgproc/resize.cpp); oracle/resize_ref.py states it in numpy.
//
// One thread per destination pixel: source taps and weights are recomputed per thread (twenty
// instructions; the kernel moves 3 B in and reads 12 B scattered over two source rows).
#include "common.h"
#include "../../include/imgan.h"
#include <math.h>

namespace {

struct Tap {
    int s0, s1, w0, w1;
};

__device__ __forceinline__ Tap make_tap(int d, double scale, int src) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) f = 0.f, s = 0;
    if (s >= src - 1) f = 0.f, s = src - 1;
    Tap t;
    t.s0 = s;
    t.s1 = min(s + 1, src - 1);
    t.w0 = (int)rintf((1.f - f) * 2048.f);
    t.w1 = (int)rintf(f * 2048.f);
    return t;
}

__global__ __launch_bounds__(256) void resize_linear_kernel(const uint8_t* __restrict__ src,
                                                            uint8_t* __restrict__ dst, int Hs, int Ws,
                                                            int Hd, int Wd, double sx, double sy,
                                                            int area2) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= Wd || y >= Hd) return;
    const uint8_t* im = src + (size_t)b * Hs * Ws * 3;
    uint8_t* o = dst + (((size_t)b * Hd + y) * Wd + x) * 3;
    if (area2) {
        const uint8_t* p = im + ((size_t)(2 * y) * Ws + 2 * x) * 3;
        const uint8_t* q = p + (size_t)Ws * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (uint8_t)((p[c] + p[3 + c] + q[c] + q[3 + c] + 2) >> 2);
        return;
    }
    const Tap tx = make_tap(x, sx, Ws), ty = make_tap(y, sy, Hs);
    const uint8_t* r0 = im + (size_t)ty.s0 * Ws * 3;
    const uint8_t* r1 = im + (size_t)ty.s1 * Ws * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int h0 = r0[tx.s0 * 3 + c] * tx.w0 + r0[tx.s1 * 3 + c] * tx.w1;
        const int h1 = r1[tx.s0 * 3 + c] * tx.w0 + r1[tx.s1 * 3 + c] * tx.w1;
        const int v = (((ty.w0 * (h0 >> 4)) >> 16) + ((ty.w1 * (h1 >> 4)) >> 16) + 2) >> 2;
        o[c] = (uint8_t)min(max(v, 0), 255);
    }
}

}  // namespace

extern "C" int imgan_resize_linear_u8c3(const uint8_t* src, uint8_t* dst, int B, int Hs, int Ws, int Hd,
                                        int Wd, imgan_stream_t stream) {
    IMGAN_REQUIRE(src && dst && src != dst, "imgan_resize_linear_u8c3: bad pointers");
    IMGAN_REQUIRE(B >= 1 && Hs >= 1 && Ws >= 1 && Hd >= 1 && Wd >= 1 && B <= 65535,
                  "imgan_resize_linear_u8c3: bad shape");
    IMGAN_REQUIRE(imgan::cdiv(Hd, 4) <= 65535, "imgan_resize_linear_u8c3: destination too tall");
    // OpenCV: inv_scale = dsize / ssize (double), scale = 1 / inv_scale
    const double sx = 1.0 / ((double)Wd / (double)Ws), sy = 1.0 / ((double)Hd / (double)Hs);
    const int area2 = (Ws == 2 * Wd && Hs == 2 * Hd) ? 1 : 0;
    dim3 grid((unsigned)imgan::cdiv(Wd, 64), (unsigned)imgan::cdiv(Hd, 4), (unsigned)B);
    hipLaunchKernelGGL(resize_linear_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, Hs, Ws, Hd,
                       Wd, sx, sy, area2);
    return imgan::check_launch("resize_linear_kernel");
}
