// 7x7 median blur of a 3-channel u8 image, replicated border.
// Replaces cv2.medianBlur(image, 7) at rag_segmentation.py:11 (exact integer median).
//
// One workgroup = a 32 x 8 output tile; the 38 x 14 input patch is staged in LDS with the
// three channels re-packed into 9-bit fields of one dword (b | g<<9 | r<<18), so the
// selection runs on all three channels at once: the median (25th of 49) is found by an
// 8-step binary search on the value; one add per sample tests "sample >= candidate" for the
// three channels (field + 256 - t sets bit 8 of the field iff sample >= t, and never carries
// into the next field: 255 + 255 < 512).  The flags are counted where they stand: once the
// data bits are masked off, the six bits above each flag are free (8-13, 17-22, 26-31), so a
// count of up to 49 needs no shift -- an and and (for two samples) one three-operand add.
#include "common.h"
#include "../../include/imgan.h"

namespace {

constexpr int MT_W = 32, MT_H = 8, R = 3, PW = MT_W + 2 * R, PH = MT_H + 2 * R;

__global__ __launch_bounds__(256) void median7_kernel(const uint8_t* __restrict__ src,
                                                      uint8_t* __restrict__ dst, int H, int W) {
    __shared__ uint32_t patch[PH * PW];
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * MT_W, y0 = blockIdx.y * MT_H;
    const uint8_t* im = src + (size_t)b * H * W * 3;
    for (int i = threadIdx.x; i < PH * PW; i += 256) {
        int py = i / PW, px = i - py * PW;
        int yy = min(max(y0 + py - R, 0), H - 1), xx = min(max(x0 + px - R, 0), W - 1);
        const uint8_t* q = im + ((size_t)yy * W + xx) * 3;
        patch[i] = (uint32_t)q[0] | ((uint32_t)q[1] << 9) | ((uint32_t)q[2] << 18);
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x = x0 + tx, y = y0 + ty;
    uint32_t v[49];
#pragma unroll
    for (int dy = 0; dy < 7; ++dy)
#pragma unroll
        for (int dx = 0; dx < 7; ++dx) v[dy * 7 + dx] = patch[(ty + dy) * PW + tx + dx];
    uint32_t t = 0;                         // packed candidate, 9-bit fields
    constexpr uint32_t FLAGS = (1u << 8) | (1u << 17) | (1u << 26);
#pragma unroll
    for (int bit = 7; bit >= 0; --bit) {
        const uint32_t trial = t | (0x00040201u << bit);
        const uint32_t add = 0x04020100u - trial;        // (256 - t) per field, t >= 1: no borrows
        uint32_t cnt = (v[48] + add) & FLAGS;
#pragma unroll
        for (int k = 0; k < 48; k += 2) cnt += ((v[k] + add) & FLAGS) + ((v[k + 1] + add) & FLAGS);
        // keep the bit in every field whose count of (sample >= trial) reaches 25
        uint32_t keep = 0;
        if (((cnt >> 8) & 63u) >= 25u) keep |= 1u << bit;
        if (((cnt >> 17) & 63u) >= 25u) keep |= 1u << (bit + 9);
        if ((cnt >> 26) >= 25u) keep |= 1u << (bit + 18);
        t |= keep;
    }
    if (x < W && y < H) {
        uint8_t* o = dst + (((size_t)b * H + y) * W + x) * 3;
        o[0] = (uint8_t)(t & 255u);
        o[1] = (uint8_t)((t >> 9) & 255u);
        o[2] = (uint8_t)((t >> 18) & 255u);
    }
}

}  // namespace

extern "C" int imgan_median7_u8c3(const uint8_t* src, uint8_t* dst, int B, int H, int W,
                                  imgan_stream_t stream) {
    IMGAN_REQUIRE(src && dst, "imgan_median7_u8c3: null pointer");
    IMGAN_REQUIRE(src != dst, "imgan_median7_u8c3: in-place operation is not supported");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && B <= 65535, "imgan_median7_u8c3: bad shape");
    dim3 grid((unsigned)imgan::cdiv(W, MT_W), (unsigned)imgan::cdiv(H, MT_H), (unsigned)B);
    hipLaunchKernelGGL(median7_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, H, W);
    return imgan::check_launch("median7_kernel");
}
