// 8-bit BGR -> CIELAB / gray (OpenCV fixed-point algorithm).
// Replaces cv2.cvtColor at slic.py:13, rag_segmentation.py:12, mpeg7Descriptors.py:22-23.
//
// HBM-bound streaming kernels: every lane moves 4 pixels = 12 B in / 12 B (or 4 B) out
// with dword accesses; the two lookup tables (512 B + 6 KiB) are staged in LDS once per
// workgroup and the grid is persistent (grid-stride) so the staging is amortised.
#include "common.h"
#include "../../include/imgan.h"
#include <string.h>

#include "lab_tables.inc"

namespace {

const uint16_t kGammaTab[256] = {IMGAN_GAMMA_VALUES};
const uint16_t kCbrtTab[3072] = {IMGAN_CBRT_VALUES};
const int32_t kLabCoeffs[9] = {IMGAN_LABCOEFF_VALUES};   // rows X,Y,Z over (R,G,B)

// statically initialised in the code object: present on every device, no upload step
__device__ const uint16_t d_gamma[256] = {IMGAN_GAMMA_VALUES};
__device__ const uint16_t d_cbrt[3072] = {IMGAN_CBRT_VALUES};

struct __attribute__((packed, aligned(4))) Dword3 {
    uint32_t a, b, c;
};

constexpr int LAB_SHIFT = 12, LAB_SHIFT2 = 15;
constexpr int LSCALE = (116 * 255 + 50) / 100;
constexpr int LSHIFT = -((16 * 255 * (1 << LAB_SHIFT2) + 50) / 100);

struct Lab3 {
    uint32_t L, a, b;
};

// a * b + c with 24-bit factors: one v_mad_i32_i24 (spelled out: left to itself the compiler turns a
// chain of three into three multiplies and a three-input add)
__device__ __forceinline__ int imad24(int a, int b, int c) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));   // b: wave uniform
    return r;
}

__device__ __forceinline__ Lab3 lab_px(const uint16_t* __restrict__ sg,
                                       const uint16_t* __restrict__ sc, uint32_t b8,
                                       uint32_t g8, uint32_t r8, const int* __restrict__ C) {
    // gamma values and coefficients are below 2^13 (12-bit fixed point), the cube-root values below
    // 2^16: 24-bit multiply-adds are exact here and a third of the instructions of the 32 / 64-bit
    // forms the compiler picks for plain ints (it cannot see the ranges)
    const int R = sg[r8], G = sg[g8], B = sg[b8];
    const int half = 1 << (LAB_SHIFT - 1);
    // table look-up at byte offset ((t >> 12) * 2) = (t >> 11) & ~1: a shift right and an AND are
    // 2-cycle instructions, the shift left the compiler would use is a 4-cycle one
    auto cbrt_at = [&](int t) {
        uint32_t off = (uint32_t)t >> (LAB_SHIFT - 1);
        asm("v_and_b32 %0, 0xfffffffe, %0" : "+v"(off));
        return (int)*reinterpret_cast<const uint16_t*>(reinterpret_cast<const unsigned char*>(sc) + off);
    };
    const int fX = cbrt_at(imad24(R, C[0], imad24(G, C[1], imad24(B, C[2], half))));
    const int fY = cbrt_at(imad24(R, C[3], imad24(G, C[4], imad24(B, C[5], half))));
    const int fZ = cbrt_at(imad24(R, C[6], imad24(G, C[7], imad24(B, C[8], half))));
    int L = imad24(fY, LSCALE, LSHIFT + (1 << (LAB_SHIFT2 - 1))) >> LAB_SHIFT2;
    int a = imad24(fX - fY, 500, 128 * (1 << LAB_SHIFT2) + (1 << (LAB_SHIFT2 - 1))) >> LAB_SHIFT2;
    int b = imad24(fY - fZ, 200, 128 * (1 << LAB_SHIFT2) + (1 << (LAB_SHIFT2 - 1))) >> LAB_SHIFT2;
    // Keep the shift and the clamp apart: hipcc (ROCm 7.2) fuses "sat_u8(x >> 15)" pairs into
    // v_ashr_pk_u8_i32 and then assumes bits 31:16 of its result are zero; on gfx950 they are
    // not, which corrupted the neighbouring byte of the packed output.
    asm volatile("" : "+v"(L), "+v"(a), "+v"(b));
    L = min(max(L, 0), 255);
    a = min(max(a, 0), 255);
    b = min(max(b, 0), 255);
    Lab3 o;
    o.L = (uint32_t)L & 255u;
    o.a = (uint32_t)a & 255u;
    o.b = (uint32_t)b & 255u;
    return o;
}

struct Coeffs {
    int c[9];
};

__global__ __launch_bounds__(256) void bgr2lab_kernel(const uint8_t* __restrict__ src,
                                                      uint8_t* __restrict__ dst, int64_t npix,
                                                      Coeffs cf) {
    __shared__ uint16_t sg[256];
    __shared__ uint16_t sc[3072];
    for (int i = threadIdx.x; i < 256; i += 256) sg[i] = d_gamma[i];
    for (int i = threadIdx.x; i < 3072; i += 256) sc[i] = d_cbrt[i];
    __syncthreads();
    const int64_t nquad = npix >> 2;
    const Dword3* __restrict__ s3 = reinterpret_cast<const Dword3*>(src);
    Dword3* __restrict__ d3 = reinterpret_cast<Dword3*>(dst);
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nquad;
         q += (int64_t)gridDim.x * 256) {
        Dword3 v = s3[q];
        // bytes: b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
        Lab3 p0 = lab_px(sg, sc, v.a & 255, (v.a >> 8) & 255, (v.a >> 16) & 255, cf.c);
        Lab3 p1 = lab_px(sg, sc, v.a >> 24, v.b & 255, (v.b >> 8) & 255, cf.c);
        Lab3 p2 = lab_px(sg, sc, (v.b >> 16) & 255, v.b >> 24, v.c & 255, cf.c);
        Lab3 p3 = lab_px(sg, sc, (v.c >> 8) & 255, (v.c >> 16) & 255, v.c >> 24, cf.c);
        Dword3 o;
        o.a = p0.L | (p0.a << 8) | (p0.b << 16) | (p1.L << 24);
        o.b = p1.a | (p1.b << 8) | (p2.L << 16) | (p2.a << 24);
        o.c = p2.b | (p3.L << 8) | (p3.a << 16) | (p3.b << 24);
        d3[q] = o;
    }
    // tail (npix % 4 pixels), one lane each
    if (blockIdx.x == 0 && threadIdx.x < (npix & 3)) {
        int64_t p = (nquad << 2) + threadIdx.x;
        Lab3 o = lab_px(sg, sc, src[3 * p], src[3 * p + 1], src[3 * p + 2], cf.c);
        dst[3 * p] = (uint8_t)o.L;
        dst[3 * p + 1] = (uint8_t)o.a;
        dst[3 * p + 2] = (uint8_t)o.b;
    }
}

__device__ __forceinline__ uint32_t gray_px(uint32_t b, uint32_t g, uint32_t r) {
    return (4899u * r + 9617u * g + 1868u * b + 8192u) >> 14;
}

__global__ __launch_bounds__(256) void bgr2gray_kernel(const uint8_t* __restrict__ src,
                                                       uint8_t* __restrict__ dst, int64_t npix) {
    const int64_t nquad = npix >> 2;
    const Dword3* __restrict__ s3 = reinterpret_cast<const Dword3*>(src);
    uint32_t* __restrict__ d1 = reinterpret_cast<uint32_t*>(dst);
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nquad;
         q += (int64_t)gridDim.x * 256) {
        Dword3 v = s3[q];
        uint32_t g0 = gray_px(v.a & 255, (v.a >> 8) & 255, (v.a >> 16) & 255);
        uint32_t g1 = gray_px(v.a >> 24, v.b & 255, (v.b >> 8) & 255);
        uint32_t g2 = gray_px((v.b >> 16) & 255, v.b >> 24, v.c & 255);
        uint32_t g3 = gray_px((v.c >> 8) & 255, (v.c >> 16) & 255, v.c >> 24);
        d1[q] = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    }
    if (blockIdx.x == 0 && threadIdx.x < (npix & 3)) {
        int64_t p = (nquad << 2) + threadIdx.x;
        dst[p] = (uint8_t)gray_px(src[3 * p], src[3 * p + 1], src[3 * p + 2]);
    }
}

int grid_for(int64_t nquad) {
    int64_t blocks = imgan::cdiv(nquad, 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8;
    return (int)blocks;
}

}  // namespace

extern "C" int imgan_lab_tables_host(uint16_t* gamma256, uint16_t* cbrt3072, int32_t* coeffs9) {
    IMGAN_REQUIRE(gamma256 && cbrt3072 && coeffs9, "imgan_lab_tables_host: null output");
    memcpy(gamma256, kGammaTab, sizeof(kGammaTab));
    memcpy(cbrt3072, kCbrtTab, sizeof(kCbrtTab));
    memcpy(coeffs9, kLabCoeffs, sizeof(kLabCoeffs));
    return 0;
}

extern "C" int imgan_bgr2lab_u8(const uint8_t* bgr, uint8_t* lab, int64_t npix,
                                imgan_stream_t stream) {
    IMGAN_REQUIRE(bgr && lab, "imgan_bgr2lab_u8: null pointer");
    IMGAN_REQUIRE(npix >= 0, "imgan_bgr2lab_u8: negative pixel count");
    IMGAN_REQUIRE(((uintptr_t)bgr & 3) == 0 && ((uintptr_t)lab & 3) == 0,
                  "imgan_bgr2lab_u8: buffers must be 4-byte aligned");
    if (npix == 0) return 0;
    Coeffs cf;
    for (int i = 0; i < 9; ++i) cf.c[i] = kLabCoeffs[i];
    hipLaunchKernelGGL(bgr2lab_kernel, dim3(grid_for(npix >> 2)), dim3(256), 0,
                       (hipStream_t)stream, bgr, lab, npix, cf);
    return imgan::check_launch("bgr2lab_kernel");
}

extern "C" int imgan_bgr2gray_u8(const uint8_t* bgr, uint8_t* gray, int64_t npix,
                                 imgan_stream_t stream) {
    IMGAN_REQUIRE(bgr && gray, "imgan_bgr2gray_u8: null pointer");
    IMGAN_REQUIRE(npix >= 0, "imgan_bgr2gray_u8: negative pixel count");
    IMGAN_REQUIRE(((uintptr_t)bgr & 3) == 0 && ((uintptr_t)gray & 3) == 0,
                  "imgan_bgr2gray_u8: buffers must be 4-byte aligned");
    if (npix == 0) return 0;
    hipLaunchKernelGGL(bgr2gray_kernel, dim3(grid_for(npix >> 2)), dim3(256), 0,
                       (hipStream_t)stream, bgr, gray, npix);
    return imgan::check_launch("bgr2gray_kernel");
}
