// MPEG-7 colour layout (CLD), edge histogram (EHD) and colour structure (CSD) descriptors.
// NOT in the reference; defined from ISO/IEC 15938-3 / the XM
// software in oracle/mpeg7_std.py, which these kernels follow bit for bit.  Every decision is
// taken in exact integer arithmetic; the only floating point is the CLD colour transform and
// 8x8 DCT, in double with a fixed operation order (build uses -ffp-contract=off).
//
// All three are one streaming pass over the BGR image (3 B/px; CSD only touches the
// sub-sampled grid).  The CLD / EHD sums walk row strips with the column-dependent indices
// fixed per thread and the sums in registers; CSD builds its 256-bin histogram in LDS
// (privatised per workgroup) and flushes once.
#include "common.h"
#include "dcd_lists.h"
#include "../../include/imgan.h"
#include <math.h>
#include <atomic>

namespace {

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v |= __shfl_xor(v, off);
    return v;
}

// OR over the wave as a scalar: four DPP steps leave every row of 16 lanes with its own OR (no
// LDS round trips, unlike the shuffles above), four lane reads join the rows
__device__ __forceinline__ uint32_t wave_or_scalar(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);   // row_half_mirror
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);   // row_mirror
    return (uint32_t)(__builtin_amdgcn_readlane((int)v, 0) | __builtin_amdgcn_readlane((int)v, 16) |
                      __builtin_amdgcn_readlane((int)v, 32) | __builtin_amdgcn_readlane((int)v, 48));
}

__device__ __forceinline__ uint32_t gray_px(uint32_t b, uint32_t g, uint32_t r) {
    return (4899u * r + 9617u * g + 1868u * b + 8192u) >> 14;
}

// ---- row-strip helpers shared by the CLD and EHD sums.  A thread owns 4 adjacent pixel
// columns and walks down a strip of rows, so everything that depends on the column (grid
// block, sub-block quadrant) is fixed per thread and the sums stay in registers until the
// strip ends.  Four rows are requested before the first is used; rows past the strip are
// clamped and masked instead of branched around.
struct __attribute__((packed, aligned(4))) Bgr4 {
    uint32_t a, b, c;
};

// px[j] = B | G << 8 | R << 16 of pixel x + j (0 outside the image)
template <bool ALIGNED>
__device__ __forceinline__ void load_quad(const uint8_t* __restrict__ row, int x, int W,
                                          uint32_t mask, uint32_t* px) {
    if (ALIGNED) {                                   // W % 4 == 0: 12 aligned bytes
        const Bgr4 v = *reinterpret_cast<const Bgr4*>(row + (size_t)x * 3);
        px[0] = v.a & 0xffffffu & mask;
        px[1] = __builtin_amdgcn_perm(v.b, v.a, 0x0c050403u) & mask;
        px[2] = __builtin_amdgcn_perm(v.c, v.b, 0x0c040302u) & mask;
        px[3] = (v.c >> 8) & mask;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = x + j < W ? x + j : W - 1;
            const uint8_t* q = row + (size_t)xx * 3;
            const uint32_t v = q[0] | (q[1] << 8) | (q[2] << 16);
            px[j] = x + j < W ? (v & mask) : 0u;
        }
    }
}

// ------------------------------------------------------------------------------ CLD
// sums: u64 [B][64][4] = B, G, R sums and pixel count of the 8x8 grid blocks (zeroed by the
// host wrapper).  Grid row gy owns the image rows with (y * 8) / H == gy; one workgroup takes
// a quarter of such a band at full width.
constexpr int CLD_PARTS = 4;

template <bool ALIGNED>
__global__ __launch_bounds__(256) void cld_sums_kernel(const uint8_t* __restrict__ img, int H,
                                                       int W, unsigned long long* __restrict__ sums) {
    // 16 copies of the sums (a lane adds to copy lane % 16): the 256 columns of a wave fall into two
    // or three grid columns, and lanes that add to one LDS word take turns
    constexpr int COPIES = 16, PITCH = 33;
    __shared__ unsigned long long tab[COPIES][PITCH];
    const int b = blockIdx.z, gy = blockIdx.y / CLD_PARTS, part = blockIdx.y % CLD_PARTS;
    const int r0 = (gy * H + 7) / 8, r1 = ((gy + 1) * H + 7) / 8;
    const int per = (r1 - r0 + CLD_PARTS - 1) / CLD_PARTS;
    const int ya = r0 + part * per, yb = min(ya + per, r1);
    const int tid = threadIdx.x;
    for (int i = tid; i < COPIES * PITCH; i += 256) (&tab[0][0])[i] = 0ull;
    __syncthreads();
    const int copy = tid & (COPIES - 1);
    const uint8_t* im = img + (size_t)b * H * W * 3;
    const int nquad = (W + 3) / 4;
    for (int q0 = 0; q0 < nquad && ya < yb; q0 += 256) {
        const int x = (q0 + tid) * 4;
        const bool live = x < W;
        const int xl = live ? x : 0;
        uint32_t e[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0};     // B | R << 16 and G
        int nrows = 0;
        auto flush = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (live && x + j < W) {
                    const int gx = ((x + j) * 8) / W;
                    unsigned long long* t4 = &tab[copy][gx * 4];
                    atomicAdd(t4, (unsigned long long)(e[j] & 0xffffu));
                    atomicAdd(t4 + 1, (unsigned long long)o[j]);
                    atomicAdd(t4 + 2, (unsigned long long)(e[j] >> 16));
                    atomicAdd(t4 + 3, (unsigned long long)nrows);
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = o[j] = 0;
            nrows = 0;
        };
        for (int y = ya; y < yb; y += 4) {
            uint32_t px[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int yy = y + u < yb ? y + u : yb - 1;
                load_quad<ALIGNED>(im + (size_t)yy * W * 3, xl, W, y + u < yb ? 0xffffffffu : 0u,
                                   px[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    e[j] += px[u][j] & 0xff00ffu;
                    o[j] += (px[u][j] >> 8) & 0xffu;
                }
            nrows += min(4, yb - y);
            if (nrows >= 252) flush();                 // the 16-bit halves of e hold 257 rows
        }
        flush();
    }
    __syncthreads();
    if (tid < 32) {
        unsigned long long v = 0;
#pragma unroll
        for (int c = 0; c < COPIES; ++c) v += tab[c][tid];
        if (v) atomicAdd(&sums[((size_t)b * 64 + gy * 8 + (tid >> 2)) * 4 + (tid & 3)], v);
    }
}

struct DctTable {
    double c[64];
};

__device__ const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                        12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                        35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                        58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

__device__ int quant_ydc(int i) {
    if (i > 192) return 112 + (i - 192) / 4;
    if (i > 160) return 96 + (i - 160) / 2;
    if (i > 96) return 32 + (i - 96);
    if (i > 64) return 16 + (i - 64) / 2;
    return i >= 0 ? i / 4 : -((-i + 3) / 4);          // floor division like the oracle's '//'
}
__device__ int quant_cdc(int i) {
    if (i > 191) return 63;
    if (i > 160) return 56 + (i - 160) / 4;
    if (i > 144) return 48 + (i - 144) / 2;
    if (i > 112) return 16 + (i - 112);
    if (i > 96) return 8 + (i - 96) / 2;
    if (i > 64) return (i - 64) / 4;
    return 0;
}
__device__ int quant_ac(int i) {
    i = min(i, 239);
    i = max(i, -256);
    int a = abs(i), j;
    if (a > 127) j = 64 + a / 4;
    else if (a > 63) j = 32 + a / 2;
    else j = a;
    j = i < 0 ? -j : j;
    return j + 128;
}

// one wave per (image, channel), one lane per grid cell / coefficient: averages -> YCbCr ->
// 8x8 DCT -> zig-zag -> quantise.  Every sum runs over k in the order of the serial statement.
// fill_empty: a grid block without pixels takes the mean of all blocks' pixels (the per-region
// form: blocks of the bounding box that the region does not reach)
__global__ __launch_bounds__(64) void cld_finalise_kernel(const unsigned long long* __restrict__ sums,
                                                          int B, int32_t* __restrict__ out,
                                                          DctTable T, int fill_empty) {
    __shared__ double s_t[64], blk[64], tmp[64];
    __shared__ int coef[64];
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= B * 3) return;
    const int b = t / 3, ch = t - b * 3;
    const unsigned long long* S = sums + (size_t)b * 64 * 4;
    s_t[lane] = T.c[lane];
    {
        const int k = lane;
        unsigned long long n = S[4 * k + 3];
        unsigned long long s0 = S[4 * k], s1 = S[4 * k + 1], s2 = S[4 * k + 2];
        if (fill_empty) {                              // totals over the 64 blocks, every lane
            unsigned long long t0 = s0, t1 = s1, t2 = s2, tn = n;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                t0 += __shfl_xor(t0, off);
                t1 += __shfl_xor(t1, off);
                t2 += __shfl_xor(t2, off);
                tn += __shfl_xor(tn, off);
            }
            if (tn == 0) {                             // a region without pixels (or left out): zeros
                out[((size_t)b * 3 + ch) * 64 + lane] = 0;
                return;
            }
            if (n == 0) s0 = t0, s1 = t1, s2 = t2, n = tn;
        }
        double Bc = 0.0, Gc = 0.0, Rc = 0.0;
        if (n) {
            Bc = (double)(s0 / n);
            Gc = (double)(s1 / n);
            Rc = (double)(s2 / n);
        }
        double yy = (0.299 * Rc + 0.587 * Gc + 0.114 * Bc) / 256.0;
        double v;
        if (ch == 0) v = (double)(int)(219.0 * yy + 16.5);
        else if (ch == 1) v = (double)(int)(224.0 * 0.564 * (Bc / 256.0 * 1.0 - yy) + 128.5);
        else v = (double)(int)(224.0 * 0.713 * (Rc / 256.0 * 1.0 - yy) + 128.5);
        blk[k] = v;
    }
    __syncthreads();
    const int i = lane >> 3, j = lane & 7;
    {
        double s = 0.0;
        for (int k = 0; k < 8; ++k) s += s_t[j * 8 + k] * blk[8 * i + k];
        tmp[8 * i + j] = s;
    }
    __syncthreads();
    {
        double s = 0.0;
        for (int k = 0; k < 8; ++k) s += s_t[i * 8 + k] * tmp[8 * k + j];
        coef[8 * i + j] = (int)floor(s + 0.499999);
    }
    __syncthreads();
    int32_t* o = out + ((size_t)b * 3 + ch) * 64;
    if (lane == 0) {
        o[0] = ch == 0 ? (quant_ydc(coef[0] / 8) >> 1) : quant_cdc(coef[0] / 8);
    } else {
        const int v = coef[kZigzag[lane]];
        o[lane] = ch == 0 ? (quant_ac(v / 2) >> 3) : (quant_ac(v) >> 3);
    }
}

// ------------------------------------------------------------------------------ EHD
// qsum: u32 [B][nby*nbx][4] quadrant gray sums.  One workgroup owns one row of quadrants
// (sub-block row by, upper or lower half) at full width, so it writes its sums with plain
// stores.  tab: dynamic LDS, nbx * 2 words.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void ehd_sums_kernel(const uint8_t* __restrict__ img, int H,
                                                       int W, int bs, int nbx, int nby,
                                                       uint32_t* __restrict__ qsum) {
    extern __shared__ __attribute__((aligned(16))) uint32_t ehd_tab[];
    const int b = blockIdx.y, by = blockIdx.x >> 1, qy = blockIdx.x & 1;
    const int hb = bs >> 1;
    const int ya = by * bs + qy * hb, yb = ya + hb;
    const int tid = threadIdx.x;
    for (int i = tid; i < nbx * 2; i += 256) ehd_tab[i] = 0;
    __syncthreads();
    const uint8_t* im = img + (size_t)b * H * W * 3;
    const int wuse = nbx * bs;                          // columns right of it belong to no block
    const int nquad = (wuse + 3) / 4;
    for (int q0 = 0; q0 < nquad; q0 += 256) {
        const int x = (q0 + tid) * 4;
        const bool live = x < wuse;
        const int xl = live ? x : 0;
        uint32_t g[4] = {0, 0, 0, 0};
        for (int y = ya; y < yb; y += 4) {
            uint32_t px[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int yy = y + u < yb ? y + u : yb - 1;
                load_quad<ALIGNED>(im + (size_t)yy * W * 3, xl, W, y + u < yb ? 0xffffffffu : 0u,
                                   px[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j)            // a masked pixel has gray 0
                    g[j] += gray_px(px[u][j] & 0xffu, (px[u][j] >> 8) & 0xffu, px[u][j] >> 16);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (live && x + j < wuse) {
                const int bx = (x + j) / bs;
                const int qx = (x + j - bx * bs) >= hb;
                atomicAdd(&ehd_tab[bx * 2 + qx], g[j]);
            }
    }
    __syncthreads();
    uint32_t* Q = qsum + ((size_t)b * nby + by) * nbx * 4;
    for (int i = tid; i < nbx * 2; i += 256) Q[(i >> 1) * 4 + qy * 2 + (i & 1)] = ehd_tab[i];
}

// ---- colour layout sums and edge quadrant sums in ONE pass over the image (imgan_cld_ehd): both
// walk strips of rows with the same four-columns-per-thread mapping, and each alone is bound by
// reading the image.  A workgroup takes the rows of one quadrant half (hb = bs / 2 rows, as
// ehd_sums_kernel; strips below the last edge block serve the colour layout alone).  A strip lies
// in at most two rows of the 8 x 8 grid (the host checks hb * 8 <= H): the colour sums of the
// rows above and below that boundary go to two LDS tables.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void cld_ehd_sums_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                           int bs, int nbx, int nby,
                                                           unsigned long long* __restrict__ sums,
                                                           uint32_t* __restrict__ qsum) {
    extern __shared__ __attribute__((aligned(16))) uint32_t ehd_tab[];
    // colour sums in 16 copies (a lane adds to copy lane % 16): the 256 columns of a wave fall into
    // two or three grid columns, and lanes that add to one LDS word take turns
    constexpr int CE_COPIES = 16, CE_PITCH = 33;
    __shared__ unsigned long long tab[2][CE_COPIES][CE_PITCH];
    const int b = blockIdx.y, strip = blockIdx.x, hb = bs >> 1;
    const int ya = strip * hb, yb = min(ya + hb, H);
    const bool edge = strip < nby * 2;
    const int by = strip >> 1, qy = strip & 1;
    const int tid = threadIdx.x;
    for (int i = tid; i < nbx * 2; i += 256) ehd_tab[i] = 0;
    for (int i = tid; i < 2 * CE_COPIES * CE_PITCH; i += 256) (&tab[0][0][0])[i] = 0ull;
    __syncthreads();
    const int copy = tid & (CE_COPIES - 1);
    const int gy0 = (int)(((long long)ya * 8) / H);
    const int yc = min((int)((((long long)gy0 + 1) * H + 7) / 8), yb);   // first row of the next grid row
    const uint8_t* im = img + (size_t)b * H * W * 3;
    const int wuse = nbx * bs;                          // columns right of it belong to no edge block
    const int nquad = (W + 3) / 4;
    for (int q0 = 0; q0 < nquad; q0 += 256) {
        const int x = (q0 + tid) * 4;
        const bool live = x < W;
        const int xl = live ? x : 0;
        uint32_t g[4] = {0, 0, 0, 0};
        for (int seg = 0; seg < 2; ++seg) {
            const int ys = seg ? yc : ya, ye = seg ? yb : yc;
            if (ys >= ye) continue;
            uint32_t e[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0};     // B | R << 16 and G
            int nrows = 0;
            auto flush = [&]() {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (live && x + j < W) {
                        const int gx = ((x + j) * 8) / W;
                        unsigned long long* t4 = &tab[seg][copy][gx * 4];
                        atomicAdd(t4, (unsigned long long)(e[j] & 0xffffu));
                        atomicAdd(t4 + 1, (unsigned long long)o[j]);
                        atomicAdd(t4 + 2, (unsigned long long)(e[j] >> 16));
                        atomicAdd(t4 + 3, (unsigned long long)nrows);
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = o[j] = 0;
                nrows = 0;
            };
            // four rows are asked for while the four before them are added up
            uint32_t nxt[4][4];
            auto fetch = [&](int y) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int yy = y + u < ye ? y + u : ye - 1;
                    load_quad<ALIGNED>(im + (size_t)yy * W * 3, xl, W, y + u < ye ? 0xffffffffu : 0u, nxt[u]);
                }
            };
            fetch(ys);
            for (int y = ys; y < ye; y += 4) {
                uint32_t px[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) px[u][j] = nxt[u][j];
                if (y + 4 < ye) fetch(y + 4);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        e[j] += px[u][j] & 0xff00ffu;
                        o[j] += (px[u][j] >> 8) & 0xffu;
                        // (a masked pixel has gray 0)
                        g[j] += gray_px(px[u][j] & 0xffu, (px[u][j] >> 8) & 0xffu, px[u][j] >> 16);
                    }
                nrows += min(4, ye - y);
                if (nrows >= 252) flush();             // the 16-bit halves of e hold 257 rows
            }
            flush();
        }
        if (edge) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (live && x + j < wuse) {
                    const int bx = (x + j) / bs;
                    const int qx = (x + j - bx * bs) >= hb;
                    atomicAdd(&ehd_tab[bx * 2 + qx], g[j]);
                }
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int seg = tid >> 5, gx = (tid >> 2) & 7, f = tid & 3;
        unsigned long long v = 0;
#pragma unroll
        for (int c = 0; c < CE_COPIES; ++c) v += tab[seg][c][gx * 4 + f];
        if (v) atomicAdd(&sums[((size_t)b * 64 + (gy0 + seg) * 8 + gx) * 4 + f], v);
    }
    if (edge) {
        uint32_t* Q = qsum + ((size_t)b * nby + by) * nbx * 4;
        for (int i = tid; i < nbx * 2; i += 256) Q[(i >> 1) * 4 + qy * 2 + (i & 1)] = ehd_tab[i];
    }
}

__global__ void ehd_classify_kernel(const uint32_t* __restrict__ qsum, int B, int H, int W, int bs,
                                    int nbx, int nby, int threshold, int32_t* __restrict__ out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int nblk = nbx * nby;
    if (t >= B * nblk) return;
    int b = t / nblk, k = t - b * nblk;
    int by = k / nbx, bx = k - by * nbx;
    const uint32_t* q = qsum + (size_t)t * 4;
    long long s1 = q[0], s2 = q[1], s3 = q[2], s4 = q[3];
    long long hbq = (long long)(bs / 2) * (bs / 2);
    long long e[5];
    long long v = s1 - s2 + s3 - s4, h = s1 + s2 - s3 - s4, d45 = s1 - s4, d135 = s2 - s3,
              nd = s1 - s2 - s3 + s4;
    e[0] = v * v;
    e[1] = h * h;
    e[2] = 2 * d45 * d45;
    e[3] = 2 * d135 * d135;
    e[4] = 4 * nd * nd;
    long long best = e[0];
    int type = 0;
    for (int i = 1; i < 5; ++i)
        if (e[i] > best) {
            best = e[i];
            type = i;
        }
    int sub = (bx * bs * 4) / W + ((by * bs * 4) / H) * 4;
    int32_t* o = out + (size_t)b * 96;
    atomicAdd(&o[80 + sub], 1);
    long long thr = (long long)threshold * hbq;
    if (best >= thr * thr) atomicAdd(&o[sub * 5 + type], 1);
}

// ------------------------------------------------------------------------------ CSD
struct HmmdTab {
    int nsub;
    int thr[6];
    int nhue[5], nsum[5], off[5];
};

// a / d for 0 <= a < 2^24, 0 < d, a / d < 2^20: float reciprocal and one correction step
// (exact, a quarter of the instructions of an integer division)
__device__ __forceinline__ int small_quot(int a, int d) {
    int q = (int)((float)a * __frcp_rn((float)d));
    const int r = a - q * d;
    return q + (r >= d) - (r < 0);
}

// a / d with inv = 1 / (float)d handed in: for 0 <= a < 2^24, 0 < d < 2^24 (a quarter of the
// instructions of the integer division the compiler expands to)
__device__ __forceinline__ int quot_inv(int a, int d, float inv) {
    int q = (int)((float)a * inv);
    const int r = a - __mul24(q, d);
    return q + (r >= d) - (r < 0);
}

__device__ __forceinline__ int hmmd_cell(int b, int g, int r, const HmmdTab& T) {
    int mx = max(max(r, g), b), mn = min(min(r, g), b);
    int d = mx - mn, s2 = mx + mn;
    int hue = 0;
    if (d > 0) {
        int num;
        if (r == mx) num = 60 * (g - b);
        else if (g == mx) num = 120 * d + 60 * (b - r);
        else num = 240 * d + 60 * (r - g);
        if (num < 0) num += 360 * d;
        hue = small_quot(2 * num + d, 2 * d);
        if (hue >= 360) hue -= 360;
    }
    int sub = 0;
    for (int i = 1; i < T.nsub; ++i) sub += (d >= T.thr[i]);
    int lo = T.thr[sub];
    int ihue = hue * T.nhue[sub] / 360;
    int isum = small_quot((s2 - lo) * T.nsum[sub], 2 * (256 - lo));
    return T.off[sub] + ihue * T.nsum[sub] + isum;
}

// The same cell with its per-subspace constants and the reciprocals of the two divisors read from
// LDS tables (region_desc_kernel: the selects from the argument arrays and the two reciprocals were
// a third of this function's instructions).  rcp2d[d] = 1 / (2 d) with the subspace of d in its
// three lowest mantissa bits (hmmd_rcp_entry; quot_inv corrects its estimate by +-1, which 2^-20
// relative on a quotient below 720 cannot exceed -- the threshold loop over the argument arrays
// was two scalar loops with a taken branch per step); sub8[sub] = {lo, nhue, nsum, off,
// 2 (256 - lo), 1 / that as float bits, -, -}.
__device__ __forceinline__ float hmmd_rcp_entry(int d, const HmmdTab& T) {
    int sub = 0;
    for (int i = 1; i < T.nsub; ++i) sub += (d >= T.thr[i]);
    return __uint_as_float((__float_as_uint(1.0f / (float)(2 * max(d, 1))) & ~7u) | (uint32_t)sub);
}
__device__ __forceinline__ int hmmd_cell_tab(int b, int g, int r, const HmmdTab& T,
                                             const float* rcp2d, const int4* sub8) {
    const int mx = max(max(r, g), b), mn = min(min(r, g), b);
    const int d = mx - mn, s2 = mx + mn;
    int hue = 0;
    const float inv2d = rcp2d[d];
    if (d > 0) {
        int num;
        if (r == mx) num = 60 * (g - b);
        else if (g == mx) num = 120 * d + 60 * (b - r);
        else num = 240 * d + 60 * (r - g);
        if (num < 0) num += 360 * d;
        hue = quot_inv(2 * num + d, 2 * d, inv2d);
        if (hue >= 360) hue -= 360;
    }
    const int sub = (int)(__float_as_uint(inv2d) & 7u);
    const int4 c0 = sub8[sub * 2], c1 = sub8[sub * 2 + 1];
    const int ihue = hue * c0.y / 360;
    const int isum = quot_inv((s2 - c0.x) * c0.z, c1.x, __int_as_float(c1.y));
    return c0.w + ihue * c0.z + isum;
}

__global__ __launch_bounds__(256) void hmmd_kernel(const uint8_t* __restrict__ img,
                                                   uint8_t* __restrict__ out, long long npix,
                                                   HmmdTab T) {
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix;
         p += (long long)gridDim.x * 256) {
        const uint8_t* q = img + p * 3;
        out[p] = (uint8_t)hmmd_cell(q[0], q[1], q[2], T);
    }
}

constexpr int CS_W = 32, CS_H = 16, CS_PW = CS_W + 7, CS_PH = CS_H + 7;

// one thread per window origin on the sub-sampled grid, two rows of origins per thread; 32 x 16
// windows per workgroup (a patch of 39 x 23 cells: 1.75 cells quantised per window, 2.3 with 8 rows)
// labels != NULL: the colour structure of ONE region over its bounding box (origin x0, y0 in an
// image of width W; cells of other regions hold no colour, out[nbins] counts the windows that hold
// a cell of the region) -- the form imgan_region_descriptors_large launches with one image.
__global__ __launch_bounds__(256) void csd_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                  int K, int nwx, int nwy, int nbins,
                                                  int32_t* __restrict__ out, HmmdTab T,
                                                  const int32_t* __restrict__ labels, int region,
                                                  int x0, int y0) {
    __shared__ uint32_t runs[CS_PH * 8 * CS_W];     // [patch row][mask word][column]
    __shared__ uint32_t hist[256];
    __shared__ uint32_t held;                       // windows that hold a cell of the region
    __shared__ float rcp2d[256];
    __shared__ int4 sub8[5 * 2];
    const int b = blockIdx.z;
    const int u0 = blockIdx.x * CS_W, v0 = blockIdx.y * CS_H;
    const uint8_t* im = img + (size_t)b * H * W * 3;
    for (int i = threadIdx.x; i < 256; i += 256) hist[i] = 0;
    if (threadIdx.x == 0) held = 0;
    for (int i = threadIdx.x; i < CS_PH * 8 * CS_W / 4; i += 256)
        reinterpret_cast<uint4*>(runs)[i] = make_uint4(0, 0, 0, 0);
    rcp2d[threadIdx.x] = hmmd_rcp_entry((int)threadIdx.x, T);
    if ((int)threadIdx.x < T.nsub) {
        const int lo = T.thr[threadIdx.x], den = 2 * (256 - lo);
        sub8[threadIdx.x * 2] = make_int4(lo, T.nhue[threadIdx.x], T.nsum[threadIdx.x], T.off[threadIdx.x]);
        sub8[threadIdx.x * 2 + 1] = make_int4(den, __float_as_int(1.0f / (float)den), 0, 0);
    }
    __syncthreads();
    // The 8 x 8 structuring element is separable for "which colours occur": first the set of
    // colours (a 256-bit mask, 8 words) of every horizontal run of 8 cells, once per patch row
    // and column, then each window ORs the 8 runs below its origin.  A cell sets its bit in the
    // eight runs that contain it with LDS atomics (building a run in registers from its eight
    // cells cost 24 instructions per cell, and this kernel is bound by instruction issue).
    for (int i = threadIdx.x; i < CS_PH * CS_PW; i += 256) {
        int pv = i / CS_PW, pu = i - pv * CS_PW;
        int u = u0 + pu, v = v0 + pv;
        uint32_t c = 0;
        bool has = true;                            // a cell past the last window: colour 0, never read
        if (u < nwx + 7 && v < nwy + 7) {
            const size_t p = (size_t)(y0 + v * K) * W + (size_t)(x0 + u * K);
            const uint8_t* q = im + p * 3;
            c = (uint32_t)hmmd_cell_tab(q[0], q[1], q[2], T, rcp2d, sub8);
            if (labels) has = labels[p] == region;
        }
        const uint32_t bit = has ? 1u << (c & 31u) : 0u;
        uint32_t* row = runs + (pv * 8 + (int)(c >> 5)) * CS_W;
#pragma unroll
        for (int du = 0; du < 8; ++du)
            if ((unsigned)(pu - du) < (unsigned)CS_W) atomicOr(&row[pu - du], bit);
    }
    __syncthreads();
    const int tu = threadIdx.x & 31;
    const int lane = threadIdx.x & 63;
    // votes[w]: lane i holds the number of this wave's windows with colour 32 w + i (the counts are
    // scalars -- one ballot per colour that occurs in the wave -- and ride in a lane of a vector
    // register until the end; an LDS atomic per colour and trip cost twice the instructions)
    int votes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int tv = threadIdx.x >> 5; tv < CS_H; tv += 8) {
        const bool valid = (u0 + tu < nwx) && (v0 + tv < nwy);
        uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (valid) {
#pragma unroll
            for (int dv = 0; dv < 8; ++dv)
#pragma unroll
                for (int w = 0; w < 8; ++w) m[w] |= runs[((tv + dv) * 8 + w) * CS_W + tu];
        }
        if (labels) {
            const unsigned long long some = __ballot(((m[0] | m[1]) | (m[2] | m[3])) | ((m[4] | m[5]) | (m[6] | m[7])));
            if (lane == 0 && some) atomicAdd(&held, (uint32_t)__popcll(some));
        }
        // every colour present in a window counts once: count lanes per bit, one add per wave
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            if (__ballot(m[w] != 0u) == 0ull) continue;    // most words are empty
            uint32_t any = wave_or_scalar(m[w]);
            while (any) {
                const int bit = __ffs((int)any) - 1;
                any &= any - 1;
                const unsigned long long vote = __ballot((m[w] & (1u << bit)) != 0u);
                votes[w] += lane == bit ? (int)__popcll(vote) : 0;
            }
        }
    }
    if (lane < 32) {
#pragma unroll
        for (int w = 0; w < 8; ++w)
            if (votes[w]) atomicAdd(&hist[w * 32 + lane], (uint32_t)votes[w]);
    }
    __syncthreads();
    int32_t* o = out + (size_t)b * (nbins + 1);
    for (int i = threadIdx.x; i < nbins; i += 256)
        if (hist[i]) atomicAdd(&o[i], (int32_t)hist[i]);
    if (labels) {
        if (threadIdx.x == 0 && held) atomicAdd(&o[nbins], (int32_t)held);
    } else if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        o[nbins] = nwx * nwy;
    }
}

// ---- regions beyond the on-chip capacity of region_desc_kernel (merged regions of the class API:
// a box of up to the whole frame), one region per call, streaming with global atomics.  Colour
// layout sums and the edge histogram's quadrant sums in one pass over the box: a thread owns a
// column of the box and walks down LG_ROWS rows, so its grid column, edge block column and
// quadrant half are fixed and its sums stay in registers while the grid row / block row / quadrant
// half do not change.
constexpr int LG_ROWS = 32;

// lg: u64 [64][4] colour layout sums, then u32 [nblk][5] = four quadrant grey sums + member count
__global__ __launch_bounds__(256) void region_large_sums_kernel(
    const uint8_t* __restrict__ im, const int32_t* __restrict__ lab, int W, int region, int x0, int y0,
    int bw, int bh, int bs, int nbx, int nby, unsigned long long* __restrict__ cld, uint32_t* __restrict__ q5) {
    const int tid = threadIdx.x;
    const int xx = blockIdx.x * 256 + tid, ya = blockIdx.y * LG_ROWS, yb = min(ya + LG_ROWS, bh);
    if (xx < bw) {
        const int gx = (int)(((long long)xx * 8) / bw);
        const int hb = bs >> 1;
        const bool inblk = xx < nbx * bs;
        const int bx = inblk ? xx / bs : 0;
        const int qx = inblk && (xx - bx * bs) >= hb ? 1 : 0;
        uint32_t sb = 0, sg = 0, sr = 0, sn = 0;        // colour layout: this column, current grid row
        int gy_cur = (int)(((long long)ya * 8) / bh);
        uint32_t eg = 0, en = 0;                        // edge histogram: current (block row, half)
        int by_cur = -1, qy_cur = 0;
        auto flush_cld = [&]() {
            if (sn) {
                unsigned long long* d = cld + (size_t)(gy_cur * 8 + gx) * 4;
                atomicAdd(&d[0], (unsigned long long)sb);
                atomicAdd(&d[1], (unsigned long long)sg);
                atomicAdd(&d[2], (unsigned long long)sr);
                atomicAdd(&d[3], (unsigned long long)sn);
            }
            sb = sg = sr = sn = 0;
        };
        auto flush_ehd = [&]() {
            if (by_cur >= 0) {
                uint32_t* d = q5 + (size_t)(by_cur * nbx + bx) * 5;
                atomicAdd(&d[qy_cur * 2 + qx], eg);
                if (en) atomicAdd(&d[4], en);
            }
            eg = en = 0;
        };
        for (int yy = ya; yy < yb; ++yy) {
            const size_t p = (size_t)(y0 + yy) * W + (size_t)(x0 + xx);
            const uint8_t* q = im + p * 3;
            const uint32_t cb = q[0], cg = q[1], cr = q[2];
            const bool mem = lab[p] == region;
            const int gy = (int)(((long long)yy * 8) / bh);
            if (gy != gy_cur) {
                flush_cld();
                gy_cur = gy;
            }
            if (mem) sb += cb, sg += cg, sr += cr, sn += 1;
            if (inblk && yy < nby * bs) {
                const int by = yy / bs, qy = (yy - by * bs) >= hb ? 1 : 0;
                if (by != by_cur || qy != qy_cur) {
                    flush_ehd();
                    by_cur = by, qy_cur = qy;
                }
                eg += gray_px(cb, cg, cr);              // all pixels of the block; it votes only if
                en += mem ? 1u : 0u;                    // the region holds every one of them
            }
        }
        flush_cld();
        flush_ehd();
    }
}

// blocks that the region fills vote (oracle/mpeg7_std.py region_edge_histogram_counts)
__global__ void region_large_ehd_kernel(const uint32_t* __restrict__ q5, int bw, int bh, int bs, int nbx,
                                        int nby, int threshold, int32_t* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nbx * nby) return;
    const uint32_t* q = q5 + (size_t)k * 5;
    if (q[4] != (uint32_t)(bs * bs)) return;
    const int by = k / nbx, bx = k - by * nbx;
    const long long s1 = q[0], s2 = q[1], s3 = q[2], s4 = q[3];
    const long long hbq = (long long)(bs / 2) * (bs / 2);
    const long long v = s1 - s2 + s3 - s4, h = s1 + s2 - s3 - s4, d45 = s1 - s4, d135 = s2 - s3,
                    nd = s1 - s2 - s3 + s4;
    const long long e[5] = {v * v, h * h, 2 * d45 * d45, 2 * d135 * d135, 4 * nd * nd};
    long long best = e[0];
    int type = 0;
    for (int i = 1; i < 5; ++i)
        if (e[i] > best) best = e[i], type = i;
    const int sub = (int)(((long long)bx * bs * 4) / bw) + (int)(((long long)by * bs * 4) / bh) * 4;
    atomicAdd(&out[80 + sub], 1);
    const long long thr = (long long)threshold * hbq;
    if (best >= thr * thr) atomicAdd(&out[sub * 5 + type], 1);
}

// ------------------------------------------------------------------------------ per region
// CLD / EHD / CSD of every region of a label map in one pass over image + labels (7 B/px): one
// workgroup per region stages the cells of the region's bounding box in LDS (grey level, HMMD
// colour cell, membership bit), adds up the 8 x 8 grid sums on the way, then walks the edge blocks
// and the colour-structure windows of the box on chip.  Definitions: oracle/mpeg7_std.py
// "per region".  A box of more than RD_CAP cells is left to the caller (counted in too_big).
constexpr int RD_THREADS = 512, RD_CAP = 24576, RD_WORDS = RD_CAP / 64;
constexpr int RD_POOL = 4608;                              // 64-bit words: 36 KB >= RD_CAP grey bytes
constexpr int RD_TABS = 8, RD_TAB_PITCH = 64 * 4 + 1;
// edge blocks of a box that fits are 2 or 4 pixels wide, so their sums and squared responses stay
// far inside 32 bits
static_assert(RD_CAP / 1100 < 36, "edge blocks wider than 4 pixels need 64-bit responses");
typedef uint32_t __attribute__((aligned(1))) u32_any;     // one global_load_dword at any address
static_assert(RD_CAP < 32768, "the colour structure's sub-sampling factor must stay 1 on chip");

__device__ __forceinline__ int ehd_block_size_dev(int H, int W) {
    const double temp = sqrt((double)W * H / 1100.0);
    const int bs = (int)(temp / 2) * 2;
    return bs ? bs : 2;
}

// 64 bits of a bitmap over the box's cells, starting `shift` bits after the start of word w
__device__ __forceinline__ unsigned long long bits_at(const unsigned long long* bm, int w, int shift,
                                                      int nwords) {
    const int ws = w + (shift >> 6), bs = shift & 63;
    const unsigned long long lo = ws < nwords ? bm[ws] : 0ull;
    const unsigned long long hi = ws + 1 < nwords ? bm[ws + 1] : 0ull;
    return bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
}

#ifdef IMGAN_RD_TIMING
// phase clocks of thread 0, summed over blocks (tools/rd_timing.py; never built into the product)
__device__ unsigned long long rd_timing[16];
#define RD_MARK(i)                                                          \
    if (tid == 0) {                                                         \
        unsigned long long now_ = wall_clock64();                           \
        atomicAdd(&rd_timing[i], now_ - t_last_);                           \
        t_last_ = now_;                                                     \
    }
#else
#define RD_MARK(i)
#endif

// LISTED: the region's pixels come from its run of the pixel lists that imgan_dcd_segments_listed
// leaves behind (positions y << 16 | x, dcd_lists.h) instead of from a scan of its bounding box
// against the label map: a superpixel's box holds 2.2 cells per pixel of its own, and every staged
// cell costs about as much as a member (address arithmetic, two loads, the ballots).
template <bool LISTED>
__global__ __launch_bounds__(RD_THREADS) void region_desc_kernel(
    const uint8_t* __restrict__ img, const int32_t* __restrict__ labels, const int32_t* __restrict__ bbox,
    int H, int W, int nseg, int nbins, HmmdTab T, unsigned long long* __restrict__ cld_sums,
    int32_t* __restrict__ ehd_out, int32_t* __restrict__ csd_out, int32_t* __restrict__ too_big,
    imgan_dcd::SegLists L) {
    // pool: first the grey levels of the box's cells (edge histogram), then, for the colour
    // structure, bitmaps over the cells: per colour of a pass its cells, then, in place, their
    // dilation along x.  member / fits: the region's cells, and the cells where an 8 x 8 window fits.
    __shared__ unsigned long long pool[RD_POOL];
    __shared__ uint8_t col[RD_CAP];
    __shared__ unsigned long long member[RD_WORDS], fits[RD_WORDS];
    // grid sums in RD_TABS copies (a cell adds to copy = lane % RD_TABS; the copies start one bank
    // apart): the 64 cells of a wave fall into a handful of blocks, and lanes that add to the
    // same word take turns
    __shared__ uint32_t tab[RD_TABS * RD_TAB_PITCH], hist[256], ehd[96], present[8], cnts[264];
    __shared__ uint8_t slot_of[256], colour_of[257];
    __shared__ float rcp2d[256];
    __shared__ int4 sub8[5 * 2];
    // edge histogram: one 64-bit word per block, {grey sums of its four quadrants, 10 bits each,
    // number of the region's pixels in it << 40} (blocks of 2 x 2 or 4 x 4 cells at this capacity)
    unsigned long long* eacc = pool;
    const int s = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const size_t gs = (size_t)b * nseg + s;
    const int* bb = bbox + gs * 4;
    const int x0 = max(bb[0], 0), y0 = max(bb[1], 0), x1 = min(bb[2], W - 1), y1 = min(bb[3], H - 1);
    unsigned long long* csum = cld_sums + gs * 64 * 4;
    int32_t* eo = ehd_out + gs * 96;
    int32_t* co = csd_out + gs * (size_t)(nbins + 1);
    const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
    const bool empty = x1 < x0 || y1 < y0;
    const long long area_ll = empty ? 0 : (long long)bw * bh;
    if (empty || area_ll > RD_CAP) {                     // nothing, or too large for the chip
        for (int i = tid; i < 64 * 4; i += RD_THREADS) csum[i] = 0ull;
        for (int i = tid; i < 96; i += RD_THREADS) eo[i] = 0;
        for (int i = tid; i <= nbins; i += RD_THREADS) co[i] = 0;
        if (!empty && tid == 0) atomicAdd(too_big, 1);
        return;
    }
    const int area = (int)area_ll, nwords = (area + 63) >> 6;
#ifdef IMGAN_RD_TIMING
    unsigned long long t_last_ = wall_clock64();
#endif
    const int ebs = ehd_block_size_dev(bh, bw), elg = ebs >= 4 ? 2 : 1;   // 2 or 4 (RD_CAP / 1100 < 36)
    const int nbx = bw >> elg, nby = bh >> elg;
    for (int i = tid; i < nbx * nby; i += RD_THREADS) eacc[i] = 0ull;
    for (int i = tid; i < RD_TABS * RD_TAB_PITCH; i += RD_THREADS) tab[i] = 0;
    for (int i = tid; i < 256; i += RD_THREADS) hist[i] = 0;
    for (int i = tid; i < 96; i += RD_THREADS) ehd[i] = 0;
    if (tid < 8) present[tid] = 0;
    if (tid < 256) rcp2d[tid] = hmmd_rcp_entry(tid, T);
    if (tid < T.nsub) {
        const int lo = T.thr[tid], den = 2 * (256 - lo);
        sub8[tid * 2] = make_int4(lo, T.nhue[tid], T.nsum[tid], T.off[tid]);
        sub8[tid * 2 + 1] = make_int4(den, __float_as_int(1.0f / (float)den), 0, 0);
    }
    __syncthreads();

    // ---- stage the box; 8 x 8 grid sums of the region's pixels on the way
    const uint8_t* im = img + (size_t)b * H * W * 3;
    const int32_t* lab = labels + (size_t)b * H * W;
    const float inv_bw = 1.0f / (float)bw, inv_bh = 1.0f / (float)bh;
    auto edge_add = [&](int x, int y, uint32_t g) {        // a pixel of the region at cell (x, y)
        const int bx = x >> elg, by = y >> elg;
        if (bx < nbx && by < nby) {
            const int q = (((y >> (elg - 1)) & 1) << 1) | ((x >> (elg - 1)) & 1);
            atomicAdd(&eacc[__mul24(by, nbx) + bx], ((unsigned long long)g << (10 * q)) + (1ull << 40));
        }
    };
    // Four cells per thread and step, every load issued before the first use (a step is bound by
    // the memory latency of its loads, not by their number); addresses are clamped into the box
    // instead of branched around, and the three colour bytes come with one unaligned dword load
    // (the very last pixel of the batch is read one byte early so that the load stays in bounds).
    constexpr int GU = 4;
    const uint32_t last_p = b == (int)gridDim.y - 1 ? (uint32_t)H * (uint32_t)W - 1u : 0xffffffffu;
    if (LISTED) {
        // The run's positions and colours are asked for a batch ahead: the first batch before the
        // bitmaps below are set up (two dependent loads per pixel and two workgroups per CU leave
        // little else to hide their latency behind), each later one while the batch before it is
        // worked through.
        constexpr int LU = 6;
        const uint32_t run_at = L.offsets[gs];
        const int run_n = (int)min(min(L.ends[gs] - run_at, L.cursor[gs]), (uint32_t)RD_CAP);
        // one scalar base for the run (kept inside the list even for a region without entries),
        // 32-bit indices below RD_CAP
        const uint32_t last_entry = gridDim.y * (uint32_t)H * (uint32_t)W - 1u;   // of the whole list
        const uint32_t* __restrict__ run = L.sxy + min(run_at, last_entry);
        uint32_t xy[LU], raw[LU], sh[LU];
        auto fetch = [&](int i0) {
#pragma unroll
            for (int u = 0; u < LU; ++u)
                xy[u] = run[max(min(i0 + u * RD_THREADS + tid, run_n - 1), 0)];
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                // clamped into the image: the lists are the caller's scratch memory
                const int xa = min((int)(xy[u] & 0xffffu), W - 1), ya = min((int)(xy[u] >> 16), H - 1);
                const uint32_t p = (uint32_t)__mul24(ya, W) + (uint32_t)xa;
                sh[u] = p == last_p ? 1u : 0u;
                raw[u] = *reinterpret_cast<const u32_any*>(im + (p * 3u - sh[u]));
            }
        };
        fetch(0);
        // the cells where a window fits follow from the box alone; the members are set bit by bit
        for (int i = tid; i < nwords; i += RD_THREADS) member[i] = 0ull;
        for (int i0 = 0; i0 < area; i0 += RD_THREADS) {
            const int i = i0 + tid;
            const int ic = i < area ? i : area - 1;
            const int y = quot_inv(ic, bw, inv_bw);
            const int x = ic - __mul24(y, bw);
            const unsigned long long fbits = __ballot(i < area && x + 8 <= bw && y + 8 <= bh);
            if (lane == 0 && i < area) fits[i >> 6] = fbits;
        }
        __syncthreads();
        uint32_t* m32 = reinterpret_cast<uint32_t*>(member);
        for (int i0 = 0; i0 < run_n; i0 += RD_THREADS * LU) {
            uint32_t cxy[LU], cv[LU];
#pragma unroll
            for (int u = 0; u < LU; ++u) cxy[u] = xy[u], cv[u] = raw[u] >> (8u * sh[u]);
            if (i0 + RD_THREADS * LU < run_n) fetch(i0 + RD_THREADS * LU);
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                const int i = i0 + u * RD_THREADS + tid;
                const int x = (int)(cxy[u] & 0xffffu) - x0, y = (int)(cxy[u] >> 16) - y0;
                // (a position outside the box is not the region's as far as this pass goes: the
                // scan of the box would not have met it either)
                if (i < run_n && (unsigned)x < (unsigned)bw && (unsigned)y < (unsigned)bh) {
                    const int cell = __mul24(y, bw) + x;
                    const uint32_t v = cv[u];
                    const uint32_t cb = v & 255u, cg = (v >> 8) & 255u, cr = (v >> 16) & 255u;
                    edge_add(x, y, gray_px(cb, cg, cr));
                    const uint32_t c = (uint32_t)hmmd_cell_tab((int)cb, (int)cg, (int)cr, T, rcp2d, sub8);
                    col[cell] = (uint8_t)c;
                    if (!((present[c >> 5] >> (c & 31u)) & 1u)) atomicOr(&present[c >> 5], 1u << (c & 31u));
                    const int blk = quot_inv(y * 8, bh, inv_bh) * 8 + quot_inv(x * 8, bw, inv_bw);
                    uint32_t* t4 = tab + (lane & (RD_TABS - 1)) * RD_TAB_PITCH + blk * 4;
                    atomicAdd(t4, cb);
                    atomicAdd(t4 + 1, cg);
                    atomicAdd(t4 + 2, cr);
                    atomicAdd(t4 + 3, 1u);
                    atomicOr(&m32[cell >> 5], 1u << (cell & 31));
                }
            }
        }
    } else
    for (int i0 = 0; i0 < area; i0 += RD_THREADS * GU) {
        int xx[GU], yy[GU], lbl[GU];
        uint32_t raw[GU], sh[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = i0 + u * RD_THREADS + tid;
            const int ic = i < area ? i : area - 1;
            const int y = quot_inv(ic, bw, inv_bw);      // cells < 2^15
            const int x = ic - __mul24(y, bw);
            xx[u] = x, yy[u] = y;
            // 32-bit offsets inside the image (H * W * 3 < 2^32, checked by the host)
            const uint32_t p = (uint32_t)__mul24(y0 + y, W) + (uint32_t)(x0 + x);
            lbl[u] = lab[p];
            sh[u] = p == last_p ? 1u : 0u;
            raw[u] = *reinterpret_cast<const u32_any*>(im + (p * 3u - sh[u]));
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = i0 + u * RD_THREADS + tid;
            const bool live = i < area;
            const bool mem = live && lbl[u] == s;
            const bool fit = live && xx[u] + 8 <= bw && yy[u] + 8 <= bh;
            const uint32_t v = raw[u] >> (8u * sh[u]);
            const uint32_t cb = v & 255u, cg = (v >> 8) & 255u, cr = (v >> 16) & 255u;
            int blk = -1;
            if (mem) {
                edge_add(xx[u], yy[u], gray_px(cb, cg, cr));
                const uint32_t c = (uint32_t)hmmd_cell_tab((int)cb, (int)cg, (int)cr, T, rcp2d, sub8);
                col[i] = (uint8_t)c;
                if (!((present[c >> 5] >> (c & 31u)) & 1u)) atomicOr(&present[c >> 5], 1u << (c & 31u));
                blk = quot_inv(yy[u] * 8, bh, inv_bh) * 8 + quot_inv(xx[u] * 8, bw, inv_bw);
            }
            // (one wave-wide sum per block and field instead of these atomics, 64 cells hitting a
            // handful of LDS words, was measured slower: 10.1 against 7.2 ms on the mosaic)
            if (blk >= 0) {
                uint32_t* t4 = tab + (lane & (RD_TABS - 1)) * RD_TAB_PITCH + blk * 4;
                atomicAdd(t4, cb);
                atomicAdd(t4 + 1, cg);
                atomicAdd(t4 + 2, cr);
                atomicAdd(t4 + 3, 1u);
            }
            const unsigned long long mbits = __ballot(mem), fbits = __ballot(fit);   // 64 consecutive cells
            if (lane == 0 && live) member[i >> 6] = mbits, fits[i >> 6] = fbits;
        }
    }
    __syncthreads();
    RD_MARK(0)
    for (int i = tid; i < 64 * 4; i += RD_THREADS) {
        uint32_t sum = 0;
#pragma unroll
        for (int c = 0; c < RD_TABS; ++c) sum += tab[c * RD_TAB_PITCH + i];
        csum[i] = sum;
    }
    auto is_member = [&](int cell) { return (member[cell >> 6] >> (cell & 63)) & 1ull; };

    // ---- edge histogram: blocks tile the box from its corner; a block votes if the region holds
    // all of its pixels (its quadrant sums were added up pixel by pixel above)
    {
        const int hb = ebs >> 1;
        const float inv_nbx = 1.0f / (float)max(nbx, 1);
        for (int k = tid; k < nbx * nby; k += RD_THREADS) {
            const unsigned long long w = eacc[k];
            if ((int)(w >> 40) != ebs * ebs) continue;
            const int by = quot_inv(k, nbx, inv_nbx), bx = k - by * nbx;
            const int s1 = (int)(w & 1023u), s2 = (int)((w >> 10) & 1023u), s3 = (int)((w >> 20) & 1023u),
                      s4 = (int)((w >> 30) & 1023u);
            const int v = s1 - s2 + s3 - s4, h = s1 + s2 - s3 - s4, d45 = s1 - s4, d135 = s2 - s3,
                      nd = s1 - s2 - s3 + s4;
            const int e[5] = {v * v, h * h, 2 * d45 * d45, 2 * d135 * d135, 4 * nd * nd};
            int best = e[0];
            int type = 0;
#pragma unroll
            for (int t = 1; t < 5; ++t)
                if (e[t] > best) best = e[t], type = t;
            const int sub = quot_inv(bx * ebs * 4, bw, inv_bw) + quot_inv(by * ebs * 4, bh, inv_bh) * 4;
            atomicAdd(&ehd[80 + sub], 1u);
            const int thr = 11 * hb * hb;
            if (best >= thr * thr) atomicAdd(&ehd[sub * 5 + type], 1u);
        }
    }

    // ---- colour structure.  A window (origin u, v) holds colour c iff one of the region's cells
    // in [u, u+8) x [v, v+8) has it: the windows of c are the 8 x 8 dilation of c's cells towards
    // the origin.  On bitmaps over the box's cells that is seven shifts by one bit (x) and seven
    // by one row pitch (y); bits that a shift carries across a row end land on origins where no
    // window fits, which the `fits` bitmap removes before the count.  Slot 0 is the region itself
    // (the number of windows that hold a region pixel), slots 1.. its colours in ascending order;
    // a pass takes as many slots as the pool holds bitmaps for (all of them for a typical
    // superpixel).
    uint32_t nwin = 0;
    RD_MARK(1)
    if (bw >= 8 && bh >= 8) {
        if (tid < 256) {
            const uint32_t c = (uint32_t)tid;
            uint32_t below = 0;
            for (uint32_t w = 0; w < (c >> 5); ++w) below += (uint32_t)__popc(present[w]);
            below += (uint32_t)__popc(present[c >> 5] & ((1u << (c & 31u)) - 1u));
            const bool here = (present[c >> 5] >> (c & 31u)) & 1u;
            slot_of[c] = here ? (uint8_t)(1u + below) : (uint8_t)0;   // <= 256 colours: slot 256 wraps
            if (here) colour_of[1 + below] = (uint8_t)c;
        }
        int nslots = 1;
        for (int w = 0; w < 8; ++w) nslots += __popc(present[w]);
        const int G = min(nslots, RD_POOL / nwords);           // >= 12
        const float inv_nw = 1.0f / (float)nwords;
        __syncthreads();                                      // the edge histogram is done with the pool
        RD_MARK(2)
        for (int g0 = 0; g0 < nslots; g0 += G) {
            const int ng = min(G, nslots - g0);
            unsigned long long* Rb = pool;
            for (int k = tid; k < ng * nwords; k += RD_THREADS) Rb[k] = 0ull;
            for (int k = tid; k < ng; k += RD_THREADS) cnts[k] = 0;
            __syncthreads();
            RD_MARK(3)
            // a wave owns the 64-bit word of its 64 consecutive cells: one ballot and one plain
            // store per slot that occurs among them
            // (four words per wave and step: the two dependent table reads of a step overlap)
            constexpr int BU = 4;
            if (LISTED) {
                // one bit per pixel of the region's run: its cell in the bitmap of its colour's slot
                // (the positions are read again: they are still in the cache)
                const uint32_t run_at = L.offsets[gs];
                const int run_n = (int)min(min(L.ends[gs] - run_at, L.cursor[gs]), (uint32_t)RD_CAP);
                const uint32_t* __restrict__ run2 =
                    L.sxy + min(run_at, gridDim.y * (uint32_t)H * (uint32_t)W - 1u);
                uint32_t* R32 = reinterpret_cast<uint32_t*>(Rb);
                if (g0 == 0)
                    for (int k = tid; k < nwords; k += RD_THREADS) Rb[k] = member[k];
                for (int i0 = 0; i0 < run_n; i0 += RD_THREADS * BU) {
                    uint32_t xy[BU];
#pragma unroll
                    for (int u = 0; u < BU; ++u)
                        xy[u] = run2[min(i0 + u * RD_THREADS + tid, run_n - 1)];
                    int cell[BU], rr[BU];
#pragma unroll
                    for (int u = 0; u < BU; ++u) {
                        const int x = (int)(xy[u] & 0xffffu) - x0, y = (int)(xy[u] >> 16) - y0;
                        const bool ok = i0 + u * RD_THREADS + tid < run_n && (unsigned)x < (unsigned)bw &&
                                        (unsigned)y < (unsigned)bh;
                        cell[u] = ok ? __mul24(y, bw) + x : -1;
                    }
#pragma unroll
                    for (int u = 0; u < BU; ++u) rr[u] = slot_of[col[max(cell[u], 0)]];
#pragma unroll
                    for (int u = 0; u < BU; ++u) {
                        int sl = rr[u];
                        if (sl == 0) sl = 256;               // slot 256 is stored as 0 in the byte table
                        const int r = sl - g0;
                        if (cell[u] >= 0 && r >= 0 && r < ng)
                            atomicOr(&R32[(size_t)r * nwords * 2 + (cell[u] >> 5)], 1u << (cell[u] & 31));
                    }
                }
            } else
            for (int i0 = 0; i0 < area; i0 += RD_THREADS * BU) {
                int rr[BU];
                bool mm[BU];
                uint8_t cc[BU];
#pragma unroll
                for (int u = 0; u < BU; ++u) {
                    const int i = i0 + u * RD_THREADS + tid;
                    mm[u] = i < area && is_member(i);
                    cc[u] = col[min(i, area - 1)];           // not a member: any byte
                }
#pragma unroll
                for (int u = 0; u < BU; ++u) rr[u] = slot_of[cc[u]];
#pragma unroll
                for (int u = 0; u < BU; ++u) {
                    const int i = i0 + u * RD_THREADS + tid;
                    int r = -1;
                    if (mm[u]) {
                        int sl = rr[u];
                        if (sl == 0) sl = 256;               // slot 256 is stored as 0 in the byte table
                        r = sl - g0;
                        if (r >= ng) r = -1;
                    }
                    if (g0 == 0) {
                        const unsigned long long all = __ballot(mm[u]);
                        if (lane == 0 && i < area) Rb[i >> 6] = all;
                    }
                    unsigned long long todo = __ballot(r >= 0);
                    while (todo) {
                        const int rsel = __builtin_amdgcn_readlane(r, __ffsll((long long)todo) - 1);
                        const unsigned long long who = __ballot(r == rsel);
                        if (lane == 0) Rb[(size_t)rsel * nwords + (i >> 6)] = who;
                        todo &= ~who;
                    }
                }
            }
            __syncthreads();
            RD_MARK(4)
            // dilation along x in place (a thread holds its words while the block reads): the pool
            // then takes twice the slots per pass
            {
                constexpr int DW = (RD_POOL + RD_THREADS - 1) / RD_THREADS;
                unsigned long long dil[DW];
#pragma unroll
                for (int q = 0; q < DW; ++q) {
                    const int k = tid + q * RD_THREADS;
                    dil[q] = 0ull;
                    if (k < ng * nwords) {
                        const int r = quot_inv(k, nwords, inv_nw), w = k - r * nwords;
                        const unsigned long long* Rr = Rb + (size_t)r * nwords;
                        const unsigned long long p = Rr[w], n = w + 1 < nwords ? Rr[w + 1] : 0ull;
                        unsigned long long d = p;
#pragma unroll
                        for (int dx = 1; dx < 8; ++dx) d |= (p >> dx) | (n << (64 - dx));
                        dil[q] = d;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < DW; ++q) {
                    const int k = tid + q * RD_THREADS;
                    if (k < ng * nwords) Rb[k] = dil[q];
                }
            }
            __syncthreads();
            RD_MARK(5)
            for (int k = tid; k < ng * nwords; k += RD_THREADS) {
                const int r = quot_inv(k, nwords, inv_nw), w = k - r * nwords;
                const unsigned long long* Dr = Rb + (size_t)r * nwords;
                unsigned long long v = Dr[w];
#pragma unroll
                for (int dy = 1; dy < 8; ++dy) v |= bits_at(Dr, w, dy * bw, nwords);
                const uint32_t c = (uint32_t)__popcll(v & fits[w]);
                if (c) atomicAdd(&cnts[r], c);
            }
            __syncthreads();
            RD_MARK(6)
            for (int r = tid; r < ng; r += RD_THREADS) {
                const int sl = g0 + r;
                if (sl == 0) nwin = cnts[r];                   // thread 0 of the first pass
                else hist[colour_of[sl]] = cnts[r];
            }
            __syncthreads();                                   // cnts and the pool are reused
            RD_MARK(7)
#ifdef IMGAN_RD_TIMING
            if (tid == 0) atomicAdd(&rd_timing[15], 1ull);
#endif
        }
    }
    __syncthreads();
    for (int i = tid; i < 96; i += RD_THREADS) eo[i] = (int32_t)ehd[i];
    for (int i = tid; i < nbins; i += RD_THREADS) co[i] = (int32_t)hist[i];
    if (tid == 0) co[nbins] = (int32_t)nwin;
    RD_MARK(8)
}

int make_hmmd(int nbins, HmmdTab* T) {
    static const int thr5[6] = {0, 6, 20, 60, 110, 256};
    static const int thr4[5] = {0, 6, 60, 110, 256};
    const int *nh, *ns;
    static const int nh256[5] = {1, 4, 16, 16, 16}, ns256[5] = {32, 8, 4, 4, 4};
    static const int nh128[5] = {1, 4, 8, 8, 8}, ns128[5] = {16, 4, 4, 4, 4};
    static const int nh64[5] = {1, 4, 4, 8, 8}, ns64[5] = {8, 4, 4, 2, 1};
    static const int nh32[4] = {1, 4, 4, 4}, ns32[4] = {8, 4, 1, 1};
    *T = HmmdTab{};
    switch (nbins) {
        case 256: nh = nh256; ns = ns256; T->nsub = 5; break;
        case 128: nh = nh128; ns = ns128; T->nsub = 5; break;
        case 64: nh = nh64; ns = ns64; T->nsub = 5; break;
        case 32: nh = nh32; ns = ns32; T->nsub = 4; break;
        default: return imgan::fail(imgan::E_INVALID, "HMMD quantisation supports 32, 64, 128 or 256 cells");
    }
    int off = 0;
    for (int i = 0; i < T->nsub; ++i) {
        T->thr[i] = T->nsub == 5 ? thr5[i] : thr4[i];
        T->nhue[i] = nh[i];
        T->nsum[i] = ns[i];
        T->off[i] = off;
        off += nh[i] * ns[i];
    }
    T->thr[T->nsub] = 256;
    return 0;
}

}  // namespace

extern "C" size_t imgan_cld_scratch_bytes(int B) { return B < 1 ? 0 : (size_t)B * 64 * 4 * 8; }

// the two launches of the colour layout; `sums` is device memory of imgan_cld_scratch_bytes(B)
static int cld_run(const uint8_t* bgr, int B, int H, int W, int32_t* cld_out,
                   unsigned long long* sums, hipStream_t st, const char* what) {
    IMGAN_HIP(hipMemsetAsync(sums, 0, imgan_cld_scratch_bytes(B), st));
    dim3 grid(1, 8 * CLD_PARTS, (unsigned)B);
    if (W % 4 == 0 && ((uintptr_t)bgr & 3) == 0)
        hipLaunchKernelGGL(cld_sums_kernel<true>, grid, dim3(256), 0, st, bgr, H, W, sums);
    else
        hipLaunchKernelGGL(cld_sums_kernel<false>, grid, dim3(256), 0, st, bgr, H, W, sums);
    DctTable T;
    for (int i = 0; i < 8; ++i) {
        double s = i == 0 ? sqrt(0.125) : 0.5;
        for (int j = 0; j < 8; ++j) T.c[i * 8 + j] = s * cos((M_PI / 8.0) * i * (j + 0.5));
    }
    hipLaunchKernelGGL(cld_finalise_kernel, dim3(B * 3), dim3(64), 0, st,
                       (const unsigned long long*)sums, B, cld_out, T, 0);
    return imgan::check_launch(what);
}

static int cld_check(const uint8_t* bgr, int B, int H, int W, const int32_t* cld_out) {
    IMGAN_REQUIRE(bgr && cld_out, "imgan_cld: null pointer");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && B <= 65535, "imgan_cld: bad shape");
    IMGAN_REQUIRE((long long)H * 8 < 0x7fffffffLL && (long long)W * 8 < 0x7fffffffLL, "imgan_cld: too large");
    return 0;
}

extern "C" int imgan_cld(const uint8_t* bgr, int B, int H, int W, int32_t* cld_out, void* scratch,
                         imgan_stream_t stream) {
    if (int rc = cld_check(bgr, B, H, W, cld_out)) return rc;
    IMGAN_REQUIRE(scratch, "imgan_cld: null scratch");
    return cld_run(bgr, B, H, W, cld_out, (unsigned long long*)scratch, (hipStream_t)stream,
                   "imgan_cld");
}

// ---- imgan_cld_static: the colour layout for callers that cannot allocate -------------------
// The block sums of imgan_cld (u64 [B][64][4]) live in a file-scope device array instead of a
// caller's workspace.  `scratch` below is the HOST-side stand-in of that array: it is never
// handed to a kernel or a memset; the device address comes from hipGetSymbolAddress, once per
// device (every GPU holds its own copy of the array at its own address).
constexpr int CLD_STATIC_MAX_B = 1024;                       // 1024 images x 2 KiB = 2 MiB per GPU
static __device__ __align__(256) unsigned long long scratch[(size_t)CLD_STATIC_MAX_B * 64 * 4];

constexpr int CLD_MAX_DEVICES = 64;
static std::atomic<unsigned long long*> g_cld_static_ptr[CLD_MAX_DEVICES];

// The device address of the array on the current device: resolved once per device, kept.
static int cld_static_address(unsigned long long** out, int* dev_out) {
    int dev = -1;
    IMGAN_HIP(hipGetDevice(&dev));
    IMGAN_REQUIRE(dev >= 0 && dev < CLD_MAX_DEVICES, "imgan_cld_static: device ordinal out of range");
    unsigned long long* ptr = g_cld_static_ptr[dev].load(std::memory_order_acquire);
    if (!ptr) {                                              // first call on this device
        void* p = nullptr;
        IMGAN_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        IMGAN_REQUIRE(p != nullptr, "imgan_cld_static: no device address for the static scratch");
        ptr = (unsigned long long*)p;
        g_cld_static_ptr[dev].store(ptr, std::memory_order_release);   // same value from every thread
    }
    *out = ptr;
    if (dev_out) *dev_out = dev;
    return 0;
}

// Set-up of imgan_cld_static for the current device, for callers that will call it where the
// look-up of the array's address is not allowed (hipGetSymbolAddress may load the code object: not
// a stream operation, not capturable): call this once per device beforehand.
extern "C" int imgan_cld_static_prepare(void) {
    unsigned long long* ptr = nullptr;
    return cld_static_address(&ptr, nullptr);
}

extern "C" int imgan_cld_static(const uint8_t* bgr, int B, int H, int W, int32_t* cld_out,
                                imgan_stream_t stream) {
    if (int rc = cld_check(bgr, B, H, W, cld_out)) return rc;
    if (B > CLD_STATIC_MAX_B)
        return imgan::fail(imgan::E_RANGE, "imgan_cld_static: at most " +
                                               std::to_string(CLD_STATIC_MAX_B) +
                                               " images per call (static scratch); use imgan_cld");
    unsigned long long* ptr = nullptr;
    int dev = -1;
    if (int rc = cld_static_address(&ptr, &dev)) return rc;
    // the array is the current device's copy: a stream of another device would run the kernels
    // there with this device's address
    if (stream) {
        hipDevice_t sdev = -1;
        if (hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess)
            IMGAN_REQUIRE((int)sdev == dev, "imgan_cld_static: the stream belongs to another device "
                                            "than the current one (hipSetDevice first)");
        else
            (void)hipGetLastError();                         // a query some contexts refuse: not fatal
    }
    return cld_run(bgr, B, H, W, cld_out, ptr, (hipStream_t)stream, "imgan_cld_static");
}

static void make_dct(DctTable* D);

static int ehd_block_size(int H, int W) {
    double temp = sqrt((double)W * H / 1100.0);
    int bs = (int)(temp / 2) * 2;
    return bs ? bs : 2;
}

extern "C" size_t imgan_ehd_scratch_bytes(int B, int H, int W) {
    if (B < 1 || H < 1 || W < 1) return 0;
    int bs = ehd_block_size(H, W);
    size_t nb = (size_t)(W / bs) * (H / bs);
    return imgan::align_up((size_t)B * (nb ? nb : 1) * 4 * 4, 256);
}

extern "C" int imgan_ehd(const uint8_t* bgr, int B, int H, int W, int32_t* ehd_out, void* scratch,
                         imgan_stream_t stream) {
    IMGAN_REQUIRE(bgr && ehd_out && scratch, "imgan_ehd: null pointer");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && B <= 65535, "imgan_ehd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    int bs = ehd_block_size(H, W);
    int nbx = W / bs, nby = H / bs;
    // a quadrant sum must fit in 32 bits
    IMGAN_REQUIRE((double)(bs / 2) * (bs / 2) * 255.0 < 4294967295.0, "imgan_ehd: image too large");
    IMGAN_HIP(hipMemsetAsync(ehd_out, 0, (size_t)B * 96 * 4, st));
    if (nbx == 0 || nby == 0) return 0;
    const size_t lds = (size_t)nbx * 2 * 4;
    IMGAN_REQUIRE(lds <= 160 * 1024 - 1024, "imgan_ehd: image too wide for the quadrant table");
    dim3 grid((unsigned)nby * 2, (unsigned)B);
    if (W % 4 == 0 && ((uintptr_t)bgr & 3) == 0) {
        auto fn = ehd_sums_kernel<true>;
        if (lds > 48 * 1024)
            IMGAN_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, bgr, H, W, bs, nbx, nby, (uint32_t*)scratch);
    } else {
        auto fn = ehd_sums_kernel<false>;
        if (lds > 48 * 1024)
            IMGAN_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, bgr, H, W, bs, nbx, nby, (uint32_t*)scratch);
    }
    int total = B * nbx * nby;
    hipLaunchKernelGGL(ehd_classify_kernel, dim3(imgan::cdiv(total, 256)), dim3(256), 0, st,
                       (const uint32_t*)scratch, B, H, W, bs, nbx, nby, 11, ehd_out);
    return imgan::check_launch("imgan_ehd");
}

// imgan_cld + imgan_ehd with one pass over the image for both sets of sums (same results: the sums
// are integers).  Shapes where a strip of edge-block rows could span more than two rows of the
// colour grid (aspect ratios beyond 60 : 1) or that hold no edge block take the two calls.
extern "C" int imgan_cld_ehd(const uint8_t* bgr, int B, int H, int W, int32_t* cld_out, int32_t* ehd_out,
                             void* cld_scratch, void* ehd_scratch, imgan_stream_t stream) {
    if (int rc = cld_check(bgr, B, H, W, cld_out)) return rc;
    IMGAN_REQUIRE(ehd_out && cld_scratch && ehd_scratch, "imgan_cld_ehd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int bs = ehd_block_size(H, W), hb = bs / 2;
    const int nbx = W / bs, nby = H / bs;
    const size_t lds = (size_t)nbx * 2 * 4;
    if (nbx == 0 || nby == 0 || (long long)hb * 8 > H || lds > 48 * 1024 ||
        (double)hb * hb * 255.0 >= 4294967295.0) {
        if (int rc = imgan_cld(bgr, B, H, W, cld_out, cld_scratch, stream)) return rc;
        return imgan_ehd(bgr, B, H, W, ehd_out, ehd_scratch, stream);
    }
    unsigned long long* sums = (unsigned long long*)cld_scratch;
    IMGAN_HIP(hipMemsetAsync(sums, 0, imgan_cld_scratch_bytes(B), st));
    IMGAN_HIP(hipMemsetAsync(ehd_out, 0, (size_t)B * 96 * 4, st));
    const int nstrips = nby * 2 + imgan::cdiv(H - nby * bs, hb);
    dim3 grid((unsigned)nstrips, (unsigned)B);
    if (W % 4 == 0 && ((uintptr_t)bgr & 3) == 0)
        hipLaunchKernelGGL(cld_ehd_sums_kernel<true>, grid, dim3(256), lds, st, bgr, H, W, bs, nbx, nby, sums,
                           (uint32_t*)ehd_scratch);
    else
        hipLaunchKernelGGL(cld_ehd_sums_kernel<false>, grid, dim3(256), lds, st, bgr, H, W, bs, nbx, nby, sums,
                           (uint32_t*)ehd_scratch);
    DctTable T;
    make_dct(&T);
    hipLaunchKernelGGL(cld_finalise_kernel, dim3(B * 3), dim3(64), 0, st, (const unsigned long long*)sums, B,
                       cld_out, T, 0);
    const int total = B * nbx * nby;
    hipLaunchKernelGGL(ehd_classify_kernel, dim3(imgan::cdiv(total, 256)), dim3(256), 0, st,
                       (const uint32_t*)ehd_scratch, B, H, W, bs, nbx, nby, 11, ehd_out);
    return imgan::check_launch("imgan_cld_ehd");
}

extern "C" int imgan_hmmd_quantise(const uint8_t* bgr, uint8_t* out, int64_t npix, int nbins,
                                   imgan_stream_t stream) {
    IMGAN_REQUIRE(bgr && out && npix >= 0, "imgan_hmmd_quantise: bad arguments");
    HmmdTab T;
    if (int rc = make_hmmd(nbins, &T)) return rc;
    if (npix == 0) return 0;
    int grid = (int)std::min<int64_t>(imgan::cdiv(npix, 256), 4096);
    hipLaunchKernelGGL(hmmd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, bgr, out,
                       (long long)npix, T);
    return imgan::check_launch("hmmd_kernel");
}

extern "C" int imgan_csd(const uint8_t* bgr, int B, int H, int W, int nbins, int32_t* csd_out,
                         imgan_stream_t stream) {
    IMGAN_REQUIRE(bgr && csd_out, "imgan_csd: null pointer");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && B <= 65535, "imgan_csd: bad shape");
    HmmdTab T;
    if (int rc = make_hmmd(nbins, &T)) return rc;
    hipStream_t st = (hipStream_t)stream;
    int p = (int)floor(0.5 * log2((double)W * H) - 8.0 + 0.5);
    int K = 1 << (p > 0 ? p : 0);
    int nwx = W >= 8 * K ? (W - 8 * K) / K + 1 : 0;
    int nwy = H >= 8 * K ? (H - 8 * K) / K + 1 : 0;
    IMGAN_HIP(hipMemsetAsync(csd_out, 0, (size_t)B * (nbins + 1) * 4, st));
    if (nwx == 0 || nwy == 0) return 0;
    dim3 grid((unsigned)imgan::cdiv(nwx, CS_W), (unsigned)imgan::cdiv(nwy, CS_H), (unsigned)B);
    hipLaunchKernelGGL(csd_kernel, grid, dim3(256), 0, st, bgr, H, W, K, nwx, nwy, nbins, csd_out, T,
                       (const int32_t*)nullptr, 0, 0, 0);
    return imgan::check_launch("imgan_csd");
}

#ifdef IMGAN_RD_TIMING
extern "C" int imgan_debug_rd_timing(unsigned long long* out16, int reset) {
    IMGAN_HIP(hipDeviceSynchronize());
    IMGAN_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(rd_timing), 128));
    if (reset) {
        unsigned long long z[16] = {0};
        IMGAN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rd_timing), z, 128));
    }
    return 0;
}
#endif

extern "C" size_t imgan_region_descriptors_scratch_bytes(int B, int nseg) {
    return B < 1 || nseg < 1 ? 0 : (size_t)B * nseg * 64 * 4 * 8;
}

static int region_descriptors(const uint8_t* bgr, const int32_t* labels, const int32_t* bbox,
                              const void* list_scratch, int B, int H, int W, int nseg, int nbins,
                              int32_t* cld_out, int32_t* ehd_out, int32_t* csd_out, int32_t* too_big_out,
                              void* scratch, imgan_stream_t stream) {
    IMGAN_REQUIRE(bgr && bbox && cld_out && ehd_out && csd_out && too_big_out && scratch,
                  "imgan_region_descriptors: null pointer");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && nseg >= 1 && B <= 65535,
                  "imgan_region_descriptors: bad shape");
    IMGAN_REQUIRE((long long)B * nseg * 3 < 0x7fffffffLL, "imgan_region_descriptors: too many regions");
    IMGAN_REQUIRE((long long)H * W * 3 < 0x100000000LL && W < (1 << 23) && H < (1 << 23),
                  "imgan_region_descriptors: image too large (H * W * 3 must stay below 2^32)");
    HmmdTab T;
    if (int rc = make_hmmd(nbins, &T)) return rc;
    hipStream_t st = (hipStream_t)stream;
    IMGAN_HIP(hipMemsetAsync(too_big_out, 0, 4, st));
    if (list_scratch) {
        IMGAN_REQUIRE(W <= 65536 && H <= 65535 && (long long)B * H * W < 0xffffffffLL,
                      "imgan_region_descriptors_listed: the pixel lists hold 16-bit positions");
        const imgan_dcd::SegLists L = imgan_dcd::bind_lists(const_cast<void*>(list_scratch), (size_t)B * nseg,
                                                            (size_t)B * H * W);
        hipLaunchKernelGGL(region_desc_kernel<true>, dim3((unsigned)nseg, (unsigned)B), dim3(RD_THREADS), 0, st,
                           bgr, labels, bbox, H, W, nseg, nbins, T, (unsigned long long*)scratch, ehd_out,
                           csd_out, too_big_out, L);
    } else {
        hipLaunchKernelGGL(region_desc_kernel<false>, dim3((unsigned)nseg, (unsigned)B), dim3(RD_THREADS), 0,
                           st, bgr, labels, bbox, H, W, nseg, nbins, T, (unsigned long long*)scratch, ehd_out,
                           csd_out, too_big_out, imgan_dcd::SegLists{});
    }
    DctTable D;
    for (int i = 0; i < 8; ++i) {
        double s = i == 0 ? sqrt(0.125) : 0.5;
        for (int j = 0; j < 8; ++j) D.c[i * 8 + j] = s * cos((M_PI / 8.0) * i * (j + 0.5));
    }
    hipLaunchKernelGGL(cld_finalise_kernel, dim3(B * nseg * 3), dim3(64), 0, st,
                       (const unsigned long long*)scratch, B * nseg, cld_out, D, 1);
    return imgan::check_launch("imgan_region_descriptors");
}

extern "C" int imgan_region_descriptors(const uint8_t* bgr, const int32_t* labels, const int32_t* bbox,
                                        int B, int H, int W, int nseg, int nbins, int32_t* cld_out,
                                        int32_t* ehd_out, int32_t* csd_out, int32_t* too_big_out,
                                        void* scratch, imgan_stream_t stream) {
    IMGAN_REQUIRE(labels, "imgan_region_descriptors: null pointer");
    return region_descriptors(bgr, labels, bbox, nullptr, B, H, W, nseg, nbins, cld_out, ehd_out, csd_out,
                              too_big_out, scratch, stream);
}

extern "C" int imgan_region_descriptors_listed(const uint8_t* bgr, const int32_t* bbox,
                                               const void* list_scratch, int B, int H, int W, int nseg,
                                               int nbins, int32_t* cld_out, int32_t* ehd_out,
                                               int32_t* csd_out, int32_t* too_big_out, void* scratch,
                                               imgan_stream_t stream) {
    IMGAN_REQUIRE(list_scratch, "imgan_region_descriptors_listed: null pointer");
    return region_descriptors(bgr, nullptr, bbox, list_scratch, B, H, W, nseg, nbins, cld_out, ehd_out,
                              csd_out, too_big_out, scratch, stream);
}

static void make_dct(DctTable* D) {
    for (int i = 0; i < 8; ++i) {
        double sc = i == 0 ? sqrt(0.125) : 0.5;
        for (int j = 0; j < 8; ++j) D->c[i * 8 + j] = sc * cos((M_PI / 8.0) * i * (j + 0.5));
    }
}

// edge blocks of a box: (bw / bs)(bh / bs) with bs = 2 floor(sqrt(bw bh / 1100) / 2), at most 2:
// about 1100 for large boxes, up to 4400 where the block size is held at 2
constexpr size_t LG_MAX_BLOCKS = 4608;

extern "C" size_t imgan_region_descriptors_large_scratch_bytes(int H, int W) {
    if (H < 1 || W < 1) return 0;
    return imgan::align_up(64 * 4 * 8 + LG_MAX_BLOCKS * 5 * 4, 256);
}

extern "C" int imgan_region_descriptors_large(const uint8_t* bgr, const int32_t* labels, int H, int W,
                                              int region, int x0, int y0, int x1, int y1, int nbins,
                                              int32_t* cld_out, int32_t* ehd_out, int32_t* csd_out,
                                              void* scratch, imgan_stream_t stream) {
    IMGAN_REQUIRE(bgr && labels && cld_out && ehd_out && csd_out && scratch,
                  "imgan_region_descriptors_large: null pointer");
    IMGAN_REQUIRE(H >= 1 && W >= 1 && W < (1 << 23) && H < (1 << 23),
                  "imgan_region_descriptors_large: bad shape");
    IMGAN_REQUIRE(0 <= x0 && x0 <= x1 && x1 < W && 0 <= y0 && y0 <= y1 && y1 < H,
                  "imgan_region_descriptors_large: the box must lie inside the image");
    HmmdTab T;
    if (int rc = make_hmmd(nbins, &T)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
    const int bs = ehd_block_size(bh, bw), nbx = bw / bs, nby = bh / bs;
    IMGAN_REQUIRE((double)(bs / 2) * (bs / 2) * 255.0 < 4294967295.0,
                  "imgan_region_descriptors_large: box too large");
    IMGAN_REQUIRE((size_t)nbx * nby <= LG_MAX_BLOCKS,
                  "imgan_region_descriptors_large: more edge blocks than the scratch holds");
    unsigned long long* cld = (unsigned long long*)scratch;
    uint32_t* q5 = (uint32_t*)((char*)scratch + 64 * 4 * 8);
    IMGAN_HIP(hipMemsetAsync(scratch, 0, 64 * 4 * 8 + (size_t)std::max(nbx * nby, 1) * 5 * 4, st));
    IMGAN_HIP(hipMemsetAsync(ehd_out, 0, 96 * 4, st));
    IMGAN_HIP(hipMemsetAsync(csd_out, 0, (size_t)(nbins + 1) * 4, st));
    dim3 grid((unsigned)imgan::cdiv(bw, 256), (unsigned)imgan::cdiv(bh, LG_ROWS));
    hipLaunchKernelGGL(region_large_sums_kernel, grid, dim3(256), 0, st, bgr, labels, W, region, x0, y0, bw,
                       bh, bs, nbx, nby, cld, q5);
    if (nbx > 0 && nby > 0)
        hipLaunchKernelGGL(region_large_ehd_kernel, dim3((unsigned)imgan::cdiv(nbx * nby, 256)), dim3(256), 0,
                           st, (const uint32_t*)q5, bw, bh, bs, nbx, nby, 11, ehd_out);
    DctTable D;
    make_dct(&D);
    hipLaunchKernelGGL(cld_finalise_kernel, dim3(3), dim3(64), 0, st, (const unsigned long long*)cld, 1,
                       cld_out, D, 1);
    // colour structure: sub-sampling from the box (oracle: csd_subsample(bh, bw))
    const int pw = (int)floor(0.5 * log2((double)bw * bh) - 8.0 + 0.5);
    const int K = 1 << (pw > 0 ? pw : 0);
    const int nwx = bw >= 8 * K ? (bw - 8 * K) / K + 1 : 0;
    const int nwy = bh >= 8 * K ? (bh - 8 * K) / K + 1 : 0;
    if (nwx > 0 && nwy > 0) {
        dim3 gc((unsigned)imgan::cdiv(nwx, CS_W), (unsigned)imgan::cdiv(nwy, CS_H), 1u);
        hipLaunchKernelGGL(csd_kernel, gc, dim3(256), 0, st, bgr, H, W, K, nwx, nwy, nbins, csd_out, T, labels,
                           region, x0, y0);
    }
    return imgan::check_launch("imgan_region_descriptors_large");
}
