// k x k median blur (k = 3, 5, 7) of a 3-channel u8 image, replicated border.
// Replaces cv2.medianBlur(image, 7) at rag_segmentation.py:11 (exact integer median).
//
// Bit-sliced radix select.  A lane owns 32 horizontally adjacent output pixels of one row and
// works on bit planes: one 32-bit word = one bit of one channel of 32 pixels, so every vector
// instruction advances 32 pixels x 64 lanes.  For the k-th largest of the N = k*k samples of
// a window, from the top bit down: count the samples that are still candidates and have the
// bit set (a carry-save adder tree over N one-bit planes -> a bit-sliced count), compare with
// the bit-sliced number of samples still wanted, keep the bit and narrow the candidates.
// About 3 + 2 instructions per (sample, bit) for 32 pixels: 230 instructions per pixel for
// the three channels of the 7 x 7 median, where a per-pixel binary search on the value with
// the three channels packed in one register (the round-1 kernel) took 980.
//
// One workgroup = a 128 x 64 output tile.  Stage in: every lane loads 32 pixels (96 bytes) of
// one row of the tile's halo region, separates the channels (v_perm) so that register i holds
// pixels i, i+8, i+16, i+24 in its four bytes, and transposes the 8 x 8 bit blocks (three
// butterfly stages of shift + v_bfi); plane words go to LDS (24 planes x 70 rows x 5 words).
// The staged region starts 4 pixels left of the tile, so global loads stay dword aligned and
// the neighbour at dx is the 64-bit pair (word, word + 1) shifted by 4 + dx (v_alignbit).
// Stage out: the eight result planes per channel go back through the same transpose and an
// interleaving v_perm pass into 96 contiguous bytes per lane.
#include "common.h"
#include "../../include/imgan.h"

namespace {

constexpr int MB_TW = 128, MB_TH = 64, MB_WORDS = MB_TW / 32 + 1, MB_LEFT = 4;

struct __attribute__((packed, aligned(4))) Quad32 {
    uint32_t a, b, c, d;
};

__device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); }

// 8 x 8 bit transpose inside every byte lane of r[0..7]: bit k of r[i] <-> bit i of r[k]
template <int S>
__device__ __forceinline__ void transpose_stage(uint32_t (&r)[8]) {
    constexpr uint32_t mask = S == 4 ? 0x0f0f0f0fu : (S == 2 ? 0x33333333u : 0x55555555u);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if ((i & S) == 0) {
            const uint32_t a = r[i], b = r[i + S];
            r[i] = bfi(mask, a, b << S);
            r[i + S] = bfi(mask, a >> S, b);
        }
}
__device__ __forceinline__ void transpose8(uint32_t (&r)[8]) {
    transpose_stage<4>(r);
    transpose_stage<2>(r);
    transpose_stage<1>(r);
}

// v_bitop3_b32: any function of three words, given by its truth table (the function applied
// to 0xf0, 0xcc, 0xaa).  Spelled out because the compiler, left to itself, re-associates the
// select / update expressions into forms with more instructions and more live registers.
constexpr uint32_t TA = 0xf0u, TB = 0xccu, TC = 0xaau;
constexpr uint32_t TT_XOR3 = (TA ^ TB ^ TC) & 0xffu;
constexpr uint32_t TT_MAJ = ((TA & TB) | (TA & TC) | (TB & TC)) & 0xffu;
// borrow out of a - b - c
constexpr uint32_t TT_BORROW = ((~TA & TB) | (~(TA ^ TB) & TC)) & 0xffu;
// a ? b : c
constexpr uint32_t TT_SELECT = ((TA & TB) | (~TA & TC)) & 0xffu;
// a & ~(b ^ c): candidates (a) whose bit (b) equals the bit taken (c)
constexpr uint32_t TT_NARROW = (TA & ~(TB ^ TC)) & 0xffu;
#define BITOP3(a, b, c, tt) __builtin_amdgcn_bitop3_b32((a), (b), (c), (tt))

// count[b] = bit b of the number of set planes among v[0..N): carry-save adders, N/2 per level,
// two instructions per full adder
template <int N, int NB>
__device__ __forceinline__ void plane_count(uint32_t (&v)[N], uint32_t (&count)[NB]) {
    if constexpr (N == 1) {
        count[0] = v[0];
#pragma unroll
        for (int b = 1; b < NB; ++b) count[b] = 0u;
    } else {
        uint32_t carry[N / 2];
        uint32_t acc = v[0];
#pragma unroll
        for (int i = 1, j = 0; i < N; i += 2, ++j) {
            if (i + 1 < N) {                                   // full adder
                carry[j] = BITOP3(acc, v[i], v[i + 1], TT_MAJ);
                acc = BITOP3(acc, v[i], v[i + 1], TT_XOR3);
            } else {                                           // half adder
                carry[j] = acc & v[i];
                acc ^= v[i];
            }
        }
        count[0] = acc;
        if constexpr (NB > 1) {
            uint32_t up[NB - 1];
            plane_count<N / 2, NB - 1>(carry, up);
#pragma unroll
            for (int b = 1; b < NB; ++b) count[b] = up[b - 1];
        }
    }
}

template <int KS>
__global__ __launch_bounds__(256, 2) void median_bits_kernel(const uint8_t* __restrict__ src,
                                                          uint8_t* __restrict__ dst, int H, int W,
                                                          int fast) {
    constexpr int R = KS / 2, N = KS * KS, ROWS = MB_TH + 2 * R;
    constexpr int NB = N > 31 ? 6 : (N > 15 ? 5 : 4);
    __shared__ uint32_t planes[24 * ROWS * MB_WORDS];          // [channel * 8 + bit][row][word]
    const int b = blockIdx.z, x0 = blockIdx.x * MB_TW, y0 = blockIdx.y * MB_TH;
    const uint8_t* im = src + (size_t)b * H * W * 3;
    const int tid = threadIdx.x;

    // ---- stage in: bit planes of the tile + halo
    for (int item = tid; item < ROWS * MB_WORDS; item += 256) {
        const int row = item / MB_WORDS, w = item - row * MB_WORDS;
        const int y = min(max(y0 + row - R, 0), H - 1);
        const int xs = x0 - MB_LEFT + 32 * w;
        const uint8_t* line = im + (size_t)y * W * 3;
        uint32_t raw[24];                                      // 32 pixels x BGR
        if (fast && xs >= 0 && xs + 32 <= W) {
            const Quad32* q = reinterpret_cast<const Quad32*>(line + (size_t)xs * 3);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const Quad32 v = q[i];
                raw[4 * i] = v.a, raw[4 * i + 1] = v.b, raw[4 * i + 2] = v.c, raw[4 * i + 3] = v.d;
            }
        } else {                                               // image border, or W % 4 != 0
#pragma unroll
            for (int i = 0; i < 24; ++i) raw[i] = 0u;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int xx = min(max(xs + j, 0), W - 1);
                const uint8_t* p = line + (size_t)xx * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    raw[(3 * j + c) >> 2] |= (uint32_t)p[c] << (8 * ((3 * j + c) & 3));
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uint32_t e[8];                                     // e[i] = pixels i, i+8, i+16, i+24
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int o0 = 3 * i + c, o1 = o0 + 24, o2 = o0 + 48, o3 = o0 + 72;   // byte offsets
                const uint32_t lo = __builtin_amdgcn_perm(raw[o1 >> 2], raw[o0 >> 2],
                                                          0x0c0c0000u | (o0 & 3) | ((4 + (o1 & 3)) << 8));
                const uint32_t hi = __builtin_amdgcn_perm(raw[o3 >> 2], raw[o2 >> 2],
                                                          0x00000c0cu | ((o2 & 3) << 16) | ((4 + (o3 & 3)) << 24));
                e[i] = lo | hi;
            }
            transpose8(e);                                     // e[k] = bit k of the 32 pixels
#pragma unroll
            for (int k = 0; k < 8; ++k) planes[((c * 8 + k) * ROWS + row) * MB_WORDS + w] = e[k];
        }
    }
    __syncthreads();

    // ---- select: lane = (row, word) of the tile.  The channel loop stays rolled (one copy of the
    // eight unrolled bit steps, about 20 KB of code, fits the instruction cache).
    const int row = tid >> 2, ow = tid & 3;
    uint32_t out0[8], out1[8], out2[8];
#pragma nounroll
    for (int c = 0; c < 3; ++c) {
        uint32_t cand[N];
#pragma unroll
        for (int n = 0; n < N; ++n) cand[n] = 0xffffffffu;
        uint32_t want[NB];                                     // bit-sliced (rank from the top) - 1
#pragma unroll
        for (int bit = 0; bit < NB; ++bit) want[bit] = (((N - 1) / 2) >> bit) & 1 ? 0xffffffffu : 0u;
        uint32_t res[8];
        // plane words of the current bit step and of the next one (loaded one step ahead)
        uint32_t lo[KS], hi[KS], nlo[KS], nhi[KS];
        int off = ((c * 8 + 7) * ROWS + row) * MB_WORDS + ow;
        uint32_t prev_take = 0u;
#pragma unroll
        for (int dy = 0; dy < KS; ++dy) lo[dy] = planes[off + dy * MB_WORDS], hi[dy] = planes[off + dy * MB_WORDS + 1];
#pragma unroll
        for (int k = 7; k >= 0; --k) {
            if (k > 0) {
                // the planes of later bit steps depend on nothing computed here, and the compiler
                // would hoist all eight steps' loads to the top (8 x 2 x KS live registers): tie
                // the next step's address to the previous step's result, so that it is loaded one
                // step ahead and no earlier
                off -= ROWS * MB_WORDS;
                asm volatile("" : "+v"(off) : "v"(prev_take));
#pragma unroll
                for (int dy = 0; dy < KS; ++dy)
                    nlo[dy] = planes[off + dy * MB_WORDS], nhi[dy] = planes[off + dy * MB_WORDS + 1];
            }
            uint32_t p[N], in[N];
#pragma unroll
            for (int dy = 0; dy < KS; ++dy) {
#pragma unroll
                for (int dx = 0; dx < KS; ++dx) {
                    p[dy * KS + dx] =
                        (uint32_t)(((((uint64_t)hi[dy]) << 32) | lo[dy]) >> (MB_LEFT - R + dx));
                    in[dy * KS + dx] = p[dy * KS + dx] & cand[dy * KS + dx];
                }
            }
            uint32_t ones[NB];
            plane_count<N, NB>(in, ones);
            // want - ones, bit-sliced; the borrow out says want < ones: enough candidates have the bit
            uint32_t borrow = 0u, diff[NB];
#pragma unroll
            for (int bit = 0; bit < NB; ++bit) {
                diff[bit] = BITOP3(want[bit], ones[bit], borrow, TT_XOR3);
                borrow = BITOP3(want[bit], ones[bit], borrow, TT_BORROW);
            }
            const uint32_t take = borrow;
            res[k] = take;
            prev_take = take;
            if (k > 0) {
#pragma unroll
                for (int bit = 0; bit < NB; ++bit) want[bit] = BITOP3(take, want[bit], diff[bit], TT_SELECT);
#pragma unroll
                for (int n = 0; n < N; ++n) cand[n] = BITOP3(cand[n], p[n], take, TT_NARROW);
            }
#pragma unroll
            for (int dy = 0; dy < KS; ++dy) lo[dy] = nlo[dy], hi[dy] = nhi[dy];
            __builtin_amdgcn_sched_barrier(0);                 // no motion across bit steps
        }
        transpose8(res);                                       // back to pixels i, i+8, i+16, i+24
        if (c == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) out0[i] = res[i];
        } else if (c == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) out1[i] = res[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) out2[i] = res[i];
        }
    }

    // ---- stage out: interleave the channels, 96 contiguous bytes per lane
    const int y = y0 + row, x = x0 + 32 * ow;
    if (y >= H || x >= W) return;
    uint32_t o[24];
#pragma unroll
    for (int t = 0; t < 24; ++t) {
        uint32_t sel_lo = 0x0c0c0000u, sel_hi = 0x00000c0cu;
        uint32_t s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int byte = 4 * t + u, j = byte / 3, c = byte - 3 * j;
            s[u] = c == 0 ? out0[j & 7] : (c == 1 ? out1[j & 7] : out2[j & 7]);
            const uint32_t lane = (uint32_t)(j >> 3) + ((u & 1) ? 4u : 0u);
            if (u < 2) sel_lo |= lane << (8 * u);
            else sel_hi |= lane << (8 * u);
        }
        o[t] = __builtin_amdgcn_perm(s[1], s[0], sel_lo) | __builtin_amdgcn_perm(s[3], s[2], sel_hi);
    }
    uint8_t* out = dst + (((size_t)b * H + y) * W + x) * 3;
    if (fast && x + 32 <= W) {
        Quad32* q = reinterpret_cast<Quad32*>(out);
#pragma unroll
        for (int i = 0; i < 6; ++i) q[i] = Quad32{o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]};
    } else {
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (x + j < W) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    out[3 * j + c] = (uint8_t)(o[(3 * j + c) >> 2] >> (8 * ((3 * j + c) & 3)));
            }
    }
}

template <int KS>
int launch_median(const uint8_t* src, uint8_t* dst, int B, int H, int W, hipStream_t st) {
    // dword loads / stores of a row need every row start dword aligned
    const int fast = (W % 4 == 0) && (((uintptr_t)src | (uintptr_t)dst) & 3) == 0;
    dim3 grid((unsigned)imgan::cdiv(W, MB_TW), (unsigned)imgan::cdiv(H, MB_TH), (unsigned)B);
    hipLaunchKernelGGL(median_bits_kernel<KS>, grid, dim3(256), 0, st, src, dst, H, W, fast);
    return imgan::check_launch("median_bits_kernel");
}

}  // namespace

// Any other odd size (cv2.medianBlur takes them all; rag_segmentation.py:10-11 passes its
// median_blur argument on): one thread per pixel, an 8-step search for the largest value v with at
// least k^2 / 2 + 1 window values >= v, the three channels side by side, the window read from the
// image with the replicated border as clamped coordinates.  k^2 byte loads per step and channel:
// a path for completeness, not for speed.
__global__ __launch_bounds__(256) void median_any_kernel(const uint8_t* __restrict__ src,
                                                         uint8_t* __restrict__ dst, int H, int W, int k) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t img = (size_t)blockIdx.z * H * W * 3;
    const int r = k >> 1, need = (k * k) / 2 + 1;
    int lo0 = 0, lo1 = 0, lo2 = 0;
    for (int bit = 7; bit >= 0; --bit) {
        const int t0 = lo0 | (1 << bit), t1 = lo1 | (1 << bit), t2 = lo2 | (1 << bit);
        int c0 = 0, c1 = 0, c2 = 0;
        for (int dy = -r; dy <= r; ++dy) {
            const int yy = min(max(y + dy, 0), H - 1);
            const uint8_t* row = src + img + (size_t)yy * W * 3;
            for (int dx = -r; dx <= r; ++dx) {
                const uint8_t* p = row + (size_t)min(max(x + dx, 0), W - 1) * 3;
                c0 += p[0] >= t0;
                c1 += p[1] >= t1;
                c2 += p[2] >= t2;
            }
        }
        lo0 = c0 >= need ? t0 : lo0;
        lo1 = c1 >= need ? t1 : lo1;
        lo2 = c2 >= need ? t2 : lo2;
    }
    uint8_t* o = dst + img + ((size_t)y * W + x) * 3;
    o[0] = (uint8_t)lo0, o[1] = (uint8_t)lo1, o[2] = (uint8_t)lo2;
}

extern "C" int imgan_median_u8c3(const uint8_t* src, uint8_t* dst, int B, int H, int W, int ksize,
                                 imgan_stream_t stream) {
    IMGAN_REQUIRE(src && dst, "imgan_median_u8c3: null pointer");
    IMGAN_REQUIRE(src != dst, "imgan_median_u8c3: in-place operation is not supported");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && B <= 65535, "imgan_median_u8c3: bad shape");
    IMGAN_REQUIRE((long long)imgan::cdiv(H, MB_TH) <= 65535, "imgan_median_u8c3: image too tall");
    hipStream_t st = (hipStream_t)stream;
    switch (ksize) {
        case 3: return launch_median<3>(src, dst, B, H, W, st);
        case 5: return launch_median<5>(src, dst, B, H, W, st);
        case 7: return launch_median<7>(src, dst, B, H, W, st);
        default:
            break;
    }
    IMGAN_REQUIRE(ksize >= 1 && ksize <= 255 && (ksize & 1),
                  "imgan_median_u8c3: ksize must be odd, 1 to 255 (cv2.medianBlur: odd and > 1)");
    IMGAN_REQUIRE((long long)imgan::cdiv(H, 4) <= 65535, "imgan_median_u8c3: image too tall for this kernel size");
    hipLaunchKernelGGL(median_any_kernel, dim3((unsigned)imgan::cdiv(W, 64), (unsigned)imgan::cdiv(H, 4), (unsigned)B),
                       dim3(256), 0, st, src, dst, H, W, ksize);
    return imgan::check_launch("median_any_kernel");
}

extern "C" int imgan_median7_u8c3(const uint8_t* src, uint8_t* dst, int B, int H, int W,
                                  imgan_stream_t stream) {
    return imgan_median_u8c3(src, dst, B, H, W, 7, stream);
}
