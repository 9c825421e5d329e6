// Dominant colour, LDS-resident path: one workgroup owns one segment, gathers its pixels
// into LDS once and runs the whole Lloyd loop of mpeg7Descriptors.py:85-107 on chip.
//
// Why: the streaming path (dcd.hip) re-reads labels + Lab from HBM every round and every
// segment pays for the slowest one (36 rounds on the bench batch).  Superpixels are compact
// (a few thousand pixels), so a segment's pixels fit LDS and each segment stops at its own
// round.  HBM traffic: one scan of the segment's bounding box + one assignment write.
//
// Arithmetic is the reference's, identical to the streaming path (bit-for-bit: float32
// distance sqrt((d0^2+d1^2)+d2^2) with strict '<', integer sums, float32 centre = sum/count,
// float32 shift sums added in float64).  sqrt is monotone, so it is only evaluated when two
// squared distances are within 2^-20 of each other (where rounding could merge them).
//
// Segments whose pixel count exceeds the LDS capacity (or whose bounding box has more than
// 65535 pixels) are left to the streaming path: SegState.handled stays 0 for them.
#include "common.h"
#include "../../include/imgan.h"
#include "dcd_common.h"

using namespace imgan_dcd;

namespace {

constexpr int SEG_THREADS = 512;
constexpr int SEG_CAP = 11776;              // pixels of one segment held in LDS (6 B each)

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = min(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}

// bbox: int [nsegs][5] = xmin, ymin, xmax, ymax, count (host pre-fills mins with INT_MAX)
__global__ __launch_bounds__(256) void seg_bbox_kernel(const int32_t* __restrict__ labels, int H,
                                                       int W, int nseg, int* __restrict__ bbox) {
    const int b = blockIdx.z;
    const int y = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    int pending = -1;
    if (x < W) {
        int l = labels[((size_t)b * H + y) * W + x];
        if (l >= 0 && l < nseg) pending = l;
    }
    const int lane = threadIdx.x & 63;
    while (true) {
        unsigned long long act = __ballot(pending >= 0);
        if (act == 0ull) break;
        int leader = __ffsll((long long)act) - 1;
        int cur = __shfl(pending, leader);
        bool mine = (pending == cur);
        unsigned long long who = __ballot(mine);
        // lanes are consecutive columns of one row: the extremes are the first / last set bit
        int first = __ffsll((long long)who) - 1, last = 63 - __clzll((long long)who);
        if (lane == leader) {
            int* bb = bbox + ((size_t)b * nseg + cur) * 5;
            int xb = x - lane;
            atomicMin(&bb[0], xb + first);
            atomicMin(&bb[1], y);
            atomicMax(&bb[2], xb + last);
            atomicMax(&bb[3], y);
            atomicAdd(&bb[4], __popcll(who));
        }
        if (mine) pending = -1;
    }
}

__global__ void seg_bbox_init_kernel(int* __restrict__ bbox, int nsegs) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nsegs) return;
    int* bb = bbox + (size_t)t * 5;
    bb[0] = 0x7fffffff;
    bb[1] = 0x7fffffff;
    bb[2] = -1;
    bb[3] = -1;
    bb[4] = 0;
}

template <int KM>
__global__ __launch_bounds__(SEG_THREADS) void dcd_segment_kernel(
    const uint8_t* __restrict__ lab, const int32_t* __restrict__ labels, int H, int W, int nseg,
    int k, int use_hash, uint32_t seed, float* __restrict__ centres, uint8_t* __restrict__ assign,
    SegState* __restrict__ seg, unsigned long long* __restrict__ stats,
    const int* __restrict__ bbox, int* __restrict__ big_count, int max_rounds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* pix = reinterpret_cast<uint32_t*>(smem);                    // Lab | assign << 24
    uint16_t* pos = reinterpret_cast<uint16_t*>(pix + SEG_CAP);           // offset in the bbox
    uint32_t* acc = reinterpret_cast<uint32_t*>(pos + SEG_CAP);           // [KM*3][64] lane-private
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(acc + KM * 3 * 64);
    float* cen = reinterpret_cast<float*>(slots + KM);                   // [KM][3]
    float* old = cen + KM * 3;                                            // [KM][3]
    uint32_t* red = reinterpret_cast<uint32_t*>(old + KM * 3);           // [KM][4] count, sums
    int* flags = reinterpret_cast<int*>(red + KM * 4);                    // no static LDS: keeps
    int& s_n = flags[0];                                                  // the dynamic base 16-B aligned
    int& s_done = flags[1];

    const int b = blockIdx.y, s = blockIdx.x;
    const size_t gs = (size_t)b * nseg + s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int* bb = bbox + gs * 5;
    const int xmin = bb[0], ymin = bb[1], xmax = bb[2], ymax = bb[3], cnt = bb[4];
    SegState st;
    st.old_sum = 0.0;
    st.done = 1;
    st.rounds = 0;
    st.handled = 1;
    st.pad = 0;
    unsigned long long* out = stats + gs * k * 8;
    if (cnt == 0) {                                     // segment id without pixels
        if (tid == 0) seg[gs] = st;
        for (int i = tid; i < k * 8; i += SEG_THREADS) out[i] = 0ull;
        return;
    }
    const int bw = xmax - xmin + 1, bh = ymax - ymin + 1;
    if (cnt > SEG_CAP || (long long)bw * bh > 65535) {  // too large: the streaming path takes it
        if (tid == 0) {
            st.done = 0;
            st.handled = 0;
            seg[gs] = st;
            atomicAdd(big_count, 1);
        }
        return;
    }
    const size_t img = (size_t)b * H * W;
    if (tid == 0) s_n = 0;
    if (tid < KM) slots[tid] = ~0ull;
    __syncthreads();

    // ---- gather the segment's pixels (rows by wave, columns by lane) + sample the init pixels
    for (int yy = wave; yy < bh; yy += SEG_THREADS / 64)
        for (int xx = lane; xx < bw; xx += 64) {
            const size_t p = (size_t)(ymin + yy) * W + xmin + xx;
            if (labels[img + p] != s) continue;
            const uint8_t* q = lab + (img + p) * 3;
            const uint32_t v = q[0] | (q[1] << 8) | (q[2] << 16);
            const int slot = atomicAdd(&s_n, 1);
            pix[slot] = v;
            pos[slot] = (uint16_t)(yy * bw + xx);
            if (use_hash) {
                // stand-in for random.sample (:79): slot h % k goes to the smallest (h, p)
                const uint32_t h = mix32((uint32_t)p * 0x9e3779b1u + seed);
                atomicMin(&slots[h % (uint32_t)k],
                          ((unsigned long long)h << 32) | (unsigned long long)(uint32_t)p);
            }
        }
    __syncthreads();
    const int n = s_n;

    // ---- initial centres
    if (tid < k) {
        float c0, c1, c2;
        if (use_hash) {
            unsigned long long fallback = ~0ull;
            for (int j = 0; j < k; ++j) fallback = slots[j] < fallback ? slots[j] : fallback;
            unsigned long long key = slots[tid] != ~0ull ? slots[tid] : fallback;
            const uint8_t* q = lab + (img + (uint32_t)key) * 3;
            c0 = (float)q[0];
            c1 = (float)q[1];
            c2 = (float)q[2];
        } else {
            const float* src = centres + (gs * k + tid) * 3;
            c0 = src[0];
            c1 = src[1];
            c2 = src[2];
        }
        cen[tid * 3] = c0;
        cen[tid * 3 + 1] = c1;
        cen[tid * 3 + 2] = c2;
        old[tid * 3] = old[tid * 3 + 1] = old[tid * 3 + 2] = 0.f;
    }
    if (tid == 0) s_done = 0;
    double old_sum = 0.0;              // only thread 0's copy is used
    int rounds = 0;
    const float NEAR = 1.0f - 9.5367431640625e-07f;       // 1 - 2^-20

    // ---- Lloyd rounds (:85-107)
    while (true) {
        for (int i = tid; i < k * 2 * 64; i += SEG_THREADS) acc[i] = 0;
        __syncthreads();
        float c[KM][3];
#pragma unroll
        for (int j = 0; j < KM; ++j)
            if (j < k) {
                c[j][0] = cen[j * 3];
                c[j][1] = cen[j * 3 + 1];
                c[j][2] = cen[j * 3 + 2];
            }
        for (int i = tid; i < n; i += SEG_THREADS) {
            const uint32_t v = pix[i];
            const uint32_t u0 = v & 255u, u1 = (v >> 8) & 255u, u2 = (v >> 16) & 255u;
            const float f0 = (float)u0, f1 = (float)u1, f2 = (float)u2;
            float best2 = __builtin_inff(), bestn = __builtin_inff();
            int bi = 0;
#pragma unroll
            for (int j = 0; j < KM; ++j)
                if (j < k) {
                    const float d0 = f0 - c[j][0], d1 = f1 - c[j][1], d2 = f2 - c[j][2];
                    const float q2 = __fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)),
                                               __fmul_rn(d2, d2));
                    bool take = q2 < bestn;                 // clearly smaller after the sqrt too
                    if (!take && q2 < best2)                // near tie: decide on the rounded sqrt
                        take = __fsqrt_rn(q2) < __fsqrt_rn(best2);
                    best2 = take ? q2 : best2;
                    bestn = take ? __fmul_rn(q2, NEAR) : bestn;
                    bi = take ? j : bi;
                }
            pix[i] = (v & 0xffffffu) | ((uint32_t)bi << 24);
            atomicAdd(&acc[(bi * 2) * 64 + lane], 1u | (u0 << 16));
            atomicAdd(&acc[(bi * 2 + 1) * 64 + lane], u1 | (u2 << 16));
        }
        __syncthreads();
        // reduce the 64 lane-private copies: 4 lanes per (cluster, word) row, 16 copies each
        if (tid < k * 2 * 4) {
            const int row = tid >> 2, q = tid & 3;
            uint32_t lo = 0, hi = 0;
            const uint4* src = reinterpret_cast<const uint4*>(acc + row * 64 + q * 16);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint4 w4 = src[t];
                lo += (w4.x & 0xffffu) + (w4.y & 0xffffu) + (w4.z & 0xffffu) + (w4.w & 0xffffu);
                hi += (w4.x >> 16) + (w4.y >> 16) + (w4.z >> 16) + (w4.w >> 16);
            }
            lo += __shfl_xor(lo, 1);
            hi += __shfl_xor(hi, 1);
            lo += __shfl_xor(lo, 2);
            hi += __shfl_xor(hi, 2);
            if (q == 0) {
                red[row * 2] = lo;          // row = cluster*2 + word: {count, s0}, {s1, s2}
                red[row * 2 + 1] = hi;
            }
        }
        __syncthreads();
        if (tid == 0) {                     // centre update + stop rule (:93-107)
            float sh0 = 0.f, sh1 = 0.f, sh2 = 0.f;
            for (int j = 0; j < k; ++j) {
                const uint32_t nj = red[j * 4];
                float c0 = cen[j * 3], c1 = cen[j * 3 + 1], c2 = cen[j * 3 + 2];
                if (nj) {
                    const float fn = (float)nj;
                    c0 = __fdiv_rn((float)red[j * 4 + 1], fn);
                    c1 = __fdiv_rn((float)red[j * 4 + 2], fn);
                    c2 = __fdiv_rn((float)red[j * 4 + 3], fn);
                    cen[j * 3] = c0;
                    cen[j * 3 + 1] = c1;
                    cen[j * 3 + 2] = c2;
                }
                sh0 = __fadd_rn(sh0, fabsf(__fsub_rn(c0, old[j * 3])));
                sh1 = __fadd_rn(sh1, fabsf(__fsub_rn(c1, old[j * 3 + 1])));
                sh2 = __fadd_rn(sh2, fabsf(__fsub_rn(c2, old[j * 3 + 2])));
            }
            const double cent_sum = ((double)sh0 + (double)sh1) + (double)sh2;
            rounds += 1;
            if (fabs(old_sum - cent_sum) < 0.1 || rounds >= max_rounds) {
                s_done = rounds;
            } else {
                for (int j = 0; j < 3 * k; ++j) old[j] = cen[j];
                old_sum = cent_sum;
            }
        }
        __syncthreads();
        if (s_done) break;
    }

    // ---- statistics (:110-116): count / sums come from the last round, squares need a pass
    for (int i = tid; i < k * 3 * 64; i += SEG_THREADS) acc[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += SEG_THREADS) {
        const uint32_t v = pix[i];
        const uint32_t u0 = v & 255u, u1 = (v >> 8) & 255u, u2 = (v >> 16) & 255u, bi = v >> 24;
        atomicAdd(&acc[(bi * 3) * 64 + lane], u0 * u0);
        atomicAdd(&acc[(bi * 3 + 1) * 64 + lane], u1 * u1);
        atomicAdd(&acc[(bi * 3 + 2) * 64 + lane], u2 * u2);
        // assignment back to the image
        const int o = pos[i];
        const int yy = o / bw, xx = o - yy * bw;
        assign[img + (size_t)(ymin + yy) * W + xmin + xx] = (uint8_t)bi;
    }
    __syncthreads();
    if (tid < k * 3 * 4) {
        const int row = tid >> 2, q = tid & 3;
        unsigned long long sum = 0;
        const uint4* src = reinterpret_cast<const uint4*>(acc + row * 64 + q * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint4 w4 = src[t];
            sum += (unsigned long long)w4.x + w4.y + w4.z + w4.w;
        }
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        if (q == 0) out[(row / 3) * 8 + 4 + (row % 3)] = sum;
    }
    if (tid < k) {
        out[tid * 8] = red[tid * 4];
        out[tid * 8 + 1] = red[tid * 4 + 1];
        out[tid * 8 + 2] = red[tid * 4 + 2];
        out[tid * 8 + 3] = red[tid * 4 + 3];
        out[tid * 8 + 7] = 0ull;                        // largest component: filled by the CC stage
        float* dst = centres + (gs * k + tid) * 3;
        dst[0] = cen[tid * 3];
        dst[1] = cen[tid * 3 + 1];
        dst[2] = cen[tid * 3 + 2];
    }
    if (tid == 0) {
        st.rounds = s_done;
        seg[gs] = st;
    }
}

size_t seg_lds_bytes(int km) {
    return (size_t)SEG_CAP * 6 + (size_t)km * 3 * 64 * 4 + (size_t)km * 8 + (size_t)km * 3 * 4 * 2 +
           (size_t)km * 4 * 4 + 16;
}

}  // namespace

extern "C" size_t imgan_dcd_bbox_bytes(int B, int nseg) {
    return (B < 1 || nseg < 1) ? 0 : imgan::align_up((size_t)B * nseg * 5 * 4 + 16, 256);
}

extern "C" int imgan_dcd_segments(const uint8_t* lab, const int32_t* labels, int B, int H, int W,
                                  int nseg, int k, int use_hash_init, uint32_t seed,
                                  float* centres, uint8_t* assign, void* state, uint64_t* stats,
                                  void* bbox_scratch, int32_t* n_big_out, int max_rounds,
                                  imgan_stream_t stream) {
    IMGAN_REQUIRE(lab && labels && centres && assign && state && stats && bbox_scratch && n_big_out,
                  "imgan_dcd_segments: null pointer");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && nseg >= 1 && B <= 65535 && H <= 65535,
                  "imgan_dcd_segments: bad shape");
    IMGAN_REQUIRE(k >= 1 && k <= KMAX, "imgan_dcd_segments: 1 <= k <= 16 dominant colours");
    IMGAN_REQUIRE((long long)H * W < 0x7fffffffLL, "imgan_dcd_segments: image too large");
    IMGAN_REQUIRE(max_rounds >= 1, "imgan_dcd_segments: max_rounds must be positive");
    hipStream_t st = (hipStream_t)stream;
    DcdLayout L = dcd_layout(B, nseg, k);
    SegState* seg = (SegState*)((char*)state + L.off_seg);
    int* bbox = (int*)bbox_scratch;
    size_t nsegs = (size_t)B * nseg;
    IMGAN_HIP(hipMemsetAsync(state, 0, L.total, st));
    IMGAN_HIP(hipMemsetAsync(n_big_out, 0, 4, st));
    IMGAN_HIP(hipMemsetAsync(assign, 0xff, (size_t)B * H * W, st));
    hipLaunchKernelGGL(seg_bbox_init_kernel, dim3((unsigned)imgan::cdiv((long long)nsegs, 256)),
                       dim3(256), 0, st, bbox, (int)nsegs);
    dim3 gb((unsigned)imgan::cdiv(W, 256), (unsigned)H, (unsigned)B);
    hipLaunchKernelGGL(seg_bbox_kernel, gb, dim3(256), 0, st, labels, H, W, nseg, bbox);
    dim3 gs((unsigned)nseg, (unsigned)B);
    if (k <= 8) {
        size_t lds = seg_lds_bytes(8);
        IMGAN_HIP(hipFuncSetAttribute((const void*)dcd_segment_kernel<8>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((dcd_segment_kernel<8>), gs, dim3(SEG_THREADS), lds, st, lab, labels, H,
                           W, nseg, k, use_hash_init, seed, centres, assign, seg,
                           (unsigned long long*)stats, (const int*)bbox, n_big_out, max_rounds);
    } else {
        size_t lds = seg_lds_bytes(16);
        IMGAN_HIP(hipFuncSetAttribute((const void*)dcd_segment_kernel<16>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((dcd_segment_kernel<16>), gs, dim3(SEG_THREADS), lds, st, lab, labels, H,
                           W, nseg, k, use_hash_init, seed, centres, assign, seg,
                           (unsigned long long*)stats, (const int*)bbox, n_big_out, max_rounds);
    }
    return imgan::check_launch("imgan_dcd_segments");
}
