// MPEG-7 dominant colour per segment: k-means in Lab + per-colour statistics.
// Replaces MPEG7Descriptors.dominant_colour (scripts/mpeg7Descriptors.py:76-130).
//
// Reference arithmetic kept: float32 distance sqrt((d0^2 + d1^2) + d2^2) with strict '<'
// (lowest cluster index wins ties, :87-92); centre = float32(sum) / count (:96-99); stop rule
// |previous shift sum - shift sum| < 0.1 with the shift summed in float32 per component and
// the three components added in float64 (:102-107).  Deviations (documented in DESIGN.md):
// initial centres are injected, an empty cluster is re-seeded by a hash rule in place of
// random.sample (:101, resample_key in dcd_common.h), spatial coherence uses true 8-connectivity.
//
// Every segment runs its own loop: a segment that met the stop rule is frozen (its pixels are
// skipped), so one "round" call advances only the segments still running.
//
// Per-(segment, colour) sums are integers, accumulated in a workgroup-local LDS hash table
// (ds_add, few distinct keys per workgroup because segments are spatially coherent) and
// flushed with 64-bit global atomics: results are order independent.
#include "common.h"
#include "../../include/imgan.h"
#include "cc_device.h"
#include "dcd_common.h"

using namespace imgan_dcd;

namespace {

template <int NF>
struct BlockHash {
    int keys[HCAP];
    uint32_t vals[HCAP * NF];

    __device__ void init() {
        for (int i = threadIdx.x; i < HCAP; i += blockDim.x) keys[i] = -1;
        for (int i = threadIdx.x; i < HCAP * NF; i += blockDim.x) vals[i] = 0;
    }
    // returns false when the table is full around the key: caller falls back to global atomics
    __device__ bool add(int key, const uint32_t* v) {
        uint32_t h = ((uint32_t)key * 2654435761u) >> 23;
        for (int probe = 0; probe < 32; ++probe) {
            int cur = keys[h];
            if (cur != key) {
                if (cur != -1) {
                    h = (h + 1) & (HCAP - 1);
                    continue;
                }
                int old = atomicCAS(&keys[h], -1, key);
                if (old != -1 && old != key) {
                    h = (h + 1) & (HCAP - 1);
                    continue;
                }
            }
#pragma unroll
            for (int f = 0; f < NF; ++f)
                if (v[f]) atomicAdd(&vals[h * NF + f], v[f]);
            return true;
        }
        return false;
    }
};

// ---- Lloyd assignment (:86-92) + sums for the centre update (:93-99)
__global__ __launch_bounds__(256) void dcd_assign_kernel(
    const uint8_t* __restrict__ lab, const int32_t* __restrict__ labels, long long npix, int nseg,
    int k, const float* __restrict__ centres, uint8_t* __restrict__ assign,
    const SegState* __restrict__ seg, unsigned long long* __restrict__ acc, int first_round) {
    __shared__ BlockHash<4> tab;
    tab.init();
    __syncthreads();
    const int b = blockIdx.y;
    const int32_t* L = labels + (size_t)b * npix;
    const uint8_t* im = lab + (size_t)b * npix * 3;
    uint8_t* A = assign + (size_t)b * npix;
    const long long chunk = (long long)blockIdx.x * (256 * 8);
    int cached = -1;
    float c[KMAX][3];
    for (int it = 0; it < 8; ++it) {
        long long p = chunk + (long long)it * 256 + threadIdx.x;
        if (p >= npix) break;
        int s = L[p];
        if (s < 0 || s >= nseg) {
            if (first_round) A[p] = 0xff;
            continue;
        }
        size_t gs = (size_t)b * nseg + s;
        if (seg[gs].done) continue;
        if (s != cached) {
            const float* src = centres + gs * k * 3;
#pragma unroll
            for (int j = 0; j < KMAX; ++j)
                if (j < k) {
                    c[j][0] = src[3 * j];
                    c[j][1] = src[3 * j + 1];
                    c[j][2] = src[3 * j + 2];
                }
            cached = s;
        }
        const uint8_t* q = im + p * 3;
        const uint32_t u0 = q[0], u1 = q[1], u2 = q[2];
        const float f0 = (float)u0, f1 = (float)u1, f2 = (float)u2;
        float best = 0.f;
        int bi = -1;
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
            if (j < k) {
                float d0 = f0 - c[j][0], d1 = f1 - c[j][1], d2 = f2 - c[j][2];
                float d = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)),
                                               __fmul_rn(d2, d2)));
                if (bi < 0 || d < best) {      // 'distance > dist' against +inf first (:86-91)
                    best = d;
                    bi = j;
                }
            }
        A[p] = (uint8_t)bi;
        uint32_t v[4] = {1u, u0, u1, u2};
        int key = s * k + bi;
        if (!tab.add(key, v)) {
            unsigned long long* a = acc + (gs * k + bi) * 4;
            atomicAdd(&a[0], 1ull);
            atomicAdd(&a[1], (unsigned long long)u0);
            atomicAdd(&a[2], (unsigned long long)u1);
            atomicAdd(&a[3], (unsigned long long)u2);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HCAP; i += 256) {
        int key = tab.keys[i];
        if (key < 0) continue;
        unsigned long long* a = acc + ((size_t)b * nseg * k + key) * 4;
#pragma unroll
        for (int f = 0; f < 4; ++f)
            if (tab.vals[i * 4 + f]) atomicAdd(&a[f], (unsigned long long)tab.vals[i * 4 + f]);
    }
}

// ---- centre update + stop rule (:93-107), one thread per segment.  An empty cluster takes its
// re-sample draw (:101): known from an earlier round, or looked for by the two kernels below, in
// which case the segment's round is finished there (pending).
__device__ __forceinline__ void finish_round(SegState& st, int k, const float* c, float* o,
                                             int* __restrict__ active) {
    float sh0 = 0.f, sh1 = 0.f, sh2 = 0.f;
    for (int j = 0; j < k; ++j) {
        sh0 = __fadd_rn(sh0, fabsf(__fsub_rn(c[3 * j], o[3 * j])));
        sh1 = __fadd_rn(sh1, fabsf(__fsub_rn(c[3 * j + 1], o[3 * j + 1])));
        sh2 = __fadd_rn(sh2, fabsf(__fsub_rn(c[3 * j + 2], o[3 * j + 2])));
    }
    const double cent_sum = ((double)sh0 + (double)sh1) + (double)sh2;
    st.rounds += 1;
    if (fabs(st.old_sum - cent_sum) < 0.1) {
        st.done = 1;
    } else {
        for (int j = 0; j < 3 * k; ++j) o[j] = c[j];
        st.old_sum = cent_sum;
        if (active) atomicAdd(active, 1);
    }
}

__global__ void dcd_update_kernel(int nsegs, int nseg, int k, long long npix,
                                  const uint8_t* __restrict__ lab, float* __restrict__ centres,
                                  SegState* __restrict__ seg, float* __restrict__ oldcen,
                                  unsigned long long* __restrict__ acc,
                                  unsigned long long* __restrict__ draw, int* __restrict__ wanted,
                                  int* __restrict__ active) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsegs) return;
    SegState st = seg[s];
    if (st.done) return;
    float* c = centres + (size_t)s * k * 3;
    float* o = oldcen + (size_t)s * k * 3;
    unsigned long long* a = acc + (size_t)s * k * 4;
    unsigned long long* dr = draw + (size_t)s * k;
    const uint8_t* im = lab + (size_t)(s / nseg) * npix * 3;
    int pending = 0;
    for (int j = 0; j < k; ++j) {
        unsigned long long n = a[4 * j];
        if (n) {
            float fn = (float)n;
            c[3 * j] = __fdiv_rn((float)a[4 * j + 1], fn);
            c[3 * j + 1] = __fdiv_rn((float)a[4 * j + 2], fn);
            c[3 * j + 2] = __fdiv_rn((float)a[4 * j + 3], fn);
        } else if (dr[j] > DRAW_WANTED) {                 // drawn in an earlier round
            const uint8_t* q = im + (size_t)(uint32_t)(~dr[j]) * 3;
            c[3 * j] = (float)q[0];
            c[3 * j + 1] = (float)q[1];
            c[3 * j + 2] = (float)q[2];
        } else {
            dr[j] = DRAW_WANTED;
            pending |= 1 << j;
        }
        a[4 * j] = a[4 * j + 1] = a[4 * j + 2] = a[4 * j + 3] = 0ull;
    }
    st.pending = pending;
    if (pending)
        atomicAdd(wanted, 1);
    else
        finish_round(st, k, c, o, active);
    seg[s] = st;
}

// the draws asked for in this round: every pixel of a pending segment bids for its clusters
__global__ __launch_bounds__(256) void dcd_draw_kernel(const int32_t* __restrict__ labels,
                                                       long long npix, int nseg, int k, uint32_t seed,
                                                       const SegState* __restrict__ seg,
                                                       unsigned long long* __restrict__ draw,
                                                       const int* __restrict__ wanted) {
    if (*wanted == 0) return;
    const int b = blockIdx.y;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix;
         p += (long long)gridDim.x * 256) {
        const int s = labels[(size_t)b * npix + p];
        if (s < 0 || s >= nseg) continue;
        const size_t gs = (size_t)b * nseg + s;
        const int pending = seg[gs].done ? 0 : seg[gs].pending;
        for (int j = 0; j < k; ++j)
            if ((pending >> j) & 1) {
                const unsigned long long bid = ~resample_key((uint32_t)p, seed, j);
                unsigned long long* slot = draw + gs * k + j;
                // the slot only ever grows: a stale read just costs an extra atomic
                if (bid > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, bid);
            }
    }
}

__global__ void dcd_draw_apply_kernel(int nsegs, int nseg, int k, long long npix,
                                      const uint8_t* __restrict__ lab, float* __restrict__ centres,
                                      SegState* __restrict__ seg, float* __restrict__ oldcen,
                                      const unsigned long long* __restrict__ draw,
                                      const int* __restrict__ wanted, int* __restrict__ active) {
    if (*wanted == 0) return;
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsegs) return;
    SegState st = seg[s];
    if (st.done || !st.pending) return;
    float* c = centres + (size_t)s * k * 3;
    const uint8_t* im = lab + (size_t)(s / nseg) * npix * 3;
    for (int j = 0; j < k; ++j)
        if ((st.pending >> j) & 1) {
            const unsigned long long d = draw[(size_t)s * k + j];
            if (d > DRAW_WANTED) {                        // else: a segment without pixels
                const uint8_t* q = im + (size_t)(uint32_t)(~d) * 3;
                c[3 * j] = (float)q[0];
                c[3 * j + 1] = (float)q[1];
                c[3 * j + 2] = (float)q[2];
            }
        }
    st.pending = 0;
    finish_round(st, k, c, oldcen + (size_t)s * k * 3, active);
    seg[s] = st;
}

// ---- statistics (:110-129)
__global__ __launch_bounds__(256) void dcd_values_kernel(
    const uint8_t* __restrict__ lab, const int32_t* __restrict__ labels,
    const uint8_t* __restrict__ assign, long long npix, int nseg, int k,
    int32_t* __restrict__ values, int32_t* __restrict__ sizes,
    unsigned long long* __restrict__ stats, const SegState* __restrict__ seg) {
    __shared__ BlockHash<7> tab;
    tab.init();
    __syncthreads();
    const int b = blockIdx.y;
    const long long chunk = (long long)blockIdx.x * (256 * 8);
    for (int it = 0; it < 8; ++it) {
        long long p = chunk + (long long)it * 256 + threadIdx.x;
        if (p >= npix) break;
        size_t gp = (size_t)b * npix + p;
        int s = labels[gp];
        int j = assign[gp];
        int key = (s >= 0 && s < nseg && j < k) ? s * k + j : -1;
        // segments finished on chip have all eight statistics already: leave them out
        if (key >= 0 && seg && seg[(size_t)b * nseg + s].handled) key = -1;
        values[gp] = key;
        sizes[gp] = 0;
        if (key < 0) continue;
        const uint8_t* q = lab + gp * 3;
        uint32_t u0 = q[0], u1 = q[1], u2 = q[2];
        uint32_t v[7] = {1u, u0, u1, u2, u0 * u0, u1 * u1, u2 * u2};
        if (!tab.add(key, v)) {
            unsigned long long* a = stats + ((size_t)b * nseg * k + key) * 8;
#pragma unroll
            for (int f = 0; f < 7; ++f) atomicAdd(&a[f], (unsigned long long)v[f]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HCAP; i += 256) {
        int key = tab.keys[i];
        if (key < 0) continue;
        unsigned long long* a = stats + ((size_t)b * nseg * k + key) * 8;
#pragma unroll
        for (int f = 0; f < 7; ++f)
            if (tab.vals[i * 7 + f]) atomicAdd(&a[f], (unsigned long long)tab.vals[i * 7 + f]);
    }
}

__global__ void dcd_zero_stats_kernel(unsigned long long* __restrict__ stats, int nsegs, int k,
                                      const SegState* __restrict__ seg) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nsegs * k) return;
    int s = t / k;
    unsigned long long* a = stats + (size_t)t * 8;
    if (seg && seg[s].handled) return;      // finished on chip, statistics included
    for (int f = 0; f < 8; ++f) a[f] = 0ull;
}

__global__ __launch_bounds__(256) void dcd_sizes_kernel(const int32_t* __restrict__ roots,
                                                        int32_t* __restrict__ sizes,
                                                        long long npix) {
    const int b = blockIdx.y;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix;
         p += (long long)gridDim.x * 256) {
        int r = roots[(size_t)b * npix + p];
        if (r >= 0) atomicAdd(&sizes[(size_t)b * npix + r], 1);
    }
}

__global__ __launch_bounds__(256) void dcd_largest_kernel(const int32_t* __restrict__ values,
                                                          const int32_t* __restrict__ roots,
                                                          const int32_t* __restrict__ sizes,
                                                          long long npix, int nsegk,
                                                          unsigned long long* __restrict__ stats) {
    const int b = blockIdx.y;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix;
         p += (long long)gridDim.x * 256) {
        size_t gp = (size_t)b * npix + p;
        if (roots[gp] == (int32_t)p)
            atomicMax(&stats[((size_t)b * nsegk + values[gp]) * 8 + 7],
                      (unsigned long long)sizes[gp]);
    }
}

// ---- deterministic stand-in for random.sample(segment, k) (:79) for device-resident batches:
// pixel p competes for slot j = h(p) % k of its segment; the slot goes to the smallest
// (h(p), p).  k distinct pixels per segment, reproducible, no host round trip.
__global__ __launch_bounds__(256) void dcd_init_pick_kernel(const int32_t* __restrict__ labels,
                                                            long long npix, int nseg, int k,
                                                            uint32_t seed,
                                                            unsigned long long* __restrict__ slots,
                                                            const SegState* __restrict__ seg) {
    const int b = blockIdx.y;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix;
         p += (long long)gridDim.x * 256) {
        int s = labels[(size_t)b * npix + p];
        if (s < 0 || s >= nseg) continue;
        if (seg && seg[(size_t)b * nseg + s].handled) continue;
        uint32_t h = mix32((uint32_t)p * 0x9e3779b1u + seed);
        unsigned long long key = ((unsigned long long)h << 32) | (unsigned long long)(uint32_t)p;
        unsigned long long* slot = slots + ((size_t)b * nseg + s) * k + (h % (uint32_t)k);
        // the slot only ever decreases: a stale read just costs an extra atomic
        if (key < __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMin(slot, key);
    }
}

__global__ void dcd_init_fetch_kernel(const uint8_t* __restrict__ lab, long long npix, int nsegs,
                                      int nseg, int k, const unsigned long long* __restrict__ slots,
                                      float* __restrict__ centres,
                                      const SegState* __restrict__ seg) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nsegs) return;
    if (seg && seg[t].handled) return;
    int b = t / nseg;
    const unsigned long long* sl = slots + (size_t)t * k;
    unsigned long long fallback = ~0ull;
    for (int j = 0; j < k; ++j) fallback = sl[j] < fallback ? sl[j] : fallback;
    for (int j = 0; j < k; ++j) {
        unsigned long long key = sl[j] != ~0ull ? sl[j] : fallback;   // empty slot: tiny segment
        float* c = centres + ((size_t)t * k + j) * 3;
        if (key == ~0ull) {
            c[0] = c[1] = c[2] = 0.f;                                  // segment without pixels
        } else {
            const uint8_t* q = lab + ((size_t)b * npix + (uint32_t)key) * 3;
            c[0] = (float)q[0];
            c[1] = (float)q[1];
            c[2] = (float)q[2];
        }
    }
}

// stats + centres -> [mean(3), var(3), percentage, coherence] as float32 (mpeg7Descriptors.py:114-129)
__global__ void dcd_features_kernel(const unsigned long long* __restrict__ stats,
                                    const float* __restrict__ centres, int nsegs, int k,
                                    float* __restrict__ out) {
    // one thread per (segment, colour): the rows of stats and out are read and written whole
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)nsegs * k) return;
    const long long seg = t / k;
    const unsigned long long* st = stats + (size_t)seg * k * 8;
    unsigned long long total = 0;
    for (int j = 0; j < k; ++j) total += st[8 * j];
    const unsigned long long* a = stats + (size_t)t * 8;
    float* o = out + (size_t)t * 8;
    const float* c = centres + (size_t)t * 3;
    float r[8];
    r[0] = c[0];
    r[1] = c[1];
    r[2] = c[2];
    const unsigned long long cnt = a[0];
    const double n = (double)cnt;
    for (int ch = 0; ch < 3; ++ch) {
        double sx = (double)a[1 + ch], sxx = (double)a[4 + ch];
        r[3 + ch] = cnt ? (float)((n * sxx - sx * sx) / (n * n)) : __builtin_nanf("");
    }
    r[6] = total ? (float)(n / (double)total) : 0.f;
    r[7] = cnt ? (float)((double)a[7] / n) : 0.f;
    reinterpret_cast<float4*>(o)[0] = make_float4(r[0], r[1], r[2], r[3]);
    reinterpret_cast<float4*>(o)[1] = make_float4(r[4], r[5], r[6], r[7]);
}

}  // namespace

extern "C" size_t imgan_dcd_state_bytes(int B, int nseg, int k) {
    if (B < 1 || nseg < 1 || k < 1) return 0;
    return dcd_layout(B, nseg, k).total;
}

extern "C" int imgan_dcd_kmeans_round(const uint8_t* lab, const int32_t* labels, int B, int H,
                                      int W, int nseg, int k, uint32_t seed, float* centres,
                                      uint8_t* assign, void* state, int first_round,
                                      int32_t* active_out, imgan_stream_t stream) {
    IMGAN_REQUIRE(lab && labels && centres && assign && state, "imgan_dcd_kmeans_round: null pointer");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && nseg >= 1 && B <= 65535,
                  "imgan_dcd_kmeans_round: bad shape");
    IMGAN_REQUIRE(k >= 1 && k <= KMAX, "imgan_dcd_kmeans_round: 1 <= k <= 16 dominant colours");
    IMGAN_REQUIRE((long long)nseg * k < 0x7fffffffLL, "imgan_dcd_kmeans_round: nseg*k too large");
    DcdLayout L = dcd_layout(B, nseg, k);
    char* base = (char*)state;
    SegState* seg = (SegState*)(base + L.off_seg);
    float* oldcen = (float*)(base + L.off_oldcen);
    unsigned long long* acc = (unsigned long long*)(base + L.off_acc);
    unsigned long long* draw = (unsigned long long*)(base + L.off_draw);
    int* wanted = (int*)(base + L.off_wanted);
    hipStream_t st = (hipStream_t)stream;
    long long npix = (long long)H * W;
    IMGAN_REQUIRE(npix < 0x7fffffffLL, "imgan_dcd_kmeans_round: image too large");
    if (active_out) IMGAN_HIP(hipMemsetAsync(active_out, 0, sizeof(int32_t), st));
    IMGAN_HIP(hipMemsetAsync(wanted, 0, sizeof(int), st));
    dim3 grid((unsigned)imgan::cdiv(npix, 256 * 8), (unsigned)B);
    hipLaunchKernelGGL(dcd_assign_kernel, grid, dim3(256), 0, st, lab, labels, npix, nseg, k,
                       (const float*)centres, assign, (const SegState*)seg, acc, first_round);
    int nsegs = B * nseg;
    hipLaunchKernelGGL(dcd_update_kernel, dim3(imgan::cdiv(nsegs, 128)), dim3(128), 0, st, nsegs, nseg,
                       k, npix, lab, centres, seg, oldcen, acc, draw, wanted, active_out);
    // both leave at once unless the update met a cluster that is empty for the first time
    dim3 gd((unsigned)std::min<long long>(imgan::cdiv(npix, 256), 8192), (unsigned)B);
    hipLaunchKernelGGL(dcd_draw_kernel, gd, dim3(256), 0, st, labels, npix, nseg, k, seed,
                       (const SegState*)seg, draw, (const int*)wanted);
    hipLaunchKernelGGL(dcd_draw_apply_kernel, dim3(imgan::cdiv(nsegs, 128)), dim3(128), 0, st, nsegs,
                       nseg, k, npix, lab, centres, seg, oldcen, (const unsigned long long*)draw,
                       (const int*)wanted, active_out);
    return imgan::check_launch("imgan_dcd_kmeans_round");
}

extern "C" int imgan_dcd_rounds(const void* state, int B, int nseg, int k, int32_t* rounds_out,
                                int32_t* done_out, imgan_stream_t stream) {
    IMGAN_REQUIRE(state && rounds_out && done_out, "imgan_dcd_rounds: null pointer");
    IMGAN_REQUIRE(B >= 1 && nseg >= 1 && k >= 1, "imgan_dcd_rounds: bad shape");
    DcdLayout L = dcd_layout(B, nseg, k);
    const char* segs = (const char*)state + L.off_seg;
    size_t n = (size_t)B * nseg;
    hipStream_t st = (hipStream_t)stream;
    IMGAN_HIP(hipMemcpy2DAsync(done_out, 4, segs + offsetof(SegState, done), sizeof(SegState), 4, n,
                               hipMemcpyDeviceToDevice, st));
    IMGAN_HIP(hipMemcpy2DAsync(rounds_out, 4, segs + offsetof(SegState, rounds), sizeof(SegState),
                               4, n, hipMemcpyDeviceToDevice, st));
    return 0;
}

extern "C" int imgan_dcd_stats(const uint8_t* lab, const int32_t* labels, const uint8_t* assign,
                               int B, int H, int W, int nseg, int k, uint64_t* stats,
                               int32_t* scratch, const void* state, imgan_stream_t stream) {
    IMGAN_REQUIRE(lab && labels && assign && stats && scratch, "imgan_dcd_stats: null pointer");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && nseg >= 1 && B <= 65535, "imgan_dcd_stats: bad shape");
    IMGAN_REQUIRE(k >= 1 && k <= KMAX, "imgan_dcd_stats: 1 <= k <= 16 dominant colours");
    IMGAN_REQUIRE((long long)H * W < 0x7fffffffLL && (long long)nseg * k < 0x7fffffffLL,
                  "imgan_dcd_stats: too large for int32 ids");
    long long npix = (long long)H * W;
    size_t plane = (size_t)B * npix;
    int32_t* values = scratch;
    int32_t* roots = scratch + plane;
    int32_t* sizes = scratch + 2 * plane;
    hipStream_t st = (hipStream_t)stream;
    const SegState* seg = state ? (const SegState*)((const char*)state + dcd_layout(B, nseg, k).off_seg)
                                : nullptr;
    int nsk = B * nseg * k;
    hipLaunchKernelGGL(dcd_zero_stats_kernel, dim3(imgan::cdiv(nsk, 256)), dim3(256), 0, st,
                       (unsigned long long*)stats, B * nseg, k, seg);
    dim3 grid8((unsigned)imgan::cdiv(npix, 256 * 8), (unsigned)B);
    hipLaunchKernelGGL(dcd_values_kernel, grid8, dim3(256), 0, st, lab, labels, assign, npix, nseg,
                       k, values, sizes, (unsigned long long*)stats, seg);
    if (int rc = imgan_cc_label(values, roots, B, H, W, stream)) return rc;
    dim3 grid((unsigned)std::min<long long>(imgan::cdiv(npix, 256), 16384), (unsigned)B);
    hipLaunchKernelGGL(dcd_sizes_kernel, grid, dim3(256), 0, st, (const int32_t*)roots, sizes, npix);
    hipLaunchKernelGGL(dcd_largest_kernel, grid, dim3(256), 0, st, (const int32_t*)values,
                       (const int32_t*)roots, (const int32_t*)sizes, npix, nseg * k,
                       (unsigned long long*)stats);
    return imgan::check_launch("imgan_dcd_stats");
}

extern "C" int imgan_dcd_init_hash(const uint8_t* lab, const int32_t* labels, int B, int H, int W,
                                   int nseg, int k, uint32_t seed, float* centres, uint64_t* slots,
                                   const void* state, imgan_stream_t stream) {
    IMGAN_REQUIRE(lab && labels && centres && slots, "imgan_dcd_init_hash: null pointer");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && nseg >= 1 && B <= 65535, "imgan_dcd_init_hash: bad shape");
    IMGAN_REQUIRE(k >= 1 && k <= KMAX, "imgan_dcd_init_hash: 1 <= k <= 16 dominant colours");
    IMGAN_REQUIRE((long long)H * W < 0x7fffffffLL, "imgan_dcd_init_hash: image too large");
    hipStream_t st = (hipStream_t)stream;
    long long npix = (long long)H * W;
    IMGAN_HIP(hipMemsetAsync(slots, 0xff, (size_t)B * nseg * k * sizeof(uint64_t), st));
    dim3 grid((unsigned)std::min<long long>(imgan::cdiv(npix, 256), 8192), (unsigned)B);
    const SegState* seg = state ? (const SegState*)((const char*)state + dcd_layout(B, nseg, k).off_seg)
                                : nullptr;
    hipLaunchKernelGGL(dcd_init_pick_kernel, grid, dim3(256), 0, st, labels, npix, nseg, k, seed,
                       (unsigned long long*)slots, seg);
    int nsegs = B * nseg;
    hipLaunchKernelGGL(dcd_init_fetch_kernel, dim3(imgan::cdiv(nsegs, 128)), dim3(128), 0, st, lab,
                       npix, nsegs, nseg, k, (const unsigned long long*)slots, centres, seg);
    return imgan::check_launch("imgan_dcd_init_hash");
}

extern "C" int imgan_dcd_features(const uint64_t* stats, const float* centres, int B, int nseg,
                                  int k, float* features, imgan_stream_t stream) {
    IMGAN_REQUIRE(stats && centres && features, "imgan_dcd_features: null pointer");
    IMGAN_REQUIRE(B >= 1 && nseg >= 1 && k >= 1 && k <= KMAX, "imgan_dcd_features: bad shape");
    IMGAN_REQUIRE(((uintptr_t)features & 15) == 0, "imgan_dcd_features: features must be 16-byte aligned");
    int nsegs = B * nseg;
    hipLaunchKernelGGL(dcd_features_kernel, dim3((unsigned)imgan::cdiv((long long)nsegs * k, 256)),
                       dim3(256), 0,
                       (hipStream_t)stream, (const unsigned long long*)stats, centres, nsegs, k,
                       features);
    return imgan::check_launch("dcd_features_kernel");
}
