// The merge of rag_segmentation.py:77-184 (find_edges, __convert2arrays, __sort_edges, kruskal_alg,
// the leave-one-out cut of concat_similar_regs, the second kruskal_alg and the new ids) on the
// device, one workgroup per image, everything in LDS: the batch path never leaves the GPU.
//
//   1. floor means from the per-label sums (rag_segmentation.py:38-48)
//   2. the image's edge hash set -> a list of keys (w2 << 24 | u << 12 | v), w2 = the exact squared
//      colour distance (metric_methods.py:5-6): sorting the keys is the reference's stable sort by
//      weight of the canonical (u < v, lexicographic) edge list
//   3. bitonic sort in LDS (<= 16,384 keys)
//   4. the minimum spanning forest by Boruvka rounds.  With a strict total order on the edges (the
//      rank after sorting) the forest is unique, so it is the one Kruskal's scan finds (:132-145)
//   5. sum and sum of squares of the forest's weights, then the cut (:163-170) for every edge at
//      once: keep iff w < mean(others) + c * std(others), in double-double arithmetic
//   6. the second Kruskal pass (:173) replayed in rank order: a set keeps the list slot of the
//      first endpoint's set (:124-130), which is what numbers the merged regions.  The order only
//      matters inside one final region, so the regions are replayed side by side, one wave each
//   7. ids = 1-based rank of a set's slot (:175-182), written as the relabel table
#include "common.h"
#include "../../include/imgan.h"
#include <math.h>
#include "cut_rule.h"

namespace {

constexpr int MG_THREADS = 1024, MG_MAXK = 4096, MG_MAXE = 16384;
constexpr uint32_t MG_NONE = 0xffffffffu;

using imgan_cut::DD;
using imgan_cut::dd_add;

__device__ __forceinline__ int w2_of(uint32_t mu, uint32_t mv) {
    const int d0 = (int)(mu & 255u) - (int)(mv & 255u);
    const int d1 = (int)((mu >> 8) & 255u) - (int)((mv >> 8) & 255u);
    const int d2 = (int)((mu >> 16) & 255u) - (int)((mv >> 16) & 255u);
    return d0 * d0 + d1 * d1 + d2 * d2;
}

// block-wide exclusive prefix sum of one int per thread (1024 threads); total returned to all
__device__ int block_scan(int v, int* wave_tot, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    __syncthreads();                                       // wave_tot may still be read by a previous use
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < MG_THREADS / 64; ++w) {
        const int t = wave_tot[w];
        if (w < wave) base += t;
        tot += t;
    }
    total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(MG_THREADS) void rag_merge_kernel(
    const unsigned long long* __restrict__ sums, const unsigned long long* __restrict__ edges,
    long long cap, const int* __restrict__ overflow, int K, double c_factor,
    int32_t* __restrict__ lut_out, int32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long smem[];
    // layout (bytes): keys u64 [MG_MAXE] | after the sort its first half holds edge u32 [MG_MAXE]
    // and its second half comp / parent / best u32 [MG_MAXK] each + state u8 [MG_MAXE];
    // then means u32 [MG_MAXK], and a few words of shared scalars
    unsigned long long* keys = smem;
    uint32_t* edge = reinterpret_cast<uint32_t*>(smem);
    uint32_t* comp = edge + MG_MAXE;
    uint32_t* parent = comp + MG_MAXK;
    uint32_t* best = parent + MG_MAXK;
    uint8_t* state = reinterpret_cast<uint8_t*>(best + MG_MAXK);
    uint32_t* meanpk = reinterpret_cast<uint32_t*>(smem + MG_MAXE);
    int* sh = reinterpret_cast<int*>(meanpk + MG_MAXK);     // [0] nE, [1] changed, [2] bad, [3..] scan
    double* shd = reinterpret_cast<double*>(sh + 32);      // reduction scratch, 16-byte aligned
    const int b = blockIdx.x, tid = threadIdx.x;
    const unsigned long long* S = sums + (size_t)b * K * 4;
    const unsigned long long* tab = edges + (size_t)b * cap;
    int32_t* lut = lut_out + (size_t)b * K;

    // ---- 1. floor means (empty label -> 0)
    for (int v = tid; v < K; v += MG_THREADS) {
        const unsigned long long n = S[4 * v + 3];
        uint32_t m = 0;
        if (n) m = (uint32_t)(S[4 * v] / n) | ((uint32_t)(S[4 * v + 1] / n) << 8) | ((uint32_t)(S[4 * v + 2] / n) << 16);
        meanpk[v] = m;
    }
    if (tid < 3) sh[tid] = 0;
    __syncthreads();

    // ---- 2. edge keys
    constexpr int KU = 8;                                  // table words in flight per thread
    for (long long i0 = 0; i0 < cap; i0 += (long long)MG_THREADS * KU) {
        unsigned long long got[KU];
#pragma unroll
        for (int q = 0; q < KU; ++q) {
            const long long i = i0 + (long long)q * MG_THREADS + tid;
            got[q] = i < cap ? tab[i] : 0ull;
        }
#pragma unroll
        for (int q = 0; q < KU; ++q) {
            const unsigned long long key = got[q];
            if (key == 0ull) continue;
            const uint32_t u = (uint32_t)(key >> 32) - 1u, v = (uint32_t)key - 1u;
            if (u >= v || v >= (uint32_t)K) {
                sh[2] = 1;                                 // not an edge of labels 0..K-1
                continue;
            }
            const int pos = atomicAdd(&sh[0], 1);
            if (pos < MG_MAXE)
                keys[pos] = ((unsigned long long)w2_of(meanpk[u], meanpk[v]) << 24) | ((unsigned long long)u << 12) | v;
        }
    }
    __syncthreads();
    const int E = sh[0];
    if (E > MG_MAXE || sh[2] || overflow[b]) {             // the caller finishes this image on the host
        if (tid == 0) status[b] = E > MG_MAXE ? 1 : (sh[2] ? 2 : 3);
        return;
    }
    int EP = 2;
    while (EP < E) EP <<= 1;
    for (int i = E + tid; i < EP; i += MG_THREADS) keys[i] = ~0ull;
    __syncthreads();

    // ---- 3. bitonic sort, ascending
    for (int k = 2; k <= EP; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < EP / 2; t += MG_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
                const unsigned long long x = keys[i], y = keys[p];
                if ((x > y) == ((i & k) == 0)) {
                    keys[i] = y;
                    keys[p] = x;
                }
            }
            __syncthreads();
        }

    // ---- keys -> (u, v) words in place (all keys are read before any word is written)
    {
        unsigned long long mine[MG_MAXE / MG_THREADS];
#pragma unroll
        for (int q = 0; q < MG_MAXE / MG_THREADS; ++q) {
            const int e = tid + q * MG_THREADS;
            mine[q] = e < E ? keys[e] : 0ull;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MG_MAXE / MG_THREADS; ++q) {
            const int e = tid + q * MG_THREADS;
            if (e < E) {
                edge[e] = (uint32_t)(((mine[q] >> 12) & 4095u) << 16) | (uint32_t)(mine[q] & 4095u);
                state[e] = 0;                              // 0 open, 1 in the forest, 2 dropped, 3 kept
            }
        }
        for (int v = tid; v < K; v += MG_THREADS) comp[v] = parent[v] = (uint32_t)v, best[v] = MG_NONE;
    }
    __syncthreads();

    // ---- 4. minimum spanning forest, Boruvka rounds (at most log2 K + 1 of them do work)
    for (int round = 0; round < 16; ++round) {
        if (tid == 0) sh[1] = 0;
        for (int e = tid; e < E; e += MG_THREADS) {
            if (state[e] != 0) continue;
            const uint32_t cu = comp[edge[e] >> 16], cv = comp[edge[e] & 0xffffu];
            if (cu == cv) {
                state[e] = 2;
            } else {
                atomicMin(&best[cu], (uint32_t)e);
                atomicMin(&best[cv], (uint32_t)e);
            }
        }
        __syncthreads();
        for (int c = tid; c < K; c += MG_THREADS) {
            if (comp[c] != (uint32_t)c || best[c] == MG_NONE) continue;
            const uint32_t e = best[c];
            state[e] = 1;
            const uint32_t cu = comp[edge[e] >> 16], cv = comp[edge[e] & 0xffffu];
            const uint32_t other = cu == (uint32_t)c ? cv : cu;
            // two components that choose the same edge would point at each other: the lower stays
            if (!(best[other] == e && (uint32_t)c < other)) parent[c] = other;
            sh[1] = 1;
        }
        __syncthreads();
        const int changed = sh[1];
        uint32_t root[MG_MAXK / MG_THREADS];
#pragma unroll
        for (int q = 0; q < MG_MAXK / MG_THREADS; ++q) {
            const int v = tid + q * MG_THREADS;
            uint32_t c = v < K ? comp[v] : 0u;
            if (v < K)
                while (parent[c] != c) c = parent[c];
            root[q] = c;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MG_MAXK / MG_THREADS; ++q) {
            const int v = tid + q * MG_THREADS;
            if (v < K) comp[v] = root[q], best[v] = MG_NONE;
        }
        __syncthreads();
        if (!changed) break;
    }

    // ---- 5. forest statistics and the cut
    DD t_loc{0.0, 0.0}, q_loc{0.0, 0.0};
    int m_loc = 0;
    for (int e = tid; e < E; e += MG_THREADS)
        if (state[e] == 1) {
            const int w2 = w2_of(meanpk[edge[e] >> 16], meanpk[edge[e] & 0xffffu]);
            const double w = sqrt((double)w2);
            t_loc = dd_add(t_loc, DD{w, 0.0});
            q_loc = dd_add(q_loc, imgan_cut::dd_square(w));    // of the rounded weight (cut_rule.h)
            ++m_loc;
        }
    // tree reduction over the workgroup: [hi | lo | q | m] x 1024 would not fit; fold through waves
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        DD o{__shfl_xor(t_loc.hi, off), __shfl_xor(t_loc.lo, off)};
        t_loc = dd_add(t_loc, o);
        DD oq{__shfl_xor(q_loc.hi, off), __shfl_xor(q_loc.lo, off)};
        q_loc = dd_add(q_loc, oq);
        m_loc += __shfl_xor(m_loc, off);
    }
    if ((tid & 63) == 0) {
        const int w = tid >> 6;
        shd[5 * w] = t_loc.hi, shd[5 * w + 1] = t_loc.lo;
        shd[5 * w + 2] = q_loc.hi, shd[5 * w + 3] = q_loc.lo;
        shd[5 * w + 4] = (double)m_loc;
    }
    __syncthreads();
    DD T{0.0, 0.0}, Q{0.0, 0.0};
    int m = 0;
    for (int w = 0; w < MG_THREADS / 64; ++w) {
        T = dd_add(T, DD{shd[5 * w], shd[5 * w + 1]});
        Q = dd_add(Q, DD{shd[5 * w + 2], shd[5 * w + 3]});
        m += (int)shd[5 * w + 4];
    }
    for (int e = tid; e < E; e += MG_THREADS) {
        if (state[e] != 1) continue;
        const int w2 = w2_of(meanpk[edge[e] >> 16], meanpk[edge[e] & 0xffffu]);
        const bool keep = imgan_cut::keep_edge(sqrt((double)w2), T, Q, m, c_factor);
        state[e] = keep ? 3 : 2;
    }
    __syncthreads();

    // ---- 6. kept edges in rank order -> best[] (the Boruvka arrays are free), then the replay
    constexpr int PER = MG_MAXE / MG_THREADS;
    int mine = 0;
    for (int q = 0; q < PER; ++q) {
        const int e = tid * PER + q;
        mine += (e < E && state[e] == 3) ? 1 : 0;
    }
    int nkept;
    int at = block_scan(mine, sh + 3, nkept);
    uint32_t* kept = best;                                 // nkept <= K - 1
    for (int q = 0; q < PER; ++q) {
        const int e = tid * PER + q;
        if (e < E && state[e] == 3) kept[at++] = edge[e];
    }
    __syncthreads();
    // union by size keeps the finds short; comp[] becomes the list slot a set sits in; the sizes
    // go where the edge words were (kept[] holds the ones still needed).
    // The replay is a chain of dependent LDS accesses per edge (0.28 us each, three quarters of
    // this kernel when one lane walked the whole list), but only the order of the edges inside
    // one final region matters: the regions -- the trees of the kept forest, found first by
    // lock-free hooking -- are dealt to the 16 waves, and every wave replays its regions' edges
    // in rank order on its own.
    uint32_t* size = edge;
    uint32_t* p2 = edge + MG_MAXK;                         // parent pointers of the kept forest
    for (int v = tid; v < K; v += MG_THREADS) comp[v] = parent[v] = p2[v] = (uint32_t)v, size[v] = 1u;
    __syncthreads();
    for (int i = tid; i < nkept; i += MG_THREADS) {
        uint32_t u = kept[i] >> 16, v = kept[i] & 0xffffu;
        while (true) {
            while (p2[u] != u) u = p2[u];
            while (p2[v] != v) v = p2[v];
            if (u == v) break;
            const uint32_t hi = max(u, v), lo = min(u, v);
            const uint32_t was = atomicMin(&p2[hi], lo);   // pointers only ever go down: no cycles
            if (was == hi) break;                          // hi was a root and hangs under lo now
            u = was, v = lo;                               // someone moved hi first: join that side too
        }
    }
    __syncthreads();
    for (int v = tid; v < K; v += MG_THREADS) {
        uint32_t r = (uint32_t)v;
        while (p2[r] != r) r = p2[r];
        p2[v] = r;                                         // a reader meets the old parent or the root
    }
    __syncthreads();
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int base = 0; base < nkept; base += 64) {
            const int i = base + lane;
            const uint32_t word = i < nkept ? kept[i] : 0u;
            const bool mine = i < nkept && (int)(p2[word >> 16] & (MG_THREADS / 64 - 1)) == wave;
            unsigned long long todo = __ballot(mine);
            while (todo) {                                 // every lane runs the same replay: uniform
                const int l = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)word, l);
                uint32_t u = w >> 16, v = w & 0xffffu;
                while (parent[u] != u) u = parent[u];
                while (parent[v] != v) v = parent[v];
                if (u == v) continue;                      // cannot happen in a forest
                const uint32_t slot = comp[u];             // the first endpoint's set keeps its slot
                const uint32_t su = size[u], sv = size[v];
                __builtin_amdgcn_wave_barrier();           // all reads of this edge before its writes
                if (lane == 0) {
                    if (su < sv) {
                        parent[u] = v;
                        size[v] = su + sv;
                        comp[v] = slot;
                    } else {
                        parent[v] = u;
                        size[u] = su + sv;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();

    // ---- 7. ids: 1-based rank of the slot among the surviving slots
    uint32_t* slot_of = best + 0;                          // kept[] is dead now
    uint8_t* is_slot = state;
    for (int v = tid; v < K; v += MG_THREADS) is_slot[v] = 0;
    __syncthreads();
    for (int v = tid; v < K; v += MG_THREADS) {
        uint32_t r = (uint32_t)v;
        while (parent[r] != r) r = parent[r];
        slot_of[v] = comp[r];
        if (r == (uint32_t)v) is_slot[comp[r]] = 1;
    }
    __syncthreads();
    constexpr int VPER = MG_MAXK / MG_THREADS;
    int cnt = 0;
    for (int q = 0; q < VPER; ++q) {
        const int v = tid * VPER + q;
        cnt += (v < K && is_slot[v]) ? 1 : 0;
    }
    int nsets;
    int id = block_scan(cnt, sh + 3, nsets);
    for (int q = 0; q < VPER; ++q) {
        const int v = tid * VPER + q;
        if (v < K && is_slot[v]) size[v] = (uint32_t)(++id);   // size[] reused: id of the set in slot v
    }
    __syncthreads();
    for (int v = tid; v < K; v += MG_THREADS) lut[v] = (int32_t)size[slot_of[v]];
    if (tid == 0) status[b] = 0;
}

constexpr size_t MG_LDS = (size_t)MG_MAXE * 8 + (size_t)MG_MAXK * 4 + 32 * 4 + 16 * 5 * 8;

}  // namespace

extern "C" int imgan_rag_merge_device(const uint64_t* sums, const uint64_t* edges, int64_t edge_cap,
                                      const int32_t* overflow, int B, int K, double c_factor,
                                      int32_t* lut_out, int32_t* status, imgan_stream_t stream) {
    IMGAN_REQUIRE(sums && edges && overflow && lut_out && status, "imgan_rag_merge_device: null pointer");
    IMGAN_REQUIRE(B >= 1 && K >= 1 && edge_cap >= 2, "imgan_rag_merge_device: bad sizes");
    if (K > MG_MAXK)
        return imgan::fail(imgan::E_RANGE, "imgan_rag_merge_device: more than 4096 labels per image; use "
                                           "imgan_rag_merge_host");
    // per device, so set on every call (a process may drive several GPUs)
    IMGAN_HIP(hipFuncSetAttribute((const void*)rag_merge_kernel,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)MG_LDS));
    hipLaunchKernelGGL(rag_merge_kernel, dim3((unsigned)B), dim3(MG_THREADS), MG_LDS, (hipStream_t)stream,
                       (const unsigned long long*)sums, (const unsigned long long*)edges,
                       (long long)edge_cap, overflow, K, c_factor, lut_out, status);
    return imgan::check_launch("rag_merge_kernel");
}
