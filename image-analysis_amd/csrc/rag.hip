// Region adjacency graph: per-label colour sums + 8-connected adjacency (device), the
// MST merge (host) and the relabel / paint passes (device).
// Replaces scripts/rag_segmentation.py:27-48 (means), :50-74 (neighbours_regions),
// :77-184 (find_edges .. concat_similar_regs).
//
// Device pass: one sweep over labels (4 B/px) + image (3 B/px).  Sums are aggregated per
// wave before touching memory (labels are spatially coherent, so a wave sees a handful of
// distinct labels): one 64-bit atomic per (wave, label, field).  Adjacent label pairs are
// only produced on region borders (a few % of the pixels) and go into a per-image
// open-addressed hash set with a read-before-CAS probe, so the common "already present"
// case costs one cached load.
#include "common.h"
#include "../../include/imgan.h"
#include <algorithm>
#include <math.h>
#include "cut_rule.h"
#include <atomic>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

namespace {

// Sum over the 64 lanes with data-parallel-primitive adds (one vector instruction per step, no
// LDS traffic): four shifts inside the rows of 16 lanes, then lane 15 of rows 0 / 2 into rows
// 1 / 3 and lane 31 into rows 2 / 3.  The total is lane 63's value, returned to every lane.
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);   // row_bcast:15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);   // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t hash64(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (uint32_t)k;
}

__device__ __forceinline__ unsigned long long edge_key(int a, int b) {
    const int u = min(a, b), v = max(a, b);
    return ((unsigned long long)(u + 1) << 32) | (unsigned long long)(v + 1);
}

__device__ void edge_insert(unsigned long long* tab, unsigned long long mask, unsigned long long key,
                            int* overflow) {
    unsigned long long h = hash64(key) & mask;
    for (unsigned long long probe = 0; probe <= mask; ++probe) {
        unsigned long long cur = tab[h];
        if (cur == key) return;
        if (cur == 0ull) {
            unsigned long long old = atomicCAS(&tab[h], 0ull, key);
            if (old == 0ull || old == key) return;
        }
        h = (h + 1) & mask;
    }
    *overflow = 1;
}

typedef uint32_t __attribute__((aligned(1))) u32_any;   // one global_load_dword at any address
constexpr int RAG_ROWS = 16;
constexpr int RAG_PEND = 512;      // pairs a block parks in LDS before they go to the table

// one lane per column; block = 256 columns x RAG_ROWS rows (grid: column strips, row strips, images).
// SUMS == false: adjacency only, the image is not read (the caller has the per-label sums already,
// e.g. from the last SLIC round: imgan_slic_label_sums).
template <bool SUMS>
__global__ __launch_bounds__(256) void rag_accumulate_kernel(
    const uint8_t* __restrict__ img, const int32_t* __restrict__ labels, int H, int W, int K,
    unsigned long long* __restrict__ sums, unsigned long long* __restrict__ edges,
    unsigned long long edge_mask, int* __restrict__ overflow, long long last_px) {
    // The pairs a block meets are parked in LDS and go to the table together when its rows are
    // done: looked up where they are met, the one lane with a new pair holds its wave's row loop
    // up for a round trip to the table in global memory.
    __shared__ unsigned long long pend[RAG_PEND];
    __shared__ int npend;
    const int b = blockIdx.z;
    if (threadIdx.x == 0) npend = 0;
    __syncthreads();
    const long long npix = (long long)H * W;
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int32_t* lab = labels + (size_t)b * npix;
    const uint8_t* im = img + (size_t)b * npix * 3;
    unsigned long long* S = sums + (size_t)b * K * 4;
    unsigned long long* E = edges ? edges + (size_t)b * (edge_mask + 1) : nullptr;
    // RAG_ROWS rows per block: one row per block leaves the launch bound by the workgroup
    // dispatch rate (a row of 256 pixels is a few hundred cycles of work).  A lane keeps the sums
    // of the label it is walking down in registers; when lanes change label (typically a whole
    // run of lanes crossing a region border together) the old label's sums of those lanes are
    // added up over the wave and leave with one atomic per field, as do all sums at the end.
    int cur = -1;
    uint32_t a0 = 0, a1 = 0, a2 = 0, an = 0;
    auto park = [&](int u, int v) {
        const unsigned long long key = edge_key(u, v);
        const int at = atomicAdd(&npend, 1);
        if (at < RAG_PEND)
            pend[at] = key;
        else
            edge_insert(E, edge_mask, key, overflow + b);
    };
    auto flush = [&](bool who) {
        int pending = (who && cur >= 0 && cur < K) ? cur : -1;
        while (true) {
            const unsigned long long act = __ballot(pending >= 0);
            if (act == 0ull) break;
            const int c = __builtin_amdgcn_readlane(pending, __ffsll((long long)act) - 1);
            const bool mine = (pending == c);
            // at most 64 lanes x RAG_ROWS rows x 255: far below 2^32
            const uint32_t t0 = wave_sum(mine ? a0 : 0u), t1 = wave_sum(mine ? a1 : 0u);
            const uint32_t t2 = wave_sum(mine ? a2 : 0u), tn = wave_sum(mine ? an : 0u);
            if ((threadIdx.x & 63) == 0) {
                unsigned long long* s = S + (size_t)c * 4;
                atomicAdd(&s[0], (unsigned long long)t0);
                atomicAdd(&s[1], (unsigned long long)t1);
                atomicAdd(&s[2], (unsigned long long)t2);
                atomicAdd(&s[3], (unsigned long long)tn);
            }
            if (mine) pending = -1;
        }
    };
    const int yend = min((int)(blockIdx.y + 1) * RAG_ROWS, H);
    const int lane = threadIdx.x & 63;
    // adjacency state: the row above (own column and the columns left / right of it); a lane takes
    // its neighbours' labels from the lanes beside it, only the two lanes at the wave's ends load
    // a label of their own for that
    int pc = -1, pcl = -1, pcr = -1;
    // what a lane reads from memory for row y: its label, its pixel, and (lanes 0 / 63 only) the
    // label left / right of the wave's 64 columns
    struct RowIn {
        int c, edge;
        uint32_t raw;
    };
    auto fetch = [&](int y, bool pixel) {
        RowIn r;
        const long long q = (long long)y * W + x;
        r.c = x < W ? lab[q] : -1;
        r.edge = -1;
        r.raw = 0;
        if (E) {
            if (lane == 0 && x > 0 && x <= W) r.edge = lab[q - 1];
            if (lane == 63 && x + 1 < W) r.edge = lab[q + 1];
        }
        if (pixel && x < W) {
            // the three bytes with one (unaligned) dword load; the very last pixel of the batch is
            // read one byte early so that the load stays inside the buffer
            const uint32_t sh = ((long long)b * npix + q == last_px) ? 1u : 0u;
            r.raw = *reinterpret_cast<const u32_any*>(im + q * 3 - sh) >> (8u * sh);
        }
        return r;
    };
    auto sides = [&](const RowIn& r, int& cl, int& cr) {
        cl = __shfl_up(r.c, 1);
        cr = __shfl_down(r.c, 1);
        if (lane == 0) cl = r.edge;
        if (lane == 63) cr = r.edge;
    };
    const int ystart = blockIdx.y * RAG_ROWS;
    if (E && ystart > 0) {
        const RowIn above = fetch(ystart - 1, false);
        pc = above.c;
        sides(above, pcl, pcr);
    }
    auto process = [&](const RowIn& here) {
    const int l = here.c;
    const uint32_t raw = here.raw;
    int cl = -1, cr = -1;
    if (E) sides(here, cl, cr);
    if (SUMS) {
        const bool chg = (l != cur);
        if (__ballot(chg) != 0ull) {                       // wave uniform
            flush(chg);
            if (chg) {
                cur = l;
                a0 = a1 = a2 = an = 0;
            }
        }
        if (l >= 0) {
            a0 += raw & 0xffu;
            a1 += (raw >> 8) & 0xffu;
            a2 += (raw >> 16) & 0xffu;
            an += 1u;
        }
    }
    // ---- adjacency: W, N, NW, NE neighbours (each unordered 8-neighbour pair once).  A border
    // runs on for many pixels: a pair this lane met in the row above (vertical border) or the lane
    // to the left meets in this row (horizontal border) is not looked up again.
    if (E) {
        if (l >= 0) {
            const bool w_ = cl >= 0 && cl != l && !(l == pc && cl == pcl);
            if (w_) park(l, cl);
            const bool n_ = pc >= 0 && pc != l && pc != cl && !(lane > 0 && cl == l && pcl == pc);
            if (n_) park(l, pc);
            if (pcl >= 0 && pcl != l && pcl != cl && pcl != pc) park(l, pcl);
            if (pcr >= 0 && pcr != l && pcr != cl && pcr != pc && pcr != pcl)
                park(l, pcr);
        }
        pc = l, pcl = cl, pcr = cr;
    }
    };
    // four rows per step, one step ahead: the loads of rows y + 4 .. y + 7 are in flight while rows
    // y .. y + 3 are worked on (one row ahead left a block waiting for memory sixteen times)
    constexpr int RC = 4;
    static_assert(RAG_ROWS % RC == 0, "rows per block in whole steps");
    RowIn ahead[RC];
#pragma unroll
    for (int u = 0; u < RC; ++u) ahead[u] = fetch(min(ystart + u, yend - 1), SUMS);
    for (int y = ystart; y < yend; y += RC) {
        RowIn now[RC];
#pragma unroll
        for (int u = 0; u < RC; ++u) now[u] = ahead[u];
        if (y + RC < yend) {
#pragma unroll
            for (int u = 0; u < RC; ++u) ahead[u] = fetch(min(y + RC + u, yend - 1), SUMS);
        }
#pragma unroll
        for (int u = 0; u < RC; ++u)
            if (y + u < yend) process(now[u]);
    }
    if (SUMS) flush(true);
    if (E) {
        __syncthreads();
        const int np = min(npend, RAG_PEND);
        for (int i = threadIdx.x; i < np; i += 256) edge_insert(E, edge_mask, pend[i], overflow + b);
    }
}

__global__ __launch_bounds__(256) void relabel_kernel(const int32_t* __restrict__ labels,
                                                      const int32_t* __restrict__ lut,
                                                      int32_t* __restrict__ out, long long npix,
                                                      int K) {
    const int b = blockIdx.y;
    const int32_t* L = lut + (size_t)b * K;
    const int32_t* src = labels + (size_t)b * npix;
    int32_t* dst = out + (size_t)b * npix;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix;
         p += (long long)gridDim.x * 256) {
        int l = src[p];
        dst[p] = (l >= 0 && l < K) ? L[l] : 0;
    }
}

// Four labels per lane and step (16-byte loads and stores), the table in LDS.  Needs npix % 4 == 0
// (every image of the batch then starts on a 16-byte boundary if the first one does) and
// K <= RELABEL_LDS_K.
constexpr int RELABEL_LDS_K = 8192;
__global__ __launch_bounds__(256) void relabel4_kernel(const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ lut,
                                                       int32_t* __restrict__ out, long long nquad,
                                                       int K) {
    __shared__ int32_t T[RELABEL_LDS_K];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < K; i += 256) T[i] = lut[(size_t)b * K + i];
    __syncthreads();
    const int4* src = reinterpret_cast<const int4*>(labels) + (size_t)b * nquad;
    int4* dst = reinterpret_cast<int4*>(out) + (size_t)b * nquad;
    auto map = [&](int l) { return ((unsigned)l < (unsigned)K) ? T[l] : 0; };
    constexpr int U = 2;                                 // loads in flight per lane
    const long long stride = (long long)gridDim.x * 256;
    for (long long q0 = (long long)blockIdx.x * 256 + threadIdx.x; q0 < nquad; q0 += stride * U) {
        int4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long q = q0 + u * stride;
            v[u] = src[q < nquad ? q : nquad - 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long q = q0 + u * stride;
            if (q < nquad) dst[q] = make_int4(map(v[u].x), map(v[u].y), map(v[u].z), map(v[u].w));
        }
    }
}

__global__ __launch_bounds__(256) void paint_kernel(uint8_t* __restrict__ img,
                                                    const int32_t* __restrict__ labels,
                                                    const uint8_t* __restrict__ means,
                                                    long long npix, int K) {
    const int b = blockIdx.y;
    const uint8_t* M = means + (size_t)b * K * 3;
    const int32_t* src = labels + (size_t)b * npix;
    uint8_t* dst = img + (size_t)b * npix * 3;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix;
         p += (long long)gridDim.x * 256) {
        int l = src[p];
        if (l >= 0 && l < K) {
            dst[3 * p] = M[3 * l];
            dst[3 * p + 1] = M[3 * l + 1];
            dst[3 * p + 2] = M[3 * l + 2];
        }
    }
}

// ---------------------------------------------------------------- host merge
struct Sets {
    // list-of-sets semantics of rag_segmentation.py:117-130: union(v1, v2) appends the set
    // of v2 to the set of v1 and removes v2's entry; a set keeps the list slot ("anchor")
    // of the set it was merged into.
    std::vector<int> parent, anchor, head, tail, next;
    int alive;
    explicit Sets(int K) : parent(K), anchor(K), head(K), tail(K), next(K, -1), alive(K) {
        std::iota(parent.begin(), parent.end(), 0);
        std::iota(anchor.begin(), anchor.end(), 0);
        std::iota(head.begin(), head.end(), 0);
        std::iota(tail.begin(), tail.end(), 0);
    }
    int find(int v) {
        int r = v;
        while (parent[r] != r) r = parent[r];
        while (parent[v] != r) {
            int nx = parent[v];
            parent[v] = r;
            v = nx;
        }
        return r;
    }
    bool unite(int v1, int v2) {
        int r1 = find(v1), r2 = find(v2);
        if (r1 == r2) return false;
        parent[r2] = r1;                 // r1 keeps its anchor and member order
        next[tail[r1]] = head[r2];
        tail[r1] = tail[r2];
        --alive;
        return true;
    }
};

}  // namespace

extern "C" int imgan_rag_accumulate(const uint8_t* img, const int32_t* labels, int B, int H, int W,
                                    int K, uint64_t* sums, uint64_t* edges, int64_t edge_cap,
                                    int32_t* overflow, imgan_stream_t stream) {
    IMGAN_REQUIRE(labels && ((img && sums) || (!img && !sums && edges)),
                  "imgan_rag_accumulate: null pointer (image and sums go together; without them the edge table is required)");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && K >= 1, "imgan_rag_accumulate: bad shape");
    IMGAN_REQUIRE(B <= 65535, "imgan_rag_accumulate: batch > 65535, split the call");
    if (edges) {
        IMGAN_REQUIRE(edge_cap >= 2 && (edge_cap & (edge_cap - 1)) == 0,
                      "imgan_rag_accumulate: edge_cap must be a power of two");
        IMGAN_REQUIRE(overflow != nullptr, "imgan_rag_accumulate: overflow flags required");
    }
    IMGAN_REQUIRE(H <= 65535 * RAG_ROWS, "imgan_rag_accumulate: too many rows");
    IMGAN_REQUIRE((long long)B * H * W >= 2, "imgan_rag_accumulate: needs at least two pixels");
    dim3 grid((unsigned)imgan::cdiv(W, 256), (unsigned)imgan::cdiv(H, RAG_ROWS), (unsigned)B);
    if (img)
        hipLaunchKernelGGL(rag_accumulate_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, img,
                           labels, H, W, K, (unsigned long long*)sums, (unsigned long long*)edges,
                           (unsigned long long)(edges ? edge_cap - 1 : 0), overflow,
                           (long long)B * H * W - 1);
    else
        hipLaunchKernelGGL(rag_accumulate_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, img,
                           labels, H, W, K, (unsigned long long*)sums, (unsigned long long*)edges,
                           (unsigned long long)(edge_cap - 1), overflow, (long long)B * H * W - 1);
    return imgan::check_launch("rag_accumulate_kernel");
}

extern "C" int imgan_relabel_lut(const int32_t* labels, const int32_t* lut, int32_t* out, int B,
                                 int64_t npix, int K, imgan_stream_t stream) {
    IMGAN_REQUIRE(labels && lut && out, "imgan_relabel_lut: null pointer");
    IMGAN_REQUIRE(B >= 1 && npix >= 1 && K >= 1 && B <= 65535, "imgan_relabel_lut: bad shape");
    if (npix % 4 == 0 && K <= RELABEL_LDS_K && ((uintptr_t)labels & 15) == 0 && ((uintptr_t)out & 15) == 0) {
        const int64_t nquad = npix / 4;
        // enough blocks per image to fill the chip, few enough that the table load stays small
        // beside the 16 KB of labels a block of 256 lanes moves per step
        dim3 grid4((unsigned)std::min<int64_t>(imgan::cdiv(nquad, 256 * 8), 4096), (unsigned)B);
        hipLaunchKernelGGL(relabel4_kernel, grid4, dim3(256), 0, (hipStream_t)stream, labels, lut, out,
                           (long long)nquad, K);
        return imgan::check_launch("relabel4_kernel");
    }
    dim3 grid((unsigned)std::min<int64_t>(imgan::cdiv(npix, 256), 8192), (unsigned)B);
    hipLaunchKernelGGL(relabel_kernel, grid, dim3(256), 0, (hipStream_t)stream, labels, lut, out,
                       (long long)npix, K);
    return imgan::check_launch("relabel_kernel");
}

extern "C" int imgan_paint_means(uint8_t* img, const int32_t* labels, const uint8_t* means_u8,
                                 int B, int64_t npix, int K, imgan_stream_t stream) {
    IMGAN_REQUIRE(img && labels && means_u8, "imgan_paint_means: null pointer");
    IMGAN_REQUIRE(B >= 1 && npix >= 1 && K >= 1 && B <= 65535, "imgan_paint_means: bad shape");
    dim3 grid((unsigned)std::min<int64_t>(imgan::cdiv(npix, 256), 8192), (unsigned)B);
    hipLaunchKernelGGL(paint_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, labels,
                       means_u8, (long long)npix, K);
    return imgan::check_launch("paint_kernel");
}

extern "C" int imgan_rag_merge_host(const int32_t* means, const double* weights, int K,
                                    const int32_t* edges_uv, int64_t E, double c_factor,
                                    int32_t* lut_out, int32_t* mst_out, double* mst_w_out,
                                    int32_t* n_mst_out, int32_t* members_out,
                                    int32_t* set_offsets_out, int32_t* n_sets_out) {
    IMGAN_REQUIRE(lut_out && n_mst_out && n_sets_out, "imgan_rag_merge_host: null pointer");
    IMGAN_REQUIRE((means != nullptr) != (weights != nullptr),
                  "imgan_rag_merge_host: give either means or weights");
    IMGAN_REQUIRE(K >= 1 && E >= 0 && (E == 0 || edges_uv), "imgan_rag_merge_host: bad sizes");
    struct Edge {
        int u, v;
        long long w2;      // exact squared weight (means mode)
        double w;          // weight as the reference sees it
    };
    std::vector<Edge> ed;
    ed.reserve((size_t)E);
    for (int64_t i = 0; i < E; ++i) {
        int u = edges_uv[2 * i], v = edges_uv[2 * i + 1];
        if (u < 0 || v < 0 || u >= K || v >= K || u == v)
            return imgan::fail(imgan::E_INVALID, "imgan_rag_merge_host: edge endpoint out of range");
        if (means) {
            if (u > v) std::swap(u, v);
            long long d0 = means[3 * u] - means[3 * v], d1 = means[3 * u + 1] - means[3 * v + 1],
                      d2 = means[3 * u + 2] - means[3 * v + 2];
            long long w2 = d0 * d0 + d1 * d1 + d2 * d2;          // metric_methods.py:5-6, squared
            ed.push_back({u, v, w2, sqrt((double)w2)});
        } else {
            ed.push_back({u, v, 0, weights[i]});
        }
    }
    if (means) {
        // canonical discovery order (:93-99 with ascending dict / set order): lexicographic (u,v)
        std::sort(ed.begin(), ed.end(), [](const Edge& a, const Edge& b) {
            return a.u != b.u ? a.u < b.u : a.v < b.v;
        });
        ed.erase(std::unique(ed.begin(), ed.end(),
                             [](const Edge& a, const Edge& b) { return a.u == b.u && a.v == b.v; }),
                 ed.end());
        // :103-114 stable insertion sort by weight (sqrt is monotone: sort by the exact square)
        std::stable_sort(ed.begin(), ed.end(),
                         [](const Edge& a, const Edge& b) { return a.w2 < b.w2; });
    } else {
        // weights mode: edges arrive in the caller's discovery order, already de-duplicated
        std::stable_sort(ed.begin(), ed.end(), [](const Edge& a, const Edge& b) { return a.w < b.w; });
    }

    // :132-145 first Kruskal
    std::vector<Edge> mst;
    {
        Sets s(K);
        for (size_t i = 0; i < ed.size() && s.alive > 1; ++i)
            if (s.unite(ed[i].u, ed[i].v)) mst.push_back(ed[i]);
    }
    // :163-170 keep edge i iff w_i < mean(others) + c * std(others), population std
    std::vector<Edge> kept;
    const size_t m = mst.size();
    if (m >= 2) {
        // both sums over the rounded float64 weights, in double-double (cut_rule.h: the rule the
        // device merge applies too)
        imgan_cut::DD T{0.0, 0.0}, Q{0.0, 0.0};
        for (const Edge& e : mst) {
            T = imgan_cut::dd_add(T, imgan_cut::DD{e.w, 0.0});
            Q = imgan_cut::dd_add(Q, imgan_cut::dd_square(e.w));
        }
        for (const Edge& e : mst)
            if (imgan_cut::keep_edge(e.w, T, Q, (int)m, c_factor)) kept.push_back(e);
    }   // m == 1: mean of an empty list is NaN -> the edge is dropped (:166-168)

    // :173 second Kruskal from singletons, :175-182 ids in surviving-list order
    Sets s(K);
    int n_mst = 0;
    for (const Edge& e : kept)
        if (s.alive > 1 && s.unite(e.u, e.v)) {
            if (mst_out) {
                mst_out[2 * n_mst] = e.u;
                mst_out[2 * n_mst + 1] = e.v;
            }
            if (mst_w_out) mst_w_out[n_mst] = e.w;
            ++n_mst;
        }
    *n_mst_out = n_mst;
    int id = 0, pos = 0;
    for (int v = 0; v < K; ++v) {
        if (s.find(v) != v) continue;          // surviving sets sit at their anchor's slot
        ++id;
        if (set_offsets_out) set_offsets_out[id - 1] = pos;
        for (int e = s.head[v]; e >= 0; e = s.next[e]) {
            lut_out[e] = id;
            if (members_out) members_out[pos] = e;
            ++pos;
        }
    }
    if (set_offsets_out) set_offsets_out[id] = pos;
    *n_sets_out = id;
    return 0;
}

// The merge of every image of a batch (means mode, relabel table only), on `threads` host
// threads that take images from a shared counter: a merge is a few hundred microseconds of
// sorting and union-find, so a batch of 4K frames is bound by this loop when it runs serially.
extern "C" int imgan_rag_merge_host_batch(const int32_t* means, int B, int K,
                                          const int32_t* edges_uv, const int64_t* edge_offsets,
                                          double c_factor, int32_t* lut_out, int threads) {
    IMGAN_REQUIRE(means && edge_offsets && lut_out, "imgan_rag_merge_host_batch: null pointer");
    IMGAN_REQUIRE(B >= 1 && K >= 1, "imgan_rag_merge_host_batch: bad sizes");
    for (int b = 0; b < B; ++b)
        IMGAN_REQUIRE(edge_offsets[b] >= 0 && edge_offsets[b] <= edge_offsets[b + 1],
                      "imgan_rag_merge_host_batch: edge offsets must ascend");
    IMGAN_REQUIRE(edge_offsets[B] == 0 || edges_uv, "imgan_rag_merge_host_batch: null edge list");
    const int nt = std::max(1, std::min(std::min(threads, B), 64));
    std::atomic<int> next(0), rc_all(0);
    std::string first_error;
    std::atomic<bool> have_error(false);
    auto work = [&]() {
        for (int b = next.fetch_add(1); b < B; b = next.fetch_add(1)) {
            int32_t n_mst = 0, n_sets = 0;
            const int64_t lo = edge_offsets[b], E = edge_offsets[b + 1] - lo;
            const int rc = imgan_rag_merge_host(means + (size_t)b * K * 3, nullptr, K,
                                                E ? edges_uv + 2 * lo : nullptr, E, c_factor,
                                                lut_out + (size_t)b * K, nullptr, nullptr, &n_mst,
                                                nullptr, nullptr, &n_sets);
            if (rc != 0 && !have_error.exchange(true)) {     // the message is per thread: keep one
                rc_all = rc;
                first_error = imgan_last_error();
            }
        }
    };
    if (nt == 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back(work);
        for (auto& th : pool) th.join();
    }
    if (have_error) return imgan::fail(rc_all, first_error);
    return 0;
}
