// SLIC superpixels for gfx950 -- replaces SLICSuperPixel.run() (scripts/slic.py:35-127).
//
// Reference semantics kept:
//   * centre order column-major (ix*ny + iy); 3x3 gradient nudge with first-minimum wins;
//   * num_iter-1 rounds; distance reset every round, labels persist;
//   * colour distance to the IMAGE PIXEL at the centre position, not to the mean colour;
//   * windows [c-S, c+S) clamped, last row / column never labelled;
//   * ties go to the lowest centre index; centres are floor(sum / count), empty -> 0.
//
// Data flow per round (all state in the caller's workspace, nothing allocated here):
//   bin:     centres -> S x S cell lists (count / scan / fill) + colour at the centre position
//   assign:  one workgroup per 64 x 32 / 64 x 64 pixel tile.  The centres whose window can reach the
//            tile are gathered into LDS (sorted by index, so "first strictly smaller wins"
//            is exactly the reference's sequential sweep), every lane evaluates its 4 pixels
//            against them with an exact integer key
//                 key = dc^2 * S^2 + ds^2 * cw^2        (same order as slic.py:101-109)
//            and the per-label sums of slic.py:118-125 are accumulated in lane-private LDS
//            slots (conflict-free ds_add), reduced once per tile and flushed with one 64-bit
//            global atomic per (tile, centre, field).  Integer sums => bit-reproducible.
//   finalise: centre = floor(sum / count).
//
// HBM traffic per round: 3 B/px read (u8 Lab) + 4 B/px written (int32 labels).
#include "common.h"
#include "../../include/imgan.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>

namespace {

constexpr int TW = 64;      // tile width  (16 lanes x 4 pixels)
constexpr int TILE_H_MAX = 64;   // tallest tile: 2 bands x 4 waves x 8 rows
constexpr int NSUBMAX = 2;      // bands per tile (pixels of all bands prefetched)
constexpr int MAXC = 48;    // most centres per tile the LDS fast path can be sized for
constexpr int NFIELD = 6;   // sum L, a, b, x, y, count
constexpr int TILE_REC = 1 + MAXC;   // 32-byte records per tile: header {m} + sorted candidates

typedef int int8v __attribute__((ext_vector_type(8)));
#define AS_CONST __attribute__((address_space(4)))   // loads through it are scalar loads

struct Layout {
    int ncx, ncy, ncell;
    int nsub, tile_h, ntx, nty;  // assignment tiles: 64 x tile_h pixels (nsub bands)
    size_t off_count, off_start, off_fill, off_scand, off_cand, off_sums, off_tiles, total;
};

// A lane owns `rows` vertically adjacent pixels of its column: everything per candidate that is
// not per pixel (the entry, the column test, the x part of the key) is paid once for them.  A
// band is 4 waves x rows tall; taller tiles amortise the per-tile prologue / flush; small steps
// keep tiles short so that the number of centres reaching a tile stays below MAXC.
int choose_rows(int S) { return S >= 28 ? 8 : 4; }
int choose_nsub(int S) { return S >= 48 ? 2 : 1; }

Layout make_layout(int B, int H, int W, int S, int n) {
    Layout L;
    L.ncx = (W + S - 1) / S;
    L.ncy = (H + S - 1) / S;
    L.ncell = L.ncx * L.ncy;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o = imgan::align_up(o + bytes, 256);
        return at;
    };
    L.off_count = take((size_t)B * L.ncell * 4);
    L.off_fill = take((size_t)B * L.ncell * 4);
    L.off_start = take((size_t)B * (L.ncell + 1) * 4);
    L.off_scand = take((size_t)B * n * 16);
    L.off_cand = take((size_t)B * n * 16);
    L.off_sums = take((size_t)B * n * NFIELD * 8);
    L.nsub = choose_nsub(S);
    L.tile_h = 4 * choose_rows(S) * L.nsub;
    L.ntx = (W - 1 + TW - 1) / TW;
    L.nty = (H - 1 + L.tile_h - 1) / L.tile_h;
    if (L.ntx < 1) L.ntx = 1;
    if (L.nty < 1) L.nty = 1;
    L.off_tiles = take((size_t)B * L.ntx * L.nty * TILE_REC * sizeof(int8v));
    L.total = o;
    return L;
}

struct Ws {
    int* count;
    int* fill;
    int* start;
    int4* scand;            // cell-sorted copy of cand with .w = centre index
    int4* cand;             // cx, cy, packed Lab (L | a<<8 | b<<16), cell
    unsigned long long* sums;
    int8v* tiles;           // [B][nty][ntx][TILE_REC]: {m,..}, then the candidates by idx
    int ncx, ncy, ncell;
    int nsub, tile_h, ntx, nty;
};

Ws bind(void* ws, const Layout& L) {
    char* p = (char*)ws;
    Ws w;
    w.count = (int*)(p + L.off_count);
    w.fill = (int*)(p + L.off_fill);
    w.start = (int*)(p + L.off_start);
    w.scand = (int4*)(p + L.off_scand);
    w.cand = (int4*)(p + L.off_cand);
    w.sums = (unsigned long long*)(p + L.off_sums);
    w.ncx = L.ncx;
    w.ncy = L.ncy;
    w.ncell = L.ncell;
    w.tiles = (int8v*)(p + L.off_tiles);
    w.nsub = L.nsub;
    w.tile_h = L.tile_h;
    w.ntx = L.ntx;
    w.nty = L.nty;
    return w;
}

// ------------------------------------------------------------------ init (slic.py:35-76)
__global__ void slic_init_kernel(const uint8_t* __restrict__ lab, int32_t* __restrict__ centres,
                                 int B, int H, int W, int S, int nx, int ny) {
    int n = nx * ny;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * n) return;
    int b = t / n, c = t - b * n;
    int ix = c / ny, iy = c - ix * ny;
    int sx = S + ix * S, sy = S + iy * S;
    const uint8_t* img = lab + (size_t)b * H * W * 3;
    int best = 0x7fffffff, bx = sx, by = sy;
    for (int i = sx - 1; i <= sx + 1; ++i)
        for (int j = sy - 1; j <= sy + 1; ++j) {
            int l1 = img[((size_t)(j + 1) * W + i) * 3];
            int l2 = img[((size_t)j * W + i + 1) * 3];
            int l3 = img[((size_t)j * W + i) * 3];
            int g = abs(l1 - l3) + abs(l2 - l3);
            if (g < best) {
                best = g;
                bx = i;
                by = j;
            }
        }
    const uint8_t* px = img + ((size_t)by * W + bx) * 3;
    int32_t* o = centres + (size_t)t * 5;
    o[0] = px[0];
    o[1] = px[1];
    o[2] = px[2];
    o[3] = bx;
    o[4] = by;
}

// ------------------------------------------------------------------ binning
__global__ void slic_prepare_kernel(const uint8_t* __restrict__ lab,
                                    const int32_t* __restrict__ centres, Ws w, int B, int H, int W,
                                    int S, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * n) return;
    int b = t / n;
    const int32_t* c = centres + (size_t)t * 5;
    int cx = min(max(c[3], 0), W - 1), cy = min(max(c[4], 0), H - 1);
    const uint8_t* px = lab + (((size_t)b * H + cy) * W + cx) * 3;
    int packed = px[0] | (px[1] << 8) | (px[2] << 16);
    int cell = (cy / S) * w.ncx + (cx / S);
    w.cand[t] = make_int4(cx, cy, packed, cell);
    atomicAdd(&w.count[(size_t)b * w.ncell + cell], 1);
    unsigned long long* s = w.sums + (size_t)t * NFIELD;
#pragma unroll
    for (int k = 0; k < NFIELD; ++k) s[k] = 0ull;
}

// one workgroup per image: exclusive scan of the cell counts.  Four cells per thread, the running
// sum inside a wave by lane shuffles, across the waves through 16 words of LDS: two barriers per
// 4096 cells (a Hillis-Steele scan over LDS took twenty per 1024: 89 us per round on the 50,000
// cells of the 16384^2 mosaic).
__global__ __launch_bounds__(1024) void slic_scan_kernel(Ws w) {
    __shared__ int wave_sum[16];
    __shared__ int carry;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int* cnt = w.count + (size_t)b * w.ncell;
    int* start = w.start + (size_t)b * (w.ncell + 1);
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < w.ncell; base += 4096) {
        const int i0 = base + tid * 4;
        int v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i0 + u < w.ncell) ? cnt[i0 + u] : 0;
        const int mine = (v[0] + v[1]) + (v[2] + v[3]);
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int q = 0; q < wave; ++q) before += wave_sum[q];
        int run = before + incl - mine;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u < w.ncell) start[i0 + u] = run;
            run += v[u];
        }
        __syncthreads();
        if (tid == 1023) carry = run;
        __syncthreads();
    }
    if (tid == 0) start[w.ncell] = carry;
}

__global__ void slic_fill_kernel(Ws w, int B, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * n) return;
    int b = t / n, c = t - b * n;
    int4 q = w.cand[t];
    int pos = w.start[(size_t)b * (w.ncell + 1) + q.w] +
              atomicAdd(&w.fill[(size_t)b * w.ncell + q.w], 1);
    q.w = c;
    w.scand[(size_t)b * n + pos] = q;
}

// The three kernels above as one, for images whose cell grid fits LDS: one workgroup per image
// counts, scans and places its centres without leaving the CU (no global atomics, no memset,
// one launch instead of three plus a fill).
constexpr int BIN_THREADS = 1024;
constexpr int BIN_PER_THREAD = 4;                          // centres and cells per thread
constexpr int BIN_MAX = BIN_THREADS * BIN_PER_THREAD;      // most centres and most cells per image
__global__ __launch_bounds__(BIN_THREADS) void slic_bin_image_kernel(
    const uint8_t* __restrict__ lab, const int32_t* __restrict__ centres, Ws w, int H, int W, int S,
    int n) {
    __shared__ int s_cnt[BIN_MAX];                         // counts, then exclusive starts
    __shared__ int s_wave[BIN_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ncell = w.ncell;
    for (int i = tid; i < ncell; i += BIN_THREADS) s_cnt[i] = 0;
    __syncthreads();
    int4 q[BIN_PER_THREAD];
    int rank[BIN_PER_THREAD];
    const float inv_s = 1.0f / (float)S;
    auto quot = [&](int a) {                               // a / S for 0 <= a < 2^24
        int v = (int)((float)a * inv_s);
        const int r = a - v * S;
        return v + (r >= S) - (r < 0);
    };
#pragma unroll
    for (int k = 0; k < BIN_PER_THREAD; ++k) {
        const int c = tid + k * BIN_THREADS;
        rank[k] = 0;
        q[k] = make_int4(0, 0, 0, 0);
        if (c < n) {
            const int32_t* ce = centres + ((size_t)b * n + c) * 5;
            const int cx = min(max(ce[3], 0), W - 1), cy = min(max(ce[4], 0), H - 1);
            const uint8_t* px = lab + (((size_t)b * H + cy) * W + cx) * 3;
            const int packed = px[0] | (px[1] << 8) | (px[2] << 16);
            const int cell = quot(cy) * w.ncx + quot(cx);
            q[k] = make_int4(cx, cy, packed, cell);
            rank[k] = atomicAdd(&s_cnt[cell], 1);
            unsigned long long* sm = w.sums + ((size_t)b * n + c) * NFIELD;
#pragma unroll
            for (int f = 0; f < NFIELD; ++f) sm[f] = 0ull;
        }
    }
    __syncthreads();
    // exclusive scan of the counts: four consecutive cells per thread, a shuffle scan along the
    // wave, the wave totals through LDS
    int v[BIN_PER_THREAD], sum = 0;
#pragma unroll
    for (int k = 0; k < BIN_PER_THREAD; ++k) {
        const int i = tid * BIN_PER_THREAD + k;
        v[k] = i < ncell ? s_cnt[i] : 0;
        sum += v[k];
    }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int before = incl - sum;
    for (int k = 0; k < wv; ++k) before += s_wave[k];
    int* start = w.start + (size_t)b * (ncell + 1);
#pragma unroll
    for (int k = 0; k < BIN_PER_THREAD; ++k) {
        const int i = tid * BIN_PER_THREAD + k;
        if (i < ncell) {
            s_cnt[i] = before;
            start[i] = before;
        }
        before += v[k];
    }
    if (tid == 0) start[ncell] = n;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BIN_PER_THREAD; ++k) {
        const int c = tid + k * BIN_THREADS;
        if (c < n) {
            const int pos = s_cnt[q[k].w] + rank[k];
            w.scand[(size_t)b * n + pos] = make_int4(q[k].x, q[k].y, q[k].z, c);
        }
    }
}

// ------------------------------------------------------------------ per-tile candidate lists
struct AssignParams {
    int B, H, W, S, n, nsub, maxc;
    long long ntiles;         // B * ntx * nty
    unsigned S2, cw2;         // integer key weights (modes 0, 1); cw2 = 0 in mode 2
    double Sd, cw;            // literal float64 evaluation (mode 2 and exact-key ties)
    int* bbox;                // NULL, or [B][n][4] xmin, ymin, xmax, ymax of the new label map
    int first;                // first round: the incoming label map counts as all -1, unread
};

// Sixteen lanes per assignment tile (four tiles per wave): collects the centres whose window
// can reach the tile from the cell lists, sorts them by centre index and writes a fixed-size
// record of 8-dword entries
//     {cx', cy', lab, idx, A, Bx, By, 0}       (cx', cy' relative to the tile origin)
// with the per-candidate terms of the expanded integer key (see slic_assign_kernel):
//     A = S^2 |c|^2 + cw^2 (cx'^2 + cy'^2),  Bx = 2 cw^2 cx',  By = 2 cw^2 cy'.
// The assignment kernel reads the entries with scalar loads, so everything that is the same
// for the whole wave (the entry itself, the row tests, the y part of the key) costs scalar
// instructions only.
// A tile sees about nine centres, so a whole wave per tile spent its time issuing instructions
// for idle lanes (the kernel was bound by instruction issue, not by memory); a quarter wave per
// tile issues a quarter of them.  Index arithmetic avoids integer division: the quotients are
// below 2^24 and come from a float reciprocal with a one-step correction.
constexpr int TG = 16;                 // lanes per tile
constexpr int TILES_PER_WG = 256 / TG;
__global__ __launch_bounds__(256) void slic_tiles_kernel(Ws w, AssignParams P) {
    __shared__ int4 u_c[TILES_PER_WG][MAXC];
    const int H = P.H, W = P.W, S = P.S, n = P.n;
    const int lane = threadIdx.x & 63, gl = lane & (TG - 1), gbase = lane & ~(TG - 1);
    const int slot_wg = threadIdx.x / TG;
    const int ntiles = w.ntx * w.nty;
    const int b = (int)(blockIdx.z * 65535u + blockIdx.y);
    if (b >= P.B) return;                                // workgroup uniform
    const int ti = (int)blockIdx.x * TILES_PER_WG + slot_wg;
    const bool live = ti < ntiles;
    auto quot = [](int a, int d, float inv) {
        int q = (int)((float)a * inv);
        const int r = a - q * d;
        return q + (r >= d) - (r < 0);
    };
    const float inv_s = 1.0f / (float)S, inv_ntx = 1.0f / (float)w.ntx;
    const int ty = quot(live ? ti : 0, w.ntx, inv_ntx), tx = (live ? ti : 0) - ty * w.ntx;
    const int x0 = tx * TW, y0 = ty * w.tile_h;
    const int x1 = min(x0 + TW - 1, W - 2), y1 = min(y0 + w.tile_h - 1, H - 2);
    // centres that can cover a pixel of the tile: cx in [x0-S+1, x1+S], cy likewise
    const int cxl = quot(max(x0 - S + 1, 0), S, inv_s), cxh = min(quot(x1 + S, S, inv_s), w.ncx - 1);
    const int cyl = quot(max(y0 - S + 1, 0), S, inv_s), cyh = min(quot(y1 + S, S, inv_s), w.ncy - 1);
    const int4* scand = w.scand + (size_t)b * n;
    const int* start = w.start + (size_t)b * (w.ncell + 1);
    const int nrow = live ? cyh - cyl + 1 : 0;
    const unsigned below = (1u << gl) - 1u;              // the lanes of the group before this one
    int4* mine = u_c[slot_wg];
    int m = 0;
    // Cell rows in chunks of TG: lane r of the group fetches the bounds of row r0 + r, a prefix
    // sum along the group places the rows' candidates side by side, and the lanes then take one
    // candidate each (two memory round trips for a tile whose candidates fit the group).
    for (int r0 = 0; __any(r0 < nrow); r0 += TG) {
        int myb = 0, cnt = 0;
        if (r0 + gl < nrow) {
            myb = start[(cyl + r0 + gl) * w.ncx + cxl];
            cnt = start[(cyl + r0 + gl) * w.ncx + cxh + 1] - myb;
        }
        int incl = cnt;                                  // inclusive prefix sum over the group
#pragma unroll
        for (int d = 1; d < TG; d <<= 1) {
            const int up = __shfl(incl, lane - d);
            if (gl >= d) incl += up;
        }
        const int total = __shfl(incl, gbase + TG - 1);
        const int rows_here = min(max(nrow - r0, 0), TG);
        for (int l0 = 0; __any(l0 < total); l0 += TG) {
            const int l = l0 + gl;
            const bool valid = l < total;
            // the row of candidate l: the number of rows whose inclusive prefix is <= l
            int r = 0;
            for (int k = 0; __any(k < rows_here - 1); ++k) {
                const int pk = __shfl(incl, gbase + k);
                r += (k < rows_here - 1 && l >= pk) ? 1 : 0;
            }
            const int rb = __shfl(myb, gbase + r);
            const int ri = __shfl(incl, gbase + r), rc = __shfl(cnt, gbase + r);
            bool hit = false;
            int4 q = make_int4(0, 0, 0, 0);
            if (valid) {
                q = scand[rb + (l - (ri - rc))];
                hit = q.x >= x0 - S + 1 && q.x <= x1 + S && q.y >= y0 - S + 1 && q.y <= y1 + S;
            }
            const unsigned vote = (unsigned)(__ballot(hit) >> gbase) & ((1u << TG) - 1u);
            const int slot = m + __popc(vote & below);
            if (hit && slot < MAXC) mine[slot] = q;
            m += __popc(vote);
        }
    }
    __builtin_amdgcn_wave_barrier();                     // a tile's list belongs to one wave
    if (!live) return;
    int8v* rec = w.tiles + ((size_t)b * ntiles + ti) * TILE_REC;
    if (gl == 0) {
        int8v h = {m, x0, y0, 0, 0, 0, 0, 0};
        rec[0] = h;
    }
    if (m > MAXC) return;
    for (int e = gl; e < m; e += TG) {
        const int4 me = mine[e];
        int rank = 0;
        for (int j = 0; j < m; ++j) rank += (mine[j].w < me.w);
        const int cxr = me.x - x0, cyr = me.y - y0;
        const uint32_t cl = (uint32_t)me.z;
        const uint32_t cc = __builtin_amdgcn_udot4(cl, cl, 0u, false);
        const uint32_t A = P.S2 * cc + P.cw2 * (uint32_t)(cxr * cxr + cyr * cyr);
        int8v e8 = {cxr, cyr, me.z, me.w, (int)A, 2 * (int)P.cw2 * cxr, 2 * (int)P.cw2 * cyr, 0};
        rec[1 + rank] = e8;
    }
}

// ------------------------------------------------------------------ assignment
struct __attribute__((packed, aligned(4))) Dword3 {
    uint32_t a, b, c;
};

// slic.py:101-109 evaluated literally in float64 (numpy order of operations, no contraction):
//   dc = sqrt(sum(square(dLab))); ds = sqrt(dx^2 + dy^2); D = sqrt((dc/cw)^2 + (ds/S)^2)
__device__ __attribute__((noinline)) double literal_dist(uint32_t p, int x, int y, uint32_t cl,
                                                         int cx, int cy, double Sd, double cw) {
    int dL = (int)(p & 255) - (int)(cl & 255);
    int da = (int)((p >> 8) & 255) - (int)((cl >> 8) & 255);
    int db = (int)((p >> 16) & 255) - (int)((cl >> 16) & 255);
    int dx = x - cx, dy = y - cy;
    double dc = sqrt((double)(dL * dL + da * da + db * db));
    double ds = sqrt((double)(dx * dx + dy * dy));
    double u = dc / cw, v = ds / Sd;
    return sqrt(u * u + v * v);
}

// Generic key: mode 1 = exact u64 (same order as the float64 expression whenever the keys
// differ: distinct integer keys below 2^64 differ by far more than the float64 rounding of
// the expression); mode 2 = the literal float64 distance (non-integral colour weight).
template <int MODE>
struct KeyT {
    using type = unsigned long long;
};
template <>
struct KeyT<2> {
    using type = double;
};

template <int MODE>
__device__ __forceinline__ typename KeyT<MODE>::type generic_key(uint32_t p, int x, int y,
                                                                 uint32_t cl, int cx, int cy,
                                                                 const AssignParams& P) {
    int dL = (int)(p & 255) - (int)(cl & 255);
    int da = (int)((p >> 8) & 255) - (int)((cl >> 8) & 255);
    int db = (int)((p >> 16) & 255) - (int)((cl >> 16) & 255);
    unsigned dc2 = dL * dL + da * da + db * db;
    int dx = x - cx, dy = y - cy;
    unsigned ds2 = dx * dx + dy * dy;
    if (MODE == 2) return (typename KeyT<MODE>::type)literal_dist(p, x, y, cl, cx, cy, P.Sd, P.cw);
    return (typename KeyT<MODE>::type)((unsigned long long)dc2 * P.S2 +
                                       (unsigned long long)ds2 * P.cw2);
}

__device__ __forceinline__ void flush_pixel_global(unsigned long long* sums, int label, uint32_t p,
                                                   int x, int y) {
    unsigned long long* s = sums + (size_t)label * NFIELD;
    atomicAdd(&s[0], (unsigned long long)(p & 255));
    atomicAdd(&s[1], (unsigned long long)((p >> 8) & 255));
    atomicAdd(&s[2], (unsigned long long)((p >> 16) & 255));
    atomicAdd(&s[3], (unsigned long long)x);
    atomicAdd(&s[4], (unsigned long long)y);
    atomicAdd(&s[5], 1ull);
}

__device__ __forceinline__ void bbox_point_global(int* bb, int xa, int ya, int xb, int yb) {
    atomicMin(&bb[0], xa);
    atomicMin(&bb[1], ya);
    atomicMax(&bb[2], xb);
    atomicMax(&bb[3], yb);
}

__global__ void slic_bbox_init_kernel(int* __restrict__ bbox, int total) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    reinterpret_cast<int4*>(bbox)[t] = make_int4(0x7fffffff, 0x7fffffff, -1, -1);
}

// MODE 0: 32-bit key with the dot4 expansion (fast path); 1: u64; 2: literal float64.
// REL (mode 0 only): the key is kept relative to its per-pixel constant (see below).
//
// One workgroup = a 64 x tile_h pixel tile made of nsub bands of 4 x R rows; wave w takes rows
// R w .. R w + R - 1 of every band.  Inside a wave LANE = COLUMN and the R pixels of a lane are
// R consecutive ROWS, so for a given candidate everything that depends on the row is the same
// for all 64 lanes and lives in scalar registers (the row-in-window tests, the y part of the
// key); per lane and candidate only the column test and the x part are left, paid once for the
// lane's R pixels.
//
// Mode 0 key.  With p, c the Lab bytes of pixel and centre, (x, y) / (cx', cy') tile relative:
//     key = S^2 |p - c|^2 + cw^2 ((x - cx')^2 + (y - cy')^2)
//         = T(p, x, y) + A(c) - Bx x - By y - 2 S^2 (p . c)
// T is the same for every candidate of a pixel, so it cannot change which one wins: REL drops
// it and compares the signed remainder (needs |key| < 2^31, checked on the host); otherwise
// T is added and the comparison is unsigned below 2^32.  p . c is one v_dot4_u32_u8.
__device__ __forceinline__ int smed3(int a, int b, int c) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// 7 waves per SIMD: what the LDS of the common configurations allows (one workgroup = one wave
// per SIMD); without the hint the compiler settles on 77 registers, one wave short of that
#ifdef IMGAN_ASSIGN_COUNTS
// which epilogue the bands of the assignment kernel take (tools/assign_counts.py; never built
// into the product): 0 fast, 1 rows not full, 2 more centres than LDS slots, 3 uncovered pixel,
// 4 bands with a tie replay
__device__ unsigned long long assign_counts[8];
#define ASSIGN_COUNT(i) do { if (lane == 0) atomicAdd(&assign_counts[i], 1ull); } while (0)
#else
#define ASSIGN_COUNT(i) do { } while (0)
#endif
#ifndef IMGAN_ASSIGN_WAVES
#define IMGAN_ASSIGN_WAVES __attribute__((amdgpu_waves_per_eu(7, 7)))
#endif
template <int MODE, bool ALIGNED, bool REL, int R>
__global__ __launch_bounds__(256) IMGAN_ASSIGN_WAVES void slic_assign_kernel(const uint8_t* __restrict__ lab,
                                                          int32_t* __restrict__ labels, Ws w,
                                                          AssignParams P) {
    // dynamic LDS, sized by the host for the number of centres a tile is expected to see
    // (P.maxc): fewer bytes per workgroup = more workgroups per CU to hide the prologue.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int maxc = P.maxc;
    int4* s_c = reinterpret_cast<int4*>(smem);                  // by rank: {cx', cy', lab, idx}
    uint32_t* acc = reinterpret_cast<uint32_t*>(s_c + MAXC);    // [maxc slots][word][lane] partial sums
    uint32_t* tr = acc + maxc * 192;                            // [wave][row][column] transposition
    // bounding boxes of the new labels (P.bbox, last round only): the columns of a slot are the
    // lanes whose count is not zero; its rows come from a lane-private min / max pair
    uint32_t* ybox = tr + 4 * 4 * 64;                           // [slot][min, max][group of 4 lanes]
    int* const bbox = P.bbox;                                   // wave uniform

    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2),
    // so XCD k takes the k-th contiguous eighth of the tile list.  Horizontally adjacent
    // tiles then run on the same XCD right after each other, and the 128-byte line that their
    // 192-byte rows share is fetched from HBM once instead of twice.
    const long long per_xcd = (P.ntiles + 7) / 8;
    const long long tile = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= P.ntiles) return;
    const int tiles_per_image = w.ntx * w.nty;
    const int b = (int)(tile / tiles_per_image);
    const int trem = (int)(tile - (long long)b * tiles_per_image);
    const int tby = trem / w.ntx, tbx = trem - tby * w.ntx;
    const int H = P.H, W = P.W, S = P.S, n = P.n, nsub = P.nsub;
    constexpr int BAND = 4 * R;                           // rows per band: 4 waves x R rows
    const int x0 = tbx * TW, y0 = tby * w.tile_h;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long* sums = w.sums + (size_t)b * n * NFIELD;

    // Every band's pixels are requested up front, before anything else: the kernel needs
    // many bytes in flight, so each wave keeps nsub x 768 B outstanding.  The loads use the
    // row-major mapping (16 lanes x 12 B per row, 4 rows); the bands are transposed to the
    // lane = column mapping through LDS when they are used.
    auto unpack = [&](const Dword3& v, uint32_t* px) {
        px[0] = v.a & 0xffffffu;
        px[1] = __builtin_amdgcn_perm(v.b, v.a, 0x0c050403u);
        px[2] = __builtin_amdgcn_perm(v.c, v.b, 0x0c040302u);
        px[3] = v.c >> 8;
    };
    Dword3 raw[NSUBMAX][R / 4];
    if (ALIGNED) {
        const int xl = x0 + (lane & 15) * 4;
        const int yl = y0 + wave * R + (lane >> 4);
        // one 64-bit address per lane, stepped by whole rows (uniform offsets)
        const uint8_t* src = lab + (((size_t)b * H + yl) * W + xl) * 3;
        const size_t row3 = (size_t)W * 3;
#pragma unroll
        for (int sub = 0; sub < NSUBMAX; ++sub)
#pragma unroll
            for (int g = 0; g < R / 4; ++g) {
                raw[sub][g].a = raw[sub][g].b = raw[sub][g].c = 0;
                const int dy = sub * BAND + g * 4;
                if (sub < nsub && yl + dy < H - 1 && xl < W - 1)
                    raw[sub][g] = *reinterpret_cast<const Dword3*>(src + (size_t)dy * row3);
            }
    }

    // the tile's sorted candidate record (slic_tiles_kernel), read through the scalar cache
    const int8v* recv = w.tiles + (size_t)tile * TILE_REC;
    const int8v AS_CONST* rec = (const int8v AS_CONST*)recv;
    const int m = rec[0][0];
    if (tid < m && m <= MAXC) {
        const int8v e = recv[1 + tid];
        s_c[tid] = make_int4(e[0], e[1], e[2], e[3]);
    }
    for (int i = tid; i < maxc * 3 * 16; i += 256)
        reinterpret_cast<uint4*>(acc)[i] = make_uint4(0, 0, 0, 0);
    if (bbox)
        for (int i = tid; i < maxc * 32; i += 256) ybox[i] = (i & 16) ? 0u : 0x7fffffffu;

    const int xr = lane;                                 // tile relative column of this lane
    const int x = x0 + xr;
    const bool colok = x < W - 1;

    // A tile that sees more centres than the LDS accumulators were sized for (maxc) still takes
    // the fast loop: its record holds up to MAXC entries, and only the pixels won by a slot
    // beyond maxc go to global memory one by one.
    const int macc = min(m, maxc);                         // slots with LDS accumulators
    if (m > MAXC) {
        // ---- slow path: more centres around this tile than a record holds (steps of a few pixels).  Every lane
        // walks the cell lists itself for one pixel at a time and flushes with global atomics.
        const int x1 = min(x0 + TW - 1, W - 2), y1 = min(y0 + w.tile_h - 1, H - 2);
        const int4* scand = w.scand + (size_t)b * n;
        const int* start = w.start + (size_t)b * (w.ncell + 1);
        const int cxl = max(x0 - S + 1, 0) / S, cxh = min((x1 + S) / S, w.ncx - 1);
        const int cyl = max(y0 - S + 1, 0) / S, cyh = min((y1 + S) / S, w.ncy - 1);
        for (int y = y0 + (tid >> 6); y <= y1; y += 4) {
            if (!colok) continue;
            const size_t li = ((size_t)b * H + y) * W + x;
            const uint8_t* q3 = lab + li * 3;
            const uint32_t px = q3[0] | (q3[1] << 8) | (q3[2] << 16);
            typename KeyT<MODE == 2 ? 2 : 1>::type best = 0;
            int bi = -1;
            int4 qb = make_int4(0, 0, 0, 0);
            for (int cyc = cyl; cyc <= cyh; ++cyc) {
                int beg = start[cyc * w.ncx + cxl], end = start[cyc * w.ncx + cxh + 1];
                for (int i = beg; i < end; ++i) {
                    int4 q = scand[i];
                    if ((unsigned)(x - q.x + S) >= (unsigned)(2 * S) ||
                        (unsigned)(y - q.y + S) >= (unsigned)(2 * S))
                        continue;
                    auto k = generic_key<MODE == 2 ? 2 : 1>(px, x, y, q.z, q.x, q.y, P);
                    bool take = bi < 0 || k < best;
                    if (!take && k == best) {
                        // equal exact keys: the reference compares the rounded float64
                        // values; on equality the lower centre index (earlier) stays
                        double dn = literal_dist(px, x, y, q.z, q.x, q.y, P.Sd, P.cw);
                        double db = literal_dist(px, x, y, qb.z, qb.x, qb.y, P.Sd, P.cw);
                        take = dn < db || (dn == db && q.w < bi);
                    }
                    if (take) {
                        best = k;
                        bi = q.w;
                        qb = q;
                    }
                }
            }
            int lbl = bi;
            if (bi >= 0 || P.first)
                labels[li] = bi;
            else
                lbl = labels[li];
            if ((unsigned)lbl < (unsigned)n) {            // a kept label comes from the caller's map
                flush_pixel_global(sums, lbl, px, x, y);
                if (bbox) bbox_point_global(bbox + ((size_t)b * n + lbl) * 4, x, y, x, y);
            }
        }
        return;
    }
    __syncthreads();

    const int twoS = 2 * S;
    const int nS2x2 = -2 * (int)P.S2;
    uint32_t* trw = tr + wave * (4 * 64);                // 4 rows at a time
    // unrolled over the (two) bands: which prefetched pixels a band uses is then fixed in the code
    // instead of picked with selects (0.988 -> 0.983 ms)
#pragma unroll
    for (int sub = 0; sub < NSUBMAX; ++sub) {
        if (sub >= nsub) break;
        const int wy0 = sub * BAND + wave * R;           // tile relative first row of this wave
        if (y0 + wy0 >= H - 1) continue;                 // whole wave below the image
        uint32_t px[R];
        if (ALIGNED) {
            // LDS operations of one wave complete in order: no barrier between the row-major
            // store and the column reads, nor before the next band overwrites the patch
#pragma unroll
            for (int g = 0; g < R / 4; ++g) {
                Dword3 v = raw[0][g];
#pragma unroll
                for (int k = 1; k < NSUBMAX; ++k)
                    if (sub == k) v = raw[k][g];
                uint32_t q[4];
                unpack(v, q);
                *reinterpret_cast<uint4*>(trw + (lane >> 4) * 64 + (lane & 15) * 4) =
                    make_uint4(q[0], q[1], q[2], q[3]);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 4; ++r) px[g * 4 + r] = trw[r * 64 + lane];
                __builtin_amdgcn_wave_barrier();
            }
        } else {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                px[j] = 0;
                const int yy = y0 + wy0 + j;
                if (yy < H - 1 && colok) {
                    const uint8_t* q = lab + (((size_t)b * H + yy) * W + x) * 3;
                    px[j] = q[0] | (q[1] << 8) | (q[2] << 16);
                }
            }
        }

        int slot[R];
#pragma unroll
        for (int j = 0; j < R; ++j) slot[j] = -1;
        unsigned tied = 0;                  // bit j: the best key of pixel j may have been met again
        int lowest = 0;                     // the lane's lowest slot: negative if one of its pixels has none
        if (MODE == 0) {
            // Issue costs measured on this chip (tools/issue_rate.hip, profiles/r03_issue_rate.txt): a
            // wave64 add / subtract / logic operation (v_bitop3 included) whose operands are vector
            // registers or inline constants takes 2 cycles of its SIMD; everything else used here
            // (v_dot4, v_mad_i32_i24, v_min, v_med3, compares, selects) and ANY operation with a
            // scalar-register operand takes 4.  Per (row, candidate) the loop is
            //     pc  = v_dot4(px, c.lab)                  4
            //     key = v_mad_i32_i24(pc, -2 S^2, Xj)      4
            //     kp  = v_bitop3: (key & ~M) | slot        2   mask and slot in vector registers
            //     b2  = v_med3(kp, best, b2)               4
            //     best = v_min(best, kp)                   4
            //     Xj -= By                                 2   By copied to a vector register
            // The winner's slot rides in the low bits of the key (M = 15 for a tile with at most 16
            // candidates, else 63), which replaces the compare / select pair of a plain arg-min
            // (measured: 0.514 -> 0.489 ms per launch at 128 images).  Keys are therefore compared
            // above those bits only: a row in which some pixel's two lowest packed keys agree there
            // ((best ^ b2) <= M) is replayed with full keys further down.
            int best[R], best2[R];          // lowest and second lowest packed key met so far
            uint32_t T[R];
            const int mbits = m <= 16 ? 15 : 63;                       // wave uniform
            int maskv, sv;                                             // ~M and the slot, per lane
            asm("v_mov_b32 %0, %1" : "=v"(maskv) : "s"(~mbits));
#pragma unroll
            for (int j = 0; j < R; ++j) {
                best[j] = best2[j] = REL ? 0x7fffffff : -1;
                T[j] = 0;
                if (!REL) {
                    const uint32_t pp = __builtin_amdgcn_udot4(px[j], px[j], 0u, false);
                    T[j] = pp * P.S2 + P.cw2 * (uint32_t)(xr * xr + (wy0 + j) * (wy0 + j));
                }
            }
            auto full_key = [&](const int8v& c, int j) {
                const int pc = (int)__builtin_amdgcn_udot4(px[j], (uint32_t)c[2], 0u, false);
                int key = __mul24(nS2x2, pc) + (__mul24(-c[5], xr) + (c[4] - c[6] * (wy0 + j)));
                if (!REL) key = (int)((uint32_t)key + T[j]);
                return key;
            };
            auto eval = [&](int s, const int8v& c, int t0, bool all_rows) {
                if ((unsigned)(xr - c[0] + S) < (unsigned)twoS) {
                    int Xj = __mul24(-c[5], xr) + (c[4] - c[6] * wy0);
                    int byv;
                    asm("v_mov_b32 %0, %1" : "=v"(byv) : "s"(c[6]));
                    asm("v_mov_b32 %0, %1" : "=v"(sv) : "s"(s));
                    const uint32_t cl = (uint32_t)c[2];
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        if (all_rows || (unsigned)(t0 + j) < (unsigned)twoS) {
                            const int pc = (int)__builtin_amdgcn_udot4(px[j], cl, 0u, false);
                            int key = __mul24(nS2x2, pc) + Xj;
                            if (!REL) key = (int)((uint32_t)key + T[j]);
                            const int kp = (int)__builtin_amdgcn_bitop3_b32(
                                (uint32_t)key, (uint32_t)maskv, (uint32_t)sv, 0xeau);   // (a & b) | c
                            best2[j] = REL ? smed3(kp, best[j], best2[j])
                                           : (int)umed3((uint32_t)kp, (uint32_t)best[j],
                                                        (uint32_t)best2[j]);
                            best[j] = REL ? min(best[j], kp)
                                          : (int)min((uint32_t)best[j], (uint32_t)kp);
                        }
                        Xj -= byv;
                    }
                }
            };
            unsigned long long partial = 0ull;           // candidates that hold some rows only
            const unsigned all_lim = (unsigned)max(twoS - (R - 1), 0);
            for (int s = 0; s < m; ++s) {
                const int8v c = rec[1 + s];
                const int t0 = wy0 - c[1] + S;           // row j is inside iff 0 <= t0 + j < 2S
                if ((unsigned)t0 < all_lim) {
                    eval(s, c, t0, true);                // every row of this wave inside
                } else if ((unsigned)(t0 + (R - 1)) < (unsigned)(twoS + (R - 1))) {
                    partial |= 1ull << s;
                }
            }
            while (partial != 0ull) {
                const int s = __builtin_ctzll(partial);
                partial &= partial - 1ull;
                const int8v c = rec[1 + s];
                eval(s, c, wy0 - c[1] + S, false);
            }
            // one test for the lane's R rows: the smallest difference of the two lowest keys
            uint32_t gap[R];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                slot[j] = best[j] & mbits;
                gap[j] = (uint32_t)(best[j] ^ best2[j]);
            }
            uint32_t least = gap[0];
#pragma unroll
            for (int j = 1; j + 1 < R; j += 2) least = umin3(least, gap[j], gap[j + 1]);
            if ((R & 1) == 0) least = min(least, gap[R - 1]);
            lowest = 0;                                  // every pixel has a slot unless the replay says otherwise
            if (__builtin_amdgcn_ballot_w64(least <= (uint32_t)mbits) != 0ull) {   // wave uniform
                ASSIGN_COUNT(4);
                // Rows that hold a pixel whose two lowest keys agree above the slot bits (or a pixel
                // no candidate covers: both still at their start value): the whole wave goes over
                // the row's candidates again with full keys.  The first strictly lower key wins
                // (= the lowest centre index among equal keys); a pixel whose lowest full key is met
                // twice goes on to the float64 replay below, as the reference compares rounded
                // float64 distances.
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    if (__builtin_amdgcn_ballot_w64(gap[j] <= (uint32_t)mbits) == 0ull) continue;
                    int kb = 0, sb = -1;
                    bool dup = false;
                    for (int s = 0; s < m; ++s) {
                        const int8v c = rec[1 + s];
                        if ((unsigned)(wy0 + j - c[1] + S) >= (unsigned)twoS) continue;   // uniform
                        if ((unsigned)(xr - c[0] + S) >= (unsigned)twoS) continue;
                        const int key = full_key(c, j);
                        const bool lt = sb < 0 || (REL ? key < kb : (uint32_t)key < (uint32_t)kb);
                        dup = lt ? false : (dup || key == kb);
                        kb = lt ? key : kb;
                        sb = lt ? s : sb;
                    }
                    slot[j] = sb;
                    tied |= (dup && sb >= 0) ? (1u << j) : 0u;
                }
#pragma unroll
                for (int j = 0; j < R; ++j) lowest = min(lowest, slot[j]);     // -1: a pixel no window covers
            }
        } else {
            typename KeyT<MODE>::type best[R];
            for (int s = 0; s < m; ++s) {
                const int4 c0 = s_c[s];
                const bool inx = (unsigned)(xr - c0.x + S) < (unsigned)twoS;
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    bool in = inx && ((unsigned)(wy0 + j - c0.y + S) < (unsigned)twoS);
                    if (!in) continue;
                    auto key = generic_key<MODE>(px[j], xr, wy0 + j, (uint32_t)c0.z, c0.x, c0.y, P);
                    if (slot[j] < 0 || key < best[j]) {
                        best[j] = key;
                        slot[j] = s;
                        tied &= ~(1u << j);
                    } else if (MODE == 1 && key == best[j]) {
                        tied |= 1u << j;
                    }
                }
            }
        }
        // ---- equal exact keys (rare): the reference compares the rounded float64 distances in
        // centre order with a strict '<', so replay the candidates that cover the pixel.
        if (tied) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (!((tied >> j) & 1u)) continue;
                const int yr = wy0 + j;
                const int4 cb = s_c[slot[j]];
                const auto kbest = generic_key<1>(px[j], xr, yr, (uint32_t)cb.z, cb.x, cb.y, P);
                double dbest = 0.0;
                int sb = -1;
                for (int s = 0; s < m; ++s) {
                    const int4 c0 = s_c[s];
                    if ((unsigned)(yr - c0.y + S) >= (unsigned)twoS ||
                        (unsigned)(xr - c0.x + S) >= (unsigned)twoS)
                        continue;
                    if (generic_key<1>(px[j], xr, yr, (uint32_t)c0.z, c0.x, c0.y, P) != kbest)
                        continue;
                    double d = literal_dist(px[j], xr, yr, (uint32_t)c0.z, c0.x, c0.y, P.Sd, P.cw);
                    if (sb < 0 || d < dbest) {
                        dbest = d;
                        sb = s;
                    }
                }
                slot[j] = sb;
            }
        }

        // ---- labels out + accumulate (a row of 64 labels per store instruction).  Row
        // addresses are wave uniform (scalar base + lane offset); LDS offsets use 24-bit
        // multiplies (the 32-bit multiply-add of this chip is a quarter-rate 64-bit one).
        int32_t* rowp = labels + ((size_t)b * H + (y0 + wy0)) * W + x0;
        if (MODE != 0) {
            lowest = slot[0];
#pragma unroll
            for (int j = 1; j < R; ++j) lowest = min(lowest, slot[j]);
        }
        const bool rows_full = y0 + wy0 + R <= H - 1;
        if (rows_full && m <= maxc && __builtin_amdgcn_ballot_w64(colok && lowest < 0) == 0ull) {
            ASSIGN_COUNT(0);
            // the common case, free of per-row branches: every pixel of the wave has a centre
            if (colok) {
                int lbl[R];
#pragma unroll
                for (int j = 0; j < R; ++j) lbl[j] = s_c[slot[j]].w;
                // the stores: one scalar base for the R rows, the row's byte offset in a vector
                // register (the compiler otherwise steps a 64-bit vector address from row to row)
                {
                    uint32_t off = (uint32_t)lane * 4u, row_bytes;
                    asm("v_mov_b32 %0, %1" : "=v"(row_bytes) : "s"((uint32_t)W * 4u));
                    char* const rows = reinterpret_cast<char*>(rowp);
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        *reinterpret_cast<int32_t*>(rows + off) = lbl[j];
                        off += row_bytes;
                    }
                }
                uint32_t xy0 = (uint32_t)xr | ((uint32_t)wy0 << 16);
                asm("" : "+v"(xy0));    // one register for the band: the row index is added as a literal
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const uint32_t p = px[j];
                    uint32_t* a = acc + (__mul24(slot[j], 192) + lane);
                    atomicAdd(a, p & 0x00ff00ffu);                                    // L | b << 16
                    atomicAdd(a + 64, __builtin_amdgcn_perm(0x01u, p, 0x0c040c01u));  // a | 1 << 16
                    atomicAdd(a + 128, xy0 + ((uint32_t)j << 16));
                }
                if (bbox) {
                    // one min / max pair per run of equal slots down the lane's rows (one run, as
                    // a rule); four lanes share a copy, which keeps the rows' extremes small enough
                    // in LDS for the seventh workgroup of a CU
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        uint32_t* q = ybox + (__mul24(slot[j], 32) + (lane >> 2));
                        if (j == 0 || slot[j] != slot[j - 1]) atomicMin(q, (uint32_t)(wy0 + j));
                        if (j == R - 1 || slot[j] != slot[j + 1]) atomicMax(q + 16, (uint32_t)(wy0 + j));
                    }
                }
            }
        } else {
            ASSIGN_COUNT(!rows_full ? 1 : m > maxc ? 2 : 3);
            int32_t* lrow = rowp + lane;
#pragma unroll
            for (int j = 0; j < R; ++j, lrow += W) {
                const int yr = wy0 + j, y = y0 + yr;
                if (!(y < H - 1 && colok)) continue;
                if (slot[j] >= maxc) {                // no LDS accumulator for this slot
                    const int lbl = s_c[slot[j]].w;
                    *lrow = lbl;
                    flush_pixel_global(sums, lbl, px[j], x, y);
                    if (bbox) bbox_point_global(bbox + ((size_t)b * n + lbl) * 4, x, y, x, y);
                } else if (slot[j] >= 0) {
                    *lrow = s_c[slot[j]].w;
                    const uint32_t p = px[j];
                    uint32_t* a = acc + (__mul24(slot[j], 192) + lane);
                    atomicAdd(a, p & 0x00ff00ffu);
                    atomicAdd(a + 64, ((p >> 8) & 0xffu) | 0x10000u);
                    atomicAdd(a + 128, (uint32_t)xr | ((uint32_t)yr << 16));
                    if (bbox) {
                        uint32_t* q = ybox + (__mul24(slot[j], 32) + (lane >> 2));
                        atomicMin(q, (uint32_t)yr);
                        atomicMax(q + 16, (uint32_t)yr);
                    }
                } else {
                    // not covered this round: keeps its label (-1 in the first round, which
                    // writes it instead of reading it)
                    if (P.first) *lrow = -1;
                    const int old = P.first ? -1 : *lrow;
                    if ((unsigned)old < (unsigned)n) {    // it comes from the caller's map
                        flush_pixel_global(sums, old, px[j], x, y);
                        if (bbox) bbox_point_global(bbox + ((size_t)b * n + old) * 4, x, y, x, y);
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- reduce the 64 lane-private copies and flush: 16 lanes per slot = 4 lanes for each of
    // its 3 packed rows (L|b, a|count, x'|y'), 16 copies per lane.  A copy holds at most
    // 4 waves x nsub bands x R rows = tile height <= 64 pixels, so its 16-bit halves cannot overflow
    // (64 * 255 < 65536); the halves are widened before they are summed.
    for (int s0 = 0; s0 < macc; s0 += 16) {
        const int s = s0 + (tid >> 4), r = (tid >> 2) & 3, q = tid & 3;
        const bool live = s < macc && r < 3;
        uint32_t lo = 0, hi = 0;
        if (live) {
            const uint4* src = reinterpret_cast<const uint4*>(acc + (s * 3 + r) * 64 + q * 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint4 v = src[k];
                lo += (v.x & 0xffffu) + (v.y & 0xffffu) + (v.z & 0xffffu) + (v.w & 0xffffu);
                hi += (v.x >> 16) + (v.y >> 16) + (v.z >> 16) + (v.w >> 16);
            }
        }
        // lanes ^ 1 and ^ 2 by DPP quad permutations (a shuffle is an LDS round trip)
        lo += (uint32_t)__builtin_amdgcn_mov_dpp((int)lo, 0xb1, 0xf, 0xf, true);
        hi += (uint32_t)__builtin_amdgcn_mov_dpp((int)hi, 0xb1, 0xf, 0xf, true);
        lo += (uint32_t)__builtin_amdgcn_mov_dpp((int)lo, 0x4e, 0xf, 0xf, true);
        hi += (uint32_t)__builtin_amdgcn_mov_dpp((int)hi, 0x4e, 0xf, 0xf, true);
        const uint32_t cnt = __shfl(hi, (lane & 48) + 4);      // row 1 of this slot: a | count
        if (live && q == 0 && cnt) {
            unsigned long long* dst = sums + (size_t)s_c[s].w * NFIELD;
            if (r == 0) {
                atomicAdd(&dst[0], (unsigned long long)lo);
                atomicAdd(&dst[2], (unsigned long long)hi);
            } else if (r == 1) {
                atomicAdd(&dst[1], (unsigned long long)lo);
                atomicAdd(&dst[5], (unsigned long long)cnt);
            } else {
                atomicAdd(&dst[3], (unsigned long long)lo + (unsigned long long)cnt * x0);
                atomicAdd(&dst[4], (unsigned long long)hi + (unsigned long long)cnt * y0);
            }
        }
    }
    if (bbox) {
        // 16 lanes per slot, 4 columns per lane: the columns that hold the slot (count != 0)
        // give the x range directly, their row extremes the y range; one global update per
        // (tile, slot) and bound
        for (int s0 = 0; s0 < macc; s0 += 16) {
            const int s = s0 + (tid >> 4), t16 = tid & 15;
            int xa = 0x7fffffff, ya = 0x7fffffff, xb = -1, yb = -1;
            if (s < macc) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int col = 4 * t16 + c;
                    if ((acc[(s * 3 + 1) * 64 + col] >> 16) != 0u) {
                        xa = min(xa, col);
                        xb = max(xb, col);
                        ya = min(ya, (int)ybox[s * 32 + t16]);       // the rows of this lane's four columns
                        yb = max(yb, (int)ybox[s * 32 + 16 + t16]);
                    }
                }
            }
            // over the slot's 16 lanes = one DPP row: quad permutations, then the two mirrors
#define ROW_STEP(ctrl)                                                                  \
            xa = min(xa, __builtin_amdgcn_mov_dpp(xa, ctrl, 0xf, 0xf, true));           \
            ya = min(ya, __builtin_amdgcn_mov_dpp(ya, ctrl, 0xf, 0xf, true));           \
            xb = max(xb, __builtin_amdgcn_mov_dpp(xb, ctrl, 0xf, 0xf, true));           \
            yb = max(yb, __builtin_amdgcn_mov_dpp(yb, ctrl, 0xf, 0xf, true));
            ROW_STEP(0xb1) ROW_STEP(0x4e) ROW_STEP(0x141) ROW_STEP(0x140)
#undef ROW_STEP
            if (s < macc && t16 == 0 && xb >= 0)
                bbox_point_global(bbox + ((size_t)b * n + s_c[s].w) * 4, x0 + xa, y0 + ya, x0 + xb,
                                  y0 + yb);
        }
    }
}

// ------------------------------------------------------------------ finalise (slic.py:124-126)
__global__ void slic_finalise_kernel(int32_t* __restrict__ centres, Ws w, int B, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * n) return;
    const unsigned long long* s = w.sums + (size_t)t * NFIELD;
    unsigned long long cnt = s[5];
    int32_t* o = centres + (size_t)t * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) o[k] = cnt ? (int32_t)(s[k] / cnt) : 0;
}

__global__ void slic_counts_kernel(int32_t* __restrict__ counts, Ws w, int total) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) counts[t] = (int32_t)w.sums[(size_t)t * NFIELD + 5];
}

// per-label {sum L, sum a, sum b, count} of the last round, in the layout of imgan_rag_accumulate
// ([B][K][4]; labels n..K-1 do not occur: zero)
__global__ void slic_label_sums_kernel(unsigned long long* __restrict__ out, Ws w, int n, int K,
                                       int total) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int b = t / K, s = t - b * K;
    unsigned long long v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    if (s < n) {
        const unsigned long long* src = w.sums + ((size_t)b * n + s) * NFIELD;
        v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[5];
    }
    unsigned long long* dst = out + (size_t)t * 4;
    dst[0] = v0, dst[1] = v1, dst[2] = v2, dst[3] = v3;
}

// 0: 32-bit key, 1: u64 key, 2: literal float64 (non-integral colour weight).  *rel: the
// 32-bit key also fits a signed word once its per-pixel constant is dropped.
int key_mode(int S, double cw, bool* rel) {
    *rel = false;
    double r = rint(cw);
    if (r != cw || cw < 1.0 || cw > 4096.0) return 2;
    double cw2 = cw * cw, S2 = (double)S * S;
    double span = (double)(TW + S) * (TW + S) + (double)(TILE_H_MAX + S) * (TILE_H_MAX + S);
    // (128 below the ends of the range: the assignment kernel rounds keys down to a multiple of 64
    // and keeps its start value above every key)
    bool fits32 = S2 * 195075.0 + cw2 * span < 4294967295.0 - 128.0 &&
                  2.0 * cw2 * (TILE_H_MAX + TW + S) < 8388608.0 && 2.0 * S2 < 8388608.0;
    *rel = fits32 && S2 * 195075.0 + cw2 * span < 2147483647.0 - 128.0;
    return fits32 ? 0 : 1;
}

int validate(const char* fn, int B, int H, int W, int S, int n) {
    if (B < 1 || H < 2 || W < 2) return imgan::fail(imgan::E_INVALID, std::string(fn) + ": bad shape");
    if (S < 1) return imgan::fail(imgan::E_INVALID, std::string(fn) + ": step must be >= 1");
    if (n < 1) return imgan::fail(imgan::E_INVALID, std::string(fn) + ": no centres");
    if ((double)H * W * 3 * B > 9.0e18) return imgan::fail(imgan::E_RANGE, std::string(fn) + ": too large");
    if ((int64_t)B * n > 0x7fffffff) return imgan::fail(imgan::E_RANGE, std::string(fn) + ": too many centres");
    return 0;
}

}  // namespace

extern "C" int imgan_slic_grid_host(int H, int W, int K, int step_in, int* step, int* nx, int* ny) {
    IMGAN_REQUIRE(step && nx && ny, "imgan_slic_grid_host: null output");
    IMGAN_REQUIRE(H > 0 && W > 0, "imgan_slic_grid_host: bad shape");
    int S = step_in;
    if (S <= 0) {
        IMGAN_REQUIRE(K > 0, "imgan_slic_grid_host: num_clusters must be positive");
        S = (int)sqrt((double)(((int64_t)H * W) / K));        // slic.py:29,33
    }
    IMGAN_REQUIRE(S >= 1, "imgan_slic_grid_host: step is 0 (more clusters than pixels)");
    auto count = [](int lo, int hi, int st) { return hi > lo ? (hi - lo + st - 1) / st : 0; };
    *step = S;
    *nx = count(S, W - S / 2, S);                            // slic.py:42
    *ny = count(S, H - S / 2, S);                            // slic.py:43
    return 0;
}

extern "C" size_t imgan_slic_workspace_bytes(int B, int H, int W, int S, int n) {
    if (B < 1 || H < 1 || W < 1 || S < 1 || n < 1) return 0;
    return make_layout(B, H, W, S, n).total;
}

extern "C" int imgan_slic_seed_centres(const uint8_t* lab, int B, int H, int W, int S, int nx,
                                       int ny, int32_t* centres, imgan_stream_t stream) {
    IMGAN_REQUIRE(lab && centres, "imgan_slic_seed_centres: null pointer");
    int n = nx * ny;
    if (int rc = validate("imgan_slic_seed_centres", B, H, W, S, n)) return rc;
    // the 3x3 search reads (x+2, y) and (x, y+2) around the last seed (slic.py:65-66)
    IMGAN_REQUIRE(S + (nx - 1) * S + 2 <= W - 1 && S + (ny - 1) * S + 2 <= H - 1,
                  "imgan_slic_init: seed grid too close to the image border (index out of bounds)");
    int total = B * n;
    hipLaunchKernelGGL(slic_init_kernel, dim3(imgan::cdiv(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, lab, centres, B, H, W, S, nx, ny);
    return imgan::check_launch("slic_init_kernel");
}

extern "C" int imgan_slic_init(const uint8_t* lab, int B, int H, int W, int S, int nx, int ny,
                               int32_t* centres, int32_t* labels, void* workspace,
                               imgan_stream_t stream) {
    IMGAN_REQUIRE(lab && centres && labels && workspace, "imgan_slic_init: null pointer");
    if (int rc = imgan_slic_seed_centres(lab, B, H, W, S, nx, ny, centres, stream)) return rc;
    IMGAN_HIP(hipMemsetAsync(labels, 0xff, (size_t)B * H * W * 4, (hipStream_t)stream));
    return 0;
}

// The last row and the last column of a label map, which no round assigns (slic.py:95,97 stop
// the windows at W - 1 and H - 1): -1 from the start.
__global__ void slic_border_kernel(int32_t* __restrict__ labels, int B, int H, int W) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = H + W;
    if (t >= (long long)B * per) return;
    const int b = (int)(t / per), i = (int)(t - (long long)b * per);
    int32_t* img = labels + (size_t)b * H * W;
    if (i < W)
        img[(size_t)(H - 1) * W + i] = -1;
    else
        img[(size_t)(i - W) * W + (W - 1)] = -1;
}

// Kernel parameters shared by the tile-record and the assignment kernels.
AssignParams make_params(int B, int H, int W, int S, int n, double colour_weight, const Layout& L,
                         int* mode_out, bool* rel_out) {
    AssignParams P;
    P.B = B; P.H = H; P.W = W; P.S = S; P.n = n;
    P.Sd = (double)S;
    P.cw = colour_weight;
    *mode_out = key_mode(S, colour_weight, rel_out);
    P.S2 = (unsigned)(S * S);
    P.cw2 = *mode_out == 2 ? 0u : (unsigned)(colour_weight * colour_weight);
    P.nsub = L.nsub;
    P.ntiles = (long long)B * L.ntx * L.nty;
    // LDS is sized for about 1.5 times the expected number of centres around a tile (a tile that
    // sees more takes the slow path): every 4 KB less is another workgroup per CU
    double m_est = (double)(TW + 2 * S) * (L.tile_h + 2 * S) / ((double)S * S);
    P.maxc = std::min(MAXC, std::max(16, ((int)(1.5 * m_est) + 4 + 7) / 8 * 8));
    P.bbox = nullptr;
    P.first = 0;
    return P;
}

extern "C" int imgan_slic_bin_centres(const uint8_t* lab, const int32_t* centres, int B, int H,
                                      int W, int S, int n, double colour_weight, void* workspace,
                                      imgan_stream_t stream) {
    IMGAN_REQUIRE(lab && centres && workspace, "imgan_slic_bin_centres: null pointer");
    if (int rc = validate("imgan_slic_bin_centres", B, H, W, S, n)) return rc;
    IMGAN_REQUIRE(colour_weight > 0.0 && isfinite(colour_weight),
                  "imgan_slic_bin_centres: colour_weight must be positive");
    Layout L = make_layout(B, H, W, S, n);
    // the tile kernel takes its quotients from float reciprocals, exact below 2^24
    IMGAN_REQUIRE((long long)L.ntx * L.nty < (1 << 24) && (long long)W + S < (1 << 24) &&
                      (long long)H + S < (1 << 24),
                  "imgan_slic_bin_centres: image too large");
    Ws w = bind(workspace, L);
    int mode;
    bool rel;
    AssignParams P = make_params(B, H, W, S, n, colour_weight, L, &mode, &rel);
    hipStream_t st = (hipStream_t)stream;
    if (n <= BIN_MAX && L.ncell <= BIN_MAX && H < (1 << 24) && W < (1 << 24)) {
        hipLaunchKernelGGL(slic_bin_image_kernel, dim3(B), dim3(BIN_THREADS), 0, st, lab, centres, w,
                           H, W, S, n);
    } else {
        // count and fill are adjacent in the layout: one memset clears both
        IMGAN_HIP(hipMemsetAsync(w.count, 0, L.off_start - L.off_count, st));
        int total = B * n;
        hipLaunchKernelGGL(slic_prepare_kernel, dim3(imgan::cdiv(total, 256)), dim3(256), 0, st,
                           lab, centres, w, B, H, W, S, n);
        hipLaunchKernelGGL(slic_scan_kernel, dim3(B), dim3(1024), 0, st, w);
        hipLaunchKernelGGL(slic_fill_kernel, dim3(imgan::cdiv(total, 256)), dim3(256), 0, st, w, B,
                           n);
    }
    // grid: (tiles of one image / 16, images); the y extent of a grid stops at 65535
    hipLaunchKernelGGL(slic_tiles_kernel,
                       dim3((unsigned)imgan::cdiv((long long)L.ntx * L.nty, TILES_PER_WG),
                            (unsigned)std::min(B, 65535), (unsigned)imgan::cdiv(B, 65535)),
                       dim3(256), 0, st, w, P);
    return imgan::check_launch("slic_bin_centres");
}

static int assign_accumulate(const uint8_t* lab, int32_t* labels, int B, int H, int W, int S, int n,
                             double colour_weight, void* workspace, int32_t* bbox_out, bool first,
                             imgan_stream_t stream) {
    IMGAN_REQUIRE(lab && labels && workspace, "imgan_slic_assign_accumulate: null pointer");
    if (int rc = validate("imgan_slic_assign_accumulate", B, H, W, S, n)) return rc;
    IMGAN_REQUIRE(colour_weight > 0.0 && isfinite(colour_weight),
                  "imgan_slic_assign_accumulate: colour_weight must be positive");
    Layout L = make_layout(B, H, W, S, n);
    Ws w = bind(workspace, L);
    int mode;
    bool rel;
    AssignParams P = make_params(B, H, W, S, n, colour_weight, L, &mode, &rel);
    bool aligned = (W % 4 == 0) && (((uintptr_t)lab & 3) == 0) && (((uintptr_t)labels & 15) == 0);
    const int rows = choose_rows(S);
    size_t lds_bytes = (size_t)MAXC * sizeof(int4) + (size_t)P.maxc * 3 * 64 * sizeof(uint32_t) +
                       (size_t)4 * 4 * 64 * 4;
    IMGAN_REQUIRE((P.ntiles + 7) / 8 * 8 < 0x7fffffffLL, "imgan_slic_assign_accumulate: too many tiles");
    dim3 grid((unsigned)((P.ntiles + 7) / 8 * 8));
    hipStream_t st = (hipStream_t)stream;
    if (first) {
        // the incoming map counts as all -1 and is not read: the kernel writes every pixel it
        // owns, the border nobody owns is written here
        P.first = 1;
        hipLaunchKernelGGL(slic_border_kernel,
                           dim3((unsigned)imgan::cdiv((long long)B * (H + W), 256)), dim3(256), 0, st,
                           labels, B, H, W);
    }
    if (bbox_out) {
        P.bbox = bbox_out;
        lds_bytes += (size_t)P.maxc * 2 * 16 * sizeof(uint32_t);
        hipLaunchKernelGGL(slic_bbox_init_kernel, dim3((unsigned)imgan::cdiv(B * n, 256)), dim3(256),
                           0, st, bbox_out, B * n);
    }
#define LAUNCH1(M, A, E, RR) hipLaunchKernelGGL((slic_assign_kernel<M, A, E, RR>), grid, dim3(256), lds_bytes, st, lab, labels, w, P)
#define LAUNCH(M, A, E) do { if (rows == 8) LAUNCH1(M, A, E, 8); else LAUNCH1(M, A, E, 4); } while (0)
    if (mode == 0 && rel) { if (aligned) LAUNCH(0, true, true); else LAUNCH(0, false, true); }
    else if (mode == 0) { if (aligned) LAUNCH(0, true, false); else LAUNCH(0, false, false); }
    else if (mode == 1) { if (aligned) LAUNCH(1, true, false); else LAUNCH(1, false, false); }
    else { if (aligned) LAUNCH(2, true, false); else LAUNCH(2, false, false); }
#undef LAUNCH
#undef LAUNCH1
    return imgan::check_launch("slic_assign_kernel");
}

extern "C" int imgan_slic_assign_accumulate(const uint8_t* lab, int32_t* labels, int B, int H,
                                            int W, int S, int n, double colour_weight,
                                            void* workspace, imgan_stream_t stream) {
    return assign_accumulate(lab, labels, B, H, W, S, n, colour_weight, workspace, nullptr, false,
                             stream);
}

extern "C" int imgan_slic_assign_accumulate_first(const uint8_t* lab, int32_t* labels, int B, int H,
                                                  int W, int S, int n, double colour_weight,
                                                  void* workspace, int32_t* bbox_out,
                                                  imgan_stream_t stream) {
    return assign_accumulate(lab, labels, B, H, W, S, n, colour_weight, workspace, bbox_out, true,
                             stream);
}

extern "C" int imgan_slic_assign_accumulate_bbox(const uint8_t* lab, int32_t* labels, int B, int H,
                                                 int W, int S, int n, double colour_weight,
                                                 void* workspace, int32_t* bbox_out,
                                                 imgan_stream_t stream) {
    IMGAN_REQUIRE(bbox_out, "imgan_slic_assign_accumulate_bbox: null pointer");
    return assign_accumulate(lab, labels, B, H, W, S, n, colour_weight, workspace, bbox_out, false,
                             stream);
}

#ifdef IMGAN_ASSIGN_COUNTS
extern "C" int imgan_debug_assign_counts(unsigned long long* out8, int reset) {
    IMGAN_HIP(hipDeviceSynchronize());
    IMGAN_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(assign_counts), 64));
    if (reset) {
        unsigned long long z[8] = {0};
        IMGAN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(assign_counts), z, 64));
    }
    return 0;
}
#endif

extern "C" int imgan_slic_finalise_centres(int32_t* centres, int B, int H, int W, int S, int n,
                                           void* workspace, imgan_stream_t stream) {
    IMGAN_REQUIRE(centres && workspace, "imgan_slic_finalise_centres: null pointer");
    if (int rc = validate("imgan_slic_finalise_centres", B, H, W, S, n)) return rc;
    Ws w = bind(workspace, make_layout(B, H, W, S, n));
    hipLaunchKernelGGL(slic_finalise_kernel, dim3(imgan::cdiv(B * n, 256)), dim3(256), 0,
                       (hipStream_t)stream, centres, w, B, n);
    return imgan::check_launch("slic_finalise_kernel");
}

extern "C" int imgan_slic_label_counts(int B, int H, int W, int S, int n, const void* workspace,
                                       int32_t* counts, imgan_stream_t stream) {
    IMGAN_REQUIRE(counts && workspace, "imgan_slic_label_counts: null pointer");
    if (int rc = validate("imgan_slic_label_counts", B, H, W, S, n)) return rc;
    Ws w = bind(const_cast<void*>(workspace), make_layout(B, H, W, S, n));
    hipLaunchKernelGGL(slic_counts_kernel, dim3(imgan::cdiv(B * n, 256)), dim3(256), 0,
                       (hipStream_t)stream, counts, w, B * n);
    return imgan::check_launch("slic_counts_kernel");
}

extern "C" int imgan_slic_label_sums(int B, int H, int W, int S, int n, const void* workspace, int K,
                                     uint64_t* sums, imgan_stream_t stream) {
    IMGAN_REQUIRE(sums && workspace, "imgan_slic_label_sums: null pointer");
    IMGAN_REQUIRE(K >= 1 && (long long)B * K < 0x7fffffffLL, "imgan_slic_label_sums: bad label count");
    if (int rc = validate("imgan_slic_label_sums", B, H, W, S, n)) return rc;
    Ws w = bind(const_cast<void*>(workspace), make_layout(B, H, W, S, n));
    hipLaunchKernelGGL(slic_label_sums_kernel, dim3(imgan::cdiv(B * K, 256)), dim3(256), 0,
                       (hipStream_t)stream, (unsigned long long*)sums, w, n, K, B * K);
    return imgan::check_launch("slic_label_sums_kernel");
}

extern "C" int imgan_slic_run(const uint8_t* lab, int B, int H, int W, int S, int nx, int ny,
                              double colour_weight, int num_iter, int32_t* centres,
                              int32_t* labels, void* workspace, imgan_stream_t stream) {
    int n = nx * ny;
    if (num_iter <= 1)                                       // no round: the map stays all -1
        return imgan_slic_init(lab, B, H, W, S, nx, ny, centres, labels, workspace, stream);
    IMGAN_REQUIRE(labels && workspace, "imgan_slic_run: null pointer");
    if (int rc = imgan_slic_seed_centres(lab, B, H, W, S, nx, ny, centres, stream)) return rc;
    for (int it = 1; it < num_iter; ++it) {                  // slic.py:84 -> num_iter - 1 rounds
        if (int rc = imgan_slic_bin_centres(lab, centres, B, H, W, S, n, colour_weight, workspace,
                                            stream))
            return rc;
        // the first round writes the whole map (no fill before it)
        if (int rc = assign_accumulate(lab, labels, B, H, W, S, n, colour_weight, workspace,
                                       nullptr, it == 1, stream))
            return rc;
        if (int rc = imgan_slic_finalise_centres(centres, B, H, W, S, n, workspace, stream))
            return rc;
    }
    return 0;
}
