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
//
// Two front ends feed the same body: the box gather (imgan_dcd_segments / _boxed: every
// workgroup scans its segment's bounding box; no memory beyond the boxes) and the pixel lists
// (imgan_dcd_segments_listed: one pass sorts all pixels by segment first; 8 B of scratch per
// pixel, every segment dispatched to the tier that fits it).
#include "common.h"
#include "../../include/imgan.h"
#include "dcd_common.h"
#include "dcd_lists.h"
#include <stdlib.h>

using namespace imgan_dcd;

namespace {

// Capacity tiers: most segments (superpixels) fit the small tier, which leaves room for two
// workgroups per CU; the few larger ones go to the large tier (one workgroup per CU); only what
// exceeds that goes to the streaming path.  The listed mode, which knows every segment's size
// before it starts, adds a tier below the small one (three workgroups per CU).
#ifdef IMGAN_SEG_TIMING
// phase clocks of thread 0, summed over blocks (tools/seg_timing.py; never built into the product)
__device__ unsigned long long seg_timing[16];
#define SEG_MARK(i)                                                         \
    if (tid == 0) {                                                         \
        unsigned long long now_ = wall_clock64();                           \
        atomicAdd(&seg_timing[i], now_ - t_last_);                          \
        t_last_ = now_;                                                     \
    }
#else
#define SEG_MARK(i)
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The small tier packs {count, sum0} and {sum1, sum2} into 16-bit halves of two words per
// lane-private copy: a copy sees at most CAP / 64 pixels, and 192 * 255 < 65536.
static_assert((CAP_SMALL / 64) * 255 < 65536, "packed accumulators would overflow");

// bbox: int [nsegs][4] = xmin, ymin, xmax, ymax.  One block = 256 columns x BBOX_ROWS rows
// (a block per row was bound by the workgroup dispatch rate, not by memory).  Each wave walks
// down its 64 columns; per row the extremes of every label run come from a ballot and go into
// a small LDS table keyed by label (fire-and-forget LDS atomics: nothing waits on global
// memory in the loop); the table is flushed with one global atomic per entry and field.
constexpr int BBOX_ROWS = 32, BBOX_SLOTS = 128;

struct BboxTable {
    int key[BBOX_SLOTS];
    int ext[BBOX_SLOTS][4];
};

__device__ __forceinline__ void bbox_global(int* __restrict__ bb, int xa, int ya, int xb, int yb) {
    atomicMin(&bb[0], xa);
    atomicMin(&bb[1], ya);
    atomicMax(&bb[2], xb);
    atomicMax(&bb[3], yb);
}

__device__ __forceinline__ void bbox_row(int pending, int x, int y, int lane, int nseg, int b,
                                         BboxTable& T, int* __restrict__ bbox) {
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
            const int xa = x - lane + first, xb = x - lane + last;
            int idx = (int)(((uint32_t)cur * 0x9e3779b1u) >> 25);
            bool placed = false;
            for (int probe = 0; probe < BBOX_SLOTS; ++probe) {
                const int old = atomicCAS(&T.key[idx], -1, cur);
                if (old == -1 || old == cur) {
                    placed = true;
                    break;
                }
                idx = (idx + 1) & (BBOX_SLOTS - 1);
            }
            if (placed) {
                atomicMin(&T.ext[idx][0], xa);
                atomicMin(&T.ext[idx][1], y);
                atomicMax(&T.ext[idx][2], xb);
                atomicMax(&T.ext[idx][3], y);
            } else {                                    // more labels than slots in this block
                bbox_global(bbox + ((size_t)b * nseg + cur) * 4, xa, y, xb, y);
            }
        }
        if (mine) pending = -1;
    }
}

__global__ __launch_bounds__(256) void seg_bbox_kernel(const int32_t* __restrict__ labels, int H,
                                                       int W, int nseg, int* __restrict__ bbox) {
    __shared__ BboxTable T;
    const int b = blockIdx.z;
    const int y0 = blockIdx.y * BBOX_ROWS;
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int xc = x < W ? x : W - 1;
    if (threadIdx.x < BBOX_SLOTS) {
        T.key[threadIdx.x] = -1;
        T.ext[threadIdx.x][0] = T.ext[threadIdx.x][1] = 0x7fffffff;
        T.ext[threadIdx.x][2] = T.ext[threadIdx.x][3] = -1;
    }
    __syncthreads();
    const int32_t* col = labels + (size_t)b * H * W + xc;
    for (int r = 0; r < BBOX_ROWS; r += 4) {
        int l[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                  // 4 rows in flight, clamped not branched
            int y = y0 + r + u;
            l[u] = col[(size_t)(y < H ? y : H - 1) * W];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int y = y0 + r + u;
            bool ok = x < W && y < H && l[u] >= 0 && l[u] < nseg;
            bbox_row(ok ? l[u] : -1, x, y, lane, nseg, b, T, bbox);
        }
    }
    __syncthreads();
    if (threadIdx.x < BBOX_SLOTS && T.key[threadIdx.x] >= 0)
        bbox_global(bbox + ((size_t)b * nseg + T.key[threadIdx.x]) * 4, T.ext[threadIdx.x][0],
                    T.ext[threadIdx.x][1], T.ext[threadIdx.x][2], T.ext[threadIdx.x][3]);
}

__global__ void seg_bbox_init_kernel(int* __restrict__ bbox, int nsegs) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nsegs) return;
    int* bb = bbox + (size_t)t * 4;
    bb[0] = 0x7fffffff;
    bb[1] = 0x7fffffff;
    bb[2] = -1;
    bb[3] = -1;
}

// Room a tier needs for a segment: its pixels, and the 2-byte cell grid of the coherence pass over
// the bounding box (which also keeps cell offsets within 16 bits).
__device__ __forceinline__ bool tier_fits(long long n, long long area, int cap) {
    return cap > 0 && n <= cap && area <= 65535 && area * 2 + 8 <= (long long)cap * 6;
}

// One workgroup per image: running sum of its counts -> the runs of its segments inside the
// image's own H*W slab of the lists, cursors cleared, every segment put on the list of the
// smallest tier that holds it (or counted in n_big: left to the streaming path).  hdr is zero
// on entry.
__global__ __launch_bounds__(1024) void seg_plan_kernel(const int32_t* __restrict__ counts,
                                                        const int* __restrict__ bbox, int nseg,
                                                        int H, int W, SegLists L, int cap0,
                                                        int cap1, int cap2,
                                                        int* __restrict__ n_big) {
    __shared__ unsigned long long wave_sum[16];
    __shared__ unsigned long long carry;
    __shared__ int fill[4], first[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    if (tid < 4) fill[tid] = 0;
    const unsigned long long slab = (unsigned long long)H * W, slab_at = slab * b;
    if (tid == 0) carry = 0ull;
    __syncthreads();
    for (int base = 0; base < nseg; base += 1024) {
        const int s = base + tid;
        const size_t i = (size_t)b * nseg + s;
        const int c = s < nseg ? max(counts[i], 0) : 0;
        unsigned long long incl = (unsigned long long)c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        unsigned long long before = carry;
        for (int w = 0; w < wave; ++w) before += wave_sum[w];
        const unsigned long long start = before + incl - (unsigned long long)c;
        if (s < nseg) {
            const unsigned long long s0 = start < slab ? start : slab;
            const unsigned long long s1 = start + c < slab ? start + c : slab;
            L.offsets[i] = (uint32_t)(slab_at + s0);
            L.ends[i] = (uint32_t)(slab_at + s1);
            L.cursor[i] = 0u;
            const long long n = (long long)(s1 - s0);
            const int* bb = bbox + i * 4;
            const int xmin = max(bb[0], 0), ymin = max(bb[1], 0);
            const int xmax = min(bb[2], W - 1), ymax = min(bb[3], H - 1);
            const bool empty = xmax < xmin || ymax < ymin || n == 0;
            // (the grid of the coherence pass: the box with a border of one cell on three sides --
            // dcd_tiers.hip lays it out that way, the plain body uses the box alone)
            const long long area = empty ? 0 : (long long)(xmax - xmin + 3) * (ymax - ymin + 2);
            const int t = tier_fits(n, area, cap0) ? 0 : tier_fits(n, area, cap1) ? 1
                          : tier_fits(n, area, cap2) ? 2 : 3;
            // place on its tier's list, for the moment relative to this image's share of it
            L.cursor[i] = ((uint32_t)t << 28) | (uint32_t)atomicAdd(&fill[t], 1);
        }
        __syncthreads();
        if (tid == 1023) carry = before + incl;
        __syncthreads();
    }
    // The image's segments stay together on every list (one reservation per image and tier):
    // workgroups that run side by side then work on neighbouring segments, whose assignment bytes
    // share cache lines -- with the lists interleaved by image the stage measured 4 % slower.
    if (tid < 3) first[tid] = atomicAdd(&L.hdr[tid], fill[tid]);
    if (tid == 3 && fill[3]) atomicAdd(n_big, fill[3]);
    __syncthreads();
    for (int s = tid; s < nseg; s += 1024) {
        const size_t i = (size_t)b * nseg + s;
        const uint32_t v = L.cursor[i];                 // written by this thread above
        const int t = (int)(v >> 28);
        if (t < 3) L.tier[t][first[t] + (int)(v & 0xfffffffu)] = (int)i;
        L.cursor[i] = 0u;
    }
}

// The pass that fills the runs (COUNT_ONLY: just counts the pixels of every label, for callers
// that have no counts).  One block = SC_WAVES rows x 512 columns: a wave takes a row, a lane
// SC_PX consecutive pixels of it (two 16-byte loads).
//   ranking: a piece that is one run of one label (the rule) joins the equal pieces of the lanes
//     beside it; the first lane of such a run of lanes enters the label in a small LDS table and
//     reserves for all of them with one returning LDS atomic; a piece with a boundary in it does
//     that run by run for itself.  About 20 vector instructions per pixel.  (Ranking wave row by
//     wave row with one ballot loop per distinct label of the row took 119 per pixel, 90 % of
//     that form's time.)
//   reservation: one global atomic per label and block takes the block's share of the segment's run.
//   output: the positions go to an LDS buffer in (label, rank) order, and every label's share is
//     copied out by the whole block with consecutive lanes on consecutive entries: whole sectors.
//     (Each lane storing its own eight entries, 32 bytes apart from its neighbour's, held the
//     pass at 2.2 TB/s.)
constexpr int SC_PX = 8, SC_WAVES = 4, SC_BLOCK_ROWS = SC_WAVES, SC_BLOCK_W = 64 * SC_PX, SC_SLOTS = 128;
constexpr int SC_THREADS = 64 * SC_WAVES, SC_BLOCK_PX = SC_THREADS * SC_PX;

struct ScatterTable {
    int key[SC_SLOTS];
    uint32_t cnt[SC_SLOTS];
    uint32_t base[SC_SLOTS];
    uint32_t end[SC_SLOTS];
    uint32_t off[SC_SLOTS];     // where the label's share starts in the staging buffer
    int active[SC_SLOTS];       // the slots in use, closed up
    int nactive, wave0_total, wave0_active;
};

template <bool COUNT_ONLY>
__global__ __launch_bounds__(SC_THREADS) void seg_scatter_kernel(
    const int32_t* __restrict__ labels, int B, int H, int W, int nseg, SegLists L,
    int32_t* __restrict__ counts) {
    __shared__ ScatterTable T;
    __shared__ uint32_t staged[COUNT_ONLY ? 1 : SC_BLOCK_PX];
    const int b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int y = blockIdx.y * SC_BLOCK_ROWS + wave;
    const int x0 = blockIdx.x * SC_BLOCK_W + lane * SC_PX;
    if (tid < SC_SLOTS) {
        T.key[tid] = -1;
        T.cnt[tid] = 0u;
    }
    static_assert(SC_PX == 8, "a piece is loaded as two int4");
    static_assert(SC_SLOTS == 128 && SC_THREADS >= 128, "the table is closed up by two waves");
    int cur[SC_PX];
    {
        // clamped, not branched; 16-byte loads where the piece is aligned and whole
        const int yc = y < H ? y : H - 1;
        const int32_t* row = labels + ((size_t)b * H + yc) * W;
        if ((W & 3) == 0 && x0 + SC_PX <= W && (reinterpret_cast<uintptr_t>(labels) & 15) == 0) {
            const int4 lo = *reinterpret_cast<const int4*>(row + x0);
            const int4 hi = *reinterpret_cast<const int4*>(row + x0 + 4);
            cur[0] = lo.x, cur[1] = lo.y, cur[2] = lo.z, cur[3] = lo.w;
            cur[4] = hi.x, cur[5] = hi.y, cur[6] = hi.z, cur[7] = hi.w;
        } else {
#pragma unroll
            for (int u = 0; u < SC_PX; ++u) cur[u] = row[min(x0 + u, W - 1)];
        }
#pragma unroll
        for (int u = 0; u < SC_PX; ++u) {
            const bool ok = x0 + u < W && y < H && cur[u] >= 0 && cur[u] < nseg;
            cur[u] = ok ? cur[u] : -1;
        }
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long upto = below | (1ull << lane);          // lanes 0 .. this one
    // table slot of a label (entered if new), 255 when the table is full
    auto slot_of = [&](int c) {
        int probe_at = (int)(((uint32_t)c * 0x9e3779b1u) >> 25);
        for (int probe = 0; probe < SC_SLOTS; ++probe) {
            const int old = atomicCAS(&T.key[probe_at], -1, c);
            if (old == -1 || old == c) return probe_at;
            probe_at = (probe_at + 1) & (SC_SLOTS - 1);
        }
        return 255;
    };
    // where[u]: table slot | rank in the block's share << 8; -1: no pixel; slot 255: table full
    int len_at[SC_PX], where[SC_PX];
    {
        int run_end = SC_PX;
#pragma unroll
        for (int u = SC_PX - 1; u >= 0; --u) {
            if (u < SC_PX - 1 && cur[u + 1] != cur[u]) run_end = u + 1;
            len_at[u] = run_end - u;                      // read at the first pixel of a run only
        }
    }
    {
        const bool whole = len_at[0] == SC_PX && cur[0] >= 0;
        const int c0 = whole ? cur[0] : -1 - lane;                   // never equal to a neighbour's
        const int left = __shfl_up(c0, 1);
        const bool head = lane == 0 || c0 != left;
        const unsigned long long heads = __ballot(head);             // bit 0 is always set
        const int first_lane = 63 - __clzll((long long)(heads & upto));
        const unsigned long long later = heads & ~upto;
        const int next_head = later ? __ffsll((long long)later) - 1 : 64;
        int packed = -1;
        if (whole && head) {
            const int idx = slot_of(c0);
            packed = idx;
            if (idx != 255)
                packed |= (int)atomicAdd(&T.cnt[idx], (uint32_t)((next_head - lane) * SC_PX)) << 8;
        }
        packed = __shfl(packed, first_lane);
        if (whole) {
            const int w0 = (packed & 255) == 255 ? 255 : packed + (((lane - first_lane) * SC_PX) << 8);
#pragma unroll
            for (int u = 0; u < SC_PX; ++u) where[u] = w0 == 255 ? 255 : w0 + (u << 8);
        } else {
#pragma unroll
            for (int u = 0; u < SC_PX; ++u) {
                const int c = cur[u];
                const bool first_px = u == 0 || c != cur[u > 0 ? u - 1 : 0];
                int w = -1;
                if (c >= 0) {
                    if (first_px) {
                        const int idx = slot_of(c);
                        w = idx;
                        if (idx != 255) w |= (int)atomicAdd(&T.cnt[idx], (uint32_t)len_at[u]) << 8;
                    } else {
                        const int before = where[u > 0 ? u - 1 : 0];
                        w = (before & 255) == 255 ? 255 : before + 256;
                    }
                }
                where[u] = w;
            }
        }
    }
    __syncthreads();
    // ---- the block's share of every label's run; where the shares start in the staging buffer
    if (tid < SC_SLOTS) {
        const bool used = T.key[tid] >= 0;
        const uint32_t n = used ? T.cnt[tid] : 0u;
        if (used) {
            const size_t gs = (size_t)b * nseg + T.key[tid];
            if (COUNT_ONLY) {
                atomicAdd(&counts[gs], (int)n);
            } else {
                T.base[tid] = L.offsets[gs] + atomicAdd(&L.cursor[gs], n);
                T.end[tid] = L.ends[gs];
            }
        }
        if (!COUNT_ONLY) {
            uint32_t incl = n;                              // running sum over the wave's 64 slots
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
                if (lane >= o) incl += up;
            }
            const unsigned long long usedm = __ballot(used);
            if (wave == 0 && lane == 63) T.wave0_total = (int)incl, T.wave0_active = __popcll(usedm);
            // (the second wave waits for the first one's totals at the barrier below)
            T.off[tid] = incl - n;                          // wave 1: plus wave 0's total, added below
            if (used) T.active[(wave == 0 ? 0 : 64) + __popcll(usedm & below)] = tid | (wave << 16);
            if (wave == 1 && lane == 63) T.nactive = __popcll(usedm);
        }
    }
    __syncthreads();
    if (!COUNT_ONLY) {
        // ---- stage the positions in (label, rank) order
        const int add1 = T.wave0_total;
#pragma unroll
        for (int u = 0; u < SC_PX; ++u) {
            const int w = where[u];
            if (w >= 0 && (w & 255) != 255) {
                const int slot = w & 255;
                const uint32_t at = T.off[slot] + (slot >= 64 ? (uint32_t)add1 : 0u) + (uint32_t)(w >> 8);
                if (at < (uint32_t)SC_BLOCK_PX) staged[at] = ((uint32_t)y << 16) | (uint32_t)(x0 + u);
            }
        }
    }
    __syncthreads();
    if (!COUNT_ONLY) {
        // ---- copy every label's share out, consecutive lanes on consecutive entries
        const int n0 = T.wave0_active, n1 = T.nactive, add1 = T.wave0_total;
        for (int k = 0; k < n0 + n1; ++k) {
            const int slot = (k < n0 ? T.active[k] : T.active[64 + k - n0]) & 0xffff;
            const uint32_t n = T.cnt[slot], from = T.off[slot] + (slot >= 64 ? (uint32_t)add1 : 0u);
            const unsigned long long to = T.base[slot], end = T.end[slot];
            for (uint32_t i = tid; i < n; i += SC_THREADS)
                if (to + i < end && from + i < (uint32_t)SC_BLOCK_PX) L.sxy[to + i] = staged[from + i];
        }
    }
    // ---- more labels in the block than table slots: these pixels reserve their places one by one
#pragma unroll
    for (int u = 0; u < SC_PX; ++u) {
        const uint32_t xy = ((uint32_t)y << 16) | (uint32_t)(x0 + u);
        int pending = (where[u] >= 0 && (where[u] & 255) == 255) ? cur[u] : -1;
        while (true) {
            const unsigned long long act = __ballot(pending >= 0);
            if (act == 0ull) break;
            const int leader = __ffsll((long long)act) - 1;
            const int sel = __shfl(pending, leader);
            const bool mine = pending == sel;
            const unsigned long long who = __ballot(mine);
            const size_t gs = (size_t)b * nseg + sel;
            if (COUNT_ONLY) {
                if (lane == leader) atomicAdd(&counts[gs], __popcll(who));
            } else {
                uint32_t first = 0;
                if (lane == leader) first = atomicAdd(&L.cursor[gs], (uint32_t)__popcll(who));
                first = (uint32_t)__shfl((int)first, leader);
                if (mine) {
                    const unsigned long long dst =
                        (unsigned long long)L.offsets[gs] + first + (uint32_t)__popcll(who & below);
                    if (dst < L.ends[gs]) L.sxy[dst] = xy;
                }
            }
            if (mine) pending = -1;
        }
    }
}

// mpeg7Descriptors.py:101 for the clusters in `empty` (a wave-uniform mask), by one whole wave while
// the others wait at the round's barrier: rare (a cluster only runs empty when sampled colours
// coincide or a cluster is crowded out).  Kept out of line on purpose: inlined into the fold, the
// 16-cluster kernels built by ROCm 7.2 never leave the pixel loop below (the same loop as a
// callee, and with a constant bound inlined, run fine; the ISA reads correct -- cause not found),
// and the call costs 1 % of the stage.  (The tier kernels spill several hundred scalar registers,
// and builds that raise the pressure a little more have come out wrong with this call present and
// right without it -- until the wave index was declared uniform: see "Known fragility" in DESIGN.md
// before touching the register budget.)
// Writes the drawn pixel's colour to cen[j], cen[stride + j], cen[2 * stride + j].
__device__ __noinline__ void resample_empty(unsigned long long empty, const uint32_t* pix,
                                            const uint16_t* pos, int n, int bw, int xmin, int ymin,
                                            int W, uint32_t seed, float* cen, int stride) {
    const int lane = threadIdx.x & 63;
    while (empty) {
        const int j = __ffsll((long long)empty) - 1;
        empty &= empty - 1;
        unsigned long long best = ~0ull;
        uint32_t colour = 0;
        for (int i = lane; i < n; i += 64) {
            const int o = pos[i], yy = o / bw, xx = o - yy * bw;
            const unsigned long long key =
                resample_key((uint32_t)((ymin + yy) * W + xmin + xx), seed, j);
            if (key < best) {
                best = key;
                colour = pix[i] & 0xffffffu;
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o);
            const uint32_t other_colour = (uint32_t)__shfl_xor((int)colour, o);
            if (other < best) {
                best = other;
                colour = other_colour;
            }
        }
        if (lane == 0 && best != ~0ull) {
            cen[j] = (float)(colour & 255u);
            cen[stride + j] = (float)((colour >> 8) & 255u);
            cen[2 * stride + j] = (float)((colour >> 16) & 255u);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// One segment, start to finish, by the whole workgroup.  MODE_FIRST: the small tier visits every
// segment, gathers its pixels from its bounding box and puts those that do not fit on the retry
// list; MODE_RETRY: the large tier works through that list and counts what is still too large
// (left to the streaming path); MODE_LISTED: the pixels come from the segment's run of the pixel
// lists, and the plan kernel has already chosen a tier that holds them.
// (The dedup tiers, which run the rounds over the distinct colours of a segment, are dcd_tiers.hip;
// a segment whose colours fit none of their tables comes here, MODE_LISTED on list 3.)
constexpr int MODE_FIRST = 0, MODE_RETRY = 1, MODE_LISTED = 2;

template <int KM, int CAP, int THREADS, int MODE>
__device__ __forceinline__ void segment_body(
    unsigned char* smem, const int b, const int s, const int B, const uint8_t* __restrict__ lab,
    const int32_t* __restrict__ labels, int H, int W, int nseg, int k, int use_hash, uint32_t seed,
    float* __restrict__ centres, uint8_t* __restrict__ assign, SegState* __restrict__ seg,
    unsigned long long* __restrict__ stats, const int* __restrict__ bbox, int* __restrict__ retry,
    int* __restrict__ big_count, int max_rounds, const SegLists& L, int tier) {
    constexpr bool FIRST = MODE == MODE_FIRST;
    constexpr bool LISTED = MODE == MODE_LISTED;
    uint32_t* pix = reinterpret_cast<uint32_t*>(smem);                    // Lab | assign << 24
    uint16_t* pos = reinterpret_cast<uint16_t*>(pix + CAP);               // offset in the bbox
    constexpr bool PACKED = (CAP / 64) * 255 < 65536;
    constexpr int WORDS = PACKED ? 2 : 4;                                 // per cluster and copy
    constexpr int ACCW = WORDS > 3 ? WORDS : 3;                           // statistics use 3
    uint32_t* acc = reinterpret_cast<uint32_t*>(pos + CAP);               // [KM*ACCW][64] lane-private
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(acc + KM * ACCW * 64);
    float* cen = reinterpret_cast<float*>(slots + KM);                   // [KM][3]
    float* old = cen + KM * 3;                                            // [KM][3]
    uint32_t* red = reinterpret_cast<uint32_t*>(old + KM * 3);           // [KM][4] count, sums
    float* sbuf = reinterpret_cast<float*>(red + KM * 4);                 // [3][KM] shifts per cluster
    int* flags = reinterpret_cast<int*>(sbuf + KM * 3);                   // no static LDS: keeps
    int& s_n = flags[0];                                                  // the dynamic base 16-B aligned
    int& s_done = flags[1];

    const size_t gs = (size_t)b * nseg + s;
    int tid_ = threadIdx.x;
    // the listed mode calls this in a loop: without the fence the compiler computes everything that
    // depends on the thread index alone once, ahead of the loop, and keeps it all in registers
    if (LISTED) asm volatile("" : "+v"(tid_));
    // wave: known to the compiler as wave uniform, so that `if (wave == 0)` is a scalar branch.  The
    // fence above hides that from it; with the serial section of a round behind a *divergent*
    // branch, builds under a little more register pressure than this one came out wrong (DESIGN.md,
    // "Known fragility")
    const int tid = tid_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef IMGAN_SEG_TIMING
    unsigned long long t_last_ = wall_clock64();
#endif
    if (MODE == MODE_RETRY && seg[gs].handled) return;
    const int* bb = bbox + gs * 4;
    // clamped into the image: the boxes may come from the caller (imgan_dcd_segments_boxed), and a
    // box outside the image must not turn into a read outside the buffers
    const int xmin = max(bb[0], 0), ymin = max(bb[1], 0);
    const int xmax = min(bb[2], W - 1), ymax = min(bb[3], H - 1);
    SegState st;
    st.old_sum = 0.0;
    st.done = 1;
    st.rounds = 0;
    st.handled = 1;
    st.pending = 0;
    unsigned long long* out = stats + gs * k * 8;
    uint32_t run_at = 0, run_n = 0;
    if (LISTED) {
        run_at = L.offsets[gs];
        run_n = min(L.ends[gs] - run_at, L.cursor[gs]);
    }
    if (xmax < xmin || ymax < ymin || (LISTED && run_n == 0)) {   // segment id without pixels
        if (tid == 0) seg[gs] = st;
        for (int i = tid; i < k * 8; i += THREADS) out[i] = 0ull;
        return;
    }
    const int bw = xmax - xmin + 1, bh = ymax - ymin + 1;
    const int area = bw * bh;
    const size_t img = (size_t)b * H * W;
    // positions are kept as 16-bit offsets, and the coherence pass lays a 2-byte cell grid of
    // the box over the pixel store
    bool big = (long long)bw * bh > 65535 || (long long)bw * bh * 2 + 8 > (long long)CAP * 6;
    if (LISTED) {
        // ---- the segment's run, straight into LDS.  (A box that does not hold the pixel -- boxes
        // and counts come from the caller -- must not turn into an offset outside the cell grid.)
        if (tid == 0) {
            s_n = (int)min(run_n, (uint32_t)CAP);
            s_done = 0;
        }
        if (tid < KM) slots[tid] = ~0ull;
        __syncthreads();
        const int nrun = (int)min(run_n, (uint32_t)CAP);
        const bool kpow2 = (k & (k - 1)) == 0;
        constexpr int LU = 4;                          // loads in flight per thread (8 and 16 measured: no gain)
        for (int i0 = 0; i0 < nrun; i0 += THREADS * LU) {
            uint32_t v[LU], xy[LU];
#pragma unroll
            for (int u = 0; u < LU; ++u) xy[u] = L.sxy[run_at + min(i0 + u * THREADS + tid, nrun - 1)];
#pragma unroll
            for (int u = 0; u < LU; ++u)                   // the colours, by position
                v[u] = lab_at(lab, img, (int)(xy[u] & 0xffffu), (int)(xy[u] >> 16), H, W,
                              (size_t)B * H * W - 1);
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                const int i = i0 + u * THREADS + tid;
                if (i >= nrun) continue;
                const int x = min(max((int)(xy[u] & 0xffffu), xmin), xmax);
                const int y = min(max((int)(xy[u] >> 16), ymin), ymax);
                pix[i] = v[u] & 0xffffffu;
                pos[i] = (uint16_t)(__mul24(y - ymin, bw) + (x - xmin));
                if (use_hash) {
                    const uint32_t p = (uint32_t)(__mul24(y, W) + x);
                    const uint32_t h = mix32(p * 0x9e3779b1u + seed);
                    const unsigned long long key = ((unsigned long long)h << 32) | (unsigned long long)p;
                    unsigned long long* sl = &slots[kpow2 ? (h & (uint32_t)(k - 1)) : h % (uint32_t)k];
                    if (key < *sl) atomicMin(sl, key);
                }
            }
        }
        SEG_MARK(7)
        __syncthreads();
        SEG_MARK(1)
        big = false;
    } else if (!big) {
        if (tid == 0) s_n = 0;
        if (tid < KM) slots[tid] = ~0ull;
        __syncthreads();
        SEG_MARK(0)
        // ---- gather the segment's pixels.  Every load of a batch is issued before the first
        // use: addresses are clamped into the box instead of branching around the loads (a
        // load under a branch makes the compiler wait for it at the join, which serialises
        // the batch and leaves the gather bound by 8 memory latencies per step).
        const float inv_bw = 1.0f / (float)bw;
        const size_t last_px = (size_t)B * H * W - 1;
        const bool kpow2 = (k & (k - 1)) == 0;
        constexpr int GU = 8;                           // cells per thread and step
        for (int i0 = 0; i0 < area; i0 += THREADS * GU) {
            int lbl[GU], off[GU];
            uint32_t pp[GU], raw[GU], sh[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int i = i0 + u * THREADS + tid;
                off[u] = i;
                const int ic = i < area ? i : area - 1;
                int yy = (int)((float)ic * inv_bw);
                // 24-bit multiplies: box cells < 2^16, rows < 2^16, W < 2^23 (checked by the host);
                // the 32-bit multiply of this chip runs at a quarter of their rate
                int xx = ic - __mul24(yy, bw);
                yy += (xx >= bw) - (xx < 0);            // the float quotient can be off by one
                xx = ic - __mul24(yy, bw);
                pp[u] = (uint32_t)(__mul24(ymin + yy, W) + xmin + xx);
                lbl[u] = labels[img + pp[u]];
                // the three Lab bytes with one (unaligned) dword load; the very last pixel of
                // the batch is read one byte early so that the load stays inside the buffer
                const size_t gp = img + pp[u];
                sh[u] = gp == last_px ? 1u : 0u;
                raw[u] = *reinterpret_cast<const u32_unaligned*>(lab + ((gp << 1) + gp) - sh[u]);
            }
            // one LDS atomic per wave and step: lane 0 reserves the slots of all GU ballots
            unsigned long long who[GU];
            int total = 0;
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                if (off[u] >= area) lbl[u] = -2;
                who[u] = __ballot(lbl[u] == s);
                total += __popcll(who[u]);
            }
            int base = 0;
            if (total) {                                // wave uniform
                if (lane == 0) base = atomicAdd(&s_n, total);
                base = __builtin_amdgcn_readfirstlane(base);
            }
            const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int slot = base + __popcll(who[u] & below);
                base += __popcll(who[u]);
                if (lbl[u] != s) continue;
                const uint32_t v = (raw[u] >> (8u * sh[u])) & 0xffffffu;
                if (slot < CAP) {
                    pix[slot] = v;
                    pos[slot] = (uint16_t)off[u];
                }
                if (use_hash) {
                    // stand-in for random.sample (:79): slot h % k goes to the smallest (h, p)
                    const uint32_t h = mix32(pp[u] * 0x9e3779b1u + seed);
                    const unsigned long long key =
                        ((unsigned long long)h << 32) | (unsigned long long)pp[u];
                    unsigned long long* sl = &slots[kpow2 ? (h & (uint32_t)(k - 1)) : h % (uint32_t)k];
                    if (key < *sl) atomicMin(sl, key);
                }
            }
        }
        SEG_MARK(7)
        __syncthreads();
        SEG_MARK(1)
        big = s_n > CAP;
        if (!big && s_n == 0) {                         // a box (the caller's) without the label in it
            if (tid == 0) seg[gs] = st;
            for (int i = tid; i < k * 8; i += THREADS) out[i] = 0ull;
            return;
        }
    }
    if (big) {                                          // too large for this tier
        if (tid == 0) {
            st.done = 0;
            st.handled = 0;
            seg[gs] = st;
            if (FIRST)
                retry[1 + atomicAdd(&retry[0], 1)] = (int)gs;    // for the large tier
            else
                atomicAdd(big_count, 1);                         // left to the streaming path
        }
        return;
    }
    const int n = s_n;
    const int nloop = n;

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
        cen[tid] = c0;
        cen[KM + tid] = c1;
        cen[2 * KM + tid] = c2;
        old[tid] = old[KM + tid] = old[2 * KM + tid] = 0.f;
    } else if (tid < KM) {
        // unused slots sit far away from every 8-bit colour, so the loop over all KM slots
        // never picks them and needs no bound check
        cen[tid] = cen[KM + tid] = cen[2 * KM + tid] = 1.0e18f;
    }
    if (tid == 0) s_done = 0;
    double old_sum = 0.0;              // only wave 0 uses its copy
    int rounds = 0;
    // wave 0, lane j < k: centre j and its value one round ago (the serial section of a round)
    float wc0 = 0.f, wc1 = 0.f, wc2 = 0.f, wo0 = 0.f, wo1 = 0.f, wo2 = 0.f;
    if (tid < k) {                     // written by this very thread above
        wc0 = cen[tid];
        wc1 = cen[KM + tid];
        wc2 = cen[2 * KM + tid];
    }
    // a q within 2^-15 of the smallest one replays the literal rule.  What the fast loop leaves
    // out, relative to q, for each of the two values compared: up to 2^-19 for the mantissa bits
    // that carry the cluster index (twice: they are not masked off for the comparison either),
    // 2^-22 for the fused multiply-adds, 2^-23 for the sqrt rounding -- under 2^-17 together.
    const float NEAR = 1.0f - 3.0517578125e-05f;           // 1 - 2^-15
    constexpr uint32_t IDXM = KM - 1;

    // ---- Lloyd rounds (:85-107).  Two barriers per round: all waves assign, then the first
    // wave alone folds the lane-private sums (clearing them for the next round), moves the
    // centres and applies the stop rule while the CU's other workgroup keeps the SIMDs busy.
    for (int i = tid; i < k * WORDS * 64; i += THREADS) acc[i] = 0;
    __syncthreads();
    while (true) {
        // centres in registers (plain fp32 operations: a packed v_pk_*_f32 takes the issue slots of
        // the two operations it replaces on this chip, and dependent ones need a wait state on top)
        float c0[KM], c1[KM], c2[KM];
#pragma unroll
        for (int j = 0; j < KM; ++j) {
            c0[j] = cen[j];
            c1[j] = cen[KM + j];
            c2[j] = cen[2 * KM + j];
        }
        for (int i = tid; i < nloop; i += THREADS) {
            const uint32_t v = pix[i];
            const uint32_t weight = 1u;
            const uint32_t u0 = v & 255u, u1 = (v >> 8) & 255u, u2 = (v >> 16) & 255u;
            const float f0 = (float)u0, f1 = (float)u1, f2 = (float)u2;
            // The reference compares sqrt-rounded distances with '<' (:88-91): the winner is the
            // first minimum of sqrt_rn(q).  The first minimum of q itself is the same cluster
            // unless another q lies within a few ulp of the smallest one, so the fast loop
            // tracks the two smallest q and only a near tie replays the literal rule.
            // q >= +0, so the float order is the order of the bit patterns: the cluster index
            // rides in the low mantissa bits (IDXM + 1 ulp of slack, covered by NEAR) and one
            // unsigned min / med3 per cluster tracks the smallest and second smallest q.
            uint32_t best = 0xffffffffu, second = 0xffffffffu;
#pragma unroll
            for (int j = 0; j < KM; ++j) {
                // fused multiply-adds: q differs from the literal (d0^2 + d1^2) + d2^2 by a few
                // 2^-24 relative roundings, far inside the margin that NEAR leaves
                const float d0 = f0 - c0[j], d1 = f1 - c1[j], d2 = f2 - c2[j];
                const float q = __builtin_fmaf(d2, d2, __builtin_fmaf(d1, d1, d0 * d0));
                const uint32_t kq = (__float_as_uint(q) & ~IDXM) | (uint32_t)j;
                second = umed3(kq, best, second);
                best = min(best, kq);
            }
            int bi = (int)(best & IDXM);
            if (__builtin_expect(__fmul_rn(__uint_as_float(second), NEAR) < __uint_as_float(best), 0)) {
                float bs = __builtin_inff();
                bi = 0;
                for (int j = 0; j < k; ++j) {
                    const float e0 = __fsub_rn(f0, cen[j]), e1 = __fsub_rn(f1, cen[KM + j]),
                                e2 = __fsub_rn(f2, cen[2 * KM + j]);
                    const float sq = __fsqrt_rn(__fadd_rn(
                        __fadd_rn(__fmul_rn(e0, e0), __fmul_rn(e1, e1)), __fmul_rn(e2, e2)));
                    if (sq < bs) {
                        bs = sq;
                        bi = j;
                    }
                }
            }
            pix[i] = (v & 0xffffffu) | ((uint32_t)bi << 24);
            if (PACKED) {
                // byte shuffles: 1 | L << 16 and a | b << 16
                atomicAdd(&acc[(bi * 2) * 64 + lane], __builtin_amdgcn_perm(1u, v, 0x0c000c04u));
                atomicAdd(&acc[(bi * 2 + 1) * 64 + lane], __builtin_amdgcn_perm(0u, v, 0x0c020c01u));
            } else {
                atomicAdd(&acc[(bi * 4) * 64 + lane], weight);
                atomicAdd(&acc[(bi * 4 + 1) * 64 + lane], __umul24(weight, u0));
                atomicAdd(&acc[(bi * 4 + 2) * 64 + lane], __umul24(weight, u1));
                atomicAdd(&acc[(bi * 4 + 3) * 64 + lane], __umul24(weight, u2));
            }
        }
        __syncthreads();
        SEG_MARK(2)
        if (wave == 0 && PACKED) {
            // ---- the serial section of a round, kept in registers: every dependent LDS access
            // here costs a few hundred cycles under the other workgroup's traffic, and eleven
            // waves wait for this one.  Reduce the 64 lane-private copies (4 lanes per
            // (cluster, word) row, 16 copies each; the 4 partial sums of a row meet through two
            // quad permutes), hand cluster j's {count, sum0} / {sum1, sum2} rows to lane j with
            // one permute each, and add the shifts up from lane reads.
            uint32_t nj = 0, s0 = 0, s1 = 0, s2 = 0;
        constexpr int ITER = (KM * 2 + 15) / 16;
        uint32_t lo[ITER], hi[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int row = it * 16 + (lane >> 2), q = lane & 3;
            lo[it] = hi[it] = 0;
            if (row < k * 2) {
                uint4* src = reinterpret_cast<uint4*>(acc + row * 64 + q * 16);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint4 w4 = src[t];
                    src[t] = make_uint4(0, 0, 0, 0);
                    lo[it] += (w4.x & 0xffffu) + (w4.y & 0xffffu) + (w4.z & 0xffffu) + (w4.w & 0xffffu);
                    hi[it] += (w4.x >> 16) + (w4.y >> 16) + (w4.z >> 16) + (w4.w >> 16);
                }
            }
            lo[it] += (uint32_t)__builtin_amdgcn_mov_dpp((int)lo[it], 0xb1, 0xf, 0xf, true);   // lanes ^ 1
            hi[it] += (uint32_t)__builtin_amdgcn_mov_dpp((int)hi[it], 0xb1, 0xf, 0xf, true);
            lo[it] += (uint32_t)__builtin_amdgcn_mov_dpp((int)lo[it], 0x4e, 0xf, 0xf, true);   // lanes ^ 2
            hi[it] += (uint32_t)__builtin_amdgcn_mov_dpp((int)hi[it], 0x4e, 0xf, 0xf, true);
        }
        // cluster j: rows 2j {count, sum0} and 2j + 1 {sum1, sum2}, i.e. lanes 8 j' and 8 j' + 4
        // of iteration j / 8 (j' = j % 8)
        const int srcl = (lane & 7) * 8;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const uint32_t a_ = (uint32_t)__shfl((int)lo[it], srcl), b_ = (uint32_t)__shfl((int)hi[it], srcl);
            const uint32_t c_ = (uint32_t)__shfl((int)lo[it], srcl + 4), d_ = (uint32_t)__shfl((int)hi[it], srcl + 4);
            if ((lane >> 3) == it) {
                nj = a_;
                s0 = b_;
                s1 = c_;
                s2 = d_;
            }
        }
        
            SEG_MARK(3)
            // centre update + stop rule (:93-107); lane j < k holds centre j and its previous value
            const unsigned long long empty = __ballot(lane < k && nj == 0);
            if (empty) {
                resample_empty(empty, pix, pos, n, bw, xmin, ymin, W, seed, cen, KM);
            }
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            if (lane < k) {
                red[lane * 4] = nj;                       // the final statistics read these back
                red[lane * 4 + 1] = s0;
                red[lane * 4 + 2] = s1;
                red[lane * 4 + 3] = s2;
                if (nj) {
                    const float fn = (float)nj;
                    wc0 = __fdiv_rn((float)s0, fn);
                    wc1 = __fdiv_rn((float)s1, fn);
                    wc2 = __fdiv_rn((float)s2, fn);
                    cen[lane] = wc0;
                    cen[KM + lane] = wc1;
                    cen[2 * KM + lane] = wc2;
                } else {                                  // the re-sampled pixel (:101)
                    wc0 = cen[lane];
                    wc1 = cen[KM + lane];
                    wc2 = cen[2 * KM + lane];
                }
                a0 = fabsf(__fsub_rn(wc0, wo0));
                a1 = fabsf(__fsub_rn(wc1, wo1));
                a2 = fabsf(__fsub_rn(wc2, wo2));
            }
            // float32 sums in cluster order (:102), every lane the same
            float sh0 = 0.f, sh1 = 0.f, sh2 = 0.f;
#pragma unroll
            for (int j = 0; j < KM; ++j)
                if (j < k) {
                    sh0 = __fadd_rn(sh0, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a0), j)));
                    sh1 = __fadd_rn(sh1, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a1), j)));
                    sh2 = __fadd_rn(sh2, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a2), j)));
                }
            const double cent_sum = ((double)sh0 + (double)sh1) + (double)sh2;
            rounds += 1;
            const bool stop = fabs(old_sum - cent_sum) < 0.1 || rounds >= max_rounds;
            if (!stop) {
                old_sum = cent_sum;
                wo0 = wc0;
                wo1 = wc1;
                wo2 = wc2;
            } else if (lane == 0) {
                s_done = rounds;
            }
        } else if (wave == 0) {
            // reduce the 64 lane-private copies: 4 lanes per (cluster, word) row, 16 copies each
            for (int row0 = 0; row0 < k * WORDS; row0 += 16) {
                const int row = row0 + (lane >> 2), q = lane & 3;
                uint32_t lo = 0, hi = 0;
                if (row < k * WORDS) {
                    uint4* src = reinterpret_cast<uint4*>(acc + row * 64 + q * 16);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const uint4 w4 = src[t];
                        src[t] = make_uint4(0, 0, 0, 0);
                        if (PACKED) {
                            lo += (w4.x & 0xffffu) + (w4.y & 0xffffu) + (w4.z & 0xffffu) + (w4.w & 0xffffu);
                            hi += (w4.x >> 16) + (w4.y >> 16) + (w4.z >> 16) + (w4.w >> 16);
                        } else {
                            lo += w4.x + w4.y + w4.z + w4.w;
                        }
                    }
                }
                lo += __shfl_xor(lo, 1);
                hi += __shfl_xor(hi, 1);
                lo += __shfl_xor(lo, 2);
                hi += __shfl_xor(hi, 2);
                if (q == 0 && row < k * WORDS) {
                    if (PACKED) {           // row = cluster*2 + word: {count, s0}, {s1, s2}
                        red[row * 2] = lo;
                        red[row * 2 + 1] = hi;
                    } else {                // row = cluster*4 + {count, s0, s1, s2}
                        red[row] = lo;
                    }
                }
            }
            __threadfence_block();          // red is read back by other lanes of this wave
            SEG_MARK(3)
            // centre update + stop rule (:93-107)
            const unsigned long long empty = __ballot(lane < k && red[(lane < k ? lane : 0) * 4] == 0u);
            if (empty) {
                resample_empty(empty, pix, pos, n, bw, xmin, ymin, W, seed, cen, KM);
            }
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
            if (lane < k) {
                const uint32_t nj = red[lane * 4];
                c0 = cen[lane];
                c1 = cen[KM + lane];
                c2 = cen[2 * KM + lane];
                if (nj) {
                    const float fn = (float)nj;
                    c0 = __fdiv_rn((float)red[lane * 4 + 1], fn);
                    c1 = __fdiv_rn((float)red[lane * 4 + 2], fn);
                    c2 = __fdiv_rn((float)red[lane * 4 + 3], fn);
                    cen[lane] = c0;
                    cen[KM + lane] = c1;
                    cen[2 * KM + lane] = c2;
                }
                a0 = fabsf(__fsub_rn(c0, old[lane]));
                a1 = fabsf(__fsub_rn(c1, old[KM + lane]));
                a2 = fabsf(__fsub_rn(c2, old[2 * KM + lane]));
            }
            // float32 sums in cluster order (:102).  The shifts go through LDS and every lane
            // adds them up itself: all the reads are issued at once (one round trip) where a
            // chain of cross-lane shuffles costs one round trip per term.
            if (lane < KM) {
                sbuf[lane] = a0;
                sbuf[KM + lane] = a1;
                sbuf[2 * KM + lane] = a2;
            }
            __builtin_amdgcn_wave_barrier();            // same wave: LDS operations are in order
            float t0[KM], t1[KM], t2[KM];
#pragma unroll
            for (int j = 0; j < KM; ++j) {
                t0[j] = sbuf[j];
                t1[j] = sbuf[KM + j];
                t2[j] = sbuf[2 * KM + j];
            }
            float sh0 = 0.f, sh1 = 0.f, sh2 = 0.f;
#pragma unroll
            for (int j = 0; j < KM; ++j)
                if (j < k) {
                    sh0 = __fadd_rn(sh0, t0[j]);
                    sh1 = __fadd_rn(sh1, t1[j]);
                    sh2 = __fadd_rn(sh2, t2[j]);
                }
            const double cent_sum = ((double)sh0 + (double)sh1) + (double)sh2;
            rounds += 1;
            const bool stop = fabs(old_sum - cent_sum) < 0.1 || rounds >= max_rounds;
            if (!stop) {
                old_sum = cent_sum;
                if (lane < k) {
                    old[lane] = c0;
                    old[KM + lane] = c1;
                    old[2 * KM + lane] = c2;
                }
            } else if (lane == 0) {
                s_done = rounds;
            }
        }
        __syncthreads();
        SEG_MARK(4)
        if (s_done) break;
    }

    // ---- statistics (:110-116): count / sums come from the last round, squares need a pass
    for (int i = tid; i < k * 3 * 64; i += THREADS) acc[i] = 0;
    __syncthreads();
    const float inv_bw = 1.0f / (float)bw;
    for (int i = tid; i < n; i += THREADS) {
        const uint32_t v = pix[i];
        const uint32_t u0 = v & 255u, u1 = (v >> 8) & 255u, u2 = (v >> 16) & 255u, bi = v >> 24;
        atomicAdd(&acc[(bi * 3) * 64 + lane], u0 * u0);
        atomicAdd(&acc[(bi * 3 + 1) * 64 + lane], u1 * u1);
        atomicAdd(&acc[(bi * 3 + 2) * 64 + lane], u2 * u2);
        // assignment back to the image
        const int o = pos[i];
        int yy = (int)((float)o * inv_bw);
        int xx = o - __mul24(yy, bw);
        if (xx < 0) {
            --yy;
            xx += bw;
        } else if (xx >= bw) {
            ++yy;
            xx -= bw;
        }
        if (assign) assign[img + (size_t)(uint32_t)(__mul24(ymin + yy, W) + xmin + xx)] = (uint8_t)bi;
    }
    __syncthreads();
    SEG_MARK(5)
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

    // ---- spatial coherence (:40-74, :118-127): largest 8-connected component of every
    // cluster, found on chip.  Each thread takes its pixels (cell in the box, cluster) into
    // registers; the pixel store is then reused as a grid over the box: first the cluster of every
    // cell (u8), from which each cell picks its link (kept in its register as a direction), then,
    // in the same place, the parents (u16).
    constexpr int NPT = CAP / THREADS;
    static_assert(NPT * THREADS == CAP, "CAP must be a multiple of the block size");
    uint32_t keep[NPT];
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
        const int i = tid + t * THREADS;
        keep[t] = i < n ? ((uint32_t)pos[i] | ((pix[i] >> 24) << 16)) : 0xffffffffu;
    
    }
    uint16_t* par = reinterpret_cast<uint16_t*>(smem);
    uint8_t* cls = smem;
    uint32_t* largest = reinterpret_cast<uint32_t*>(slots);
    __syncthreads();
    SEG_MARK(9)
    for (int i = tid; i < (area + 3) / 4; i += THREADS) reinterpret_cast<uint32_t*>(cls)[i] = 0xffffffffu;
    if (tid < KM) largest[tid] = 0;
    // the cells that need an explicit union below, collected in the accumulators' place
    uint16_t* joins = reinterpret_cast<uint16_t*>(acc);
    constexpr int JOINS = KM * ACCW * 64 * 2;
    int& s_joins = flags[3];
    if (tid == 0) s_joins = 0;
    __syncthreads();
    SEG_MARK(10)
#pragma unroll
    for (int t = 0; t < NPT; ++t)
        if (keep[t] != 0xffffffffu) cls[keep[t] & 0xffffu] = (uint8_t)(keep[t] >> 16);
    __syncthreads();
    SEG_MARK(11)
    // Every cell hangs itself under one earlier neighbour of its cluster (N, else W, else NW,
    // else NE; all have a smaller index, so the links form a forest) with a plain store.
    // Of the four backward links W / NW / N / NE only one more can be needed: with N present
    // the others follow from the neighbours' own links (NW-N, N-NE, W-N), and W covers NW;
    // what is left is NE next to a missing N, which is united explicitly below.  Those cells go
    // on a list (one LDS atomic per such cell; reserving a wave's entries with one atomic behind a
    // ballot gave wrong component sizes in the dedup instantiations only, with __shfl and with
    // v_readlane alike -- cause not found, left alone) that the whole block then works through evenly:
    // taken where they stand, a wave would run one union after the other for as many steps as any
    // of its lanes has such a cell, nearly all fourteen.  (Bit 25: no room on the list.)
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
        bool join = false;
        if (keep[t] != 0xffffffffu) {
            const uint32_t cell = keep[t] & 0xffffu;
            const uint8_t c = (uint8_t)(keep[t] >> 16);
            int yy = (int)((float)cell * inv_bw);
            int xx = (int)cell - __mul24(yy, bw);
            yy += (xx >= bw) - (xx < 0);
            xx = (int)cell - __mul24(yy, bw);
            const bool up = yy > 0, lf = xx > 0, rt = xx < bw - 1;
            const bool n_ = up && cls[cell - bw] == c, w_ = lf && cls[cell - 1] == c;
            const bool nw = up && lf && cls[cell - bw - 1] == c;
            const bool ne = up && rt && cls[cell - bw + 1] == c;
            // bits 26-28: 0 self, 1 N, 2 W, 3 NW, 4 NE
            keep[t] |= (n_ ? 1u : w_ ? 2u : nw ? 3u : ne ? 4u : 0u) << 26;
            join = !n_ && (w_ || nw) && ne;
        }
        if (join) {
            const int at = atomicAdd(&s_joins, 1);
            if (at < JOINS)
                joins[at] = (uint16_t)(keep[t] & 0xffffu);
            else
                keep[t] |= 0x2000000u;
        }
    }
    __syncthreads();                                    // every cluster byte has been read
#pragma unroll
    for (int t = 0; t < NPT; ++t)
        if (keep[t] != 0xffffffffu) {
            const uint32_t cell = keep[t] & 0xffffu, dir = (keep[t] >> 26) & 7u;
            const uint32_t link = dir == 1u ? cell - bw : dir == 2u ? cell - 1 : dir == 3u ? cell - bw - 1
                                                        : dir == 4u ? cell - bw + 1 : cell;
            par[cell] = (uint16_t)link;
        }
    __syncthreads();
    SEG_MARK(12)
    for (int e = tid, ne_ = min(s_joins, JOINS); e < ne_; e += THREADS) {
        const uint32_t cell = joins[e];
        unite16(par, cell, cell - bw + 1);
    }
    if (s_joins > JOINS)
#pragma unroll
    for (int t = 0; t < NPT; ++t)
        if (keep[t] != 0xffffffffu && (keep[t] & 0x2000000u)) {
            const uint32_t cell = keep[t] & 0xffffu;
            unite16(par, cell, cell - bw + 1);
        }
    __syncthreads();
    SEG_MARK(13)
    // roots into registers (bit 24: this cell is a root), then the parent grid turns into
    // 16-bit size counters (two per word; a segment has < 65536 pixels, so no carry)
#pragma unroll
    for (int t = 0; t < NPT; ++t)
        if (keep[t] != 0xffffffffu) {
            const uint32_t cell = keep[t] & 0xffffu;
            const uint32_t root = find16(par, cell);
            par[cell] = (uint16_t)root;                 // shortens the walks of later cells
            keep[t] = root | (keep[t] & 0xff0000u) | (root == cell ? 0x1000000u : 0u);   // drops bits 25-28
        }
    __syncthreads();
    SEG_MARK(14)
    for (int i = tid; i < (area + 1) / 2; i += THREADS) reinterpret_cast<uint32_t*>(par)[i] = 0;
    __syncthreads();
    SEG_MARK(6)
#pragma unroll
    for (int t = 0; t < NPT; ++t)
        if (keep[t] != 0xffffffffu) {
            const uint32_t root = keep[t] & 0xffffu;
            atomicAdd(reinterpret_cast<uint32_t*>(par) + (root >> 1), 1u << ((root & 1u) * 16u));
        }
    __syncthreads();
    SEG_MARK(15)
#pragma unroll
    for (int t = 0; t < NPT; ++t)
        if (keep[t] != 0xffffffffu && (keep[t] & 0x1000000u))
            atomicMax(&largest[(keep[t] >> 16) & 0xffu], (uint32_t)par[keep[t] & 0xffffu]);
    __syncthreads();
    SEG_MARK(8)
    if (tid < k) {
        out[tid * 8] = red[tid * 4];
        out[tid * 8 + 1] = red[tid * 4 + 1];
        out[tid * 8 + 2] = red[tid * 4 + 2];
        out[tid * 8 + 3] = red[tid * 4 + 3];
        out[tid * 8 + 7] = largest[tid];
        float* dst = centres + (gs * k + tid) * 3;
        dst[0] = cen[tid];
        dst[1] = cen[KM + tid];
        dst[2] = cen[2 * KM + tid];
    }
    if (tid == 0) {
        st.rounds = s_done;
        seg[gs] = st;
    }
}

// LDS of one workgroup: the layout at the top of segment_body; the last 16 bytes are its flags
// (s_n, s_done) and the ticket of the listed mode
__host__ __device__ constexpr size_t seg_lds_bytes(int km, int cap) {
    return (size_t)cap * 6 + (size_t)km * ((cap / 64) * 255 < 65536 ? 3 : 4) * 64 * 4 +
           (size_t)km * 8 + (size_t)km * 3 * 4 * 2 + (size_t)km * 4 * 4 + (size_t)km * 3 * 4 + 16;
}

// waves per SIMD when as many workgroups as the LDS allows share a CU: the register budget the
// compiler is held to (the listed mode's body otherwise settles above it, and a workgroup fewer fits)
constexpr int seg_waves_per_simd(int km, int cap, int threads) {
    return (threads / 256) * (int)((160 * 1024) / seg_lds_bytes(km, cap));
}

template <int KM, int CAP, int THREADS, int MODE>
__global__ __launch_bounds__(THREADS, seg_waves_per_simd(KM, CAP, THREADS))
void dcd_segment_kernel(
    const uint8_t* __restrict__ lab, const int32_t* __restrict__ labels, int B, int H, int W,
    int nseg, int k, int use_hash, uint32_t seed, float* __restrict__ centres,
    uint8_t* __restrict__ assign, SegState* __restrict__ seg, unsigned long long* __restrict__ stats,
    const int* __restrict__ bbox, int* __restrict__ retry, int* __restrict__ big_count,
    int max_rounds, SegLists L, int tier) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (MODE == MODE_FIRST) {
        segment_body<KM, CAP, THREADS, MODE_FIRST>(smem, blockIdx.y, blockIdx.x, B, lab, labels, H, W,
                                                   nseg, k, use_hash, seed, centres, assign, seg,
                                                   stats, bbox, retry, big_count, max_rounds, L, 0);
    } else if (MODE == MODE_LISTED) {
        // the workgroups draw tickets for the entries of their list: segments differ in size and
        // in rounds by an order of magnitude, so a fixed split would leave a long tail
        int* ticket = reinterpret_cast<int*>(smem + seg_lds_bytes(KM, CAP) - 8);
        const int count = L.hdr[tier];
        while (true) {
            if (threadIdx.x == 0) *ticket = atomicAdd(&L.hdr[8 + tier], 1);
            __syncthreads();
            const int e = *ticket;
            __syncthreads();
            if (e >= count) break;
            const int gs = L.tier[tier][e];
            segment_body<KM, CAP, THREADS, MODE_LISTED>(smem, gs / nseg, gs % nseg, B, lab, labels, H, W, nseg,
                                                        k, use_hash, seed, centres, assign, seg, stats, bbox,
                                                        retry, big_count, max_rounds, L, tier);
            __syncthreads();
        }
    } else {
        // a fixed, small grid walks the retry list (launching one workgroup per segment just to
        // find most of them done cost more than the segments that needed the retry)
        const int count = retry[0];
        for (int e = blockIdx.x; e < count; e += gridDim.x) {
            const int gs = retry[1 + e];
            segment_body<KM, CAP, THREADS, MODE_RETRY>(smem, gs / nseg, gs % nseg, B, lab, labels, H,
                                                       W, nseg, k, use_hash, seed, centres, assign,
                                                       seg, stats, bbox, retry, big_count,
                                                       max_rounds, L, 0);
            __syncthreads();
        }
    }
}

template <int KM, int CAP, int THREADS, int MODE>
int launch_tier(dim3 grid, hipStream_t st, const uint8_t* lab, const int32_t* labels, int B, int H,
                int W, int nseg, int k, int use_hash, uint32_t seed, float* centres, uint8_t* assign,
                SegState* seg, unsigned long long* stats, const int* bbox, int* retry, int* n_big,
                int max_rounds, const SegLists& L, int tier) {
    size_t lds = seg_lds_bytes(KM, CAP);
    auto fn = dcd_segment_kernel<KM, CAP, THREADS, MODE>;
    IMGAN_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
    hipLaunchKernelGGL(fn, grid, dim3(THREADS), lds, st, lab, labels, B, H, W, nseg, k, use_hash, seed,
                       centres, assign, seg, stats, bbox, retry, n_big, max_rounds, L, tier);
    return 0;
}

}  // namespace

#ifdef IMGAN_SEG_TIMING
extern "C" int imgan_debug_seg_timing(unsigned long long* out16, int reset) {
    IMGAN_HIP(hipDeviceSynchronize());
    IMGAN_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(seg_timing), 128));
    if (reset) {
        unsigned long long z[16] = {0};
        IMGAN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(seg_timing), z, 128));
    }
    return 0;
}
#endif

extern "C" size_t imgan_dcd_bbox_bytes(int B, int nseg) {
    // boxes [B*nseg][4] + the retry list of the large tier {count, ids[B*nseg]}
    return (B < 1 || nseg < 1) ? 0 : imgan::align_up((size_t)B * nseg * 4 * 4 + 16, 256) +
                                         imgan::align_up(((size_t)B * nseg + 1) * 4, 256);
}

static int dcd_segments(const uint8_t* lab, const int32_t* labels, int B, int H, int W, int nseg,
                        int k, int use_hash_init, uint32_t seed, float* centres, uint8_t* assign,
                        void* state, uint64_t* stats, void* bbox_scratch, bool have_bbox,
                        int32_t* n_big_out, int max_rounds, imgan_stream_t stream) {
    IMGAN_REQUIRE(lab && labels && centres && assign && state && stats && bbox_scratch && n_big_out,
                  "imgan_dcd_segments: null pointer");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && nseg >= 1 && B <= 65535 && H <= 65535,
                  "imgan_dcd_segments: bad shape");
    IMGAN_REQUIRE(k >= 1 && k <= KMAX, "imgan_dcd_segments: 1 <= k <= 16 dominant colours");
    IMGAN_REQUIRE((long long)H * W < 0x7fffffffLL && W < (1 << 23),
                  "imgan_dcd_segments: image too large");
    IMGAN_REQUIRE((long long)B * H * W >= 2, "imgan_dcd_segments: needs at least two pixels");
    IMGAN_REQUIRE(max_rounds >= 1, "imgan_dcd_segments: max_rounds must be positive");
    hipStream_t st = (hipStream_t)stream;
    DcdLayout L = dcd_layout(B, nseg, k);
    SegState* seg = (SegState*)((char*)state + L.off_seg);
    int* bbox = (int*)bbox_scratch;
    size_t nsegs = (size_t)B * nseg;
    IMGAN_HIP(hipMemsetAsync(state, 0, L.total, st));
    IMGAN_HIP(hipMemsetAsync(n_big_out, 0, 4, st));
    IMGAN_HIP(hipMemsetAsync(assign, 0xff, (size_t)B * H * W, st));
    if (!have_bbox) {
        hipLaunchKernelGGL(seg_bbox_init_kernel, dim3((unsigned)imgan::cdiv((long long)nsegs, 256)),
                           dim3(256), 0, st, bbox, (int)nsegs);
        dim3 gb((unsigned)imgan::cdiv(W, 256), (unsigned)imgan::cdiv(H, BBOX_ROWS), (unsigned)B);
        hipLaunchKernelGGL(seg_bbox_kernel, gb, dim3(256), 0, st, labels, H, W, nseg, bbox);
    }
    int* retry = (int*)((char*)bbox_scratch + imgan::align_up(nsegs * 4 * 4 + 16, 256));
    IMGAN_HIP(hipMemsetAsync(retry, 0, 4, st));
    dim3 gs((unsigned)nseg, (unsigned)B);
    dim3 gr((unsigned)std::min<size_t>(nsegs, 2048));
    unsigned long long* stt = (unsigned long long*)stats;
    SegLists none = {};
#define TIER(KM, CAP, THR, MODE)                                                                     \
    if (int rc = launch_tier<KM, CAP, THR, MODE>(MODE == MODE_FIRST ? gs : gr, st, lab, labels, B, H, \
                                                 W, nseg, k, use_hash_init, seed, centres, assign,   \
                                                 seg, stt, bbox, retry, n_big_out, max_rounds, none, \
                                                 0))                                                 \
        return rc;
    if (k <= 8) {
        TIER(8, CAP_SMALL, THREADS_SMALL, MODE_FIRST)
        TIER(8, CAP_LARGE, THREADS_LARGE, MODE_RETRY)
    } else {
        TIER(16, CAP_SMALL_K16, THREADS_SMALL, MODE_FIRST)
        TIER(16, CAP_LARGE, THREADS_LARGE, MODE_RETRY)
    }
#undef TIER
    return imgan::check_launch("imgan_dcd_segments");
}

extern "C" size_t imgan_dcd_list_bytes(int B, int H, int W, int nseg) {
    if (B < 1 || H < 1 || W < 1 || nseg < 1) return 0;
    return seg_lists_bytes((size_t)B * nseg, (size_t)B * H * W);
}

extern "C" int imgan_label_counts(const int32_t* labels, int B, int H, int W, int nseg,
                                  int32_t* counts, imgan_stream_t stream) {
    IMGAN_REQUIRE(labels && counts, "imgan_label_counts: null pointer");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && nseg >= 1 && B <= 65535 && H <= 65535 * SC_BLOCK_ROWS,
                  "imgan_label_counts: bad shape");
    IMGAN_REQUIRE((long long)H * W < 0x7fffffffLL, "imgan_label_counts: image too large");
    hipStream_t st = (hipStream_t)stream;
    IMGAN_HIP(hipMemsetAsync(counts, 0, (size_t)B * nseg * 4, st));
    dim3 grid((unsigned)imgan::cdiv(W, SC_BLOCK_W), (unsigned)imgan::cdiv(H, SC_BLOCK_ROWS), (unsigned)B);
    SegLists none = {};
    hipLaunchKernelGGL(seg_scatter_kernel<true>, grid, dim3(SC_THREADS), 0, st,
                       labels, B, H, W, nseg, none, counts);
    return imgan::check_launch("seg_scatter_kernel<count>");
}

extern "C" int imgan_dcd_segments_listed(const uint8_t* lab, const int32_t* labels, int B, int H,
                                         int W, int nseg, int k, int use_hash_init, uint32_t seed,
                                         float* centres, uint8_t* assign, void* state,
                                         uint64_t* stats, const int32_t* bbox, const int32_t* counts,
                                         void* list_scratch, int32_t* n_big_out, int max_rounds,
                                         imgan_stream_t stream) {
    IMGAN_REQUIRE(lab && labels && centres && state && stats && bbox && counts && list_scratch &&
                      n_big_out,
                  "imgan_dcd_segments_listed: null pointer");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && nseg >= 1 && B <= 65535 && H <= 65535 && W <= 65536,
                  "imgan_dcd_segments_listed: bad shape");
    IMGAN_REQUIRE(k >= 1 && k <= KMAX, "imgan_dcd_segments_listed: 1 <= k <= 16 dominant colours");
    IMGAN_REQUIRE((long long)H * W < 0x7fffffffLL && (long long)B * H * W < 0xffffffffLL &&
                      (long long)B * nseg < 0x7fffffffLL,
                  "imgan_dcd_segments_listed: batch too large");
    IMGAN_REQUIRE((long long)B * H * W >= 2, "imgan_dcd_segments_listed: needs at least two pixels");
    IMGAN_REQUIRE(max_rounds >= 1, "imgan_dcd_segments_listed: max_rounds must be positive");
    hipStream_t st = (hipStream_t)stream;
    DcdLayout D = dcd_layout(B, nseg, k);
    SegState* seg = (SegState*)((char*)state + D.off_seg);
    const size_t nsegs = (size_t)B * nseg, npx = (size_t)B * H * W;
    SegLists L = bind_lists(list_scratch, nsegs, npx);
    IMGAN_HIP(hipMemsetAsync(state, 0, D.total, st));
    IMGAN_HIP(hipMemsetAsync(n_big_out, 0, 4, st));
    if (assign) IMGAN_HIP(hipMemsetAsync(assign, 0xff, npx, st));
    // IMGAN_DCD_DEDUP=0: the plain listed kernels on every tier (for comparisons)
    static const bool dedup = [] {
        const char* e = getenv("IMGAN_DCD_DEDUP");
        return !(e && e[0] == '0');
    }();
    const int cap0 = k <= 8 ? CAP_TINY_D : 0;
    const int cap1 = k <= 8 ? CAP_SMALL_D : CAP_SMALL_K16;
    IMGAN_HIP(hipMemsetAsync(L.hdr, 0, 256, st));
    hipLaunchKernelGGL(seg_plan_kernel, dim3((unsigned)B), dim3(1024), 0, st, counts, bbox, nseg, H, W, L,
                       cap0, cap1, CAP_LARGE, n_big_out);
    dim3 gsc((unsigned)imgan::cdiv(W, SC_BLOCK_W), (unsigned)imgan::cdiv(H, SC_BLOCK_ROWS), (unsigned)B);
    hipLaunchKernelGGL(seg_scatter_kernel<false>, gsc, dim3(SC_THREADS), 0, st, labels, B, H, W, nseg, L,
                       (int32_t*)nullptr);
    unsigned long long* stt = (unsigned long long*)stats;
    // IMGAN_DCD_STOP_AFTER_SORT: leave after the sort pass, so that tools/check_lists.py can hold the
    // lists against the label map
    static const bool stop_after_sort = getenv("IMGAN_DCD_STOP_AFTER_SORT") != nullptr;
    if (stop_after_sort) return imgan::check_launch("seg_scatter_kernel");
    if (dedup) {
        // the dedup tiers (dcd_tiers.hip), smallest first; what overflows every colour table ends
        // up on list 3 and takes the plain body here, one workgroup per CU
        if (int rc = imgan_dcd_launch_dedup_tiers(lab, B, H, W, nseg, k, use_hash_init, seed, centres,
                                                  assign, seg, stt, bbox, max_rounds, L, st))
            return rc;
        dim3 grid((unsigned)std::min<size_t>(nsegs, 256));
        if (k <= 8) {
            if (int rc = launch_tier<8, CAP_LARGE, THREADS_LARGE, MODE_LISTED>(
                    grid, st, lab, labels, B, H, W, nseg, k, use_hash_init, seed, centres, assign, seg, stt,
                    bbox, nullptr, n_big_out, max_rounds, L, 3))
                return rc;
        } else {
            if (int rc = launch_tier<16, CAP_LARGE, THREADS_LARGE, MODE_LISTED>(
                    grid, st, lab, labels, B, H, W, nseg, k, use_hash_init, seed, centres, assign, seg, stt,
                    bbox, nullptr, n_big_out, max_rounds, L, 3))
                return rc;
        }
        return imgan::check_launch("imgan_dcd_segments_listed");
    }
    // IMGAN_DCD_DEDUP=0: every segment through the plain body, the tiers as the plan made them.
    // The large tier first (its few long jobs should not be what everything else waits for at the
    // end); a grid of a few workgroups per CU slot, which draw segments until the tier's list is
    // empty.  (Measured: the tiers on three streams side by side gain 2 % at 64 images and lose
    // 1 % at 256; they stay on the caller's stream.)
#define TIER(KM, CAP, THR, MODE, PER_CU, T)                                                           \
    if (int rc = launch_tier<KM, CAP, THR, MODE>(                                                     \
            dim3((unsigned)std::min<size_t>(nsegs, 256 * PER_CU)), st, lab, labels, B, H, W, nseg, k, \
            use_hash_init, seed, centres, assign, seg, stt, bbox, nullptr, n_big_out, max_rounds, L,  \
            T))                                                                                       \
        return rc;
    if (k <= 8) {
        TIER(8, CAP_LARGE, THREADS_LARGE, MODE_LISTED, 1, 2)
        TIER(8, CAP_SMALL_D, THREADS_SMALL, MODE_LISTED, 2, 1)
        TIER(8, CAP_TINY_D, THREADS_TINY, MODE_LISTED, 3, 0)
    } else {
        TIER(16, CAP_LARGE, THREADS_LARGE, MODE_LISTED, 1, 2)
        TIER(16, CAP_SMALL_K16, THREADS_SMALL, MODE_LISTED, 2, 1)
    }
#undef TIER
    return imgan::check_launch("imgan_dcd_segments_listed");
}

extern "C" int imgan_dcd_segments(const uint8_t* lab, const int32_t* labels, int B, int H, int W,
                                  int nseg, int k, int use_hash_init, uint32_t seed,
                                  float* centres, uint8_t* assign, void* state, uint64_t* stats,
                                  void* bbox_scratch, int32_t* n_big_out, int max_rounds,
                                  imgan_stream_t stream) {
    return dcd_segments(lab, labels, B, H, W, nseg, k, use_hash_init, seed, centres, assign, state,
                        stats, bbox_scratch, false, n_big_out, max_rounds, stream);
}

extern "C" int imgan_dcd_segments_boxed(const uint8_t* lab, const int32_t* labels, int B, int H,
                                        int W, int nseg, int k, int use_hash_init, uint32_t seed,
                                        float* centres, uint8_t* assign, void* state,
                                        uint64_t* stats, void* bbox_scratch, int32_t* n_big_out,
                                        int max_rounds, imgan_stream_t stream) {
    return dcd_segments(lab, labels, B, H, W, nseg, k, use_hash_init, seed, centres, assign, state,
                        stats, bbox_scratch, true, n_big_out, max_rounds, stream);
}

extern "C" int imgan_label_bbox(const int32_t* labels, int B, int H, int W, int nseg,
                                int32_t* bbox, imgan_stream_t stream) {
    IMGAN_REQUIRE(labels && bbox, "imgan_label_bbox: null pointer");
    IMGAN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && nseg >= 1 && B <= 65535, "imgan_label_bbox: bad shape");
    hipStream_t st = (hipStream_t)stream;
    size_t nsegs = (size_t)B * nseg;
    IMGAN_REQUIRE(nsegs < 0x7fffffffULL, "imgan_label_bbox: too many segments");
    hipLaunchKernelGGL(seg_bbox_init_kernel, dim3((unsigned)imgan::cdiv((long long)nsegs, 256)),
                       dim3(256), 0, st, bbox, (int)nsegs);
    dim3 gb((unsigned)imgan::cdiv(W, 256), (unsigned)imgan::cdiv(H, BBOX_ROWS), (unsigned)B);
    IMGAN_REQUIRE(gb.y <= 65535, "imgan_label_bbox: image too tall");
    hipLaunchKernelGGL(seg_bbox_kernel, gb, dim3(256), 0, st, labels, H, W, nseg, bbox);
    return imgan::check_launch("seg_bbox_kernel");
}
