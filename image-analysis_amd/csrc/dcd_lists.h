// Shared by the LDS-resident dominant-colour kernels (dcd_seg.hip: box gather and plain pixel
// lists; dcd_tiers.hip: the dedup tiers): the pixel lists, the capacity tiers and the 16-bit
// union-find of the coherence pass.
#pragma once
#include "common.h"
#include "dcd_common.h"

namespace imgan_dcd {

// median of three unsigned words in one instruction (the compiler has no builtin for it)
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;   // one global_load_dword at any address

// Lab of the pixel at (x, y) of image b as L | a << 8 | b << 16 | junk << 24: one unaligned dword
// load; the very last pixel of the batch is read one byte early so that the load stays in bounds.
// x and y are clamped into the image (positions come from a list in scratch memory).
__device__ __forceinline__ uint32_t lab_at(const uint8_t* __restrict__ lab, size_t img_px, int x, int y,
                                           int H, int W, size_t last_px) {
    const size_t gp = img_px + (size_t)(uint32_t)(__mul24(min(max(y, 0), H - 1), W) + min(max(x, 0), W - 1));
    const uint32_t sh = gp >= last_px ? 1u : 0u;
    return *reinterpret_cast<const u32_unaligned*>(lab + gp * 3 - sh) >> (8u * sh);
}

// The same with 32-bit arithmetic, for images of less than 2^32 / 3 pixels: lab_img points at the
// image's first byte (the same in every lane: a scalar base), last_off is the byte offset of the
// batch's very last pixel when this is the last image, else 0xffffffff.  (The 64-bit form costs
// eight 64-bit vector operations per pixel.)
__device__ __forceinline__ uint32_t lab_at32(const uint8_t* __restrict__ lab_img, int x, int y, int H, int W,
                                             uint32_t last_off) {
    const uint32_t p = (uint32_t)(__mul24(min(max(y, 0), H - 1), W) + min(max(x, 0), W - 1));
    const uint32_t off = p * 3u;
    const uint32_t sh = off == last_off ? 1u : 0u;
    return *reinterpret_cast<const u32_unaligned*>(lab_img + (off - sh)) >> (8u * sh);
}

// Capacity tiers (pixels a workgroup holds in LDS at 6 B each, and its thread count).
// 768 threads = 3 waves on each SIMD of the CU per workgroup, 6 with the second workgroup: measured
// against 512 / 640 / 896 / 1024 threads (5.6 / 6.3 / 8.0 / 7.4 ms where this takes 5.1): the
// Lloyd rounds are latency bound per workgroup, and wave counts that do not divide by four leave
// one SIMD holding the others up at every barrier.
constexpr int CAP_SMALL = 12288, THREADS_SMALL = 768;
// listed mode only: three workgroups per CU for the segments that leave room for them; the two
// smaller tiers sized so that the accumulators of the dedup kernels fit beside the pixels
constexpr int CAP_TINY_D = 7168, THREADS_TINY = 512;       // 512 x 14
constexpr int CAP_SMALL_D = 11520;                         // 768 x 15
// (A fourth tier below these, 5,376 px / 384 threads = four workgroups per CU, was measured: 4.46
// against 4.16 ms at K = 400 and 5.6 against 4.9 at K = 800 -- six waves do not spread over four
// SIMDs, the same finding as for 640 and 896 threads.)
constexpr int CAP_SMALL_K16 = 10752;   // 9 .. 16 colours: twice the accumulators, still two workgroups per CU
constexpr int CAP_LARGE = 22528, THREADS_LARGE = 1024;     // >= 4 S^2 of the 1080p / K=400 grid

constexpr uint32_t NO_COLOUR = 0xffffffffu;            // an empty slot of the colour table
constexpr int DEDUP_PROBES = 64;                       // linear probes before a segment is given up (16 / 512 measured: +6 % / +11 %)

// ---- pixel lists (imgan_dcd_segments_listed).  With the pixel count of every segment known up
// front (the last SLIC round has it), one streaming pass sorts the pixels by segment: every
// segment owns a contiguous run of positions y << 16 | x (4 B per pixel; the pass reads the labels
// only).  The segment kernel then walks its run with straight reads and fetches the colours by
// position -- a chunk of a run is a piece of an image row, so neighbouring lanes read neighbouring
// bytes -- instead of scanning its bounding box (about 2.5 cells per pixel for a superpixel, with
// a ballot and an LDS atomic per 64 cells), and every segment goes directly to the capacity tier
// that fits it.  (Lists of {Lab, position}, 8 B per pixel, had the sorting pass read the colours
// too and write twice as much.  Its time is the same either way, 2.0 ms for 256 x 1080p -- it
// waits for its reserving atomics and for its stores to be acknowledged, not for bandwidth:
// measured with the stores, the atomics and the ranking removed in turn -- but the lists now take
// 2.1 GB instead of 4.2 for that batch.)
struct SegLists {
    int* hdr;               // [64]: 0..3 length of the lists below, 8..11 their ticket counters
    uint32_t* offsets;      // [nsegs] start of every segment's run (inside its image's H*W slab)
    uint32_t* ends;         // [nsegs] end of the run: start + count, clamped to the slab
    uint32_t* cursor;       // [nsegs] entries written so far
    int* tier[4];           // [nsegs] each: segment ids by tier; [3]: what the dedup tiers hand to the plain body
    uint32_t* sxy;          // [B*H*W] y << 16 | x
};

inline size_t seg_lists_bytes(size_t nsegs, size_t npx) {
    return 256 + 7 * imgan::align_up(nsegs * 4, 256) + imgan::align_up(npx * 4, 256);
}

inline SegLists bind_lists(void* scratch, size_t nsegs, size_t npx) {
    char* p = (char*)scratch;
    SegLists L;
    L.hdr = (int*)p;
    p += 256;
    L.offsets = (uint32_t*)p;
    p += imgan::align_up(nsegs * 4, 256);
    L.ends = (uint32_t*)p;
    p += imgan::align_up(nsegs * 4, 256);
    L.cursor = (uint32_t*)p;
    p += imgan::align_up(nsegs * 4, 256);
    for (int t = 0; t < 4; ++t) {
        L.tier[t] = (int*)p;
        p += imgan::align_up(nsegs * 4, 256);
    }
    L.sxy = (uint32_t*)p;
    return L;
}

// ---- union-find on 16-bit parents in LDS (the hardware has no 16-bit LDS atomics: a
// compare-and-swap on the containing word stands in for atomicMin)
// find with path halving.  The halving store may race with a concurrent atomic_min16 on the
// same entry; either value is an ancestor of the cell, so no link is lost (the uniting thread
// carries on with the value it displaced).
__device__ __forceinline__ uint32_t find16(volatile uint16_t* par, uint32_t a) {
    while (true) {
        const uint32_t p = par[a];
        if (p == a) return a;
        const uint32_t g = par[p];
        if (g == p) return p;
        par[a] = (uint16_t)g;
        a = g;
    }
}

__device__ __forceinline__ uint32_t atomic_min16(uint16_t* par, uint32_t idx, uint32_t val) {
    uint32_t* w = reinterpret_cast<uint32_t*>(par) + (idx >> 1);
    const uint32_t sh = (idx & 1u) * 16u;
    uint32_t old = *reinterpret_cast<volatile uint32_t*>(w);
    while (true) {
        const uint32_t cur = (old >> sh) & 0xffffu;
        if (cur <= val) return cur;
        const uint32_t nw = (old & ~(0xffffu << sh)) | (val << sh);
        const uint32_t prev = atomicCAS(w, old, nw);
        if (prev == old) return cur;
        old = prev;
    }
}

__device__ __forceinline__ void unite16(uint16_t* par, uint32_t a, uint32_t b) {
    while (true) {
        a = find16(par, a);
        b = find16(par, b);
        if (a == b) return;
        if (a < b) {
            uint32_t t = a;
            a = b;
            b = t;
        }
        const uint32_t old = atomic_min16(par, a, b);      // hang the larger root under the smaller
        if (old == a) return;
        a = old;                                           // somebody re-parented a meanwhile
    }
}

}  // namespace imgan_dcd

// dcd_tiers.hip: the dedup tier launches of imgan_dcd_segments_listed
int imgan_dcd_launch_dedup_tiers(const uint8_t* lab, int B, int H, int W, int nseg, int k, int use_hash,
                                 uint32_t seed, float* centres, uint8_t* assign,
                                 imgan_dcd::SegState* seg, unsigned long long* stats, const int* bbox,
                                 int max_rounds, const imgan_dcd::SegLists& L, hipStream_t st);
