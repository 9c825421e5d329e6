// Shared declarations of the dominant-colour kernels (dcd.hip: streaming path for segments of
// any size; dcd_seg.hip: LDS-resident path for compact segments).
#pragma once
#include "common.h"

namespace imgan_dcd {

constexpr int KMAX = 16;          // dominant colours per segment supported by the kernels
constexpr int HCAP = 512;         // LDS hash capacity (entries)

struct SegState {
    double old_sum;
    int done;
    int rounds;
    int handled;      // 1: finished by the LDS-resident segment kernel (streaming kernels skip it)
    int pending;      // streaming rounds: mask of the clusters waiting for their re-sample draw
};

struct DcdLayout {
    size_t off_seg, off_oldcen, off_acc, off_draw, off_wanted, total;
};

// draw[segment][cluster] (streaming rounds): 0 = never needed, DRAW_WANTED = being looked for,
// anything else = ~resample_key of the pixel the cluster takes when it is empty (inverted so that
// a zero-filled state means "never needed" and atomicMax finds the smallest key)
constexpr unsigned long long DRAW_WANTED = 1ull;

inline DcdLayout dcd_layout(int B, int nseg, int k) {
    DcdLayout L;
    size_t nsegs = (size_t)B * nseg;
    size_t o = 0;
    L.off_seg = o;
    o = imgan::align_up(o + nsegs * sizeof(SegState), 256);
    L.off_oldcen = o;
    o = imgan::align_up(o + nsegs * k * 3 * sizeof(float), 256);
    L.off_acc = o;
    o = imgan::align_up(o + nsegs * k * 4 * sizeof(unsigned long long), 256);
    L.off_draw = o;
    o = imgan::align_up(o + nsegs * k * sizeof(unsigned long long), 256);
    L.off_wanted = o;
    o = imgan::align_up(o + sizeof(int), 256);
    L.total = o;
    return L;
}

__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

// Stand-in for the unseeded random.sample(segment, 1) that re-seeds an empty cluster
// (mpeg7Descriptors.py:101): cluster j takes the pixel of its segment with the smallest
// (resample_hash, p), p = y * W + x.  The draw depends on the segment, the seed and j only, not on
// the round, so a cluster that stays empty keeps proposing the same pixel and the loop settles.
__device__ __forceinline__ unsigned long long resample_key(uint32_t p, uint32_t seed, int j) {
    const uint32_t h = mix32((p * 0x9e3779b1u + seed) ^ ((uint32_t)(j + 1) * 0x85ebca6bu));
    return ((unsigned long long)h << 32) | (unsigned long long)p;
}

}  // namespace imgan_dcd
