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
    int pad;
};

struct DcdLayout {
    size_t off_seg, off_oldcen, off_acc, total;
};

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


}  // namespace imgan_dcd
