// Device-side union-find primitives shared by cc.hip and dcd.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace imgan {

__device__ __forceinline__ int cc_find(const int32_t* parent, int x) {
    // parents only ever decrease, so a stale read costs extra steps, never a wrong root
    int p = __atomic_load_n(&parent[x], __ATOMIC_RELAXED);
    while (p != x) {
        x = p;
        p = __atomic_load_n(&parent[x], __ATOMIC_RELAXED);
    }
    return x;
}

__device__ __forceinline__ void cc_union(int32_t* parent, int a, int b) {
    while (true) {
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) return;
        if (a < b) {
            int t = a;
            a = b;
            b = t;
        }
        int old = atomicMin(&parent[a], b);     // hook the larger root under the smaller
        if (old == a) return;
        a = old;
    }
}

}  // namespace imgan
