// The leave-one-out cut of rag_segmentation.py:163-170, shared by the device merge
// (rag_merge.hip) and the host merge (rag.hip) so that both decide every edge the same way:
//     keep edge i  iff  w_i < mean(others) + c * std(others)      (population std, float64)
// for all edges at once from two sums over the whole forest, T = sum w and Q = sum w * w, both
// over the SAME rounded float64 weights and both carried in double-double (an error-free product
// for w * w): the second moment then agrees with the mean to ~1e-32, so when the other weights
// are all equal the variance comes out as 0 -- as numpy's does for one other weight (m = 2) or a
// uniform forest -- and not as the +-2e-16 w^2 that an exact integer w^2 beside a rounded sqrt
// leaves (which kept an edge of weight w beside others of the same weight, where the reference's
// strict '<' drops it).  A variance below 1e-24 of the mean square is arithmetic noise (weights
// that are square roots of integers <= 195,075 and not all equal have a variance above 1e-15 of
// it) and counts as 0.
#pragma once
#include <math.h>

#ifndef IMGAN_HD
#ifdef __HIPCC__
#define IMGAN_HD __host__ __device__ inline
#else
#define IMGAN_HD inline
#endif
#endif

namespace imgan_cut {

struct DD {
    double hi, lo;
};
IMGAN_HD DD two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return DD{s, (a - (s - bb)) + (b - bb)};
}
IMGAN_HD DD dd_add(DD x, DD y) {
    DD s = two_sum(x.hi, y.hi);
    const double e = s.lo + (x.lo + y.lo);
    const double hi = s.hi + e;
    return DD{hi, e - (hi - s.hi)};
}
IMGAN_HD DD dd_neg(DD x) { return DD{-x.hi, -x.lo}; }
IMGAN_HD DD dd_square(double w) {                 // w * w, error free
    const double p = w * w;
    return DD{p, fma(w, w, -p)};
}
IMGAN_HD DD dd_div(DD x, double n) {              // x / n for a small integer n
    const double q = x.hi / n;
    const double r = (fma(-q, n, x.hi) + x.lo) / n;
    const double hi = q + r;
    return DD{hi, r - (hi - q)};
}

// w: the edge's weight; T, Q: sum of the weights and of their squares over the m forest edges
IMGAN_HD bool keep_edge(double w, DD T, DD Q, int m, double c_factor) {
    if (m < 2) return false;                      // mean of nothing is NaN: dropped (:166-168)
    const double n1 = (double)(m - 1);
    const DD mean = dd_div(dd_add(T, DD{-w, 0.0}), n1);
    const DD second = dd_div(dd_add(Q, dd_neg(dd_square(w))), n1);
    const double sq_hi = mean.hi * mean.hi;
    const double sq_lo = fma(mean.hi, mean.hi, -sq_hi) + 2.0 * mean.hi * mean.lo;
    double var = (second.hi - sq_hi) + (second.lo - sq_lo);
    if (!(var > 1.0e-24 * sq_hi)) var = 0.0;
    return w < mean.hi + (mean.lo + c_factor * sqrt(var));
}

}  // namespace imgan_cut
