/* TEST INFRASTRUCTURE -- plain C restatement of the two loops that carry the CPU cost of the
 * path, for (i) a second, independent statement of the algorithm that the numpy oracle is checked
 * against (tests/test_oracle_c.py) and (ii) a fair multi-core CPU baseline in bench.py
 * (one image per OpenMP thread).  Nothing under image-analysis_amd/ loads this library.
 *
 *   oracle_slic_run     <- scripts/slic.py:35-127 (SLICSuperPixel.init_clusters / run), as restated
 *                          in oracle/slic_ref.py, literal float64 distance (slic.py:101-109)
 *   oracle_dcd_segment  <- scripts/mpeg7Descriptors.py:76-130 (dominant_colour), as restated in
 *                          oracle/dcd_ref.py (same stated deviations: injected initial points and
 *                          re-sample points, true 8-connected coherence)
 *
 * Build: oracle/c/build.sh (gcc -O2 -fopenmp -ffp-contract=off: no fused multiply-adds, so the
 * float results are those of the numpy expressions evaluated operation by operation).
 * Parity unpinned by the reference (it holds no fixtures for this path). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int count_range(int lo, int hi, int st) { return hi > lo ? (hi - lo + st - 1) / st : 0; }

/* slic.py:29,33 + :42-43: step and seed grid */
int oracle_slic_grid(int H, int W, int K, int* step, int* nx, int* ny) {
    if (K <= 0 || H <= 0 || W <= 0) return -1;
    int S = (int)sqrt((double)(((int64_t)H * W) / K));
    if (S < 1) return -1;
    *step = S;
    *nx = count_range(S, W - S / 2, S);
    *ny = count_range(S, H - S / 2, S);
    return 0;
}

/* lab: H x W x 3 u8.  labels: H x W int32 (-1 = never assigned).  centres: n x 5 int32
 * (L, a, b, x, y), n = nx * ny, column-major seed order.  Returns n, or < 0. */
int oracle_slic_run(const uint8_t* lab, int H, int W, int K, double cw, int num_iter,
                    int32_t* labels, int32_t* centres) {
    int S, nx, ny;
    if (oracle_slic_grid(H, W, K, &S, &nx, &ny)) return -1;
    const int n = nx * ny;
    const double Sd = (double)S;
#define PX(y, x, c) ((int)lab[((size_t)(y) * W + (x)) * 3 + (c)])
    /* init_clusters (:35-54) with find_local_minimum (:57-76): x outer, y inner, first minimum */
    for (int ix = 0; ix < nx; ++ix)
        for (int iy = 0; iy < ny; ++iy) {
            const int sx = S + ix * S, sy = S + iy * S;
            if (sx + 2 > W - 1 || sy + 2 > H - 1) return -3;   /* the reference raises IndexError here */
            int best = 0x7fffffff, bx = sx, by = sy;
            for (int i = sx - 1; i <= sx + 1; ++i)
                for (int j = sy - 1; j <= sy + 1; ++j) {
                    const int l1 = PX(j + 1, i, 0), l2 = PX(j, i + 1, 0), l3 = PX(j, i, 0);
                    const int g = abs(l1 - l3) + abs(l2 - l3);
                    if (g < best) {
                        best = g;
                        bx = i;
                        by = j;
                    }
                }
            int32_t* c = centres + (size_t)(ix * ny + iy) * 5;
            c[0] = PX(by, bx, 0);
            c[1] = PX(by, bx, 1);
            c[2] = PX(by, bx, 2);
            c[3] = bx;
            c[4] = by;
        }
    double* dist = (double*)malloc((size_t)H * W * sizeof(double));
    int64_t* sums = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * 6 * sizeof(int64_t));
    if (!dist || !sums) {
        free(dist);
        free(sums);
        return -2;
    }
    for (size_t i = 0; i < (size_t)H * W; ++i) labels[i] = -1;
    for (int round = 1; round < num_iter; ++round) {             /* num_iter - 1 rounds (:84) */
        for (size_t i = 0; i < (size_t)H * W; ++i) dist[i] = INFINITY;   /* reset every round (:86) */
        for (int c = 0; c < n; ++c) {                            /* :88-116, centre order */
            const int cx = centres[(size_t)c * 5 + 3], cy = centres[(size_t)c * 5 + 4];
            const int xl = cx - S >= 0 ? cx - S : 0, xr = cx + S < W ? cx + S : W - 1;
            const int yl = cy - S >= 0 ? cy - S : 0, yr = cy + S < H ? cy + S : H - 1;
            if (xr <= xl || yr <= yl) continue;
            /* the colour of the image pixel AT the centre position (:98), not the mean colour */
            const int c0 = PX(cy, cx, 0), c1 = PX(cy, cx, 1), c2 = PX(cy, cx, 2);
            for (int y = yl; y < yr; ++y)
                for (int x = xl; x < xr; ++x) {
                    const int d0 = PX(y, x, 0) - c0, d1 = PX(y, x, 1) - c1, d2 = PX(y, x, 2) - c2;
                    const double dc = sqrt((double)(d0 * d0 + d1 * d1 + d2 * d2));
                    const int dx = x - cx, dy = y - cy;
                    const double ds = sqrt((double)(dx * dx + dy * dy));
                    const double u = dc / cw, v = ds / Sd;
                    const double e = sqrt(u * u + v * v);
                    const size_t i = (size_t)y * W + x;
                    if (dist[i] > e) {                           /* strict: the earlier centre stays */
                        dist[i] = e;
                        labels[i] = c;
                    }
                }
        }
        /* :118-126: centre = floor(sum / count) over the label map, empty -> 0 */
        memset(sums, 0, (size_t)(n > 0 ? n : 1) * 6 * sizeof(int64_t));
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const int l = labels[(size_t)y * W + x];
                if (l < 0) continue;
                int64_t* s = sums + (size_t)l * 6;
                s[0] += PX(y, x, 0);
                s[1] += PX(y, x, 1);
                s[2] += PX(y, x, 2);
                s[3] += x;
                s[4] += y;
                s[5] += 1;
            }
        for (int c = 0; c < n; ++c)
            for (int f = 0; f < 5; ++f)
                centres[(size_t)c * 5 + f] = sums[(size_t)c * 6 + 5] ? (int32_t)(sums[(size_t)c * 6 + f] / sums[(size_t)c * 6 + 5]) : 0;
    }
#undef PX
    free(dist);
    free(sums);
    return n;
}

static int uf_find(int* par, int a) {
    while (par[a] != a) {
        par[a] = par[par[a]];
        a = par[a];
    }
    return a;
}

/* One segment.  pts: N x 3 u8 Lab; rc: N x 2 int32 (row, col) of the same pixels; init: k point
 * indices; resample: k point indices (the point cluster j takes whenever it is empty, :101) or
 * NULL (it keeps its centre).  Out: assign int32[N], centres float[k*3], stats uint64[k*8] = {count, sum0, sum1,
 * sum2, sumsq0, sumsq1, sumsq2, largest 8-connected component}, rounds.  Returns 0, or < 0. */
int oracle_dcd_segment(const uint8_t* pts, const int32_t* rc, int N, const int32_t* init,
                       const int32_t* resample, int k, int max_rounds, int32_t* assign, float* centres, uint64_t* stats,
                       int32_t* rounds_out) {
    if (N < 1 || k < 1 || k > 64) return -1;
    float old[64][3];
    double old_sum = 0.0;
    for (int j = 0; j < k; ++j)
        for (int f = 0; f < 3; ++f) {
            centres[j * 3 + f] = (float)pts[(size_t)init[j] * 3 + f];     /* :79-81 */
            old[j][f] = 0.0f;
        }
    for (int i = 0; i < N; ++i) assign[i] = -1;
    int rounds = 0;
    uint64_t cnt[64], sum[64][3];
    while (1) {                                                           /* :85-107 */
        ++rounds;
        for (int i = 0; i < N; ++i) {
            const float f0 = (float)pts[(size_t)i * 3], f1 = (float)pts[(size_t)i * 3 + 1],
                        f2 = (float)pts[(size_t)i * 3 + 2];
            double best = INFINITY;                                       /* float64 inf (:87) */
            int bi = assign[i];
            for (int j = 0; j < k; ++j) {
                const float e0 = f0 - centres[j * 3], e1 = f1 - centres[j * 3 + 1],
                            e2 = f2 - centres[j * 3 + 2];
                const float q = (e0 * e0 + e1 * e1) + e2 * e2;            /* float32 sum, in order */
                const float d = sqrtf(q);
                if (best > (double)d) {                                   /* strict '>' (:90) */
                    best = (double)d;
                    bi = j;
                }
            }
            assign[i] = bi;
        }
        for (int j = 0; j < k; ++j) cnt[j] = sum[j][0] = sum[j][1] = sum[j][2] = 0;
        for (int i = 0; i < N; ++i) {
            const int j = assign[i];
            cnt[j] += 1;
            for (int f = 0; f < 3; ++f) sum[j][f] += pts[(size_t)i * 3 + f];
        }
        for (int j = 0; j < k; ++j)
            if (cnt[j])                                   /* float32 sums of small integers are exact */
                for (int f = 0; f < 3; ++f) centres[j * 3 + f] = (float)sum[j][f] / (float)cnt[j];
            else if (resample)                            /* :101, injected draw */
                for (int f = 0; f < 3; ++f) centres[j * 3 + f] = (float)pts[(size_t)resample[j] * 3 + f];
        float sh[3] = {0.0f, 0.0f, 0.0f};
        for (int j = 0; j < k; ++j)
            for (int f = 0; f < 3; ++f) sh[f] = sh[f] + fabsf(centres[j * 3 + f] - old[j][f]);
        const double cent_sum = ((double)sh[0] + (double)sh[1]) + (double)sh[2];   /* :102 */
        if (fabs(old_sum - cent_sum) < 0.1 || rounds >= max_rounds) break;
        for (int j = 0; j < k; ++j)
            for (int f = 0; f < 3; ++f) old[j][f] = centres[j * 3 + f];
        old_sum = cent_sum;
    }
    *rounds_out = rounds;
    /* :110-127: per colour count, sums, squares, largest 8-connected component */
    int rmin = rc[0], rmax = rc[0], cmin = rc[1], cmax = rc[1];
    for (int i = 1; i < N; ++i) {
        if (rc[2 * i] < rmin) rmin = rc[2 * i];
        if (rc[2 * i] > rmax) rmax = rc[2 * i];
        if (rc[2 * i + 1] < cmin) cmin = rc[2 * i + 1];
        if (rc[2 * i + 1] > cmax) cmax = rc[2 * i + 1];
    }
    const int bw = cmax - cmin + 3, bh = rmax - rmin + 3;              /* one cell of margin */
    int* grid = (int*)malloc((size_t)bw * bh * sizeof(int));           /* point index + 1, 0 = none */
    int* par = (int*)malloc((size_t)N * sizeof(int));
    int* size = (int*)calloc((size_t)N, sizeof(int));
    if (!grid || !par || !size) {
        free(grid);
        free(par);
        free(size);
        return -2;
    }
    memset(grid, 0, (size_t)bw * bh * sizeof(int));
    for (int i = 0; i < N; ++i) {
        grid[(size_t)(rc[2 * i] - rmin + 1) * bw + (rc[2 * i + 1] - cmin + 1)] = i + 1;
        par[i] = i;
    }
    static const int NB[4][2] = {{-1, -1}, {-1, 0}, {-1, 1}, {0, -1}};   /* earlier neighbours */
    for (int i = 0; i < N; ++i) {
        const int r = rc[2 * i] - rmin + 1, c = rc[2 * i + 1] - cmin + 1;
        for (int t = 0; t < 4; ++t) {
            const int o = grid[(size_t)(r + NB[t][0]) * bw + (c + NB[t][1])];
            if (o && assign[o - 1] == assign[i]) {
                const int a = uf_find(par, i), b = uf_find(par, o - 1);
                if (a != b) par[a > b ? a : b] = a > b ? b : a;
            }
        }
    }
    for (int j = 0; j < k; ++j)
        for (int f = 0; f < 8; ++f) stats[j * 8 + f] = 0;
    for (int i = 0; i < N; ++i) {
        const int j = assign[i];
        stats[j * 8] += 1;
        for (int f = 0; f < 3; ++f) {
            const uint64_t v = pts[(size_t)i * 3 + f];
            stats[j * 8 + 1 + f] += v;
            stats[j * 8 + 4 + f] += v * v;
        }
        const int root = uf_find(par, i);
        size[root] += 1;
        if ((uint64_t)size[root] > stats[j * 8 + 7]) stats[j * 8 + 7] = (uint64_t)size[root];
    }
    free(grid);
    free(par);
    free(size);
    return 0;
}

/* The CPU baseline of bench.py: B images, one per OpenMP thread; SLIC, then the dominant colours
 * of every superpixel (k initial points spread evenly over the segment's raster order).
 * lab: B x H x W x 3.  labels: B x H x W scratch / output.  Returns the threads used, or < 0. */
#ifdef _OPENMP
#include <omp.h>
#endif
int oracle_path_batch(const uint8_t* lab, int B, int H, int W, int K, double cw, int num_iter,
                      int k, int threads, int32_t* labels, int64_t* rounds_total) {
    int S, nx, ny;
    if (oracle_slic_grid(H, W, K, &S, &nx, &ny)) return -1;
    const int n = nx * ny;
    int used = 1, fail = 0;
    int64_t rt = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = threads > 0 ? threads : omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : rt) reduction(| : fail)
    for (int b = 0; b < B; ++b) {
        const uint8_t* im = lab + (size_t)b * H * W * 3;
        int32_t* lb = labels + (size_t)b * H * W;
        int32_t* cen = (int32_t*)malloc((size_t)(n > 0 ? n : 1) * 5 * sizeof(int32_t));
        if (!cen || oracle_slic_run(im, H, W, K, cw, num_iter, lb, cen) < 0) {
            fail |= 1;
            free(cen);
            continue;
        }
        free(cen);
        /* bucket the pixels by label (raster order inside a bucket) */
        int* start = (int*)calloc((size_t)n + 2, sizeof(int));
        int* order = (int*)malloc((size_t)H * W * sizeof(int));
        for (size_t i = 0; i < (size_t)H * W; ++i)
            if (lb[i] >= 0) start[lb[i] + 2] += 1;
        for (int s = 0; s < n; ++s) start[s + 2] += start[s + 1];
        for (size_t i = 0; i < (size_t)H * W; ++i)
            if (lb[i] >= 0) order[start[lb[i] + 1]++] = (int)i;
        for (int s = 0; s < n; ++s) {
            const int lo = start[s], N = start[s + 1] - lo;
            if (N < k) continue;
            uint8_t* pts = (uint8_t*)malloc((size_t)N * 3);
            int32_t* rc = (int32_t*)malloc((size_t)N * 2 * sizeof(int32_t));
            int32_t* as = (int32_t*)malloc((size_t)N * sizeof(int32_t));
            int32_t init[64], resample[64];
            float c[64 * 3];
            uint64_t st[64 * 8];
            int32_t rounds = 0;
            for (int i = 0; i < N; ++i) {
                const int p = order[lo + i];
                memcpy(pts + (size_t)i * 3, im + (size_t)p * 3, 3);
                rc[2 * i] = p / W;
                rc[2 * i + 1] = p % W;
            }
            for (int j = 0; j < k; ++j) {
                init[j] = (int)(((int64_t)j * N) / k);
                resample[j] = (int)(((int64_t)(2 * j + 1) * N) / (2 * k));
            }
            if (oracle_dcd_segment(pts, rc, N, init, resample, k, 1000, as, c, st, &rounds)) fail |= 1;
            rt += (int64_t)rounds * N;
            free(pts);
            free(rc);
            free(as);
        }
        free(start);
        free(order);
    }
    if (rounds_total) *rounds_total = rt;
    return fail ? -2 : used;
}
