// Dominant colours, the tier kernels of imgan_dcd_segments_listed (mpeg7Descriptors.py:76-129):
// one workgroup owns one segment, reads its run of the pixel lists once and runs the whole Lloyd
// loop, the statistics and the coherence pass on chip.
//
// The rounds walk the DISTINCT COLOURS of the segment: the assignment of a pixel depends on its
// colour alone and the sums are integers, so one entry per distinct colour, weighted by the number
// of pixels that have it, gives the same centres, assignments and round counts at a fraction of
// the distance evaluations (0.41 distinct colours per pixel on the bench batch).  The colours are
// collected in an LDS hash table while the run is loaded, closed up into dense arrays
// {colour | cluster << 24, count}, and every pixel keeps the index of its colour (2 B).
//
// What differs from the kernels in dcd_seg.hip (which stay as the box-gather front end and as the
// plain body for segments whose colours do not fit a table):
//   * TWO barriers per Lloyd round and no clearing.  The accumulators are two buffers used in
//     turn and never cleared (a round's sums are the buffer's totals minus the totals seen two
//     rounds ago).  After the round's first barrier wave 0 alone folds the buffer, moves the centres
//     and applies the stop rule, and leaves the new centres and the decision in LDS; after the
//     second barrier every wave reads them (one LDS word per lane, 24 lane reads).  Every wave
//     folding and updating for itself saved the second barrier but cost 8 x 200 vector
//     instructions per round in a kernel whose vector pipe is the busiest unit (90 % by
//     SQ_ACTIVE_INST_VALU): measured 3.77 -> 3.63 ms per stage at 64 images with the fold on one
//     wave.
//   * the distances of a round in expanded, centred form, two clusters per v_pk_fma_f32 with a
//     scalar register pair for the centres' side, and a per-entry margin that sends near ties to
//     the literal rule (comment at the rounds): 12 packed instructions per entry instead of 48
//     plain ones, 12.95 -> 12.5 ms per call at 256 images.  (Packed fp32 on the subtract form had
//     measured no gain: a v_pk_* takes the issue time of the two plain operations it replaces --
//     unless those carry a scalar operand, which doubles theirs and not the packed one's.)
//   * no calls and no second body: an empty cluster is re-seeded by the whole block in line (the
//     mask is the same in every wave, so the branch is uniform over the block), and a segment
//     whose colours overflow the table is put on the next tier's list (larger table) -- the last
//     tier hands it to the plain body of dcd_seg.hip, launched after the tiers.
//   * the assignment map is optional (the batch pipeline does not read it).
// Tried and left out: thread 0 fetching the workgroup's next ticket, list entry, run and box while
// the current segment is worked on (three round trips to global memory off the critical path).  The
// build with it hung in the test suite: nine more values alive across the whole body under a
// one-lane branch, on top of ~400 scalar-register spills, is the situation DESIGN.md's "fragility"
// note describes; the chain costs about 3 % of a segment.
//
// Arithmetic is the reference's, as in dcd_seg.hip: float32 distance sqrt((d0^2+d1^2)+d2^2) with a
// strict '<' (evaluated literally only when an entry's two smallest keys are within the error margin of
// the expanded form, see the rounds),
// integer sums, float32 centre = sum / count, float32 shift sums added in float64.
#include "common.h"
#include "../../include/imgan.h"
#include "dcd_common.h"
#include "dcd_lists.h"

using namespace imgan_dcd;

namespace {

#ifdef IMGAN_TIER_TIMING
// phase clocks of thread 0, summed over the segments (tools/tier_timing.py; never built into the
// product): 100 MHz ticks per phase, [31] = segments
__device__ unsigned long long tier_timing[32];
#define TIER_MARK(i)                                                    \
    if (tid == 0) {                                                     \
        const unsigned long long now_ = wall_clock64();                 \
        atomicAdd(&tier_timing[i], now_ - t_last_);                     \
        t_last_ = now_;                                                 \
    }
#elif defined(IMGAN_TIER_PROGRESS)
// debugging aid (tools/hang_watch.py; never built into the product): every workgroup's thread 0 leaves
// {segment, phase} in host-visible memory
__device__ volatile unsigned* tier_progress;
#define TIER_MARK(i)                                                                      \
    if (tid == 0 && tier_progress) {                                                      \
        tier_progress[blockIdx.x * 4 + 0] = (unsigned)gs;                                 \
        tier_progress[blockIdx.x * 4 + 1] = (unsigned)(i) | ((unsigned)CAP << 8);         \
        tier_progress[blockIdx.x * 4 + 2] += 1u;                                          \
        __threadfence_system();                                                           \
    }
#else
#define TIER_MARK(i)
#endif

#ifdef IMGAN_TIE_COUNTS
// how often the expanded distances hand an entry to the exact rule (never built into the product):
// [0] wave trips, [1] trips with a hand-over, [2] entries, [3] handed over, [4] of those with equal keys,
// [5] handed over in a segment's first round, [7] second round, [6] handed over with two candidates at
// one literal distance
__device__ unsigned long long tie_counts[8];
#endif

struct TierArgs {
    const uint8_t* lab;
    int B, H, W, nseg, k, use_hash;
    uint32_t seed;
    float* centres;
    uint8_t* assign;                    // may be NULL: the map is not written
    SegState* seg;
    unsigned long long* stats;
    const int* bbox;
    int max_rounds;
    SegLists L;
    int tier, next_tier;                // list this launch works through / list for table overflows
};

// bytes behind the 64-bit slots: flags, centres, expanded centres
#define TIER_TAIL(km) ((size_t)128 + (size_t)(km) * 32)

// lanes that share an accumulator word (LDS atomics): a buffer is 1024 words either way
__host__ __device__ constexpr int tier_copies(int km) { return 256 / km; }

__host__ __device__ constexpr size_t tier_lds_bytes(int km, int cap) {
    // table + counts + colour indices (6 B per pixel of capacity), two accumulator buffers,
    // the 64-bit slots, 32 words of flags, the centres of the round (km float4) and their
    // expanded form (4 km floats: -2 c', |c'|^2)
    return (size_t)cap * 6 + (size_t)2 * km * 4 * tier_copies(km) * 4 + (size_t)km * 8 + TIER_TAIL(km);
}

constexpr int tier_waves_per_simd(int km, int cap, int threads) {
    return (threads / 256) * (int)((160 * 1024) / tier_lds_bytes(km, cap));
}

// plain v_min_f32 / v_med3_f32 (the builtins come with a canonicalising instruction each)
__device__ __forceinline__ float fmin2(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float fmed3(float a, float b, float c) {
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// the cluster index in the low mantissa bits of a key: (t & ~(KM - 1)) | j, one v_and_or_b32 (the same
// as a v_bitop3_b32 with both constants in line, a 2-cycle instruction on paper, was measured slower)
template <int KM>
__device__ __forceinline__ float with_index(float t, int j) {
    return __uint_as_float((__float_as_uint(t) & ~(uint32_t)(KM - 1)) | (uint32_t)j);
}

__device__ __forceinline__ float lane_f(float v, int j) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}

// One segment, start to finish, by the whole workgroup.
template <int KM, int CAP, int THREADS>
__device__ __forceinline__ void dedup_segment(unsigned char* smem, const TierArgs& A, const int gs) {
    constexpr int TU = (CAP * 2 / 3) / 64 * 64;            // table slots: 4 TU + 2 TU + 2 CAP <= 6 CAP
    constexpr int TS = 64;                                 // of which the last TS are not hashed to
    constexpr int TH = TU - TS;
    constexpr int ROWS = KM * 4;                           // {count, sum0, sum1, sum2} per cluster
    constexpr int COPIES = tier_copies(KM);                // 32 (KM = 8) or 16
    constexpr int BUF = ROWS * COPIES;                     // words per accumulator buffer: 1024
    static_assert(BUF == 1024, "the fold takes 16 words per lane");
    uint32_t* tab = reinterpret_cast<uint32_t*>(smem);     // hash table, then dense colour | cluster << 24
    uint16_t* ucnt = reinterpret_cast<uint16_t*>(tab + TU);
    uint16_t* idx = ucnt + TU;                             // per pixel: slot, then dense index
    // The accumulators start where the plan's 6 bytes per pixel end -- NOT right behind idx: TU is
    // rounded down to a multiple of 64, so table + counts + indices can be up to 256 bytes short of
    // 6 CAP, and the coherence pass lays its 2-byte cell grid over everything up to 6 CAP
    // (tier_fits in dcd_seg.hip: 2 area + 8 <= 6 CAP).  With the accumulators right behind idx a box
    // of 21,377 .. 21,500 cells (CAP 7168) put its last parents on top of the first 128 entries of
    // the join list, which lives in the accumulators' place: union-find walks over garbage, wrong
    // coherence values or a workgroup that never comes back, depending on the order in which the
    // join list was filled.
    static_assert((size_t)TU * 6 + (size_t)CAP * 2 <= (size_t)CAP * 6, "the table does not fit its share");
    uint32_t* acc = reinterpret_cast<uint32_t*>(smem + (size_t)CAP * 6);
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(acc + 2 * BUF);
    int* flags = reinterpret_cast<int*>(slots + KM);
    float* cen4 = reinterpret_cast<float*>(flags + 32);     // [cluster][component, 4th unused], written by wave 0
    float* cenx = cen4 + KM * 4;                           // [-2 c'0, -2 c'1, -2 c'2, |c'|^2][cluster]
    float* fflags = reinterpret_cast<float*>(flags);       // [8..10] centring point g, [11..12] the round's tie margin
    int& s_round = flags[5];                               // bit 0: stop, bit 1: a cluster ran empty
    int& s_full = flags[0];
    int& s_unique = flags[1];                              // ([2]: the ticket)
    int& s_joins = flags[3];
    int& s_stack = flags[4];

    // This body runs in the ticket loop: without the fence the compiler computes everything that
    // depends on the thread index alone (a few dozen LDS addresses) once, ahead of the loop, keeps
    // it alive across the whole kernel and ends up reloading it from scratch memory before every
    // use (the close-up of the table took 19 % of a segment's time that way).  The wave index is
    // declared uniform, so that tests on it are scalar branches.
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = A.k, W = A.W, H = A.H;
    const SegLists& L = A.L;
    const int b = gs / A.nseg;
    const int* bb = A.bbox + (size_t)gs * 4;
    // clamped into the image: boxes may come from the caller
    const int xmin = max(bb[0], 0), ymin = max(bb[1], 0);
    const int xmax = min(bb[2], W - 1), ymax = min(bb[3], H - 1);
    SegState st;
    st.old_sum = 0.0;
    st.done = 1;
    st.rounds = 0;
    st.handled = 1;
    st.pending = 0;
    unsigned long long* out = A.stats + (size_t)gs * k * 8;
    const uint32_t run_at = L.offsets[gs];
    const uint32_t run_n = min(L.ends[gs] - run_at, L.cursor[gs]);
    if (xmax < xmin || ymax < ymin || run_n == 0) {        // segment id without pixels
        if (tid == 0) A.seg[gs] = st;
        for (int i = tid; i < k * 8; i += THREADS) out[i] = 0ull;
        return;
    }
    const int bw = xmax - xmin + 1, bh = ymax - ymin + 1;
    // The grid of the coherence pass is the box with a border of one cell above, left and right (its
    // cells never hold a cluster): a cell's four earlier neighbours are then four plain reads, with
    // no test for the box's edges (seg_plan_kernel sizes the tiers for the padded grid).
    const int pw = bw + 2;
    const int area = pw * (bh + 1);
    const size_t img = (size_t)b * H * W;
    const int n = (int)min(run_n, (uint32_t)CAP);
    // the segment's run: one scalar base, 32-bit indices below CAP (indexing the whole list takes
    // 64-bit vector arithmetic per access)
    const uint32_t* __restrict__ run = L.sxy + run_at;
#ifdef IMGAN_TIER_TIMING
    unsigned long long t_last_ = wall_clock64();
    if (tid == 0) atomicAdd(&tier_timing[31], 1ull);
#endif

    // ---- the segment's run into the colour table
    if (tid == 0) s_full = 0, s_stack = 0;
    if (tid < KM) slots[tid] = ~0ull;
    for (int i = tid; i < TU / 4; i += THREADS)
        reinterpret_cast<uint4*>(tab)[i] = make_uint4(NO_COLOUR, NO_COLOUR, NO_COLOUR, NO_COLOUR);
    for (int i = tid; i < TU / 8; i += THREADS) reinterpret_cast<uint4*>(ucnt)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < 2 * BUF / 4; i += THREADS) reinterpret_cast<uint4*>(acc)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    TIER_MARK(0)
    {
        const bool kpow2 = (k & (k - 1)) == 0;
        constexpr int LU = 4;                              // loads in flight per thread
        // 32-bit offsets from the image's first byte where the image allows them
        const bool narrow = (unsigned long long)H * (unsigned long long)W * 3ull < 0xffffffffull;
        const uint8_t* lab_img = A.lab + img * 3;
        const uint32_t last_off = b == A.B - 1 ? ((uint32_t)H * (uint32_t)W - 1u) * 3u : 0xffffffffu;
        for (int i0 = 0; i0 < n; i0 += THREADS * LU) {
            // a table that has been given up is not worth the rest of the run
            if (*reinterpret_cast<volatile int*>(&s_full)) break;
            uint32_t v[LU], xy[LU];
#pragma unroll
            for (int u = 0; u < LU; ++u) xy[u] = run[min(i0 + u * THREADS + tid, n - 1)];
            if (narrow) {                                  // the colours, by position
#pragma unroll
                for (int u = 0; u < LU; ++u)
                    v[u] = lab_at32(lab_img, (int)(xy[u] & 0xffffu), (int)(xy[u] >> 16), H, W, last_off);
            } else {
#pragma unroll
                for (int u = 0; u < LU; ++u)
                    v[u] = lab_at(A.lab, img, (int)(xy[u] & 0xffffu), (int)(xy[u] >> 16), H, W,
                                  (size_t)A.B * H * W - 1);
            }
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                const int i = i0 + u * THREADS + tid;
                if (i >= n) continue;
                // the colour's slot: double hashing (the step comes from the hash's low bits), which
                // keeps the probe sequences short at the loads met here -- linear probing gave up
                // (64 probes) on 8 % of the bench batch's segments at loads of 0.7 to 0.9.  A colour
                // that still finds no slot takes one of the last TS slots as an entry of its own
                // (entries may repeat a colour: the sums do not change); only when those are gone
                // too is the segment handed on.
                const uint32_t c = v[u] & 0xffffffu;
                const uint32_t hc = mix32(c);
                uint32_t slot = __umulhi(hc, (uint32_t)TH);
                const uint32_t step = 1u + (hc & 62u);     // odd steps < 64 << TH
                int probe = 0;
                // NOT unrolled: as 64 nested one-lane branches this loop was 6,000 lines of code and the
                // source of nearly all of the 400 scalar-register spills the kernel once had (it ran
                // 2 % faster that way; partly unrolled, by 8, it is 5 % slower than either)
#pragma unroll 1
                for (; probe < DEDUP_PROBES; ++probe) {
                    const uint32_t seen = atomicCAS(&tab[slot], NO_COLOUR, c);
                    if (seen == NO_COLOUR || seen == c) break;
                    slot += step;
                    slot -= slot >= (uint32_t)TH ? (uint32_t)TH : 0u;
                }
                if (probe == DEDUP_PROBES) {
                    const int at = atomicAdd(&s_stack, 1);
                    if (at < TS) {
                        slot = (uint32_t)(TH + at);
                        tab[slot] = c;
                    } else {
                        s_full = 1;
                        slot = 0;
                    }
                }
                idx[i] = (uint16_t)slot;
                // two 16-bit counts per word; no carry: a count is at most n < 65536
                atomicAdd(reinterpret_cast<uint32_t*>(ucnt) + (slot >> 1), 1u << ((slot & 1u) * 16u));
                if (A.use_hash) {
                    // stand-in for random.sample (:79): slot h % k goes to the smallest (h, p)
                    const int x = min(max((int)(xy[u] & 0xffffu), xmin), xmax);
                    const int y = min(max((int)(xy[u] >> 16), ymin), ymax);
                    const uint32_t p = (uint32_t)(__mul24(y, W) + x);
                    const uint32_t h = mix32(p * 0x9e3779b1u + A.seed);
                    const unsigned long long key = ((unsigned long long)h << 32) | (unsigned long long)p;
                    unsigned long long* sl = &slots[kpow2 ? (h & (uint32_t)(k - 1)) : h % (uint32_t)k];
                    if (key < *sl) atomicMin(sl, key);
                }
            }
        }
    }
    __syncthreads();
    TIER_MARK(1)
    if (s_full) {
        // more colours than the table holds: the next tier has a larger one; the last tier's list
        // goes to the plain body (dcd_seg.hip)
        if (tid == 0) {
            const int at = atomicAdd(&L.hdr[A.next_tier], 1);
            L.tier[A.next_tier][at] = gs;
        }
        return;
    }
    // ---- close the table up.  Every thread takes TC consecutive slots (colour and count) into
    // registers; a running sum of the occupied ones over the block gives every colour its dense
    // index, left in its slot for the moment; the pixels swap their slot for the index; the
    // colours and counts move to their dense places.
    {
        constexpr int TC = (TU + THREADS - 1) / THREADS;
        uint32_t* wsum = acc;                              // a few words of a buffer cleared again below
        uint32_t held[TC], occupied = 0;
        uint16_t held_n[TC];
#pragma unroll
        for (int j = 0; j < TC; ++j) {
            const int slot = min(tid * TC + j, TU - 1);
            held[j] = tab[slot];
            held_n[j] = ucnt[slot];
            if (tid * TC + j < TU && held[j] != NO_COLOUR) occupied |= 1u << j;
        }
        const int mine = __popc(occupied);
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wsum[wave] = (uint32_t)incl;
        __syncthreads();
        int before = 0, total = 0;
        for (int q = 0; q < THREADS / 64; ++q) {
            const int ws = (int)wsum[q];
            before += q < wave ? ws : 0;
            total += ws;
        }
        const int first = before + incl - mine;
        {
            int id = first;
#pragma unroll
            for (int j = 0; j < TC; ++j)
                if ((occupied >> j) & 1u) tab[tid * TC + j] = (uint32_t)id++;
        }
        if (tid == 0) s_unique = total;
        __syncthreads();                                   // every slot holds its index now
        if (tid < THREADS / 64) wsum[tid] = 0u;            // the accumulators start from zero
        for (int i = tid; i < n; i += THREADS) idx[i] = (uint16_t)tab[idx[i]];
        __syncthreads();
        {
            int id = first;
#pragma unroll
            for (int j = 0; j < TC; ++j)
                if ((occupied >> j) & 1u) {
                    tab[id] = held[j];
                    ucnt[id] = held_n[j];
                    ++id;
                }
        }
        __syncthreads();
    }
    const int nloop = s_unique;
    TIER_MARK(2)

    // ---- Lloyd rounds (:85-107).  Lane layout of wave 0's per-round state: the
    // 64 lanes of a wave cover the KM x 4 rows {count, sum0, sum1, sum2} of the accumulators, GL
    // lanes per cluster and GL / 4 per row; lane (j, f) holds field f of cluster j after the fold
    // and, for f = 1..3, component f - 1 of centre j, its previous value and its shift.
    constexpr int GL = 64 / KM;                            // lanes per cluster: 8 (KM = 8), 4 (KM = 16)
    constexpr int FL = GL / 4;                             // lanes per row
    const int cj = lane / GL, cf = (lane / FL) & 3;        // this lane's cluster and field
    const bool comp = cf >= 1 && cj < k && (lane & (FL - 1)) == 0;   // holds a centre component
    // lane that owns entry (cj, cf - 1) of the centres in LDS
    const bool holds = cf >= 1 && (lane & (FL - 1)) == 0;
    float wc = 1.0e18f, wo = 0.f;                          // (unused clusters: never the nearest)
    float gl = 0.f;                                        // wave 0: the centring point g in this lane's component
    // the tie margin's coefficients (see the rounds), in units of 2^-24
    constexpr float TIE_UNIT = 5.9604644775390625e-08f;
    constexpr float TIE_U1 = KM == 8 ? 52.f : 86.f, TIE_U2 = 32.f;
    // wave 0: the centres of the next round, as they are (for the exact replay) and expanded, and
    // the coefficients of the round's tie margin (see the rounds)
    // `fresh`: the centres may all be colours of pixels (the initial centres are), i.e. integers.  Then
    // every step of the expanded form is exact (integers below 2^20, whose low mantissa bits are free
    // for the index), the literal rule is the order of the exact q with the lower index first among
    // equals, and with a constant on top that makes every key positive the keys give exactly that:
    // such a round needs no margin.  It matters: between integer centres a lattice of integer
    // colours has many points at equal distance -- 2.8 % of the entries of a segment's first round on
    // the bench batch, so that nine wave trips in ten of that round went through the literal rule.
    auto publish = [&](bool fresh) {
        if (holds) cen4[cj * 4 + (cf - 1)] = wc;           // (LDS operations of a wave stay in order)
        const bool whole = fresh && __ballot(comp && !(wc == rintf(wc) && wc >= 0.f && wc <= 255.f)) == 0ull;
        const float lift = whole ? 195076.f : 0.f;         // 3 x 255^2 + 1 >= |f'|^2 + 1
        const float cp = __fsub_rn(wc, gl);                // the centred component c' (on `comp` lanes)
        float sq = comp ? __fmul_rn(cp, cp) : 0.f;         // |c'|^2 over the cluster's GL lanes
        sq = __fadd_rn(sq, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sq), 0xb1, 0xf, 0xf, true)));
        sq = __fadd_rn(sq, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sq), 0x4e, 0xf, 0xf, true)));
        if (GL == 8)
            sq = __fadd_rn(sq, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sq), 0x141, 0xf, 0xf, true)));
        const bool head = cf == 0 && (lane & (FL - 1)) == 0;         // the lane that holds |c'_j|^2
        const float word = cf == 0 ? (cj < k ? __fadd_rn(sq, lift) : 3.0e38f) : (comp ? -2.f * cp : 0.f);
        if ((lane & (FL - 1)) == 0) cenx[(cf == 0 ? 3 : cf - 1) * KM + cj] = word;
        // C2 = the largest |c'_j|^2 (non-negative floats order as integers)
        uint32_t c2 = head && cj < k ? __float_as_uint(sq) : 0u;
#define ROW_MAX(ctrl) c2 = max(c2, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c2, ctrl, 0xf, 0xf, true));
        ROW_MAX(0xb1) ROW_MAX(0x4e) ROW_MAX(0x141) ROW_MAX(0x140)
#undef ROW_MAX
        const uint32_t c2u = max(max((uint32_t)__builtin_amdgcn_readlane((int)c2, 0), (uint32_t)__builtin_amdgcn_readlane((int)c2, 16)),
                                 max((uint32_t)__builtin_amdgcn_readlane((int)c2, 32), (uint32_t)__builtin_amdgcn_readlane((int)c2, 48)));
        const float C2 = __uint_as_float(c2u);
        if (lane == 0) {
            // centres too large for the expanded form (a caller's initial centres): no margin, every entry
            // of the round takes the exact rule
            const bool sane = C2 < 1.0e30f;
            fflags[11] = whole ? -__builtin_inff()
                         : sane ? (TIE_U1 + TIE_U2) * TIE_UNIT * C2 + 1.0e-30f : __builtin_nanf("");
            fflags[12] = 2.002f * TIE_U1 * TIE_UNIT * __builtin_amdgcn_sqrtf(C2);
        }
    };
    if (wave == 0) {
        const int j = cj < k ? cj : 0, f = cf >= 1 ? cf - 1 : 0;
        float v0;
        if (A.use_hash) {
            unsigned long long fallback = ~0ull;
            for (int q = 0; q < k; ++q) fallback = slots[q] < fallback ? slots[q] : fallback;
            const unsigned long long key = slots[j] != ~0ull ? slots[j] : fallback;
            v0 = (float)A.lab[(img + (uint32_t)key) * 3 + f];
        } else {
            v0 = A.centres[((size_t)gs * k + j) * 3 + f];
        }
        if (comp) wc = v0;
        // The centring point g of the expanded distances: the mean of the initial centres, rounded to
        // integers (pixels of the segment drawn at random: close to its mean colour).  Any integers do
        // for the results; how small |f - g| and |c - g| are decides how rarely the exact rule is called.
        if (holds) cen4[cj * 4 + (cf - 1)] = wc;           // (LDS operations of a wave stay in order)
        {
            float sum = 0.f;
            for (int q = 0; q < k; ++q) sum += cen4[q * 4 + (cf >= 1 ? cf - 1 : 0)];
            gl = rintf(fminf(fmaxf(sum / (float)k, 0.f), 255.f));
            if (holds && cj == 0) fflags[8 + (cf - 1)] = gl;
        }
        publish(true);
    }
    __syncthreads();
    // the centring point, the same in every lane of the block
    const float g0s = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fflags[8])));
    const float g1s = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fflags[9])));
    const float g2s = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fflags[10])));
    double old_sum = 0.0;
    int rounds = 0;
    uint32_t field = 0;                                    // lane (j, f): field f of cluster j, last round
    uint32_t seen0 = 0, seen1 = 0;                         // this lane's running totals of the two buffers
    constexpr uint32_t IDXM = KM - 1;
    // The accumulators are never cleared: two buffers take the rounds in turn (nothing adds to a
    // buffer while a wave may still be reading it: round r adds before barrier r, round r + 2
    // after barrier r + 1, and the reads of round r lie between those two barriers), and a round's
    // sums are the buffer's totals minus the totals this lane saw two rounds ago (mod 2^32).
    int buf = 0;
    const int copy = lane & (COPIES - 1);
    // The distances of a round in expanded form.  With g a fixed point near the segment's mean colour
    // (integers), f' = f - g (exact) and c' = c - g:
    //     |f - c|^2 - |f'|^2 = |c'|^2 - 2 c'0 f'0 - 2 c'1 f'1 - 2 c'2 f'2 =: t
    // -- three fused multiply-adds per cluster instead of three subtracts, a multiply and two of them,
    // and two clusters per v_pk_fma_f32 with the centres' side in a scalar register pair (measured,
    // tools/issue_rate.hip: 4 cycles for the pair; the subtract form is 12 cycles per cluster, its
    // operations with a scalar operand being 4-cycle ones).  |f'|^2 is the same for every cluster of an
    // entry, so the order of the t is the order of the squared distances; t may be negative, the keys
    // are compared as floats.  What the form costs is precision, and the margin pays for it.  With
    // C2 = the largest |c'_j|^2 of the round and S1 = |f'0| + |f'1| + |f'2| >= |f'|, every value met on
    // the way to a key is at most M = C2 + 2 sqrt(C2) S1 in magnitude, so a key differs from its exact
    // value by at most (2 [c' rounded] + 2.5 [|c'|^2] + 3 [the chain] + 2 KM [index in the low mantissa
    // bits: 8 or 16 ulp]) 2^-24 M, and the literal rule (float32 sqrt of a float32 sum of squares,
    // strict '<') follows the exact order whenever two exact q differ by more than 14 x 2^-24 of the
    // larger, which is at most 2 S1^2 + 2 C2.  An entry whose two smallest keys are closer than
    //     margin = A + B S1 + D S1^2,  A = (U1 + 32) 2^-24 C2,  B = 2 U1 2^-24 sqrt(C2),  D = 32 x 2^-24
    // (U1 = 52 for 8 clusters, 86 for 16: the sums above come to 47 and 79 for the two keys; a key
    // further up only gains on its own error) is assigned by the literal rule instead, evaluated for
    // the clusters whose keys lie within that margin of the smallest: every other cluster is farther
    // by the literal rule too.  The margin is a few thousandths for a superpixel of one colour mode
    // and a tenth for one that straddles two (colours and centres 50 and more from g); nothing here
    // depends on g but how often the literal rule is called.  On the bench batch 0.28 % of the
    // entries go there, nearly all of them true ties (clusters with one and the same centre, which
    // the subtract form sent there as well): tighter margins from |f'|^2 and the keys themselves
    // (0.27 %) cost more per entry than they saved.
    typedef float float2v __attribute__((ext_vector_type(2)));
#ifdef IMGAN_TIE_COUNTS
    int rnd_ = -1;
#endif
    while (true) {
#ifdef IMGAN_TIE_COUNTS
        ++rnd_;
#endif
        float2v m0[KM / 2], m1[KM / 2], m2[KM / 2], cc[KM / 2];   // m: scalars, the same in every lane
        {
            const float cv = cenx[lane & (3 * KM <= 32 ? 31 : 63)];
#pragma unroll
            for (int jp = 0; jp < KM / 2; ++jp) {
                m0[jp] = float2v{lane_f(cv, 2 * jp), lane_f(cv, 2 * jp + 1)};
                m1[jp] = float2v{lane_f(cv, KM + 2 * jp), lane_f(cv, KM + 2 * jp + 1)};
                m2[jp] = float2v{lane_f(cv, 2 * KM + 2 * jp), lane_f(cv, 2 * KM + 2 * jp + 1)};
                cc[jp] = *reinterpret_cast<const float2v*>(cenx + 3 * KM + 2 * jp);   // one address for the wave
            }
        }
        // the margin's coefficients, the same in every lane (vector registers: two scalar operands do
        // not fit one instruction)
        const float tolA = fflags[11], tolB = fflags[12];
        float tolD = TIE_U2 * TIE_UNIT;
        asm("" : "+v"(tolD));
        uint32_t* ab = acc + buf * BUF + copy;
        for (int i = tid; i < nloop; i += THREADS) {
            const uint32_t v = tab[i];
            const uint32_t weight = (uint32_t)ucnt[i];
            const uint32_t u0 = v & 255u, u1 = (v >> 8) & 255u, u2 = (v >> 16) & 255u;
            const float f0 = (float)u0 - g0s, f1 = (float)u1 - g1s, f2 = (float)u2 - g2s;
            const float2v F0 = float2v{f0, f0}, F1 = float2v{f1, f1}, F2 = float2v{f2, f2};
            // the two smallest keys (cluster index in the low mantissa bits), compared as floats
            float best = __builtin_inff(), second = __builtin_inff();
#pragma unroll
            for (int jp = 0; jp < KM / 2; ++jp) {
                const float2v t = __builtin_elementwise_fma(
                    F2, m2[jp], __builtin_elementwise_fma(F1, m1[jp], __builtin_elementwise_fma(F0, m0[jp], cc[jp])));
                const float ka = with_index<KM>(t.x, 2 * jp);
                second = fmed3(ka, best, second);
                best = fmin2(best, ka);
                const float kb = with_index<KM>(t.y, 2 * jp + 1);
                second = fmed3(kb, best, second);
                best = fmin2(best, kb);
            }
            int bi = (int)(__float_as_uint(best) & IDXM);
            const float s1 = (fabsf(f0) + fabsf(f1)) + fabsf(f2);
            const float margin = __builtin_fmaf(s1, __builtin_fmaf(s1, tolD, tolB), tolA);
#ifdef IMGAN_TIE_COUNTS
            {
                const bool hit = !(__fsub_rn(second, best) >= margin);
                const bool zero = __fsub_rn(second, best) == 0.f;
                const unsigned long long bh = __ballot(hit), bz = __ballot(zero), ba = __ballot(true);
                if (lane == 0) {
                    atomicAdd(&tie_counts[0], 1ull);
                    atomicAdd(&tie_counts[1], bh != 0ull ? 1ull : 0ull);
                    atomicAdd(&tie_counts[2], (unsigned long long)__popcll(ba));
                    atomicAdd(&tie_counts[3], (unsigned long long)__popcll(bh));
                    atomicAdd(&tie_counts[4], (unsigned long long)__popcll(bz));
                    atomicAdd(&tie_counts[5], rnd_ == 0 ? (unsigned long long)__popcll(bh) : 0ull);
                    atomicAdd(&tie_counts[7], rnd_ == 1 ? (unsigned long long)__popcll(bh) : 0ull);
                }
            }
#endif
            // (negated: a margin that is not a number sends every entry to the exact rule)
            if (__builtin_expect(!(__fsub_rn(second, best) >= margin), 0)) {
                // ---- the literal rule (:88-91), in cluster order with a strict '<', over the clusters
                // whose key is within the margin of the smallest: the others are farther by that rule too
                uint32_t cold = 0;
                asm volatile("" : "+v"(cold));             // (keeps the centre reads inside this block)
                const float4* cq = reinterpret_cast<const float4*>(cen4 + cold);
                const float l0 = (float)u0, l1 = (float)u1, l2 = (float)u2;
                float bs = __builtin_inff();
                bi = 0;
#ifdef IMGAN_TIE_COUNTS
                bool dupl_ = false;
#endif
#pragma unroll
                for (int jp = 0; jp < KM / 2; ++jp) {
                    const float2v t = __builtin_elementwise_fma(
                        F2, m2[jp], __builtin_elementwise_fma(F1, m1[jp], __builtin_elementwise_fma(F0, m0[jp], cc[jp])));
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int j = 2 * jp + h;
                        if (j >= k) continue;                          // (the same in every lane)
                        const float kj = with_index<KM>(h ? t.y : t.x, j);
                        if (!(__fsub_rn(kj, best) >= margin)) {
                            const float4 c = cq[j];
                            const float e0 = __fsub_rn(l0, c.x), e1 = __fsub_rn(l1, c.y), e2 = __fsub_rn(l2, c.z);
                            const float sq = __fsqrt_rn(__fadd_rn(
                                __fadd_rn(__fmul_rn(e0, e0), __fmul_rn(e1, e1)), __fmul_rn(e2, e2)));
#ifdef IMGAN_TIE_COUNTS
                            if (sq == bs) dupl_ = true;
                            if (sq < bs) dupl_ = false;
#endif
                            if (sq < bs) {
                                bs = sq;
                                bi = j;
                            }
                        }
                    }
                }
#ifdef IMGAN_TIE_COUNTS
                atomicAdd(&tie_counts[6], dupl_ ? 1ull : 0ull);
#endif
            }
            tab[i] = (v & 0xffffffu) | ((uint32_t)bi << 24);
            uint32_t* a4 = ab + bi * (4 * COPIES);
            atomicAdd(a4, weight);
            atomicAdd(a4 + COPIES, __umul24(weight, u0));
            atomicAdd(a4 + 2 * COPIES, __umul24(weight, u1));
            atomicAdd(a4 + 3 * COPIES, __umul24(weight, u2));
        }
        TIER_MARK(3)
        __syncthreads();                                   // A: the round's sums are complete
        TIER_MARK(4)
        uint32_t nj = 0;
        // centre update + stop rule (:93-107) on wave 0's lanes; leaves the new centres and the
        // decision in LDS
        auto update = [&](float drawn, bool fresh) {
            // one division for all components
            float shift = 0.f;
            if (comp) {
                wc = nj ? __fdiv_rn((float)field, (float)nj) : drawn;      // or the re-sampled pixel (:101)
                shift = fabsf(__fsub_rn(wc, wo));
            }
            // float32 sums in cluster order (:102): the shifts of component f move to lanes
            // 16 f .. 16 f + KM - 1, a running sum along each row of 16 lanes adds them in order (after
            // s steps the lanes up to s hold their exact prefix), lane 16 f + k - 1 has the total
            const int from = (lane & 15) * GL + ((lane >> 4) + 1) * FL;
            const float moved = __shfl(shift, from);        // every lane takes part: not under the test
            const float x = ((lane >> 4) < 3 && (lane & 15) < k) ? moved : 0.f;
            float run = x;
#pragma unroll
            for (int sft = 1; sft < KM; ++sft) {
                const float up = __int_as_float(
                    __builtin_amdgcn_update_dpp(0, __float_as_int(run), 0x111, 0xf, 0xf, true));   // row_shr:1
                run = __fadd_rn(up, x);
            }
            const float sh0 = lane_f(run, k - 1), sh1 = lane_f(run, 16 + k - 1), sh2 = lane_f(run, 32 + k - 1);
            const double cent_sum = ((double)sh0 + (double)sh1) + (double)sh2;
            rounds += 1;
            const int stop = (fabs(old_sum - cent_sum) < 0.1 || rounds >= A.max_rounds) ? 1 : 0;
            publish(fresh);
            if (lane == 0) s_round = stop;
            old_sum = cent_sum;
            wo = wc;
        };
        if (wave == 0) {
            // ---- fold: a buffer is KM * 4 rows of COPIES words = 16 words per lane
            {
                const uint4* src = reinterpret_cast<const uint4*>(acc + buf * BUF + lane * 16);
                const uint4 p = src[0], q = src[1], r = src[2], t4 = src[3];
                uint32_t tot = (((p.x + p.y) + (p.z + p.w)) + ((q.x + q.y) + (q.z + q.w))) +
                               (((r.x + r.y) + (r.z + r.w)) + ((t4.x + t4.y) + (t4.z + t4.w)));
                if (FL == 2) tot += (uint32_t)__builtin_amdgcn_mov_dpp((int)tot, 0xb1, 0xf, 0xf, true);   // lane ^ 1
                if (buf == 0) {
                    field = tot - seen0;
                    seen0 = tot;
                } else {
                    field = tot - seen1;
                    seen1 = tot;
                }
            }
            // the cluster's count = `field` of the first lane of its group of GL lanes, by DPP (a shuffle
            // is an LDS round trip on the serial path of every round)
            {
                const int q0 = __builtin_amdgcn_mov_dpp((int)field, 0x00, 0xf, 0xf, true);       // quad_perm [0,0,0,0]
                const int up4 = __builtin_amdgcn_update_dpp(0, q0, 0x114, 0xf, 0xf, true);       // row_shr:4
                nj = (uint32_t)(GL == 8 && (lane & 4) ? up4 : q0);
            }
            const unsigned long long empty = __ballot(cj < k && (lane & (GL - 1)) == 0 && nj == 0);
            if (empty == 0ull) {
                update(wc, false);
            } else if (lane == 0) {
                s_round = 2;
                *reinterpret_cast<unsigned long long*>(&flags[6]) = empty;
            }
        }
        __syncthreads();                                   // B: centres and decision are in LDS
        int decision = *reinterpret_cast<volatile int*>(&s_round);
        if (decision & 2) {
            // ---- empty clusters (:101), re-seeded by the whole block.  Cluster j takes the segment's
            // pixel with the smallest resample_key (dcd_common.h).
            if (tid < KM) slots[tid] = ~0ull;
            __syncthreads();
            unsigned long long m = *reinterpret_cast<volatile unsigned long long*>(&flags[6]);
            while (m) {
                const int j = (__ffsll((long long)m) - 1) / GL;
                m &= m - 1;
                unsigned long long best = ~0ull;
                for (int i = tid; i < n; i += THREADS) {
                    const uint32_t xy = run[i];
                    const int x = min(max((int)(xy & 0xffffu), xmin), xmax);
                    const int y = min(max((int)(xy >> 16), ymin), ymax);
                    const unsigned long long key = resample_key((uint32_t)(y * W + x), A.seed, j);
                    best = key < best ? key : best;
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    const unsigned long long other = __shfl_xor(best, o);
                    best = other < best ? other : best;
                }
                if (lane == 0 && best != ~0ull) atomicMin(&slots[j], best);
            }
            __syncthreads();
            if (wave == 0) {
                float drawn = wc;
                if (comp && nj == 0) drawn = (float)A.lab[(img + (uint32_t)slots[cj]) * 3 + (cf - 1)];
                update(drawn, true);
            }
            __syncthreads();
            decision = *reinterpret_cast<volatile int*>(&s_round);
        }
        TIER_MARK(5)
        if (__builtin_amdgcn_readfirstlane(decision) & 1) break;
        buf ^= 1;
    }

    // the positions of this thread's pixels, wanted by the coherence pass: asked for now, all at
    // once (clamped, not branched around), so that they arrive while the statistics are made
    constexpr int NPT = CAP / THREADS;
    static_assert(NPT * THREADS == CAP, "CAP must be a multiple of the block size");
    uint32_t where[NPT];
#pragma unroll
    for (int t = 0; t < NPT; ++t) where[t] = run[min(tid + t * THREADS, n - 1)];

    // ---- statistics (:110-116): counts and sums are the last round's, the squares need a pass
    // over the distinct colours, weighted (a copy's sums stay below n * 255^2 < 2^32)
    __syncthreads();                                       // every wave has left the rounds
    for (int i = tid; i < KM * 3 * COPIES; i += THREADS) acc[i] = 0u;
    __syncthreads();
    for (int i = tid; i < nloop; i += THREADS) {
        const uint32_t v = tab[i], weight = ucnt[i];
        const uint32_t u0 = v & 255u, u1 = (v >> 8) & 255u, u2 = (v >> 16) & 255u, bi = v >> 24;
        uint32_t* a3 = acc + bi * (3 * COPIES) + copy;
        atomicAdd(a3, weight * (u0 * u0));
        atomicAdd(a3 + COPIES, weight * (u1 * u1));
        atomicAdd(a3 + 2 * COPIES, weight * (u2 * u2));
    }
    __syncthreads();
    if (tid < k * 3) {
        const uint4* src = reinterpret_cast<const uint4*>(acc + tid * COPIES);
        unsigned long long sum = 0;
#pragma unroll
        for (int t = 0; t < COPIES / 4; ++t) {
            const uint4 w4 = src[t];
            sum += (unsigned long long)w4.x + w4.y + w4.z + w4.w;
        }
        out[(tid / 3) * 8 + 4 + (tid % 3)] = sum;
    }
    if (wave == 0 && cj < k && (lane & (FL - 1)) == 0) {
        out[cj * 8 + cf] = field;                          // count, sum0, sum1, sum2 of the last round
        if (cf >= 1) A.centres[((size_t)gs * k + cj) * 3 + (cf - 1)] = wc;
    }

    TIER_MARK(6)
    // ---- spatial coherence (:40-74, :118-127): largest 8-connected component of every cluster,
    // found on chip.  Each thread takes its pixels (cell in the box, cluster) into registers -- the
    // positions come from the list again --; the table's place is then reused as a grid over the
    // box: first the cluster of every cell (u8), from which each cell picks its link (kept in its
    // register as a direction), then, in the same place, the parents (u16).
    // (the two table reads of a pixel are not branched around: all of a thread's reads of a kind
    // are asked for together, a slot beyond the run repeats the run's last pixel and is dropped)
    uint32_t keep[NPT];
    constexpr int KG = NPT % 7 == 0 ? 7 : NPT % 5 == 0 ? 5 : NPT % 4 == 0 ? 4 : 1;   // pixels per group of reads
#pragma unroll
    for (int t0 = 0; t0 < NPT; t0 += KG) {
#pragma unroll
        for (int t = t0; t < t0 + KG; ++t) keep[t] = idx[min(tid + t * THREADS, n - 1)];
#pragma unroll
        for (int t = t0; t < t0 + KG; ++t) keep[t] = reinterpret_cast<const uint8_t*>(tab)[keep[t] * 4u + 3u];
#pragma unroll
        for (int t = t0; t < t0 + KG; ++t) {
            const uint32_t xy = where[t], cl = keep[t];
            const int x = min(max((int)(xy & 0xffffu), xmin), xmax);
            const int y = min(max((int)(xy >> 16), ymin), ymax);
            const bool valid = tid + t * THREADS < n;
            keep[t] = valid ? (uint32_t)(__mul24(y - ymin + 1, pw) + (x - xmin + 1)) | (cl << 16) : 0xffffffffu;
            if (A.assign && valid) A.assign[img + (size_t)(uint32_t)(__mul24(y, W) + x)] = (uint8_t)cl;
        }
        __builtin_amdgcn_sched_barrier(0);                 // (keeps the groups apart: registers)
    }
    TIER_MARK(7)
    uint16_t* par = reinterpret_cast<uint16_t*>(smem);
    uint8_t* cls = smem;
    uint32_t* largest = reinterpret_cast<uint32_t*>(slots);
    __syncthreads();
    for (int i = tid; i < (area + 3) / 4; i += THREADS) reinterpret_cast<uint32_t*>(cls)[i] = 0xffffffffu;
    if (tid < KM) largest[tid] = 0;
    // the cells that need an explicit union below, collected in the accumulators' place
    uint16_t* joins = reinterpret_cast<uint16_t*>(acc);
    constexpr int JOINS = 2 * BUF * 2;
    if (tid == 0) s_joins = 0;
    __syncthreads();
    TIER_MARK(9)
#pragma unroll
    for (int t = 0; t < NPT; ++t)
        if (keep[t] != 0xffffffffu) cls[keep[t] & 0xffffu] = (uint8_t)(keep[t] >> 16);
    __syncthreads();
    TIER_MARK(10)
    // Every cell hangs itself under one earlier neighbour of its cluster (N, else W, else NW, else
    // NE; all have a smaller index, so the links form a forest) with a plain store.  Of the four
    // backward links only one more can be needed: with N present the others follow from the
    // neighbours' own links, and W covers NW; what is left is NE next to a missing N, which is
    // united explicitly below.  Those cells go on a list that the whole block works through evenly.
    // (Bit 25: no room on the list.)
    // (no branch per pixel: a thread without a pixel in this slot reads the grid's first cells and
    // compares them with a cluster nothing has; the four reads of a pixel are asked for together)
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
        const bool valid = keep[t] != 0xffffffffu;
        const uint32_t cell = valid ? keep[t] & 0xffffu : (uint32_t)(pw + 1);
        const uint32_t c = valid ? (keep[t] >> 16) & 0xffu : 0xfeu;
        const uint8_t* above = cls + (cell - (uint32_t)pw - 1u);
        const uint32_t vnw = above[0], vn = above[1], vne = above[2], vw = cls[cell - 1u];
        const bool n_ = vn == c, w_ = vw == c, nw = vnw == c, ne = vne == c;
        // bits 26-28, the link as (rows up) << 2 | (columns right + 1): self 1, N 5, W 0, NW 4, NE 6
        const uint32_t dir = n_ ? 5u : w_ ? 0u : nw ? 4u : ne ? 6u : 1u;
        keep[t] |= valid ? dir << 26 : 0u;
        const bool join = !n_ & (w_ | nw) & ne;
        if ((t & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // two pixels' reads in flight, not all
        if (join) {
            const int at = atomicAdd(&s_joins, 1);
            if (at < JOINS)
                joins[at] = (uint16_t)(keep[t] & 0xffffu);
            else
                keep[t] |= 0x2000000u;
        }
    }
    __syncthreads();                                       // every cluster byte has been read
    TIER_MARK(11)
#pragma unroll
    for (int t = 0; t < NPT; ++t)
        if (keep[t] != 0xffffffffu) {
            const uint32_t cell = keep[t] & 0xffffu, dir = (keep[t] >> 26) & 7u;
            par[cell] = (uint16_t)(cell + (dir & 3u) - 1u - __umul24(dir >> 2, (uint32_t)pw));
        }
    if (tid == 0) par[0] = 0;                              // (a border cell: where threads without a pixel point)
    __syncthreads();
    TIER_MARK(12)
    for (int e = tid, ne_ = min(s_joins, JOINS); e < ne_; e += THREADS) {
        const uint32_t cell = joins[e];
        unite16(par, cell, cell - pw + 1);
    }
    if (s_joins > JOINS) {
#pragma unroll
        for (int t = 0; t < NPT; ++t)
            if (keep[t] != 0xffffffffu && (keep[t] & 0x2000000u)) {
                const uint32_t cell = keep[t] & 0xffffu;
                unite16(par, cell, cell - pw + 1);
            }
    }
    __syncthreads();
    TIER_MARK(13)
    // roots into registers (bit 24: this cell is a root), then the parent grid turns into 16-bit
    // size counters (two per word; a segment has < 65536 pixels, so no carry)
    // (Pointer jumping with all of a thread's pixels a step at a time -- NPT reads in flight instead of
    // NPT walks of dependent reads -- was measured slower: waves do not run in step, so a wave's
    // pointers advance one cell at a time through the cells of waves that have not got there yet;
    // with a block barrier per pass slower still.)
#pragma unroll
    for (int t = 0; t < NPT; ++t)
        if (keep[t] != 0xffffffffu) {
            const uint32_t cell = keep[t] & 0xffffu;
            const uint32_t root = find16(par, cell);
            par[cell] = (uint16_t)root;                    // shortens the walks of later cells
            keep[t] = root | (keep[t] & 0xff0000u) | (root == cell ? 0x1000000u : 0u);
        }
    __syncthreads();
    TIER_MARK(14)
    for (int i = tid; i < (area + 1) / 2; i += THREADS) reinterpret_cast<uint32_t*>(par)[i] = 0;
    __syncthreads();
    TIER_MARK(15)
#pragma unroll
    for (int t = 0; t < NPT; ++t)
        if (keep[t] != 0xffffffffu) {
            const uint32_t root = keep[t] & 0xffffu;
            atomicAdd(reinterpret_cast<uint32_t*>(par) + (root >> 1), 1u << ((root & 1u) * 16u));
        }
    __syncthreads();
    TIER_MARK(16)
#pragma unroll
    for (int t = 0; t < NPT; ++t)
        if (keep[t] != 0xffffffffu && (keep[t] & 0x1000000u))
            atomicMax(&largest[(keep[t] >> 16) & 0xffu], (uint32_t)par[keep[t] & 0xffffu]);
    __syncthreads();
    TIER_MARK(17)
    if (tid < k) out[tid * 8 + 7] = largest[tid];
    if (tid == 0) {
        st.rounds = rounds;
        A.seg[gs] = st;
    }
}

template <int KM, int CAP, int THREADS>
__global__ __launch_bounds__(THREADS, tier_waves_per_simd(KM, CAP, THREADS))
void dcd_dedup_kernel(TierArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the workgroups draw tickets for the entries of their list: segments differ in size and in
    // rounds by an order of magnitude, so a fixed split would leave a long tail
    int* ticket = reinterpret_cast<int*>(smem + tier_lds_bytes(KM, CAP) - TIER_TAIL(KM)) + 2;
    const int count = A.L.hdr[A.tier];
#ifdef IMGAN_TIER_TIMING
    const unsigned long long t_begin_ = wall_clock64();
#endif
    while (true) {
        if (threadIdx.x == 0) *ticket = atomicAdd(&A.L.hdr[8 + A.tier], 1);
        __syncthreads();
        const int e = __builtin_amdgcn_readfirstlane(*ticket);
        __syncthreads();
        if (e >= count) break;
        const int gs = __builtin_amdgcn_readfirstlane(A.L.tier[A.tier][e]);
        dedup_segment<KM, CAP, THREADS>(smem, A, gs);
        __syncthreads();
    }
#ifdef IMGAN_TIER_TIMING
    // when the workgroups of a launch start and end: [18 + 4 tier ..] = {~first start, last end, sum of
    // the ends, workgroups} -- last end minus mean end is what the launch's tail costs per slot
    if (threadIdx.x == 0) {
        const unsigned long long t_end_ = wall_clock64();
        atomicMax(&tier_timing[18 + A.tier * 4 + 0], ~t_begin_);
        atomicMax(&tier_timing[18 + A.tier * 4 + 1], t_end_);
        atomicAdd(&tier_timing[18 + A.tier * 4 + 2], t_end_);
        atomicAdd(&tier_timing[18 + A.tier * 4 + 3], 1ull);
    }
#endif
}

template <int KM, int CAP, int THREADS>
int launch(const TierArgs& A, size_t nsegs, int per_cu, hipStream_t st) {
    const size_t lds = tier_lds_bytes(KM, CAP);
    auto fn = dcd_dedup_kernel<KM, CAP, THREADS>;
    IMGAN_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)std::min<size_t>(nsegs, (size_t)256 * per_cu);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(THREADS), lds, st, A);
    return 0;
}

}  // namespace

#ifdef IMGAN_TIER_PROGRESS
extern "C" int imgan_debug_tier_progress(unsigned* host_visible) {
    IMGAN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(tier_progress), &host_visible, sizeof(host_visible)));
    return 0;
}
#endif

#ifdef IMGAN_TIE_COUNTS
extern "C" int imgan_debug_tie_counts(unsigned long long* out8) {
    IMGAN_HIP(hipDeviceSynchronize());
    IMGAN_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(tie_counts), 64));
    return 0;
}
#endif

#ifdef IMGAN_TIER_TIMING
extern "C" int imgan_debug_tier_timing(unsigned long long* out32, int reset) {
    IMGAN_HIP(hipDeviceSynchronize());
    IMGAN_HIP(hipMemcpyFromSymbol(out32, HIP_SYMBOL(tier_timing), 256));
    if (reset) {
        unsigned long long z[32] = {0};
        IMGAN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(tier_timing), z, 256));
    }
    return 0;
}
#endif

// The tier launches of imgan_dcd_segments_listed, smallest tier first: a segment whose colours
// overflow a tier's table moves to the next tier's list, and from the last tier to list 3, which
// the caller hands to the plain body.  Lists and counters are those of seg_plan_kernel.
int imgan_dcd_launch_dedup_tiers(const uint8_t* lab, int B, int H, int W, int nseg, int k, int use_hash,
                                 uint32_t seed, float* centres, uint8_t* assign, SegState* seg,
                                 unsigned long long* stats, const int* bbox, int max_rounds,
                                 const SegLists& L, hipStream_t st) {
    TierArgs A;
    A.lab = lab;
    A.B = B, A.H = H, A.W = W, A.nseg = nseg, A.k = k, A.use_hash = use_hash;
    A.seed = seed;
    A.centres = centres;
    A.assign = assign;
    A.seg = seg;
    A.stats = stats;
    A.bbox = bbox;
    A.max_rounds = max_rounds;
    A.L = L;
    const size_t nsegs = (size_t)B * nseg;
    if (k <= 8) {
        A.tier = 0, A.next_tier = 1;
#ifndef TIER_T0
#define TIER_T0 THREADS_TINY
#endif
#ifndef TIER_T1
#define TIER_T1 THREADS_SMALL
#endif
        if (int rc = launch<8, CAP_TINY_D, TIER_T0>(A, nsegs, 3, st)) return rc;
        A.tier = 1, A.next_tier = 2;
        if (int rc = launch<8, CAP_SMALL_D, TIER_T1>(A, nsegs, 2, st)) return rc;
        A.tier = 2, A.next_tier = 3;
        if (int rc = launch<8, CAP_LARGE, THREADS_LARGE>(A, nsegs, 1, st)) return rc;
    } else {
        A.tier = 1, A.next_tier = 2;
        if (int rc = launch<16, CAP_SMALL_K16, THREADS_SMALL>(A, nsegs, 2, st)) return rc;
        A.tier = 2, A.next_tier = 3;
        if (int rc = launch<16, CAP_LARGE, THREADS_LARGE>(A, nsegs, 1, st)) return rc;
    }
    return 0;
}
