// VALU issue cost of one SIMD of gfx950, measured: time per wave-instruction for eight independent
// chains of one instruction at 1, 2, 4 and 8 waves per SIMD, every CU busy.  Printed as ns per
// wave-instruction and SIMD and as shader cycles at the clock the run sustained (the elapsed
// s_memtime ticks of the waves against the elapsed time give that clock).  Every "% of the issue
// slots" in DESIGN.md and profiles/README.md is priced with what this prints (profiles/r03_issue_rate.txt).
//   hipcc --offload-arch=gfx950 -O3 tools/issue_rate.hip -o tools/bin/issue_rate && tools/bin/issue_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int CHAINS = 8, UNROLL = 16;

#define OPS(X)                                                                                   \
    X(0, "v_add_u32 (v,v)", "v_add_u32 %0, %0, %1", V1)                                          \
    X(1, "v_sub_u32 (v,s)", "v_sub_u32 %0, %0, %3", V1)                                          \
    X(2, "v_and_b32", "v_and_b32 %0, %0, %1", V1)                                                \
    X(3, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %0", V1)                                         \
    X(4, "v_min_i32", "v_min_i32 %0, %0, %1", V1)                                                \
    X(5, "v_max_u32", "v_max_u32 %0, %0, %1", V1)                                                \
    X(6, "v_cndmask_b32 (vcc)", "v_cndmask_b32 %0, %0, %1, vcc", V1)                             \
    X(7, "v_cmp_lt_i32 -> vcc", "v_cmp_lt_i32 vcc, %0, %1", V1)                                  \
    X(8, "v_cmp_lt_i32 -> sgpr pair", "v_cmp_lt_i32 s[40:41], %0, %1", V1)                       \
    X(9, "v_mov_b32", "v_mov_b32 %0, %1", V1)                                                    \
    X(10, "v_mul_i32_i24 (v,v)", "v_mul_i32_i24 %0, %0, %1", V1)                                 \
    X(11, "v_mul_u32_u24 (v,s)", "v_mul_u32_u24 %0, %3, %0", V1)                                 \
    X(12, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1", V1)                                         \
    X(13, "v_mad_i32_i24 (v,v,v)", "v_mad_i32_i24 %0, %0, %1, %2", V1)                           \
    X(14, "v_mad_i32_i24 (v,s,v)", "v_mad_i32_i24 %0, %0, %3, %2", V1)                           \
    X(15, "v_mad_u32_u24 (v,v,v)", "v_mad_u32_u24 %0, %0, %1, %2", V1)                           \
    X(16, "v_dot4_u32_u8 (v,v,v)", "v_dot4_u32_u8 %0, %0, %1, %2", V1)                           \
    X(17, "v_dot4_u32_u8 (v,s,0)", "v_dot4_u32_u8 %0, %0, %3, 0", V1)                            \
    X(18, "v_dot4_i32_i8 (v,s,v)", "v_dot4_i32_i8 %0, %0, %3, %2", V1)                           \
    X(19, "v_dot2_i32_i16 (v,v,v)", "v_dot2_i32_i16 %0, %0, %1, %2", V1)                         \
    X(20, "v_med3_i32", "v_med3_i32 %0, %0, %1, %2", V1)                                         \
    X(21, "v_min3_i32", "v_min3_i32 %0, %0, %1, %2", V1)                                         \
    X(22, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2", V1)                                         \
    X(23, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %1", V1)                                  \
    X(24, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2", V1)                                         \
    X(25, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 8", V1)                                             \
    X(26, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %2", V1)                                           \
    X(27, "v_bitop3_b32", "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96", V1)                         \
    X(28, "v_sad_u32", "v_sad_u32 %0, %0, %1, %2", V1)                                           \
    X(29, "v_add_f32", "v_add_f32 %0, %0, %1", V1)                                               \
    X(30, "v_mul_f32", "v_mul_f32 %0, %0, %1", V1)                                               \
    X(31, "v_fma_f32 (v,v,v)", "v_fma_f32 %0, %0, %1, %2", V1)                                   \
    X(32, "v_fma_f32 (v,s,v)", "v_fma_f32 %0, %0, %3, %2", V1)                                   \
    X(33, "v_fmac_f32", "v_fmac_f32 %0, %1, %2", V1)                                             \
    X(34, "v_max_f32", "v_max_f32 %0, %0, %1", V1)                                               \
    X(35, "v_med3_f32", "v_med3_f32 %0, %0, %1, %2", V1)                                         \
    X(36, "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 %0, %0", V1)                                     \
    X(37, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0", V1)                                           \
    X(38, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1", V1)                                         \
    X(39, "v_pk_mul_lo_u16", "v_pk_mul_lo_u16 %0, %0, %1", V1)                                   \
    X(40, "v_pk_mad_u16", "v_pk_mad_u16 %0, %0, %1, %2", V1)                                     \
    X(41, "v_pk_min_i16", "v_pk_min_i16 %0, %0, %1", V1)                                         \
    X(42, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %1, %1", V2)                                     \
    X(43, "v_pk_add_f32", "v_pk_add_f32 %0, %0, %1", V2)                                         \
    X(44, "v_mad_u64_u32", "v_mad_u64_u32 %0, vcc, %2, %3, %0", V2)                              \
    X(45, "v_add_u32 dpp row_shr:1", "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", V1) \
    X(46, "v_mov_b32 dpp quad_perm", "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", V1) \
    X(47, "v_cmp_eq_u32 + v_cndmask (pair)", "v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc", V1) \
    X(48, "v_add_co_u32 / v_addc (pair)", "v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %2, vcc", V1) \
    X(49, "v_mul_hi_u32", "v_mul_hi_u32 %0, %0, %1", V1)                                         \
    X(50, "v_lshrrev_b64", "v_lshrrev_b64 %0, 1, %0", V2)                                        \
    X(51, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 7", V1)                                  \
    X(52, "v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 %0, %1, %0", V1)                             \
    X(53, "v_bcnt_u32_b32", "v_bcnt_u32_b32 %0, %0, %1", V1)                                     \
    X(54, "v_rcp_f32", "v_rcp_f32 %0, %0", V1)                                                   \
    X(55, "v_sqrt_f32", "v_sqrt_f32 %0, %0", V1)                                                 \
    X(56, "v_fma_f64", "v_fma_f64 %0, %0, %1, %1", V2)                                           \
    X(57, "v_add_f64", "v_add_f64 %0, %0, %1", V2)                                               \
    X(58, "v_add_u32 (v,s)", "v_add_u32 %0, %3, %0", V1)                                         \
    X(59, "v_add_u32 (v,inline 5)", "v_add_u32 %0, 5, %0", V1)                                   \
    X(60, "v_add_u32 (v,literal)", "v_add_u32 %0, 0x12345, %0", V1)                              \
    X(61, "v_sub_u32 (v,v)", "v_sub_u32 %0, %0, %1", V1)                                         \
    X(62, "v_subrev_u32 (v,v)", "v_subrev_u32 %0, %0, %1", V1)                                   \
    X(63, "v_or_b32", "v_or_b32 %0, %0, %1", V1)                                                 \
    X(64, "v_xor_b32", "v_xor_b32 %0, %0, %1", V1)                                               \
    X(65, "v_and_b32 (v,s)", "v_and_b32 %0, %3, %0", V1)                                         \
    X(66, "v_and_b32 (v,inline)", "v_and_b32 %0, 7, %0", V1)                                     \
    X(67, "v_lshlrev_b32 (v,v)", "v_lshlrev_b32 %0, %1, %0", V1)                                 \
    X(68, "v_lshrrev_b32 (v,v)", "v_lshrrev_b32 %0, %1, %0", V1)                                 \
    X(69, "v_min_u32", "v_min_u32 %0, %0, %1", V1)                                               \
    X(70, "v_max_i32", "v_max_i32 %0, %0, %1", V1)                                               \
    X(71, "v_min_f32", "v_min_f32 %0, %0, %1", V1)                                               \
    X(72, "v_sub_f32 (v,v)", "v_sub_f32 %0, %0, %1", V1)                                         \
    X(73, "v_sub_f32 (s,v)", "v_sub_f32 %0, %3, %0", V1)                                         \
    X(74, "v_subrev_f32 (v,v)", "v_subrev_f32 %0, %0, %1", V1)                                   \
    X(75, "v_add_f32 (s,v)", "v_add_f32 %0, %3, %0", V1)                                         \
    X(76, "v_add_f32 (inline 1.0)", "v_add_f32 %0, 1.0, %0", V1)                                 \
    X(77, "v_mul_f32 (s,v)", "v_mul_f32 %0, %3, %0", V1)                                         \
    X(78, "v_fma_f32 (v,v,-v) neg", "v_fma_f32 %0, %0, %1, -%2", V1)                             \
    X(79, "v_fma_f32 (v,v,inline)", "v_fma_f32 %0, %0, %1, 1.0", V1)                             \
    X(80, "v_fmac_f32 (s,v)", "v_fmac_f32 %0, %3, %1", V1)                                       \
    X(81, "v_mov_b32 (s)", "v_mov_b32 %0, %3", V1)                                               \
    X(82, "v_mov_b32 (inline)", "v_mov_b32 %0, 7", V1)                                           \
    X(83, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2", V1)                                     \
    X(84, "v_or3_b32", "v_or3_b32 %0, %0, %1, %2", V1)                                           \
    X(85, "v_xad_u32", "v_xad_u32 %0, %0, %1, %2", V1)                                           \
    X(86, "v_add_lshl_u32", "v_add_lshl_u32 %0, %0, %1, 2", V1)                                  \
    X(87, "v_bitop3_b32 (v,s,v)", "v_bitop3_b32 %0, %0, %3, %2 bitop3:0x96", V1)                 \
    X(88, "v_cndmask_b32 (v,v,vcc const)", "v_cndmask_b32 %0, %1, %2, vcc", V0)                  \
    X(89, "v_cndmask_b32 (v,v,sgpr pair)", "v_cndmask_b32 %0, %1, %2, s[42:43]", V0)             \
    X(90, "v_cvt_u32_f32", "v_cvt_u32_f32 %0, %0", V1)                                           \
    X(91, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %0", V1)                                           \
    X(92, "v_mul_u32_u24 (v,v)", "v_mul_u32_u24 %0, %0, %1", V1)                                 \
    X(93, "v_ashrrev_i32", "v_ashrrev_i32 %0, 1, %0", V1)                                        \
    X(94, "v_add_u16", "v_add_u16 %0, %0, %1", V1)                                               \
    X(95, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1", V1)                                         \
    X(96, "v_pk_mul_f32", "v_pk_mul_f32 %0, %0, %1", V2)                                         \
    X(97, "v_max3_f32", "v_max3_f32 %0, %0, %1, %2", V1)                                         \
    X(98, "v_min3_f32", "v_min3_f32 %0, %0, %1, %2", V1)                                         \
    X(99, "v_readlane+s (v_readfirstlane)", "v_readfirstlane_b32 s44, %0", V1)                   \
    X(100, "v_fma_f32 (v,v,v) distinct dst", "v_fma_f32 %0, %1, %2, %0", V1)                     \
    X(101, "v_mul_f32 (v,v) e64 abs", "v_mul_f32 %0, |%0|, %1", V1)                              \
    X(102, "v_add_f32 dpp row_shr", "v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", V1) \
    X(103, "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1 %0, %0", V1)                                    \
    X(104, "v_sub_u32 (v,v) rev operands", "v_sub_u32 %0, %1, %0", V1)                           \
    X(105, "v_mul_i32_i24 (v, inline)", "v_mul_i32_i24 %0, 3, %0", V1)                               \
    X(106, "mix: v_min_i32 + v_add_u32 (pair)", "v_min_i32 %0, %0, %1\n v_add_u32 %0, %0, %2", V1) \
    X(107, "mix: v_min_i32 + v_min_i32 (pair)", "v_min_i32 %0, %0, %1\n v_max_i32 %0, %0, %2", V1) \
    X(108, "mix: v_add_u32 + v_add_u32 (pair)", "v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %2", V1) \
    X(109, "mix: dot4,mad24,bitop3,med3,min,sub (six)", "v_dot4_u32_u8 %0, %0, %3, 0\n v_mad_i32_i24 %0, %0, %3, %1\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xea\n v_med3_i32 %0, %0, %1, %2\n v_min_i32 %0, %0, %1\n v_sub_u32 %0, %0, %2", V1) \
    X(110, "mix: 3 fast + 1 slow (four)", "v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %2\n v_sub_u32 %0, %0, %1\n v_min_i32 %0, %0, %2", V1) \
    X(111, "v_pk_fma_f32 (v2,s2,v2)", "v_pk_fma_f32 %0, %1, s[42:43], %0", V2)                   \
    X(112, "v_pk_fma_f32 (v splat,s2,v2)", "v_pk_fma_f32 %0, %1, s[42:43], %0 op_sel_hi:[0,1,1]", V2) \
    X(113, "v_pk_mul_f32 (v2,s2)", "v_pk_mul_f32 %0, %0, s[42:43]", V2)                           \
    X(114, "mix: 3 pk_fma(s2) + 2 x (and_or, med3, min) (nine)", "v_pk_fma_f32 %1, %2, s[42:43], %1 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %2, s[42:43], %1 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %2, s[42:43], %1 op_sel_hi:[0,1,1]\n v_and_or_b32 %0, %0, %3, %4\n v_med3_u32 %0, %0, %3, %4\n v_min_u32 %0, %0, %3\n v_and_or_b32 %0, %0, %3, %4\n v_med3_u32 %0, %0, %3, %4\n v_min_u32 %0, %0, %3", V3) \
    X(115, "mix: sub(s), mul, pk_add(s2), fma, fma, and_or, med3, min (eight)", "v_sub_f32 %0, %5, %0\n v_mul_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, s[42:43]\n v_fma_f32 %0, %3, %3, %0\n v_fma_f32 %0, %4, %4, %0\n v_and_or_b32 %0, %0, %3, %4\n v_med3_u32 %0, %0, %3, %4\n v_min_u32 %0, %0, %3", V3)

template <int OP>
__global__ void rate_kernel(uint32_t* out, long long* cycles, int iters, uint32_t sarg) {
    uint32_t a[CHAINS];
    uint64_t p[CHAINS];
    const uint32_t x = threadIdx.x * 2654435761u + 12345u, y = blockIdx.x + 7u;
    const uint32_t sv = __builtin_amdgcn_readfirstlane(sarg);
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
        a[c] = x + c;
        p[c] = ((uint64_t)x << 32) | (uint32_t)(y + c);
    }
    const uint64_t px = ((uint64_t)y << 32) | x;
    asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[42:43], 0x3333" ::: "vcc", "s42", "s43");
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
#define V1(txt) asm volatile(txt : "+v"(a[c]) : "v"(x), "v"(y), "s"(sv) : "vcc", "s40", "s41", "s44");
#define V0(txt) asm volatile(txt : "+v"(a[c]) : "v"(x), "v"(y), "s"(sv));
#define V2(txt) asm volatile(txt : "+v"(p[c]) : "v"(px), "v"(x), "v"(y) : "vcc");
#define V3(txt) asm volatile(txt : "+v"(a[c]), "+v"(p[c]) : "v"(px), "v"(x), "v"(y), "s"(sv));
#define X(id, name, txt, kind) if (OP == id) { kind(txt) }
                OPS(X)
#undef X
            }
    }
    const long long t1 = clock64();
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) r ^= a[c] ^ (uint32_t)p[c] ^ (uint32_t)(p[c] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0)
        cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

static uint32_t* g_out;
static long long *g_cyc, *g_host;

static int g_first = 0;

template <int OP>
void run(const char* name, int per_step) {
    if (OP < g_first) return;
    const int iters = 3000;
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    printf("%-34s", name);
    for (int wps : {2, 4, 8}) {
        // wps waves per SIMD = 4 wps waves per CU, as ONE workgroup per CU (a second one is not
        // guaranteed to land on the same CU)
        const int threads = 256 * wps > 1024 ? 1024 : 256 * wps;
        const int grid = ncu * (256 * wps / threads);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        rate_kernel<OP><<<grid, threads>>>(g_out, g_cyc, 10, 3u);          // warm-up
        hipEventRecord(e0);
        rate_kernel<OP><<<grid, threads>>>(g_out, g_cyc, iters, 3u);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const int nw = grid * threads / 64;
        hipMemcpy(g_host, g_cyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
        double mean = 0;
        for (int i = 0; i < nw; ++i) mean += (double)g_host[i];
        mean /= nw;
        const double instr = (double)iters * UNROLL * CHAINS * per_step;   // per wave
        const int wave_per_simd = nw / (ncu * 4);
        // ns per wave-instruction on one SIMD; ticks of s_memtime per wave-instruction and SIMD
        printf("  w%d: %.3f ns %.2f tk", wps, (double)ms * 1e6 / (instr * wave_per_simd),
               mean / (instr * wave_per_simd));
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
    printf("\n");
}

int main(int argc, char** argv) {
    if (argc > 1) g_first = atoi(argv[1]);       // first row to run
    hipMalloc(&g_out, 1024 * 1024 * 4);
    hipMalloc(&g_cyc, 1024 * 16 * 8);
    g_host = (long long*)malloc(1024 * 16 * 8);
    for (int i = 0; i < 40; ++i) rate_kernel<0><<<1024, 1024>>>(g_out, g_cyc, 3000, 3u);   // clocks up
    hipDeviceSynchronize();
    printf("ns = elapsed time per wave-instruction and SIMD; tk = s_memtime ticks of a wave per wave-instruction and SIMD\n");
#define X(id, name, txt, kind) run<id>(name, (id == 47 || id == 48 || (id >= 106 && id <= 108)) ? 2 : id == 109 ? 6 : id == 110 ? 4 : id == 114 ? 9 : id == 115 ? 8 : 1);
    OPS(X)
#undef X
    return 0;
}
