// VALU issue rate of one SIMD of gfx950, measured: wave-instructions per cycle for chains of
// independent v_fma_f32 / v_dot4_u32_u8 / v_mad_u32_u24 / v_med3_u32 / v_perm_b32 / v_pk_fma_f32
// at 1, 2, 4 and 8 waves per SIMD.  Every "% of the issue slots" in DESIGN.md and
// profiles/README.md is priced with the rate this prints.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_rate.hip -o build/issue_rate && build/issue_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int CHAINS = 8, UNROLL = 16;

template <int OP>
__global__ void rate_kernel(uint32_t* out, long long* cycles, int iters) {
    uint32_t a[CHAINS];
    uint64_t p[CHAINS];
    const uint32_t x = threadIdx.x * 2654435761u + 12345u, y = blockIdx.x + 7u;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
        a[c] = x + c;
        p[c] = ((uint64_t)x << 32) | (uint32_t)(y + c);
    }
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(x), "v"(y));
                if (OP == 1) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[c]) : "v"(x), "v"(y));
                if (OP == 2) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[c]) : "v"(x), "v"(y));
                if (OP == 3) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(x), "v"(y));
                if (OP == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(x), "v"(y));
                if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[c]) : "v"(p[(c + 1) % CHAINS]));
                if (OP == 6) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p[c]) : "v"(x), "v"(y) : "vcc");
                if (OP == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(x));
                if (OP == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "v"(x));
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

template <int OP>
void run(const char* name, uint32_t* out, long long* cyc, long long* host) {
    const int iters = 2000;
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    for (int wps : {1, 2, 4, 8}) {
        // wps waves per SIMD = 4 wps waves per CU; one workgroup per CU up to 16 waves, two beyond
        const int wg_per_cu = wps > 4 ? 2 : 1, threads = 256 * wps / wg_per_cu;
        const int grid = ncu * wg_per_cu;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        rate_kernel<OP><<<grid, threads>>>(out, cyc, 10);          // warm-up
        hipEventRecord(e0);
        rate_kernel<OP><<<grid, threads>>>(out, cyc, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const int nw = grid * threads / 64;
        hipMemcpy(host, cyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
        double mean = 0;
        for (int i = 0; i < nw; ++i) mean += (double)host[i];
        mean /= nw;
        const double per_wave = (double)iters * UNROLL * CHAINS;
        // clock64 counts at a fixed 100 MHz on this chip: the rate per shader cycle comes from the
        // elapsed time and the shader clock
        int khz = 0;
        hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
        const double shader_cycles = (double)ms * 1e-3 * khz * 1e3;
        printf("%-14s waves/SIMD %d: %.3f ms, %.1f ticks/wave, wave-instr per SIMD per shader cycle (at %d MHz) %.3f\n",
               name, wps, ms, mean, khz / 1000, per_wave * wps / shader_cycles);
    }
}

int main() {
    uint32_t* out;
    long long *cyc, *host;
    hipMalloc(&out, 512 * 1024 * 4);
    hipMalloc(&cyc, 512 * 16 * 8);
    host = (long long*)malloc(512 * 16 * 8);
    run<0>("v_fma_f32", out, cyc, host);
    run<1>("v_dot4_u32_u8", out, cyc, host);
    run<2>("v_mad_u32_u24", out, cyc, host);
    run<3>("v_med3_u32", out, cyc, host);
    run<4>("v_perm_b32", out, cyc, host);
    run<5>("v_pk_fma_f32", out, cyc, host);
    run<6>("v_mad_u64_u32", out, cyc, host);
    run<7>("v_add_u32", out, cyc, host);
    run<8>("v_mul_lo_u32", out, cyc, host);
    return 0;
}
