// dev tool (measured round 4: a lone wave issues one VALU instruction per ~8 cycles, 64-bit shifts / adds cost the same as 32-bit ops, v_alignbit 9.5): issue cost of the 64-bit integer VALU forms the bit-vector aligner uses, on gfx950.  One workgroup of W waves per CU-sized grid; every wave runs
// N x 8 independent instructions of one kind and reports wall-clock cycles / instruction (s_memtime).     hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
#define REP8(X) X X X X X X X X
template <int OP> __global__ __launch_bounds__(1024) void k(u64* out, int n, u64 seed)
{
    u64 a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; u64 b = seed | 1; unsigned sh = (unsigned)(seed & 31) | 1;
    unsigned x0 = (unsigned)a0, x1 = (unsigned)a1, x2 = (unsigned)a2, x3 = (unsigned)a3, x4 = (unsigned)a4, x5 = (unsigned)a5, x6 = (unsigned)a6, x7 = (unsigned)a7;
    const u64 t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        if (OP == 0) { asm volatile("v_lshlrev_b64 %0, 1, %0\nv_lshlrev_b64 %1, 1, %1\nv_lshlrev_b64 %2, 1, %2\nv_lshlrev_b64 %3, 1, %3\nv_lshlrev_b64 %4, 1, %4\nv_lshlrev_b64 %5, 1, %5\nv_lshlrev_b64 %6, 1, %6\nv_lshlrev_b64 %7, 1, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        if (OP == 1) { asm volatile("v_lshl_add_u64 %0, %0, 0, %8\nv_lshl_add_u64 %1, %1, 0, %8\nv_lshl_add_u64 %2, %2, 0, %8\nv_lshl_add_u64 %3, %3, 0, %8\nv_lshl_add_u64 %4, %4, 0, %8\nv_lshl_add_u64 %5, %5, 0, %8\nv_lshl_add_u64 %6, %6, 0, %8\nv_lshl_add_u64 %7, %7, 0, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
        if (OP == 2) { asm volatile("v_lshrrev_b64 %0, %8, %0\nv_lshrrev_b64 %1, %8, %1\nv_lshrrev_b64 %2, %8, %2\nv_lshrrev_b64 %3, %8, %3\nv_lshrrev_b64 %4, %8, %4\nv_lshrrev_b64 %5, %8, %5\nv_lshrrev_b64 %6, %8, %6\nv_lshrrev_b64 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sh)); }
        if (OP == 3) { asm volatile("v_xor_b32 %0, %0, %8\nv_xor_b32 %1, %1, %8\nv_xor_b32 %2, %2, %8\nv_xor_b32 %3, %3, %8\nv_xor_b32 %4, %4, %8\nv_xor_b32 %5, %5, %8\nv_xor_b32 %6, %6, %8\nv_xor_b32 %7, %7, %8" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(sh)); }
        if (OP == 4) { asm volatile("v_alignbit_b32 %0, %0, %1, 31\nv_alignbit_b32 %1, %1, %2, 31\nv_alignbit_b32 %2, %2, %3, 31\nv_alignbit_b32 %3, %3, %4, 31\nv_alignbit_b32 %4, %4, %5, 31\nv_alignbit_b32 %5, %5, %6, 31\nv_alignbit_b32 %6, %6, %7, 31\nv_alignbit_b32 %7, %7, %0, 31" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (OP == 5) { asm volatile("v_bfi_b32 %0, %0, %8, %1\nv_bfi_b32 %1, %1, %8, %2\nv_bfi_b32 %2, %2, %8, %3\nv_bfi_b32 %3, %3, %8, %4\nv_bfi_b32 %4, %4, %8, %5\nv_bfi_b32 %5, %5, %8, %6\nv_bfi_b32 %6, %6, %8, %7\nv_bfi_b32 %7, %7, %8, %0" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(sh)); }
        if (OP == 6) { asm volatile("v_add_co_u32 %0, vcc, %0, %8\nv_addc_co_u32 %1, vcc, %1, %8, vcc\nv_add_co_u32 %2, vcc, %2, %8\nv_addc_co_u32 %3, vcc, %3, %8, vcc\nv_add_co_u32 %4, vcc, %4, %8\nv_addc_co_u32 %5, vcc, %5, %8, vcc\nv_add_co_u32 %6, vcc, %6, %8\nv_addc_co_u32 %7, vcc, %7, %8, vcc" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(sh) : "vcc"); }
        if (OP == 7) { asm volatile("v_xor_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1" : "+v"(x0) : "v"(sh)); }      // dependent chain
        if (OP == 8) { asm volatile("v_lshl_add_u64 %0, %0, 0, %1\nv_lshl_add_u64 %0, %0, 0, %1\nv_lshl_add_u64 %0, %0, 0, %1\nv_lshl_add_u64 %0, %0, 0, %1\nv_lshl_add_u64 %0, %0, 0, %1\nv_lshl_add_u64 %0, %0, 0, %1\nv_lshl_add_u64 %0, %0, 0, %1\nv_lshl_add_u64 %0, %0, 0, %1" : "+v"(a0) : "v"(b)); }   // dependent chain
    }
    const u64 t1 = __builtin_readcyclecounter();
    u64 r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (threadIdx.x & 63) == 0 ? (t1 - t0) : r;
}
template <int OP> void run(const char* name, int waves_per_cu)
{
    const int n = 20000, blocks = 256, threads = 64 * waves_per_cu; u64* d; hipMalloc(&d, sizeof(u64) * blocks * 1024);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, n, 12345ull); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, n, 12345ull);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<u64> h(blocks * 1024); hipMemcpy(h.data(), d, sizeof(u64) * blocks * threads, hipMemcpyDeviceToHost);
    // instructions per SIMD = waves per SIMD x n x 8; SIMD time = ms
    const double per_simd = (double)waves_per_cu / 4.0 * n * 8.0;
    printf("%-28s waves/CU %2d: %.2f ms  -> %.2f ns per instruction per SIMD (= %.2f cycles at 2.4 GHz); s_memtime ticks per instruction of wave 0: %.3f\n", name, waves_per_cu, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, (double)h[0] / (n * 8.0));
    fflush(stdout); hipFree(d);
}
int main()
{
    for (int w : {4, 8, 16}) {
        run<3>("v_xor_b32 (independent)", w); run<7>("v_xor_b32 (dependent)", w); run<0>("v_lshlrev_b64 x,1", w); run<2>("v_lshrrev_b64 x,v", w); run<1>("v_lshl_add_u64", w); run<8>("v_lshl_add_u64 (dependent)", w);
        run<6>("v_add_co + v_addc_co", w); run<4>("v_alignbit_b32", w); run<5>("v_bfi_b32", w);
    }
    return 0;
}
