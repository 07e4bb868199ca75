// round 6: 8 waves per SIMD (two 1024-thread workgroups per CU), 16 independent registers per stream (BODY16), and two MIXES at the ratios of the DP kernels' inner loops.
// dev tool (round 5): issue rate of the VALU forms the two DP kernels are made of, on gfx950, at 1 / 2 / 4 waves per SIMD.
// Every wave runs N x 8 INDEPENDENT instructions of one kind (eight accumulators, no dependency between neighbours); 256 workgroups = one per CU,
// 4 / 8 / 16 waves each.  Reported: wall time of the launch from HIP events -> cycles per wave-instruction and SIMD at the clock the device reports
// AND at the clock measured in the run (s_memrealtime 100 MHz against the shader cycle counter), plus the implied chip-wide wave-instructions per second.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue > profiles/r05_valu_rates.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;

#define BODY8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(s), "v"(t))
#define BODY16(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(8) INS(9) INS(10) INS(11) INS(12) INS(13) INS(14) INS(15) \
    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(s), "v"(t))
#define J_XOR(i)     "v_xor_b32 %" #i ", %" #i ", %16\n"
#define J_MAXI(i)    "v_max_i32 %" #i ", %" #i ", %16\n"
#define J_PKADD(i)   "v_pk_add_i16 %" #i ", %" #i ", %16\n"
#define J_MAXDPP(i)  "v_max_i32_dpp %" #i ", %16, %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define J_FMA(i)     "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define J_MOV(i)     "v_mov_b32 %" #i ", %16\n"
#define I_XOR(i)     "v_xor_b32 %" #i ", %" #i ", %8\n"
#define I_ADD(i)     "v_add_u32 %" #i ", %" #i ", %8\n"
#define I_MAXI(i)    "v_max_i32 %" #i ", %" #i ", %8\n"
#define I_PKADD(i)   "v_pk_add_i16 %" #i ", %" #i ", %8\n"
#define I_PKMAX(i)   "v_pk_max_i16 %" #i ", %" #i ", %8\n"
#define I_ADDDPP(i)  "v_add_u32_dpp %" #i ", %8, %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_MAXDPP(i)  "v_max_i32_dpp %" #i ", %8, %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_FMA(i)     "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_BFI(i)     "v_bfi_b32 %" #i ", %8, %9, %" #i "\n"
#define I_ANDOR(i)   "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define I_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define I_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define I_MAX3(i)    "v_max3_i32 %" #i ", %" #i ", %8, %9\n"
#define I_PKSUB(i)   "v_pk_sub_i16 %" #i ", %" #i ", %8\n"
#define I_MOV(i)     "v_mov_b32 %" #i ", %8\n"

template <int OP> __global__ __launch_bounds__(1024) void k(u64* out, int n, unsigned seed)
{
    unsigned x0 = seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 11, x5 = x0 * 13, x6 = x0 * 17, x7 = x0 * 19; unsigned s = seed | 1, t = seed * 7 + 3;
    unsigned y0 = x0 * 23, y1 = x0 * 29, y2 = x0 * 31, y3 = x0 * 37, y4 = x0 * 41, y5 = x0 * 43, y6 = x0 * 47, y7 = x0 * 53;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f}, p2 = {5.f, 6.f}, p3 = {7.f, 8.f}, p4 = {1.5f, 2.5f}, p5 = {3.5f, 4.5f}, p6 = {5.5f, 6.5f}, p7 = {7.5f, 8.5f}, ps = {1.0001f, 0.9999f}, pt = {0.5f, 0.25f};
    const u64 c0 = __builtin_readcyclecounter(); const u64 r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
        if (OP == 0) BODY8(I_XOR);
        if (OP == 1) BODY8(I_ADD);
        if (OP == 2) BODY8(I_MAXI);
        if (OP == 3) BODY8(I_PKADD);
        if (OP == 4) BODY8(I_PKMAX);
        if (OP == 5) BODY8(I_ADDDPP);
        if (OP == 6) BODY8(I_FMA);
        if (OP == 7) asm volatile("v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\nv_pk_fma_f32 %3, %3, %8, %9\nv_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\nv_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9\n"
                                  : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(ps), "v"(pt));
        if (OP == 8) BODY8(I_BFI);
        if (OP == 9) BODY8(I_ANDOR);
        if (OP == 10) BODY8(I_CNDMASK);
        if (OP == 11) BODY8(I_LSHLADD);
        if (OP == 12) BODY8(I_MAX3);
        if (OP == 13) BODY8(I_MAXDPP);
        if (OP == 14) BODY8(I_PKSUB);
        if (OP == 15) BODY8(I_MOV);
        if (OP == 16) { BODY16(J_XOR); }
        if (OP == 17) { BODY16(J_MAXI); }
        if (OP == 18) { BODY16(J_PKADD); }
        if (OP == 19) { BODY16(J_MAXDPP); }
        if (OP == 20) { BODY16(J_FMA); }
        if (OP == 21) { BODY16(J_MOV); }
        // mix A = the aligner's step loop (k_sg_align16p): packed int16 add / sub / max, a DPP hand-off, bfi / and_or for the direction words - 8 independent instructions
        if (OP == 22) asm volatile("v_pk_add_i16 %0, %0, %8\nv_pk_max_i16 %1, %1, %8\nv_pk_sub_i16 %2, %2, %8\nv_pk_max_i16 %3, %3, %9\nv_mov_b32_dpp %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\nv_bfi_b32 %5, %8, %9, %5\nv_and_or_b32 %6, %6, %8, %9\nv_pk_add_i16 %7, %7, %9\n"
                                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(s), "v"(t));
        // mix B = the POA tight row (k_poa_tile): one fused DPP add, a six-step max scan (DPP max), compares / selects for the direction code, a plain add - 8 independent instructions
        if (OP == 23) asm volatile("v_add_u32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_max_i32_dpp %1, %8, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_max_i32_dpp %2, %9, %2 row_shr:2 row_mask:0xf bank_mask:0xf\nv_max_i32 %3, %3, %8\nv_max_i32_dpp %4, %8, %4 row_shr:4 row_mask:0xf bank_mask:0xf\nv_add_u32 %5, %5, %8\nv_lshl_add_u32 %6, %6, 1, %9\nv_max3_i32 %7, %7, %8, %9\n"
                                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(s), "v"(t));
    }
    const u64 c1 = __builtin_readcyclecounter(); const u64 r1 = __builtin_amdgcn_s_memrealtime();
    unsigned r = y0 ^ y1 ^ y2 ^ y3 ^ y4 ^ y5 ^ y6 ^ y7 ^ x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ (unsigned)(p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y);
    const size_t o = ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
    out[o] = (threadIdx.x & 63) == 0 ? (c1 - c0) : r;
    if (threadIdx.x == 1) out[o] = r1 - r0;
}

static double g_nominal_ghz = 2.4;
template <int OP> void run(const char* name, int waves_per_cu, int n_cu)
{
    const int per_iter = (OP >= 16 && OP <= 21) ? 16 : 8;
    const int wg_per_cu = waves_per_cu > 16 ? 2 : 1;               // 8 waves per SIMD = 32 waves per CU = two 1024-thread workgroups per CU
    const int n = 20000, blocks = n_cu * wg_per_cu, threads = 64 * waves_per_cu / wg_per_cu; u64* d; hipMalloc(&d, sizeof(u64) * blocks * 1024);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, n, 12345u); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, n, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<u64> h((size_t)blocks * threads); hipMemcpy(h.data(), d, sizeof(u64) * blocks * threads, hipMemcpyDeviceToHost);
    const double cyc = (double)h[0], real_ticks = (double)h[1];                       // wave 0 of workgroup 0: shader cycles and 100 MHz ticks over the loop
    const double ghz_measured = real_ticks > 0 ? cyc / (real_ticks * 10.0) : 0.0;     // cycles per ns
    const double per_simd = (double)waves_per_cu / 4.0 * n * per_iter;                     // wave-instructions one SIMD issued
    const double ns_per = ms * 1e6 / per_simd;
    const double chip = (double)n_cu * 4.0 / ns_per;                                  // G wave-instructions / s over the chip
    printf("%-34s waves/SIMD %d: %7.3f ms | %.3f ns per wave-instruction and SIMD = %.2f cycles at the nominal %.2f GHz, %.2f cycles at the measured %.3f GHz | wave 0: %.2f shader cycles per own instruction | chip: %.0f G wave-instr/s\n",
           name, waves_per_cu / 4, ms, ns_per, ns_per * g_nominal_ghz, g_nominal_ghz, ns_per * ghz_measured, ghz_measured, cyc / (n * (double)per_iter), chip);
    fflush(stdout); hipFree(d);
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    g_nominal_ghz = p.clockRate / 1e6;
    printf("# device %s, %d CUs, clockRate %.3f GHz; one workgroup per CU, N = 20000 x 8 (or 16) independent instructions per wave; waves/SIMD 8 = two workgroups of 1024 threads per CU\n", p.name, p.multiProcessorCount, g_nominal_ghz);
    printf("# MI355X_MICROARCH.md:52-54 states SIMD-32 lanes x 2 passes = 2 cycles per wave64 VALU instruction (1 229 G wave-instr/s at 2.4 GHz x 1024 SIMDs); 4 cycles = 614 G\n");
    const int ncu = p.multiProcessorCount;
    for (int w : {4, 8, 16, 32}) {
        run<0>("v_xor_b32", w, ncu); run<1>("v_add_u32", w, ncu); run<2>("v_max_i32", w, ncu); run<3>("v_pk_add_i16", w, ncu); run<14>("v_pk_sub_i16", w, ncu); run<4>("v_pk_max_i16", w, ncu);
        run<5>("v_add_u32_dpp row_shr:1", w, ncu); run<13>("v_max_i32_dpp row_shr:1", w, ncu); run<6>("v_fma_f32", w, ncu); run<7>("v_pk_fma_f32", w, ncu);
        run<8>("v_bfi_b32", w, ncu); run<9>("v_and_or_b32", w, ncu); run<10>("v_cndmask_b32", w, ncu); run<11>("v_lshl_add_u32", w, ncu); run<12>("v_max3_i32", w, ncu); run<15>("v_mov_b32", w, ncu);
        run<16>("16 regs: v_xor_b32", w, ncu); run<17>("16 regs: v_max_i32", w, ncu); run<18>("16 regs: v_pk_add_i16", w, ncu); run<19>("16 regs: v_max_i32_dpp row_shr:1", w, ncu); run<20>("16 regs: v_fma_f32", w, ncu); run<21>("16 regs: v_mov_b32", w, ncu);
        run<22>("MIX A (aligner step: pk_i16 + dpp + bfi)", w, ncu); run<23>("MIX B (POA tight row: dpp add / max scan)", w, ncu);
        printf("\n");
    }
    return 0;
}
