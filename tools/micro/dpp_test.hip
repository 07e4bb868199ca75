// micro test: DPP wave_shr:1 and row_shr:n semantics on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
    int lane = threadIdx.x;
    int v = lane * 10;
    out[lane] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);          // wave_shr:1
    out[64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xf, false);     // row_shr:1
    out[128 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x114, 0xf, 0xf, false);    // row_shr:4
    out[192 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x118, 0xf, 0xf, false);    // row_shr:8
}
int main() {
    int* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); int h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) { int e = l == 0 ? -1 : (l - 1) * 10; if (h[l] != e) { ok = 0; printf("wave_shr1 lane %d got %d exp %d\n", l, h[l], e); } }
    for (int l = 0; l < 64; ++l) { int e = (l % 16) < 1 ? -1 : (l - 1) * 10; if (h[64 + l] != e) { ok = 0; printf("row_shr1 lane %d got %d exp %d\n", l, h[64 + l], e); } }
    for (int l = 0; l < 64; ++l) { int e = (l % 16) < 4 ? -1 : (l - 4) * 10; if (h[128 + l] != e) { ok = 0; printf("row_shr4 lane %d got %d exp %d\n", l, h[128 + l], e); } }
    for (int l = 0; l < 64; ++l) { int e = (l % 16) < 8 ? -1 : (l - 8) * 10; if (h[192 + l] != e) { ok = 0; printf("row_shr8 lane %d got %d exp %d\n", l, h[192 + l], e); } }
    printf(ok ? "DPP OK\n" : "DPP MISMATCH\n");
    return 0;
}
