// Probe of gfx950 cross-lane primitive semantics used by vit_pk.hip (debug tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* o) {
    uint32_t lane = threadIdx.x;
    uint32_t x = lane, y = 100 + lane;
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    o[0 * 64 + lane] = r[0];
    o[1 * 64 + lane] = r[1];
    o[2 * 64 + lane] = __builtin_amdgcn_update_dpp(x, y, 0x128, 0xF, 0xC, false);  // row_ror:8 banks 2,3
    o[3 * 64 + lane] = __builtin_amdgcn_update_dpp(x, y, 0x114, 0xF, 0xA, false);  // row_shr:4 banks 1,3
    o[4 * 64 + lane] = __builtin_amdgcn_update_dpp(x, y, 0x104, 0xF, 0x5, false);  // row_shl:4 banks 0,2
    o[5 * 64 + lane] = __builtin_amdgcn_update_dpp(0u, y, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    o[6 * 64 + lane] = __builtin_amdgcn_update_dpp(0u, y, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    o[7 * 64 + lane] = (uint32_t)__builtin_amdgcn_ds_swizzle((int)y, 0);
    o[8 * 64 + lane] = __builtin_amdgcn_perm(0x77665544u, 0x33221100u, 0x0D040C01u);
    o[9 * 64 + lane] = __builtin_amdgcn_lerp(0x00FF10FEu, 0x01FF1101u, 0x01010101u);
    o[10 * 64 + lane] = __shfl_down(y, 1);
}
int main() {
    uint32_t* d; hipMalloc(&d, 11 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[11 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[] = {"swap.vdst(x)", "swap.src(y)", "ror8 b=C", "shr4 b=A", "shl4 b=5", "qp2301", "qp1032", "swz0", "perm", "lerp", "shfl_down"};
    for (int r = 0; r < 11; r++) { printf("%-12s:", nm[r]); for (int l = 0; l < 64; l++) printf(r>=8&&r<10? " %08x":" %3u", h[r*64+l]); printf("\n"); if (r>=8&&r<10) {} }
    return 0;
}
