// Dependent-chain latency of the latency kernel's trellis step, one wave alone on the chip (what a single
// deconvolve() call is bound by).  Variants isolate the partner fetch, the 16-bit clamp adds and the renormalisation hop.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u32;
typedef unsigned short u16;
typedef unsigned long long u64;
__device__ __forceinline__ u32 add_sat16(u32 a, u32 b) { return __builtin_elementwise_add_sat((u16)a, (u16)b); }
__device__ __forceinline__ u32 sub_sat16(u32 a, u32 k) { return __builtin_elementwise_sub_sat((u16)a, (u16)k); }
__device__ __forceinline__ u32 min16(u32 a, u32 b) { return __builtin_elementwise_min((u16)a, (u16)b); }
constexpr int ITER = 8192;
// V: 0 = dpp + 2 add + min; 1 = + renorm (v_cmp, s_bitcmp1, s_cselect, v_sub clamp); 2 = no dpp (partner = own);
//    3 = 0 with 32-bit add/min (clamp by a second min); 4 = 1 with the renorm through v_readfirstlane + s_cmp
template <int V>
__global__ __launch_bounds__(64) void k(u32* out, u32 x0, u32 x1) {
    u32 m = threadIdx.x + x0;
    const u32 X = x0 & 63u, Y = x1 & 63u;
    const long long t0 = wall_clock64();
    for (int i = 0; i < ITER; i++) {
        u32 p = m;
        if (V != 2) p = (u32)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, true);
        u32 om, pm, n;
        if (V == 3) {
            om = min(m + X, 255u + 0xFF00u); pm = min(p + Y, 255u + 0xFF00u); n = min(om, pm);
        } else {
            om = add_sat16(m, X); asm("" : "+v"(om)); pm = add_sat16(p, Y); asm("" : "+v"(pm)); n = min16(om, pm);
        }
        if (V == 1) {
            const u64 gt = __builtin_amdgcn_ballot_w64((u16)n > (u16)(0xFF00u + 150u));
            u32 K;
            asm("s_bitcmp1_b32 %1, 0\n\ts_cselect_b32 %0, %2, %3" : "=s"(K) : "s"((u32)gt), "s"(0xFF00u + 63u), "s"(0xFF00u) : "scc");
            m = sub_sat16(n, K) + 0xFF00u;
        } else if (V == 4) {
            const u32 m0 = (u32)__builtin_amdgcn_readfirstlane((int)n) & 0xFFFFu;
            const u32 K = m0 > 0xFF00u + 150u ? 0xFF00u + 63u : 0xFF00u;
            m = sub_sat16(n, K) + 0xFF00u;
        } else {
            m = n;
        }
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = (u32)(t1 - t0); out[1] = m; }
}
int main() {
    u32* d; CK(hipMalloc((void**)&d, 64));
    int rate = 0; CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));  // kHz
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const char* names[5] = {"dpp + 2 x v_add_u16 clamp + v_min_u16", "  + renorm: v_cmp, s_bitcmp1, s_cselect, v_sub_u16 clamp (+ v_add rebias)",
                            "no partner fetch (2 adds + min)", "32-bit add + min clamp + min (dpp)", "  + renorm through v_readfirstlane + s_cmp + s_cselect"};
    for (int rep = 0; rep < 2; rep++)
        for (int v = 0; v < 5; v++) {
            switch (v) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, 0xFF00u + 3u, 5u); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, 0xFF00u + 3u, 5u); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, 0xFF00u + 3u, 5u); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, 0xFF00u + 3u, 5u); break;
                default: hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d, 0xFF00u + 3u, 5u); break;
            }
            u32 h[2]; CK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
            if (rep) printf("%-78s %7.1f ns per step (%5.1f cycles at %.2f GHz)\n", names[v], h[0] * 1e6 / rate / ITER,
                            h[0] * 1e6 / rate / ITER * pr.clockRate / 1e6, pr.clockRate / 1e6);
        }
    return 0;
}
