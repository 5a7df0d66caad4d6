// How many independent ACS chains must a SIMD hold to keep its VALU busy?  (design input for "8 frames per
// wavefront at 2 waves per SIMD").  Each wave runs C independent copies of the packed ACS
// step's dependent chain (4 v_pk_add_u16 clamp, 2 v_pk_min_u16, 2 v_pk_sub_i16, 2 v_bfi_b32, 1 v_sub_u32; the
// survivors feed the next step), W waves per SIMD.  Prints SIMD cycles per chain-step.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32;
__device__ __forceinline__ us2 U(u32 x) { return __builtin_bit_cast(us2, x); }
__device__ __forceinline__ u32 W_(us2 x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ u32 bfi(u32 mask, u32 a, u32 b) {
    u32 d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(mask), "v"(a), "v"(b));
    return d;
}
constexpr int STEPS = 4096;
template <int C>
__global__ __launch_bounds__(64) void chain_kernel(u32* out, u32 seed) {
    u32 A[C], B[C], acc0[C], acc1[C];
    for (int c = 0; c < C; c++) { A[c] = threadIdx.x * 2654435761u + seed + c; B[c] = A[c] ^ 0x5bd1e995u; acc0[c] = acc1[c] = 0; }
    u32 mt = (threadIdx.x & 63u) | 0x00200000u;
    for (int t = 0; t < STEPS; t++) {
#pragma unroll
        for (int c = 0; c < C; c++) {
            const us2 a = U(A[c]), b = U(B[c]), M = U(mt), MM = U(0x003F003Fu - mt);
            const us2 m0 = __builtin_elementwise_add_sat(a, M), m1 = __builtin_elementwise_add_sat(b, MM);
            const us2 m2 = __builtin_elementwise_add_sat(a, MM), m3 = __builtin_elementwise_add_sat(b, M);
            const us2 n0 = __builtin_elementwise_min(m0, m1), n1 = __builtin_elementwise_min(m2, m3);
            const us2 x01 = m0 - m1, x23 = m2 - m3;
            acc0[c] = bfi(0x80008000u, W_(x01), acc0[c] >> 1);
            acc1[c] = bfi(0x80008000u, W_(x23), acc1[c] >> 1);
            A[c] = W_(n0);
            B[c] = W_(n1);
        }
        mt ^= 0x00010001u;
    }
    u32 s = 0;
    for (int c = 0; c < C; c++) s ^= A[c] ^ B[c] ^ acc0[c] ^ acc1[c];
    if (s == 0x12345678u) out[threadIdx.x] = s;
}
template <int C>
int run(int wps, u32* d_out, int cus, double ghz) {
    const int grid = cus * 4 * wps;  // one 64-thread workgroup per wave slot
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(chain_kernel<C>, dim3(grid), dim3(64), 0, 0, d_out, 1u);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(chain_kernel<C>, dim3(grid), dim3(64), 0, 0, d_out, 1u);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double cyc = ms / 10 * 1e-3 * ghz * 1e9 / ((double)STEPS * C * wps);
    printf("waves/SIMD %d  chains/wave %d  chains/SIMD %2d : %6.1f SIMD-cycles per chain-step (13 VALU instructions)\n", wps, C, wps * C, cyc);
    return 0;
}
int main() {
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const double ghz = pr.clockRate / 1e6;
    printf("device %s, %d CUs, nominal %.2f GHz\n", pr.gcnArchName, pr.multiProcessorCount, ghz);
    u32* d_out;
    CK(hipMalloc((void**)&d_out, 4096));
    for (int wps : {1, 2, 4, 8}) {
        if (run<1>(wps, d_out, pr.multiProcessorCount, ghz)) return 1;
        if (run<2>(wps, d_out, pr.multiProcessorCount, ghz)) return 1;
        if (wps <= 4 && run<4>(wps, d_out, pr.multiProcessorCount, ghz)) return 1;
    }
    return 0;
}
