// launchgap.hip -- what a launch of N one-wave workgroups with 10 KB of dynamic LDS costs when the kernel does nothing
// (dispatch rate + launch-to-launch gap), and when every wave just sleeps ~100 us (a stand-in for one round of decode waves).
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe/launchgap.bin tools/probe/launchgap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_k(int* p) { extern __shared__ int lds[]; if (p == (int*)1) lds[threadIdx.x] = 1; }
__global__ void sleep_k(int* p, int loops) {
    extern __shared__ int lds[];
    if (p == (int*)1) lds[threadIdx.x] = 1;
    for (int i = 0; i < loops; i++) __builtin_amdgcn_s_sleep(127);  // 127 * 64 cycles
}
int main() {
    hipFuncSetAttribute((const void*)empty_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)sleep_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int grid : {1, 4096, 16384, 65536}) {
        for (int lds : {0, 10240}) {
            for (int i = 0; i < 20; i++) hipLaunchKernelGGL(empty_k, dim3(grid), dim3(64), lds, 0, (int*)nullptr);
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (int i = 0; i < 200; i++) hipLaunchKernelGGL(empty_k, dim3(grid), dim3(64), lds, 0, (int*)nullptr);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("empty  grid %6d lds %5d : %.2f us per launch\n", grid, lds, ms * 1000 / 200);
        }
    }
    for (int grid : {4096, 16384}) {
        const int loops = 30;  // 30 * 127 * 64 cycles = 243840 cycles ~ 100 us at 2.4 GHz
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(sleep_k, dim3(grid), dim3(64), 10240, 0, (int*)nullptr, loops);
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(sleep_k, dim3(grid), dim3(64), 10240, 0, (int*)nullptr, loops);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("sleep  grid %6d lds 10240 : %.2f us per launch (each wave sleeps %d x 127 x 64 cycles)\n", grid, ms * 1000 / 50, loops);
    }
    return 0;
}
