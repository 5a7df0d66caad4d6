// Where one deconvolve() call's kernel time goes: the latency kernel (vit_lat.hip, compiled into this probe with
// VIT_LAT_STAMPS) stamps the shader clock at its phase borders; one frame, symbols in mapped host memory (as the export
// hands them over) and in device memory.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVIT_LAT_STAMPS -I include -I viterbi.dll_amd/csrc tools/probe/lat_phases.hip -o lat_phases
#include "../../viterbi.dll_amd/csrc/vit_lat.hip"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

int vit_device_cus(int) { return 256; }
hipError_t vit_optin_dynamic_lds(const void* const* kernels, int nkernels, int bytes, int, uint64_t* done) {
    if (*done) return hipSuccess;
    for (int i = 0; i < nkernels; i++) {
        hipError_t e = hipFuncSetAttribute(kernels[i], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
    }
    *done = 1;
    return hipSuccess;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const unsigned fb = argc > 1 ? atoi(argv[1]) : 768;
    const size_t nsym = 4 * (fb + 6);
    uint8_t *h = nullptr, *hd = nullptr, *d = nullptr;
    CK(hipHostMalloc((void**)&h, 65536, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void**)&hd, h, 0));
    CK(hipMalloc((void**)&d, 65536));
    srand(1);
    for (size_t i = 0; i < nsym; i++) h[i] = (uint8_t)(rand() & 1 ? 200 + rand() % 30 : 30 + rand() % 30);
    CK(hipMemcpy(d, h, nsym, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t out_off = 40960, flag_off = 49152;
    for (int where = 0; where < 2; where++) {
        uint8_t* sym = where ? d : hd;
        uint8_t* out = where ? d + out_off : hd + out_off;
        double sums[6] = {0};
        double wall = 0, clk = 0;
        const int reps = 300;
        for (int r = 0; r < reps + 20; r++) {
            volatile uint32_t* flag = (volatile uint32_t*)(h + flag_off);
            *flag = 0;
            const auto t0 = std::chrono::steady_clock::now();
            CK(vit_launch_lat(sym, false, out, nullptr, fb, fb, 1, s, (uint32_t*)(hd + flag_off), (uint32_t)(r + 1), false));
            while (*flag != (uint32_t)(r + 1)) __builtin_ia32_pause();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            CK(hipStreamSynchronize(s));
            unsigned long long st[16];
            CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_lat_stamps), sizeof st));
            if (r >= 20) {
                wall += us;
                for (int k = 0; k < 5; k++) sums[k] += (double)(st[k + 1] - st[k]);
                clk += (double)(st[5] - st[0]);
            }
        }
        const char* names[5] = {"stage symbols", "ACS (pre-pass + steps)", "traceback first pass", "traceback re-trace passes", "image + output"};
        printf("framebits %u, symbols in %s: launch -> flag seen %.1f us; kernel body %.0f shader cycles (100 MHz ticks of s_memtime? see note)\n", fb,
               where ? "device memory" : "mapped host memory", wall / reps, clk / reps);
        for (int k = 0; k < 5; k++) printf("  %-28s %9.0f ticks  %5.1f %%\n", names[k], sums[k] / reps, 100.0 * sums[k] / clk);
    }
    return 0;
}
