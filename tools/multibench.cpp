// multibench.cpp -- native C++ host for the multi-GPU entry point (include/viterbi_amd.h Part 3): no Python, no torch.
// Allocates a stream of FIC frames on device 0, decodes it once with vit_decode_batch_dev (one launch, one GPU) and
// once with vit_decode_stream_multi over the GPUs given on the command line, compares every output byte and times both.
//   hipcc -O2 -std=c++17 -I include tools/multibench.cpp -o tools/multibench.bin -ldl
//   tools/multibench.bin viterbi.dll_amd/libviterbi.so <nframes> <chunk_frames> <root_frames|-1> <loopback 0|1> <dev> [<dev> ...]
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "viterbi_amd.h"

typedef int (*BATCH)(const uint8_t*, uint8_t*, uint32_t, int64_t, void*);
typedef int (*MULTI)(const uint8_t*, uint8_t*, uint32_t, int64_t, const int*, int, int64_t, int64_t, unsigned, void*);
typedef const char* (*LASTERR)(void);

#define CHECK(x)                                                                    \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            return 2;                                                               \
        }                                                                           \
    } while (0)

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: %s libviterbi.so nframes chunk_frames root_frames loopback dev [dev ...]\n", argv[0]);
        return 2;
    }
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    BATCH batch = (BATCH)dlsym(h, "vit_decode_batch_dev");
    MULTI multi = (MULTI)dlsym(h, "vit_decode_stream_multi");
    LASTERR lasterr = (LASTERR)dlsym(h, "vit_last_error");
    if (!batch || !multi || !lasterr) { fprintf(stderr, "missing exports\n"); return 2; }
    const int64_t nframes = atoll(argv[2]), chunk = atoll(argv[3]), rootf = atoll(argv[4]);
    const unsigned flags = atoi(argv[5]) ? VIT_MULTI_LOOPBACK : 0u;
    std::vector<int> devs;
    for (int i = 6; i < argc; i++) devs.push_back(atoi(argv[i]));
    const uint32_t fb = 768;
    const size_t symlen = 4 * (fb + 6), olen = fb / 8;

    CHECK(hipSetDevice(devs[0]));
    std::vector<uint8_t> hsym((size_t)nframes * symlen);
    uint64_t x = 88172645463325252ull;  // xorshift64: uniform bytes - the decoder does not care what it decodes
    for (size_t i = 0; i < hsym.size(); i += 8) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        memcpy(&hsym[i], &x, hsym.size() - i < 8 ? hsym.size() - i : 8);
    }
    uint8_t *d_sym, *d_ref, *d_out;
    CHECK(hipMalloc((void**)&d_sym, hsym.size()));
    CHECK(hipMalloc((void**)&d_ref, (size_t)nframes * olen));
    CHECK(hipMalloc((void**)&d_out, (size_t)nframes * olen));
    CHECK(hipMemcpy(d_sym, hsym.data(), hsym.size(), hipMemcpyHostToDevice));
    CHECK(hipMemset(d_out, 0xEE, (size_t)nframes * olen));

    if (batch(d_sym, d_ref, fb, nframes, nullptr) != VIT_OK) { fprintf(stderr, "batch: %s\n", lasterr()); return 1; }
    CHECK(hipDeviceSynchronize());
    double t0 = now_s();
    for (int i = 0; i < 5; i++) batch(d_sym, d_ref, fb, nframes, nullptr);
    CHECK(hipDeviceSynchronize());
    const double t_one = (now_s() - t0) / 5;

    if (multi(d_sym, d_out, fb, nframes, devs.data(), (int)devs.size(), chunk, rootf, flags, nullptr) != VIT_OK) {
        fprintf(stderr, "multi: %s\n", lasterr());
        return 1;
    }
    t0 = now_s();
    for (int i = 0; i < 5; i++) multi(d_sym, d_out, fb, nframes, devs.data(), (int)devs.size(), chunk, rootf, flags, nullptr);
    const double t_multi = (now_s() - t0) / 5;  // synchronous call

    std::vector<uint8_t> a((size_t)nframes * olen), b(a.size());
    CHECK(hipMemcpy(a.data(), d_ref, a.size(), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), d_out, b.size(), hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (int64_t f = 0; f < nframes; f++) diff += memcmp(&a[f * olen], &b[f * olen], olen) != 0;
    int cur = -1;
    CHECK(hipGetDevice(&cur));
    printf("{\"frames\": %lld, \"gpus\": %zu, \"loopback\": %d, \"chunk_frames\": %lld, \"root_frames\": %lld, "
           "\"ms_one_gpu_one_launch\": %.3f, \"ms_multi\": %.3f, \"Mbit_s_multi\": %.1f, \"frames_differing\": %zu, "
           "\"current_device_restored\": %s}\n",
           (long long)nframes, devs.size(), flags ? 1 : 0, (long long)chunk, (long long)rootf, t_one * 1e3, t_multi * 1e3,
           nframes * 768.0 / t_multi / 1e6, diff, cur == devs[0] ? "true" : "false");
    return diff ? 1 : 0;
}
