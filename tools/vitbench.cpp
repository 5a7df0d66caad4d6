// vitbench.cpp -- native counterpart of the reference's viterbi-benchmark.cpp for libviterbi.so.
//
// Loads the library by name exactly like the reference harness (LoadLibrary/GetProcAddress ->
// dlopen/dlsym, viterbi-benchmark.cpp:201-229), then:
//   1. BER/FER loop over the drop-in `deconvolve` export (:293-329: random bits, DAB mother code,
//      AWGN at Eb/N0 = 3 dB, gain 32, offset 127.5, clip 0..255);
//   2. timing of `deconvolve` for the four frame sizes of :332-346 (768/1536/2304/3072 bits);
//   3. the same calls from several threads (QIRX >= 4.0 calls from several threads, README.md:56),
//      without and with the micro-batching ingest stage;
//   4. `RScheckSuperframe` on a valid block with injected errors and on a hopeless one;
//   5. the batched host entry point.
// No HIP, no Python: only the C ABI in include/viterbi_amd.h.
//   g++ -O2 -std=c++17 -I include tools/vitbench.cpp -o /tmp/vitbench -ldl -lpthread
//   /tmp/vitbench viterbi.dll_amd/libviterbi.so [frames] [loops] [sweep]
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "viterbi_amd.h"

typedef int (*DECONVOLVE)(unsigned, unsigned*, int, unsigned char*);
typedef int (*RSCHECK)(unsigned char*, int, unsigned, unsigned char*);
typedef unsigned char (*INITIALIZE)(void);
typedef int (*GETCPUCAPS)(void);
typedef int (*BATCHHOST)(const uint8_t*, uint8_t*, uint32_t, int64_t);
typedef int (*SETWINDOW)(int);
typedef const char* (*LASTERR)(void);

static uint64_t rng_state = 88172645463325252ull;
static inline uint64_t xorshift64() {
    uint64_t x = rng_state;
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    return rng_state = x;
}
static inline double uniform01() { return (double)(xorshift64() >> 11) * (1.0 / 9007199254740992.0); }
static double gauss() {
    double u1; do { u1 = uniform01(); } while (u1 <= 0.0);
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * uniform01());
}
static inline int parity8(unsigned x) { x ^= x >> 4; x ^= x >> 2; x ^= x >> 1; return x & 1; }
static const int POLYS[4] = {109, 79, 83, 109};

static void make_frame(unsigned framebits, double gain, unsigned* sym, unsigned char* bits_packed) {
    unsigned sr = 0;
    memset(bits_packed, 0, framebits / 8);
    for (unsigned i = 0; i < framebits + 6; i++) {
        const unsigned bit = i < framebits ? (unsigned)(xorshift64() >> 63) : 0u;
        if (i < framebits && bit) bits_packed[i >> 3] |= 0x80u >> (i & 7);
        sr = ((sr << 1) | bit) & 0xFF;
        for (int j = 0; j < 4; j++) {
            const int hard = parity8(sr & POLYS[j]);
            int s = (int)(127.5 + 32.0 * ((hard ? gain : -gain) + gauss()));
            sym[4 * i + j] = (unsigned)(s < 0 ? 0 : s > 255 ? 255 : s);
        }
    }
}
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "libviterbi.so";
    const int frames = argc > 2 ? atoi(argv[2]) : 500;
    const int loops = argc > 3 ? atoi(argv[3]) : 2000;
    void* h = dlopen(path, RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    auto deconvolve = (DECONVOLVE)dlsym(h, "deconvolve");
    auto rscheck = (RSCHECK)dlsym(h, "RScheckSuperframe");
    auto initialize = (INITIALIZE)dlsym(h, "initialize");
    auto getcaps = (GETCPUCAPS)dlsym(h, "GetCPUCaps");
    auto batch_host = (BATCHHOST)dlsym(h, "vit_decode_batch_host");
    auto set_window = (SETWINDOW)dlsym(h, "vit_set_batch_window_us");
    auto last_error = (LASTERR)dlsym(h, "vit_last_error");
    auto set_minc = (SETWINDOW)dlsym(h, "vit_set_batch_min_callers");
    auto set_depth = (SETWINDOW)dlsym(h, "vit_set_batch_depth");
    auto set_spin = (SETWINDOW)dlsym(h, "vit_set_batch_spin_cpus");
    const bool sweep = argc > 4 && !strcmp(argv[4], "sweep");
    if (!deconvolve || !rscheck || !initialize || !getcaps || !batch_host || !set_window || !set_minc || !set_depth || !set_spin) { fprintf(stderr, "missing export\n"); return 2; }
    initialize();
    const int caps = getcaps();
    printf("GetCPUCaps() = 0x%x (%s, %d CUs)\n", caps, (caps & VIT_CAPS_GFX950) ? "gfx950" : "no GPU", caps >> 8);
    if (!(caps & VIT_CAPS_GFX950)) { printf("no usable GPU: deconvolve returns %d (%s)\n", deconvolve(768, (unsigned*)path, 0, (unsigned char*)path), last_error()); return 1; }

    const double esn0 = 3.0 + 10.0 * std::log10(0.25), gain = 1.0 / std::sqrt(0.5 / std::pow(10.0, esn0 / 10.0));
    // 1. BER / FER at Eb/N0 = 3 dB, 3072-bit frames (viterbi-benchmark.cpp:296-329)
    {
        const unsigned fb = 3072;
        std::vector<unsigned> sym(4 * (fb + 6));
        std::vector<unsigned char> bits(fb / 8), out(fb / 8);
        long errs = 0, bad = 0;
        for (int f = 0; f < frames; f++) {
            make_frame(fb, gain, sym.data(), bits.data());
            if (deconvolve(fb, sym.data(), 0, out.data()) != 0) { fprintf(stderr, "deconvolve failed: %s\n", last_error()); return 1; }
            long e = 0;
            for (unsigned i = 0; i < fb / 8; i++) e += __builtin_popcount(out[i] ^ bits[i]);
            errs += e; bad += e != 0;
        }
        printf("BER %ld/%ld (%10.3g) FER %ld/%d (%10.3g)\n", errs, (long)fb * frames, errs / ((double)fb * frames), bad, frames, (double)bad / frames);
    }
    // 2. timing per frame size (:332-346)
    std::vector<unsigned> sym(4 * (3072 + 6));
    std::vector<unsigned char> bits(3072 / 8), out(3072 / 8);
    make_frame(3072, gain, sym.data(), bits.data());
    for (int bitrate = 32; bitrate <= 128; bitrate += 32) {
        const unsigned fb = bitrate * 24;
        for (int i = 0; i < loops / 2; i++) deconvolve(fb, sym.data(), 0, out.data());
        const double t0 = now_s();
        for (int i = 0; i < loops; i++) deconvolve(fb, sym.data(), 0, out.data());
        const double dt = now_s() - t0;
        printf("Bitrate: %5d\tFramebits: %5u\tTime: %8.4f sec  (%.1f us/call)\n", bitrate, fb, dt, dt / loops * 1e6);
    }
    // 3. concurrent callers, every thread with its own frame; every result compared with the single-call result
    {
        const int MAXT = 64;
        const unsigned fbs[4] = {768, 768, 768, 768};
        std::vector<std::vector<unsigned>> tsym(MAXT);
        std::vector<std::vector<unsigned char>> want(MAXT);
        for (int t = 0; t < MAXT; t++) {
            const unsigned fb = fbs[t & 3];
            tsym[t].resize(4 * (fb + 6));
            std::vector<unsigned char> b(fb / 8);
            make_frame(fb, gain, tsym[t].data(), b.data());
            want[t].resize(fb / 8);
            set_window(0);
            deconvolve(fb, tsym[t].data(), 0, want[t].data());
        }
        auto run = [&](int nt, int window, int minc, int depth, const char* tag) {
            set_window(window);
            const int old_minc = minc > 0 ? set_minc(minc) : -1;
            const int old_depth = depth > 0 ? set_depth(depth) : -1;
            std::vector<std::thread> th;
            std::atomic<long> bad{0};
            const double t0 = now_s();
            for (int t = 0; t < nt; t++)
                th.emplace_back([&, t] {
                    const unsigned fb = fbs[t & 3];
                    std::vector<unsigned char> o(fb / 8);
                    for (int i = 0; i < loops; i++) {
                        if (deconvolve(fb, tsym[t].data(), 0, o.data()) != 0 || memcmp(o.data(), want[t].data(), fb / 8) != 0) bad++;
                    }
                });
            for (auto& x : th) x.join();
            const double dt = now_s() - t0;
            const double cps = nt * (double)loops / dt;
            printf("threads %2d  %s: %9.0f calls/s = %6.1f Mbit/s  (%.1f us per call per thread)%s\n", nt, tag, cps, cps * 768 / 1e6,
                   dt / loops * 1e6, bad.load() ? "  *** WRONG RESULTS ***" : "");
            if (bad.load()) exit(3);
            if (old_minc > 0) set_minc(old_minc);
            if (old_depth > 0) set_depth(old_depth);
        };
        for (int nt : {1, 2, 4, 8, 16, 32, 64}) run(nt, 0, 0, 0, "ingest stage off            ");
        for (int nt : {1, 2, 4, 8, 16, 32, 64}) run(nt, 50, 0, 0, "ingest stage on (defaults)  ");
        if (sweep) {
            for (int spin : {-1, 0})  // -1: the library's default (CPU budget of the process), 0: waiting callers always sleep
                for (int depth : {2, 3, 4, 6}) {
                    const int old_spin = spin >= 0 ? set_spin(spin) : -1;
                    for (int nt : {1, 2, 4, 8, 16, 32, 64}) {
                        char tag[64];
                        snprintf(tag, sizeof tag, "window 50 depth %d spin_cpus %s", depth, spin < 0 ? "default" : "0");
                        run(nt, 50, 1, depth, tag);
                    }
                    if (old_spin >= 0) set_spin(old_spin);
                }
        }
    }
    set_window(0);
    // 4. RScheckSuperframe: all-zero block with 3 flipped bytes, then 6 errors in one column
    {
        const unsigned rs = 12;
        std::vector<unsigned char> p(120 * rs, 0), o(110 * rs, 0x77);
        p[3] = 0x55; p[500] = 0x01; p[1300] = 0xFF;
        const int r1 = rscheck(p.data(), 0, rs, o.data());
        int nz = 0; for (auto b : o) nz += b != 0;
        std::fill(p.begin(), p.end(), 0);
        const int pos[6] = {1, 9, 20, 33, 47, 90}; const unsigned char val[6] = {7, 99, 3, 200, 5, 66};
        for (int k = 0; k < 6; k++) p[5 + rs * pos[k]] = val[k];
        const int r2 = rscheck(p.data(), 0, rs, o.data());
        printf("RScheckSuperframe: 3 flipped bytes -> %d (nonzero out bytes %d), 6 errors in a column -> %d\n", r1, nz, r2);
        // per-call time of the drop-in export (the reference runs it once per 5 audio frames)
        for (unsigned rsd : {12u, 24u}) {
            std::vector<unsigned char> q(120 * rsd, 0), oo(110 * rsd, 0);
            q[7] = 0x21;
            for (int i = 0; i < 50; i++) rscheck(q.data(), 0, rsd, oo.data());
            const auto t0 = std::chrono::steady_clock::now();
            const int reps = 1000;
            for (int i = 0; i < reps; i++) rscheck(q.data(), 0, rsd, oo.data());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            printf("RScheckSuperframe RSDims %2u: %.1f us/call\n", rsd, us);
        }
    }
    // 5. batched host entry point
    {
        const unsigned fb = 768; const int64_t n = 65536;
        std::vector<uint8_t> s8((size_t)n * 4 * (fb + 6)), o8((size_t)n * fb / 8);
        for (size_t i = 0; i < (size_t)4 * (fb + 6) * 256; i++) s8[i] = (uint8_t)(xorshift64() >> 11);
        for (int64_t f = 256; f < n; f++) memcpy(&s8[(size_t)f * 4 * (fb + 6)], &s8[(size_t)(f % 256) * 4 * (fb + 6)], 4 * (fb + 6));
        batch_host(s8.data(), o8.data(), fb, n);
        const double t0 = now_s();
        for (int i = 0; i < 3; i++) batch_host(s8.data(), o8.data(), fb, n);
        const double dt = (now_s() - t0) / 3;
        printf("vit_decode_batch_host: %lld FIC frames in %.2f ms = %.1f Mbit/s (PCIe inclusive)\n", (long long)n, dt * 1e3, n * fb / dt / 1e6);
    }
    return 0;
}
