// vit_api.hip -- the C ABI of libviterbi.so (include/viterbi_amd.h).
//
// Host side of the drop-in: argument validation, per-thread HIP stream and
// staging buffers, the fault ("save mode") flag, kernel selection.  No decode
// arithmetic happens on the host; when no gfx950 device is usable every entry
// point fails loudly with the documented error value.
//
// Reference boundary being replaced: viterbi.def:4-8, deconvolve.cpp:551-554,
// rschecksf.cpp:65-93, dllmain.cpp:156-160, setupdll.cpp:195-270 (dispatcher),
// exc_handler.cpp:150-249 (fault -> save mode).
#include <emmintrin.h>
#include <linux/futex.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <climits>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "vit_internal.h"

static thread_local char t_err[256] = "";
#define set_err vit_set_err

namespace {

// ---- opt-in call log, the analogue of the reference's VIT_WRITE_LOGFILE build -------------------
// (deconvolve.cpp:568-649, rschecksf.cpp:94-186): VITERBI_AMD_LOG=<path> appends one line per
// exported call: sequence number, wall-clock time, thread id, call, size, duration, return value.
struct CallLog {
    FILE* f = nullptr;
    std::mutex mu;
    std::atomic<unsigned> seq{0};
    CallLog() {
        if (const char* p = getenv("VITERBI_AMD_LOG")) f = fopen(p, "a");
    }
    void line(const char* what, unsigned size, double us, int rc) {
        if (!f) return;
        const auto now = std::chrono::system_clock::now().time_since_epoch();
        const long long usec = std::chrono::duration_cast<std::chrono::microseconds>(now).count();
        std::lock_guard<std::mutex> lk(mu);
        fprintf(f, "%6u  %lld.%06lld  TID: %zu  %s  size: %u  dur: %.1f us  ret: %d\n", seq++, usec / 1000000,
                usec % 1000000, std::hash<std::thread::id>()(std::this_thread::get_id()) % 100000, what, size, us, rc);
        fflush(f);
    }
};
CallLog* g_log = new CallLog();  // never destroyed: calls may come from other threads during exit
struct ScopedCall {
    const char* what; unsigned size; int* rc; std::chrono::steady_clock::time_point t0;
    ScopedCall(const char* w, unsigned s, int* r) : what(w), size(s), rc(r) {
        if (g_log->f) t0 = std::chrono::steady_clock::now();
    }
    ~ScopedCall() {
        if (g_log->f)
            g_log->line(what, size, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), *rc);
    }
};

// ---- process-wide state (written at init only, like deconJumpTarget) -------
std::once_flag g_once;
int g_ndev = 0;           // usable gfx950 devices
int g_device = -1;        // selected device (host-buffer entry points; *_dev calls use the caller's current device)
int g_cus = 0;
char g_init_err[160] = "no usable gfx950 (MI355X) HIP device; libviterbi has no CPU path";
std::atomic<int> g_fault{0};   // reference: exceptCounter / decon_savemode
std::atomic<int> g_kernel{0};  // 0 auto, 1 wave, 2 packed, 3 latency
// Renormalisation comparator of every decoder kernel: 1 = `>= 150`, the reference's MASM decoders (decon_avx2.asm:97,114
// `cmp sil,150 ; jb mainloop`; configuration Rel_asm - the one the reference's README tells users to build (README.md:50-52),
// i.e. what an installed viterbi.dll runs, and therefore the DEFAULT of this drop-in since round 4); 0 = `> 150`, its C
// decoders (deconvolve.cpp:399,408, configuration Rel_cpp - the one that can be compiled outside Windows and that the
// timed AVX2 port and bench.py's parity check implement).  Environment VITERBI_AMD_RENORM_GE=0/1 sets the start-up value;
// vit_set_renorm_ge() changes it.
std::atomic<int> g_renorm_ge{[] {
    const char* e = getenv("VITERBI_AMD_RENORM_GE");
    return e ? (atoi(e) != 0 ? 1 : 0) : 1;
}()};
// what an exported entry point reads ONCE per call
struct DecodeMode {
    int kernel;
    bool ge;
};
DecodeMode decode_mode() { return DecodeMode{g_kernel.load(), g_renorm_ge.load() != 0}; }

void probe_devices() {
    // Callers are threads (README.md:56), each with its own stream.  ROCclr multiplexes a process's streams onto
    // GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels that share a queue run one after the other, so
    // a host with many concurrent deconvolve() callers wants more.  That is the host's policy:
    // the library only touches the environment when VITERBI_AMD_HW_QUEUES explicitly asks it to, and only if this
    // is early enough (before the process's first HIP call) and GPU_MAX_HW_QUEUES is not set already.
    if (const char* q = getenv("VITERBI_AMD_HW_QUEUES"))
        if (atoi(q) > 0) setenv("GPU_MAX_HW_QUEUES", q, 0);
    // The first HIP call of a process can fail transiently right after another process has released the GPU (seen once on a
    // freshly acquired box: torch saw the device, this probe did not).  The probe runs once per process, so it does not give up
    // at the first answer: up to 2 s of retries, and the error text says what HIP reported.
    int n = 0;
    hipError_t pe = hipSuccess;
    for (int attempt = 0; attempt < 20; attempt++) {
        pe = hipGetDeviceCount(&n);
        if (pe == hipSuccess && n > 0) break;
        (void)hipGetLastError();
        n = 0;
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    if (n <= 0) {
        snprintf(g_init_err, sizeof g_init_err, "no usable gfx950 (MI355X) HIP device (hipGetDeviceCount: %s); libviterbi has no CPU path",
                 pe == hipSuccess ? "0 devices" : hipGetErrorString(pe));
        g_ndev = 0;
        g_device = -1;
        return;
    }
    int want = 0;
    if (const char* e = getenv("VITERBI_AMD_DEVICE")) want = atoi(e);
    int usable = 0, chosen = -1;
    for (int d = 0; d < n; d++) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, d) != hipSuccess) continue;
        if (strncmp(pr.gcnArchName, "gfx950", 6) != 0) continue;  // kernels exist for gfx950 only
        if (usable == want) { chosen = d; g_cus = pr.multiProcessorCount; }
        usable++;
    }
    g_ndev = usable;
    g_device = chosen;
    if (usable > 0 && chosen < 0)  // never fall back silently to another GPU than the one asked for
        snprintf(g_init_err, sizeof g_init_err, "VITERBI_AMD_DEVICE=%d is out of range: %d usable gfx950 device(s)", want,
                 usable);
}
void ensure_init() { std::call_once(g_once, probe_devices); }

// ---- per-thread context: stream + staging (README.md:56: callers are threads) ----
// Everything in it belongs to ONE device (`dev`).
struct ThreadCtx {
    int dev = -1;
    hipStream_t stream = nullptr;
    void* h_pin = nullptr;   size_t h_cap = 0;   // pinned host staging (mapped: h_pin_dev is its device view)
    void* h_pin_dev = nullptr;
    void* d_in = nullptr;    size_t din_cap = 0; // device input (u32 or u8 / RS block)
    void* d_sym8 = nullptr;  size_t d8_cap = 0;  // packed symbols
    void* d_out = nullptr;   size_t dout_cap = 0;
    void* d_ret = nullptr;   size_t dret_cap = 0;
    void* d_desc = nullptr;  size_t ddesc_cap = 0;  // bounds-checked descriptor copy (vit_decode_varlen_dev_checked)
    hipEvent_t scratch_ev = nullptr;  // last use of d_sym8 / d_desc by a *_dev call on a caller-owned stream
    uint32_t seq = 0;                 // completion sequence number of the single-call latency path
    bool ready = false;
    void release() {
        if (!ready) return;
        // process teardown may already have destroyed the runtime: ignore errors
        if (h_pin) (void)hipHostFree(h_pin);
        if (d_in) (void)hipFree(d_in);
        if (d_sym8) (void)hipFree(d_sym8);
        if (d_out) (void)hipFree(d_out);
        if (d_ret) (void)hipFree(d_ret);
        if (d_desc) (void)hipFree(d_desc);
        if (scratch_ev) (void)hipEventDestroy(scratch_ev);
        if (stream) (void)hipStreamDestroy(stream);
        *this = ThreadCtx();
    }
    ~ThreadCtx() { release(); }
};
// One context per (thread, device): a thread that alternates between devices (vit_decode_stream_multi drives all its
// ranks from one host thread) keeps every device's stream and buffers instead of freeing and re-creating them.
constexpr int VIT_MAX_DEVS = 64;
struct ThreadCtxs {
    ThreadCtx by_dev[VIT_MAX_DEVS];
};
thread_local ThreadCtxs t_ctxs;
thread_local ThreadCtx* t_ctx_cur = &t_ctxs.by_dev[0];  // set by ctx_prepare()
#define t_ctx (*t_ctx_cur)

#define HIPCHK(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            set_err("%s failed: %s", #call, hipGetErrorString(e_));                        \
            return VIT_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

int hip_device_ready() {
    ensure_init();
    if (g_device < 0) {
        set_err("%s", g_init_err);
        return VIT_ERR_NO_DEVICE;
    }
    return VIT_OK;
}

// Makes the calling thread's context usable on device `dev`, which must be the CURRENT device (the exported
// entry points hold a VitDeviceGuard, so the caller's own current device is restored when they return).
int ctx_prepare(int dev) {
    if (dev < 0 || dev >= VIT_MAX_DEVS) {
        set_err("HIP device ordinal %d out of range", dev);
        return VIT_ERR_ARG;
    }
    t_ctx_cur = &t_ctxs.by_dev[dev];
    if (!t_ctx.ready) {
        HIPCHK(hipStreamCreateWithFlags(&t_ctx.stream, hipStreamNonBlocking));
        t_ctx.dev = dev;
        t_ctx.ready = true;
    }
    return VIT_OK;
}
int grow_dev(void** p, size_t* cap, size_t need) {
    if (*cap >= need) return VIT_OK;
    if (*p) HIPCHK(hipFree(*p));
    *p = nullptr; *cap = 0;
    size_t sz = need < 65536 ? 65536 : need + need / 4;
    HIPCHK(hipMalloc(p, sz));
    *cap = sz;
    return VIT_OK;
}
int grow_pin(size_t need) {
    if (t_ctx.h_cap >= need) return VIT_OK;
    if (t_ctx.h_pin) HIPCHK(hipHostFree(t_ctx.h_pin));
    t_ctx.h_pin = nullptr; t_ctx.h_pin_dev = nullptr; t_ctx.h_cap = 0;
    size_t sz = need < 65536 ? 65536 : need + need / 4;
    HIPCHK(hipHostMalloc(&t_ctx.h_pin, sz, hipHostMallocMapped));
    HIPCHK(hipHostGetDevicePointer(&t_ctx.h_pin_dev, t_ctx.h_pin, 0));
    t_ctx.h_cap = sz;
    return VIT_OK;
}

bool valid_framebits(uint32_t fb) { return fb <= VIT_MAX_FRAMEBITS && (fb & 1u) == 0; }

// the analogue of setupdll.cpp:195-270's dispatcher: choose the kernel for a batch.  `choice` is the value of
// vit_set_kernel() READ ONCE by the exported entry point (a concurrent vit_set_kernel must not flip the decision
// between the check that sizes the scratch buffers and the launch).
enum { K_AUTO = 0, K_WAVE = 1, K_PACKED = 2, K_LATENCY = 3, K_PACKED8 = 4 };
// The 8-frames-per-wavefront kernel (vit_pk8.hip; round 3: 12 % fewer instructions per frame, 19 % slower than vit_pk.hip -
// two wavefronts per SIMD cannot hide the LDS round trips, profiles/r03_ab_pk8.txt) is an experiment: it is compiled in
// only with -DVIT_WITH_PK8 (viterbi.dll_amd/build.py: build(extra=["-DVIT_WITH_PK8"])).  The product library does not
// carry it: vit_set_kernel(4) then selects 0 (auto).
#ifdef VIT_WITH_PK8
constexpr int K_MAX = K_PACKED8;
bool pk8_default() {
    static const bool on = [] {
        const char* e = getenv("VITERBI_AMD_PK8");
        return e ? atoi(e) != 0 : false;
    }();
    return on;
}
#else
constexpr int K_MAX = K_LATENCY;
bool pk8_default() { return false; }
#endif
// which kernel runs a batch: the explicit choice, or for K_AUTO the latency kernel for launches that cannot fill
// the chip (<= VIT_LAT_MAX_FRAMES wavefronts) and the packed kernel otherwise
int pick_kernel(int choice, uint32_t max_framebits, int64_t nframes) {
    if (choice == K_WAVE || choice == K_LATENCY) return choice;
    if (choice == K_PACKED8) return vit_pk8_supported(max_framebits) ? K_PACKED8 : -1;
    if (choice == K_AUTO && nframes <= VIT_LAT_MAX_FRAMES) {  // small launch: does it fit the latency kernel's residency?
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && nframes <= vit_lat_capacity(max_framebits, dev)) return K_LATENCY;
    }
    return vit_pk_supported(max_framebits) ? K_PACKED : (choice == K_PACKED ? -1 : K_WAVE);
}
int launch_decode(DecodeMode mode, const uint8_t* d_sym, uint8_t* d_out, const vit_frame_desc* d_desc, uint32_t framebits,
                  uint32_t max_framebits, int64_t nframes, hipStream_t s) {
    const int choice = mode.kernel;
    int k = pick_kernel(choice, max_framebits, nframes);
    if (k < 0) {
        set_err("packed kernel does not support framebits=%u", max_framebits);
        return VIT_ERR_ARG;
    }
    if (k == K_PACKED && !d_desc && pk8_default() && vit_pk8_supported(max_framebits)) k = K_PACKED8;
    hipError_t e = k == K_PACKED8   ? vit_launch_pk8(d_sym, false, d_out, d_desc, framebits, max_framebits, nframes, s, mode.ge)
                   : k == K_PACKED  ? vit_launch_pk(d_sym, false, d_out, d_desc, framebits, max_framebits, nframes, s, mode.ge)
                   : k == K_LATENCY ? vit_launch_lat(d_sym, false, d_out, d_desc, framebits, max_framebits, nframes, s, nullptr, 0, mode.ge)
                                    : vit_launch_wave(d_sym, d_out, d_desc, framebits, max_framebits, nframes, s, mode.ge);
    if (e != hipSuccess) {
        set_err("kernel launch failed: %s", hipGetErrorString(e));
        return VIT_ERR_HIP;
    }
    return VIT_OK;
}
// Symbols still in the reference ABI's u32 format (deconvolve.cpp:158-165).  The packed and the latency kernel
// read them directly (narrowing fused into their symbol loads); otherwise they are narrowed into `d_scratch8` first.
bool u32_in_place(int choice, const void* d_sym32, uint32_t max_framebits, int64_t nframes) {
    const int k = pick_kernel(choice, max_framebits, nframes);
    return (k == K_PACKED || k == K_PACKED8 || k == K_LATENCY) && (reinterpret_cast<uintptr_t>(d_sym32) & 15u) == 0;
}
int launch_decode_u32(DecodeMode mode, const uint32_t* d_sym32, uint8_t* d_scratch8, uint8_t* d_out, const vit_frame_desc* d_desc,
                      uint32_t framebits, uint32_t max_framebits, int64_t nframes, int64_t nsym, hipStream_t s) {
    const int choice = mode.kernel;
    if (u32_in_place(choice, d_sym32, max_framebits, nframes)) {
        int k = pick_kernel(choice, max_framebits, nframes);
        if (k == K_PACKED && !d_desc && pk8_default() && vit_pk8_supported(max_framebits)) k = K_PACKED8;
        hipError_t e = k == K_PACKED8 ? vit_launch_pk8(d_sym32, true, d_out, d_desc, framebits, max_framebits, nframes, s, mode.ge)
                       : k == K_PACKED ? vit_launch_pk(d_sym32, true, d_out, d_desc, framebits, max_framebits, nframes, s, mode.ge)
                                     : vit_launch_lat(d_sym32, true, d_out, d_desc, framebits, max_framebits, nframes, s, nullptr, 0, mode.ge);
        if (e != hipSuccess) {
            set_err("kernel launch failed: %s", hipGetErrorString(e));
            return VIT_ERR_HIP;
        }
        return VIT_OK;
    }
    if (!d_scratch8) {
        set_err("internal: no scratch buffer for the u32 narrowing");
        return VIT_ERR_ARG;
    }
    hipError_t e = vit_launch_pack(d_sym32, d_scratch8, nsym, s);
    if (e != hipSuccess) { set_err("pack launch failed: %s", hipGetErrorString(e)); return VIT_ERR_HIP; }
    return launch_decode(mode, d_scratch8, d_out, d_desc, framebits, max_framebits, nframes, s);
}

// ---- ingest stage: concurrent deconvolve() callers share launches ------------------
// The reference is re-entrant and QIRX calls it from several threads (README.md:56).  One call is one 774-step
// dependent chain on one wavefront (~35 us) however empty the GPU is, and a process's streams share a handful of
// hardware queues, so separate launches of concurrent callers queue up behind each other.  The stage:
//
//   * ONE mapped pinned buffer cut into slots.  A caller claims a slot, copies its OWN symbols into it - narrowing
//     the reference ABI's u32 to the device format's one byte per symbol on the way (low byte, deconvolve.cpp:158-165:
//     a quarter of the PCIe traffic) - and spins on the slot's completion word.  Callers copy in parallel; there is
//     no worker thread, no H2D/D2H copy, no descriptor table in memory, no condition variable.
//   * The first caller to arrive while no batch is open becomes that batch's LEADER: it holds the batch open until
//     one of `depth` launch credits is free (i.e. while `depth` earlier batches are still on the GPU, at most
//     `window` microseconds), closes it, and issues ONE bounded launch of the latency kernel with grid = members
//     (vit_lat_ring_kernel: the slot table travels by value in the kernel arguments).  Batches therefore form
//     exactly when launches would otherwise queue: a lone caller finds a credit and launches at once.
//   * Every workgroup publishes the batch's sequence number in its slot's completion word after its last output
//     byte (system-scope release); the caller copies its bytes out and frees the slot.
struct SpinLock {  // test-and-test-and-set; held for a few dozen instructions (join / close a batch)
    std::atomic<int> f{0};
    void lock() {
        for (unsigned n = 1;; n++) {
            if (f.load(std::memory_order_relaxed) == 0 && f.exchange(1, std::memory_order_acquire) == 0) return;
            __builtin_ia32_pause();
            if ((n & 1023u) == 0) sched_yield();  // the holder may have lost its core
        }
    }
    void unlock() { f.store(0, std::memory_order_release); }
};

// u32 -> u8 (low byte) while copying a caller's symbols into pinned memory; n is a multiple of 8
void narrow_symbols(const unsigned int* src, uint8_t* dst, size_t n) {
    size_t i = 0;
    const __m128i lo = _mm_set1_epi32(0xFF);
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_and_si128(_mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i)), lo);
        const __m128i b = _mm_and_si128(_mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 4)), lo);
        const __m128i c = _mm_and_si128(_mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 8)), lo);
        const __m128i d = _mm_and_si128(_mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 12)), lo);
        // values are 0..255: the signed/unsigned saturating packs are exact
        _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i), _mm_packus_epi16(_mm_packs_epi32(a, b), _mm_packs_epi32(c, d)));
    }
    for (; i < n; i++) dst[i] = (uint8_t)src[i];
}

constexpr uint32_t RING_SLOTS = VIT_RING_MAXB;   // concurrent callers the ring serves; further ones take the direct path
constexpr uint32_t RING_SYM_CAP = 36992;         // >= 4 * (9216 + 6) one-byte symbols
constexpr uint32_t RING_OUT_OFF = RING_SYM_CAP;  // 1152 output bytes
constexpr uint32_t RING_FLAG_OFF = RING_OUT_OFF + 1152;  // completion word, on a cache line of its own
constexpr uint32_t RING_STRIDE = 38400;
static_assert(RING_SYM_CAP >= 4 * (VIT_MAX_FRAMEBITS + VIT_TAIL) && RING_FLAG_OFF % 64 == 0 && RING_FLAG_OFF + 64 <= RING_STRIDE &&
              RING_STRIDE % 128 == 0, "ring slot layout");
enum { RING_OK = 0, RING_FAILED = 1, RING_DECLINED = 2 };
constexpr int RING_MAX_DEPTH = 16;

long futex_wait(std::atomic<uint32_t>* a, uint32_t expected, long timeout_ns) {
    timespec ts{0, timeout_ns};
    return syscall(SYS_futex, reinterpret_cast<uint32_t*>(a), FUTEX_WAIT_PRIVATE, expected, &ts, nullptr, 0);
}
void futex_wake_all(std::atomic<uint32_t>* a) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(a), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}
// CPUs this process may keep busy: the affinity mask, capped by a cgroup CPU quota (a container that may use 16 CPUs'
// worth of time on a 256-CPU host is THROTTLED when 32 threads spin).  Waiting callers spin only while the calls in
// flight fit this budget; beyond it all but one per batch sleep on a futex.
int detect_cpu_budget() {
    cpu_set_t set;
    int n = sched_getaffinity(0, sizeof set, &set) == 0 ? CPU_COUNT(&set) : 1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
        char q[32] = "";
        long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long cpus = (atol(q) + period - 1) / period;
            if (cpus > 0 && cpus < n) n = (int)cpus;
        }
        fclose(f);
    } else {
        long quota = -1, period = 0;  // cgroup v1
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%ld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%ld", &period) != 1) period = 0; fclose(g); }
        if (quota > 0 && period > 0 && (quota + period - 1) / period < n) n = (int)((quota + period - 1) / period);
    }
    return n < 1 ? 1 : n;
}

// VITERBI_AMD_RING_STATS=1: where a call's time goes, printed to stderr (and reset) by every vit_set_batch_window_us() call
struct RingStats {
    const bool on = getenv("VITERBI_AMD_RING_STATS") != nullptr;
    std::atomic<uint64_t> calls{0}, batches{0}, declined{0}, ns_call{0}, ns_copy{0}, ns_token{0}, ns_members{0}, ns_launch{0},
        ns_leader_wait{0}, ns_follower_wait{0}, ns_wake_delay{0}, wakes{0}, slept{0}, takeovers{0}, no_token{0};
    static uint64_t now() {
        return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    void print_and_reset() {
        const uint64_t c = calls.exchange(0), b = batches.exchange(0);
        if (c && b)
            fprintf(stderr,
                    "[ring] calls %llu batches %llu (%.1f per batch) declined %llu | per call: total %.1f us copy %.1f | per batch: token wait %.1f "
                    "members' copies %.1f launch call %.1f leader wait %.1f | followers: wait %.1f us, slept %llu, wake delay %.1f us (%llu), "
                    "poller take-overs %llu, launches without token %llu\n",
                    (unsigned long long)c, (unsigned long long)b, (double)c / b, (unsigned long long)declined.load(), ns_call.load() / 1e3 / c,
                    ns_copy.load() / 1e3 / c, ns_token.load() / 1e3 / b, ns_members.load() / 1e3 / b, ns_launch.load() / 1e3 / b,
                    ns_leader_wait.load() / 1e3 / b, c > b ? ns_follower_wait.load() / 1e3 / (c - b) : 0.0, (unsigned long long)slept.load(),
                    wakes.load() ? ns_wake_delay.load() / 1e3 / wakes.load() : 0.0, (unsigned long long)wakes.load(),
                    (unsigned long long)takeovers.load(), (unsigned long long)no_token.load());
        for (auto* a : {&declined, &ns_call, &ns_copy, &ns_token, &ns_members, &ns_launch, &ns_leader_wait, &ns_follower_wait, &ns_wake_delay,
                        &wakes, &slept, &takeovers, &no_token})
            a->store(0);
    }
};
RingStats g_rstat;

struct RingBatch {
    VitRingTable tbl;                  // members (what the kernel receives by value)
    uint32_t maxfb = 0;
    bool ge = false;
    int token = -1;                    // launch token (= index of the ring stream) the leader holds, -1: none
    hipStream_t stream = nullptr;      // the stream the batch was launched on
    std::atomic<uint32_t> copied{0};   // members whose symbols are in their slot
    std::atomic<int> state{0};         // 0 not launched yet, 1 launched, 2 failed before the launch, 3 failed after it
    std::atomic<uint32_t> refs{0};     // members that have not left yet
    // waiting: ONE member at a time (the leader first) is the batch's poller: it spins on the completion words and
    // wakes the members that sleep; the others spin on their own word while CPUs are to spare, or sleep
    std::atomic<int> poller{0};                        // 1 while a member polls for the others
    std::atomic<uint32_t> wake[VIT_RING_MAXB];         // per member futex word: 0 = must wait, WAKE_DONE, WAKE_POLL
    std::atomic<uint8_t> asleep[VIT_RING_MAXB];        // member sleeps (or is about to) on its futex word
    std::atomic<uint8_t> gone[VIT_RING_MAXB];          // member has seen its own completion (its slot may be reused)
    std::atomic<uint64_t> t_seen[VIT_RING_MAXB];       // statistics only: when the poller saw the member's completion
};
enum : uint32_t { WAKE_DONE = 1, WAKE_POLL = 2 };
struct Ring {
    SpinLock mu;                       // joining and closing the open batch; everything else is lock-free
    std::once_flag once;
    int init_rc = VIT_ERR_HIP;
    uint8_t* h_base = nullptr;
    uint8_t* d_base = nullptr;
    // free slots / batch records: bit masks; claimed under `mu` (one claimer at a time), released with one atomic OR
    std::atomic<uint64_t> slot_free[2] = {};
    std::atomic<uint64_t> batch_free[2] = {};
    RingBatch batches[RING_SLOTS];     // every live batch has a member holding a slot: never more than RING_SLOTS
    RingBatch* open = nullptr;         // the batch a new caller joins (under mu)
    uint32_t seq = 0;
    // launch tokens: token i = the right to have one batch in flight on streams[i].  Batches in flight therefore never
    // share a stream (streams of arbitrary caller threads would: a process's streams are multiplexed onto a few
    // hardware queues, and two batches on one queue run one after the other).
    hipStream_t streams[RING_MAX_DEPTH] = {};
    std::atomic<uint32_t> token_free{(1u << RING_MAX_DEPTH) - 1u};
    std::atomic<int> depth{4};         // batches in flight
    std::atomic<int> window_us{50};    // 0 = stage off
    std::atomic<int> min_callers{1};   // engage only while at least this many deconvolve() calls are in flight
    std::atomic<int> inflight{0};      // deconvolve() calls currently executing (any path)
    std::atomic<int> spin_cpus{0};     // waiting callers spin while the calls in flight do not exceed this

    Ring() {
        // a host that only binds the five reference exports configures the stage through the environment
        // (the analogue of the reference's viterbi.txt)
        if (const char* e = getenv("VITERBI_AMD_BATCH_WINDOW_US")) {
            const int w = atoi(e);
            window_us.store(w < 0 ? 0 : w > 100000 ? 100000 : w);
        }
        if (const char* e = getenv("VITERBI_AMD_BATCH_MIN_CALLERS")) {
            const int n = atoi(e);
            min_callers.store(n < 1 ? 1 : n);
        }
        if (const char* e = getenv("VITERBI_AMD_BATCH_DEPTH")) set_depth(atoi(e));
        // waiting callers spin only while the calls in flight use at most a quarter of the process's CPU budget
        // (profiles/r04_vitbench_sweep.txt: spinning up to the whole budget throttles a quota-limited container)
        const char* e = getenv("VITERBI_AMD_SPIN_CPUS");
        spin_cpus.store(e ? (atoi(e) < 0 ? 0 : atoi(e)) : detect_cpu_budget() / 4);
    }
    int set_depth(int n) { return depth.exchange(n < 1 ? 1 : n > RING_MAX_DEPTH ? RING_MAX_DEPTH : n); }
    int take_token() {  // -1: `depth` batches are in flight
        uint32_t f = token_free.load(std::memory_order_relaxed);
        for (;;) {
            if (RING_MAX_DEPTH - __builtin_popcount(f) >= depth.load(std::memory_order_relaxed) || f == 0) return -1;
            const int t = __builtin_ctz(f);
            if (token_free.compare_exchange_weak(f, f & ~(1u << t), std::memory_order_acquire)) return t;
        }
    }
    static int claim_bit(std::atomic<uint64_t> (&m)[2]) {  // under mu; -1: none
        for (int w = 0; w < 2; w++) {
            const uint64_t v = m[w].load(std::memory_order_acquire);
            if (v) {
                const int bit = __builtin_ctzll(v);
                m[w].fetch_and(~(1ull << bit), std::memory_order_acquire);
                return w * 64 + bit;
            }
        }
        return -1;
    }
    static void free_bit(std::atomic<uint64_t> (&m)[2], uint32_t i) { m[i >> 6].fetch_or(1ull << (i & 63u), std::memory_order_release); }
    void allocate() {
        VitDeviceGuard guard(g_device);
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, (size_t)RING_SLOTS * RING_STRIDE, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (h) (void)hipHostFree(h);
            return;
        }
        for (int i = 0; i < RING_MAX_DEPTH; i++)
            if (hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking) != hipSuccess) {
                (void)hipGetLastError();
                return;
            }
        h_base = (uint8_t*)h;
        d_base = (uint8_t*)d;
        for (uint32_t i = 0; i < RING_SLOTS; i++) {
            batches[i].tbl.stride = RING_STRIDE;
            batches[i].tbl.out_off = RING_OUT_OFF;
            batches[i].tbl.flag_off = RING_FLAG_OFF;
        }
        static_assert(RING_SLOTS == 128, "two 64-bit masks");
        slot_free[0] = slot_free[1] = batch_free[0] = batch_free[1] = ~0ull;
        init_rc = VIT_OK;
    }
    bool engaged(const DecodeMode& mode) {
        if (window_us.load(std::memory_order_relaxed) <= 0) return false;
        if (mode.kernel != K_AUTO && mode.kernel != K_LATENCY) return false;  // a forced kernel keeps the direct path
        return inflight.load(std::memory_order_relaxed) >= min_callers.load(std::memory_order_relaxed);
    }
    uint32_t* flag_of(uint32_t slot) { return reinterpret_cast<uint32_t*>(h_base + (size_t)slot * RING_STRIDE + RING_FLAG_OFF); }
    int call(const DecodeMode& mode, uint32_t framebits, const unsigned int* symbols, unsigned char* out);
    int wait_done(RingBatch* b, uint32_t idx, uint32_t* flag, uint32_t my_seq, bool leader);
};
Ring* g_ring = new Ring();  // intentionally never destroyed (calls may come from other threads during exit)

// Waits until this member's completion word carries the batch's sequence number.  Returns RING_OK / RING_FAILED.
int Ring::wait_done(RingBatch* b, uint32_t idx, uint32_t* flag, uint32_t my_seq, bool leader) {
    auto mine_done = [&] { return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == my_seq; };
    const auto t0 = std::chrono::steady_clock::now();
    bool synced = false;
    // Every few hundred iterations: has the launch failed, is it overdue?  (true = give up)
    auto overdue = [&]() -> bool {
        if (b->state.load(std::memory_order_acquire) >= 2) return !mine_done();
        const auto waited = std::chrono::steady_clock::now() - t0;
        if (leader && !synced && waited > std::chrono::milliseconds(20)) {
            // not what a healthy launch does: fall back to the stream's own completion, once
            const hipError_t e = b->stream ? hipStreamSynchronize(b->stream) : hipErrorUnknown;
            synced = true;
            if (e != hipSuccess || !mine_done()) {
                set_err("deconvolve: the batch launch did not complete (%s)", hipGetErrorString(e));
                b->state.store(3, std::memory_order_release);
                return true;
            }
        } else if (waited > std::chrono::seconds(10)) {
            set_err("deconvolve: no completion from the batch launch");
            b->state.store(3, std::memory_order_release);
            return true;
        }
        return false;
    };
    auto wake_member = [&](uint32_t i, uint32_t what) {
        b->wake[i].store(what, std::memory_order_seq_cst);
        if (b->asleep[i].load(std::memory_order_seq_cst))
            (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(&b->wake[i]), FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0);
    };
    // This member leaves (or failed): if others still wait asleep, ONE of them must poll from here on.
    auto pass_on = [&] {
        const uint32_t n = b->tbl.n;
        for (uint32_t i = 0; i < n; i++) {
            if (i == idx || b->gone[i].load(std::memory_order_relaxed)) continue;
            if (__atomic_load_n(flag_of(b->tbl.slot[i]), __ATOMIC_RELAXED) == my_seq) { wake_member(i, WAKE_DONE); continue; }
            if (b->asleep[i].load(std::memory_order_seq_cst)) { wake_member(i, WAKE_POLL); return; }
        }
    };
    bool polling = leader;  // the leader has taken the poller's role before the launch
    bool seen[VIT_RING_MAXB];
    for (;;) {
        if (polling) {
            // ---- poller: watch every member's word, wake the sleeper whose frame is done ----
            const uint32_t n = b->tbl.n;  // final: the batch was closed before its launch, and only a launched batch is polled
            for (uint32_t i = 0; i < n; i++) seen[i] = i == idx;
            uint32_t left = n - 1;
            for (unsigned it = 1;; it++) {
                for (uint32_t i = 0; i < n && left; i++) {
                    if (seen[i]) continue;
                    if (b->gone[i].load(std::memory_order_relaxed)) { seen[i] = true; left--; continue; }
                    if (__atomic_load_n(flag_of(b->tbl.slot[i]), __ATOMIC_RELAXED) == my_seq) {
                        seen[i] = true;
                        left--;
                        if (g_rstat.on) b->t_seen[i].store(RingStats::now(), std::memory_order_relaxed);
                        wake_member(i, WAKE_DONE);
                    }
                }
                const bool failed = (it & 255u) == 0 && overdue();
                if (mine_done() || failed) {
                    // hand the role to a member that still waits asleep (a spinning one needs nobody)
                    b->poller.store(0, std::memory_order_seq_cst);
                    if (failed) {
                        for (uint32_t i = 0; i < n; i++)
                            if (!seen[i]) wake_member(i, WAKE_DONE);  // they find the batch's state themselves
                    } else if (left) {
                        pass_on();
                    }
                    return failed ? RING_FAILED : RING_OK;
                }
                __builtin_ia32_pause();
            }
        }
        // ---- not the poller: spin on the own word while CPUs are to spare, else sleep until the poller wakes us ----
        if (inflight.load(std::memory_order_relaxed) <= spin_cpus.load(std::memory_order_relaxed)) {
            for (unsigned it = 1; !mine_done(); it++) {
                __builtin_ia32_pause();
                if ((it & 255u) == 0) {
                    if (overdue()) return RING_FAILED;
                    if (inflight.load(std::memory_order_relaxed) > spin_cpus.load(std::memory_order_relaxed)) break;  // crowded now
                }
            }
            if (mine_done()) return RING_OK;
        }
        b->asleep[idx].store(1, std::memory_order_seq_cst);
        const uint32_t w = b->wake[idx].load(std::memory_order_seq_cst);
        if (mine_done()) {
            if (w == WAKE_POLL) pass_on();  // asked to poll, but already done
            return RING_OK;
        }
        int expect = 0;
        if (w == WAKE_POLL ||
            (b->state.load(std::memory_order_acquire) == 1 && b->poller.load(std::memory_order_seq_cst) == 0 &&
             b->poller.compare_exchange_strong(expect, 1, std::memory_order_seq_cst))) {
            if (w == WAKE_POLL) b->poller.store(1, std::memory_order_seq_cst);
            b->wake[idx].store(0, std::memory_order_relaxed);
            b->asleep[idx].store(0, std::memory_order_seq_cst);
            polling = true;  // the poller has left with its own frame done: this member polls from here on
            if (g_rstat.on) g_rstat.takeovers++;
            continue;
        }
        if (w == 0) {
            if (g_rstat.on) g_rstat.slept++;
            futex_wait(&b->wake[idx], 0, 2000000);  // 2 ms: a missed wake-up costs time, never the result
        }
        if (mine_done()) {
            if (g_rstat.on) {
                const uint64_t ts = b->t_seen[idx].load(std::memory_order_relaxed);
                if (ts) { g_rstat.ns_wake_delay += RingStats::now() - ts; g_rstat.wakes++; }
            }
            if (b->wake[idx].load(std::memory_order_seq_cst) == WAKE_POLL) pass_on();
            return RING_OK;
        }
        if (overdue()) return RING_FAILED;
    }
}

int Ring::call(const DecodeMode& mode, uint32_t framebits, const unsigned int* symbols, unsigned char* out) {
    std::call_once(once, [this] { allocate(); });
    if (init_rc != VIT_OK) return RING_DECLINED;
    const bool st = g_rstat.on;
    const uint64_t t_in = st ? RingStats::now() : 0;
    // ---- join the open batch, or open one ----
    mu.lock();
    RingBatch* b = open;
    const int slot_i = (b && b->ge != mode.ge) ? -1 : claim_bit(slot_free);
    if (slot_i < 0) {
        mu.unlock();
        if (st) g_rstat.declined++;
        return RING_DECLINED;
    }
    const uint32_t slot = (uint32_t)slot_i;
    const bool leader = b == nullptr;
    if (leader) {
        b = &batches[claim_bit(batch_free)];  // never fails: a live batch holds a slot, and there are as many records as slots
        b->tbl.n = 0;
        if (++seq == 0) ++seq;
        b->tbl.seq = seq;
        b->maxfb = 0;
        b->ge = mode.ge;
        b->token = -1;
        b->stream = nullptr;
        b->copied.store(0, std::memory_order_relaxed);
        b->state.store(0, std::memory_order_relaxed);
        b->refs.store(0, std::memory_order_relaxed);
        b->poller.store(1, std::memory_order_relaxed);  // the leader
        open = b;
    }
    const uint32_t idx = b->tbl.n++;
    b->tbl.slot[idx] = (uint16_t)slot;
    b->tbl.fb[idx] = (uint16_t)framebits;
    b->gone[idx].store(0, std::memory_order_relaxed);
    b->asleep[idx].store(0, std::memory_order_relaxed);
    b->wake[idx].store(0, std::memory_order_relaxed);
    b->t_seen[idx].store(0, std::memory_order_relaxed);
    if (framebits > b->maxfb) b->maxfb = framebits;
    b->refs.fetch_add(1, std::memory_order_relaxed);
    if (b->tbl.n == VIT_RING_MAXB) open = nullptr;
    const uint32_t my_seq = b->tbl.seq;
    mu.unlock();

    // ---- the caller's own copy, in parallel with everybody else's ----
    uint8_t* hs = h_base + (size_t)slot * RING_STRIDE;
    uint32_t* flag = flag_of(slot);
    __atomic_store_n(flag, 0u, __ATOMIC_RELAXED);
    narrow_symbols(symbols, hs, 4u * ((size_t)framebits + VIT_TAIL));
    b->copied.fetch_add(1, std::memory_order_release);
    const uint64_t t_copied = st ? RingStats::now() : 0;
    uint64_t t_wait0 = t_copied;

    if (leader) {
        // hold the batch open while `depth` launches are in flight (bounded by the window), then close it
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(window_us.load(std::memory_order_relaxed));
        int token = -1;
        for (unsigned spins = 1; (token = take_token()) < 0; spins++) {
            __builtin_ia32_pause();
            if ((spins & 31u) == 0 && std::chrono::steady_clock::now() > deadline) break;
        }
        const uint64_t t_tok = st ? RingStats::now() : 0;
        mu.lock();
        if (open == b) open = nullptr;
        const uint32_t n = b->tbl.n;
        mu.unlock();
        for (unsigned spins = 1; b->copied.load(std::memory_order_acquire) < n; spins++) {  // members still copying
            __builtin_ia32_pause();
            if ((spins & 1023u) == 0) sched_yield();
        }
        const uint64_t t_mem = st ? RingStats::now() : 0;
        hipError_t e = hipErrorUnknown;
        {
            VitDeviceGuard guard(g_device);
            hipStream_t s = nullptr;
            if (token >= 0) s = streams[token];
            else if (ctx_prepare(g_device) == VIT_OK) s = t_ctx.stream;  // the window ran out: one launch beyond `depth`, on this thread's stream
            if (s) {
                b->token = token;
                b->stream = s;
                e = vit_launch_lat_ring(d_base, b->tbl, b->maxfb, s, b->ge);
                if (e != hipSuccess) set_err("deconvolve: batch launch: %s", hipGetErrorString(e));
            }
        }
        b->state.store(e == hipSuccess ? 1 : 2, std::memory_order_release);
        if (st) {
            t_wait0 = RingStats::now();
            g_rstat.batches++;
            g_rstat.ns_token += t_tok - t_copied;
            g_rstat.ns_members += t_mem - t_tok;
            g_rstat.ns_launch += t_wait0 - t_mem;
            if (token < 0) g_rstat.no_token++;
        }
        if (e != hipSuccess)  // nobody will ever be woken by a completion word
            for (uint32_t i = 0; i < n; i++)
                if (i != idx) {
                    b->wake[i].store(WAKE_DONE, std::memory_order_seq_cst);
                    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(&b->wake[i]), FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0);
                }
    }

    const int rc = wait_done(b, idx, flag, my_seq, leader);
    if (rc == RING_OK) memcpy(out, hs + RING_OUT_OFF, (framebits + 7u) >> 3);
    else if (!leader) set_err("deconvolve: the shared launch failed");
    if (st) {
        const uint64_t t_out = RingStats::now();
        g_rstat.calls++;
        g_rstat.ns_call += t_out - t_in;
        g_rstat.ns_copy += t_copied - t_in;
        (leader ? g_rstat.ns_leader_wait : g_rstat.ns_follower_wait) += t_out - t_wait0;
    }
    // ---- leave: no lock ----
    // the leader's frame is done: its launch has at least started to retire, the next batch may go
    if (leader && b->token >= 0) token_free.fetch_or(1u << b->token, std::memory_order_release);
    const int final_state = b->state.load(std::memory_order_acquire);
    b->gone[idx].store(1, std::memory_order_release);
    // a slot whose kernel may still be running (failure after the launch) is never handed out again
    if (rc == RING_OK || final_state == 2) free_bit(slot_free, slot);
    if (b->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) free_bit(batch_free, (uint32_t)(b - batches));
    return rc;
}
struct InflightGuard {
    std::atomic<int>& n;
    explicit InflightGuard(std::atomic<int>& c) : n(c) { n.fetch_add(1, std::memory_order_relaxed); }
    ~InflightGuard() { n.fetch_sub(1, std::memory_order_relaxed); }
};

}  // namespace

#ifndef VIT_WITH_PK8  // the experiment is not compiled in: the launcher's names exist, the kernel does not
bool vit_pk8_supported(uint32_t) { return false; }
hipError_t vit_launch_pk8(const void*, bool, uint8_t*, const vit_frame_desc*, uint32_t, uint32_t, int64_t, hipStream_t, bool) {
    return hipErrorNotSupported;
}
#endif

void vit_set_err(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof t_err, fmt, ap);
    va_end(ap);
    if (getenv("VITERBI_AMD_VERBOSE")) fprintf(stderr, "[libviterbi] %s\n", t_err);
}

int vit_device_cus(int dev) {
    static std::mutex mu;
    static int cus[64] = {0};
    if (dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lk(mu);
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

hipError_t vit_optin_dynamic_lds(const void* const* kernels, int nkernels, int bytes, int dev, uint64_t* done) {
    static std::mutex mu;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(mu);
    if ((*done >> dev) & 1u) return hipSuccess;
    for (int i = 0; i < nkernels; i++) {
        const hipError_t e = hipFuncSetAttribute(kernels[i], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
    }
    *done |= 1ull << dev;
    return hipSuccess;
}

extern "C" {

const char* vit_last_error(void) { return t_err; }

int vit_device_count(void) {
    ensure_init();
    if (g_ndev == 0) set_err("%s", g_init_err);
    return g_ndev;
}

int vit_set_batch_window_us(int microseconds) {
    if (microseconds < 0) microseconds = 0;
    if (microseconds > 100000) microseconds = 100000;
    if (g_rstat.on) g_rstat.print_and_reset();
    return g_ring->window_us.exchange(microseconds);
}

int vit_set_batch_min_callers(int n) {
    if (n < 1) n = 1;
    return g_ring->min_callers.exchange(n);
}

int vit_set_batch_depth(int launches_in_flight) { return g_ring->set_depth(launches_in_flight); }

int vit_set_batch_spin_cpus(int cpus) { return g_ring->spin_cpus.exchange(cpus < 0 ? 0 : cpus); }

int vit_set_kernel(int which) {
    if (which < K_AUTO || which > K_MAX) which = K_AUTO;
    return g_kernel.exchange(which);
}

int vit_set_renorm_ge(int on) { return g_renorm_ge.exchange(on ? 1 : 0); }

unsigned char initialize(void) {
    // dllmain.cpp:156-160: clear the fault counter and re-run the (idempotent) set-up.
    g_fault.store(0);
    ensure_init();
    (void)hipGetLastError();
    return 1;
}

int GetCPUCaps(void) {
    ensure_init();
    if (g_device < 0) return 0;
    return VIT_CAPS_GFX950 | (g_cus << 8);
}

void WakeUpYMM(void) {
    if (hip_device_ready() != VIT_OK) return;
    VitDeviceGuard guard(g_device);
    if (ctx_prepare(g_device) != VIT_OK) return;
    (void)grow_pin(65536);
    (void)grow_dev(&t_ctx.d_in, &t_ctx.din_cap, 65536);
    (void)grow_dev(&t_ctx.d_sym8, &t_ctx.d8_cap, 65536);
    (void)grow_dev(&t_ctx.d_out, &t_ctx.dout_cap, 65536);
}

int vit_pack_symbols_dev(const uint32_t* d_symbols_u32, uint8_t* d_symbols_u8, int64_t nsym, void* stream) {
    if (hip_device_ready() != VIT_OK) return VIT_ERR_NO_DEVICE;
    if (nsym < 0 || (nsym > 0 && (!d_symbols_u32 || !d_symbols_u8))) {
        set_err("vit_pack_symbols_dev: bad arguments");
        return VIT_ERR_ARG;
    }
    hipError_t e = vit_launch_pack(d_symbols_u32, d_symbols_u8, nsym, (hipStream_t)stream);
    if (e != hipSuccess) { set_err("pack launch failed: %s", hipGetErrorString(e)); return VIT_ERR_HIP; }
    return VIT_OK;
}

int vit_decode_batch_dev(const uint8_t* d_symbols_u8, uint8_t* d_decoded, uint32_t framebits, int64_t nframes,
                         void* stream) {
    if (hip_device_ready() != VIT_OK) return VIT_ERR_NO_DEVICE;
    if (!valid_framebits(framebits) || nframes < 0 || (nframes > 0 && framebits > 0 && (!d_symbols_u8 || !d_decoded))) {
        set_err("vit_decode_batch_dev: bad arguments (framebits=%u nframes=%lld)", framebits, (long long)nframes);
        return VIT_ERR_ARG;
    }
    if (framebits == 0 || nframes == 0) return VIT_OK;
    return launch_decode(decode_mode(), d_symbols_u8, d_decoded, nullptr, framebits, framebits, nframes,
                         (hipStream_t)stream);
}

int vit_decode_batch_dev_u32(const uint32_t* d_symbols_u32, uint8_t* d_decoded, uint32_t framebits,
                             int64_t nframes, void* stream) {
    if (hip_device_ready() != VIT_OK) return VIT_ERR_NO_DEVICE;
    if (!valid_framebits(framebits) || nframes < 0 || (nframes > 0 && framebits > 0 && (!d_symbols_u32 || !d_decoded))) {
        set_err("vit_decode_batch_dev_u32: bad arguments");
        return VIT_ERR_ARG;
    }
    if (framebits == 0 || nframes == 0) return VIT_OK;
    const size_t nsym = (size_t)nframes * 4u * (framebits + VIT_TAIL);
    const DecodeMode mode = decode_mode();
    const int choice = mode.kernel;
    if (u32_in_place(choice, d_symbols_u32, framebits, nframes))  // read in place: no scratch, no extra launch
        return launch_decode_u32(mode, d_symbols_u32, nullptr, d_decoded, nullptr, framebits, framebits, nframes,
                                 (int64_t)nsym, (hipStream_t)stream);
    // The narrowed symbols go to this thread's scratch buffer ON THE CALLER'S CURRENT DEVICE (the device its
    // pointers and stream belong to), not on the library's default device.
    int dev = -1;
    HIPCHK(hipGetDevice(&dev));
    int rc = ctx_prepare(dev);
    if (rc != VIT_OK) return rc;
    rc = grow_dev(&t_ctx.d_sym8, &t_ctx.d8_cap, nsym);  // may synchronise the device (hipMalloc)
    if (rc != VIT_OK) return rc;
    // order the scratch buffer's reuse across the caller's streams
    if (!t_ctx.scratch_ev) HIPCHK(hipEventCreateWithFlags(&t_ctx.scratch_ev, hipEventDisableTiming));
    else HIPCHK(hipStreamWaitEvent((hipStream_t)stream, t_ctx.scratch_ev, 0));
    rc = launch_decode_u32(mode, d_symbols_u32, (uint8_t*)t_ctx.d_sym8, d_decoded, nullptr, framebits, framebits, nframes,
                           (int64_t)nsym, (hipStream_t)stream);
    if (rc != VIT_OK) return rc;
    HIPCHK(hipEventRecord(t_ctx.scratch_ev, (hipStream_t)stream));
    return VIT_OK;
}

int vit_decode_varlen_dev(const uint8_t* d_symbols_u8, uint8_t* d_decoded, const vit_frame_desc* d_desc,
                          int64_t nframes, uint32_t max_framebits, void* stream) {
    if (hip_device_ready() != VIT_OK) return VIT_ERR_NO_DEVICE;
    if (!valid_framebits(max_framebits) || nframes < 0 ||
        (nframes > 0 && (!d_symbols_u8 || !d_decoded || !d_desc))) {
        set_err("vit_decode_varlen_dev: bad arguments");
        return VIT_ERR_ARG;
    }
    if (nframes == 0 || max_framebits == 0) return VIT_OK;
    return launch_decode(decode_mode(), d_symbols_u8, d_decoded, d_desc, 0, max_framebits, nframes, (hipStream_t)stream);
}

int vit_decode_varlen_dev_checked(const uint8_t* d_symbols_u8, uint64_t sym_bytes, uint8_t* d_decoded, uint64_t out_bytes,
                                  const vit_frame_desc* d_desc, int64_t nframes, uint32_t max_framebits, void* stream) {
    if (hip_device_ready() != VIT_OK) return VIT_ERR_NO_DEVICE;
    if (!valid_framebits(max_framebits) || nframes < 0 ||
        (nframes > 0 && (!d_symbols_u8 || !d_decoded || !d_desc))) {
        set_err("vit_decode_varlen_dev_checked: bad arguments");
        return VIT_ERR_ARG;
    }
    if (nframes == 0 || max_framebits == 0) return VIT_OK;
    // the checked copy lives in this thread's scratch ON THE CALLER'S CURRENT DEVICE; its reuse across the caller's
    // streams is ordered by an event, like the u32 path's narrowing buffer
    int dev = -1;
    HIPCHK(hipGetDevice(&dev));
    int rc = ctx_prepare(dev);
    if (rc != VIT_OK) return rc;
    const bool fresh = !t_ctx.scratch_ev;
    if (fresh) HIPCHK(hipEventCreateWithFlags(&t_ctx.scratch_ev, hipEventDisableTiming));
    if ((rc = grow_dev(&t_ctx.d_desc, &t_ctx.ddesc_cap, (size_t)nframes * sizeof(vit_frame_desc))) != VIT_OK) return rc;
    if (!fresh) HIPCHK(hipStreamWaitEvent((hipStream_t)stream, t_ctx.scratch_ev, 0));
    hipError_t e = vit_check_descs_launch(d_desc, (vit_frame_desc*)t_ctx.d_desc, nframes, sym_bytes, out_bytes, (hipStream_t)stream);
    if (e != hipSuccess) { set_err("descriptor check launch failed: %s", hipGetErrorString(e)); return VIT_ERR_HIP; }
    rc = launch_decode(decode_mode(), d_symbols_u8, d_decoded, (const vit_frame_desc*)t_ctx.d_desc, 0, max_framebits, nframes,
                       (hipStream_t)stream);
    if (rc != VIT_OK) return rc;
    HIPCHK(hipEventRecord(t_ctx.scratch_ev, (hipStream_t)stream));
    return VIT_OK;
}

void vit_sort_descs(vit_frame_desc* h_desc, int64_t nframes) {
    if (!h_desc || nframes <= 1) return;
    std::stable_sort(h_desc, h_desc + nframes,
                     [](const vit_frame_desc& a, const vit_frame_desc& b) { return a.framebits > b.framebits; });
}

int vit_decode_batch_host(const uint8_t* h_symbols_u8, uint8_t* h_decoded, uint32_t framebits, int64_t nframes) {
    if (!valid_framebits(framebits) || nframes < 0 || (nframes > 0 && framebits > 0 && (!h_symbols_u8 || !h_decoded))) {
        set_err("vit_decode_batch_host: bad arguments");
        return VIT_ERR_ARG;
    }
    if (framebits == 0 || nframes == 0) return VIT_OK;
    int rc = hip_device_ready();
    if (rc != VIT_OK) return rc;
    VitDeviceGuard guard(g_device);
    if ((rc = ctx_prepare(g_device)) != VIT_OK) return rc;
    const size_t in_sz = (size_t)nframes * 4u * (framebits + VIT_TAIL);
    const size_t out_sz = (size_t)nframes * ((framebits + 7u) >> 3);
    if ((rc = grow_dev(&t_ctx.d_sym8, &t_ctx.d8_cap, in_sz)) != VIT_OK) return rc;
    if ((rc = grow_dev(&t_ctx.d_out, &t_ctx.dout_cap, out_sz)) != VIT_OK) return rc;
    HIPCHK(hipMemcpyAsync(t_ctx.d_sym8, h_symbols_u8, in_sz, hipMemcpyHostToDevice, t_ctx.stream));
    rc = launch_decode(decode_mode(), (const uint8_t*)t_ctx.d_sym8, (uint8_t*)t_ctx.d_out, nullptr, framebits, framebits,
                       nframes, t_ctx.stream);
    if (rc != VIT_OK) return rc;
    HIPCHK(hipMemcpyAsync(h_decoded, t_ctx.d_out, out_sz, hipMemcpyDeviceToHost, t_ctx.stream));
    HIPCHK(hipStreamSynchronize(t_ctx.stream));
    return VIT_OK;
}

static int deconvolve_impl(unsigned int framebits, unsigned int* symbols, unsigned char* decodedBits);
int deconvolve(unsigned int framebits, unsigned int* symbols, int unused, unsigned char* decodedBits) {
    (void)unused;  // never read by the reference either (deconvolve.cpp:447-526)
    int rc = 1;
    ScopedCall log("deconvolve", framebits, &rc);
    rc = deconvolve_impl(framebits, symbols, decodedBits);
    return rc;
}
static int deconvolve_impl(unsigned int framebits, unsigned int* symbols, unsigned char* decodedBits) {
    if (framebits == 0) return 0;  // C path: loop count 0, no memory touched
    if (g_fault.load()) return 1;  // save mode until initialize() (exc_handler.cpp:214,243)
    if (!symbols || !decodedBits || !valid_framebits(framebits)) {
        set_err("deconvolve: bad arguments (framebits=%u)", framebits);
        return 1;
    }
    if (hip_device_ready() != VIT_OK) return 1;
    InflightGuard inflight(g_ring->inflight);
    const DecodeMode mode = decode_mode();
    if (g_ring->engaged(mode)) {  // ingest stage: share a launch with the other callers in flight
        const int r = g_ring->call(mode, framebits, symbols, decodedBits);
        if (r == RING_OK) return 0;
        if (r == RING_FAILED) {
            g_fault.store(1);
            return 1;
        }
        // declined (no free slot, comparator mode differs from the open batch's): the direct path below
    }
    VitDeviceGuard guard(g_device);
    if (ctx_prepare(g_device) != VIT_OK) return 1;
    const size_t nsym = 4u * ((size_t)framebits + VIT_TAIL);
    const size_t out_sz = (framebits + 7u) >> 3;
    const size_t out_off = (nsym + 15u) & ~(size_t)15u;
    const size_t flag_off = (out_off + out_sz + 63u) & ~(size_t)63u;
    if (grow_pin(flag_off + 64) != VIT_OK) {
        g_fault.store(1);
        return 1;
    }
    auto fail = [&](const char* what, hipError_t e) {
        set_err("deconvolve: %s: %s", what, hipGetErrorString(e));
        g_fault.store(1);
        return 1;
    };
    // Zero-copy staging: the pinned buffer is mapped into the device's address space.  The caller's u32 symbols are
    // narrowed to the device format (one byte each, the low byte: deconvolve.cpp:158-165) while they are copied into
    // it; the decode kernel reads those 3 KB (FIC) straight from host memory, in one batch of loads, and writes its
    // (framebits+7)/8 bytes straight back: no hipMemcpy round trips, ONE launch, one wait.
    unsigned char* h_out = (unsigned char*)t_ctx.h_pin + out_off;
    narrow_symbols(symbols, (uint8_t*)t_ctx.h_pin, nsym);
    hipError_t e;
    if (pick_kernel(mode.kernel, framebits, 1) == K_LATENCY) {
        // Latency path: the kernel publishes a sequence number in the mapped buffer after its last output byte and
        // this thread spins on it - the end-of-kernel signal and hipStreamSynchronize's wake-up are off the call's
        // critical path.  A kernel that does not finish within the spin budget falls back to the stream sync.
        volatile uint32_t* h_flag = reinterpret_cast<volatile uint32_t*>((unsigned char*)t_ctx.h_pin + flag_off);
        const uint32_t seq = ++t_ctx.seq ? t_ctx.seq : ++t_ctx.seq;  // never 0
        *h_flag = 0;
        e = vit_launch_lat(t_ctx.h_pin_dev, false, (uint8_t*)t_ctx.h_pin_dev + out_off, nullptr, framebits, framebits, 1,
                           t_ctx.stream, reinterpret_cast<uint32_t*>((unsigned char*)t_ctx.h_pin_dev + flag_off), seq, mode.ge);
        if (e != hipSuccess) return fail("launch", e);
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (__atomic_load_n(const_cast<uint32_t*>(h_flag), __ATOMIC_ACQUIRE) != seq) {
            __builtin_ia32_pause();
            if (++spins > 4096u && (spins & 63u) == 0) sched_yield();  // more callers than cores: let the others run
            if ((spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
                if ((e = hipStreamSynchronize(t_ctx.stream)) != hipSuccess) return fail("sync", e);
                if (__atomic_load_n(const_cast<uint32_t*>(h_flag), __ATOMIC_ACQUIRE) != seq) {
                    set_err("deconvolve: the kernel ended without publishing its result");
                    g_fault.store(1);
                    return 1;
                }
            }
        }
        memcpy(decodedBits, h_out, out_sz);
        return 0;
    }
    if (launch_decode(mode, (const uint8_t*)t_ctx.h_pin_dev, (uint8_t*)t_ctx.h_pin_dev + out_off, nullptr, framebits, framebits, 1,
                      t_ctx.stream) != VIT_OK) {
        g_fault.store(1);
        return 1;
    }
    if ((e = hipStreamSynchronize(t_ctx.stream)) != hipSuccess) return fail("sync", e);
    memcpy(decodedBits, h_out, out_sz);
    return 0;
}

int vit_rs_batch_dev(const uint8_t* d_p, uint8_t* d_out, int32_t* d_ret, uint32_t RSDims, int64_t nsf, void* stream) {
    if (hip_device_ready() != VIT_OK) return VIT_ERR_NO_DEVICE;
    if (nsf < 0 || (nsf > 0 && RSDims > 0 && (!d_p || !d_out || !d_ret)) || RSDims > 65535u) {
        set_err("vit_rs_batch_dev: bad arguments");
        return VIT_ERR_ARG;
    }
    if (nsf == 0) return VIT_OK;
    if (RSDims == 0) {  // zero columns: the reference's loop does not run, returns 0
        hipError_t e0 = hipMemsetAsync(d_ret, 0, (size_t)nsf * sizeof(int32_t), (hipStream_t)stream);
        if (e0 != hipSuccess) { set_err("memset: %s", hipGetErrorString(e0)); return VIT_ERR_HIP; }
        return VIT_OK;
    }
    hipError_t e = rs_launch(d_p, d_out, d_ret, RSDims, nsf, (hipStream_t)stream);
    if (e != hipSuccess) { set_err("rs launch failed: %s", hipGetErrorString(e)); return VIT_ERR_HIP; }
    return VIT_OK;
}

int vit_dabplus_superframes_dev(const uint8_t* d_symbols_u8, uint8_t* d_work, uint8_t* d_rs_out, int32_t* d_ret,
                                uint32_t RSDims, int64_t nsf, void* stream) {
    if (hip_device_ready() != VIT_OK) return VIT_ERR_NO_DEVICE;
    const uint64_t framebits = 192ull * RSDims;  // 24 ms frame of an RSDims*8 kbit/s sub-channel
    if (RSDims == 0 || framebits > VIT_MAX_FRAMEBITS || nsf < 0 ||
        (nsf > 0 && (!d_symbols_u8 || !d_work || !d_rs_out || !d_ret))) {
        set_err("vit_dabplus_superframes_dev: bad arguments (RSDims=%u)", RSDims);
        return VIT_ERR_ARG;
    }
    if (nsf == 0) return VIT_OK;
    int rc = launch_decode(decode_mode(), d_symbols_u8, d_work, nullptr, (uint32_t)framebits, (uint32_t)framebits, 5 * nsf,
                           (hipStream_t)stream);
    if (rc != VIT_OK) return rc;
    return vit_rs_batch_dev(d_work, d_rs_out, d_ret, RSDims, nsf, stream);
}

int vit_rs_batch_host(const uint8_t* h_p, uint8_t* h_out, int32_t* h_ret, uint32_t RSDims, int64_t nsf) {
    if (nsf < 0 || RSDims > 65535u || (nsf > 0 && (!h_ret || (RSDims > 0 && (!h_p || !h_out))))) {
        set_err("vit_rs_batch_host: bad arguments");
        return VIT_ERR_ARG;
    }
    if (nsf == 0) return VIT_OK;
    if (RSDims == 0) { memset(h_ret, 0, (size_t)nsf * sizeof(int32_t)); return VIT_OK; }
    int rc = hip_device_ready();
    if (rc != VIT_OK) return rc;
    VitDeviceGuard guard(g_device);
    if ((rc = ctx_prepare(g_device)) != VIT_OK) return rc;
    const size_t in_sz = (size_t)nsf * 120u * RSDims, out_sz = (size_t)nsf * 110u * RSDims;
    if ((rc = grow_dev(&t_ctx.d_in, &t_ctx.din_cap, in_sz)) != VIT_OK) return rc;
    if ((rc = grow_dev(&t_ctx.d_out, &t_ctx.dout_cap, out_sz)) != VIT_OK) return rc;
    if ((rc = grow_dev(&t_ctx.d_ret, &t_ctx.dret_cap, (size_t)nsf * 4)) != VIT_OK) return rc;
    HIPCHK(hipMemcpyAsync(t_ctx.d_in, h_p, in_sz, hipMemcpyHostToDevice, t_ctx.stream));
    // columns at/after the first failure must keep the caller's bytes: seed the device copy
    HIPCHK(hipMemcpyAsync(t_ctx.d_out, h_out, out_sz, hipMemcpyHostToDevice, t_ctx.stream));
    hipError_t e = rs_launch((const uint8_t*)t_ctx.d_in, (uint8_t*)t_ctx.d_out, (int32_t*)t_ctx.d_ret, RSDims, nsf,
                             t_ctx.stream);
    if (e != hipSuccess) { set_err("rs launch failed: %s", hipGetErrorString(e)); return VIT_ERR_HIP; }
    HIPCHK(hipMemcpyAsync(h_out, t_ctx.d_out, out_sz, hipMemcpyDeviceToHost, t_ctx.stream));
    HIPCHK(hipMemcpyAsync(h_ret, t_ctx.d_ret, (size_t)nsf * 4, hipMemcpyDeviceToHost, t_ctx.stream));
    HIPCHK(hipStreamSynchronize(t_ctx.stream));
    return VIT_OK;
}

static int rscheck_impl(unsigned char* p, unsigned int RSDims, unsigned char* outVector);
int RScheckSuperframe(unsigned char* p, int startIx, unsigned int RSDims, unsigned char* outVector) {
    (void)startIx;  // rschecksf.cpp:69
    int rc = -1;
    ScopedCall log("RScheckSuperframe", RSDims, &rc);
    rc = rscheck_impl(p, RSDims, outVector);
    return rc;
}
static int rscheck_impl(unsigned char* p, unsigned int RSDims, unsigned char* outVector) {
    if (RSDims == 0) return 0;
    if (g_fault.load()) return -1;
    if (!p || !outVector || RSDims > 65535u) {
        set_err("RScheckSuperframe: bad arguments");
        return -1;
    }
    // Zero-copy like deconvolve(): the 120*RSDims input bytes, the caller's current output bytes (columns at
    // and after the first failure must keep them) and the return value live in the thread's mapped pinned
    // buffer; the kernel reads and writes host memory directly - one launch, one sync, no hipMemcpy.
    if (hip_device_ready() != VIT_OK) return -1;
    VitDeviceGuard guard(g_device);
    if (ctx_prepare(g_device) != VIT_OK) return -1;
    const size_t in_sz = 120u * (size_t)RSDims, out_sz = 110u * (size_t)RSDims;
    const size_t in_pad = (in_sz + 15u) & ~(size_t)15u, out_pad = (out_sz + 15u) & ~(size_t)15u;
    if (grow_pin(in_pad + out_pad + 64) != VIT_OK) {
        g_fault.store(1);
        return -1;
    }
    unsigned char* h = (unsigned char*)t_ctx.h_pin;
    unsigned char* d = (unsigned char*)t_ctx.h_pin_dev;
    memcpy(h, p, in_sz);
    memcpy(h + in_pad, outVector, out_sz);
    int32_t* h_ret = reinterpret_cast<int32_t*>(h + in_pad + out_pad);
    constexpr int32_t PENDING = 0x7FFFFFFF;  // never a return value: those are -1 or a root count
    const bool poll = RSDims <= 256u;      // the one-workgroup kernel publishes the return value last (see rs_kernel)
    *h_ret = poll ? PENDING : -1;
    hipError_t e = rs_launch(d, d + in_pad, reinterpret_cast<int32_t*>(d + in_pad + out_pad), RSDims, 1, t_ctx.stream, poll);
    if (e == hipSuccess && poll) {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (__atomic_load_n(h_ret, __ATOMIC_ACQUIRE) == PENDING) {
            __builtin_ia32_pause();
            if (++spins > 4096u && (spins & 63u) == 0) sched_yield();
            if ((spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
                e = hipStreamSynchronize(t_ctx.stream);
                if (e == hipSuccess && __atomic_load_n(h_ret, __ATOMIC_ACQUIRE) == PENDING) e = hipErrorUnknown;
                break;
            }
        }
    } else if (e == hipSuccess) {
        e = hipStreamSynchronize(t_ctx.stream);
    }
    if (e != hipSuccess) {
        set_err("RScheckSuperframe: %s", hipGetErrorString(e));
        g_fault.store(1);
        return -1;
    }
    memcpy(outVector, h + in_pad, out_sz);
    return *h_ret;
}

int RSCheckSuperframe(unsigned char* p, int startIx, unsigned int RSDims, unsigned char* outVector) {
    return RScheckSuperframe(p, startIx, RSDims, outVector);
}

}  // extern "C"
