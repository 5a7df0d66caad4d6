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
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
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
// Renormalisation comparator of every decoder kernel: 0 = `> 150` (the reference's C decoders, deconvolve.cpp:399,408,
// configuration Rel_cpp), 1 = `>= 150` (its MASM decoders, decon_avx2.asm:97,114 `cmp sil,150 ; jb mainloop`,
// configuration Rel_asm).  Environment VITERBI_AMD_RENORM_GE=1 sets the start-up value; vit_set_renorm_ge() changes it.
std::atomic<int> g_renorm_ge{[] {
    const char* e = getenv("VITERBI_AMD_RENORM_GE");
    return e && atoi(e) != 0 ? 1 : 0;
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
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
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
// Frames of one segment (<= 778 bits) in a uniform-length batch have two packed kernels: 4 frames per wavefront at 4
// wavefronts per SIMD (vit_pk.hip) and 8 frames per wavefront at 2 per SIMD (vit_pk8.hip).  The second executes 12 % fewer
// instructions per frame (8 % fewer than the shipped kernel since its fast traceback form) and is 14 % SLOWER on the benchmark batch (two wavefronts cannot hide the LDS round trips of the
// exchange and of the traceback: profiles/r03_ab_pk8.txt), so it is not the default: VITERBI_AMD_PK8=1 makes K_AUTO /
// K_PACKED take it, vit_set_kernel(4) forces it (tests, A/B runs).
bool pk8_default() {
    static const bool on = [] {
        const char* e = getenv("VITERBI_AMD_PK8");
        return e ? atoi(e) != 0 : false;
    }();
    return on;
}
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

// ---- ingest stage: micro-batching of concurrent deconvolve() callers -------------
// Off by default (window 0): every call then runs on its own thread's stream.  With a window of
// w microseconds, callers park their request in a queue; one worker thread collects what arrives
// within w us of the first request (or up to MAX_BATCH), stages all symbol buffers in pinned
// memory, and runs ONE u32->u8 pack + ONE variable-length decode for the batch.
struct BatchReq {
    uint32_t framebits;
    const unsigned int* symbols;
    unsigned char* out;
    int rc;
    bool done;
};
struct Batcher {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<BatchReq*> q;
    std::atomic<int> window_us{0};
    std::atomic<int> min_callers{8};  // batching engages only while at least this many deconvolve() calls are in flight
    std::atomic<int> inflight{0};     // deconvolve() calls currently executing (any path)
    int committed = 0;                // callers that chose the batch path and are not in a batch yet (under mu)
    bool started = false;
    static constexpr size_t MAX_BATCH = 256;
    Batcher() {
        // a host that only binds the five reference exports configures the stage through the environment
        // (the analogue of the reference's viterbi.txt): window in microseconds (0 = off) and the engagement threshold
        if (const char* e = getenv("VITERBI_AMD_BATCH_WINDOW_US")) {
            const int w = atoi(e);
            window_us.store(w < 0 ? 0 : w > 100000 ? 100000 : w);
        }
        if (const char* e = getenv("VITERBI_AMD_BATCH_MIN_CALLERS")) {
            const int n = atoi(e);
            min_callers.store(n < 1 ? 1 : n);
        }
    }
    std::vector<vit_frame_desc> h_desc;
    void* d_desc = nullptr; size_t ddesc_cap = 0;

    int process(std::vector<BatchReq*>& b) {
        VitDeviceGuard guard(g_device);
        int rc = ctx_prepare(g_device);  // the worker thread has its own stream and buffers
        if (rc != VIT_OK) return rc;
        size_t nsym = 0, nout = 0;
        uint32_t maxfb = 0;
        h_desc.resize(b.size());
        for (size_t i = 0; i < b.size(); i++) {
            const uint32_t fb = b[i]->framebits;
            h_desc[i].sym_offset = nsym;  // one byte per symbol after narrowing; multiple of 4
            h_desc[i].out_offset = nout;
            h_desc[i].framebits = fb;
            h_desc[i].reserved = 0;
            nsym += 4u * ((size_t)fb + VIT_TAIL);
            nout += (fb + 7u) >> 3;
            maxfb = fb > maxfb ? fb : maxfb;
        }
        const size_t desc_bytes = b.size() * sizeof(vit_frame_desc);
        const size_t out_pad = (nout + 15u) & ~(size_t)15u;
        if ((rc = grow_pin(nsym * 4 + out_pad + desc_bytes + 64)) != VIT_OK) return rc;
        if ((rc = grow_dev(&t_ctx.d_in, &t_ctx.din_cap, nsym * 4)) != VIT_OK) return rc;
        if ((rc = grow_dev(&t_ctx.d_sym8, &t_ctx.d8_cap, nsym)) != VIT_OK) return rc;
        if ((rc = grow_dev(&t_ctx.d_out, &t_ctx.dout_cap, nout)) != VIT_OK) return rc;
        if ((rc = grow_dev(&d_desc, &ddesc_cap, desc_bytes)) != VIT_OK) return rc;
        unsigned char* pin = (unsigned char*)t_ctx.h_pin;
        unsigned char* h_out = pin + nsym * 4;
        unsigned char* h_d = h_out + out_pad;
        size_t off = 0;
        for (BatchReq* r : b) {
            const size_t n = 4u * ((size_t)r->framebits + VIT_TAIL);
            memcpy(pin + off * 4, r->symbols, n * 4);
            off += n;
        }
        memcpy(h_d, h_desc.data(), desc_bytes);
        hipStream_t s = t_ctx.stream;
        HIPCHK(hipMemcpyAsync(t_ctx.d_in, pin, nsym * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(d_desc, h_d, desc_bytes, hipMemcpyHostToDevice, s));
        // sym_offset counts symbols: the same table addresses the u32 buffer and its narrowed copy
        rc = launch_decode_u32(decode_mode(), (const uint32_t*)t_ctx.d_in, (uint8_t*)t_ctx.d_sym8, (uint8_t*)t_ctx.d_out,
                               (const vit_frame_desc*)d_desc, 0, maxfb, (int64_t)b.size(), (int64_t)nsym, s);
        if (rc != VIT_OK) return rc;
        HIPCHK(hipMemcpyAsync(h_out, t_ctx.d_out, nout, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (size_t i = 0; i < b.size(); i++)
            memcpy(b[i]->out, h_out + h_desc[i].out_offset, (b[i]->framebits + 7u) >> 3);
        return VIT_OK;
    }

    void run() {
        std::vector<BatchReq*> batch;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return !q.empty(); });
                // The window is an upper bound, not a delay: the batch closes as soon as every caller that has
                // chosen the batch path has arrived (`committed` counts them from their decision on).
                const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(window_us.load());
                while (q.size() < MAX_BATCH && (int)q.size() < committed &&
                       cv_work.wait_until(lk, deadline) != std::cv_status::timeout) {}
                committed -= (int)q.size();
                batch.swap(q);
            }
            const int rc = process(batch);
            {
                std::lock_guard<std::mutex> lk(mu);
                for (BatchReq* r : batch) { r->rc = rc; r->done = true; }
            }
            cv_done.notify_all();
            batch.clear();
        }
    }

    // A caller takes the batch path only while enough calls are in flight to make a shared launch pay: with a
    // handful of threads every call keeps its own stream and never waits for anybody (profiles/r02_vitbench.txt).
    bool should_batch() {
        if (window_us.load(std::memory_order_relaxed) <= 0) return false;
        if (inflight.load(std::memory_order_relaxed) < min_callers.load(std::memory_order_relaxed)) return false;
        std::lock_guard<std::mutex> lk(mu);
        committed++;
        return true;
    }
    int submit(uint32_t framebits, const unsigned int* symbols, unsigned char* out) {  // after should_batch() == true
        BatchReq r{framebits, symbols, out, VIT_ERR_HIP, false};
        std::unique_lock<std::mutex> lk(mu);
        if (!started) {
            started = true;
            std::thread([this] { run(); }).detach();  // lives until the process ends
        }
        q.push_back(&r);
        cv_work.notify_one();
        cv_done.wait(lk, [&] { return r.done; });
        return r.rc;
    }
};
struct InflightGuard {
    std::atomic<int>& n;
    explicit InflightGuard(std::atomic<int>& c) : n(c) { n.fetch_add(1, std::memory_order_relaxed); }
    ~InflightGuard() { n.fetch_sub(1, std::memory_order_relaxed); }
};
Batcher* g_batcher = new Batcher();  // intentionally never destroyed (worker may outlive static dtors)

}  // namespace

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
    return g_ndev;
}

int vit_set_batch_window_us(int microseconds) {
    if (microseconds < 0) microseconds = 0;
    if (microseconds > 100000) microseconds = 100000;
    return g_batcher->window_us.exchange(microseconds);
}

int vit_set_batch_min_callers(int n) {
    if (n < 1) n = 1;
    return g_batcher->min_callers.exchange(n);
}

int vit_set_kernel(int which) {
    if (which < K_AUTO || which > K_PACKED8) which = K_AUTO;
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
    if (hip_device_ready() != VIT_OK) return 1;  // before should_batch(): a caller it has counted must reach submit()
    InflightGuard inflight(g_batcher->inflight);
    if (g_batcher->should_batch()) {
        if (g_batcher->submit(framebits, symbols, decodedBits) != VIT_OK) {
            g_fault.store(1);
            return 1;
        }
        return 0;
    }
    VitDeviceGuard guard(g_device);
    if (ctx_prepare(g_device) != VIT_OK) return 1;
    const size_t nsym = 4u * ((size_t)framebits + VIT_TAIL);
    const size_t out_sz = (framebits + 7u) >> 3;
    int rc;
    if ((rc = grow_pin(nsym * 4 + out_sz + 192)) != VIT_OK || (rc = grow_dev(&t_ctx.d_sym8, &t_ctx.d8_cap, nsym)) != VIT_OK) {
        g_fault.store(1);
        return 1;
    }
    auto fail = [&](const char* what, hipError_t e) {
        set_err("deconvolve: %s: %s", what, hipGetErrorString(e));
        g_fault.store(1);
        return 1;
    };
    // Zero-copy staging: the pinned buffer is mapped into the device's address space.  The decode kernel reads
    // the caller's u32 symbols straight from host memory (12 KB over PCIe; its pre-pass loads run 32 steps
    // ahead of their use, which covers the PCIe latency) and writes its (framebits+7)/8 bytes straight back:
    // no hipMemcpy round trips, ONE launch, one sync.  (With the wave-per-frame kernel selected the ingest
    // kernel narrows the symbols first.)
    unsigned char* h_out = (unsigned char*)t_ctx.h_pin + nsym * 4;
    memcpy(t_ctx.h_pin, symbols, nsym * 4);
    const DecodeMode mode = decode_mode();
    hipError_t e;
    if (pick_kernel(mode.kernel, framebits, 1) == K_LATENCY) {
        // Latency path: the kernel publishes a sequence number in the mapped buffer after its last output byte and
        // this thread spins on it - the end-of-kernel signal and hipStreamSynchronize's wake-up are off the call's
        // critical path.  A kernel that does not finish within the spin budget falls back to the stream sync.
        const size_t flag_off = (nsym * 4 + out_sz + 63u) & ~(size_t)63u;
        volatile uint32_t* h_flag = reinterpret_cast<volatile uint32_t*>((unsigned char*)t_ctx.h_pin + flag_off);
        const uint32_t seq = ++t_ctx.seq ? t_ctx.seq : ++t_ctx.seq;  // never 0
        *h_flag = 0;
        e = vit_launch_lat(t_ctx.h_pin_dev, true, (uint8_t*)t_ctx.h_pin_dev + nsym * 4, nullptr, framebits, framebits, 1,
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
    if (launch_decode_u32(mode, (const uint32_t*)t_ctx.h_pin_dev, (uint8_t*)t_ctx.d_sym8, (uint8_t*)t_ctx.h_pin_dev + nsym * 4,
                          nullptr, framebits, framebits, 1, (int64_t)nsym, t_ctx.stream) != VIT_OK) {
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
