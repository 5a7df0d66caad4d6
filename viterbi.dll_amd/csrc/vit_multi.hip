// vit_multi.hip -- ONE host process, several gfx950 devices: a stream of equal-length frames that lives
// on the first ("root") device is decoded by all of them.  BASELINE.json configs[3] / SURVEY 8e through the
// C ABI (vit_decode_stream_multi, include/viterbi_amd.h Part 3); the torch.distributed twin for one process
// per GPU is viterbi.dll_amd/sharding.py:decode_stream.
//
// Data movement is RCCL point-to-point over xGMI: ncclSend/ncclRecv pairs fused in one ncclGroupStart/End per
// pipeline step.  librccl is dlopen'ed on first use: single-GPU users of the drop-in never load it.
//
// Plan (same as sharding.StreamPlan).  The stream is cut into chunks of  span = root_frames + (W-1)*chunk_frames
// consecutive frames (W = number of ranks).  Inside a chunk the root keeps the first root_frames and rank r the
// r-th block of chunk_frames: round-robin at BLOCK granularity, so every slice is contiguous in the root's
// buffers and is sent from / received into place - no packing kernel, no staging copy on the root.
//
// Pipeline.  Per rank: a transfer stream s_x and a compute stream s_c, two symbol and two output buffers.
//   transfer step j : { S_j : block j of every peer leaves the root }  +  { G_(j-2) : decoded block j-2 comes back }
//                     both directions of every link in ONE group; waits for decode j-2 (frees the buffer halves)
//   compute  step j : waits for transfer step j, decodes block j
// so block j+1 is on the wire while block j is decoded, and the results of j-1 travel back behind it.
//
// What it cannot do: the root's xGMI egress bounds the scatter; per-GPU ingestion
// (bench.py's default shard mode) is the path that scales with the number of GPUs.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library itself is dlopen'ed

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "vit_internal.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;  // optional
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load() {
        if (handle) return true;
        const char* names[3] = {getenv("VITERBI_AMD_RCCL_LIB"), "librccl.so.1", "librccl.so"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            if ((handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        }
        if (!handle) {
            vit_set_err("vit_decode_stream_multi: cannot load librccl (%s)", dlerror());
            return false;
        }
#define VIT_SYM(field, name)                                                    \
    field = reinterpret_cast<decltype(field)>(dlsym(handle, name));             \
    if (!field) {                                                               \
        vit_set_err("vit_decode_stream_multi: librccl has no %s", name);        \
        return false;                                                           \
    }
        VIT_SYM(CommInitAll, "ncclCommInitAll")
        VIT_SYM(CommDestroy, "ncclCommDestroy")
        VIT_SYM(Send, "ncclSend")
        VIT_SYM(Recv, "ncclRecv")
        VIT_SYM(GroupStart, "ncclGroupStart")
        VIT_SYM(GroupEnd, "ncclGroupEnd")
        VIT_SYM(GetErrorString, "ncclGetErrorString")
#undef VIT_SYM
        CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(handle, "ncclCommAbort"));
        return true;
    }
};

struct Rank {
    int dev = -1;
    int comm = 0;                // index into MultiCtx::comms (= index of its device in the device list)
    bool own_sx = false;         // the loop-back rank shares the root's transfer stream
    hipStream_t s_x = nullptr, s_c = nullptr;
    hipEvent_t ev_x[2] = {nullptr, nullptr}, ev_dec[2] = {nullptr, nullptr};
    uint8_t* rbuf[2] = {nullptr, nullptr};  // received symbol blocks (peers only)
    uint8_t* obuf[2] = {nullptr, nullptr};  // decoded blocks waiting for their way back
    size_t rcap = 0, ocap = 0;
};

struct MultiCtx {
    std::vector<int> devices;
    bool loopback = false;
    std::vector<ncclComm_t> comms;
    std::vector<Rank> ranks;  // ranks[0] = root
    hipEvent_t ev_in = nullptr;
    bool ready = false;
    bool group_failed = false;  // a send/recv group was closed on an error path: it may hold unmatched transfers
};

std::mutex g_mu;  // one multi-device call at a time
Rccl g_rccl;
MultiCtx g_ctx;

#define MHIP(call)                                                                          \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            vit_set_err("vit_decode_stream_multi: %s: %s", #call, hipGetErrorString(e_));   \
            return VIT_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)
#define MNCCL(call)                                                                                  \
    do {                                                                                             \
        ncclResult_t r_ = (call);                                                                    \
        if (r_ != ncclSuccess) {                                                                     \
            vit_set_err("vit_decode_stream_multi: %s: %s", #call, g_rccl.GetErrorString(r_));        \
            return VIT_ERR_HIP;                                                                      \
        }                                                                                            \
    } while (0)

void destroy_ctx(MultiCtx& c) {
    for (Rank& r : c.ranks) {
        if (r.dev >= 0) (void)hipSetDevice(r.dev);
        for (int i = 0; i < 2; i++) {
            if (r.rbuf[i]) (void)hipFree(r.rbuf[i]);
            if (r.obuf[i]) (void)hipFree(r.obuf[i]);
            if (r.ev_x[i]) (void)hipEventDestroy(r.ev_x[i]);
            if (r.ev_dec[i]) (void)hipEventDestroy(r.ev_dec[i]);
        }
        if (r.s_c) (void)hipStreamDestroy(r.s_c);
        if (r.own_sx && r.s_x) (void)hipStreamDestroy(r.s_x);
    }
    if (c.ev_in) (void)hipEventDestroy(c.ev_in);
    // After a group that was closed on an error path the communicators may hold transfers whose partner was never
    // posted: ncclCommDestroy would wait for them, ncclCommAbort ends them (round-3 advisor finding).
    for (ncclComm_t cm : c.comms)
        if (cm) (void)((c.group_failed && g_rccl.CommAbort) ? g_rccl.CommAbort(cm) : g_rccl.CommDestroy(cm));
    c = MultiCtx();
}

int build_ctx(const int* devices, int ndev, bool loopback) {
    MultiCtx& c = g_ctx;
    if (c.ready && c.loopback == loopback && (int)c.devices.size() == ndev &&
        memcmp(c.devices.data(), devices, sizeof(int) * (size_t)ndev) == 0)
        return VIT_OK;
    if (c.ready) destroy_ctx(c);
    c.devices.assign(devices, devices + ndev);
    c.loopback = loopback;
    const bool need_comm = ndev > 1 || loopback;
    if (need_comm) {
        if (!g_rccl.load()) return VIT_ERR_HIP;
        c.comms.assign((size_t)ndev, nullptr);
        MNCCL(g_rccl.CommInitAll(c.comms.data(), ndev, devices));  // rank i of the communicator = devices[i]
    }
    const int W = ndev + (loopback ? 1 : 0);
    c.ranks.assign((size_t)W, Rank());
    for (int r = 0; r < W; r++) {
        Rank& k = c.ranks[(size_t)r];
        const bool is_loop = loopback && r == W - 1;  // an extra rank on the root's device that talks to it over RCCL
        k.comm = is_loop ? 0 : r;
        k.dev = devices[k.comm];
        MHIP(hipSetDevice(k.dev));
        MHIP(hipStreamCreateWithFlags(&k.s_c, hipStreamNonBlocking));
        if (is_loop) {
            k.s_x = c.ranks[0].s_x;  // one communicator, one transfer stream
        } else {
            MHIP(hipStreamCreateWithFlags(&k.s_x, hipStreamNonBlocking));
            k.own_sx = true;
        }
        for (int i = 0; i < 2; i++) {
            MHIP(hipEventCreateWithFlags(&k.ev_x[i], hipEventDisableTiming));
            MHIP(hipEventCreateWithFlags(&k.ev_dec[i], hipEventDisableTiming));
        }
    }
    MHIP(hipSetDevice(devices[0]));
    MHIP(hipEventCreateWithFlags(&c.ev_in, hipEventDisableTiming));
    c.ready = true;
    return VIT_OK;
}

struct Plan {
    int64_t nframes, chunk, rootf, span, nchunks;
    int W;
    // (first frame, count) of the block rank r decodes in chunk k; rank 0 = root
    void block(int64_t k, int r, int64_t* lo, int64_t* n) const {
        const int64_t off = r == 0 ? 0 : rootf + (int64_t)(r - 1) * chunk;
        const int64_t len = r == 0 ? rootf : chunk;
        int64_t a = k * span + off, b = a + len;
        if (a > nframes) a = nframes;
        if (b > nframes) b = nframes;
        *lo = a;
        *n = (k < 0 || k >= nchunks) ? 0 : b - a;
    }
};

int grow(uint8_t** p, size_t* cap, size_t need) {
    if (*cap >= need) return VIT_OK;
    if (*p) MHIP(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    MHIP(hipMalloc(reinterpret_cast<void**>(p), need));
    *cap = need;
    return VIT_OK;
}

int run(const uint8_t* d_sym, uint8_t* d_out, uint32_t framebits, int64_t nframes, const Plan& P, hipStream_t in_stream) {
    MultiCtx& c = g_ctx;
    const size_t symlen = 4u * ((size_t)framebits + VIT_TAIL), olen = (framebits + 7u) >> 3;
    const int W = P.W;
    Rank& root = c.ranks[0];
    // buffers of the peers: two halves each way
    for (int r = 1; r < W; r++) {
        Rank& k = c.ranks[(size_t)r];
        MHIP(hipSetDevice(k.dev));
        int rc;
        for (int i = 0; i < 2; i++) {
            size_t cap = k.rcap, ocap = k.ocap;
            if ((rc = grow(&k.rbuf[i], &cap, (size_t)P.chunk * symlen)) != VIT_OK) return rc;
            if ((rc = grow(&k.obuf[i], &ocap, (size_t)P.chunk * olen)) != VIT_OK) return rc;
            if (i == 1) { k.rcap = cap; k.ocap = ocap; }
        }
    }
    // the caller's stream has produced the symbols: both root streams start behind it
    MHIP(hipSetDevice(root.dev));
    MHIP(hipEventRecord(c.ev_in, in_stream));
    MHIP(hipStreamWaitEvent(root.s_x, c.ev_in, 0));
    MHIP(hipStreamWaitEvent(root.s_c, c.ev_in, 0));

    for (int64_t j = 0; j < P.nchunks + 2; j++) {
        const int h = (int)(j & 1);
        // ---- transfer step j: S_j out, G_(j-2) back ----
        bool any = false;
        for (int r = 1; r < W; r++) {
            Rank& k = c.ranks[(size_t)r];
            int64_t lo, n2;
            P.block(j - 2, r, &lo, &n2);
            if (n2) {  // decode j-2 has finished: its output half is complete and its symbol half is free
                MHIP(hipSetDevice(k.dev));
                MHIP(hipStreamWaitEvent(k.s_x, k.ev_dec[h], 0));
            }
            int64_t n0;
            P.block(j, r, &lo, &n0);
            any = any || n0 || n2;
        }
        if (any) {
            MNCCL(g_rccl.GroupStart());
            // an error return between GroupStart and GroupEnd must still close the group: RCCL's group depth is
            // per thread, and the clean-up after a failed call (ncclCommDestroy) runs on this thread
            struct GroupGuard {
                MultiCtx& ctx;
                bool open = true;
                ~GroupGuard() {
                    if (!open) return;
                    ctx.group_failed = true;  // the exported call aborts the communicators before it waits for anything
                    (void)g_rccl.GroupEnd();
                }
            } group{c};
            // TEST HOOK (tests/test_gpu_multi.py): VITERBI_AMD_TEST_MULTI_FAULT=<j> fails transfer step j after its first
            // complete send/recv pair, i.e. on the error path between GroupStart and GroupEnd
            const char* fault_env = getenv("VITERBI_AMD_TEST_MULTI_FAULT");
            const int64_t fault_at = fault_env ? atoll(fault_env) : -1;
            bool posted = false;
            for (int r = 1; r < W; r++) {
                Rank& k = c.ranks[(size_t)r];
                int64_t lo, n;
                P.block(j, r, &lo, &n);
                if (n) {
                    MNCCL(g_rccl.Send(d_sym + (size_t)lo * symlen, (size_t)n * symlen, ncclUint8, k.comm, c.comms[0], root.s_x));
                    MNCCL(g_rccl.Recv(k.rbuf[h], (size_t)n * symlen, ncclUint8, 0, c.comms[(size_t)k.comm], k.s_x));
                    posted = true;
                }
                if (posted && fault_at == j) {
                    vit_set_err("vit_decode_stream_multi: injected fault in transfer step %lld (VITERBI_AMD_TEST_MULTI_FAULT)", (long long)j);
                    return VIT_ERR_HIP;
                }
                P.block(j - 2, r, &lo, &n);
                if (n) {
                    MNCCL(g_rccl.Send(k.obuf[h], (size_t)n * olen, ncclUint8, 0, c.comms[(size_t)k.comm], k.s_x));
                    MNCCL(g_rccl.Recv(d_out + (size_t)lo * olen, (size_t)n * olen, ncclUint8, k.comm, c.comms[0], root.s_x));
                }
            }
            group.open = false;
            MNCCL(g_rccl.GroupEnd());
        }
        // ---- compute step j ----
        if (j >= P.nchunks) continue;
        for (int r = 0; r < W; r++) {
            Rank& k = c.ranks[(size_t)r];
            int64_t lo, n;
            P.block(j, r, &lo, &n);
            if (!n) continue;
            MHIP(hipSetDevice(k.dev));
            int rc;
            if (r == 0) {
                rc = vit_decode_batch_dev(d_sym + (size_t)lo * symlen, d_out + (size_t)lo * olen, framebits, n, k.s_c);
            } else {
                MHIP(hipEventRecord(k.ev_x[h], k.s_x));       // transfer step j on this rank's stream
                MHIP(hipStreamWaitEvent(k.s_c, k.ev_x[h], 0));  // block j has arrived, obuf[h] has left
                rc = vit_decode_batch_dev(k.rbuf[h], k.obuf[h], framebits, n, k.s_c);
                if (rc == VIT_OK) MHIP(hipEventRecord(k.ev_dec[h], k.s_c));
            }
            if (rc != VIT_OK) return rc;
        }
    }
    for (int r = 0; r < W; r++) {
        Rank& k = c.ranks[(size_t)r];
        MHIP(hipSetDevice(k.dev));
        MHIP(hipStreamSynchronize(k.s_c));
        MHIP(hipStreamSynchronize(k.s_x));
    }
    (void)nframes;
    return VIT_OK;
}

}  // namespace

extern "C" int vit_decode_stream_multi(const uint8_t* d_symbols_u8, uint8_t* d_decoded, uint32_t framebits, int64_t nframes,
                                       const int* devices, int ndev, int64_t chunk_frames, int64_t root_frames,
                                       unsigned flags, void* stream) {
    if (vit_device_count() <= 0) {
        vit_set_err("no usable gfx950 (MI355X) HIP device; libviterbi has no CPU path");
        return VIT_ERR_NO_DEVICE;
    }
    const bool loopback = (flags & VIT_MULTI_LOOPBACK) != 0;
    const int W = ndev + (loopback ? 1 : 0);
    if (!devices || ndev < 1 || ndev > 64 || framebits > VIT_MAX_FRAMEBITS || (framebits & 1u) || nframes < 0 ||
        chunk_frames <= 0 || root_frames < -1 || (W == 1 && root_frames == 0) || (flags & ~VIT_MULTI_LOOPBACK) ||
        (nframes > 0 && framebits > 0 && (!d_symbols_u8 || !d_decoded))) {
        vit_set_err("vit_decode_stream_multi: bad arguments (ndev=%d framebits=%u nframes=%lld chunk_frames=%lld)", ndev,
                    framebits, (long long)nframes, (long long)chunk_frames);
        return VIT_ERR_ARG;
    }
    int nvisible = 0;
    if (hipGetDeviceCount(&nvisible) != hipSuccess) nvisible = 0;
    for (int i = 0; i < ndev; i++) {
        bool ok = devices[i] >= 0 && devices[i] < nvisible;
        for (int k = 0; ok && k < i; k++) ok = devices[k] != devices[i];
        hipDeviceProp_t pr;
        if (ok) ok = hipGetDeviceProperties(&pr, devices[i]) == hipSuccess && strncmp(pr.gcnArchName, "gfx950", 6) == 0;
        if (!ok) {
            vit_set_err("vit_decode_stream_multi: devices[%d]=%d is not a usable, distinct gfx950 device", i, devices[i]);
            return VIT_ERR_ARG;
        }
    }
    if (nframes == 0 || framebits == 0) return VIT_OK;
    Plan P;
    P.nframes = nframes;
    P.W = W;
    P.chunk = chunk_frames;
    P.rootf = root_frames < 0 ? chunk_frames : root_frames;
    P.span = P.rootf + (int64_t)(W - 1) * P.chunk;
    P.nchunks = (nframes + P.span - 1) / P.span;
    std::lock_guard<std::mutex> lk(g_mu);
    VitDeviceGuard guard(-1);  // whatever happens below, the caller's current device comes back
    guard.changed = true;
    int rc = build_ctx(devices, ndev, loopback);
    if (rc == VIT_OK) rc = run(d_symbols_u8, d_decoded, framebits, nframes, P, (hipStream_t)stream);
    if (rc != VIT_OK) {  // drain whatever was enqueued, drop the context: the next call starts clean
        if (g_ctx.group_failed && g_rccl.CommAbort) {  // first end transfers that may never find their partner
            for (ncclComm_t& cm : g_ctx.comms)
                if (cm) { (void)g_rccl.CommAbort(cm); cm = nullptr; }
        }
        for (int i = 0; i < ndev; i++)
            if (hipSetDevice(devices[i]) == hipSuccess) (void)hipDeviceSynchronize();
        destroy_ctx(g_ctx);
    }
    return rc;
}
