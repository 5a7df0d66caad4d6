// vit_internal.h -- launchers shared between the kernel TUs and the C-ABI TU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "viterbi_amd.h"

#define VIT_MAX_FRAMEBITS 9216u  // deconvolve.cpp:93,127 (384*24)
#define VIT_TAIL 6u              // K-1 tail steps
#define VIT_SORT_BINS (VIT_MAX_FRAMEBITS / 8u + 1u)  // counting-sort keys framebits/8

// renorm_ge (all decoder launchers): the renormalisation test on state 0.  false = `> 150`, the reference's C decoders
// (deconvolve.cpp:399,408; build configuration Rel_cpp); true = `>= 150`, its MASM decoders (decon_avx2.asm:97,114
// `cmp sil,150 ; jb mainloop`; Rel_asm, the configuration QIRX ships).  The two differ on hard-decision input from a
// poor channel, where path metrics reach the 0 / 255 clamps.
// Wave-per-frame kernel (lane = trellis state); any even framebits <= 9216.
hipError_t vit_launch_wave(const uint8_t* d_sym, uint8_t* d_out, const vit_frame_desc* d_desc,
                           uint32_t framebits, uint32_t max_framebits, int64_t nframes,
                           hipStream_t stream, bool renorm_ge);
// Packed kernel: 4 frames per wavefront, 2 states x 2 frames per lane register.
// Every even framebits <= 9216 (frames longer than 778 bits spill their history through HBM).
bool vit_pk_supported(uint32_t max_framebits);
// sym32: d_symbols holds the reference ABI's u32-per-symbol format (16-byte aligned; sym_offset
// then counts symbols); the narrowing to the low byte is fused into the kernel's pre-pass.
hipError_t vit_launch_pk(const void* d_symbols, bool sym32, uint8_t* d_out, const vit_frame_desc* d_desc,
                         uint32_t framebits, uint32_t max_framebits, int64_t nframes,
                         hipStream_t stream, bool renorm_ge);
// Packed kernel, 8 frames per wavefront at 2 wavefronts per SIMD (vit_pk8.hip): frames of one segment (framebits <= 778).
// An experiment, compiled in only with -DVIT_WITH_PK8 (otherwise vit_api.hip holds stubs that report "not supported").
bool vit_pk8_supported(uint32_t max_framebits);
hipError_t vit_launch_pk8(const void* d_symbols, bool sym32, uint8_t* d_out, const vit_frame_desc* d_desc,
                          uint32_t framebits, uint32_t max_framebits, int64_t nframes, hipStream_t stream, bool renorm_ge);
// Latency kernel: one frame per wavefront, one path metric per lane, DPP partner fetches (small launches).
#define VIT_LAT_MAX_FRAMES 2048  // auto selection: up to two waves per SIMD; beyond that the packed kernel's throughput wins
// done_flag (optional, nframes == 1 only): a word in host-visible memory that receives done_seq, with system-scope
// release semantics, after the frame's last output byte.
hipError_t vit_launch_lat(const void* d_symbols, bool sym32, uint8_t* d_out, const vit_frame_desc* d_desc,
                          uint32_t framebits, uint32_t max_framebits, int64_t nframes,
                          hipStream_t stream, uint32_t* done_flag, uint32_t done_seq, bool renorm_ge);
// Ingest-stage launch (vit_api.hip): one frame per slot of a mapped pinned ring.  The table travels BY VALUE in the
// kernel arguments (no descriptor fetch over PCIe in front of the symbol loads).
#define VIT_RING_MAXB 128u  // frames per launch
struct VitRingTable {
    uint32_t n;         // frames of this launch (= grid)
    uint32_t seq;       // what every slot's completion word receives (never 0)
    uint32_t stride;    // bytes per slot
    uint32_t out_off;   // offset of a slot's (framebits+7)/8 output bytes
    uint32_t flag_off;  // offset of a slot's completion word
    uint16_t slot[VIT_RING_MAXB];
    uint16_t fb[VIT_RING_MAXB];  // framebits of the frame in that slot (even, 2..9216)
};
hipError_t vit_launch_lat_ring(uint8_t* d_ring, const VitRingTable& tbl, uint32_t max_framebits, hipStream_t stream,
                               bool renorm_ge);
// frames the latency kernel can keep resident at one wave per SIMD or so for this frame length (LDS-limited)
int64_t vit_lat_capacity(uint32_t max_framebits, int dev);
// Length-sorted (longest first) copy of a device descriptor table; d_bins = 2*VIT_SORT_BINS words of scratch.
// bins_clean (optional): the caller's flag "the histogram half of d_bins is zero" - the scan kernel leaves it so; d_hdr (optional): 64
// words the scan kernel clears (the persistent kernel's counter header in front of the bins).
hipError_t vit_sort_descs_launch(const vit_frame_desc* d_desc, vit_frame_desc* d_sorted, int64_t nframes,
                                 uint32_t max_framebits, unsigned* d_bins, hipStream_t stream, bool* bins_clean = nullptr,
                                 unsigned* d_hdr = nullptr);
// Copy of a device descriptor table in which every descriptor that reaches outside [0, sym_bytes) / [0, out_bytes)
// has its framebits replaced by 0xFFFFFFFF (skipped by every kernel).
hipError_t vit_check_descs_launch(const vit_frame_desc* d_desc, vit_frame_desc* d_checked, int64_t nframes,
                                  uint64_t sym_bytes, uint64_t out_bytes, hipStream_t stream);
// u32 -> u8 narrowing (low byte), the reference ABI's symbol format to the device format.
hipError_t vit_launch_pack(const uint32_t* d_sym32, uint8_t* d_sym8, int64_t nsym,
                           hipStream_t stream);
// RS(120,110) superframe check, one lane per column.
// host_polls_ret (nsf == 1, rsdims <= 256): d_ret is host-visible and receives its value with system-scope release
// semantics after the last output byte, so the host may spin on it instead of synchronising the stream.
hipError_t rs_launch(const uint8_t* d_p, uint8_t* d_out, int32_t* d_ret, uint32_t rsdims,
                     int64_t nsf, hipStream_t stream, bool host_polls_ret = false);

// ---- host-side helpers shared by the TUs -------------------------------------------------------
// per-thread error text behind vit_last_error() (printf-style)
void vit_set_err(const char* fmt, ...);
// CU count of a HIP device, cached (persistent grids are sized by it)
int vit_device_cus(int dev);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: opt the kernels in once per (device, caller).
// `done` is the caller's bitmask of devices already handled (guarded by an internal mutex).
hipError_t vit_optin_dynamic_lds(const void* const* kernels, int nkernels, int bytes, int dev, uint64_t* done);
// Restores the calling thread's current HIP device on scope exit (library calls must not leave it changed).
struct VitDeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit VitDeviceGuard(int want) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (want >= 0 && prev != want && hipSetDevice(want) == hipSuccess) changed = true;
    }
    ~VitDeviceGuard() {
        if (changed && prev >= 0) (void)hipSetDevice(prev);
    }
    VitDeviceGuard(const VitDeviceGuard&) = delete;
    VitDeviceGuard& operator=(const VitDeviceGuard&) = delete;
};
