/*
 * viterbi_amd.h -- C ABI of libviterbi.so, the MI355X (gfx950) drop-in for the
 * compute path of Drehrumbum/viterbi.dll.
 *
 * Part 1 are the five exports of the reference's viterbi.def:4-8, same names,
 * argument meaning and return values, so a caller that binds viterbi.dll
 * (QIRX via P/Invoke, viterbi-benchmark.cpp:201-229 via GetProcAddress) binds
 * this library unchanged (SysV x86-64 instead of Win64).  Part 2 is the build's
 * own batched, device-resident extension: one 96-byte call cannot feed a GPU,
 * so the throughput path takes many frames per call.
 *
 * All entry points are thread-safe.  No entry point ever computes on the CPU:
 * when no HIP device is usable they return the error codes below.
 */
#ifndef VITERBI_AMD_H
#define VITERBI_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ *
 * Part 1 -- drop-in exports (reference: viterbi.def:4-8)
 * ------------------------------------------------------------------------ */

/* Replaces `deconvolve` (deconvolve.cpp:551-554, typedef DECON viterbi.h:113;
 * caller's view viterbi-benchmark.cpp:72-73).
 *   framebits : decoded bits per frame, even, <= 9216 (deconvolve.cpp:126-127)
 *   symbols   : 4*(framebits+6) soft symbols, one per u32, low byte used
 *               (0 = strong "0", 255 = strong "1"; deconvolve.cpp:141-165)
 *   unused    : ignored, like the reference's `inputLength`
 *   decodedBits: receives (framebits+7)/8 bytes, MSB first
 * Returns 0 on success, 1 on any failure ("save mode" value of
 * viterbi_helpers.asm:184-186): bad arguments, no GPU, HIP error.
 * framebits == 0 returns 0 without touching memory (C path behaviour). */
int deconvolve(unsigned int framebits, unsigned int *symbols, int unused,
               unsigned char *decodedBits);

/* Replaces `initialize` (dllmain.cpp:156-160): clears the fault state ("save
 * mode") and makes sure the device probe has run; cheap, idempotent; returns
 * non-zero (true).  The GPU is chosen once per process, at the first call into
 * the library, from the environment variable VITERBI_AMD_DEVICE (index among the
 * gfx950 devices, default 0) -- the analogue of the reference's viterbi.txt. */
unsigned char initialize(void);

/* Replaces `RScheckSuperframe` (rschecksf.cpp:65-93).  RS(120,110) over
 * GF(2^8)/0x11D on the RSDims columns of p[120*RSDims]; corrected first 110
 * rows go to outVector[110*RSDims].  Returns the summed root counts, or -1 at
 * the first uncorrectable column; that column and all later ones are left
 * unwritten in outVector.  startIx is ignored (rschecksf.cpp:69).
 * Also -1 on bad arguments / no GPU / HIP error (see vit_last_error()). */
int RScheckSuperframe(unsigned char *p, int startIx, unsigned int RSDims,
                      unsigned char *outVector);
/* BASELINE.json spells it with a capital C; same function. */
int RSCheckSuperframe(unsigned char *p, int startIx, unsigned int RSDims,
                      unsigned char *outVector);

/* Replaces `GetCPUCaps` (viterbi_helpers.asm:48-157, bit masks
 * getcpucaps.h:27-38).  There is no x86 dispatch here: returns 0 when no
 * usable GPU was found, otherwise VIT_CAPS_GFX950 | number of CUs << 8. */
int GetCPUCaps(void);
#define VIT_CAPS_GFX950 0x1

/* Replaces `WakeUpYMM` (dllmain.cpp:54-56 / viterbi_helpers.asm:160-176): a
 * warm-up hook.  Here it creates the calling thread's HIP stream and staging
 * buffers so the first deconvolve() does not pay for them. */
void WakeUpYMM(void);

/* ------------------------------------------------------------------------ *
 * Part 2 -- batched extension (not in the reference)
 * ------------------------------------------------------------------------ */

#define VIT_OK 0
#define VIT_ERR_ARG 1
#define VIT_ERR_NO_DEVICE 2
#define VIT_ERR_HIP 3

/* Last error text of the calling thread ("" if none). */
const char *vit_last_error(void);
/* Number of usable gfx950 devices (0 = none). */
int vit_device_count(void);

/* Frame descriptor for variable-length batches.
 * sym_offset: byte offset of the frame's first soft symbol in the u8 symbol
 * buffer; MUST be a multiple of 4 (the kernels load one dword per trellis step) -
 * a descriptor that is not is skipped like one with an invalid length, its
 * output stays untouched; the frame owns 4*(framebits+6) bytes from there.
 * out_offset: byte offset of its (framebits+7)/8 output bytes. */
typedef struct vit_frame_desc {
    uint64_t sym_offset;
    uint64_t out_offset;
    uint32_t framebits; /* even, <= 9216 */
    uint32_t reserved;
} vit_frame_desc;

/* Device format of the soft symbols: one byte per symbol (the low byte of the
 * reference's u32), frames back to back: frame f at f*4*(framebits+6).
 * Decoded output: frame f at f*((framebits+7)/8), MSB first; framebits may be
 * any even number up to 9216 (a partial last byte is padded with zero bits,
 * like the reference's ChainBack writes it).
 * All *_dev calls take DEVICE pointers and enqueue on `stream` (a hipStream_t,
 * NULL = default stream) without synchronising. */
int vit_decode_batch_dev(const uint8_t *d_symbols_u8, uint8_t *d_decoded,
                         uint32_t framebits, int64_t nframes, void *stream);
/* Same, symbols still in the reference ABI format (u32 per symbol, low byte
 * used).  With a 16-byte aligned buffer the decoder
 * reads them in place (narrowing fused into the kernel); otherwise they are
 * narrowed on the device into an internal scratch buffer first. */
int vit_decode_batch_dev_u32(const uint32_t *d_symbols_u32, uint8_t *d_decoded,
                             uint32_t framebits, int64_t nframes, void *stream);
/* Variable-length batch; d_desc is a DEVICE array of nframes descriptors,
 * max_framebits the largest framebits in it (host-known).  A descriptor whose
 * framebits exceeds max_framebits (or is odd) is skipped: its output bytes stay
 * untouched.  Tables of 16 or more frames
 * are length-sorted on the device into an internal copy first (longest frame
 * first: a wavefront decodes four consecutive descriptors and runs as long as
 * the longest); d_desc itself is never modified and the order changes no
 * output byte.  Frames longer than 778 bits use a per-thread HBM scratch
 * buffer for their decision history (grown on demand). */
int vit_decode_varlen_dev(const uint8_t *d_symbols_u8, uint8_t *d_decoded,
                          const vit_frame_desc *d_desc, int64_t nframes,
                          uint32_t max_framebits, void *stream);
/* Same, for a table the caller does not trust: sym_bytes / out_bytes are the sizes of the two buffers, and a
 * descriptor whose 4*(framebits+6) symbol bytes or (framebits+7)/8 output bytes would lie (even partly) outside them
 * is skipped like the other invalid ones (checked on the device, in a copy of the table; nothing is read or written
 * for it).  vit_decode_varlen_dev itself takes no sizes: there a bad offset is an out-of-bounds device access. */
int vit_decode_varlen_dev_checked(const uint8_t *d_symbols_u8, uint64_t sym_bytes, uint8_t *d_decoded,
                                  uint64_t out_bytes, const vit_frame_desc *d_desc, int64_t nframes,
                                  uint32_t max_framebits, void *stream);
/* Host helper: reorder a HOST array of descriptors by framebits (longest first, stable) before
 * uploading it.  Optional since the device-side sort above; kept for callers that build tables of
 * fewer than 16 frames or want a deterministic order.  Every descriptor carries its own offsets,
 * so the order does not change any output byte. */
void vit_sort_descs(vit_frame_desc *h_desc, int64_t nframes);
/* u32 -> u8 narrowing of nsym symbols on the device (ingest stage). */
int vit_pack_symbols_dev(const uint32_t *d_symbols_u32, uint8_t *d_symbols_u8,
                         int64_t nsym, void *stream);

/* Host-buffer convenience: H2D, decode, D2H, synchronous. */
int vit_decode_batch_host(const uint8_t *h_symbols_u8, uint8_t *h_decoded,
                          uint32_t framebits, int64_t nframes);

/* Batched RScheckSuperframe: nsf superframes of 120*RSDims bytes each (device),
 * outputs 110*RSDims bytes each; d_ret[s] receives what RScheckSuperframe
 * would return for superframe s.  Output columns at and after the first
 * uncorrectable column of a superframe are left untouched. */
int vit_rs_batch_dev(const uint8_t *d_p, uint8_t *d_out, int32_t *d_ret,
                     uint32_t RSDims, int64_t nsf, void *stream);
int vit_rs_batch_host(const uint8_t *h_p, uint8_t *h_out, int32_t *h_ret,
                      uint32_t RSDims, int64_t nsf);

/* DAB+ superframe path: nsf superframes, each = 5 consecutive frames of
 * framebits = 192*RSDims bits in d_symbols_u8.  Decodes the 5*nsf frames into d_work
 * (nsf*120*RSDims bytes, device) -- five decoded frames ARE the RS input block p[j + k*RSDims] --
 * and runs the batched RScheckSuperframe on it.  Both kernels are enqueued on `stream`. */
int vit_dabplus_superframes_dev(const uint8_t *d_symbols_u8, uint8_t *d_work, uint8_t *d_rs_out,
                                int32_t *d_ret, uint32_t RSDims, int64_t nsf, void *stream);

/* Ingest stage for concurrent callers of deconvolve() (the reference is re-entrant and QIRX calls it from several
 * threads, README.md:56).  `microseconds` = 0 switches it off: every call is a launch of its own on its thread's stream.
 * With a window > 0, a call that finds at least `min_callers` deconvolve() calls in flight (itself included)
 * claims a slot of one mapped pinned ring, copies its own symbols into it (narrowed to one byte each) and joins the
 * open batch; the batch's first caller holds it open while `launches_in_flight` earlier batches are still on the
 * GPU - never longer than the window - and then issues ONE launch for all members; each workgroup publishes its
 * slot's completion word, on which the caller spins.  A lone caller finds a free launch credit and is launched at
 * once, so nobody waits for callers that do not exist.  No worker thread, no copy by anyone but the caller itself.
 * Environment (read when the library is loaded, for hosts that only bind the five reference exports):
 * VITERBI_AMD_BATCH_WINDOW_US, VITERBI_AMD_BATCH_MIN_CALLERS, VITERBI_AMD_BATCH_DEPTH, VITERBI_AMD_SPIN_CPUS.
 * Waiting: the batch's first caller polls the completion words; the others spin on their own word while the calls in
 * flight do not exceed `cpus` (default: the process's CPU budget - affinity mask capped by a cgroup CPU quota) and
 * otherwise sleep on a futex until the polling member wakes them (32 spinning callers in a 16-CPU container get the
 * whole process throttled).  vit_set_batch_spin_cpus(0): always sleep.
 * All setters return the previous value. */
int vit_set_batch_window_us(int microseconds);
int vit_set_batch_min_callers(int min_callers);
int vit_set_batch_depth(int launches_in_flight);
int vit_set_batch_spin_cpus(int cpus);

/* Kernel selection (the analogue of the reference's dispatcher, setupdll.cpp:195-270):
 *   0 = auto: launches of up to 2048 frames (they cannot fill the chip) take the latency kernel - one
 *       frame per wavefront, ~20 us per FIC frame -, larger ones the packed throughput kernel;
 *   1 = wave-per-frame cross-check kernel, 2 = packed 4-frames-per-wave kernel, 3 = latency kernel,
 *   4 = packed 8-frames-per-wave kernel (frames <= 778 bits; an experiment kept for comparison: fewer instructions per
 *       frame, slower - see csrc/vit_pk8.hip).
 * Returns the old value.  Affects later vit_decode_* / deconvolve calls of the whole process. */
int vit_set_kernel(int which);

/* Renormalisation comparator of the decoder (process-wide, affects later vit_decode_* / deconvolve calls).
 * The reference exists in two build configurations that differ in ONE comparison on the hot path:
 *   1 (default): renormalise when the metric of state 0 is >= 150 -- the MASM decoders, decon_avx2.asm:97,114
 *                `cmp sil,150 ; jb mainloop` (also decon_avx.asm:142, decon_ssse3.asm:163,
 *                decon_sse2_lut32.asm:173; configuration Rel_asm, the one the reference's README tells users to
 *                build (README.md:50-52): what an installed viterbi.dll runs);
 *   0          : renormalise when it is                    >  150 -- the C decoders, deconvolve.cpp:399,408
 *                (configuration Rel_cpp, the one that can be compiled and run outside Windows; bench.py selects it
 *                because its timed CPU baseline, this repo's AVX2 port, implements it).
 * The two give identical output on soft-decision input at any usable SNR and DIFFERENT output on hard-decision
 * (0/255) input from a poor channel, where path metrics reach the 0 and 255 clamps (tests/test_gpu_parity.py:
 * test_renorm_ge_mode).  Environment variable VITERBI_AMD_RENORM_GE=0/1 selects the mode at start-up for a host that
 * only binds the five reference exports.  Returns the previous value.  (Until round 3 the default was 0.) */
int vit_set_renorm_ge(int on);

/* ------------------------------------------------------------------------ *
 * Part 3 -- several GPUs behind one call (BASELINE.json configs[3], SURVEY 8e;
 * not in the reference, whose only concurrency is caller threads, README.md:56)
 * ------------------------------------------------------------------------ */

/* ONE host process, ndev gfx950 devices (HIP ordinals in devices[], distinct; devices[0] is the "root").
 * nframes equal-length frames in the device format live on the root (d_symbols_u8), the decoded bytes
 * are wanted there too (d_decoded).  The stream is cut into chunks of
 *     root_frames + (ndev-1) * chunk_frames
 * consecutive frames; of every chunk the root decodes the first root_frames itself and devices[i]
 * (i >= 1) the i-th block of chunk_frames frames: round-robin at block granularity, so each block is
 * a contiguous slice that RCCL sends from / receives into place (ncclSend/ncclRecv over xGMI, one
 * ncclGroupStart/End per pipeline step, librccl dlopen'ed on first use).  Block k+1 travels while
 * block k is being decoded and the decoded bytes of block k-1 come back (double buffers per device).
 *   root_frames : -1 = chunk_frames; 0 = the root only distributes (ndev > 1); larger values give the
 *                 root a bigger share (its peers are fed through one xGMI link each)
 *   stream      : the root-device stream that produced d_symbols_u8 (NULL = default stream); the
 *                 transfers start behind it
 *   flags       : VIT_MULTI_LOOPBACK adds one more rank ON THE ROOT DEVICE that is fed through RCCL
 *                 like a remote peer (self send/recv) - a self-test of the pipeline on a one-GPU box
 * SYNCHRONOUS: returns when every byte of d_decoded is in place.  One call at a time per process
 * (internal mutex); streams, communicators and buffers are cached between calls with the same devices.
 * The caller's current device is restored.  ndev == 1 without the flag needs no RCCL. */
#define VIT_MULTI_LOOPBACK 0x1u
int vit_decode_stream_multi(const uint8_t *d_symbols_u8, uint8_t *d_decoded, uint32_t framebits,
                            int64_t nframes, const int *devices, int ndev, int64_t chunk_frames,
                            int64_t root_frames, unsigned flags, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VITERBI_AMD_H */
