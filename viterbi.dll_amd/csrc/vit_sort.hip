// vit_sort.hip -- device-side ordering of a variable-length descriptor table, longest frame first.
//
// The packed decoder gives four consecutive descriptors to one wavefront, which then runs as long
// as the longest of them; with persistent workgroups the table is also consumed front to back.
// A length-sorted copy of the table (counting sort on framebits/8, 1153 bins) therefore makes a
// mixed batch (BASELINE config 3: 288..6912 bits) run like a uniform one.  The order is not
// stable - it does not have to be: every descriptor carries its own symbol/output offsets, so no
// output byte depends on the order.  Descriptors the launch was not sized for (framebits above
// max_framebits or odd) sort to the end and are skipped by the decoder as before.
//
// Three small kernels on the caller's stream: histogram, descending exclusive scan, scatter.
// Histogram and scatter privatise the bins in LDS (one global atomic per workgroup and length),
// so a batch of identical lengths does not serialise on one L2 atomic.
#include "vit_internal.h"

namespace {

typedef uint32_t u32;
constexpr u32 BINS = VIT_SORT_BINS;  // framebits/8 = 0..1152
constexpr u32 TPB = 1024;

__device__ __forceinline__ u32 key_of(const vit_frame_desc& d, u32 maxfb) {
    const u32 fb = d.framebits;
    return (fb <= maxfb && (fb & 1u) == 0) ? ((fb + 7u) >> 3) : 0u;
}

__global__ __launch_bounds__(TPB) void desc_hist_kernel(const vit_frame_desc* __restrict__ desc, long long n, u32 maxfb,
                                                        unsigned* __restrict__ hist) {
    __shared__ unsigned cnt[BINS];
    for (u32 k = threadIdx.x; k < BINS; k += TPB) cnt[k] = 0;
    __syncthreads();
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i < n) atomicAdd(&cnt[key_of(desc[i], maxfb)], 1u);
    __syncthreads();
    for (u32 k = threadIdx.x; k < BINS; k += TPB)
        if (cnt[k]) atomicAdd(&hist[k], cnt[k]);
}

// hist[k] = count of key k  ->  start[k] = number of descriptors with a LARGER key (descending order)
// It also leaves the histogram zeroed for the next sort and clears the 256-byte header in front of it (the persistent kernel's group
// counter): two memset dispatches less per call (profiles/r04_ab_long_inflight.txt, section 22).
__global__ __launch_bounds__(256) void desc_scan_kernel(unsigned* __restrict__ hist, unsigned* __restrict__ start, unsigned* __restrict__ hdr) {
    constexpr u32 PER = (BINS + 255u) / 256u;  // 5 bins per thread, taken from the top
    __shared__ unsigned part[256];
    const u32 t = threadIdx.x;
    unsigned c[PER], sum = 0;
#pragma unroll
    for (u32 j = 0; j < PER; j++) {
        const u32 r = t * PER + j;  // rank from the top: bin BINS-1-r
        c[j] = r < BINS ? hist[BINS - 1u - r] : 0u;
        if (r < BINS) hist[BINS - 1u - r] = 0u;
        sum += c[j];
    }
    if (hdr && t < 64u) hdr[t] = 0u;
    part[t] = sum;
    __syncthreads();
    for (u32 d = 1; d < 256u; d <<= 1) {  // inclusive Hillis-Steele scan over the thread sums
        const unsigned v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned run = part[t] - sum;  // exclusive
#pragma unroll
    for (u32 j = 0; j < PER; j++) {
        const u32 r = t * PER + j;
        if (r < BINS) start[BINS - 1u - r] = run;
        run += c[j];
    }
}

__global__ __launch_bounds__(TPB) void desc_scatter_kernel(const vit_frame_desc* __restrict__ desc, long long n, u32 maxfb,
                                                           unsigned* __restrict__ cursor,
                                                           vit_frame_desc* __restrict__ sorted) {
    __shared__ unsigned cnt[BINS];  // per-workgroup count, then the workgroup's base position per key
    for (u32 k = threadIdx.x; k < BINS; k += TPB) cnt[k] = 0;
    __syncthreads();
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    vit_frame_desc d;
    u32 key = 0, rank = 0;
    if (i < n) {
        d = desc[i];
        key = key_of(d, maxfb);
        rank = atomicAdd(&cnt[key], 1u);
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < BINS; k += TPB)
        if (cnt[k]) cnt[k] = atomicAdd(&cursor[k], cnt[k]);
    __syncthreads();
    if (i < n) sorted[cnt[key] + rank] = d;
}

// Bounds check of a descriptor table against the sizes of the caller's two buffers (vit_decode_varlen_dev_checked):
// the copy keeps every descriptor whose symbols and output bytes lie inside them and gives the others a length no
// launch is sized for (0xFFFFFFFF), which every decoder kernel skips and the sort puts last.
__global__ __launch_bounds__(256) void desc_check_kernel(const vit_frame_desc* __restrict__ desc, long long n,
                                                         unsigned long long sym_bytes, unsigned long long out_bytes,
                                                         vit_frame_desc* __restrict__ checked) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    vit_frame_desc d = desc[i];
    if (d.framebits <= VIT_MAX_FRAMEBITS) {
        const unsigned long long need_in = 4ull * (d.framebits + VIT_TAIL), need_out = (d.framebits + 7u) >> 3;
        const bool inside = d.sym_offset <= sym_bytes && need_in <= sym_bytes - d.sym_offset &&
                            d.out_offset <= out_bytes && need_out <= out_bytes - d.out_offset;
        if (!inside) d.framebits = 0xFFFFFFFFu;
    }
    checked[i] = d;
}

}  // namespace

hipError_t vit_check_descs_launch(const vit_frame_desc* d_desc, vit_frame_desc* d_checked, int64_t nframes,
                                  uint64_t sym_bytes, uint64_t out_bytes, hipStream_t stream) {
    if (nframes <= 0) return hipSuccess;
    hipLaunchKernelGGL(desc_check_kernel, dim3((unsigned)((nframes + 255) / 256)), dim3(256), 0, stream, d_desc,
                       (long long)nframes, (unsigned long long)sym_bytes, (unsigned long long)out_bytes, d_checked);
    return hipGetLastError();
}

hipError_t vit_sort_descs_launch(const vit_frame_desc* d_desc, vit_frame_desc* d_sorted, int64_t nframes,
                                 uint32_t max_framebits, unsigned* d_bins, hipStream_t stream, bool* bins_clean, unsigned* d_hdr) {
    if (nframes <= 0) return hipSuccess;
    hipError_t e;
    if (!bins_clean || !*bins_clean) {  // first sort in this buffer (or after a failed one): the histogram is not known to be zero
        if ((e = hipMemsetAsync(d_bins, 0, BINS * sizeof(unsigned), stream)) != hipSuccess) return e;
    }
    if (bins_clean) *bins_clean = false;
    const unsigned blocks = (unsigned)((nframes + TPB - 1) / TPB);
    hipLaunchKernelGGL(desc_hist_kernel, dim3(blocks), dim3(TPB), 0, stream, d_desc, (long long)nframes, max_framebits,
                       d_bins);
    hipLaunchKernelGGL(desc_scan_kernel, dim3(1), dim3(256), 0, stream, d_bins, d_bins + BINS, d_hdr);
    if (bins_clean && hipPeekAtLastError() == hipSuccess) *bins_clean = true;  // the scan leaves the histogram zeroed
    hipLaunchKernelGGL(desc_scatter_kernel, dim3(blocks), dim3(TPB), 0, stream, d_desc, (long long)nframes,
                       max_framebits, d_bins + BINS, d_sorted);
    return hipGetLastError();
}
