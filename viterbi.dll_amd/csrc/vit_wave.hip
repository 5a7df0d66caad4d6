// vit_wave.hip -- wave-per-frame K=7 r=1/4 Viterbi decoder for gfx950.
//
// One 64-lane wavefront owns one frame; lane s holds the 8-bit path metric of
// trellis state s in a VGPR.  This is the general kernel (any even framebits
// <= 9216, one workgroup = one wave, decisions in LDS) and the in-tree
// cross-check for the packed kernel in vit_pk.hip, which is the fast path.
//
// Replaces, from scratch: decon_avx2 / Butterfly256 (deconvolve.cpp:334-387,
// 514-526), Renormalize256 (:407-412), ChainBack (:416-435) and the constant
// block const.asm:19-63 (the masks are recomputed from the polynomials).
#include <mutex>

#include "vit_internal.h"

namespace {

__device__ __forceinline__ uint32_t par32(uint32_t x) { return __builtin_popcount(x) & 1u; }

// v_lerp_u8 with S2 = 0x01010101 is pavgb on four packed bytes:
// D.u8[i] = (S0.u8[i] + S1.u8[i] + 1) >> 1   (deconvolve.cpp:338-351 vpavgb)
__device__ __forceinline__ uint32_t avg4(uint32_t a, uint32_t b) {
    return __builtin_amdgcn_lerp(a, b, 0x01010101u);
}

__global__ __launch_bounds__(64) void vit_wave_kernel(const uint8_t* __restrict__ sym,
                                                      uint8_t* __restrict__ out,
                                                      const vit_frame_desc* __restrict__ desc,
                                                      uint32_t framebits_uniform, uint32_t max_framebits,
                                                      long long nframes, uint32_t renorm_thr) {
    extern __shared__ unsigned long long dec[];  // one 64-bit decision word per step
    const uint32_t lane = threadIdx.x;
    // Branch mask of this lane's butterfly i = lane>>1: byte j = 0xFF iff
    // parity((2i) & poly_j), polys {109,79,83,109} (const.asm:27-63).
    const uint32_t two_i = lane & ~1u;
    const uint32_t mask = (par32(two_i & 109u) ? 0x000000FFu : 0u) | (par32(two_i & 79u) ? 0x0000FF00u : 0u) |
                          (par32(two_i & 83u) ? 0x00FF0000u : 0u) | (par32(two_i & 109u) ? 0xFF000000u : 0u);
    const bool odd = lane & 1u;
    const int src_a = lane >> 1, src_b = (lane >> 1) + 32;

    for (long long f = blockIdx.x; f < nframes; f += gridDim.x) {
        uint32_t framebits = framebits_uniform;
        size_t sym_off, out_off;
        if (desc) {
            framebits = desc[f].framebits;
            sym_off = desc[f].sym_offset;
            out_off = desc[f].out_offset;
            // the launch's LDS was not sized for it, an odd length (not a valid frame) or symbols that are not
            // dword aligned: skipped, output untouched
            if (framebits > max_framebits || (framebits & 1u) || (sym_off & 3u)) continue;
        } else {
            sym_off = (size_t)f * 4u * (framebits + VIT_TAIL);
            out_off = (size_t)f * ((framebits + 7u) >> 3);
        }
        const uint32_t T = ((framebits + VIT_TAIL) >> 1) << 1;  // deconvolve.cpp:126: 2 steps per iteration
        const uint32_t* s32 = reinterpret_cast<const uint32_t*>(sym + sym_off);

        uint32_t m = lane == 0 ? 0u : 63u;  // const.asm:19-25
        for (uint32_t t = 0; t < T; ++t) {
            const uint32_t s = s32[t];  // 4 soft symbols of this step, wave-uniform
            const uint32_t x = s ^ mask;
            const uint32_t p = avg4(x, x >> 8);        // byte0 = avg(x0,x1), byte2 = avg(x2,x3)
            const uint32_t q = avg4(p, p >> 16);       // byte0 = avg(avg01, avg23)
            const uint32_t metric = (q & 0xFFu) >> 2;  // psrlw 2 + pand 63
            const uint32_t bm_a = odd ? 63u - metric : metric;
            const uint32_t bm_b = 63u - bm_a;
            const uint32_t a = __shfl(m, src_a), b = __shfl(m, src_b);
            const uint32_t ca = min(a + bm_a, 255u), cb = min(b + bm_b, 255u);  // paddusb
            const bool d = cb <= ca;  // pminub + pcmpeqb(survivor, m1): tie -> 1
            m = d ? cb : ca;
            const unsigned long long dw = __ballot(d);
            if (lane == 0) dec[t] = dw;
            if (t & 1u) {  // Renormalize256 after every second step, state 0 only, > 150 (MASM twins: >= 150, thr = 149)
                const uint32_t m0 = __builtin_amdgcn_readfirstlane(m);
                if (m0 > renorm_thr) m = m > 63u ? m - 63u : 0u;  // psubusb
            }
        }
        __syncthreads();
        if (lane == 0) {  // ChainBack, deconvolve.cpp:416-435
            uint8_t* o = out + out_off;
            uint32_t E = 0;
            for (uint32_t n = framebits; n-- > 0;) {
                const uint32_t k = (uint32_t)(dec[n + VIT_TAIL] >> (E >> 2)) & 1u;
                E = ((E >> 1) | (k << 7)) & 0xFFu;
                if ((n & 7u) == 0) o[n >> 3] = (uint8_t)E;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void vit_pack_kernel(const uint32_t* __restrict__ in,
                                                       uint8_t* __restrict__ outp, long long nsym) {
    // four symbols per thread: u32x4 in, one packed dword out (low bytes; the
    // reference clamps with `& 0xFF`, deconvolve.cpp:158-165)
    const long long nquad = nsym >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nquad;
         i += (long long)gridDim.x * blockDim.x) {
        const uint4 v = reinterpret_cast<const uint4*>(in)[i];
        reinterpret_cast<uint32_t*>(outp)[i] =
            (v.x & 0xFFu) | ((v.y & 0xFFu) << 8) | ((v.z & 0xFFu) << 16) | (v.w << 24);
    }
    const long long tail = nquad << 2;
    if (blockIdx.x == 0 && threadIdx.x < (nsym - tail)) outp[tail + threadIdx.x] = (uint8_t)in[tail + threadIdx.x];
}

}  // namespace

hipError_t vit_launch_wave(const uint8_t* d_sym, uint8_t* d_out, const vit_frame_desc* d_desc,
                           uint32_t framebits, uint32_t max_framebits, int64_t nframes,
                           hipStream_t stream, bool renorm_ge) {
    if (nframes <= 0) return hipSuccess;
    const size_t lds = (size_t)(max_framebits + VIT_TAIL) * 8u;
    static uint64_t optin_done = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const void* ks[1] = {reinterpret_cast<const void*>(vit_wave_kernel)};
    if ((e = vit_optin_dynamic_lds(ks, 1, 80 * 1024, dev, &optin_done)) != hipSuccess) return e;
    const long long grid = nframes < (1 << 20) ? nframes : (1 << 20);
    hipLaunchKernelGGL(vit_wave_kernel, dim3((unsigned)grid), dim3(64), lds, stream, d_sym, d_out, d_desc,
                       framebits, max_framebits, (long long)nframes, renorm_ge ? 149u : 150u);
    return hipGetLastError();
}

hipError_t vit_launch_pack(const uint32_t* d_sym32, uint8_t* d_sym8, int64_t nsym, hipStream_t stream) {
    if (nsym <= 0) return hipSuccess;
    long long blocks = ((nsym >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(vit_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_sym32, d_sym8,
                       (long long)nsym);
    return hipGetLastError();
}
