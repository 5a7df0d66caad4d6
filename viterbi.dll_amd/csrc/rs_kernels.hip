// rs_kernels.hip -- batched RScheckSuperframe for gfx950: RS(120,110) over
// GF(2^8)/0x11D (RS(255,245) shortened by 135, roots alpha^0..alpha^9).
//
// One lane decodes one column (codeword); a 256-thread workgroup takes whole
// superframes so the reference's "stop at the first uncorrectable column"
// rule (rschecksf.cpp:80-88) is a workgroup-local min-reduction.  Codeword
// bytes live in LDS transposed ([row][lane]) so the strided column gather of
// rschecksf.cpp:75-76 is a coalesced byte load across lanes.
//
// Replaces, from scratch: RScheckSuperframe (rschecksf.cpp:65-93), DECODE_RS
// (:199-377), Mod255 (:50-52) and CreateLookupTables (dllmain.cpp:124-146).
#include "vit_internal.h"

namespace {

constexpr int NN = 255;      // viterbi.h:95
constexpr int NROOTS = 10;   // viterbi.h:97
constexpr int PADN = 135;    // rschecksf.cpp:45
constexpr int NCW = 120, NMSG = 110;
constexpr int RS_THREADS = 256;

struct GfTables {
    uint8_t ato[768];  // alpha_to[i % 255], viterbi.h:101-105
    uint8_t iof[256];  // index_of, index_of[0] = 255
};
constexpr GfTables make_tables() {
    GfTables t{};
    uint8_t alpha[256] = {};
    int sr = 1;
    t.iof[0] = NN;
    for (int i = 0; i < NN; i++) {
        t.iof[sr] = (uint8_t)i;
        alpha[i] = (uint8_t)sr;
        sr <<= 1;
        if (sr & 256) sr ^= 285;  // c_gfpoly, viterbi.h:96
        sr &= NN;
    }
    for (int i = 0; i < 768; i++) t.ato[i] = alpha[i % 255];
    return t;
}
__constant__ GfTables g_gf = make_tables();

__device__ __forceinline__ uint32_t mod255(uint32_t x) { return (x * 0x1010102u) >> 24; }

// Decode the codeword stored at col[k * RS_THREADS], k = 0..119 (LDS, transposed).
// Returns root count, 0 when clean, -1 when uncorrectable; patches in place.
__device__ int decode_rs(uint8_t* col, const uint8_t* __restrict__ ato, const uint8_t* __restrict__ iof) {
    uint8_t s[16], lambda[16], b[16], root[16];
    const uint32_t d0 = col[0];
#pragma unroll
    for (int i = 0; i < 16; i++) { s[i] = (uint8_t)d0; root[i] = 0; }
    for (int j = 1; j < NCW; j++) {  // syndromes, Horner (rschecksf.cpp:212-219)
        const uint32_t d = col[j * RS_THREADS];
#pragma unroll
        for (int i = 0; i < NROOTS; i++) s[i] = s[i] == 0 ? (uint8_t)d : (uint8_t)(d ^ ato[iof[s[i]] + i]);
    }
    uint32_t syn = 0;
#pragma unroll
    for (int i = 0; i < NROOTS; i++) syn |= s[i];
    if (!syn) return 0;
#pragma unroll
    for (int i = 0; i <= NROOTS; i++) s[i] = iof[s[i]];
#pragma unroll
    for (int i = 0; i < 16; i++) { b[i] = 0xFF; lambda[i] = 0; }
    b[0] = 0;
    lambda[0] = 1;

    int el = 0;
    for (int r = 1; r <= NROOTS; r++) {  // Berlekamp-Massey (rschecksf.cpp:240-284)
        uint32_t discr = 0;
        for (int i = 0; i < r; i++)
            if (lambda[i] != 0 && s[r - i - 1] != NN) discr ^= ato[iof[lambda[i]] + s[r - i - 1]];
        discr = iof[discr];
        if (discr == NN) {
            for (int i = 15; i > 0; i--) b[i] = b[i - 1];
            b[0] = NN;
        } else {
            root[0] = lambda[0];
            for (int i = 0; i < NROOTS; i++) {
                root[i + 1] = lambda[i + 1];
                if (b[i] != NN) root[i + 1] ^= ato[discr + b[i]];
            }
            if (2 * el <= r - 1) {
                el = r - el;
                for (int i = 0; i <= NROOTS; i++)
                    b[i] = lambda[i] == 0 ? (uint8_t)NN : (uint8_t)mod255(iof[lambda[i]] - discr + NN);
            } else {
                for (int i = 15; i > 0; i--) b[i] = b[i - 1];
                b[0] = NN;
            }
            for (int i = 0; i < 16; i++) lambda[i] = root[i];
        }
    }
    int deg_lambda = 0;
    for (int i = 0; i < NROOTS + 1; i++) {
        lambda[i] = iof[lambda[i]];
        if (lambda[i] != NN) deg_lambda = i;
    }
    for (int i = 0; i < 16; i++) b[i] = lambda[i];
    int count = 0;
    for (int i = 1; i <= NN; i++) {  // Chien search (rschecksf.cpp:299-320)
        uint32_t q = 1;
        for (int j = deg_lambda; j > 0; j--)
            if (b[j] != NN) {
                b[j] = (uint8_t)mod255(b[j] + j);
                q ^= ato[b[j]];
            }
        if (q != 0) continue;
        root[count] = (uint8_t)i;
        if (++count == deg_lambda) break;
    }
    if (deg_lambda != count) return -1;

    const int deg_omega = deg_lambda - 1;
    for (int i = 0; i <= deg_omega; i++) {  // omega (rschecksf.cpp:331-341)
        uint32_t tmp = 0;
        for (int j = i; j >= 0; j--)
            if (s[i - j] != NN && lambda[j] != NN) tmp ^= ato[s[i - j] + lambda[j]];
        b[i] = iof[tmp];
    }
    for (int j = count - 1; j >= 0; j--) {  // Forney (rschecksf.cpp:346-374)
        const uint32_t rt = root[j];
        if (rt < PADN + 1) continue;  // error in the virtual padding: skipped, still counted
        uint32_t num1 = 0;
        for (int i = deg_omega; i >= 0; i--)
            if (b[i] != NN) num1 ^= ato[mod255(b[i] + i * rt)];
        if (!num1) continue;
        const uint32_t num2 = ato[NN - rt];
        uint32_t den = 0;
        const int top = deg_lambda < NROOTS - 1 ? deg_lambda : NROOTS - 1;
        for (int i = top & ~1; i >= 0; i -= 2)
            if (lambda[i + 1] != NN) den ^= ato[mod255(lambda[i + 1] + i * rt)];
        const uint32_t tmp = (uint32_t)iof[num1] + iof[num2] + (NN - iof[den]);  // <= 763 < 768
        col[(rt - 1 - PADN) * RS_THREADS] ^= ato[tmp];
    }
    return count;
}

// Workgroup = 256 lanes.  For rsdims <= 256 it takes spb = 256/rsdims superframes
// per pass; for wider superframes it walks the columns in 256-wide chunks, in
// order, and stops after the first chunk that holds a failure.
__global__ __launch_bounds__(RS_THREADS) void rs_kernel(const uint8_t* __restrict__ p, uint8_t* __restrict__ out,
                                                        int32_t* __restrict__ ret, uint32_t rsdims,
                                                        long long nsf) {
    __shared__ uint8_t cw[NCW * RS_THREADS];  // [row][lane]
    __shared__ uint8_t ato[768];
    __shared__ uint8_t iof[256];
    __shared__ int s_minfail[RS_THREADS];
    __shared__ int s_sum[RS_THREADS];
    __shared__ int s_fail[RS_THREADS];
    const int tid = threadIdx.x;
    for (int i = tid; i < 768; i += RS_THREADS) ato[i] = g_gf.ato[i];
    iof[tid] = g_gf.iof[tid];
    __syncthreads();

    const uint32_t spb = rsdims <= RS_THREADS ? RS_THREADS / rsdims : 1u;  // superframes per pass
    const uint32_t nchunk = rsdims <= RS_THREADS ? 1u : (rsdims + RS_THREADS - 1) / RS_THREADS;
    const long long ngroups = (nsf + spb - 1) / spb;
    const size_t in_sz = (size_t)NCW * rsdims, out_sz = (size_t)NMSG * rsdims;

    for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const uint32_t lsf = rsdims <= RS_THREADS ? tid / rsdims : 0u;  // local superframe of this lane
        const uint32_t lsfc = lsf < spb ? lsf : 0u;
        const long long sf = g * spb + lsf;
        if (tid < (int)spb) {
            s_sum[tid] = 0;
            s_fail[tid] = 0;
        }
        __syncthreads();
        for (uint32_t ch = 0; ch < nchunk; ch++) {
            if (tid < (int)spb) s_minfail[tid] = 0x7FFFFFFF;
            __syncthreads();
            const uint32_t colidx = rsdims <= RS_THREADS ? tid - lsf * rsdims : ch * RS_THREADS + tid;
            const bool active = lsf < spb && sf < nsf && colidx < rsdims && !s_fail[lsfc];
            int res = 0;
            if (active) {
                const uint8_t* src = p + (size_t)sf * in_sz + colidx;
                for (int k = 0; k < NCW; k++) cw[k * RS_THREADS + tid] = src[(size_t)k * rsdims];
                res = decode_rs(&cw[tid], ato, iof);
                if (res < 0) atomicMin(&s_minfail[lsf], (int)colidx);
            }
            __syncthreads();
            const int mf = s_minfail[lsfc];
            if (active && (int)colidx < mf) {  // columns before the first failure are written
                uint8_t* dst = out + (size_t)sf * out_sz + colidx;
                for (int k = 0; k < NMSG; k++) dst[(size_t)k * rsdims] = cw[k * RS_THREADS + tid];
                atomicAdd(&s_sum[lsf], res);
            }
            __syncthreads();
            if (tid < (int)spb && s_minfail[tid] != 0x7FFFFFFF) s_fail[tid] = 1;
            __syncthreads();
        }
        if (tid < (int)spb && g * spb + tid < nsf) ret[g * spb + tid] = s_fail[tid] ? -1 : s_sum[tid];
        __syncthreads();
    }
}

}  // namespace

hipError_t rs_launch(const uint8_t* d_p, uint8_t* d_out, int32_t* d_ret, uint32_t rsdims, int64_t nsf,
                     hipStream_t stream) {
    if (nsf <= 0 || rsdims == 0) return hipSuccess;
    const uint32_t spb = rsdims <= RS_THREADS ? RS_THREADS / rsdims : 1u;
    long long groups = (nsf + spb - 1) / spb;
    if (groups > (1 << 20)) groups = 1 << 20;
    hipLaunchKernelGGL(rs_kernel, dim3((unsigned)groups), dim3(RS_THREADS), 0, stream, d_p, d_out, d_ret, rsdims,
                       (long long)nsf);
    return hipGetLastError();
}
