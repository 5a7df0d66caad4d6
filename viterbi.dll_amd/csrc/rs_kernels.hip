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
// Every small polynomial array is indexed with compile-time constants only (loops fully
// unrolled, data-dependent bounds turned into predicates): dynamically indexed register arrays
// cost a v_cndmask chain per access on this target and made the error path ~8x slower.
__device__ int decode_rs(uint8_t* col, const uint8_t* __restrict__ ato, const uint8_t* __restrict__ iof,
                         const uint8_t* __restrict__ mulp) {
    uint32_t s[NROOTS];
    const uint32_t d0 = col[0];
#pragma unroll
    for (int i = 0; i < NROOTS; i++) s[i] = d0;
    for (int j = 1; j < NCW; j++) {  // syndromes, Horner (rschecksf.cpp:212-219)
        const uint32_t d = col[j * RS_THREADS];
        // s*alpha^i from a per-root product table (mulp[i][x] = x ? alpha_to[index_of[x]+i] : 0):
        // one LDS byte per multiply-add instead of the log + antilog pair, same field arithmetic
        s[0] ^= d;
#pragma unroll
        for (int i = 1; i < NROOTS; i++) s[i] = d ^ mulp[i * 256 + s[i]];
    }
    uint32_t syn = 0;
#pragma unroll
    for (int i = 0; i < NROOTS; i++) syn |= s[i];
    if (!syn) return 0;
#pragma unroll
    for (int i = 0; i < NROOTS; i++) s[i] = iof[s[i]];  // index form (s[10] of the reference is never used)

    // lambda / b: only entries 0..10 of the reference's 16-byte vectors are ever read
    uint32_t lam[NROOTS + 1], b[NROOTS + 1];
#pragma unroll
    for (int i = 0; i <= NROOTS; i++) { lam[i] = 0; b[i] = NN; }
    lam[0] = 1;
    b[0] = 0;
    int el = 0;
#pragma unroll
    for (int r = 1; r <= NROOTS; r++) {  // Berlekamp-Massey (rschecksf.cpp:240-284)
        uint32_t discr = 0;
#pragma unroll
        for (int i = 0; i < r; i++)
            if (lam[i] != 0 && s[r - i - 1] != NN) discr ^= ato[iof[lam[i]] + s[r - i - 1]];
        discr = iof[discr];
        const bool zero = discr == NN;
        const bool grow = !zero && 2 * el <= r - 1;
        uint32_t t[NROOTS + 1];
        t[0] = lam[0];
#pragma unroll
        for (int i = 0; i < NROOTS; i++) {
            t[i + 1] = lam[i + 1];
            if (!zero && b[i] != NN) t[i + 1] ^= ato[discr + b[i]];
        }
        if (grow) el = r - el;
        // b <- inv(discr) * lambda (grow) or x * b (otherwise: _mm_slli_si128(b,1), b[0] = 255)
#pragma unroll
        for (int i = NROOTS; i >= 0; i--) {
            const uint32_t scaled = lam[i] == 0 ? (uint32_t)NN : mod255(iof[lam[i]] - discr + NN);
            const uint32_t shifted = i ? b[i - 1] : (uint32_t)NN;
            b[i] = grow ? scaled : shifted;
        }
#pragma unroll
        for (int i = 0; i <= NROOTS; i++) lam[i] = zero ? lam[i] : t[i];
    }
    int deg_lambda = 0;
#pragma unroll
    for (int i = 0; i <= NROOTS; i++) {
        lam[i] = iof[lam[i]];
        if (lam[i] != NN) deg_lambda = i;
    }
    // Chien search (rschecksf.cpp:299-320): the reference advances the logs b[j] += j and sums
    // alpha_to[b[j]]; here the same terms lambda_j * alpha^(j*i) are kept in polynomial form and
    // advanced with the product tables (one LDS byte per term, no mod-255 arithmetic).
    uint32_t c[NROOTS + 1];
#pragma unroll
    for (int i = 0; i <= NROOTS; i++) c[i] = lam[i] == NN ? 0u : (uint32_t)ato[lam[i]];
    uint32_t root[NROOTS];
#pragma unroll
    for (int k = 0; k < NROOTS; k++) root[k] = 0;
    int count = 0;
    bool searching = true;
    for (int i = 1; i <= NN; i++) {
        if (searching) {
            uint32_t q = 1;  // lambda[0] is always 1
#pragma unroll
            for (int j = NROOTS; j > 0; j--) {
                c[j] = mulp[j * 256 + c[j]];
                q ^= c[j];
            }
            if (q == 0) {
#pragma unroll
                for (int k = 0; k < NROOTS; k++)
                    if (count == k) root[k] = (uint32_t)i;
                if (++count == deg_lambda) searching = false;
            }
        }
        if (!__any(searching)) break;
    }
    if (deg_lambda != count) return -1;

    const int deg_omega = deg_lambda - 1;
    uint32_t om[NROOTS];
#pragma unroll
    for (int i = 0; i < NROOTS; i++) {  // omega (rschecksf.cpp:331-341)
        uint32_t tmp = 0;
#pragma unroll
        for (int j = 0; j <= i; j++)
            if (s[i - j] != NN && lam[j] != NN) tmp ^= ato[s[i - j] + lam[j]];
        om[i] = i <= deg_omega ? (uint32_t)iof[tmp] : (uint32_t)NN;
    }
    const int top = (deg_lambda < NROOTS - 1 ? deg_lambda : NROOTS - 1) & ~1;
#pragma unroll
    for (int j = NROOTS - 1; j >= 0; j--) {  // Forney (rschecksf.cpp:346-374)
        const uint32_t rt = root[j];
        if (j < count && rt >= PADN + 1) {  // roots in the virtual padding are skipped, still counted
            uint32_t num1 = 0;
#pragma unroll
            for (int i = 0; i < NROOTS; i++)
                if (i <= deg_omega && om[i] != NN) num1 ^= ato[mod255(om[i] + i * rt)];
            if (num1) {
                const uint32_t num2 = ato[NN - rt];
                uint32_t den = 0;
#pragma unroll
                for (int i = 0; i < NROOTS; i += 2)
                    if (i <= top && lam[i + 1] != NN) den ^= ato[mod255(lam[i + 1] + i * rt)];
                const uint32_t tmp = (uint32_t)iof[num1] + iof[num2] + (NN - iof[den]);  // <= 763 < 768
                col[(rt - 1 - PADN) * RS_THREADS] ^= ato[tmp];
            }
        }
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
    __shared__ uint8_t mulp[(NROOTS + 1) * 256];  // mulp[i][x] = x * alpha^i, i = 0..10
    __shared__ int s_minfail[RS_THREADS];
    __shared__ int s_sum[RS_THREADS];
    __shared__ int s_fail[RS_THREADS];
    const int tid = threadIdx.x;
    for (int i = tid; i < 768; i += RS_THREADS) ato[i] = g_gf.ato[i];
    iof[tid] = g_gf.iof[tid];
    for (int i = 0; i <= NROOTS; i++) mulp[i * 256 + tid] = tid ? g_gf.ato[g_gf.iof[tid] + i] : 0;
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
                res = decode_rs(&cw[tid], ato, iof, mulp);
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
