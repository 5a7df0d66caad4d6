// rs_kernels.hip -- batched RScheckSuperframe for gfx950: RS(120,110) over
// GF(2^8)/0x11D (RS(255,245) shortened by 135, roots alpha^0..alpha^9).
//
// One lane decodes one column (codeword); a 256-thread workgroup takes whole
// superframes so the reference's "stop at the first uncorrectable column"
// rule (rschecksf.cpp:80-88) is a workgroup-local min-reduction.
//
// rs_kernel (RSDims <= 256, the DAB+ range): the workgroup's superframes are copied HBM -> LDS
// -> HBM with 16-byte accesses in their natural layout p[j + k*RSDims] (rschecksf.cpp:75-76: lanes
// of one superframe read consecutive bytes of a row, so the column gather needs no transposition).
// The 1190 multiply-adds of the syndrome loop (rschecksf.cpp:210-219) are replaced by the remainder
// of the codeword modulo the generator polynomial: two conflict-free 16-byte LDS lookups per data byte
// (low and high nibble of x -> x*g_0..x*g_9, see NibTable) instead of nine byte lookups; a zero remainder
// is a clean codeword, otherwise the ten syndromes are the remainder evaluated at alpha^i (same field
// elements as the reference's Horner sums, since g(alpha^i) = 0).  The correction path (rs_correct) is
// entered by a whole wavefront as soon as one of its columns has an error: Berlekamp-Massey on logs,
// locator roots in closed form (degree 1, 2), a wavefront per column (chien_wave) or four positions per
// lookup (chien_quad), Forney.  rs_kernel_wide keeps the transposed-LDS form for RSDims > 256.
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
constexpr int ATO_ZERO = 512;   // index form of a zero coefficient where a lookup must give 0: ato[512..767] = 0
constexpr int ATO_SIZE = 768;
#ifndef COOP_MAX3
#define COOP_MAX3 12  /* most columns of degree 3 and above in a wave for the column-at-a-time search chien_wave (wave's largest degree 3 / 4 or 5) */
#endif
#ifndef COOP_MAX5
#define COOP_MAX5 28  /* the stress mix of tests/tools/bench_rs.py holds 20 +- 4 per wave (and ends most superframes early), a mix without failures but 0..5 errors in every column 32 +- 4 */
#endif

struct GfTables {
    uint8_t ato[ATO_SIZE];  // alpha_to[i % 255] for i < 512 (viterbi.h:101-105), zeros from ATO_ZERO on
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
    for (int i = 0; i < ATO_ZERO; i++) t.ato[i] = alpha[i % 255];
    return t;
}
__constant__ GfTables g_gf = make_tables();

// Q[c] = a solution y of y^2 + y = c in GF(2^8) (the other one is y ^ 1), 0 when there is none (trace(c) = 1) or
// c = 0: the closed form behind the degree-2 locator roots (Chien's 255-step scan is for degrees above 2 only).
struct QuadTable {
    uint8_t q[256];
};
constexpr QuadTable make_quad_table() {
    const GfTables t = make_tables();
    QuadTable Q{};
    for (int y = 2; y < 256; y++) {
        const int y2 = t.ato[(2 * t.iof[y]) % 255];
        const int c = y2 ^ y;
        if (Q.q[c] == 0) Q.q[c] = (uint8_t)y;
    }
    return Q;
}

// Feedback table of the remainder LFSR.  Row(x) = (x*g_0, ..., x*g_9) in 16 bytes (g_0..g_7, four empty, g_8, g_9, two empty),
// g(x) = prod_{i=0..9}(x + alpha^i) (monic, degree 10).  The row is linear in x over GF(2), so it is kept as TWO
// 16-row tables, Row(x) = LO[x & 15] ^ HI[x >> 4]: 16 rows of 16 bytes are 256 B = one row per group of four LDS
// banks, so a lookup never has a bank conflict - lanes with the same nibble read the same row (broadcast), lanes with
// different nibbles different banks.  One 256-row table needs one lookup per step instead of two, but 64 random rows
// cost it ~3x the conflict-free cycles: 0.27 ms per 131072 superframes against 0.18 ms with conflict-free rows
// (measured with a fake index, profiles/r02_ab_rs_chien.txt), and that loop is LDS-bound.
#ifndef RS_LFSR2
#define RS_LFSR2 1  /* two data bytes per LFSR step: four independent lookups, half the dependent chain, 9 instead of 11
                       instructions per byte (0.195 against 0.201 ms per 131072 clean superframes, profiles/r03_ab_rs_lfsr2.txt) */
#endif
constexpr int NIB_TABLES = RS_LFSR2 ? 4 : 2;
struct NibTable {
    uint32_t w[NIB_TABLES * 16 * 4];  // LO rows, then HI rows (RS_LFSR2: then the LO and HI rows of the two-step table)
};
constexpr NibTable make_nib_table() {
    const GfTables t = make_tables();
    uint8_t g[NROOTS + 1] = {1};
    int deg = 0;
    for (int i = 0; i < NROOTS; i++) {  // multiply by (x + alpha^i)
        const uint8_t root = t.ato[i];
        uint8_t ng[NROOTS + 1] = {};
        for (int j = 0; j <= deg; j++) {
            ng[j + 1] ^= g[j];
            if (g[j]) ng[j] ^= t.ato[t.iof[g[j]] + t.iof[root]];
        }
        deg++;
        for (int j = 0; j <= deg; j++) g[j] = ng[j];
    }
    NibTable G{};
    for (int half = 0; half < 2; half++)
        for (int n = 1; n < 16; n++) {
            const int x = half ? n << 4 : n;
            for (int j = 0; j < NROOTS; j++) {
                const uint32_t prod = g[j] ? t.ato[t.iof[x] + t.iof[g[j]]] : 0u;
                G.w[(half * 16 + n) * 4 + (j < 8 ? j / 4 : 3)] |= prod << (8 * (j % 4));  // g_8, g_9 in dword 3
            }
        }
#if RS_LFSR2
    // Two steps at once: r'' = shift2(r; d1, d2) + Row(r_8) + Row2(r_9) with Row2(a)_j = a * (g_(j-1) + g_9 * g_j), g_(-1) = 0
    // (the second step's feedback byte is r_8 + r_9 * g_9, and Row is linear)
    for (int half = 0; half < 2; half++)
        for (int n = 1; n < 16; n++) {
            const int x = half ? n << 4 : n;
            for (int j = 0; j < NROOTS; j++) {
                uint8_t c = j ? g[j - 1] : 0;
                if (g[9] && g[j]) c ^= t.ato[t.iof[g[9]] + t.iof[g[j]]];
                const uint32_t prod = c ? t.ato[t.iof[x] + t.iof[c]] : 0u;
                G.w[((2 + half) * 16 + n) * 4 + (j < 8 ? j / 4 : 3)] |= prod << (8 * (j % 4));
            }
        }
#endif
    return G;
}
__constant__ NibTable g_nib = make_nib_table();
__constant__ QuadTable g_quad = make_quad_table();

// Four Chien steps per lookup: W[j-1][c] = (c*a^j, c*a^2j, c*a^3j, c*a^4j) packed little-endian, j = 1..5 (a = alpha).
constexpr int STEP_TERMS = 5;
struct StepTable {
    uint32_t w[STEP_TERMS * 256];
};
constexpr StepTable make_step_table() {
    const GfTables t = make_tables();
    StepTable S{};
    for (int j = 1; j <= STEP_TERMS; j++)
        for (int c = 1; c < 256; c++)
            for (int k = 1; k <= 4; k++) S.w[(j - 1) * 256 + c] |= (uint32_t)t.ato[t.iof[c] + j * k] << (8 * (k - 1));
    return S;
}
__constant__ StepTable g_step = make_step_table();

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LDS __attribute__((address_space(3)))
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t mod255(uint32_t x) { return (x * 0x1010102u) >> 24; }  // x < 65536
__device__ __forceinline__ uint32_t mod510(uint32_t x) { return x < x - NN ? x : x - NN; }  // x < 510 (x - 255 wraps below 255)

// Chien search over i = 1..255 (rschecksf.cpp:299-320) in the reference's own form - the logs b[j] advance by j, the sum
// is over alpha_to[b[j]] - for the rare wave that holds a locator above degree 5 (never correctable, but its roots are
// counted).  Roots go to the column's parity rows like chien_wave's.  Lanes with need == false find nothing.
__device__ __forceinline__ int chien_log(const uint32_t (&lam)[NROOTS + 1], bool need, int deg_lambda, uint8_t* col,
                                         uint32_t stride, const uint8_t* __restrict__ ato) {
    uint32_t b[NROOTS + 1], live[NROOTS + 1];
#pragma unroll
    for (int j = 1; j <= NROOTS; j++) {
        live[j] = need && lam[j] != NN ? 0xFFu : 0u;
        b[j] = live[j] ? lam[j] : 0u;
    }
    bool searching = need;
    int count = 0;
    for (int i = 1; i <= NN; i++) {
        if (searching) {
            uint32_t q = 1;  // lambda[0] is always 1
#pragma unroll
            for (int j = 1; j <= NROOTS; j++) {
                b[j] += (uint32_t)j;
                b[j] = b[j] < (uint32_t)NN ? b[j] : b[j] - NN;
                q ^= ato[b[j]] & live[j];
            }
            if (q == 0) {
                col[(uint32_t)(NMSG + count) * stride] = (uint8_t)i;
                if (++count == deg_lambda) searching = false;
            }
        }
        if (!__any(searching)) break;
    }
    return count;
}

// Chien search, four positions per lookup, every lane on its own column, for locators of degree <= D <= 5.
// adr[j] = 4 * lambda_j * alpha^(j*i0): one dword of the step table gives the term at i0+1..i0+4 and the next address;
// the XOR of the terms' dwords evaluates the locator at four positions at once, a zero byte is a root.
// Roots go to the column's parity rows (rows 110..119 are dead once the syndromes are known and are never output; Forney
// takes the set in any order).  Lanes with need == false carry zero terms and find nothing.  Returns the number of roots.
// (The 255-step byte scan it replaces cost 0.63 ms of the 1.01 ms a batch with three-error columns took; this 0.11.)
template <int D>
__device__ __forceinline__ int chien_quad(const uint32_t (&lam)[NROOTS + 1], bool need, uint8_t* col, uint32_t stride,
                                          const uint8_t* __restrict__ ato, const uint32_t* __restrict__ step) {
    static_assert(D <= STEP_TERMS, "step table");
    uint32_t adr[D + 1];
#pragma unroll
    for (int j = 1; j <= D; j++) adr[j] = need && lam[j] != NN ? (uint32_t)ato[lam[j]] << 2 : 0u;  // lam[] in index form
    int count = 0;
    for (int blk = 0; blk < 8; blk++) {
        uint32_t m = 0;
#pragma unroll
        for (int it = 0; it < 8; it++) {
            uint32_t q = 0x01010101u;  // lambda_0 = 1 at each of the four positions
#pragma unroll
            for (int j = 1; j <= D; j++) {
                const uint32_t t =
                    *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(step) + (j - 1) * 1024 + adr[j]);
                q ^= t;
                adr[j] = (t >> 22) & 0x3FCu;
            }
            const uint32_t z = ~(((q & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | q | 0x7F7F7F7Fu);  // 0x80 exactly where a byte is 0
            m = (m >> 1) | z;  // step `it`, byte k ends up in bit 8k + it
        }
        if (blk == 7) m &= 0x7FFFFFFFu;  // position 256 = position 1 again
        if (__any(m != 0)) {
            while (m) {
                const uint32_t bit = (uint32_t)__ffs((int)m) - 1u;
                m &= m - 1u;
                col[(uint32_t)(NMSG + count) * stride] = (uint8_t)(32u * blk + 4u * (bit & 7u) + (bit >> 3) + 1u);
                count++;
            }
        }
    }
    return count;
}

// The same search one column at a time with the whole wavefront on it: lane L evaluates the locator at the four
// positions i = 4L+1 .. 4L+4, so 64 lanes cover 1..256 in one step (256 = position 1 again, masked).  Term j at those
// positions is lambda_j * alpha^(4jL) * (alpha^j, alpha^2j, alpha^3j, alpha^4j): one byte of alpha_to at
// log(lambda_j) + (4jL mod 255) - a lane-strided read, the log broadcast from the owning lane - and one dword of the
// step table.  Cost is per column (about 1/23 of chien_quad<5>'s fixed cost), and the columns are taken in lane order =
// column order within a superframe, which lets the reference's rule work for us: once a column of a superframe has
// failed (here, or in `failed` = the lanes whose closed form already did), the later columns of that superframe are
// never output and are dropped from the list (`dropped`: their lanes return 0 and patch nothing).  In a batch with
// uncorrectable columns that is most of the work.  Must be called by all 64 lanes.
template <int D>
__device__ __forceinline__ void chien_wave(const uint32_t (&lam)[NROOTS + 1], bool need, bool failed, int deg_lambda,
                                           uint32_t sfid, int& count, bool& dropped, uint8_t* cwbase, uint32_t coloff,
                                           uint32_t stride, const uint8_t* __restrict__ ato,
                                           const uint32_t* __restrict__ step) {
    static_assert(D <= STEP_TERMS, "step table");
    const uint32_t lane = __lane_id();
    uint32_t e[D + 1], lgz[D + 1];
#pragma unroll
    for (int j = 1; j <= D; j++) {
        e[j] = mod255(4u * (uint32_t)j * lane);
        lgz[j] = lam[j] == (uint32_t)NN ? (uint32_t)ATO_ZERO : lam[j];  // a zero coefficient reads alpha_to's zero block
    }
    const uint32_t keep = lane == 63u ? 0x7FFFFFFFu : 0xFFFFFFFFu;  // position 256 = position 1 again
    const uint64_t failm = __ballot(failed);
    uint64_t work = __ballot(need) | failm;
    uint64_t gone = 0;
    while (work) {
        const int owner = __builtin_ctzll(work);
        work &= work - 1;
        bool fail = true;
        if (!((failm >> owner) & 1u)) {
            uint32_t q = 0x01010101u;  // lambda_0 = 1 at each of the four positions
#pragma unroll
            for (int j = 1; j <= D; j++) {
                const uint32_t c = ato[(uint32_t)__builtin_amdgcn_readlane(lgz[j], owner) + e[j]];  // <= 512 + 254
                q ^= step[(j - 1) * 256 + c];
            }
            uint32_t z = ~(((q & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | q | 0x7F7F7F7Fu) & keep;  // 0x80 exactly where a byte is 0
            uint8_t* ocol = cwbase + __builtin_amdgcn_readlane(coloff, owner);
            uint32_t found = 0;
            uint64_t hit;
            while ((hit = __ballot(z != 0)) != 0) {  // a second trip only when a lane holds two roots
                if (z) {
                    const uint32_t below =
                        __builtin_amdgcn_mbcnt_hi((uint32_t)(hit >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hit, 0u));
                    const uint32_t k = ((uint32_t)__ffs((int)z) - 1u) >> 3;
                    ocol[__umul24(NMSG + found + below, stride)] = (uint8_t)(4u * lane + k + 1u);
                    z &= z - 1u;
                }
                found += (uint32_t)__popcll(hit);
            }
            if (lane == (uint32_t)owner) count = (int)found;
            fail = (int)found != __builtin_amdgcn_readlane(deg_lambda, owner);
        }
        if (fail) {
            const uint64_t later = __ballot(sfid == (uint32_t)__builtin_amdgcn_readlane(sfid, owner)) & ~((2ull << owner) - 1ull);
            work &= ~later;
            gone |= later;
        }
    }
    dropped = (gone >> lane) & 1u;
}

// Everything after the syndromes, for the lanes with err set: s[] = their ten syndromes in polynomial form, not all
// zero.  Codeword byte k of a lane is cwbase[coloff + k * stride].  Returns root count or -1 when uncorrectable (0 for
// the other lanes); patches in place.  ALL 64 lanes of the wavefront must call it together (chien_wave).
// Every small polynomial array is indexed with compile-time constants only (loops fully
// unrolled, data-dependent bounds turned into predicates): dynamically indexed register arrays
// cost a v_cndmask chain per access on this target and made the error path ~8x slower.
// A wavefront runs this path as soon as ONE of its 64 columns has an error, so its cost is set by the
// worst column of the wave; wave-uniform guards (__any, the wave's largest locator degree) skip the
// terms that are zero in every lane - after a few single-symbol errors that is most of them.
__device__ int rs_correct(uint32_t (&s)[NROOTS], bool err, uint32_t sfid, uint8_t* cwbase, uint32_t coloff,
                          uint32_t stride, const uint8_t* __restrict__ ato, const uint8_t* __restrict__ iof,
                          const uint8_t* __restrict__ qsol, const uint32_t* __restrict__ step) {
    uint8_t* const col = cwbase + coloff;
    uint32_t lam[NROOTS + 1];
#pragma unroll
    for (int i = 0; i <= NROOTS; i++) lam[i] = NN;
    int deg_lambda = 0;
    if (err) {
#pragma unroll
        for (int i = 0; i < NROOTS; i++) s[i] = iof[s[i]];  // index form (s[10] of the reference is never used)

        // lambda / b: only entries 0..10 of the reference's 16-byte vectors are ever read.  lam[] is kept in index form
        // (255 = zero coefficient) - every use but the XOR of the update wants the log - next to its polynomial form pl[].
        // Before step r both polynomials have degree < r, which bounds every loop of the step at compile time.
        uint32_t pl[NROOTS + 1], b[NROOTS + 1];
#pragma unroll
        for (int i = 0; i <= NROOTS; i++) { pl[i] = 0; b[i] = NN; }
        pl[0] = 1;
        lam[0] = 0;
        b[0] = 0;
        int el = 0;
        // Single-symbol shortcut.  One error e at position p gives S_i = e * alpha^(i*p): a geometric sequence, and then
        // Berlekamp-Massey's answer is the unique connection polynomial 1 + (S_1/S_0) x (2L <= 10).  What the Viterbi decoder
        // leaves behind is almost always that (a burst of a few bits lands in one or two different columns), so when EVERY
        // erroneous column of the wave passes the test the ten BM iterations are skipped; anything else takes them.
        bool geo = s[0] != NN && s[1] != NN;
        const uint32_t lr = mod510(s[1] + NN - s[0]);  // log(S_1 / S_0); meaningless unless geo
        uint32_t expect = s[1];
#pragma unroll
        for (int i = 2; i < NROOTS; i++) {
            expect = mod510(expect + lr);  // log S_0 + i * lr
            geo = geo && s[i] == expect;   // a zero S_i (255) never matches
        }
        const bool shortcut = __all(geo);
        if (shortcut) lam[1] = lr;
#pragma unroll
        for (int r = 1; r <= NROOTS; r++) {  // Berlekamp-Massey (rschecksf.cpp:240-284)
            if (shortcut) break;
            uint32_t discr = 0;
#pragma unroll
            for (int i = 0; i < r; i++) {
                const bool term = lam[i] != NN && s[r - i - 1] != NN;
                if (__any(term)) {
                    if (term) discr ^= ato[lam[i] + s[r - i - 1]];
                }
            }
            discr = iof[discr];
            const bool zero = discr == NN;
            if (__any(!zero)) {
                const bool grow = !zero && 2 * el <= r - 1;
                uint32_t t[NROOTS + 1];
#pragma unroll
                for (int i = 0; i < r; i++) {
                    t[i + 1] = pl[i + 1];
                    if (!zero && b[i] != NN) t[i + 1] ^= ato[discr + b[i]];
                }
                if (grow) el = r - el;
                // b <- inv(discr) * lambda (grow) or x * b (otherwise: _mm_slli_si128(b,1), b[0] = 255)
#pragma unroll
                for (int i = r; i >= 0; i--) {
                    const uint32_t scaled = lam[i] == NN ? (uint32_t)NN : mod510(lam[i] + NN - discr);
                    const uint32_t shifted = i ? b[i - 1] : (uint32_t)NN;
                    b[i] = grow ? scaled : shifted;
                }
#pragma unroll
                for (int i = 1; i <= r; i++) {
                    pl[i] = zero ? pl[i] : t[i];
                    lam[i] = iof[pl[i]];
                }
            } else {  // zero discrepancy in every lane: lambda stays, b <- x * b
#pragma unroll
                for (int i = r; i > 0; i--) b[i] = b[i - 1];
                b[0] = NN;
            }
        }
#pragma unroll
        for (int i = 0; i <= NROOTS; i++)
            if (lam[i] != NN) deg_lambda = i;
    }
    int dmax = 0;  // largest locator degree in this wave (uniform; deg_lambda = 0 in the lanes without an error)
#pragma unroll
    for (int d = 1; d <= NROOTS; d++)
        if (__any(deg_lambda >= d)) dmax = d;
    dmax = __builtin_amdgcn_readfirstlane(dmax);

    uint32_t root[NROOTS];
#pragma unroll
    for (int k = 0; k < NROOTS; k++) root[k] = 0;
    int count = 0;
    // Degrees 1 and 2 in closed form, whatever the rest of the wave needs.
    // Degree 1: 1 + lambda_1 * alpha^i = 0 has exactly one solution in i = 1..255: i = 255 - log(lambda_1).
    // Degree 2: 1 + l1 x + l2 x^2 = 0 with x = (l1/l2) y becomes y^2 + y = l2 / l1^2; its solutions y0, y0 ^ 1 come from a
    // 256-byte table, the Chien index of a root x is log x (0 -> 255).  No solution, or l1 = 0 (a double root, which the
    // scan counts once): count stays below the degree and the column fails, as in the scan.
    // Two symbol errors in a column are what is left of the decoder's longer bursts; a 255-step scan would make every such
    // wave - and, through the barriers, its whole workgroup - several times slower than its neighbours.
    if (deg_lambda == 1) {
        root[0] = (uint32_t)NN - lam[1];
        count = 1;
    } else if (dmax >= 2 && deg_lambda == 2) {
        if (lam[1] != NN) {
            const uint32_t y0 = qsol[ato[mod255(lam[2] + 2u * (NN - lam[1]))]];
            if (y0) {
                const uint32_t scale = lam[1] + NN - lam[2];  // log(l1 / l2) + 255
                uint32_t r0 = mod255(scale + iof[y0]), r1 = mod255(scale + iof[y0 ^ 1u]);
                r0 = r0 ? r0 : (uint32_t)NN;
                r1 = r1 ? r1 : (uint32_t)NN;
                root[0] = r0 < r1 ? r0 : r1;  // the scan finds them in increasing order
                root[1] = r0 < r1 ? r1 : r0;
                count = 2;
            }
        } else {
            // x^2 = 1 / l2: one root (every element of GF(2^8) has exactly one square root), counted once
            count = 1;
        }
    }
    bool dropped = false;
    if (dmax >= 3) {  // the scan, for the lanes of degree 3 and above
        const bool need = deg_lambda >= 3;
        if (dmax <= STEP_TERMS) {
            // few such columns, or failures that end their superframes early: one step per column; a wave full of them:
            // one column per lane (chien_quad's cost does not depend on how many lanes need it)
            const int heavy = __popcll(__ballot(need));
            if (heavy <= (dmax <= 3 ? COOP_MAX3 : COOP_MAX5)) {
                const bool failed = err && !need && deg_lambda != count;
                if (dmax <= 3) chien_wave<3>(lam, need, failed, deg_lambda, sfid, count, dropped, cwbase, coloff, stride, ato, step);
                else chien_wave<STEP_TERMS>(lam, need, failed, deg_lambda, sfid, count, dropped, cwbase, coloff, stride, ato, step);
            } else {
                const int found = dmax <= 3 ? chien_quad<3>(lam, need, col, stride, ato, step)
                                            : chien_quad<STEP_TERMS>(lam, need, col, stride, ato, step);
                if (need) count = found;
            }
        } else if (need) {
            count = chien_log(lam, need, deg_lambda, col, stride, ato);
        }
        if (need && !dropped) {
#pragma unroll
            for (int k = 0; k < NROOTS; k++)
                if (k < dmax && k < count) root[k] = col[(uint32_t)(NMSG + k) * stride];
        }
    }
    if (!err || dropped) return 0;
    if (deg_lambda != count) return -1;

    const int deg_omega = deg_lambda - 1;
    uint32_t om[NROOTS];
#pragma unroll
    for (int i = 0; i < NROOTS; i++) {  // omega (rschecksf.cpp:331-341); only om[0..deg_omega] is used
        om[i] = NN;
        if (i < dmax) {
            uint32_t tmp = 0;
#pragma unroll
            for (int j = 0; j <= i; j++)
                if (s[i - j] != NN && lam[j] != NN) tmp ^= ato[s[i - j] + lam[j]];
            om[i] = i <= deg_omega ? (uint32_t)iof[tmp] : (uint32_t)NN;
        }
    }
    const int top = (deg_lambda < NROOTS - 1 ? deg_lambda : NROOTS - 1) & ~1;
#pragma unroll
    for (int j = NROOTS - 1; j >= 0; j--) {  // Forney (rschecksf.cpp:346-374)
        if (j >= dmax) continue;  // count <= dmax in every lane
        const uint32_t rt = root[j];
        if (j < count && rt >= PADN + 1) {  // roots in the virtual padding are skipped, still counted
            uint32_t num1 = 0;
#pragma unroll
            for (int i = 0; i < NROOTS; i++)
                if (i < dmax && i <= deg_omega && om[i] != NN) num1 ^= ato[mod255(om[i] + i * rt)];
            if (num1) {
                const uint32_t num2 = ato[NN - rt];
                uint32_t den = 0;
#pragma unroll
                for (int i = 0; i < NROOTS; i += 2)
                    if (i < dmax && i <= top && lam[i + 1] != NN) den ^= ato[mod255(lam[i + 1] + i * rt)];
                const uint32_t tmp = (uint32_t)iof[num1] + iof[num2] + (NN - iof[den]);  // <= 763
                col[(rt - 1 - PADN) * stride] ^= ato[mod255(tmp)];
            }
        }
    }
    return count;
}

// Transposed-LDS front end (codeword byte k at cw[tid + k * RS_THREADS]): Horner syndromes (rschecksf.cpp:212-219)
// with per-root product tables (mulp[i][x] = x * alpha^i: one LDS byte per multiply-add).  Called by all lanes.
__device__ int decode_rs(bool active, uint32_t sfid, uint8_t* cw, uint32_t tid, const uint8_t* __restrict__ ato,
                         const uint8_t* __restrict__ iof, const uint8_t* __restrict__ mulp,
                         const uint8_t* __restrict__ qsol, const uint32_t* __restrict__ step) {
    uint32_t s[NROOTS];
    uint32_t syn = 0;
    if (active) {
        const uint8_t* col = cw + tid;
        const uint32_t d0 = col[0];
#pragma unroll
        for (int i = 0; i < NROOTS; i++) s[i] = d0;
        for (int j = 1; j < NCW; j++) {
            const uint32_t d = col[j * RS_THREADS];
            s[0] ^= d;
#pragma unroll
            for (int i = 1; i < NROOTS; i++) s[i] = d ^ mulp[i * 256 + s[i]];
        }
#pragma unroll
        for (int i = 0; i < NROOTS; i++) syn |= s[i];
    }
    if (!__any(syn != 0)) return 0;
    return rs_correct(s, syn != 0, sfid, cw, tid, RS_THREADS, ato, iof, qsol, step);
}

// Natural-layout front end (codeword byte k at cwbase[coloff + k * stride]): remainder modulo g(x) by LFSR,
// r <- r*x + d_k - r_9*(x^10 + g(x)): two conflict-free 16-byte lookups (low and high nibble of the feedback byte,
// see NibTable) per data byte; two bytes per trip (RS_LFSR2: the second feedback byte is linear in r_8 and r_9, so its
// row is folded into a second pair of tables and all four lookups of a trip start from the same register).  Called by all lanes; the ones without a column (active == false) walk a valid one
// and drop the result.
__device__ int decode_rs_lfsr(bool active, uint32_t sfid, uint8_t* cwbase, uint32_t coloff, uint32_t stride,
                              const uint8_t* __restrict__ ato, const uint8_t* __restrict__ iof,
                              const uint8_t* __restrict__ qsol, const uint32_t* __restrict__ gnib,
                              const uint32_t* __restrict__ step) {
    uint32_t r0 = 0, r1 = 0, r2 = 0;  // coefficient r_j = byte j of the 80-bit register (r2 above bit 15: junk)
    const uint8_t* q = cwbase + coloff;
    const uint32_t nibbase = (uint32_t)(uintptr_t)(const LDS uint32_t*)gnib;  // LDS byte address
#if RS_LFSR2
    static_assert(NCW % 2 == 0, "two data bytes per step");
#pragma unroll 4
    for (int k = 0; k < NCW; k += 2, q += 2 * stride) {
        const uint32_t d1 = q[0], d2 = q[stride];
        const uint32_t a4 = r2 >> 4;  // bits 4..11 = r_9; r_8 = bits 0..7 of r2
        const u32x4 lo2 = *reinterpret_cast<const LDS u32x4*>(nibbase + 512u + (a4 & 0xF0u));
        const u32x4 hi2 = *reinterpret_cast<const LDS u32x4*>(nibbase + 768u + ((a4 >> 4) & 0xF0u));
        const u32x4 lo1 = *reinterpret_cast<const LDS u32x4*>(nibbase + ((r2 << 4) & 0xF0u));
        const u32x4 hi1 = *reinterpret_cast<const LDS u32x4*>(nibbase + 256u + (r2 & 0xF0u));
        r2 = xor3(xor3(__builtin_amdgcn_alignbit(r2, r1, 16), lo1.w, hi1.w), lo2.w, hi2.w);
        r1 = xor3(xor3(__builtin_amdgcn_alignbit(r1, r0, 16), lo1.y, hi1.y), lo2.y, hi2.y);
        r0 = xor3(xor3((((r0 << 16) | d2) | (d1 << 8)), lo1.x, hi1.x), lo2.x, hi2.x);
        asm volatile("" ::"v"(lo1.z), "v"(hi1.z), "v"(lo2.z), "v"(hi2.z));
    }
#else
#pragma unroll 8
    for (int k = 0; k < NCW; k++, q += stride) {
#ifdef RS_DIAG_NO_DATA
        const uint32_t d = (uint32_t)k;  // timing-only diagnostic: no data byte read (outputs wrong)
#else
        const uint32_t d = *q;
#endif
#ifdef RS_DIAG_FAKE_IDX
        const uint32_t f4 = (uint32_t)k << 4;  // timing-only diagnostic: every lane reads the same rows (outputs wrong)
#else
        const uint32_t f4 = r2 >> 4;  // bits 4..11 = r_9
#endif
        // g_8, g_9 sit in the LAST dword of a row (the third is empty) so that the access stays ONE ds_read_b128
        // (4 LDS cycles per wave): with the payload in the first 12 bytes the compiler narrows it to ds_read_b96,
        // which takes 8 (measured: 0.36 ms against 0.20 per 131072 superframes of RSDims 24)
        const u32x4 lo = *reinterpret_cast<const LDS u32x4*>(nibbase + (f4 & 0xF0u));
        const u32x4 hi = *reinterpret_cast<const LDS u32x4*>(nibbase + 256u + ((f4 >> 4) & 0xF0u));
        r2 = xor3(__builtin_amdgcn_alignbit(r2, r1, 24), lo.w, hi.w);
        r1 = xor3(__builtin_amdgcn_alignbit(r1, r0, 24), lo.y, hi.y);
        r0 = xor3((r0 << 8) | d, lo.x, hi.x);
        // keep the empty dwords' registers allocated until the rows have arrived: reused for the next address they
        // would put a wait for the lookup in front of its computation
        asm volatile("" ::"v"(lo.z), "v"(hi.z));
    }
#endif
    r2 &= 0xFFFFu;
    const bool err = active && (r0 | r1 | r2) != 0;
    if (!__any(err)) return 0;
    // syndromes S_i = r(alpha^i) = sum_j r_j alpha^(i*j) (= the reference's Horner sums over the whole codeword), through
    // the logs of the ten coefficients: independent lookups at ato[log r_j + i*j], a zero coefficient reads the zero block
    uint32_t s[NROOTS];
    if (err) {
        uint32_t lg[NROOTS];
        s[0] = 0;
#pragma unroll
        for (int j = 0; j < NROOTS; j++) {
            const uint32_t c = ((j < 4 ? r0 : j < 8 ? r1 : r2) >> (8 * (j & 3))) & 0xFFu;
            s[0] ^= c;
            lg[j] = c ? (uint32_t)iof[c] : (uint32_t)ATO_ZERO;
        }
#pragma unroll
        for (int i = 1; i < NROOTS; i++) {
            uint32_t v = 0;
#pragma unroll
            for (int j = 0; j < NROOTS; j++) v ^= ato[lg[j] + (uint32_t)(i * j)];  // <= 512 + 81
            s[i] = v;
            if (i % 3 == 0) __builtin_amdgcn_sched_barrier(0);  // 30 lookups in flight are plenty; all 90 cost spills
        }
    }
    return rs_correct(s, err, sfid, cwbase, coloff, stride, ato, iof, qsol, step);
}

// General form (used for rsdims > 256).  Workgroup = 256 lanes.  For rsdims <= 256 it takes spb = 256/rsdims
// superframes per pass; for wider superframes it walks the columns in 256-wide chunks, in
// order, and stops after the first chunk that holds a failure.
__global__ __launch_bounds__(RS_THREADS) void rs_kernel_wide(const uint8_t* __restrict__ p, uint8_t* __restrict__ out,
                                                        int32_t* __restrict__ ret, uint32_t rsdims,
                                                        long long nsf) {
    __shared__ uint8_t cw[NCW * RS_THREADS];  // [row][lane]
    __shared__ uint8_t ato[ATO_SIZE];
    __shared__ uint8_t iof[256];
    __shared__ uint8_t mulp[(NROOTS + 1) * 256];  // mulp[i][x] = x * alpha^i, i = 0..10
    __shared__ uint8_t qsol[256];
    __shared__ uint32_t step[STEP_TERMS * 256];
    __shared__ int s_minfail[RS_THREADS];
    __shared__ int s_sum[RS_THREADS];
    __shared__ int s_fail[RS_THREADS];
    const int tid = threadIdx.x;
    for (int i = tid; i < ATO_SIZE; i += RS_THREADS) ato[i] = g_gf.ato[i];
    iof[tid] = g_gf.iof[tid];
    qsol[tid] = g_quad.q[tid];
    for (int i = 0; i <= NROOTS; i++) mulp[i * 256 + tid] = tid ? g_gf.ato[g_gf.iof[tid] + i] : 0;
    for (int i = tid; i < STEP_TERMS * 256; i += RS_THREADS) step[i] = g_step.w[i];
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
            if (active) {
                const uint8_t* src = p + (size_t)sf * in_sz + colidx;
                for (int k = 0; k < NCW; k++) cw[k * RS_THREADS + tid] = src[(size_t)k * rsdims];
            }
            const int res = decode_rs(active, lsf, cw, tid, ato, iof, mulp, qsol, step);  // all lanes
            if (res < 0) atomicMin(&s_minfail[lsf], (int)colidx);
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


// Linear copy of n bytes between global memory and LDS with the widest access the alignment allows.
template <typename V>
__device__ __forceinline__ void copy_vec(uint8_t* dst, const uint8_t* src, uint32_t n, uint32_t tid) {
    const uint32_t nv = n / (uint32_t)sizeof(V);
#pragma clang loop vectorize(disable) interleave(disable)
    for (uint32_t i = tid; i < nv; i += RS_THREADS) reinterpret_cast<V*>(dst)[i] = reinterpret_cast<const V*>(src)[i];
#pragma clang loop vectorize(disable) interleave(disable)
    for (uint32_t i = nv * (uint32_t)sizeof(V) + tid; i < n; i += RS_THREADS) dst[i] = src[i];
}
// (not inlined: the fall-back for an input pointer that is not 8-byte aligned should cost the main loop no registers)
__device__ __attribute__((noinline)) void copy_linear(uint8_t* dst, const uint8_t* src, uint32_t n, uint32_t tid) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src);
    if ((a & 15u) == 0) copy_vec<uint4>(dst, src, n, tid);
    else if ((a & 7u) == 0) copy_vec<uint2>(dst, src, n, tid);
    else if ((a & 3u) == 0) copy_vec<uint32_t>(dst, src, n, tid);
    else copy_vec<uint8_t>(dst, src, n, tid);
}

// Output of the superframes without a failure: their first 110 rows are one linear block each (out_sz bytes of the in_sz).
// (l, o) = superframe and piece (of sizeof(V) bytes) of this thread, stepped RS_THREADS pieces at a time without a division.
template <typename V>
__device__ __forceinline__ void copy_out(uint8_t* dst0, const uint8_t* cw, const int* s_minfail, uint32_t nloc,
                                         uint32_t in_sz, uint32_t out_sz, uint32_t tid) {
    const uint32_t wps = out_sz / (uint32_t)sizeof(V);
    uint32_t l = tid / wps, o = tid - l * wps;
    while (l < nloc) {
        if (s_minfail[l] == 0x7FFFFFFF)
            reinterpret_cast<V*>(dst0 + l * out_sz)[o] = reinterpret_cast<const V*>(cw + l * in_sz)[o];
        o += RS_THREADS;
        while (o >= wps) {
            o -= wps;
            l++;
        }
    }
}

// rsdims <= 256: spb = 256/rsdims superframes per pass, natural layout in LDS (see the header comment).
// LDS: 30720 (codewords) + 5120 (Chien steps) + 1024 (LFSR nibble rows) + 768 + 256 + 256 + 2048 = 40192 B: four workgroups
// share a CU (second launch bound: 4 waves per SIMD); a three-workgroup build was 6 % slower on clean data
// (profiles/r02_ab_rs_chien.txt).
__global__ __launch_bounds__(RS_THREADS, 4) void rs_kernel(const uint8_t* __restrict__ p, uint8_t* __restrict__ out,
                                                        int32_t* __restrict__ ret, uint32_t rsdims,
                                                        long long nsf, int host_polls_ret) {
    __shared__ __attribute__((aligned(16))) uint8_t cw[NCW * RS_THREADS];  // [superframe][row][column]
    __shared__ __attribute__((aligned(256))) uint32_t gnib[NIB_TABLES * 16 * 4];
    __shared__ uint32_t step[STEP_TERMS * 256];
    __shared__ uint8_t ato[ATO_SIZE];
    __shared__ uint8_t iof[256];
    __shared__ uint8_t qsol[256];
    __shared__ int s_minfail[RS_THREADS];
    __shared__ int s_sum[RS_THREADS];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < ATO_SIZE; i += RS_THREADS) ato[i] = g_gf.ato[i];
    iof[tid] = g_gf.iof[tid];
    qsol[tid] = g_quad.q[tid];
    if (tid < (uint32_t)NIB_TABLES * 16u * 4u) gnib[tid] = g_nib.w[tid];
    for (uint32_t i = tid; i < STEP_TERMS * 256u; i += RS_THREADS) step[i] = g_step.w[i];
    __syncthreads();

    const uint32_t spb = RS_THREADS / rsdims;
    const long long ngroups = (nsf + spb - 1) / spb;
    const uint32_t in_sz = NCW * rsdims, out_sz = NMSG * rsdims;
    const uint32_t lsf = tid / rsdims, colidx = tid - lsf * rsdims;
    constexpr int NOFAIL = 0x7FFFFFFF;

    // The next group's input block is fetched into registers (8-byte words, 15 per thread) while the current
    // one is decoded, so the HBM latency of the load phase overlaps the LDS-bound decode of the same workgroup.
    constexpr uint32_t PRE = NCW * RS_THREADS / 8u / RS_THREADS;  // 15
    const bool pre_ok = (reinterpret_cast<uintptr_t>(p) & 7u) == 0;  // in_sz is a multiple of 8
    uint2 pre[PRE];
    auto prefetch = [&](long long gg) {
        const long long s0 = gg * spb;
        const uint32_t nl = nsf - s0 < (long long)spb ? (uint32_t)(nsf - s0) : spb;
        const uint32_t nw = nl * in_sz / 8u;
        const uint2* src = reinterpret_cast<const uint2*>(p + (size_t)s0 * in_sz);
#pragma unroll
        for (uint32_t k = 0; k < PRE; k++) {
            const uint32_t i = tid + k * RS_THREADS;
            pre[k] = i < nw ? src[i] : make_uint2(0u, 0u);
        }
    };
    if (pre_ok && (long long)blockIdx.x < ngroups) prefetch(blockIdx.x);

#ifndef RS_PRIO_ROT
#define RS_PRIO_ROT 1
#endif
#if RS_PRIO_ROT
    // Issue priority that rotates pass by pass (see vit_pk.hip: the hardware's arbitration otherwise lets the waves of a SIMD - here
    // one of each of the CU's four workgroups - advance one after the other): -3 % on clean data, -6 % with errors in 6 % of the columns
    // (profiles/r03_ab_rs_lfsr2.txt).  RS_PRIO_ROT = 2: the four waves of a workgroup share the level (the slot of wave 0).
    uint32_t rs_slot, rs_pass = 0;
    {
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        rs_slot = hwid & 3u;
#if RS_PRIO_ROT == 2
        __shared__ uint32_t s_slot;
        if (tid == 0) s_slot = rs_slot;
        __syncthreads();
        rs_slot = s_slot;
        rs_slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)rs_slot);
#endif
    }
#endif
    for (long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
#if RS_PRIO_ROT
        if (ngroups > 2 * (long long)gridDim.x)  // a launch of one or two passes has nothing to level (and measured 2 % slower with it)
        switch ((rs_slot + rs_pass) & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
        rs_pass++;
#endif
        const long long sf0 = g * spb;
        const uint32_t nloc = nsf - sf0 < (long long)spb ? (uint32_t)(nsf - sf0) : spb;
        if (tid < spb) {
            s_minfail[tid] = NOFAIL;
            s_sum[tid] = 0;
        }
        if (pre_ok) {
            const uint32_t nw = nloc * in_sz / 8u;
#pragma unroll
            for (uint32_t k = 0; k < PRE; k++) {
                const uint32_t i = tid + k * RS_THREADS;
                if (i < nw) reinterpret_cast<uint2*>(cw)[i] = pre[k];
            }
            if (g + gridDim.x < ngroups) prefetch(g + gridDim.x);
        } else {
            copy_linear(cw, p + (size_t)sf0 * in_sz, nloc * in_sz, tid);
        }
        __syncthreads();
        const bool active = lsf < nloc;
        const int res = decode_rs_lfsr(active, lsf, cw, active ? lsf * in_sz + colidx : 0u, rsdims, ato, iof, qsol, gnib, step);  // all lanes
        if (res < 0) atomicMin(&s_minfail[lsf], (int)colidx);
        __syncthreads();
        uint8_t* dst0 = out + (size_t)sf0 * out_sz;
        // superframes without a failure: their first 110 rows go out as one linear block each
        if ((out_sz & 3u) == 0 && (reinterpret_cast<uintptr_t>(dst0) & 3u) == 0) {
            // 16-byte pieces when the block sizes and the destination allow it (in_sz is a multiple of 8, both are of 16
            // when rsdims is a multiple of 8), else dwords (the quotient c / wps in every trip of the plain loop was a
            // quarter of the clean path's instructions)
            if ((out_sz & 15u) == 0 && (in_sz & 15u) == 0 && (reinterpret_cast<uintptr_t>(dst0) & 15u) == 0)
                copy_out<uint4>(dst0, cw, s_minfail, nloc, in_sz, out_sz, tid);
            else
                copy_out<uint32_t>(dst0, cw, s_minfail, nloc, in_sz, out_sz, tid);
            if (active) {
                const int mf = s_minfail[lsf];
                if (mf != NOFAIL && (int)colidx < mf) {  // columns before the first failure are written
                    uint8_t* dst = dst0 + (lsf * out_sz + colidx);  // < 110 * 256
                    const uint8_t* src = cw + lsf * in_sz + colidx;
#pragma clang loop vectorize(disable) interleave(disable)
                    for (int k = 0; k < NMSG; k++) dst[(size_t)k * rsdims] = src[k * rsdims];
                }
                if ((int)colidx < mf) atomicAdd(&s_sum[lsf], res);
            }
        } else if (active) {
            const int mf = s_minfail[lsf];
            if ((int)colidx < mf) {
                uint8_t* dst = dst0 + (lsf * out_sz + colidx);  // < 110 * 256
                const uint8_t* src = cw + lsf * in_sz + colidx;
#pragma clang loop vectorize(disable) interleave(disable)
                for (int k = 0; k < NMSG; k++) dst[(size_t)k * rsdims] = src[k * rsdims];
                atomicAdd(&s_sum[lsf], res);
            }
        }
        // single-call path (RScheckSuperframe): the host spins on ret[0] in its mapped buffer instead of waiting for the
        // end-of-kernel signal, so every output byte must be visible system-wide before the return value is
        if (host_polls_ret) __threadfence_system();
        __syncthreads();
        if (tid < nloc) {
            const int32_t r = s_minfail[tid] != NOFAIL ? -1 : s_sum[tid];
            if (host_polls_ret) __hip_atomic_store(&ret[sf0 + tid], r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            else ret[sf0 + tid] = r;
        }
        __syncthreads();
    }
}

}  // namespace

hipError_t rs_launch(const uint8_t* d_p, uint8_t* d_out, int32_t* d_ret, uint32_t rsdims, int64_t nsf,
                     hipStream_t stream, bool host_polls_ret) {
    if (nsf <= 0 || rsdims == 0) return hipSuccess;
    if (host_polls_ret && (nsf != 1 || rsdims > RS_THREADS)) return hipErrorInvalidValue;
    const uint32_t spb = rsdims <= RS_THREADS ? RS_THREADS / rsdims : 1u;
    long long groups = (nsf + spb - 1) / spb;
    if (groups > (1 << 20)) groups = 1 << 20;
    if (rsdims <= RS_THREADS) {
        // persistent workgroups (4 fit a CU): each walks its groups with the next input block already in flight
        int dev = 0;
        const int cus = hipGetDevice(&dev) == hipSuccess ? vit_device_cus(dev) : 256;
        const long long resident = 4LL * cus;
        if (groups > resident) groups = resident;
    }
    if (rsdims <= RS_THREADS)
        hipLaunchKernelGGL(rs_kernel, dim3((unsigned)groups), dim3(RS_THREADS), 0, stream, d_p, d_out, d_ret, rsdims,
                           (long long)nsf, host_polls_ret ? 1 : 0);
    else
        hipLaunchKernelGGL(rs_kernel_wide, dim3((unsigned)groups), dim3(RS_THREADS), 0, stream, d_p, d_out, d_ret,
                           rsdims, (long long)nsf);
    return hipGetLastError();
}
