/*
 * oracle/vit_avx2.c -- TEST INFRASTRUCTURE ONLY (see vit_oracle.h).
 *
 * Own AVX2 implementation of the integer specification in SURVEY.md Appendix A
 * (what the reference's decon_avx2 computes, deconvolve.cpp:514-526), kept in
 * NATURAL state order: `lo` = path metrics of states 0..31, `hi` = states
 * 32..63; the interleave of the survivors is undone with two cross-lane
 * permutes instead of the reference's pre-permuted constants.  It exists to be
 * the CPU baseline timed beside the GPU on the GPU box's host cores ("port");
 * tests check it bit-for-bit against the scalar restatement.
 */
#include "vit_oracle.h"

#include <immintrin.h>
#include <stdlib.h>
#include <string.h>

int vo_has_avx2(void) { return __builtin_cpu_supports("avx2") ? 1 : 0; }

static inline unsigned par8(unsigned x) {
    x ^= x >> 4;
    x ^= x >> 2;
    x ^= x >> 1;
    return x & 1u;
}

/* branch masks (parity((2i) & poly_j), const.asm:27-63 in natural order): built once per process, not per frame */
static uint8_t g_mk[4][32] __attribute__((aligned(32)));
__attribute__((constructor)) static void vo_avx2_init_masks(void) {
    static const int polys[4] = {109, 79, 83, 109};
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 32; i++) g_mk[j][i] = par8((2u * i) & polys[j]) ? 0xFF : 0;
}

__attribute__((target("avx2")))
int vo_deconvolve_avx2_u8(unsigned framebits, const uint8_t *sym,
                          unsigned char *out) {
    if (framebits > 9216) return 1;
    const __m256i k0 = _mm256_load_si256((const __m256i *)g_mk[0]);
    const __m256i k1 = _mm256_load_si256((const __m256i *)g_mk[1]);
    const __m256i k2 = _mm256_load_si256((const __m256i *)g_mk[2]);
    const __m256i k3 = _mm256_load_si256((const __m256i *)g_mk[3]);
    const __m256i c63 = _mm256_set1_epi8(63);

    uint64_t dec[9216 + 6]; /* on the stack like the reference (deconvolve.cpp:127) */

    uint8_t init[32];
    memset(init, 63, 32);
    __m256i hi = _mm256_loadu_si256((const __m256i *)init);
    init[0] = 0;
    __m256i lo = _mm256_loadu_si256((const __m256i *)init);

    unsigned nb = (framebits + 6) / 2;
    unsigned t = 0;
    for (unsigned it = 0; it < nb; it++) {
        for (int half = 0; half < 2; half++, t++) {
            const uint8_t *s = sym + 4u * t;
            __m256i x0 = _mm256_xor_si256(_mm256_set1_epi8((char)s[0]), k0);
            __m256i x1 = _mm256_xor_si256(_mm256_set1_epi8((char)s[1]), k1);
            __m256i x2 = _mm256_xor_si256(_mm256_set1_epi8((char)s[2]), k2);
            __m256i x3 = _mm256_xor_si256(_mm256_set1_epi8((char)s[3]), k3);
            __m256i met = _mm256_avg_epu8(_mm256_avg_epu8(x0, x1), _mm256_avg_epu8(x2, x3));
            met = _mm256_and_si256(_mm256_srli_epi16(met, 2), c63);
            __m256i mm = _mm256_subs_epu8(c63, met);
            __m256i m0 = _mm256_adds_epu8(lo, met), m1 = _mm256_adds_epu8(hi, mm);
            __m256i m2 = _mm256_adds_epu8(lo, mm), m3 = _mm256_adds_epu8(hi, met);
            __m256i sv0 = _mm256_min_epu8(m0, m1), sv1 = _mm256_min_epu8(m2, m3);
            __m256i d0 = _mm256_cmpeq_epi8(sv0, m1), d1 = _mm256_cmpeq_epi8(sv1, m3);
            __m256i ul = _mm256_unpacklo_epi8(sv0, sv1), uh = _mm256_unpackhi_epi8(sv0, sv1);
            lo = _mm256_permute2x128_si256(ul, uh, 0x20);
            hi = _mm256_permute2x128_si256(ul, uh, 0x31);
            uint32_t a = (uint32_t)_mm256_movemask_epi8(_mm256_unpacklo_epi8(d0, d1));
            uint32_t b = (uint32_t)_mm256_movemask_epi8(_mm256_unpackhi_epi8(d0, d1));
            dec[t] = (uint64_t)(a & 0xFFFFu) | ((uint64_t)(b & 0xFFFFu) << 16) |
                     ((uint64_t)(a >> 16) << 32) | ((uint64_t)(b >> 16) << 48);
        }
        /* renormalise on state 0 only, `>150` (deconvolve.cpp:407-412) */
        if ((uint8_t)_mm256_extract_epi8(lo, 0) > 150) {
            lo = _mm256_subs_epu8(lo, c63);
            hi = _mm256_subs_epu8(hi, c63);
        }
    }
    /* traceback, deconvolve.cpp:416-435 */
    unsigned E = 0;
    const uint64_t *D = dec + 6;
    unsigned n = framebits;
    while (n--) {
        unsigned k = (unsigned)(D[n] >> (E >> 2)) & 1u;
        E = ((E >> 1) | (k << 7)) & 0xFFu;
        out[n >> 3] = (unsigned char)E;
    }
    return 0;
}
