/*
 * oracle/vit_oracle.c -- TEST INFRASTRUCTURE ONLY (see vit_oracle.h).
 *
 * Plain-C restatement of the reference algorithm, written from the integer
 * specification in SURVEY.md Appendix A/B.  Each function cites the reference
 * lines (under /root/reference) whose behaviour it restates.
 */
#include "vit_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define VO_MAXBITS 9216 /* deconvolve.cpp:93 (384*24) */
#define VO_RENORM_THRESHOLD 150 /* viterbi.h:86 */

static const int vo_polys[4] = {109, 79, 83, 109}; /* viterbi-benchmark.cpp:64 */

static inline unsigned parity8(unsigned x) {
    x ^= x >> 4;
    x ^= x >> 2;
    x ^= x >> 1;
    return x & 1u;
}

/* pavgb: deconvolve.cpp:237-239 (_mm_avg_epu8) */
static inline unsigned avg8(unsigned a, unsigned b) { return (a + b + 1u) >> 1; }
static inline unsigned sat255(unsigned a) { return a > 255u ? 255u : a; }

/* branch masks b_j(i) = parity((2i) & poly_j): const.asm:35-49 */
static uint8_t vo_mask[4][32];
static int vo_mask_ready;
static void vo_init_masks(void) {
    if (vo_mask_ready) return;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 32; i++)
            vo_mask[j][i] = parity8((2u * i) & vo_polys[j]) ? 0xFF : 0x00;
    vo_mask_ready = 1;
}

/* One trellis step in natural state order.
 * deconvolve.cpp:233-279 (ButterFly), :335-387 (Butterfly256). */
static inline uint64_t vo_step(const uint8_t s[4], const uint8_t *old,
                               uint8_t *neu) {
    uint64_t d = 0;
    for (int i = 0; i < 32; i++) {
        unsigned x0 = s[0] ^ vo_mask[0][i], x1 = s[1] ^ vo_mask[1][i];
        unsigned x2 = s[2] ^ vo_mask[2][i], x3 = s[3] ^ vo_mask[3][i];
        unsigned metric = avg8(avg8(x0, x1), avg8(x2, x3)) >> 2; /* 0..63 */
        unsigned mm = 63u - metric;
        unsigned m0 = sat255(old[i] + metric), m1 = sat255(old[i + 32] + mm);
        unsigned m2 = sat255(old[i] + mm), m3 = sat255(old[i + 32] + metric);
        /* min + cmpeq(survivor, m1): tie -> decision 1 */
        unsigned d0 = m1 <= m0, d1 = m3 <= m2;
        neu[2 * i] = (uint8_t)(d0 ? m1 : m0);
        neu[2 * i + 1] = (uint8_t)(d1 ? m3 : m2);
        d |= (uint64_t)d0 << (2 * i);
        d |= (uint64_t)d1 << (2 * i + 1);
    }
    return d;
}

/* deconvolve.cpp:407-412 Renormalize256 (C path `>`); decon_avx2.asm:97,114
 * (`jb` => `>=`) selectable for the A.6 experiment. */
static inline void vo_renorm(uint8_t *m, int ge) {
    int hit = ge ? (m[0] >= VO_RENORM_THRESHOLD) : (m[0] > VO_RENORM_THRESHOLD);
    if (hit)
        for (int s = 0; s < 64; s++) m[s] = m[s] > 63 ? (uint8_t)(m[s] - 63) : 0;
}

/* deconvolve.cpp:416-435 ChainBack / chainback.inc:18-41 */
static void vo_chainback(unsigned framebits, const uint64_t *dec,
                         unsigned char *out) {
    unsigned E = 0;
    const uint64_t *D = dec + 6;
    unsigned n = framebits;
    while (n--) {
        unsigned k = (unsigned)(D[n] >> (E >> 2)) & 1u;
        E = ((E >> 1) | (k << 7)) & 0xFFu;
        out[n >> 3] = (unsigned char)E;
    }
}

static int vo_decode_core_t(unsigned framebits, const uint32_t *s32,
                            const uint8_t *s8, unsigned char *out, int ge, uint8_t *trace0);
static int vo_decode_core(unsigned framebits, const uint32_t *s32,
                          const uint8_t *s8, unsigned char *out, int ge) {
    return vo_decode_core_t(framebits, s32, s8, out, ge, NULL);
}
/* trace0 (optional, framebits+6 entries): the metric of state 0 after every trellis step, after the renormalisation
 * where there is one - what tests/test_oracle_kat.py checks against a hand-derived trajectory */
static int vo_decode_core_t(unsigned framebits, const uint32_t *s32,
                            const uint8_t *s8, unsigned char *out, int ge, uint8_t *trace0) {
    if (framebits > VO_MAXBITS) return 1;
    vo_init_masks();
    uint64_t dec[VO_MAXBITS + 6]; /* deconvolve.cpp:93,127: on the stack */
    uint8_t a[64], b[64];
    /* const.asm:19-25: state 0 = 0, others 63 */
    a[0] = 0;
    for (int s = 1; s < 64; s++) a[s] = 63;
    unsigned nb = (framebits + 6) / 2; /* deconvolve.cpp:126 */
    unsigned t = 0;
    for (unsigned it = 0; it < nb; it++) {
        uint8_t sy[4];
        for (int half = 0; half < 2; half++, t++) {
            for (int j = 0; j < 4; j++)
                sy[j] = s32 ? (uint8_t)(s32[4 * t + j] & 0xFF) : s8[4 * t + j];
            if (half == 0) {
                dec[t] = vo_step(sy, a, b);
                if (trace0) trace0[t] = b[0];
            } else {
                dec[t] = vo_step(sy, b, a);
            }
        }
        vo_renorm(a, ge);
        if (trace0) trace0[t - 1] = a[0];
    }
    vo_chainback(framebits, dec, out);
    return 0;
}

int vo_deconvolve(unsigned framebits, const uint32_t *symbols, int unused,
                  unsigned char *out) {
    (void)unused; /* deconvolve.cpp:447-526 never read inputLength */
    return vo_decode_core(framebits, symbols, NULL, out, 0);
}
int vo_deconvolve_opt(unsigned framebits, const uint32_t *symbols,
                      unsigned char *out, int ge) {
    return vo_decode_core(framebits, symbols, NULL, out, ge);
}
int vo_deconvolve_u8(unsigned framebits, const uint8_t *symbols,
                     unsigned char *out) {
    return vo_decode_core(framebits, NULL, symbols, out, 0);
}
/* the MASM twins' comparator (decon_avx2.asm:97,114: `cmp sil,150 ; jb mainloop`) */
int vo_deconvolve_u8_ge(unsigned framebits, const uint8_t *symbols,
                        unsigned char *out) {
    return vo_decode_core(framebits, NULL, symbols, out, 1);
}

int vo_trace_state0_u8(unsigned framebits, const uint8_t *symbols, int ge, uint8_t *trace0) {
    unsigned char out[(VO_MAXBITS + 7) / 8];
    return vo_decode_core_t(framebits, NULL, symbols, out, ge, trace0);
}

/* ---- batch drivers --------------------------------------------------------- */
typedef int (*vo_dec_fn)(unsigned, const uint8_t *, unsigned char *);
struct vo_job {
    vo_dec_fn fn;
    unsigned framebits;
    const uint8_t *sym;
    unsigned char *out;
    long f0, f1;
};
static void *vo_worker(void *p) {
    struct vo_job *j = (struct vo_job *)p;
    size_t ssz = 4u * (j->framebits + 6), osz = (j->framebits + 7) / 8; /* a partial last byte is written too (ChainBack: out[n>>3]) */
    for (long f = j->f0; f < j->f1; f++)
        j->fn(j->framebits, j->sym + ssz * f, j->out + osz * f);
    return NULL;
}
static int vo_batch(vo_dec_fn fn, unsigned framebits, const uint8_t *sym,
                    unsigned char *out, long nframes, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    if (nthreads == 1) {
        struct vo_job j = {fn, framebits, sym, out, 0, nframes};
        vo_worker(&j);
        return 0;
    }
    pthread_t th[256];
    struct vo_job jobs[256];
    long per = (nframes + nthreads - 1) / nthreads;
    int started = 0;
    for (int i = 0; i < nthreads; i++) {
        long f0 = per * i, f1 = f0 + per > nframes ? nframes : f0 + per;
        if (f0 >= f1) break;
        jobs[i] = (struct vo_job){fn, framebits, sym, out, f0, f1};
        pthread_create(&th[i], NULL, vo_worker, &jobs[i]);
        started++;
    }
    for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
    return 0;
}
int vo_decode_batch_u8(unsigned framebits, const uint8_t *symbols,
                       unsigned char *out, long nframes, int nthreads) {
    return vo_batch(vo_deconvolve_u8, framebits, symbols, out, nframes, nthreads);
}
int vo_decode_batch_u8_opt(unsigned framebits, const uint8_t *symbols,
                           unsigned char *out, long nframes, int nthreads,
                           int ge_threshold) {
    return vo_batch(ge_threshold ? vo_deconvolve_u8_ge : vo_deconvolve_u8,
                    framebits, symbols, out, nframes, nthreads);
}
int vo_decode_batch_avx2_u8(unsigned framebits, const uint8_t *symbols,
                            unsigned char *out, long nframes, int nthreads) {
    if (!vo_has_avx2()) return -1;
    return vo_batch(vo_deconvolve_avx2_u8, framebits, symbols, out, nframes,
                    nthreads);
}

/* ---- vector builders -------------------------------------------------------- */

/* viterbi-benchmark.cpp:304-311 */
void vo_encode(unsigned framebits, const uint8_t *bits, uint8_t *hard) {
    unsigned sr = 0;
    for (unsigned i = 0; i < framebits + 6; i++) {
        unsigned bit = i < framebits ? (bits[i] & 1u) : 0u;
        sr = ((sr << 1) | bit) & 0xFFu;
        for (int j = 0; j < 4; j++) hard[4 * i + j] = (uint8_t)parity8(sr & vo_polys[j]);
    }
}

uint64_t vo_xorshift64(uint64_t *state) {
    uint64_t x = *state;
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    *state = x;
    return x;
}
void vo_fill_uniform(uint64_t *state, uint8_t *sym, long n) {
    for (long i = 0; i < n; i++) sym[i] = (uint8_t)((vo_xorshift64(state) >> 11) & 255u);
}
static double vo_uniform01(uint64_t *state) {
    return (double)(vo_xorshift64(state) >> 11) * (1.0 / 9007199254740992.0);
}
/* Box-Muller; recipe mirrors viterbi-benchmark.cpp:637-670 but is seeded and
 * portable (the reference uses MSVC rand()). */
static double vo_gauss(uint64_t *state) {
    double u1, u2;
    do {
        u1 = vo_uniform01(state);
    } while (u1 <= 0.0);
    u2 = vo_uniform01(state);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
void vo_make_noisy_frame(uint64_t *state, unsigned framebits, double ebn0_db,
                         uint8_t *sym, uint8_t *bits_out) {
    /* viterbi-benchmark.cpp:293-294 */
    double esn0 = ebn0_db + 10.0 * log10(1.0 / 4.0);
    double gain = 1.0 / sqrt(0.5 / pow(10.0, esn0 / 10.0));
    unsigned sr = 0;
    for (unsigned i = 0; i < framebits + 6; i++) {
        unsigned bit = i < framebits ? (unsigned)(vo_xorshift64(state) >> 63) : 0u;
        if (bits_out && i < framebits) bits_out[i] = (uint8_t)bit;
        sr = ((sr << 1) | bit) & 0xFFu;
        for (int j = 0; j < 4; j++) {
            int hard = (int)parity8(sr & vo_polys[j]);
            double v = 127.5 + 32.0 * ((hard ? gain : -gain) + vo_gauss(state));
            int sample = (int)v; /* C truncation as in addnoise() :664 */
            if (sample < 0) sample = 0;
            else if (sample > 255) sample = 255;
            sym[4 * i + j] = (uint8_t)sample;
        }
    }
}
uint64_t vo_fnv1a64(const uint8_t *p, long n) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (long i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ULL;
    }
    return h;
}

/* ---- Reed-Solomon ----------------------------------------------------------- */
#define C_NN 255 /* viterbi.h:95 */
#define C_GFPOLY 285 /* viterbi.h:96 */
#define C_NROOTS 10 /* viterbi.h:97 */
#define PAD 135 /* rschecksf.cpp:45 */

static uint8_t g_ato[768], g_iof[256];
static int g_rs_ready;

/* dllmain.cpp:124-146 CreateLookupTables */
void vo_rs_tables(uint8_t *ato_mod, uint8_t *index_of) {
    uint8_t alpha[256];
    int sr = 1;
    index_of[0] = C_NN;
    alpha[C_NN] = 0;
    for (int i = 0; i < C_NN; i++) {
        index_of[sr] = (uint8_t)i;
        alpha[i] = (uint8_t)sr;
        sr <<= 1;
        if (sr & 256) sr ^= C_GFPOLY;
        sr &= C_NN;
    }
    for (int i = 0; i < 768; i++) ato_mod[i] = alpha[i % 255];
}
static void vo_rs_init(void) {
    if (!g_rs_ready) {
        vo_rs_tables(g_ato, g_iof);
        g_rs_ready = 1;
    }
}
/* rschecksf.cpp:50-52 */
static inline unsigned mod255(unsigned x) { return (x * 0x1010102u) >> 24; }

/* rschecksf.cpp:199-377 DECODE_RS.  Arrays are 16 bytes like the reference's
 * XMM-sized locals; entries the reference leaves as stack garbage (root[11..15]
 * copied into lambda[11..15]) are never read, so zero-filling is equivalent. */
int vo_decode_rs(uint32_t *data) {
    vo_rs_init();
    const uint8_t *ato = g_ato, *iof = g_iof;
    uint8_t root[16] = {0}, lambda[16], s[16], b[16];
    unsigned q, tmp, num1, num2, den, discr_r;
    int el, deg_lambda, deg_omega, syn_error, count, r, i, j;

    memset(s, (int)(data[0] & 0xFF), 16); /* :210 */
    for (j = 1; j < C_NN - PAD; j++)      /* :212-219 */
        for (i = 0; i < C_NROOTS; i++) {
            if (s[i] == 0)
                s[i] = (uint8_t)data[j];
            else
                s[i] = (uint8_t)(data[j] ^ ato[iof[s[i]] + i]);
        }
    syn_error = 0; /* :222-230 */
    for (i = 0; i < C_NROOTS; i++) syn_error |= s[i];
    if (!syn_error) return 0;
    for (i = 0; i <= C_NROOTS; i++) s[i] = iof[s[i]]; /* :232-233 */

    memset(b, 0xFF, 16); /* :188-194,235-236 */
    b[0] = 0;
    memset(lambda, 0, 16);
    lambda[0] = 1;

    r = el = 0; /* :240-284 Berlekamp-Massey */
    while (++r <= C_NROOTS) {
        discr_r = 0;
        for (i = 0; i < r; i++)
            if (lambda[i] != 0 && s[r - i - 1] != C_NN)
                discr_r ^= ato[iof[lambda[i]] + s[r - i - 1]];
        discr_r = iof[discr_r];
        if (discr_r == C_NN) {
            memmove(b + 1, b, 15); /* _mm_slli_si128(b,1) :252 */
            b[0] = C_NN;
        } else {
            root[0] = lambda[0];
            for (i = 0; i < C_NROOTS; i++) {
                root[i + 1] = lambda[i + 1];
                if (b[i] != C_NN) root[i + 1] ^= ato[discr_r + b[i]];
            }
            if (2 * el <= r - 1) {
                el = r - el;
                for (i = 0; i <= C_NROOTS; i++)
                    b[i] = (lambda[i] == 0)
                               ? C_NN
                               : (uint8_t)mod255(iof[lambda[i]] - discr_r + C_NN);
            } else {
                memmove(b + 1, b, 15);
                b[0] = C_NN;
            }
            memcpy(lambda, root, 16);
        }
    }

    deg_lambda = 0; /* :287-293 */
    for (i = 0; i < C_NROOTS + 1; i++) {
        lambda[i] = iof[lambda[i]];
        if (lambda[i] != C_NN) deg_lambda = i;
    }

    memcpy(b, lambda, 16); /* :296-320 Chien */
    count = 0;
    for (i = 1; i <= C_NN; i++) {
        q = 1;
        for (j = deg_lambda; j > 0; j--)
            if (b[j] != C_NN) {
                b[j] = (uint8_t)mod255(b[j] + j);
                q ^= ato[b[j]];
            }
        if (q != 0) continue;
        root[count] = (uint8_t)i;
        if (++count == deg_lambda) break;
    }
    if (deg_lambda != count) return -1; /* :325-326 */

    deg_omega = deg_lambda - 1; /* :331-341 */
    for (i = 0; i <= deg_omega; i++) {
        tmp = 0;
        for (j = i; j >= 0; j--)
            if (s[i - j] != C_NN && lambda[j] != C_NN)
                tmp ^= ato[s[i - j] + lambda[j]];
        b[i] = iof[tmp];
    }

    for (j = count - 1; j >= 0; j--) { /* :346-374 Forney */
        if (root[j] < PAD + 1) continue;
        num1 = 0;
        for (i = deg_omega; i >= 0; i--)
            if (b[i] != C_NN) num1 ^= ato[mod255(b[i] + i * root[j])];
        if (!num1) continue;
        num2 = ato[C_NN - root[j]];
        den = 0;
        int top = deg_lambda < C_NROOTS - 1 ? deg_lambda : C_NROOTS - 1;
        for (i = top & ~1; i >= 0; i -= 2)
            if (lambda[i + 1] != C_NN)
                den ^= ato[mod255(lambda[i + 1] + i * root[j])];
        tmp = (iof[num1] + iof[num2]) + (C_NN - iof[den]);
        data[root[j] - 1 - PAD] ^= ato[tmp]; /* ATO_MOD_SIZE == 768 branch */
    }
    return count;
}

/* rschecksf.cpp:65-93 */
int vo_rs_check_superframe(const unsigned char *p, int startIx, unsigned RSDims,
                           unsigned char *outVector) {
    (void)startIx; /* :69 */
    int errors = 0;
    uint32_t blk[128];
    for (unsigned j = 0; j < RSDims; j++) {
        for (unsigned k = 0; k < 120; k++) blk[k] = p[j + (size_t)k * RSDims];
        int res = vo_decode_rs(blk);
        if (res == -1) return -1; /* later columns stay untouched :85-88 */
        errors += res;
        for (unsigned k = 0; k < 110; k++)
            outVector[j + (size_t)k * RSDims] = (unsigned char)blk[k];
    }
    return errors;
}

/* systematic encoder for building valid test codewords */
void vo_rs_encode(const uint8_t *msg, uint8_t *cw) {
    vo_rs_init();
    /* g(x) = prod_{i=0..9} (x + alpha^i), g[0] = x^10 coefficient */
    uint8_t g[11] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int deg = 0;
    for (int i = 0; i < C_NROOTS; i++) {
        /* multiply by (x + alpha^i): coefficients stored highest first */
        uint8_t ng[11] = {0};
        for (int k = 0; k <= deg; k++) {
            ng[k] ^= g[k];
            if (g[k]) ng[k + 1] ^= g_ato[g_iof[g[k]] + i];
        }
        deg++;
        memcpy(g, ng, 11);
    }
    uint8_t rem[10] = {0};
    for (int k = 0; k < 110; k++) {
        uint8_t fb = (uint8_t)(msg[k] ^ rem[0]);
        memmove(rem, rem + 1, 9);
        rem[9] = 0;
        if (fb)
            for (int m = 0; m < 10; m++)
                if (g[m + 1]) rem[m] ^= g_ato[g_iof[fb] + g_iof[g[m + 1]]];
    }
    memcpy(cw, msg, 110);
    memcpy(cw + 110, rem, 10);
}
