/* tests/tools/merge_depth.c -- TEST INFRASTRUCTURE (measurement tool; includes the oracle's source).
 * How many steps back until a traceback started in state 0 (mode < 10) or in the best-metric state (mode >= 10) meets the
 * true survivor path, for reference-style noise at Eb/N0 = 3 dB (mode 1), 0 dB (3), uniform random bytes (0) and random hard
 * decisions (2).  Behind profiles/r03_merge_depth.txt and the warm-up constants of csrc/vit_pk.hip.
 * build: gcc -O2 -I oracle -o /tmp/merge_depth tests/tools/merge_depth.c oracle/vit_avx2.c -lm -lpthread */
#include "../../oracle/vit_oracle.c"
#include <stdio.h>
// merge depth statistics: trace from state 0 at step t and compare with the true survivor path
int main(int argc, char** argv) {
    int mode = argc > 1 ? atoi(argv[1]) : 0; int dm = mode % 10;  // 0 uniform bytes, 1 noisy 3dB, 2 hard random, 3 noisy 0 dB
    unsigned fb = 3072;
    static uint8_t sym[4 * (9216 + 6)];
    uint64_t st = 12345;
    long hist[400] = {0}; long total = 0;
    vo_init_masks();
    for (int frame = 0; frame < 200; frame++) {
        if (dm == 0) vo_fill_uniform(&st, sym, 4 * (fb + 6));
        else if (dm == 1) vo_make_noisy_frame(&st, fb, 3.0, sym, NULL);
        else if (dm == 3) vo_make_noisy_frame(&st, fb, 0.0, sym, NULL);
        else { vo_fill_uniform(&st, sym, 4 * (fb + 6)); for (unsigned i = 0; i < 4 * (fb + 6); i++) sym[i] = (sym[i] & 1) ? 255 : 0; }
        static uint64_t dec[9216 + 6]; static uint8_t best[9216+6];
        uint8_t a[64], b[64];
        a[0] = 0; for (int s = 1; s < 64; s++) a[s] = 63;
        unsigned t = 0;
        for (unsigned it = 0; it < (fb + 6) / 2; it++) {
            for (int half = 0; half < 2; half++, t++) {
                uint8_t* nw = half==0 ? b : a; if (half == 0) dec[t] = vo_step(sym + 4 * t, a, b); else dec[t] = vo_step(sym + 4 * t, b, a); { int bi=0; for(int q=1;q<64;q++) if(nw[q]<nw[bi]) bi=q; best[t]=bi; }
            }
            vo_renorm(a, 0);
        }
        // true path states: state after step t (t = fb+5 -> 0): S[t]
        static unsigned S[9216 + 7]; 
        unsigned s = 0; // state at end (after last step) = 0
        for (int tt = (int)fb + 5; tt >= 0; tt--) { S[tt + 1] = s; unsigned k = (dec[tt] >> s) & 1; s = (s >> 1) | (k << 5); }
        S[0] = s;
        for (int t0 = 400; t0 + 1 < (int)fb; t0 += 37) {
            unsigned z = (mode>=10)? best[t0] : 0; int depth = 0; int tt = t0;
            while (tt >= 0 && z != S[tt + 1]) { unsigned k = (dec[tt] >> z) & 1; z = (z >> 1) | (k << 5); tt--; depth++; }
            if (depth > 399) depth = 399;
            hist[depth]++; total++;
        }
    }
    long cum = 0; int marks[] = {10, 20, 25, 30, 40, 50, 60, 80, 100, 150, 200, 300};
    for (int m = 0, d = 0; m < 12; m++) { for (; d <= marks[m]; d++) cum += hist[d]; printf("P(depth > %3d) = %.4f\n", marks[m], 1.0 - (double)cum / total); }
    return 0;
}
