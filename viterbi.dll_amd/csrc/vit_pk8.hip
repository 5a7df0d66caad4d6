// vit_pk8.hip -- packed K=7 r=1/4 Viterbi decoder for gfx950: EIGHT frames per wavefront, two wavefronts per SIMD.
//
// STATUS: an experiment that is kept, tested and selectable (vit_set_kernel(4) / VITERBI_AMD_PK8=1) but NOT the default.
// Round-3 question: does twice the frames per wave pay?  vit_pk.hip (4 frames per wave, 4 waves per SIMD)
// is bound by VALU issue, and part of its instructions does not scale with the frames in a wave (profiles/r03_ab_diag.txt:
// renormalisation constant 7 %, traceback - mostly its speculative warm-up - 11 %, per-workgroup set-up 3 %).  This kernel
// carries eight frames for the same per-wave work of that kind and moves no data in one trellis step of five:
// 3671 VALU instructions per frame instead of 4162 (-11.8 %, PMC; the shipped kernel has since come down to 3975 with its
// fast traceback form).  Measured on the 65536-frame FIC batch it is 9 % SLOWER than the shipped kernel was at the time
// (0.505 vs 0.461 ms, profiles/r03_ab_pk8.txt; 0.442 ms at the end of round 3): 18.3 KB of LDS and ~210 VGPRs per wave allow two waves per SIMD,
// and two instruction streams do not hide the LDS round trips of the exchange (ACS phase: 94 % VALU busy after the
// two-stage software pipeline below, 0.420 vs 0.409 ms without traceback) nor, above all, the dependent LDS reads of the
// traceback (one chain per wave: 0.085 ms against 0.052 ms for twice the waves in vit_pk.hip).
//
// Layout.  A wave = 4 rows of 16 lanes; a row owns a frame pair (the two 16-bit halves of a VGPR).  The 64 path metrics of
// a frame sit in FOUR registers N[x][y] x 16 lanes: a state's six bits are held by the four lane bits L3..L0 and the two
// register bits X, Y.  Time runs in cycles of five steps with ONE active register bit R (X in even cycles, Y in odd ones):
//     step p = 0     : butterfly on R - R's bit has aged into s5 while R lay dormant: nothing moves;
//     step p = 1..4  : butterfly on R after register bit R and lane bit J = 4 - p have changed places (the transposition of
//                      vit_pk.hip, for both values of the dormant bit; all inside a 16-lane row, so it is DPP or ds_swizzle);
// afterwards the dormant bit has aged into s5 and becomes the active one.  A butterfly (i, i+32) -> (2i, 2i+1) reads and
// writes the two registers that differ in R, in its own lane.  State 0 is always lane 0 of the row, register N[0][0].
// (tests/tools/emulate_pk8.py replays exactly this map, the table classes and the traceback below against the oracle.)
//
// Arithmetic, branch-metric table, decision history (sign bits of m0-m1 / m2-m3, seven-copy insertion) are those of
// vit_pk.hip; the table holds 16 steps x 4 rows (2 KB), a decision block is 64 lanes x 16 B.
// Per segment of 49 blocks the first 32 stay in VGPRs (4 x 32 dwords), 16 go to LDS and the last one lands on the dead
// table: 18.3 KB of LDS per wave, 8 waves per CU = 2 per SIMD, up to 256 VGPRs each.
//
// Traceback: the blocked speculative scheme of vit_pk.hip with lane = (frame, block of BL steps), 8 blocks per frame, so
// a part costs BL + 30 steps for EIGHT frames (BL = 40 for a FIC frame's 272-step parts) instead of 20 + 30 for four.
// The path is tracked in complemented physical coordinates P = l4c<<4 | xc<<3 | yc<<2, which is the byte offset of the
// history word inside its row's 256 bytes of a decision block.
//
// Replaces, from scratch, the same reference code as vit_pk.hip: decon_avx2 / Butterfly256 (deconvolve.cpp:334-387,
// 514-526), Load8Syms256 (:219-228), Renormalize256 (:407-412; decon_avx2.asm:94-118 for the >= mode), ChainBack
// (:416-435), chainback.inc:18-41, const.asm:19-63.
#include "vit_internal.h"
#include "vit_pk_dev.h"

namespace {

constexpr u32 P8_TAB_BYTES = 2048;  // 16 steps x 4 rows x 8 classes x 4 B (M of both frames of the row)
constexpr u32 P8_DEC_BLOCK = 1024;  // 16 steps of decisions: 64 lanes x 16 B
constexpr u32 P8_VREG_BLOCKS = 32;  // decision blocks that can stay in VGPRs (4 x 32 dwords)
constexpr u32 P8_DUMP_GROUP = 16;
constexpr u32 P8_SEG_BLOCKS = P8_VREG_BLOCKS + P8_DUMP_GROUP + 1u;  // 49 blocks = 784 steps: one FIC frame
constexpr u32 P8_TB_WARM = 30;      // warm-up steps of a speculative block (multiple of 10)

#ifndef VIT_P8_FENCE
#define VIT_P8_FENCE 0  /* keep the history inserts between a step's swizzles and the selects that consume them */
#endif
#ifndef VIT_P8_SWZ
#define VIT_P8_SWZ 0xB  /* bit J set: lane bit J is exchanged through ds_swizzle (LDS crossbar, no VALU slot for the move),
                           clear: through DPP (bits 3, 2: masked row moves; bits 1, 0: quad_perm + select) */
#endif

// ---- the schedule ------------------------------------------------------------------------------------------------
// PH = t mod 10.  Active register bit: X for PH < 5, Y otherwise; p = PH mod 5; the transposition AFTER the butterfly of
// step t swaps the active register bit with lane bit 3 - p (none for p = 4).
constexpr bool p8_active_x(int ph) { return ph < 5; }
constexpr int p8_swap_bit(int ph) { return (ph % 5) < 4 ? 3 - (ph % 5) : -1; }

// Table class of the butterflies of a lane at phase PH, for the dormant register bit o = 0 / 1.  State bit k of the
// predecessor sits (before step t) at: s5 -> active register, and with p = PH mod 5, q = the other bits in age order:
// the bit that will be s5 in d steps is s(5-d); lane bit 3-p' is consumed at phase p'+1 ... written out from the
// construction in emulate_pk8.py: at p = 0: s4->L3 s3->L2 s2->L1 s1->L0 s0->dormant; every butterfly makes the active
// register's bit the new s0 and ages the others; the transposition then parks that new s0 in the lane bit it swapped with.
struct P8Map {
    int pos[6];  // position of state bit k: 0..3 lane bit, 4 = X, 5 = Y
};
constexpr P8Map p8_map(int ph) {
    P8Map m = {{5, 0, 1, 2, 3, 4}};  // t = 0: s0->Y s1->L0 s2->L1 s3->L2 s4->L3 s5->X
    for (int t = 0; t < ph; t++) {
        const int R = p8_active_x(t % 10) ? 4 : 5, J = p8_swap_bit(t % 10);
        P8Map n = {{R, m.pos[0], m.pos[1], m.pos[2], m.pos[3], m.pos[4]}};
        if (J >= 0)
            for (int k = 0; k < 6; k++) n.pos[k] = n.pos[k] == R ? J : n.pos[k] == J ? R : n.pos[k];
        m = n;
    }
    return m;
}
template <int PH>
DEV u32 p8_class(u32 l4, u32 o) {
    constexpr P8Map m = p8_map(PH);
    static_assert(m.pos[5] == (p8_active_x(PH) ? 4 : 5), "s5 must sit in the active register bit");
    u32 ib[5];
#pragma unroll
    for (int k = 0; k < 5; k++) ib[k] = m.pos[k] < 4 ? (l4 >> m.pos[k]) & 1u : o;  // the only register position left is the dormant one
    return (ib[1] ^ ib[2] ^ ib[4]) | ((ib[0] ^ ib[1] ^ ib[2]) << 1) | ((ib[0] ^ ib[3]) << 2);  // parity((2i)&poly_j), const.asm:27-63
}

struct Lanes8 {
    u32 toff[10][2];  // LDS address of this lane's table entry for step 0 of a block: [phase][dormant bit]
};

// transposition of the active register bit with lane bit J (inside a 16-lane row):
//   A = bit_J(lane) ? N1[lane ^ 2^J] : N0        B = bit_J(lane) ? N1 : N0[lane ^ 2^J]
template <int J>
DEV void exchange8(u32& A, u32& B, u32 N0, u32 N1, u32 lane) {
    if constexpr (((VIT_P8_SWZ >> J) & 1) == 0 && J == 3) {
        A = __builtin_amdgcn_update_dpp(N0, N1, 0x128 /*row_ror:8*/, 0xF, 0xC, false);
        B = __builtin_amdgcn_update_dpp(N1, N0, 0x128, 0xF, 0x3, false);
    } else if constexpr (((VIT_P8_SWZ >> J) & 1) == 0 && J == 2) {
        A = __builtin_amdgcn_update_dpp(N0, N1, 0x114 /*row_shr:4*/, 0xF, 0xA, false);
        B = __builtin_amdgcn_update_dpp(N1, N0, 0x104 /*row_shl:4*/, 0xF, 0x5, false);
    } else {
        const bool hi = (lane >> J) & 1u;
        u32 p0, p1;
        if constexpr (((VIT_P8_SWZ >> J) & 1) == 0) {
            constexpr int qp = J == 1 ? 0x4E /*quad_perm:[2,3,0,1]*/ : 0xB1 /*quad_perm:[1,0,3,2]*/;
            p1 = __builtin_amdgcn_update_dpp(0u, N1, qp, 0xF, 0xF, true);
            p0 = __builtin_amdgcn_update_dpp(0u, N0, qp, 0xF, 0xF, true);
        } else {
            constexpr int pat = 0x1F | ((1 << J) << 10);  // BitMode: src lane = lane ^ 2^J
            p1 = (u32)__builtin_amdgcn_ds_swizzle((int)N1, pat);
            p0 = (u32)__builtin_amdgcn_ds_swizzle((int)N0, pat);
        }
        A = hi ? p1 : N0;
        B = hi ? N1 : p0;
    }
}

// One trellis step for 8 frames (deconvolve.cpp:352-374 in packed u16 form): two butterflies per lane.
// N[x][y]: path metrics; acc[x][y]: decision history of the state that lands in N[x][y].
template <int PH, int J>
DEV void acs_step8(u32 (&N)[2][2], u32 (&acc)[2][2], u32 mt0, u32 mt1, u32 lane, const Consts& C) {
    constexpr bool RX = p8_active_x(PH), ODD = (J & 1) != 0;
    constexpr int SW = p8_swap_bit(PH);
    constexpr bool SWZ = SW >= 0 && ((VIT_P8_SWZ >> (SW < 0 ? 0 : SW)) & 1) != 0;  // this step's transposition goes through ds_swizzle
    static_assert(((PH ^ J) & 1) == 0, "the phase and the step share their parity");
    u32 n0[2], n1[2], x01[2], x23[2];
#pragma unroll
    for (int o = 0; o < 2; o++) {
        const u32 a_ = RX ? N[0][o] : N[o][0], b_ = RX ? N[1][o] : N[o][1];
        const u32 mt = o ? mt1 : mt0;
        // 63 - M per half as ONE 32-bit subtract (see vit_pk.hip: even steps carry the +0xFF00 bias)
        const us2 a = U(a_), b = U(b_), M = U(mt), MM = U((ODD ? 0x003F003Fu : 0xFE40FE3Fu) - mt);
        const us2 m0 = __builtin_elementwise_add_sat(a, M), m1 = __builtin_elementwise_add_sat(b, MM);
        const us2 m2 = __builtin_elementwise_add_sat(a, MM), m3 = __builtin_elementwise_add_sat(b, M);
        n0[o] = W(__builtin_elementwise_min(m0, m1));
        n1[o] = W(__builtin_elementwise_min(m2, m3));
        x01[o] = W(m0 - m1);
        x23[o] = W(m2 - m3);
    }
    // Everything that travels through the LDS crossbar is requested first ...
    u32 z = 0, p0[2] = {0u, 0u}, p1[2] = {0u, 0u};
    if constexpr (ODD) z = (u32)__builtin_amdgcn_ds_swizzle((int)n0[0], 0x10);  // state 0: lane 0 of each 16-lane row, N[0][0]
    if constexpr (SWZ) {
        constexpr int pat = 0x1F | ((1 << (SW < 0 ? 0 : SW)) << 10);  // BitMode: src lane = lane ^ 2^SW
#pragma unroll
        for (int o = 0; o < 2; o++) {
            p1[o] = (u32)__builtin_amdgcn_ds_swizzle((int)n1[o], pat);
            p0[o] = (u32)__builtin_amdgcn_ds_swizzle((int)n0[o], pat);
        }
    }
    // ... and the decision history, which nothing in the metric chain waits for, is inserted while it is under way
    // (two waves per SIMD: a wave that stalls on its own swizzle right after issuing it leaves the SIMD to ONE other wave).
    // History: bits 9..15 of each half of m0 - m1 are seven copies of its sign (= NOT decision); step j of the block ends
    // up at bit j of its half with two 32-bit shifts per 16 steps (vit_pk.hip, acs_step).
#pragma unroll
    for (int o = 0; o < 2; o++) {
        constexpr int pos = J < 2 ? 14 + J : J < 9 ? 7 + J : J;
        constexpr u32 mask = 0x00010001u << pos;
        u32& h0 = RX ? acc[0][o] : acc[o][0];
        u32& h1 = RX ? acc[1][o] : acc[o][1];
        if constexpr (J == 2 || J == 9) {
            h0 >>= 7;
            h1 >>= 7;
        }
        h0 = bfi(mask, x01[o], h0);
        h1 = bfi(mask, x23[o], h1);
    }
#if VIT_P8_FENCE
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int o = 0; o < 2; o++) {
        u32& A = RX ? N[0][o] : N[o][0];
        u32& B = RX ? N[1][o] : N[o][1];
        if constexpr (SWZ) {
            const bool hi = (lane >> (SW < 0 ? 0 : SW)) & 1u;
            A = hi ? p1[o] : n0[o];
            B = hi ? n1[o] : p0[o];
        } else if constexpr (SW >= 0) {
            exchange8<(SW < 0 ? 0 : SW)>(A, B, n0[o], n1[o], lane);
        } else {
            A = n0[o];
            B = n1[o];
        }
    }
    if constexpr (ODD) {
        // Renormalize256 (deconvolve.cpp:407-412; C.rc selects > 150 or the MASM decoders' >= 150): ONE constant for the
        // row's two frames, shared by all four registers; lands in the 0-based representation
        const u32 w = z + C.rc;
        const u32 t = (w >> 15) & 0x00010001u;
        const u32 K = t * 63u + C.hi;
#pragma unroll
        for (int x = 0; x < 2; x++)
#pragma unroll
            for (int y = 0; y < 2; y++) N[x][y] = W(__builtin_elementwise_sub_sat(U(N[x][y]), U(K)));
    }
}

// table entry of this lane for step J of the current block: an LDS read by absolute address (no base register to add)
DEV u32 tab_read(u32 addr) { return *reinterpret_cast<const __attribute__((address_space(3))) u32*>(addr); }

// ---- the same step as a two-stage software pipeline over the lane's two butterflies ("chains" o = 0, 1) --------------
// Inside a cycle the two chains do not touch each other's registers (only the renormalisation constant is shared), and a
// wave issues in order: written one step after the other, both chains request their swizzles at about the same time and
// then both wait for them - with two waves per SIMD that wait is exposed.  Here the chains run HALF A STEP APART:
//     A(o, t): adds, mins, differences of chain o; requests its swizzles (and state 0's broadcast on odd steps, chain 0)
//     B(o, t): history inserts, then the selects that consume the swizzles, renormalisation subtract
// in the order  A(0,t) A(1,t) | B(0,t) A(0,t+1) | B(1,t) A(1,t+1) | ...  so that between a chain's request and its use
// lies the other chain's whole stage pair.  A step that opens a new cycle (p = 0) pairs registers of BOTH old chains: its
// two A stages come after both B stages of the step before (which has no transposition to wait for anyway).
struct Half8 {
    u32 n0, n1, x01, x23, p0, p1;
};
template <int PH, int J, int O>
DEV void stage_a(const u32 (&N)[2][2], u32 mt, Half8& h, u32& z) {
    constexpr bool RX = p8_active_x(PH), ODD = (J & 1) != 0;
    constexpr int SW = p8_swap_bit(PH);
    constexpr bool SWZ = SW >= 0 && ((VIT_P8_SWZ >> (SW < 0 ? 0 : SW)) & 1) != 0;
    const u32 a_ = RX ? N[0][O] : N[O][0], b_ = RX ? N[1][O] : N[O][1];
    const us2 a = U(a_), b = U(b_), M = U(mt), MM = U((ODD ? 0x003F003Fu : 0xFE40FE3Fu) - mt);
    const us2 m0 = __builtin_elementwise_add_sat(a, M), m1 = __builtin_elementwise_add_sat(b, MM);
    const us2 m2 = __builtin_elementwise_add_sat(a, MM), m3 = __builtin_elementwise_add_sat(b, M);
    h.n0 = W(__builtin_elementwise_min(m0, m1));
    h.n1 = W(__builtin_elementwise_min(m2, m3));
    if constexpr (ODD && O == 0) z = (u32)__builtin_amdgcn_ds_swizzle((int)h.n0, 0x10);  // state 0: lane 0 of each row, N[0][0]
    h.p0 = h.p1 = 0;
    if constexpr (SWZ) {
        constexpr int pat = 0x1F | ((1 << (SW < 0 ? 0 : SW)) << 10);
        h.p1 = (u32)__builtin_amdgcn_ds_swizzle((int)h.n1, pat);
        h.p0 = (u32)__builtin_amdgcn_ds_swizzle((int)h.n0, pat);
    }
    h.x01 = W(m0 - m1);
    h.x23 = W(m2 - m3);
}
template <int PH, int J, int O>
DEV void stage_b(u32 (&N)[2][2], u32 (&acc)[2][2], const Half8& h, u32 z, u32& K, u32 lane, const Consts& C) {
    constexpr bool RX = p8_active_x(PH), ODD = (J & 1) != 0;
    constexpr int SW = p8_swap_bit(PH);
    constexpr bool SWZ = SW >= 0 && ((VIT_P8_SWZ >> (SW < 0 ? 0 : SW)) & 1) != 0;
    constexpr int pos = J < 2 ? 14 + J : J < 9 ? 7 + J : J;
    constexpr u32 mask = 0x00010001u << pos;
    u32& h0 = RX ? acc[0][O] : acc[O][0];
    u32& h1 = RX ? acc[1][O] : acc[O][1];
    if constexpr (J == 2 || J == 9) {
        h0 >>= 7;
        h1 >>= 7;
    }
    h0 = bfi(mask, h.x01, h0);
    h1 = bfi(mask, h.x23, h1);
    u32& A = RX ? N[0][O] : N[O][0];
    u32& B = RX ? N[1][O] : N[O][1];
    if constexpr (SWZ) {
        const bool hi = (lane >> (SW < 0 ? 0 : SW)) & 1u;
        A = hi ? h.p1 : h.n0;
        B = hi ? h.n1 : h.p0;
    } else if constexpr (SW >= 0) {
        exchange8<(SW < 0 ? 0 : SW)>(A, B, h.n0, h.n1, lane);
    } else {
        A = h.n0;
        B = h.n1;
    }
    if constexpr (ODD) {
        if constexpr (O == 0) {
            const u32 w = z + C.rc;
            const u32 t = (w >> 15) & 0x00010001u;
            K = t * 63u + C.hi;
        }
        A = W(__builtin_elementwise_sub_sat(U(A), U(K)));
        B = W(__builtin_elementwise_sub_sat(U(B), U(K)));
    }
}
#define P8_FENCE() __builtin_amdgcn_sched_barrier(VIT_P8_PIPE_MASK)
#ifndef VIT_P8_PIPE_MASK
#define VIT_P8_PIPE_MASK 0x4  /* what may cross a stage boundary in the scheduler: 0x4 = scalar ALU only */
#endif
template <int V, int J, int JEND>
struct Pipe8 {
    // on entry both A stages of step J are issued (h0, h1, z) and the table values of step J+1 are on their way (nx0, nx1)
    static DEV void run(u32 (&N)[2][2], u32 (&acc)[2][2], const Half8& h0, const Half8& h1, u32 z, u32 nx0, u32 nx1,
                        const Lanes8& L, u32 lane, const Consts& C) {
        constexpr int PH = (6 * V + J) % 10, PN = (6 * V + J + 1) % 10;
        constexpr bool LAST = J + 1 == JEND;
        constexpr bool OPENS = !LAST && (PN % 5) == 0;  // step J+1 opens a new cycle
        u32 mm0 = 0, mm1 = 0, K = 0, zn = 0;
        Half8 g0 = {}, g1 = {};
        if constexpr (J + 2 < JEND) {
            constexpr int P2 = (6 * V + J + 2) % 10;
            mm0 = tab_read(L.toff[P2][0] + (J + 2) * 128);
            mm1 = tab_read(L.toff[P2][1] + (J + 2) * 128);
        }
        stage_b<PH, J, 0>(N, acc, h0, z, K, lane, C);
        if constexpr (!LAST && !OPENS) stage_a<PN, J + 1, 0>(N, nx0, g0, zn);
        P8_FENCE();
        stage_b<PH, J, 1>(N, acc, h1, z, K, lane, C);
        if constexpr (OPENS) stage_a<PN, J + 1, 0>(N, nx0, g0, zn);
        if constexpr (!LAST) stage_a<PN, J + 1, 1>(N, nx1, g1, zn);
        P8_FENCE();
        if constexpr (!LAST) Pipe8<V, J + 1, JEND>::run(N, acc, g0, g1, zn, mm0, mm1, L, lane, C);
    }
};
template <int V, int N16>
DEV void pipe8v(u32 (&N)[2][2], u32 (&acc)[2][2], const Lanes8& L, u32 lane, const Consts& C) {
    constexpr int P0 = (6 * V) % 10, P1 = (6 * V + 1) % 10;
    const u32 mt0 = tab_read(L.toff[P0][0]), mt1 = tab_read(L.toff[P0][1]);
    const u32 nx0 = tab_read(L.toff[P1][0] + 128), nx1 = tab_read(L.toff[P1][1] + 128);
    Half8 h0, h1;
    u32 z = 0;
    stage_a<P0, 0, 0>(N, mt0, h0, z);
    stage_a<P0, 0, 1>(N, mt1, h1, z);
    P8_FENCE();
    Pipe8<V, 0, N16>::run(N, acc, h0, h1, z, nx0, nx1, L, lane, C);
}

// The table values of step J+1 are requested BEFORE step J is computed: with two waves per SIMD an LDS round trip in
// front of every step's first add would be exposed (the 4-frames-per-wave kernel has four waves to hide it).
template <int V, int J, int JEND>
struct Steps8 {
    static DEV void run(u32 (&N)[2][2], u32 (&acc)[2][2], u32 mt0, u32 mt1, const Lanes8& L, u32 lane, const Consts& C) {
        constexpr int PH = (6 * V + J) % 10;  // t = 16 (5k + V) + J
        u32 nx0 = 0, nx1 = 0;
        if constexpr (J + 1 < JEND) {
            constexpr int PN = (6 * V + J + 1) % 10;
            nx0 = tab_read(L.toff[PN][0] + (J + 1) * 128);
            nx1 = tab_read(L.toff[PN][1] + (J + 1) * 128);
        }
        acs_step8<PH, J>(N, acc, mt0, mt1, lane, C);
        Steps8<V, J + 1, JEND>::run(N, acc, nx0, nx1, L, lane, C);
    }
};
template <int V, int JEND>
struct Steps8<V, JEND, JEND> {
    static DEV void run(u32 (&)[2][2], u32 (&)[2][2], u32, u32, const Lanes8&, u32, const Consts&) {}
};
template <int V, int N16>
DEV void steps8v(u32 (&N)[2][2], u32 (&acc)[2][2], const Lanes8& L, u32 lane, const Consts& C) {
    constexpr int P0 = (6 * V) % 10;
    Steps8<V, 0, N16>::run(N, acc, tab_read(L.toff[P0][0]), tab_read(L.toff[P0][1]), L, lane, C);
}
#ifndef VIT_P8_PIPE
#define VIT_P8_PIPE 1  /* 1: the two chains of a lane run half a step apart (Pipe8); 0: step after step (Steps8) */
#endif
template <int N16>
DEV void steps8(u32 v, u32 (&N)[2][2], u32 (&acc)[2][2], const Lanes8& L, u32 lane, const Consts& C) {
#if VIT_P8_PIPE
    switch (v) {
        case 0: pipe8v<0, N16>(N, acc, L, lane, C); break;
        case 1: pipe8v<1, N16>(N, acc, L, lane, C); break;
        case 2: pipe8v<2, N16>(N, acc, L, lane, C); break;
        case 3: pipe8v<3, N16>(N, acc, L, lane, C); break;
        default: pipe8v<4, N16>(N, acc, L, lane, C); break;
    }
#else
    switch (v) {
        case 0: steps8v<0, N16>(N, acc, L, lane, C); break;
        case 1: steps8v<1, N16>(N, acc, L, lane, C); break;
        case 2: steps8v<2, N16>(N, acc, L, lane, C); break;
        case 3: steps8v<3, N16>(N, acc, L, lane, C); break;
        default: steps8v<4, N16>(N, acc, L, lane, C); break;
    }
#endif
}

typedef u32 v32u __attribute__((ext_vector_type(32)));

// ---- layout ------------------------------------------------------------------------------------------------------
__host__ __device__ inline u32 p8_reg_blocks(u32 nb) {
    if (nb <= P8_DUMP_GROUP + 1u) return 0;
    const u32 r = nb - (P8_DUMP_GROUP + 1u);
    return r < P8_VREG_BLOCKS ? r : P8_VREG_BLOCKS;
}
__host__ __device__ inline u32 p8_img_stride(u32 maxfb) { return ((maxfb + 31u) >> 5) + 2u; }  // dwords per frame
__host__ __device__ inline u32 p8_block_len(u32 span) { return 10u * ((span + 79u) / 80u); }  // 8 blocks cover the span; multiple of the schedule's period
__host__ __device__ inline u32 p8_scratch_words(u32 maxfb) {
    u32 nb = (maxfb + VIT_TAIL + 15u) >> 4;
    if (nb > P8_SEG_BLOCKS) nb = P8_SEG_BLOCKS;
    const u32 tail = (nb - p8_reg_blocks(nb)) * 16u;
    const u32 span = tail > 256u ? tail : 256u;
    return (p8_block_len(span) + 31u) >> 5;
}
struct P8Layout {
    u32 dec_bytes;  // the table starts here
    u32 img_off;    // output bit image
    u32 total;
    u32 maxfb;
};
__host__ __device__ inline P8Layout p8_layout(u32 maxfb) {  // single-segment kernel (nb <= 49)
    const u32 nb = (maxfb + VIT_TAIL + 15u) >> 4;
    P8Layout l;
    l.maxfb = maxfb;
    l.dec_bytes = (nb - p8_reg_blocks(nb) - 1u) * P8_DEC_BLOCK;
    const u32 scratch = 64u * 4u * p8_scratch_words(maxfb), img = 8u * 4u * p8_img_stride(maxfb);
    u32 tabregion = P8_DEC_BLOCK + scratch + img;  // after the ACS the dead table holds the last block, scratch and image
    tabregion = tabregion > P8_TAB_BYTES ? ((tabregion + 15u) & ~15u) : P8_TAB_BYTES;
    l.img_off = l.dec_bytes + P8_DEC_BLOCK + scratch;
    l.total = l.dec_bytes + tabregion;
    return l;
}

// ---- traceback ---------------------------------------------------------------------------------------------------
// Tracked value PC = g<<8 | l4c<<4 | xc<<3 | yc<<2 | h<<1: row, complemented position, frame half = the byte offset of the
// 16 history bits of (frame, position) inside a decision block.  The position is the one the state had right AFTER the
// butterfly of the step being read.  One step back from step t (phase PH = t mod 10, active register bit RB):
//     kb = stored bit (NOT decision) of step t at PC;   RB := kb;   then undo the transposition that followed step t-1
//     (same cycle, p >= 1: swap RB with lane bit 4 - p)  =>  lane bit := kb, RB := old lane bit      [one v_bfi]
//     (p = 0: step t-1 closed the other register's cycle without a transposition)  =>  RB := kb
// (ChainBack's E = (E>>1)|(k<<7), deconvolve.cpp:424-433, seen through the map above.)  State 0 is PC & 0xFC = 252.
constexpr u32 P8_ZERO = 252u;
DEV u32 p8_dec_slot(u32 lane) { return (lane >> 4) * 256u + (15u - (lane & 15u)) * 16u; }  // [row][15 - l4] -> (acc11, acc10, acc01, acc00)

template <int PH>
DEV void tb_step8(u32& PC, u32& kb, u32 x) {
    // x = (t - 16*slot0) << 6 with the LDS address of dec (a multiple of 1024) folded in
    const u32 w = *reinterpret_cast<const __attribute__((address_space(3))) unsigned short*>((x & ~1023u) | PC);
    kb = __builtin_amdgcn_ubfe(w, (x >> 6) & 15u, 1u);
    constexpr int RB = p8_active_x(PH) ? 3 : 2, p = PH % 5;
    if constexpr (p == 0) {
        PC = bfi(1u << RB, kb << RB, PC);
    } else {
        constexpr int JJ = 4 + (4 - p);  // lane bit 4 - p inside PC
        const u32 t = ((PC >> (JJ - RB)) & (1u << RB)) | (kb << JJ);
        PC = bfi((1u << JJ) | (1u << RB), t, PC);
    }
}

// block-relative indices i_from .. i_to, downwards; i_from + 1 and i_to are multiples of 10, T0 = phase of index 0
template <bool RECORD, int T0>
DEV void tb_loop8(u32& PC, u32* scratch, int i_from, int i_to, bool on, u32 i_start, u32 xbase) {
    u32 cur = 0;
    for (int i = i_from; i >= i_to; i -= 10) {
#define TB8_ONE(K)                                                                         \
    {                                                                                      \
        const int ii = i - (K);                                                            \
        u32 kb = 0;                                                                        \
        if (on && (u32)ii <= i_start) tb_step8<(T0 + 9 - (K)) % 10>(PC, kb, xbase + ((u32)ii << 6)); \
        if (RECORD) {                                                                      \
            cur |= kb << (ii & 31);                                                        \
            if ((ii & 31) == 0) {                                                          \
                if (on) scratch[ii >> 5] = cur;                                            \
                cur = 0;                                                                   \
            }                                                                              \
        }                                                                                  \
    }
        TB8_ONE(0) TB8_ONE(1) TB8_ONE(2) TB8_ONE(3) TB8_ONE(4) TB8_ONE(5) TB8_ONE(6) TB8_ONE(7) TB8_ONE(8) TB8_ONE(9)
#undef TB8_ONE
    }
}
template <bool RECORD>
DEV void tb_run8(u32& PC, u32* scratch, int i_from, int i_to, bool on, u32 i_start, u32 xbase, u32 t0) {
    switch (t0) {  // phase of block-relative index 0: ts mod 10, always even (ts = 6 or a multiple of 16)
        case 0: tb_loop8<RECORD, 0>(PC, scratch, i_from, i_to, on, i_start, xbase); break;
        case 2: tb_loop8<RECORD, 2>(PC, scratch, i_from, i_to, on, i_start, xbase); break;
        case 4: tb_loop8<RECORD, 4>(PC, scratch, i_from, i_to, on, i_start, xbase); break;
        case 6: tb_loop8<RECORD, 6>(PC, scratch, i_from, i_to, on, i_start, xbase); break;
        default: tb_loop8<RECORD, 8>(PC, scratch, i_from, i_to, on, i_start, xbase); break;
    }
}

// One traceback part over steps [ts, te) of every frame (te per lane's frame, te_max uniform); decisions of block b at
// dec + (b - slot0)*1024.  Lane = (frame fi = lane>>3, block q = lane&7).  Same scheme as vit_pk.hip's traceback_part:
// speculative blocks start P8_TB_WARM steps early from state 0, are checked against the block above and re-traced until
// nothing changes - the fixed point is the serial chainback.  ORs the decoded bits into img; returns P after step ts.
DEV u32 traceback_part8(const char* dec, u32* scratch, u32* img, u32 fstride, u32 lane, u32 ts, u32 te, u32 te_max, u32 slot0,
                        u32 P_top) {
#ifdef VIT_DIAG_NO_TB
    return P_top;  // timing-only diagnostic build: outputs are wrong
#endif
    const u32 fi = lane >> 3, q = lane & 7u;
    const u32 span = te_max > ts ? te_max - ts : 0u;
    if (span == 0) return P_top;
    const u32 BL = p8_block_len(span);
    const u32 tbase = ts + q * BL;
    const bool has_work = tbase < te;
    const u32 i_last = has_work ? te - 1u - tbase : 0u;
    const u32 q_top = te > ts ? (te - 1u - ts) / BL : 0u;
    const u32 i_warm = BL - 1u + P8_TB_WARM;
    const u32 i_start = i_last < i_warm ? i_last : i_warm;
    const bool fixed = has_work && i_last <= i_warm;  // starts from the true position: never re-traced
    const u32 Cb = (fi >> 1) * 256u + (fi & 1u) * 2u;  // row and half ride along in the tracked position
    const u32 dbase = (u32)(uintptr_t)(const __attribute__((address_space(3))) char*)dec;
    if (dbase & 1023u) __builtin_trap();
    const u32 xbase = ((tbase - slot0 * 16u) << 6) + dbase;
    const u32 t0 = ts % 10u;
    const u32 PC_top = P_top | Cb;

    u32 P = fixed ? PC_top : (P8_ZERO | Cb), P_out = PC_top;
    tb_run8<false>(P, scratch, (int)i_warm, (int)BL, has_work, i_start, xbase, t0);
    u32 P_in = P;
    tb_run8<true>(P, scratch, (int)BL - 1, 0, has_work, i_start, xbase, t0);
    if (has_work) P_out = P;
    for (int pass = 0; pass < 9; pass++) {
        const u32 nxt = __shfl_down(P_out, 1);  // the block above belongs to the same frame
        const u32 new_in = (q < q_top) ? nxt : PC_top;
        const bool changed = has_work && !fixed && new_in != P_in;
        if (!__any(changed)) break;
        if (changed) P_in = new_in;
        P = new_in;
        tb_run8<true>(P, scratch, (int)BL - 1, 0, changed, BL - 1u, xbase, t0);
        if (changed) P_out = P;
    }
    if (has_work) {  // decoded bit index of step t is t - 6; decoded bit = NOT stored bit
        const u32 nvalid = i_last + 1u < BL ? i_last + 1u : BL;
        const u32 nw = (BL + 31u) >> 5;
        for (u32 w = 0; w < nw; w++) {
            const u32 lo = 32u * w;
            const u32 cnt = nvalid > lo ? nvalid - lo : 0u;
            const u32 mask = cnt >= 32u ? 0xFFFFFFFFu : ((1u << cnt) - 1u);
            const u32 val = ~scratch[w] & mask;
            const u32 b0 = tbase - VIT_TAIL + lo;
            const u32 d = b0 >> 5, sft = b0 & 31u;
            if (val) {
                atomicOr(&img[fi * fstride + d], val << sft);
                if (sft) atomicOr(&img[fi * fstride + d + 1u], val >> (32u - sft));
            }
        }
    }
    return __shfl(P_out, (int)(fi * 8u)) & 0xFCu;
}

// ---- the kernel: every frame of the launch fits one segment (framebits <= 778, the FIC fast path) ---------------------
struct FrameP {
    u32 fb;  // 0 = nothing to decode
    size_t soff, ooff;
};
DEV FrameP frame_params(const vit_frame_desc* __restrict__ desc, long long f, long long nframes, u32 framebits_uniform, u32 vmax) {
    FrameP p = {0u, 0, 0};
    if (f < nframes) {
        if (desc) {
            p.fb = desc[f].framebits;
            p.soff = desc[f].sym_offset;
            p.ooff = desc[f].out_offset;
            if (p.fb > vmax || (p.fb & 1u) || (p.soff & 3u)) p.fb = 0;  // not what the launch was sized for / misaligned: skipped
        } else {
            p.fb = framebits_uniform;
            p.soff = (size_t)f * 4u * (framebits_uniform + VIT_TAIL);
            p.ooff = (size_t)f * ((framebits_uniform + 7u) >> 3);
        }
    }
    return p;
}

template <bool SYM32>
__global__ __launch_bounds__(64, 2) void vit_pk8_kernel(const uint8_t* __restrict__ sym, uint8_t* __restrict__ out,
                                                         const vit_frame_desc* __restrict__ desc, u32 framebits_uniform,
                                                         long long nframes, P8Layout lay, u32 vmax, u32 renorm_c) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* dec = lds;                  // [block - R][row][15 - l4] -> (acc11, acc10, acc01, acc00); the last block spills into tab
    char* tab = lds + lay.dec_bytes;  // [tau][row][c] -> M; after the ACS: last block, scratch, image
    u32* img = reinterpret_cast<u32*>(lds + lay.img_off);
    const u32 lane = threadIdx.x;
    const long long f0 = (long long)blockIdx.x * 8;

    // ---- frames: the traceback lane's own frame (fi = lane>>3), the pre-pass lane's two frames (row pp = lane&3) ----
    const FrameP mine = frame_params(desc, f0 + (lane >> 3), nframes, framebits_uniform, vmax);
    u32 maxfb = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const u32 v = (u32)__builtin_amdgcn_readlane((int)mine.fb, k * 8);
        maxfb = v > maxfb ? v : maxfb;
    }
    if (maxfb == 0 || maxfb > lay.maxfb) return;
    const u32 nb = (maxfb + VIT_TAIL + 15u) >> 4, R = p8_reg_blocks(nb);
    const u32 fstride = p8_img_stride(maxfb);
    const u32 T_max = maxfb + VIT_TAIL;

    // ---- ACS lane constants ----
    const u32 l4 = lane & 15u, row = lane >> 4;
    const u32 dslot = p8_dec_slot(lane);
    Lanes8 L;
    const u32 tab_lds = (u32)(uintptr_t)(const __attribute__((address_space(3))) char*)tab;
#define P8_TOFF(PH)                                          \
    L.toff[PH][0] = tab_lds + row * 32u + p8_class<PH>(l4, 0u) * 4u;   \
    L.toff[PH][1] = tab_lds + row * 32u + p8_class<PH>(l4, 1u) * 4u;
    P8_TOFF(0) P8_TOFF(1) P8_TOFF(2) P8_TOFF(3) P8_TOFF(4) P8_TOFF(5) P8_TOFF(6) P8_TOFF(7) P8_TOFF(8) P8_TOFF(9)
#undef P8_TOFF
#pragma unroll
    for (int ph = 0; ph < 10; ph++) asm volatile("" : "+v"(L.toff[ph][0]), "+v"(L.toff[ph][1]));  // addresses, not sums to redo per step
    Consts C;
    C.hi = HI;
    C.rc = renorm_c;
    asm volatile("" : "+v"(C.hi));
    // ---- pre-pass lane constants: lane = (tau = lane>>2, row pp = lane&3) ----
    const u32 tau = lane >> 2, pp = lane & 3u;
    const FrameP fa = frame_params(desc, f0 + 2 * pp, nframes, framebits_uniform, vmax);
    const FrameP fb = frame_params(desc, f0 + 2 * pp + 1, nframes, framebits_uniform, vmax);
    const u32 a_T = fa.fb ? fa.fb + VIT_TAIL : 0u, b_T = fb.fb ? fb.fb + VIT_TAIL : 0u;
    constexpr size_t SB = SYM32 ? 4 : 1;
    const uint8_t* a_sym = sym + SB * fa.soff;
    const uint8_t* b_sym = sym + SB * fb.soff;
    u32 sel[4];
    {
        const u32 hb = (tau & 1u) ? 0x0C000C00u : 0x0D000D00u;  // even step: 0xFF high bytes (= +0xFF00)
#pragma unroll
        for (int k = 0; k < 4; k++) sel[k] = hb | (0x00040000u + 0x00010001u * k);
    }
    const PrepassLane PL = prepass_lane(lane);
    u32 N[2][2] = {{l4 == 0 ? 0u : 0x003F003Fu, 0x003F003Fu}, {0x003F003Fu, 0x003F003Fu}};  // const.asm:19-25 (0-based, step 0 is even)
    u32 acc[2][2] = {{0u, 0u}, {0u, 0u}};
    v32u r00, r01, r10, r11;  // register-resident decisions of blocks [0,R)

    // ---- ACS over the blocks ----
    {
        auto sa = load_step<SYM32>(a_sym, tau, tau < a_T), sb = load_step<SYM32>(b_sym, tau, tau < b_T);
        u32 v = 0;
        for (u32 rb = 0; rb < nb; rb++) {
            __syncthreads();  // every lane is done with the previous table
            prepass(pack_step(sa), pack_step(sb), tab, PL, sel);
            const u32 tn = (rb + 1u) * 16u + tau;
            sa = load_step<SYM32>(a_sym, tn, tn < a_T);  // prefetch the next 16 steps' symbols
            sb = load_step<SYM32>(b_sym, tn, tn < b_T);
            __syncthreads();
            if (rb + 1u == nb && (T_max & 15u) == 6u) {  // every DAB size: the ten padding steps of the last block are not computed
                steps8<6>(v, N, acc, L, lane, C);
#pragma unroll
                for (int x = 0; x < 2; x++)
#pragma unroll
                    for (int y = 0; y < 2; y++) acc[x][y] >>= 7;
            } else {
                steps8<16>(v, N, acc, L, lane, C);
            }
            if (rb < R) {
                r00[rb] = acc[0][0];
                r01[rb] = acc[0][1];
                r10[rb] = acc[1][0];
                r11[rb] = acc[1][1];
            } else {
                if (rb + 1u == nb) __syncthreads();  // the last block lands on the table: all reads done first
                *reinterpret_cast<uint4*>(dec + (rb - R) * P8_DEC_BLOCK + dslot) = make_uint4(acc[1][1], acc[1][0], acc[0][1], acc[0][0]);
            }
            v = v == 4 ? 0 : v + 1;
        }
    }
    __syncthreads();
    for (u32 i = lane; i < 8u * fstride; i += 64u) img[i] = 0;  // the image aliases the dead table region

    // ---- traceback, last part first: lane = (frame fi, block q) ----
    const u32 t_T = mine.fb ? mine.fb + VIT_TAIL : 0u;
    u32* scratch = reinterpret_cast<u32*>(tab + P8_DEC_BLOCK) + lane * p8_scratch_words(maxfb);
    const u32 t_lo = R * 16u;
    u32 P_part = traceback_part8(dec, scratch, img, fstride, lane, t_lo > VIT_TAIL ? t_lo : VIT_TAIL, t_T, T_max, R, P8_ZERO);
    for (u32 g1 = R; g1 > 0;) {
        const u32 g0 = g1 > P8_DUMP_GROUP ? g1 - P8_DUMP_GROUP : 0u;  // group = blocks [g0, g1)
        __syncthreads();
#pragma unroll
        for (u32 b = 0; b < P8_VREG_BLOCKS; b++)
            if (b >= g0 && b < g1)
                *reinterpret_cast<uint4*>(dec + (b - g0) * P8_DEC_BLOCK + dslot) = make_uint4(r11[b], r10[b], r01[b], r00[b]);
        __syncthreads();
        const u32 tsg = g0 ? g0 * 16u : VIT_TAIL, tend = g1 * 16u;
        const u32 te = t_T < tend ? t_T : tend, te_max = T_max < tend ? T_max : tend;
        const u32 P_top = t_T > tend ? P_part : P8_ZERO;
        P_part = traceback_part8(dec, scratch, img, fstride, lane, tsg, te, te_max, g0, P_top);
        g1 = g0;
    }
    __syncthreads();

    // bit b of the image is decoded bit b; output bytes are MSB-first (deconvolve.cpp:432-433); 8 lanes per frame
    {
        const u32 fi = lane >> 3, m0 = lane & 7u;
        const u32 nbytes = (mine.fb + 7u) >> 3;  // a partial last byte is padded with zero bits (ChainBack starts from E = 0)
        uint8_t* o = out + mine.ooff;
        if (((mine.ooff | nbytes) & 3u) == 0) {
            for (u32 m = m0; m < (nbytes >> 2); m += 8u)
                reinterpret_cast<u32*>(o)[m] = __builtin_bswap32(__builtin_bitreverse32(img[fi * fstride + m]));
        } else {
            for (u32 j = m0; j < nbytes; j += 8u) {
                const u32 byte = (img[fi * fstride + (j >> 2)] >> (8u * (j & 3u))) & 0xFFu;
                o[j] = (uint8_t)(__builtin_bitreverse32(byte) >> 24);
            }
        }
    }
}

}  // namespace

bool vit_pk8_supported(uint32_t max_framebits) {
    if (max_framebits < 2 || (max_framebits & 1u)) return false;
    const u32 nb = (max_framebits + VIT_TAIL + 15u) >> 4;
    return nb <= P8_SEG_BLOCKS && p8_layout(max_framebits).total <= 160u * 1024u;
}

hipError_t vit_launch_pk8(const void* d_symbols, bool sym32, uint8_t* d_out, const vit_frame_desc* d_desc, uint32_t framebits,
                          uint32_t max_framebits, int64_t nframes, hipStream_t stream, bool renorm_ge) {
    const uint8_t* d_sym = static_cast<const uint8_t*>(d_symbols);
    if (sym32 && (reinterpret_cast<uintptr_t>(d_symbols) & 15u)) return hipErrorInvalidValue;  // uint4 loads
    if (nframes <= 0) return hipSuccess;
    if (!vit_pk8_supported(max_framebits)) return hipErrorInvalidValue;
    hipError_t e;
    int dev = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    {
        static uint64_t optin_done = 0;
        const void* ks[2] = {reinterpret_cast<const void*>(vit_pk8_kernel<false>), reinterpret_cast<const void*>(vit_pk8_kernel<true>)};
        if ((e = vit_optin_dynamic_lds(ks, 2, 160 * 1024, dev, &optin_done)) != hipSuccess) return e;
    }
    const long long groups = (nframes + 7) / 8;
    if (groups > 0x7FFFFFFFLL) return hipErrorInvalidValue;
    const P8Layout lay = p8_layout(max_framebits);
    const u32 rc = pk_renorm_const(renorm_ge);
    if (sym32)
        hipLaunchKernelGGL(vit_pk8_kernel<true>, dim3((unsigned)groups), dim3(64), lay.total, stream, d_sym, d_out, d_desc,
                           framebits, (long long)nframes, lay, lay.maxfb, rc);
    else
        hipLaunchKernelGGL(vit_pk8_kernel<false>, dim3((unsigned)groups), dim3(64), lay.total, stream, d_sym, d_out, d_desc,
                           framebits, (long long)nframes, lay, lay.maxfb, rc);
    return hipGetLastError();
}
