// vit_pk.hip -- packed K=7 r=1/4 Viterbi decoder for gfx950: FOUR frames per
// wavefront, no cross-lane traffic through LDS memory in the add-compare-select loop.
//
// Layout.  A wave owns frames F(p,h), p = lane>>5 ("pair"), h = 16-bit half of a
// VGPR.  Within a pair, the 64 path metrics of a frame live in two registers x 32
// lanes: register A holds the state with s5 = 0, register B the one with s5 = 1,
// of the butterfly i = rol5(lane&31, t mod 5).  A butterfly therefore reads both
// of its predecessors (i, i+32) from its own lane, and both survivors
// (2i, 2i+1) stay in the lane.  After each step ONE register bit and ONE lane
// bit swap roles (lane bit 4,3,2,1,0,4,... = 4 - t mod 5): v_permlane16_swap for
// bit 4, masked DPP moves for bits 3/2, ds_swizzle + select for bits 1/0.  The
// state <-> lane map rotates with period 5, so the step body is unrolled
// 5-periodically (16 steps x 5 variants).
//
// Arithmetic.  Metrics are u16 lanes of v_pk_* instructions holding m + 0xFF00,
// so `v_pk_add_u16 clamp` IS paddusb (saturation at 255).  The renormalisation
// (`psubusb 63` when state 0 > 150, every second step) is `v_pk_sub_u16 clamp`
// against 0xFF00 + {0,63}, which lands in a 0-based representation; the branch
// metric table of the following (even) step carries the +0xFF00 back.  Decisions
// are the sign bits of m0-m1 / m2-m3, shifted into two per-lane history registers,
// one 32-bit word pair (8 B/lane) per 16 steps.
//
// Branch metrics.  Only 8 distinct (b0,b1,b2) mask triples exist, so a 32-step
// pre-pass (lane = pair x step) computes the 8 pavgb-tree metrics of both frames of
// its pair with byte-wide v_perm_b32 / v_lerp_u8 and writes a table of M to LDS; an
// ACS lane reads its 4 bytes per step and forms 63-M with one v_sub.
//
// Decision storage.  LDS (160 KB/CU) is what limits resident waves.  Per segment of
// 49 blocks (784 steps) the first 32 blocks of history stay in VGPRs (two 32-dword
// register arrays indexed with s_set_gpr_idx), 16 go to LDS and the last one to the
// start of the then-dead metric table: 10 KB of LDS per wave, 16 waves per CU.
// Longer frames (vit_pk_long_kernel): same forward pass; the frame is traced back 256 steps at a time WHILE
// the pass runs, from a 16-block window in LDS, and every block is also written - once, 8 B per frame-step -
// to a per-workgroup slice of HBM from which only a part that fails its check is read back.
//
// Traceback.  Per segment three parts (LDS tail, then the register blocks 16 at a
// time through the same LDS region), each blocked and speculative: lane = (frame,
// block of BL steps); a block is traced from state 0 after a 30-step warm-up (or
// from the true position when that reaches the frame end), checked against the
// block above and re-traced until nothing changes.  The last block really starts
// in state 0 (tail-terminated), so the fixed point is exactly the serial chainback.
// Two forms: the general one (any mix of lengths in a wave, 20-step blocks, a loop
// with a predicate per step) and, since round 3, a fast one for waves of four equally
// long frames of a multiple of 16 bits - every DAB size -: 16-step blocks in
// straight-line code, 5 instead of 9 instructions per step back (traceback_part16).
//
// Scheduling (round 3, from a per-workgroup timeline of the launch, tools/exp/timeline.py): the hardware's arbitration lets the
// waves of a SIMD advance one after the other.  In a launch of many rounds of waves that is welcome - the latency-bound traceback
// of one wave runs under the ACS of the others - and ONE wave per SIMD at a lower static issue priority is the best of eight
// maps measured (four distinct levels starve the lowest: 63 ... 248 us per wave and a 50 us drain).  In a launch of ONE round
// nothing replaces a finished wave, so the priority rotates block by block and the four waves finish together (a separate
// instantiation of the single-segment kernel; a run-time branch in the persistent long-frame kernel, whose workgroups also take
// their first group statically instead of queueing at one atomic counter).  Round 4 (profiles/r04_ab_long_inflight.txt, sections 13-21): the
// persistent kernel's multi-round launches follow "more work left -> more issue slots" - in a uniform launch the last round of
// fetches runs at a priority that grows with the fetch order, in a length-sorted table the long groups run above the short ones -;
// an s_setprio per block is not free, so launches of at most one wave per SIMD do not rotate.
//
// Instruction costs that shaped this (profiles/r01_valu_issue_rates_ubench.txt): v_pk_*,
// VOP3 three-operand, DPP, SDWA, v_cmp = 4 cycles per wave; plain VOP2 = 2;
// v_permlane*_swap = 8; v_cndmask_b32 via VCC ~22; SALU 1 instr/cycle/CU.
//
// Replaces, from scratch: decon_avx2 / Butterfly256 (deconvolve.cpp:334-387,
// 514-526), Load8Syms256 (:219-228), Renormalize256 (:407-412), ChainBack
// (:416-435), chainback.inc:18-41 and const.asm:19-63.
#include <cstdlib>
#include <mutex>

#include "vit_internal.h"
#include "vit_pk_dev.h"

namespace {

constexpr int TAB_BYTES = 2048;  // 32 steps x 2 pairs x 8 triples x M (4 B); 63-M is one v_sub in the ACS
constexpr int DEC_BLOCK = 512;   // 16 steps of decisions: 64 lanes x 8 B
constexpr unsigned PK_SPLIT_DEN = 8;  // a sorted table is split between the kernels when < 1/8 of its frames are long

#ifndef VIT_TAB_STATIC
#define VIT_TAB_STATIC 1  /* one set of table ADDRESSES per table half (even / odd block), the half chosen by a uniform branch: no
                             address add in front of the table reads (5 per block; 0.4363 -> 0.4306 ms, profiles/r03_ab_tabs.txt) */
#endif
struct Lanes {
    u32 toff[5];  // LDS byte offset of this lane's (M,MM) entry for phase rho
};


// lane-bit <-> register-bit transpose on lane bit J: afterwards A holds the s5=0
// member and B the s5=1 member of the next butterfly.
//   A = bit_J(lane) ? N1[lane ^ 2^J] : N0        B = bit_J(lane) ? N1 : N0[lane ^ 2^J]
// J = 4: v_permlane16_swap.  J < 4: two v_cndmask_b32_dpp (DPP on src0, select by VCC);
// `s_nop 1` covers the VALU-write -> DPP-read hazard of N0/N1, which hipcc cannot see
// inside an asm statement.
// Issue cost on gfx950 (profiles/r01_valu_issue_rates_ubench.txt): every v_pk_*, VOP3 three-operand,
// DPP and v_cmp instruction holds the SIMD for 4 cycles per wave, v_permlane*_swap for 8, and
// v_cndmask_b32 through VCC for ~22.  So for J < 4 the partner values travel through the LDS
// crossbar (ds_swizzle: no VALU slot, no LDS memory) and two v_cndmask_b32_e64 pick them up.
#ifndef VIT_SWZ_ALL
#define VIT_SWZ_ALL 1  /* which of lane bits 3 (bit 0 of this mask) and 2 (bit 1) are exchanged through ds_swizzle instead of
                          masked DPP moves.  Since the round-2 traceback the kernel sits between the VALU and the LDS limit:
                          none 0.467 ms, bit 3 only 0.458, bit 2 only 0.459, both 0.462 (profiles/r02_ab_k3_swz.txt) */
#endif
#ifndef VIT_K3
#define VIT_K3 2  /* the renormalisation subtrahend in three instructions (v_add_u32, v_pk_ashrrev_i16, v_and_or_b32) instead of four
                     (add, shift, and, multiply-add): no faster in round 2, 0.8 % on the headline batch and 1-2 % on config 3 since the
                     kernel has fewer instructions elsewhere (profiles/r03_ab_swz_k3.txt) */
#endif
#ifndef VIT_STEPS6
#define VIT_STEPS6 1  /* skip the ten padding steps of the last block when T = 6 mod 16 */
#endif
#ifndef VIT_PRIO
#define VIT_PRIO 1  /* measured: ~1 % on the 65536-frame batch */
#endif
#ifndef VIT_PAIR_LSB
#define VIT_PAIR_LSB 0  /* 1: the pair index is lane bit 0 and the five state bits are lane bits 1..5 (two swap instructions) */
#endif
// ACS lane roles: which lane bits carry the butterfly index (l5) and which one the pair.
DEV u32 acs_l5(u32 lane) { return VIT_PAIR_LSB ? lane >> 1 : lane & 31u; }
DEV u32 acs_pair(u32 lane) { return VIT_PAIR_LSB ? lane & 1u : lane >> 5; }
// toff by lane, tabulated at compile time (80 VALU instructions per wave as arithmetic; two loads that are back long before the
// first table read).  Butterfly index of ACS lane l5 at phase rho = rol5(l5, rho); its class = parity((2i) & poly_j), const.asm:27-63.
struct alignas(32) ToffTable {  // load_toff reads a row with one 16-byte load
    u32 v[64][8];  // [lane][rho], rows padded to 32 bytes
};
constexpr ToffTable make_toff_table() {
    ToffTable t{};
    for (u32 lane = 0; lane < 64; lane++) {
        const u32 l5 = VIT_PAIR_LSB ? lane >> 1 : lane & 31u, pair = VIT_PAIR_LSB ? lane & 1u : lane >> 5;
        for (u32 rho = 0; rho < 5; rho++) {
            const u32 i = ((l5 << rho) | (l5 >> (5 - rho))) & 31u;
            const u32 i0 = i & 1u, i1 = (i >> 1) & 1u, i2 = (i >> 2) & 1u, i3 = (i >> 3) & 1u, i4 = (i >> 4) & 1u;
            const u32 c = (i1 ^ i2 ^ i4) | ((i0 ^ i1 ^ i2) << 1) | ((i0 ^ i3) << 2);
            t.v[lane][rho] = pair * 32u + c * 4u;
        }
    }
    return t;
}
__constant__ ToffTable g_toff = make_toff_table();
DEV void load_toff(Lanes& L, u32 lane) {
    const uint4 a = *reinterpret_cast<const uint4*>(&g_toff.v[lane][0]);
    L.toff[0] = a.x;
    L.toff[1] = a.y;
    L.toff[2] = a.z;
    L.toff[3] = a.w;
    L.toff[4] = g_toff.v[lane][4];
}
template <int J>  // J = state lane bit (bit J of l5)
DEV void exchange(u32& A, u32& B, u32 N0, u32 N1, u32 lane) {
    constexpr int P = J + VIT_PAIR_LSB;  // physical lane bit
    if constexpr (P == 5) {
        // swap N0's upper 32 lanes with N1's lower 32 lanes
        auto r = __builtin_amdgcn_permlane32_swap(N0, N1, false, false);
        A = r[0];
        B = r[1];
    } else if constexpr (P == 4) {
        // swap N0's odd rows with N1's even rows (rows = 16 lanes)
        auto r = __builtin_amdgcn_permlane16_swap(N0, N1, false, false);
        A = r[0];
        B = r[1];
    } else if constexpr (P == 3 && !(VIT_SWZ_ALL & 1)) {
        // masked DPP moves stay in the VALU (10 cycles incl. one copy) and keep LDS latency off this step
        A = __builtin_amdgcn_update_dpp(N0, N1, 0x128 /*row_ror:8*/, 0xF, 0xC, false);
        B = __builtin_amdgcn_update_dpp(N1, N0, 0x128, 0xF, 0x3, false);
    } else if constexpr (P == 2 && !(VIT_SWZ_ALL & 2)) {
        A = __builtin_amdgcn_update_dpp(N0, N1, 0x114 /*row_shr:4*/, 0xF, 0xA, false);
        B = __builtin_amdgcn_update_dpp(N1, N0, 0x104 /*row_shl:4*/, 0xF, 0x5, false);
    } else {
        // lane bits 0/1 cannot be masked by DPP bank masks (and DPP needs the source lane active)
        const bool hi = (lane >> P) & 1u;
        // partner values through ds_swizzle (LDS crossbar, no VALU slot): 8 VALU cycles for the selects
        constexpr int pat = 0x1F | ((1 << P) << 10);  // BitMode: src lane = lane ^ 2^P within 32
        const u32 p1 = (u32)__builtin_amdgcn_ds_swizzle((int)N1, pat);
        const u32 p0 = (u32)__builtin_amdgcn_ds_swizzle((int)N0, pat);
        A = hi ? p1 : N0;
        B = hi ? N1 : p0;
    }
}

// One trellis step for 4 frames (deconvolve.cpp:352-374 in packed u16 form).
template <int RHO, int J, bool HIST>
DEV void acs_step(u32& A, u32& B, u32& acc0, u32& acc1, u32 mt, u32 lane, const Consts& C) {
    constexpr bool ODD = (J & 1) != 0;
    // 63 - M per half as ONE 32-bit subtract.  Odd steps: M <= 63, no borrow.  Even steps carry the
    // +0xFF00 bias: (0xFE3F - M') mod 2^16 per half; the low half always borrows, hence 0xFE40 on top.
    const us2 a = U(A), b = U(B), M = U(mt), MM = U((ODD ? 0x003F003Fu : 0xFE40FE3Fu) - mt);
    const us2 m0 = __builtin_elementwise_add_sat(a, M), m1 = __builtin_elementwise_add_sat(b, MM);
    const us2 m2 = __builtin_elementwise_add_sat(a, MM), m3 = __builtin_elementwise_add_sat(b, M);
    us2 n0 = __builtin_elementwise_min(m0, m1), n1 = __builtin_elementwise_min(m2, m3);
    // sign(m0-m1) = 1  <=>  m0 < m1  <=>  decision bit 0 (tie -> decision 1)
    if constexpr (HIST) {
        const us2 x01 = m0 - m1, x23 = m2 - m3;
        // history: |m0 - m1| <= 255, so bits 9..15 of each half of the difference are seven copies
        // of its sign, and one v_bfi can drop the decision at any of those positions.  Step j of
        // the block ends up at bit j of its half with only TWO 32-bit shifts per 16 steps:
        //   steps 0,1 -> bits 14,15 | >>7 | steps 2..8 -> bits 9..15 | >>7 | steps 9..15 -> bits 9..15
        // (what a shift carries from the upper half into bits 9..15 is overwritten by the seven
        // inserts that follow it; the stale bits of the previous block are shifted out).
        constexpr int pos = J < 2 ? 14 + J : J < 9 ? 7 + J : J;
        constexpr u32 mask = 0x00010001u << pos;
        if constexpr (J == 2 || J == 9) {
            acc0 >>= 7;
            acc1 >>= 7;
        }
        acc0 = bfi(mask, W(x01), acc0);
        acc1 = bfi(mask, W(x23), acc1);
    }
    if constexpr (ODD) {
        // Renormalize256: state 0 (lane 0 of the pair, register N0) > 150 (or >= 150, the MASM decoders' test) ->
        // psubusb 63.  z = m + 0xFF00 per half.  z + 0x8069 has bit 15 set iff m >= 151; done as ONE 32-bit add:
        // the low half always carries out (0xFF00 + 0x8069 >= 2^16), so the high constant is 0x8068 (C.rc).
        // The subtrahend is per frame, i.e. the same for every lane and both registers of a pair, so it
        // commutes with the lane exchange: the broadcast and the exchange are issued together and
        // the subtraction lands on the exchanged registers (one LDS latency instead of two in a row).
#if defined(VIT_DIAG_NO_RENORM)
        exchange<4 - RHO>(A, B, W(n0), W(n1), lane);  // timing-only diagnostic builds: outputs are wrong
        return;
#elif defined(VIT_DIAG_NO_K)
        exchange<4 - RHO>(A, B, W(n0), W(n1), lane);
        A = W(__builtin_elementwise_sub_sat(U(A), U(C.hi)));
        B = W(__builtin_elementwise_sub_sat(U(B), U(C.hi)));
        return;
#endif
#if VIT_PAIR_LSB
        const u32 z = (u32)__builtin_amdgcn_ds_bpermute((int)((lane & 1u) << 2), (int)W(n0));  // state 0 of the pair: lane 0 / 1
#else
        const u32 z = (u32)__builtin_amdgcn_ds_swizzle((int)W(n0), 0);  // lane 0 of each 32-lane group
#endif
        exchange<4 - RHO>(A, B, W(n0), W(n1), lane);
#if VIT_K3 == 2
        // v_add_u32 (2.7 cycles; bit 15 of each half := m >= 151), v_pk_ashrrev_i16 15, v_and_or_b32
        u32 K;
        {
            const u32 w = z + C.rc;
            asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]\n\t"
                "v_and_or_b32 %0, %0, %2, %3"
                : "=&v"(K)
                : "v"(w), "s"(0x003F003Fu), "v"(C.hi));
        }
#elif VIT_K3
        // three instructions: v_pk_add_u16 (bit 15 of each half := m >= 151), v_pk_ashrrev_i16 15, v_and_or_b32
        u32 K;
        asm("v_pk_add_u16 %0, %1, %2\n\t"
            "v_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]\n\t"
            "v_and_or_b32 %0, %0, %3, %4"
            : "=&v"(K)
            : "v"(z), "s"(C.rc + 0x00010000u), "s"(0x003F003Fu), "v"(C.hi));
#else
        const u32 w = z + C.rc;
        const u32 t = (w >> 15) & 0x00010001u;
        const u32 K = t * 63u + C.hi;  // v_mad_u32_u24: 0xFF00 + {0,63} per half
#endif
        A = W(__builtin_elementwise_sub_sat(U(A), U(K)));  // -> 0-based representation
        B = W(__builtin_elementwise_sub_sat(U(B), U(K)));
    } else {
#ifdef VIT_DIAG_SKIP_X
        if constexpr (RHO == 2) {  // timing-only diagnostic: one exchange in five left out (outputs are wrong)
            A = W(n0);
            B = W(n1);
            return;
        }
#endif
        exchange<4 - RHO>(A, B, W(n0), W(n1), lane);
    }
}

template <int V, int J, int JEND, bool HIST>
struct Steps {
    static DEV void run(u32& A, u32& B, u32& acc0, u32& acc1, const char* tab, const Lanes& L, u32 lane,
                        const Consts& C) {
        constexpr int RHO = (V + J) % 5;
#if VIT_TAB_STATIC
        // L.toff holds LDS addresses of the table half this block reads: no address arithmetic left in the loop
        const u32 mt = *reinterpret_cast<const __attribute__((address_space(3))) u32*>(L.toff[RHO] + (u32)(J * 64));
#else
        const u32 mt = *reinterpret_cast<const u32*>(tab + L.toff[RHO] + J * 64);
#endif
        acs_step<RHO, J, HIST>(A, B, acc0, acc1, mt, lane, C);
        Steps<V, J + 1, JEND, HIST>::run(A, B, acc0, acc1, tab, L, lane, C);
    }
};
template <int V, int JEND, bool HIST>
struct Steps<V, JEND, JEND, HIST> {
    static DEV void run(u32&, u32&, u32&, u32&, const char*, const Lanes&, u32, const Consts&) {}
};
template <bool HIST, int N = 16>
DEV void steps16(u32 v, u32& A, u32& B, u32& acc0, u32& acc1, const char* th, const Lanes& L, u32 lane, const Consts& C) {
    switch (v) {
        case 0: Steps<0, 0, N, HIST>::run(A, B, acc0, acc1, th, L, lane, C); break;
        case 1: Steps<1, 0, N, HIST>::run(A, B, acc0, acc1, th, L, lane, C); break;
        case 2: Steps<2, 0, N, HIST>::run(A, B, acc0, acc1, th, L, lane, C); break;
        case 3: Steps<3, 0, N, HIST>::run(A, B, acc0, acc1, th, L, lane, C); break;
        default: Steps<4, 0, N, HIST>::run(A, B, acc0, acc1, th, L, lane, C); break;
    }
}
// Last block of a frame whose step count is 6 mod 16 - every DAB size (framebits = 96*m, and the FIC's
// 768): the ten padding steps are not computed.  After six steps the history sits at bits 7..12 of
// each half (see acs_step); one more shift puts step j at bit j like in a full block.
DEV void steps6(u32 v, u32& A, u32& B, u32& acc0, u32& acc1, const char* th, const Lanes& L, u32 lane, const Consts& C) {
    steps16<true, 6>(v, A, B, acc0, acc1, th, L, lane, C);
    acc0 >>= 7;
    acc1 >>= 7;
}

typedef u32 v32u __attribute__((ext_vector_type(32)));
constexpr u32 VREG_BLOCKS = 32;  // decision blocks that can stay in VGPRs (2 x 32 dwords)

constexpr u32 DUMP_GROUP = 16;   // register blocks are dumped to LDS 16 at a time
constexpr u32 SEG_BLOCKS = VREG_BLOCKS + DUMP_GROUP + 1u;  // 49 blocks = 784 steps: one FIC frame

// Decision history of ONE SEGMENT (<= 49 blocks).  Blocks [0,R) of the segment stay in VGPRs, the
// other nb - R in LDS: Ld of them in the `dec` region, the LAST one where the (by then dead)
// branch-metric table starts, right behind dec - so a FIC frame needs 16 x 512 + 2048 = 10 KB of
// LDS and 16 waves fit a CU.
//   nb <= 17 : R = 0,                Ld = nb - 1
//   else     : R = min(32, nb - 17), Ld = nb - R - 1  (>= 16 = one dump group)
// Frames longer than a segment take vit_pk_long_kernel below (blocks beyond the last 17 go through HBM).
__host__ __device__ inline u32 pk_reg_blocks(u32 nb) {
    if (nb <= DUMP_GROUP + 1u) return 0;
    const u32 r = nb - (DUMP_GROUP + 1u);
    return r < VREG_BLOCKS ? r : VREG_BLOCKS;
}
__host__ __device__ inline u32 pk_img_stride(u32 maxfb) { return ((maxfb + 31u) >> 5) + 2u; }  // dwords per frame
__host__ __device__ inline u32 pk_scratch_words(u32 maxfb) {
    // words per lane of traceback bit scratch: the longest part is a segment's LDS tail (<= 17 blocks)
    // or a 256-step register group -> BL <= 20 for full segments; short frames are all "tail"
    u32 nblk = (maxfb + VIT_TAIL + 15u) >> 4;
    if (nblk > SEG_BLOCKS) nblk = SEG_BLOCKS;
    const u32 tail = (nblk - pk_reg_blocks(nblk)) * 16u;
    const u32 span = tail > 256u ? tail : 256u;
    const u32 bl = 5u * ((span + 79u) / 80u);
    return (bl + 31u) >> 5;
}
struct PkLayout {
    u32 dec_bytes;  // tab starts here
    u32 img_off;    // output bit image
    u32 total;
    u32 maxfb;      // the framebits this layout was sized for
};
__host__ __device__ inline PkLayout pk_layout(u32 maxfb) {  // single-segment kernel (nblk <= 49)
    const u32 nb = (maxfb + VIT_TAIL + 15u) >> 4;
    PkLayout l;
    l.maxfb = maxfb;
    l.dec_bytes = (nb - pk_reg_blocks(nb) - 1u) * DEC_BLOCK;
    const u32 scratch = 64u * 4u * pk_scratch_words(maxfb), img = 16u * pk_img_stride(maxfb);
    u32 tabregion = DEC_BLOCK + scratch + img;  // the image lives in the dead table region too
    tabregion = tabregion > (u32)TAB_BYTES ? ((tabregion + 15u) & ~15u) : (u32)TAB_BYTES;
    l.img_off = l.dec_bytes + DEC_BLOCK + scratch;
    l.total = l.dec_bytes + tabregion;
    return l;
}

// ---- traceback -------------------------------------------------------------------------------
// The path is followed in PHYSICAL coordinates: (l, n) = lane (within the pair) and register
// (N0/N1) that held the survivor decision of the state on the path after step t.  With
// j = 4 - ((t-1) mod 5), the lane bit exchanged before step t, one step back is
//     n' = bit j of l,   l' = l with bit j replaced by the decision k read at step t
// (this is ChainBack's E = (E>>1)|(k<<7), deconvolve.cpp:424-433, seen through the rotating
// lane<->state map).  The history words hold NOT k, so the code tracks the complement
// P = (31-l)<<3 | (1-n)<<2 - the stored bit is inserted unchanged - and the decision blocks are laid
// out in LDS in those same complemented coordinates (dec_slot() below), so that P IS the byte offset
// of the history word inside its block.  State 0 is P = 252.
constexpr u32 P_ZERO = 252u;
#ifndef VIT_TB_WARM
#define VIT_TB_WARM 30
#endif
#ifndef VIT_TB16
#define VIT_TB16 (VIT_TB_WARM == 30)  /* the fast traceback form (16-step blocks, straight-line code written for a 30-step
                                         warm-up) for waves of four equally long frames */
#endif
constexpr u32 TB_WARM = VIT_TB_WARM;  // warm-up steps (multiple of 5) a speculative block starts above its own range
// Input without signal (uniform random bytes, hard decisions from a dead channel) merges late: after 30 steps back from
// state 0 half of the blocks are still off the survivor path (3 % at Eb/N0 = 3 dB), after 90 steps 12 %
// (profiles/r03_merge_depth.txt).  A wave that sees at least TB_HARD_MISSES of its 64 blocks miss in the first pass of a
// part traces the parts that follow with the longer warm-up: one long pass instead of a chain of re-trace passes
// (profiles/r03_inputs.txt: 73.8 -> 75.4 Gbit/s on uniform random bytes, 3 dB input unchanged).
#ifndef VIT_TB_WARM_HARD
#define VIT_TB_WARM_HARD 90
#endif
#ifndef VIT_TB_HARD_MISSES
#define VIT_TB_HARD_MISSES 8
#endif
constexpr u32 TB_WARM_HARD = VIT_TB_WARM_HARD, TB_HARD_MISSES = VIT_TB_HARD_MISSES;

// LDS byte offset, inside a 512-byte decision block, of the (acc1, acc0) pair of ACS lane `lane`:
// [pair][31 - l][1 - n] - the register with n = 1 first.  Every store of a block uses it.
DEV u32 dec_slot(u32 lane) { return acs_pair(lane) * 256u + (31u - acs_l5(lane)) * 8u; }

// One step back for every active lane.  JJ = 3 + j is the position of lane bit j inside P.
//   x     = (t - 16*slot0) << 5: bits 9.. select the block, bits 5..8 are t & 15
//   PC    = P | pair << 8 | half << 1: with the mirrored block layout the 16 history bits of (frame, l, n)
//           for block t >> 4 are the halfword at (x & ~511) | PC
//   kb    = bit t & 15 of it (= NOT decision);   P: bit JJ := kb, bit 2 := old bit JJ
template <int JJ>
DEV void tb_step(u32& PC, u32& kb, u32 x) {
    // x already carries the LDS address of dec (a multiple of 512, checked in traceback_part): no add left here
    const u32 w = *reinterpret_cast<const __attribute__((address_space(3))) unsigned short*>((x & ~511u) | PC);
    kb = __builtin_amdgcn_ubfe(w, (x >> 5) & 15u, 1u);
    const u32 t = ((PC >> (JJ - 2)) & 4u) | (kb << JJ);
    PC = bfi((1u << JJ) | 4u, t, PC);
}

// Runs block-relative indices i = i_from .. i_to (downwards; i_from + 1 and i_to are multiples of 5
// so that the phase of every unrolled position is static: j(i) = (j0 - i) mod 5, JA = j(i_from)).
// Lanes take part while `on` and i <= i_start.  RECORD: collect kb bits of index i into words
// (bit i&31 of word i>>5), flushed to scratch when a word is complete.  xbase = x of index 0.
template <bool RECORD, int JA>
DEV void tb_loop(u32& PC, u32* scratch, int i_from, int i_to, bool on, u32 i_start, u32 xbase) {
    u32 cur = 0;
    for (int i = i_from; i >= i_to; i -= 5) {
#define TB_ONE(K)                                                                  \
    {                                                                              \
        const int ii = i - (K);                                                    \
        u32 kb = 0;                                                                \
        if (on && (u32)ii <= i_start) tb_step<3 + (JA + (K)) % 5>(PC, kb, xbase + ((u32)ii << 5)); \
        if (RECORD) {                                                              \
            cur |= kb << (ii & 31);                                                \
            if ((ii & 31) == 0) {                                                  \
                if (on) scratch[ii >> 5] = cur;                                    \
                cur = 0;                                                           \
            }                                                                      \
        }                                                                          \
    }
        TB_ONE(0) TB_ONE(1) TB_ONE(2) TB_ONE(3) TB_ONE(4)
#undef TB_ONE
    }
}
template <bool RECORD>
DEV void tb_run(u32& PC, u32* scratch, int i_from, int i_to, bool on, u32 i_start, u32 xbase, u32 j0) {
    // i_from % 5 == 4  ->  j(i_from) = (j0 - 4) mod 5 = (j0 + 1) % 5
    switch ((j0 + 1u) % 5u) {
        case 0: tb_loop<RECORD, 0>(PC, scratch, i_from, i_to, on, i_start, xbase); break;
        case 1: tb_loop<RECORD, 1>(PC, scratch, i_from, i_to, on, i_start, xbase); break;
        case 2: tb_loop<RECORD, 2>(PC, scratch, i_from, i_to, on, i_start, xbase); break;
        case 3: tb_loop<RECORD, 3>(PC, scratch, i_from, i_to, on, i_start, xbase); break;
        default: tb_loop<RECORD, 4>(PC, scratch, i_from, i_to, on, i_start, xbase); break;
    }
}

// One traceback part over steps [ts, te) of every frame (te per lane's frame, te_max uniform),
// decisions of block b at dec + (b - slot0)*512.  Lane = (frame fi = lane>>4, block q = lane&15)
// takes the BL steps from ts + q*BL.  A block whose warm-up reaches the frame's last step of the
// part starts from the true position P_top; the others start TB_WARM steps early from state 0,
// which merges with the true path with high probability.  Afterwards every speculative block is
// checked against the block above it and re-traced until nothing changes, so the result is
// exactly the serial chainback.  ORs the decoded bits into img; returns P after step ts.
#ifndef VIT_TB_INLINE
#define VIT_TB_INLINE 1
#endif
#if VIT_TB_INLINE
DEV
#else
__device__ __attribute__((noinline))
#endif
u32 traceback_part(const char* dec, u32* scratch, u32* img, u32 fstride, u32 lane, u32 ts, u32 te, u32 te_max,
                       u32 slot0, u32 P_top, u32& warm, u32 dmask = 0xFFFFFFFFu) {
#ifdef VIT_DIAG_NO_TB
    return P_top;  // timing-only diagnostic build: outputs are wrong
#endif
    const u32 fi = lane >> 4, q = lane & 15u;
    const u32 span = te_max > ts ? te_max - ts : 0u;
    if (span == 0) return P_top;
    const u32 BL = 5u * ((span + 79u) / 80u);  // 16*BL >= span, multiple of the phase period 5
    const u32 tbase = ts + q * BL;
    const bool has_work = tbase < te;
    const u32 i_last = has_work ? te - 1u - tbase : 0u;  // block-relative index of the frame's last step here
    const u32 q_top = te > ts ? (te - 1u - ts) / BL : 0u;
    const u32 i_warm = BL - 1u + warm;
    const u32 i_start = i_last < i_warm ? i_last : i_warm;
    const bool fixed = has_work && i_last <= i_warm;  // starts from the true position: never re-traced
    // PC = P | C: the lane's constant address bits (pair, half) ride along in the tracked position
    const u32 C = (fi >> 1) * 256u + (fi & 1u) * 2u;
    // x of block-relative index 0 (every t traced is >= 16*slot0), with the LDS address of `dec` folded in: the
    // block select is (x & ~511), so that address must be a multiple of 512 (it is 0: dynamic LDS, no static LDS)
    const u32 dbase = (u32)(uintptr_t)(const __attribute__((address_space(3))) char*)dec;
    if (dbase & 511u) __builtin_trap();
    const u32 xbase = ((tbase - slot0 * 16u) << 5) + dbase;
    const u32 j0 = (4u + 5u - ((ts + 5u - 1u) % 5u)) % 5u;  // j of block-relative index 0: 4 - ((ts-1) mod 5)
    const u32 PC_top = P_top | C;

    u32 P = fixed ? PC_top : (P_ZERO | C), P_out = PC_top;
    // pass 0: warm-up (no bits kept) then the block itself
    tb_run<false>(P, scratch, (int)i_warm, (int)BL, has_work, i_start, xbase, j0);
    u32 P_in = P;  // position the trace passed through at the block's top (meaningless for short top blocks)
    tb_run<true>(P, scratch, (int)BL - 1, 0, has_work, i_start, xbase, j0);
    if (has_work) P_out = P;
    for (int pass = 0; pass < 17; pass++) {
        const u32 nxt = __shfl_down(P_out, 1);  // the block above belongs to the same frame: same C
        const u32 new_in = (q < q_top) ? nxt : PC_top;
        const bool changed = has_work && !fixed && new_in != P_in;
        const unsigned long long miss = __ballot(changed);
        if (miss == 0) break;
        if (pass == 0 && warm == TB_WARM && (u32)__popcll(miss) >= TB_HARD_MISSES) warm = TB_WARM_HARD;  // for the parts that follow
        if (changed) P_in = new_in;
        P = new_in;
        tb_run<true>(P, scratch, (int)BL - 1, 0, changed, BL - 1u, xbase, j0);
        if (changed) P_out = P;
    }
    // decoded bit index of step t is t - 6 (chainback skips the 6 tail decisions); decoded bit = NOT stored bit
    if (has_work) {
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
                atomicOr(&img[fi * fstride + (d & dmask)], val << sft);  // dmask: the long-frame kernel's image is a ring
                if (sft) atomicOr(&img[fi * fstride + ((d + 1u) & dmask)], val >> (32u - sft));
            }
        }
    }
    return __shfl(P_out, (int)(fi * 16u)) & 0xFCu;  // block 0 of the frame ends at step ts; C stripped
}

// ---- traceback, fast form: blocks of 16 steps, straight-line code (round 3) ---------------------------------------------
// For a wave whose four frames have the SAME length, a multiple of 16 bits (every DAB size: the FIC's 768, 96*m), the parts
// are cut top-down, 256 steps each (the lowest one may be shorter, a multiple of 16): part = [lo, lo + 16*nl), lo = 6 mod
// 16, lane (frame, q) takes the 16 steps from lo + 16q.  Then
//   * every lane's block-relative index ii has the same bit position (6 + ii) & 15 in its history word and crosses a
//     16-step history block at the same ii: the bit position is an immediate of v_bfe, the block an immediate offset of
//     the ds_read, and there is no address arithmetic left but ONE v_or;
//   * which lanes take part changes at two fixed points only (the two top lanes of a part start late, from the true
//     position): three straight-line segments under one exec mask each instead of a compare + exec save/restore per step;
//   * a lane's 16 decoded bits are one halfword of the image: a plain ds_write_b16, no scratch words, no atomics.
// What is NOT uniform any more is the phase of the lane <-> state map (16 is not a multiple of its period 5): the shift,
// the bit number and the v_bfi mask of a step come from five per-lane register triples indexed by ii mod 5 (static).
// A step back is 5 VALU instructions and 18.9 issue cycles (v_bfe, v_lshrrev, v_and, v_lshl_or, v_bfi; the tracked value is
// the LDS address itself) instead of 9 and 37, and a part is 16 + 30 steps instead of 20 + 30.  Same fixed point, same re-trace rule: exactly ChainBack.
struct Tb16 {
    u32 sh[5], jj[5], mk[5];  // by ii mod 5: JJ - 2, JJ, (1 << JJ) | 4 with JJ = 7 - ((t - 1) mod 5), t = tbase + ii
};
DEV u32 bfi_v(u32 mask, u32 a, u32 b) {
    u32 d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "v"(a), "v"(b));
    return d;
}
// PC carries the lane's block base (a multiple of 512, bits 9 and up, which the v_bfi never touches): it IS the LDS address.
template <int II, bool REC>
DEV void tb16_step(u32& PC, u32& cur, const Tb16& L) {
    constexpr int r = II % 5, blk = (6 + II) >> 4, idx = (6 + II) & 15;
    const u32 w = *reinterpret_cast<const __attribute__((address_space(3))) unsigned short*>(PC + (u32)(blk * DEC_BLOCK));
    // asm: left to itself hipcc narrows the extract of a zero-extended halfword to v_lshrrev_b16 + v_and 1 + v_and 0xffff
    u32 kb;
    asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(kb) : "v"(w), "n"(idx));
    const u32 m = (PC >> L.sh[r]) & 4u;
    const u32 t = (kb << L.jj[r]) | m;
    PC = bfi_v(L.mk[r], t, PC);
    if constexpr (REC) cur |= kb << II;
}
template <int IFROM, int ITO, bool REC>
struct Tb16Run {
    static DEV void run(u32& PC, u32& cur, const Tb16& L) {
        tb16_step<IFROM, REC>(PC, cur, L);
        if constexpr (IFROM > ITO) Tb16Run<IFROM - 1, ITO, REC>::run(PC, cur, L);
    }
};
// One part [lo, lo + 16*nl) of four equally long frames; decisions of block b at dec + (b - slot0)*512.  Returns P after step lo.
// SPEC (round 4, the long-frame kernel's in-flight parts): the position at the part's top is not known yet.  Every lane - the
// two top ones too - then starts 30 steps above its block from state 0 (the three history blocks above the part must be in LDS),
// the top lane is trusted, and *p_spec receives the position it passed through at the part's top: the part's output is final iff
// that equals what the part above ends in, which the caller checks once the frame's last part has been traced from the true end
// state.  *misses = speculative blocks of the wave that missed in the first pass.  TOMEM (gout = per lane: its frame's output):
// the lane's 16 decoded bits go straight to memory as two MSB-first bytes (deconvolve.cpp:432-433) instead of into the LDS image.
template <bool SPEC = false, bool TOMEM = false>
DEV u32 traceback_part16(const char* dec, u32* img, u32 fstride, u32 lane, u32 lo, u32 nl, u32 slot0, u32 P_top, u32 dmask,
                         uint8_t* gout = nullptr, u32* p_spec = nullptr, u32* misses = nullptr) {
#ifdef VIT_DIAG_NO_TB
    return P_top;
#endif
    constexpr int W = 30;  // warm-up; with 16-step blocks the chain of a speculative lane starts at ii = 45
    const u32 fi = lane >> 4, q = lane & 15u;
    const u32 tbase = lo + q * 16u;
    const bool has_work = q < nl;
    const u32 above = SPEC ? 3u : nl - q;  // blocks from this one up to the top of the part (has_work: >= 1)
    const bool fixed = has_work && above <= 2u;  // its warm-up would cross the part's top: it starts there, from the true position
    const u32 C = (fi >> 1) * 256u + (fi & 1u) * 2u;
    const u32 dbase = (u32)(uintptr_t)(const __attribute__((address_space(3))) char*)dec;
    if (dbase & 511u) __builtin_trap();
    const u32 bbq = dbase + ((lo >> 4) - slot0 + q) * DEC_BLOCK;
    Tb16 L;
    {
        const u32 c = (tbase + 4u) % 5u;  // (tbase - 1) mod 5
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const u32 e = (c + r) % 5u, JJ = 7u - e;
            L.sh[r] = JJ - 2u;
            L.jj[r] = JJ;
            L.mk[r] = (1u << JJ) | 4u;
        }
    }
    // positions travel between lanes without the block base (P & 0x1FF); a lane adds its own
    const u32 PC_top = P_top | C;
    u32 P = (fixed ? PC_top : (P_ZERO | C)) | bbq, P_out = PC_top, cur = 0;
    if (has_work && above >= 3u) Tb16Run<15 + W, 32, false>::run(P, cur, L);
    if (has_work && above >= 2u) Tb16Run<31, 16, false>::run(P, cur, L);
    u32 P_in = P & 0x1FFu;
    if (has_work) {
        Tb16Run<15, 0, true>::run(P, cur, L);
        P_out = P & 0x1FFu;
    }
    for (int pass = 0; pass < 17; pass++) {
        // the block above = the next lane of the same 16-lane row: one DPP move (row_shl:1), no LDS round trip in the pass loop
        const u32 nxt = (u32)__builtin_amdgcn_update_dpp(0, (int)P_out, 0x101, 0xF, 0xF, true);
        const u32 new_in = (q + 1u < nl) ? nxt : (SPEC ? P_in : PC_top);
        const bool changed = has_work && !fixed && new_in != P_in;
        if constexpr (SPEC) {
            const unsigned long long miss = __ballot(changed);
            if (pass == 0) *misses = (u32)__popcll(miss);
            if (miss == 0) break;
        } else {
            if (!__any(changed)) break;
        }
        if (changed) {
            P_in = new_in;
            P = new_in | bbq;
            cur = 0;
            Tb16Run<15, 0, true>::run(P, cur, L);
            P_out = P & 0x1FFu;
        }
    }
    // decoded bit index of step t is t - 6 (a multiple of 16 here); decoded bit = NOT stored bit
    if (has_work) {
        const u32 h = (tbase - VIT_TAIL) >> 4;  // halfword index in the frame's bit image
        if constexpr (TOMEM) {
            const u32 v = __builtin_bitreverse32(~cur & 0xFFFFu);  // byte 3 = bits 0..7 reversed, byte 2 = bits 8..15 reversed
            gout[2u * h] = (uint8_t)(v >> 24);
            gout[2u * h + 1u] = (uint8_t)(v >> 16);
        } else {
            reinterpret_cast<unsigned short*>(img + fi * fstride)[((h >> 1) & dmask) * 2u + (h & 1u)] = (unsigned short)~cur;
        }
    }
    if constexpr (SPEC) *p_spec = __shfl(P_in, (int)(fi * 16u + nl - 1u)) & 0xFCu;  // the top lane's position at the part's top
    return __shfl(P_out, (int)(fi * 16u)) & 0xFCu;
}

// Single-segment kernel: every frame of the launch fits 49 blocks (framebits <= 778; the FIC fast path).
// One workgroup (= one wave) per group of 4 frames, dispatched by the hardware.  A persistent form of this kernel
// (workgroups looping over groups, static stride or atomic counter, with and without a start-up stagger) was
// measured 7-12 % SLOWER on the benchmark batch and is not kept: profiles/r02_ab_persist.txt, r02_ab_stagger.txt,
// r02_ab_noprio.txt.  Even the loop scaffolding alone (grid = groups, one trip) cost 12 % through the register
// allocation it led to (128 VGPRs / 104 SGPRs instead of 113 / 63): profiles/r02_ab_r1kernel_vs_loop.txt.
#ifdef VIT_DIAG_TIMES
__device__ unsigned long long g_diag_times[16384 * 4];
#endif
// ROT: a launch of ONE round of waves (no more workgroups than the device holds at once): nothing takes the place of a wave that
// is done, so the static per-slot priorities - under which a wave takes between 63 and 248 us - would leave the second half of the
// launch to ever fewer waves; the priority rotates block by block instead and the four waves of a SIMD finish together
// (16384 FIC frames: 0.130 -> 0.117 ms).  A separate instantiation: the rotation's test inside the block loop cost launches that
// do not use it up to 6 % (profiles/r03_ab_long_oneround.txt).
template <bool SYM32, bool ROT>
__global__ __launch_bounds__(64, 4) void vit_pk_kernel(const uint8_t* __restrict__ sym, uint8_t* __restrict__ out,
                                                        const vit_frame_desc* __restrict__ desc, u32 framebits_uniform,
                                                        long long nframes, PkLayout lay, u32 vmax,
                                                        const unsigned* __restrict__ split_gate, u32 renorm_c) {
    // second launch behind the long-frame kernel on a length-sorted table: runs only if that kernel left the short
    // groups to it (same test there, see vit_launch_pk)
    if (split_gate && (unsigned long long)*split_gate * PK_SPLIT_DEN >= (unsigned long long)nframes) return;
#ifdef VIT_DIAG_TIMES
    const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();  // s_memtime
#endif
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* dec = lds;                  // [block - R][lane] -> (acc0, acc1); the last block spills into tab
    char* tab = lds + lay.dec_bytes;  // [tau][pair][c] -> M; after the ACS: last block, scratch, image
    u32* img = reinterpret_cast<u32*>(lds + lay.img_off);  // output bit image, 4 frames
    const u32 lane = threadIdx.x;
    const long long f0 = (long long)blockIdx.x * 4;
    u32 prio_slot = 0;
#if VIT_PRIO
    // Stagger the waves that share a SIMD: different issue priorities make them drift apart, so the
    // latency-bound traceback of one overlaps the ACS of the others instead of all four hitting it together.
    {
        u32 hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        prio_slot = hwid & 3u;
#ifdef VIT_PRIO_FIRST
        if (blockIdx.x < VIT_PRIO_FIRST)  // experiment: priorities only for the waves of the first round
#endif
#ifndef VIT_PRIO_MAP
#define VIT_PRIO_MAP(slot) ((slot) == 0u ? 0u : 1u)  /* ONE wave of the four at low priority, three equal: four distinct levels starve the
                                                         lowest (a wave then takes 63 ... 248 us) and the launch drains for ~50 us; this map is
                                                         1.3 % faster at 3 dB and 4 % on input without signal (profiles/r03_ab_priomap.txt) */
#endif
        switch (VIT_PRIO_MAP(hwid & 3u)) {  // wave slot within the SIMD
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
#ifdef VIT_STAGGER
        // experiment: a start-up offset per wave slot (and SIMD) in units of 64 cycles
        {
            const u32 k = (hwid & 3u) * 4u + ((hwid >> 4) & 3u);
            for (u32 i = 0; i < k; i++) __builtin_amdgcn_s_sleep(VIT_STAGGER);
        }
#endif
    }
#endif

    // ---- per-frame parameters (wave-uniform loads) ----
    u32 fbits[4];
    size_t soff[4], ooff[4];
    u32 maxfb = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long f = f0 + k;
        fbits[k] = 0;
        soff[k] = ooff[k] = 0;
        if (f < nframes) {
            if (desc) {
                fbits[k] = desc[f].framebits;
                soff[k] = desc[f].sym_offset;
                ooff[k] = desc[f].out_offset;
                // a descriptor the launch was not sized for (longer than max_framebits, or odd) is skipped rather
                // than allowed to run off the LDS layout; so is one whose symbols are not dword aligned.
                // (vmax = the launch's max_framebits; it exceeds lay.maxfb when this kernel only takes the short
                // groups of a length-sorted mixed table, see vit_launch_pk)
                if (fbits[k] > vmax || (fbits[k] & 1u) || (soff[k] & 3u)) fbits[k] = 0;
            } else {
                fbits[k] = framebits_uniform;
                soff[k] = (size_t)f * 4u * (framebits_uniform + VIT_TAIL);
                ooff[k] = (size_t)f * ((framebits_uniform + 7u) >> 3);
            }
        }
        maxfb = fbits[k] > maxfb ? fbits[k] : maxfb;
    }
    if (maxfb == 0 || maxfb > lay.maxfb) return;  // nothing valid / a group of the long-frame kernel
    const u32 nb = (maxfb + VIT_TAIL + 15u) >> 4, R = pk_reg_blocks(nb);
    const u32 fstride = pk_img_stride(maxfb);  // image dwords per frame (+ slack for the shifted spill)
    const u32 T_max = maxfb + VIT_TAIL;

    // ---- ACS lane constants ----
    const u32 l5 = acs_l5(lane);
    const u32 dslot = dec_slot(lane);  // where this lane's history words go inside a decision block
    Lanes L;
    load_toff(L, lane);
#if VIT_TAB_STATIC
    Lanes L1;
    {
        const u32 tb = (u32)(uintptr_t)(const __attribute__((address_space(3))) char*)tab;
#pragma unroll
        for (int rho = 0; rho < 5; rho++) {
            L.toff[rho] += tb;
            L1.toff[rho] = L.toff[rho] + 1024u;
            asm volatile("" : "+v"(L1.toff[rho]));  // its own register: as "L + 1024" the ds_read2 pairs would need an add again
        }
    }
#endif
    Consts C;
    C.hi = HI;
    C.rc = renorm_c;
    asm volatile("" : "+v"(C.hi));  // keep it in a VGPR
    // ---- pre-pass lane constants: lane = (tau = lane>>1, pair pp = lane&1) ----
    const u32 tau = lane >> 1, pp = lane & 1u;
    const u32 a_fb = pp ? fbits[2] : fbits[0], b_fb = pp ? fbits[3] : fbits[1];
    const u32 a_T = a_fb ? a_fb + VIT_TAIL : 0u, b_T = b_fb ? b_fb + VIT_TAIL : 0u;
    constexpr size_t SB = SYM32 ? 4 : 1;  // bytes per soft symbol in memory
    const uint8_t* a_sym = sym + SB * (pp ? soff[2] : soff[0]);
    const uint8_t* b_sym = sym + SB * (pp ? soff[3] : soff[1]);
    u32 sel[4];
    {
        const u32 hb = (tau & 1u) ? 0x0C000C00u : 0x0D000D00u;  // even step: 0xFF high bytes (= +0xFF00)
#pragma unroll
        for (int k = 0; k < 4; k++) sel[k] = hb | (0x00040000u + 0x00010001u * k);  // byte k of half 0 / half 1
    }
    const PrepassLane PL = prepass_lane(lane);
    u32 A = l5 == 0 ? 0u : 0x003F003Fu, B = 0x003F003Fu;  // const.asm:19-25 (0-based, step 0 is even)
    u32 acc0 = 0, acc1 = 0;
    v32u r0, r1;  // register-resident decisions of blocks [0,R)

    // ---- ACS over the blocks ----
    {
        auto sa = load_step<SYM32>(a_sym, tau, tau < a_T), sb = load_step<SYM32>(b_sym, tau, tau < b_T);
        u32 v = 0;
        // Two blocks per trip - the pre-pass's 32 steps: the even block reads table half 0, the odd one half 1.  (One block per trip cost
        // four v_mov of loop-carried values per block; per pair of blocks it is the same four.)
        const bool last6 = VIT_STEPS6 && (T_max & 15u) == 6u;  // the frame's last block has six steps
        auto put = [&](const u32 rbx) {  // the block's history words: registers for blocks < R, else LDS (the last one on the dead table)
            u32 rbs = rbx;
            asm volatile("" : "+s"(rbs));  // the address is formed from the scalar block index here: as an induction variable it cost a VALU add in every block
            if (rbx < R) {
                r0[rbx] = acc0;  // s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off
                r1[rbx] = acc1;
            } else {
                if (rbx + 1u == nb) __syncthreads();  // the last block lands on the table: all reads done first
                *reinterpret_cast<uint2*>(dec + (rbs - R) * DEC_BLOCK + dslot) = make_uint2(acc1, acc0);
            }
        };
        for (u32 rb = 0; rb < nb; rb += 2u) {
            u32 rbs = rb;
            asm volatile("" : "+s"(rbs));
            auto rotate = [&](const u32 rbx) {  // one-round launches: the issue priority rotates block by block
                if constexpr (ROT) {
                    switch ((prio_slot + rbx) & 3u) {
                        case 0: __builtin_amdgcn_s_setprio(0); break;
                        case 1: __builtin_amdgcn_s_setprio(1); break;
                        case 2: __builtin_amdgcn_s_setprio(2); break;
                        default: __builtin_amdgcn_s_setprio(3); break;
                    }
                }
            };
            rotate(rb);
            __syncthreads();  // every lane is done with the previous table
            prepass(pack_step(sa), pack_step(sb), tab, PL, sel);
            const u32 tn = (rbs + 2u) * 16u + tau;
            sa = load_step<SYM32>(a_sym, tn, tn < a_T);  // prefetch the next 32 steps' symbols
            sb = load_step<SYM32>(b_sym, tn, tn < b_T);
            __syncthreads();
            if (last6 && rb + 1u == nb) steps6(v, A, B, acc0, acc1, tab, L, lane, C);
            else steps16<true>(v, A, B, acc0, acc1, tab, L, lane, C);
            put(rb);
            v = v == 4 ? 0 : v + 1;
            if (rb + 1u < nb) {
                rotate(rb + 1u);
                if (last6 && rb + 2u == nb) steps6(v, A, B, acc0, acc1, tab, L1, lane, C);
                else steps16<true>(v, A, B, acc0, acc1, tab, L1, lane, C);
                put(rb + 1u);
                v = v == 4 ? 0 : v + 1;
            }
        }
    }
    __syncthreads();
#pragma nounroll
    for (u32 i = lane; i < 4u * fstride; i += 64u) img[i] = 0;  // the image aliases the dead table region (two trips for FIC frames)

    // ---- traceback, last part first: lane = (frame fi, block q) ----
#ifdef VIT_DIAG_TIMES
    const unsigned long long diag_t1 = __builtin_amdgcn_s_memrealtime();
#endif
    const u32 fi = lane >> 4;
    const u32 t_fb = fi == 0 ? fbits[0] : fi == 1 ? fbits[1] : fi == 2 ? fbits[2] : fbits[3];
    const u32 t_T = t_fb ? t_fb + VIT_TAIL : 0u;  // steps of this lane's frame
    u32* scratch = reinterpret_cast<u32*>(tab + DEC_BLOCK) + lane * pk_scratch_words(maxfb);  // traceback bit words
#if VIT_TB16
    if (fbits[0] == fbits[1] && fbits[1] == fbits[2] && fbits[2] == fbits[3] && (maxfb & 15u) == 0) {
        // ---- fast form (four equally long frames, a multiple of 16 bits): 256-step parts from the top, a window of 17 blocks ----
        u32 hi = T_max, slot0 = R, P16 = P_ZERO;
        for (;;) {
            const u32 lo = hi > 256u + VIT_TAIL ? hi - 256u : VIT_TAIL;
            P16 = traceback_part16(dec, img, fstride, lane, lo, (hi - lo) >> 4, slot0, P16, 0xFFFFFFFFu);
            if (lo == VIT_TAIL) break;
            // the window moves down by d blocks: what stays goes d slots up (every lane moves its own 8 bytes of a block), the
            // d register blocks below come in
            const u32 d = slot0 < DUMP_GROUP ? slot0 : DUMP_GROUP;
            __syncthreads();
            for (u32 sl = DUMP_GROUP; sl >= d; sl--)  // sl = 16 .. d (d >= 1: there are blocks below this part)
                *reinterpret_cast<uint2*>(dec + sl * DEC_BLOCK + dslot) = *reinterpret_cast<const uint2*>(dec + (sl - d) * DEC_BLOCK + dslot);
            const u32 g0 = slot0 - d;
#pragma unroll
            for (u32 b = 0; b < VREG_BLOCKS; b++)
                if (b >= g0 && b < slot0)
                    *reinterpret_cast<uint2*>(dec + (b - g0) * DEC_BLOCK + dslot) = make_uint2(r1[b], r0[b]);
            __syncthreads();
            slot0 = g0;
            hi = lo;
        }
    } else {
#endif
    // LDS-resident blocks [R, nb)
    const u32 t_lo = R * 16u;
    u32 warm = TB_WARM;
    u32 P_part = traceback_part(dec, scratch, img, fstride, lane, t_lo > VIT_TAIL ? t_lo : VIT_TAIL, t_T, T_max, R,
                                P_ZERO, warm);
    // register-resident blocks, 16 at a time from the top
    for (u32 g1 = R; g1 > 0;) {
        const u32 g0 = g1 > DUMP_GROUP ? g1 - DUMP_GROUP : 0u;  // group = blocks [g0, g1)
        __syncthreads();
#pragma unroll
        for (u32 b = 0; b < VREG_BLOCKS; b++)
            if (b >= g0 && b < g1)
                *reinterpret_cast<uint2*>(dec + (b - g0) * DEC_BLOCK + dslot) = make_uint2(r1[b], r0[b]);
        __syncthreads();
        const u32 tsg = g0 ? g0 * 16u : VIT_TAIL, tend = g1 * 16u;
        const u32 te = t_T < tend ? t_T : tend, te_max = T_max < tend ? T_max : tend;
        // a frame that reaches beyond this group continues from the position the later part ended in
        const u32 P_top = t_T > tend ? P_part : P_ZERO;
        P_part = traceback_part(dec, scratch, img, fstride, lane, tsg, te, te_max, g0, P_top, warm);
        g1 = g0;
    }
#if VIT_TB16
    }
#endif
    __syncthreads();

#ifdef VIT_DIAG_TIMES
    if (threadIdx.x == 0 && blockIdx.x < 16384u) {  // timeline diagnostic: start, start of the traceback, end, hardware slot
        u32 hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        g_diag_times[blockIdx.x * 4u + 0u] = diag_t0;
        g_diag_times[blockIdx.x * 4u + 1u] = diag_t1;
        g_diag_times[blockIdx.x * 4u + 2u] = __builtin_amdgcn_s_memrealtime();
        g_diag_times[blockIdx.x * 4u + 3u] = hwid;
    }
#endif
    // bit b of the image is decoded bit b; output bytes are MSB-first (deconvolve.cpp:432-433).  Sixteen lanes per frame,
    // every lane its dwords m = j, j + 16, ...: two trips for a FIC frame (a loop over the frames with 64 lanes each was
    // unrolled eightfold by the compiler: 376 VALU instructions per wave for 4 x 96 bytes).
    {
        const u32 k = lane >> 4, j = lane & 15u;
        const u32 fb_k = k == 0 ? fbits[0] : k == 1 ? fbits[1] : k == 2 ? fbits[2] : fbits[3];
        const size_t oo_k = k == 0 ? ooff[0] : k == 1 ? ooff[1] : k == 2 ? ooff[2] : ooff[3];
        const u32 nbytes = (fb_k + 7u) >> 3;  // a partial last byte is padded with zero bits (ChainBack starts from E = 0)
        uint8_t* o = out + oo_k;
        const u32* im = img + k * fstride;
        if ((((u32)oo_k | nbytes) & 3u) == 0) {
#pragma nounroll
            for (u32 m = j; m < (nbytes >> 2); m += 16u)
                reinterpret_cast<u32*>(o)[m] = __builtin_bswap32(__builtin_bitreverse32(im[m]));
        } else {
#pragma nounroll
            for (u32 b = j; b < nbytes; b += 16u) {
                const u32 byte = (im[b >> 2] >> (8u * (b & 3u))) & 0xFFu;
                o[b] = (uint8_t)(__builtin_bitreverse32(byte) >> 24);
            }
        }
    }
}

// ---- frames longer than one segment: traced back in flight from a window in LDS; decisions written once to HBM -------------
// (history: round 1 checkpoint + recompute, 1.7x the ACS work; rounds 2-3 one forward pass whose blocks were spilled to HBM and
// read back 16 at a time for a traceback after the forward pass: 8 B per frame-step each way.)
// Round 4: the forward pass writes every block but a frame's last 16 to the workgroup's slice of `spill` (512 B per block,
// coalesced) WITHOUT reading it back, and keeps what a traceback may need next in a window of LDS (see below):
//   * a group of four equally long frames of a multiple of 16 bits (every DAB size) is traced back IN FLIGHT.  Its frames are cut
//     into parts of 256 steps from the top (part 0 = the frame's end); as soon as the ACS is two blocks past the top of part p >= 1,
//     at the next point where the branch-metric table is dead (so that 19 blocks fit the workgroup's 10 KB of LDS), the part is
//     traced speculatively (traceback_part16<SPEC>: every lane 30 steps above its block from state 0, the top lane trusted), its
//     decoded bits go straight to `out`, and the position at its top (spec) and at its bottom (out) are recorded - lane p of two
//     registers holds part p, four frames x 8 bits.
//   * after the forward pass part 0 is traced from the true end state (state 0), and the chain is checked from the top down:
//     part p is final iff spec(p) = out(p - 1).  The first part that fails - 1.1 % of the wave-parts at Eb/N0 = 3 dB, 7 % at 2 dB
//     (profiles/r04_spec_stats.jsonl) - is reloaded from the spill (17 blocks), traced from its true top and checked again:
//     the fixed point is the serial ChainBack (deconvolve.cpp:416-435), as before.
//   * a wave whose first in-flight part shows >= TB_HARD_MISSES missed blocks (input without signal: half of all 30-step
//     speculations fail) stops tracing in flight: its parts stay marked "unchecked" and the top-down loop traces them all from
//     the spill - the behaviour of rounds 2-3.
//   * other groups (mixed lengths in a wave, lengths that are not a multiple of 16) take the general traceback form after the
//     forward pass: the last 17 blocks from the window, the others read back from the spill as before.
// Workgroups are persistent: each takes the next group of 4 frames from an atomic counter, so the spill buffer is sized by the
// resident waves, not by the batch, and a length-sorted descriptor table is consumed longest-first.
constexpr u32 LONG_LDS_BLOCKS = DUMP_GROUP + 1u;  // 17: the blocks of one 256-step part
constexpr u32 LONG_KEEP = DUMP_GROUP;             // the last 16 blocks of a frame are never spilled
constexpr u32 SPEC_BLOCKS = LONG_LDS_BLOCKS + 2u; // an in-flight part and the two blocks above it (30-step warm-up of its top lane)

constexpr u32 IMG_RING = 16;  // output bit image of the general form: a ring of 16 words (512 bits) per frame

__host__ __device__ inline PkLayout pk_layout_long(u32 maxfb) {
    PkLayout l;
    l.maxfb = maxfb;
    l.dec_bytes = DUMP_GROUP * DEC_BLOCK;
    const u32 scratch = 64u * 4u * pk_scratch_words(maxfb), img = 4u * 4u * IMG_RING;
    u32 tabregion = DEC_BLOCK + scratch + img;
    tabregion = tabregion > (u32)TAB_BYTES ? ((tabregion + 15u) & ~15u) : (u32)TAB_BYTES;
    l.img_off = l.dec_bytes + DEC_BLOCK + scratch;
    l.total = l.dec_bytes + tabregion;  // 10 KB for every length: 16 waves per CU
    return l;
}
static_assert(SPEC_BLOCKS * DEC_BLOCK <= DUMP_GROUP * DEC_BLOCK + (u32)TAB_BYTES, "an in-flight part must fit dec + the dead table");

// Where the forward pass keeps the history it may still need (besides the write-only spill): a WINDOW whose slot 0 holds block
// `base` (a signed block number: the short bottom part of a frame starts "below block 0").  Block base + d goes to LDS slot d for
// d < 16 (the dec region); d = 16 .. 19 would land on the branch-metric table, which is live, so those four blocks wait in eight
// VGPRs (the carry) and reach LDS slots 16 .. 19 only when the table is dead: for an in-flight part (slots 0 .. 18 = its 17 blocks and
// the two above), or after the forward pass (slot 16 = the frame's last block).  After an in-flight part the window moves up 16
// blocks and the carried blocks become its slots 0 .. 3.
// The carry is four register pairs, written by ONE inline-asm statement with scalar branches inside: to the compiler a single
// instruction that updates eight registers in place.  Every plainer form cost more than the whole scheme is worth
// (profiles/r04_ab_long_inflight.txt): a switch over the elements of an array or struct - the optimiser merges the four stores into one
// with a computed index and the object lives in scratch memory, i.e. a scratch load behind `s_waitcnt vmcnt(0)` (= behind the
// acknowledgement of the spill stores just issued) at every part, +14 % on a multi-round launch; four separate locals - a web of phi
// copies, eight v_mov_b64 in EVERY block, +3 %; an 8-dword vector with a dynamic index - 16 to 24 v_cndmask per block.
struct Carry {
    u32 a0, b0, a1, b1, a2, b2, a3, b3;  // (acc1, acc0) of window positions 16, 17, 18, 19
};
DEV void carry_put(Carry& c, int d, u32 x, u32 y) {  // d wave-uniform; positions below 16 leave the carry alone
    asm volatile(
        "s_cmp_lt_i32 %[d], 16\n\t"
        "s_cbranch_scc1 .Lcy_end%=\n\t"
        "s_cmp_lg_u32 %[d], 16\n\t"
        "s_cbranch_scc1 .Lcy_1%=\n\t"
        "v_mov_b32 %[a0], %[x]\n\t"
        "v_mov_b32 %[b0], %[y]\n\t"
        "s_branch .Lcy_end%=\n"
        ".Lcy_1%=:\n\t"
        "s_cmp_lg_u32 %[d], 17\n\t"
        "s_cbranch_scc1 .Lcy_2%=\n\t"
        "v_mov_b32 %[a1], %[x]\n\t"
        "v_mov_b32 %[b1], %[y]\n\t"
        "s_branch .Lcy_end%=\n"
        ".Lcy_2%=:\n\t"
        "s_cmp_lg_u32 %[d], 18\n\t"
        "s_cbranch_scc1 .Lcy_3%=\n\t"
        "v_mov_b32 %[a2], %[x]\n\t"
        "v_mov_b32 %[b2], %[y]\n\t"
        "s_branch .Lcy_end%=\n"
        ".Lcy_3%=:\n\t"
        "v_mov_b32 %[a3], %[x]\n\t"
        "v_mov_b32 %[b3], %[y]\n"
        ".Lcy_end%=:"
        : [a0] "+v"(c.a0), [b0] "+v"(c.b0), [a1] "+v"(c.a1), [b1] "+v"(c.b1), [a2] "+v"(c.a2), [b2] "+v"(c.b2), [a3] "+v"(c.a3), [b3] "+v"(c.b3)
        : [d] "s"(d), [x] "v"(x), [y] "v"(y)
        : "scc");
}
// carried blocks 0 .. n - 1 -> LDS slots first .. first + n - 1
DEV void carry_to_lds(const Carry& c, char* dec, u32 dslot, u32 first, u32 n) {
    if (n > 0u) *reinterpret_cast<uint2*>(dec + (first + 0u) * DEC_BLOCK + dslot) = make_uint2(c.a0, c.b0);
    if (n > 1u) *reinterpret_cast<uint2*>(dec + (first + 1u) * DEC_BLOCK + dslot) = make_uint2(c.a1, c.b1);
    if (n > 2u) *reinterpret_cast<uint2*>(dec + (first + 2u) * DEC_BLOCK + dslot) = make_uint2(c.a2, c.b2);
    if (n > 3u) *reinterpret_cast<uint2*>(dec + (first + 3u) * DEC_BLOCK + dslot) = make_uint2(c.a3, c.b3);
}
// four per-frame positions (uniform within each 16-lane row) as one word, frame k in byte k
DEV u32 pack_rows(u32 P) {
    return (u32)__builtin_amdgcn_readlane((int)P, 0) | ((u32)__builtin_amdgcn_readlane((int)P, 16) << 8) |
           ((u32)__builtin_amdgcn_readlane((int)P, 32) << 16) | ((u32)__builtin_amdgcn_readlane((int)P, 48) << 24);
}

#ifdef VIT_DIAG_SPEC  /* test build: what the fast groups did - [0] groups, [1] parts traced in flight, [2] groups that gave up tracing in
                        flight (input without signal), [3] parts traced after the forward pass beyond part 0, [4] of those: parts that had
                        been traced in flight and failed their check */
__device__ unsigned long long g_diag_spec[8];
#define SPEC_COUNT(i) do { if (lane == 0) atomicAdd(&g_diag_spec[i], 1ull); } while (0)
#else
#define SPEC_COUNT(i) do { } while (0)
#endif
#ifndef VIT_LONG_BASE_PRIO
#define VIT_LONG_BASE_PRIO 1
#endif
#ifndef VIT_LONG_INFLIGHT
#define VIT_LONG_INFLIGHT 1  /* 0: no in-flight parts - every part is traced after the forward pass from the spill (A/B; the rounds 2-3 traffic) */
#endif

// The workgroups of these kernels are ONE wavefront, and a wavefront's LDS instructions execute in issue order: what a
// __syncthreads() has to provide here is only that the compiler keeps LDS accesses on their side of it.  __syncthreads() itself
// is a workgroup-scope fence, i.e. `s_waitcnt vmcnt(0)` as well - in the long-frame kernel that is a wait for the acknowledgement
// of the spill and output stores just issued (1-2 us each time, 4.5 us per in-flight part: profiles/r04_ab_long_inflight.txt).
DEV void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <bool SYM32>
__global__ __launch_bounds__(64, 4) void vit_pk_long_kernel(const uint8_t* __restrict__ sym, uint8_t* __restrict__ out,
                                                             const vit_frame_desc* __restrict__ desc,
                                                             u32 framebits_uniform, long long nframes, PkLayout lay,
                                                             uint2* spill, u32 spill_blocks, unsigned* counter,
                                                             u32 ngroups, u32 short_max,
                                                             const unsigned* __restrict__ split_gate, u32 renorm_c, u32 nsimd) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* dec = lds;                  // 16 blocks; 17th ... 20th land on the table when it is dead
    char* tab = lds + lay.dec_bytes;  // [tau][pair][c] -> M; after the ACS: last block, scratch, image
    u32* img = reinterpret_cast<u32*>(lds + lay.img_off);
    const u32 lane = threadIdx.x;
    uint2* wspill = spill + (size_t)blockIdx.x * spill_blocks * 64u + lane;
    // few long frames in the table: leave the short groups to the single-segment kernel (no spill for them);
    // many: keep them - they are the small jobs that level the end of this kernel's longest-first schedule
    const bool split = split_gate && (unsigned long long)*split_gate * PK_SPLIT_DEN < (unsigned long long)nframes;

    // ---- lane constants (same roles as in vit_pk_kernel) ----
    const u32 l5 = acs_l5(lane);
    const u32 dslot = dec_slot(lane);  // where this lane's history words go inside a decision block
    Lanes L;
    load_toff(L, lane);
#if VIT_TAB_STATIC
    Lanes L1;
    {
        const u32 tb = (u32)(uintptr_t)(const __attribute__((address_space(3))) char*)tab;
#pragma unroll
        for (int rho = 0; rho < 5; rho++) {
            L.toff[rho] += tb;
            L1.toff[rho] = L.toff[rho] + 1024u;
            asm volatile("" : "+v"(L1.toff[rho]));  // its own register: as "L + 1024" the ds_read2 pairs would need an add again
        }
    }
#endif
    Consts C;
    C.hi = HI;
    C.rc = renorm_c;
    asm volatile("" : "+v"(C.hi));
    const u32 tau = lane >> 1, pp = lane & 1u;
    u32 sel[4];
    {
        const u32 hb = (tau & 1u) ? 0x0C000C00u : 0x0D000D00u;
#pragma unroll
        for (int k = 0; k < 4; k++) sel[k] = hb | (0x00040000u + 0x00010001u * k);
    }
    const PrepassLane PL = prepass_lane(lane);

#ifndef VIT_LONG_FIRST_STATIC
#define VIT_LONG_FIRST_STATIC 1
#endif
#ifndef VIT_LONG_ROT
#define VIT_LONG_ROT 2  /* 0: no issue priorities in this kernel, 2: rotating priorities in a one-round launch, 1: in the last round of every launch */
#endif
    u32 prio_slot;
    {
        u32 hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        prio_slot = hwid & 3u;  // wave slot within the SIMD
    }
#if VIT_LONG_ROT
#if VIT_LONG_BASE_PRIO
    // every wave of this kernel at the level of the single-segment kernel's three regular waves: when a split table runs both kernels
    // side by side, the few long groups are the critical path and must not rank below the other kernel's waves
    __builtin_amdgcn_s_setprio(1);
#endif
    // A launch with no more groups than workgroups is ONE round of waves: nothing takes the place of a wave that is done, and the
    // hardware's arbitration lets the four waves of a SIMD finish one after the other (profiles/r03_timeline.jsonl: at 320, 385,
    // 450 and 550 us of a 558 us launch), i.e. the SIMD runs on three, two, one wave for the second half.  An issue priority that
    // rotates block by block gives every wave the same share, and they finish together: 16384 x 3072 bits 0.561 -> 0.473 ms,
    // 16384 x 4608 bits 0.812 -> 0.700 (profiles/r03_ab_long_oneround.txt).  Doing the same for the LAST round of a longer launch
    // (VIT_LONG_ROT = 1: the groups from ngroups - gridDim.x on) measured worse than leaving multi-round launches alone.
#endif
    bool first_group = true;
    for (;;) {
        u32 grp = 0;
#if VIT_LONG_FIRST_STATIC
        // a workgroup's first group is its own index; only the later ones come from the counter (4096 workgroups that start
        // with an atomic on one address are served one at a time: the last one waited 46 us, profiles/r03_timeline.jsonl).
        // Worth 3-7 % on most multi-round shapes and on config 3; launches of an exact number of rounds lose 2-3 % (the
        // staggered start had kept their rounds apart): profiles/r03_ab_long_oneround.txt
        if (first_group) {
            grp = blockIdx.x;
        } else {
            if (lane == 0) grp = gridDim.x + atomicAdd(counter, 1u);
            grp = (u32)__builtin_amdgcn_readfirstlane((int)grp);
        }
        first_group = false;
#else
        if (lane == 0) grp = atomicAdd(counter, 1u);
        grp = (u32)__builtin_amdgcn_readfirstlane((int)grp);
#endif
        if (grp >= ngroups) break;
#ifndef VIT_LONG_TAILPRIO
#define VIT_LONG_TAILPRIO 1
#endif
#if VIT_LONG_TAILPRIO
        // The launch's LAST round of fetches (nothing replaces a wave that finishes it): the issue priority grows with the fetch order, one
        // level per quarter of the round, so the waves that start their last group late get the slots first and a SIMD's last waves finish
        // closer together.  Uniform launches of >= 2.25 rounds only: measured over 1 ... 10 rounds x four frame lengths
        // (profiles/r04_ab_long_inflight.txt, section 13) it takes 3-11 % off 2.5 ... 5 rounds (config 5's decode, 5 rounds: -3 %), is within
        // +-2 % from 5.5 rounds up, costs 7 % at exactly two rounds (all waves fetch together: the levels only unbalance them) and 2-3 % on a
        // length-sorted table, whose last fetches are its shortest frames.
        // (a sorted table whose first and last frame are equally long is a uniform launch too)
        const bool uniform_launch = !desc || desc[0].framebits == desc[nframes - 1].framebits;
        if (uniform_launch && (unsigned long long)ngroups * 4ull >= 9ull * gridDim.x && grp + gridDim.x >= ngroups) {
            switch (((grp - (ngroups - gridDim.x)) * 4u) / gridDim.x) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
            }
        }
#endif
#ifdef VIT_DIAG_TIMES
        const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long diag_tr = 0;  // time spent in in-flight parts
#endif
        const long long f0 = (long long)grp * 4;
        u32 fbits[4];
        size_t soff[4], ooff[4];
        u32 maxfb = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const long long f = f0 + k;
            fbits[k] = 0;
            soff[k] = ooff[k] = 0;
            if (f < nframes) {
                if (desc) {
                    fbits[k] = desc[f].framebits;
                    soff[k] = desc[f].sym_offset;
                    ooff[k] = desc[f].out_offset;
                    if (fbits[k] > lay.maxfb || (fbits[k] & 1u) || (soff[k] & 3u)) fbits[k] = 0;  // not what the launch was sized for / misaligned
                } else {
                    fbits[k] = framebits_uniform;
                    soff[k] = (size_t)f * 4u * (framebits_uniform + VIT_TAIL);
                    ooff[k] = (size_t)f * ((framebits_uniform + 7u) >> 3);
                }
            }
            maxfb = fbits[k] > maxfb ? fbits[k] : maxfb;
        }
        if (maxfb == 0) continue;
        // length-sorted mixed table: the groups that fit the single-segment kernel are that kernel's (it keeps two
        // thirds of the history in VGPRs instead of spilling it).  The table is ordered by the sort's 8-bit-wide bins
        // only: bin 98 holds 778-bit frames (short) next to 780/782/784-bit ones (long) in no particular order, so a
        // short group inside that bin is skipped, not taken as the end; from the first group of the bins below
        // (maxfb <= 776) on, every later group is short as well.
        if (split && maxfb <= short_max) {
            if (maxfb <= (short_max & ~7u)) break;
            continue;
        }
        const u32 nblk = (maxfb + VIT_TAIL + 15u) >> 4;
        const u32 G = nblk > LONG_KEEP ? nblk - LONG_KEEP : 0u;  // spilled blocks (<= spill_blocks)
#ifndef VIT_LONG_LENPRIO
#define VIT_LONG_LENPRIO 1
#endif
#if VIT_LONG_LENPRIO
        // A descriptor table is consumed longest first, and the launch ends when the workgroup with the longest frames does: in a multi-round
        // table the groups of more than 3/4 (1/2) of the launch's longest frame run at issue priority 3 (2), the rest at 1 - the long groups
        // finish sooner, the dynamic counter levels the short ones behind them.  Config 3 as drawn: 1.125 -> 1.058 ms at 3 dB, the same 6 % at
        // 0 dB and on input without signal; with the 1/2 threshold alone or thresholds of 7/8 and 3/4: nothing
        // (profiles/r04_ab_long_inflight.txt, section 19).  A table of equal lengths runs at one level throughout, as before.
        if (desc && ngroups > gridDim.x && desc[0].framebits != desc[nframes - 1].framebits) {
            // the launch's longest frame: the first descriptor of the (sorted) table; what the caller declared as max_framebits may be generous
            u32 ref = desc[0].framebits;
            if (ref > lay.maxfb || ref < maxfb) ref = lay.maxfb;
            if ((unsigned long long)maxfb * 4ull > (unsigned long long)ref * 3ull) __builtin_amdgcn_s_setprio(3);
            else if ((unsigned long long)maxfb * 2ull > (unsigned long long)ref) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(1);
        }
#endif
        const u32 T_max = maxfb + VIT_TAIL;
        const u32 a_fb = pp ? fbits[2] : fbits[0], b_fb = pp ? fbits[3] : fbits[1];
        const u32 a_T = a_fb ? a_fb + VIT_TAIL : 0u, b_T = b_fb ? b_fb + VIT_TAIL : 0u;
        constexpr size_t SB = SYM32 ? 4 : 1;  // bytes per soft symbol in memory
        const uint8_t* a_sym = sym + SB * (pp ? soff[2] : soff[0]);
        const uint8_t* b_sym = sym + SB * (pp ? soff[3] : soff[1]);
        const u32 fi = lane >> 4;
        const u32 t_fb = fi == 0 ? fbits[0] : fi == 1 ? fbits[1] : fi == 2 ? fbits[2] : fbits[3];
        uint8_t* o_f = out + (fi == 0 ? ooff[0] : fi == 1 ? ooff[1] : fi == 2 ? ooff[2] : ooff[3]);
#if VIT_TB16
        const bool fast = fbits[0] == fbits[1] && fbits[1] == fbits[2] && fbits[2] == fbits[3] && (maxfb & 15u) == 0;
#else
        const bool fast = false;
#endif
        // parts of a fast group: part p = steps [max(hi - 256, 6), hi), hi = T_max - 256 p; its top block is nblk - 1 - 16 p and its
        // window base nblk - 17 - 16 p (part 0 = the frame's last 17 blocks)
        const u32 NP = fast ? (maxfb + 255u) >> 8 : 0u;
        u32 p_next = (VIT_LONG_INFLIGHT && NP > 1u) ? NP - 1u : 0u;  // bottom part first; 0 = nothing (left) to trace in flight
        // the window the ACS is filling: the next in-flight part's, else the frame's last 17 blocks (all of a shorter frame)
        int base = (fast ? (int)nblk - (int)LONG_LDS_BLOCKS : (int)(nblk > LONG_LDS_BLOCKS ? nblk - LONG_LDS_BLOCKS : 0u)) - 16 * (int)p_next;
        u32 rec_spec = 0xFFFFFFFFu, rec_out = 0u;  // lane p: part p, frame k in byte k; 0xFFFFFFFF = not traced (never equals a position)

        // ---- forward pass: ACS with history over all blocks ----
        u32 A = l5 == 0 ? 0u : 0x003F003Fu, B = 0x003F003Fu;
        u32 acc0 = 0, acc1 = 0;
        Carry cy = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        {
            auto sa = load_step<SYM32>(a_sym, tau, tau < a_T), sb = load_step<SYM32>(b_sym, tau, tau < b_T);
            u32 v = 0;
            // Two blocks per trip - the pre-pass's 32 steps: the even block reads table half 0, the odd one half 1 (see vit_pk_kernel:
            // the copies of loop-carried values once per pair).  In-flight parts start behind ODD blocks only: the table is dead there.
            const bool last6 = VIT_STEPS6 && (T_max & 15u) == 6u;  // the frame's last block has six steps
            // (a launch of at most one wave per SIMD - ngroups <= nsimd - has nothing to rotate between, and the s_setprio per block costs it 4.5 %)
            const bool rot = VIT_LONG_ROT != 0 && (VIT_LONG_ROT == 3 ? true : VIT_LONG_ROT == 2 ? (ngroups <= gridDim.x && ngroups > nsimd) : grp + gridDim.x >= ngroups);
            auto rotate = [&](const u32 rbx) {
                if (rot) {
                    switch ((prio_slot + rbx) & 3u) {
                        case 0: __builtin_amdgcn_s_setprio(0); break;
                        case 1: __builtin_amdgcn_s_setprio(1); break;
                        case 2: __builtin_amdgcn_s_setprio(2); break;
                        default: __builtin_amdgcn_s_setprio(3); break;
                    }
                }
            };
            // a finished block's history words: the write-only spill, and the window (LDS slot or carry)
            auto put = [&](const u32 rbx) {
                const uint2 hw = make_uint2(acc1, acc0);  // the order of the LDS blocks: a reload is a plain copy
                if (rbx < G) wspill[(size_t)rbx * 64u] = hw;
                const int dx = (int)rbx - base;  // position in the window (< 0: a block no later part of this wave needs in LDS)
                if (dx >= 0 && dx < (int)DUMP_GROUP) *reinterpret_cast<uint2*>(dec + (u32)dx * DEC_BLOCK + dslot) = hw;
                carry_put(cy, dx, acc1, acc0);
            };
            for (u32 rb0 = 0; rb0 < nblk; rb0 += 2u) {
                rotate(rb0);
                wave_sync();
                prepass(pack_step(sa), pack_step(sb), tab, PL, sel);
                const u32 tn = (rb0 + 2u) * 16u + tau;
                sa = load_step<SYM32>(a_sym, tn, tn < a_T);
                sb = load_step<SYM32>(b_sym, tn, tn < b_T);
                wave_sync();
                if (last6 && rb0 + 1u == nblk) steps6(v, A, B, acc0, acc1, tab, L, lane, C);
                else steps16<true>(v, A, B, acc0, acc1, tab, L, lane, C);
                put(rb0);
                v = v == 4 ? 0 : v + 1;
                if (rb0 + 1u >= nblk) break;
                const u32 rb = rb0 + 1u;  // the odd block
                rotate(rb);
                if (last6 && rb + 1u == nblk) steps6(v, A, B, acc0, acc1, tab, L1, lane, C);
                else steps16<true>(v, A, B, acc0, acc1, tab, L1, lane, C);
                put(rb);
                const int d = (int)rb - base;
                // part p_next is traced in flight after the first ODD block (the table is dead then) that is >= two blocks above its top
                if (p_next && d >= (int)LONG_LDS_BLOCKS + 1) {
                    const u32 hi = T_max - 256u * p_next;
                    const u32 lo = hi > 256u + VIT_TAIL ? hi - 256u : VIT_TAIL;
                    const u32 nc = (u32)d - (DUMP_GROUP - 1u);  // carried blocks: 3 or 4
#ifndef VIT_LONG_TRACE_PRIO
#define VIT_LONG_TRACE_PRIO (-1)  /* issue priority of an in-flight part, -1 = leave it alone.  A part is a chain of ~50 dependent LDS reads with
                                     five instructions each; raising its priority (3) so that none of them queues behind the other waves' ACS
                                     measured 2-10 % SLOWER on multi-round launches, lowering it (0) no better: profiles/r04_ab_long_inflight.txt */
#endif
                    if (VIT_LONG_TRACE_PRIO >= 0) __builtin_amdgcn_s_setprio(VIT_LONG_TRACE_PRIO >= 0 ? VIT_LONG_TRACE_PRIO : 0);
#ifdef VIT_DIAG_TIMES
                    const unsigned long long diag_ta = __builtin_amdgcn_s_memrealtime();
#endif
                    wave_sync();  // every table read of this block is done
                    carry_to_lds(cy, dec, dslot, DUMP_GROUP, nc);
                    wave_sync();
                    u32 pspec = 0, misses = 0;
                    const u32 pout = traceback_part16<true, true>(dec, nullptr, 0u, lane, lo, (hi - lo) >> 4, (u32)base, 0u, 0u, o_f, &pspec, &misses);
#ifdef VIT_SPEC_SABOTAGE  /* test build: every in-flight part fails its check and comes back from the spill (outputs must not change) */
                    const u32 ks = pack_rows(pspec) ^ ((p_next & 1u) ? 0x04040404u : 0x00000800u), ko = pack_rows(pout);
#else
                    const u32 ks = pack_rows(pspec), ko = pack_rows(pout);
#endif
                    if (lane == p_next) {
                        rec_spec = ks;
                        rec_out = ko;
                    }
                    SPEC_COUNT(1);
                    if (misses >= TB_HARD_MISSES && p_next > 1u) SPEC_COUNT(2);
                    p_next--;
                    wave_sync();  // the part's blocks have been read
                    if (misses >= TB_HARD_MISSES && p_next && (int)rb < (int)nblk - (int)LONG_LDS_BLOCKS) {
                        // input without signal (half of all 30-step speculations fail): no more parts in flight, the top-down loop
                        // below takes them from the spill; the window jumps to the frame's last 17 blocks (always ahead of the ACS
                        // here: a part that is not the last but one ends >= 32 blocks below the top - the test is belt and braces)
                        p_next = 0;
                        base = (int)nblk - (int)LONG_LDS_BLOCKS;
                    } else {
                        base += (int)DUMP_GROUP;  // the next part up (part 0 included): the carried blocks are its first ones
                        carry_to_lds(cy, dec, dslot, 0u, nc);
                    }
                    if (VIT_LONG_TRACE_PRIO >= 0) __builtin_amdgcn_s_setprio(VIT_LONG_BASE_PRIO);  // (a one-round launch sets its rotating level at the next block)
#ifdef VIT_DIAG_TIMES
                    diag_tr += __builtin_amdgcn_s_memrealtime() - diag_ta;
#endif
                }
                v = v == 4 ? 0 : v + 1;
            }
        }
        // the frame's last block (window position 16) waited in the carry for the table to die
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // once per group: the spill is complete before anything is read back
        wave_sync();
        if ((int)nblk - 1 - base == (int)DUMP_GROUP) carry_to_lds(cy, dec, dslot, DUMP_GROUP, 1u);
#ifdef VIT_DIAG_TIMES
        const unsigned long long diag_t1 = __builtin_amdgcn_s_memrealtime();
#endif

#if VIT_TB16
        if (fast) {
            // ---- part 0 from the window and the true end state, then the chain of parts from the top down: a part whose recorded
            // top position is not what the part above ends in (or that was never traced) comes back from the spill ----
            u32 p = 0;
            SPEC_COUNT(0);
            uint2 d[LONG_LDS_BLOCKS];  // a part's 17 blocks on their way from the spill to LDS
#pragma unroll
            for (u32 k = 0; k < LONG_LDS_BLOCKS; k++) d[k] = make_uint2(0u, 0u);
            u32 pre = 0;               // the part whose blocks were requested ahead (0 = none)
            auto request = [&](const u32 q) {  // issue the loads of part q's blocks
                const u32 qhi = T_max - 256u * q;
                const u32 qlo = qhi > 256u + VIT_TAIL ? qhi - 256u : VIT_TAIL;
                const u32 qb = qlo >> 4, qn = (qhi - qlo) >> 4;
#pragma unroll
                for (u32 k = 0; k < LONG_LDS_BLOCKS; k++) d[k] = k <= qn ? wspill[(size_t)(qb + k) * 64u] : make_uint2(0u, 0u);
            };
            for (u32 it = 0; it <= NP; it++) {
                if (it) {
                    const u32 up = (u32)__shfl_up((int)rec_out, 1);
                    const unsigned long long bad = __ballot(lane >= 1u && lane < NP && rec_spec != up);
                    if (bad == 0) break;
                    p = (u32)__builtin_ctzll(bad);  // the topmost such part: everything above it is final
                    SPEC_COUNT(3);
                    if ((u32)__builtin_amdgcn_readlane((int)rec_spec, (int)p) != 0xFFFFFFFFu) SPEC_COUNT(4);
                }
                const u32 hi = T_max - 256u * p;
                const u32 lo = hi > 256u + VIT_TAIL ? hi - 256u : VIT_TAIL;
                const u32 nl = (hi - lo) >> 4;
                u32 slot0 = (u32)base;  // part 0 sits in the window the forward pass left behind
                u32 ktop = 0x01010101u * P_ZERO;
                if (p) {
                    slot0 = lo >> 4;
                    wave_sync();
                    ktop = (u32)__builtin_amdgcn_readlane((int)rec_out, (int)(p - 1u));
                    if (pre != p) request(p);
#pragma unroll
                    for (u32 k = 0; k < LONG_LDS_BLOCKS; k++)
                        if (k <= nl) *reinterpret_cast<uint2*>(dec + k * DEC_BLOCK + dslot) = d[k];
                }
                // (option) the part below was never traced (a wave that gave up tracing in flight): it is certain to come next, so its
                // blocks could be requested now and arrive while this part is traced, as rounds 2-3 hid the read-back latency
                pre = 0;
#ifndef VIT_LONG_PREFETCH
#define VIT_LONG_PREFETCH 0  /* measured, not adopted: config 3 on input without signal 1.37 ms without, 1.39 ms with (profiles/r04_ab_long_inflight.txt §7) */
#endif
                if (VIT_LONG_PREFETCH && p + 1u < NP && (u32)__builtin_amdgcn_readlane((int)rec_spec, (int)(p + 1u)) == 0xFFFFFFFFu) {
                    wave_sync();  // the stores above have taken their data
                    request(p + 1u);
                    pre = p + 1u;
                }
                wave_sync();
                const u32 P_top = (ktop >> (8u * fi)) & 0xFFu;
                const u32 pout = traceback_part16<false, true>(dec, nullptr, 0u, lane, lo, nl, slot0, P_top, 0u, o_f);
                const u32 ko = pack_rows(pout);
                if (lane == p) {
                    rec_spec = ktop;
                    rec_out = ko;
                }
            }
            wave_sync();  // the part's LDS blocks are read before the next group's pre-pass reuses the region
#ifdef VIT_DIAG_TIMES
            if (lane == 0 && grp < 16384u) {
                u32 hwid;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                g_diag_times[grp * 4u + 0u] = diag_t0;
                g_diag_times[grp * 4u + 1u] = diag_t1;
                g_diag_times[grp * 4u + 2u] = __builtin_amdgcn_s_memrealtime();
                g_diag_times[grp * 4u + 3u] = diag_tr;
            }
#endif
            continue;
        }
#endif
        // ---- general form: LDS tail (the last 17 blocks: the window), then the spilled blocks 16 at a time from the top ----
        const u32 Gg = nblk > LONG_LDS_BLOCKS ? nblk - LONG_LDS_BLOCKS : 0u;  // blocks below the LDS tail (= base: the window never moved)
        img[lane] = 0;  // 4 frames x IMG_RING words (the ring aliases the dead table region behind the 17th block)
        wave_sync();
        const u32 slot = lane & (IMG_RING - 1u);
        const u32 t_T = t_fb ? t_fb + VIT_TAIL : 0u;
        const u32 nbytes_f = (t_fb + 7u) >> 3;  // a partial last byte is padded with zero bits (ChainBack starts from E = 0)
        const bool o_aligned = (reinterpret_cast<uintptr_t>(o_f) & 3u) == 0;
        u32 d_hi = (t_fb + 31u) >> 5;  // image words [d_lo, d_hi) of this lane's frame are not written out yet
        // The image is a ring: after a part has been traced back, every decoded bit from its first step up is
        // final.  Lane (frame, slot) owns the one word of [d_lo, d_hi) that maps to its ring slot: it goes out as
        // 4 MSB-first bytes (deconvolve.cpp:432-433) and the slot is cleared for the words further down.
        auto flush = [&](const u32 ts_done) {
            const u32 d_lo = ts_done <= VIT_TAIL ? 0u : (ts_done - VIT_TAIL + 31u) >> 5;
            wave_sync();  // all atomicOr of the part have landed
            const u32 d = d_lo + ((slot - d_lo) & (IMG_RING - 1u));
            if (d < d_hi) {
                const u32 v = __builtin_bswap32(__builtin_bitreverse32(img[lane]));  // byte k = bit-reversed byte k
                img[lane] = 0;
                const u32 b = 4u * d;
                if (o_aligned && b + 4u <= nbytes_f) {
                    *reinterpret_cast<u32*>(o_f + b) = v;
                } else {
#pragma unroll
                    for (u32 k = 0; k < 4u; k++)
                        if (b + k < nbytes_f) o_f[b + k] = (uint8_t)(v >> (8u * k));
                }
            }
            d_hi = d_hi < d_lo ? d_hi : d_lo;
            wave_sync();
        };
        u32* scratch = reinterpret_cast<u32*>(tab + DEC_BLOCK) + lane * pk_scratch_words(lay.maxfb);
        const u32 t_lo = Gg * 16u;
        const u32 ts_top = t_lo > VIT_TAIL ? t_lo : VIT_TAIL;
        // the spilled blocks come back 16 at a time; a group is fetched into registers while the part above it
        // is being traced back, so its HBM latency is off the wave's critical path
        uint2 d[DUMP_GROUP];
#pragma unroll
        for (u32 k = 0; k < DUMP_GROUP; k++) d[k] = make_uint2(0u, 0u);  // defined on EVERY path: left undefined on the path without a fetch, the array counted as live
                                                                          // across the whole group loop - 29 more VGPRs (128 instead of 99) and 2-6 % of the 3 dB rate (r04_ab_long_inflight.txt §7)
        auto fetch = [&](const u32 g0, const u32 g1) {
#pragma unroll
            for (u32 k = 0; k < DUMP_GROUP; k++)
                d[k] = (g0 + k < g1) ? wspill[(size_t)(g0 + k) * 64u] : make_uint2(0u, 0u);
        };
        if (Gg) fetch(Gg > DUMP_GROUP ? Gg - DUMP_GROUP : 0u, Gg);
        u32 warm = TB_WARM;
        u32 P_part = traceback_part(dec, scratch, img, IMG_RING, lane, ts_top, t_T, T_max, Gg, P_ZERO, warm, IMG_RING - 1u);
        flush(ts_top);
        for (u32 g1 = Gg; g1 > 0;) {
            const u32 g0 = g1 > DUMP_GROUP ? g1 - DUMP_GROUP : 0u;
            wave_sync();
#pragma unroll
            for (u32 k = 0; k < DUMP_GROUP; k++)
                *reinterpret_cast<uint2*>(dec + k * DEC_BLOCK + dslot) = d[k];
            wave_sync();
            if (g0) fetch(g0 > DUMP_GROUP ? g0 - DUMP_GROUP : 0u, g0);  // next group down, in flight during this part
            const u32 tsg = g0 ? g0 * 16u : VIT_TAIL, tend = g1 * 16u;
            const u32 te = t_T < tend ? t_T : tend, te_max = T_max < tend ? T_max : tend;
            const u32 P_top = t_T > tend ? P_part : P_ZERO;
            P_part = traceback_part(dec, scratch, img, IMG_RING, lane, tsg, te, te_max, g0, P_top, warm, IMG_RING - 1u);
            flush(tsg);
            g1 = g0;
        }
        wave_sync();  // the image is read before the next group's pre-pass reuses the region
    }
}

constexpr u32 PK_MAX_FRAMEBITS = VIT_MAX_FRAMEBITS;
constexpr u32 PK_SHORT_MAX = SEG_BLOCKS * 16u - VIT_TAIL;  // 778: the longest frame of one segment (49 blocks)

}  // namespace

#ifdef VIT_DIAG_SPEC
extern "C" __attribute__((visibility("default"))) int vit_diag_spec(void* host_buf, int reset) {
    if (reset) {
        unsigned long long z[8] = {};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_diag_spec), z, sizeof z);
    }
    return (int)hipMemcpyFromSymbol(host_buf, HIP_SYMBOL(g_diag_spec), sizeof(unsigned long long) * 8);
}
#endif
#ifdef VIT_DIAG_TIMES
extern "C" __attribute__((visibility("default"))) int vit_diag_times(void* host_buf) {
    return (int)hipMemcpyFromSymbol(host_buf, HIP_SYMBOL(g_diag_times), sizeof(unsigned long long) * 16384 * 4);
}
#endif
bool vit_pk_supported(uint32_t max_framebits) {
    if (max_framebits < 2 || max_framebits > PK_MAX_FRAMEBITS || (max_framebits % 2u) != 0) return false;
    const u32 nblk = (max_framebits + VIT_TAIL + 15u) >> 4;
    return (nblk <= SEG_BLOCKS ? pk_layout(max_framebits) : pk_layout_long(max_framebits)).total <= 160u * 1024u;
}

// Per-thread device scratch of the launcher, grown on demand; its reuse is ordered by an event (a
// thread may alternate between streams).  Layout: [0,256) group counter of the persistent kernel,
// [256,16K) counting-sort bins, then the length-sorted descriptor copy, then the spill slices
// (grid x spill_blocks x 512 B).
struct ScratchCtx {
    void* buf = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    int dev = -1;
    // a split table runs its two kernels side by side: the long-frame kernel goes to `side`, forked from and joined
    // back into the caller's stream with these events
    bool bins_clean = false;  // the sort's histogram in this buffer is zero (the scan kernel leaves it so)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    void drop_side() {
        if (side) (void)hipStreamDestroy(side);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        side = nullptr;
        ev_fork = ev_join = nullptr;
    }
    ~ScratchCtx() {
        drop_side();
        if (buf) (void)hipFree(buf);
        if (ev) (void)hipEventDestroy(ev);
    }
};
// one per (thread, device): vit_decode_stream_multi drives every rank from one host thread, and freeing the other
// device's scratch at every step (hipFree = device-wide sync) would serialise its double-buffered pipeline
constexpr int PK_MAX_DEVS = 64;
struct ScratchCtxs {
    ScratchCtx by_dev[PK_MAX_DEVS];
};
thread_local ScratchCtxs t_scratch;
constexpr size_t SCRATCH_HDR = 16384;
constexpr int64_t SORT_MIN_FRAMES = 16;  // below this a table is not worth three extra launches

bool sort_enabled() {
    static const bool on = getenv("VITERBI_AMD_NO_SORT") == nullptr;
    return on;
}

hipError_t vit_launch_pk(const void* d_symbols, bool sym32, uint8_t* d_out, const vit_frame_desc* d_desc,
                         uint32_t framebits, uint32_t max_framebits, int64_t nframes, hipStream_t stream, bool renorm_ge) {
    const u32 rc = pk_renorm_const(renorm_ge);
    const uint8_t* d_sym = static_cast<const uint8_t*>(d_symbols);
    if (sym32 && (reinterpret_cast<uintptr_t>(d_symbols) & 15u)) return hipErrorInvalidValue;  // uint4 loads
    if (nframes <= 0) return hipSuccess;
    if (!vit_pk_supported(max_framebits)) return hipErrorInvalidValue;
    hipError_t e;
    int dev = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    {  // the dynamic-LDS opt-in is per device
        static uint64_t optin_done = 0;
        const void* ks[6] = {reinterpret_cast<const void*>(vit_pk_kernel<false, false>),
                             reinterpret_cast<const void*>(vit_pk_kernel<true, false>),
                             reinterpret_cast<const void*>(vit_pk_kernel<false, true>),
                             reinterpret_cast<const void*>(vit_pk_kernel<true, true>),
                             reinterpret_cast<const void*>(vit_pk_long_kernel<false>),
                             reinterpret_cast<const void*>(vit_pk_long_kernel<true>)};
        if ((e = vit_optin_dynamic_lds(ks, 6, 160 * 1024, dev, &optin_done)) != hipSuccess) return e;
    }
    const long long groups = (nframes + 3) / 4;
    if (groups > 0x7FFFFFFFLL) return hipErrorInvalidValue;
    const u32 nblk = (max_framebits + VIT_TAIL + 15u) >> 4;
#ifdef VIT_FORCE_LONG
    const bool is_long = true;  // experiment: every length through the persistent spill kernel
#else
    const bool is_long = nblk > SEG_BLOCKS;
#endif
    const bool sort = d_desc != nullptr && nframes >= SORT_MIN_FRAMES && sort_enabled();
    const PkLayout lay = is_long ? pk_layout_long(max_framebits) : pk_layout(max_framebits);
    long long grid = groups;
    if (is_long) {  // the long-frame kernel is persistent: as many workgroups as fit the chip
        u32 per_cu = (160u * 1024u) / lay.total;
        if (per_cu > 16u) per_cu = 16u;  // 4 waves per SIMD (launch bounds)
        grid = (long long)per_cu * vit_device_cus(dev);
        if (grid > groups) grid = groups;
    }
    const bool need_counter = is_long;
    // single-segment kernel: the instantiation with rotating priorities for a launch of one round of waves
    auto launch_short = [&](long long g, const PkLayout& l, u32 vmax, const unsigned* gate, bool may_rotate) {
        u32 per_cu = (160u * 1024u) / l.total;
        if (per_cu > 16u) per_cu = 16u;  // 4 waves per SIMD (launch bounds)
        // (a launch of at most one wave per SIMD has nothing to rotate between, and the s_setprio per block costs it time)
        const bool rot = may_rotate && g <= (long long)per_cu * vit_device_cus(dev) && g > 4ll * vit_device_cus(dev);
#define VIT_LAUNCH_SHORT(S32, R)                                                                                          \
    hipLaunchKernelGGL((vit_pk_kernel<S32, R>), dim3((unsigned)g), dim3(64), l.total, stream, d_sym, d_out, d_desc, framebits, \
                       (long long)nframes, l, vmax, gate, rc)
        if (sym32) {
            if (rot) VIT_LAUNCH_SHORT(true, true);
            else VIT_LAUNCH_SHORT(true, false);
        } else {
            if (rot) VIT_LAUNCH_SHORT(false, true);
            else VIT_LAUNCH_SHORT(false, false);
        }
#undef VIT_LAUNCH_SHORT
    };
    if (!need_counter && !sort) {
        launch_short(grid, lay, lay.maxfb, nullptr, true);
        return hipGetLastError();
    }
    const u32 spill_blocks = is_long ? nblk - LONG_KEEP : 0u;
    const size_t desc_bytes = sort ? (((size_t)nframes * sizeof(vit_frame_desc) + 255u) & ~(size_t)255u) : 0u;
    const size_t need = SCRATCH_HDR + desc_bytes + (size_t)grid * spill_blocks * DEC_BLOCK;
    if (dev < 0 || dev >= PK_MAX_DEVS) return hipErrorInvalidDevice;
    ScratchCtx& sc = t_scratch.by_dev[dev];
    if (sc.cap < need) {
        if (sc.buf) (void)hipFree(sc.buf);  // synchronises with the kernels still using it
        sc.buf = nullptr;
        sc.cap = 0;
        sc.bins_clean = false;
        if (sc.ev) (void)hipEventDestroy(sc.ev);
        sc.ev = nullptr;
        if ((e = hipMalloc(&sc.buf, need + need / 4)) != hipSuccess) return e;
        sc.cap = need + need / 4;
        sc.dev = dev;
        if ((e = hipEventCreateWithFlags(&sc.ev, hipEventDisableTiming)) != hipSuccess) return e;
    } else if ((e = hipStreamWaitEvent(stream, sc.ev, 0)) != hipSuccess) {
        return e;
    }
    char* base = static_cast<char*>(sc.buf);
    if (sort) {
        vit_frame_desc* sorted = reinterpret_cast<vit_frame_desc*>(base + SCRATCH_HDR);
        if ((e = vit_sort_descs_launch(d_desc, sorted, nframes, max_framebits, reinterpret_cast<unsigned*>(base + 256),
                                       stream, &sc.bins_clean, reinterpret_cast<unsigned*>(base))) != hipSuccess)
            return e;
        d_desc = sorted;
    }
    unsigned* counter = reinterpret_cast<unsigned*>(base);
    if (need_counter && !sort && (e = hipMemsetAsync(base, 0, 256, stream)) != hipSuccess) return e;  // (the sort's scan kernel clears it)
    if (is_long) {
        // A length-sorted table is split between the two kernels: the long-frame kernel stops at the first group
        // that fits one segment, the single-segment kernel (second launch, same stream) skips the groups before it.
        // That pays only when few frames are long (measured: config 3, 91 % long frames, lost 12 % to the split - the
        // short groups are what levels the tail of the longest-first schedule), so both kernels test the same
        // device-side count: after the sort, bin cursor [98] = number of frames of >= 777 bits.
        const u32 short_max = sort ? PK_SHORT_MAX : 0u;
        const unsigned* gate = sort ? reinterpret_cast<const unsigned*>(base + 256) + VIT_SORT_BINS + ((PK_SHORT_MAX + 7u) >> 3) : nullptr;
        uint2* spill = reinterpret_cast<uint2*>(base + SCRATCH_HDR + desc_bytes);
        // With a split in prospect the long-frame kernel runs on a side stream, forked behind the sort: a handful of
        // long groups cannot fill the chip, the single-segment kernel's workgroups take the rest of it meanwhile.
        hipStream_t ls = stream;
        if (short_max) {
            if (!sc.side) {
                if ((e = hipStreamCreateWithFlags(&sc.side, hipStreamNonBlocking)) != hipSuccess) return e;
                if ((e = hipEventCreateWithFlags(&sc.ev_fork, hipEventDisableTiming)) != hipSuccess) return e;
                if ((e = hipEventCreateWithFlags(&sc.ev_join, hipEventDisableTiming)) != hipSuccess) return e;
            }
            if ((e = hipEventRecord(sc.ev_fork, stream)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(sc.side, sc.ev_fork, 0)) != hipSuccess) return e;
            ls = sc.side;
        }
        if (sym32)
            hipLaunchKernelGGL(vit_pk_long_kernel<true>, dim3((unsigned)grid), dim3(64), lay.total, ls, d_sym, d_out,
                               d_desc, framebits, (long long)nframes, lay, spill, spill_blocks, counter, (u32)groups,
                               short_max, gate, rc, 4u * (u32)vit_device_cus(dev));
        else
            hipLaunchKernelGGL(vit_pk_long_kernel<false>, dim3((unsigned)grid), dim3(64), lay.total, ls, d_sym, d_out,
                               d_desc, framebits, (long long)nframes, lay, spill, spill_blocks, counter, (u32)groups,
                               short_max, gate, rc, 4u * (u32)vit_device_cus(dev));
        if (short_max && (e = hipGetLastError()) == hipSuccess) {
            const PkLayout lsh = pk_layout(PK_SHORT_MAX);
            launch_short(groups, lsh, max_framebits, gate, false);  // shares the chip with the long-frame kernel: static priorities
            // join: whatever the caller enqueues next waits for both kernels
            if ((e = hipGetLastError()) != hipSuccess) return e;
            if ((e = hipEventRecord(sc.ev_join, sc.side)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(stream, sc.ev_join, 0)) != hipSuccess) return e;
        }
    } else {
        launch_short(grid, lay, lay.maxfb, nullptr, true);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return hipEventRecord(sc.ev, stream);
}
