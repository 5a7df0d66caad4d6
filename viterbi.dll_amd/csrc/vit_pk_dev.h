// vit_pk_dev.h -- device helpers shared by the packed decoder kernels (vit_pk.hip: 4 frames per wavefront,
// vit_pk8.hip: 8 frames per wavefront): packed-u16 views, the pavgb-tree branch metrics (deconvolve.cpp:335-351),
// the pre-pass that tabulates them, the symbol loads (u8 device format or the reference ABI's u32).
#pragma once
#include "vit_internal.h"

namespace {

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32;

#define DEV __device__ __forceinline__

DEV us2 U(u32 x) { return __builtin_bit_cast(us2, x); }
DEV u32 W(us2 x) { return __builtin_bit_cast(u32, x); }
// v_bfi_b32: (mask & a) | (~mask & b); asm so that hipcc does not split it into and/or chains
DEV u32 bfi(u32 mask, u32 a, u32 b) {
    u32 d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(mask), "v"(a), "v"(b));
    return d;
}
DEV u32 avg4(u32 a, u32 b) { return __builtin_amdgcn_lerp(a, b, 0x01010101u); }  // pavgb x4

constexpr u32 HI = 0xFF00FF00u;  // +0xFF00 in both halves

struct Consts {
    u32 hi;  // 0xFF00FF00 in a VGPR
    u32 rc;  // renormalisation test constant (wave-uniform, SGPR), see pk_renorm_const()
};
// Renormalize256's threshold test as ONE 32-bit add on z = m + 0xFF00 per half: z + c has bit 15 set iff m exceeds the
// threshold.  The low half always carries out (0xFF00 + 0x80xx >= 2^16), so the high half's constant is one less.
//   C twins,    deconvolve.cpp:408  `> 150`  (m >= 151): c = 0x8069 -> 0x80688069
//   MASM twins, decon_avx2.asm:97,114 `cmp sil,150 ; jb` = `>= 150`: c = 0x806A -> 0x8069806A
__host__ __device__ inline u32 pk_renorm_const(bool ge) { return ge ? 0x8069806Au : 0x80688069u; }

// The 8 pavgb-tree metrics of one frame-step: s = its 4 soft symbols (bytes), ns = ~s.
// metric(c) = avg(avg(s0^B0, s1^B1), avg(s2^B2, s3^B0)) >> 2 for the mask triple
// c = b0 | b1<<1 | b2<<2 (b3 = b0); bytes of lo = c 0..3, bytes of hi = c 4..7.
// x ^ 0xFF = ~x, so one v_perm_b32 over {s, ~s} yields the four masked variants of a symbol.
// q0 / q1: the v_perm selectors that align Q with P for the first / second output word.  (0x01000100, 0x03020302) gives
// lo = classes 0..3 (b2 = 0), hi = classes 4..7; a lane that passes them SWAPPED gets the two words swapped (see prepass).
DEV void met8(u32 s, u32& lo, u32& hi, u32 q0 = 0x01000100u, u32 q1 = 0x03020302u) {
    const u32 ns = ~s;
    const u32 r0 = __builtin_amdgcn_perm(ns, s, 0x04000400u);  // s0 ^ B0, byte pos = b0 + 2*b1
    const u32 r1 = __builtin_amdgcn_perm(ns, s, 0x05050101u);  // s1 ^ B1
    const u32 r2 = __builtin_amdgcn_perm(ns, s, 0x06060202u);  // s2 ^ B2, byte pos = b0 + 2*b2
    const u32 r3 = __builtin_amdgcn_perm(ns, s, 0x07030703u);  // s3 ^ B0
    const u32 P = avg4(r0, r1), Q = avg4(r2, r3);
    const u32 qlo = __builtin_amdgcn_perm(Q, Q, q0);  // Q(b0, b2=0) aligned to P's (b0,b1)
    const u32 qhi = __builtin_amdgcn_perm(Q, Q, q1);  // b2 = 1
    lo = (avg4(P, qlo) >> 2) & 0x3F3F3F3Fu;
    hi = (avg4(P, qhi) >> 2) & 0x3F3F3F3Fu;
}

// Per-lane constants of the pre-pass.  A ds_write_b128 is served in groups of 8 consecutive lanes; with every lane
// storing its classes 0..3 first (16 bytes at lane * 32) lanes k and k + 4 of a group hit the same four banks - a 2-way
// conflict on each of the two stores, ~400 LDS cycles per wave (30 % of the kernel's bank-conflict cycles, round-4 PMC
// attribution in profiles/README.md).  Lanes with bit 2 set therefore produce and store their two halves in the OTHER
// order: the swap costs nothing (it is the order of two v_perm selectors, and two address registers instead of one).
struct PrepassLane {
    u32 q0, q1;      // met8's Q selectors for the first / second store
    u32 off0, off1;  // byte offsets of the two stores inside the 2 KB table
};
DEV PrepassLane prepass_lane(u32 lane) {
    PrepassLane p;
#ifdef VIT_NO_PREPASS_SWAP  /* A/B: the round-3 store order (2-way bank conflict on both stores) */
    const bool sw = false;
#else
    const bool sw = (lane & 4u) != 0;
#endif
    p.q0 = sw ? 0x03020302u : 0x01000100u;
    p.q1 = sw ? 0x01000100u : 0x03020302u;
    p.off0 = lane * 32u + (sw ? 16u : 0u);
    p.off1 = lane * 32u + (sw ? 0u : 16u);
    return p;
}

// Pre-pass for 32 steps: lane = (tau = lane>>1, pair = lane&1) computes the 8 branch metrics of both
// frames of its pair for step t0+tau and writes its 32 table bytes (M only); no cross-lane traffic.
DEV void prepass(u32 sa, u32 sb, char* tab, const PrepassLane& pl, const u32 (&sel)[4]) {
    u32 a0, a1, b0, b1;
    met8(sa, a0, a1, pl.q0, pl.q1);  // frame half 0 (low 16 bits of the ACS registers)
    met8(sb, b0, b1, pl.q0, pl.q1);  // frame half 1
    // sel[k]: byte k of the half-0 word, byte k of the half-1 word, and 0xFF high bytes (+0xFF00) on even steps
    *reinterpret_cast<uint4*>(tab + pl.off0) = make_uint4(__builtin_amdgcn_perm(b0, a0, sel[0]), __builtin_amdgcn_perm(b0, a0, sel[1]),
                                                          __builtin_amdgcn_perm(b0, a0, sel[2]), __builtin_amdgcn_perm(b0, a0, sel[3]));
    *reinterpret_cast<uint4*>(tab + pl.off1) = make_uint4(__builtin_amdgcn_perm(b1, a1, sel[0]), __builtin_amdgcn_perm(b1, a1, sel[1]),
                                                          __builtin_amdgcn_perm(b1, a1, sel[2]), __builtin_amdgcn_perm(b1, a1, sel[3]));
}

// The 4 soft symbols of step t of one frame.  SYM32 = the reference ABI's format (one u32 per symbol, low
// byte used: deconvolve.cpp:158-165) read straight from memory - the ingest narrowing fused into the
// pre-pass (16 B per step instead of 4, no intermediate u8 buffer).  The load returns the RAW words and the
// narrowing happens where the symbols are consumed, 32 steps later: packing right after the load would
// make the wave wait for HBM at every pre-pass.
template <bool SYM32>
struct RawStep {
    typedef u32 type;
};
template <>
struct RawStep<true> {
    typedef uint4 type;
};
template <bool SYM32>
DEV typename RawStep<SYM32>::type load_step(const uint8_t* frame, u32 t, bool valid) {
    typedef typename RawStep<SYM32>::type T;
    if constexpr (!SYM32) {
        return valid ? reinterpret_cast<const T*>(frame)[t] : 0u;
    } else {
        return valid ? reinterpret_cast<const T*>(frame)[t] : make_uint4(0u, 0u, 0u, 0u);
    }
}
DEV u32 pack_step(u32 raw) { return raw; }
DEV u32 pack_step(const uint4& v) {
    return __builtin_amdgcn_perm(v.y, v.x, 0x0C0C0400u) | __builtin_amdgcn_perm(v.w, v.z, 0x04000C0Cu);
}

}  // namespace
