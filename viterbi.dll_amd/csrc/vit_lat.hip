// vit_lat.hip -- latency kernel for small launches: ONE frame per wavefront, ONE path metric per lane,
// no LDS round trip in the add-compare-select dependency chain.
//
// The packed kernel (vit_pk.hip) is built for throughput: 4 frames per wave, ~23 VALU instructions per
// trellis step, and a step's dependent chain runs through ds_swizzle - a lone wave needs ~200 cycles per
// step, 68 us for a FIC frame.  A single `deconvolve()` call, or a launch that cannot fill the chip, is
// bound by that chain, not by issue slots.  Here:
//
// * Lane = position of the trellis state, ROTATING with the step: state bit k sits in lane bit
//   (k - t) mod 6.  The two predecessors (i, i+32) of a butterfly then differ in lane bit j = (5 - t) mod 6
//   only, and both survivors (2i, 2i+1) stay in those same two lanes: a step moves NO metric, it only
//   fetches the partner lane's value (lane ^ 2^j): DPP for j <= 3, v_permlane{16,32}_swap + select for
//   j = 4, 5.  State 0 is always lane 0.
// * n = min3(own + M, partner + (63-M), 255) on plain 32-bit values (= paddusb + pminub: the clamp commutes with
//   the min); renormalisation (state 0 > 150 -> psubusb 63, every second step) is v_cmp + s_bitcmp1 + s_cselect
//   + one v_sub_u32 clamp.
// * Decision bit (tie -> 1, deconvolve.cpp:352-374) = [survivor == the candidate that came from state i+32]: one
//   select by a constant lane mask (the roles of "own" and "partner" swap with lane bit j) and one v_cmp_eq,
//   shifted into a per-lane history word by ONE v_addc_co_u32 (acc = 2*acc + carry-in); one ds_write per 32 steps.
// * Branch metrics: the frame's symbols are staged in LDS once (bulk load, one memory latency - it may be
//   PCIe: deconvolve() hands the kernel mapped host memory); every 64 steps a lane-per-step pre-pass
//   writes the 8 class metrics (M, 63-M) of those steps.
// * Traceback: the blocked speculative scheme of vit_pk.hip (lane = block of BL steps, 30-step warm-up,
//   re-trace until consistent = exactly the serial ChainBack), on the per-lane history words.
//
// ~9 VALU instructions and ~50 cycles of dependent chain per step: a FIC frame in ~20 us.
// Replaces, from scratch, the same reference code as vit_pk.hip (deconvolve.cpp:219-228, 334-435).
#include "vit_internal.h"

namespace {

typedef uint32_t u32;
typedef unsigned long long u64;
#define DEV __device__ __forceinline__

DEV u32 avg4(u32 a, u32 b) { return __builtin_amdgcn_lerp(a, b, 0x01010101u); }  // pavgb x4

// the 8 pavgb-tree metrics of one step (see met8 in vit_pk.hip): bytes of lo = classes 0..3, hi = 4..7
DEV void met8(u32 s, u32& lo, u32& hi) {
    const u32 ns = ~s;
    const u32 r0 = __builtin_amdgcn_perm(ns, s, 0x04000400u);
    const u32 r1 = __builtin_amdgcn_perm(ns, s, 0x05050101u);
    const u32 r2 = __builtin_amdgcn_perm(ns, s, 0x06060202u);
    const u32 r3 = __builtin_amdgcn_perm(ns, s, 0x07030703u);
    const u32 P = avg4(r0, r1), Q = avg4(r2, r3);
    const u32 qlo = __builtin_amdgcn_perm(Q, Q, 0x01000100u);
    const u32 qhi = __builtin_amdgcn_perm(Q, Q, 0x03020302u);
    lo = (avg4(P, qlo) >> 2) & 0x3F3F3F3Fu;
    hi = (avg4(P, qhi) >> 2) & 0x3F3F3F3Fu;
}

// Diagnostic builds only (tools/probe/lat_phases.hip defines VIT_LAT_STAMPS): shader-clock stamps at the phase borders
#ifdef VIT_LAT_STAMPS
__device__ unsigned long long g_lat_stamps[16];
#define LAT_STAMP(k)                                                     \
    do {                                                                 \
        if (threadIdx.x == 0 && blockIdx.x == 0) g_lat_stamps[k] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define LAT_STAMP(k) do {} while (0)
#endif

constexpr u32 CHUNK = 96;       // steps per pre-pass and per unrolled ACS body: a multiple of 6 (phases) and 32 (history words)
constexpr u32 TB_WARM = 30;     // warm-up steps of a speculative traceback block (multiple of 6)

struct LatLayout {
    u32 sym_off, tab_off, dec_off, img_off, scr_off, total;
};
__host__ __device__ inline LatLayout lat_layout(u32 maxfb) {
    const u32 T = maxfb + VIT_TAIL;
    LatLayout l;
    l.sym_off = 0;
    l.tab_off = (T * 4u + 15u) & ~15u;                     // staged symbols: one dword per step
    l.dec_off = l.tab_off + CHUNK * 64u;                   // table: 8 classes x (M, 63-M) per step
    l.img_off = l.dec_off + ((T + CHUNK - 1u) / CHUNK) * (CHUNK / 32u) * 256u;  // history: one dword per lane and 32 steps
    l.scr_off = l.img_off + (((maxfb + 31u) >> 5) + 2u) * 4u;  // output bit image (+ slack for shifted ORs)
    l.total = l.scr_off + 5u * 256u;                            // traceback bit words: <= 144 steps per lane
    return l;
}

// partner lane's value for lane bit J
template <int J>
DEV u32 partner(u32 m, u32 lane) {
    if constexpr (J == 5) {
        auto r = __builtin_amdgcn_permlane32_swap(m, m, false, false);  // r[0] = (lo,lo), r[1] = (hi,hi)
        return lane < 32u ? r[1] : r[0];
    } else if constexpr (J == 4) {
        auto r = __builtin_amdgcn_permlane16_swap(m, m, false, false);  // rows: r[0] = (0,0,2,2), r[1] = (1,1,3,3)
        return (lane & 16u) ? r[0] : r[1];
    } else if constexpr (J == 3) {
        return (u32)__builtin_amdgcn_update_dpp(0, (int)m, 0x128 /*row_ror:8*/, 0xF, 0xF, true);
    } else if constexpr (J == 2) {
        int p = __builtin_amdgcn_update_dpp(0, (int)m, 0x114 /*row_shr:4*/, 0xF, 0xA, false);  // banks 1,3 <- lane-4
        p = __builtin_amdgcn_update_dpp(p, (int)m, 0x104 /*row_shl:4*/, 0xF, 0x5, false);       // banks 0,2 <- lane+4
        return (u32)p;
    } else if constexpr (J == 1) {
        return (u32)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E /*quad_perm:[2,3,0,1]*/, 0xF, 0xF, true);
    } else {
        return (u32)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1 /*quad_perm:[1,0,3,2]*/, 0xF, 0xF, true);
    }
}

// Metrics are plain 32-bit values 0..255 here: measured on one wave alone (tools/probe/latchain.hip,
// profiles/r02_lat_chain_ubench.txt) a dependent 16-bit VOP3 clamp add costs ~9 cycles, a 32-bit VOP2 add or min ~4.7,
// and v_min3_u32 folds the 255 clamp of both candidates into the survivor select.

// Decision bit of a step.  Lanes with bit J clear hold predecessor i: decision = [partner cand <= own cand]
// (m1 <= m0, tie -> 1); lanes with bit J set hold predecessor i+32: decision = [own cand <= partner cand] (m3 <= m2,
// tie -> 1).  Both are "the survivor equals the candidate that came from i+32": decision = [n == hi], where hi is the
// partner's candidate for the first kind of lane and the own candidate for the second - one select by a constant
// lane mask and ONE compare, no scalar instruction (the first version combined two compare masks with four).
// `after` is not used by the instruction: it only pins it behind the value's definition in the schedule.
DEV void push_decisions(u32& acc, u64 d, u32 after = 0) {
    u64 cout;
    asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(acc), "=s"(cout) : "s"(d), "v"(after));  // acc = 2*acc + decision
}

// One trellis step.  RHO = t mod 6 (static): partner bit J = (5 - RHO) mod 6, parity of t = parity of RHO.
// The wave issues in order, so the v_addc that consumes a step's decision mask would stall the next step's metric
// chain behind a VALU -> SGPR -> VALU hand-off: the mask (pd) of step t is therefore consumed one step LATER,
// between the adds and the compare of step t+1 (PENDING = there is one).
template <int RHO, bool PENDING>
DEV void step(u32& m, u32& acc, u64& pd, const char* tabrow, u32 toff, u32 lane, u32 thr) {
    constexpr int J = (5 - RHO + 6) % 6;
    const uint2 X = *reinterpret_cast<const uint2*>(tabrow + toff);  // x: M, y: 63 - M
    const u32 om = m + X.x;                 // candidates, NOT yet clamped (<= 255 + 63)
    const u32 p = partner<J>(m, lane);      // after the own add: m is dead here, the in-place lane swaps need one copy only
    const u32 pm = p + X.y;
    if constexpr (PENDING) push_decisions(acc, pd, pm);  // not before this step's adds have been issued
    const u32 n = min(min(om, pm), 255u);  // v_min3_u32: the survivor with paddusb's clamp, min(min(om,255), min(pm,255))
    const u32 hi = ((lane >> J) & 1u) ? om : pm;         // the candidate from state i+32 (v_cndmask, constant lane mask)
    pd = __builtin_amdgcn_ballot_w64(min(hi, 255u) == n);
    if constexpr (RHO & 1) {
        // Renormalize256 (deconvolve.cpp:407-412: state 0 > 150; the MASM twins test >= 150, decon_avx2.asm:97,114:
        // thr = 150 or 149): state 0 = lane 0
        // (compare in every lane, take lane 0's bit: one hop shorter than v_readfirstlane + scalar compare)
        const u64 gt = __builtin_amdgcn_ballot_w64(n > thr);
        u32 K;
        asm("s_bitcmp1_b32 %1, 0\n\ts_cselect_b32 %0, 63, 0" : "=s"(K) : "s"((u32)gt) : "scc");
        m = __builtin_elementwise_sub_sat(n, K);  // psubusb
    } else {
        m = n;
    }
}

// CHUNK steps, fully unrolled: no branch, table reads at immediate offsets (the scheduler can run them ahead of the
// dependent chain), history word stored once the decisions of every 32nd step are in.
template <int S>
struct ChunkSteps {
    static DEV void run(u32& m, u32& acc, u64& pd, const char* tab, const u32 (&toff)[6], u32 lane, u32* decw, u32 thr) {
        step<S % 6, (S > 0)>(m, acc, pd, tab + S * 64, toff[S % 6], lane, thr);
        if constexpr (S > 0 && S % 32 == 0) decw[(S / 32 - 1) * 64] = acc;  // steps S-32 .. S-1; step t at bit 31 - (t & 31)
        ChunkSteps<S + 1>::run(m, acc, pd, tab, toff, lane, decw, thr);
    }
};
template <>
struct ChunkSteps<(int)CHUNK> {
    static DEV void run(u32&, u32& acc, u64& pd, const char*, const u32 (&)[6], u32, u32* decw, u32) {
        push_decisions(acc, pd);  // the chunk's last step
        decw[(CHUNK / 32u - 1u) * 64u] = acc;
    }
};

// class of this lane's butterfly per phase: state bit k sits in lane bit (k - rho) mod 6
DEV void lat_class_offsets(u32 lane, u32 (&toff)[6]) {
#pragma unroll
    for (int rho = 0; rho < 6; rho++) {
        u32 i = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) i |= ((lane >> ((k - rho + 6) % 6)) & 1u) << k;
        const u32 i0 = i & 1u, i1 = (i >> 1) & 1u, i2 = (i >> 2) & 1u, i3 = (i >> 3) & 1u, i4 = (i >> 4) & 1u;
        const u32 c = (i1 ^ i2 ^ i4) | ((i0 ^ i1 ^ i2) << 1) | ((i0 ^ i3) << 2);  // parity((2i) & poly_j), const.asm:27-63
        toff[rho] = c * 8u;
    }
}

// One frame on one wavefront: `sym` + soff = its symbols (soff counts bytes for the u8 format, symbols for u32),
// o = its (fb+7)/8 output bytes, fb even and 0 < fb <= the framebits `lay` was sized for.
template <bool SYM32>
DEV void lat_decode_frame(const uint8_t* __restrict__ sym, size_t soff, uint8_t* __restrict__ o, u32 fb,
                          const LatLayout& lay, char* lds, const u32 (&toff)[6], u32 lane, u32 renorm_thr) {
    u32* symb = reinterpret_cast<u32*>(lds + lay.sym_off);
    char* tab = lds + lay.tab_off;
    u32* dec = reinterpret_cast<u32*>(lds + lay.dec_off);
    u32* img = reinterpret_cast<u32*>(lds + lay.img_off);
    {
        const u32 T = fb + VIT_TAIL;
        LAT_STAMP(0);

        // ---- stage the frame's symbols: one dword (4 soft symbols, low bytes) per step ----
        // All loads of a batch of 16 x 64 steps are issued before the first one is consumed: a FIC frame costs ONE
        // memory round trip (deconvolve() hands over mapped host memory: that round trip is PCIe).
        constexpr u32 NB = 16;
        if constexpr (SYM32) {
            const uint4* g = reinterpret_cast<const uint4*>(sym) + (soff >> 2);  // soff counts symbols: 4 (one uint4) per step
            for (u32 t0 = 0; t0 < T; t0 += 64u * NB) {
                uint4 v[NB];
#pragma unroll
                for (u32 k = 0; k < NB; k++) {
                    const u32 t = t0 + k * 64u + lane;
                    v[k] = t < T ? g[t] : make_uint4(0u, 0u, 0u, 0u);
                }
#pragma unroll
                for (u32 k = 0; k < NB; k++) {
                    const u32 t = t0 + k * 64u + lane;
                    if (t < T)
                        symb[t] = __builtin_amdgcn_perm(v[k].y, v[k].x, 0x0C0C0400u) | __builtin_amdgcn_perm(v[k].w, v[k].z, 0x04000C0Cu);
                }
            }
        } else {
            const u32* g = reinterpret_cast<const u32*>(sym + soff);
            for (u32 t0 = 0; t0 < T; t0 += 64u * NB) {
                u32 v[NB];
#pragma unroll
                for (u32 k = 0; k < NB; k++) {
                    const u32 t = t0 + k * 64u + lane;
                    v[k] = t < T ? g[t] : 0u;
                }
#pragma unroll
                for (u32 k = 0; k < NB; k++) {
                    const u32 t = t0 + k * 64u + lane;
                    if (t < T) symb[t] = v[k];
                }
            }
        }
        for (u32 i = lane; i < ((fb + 31u) >> 5) + 2u; i += 64u) img[i] = 0;
        __syncthreads();
        LAT_STAMP(1);

        // ---- ACS ----
        u32 m = lane == 0 ? 0u : 63u;  // const.asm:19-25 (0-based; step 0 is even)
        u32 acc = 0;
        // pre-pass: lane = step; the 8 class metrics (M, 63-M) of CHUNK steps -> tab[step][class]
        auto prepass_row = [&](u32 t0, u32 row) {
            const u32 t = t0 + row;
            const u32 s = t < T ? symb[t] : 0u;  // steps past the frame's end: run, never traced back
            u32 lo, hi;
            met8(s, lo, hi);
            u32 M[8];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                M[c] = (lo >> (8 * c)) & 0xFFu;
                M[4 + c] = (hi >> (8 * c)) & 0xFFu;
            }
            uint4* dst = reinterpret_cast<uint4*>(tab + row * 64u);
#pragma unroll
            for (int c = 0; c < 8; c += 2) dst[c >> 1] = make_uint4(M[c], 63u - M[c], M[c + 1], 63u - M[c + 1]);
        };
        const u32 nch = (T + CHUNK - 1u) / CHUNK;
        for (u32 ch = 0; ch < nch; ch++) {
            __syncthreads();  // the previous chunk's table has been read
            prepass_row(ch * CHUNK, lane);
            if (lane < CHUNK - 64u) prepass_row(ch * CHUNK, 64u + lane);
            __syncthreads();
            u64 pd = 0;
            ChunkSteps<0>::run(m, acc, pd, tab, toff, lane, dec + ch * (CHUNK / 32u) * 64u + lane, renorm_thr);
        }
        __syncthreads();
        LAT_STAMP(2);

        // ---- traceback: lane q takes steps [6 + q*BL, 6 + (q+1)*BL) ----
        // One step back from time t: j = (5 - t) mod 6, the decision d of the state on the path (lane L) is
        // bit 31-(t&31) of dec[t>>5][L]; the predecessor is L with bit j := d, and d is decoded bit t-6
        // (ChainBack, deconvolve.cpp:416-435, seen through the rotating lane <-> state map).
        const u32 BL = 6u * ((fb + 64u * 6u - 1u) / (64u * 6u));
        const u32 tbase = VIT_TAIL + lane * BL;
        const bool has_work = tbase < T;
        const u32 i_last = has_work ? T - 1u - tbase : 0u;  // block-relative index of the frame's last step
        const u32 q_top = (T - 1u - VIT_TAIL) / BL;
        const u32 i_warm = BL - 1u + TB_WARM;
        const u32 i_start = i_last < i_warm ? i_last : i_warm;
        const bool fixed = has_work && i_last <= i_warm;  // starts from the true end state (0): never re-traced
        u32* scr = reinterpret_cast<u32*>(lds + lay.scr_off) + lane;  // decoded bits of this block, word x at scr[x*64]
        // i_from + 1 and i_to are multiples of 6 and tbase = 0 mod 6: index ii = i - k has phase j = k
        auto trace = [&](u32& L, int i_from, int i_to, bool on, u32 i_max, bool record) {
            u32 cur = 0;
            for (int i = i_from; i >= i_to; i -= 6) {
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const int ii = i - k;
                    u32 d = 0;
                    if (on && (u32)ii <= i_max) {
                        const u32 t = tbase + (u32)ii;
                        const u32 w = dec[(t >> 5) * 64u + L];
                        d = (w >> (31u - (t & 31u))) & 1u;
                        L = (L & ~(1u << k)) | (d << k);
                    }
                    if (record) {
                        cur |= d << (ii & 31);
                        if ((ii & 31) == 0) {
                            if (on) scr[(ii >> 5) * 64] = cur;
                            cur = 0;
                        }
                    }
                }
            }
        };
        u32 L = 0, L_out = 0;
        trace(L, (int)i_warm, (int)BL, has_work, i_start, false);
        u32 L_in = L;
        trace(L, (int)BL - 1, 0, has_work, i_start, true);
        if (has_work) L_out = L;
        LAT_STAMP(3);
        for (int pass = 0; pass < 65; pass++) {
            const u32 nxt = __shfl_down(L_out, 1);
            const u32 new_in = (lane < q_top) ? nxt : 0u;
            const bool changed = has_work && !fixed && new_in != L_in;
            if (!__any(changed)) break;
            if (changed) L_in = new_in;
            L = new_in;
            trace(L, (int)BL - 1, 0, changed, BL - 1u, true);
            if (changed) L_out = L;
        }
        LAT_STAMP(4);
        if (has_work) {
            const u32 nvalid = i_last + 1u < BL ? i_last + 1u : BL;
            const u32 b0 = lane * BL;  // decoded bit index of the block's first step
            for (u32 x = 0; x < (BL + 31u) >> 5; x++) {
                const u32 lo = 32u * x;
                if (lo < nvalid) {
                    const u32 cnt = nvalid - lo;
                    const u32 val = scr[x * 64u] & (cnt >= 32u ? 0xFFFFFFFFu : ((1u << cnt) - 1u));
                    const u32 b = b0 + lo, dw = b >> 5, sft = b & 31u;
                    if (val) {
                        atomicOr(&img[dw], val << sft);
                        if (sft) atomicOr(&img[dw + 1u], val >> (32u - sft));
                    }
                }
            }
        }
        __syncthreads();
        // bit b of the image is decoded bit b; output bytes are MSB-first (deconvolve.cpp:432-433)
        const u32 nbytes = (fb + 7u) >> 3;
        if (((reinterpret_cast<uintptr_t>(o) | nbytes) & 3u) == 0) {
            for (u32 k = lane; k < (nbytes >> 2); k += 64u)
                reinterpret_cast<u32*>(o)[k] = __builtin_bswap32(__builtin_bitreverse32(img[k]));
        } else {
            for (u32 k = lane; k < nbytes; k += 64u) {
                const u32 byte = (img[k >> 2] >> (8u * (k & 3u))) & 0xFFu;
                o[k] = (uint8_t)(__builtin_bitreverse32(byte) >> 24);
            }
        }
        __syncthreads();
        LAT_STAMP(5);
    }
}

template <bool SYM32>
__global__ __launch_bounds__(64) void vit_lat_kernel(const uint8_t* __restrict__ sym, uint8_t* __restrict__ out,
                                                     const vit_frame_desc* __restrict__ desc, u32 framebits_uniform,
                                                     u32 max_framebits, long long nframes, LatLayout lay,
                                                     u32* done_flag, u32 done_seq, u32 renorm_thr) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const u32 lane = threadIdx.x;
    u32 toff[6];
    lat_class_offsets(lane, toff);

    for (long long f = blockIdx.x; f < nframes; f += gridDim.x) {
        u32 fb = framebits_uniform;
        size_t soff, ooff;
        if (desc) {
            fb = desc[f].framebits;
            soff = desc[f].sym_offset;
            ooff = desc[f].out_offset;
            if (fb > max_framebits || (fb & 1u) || (soff & 3u)) continue;  // not what the launch was sized for
        } else {
            soff = (size_t)f * 4u * (fb + VIT_TAIL);
            ooff = (size_t)f * ((fb + 7u) >> 3);
        }
        if (fb == 0) continue;
        lat_decode_frame<SYM32>(sym, soff, out + ooff, fb, lay, lds, toff, lane, renorm_thr);
    }
    // Completion flag for the single-call path (deconvolve()): the host spins on a word of its mapped staging
    // buffer instead of waiting for the end-of-kernel signal.  Every output byte of this wave is written (and
    // made visible system-wide by the release) before the flag.
    if (done_flag && lane == 0)
        __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Ingest-stage launch (vit_api.hip): workgroup b decodes the frame a deconvolve() caller has put into
// slot tbl.slot[b] of the mapped pinned ring (symbols already narrowed to one byte each by the caller's copy) and
// publishes the batch's sequence number in the slot's completion word after its last output byte, with system
// scope - the caller spins on that word.  One bounded launch per batch, no polling on the device.
__global__ __launch_bounds__(64) void vit_lat_ring_kernel(uint8_t* __restrict__ ring, VitRingTable tbl, LatLayout lay,
                                                          u32 renorm_thr) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const u32 lane = threadIdx.x;
    const u32 b = blockIdx.x;
    if (b >= tbl.n) return;
    u32 toff[6];
    lat_class_offsets(lane, toff);
    const u32 slot = tbl.slot[b], fb = tbl.fb[b];
    uint8_t* base = ring + (size_t)slot * tbl.stride;
    lat_decode_frame<false>(base, 0, base + tbl.out_off, fb, lay, lds, toff, lane, renorm_thr);
    if (lane == 0)
        __hip_atomic_store(reinterpret_cast<u32*>(base + tbl.flag_off), tbl.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

int64_t vit_lat_capacity(uint32_t max_framebits, int dev) {
    const u32 per_cu = (160u * 1024u) / lat_layout(max_framebits).total;  // workgroups (= waves) per CU the LDS admits
    const int64_t cap = (int64_t)(per_cu < 8u ? per_cu : 8u) * vit_device_cus(dev);  // at most two waves per SIMD
    return cap < VIT_LAT_MAX_FRAMES ? cap : VIT_LAT_MAX_FRAMES;
}

hipError_t vit_launch_lat(const void* d_symbols, bool sym32, uint8_t* d_out, const vit_frame_desc* d_desc,
                          uint32_t framebits, uint32_t max_framebits, int64_t nframes, hipStream_t stream,
                          uint32_t* done_flag, uint32_t done_seq, bool renorm_ge) {
    const u32 thr = renorm_ge ? 149u : 150u;  // the kernels test `> thr`
    if (done_flag && nframes != 1) return hipErrorInvalidValue;  // one workgroup writes the flag
    if (nframes <= 0) return hipSuccess;
    if (sym32 && (reinterpret_cast<uintptr_t>(d_symbols) & 15u)) return hipErrorInvalidValue;  // uint4 loads
    static uint64_t optin_done = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const void* ks[2] = {reinterpret_cast<const void*>(vit_lat_kernel<false>),
                         reinterpret_cast<const void*>(vit_lat_kernel<true>)};
    if ((e = vit_optin_dynamic_lds(ks, 2, 160 * 1024, dev, &optin_done)) != hipSuccess) return e;
    const LatLayout lay = lat_layout(max_framebits);
    const long long grid = nframes < (1 << 20) ? nframes : (1 << 20);
    const uint8_t* d_sym = static_cast<const uint8_t*>(d_symbols);
    if (sym32)
        hipLaunchKernelGGL(vit_lat_kernel<true>, dim3((unsigned)grid), dim3(64), lay.total, stream, d_sym, d_out, d_desc,
                           framebits, max_framebits, (long long)nframes, lay, done_flag, done_seq, thr);
    else
        hipLaunchKernelGGL(vit_lat_kernel<false>, dim3((unsigned)grid), dim3(64), lay.total, stream, d_sym, d_out, d_desc,
                           framebits, max_framebits, (long long)nframes, lay, done_flag, done_seq, thr);
    return hipGetLastError();
}

hipError_t vit_launch_lat_ring(uint8_t* d_ring, const VitRingTable& tbl, uint32_t max_framebits, hipStream_t stream,
                               bool renorm_ge) {
    if (tbl.n == 0) return hipSuccess;
    if (tbl.n > VIT_RING_MAXB || (reinterpret_cast<uintptr_t>(d_ring) & 15u) || (tbl.stride & 15u) || (tbl.out_off & 3u) ||
        (tbl.flag_off & 3u))
        return hipErrorInvalidValue;
    for (u32 i = 0; i < tbl.n; i++)
        if (tbl.fb[i] == 0 || tbl.fb[i] > max_framebits || (tbl.fb[i] & 1u)) return hipErrorInvalidValue;
    static uint64_t optin_done = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const void* ks[1] = {reinterpret_cast<const void*>(vit_lat_ring_kernel)};
    if ((e = vit_optin_dynamic_lds(ks, 1, 160 * 1024, dev, &optin_done)) != hipSuccess) return e;
    const LatLayout lay = lat_layout(max_framebits);
    hipLaunchKernelGGL(vit_lat_ring_kernel, dim3(tbl.n), dim3(64), lay.total, stream, d_ring, tbl, lay,
                       renorm_ge ? 149u : 150u);
    return hipGetLastError();
}
