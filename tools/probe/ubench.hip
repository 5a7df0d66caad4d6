// VALU issue-rate microbenchmark for the instructions of the packed ACS loop (gfx950).
// Each kernel runs REP x 8 independent instructions per lane; grid fills every SIMD
// with WPS waves.  Prints cycles per wave-instruction per SIMD (2.0 = full rate).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int REP = 2000;

#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define KERNEL(NAME, INS)                                                             \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {       \
        uint32_t rr = 0; (void)rr; uint32_t r[8], a = threadIdx.x * 2654435761u + seed, b = a ^ 0x5bd1e995u;     \
        for (int i = 0; i < 8; i++) r[i] = a + i * 77u;                               \
        for (int it = 0; it < REP; it++) { BODY8(INS) }                               \
        uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= r[i];                        \
        if (s == 0x12345678u) out[threadIdx.x] = s;                                    \
    }
#define I_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define I_PKADD(i) asm volatile("v_pk_add_u16 %0, %0, %1 clamp" : "+v"(r[i]) : "v"(b));
#define I_PKMIN(i) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define I_PKSUB(i) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define I_PKLSHR(i) asm volatile("v_pk_lshrrev_b16 %0, 1, %0 op_sel_hi:[0,1]" : "+v"(r[i]));
#define I_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(b));
#define I_DPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(r[i]) : "v"(b));
#define I_CNDDPP(i) asm volatile("v_cndmask_b32_dpp %0, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]) : "v"(b) : "vcc");
#define I_SWAP(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r[i]), "+v"(b));
#define I_LERP(i) asm volatile("v_lerp_u8 %0, %0, %1, %1" : "+v"(r[i]) : "v"(b));
#define I_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(b));
#define I_SMOV(i) asm volatile("s_mov_b64 vcc, exec" ::: "vcc");
#define I_SNOP(i) asm volatile("s_nop 0");
#define I_MIX(i) asm volatile("v_pk_add_u16 %0, %0, %1 clamp\n\ts_mov_b64 vcc, exec" : "+v"(r[i]) : "v"(b) : "vcc");
#define I_SWZ(i) asm volatile("ds_swizzle_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(r[i]));
KERNEL(k_add, I_ADD) KERNEL(k_pkadd, I_PKADD) KERNEL(k_pkmin, I_PKMIN) KERNEL(k_pksub, I_PKSUB)
KERNEL(k_pklshr, I_PKLSHR) KERNEL(k_andor, I_ANDOR) KERNEL(k_dpp, I_DPP) KERNEL(k_cnddpp, I_CNDDPP)
KERNEL(k_swap, I_SWAP) KERNEL(k_lerp, I_LERP) KERNEL(k_perm, I_PERM) KERNEL(k_smov, I_SMOV)
KERNEL(k_snop, I_SNOP) KERNEL(k_mix, I_MIX) KERNEL(k_swz, I_SWZ)


#define I_ADD64(i) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define I_MINU32(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define I_MINU16(i) asm volatile("v_min_u16 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define I_ADDU16(i) asm volatile("v_add_u16 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define I_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define I_LSHR(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r[i]));
#define I_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i]) : "v"(b));
#define I_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(b));
#define I_CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(r[i]) : "v"(b));
#define I_SDWA(i) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(r[i]) : "v"(b));
#define I_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(r[i]));
#define I_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 1" : "+v"(r[i]) : "v"(b));
#define I_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(b));
#define I_MIN3(i) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(b));
#define I_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r[i]) : "v"(b));
#define I_CMP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(r[i]), "v"(b) : "vcc");
#define I_CMP64(i) asm volatile("v_cmp_lt_u32_e64 s[10:11], %0, %1" :: "v"(r[i]), "v"(b) : "s10", "s11");
#define I_CMPSDWA(i) asm volatile("v_cmp_lt_u16_sdwa vcc, %0, %1 src0_sel:WORD_1 src1_sel:WORD_1" :: "v"(r[i]), "v"(b) : "vcc");
#define I_ADDC(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(r[i]) :: "vcc");
#define I_READLANE(i) asm volatile("v_readlane_b32 s10, %0, 0" :: "v"(r[i]) : "s10");
#define I_PKADDNC(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define I_PKMAD(i) asm volatile("v_pk_mad_u16 %0, %0, %1, %1" : "+v"(r[i]) : "v"(b));
#define I_MOVDPPQ(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(b));
#define I_ADDDPP(i) asm volatile("v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(b));
#define I_MINDPP(i) asm volatile("v_min_u32_dpp %0, %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(b));
#define I_SWAP32(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(b));
#define I_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(r[i]) : "v"(b));
#define I_DSRD(i) asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rr) : "v"(b & 0xff8));
#define I_SADD(i) asm volatile("s_add_u32 s10, s10, 1" ::: "s10", "scc");
#define I_VSUBREV(i) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(r[i]) : "v"(b));
#define I_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define I_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(b) : "vcc");
#define I_SAT8(i) asm volatile("v_add_u16 %0, %0, %1 clamp" : "+v"(r[i]) : "v"(b));
KERNEL(k_add64, I_ADD64) KERNEL(k_minu32, I_MINU32) KERNEL(k_minu16, I_MINU16) KERNEL(k_addu16, I_ADDU16)
KERNEL(k_and, I_AND) KERNEL(k_lshr, I_LSHR) KERNEL(k_mov, I_MOV) KERNEL(k_cnd, I_CND) KERNEL(k_cnd64, I_CND64)
KERNEL(k_sdwa, I_SDWA) KERNEL(k_bfe, I_BFE) KERNEL(k_align, I_ALIGN) KERNEL(k_add3, I_ADD3) KERNEL(k_min3, I_MIN3)
KERNEL(k_lshlor, I_LSHLOR) KERNEL(k_cmp, I_CMP) KERNEL(k_cmp64, I_CMP64) KERNEL(k_cmpsdwa, I_CMPSDWA)
KERNEL(k_addc, I_ADDC) KERNEL(k_readlane, I_READLANE) KERNEL(k_pkaddnc, I_PKADDNC) KERNEL(k_pkmad, I_PKMAD)
KERNEL(k_movdppq, I_MOVDPPQ) KERNEL(k_adddpp, I_ADDDPP) KERNEL(k_mindpp, I_MINDPP) KERNEL(k_swap32, I_SWAP32)
KERNEL(k_bperm, I_BPERM) KERNEL(k_sadd, I_SADD) KERNEL(k_subrev, I_VSUBREV) KERNEL(k_xor, I_XOR)
KERNEL(k_addco, I_ADDCO) KERNEL(k_sat16, I_SAT8)

typedef void (*kfn)(uint32_t*, uint32_t);
int main() {
    uint32_t* d; CK(hipMalloc(&d, 4096));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    const double mhz = pr.clockRate / 1000.0;
    struct { const char* n; kfn f; int per; } ks[] = {
        {"v_add_u32", k_add, 1}, {"v_pk_add_u16 clamp", k_pkadd, 1}, {"v_pk_min_u16", k_pkmin, 1},
        {"v_pk_sub_i16", k_pksub, 1}, {"v_pk_lshrrev_b16", k_pklshr, 1}, {"v_and_or_b32", k_andor, 1},
        {"v_mov_b32_dpp", k_dpp, 1}, {"v_cndmask_b32_dpp", k_cnddpp, 1}, {"v_permlane16_swap", k_swap, 1},
        {"v_lerp_u8", k_lerp, 1}, {"v_perm_b32", k_perm, 1}, {"s_mov_b64", k_smov, 1}, {"s_nop 0", k_snop, 1},
        {"pk_add + s_mov pair", k_mix, 1}, {"ds_swizzle+wait", k_swz, 1},
        {"v_add_u32_e64", k_add64, 1}, {"v_min_u32", k_minu32, 1}, {"v_min_u16", k_minu16, 1}, {"v_add_u16", k_addu16, 1},
        {"v_add_u16 clamp(e64)", k_sat16, 1}, {"v_and_b32", k_and, 1}, {"v_xor_b32", k_xor, 1}, {"v_lshrrev_b32", k_lshr, 1},
        {"v_mov_b32", k_mov, 1}, {"v_subrev_u32", k_subrev, 1}, {"v_cndmask_b32 vcc", k_cnd, 1}, {"v_cndmask_b32_e64", k_cnd64, 1},
        {"v_add_u32_sdwa", k_sdwa, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_alignbit_b32", k_align, 1}, {"v_add3_u32", k_add3, 1},
        {"v_min3_u32", k_min3, 1}, {"v_lshl_or_b32", k_lshlor, 1}, {"v_cmp_lt_u32 vcc", k_cmp, 1},
        {"v_cmp_lt_u32_e64 sgpr", k_cmp64, 1}, {"v_cmp_lt_u16_sdwa", k_cmpsdwa, 1}, {"v_addc_co_u32", k_addc, 1},
        {"v_add_co_u32", k_addco, 1}, {"v_readlane_b32", k_readlane, 1}, {"v_pk_add_u16 (no clamp)", k_pkaddnc, 1},
        {"v_pk_mad_u16", k_pkmad, 1}, {"v_mov_b32_dpp quad", k_movdppq, 1}, {"v_add_u32_dpp", k_adddpp, 1},
        {"v_min_u32_dpp ror8", k_mindpp, 1}, {"v_permlane32_swap", k_swap32, 1}, {"ds_bpermute+wait", k_bperm, 1},
        {"s_add_u32", k_sadd, 1}};
    printf("device %s, %d CUs, clock %.0f MHz\n", pr.gcnArchName, cus, mhz);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wps : {2, 4}) {  // waves per SIMD
        printf("-- %d wave(s) per SIMD --\n", wps);
        for (auto& k : ks) {
            const int blocks = cus * wps;  // 256-thread blocks = 4 waves = one per SIMD
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d, 1u);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d, 1u);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            const double instr_per_simd = (double)REP * 8 * wps;
            printf("%-22s %8.3f ms  -> %.2f cycles per wave-instr per SIMD (at %.0f MHz nominal)\n", k.n, ms,
                   ms * 1e-3 * mhz * 1e6 / instr_per_simd, mhz);
        }
    }
    return 0;
}
