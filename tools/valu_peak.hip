// valu_peak.hip — measures per-instruction VALU issue rates on gfx950 for the ops the
// Smith-Waterman kernel is made of.  Build+run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o valu_peak tools/valu_peak.hip && ./valu_peak
// Output: lane-ops/s (wave64 instr x 64) and cycles per wave-instruction per SIMD
// at the measured clock (s_memtime) for 8 independent chains per thread, 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 4096

#define KERNEL(NAME, ASM)                                                                  \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {            \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7,           \
                 a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;                    \
        uint32_t b = seed | 0x00010001u, c = (seed * 31) | 0x00020002u;                     \
        for (int i = 0; i < ITERS; ++i) {                                                  \
            asm volatile(ASM(%0) "\n" ASM(%1) "\n" ASM(%2) "\n" ASM(%3) "\n" ASM(%4) "\n"    \
                         ASM(%5) "\n" ASM(%6) "\n" ASM(%7)                                   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),       \
                           "+v"(a6), "+v"(a7)                                                \
                         : "v"(b), "v"(c) : "vcc", "s20", "s21");                                 \
        }                                                                                  \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;       \
    }

#define OP_PKMAX(x) "v_pk_max_i16 " #x ", " #x ", %8"
#define OP_PKSUBC(x) "v_pk_sub_u16 " #x ", " #x ", %8 clamp"
#define OP_PKADD(x) "v_pk_add_u16 " #x ", " #x ", %8"
#define OP_PKMIN(x) "v_pk_min_u16 " #x ", " #x ", %8"
#define OP_PKMAD(x) "v_pk_mad_i16 " #x ", " #x ", %8, %9"
#define OP_XOR(x) "v_xor_b32 " #x ", " #x ", %8"
#define OP_MAXI32(x) "v_max_i32 " #x ", " #x ", %8"
#define OP_ADDU32(x) "v_add_u32 " #x ", " #x ", %8"
#define OP_MAX3(x) "v_max3_i32 " #x ", " #x ", %8, %9"
#define OP_DPP(x) "v_mov_b32_dpp " #x ", " #x " row_shr:1 row_mask:0xf bank_mask:0xf"
#define OP_FMA(x) "v_fma_f32 " #x ", " #x ", %8, %9"
#define OP_PKFMA(x) "v_pk_fma_f32 " #x ", " #x ", %8, %9"   /* placeholder: invalid on 1 reg, not used */
#define OP_MAXDPP(x) "v_max_i32_dpp " #x ", " #x ", %8 row_shr:1 row_mask:0xf bank_mask:0xf"
#define OP_PERM(x) "v_perm_b32 " #x ", " #x ", %8, %9"
#define OP_BFI(x) "v_bfi_b32 " #x ", %8, " #x ", %9"
#define OP_CNDMASK(x) "v_cndmask_b32 " #x ", " #x ", %8, vcc"
#define OP_SAD(x) "v_sad_u8 " #x ", " #x ", %8, %9"
#define OP_MAXU16(x) "v_max_u16 " #x ", " #x ", %8"


#define OP_MAX3F(x) "v_max3_f32 " #x ", " #x ", %8, %9"
#define OP_MAXF(x) "v_max_f32 " #x ", " #x ", %8"
#define OP_ADDF(x) "v_add_f32 " #x ", " #x ", %8"
#define OP_MED3F(x) "v_med3_f32 " #x ", " #x ", %8, %9"
#define OP_MAX3I16(x) "v_max3_i16 " #x ", " #x ", %8, %9"
#define OP_MAX3U16(x) "v_max3_u16 " #x ", " #x ", %8, %9"
#define OP_MAXI16(x) "v_max_i16 " #x ", " #x ", %8"
#define OP_SUBU16(x) "v_sub_u16 " #x ", " #x ", %8"
#define OP_SUBU16C(x) "v_sub_u16 " #x ", " #x ", %8 clamp"
#define OP_ADDU16(x) "v_add_u16 " #x ", " #x ", %8"
#define OP_MINU16(x) "v_min_u16 " #x ", " #x ", %8"
#define OP_MADU16(x) "v_mad_u16 " #x ", " #x ", %8, %9"
#define OP_MADI16(x) "v_mad_i16 " #x ", " #x ", %8, %9"
#define OP_ADDU16S(x) "v_add_u16_sdwa " #x ", " #x ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
#define OP_MAXU32(x) "v_max_u32 " #x ", " #x ", %8"
#define OP_SUBU32(x) "v_sub_u32 " #x ", " #x ", %8"
#define OP_ADD3(x) "v_add3_u32 " #x ", " #x ", %8, %9"
#define OP_AND(x) "v_and_b32 " #x ", " #x ", %8"
#define OP_ANDOR(x) "v_and_or_b32 " #x ", " #x ", %8, %9"
#define OP_MOV(x) "v_mov_b32 " #x ", %8"
#define OP_LSHLADD(x) "v_lshl_add_u32 " #x ", " #x ", 1, %8"
#define OP_MAXF16(x) "v_max_f16 " #x ", " #x ", %8"
#define OP_PKMAXF16(x) "v_pk_max_f16 " #x ", " #x ", %8"
#define OP_PKADDF16(x) "v_pk_add_f16 " #x ", " #x ", %8"
#define OP_PKFMAF16(x) "v_pk_fma_f16 " #x ", " #x ", %8, %9"
#define OP_MAX3F16(x) "v_max3_f16 " #x ", " #x ", %8, %9"
#define OP_SUBREVC(x) "v_sub_i16 " #x ", " #x ", %8 clamp"
#define OP_MIN3U16(x) "v_min3_u16 " #x ", " #x ", %8, %9"
#define OP_PKMAXU16(x) "v_pk_max_u16 " #x ", " #x ", %8"
#define OP_DOT(x) "v_dot2_i32_i16 " #x ", " #x ", %8, %9"
KERNEL(k_pkmax, OP_PKMAX)
KERNEL(k_pksubc, OP_PKSUBC)
KERNEL(k_pkadd, OP_PKADD)
KERNEL(k_pkmin, OP_PKMIN)
KERNEL(k_pkmad, OP_PKMAD)
KERNEL(k_xor, OP_XOR)
KERNEL(k_maxi32, OP_MAXI32)
KERNEL(k_addu32, OP_ADDU32)
KERNEL(k_max3, OP_MAX3)
KERNEL(k_dpp, OP_DPP)
KERNEL(k_fma, OP_FMA)
KERNEL(k_maxdpp, OP_MAXDPP)
KERNEL(k_perm, OP_PERM)
KERNEL(k_bfi, OP_BFI)
KERNEL(k_cnd, OP_CNDMASK)
KERNEL(k_sad, OP_SAD)
KERNEL(k_maxu16, OP_MAXU16)

KERNEL(k_max3f, OP_MAX3F)
KERNEL(k_maxf, OP_MAXF)
KERNEL(k_addf, OP_ADDF)
KERNEL(k_med3f, OP_MED3F)
KERNEL(k_max3i16, OP_MAX3I16)
KERNEL(k_max3u16, OP_MAX3U16)
KERNEL(k_maxi16, OP_MAXI16)
KERNEL(k_subu16, OP_SUBU16)
KERNEL(k_subu16c, OP_SUBU16C)
KERNEL(k_addu16, OP_ADDU16)
KERNEL(k_minu16, OP_MINU16)
KERNEL(k_madu16, OP_MADU16)
KERNEL(k_madi16, OP_MADI16)
KERNEL(k_addu16s, OP_ADDU16S)
KERNEL(k_maxu32, OP_MAXU32)
KERNEL(k_subu32, OP_SUBU32)
KERNEL(k_add3, OP_ADD3)
KERNEL(k_and, OP_AND)
KERNEL(k_andor, OP_ANDOR)
KERNEL(k_mov, OP_MOV)
KERNEL(k_lshladd, OP_LSHLADD)
KERNEL(k_maxf16, OP_MAXF16)
KERNEL(k_pkmaxf16, OP_PKMAXF16)
KERNEL(k_pkaddf16, OP_PKADDF16)
KERNEL(k_pkfmaf16, OP_PKFMAF16)
KERNEL(k_max3f16, OP_MAX3F16)
KERNEL(k_subi16c, OP_SUBREVC)
KERNEL(k_min3u16, OP_MIN3U16)
KERNEL(k_pkmaxu16, OP_PKMAXU16)
#define OP_X_LSHL(x) "v_lshlrev_b32 " #x ", 1, " #x
#define OP_X_LSHR(x) "v_lshrrev_b32 " #x ", 1, " #x
#define OP_X_OR(x) "v_or_b32 " #x ", " #x ", %8"
#define OP_X_NOT(x) "v_not_b32 " #x ", " #x
#define OP_X_FFBL(x) "v_ffbl_b32 " #x ", " #x
#define OP_X_BFE(x) "v_bfe_u32 " #x ", " #x ", 3, 8"
#define OP_X_ALIGNBIT(x) "v_alignbit_b32 " #x ", " #x ", %8, 7"
#define OP_X_MULLO(x) "v_mul_lo_u32 " #x ", " #x ", %8"
#define OP_X_MUL24(x) "v_mul_u32_u24 " #x ", " #x ", %8"
#define OP_X_MAD24(x) "v_mad_u32_u24 " #x ", " #x ", %8, %9"
#define OP_X_OR3(x) "v_or3_b32 " #x ", " #x ", %8, %9"
#define OP_X_LSHLOR(x) "v_lshl_or_b32 " #x ", " #x ", 3, %8"
#define OP_X_BCNT(x) "v_bcnt_u32_b32 " #x ", " #x ", %8"
#define OP_X_MINI32(x) "v_min_i32 " #x ", " #x ", %8"
#define OP_X_MINU32(x) "v_min_u32 " #x ", " #x ", %8"
#define OP_X_CMPCND(x) "v_cmp_lt_u32 vcc, " #x ", %8\n v_cndmask_b32 " #x ", " #x ", %9, vcc"
#define OP_X_CMPE64(x) "v_cmp_eq_u32 s[20:21], " #x ", %8\n v_cndmask_b32 " #x ", " #x ", %9, s[20:21]"
#define OP_X_SDWAADD(x) "v_add_u32_sdwa " #x ", " #x ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
#define OP_X_XAD(x) "v_xad_u32 " #x ", " #x ", %8, %9"
KERNEL(k_x_lshl, OP_X_LSHL)
KERNEL(k_x_lshr, OP_X_LSHR)
KERNEL(k_x_or, OP_X_OR)
KERNEL(k_x_not, OP_X_NOT)
KERNEL(k_x_ffbl, OP_X_FFBL)
KERNEL(k_x_bfe, OP_X_BFE)
KERNEL(k_x_alignbit, OP_X_ALIGNBIT)
KERNEL(k_x_mullo, OP_X_MULLO)
KERNEL(k_x_mul24, OP_X_MUL24)
KERNEL(k_x_mad24, OP_X_MAD24)
KERNEL(k_x_or3, OP_X_OR3)
KERNEL(k_x_lshlor, OP_X_LSHLOR)
KERNEL(k_x_bcnt, OP_X_BCNT)
KERNEL(k_x_mini32, OP_X_MINI32)
KERNEL(k_x_minu32, OP_X_MINU32)
KERNEL(k_x_cmpcnd, OP_X_CMPCND)
KERNEL(k_x_cmpe64, OP_X_CMPE64)
KERNEL(k_x_sdwaadd, OP_X_SDWAADD)
KERNEL(k_x_xad, OP_X_XAD)
typedef void (*kern_t)(uint32_t*, uint32_t);

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz\n", p.gcnArchName, cus, p.clockRate);
    const int blocks = cus * 8 * 4;   // 8 blocks of 4 waves per CU resident (8 waves/SIMD), x4 rounds
    uint32_t* d;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    struct { const char* name; kern_t k; } ks[] = {
        {"v_pk_max_i16", k_pkmax}, {"v_pk_sub_u16 clamp", k_pksubc}, {"v_pk_add_u16", k_pkadd},
        {"v_pk_min_u16", k_pkmin}, {"v_pk_mad_i16", k_pkmad}, {"v_xor_b32", k_xor}, {"v_max_i32", k_maxi32},
        {"v_add_u32", k_addu32}, {"v_max3_i32", k_max3}, {"v_mov_b32_dpp row_shr:1", k_dpp}, {"v_fma_f32", k_fma},
        {"v_max_i32_dpp", k_maxdpp}, {"v_perm_b32", k_perm}, {"v_bfi_b32", k_bfi}, {"v_cndmask_b32", k_cnd},
        {"v_sad_u8", k_sad}, {"v_max_u16", k_maxu16}, {"v_max3_f32", k_max3f}, {"v_max_f32", k_maxf}, {"v_add_f32", k_addf}, {"v_med3_f32", k_med3f}, {"v_max3_i16", k_max3i16}, {"v_max3_u16", k_max3u16}, {"v_max_i16", k_maxi16}, {"v_sub_u16", k_subu16}, {"v_sub_u16 clamp", k_subu16c}, {"v_add_u16", k_addu16}, {"v_min_u16", k_minu16}, {"v_mad_u16", k_madu16}, {"v_mad_i16", k_madi16}, {"v_add_u16_sdwa", k_addu16s}, {"v_max_u32", k_maxu32}, {"v_sub_u32", k_subu32}, {"v_add3_u32", k_add3}, {"v_and_b32", k_and}, {"v_and_or_b32", k_andor}, {"v_mov_b32", k_mov}, {"v_lshl_add_u32", k_lshladd}, {"v_max_f16", k_maxf16}, {"v_pk_max_f16", k_pkmaxf16}, {"v_pk_add_f16", k_pkaddf16}, {"v_pk_fma_f16", k_pkfmaf16}, {"v_max3_f16", k_max3f16}, {"v_sub_i16 clamp", k_subi16c}, {"v_min3_u16", k_min3u16}, {"v_pk_max_u16", k_pkmaxu16}, {"v_lshlrev_b32", k_x_lshl}, {"v_lshrrev_b32", k_x_lshr}, {"v_or_b32", k_x_or}, {"v_not_b32", k_x_not}, {"v_ffbl_b32", k_x_ffbl}, {"v_bfe_u32", k_x_bfe}, {"v_alignbit_b32", k_x_alignbit}, {"v_mul_lo_u32", k_x_mullo}, {"v_mul_u32_u24", k_x_mul24}, {"v_mad_u32_u24", k_x_mad24}, {"v_or3_b32", k_x_or3}, {"v_lshl_or_b32", k_x_lshlor}, {"v_bcnt_u32_b32", k_x_bcnt}, {"v_min_i32", k_x_mini32}, {"v_min_u32", k_x_minu32}, {"v_cmp_lt_u32 + v_cndmask_b32 (pair)", k_x_cmpcnd}, {"v_cmp_eq_u32_e64 + v_cndmask_b32_e64 (pair, SGPR mask)", k_x_cmpe64}, {"v_add_u32_sdwa", k_x_sdwaadd}, {"v_xad_u32", k_x_xad}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto& kk : ks) {
        hipLaunchKernelGGL(kk.k, dim3(blocks), dim3(256), 0, 0, d, 1u);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kk.k, dim3(blocks), dim3(256), 0, 0, d, 2u + rep);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double wave_instr = (double)blocks * 4 * ITERS * 8;
        const double lane_ops = wave_instr * 64;
        const double per_simd = wave_instr / (cus * 4);
        printf("%-26s %8.3f ms  %7.2f T lane-ops/s  %5.2f cycles/wave-instr/SIMD @2.4GHz\n", kk.name, best,
               lane_ops / (best * 1e-3) / 1e12, best * 1e-3 * 2.4e9 / per_simd);
    }
    return 0;
}
