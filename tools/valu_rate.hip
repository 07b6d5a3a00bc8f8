// valu_rate.hip — issue rate of the vector instructions the gradient-stage kernel is built from, on gfx950.
// For every instruction: 8 independent dependency chains per wave, W waves per SIMD (W = 1, 2, 4, 8), every CU
// busy; reports cycles per wave-instruction per SIMD (from wall time at the clock read with s_memtime /
// s_memrealtime inside the kernel).  Design input only; nothing in the library depends on it.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_rate tools/valu_rate.hip && tools/bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHAINS 8
#define REP 16 // instructions per chain per loop iteration

#define BODY8(INS)                                                                                                     \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                               \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])       \
                 : "v"(b), "v"(c))

#define KERNEL(NAME, INS)                                                                                              \
    __global__ __launch_bounds__(512) void NAME(uint32_t* out, int iters, unsigned long long* clk)                     \
    {                                                                                                                  \
        uint32_t a[CHAINS];                                                                                            \
        for (int i = 0; i < CHAINS; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;                   \
        uint32_t b = threadIdx.x * 97u + 13u, c = blockIdx.x * 31u + 7u;                                               \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                   \
        for (int it = 0; it < iters; ++it) {                                                                           \
            _Pragma("unroll") for (int r = 0; r < REP; ++r) { BODY8(INS); }                                            \
        }                                                                                                              \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                   \
        uint32_t s = 0;                                                                                                \
        for (int i = 0; i < CHAINS; ++i) s ^= a[i];                                                                    \
        if (s == 0x12345678u) out[0] = s;                                                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                                                     \
            clk[0] = t1 - t0;                                                                                          \
            clk[1] = r1 - r0;                                                                                          \
        }                                                                                                              \
    }

#define I_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_DOT2U(n) "v_dot2_u32_u16 %" #n ", %" #n ", %8, %9\n"
#define I_DOT2I(n) "v_dot2_i32_i16 %" #n ", %" #n ", %8, %9\n"
#define I_DOT4U(n) "v_dot4_u32_u8 %" #n ", %" #n ", %8, %9\n"
#define I_PKADD(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n"
#define I_PKSUB(n) "v_pk_sub_i16 %" #n ", %" #n ", %8\n"
#define I_PKMAD(n) "v_pk_mad_u16 %" #n ", %" #n ", %8, %9\n"
#define I_PKMUL(n) "v_pk_mul_lo_u16 %" #n ", %" #n ", %8\n"
#define I_PKMAX(n) "v_pk_max_i16 %" #n ", %" #n ", %8\n"
#define I_PKLSHR(n) "v_pk_lshrrev_b16 %" #n ", 8, %" #n "\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define I_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 16\n"
#define I_ALIGNBYTE(n) "v_alignbyte_b32 %" #n ", %" #n ", %8, 1\n"
#define I_MOVDPP_WSHR(n) "v_mov_b32_dpp %" #n ", %" #n " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_MOVDPP_WSHL(n) "v_mov_b32_dpp %" #n ", %" #n " wave_shl:1 row_mask:0xf bank_mask:0xf\n"
#define I_MOVDPP_RSHR(n) "v_mov_b32_dpp %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADDDPP_WSHR(n) "v_add_u32_dpp %" #n ", %" #n ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define I_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 8, %9\n"
#define I_BCNT(n) "v_bcnt_u32_b32 %" #n ", %" #n ", %8\n"
#define I_FFBL(n) "v_ffbl_b32 %" #n ", %" #n "\n"
// NOTE: selects on VCC, which the probe loop's own control flow rewrites between the instructions: this line measures that
// dependency (22.5 cycles), not the instruction; I_CNDS (mask in an SGPR pair set once) is the issue cost: 4.06 cycles.
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CMP(n) "v_cmp_gt_i32 vcc, %" #n ", %8\n"
#define I_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define I_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define I_MAX3(n) "v_max3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 8, 8\n"
#define I_LSHL(n) "v_lshlrev_b32 %" #n ", %8, %" #n "\n"
#define I_MADU16(n) "v_mad_u16 %" #n ", %" #n ", %8, %9\n"
#define I_SAD(n) "v_sad_u8 %" #n ", %" #n ", %8, %9\n"
#define I_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I_SUBREV(n) "v_sub_u32 %" #n ", %8, %" #n "\n"

#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_OR(n) "v_or_b32 %" #n ", %" #n ", %8\n"
#define I_XOR2(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I_SUB(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_MAXI(n) "v_max_i32 %" #n ", %" #n ", %8\n"
#define I_MINU(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define I_LSHR(n) "v_lshrrev_b32 %" #n ", 3, %" #n "\n"
#define I_ASHR(n) "v_ashrrev_i32 %" #n ", 3, %" #n "\n"
#define I_ADDF(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define I_MULF(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_FMAF(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_FMACF(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define I_MACU24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define I_ADDLSHL(n) "v_add_lshl_u32 %" #n ", %" #n ", %8, 2\n"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %8\n"
#define I_OR3(n) "v_or3_b32 %" #n ", %" #n ", %8, %9\n"
#define I_XAD(n) "v_xad_u32 %" #n ", %" #n ", %8, %9\n"
#define I_MED3(n) "v_med3_i32 %" #n ", %" #n ", %8, %9\n"
#define I_CVTU8(n) "v_cvt_f32_ubyte0 %" #n ", %" #n "\n"
#define I_CVTI(n) "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define I_CNDS(n) "v_cndmask_b32 %" #n ", %" #n ", %8, s[20:21]\n"
#define I_ADDSDWA(n) "v_add_u32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n"
#define I_PKADDF16(n) "v_pk_add_f16 %" #n ", %" #n ", %8\n"
#define I_MADI24(n) "v_mad_i32_i24 %" #n ", %" #n ", %8, %9\n"
#define I_SUBREV2(n) "v_subrev_u32 %" #n ", %" #n ", %8\n"
#define I_ADDCO(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %8\n"
#define I_MADU32U16(n) "v_mad_u32_u16 %" #n ", %" #n ", %8, %9\n"
KERNEL(k_add, I_ADD)
KERNEL(k_dot2u, I_DOT2U)
KERNEL(k_dot2i, I_DOT2I)
KERNEL(k_dot4u, I_DOT4U)
KERNEL(k_pkadd, I_PKADD)
KERNEL(k_pksub, I_PKSUB)
KERNEL(k_pkmad, I_PKMAD)
KERNEL(k_pkmul, I_PKMUL)
KERNEL(k_pkmax, I_PKMAX)
KERNEL(k_pklshr, I_PKLSHR)
KERNEL(k_perm, I_PERM)
KERNEL(k_alignbit, I_ALIGNBIT)
KERNEL(k_alignbyte, I_ALIGNBYTE)
KERNEL(k_movdpp_wshr, I_MOVDPP_WSHR)
KERNEL(k_movdpp_wshl, I_MOVDPP_WSHL)
KERNEL(k_movdpp_rshr, I_MOVDPP_RSHR)
KERNEL(k_adddpp_wshr, I_ADDDPP_WSHR)
KERNEL(k_andor, I_ANDOR)
KERNEL(k_add3, I_ADD3)
KERNEL(k_lshlor, I_LSHLOR)
KERNEL(k_bcnt, I_BCNT)
KERNEL(k_ffbl, I_FFBL)
KERNEL(k_cndmask, I_CNDMASK)
KERNEL(k_cmp, I_CMP)
KERNEL(k_mad24, I_MAD24)
KERNEL(k_mullo, I_MULLO)
KERNEL(k_max3, I_MAX3)
KERNEL(k_bfe, I_BFE)
KERNEL(k_lshl, I_LSHL)
KERNEL(k_madu16, I_MADU16)
KERNEL(k_sad, I_SAD)

KERNEL(k_and, I_AND)
KERNEL(k_or, I_OR)
KERNEL(k_xor2, I_XOR2)
KERNEL(k_sub, I_SUB)
KERNEL(k_mov, I_MOV)
KERNEL(k_maxi, I_MAXI)
KERNEL(k_minu, I_MINU)
KERNEL(k_lshr, I_LSHR)
KERNEL(k_ashr, I_ASHR)
KERNEL(k_addf, I_ADDF)
KERNEL(k_mulf, I_MULF)
KERNEL(k_fmaf, I_FMAF)
KERNEL(k_fmacf, I_FMACF)
KERNEL(k_mulu24, I_MACU24)
KERNEL(k_addlshl, I_ADDLSHL)
KERNEL(k_lshladd, I_LSHLADD)
KERNEL(k_or3, I_OR3)
KERNEL(k_xad, I_XAD)
KERNEL(k_med3, I_MED3)
KERNEL(k_cvtu8, I_CVTU8)
KERNEL(k_cvti, I_CVTI)
KERNEL(k_cnds, I_CNDS)
KERNEL(k_addsdwa, I_ADDSDWA)
KERNEL(k_pkaddf16, I_PKADDF16)
KERNEL(k_madi24, I_MADI24)
KERNEL(k_subrev2, I_SUBREV2)
KERNEL(k_addco, I_ADDCO)
KERNEL(k_madu32u16, I_MADU32U16)

struct Entry {
    const char* name;
    void (*fn)(uint32_t*, int, unsigned long long*);
};

int main()
{
    Entry tab[] = {{"v_add_u32", k_add}, {"v_dot2_u32_u16", k_dot2u}, {"v_dot2_i32_i16", k_dot2i}, {"v_dot4_u32_u8", k_dot4u},
                   {"v_pk_add_u16", k_pkadd}, {"v_pk_sub_i16", k_pksub}, {"v_pk_mad_u16", k_pkmad}, {"v_pk_mul_lo_u16", k_pkmul},
                   {"v_pk_max_i16", k_pkmax}, {"v_pk_lshrrev_b16", k_pklshr}, {"v_perm_b32", k_perm}, {"v_alignbit_b32", k_alignbit},
                   {"v_alignbyte_b32", k_alignbyte}, {"v_mov_dpp wave_shr", k_movdpp_wshr}, {"v_mov_dpp wave_shl", k_movdpp_wshl},
                   {"v_mov_dpp row_shr", k_movdpp_rshr}, {"v_add_dpp wave_shr", k_adddpp_wshr}, {"v_and_or_b32", k_andor},
                   {"v_add3_u32", k_add3}, {"v_lshl_or_b32", k_lshlor}, {"v_bcnt_u32_b32", k_bcnt}, {"v_ffbl_b32", k_ffbl},
                   {"v_cndmask_b32", k_cndmask}, {"v_cmp_gt_i32", k_cmp}, {"v_mad_u32_u24", k_mad24}, {"v_mul_lo_u32", k_mullo},
                   {"v_max3_u32", k_max3}, {"v_bfe_u32", k_bfe}, {"v_lshlrev_b32", k_lshl}, {"v_mad_u16", k_madu16}, {"v_sad_u8", k_sad}, {"v_and_b32", k_and}, {"v_or_b32", k_or}, {"v_xor_b32", k_xor2}, {"v_sub_u32", k_sub}, {"v_mov_b32", k_mov}, {"v_max_i32", k_maxi}, {"v_min_u32", k_minu}, {"v_lshrrev_b32", k_lshr}, {"v_ashrrev_i32", k_ashr}, {"v_add_f32", k_addf}, {"v_mul_f32", k_mulf}, {"v_fma_f32", k_fmaf}, {"v_fmac_f32", k_fmacf}, {"v_mul_u32_u24", k_mulu24}, {"v_add_lshl_u32", k_addlshl}, {"v_lshl_add_u32", k_lshladd}, {"v_or3_b32", k_or3}, {"v_xad_u32", k_xad}, {"v_med3_i32", k_med3}, {"v_cvt_f32_ubyte0", k_cvtu8}, {"v_cvt_f32_i32", k_cvti}, {"v_cndmask_b32 sgpr", k_cnds}, {"v_add_u32_sdwa", k_addsdwa}, {"v_pk_add_f16", k_pkaddf16}, {"v_mad_i32_i24", k_madi24}, {"v_subrev_u32", k_subrev2}, {"v_add_co_u32", k_addco}, {"v_mad_u32_u16", k_madu32u16}};
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t* d_out;
    unsigned long long* d_clk;
    hipMalloc(&d_out, 64);
    hipMalloc(&d_clk, 64);
    const int iters = 2000;
    printf("device %s, %d CUs; cycles per wave-instruction per SIMD (lower = faster; 2.0 = one wave64 op per 2 cycles)\n", prop.name, cus);
    printf("%-22s %8s %8s %8s %8s\n", "instruction", "W=1", "W=2", "W=4", "W=8");
    for (auto& e : tab) {
        printf("%-22s", e.name);
        for (int W : {1, 2, 4, 8}) {
            // W waves per SIMD: blocks of 256 threads (one wave per SIMD), W blocks per CU
            const int threads = 256, blocks = cus * W;
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(threads), 0, 0, d_out, 50, d_clk);
            hipDeviceSynchronize();
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(threads), 0, 0, d_out, iters, d_clk);
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            unsigned long long clk[2];
            hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost);
            // wave 0's own view: cycles it needed / instructions it issued, divided by waves sharing the SIMD
            // whole-launch view: wall time x in-kernel clock (s_memtime ticks per 10 ns of s_memrealtime) over the
            // instructions one SIMD issued (W waves x iters x REP x CHAINS)
            const double n_instr = (double)iters * REP * CHAINS;
            const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);
            const double cyc_wall = (double)ms * 1e6 * ghz;
            printf(" %8.2f", cyc_wall / (n_instr * W));
            if (W == 8) printf("   [wave 0 alone: %.2f cyc/instr at W=8, clock %.2f GHz]", (double)clk[0] / n_instr, ghz);
            hipEventDestroy(a);
            hipEventDestroy(b);
        }
        printf("\n");
    }
    hipFree(d_out);
    hipFree(d_clk);
    return 0;
}
