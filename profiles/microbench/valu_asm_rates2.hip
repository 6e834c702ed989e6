// Issue rates of the VALU instructions the emit kernels choose between (round 2): SDWA forms against
// their plain replacements, compares writing VCC / an SGPR pair, bit-field and byte-placing ops.
// Four independent chains per lane, 8 waves per SIMD, every CU busy; inline asm so that the compiler
// cannot fuse or reselect.  Build: hipcc --offload-arch=gfx950 -O3 valu_asm_rates2.hip -o valu_asm_rates2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 1024
#define REP8(x) x x x x x x x x
#define FOUR(ins) asm volatile(ins(%0, %1) "\n" ins(%1, %2) "\n" ins(%2, %3) "\n" ins(%3, %0) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(k2) : "vcc");
// each macro: dst/src0 = x, src1 = y, %4 = m (constant), %5 = k2
#define I_ADD(x, y) "v_add_u32 " #x ", " #x ", " #y
#define I_MOV_SDWA(x, y) "v_mov_b32_sdwa " #x ", " #y " dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2"
#define I_CND_SDWA(x, y) "v_cndmask_b32_sdwa " #x ", " #y ", " #y ", vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0"
#define I_CND(x, y) "v_cndmask_b32 " #x ", " #x ", " #y ", vcc"
#define I_LSHL_SDWA(x, y) "v_lshlrev_b32_sdwa " #x ", %5, " #y " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"
#define I_AND_SDWA(x, y) "v_and_b32_sdwa " #x ", " #y ", %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
#define I_CMP_SDWA(x, y) "v_cmp_lt_u32_sdwa vcc, " #x ", " #y " src0_sel:WORD_0 src1_sel:DWORD\n v_add_u32 " #x ", " #x ", %4"
#define I_CMP(x, y) "v_cmp_lt_u32 vcc, " #x ", " #y "\n v_add_u32 " #x ", " #x ", %4"
#define I_CMP_CND(x, y) "v_cmp_lt_u32 vcc, " #x ", " #y "\n s_nop 1\n v_cndmask_b32 " #x ", " #x ", " #y ", vcc"
#define I_CMP64_CND64(x, y) "v_cmp_lt_u32_e64 s[10:11], " #x ", " #y "\n s_nop 1\n v_cndmask_b32_e64 " #x ", " #x ", " #y ", s[10:11]"
#define I_BFE(x, y) "v_bfe_u32 " #x ", " #y ", 3, 10"
#define I_BFI(x, y) "v_bfi_b32 " #x ", %4, " #x ", " #y
#define I_ANDOR(x, y) "v_and_or_b32 " #x ", " #x ", %4, " #y
#define I_PERM(x, y) "v_perm_b32 " #x ", " #x ", " #y ", %5"
#define I_LSHLADD(x, y) "v_lshl_add_u32 " #x ", " #x ", 3, " #y
#define I_AND(x, y) "v_and_b32 " #x ", " #x ", " #y
#define I_OR3(x, y) "v_or3_b32 " #x ", " #x ", " #y ", %4"
#define I_SAD(x, y) "v_sad_u8 " #x ", " #x ", " #y ", %4"
#define I_BCNT(x, y) "v_bcnt_u32_b32 " #x ", " #x ", " #y
#define I_MUL24(x, y) "v_mul_u32_u24 " #x ", " #x ", " #y
#define I_MAD24(x, y) "v_mad_u32_u24 " #x ", " #x ", " #y ", %4"
#define I_MADHI24(x, y) "v_mul_hi_u32_u24 " #x ", " #x ", " #y
#define I_ADD_DPP(x, y) "v_add_u32_dpp " #x ", " #x ", " #y " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define I_LSHR(x, y) "v_lshrrev_b32 " #x ", 3, " #y
#define I_ALIGNBYTE(x, y) "v_alignbyte_b32 " #x ", " #x ", " #y ", 1"
#define I_DOT4(x, y) "v_dot4_u32_u8 " #x ", " #x ", " #y ", %4"
#define I_SUBREV_SDWA(x, y) "v_sub_u32_sdwa " #x ", " #x ", " #y " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD"
#define I_CVT_PK(x, y) "v_cvt_pk_u8_f32 " #x ", " #y ", 1, " #x
#define I_PKADD16(x, y) "v_pk_add_u16 " #x ", " #x ", " #y
#define I_PKMUL16(x, y) "v_pk_mul_lo_u16 " #x ", " #x ", " #y
#define I_MADU64(x, y) "v_mad_u64_u32 v[20:21], vcc, " #x ", " #y ", 0\n"
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x9e3779b9u, d = b + 7;
  const uint32_t m = 0x1ff8u, k2 = 0x06010403u;
  for (int i = 0; i < ITERS; i++) {
#define CASE(n, ins) if (OP == n) { REP8(FOUR(ins)) }
    CASE(0, I_ADD) CASE(1, I_MOV_SDWA) CASE(2, I_CND_SDWA) CASE(3, I_CND) CASE(4, I_LSHL_SDWA) CASE(5, I_AND_SDWA)
    CASE(6, I_CMP_SDWA) CASE(7, I_CMP) CASE(8, I_CMP_CND) CASE(9, I_CMP64_CND64) CASE(10, I_BFE) CASE(11, I_BFI)
    CASE(12, I_ANDOR) CASE(13, I_PERM) CASE(14, I_LSHLADD) CASE(15, I_AND) CASE(16, I_OR3) CASE(17, I_SAD)
    CASE(18, I_BCNT) CASE(19, I_MUL24) CASE(20, I_MAD24) CASE(21, I_MADHI24) CASE(22, I_ADD_DPP) CASE(23, I_LSHR)
    CASE(24, I_ALIGNBYTE) CASE(25, I_DOT4) CASE(26, I_SUBREV_SDWA) CASE(27, I_PKADD16) CASE(28, I_PKMUL16)
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}
template <int OP> double run(uint32_t* d_out, int grid, int per) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<OP><<<grid, 256>>>(d_out, 1); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<OP><<<grid, 256>>>(d_out, 2); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return (double)grid * 4 /*waves*/ * ITERS * 8 * 4 * per / (ms * 1e-3);
}
int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  int grid = p.multiProcessorCount * 8;
  uint32_t* d; (void)hipMalloc(&d, (size_t)grid * 256 * 4);
  const char* names[] = {"v_add_u32", "v_mov_b32_sdwa (byte insert)", "v_cndmask_b32_sdwa", "v_cndmask_b32 (vcc)", "v_lshlrev_b32_sdwa",
    "v_and_b32_sdwa", "v_cmp_lt_u32_sdwa + v_add", "v_cmp_lt_u32 + v_add", "v_cmp + s_nop 1 + v_cndmask", "v_cmp_e64 + s_nop 1 + v_cndmask_e64",
    "v_bfe_u32", "v_bfi_b32", "v_and_or_b32", "v_perm_b32", "v_lshl_add_u32", "v_and_b32", "v_or3_b32", "v_sad_u8", "v_bcnt_u32_b32",
    "v_mul_u32_u24", "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_add_u32_dpp quad_perm", "v_lshrrev_b32", "v_alignbyte_b32", "v_dot4_u32_u8",
    "v_sub_u32_sdwa", "v_pk_add_u16", "v_pk_mul_lo_u16"};
  const int per[] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
  double r[29];
#define RUN(n) r[n] = run<n>(d, grid, per[n]);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16)
  RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23) RUN(24) RUN(25) RUN(26) RUN(27) RUN(28)
  for (int i = 0; i < 29; i++)
    printf("%-38s %.3e wave-instr/s  = %.2f x the time of v_add_u32 per instruction\n", names[i], r[i], r[0] / r[i]);
  return 0;
}
