// Issue rate of single VALU instructions on gfx950, written as inline asm so the compiler cannot
// fuse or reselect them.  Four independent chains per lane, 8 waves per SIMD, every CU busy.
// Build: hipcc --offload-arch=gfx950 -O3 valu_asm_rates.hip -o valu_asm_rates ; run: ./valu_asm_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 2048
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x9e3779b9u, d = b + 7;
  uint64_t p = a, q = b, r = c, s = d;
  const uint32_t m = 0xD2511F53u;
  for (int i = 0; i < ITERS; i++) {
    if (OP == 0) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
    if (OP == 1) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %4, %6, 0\n v_mad_u64_u32 %2, vcc, %4, %7, 0\n v_mad_u64_u32 %3, vcc, %4, %8, 0" : "+v"(p), "+v"(q), "+v"(r), "+v"(s) : "v"(m), "v"(a), "v"(b), "v"(c), "v"(d) : "vcc"); a ^= (uint32_t)p; ) }
    if (OP == 2) { REP8(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
    if (OP == 3) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
    if (OP == 4) { REP8(asm volatile("v_bitop3_b32 %0, %0, %1, %4 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %4 bitop3:0x96\n v_bitop3_b32 %2, %2, %3, %4 bitop3:0x96\n v_bitop3_b32 %3, %3, %0, %4 bitop3:0x96" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
    if (OP == 5) { REP8(asm volatile("v_perm_b32 %0, %0, %1, %4\n v_perm_b32 %1, %1, %2, %4\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %3, %3, %0, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(0x06010403u));) }
    if (OP == 6) { REP8(asm volatile("v_alignbit_b32 %0, %0, %1, 2\n v_alignbit_b32 %1, %1, %2, 2\n v_alignbit_b32 %2, %2, %3, 2\n v_alignbit_b32 %3, %3, %0, 2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 7) { REP8(asm volatile("v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n v_cndmask_b32_sdwa %1, %1, %2, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n v_cndmask_b32_sdwa %2, %2, %3, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n v_cndmask_b32_sdwa %3, %3, %0, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");) }
    if (OP == 8) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_cmp_lt_u32 vcc, %1, %3\n v_addc_co_u32 %3, vcc, %3, %3, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");) }
    if (OP == 9) { REP8(asm volatile("v_lshl_or_b32 %0, %0, 3, %1\n v_lshl_or_b32 %1, %1, 3, %2\n v_lshl_or_b32 %2, %2, 3, %3\n v_lshl_or_b32 %3, %3, 3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (OP == 10) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0" : "+v"(p), "+v"(q), "+v"(r), "+v"(s));) }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (uint32_t)(p + q + r + s) + (uint32_t)((p + q + r + s) >> 32);
}
template <int OP> double run(uint32_t* d_out, int grid) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<OP><<<grid, 256>>>(d_out, 1); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<OP><<<grid, 256>>>(d_out, 2); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return (double)grid * 4 /*waves*/ * ITERS * 8 * 4 / (ms * 1e-3);
}
int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  int grid = p.multiProcessorCount * 8;
  uint32_t* d; (void)hipMalloc(&d, (size_t)grid * 256 * 4);
  const char* names[] = {"v_add_u32", "v_mad_u64_u32", "v_mul_hi_u32", "v_mul_lo_u32", "v_bitop3_b32", "v_perm_b32", "v_alignbit_b32",
                         "v_cndmask_b32_sdwa", "v_cmp + v_addc_co", "v_lshl_or_b32", "v_lshl_add_u64"};
  double r[11] = {run<0>(d, grid), run<1>(d, grid), run<2>(d, grid), run<3>(d, grid), run<4>(d, grid), run<5>(d, grid),
                  run<6>(d, grid), run<7>(d, grid), run<8>(d, grid), run<9>(d, grid), run<10>(d, grid)};
  for (int i = 0; i < 11; i++)
    printf("%-22s %.3e wave-instr/s  = %.2f x the time of v_add_u32\n", names[i], r[i], r[0] / r[i]);
  return 0;
}
