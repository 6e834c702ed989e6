// Issue-rate microbenchmark for the integer ops the RNG choices rest on (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 int_rates.hip -o int_rates ; run: ./int_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 4096
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x9e3779b9u, d = b + 7;
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (OP == 0) { a += b; b ^= c; c += d; d ^= a; }                                     // add / xor
      if (OP == 1) { a = __builtin_rotateleft32(a ^ b, 7); b = __builtin_rotateleft32(b ^ c, 9); c = __builtin_rotateleft32(c ^ d, 13); d = __builtin_rotateleft32(d ^ a, 18); }  // xor + alignbit
      if (OP == 2) { a = a * b + 1; b = b * c + 3; c = c * d + 5; d = d * a + 7; }  // v_mul_lo_u32 (+add)
      if (OP == 3) { a = __umulhi(a, 0xD2511F53u); b = __umulhi(b, 0xCD9E8D57u); c = __umulhi(c, 0x9E3779B9u) | 1; d = __umulhi(d, 0xBB67AE85u) | 3; }  // v_mul_hi_u32
      if (OP == 4) { a = __umul24(a, b) + 1; b = __umul24(b, c) + 3; c = __umul24(c, d) + 5; d = __umul24(d, a) + 7; }  // v_mul_u32_u24 (+add)
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}
template <int OP> double run(uint32_t* d_out, int grid, int ops_per_inner) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<OP><<<grid, 256>>>(d_out, 1); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<OP><<<grid, 256>>>(d_out, 2); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double wave_instr = (double)grid * 4 /*waves*/ * ITERS * 8 * ops_per_inner;
  return wave_instr / (ms * 1e-3);
}
int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  int grid = p.multiProcessorCount * 8;  // 8 waves per SIMD
  uint32_t* d; (void)hipMalloc(&d, (size_t)grid * 256 * 4);
  const char* names[] = {"add/xor", "xor+rotate(alignbit)", "mul_lo_u32", "mul_hi_u32", "mul_u32_u24+add"};
  double r[5] = {run<0>(d, grid, 4), run<1>(d, grid, 8), run<2>(d, grid, 8), run<3>(d, grid, 6), run<4>(d, grid, 8)};
  double simds = p.multiProcessorCount * 4.0;
  for (int i = 0; i < 5; i++)
    printf("%-24s %.3e wave-instr/s  = %.2f cycles per wave64 instruction per SIMD at %.2f GHz (nominal)\n", names[i], r[i],
           simds * p.clockRate * 1e3 / r[i], p.clockRate * 1e-6);
  return 0;
}
