/*
 * rocrand_pin.hip — ORACLE (test infrastructure only; see oracle.h).
 *
 * BASELINE.json's north_star names "counter-based rocRAND (Philox)" for the per-base draws.  The product's
 * generator is hand-written (simmr_amd/csrc/kernels.hip: philox4x32_10, two v_mad_u64_u32 and two v_bitop3_b32
 * per round) and specified in include/simmr_hip.h / oracle/philox.c; this file ties that specification to
 * rocRAND itself: it runs rocRAND's OWN engine class — rocrand_device::philox4x32_10_engine of the ROCm
 * installation's <rocrand/rocrand_philox4x32_10.h>, whose members are __host__ __device__ — on the host, so that
 * tests/test_oracle_kat.py can check, without a GPU, that
 *
 *     word j of the block with key (k0, k1) and counter (c0, c1, c2, c3)
 *       ==  the j-th `next()` of  philox4x32_10_engine(seed = k0 | k1 << 32,
 *                                                     subsequence = c2 | c3 << 32,
 *                                                     offset = 4 * (c0 | c1 << 32))
 *
 * i.e. a read's draws are the rocRAND Philox4x32-10 stream of seed = the read's Phred seed, subsequence =
 * 0x7200000373696D6D ('simm', 'r\0\0\3'), read at offset 4 * (3 g + c) for level 1 and 4 * ((b >> 2) | 1 << 32)
 * for level 2.  rocRAND is a third-party library of the image (ROCm 7.2), not the reference; nothing of it is
 * copied: the header is included where it lies.
 */
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_philox4x32_10.h>

#include <stdint.h>

extern "C" {

/* n consecutive 32-bit outputs of rocRAND's Philox4x32-10 engine */
void rr_philox_stream(uint64_t seed, uint64_t subsequence, uint64_t offset, uint32_t n, uint32_t* out) {
  rocrand_device::philox4x32_10_engine eng(seed, subsequence, offset);
  for (uint32_t i = 0; i < n; i++) out[i] = eng.next();
}

/* the same through the C-style device API (rocrand_init / rocrand4), which is what a kernel written against
 * rocRAND would call */
void rr_philox_block(uint64_t seed, uint64_t subsequence, uint64_t offset, uint32_t out[4]) {
  rocrand_state_philox4x32_10 st;
  rocrand_init(seed, subsequence, offset, &st);
  const uint4 v = rocrand4(&st);
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}

}  // extern "C"
