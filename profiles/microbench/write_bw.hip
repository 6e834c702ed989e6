// HBM write-path ceilings on MI355X for the store patterns the emit kernels use.
// Build: hipcc --offload-arch=gfx950 -O3 write_bw.hip -o write_bw ; run: ./write_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint4 __attribute__((aligned(1))) uint4_u;
// V0: one aligned 16-byte store per thread, whole grid contiguous (memset-like)
__global__ void __launch_bounds__(256) fill_linear(uint4* a, uint64_t n16) {
  const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) a[i] = v;
}
// V1/V2: workgroup b owns chunk b, b + grid, ... of `chunk` bytes in each of `streams` arrays; lanes store
// consecutive 16-byte pieces (aligned when chunk % 16 == 0, byte-misaligned otherwise)
__global__ void __launch_bounds__(256) fill_chunks(uint8_t* a, uint8_t* b, uint64_t n_chunks, uint32_t chunk, int streams) {
  const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
  for (uint64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const uint64_t base = c * chunk;
    for (uint32_t o = threadIdx.x * 16; o + 16 <= chunk; o += 256 * 16) {
      *reinterpret_cast<uint4_u*>(a + base + o) = v;
      if (streams > 1) *reinterpret_cast<uint4_u*>(b + base + o) = v;
    }
  }
}
static float timeit(void (*f)(void*), void* ctx) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(ctx); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(ctx); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
struct Ctx { uint8_t *a, *b; uint64_t bytes; int grid; uint32_t chunk; int streams; };
int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  Ctx c; c.bytes = 15ull << 30; c.grid = p.multiProcessorCount * 8;
  (void)hipMalloc(&c.a, c.bytes + 4096); (void)hipMalloc(&c.b, c.bytes + 4096);
  float ms = timeit([](void* q) { Ctx* c = (Ctx*)q; fill_linear<<<c->grid, 256>>>((uint4*)c->a, c->bytes / 16); }, &c);
  printf("linear aligned fill, 1 stream          : %7.2f ms  %6.2f TB/s\n", ms, c.bytes / ms * 1e-9);
  for (int streams = 1; streams <= 2; streams++)
    for (uint32_t chunk : {38400u, 38390u, 4096u, 4090u}) {
      c.chunk = chunk; c.streams = streams;
      ms = timeit([](void* q) { Ctx* c = (Ctx*)q; fill_chunks<<<c->grid, 256>>>(c->a, c->b, c->bytes / c->chunk, c->chunk, c->streams); }, &c);
      const double wr = (double)(c.bytes / chunk) * (chunk / 16 * 16) * streams;
      printf("chunked fill, chunk %5u B, %d stream(s)  : %7.2f ms  %6.2f TB/s\n", chunk, streams, ms, wr / ms * 1e-9);
    }
  return 0;
}
