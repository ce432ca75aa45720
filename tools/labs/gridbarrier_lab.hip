// LAB: what does a grid-wide barrier cost on MI355X?  (budget for the persistent multi-phase step kernel of DESIGN.md
// section 8: it would replace 4 kernel boundaries per step and layer.)
// A resident grid (WGS workgroups per CU x 256 CUs, 256 threads each) passes NB barriers built from one global atomic
// counter (sense reversal by generation).  Every spin is BOUNDED: a workgroup that waits longer than SPIN_MAX polls sets an
// error flag and leaves, so the kernel terminates even if the grid is not resident.
//   hipcc -O3 --offload-arch=gfx950 tools/gridbarrier_lab.hip -o tools/gridbarrier_lab && tools/gridbarrier_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr long SPIN_MAX = 2000000;

__global__ __launch_bounds__(256) void k_barriers(unsigned* counter, unsigned* gen, int* err, int nb, float* sink, int work) {
  const unsigned total = gridDim.x;
  float acc = threadIdx.x;
  for (int b = 0; b < nb; ++b) {
    for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;     // optional filler between barriers
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned g = __atomic_load_n(gen, __ATOMIC_ACQUIRE);
      __threadfence();
      const unsigned arrived = atomicAdd(counter, 1u) + 1u;
      if (arrived == total) {
        atomicExch(counter, 0u);
        __threadfence();
        atomicAdd(gen, 1u);
      } else {
        long spins = 0;
        while (__atomic_load_n(gen, __ATOMIC_ACQUIRE) == g) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > SPIN_MAX) { atomicExch(err, 1); break; }
        }
      }
    }
    __syncthreads();
    if (__atomic_load_n(err, __ATOMIC_RELAXED)) break;      // somebody gave up: everybody leaves
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  unsigned *counter, *gen; int* err; float* sink;
  CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&gen, 4)); CHECK(hipMalloc(&err, 4)); CHECK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int wgs = 1; wgs <= 4; wgs *= 2) {
    const int grid = 256 * wgs;
    int maxBlocks = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxBlocks, k_barriers, 256, 0));
    if (maxBlocks < wgs) { printf("grid %d: only %d blocks per CU resident, skipped\n", grid, maxBlocks); continue; }
    for (int nb : {1, 101, 401}) {
      CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(gen, 0, 4)); CHECK(hipMemset(err, 0, 4));
      hipLaunchKernelGGL(k_barriers, dim3(grid), dim3(256), 0, 0, counter, gen, err, nb, sink, 0);   // warm
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(gen, 0, 4));
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_barriers, dim3(grid), dim3(256), 0, 0, counter, gen, err, nb, sink, 0);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
      int h = 0; CHECK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
      printf("grid %4d workgroups (%d / CU), %3d barriers: %8.1f us total%s\n", grid, wgs, nb, ms * 1e3, h ? "  [a workgroup gave up: grid not resident]" : "");
    }
  }
  // the alternative: the same number of boundaries as empty dependent kernel launches
  for (int n : {1, 101, 401}) {
    CHECK(hipMemset(err, 0, 4));
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_barriers, dim3(512), dim3(256), 0, 0, counter, gen, err, 0, sink, 0);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%3d empty dependent launches of 512 workgroups: %8.1f us total\n", n, ms * 1e3);
  }
  return 0;
}
