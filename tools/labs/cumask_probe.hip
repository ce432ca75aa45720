// LAB: how do the bits of a HIP stream's CU mask (hipExtStreamCreateWithCUMask) map onto the 8 XCDs of an MI355X, and
// can two streams with disjoint masks run kernels side by side?  (Round 3: the two layers' recurrent chains phase-lock
// when they share the chip - DESIGN.md section 4 - so giving each chain its own XCDs is worth a measurement.)
//   hipcc -O3 --offload-arch=gfx950 tools/labs/cumask_probe.hip -o tools/labs/cumask_probe && tools/labs/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_where(unsigned* out, int spin) {
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;   // HW_REG_XCC_ID[3:0]
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);          // HW_REG_HW_ID
  float acc = threadIdx.x;
  for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | ((hw >> 8) & 0xff) | (acc == 1.234f ? 1u << 31 : 0u);   // cu_id[3:0], sh_id, se_id
}

__global__ __launch_bounds__(256) void k_busy(float* sink, int iters) {
  float a = threadIdx.x, b = blockIdx.x;
  for (int i = 0; i < iters; ++i) { a = a * 1.0001f + b; b = b * 0.9999f + a; }
  if (a == 1.2345f) sink[0] = b;
}

static int histogram(hipStream_t s, unsigned* dev, const char* tag) {
  const int grid = 4096;
  std::vector<unsigned> h(grid);
  hipLaunchKernelGGL(k_where, dim3(grid), dim3(256), 0, s, dev, 2000);
  CHECK(hipStreamSynchronize(s));
  CHECK(hipMemcpy(h.data(), dev, grid * 4, hipMemcpyDeviceToHost));
  int perX[16] = {0};
  bool seen[16][256];
  memset(seen, 0, sizeof(seen));
  for (unsigned v : h) { perX[(v >> 16) & 15]++; seen[(v >> 16) & 15][v & 0xff] = true; }
  printf("%-28s workgroups per XCC:", tag);
  for (int x = 0; x < 8; ++x) printf(" %4d", perX[x]);
  printf("   distinct CUs per XCC:");
  for (int x = 0; x < 8; ++x) { int c = 0; for (int i = 0; i < 256; ++i) c += seen[x][i]; printf(" %2d", c); }
  printf("\n");
  return 0;
}

int main() {
  unsigned* dev; float* sink;
  CHECK(hipMalloc(&dev, 4096 * 4)); CHECK(hipMalloc(&sink, 4));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  hipStream_t plain; CHECK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
  if (histogram(plain, dev, "no mask")) return 1;
  struct Pat { const char* tag; unsigned w[8]; };
  Pat pats[] = {
      {"bits 0-31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
      {"bits 0-127", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0}},
      {"bits 128-255", {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}},
      {"every 8th bit (0,8,..)", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}},
      {"bits with (i%8)<4", {0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu}},
      {"bits with (i%8)>=4", {0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u}},
      {"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
  };
  hipStream_t lowS = nullptr, highS = nullptr;
  for (auto& p : pats) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, p.w);
    if (e != hipSuccess) { printf("%-28s hipExtStreamCreateWithCUMask failed: %s\n", p.tag, hipGetErrorString(e)); continue; }
    if (histogram(s, dev, p.tag)) return 1;
    if (!strcmp(p.tag, "bits 0-127")) lowS = s;
    else if (!strcmp(p.tag, "bits 128-255")) highS = s;
  }
  // do two kernels on disjoint masks overlap?  a busy kernel of 1280 workgroups alone on the whole chip, alone on a
  // half, and two of them on the two halves at the same time
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto timeit = [&](hipStream_t a, hipStream_t b, const char* tag) -> int {
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0, plain));
      CHECK(hipStreamWaitEvent(a, e0, 0));
      if (b) CHECK(hipStreamWaitEvent(b, e0, 0));
      hipLaunchKernelGGL(k_busy, dim3(1280), dim3(256), 0, a, sink, 40000);
      if (b) hipLaunchKernelGGL(k_busy, dim3(1280), dim3(256), 0, b, sink, 40000);
      hipEvent_t ea, eb; CHECK(hipEventCreate(&ea)); CHECK(hipEventCreate(&eb));
      CHECK(hipEventRecord(ea, a)); CHECK(hipStreamWaitEvent(plain, ea, 0));
      if (b) { CHECK(hipEventRecord(eb, b)); CHECK(hipStreamWaitEvent(plain, eb, 0)); }
      CHECK(hipEventRecord(e1, plain));
      CHECK(hipDeviceSynchronize());
      float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("%-40s %8.1f us\n", tag, ms * 1e3);
    }
    return 0;
  };
  if (timeit(plain, nullptr, "one busy kernel, whole chip")) return 1;
  if (lowS && highS) {
    if (timeit(lowS, nullptr, "one busy kernel, half mask")) return 1;
    if (timeit(lowS, highS, "two busy kernels, disjoint half masks")) return 1;
    hipStream_t plain2; CHECK(hipStreamCreateWithFlags(&plain2, hipStreamNonBlocking));
    if (timeit(plain, plain2, "two busy kernels, two unmasked streams")) return 1;
  }
  return 0;
}
