// LAB (round 3): the XCD-hierarchical grid barrier of MI355X_MICROARCH.md (price-list row "barrier-xcd") against the
// single-counter barrier tools/gridbarrier_lab.hip measured in round 2 and against an empty dependent launch - the
// budget question of a persistent multi-phase step kernel.
// Per-XCC arrival counter -> the XCC's last arriver (its leader) runs the release fence and arrives at the top counter
// -> leaders poll the top counter, acquire, and publish their XCC's generation word -> everybody else polls ITS XCC's
// word and acquires.  Monotonic counters (no reset race), relaxed sc1 polls with s_sleep, every spin BOUNDED.
// Optional payload: every workgroup writes PAYLOAD_KB of plain stores before the barrier and reads the slab of the
// workgroup 37 places further on after it (checked: a stale read is counted).
//   hipcc -O3 --offload-arch=gfx950 tools/labs/xcdbarrier_lab.hip -o tools/labs/xcdbarrier_lab && tools/labs/xcdbarrier_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr long SPIN_MAX = 4000000;
constexpr int LINE = 32;   // unsigned words per 128-byte line

struct Bar {
  unsigned members[8 * LINE];   // workgroups resident per XCC (registered in the setup phase)
  unsigned setup[LINE];         // single-counter barrier of the setup phase
  unsigned xccCount[8 * LINE];
  unsigned top[LINE];
  unsigned xccGen[8 * LINE];
  unsigned err[LINE];
  unsigned stale[LINE];
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_relaxed(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool spin_until(const unsigned* p, unsigned want, unsigned* err) {
  long spins = 0;
  while (ld_relaxed(p) < want) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > SPIN_MAX) { atomicExch(err, 1u); return false; }
    if ((spins & 1023) == 0 && ld_relaxed(err) != 0u) return false;   // somebody gave up: everybody leaves
  }
  return true;
}

template <bool HIER>
__global__ __launch_bounds__(256) void k_barriers(Bar* bar, int nb, float* slabs, int payloadFloats, float* sink) {
  __shared__ unsigned sh[4];
  const unsigned total = gridDim.x;
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;
    atomicAdd(&bar->members[xcc * LINE], 1u);
    atomicAdd(&bar->setup[0], 1u);
    bool ok = spin_until(&bar->setup[0], total, &bar->err[0]);
    unsigned nx = 0;
    for (int x = 0; x < 8; ++x) nx += ld_relaxed(&bar->members[x * LINE]) > 0 ? 1u : 0u;
    sh[0] = xcc; sh[1] = ld_relaxed(&bar->members[xcc * LINE]); sh[2] = nx; sh[3] = ok ? 1u : 0u;
  }
  __syncthreads();
  const unsigned xcc = sh[0], mine = sh[1], nx = sh[2];
  bool alive = sh[3] != 0;
  float acc = 0.f;
  float* my = slabs + (size_t)blockIdx.x * payloadFloats;
  const float* other = slabs + (size_t)((blockIdx.x + 37) % total) * payloadFloats;
  for (int b = 0; b < nb && alive; ++b) {
    const unsigned g = (unsigned)b + 1u;
    for (int i = threadIdx.x * 4; i < payloadFloats; i += 1024)
      *reinterpret_cast<float4*>(my + i) = make_float4((float)g, (float)g, (float)g, (float)g);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      bool ok = true;
      if (HIER) {
        const unsigned old = atomicAdd(&bar->xccCount[xcc * LINE], 1u);
        if (old + 1u == mine * g) {                       // this XCC's last arriver: its leader for this generation
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          atomicAdd(&bar->top[0], 1u);
          ok = spin_until(&bar->top[0], nx * g, &bar->err[0]);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          st_relaxed(&bar->xccGen[xcc * LINE], g);
        } else {
          ok = spin_until(&bar->xccGen[xcc * LINE], g, &bar->err[0]);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
      } else {                                              // one monotonic counter for everybody
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        atomicAdd(&bar->top[0], 1u);
        ok = spin_until(&bar->top[0], total * g, &bar->err[0]);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      sh[3] = (ok && ld_relaxed(&bar->err[0]) == 0u) ? 1u : 0u;
    }
    __syncthreads();
    alive = sh[3] != 0;
    if (payloadFloats > 0 && alive) {
      float bad = 0.f;
      for (int i = threadIdx.x * 4; i < payloadFloats; i += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(other + i);
        // fresh = generation g, or g + 1 from a writer that is already a phase ahead; anything older is a stale read
        bad += (v.x < (float)g) + (v.y < (float)g) + (v.z < (float)g) + (v.w < (float)g);
      }
      if (bad != 0.f) atomicAdd(&bar->stale[0], 1u);
      acc += bad;
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  Bar* bar; float* sink; float* slabs;
  const int maxGrid = 1024, maxPayload = 16384;
  CHECK(hipMalloc(&bar, sizeof(Bar))); CHECK(hipMalloc(&sink, 4)); CHECK(hipMalloc(&slabs, (size_t)maxGrid * maxPayload * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  int occ = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_barriers<true>, 256, 0));
  printf("occupancy query: %d workgroups of 256 threads per CU\n", occ);
  for (int payloadKB : {0, 4, 64}) {
    for (int grid : {256, 512, 806, 1024}) {
      if ((grid + 255) / 256 > occ) continue;
      for (int hier = 1; hier >= 0; --hier) {
        float perBarrier[2] = {0, 0};
        unsigned err = 0, stale = 0, members[8] = {0};
        int idx = 0;
        for (int nb : {11, 211}) {
          for (int rep = 0; rep < 2; ++rep) {   // rep 0 warms
            CHECK(hipMemset(bar, 0, sizeof(Bar)));
            CHECK(hipEventRecord(e0));
            if (hier) hipLaunchKernelGGL(k_barriers<true>, dim3(grid), dim3(256), 0, 0, bar, nb, slabs, payloadKB * 256, sink);
            else hipLaunchKernelGGL(k_barriers<false>, dim3(grid), dim3(256), 0, 0, bar, nb, slabs, payloadKB * 256, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            perBarrier[idx] = ms * 1e3f;
          }
          Bar h; CHECK(hipMemcpy(&h, bar, sizeof(Bar), hipMemcpyDeviceToHost));
          err |= h.err[0]; stale += h.stale[0];
          for (int x = 0; x < 8; ++x) members[x] = h.members[x * LINE];
          ++idx;
        }
        // the difference quotient removes launch + setup: (t(211) - t(11)) / 200
        printf("payload %2d KB  grid %4d  %-14s %6.2f us per barrier  (11: %7.1f us, 211: %8.1f us)%s%s  XCC members:",
               payloadKB, grid, hier ? "xcd-hierarchic" : "single-counter", (perBarrier[1] - perBarrier[0]) / 200.f,
               perBarrier[0], perBarrier[1], err ? "  [SPIN LIMIT HIT]" : "", stale ? "  [STALE READS]" : "");
        for (int x = 0; x < 8; ++x) printf(" %u", members[x]);
        printf("\n");
      }
    }
  }
  return 0;
}
