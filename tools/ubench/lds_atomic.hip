// LDS float-atomic throughput (gfx950): can a wave's per-Gaussian partial sums go through
// ds_add_f32 instead of a cross-lane reduction tree? One workgroup of 256*w threads = w waves per
// SIMD on one CU; every wave issues `n` returnless ds_add_f32 per trip into its own LDS region,
// lanes l and l+32 hitting the same address (32 distinct addresses per instruction, stride 9
// floats = conflict-free banks), interleaved with `f` independent v_fma_f32 per atomic.
// Prints cycles per loop trip per wave and the implied LDS cycles per ds_add on the CU.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_atomic.hip -o lds_atomic && ./lds_atomic
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

template <int FMAS, int SLOTS>
__global__ void __launch_bounds__(1024) k(float *out, unsigned long long *stamps, int iters) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *mine = lds + wave * (4 * 32 * 9);                  // 4.6 KB per wave
  for (int i = lane; i < 4 * 32 * 9; i += 64) mine[i] = 0.f;
  __syncthreads();
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = lane * 0.001f + i;
  const float a = 1.0001f + lane * 1e-9f, b = 0.5f;
  float *slot = mine + (lane & (SLOTS - 1)) * 9;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      __hip_atomic_fetch_add(slot + s + (it & 3) * 288, r[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
      for (int f = 0; f < FMAS; ++f) r[(s + f) & 15] = __builtin_fmaf(r[(s + f) & 15], a, b);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
  out[threadIdx.x] = s + mine[lane];
  if (lane == 0) {
    stamps[2 * wave] = c0;
    stamps[2 * wave + 1] = c1;
  }
}

template <int FMAS, int SLOTS>
void run() {
  float *out;
  unsigned long long *st;
  hipMalloc(&out, 4096);
  hipMalloc(&st, 16 * 2 * 8);
  const int iters = 20000;
  for (int w : {1, 2, 4, 5}) {
    const int threads = 256 * w;
    const size_t lds = (size_t)(threads / 64) * 4 * 32 * 9 * 4;
    if (threads > 1024) {           // 5 waves/SIMD: 20 waves -> two workgroups of 640 cannot be forced on one CU
      continue;
    }
    hipLaunchKernelGGL((k<FMAS, SLOTS>), dim3(1), dim3(threads), lds, 0, out, st, iters);
    hipLaunchKernelGGL((k<FMAS, SLOTS>), dim3(1), dim3(threads), lds, 0, out, st, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * (threads / 64));
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long s0 = ~0ull, s1 = 0;
    for (size_t i = 0; i < h.size() / 2; ++i) {
      s0 = std::min(s0, h[2 * i]);
      s1 = std::max(s1, h[2 * i + 1]);
    }
    const double span = (double)(s1 - s0);
    const double per_trip = span / iters;                       // all waves run concurrently
    const double lds_per_add = span / ((double)iters * 9 * (threads / 64));   // if LDS were the only limit
    printf("{\"fmas_per_atomic\": %d, \"slots\": %d, \"waves_per_simd\": %d, \"cycles_per_trip\": %.1f, "
           "\"cycles_per_ds_add_per_cu\": %.2f}\n", FMAS, SLOTS, w, per_trip, lds_per_add);
  }
  hipFree(out);
  hipFree(st);
}

int main() {
  run<0, 32>();     // atomics only, 2 lanes per address
  run<0, 16>();     // 4 lanes per address
  run<0, 64>();     // every lane its own address
  run<8, 32>();     // 8 fmas of VALU work behind each atomic (the kernel's ratio is ~20)
  run<20, 32>();
  return 0;
}
