// VALU issue-rate microbenchmark (experiment): independent v_fma_f32 / mixed streams,
// waves per SIMD swept. Prints cycles per wave64 instruction per SIMD at an assumed clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, int iters, float a, float b) {
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) r[i] = __builtin_fmaf(r[i], a, b);                       // independent fma
      if (MODE == 1) r[i] = __builtin_fmaf(r[(i + 15) & 15], a, r[i]);        // chain across regs
      if (MODE == 2) r[i] = (i & 3) == 0 ? __builtin_amdgcn_exp2f(r[i]) : __builtin_fmaf(r[i], a, b);
      if (MODE == 3) { float t = r[i] * a; r[i] = t > b ? t : r[i]; }         // mul + cmp + cndmask
      if (MODE == 4) r[i] = __builtin_fmaf(r[(i + 14) & 15], a, r[i]);        // 2 interleaved chains
      if (MODE == 5) r[i] = __builtin_fmaf(r[(i + 12) & 15], a, r[i]);        // 4 interleaved chains
      if (MODE == 6) r[i] = __builtin_fmaf(r[(i + 13) & 15], a, r[i]);        // 3 interleaved chains
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int insts_per_iter) {
  float *out;
  hipMalloc(&out, sizeof(float) * 64 * 1024 * 16);
  const int iters = 20000;
  for (int wps : {1, 2, 4, 8}) {
    int blocks = 1024 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst_per_simd = (double)wps * iters * insts_per_iter;
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n",
           name, wps, ms, cyc / inst_per_simd);
  }
  hipFree(out);
}

int main() {
  run<0>("independent v_fma_f32", 16);
  run<1>("dependent-across v_fma_f32", 16);
  run<2>("1/4 v_exp + 3/4 fma", 16);
  run<3>("mul+cmp+cndmask", 48);
  run<4>("2 interleaved fma chains", 16);
  run<6>("3 interleaved fma chains", 16);
  run<5>("4 interleaved fma chains", 16);
  return 0;
}
