// VALU issue-rate microbenchmark for gfx950 (evidence for DESIGN.md section 4 / bench.py's
// `valu_issue`): cycles per wave64 VALU instruction per SIMD for instruction streams of
// different dependency structure, swept over waves per SIMD, with the shader clock MEASURED
// in the kernel (delta s_memtime / delta s_memrealtime x 100 MHz, MI355X_MICROARCH.md DVFS
// item 6) instead of assumed. Prints one JSON object per (mode, waves/SIMD).
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                       \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));        \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

enum Mode {
  INDEP_FMA = 0,   // 16 independent v_fma_f32 per trip
  CHAIN1,          // one dependent chain through all 16
  CHAIN2,          // two interleaved chains
  CHAIN4,          // four interleaved chains
  MUL_CMP_CND,     // v_mul + v_cmp + v_cndmask, independent
  EXP_QUARTER,     // 1/4 v_exp_f32, 3/4 v_fma_f32, independent
  COMPOSITE1,      // the compositing body's shape: sigma -> exp -> alpha -> T chain, one Gaussian
  COMPOSITE2,      // the same, two Gaussians per trip (two independent sigma/exp/alpha chains)
  N_MODES
};
static const char *kName[N_MODES] = {"indep_fma",   "chain1",      "chain2",     "chain4",
                                     "mul_cmp_cnd", "exp_quarter", "composite1", "composite2"};
// VALU instructions per loop trip (checked against the ISA: --save-temps and count)
static const int kInsts[N_MODES] = {16, 16, 16, 16, 48, 16, 0, 0};

template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, unsigned long long *stamps, int iters, float a,
                                        float b) {
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 0.001f + i;
  float T = 1.0f, acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == COMPOSITE1 || MODE == COMPOSITE2) {
      constexpr int G = MODE == COMPOSITE1 ? 1 : 2;
      float al[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float dx = r[0 + g] - a, dy = r[2 + g] - b;
        const float sig = dx * (r[4 + g] * dx + r[6 + g] * dy) + r[8 + g] * dy * dy;
        float e = r[10 + g] * __builtin_amdgcn_exp2f(-sig);
        e = fminf(e, 0.999f);
        al[g] = (sig >= 0.f && e >= 0.00392f) ? e : 0.f;
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float w = al[g] * T;
        acc0 = __builtin_fmaf(w, r[12], acc0);
        acc1 = __builtin_fmaf(w, r[13], acc1);
        acc2 = __builtin_fmaf(w, r[14], acc2);
        T = T - w;
        r[0 + g] += 1e-7f;  // keep the loads "live": next Gaussian's parameters differ
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (MODE == INDEP_FMA) r[i] = __builtin_fmaf(r[i], a, b);
        if (MODE == CHAIN1) r[i] = __builtin_fmaf(r[(i + 15) & 15], a, r[i]);
        if (MODE == CHAIN2) r[i] = __builtin_fmaf(r[(i + 14) & 15], a, r[i]);
        if (MODE == CHAIN4) r[i] = __builtin_fmaf(r[(i + 12) & 15], a, r[i]);
        if (MODE == MUL_CMP_CND) {
          float t = r[i] * a;
          r[i] = t > b ? t : r[i];
        }
        if (MODE == EXP_QUARTER)
          r[i] = (i & 3) == 0 ? __builtin_amdgcn_exp2f(r[i]) : __builtin_fmaf(r[i], a, b);
      }
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  float s = T + acc0 + acc1 + acc2;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;      // shader cycles
    stamps[2 * blockIdx.x + 1] = w1 - w0;  // 100 MHz ticks
  }
}

template <int MODE>
int run(int insts_per_iter) {
  const int max_blocks = 1024 * 8;
  float *out;
  unsigned long long *stamps;
  CHECK(hipMalloc(&out, sizeof(float) * 64 * max_blocks));
  CHECK(hipMalloc(&stamps, sizeof(unsigned long long) * 2 * max_blocks));
  std::vector<unsigned long long> h(2 * max_blocks);
  const int iters = 40000;
  for (int wps : {1, 2, 4, 5, 8}) {
    const int blocks = 1024 * wps;  // 256 CUs x 4 SIMDs x wps waves
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w)  // warm the clock governor
      hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, stamps, iters, 1.0001f, 0.5f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, stamps, iters, 1.0001f, 0.5f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
    std::vector<double> ghz(blocks), cyc(blocks);
    for (int b = 0; b < blocks; ++b) {
      cyc[b] = (double)h[2 * b];
      ghz[b] = (double)h[2 * b] / (double)h[2 * b + 1] * 0.1;
    }
    std::sort(ghz.begin(), ghz.end());
    std::sort(cyc.begin(), cyc.end());
    const double clock = ghz[blocks / 2];
    // a wave's own lifetime in cycles / its instructions, and the per-SIMD aggregate rate
    const double wave_cpi = cyc[blocks / 2] / ((double)iters * insts_per_iter);
    const double simd_cpi = ms * 1e-3 * clock * 1e9 / ((double)wps * iters * insts_per_iter);
    printf("{\"mode\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"clock_ghz\": %.3f, "
           "\"insts_per_trip\": %d, \"cycles_per_inst_per_simd\": %.3f, "
           "\"cycles_per_inst_one_wave\": %.3f}\n",
           kName[MODE], wps, ms, clock, insts_per_iter, simd_cpi, wave_cpi);
    fflush(stdout);
  }
  CHECK(hipFree(out));
  CHECK(hipFree(stamps));
  return 0;
}

int main(int argc, char **argv) {
  // instruction counts of the two composite bodies are passed in (counted from the ISA by
  // tools/ubench/run_valu_rate.sh); the pure modes have fixed counts
  const int c1 = argc > 1 ? atoi(argv[1]) : 0, c2 = argc > 2 ? atoi(argv[2]) : 0;
  int rc = 0;
  rc |= run<INDEP_FMA>(kInsts[INDEP_FMA]);
  rc |= run<CHAIN1>(kInsts[CHAIN1]);
  rc |= run<CHAIN2>(kInsts[CHAIN2]);
  rc |= run<CHAIN4>(kInsts[CHAIN4]);
  rc |= run<MUL_CMP_CND>(kInsts[MUL_CMP_CND]);
  rc |= run<EXP_QUARTER>(kInsts[EXP_QUARTER]);
  if (c1 > 0) rc |= run<COMPOSITE1>(c1);
  if (c2 > 0) rc |= run<COMPOSITE2>(c2);
  return rc;
}
