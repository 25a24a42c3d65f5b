// VALU issue-rate microbenchmark for gfx950 (evidence for DESIGN.md section 4 / bench.py's
// `valu_issue`): cycles per wave64 VALU instruction per SIMD for instruction streams of
// different dependency structure, at a CONTROLLED number of waves per SIMD.
//
// Placement: ONE workgroup of 64*4*w threads on one CU puts exactly w waves on each of the
// CU's 4 SIMDs (a workgroup's waves are dealt round-robin to the SIMDs); a grid of many small
// workgroups, as used first, leaves the per-SIMD count to the dispatcher and made the rates
// unreadable. The rate is taken inside the kernel: every wave stamps s_memtime before and
// after its loop, rate = (last end - first start) / (instructions per wave * w).
// The shader clock is MEASURED (delta s_memtime / delta s_memrealtime x 100 MHz,
// MI355X_MICROARCH.md DVFS item 6), both for the idle chip (one workgroup) and with all
// 256 CUs running the same loop (`loaded`).
//
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize tools/ubench/valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                       \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));        \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

enum Mode {
  INDEP_FMA = 0,   // 16 independent v_fma_f32 per trip
  CHAIN1,          // one dependent chain through all 16
  CHAIN2,          // two interleaved chains
  MUL_CMP_CND,     // v_mul + v_cmp + v_cndmask (VCC hazard: s_nop between cmp and cndmask)
  EXP_QUARTER,     // 1/4 v_exp_f32, 3/4 v_fma_f32, independent
  PK_FMA,          // 8 independent v_pk_fma_f32 (16 fp32 fmas) per trip
  PK_CHAIN,        // 8 v_pk_fma_f32 in one dependent chain
  COMPOSITE1,      // the compositing body's shape: sigma -> exp -> alpha -> T chain, one Gaussian
  COMPOSITE2,      // the same, two Gaussians per trip (two independent sigma/exp/alpha chains)
  RCP_ALL,         // 16 v_rcp_f32 (transcendental pipe alone)
  PERMLANE32,      // 8 x (v_permlane32_swap + v_add): the backward's halving-tree step
  PERMLANE16,      // 8 x (v_permlane16_swap + v_add)
  DPP_ROW,         // 16 x v_add_f32 row_ror:4 (DPP, full rate?)
  DPP_BCAST,       // 8 x (row_bcast:15 + row_bcast:31 adds, with their s_nop)
  READFIRST,       // 16 x (v_readfirstlane + s_add dependent on it)
  CMP_E64_CND,     // 16 x (v_cmp_gt_f32_e64 s[..] ; v_cndmask_b32_e64 with that SGPR pair)
  N_MODES
};
static const char *kName[N_MODES] = {"indep_fma", "chain1",   "chain2",     "mul_cmp_cnd", "exp_quarter",
                                     "pk_fma",    "pk_chain", "composite1", "composite2",  "rcp_all",
                                     "permlane32_swap_add", "permlane16_swap_add", "dpp_row_ror_add",
                                     "dpp_bcast_add", "readfirstlane_salu", "cmp_e64_cndmask"};

template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, unsigned long long *stamps, int iters, float a_,
                                          float b_) {
  float r[16];
  // loop constants in VGPRs (an SGPR operand pair would hit the constant-bus limit)
  const float a = a_ + threadIdx.x * 1e-9f, b = b_ + threadIdx.x * 1e-9f;
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 0.001f + i;
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = f2{r[2 * i], r[2 * i + 1]};
  const f2 pa = f2{a, a}, pb = f2{b, b};
  float T = 1.0f, acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
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
        r[0 + g] += 1e-7f;  // the next Gaussian's parameters differ
      }
    } else if constexpr (MODE == RCP_ALL) {
#pragma unroll
      for (int i = 0; i < 16; ++i) r[i] = __builtin_amdgcn_rcpf(r[i]);
    } else if constexpr (MODE == PERMLANE32 || MODE == PERMLANE16) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        auto v = MODE == PERMLANE32
                     ? __builtin_amdgcn_permlane32_swap(__float_as_uint(r[2 * i]), __float_as_uint(r[2 * i + 1]), false, false)
                     : __builtin_amdgcn_permlane16_swap(__float_as_uint(r[2 * i]), __float_as_uint(r[2 * i + 1]), false, false);
        const unsigned v0 = v[0], v1 = v[1];
        r[2 * i] = __uint_as_float(v0) + __uint_as_float(v1);
      }
    } else if constexpr (MODE == DPP_ROW) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r[i]), 0x124, 0xf, 0xf, false);
        r[i] += __builtin_bit_cast(float, m);
      }
    } else if constexpr (MODE == DPP_BCAST) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(r[i]));
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(r[i]));
      }
    } else if constexpr (MODE == READFIRST) {
      int acc_s = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc_s += __builtin_amdgcn_readfirstlane(__float_as_int(r[i])) & 0xff;
        r[i] += 1e-7f;
      }
      if (acc_s == 12345) r[0] += 1.f;           // keep the scalar chain alive
    } else if constexpr (MODE == CMP_E64_CND) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool c = r[i] > r[(i + 1) & 15];     // both operands VGPRs -> SGPR-pair result
        r[i] = c ? a : b;
      }
    } else if constexpr (MODE == PK_FMA || MODE == PK_CHAIN) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == PK_FMA) p[i] = __builtin_elementwise_fma(p[i], pa, pb);
        if (MODE == PK_CHAIN) p[i] = __builtin_elementwise_fma(p[(i + 7) & 7], pa, p[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (MODE == INDEP_FMA) r[i] = __builtin_fmaf(r[i], a, b);
        if (MODE == CHAIN1) r[i] = __builtin_fmaf(r[(i + 15) & 15], a, r[i]);
        if (MODE == CHAIN2) r[i] = __builtin_fmaf(r[(i + 14) & 15], a, r[i]);
        if (MODE == MUL_CMP_CND) {
          float t = r[i] * a;
          r[i] = t > b ? t : r[i];
        }
        if (MODE == EXP_QUARTER)
          r[i] = (i & 3) == 0 ? __builtin_amdgcn_exp2f(r[i]) : __builtin_fmaf(r[i], a, b);
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  float s = T + acc0 + acc1 + acc2;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    stamps[4 * w + 0] = c0;
    stamps[4 * w + 1] = c1;
    stamps[4 * w + 2] = w0;
    stamps[4 * w + 3] = w1;
  }
}

template <int MODE>
int run(int insts_per_iter) {
  if (insts_per_iter <= 0) return 0;
  const int max_blocks = 256 * 2;
  float *out;
  unsigned long long *stamps;
  CHECK(hipMalloc(&out, sizeof(float) * 1024 * max_blocks));
  CHECK(hipMalloc(&stamps, sizeof(unsigned long long) * 4 * 16 * max_blocks));
  std::vector<unsigned long long> h(4 * 16 * max_blocks);
  const int iters = 40000;
  for (int loaded = 0; loaded < 2; ++loaded) {
    for (int w : {1, 2, 4}) {            // waves per SIMD
      const int threads = 256 * w;       // 4 SIMDs x w waves
      // loaded: one such workgroup per CU (256) -- at w <= 2 two workgroups could share a CU,
      // so the per-SIMD figure is taken from each workgroup's own span either way
      const int blocks = loaded ? 256 : 1;
      for (int rep = 0; rep < 2; ++rep)
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, stamps, iters, 1.0001f, 0.5f);
      CHECK(hipDeviceSynchronize());
      const int waves_per_block = threads / 64;
      CHECK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 4 * waves_per_block * blocks,
                      hipMemcpyDeviceToHost));
      std::vector<double> span(blocks), ghz(blocks), onewave(blocks);
      for (int b = 0; b < blocks; ++b) {
        unsigned long long s0 = ~0ull, s1 = 0, wt0 = ~0ull, wt1 = 0;
        double own = 0;
        for (int wv = 0; wv < waves_per_block; ++wv) {
          const unsigned long long *q = &h[4 * ((size_t)b * waves_per_block + wv)];
          s0 = std::min(s0, q[0]); s1 = std::max(s1, q[1]);
          wt0 = std::min(wt0, q[2]); wt1 = std::max(wt1, q[3]);
          own += (double)(q[1] - q[0]);
        }
        span[b] = (double)(s1 - s0);
        ghz[b] = (double)(s1 - s0) / (double)(wt1 - wt0) * 0.1;
        onewave[b] = own / waves_per_block;
      }
      std::sort(span.begin(), span.end());
      std::sort(ghz.begin(), ghz.end());
      std::sort(onewave.begin(), onewave.end());
      const double n_inst = (double)iters * insts_per_iter;
      printf("{\"mode\": \"%s\", \"chip\": \"%s\", \"waves_per_simd\": %d, \"clock_ghz\": %.3f, "
             "\"insts_per_trip\": %d, \"cycles_per_inst_per_simd\": %.3f, "
             "\"cycles_per_inst_one_wave\": %.3f}\n",
             kName[MODE], loaded ? "all 256 CUs busy" : "one CU", w, ghz[blocks / 2], insts_per_iter,
             span[blocks / 2] / (n_inst * w), onewave[blocks / 2] / n_inst);
      fflush(stdout);
    }
  }
  CHECK(hipFree(out));
  CHECK(hipFree(stamps));
  return 0;
}

int main(int argc, char **argv) {
  // VALU instructions per loop trip of every mode, counted from the ISA by
  // tools/ubench/run_valu_rate.sh (argv[1..N_MODES]); 0 skips a mode
  int n[N_MODES] = {16, 16, 16, 48, 16, 8, 8, 0, 0, 16, 16, 16, 32, 16, 32, 32};
  for (int m = 0; m < N_MODES && m + 1 < argc; ++m) n[m] = atoi(argv[m + 1]);
  int rc = 0;
  rc |= run<INDEP_FMA>(n[INDEP_FMA]);
  rc |= run<CHAIN1>(n[CHAIN1]);
  rc |= run<CHAIN2>(n[CHAIN2]);
  rc |= run<MUL_CMP_CND>(n[MUL_CMP_CND]);
  rc |= run<EXP_QUARTER>(n[EXP_QUARTER]);
  rc |= run<PK_FMA>(n[PK_FMA]);
  rc |= run<PK_CHAIN>(n[PK_CHAIN]);
  rc |= run<COMPOSITE1>(n[COMPOSITE1]);
  rc |= run<COMPOSITE2>(n[COMPOSITE2]);
  rc |= run<RCP_ALL>(n[RCP_ALL]);
  rc |= run<PERMLANE32>(n[PERMLANE32]);
  rc |= run<PERMLANE16>(n[PERMLANE16]);
  rc |= run<DPP_ROW>(n[DPP_ROW]);
  rc |= run<DPP_BCAST>(n[DPP_BCAST]);
  rc |= run<READFIRST>(n[READFIRST]);
  rc |= run<CMP_E64_CND>(n[CMP_E64_CND]);
  return rc;
}
