// depthnet.hip -- B10: the dense monocular depth network (Metric3D v2: DINOv2-reg ViT encoder +
// RAFT-DPT decoder) on the gfx950 matrix cores, fp16 operands / fp32 accumulation.
//
// Architecture spec (read as text, nothing copied): /root/reference/gs_init_compare/third_party/
// metric3d/mono/model/backbones/ViT_DINO_reg.py:755-1270 (DinoVisionTransformer; vit_small_reg
// :1192, vit_large_reg :1227) and .../decode_heads/RAFTDepthNormalDPTDecoder5.py:736-1035, reached
// from depth_prediction/predictors/metric3d.py:87-88 (`model.inference`).
//
// Building blocks (all hand-written, no BLAS / MIOpen):
//   gemm_kernel        C = epilogue(A[M,K] * W[N,K]^T): 128x128 / 128x64 / 64x64 tiles x K 64, 4 waves of
//                      v_mfma_f32_32x32x16_f16 tiles, LDS-DMA double buffer, fused bias / GELU / ReLU /
//                      sigmoid / tanh / layer-scale / residual epilogue. Every Linear, every 1x1
//                      convolution (NHWC) and every 3x3 convolution (taps gathered straight from the
//                      map, optionally from a virtual concatenation of two maps).
//   gemm8p_kernel      the same contract on a 256x256x64 tile in eight phases per two K-tiles (8 waves
//                      of 128x64, v_mfma_f32_16x16x32_f16, half-tile staging seven phases ahead with one
//                      counted wait per K-tile, epilogue through LDS): the launches that fill the chip.
//   conv3_head_kernel  3x3 convolution with <= 8 output channels accumulated into an fp32 field.
//   attention_kernel   flash-style softmax(QK^T/sqrt(d))V for head_dim 64: S^T = K Q^T on MFMA so
//                      that a query is a LANE (softmax over keys = over the lane's registers + one
//                      cross-half exchange), P stays in registers as the B operand of O^T += V^T P.
//   layernorm / im2col / resize / pooling / GRU gates / softmax-expectation / convex upsampling:
//                      HBM-bound element-wise kernels around the GEMMs.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.h"

namespace gsr {
namespace dn {

typedef _Float16 h16;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDT = 72;   // LDS row stride in halves: 144 B = 36 dwords, so the 16 rows of a
                          // quarter-wave's ds_read_b128 start in 16 distinct 4-bank groups

enum Act { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_SIGMOID = 3, ACT_TANH = 4 };

struct GemmArgs {
  int M, N, K;
  const h16 *A;
  int lda;
  const h16 *W;        // [N, K] row-major (nn.Linear / flattened conv weight), K % 64 == 0
  const float *bias;   // [N] or null
  const float *gamma;  // [N] or null: layer scale applied after the activation
  const float *residual;   // [M, ldr] fp32 or null, added last (may alias out32)
  int ldr;
  const h16 *residual16;   // [M, ldr16] fp16 or null, added last
  int ldr16;
  h16 *out16;
  int ldo16;
  float *out32;
  int ldo32;
  int act;
  // implicit-im2col mode (CONV): A = NHWC map [cH*cW, lda], cC channels, cKS x cKS taps, stride 1
  int cH, cW, cC, cKS, cPad;
  const h16 *zero_page;   // >= 16 bytes of zeros
  int pad_to;             // columns [N, pad_to) of out16 are written as zeros (the map's zero channels)
  // CONV only: the map may be a VIRTUAL CONCATENATION -- channels [0, c_split) are read from A2 (row stride
  // lda2), channels [c_split, cC) from A; c_split % 64 == 0, so a K-tile never straddles the seam
  const h16 *A2;
  int lda2, c_split;
};

// GELU with the erf of Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 rounding
// of the outputs): one exp, one reciprocal and a degree-5 polynomial instead of the library erff's
// two-range evaluation -- the epilogue of the 3349 x 4096 MLP GEMM evaluates it 13.7 M times
// (76 -> 62 us for that launch). 1 + erf(x) is formed as 2 - tail(x) / tail(-x) without cancellation.
__device__ __forceinline__ float gelu_fast(float v) {
  const float x = fabsf(v) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f),
                              0.254829592f);
  const float tail = poly * __expf(-x * x);            // 1 - erf(|x|)
  return 0.5f * v * (v >= 0.f ? 2.0f - tail : tail);
}

template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
  if (ACT == ACT_GELU) return gelu_fast(v);
  if (ACT == ACT_RELU) return fmaxf(v, 0.f);
  if (ACT == ACT_SIGMOID) return 1.0f / (1.0f + __expf(-v));
  if (ACT == ACT_TANH) return tanhf(v);
  return v;
}

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
    case ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// LDS image of an operand tile: [128 rows][64 halves], rows of 128 B, UNPADDED (the LDS-DMA below
// writes wave-uniform base + lane*16, so the image must be lane-linear); the eight 16-byte chunks of
// a row are stored XOR-swizzled, physical chunk = logical chunk ^ ((row >> 1) & 7): the 16 rows a
// quarter-wave's ds_read_b128 touches then start in 16 distinct 4-bank groups (conflict-free). The
// swizzle is applied on the SOURCE side: the lane that fills physical chunk p of row r fetches the
// row's logical chunk p ^ ((r >> 1) & 7) from global memory.
//
// CONV: A is not a matrix in memory but the im2col view of an NHWC map (3x3 / 1x1, stride 1):
// row m = output pixel, column k = tap * C + c. With C a multiple of 64 a 64-wide K-tile lies inside
// one tap, so every lane's chunk is one 16-byte piece of a neighbouring pixel -- fetched straight from
// the map (out-of-image taps and the K padding come from a zero page): no im2col buffer is written
// or re-read.
// BNT = 128: the 128x128 tile, 2x2 MFMA tiles per wave. BNT = 64: a 128x64 tile (each wave 64x32,
// 48 KB of LDS) for the launches whose 128x128 grid would leave the chip underfilled -- M = 3349
// tokens are 27 row tiles, so N = 1024 gives 216 workgroups for 256 CUs with nothing to overlap
// with; twice as many half-size workgroups, three resident per CU, finish sooner.
// BMT = 64 (with BNT = 64): a 64x64 tile, each wave ONE 32x32 MFMA tile, 32 KB of LDS -- for the launches that
// have fewer 128-row workgroups than the chip has CUs (the 1/14-resolution maps: 3344 rows = 27 row tiles):
// each workgroup there walks its whole K alone on its CU, so twice as many workgroups of half the size finish sooner.
template <int ACT, bool CONV, int BNT, int BMT = BM>
__global__ void __launch_bounds__(256, 2)
gemm_kernel(GemmArgs p) {
  constexpr int NAS = BMT / 32;     // A staging instructions per wave
  constexpr int NI = BMT / 64;      // 32-row MFMA tiles per wave
  constexpr int NBS = BNT / 32;     // B staging instructions per wave
  constexpr int NJ = BNT / 64;      // 32-column MFMA tiles per wave
  __shared__ __attribute__((aligned(1024))) h16 smem[2][(BMT + BNT) * BK];   // [buffer][A rows | B rows][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware order: consecutive workgroup ids are dealt round-robin to the 8 XCDs; remap so that
  // each XCD walks a contiguous run of tiles of the same row panel (A panel stays in its L2)
  const int nbx = gridDim.x, nby = gridDim.y, nwg = nbx * nby;
  int wg = blockIdx.y * nbx + blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, x = wg & 7;
    wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (wg >> 3);
  }
  const int m0 = (wg / nbx) * BMT, n0 = (wg % nbx) * BNT;
  const int lr = lane & 31, lh = lane >> 5;

  // staging: wave w, instruction i moves rows (4w + i)*8 .. +7 of A (NBS*w + i for B);
  // lane -> (row, physical chunk)
  const int srow = lane >> 3, pch = lane & 7;
  const h16 *ga[NAS], *gb[NBS];
  int pix_y[NAS], pix_x[NAS], lch[NAS];
#pragma unroll
  for (int i = 0; i < NAS; ++i) {
    const int r = (wave * NAS + i) * 8 + srow;
    lch[i] = (pch ^ ((r >> 1) & 7)) * 8;                 // logical chunk offset in halves
    const int m = min(m0 + r, p.M - 1);
    if (CONV) {
      pix_y[i] = m / p.cW;
      pix_x[i] = m - pix_y[i] * p.cW;
      ga[i] = nullptr;
    } else {
      ga[i] = p.A + (int64_t)m * p.lda + lch[i];
    }
  }
#pragma unroll
  for (int i = 0; i < NBS; ++i) {
    const int r = (wave * NBS + i) * 8 + srow;
    gb[i] = p.W + (int64_t)min(n0 + r, p.N - 1) * p.K + (pch ^ ((r >> 1) & 7)) * 8;
  }
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  auto stage = [&](int buf, int k0) {
    int ky = 0, kx = 0, c0 = 0;
    bool tap_ok = true;
    if (CONV) {
      const int tap = k0 / p.cC;                          // wave-uniform
      c0 = k0 - tap * p.cC;
      ky = tap / p.cKS;
      kx = tap - ky * p.cKS;
      tap_ok = tap < p.cKS * p.cKS;
    }
#pragma unroll
    for (int i = 0; i < NAS; ++i) {
      const h16 *src;
      if (CONV) {
        const int iy = pix_y[i] + ky - p.cPad, ix = pix_x[i] + kx - p.cPad;
        const bool ok = tap_ok && iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW;
        const bool second = c0 < p.c_split;                // wave-uniform: this K-tile's channels come from A2
        src = ok ? (second ? p.A2 + ((int64_t)iy * p.cW + ix) * p.lda2 : p.A + ((int64_t)iy * p.cW + ix) * p.lda) + c0 + lch[i]
                 : p.zero_page;
      } else {
        src = ga[i] + k0;
      }
      h16 *dA = &smem[buf][(wave * NAS + i) * 8 * BK];
      __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)dA, 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NBS; ++i) {
      h16 *dB = &smem[buf][BMT * BK + (wave * NBS + i) * 8 * BK];
      __builtin_amdgcn_global_load_lds((glb_void *)(gb[i] + k0), (lds_void *)dB, 16, 0, 0);
    }
  };

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment rows of this lane and their swizzle terms
  const int ra0 = wm * (BMT / 2) + lr, ra1 = ra0 + 32, rb0 = wn * (BNT / 2) + lr, rb1 = rb0 + 32;
  const int xa0 = (ra0 >> 1) & 7, xa1 = (ra1 >> 1) & 7, xb0 = (rb0 >> 1) & 7, xb1 = (rb1 >> 1) & 7;

  const int nk = p.K / BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA has landed: explicit, not left
  __syncthreads();            // to the barrier's lowering (which happens to wait vmcnt(0) on ROCm 7.2)
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(buf ^ 1, (kt + 1) * BK);
    const h16 *sA = smem[buf], *sB = smem[buf] + BMT * BK;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      const int cl = 2 * s + lh;
      const half8 af0 = *reinterpret_cast<const half8 *>(sA + ra0 * BK + ((cl ^ xa0) << 3));
      const half8 bf0 = *reinterpret_cast<const half8 *>(sB + rb0 * BK + ((cl ^ xb0) << 3));
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af0, bf0, acc[0][0], 0, 0, 0);
      if constexpr (NI == 2) {
        const half8 af1 = *reinterpret_cast<const half8 *>(sA + ra1 * BK + ((cl ^ xa1) << 3));
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af1, bf0, acc[1][0], 0, 0, 0);
        if constexpr (NJ == 2) {
          const half8 bf1 = *reinterpret_cast<const half8 *>(sB + rb1 * BK + ((cl ^ xb1) << 3));
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af0, bf1, acc[0][1], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af1, bf1, acc[1][1], 0, 0, 0);
        }
      } else {
        static_assert(NI == 2 || NJ == 1, "the 64-row tile is built 64 columns wide");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (as above: not left to the barrier's lowering)
    __syncthreads();          // tile kt+1 has landed, and every wave is done reading tile kt
    buf ^= 1;
  }

  // epilogue: lane holds column n, rows (reg&3) + 8*(reg>>2) + 4*(lane>>5) of each 32x32 tile
  auto emit = [&](const f32x16 &a, int i, int j) {
    const int n = n0 + wn * (BNT / 2) + j * 32 + lr;
    if (n >= p.N) {
      if (n < p.pad_to) {      // zero channels of the output map, written here instead of by a fill launch
        const int mb0 = m0 + wm * (BMT / 2) + i * 32 + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb0 + (r & 3) + 8 * (r >> 2);
          if (m < p.M) p.out16[(int64_t)m * p.ldo16 + n] = (h16)0.f;
        }
      }
      return;
    }
    const float b = p.bias ? p.bias[n] : 0.f;
    const float g = p.gamma ? p.gamma[n] : 1.f;
    const int mb = m0 + wm * (BMT / 2) + i * 32 + 4 * lh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mb + (r & 3) + 8 * (r >> 2);
      if (m >= p.M) continue;
      float v = act_fn<ACT>(a[r] + b) * g;
      if (p.residual) v += p.residual[(int64_t)m * p.ldr + n];
      if (p.residual16) v += (float)p.residual16[(int64_t)m * p.ldr16 + n];
      if (p.out32) p.out32[(int64_t)m * p.ldo32 + n] = v;
      if (p.out16) p.out16[(int64_t)m * p.ldo16 + n] = (h16)v;
    }
  };
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) emit(acc[i][j], i, j);
}

constexpr int BM2 = 256, BN2 = 256;
// LDS-DMA as asm statements (raster_common.h explains why): destination = wave-uniform LDS byte address
__device__ __forceinline__ void dn_dma_16B(const void *src_lane, uint32_t lds_dst_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src_lane), "s"(lds_dst_uniform)
      : "memory");
}
// the same with a wave-uniform 64-bit base (SGPR pair) and a 32-bit byte offset per lane
__device__ __forceinline__ void dn_dma_16B_off(const void *base_uniform, uint32_t byte_off_lane, uint32_t lds_dst_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(byte_off_lane), "s"(base_uniform), "s"(lds_dst_uniform)
      : "memory");
}
// ---- 256x256x64 tile in eight phases per two K-tiles (the guide's verified structure, section 5) -------
// Eight waves as 2 (rows) x 4 (columns), 128x64 outputs each, v_mfma_f32_16x16x32_f16. A K-tile is
// staged as FOUR half-tiles of 16 KB -- A0 / A1 = the upper / lower 64 rows of each wave row's 128,
// B0 / B1 = the left / right 32 columns of each wave column's 64 -- and computed as four QUADRANTS
// (64 x 32 outputs x K = 64 = 16 MFMAs per wave): (A0,B0) (A0,B1) (A1,B1) (A1,B0); a phase reads the
// fragments its quadrant is missing (12 / 4 / 8 / 0 ds_read_b128), stages ONE half-tile seven ahead of
// the phase number (two DMA instructions per wave), and issues its 16 MFMAs. The two wave rows run one
// barrier apart, so in every barrier interval one wave per SIMD issues MFMAs while the other reads
// and stages. Eight half-tile slots (two K-tiles); the DMAs are waited for ONCE per K-tile with a
// counted vmcnt(6) that leaves three half-tiles in flight across the barriers:
//   half-tile q = 4t + i in order [A0 B0 B1 A1] is read in phase 4t + {0,0,1,2}[i]; phase P stages
//   q = P + 7 into slot q & 7, whose previous tenant q - 8 was last read in a phase <= P - 1 and whose
//   reads were retired (lgkmcnt(0)) before that phase's barrier (write-after-read); the wait in phase
//   4t + 3 retires all of tile t + 1, both wave rows pass a barrier after their wait before anyone reads
//   it in phase 4t + 4 (read-after-write: an LDS-DMA is ordered for a ds_read only by the issuing
//   wave's vmcnt followed by a barrier).
// Operands are passed to the MFMA swapped (W fragment first), so a lane ends up with FOUR CONSECUTIVE
// columns of one output row: 8-byte fp16 / 16-byte fp32 stores in the epilogue.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifdef GSR_GEMM_TIMELINE
// diagnostic build: s_memrealtime (100 MHz) at kernel start / tile 0 landed / main loop done / epilogue done, per workgroup
__device__ unsigned long long *g_gemm_timeline = nullptr;
#define GEMM_STAMP(k)                                                                      \
  do {                                                                                     \
    if (threadIdx.x == 0 && g_gemm_timeline) {                                             \
      unsigned long long t_;                                                               \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
      g_gemm_timeline[4 * (blockIdx.y * gridDim.x + blockIdx.x) + (k)] = t_;               \
    }                                                                                      \
  } while (0)
#else
#define GEMM_STAMP(k)
#endif
template <int ACT, bool CONV>
__global__ void __launch_bounds__(512)
gemm8p_kernel(GemmArgs p) {
  constexpr int HT = 128 * BK;                     // halves per half-tile (16 KB)
  __shared__ __attribute__((aligned(1024))) h16 smem[8 * HT];   // 128 KB: slot = (tile & 1) * 4 + i
  GEMM_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int nbx = gridDim.x, nby = gridDim.y, nwg = nbx * nby;
  int wg = blockIdx.y * nbx + blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, x = wg & 7;
    wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (wg >> 3);
  }
  const int m0 = (wg / nbx) * BM2, n0 = (wg % nbx) * BN2;
  const int nk = p.K / BK, NQ = 4 * nk;

  // staging: a half-tile is 16 units of 8 rows; wave w moves units 2w and 2w+1. lane -> (row, physical chunk).
  // local row r of an A half h is tile row (r >> 6) * 128 + h * 64 + (r & 63); of a B half h it is
  // tile column (r >> 5) * 64 + h * 32 + (r & 31).
  const int srow = lane >> 3, pch = lane & 7;
  uint32_t ga[2][2], gb[2][2];                    // [half][unit of this wave]: BYTE offsets from p.A / p.W
  int pix_y[2][2], pix_x[2][2], lchs[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 8 + srow;
    lchs[j] = (pch ^ ((r >> 1) & 7)) * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = min(m0 + (r >> 6) * 128 + h * 64 + (r & 63), p.M - 1);
      if (CONV) {
        pix_y[h][j] = m / p.cW;
        pix_x[h][j] = m - pix_y[h][j] * p.cW;
        ga[h][j] = 0;
      } else {
        ga[h][j] = (uint32_t)(((int64_t)m * p.lda + lchs[j]) * 2);      // < 4 GB: checked by the launcher
      }
      const int n = min(n0 + (r >> 5) * 64 + h * 32 + (r & 31), p.N - 1);
      gb[h][j] = (uint32_t)(((int64_t)n * p.K + lchs[j]) * 2);
    }
  }
  typedef __attribute__((address_space(3))) void lds_void;
  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_void *)&smem[0]);
  const uint32_t lds_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + (uint32_t)wave * (2 * 8 * BK * 2)));
  // stage half-tile i (0: A0, 1: B0, 2: B1, 3: A1) of K-tile `tile`
  auto stage = [&](int tile, int i) {
    const int k0 = tile * BK;
    const uint32_t dst = lds_w + (uint32_t)(((tile & 1) * 4 + i) * HT * 2);
    const bool isA = i == 0 || i == 3;
    const int h = (i == 0 || i == 1) ? 0 : 1;
    int ky = 0, kx = 0, c0 = 0;
    bool tap_ok = true;
    if (CONV && isA) {
      const int tap = k0 / p.cC;
      c0 = k0 - tap * p.cC;
      ky = tap / p.cKS;
      kx = tap - ky * p.cKS;
      tap_ok = tap < p.cKS * p.cKS;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t d = dst + (uint32_t)j * (8 * BK * 2);
      if (!isA) {
        dn_dma_16B_off(p.W + k0, gb[h][j], d);
      } else if (CONV) {
        const int iy = pix_y[h][j] + ky - p.cPad, ix = pix_x[h][j] + kx - p.cPad;
        const bool ok = tap_ok && iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW;
        const bool second = c0 < p.c_split;
        dn_dma_16B(ok ? (second ? p.A2 + ((int64_t)iy * p.cW + ix) * p.lda2 : p.A + ((int64_t)iy * p.cW + ix) * p.lda) + c0 + lchs[j]
                      : p.zero_page, d);
      } else {
        dn_dma_16B_off(p.A + k0, ga[h][j], d);
      }
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
  half8 af[4][2], bf[2][2][2];   // A: [row tile of the half][k-step]; B: [half][column tile][k-step]

  // fragment reads: lane l holds row (l & 15), halves 8 * (l >> 4) .. + 7 of k-step ks (32 wide)
  const int l15 = lane & 15, lq = lane >> 4, swz = (l15 >> 1) & 7;
  const int a_row = wr * 64 + l15, b_row = wc * 32 + l15;
  auto read_a = [&](int slot_halves) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        af[m][ks] = *reinterpret_cast<const half8 *>(smem + slot_halves + (a_row + m * 16) * BK + (((lq + 4 * ks) ^ swz) << 3));
  };
  auto read_b = [&](int slot_halves, int h) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int n = 0; n < 2; ++n)
        bf[h][n][ks] = *reinterpret_cast<const half8 *>(smem + slot_halves + (b_row + n * 16) * BK + (((lq + 4 * ks) ^ swz) << 3));
  };
  // quadrant (ah, bh) with the B fragments of register set `set`
  auto mfma_quadrant_x = [&](int ah, int bh, int set) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[ah * 4 + m][bh * 2 + n] =
              __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[set][n][ks], af[m][ks], acc[ah * 4 + m][bh * 2 + n], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto mfma_quadrant = [&](int ah, int bh) { mfma_quadrant_x(ah, bh, bh); };
#define GSR_BAR()                        \
  __builtin_amdgcn_sched_barrier(0);     \
  __builtin_amdgcn_s_barrier();          \
  __builtin_amdgcn_sched_barrier(0)
#define GSR_LOADS_DONE()                                   \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
  GSR_BAR()

  // prologue: half-tiles 0..6 in flight, tile 0 waited for
#pragma unroll
  for (int q = 0; q < 7; ++q)
    if (q < NQ) stage(q >> 2, q & 3);
  if (NQ >= 7) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GSR_BAR();
  GEMM_STAMP(1);
  if (wr == 1) { GSR_BAR(); }                      // the lower wave row runs one barrier behind
  // one K-tile; PAR = t & 1 at compile time (the slot addresses are then constants)
  // (tried: reads balanced 8 / 4 / 8 / 4 by fetching the next tile's B0 fragments in phase 3, with a
  // counted wait in every phase -- 5 % slower at 4096^3, 251 registers instead of 222)
  auto tile = [&](int t, auto par_c) {
    constexpr int sb = decltype(par_c)::value * 4 * HT;     // this tile's slots, in halves
    read_b(sb + 1 * HT, 0);                                 // phase 4t: (A0, B0); stage (tile t + 1, A1)
    __builtin_amdgcn_sched_barrier(0);
    read_a(sb + 0 * HT);
    if (t + 1 < nk) stage(t + 1, 3);
    GSR_LOADS_DONE();
    mfma_quadrant(0, 0);
    GSR_BAR();
    read_b(sb + 2 * HT, 1);                                 // phase 4t + 1: (A0, B1); stage (tile t + 2, A0)
    if (t + 2 < nk) stage(t + 2, 0);
    GSR_LOADS_DONE();
    mfma_quadrant(0, 1);
    GSR_BAR();
    read_a(sb + 3 * HT);                                    // phase 4t + 2: (A1, B1); stage (tile t + 2, B0)
    if (t + 2 < nk) stage(t + 2, 1);
    GSR_LOADS_DONE();
    mfma_quadrant(1, 1);
    GSR_BAR();
    // phase 4t + 3: (A1, B0), fragments already in registers; stage (tile t + 2, B1); the one DMA wait of
    // the K-tile: everything but the last three half-tiles, i.e. all of tile t + 1
    if (t + 2 < nk) {
      stage(t + 2, 2);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    GSR_BAR();
    mfma_quadrant(1, 0);
    GSR_BAR();
  };
  for (int t = 0; t + 1 < nk; t += 2) {
    tile(t, std::integral_constant<int, 0>{});
    tile(t + 1, std::integral_constant<int, 1>{});
  }
  if (nk & 1) tile(nk - 1, std::integral_constant<int, 0>{});
  if (wr == 0) { GSR_BAR(); }
  GEMM_STAMP(2);
#undef GSR_BAR
#undef GSR_LOADS_DONE

  // Epilogue through LDS. In the accumulator layout a store instruction covers 16 rows x 32 bytes --
  // 16 partial lines per instruction, 8 us of store issue for a 256x256 fp16 tile (in-kernel timeline).
  // Each wave instead writes its 128x64 outputs, 64 rows at a time, as fp32 into a private 16 KB of the
  // (now idle) staging memory -- rows of 256 B, 16-byte chunks XOR-ed with the row so that both the
  // accumulator-layout writes and the row-layout reads are conflict-free -- and reads them back with
  // 8 lanes per row, 8 consecutive columns per lane: bias / LayerScale are per-lane constants, the
  // residual loads and the stores are whole 128-byte (fp16) or 256-byte (fp32) row segments.
  {
    const int nw0 = n0 + wc * 64;                                        // this wave's 64 columns
    const bool rows_ok = (!p.out16 || ((((uintptr_t)p.out16) & 15) == 0 && (p.ldo16 & 7) == 0)) &&
                         (!p.out32 || ((((uintptr_t)p.out32) & 15) == 0 && (p.ldo32 & 3) == 0)) &&
                         (!p.residual || ((((uintptr_t)p.residual) & 15) == 0 && (p.ldr & 3) == 0)) &&
                         (!p.residual16 || ((((uintptr_t)p.residual16) & 15) == 0 && (p.ldr16 & 7) == 0)) &&
                         ((((uintptr_t)p.bias) | ((uintptr_t)p.gamma)) & 15) == 0;
    if (rows_ok && nw0 + 64 <= p.N) {                                    // wave-uniform
      float *cw = reinterpret_cast<float *>(smem) + wave * (HT / 2);     // 16 KB = [64 rows][64 floats]
      const int rr = lane >> 3, c8 = lane & 7, ncol = nw0 + c8 * 8;
      f32x4 bia[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, gam[2] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}};
      if (p.bias) {
        bia[0] = *reinterpret_cast<const f32x4 *>(p.bias + ncol);
        bia[1] = *reinterpret_cast<const f32x4 *>(p.bias + ncol + 4);
      }
      if (p.gamma) {
        gam[0] = *reinterpret_cast<const f32x4 *>(p.gamma + ncol);
        gam[1] = *reinterpret_cast<const f32x4 *>(p.gamma + ncol + 4);
      }
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const int row = mt * 16 + l15, ch = (nt >> 1) * 8 + (nt & 1) * 4 + lq;
            *reinterpret_cast<f32x4 *>(cw + row * 64 + ((ch ^ l15) << 2)) = acc[ph * 4 + mt][nt];
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = it * 8 + rr;
          const int m = m0 + wr * 128 + ph * 64 + row;
          f32x4 v[2];
          v[0] = *reinterpret_cast<const f32x4 *>(cw + row * 64 + (((2 * c8) ^ (row & 15)) << 2));
          v[1] = *reinterpret_cast<const f32x4 *>(cw + row * 64 + (((2 * c8 + 1) ^ (row & 15)) << 2));
          if (m >= p.M) continue;
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[hh][r] = act_fn<ACT>(v[hh][r] + bia[hh][r]) * gam[hh][r];
          if (p.residual) {
            const float *rp = p.residual + (int64_t)m * p.ldr + ncol;
            v[0] += *reinterpret_cast<const f32x4 *>(rp);
            v[1] += *reinterpret_cast<const f32x4 *>(rp + 4);
          }
          if (p.residual16) {
            const half8 r16 = *reinterpret_cast<const half8 *>(p.residual16 + (int64_t)m * p.ldr16 + ncol);
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r >> 2][r & 3] += (float)r16[r];
          }
          if (p.out32) {
            float *op = p.out32 + (int64_t)m * p.ldo32 + ncol;
            *reinterpret_cast<f32x4 *>(op) = v[0];
            *reinterpret_cast<f32x4 *>(op + 4) = v[1];
          }
          if (p.out16) {
            half8 o;
#pragma unroll
            for (int r = 0; r < 8; ++r) o[r] = (h16)v[r >> 2][r & 3];
            *reinterpret_cast<half8 *>(p.out16 + (int64_t)m * p.ldo16 + ncol) = o;
          }
        }
      }
#ifdef GSR_GEMM_TIMELINE
      __syncthreads();
      GEMM_STAMP(3);
#endif
      return;
    }
  }
  // (partial column tiles, unaligned rows: element-wise in the accumulator layout)
  const bool vec_ok = (!p.out16 || ((((uintptr_t)p.out16) & 7) == 0 && (p.ldo16 & 3) == 0)) &&
                      (!p.out32 || ((((uintptr_t)p.out32) & 15) == 0 && (p.ldo32 & 3) == 0)) &&
                      (!p.residual || ((((uintptr_t)p.residual) & 15) == 0 && (p.ldr & 3) == 0)) &&
                      (!p.residual16 || ((((uintptr_t)p.residual16) & 7) == 0 && (p.ldr16 & 3) == 0)) &&
                      ((((uintptr_t)p.bias) | ((uintptr_t)p.gamma)) & 15) == 0;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int n = n0 + wc * 64 + (nt >> 1) * 32 + (nt & 1) * 16 + 4 * lq;
    const bool fast = vec_ok && n + 3 < p.N;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, gam4 = {1.f, 1.f, 1.f, 1.f};
    if (fast) {
      if (p.bias) bias4 = *reinterpret_cast<const f32x4 *>(p.bias + n);    // n % 4 == 0; cudaMalloc-aligned vectors
      if (p.gamma) gam4 = *reinterpret_cast<const f32x4 *>(p.gamma + n);
    }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int m = m0 + wr * 128 + (mt >> 2) * 64 + (mt & 3) * 16 + l15;
      if (m >= p.M) continue;
      const f32x4 a = acc[mt][nt];
      if (fast) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act_fn<ACT>(a[r] + bias4[r]) * gam4[r];
        if (p.residual) v += *reinterpret_cast<const f32x4 *>(p.residual + (int64_t)m * p.ldr + n);
        if (p.residual16) {
          const half4 r16 = *reinterpret_cast<const half4 *>(p.residual16 + (int64_t)m * p.ldr16 + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)r16[r];
        }
        if (p.out32) *reinterpret_cast<f32x4 *>(p.out32 + (int64_t)m * p.ldo32 + n) = v;
        if (p.out16) {
          half4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (h16)v[r];
          *reinterpret_cast<half4 *>(p.out16 + (int64_t)m * p.ldo16 + n) = o;
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = n + r;
        if (nn >= p.N) {
          if (nn < p.pad_to) p.out16[(int64_t)m * p.ldo16 + nn] = (h16)0.f;
          continue;
        }
        const float b = p.bias ? p.bias[nn] : 0.f;
        const float g = p.gamma ? p.gamma[nn] : 1.f;
        float v = act_fn<ACT>(a[r] + b) * g;
        if (p.residual) v += p.residual[(int64_t)m * p.ldr + nn];
        if (p.residual16) v += (float)p.residual16[(int64_t)m * p.ldr16 + nn];
        if (p.out32) p.out32[(int64_t)m * p.ldo32 + nn] = v;
        if (p.out16) p.out16[(int64_t)m * p.ldo16 + nn] = (h16)v;
      }
    }
  }
#ifdef GSR_GEMM_TIMELINE
  __syncthreads();
  GEMM_STAMP(3);
#endif
}

// ---- 3x3 convolution with a handful of output channels, accumulated into an fp32 field -------------
// The flow head's last layers (RAFTDepthNormalDPTDecoder5.py:282-297): 128 -> 2 and 128 -> 4 channels at
// 1/4 resolution, added to the fp32 flow field, eight times per image. As im2col rows + GEMM that is a
// 94 MB buffer written and read for 0.4 GFLOP (25 + 40 us per head and iteration); here sixteen lanes
// share a pixel -- lane = (pixel of 4, 8-channel chunk of 16) -- read the nine taps straight from the map,
// take their dot products with the weights (staged once per workgroup in LDS), and a 4-step lane
// reduction leaves the N sums: the map is read once from memory (taps hit the caches), nothing is written
// but the N values.
constexpr int HEAD_MAX_N = 8;
template <int N>
__global__ void __launch_bounds__(256)
conv3_head_kernel(int H, int W, int C, const h16 *__restrict__ in, int ldi, const h16 *__restrict__ wt,
                  int K_pad, const float *__restrict__ bias, float *__restrict__ out, int ldo) {
  extern __shared__ h16 sW[];                       // [N][9 * C]
  const int K = 9 * C;
  for (int i = threadIdx.x * 8; i < N * K; i += 256 * 8) {
    const int n = i / K, k = i - n * K;
    *reinterpret_cast<half8 *>(sW + i) = *reinterpret_cast<const half8 *>(wt + (int64_t)n * K_pad + k);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, chunk = lane & 15, sub = lane >> 4;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
  const int P = H * W;
  for (int p0 = wave * 4; p0 < P; p0 += nwaves * 4) {
    const int pix = min(p0 + sub, P - 1);
    const int y = pix / W, x = pix - y * W;
    float acc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = 0.f;
    for (int c0 = chunk * 8; c0 < C; c0 += 128) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const half8 v = *reinterpret_cast<const half8 *>(in + ((int64_t)iy * W + ix) * ldi + c0);
#pragma unroll
        for (int n = 0; n < N; ++n) {
          const half8 w = *reinterpret_cast<const half8 *>(sW + n * K + tap * C + c0);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[n] = fmaf((float)v[e], (float)w[e], acc[n]);
        }
      }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) acc[n] += __shfl_xor(acc[n], m, 64);
    }
    if (chunk == 0 && p0 + sub < P) {
#pragma unroll
      for (int n = 0; n < N; ++n) out[(int64_t)pix * ldo + n] += acc[n] + (bias ? bias[n] : 0.f);
    }
  }
}

// ---- LayerNorm over the last dimension: one wave per row, fp32 statistics -------------------
template <typename TIN>
__global__ void __launch_bounds__(256)
layernorm_kernel(int M, int D, const TIN *__restrict__ x, int ldx, const float *__restrict__ gamma,
                 const float *__restrict__ beta, float eps, h16 *__restrict__ out16, int ldo16,
                 float *__restrict__ out32, int ldo32, int relu) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const TIN *xr = x + (int64_t)row * ldx;
  if constexpr (sizeof(TIN) == 4) {
    // fp32 rows of 256 / 512 / 768 / 1024 elements (the encoder's token rows): four consecutive elements per
    // lane and 256-element chunk -- float4 loads of x, gamma, beta, 8-byte fp16 stores: a quarter of the memory
    // instructions of the element-per-lane form below (49 launches per ViT-L image, 11 -> 7 us each)
    const bool vec = (D & 255) == 0 && D <= 1024 && (ldx & 3) == 0 && (((uintptr_t)x) & 15) == 0 &&
                     ((((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0 &&
                     (!out16 || ((ldo16 & 3) == 0 && (((uintptr_t)out16) & 7) == 0)) &&
                     (!out32 || ((ldo32 & 3) == 0 && (((uintptr_t)out32) & 15) == 0));
    if (vec) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      const int nc = D >> 8;
      f4 v[4];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = (k < nc) ? *reinterpret_cast<const f4 *>(reinterpret_cast<const float *>(xr) + 4 * lane + 256 * k)
                        : f4{0.f, 0.f, 0.f, 0.f};
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
      }
      const float mean = wave_sum(s) / (float)D;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < nc) {
          const f4 d = v[k] - mean;
          q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
      const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < nc) {
          const int i = 4 * lane + 256 * k;
          f4 y = (v[k] - mean) * rstd * *reinterpret_cast<const f4 *>(gamma + i) + *reinterpret_cast<const f4 *>(beta + i);
          if (relu) y = f4{fmaxf(y.x, 0.f), fmaxf(y.y, 0.f), fmaxf(y.z, 0.f), fmaxf(y.w, 0.f)};
          if (out16) {
            half4 o;
            o[0] = (h16)y.x; o[1] = (h16)y.y; o[2] = (h16)y.z; o[3] = (h16)y.w;
            *reinterpret_cast<half4 *>(out16 + (int64_t)row * ldo16 + i) = o;
          }
          if (out32) *reinterpret_cast<f4 *>(out32 + (int64_t)row * ldo32 + i) = y;
        }
      return;
    }
  }
  if (D <= 1024) {   // the row is read ONCE: 16 elements per lane stay in registers for both moments
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = lane + 64 * k;
      v[k] = i < D ? (float)xr[i] : 0.f;
      s += v[k];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float d = (lane + 64 * k < D) ? v[k] - mean : 0.f;
      q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = lane + 64 * k;
      if (i < D) {
        float y = (v[k] - mean) * rstd * gamma[i] + beta[i];
        if (relu) y = fmaxf(y, 0.f);
        if (out16) out16[(int64_t)row * ldo16 + i] = (h16)y;
        if (out32) out32[(int64_t)row * ldo32 + i] = y;
      }
    }
    return;
  }
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += (float)xr[i];
  s = wave_sum(s);
  const float mean = s / (float)D;
  float v = 0.f;
  for (int i = lane; i < D; i += 64) {
    const float d = (float)xr[i] - mean;
    v += d * d;
  }
  v = wave_sum(v);
  const float rstd = rsqrtf(v / (float)D + eps);
  for (int i = lane; i < D; i += 64) {
    float y = ((float)xr[i] - mean) * rstd * gamma[i] + beta[i];
    if (relu) y = fmaxf(y, 0.f);
    if (out16) out16[(int64_t)row * ldo16 + i] = (h16)y;
    if (out32) out32[(int64_t)row * ldo32 + i] = y;
  }
}

// ---- V^T for the attention kernel: Vt[h][d][key], keys zero-padded to n_pad -----------------
__global__ void __launch_bounds__(256)
transpose_v_kernel(int n_tok, int n_pad, int heads, const h16 *__restrict__ qkv, int ld,
                   h16 *__restrict__ vt) {
  __shared__ h16 tile[64][66];
  const int h = blockIdx.y, t0 = blockIdx.x * 64;
  const int D = heads * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int t = i >> 6, d = i & 63;
    tile[t][d] = (t0 + t < n_tok) ? qkv[(int64_t)(t0 + t) * ld + 2 * D + h * 64 + d] : (h16)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int d = i >> 6, t = i & 63;
    if (t0 + t < n_pad) vt[((int64_t)h * 64 + d) * n_pad + t0 + t] = tile[t][d];
  }
}

// ---- attention, head_dim 64 ---------------------------------------------------------------
// grid (ceil(n_tok / 128), heads), 4 waves; wave w owns queries q0 + 32 w .. + 31 (one per
// lane column). Per 32-key tile: S^T = K Q^T (4 MFMA), online softmax in registers, then
// O^T += V^T P (4 MFMA) with P taken straight from the S^T accumulator registers: after
// v_mfma_f32_32x32x16 register e of lane-half h is key row (e&3) + 8(e>>2) + 4h, so the B
// fragment of k-step s is registers 8s..8s+7 and element j stands for key 16s + 8(j>>2) + 4h +
// (j&3); the V^T fragment (A operand) is read from LDS in that same key order.
constexpr int KT = 32;        // keys per tile
constexpr int LDK = 72;       // sK row stride (halves)
constexpr int LDV = 36;       // sVt row stride (halves): 32 keys + 4 pad = 18 dwords, so the 32 rows of a
                              // half-wave's ds_read_b64 start in 32 distinct bank pairs (40 halves = 20
                              // dwords put rows r and r + 16 on the same banks: SQ_LDS_BANK_CONFLICT 36 %)

// SPLIT groups of 4 waves share a workgroup's 128 queries and take every SPLIT-th key tile each
// (flash-decoding inside the workgroup): the sequence of this network is short -- 3349 tokens x
// 16 heads are 1675 query waves for 1024 SIMDs, and a single wave per SIMD issues a VALU instruction
// only every ~6 cycles (profiles/r02_valu_rate.jsonl), which is what bounds the softmax stream. The
// groups' partial (O, m, l) are merged through LDS at the end.
template <int SPLIT>
__global__ void __launch_bounds__(256 * SPLIT)
attention_kernel(int n_tok, int n_pad, int heads, const h16 *__restrict__ qkv, int ld,
                 const h16 *__restrict__ vt, float scale_log2e, h16 *__restrict__ out, int ldo) {
  constexpr int TILE_H = KT * LDK + 64 * LDV;                 // halves of one staged (K, Vt) tile pair
  constexpr int STAGE_B = 2 * SPLIT * TILE_H * 2, MERGE_B = 4 * 34 * 64 * 4;
  __shared__ __attribute__((aligned(16))) char smem_raw[STAGE_B > MERGE_B ? STAGE_B : MERGE_B];
  h16 *smem = reinterpret_cast<h16 *>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, w4 = wave & 3, gt = tid & 255;
  const int lr = lane & 31, lh = lane >> 5;
  const int h = blockIdx.y, D = heads * 64;
  const int q = blockIdx.x * 128 + w4 * 32 + lr;
  const int qc = min(q, n_tok - 1);

  half8 qf[4];   // B operand of S^T: B[k = d][col = query]: lane (r, h) holds Q[q][16s + 8h + j]
#pragma unroll
  for (int s = 0; s < 4; ++s)
    qf[s] = *reinterpret_cast<const half8 *>(qkv + (int64_t)qc * ld + h * 64 + s * 16 + lh * 8);

  // staging (per group): K tile = 32 keys x 64 d = 256 pieces of 16 B; Vt tile = 64 d x 32 keys
  const int k_row = gt >> 3, k_pc = (gt & 7) * 8;
  const int v_row = gt >> 2, v_pc = (gt & 3) * 8;
  const h16 *gk = qkv + D + h * 64 + k_pc;
  const h16 *gv = vt + ((int64_t)h * 64 + v_row) * n_pad + v_pc;
  uint4 rk, rv;
  auto gload = [&](int key0) {   // (tiles past the sequence are clamped: read, never used)
    rk = *reinterpret_cast<const uint4 *>(gk + (int64_t)min(key0 + k_row, n_tok - 1) * ld);
    rv = *reinterpret_cast<const uint4 *>(gv + min(key0, n_pad - KT));
  };
  auto sstore = [&](int buf) {
    h16 *sK = smem + (buf * SPLIT + grp) * TILE_H, *sV = sK + KT * LDK;
    *reinterpret_cast<uint4 *>(&sK[k_row * LDK + k_pc]) = rk;
    // (Vt rows are 72 bytes apart: 8-byte aligned pieces)
    *reinterpret_cast<uint2 *>(&sV[v_row * LDV + v_pc]) = make_uint2(rv.x, rv.y);
    *reinterpret_cast<uint2 *>(&sV[v_row * LDV + v_pc + 4]) = make_uint2(rv.z, rv.w);
  };

  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int n_tiles = (n_tok + KT - 1) / KT;
  const int n_iter = (n_tiles + SPLIT - 1) / SPLIT;
  gload(grp * KT);
  sstore(0);
  __syncthreads();
  int buf = 0;
  for (int it = 0; it < n_iter; ++it) {
    const int key0 = (it * SPLIT + grp) * KT;
    if (it + 1 < n_iter) gload(key0 + SPLIT * KT);
    if (key0 < n_tok) {   // (wave-uniform; a group's surplus tile at the end is skipped)
      const h16 *sK = smem + (buf * SPLIT + grp) * TILE_H, *sV = sK + KT * LDK;
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const half8 kf = *reinterpret_cast<const half8 *>(&sK[lr * LDK + s * 16 + lh * 8]);
        st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], st, 0, 0, 0);
      }
      if (key0 + KT > n_tok) {   // last tile: keys beyond the sequence take no weight
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (key0 + (r & 3) + 8 * (r >> 2) + 4 * lh >= n_tok) st[r] = -1e30f;
      }
      float m_loc = st[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) m_loc = fmaxf(m_loc, st[r]);
      m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 32, 64));
      // Deferred maximum: the running reference m_run only moves when some query's tile maximum
      // exceeds it by more than 2^8 in probability units -- until then the weights are taken
      // relative to the stale reference (p <= 256, exact in the fp32 sums, 11-bit mantissa in the
      // fp16 P operand either way) and the 32 accumulator registers are NOT rescaled. The final
      // division by l_run cancels the reference, so the result is the same softmax.
      if (__any((m_loc - m_run) * scale_log2e > 8.0f)) {
        const float m_new = fmaxf(m_run, m_loc);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
        l_run *= alpha;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      }
      const float mc = m_run * scale_log2e;
      float l_loc = 0.f;
      half8 pf[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(st[r], scale_log2e, -mc));
        l_loc += pv;
        pf[r >> 3][r & 7] = (h16)pv;
      }
      l_loc += __shfl_xor(l_loc, 32, 64);
      l_run += l_loc;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // A[row = d][k = 8h + j] = Vt[d][16s + 8(j>>2) + 4h + (j&3)]: two 8-byte reads
          const h16 *vrow = &sV[(t * 32 + lr) * LDV + s * 16 + lh * 4];
          const half4 lo = *reinterpret_cast<const half4 *>(vrow);
          const half4 hi = *reinterpret_cast<const half4 *>(vrow + 8);
          half8 vf;
          vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
          vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s], o[t], 0, 0, 0);
        }
      }
    }
    if (it + 1 < n_iter) {
      sstore(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }
  // merge the groups' partial results into group 0, one group at a time through LDS
  // (float mb[4 waves][34][64 lanes]: 32 accumulators, m, l)
  if (SPLIT > 1) {
    float *mb = reinterpret_cast<float *>(smem_raw) + w4 * 34 * 64 + lane;
#pragma unroll 1
    for (int g = 1; g < SPLIT; ++g) {
      __syncthreads();   // staging reads / the previous round's merge reads are done
      if (grp == g) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) mb[(t * 16 + r) * 64] = o[t][r];
        mb[32 * 64] = m_run;
        mb[33 * 64] = l_run;
      }
      __syncthreads();
      if (grp == 0) {
        const float m_o = mb[32 * 64], l_o = mb[33 * 64];
        const float m_new = fmaxf(m_run, m_o);
        const float a0 = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
        const float a1 = __builtin_amdgcn_exp2f((m_o - m_new) * scale_log2e);
        l_run = l_run * a0 + l_o * a1;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[t][r] = o[t][r] * a0 + mb[(t * 16 + r) * 64] * a1;
      }
    }
  }
  if (grp == 0 && q < n_tok) {
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        out[(int64_t)q * ldo + h * 64 + d] = (h16)(o[t][r] * inv);
      }
  }
}

// ---- patch / convolution rows --------------------------------------------------------------
// Patch embedding (Conv2d k = stride = P on an NCHW fp32 image): row = patch, column
// c*P*P + ky*P + kx (the flattening of the conv weight [D, 3, P, P]); columns >= 3*P*P are zero.
__global__ void __launch_bounds__(256)
patch_rows_kernel(int H, int W, int P, int K_pad, const float *__restrict__ img, h16 *__restrict__ rows) {
  const int gw = W / P;
  const int64_t total = (int64_t)(H / P) * gw * K_pad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % K_pad);
    const int patch = (int)(i / K_pad);
    float v = 0.f;
    if (col < 3 * P * P) {
      const int c = col / (P * P), rem = col - c * P * P, ky = rem / P, kx = rem - ky * P;
      const int py = patch / gw, px = patch - py * gw;
      v = img[((int64_t)c * H + py * P + ky) * W + px * P + kx];
    }
    rows[i] = (h16)v;
  }
}

// im2col for a KSxKS convolution over an NHWC fp16 map: row = output pixel, column
// (ky*KS + kx)*C + c; zero outside the image and in the K padding.
__global__ void __launch_bounds__(256)
im2col_kernel(int H, int W, int C, int ldi, int KS, int stride, int pad, int Ho, int Wo, int K_pad,
              const h16 *__restrict__ in, h16 *__restrict__ rows, int relu) {
  const int64_t total = (int64_t)Ho * Wo * (K_pad / 8);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int col8 = (int)(i % (K_pad / 8)) * 8;
    const int pix = (int)(i / (K_pad / 8));
    const int oy = pix / Wo, ox = pix - oy * Wo;
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (h16)0;
    if ((C & 7) == 0) {   // the 8 columns share one tap: one 16-byte read
      if (col8 < KS * KS * C) {
        const int tap = col8 / C, c = col8 - tap * C, ky = tap / KS, kx = tap - ky * KS;
        const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W)
          v = *reinterpret_cast<const half8 *>(in + ((int64_t)iy * W + ix) * ldi + c);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = col8 + e;
        if (col < KS * KS * C) {
          const int tap = col / C, c = col - tap * C, ky = tap / KS, kx = tap - ky * KS;
          const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) v[e] = in[((int64_t)iy * W + ix) * ldi + c];
        }
      }
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] > (h16)0 ? v[e] : (h16)0;
    }
    *reinterpret_cast<half8 *>(rows + (int64_t)pix * K_pad + col8) = v;
  }
}

// ---- element-wise pieces of the decoder (NHWC fp16 maps, `ld` = channel stride of a pixel) ----
// out = resize(in) [+ add]; mode 0 nearest (floor(dst * scale)), 1 bilinear align_corners=True,
// 2 bilinear align_corners=False (half-pixel centres, clamped)
__global__ void __launch_bounds__(256)
resize_kernel(int Hi, int Wi, int C, const h16 *__restrict__ in, int ldi, int Ho, int Wo,
              h16 *__restrict__ out, int ldo, int mode, float sy, float sx) {
  const int64_t total = (int64_t)Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int pix = (int)(i / C);
    const int oy = pix / Wo, ox = pix - oy * Wo;
    float v;
    if (mode == 0) {
      const int iy = min((int)floorf(oy * sy), Hi - 1), ix = min((int)floorf(ox * sx), Wi - 1);
      v = (float)in[((int64_t)iy * Wi + ix) * ldi + c];
    } else {
      float fy, fx;
      if (mode == 1) {
        fy = oy * sy;
        fx = ox * sx;
      } else {
        fy = fmaxf((oy + 0.5f) * sy - 0.5f, 0.f);
        fx = fmaxf((ox + 0.5f) * sx - 0.5f, 0.f);
      }
      const int y0 = min((int)fy, Hi - 1), x0 = min((int)fx, Wi - 1);
      const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
      const float wy = fy - (float)y0, wx = fx - (float)x0;
      const float a = (float)in[((int64_t)y0 * Wi + x0) * ldi + c], b = (float)in[((int64_t)y0 * Wi + x1) * ldi + c];
      const float d = (float)in[((int64_t)y1 * Wi + x0) * ldi + c], e = (float)in[((int64_t)y1 * Wi + x1) * ldi + c];
      v = (1.f - wy) * ((1.f - wx) * a + wx * b) + wy * ((1.f - wx) * d + wx * e);
    }
    out[(int64_t)pix * ldo + c] = (h16)v;
  }
}

// F.avg_pool2d(x, 3, stride 2, padding 1), count_include_pad=True (divide by 9)
__global__ void __launch_bounds__(256)
avgpool3s2_kernel(int Hi, int Wi, int C, const h16 *__restrict__ in, int ldi, int Ho, int Wo,
                  h16 *__restrict__ out, int ldo) {
  const int64_t total = (int64_t)Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int pix = (int)(i / C);
    const int oy = pix / Wo, ox = pix - oy * Wo;
    float s = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int iy = 2 * oy + dy, ix = 2 * ox + dx;
        if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) s += (float)in[((int64_t)iy * Wi + ix) * ldi + c];
      }
    out[(int64_t)pix * ldo + c] = (h16)(s * (1.0f / 9.0f));
  }
}

// generic strided copy / add / scale of channel slices: out[p, co + c] = a*in[p, ci + c] (+ out)
__global__ void __launch_bounds__(256)
slice_kernel(int64_t P, int C, const h16 *__restrict__ in, int ldi, h16 *__restrict__ out, int ldo,
             float a, int accumulate, int act) {
  const int64_t total = P * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t pix = i / C;
    float v = a * (float)in[pix * ldi + c];
    if (accumulate) v += (float)out[pix * ldo + c];
    out[pix * ldo + c] = (h16)apply_act(v, act);
  }
}

// SwiGLU gate of the ViT-giant FFN (ViT_DINO_reg.py SwiGLUFFN.forward :335-345):
// out[p, c] = silu(x12[p, c]) * x12[p, h + c], c < h (x12 = w12(x), two halves of 2h columns)
__global__ void __launch_bounds__(256)
swiglu_kernel(int64_t P, int h, const h16 *__restrict__ x12, int ldx, h16 *__restrict__ out, int ldo) {
  const int64_t total = P * (h / 2);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (h / 2)) * 2;
    const int64_t p = i / (h / 2);
    const h16 *row = x12 + p * ldx;
    const float a0 = (float)row[c], a1 = (float)row[c + 1], b0 = (float)row[h + c], b1 = (float)row[h + c + 1];
    out[p * ldo + c] = (h16)(a0 / (1.0f + __expf(-a0)) * b0);
    out[p * ldo + c + 1] = (h16)(a1 / (1.0f + __expf(-a1)) * b1);
  }
}

// ConvGRU gate algebra (RAFTDepthNormalDPTDecoder5.py ConvGRU.forward):
//   stage 0: z = sigmoid(zr[:, 0:C] + cz), r = sigmoid(zr[:, C:2C] + cr); writes z and r*h
//   stage 1: q = tanh(qin + cq); h = (1 - z) h + z q
__global__ void __launch_bounds__(256)
gru_gate_kernel(int64_t P, int C, int stage, const h16 *__restrict__ zr, int ldzr,
                const h16 *__restrict__ ctx, int ldc, h16 *__restrict__ h, int ldh,
                h16 *__restrict__ z, int ldz, h16 *__restrict__ rh, int ldrh) {
  const int64_t total = P * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t p = i / C;
    if (stage == 0) {
      const float zz = 1.f / (1.f + __expf(-((float)zr[p * ldzr + c] + (float)ctx[p * ldc + c])));
      const float rr = 1.f / (1.f + __expf(-((float)zr[p * ldzr + C + c] + (float)ctx[p * ldc + C + c])));
      z[p * ldz + c] = (h16)zz;
      rh[p * ldrh + c] = (h16)(rr * (float)h[p * ldh + c]);
    } else {
      const float qq = tanhf((float)zr[p * ldzr + c] + (float)ctx[p * ldc + 2 * C + c]);
      const float zz = (float)z[p * ldz + c], hh = (float)h[p * ldh + c];
      h[p * ldh + c] = (h16)((1.f - zz) * hh + zz * qq);
    }
  }
}

// depth head: softmax over the `bins` logits of a pixel, expectation over log-spaced depth bins,
// clamp to [min, max], then (d - max) / regress_scale (regress_depth, decoder :806-838)
__global__ void __launch_bounds__(256)
depth_expectation_kernel(int64_t P, int bins, const h16 *__restrict__ logits, int ld, float log_min,
                         float log_max, float min_val, float max_val, float regress_scale,
                         float *__restrict__ out, int ldo) {
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (p >= P) return;
  const h16 *l = logits + p * ld;
  float m = -1e30f;
  for (int i = lane; i < bins; i += 64) m = fmaxf(m, (float)l[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  float s = 0.f, e = 0.f;
  for (int i = lane; i < bins; i += 64) {
    const float w = __expf((float)l[i] - m);
    const float depth = __expf(log_min + (log_max - log_min) * (float)i / (float)(bins - 1));
    s += w;
    e += w * depth;
  }
  s = wave_sum(s);
  e = wave_sum(e);
  if (lane == 0) {
    float d = e / s;
    d = fminf(fmaxf(d, min_val), max_val);
    out[p * ldo] = (d - max_val) / regress_scale;
  }
}

// norm_normalize (decoder :252-258): xyz / (|xyz| + 1e-10), kappa = elu(k) + 1 + 0.01
__device__ __forceinline__ void norm_normalize4(float &x, float &y, float &z, float &k) {
  const float n = sqrtf(x * x + y * y + z * z) + 1e-10f;
  x /= n;
  y /= n;
  z /= n;
  k = (k > 0.f ? k : (__expf(k) - 1.f)) + 1.0f + 0.01f;
}

__global__ void __launch_bounds__(256)
normal_head_kernel(int64_t P, const h16 *__restrict__ nrm, int ldn, const h16 *__restrict__ conf,
                   int ldc, float *__restrict__ out, int ldo) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float x = (float)nrm[p * ldn], y = (float)nrm[p * ldn + 1], z = (float)nrm[p * ldn + 2];
  float k = (float)conf[p * ldc];
  norm_normalize4(x, y, z, k);
  out[p * ldo] = x;
  out[p * ldo + 1] = y;
  out[p * ldo + 2] = z;
  out[p * ldo + 3] = k;
}

// Convex upsampling of the 6-channel flow (upsample_flow, decoder :870-884) fused with the
// output heads (:985-987): for low-res pixel (y, x) and sub-position (i, j) of the FxF cell,
// weights = softmax over the 9 taps of mask[(tap*F + i)*F + j], value = sum_tap w * flow[3x3
// neighbour, zero padded]. Writes depth = clamp(v0 * regress_scale + max), confidence = v1,
// normal = norm_normalize(v2..5) as [1,*,H*F,W*F] planes.
__global__ void __launch_bounds__(256)
convex_upsample_kernel(int H, int W, int F, const float *__restrict__ flow /* [H*W,6] */,
                       const h16 *__restrict__ mask, int ldm, float min_val, float max_val,
                       float regress_scale, float *__restrict__ depth, float *__restrict__ conf,
                       float *__restrict__ normal /* [4, H*F, W*F] */) {
  const int64_t total = (int64_t)H * W * F * F;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int sub = (int)(i % (F * F));
  const int pix = (int)(i / (F * F));
  const int si = sub / F, sj = sub - si * F;
  const int y = pix / W, x = pix - y * W;
  float w[9], m = -1e30f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    w[t] = (float)mask[(int64_t)pix * ldm + (t * F + si) * F + sj];
    m = fmaxf(m, w[t]);
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    w[t] = __expf(w[t] - m);
    s += w[t];
  }
  float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
    const float ww = w[t] / s;
    const float *f = flow + ((int64_t)yy * W + xx) * 6;
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] += ww * f[c];
  }
  const int HF = H * F, WF = W * F;
  const int64_t o = (int64_t)(y * F + si) * WF + x * F + sj;
  float d = v[0] * regress_scale + max_val;
  depth[o] = fminf(fmaxf(d, min_val), max_val);
  conf[o] = v[1];
  norm_normalize4(v[2], v[3], v[4], v[5]);
  const int64_t plane = (int64_t)HF * WF;
  normal[o] = v[2];
  normal[plane + o] = v[3];
  normal[2 * plane + o] = v[4];
  normal[3 * plane + o] = v[5];
}

// fp32 <-> fp16 row conversions with strides (token rows, flow maps)
__global__ void __launch_bounds__(256)
cvt_kernel(int64_t P, int C, const float *__restrict__ in, int ldi, h16 *__restrict__ out, int ldo) {
  const int64_t total = P * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t p = i / C;
    out[p * ldo + c] = (h16)in[p * ldi + c];
  }
}

static inline unsigned grid_for(int64_t total) {
  const int64_t b = ceil_div64(total, 256);
  return (unsigned)(b < 65536 ? b : 65536);
}

}  // namespace dn
}  // namespace gsr

using namespace gsr::dn;

#ifdef GSR_GEMM_TIMELINE
extern "C" int gsr_debug_set_gemm_timeline(void *buf) {
  unsigned long long *p = (unsigned long long *)buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(gsr::dn::g_gemm_timeline), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

template <bool CONV>
static int launch_gemm(const GemmArgs &p, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  const int rows = gsr::ceil_div(p.M, gsr::dn::BM);
  // Tile width by a wave-quantisation model of the 256 CUs: 2 resident 128x128 workgroups per CU
  // (64 KB of LDS each) or 3 of 128x64 (48 KB), a 128x64 workgroup taking ~0.7 of the time of a
  // 128x128 one; the launch runs ceil(workgroups / resident slots) rounds. Measured at 3349 tokens:
  // N = 1024 (216 -> 432 workgroups) 26.6 -> 18.5 us at K = 1024 and 59 -> 48 us at K = 4096,
  // N = 3072 (648 = 2 rounds -> 1296) 53 -> 45 us; the 1/7-resolution convolution (420 workgroups, one
  // round) stays wide: 86 us against 116 us narrow.
  const int b128 = rows * gsr::ceil_div(p.N, 128), b64 = rows * gsr::ceil_div(p.N, 64);
  // (only where the rounding matters, up to two rounds: 8192^3 is 1390 us wide, 1560 us narrow)
  // (N <= 64: one column tile either way, and the 64-wide one does half the MFMA work of the 128-wide --
  // Metric3D-small's 48-channel maps)
  static const int wide_small = gsr::gsr_knob_int("GSR_DN_GEMM_WIDE_SMALL_N", 0);   // bench knob
  const bool narrow = (p.N <= 64 && !wide_small) ||
                      (p.N > 64 && b128 <= 1024 && 0.7 * gsr::ceil_div(b64, 768) < 1.0 * gsr::ceil_div(b128, 512));
  // The 256x256 eight-phase core wherever its grid occupies the chip: from 112 workgroups (one round on
  // 256 CUs) up. Same box, plain fp16 GEMM, TFLOP/s (profiles/r03_depthnet_gemm.md): 3349x3072x1024
  // 459 -> 679, 3349x4096x1024 495 -> 860, 40964x256x2304 582 -> 811, 40964x512x2880 666 -> 908,
  // 4096^3 943 -> 1279; with 56-106 workgroups (N = 1024 at 3349 rows, the 1/7 and 1/14 maps) the
  // 128-row tiles below stay ahead. GSR_DN_GEMM_CORE=1 / 4 forces a core (bench knob).
  static const int force_core = gsr::gsr_knob_int("GSR_DN_GEMM_CORE", 0);
  const int b256 = gsr::ceil_div(p.M, 256) * gsr::ceil_div(p.N, 256);
  // (gemm8p addresses its operands with 32-bit byte offsets)
  const bool fits32 = (int64_t)p.M * p.lda * 2 < (1ll << 32) && (int64_t)p.N * p.K * 2 < (1ll << 32);
  // (only where the 256-wide tiles are at least 7/8 full: N = 128 would leave half of every tile idle -- 40964 x
  // 128 x 2880 takes 71 us there, 54 us on the 128 x 64 tiles --, N = 384 a quarter: 101 against 69 us)
  const bool tiles_full = (int64_t)p.N * 8 >= (int64_t)gsr::ceil_div(p.N, 256) * 256 * 7;
  if (fits32 && (force_core == 4 || (force_core == 0 && p.K >= 256 && b256 >= 112 && tiles_full))) {
    const dim3 grid((unsigned)gsr::ceil_div(p.N, 256), (unsigned)gsr::ceil_div(p.M, 256));
#define GSR_GEMM4(A) hipLaunchKernelGGL((gemm8p_kernel<A, CONV>), grid, dim3(512), 0, st, p)
    switch (p.act) {
      case ACT_GELU: GSR_GEMM4(ACT_GELU); break;
      case ACT_RELU: GSR_GEMM4(ACT_RELU); break;
      case ACT_SIGMOID: GSR_GEMM4(ACT_SIGMOID); break;
      case ACT_TANH: GSR_GEMM4(ACT_TANH); break;
      default: GSR_GEMM4(ACT_NONE); break;
    }
#undef GSR_GEMM4
    GSR_CHECK_LAUNCH("dn_gemm8p");
    return GSR_OK;
  }
#define GSR_GEMM(A, BNT_)                                                                             \
  hipLaunchKernelGGL((gemm_kernel<A, CONV, BNT_>), dim3((unsigned)gsr::ceil_div(p.N, BNT_), (unsigned)rows), \
                     dim3(256), 0, st, p)
#define GSR_GEMM_ACT(BNT_)                         \
  switch (p.act) {                                 \
    case ACT_GELU: GSR_GEMM(ACT_GELU, BNT_); break;       \
    case ACT_RELU: GSR_GEMM(ACT_RELU, BNT_); break;       \
    case ACT_SIGMOID: GSR_GEMM(ACT_SIGMOID, BNT_); break; \
    case ACT_TANH: GSR_GEMM(ACT_TANH, BNT_); break;       \
    default: GSR_GEMM(ACT_NONE, BNT_); break;             \
  }
  // few 128x64 workgroups: 64x64 tiles (twice the workgroups, each walks half the rows): the 1/14-resolution
  // convolutions 33 -> 21 us each
  static const int no_small = gsr::gsr_knob_int("GSR_DN_GEMM_NO64", 0);   // bench knobs
  static const int max64 = gsr::gsr_knob_int("GSR_DN_GEMM_64_MAX", 450);
  // (measured per shape, tools/depthnet_shapes.py: up to 450 workgroups of 128 x 64 the smaller tiles win or tie
  // -- 3349 x 1024 x 1024 31 -> 24 us, 13376 x 256 x 3456 59 -> 55 us -- except at K = 9216: 140 -> 148 us)
  // (and from 256 workgroups up only to K = 3584: 3349 x 1024 x 4096 takes 48.6 us on 128 x 64, 51.7 us on 64 x 64)
  if (narrow && !no_small && b64 < max64 && p.K >= 512 && p.K <= (b64 < 256 ? 4608 : 3584)) {
    const dim3 grid64((unsigned)gsr::ceil_div(p.N, 64), (unsigned)gsr::ceil_div(p.M, 64));
#define GSR_GEMM64(A) hipLaunchKernelGGL((gemm_kernel<A, CONV, 64, 64>), grid64, dim3(256), 0, st, p)
    switch (p.act) {
      case ACT_GELU: GSR_GEMM64(ACT_GELU); break;
      case ACT_RELU: GSR_GEMM64(ACT_RELU); break;
      case ACT_SIGMOID: GSR_GEMM64(ACT_SIGMOID); break;
      case ACT_TANH: GSR_GEMM64(ACT_TANH); break;
      default: GSR_GEMM64(ACT_NONE); break;
    }
#undef GSR_GEMM64
    GSR_CHECK_LAUNCH("dn_gemm64");
    return GSR_OK;
  }
  if (narrow) {
    GSR_GEMM_ACT(64)
  } else {
    GSR_GEMM_ACT(128)
  }
#undef GSR_GEMM_ACT
#undef GSR_GEMM
  GSR_CHECK_LAUNCH("dn_gemm");
  return GSR_OK;
}

extern "C" int gsr_dn_gemm(int M, int N, int K, const void *A, int lda, const void *W,
                           const float *bias, int act, const float *gamma, const float *residual,
                           int ldr, const void *residual16, int ldr16, void *out16, int ldo16,
                           float *out32, int ldo32, int out16_pad_to, void *stream) {
  GSR_REQUIRE(M >= 0 && N >= 0 && K > 0 && K % gsr::dn::BK == 0, "dn_gemm: bad sizes M=%d N=%d K=%d (K %% 64)", M, N, K);
  if (M == 0 || N == 0) return GSR_OK;
  GSR_REQUIRE(A && W && (out16 || out32), "dn_gemm: null pointer");
  GSR_REQUIRE(lda >= K && (lda % 8) == 0, "dn_gemm: lda %d (>= K, multiple of 8 halves)", lda);
  GSR_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, "dn_gemm: A and W must be 16-byte aligned");
  GSR_REQUIRE(act >= 0 && act <= 4, "dn_gemm: act %d", act);
  GemmArgs p = {};
  p.M = M; p.N = N; p.K = K;
  p.A = (const h16 *)A; p.lda = lda; p.W = (const h16 *)W;
  p.bias = bias; p.gamma = gamma; p.residual = residual; p.ldr = ldr;
  p.residual16 = (const h16 *)residual16; p.ldr16 = ldr16;
  p.out16 = (h16 *)out16; p.ldo16 = ldo16; p.out32 = out32; p.ldo32 = ldo32;
  p.act = act;
  GSR_REQUIRE(out16_pad_to <= ldo16 && (out16_pad_to <= N || out16) && out16_pad_to <= gsr::ceil_div(N, 64) * 64,
              "dn_gemm: out16_pad_to %d (N %d, ldo16 %d)", out16_pad_to, N, ldo16);
  p.pad_to = out16_pad_to;
  return launch_gemm<false>(p, stream);
}

extern "C" int gsr_dn_conv_gemm(int H, int Wd, int C, const void *in, int ldi, int KS, int N, int K_pad,
                                const void *W, const float *bias, int act, const void *residual16,
                                int ldr16, void *out16, int ldo16, const void *zero_page, int out16_pad_to,
                                void *stream) {
  GSR_REQUIRE(H > 0 && Wd > 0 && C > 0 && C % 64 == 0 && (KS == 1 || KS == 3) && N > 0 &&
                  K_pad >= KS * KS * C && K_pad % gsr::dn::BK == 0,
              "dn_conv_gemm: bad sizes H=%d W=%d C=%d (C %% 64) KS=%d K_pad=%d", H, Wd, C, KS, K_pad);
  GSR_REQUIRE(in && W && out16 && zero_page, "dn_conv_gemm: null pointer");
  GSR_REQUIRE((ldi % 8) == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)W & 15) == 0 &&
                  ((uintptr_t)zero_page & 15) == 0,
              "dn_conv_gemm: 16-byte alignment of the map, the weights and the zero page");
  GSR_REQUIRE(act >= 0 && act <= 4, "dn_conv_gemm: act %d", act);
  GemmArgs p = {};
  p.M = H * Wd; p.N = N; p.K = K_pad;
  p.A = (const h16 *)in; p.lda = ldi; p.W = (const h16 *)W;
  p.bias = bias;
  p.residual16 = (const h16 *)residual16; p.ldr16 = ldr16;
  p.out16 = (h16 *)out16; p.ldo16 = ldo16;
  p.act = act;
  p.cH = H; p.cW = Wd; p.cC = C; p.cKS = KS; p.cPad = KS / 2;
  p.zero_page = (const h16 *)zero_page;
  GSR_REQUIRE(out16_pad_to <= ldo16 && out16_pad_to <= gsr::ceil_div(N, 64) * 64,
              "dn_conv_gemm: out16_pad_to %d (N %d, ldo16 %d)", out16_pad_to, N, ldo16);
  p.pad_to = out16_pad_to;
  return launch_gemm<true>(p, stream);
}

// gsr_dn_conv_gemm over a virtual concatenation: channels [0, c_split) of the input are read from `first`
// (row stride ld_first), channels [c_split, C) from `in` (whose first c_split channels are never read). The
// ConvGRU's convolutions take [h | x] and [r * h | x] this way: no copy of the hidden state into the
// concatenated input (RAFTDepthNormalDPTDecoder5.py:318-330, torch.cat([h, x], dim=1)).
extern "C" int gsr_dn_conv_gemm2(int H, int Wd, int C, const void *in, int ldi, const void *first, int ld_first,
                                 int c_split, int KS, int N, int K_pad, const void *W, const float *bias, int act,
                                 const void *residual16, int ldr16, void *out16, int ldo16, const void *zero_page,
                                 int out16_pad_to, void *stream) {
  GSR_REQUIRE(H > 0 && Wd > 0 && C > 0 && C % 64 == 0 && (KS == 1 || KS == 3) && N > 0 &&
                  K_pad >= KS * KS * C && K_pad % gsr::dn::BK == 0 && c_split > 0 && c_split < C && c_split % 64 == 0,
              "dn_conv_gemm2: bad sizes H=%d W=%d C=%d (C %% 64) c_split=%d KS=%d K_pad=%d", H, Wd, C, c_split, KS, K_pad);
  GSR_REQUIRE(in && first && W && out16 && zero_page, "dn_conv_gemm2: null pointer");
  GSR_REQUIRE((ldi % 8) == 0 && (ld_first % 8) == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)first & 15) == 0 &&
                  ((uintptr_t)W & 15) == 0 && ((uintptr_t)zero_page & 15) == 0,
              "dn_conv_gemm2: 16-byte alignment of the maps, the weights and the zero page");
  GSR_REQUIRE(act >= 0 && act <= 4, "dn_conv_gemm2: act %d", act);
  GemmArgs p = {};
  p.M = H * Wd; p.N = N; p.K = K_pad;
  p.A = (const h16 *)in; p.lda = ldi; p.W = (const h16 *)W;
  p.A2 = (const h16 *)first; p.lda2 = ld_first; p.c_split = c_split;
  p.bias = bias;
  p.residual16 = (const h16 *)residual16; p.ldr16 = ldr16;
  p.out16 = (h16 *)out16; p.ldo16 = ldo16;
  p.act = act;
  p.cH = H; p.cW = Wd; p.cC = C; p.cKS = KS; p.cPad = KS / 2;
  p.zero_page = (const h16 *)zero_page;
  GSR_REQUIRE(out16_pad_to <= ldo16 && out16_pad_to <= gsr::ceil_div(N, 64) * 64,
              "dn_conv_gemm2: out16_pad_to %d (N %d, ldo16 %d)", out16_pad_to, N, ldo16);
  p.pad_to = out16_pad_to;
  return launch_gemm<true>(p, stream);
}

extern "C" int gsr_dn_layernorm(int M, int D, const void *x, int ldx, int x_is_f16, const float *gamma,
                                const float *beta, float eps, void *out16, int ldo16, float *out32,
                                int ldo32, int relu, void *stream) {
  GSR_REQUIRE(M >= 0 && D > 0, "dn_layernorm: bad sizes");
  if (M == 0) return GSR_OK;
  GSR_REQUIRE(x && gamma && beta && (out16 || out32), "dn_layernorm: null pointer");
  dim3 grid((unsigned)gsr::ceil_div(M, 4));
  if (x_is_f16)
    hipLaunchKernelGGL(layernorm_kernel<h16>, grid, dim3(256), 0, (hipStream_t)stream, M, D,
                       (const h16 *)x, ldx, gamma, beta, eps, (h16 *)out16, ldo16, out32, ldo32, relu);
  else
    hipLaunchKernelGGL(layernorm_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, M, D,
                       (const float *)x, ldx, gamma, beta, eps, (h16 *)out16, ldo16, out32, ldo32, relu);
  GSR_CHECK_LAUNCH("dn_layernorm");
  return GSR_OK;
}

extern "C" int gsr_dn_attention(int n_tok, int n_pad, int heads, const void *qkv, int ld, void *vt_scratch,
                                float scale, void *out, int ldo, void *stream) {
  GSR_REQUIRE(n_tok > 0 && heads > 0 && n_pad >= n_tok && n_pad % 32 == 0,
              "dn_attention: bad sizes n_tok=%d n_pad=%d (multiple of 32)", n_tok, n_pad);
  GSR_REQUIRE(qkv && vt_scratch && out && ld >= 3 * heads * 64 && (ld % 8) == 0, "dn_attention: bad arguments");
  hipLaunchKernelGGL(transpose_v_kernel, dim3((unsigned)gsr::ceil_div(n_pad, 64), (unsigned)heads),
                     dim3(256), 0, (hipStream_t)stream, n_tok, n_pad, heads, (const h16 *)qkv, ld,
                     (h16 *)vt_scratch);
  // key split: enough waves for ~3 per SIMD (1024 SIMDs), at most 4 groups per workgroup
  const int q_waves = gsr::ceil_div(n_tok, 128) * 4 * heads;
  // (measured at 3349 tokens: 16 heads 133 / 93 / 104 us with 1 / 2 / 4 groups, 6 heads 116 / 68 / 55)
  const int split = (q_waves >= 3072 || n_tok <= 4 * 32) ? 1 : (q_waves >= 1536 ? 2 : 4);
  const dim3 grid((unsigned)gsr::ceil_div(n_tok, 128), (unsigned)heads);
#define GSR_ATT(S)                                                                                  \
  hipLaunchKernelGGL(attention_kernel<S>, grid, dim3(256 * S), 0, (hipStream_t)stream, n_tok, n_pad, \
                     heads, (const h16 *)qkv, ld, (const h16 *)vt_scratch,                           \
                     scale * 1.4426950408889634f, (h16 *)out, ldo)
  if (split == 1) GSR_ATT(1);
  else if (split == 2) GSR_ATT(2);
  else GSR_ATT(4);
#undef GSR_ATT
  GSR_CHECK_LAUNCH("dn_attention");
  return GSR_OK;
}

extern "C" int gsr_dn_patch_rows(int H, int W, int P, int K_pad, const float *img, void *rows, void *stream) {
  GSR_REQUIRE(H > 0 && W > 0 && P > 0 && H % P == 0 && W % P == 0 && K_pad >= 3 * P * P, "dn_patch_rows: bad sizes");
  GSR_REQUIRE(img && rows, "dn_patch_rows: null pointer");
  const int64_t total = (int64_t)(H / P) * (W / P) * K_pad;
  hipLaunchKernelGGL(patch_rows_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, H, W, P,
                     K_pad, img, (h16 *)rows);
  GSR_CHECK_LAUNCH("dn_patch_rows");
  return GSR_OK;
}

extern "C" int gsr_dn_im2col(int H, int W, int C, int ldi, int KS, int stride, int pad, int Ho, int Wo,
                             int K_pad, const void *in, void *rows, int relu, void *stream) {
  GSR_REQUIRE(H > 0 && W > 0 && C > 0 && KS > 0 && stride > 0 && Ho > 0 && Wo > 0 &&
                  K_pad >= KS * KS * C && K_pad % 8 == 0,
              "dn_im2col: bad sizes");
  GSR_REQUIRE(in && rows, "dn_im2col: null pointer");
  GSR_REQUIRE((C & 7) != 0 || (ldi & 7) == 0, "dn_im2col: ldi must be a multiple of 8 for vector reads");
  const int64_t total = (int64_t)Ho * Wo * (K_pad / 8);
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, H, W, C, ldi,
                     KS, stride, pad, Ho, Wo, K_pad, (const h16 *)in, (h16 *)rows, relu);
  GSR_CHECK_LAUNCH("dn_im2col");
  return GSR_OK;
}

extern "C" int gsr_dn_conv3_head(int H, int W, int C, const void *in, int ldi, int N, const void *Wt, int K_pad,
                                 const float *bias, float *out32, int ldo, void *stream) {
  GSR_REQUIRE(H > 0 && W > 0 && C > 0 && C % 8 == 0 && N >= 1 && N <= gsr::dn::HEAD_MAX_N && K_pad >= 9 * C && ldo >= N,
              "dn_conv3_head: bad sizes H=%d W=%d C=%d N=%d K_pad=%d ldo=%d", H, W, C, N, K_pad, ldo);
  GSR_REQUIRE(in && Wt && out32, "dn_conv3_head: null pointer");
  GSR_REQUIRE((ldi & 7) == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)Wt & 15) == 0 && (K_pad & 7) == 0,
              "dn_conv3_head: rows of the map and of the weights must be 16-byte aligned");
  const size_t lds = (size_t)N * 9 * C * sizeof(h16);
  GSR_REQUIRE(lds <= 64 * 1024, "dn_conv3_head: %d x 9 x %d weights do not fit the LDS staging", N, C);
  const int64_t P = (int64_t)H * W;
  int grid = (int)gsr::ceil_div64(P, 16 * 4);            // four pixel groups per wave and trip, ~4 trips
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
#define GSR_HEAD(N_) case N_: hipLaunchKernelGGL((conv3_head_kernel<N_>), dim3(grid), dim3(256), lds, (hipStream_t)stream, \
                                                 H, W, C, (const h16 *)in, ldi, (const h16 *)Wt, K_pad, bias, out32, ldo); break
  switch (N) {
    GSR_HEAD(1); GSR_HEAD(2); GSR_HEAD(3); GSR_HEAD(4); GSR_HEAD(5); GSR_HEAD(6); GSR_HEAD(7); GSR_HEAD(8);
  }
#undef GSR_HEAD
  GSR_CHECK_LAUNCH("dn_conv3_head");
  return GSR_OK;
}

extern "C" int gsr_dn_resize(int Hi, int Wi, int C, const void *in, int ldi, int Ho, int Wo, void *out,
                             int ldo, int mode, void *stream) {
  GSR_REQUIRE(Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && mode >= 0 && mode <= 2, "dn_resize: bad sizes");
  GSR_REQUIRE(in && out, "dn_resize: null pointer");
  float sy, sx;
  if (mode == 1) {
    sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
    sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  } else {
    sy = (float)Hi / (float)Ho;
    sx = (float)Wi / (float)Wo;
  }
  hipLaunchKernelGGL(resize_kernel, dim3(grid_for((int64_t)Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream,
                     Hi, Wi, C, (const h16 *)in, ldi, Ho, Wo, (h16 *)out, ldo, mode, sy, sx);
  GSR_CHECK_LAUNCH("dn_resize");
  return GSR_OK;
}

extern "C" int gsr_dn_avgpool3s2(int Hi, int Wi, int C, const void *in, int ldi, void *out, int ldo,
                                 void *stream) {
  GSR_REQUIRE(Hi > 0 && Wi > 0 && C > 0 && in && out, "dn_avgpool3s2: bad arguments");
  const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(avgpool3s2_kernel, dim3(grid_for((int64_t)Ho * Wo * C)), dim3(256), 0,
                     (hipStream_t)stream, Hi, Wi, C, (const h16 *)in, ldi, Ho, Wo, (h16 *)out, ldo);
  GSR_CHECK_LAUNCH("dn_avgpool3s2");
  return GSR_OK;
}

extern "C" int gsr_dn_slice(int64_t P, int C, const void *in, int ldi, void *out, int ldo, float a,
                            int accumulate, int act, void *stream) {
  GSR_REQUIRE(P >= 0 && C > 0 && act >= 0 && act <= 4, "dn_slice: bad sizes");
  if (P == 0) return GSR_OK;
  GSR_REQUIRE(in && out, "dn_slice: null pointer");
  hipLaunchKernelGGL(slice_kernel, dim3(grid_for(P * C)), dim3(256), 0, (hipStream_t)stream, P, C,
                     (const h16 *)in, ldi, (h16 *)out, ldo, a, accumulate, act);
  GSR_CHECK_LAUNCH("dn_slice");
  return GSR_OK;
}

extern "C" int gsr_dn_swiglu(int64_t P, int h, const void *x12, int ldx, void *out, int ldo, void *stream) {
  GSR_REQUIRE(P >= 0 && h > 0 && h % 2 == 0 && ldx >= 2 * h && ldo >= h, "dn_swiglu: bad sizes");
  if (P == 0) return GSR_OK;
  GSR_REQUIRE(x12 && out, "dn_swiglu: null pointer");
  hipLaunchKernelGGL(swiglu_kernel, dim3(grid_for(P * (h / 2))), dim3(256), 0, (hipStream_t)stream, P, h,
                     (const h16 *)x12, ldx, (h16 *)out, ldo);
  GSR_CHECK_LAUNCH("dn_swiglu");
  return GSR_OK;
}

extern "C" int gsr_dn_gru_gate(int64_t P, int C, int stage, const void *zr, int ldzr, const void *ctx,
                               int ldc, void *h, int ldh, void *z, int ldz, void *rh, int ldrh,
                               void *stream) {
  GSR_REQUIRE(P >= 0 && C > 0 && (stage == 0 || stage == 1), "dn_gru_gate: bad sizes");
  if (P == 0) return GSR_OK;
  GSR_REQUIRE(zr && ctx && h && z && (stage == 1 || rh), "dn_gru_gate: null pointer");
  hipLaunchKernelGGL(gru_gate_kernel, dim3(grid_for(P * C)), dim3(256), 0, (hipStream_t)stream, P, C, stage,
                     (const h16 *)zr, ldzr, (const h16 *)ctx, ldc, (h16 *)h, ldh, (h16 *)z, ldz,
                     (h16 *)rh, ldrh);
  GSR_CHECK_LAUNCH("dn_gru_gate");
  return GSR_OK;
}

extern "C" int gsr_dn_depth_expectation(int64_t P, int bins, const void *logits, int ld, float min_val,
                                        float max_val, float regress_scale, float *out, int ldo,
                                        void *stream) {
  GSR_REQUIRE(P >= 0 && bins > 1 && min_val > 0.f && max_val > min_val, "dn_depth_expectation: bad sizes");
  if (P == 0) return GSR_OK;
  GSR_REQUIRE(logits && out, "dn_depth_expectation: null pointer");
  hipLaunchKernelGGL(depth_expectation_kernel, dim3((unsigned)gsr::ceil_div64(P, 4)), dim3(256), 0,
                     (hipStream_t)stream, P, bins, (const h16 *)logits, ld, logf(min_val), logf(max_val),
                     min_val, max_val, regress_scale, out, ldo);
  GSR_CHECK_LAUNCH("dn_depth_expectation");
  return GSR_OK;
}

extern "C" int gsr_dn_normal_head(int64_t P, const void *nrm, int ldn, const void *conf, int ldc,
                                  float *out, int ldo, void *stream) {
  GSR_REQUIRE(P >= 0, "dn_normal_head: bad sizes");
  if (P == 0) return GSR_OK;
  GSR_REQUIRE(nrm && conf && out, "dn_normal_head: null pointer");
  hipLaunchKernelGGL(normal_head_kernel, dim3((unsigned)gsr::ceil_div64(P, 256)), dim3(256), 0,
                     (hipStream_t)stream, P, (const h16 *)nrm, ldn, (const h16 *)conf, ldc, out, ldo);
  GSR_CHECK_LAUNCH("dn_normal_head");
  return GSR_OK;
}

extern "C" int gsr_dn_convex_upsample(int H, int W, int F, const float *flow, const void *mask, int ldm,
                                      float min_val, float max_val, float regress_scale, float *depth,
                                      float *conf, float *normal, void *stream) {
  GSR_REQUIRE(H > 0 && W > 0 && F > 0 && ldm >= 9 * F * F, "dn_convex_upsample: bad sizes");
  GSR_REQUIRE(flow && mask && depth && conf && normal, "dn_convex_upsample: null pointer");
  const int64_t total = (int64_t)H * W * F * F;
  hipLaunchKernelGGL(convex_upsample_kernel, dim3((unsigned)gsr::ceil_div64(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, H, W, F, flow, (const h16 *)mask, ldm, min_val, max_val,
                     regress_scale, depth, conf, normal);
  GSR_CHECK_LAUNCH("dn_convex_upsample");
  return GSR_OK;
}

extern "C" int gsr_dn_cvt_f32_f16(int64_t P, int C, const float *in, int ldi, void *out, int ldo,
                                  void *stream) {
  GSR_REQUIRE(P >= 0 && C > 0, "dn_cvt: bad sizes");
  if (P == 0) return GSR_OK;
  GSR_REQUIRE(in && out, "dn_cvt: null pointer");
  hipLaunchKernelGGL(cvt_kernel, dim3(grid_for(P * C)), dim3(256), 0, (hipStream_t)stream, P, C, in, ldi,
                     (h16 *)out, ldo);
  GSR_CHECK_LAUNCH("dn_cvt");
  return GSR_OK;
}
