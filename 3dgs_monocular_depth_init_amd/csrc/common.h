// common.h -- shared host/device helpers for libgsrast (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gsrast.h"

namespace gsr {

void set_error(const char *fmt, ...);

#define GSR_REQUIRE(cond, ...)         \
  do {                                 \
    if (!(cond)) {                     \
      gsr::set_error(__VA_ARGS__);     \
      return GSR_EINVAL;               \
    }                                  \
  } while (0)

#define GSR_CHECK_LAUNCH(name)                                                     \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      gsr::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));       \
      return GSR_EHIP;                                                             \
    }                                                                              \
  } while (0)

#define GSR_CHECK_HIP(expr)                                                        \
  do {                                                                             \
    hipError_t e__ = (expr);                                                       \
    if (e__ != hipSuccess) {                                                       \
      gsr::set_error("%s failed: %s", #expr, hipGetErrorString(e__));              \
      return GSR_EHIP;                                                             \
    }                                                                              \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Workgroups are dealt round-robin over the 8 XCDs (block b and b+8 share an
// XCD and its L2). Remap so that every XCD walks one contiguous range of
// work items: neighbouring tiles (which share Gaussians) then hit the same L2.
// Speed only; any placement is correct. grid must be xcd_grid(n).
static inline int xcd_grid(int n) { return ceil_div(n, 8) * 8; }
__device__ __forceinline__ int xcd_remap(int b, int n) {
  int per = (n + 7) >> 3;
  return (b & 7) * per + (b >> 3);
}

// Experiment knobs (A/B scaffolding: occupancy padding of the compositing kernels, GEMM core selection of the
// depth network) are environment variables ONLY in builds made with -DGSR_EXPERIMENT_KNOBS=1
// (tools/build_variants.sh knobs "-DGSR_EXPERIMENT_KNOBS=1"); the product library never calls getenv.
#ifdef GSR_EXPERIMENT_KNOBS
#include <stdlib.h>
static inline int gsr_knob_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v ? atoi(v) : dflt;
}
#else
static inline int gsr_knob_int(const char *, int dflt) { return dflt; }
#endif

#if defined(__HIPCC__)
// ---- wave64 reductions on DPP (no LDS traffic) ------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v) {
  int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, moved);
}
// Sum over the 64 lanes; the total is valid in lane 63 (and returned
// wave-uniformly through readlane).
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xb1>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4e>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x124>(v);   // row_ror:4
  v = dpp_add<0x128>(v);   // row_ror:8
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
  return __builtin_bit_cast(float,
                            __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// DPP steps the compiler does not fold into one instruction (it emits v_mov 0 +
// v_mov_dpp + v_add for them): written out. The leading s_nop covers the
// VALU-write -> DPP-read hazard, which the compiler cannot see inside the asm.
__device__ __forceinline__ float dpp_add_half_mirror(float v) {
  float r;
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf"
      : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ float dpp_add_bcast15(float v) {   // rows 1,3 += lane 15 of rows 0,2
  asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
  return v;
}
__device__ __forceinline__ float dpp_add_bcast31(float v) {   // rows 2,3 += lane 31
  asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
  return v;
}
// Same sum, total left in lane 63 only (no readlane / broadcast).
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v = dpp_add<0xb1>(v);
  v = dpp_add<0x4e>(v);
  v = dpp_add<0x124>(v);
  v = dpp_add<0x128>(v);
  v = dpp_add_bcast15(v);
  v = dpp_add_bcast31(v);
  return v;
}
// ---- 8 values x 64 lanes -> 8 totals in 18 VALU ops (instead of 8 x 6) ------
// Halving tree on the gfx950 lane-swap instructions: each level folds the wave
// in half AND packs two values into one register, so the work shrinks
// 8 -> 4 -> 2 -> 1 registers. On return, every lane of (row r = lane/16,
// half h = (lane/8)&1) holds the 64-lane total of v[TREE8_INDEX[h][r]].
#ifndef GSR_XLANE_LDS
#define GSR_XLANE_LDS 0
#endif
#if GSR_XLANE_LDS
// The two widest exchanges (across the 32-lane halves, across the 16-lane rows) through the LDS crossbar
// (ds_bpermute_b32 / ds_swizzle_b32: no LDS memory is touched) instead of v_permlane32_swap /
// v_permlane16_swap: a swap costs the VALU ~10 cycles (profiles/r02_valu_rate.jsonl), two v_cndmask + the
// add 6, and the LDS pipe of the compositing backward is otherwise idle (profiles/r04a_pmc.json: 14 M LDS
// against 237 M VALU instructions per launch). Same result layout as the swap versions.
__device__ __forceinline__ float swap32_add(float a, float b, int lane) {
  const bool hi = (lane & 32) != 0;
  const float u = hi ? a : b;            // what the OTHER half adds up
  const float t = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, u)));
  return (hi ? b : a) + t;
}
__device__ __forceinline__ float swap16_add(float p, float q, int lane) {
  const bool odd = (lane & 16) != 0;
  const float u = odd ? p : q;
  // bitmask mode: and 0x1f, or 0, xor 0x10 -> lane ^ 16 inside each 32-lane half
  const float t = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, u), 0x401f));
  return (odd ? q : p) + t;
}
// all lanes end with the 64-lane total
__device__ __forceinline__ float wave_sum_xlane(float v, int lane) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, v)));
  v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401f));
  v = dpp_add<0xb1>(v);
  v = dpp_add<0x4e>(v);
  v = dpp_add<0x124>(v);
  v = dpp_add<0x128>(v);
  return v;
}
#else
__device__ __forceinline__ float swap32_add(float a, float b, int) {
  // lanes 0-31: a folded over the two 32-lane halves; lanes 32-63: b folded
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  // NB: copy the elements out first -- `__builtin_bit_cast(float, r[1])` applied to
  // the vector element directly is miscompiled by ROCm 7.2 clang (reads r[0] twice).
  const unsigned r0 = r[0], r1 = r[1];
  return __uint_as_float(r0) + __uint_as_float(r1);
}
__device__ __forceinline__ float swap16_add(float p, float q, int) {
  // rows: [p0+p1, q0+q1, p2+p3, q2+q3]
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(q), false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return __uint_as_float(r0) + __uint_as_float(r1);
}
__device__ __forceinline__ float wave_sum_xlane(float v, int) { return wave_sum_lane63(v); }   // (lane 63 only)
#endif
__device__ __forceinline__ float tree_reduce8(const float v[8], int lane) {
  const float s01 = swap32_add(v[0], v[1], lane), s23 = swap32_add(v[2], v[3], lane);
  const float s45 = swap32_add(v[4], v[5], lane), s67 = swap32_add(v[6], v[7], lane);
  const float r1 = swap16_add(s01, s23, lane);     // rows: v0, v2, v1, v3 (16 partial lanes each)
  const float r2 = swap16_add(s45, s67, lane);     // rows: v4, v6, v5, v7
  const float t1 = dpp_add<0x128>(r1);       // row_ror:8 -> fold the two 8-lane halves
  const float t2 = dpp_add<0x128>(r2);
  float u = (lane & 8) ? t2 : t1;
  u = dpp_add<0xb1>(u);                      // quad_perm [1,0,3,2]
  u = dpp_add<0x4e>(u);                      // quad_perm [2,3,0,1]
  u = dpp_add_half_mirror(u);                // the other quad of the 8-lane half
  return u;
}
// value index held by (half h, row r) after tree_reduce8
__device__ __forceinline__ int tree8_index(int lane) {
  const int r = lane >> 4, h = (lane >> 3) & 1;
  const int base = (r == 0) ? 0 : (r == 1) ? 2 : (r == 2) ? 1 : 3;
  return base + 4 * h;
}

// Tile rectangle [x0,x1) x [y0,y1) a projected Gaussian touches (A.3): mean +- radius in
// tile units, clamped to the grid. false = culled / touches nothing.
__device__ __forceinline__ bool tile_rect_v(float mean_x, float mean_y, int rx, int ry, int tile_w,
                                            int tile_h, int &x0, int &x1, int &y0, int &y1) {
  if (rx <= 0 || ry <= 0) return false;
  const float inv = 1.0f / (float)GSR_TILE;
  const float mx = mean_x * inv, my = mean_y * inv;
  const float trx = (float)rx * inv, try_ = (float)ry * inv;
  x0 = min(max(0, (int)floorf(mx - trx)), tile_w);
  x1 = min(max(0, (int)ceilf(mx + trx)), tile_w);
  y0 = min(max(0, (int)floorf(my - try_)), tile_h);
  y1 = min(max(0, (int)ceilf(my + try_)), tile_h);
  return (x1 > x0) && (y1 > y0);
}
__device__ __forceinline__ bool tile_rect(const float *__restrict__ means2d,
                                          const int32_t *__restrict__ radii, int64_t g,
                                          int tile_w, int tile_h, int &x0, int &x1, int &y0,
                                          int &y1) {
  return tile_rect_v(means2d[g * 2], means2d[g * 2 + 1], radii[g * 2], radii[g * 2 + 1], tile_w,
                     tile_h, x0, x1, y0, y1);
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    int o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// All-lanes reductions of a 32-bit value without lane-address registers: four DPP steps inside the rows, then the
// two lane-swap instructions (a = b = v: every lane ends with the folded value). The __shfl_xor versions above need
// six address VGPRs (ds_bpermute), which the compiler computes once and keeps live across a whole kernel.
template <typename Op>
__device__ __forceinline__ int wave_allreduce_i32(int v, Op op) {
  v = op(v, __builtin_amdgcn_update_dpp(v, v, 0xb1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x4e, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));   // row_half_mirror
  v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));   // row_mirror
  {
    auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    v = op((int)r0, (int)r1);
  }
  {
    auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    v = op((int)r0, (int)r1);
  }
  return v;
}
__device__ __forceinline__ int wave_sum_i32_dpp(int v) { return wave_allreduce_i32(v, [](int a, int b) { return a + b; }); }
__device__ __forceinline__ int wave_max_i32_dpp(int v) { return wave_allreduce_i32(v, [](int a, int b) { return a > b ? a : b; }); }
__device__ __forceinline__ uint32_t wave_min_u32_dpp(uint32_t v) {
  return (uint32_t)wave_allreduce_i32((int)v, [](int a, int b) { return (uint32_t)a < (uint32_t)b ? a : b; });
}
__device__ __forceinline__ uint32_t wave_max_u32_dpp(uint32_t v) {
  return (uint32_t)wave_allreduce_i32((int)v, [](int a, int b) { return (uint32_t)a > (uint32_t)b ? a : b; });
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
#endif

}  // namespace gsr
