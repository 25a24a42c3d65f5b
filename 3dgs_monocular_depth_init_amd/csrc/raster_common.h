// raster_common.h -- pieces shared by the tile-list builder and the compositing kernels.
//
// Wave/tile shape (both compositing kernels): ONE wave64 per 16x16 tile. The tile is cut
// into four 8x8 quadrants; lane l owns pixel (l & 7, l >> 3) of EVERY quadrant
// (4 pixels per lane). Every (tile, Gaussian) pair of the tile lists carries a 4-bit
// quadrant mask from an exact ellipse-vs-rectangle test (min of sigma over the
// quadrant's pixel centres <= ln(255*opacity), i.e. at least one pixel could reach
// alpha >= 1/255). The mask is wave-uniform, so the per-quadrant body is skipped with
// SCALAR branches: with ~6 px radii a Gaussian touches 1.8 of the 4 quadrants, which
// halves the VALU work without changing any result (a skipped quadrant has
// alpha < 1/255 everywhere).
//
// Round 3: the mask is computed ONCE per pair, where the pair is created (the emit pass
// of isect_bucket.hip, or gsr_pair_masks for lists built elsewhere), and travels in the
// top four bits of the pair word the compositing kernels read:
//     pair_ids[i] = g | clamp << 27 | mask << 28     (g = camera * N + Gaussian < 2^27;
//                                                    clamp = opacity > 0.999: alpha may reach its cap)
// Before, both compositing kernels re-derived it while staging each batch (four
// min_sigma_rect evaluations per pair and kernel, ~14 live registers that pushed the
// forward into scratch and made it wait for its own prefetch: profiles/r03_fwd_timeline_before.json).
// With `tight` lists a pair whose mask is 0 is never emitted at all (17.5 % of the c4 pairs,
// profiles/r03_pair_stats.jsonl).
#pragma once
#include "common.h"
#include "gs_math.h"

namespace gsr {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float PIX_DONE = 1.0e18f;   // pixel x of a finished / outside pixel: sigma -> huge, alpha -> 0

constexpr int PAIR_MASK_SHIFT = 28;
constexpr uint32_t PAIR_CLAMP_BIT = 1u << 27;   // opacity > 0.999: alpha can reach the clamp
constexpr uint32_t PAIR_ID_MASK = 0x07ffffffu;

// sigma' = log2(e) * sigma = ha*dx^2 + bb*dx*dy + hc*dy^2, evaluated identically in
// forward and backward as dx*(ha*dx + B) + C with the row terms B = bb*dy, C = hc*dy*dy:
// the two quadrants of a row share dy, so B and C cost 3 VALU per (Gaussian, row) and
// sigma 2 per pixel.
__device__ __forceinline__ void sigma_row_terms(float bb, float hc, float dy, float &B, float &C) {
  B = bb * dy;
  C = (hc * dy) * dy;
}
__device__ __forceinline__ float sigma_l2(float ha, float dx, float B, float C) {
  return fmaf(dx, fmaf(ha, dx, B), C);
}
// ov with sigma's sign bit OR-ed in: (result >= threshold) <=> (sigma >= 0 and ov >= threshold)
__device__ __forceinline__ float with_sign_of(float ov, float sigma) {
  return __uint_as_float(__float_as_uint(ov) | (__float_as_uint(sigma) & 0x80000000u));
}

// alpha >= 1/255  <=>  sigma <= ln(255*op); the small margin keeps the test conservative
__device__ __forceinline__ float alpha_tau(float op) {
  const float tau = logf(op * 255.0f);
  return tau + 1e-4f * (1.0f + fabsf(tau));
}

// 4-bit quadrant mask of a Gaussian against the tile whose top-left pixel is (tx0, ty0): bit q set
// <=> some pixel centre of quadrant q (x half q & 1, y half q >> 1) can reach alpha >= 1/255, i.e.
// min over the quadrant's rectangle of pixel centres of sigma(d) = (a dx^2 + c dy^2)/2 + b dx dy is
// <= tau_m. sigma is convex with its minimum at the mean, so over a rectangle that does not contain
// the mean it is minimal on the edge(s) FACING the mean: one vertical edge when the mean lies beside
// the rectangle, one horizontal edge when above / below, both at a corner. Along a vertical edge at
// offset dx the minimum over dy is at dy* = -b dx / c clamped to the edge, with value
// kx dx^2 + c/2 (dy - dy*)^2, kx = (a - b^2/c)/2; likewise for horizontal edges. The per-Gaussian
// constants are set up once (PairConic), the x- and y-halves of the tile share their edge terms.
struct PairConic {
  float mx, my, ha, hc, sx, sy, kx, ky, tau_m;   // ha = a/2, hc = c/2, sx = -b/a, sy = -b/c
  bool clamp;                                    // opacity > 0.999 (gs::ALPHA_MAX)
};
__device__ __forceinline__ PairConic make_pair_conic(float mx, float my, float a, float b, float c,
                                                     float op) {
  PairConic p;
  const float ia = __builtin_amdgcn_rcpf(a), ic = __builtin_amdgcn_rcpf(c);
  p.mx = mx;
  p.my = my;
  p.ha = 0.5f * a;
  p.hc = 0.5f * c;
  p.sx = -b * ia;
  p.sy = -b * ic;
  // kx = (a - b^2/c)/2 = det/(2c), ky = det/(2a) with det = a c - b^2 as a compensated difference of
  // products (Kahan): for a needle-like rotated Gaussian (conic eigenvalues ~3 and ~1e-4) the plain
  // a - b*b*rcp(c) cancels to ~1e-3 relative error, which times de^2 near the alpha = 1/255 boundary
  // exceeded tau's safety margin -- a marginally visible pair or quadrant could be dropped.
  const float w = b * b;
  const float det = fmaf(a, c, -w) + fmaf(-b, b, w);
  p.kx = 0.5f * det * ic;
  p.ky = 0.5f * det * ia;
  p.tau_m = alpha_tau(op);
  p.clamp = op > gs::ALPHA_MAX;
  return p;
}
// One axis of one tile: per half h (0: pixels 0..7, 1: pixels 8..15) the offsets of the mean from the
// first and last pixel centre, whether the mean lies inside the half's span, and the terms of the
// facing edge (offset de): pe = s * de (the minimiser along the edge), ke = k * de^2.
// x axis: s = sy, k = kx (vertical edges); y axis: s = sx, k = ky (horizontal edges). Computed once
// per tile column / tile row and shared by the tiles of a rectangle (the emit pass walks rows).
struct AxisTerms {
  float lo[2], hi[2], pe[2], ke[2];
  bool in[2];
};
__device__ __forceinline__ AxisTerms axis_terms(float mean, float s, float k, float t0) {
  AxisTerms a;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    a.hi[h] = mean - (t0 + (8.f * (float)h + 0.5f));
    a.lo[h] = a.hi[h] - 7.f;
    a.in[h] = a.lo[h] * a.hi[h] <= 0.f;
    const float de = a.hi[h] < 0.f ? a.hi[h] : a.lo[h];
    a.pe[h] = s * de;
    a.ke[h] = k * de * de;
  }
  return a;
}
__device__ __forceinline__ int pair_quadrant_mask(const PairConic &p, const AxisTerms &X, const AxisTerms &Y) {
  int qmask = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int xh = q & 1, yh = q >> 1;
    const float tv = __builtin_amdgcn_fmed3f(X.pe[xh], Y.lo[yh], Y.hi[yh]) - X.pe[xh];
    const float sv = fmaf(p.hc * tv, tv, X.ke[xh]);
    const float th = __builtin_amdgcn_fmed3f(Y.pe[yh], X.lo[xh], X.hi[xh]) - Y.pe[yh];
    const float sh = fmaf(p.ha * th, th, Y.ke[yh]);
    // inside along x: only the horizontal edge faces the mean; inside along y: only the vertical one
    float m = fminf(X.in[xh] ? 3.0e38f : sv, Y.in[yh] ? 3.0e38f : sh);
    m = (X.in[xh] && Y.in[yh]) ? 0.f : m;
    qmask |= (m <= p.tau_m) ? (1 << q) : 0;
  }
  return qmask;
}
__device__ __forceinline__ int pair_quadrant_mask(const PairConic &p, float tx0, float ty0) {
  return pair_quadrant_mask(p, axis_terms(p.mx, p.sy, p.kx, tx0), axis_terms(p.my, p.sx, p.ky, ty0));
}

// Packed per-(camera,Gaussian) compositing record, one 64-byte row (= one cache line / one
// fabric request per gather), written by gsr_project_fwd or gsr_pack_records with the conic
// ALREADY in the units the compositing loops use:
//   [0] mx  my  ha  bb    [1] hc  opacity  col0  col1    [2] col2  col3  col4  -    [3] unused
//   ha = 0.5*a*log2(e), bb = b*log2(e), hc = 0.5*c*log2(e)
// The compositing kernels copy rows into LDS with LDS-DMA (no registers, no arithmetic).
constexpr int REC_FLOATS = 16;
__device__ __forceinline__ void write_record(float *__restrict__ records, int64_t g, float mx, float my,
                                             float a, float b, float c, float op, const float col[5]) {
  float4 *row = reinterpret_cast<float4 *>(records + g * REC_FLOATS);
  row[0] = make_float4(mx, my, 0.5f * a * LOG2E, b * LOG2E);
  row[1] = make_float4(0.5f * c * LOG2E, op, col[0], col[1]);
  row[2] = make_float4(col[2], col[3], col[4], 0.f);
}

// LDS image of a batch: RBATCH rows of 64 bytes, double buffered. One LDS-DMA instruction
// (global_load_lds_dwordx4: 64 lanes x 16 B, lane-linear in LDS) moves 16 whole rows:
// lanes 4r .. 4r+3 fetch the four 16-byte chunks of row r.
constexpr int RBATCH = 32;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// A value the compiler must treat as freshly produced here: keeps loop-invariant address
// arithmetic of rarely executed paths (and of the once-per-batch staging) from being hoisted
// into registers that stay live across the compositing loops.
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// The DMAs are asm statements on purpose: hipcc counts a __builtin_amdgcn_global_load_lds as an LDS
// store and, unable to prove that the buffer being filled is not the one being read, drains it
// (s_waitcnt vmcnt(0)) before the next LDS read -- i.e. right after issuing it. An asm statement is
// absent from its bookkeeping; the kernels wait for the data themselves (GSR_WAIT_VMEM at the top of
// the batch that reads it). M0 (the LDS destination base) is saved and restored inside the
// statement (cdna_hip_programming.md section 5.7). Only ACTIVE lanes transfer; lane l writes
// lds_dst + l * (16 or 4) bytes.
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_void_t *)p);
}
__device__ __forceinline__ void dma_16B(const void *src_lane, const void *lds_dst_uniform) {
  const uint32_t dst = lds_addr(lds_dst_uniform);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src_lane), "s"(dst)
      : "memory");
}
// the same with the non-temporal policy (once-read streams: the shN block of the projection forward)
__device__ __forceinline__ void dma_16B_nt(const void *src_lane, const void *lds_dst_uniform) {
  const uint32_t dst = lds_addr(lds_dst_uniform);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src_lane), "s"(dst)
      : "memory");
}
__device__ __forceinline__ void dma_4B(const void *src_lane, const void *lds_dst_uniform) {
  const uint32_t dst = lds_addr(lds_dst_uniform);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src_lane), "s"(dst)
      : "memory");
}
#define GSR_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// Diagnostic build (-DGSR_RASTER_TIMELINE=1, tools/raster_timeline.py): per tile (= wave) the compositing
// kernels add up shader-clock segments -- waiting for the staged batch, issuing the next DMAs (backward:
// + flushing the previous batch's rows), the compositing loop -- into 8 words per tile of a side buffer.
// The stamps serialise what the product kernel overlaps: read the SHARES, not the times.
#ifdef GSR_RASTER_TIMELINE
static __device__ unsigned long long *g_raster_timeline[2] = {nullptr, nullptr};   // [0] forward, [1] backward; one copy per translation unit
__device__ __forceinline__ unsigned long long tl_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define TL_DECL() unsigned long long tl_seg[4] = {0, 0, 0, 0}, tl_t = tl_now(); const unsigned long long tl_t0 = tl_t; int tl_batches = 0
#define TL_MARK(k) do { const unsigned long long n_ = tl_now(); tl_seg[k] += n_ - tl_t; tl_t = n_; } while (0)
#define TL_STORE(which, tile, pairs)                                                         \
  do {                                                                                       \
    unsigned long long *b_ = g_raster_timeline[which];                                       \
    if (b_ && threadIdx.x == 0) {                                                            \
      b_ += 8 * (int64_t)(tile);                                                             \
      b_[0] = tl_seg[0]; b_[1] = tl_seg[1]; b_[2] = tl_seg[2]; b_[3] = tl_seg[3];            \
      b_[4] = tl_now() - tl_t0; b_[5] = (unsigned long long)tl_batches;                      \
      b_[6] = (unsigned long long)(pairs); b_[7] = tl_t0;                                    \
    }                                                                                        \
  } while (0)
#else
#define TL_DECL()
#define TL_MARK(k)
#define TL_STORE(which, tile, pairs)
#endif

// Pair words travel through LDS too (a 3-slot ring of 32 words, filled two batches ahead): held in
// registers across the compositing loop they were what the register allocator spilled first -- and
// a spill of a just-loaded register is a wait for it. Slot (batch number % 3), word r = the batch's
// r-th pair: list position p0 + r*dir clamped to [lo, hi] (dir = +1 forward, -1 backward; clamped
// entries repeat a valid id and are never composited: the loops stop at the batch's own length).
constexpr int PW_SLOTS = 3;
template <int DIR>
__device__ __forceinline__ void dma_pair_words(const int32_t *__restrict__ pair_ids, int p0, int lo, int hi,
                                               int lane_, uint32_t *slot) {
  const int lane = opaque(lane_);
  if (lane < RBATCH) dma_4B(pair_ids + min(max(p0 + DIR * lane, lo), hi), slot);
}
// Stage the 32 records named by a slot of pair words into dst[32][4].
__device__ __forceinline__ void dma_stage_batch(const float *__restrict__ records, const uint32_t *slot,
                                                int lane_, float4 (*dst)[4]) {
  const int lane = opaque(lane_);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t id = slot[16 * h + (lane >> 2)] & PAIR_ID_MASK;
    dma_16B(records + (int64_t)id * REC_FLOATS + 4 * (lane & 3), &dst[16 * h][0]);
  }
}

// The same 32 records as three planes of 16-byte chunks, dst[c][r] = chunk c of record r (48 bytes
// per record instead of a whole 64-byte row: the backward needs the LDS for a seventh wave per
// SIMD). One DMA instruction per plane, lanes 0..31 active.
__device__ __forceinline__ void dma_stage_batch_planes(const float *__restrict__ records, const uint32_t *slot,
                                                       int lane_, float4 (*dst)[RBATCH]) {
  const int lane = opaque(lane_);
  if (lane < RBATCH) {
    const uint32_t id = slot[lane] & PAIR_ID_MASK;
    const float *src = records + (int64_t)id * REC_FLOATS;
#pragma unroll
    for (int c = 0; c < 3; ++c) dma_16B(src + 4 * c, &dst[c][0]);
  }
}

}  // namespace gsr
