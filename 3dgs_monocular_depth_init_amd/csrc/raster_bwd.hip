// raster_bwd.hip -- A7: per-Gaussian gradients of the compositing.
//
// Replaces gsplat's rasterize_to_pixels backward that loss.backward()
// (gs_init_compare/runner.py:547) triggers. Same shape as the forward: one
// wave64 per 16x16 tile, 2x2 pixels per lane, tile list replayed back to front
// in LDS batches of 32 (records and pair words by LDS-DMA, one batch ahead; the
// quadrant masks arrive with the pair words). Per Gaussian the lane sums its 4
// pixels in registers, the wave reduces on DPP / lane swaps (no LDS), and the
// batch's totals are parked in LDS as [32][16] rows; each row is then flushed with
// ONE 64-byte-aligned group of float atomics into grad_rows[g][16] (16 adjacent
// lanes = one memory-side atomic request), instead of 9-11 scattered dword
// atomics per Gaussian. The flush of batch k runs at the top of batch k+1, BEFORE
// the DMA of batch k+2 is issued: the batch-top wait (vmcnt(0), the DMA has no other
// completion signal) then only ever waits for atomics that are a whole batch old.

#include <type_traits>

#include "raster_common.h"

#ifndef GSR_BWD_MASK_SWITCH
#define GSR_BWD_MASK_SWITCH 0   // 1: fifteen straight-line bodies (one per quadrant mask) in the FAST batches
#endif
#ifndef GSR_BWD_WAVES
#define GSR_BWD_WAVES 6   // waves per SIMD the CH <= 3 instantiation is register-allocated for
#endif

namespace gsr {

#ifdef GSR_BWD_COUNT_EMPTY
// Diagnostic build (tools/build_variants.sh count "-DGSR_BWD_COUNT_EMPTY=1"): [0] pairs composited, [1] pairs in
// which NO pixel of the tile passed the alpha >= 1/255 test (their reduction and LDS row are all zeros),
// [2] pairs composited in FAST batches. Read back with gsr_debug_bwd_counts.
static __device__ unsigned long long g_bwd_counts[4] = {0, 0, 0, 0};
#endif

// Sum over the 16 lanes of each DPP row; every lane of a row ends with the row sum.
__device__ __forceinline__ float row_sum16(float v) {
  v = dpp_add<0xb1>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4e>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x124>(v);   // row_ror:4
  v = dpp_add<0x128>(v);   // row_ror:8
  return v;
}

template <int CH, bool ABSGRAD>
__global__ void __launch_bounds__(64, (CH <= 3 && !ABSGRAD) ? GSR_BWD_WAVES : 4)
raster_bwd_kernel(int n_tiles, const float *__restrict__ records,
                  const float *__restrict__ backgrounds, int width, int height, int tile_w,
                  int tile_h, const int32_t *__restrict__ tile_offsets,
                  const int32_t *__restrict__ tile_order,
                  const int32_t *__restrict__ pair_ids,
                  const float *__restrict__ render_alphas, const int32_t *__restrict__ last_ids,
                  const float *__restrict__ v_render_colors,
                  const float *__restrict__ v_render_alphas, float *__restrict__ grad_rows, int planar) {
  // staged batches: chunk 0 = {mx, my, ha, bb}, 1 = {hc, opacity, col0, col1}, 2 = {col2, col3, col4, -}
  // of the batch's r-th Gaussian (r = 0 is the LAST list position of the batch)
  __shared__ __attribute__((aligned(16))) float4 sRec[2][3][RBATCH];   // [buffer][16-byte chunk][Gaussian]
  __shared__ uint32_t sPw[PW_SLOTS][RBATCH];
  __shared__ __attribute__((aligned(16))) float sG[RBATCH][GSR_GRAD_ROW];  // batch gradient rows

  if ((int)blockIdx.x >= n_tiles) return;
  const int tile = tile_order ? tile_order[blockIdx.x] : (int)blockIdx.x;
  const int tiles_per_cam = tile_w * tile_h;
  const int cam = tile / tiles_per_cam;
  const int tin = tile - cam * tiles_per_cam;
  const int ty = tin / tile_w, tx = tin - ty * tile_w;
  const int lane = threadIdx.x;
  const int lx = lane & 7, ly = lane >> 3;
  const int tx0 = tx * GSR_TILE, ty0 = ty * GSR_TILE;

  // Which field of a gradient row this lane stores after the reductions (-1: none):
  // lanes with (lane & 7) == 0 hold one tree_reduce8 total each -- tree values 0,1 = first
  // moments (slots MEAN2D), 2..4 = conic, 5..7 = colours 0..2; lane 63 holds the wave sum
  // of v_sigma (slot OPAC, scaled by -1/opacity at the flush); lane 1 raises the row's
  // "touched" flag (slot 15). One predicated ds_write per Gaussian stores all of them.
  int wfield = -1;
  if ((lane & 7) == 0) {
    const int ti = tree8_index(lane);
    wfield = (ti < 2) ? GSR_GR_MEAN2D + ti
             : (ti < 5) ? GSR_GR_CONIC + (ti - 2)
             : ((ti - 5) < CH ? GSR_GR_COLOR + (ti - 5) : -1);
  }
  if (lane == 63) wfield = GSR_GR_OPAC;
  if (lane == 1) wfield = 15;

  // R = T_final*(v_alpha_out - <background, v_out>) - <buf, v_out>, where buf is the colour
  // accumulated behind the current Gaussian: only its dot product with v_out is ever
  // needed, so one scalar per pixel replaces the CH-vector
  // pixel q of this lane: x = px[q & 1], y = py[q >> 1] (the lane's column / row in the left / right and
  // upper / lower quadrants). No per-pixel copy of x is needed: a pixel that takes no part (outside
  // the image, or nothing blended into it) is parked by T = R = v_out = 0, with which every
  // contribution below is an exact zero (T stays 0 under T *= 1/(1-a), a <= 0.999).
  float px[2], py[2], T[4], R[4], vout[4][CH];
  int last[4], qmax[4];
  px[0] = (float)(tx0 + lx) + 0.5f;
  px[1] = px[0] + 8.0f;
  py[0] = (float)(ty0 + ly) + 0.5f;
  py[1] = py[0] + 8.0f;
  const int s = tile_offsets[tile], e = tile_offsets[tile + 1];
  if (e <= s) return;
  int max_last = -1, my_min_last = 0x7fffffff;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int x = tx0 + 8 * (q & 1) + lx, y = ty0 + 8 * (q >> 1) + ly;
    last[q] = -1;
    T[q] = 0.f;
    R[q] = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) vout[q][k] = 0.f;
    if (x < width && y < height) {
      const int64_t pix = ((int64_t)cam * height + y) * width + x;
      last[q] = last_ids[pix];
      T[q] = 1.0f - render_alphas[pix];
      float kk = v_render_alphas ? v_render_alphas[pix] : 0.f;
#pragma unroll
      for (int k = 0; k < CH; ++k)      // [C,H,W,CH], or planes [C,CH,H,W] (gsr_rasterize_bwd_planar)
        vout[q][k] = planar ? v_render_colors[pix + ((int64_t)cam * (CH - 1) + k) * height * width] : v_render_colors[pix * CH + k];
      if (backgrounds) {
#pragma unroll
        for (int k = 0; k < CH; ++k) kk -= backgrounds[cam * CH + k] * vout[q][k];
      }
      R[q] = T[q] * kk;
    }
    qmax[q] = __builtin_amdgcn_readfirstlane(wave_max_i32(last[q]));
    max_last = max(max_last, qmax[q]);
    // a pixel outside the image or one nothing was blended into takes no part: park it (see
    // above) so that it never holds back the fast path below
    if (last[q] < s) {
      T[q] = 0.f;
      R[q] = 0.f;
#pragma unroll
      for (int k = 0; k < CH; ++k) vout[q][k] = 0.f;
      last[q] = 0x7fffffff;
    }
    my_min_last = min(my_min_last, last[q]);
  }
  if (max_last < s) return;  // nothing was blended into this tile (wave-uniform)
  const int start = min(max_last, e - 1);
  // every pixel of the tile is active for list positions <= min_last: batches entirely
  // below it need neither the per-pixel "idx <= last" test nor the per-quadrant one
  const int min_last = -__builtin_amdgcn_readfirstlane(wave_max_i32(-my_min_last));

  // Flush the parked rows of a finished batch: 4 Gaussians per wave instruction, 16 lanes = one
  // 64-byte row. Reads the batch's records (conic for v_xy, opacity) and pair words (the ids).
  auto flush = [&](int n, const float4(*rec)[RBATCH], const uint32_t *pw) {
    const int f = lane & 15;
    const bool field_used =
        (f < GSR_GR_COLOR + CH) || (ABSGRAD && (f == GSR_GR_ABS || f == GSR_GR_ABS + 1));
    for (int j0 = 0; j0 < n; j0 += 4) {
      const int j = j0 + (lane >> 4);
      if (j < n && field_used && sG[j][15] != 0.f) {
        const uint32_t g = pw[j] & PAIR_ID_MASK;
        float val = sG[j][f];
        if (f < 2) {   // v_xy = conic * (first moments): row holds (m_x, m_y)
          const float4 Aj = rec[0][j];
          const float4 Bj = rec[1][j];
          const float ca = Aj.z * (2.0f / LOG2E), cb = Aj.w * (1.0f / LOG2E),
                      cc = Bj.x * (2.0f / LOG2E);
          const float mx_ = sG[j][0], my_ = sG[j][1];
          val = (f == 0) ? fmaf(ca, mx_, cb * my_) : fmaf(cb, mx_, cc * my_);
        }
        if (f == GSR_GR_OPAC) val = -val / rec[1][j].y;   // row holds sum v_sigma
        if (f == GSR_GR_CONIC || f == GSR_GR_CONIC + 2) val *= 0.5f;
#if defined(GSR_BWD_KO) && GSR_BWD_KO == 3   // knock-out (timing only): no atomics
        if (val == 1.2345e-30f) grad_rows[(int64_t)g * GSR_GRAD_ROW + f] = val;
#else
        atomicAdd(grad_rows + (int64_t)g * GSR_GRAD_ROW + f, val);
#endif
      }
    }
  };

  // batch k covers the list positions start - 32k - 31 .. start - 32k; its r-th Gaussian is
  // position start - 32k - r. Pipeline: pair words two batches ahead, records one batch ahead.
  dma_pair_words<-1>(pair_ids, start, s, start, lane, sPw[0]);
  dma_pair_words<-1>(pair_ids, start - RBATCH, s, start, lane, sPw[1]);
  GSR_WAIT_VMEM();
  dma_stage_batch_planes(records, sPw[0], lane, sRec[0]);
  int buf = 0, slot = 0, n_prev = 0;
  TL_DECL();
  for (int batch_end = start; batch_end >= s; batch_end -= RBATCH, buf ^= 1, slot = (slot == 2) ? 0 : slot + 1) {
    const int n = min(RBATCH, batch_end - s + 1);
    TL_MARK(3);        // (3) = everything outside the three segments below
    GSR_WAIT_VMEM();   // this batch's records and the next batch's words are in LDS
    TL_MARK(0);        // (0) waiting for the staged batch
    const int s1 = (slot == 2) ? 0 : slot + 1, s2 = (s1 == 2) ? 0 : s1 + 1;   // s2 = previous batch's slot
    if (n_prev > 0) flush(n_prev, sRec[buf ^ 1], sPw[s2]);
    if (batch_end - RBATCH >= s) {
      dma_stage_batch_planes(records, sPw[s1], lane, sRec[buf ^ 1]);
      dma_pair_words<-1>(pair_ids, batch_end - 2 * RBATCH, s, start, lane, sPw[s2]);
    }
    TL_MARK(1);        // (1) flushing the previous batch's rows (atomics) + issuing the next DMAs
#ifdef GSR_RASTER_TIMELINE
    ++tl_batches;
#endif
    const float4(*rec)[RBATCH] = sRec[buf];
    const uint32_t pw = sPw[slot][lane & 31];   // lane j < n: pair j's word (mask in the top bits)
    // opacity > 0.999 somewhere in the batch (the pair words' clamp flags): alpha may hit the clamp
    // (no gradient there)
    const bool can_clamp = (lane < n) && (pw & PAIR_CLAMP_BIT);
    if (lane < RBATCH) sG[lane][15] = 0.f;   // "row touched" flag of Gaussian `lane` of this batch

    // One Gaussian against the tile. FAST (wave-uniform, decided per batch): every pixel
    // is active and no opacity of the batch exceeds the alpha clamp, so the tests against
    // last[] / qmax[] and the clamp handling are compiled out.
#ifdef GSR_BWD_NO_FAST
    const bool fast_batch = false;   // (experiment: one copy of the quadrant bodies instead of two)
#else
    const bool fast_batch = (batch_end <= min_last) && !__any(can_clamp);
#endif
    auto composite = [&](auto fast_tag, auto mask_tag, int j, unsigned qm_dyn) {
      constexpr bool FAST = decltype(fast_tag)::value;
      constexpr unsigned QMC = decltype(mask_tag)::value;   // 1..15: the pair's quadrant mask as a compile-time constant
      unsigned qm = QMC ? QMC : qm_dyn;
      const int idx = batch_end - j;
      if (!FAST) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (idx > qmax[q]) qm &= ~(1u << q);   // scalar: no pixel of q blended this far down
      }
      if (qm == 0) return;
      const float4 Ac = rec[0][j], Bc = rec[1][j];
      float col[CH];
      col[0] = Bc.z;
      if (CH > 1) col[1] = Bc.w;
      if (CH > 2) col[2] = rec[2][j].x;
      if (CH > 3) col[3] = rec[2][j].y;
      if (CH > 4) col[4] = rec[2][j].z;
      const float opac = Bc.y;
      // conic in natural units for the gradient formulas: a = 2*ha/log2e etc.
      const float ca = Ac.z * (2.0f / LOG2E), cb = Ac.w * (1.0f / LOG2E), cc = Bc.x * (2.0f / LOG2E);

#ifdef GSR_BWD_COUNT_EMPTY
      bool any_valid = false;
#endif
      float g_xy[2] = {0.f, 0.f}, g_con[3] = {0.f, 0.f, 0.f}, g_vs = 0.f, g_col[CH];
      float g_abs[2] = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < CH; ++k) g_col[k] = 0.f;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
       if (qm & (3u << (2 * r))) {   // scalar: this row of quadrants is touched
        const float dy = Ac.y - py[r];
        float Br, Cr;
        sigma_row_terms(Ac.w, Bc.x, dy, Br, Cr);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
         const int q = 2 * r + h;
         if (qm & (1u << q)) {   // scalar branch
          const float dx = Ac.x - px[h];
          const float sg = sigma_l2(Ac.z, dx, Br, Cr);
#if defined(GSR_BWD_KO) && GSR_BWD_KO == 2   // knock-out (timing only): no transcendental
          const float vis = 1.0f - 0.001f * sg;
#else
          const float vis = __builtin_amdgcn_exp2f(-sg);
#endif
          const float ov = opac * vis;
          // valid <=> sigma >= 0 and alpha = min(0.999, ov) >= 1/255 (<=> ov >= 1/255):
          // sigma's sign bit is OR-ed into ov, so one compare covers both
          bool valid = with_sign_of(ov, sg) >= gs::ALPHA_THRESHOLD;
          if (!FAST) valid = valid && (idx <= last[q]);
#ifdef GSR_BWD_COUNT_EMPTY
          any_valid |= valid;
#endif
          const float ovv = valid ? ov : 0.f;
          // FAST: opacity <= 0.999 and exp2(-sigma) <= 1, so the clamp cannot bind
          const float a = FAST ? ovv : fminf(gs::ALPHA_MAX, ovv);
#if defined(GSR_BWD_KO) && GSR_BWD_KO == 2
          const float ra = 1.0f + a;
#else
          const float ra = __builtin_amdgcn_rcpf(1.0f - a);
#endif
          T[q] *= ra;
          const float fac = a * T[q];
          float D = col[0] * vout[q][0];      // <colour, v_out>
#pragma unroll
          for (int k = 1; k < CH; ++k) D = fmaf(col[k], vout[q][k], D);
#pragma unroll
          for (int k = 0; k < CH; ++k) g_col[k] = fmaf(fac, vout[q][k], g_col[k]);
          // v_alpha = sum_k (col_k*T - buf_k/(1-a)) * v_out_k + T_final/(1-a) * (...)
          const float v_alpha = fmaf(T[q], D, R[q] * ra);
          R[q] = fmaf(-fac, D, R[q]);         // buf += col*fac  =>  <buf, v_out> += fac*D
          // no gradient through a clamped alpha; ovv = 0 already silences an invalid pair
          const float va = (FAST || ovv <= gs::ALPHA_MAX) ? v_alpha : 0.f;
          const float v_sigma = -ovv * va;
          const float tdx = v_sigma * dx, tdy = v_sigma * dy;
          g_con[0] = fmaf(tdx, dx, g_con[0]);   // second moments; the 1/2 of v_conic.a / .c
          g_con[1] = fmaf(tdx, dy, g_con[1]);   // is applied at the flush
          g_con[2] = fmaf(tdy, dy, g_con[2]);
          // v_xy = conic * (sum v_sigma*dx, sum v_sigma*dy): only the two first
          // moments are summed per pixel, the 2x2 product is applied once after
          // the reduction (absgrad needs the per-pixel value).
          g_xy[0] += tdx;
          g_xy[1] += tdy;
          if (ABSGRAD) {
            g_abs[0] += fabsf(fmaf(ca, tdx, cb * tdy));
            g_abs[1] += fabsf(fmaf(cb, tdx, cc * tdy));
          }
          // v_opacity = sum vis*va = -(sum v_sigma) / opacity: the zeroth moment is
          // summed, the division happens once at the flush
          g_vs += v_sigma;
         }
        }
       }
      }
#ifdef GSR_BWD_COUNT_EMPTY
      {
        const bool none = !__any(any_valid);
        if (lane == 0) {
          atomicAdd(&g_bwd_counts[0], 1ull);
          if (none) atomicAdd(&g_bwd_counts[1], 1ull);
          if (FAST) atomicAdd(&g_bwd_counts[2], 1ull);
        }
      }
#endif
      // 8 of the sums go through the lane-swap halving tree (18 VALU for all 8), the
      // zeroth moment through a plain wave sum that ends in lane 63; the writer lanes
      // (wfield) store everything with one predicated ds_write into the 64-byte LDS row.
      {
        const float tv[8] = {g_xy[0], g_xy[1], g_con[0], g_con[1], g_con[2], g_col[0],
                             (CH > 1) ? g_col[1] : 0.f, (CH > 2) ? g_col[2] : 0.f};
#if defined(GSR_BWD_KO) && GSR_BWD_KO == 1   // knock-out (timing only): no cross-lane reduction
        float u = tv[0] + tv[1] + tv[2] + tv[3] + tv[4] + tv[5] + tv[6] + tv[7];
        const float r_s = g_vs;
#else
        float u = tree_reduce8(tv, lane);
        const float r_s = wave_sum_xlane(g_vs, lane);
#endif
        u = (lane == 63) ? r_s : u;
        u = (lane == 1) ? 1.0f : u;
        float *row = &sG[j][0];
        if (wfield >= 0) row[wfield] = u;
        if (CH > 3) {
          const float r3 = wave_sum(g_col[3]);
          if (lane == 0) row[GSR_GR_COLOR + 3] = r3;
        }
        if (CH > 4) {
          const float r4 = wave_sum(g_col[4]);
          if (lane == 0) row[GSR_GR_COLOR + 4] = r4;
        }
        if (ABSGRAD) {
          const float r_ax = wave_sum(g_abs[0]), r_ay = wave_sum(g_abs[1]);
          if (lane == 0) {
            row[GSR_GR_ABS] = r_ax;
            row[GSR_GR_ABS + 1] = r_ay;
          }
        }
      }
    };
    if (fast_batch) {
#if GSR_BWD_MASK_SWITCH
      // one straight-line body per quadrant mask (FAST batches: the mask is exactly the pair word's): the
      // per-quadrant scalar tests disappear and the bodies of a mask's quadrants can be scheduled together
      for (int j = 0; j < n; ++j) {
        const unsigned qm = (unsigned)__builtin_amdgcn_readlane((int)pw, j) >> PAIR_MASK_SHIFT;
        switch (qm) {
#define GSR_CASE(M) case M: composite(std::true_type{}, std::integral_constant<unsigned, M>{}, j, qm); break;
          GSR_CASE(1) GSR_CASE(2) GSR_CASE(3) GSR_CASE(4) GSR_CASE(5) GSR_CASE(6) GSR_CASE(7) GSR_CASE(8)
          GSR_CASE(9) GSR_CASE(10) GSR_CASE(11) GSR_CASE(12) GSR_CASE(13) GSR_CASE(14) GSR_CASE(15)
#undef GSR_CASE
          default: break;
        }
      }
#else
      for (int j = 0; j < n; ++j)
        composite(std::true_type{}, std::integral_constant<unsigned, 0>{}, j,
                  (unsigned)__builtin_amdgcn_readlane((int)pw, j) >> PAIR_MASK_SHIFT);
#endif
    } else {
      for (int j = 0; j < n; ++j)
        composite(std::false_type{}, std::integral_constant<unsigned, 0>{}, j,
                  (unsigned)__builtin_amdgcn_readlane((int)pw, j) >> PAIR_MASK_SHIFT);
    }
    TL_MARK(2);        // (2) the compositing loop
    n_prev = n;
  }
  // the last batch's rows: its records are in sRec[buf ^ 1] (buf was flipped on the way out),
  // its words in the slot before `slot`
  GSR_WAIT_VMEM();
  flush(n_prev, sRec[buf ^ 1], sPw[(slot == 0) ? 2 : slot - 1]);
  TL_MARK(1);
  TL_STORE(1, tile, start - s + 1);
}

// Test hook for the lane-swap tree (the semantics of v_permlane{16,32}_swap are
// checked on the GPU by tests/test_gpu_rasterization.py::test_tree_reduce8).
__global__ void debug_tree_reduce8_kernel(const float *__restrict__ in, float *__restrict__ out,
                                          int *__restrict__ idx_out) {
  const int lane = threadIdx.x;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = in[k * 64 + lane];
  out[lane] = tree_reduce8(v, lane);
  idx_out[lane] = tree8_index(lane);
  out[64 + lane] = wave_sum(v[0]);
  out[128 + lane] = wave_sum_xlane(v[1], lane);   // only lane 63 is meaningful
}

template <int CH>
static int launch_bwd(int n_tiles, const float *records, const float *backgrounds, int width,
                      int height, int tile_w, int tile_h, const int32_t *tile_offsets,
                      const int32_t *tile_order, const int32_t *pair_ids,
                      const float *render_alphas, const int32_t *last_ids,
                      const float *v_render_colors, const float *v_render_alphas, int absgrad,
                      float *grad_rows, hipStream_t stream, int planar) {
  const unsigned pad = (unsigned)gsr_knob_int("GSR_BWD_LDS_PAD", 0);   // experiment knob: see raster_fwd.hip (0 in the product build)
  // experiment knob (timing only, wrong gradients): launch just the GSR_BWD_TILE_LIMIT longest tiles, or (negative)
  // skip the -limit longest -- how much of the launch is the under-occupied tail of its 8160 / 6144-slot schedule?
  {
    const int lim = gsr_knob_int("GSR_BWD_TILE_LIMIT", 0);
    if (lim > 0 && lim < n_tiles) n_tiles = lim;
    if (lim < 0 && -lim < n_tiles && tile_order) {
      tile_order += -lim;
      n_tiles -= -lim;
    }
  }
  if (absgrad)
    hipLaunchKernelGGL((raster_bwd_kernel<CH, true>), dim3(n_tiles), dim3(64), pad, stream, n_tiles,
                       records, backgrounds, width, height, tile_w, tile_h, tile_offsets,
                       tile_order, pair_ids, render_alphas, last_ids, v_render_colors,
                       v_render_alphas, grad_rows, planar);
  else
    hipLaunchKernelGGL((raster_bwd_kernel<CH, false>), dim3(n_tiles), dim3(64), pad, stream, n_tiles,
                       records, backgrounds, width, height, tile_w, tile_h, tile_offsets,
                       tile_order, pair_ids, render_alphas, last_ids, v_render_colors,
                       v_render_alphas, grad_rows, planar);
  GSR_CHECK_LAUNCH("rasterize_bwd");
  return GSR_OK;
}

}  // namespace gsr

extern "C" int gsr_debug_tree_reduce8(const float *in, float *out, int32_t *idx_out,
                                      void *stream) {
  GSR_REQUIRE(in && out && idx_out, "debug_tree_reduce8: null pointer");
  hipLaunchKernelGGL(gsr::debug_tree_reduce8_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     in, out, idx_out);
  GSR_CHECK_LAUNCH("debug_tree_reduce8");
  return GSR_OK;
}

static int rasterize_bwd_impl(int C, int CH, const float *records, const float *backgrounds,
                              int width, int height, int tile_w, int tile_h,
                              const int32_t *tile_offsets, const int32_t *tile_order,
                              const int32_t *pair_ids, const float *render_alphas,
                              const int32_t *last_ids, const float *v_render_colors,
                              const float *v_render_alphas, int absgrad, float *grad_rows,
                              int planar, void *stream) {
  GSR_REQUIRE(C >= 0 && width > 0 && height > 0, "rasterize_bwd: bad sizes");
  GSR_REQUIRE(tile_w == gsr::ceil_div(width, GSR_TILE) && tile_h == gsr::ceil_div(height, GSR_TILE),
              "rasterize_bwd: tile grid does not match image");
  GSR_REQUIRE(CH >= 1 && CH <= 5, "rasterize_bwd: CH=%d", CH);
  if (C == 0) return GSR_OK;
  GSR_REQUIRE(tile_offsets && render_alphas && last_ids && v_render_colors && grad_rows,
              "rasterize_bwd: null pointer");
  int n_tiles = C * tile_w * tile_h;
  hipStream_t st = (hipStream_t)stream;
#define GSR_BWD_CASE(K)                                                                         \
  case K:                                                                                       \
    return gsr::launch_bwd<K>(n_tiles, records, backgrounds, width, height, tile_w, tile_h,     \
                              tile_offsets, tile_order, pair_ids, render_alphas, last_ids,      \
                              v_render_colors, v_render_alphas, absgrad, grad_rows, st, planar);
  switch (CH) {
    GSR_BWD_CASE(1)
    GSR_BWD_CASE(2)
    GSR_BWD_CASE(3)
    GSR_BWD_CASE(4)
    GSR_BWD_CASE(5)
  }
#undef GSR_BWD_CASE
  return GSR_EINVAL;
}

extern "C" int gsr_rasterize_bwd(int C, int CH, const float *records, const float *backgrounds,
                                 int width, int height, int tile_w, int tile_h,
                                 const int32_t *tile_offsets, const int32_t *tile_order,
                                 const int32_t *pair_ids, const float *render_alphas,
                                 const int32_t *last_ids, const float *v_render_colors,
                                 const float *v_render_alphas, int absgrad, float *grad_rows,
                                 void *stream) {
  return rasterize_bwd_impl(C, CH, records, backgrounds, width, height, tile_w, tile_h, tile_offsets, tile_order,
                            pair_ids, render_alphas, last_ids, v_render_colors, v_render_alphas, absgrad, grad_rows, 0,
                            stream);
}
// The same with v_render_colors laid out [C,CH,H,W] (the gradient of a render produced by gsr_rasterize_fwd_planar).
extern "C" int gsr_rasterize_bwd_planar(int C, int CH, const float *records, const float *backgrounds,
                                        int width, int height, int tile_w, int tile_h,
                                        const int32_t *tile_offsets, const int32_t *tile_order,
                                        const int32_t *pair_ids, const float *render_alphas,
                                        const int32_t *last_ids, const float *v_render_colors,
                                        const float *v_render_alphas, int absgrad, float *grad_rows,
                                        void *stream) {
  return rasterize_bwd_impl(C, CH, records, backgrounds, width, height, tile_w, tile_h, tile_offsets, tile_order,
                            pair_ids, render_alphas, last_ids, v_render_colors, v_render_alphas, absgrad, grad_rows, 1,
                            stream);
}

#ifdef GSR_BWD_COUNT_EMPTY
extern "C" int gsr_debug_bwd_counts(unsigned long long *out4, int reset) {
  GSR_CHECK_HIP(hipDeviceSynchronize());
  GSR_CHECK_HIP(hipMemcpyFromSymbol(out4, HIP_SYMBOL(gsr::g_bwd_counts), 4 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[4] = {0, 0, 0, 0};
    GSR_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gsr::g_bwd_counts), z, sizeof(z)));
  }
  return GSR_OK;
}
#endif

#ifdef GSR_RASTER_TIMELINE
// buf: [n_tiles, 8] uint64 on the device (or NULL to switch off)
extern "C" int gsr_debug_set_bwd_timeline(void *buf) {
  unsigned long long *p = (unsigned long long *)buf;
  GSR_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gsr::g_raster_timeline), &p, sizeof(p), sizeof(p)));
  return GSR_OK;
}
#endif
