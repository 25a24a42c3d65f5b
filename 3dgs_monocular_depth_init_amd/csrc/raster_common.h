// raster_common.h -- pieces shared by the compositing forward and backward.
//
// Wave/tile shape (both kernels): ONE wave64 per 16x16 tile. The tile is cut
// into four 8x8 quadrants; lane l owns pixel (l & 7, l >> 3) of EVERY quadrant
// (4 pixels per lane). For each Gaussian of the tile list the staging lane
// pre-computes a 4-bit quadrant mask by an exact ellipse-vs-rectangle test
// (min of sigma over the quadrant's pixel centres <= ln(255*opacity), i.e. at
// least one pixel could reach alpha >= 1/255). The mask is wave-uniform, so
// the per-quadrant body is skipped with SCALAR branches: with ~5 px radii a
// Gaussian touches ~2 of the 4 quadrants, which halves the VALU work without
// changing any result (a skipped quadrant has alpha < 1/255 everywhere).
#pragma once
#include "common.h"
#include "gs_math.h"

namespace gsr {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float PIX_DONE = 1.0e18f;   // pixel x of a finished / outside pixel: sigma -> huge, alpha -> 0

template <int CH>
struct TileRec {      // LDS image of one staged Gaussian
  float4 a;           // mx, my, ha = 0.5*a*log2e, bb = b*log2e
  float4 b;           // hc = 0.5*c*log2e, opacity, col0, col1
  float4 c;           // col2, col3, col4, quadrant mask (int bits)
};

// sigma' = log2(e) * sigma = ha*dx^2 + bb*dx*dy + hc*dy^2, evaluated identically in
// forward and backward as dx*(ha*dx + B) + C with the row terms B = bb*dy, C = hc*dy*dy:
// the two quadrants of a row share dy, so B and C cost 3 VALU per (Gaussian, row) and
// sigma 2 per pixel (the kernels issue one VALU instruction per 4 cycles per SIMD and
// are bound by exactly that, so instruction counts are what the layout is chosen for).
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

// min over the rectangle of pixel centres [x0,x1]x[y0,y1] of
// sigma(d) = 0.5*(a dx^2 + c dy^2) + b dx dy, d = mean - pixel.
__device__ __forceinline__ float min_sigma_rect(float a, float b, float c, float mx, float my,
                                                float x0, float x1, float y0, float y1) {
  // (x1 - x0 and y1 - y0 are compile-time constants at the call sites: written so that the
  // wave-uniform x0 / y0 are each used once and can stay in scalar registers)
  const float dxhi = mx - x0, dyhi = my - y0;
  const float dxlo = dxhi - (x1 - x0), dylo = dyhi - (y1 - y0);
  if (dxlo <= 0.f && dxhi >= 0.f && dylo <= 0.f && dyhi >= 0.f) return 0.f;
  auto sig = [&](float dx, float dy) { return 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy; };
  const float ic = 1.0f / c, ia = 1.0f / a;
  float m = sig(dxlo, fminf(fmaxf(-b * dxlo * ic, dylo), dyhi));
  m = fminf(m, sig(dxhi, fminf(fmaxf(-b * dxhi * ic, dylo), dyhi)));
  m = fminf(m, sig(fminf(fmaxf(-b * dylo * ia, dxlo), dxhi), dylo));
  m = fminf(m, sig(fminf(fmaxf(-b * dyhi * ia, dxlo), dxhi), dyhi));
  return m;
}

// Packed per-(camera,Gaussian) compositing record, one 64-byte row (= one cache
// line / one fabric request per gather instead of four scattered arrays):
//   [0] mx  my  conic.a conic.b   [1] conic.c opacity col0 col1   [2] col2 col3 col4 -
// written by gsr_project_fwd (or gsr_pack_records for caller-supplied colours).
constexpr int REC_FLOATS = 16;

// The gather and the LDS image are split so that the kernels can software-pipeline:
// the loads of batch b+1 are issued before the compositing loop of batch b and only
// consumed (make_rec) after it, so their latency is covered by the wave's own work.
template <int CH>
struct RawRec {
  float4 r0, r1, r2;
};

template <int CH>
__device__ __forceinline__ void load_raw(int g, const float *__restrict__ records, RawRec<CH> &w) {
  const float4 *row = reinterpret_cast<const float4 *>(records + (int64_t)g * REC_FLOATS);
  w.r0 = row[0];
  w.r1 = row[1];
  if (CH > 3) w.r2 = row[2];
  else if (CH > 2) w.r2.x = records[(int64_t)g * REC_FLOATS + 8];
}

// Build the LDS image of a gathered record for the tile at (tx0, ty0).
template <int CH>
__device__ __forceinline__ void make_rec(const RawRec<CH> &w, float tx0, float ty0, TileRec<CH> &r) {
  const float4 r0 = w.r0, r1 = w.r1;
  const float a = r0.z, b = r0.w, c = r1.x, op = r1.y;
  // alpha >= 1/255  <=>  sigma <= ln(255*op); small margin keeps the test conservative
  const float tau = logf(op * 255.0f);
  const float tau_m = tau + 1e-4f * (1.0f + fabsf(tau));
  int qmask = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = tx0 + (8.f * (float)(q & 1) + 0.5f), y0 = ty0 + (8.f * (float)(q >> 1) + 0.5f);
    const float ms = min_sigma_rect(a, b, c, r0.x, r0.y, x0, x0 + 7.f, y0, y0 + 7.f);
    qmask |= (ms <= tau_m) ? (1 << q) : 0;
  }
  r.a = make_float4(r0.x, r0.y, 0.5f * a * LOG2E, b * LOG2E);
  r.b = make_float4(0.5f * c * LOG2E, op, r1.z, r1.w);
  r.c = make_float4(CH > 2 ? w.r2.x : 0.f, CH > 3 ? w.r2.y : 0.f, CH > 4 ? w.r2.z : 0.f,
                    __int_as_float(qmask));
}

}  // namespace gsr
