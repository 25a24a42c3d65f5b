// gs_math.h -- per-Gaussian math of the projection / SH stages (forward and
// hand-derived backward). Pure functions on scalars, usable from HIP kernels
// and (for unit tests of the derivations only) from a host compile: the
// `tests/` harness builds this header with g++ and checks it against the
// autograd oracle without a GPU. The product never runs the host build.
//
// Conventions (stands in for what
// gsplat.rendering.rasterization computes at gs_init_compare/runner.py:341):
//   quaternions wxyz, normalised here; viewmat row-major world->camera;
//   conic = (a, b, c) = inverse(cov2d + eps2d I) as (xx, xy, yy).
#pragma once

#if defined(__HIPCC__)
#define GS_HD __host__ __device__ __forceinline__
#else
#define GS_HD inline
#include <cmath>
#endif

namespace gs {

constexpr float ALPHA_THRESHOLD = 1.0f / 255.0f;  // skip below this alpha
constexpr float ALPHA_MAX = 0.999f;                // alpha clamp
constexpr float T_THRESHOLD = 1e-4f;               // transmittance early stop
constexpr float EXTENT_MAX = 3.33f;                // gsplat 1.5.x extent cap (sigmas)

struct Mat3 {
  float m[3][3];
};

GS_HD Mat3 mat3_zero() {
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = 0.f;
  return r;
}

GS_HD Mat3 mat3_mul(const Mat3 &a, const Mat3 &b) {
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}

GS_HD Mat3 mat3_transpose(const Mat3 &a) {
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
  return r;
}

// Rotation matrix of a unit quaternion (w,x,y,z).
GS_HD Mat3 quat_to_rotmat(float w, float x, float y, float z) {
  Mat3 R;
  R.m[0][0] = 1.f - 2.f * (y * y + z * z);
  R.m[0][1] = 2.f * (x * y - w * z);
  R.m[0][2] = 2.f * (x * z + w * y);
  R.m[1][0] = 2.f * (x * y + w * z);
  R.m[1][1] = 1.f - 2.f * (x * x + z * z);
  R.m[1][2] = 2.f * (y * z - w * x);
  R.m[2][0] = 2.f * (x * z - w * y);
  R.m[2][1] = 2.f * (y * z + w * x);
  R.m[2][2] = 1.f - 2.f * (x * x + y * y);
  return R;
}

// d(loss)/d(unit quat) from d(loss)/dR.
GS_HD void quat_to_rotmat_vjp(float w, float x, float y, float z, const Mat3 &v, float vq[4]) {
  vq[0] = 2.f * (x * (v.m[2][1] - v.m[1][2]) + y * (v.m[0][2] - v.m[2][0]) +
                 z * (v.m[1][0] - v.m[0][1]));
  vq[1] = 2.f * (-2.f * x * (v.m[1][1] + v.m[2][2]) + y * (v.m[0][1] + v.m[1][0]) +
                 z * (v.m[0][2] + v.m[2][0]) + w * (v.m[2][1] - v.m[1][2]));
  vq[2] = 2.f * (x * (v.m[0][1] + v.m[1][0]) - 2.f * y * (v.m[0][0] + v.m[2][2]) +
                 z * (v.m[1][2] + v.m[2][1]) + w * (v.m[0][2] - v.m[2][0]));
  vq[3] = 2.f * (x * (v.m[0][2] + v.m[2][0]) + y * (v.m[1][2] + v.m[2][1]) -
                 2.f * z * (v.m[0][0] + v.m[1][1]) + w * (v.m[1][0] - v.m[0][1]));
}

// Sigma = (R diag(s)) (R diag(s))^T for raw quaternion q (normalised here).
GS_HD Mat3 quat_scale_to_covar(const float q[4], const float s[3]) {
  float inv = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  Mat3 R = quat_to_rotmat(q[0] * inv, q[1] * inv, q[2] * inv, q[3] * inv);
  Mat3 M;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M.m[i][j] = R.m[i][j] * s[j];
  return mat3_mul(M, mat3_transpose(M));
}

// Backward of quat_scale_to_covar: v_covar (any 3x3) -> v_quat (raw), v_scale.
GS_HD void quat_scale_to_covar_vjp(const float q[4], const float s[3], const Mat3 &v_covar,
                                   float v_q[4], float v_s[3]) {
  float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  float inv = 1.0f / sqrtf(n2);
  float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
  Mat3 R = quat_to_rotmat(w, x, y, z);
  Mat3 M;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M.m[i][j] = R.m[i][j] * s[j];
  // v_M = (v_covar + v_covar^T) M
  Mat3 sym;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) sym.m[i][j] = v_covar.m[i][j] + v_covar.m[j][i];
  Mat3 v_M = mat3_mul(sym, M);
  Mat3 v_R;
  for (int j = 0; j < 3; ++j) {
    float acc = 0.f;
    for (int i = 0; i < 3; ++i) {
      v_R.m[i][j] = v_M.m[i][j] * s[j];
      acc += R.m[i][j] * v_M.m[i][j];
    }
    v_s[j] = acc;
  }
  float vqn[4];
  quat_to_rotmat_vjp(w, x, y, z, v_R, vqn);
  // through q / |q|
  float dot = vqn[0] * w + vqn[1] * x + vqn[2] * y + vqn[3] * z;
  v_q[0] = (vqn[0] - dot * w) * inv;
  v_q[1] = (vqn[1] - dot * x) * inv;
  v_q[2] = (vqn[2] - dot * y) * inv;
  v_q[3] = (vqn[3] - dot * z) * inv;
}

struct Camera {
  float R[3][3];  // world->camera rotation (rows of viewmat[:3,:3])
  float t[3];     // viewmat[:3,3]
  float fx, fy, cx, cy;
};

GS_HD Camera load_camera(const float *viewmat /*16*/, const float *K /*9*/) {
  Camera c;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) c.R[i][j] = viewmat[i * 4 + j];
    c.t[i] = viewmat[i * 4 + 3];
  }
  c.fx = K[0];
  c.cx = K[2];
  c.fy = K[4];
  c.cy = K[5];
  return c;
}

struct Proj {
  int rx, ry;           // integer radii; 0,0 = culled
  float mx, my;         // pixel-space mean
  float depth;          // camera-space z
  float ca, cb, cc;     // conic
  float comp;           // antialias compensation
};

// Pinhole EWA projection of one Gaussian under one camera (forward).
// opacity < 0 means "no opacity given" (extent = EXTENT_MAX).
GS_HD Proj project_ewa(const Camera &cam, const float mean[3], const Mat3 &covar, float opacity,
                       int width, int height, float eps2d, float near_plane, float far_plane,
                       float radius_clip, bool comp_scales_opacity) {
  Proj o;
  o.rx = o.ry = 0;
  o.mx = o.my = o.depth = o.ca = o.cb = o.cc = 0.f;
  o.comp = 0.f;
  float pc[3];
  for (int i = 0; i < 3; ++i)
    pc[i] = cam.R[i][0] * mean[0] + cam.R[i][1] * mean[1] + cam.R[i][2] * mean[2] + cam.t[i];
  if (pc[2] < near_plane || pc[2] > far_plane) return o;
  // covar_c = R covar R^T
  Mat3 Rm;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rm.m[i][j] = cam.R[i][j];
  Mat3 cc = mat3_mul(mat3_mul(Rm, covar), mat3_transpose(Rm));
  float x = pc[0], y = pc[1], z = pc[2];
  float tanx = 0.5f * width / cam.fx, tany = 0.5f * height / cam.fy;
  float lim_xp = (width - cam.cx) / cam.fx + 0.3f * tanx;
  float lim_xn = cam.cx / cam.fx + 0.3f * tanx;
  float lim_yp = (height - cam.cy) / cam.fy + 0.3f * tany;
  float lim_yn = cam.cy / cam.fy + 0.3f * tany;
  float rz = 1.f / z, rz2 = rz * rz;
  float tx = z * fminf(lim_xp, fmaxf(-lim_xn, x * rz));
  float ty = z * fminf(lim_yp, fmaxf(-lim_yn, y * rz));
  // J = [[fx rz, 0, -fx tx rz2], [0, fy rz, -fy ty rz2]]
  float j00 = cam.fx * rz, j02 = -cam.fx * tx * rz2;
  float j11 = cam.fy * rz, j12 = -cam.fy * ty * rz2;
  // cov2d = J cc J^T
  float a0 = j00 * cc.m[0][0] + j02 * cc.m[2][0];
  float a1 = j00 * cc.m[0][1] + j02 * cc.m[2][1];
  float a2 = j00 * cc.m[0][2] + j02 * cc.m[2][2];
  float b1 = j11 * cc.m[1][1] + j12 * cc.m[2][1];
  float b2 = j11 * cc.m[1][2] + j12 * cc.m[2][2];
  float c00 = a0 * j00 + a2 * j02;
  float c01 = a1 * j11 + a2 * j12;
  float c11 = b1 * j11 + b2 * j12;
  float det_orig = c00 * c11 - c01 * c01;
  c00 += eps2d;
  c11 += eps2d;
  float det = c00 * c11 - c01 * c01;
  if (det <= 0.f) return o;
  float comp = sqrtf(fmaxf(0.f, det_orig / det));
  float extent = EXTENT_MAX;
  if (opacity >= 0.f) {
    float op = comp_scales_opacity ? opacity * comp : opacity;
    if (op < ALPHA_THRESHOLD) return o;
    extent = fminf(extent, sqrtf(2.0f * logf(op / ALPHA_THRESHOLD)));
  }
  float rxf = ceilf(extent * sqrtf(c00));
  float ryf = ceilf(extent * sqrtf(c11));
  if (rxf <= radius_clip && ryf <= radius_clip) return o;
  float mx = cam.fx * x * rz + cam.cx;
  float my = cam.fy * y * rz + cam.cy;
  if (mx + rxf <= 0.f || mx - rxf >= (float)width || my + ryf <= 0.f || my - ryf >= (float)height)
    return o;
  float idet = 1.f / det;
  o.rx = (int)rxf;
  o.ry = (int)ryf;
  o.mx = mx;
  o.my = my;
  o.depth = z;
  o.ca = c11 * idet;
  o.cb = -c01 * idet;
  o.cc = c00 * idet;
  o.comp = comp;
  return o;
}

// Backward of project_ewa for one visible (camera, Gaussian) pair.
// Inputs: v_mean2d[2], v_depth, v_conic[3] (a,b,c), v_comp (0 if unused).
// Adds into v_mean[3] (world) and v_covar (world, 3x3, generally symmetric).
GS_HD void project_ewa_vjp(const Camera &cam, const float mean[3], const Mat3 &covar, int width,
                           int height, float eps2d, const float v_mean2d[2], float v_depth,
                           const float v_conic[3], float v_comp, float v_mean[3],
                           Mat3 &v_covar) {
  float pc[3];
  for (int i = 0; i < 3; ++i)
    pc[i] = cam.R[i][0] * mean[0] + cam.R[i][1] * mean[1] + cam.R[i][2] * mean[2] + cam.t[i];
  Mat3 Rm;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rm.m[i][j] = cam.R[i][j];
  Mat3 cc = mat3_mul(mat3_mul(Rm, covar), mat3_transpose(Rm));
  float x = pc[0], y = pc[1], z = pc[2];
  float tanx = 0.5f * width / cam.fx, tany = 0.5f * height / cam.fy;
  float lim_xp = (width - cam.cx) / cam.fx + 0.3f * tanx;
  float lim_xn = cam.cx / cam.fx + 0.3f * tanx;
  float lim_yp = (height - cam.cy) / cam.fy + 0.3f * tany;
  float lim_yn = cam.cy / cam.fy + 0.3f * tany;
  float rz = 1.f / z, rz2 = rz * rz, rz3 = rz2 * rz;
  float xr = x * rz, yr = y * rz;
  bool x_in = (xr <= lim_xp) && (xr >= -lim_xn);
  bool y_in = (yr <= lim_yp) && (yr >= -lim_yn);
  float tx = z * fminf(lim_xp, fmaxf(-lim_xn, xr));
  float ty = z * fminf(lim_yp, fmaxf(-lim_yn, yr));
  float J[2][3] = {{cam.fx * rz, 0.f, -cam.fx * tx * rz2}, {0.f, cam.fy * rz, -cam.fy * ty * rz2}};
  // forward cov2d (+blur) for the inverse vjp
  float JC[2][3];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j)
      JC[i][j] = J[i][0] * cc.m[0][j] + J[i][1] * cc.m[1][j] + J[i][2] * cc.m[2][j];
  float c00 = JC[0][0] * J[0][0] + JC[0][1] * J[0][1] + JC[0][2] * J[0][2];
  float c01 = JC[0][0] * J[1][0] + JC[0][1] * J[1][1] + JC[0][2] * J[1][2];
  float c11 = JC[1][0] * J[1][0] + JC[1][1] * J[1][1] + JC[1][2] * J[1][2];
  float det_orig = c00 * c11 - c01 * c01;
  float b00 = c00 + eps2d, b11 = c11 + eps2d;
  float det = b00 * b11 - c01 * c01;
  float idet = 1.f / det;
  float ia = b11 * idet, ib = -c01 * idet, ic = b00 * idet;  // conic
  // v_inv (symmetric) = [[va, vb/2],[vb/2, vc]];  v_cov2d = -inv v_inv inv
  float va = v_conic[0], vb = 0.5f * v_conic[1], vc = v_conic[2];
  // tmp = inv * v_inv
  float t00 = ia * va + ib * vb, t01 = ia * vb + ib * vc;
  float t10 = ib * va + ic * vb, t11 = ib * vb + ic * vc;
  float g00 = -(t00 * ia + t01 * ib);
  float g01 = -(t00 * ib + t01 * ic);
  float g10 = -(t10 * ia + t11 * ib);
  float g11 = -(t10 * ib + t11 * ic);
  if (v_comp != 0.f) {
    // comp = sqrt(max(0, det_orig/det)); d comp / d cov2d (pre-blur entries)
    float comp = sqrtf(fmaxf(0.f, det_orig / det));
    float v_sqr = v_comp * 0.5f / (comp + 1e-6f);
    float om = 1.f - comp * comp;
    float det_conic = idet;
    g00 += v_sqr * (om * ia - eps2d * det_conic);
    g01 += v_sqr * (om * ib);
    g10 += v_sqr * (om * ib);
    g11 += v_sqr * (om * ic - eps2d * det_conic);
  }
  float G[2][2] = {{g00, g01}, {g10, g11}};
  // v_cc = J^T G J
  float GJ[2][3];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j) GJ[i][j] = G[i][0] * J[0][j] + G[i][1] * J[1][j];
  Mat3 v_cc;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) v_cc.m[i][j] = J[0][i] * GJ[0][j] + J[1][i] * GJ[1][j];
  // v_J = G J cc^T + G^T J cc
  float v_J[2][3];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j) {
      float s1 = 0.f, s2 = 0.f;
      for (int k = 0; k < 3; ++k) {
        s1 += GJ[i][k] * cc.m[j][k];
        float gtj = G[0][i] * J[0][k] + G[1][i] * J[1][k];
        s2 += gtj * cc.m[k][j];
      }
      v_J[i][j] = s1 + s2;
    }
  float v_pc[3];
  v_pc[0] = cam.fx * rz * v_mean2d[0];
  v_pc[1] = cam.fy * rz * v_mean2d[1];
  v_pc[2] = -(cam.fx * x * v_mean2d[0] + cam.fy * y * v_mean2d[1]) * rz2 + v_depth;
  if (x_in)
    v_pc[0] += -cam.fx * rz2 * v_J[0][2];
  else
    v_pc[2] += -cam.fx * rz3 * v_J[0][2] * tx;
  if (y_in)
    v_pc[1] += -cam.fy * rz2 * v_J[1][2];
  else
    v_pc[2] += -cam.fy * rz3 * v_J[1][2] * ty;
  v_pc[2] += -cam.fx * rz2 * v_J[0][0] - cam.fy * rz2 * v_J[1][1] +
             2.f * cam.fx * tx * rz3 * v_J[0][2] + 2.f * cam.fy * ty * rz3 * v_J[1][2];
  // world space
  for (int j = 0; j < 3; ++j)
    v_mean[j] += cam.R[0][j] * v_pc[0] + cam.R[1][j] * v_pc[1] + cam.R[2][j] * v_pc[2];
  Mat3 tmp = mat3_mul(mat3_transpose(Rm), mat3_mul(v_cc, Rm));
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) v_covar.m[i][j] += tmp.m[i][j];
}

// ---- spherical harmonics (real basis, "fast" polynomial form; A.2) --------
// basis[k] for k < (degree+1)^2 at unit direction (x,y,z).
GS_HD void sh_basis(int degree, float x, float y, float z, float b[16]) {
  b[0] = 0.2820947917738781f;
  if (degree < 1) return;
  b[1] = -0.48860251190292f * y;
  b[2] = 0.48860251190292f * z;
  b[3] = -0.48860251190292f * x;
  if (degree < 2) return;
  float z2 = z * z;
  float fTmp0B = -1.092548430592079f * z;
  float fC1 = x * x - y * y;
  float fS1 = 2.f * x * y;
  b[4] = 0.5462742152960395f * fS1;
  b[5] = fTmp0B * y;
  b[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
  b[7] = fTmp0B * x;
  b[8] = 0.5462742152960395f * fC1;
  if (degree < 3) return;
  float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
  float fTmp1B = 1.445305721320277f * z;
  float fC2 = x * fC1 - y * fS1;
  float fS2 = x * fS1 + y * fC1;
  b[9] = -0.5900435899266435f * fS2;
  b[10] = fTmp1B * fS1;
  b[11] = fTmp0C * y;
  b[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
  b[13] = fTmp0C * x;
  b[14] = fTmp1B * fC1;
  b[15] = -0.5900435899266435f * fC2;
}

// d basis[k] / d (x,y,z) (treating x,y,z as independent).
GS_HD void sh_basis_grad(int degree, float x, float y, float z, float dx[16], float dy[16],
                         float dz[16]) {
  dx[0] = dy[0] = dz[0] = 0.f;
  if (degree < 1) return;
  dx[1] = 0.f; dy[1] = -0.48860251190292f; dz[1] = 0.f;
  dx[2] = 0.f; dy[2] = 0.f; dz[2] = 0.48860251190292f;
  dx[3] = -0.48860251190292f; dy[3] = 0.f; dz[3] = 0.f;
  if (degree < 2) return;
  float z2 = z * z;
  float fTmp0B = -1.092548430592079f * z;
  float fC1 = x * x - y * y;
  float fS1 = 2.f * x * y;
  float fS1_x = 2.f * y, fS1_y = 2.f * x;
  float fC1_x = 2.f * x, fC1_y = -2.f * y;
  float fTmp0B_z = -1.092548430592079f;
  dx[4] = 0.5462742152960395f * fS1_x; dy[4] = 0.5462742152960395f * fS1_y; dz[4] = 0.f;
  dx[5] = 0.f; dy[5] = fTmp0B; dz[5] = fTmp0B_z * y;
  dx[6] = 0.f; dy[6] = 0.f; dz[6] = 2.f * 0.9461746957575601f * z;
  dx[7] = fTmp0B; dy[7] = 0.f; dz[7] = fTmp0B_z * x;
  dx[8] = 0.5462742152960395f * fC1_x; dy[8] = 0.5462742152960395f * fC1_y; dz[8] = 0.f;
  if (degree < 3) return;
  float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
  float fTmp1B = 1.445305721320277f * z;
  float fTmp0C_z = -2.285228997322329f * 2.f * z;
  float fTmp1B_z = 1.445305721320277f;
  float fC2_x = fC1 + x * fC1_x - y * fS1_x;
  float fC2_y = x * fC1_y - fS1 - y * fS1_y;
  float fS2_x = fS1 + x * fS1_x + y * fC1_x;
  float fS2_y = x * fS1_y + fC1 + y * fC1_y;
  dx[9] = -0.5900435899266435f * fS2_x; dy[9] = -0.5900435899266435f * fS2_y; dz[9] = 0.f;
  dx[10] = fTmp1B * fS1_x; dy[10] = fTmp1B * fS1_y; dz[10] = fTmp1B_z * fS1;
  dx[11] = 0.f; dy[11] = fTmp0C; dz[11] = fTmp0C_z * y;
  dx[12] = 0.f; dy[12] = 0.f;
  dz[12] = (1.865881662950577f * z2 - 1.119528997770346f) + z * (2.f * 1.865881662950577f * z);
  dx[13] = fTmp0C; dy[13] = 0.f; dz[13] = fTmp0C_z * x;
  dx[14] = fTmp1B * fC1_x; dy[14] = fTmp1B * fC1_y; dz[14] = fTmp1B_z * fC1;
  dx[15] = -0.5900435899266435f * fC2_x; dy[15] = -0.5900435899266435f * fC2_y; dz[15] = 0.f;
}


// Visitor form: calls f(k, b, dbx, dby, dbz) for k = 0 .. (degree+1)^2-1 with
// the basis value and its partial derivatives, one coefficient at a time, so
// a consumer never holds the four 16-entry arrays at once (register pressure
// of the fused backward).
template <typename F>
GS_HD void sh_visit(int degree, float x, float y, float z, F &&f) {
  f(0, 0.2820947917738781f, 0.f, 0.f, 0.f);
  if (degree < 1) return;
  f(1, -0.48860251190292f * y, 0.f, -0.48860251190292f, 0.f);
  f(2, 0.48860251190292f * z, 0.f, 0.f, 0.48860251190292f);
  f(3, -0.48860251190292f * x, -0.48860251190292f, 0.f, 0.f);
  if (degree < 2) return;
  const float z2 = z * z;
  const float fTmp0B = -1.092548430592079f * z;
  const float fC1 = x * x - y * y;
  const float fS1 = 2.f * x * y;
  f(4, 0.5462742152960395f * fS1, 0.5462742152960395f * 2.f * y, 0.5462742152960395f * 2.f * x, 0.f);
  f(5, fTmp0B * y, 0.f, fTmp0B, -1.092548430592079f * y);
  f(6, 0.9461746957575601f * z2 - 0.3153915652525201f, 0.f, 0.f, 2.f * 0.9461746957575601f * z);
  f(7, fTmp0B * x, fTmp0B, 0.f, -1.092548430592079f * x);
  f(8, 0.5462742152960395f * fC1, 0.5462742152960395f * 2.f * x, -0.5462742152960395f * 2.f * y, 0.f);
  if (degree < 3) return;
  const float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
  const float fTmp1B = 1.445305721320277f * z;
  const float fTmp0C_z = -2.285228997322329f * 2.f * z;
  const float fC2 = x * fC1 - y * fS1;
  const float fS2 = x * fS1 + y * fC1;
  const float fC2_x = 3.f * fC1;          // fC1 + 2x^2 - 2y^2
  const float fC2_y = -3.f * fS1;         // -2xy - fS1 - 2xy
  const float fS2_x = 3.f * fS1;          // fS1 + 2xy + 2xy
  const float fS2_y = 3.f * fC1;          // 2x^2 + fC1 - 2y^2
  f(9, -0.5900435899266435f * fS2, -0.5900435899266435f * fS2_x, -0.5900435899266435f * fS2_y, 0.f);
  f(10, fTmp1B * fS1, fTmp1B * 2.f * y, fTmp1B * 2.f * x, 1.445305721320277f * fS1);
  f(11, fTmp0C * y, 0.f, fTmp0C, fTmp0C_z * y);
  f(12, z * (1.865881662950577f * z2 - 1.119528997770346f), 0.f, 0.f,
    3.f * 1.865881662950577f * z2 - 1.119528997770346f);
  f(13, fTmp0C * x, fTmp0C, 0.f, fTmp0C_z * x);
  f(14, fTmp1B * fC1, fTmp1B * 2.f * x, -fTmp1B * 2.f * y, 1.445305721320277f * fC1);
  f(15, -0.5900435899266435f * fC2, -0.5900435899266435f * fC2_x, -0.5900435899266435f * fC2_y, 0.f);
}

}  // namespace gs
