// Host build of csrc/gs_math.h for unit tests of the hand-derived math
// (compiled with g++ by tests/test_host_math.py; never part of the product).
#include <cstdint>
#include <cmath>
#include "gs_math.h"

extern "C" {

void hm_project_fwd(int N, const float *means, const float *quats, const float *scales,
                    const float *opac, const float *viewmat, const float *K, int W, int H,
                    float eps2d, float near_p, float far_p, float radius_clip, int comp_flag,
                    int32_t *radii, float *means2d, float *depths, float *conics, float *comps) {
  gs::Camera cam = gs::load_camera(viewmat, K);
  for (int i = 0; i < N; ++i) {
    gs::Mat3 cov = gs::quat_scale_to_covar(quats + 4 * i, scales + 3 * i);
    gs::Proj p = gs::project_ewa(cam, means + 3 * i, cov, opac ? opac[i] : -1.f, W, H, eps2d,
                                 near_p, far_p, radius_clip, comp_flag != 0);
    radii[2 * i] = p.rx; radii[2 * i + 1] = p.ry;
    means2d[2 * i] = p.mx; means2d[2 * i + 1] = p.my;
    depths[i] = p.depth;
    conics[3 * i] = p.ca; conics[3 * i + 1] = p.cb; conics[3 * i + 2] = p.cc;
    comps[i] = p.comp;
  }
}

void hm_project_bwd(int N, const float *means, const float *quats, const float *scales,
                    const float *viewmat, const float *K, int W, int H, float eps2d,
                    const int32_t *radii, const float *v_means2d, const float *v_depths,
                    const float *v_conics, const float *v_comps, float *v_means, float *v_quats,
                    float *v_scales) {
  gs::Camera cam = gs::load_camera(viewmat, K);
  for (int i = 0; i < N; ++i) {
    float vm[3] = {0, 0, 0};
    gs::Mat3 vc = gs::mat3_zero();
    gs::Mat3 cov = gs::quat_scale_to_covar(quats + 4 * i, scales + 3 * i);
    if (radii[2 * i] > 0)
      gs::project_ewa_vjp(cam, means + 3 * i, cov, W, H, eps2d, v_means2d + 2 * i, v_depths[i],
                          v_conics + 3 * i, v_comps ? v_comps[i] : 0.f, vm, vc);
    float vq[4], vs[3];
    gs::quat_scale_to_covar_vjp(quats + 4 * i, scales + 3 * i, vc, vq, vs);
    for (int k = 0; k < 3; ++k) { v_means[3 * i + k] = vm[k]; v_scales[3 * i + k] = vs[k]; }
    for (int k = 0; k < 4; ++k) v_quats[4 * i + k] = vq[k];
  }
}

// colours = sum_k basis_k(dir) * coeff_k  (no +0.5 / clamp)
void hm_sh_fwd(int N, int degree, const float *dirs, const float *coeffs /*[N,16,3]*/,
               float *out) {
  for (int i = 0; i < N; ++i) {
    float dx = dirs[3 * i], dy = dirs[3 * i + 1], dz = dirs[3 * i + 2];
    float inv = 1.f / std::sqrt(dx * dx + dy * dy + dz * dz);
    float b[16];
    gs::sh_basis(degree, dx * inv, dy * inv, dz * inv, b);
    float acc[3] = {0, 0, 0};
    // cross-check the visitor against the array form
    gs::sh_visit(degree, dx * inv, dy * inv, dz * inv,
                 [&](int k, float bk, float, float, float) {
                   for (int c = 0; c < 3; ++c) acc[c] += 0.5f * (bk + b[k]) * coeffs[(i * 16 + k) * 3 + c];
                 });
    for (int c = 0; c < 3; ++c) out[3 * i + c] = acc[c];
  }
}

void hm_sh_bwd(int N, int degree, const float *dirs, const float *coeffs, const float *v_out,
               float *v_coeffs /*[N,16,3]*/, float *v_dirs) {
  for (int i = 0; i < N; ++i) {
    float dx = dirs[3 * i], dy = dirs[3 * i + 1], dz = dirs[3 * i + 2];
    float nrm = std::sqrt(dx * dx + dy * dy + dz * dz), inv = 1.f / nrm;
    float ux = dx * inv, uy = dy * inv, uz = dz * inv;
    float vx = 0, vy = 0, vz = 0;
    for (int k = 0; k < 48; ++k) v_coeffs[i * 48 + k] = 0.f;
    gs::sh_visit(degree, ux, uy, uz, [&](int k, float b, float bx, float by, float bz) {
      const float *ck = coeffs + (i * 16 + k) * 3;
      float dotc = ck[0] * v_out[3 * i] + ck[1] * v_out[3 * i + 1] + ck[2] * v_out[3 * i + 2];
      vx += bx * dotc; vy += by * dotc; vz += bz * dotc;
      for (int c = 0; c < 3; ++c) v_coeffs[(i * 16 + k) * 3 + c] = b * v_out[3 * i + c];
    });
    float dot = vx * ux + vy * uy + vz * uz;
    v_dirs[3 * i] = (vx - dot * ux) * inv;
    v_dirs[3 * i + 1] = (vy - dot * uy) * inv;
    v_dirs[3 * i + 2] = (vz - dot * uz) * inv;
  }
}
}
