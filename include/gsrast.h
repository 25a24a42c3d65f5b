/*
 * gsrast.h -- C ABI of libgsrast.so, the MI355X (gfx950) hot path of
 * deivse/3dgs_monocular_depth_init.
 *
 * Every entry point takes raw DEVICE pointers, sizes, scalar parameters and a
 * HIP stream (passed as void*; NULL = the null stream). Nothing here knows
 * about torch. All functions are asynchronous on `stream`, allocate nothing
 * (the caller owns every buffer, including scratch) and return 0 on success
 * or a negative GSR_E* code; gsr_last_error() gives the message of the last
 * failure on the calling thread.
 *
 * What each group replaces in the reference (paths under /root/reference):
 *   - gsr_project_*, gsr_isect_*, gsr_tile_sort, gsr_rasterize_*:
 *       the third-party call `gsplat.rendering.rasterization(...)` made at
 *       gs_init_compare/runner.py:341-362 (again via rasterize_splats from
 *       nerfbaselines_integration/method.py:754,829) and its autograd
 *       backward triggered by `loss.backward()` at runner.py:547.
 *   - gsr_lstsq_*, gsr_ransac_*:
 *       gs_init_compare/depth_alignment/alignment/lstsqrs.py:9-54 and
 *       gs_init_compare/depth_alignment/alignment/ransacs.py:100-189.
 *   - gsr_subsample_*, gsr_sfm_patch_mask, gsr_unproject_*:
 *       gs_init_compare/depth_subsampling/static_subsampler.py:8-22,
 *       adaptive_subsampling.py:48-122, num_sfm_points_mask.py:7-64 and
 *       gs_init_compare/depth_prediction/points_from_depth.py:111-180,270-312.
 *
 * Layout conventions: row-major, fp32 unless stated, indices int32.
 *   N  = number of Gaussians, C = cameras in this batch, g = c*N + i is the
 *   flat (camera, Gaussian) index, CH = colour channels composited (1..5).
 */
#ifndef GSRAST_H
#define GSRAST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_OK 0
#define GSR_EINVAL (-1)   /* bad argument (shape / null pointer / unsupported value) */
#define GSR_EHIP (-2)     /* a HIP runtime call or kernel launch failed              */
#define GSR_ECAPACITY (-3)/* caller-provided buffer too small                         */

#define GSR_TILE 16       /* tile edge in pixels (gsplat tile_size default)            */
#define GSR_GRAD_ROW 16   /* floats per (camera,Gaussian) row of the gradient scratch  */
/* gradient-row field offsets (floats): */
#define GSR_GR_MEAN2D 0   /* 2: dL/d means2d (pixels)        */
#define GSR_GR_CONIC 2    /* 3: dL/d conic (a,b,c)           */
#define GSR_GR_OPAC 5     /* 1: dL/d opacity                 */
#define GSR_GR_COLOR 6    /* CH (<=5): dL/d colour channels  */
#define GSR_GR_ABS 12     /* 2: sum |dL/d means2d| (absgrad) */
#define GSR_PACKED_ROW_H 5 /* DWORDS of a half-packed row (gsr_pack_grad_rows_h): int16 exponent + 9 halves */
#define GSR_PACKED_ROW 9  /* floats of a packed row (gsr_pack_grad_rows): slots 0..8 as above;  */
                          /*   an all-zero row = invisible pair (or one that contributes nothing) */
/* `activations` bits of gsr_project_fwd/bwd: the reference's A1 step
 * (gs_init_compare/runner.py:324-325) fused into the kernels. */
#define GSR_ACT_EXP_SCALES 1   /* `scales` holds log-scales: scale = exp(raw)          */
#define GSR_ACT_SIGMOID_OPAC 2 /* `opacities` holds logits: opacity = sigmoid(raw)     */

int gsr_version(void);
const char *gsr_last_error(void);
/* Name of the code-object architecture the library was built for ("gfx950"). */
const char *gsr_arch(void);

/* ---------------------------------------------------------------------------
 * A3 + A4: 3D->2D EWA projection fused with SH colour evaluation.
 * One thread per (camera, Gaussian).
 * sh0/shN may be NULL (no SH: colours are supplied by the caller elsewhere).
 * sh0 points at coefficient 0 of Gaussian 0, consecutive Gaussians are
 * sh0_stride floats apart; shN points at coefficient 1, shN_stride apart
 * (a concatenated [N,K,3] tensor is sh0=p, shN=p+3, both strides 3K).
 * colors_out is [C*N, color_stride]; RGB goes to channels 0..2 and, when
 * depth_channel >= 0, the camera-space depth to that channel.
 * --------------------------------------------------------------------------*/
int gsr_project_fwd(int C, int N, const float *means, const float *quats, const float *scales,
                    const float *opacities /* [N] or NULL */, const float *viewmats /* [C,4,4] */,
                    const float *Ks /* [C,3,3] */, const float *campos /* [C,3] */, int width,
                    int height, float eps2d, float near_plane, float far_plane, float radius_clip,
                    int calc_compensations, int sh_degree /* -1: no SH */, const float *sh0,
                    int sh0_stride, const float *shN, int shN_stride,
                    int32_t *radii /* [C,N,2] */, float *means2d /* [C,N,2] */,
                    float *depths /* [C,N] */, float *conics /* [C,N,3] */,
                    float *compensations /* [C,N] or NULL */, float *colors_out /* or NULL */,
                    int color_stride, int depth_channel, int activations /* GSR_ACT_* */,
                    float *opacities_out /* [N] activated opacities (with GSR_ACT_SIGMOID_OPAC) */,
                    int tile_w, int tile_h,
                    int32_t *tile_counts /* NULL, or [C*tile_h*tile_w]: fused gsr_isect_count */,
                    float *records /* NULL, or [C*N,16]: packed compositing records */,
                    void *stream);

/* Backward of gsr_project_fwd. grad_rows is the [C*N, GSR_GRAD_ROW] scratch
 * filled by gsr_rasterize_bwd (fields GSR_GR_*). v_depths [C,N] may be NULL
 * (then the depth gradient is taken from colour channel depth_channel if
 * >= 0). v_compensations [C,N] may be NULL. Outputs are WRITTEN (not
 * accumulated); camera contributions are summed inside the kernel. */
int gsr_project_bwd(int C, int N, const float *means, const float *quats, const float *scales,
                    const float *viewmats, const float *Ks, const float *campos, int width,
                    int height, float eps2d, int sh_degree, const float *sh0, int sh0_stride,
                    const float *shN, int shN_stride, const int32_t *radii, const float *conics,
                    const float *compensations, const float *grad_rows, const float *v_depths,
                    const float *v_compensations, int depth_channel, float *v_means /* [N,3] */,
                    float *v_quats /* [N,4] */, float *v_scales /* [N,3] */,
                    float *v_sh0 /* or NULL */, int v_sh0_stride, float *v_shN, int v_shN_stride,
                    int sh_K /* coefficients stored per Gaussian */, int activations,
                    const float *opacities_act /* [N], needed with GSR_ACT_SIGMOID_OPAC */,
                    float *v_opacities /* NULL, or [N]: sum over cameras of grad_rows[.][GSR_GR_OPAC]
                                          (times o(1-o) with GSR_ACT_SIGMOID_OPAC) */,
                    void *stream);

/* Multi-GPU exchange of VIEW-SPACE gradients: instead of all-reducing
 * the 59 floats per Gaussian of parameter gradients, every rank packs the 9 floats per Gaussian
 * its compositing backward produced (36 bytes; rows of invisible pairs are zero), the ranks all-gather the
 * packed rows, and each rank runs gsr_project_bwd_adam over ALL ranks' cameras (C = world size):
 * the same sum of per-view gradients, 6x fewer bytes on the wire, and the Adam update stays
 * fused in the backward. packed [n,GSR_PACKED_ROW]; rows of invisible pairs are zeroed, and the
 * backward skips all-zero rows. */
int gsr_pack_grad_rows(int64_t n, const float *grad_rows /* [n,16] */, const int32_t *radii,
                       float *packed, void *stream);
/* The same rows in 20 bytes: a shared power-of-two exponent (int16) and the 9 values as IEEE halves
 * (value = half * 2^exponent). Consumed by gsr_project_bwd_adam / gsr_project_bwd_rows with
 * grad_stride = GSR_PACKED_ROW_H. Opt-in: the gradients then carry 11 significant bits. */
int gsr_pack_grad_rows_h(int64_t n, const float *grad_rows, const int32_t *radii, void *packed,
                         void *stream);

/* gsr_project_bwd over gathered PACKED rows (grad_stride = GSR_PACKED_ROW, radii NULL:
 * visibility from the row) or scratch rows (GSR_GRAD_ROW, radii required): the non-fused
 * companion of gsr_project_bwd_adam for the steps on which something must see the parameter
 * gradients between backward and optimizer step (the densification strategy on refine /
 * reset steps, runner.py:638-679). Colour-only views (no depth / compensation gradients). */
int gsr_project_bwd_rows(int C, int N, const float *means, const float *quats, const float *scales,
                         const float *viewmats, const float *Ks, const float *campos, int width,
                         int height, float eps2d, int sh_degree, const float *sh0, int sh0_stride,
                         const float *shN, int shN_stride, const int32_t *radii,
                         const float *grad_rows, int grad_stride, float *v_means, float *v_quats,
                         float *v_scales, float *v_sh0, int v_sh0_stride, float *v_shN,
                         int v_shN_stride, int sh_K, int activations, const float *opacities_act,
                         float *v_opacities, void *stream);

/* Optimizer in backward (A8 fused into the backward of A1+A3+A4): gsr_project_bwd and
 * gsr_adam_step in ONE pass over the parameters, for the single-process case where nothing
 * (no all-reduce, no other loss term) has to see the parameter gradients. params /
 * exp_avg / exp_avg_sq: HOST arrays of 6 device pointers in the order
 * means [N,3], quats [N,4], raw scales [N,3], raw opacities [N], sh0 [N,1,3], shN [N,15,3];
 * step_size[6] = lr/(1-beta1^t), bc2_sqrt[6] = sqrt(1-beta2^t) (HOST). The parameters are
 * read as the inputs of the backward and updated in place; no gradient is written.
 * Requires activations == GSR_ACT_EXP_SCALES | GSR_ACT_SIGMOID_OPAC. */
int gsr_project_bwd_adam(int C, int N, const float *viewmats, const float *Ks, const float *campos,
                         int width, int height, float eps2d, int sh_degree,
                         const int32_t *radii /* NULL with packed rows: visibility from the row */,
                         const float *grad_rows,
                         int grad_stride /* GSR_GRAD_ROW, or GSR_PACKED_ROW for gathered rows */,
                         const float *v_depths,
                         const float *v_compensations, int depth_channel, int activations,
                         const float *opacities_act, void *const *params, void *const *exp_avg,
                         void *const *exp_avg_sq, const float *step_size, const float *bc2_sqrt,
                         double beta1, double beta2, double eps, void *stream);
/* What else a training step does to every Gaussian, riding along in the same pass (all optional):
 *  - the reference's "mcmc" preset (trainer.py:83-92): the position noise of gsplat's
 *    MCMCStrategy.step_post_backward -> inject_noise_to_position (runner.py:649-656; means += covar . (noise *
 *    gate(1 - opacity) * noise_scale), from the PRE-update parameters, applied before the Adam update of the means as
 *    in the reference's strategy-then-optimizer order) and the gradients of the two regularisers
 *    opacity_reg * mean(sigmoid(opacities)) + scale_reg * mean(exp(scales)) (runner.py:535-545);
 *  - gsplat's DefaultStrategy._update_state (runner.py:639-647): grad2d[i] += |(g.x sx, g.y sy)| (g = the 2-D mean
 *    gradient of the row, or its absgrad fields), count[i] += 1 per camera rendering i, radii[i] = max(radii[i],
 *    max(rx, ry) * inv_max_wh); needs the fp32 scratch rows and the radii. */
typedef struct gsr_step_extras {
  const float *noise;        /* NULL or [N,3] standard-normal draws */
  double noise_scale;        /* lr(means) * noise_lr */
  double opacity_reg, scale_reg;
  float *stat_grad2d, *stat_count /* [N], NULL: no statistics */, *stat_radii /* [N] or NULL */;
  double stat_sx, stat_sy, stat_inv_max_wh;
  int stat_use_absgrad;
} gsr_step_extras;
int gsr_project_bwd_adam_ex(int C, int N, const float *viewmats, const float *Ks, const float *campos,
                            int width, int height, float eps2d, int sh_degree, const int32_t *radii,
                            const float *grad_rows, int grad_stride, const float *v_depths,
                            const float *v_compensations, int depth_channel, int activations,
                            const float *opacities_act, void *const *params, void *const *exp_avg,
                            void *const *exp_avg_sq, const float *step_size, const float *bc2_sqrt,
                            double beta1, double beta2, double eps, const gsr_step_extras *extras /* or NULL */,
                            void *stream);

/* ---------------------------------------------------------------------------
 * A5: per-tile depth-sorted intersection lists.
 * tile ids are c*tile_h*tile_w + ty*tile_w + tx; n_tiles = C*tile_h*tile_w.
 * gsr_isect_count : tile_counts[n_tiles] (zeroed inside) and, optionally,
 *                   tiles_per_gauss[C*N].
 * gsr_isect_scan  : exclusive scan -> tile_offsets[n_tiles+1] (last = n_isects);
 *                   optionally tile_order[n_tiles] = tile ids, longest list first
 *                   (the work order of gsr_rasterize_fwd/bwd).
 * gsr_isect_emit  : scatter (depth_bits<<32 | g) into the tile buckets.
 *                   tile_cursor[n_tiles] is scratch (zeroed inside).
 * gsr_tile_sort   : sort every bucket ascending (depth, then g) in place and
 *                   write flatten_ids[n_isects] = g. big_list is scratch for
 *                   the queue of buckets too long for the 16 KB LDS sorter.
 * --------------------------------------------------------------------------*/
int gsr_isect_count(int C, int N, const float *means2d, const int32_t *radii, int tile_w,
                    int tile_h, int32_t *tiles_per_gauss /* or NULL */, int32_t *tile_counts,
                    void *stream);
int gsr_isect_scan(int n_tiles, const int32_t *tile_counts, int32_t *tile_offsets,
                   int32_t *tile_order /* [n_tiles] or NULL */, void *stream);
/* Same scan; the counts are zeroed once read (buffer reused as a cursor, kept across frames).
 * total_host: NULL, or a device-accessible HOST address (pinned memory) that receives offsets[n]. */
int gsr_isect_scan_clear(int n, int32_t *counts, int32_t *offsets, int32_t *order,
                         int32_t *total_host, void *stream);
int gsr_isect_emit(int C, int N, const float *means2d, const int32_t *radii, const float *depths,
                   int tile_w, int tile_h, const int32_t *tile_offsets, int32_t *tile_cursor,
                   uint64_t *isect_keys, int64_t capacity, void *stream);
int gsr_tile_sort(int n_tiles, const int32_t *tile_offsets,
                  const int32_t *tile_order /* or NULL */, uint64_t *isect_keys,
                  int32_t *flatten_ids, int32_t *big_list /* scratch [n_tiles+1] */,
                  void *stream);

/* ---------------------------------------------------------------------------
 * A6 / A7: alpha compositing, one wave64 per 16x16 tile (four 8x8 quadrants,
 * one pixel of each per lane). backgrounds [C,CH] or NULL. last_ids [C,H,W] int32 is
 * the position (in flatten_ids) after which nothing is blended into a pixel: the entry
 * before the one at which the pixel's transmittance fell to 1e-4, else the last entry of
 * the tile's list (first entry - 1 for an empty list). tile_order (from gsr_isect_scan) is the order workgroups take tiles in;
 * NULL = natural order.
 * --------------------------------------------------------------------------*/
/* A5, bucketed variant (isect_bucket.hip): same outputs as count/scan/emit/sort
 * above -- tile_offsets, flatten_ids in (depth, g) order per tile -- without one
 * global atomic per intersection: the first digit is a bucket of 8 tiles along x.
 * It also produces what the compositing kernels read: pair_ids[i] = g | mask << 28,
 * mask = the pair's 4-bit quadrant mask (bit q: the ellipse alpha >= 1/255 reaches the
 * 8x8 quadrant q of the tile; exact test, csrc/raster_common.h).
 *   gsr_bucket_layout : buckets per row (bw) and in total; GSR_ECAPACITY when the
 *                       total exceeds the LDS histogram (8192) -> use the calls above.
 *   gsr_bucket_count  : bucket_counts[n_buckets] under gsplat's rule (every tile the rectangle
 *                       mean +- radius touches); also zeroes clear_a / clear_b (the emit
 *                       pass's cursor and real counts) when given.
 *   gsr_bucket_emit   : scans the counts (bucket_offsets[n_buckets+1], bucket_order, and the
 *                       number of reserved slots -> total_host[0]), evaluates the exact pair
 *                       test and scatters the composite keys (tile-in-bucket | depth | g |
 *                       mask) into the buckets. Needs C*N < 2^26.
 *   gsr_bucket_sort   : sorts every bucket in LDS and writes, compacted, flatten_ids[I],
 *                       pair_ids[I], (optional) keys_sorted[I], tile_offsets[n_tiles+1], the
 *                       number of listed pairs I -> total_host[0], and (optional)
 *                       tile_order[n_tiles].
 * tight = 0: gsplat's lists, bit for bit. tight = 1 (runner.rasterize_splats): pairs whose
 *   mask is 0 -- no pixel of the tile can reach alpha >= 1/255 -- are not listed; same image
 *   and gradients, fewer pairs to sort, gather and composite; the slots the count pass had
 *   reserved for them hold sentinel keys that the sort pass skips.
 * conics [C,N,3] (natural units), opacities [N] or [C,N]. keys / flatten_ids / pair_ids hold
 * `capacity` entries (>= the reserved slots, else the lists are truncated: check total_host). */
int gsr_bucket_layout(int C, int tile_w, int tile_h, int *bw_out, int *n_buckets_out);
int gsr_bucket_count(int C, int N, const float *means2d, const int32_t *radii, int tile_w,
                     int tile_h, int32_t *bucket_counts, int32_t *clear_a /* NULL or [n_buckets] */,
                     int32_t *clear_b /* NULL or [n_buckets] */,
                     int assume_zero /* 1: bucket_counts is already zero, no memset launch */,
                     int32_t *wg_hist /* NULL or [256, n_buckets]: each workgroup's own counts, which
                                         gsr_bucket_emit (same grid) then reserves from without a second walk */,
                     void *stream);
int gsr_bucket_emit(int C, int N, const float *means2d, const int32_t *radii, const float *depths,
                    const float *conics, const float *opacities, int opac_per_camera, int tile_w,
                    int tile_h, int tight, const int32_t *bucket_counts,
                    int32_t *bucket_cursor /* [n_buckets], zero */,
                    int32_t *real_counts /* [n_buckets], zero; out: listed pairs per bucket */,
                    int32_t *bucket_offsets /* out [n_buckets+1] */,
                    int32_t *bucket_order /* out [n_buckets] or NULL */,
                    int32_t *tile_order /* out [n_tiles] or NULL: compositing work order, longest
                                           bucket first (gsr_bucket_sort can produce the tile-exact one) */,
                    int32_t *total_host /* NULL or host-visible: reserved slots */, uint64_t *keys,
                    int64_t capacity, const int32_t *wg_hist /* NULL or what gsr_bucket_count wrote */, void *stream);
int gsr_bucket_sort(int C, int tile_w, int tile_h, const int32_t *bucket_offsets,
                    const int32_t *bucket_order, const int32_t *real_counts, uint64_t *keys,
                    uint64_t *keys_sorted /* NULL or [capacity] */, int32_t *flatten_ids,
                    int32_t *pair_ids, int32_t *tile_offsets, int32_t *tile_order,
                    int64_t capacity /* entries in keys / flatten_ids / pair_ids; offsets are clamped to it */,
                    int32_t *clear_counts /* NULL, or [n_buckets]: zeroed per bucket on the way out
                                             (the count buffer kept across frames) */,
                    int32_t *total_host /* NULL or host-visible: listed pairs */,
                    int32_t *done_host /* NULL or host-visible coherent int32[3]: [0] reserved slots, [1] listed pairs,
                                          then [2] = seq stored LAST with system-scope release -- a host may poll [2]
                                          for this frame's seq instead of waiting on an event */,
                    int seq, void *stream);
/* Pair words for lists built by gsr_isect_* / gsr_tile_sort (or by the caller): one launch
 * that evaluates the quadrant mask of every (tile, Gaussian) entry. Needs C*N < 2^28. */
int gsr_pair_masks(int C, int N, int tile_w, int tile_h, const int32_t *tile_offsets,
                   const int32_t *flatten_ids, const float *means2d, const float *conics,
                   const float *opacities, int opac_per_camera, int32_t *pair_ids, void *stream);

/* Compositing reads ONE packed 64-byte record per (camera, Gaussian), with the conic in
 * the units of the compositing loops (ha = a/2 * log2 e, bb = b * log2 e, hc = c/2 * log2 e):
 *   float[16] = {mx, my, ha, bb | hc, opacity, col0, col1 | col2, col3, col4, - | pad}
 * gsr_project_fwd writes the records on the SH path; gsr_pack_records builds them
 * from separate arrays (caller-supplied colours; opacities [N] or [C,N]; natural-unit
 * conics). records must be 16-byte aligned (rows are copied to LDS by LDS-DMA).
 * pair_ids: see above (replaces gsplat's flatten_ids at this boundary; the low 28 bits ARE
 * flatten_ids). Replaces gsplat's rasterize_to_pixels (gs_init_compare/runner.py:341). */
#define GSR_REC_FLOATS 16
int gsr_pack_records(int C, int N, int CH, const float *means2d, const float *conics,
                     const float *colors, int color_stride, const float *opacities,
                     int opac_per_camera, float *records, void *stream);
int gsr_rasterize_fwd(int C, int CH, const float *records, const float *backgrounds, int width,
                      int height, int tile_w, int tile_h, const int32_t *tile_offsets,
                      const int32_t *tile_order, const int32_t *pair_ids,
                      float *render_colors, float *render_alphas, int32_t *last_ids,
                      float *zero_rows /* NULL, or the [n_zero_rows, 16] grad_rows buffer the backward of
                                          this render will accumulate into: cleared here, on the side */,
                      int64_t n_zero_rows, void *stream);
/* gsr_rasterize_fwd / _bwd with the colour image (render_colors, v_render_colors) laid out in planes [C,CH,H,W] instead of
 * [C,H,W,CH]: the fused L1 + SSIM loss (gsr_ssim_l1_fwd / _bwd take element strides per image) reads and writes a plane
 * at a time, and on channel-interleaved memory every line is fetched / written by all three planes' workgroups
 * (forward 0.058 -> 0.052 ms, backward 0.064 -> 0.042 ms at 1080p). Same arguments as the functions they stand in for. */
int gsr_rasterize_fwd_planar(int C, int CH, const float *records, const float *backgrounds, int width,
                             int height, int tile_w, int tile_h, const int32_t *tile_offsets,
                             const int32_t *tile_order, const int32_t *pair_ids, float *render_colors,
                             float *render_alphas, int32_t *last_ids, float *zero_rows, int64_t n_zero_rows,
                             void *stream);
int gsr_rasterize_bwd_planar(int C, int CH, const float *records, const float *backgrounds, int width,
                             int height, int tile_w, int tile_h, const int32_t *tile_offsets,
                             const int32_t *tile_order, const int32_t *pair_ids, const float *render_alphas,
                             const int32_t *last_ids, const float *v_render_colors,
                             const float *v_render_alphas, int absgrad, float *grad_rows, void *stream);
/* gsr_rasterize_fwd for a training step whose loss is the plain L1 (F.l1_loss(colors, pixels), runner.py:506, three
 * channels): the loss is taken while the finished pixels are in registers. grad_out [C,H,W,3] receives
 * d mean|render - target| / d render = sign(render - target) / (C H W 3) -- what gsr_rasterize_bwd takes as
 * v_render_colors under a root gradient of 1 -- and mean_out[0] (device) the loss; the render itself is not written.
 * l1_partials: C*tile_h*tile_w device doubles (scratch). Replaces gsr_rasterize_fwd + gsr_l1_fwd of that step. */
int gsr_rasterize_fwd_l1(int C, const float *records, const float *backgrounds, int width, int height,
                         int tile_w, int tile_h, const int32_t *tile_offsets, const int32_t *tile_order,
                         const int32_t *pair_ids, const float *target /* [C,H,W,3] */, float *grad_out,
                         float *render_alphas, int32_t *last_ids, float *zero_rows, int64_t n_zero_rows,
                         double *l1_partials, float *mean_out, void *stream);
int gsr_rasterize_bwd(int C, int CH, const float *records, const float *backgrounds, int width,
                      int height, int tile_w, int tile_h, const int32_t *tile_offsets,
                      const int32_t *tile_order, const int32_t *pair_ids,
                      const float *render_alphas, const int32_t *last_ids,
                      const float *v_render_colors,
                      const float *v_render_alphas /* NULL = no gradient on the alphas */,
                      int absgrad,
                      float *grad_rows /* [C*N,16], zero on entry (by the caller, or by the forward's
                                          zero_rows), accumulated */, void *stream);

/* ---------------------------------------------------------------------------
 * A8 / F2: fused multi-tensor Adam over the Gaussian parameters, one launch
 * for all tensors (replaces the six torch.optim.Adam steps of
 * gs_init_compare/runner.py:129-137, 676-679; same update as
 * torch.optim.Adam without weight decay / amsgrad). The pointer arrays and the
 * per-tensor scalars are HOST arrays of n <= 8 entries; the tensors they point
 * to are device memory, 16-byte aligned. step_size[i] = lr_i / (1 - beta1^t_i),
 * bc2_sqrt[i] = sqrt(1 - beta2^t_i).
 * --------------------------------------------------------------------------*/
int gsr_adam_step(int n, void *const *params, const void *const *grads, void *const *exp_avg,
                  void *const *exp_avg_sq, const int64_t *numel, const float *step_size,
                  const float *bc2_sqrt, double beta1, double beta2, double eps, void *stream);

/* ---------------------------------------------------------------------------
 * F2: DefaultStrategy refine step in one pass (duplicate -> split -> prune of
 * gsplat.strategy.ops, driven by gs_init_compare/runner.py:639-647; thresholds
 * gs_init_compare/config.py:204-221).
 * gsr_refine_decide: flags4 int32 [5,N] = {original kept, duplicate kept, split children kept,
 *   is split, is duplicated} from the strategy's statistics and the raw (log / logit) parameters. Thresholds are
 *   absolute (already multiplied by scene_scale); grow_scale2d / prune_scale2d < 0 switch the
 *   screen-space tests off (step >= refine_scale2d_stop_iter), prune_scale3d < 0 the size prune
 *   (step <= reset_every).
 * gsr_refine_plan: from the inclusive scans incl4 (rows 0..2 are read) of flags4 and the totals n0, n1, n2 of
 *   rows 0..2: src_row / kind for the M = n0 + n1 + 2*n2 output rows (kind 0 original,
 *   1 duplicate, 2 / 3 split child with noise sample 0 / 1).
 * gsr_refine_gather: dst[t][r] = src[t][src_row[r]] for n tensors in one launch; tensors with
 *   zero_new[t] (Adam moments) get zeros in rows of kind != 0. HOST pointer arrays, n <= 24.
 * --------------------------------------------------------------------------*/
int gsr_refine_decide(int N, const float *log_scales, const float *logit_opacities, const float *grad2d,
                      const float *count, const float *radii_state /* or NULL */, float grow_grad2d,
                      float grow_scale3d, float grow_scale2d, float prune_opa, float prune_scale3d,
                      float prune_scale2d, int revised_opacity, int32_t *flags4, void *stream);
int gsr_refine_plan(int N, const int32_t *flags4, const int32_t *incl4, int n0, int n1, int n2,
                    int32_t *src_row, uint8_t *kind, void *stream);
int gsr_refine_gather(int n, int M, const int32_t *src_row, const uint8_t *kind, const void *const *src,
                      void *const *dst, const int32_t *row_len, const int32_t *zero_new, void *stream);

/* ---------------------------------------------------------------------------
 * F1: fused L1 + SSIM loss (gs_init_compare/runner.py:506-510; replaces the
 * third-party `fused_ssim`, setup.py:14). Images are logical [N,CH,H,W] fp32
 * addressed by ELEMENT strides (HOST int64[4]), so NHWC renders are used in
 * place. 11x11 Gaussian window (sigma 1.5), zero padding, C1=0.01^2, C2=0.03^2.
 * fwd: out[3] (device float) = {mean SSIM over the counted region (all pixels, or the map
 *      cropped by 5 px when valid_only), mean |img1-img2|, the reference's loss
 *      (1-ssim_lambda)*L1 + ssim_lambda*(1-SSIM) of runner.py:506-510}, finalised on the
 *      device; workspace = gsr_ssim_workspace_doubles(N,CH,H,W) device doubles (per-workgroup
 *      partial sums, any content on entry); dm_* [N,CH,H,W] receive
 *      dSSIM/d{mu1,sigma1^2,sigma12} (all three or all NULL).
 * bwd: grad = w_l1*sign(img1-img2) + w_ssim*dSSIM/dimg1, (w_ssim, w_l1) = weights[0..1]
 *      (device float[2]) or, weights NULL, upstream[0] * (scale_ssim, scale_l1) with upstream
 *      the device scalar autograd hands the loss (NULL = 1).
 * --------------------------------------------------------------------------*/
int64_t gsr_ssim_workspace_doubles(int N, int CH, int H, int W);
int gsr_ssim_l1_fwd(int N, int CH, int H, int W, const float *img1, const int64_t *strides1,
                    const float *img2, const int64_t *strides2, int valid_only,
                    double *workspace, float *out, float ssim_lambda, float *dm_mu1,
                    float *dm_s1, float *dm_s12, void *stream);
int gsr_ssim_l1_bwd(int N, int CH, int H, int W, const float *img1, const int64_t *strides1,
                    const float *img2, const int64_t *strides2, const float *dm_mu1,
                    const float *dm_s1, const float *dm_s12, const float *weights,
                    const float *upstream, float scale_ssim, float scale_l1, float *grad,
                    const int64_t *stridesg, void *stream);

/* Plain L1 over n contiguous floats (16-byte aligned) (replaces F.l1_loss, runner.py:506):
 * gsr_l1_fwd: mean_out[0] (device float) = mean |a-b|. workspace = GSR_L1_WS_DOUBLES device
 *   doubles, any content (one partial sum per workgroup; a second, one-workgroup launch adds them).
 *   unit_grad (NULL or n floats) receives sign(a-b)/n, the gradient w.r.t. a under an upstream
 *   gradient of 1: the backward is then free unless the loss is scaled further.
 * gsr_l1_bwd: grad = (upstream ? upstream[0] : 1) * scale * sign(a-b); upstream is the device
 *   scalar autograd hands the loss, scale = 1/n. */
#define GSR_L1_WS_DOUBLES 512
int gsr_l1_fwd(int64_t n, const float *a, const float *b, double *workspace, float *mean_out,
               float *unit_grad, void *stream);
int gsr_l1_bwd(int64_t n, const float *a, const float *b, const float *upstream, float scale,
               float *grad, void *stream);

/* Test hook: in [8][64] -> out[0..63] = per-lane result of the 8-value lane-swap
 * reduction tree used by gsr_rasterize_bwd, out[64..127] = wave sum of in[0],
 * idx_out[64] = which input each lane holds the total of. */
int gsr_debug_tree_reduce8(const float *in, float *out, int32_t *idx_out, void *stream);

/* Batched inverse of C row-major 4x4 matrices (viewmats = inv(camtoworlds),
 * gs_init_compare/runner.py:347). in_translation / out_translation [C,3] (optional) receive
 * the translation column of the INPUT / of the INVERSE: the camera positions when `in` is
 * camera-to-world / world-to-camera. */
int gsr_inverse4x4(int C, const float *in, float *out, float *in_translation,
                   float *out_translation, void *stream);

/* DefaultStrategy statistics (gsplat's DefaultStrategy._update_state, which the
 * reference drives at gs_init_compare/runner.py:639-647), one launch, no host sync: for every
 * (camera, Gaussian) pair with both radii > 0
 *   grad2d[i] += hypot(g.x*sx, g.y*sy);  count[i] += 1;
 *   radii_state[i] = max(radii_state[i], max(rx, ry) / max_wh)          (radii_state may be NULL)
 * with g the pair's means2d gradient at grad[(c*N+i)*grad_stride + {0,1}] (grad_stride = 2 for a
 * dense [C,N,2] tensor, 16 for the view into the 64-byte gradient rows), sx = width/2*C,
 * sy = height/2*C. */
int gsr_strategy_accumulate(int C, int N, const float *grad, int grad_stride, const int32_t *radii,
                            float sx, float sy, float *grad2d, float *count, float *radii_state,
                            float max_wh, void *stream);

/* MCMC densification strategy (the reference drives gsplat's MCMCStrategy at
 * gs_init_compare/runner.py:214-215, 649-658 with the "mcmc" preset of trainer.py:83-92).
 * gsr_relocation replaces gsplat's `compute_relocation` CUDA op: for i < n,
 *   N = clamp(ratios[i], 1, n_max); new_opacities[i] = 1 - (1 - opacities[i])^(1/N);
 *   new_scales[i,:] = scales[i,:] * opacities[i] /
 *       sum_{a=1..N} sum_{k=0..a-1} binoms[(a-1)*n_max + k] * (-1)^k / sqrt(k+1) * new_opacity^(k+1)
 * (opacities / scales are ACTIVATED values, binoms is the [n_max,n_max] table C(n,k)).
 * gsr_inject_noise replaces `inject_noise_to_position`: in place,
 *   means[i] += Sigma_i * noise[i] * scaler / (1 + exp(-100 * ((1 - sigmoid(logit_opac[i])) - 0.995)))
 * with Sigma_i from the un-normalised quats [N,4] (wxyz) and exp(log_scales [N,3]). */
int gsr_relocation(int n, const float *opacities, const float *scales, const int32_t *ratios,
                   const float *binoms, int n_max, float *new_opacities, float *new_scales,
                   void *stream);
int gsr_inject_noise(int N, float *means, const float *quats, const float *log_scales,
                     const float *logit_opacities, const float *noise, float scaler, void *stream);
/* gsplat.strategy.ops.reset_opa in place: logit_opacities[i] = min(logit_opacities[i], max_logit) and the
 * opacity optimizer's moments (either may be NULL) cleared (DefaultStrategy.reset_every). */
int gsr_reset_opacity(int64_t n, float *logit_opacities, float *exp_avg, float *exp_avg_sq, float max_logit,
                      void *stream);
/* torch.optim.SparseAdam over the rows a step rendered (cfg.sparse_grad: runner.py:130 builds SparseAdam,
 * runner.py:661-672 turns every gradient into a sparse tensor over info["gaussian_ids"]): for each of the n
 * tensors [rows, row_len[t]], row r is updated iff visible[r] != 0 -- parameter and both moments; all other
 * rows are untouched. visible[r] = the number of cameras that render row r (<= 255): the row steps on that
 * many times its dense gradient, as SparseAdam's coalesce() of the per-(camera, Gaussian) entries does. step_size[t] = lr sqrt(1 - beta2^t) / (1 - beta1^t); the arithmetic order is
 * torch/optim/_functional.py sparse_adam's. HOST pointer arrays, n <= 8. */
int gsr_sparse_adam_step(int n, int64_t rows, const uint8_t *visible, void *const *params,
                         const void *const *grads, void *const *exp_avg, void *const *exp_avg_sq,
                         const int32_t *row_len, const float *step_size, double beta1, double beta2,
                         double eps, void *stream);

/* ---------------------------------------------------------------------------
 * Init path (monocular depth -> seed point cloud), SURVEY.md rows B1-B9.
 * coords arrays are int64 [2,M], row 0 = x, row 1 = y (the reference's
 * sfm_points_camera_coords). Masks are uint8 (0/1) [H*W].
 * --------------------------------------------------------------------------*/
/* B1 (points_from_depth.py:111-180): project M SfM points with P [3,4],
 * round half-to-even to integer pixels, inbounds[i] = in image && z >= 0,
 * valid[i] = inbounds && pred_mask[y,x]; coords of out-of-bounds points are
 * zeroed as in the reference. depth_out[i] = z. */
int gsr_project_sfm(int M, const float *pts /* [M,3] */, const float *P, int W, int H,
                    const uint8_t *pred_mask, int64_t *coords, float *depth_out,
                    uint8_t *inbounds, uint8_t *valid, void *stream);
/* out[i] = depth_map[y_i, x_i]. */
int gsr_gather_depth(int M, const float *depth_map, int depth_w, const int64_t *coords,
                     float *out, void *stream);
/* B2/B3: normal-equation sums of (d,1) against g (lstsqrs.py:22-25) for T
 * subsets of the M correspondences, fp64: sums_out[T,5] = {sum d^2, sum d,
 * count, sum d*g, sum g}. mode 0: all M (T = 1); mode 1: the S sample indices
 * sample_idx[T,S]; mode 2: the inliers ((s*d+t-g)^2 < thr) of hyp[T,2]. */
int gsr_lsq_sums(int T, int M, int mode, const float *d, const float *g,
                 const int64_t *sample_idx, int S, const float *hyp, float thr, double *sums_out,
                 void *stream);
/* hyp[T,2] = pinv([[Sdd,Sd],[Sd,n]]) @ [Sdg,Sg]  (scale, shift), fp64 -> fp32. */
int gsr_solve_scale_shift(int T, const double *sums, float *hyp, void *stream);
/* B3 (ransacs.py:60-65, 94-97): squared residuals of T hypotheses over all M
 * in one launch: out_ransac[T] = #(r >= thr), out_msac[T] = sum min(r, thr),
 * out_inliers[T] = #(r < thr). */
int gsr_ransac_score(int T, int M, const float *hyp, const float *d, const float *g, float thr,
                     int32_t *out_ransac, float *out_msac, int32_t *out_inliers, void *stream);
/* aligned = depth*hs[0] + hs[1] (two rounded fp32 ops), hs on the device. */
int gsr_affine_depth(int64_t n, const float *depth, const float *hs, float *out, void *stream);

/* B5/B6: keep[i] = y%f==0 && x%f==0 && valid[i]. mode 0: f = static_k
 * (static_subsampler.py:8-22). mode 1 (adaptive_subsampling.py:89-122):
 * m = 1 - clamp((d-lo)/(hi-lo),0,1) (0.5 where !valid), f = trunc(clamp(fmin +
 * (fmax-fmin)*m, fmin, fmax)), lo/hi = range[0..1] on the device. */
int gsr_subsample_mask(int H, int W, int mode, int static_k, const float *depth,
                       const uint8_t *valid, const float *range, int fmin, int fmax,
                       uint8_t *keep, void *stream);
/* B7 (num_sfm_points_mask.py:38-64): patch_counts[gh*gw] (zeroed inside) =
 * histogram of the M points over patches of (ph,pw) pixels; mask[i] = 0 where
 * the pixel's patch holds > threshold points. */
int gsr_sfm_patch_mask(int H, int W, int M, const int64_t *coords, int ph, int pw, int gh,
                       int gw, int threshold, int32_t *patch_counts, uint8_t *mask, void *stream);
/* F4 tail (depth_alignment/alignment/interp.py:77-110): scipy's LinearNDInterpolator evaluated on
 * the integer pixel grid: xy fp64 [P,2] vertex coordinates (x, y), tris int32 [n_tri,3] (scipy's
 * Delaunay simplices), values fp64 [P]; writes out[y*W + x] for every pixel inside a triangle
 * (`out` should be pre-filled with the fill value). */
int gsr_tri_interp(int H, int W, int n_tri, const double *xy, const int32_t *tris, const double *values,
                   float *out, void *stream);

/* ---------------------------------------------------------------------------
 * F4: interp.method = "rbf" (gs_init_compare/depth_alignment/alignment/interp.py:30-72): the reference fits a
 * torchrbf.RBFInterpolator (= scipy.interpolate.RBFInterpolator's algorithm, no neighbours) to the scale
 * factors at the SfM pixels normalised to [0,1]^2, evaluates it on a grid 256 pixels wide (x-major) and
 * upsamples bilinearly with align_corners. kernel: 0 linear, 1 thin_plate_spline (the config's default),
 * 2 cubic; polynomial degree = the kernel's minimum (0 / 1 / 1); epsilon = 1. All float64.
 * gsr_rbf_fit: sites_xy [P,2] / values [P] fp32 -> coeffs [P+3] (P + 1 used by the linear kernel) and
 *   shift_scale [4] = {shift_x, shift_y, scale_x, scale_y}; workspace of gsr_rbf_workspace_bytes(P).
 *   Dense LU with partial pivoting; synchronises the stream once (singular-system report).
 * gsr_rbf_eval_grid: out [qw, qh] fp32 = the interpolant at (a / (qw-1), b / (qh-1)).
 * gsr_bilinear_ac_t: out [H, W] = F.interpolate(src[None, None] as [qw, qh], (W, H), "bilinear",
 *   align_corners=True)[0, 0].T
 * --------------------------------------------------------------------------*/
int64_t gsr_rbf_workspace_bytes(int P);
int gsr_rbf_fit(int P, const float *sites_xy, const float *values, double smoothing, int kernel, void *workspace,
                int64_t workspace_bytes, double *coeffs, double *shift_scale, void *stream);
int gsr_rbf_eval_grid(int P, const float *sites_xy, const double *coeffs, const double *shift_scale, int kernel,
                      int qw, int qh, float *out, void *stream);
int gsr_bilinear_ac_t(int qw, int qh, const float *src, int W, int H, float *out, void *stream);
/* F4 tail (depth_alignment/segmentation/region_margin.py:21-35, `calculate_region_margin_mask`):
 * mask[i] = 1 where the (2*half_width+1)^2 box mean of the int32 label map (replicate padding),
 * snapped to the nearest integer when torch.isclose to it, equals the pixel's own label -- the
 * pixels farther than half_width from every region boundary. row_sums: int64 [H*W] scratch. */
int gsr_region_margin_mask(int H, int W, int half_width, const int32_t *labels, int64_t *row_sums,
                           uint8_t *mask, void *stream);
/* B8 (points_from_depth.py:203-208): |dx| + |dy| backward differences. */
int gsr_depth_grad(int H, int W, const float *depth, float *grad, void *stream);
/* B9 (points_from_depth.py:270-312): fused mask -> ordered stream compaction ->
 * unprojection. keep = valid & subsample & (depth >= 0) [& extra].
 * gsr_unproject_count writes block_counts[gsr_unproject_num_blocks(H,W)];
 * the caller scans them with gsr_isect_scan; gsr_unproject_emit then writes
 * pts[n,3] (world), rgb_out[n,3] (optional) in pixel order -- identical to
 * boolean-mask indexing -- and final_mask[H*W] (optional). Kinv [3,3],
 * c2w [4,4] are device pointers. */
int gsr_unproject_num_blocks(int H, int W);
int gsr_unproject_count(int H, int W, const float *depth, const uint8_t *valid,
                        const uint8_t *subsample, const uint8_t *extra, int32_t *block_counts,
                        void *stream);
int gsr_unproject_emit(int H, int W, const float *depth, const uint8_t *valid,
                       const uint8_t *subsample, const uint8_t *extra, const float *rgb,
                       const float *Kinv, const float *c2w, const int32_t *block_offsets,
                       float *pts, float *rgb_out, uint8_t *final_mask, void *stream);

/* F3 / A9: exact K-nearest-neighbour DISTANCES (K = 4 or 8, the point itself
 * included), replacing sklearn's NearestNeighbors of utils/runner_utils.py:142-146.
 * gsr_knn_cell_keys: 64-bit cell key of every point for cubic cells of edge h
 *   (origin = per-axis minimum, device float[3]).
 * gsr_knn_grid: N queries against the points sorted by key (sorted_pts), unique
 *   keys ukeys[U] with start offsets ustart[U+1] into sorted_pts; ring-by-ring
 *   exact search up to max_ring (<= 8) rings; out[N,K] in ORIGINAL order,
 *   ascending; unresolved[N] = 1 for queries whose result is not proven exact
 *   within max_ring (finish those with gsr_knn_brute).
 * gsr_knn_brute: Q queries against all N points (one workgroup per query). */
int gsr_knn_cell_keys(int N, const float *pts, const float *origin, float h, int64_t *keys,
                      void *stream);
int gsr_knn_grid(int N, int K, const float *queries /* [N,3] */, const float *sorted_pts,
                 const int64_t *order /* result row of query i */,
                 const int64_t *ukeys, const int64_t *ustart, int U, const float *origin, float h,
                 int max_ring, float *out, uint8_t *unresolved, void *stream);
int gsr_knn_brute(int Q, int N, int K, const float *queries, const float *pts, float *out,
                  void *stream);
/* F4 tail (point_cloud_postprocess/postprocess.py:16-22: sklearn LocalOutlierFactor on the CPU).
 * gsr_knn_grid_idx: like gsr_knn_grid with K <= 64 and the neighbours' indices: out_dist fp64
 *   [*,K] ascending (sqrt of fp64 squared distances), out_idx int32 [*,K] = sorted_ids of the
 *   neighbours; self_pos[i] (or NULL) = position in sorted_pts of query i itself, skipped;
 *   qorder[i] (or NULL: i) = output row of query i; unresolved as in gsr_knn_grid.
 * gsr_lof: local reachability density and negative outlier factor (sklearn/neighbors/_lof.py
 *   `_local_reachability_density`, `fit`): lrd[N] scratch/out, negative_outlier_factor[N] or NULL,
 *   outlier[i] = negative_outlier_factor < offset (-1.5 for contamination="auto"). */
int gsr_knn_grid_idx(int Q, int K, const float *queries, const int64_t *self_pos, const float *sorted_pts,
                     const int64_t *sorted_ids, const int64_t *qorder, const int64_t *ukeys,
                     const int64_t *ustart, int U, const float *origin, float h, int max_ring,
                     double *out_dist, int32_t *out_idx, uint8_t *unresolved, void *stream);
int gsr_lof(int N, int K, const double *dist, const int32_t *idx, double offset, double *lrd,
            double *negative_outlier_factor, uint8_t *outlier, void *stream);

/* B10: Metric3D pre/post-processing (depth_prediction/predictors/metric3d.py:42-83,
 * 96-131) around the depth network. preprocess: float RGB [H,W,3] in [0,1] ->
 * uint8, channel flip, bilinear resize to (rh,rw), mean-colour border to
 * (out_h,out_w), (x-mean)/std, planar [3,out_h,out_w]. postprocess: un-pad one
 * [in_h,in_w] map, bilinear upsample (align_corners=False) to (H,W), * scale,
 * optional clamp to [lo,hi]. */
int gsr_m3d_preprocess(int H, int W, const float *img, int rh, int rw, int pad_top, int pad_left,
                       int out_h, int out_w, float *out, void *stream);
int gsr_m3d_postprocess(int in_h, int in_w, const float *in, int pad_top, int pad_bot,
                        int pad_left, int pad_right, int H, int W, float scale, float lo, float hi,
                        int do_clamp, float *out, void *stream);

/* ---------------------------------------------------------------------------
 * B10: the dense monocular depth network on the matrix cores (csrc/depthnet.hip).
 * Replaces the torch.hub Metric3D v2 model behind `model.inference({"input": rgb})`
 * (gs_init_compare/depth_prediction/predictors/metric3d.py:27-31, 87-88); architecture:
 * third_party/metric3d/mono/model/backbones/ViT_DINO_reg.py:755-1270 and
 * .../decode_heads/RAFTDepthNormalDPTDecoder5.py:736-1035. All `void *` operands are fp16
 * (IEEE half) device buffers, `float *` fp32; activations are row-major [rows, ld] with
 * rows = tokens or pixels (NHWC maps), weights [N, K] as nn.Linear / flattened conv weights
 * with K padded to a multiple of 64 by zeros.
 * --------------------------------------------------------------------------*/
/* out = residual + gamma * act(A[M,K] W[N,K]^T + bias): fp16 MFMA (v_mfma_f32_16x16x32_f16 in the
 * 256x256 eight-phase core, v_mfma_f32_32x32x16_f16 in the 128-row tiles), fp32 accumulate.
 * act: 0 none, 1 GELU(erf), 2 ReLU, 3 sigmoid, 4 tanh. bias/gamma [N] fp32 or NULL; residual
 * fp32 [M,ldr] and/or residual16 fp16 [M,ldr16] or NULL (may alias the outputs); out16 and/or
 * out32 receive the result. K % 64 == 0, lda >= K, lda % 8 == 0. out16_pad_to: 0, or the number of
 * columns of out16 (<= ldo16, <= N rounded up to 64) of which [N, out16_pad_to) are written as
 * zeros -- the zero channels of a map that a 3x3 convolution will read (no fill launch). */
int gsr_dn_gemm(int M, int N, int K, const void *A, int lda, const void *W, const float *bias, int act,
                const float *gamma, const float *residual, int ldr, const void *residual16, int ldr16,
                void *out16, int ldo16, float *out32, int ldo32, int out16_pad_to, void *stream);
/* The same GEMM with A read as the im2col view of an NHWC fp16 map [H*W, ldi] (KS = 1 or 3,
 * stride 1, "same" padding, C % 64 == 0): row = output pixel, column = tap*C + c, gathered straight
 * from the map by the LDS-DMA staging (no im2col buffer). zero_page: >= 16 bytes of zeros. */
int gsr_dn_conv_gemm(int H, int W, int C, const void *in, int ldi, int KS, int N, int K_pad,
                     const void *Wt, const float *bias, int act, const void *residual16, int ldr16,
                     void *out16, int ldo16, const void *zero_page, int out16_pad_to, void *stream);
/* out32[p, 0:N] += bias + conv3x3(in)[p, 0:N] for N <= 8 output channels (stride 1, "same" padding): the
 * flow head's last layers (RAFTDepthNormalDPTDecoder5.py:282-297), which add 2 / 4 channels to the
 * fp32 flow field. in: NHWC fp16 map [H*W, ldi] with C % 8 == 0 channels; Wt [N, K_pad] fp16 in im2col
 * column order (tap * C + c); no im2col buffer, no GEMM tile of which 2 columns are used. */
int gsr_dn_conv3_head(int H, int W, int C, const void *in, int ldi, int N, const void *Wt, int K_pad,
                      const float *bias, float *out32, int ldo, void *stream);
/* gsr_dn_conv_gemm over a VIRTUAL CONCATENATION: channels [0, c_split) of the input come from `first` (row
 * stride ld_first), channels [c_split, C) from `in`, whose first c_split channels are never read. c_split a
 * multiple of 64. The ConvGRU's torch.cat([h, x]) / torch.cat([r * h, x]) (RAFTDepthNormalDPTDecoder5.py:318-330)
 * without copying the hidden state into the concatenated input. */
int gsr_dn_conv_gemm2(int H, int W, int C, const void *in, int ldi, const void *first, int ld_first, int c_split,
                      int KS, int N, int K_pad, const void *Wt, const float *bias, int act, const void *residual16,
                      int ldr16, void *out16, int ldo16, const void *zero_page, int out16_pad_to, void *stream);
/* LayerNorm over the last dimension of [M,D] (fp32 or fp16 input), optional ReLU. */
int gsr_dn_layernorm(int M, int D, const void *x, int ldx, int x_is_f16, const float *gamma,
                     const float *beta, float eps, void *out16, int ldo16, float *out32, int ldo32,
                     int relu, void *stream);
/* Multi-head self-attention, head_dim 64 (Attention.forward, ViT_DINO_reg.py:430-470):
 * qkv fp16 [n_tok, ld >= 3*heads*64] as the qkv Linear writes it ([3][heads][64] per token);
 * vt_scratch fp16 [heads*64*n_pad] (n_pad = n_tok rounded up to 64); out fp16 [n_tok, ldo]. */
int gsr_dn_attention(int n_tok, int n_pad, int heads, const void *qkv, int ld, void *vt_scratch,
                     float scale, void *out, int ldo, void *stream);
/* Patch-embedding rows of an NCHW fp32 image [3,H,W]: [H/P*W/P, K_pad] fp16, column
 * c*P*P + ky*P + kx (PatchEmbed.proj, ViT_DINO_reg.py:212). */
int gsr_dn_patch_rows(int H, int W, int P, int K_pad, const float *img, void *rows, void *stream);
/* im2col rows of a KSxKS convolution over an NHWC fp16 map [H*W, ldi] (C channels used):
 * [Ho*Wo, K_pad], column (ky*KS + kx)*C + c; relu != 0 applies ReLU to the gathered values. */
int gsr_dn_im2col(int H, int W, int C, int ldi, int KS, int stride, int pad, int Ho, int Wo, int K_pad,
                  const void *in, void *rows, int relu, void *stream);
/* NHWC resize: mode 0 nearest, 1 bilinear align_corners=True, 2 bilinear align_corners=False. */
int gsr_dn_resize(int Hi, int Wi, int C, const void *in, int ldi, int Ho, int Wo, void *out, int ldo,
                  int mode, void *stream);
/* F.avg_pool2d(x, 3, stride=2, padding=1) on an NHWC map (pool2x, decoder :333). */
int gsr_dn_avgpool3s2(int Hi, int Wi, int C, const void *in, int ldi, void *out, int ldo, void *stream);
/* out[p, 0:C] = act(a * in[p, 0:C] (+ out[p, 0:C])) over strided channel slices. */
int gsr_dn_slice(int64_t P, int C, const void *in, int ldi, void *out, int ldo, float a, int accumulate,
                 int act, void *stream);
/* SwiGLU gate of vit_giant2_reg's FFN (ViT_DINO_reg.py:335-345): out[p, c] = silu(x12[p, c]) *
 * x12[p, h + c] for c < h; fp16 in and out. */
int gsr_dn_swiglu(int64_t P, int h, const void *x12, int ldx, void *out, int ldo, void *stream);
/* ConvGRU gate algebra (decoder :318-330). stage 0: zr = [convz(hx) | convr(hx)] [P,2C], ctx =
 * [cz | cr | cq] [P,3C]: writes z = sigmoid(.) and rh = sigmoid(.) * h. stage 1: zr = convq([r*h, x])
 * [P,C]: h = (1 - z) h + z tanh(zr + cq). */
int gsr_dn_gru_gate(int64_t P, int C, int stage, const void *zr, int ldzr, const void *ctx, int ldc,
                    void *h, int ldh, void *z, int ldz, void *rh, int ldrh, void *stream);
/* regress_depth (decoder :806-838): softmax over `bins` logits, expectation over log-spaced bins in
 * [min_val, max_val], clamp, (d - max_val) / regress_scale -> out[p*ldo]. */
int gsr_dn_depth_expectation(int64_t P, int bins, const void *logits, int ld, float min_val,
                             float max_val, float regress_scale, float *out, int ldo, void *stream);
/* pred_normal's norm_normalize (decoder :252-258, 840-850): out[p*ldo + 0..3]. */
int gsr_dn_normal_head(int64_t P, const void *nrm, int ldn, const void *conf, int ldc, float *out,
                       int ldo, void *stream);
/* upsample_flow (decoder :870-884) fused with the output heads (:985-987): flow fp32 [H*W,6],
 * mask fp16 [H*W, ldm >= 9 F F] -> depth, confidence [H F, W F] and normal [4, H F, W F] fp32. */
int gsr_dn_convex_upsample(int H, int W, int F, const float *flow, const void *mask, int ldm,
                           float min_val, float max_val, float regress_scale, float *depth,
                           float *conf, float *normal, void *stream);
int gsr_dn_cvt_f32_f16(int64_t P, int C, const float *in, int ldi, void *out, int ldo, void *stream);

/* ---------------------------------------------------------------------------
 * F4: the reference's native point-cloud subsampler (native_modules/subsampling, C++/Eigen,
 * pybind `_pointcloud_subsampling.subsample_pointcloud`, pointcloud_subsampling.cpp:22-67).
 * --------------------------------------------------------------------------*/
/* compute_minimal_gaussian_extents (impl.cpp:70-126): extents[i] = min over the cameras that see
 * point i of 2*depth/min(fx,fy); -1 when none does. Ks [C,3,3], Ps [C,3,4] (K R [I|-C]),
 * image_sizes [C,2] = (width, height). */
int gsr_pc_min_extents(int N, int C, const float *points, const float *Ks, const float *Ps,
                       const int32_t *image_sizes, float *extents, void *stream);
/* subsample_pointcloud_impl (impl.cpp:313-426): spatial-median tree over the points' bounding cube,
 * nodes merged into their mean when compact enough. Outputs are written in tree (bit-path) order:
 * out_points / out_rgbs [<= N,3], *out_count on the device. workspace: device scratch of at least
 * gsr_pc_subsample_workspace_bytes(N) bytes (returns -1 for N < 0). */
/* The subsampler's key sort by itself: (uint64 key, uint32 value) pairs sorted in place by key, ascending,
 * stable (hand-written LSD radix sort, eight 8-bit passes). keys_alt / vals_alt [n] and
 * hist [256 * ceil(n / 2048)] int32 are scratch. */
int gsr_sort_pairs_u64(int64_t n, uint64_t *keys, uint64_t *keys_alt, uint32_t *vals, uint32_t *vals_alt,
                       int32_t *hist, int64_t hist_ints, void *stream);
int64_t gsr_pc_subsample_workspace_bytes(int N);
int gsr_pc_subsample(int N, const float *points, const float *rgbs, const float *extents,
                     float max_bbox_aspect_ratio, float min_extent_multiplier, void *workspace,
                     int64_t workspace_bytes, float *out_points, float *out_rgbs, int32_t *out_count,
                     void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSRAST_H */
