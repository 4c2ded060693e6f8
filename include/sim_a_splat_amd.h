/*
 * sim_a_splat_amd.h -- C ABI of the MI355X (gfx950) Gaussian-splat rasterizer that sits behind
 * sim_a_splat's render-image calls.
 *
 * The reference reaches its renderer through two Python call sites and has no native FFI of its
 * own, so every entry point cites the reference call it serves
 * (paths relative to the reference tree):
 *
 *   Door A  GaussianSplat.render(pose) -> pipeline.model.get_outputs_for_camera(cameras, obb_box=None)
 *           sim_a_splat/ns_utils/nerfstudio_utils.py:123-177 (call at :166-172)
 *   Door B  client.get_render(height, width, wxyz, position)
 *           sim_a_splat/env/splat/splat_env_wrapper.py:148-157, sim_a_splat/splat/splat_handler.py:339-344
 *           scene.add_gaussian_splats(...) registration      sim_a_splat/splat/splat_handler.py:106-141
 *           handle.wxyz / handle.position updates             sim_a_splat/splat/splat_handler.py:283-288
 *
 * Conventions: plain C types only.  `means`, `quats`, ... of sas_scene_upload may be host or
 * device pointers (copied with hipMemcpyDefault).  Output pointers of sas_render are DEVICE
 * pointers owned by the caller (e.g. torch tensors' data_ptr()).  viewmat / K / background /
 * group poses are small HOST arrays.  No exceptions cross the ABI: every call returns 0 or a
 * negative sas_status and sas_last_error() describes the failure.  One ctx per (device, stream);
 * calls on one ctx are not re-entrant.  INTEGRATION.md shows the ctypes stub a maintainer adds.
 */
#ifndef SIM_A_SPLAT_AMD_H
#define SIM_A_SPLAT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sas_ctx sas_ctx;

typedef enum {
    SAS_OK = 0,
    SAS_ERR_INVALID = -1,   /* bad argument */
    SAS_ERR_HIP = -2,       /* a HIP runtime call failed */
    SAS_ERR_NO_SCENE = -3,  /* render before sas_scene_upload */
    SAS_ERR_OOM = -4,       /* device allocation failed */
    SAS_ERR_NO_DEVICE = -5  /* no gfx950 device visible */
} sas_status;

/* sas_render flags */
#define SAS_DEPTH_FILL_MAX 1u /* depth = where(alpha > 0, ED, max(ED)): nerfstudio get_outputs (T0) */
#define SAS_ASYNC 2u          /* enqueue only (<= 4 frames in flight).  Frames COMPLETE in submission order, inside later
                                 sas_render* calls (when a slot is needed) or sas_wait(): completion = the host has checked
                                 that the frame fitted its intersection buffer (a frame that did not is rendered again
                                 first).  A frame's outputs may be consumed -- by the host after a synchronisation, or by
                                 work put on `stream` afterwards, which is then ordered behind the frame -- only once it
                                 is complete; sas_frames_completed() tells how many are. */
#define SAS_FAST_EXP 4u       /* v_exp_f32 instead of the contract polynomial: NOT bit-exact with the oracle */
#define SAS_TIMING 8u         /* record per-stage hipEvents (readable with sas_stage_times) */
#define SAS_TIME_TILES 32u    /* HIP events around the tile kernel only (SAS_T_BLEND); frames still pipeline */
#define SAS_FULL_SORT 16u     /* order every tile list completely and keep it (sas_read_tile_lists); same image */

/* sas_stage_times slots (milliseconds of the last completed frame rendered with SAS_TIMING).  SAS_T_SCAN reads ~0:
 * the offsets scan is the tail of the projection kernel (its last workgroup), not a launch of its own.  SAS_T_SCATTER
 * reads ~0 as well on the product path (single-pass binning: the projection emits the intersection keys itself into
 * fixed-stride tile segments); it times k_scatter for SAS_FULL_SORT frames and with SAS_DIRECT=0 in the environment
 * (two-pass binning: count, scan, scatter), which also serves frames whose segments would exceed SAS_DIRECT_BUDGET_MB
 * (default 6144 MB per frame in flight). */
enum { SAS_T_PROJECT = 0, SAS_T_SCAN, SAS_T_SCATTER, SAS_T_SORT /* full path only */,
       SAS_T_BLEND /* k_tile_lazy, or k_blend on the full path */, SAS_T_TAIL /* depth fill */,
       SAS_T_TOTAL, SAS_T_COUNT };

/* sas_frame_stats slots (int64) of the last completed frame.  SAS_S_NISECT counts Gaussian x 16-pixel-tile intersections
 * (gsplat's isect count) whatever the frame's own binning; SAS_S_NKEYS what the frame actually binned (fewer on single-pass
 * frames, whose lists leave out the tiles of a rectangle the Gaussian cannot reach -- no pixel changes; SAS_CULL=0 bins whole
 * rectangles -- and more in the quad layout, which bins in 8-pixel tiles); SAS_S_MAX_TILE_LEN is the longest list of the frame's own tiles;
 * SAS_S_CAPACITY the keys the frame's buffer holds (single-pass binning: tiles x segment; two-pass: the compact buffer). */
enum { SAS_S_NVISIBLE = 0, SAS_S_NISECT, SAS_S_MAX_TILE_LEN, SAS_S_CAPACITY, SAS_S_REGROWS,
       SAS_S_WINDOW_MISSES /* workgroups that binned with per-intersection atomics */,
       SAS_S_FALLBACK_TILES /* tiles the lazy kernel had to order completely */,
       SAS_S_QUAD_LAYOUT /* 1: the frame ran in the quad layout (views of a few hundred tiles): binned in 8-pixel tiles, one
                            workgroup per 8x8 quadrant with one wave per 4x4 block */,
       SAS_S_LAUNCH_VIEWS /* views that shared the frame's launches (1, or the size of its launch group) */,
       SAS_S_NKEYS /* intersection keys written at the frame's own tile size */, SAS_S_COUNT };

/* Create / destroy a rasterizer context on HIP device `device`. */
int sas_create(int device, sas_ctx **out);
int sas_destroy(sas_ctx *ctx);

/*
 * Upload (replace) the scene.  Serves GSplatLoader -> scene registration
 * (sim_a_splat/splat/splat_utils.py:33-45, splat_handler.py:106-141).
 *   means      [n,3]   world positions
 *   quats      [n,4]   wxyz, any norm   } Door A form (gsplat normalises), or both NULL and
 *   scales     [n,3]   exp() applied    }
 *   cov6       [n,6]   xx xy xz yy yz zz  Door B form (viser takes 3x3 covariances)
 *   opacities  [n]     sigmoid() applied
 *   colors     sh_degree >= 0: [n,(sh_degree+1)^2,3] SH coefficients (features_dc ++ features_rest)
 *              sh_degree <  0: [n,3] final RGB in 0..1 (Door B: SH2RGB already applied)
 *   group_id   [n] uint8 splat-group index, or NULL (single static group)
 *   n_groups   number of groups (<= 256); poses start as identity
 * Values are taken as they are (no validation pass over the scene): every float32 is defined input.  A Gaussian whose projection
 * is not finite is culled; NaN / Inf / out-of-range opacities and colours render exactly as the oracle renders them (DESIGN.md 3,
 * "Inputs outside the reference's range"; colours leave the projection clamped to +-FLT_MAX); no input makes a kernel leave its
 * buffers (tests/tools/oracle_fuzz.py poisons scenes on the bounds-checked build).
 */
int sas_scene_upload(sas_ctx *ctx, int64_t n, const float *means, const float *quats, const float *scales,
                     const float *cov6, const float *opacities, const float *colors, int sh_degree,
                     const uint8_t *group_id, int n_groups);

/* Per-group rigid poses, [n_groups,12] row-major (R|t), host pointer.  Serves the per-step
 * `splat_links_handler[i].wxyz/.position = ...` assignments (splat_handler.py:283-288).
 * The poses apply to the frames submitted AFTER the call: every frame carries a snapshot of the poses it was
 * submitted with, so frames in flight (SAS_ASYNC) are neither disturbed nor waited for. */
int sas_set_group_poses(sas_ctx *ctx, int n_groups, const float *Rt);

/*
 * The per-link pose algebra of SplatHandler.draw_handler (splat_handler.py:239-288) inside the library, float64:
 *   R_k = Ri Rm_k Rfk_k^T Ri^T,   t_k = ti - R_k ti + s Ri (tm_k - Rm_k Rfk_k^T tfk_k),   tm_k = p_msg_k + weld,
 * with Rm_k the rotation of the message quaternion; the rotation is rounded through a unit quaternion, as the
 * reference's handles store one (handle.wxyz = ..., :283-288).
 * sas_set_link_constants, once per scene (after sas_scene_upload, which forgets them): the ICP similarity (scale s,
 *   rotation Ri [9], translation ti [3]: splat_handler.py:66-83), per link k the forward kinematics of its visual mesh
 *   at mask time (Rfk [n_links,9], tfk [n_links,3]: :147-200), the weld translation (:229; NULL = 0) and the splat
 *   group each link drives (NULL: group k).
 * sas_set_link_poses, per env step: the first k_links links' message poses (q_msg [k,4] wxyz any norm, p_msg [k,3])
 *   become the poses of their groups (the other groups keep theirs), exactly as sas_set_group_poses would set them;
 *   Rt_out (or NULL) receives all current group poses [n_groups,12].
 * sas_get_group_poses: the current poses.
 */
int sas_set_link_constants(sas_ctx *ctx, int n_links, double scale, const double *Ri, const double *ti, const double *Rfk,
                           const double *tfk, const double *weld, const int *group);
int sas_set_link_poses(sas_ctx *ctx, int k_links, const double *q_msg, const double *p_msg, float *Rt_out);
int sas_get_group_poses(sas_ctx *ctx, int n_groups, float *Rt);
/* Camera riding on a link (SplatHandler.get_attached_frame, splat_handler.py:316-332) with the context's ICP
 * similarity: pose = icp o SE3(q_link, (p_link + local_xyz) * s) -- the local offset is ADDED in world axes, as the
 * reference does.  Outputs: unit quaternion wxyz [4] and position [3] of the camera-to-world pose. */
int sas_link_attached_frame(sas_ctx *ctx, const double *q_link, const double *p_link, const double *local_xyz,
                            double *wxyz_out, double *xyz_out);

/* The same algebra as pure functions (no context, no GPU; float64 in, what the GPU gets out): the CPU tests hold them
 * against the NumPy forms of sim_a_splat_amd/poses.py.
 *   sas_link_group_poses   k link message poses -> Rt_out [k,12] float32 (the expression of sas_set_link_constants)
 *   sas_attached_frame     as sas_link_attached_frame with the similarity given
 *   sas_camera_matrices    n camera-to-world poses (wxyz [n,4] any norm, position [n,3], OpenCV axes) + vertical field of
 *                          view -> viewmats [n,16] and Ks [n,9] float32: V = [R^T | -R^T p], f = (H/2) / tan(fov/2),
 *                          principal point at the image centre (what get_render's camera means) */
int sas_link_group_poses(int k_links, double scale, const double *Ri, const double *ti, const double *Rfk, const double *tfk,
                         const double *weld, const double *q_msg, const double *p_msg, float *Rt_out);
int sas_attached_frame(double scale, const double *Ri, const double *ti, const double *q_link, const double *p_link,
                       const double *local_xyz, double *wxyz_out, double *xyz_out);
int sas_camera_matrices(int n, const double *wxyz, const double *position, double fov, int width, int height,
                        float *viewmats, float *Ks);

/*
 * Render one view.  Serves get_outputs_for_camera (Door A) and get_render (Door B).
 *   viewmat     [16] row-major world->camera, OpenCV axes (+z forward)
 *   K           [9]  row-major intrinsics
 *   background  [3]
 *   rgb   [H,W,3] f32 or NULL     clamp(render + (1-alpha)*background, 0, 1)
 *   alpha [H,W]   f32 or NULL     accumulation
 *   depth [H,W]   f32 or NULL     expected depth (see SAS_DEPTH_FILL_MAX)
 *   rgb8  [H,W,3] u8  or NULL     floor(rgb*255 + 0.5)
 *   stream      hipStream_t (NULL = default stream): the frame's writes to the output buffers are ordered
 *               behind everything enqueued on it before the call (the scene itself is synchronised by
 *               sas_scene_upload / sas_set_group_poses, which return only when their data is in place)
 * Without SAS_ASYNC the call returns after the frame is complete.
 */
int sas_render(sas_ctx *ctx, const float *viewmat, const float *K, int width, int height,
               const float *background, unsigned flags, float *rgb, float *alpha, float *depth,
               uint8_t *rgb8, void *stream);

/*
 * sas_render with the RGB-D consumer fused into the depth pass.  Replaces the unprojection of
 * GaussianSplat.generate_RGBD_point_cloud (ns_utils/nerfstudio_utils.py:424-445):
 *   points [H,W,3]  camera-frame (x, y, z) = ((u - cx) * d / fx, (v - cy) * d / fy, d), d = the depth output
 *                   (after the SAS_DEPTH_FILL_MAX fill when that flag is set); u, v integer pixel indices
 *   mask   [H,W]    uint8: d < *max_depth, or all ones when max_depth is NULL
 * `depth` is required when points or mask is given; points / mask may each be NULL.
 */
int sas_render_rgbd(sas_ctx *ctx, const float viewmat[16], const float K[9], int width, int height,
                    const float background[3], unsigned flags, const float *max_depth, float *rgb, float *alpha,
                    float *depth, float *points, uint8_t *mask, void *stream);

/*
 * Render n_views views of the same size in one call.  Serves the per-camera loops of
 * SplatHandler.render / SplatEnvWrapper.render (splat_handler.py:337-345, splat_env_wrapper.py:147-158).
 *   viewmats [n_views,16], Ks [n_views,9] host arrays; outputs are [n_views,H,W,...] device arrays
 *   (any may be NULL).  Scenes of >= 0.5 M Gaussians: the views go through the frame slots two at a time, a pair
 * sharing one projection pass over the scene, consecutive pairs overlapping on the GPU.  Smaller scenes: launch
 * groups of (by default) two views that share every launch -- one projection, one scatter, one tile kernel with the
 * views interleaved in dispatch order.  The call returns when all views are complete unless SAS_ASYNC is given (then
 * see SAS_ASYNC: views complete in order inside later calls).
 */
int sas_render_batch(sas_ctx *ctx, int n_views, const float *viewmats, const float *Ks, int width, int height,
                     const float *background, unsigned flags, float *rgb, float *alpha, float *depth,
                     uint8_t *rgb8, void *stream);

/*
 * sas_render_batch for frames that are wanted on the HOST: get_render returns np.uint8 arrays
 * (splat_handler.py:339-344, splat_env_wrapper.py:148-157).  rgb8_host [n_views,H,W,3] is HOST memory (pinned --
 * hipHostMalloc, torch pin_memory -- for speed; pageable memory works through the runtime's staging).  When
 * width and height are multiples of 16 and the destination is pinned, the tile kernel stores each finished tile
 * straight into rgb8_host (its rows packed in LDS, 16 B per lane); otherwise the frames are rendered into a
 * staging buffer of the context and copied out on the frames' own streams right behind the tile kernels.  Either
 * way the call returns with the pixels in place and no second round trip (device-to-host copy issued by the
 * caller after the frame) is needed.  Blocking only (SAS_ASYNC is rejected).  Nothing of the caller's on the device is
 * read or written, so the frames are NOT ordered against work pending on `stream` (two event records and two
 * stream waits per step that a 120-microsecond Gym step notices: docs/EXPERIMENTS.md 5.34).
 */
int sas_render_batch_host(sas_ctx *ctx, int n_views, const float *viewmats, const float *Ks, int width, int height,
                          const float *background, unsigned flags, uint8_t *rgb8_host, void *stream);

/*
 * sas_render_batch / sas_render_batch_host with PER-VIEW pose sets: view v is rendered with the group poses
 * Rt[pose_set[v]] ([n_sets, n_groups, 12] row-major (R|t), host).  Serves vectorised Gym rollouts -- E envs, each with
 * its own link poses, C cameras per env (splat_env_wrapper.py:121-159 poses the scene per env step): one call renders
 * all E*C views, nothing drains between envs, and views of different envs may share a launch group.  The context's
 * current poses (sas_set_group_poses) are not changed.
 */
int sas_render_batch_posed(sas_ctx *ctx, int n_views, const float *viewmats, const float *Ks, const int *pose_set,
                           int n_sets, const float *Rt, int width, int height, const float *background, unsigned flags,
                           float *rgb, float *alpha, float *depth, uint8_t *rgb8, void *stream);
int sas_render_batch_host_posed(sas_ctx *ctx, int n_views, const float *viewmats, const float *Ks, const int *pose_set,
                                int n_sets, const float *Rt, int width, int height, const float *background,
                                unsigned flags, uint8_t *rgb8_host, void *stream);

/* sas_render_batch_host from camera POSES: n_views camera-to-world poses (wxyz [n,4], position [n,3], float64, OpenCV
 * axes: what client.get_render(height, width, wxyz, position) takes, splat_env_wrapper.py:148-157) and one vertical
 * field of view; the view matrices and intrinsics are those of sas_camera_matrices. */
int sas_render_cameras_host(sas_ctx *ctx, int n_views, const double *wxyz, const double *position, double fov, int width,
                            int height, const float *background, unsigned flags, uint8_t *rgb8_host, void *stream);

/* Complete every SAS_ASYNC frame in flight: synchronise with each, and where its intersection buffer
 * overflowed grow it and render the frame again. */
int sas_wait(sas_ctx *ctx);

/* Frames submitted / completed (see SAS_ASYNC) since sas_create; either pointer may be NULL.  Frames
 * 0 .. *completed-1 (in submission order) are final and `stream` is ordered behind them. */
int sas_frames_completed(sas_ctx *ctx, int64_t *submitted, int64_t *completed);

const char *sas_last_error(sas_ctx *ctx);
int sas_stage_times(sas_ctx *ctx, float *ms, int n);
/* Mean stage times over the frames completed with SAS_TIMING / SAS_TIME_TILES since the last reset
 * (slots without events read 0); *frames receives the number of frames averaged. */
int sas_stage_time_means(sas_ctx *ctx, float *ms, int n, int64_t *frames, int reset);
int sas_frame_stats(sas_ctx *ctx, int64_t *stats, int n);

/* Parity hooks (HOST output pointers, any may be NULL): per-Gaussian projection results and the
 * per-tile sorted lists of the last completed frame, in the layout of gsplat's intermediate
 * tensors (radii [n,2] i32, means2d [n,2], depths [n], conics [n,3], colors [n,3];
 * tile_offsets [tiles+1] i32, sorted_ids [<=cap] i32).  sorted_ids is complete only for a frame
 * rendered with SAS_FULL_SORT (the default path orders lists lazily, front chunk by front chunk).
 * sas_read_projection: the product path does not keep the rectangles and radii of a frame (nothing on the device reads
 * them); the hook projects the last frame once more to obtain them -- same camera, same pose snapshot -- before it
 * reads back: a test facility, not a per-frame call. */
int sas_read_projection(sas_ctx *ctx, int32_t *radii, float *means2d, float *depths, float *conics,
                        float *colors);
int sas_read_tile_lists(sas_ctx *ctx, int32_t *tile_offsets, int32_t *sorted_ids, int64_t cap);

const char *sas_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SIM_A_SPLAT_AMD_H */
