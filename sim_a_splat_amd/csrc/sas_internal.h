// Shared between sas_kernels.hip (device code + launchers) and sas_api.cpp (context, C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SAS_TILE 16
#define SAS_MAX_GROUP 4   // views per launch group (SasMulti)

// Per-frame camera constants, computed on the host with the same f32 operations the oracle uses.
struct SasCam {
    float R[9], t[3];
    float campos[3];
    float fx, fy, cx, cy;
    float lim_x_pos, lim_x_neg, lim_y_pos, lim_y_neg;
    float Wf, Hf;
    int W, H, tw, th;
};

// Scene in HBM.  At upload the Gaussians are re-ordered along a 3-D Hilbert curve (per splat
// group) so that the 256 Gaussians of one workgroup project to a compact screen window, and
// re-laid out into 16-byte planes so that lane i of a wave reads
// bytes [16 i, 16 i + 16) of every plane: each wave-instruction is one contiguous 1 KiB.
//   g0[n] = (mean.x, mean.y, mean.z, opacity)
//   g1[n] = quat wxyz                     | cov xx xy xz yy
//   g2[n] = (scale.x, scale.y, scale.z, group id bits) | (cov yz, cov zz, 0, group id bits)
//   col[p*n_pad + n] = floats 4p..4p+3 of the Gaussian's flattened [K,3] coefficient block
struct SasScene {
    const float4 *g0, *g1, *g2, *col;
    const float *group_Rt;  // [n_groups,12] or nullptr
    const int *perm;        // [n] slot j holds the caller's Gaussian perm[j] (Hilbert order, by group)
    int64_t n;
    int64_t n_pad;     // plane stride
    int sh_degree;     // -1: col plane 0 holds final rgb
    int cov_mode;      // 1: g1/g2 hold a covariance
    int n_groups;
};

// Per-frame scratch owned by the context.
//   rec[3n..3n+2]  projected record of Gaussian n, 48 B:
//        (mean2d.x, mean2d.y, conic.a, conic.b) (conic.c, opacity, skip_threshold, depth) (r, g, b, -)
//   info[n]        (x0 | x1<<16, y0 | y1<<16, depth bits, rx | ry<<16): tile rectangle, 0 if culled
//   stats          [0] n_visible [1] n_isect [2] overflow [3] max ED bits [4] max tile length
//                  [5] workgroups whose screen window did not fit the LDS histogram (direct atomics)
//                  [6] tiles the lazy kernel had to order completely (one depth bucket > chunk)
struct SasFrame {
    float4 *rec;
    uint4 *info;
    int *tile_count;   // [tiles+1]
    int *tile_offset;  // [tiles+1]
    int *tile_cursor;  // [tiles]
    int *tile_order;   // [tiles] tiles by descending list length (blend launch order)
    int *sort_class;   // [6] starts of the large / mid / small sort class in tile_order, tiles; then {0, tiles}
    unsigned long long *keys;  // [cap]  depth bits << 32 | caller index
    int *sorted_ids;           // [cap]  storage slots, per tile, front to back
    long long cap;
    unsigned *stats;           // [8]
    int *wg_vis;               // [ceil(n/256)] visible Gaussians per projection workgroup
    unsigned *tile_max;        // [4 * tiles] per-tile (quad layout: per-quadrant) max expected depth (bits), written when depth is filled
    int n_wg;
    int n_tiles;               // tw * th
};

struct SasOutputs {
    float *rgb, *alpha, *depth;
    uint8_t *rgb8;
    float bg[3];
    // RGB-D tail (sas_render_rgbd): camera-frame points [H,W,3] and depth mask [H,W], or nullptr
    float *points;
    uint8_t *mask;
    float max_depth;      // mask = depth < max_depth when use_max_depth, else all ones
    int use_max_depth;
    long long n_pixels;   // W * H
};

// Per-frame parameters, resident in device memory (one block per frame slot, uploaded on the frame's
// own stream ahead of its kernels).
struct SasParams {
    SasCam cam;
    SasOutputs out;
};

// Up to SAS_MAX_GROUP same-sized views rendered by ONE set of launches (grid.y = view): the cameras of a Gym
// step.  Passed to the *_multi kernels by value.
struct SasMulti {
    SasFrame f[SAS_MAX_GROUP];
    const SasParams *P[SAS_MAX_GROUP];
    int nv;
};

// Frame prologue / epilogue as ONE small kernel each instead of a chain of runtime blits (each hipMemcpyAsync /
// hipMemsetAsync of a few hundred bytes is its own ~6 us command on the stream): the prologue kernel receives the
// views' parameter blocks and the group poses in its argument segment (pose sets too large for that: read from
// PINNED HOST memory by the kernel itself) and zeroes the counter blocks; the epilogue kernel writes the views' 8 statistics words to pinned host memory.
struct SasFrameIo {
    int nv;
    SasParams *params_dev[SAS_MAX_GROUP];
    const SasParams *params_host[SAS_MAX_GROUP];   // pinned
    unsigned *counters[SAS_MAX_GROUP];              // zeroed: counter_words[k] words (stats + tile counts)
    int counter_words[SAS_MAX_GROUP];
    unsigned *stats_host[SAS_MAX_GROUP];            // pinned, 8 words each (epilogue)
    // frames wanted on the host (sas_render_batch_host, pinned destination): copied by the epilogue kernel itself,
    // host_bytes each (0: none) -- a runtime copy between two kernels costs two switches between the compute
    // queue and a copy engine, longer than the copy
    const uint8_t *host_src[SAS_MAX_GROUP];
    uint8_t *host_dst[SAS_MAX_GROUP];
    size_t host_bytes;
    int want_stats;                                 // epilogue: also write the statistics words
    float *groups_dev;                              // or nullptr
    const float *groups_host;                       // pinned
    int group_floats;
};
void sas_launch_frame_prologue(hipStream_t st, const SasFrameIo &io);
void sas_launch_frame_epilogue(hipStream_t st, const SasFrameIo &io);

// launchers (sas_kernels.hip)
void sas_launch_relayout(hipStream_t st, int64_t n, int64_t n_pad, const int *perm, const float *means, const float *quats,
                         const float *scales, const float *cov6, const float *opac, const float *colors,
                         int coeff_floats, int planes, const uint8_t *gid, float4 *g0, float4 *g1, float4 *g2,
                         float4 *col);
void sas_launch_project(hipStream_t st, const SasScene &s, const SasParams *P, const SasFrame &f);
// two views of the scene in one pass over the Gaussians (same scene, same image grid not required)
void sas_launch_project2(hipStream_t st, const SasScene &s, const SasParams *P0, const SasFrame &f0, const SasParams *P1,
                         const SasFrame &f1);
void sas_launch_scan(hipStream_t st, int tiles, const SasFrame &f);
// one launch for all views of a group (same image size): project (one pass over the scene per view), scan,
// scatter, lazy tile kernel
void sas_launch_project_multi(hipStream_t st, const SasScene &s, const SasMulti &mf);
void sas_launch_scan_multi(hipStream_t st, int tiles, const SasMulti &mf);
void sas_launch_scatter_multi(hipStream_t st, const SasScene &s, int tw, const SasMulti &mf);
void sas_launch_tiles_lazy_multi(hipStream_t st, const SasScene &s, int tiles, const SasMulti &mf, bool fast_exp, bool want_max,
                                 bool quad, hipEvent_t ev_start, hipEvent_t ev_stop);
void sas_launch_scatter(hipStream_t st, const SasScene &s, int tw, const SasFrame &f);
struct SasSortStreams {
    hipStream_t side[2];   // nullptr: run the classes back to back on the frame's stream
    hipEvent_t fork, join[2];
};
void sas_launch_sort(hipStream_t st, const SasScene &s, int tiles, const SasFrame &f, const SasSortStreams &ss);
void sas_launch_blend(hipStream_t st, const SasScene &s, int tiles, const SasParams *P, const SasFrame &f,
                      bool fast_exp, bool want_max);
// ev_start/ev_stop (optional): stamped with the kernel's own begin/end (hipExtLaunchKernelGGL).
// quad: four workgroups per tile, one per 8x8 quadrant (small frames; sas_tiles_lazy_quad_ok says whether the
// build has that layout for the requested exponential); tile_max then holds 4 x tiles entries.
bool sas_tiles_lazy_quad_ok(bool fast_exp);
void sas_launch_tiles_lazy(hipStream_t st, const SasScene &s, int tiles, const SasParams *P, const SasFrame &f,
                           bool fast_exp, bool want_max, bool quad, hipEvent_t ev_start, hipEvent_t ev_stop);
// depth tail: fill depth where nothing was composited (fill) and/or unproject it (points)
void sas_launch_depth_tail(hipStream_t st, int tiles, const SasParams *P, const SasFrame &f, bool fill, bool points);
