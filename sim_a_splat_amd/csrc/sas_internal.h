// Shared between sas_kernels.hip (device code + launchers) and sas_api.cpp (context, C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SAS_TILE 16
#ifndef SAS_TILE_GROUP
#define SAS_TILE_GROUP 4   // tiles per unit of the tile order a single-pass projection's tail writes (A/B builds: 1)
#endif
// record / colour strides in float4: a 32-byte record and a colour array of its own.  (Measured against the colour inside a
// 48-byte record as rounds 1-4 had it, the two roles of the projection each writing part of every line: projection +5 us, tile
// kernel +1.5 us, pair bench -5 %: profiles/r05_ab_record_layout.txt.)
#define SAS_RS 2
#define SAS_CS 1
#define SAS_MAX_GROUP 4   // views per launch group (SasMulti)

// Per-frame camera constants, computed on the host with the same f32 operations the oracle uses.
struct SasCam {
    float R[9], t[3];
    float campos[3];
    float fx, fy, cx, cy;
    float lim_x_pos, lim_x_neg, lim_y_pos, lim_y_neg;
    float Wf, Hf;
    int W, H, tw, th;      // tw x th tiles of tile_px pixels
    int tile_px;           // SAS_TILE; 8 in the quad layout, whose 8x8 quadrants are binned as tiles of their own
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
    const uint8_t *gid8;    // [n_pad] group id per slot once more, one byte each: the projection's colour role needs only that of g2
    const int *perm;        // [n] slot j holds the caller's Gaussian perm[j] (Hilbert order, by group)
    int64_t n;
    int64_t n_pad;     // plane stride
    int sh_degree;     // -1: col plane 0 holds final rgb
    int cov_mode;      // 1: g1/g2 hold a covariance
    int n_groups;      // 0: no splat groups (the poses themselves belong to a frame: SasFrame::group_Rt)
};

// Per-frame scratch owned by the context.
//   rec[2n..2n+1]  projected record of Gaussian n, 32 B, written by the projection's GEOMETRY role:
//        (mean2d.x, mean2d.y, conic.a, conic.b) (conic.c, opacity, skip_threshold, depth)
//   col[n]         (r, g, b, -), 16 B, written by the projection's COLOUR role (round 5: the two roles are different
//                  workgroups of one launch, so the colour has an array of its own -- every store instruction of either
//                  role covers whole cache lines)
//   info[n]        (x0 | x1<<16, y0 | y1<<16, radius x, radius y): tile rectangle and radii (parity hook), 0 if culled
//   stats          device counters: [5] workgroups whose screen window did not fit the LDS histogram (direct atomics),
//                  reset by the projection's tail
//   tickets        [65 * 32] words: 64 sub-tickets (one cache line each) + the master ticket of the projection's
//                  workgroups (who is last?); each is reset by its last taker
//   stats_host     PINNED host words the frame reports through, written by the projection's tail (no copy at the
//                  end of a frame): [0] n_visible [1] keys written (= n_isect at 16-pixel binning) [2] overflow [3] n_isect of the contract's 16-pixel
//                  tiles (quad layout) [4] max tile length [5] window misses;
//                  [6] tiles the lazy kernel had to order completely (one depth bucket > chunk): zeroed by the tail,
//                  counted by the tile kernel with a system-scope atomic (rare)
// INVARIANT: tile_count[] and tile_big[] are all zero between frames -- the tile kernel (k_blend on the full path)
// clears the counts of the tile it has just rendered, so a frame needs no memset in front of its projection.
// SINGLE-PASS BINNING (seg > 0; round 4, the product path): every tile owns a segment of `seg` keys at keys + tile * seg
// (HBM is plentiful: 1.6 GB per slot at config 3), so a key's place is known the moment its tile's count atomic returns
// -- the projection emits the keys itself: there is no scatter kernel, no wg_base, no offset scan.  A tile longer than
// `seg` sets the overflow flag (the projection's tail sees the counts); the frame is then rendered again with larger
// segments, like a frame that outgrew `cap` on the two-pass path below.
// TWO-PASS binning (seg == 0: SAS_FULL_SORT frames -- their lists are compact, the parity hooks read them -- and frames whose
// segments would not fit the memory budget), by RESERVATION:
// a projection workgroup counts its intersections per tile in an LDS window and adds each
// window bin to tile_count with ONE returning atomic; what it gets back -- where its run starts inside the tile's
// segment -- goes to wg_base[workgroup][bin].  k_scatter (same workgroups, same windows) then needs no global atomic
// and no counting pass: position = tile_offset + wg_base + rank in LDS.  Gaussians outside the window scheme (large
// rectangles) count into tile_big with one atomic per intersection and are placed behind the tile's window runs
// through tile_cursor, as before.
struct SasFrame {
    float4 *rec;
    float4 *col;
    uint4 *info;
    int *tile_count;   // [tiles] window intersections per tile (+ zero padding the tail's 16-byte loads may touch)
    int *tile_big;     // [tiles] per-intersection counts of Gaussians outside the window scheme (same padding)
    int *wg_base;      // [n_wg * SAS_WIN_BINS] start of each projection workgroup's run inside its tiles' segments
    int *class_cursor; // [16] next free position in tile_order per list-length class (descending), set by the projection's tail
    int *tile_offset;  // [tiles+1]
    int *tile_cursor;  // [tiles] next free position for a tile's `big` entries: starts at tile_offset + tile_count
    int *tile_order;   // [tiles] tiles by descending list-length class (blend launch order)
    int *sort_class;   // [6] starts of the large / mid / small sort class in tile_order, tiles; then {0, tiles}
    unsigned long long *keys;  // [cap]  depth bits << 32 | storage slot
    int *sorted_ids;           // [cap]  storage slots, per tile, front to back
    long long cap;
    int seg;                   // > 0: single-pass binning, tile t's keys (and ids) live at [t * seg, t * seg + count); cap = tiles * seg
    int cull;                  // (single-pass binning only) 1: tiles of its rectangle a Gaussian cannot reach are left out of the lists
    int group_fill;            // 1 (frames that share the chip with others: SAS_ASYNC, batches): ONE workgroup paints the four tiles of an all-empty tile group;
                               // 0 (a blocking frame alone on the GPU): every tile has its own workgroup -- the kernel's end is shorter, its slot time larger
    int keep_info;             // 1: the geometry role writes info[] (two-pass frames: k_scatter reads it; the parity hook; -DSAS_TUNE_STATS builds).
                               // Single-pass product frames skip it: nothing on the device reads it, 16 B per Gaussian less to write
    unsigned *stats;           // [8] device counters
    unsigned *tickets;         // [65 * 32]
    unsigned *stats_host;      // [8] pinned
    int *wg_vis;               // [ceil(n/256)] visible Gaussians per projection workgroup
    int *wg_isect16;           // quad layout (8-pixel binning) only, else nullptr: [ceil(n/256)] intersections with the 16-pixel
                               // tiles of the contract per projection workgroup (the n_isect the frame reports)
    unsigned *tile_max;        // [tiles] per-tile max expected depth (bits), written when depth is filled
    const float *group_Rt;     // [n_groups,12] poses of the splat groups for THIS view (device), or nullptr
    const float *group_host;   // the same rows on the host (the slot's pinned snapshot): small pose blocks ride in the
                               // projection's argument segment instead of being uploaded (sas_poses_inline)
    int n_wg;
    int n_tiles;               // tw * th
};

struct SasOutputs {
    float *rgb, *alpha, *depth;
    uint8_t *rgb8;
    // uint8 frame wanted in PINNED HOST memory and every tile complete (W, H multiples of 16): the tile kernel packs its
    // tile's rows in LDS and stores them to the host itself (16 bytes per lane; 8 in the quad layout), tile by tile while
    // the kernel runs -- no device staging frame, no copy kernel behind the frame.  nullptr otherwise.
    uint8_t *rgb8_host;
    float bg[3];
    // RGB-D tail (sas_render_rgbd): camera-frame points [H,W,3] and depth mask [H,W], or nullptr
    float *points;
    uint8_t *mask;
    float max_depth;      // mask = depth < max_depth when use_max_depth, else all ones
    int use_max_depth;
    long long n_pixels;   // W * H
};

// Per-frame parameters.  They travel in the ARGUMENT SEGMENT of every kernel of the frame (scalar loads, no
// upload in front of the frame).
struct SasParams {
    SasCam cam;
    SasOutputs out;
};

// Up to SAS_MAX_GROUP same-sized views rendered by ONE set of launches: the cameras of a Gym step.  Passed to the
// *_multi kernels by value.
#ifndef SAS_MULTI_INLINE_ROWS
#define SAS_MULTI_INLINE_ROWS 32   // pose rows (all views together) a launch group carries in its argument segment
#endif
#ifndef SAS_PROJ_INLINE_ROWS
#define SAS_PROJ_INLINE_ROWS 16    // ... a single view / a view pair
#endif
struct SasMulti {
    SasFrame f[SAS_MAX_GROUP];
    SasParams P[SAS_MAX_GROUP];
    int nv;
    int mix_k;                                  // geometry blocks per 8 leading blocks of the projection launch (set by the launcher)
    int pose_inline;                            // 1: view k's poses are pose_rows + pose_off[k] (set by the launcher)
    int pose_off[SAS_MAX_GROUP];
    float pose_rows[12 * SAS_MULTI_INLINE_ROWS];
};
// Will the projection launch carry the poses itself (no upload kernel, no dependent-kernel gap in front of it)?
static inline bool sas_poses_inline(int n_groups, int n_views, bool multi)
{
    return n_groups > 0 && (multi ? n_groups * n_views <= SAS_MULTI_INLINE_ROWS : n_groups <= SAS_PROJ_INLINE_ROWS);
}

// Group poses of the views of a launch: ONE small kernel copies each view's [rows, 12] block into that view's device
// buffer -- from its own argument segment when the rows of all views fit (SAS_POSE_INLINE_ROWS), else from PINNED
// host memory.  Frames of scenes without splat groups have no prologue at all.
#define SAS_POSE_INLINE_ROWS 64
struct SasPoseUpload {
    int nv;
    float *dst[SAS_MAX_GROUP];
    const float *src_host[SAS_MAX_GROUP];   // pinned
    int floats[SAS_MAX_GROUP];              // 12 * rows of view k
};
void sas_launch_pose_upload(hipStream_t st, const SasPoseUpload &u);

// Frames wanted on the host (sas_render_batch_host, pinned destination): copied by a kernel behind the tile kernel,
// host_bytes each -- a runtime copy between two kernels costs two switches between the compute queue and a copy
// engine, longer than the copy.
struct SasHostCopy {
    int nv;
    const uint8_t *src[SAS_MAX_GROUP];
    uint8_t *dst[SAS_MAX_GROUP];
    size_t bytes;
};
void sas_launch_host_copy(hipStream_t st, const SasHostCopy &h);

// launchers (sas_kernels.hip)
void sas_launch_relayout(hipStream_t st, int64_t n, int64_t n_pad, const int *perm, const float *means, const float *quats,
                         const float *scales, const float *cov6, const float *opac, const float *colors,
                         int coeff_floats, int planes, const uint8_t *gid, float4 *g0, float4 *g1, float4 *g2,
                         float4 *col, uint8_t *gid8);
// The projection's LAST workgroup to finish also scans the tile counts (offsets, scatter cursors, tile order,
// statistics to stats_host) and resets the ticket: there is no scan kernel.
void sas_launch_project(hipStream_t st, const SasScene &s, const SasParams &P, const SasFrame &f);
// two views of the scene in one pass over the Gaussians (same scene and group poses, same image grid not required)
void sas_launch_project2(hipStream_t st, const SasScene &s, const SasParams &P0, const SasFrame &f0, const SasParams &P1,
                         const SasFrame &f1);
// one launch for all views of a group (same image size): project (one pass over the scene per view), scatter,
// lazy tile kernel
void sas_launch_project_multi(hipStream_t st, const SasScene &s, const SasMulti &mf);
void sas_launch_scatter_multi(hipStream_t st, const SasScene &s, int tw, const SasMulti &mf);
void sas_launch_tiles_lazy_multi(hipStream_t st, const SasScene &s, int tiles, const SasMulti &mf, bool fast_exp, bool want_max,
                                 bool quad, hipEvent_t ev_start, hipEvent_t ev_stop);
void sas_launch_scatter(hipStream_t st, const SasScene &s, int tw, const SasFrame &f);
struct SasSortStreams {
    hipStream_t side[2];   // nullptr: run the classes back to back on the frame's stream
    hipEvent_t fork, join[2];
};
void sas_launch_sort(hipStream_t st, const SasScene &s, int tiles, const SasFrame &f, const SasSortStreams &ss);
void sas_launch_blend(hipStream_t st, const SasScene &s, int tiles, const SasParams &P, const SasFrame &f,
                      bool fast_exp, bool want_max);
// ev_start/ev_stop (optional): stamped with the kernel's own begin/end (hipExtLaunchKernelGGL).
// quad: the frame was projected and binned in 8-pixel tiles (SasCam::tile_px == 8; `tiles` counts those): one workgroup
// per 8x8 quadrant of a 16-pixel tile (small frames; sas_tiles_lazy_quad_ok says whether the build has that layout for
// the requested exponential).
bool sas_tiles_lazy_quad_ok(bool fast_exp);
void sas_launch_tiles_lazy(hipStream_t st, const SasScene &s, int tiles, const SasParams &P, const SasFrame &f,
                           bool fast_exp, bool want_max, bool quad, hipEvent_t ev_start, hipEvent_t ev_stop);
// depth tail: fill depth where nothing was composited (fill) and/or unproject it (points)
void sas_launch_depth_tail(hipStream_t st, int tiles, const SasParams &P, const SasFrame &f, bool fill, bool points);
// size in ints of a frame's counter block ([8 statistics words][tile counts + zero padding]) and of one of the
// three per-tile int arrays of `tilebuf` (offsets, cursors, order; 16-byte aligned strides)
#define SAS_TICKET_INTS (65 * 32)
#define SAS_WIN_BINS 2048   // bins of a binning workgroup's LDS window (8 KiB)
static inline size_t sas_count_stride(int tiles) { return (((size_t)tiles + 1 + 1023) & ~(size_t)1023) + 1024; }
// [tickets][8 statistics words + 16 class cursors + 8 pad][tile_count][tile_big]
static inline size_t sas_counter_ints(int tiles) { return SAS_TICKET_INTS + 32 + 2 * sas_count_stride(tiles); }
static inline size_t sas_tile_stride(int tiles) { return ((size_t)tiles + 1 + 3) & ~(size_t)3; }
