// sas_api.cpp -- context management and the C ABI declared in include/sim_a_splat_amd.h.
// Compiled by hipcc together with sas_kernels.hip into libsas_hip.so.  Host float arithmetic
// that feeds the kernels (camera constants) follows the arithmetic contract: built with
// -ffp-contract=off, fused only where fmaf() is written.
#include "../../include/sim_a_splat_amd.h"
#include "sas_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <string>
#include <vector>

#pragma clang fp contract(off)

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct RenderArgs {
    float viewmat[16], K[9], bg[3];
    int W = 0, H = 0;
    unsigned flags = 0;
    float *rgb = nullptr, *alpha = nullptr, *depth = nullptr;
    uint8_t *rgb8 = nullptr;
    uint8_t *rgb8_host = nullptr;   // sas_render_batch_host: host copy of rgb8, made on the frame's stream
    float *points = nullptr;   // RGB-D tail (sas_render_rgbd)
    uint8_t *mask = nullptr;
    float max_depth = 0.0f;
    bool use_max_depth = false;
    hipStream_t stream = nullptr;
    bool order_caller = true;  // the frame writes device buffers of the caller: it runs behind what the caller's stream holds, and the
                               // caller's stream is ordered behind it (false: frames delivered to host memory, sas_render_batch_host)
    bool solo = false;         // a blocking call for this one view with nothing else in flight: the caller waits for the frame's chain
    bool valid = false;
};

}  // namespace

// Per-frame scratch.  Every slot owns one, so several frames can be on the GPU at the same time.
struct Scratch {
    DevBuf rec, col, info, tilebuf, keys, ids, counters, wgvis, wgbase, tilemax;
    long long cap = 0;
    long long seg = 0;            // single-pass binning: keys per tile segment (grown when a tile outgrows it)
    bool seg_too_big = false;     // ... segments for this slot's frames would exceed the memory budget: two-pass binning instead
    int seg_tiles = -1;           // the tile count and scene size `seg` / `seg_too_big` were established for: another frame size or scene
    int64_t seg_n = -1;           // sizes the segments afresh (a slot that once met a pathological frame does not stay two-pass for good)
    // what the slot has learned about other (tile count, scene size) pairs it has rendered: a slot that alternates between two
    // camera sizes neither carries one size's segment length over to the other nor guesses (and overflows) anew at every change
    struct SegMemo { int tiles; int64_t n; long long seg; };
    SegMemo seg_memo[4] = {{-1, -1, 0}, {-1, -1, 0}, {-1, -1, 0}, {-1, -1, 0}};
    int seg_memo_next = 0;
    bool counters_zero = false;   // the counter block is known to be all zero (SasFrame invariant)
};

// One in-flight frame.  Each slot has its own internal stream (plus two side streams for the
// concurrent sort classes) and its own scratch: the stages of a frame are each too short on
// parallelism to fill 256 CUs (a few thousand tiles), so consecutive frames overlap on the chip.
// A frame writes its output buffers only after everything the caller had enqueued on `stream` at the
// time of sas_render (its projection and binning, which touch only the scene and the slot's scratch,
// do not wait for the caller); the caller's stream is made to wait for frame i when it is complete.
// A frame is: [group poses: one small upload kernel] projection (+ key emit, scan and tile order in its tail) [-> scatter: two-pass binning only] -> tile kernel
// [-> depth tail] [-> host copy].  Parameters travel in the kernels' argument segments, the counters are left
// zeroed by the tile kernel, the statistics reach the host through pinned words the projection's tail writes:
// no memset, no upload and no read-back copy around a frame.
struct Slot {
    RenderArgs args;
    hipStream_t fs = nullptr;
    SasSortStreams sort_streams{};
    hipEvent_t start = nullptr, done = nullptr;
    hipEvent_t pair_ev = nullptr;   // leader of a view pair: both projections are done
    hipEvent_t ev[SAS_T_COUNT + 1] = {};
    unsigned *stats_host = nullptr;  // pinned, 8 words, written by the projection's tail (SasFrame::stats_host)
    float *poses_host = nullptr;     // pinned [256 * 12]: the group poses this slot's frame was submitted with
    DevBuf poses_dev;                // ... and their device copy, uploaded in front of the projection
    Scratch scr;
    SasCam cam{};
    SasParams params{};
    bool busy = false, timed = false, timed_tiles = false;
    bool quad = false;   // the frame runs in the quad layout: projected, binned and composited in 8-pixel tiles (prepare_frame)
    bool direct = false; // single-pass binning (SasFrame::seg > 0): the projection emits the keys, no scatter launch (prepare_frame)
    bool host_direct = false;   // the tile kernel delivers the uint8 frame to pinned host memory itself
    bool info_kept = false;     // the frame's projection wrote info[] (SasFrame::keep_info): sas_read_projection need not project again
    int group = 1;   // slots of the launch group this slot LEADS (enqueue_group); 0: member of the group led by an earlier slot
};

constexpr int kMaxSlots = 8;

// Constants of the per-link pose algebra (sas_set_link_constants), float64.
struct LinkConsts {
    int n = 0;
    double scale = 1.0, Ri[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, ti[3] = {0, 0, 0}, weld[3] = {0, 0, 0};
    std::vector<double> Rfk, tfk;   // [n,9], [n,3]
    std::vector<int> group;         // [n] splat group driven by link k
};

struct sas_ctx {
    int device = 0;
    std::string err;
    // scene
    DevBuf g0, g1, g2, col, gid8, perm;
    DevBuf host_stage;   // device staging of sas_render_batch_host's uint8 frames
    // answer of the pinned-memory query for the host buffer of the sas_render_batch_host call being served (cleared when
    // the call returns: nothing is remembered across calls)
    const uint8_t *host_query_base = nullptr, *host_query_end = nullptr;
    bool host_query_ok = false;
    std::vector<int> perm_host;
    SasScene scene{};
    bool has_scene = false;
    std::vector<float> group_host;    // [n_groups * 12] current poses: what a frame submitted now is rendered with
    LinkConsts links;
    // frames
    Slot slots[kMaxSlots];
    int n_slots = 4;     // frames that may be enqueued (SAS_SLOTS=1..8; 6 and 8 are slower)
    int head = 0;        // oldest busy slot
    int inflight = 0;
    int last_slot = 0;   // most recently enqueued (parity hooks)
    hipStream_t stream = nullptr;   // caller's stream of the in-flight frames
    bool has_frame = false;
    // sas_render_batch projects two views per pass over the scene when that pass is long enough to pay
    // (measured: +5 % frames/s at 1 M Gaussians, +16 % at 5 M, -5 % at 0.3 M).  SAS_PAIR=0/1 forces it.
    int pair_views = -1;            // -1: by scene size
    // quad layout (the frame binned in 8-pixel tiles, one workgroup per 8x8 quadrant): -1 = for views of at most
    // quad_max_tiles 16-pixel tiles, 0 = never, 1 = always (SAS_QUAD, SAS_QUAD_TILES)
    // single-pass binning (fixed-stride tile segments, the projection emits the keys): -1 = whenever the segments fit
    // direct_budget bytes per CONTEXT (all its frame slots together: a slot's keys + ids may take direct_budget / n_slots), 0 =
    // never (SAS_DIRECT=0: the two-pass path of rounds 1-3).  24 GB: config 5's 6.4 GB per slot; of 288 GB of HBM
    int direct_mode = -1;
    long long direct_budget = 24ll << 30;
    // exact tile culling on single-pass frames (sas_kernels.hip: tile_reached); SAS_CULL=0 bins whole rectangles as T3 does
    int cull_mode = 1;
    int seg_guess_factor = 16;    // first guess of a tile segment = this x the mean list of a frame with five intersections per Gaussian (SAS_SEG_FACTOR)
    int quad_mode = -1;
    int quad_max_tiles = 640;        // views of frames that share the chip (SAS_ASYNC, batches): beyond, the layout's 4 x workgroups lose
    int quad_max_tiles_solo = 960;   // one blocking view alone on the GPU: its heaviest tile's chain is the frame (tools/quad_threshold.py)
    // sas_render_batch renders the views of a SMALL scene (< kPairMinGaussians: launch-bound frames, the Gym
    // cameras) in groups that share one set of launches (grid.y = view).  SAS_GROUP=1 disables, 2..4 sets the size.
    int group_views = -1;           // -1: half of the slots (two groups can be in flight)
    static constexpr int64_t kPairMinGaussians = 500000;
    uint64_t scene_version = 0;
    int64_t frames_submitted = 0, frames_completed = 0;   // sas_frames_completed
    int64_t stats[SAS_S_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int64_t regrows = 0;
    float stage_ms[SAS_T_COUNT] = {0, 0, 0, 0, 0, 0, 0};
    double stage_sum[SAS_T_COUNT] = {0, 0, 0, 0, 0, 0, 0};
    int64_t stage_frames = 0;
};


namespace {

int fail(sas_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, e_ == hipErrorOutOfMemory ? SAS_ERR_OOM : SAS_ERR_HIP, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                                    \
    } while (0)

int ensure(sas_ctx *c, DevBuf &b, size_t bytes)
{
    if (bytes <= b.bytes && b.p) return SAS_OK;
    if (b.p) {
        HIP_TRY(c, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    if (bytes == 0) bytes = 16;
    HIP_TRY(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return SAS_OK;
}

void release(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

// 3-D Hilbert index (Skilling's transpose form), `bits` per axis.  Storage order of the scene:
// consecutive Gaussians are spatial neighbours, so a workgroup's tile window is compact.
uint32_t hilbert3(uint32_t x, uint32_t y, uint32_t z, int bits)
{
    uint32_t X[3] = {x, y, z};
    const uint32_t M = 1u << (bits - 1);
    for (uint32_t Q = M; Q > 1; Q >>= 1) {
        const uint32_t P = Q - 1;
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    for (int i = 0; i < 3; ++i) X[i] ^= t;
    uint32_t code = 0;
    for (int b = bits - 1; b >= 0; --b)
        for (int i = 0; i < 3; ++i) code = (code << 1) | ((X[i] >> b) & 1u);
    return code;
}

// perm[slot] = caller index, ordered by (group, Hilbert index of the mean, caller index).
void storage_order(int64_t n, const float *means, const uint8_t *gid, std::vector<int> &perm)
{
    perm.resize((size_t)n);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            const float v = means[3 * i + k];
            if (std::isfinite(v)) { lo[k] = std::min(lo[k], v); hi[k] = std::max(hi[k], v); }
        }
    float inv[3];
    for (int k = 0; k < 3; ++k) inv[k] = (hi[k] > lo[k]) ? 1023.0f / (hi[k] - lo[k]) : 0.0f;
    std::vector<uint64_t> key((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        uint32_t q[3];
        for (int k = 0; k < 3; ++k) {
            const float v = means[3 * i + k];
            float u = std::isfinite(v) ? (v - lo[k]) * inv[k] : 0.0f;
            u = std::min(std::max(u, 0.0f), 1023.0f);
            q[k] = (uint32_t)u;
        }
        const uint64_t g = gid ? gid[i] : 0;
        key[(size_t)i] = (g << 32) | hilbert3(q[0], q[1], q[2], 10);
    }
    for (int64_t i = 0; i < n; ++i) perm[(size_t)i] = (int)i;
    std::sort(perm.begin(), perm.end(), [&](int a, int b) { return key[a] != key[b] ? key[a] < key[b] : a < b; });
}

// Camera constants in the oracle's operation order (oracle/sas_oracle.c cam_from, project_one).
void make_cam(const float *V, const float *K, int W, int H, int tile_px, SasCam &c)
{
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) c.R[3 * i + j] = V[4 * i + j];
        c.t[i] = V[4 * i + 3];
    }
    for (int i = 0; i < 3; ++i) {
        float d = fmaf(c.R[6 + i], c.t[2], fmaf(c.R[3 + i], c.t[1], c.R[0 + i] * c.t[0]));
        c.campos[i] = -d;
    }
    c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
    c.W = W; c.H = H;
    c.tile_px = tile_px;
    c.tw = (W + tile_px - 1) / tile_px;
    c.th = (H + tile_px - 1) / tile_px;
    c.Wf = (float)W; c.Hf = (float)H;
    const float tan_fovx = (0.5f * c.Wf) / c.fx;
    const float tan_fovy = (0.5f * c.Hf) / c.fy;
    c.lim_x_pos = fmaf(0.3f, tan_fovx, (c.Wf - c.cx) / c.fx);
    c.lim_x_neg = fmaf(0.3f, tan_fovx, c.cx / c.fx);
    c.lim_y_pos = fmaf(0.3f, tan_fovy, (c.Hf - c.cy) / c.fy);
    c.lim_y_neg = fmaf(0.3f, tan_fovy, c.cy / c.fy);
}

// (a multiple of four ints and four to spare: the projection's tail reads both arrays with 16-byte loads)
static size_t f_wg_stride(const sas_ctx *c) { return (((size_t)((c->scene.n + 255) / 256) + 3) & ~(size_t)3) + 4; }

SasFrame frame_of(sas_ctx *c, Slot &sl, int tiles, bool keep_info = false)
{
    Scratch &q = sl.scr;
    SasFrame f{};
    f.rec = (float4 *)q.rec.p;
    f.col = (float4 *)q.col.p;
    f.info = (uint4 *)q.info.p;
    // counters block: [tickets][8 device counters, 16 class cursors][tile_count][tile_big]   (left zeroed by every frame)
    f.tickets = (unsigned *)q.counters.p;
    f.stats = (unsigned *)q.counters.p + SAS_TICKET_INTS;
    f.class_cursor = (int *)q.counters.p + SAS_TICKET_INTS + 8;
    f.stats_host = sl.stats_host;
    f.tile_count = (int *)q.counters.p + SAS_TICKET_INTS + 32;
    f.tile_big = f.tile_count + sas_count_stride(tiles);
    f.wg_base = (int *)q.wgbase.p;
    const size_t ts = sas_tile_stride(tiles);
    f.tile_offset = (int *)q.tilebuf.p;
    f.tile_cursor = (int *)q.tilebuf.p + ts;
    f.tile_order = (int *)q.tilebuf.p + 2 * ts;
    f.sort_class = (int *)q.tilebuf.p + 3 * ts;
    f.keys = (unsigned long long *)q.keys.p;
    f.sorted_ids = (int *)q.ids.p;
    f.seg = sl.direct ? (int)q.seg : 0;
    f.cap = sl.direct ? (long long)tiles * q.seg : q.cap;
    f.wg_vis = (int *)q.wgvis.p;
    f.cull = (sl.direct && c->cull_mode != 0) ? 1 : 0;
#ifdef SAS_TUNE_STATS
    keep_info = true;   // (the statistics build reads the radii in the tile kernel)
#endif
    f.keep_info = (keep_info || !sl.direct) ? 1 : 0;   // two-pass frames: k_scatter reads the rectangles
    f.group_fill = sl.args.solo ? 0 : 1;               // (measured: pair bench +1.1 % with it, the blocking frame -1.6 %: profiles/r05_ab_empty_tile_groups.txt)
    sl.info_kept = f.keep_info != 0;
    f.wg_isect16 = (sl.quad || f.cull) ? (int *)q.wgvis.p + f_wg_stride(c) : nullptr;   // the lists are not T3's: T3's count is kept beside them
    f.tile_max = (unsigned *)q.tilemax.p;
    f.group_Rt = c->scene.n_groups > 0 ? (const float *)sl.poses_dev.p : nullptr;
    f.group_host = c->scene.n_groups > 0 ? sl.poses_host : nullptr;
    f.n_wg = (int)std::max<int64_t>(1, (c->scene.n + 255) / 256);
    f.n_tiles = tiles;
    return f;
}

// Roles of a slot in a view pair (sas_render_batch): the LEADER's stream runs one two-view projection (whose tail
// scans both views' counts); the FOLLOWER's stream waits for it and continues with its own binning and tiles.
enum { ROLE_SINGLE = 0, ROLE_LEADER = 1, ROLE_FOLLOWER = 2 };

// quad layout for views of `launch_tiles` tiles each?  By the view's own size: counting the frames in
// flight as well measured worse -- 32 Gym cameras per step in launch groups of two run 30 % faster in the quad
// layout than in the ordinary one even with four groups in flight (tools/vec_env_probe.py); what loses is a
// launch that fills the chip by itself (1 200 tiles of 640x480: docs/EXPERIMENTS.md 5.21).
bool use_quad(const sas_ctx *c, int launch_tiles, unsigned flags, bool solo)
{
    if ((flags & SAS_FULL_SORT) || !sas_tiles_lazy_quad_ok((flags & SAS_FAST_EXP) != 0)) return false;
    return c->quad_mode < 0 ? launch_tiles <= (solo ? c->quad_max_tiles_solo : c->quad_max_tiles) : c->quad_mode != 0;
}

// Can a kernel store to this host address (pinned / registered memory)?  Asked on every call: remembering the
// answer per address would be wrong the day a pinned block is freed and a pageable one takes its place.
bool kernel_can_write_host(sas_ctx *c, const void *p)
{
    // (within ONE sas_render_batch_host call the views' frames lie in one caller buffer: asked once, see host_query)
    if (c->host_query_base && p >= c->host_query_base && p < c->host_query_end) return c->host_query_ok;
    hipPointerAttribute_t at{};
    const bool ok = hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost;
    (void)hipGetLastError();   // a pageable pointer makes the query fail: not an error of ours
    return ok;
}

// Nothing pending on the caller's stream?
static bool stream_idle(hipStream_t s)
{
    const hipError_t e = hipStreamQuery(s);
    if (e != hipSuccess) (void)hipGetLastError();   // hipErrorNotReady is an answer, not an error of ours
    return e == hipSuccess;
}

// Group poses of the n frames (one launch): each slot's snapshot goes to its own device block, so frames in flight
// may carry different poses (vectorised envs, a pose update per Gym step) and nothing drains between them.
int enqueue_poses(sas_ctx *c, Slot *const *sl, int n, hipStream_t st, bool multi)
{
    if (c->scene.n_groups <= 0) return SAS_OK;
    if (sas_poses_inline(c->scene.n_groups, n, multi)) return SAS_OK;   // small blocks ride in the projection's arguments
    SasPoseUpload u{};
    u.nv = n;
    for (int k = 0; k < n; ++k) {
        u.dst[k] = (float *)sl[k]->poses_dev.p;
        u.src_host[k] = sl[k]->poses_host;
        u.floats[k] = 12 * c->scene.n_groups;
    }
    sas_launch_pose_upload(st, u);
    return SAS_OK;
}

// Camera constants, scratch sizes and the parameter block of the slot's frame.  `init_st`: the stream the frame's
// projection will run on (the slot's own, or its pair / group leader's): a new counter block is cleared there.
int prepare_frame(sas_ctx *c, Slot &sl, hipStream_t init_st)
{
    const RenderArgs &a = sl.args;
    // The layout is a property of the whole frame: in the quad layout the view is BINNED in 8-pixel tiles (every 8x8
    // quadrant has its own list), so the choice is made here, in front of the projection.
    const int tiles16 = ((a.W + SAS_TILE - 1) / SAS_TILE) * ((a.H + SAS_TILE - 1) / SAS_TILE);
    sl.quad = use_quad(c, tiles16, a.flags, a.solo) && a.W <= 65535 * (SAS_TILE / 2) && a.H <= 65535 * (SAS_TILE / 2);   // (tile coordinates are 16 bits in info)
    make_cam(a.viewmat, a.K, a.W, a.H, sl.quad ? SAS_TILE / 2 : SAS_TILE, sl.cam);
    const SasCam &cam = sl.cam;
    const int tiles = cam.tw * cam.th;
    const int64_t n = c->scene.n;
    Scratch &q = sl.scr;
    int rc;
    if (q.cap == 0) {
        // 8 intersections per Gaussian to start with (trained scenes and the BASELINE configs need 4-5);
        // 12 bytes each: 96 MB per slot at 1 M Gaussians.  A frame that needs more is rendered again.
        long long want = 8 * (long long)n;
        if (want < (1ll << 20)) want = 1ll << 20;
        q.cap = want;
    }
    if ((rc = ensure(c, q.rec, sizeof(float4) * SAS_RS * (size_t)(n > 0 ? n : 1)))) return rc;
    if ((rc = ensure(c, q.col, sizeof(float4) * (size_t)(n > 0 ? n : 1)))) return rc;
    if ((rc = ensure(c, q.info, sizeof(uint4) * (size_t)(n > 0 ? n : 1)))) return rc;
    if ((rc = ensure(c, q.tilebuf, sizeof(int) * (3 * sas_tile_stride(tiles) + 16)))) return rc;
    {
        const size_t cb = sizeof(int) * sas_counter_ints(tiles);
        if (cb > q.counters.bytes || !q.counters.p) q.counters_zero = false;
        if ((rc = ensure(c, q.counters, cb))) return rc;
        if (!q.counters_zero) {   // new block, or a frame that failed half way: zero all of it once, on the slot's stream
            HIP_TRY(c, hipMemsetAsync(q.counters.p, 0, q.counters.bytes, init_st));
            q.counters_zero = true;
        }
    }
    if ((rc = ensure(c, q.wgvis, sizeof(int) * 2 * f_wg_stride(c)))) return rc;   // visible counts | 16-pixel intersections (quad layout)
    if ((rc = ensure(c, q.wgbase, sizeof(int) * SAS_WIN_BINS * (size_t)((n + 255) / 256 + 1)))) return rc;
    if ((rc = ensure(c, q.tilemax, sizeof(unsigned) * (size_t)tiles))) return rc;
    // Single-pass binning: every tile owns a segment of q.seg keys.  First guess: 16 x the mean list of a frame with five
    // intersections per Gaussian, a power of two (config 3: 16 384 keys = 1.6 GB of keys + ids per slot; its longest list
    // is ~6 k); a frame whose longest list outgrows it is rendered again with larger segments (complete_oldest).
    const long long slot_budget = c->direct_budget / std::max(1, c->n_slots);   // the budget is the context's: its slots share it
    if (q.seg_tiles != tiles || q.seg_n != n) {
        // another frame size or scene: remember what this one had learned, take up what the slot knows about the new one
        // (else 0: guessed below -- a segment length learned on a 300-tile frame says nothing about a 1 200-tile one)
        if (q.seg_tiles >= 0 && q.seg > 0) {
            int k = 0;
            while (k < 4 && !(q.seg_memo[k].tiles == q.seg_tiles && q.seg_memo[k].n == q.seg_n)) ++k;
            if (k == 4) { k = q.seg_memo_next; q.seg_memo_next = (q.seg_memo_next + 1) % 4; }
            q.seg_memo[k] = {q.seg_tiles, q.seg_n, q.seg};
        }
        q.seg = 0;
        for (const auto &m : q.seg_memo)
            if (m.tiles == tiles && m.n == n) q.seg = m.seg;
        q.seg_too_big = false;
        q.seg_tiles = tiles;
        q.seg_n = n;
    }
    sl.direct = c->direct_mode != 0 && !(a.flags & SAS_FULL_SORT) && !q.seg_too_big;
    if (sl.direct) {
        if (q.seg == 0) {
            long long guess = c->seg_guess_factor * ((5 * n) / (tiles > 0 ? tiles : 1) + 1);
            long long s2 = 1024;
            while (s2 < guess) s2 <<= 1;
            q.seg = s2;
        }
        if ((long long)tiles * q.seg * 12 > slot_budget || q.seg > (1ll << 30)) {
            sl.direct = false;          // pathological concentration (or a huge frame): the two-pass path has no such limit
            q.seg_too_big = true;
        }
    }
    size_t n_keys = sl.direct ? (size_t)tiles * (size_t)q.seg : (size_t)q.cap;
    {   // buffers sized for a much larger frame (or for segments the slot has since given up) go back to the allocator:
        // several contexts share a card (vectorised ranks, torch), and ensure() by itself only ever grows
        const size_t want_keys = sizeof(unsigned long long) * std::max(n_keys, (size_t)q.cap);
        if (q.keys.bytes > 4 * want_keys && q.keys.bytes > (256u << 20)) { release(q.keys); release(q.ids); }
    }
    if (sl.direct && (ensure(c, q.keys, sizeof(unsigned long long) * std::max(n_keys, (size_t)q.cap)) ||
                      ensure(c, q.ids, sizeof(int) * std::max(n_keys, (size_t)q.cap)))) {
        // the segments do not fit this GPU's free memory: not an error, the two-pass path needs 12 bytes per intersection only
        // (whatever of the segment-sized pair was allocated is released: the compact lists take a fraction of it)
        (void)hipGetLastError();
        c->err.clear();
        release(q.keys);
        release(q.ids);
        sl.direct = false;
        q.seg_too_big = true;
        n_keys = (size_t)q.cap;
    }
    if ((rc = ensure(c, q.keys, sizeof(unsigned long long) * std::max(n_keys, (size_t)q.cap)))) return rc;
    if ((rc = ensure(c, q.ids, sizeof(int) * std::max(n_keys, (size_t)q.cap)))) return rc;
    if (c->scene.n_groups > 0 && (rc = ensure(c, sl.poses_dev, sizeof(float) * 12 * 256))) return rc;

    SasParams &hp = sl.params;
    hp.cam = cam;
    hp.out.rgb = a.rgb; hp.out.alpha = a.alpha; hp.out.depth = a.depth; hp.out.rgb8 = a.rgb8;
    // frame wanted in pinned host memory and every tile complete: the tile kernel stores its rows there itself
    // (no device staging frame, no copy kernel); SAS_FULL_SORT frames keep the staging path
    sl.host_direct = a.rgb8_host && a.W % SAS_TILE == 0 && a.H % SAS_TILE == 0 && ((size_t)a.rgb8_host & 15) == 0 &&
                     !(a.flags & SAS_FULL_SORT) && kernel_can_write_host(c, a.rgb8_host);
    hp.out.rgb8_host = sl.host_direct ? a.rgb8_host : nullptr;
    if (sl.host_direct) hp.out.rgb8 = nullptr;
    hp.out.bg[0] = a.bg[0]; hp.out.bg[1] = a.bg[1]; hp.out.bg[2] = a.bg[2];
    hp.out.points = a.points; hp.out.mask = a.mask;
    hp.out.max_depth = a.max_depth; hp.out.use_max_depth = a.use_max_depth ? 1 : 0;
    hp.out.n_pixels = (long long)a.W * a.H;
    return SAS_OK;
}

// Enqueue the slot's frame on its internal stream (the slot must be idle on the GPU).  The two frames of a view
// pair (role, partner; both prepared by the caller) share the leader's projection.
int enqueue_frame(sas_ctx *c, Slot &sl, int role = ROLE_SINGLE, Slot *partner = nullptr)
{
    const RenderArgs &a = sl.args;
    int rc;
    if (role == ROLE_SINGLE && (rc = prepare_frame(c, sl, sl.fs))) return rc;
    const bool timing = (a.flags & SAS_TIMING) != 0;
    const bool full = (a.flags & SAS_FULL_SORT) != 0;
    const bool ttiles = (a.flags & SAS_TIME_TILES) != 0 && !timing && !full;
    hipStream_t st = sl.fs;
    const SasCam &cam = sl.cam;
    const int tiles = cam.tw * cam.th;
    const SasFrame f = frame_of(c, sl, tiles);
    const SasParams &P = sl.params;

    // after whatever the caller has enqueued on its stream so far: timed frames as a whole, the others from the
    // tile kernel on (the first thing that writes an output buffer; everything before touches only the scene and
    // the slot's scratch)
    // (a blocking single frame whose caller's stream has nothing pending needs no ordering: one query instead of an event
    // record, a stream wait and the wait packet between the scatter and the tile kernel -- +2.4 % on the blocking
    // config-3 frame; pipelined frames keep the event: no gain measured there)
    const bool order = timing || (a.order_caller && !(a.solo && stream_idle(a.stream)));
    if (order) HIP_TRY(c, hipEventRecord(sl.start, a.stream));
    if (timing) HIP_TRY(c, hipStreamWaitEvent(st, sl.start, 0));
    if (role != ROLE_FOLLOWER) {
        Slot *mem[1] = {&sl};
        if ((rc = enqueue_poses(c, mem, 1, st, false))) return rc;   // a pair shares its poses: the leader's block serves both views
    }
    if (role == ROLE_FOLLOWER) HIP_TRY(c, hipStreamWaitEvent(st, partner->pair_ev, 0));
    if (timing) HIP_TRY(c, hipEventRecord(sl.ev[0], st));
    if (role == ROLE_LEADER) {
        const int ptiles = partner->cam.tw * partner->cam.th;
        sas_launch_project2(st, c->scene, P, f, partner->params, frame_of(c, *partner, ptiles));
        HIP_TRY(c, hipEventRecord(sl.pair_ev, st));
    } else if (role == ROLE_SINGLE) {
        sas_launch_project(st, c->scene, P, f);
    }
    if (timing) {
        HIP_TRY(c, hipEventRecord(sl.ev[1], st));
        HIP_TRY(c, hipEventRecord(sl.ev[2], st));   // SAS_T_SCAN: the scan is the projection's tail
    }
    if (!sl.direct) sas_launch_scatter(st, c->scene, cam.tw, f);   // (single-pass binning: keys and tile order are in place when the projection ends)
    if (timing) HIP_TRY(c, hipEventRecord(sl.ev[3], st));
    if (!timing && order) HIP_TRY(c, hipStreamWaitEvent(st, sl.start, 0));
    if (full) sas_launch_sort(st, c->scene, tiles, f, sl.sort_streams);
    if (timing) HIP_TRY(c, hipEventRecord(sl.ev[4], st));
    const bool fill = a.depth && (a.flags & SAS_DEPTH_FILL_MAX);
    const bool quad = sl.quad;   // (prepare_frame; never for SAS_FULL_SORT frames)
    if (full) sas_launch_blend(st, c->scene, tiles, P, f, (a.flags & SAS_FAST_EXP) != 0, fill);
    else sas_launch_tiles_lazy(st, c->scene, tiles, P, f, (a.flags & SAS_FAST_EXP) != 0, fill, quad,
                               ttiles ? sl.ev[4] : nullptr, ttiles ? sl.ev[5] : nullptr);
    if (timing) HIP_TRY(c, hipEventRecord(sl.ev[5], st));
    const bool pts = a.depth && (a.points || a.mask);
    if (fill || pts) sas_launch_depth_tail(st, tiles, P, f, fill, pts);
    if (timing) HIP_TRY(c, hipEventRecord(sl.ev[6], st));
    if (a.rgb8_host && a.rgb8 && !sl.host_direct) {   // frame wanted on the host and not delivered by the tile kernel: by a copy kernel when the destination is pinned (no copy-engine hop)
        const size_t fb = 3 * (size_t)a.W * (size_t)a.H;
        if (kernel_can_write_host(c, a.rgb8_host)) {
            SasHostCopy h{};
            h.nv = 1;
            h.src[0] = a.rgb8;
            h.dst[0] = a.rgb8_host;
            h.bytes = fb;
            sas_launch_host_copy(st, h);
        } else {
            HIP_TRY(c, hipMemcpyAsync(a.rgb8_host, a.rgb8, fb, hipMemcpyDeviceToHost, st));
        }
    }
    HIP_TRY(c, hipGetLastError());
    sl.timed = timing;
    sl.timed_tiles = ttiles;
    HIP_TRY(c, hipEventRecord(sl.done, st));
    sl.busy = true;
    c->has_frame = true;
    return SAS_OK;
}

// Enqueue n same-sized views (consecutive idle slots sl[0..n-1], args filled) as ONE launch group on the leader's
// stream: one pose upload, one projection, one scatter and one tile launch with grid.y = view.  A Gym step's cameras
// on a small scene are launch-bound; the group's kernels also fill more of the chip.
int enqueue_group(sas_ctx *c, Slot **sl, int n)
{
    int rc;
    Slot &ld = *sl[0];
    hipStream_t st = ld.fs;
    for (int k = 0; k < n; ++k)
        if ((rc = prepare_frame(c, *sl[k], st))) return rc;
    // one launch, one binning scheme: single-pass only when every view of the group has its segments
    bool all_direct = true;
    for (int k = 0; k < n; ++k) all_direct = all_direct && sl[k]->direct;
    for (int k = 0; k < n; ++k) sl[k]->direct = all_direct;   // (the key buffers hold max(tiles * seg, cap) keys either way)
    const RenderArgs &a = ld.args;
    const int tiles = ld.cam.tw * ld.cam.th;
    SasMulti mf{};
    mf.nv = n;
    for (int k = 0; k < n; ++k) {
        mf.f[k] = frame_of(c, *sl[k], tiles);
        mf.P[k] = sl[k]->params;
    }
    if ((rc = enqueue_poses(c, sl, n, st, true))) return rc;
    bool order = false;
    for (int k = 0; k < n; ++k) order = order || sl[k]->args.order_caller;
    if (order) HIP_TRY(c, hipEventRecord(ld.start, a.stream));
    sas_launch_project_multi(st, c->scene, mf);
    if (!all_direct) sas_launch_scatter_multi(st, c->scene, ld.cam.tw, mf);
    if (order) HIP_TRY(c, hipStreamWaitEvent(st, ld.start, 0));   // outputs are first written by the tile kernel
    const bool ttiles = (a.flags & SAS_TIME_TILES) != 0;
    bool any_fill = false;
    for (int k = 0; k < n; ++k) any_fill = any_fill || (sl[k]->args.depth && (a.flags & SAS_DEPTH_FILL_MAX));
    const bool quad = ld.quad;   // by the size of one view (prepare_frame): groups of four 300-tile views still gain (vec_env_probe)
    sas_launch_tiles_lazy_multi(st, c->scene, tiles, mf, (a.flags & SAS_FAST_EXP) != 0, any_fill, quad,
                                ttiles ? ld.ev[4] : nullptr, ttiles ? ld.ev[5] : nullptr);
    for (int k = 0; k < n; ++k) {
        const RenderArgs &ak = sl[k]->args;
        const bool fill = ak.depth && (ak.flags & SAS_DEPTH_FILL_MAX);
        const bool pts = ak.depth && (ak.points || ak.mask);
        if (fill || pts) sas_launch_depth_tail(st, tiles, mf.P[k], mf.f[k], fill, pts);
    }
    {   // frames wanted on the host (sas_render_batch_host): by a kernel when the destination is pinned
        const size_t fb = 3 * (size_t)a.W * (size_t)a.H;
        bool any = false;
        for (int k = 0; k < n; ++k) any = any || (sl[k]->args.rgb8_host && sl[k]->args.rgb8 && !sl[k]->host_direct);
        if (any && kernel_can_write_host(c, a.rgb8_host ? a.rgb8_host : sl[n - 1]->args.rgb8_host)) {
            SasHostCopy h{};
            h.nv = n;
            for (int k = 0; k < n; ++k) {
                h.src[k] = (sl[k]->args.rgb8_host && !sl[k]->host_direct) ? sl[k]->args.rgb8 : nullptr;
                h.dst[k] = sl[k]->args.rgb8_host;
            }
            h.bytes = fb;
            sas_launch_host_copy(st, h);
        } else if (any) {
            for (int k = 0; k < n; ++k) {
                const RenderArgs &ak = sl[k]->args;
                if (ak.rgb8_host && ak.rgb8 && !sl[k]->host_direct) HIP_TRY(c, hipMemcpyAsync(ak.rgb8_host, ak.rgb8, fb, hipMemcpyDeviceToHost, st));
            }
        }
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(ld.done, st));
    for (int k = 0; k < n; ++k) {
        sl[k]->busy = true;
        sl[k]->timed = false;
        sl[k]->timed_tiles = false;
        sl[k]->group = k == 0 ? n : 0;
    }
    ld.timed_tiles = ttiles;
    c->has_frame = true;
    return SAS_OK;
}

// After a failure somewhere inside a frame's enqueue the counters can no longer be assumed zero.
void mark_dirty(sas_ctx *c)
{
    (void)hipDeviceSynchronize();
    for (Slot &sl : c->slots) sl.scr.counters_zero = false;
}

// Verify the oldest in-flight frame; on overflow grow its intersection buffer and render it again.
int complete_oldest(sas_ctx *c)
{
    if (c->inflight <= 0) return SAS_OK;
    Slot &sl = c->slots[c->head];
    const int g = sl.group > 1 ? sl.group : 1;   // a launch group completes as a whole (one stream, one done event)
    Slot *mem[SAS_MAX_GROUP];
    for (int k = 0; k < g; ++k) mem[k] = &c->slots[(c->head + k) % c->n_slots];
    for (int attempt = 0; attempt < 4; ++attempt) {
        HIP_TRY(c, hipEventSynchronize(sl.done));
        bool overflow = false;
        for (int k = 0; k < g; ++k) {
            const volatile unsigned *s = mem[k]->stats_host;   // written by the projection's tail (+ the tile kernel's [6])
            c->stats[SAS_S_NVISIBLE] = s[0];
            c->stats[SAS_S_NISECT] = s[3];   // intersections with the contract's 16-pixel tiles ([1]: keys written, at the frame's own binning)
            c->stats[SAS_S_NKEYS] = s[1];
            c->stats[SAS_S_MAX_TILE_LEN] = s[4];
            c->stats[SAS_S_CAPACITY] = mem[k]->direct ? (long long)mem[k]->cam.tw * mem[k]->cam.th * mem[k]->scr.seg : mem[k]->scr.cap;   // keys the frame's buffer holds
            c->stats[SAS_S_REGROWS] = c->regrows;
            c->stats[SAS_S_WINDOW_MISSES] = s[5];
            c->stats[SAS_S_FALLBACK_TILES] = s[6];
            c->stats[SAS_S_QUAD_LAYOUT] = mem[k]->quad ? 1 : 0;
            c->stats[SAS_S_LAUNCH_VIEWS] = g;
            overflow = overflow || s[2] != 0;
        }
        if (sl.timed) {
            for (int k = 0; k < 6; ++k) (void)hipEventElapsedTime(&c->stage_ms[k], sl.ev[k], sl.ev[k + 1]);
            (void)hipEventElapsedTime(&c->stage_ms[SAS_T_TOTAL], sl.ev[0], sl.ev[6]);
            for (int k = 0; k < SAS_T_COUNT; ++k) c->stage_sum[k] += c->stage_ms[k];
            c->stage_frames++;
        } else if (sl.timed_tiles) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, sl.ev[4], sl.ev[5]) == hipSuccess) {
                c->stage_ms[SAS_T_BLEND] = ms;
                c->stage_sum[SAS_T_BLEND] += ms;
                c->stage_frames++;
            }
        }
        if (!overflow) {
            // Only now -- the host has seen the frame finished AND that it did not overflow its intersection buffer -- may
            // the frame be consumed.  Whatever the caller puts on its stream from here on starts after the frame in real
            // time: the frame's done event has been waited for by the host (above), so a stream wait on it would be a
            // no-op on the GPU and one more runtime call and barrier packet per frame (it was issued until round 3).
            for (int k = 0; k < g; ++k) {
                mem[k]->busy = false;
                mem[k]->group = 1;
            }
            c->head = (c->head + g) % c->n_slots;
            c->inflight -= g;
            c->frames_completed += g;
            return SAS_OK;
        }
        // intersection buffer too small: grow to the measured need (+25 %) and render the frame (group) again,
        // with the poses it was submitted with (the slot's snapshot)
        long long want = 0, want_seg = 0;
        for (int k = 0; k < g; ++k) {
            const long long need = (long long)mem[k]->stats_host[1];
            want = std::max(want, need + need / 4 + 1024);
            if (mem[k]->direct) {   // single-pass binning: the longest list (+25 %), as a power of two
                const long long longest = (long long)mem[k]->stats_host[4];
                long long s2 = mem[k]->scr.seg;
                while (s2 < longest + longest / 4) s2 <<= 1;
                want_seg = std::max(want_seg, s2);
            }
        }
        for (int k = 0; k < g; ++k) {
            if (mem[k]->direct) { if (want_seg > mem[k]->scr.seg) mem[k]->scr.seg = want_seg; }
            else if (want > mem[k]->scr.cap) mem[k]->scr.cap = want;
        }
        for (Slot &o : c->slots) {   // the other slots will need it too (those set up for the same frame size and scene)
            if (o.busy) continue;
            if (want_seg && o.scr.seg && o.scr.seg < want_seg && o.scr.seg_tiles == sl.scr.seg_tiles && o.scr.seg_n == sl.scr.seg_n) o.scr.seg = want_seg;
            if (!want_seg && o.scr.cap && o.scr.cap < want) o.scr.cap = want;
        }
        c->regrows++;
        int rc = g > 1 ? enqueue_group(c, mem, g) : enqueue_frame(c, sl);
        if (rc) { mark_dirty(c); return rc; }
    }
    return fail(c, SAS_ERR_HIP, "intersection buffer kept overflowing");
}

int complete_all(sas_ctx *c)
{
    while (c->inflight > 0) {
        int rc = complete_oldest(c);
        if (rc) return rc;
    }
    return SAS_OK;
}

}  // namespace

extern "C" {

const char *sas_version(void) { return "sim_a_splat_amd 0.1 (gfx950)"; }

int sas_create(int device, sas_ctx **out)
{
    if (!out) return SAS_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SAS_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return SAS_ERR_INVALID;
    sas_ctx *c = new (std::nothrow) sas_ctx();
    if (!c) return SAS_ERR_OOM;
    c->device = device;
    bool ok = hipSetDevice(device) == hipSuccess;
    if (const char *e = getenv("SAS_SLOTS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= kMaxSlots) c->n_slots = v;
    }
    if (const char *e = getenv("SAS_PAIR")) c->pair_views = atoi(e) != 0 ? 1 : 0;
    if (const char *e = getenv("SAS_GROUP")) {
        const int v = atoi(e);
        if (v >= 1 && v <= SAS_MAX_GROUP) c->group_views = v;
    }
    if (const char *e = getenv("SAS_QUAD")) c->quad_mode = atoi(e) != 0 ? 1 : 0;
    if (const char *e = getenv("SAS_DIRECT")) c->direct_mode = atoi(e) != 0 ? -1 : 0;
    if (const char *e = getenv("SAS_CULL")) c->cull_mode = atoi(e) != 0;
    if (const char *e = getenv("SAS_SEG_FACTOR")) c->seg_guess_factor = std::max(1, atoi(e));
    if (const char *e = getenv("SAS_DIRECT_BUDGET_MB")) c->direct_budget = std::max(1ll, atoll(e)) << 20;
    if (const char *e = getenv("SAS_QUAD_TILES")) {
        const int v = atoi(e);
        if (v >= 0) c->quad_max_tiles = c->quad_max_tiles_solo = v;
    }
    for (Slot &sl : c->slots) {
        ok = ok && hipHostMalloc((void **)&sl.stats_host, 8 * sizeof(unsigned)) == hipSuccess;
        ok = ok && hipHostMalloc((void **)&sl.poses_host, sizeof(float) * 12 * 256) == hipSuccess;
        ok = ok && hipStreamCreateWithFlags(&sl.fs, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&sl.start, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&sl.pair_ev, hipEventDisableTiming) == hipSuccess;
        for (auto &e : sl.ev) ok = ok && hipEventCreate(&e) == hipSuccess;
        for (auto &sd : sl.sort_streams.side) ok = ok && hipStreamCreateWithFlags(&sd, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&sl.sort_streams.fork, hipEventDisableTiming) == hipSuccess;
        for (auto &e : sl.sort_streams.join) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (ok) memset(sl.stats_host, 0, 8 * sizeof(unsigned));
    }
    if (!ok) {
        sas_destroy(c);
        return SAS_ERR_HIP;
    }
    *out = c;
    return SAS_OK;
}

int sas_destroy(sas_ctx *c)
{
    if (!c) return SAS_ERR_INVALID;
    (void)hipSetDevice(c->device);
    for (Slot &sl : c->slots) {
        if (sl.fs) (void)hipStreamSynchronize(sl.fs);
        for (auto &sd : sl.sort_streams.side)
            if (sd) { (void)hipStreamSynchronize(sd); (void)hipStreamDestroy(sd); }
        if (sl.sort_streams.fork) (void)hipEventDestroy(sl.sort_streams.fork);
        for (auto &e : sl.sort_streams.join)
            if (e) (void)hipEventDestroy(e);
        if (sl.fs) (void)hipStreamDestroy(sl.fs);
        if (sl.stats_host) (void)hipHostFree(sl.stats_host);
        if (sl.poses_host) (void)hipHostFree(sl.poses_host);
        release(sl.poses_dev);
        if (sl.start) (void)hipEventDestroy(sl.start);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.pair_ev) (void)hipEventDestroy(sl.pair_ev);
        for (auto &e : sl.ev)
            if (e) (void)hipEventDestroy(e);
        for (DevBuf *b : {&sl.scr.rec, &sl.scr.col, &sl.scr.info, &sl.scr.tilebuf, &sl.scr.keys, &sl.scr.ids, &sl.scr.counters,
                          &sl.scr.wgvis, &sl.scr.wgbase, &sl.scr.tilemax})
            release(*b);
    }
    for (DevBuf *b : {&c->g0, &c->g1, &c->g2, &c->col, &c->gid8, &c->perm, &c->host_stage}) release(*b);
    delete c;
    return SAS_OK;
}

const char *sas_last_error(sas_ctx *c) { return c ? c->err.c_str() : "null context"; }

int sas_scene_upload(sas_ctx *c, int64_t n, const float *means, const float *quats, const float *scales,
                     const float *cov6, const float *opacities, const float *colors, int sh_degree,
                     const uint8_t *group_id, int n_groups)
{
    if (!c) return SAS_ERR_INVALID;
    if (n < 0 || n > 0x7fffffffll) return fail(c, SAS_ERR_INVALID, "n=%lld out of range", (long long)n);
    if (sh_degree > 3) return fail(c, SAS_ERR_INVALID, "sh_degree %d > 3", sh_degree);
    if (n > 0 && (!means || !opacities || !colors)) return fail(c, SAS_ERR_INVALID, "means/opacities/colors required");
    const bool quat_mode = quats && scales;
    if (n > 0 && !quat_mode && !cov6) return fail(c, SAS_ERR_INVALID, "need quats+scales or cov6");
    if (n_groups < 0 || n_groups > 256) return fail(c, SAS_ERR_INVALID, "n_groups %d out of [0,256]", n_groups);
    if (group_id && n_groups <= 0) return fail(c, SAS_ERR_INVALID, "group_id given but n_groups == 0");
    HIP_TRY(c, hipSetDevice(c->device));
    {
        int rcw = complete_all(c);
        if (rcw) return rcw;
    }
    c->has_scene = false;

    const int deg = sh_degree < 0 ? -1 : sh_degree;
    const int coeff_floats = deg < 0 ? 3 : 3 * (deg + 1) * (deg + 1);
    const int planes = (coeff_floats + 3) / 4;
    const int64_t n_pad = (n + 63) & ~63ll;
    int rc;
    const size_t np = (size_t)(n_pad > 0 ? n_pad : 64);
    if ((rc = ensure(c, c->g0, sizeof(float4) * np))) return rc;
    if ((rc = ensure(c, c->g1, sizeof(float4) * np))) return rc;
    if ((rc = ensure(c, c->g2, sizeof(float4) * np))) return rc;
    if ((rc = ensure(c, c->col, sizeof(float4) * np * planes))) return rc;
    if ((rc = ensure(c, c->gid8, np))) return rc;

    c->perm_host.clear();
    if ((rc = ensure(c, c->perm, sizeof(int) * np))) return rc;
    if (n > 0) {
        // storage order from host copies of the means / group ids
        std::vector<float> h_means((size_t)3 * n);
        std::vector<uint8_t> h_gid;
        HIP_TRY(c, hipMemcpy(h_means.data(), means, sizeof(float) * 3 * n, hipMemcpyDefault));
        if (group_id) {
            h_gid.resize((size_t)n);
            HIP_TRY(c, hipMemcpy(h_gid.data(), group_id, (size_t)n, hipMemcpyDefault));
            for (int64_t i = 0; i < n; ++i)
                if (h_gid[(size_t)i] >= n_groups) return fail(c, SAS_ERR_INVALID, "group_id[%lld]=%d >= n_groups=%d", (long long)i, (int)h_gid[(size_t)i], n_groups);
        }
        storage_order(n, h_means.data(), group_id ? h_gid.data() : nullptr, c->perm_host);
        HIP_TRY(c, hipMemcpy(c->perm.p, c->perm_host.data(), sizeof(int) * n, hipMemcpyHostToDevice));

        // stage the caller's arrays (host or device) and re-lay them out on the device
        DevBuf s_means, s_q, s_s, s_cov, s_op, s_col, s_gid;
        auto stage = [&](DevBuf &b, const void *src, size_t bytes) -> int {
            int r = ensure(c, b, bytes);
            if (r) return r;
            HIP_TRY(c, hipMemcpy(b.p, src, bytes, hipMemcpyDefault));
            return SAS_OK;
        };
        rc = stage(s_means, means, sizeof(float) * 3 * n);
        if (!rc && quat_mode) rc = stage(s_q, quats, sizeof(float) * 4 * n);
        if (!rc && quat_mode) rc = stage(s_s, scales, sizeof(float) * 3 * n);
        if (!rc && !quat_mode) rc = stage(s_cov, cov6, sizeof(float) * 6 * n);
        if (!rc) rc = stage(s_op, opacities, sizeof(float) * n);
        if (!rc) rc = stage(s_col, colors, sizeof(float) * (size_t)coeff_floats * n);
        if (!rc && group_id) rc = stage(s_gid, group_id, (size_t)n);
        if (!rc) {
            sas_launch_relayout(nullptr, n, n_pad, (const int *)c->perm.p, (const float *)s_means.p,
                                (const float *)s_q.p, (const float *)s_s.p, (const float *)s_cov.p,
                                (const float *)s_op.p, (const float *)s_col.p, coeff_floats, planes,
                                (const uint8_t *)s_gid.p, (float4 *)c->g0.p, (float4 *)c->g1.p, (float4 *)c->g2.p,
                                (float4 *)c->col.p, (uint8_t *)c->gid8.p);
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) rc = fail(c, SAS_ERR_HIP, "relayout: %s", hipGetErrorString(e));
        }
        for (DevBuf *b : {&s_means, &s_q, &s_s, &s_cov, &s_op, &s_col, &s_gid}) release(*b);
        if (rc) return rc;
    }

    c->scene = SasScene{};
    c->scene.g0 = (const float4 *)c->g0.p;
    c->scene.g1 = (const float4 *)c->g1.p;
    c->scene.g2 = (const float4 *)c->g2.p;
    c->scene.col = (const float4 *)c->col.p;
    c->scene.gid8 = (const uint8_t *)c->gid8.p;
    c->scene.perm = (const int *)c->perm.p;
    c->scene.n = n;
    c->scene.n_pad = n_pad;
    c->scene.sh_degree = deg;
    c->scene.cov_mode = quat_mode ? 0 : 1;
    c->scene.n_groups = group_id ? n_groups : 0;
    c->group_host.clear();
    c->links = LinkConsts{};
    if (group_id) {   // poses start as identity
        c->group_host.assign((size_t)12 * n_groups, 0.0f);
        for (int g = 0; g < n_groups; ++g) c->group_host[12 * g + 0] = c->group_host[12 * g + 5] = c->group_host[12 * g + 10] = 1.0f;
    }
    for (Slot &sl : c->slots) sl.scr.cap = 0;  // re-derive the intersection capacity for the new scene
    c->has_frame = false;
    c->scene_version++;
    c->has_scene = true;
    return SAS_OK;
}

int sas_set_group_poses(sas_ctx *c, int n_groups, const float *Rt)
{
    if (!c || !Rt) return SAS_ERR_INVALID;
    if (!c->has_scene) return fail(c, SAS_ERR_NO_SCENE, "no scene uploaded");
    if (n_groups != c->scene.n_groups) return fail(c, SAS_ERR_INVALID, "scene has %d groups, got %d", c->scene.n_groups, n_groups);
    // Every frame carries a snapshot of the poses it was submitted with (its slot's block, uploaded in front of its
    // projection; kept for the case that it has to be rendered again), so frames in flight are not disturbed and
    // nothing is drained here: the new poses apply to the frames submitted from now on.
    c->group_host.assign(Rt, Rt + (size_t)12 * n_groups);
    return SAS_OK;
}

// ---- per-link pose algebra (rows a8 of SURVEY.md 8a; reference: splat_handler.py:239-288) -----------------------
// float64, the same expressions in the same order as sim_a_splat_amd/poses.py (quats_wxyz_to_matrices,
// link_splat_poses, matrices_to_quats_wxyz) and SplatScene._sync (quaternion -> matrix -> float32): the handle of a
// splat group stores a quaternion in the reference (handle.wxyz = ...), so the rotation goes matrix -> quaternion ->
// matrix here as well.  Built with -ffp-contract=off like everything else: no fused multiply-adds.
namespace {

void quat_to_matrix(const double *qin, double *R)
{
    const double n = std::sqrt(qin[0] * qin[0] + qin[1] * qin[1] + qin[2] * qin[2] + qin[3] * qin[3]);
    const double q[4] = {qin[0] / n, qin[1] / n, qin[2] / n, qin[3] / n};
    double qq[4][4];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) qq[a][b] = q[a] * q[b];
    R[0] = 1 - 2 * (qq[2][2] + qq[3][3]); R[1] = 2 * (qq[1][2] - qq[0][3]); R[2] = 2 * (qq[1][3] + qq[0][2]);
    R[3] = 2 * (qq[1][2] + qq[0][3]); R[4] = 1 - 2 * (qq[1][1] + qq[3][3]); R[5] = 2 * (qq[2][3] - qq[0][1]);
    R[6] = 2 * (qq[1][3] - qq[0][2]); R[7] = 2 * (qq[2][3] + qq[0][1]); R[8] = 1 - 2 * (qq[1][1] + qq[2][2]);
}

void matrix_to_quat(const double *R, double *q)
{
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        const double s = std::sqrt(tr + 1.0) * 2;
        q[0] = 0.25 * s; q[1] = (R[7] - R[5]) / s; q[2] = (R[2] - R[6]) / s; q[3] = (R[3] - R[1]) / s;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        const double s = std::sqrt(1.0 + R[4 * i] - R[4 * j] - R[4 * k]) * 2;
        q[0] = (R[3 * k + j] - R[3 * j + k]) / s;
        q[1 + i] = 0.25 * s;
        q[1 + j] = (R[3 * j + i] + R[3 * i + j]) / s;
        q[1 + k] = (R[3 * k + i] + R[3 * i + k]) / s;
    }
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int a = 0; a < 4; ++a) q[a] /= n;
}

// C = A B (3x3, row-major); tb: B transposed.  Sums left to right, as a plain triple loop does.
void mul33(const double *A, const double *B, bool tb, double *C)
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += A[3 * r + k] * (tb ? B[3 * c + k] : B[3 * k + c]);
            C[3 * r + c] = acc;
        }
}

}  // namespace

int sas_set_link_constants(sas_ctx *c, int n_links, double scale, const double *Ri, const double *ti, const double *Rfk,
                           const double *tfk, const double *weld, const int *group)
{
    if (!c) return SAS_ERR_INVALID;
    if (!c->has_scene) return fail(c, SAS_ERR_NO_SCENE, "no scene uploaded");
    if (n_links < 0 || n_links > c->scene.n_groups) return fail(c, SAS_ERR_INVALID, "n_links %d out of [0, %d groups]", n_links, c->scene.n_groups);
    if (n_links > 0 && (!Ri || !ti || !Rfk || !tfk)) return fail(c, SAS_ERR_INVALID, "Ri, ti, Rfk, tfk are required");
    LinkConsts L;
    L.n = n_links;
    L.scale = scale;
    if (n_links > 0) {
        memcpy(L.Ri, Ri, sizeof(L.Ri));
        memcpy(L.ti, ti, sizeof(L.ti));
        if (weld) memcpy(L.weld, weld, sizeof(L.weld));
        L.Rfk.assign(Rfk, Rfk + 9 * (size_t)n_links);
        L.tfk.assign(tfk, tfk + 3 * (size_t)n_links);
        L.group.resize((size_t)n_links);
        for (int k = 0; k < n_links; ++k) {
            const int g = group ? group[k] : k;
            if (g < 0 || g >= c->scene.n_groups) return fail(c, SAS_ERR_INVALID, "group[%d]=%d out of [0,%d)", k, g, c->scene.n_groups);
            L.group[(size_t)k] = g;
        }
    }
    c->links = L;
    return SAS_OK;
}

// ---- pure host functions (no context, no GPU): the float64 pose / camera algebra, testable on any machine ----------
int sas_link_group_poses(int k_links, double scale, const double *Ri, const double *ti, const double *Rfk, const double *tfk,
                         const double *weld, const double *q_msg, const double *p_msg, float *Rt_out)
{
    if (k_links < 0 || (k_links > 0 && (!Ri || !ti || !Rfk || !tfk || !q_msg || !p_msg || !Rt_out))) return SAS_ERR_INVALID;
    const double w0[3] = {0, 0, 0};
    const double *wd = weld ? weld : w0;
    for (int k = 0; k < k_links; ++k) {
        double Rm[9], RmF[9], T1[9], R[9], q[4], Rq[9];
        quat_to_matrix(q_msg + 4 * k, Rm);
        const double tm[3] = {p_msg[3 * k] + wd[0], p_msg[3 * k + 1] + wd[1], p_msg[3 * k + 2] + wd[2]};
        mul33(Rm, Rfk + 9 * (size_t)k, true, RmF);    // Rm Rfk^T
        mul33(Ri, RmF, false, T1);
        mul33(T1, Ri, true, R);                        // R = Ri Rm Rfk^T Ri^T
        // t = ti - R ti + s (tm - RmF tfk) Ri^T
        double u[3], t[3];
        for (int r = 0; r < 3; ++r) {
            double acc = 0.0;
            for (int j = 0; j < 3; ++j) acc += RmF[3 * r + j] * tfk[3 * (size_t)k + j];
            u[r] = scale * (tm[r] - acc);
        }
        for (int r = 0; r < 3; ++r) {
            double rti = 0.0, uri = 0.0;
            for (int j = 0; j < 3; ++j) rti += R[3 * r + j] * ti[j];
            for (int j = 0; j < 3; ++j) uri += u[j] * Ri[3 * r + j];
            t[r] = (ti[r] - rti) + uri;
        }
        matrix_to_quat(R, q);        // what the handle stores ...
        quat_to_matrix(q, Rq);       // ... and what the scene uploads
        float *dst = Rt_out + 12 * (size_t)k;
        for (int r = 0; r < 3; ++r) {
            for (int j = 0; j < 3; ++j) dst[4 * r + j] = (float)Rq[3 * r + j];
            dst[4 * r + 3] = (float)t[r];
        }
    }
    return SAS_OK;
}

int sas_attached_frame(double scale, const double *Ri, const double *ti, const double *q_link, const double *p_link,
                       const double *local_xyz, double *wxyz_out, double *xyz_out)
{
    if (!Ri || !ti || !q_link || !p_link || !local_xyz || !wxyz_out || !xyz_out) return SAS_ERR_INVALID;
    double Rl[9], R[9];
    quat_to_matrix(q_link, Rl);
    mul33(Ri, Rl, false, R);
    const double p[3] = {(p_link[0] + local_xyz[0]) * scale, (p_link[1] + local_xyz[1]) * scale, (p_link[2] + local_xyz[2]) * scale};
    for (int r = 0; r < 3; ++r) {
        double acc = 0.0;
        for (int j = 0; j < 3; ++j) acc += Ri[3 * r + j] * p[j];
        xyz_out[r] = acc + ti[r];
    }
    matrix_to_quat(R, wxyz_out);
    return SAS_OK;
}

int sas_camera_matrices(int n, const double *wxyz, const double *position, double fov, int width, int height, float *viewmats,
                        float *Ks)
{
    if (n < 0 || (n > 0 && (!wxyz || !position || !viewmats || !Ks))) return SAS_ERR_INVALID;
    const double f = 0.5 * height / std::tan(0.5 * fov);      // vertical FOV, square pixels
    for (int c = 0; c < n; ++c) {
        double R[9];
        quat_to_matrix(wxyz + 4 * c, R);                        // camera-to-world, OpenCV axes
        float *V = viewmats + 16 * (size_t)c;
        for (int r = 0; r < 3; ++r) {
            double acc = 0.0;
            for (int j = 0; j < 3; ++j) {
                V[4 * r + j] = (float)R[3 * j + r];             // R^T
                acc += R[3 * j + r] * position[3 * c + j];
            }
            V[4 * r + 3] = (float)(-acc);
        }
        V[12] = V[13] = V[14] = 0.0f;
        V[15] = 1.0f;
        float *K = Ks + 9 * (size_t)c;
        K[0] = (float)f; K[1] = 0.0f; K[2] = (float)(0.5 * width);
        K[3] = 0.0f; K[4] = (float)f; K[5] = (float)(0.5 * height);
        K[6] = 0.0f; K[7] = 0.0f; K[8] = 1.0f;
    }
    return SAS_OK;
}

int sas_set_link_poses(sas_ctx *c, int k_links, const double *q_msg, const double *p_msg, float *Rt_out)
{
    if (!c) return SAS_ERR_INVALID;
    if (!c->has_scene) return fail(c, SAS_ERR_NO_SCENE, "no scene uploaded");
    const LinkConsts &L = c->links;
    if (k_links < 0 || k_links > L.n) return fail(c, SAS_ERR_INVALID, "%d link poses, constants for %d (sas_set_link_constants)", k_links, L.n);
    if (k_links > 0 && (!q_msg || !p_msg)) return fail(c, SAS_ERR_INVALID, "q_msg and p_msg are required");
    float rows[12 * 256];
    if (k_links > 0)
        sas_link_group_poses(k_links, L.scale, L.Ri, L.ti, L.Rfk.data(), L.tfk.data(), L.weld, q_msg, p_msg, rows);
    for (int k = 0; k < k_links; ++k) memcpy(&c->group_host[12 * (size_t)L.group[(size_t)k]], rows + 12 * k, sizeof(float) * 12);
    if (Rt_out && !c->group_host.empty()) memcpy(Rt_out, c->group_host.data(), sizeof(float) * c->group_host.size());
    return SAS_OK;
}

int sas_link_attached_frame(sas_ctx *c, const double *q_link, const double *p_link, const double *local_xyz, double *wxyz_out,
                            double *xyz_out)
{
    if (!c) return SAS_ERR_INVALID;
    if (c->links.n <= 0) return fail(c, SAS_ERR_INVALID, "sas_set_link_constants first (the ICP similarity)");
    const int rc = sas_attached_frame(c->links.scale, c->links.Ri, c->links.ti, q_link, p_link, local_xyz, wxyz_out, xyz_out);
    return rc ? fail(c, rc, "sas_link_attached_frame: null argument") : SAS_OK;
}

int sas_get_group_poses(sas_ctx *c, int n_groups, float *Rt)
{
    if (!c || !Rt) return SAS_ERR_INVALID;
    if (!c->has_scene) return fail(c, SAS_ERR_NO_SCENE, "no scene uploaded");
    if (n_groups != c->scene.n_groups) return fail(c, SAS_ERR_INVALID, "scene has %d groups, got %d", c->scene.n_groups, n_groups);
    if (n_groups > 0) memcpy(Rt, c->group_host.data(), sizeof(float) * 12 * (size_t)n_groups);
    return SAS_OK;
}

struct ViewCall {
    const float *viewmat, *K;
    float *rgb, *alpha, *depth;
    uint8_t *rgb8;
    float *points;
    uint8_t *mask;
    uint8_t *rgb8_host = nullptr;
    const float *poses = nullptr;   // [n_groups,12] group poses of THIS view (a pose set), or nullptr: the context's current poses
};

static int check_view(sas_ctx *c, const ViewCall &v, int width, int height)
{
    if (!c->has_scene) return fail(c, SAS_ERR_NO_SCENE, "sas_render before sas_scene_upload");
    if (!v.viewmat || !v.K) return fail(c, SAS_ERR_INVALID, "viewmat and K are required");
    if (width <= 0 || height <= 0 || width > 65535 * SAS_TILE || height > 65535 * SAS_TILE)
        return fail(c, SAS_ERR_INVALID, "bad image size %dx%d", width, height);
    if (!(v.K[0] > 0.0f) || !(v.K[4] > 0.0f)) return fail(c, SAS_ERR_INVALID, "focal lengths must be positive");
    if ((v.points || v.mask) && !v.depth) return fail(c, SAS_ERR_INVALID, "points / mask need the depth output");
    return SAS_OK;
}

static void fill_args(RenderArgs &a, const ViewCall &v, int width, int height, const float *background, unsigned flags,
                      const float *max_depth, hipStream_t st, bool solo)
{
    a.solo = solo;
    // rgb8 beside rgb8_host is the context's own staging frame (sas_render_batch_host): nothing of the caller's on the device
    a.order_caller = v.rgb || v.alpha || v.depth || v.points || v.mask || (v.rgb8 && !v.rgb8_host);
    memcpy(a.viewmat, v.viewmat, sizeof(a.viewmat));
    memcpy(a.K, v.K, sizeof(a.K));
    for (int k = 0; k < 3; ++k) a.bg[k] = background ? background[k] : 0.0f;
    a.W = width; a.H = height; a.flags = flags;
    a.rgb = v.rgb; a.alpha = v.alpha; a.depth = v.depth; a.rgb8 = v.rgb8;
    a.points = v.points; a.mask = v.mask;
    a.rgb8_host = v.rgb8_host;
    a.use_max_depth = max_depth != nullptr;
    a.max_depth = max_depth ? *max_depth : 0.0f;
    a.stream = st;
    a.valid = true;
}

// the poses the slot's frame is rendered with: the view's own set, else the context's current ones
static void snapshot_poses(sas_ctx *c, Slot &sl, const ViewCall &v)
{
    if (c->scene.n_groups <= 0) return;
    memcpy(sl.poses_host, v.poses ? v.poses : c->group_host.data(), sizeof(float) * 12 * (size_t)c->scene.n_groups);
}

// One view (n == 1), a pair of views that share one projection pass (n == 2), or -- `grouped` -- up to
// SAS_MAX_GROUP views that share every launch (enqueue_group).
static int render_views(sas_ctx *c, const ViewCall *views, int n, int width, int height, const float *background,
                        unsigned flags, const float *max_depth, void *stream, bool grouped = false, bool solo = false)
{
    if (!c) return SAS_ERR_INVALID;
    for (int k = 0; k < n; ++k) {
        const int rc = check_view(c, views[k], width, height);
        if (rc) return rc;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    if (c->inflight > 0 && (st != c->stream || (flags & SAS_TIMING))) {
        int rc = complete_all(c);   // one caller stream at a time; timed frames run alone
        if (rc) return rc;
    }
    while (c->inflight > c->n_slots - n) {
        int rc = complete_oldest(c);
        if (rc) return rc;
    }
    c->stream = st;
    // nothing in flight: the ring restarts at slot 0, so that a sequence of frames always meets the slots in the same order
    // (SAS_RING_RESTART=0: the ring goes on where it stood -- bench passes of 25 steps then start on alternating slot pairs)
    static const bool ring_restart = [] { const char *e = getenv("SAS_RING_RESTART"); return !e || atoi(e) != 0; }();
    if (ring_restart && c->inflight == 0) c->head = 0;
    Slot *sl[SAS_MAX_GROUP] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < n; ++k) {
        sl[k] = &c->slots[(c->head + c->inflight + k) % c->n_slots];
        fill_args(sl[k]->args, views[k], width, height, background, flags, max_depth, st, solo && n == 1 && c->inflight == 0);
        snapshot_poses(c, *sl[k], views[k]);
    }
    if (grouped) {
        const int rc = enqueue_group(c, sl, n);
        if (rc) { mark_dirty(c); return rc; }
        c->inflight += n;
        c->frames_submitted += n;
        c->last_slot = (int)(sl[n - 1] - c->slots);
        if (flags & SAS_ASYNC) return SAS_OK;
        return complete_all(c);
    }
    if (n == 2) {
        for (int k = 0; k < 2; ++k) {
            const int rc = prepare_frame(c, *sl[k], sl[0]->fs);
            if (rc) return rc;
        }
    }
    for (int k = 0; k < n; ++k) {
        const int role = n == 1 ? ROLE_SINGLE : (k == 0 ? ROLE_LEADER : ROLE_FOLLOWER);
        int rc = enqueue_frame(c, *sl[k], role, n == 2 ? sl[1 - k] : nullptr);
        if (rc) { mark_dirty(c); return rc; }
        c->inflight++;
        c->frames_submitted++;
        c->last_slot = (int)(sl[k] - c->slots);
    }
    if (flags & SAS_ASYNC) return SAS_OK;
    return complete_all(c);
}

static int render_impl(sas_ctx *c, const float *viewmat, const float *K, int width, int height, const float *background,
                       unsigned flags, float *rgb, float *alpha, float *depth, uint8_t *rgb8, float *points,
                       uint8_t *mask, const float *max_depth, void *stream)
{
    const ViewCall v = {viewmat, K, rgb, alpha, depth, rgb8, points, mask};
    return render_views(c, &v, 1, width, height, background, flags, max_depth, stream, false, !(flags & SAS_ASYNC));
}

int sas_render(sas_ctx *c, const float *viewmat, const float *K, int width, int height, const float *background,
               unsigned flags, float *rgb, float *alpha, float *depth, uint8_t *rgb8, void *stream)
{
    return render_impl(c, viewmat, K, width, height, background, flags, rgb, alpha, depth, rgb8, nullptr, nullptr,
                       nullptr, stream);
}

int sas_render_rgbd(sas_ctx *c, const float *viewmat, const float *K, int width, int height, const float *background,
                    unsigned flags, const float *max_depth, float *rgb, float *alpha, float *depth, float *points,
                    uint8_t *mask, void *stream)
{
    return render_impl(c, viewmat, K, width, height, background, flags, rgb, alpha, depth, nullptr, points, mask,
                       max_depth, stream);
}

// pose sets of a batch: view v is rendered with rows pose_sets[pose_set[v]] (each set [n_groups,12]); no sets: the
// context's current poses for every view
struct PoseSets {
    const int *pose_set = nullptr;
    int n_sets = 0;
    const float *Rt = nullptr;
};

static int render_batch_impl(sas_ctx *c, int n_views, const float *viewmats, const float *Ks, int width, int height,
                             const float *background, unsigned flags, float *rgb, float *alpha, float *depth, uint8_t *rgb8,
                             uint8_t *rgb8_host, void *stream, const PoseSets &ps = PoseSets())
{
    if (!c) return SAS_ERR_INVALID;
    if (n_views < 0 || (n_views > 0 && (!viewmats || !Ks))) return fail(c, SAS_ERR_INVALID, "bad view batch");
    if (ps.Rt) {
        if (!c->has_scene) return fail(c, SAS_ERR_NO_SCENE, "no scene uploaded");
        if (c->scene.n_groups <= 0) return fail(c, SAS_ERR_INVALID, "pose sets given but the scene has no splat groups");
        if (ps.n_sets <= 0 || !ps.pose_set) return fail(c, SAS_ERR_INVALID, "pose sets need pose_set[n_views] and n_sets > 0");
        for (int v = 0; v < n_views; ++v)
            if (ps.pose_set[v] < 0 || ps.pose_set[v] >= ps.n_sets)
                return fail(c, SAS_ERR_INVALID, "pose_set[%d]=%d out of [0,%d)", v, ps.pose_set[v], ps.n_sets);
    }
    const size_t px = (size_t)width * (size_t)height;
    auto view = [&](int v) {
        ViewCall vc{viewmats + 16 * v, Ks + 9 * v, rgb ? rgb + 3 * px * v : nullptr, alpha ? alpha + px * v : nullptr,
                    depth ? depth + px * v : nullptr, rgb8 ? rgb8 + 3 * px * v : nullptr, nullptr, nullptr};
        vc.rgb8_host = rgb8_host ? rgb8_host + 3 * px * v : nullptr;
        vc.poses = ps.Rt ? ps.Rt + (size_t)12 * c->scene.n_groups * ps.pose_set[v] : nullptr;
        return vc;
    };
    // Views go through the frame slots two at a time: one pass over the scene projects both
    // (timed and full-sort frames keep to one view per pass; so do two views of different pose sets).
    const bool want_pairs = c->pair_views < 0 ? c->scene.n >= sas_ctx::kPairMinGaussians : c->pair_views != 0;
    const bool pair = want_pairs && c->n_slots >= 2 && !(flags & (SAS_TIMING | SAS_FULL_SORT));
    // small scenes (not paired): launch groups, by default half of the slots each so that two can be in flight
    int gsz = c->group_views > 0 ? c->group_views : std::max(2, c->n_slots / 2);
    gsz = std::min(std::min(gsz, SAS_MAX_GROUP), c->n_slots);
    const bool group = !pair && gsz >= 2 && n_views >= 2 && !(flags & (SAS_TIMING | SAS_FULL_SORT));
    for (int v = 0; v < n_views;) {
        if (group && v + 1 < n_views) {
            const int n = std::min(gsz, n_views - v);
            ViewCall vc[SAS_MAX_GROUP];
            for (int k = 0; k < n; ++k) vc[k] = view(v + k);
            int rc = render_views(c, vc, n, width, height, background, flags | SAS_ASYNC, nullptr, stream, true);
            if (rc) return rc;
            v += n;
            continue;
        }
        const ViewCall v0 = view(v);
        const int n = (pair && v + 1 < n_views && view(v + 1).poses == v0.poses) ? 2 : 1;
        const ViewCall vc[2] = {v0, view(n == 2 ? v + 1 : v)};
        int rc = render_views(c, vc, n, width, height, background, flags | SAS_ASYNC, nullptr, stream, false, n_views == 1 && !(flags & SAS_ASYNC));
        if (rc) return rc;
        v += n;
    }
    if (flags & SAS_ASYNC) return SAS_OK;
    return sas_wait(c);
}

int sas_render_batch(sas_ctx *c, int n_views, const float *viewmats, const float *Ks, int width, int height,
                     const float *background, unsigned flags, float *rgb, float *alpha, float *depth, uint8_t *rgb8,
                     void *stream)
{
    return render_batch_impl(c, n_views, viewmats, Ks, width, height, background, flags, rgb, alpha, depth, rgb8, nullptr, stream);
}

int sas_render_batch_posed(sas_ctx *c, int n_views, const float *viewmats, const float *Ks, const int *pose_set, int n_sets,
                           const float *Rt, int width, int height, const float *background, unsigned flags, float *rgb,
                           float *alpha, float *depth, uint8_t *rgb8, void *stream)
{
    PoseSets ps;
    ps.pose_set = pose_set; ps.n_sets = n_sets; ps.Rt = Rt;
    if (!Rt) return fail(c, SAS_ERR_INVALID, "sas_render_batch_posed: Rt is required");
    return render_batch_impl(c, n_views, viewmats, Ks, width, height, background, flags, rgb, alpha, depth, rgb8, nullptr, stream, ps);
}

static int render_batch_host_impl(sas_ctx *c, int n_views, const float *viewmats, const float *Ks, int width, int height,
                                  const float *background, unsigned flags, uint8_t *rgb8_host, void *stream, const PoseSets &ps)
{
    if (!c) return SAS_ERR_INVALID;
    if (!rgb8_host || (flags & SAS_ASYNC)) return fail(c, SAS_ERR_INVALID, "sas_render_batch_host: host buffer required, blocking only");
    if (n_views <= 0) return n_views == 0 ? SAS_OK : fail(c, SAS_ERR_INVALID, "bad view batch");
    if (width <= 0 || height <= 0) return fail(c, SAS_ERR_INVALID, "bad image size");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->inflight > 0) {   // the staging buffer below may still be the target of frames in flight
        const int rc = complete_all(c);
        if (rc) return rc;
    }
    const size_t bytes = 3 * (size_t)width * (size_t)height * (size_t)n_views;
    c->host_query_base = nullptr;
    const bool pinned = kernel_can_write_host(c, rgb8_host);
    // the staging frames are needed only when the tile kernel cannot deliver the frames itself (prepare_frame's rule)
    const bool direct = pinned && width % SAS_TILE == 0 && height % SAS_TILE == 0 && ((size_t)rgb8_host & 15) == 0 &&
                        ((3 * (size_t)width * (size_t)height) & 15) == 0 && !(flags & SAS_FULL_SORT);
    if (!direct || !c->host_stage.p) {
        const int rc = ensure(c, c->host_stage, bytes);
        if (rc) return rc;
    }
    c->host_query_base = rgb8_host;
    c->host_query_end = rgb8_host + bytes;
    c->host_query_ok = pinned;
    const int rc = render_batch_impl(c, n_views, viewmats, Ks, width, height, background, flags, nullptr, nullptr, nullptr,
                                     (uint8_t *)c->host_stage.p, rgb8_host, stream, ps);
    c->host_query_base = nullptr;
    return rc;
}

int sas_render_batch_host(sas_ctx *c, int n_views, const float *viewmats, const float *Ks, int width, int height,
                          const float *background, unsigned flags, uint8_t *rgb8_host, void *stream)
{
    return render_batch_host_impl(c, n_views, viewmats, Ks, width, height, background, flags, rgb8_host, stream, PoseSets());
}

int sas_render_batch_host_posed(sas_ctx *c, int n_views, const float *viewmats, const float *Ks, const int *pose_set, int n_sets,
                                const float *Rt, int width, int height, const float *background, unsigned flags,
                                uint8_t *rgb8_host, void *stream)
{
    PoseSets ps;
    ps.pose_set = pose_set; ps.n_sets = n_sets; ps.Rt = Rt;
    if (!Rt) return fail(c, SAS_ERR_INVALID, "sas_render_batch_host_posed: Rt is required");
    return render_batch_host_impl(c, n_views, viewmats, Ks, width, height, background, flags, rgb8_host, stream, ps);
}

int sas_render_cameras_host(sas_ctx *c, int n_views, const double *wxyz, const double *position, double fov, int width,
                            int height, const float *background, unsigned flags, uint8_t *rgb8_host, void *stream)
{
    if (!c) return SAS_ERR_INVALID;
    if (n_views <= 0) return n_views == 0 ? SAS_OK : fail(c, SAS_ERR_INVALID, "bad view batch");
    if (!wxyz || !position || !(fov > 0.0)) return fail(c, SAS_ERR_INVALID, "camera poses and a positive field of view are required");
    std::vector<float> V((size_t)16 * n_views), K((size_t)9 * n_views);
    sas_camera_matrices(n_views, wxyz, position, fov, width, height, V.data(), K.data());
    return render_batch_host_impl(c, n_views, V.data(), K.data(), width, height, background, flags, rgb8_host, stream, PoseSets());
}

int sas_wait(sas_ctx *c)
{
    if (!c) return SAS_ERR_INVALID;
    if (c->inflight <= 0) return SAS_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    return complete_all(c);
}

int sas_frames_completed(sas_ctx *c, int64_t *submitted, int64_t *completed)
{
    if (!c) return SAS_ERR_INVALID;
    if (submitted) *submitted = c->frames_submitted;
    if (completed) *completed = c->frames_completed;
    return SAS_OK;
}

#ifdef SAS_DEBUG_BOUNDS
extern "C" int sas_debug_bounds_kernels(unsigned long long *out, int reset);
extern "C" int sas_debug_bounds_tiles(unsigned long long *out, int reset);
/* Bounds-checked build only: out[0] = guarded accesses found out of range since the last reset (they were
 * skipped, not executed), out[1..3] = code, index and limit of the first one (0 if none). */
int sas_debug_bounds(unsigned long long *out, int reset)
{
    unsigned long long a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess || sas_debug_bounds_kernels(a, reset) || sas_debug_bounds_tiles(b, reset)) return SAS_ERR_HIP;
    const unsigned long long *first = a[0] ? a : b;
    out[0] = a[0] + b[0];
    out[1] = first[1]; out[2] = first[2]; out[3] = first[3];
    return SAS_OK;
}
#endif

int sas_stage_times(sas_ctx *c, float *ms, int n)
{
    if (!c || !ms) return SAS_ERR_INVALID;
    for (int k = 0; k < n && k < SAS_T_COUNT; ++k) ms[k] = c->stage_ms[k];
    return SAS_OK;
}

int sas_stage_time_means(sas_ctx *c, float *ms, int n, int64_t *frames, int reset)
{
    if (!c || !ms) return SAS_ERR_INVALID;
    for (int k = 0; k < n && k < SAS_T_COUNT; ++k)
        ms[k] = c->stage_frames > 0 ? (float)(c->stage_sum[k] / (double)c->stage_frames) : 0.0f;
    if (frames) *frames = c->stage_frames;
    if (reset) {
        for (double &v : c->stage_sum) v = 0.0;
        c->stage_frames = 0;
    }
    return SAS_OK;
}

int sas_frame_stats(sas_ctx *c, int64_t *stats, int n)
{
    if (!c || !stats) return SAS_ERR_INVALID;
    for (int k = 0; k < n && k < SAS_S_COUNT; ++k) stats[k] = c->stats[k];
    return SAS_OK;
}

int sas_read_projection(sas_ctx *c, int32_t *radii, float *means2d, float *depths, float *conics, float *colors)
{
    if (!c) return SAS_ERR_INVALID;
    if (!c->has_scene || !c->has_frame) return fail(c, SAS_ERR_NO_SCENE, "no frame rendered");
    HIP_TRY(c, hipSetDevice(c->device));
    {
        int rc = complete_all(c);
        if (rc) return rc;
    }
    Slot &ls = c->slots[c->last_slot];
    const int64_t n = c->scene.n;
    if (!ls.info_kept && n > 0) {
        // Single-pass product frames do not write info[] (rectangles, radii: nothing on the device reads them).  The hook
        // projects the slot's frame once more with it -- same camera, same pose snapshot, the slot's own scratch; that
        // launch bins again, so the slot's counters are cleared in front of its next frame.
        const int tiles = ls.cam.tw * ls.cam.th;
        Slot *mem[1] = {&ls};
        int rcp = enqueue_poses(c, mem, 1, ls.fs, false);   // (a pair's follower never uploaded its own copy of the snapshot)
        if (rcp) return rcp;
        sas_launch_project(ls.fs, c->scene, ls.params, frame_of(c, ls, tiles, true));
        HIP_TRY(c, hipStreamSynchronize(ls.fs));
        ls.scr.counters_zero = false;
    }
    const Scratch &q = ls.scr;
    std::vector<float> rec((size_t)8 * n), col((size_t)4 * n);
    std::vector<uint32_t> info((size_t)4 * n);
    if (n > 0) {
        HIP_TRY(c, hipMemcpy(rec.data(), q.rec.p, sizeof(float) * 8 * n, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(col.data(), q.col.p, sizeof(float) * 4 * n, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(info.data(), q.info.p, sizeof(uint32_t) * 4 * n, hipMemcpyDeviceToHost));
    }
    for (int64_t j = 0; j < n; ++j) {
        const int64_t i = c->perm_host[(size_t)j];   // slot j holds the caller's Gaussian i
        const uint32_t rx = info[4 * j + 2], ry = info[4 * j + 3];   // radii: 32 bits each
        const bool vis = rx != 0;
        const float *r = &rec[8 * j], *cl = &col[4 * j];
        if (radii) {
            radii[2 * i] = vis ? (int32_t)rx : 0;
            radii[2 * i + 1] = vis ? (int32_t)ry : 0;
        }
        if (means2d) { means2d[2 * i] = vis ? r[0] : 0.f; means2d[2 * i + 1] = vis ? r[1] : 0.f; }
        if (depths) depths[i] = vis ? r[7] : 0.f;
        if (conics) { conics[3 * i] = vis ? r[2] : 0.f; conics[3 * i + 1] = vis ? r[3] : 0.f; conics[3 * i + 2] = vis ? r[4] : 0.f; }
        if (colors) { colors[3 * i] = vis ? cl[0] : 0.f; colors[3 * i + 1] = vis ? cl[1] : 0.f; colors[3 * i + 2] = vis ? cl[2] : 0.f; }
    }
    return SAS_OK;
}

int sas_read_tile_lists(sas_ctx *c, int32_t *tile_offsets, int32_t *sorted_ids, int64_t cap)
{
    if (!c) return SAS_ERR_INVALID;
    if (!c->has_scene || !c->has_frame) return fail(c, SAS_ERR_NO_SCENE, "no frame rendered");
    HIP_TRY(c, hipSetDevice(c->device));
    {
        int rc = complete_all(c);
        if (rc) return rc;
    }
    const Slot &ls = c->slots[c->last_slot];
    const Scratch &q = ls.scr;
    const int tiles = ls.cam.tw * ls.cam.th;
    if (tile_offsets)
        HIP_TRY(c, hipMemcpy(tile_offsets, q.tilebuf.p, sizeof(int) * (size_t)(tiles + 1), hipMemcpyDeviceToHost));
    if (sorted_ids) {
        int64_t m = c->stats[SAS_S_NISECT];
        if (m > cap) m = cap;
        if (m > q.cap) m = q.cap;
        if (m > 0) HIP_TRY(c, hipMemcpy(sorted_ids, q.ids.p, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < m; ++k) {   // storage slots -> caller indices
            const int j = sorted_ids[k];
            sorted_ids[k] = (j >= 0 && j < c->scene.n) ? c->perm_host[(size_t)j] : -1;
        }
    }
    return SAS_OK;
}

}  // extern "C"
