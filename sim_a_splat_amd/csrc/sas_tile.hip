// sas_tile.hip -- per-tile work: depth ordering (T4/T5) and front-to-back compositing (T6 + T0).
//
// Production path: k_tile_lazy.  Compositing stops once every pixel of a tile is opaque, which in
// practice happens a fraction of the way into the tile's list, so ordering the whole list (what a
// global radix sort does) is mostly wasted work.  One workgroup per 16x16 tile (longest list
// first) buckets the tile's UNSORTED keys by depth (256 buckets over the tile's own depth range),
// pulls the nearest <= 512 entries (kChunk) into LDS, radix-sorts only that chunk, composites it, and
// fetches the next buckets only while some pixel is still alive.
//
// Full path (k_sort_* + k_blend, SAS_FULL_SORT): orders every list completely and writes the
// per-tile lists to memory, exactly the T4/T5 products of the reference (read back by
// sas_read_tile_lists for the parity tests).  A tile with more than kChunk entries inside one depth
// bucket (e.g. thousands of coplanar splats) is ordered in place by the lazy kernel itself.
//
// Both paths composite through the same code (blend_range) and therefore produce identical bits.
#include "sas_device.h"

#include <hip/hip_ext.h>

#pragma clang fp contract(off)

namespace {

constexpr int kSortThreads = 256;   // workgroup size of the lazy tile kernel (lds_bucket_rank_sort)

DEV unsigned hi32(unsigned long long k) { return (unsigned)(k >> 32); }
DEV unsigned *s_queue_u32(unsigned char *raw) { return reinterpret_cast<unsigned *>(raw); }   // scratch words in a free LDS region
DEV unsigned lo32(unsigned long long k) { return (unsigned)k; }

// order of the reference: depth bits, then the CALLER's Gaussian index.  Keys carry the storage
// slot in the low word; perm[slot] is looked up only when two depth words are identical.
DEV bool key_greater(unsigned long long a, unsigned long long b, const int *perm)
{
    const unsigned ha = hi32(a), hb = hi32(b);
    if (ha != hb) return ha > hb;
    return perm[lo32(a)] > perm[lo32(b)];
}

// ================================================================================================
// LDS radix sort building blocks
// ================================================================================================

// Stable LSD radix sort (8-bit digits) of m 64-bit keys in LDS by their HIGH word, for a workgroup
// of W waves.  Only the bytes below the top set bit of `span` (largest high word) get a pass.
// Ranks come from wave ballots, so equal digits keep their order (a returning LDS atomic would
// not).  Between barriers the keys live in registers, which makes one buffer enough.
// Afterwards runs of identical high words are ordered by perm[low word].
// AND / OR of one WAVE-UNIFORM flag per wave over the workgroup's four waves, with ONE barrier.  (hipcc's __syncthreads_and / _or is
// an LDS word, a write, an atomic by one lane per wave, a read -- and THREE barriers; the compositing loop runs one per batch of
// 256 entries.)  Two flag rows used alternately (`phase`, uniform, counted by the caller): a wave that is already at the next
// reduction writes the other row, so no barrier is needed between consecutive reductions.
template <bool AND>
DEV bool wg_reduce_wave_flags(bool wave_flag, unsigned &phase)
{
    __shared__ __attribute__((aligned(16))) unsigned s_flags[8];
    unsigned *row = s_flags + 4u * (phase & 1u);
    ++phase;
    if ((threadIdx.x & 63u) == 0u) row[threadIdx.x >> 6] = wave_flag ? 1u : 0u;
    __syncthreads();
    const uint4 v = *reinterpret_cast<const uint4 *>(row);
    return AND ? ((v.x & v.y & v.z & v.w) != 0u) : ((v.x | v.y | v.z | v.w) != 0u);
}

template <int W, int NB>
DEV void lds_radix_sort(unsigned long long *buf, int m, unsigned span, const int *perm, unsigned *cnt /*[W][256]*/,
                        unsigned *dbase /*[256]*/, unsigned *wsum /*[4]*/)
{
    constexpr int THREADS = W * 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nbu = (((m + W - 1) / W) + 63) >> 6;   // 64-key batches per wave, <= NB
    const int base = wv * (nbu * 64) + lane;
    unsigned long long k[NB];
    for (int byte = 0; byte < 4; ++byte) {
        if ((span >> (8 * byte)) == 0u) break;   // uniform
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int i = base + 64 * b;
            k[b] = (b < nbu && i < m) ? buf[i] : ~0ull;
        }
        for (int d = lane; d < 256; d += 64) cnt[wv * 256 + d] = 0u;
        unsigned rank[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b >= nbu) break;   // uniform
            const bool act = base + 64 * b < m;
            const unsigned d = (hi32(k[b]) >> (8 * byte)) & 255u;
            unsigned long long mm = __ballot(act);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool on = (d >> bit) & 1u;
                const unsigned long long bm = __ballot(on);
                mm &= on ? bm : ~bm;
            }
            const unsigned below = mbcnt64(mm);
            const unsigned total = (unsigned)__popcll(mm);
            const unsigned prev = act ? cnt[wv * 256 + d] : 0u;
            if (act && below == 0u) cnt[wv * 256 + d] = prev + total;
            rank[b] = prev + below;
        }
        __syncthreads();
        unsigned tot = 0u, incl = 0u;
        if (tid < 256) {
            unsigned run = 0u;
#pragma unroll
            for (int w = 0; w < W; ++w) { const unsigned cc = cnt[w * 256 + tid]; cnt[w * 256 + tid] = run; run += cc; }
            tot = run;
            incl = tot;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            if (lane == 63) wsum[wv] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            unsigned off = incl - tot;
            for (int w = 0; w < wv; ++w) off += wsum[w];
            dbase[tid] = off;
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int i = base + 64 * b;
            if (b < nbu && i < m) {
                const unsigned d = (hi32(k[b]) >> (8 * byte)) & 255u;
                if (SAS_IN(dbase[d] + cnt[wv * 256 + d] + rank[b], m, 211)) buf[dbase[d] + cnt[wv * 256 + d] + rank[b]] = k[b];
            }
        }
        __syncthreads();
    }
    // runs of identical depth: order by caller index (the run's first element does it; the high
    // words other threads compare do not change under the permutation)
    for (int i = tid; i < m; i += THREADS) {
        const unsigned hd = hi32(buf[i]);
        const bool lead = (i == 0 || hi32(buf[i - 1]) != hd) && (i + 1 < m) && hi32(buf[i + 1]) == hd;
        if (lead) {
            int j = i + 1;
            while (j < m && hi32(buf[j]) == hd) ++j;
            for (int a = i + 1; a < j; ++a) {
                const unsigned long long v = buf[a];
                const int pv = perm[lo32(v)];
                int q = a - 1;
                while (q >= i && perm[lo32(buf[q])] > pv) { buf[q + 1] = buf[q]; --q; }
                buf[q + 1] = v;
            }
        }
    }
    __syncthreads();
}

// Order a chunk whose keys are already grouped by depth bucket (bucket b occupies [cur[b] - hist[b], cur[b])
// of buf; key = depth word relative to the chunk's base << 32 | storage slot): every key counts the keys of
// its own bucket that precede it in the reference order (depth word, then the caller's index perm[slot]) and
// moves to bucket start + count.  Buckets hold a handful of keys (256 buckets over the tile's depth range), so
// this is a few compares per key and three barriers, against three 4-barrier passes of the radix sort.
template <int NK>
DEV void lds_bucket_rank_sort(unsigned long long *buf, int m, int b0, int shift, const unsigned *hist, const unsigned *cur,
                              const int *perm)
{
    const int tid = threadIdx.x;
    unsigned long long key[NK];
    int st[NK], len[NK], rank[NK];
    int maxlen = 0;
#pragma unroll
    for (int u = 0; u < NK; ++u) {
        const int i = u * kSortThreads + tid;
        key[u] = 0ull;
        st[u] = len[u] = rank[u] = 0;
        if (i < m) {
            key[u] = buf[i];
            const int b = b0 + (int)(hi32(key[u]) >> shift);
            len[u] = (int)hist[b];
            st[u] = (int)cur[b] - len[u];
        }
        maxlen = max(maxlen, len[u]);
    }
    // the thread's keys walk their buckets side by side: their LDS reads are independent
    for (int j = 0; j < maxlen; ++j) {
#pragma unroll
        for (int u = 0; u < NK; ++u) {
            if (j < len[u] && SAS_IN(st[u] + j, m, 210)) {
                const unsigned long long kj = buf[st[u] + j];
                const unsigned hj = hi32(kj), hk = hi32(key[u]);
                bool less = hj < hk;
                if (hj == hk && st[u] + j != u * kSortThreads + tid)
                    less = perm[lo32(kj)] < perm[lo32(key[u])];   // identical depth words: rare
                rank[u] += less ? 1 : 0;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NK; ++u)
        if (len[u] > 0 && SAS_IN(st[u] + rank[u], m, 209)) buf[st[u] + rank[u]] = key[u];
    __syncthreads();
}

// Lists longer than every LDS class: bitonic network in its all-ascending form (the first step
// of each merge mirrors), virtual +inf padding, in place on the global segment.
DEV void sort_global_bitonic(unsigned long long *g, int *out, int n, const int *perm, int tid, int nthreads)
{
    int P = 2;
    while (P < n) P <<= 1;
    for (int k = 2; k <= P; k <<= 1) {
        const int hk = k >> 1;
        for (int p = tid; p < (P >> 1); p += nthreads) {
            const int blk = (p / hk) * k, o = p % hk;
            const int l = blk + o, r = blk + k - 1 - o;
            if (r < n) {
                const unsigned long long a = g[l], b = g[r];
                if (key_greater(a, b, perm)) { g[l] = b; g[r] = a; }
            }
        }
        __syncthreads();
        for (int j = k >> 2; j > 0; j >>= 1) {
            for (int p = tid; p < (P >> 1); p += nthreads) {
                const int l = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const int r = l | j;
                if (r < n) {
                    const unsigned long long a = g[l], b = g[r];
                    if (key_greater(a, b, perm)) { g[l] = b; g[r] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += nthreads) out[i] = (int)lo32(g[i]);
}

// ================================================================================================
// Full path, stage 1: k_sort_* write every tile's complete front-to-back list (storage slots)
// ================================================================================================
// Tiles come from a list (`tl[range[0] .. range[1])`): the three size classes are contiguous
// ranges of the length-ordered tile list (tile_order: k_scatter's front workgroups), and the lazy kernel's fallback tiles are
// a list of their own.  Workgroups stride over the range.

template <int CAP, int THREADS, bool LAST_CLASS>
__global__ __launch_bounds__(THREADS) void k_sort_radix(SasFrame f, const int *perm, const int *tl, const int *range)
{
    constexpr int W = THREADS / 64, NB = CAP / THREADS;
    __shared__ unsigned long long buf[CAP];
    __shared__ unsigned cnt[W * 256];
    __shared__ unsigned s_dbase[256];
    __shared__ unsigned s_wsum[4];
    __shared__ unsigned s_mx, s_mn;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int oi = range[0] + (int)blockIdx.x; oi < range[1]; oi += (int)gridDim.x) {
        const int t = tl[oi];
        const long long beg = f.tile_offset[t];
        long long end = f.tile_offset[t + 1];
        if (end > f.cap) end = f.cap;
        const int n = (int)(end - beg);
        if (n <= 0 || (!LAST_CLASS && n > CAP)) continue;
        unsigned long long *g = f.keys + beg;
        int *out = f.sorted_ids + beg;
        if (n == 1) {
            if (tid == 0) out[0] = (int)lo32(g[0]);
            continue;
        }
        if (LAST_CLASS && n > CAP) {
            sort_global_bitonic(g, out, n, perm, tid, THREADS);
            continue;
        }
        if (tid == 0) { s_mx = 0u; s_mn = ~0u; }
        __syncthreads();
        unsigned mn = ~0u, mx = 0u;
        for (int i = tid; i < n; i += THREADS) {
            const unsigned long long key = g[i];
            buf[i] = key;
            mn = min(mn, hi32(key));
            mx = max(mx, hi32(key));
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { mn = min(mn, (unsigned)__shfl_xor((int)mn, d)); mx = max(mx, (unsigned)__shfl_xor((int)mx, d)); }
        if (lane == 0) { atomicMax(&s_mx, mx); atomicMin(&s_mn, mn); }
        __syncthreads();
        // sort depth - min(depth): same order, fewer significant bytes
        const unsigned dmin = s_mn, span = s_mx - s_mn;
        for (int i = tid; i < n; i += THREADS) buf[i] -= (unsigned long long)dmin << 32;
        __syncthreads();
        lds_radix_sort<W, NB>(buf, n, span, perm, cnt, s_dbase, s_wsum);
        for (int i = tid; i < n; i += THREADS) out[i] = (int)lo32(buf[i]);
        __syncthreads();
    }
}

// Lists shorter than 1024: one wave per tile, no workgroup barrier anywhere (a wave executes its
// LDS operations in order).  Sorts (depth word, position in the unsorted segment): 6 bytes of LDS
// per entry.
__global__ __launch_bounds__(64) void k_sort_wave(SasFrame f, const int *perm, const int *tl, const int *range)
{
    constexpr int CAP = 1024, NB = 16;
    __shared__ unsigned sd[CAP];
    __shared__ unsigned short si[CAP];
    __shared__ __attribute__((aligned(16))) unsigned cnt[256];
    const int lane = threadIdx.x;
    for (int oi = range[0] + (int)blockIdx.x; oi < range[1]; oi += (int)gridDim.x) {
        const int t = tl[oi];
        const long long beg = f.tile_offset[t];
        long long end = f.tile_offset[t + 1];
        if (end > f.cap) end = f.cap;
        const int n = (int)(end - beg);
        if (n <= 0 || n > CAP) continue;
        const unsigned long long *g = f.keys + beg;
        int *out = f.sorted_ids + beg;
        if (n == 1) {
            if (lane == 0) out[0] = (int)lo32(g[0]);
            continue;
        }
        const int nb = (n + 63) >> 6;
        unsigned kd[NB], ki[NB];
        unsigned mn = ~0u, mx = 0u;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int i = 64 * b + lane;
            const bool in = b < nb && i < n;
            kd[b] = in ? hi32(g[i]) : ~0u;
            ki[b] = (unsigned)i;
            if (in) { mn = min(mn, kd[b]); mx = max(mx, kd[b]); }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { mn = min(mn, (unsigned)__shfl_xor((int)mn, d)); mx = max(mx, (unsigned)__shfl_xor((int)mx, d)); }
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (b < nb && 64 * b + lane < n) kd[b] -= mn;
        const unsigned span = mx - mn;
        for (int byte = 0; byte < 4; ++byte) {
            if ((span >> (8 * byte)) == 0u) break;   // uniform
            reinterpret_cast<uint4 *>(cnt)[lane] = make_uint4(0u, 0u, 0u, 0u);
            unsigned rank[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b >= nb) break;   // uniform
                const bool act = 64 * b + lane < n;
                const unsigned d = (kd[b] >> (8 * byte)) & 255u;
                unsigned long long m = __ballot(act);
#pragma unroll
                for (int bit = 0; bit < 8; ++bit) {
                    const bool on = (d >> bit) & 1u;
                    const unsigned long long bm = __ballot(on);
                    m &= on ? bm : ~bm;
                }
                const unsigned below = mbcnt64(m);
                const unsigned total = (unsigned)__popcll(m);
                const unsigned prev = act ? cnt[d] : 0u;
                if (act && below == 0u) cnt[d] = prev + total;
                rank[b] = prev + below;
            }
            const uint4 c4 = reinterpret_cast<uint4 *>(cnt)[lane];
            const unsigned s3 = c4.x + c4.y + c4.z + c4.w;
            unsigned incl = s3;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            const unsigned ex = incl - s3;
            reinterpret_cast<uint4 *>(cnt)[lane] = make_uint4(ex, ex + c4.x, ex + c4.x + c4.y, ex + c4.x + c4.y + c4.z);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b < nb && 64 * b + lane < n) {
                    const unsigned pos = cnt[(kd[b] >> (8 * byte)) & 255u] + rank[b];
                    sd[pos] = kd[b];
                    si[pos] = (unsigned short)ki[b];
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int i = 64 * b + lane;
                if (b < nb && i < n) { kd[b] = sd[i]; ki[b] = si[i]; }
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int i = 64 * b + lane;
            if (b < nb && i < n) { sd[i] = kd[b]; si[i] = (unsigned short)ki[b]; }
        }
        for (int i = lane; i < n; i += 64) {
            const unsigned hd = sd[i];
            const bool lead = (i == 0 || sd[i - 1] != hd) && (i + 1 < n) && sd[i + 1] == hd;
            if (lead) {
                int j = i + 1;
                while (j < n && sd[j] == hd) ++j;
                for (int a = i + 1; a < j; ++a) {
                    const unsigned short va = si[a];
                    const int pa = perm[lo32(g[va])];
                    int q = a - 1;
                    while (q >= i && perm[lo32(g[si[q]])] > pa) { si[q + 1] = si[q]; --q; }
                    si[q + 1] = va;
                }
            }
        }
        unsigned lo[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int i = 64 * b + lane;
            lo[b] = (b < nb && i < n) ? lo32(g[si[i]]) : 0u;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int i = 64 * b + lane;
            if (b < nb && i < n) out[i] = (int)lo[b];
        }
    }
}

// ================================================================================================
// Compositing
// ================================================================================================

// 16-bit mask of the 4x4-pixel blocks of tile (tx,ty) the Gaussian can reach.  Bit = cx + 4 cy.
// sqrt(sigma) is a seminorm N (the conic is positive semi-definite), so for a pixel p of a block
// with centre c:  N(p - m) >= N(c - m) - N(p - c) >= N(c - m) - r, where r bounds N over the
// block's half extent (|vx|, |vy| <= 1.5 between pixel centres).  The block is dropped when
// sigma(c - m) > (sqrt(thr + 0.05) + r)^2: no pixel of it can pass the loop's `sigma <= thr`.
// The 0.05 margin is four orders of magnitude above any rounding difference between this
// estimate and the contract's per-pixel sigma, so dropping a block never changes a pixel.
DEV unsigned block_mask16(int tx, int ty, float mx, float my, float A, float B, float C, float thr)
{
    const float r2 = 1.125f * (A + C) + 2.25f * fabsf(B);
    // hardware square roots (1 ulp, no IEEE fix-up): the limit only feeds this conservative test, whose 0.05
    // margin is five orders of magnitude above their error (the IEEE forms cost 30 instructions per entry)
    const float lim = __builtin_amdgcn_sqrtf(thr + 0.05f) + __builtin_amdgcn_sqrtf(r2);
    const float lim2 = lim * lim;
    float hx[4], bx[4], hy[4], dy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float dx = mx - ((float)(tx * SAS_TILE + 4 * k) + 2.0f);
        hx[k] = (0.5f * A) * dx * dx;
        bx[k] = B * dx;
        dy[k] = my - ((float)(ty * SAS_TILE + 4 * k) + 2.0f);
        hy[k] = (0.5f * C) * dy[k] * dy[k];
    }
    unsigned m = 0;
#pragma unroll
    for (int cy = 0; cy < 4; ++cy)
#pragma unroll
        for (int cx = 0; cx < 4; ++cx) {
            const float sc = fma_(bx[cx], dy[cy], hx[cx] + hy[cy]);
            if (!(sc > lim2)) m |= 1u << (cx + 4 * cy);
        }
    return m;
}

// Pixel of (wave, lane) inside the tile: wave w owns the 8x8 quadrant (w & 1, w >> 1); its four
// 16-lane groups own the quadrant's 4x4 blocks, so that each group can walk its own splat queue.
DEV void pixel_of(int wv, int lane, int &ox, int &oy)
{
    const int g = lane >> 4, q = lane & 15;
    ox = (wv & 1) * 8 + (g & 1) * 4 + (q & 3);
    oy = (wv >> 1) * 8 + (g >> 1) * 4 + (q >> 2);
}

// ---- quad layout (k_tile_lazy<..., QUAD>: frames of a few hundred tiles, binned in 8-pixel tiles) ----
// One workgroup per 8x8 quadrant qd = (qx, qy) of a 16-pixel tile, with the quadrant's own list; wave w of a workgroup owns the quadrant's
// 4x4 block w; a lane is (pixel q = lane >> 2 of the block, entry slot e = lane & 3): the four lanes of a
// DPP quad work on four consecutive queue entries of the SAME pixel.
DEV void pixel_of_quad(int qd, int wv, int lane, int &ox, int &oy)
{
    const int q = lane >> 2;
    ox = ((qd & 1) * 2 + (wv & 1)) * 4 + (q & 3);
    oy = ((qd >> 1) * 2 + (wv >> 1)) * 4 + (q >> 2);
}
// bit w = block w of quadrant qd can be reached (the test of block_mask16)
DEV unsigned block_mask4(int tx, int ty, int qd, float mx, float my, float A, float B, float C, float thr)
{
    const float r2 = 1.125f * (A + C) + 2.25f * fabsf(B);
    const float lim = __builtin_amdgcn_sqrtf(thr + 0.05f) + __builtin_amdgcn_sqrtf(r2);
    const float lim2 = lim * lim;
    float hx[2], bx[2], hy[2], dy[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float dx = mx - ((float)(tx * SAS_TILE + 4 * ((qd & 1) * 2 + k)) + 2.0f);
        hx[k] = (0.5f * A) * dx * dx;
        bx[k] = B * dx;
        dy[k] = my - ((float)(ty * SAS_TILE + 4 * ((qd >> 1) * 2 + k)) + 2.0f);
        hy[k] = (0.5f * C) * dy[k] * dy[k];
    }
    unsigned m = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float sc = fma_(bx[w & 1], dy[w >> 1], hx[w & 1] + hy[w >> 1]);
        if (!(sc > lim2)) m |= 1u << w;
    }
    return m;
}

// Per-pixel compositing state.  x is the pixel centre relative to the tile's centre.  A terminated
// pixel (transmittance test fired, or outside the image) is parked at x = NaN: every later sigma is
// then NaN and fails `sigma <= thr` by itself, so the inner loop carries no "done" flag.
struct PixState {
    float T, r, g, b, d;
    float x;
};
// the pixel's tile-local row (its column is PixState::x, which doubles as the parking flag)
struct PixConst {
    float y;
};
DEV bool pix_dead(const PixState &p) { return p.x != p.x; }
// Contract T6: the tile-local frame (u, v, x, y) has its origin at the CENTRE of the 16-pixel tile (x, y in -7.5 .. 7.5)
constexpr float kTileCentre = 8.0f;
DEV PixState pix_init(bool inside, int ox) { return PixState{1.0f, 0.f, 0.f, 0.f, 0.f, inside ? ((float)ox + 0.5f) - kTileCentre : __builtin_nanf("")}; }
DEV PixConst pix_const(int ox, int oy)
{
    (void)ox;
    return PixConst{((float)oy + 0.5f) - kTileCentre};
}
typedef float f32x3 __attribute__((ext_vector_type(3)));

// Contract exp for an argument the caller has clamped to [-86, rounding noise] (sas_oracle_expf clamps to
// [-86, 86]; -sigma never comes near the upper end).  n = round(x log2 e) is read off the low mantissa bits
// of x log2 e + 1.5 * 2^23 and added straight into the exponent field.
DEV float c_expf_neg(float x, float e5 /* 0.0013400432653725147f, pinned in a register */)
{
    const float magic = 12582912.0f;
    const float tm = fma_(x, 1.4426950408889634f, magic);
    const float n = tm - magic;
    const float fr = fma_(x, 1.4426950408889634f, -n);
    float p = fma_(e5, fr, 0.009676037356257439f);
    p = fma_(p, fr, 0.05550327152013779f);
    p = fma_(p, fr, 0.2402210682630539f);
    p = fma_(p, fr, 0.6931471824645996f);
    p = fma_(p, fr, 1.0000001192092896f);
    return __uint_as_float(__float_as_uint(p) + (__float_as_uint(tm) << 23));
}

// a constant pinned in a VGPR (the compiler would re-materialise it with a v_mov inside the loop:
// an SGPR cannot sit beside the literal of v_fmaak on gfx9's one-read constant bus)
DEV float vgpr_const(unsigned bits)
{
    float r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(bits));
    return r;
}

#ifdef SAS_TUNE_WGTIME
// A/B builds only: begin / end tick (100 MHz) and list length of every workgroup of k_tile_lazy,
// by launch index (tools/wg_time.py).  Plain stores: same-address atomics would dominate the kernel.
constexpr int kDbgWgMax = 16384;
__device__ unsigned long long g_dbg_wg[3 * kDbgWgMax];
// ... and where the time went (ticks, thread 0's view): [0] ordering the chunks, [1] compositing (staging, masks, queues, trips),
// [2] the trips alone, [3] batches
__device__ unsigned long long g_dbg_ph[8 * kDbgWgMax];   // [4..7] quad layout, per batch: first barrier, mask + staging, second barrier, queue building
extern "C" int sas_debug_wg(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_wg), sizeof(unsigned long long) * 3 * (size_t)(n < kDbgWgMax ? n : kDbgWgMax)) == hipSuccess ? 0 : -1;
}
extern "C" int sas_debug_wg_phases(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_ph), sizeof(unsigned long long) * 8 * (size_t)(n < kDbgWgMax ? n : kDbgWgMax)) == hipSuccess ? 0 : -1;
}
#define PH_ADD(i, v) do { if (threadIdx.x == 0 && blockIdx.x < kDbgWgMax) g_dbg_ph[8 * blockIdx.x + (i)] += (v); } while (0)
#define PH_T() wall_clock64()
// ... and an EXCLUSIVE partition of thread 0's wall time (ordinary layout): PH_LAP(i) books the ticks since the previous lap on slot i.
// [0] tile order + offsets, [1] min / max pass, [2] histogram pass, [3] bucket selection, [4] partition, [5] collect, [6] ordering a chunk,
// [7] first batch of a chunk: records + barrier, [8] staging + masks + barrier, [9] queue building, [10] trips, [11] batch barrier (other waves),
// [12] epilogue, [13] short list: loads + ordering
__device__ unsigned long long g_dbg_lap[16 * kDbgWgMax];
extern "C" int sas_debug_wg_laps(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_lap), sizeof(unsigned long long) * 16 * (size_t)(n < kDbgWgMax ? n : kDbgWgMax)) == hipSuccess ? 0 : -1;
}
#define PH_LAP(i) do { if (threadIdx.x == 0 && blockIdx.x < kDbgWgMax) { const unsigned long long now_ = wall_clock64(); g_dbg_lap[16 * blockIdx.x + (i)] += now_ - ph_lap_; ph_lap_ = now_; } } while (0)
#else
#define PH_ADD(i, v) do { (void)(v); } while (0)
#define PH_T() 0ull
#define PH_LAP(i) do { } while (0)
#endif
#ifdef SAS_TUNE_STATS
// A/B builds only: [0] wave-iterations of the compositing loop, [1] trips on which a pixel terminated,
// [2] pixels that terminated, [3] lanes that composited, [4] staged entries, [5] queued (entry, block) pairs,
// [6] / [7] wave cycles waiting at the batch barrier / inside the compositing loop,
// [8] (entry, 16-lane group) slots in which at least one lane composited, [9] slots that held the sentinel,
// [10] slots of a group whose 16 pixels had all terminated (after the trip), [11] slots of a live group in which no lane passed the alpha test (includes sentinels)
// [12] staged entries whose 16-bit block mask is empty (the splat reaches no 4x4 block of the tile), [13] staged entries in all waves' eyes (256 per batch)
__device__ unsigned long long g_dbg[16];
extern "C" int sas_debug_counters(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(g_dbg)) != hipSuccess) return -1;

    if (reset) {
        unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define DBG_ADD(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_dbg[i], (unsigned long long)(v)); } while (0)
#else
#define DBG_ADD(i, v) do { } while (0)
#endif

// LDS of the compositing loop: one staged batch of 256 records (+ one sentinel record that no
// pixel accepts) and, per wave, one compacted queue for each of its four 4x4 blocks.
constexpr int kStage = 257;
struct BlendLds {
    float4 *q0, *q1, *q2;       // [257] each
    unsigned *mask;             // [256] 16 block bits per staged entry
    unsigned short *queue;      // [4 waves][4 blocks][256] byte offsets (16 * entry) into q0/q1/q2
};
constexpr int kBlendLdsBytes = 3 * kStage * 16 + 256 * 4 + 16 * 256 * 2;   // 21552
DEV BlendLds blend_lds(unsigned char *raw)
{
    BlendLds L;
    L.q0 = reinterpret_cast<float4 *>(raw);
    L.q1 = L.q0 + kStage;
    L.q2 = L.q1 + kStage;
    L.mask = reinterpret_cast<unsigned *>(L.q2 + kStage);
    L.queue = reinterpret_cast<unsigned short *>(L.mask + 256);
    return L;
}

// Composite entries [0, count) of a depth-ordered list onto this thread's pixel.  Wave w owns the
// 8x8 quadrant w of the tile, its 16-lane group g the quadrant's 4x4 block g (pixel_of).  Per batch
// of 256 entries every thread stages one 48-byte record with its 16-bit block mask; each wave
// ballots, per block, the entries that name it into that block's queue, and the four groups walk
// their own queues in lockstep, front to back (a group that runs out reads the sentinel record,
// which no pixel accepts).  Returns true when every pixel of the tile has terminated (uniform over
// the workgroup).  `slot_at(i)` gives the storage slot of entry i.
#ifndef SAS_TUNE_LATE_COLOUR
#define SAS_TUNE_LATE_COLOUR 0
#endif
template <bool FAST_EXP, typename SlotAt>
DEV bool blend_range(const SasFrame &f, long long n_gauss, int tx, int ty, const PixConst pc, int count, SlotAt slot_at,
                     const BlendLds &L, PixState &p, bool &wdone, unsigned long long &ph_lap_, unsigned &sync_phase)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float4 ra = make_float4(0, 0, 0, 0), rb = ra, rc = ra;
    bool have = false;
#ifdef SAS_TUNE_STATS
    unsigned dbg_rx = 0u, dbg_ry = 0u;
#endif
    const float sE5 = vgpr_const(0x3aafa464u);   // leading exp coefficient
    // only issues the loads: nothing here may depend on their results
    auto fetch = [&](int at) {
        const int idx = at + tid;
        have = idx < count;
        if (have) {
            long long id = slot_at(idx);
            if (!SAS_IN(id, n_gauss, 201) || id >= n_gauss) id = n_gauss - 1;   // never dereference a bad index
            ra = f.rec[SAS_RS * id + 0];
            rb = f.rec[SAS_RS * id + 1];
            rc = f.col[SAS_CS * id];
#ifdef SAS_TUNE_STATS
            dbg_rx = f.info[id].z;
            dbg_ry = f.info[id].w;
#endif
        }
    };
    if (count > 0) fetch(0);
    if (tid == 0) {   // sentinel record: opacity 0 (built here from opaque registers: hipcc otherwise keeps the
        // constant vectors alive across the tile's rounds and spills them)
        const float z0 = vgpr_const(0u);
        L.q0[256] = make_float4(z0, z0, z0, z0);
        L.q1[256] = make_float4(z0, z0, z0, z0);
        L.q2[256] = make_float4(z0, z0, z0, z0);
    }
    const float X0 = (float)(tx * SAS_TILE) + kTileCentre, Y0 = (float)(ty * SAS_TILE) + kTileCentre;   // the polynomial's origin: the tile's centre
    // this lane's block: bit in the entry masks, and its queue
    const int grp = lane >> 4;
    const int my_bit = ((wv & 1) * 2 + (grp & 1)) + 4 * ((wv >> 1) * 2 + (grp >> 1));
    unsigned short *wq = L.queue + wv * 1024;        // this wave's four queues
    const unsigned short *myq = wq + grp * 256;
    const char *q0b = reinterpret_cast<const char *>(L.q0);
    const char *q1b = reinterpret_cast<const char *>(L.q1);
    const char *q2b = reinterpret_cast<const char *>(L.q2);
    bool all_done = false;
#ifdef SAS_TUNE_STATS
    long long t_loop_end = 0;
#endif
    for (int at = 0; at < count; at += 256) {
        // the previous batch is fully consumed; leave once every wave has terminated
        const bool every_done = wg_reduce_wave_flags<true>(wdone, sync_phase);   // (wdone is wave-uniform)
        PH_LAP(11);
#ifdef SAS_TUNE_STATS
        if (t_loop_end) DBG_ADD(6, clock64() - t_loop_end);   // [6] cycles waves waited for the slowest wave of a batch
#endif
        if (every_done) { all_done = true; break; }
        unsigned ment = 0u;
        {
            // (the tile's coordinates, opaque per batch: hipcc otherwise hoists block_mask16's eight block-centre coordinates out of
            // the tile's loops and, short of registers, parks them in scratch -- eight reloads with a full vmcnt wait per batch)
            int txo = tx, tyo = ty;
            asm volatile("" : "+s"(txo), "+s"(tyo));
            if (have) ment = block_mask16(txo, tyo, ra.x, ra.y, ra.z, ra.w, rb.x, rb.z);
        }
#ifdef SAS_TUNE_STATS
        DBG_ADD(12, __popcll(__ballot(have && ment == 0u)));
        DBG_ADD(13, __popcll(__ballot(have)));
        {   // how many queued (entry, block) pairs would the splat's own bounding box (its radii) remove?
            unsigned keep = 0u;
            if (have) {
                const float rx = (float)(int)dbg_rx, ry = (float)(int)dbg_ry;
                for (int b = 0; b < 16; ++b) {
                    const float x0 = (float)(tx * SAS_TILE + 4 * (b & 3)) + 0.5f, y0 = (float)(ty * SAS_TILE + 4 * (b >> 2)) + 0.5f;
                    const bool out = x0 > ra.x + rx || x0 + 3.0f < ra.x - rx || y0 > ra.y + ry || y0 + 3.0f < ra.y - ry;
                    if (!out) keep |= 1u << b;
                }
            }
            unsigned a = __popc(ment), c = __popc(ment & keep);
            for (int d = 32; d > 0; d >>= 1) { a += __shfl_xor(a, d); c += __shfl_xor(c, d); }
            DBG_ADD(14, a);
            DBG_ADD(15, c);
        }
#endif
        // Contract T6 (round 5): sigma on (dx, dy) = (u - x, v - y) in the frame of the tile's centre, u = mx - X0, v = my - Y0:
        //   sigma = fma(dx, fma(B, dy, hA dx), (hC dy) dy),   hA = A/2, hC = C/2
        // -- seven operations per pixel-splat pair.  (Rounds 1-4 expanded it into a polynomial in (x, y): five, but its terms
        // cancel; the (dx, dy) form costs the tile kernel 3 % and sits at the float32 noise floor of gsplat's written form.)
        {
            const float u = ra.x - X0, v = ra.y - Y0;
            const float A = ra.z, B = ra.w, C = rb.x;
            // (what a trip reads is packed: q0 whole, q1's first half -- a 16-byte and an 8-byte LDS read per entry, six registers)
            L.q0[tid] = make_float4(u, v, 0.5f * A, B);
            L.q1[tid] = make_float4(0.5f * C, rb.y, 0.0f, 0.0f);   // .y opacity
            L.q2[tid] = make_float4(rc.x, rc.y, rc.z, rb.w);     // colour, depth
            L.mask[tid] = ment;
        }
        if (!wdone) {   // all four queues of the wave start as sentinels (2 KiB: 32 bytes per lane)
            uint4 *z = reinterpret_cast<uint4 *>(wq) + 2 * lane;
            // (materialised here: hipcc otherwise keeps the four registers of the constant alive across the
            // whole tile and spills them to scratch, one 16-byte reload per batch)
            const unsigned sw = __float_as_uint(vgpr_const((256u << 4) | ((256u << 4) << 16)));
            z[0] = make_uint4(sw, sw, sw, sw);
            z[1] = make_uint4(sw, sw, sw, sw);
        }
        __syncthreads();
        PH_LAP(at == 0 ? 7 : 8);   // (the first batch's records were requested just before: their latency is in this lap)
        if (at + 256 < count) fetch(at + 256);   // next batch in flight while this one is blended
        if (!wdone) {
            const int cnt = (count - at) < 256 ? (count - at) : 256;
            int qn[4] = {0, 0, 0, 0};
            // 16 * lane, opaque to the optimiser: it otherwise hoists the four queue values 16 (64 j + lane) out
            // of the batch loop and spills them (16 scratch reloads per batch, each with a full vmcnt wait)
            unsigned lane16 = (unsigned)lane << 4;
            asm volatile("" : "+v"(lane16));
            // bit of the wave's block 0 in the entry masks, opaque per batch (hipcc otherwise keeps the four masks 1 << bit alive
            // across the tile's loops and spills them: sixteen scratch reloads per batch)
            unsigned bit0 = (unsigned)((wv & 1) * 2 + 8 * (wv >> 1));
            asm volatile("" : "+v"(bit0));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (64 * j >= cnt) break;   // partial batch (uniform)
                const int e = j * 64 + lane;
                const unsigned me = e < cnt ? L.mask[e] : 0u;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned bit = bit0 + (unsigned)((g & 1) + 4 * (g >> 1));
                    const bool has = (me >> bit) & 1u;
                    const unsigned long long m = __ballot(has);
                    const int below = (int)mbcnt64(m);
                    if (has && SAS_IN(qn[g] + below, 256, 202)) wq[g * 256 + qn[g] + below] = (unsigned short)(lane16 + 1024u * j);
                    qn[g] += (int)__popcll(m);
                }
            }
            // a 4x4 block whose 16 pixels have all terminated does not prolong the walk (13 % of the queue slots
            // walked at config 3 belonged to such blocks; a trip lasts as long as the wave's longest queue)
            {
                const unsigned long long dm = __ballot(pix_dead(p));
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (((dm >> (16 * g)) & 0xffffull) == 0xffffull) qn[g] = 0;
            }
            const int kmax = max(max(qn[0], qn[1]), max(qn[2], qn[3]));
            PH_LAP(9);
#ifdef SAS_TUNE_STATS
            const long long t_loop = clock64();
#endif
            DBG_ADD(5, qn[0] + qn[1] + qn[2] + qn[3]);
            if (wv == 0) DBG_ADD(4, cnt);
            // Two queue entries per trip: their record loads, sigmas and exponentials are independent
            // (one wave alone cannot hide the two dependent LDS round trips of an entry); only the
            // transmittance chain is sequential.  A queue of odd length ends on the sentinel.
#if SAS_TUNE_LATE_COLOUR
            struct Trip { float4 K0, K1; float2 H0, H1; unsigned o0, o1; };
#else
            struct Trip { float4 K0, K1, C0, C1; float2 H0, H1; };
#endif
            auto load_trip = [&](Trip &t, int k, unsigned pair) {   // pair = queue entries k, k + 1
                unsigned off0 = pair & 0xffffu, off1 = pair >> 16;
#ifdef SAS_TUNE_STATS
                {
                    const unsigned long long e0 = __ballot(off0 == (256u << 4)), e1 = __ballot(off1 == (256u << 4));
                    DBG_ADD(9, (__popcll(e0) + __popcll(e1)) / 16);
                }
#endif
                if (!SAS_IN(k + 1, 256, 203) || !SAS_IN(off0 >> 4, kStage, 204) || !SAS_IN(off1 >> 4, kStage, 205)) off0 = off1 = 256u << 4;
                t.K0 = *reinterpret_cast<const float4 *>(q0b + off0);
                t.H0 = *reinterpret_cast<const float2 *>(q1b + off0);
                t.K1 = *reinterpret_cast<const float4 *>(q0b + off1);
                t.H1 = *reinterpret_cast<const float2 *>(q1b + off1);
#if SAS_TUNE_LATE_COLOUR
                t.o0 = off0; t.o1 = off1;   // (experiment: colour and depth are fetched when the alphas are known -- 8 registers fewer at the trip's peak)
#else
                t.C0 = *reinterpret_cast<const float4 *>(q2b + off0);   // colour, depth
                t.C1 = *reinterpret_cast<const float4 *>(q2b + off1);
#endif
            };
            // returns true when every pixel of the wave has terminated
            auto composite_trip = [&](const Trip &t) -> bool {
                const float dx0 = t.K0.x - p.x, dy0 = t.K0.y - pc.y, dx1 = t.K1.x - p.x, dy1 = t.K1.y - pc.y;
                const float sg0 = fma_(dx0, fma_(t.K0.w, dy0, t.K0.z * dx0), (t.H0.x * dy0) * dy0);
                const float sg1 = fma_(dx1, fma_(t.K1.w, dy1, t.K1.z * dx1), (t.H1.x * dy1) * dy1);
                // Every decision below is a per-lane select on a value, not a wave mask combined on the
                // scalar unit (which the CU's four SIMDs share: a scalar instruction costs as much issue time
                // as a vector one).  A lane the splat does not reach has a large sigma: the contract's clamp
                // of the exponent argument makes its alpha underflow and fail the 1/255 test by itself; a
                // parked pixel (x = NaN -> sigma = NaN) is clamped the same way (max returns the number).
                DBG_ADD(0, 2);
                float E0, E1;
                if (FAST_EXP) { E0 = __expf(fmaxf(-sg0, -86.0f)); E1 = __expf(fmaxf(-sg1, -86.0f)); }
                else { E0 = c_expf_neg(fmaxf(-sg0, -86.0f), sE5); E1 = c_expf_neg(fmaxf(-sg1, -86.0f), sE5); }
                const float al0 = fminf(kMaxAlpha, t.H0.y * E0);
                const float al1 = fminf(kMaxAlpha, t.H1.y * E1);
                // weight w = alpha T (0 when the splat is skipped), next T = T - w, for both entries as if no
                // pixel terminated; T only falls, so one test of the last T tells whether any did
                const float w0 = (al0 < kAlphaThr) ? 0.0f : al0 * p.T;
                const float nT0 = p.T - w0;
                const float w1 = (al1 < kAlphaThr) ? 0.0f : al1 * nT0;
                const float nT1 = nT0 - w1;
                float vis0 = w0, vis1 = w1, Tn = nT1;
                bool all_dead = false;
                if (__ballot(nT1 <= kTStop)) {   // 31 % of the trips at configs 2 and 3
                    // the splat that ends a pixel is not added, and the pixel takes nothing after it
                    // (a live pixel has T > 1e-4, so a skipped splat never stops it)
                    const bool stop0 = nT0 <= kTStop, stopped = nT1 <= kTStop;
                    vis0 = stop0 ? 0.0f : w0;
                    vis1 = stopped ? 0.0f : w1;
                    Tn = stop0 ? p.T : (stopped ? nT0 : nT1);
                    if (stopped) p.x = __builtin_nanf("");
                    all_dead = __all(pix_dead(p));
                }
                p.T = Tn;
                // lanes that do not composite add with weight +0: fmaf(c, 0, x) == x for the finite
                // colours and depths of the path
#if SAS_TUNE_LATE_COLOUR
                __builtin_amdgcn_sched_barrier(0);
                const float4 lc0 = *reinterpret_cast<const float4 *>(q2b + t.o0), lc1 = *reinterpret_cast<const float4 *>(q2b + t.o1);
                p.r = fma_(lc1.x, vis1, fma_(lc0.x, vis0, p.r));
                p.g = fma_(lc1.y, vis1, fma_(lc0.y, vis0, p.g));
                p.b = fma_(lc1.z, vis1, fma_(lc0.z, vis0, p.b));
                p.d = fma_(lc1.w, vis1, fma_(lc0.w, vis0, p.d));
#else
                p.r = fma_(t.C1.x, vis1, fma_(t.C0.x, vis0, p.r));
                p.g = fma_(t.C1.y, vis1, fma_(t.C0.y, vis0, p.g));
                p.b = fma_(t.C1.z, vis1, fma_(t.C0.z, vis0, p.b));
                p.d = fma_(t.C1.w, vis1, fma_(t.C0.w, vis0, p.d));
#endif
#ifdef SAS_TUNE_STATS
                {
                    const unsigned long long m0 = __ballot(vis0 > 0.0f), m1 = __ballot(vis1 > 0.0f);
                    const unsigned long long a0 = __ballot(al0 >= kAlphaThr), a1m = __ballot(al1 >= kAlphaThr);
                    const unsigned long long dm = __ballot(pix_dead(p)), sb = __ballot(nT1 <= kTStop);
                    int used = 0, gdead = 0, nopass = 0;
                    for (int gq = 0; gq < 4; ++gq) {
                        const bool dead = ((dm >> (16 * gq)) & 0xffffull) == 0xffffull;
                        used += (int)(((m0 >> (16 * gq)) & 0xffffull) != 0) + (int)(((m1 >> (16 * gq)) & 0xffffull) != 0);
                        gdead += dead ? 2 : 0;
                        if (!dead) nopass += (int)(((a0 >> (16 * gq)) & 0xffffull) == 0) + (int)(((a1m >> (16 * gq)) & 0xffffull) == 0);
                    }
                    DBG_ADD(3, __popcll(m0) + __popcll(m1));
                    DBG_ADD(8, used);
                    DBG_ADD(10, gdead);
                    DBG_ADD(11, nopass);
                    DBG_ADD(1, sb != 0);
                    DBG_ADD(2, __popcll(sb));
                }
#endif
                return all_dead;
            };
            // One exit: the trip counter lives on the scalar unit, and a wave whose pixels have all terminated
            // jumps it to the end (a `break` makes hipcc merge two exits through lane masks: six more
            // scalar instructions per trip, -2 % frames/s).
            const int kend = __builtin_amdgcn_readfirstlane(kmax);
            if (kend > 0) {
                int k = 0;
                // the next trip's pair of queue entries is fetched a trip ahead: one LDS round trip leaves
                // the wave's dependent chain for one more VGPR (-1 % at config 3, -3 % on the Gym cameras)
                const unsigned *myq2 = reinterpret_cast<const unsigned *>(myq);
                unsigned pair = myq2[0];
                do {
                    Trip t;
                    load_trip(t, k, pair);
                    k += 2;
                    pair = myq2[min(k, 254) >> 1];
                    if (composite_trip(t)) k = kend;
                } while (k < kend);
            }
            wdone = __all(pix_dead(p));
            PH_LAP(10);
#ifdef SAS_TUNE_STATS
            t_loop_end = clock64();
            DBG_ADD(7, t_loop_end - t_loop);   // [7] cycles in the compositing loop
#endif
        }
    }
    if (!all_done) all_done = wg_reduce_wave_flags<true>(wdone, sync_phase);   // also fences the staging buffers
    return all_done;
}

// blend_range for the quad layout (pixel_of_quad): the same batches of 256 staged records, a 4-bit block mask
// per entry, ONE queue per wave, and trips of FOUR entries: lane (q, e) evaluates alpha of entry k + e for pixel
// q; the transmittance chain then runs over the four entries in order, identically in the four lanes of the
// quad, taking alpha and colour of entry j from lane j through DPP quad broadcasts.  Same arithmetic per
// (pixel, splat), same order per pixel as blend_range: bit-identical images.  A wave alone on its SIMD issues one
// instruction per ~7 cycles whatever it holds in flight, so the four-wide alpha evaluation halves the
// instructions per composited entry of the latency-bound small frames; it wastes lanes where the chip is full.
template <int J>
DEV float quad_bcast(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, true));
}
template <bool FAST_EXP, typename SlotAt>
DEV bool blend_range_quad(const SasFrame &f, long long n_gauss, int tx, int ty, int qd, const PixConst pc, int count, SlotAt slot_at,
                          const BlendLds &L, PixState &p, bool &wdone, unsigned &sync_phase)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float4 ra = make_float4(0, 0, 0, 0), rb = ra, rc = ra;
    bool have = false;
    const float sE5 = vgpr_const(0x3aafa464u);
    auto fetch = [&](int at) {
        const int idx = at + tid;
        have = idx < count;
        if (have) {
            long long id = slot_at(idx);
#if defined(SAS_TUNE_ABLATE) && (SAS_TUNE_ABLATE & 8)
            id &= 1023;   // timing experiment: the records come from 48 KiB that stay in the caches (wrong images)
#endif
            if (!SAS_IN(id, n_gauss, 231) || id >= n_gauss) id = n_gauss - 1;
            ra = f.rec[SAS_RS * id + 0];
            rb = f.rec[SAS_RS * id + 1];
            rc = f.col[SAS_CS * id];
        }
    };
    if (count > 0) fetch(0);
    if (tid == 0) {
        const float z0 = vgpr_const(0u);
        L.q0[256] = make_float4(z0, z0, z0, z0);
        L.q1[256] = make_float4(z0, z0, z0, z0);
        L.q2[256] = make_float4(z0, z0, z0, z0);
    }
    const float X0 = (float)(tx * SAS_TILE) + kTileCentre, Y0 = (float)(ty * SAS_TILE) + kTileCentre;   // the polynomial's origin: the tile's centre
    const int e = lane & 3;
    unsigned short *wq = L.queue + wv * 1024;   // this wave's queue: 256 entries (+ the look-ahead's slack inside its 1024)
    const char *q0b = reinterpret_cast<const char *>(L.q0);
    const char *q1b = reinterpret_cast<const char *>(L.q1);
    const char *q2b = reinterpret_cast<const char *>(L.q2);
    bool all_done = false;
    for (int at = 0; at < count; at += 256) {
        const unsigned long long t_b0 = PH_T();
        const bool every_done = wg_reduce_wave_flags<true>(wdone, sync_phase);
        if (every_done) { all_done = true; break; }
        const unsigned long long t_b1 = PH_T();
        PH_ADD(4, t_b1 - t_b0);
        unsigned ment = 0u;
        if (have) ment = block_mask4(tx, ty, qd, ra.x, ra.y, ra.z, ra.w, rb.x, rb.z);
        {   // (contract T6: see blend_range)
            const float u = ra.x - X0, v = ra.y - Y0;
            const float A = ra.z, B = ra.w, C = rb.x;
            L.q0[tid] = make_float4(u, v, 0.5f * A, B);              // (packed as in blend_range: q0 whole, q1's first half)
            L.q1[tid] = make_float4(0.5f * C, rb.y, 0.0f, 0.0f);
            L.q2[tid] = make_float4(rc.x, rc.y, rc.z, rb.w);
            L.mask[tid] = ment;
        }
        if (!wdone) {   // the queue starts as sentinels: 264 entries (a trip reads four, the look-ahead four more)
            const unsigned sw = __float_as_uint(vgpr_const((256u << 4) | ((256u << 4) << 16)));
            uint2 *z = reinterpret_cast<uint2 *>(wq) + lane;
            z[0] = make_uint2(sw, sw);
            if (lane < 2) z[64] = make_uint2(sw, sw);
        }
        const unsigned long long t_b2 = PH_T();
        PH_ADD(5, t_b2 - t_b1);
        __syncthreads();
        const unsigned long long t_b3 = PH_T();
        PH_ADD(6, t_b3 - t_b2);
        if (at + 256 < count) fetch(at + 256);
        if (!wdone) {
            const int cnt = (count - at) < 256 ? (count - at) : 256;
            int qn = 0;
            unsigned lane16 = (unsigned)lane << 4;
            asm volatile("" : "+v"(lane16));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (64 * j >= cnt) break;
                const int ee = j * 64 + lane;
                const unsigned me = ee < cnt ? L.mask[ee] : 0u;
                const bool has = (me >> wv) & 1u;
                const unsigned long long m = __ballot(has);
                const int below = (int)mbcnt64(m);
                if (has && SAS_IN(qn + below, 256, 232)) wq[qn + below] = (unsigned short)(lane16 + 1024u * j);
                qn += (int)__popcll(m);
            }
#if defined(SAS_TUNE_ABLATE) && (SAS_TUNE_ABLATE & 4)
            const int kend = 0;   // timing experiment: staging, masks and queues, but no trips (wrong images)
#else
            const int kend = __builtin_amdgcn_readfirstlane(qn);
#endif
            PH_ADD(3, 1);
            const unsigned long long t_t = PH_T();
            PH_ADD(7, t_t - t_b3);
            if (kend > 0) {
                int k = 0;
                unsigned off = wq[e];
                do {
                    if (!SAS_IN(off >> 4, kStage, 233)) off = 256u << 4;
                    const float4 K = *reinterpret_cast<const float4 *>(q0b + off);
                    const float2 H = *reinterpret_cast<const float2 *>(q1b + off);
                    const float4 C = *reinterpret_cast<const float4 *>(q2b + off);
                    k += 4;
                    off = wq[k + e];   // next trip's entry, one trip ahead
                    const float dxq = K.x - p.x, dyq = K.y - pc.y;
                    const float sg = fma_(dxq, fma_(K.w, dyq, K.z * dxq), (H.x * dyq) * dyq);
                    const float E = FAST_EXP ? __expf(fmaxf(-sg, -86.0f)) : c_expf_neg(fmaxf(-sg, -86.0f), sE5);
                    const float al = fminf(kMaxAlpha, H.y * E);
                    // A skipped splat weighs 0: decided HERE, once per lane for its own entry, so that the chain below is one
                    // multiply (its operand the quad broadcast of entry j's alpha: v_mul_f32_dpp) and one subtraction per
                    // entry instead of broadcast + compare + multiply + select (0 * T == +0 for the finite T of the path:
                    // the same bits as the select's 0).
                    const float alz = (al < kAlphaThr) ? 0.0f : al;
                    // as if no pixel terminated: T only falls, one test of the last T tells
                    const float T0 = p.T;
                    const float w0 = quad_bcast<0>(alz) * T0;
                    const float T1 = T0 - w0;
                    const float w1 = quad_bcast<1>(alz) * T1;
                    const float T2 = T1 - w1;
                    const float w2 = quad_bcast<2>(alz) * T2;
                    const float T3 = T2 - w2;
                    const float w3 = quad_bcast<3>(alz) * T3;
                    const float T4 = T3 - w3;
                    float v0 = w0, v1 = w1, v2 = w2, v3 = w3, Tn = T4;
                    bool all_dead = false;
                    if (__ballot(T4 <= kTStop)) {
                        // the splat that ends a pixel is not added and the pixel takes nothing after it
                        const bool s0 = T1 <= kTStop, s1 = T2 <= kTStop, s2 = T3 <= kTStop, s3 = T4 <= kTStop;   // s0 => s1 => s2 => s3
                        v0 = s0 ? 0.0f : w0;
                        v1 = s1 ? 0.0f : w1;
                        v2 = s2 ? 0.0f : w2;
                        v3 = s3 ? 0.0f : w3;
                        Tn = s0 ? T0 : (s1 ? T1 : (s2 ? T2 : (s3 ? T3 : T4)));
                        if (s3) p.x = __builtin_nanf("");
                        all_dead = __all(pix_dead(p));
                    }
                    p.T = Tn;
                    // acc = fma(colour of entry j (lane j of the quad), weight, acc): the broadcast rides on the
                    // multiply-add as a DPP operand (hipcc leaves a v_mov_dpp + s_nop in front of each otherwise)
#define SAS_QFMAC(ACC_, COL_, WGT_, J) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[" #J "," #J "," #J "," #J "] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(ACC_) : "v"(COL_), "v"(WGT_))
#define SAS_QACC(WGT_, J) SAS_QFMAC(p.r, C.x, WGT_, J); SAS_QFMAC(p.g, C.y, WGT_, J); SAS_QFMAC(p.b, C.z, WGT_, J); SAS_QFMAC(p.d, C.w, WGT_, J)
                    // Two wait states must lie between a VALU write of a VGPR and a DPP read of it (gfx9), and the
                    // hazard recogniser does not look inside inline asm.  C comes straight from ds_read_b128 today; the
                    // s_nop covers a copy of it a future register allocation might put in front of this block (the
                    // statements are volatile: they stay behind it, in this order).
                    asm volatile("s_nop 1");
                    SAS_QACC(v0, 0);
                    SAS_QACC(v1, 1);
                    SAS_QACC(v2, 2);
                    SAS_QACC(v3, 3);
#undef SAS_QACC
#undef SAS_QFMAC
                    if (all_dead) k = kend;
                } while (k < kend);
            }
            PH_ADD(2, PH_T() - t_t);
            wdone = __all(pix_dead(p));
        }
    }
    if (!all_done) all_done = wg_reduce_wave_flags<true>(wdone, sync_phase);
    return all_done;
}

// T0 epilogue for this thread's pixel; returns its expected depth (0 outside the image).
// pack4 (uniform; the ordinary layout, image width and rgb8 base multiples of four): lanes 4 j .. 4 j + 3 hold four pixels that are
// neighbours in a row (pixel_of), all inside the image or all outside: their twelve bytes leave as ONE 12-byte store from lane 4 j
// instead of three one-byte stores from each of the four lanes (three store instructions of 64 scattered bytes per wave).
template <bool PACK4_OK = false>
DEV float write_pixel(const SasOutputs &o, const PixState &p, bool inside, int ix, int iy, int W, unsigned &rgb8_packed, bool pack4 = false)
{
    rgb8_packed = 0u;
    if (!inside) return 0.0f;
    const float a = 1.0f - p.T;
    const float ED = p.d / fmaxf(a, 1e-10f);
    const long long pix = (long long)iy * W + ix;
    if (!SAS_IN(pix, o.n_pixels, 212)) return 0.0f;
    const float w = 1.0f - a;
    float v0 = p.r + w * o.bg[0], v1 = p.g + w * o.bg[1], v2 = p.b + w * o.bg[2];
    v0 = fminf(fmaxf(v0, 0.0f), 1.0f);
    v1 = fminf(fmaxf(v1, 0.0f), 1.0f);
    v2 = fminf(fmaxf(v2, 0.0f), 1.0f);
    // one 12-byte store per pixel: a wave's 8-pixel rows are then whole 96-byte runs instead of
    // three passes of strided 4-byte stores over the same lines
    if (o.rgb) {
        // (hipcc splits a plain 12-byte struct store here into three dword stores)
        const f32x3 v = {v0, v1, v2};
        asm volatile("global_store_dwordx3 %0, %1, off\n\ts_nop 1" ::"v"(o.rgb + 3 * pix), "v"(v) : "memory");
    }
    if (o.alpha) o.alpha[pix] = a;
    if (o.depth) o.depth[pix] = ED;
    if (o.rgb8 || o.rgb8_host) {
        const unsigned b0 = (unsigned)(int)floorf(fma_(v0, 255.0f, 0.5f)), b1 = (unsigned)(int)floorf(fma_(v1, 255.0f, 0.5f)),
                       b2 = (unsigned)(int)floorf(fma_(v2, 255.0f, 0.5f));
        rgb8_packed = b0 | (b1 << 8) | (b2 << 16);
        if (o.rgb8) {
            if (PACK4_OK && pack4) {
                // (the four lanes of a DPP quad are in this branch together: same row, x = 4 j .. 4 j + 3, width a multiple of four)
                const unsigned p0 = rgb8_packed;
                const unsigned p1 = __float_as_uint(quad_bcast<1>(__uint_as_float(p0))), p2 = __float_as_uint(quad_bcast<2>(__uint_as_float(p0))),
                               p3 = __float_as_uint(quad_bcast<3>(__uint_as_float(p0)));
                if ((threadIdx.x & 3) == 0) {
                    typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
                    const u32x3 v = {p0 | (p1 << 24), (p1 >> 8) | (p2 << 16), (p2 >> 16) | (p3 << 8)};
                    *reinterpret_cast<u32x3 *>(o.rgb8 + 3 * pix) = v;
                }
            } else {
                o.rgb8[3 * pix] = (uint8_t)b0;
                o.rgb8[3 * pix + 1] = (uint8_t)b1;
                o.rgb8[3 * pix + 2] = (uint8_t)b2;
            }
        }
    }
    return ED;
}

// The workgroup's pixels (a 16x16 tile, or an 8x8 quadrant in the quad layout; every pixel inside the image) as packed
// rows straight to PINNED HOST memory: bytes gathered in LDS, then ROWS x 3 lanes store 16 (8) bytes each.  All threads
// call it; `lds` is free at this point (the compositing is over).
template <int SIDE>   // 16: tile, 8: quadrant
DEV void store_rows_to_host(uint8_t *host, int W, int X0, int Y0, int ox, int oy, bool writer, unsigned rgb8_packed, unsigned char *lds)
{
    constexpr int ROWB = 3 * SIDE;          // 48 or 24 bytes per row
    __syncthreads();                        // the staging buffers of the last batch are no longer read
    if (writer) {
        unsigned char *d = lds + oy * ROWB + 3 * ox;
        d[0] = (unsigned char)rgb8_packed; d[1] = (unsigned char)(rgb8_packed >> 8); d[2] = (unsigned char)(rgb8_packed >> 16);
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 3 * SIDE) {
        const int row = t / 3, seg = t - 3 * row;
        uint8_t *dst = host + ((long long)(Y0 + row) * W + X0) * 3;
        if constexpr (SIDE == 16) reinterpret_cast<uint4 *>(dst)[seg] = reinterpret_cast<const uint4 *>(lds + row * ROWB)[seg];
        else reinterpret_cast<uint2 *>(dst)[seg] = reinterpret_cast<const uint2 *>(lds + row * ROWB)[seg];
    }
}

// per-tile max of the expected depth (reduced over tiles by k_depth_tail); all threads call it
DEV void store_tile_max(const SasFrame &f, int tile, float ED, unsigned *s_wmax)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // (expected depths are >= 0: their order is the order of their bit patterns, as the reduction over waves and tiles already takes it)
    const unsigned maxed = wave_max_u32(__float_as_uint(ED));
    if (lane == 0) s_wmax[wv] = maxed;
    __syncthreads();
    if (tid == 0) f.tile_max[tile] = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
    __syncthreads();
}

// ---- full path, stage 2: composite the complete sorted lists -----------------------------------
template <bool FAST_EXP, bool WANT_MAX>
__global__ __launch_bounds__(256) void k_blend(SasParams P, SasFrame f, long long n_gauss,
                                               const int *tl, const int *range)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[kBlendLdsBytes];
    __shared__ unsigned s_wmax[4];
    const SasCam &c = P.cam;
    const SasOutputs &o = P.out;
    const BlendLds L = blend_lds(s_raw);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned sync_phase = 0u;
    for (int oi = range[0] + (int)blockIdx.x; oi < range[1]; oi += (int)gridDim.x) {
        const int tile = tl[oi];
        const int tx = tile % c.tw, ty = tile / c.tw;
        int ox, oy;
        pixel_of(wv, lane, ox, oy);
        const int ix = tx * SAS_TILE + ox, iy = ty * SAS_TILE + oy;
        const bool inside = ix < c.W && iy < c.H;
        PixState p = pix_init(inside, ox);
        bool wdone = __all(!inside);
        const long long beg = f.tile_offset[tile];
        long long end = f.tile_offset[tile + 1];
        if (end > f.cap) end = f.cap;
        const int *ids = f.sorted_ids + beg;
        unsigned long long ph_lap_ = 0ull;
        blend_range<FAST_EXP>(f, n_gauss, tx, ty, pix_const(ox, oy), (int)(end - beg),
                              [&](int i) { return (long long)(unsigned)ids[i]; }, L, p, wdone, ph_lap_, sync_phase);
        unsigned packed;
        const float ED = write_pixel(o, p, inside, ix, iy, c.W, packed);
        if (o.rgb8_host) store_rows_to_host<16>(o.rgb8_host, c.W, tx * SAS_TILE, ty * SAS_TILE, ox, oy, true, packed, s_raw);
        if (WANT_MAX) store_tile_max(f, tile, ED, s_wmax);
        if (tid == 0) { f.tile_count[tile] = 0; f.tile_big[tile] = 0; }   // the frame's counters leave the frame zeroed (SasFrame invariant)
    }
}

// ================================================================================================
// Production path: lazy depth ordering fused with compositing
// ================================================================================================
// (the SAS_TUNE_* macros exist for A/B builds only: SAS_HIPCC_FLAGS="-DSAS_TUNE_CHUNK=2048" python -m sim_a_splat_amd.build)
#ifndef SAS_TUNE_CHUNK
#define SAS_TUNE_CHUNK 512
#endif
#ifndef SAS_TUNE_OCC
#define SAS_TUNE_OCC 5
#endif
#ifndef SAS_TUNE_U
#define SAS_TUNE_U 8
#endif
constexpr int kChunk = SAS_TUNE_CHUNK;   // entries ordered and composited per round
#ifndef SAS_TUNE_QCHUNK
#define SAS_TUNE_QCHUNK 1024
#endif
constexpr int kChunkQuad = SAS_TUNE_QCHUNK;   // ... in the quad layout
#ifndef SAS_TUNE_QPART
#define SAS_TUNE_QPART 1
#endif
#ifndef SAS_TUNE_RANKMAX
#define SAS_TUNE_RANKMAX 32
#endif
constexpr int kRankMax = SAS_TUNE_RANKMAX;   // largest depth bucket a chunk is ordered by counting (else radix passes)
#ifndef SAS_TUNE_PARTMIN
#define SAS_TUNE_PARTMIN (8 * SAS_TUNE_CHUNK)
#endif
// A list that still holds more than this many keys when its SECOND round starts is laid out by bucket once
// (later rounds then read only their own chunk); shorter remainders are cheaper to re-scan (measured: the
// layout pass + the per-chunk depth gathers cost 3 % at config 2, lists of ~4 000).
constexpr int kPartitionMin = SAS_TUNE_PARTMIN;
constexpr int kLazyThreads = 256;
#ifndef SAS_EMPTY_GROUPS_OFF
#define SAS_EMPTY_GROUPS_OFF 0   // A/B builds: 1 = every tile of an all-empty group takes a workgroup of its own (round 4)
#endif

// Lay the keys of buckets >= b_first out by bucket: slot ids into `ids` at the positions handed out by the
// per-bucket cursors `cur` (LDS, preset to each bucket's start).  A real call, not inlined: it runs once for
// the rare tile that needs many rounds, and inlining it costs the common path registers (+1 % at configs 2, 3).
__device__ __attribute__((noinline)) void partition_by_bucket(const unsigned long long *g, int n, unsigned dmin, int shift, int b_first,
                                                              unsigned *cur, int *ids)
{
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += kLazyThreads) {
        const unsigned long long key = g[i];
        const int b = (int)((hi32(key) - dmin) >> shift);
        if (b >= b_first && SAS_IN(b, 256, 214)) {
            const unsigned pos = atomicAdd(&cur[b], 1u);
            if (SAS_IN(pos, n, 215)) ids[pos] = (int)lo32(key);
        }
    }
    __threadfence_block();
    __syncthreads();
}

template <bool FAST_EXP, bool WANT_MAX, bool QUAD>
DEV void tile_lazy_body(const SasParams &P, const SasFrame &f, long long n_gauss, const int *perm, unsigned wg /* workgroup index within the view */)
{
    // timing experiments only (-DSAS_TUNE_ABLATE=1: no chunk sort, =2: no compositing): wrong images
#ifdef SAS_TUNE_ABLATE
    constexpr int ablate = SAS_TUNE_ABLATE;
#else
    constexpr int ablate = 0;
#endif
    // LDS: the chunk of keys, then a region shared in time by the sort scratch and the blend staging
    // entries ordered and composited per round: in the quad layout (frames of a few hundred tiles: the chip is not
    // full, every round's passes and barriers sit on the chain of a wave that is alone on its SIMD) twice as many
    constexpr int CH = QUAD ? kChunkQuad : kChunk;
    __shared__ unsigned long long ck[CH];                                           // 4 KiB (quad layout: 8 KiB)
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[kBlendLdsBytes];    // 21 KiB
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_wsum[4], s_wmax[4];
    __shared__ unsigned s_mn, s_mx, s_m;
    __shared__ int s_b1;
    const SasCam &c = P.cam;
    const SasOutputs &o = P.out;
    const BlendLds L = blend_lds(s_raw);
    unsigned *cnt = reinterpret_cast<unsigned *>(s_raw);          // [4][256]   (sort phase)
    unsigned *dbase = cnt + 4 * 256;                              // [256]
    unsigned *s_cur = dbase + 256;                                // [256] per-bucket write cursor of the chunk being collected (sort phase)
    __shared__ unsigned s_rem;

#ifdef SAS_TUNE_WGTIME
    const unsigned long long t_wg0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x < kDbgWgMax)
        for (int k = 0; k < 8; ++k) g_dbg_ph[8 * blockIdx.x + k] = 0ull;
#endif
    // QUAD: the frame is binned in 8-pixel tiles (c.tile_px == 8): this workgroup's tile is the 8x8 quadrant qd of the
    // contract's 16-pixel tile (tx, ty), whose origin the sigma polynomials refer to (pixel_of_quad); the list is its own
    // single-pass frames: the projection's tail orders GROUPS of four consecutive tiles (by their longest list); two-pass frames: tiles
    // (the grid is the tile count rounded up to whole groups: the last group may be ragged; all of this is uniform)
    int tile;
    if (f.seg > 0) {
#if SAS_TILE_GROUP == 1
        if (wg >= (unsigned)f.n_tiles) return;
        tile = f.tile_order[wg];
#else
        // Groups whose four tiles are all empty are the last class of the order; class_cursor[15] says where it starts.  Such a
        // group costs ONE workgroup that paints its four tiles' background (the other three leave after a scalar load), instead
        // of four that each fetch their tile, its count and paint 256 pixels: the empty tiles were 6 % of the kernel's slot time
        // (3 880 of config 3's 8 160 tiles), which is what frames in flight compete for.  A blocking frame alone on the GPU keeps a
        // workgroup per tile (SasFrame::group_fill): there the kernel's END counts, and four tiles in a row lengthen it.
        if (!QUAD && !SAS_EMPTY_GROUPS_OFF && f.group_fill && (int)(wg >> 2) >= f.class_cursor[15]) {
            if (wg & 3u) return;
            const int g4 = 4 * f.tile_order[wg >> 2];
            const SasCam &cc = P.cam;
            const SasOutputs &oo = P.out;
            const int tid_ = threadIdx.x, lane_ = tid_ & 63, wv_ = tid_ >> 6;
            int ox_, oy_;
            pixel_of(wv_, lane_, ox_, oy_);
            const bool pack4 = oo.rgb8 && (cc.W & 3) == 0 && ((size_t)oo.rgb8 & 3) == 0;
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                const int t = g4 + q;
                if (t >= f.n_tiles) break;
                const int tx_ = t % cc.tw, ty_ = t / cc.tw;
                const int ix_ = tx_ * SAS_TILE + ox_, iy_ = ty_ * SAS_TILE + oy_;
                const bool in_ = ix_ < cc.W && iy_ < cc.H;
                const PixState p0 = pix_init(in_, ox_);
                unsigned packed_;
                const float ED_ = write_pixel<true>(oo, p0, in_, ix_, iy_, cc.W, packed_, pack4);
                if (oo.rgb8_host) store_rows_to_host<16>(oo.rgb8_host, cc.W, tx_ * SAS_TILE, ty_ * SAS_TILE, ox_, oy_, true, packed_, s_raw);
                if (WANT_MAX) store_tile_max(f, t, ED_, s_wmax);
            }
            return;
        }
        tile = 4 * f.tile_order[wg >> 2] + (int)(wg & 3u);
        if (tile >= f.n_tiles) return;
#endif
    } else {
        if (wg >= (unsigned)f.n_tiles) return;
        tile = f.tile_order[wg];
    }
    if (!SAS_IN(tile, f.n_tiles, 213)) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int tx = tile % c.tw, ty = tile / c.tw;
    int qd = 0;
    if (QUAD) {
        qd = (tx & 1) | ((ty & 1) << 1);
        tx >>= 1;
        ty >>= 1;
    }
    int ox, oy;
    if (QUAD) pixel_of_quad(qd, wv, lane, ox, oy);
    else pixel_of(wv, lane, ox, oy);
    int ix = tx * SAS_TILE + ox, iy = ty * SAS_TILE + oy;
    PixConst pc = pix_const(ox, oy);
    bool inside = ix < c.W && iy < c.H;
    const bool writer = !QUAD || (lane & 3) == 0;   // QUAD: the four lanes of a pixel hold the same state, one stores it
    const int out_side = QUAD ? 8 : 16;       // what this workgroup hands out: its quadrant or the whole tile
    PixState p = pix_init(inside, ox);
    bool wdone = __all(!inside);
#ifdef SAS_TUNE_WGTIME
    unsigned long long ph_lap_ = t_wg0;
    if (threadIdx.x == 0 && blockIdx.x < kDbgWgMax)
        for (int k = 0; k < 16; ++k) g_dbg_lap[16 * blockIdx.x + k] = 0ull;
#else
    unsigned long long ph_lap_ = 0ull;
#endif
    unsigned sync_phase = 0u;   // (wg_reduce_wave_flags)
    // composite `count` ordered entries in this kernel's layout
    auto blend = [&](int count, auto slot_at) -> bool {
        const unsigned long long t_b = PH_T();
        bool r;
        if constexpr (QUAD) r = blend_range_quad<FAST_EXP>(f, n_gauss, tx, ty, qd, pc, count, slot_at, L, p, wdone, sync_phase);
        else r = blend_range<FAST_EXP>(f, n_gauss, tx, ty, pc, count, slot_at, L, p, wdone, ph_lap_, sync_phase);
        PH_ADD(1, PH_T() - t_b);
        return r;
    };

    long long beg, end;
    if (f.seg > 0) {   // (uniform) single-pass binning: the tile's own segment, as many keys as were counted (at most the segment)
        beg = (long long)tile * f.seg;
        end = beg + min(f.tile_count[tile], f.seg);   // (tile_big stays zero in this mode)
    } else {
        beg = f.tile_offset[tile];
        end = f.tile_offset[tile + 1];
        if (end > f.cap) end = f.cap;
    }
    const int n = (int)(end - beg);
    const unsigned long long *g = f.keys + beg;
    if (n >= 0) PH_LAP(0);   // (n: the dependent loads of the tile's index and its offsets have returned)
    // the tile kernel's waves issue ahead of the co-resident binning waves of the next frames (+1.2 % frames/s at config 3:
    // what a step costs is the tile kernel's slot time, docs/EXPERIMENTS.md s5.30; the reverse priority costs 0.5 %)
    __builtin_amdgcn_s_setprio(3);

    if (n > 0 && n <= CH) {
        // ---- short list: one pass, keys loaded once into registers, whole list is the chunk
        constexpr int NK = CH / kLazyThreads;
        if (tid == 0) { s_mn = ~0u; s_mx = 0u; }
        s_hist[tid] = 0u;
        __syncthreads();
        unsigned long long kk[NK];
        unsigned mn = ~0u, mx = 0u;
        // (the list's base, opaque per phase: hipcc otherwise forms this thread's key address once per tile and parks the 64-bit
        // value in scratch between the phases that read keys)
        const unsigned long long *g1 = g;
        asm volatile("" : "+s"(g1));
#pragma unroll
        for (int u = 0; u < NK; ++u) {
            const int i = u * kLazyThreads + tid;
            kk[u] = (i < n) ? g1[i] : ~0ull;
            if (i < n) { mn = min(mn, hi32(kk[u])); mx = max(mx, hi32(kk[u])); }
        }
        mn = wave_min_u32(mn); mx = wave_max_u32(mx);   // (DPP steps: sas_device.h)
        if (lane == 0) { atomicMin(&s_mn, mn); atomicMax(&s_mx, mx); }
        __syncthreads();
        const unsigned dmin = s_mn, span = s_mx - s_mn;
#ifdef SAS_TUNE_SHORT_RADIX
#pragma unroll
        for (int u = 0; u < NK; ++u) {
            const int i = u * kLazyThreads + tid;
            if (i < n) ck[i] = kk[u] - ((unsigned long long)dmin << 32);
        }
        __syncthreads();
        if (!(ablate & 1)) lds_radix_sort<4, NK>(ck, n, span, perm, cnt, dbase, s_wsum);
#else
        // Ordered as a chunk of a long list is: 256 depth buckets over the list's range, keys placed grouped by
        // bucket, every key ranked inside its own bucket (a handful of compares) -- a third of the instructions
        // of three radix passes.  A crowded bucket (coplanar splats) falls back to the radix passes.
        const int sbits = span ? 32 - __clz(span) : 0;
        const int shift = sbits > 8 ? sbits - 8 : 0;
#pragma unroll
        for (int u = 0; u < NK; ++u)
            if (u * kLazyThreads + tid < n && SAS_IN((hi32(kk[u]) - dmin) >> shift, 256, 219)) atomicAdd(&s_hist[(hi32(kk[u]) - dmin) >> shift], 1u);
        __syncthreads();
        {
            const unsigned hv = s_hist[tid];
            unsigned incl = wave_inclusive_sum_u32(hv);
            if (lane == 63) s_wsum[wv] = incl;
            __syncthreads();
            for (int w = 0; w < wv; ++w) incl += s_wsum[w];
            s_cur[tid] = incl - hv;   // start of bucket tid
            const bool big = wg_reduce_wave_flags<false>(__any(hv > (unsigned)kRankMax), sync_phase);
#pragma unroll
            for (int u = 0; u < NK; ++u) {
                if (u * kLazyThreads + tid < n) {
                    const unsigned rel = hi32(kk[u]) - dmin;
                    if (SAS_IN(rel >> shift, 256, 220)) {
                        const unsigned pos = atomicAdd(&s_cur[rel >> shift], 1u);   // leaves the END of every bucket
                        if (SAS_IN(pos, CH, 221)) ck[pos] = ((unsigned long long)rel << 32) | lo32(kk[u]);
                    }
                }
            }
            __syncthreads();
            if (!(ablate & 1)) {
                const unsigned long long t_s = PH_T();
                if (!big) lds_bucket_rank_sort<NK>(ck, n, 0, shift, s_hist, s_cur, perm);
                else lds_radix_sort<4, NK>(ck, n, span, perm, cnt, dbase, s_wsum);
                PH_ADD(0, PH_T() - t_s);
            }
        }
#endif
        PH_LAP(13);
        if (!(ablate & 2)) blend(n, [&](int i) { return (long long)lo32(ck[i]); });
    } else if (n > CH) {
        // ---- long list: every pass over the keys keeps U independent loads per thread in flight
        constexpr int U = SAS_TUNE_U;
        const unsigned long long t_p = PH_T();
        if (tid == 0) { s_mn = ~0u; s_mx = 0u; }
        s_hist[tid] = 0u;
        __syncthreads();
        // Lists of up to kKeyCache keys -- most of the long ones -- are read from memory ONCE: the min / max pass parks them in the
        // part of the staging region that the ordering scratch leaves free (15 KiB), the histogram and the first round's collect
        // pass read them from there (two dependent global round trips per tile less).  The first compositing overwrites the copy:
        // later rounds (the exception: the first chunk did not saturate the tile) read the segment again.
        constexpr int kKeyCache = (kBlendLdsBytes - 6 * 1024) / 8;   // behind cnt / dbase / s_cur (6 KiB)
        unsigned long long *const kc = reinterpret_cast<unsigned long long *>(s_raw + 6 * 1024);
        const bool cached = n <= kKeyCache;   // (uniform)
        unsigned mn = ~0u, mx = 0u;
        if (cached) {
            const unsigned long long *g2 = g;
            asm volatile("" : "+s"(g2));
            for (int i0 = 0; i0 < n; i0 += kLazyThreads * U) {
                unsigned long long kk[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u * kLazyThreads + tid;
                    kk[u] = (i < n) ? g2[i] : 0ull;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u * kLazyThreads + tid;
                    if (i < n) { mn = min(mn, hi32(kk[u])); mx = max(mx, hi32(kk[u])); if (SAS_IN(i, kKeyCache, 222)) kc[i] = kk[u]; }
                }
            }
        } else {
            for (int i0 = 0; i0 < n; i0 += kLazyThreads * U) {
                unsigned dd[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u * kLazyThreads + tid;
                    dd[u] = (i < n) ? hi32(g[i]) : 0u;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (i0 + u * kLazyThreads + tid < n) { mn = min(mn, dd[u]); mx = max(mx, dd[u]); }
            }
        }
        mn = wave_min_u32(mn); mx = wave_max_u32(mx);   // (DPP steps: sas_device.h)
        if (lane == 0) { atomicMin(&s_mn, mn); atomicMax(&s_mx, mx); }
        __syncthreads();
        PH_LAP(1);
        const unsigned dmin = s_mn, span = s_mx - s_mn;
        const int sbits = span ? 32 - __clz(span) : 0;
        const int shift = sbits > 8 ? sbits - 8 : 0;   // 256 depth buckets over the tile's range
        if (cached) {
            for (int i = tid; i < n; i += kLazyThreads) {
                const unsigned dd = hi32(kc[i]);
                if (SAS_IN((dd - dmin) >> shift, 256, 208)) atomicAdd(&s_hist[(dd - dmin) >> shift], 1u);
            }
        } else {
            for (int i0 = 0; i0 < n; i0 += kLazyThreads * U) {
                unsigned dd[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u * kLazyThreads + tid;
                    dd[u] = (i < n) ? hi32(g[i]) : 0u;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (i0 + u * kLazyThreads + tid < n && SAS_IN((dd[u] - dmin) >> shift, 256, 208)) atomicAdd(&s_hist[(dd[u] - dmin) >> shift], 1u);
            }
        }
        __syncthreads();
        PH_LAP(2);
        int b_next = 0;
        if (!QUAD) PH_ADD(2, PH_T() - t_p);   // (ordinary layout: [2] = the passes over the keys, [3] = of which the rounds' collect passes)
        bool bail = false;
        bool partitioned = false;                 // the keys left after the first round have been laid out by bucket
        int p_consumed = 0;                       // ... and this many of them have been composited since
        int *const ids = f.sorted_ids + beg;      // ... as storage slots in the tile's (otherwise unused) id segment
        while (!bail) {
            // ---- next bucket range [b0, b1]: b0 = first non-empty bucket >= b_next, b1 = last bucket
            //      whose running count from b0 stays <= CH.  Thread t owns bucket t.
            unsigned my_hv, my_incl;
            {
                const unsigned hv = (tid >= b_next) ? s_hist[tid] : 0u;
                unsigned incl = wave_inclusive_sum_u32(hv);
                if (lane == 63) s_wsum[wv] = incl;
                __syncthreads();
                for (int w = 0; w < wv; ++w) incl += s_wsum[w];        // inclusive count of buckets b_next..tid
                my_hv = hv;
                my_incl = incl;
                if (tid == kLazyThreads - 1) s_rem = incl;             // keys not yet consumed (buckets >= b_next)
                const unsigned long long nz = __ballot(hv != 0u);
                const unsigned long long fit = __ballot(hv != 0u && incl <= (unsigned)CH);
                if (lane == 0) {
                    s_queue_u32(s_raw)[wv] = nz ? (unsigned)(wv * 64 + __ffsll((long long)nz) - 1) : 256u;          // first non-empty
                    s_queue_u32(s_raw)[4 + wv] = fit ? (unsigned)(wv * 64 + 63 - __clzll((long long)fit)) : 0xffffffffu;   // last fitting
                }
                __syncthreads();
                if (tid == 0) {
                    unsigned first = 256u;
                    int last = -1;
                    for (int w = 3; w >= 0; --w) {
                        if (s_queue_u32(s_raw)[w] != 256u) first = s_queue_u32(s_raw)[w];
                    }
                    for (int w = 0; w < 4; ++w)
                        if (s_queue_u32(s_raw)[4 + w] != 0xffffffffu) last = (int)s_queue_u32(s_raw)[4 + w];
                    s_m = first;
                    s_b1 = (first >= 256u) ? 256 : last;   // last < first (i.e. -1): first bucket alone exceeds the chunk
                }
            }
            __syncthreads();
            const int b0 = (int)s_m, b1 = s_b1;
            __syncthreads();
            PH_LAP(3);
            if (b1 == 256) break;                 // nothing left
            if (b1 < 0) { bail = true; break; }   // one bucket larger than the chunk: full path
            // ---- a SECOND round starts (the first chunk did not saturate the tile, which is the exception):
            //      lay the remaining keys out by bucket, once, so that every later round reads only its own
            //      chunk instead of scanning the whole list again (n^2 / 512 key reads on a long translucent list;
            //      only worth it when many rounds are still to come: kPartitionMin).
            //      Bucket t of the remainder starts at the exclusive count of buckets b_next .. t - 1: the scan above.
            if ((!QUAD || SAS_TUNE_QPART) && !partitioned && b_next > 0 && s_rem > (unsigned)kPartitionMin) {
                s_cur[tid] = my_incl - my_hv;
                __syncthreads();
                partition_by_bucket(g, n, dmin, shift, b_next, s_cur, ids);
                partitioned = true;
                PH_LAP(4);
            }
            // ---- collect the range into LDS grouped by bucket (bucket t starts at the exclusive count of
            //      the buckets before it), depth words relative to the range's base
            const unsigned base = dmin + ((unsigned)b0 << shift);
            const bool mine = tid >= b0 && tid <= b1;
            if (tid == b1) s_m = my_incl;                            // entries in the chunk
            if (!partitioned) {
                if (mine) s_cur[tid] = my_incl - my_hv;
            } else if (mine) {
                s_cur[tid] = my_incl;   // END of bucket t inside the chunk (what the collect pass leaves): b0 is the first non-empty bucket of the scan
            }
            const bool big = wg_reduce_wave_flags<false>(__any(mine && my_hv > (unsigned)kRankMax), sync_phase);
            const unsigned long long t_c = PH_T();
            if (!partitioned && cached && b_next == 0) {
                // first round of a cached list: the keys are in LDS
                for (int i = tid; i < n; i += kLazyThreads) {
                    const unsigned long long key = kc[i];
                    const int b = (int)((hi32(key) - dmin) >> shift);
                    if (b >= b0 && b <= b1 && SAS_IN(b, 256, 206)) {
                        const unsigned pos = atomicAdd(&s_cur[b], 1u);
                        if (SAS_IN(pos, CH, 207)) ck[pos] = ((unsigned long long)(hi32(key) - base) << 32) | lo32(key);
                    }
                }
            } else if (!partitioned) {
                const unsigned long long *g3 = g;
                asm volatile("" : "+s"(g3));
                for (int i0 = 0; i0 < n; i0 += kLazyThreads * U) {
                    unsigned long long kk[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = i0 + u * kLazyThreads + tid;
                        kk[u] = (i < n) ? g3[i] : ~0ull;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (i0 + u * kLazyThreads + tid < n) {
                            const int b = (int)((hi32(kk[u]) - dmin) >> shift);
                            if (b >= b0 && b <= b1 && SAS_IN(b, 256, 206)) {
                                const unsigned pos = atomicAdd(&s_cur[b], 1u);
                                if (SAS_IN(pos, CH, 207)) ck[pos] = ((unsigned long long)(hi32(kk[u]) - base) << 32) | lo32(kk[u]);
                            }
                        }
                    }
                }
            } else {
                // the chunk is one contiguous run of the bucket-ordered slots; depth words come from the projection's records
                // buckets are consumed whole and in order, so the chunk starts where the previous ones ended
                const int st = p_consumed, cnt_chunk = (int)s_m;
                for (int i = tid; i < cnt_chunk; i += kLazyThreads) {
                    if (!SAS_IN(st + i, n, 216) || !SAS_IN(i, CH, 217)) continue;
                    const unsigned slot = (unsigned)ids[st + i];
                    const unsigned dbits = SAS_IN(slot, n_gauss, 218) ? __float_as_uint(f.rec[SAS_RS * (long long)slot + 1].w) : dmin;
                    ck[i] = ((unsigned long long)(dbits - base) << 32) | slot;
                }
            }
            __syncthreads();
            PH_LAP(5);
            if (!QUAD) { PH_ADD(2, PH_T() - t_c); PH_ADD(3, PH_T() - t_c); }
            const int m = (int)s_m;
            // ---- order the chunk, then composite it
            const unsigned long long t_s = PH_T();
            if (!(ablate & 1)) {
                if (!big) {
                    lds_bucket_rank_sort<CH / kLazyThreads>(ck, m, b0, shift, s_hist, s_cur, perm);
                } else {   // a crowded bucket (coplanar splats): radix passes cost the same whatever the distribution
                    const unsigned long long hi_excl = ((unsigned long long)(b1 - b0 + 1) << shift);
                    const unsigned rel_span = (unsigned)min((unsigned long long)(span - ((unsigned)b0 << shift)), hi_excl - 1ull);
                    lds_radix_sort<4, CH / kLazyThreads>(ck, m, rel_span, perm, cnt, dbase, s_wsum);
                }
            }
            PH_ADD(0, PH_T() - t_s);
            PH_LAP(6);
            bool all_done = true;   // ablation build (SAS_TUNE_ABLATE): pretend the first chunk saturates
            if (!(ablate & 2))
                all_done = blend(m, [&](int i) { return (long long)lo32(ck[i]); });
            if (all_done) break;
            if (partitioned) p_consumed += m;
            b_next = b1 + 1;
            if (b_next > 255) break;
        }
        if (bail) {
            // More than CH entries in one depth bucket (e.g. thousands of coplanar splats): order
            // the whole segment in place (slow, rare) and composite it from scratch.
            if (tid == 0) __hip_atomic_fetch_add(&f.stats_host[6], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            int *out = f.sorted_ids + beg;
            sort_global_bitonic(f.keys + beg, out, n, perm, tid, kLazyThreads);
            __syncthreads();
            p = pix_init(inside, ox);
            wdone = __all(!inside);
            blend(n, [&](int i) { return (long long)(unsigned)out[i]; });
        }
    }
    unsigned packed;
    float ED;
    if constexpr (QUAD) ED = write_pixel(o, p, inside && writer, ix, iy, c.W, packed);
    else ED = write_pixel<true>(o, p, inside, ix, iy, c.W, packed, o.rgb8 && (c.W & 3) == 0 && ((size_t)o.rgb8 & 3) == 0);
    if (o.rgb8_host) {   // (uniform: a frame property; out_side is uniform over the workgroup)
        if (out_side == 16) store_rows_to_host<16>(o.rgb8_host, c.W, tx * SAS_TILE, ty * SAS_TILE, ox, oy, writer, packed, s_raw);
        else if (out_side == 8) store_rows_to_host<8>(o.rgb8_host, c.W, tx * SAS_TILE + (qd & 1) * 8, ty * SAS_TILE + (qd >> 1) * 8,
                                                      ox & 7, oy & 7, writer, packed, s_raw);
    }
    if (WANT_MAX) store_tile_max(f, tile, ED, s_wmax);
    if (tid == 0) { f.tile_count[tile] = 0; f.tile_big[tile] = 0; }   // the frame's counters leave the frame zeroed (SasFrame invariant)
    PH_LAP(12);
#ifdef SAS_TUNE_WGTIME
    if (tid == 0 && blockIdx.x < kDbgWgMax) {
        g_dbg_wg[3 * blockIdx.x] = t_wg0;
        g_dbg_wg[3 * blockIdx.x + 1] = wall_clock64();
        g_dbg_wg[3 * blockIdx.x + 2] = (unsigned long long)n;
    }
#endif
}

#define SAS_LAZY_ATTRS __attribute__((amdgpu_flat_work_group_size(kLazyThreads, kLazyThreads), amdgpu_waves_per_eu(SAS_TUNE_OCC, SAS_TUNE_OCC)))
template <bool FAST_EXP, bool WANT_MAX, bool QUAD>
__global__ SAS_LAZY_ATTRS void k_tile_lazy(SasParams P, SasFrame f, long long n_gauss, const int *perm)
{
    tile_lazy_body<FAST_EXP, WANT_MAX, QUAD>(P, f, n_gauss, perm, blockIdx.x);
}
// All views of a group in one launch, INTERLEAVED in dispatch order (view = blockIdx.x mod nv): workgroups are
// handed out in blockIdx order and every view's tile list starts with its longest tiles, so with one view per
// blockIdx.y the later views' heavy tiles would start only once the first view had been dispatched entirely.
template <bool FAST_EXP, bool WANT_MAX, bool QUAD>
__global__ SAS_LAZY_ATTRS void k_tile_lazy_multi(SasMulti mf, long long n_gauss, const int *perm)
{
    const unsigned nv = (unsigned)mf.nv, v = blockIdx.x % nv;
    tile_lazy_body<FAST_EXP, WANT_MAX, QUAD>(mf.P[v], mf.f[v], n_gauss, perm, blockIdx.x / nv);
}

// Depth tail, one pass over the depth image after the tile kernel.
// FILL: depth = where(alpha > 0, ED, max ED)  (T0).  alpha == 0 <=> nothing blended <=> ED == 0.
//   Every workgroup first reduces the per-tile maxima (a few KB, L2-resident): no extra launch,
//   no same-address atomics.  ED >= 0, so the float order is the order of the bit patterns.
// PTS: the RGB-D consumer of nerfstudio_utils.py:424-445 fused into the same pass:
//   x = (u - cx) * d / fx, y = (v - cy) * d / fy, z = d;  mask = d < max_depth.
template <bool FILL, bool PTS>
__global__ __launch_bounds__(256) void k_depth_tail(const unsigned *tile_max, int tiles, SasParams P)
{
    __shared__ unsigned s_max[4];
    const SasOutputs &o = P.out;
    float *depth = o.depth;
    const int W = P.cam.W;
    const long long npix = (long long)W * P.cam.H;
    float mx = 0.0f;
    if (FILL) {
        unsigned m = 0;
        for (int t = threadIdx.x; t < tiles; t += 256) m = max(m, tile_max[t]);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, d));
        if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
        __syncthreads();
        mx = __uint_as_float(max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])));
    }
    const float fx = P.cam.fx, fy = P.cam.fy, cx = P.cam.cx, cy = P.cam.cy;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        float d = depth[p];
        if (FILL && d == 0.0f) {
            d = mx;
            depth[p] = d;
        }
        if (PTS) {
            const int v = (int)(p / W), u = (int)(p - (long long)v * W);
            if (o.points) {
                float *q = o.points + 3 * p;
                q[0] = ((float)u - cx) * d / fx;
                q[1] = ((float)v - cy) * d / fy;
                q[2] = d;
            }
            if (o.mask) o.mask[p] = o.use_max_depth ? (d < o.max_depth ? 1 : 0) : 1;
        }
    }
}

}  // namespace

SAS_BOUNDS_ACCESSOR(sas_debug_bounds_tiles)
#ifdef SAS_DEBUG_BOUNDS
// the checker checked: one guarded access that IS out of range (index 5 of 4) must be counted and skipped
namespace {
__global__ void k_bounds_selftest(int *hit)
{
    if (SAS_IN(5, 4, 999)) *hit = 1;
}
}  // namespace
extern "C" int sas_debug_bounds_selftest(void)
{
    int *d = nullptr, h = 0;
    if (hipMalloc(&d, sizeof(int)) != hipSuccess) return -1;
    (void)hipMemset(d, 0, sizeof(int));
    hipLaunchKernelGGL(k_bounds_selftest, dim3(1), dim3(1), 0, nullptr, d);
    const bool ok = hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    return ok ? h : -1;   // 0: the access was skipped
}
#endif

// ---- launchers -------------------------------------------------------------------------------------
constexpr int kSortMid = 4096, kSortLarge = 16384;

// Full ordering of every tile.  class 0: >= 4096 (LDS up to 16384, longer lists in place),
// class 1: 1024..4095, class 2: < 1024.  The classes are independent and each alone leaves most of
// the chip idle (few long lists), so they run concurrently: classes 0 and 1 on two side streams
// forked from / joined to `st`.
void sas_launch_sort(hipStream_t st, const SasScene &s, int ntiles, const SasFrame &f, const SasSortStreams &ss)
{
    const unsigned tiles = (unsigned)ntiles;
    hipStream_t s0 = st, s1 = st;
    if (ss.side[0]) {
        (void)hipEventRecord(ss.fork, st);
        (void)hipStreamWaitEvent(ss.side[0], ss.fork, 0);
        (void)hipStreamWaitEvent(ss.side[1], ss.fork, 0);
        s0 = ss.side[0];
        s1 = ss.side[1];
    }
    hipLaunchKernelGGL((k_sort_radix<kSortLarge, 1024, true>), dim3(tiles), dim3(1024), 0, s0, f, s.perm, f.tile_order,
                       f.sort_class + 0);
    hipLaunchKernelGGL((k_sort_radix<kSortMid, 256, false>), dim3(tiles), dim3(256), 0, s1, f, s.perm, f.tile_order,
                       f.sort_class + 1);
    hipLaunchKernelGGL(k_sort_wave, dim3(tiles), dim3(64), 0, st, f, s.perm, f.tile_order, f.sort_class + 2);
    if (ss.side[0]) {
        (void)hipEventRecord(ss.join[0], s0);
        (void)hipEventRecord(ss.join[1], s1);
        (void)hipStreamWaitEvent(st, ss.join[0], 0);
        (void)hipStreamWaitEvent(st, ss.join[1], 0);
    }
}

template <bool FAST, bool WMAX>
static void launch_blend_list(hipStream_t st, unsigned grid, const SasParams &P, const SasFrame &f, long long n,
                              const int *tl, const int *range)
{

    hipLaunchKernelGGL((k_blend<FAST, WMAX>), dim3(grid), dim3(256), 0, st, P, f, n, tl, range);
}

static void blend_list(hipStream_t st, unsigned grid, const SasParams &P, const SasFrame &f, long long n, const int *tl,
                       const int *range, bool fast_exp, bool want_max)
{
    if (fast_exp) {
        if (want_max) launch_blend_list<true, true>(st, grid, P, f, n, tl, range);
        else launch_blend_list<true, false>(st, grid, P, f, n, tl, range);
    } else {
        if (want_max) launch_blend_list<false, true>(st, grid, P, f, n, tl, range);
        else launch_blend_list<false, false>(st, grid, P, f, n, tl, range);
    }
}

// Full path, stage 2: composite all tiles from their complete lists (longest first).
void sas_launch_blend(hipStream_t st, const SasScene &s, int tiles, const SasParams &P, const SasFrame &f,
                      bool fast_exp, bool want_max)
{
    const long long n = s.n > 0 ? s.n : 1;
    blend_list(st, (unsigned)tiles, P, f, n, f.tile_order, f.sort_class + 4, fast_exp, want_max);
}

// Production path: lazy ordering + compositing of every tile in one launch.
// quad: frames of a few hundred tiles, binned in 8-pixel tiles by their projection (`tiles` counts those): one
// workgroup per 8x8 quadrant (pixel_of_quad); exact exponential only (SAS_FAST_EXP frames take the ordinary layout).
// experiments: SAS_TILE_DYN_LDS = bytes of unused dynamic LDS per tile workgroup (the compiled kernel, fewer workgroups per CU:
// 6656 -> four, 14000 -> three; profiles/r05_ab_tile_workgroups_per_cu.txt)
static unsigned lazy_dyn_lds()
{
    static const unsigned v = [] { const char *e = getenv("SAS_TILE_DYN_LDS"); return e ? (unsigned)atoi(e) : 0u; }();
    return v;
}

template <bool FAST, bool WMAX, bool QUAD>
static void launch_lazy(hipStream_t st, unsigned grid, const SasParams &P, const SasFrame &f, long long n, const int *perm,
                        hipEvent_t e0, hipEvent_t e1)
{
    if (e0 && e1)
        hipExtLaunchKernelGGL((k_tile_lazy<FAST, WMAX, QUAD>), dim3(grid), dim3(kLazyThreads), lazy_dyn_lds(), st, e0, e1, 0, P, f, n, perm);
    else
        hipLaunchKernelGGL((k_tile_lazy<FAST, WMAX, QUAD>), dim3(grid), dim3(kLazyThreads), lazy_dyn_lds(), st, P, f, n, perm);
}

bool sas_tiles_lazy_quad_ok(bool fast_exp) { return !fast_exp; }

void sas_launch_tiles_lazy(hipStream_t st, const SasScene &s, int tiles, const SasParams &P, const SasFrame &f,
                           bool fast_exp, bool want_max, bool quad, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    const unsigned grid = ((unsigned)tiles + 3u) & ~3u;   // whole groups of four tiles (tile_lazy_body)
    const long long n = s.n > 0 ? s.n : 1;
    if (fast_exp) {
        if (want_max) launch_lazy<true, true, false>(st, grid, P, f, n, s.perm, ev_start, ev_stop);
        else launch_lazy<true, false, false>(st, grid, P, f, n, s.perm, ev_start, ev_stop);
    } else if (quad) {
        if (want_max) launch_lazy<false, true, true>(st, grid, P, f, n, s.perm, ev_start, ev_stop);
        else launch_lazy<false, false, true>(st, grid, P, f, n, s.perm, ev_start, ev_stop);
    } else {
        if (want_max) launch_lazy<false, true, false>(st, grid, P, f, n, s.perm, ev_start, ev_stop);
        else launch_lazy<false, false, false>(st, grid, P, f, n, s.perm, ev_start, ev_stop);
    }
}

template <bool FAST, bool WMAX, bool QUAD>
static void launch_lazy_multi(hipStream_t st, dim3 grid, const SasMulti &mf, long long n, const int *perm, hipEvent_t e0, hipEvent_t e1)
{
    if (e0 && e1)
        hipExtLaunchKernelGGL((k_tile_lazy_multi<FAST, WMAX, QUAD>), grid, dim3(kLazyThreads), lazy_dyn_lds(), st, e0, e1, 0, mf, n, perm);
    else
        hipLaunchKernelGGL((k_tile_lazy_multi<FAST, WMAX, QUAD>), grid, dim3(kLazyThreads), lazy_dyn_lds(), st, mf, n, perm);
}

void sas_launch_tiles_lazy_multi(hipStream_t st, const SasScene &s, int tiles, const SasMulti &mf, bool fast_exp, bool want_max,
                                 bool quad, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    const dim3 grid((((unsigned)tiles + 3u) & ~3u) * (unsigned)mf.nv);   // views interleaved (k_tile_lazy_multi); whole groups of four tiles per view
    const long long n = s.n > 0 ? s.n : 1;
    if (fast_exp) {
        if (want_max) launch_lazy_multi<true, true, false>(st, grid, mf, n, s.perm, ev_start, ev_stop);
        else launch_lazy_multi<true, false, false>(st, grid, mf, n, s.perm, ev_start, ev_stop);
    } else if (quad) {
        if (want_max) launch_lazy_multi<false, true, true>(st, grid, mf, n, s.perm, ev_start, ev_stop);
        else launch_lazy_multi<false, false, true>(st, grid, mf, n, s.perm, ev_start, ev_stop);
    } else {
        if (want_max) launch_lazy_multi<false, true, false>(st, grid, mf, n, s.perm, ev_start, ev_stop);
        else launch_lazy_multi<false, false, false>(st, grid, mf, n, s.perm, ev_start, ev_stop);
    }
}

void sas_launch_depth_tail(hipStream_t st, int tiles, const SasParams &P, const SasFrame &f, bool fill, bool points)
{
    const unsigned *tm = (const unsigned *)f.tile_max;
    if (fill && points) hipLaunchKernelGGL((k_depth_tail<true, true>), dim3(1024), dim3(256), 0, st, tm, tiles, P);
    else if (fill) hipLaunchKernelGGL((k_depth_tail<true, false>), dim3(1024), dim3(256), 0, st, tm, tiles, P);
    else if (points) hipLaunchKernelGGL((k_depth_tail<false, true>), dim3(1024), dim3(256), 0, st, tm, tiles, P);
}
