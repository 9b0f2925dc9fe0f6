// A6: 3-D mask merging -- seq_merge / hierarchical_merge / merge_3d_masks
// (fsr_vln/memory/hmsg/utils/graph_utils.py:620-679, 883-1038) and the small-cloud drop of
// graph.py:445-448.
//
// What runs where
//   device : per-cloud uniform grids, "fraction of X within r of Y" counting in float32
//            (find_overlapping_ratio_faiss), concatenation of merged components, segmented exact DBSCAN
//            (merge_point_clouds_list -> pcd_denoise_dbscan(eps 0.1, min 10)), AABBs.
//   host   : AABB-IoU pair filter, threshold + connected components (tens..thousands of clouds), list
//            bookkeeping.  The merge is a fold over frames (inherently sequential across steps); each
//            step is two device->host syncs.
//
// EXACT shortcuts make a 1000-frame fold tractable (the reference re-does this work every step):
//   (1) fixed points: a cloud whose DBSCAN kept every point -- or dropped only points that are not
//       eps-neighbours of a kept core point (no "contested" border point, see k_db_label) -- is returned
//       unchanged by pcd_denoise_dbscan, so unchanged singleton components are not re-clustered;
//   (2) sequential merge: two clouds that were both inputs of the previous step and both survived it
//       unchanged were separate components there, so their overlap is <= threshold again -- only pairs
//       involving a new or changed cloud are evaluated.  (Hierarchical merge changes the threshold per
//       level, so there overlap VALUES are cached per pair of unchanged clouds instead.)
//   (3) anchors: the core flags of a fixed single-cluster cloud are persisted next to its pool points; when it
//       takes part in a merge they are handed to the DBSCAN batch, which neither re-counts nor re-connects
//       those points (CloudOps::dbscan_keep_largest, core0);
//   (4) empty masks never pair and are dropped by the min-points filter at the end: they are left out.
//   HMSG_DEBUG_NOANCHOR=1 disables (3) (tests compare the two folds bit for bit).
#include "hmsg_cloudops.h"

#include <algorithm>
#include <cmath>
#include <numeric>
#include <chrono>
#include <unordered_map>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

namespace {

struct Cloud {
    long long off = 0;      // into the point pool
    int n = 0;
    double mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    bool fixed = false;     // pcd_denoise_dbscan(merge eps/min) is known to return it unchanged
    bool fresh = true;      // new or changed since it last went through merge_3d_masks
    bool raw = false;       // a frame mask that has not been through a DBSCAN yet (statistics only)
    bool anchor = false;    // fixed AND one single cluster, with its core flags persisted next to the pool points:
                            // usable as the pre-connected anchor of the next DBSCAN it takes part in
    unsigned long long uid = 0;
    unsigned id = 0;        // incremental fold: cloud id in the persistent index
    int cap = 0;            // incremental fold: room of the cloud's pool region (points)
    // overlap grid (cell side >= r): cellstart[ncell+1] (absolute positions) + cell-sorted float32 points.
    // Two grids at most: a BASE grid over the cloud's first nb points, laid out on the box the cloud had when it was built, and
    // a DELTA grid over the points behind them, laid out on the cloud's box of today.  A merged cloud whose first member came
    // through its DBSCAN whole (DbscanResult::first_kept) takes that member's base grid over -- a grid holds its own sorted
    // copy of the points, and "some point within r" / "how many points within r" do not care which of two grids answers --
    // so a fold step re-indexes the points a surface GAINED, not the surface (the floor was 9/10 of the index work).
    bool has_index = false;          // every point is in one of the two grids
    bool has_index_before = false;   // (statistics)
    int nb = 0;                      // points in the base grid (0: none)
    long long ix_cell = 0, ix_pt = 0;
    int gd[3] = {0, 0, 0};
    float bmn[3] = {0, 0, 0};        // float32 lower corner the base grid was laid out on
    bool has_delta = false;
    long long ix_cell2 = 0, ix_pt2 = 0;
    int gd2[3] = {0, 0, 0};
    float dmn[3] = {0, 0, 0};        // float32 lower corner the delta grid was laid out on
};

// Work lists: the per-cloud kernels below get ONE workgroup per chunk of a cloud (blk0 = first workgroup of the
// entry, a prefix sum the host fills in; a workgroup finds its entry by binary search).  A 2-D grid sized for the
// largest cloud of a launch would start ~10^5 workgroups per fold step that find nothing to do.
template <typename T>
__device__ __forceinline__ int find_entry(const T* __restrict__ e, int n, unsigned blk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((unsigned)e[mid].blk0 <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

struct OvGrid {             // device view of one cloud for the overlap kernels
    int blk0, next;         // first workgroup of this entry in k_ov_count / k_ov_fill (work list); query tables: blk0 = points in the base grid, next = the cloud's delta grid (-1: none)
    long long pt_off;       // f64 points in the pool: the points this grid indexes (building) / the cloud's first point (query tables)
    long long ix_pt;        // its cell-sorted float32 copy in ix_pts (cell starts are relative to it)
    long long ix_cell;
    int n, gx, gy, gz;      // n: points this grid indexes (building) / points of the whole cloud (query tables)
    float mnx, mny, mnz, mxx, mxy, mxz;   // float32 AABB of the whole cloud
    double ox, oy, oz, cell;
};

struct OvTask {             // count points of grid[x] within r of grid[y]
    int x, y;
    int dep_n;              // second-direction tasks: number of points of the pair's smaller cloud (see k_ov_query)
    int blk0;               // first workgroup of this task (work list)
};

// Round 6 -- the overlap grids of a fold step's OUTPUT clouds are built while the host still waits for that step's DBSCAN batch
// (Merger::speculate_indices).  The host cannot know yet how many points a merged cloud keeps, nor whether its first member came
// through whole (only then its base grid is taken over) -- the device can: one entry per output cloud, laid out by the host on the
// segment's INPUT box (every kept point lies in it) with room for all input points, and k_ov_spec fills in which points the grid
// indexes from the batch's result words.  ov_spec_decide is that decision, evaluated by the device for the build and by the host,
// once the results have arrived, for its bookkeeping: 0 nothing to build (a single cloud that came through unchanged keeps the
// grids it has; an empty cloud has none), 1 the first member's base grid covers the whole cloud, 2 that base grid + a delta grid
// over the points behind it, 3 a grid over the whole cloud.
struct OvSpec {
    OvGrid g;
    int seg, singleton, inherit, nb_first, n_first, n_in, out_mode, delta_always;
    long long out_off;      // mode 1 / 2: the cloud's first point in the pool; mode 0: first point of the dense outputs (its place among them is added)
    long long ncell;
};
__host__ __device__ inline int ov_spec_decide(const OvSpec& sp, int n_out, int first_kept, int* first, int* n_idx) {
    *first = 0;
    *n_idx = 0;
    if (n_out <= 0) return 0;
    if (sp.singleton && n_out == sp.n_in) return 0;
    const bool inherit = sp.inherit && first_kept >= 0 && first_kept == sp.n_first && sp.nb_first > 0;
    if (inherit && sp.nb_first == n_out) return 1;
    // (build_indices' rule: an inherited base grid stays while what the cloud has gained since is a quarter of it at most and the
    //  grid of the gain, laid out on the whole box, is not mostly cells)
    const bool delta = inherit && sp.nb_first < n_out &&
                       (sp.delta_always || ((long long)(n_out - sp.nb_first) * 4 <= (long long)sp.nb_first && sp.ncell <= 4ll * sp.nb_first));
    *first = delta ? sp.nb_first : 0;
    *n_idx = n_out - *first;
    return delta ? 2 : 3;
}
// the batch's result words -> (points kept, points kept of the first member): CloudOps::dbscan_keep_largest's own read-back formulas
__host__ __device__ inline void ov_spec_result(const OvSpec& sp, const unsigned* res, int K, int* n_out, int* first_kept, int* start) {
    const int oend = (int)res[sp.seg], ostart = (int)res[(size_t)K * 16 + 8 + sp.seg], ofirst = (int)res[(size_t)K * 17 + 8 + sp.seg];
    *start = ostart;
    *n_out = sp.out_mode == 2 ? sp.n_first + oend : oend - ostart;
    *first_kept = sp.n_first <= 0 ? -1 : (sp.out_mode == 2 ? sp.n_first : (sp.n_first < sp.n_in ? ofirst - ostart : *n_out));
}
__global__ void k_ov_spec(const OvSpec* __restrict__ spec /* pinned host memory */, int nspec, const unsigned* __restrict__ res, int K,
                          OvGrid* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nspec) return;
    const OvSpec sp = spec[j];
    int n_out, first_kept, start, first, n_idx;
    ov_spec_result(sp, res, K, &n_out, &first_kept, &start);
    ov_spec_decide(sp, n_out, first_kept, &first, &n_idx);
    OvGrid g = sp.g;
    g.n = n_idx;
    g.pt_off = sp.out_off + (sp.out_mode == 0 ? (long long)start : 0ll) + first;
    out[j] = g;
}

}  // namespace

// ------------------------------------------------------------------------------------------ overlap index
__device__ __forceinline__ long long ov_cell(const OvGrid& g, float x, float y, float z) {
    int ix = (int)floor(((double)x - g.ox) / g.cell), iy = (int)floor(((double)y - g.oy) / g.cell),
        iz = (int)floor(((double)z - g.oz) / g.cell);
    ix = min(max(ix, 0), g.gx - 1);
    iy = min(max(iy, 0), g.gy - 1);
    iz = min(max(iz, 0), g.gz - 1);
    return g.ix_cell + ((long long)ix * g.gy + iy) * g.gz + iz;
}
// (counts go to `cursor`, indexed relative to the batch's first cell; the scan turns them into cell starts in
//  `cells`, and k_ov_fill hands a cell's slots out from its end by counting `cursor` back down -- the order of
//  points inside a cell is irrelevant)
static const int OVI_CHUNK = 512;      /* points per workgroup of the index kernels */
__global__ void k_ov_count(const double* __restrict__ pool, const OvGrid* __restrict__ gr, int ngr, unsigned* __restrict__ cursor,
                           long long cursor_base, int chunk) {
    const OvGrid g = gr[find_entry(gr, ngr, blockIdx.x)];
    const int i0 = (int)(blockIdx.x - (unsigned)g.blk0) * chunk, i1 = min(g.n, i0 + chunk);
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const double* p = pool + (size_t)(g.pt_off + i) * 3;
        atomicAdd(&cursor[ov_cell(g, (float)p[0], (float)p[1], (float)p[2]) - cursor_base], 1u);
    }
}
__global__ void k_ov_fill(const double* __restrict__ pool, const OvGrid* __restrict__ gr, int ngr, const unsigned* __restrict__ cells,
                          unsigned* __restrict__ cursor, long long cursor_base, float* __restrict__ sorted, int chunk) {
    const OvGrid g = gr[find_entry(gr, ngr, blockIdx.x)];
    const int i0 = (int)(blockIdx.x - (unsigned)g.blk0) * chunk, i1 = min(g.n, i0 + chunk);
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const double* p = pool + (size_t)(g.pt_off + i) * 3;
        float x = (float)p[0], y = (float)p[1], z = (float)p[2];
        long long c = ov_cell(g, x, y, z);
        size_t pos = (size_t)g.ix_pt + cells[c] + (atomicSub(&cursor[c - cursor_base], 1u) - 1u);
        sorted[(size_t)pos * 3] = x;
        sorted[(size_t)pos * 3 + 1] = y;
        sorted[(size_t)pos * 3 + 2] = z;
    }
}

// one z-run of Y's cell-sorted points: is any of them closer than r to (x, y, z)?  (float32 arithmetic of
// find_overlapping_ratio_faiss: (dx*dx + dy*dy) + dz*dz < r2)
#define OV_UNROLL 4     /* (16 was measured: 21 -> 46 us per launch -- short candidate lists dominate, and every step then issues 48 loads) */
// faiss's OTHER distance form (hmsg_config::overlap_distance_form = HMSG_OVERLAP_FAISS_BLAS): IndexFlatL2.search with 20 or more
// queries does not evaluate (dx*dx + dy*dy) + dz*dz per pair but |x|^2 + |y|^2 - 2 x.y with the inner products from sgemm, clamped at
// zero (faiss/utils/distances.cpp exhaustive_L2sqr_blas).  Stated order here and in oracle/hmsg_oracle.py (the BLAS kernel's own
// order is not ours to know): |p|^2 = (p0*p0 + p1*p1) + p2*p2, x.y = fma(x2, y2, fma(x1, y1, x0*y0)), dis = (|x|^2 + |y|^2) - 2*(x.y),
// every operation rounded to float32.
__device__ __forceinline__ float ov_norm2(float a, float b, float c) { return __fadd_rn(__fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b)), __fmul_rn(c, c)); }
template <bool BLAS>
__device__ __forceinline__ float ov_dist2(float x, float y, float z, float qx, float qy, float qz, float nx) {
    if (!BLAS) {
        const float ddx = __fsub_rn(x, qx), ddy = __fsub_rn(y, qy), ddz = __fsub_rn(z, qz);
        return __fadd_rn(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)), __fmul_rn(ddz, ddz));
    }
    const float ip = __fmaf_rn(z, qz, __fmaf_rn(y, qy, __fmul_rn(x, qx)));
    const float d = __fsub_rn(__fadd_rn(nx, ov_norm2(qx, qy, qz)), __fmul_rn(2.0f, ip));
    return d < 0.f ? 0.f : d;
}
// (BLAS is a template parameter down to the kernel: as a run-time flag in this innermost loop it cost the DEFAULT form 27.6 -> 32.0 us
//  per launch on the MI355X, profiles/r05_ov_experiments.txt)
template <bool BLAS = false>
__device__ __forceinline__ bool ov_scan(const float* __restrict__ sorted, unsigned s0, unsigned e0, float x, float y, float z,
                                        float r2, unsigned* ncand = nullptr, float nx = 0.f) {
    // OV_UNROLL candidates per step with independent loads: the scan is a serial latency chain per lane (one L2 round trip per
    // step: the early exit keeps the next step's loads from being issued ahead), and the kernel lasts as long as its slowest lane --
    // a point whose witness sits deep in a cell that has piled up hundreds of re-observations
    // Round 5, measured and withdrawn (profiles/r05_ov_probe_stats.txt, r05_ov_adaptive_unroll.txt, r05_ov_wave_cooperative.txt): a fold
    // step makes 1.5 * 10^5 probes that test 1.4 candidates each on average, but 291 of them walk more than 32 candidates (160 on
    // average, 581 at most: a point next to, but not within r of, a cell where a floor has piled up its re-observations).  Walking
    // what lies beyond the first eight candidates sixteen at a time: 25.6 -> 39.4 us per launch (the second loop's registers cost
    // every lane its occupancy); handing such probes to the whole wave, 64 candidates per trip: 52.5 us (they come in clusters -- a
    // mask's points share their neighbourhood --, a wave then walks its 30 heavy probes one after the other).
    for (unsigned k = s0; k < e0; k += OV_UNROLL) {
        bool h = false;
#pragma unroll
        for (int j = 0; j < OV_UNROLL; ++j) {
            const unsigned kk = min(k + (unsigned)j, e0 - 1u);
            const float d2 = ov_dist2<BLAS>(x, y, z, sorted[(size_t)kk * 3], sorted[(size_t)kk * 3 + 1], sorted[(size_t)kk * 3 + 2], nx);
            h = h || (k + (unsigned)j < e0 && d2 < r2);
        }
        if (ncand) *ncand += min(OV_UNROLL, (int)(e0 - k));
        if (h) return true;
    }
    return false;
}

// does (x, y, z) have a point of Y closer than r?  The point's own cell is scanned first: tested pairs passed
// the box filter, most points DO overlap, and on a surface that has piled up hundreds of re-observations per
// cell the witness sits in the own cell -- scanning the 27 cells in grid order found it after thousands of
// far candidates.
struct OvProbe {            // where a point falls in one grid
    int cx, cy, cz, z0, z1;
    bool own, any;
};
__device__ __forceinline__ OvProbe ov_probe(const OvGrid& Y, float x, float y, float z) {
    OvProbe p;
    p.cx = (int)floor(((double)x - Y.ox) / Y.cell);
    p.cy = (int)floor(((double)y - Y.oy) / Y.cell);
    p.cz = (int)floor(((double)z - Y.oz) / Y.cell);
    p.z0 = max(p.cz - 1, 0);
    p.z1 = min(p.cz + 1, Y.gz - 1);
    p.any = p.z1 >= p.z0;
    p.own = p.cx >= 0 && p.cx < Y.gx && p.cy >= 0 && p.cy < Y.gy && p.cz >= 0 && p.cz < Y.gz;
    return p;
}
// candidate range of the point's own cell
__device__ __forceinline__ void ov_own_range(const OvGrid& Y, const OvProbe& p, const unsigned* __restrict__ cells, unsigned& s0, unsigned& e0) {
    s0 = e0 = 0u;
    if (p.any && p.own) {
        const long long c = Y.ix_cell + ((long long)p.cx * Y.gy + p.cy) * Y.gz + p.cz;
        s0 = cells[c];
        e0 = cells[c + 1];
    }
}
// a MISS has to rule out all 27 cells: the candidate ranges of the 9 columns around the point (z-cells are contiguous; the own
// cell, already scanned, is cut out of its column: below it in [4], above it in [9]) -- independent loads, one round trip
__device__ __forceinline__ void ov_col_ranges(const OvGrid& Y, const OvProbe& p, const unsigned* __restrict__ cells, unsigned* rs, unsigned* re) {
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int jx = p.cx + q / 3 - 1, jy = p.cy + q % 3 - 1;
        rs[q] = re[q] = 0u;
        if (!p.any || jx < 0 || jx >= Y.gx || jy < 0 || jy >= Y.gy) continue;
        const long long c0 = Y.ix_cell + ((long long)jx * Y.gy + jy) * Y.gz;
        rs[q] = cells[c0 + p.z0];
        re[q] = (p.own && q == 4) ? cells[c0 + p.cz] : cells[c0 + p.z1 + 1];
    }
    rs[9] = re[9] = 0u;
    if (p.any && p.own) {
        const long long c0 = Y.ix_cell + ((long long)p.cx * Y.gy + p.cy) * Y.gz;
        rs[9] = cells[c0 + p.cz + 1];
        re[9] = cells[c0 + p.z1 + 1];
    }
}
// (st: HMSG_DEBUG_MERGESTATS counters -- [0] probes, [1] outside Y's box, [2] hits in the own cell, [3] hits in a neighbour cell,
//  [4] misses, [5] candidates tested in own cells, [6] candidates tested in neighbour cells; nullptr in every other run)
__device__ __forceinline__ void ov_stat(unsigned long long* st, int k, unsigned v = 1u) {
    if (st) atomicAdd(&st[k], (unsigned long long)v);
}
// (STATS, like BLAS, is a template parameter: the statistics' candidate counter has its address taken, and with a run-time `st` the
//  default kernel carried 8 bytes of scratch per lane for it)
template <bool BLAS = false, bool STATS = false>
__device__ __forceinline__ bool ov_hit(const OvGrid& Y, const unsigned* __restrict__ cells, const float* __restrict__ sorted,
                                       float x, float y, float z, float r2, float r, unsigned long long* st_in = nullptr) {
    unsigned long long* const st = STATS ? st_in : nullptr;
    const float nx = BLAS ? ov_norm2(x, y, z) : 0.f;
    ov_stat(st, 0);
    if (x < Y.mnx - r || x > Y.mxx + r || y < Y.mny - r || y > Y.mxy + r || z < Y.mnz - r || z > Y.mxz + r) {
        ov_stat(st, 1);
        return false;
    }
    const OvProbe p = ov_probe(Y, x, y, z);
    if (!p.any) return false;
    const float* const sy = sorted + (size_t)Y.ix_pt * 3;
    unsigned s0, e0, nc = 0u;
    ov_own_range(Y, p, cells, s0, e0);
    if (ov_scan<BLAS>(sy, s0, e0, x, y, z, r2, STATS ? &nc : nullptr, nx)) {
        ov_stat(st, 2);
        ov_stat(st, 5, nc);
        if (st && nc > 32u) atomicAdd(&st[8], 1ull), atomicAdd(&st[9], (unsigned long long)nc);
        if (st) atomicMax(&st[7], (unsigned long long)nc);
        return true;
    }
    ov_stat(st, 5, nc);
    const unsigned nc_own = nc;
    nc = 0u;
    unsigned rs[10], re[10];
    ov_col_ranges(Y, p, cells, rs, re);
#pragma unroll
    for (int q = 0; q < 10; ++q)
        if (ov_scan<BLAS>(sy, rs[q], re[q], x, y, z, r2, STATS ? &nc : nullptr, nx)) {
            ov_stat(st, 3);
            ov_stat(st, 6, nc);
            return true;
        }
    ov_stat(st, 4);
    ov_stat(st, 6, nc);
    if (st && nc + nc_own > 32u) atomicAdd(&st[8], 1ull), atomicAdd(&st[9], (unsigned long long)(nc + nc_own));
    if (st) atomicMax(&st[7], (unsigned long long)(nc + nc_own));
    return false;
}
// the same against a cloud with two grids (base + delta, hmsg_merge.hip: Cloud::nb): the two grids' table look-ups go out
// side by side -- every round of look-ups is a round trip, and the kernel lasts as long as a lane's chain of them
template <bool BLAS = false, bool STATS = false>
__device__ __forceinline__ bool ov_hit2(const OvGrid& Y, const OvGrid& Y2, const unsigned* __restrict__ cells, const float* __restrict__ sorted,
                                        float x, float y, float z, float r2, float r, unsigned long long* st = nullptr) {
    if (STATS) {                                   // (statistics runs: the two grids one after the other)
        if (ov_hit<BLAS, STATS>(Y, cells, sorted, x, y, z, r2, r, st)) return true;
        return ov_hit<BLAS, STATS>(Y2, cells, sorted, x, y, z, r2, r, st);
    }
    const float nx = BLAS ? ov_norm2(x, y, z) : 0.f;
    if (x < Y.mnx - r || x > Y.mxx + r || y < Y.mny - r || y > Y.mxy + r || z < Y.mnz - r || z > Y.mxz + r) return false;   // (both carry the cloud's box)
    const OvProbe p = ov_probe(Y, x, y, z), p2 = ov_probe(Y2, x, y, z);
    const float* const sy = sorted + (size_t)Y.ix_pt * 3;
    const float* const sy2 = sorted + (size_t)Y2.ix_pt * 3;
    unsigned s0, e0, s2, e2;
    ov_own_range(Y, p, cells, s0, e0);
    ov_own_range(Y2, p2, cells, s2, e2);
    if (ov_scan<BLAS>(sy, s0, e0, x, y, z, r2, nullptr, nx)) return true;
    if (ov_scan<BLAS>(sy2, s2, e2, x, y, z, r2, nullptr, nx)) return true;
    unsigned rs[10], re[10], rs2[10], re2[10];
    ov_col_ranges(Y, p, cells, rs, re);
    ov_col_ranges(Y2, p2, cells, rs2, re2);
#pragma unroll
    for (int q = 0; q < 10; ++q)
        if (ov_scan<BLAS>(sy, rs[q], re[q], x, y, z, r2, nullptr, nx)) return true;
#pragma unroll
    for (int q = 0; q < 10; ++q)
        if (ov_scan<BLAS>(sy2, rs2[q], re2[q], x, y, z, r2, nullptr, nx)) return true;
    return false;
}

static const int OV_CHUNK = 256;       /* points per workgroup of the overlap scans (128 .. 512 measured within 3 us of each other) */
// find_overlapping_ratio_faiss (graph_utils.py:645-662): a point of X overlaps when its exact float32
// nearest neighbour in Y is closer than r^2, i.e. when SOME y has (dx*dx + dy*dy) + dz*dz < r2 in float32.
// dep_counts (optional): the pair's first direction (the SMALLER cloud against the larger) has already been counted;
// when that ratio alone exceeds the threshold the pair merges whatever this direction gives -- max(a, b) > th --
// so the scan of the larger cloud is skipped (sequential merge: only the decision is needed, not the value).
// Round 5: BOTH directions of every pair in ONE launch (tasks [0, P): smaller -> larger, [P, 2P): larger -> smaller; blk_task maps a
// workgroup to its task -- one load instead of a binary search's chain of eight).  The second direction used to be a second,
// dependent launch that skipped the pairs the first ratio had decided; the kernel is a chain of dependent look-ups per lane and
// was expected to last ~25 us whether it covers one direction or both.  Measured on the MI355X (profiles/r05_ov_one_launch.txt): it
// does not -- 54.2 us for the one launch against 2 x 24.8 us: the scans the first ratio lets the second launch skip are real work
// (the kernel is bound by its L2 transactions, not by a lane's chain), so the DEPENDENT form stays the default (dep_counts !=
// nullptr, blk_off = first workgroup of the launch) and HMSG_OV_ONE_LAUNCH=1 selects the single launch.
// Decisions are the same either way: max(a, b) > th does not care about b once a > th.
// "More than th * |X| points of X (the larger cloud) have a point of Y within r" needs at least that many points of X inside Y's box
// grown by r, and X's own grid says how many it has there: the sum of its cell counts over that box (z-cells of a column are
// contiguous: two loads per column), plus the same over X's delta grid.  A floor against a chair-sized mask stays far below the
// threshold -- no scan of its 10^5 points then.  Evaluated by one workgroup (all 256 threads); true = the pair's second direction is
// decided (no merge by this direction) without a scan.
__device__ __forceinline__ bool ov_second_bounded(const OvGrid* __restrict__ gr, const OvGrid& X, const OvGrid& Y, const unsigned* __restrict__ cells,
                                                  float r, double th, unsigned* s_in /* [4] LDS */) {
    const float m = r + 1e-4f;
    int lo[3], hi[3];
    const double xo[3] = {X.ox, X.oy, X.oz};
    const float ymn[3] = {Y.mnx, Y.mny, Y.mnz}, ymx[3] = {Y.mxx, Y.mxy, Y.mxz};
    const int gd[3] = {X.gx, X.gy, X.gz};
    bool empty = false;
    for (int a = 0; a < 3; ++a) {
        lo[a] = (int)floor(((double)(ymn[a] - m) - xo[a]) / X.cell);
        hi[a] = (int)floor(((double)(ymx[a] + m) - xo[a]) / X.cell);
        lo[a] = max(lo[a], 0);
        hi[a] = min(hi[a], gd[a] - 1);
        empty = empty || hi[a] < lo[a];
    }
    unsigned inbox = 0;
    if (!empty) {
        const int ncol = (hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1), wy = hi[1] - lo[1] + 1;
        for (int q = (int)threadIdx.x; q < ncol; q += (int)blockDim.x) {
            const long long c0 = X.ix_cell + ((long long)(lo[0] + q / wy) * X.gy + (lo[1] + q % wy)) * X.gz;
            inbox += cells[c0 + hi[2] + 1] - cells[c0 + lo[2]];
        }
    }
    if (X.next >= 0) {                  // ... and the same sum over X's delta grid (the two grids split X's points)
        const OvGrid X2 = gr[X.next];
        const double xo2[3] = {X2.ox, X2.oy, X2.oz};
        const int gd2[3] = {X2.gx, X2.gy, X2.gz};
        bool empty2 = false;
        for (int a = 0; a < 3; ++a) {
            lo[a] = max((int)floor(((double)(ymn[a] - m) - xo2[a]) / X2.cell), 0);
            hi[a] = min((int)floor(((double)(ymx[a] + m) - xo2[a]) / X2.cell), gd2[a] - 1);
            empty2 = empty2 || hi[a] < lo[a];
        }
        if (!empty2) {
            const int ncol = (hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1), wy = hi[1] - lo[1] + 1;
            for (int q = (int)threadIdx.x; q < ncol; q += (int)blockDim.x) {
                const long long c0 = X2.ix_cell + ((long long)(lo[0] + q / wy) * X2.gy + (lo[1] + q % wy)) * X2.gz;
                inbox += cells[c0 + hi[2] + 1] - cells[c0 + lo[2]];
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) inbox += __shfl_xor(inbox, o);
    __syncthreads();                                        // (s_in may still be read from an earlier use)
    if ((threadIdx.x & 63) == 0) s_in[threadIdx.x >> 6] = inbox;
    __syncthreads();
    inbox = s_in[0] + s_in[1] + s_in[2] + s_in[3];
    return !((double)inbox / (double)X.n > th);
}
// points [b0, b1) of X against Y: how many have a point of Y within r (this thread's share; the caller reduces)
template <bool BLAS, bool STATS>
__device__ __forceinline__ unsigned ov_probe_chunk(const double* __restrict__ pool, const OvGrid* __restrict__ gr, const OvGrid& X, const OvGrid& Y,
                                                   int ix, int iy, const unsigned* __restrict__ cells, const float* __restrict__ sorted, float r2,
                                                   float r, int b0, int b1, int sorted_src, unsigned long long* __restrict__ st) {
    unsigned local = 0;
    const OvGrid Y2 = gr[Y.next >= 0 ? Y.next : iy];        // (Y's delta grid, if it has one)
    const bool blas = BLAS && X.n >= 20;                   // (faiss: the BLAS route from 20 queries on -- the cloud whose points are looked up)
    // Round 6: X's points are taken from X's OWN cell-sorted float32 copies (base grid, then delta grid: together they hold
    // every point of the cloud once, rounded exactly as `(float)p[a]` rounds it), not from the pool in pool order -- the count
    // does not care in which order the points are asked, and the lanes of a wave then stand in one or two cells of X, i.e. in
    // a handful of cells of Y; a point costs 12 bytes instead of 24 (profiles/r06_map_accum.txt: 4.7 -> 3.6 MB per launch, the
    // launch itself no faster).
    const OvGrid X2 = gr[X.next >= 0 ? X.next : ix];
    const int nbx = sorted_src ? X.blk0 : 0;               // (query tables: points in the base grid)
    const float* sx1 = nullptr;
    const float* sx2 = nullptr;
    if (sorted_src) {
        sx1 = sorted + ((size_t)X.ix_pt + cells[X.ix_cell]) * 3;
        if (X.next >= 0) sx2 = sorted + ((size_t)X2.ix_pt + cells[X2.ix_cell]) * 3 - (size_t)nbx * 3;
    }
    for (int i = b0 + (int)threadIdx.x; i < b1; i += blockDim.x) {
        float x, y, z;
        if (sorted_src) {
            const float* p = (i < nbx ? sx1 : sx2) + (size_t)i * 3;
            x = p[0], y = p[1], z = p[2];
        } else {
            const double* p = pool + (size_t)(X.pt_off + i) * 3;
            x = (float)p[0], y = (float)p[1], z = (float)p[2];
        }
        bool hit;
        if (BLAS && blas) hit = Y.next >= 0 ? ov_hit2<true, STATS>(Y, Y2, cells, sorted, x, y, z, r2, r, st) : ov_hit<true, STATS>(Y, cells, sorted, x, y, z, r2, r, st);
        else hit = Y.next >= 0 ? ov_hit2<false, STATS>(Y, Y2, cells, sorted, x, y, z, r2, r, st) : ov_hit<false, STATS>(Y, cells, sorted, x, y, z, r2, r, st);
        local += hit ? 1u : 0u;
    }
    return local;
}
// `bound_ahead` (round 6, with k_ov_query_second; first-direction launches of a sequential-merge step only): npairs extra workgroups
// evaluate the box bound of every pair's second direction and leave the verdict in counts[npairs + pair] (0x80000000 = decided), so
// that the second launch can be PLANNED from what lies in device memory (see there) instead of starting a workgroup per chunk of
// every larger cloud only to find out that nearly all of them have nothing to do.
template <bool BLAS, bool STATS>
__global__ void k_ov_query(const double* __restrict__ pool, const OvGrid* __restrict__ gr, const OvTask* __restrict__ tasks,
                           const unsigned* __restrict__ cells, const float* __restrict__ sorted, float r2, float r,
                           int npairs, const int* __restrict__ blk_task, unsigned blk_off, unsigned* __restrict__ counts,
                           const unsigned* __restrict__ dep_counts, double th, int chunk, unsigned long long* __restrict__ st,
                           int sorted_src, int bound_ahead) {
    __shared__ unsigned s_in[4];
    if (bound_ahead) {
        // the launch's first npairs workgroups only evaluate the box bound of their pair's second direction (X and Y change
        // places), beside the probes of the others -- as a prologue of a task's first probing workgroup it lengthened that
        // workgroup's chain, and the launch lasts as long as its longest chain
        if ((int)blockIdx.x < npairs) {
            const OvTask t0 = tasks[blockIdx.x];
            const OvGrid A = gr[t0.x], B = gr[t0.y];
            if (ov_second_bounded(gr, B, A, cells, r, th, s_in) && threadIdx.x == 0) counts[npairs + (int)blockIdx.x] = 0x80000000u;
            return;
        }
    }
    const unsigned blk = blockIdx.x - (bound_ahead ? (unsigned)npairs : 0u) + blk_off;
    const int ti = blk_task[blk];
    const OvTask t = tasks[ti];
    const bool second = ti >= npairs;
    bool run = !(second && dep_counts && (double)dep_counts[ti - npairs] / (double)t.dep_n > th);
    const OvGrid X = gr[t.x], Y = gr[t.y];
    if (run && second && th >= 0.0 && !bound_ahead) {
        // Second direction, first ratio <= th: the pair merges only if MORE than th * |X| points of X (the larger cloud) have
        // a point of Y (the smaller one) within r (ov_second_bounded).
        if (ov_second_bounded(gr, X, Y, cells, r, th, s_in)) {
            if (blk == (unsigned)t.blk0 && threadIdx.x == 0) counts[ti] = 0x80000000u;   // (decided by the bound: no scan)
            run = false;
        }
    }
    unsigned local = 0;
    // a workgroup takes OV_CHUNK consecutive points: every point is a serial chain of L2 round trips, so a
    // million-point X is spread over many workgroups (work list: exactly ceil(n / OV_CHUNK) of them per task)
    if (run) {
        const int b0 = (int)(blk - (unsigned)t.blk0) * chunk;
        const int b1 = b0 + chunk < X.n ? b0 + chunk : X.n;
        local = ov_probe_chunk<BLAS, STATS>(pool, gr, X, Y, t.x, t.y, cells, sorted, r2, r, b0, b1, sorted_src, st);
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&counts[ti], local);
}
// Round 6 -- the second direction of a sequential-merge step, planned on the device.  Until now the second launch started a
// workgroup per OV_CHUNK points of every pair's LARGER cloud (~10^4 workgroups a step), each of which loaded its task and both
// grids and summed the box bound, only for nearly all of them to find the pair decided (by the first ratio or by the bound: 22 000 of a
// step's ~10^6 larger-cloud points are ever scanned) -- the launch took 25 us for a quarter of the first launch's probes.  Now the
// verdicts lie in device memory when the first launch ends (first counts; the bound, evaluated once per pair by the first launch),
// and a FIXED grid plans its own work: every workgroup reads the <= OV2_MAX_PAIRS verdicts (a coalesced load or two), lays the chunks of the
// undecided pairs end to end in LDS (a block scan) and takes the chunks blockIdx.x, blockIdx.x + gridDim.x, ... of that list.  No
// host round trip (the grid does not depend on how many pairs are left), no workgroup per skipped chunk.  Same counts, same flags.
#define OV2_MAX_PAIRS 2048
template <bool BLAS, bool STATS>
__global__ void __launch_bounds__(256) k_ov_query_second(const double* __restrict__ pool, const OvGrid* __restrict__ gr, const OvTask* __restrict__ tasks,
                                                         const unsigned* __restrict__ cells, const float* __restrict__ sorted, float r2, float r,
                                                         int npairs, unsigned* __restrict__ counts, double th, int chunk,
                                                         unsigned long long* __restrict__ st, int sorted_src) {
    __shared__ unsigned s_off[OV2_MAX_PAIRS + 1];           // first chunk of every pair in the list of chunks to scan
    __shared__ unsigned s_w[4], s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_carry = 0u;
    __syncthreads();
    for (int k0 = 0; k0 < npairs; k0 += 256) {              // (block-uniform)
        const int k = k0 + tid;
        unsigned nch = 0;
        if (k < npairs) {
            const OvTask t = tasks[npairs + k];
            const bool first_decides = (double)counts[k] / (double)t.dep_n > th;
            const bool bounded = counts[npairs + k] == 0x80000000u;
            if (!first_decides && !bounded) nch = ((unsigned)gr[t.x].n + (unsigned)chunk - 1u) / (unsigned)chunk;
        }
        unsigned incl = nch;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        unsigned base = s_carry;
        for (int q = 0; q < wv; ++q) base += s_w[q];
        if (k < npairs) s_off[k] = base + incl - nch;
        __syncthreads();
        if (tid == 0) s_carry += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    if (tid == 0) s_off[npairs] = s_carry;
    __syncthreads();
    const unsigned total = s_off[npairs];
    for (unsigned j = blockIdx.x; j < total; j += gridDim.x) {      // (block-uniform)
        int lo = 0, hi = npairs - 1;                        // the pair of chunk j: the last one whose first chunk is <= j
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_off[mid] <= j) lo = mid; else hi = mid - 1;
        }
        const OvTask t = tasks[npairs + lo];
        const OvGrid X = gr[t.x], Y = gr[t.y];
        const int b0 = (int)(j - s_off[lo]) * chunk;
        const int b1 = b0 + chunk < X.n ? b0 + chunk : X.n;
        unsigned local = ov_probe_chunk<BLAS, STATS>(pool, gr, X, Y, t.x, t.y, cells, sorted, r2, r, b0, b1, sorted_src, st);
        for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
        if (lane == 0 && local) atomicAdd(&counts[npairs + lo], local);
    }
}

static const int CAT_CHUNK = 512;      /* points per workgroup of k_concat */
__global__ void k_concat(const double* __restrict__ pool, const CatSeg* __restrict__ segs, int nsegs, double* __restrict__ dst,
                         const unsigned char* __restrict__ poolcore, unsigned char* __restrict__ dstcore, int chunk) {
    const CatSeg sg = segs[find_entry(segs, nsegs, blockIdx.x)];
    const long long p0 = (long long)(blockIdx.x - (unsigned)sg.blk0) * chunk, p1 = min((long long)sg.n, p0 + chunk);
    for (long long i = p0 * 3 + threadIdx.x; i < p1 * 3; i += blockDim.x) dst[sg.dst * 3 + i] = pool[sg.src * 3 + i];
    if (dstcore)
        for (long long i = p0 + threadIdx.x; i < p1; i += blockDim.x)
            dstcore[sg.dst + i] = sg.anchor ? poolcore[sg.src + i] : (unsigned char)0;
}

// ------------------------------------------------------------------------------------------ host side
namespace {

double bbox_iou(const Cloud& a, const Cloud& b) {   // graph_utils.py:883-915; empty cloud -> (0,0,0) box
    // boxes disjoint along an axis: overlap volume 0 -> IoU 0 (or 0/0): never > iou_thresh (>= 0 by contract)
    if (a.mx[0] <= b.mn[0] || b.mx[0] <= a.mn[0] || a.mx[1] <= b.mn[1] || b.mx[1] <= a.mn[1] || a.mx[2] <= b.mn[2] ||
        b.mx[2] <= a.mn[2])
        return 0.0;
    double ov = 1, va = 1, vb = 1;
    for (int k = 0; k < 3; ++k) {
        double omin = std::max(a.mn[k], b.mn[k]), omax = std::min(a.mx[k], b.mx[k]);
        ov *= std::max(omax - omin, 0.0);
        va *= a.mx[k] - a.mn[k];
        vb *= b.mx[k] - b.mn[k];
    }
    return ov / (va + vb - ov);   // 0/0 -> NaN -> comparison false, like numpy
}

// cm[j] = box j meets the query box q = {lo0, hi0, lo1, hi1, lo2, hi2} with positive extent on every axis
// (thousands of boxes per fresh cloud per step: compiled for AVX2 when the host has it)
#define HMSG_BOX_MASK_BODY                                                                                          \
    for (int j = 0; j < n; ++j)                                                                                     \
        cm[j] = (unsigned char)!((q[1] <= lo0[j]) | (hi0[j] <= q[0]) | (q[3] <= lo1[j]) | (hi1[j] <= q[2]) |        \
                                 (q[5] <= lo2[j]) | (hi2[j] <= q[4]));
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
__attribute__((target("avx2"))) void box_mask_avx2(int n, const double* lo0, const double* hi0, const double* lo1,
                                                   const double* hi1, const double* lo2, const double* hi2, const double* q,
                                                   unsigned char* cm) {
    HMSG_BOX_MASK_BODY
}
#endif
void box_mask(int n, const double* lo0, const double* hi0, const double* lo1, const double* hi1, const double* lo2,
              const double* hi2, const double* q, unsigned char* cm) {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    if (have_avx2) {
        box_mask_avx2(n, lo0, hi0, lo1, hi1, lo2, hi2, q, cm);
        return;
    }
#endif
    HMSG_BOX_MASK_BODY
}

struct IntSpan {
    const int* b;
    const int* e;
    size_t size() const { return (size_t)(e - b); }
    int operator[](size_t k) const { return b[k]; }
    const int* begin() const { return b; }
    const int* end() const { return e; }
};
struct CompList {               // components as CSR: members of component c = mem[off[c] .. off[c+1])
    std::vector<int> off, mem;
    size_t size() const { return off.empty() ? 0 : off.size() - 1; }
    IntSpan operator[](size_t c) const { return IntSpan{mem.data() + off[c], mem.data() + off[c + 1]}; }
};

struct Merger {
    hmsg_ctx* h = nullptr;
    hipStream_t s = nullptr;
    CloudOps ops;
    DevBuf<double> pool;
    DevBuf<unsigned char> poolcore; // core flag of every pool point (valid for `anchor` clouds)
    long long pool_used = 0;        // points
    DevBuf<unsigned> ix_cells;      // concatenated cellstart arrays (absolute positions into ix_pts)
    long long ix_cells_used = 0;
    DevBuf<float> ix_pts;           // concatenated cell-sorted float32 points
    long long ix_pts_used = 0;      // points
    DevBuf<double> concat;
    DevBuf<unsigned char> concat_core;
    DevBuf<OvGrid> d_grids;
    PinnedBuf<OvGrid> h_grids;
    PinnedBuf<CatSeg> h_cat;
    DevBuf<unsigned> d_cursor;
    DevBuf<CatSeg> d_cat;
    DevBuf<unsigned long long> ovstat;  // HMSG_DEBUG_MERGESTATS: probe statistics of k_ov_query (ov_stat)
    Publisher pub_counts;           // overlap counts of a step, read back through pinned memory (hmsg_cloudops.h)
    // one packed upload per overlap step: [grids | tasks | zeroed counts], staged in pinned memory (the previous step's
    // copy has completed: every overlap step ends with a wait on the stream)
    PinnedBuf<char> h_ovpack;
    DevBuf<char> d_ovpack;
    // grids of a step's output clouds, built behind the DBSCAN batch's publish (speculate_indices)
    PinnedBuf<OvSpec> h_spec[2];               // (two sets, used in turn: a step without candidate pairs waits for nothing between two
    DevBuf<OvGrid> d_spec[2];                  //  batches, and the GPU may still be reading the previous step's table)
    int spec_turn = 0;
    std::vector<OvSpec> spec;                  // this step's entries (host copy: the bookkeeping replays the device's decisions)
    std::vector<int> spec_of_seg;
    bool spec_wanted = getenv("HMSG_DEBUG_NO_SPEC_INDEX") == nullptr;   // HMSG_DEBUG_NO_SPEC_INDEX=1: every grid is built at the start of the next step (round 5)
    double spec_built = 0, spec_built_pts = 0, spec_skipped = 0;        // (statistics)
    size_t cursor_clean = 0;        // entries of d_cursor known to be zero (k_ov_fill counts every cell back down to zero)
    SpinWait spin;
    unsigned long long next_uid = 1;
    std::unordered_map<unsigned long long, double> ratio_cache;   // hierarchical merge only
    bool use_cache = false;
    bool use_anchor = true;         // HMSG_DEBUG_NOANCHOR: plain DBSCAN of every batch (tests compare the two)
    bool inplace_wanted = getenv("HMSG_DEBUG_NO_INPLACE") == nullptr;   // HMSG_DEBUG_NO_INPLACE=1: every output is a dense copy (round 4)
    double radius = 0;              // 1.5 * voxel_size  (merge_3d_masks passes radius=1.5*radius)
    double reach = 0;               // the distance inside which a point can still count as "closer than radius": radius itself, or
                                    // sqrt(radius^2 + E) with E the rounding of faiss's BLAS form at this scene's coordinates
    int faiss_form = 0;             // hmsg_config::overlap_distance_form
    double cell = 0;
    double eps = 0.1;
    int minpts = 10;
    double iou_thresh = 0.05;
    double tphase[6] = {0, 0, 0, 0, 0, 0};   // host wall time per phase (HMSG_DEBUG_TIMING)
    // HMSG_DEBUG_MERGESTATS: what a fold step is made of (sums over the fold)
    struct Stats {
        double steps = 0, clouds = 0, fresh_raw = 0, fresh_g = 0, pairs_raw = 0, pairs_g = 0;
        double scan1_raw = 0, scan2_raw = 0, scan1_g = 0, scan2_g = 0;      // points scanned, by direction and pair class
        double idx_pts = 0;                                                 // points (re)indexed
        // components that go through DBSCAN, by class: 0 singleton raw, 1 singleton non-fixed, 2 anchor first + raw rest
        // (|A| > |B|), 3 anchor first + any rest (|A| > |B|), 4 other
        double ccount[5] = {0, 0, 0, 0, 0}, cpts[5] = {0, 0, 0, 0, 0}, cB[5] = {0, 0, 0, 0, 0}, cmem[5] = {0, 0, 0, 0, 0};
        double cchanged[5] = {0, 0, 0, 0, 0}, cmulti[5] = {0, 0, 0, 0, 0}, ccontested[5] = {0, 0, 0, 0, 0}, cnonfixed[5] = {0, 0, 0, 0, 0};
    } st;
    bool want_stats = false;

    template <typename T>
    void grow(DevBuf<T>& b, size_t used_elems, size_t need_elems) {
        if (need_elems <= b.n) return;
        CarveScope carve;           // (the merger's arenas die with it)
        DevBuf<T> nb;
        nb.alloc(std::max(need_elems * 2, (size_t)1 << 18));
        if (used_elems) HIP_TRY(hipMemcpyAsync(nb.p, b.p, used_elems * sizeof(T), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        nb.swap(b);
    }

    // float32 rounding can move a coordinate slightly outside the f64 box: the origin is padded
    OvGrid grid_at(const Cloud& c, const float* lo, const int* gd, long long ix_cell, long long ix_pt, long long pt_off, int n) const {
        OvGrid g;
        g.blk0 = 0;
        g.next = -1;
        g.pt_off = pt_off;
        g.ix_cell = ix_cell;
        g.ix_pt = ix_pt;
        g.n = n;
        g.gx = gd[0];
        g.gy = gd[1];
        g.gz = gd[2];
        g.mnx = (float)c.mn[0]; g.mny = (float)c.mn[1]; g.mnz = (float)c.mn[2];
        g.mxx = (float)c.mx[0]; g.mxy = (float)c.mx[1]; g.mxz = (float)c.mx[2];
        g.ox = (double)lo[0] - 1e-3;
        g.oy = (double)lo[1] - 1e-3;
        g.oz = (double)lo[2] - 1e-3;
        g.cell = cell;
        return g;
    }
    // query-table entries of a cloud: its base grid (carrying the whole cloud's point range) and, if it has one, its delta grid
    void push_grids(const Cloud& c, std::vector<OvGrid>& g) const {
        g.push_back(grid_at(c, c.bmn, c.gd, c.ix_cell, c.ix_pt, c.off, c.n));
        g.back().blk0 = c.nb;           // (query tables: the points of the base grid -- k_ov_query walks X's sorted copies)
        if (c.has_delta) {
            g.back().next = (int)g.size();
            g.push_back(grid_at(c, c.dmn, c.gd2, c.ix_cell2, c.ix_pt2, c.off + c.nb, c.n - c.nb));
        }
    }

    // ---- overlap grids for the clouds that do not have one yet
    void build_indices(std::vector<Cloud>& L) {
        std::vector<int> todo;
        long long ncell_new = 0, npts_new = 0;
        int maxn = 0;
        std::vector<OvGrid> g;
        unsigned nblk = 0;
        for (int i = 0; i < (int)L.size(); ++i) {
            Cloud& c = L[i];
            if (c.has_index || c.n == 0) continue;
            const float lo[3] = {(float)c.mn[0], (float)c.mn[1], (float)c.mn[2]};
            int gd[3];
            for (int a = 0; a < 3; ++a) gd[a] = (int)std::floor(((double)(float)c.mx[a] - (double)lo[a] + 2e-3) / cell) + 1;
            const long long ncell = (long long)gd[0] * gd[1] * gd[2] + 1;   // +1: end sentinel
            // an inherited base grid stays while what the cloud has gained since is a quarter of it at most (and the grid of the
            // gain, laid out on the whole box, is not mostly cells): else the cloud is indexed afresh
            const bool delta = c.nb > 0 && c.nb < c.n && (delta_always || ((long long)(c.n - c.nb) * 4 <= (long long)c.nb && ncell <= 4ll * c.nb));
            (delta ? grid_delta : grid_full) += 1;
            (delta ? grid_delta_pts : grid_full_pts) += delta ? c.n - c.nb : c.n;
            int first = 0;
            if (delta) {
                first = c.nb;
                c.has_delta = true;
                for (int a = 0; a < 3; ++a) {
                    c.gd2[a] = gd[a];
                    c.dmn[a] = lo[a];
                }
                c.ix_cell2 = ix_cells_used + ncell_new;
                c.ix_pt2 = ix_pts_used;                  // (cell starts are relative to the BATCH's first sorted point)
                g.push_back(grid_at(c, lo, c.gd2, c.ix_cell2, c.ix_pt2, c.off + first, c.n - first));
            } else {
                c.nb = c.n;
                c.has_delta = false;
                for (int a = 0; a < 3; ++a) {
                    c.gd[a] = gd[a];
                    c.bmn[a] = lo[a];
                }
                c.ix_cell = ix_cells_used + ncell_new;
                c.ix_pt = ix_pts_used;
                g.push_back(grid_at(c, lo, c.gd, c.ix_cell, c.ix_pt, c.off, c.n));
            }
            g.back().blk0 = (int)nblk;
            nblk += cdiv((size_t)(c.n - first), OVI_CHUNK);
            ncell_new += ncell;
            npts_new += c.n - first;
            maxn = std::max(maxn, c.n - first);
            todo.push_back(i);
        }
        if (todo.empty()) return;
        // (arena offsets are 64-bit; cell starts are relative to the grid's first sorted point, so only a BATCH is bounded)
        HMSG_REQUIRE(ncell_new < (1ll << 32) && npts_new < (1ll << 32), HMSG_ERR_UNSUPPORTED,
                     "one batch of overlap grids exceeds 2^32 entries");
        grow(ix_cells, (size_t)ix_cells_used, (size_t)(ix_cells_used + ncell_new));
        grow(ix_pts, (size_t)ix_pts_used * 3, (size_t)(ix_pts_used + npts_new) * 3);
        d_grids.ensure(g.size());
        // (pinned staging: an async copy from pageable memory is staged by the runtime, ~3x the host time per call;
        //  the previous use has completed -- every step and every prebuild ends with a wait on the stream)
        h_grids.ensure(g.size());
        memcpy(h_grids.p, g.data(), g.size() * sizeof(OvGrid));
        upload_pinned(d_grids.p, h_grids.p, g.size() * sizeof(OvGrid), s);
        unsigned* cells = ix_cells.p + ix_cells_used;
        {   // the cursor is zero whenever k_ov_fill has run over what k_ov_count counted: only fresh memory is cleared
            const unsigned* before = d_cursor.p;
            d_cursor.ensure((size_t)ncell_new);
            if (d_cursor.p != before) cursor_clean = 0;
            if (cursor_clean < (size_t)ncell_new) {
                HIP_TRY(hipMemsetAsync(d_cursor.p, 0, d_cursor.n * 4, s));
                cursor_clean = d_cursor.n;
            }
        }
        hipLaunchKernelGGL(k_ov_count, dim3(nblk), dim3(256), 0, s, (const double*)pool.p, (const OvGrid*)d_grids.p, (int)g.size(),
                           d_cursor.p, ix_cells_used, OVI_CHUNK);
        HMSG_CHECK_LAUNCH();
        // (cell starts stay relative to this batch's first sorted point: Cloud::ix_pt)
        hmsg_scan_u32(d_cursor.p, cells, (size_t)ncell_new, s, ops.scan_tmp, nullptr);
        hipLaunchKernelGGL(k_ov_fill, dim3(nblk), dim3(256), 0, s, (const double*)pool.p, (const OvGrid*)d_grids.p, (int)g.size(),
                           (const unsigned*)ix_cells.p, d_cursor.p, ix_cells_used, ix_pts.p, OVI_CHUNK);
        HMSG_CHECK_LAUNCH();
        for (int i : todo) L[i].has_index = true;
        ix_cells_used += ncell_new;
        ix_pts_used += npts_new;
    }

    // ---- overlap grids of the frame masks of frames [a, b) in one batch (inputs of the fold: built ahead, a window at a
    // time, so that a very long episode neither exceeds a batch nor holds grids of frames that are hours away)
    void prebuild(std::vector<std::vector<Cloud>>& frames, size_t a, size_t b) {
        b = std::min(b, frames.size());
        std::vector<Cloud> all;
        for (size_t f = a; f < b; ++f) all.insert(all.end(), frames[f].begin(), frames[f].end());
        build_indices(all);
        spin.wait(s);                       // (h_grids is re-used by the next call)
        size_t k = 0;
        for (size_t f = a; f < b; ++f)
            for (auto& cl : frames[f]) cl = all[k++];
    }

    // ---- The point pool and the grid arenas are append-only: every fold step adds its merged clouds and their grids,
    // and what they replace stays behind.  Fine for 10^3 frames (a few GB); a 10^4-frame episode would leave ~70 GB of
    // dead points and > 2^32 dead cells.  collect() moves the LIVE clouds (the global list + the frames still to come)
    // to a fresh pool and drops every grid; grids come back through prebuild / build_indices.  Results do not change.
    bool needs_collect() const {
        return pool_used > (long long)gc_pool_points || ix_cells_used > (long long)gc_index_entries || ix_pts_used > (long long)gc_index_entries;
    }
    void collect(std::vector<Cloud>& G, std::vector<std::vector<Cloud>>& frames, size_t f_next) {
        long long live = 0;
        unsigned blocks = 0;
        std::vector<CatSeg> cat;
        auto add = [&](Cloud& c) {
            cat.push_back(CatSeg{c.off, live, c.n, c.anchor ? 1 : 0, (int)blocks, 0});
            blocks += cdiv((size_t)c.n, CAT_CHUNK);
            c.off = live;
            c.cap = c.n;                 // (packed: no room behind it any more)
            live += c.n;
            c.has_index = false;
            c.has_delta = false;
            c.nb = 0;
        };
        for (auto& c : G) add(c);
        for (size_t f = f_next; f < frames.size(); ++f)
            for (auto& c : frames[f]) add(c);
        // the fresh pool has room for everything up to the next collection, so the pool stops growing; the old one is
        // parked in the allocator's cache and comes back at the next collection (two pools ping-pong, no hipMalloc)
        DevBuf<double> np;
        DevBuf<unsigned char> nc;
        CarveScope carve;
        const size_t cap = std::max<size_t>((size_t)live + (size_t)live / 2, gc_pool_points + ((size_t)1 << 27));
        np.alloc(cap * 3);
        nc.alloc(cap);
        if (!cat.empty()) {
            d_cat.ensure(cat.size());
            HIP_TRY(hipMemcpyAsync(d_cat.p, cat.data(), cat.size() * sizeof(CatSeg), hipMemcpyHostToDevice, s));
            if (blocks)
                hipLaunchKernelGGL(k_concat, dim3(blocks), dim3(256), 0, s, (const double*)pool.p, (const CatSeg*)d_cat.p, (int)cat.size(),
                                   np.p, (const unsigned char*)poolcore.p, nc.p, CAT_CHUNK);
            HMSG_CHECK_LAUNCH();
        }
        HIP_TRY(hipStreamSynchronize(s));
        pool.swap(np);
        poolcore.swap(nc);
        np.release();
        nc.release();
        pool_used = live;                // (the grid arenas keep their buffers: only their fill level is reset)
        ix_cells_used = ix_pts_used = 0;
        ++n_collects;
        // an episode whose LIVE clouds alone pass the threshold must not collect on every frame
        gc_pool_points = std::max(gc_pool_points, (size_t)live * 2);
    }
    // HMSG_DEBUG_NO_GRID_INHERIT=1: every merged cloud is indexed afresh (the form before round 4; comparison runs and tests)
    bool inherit_grids = getenv("HMSG_DEBUG_NO_GRID_INHERIT") == nullptr;
    // HMSG_DEBUG_GRID_DELTA_ALWAYS=1: an inherited base grid is kept whatever the cloud has gained (tests: small scenes never
    // meet the thresholds)
    bool delta_always = getenv("HMSG_DEBUG_GRID_DELTA_ALWAYS") != nullptr;
    double grid_full = 0, grid_full_pts = 0, grid_delta = 0, grid_delta_pts = 0;     // (statistics)
    size_t gc_pool_points = (size_t)1 << 30;         // 1.07 * 10^9 points = 26 GB
    size_t gc_index_entries = (size_t)1 << 31;       // grid cells (8 GB) / sorted points (26 GB)
    int n_collects = 0;
    static constexpr size_t PREBUILD_WINDOW = 64;    // frames (one fusion batch; measured: the overlap scans of a 1000-frame fold take 90 ms with 64-frame windows, 110 ms with one 1024-frame window)

    // ---- Overlap grids of the clouds a DBSCAN batch is about to put out, enqueued BEHIND the batch's publish: the GPU builds them
    // while the host waits for the results, reads them and does its bookkeeping (the ~40 us of index kernels used to open the
    // next step, in front of its overlap scans).  One entry per segment, merged components first and single clouds last (a single
    // cloud nearly always comes through unchanged and keeps its grids: the cells reserved for the trailing entries nobody used
    // are handed back).  Layout on the segment's input box; what is indexed is decided on the device (k_ov_spec).
    void speculate_indices(const std::vector<Cloud>& L, const CompList& comps, const std::vector<int>& seg_of_comp,
                           const std::vector<SegDesc>& segs, long long out_base, const unsigned* d_res, int K) {
        spec.clear();
        spec_of_seg.assign(segs.size(), -1);
        long long ncell_new = 0, npts_max = 0;
        unsigned nblk = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (size_t c = 0; c < comps.size(); ++c) {
                const int sg = seg_of_comp[c];
                if (sg < 0) continue;
                const auto& mem = comps[c];
                if ((mem.size() == 1) != (pass == 1)) continue;
                const SegDesc& sd = segs[(size_t)sg];
                if (sd.n <= 0) continue;
                int f = -1;
                for (int i : mem)
                    if (L[i].n > 0) {
                        f = i;
                        break;
                    }
                Cloud box;                                   // (carrier of the input box for grid_at)
                for (int a = 0; a < 3; ++a) {
                    box.mn[a] = sd.mn[a];
                    box.mx[a] = sd.mx[a];
                }
                const float lo[3] = {(float)sd.mn[0], (float)sd.mn[1], (float)sd.mn[2]};
                int gd[3];
                for (int a = 0; a < 3; ++a) gd[a] = (int)std::floor(((double)(float)sd.mx[a] - (double)lo[a] + 2e-3) / cell) + 1;
                const long long ncell = (long long)gd[0] * gd[1] * gd[2] + 1;   // +1: end sentinel
                OvSpec sp;
                sp.g = grid_at(box, lo, gd, ix_cells_used + ncell_new, ix_pts_used, 0, 0);
                sp.g.blk0 = (int)nblk;
                sp.seg = sg;
                sp.singleton = mem.size() == 1 ? 1 : 0;
                sp.inherit = (inherit_grids && f >= 0 && L[f].has_index && L[f].nb > 0) ? 1 : 0;
                sp.nb_first = f >= 0 ? L[f].nb : 0;
                sp.n_first = sd.n_first;
                sp.n_in = sd.n;
                sp.out_mode = sd.out_mode;
                sp.delta_always = delta_always ? 1 : 0;
                sp.out_off = sd.out_mode ? sd.out_off : out_base;
                sp.ncell = ncell;
                spec_of_seg[(size_t)sg] = (int)spec.size();
                spec.push_back(sp);
                nblk += cdiv((size_t)sd.n, OVI_CHUNK);
                ncell_new += ncell;
                npts_max += sd.n;
            }
        if (spec.empty()) return;
        if (!(ncell_new < (1ll << 31) && npts_max < (1ll << 31))) {      // (a batch of grids is bounded by its 32-bit cell starts: the next step builds them)
            spec.clear();
            spec_of_seg.assign(segs.size(), -1);
            return;
        }
        grow(ix_cells, (size_t)ix_cells_used, (size_t)(ix_cells_used + ncell_new));
        grow(ix_pts, (size_t)ix_pts_used * 3, (size_t)(ix_pts_used + npts_max) * 3);
        spec_turn ^= 1;
        PinnedBuf<OvSpec>& hs = h_spec[spec_turn];
        DevBuf<OvGrid>& ds = d_spec[spec_turn];
        hs.ensure(spec.size());
        ds.ensure(spec.size());
        memcpy(hs.p, spec.data(), spec.size() * sizeof(OvSpec));
        hipLaunchKernelGGL(k_ov_spec, dim3(cdiv(spec.size(), 64)), dim3(64), 0, s, (const OvSpec*)hs.p, (int)spec.size(), d_res, K, ds.p);
        {   // (the cursor is zero whenever k_ov_fill has run over what k_ov_count counted: only fresh memory is cleared)
            const unsigned* before = d_cursor.p;
            d_cursor.ensure((size_t)ncell_new);
            if (d_cursor.p != before) cursor_clean = 0;
            if (cursor_clean < (size_t)ncell_new) {
                HIP_TRY(hipMemsetAsync(d_cursor.p, 0, d_cursor.n * 4, s));
                cursor_clean = d_cursor.n;
            }
        }
        hipLaunchKernelGGL(k_ov_count, dim3(nblk), dim3(256), 0, s, (const double*)pool.p, (const OvGrid*)ds.p, (int)spec.size(),
                           d_cursor.p, ix_cells_used, OVI_CHUNK);
        HMSG_CHECK_LAUNCH();
        hmsg_scan_u32(d_cursor.p, ix_cells.p + ix_cells_used, (size_t)ncell_new, s, ops.scan_tmp, nullptr);
        hipLaunchKernelGGL(k_ov_fill, dim3(nblk), dim3(256), 0, s, (const double*)pool.p, (const OvGrid*)ds.p, (int)spec.size(),
                           (const unsigned*)ix_cells.p, d_cursor.p, ix_cells_used, ix_pts.p, OVI_CHUNK);
        HMSG_CHECK_LAUNCH();
    }

    // ---- overlap ratios for a list of (i, j) pairs of L.  `decide_th` >= 0 (sequential merge): only `ratio > th`
    // is needed downstream, so the pair's smaller cloud is counted first and the scan of the larger one is skipped on
    // the device when the first ratio already exceeds the threshold (its entry then reports the first ratio).
    void overlap_ratios(const std::vector<Cloud>& L, const std::vector<std::pair<int, int>>& pairs, std::vector<double>& ratio,
                        double decide_th, std::vector<unsigned char>* second_ran = nullptr) {
        ratio.assign(pairs.size(), 0.0);
        if (second_ran) second_ran->assign(pairs.size(), 0);
        if (pairs.empty()) return;
        const size_t P = pairs.size();
        // compact table of the clouds involved
        std::vector<int> slot(L.size(), -1);
        std::vector<OvGrid> g;
        std::vector<OvTask> tasks(P * 2);          // [0, P): smaller -> larger,  [P, 2P): larger -> smaller
        unsigned nblk1 = 0, nblk2 = 0;          // workgroups of the two directions (work lists)
        for (size_t k = 0; k < P; ++k) {
            int a = pairs[k].first, b = pairs[k].second;
            for (int v : {a, b})
                if (slot[v] < 0) {
                    slot[v] = (int)g.size();
                    push_grids(L[v], g);
                }
            if (L[a].n > L[b].n) std::swap(a, b);  // a = the smaller cloud
            tasks[k] = OvTask{slot[a], slot[b], 0, (int)nblk1};
            tasks[P + k] = OvTask{slot[b], slot[a], L[a].n, (int)nblk2};
            nblk1 += cdiv((size_t)L[a].n, OV_CHUNK);
            nblk2 += cdiv((size_t)L[b].n, OV_CHUNK);
        }
        for (size_t k = 0; k < P; ++k) tasks[P + k].blk0 += (int)nblk1;      // (one work list over both directions)
        const size_t nblk = (size_t)nblk1 + nblk2;
        // (the planned second launch finds its work itself: the workgroup -> task table then covers the first direction only --
        //  the second direction's ~10^4 entries were two thirds of the packed upload and of the host's time to fill it)
        static const bool two_launch = getenv("HMSG_OV_ONE_LAUNCH") == nullptr;
        static const bool plan_wanted = getenv("HMSG_OV_LEGACY_SECOND") == nullptr;   // HMSG_OV_LEGACY_SECOND=1: a workgroup per chunk of every larger cloud (until round 6)
        const bool plan = plan_wanted && two_launch && decide_th >= 0.0 && P <= (size_t)OV2_MAX_PAIRS;
        const size_t nblk_tab = plan ? (size_t)nblk1 : nblk;
        const size_t off_t = (g.size() * sizeof(OvGrid) + 15) & ~(size_t)15, off_c = off_t + tasks.size() * sizeof(OvTask),
                     off_b = off_c + tasks.size() * 4, pack = off_b + nblk_tab * 4;
        h_ovpack.ensure(pack);
        d_ovpack.ensure(pack);
        memcpy(h_ovpack.p, g.data(), g.size() * sizeof(OvGrid));
        memcpy(h_ovpack.p + off_t, tasks.data(), tasks.size() * sizeof(OvTask));
        memset(h_ovpack.p + off_c, 0, tasks.size() * 4);
        {   // workgroup -> task
            int* bt = (int*)(h_ovpack.p + off_b);
            for (size_t k = 0; k < (plan ? P : 2 * P); ++k) {
                const size_t e = k + 1 < 2 * P ? (size_t)tasks[k + 1].blk0 : nblk;
                for (size_t b = (size_t)tasks[k].blk0; b < e; ++b) bt[b] = (int)k;
            }
        }
        upload_pinned(d_ovpack.p, h_ovpack.p, pack, s);
        const OvGrid* const dg = (const OvGrid*)d_ovpack.p;
        const OvTask* const dt = (const OvTask*)(d_ovpack.p + off_t);
        unsigned* const dc = (unsigned*)(d_ovpack.p + off_c);
        const int* const db = (const int*)(d_ovpack.p + off_b);
        const float r = (float)reach;                // how far a witness can be (= radius; more in faiss's BLAS form, merger_init)
        const float r2 = (float)(radius * radius);   // `D < radius**2` with a float32 D (graph_utils.py:654-655)
        const size_t prof_idx = ops.prof->ev.size();        // (algorithmic bytes are filled in after the read-back)
        static const int ov_sorted_src = getenv("HMSG_OV_POOL_ORDER") == nullptr ? 1 : 0;   // HMSG_OV_POOL_ORDER=1: X's points from the pool (until round 5)
        // (Round 5, measured and withdrawn: the counts sent to the host by the LAST workgroup of the last launch instead of a
        //  k_publish launch behind it -- "count yourself done" needs an agent-scope release fence per workgroup, which on this
        //  chip writes the XCD's L2 back: k_ov_query 25.6 -> 157 us per launch, k_db_compact 13.7 -> 26.9 us; profiles/r05_fused_publish.txt.)
        unsigned long long* d_ovstat = nullptr;
        if (want_stats) {
            if (!ovstat.p) {
                ovstat.alloc(16);
                HIP_TRY(hipMemsetAsync(ovstat.p, 0, 128, s));
            }
            d_ovstat = ovstat.p;
        }
        auto* const ovk = d_ovstat ? (faiss_form ? k_ov_query<true, true> : k_ov_query<false, true>)
                                   : (faiss_form ? k_ov_query<true, false> : k_ov_query<false, false>);
        auto* const ovk2 = d_ovstat ? (faiss_form ? k_ov_query_second<true, true> : k_ov_query_second<false, true>)
                                    : (faiss_form ? k_ov_query_second<true, false> : k_ov_query_second<false, false>);
        {
            ProfScope ps(ops.prof, s, "k_ov_query", 0.0);       // (the kernel launches only: not the read-back below)
            if (!two_launch) {
                if (nblk)
                    hipLaunchKernelGGL(ovk, dim3((unsigned)nblk), dim3(256), 0, s, (const double*)pool.p, dg, dt,
                                       (const unsigned*)ix_cells.p, (const float*)ix_pts.p, r2, r, (int)P, db, 0u, dc, (const unsigned*)nullptr, decide_th, OV_CHUNK,
                                       d_ovstat, ov_sorted_src, 0);
            } else {
                for (int dir = 0; dir < 2; ++dir) {
                    const unsigned nb = dir ? nblk2 : nblk1;
                    if (!nb) continue;
                    if (dir && plan) {
                        // (a fixed grid that plans its own work from the verdicts in device memory; 512 workgroups take the ~90 chunks of
                        //  a usual step in one trip and the 400 of a floor that does get scanned as well)
                        hipLaunchKernelGGL(ovk2, dim3(std::min(nb, 512u)), dim3(256), 0, s, (const double*)pool.p, dg, dt,
                                           (const unsigned*)ix_cells.p, (const float*)ix_pts.p, r2, r, (int)P, dc, decide_th, OV_CHUNK, d_ovstat, ov_sorted_src);
                        continue;
                    }
                    hipLaunchKernelGGL(ovk, dim3(nb + ((!dir && plan) ? (unsigned)P : 0u)), dim3(256), 0, s, (const double*)pool.p, dg, dt,
                                       (const unsigned*)ix_cells.p, (const float*)ix_pts.p, r2, r, (int)P, db, dir ? nblk1 : 0u, dc,
                                       (dir && decide_th >= 0.0) ? (const unsigned*)dc : (const unsigned*)nullptr, decide_th, OV_CHUNK, d_ovstat, ov_sorted_src,
                                       (!dir && plan) ? 1 : 0);
                }
            }
        }
        HMSG_CHECK_LAUNCH();
        pub_counts.launch(s, (const unsigned*)dc, tasks.size());
        pub_counts.wait();
        const unsigned* hc = pub_counts.data();
        // Algorithmic bytes of the step (merge): `sum over bbox-overlapping pairs (n_A + n_B) * 12` -- the float32 points
        // of both clouds of every pair the filter passes, each read once.  What the implementation skips (the larger cloud's scan
        // when the smaller one's ratio decides, or when the box bound does) is its own cleverness, not a smaller problem.
        double ov_work = 0;
        for (size_t k = 0; k < P; ++k) {
            const int na = std::min(L[pairs[k].first].n, L[pairs[k].second].n), nb = std::max(L[pairs[k].first].n, L[pairs[k].second].n);
            const bool bounded = (hc[P + k] & 0x80000000u) != 0;      // second direction decided by the box bound (k_ov_query)
            ratio[k] = std::max((double)hc[k] / (double)na, bounded ? 0.0 : (double)hc[P + k] / (double)nb);
            ov_work += 12.0 * ((double)na + (double)nb);
            if (!(decide_th >= 0.0 && (double)hc[k] / (double)na > decide_th) && !bounded) {
                if (second_ran) (*second_ran)[k] = 1;
            }
        }
        if (ops.prof->enabled && prof_idx < ops.prof->ev.size()) ops.prof->ev[prof_idx].work = ov_work;
    }

    // ---- candidate pairs of merge_3d_masks (graph_utils.py:937-941): AABB IoU above the threshold; sequential merge:
    // only pairs with a new or changed member (shortcut 2); hierarchical merge: cached ratios of unchanged pairs
    void find_pairs(const std::vector<Cloud>& L, std::vector<std::pair<int, int>>& pairs, std::vector<double>& known,
                    std::vector<std::pair<int, int>>& known_pairs) {
        const int n = (int)L.size();
        auto consider = [&](int i, int j) {
            if (L[i].n == 0 || L[j].n == 0) return;   // find_overlapping_ratio_faiss returns 0 for empty clouds
            if (!(bbox_iou(L[i], L[j]) > iou_thresh)) return;
            if (use_cache && !L[i].fresh && !L[j].fresh) {
                auto it = ratio_cache.find(L[i].uid * 0x100000000ull + L[j].uid);
                if (it != ratio_cache.end()) {
                    known_pairs.emplace_back(i, j);
                    known.push_back(it->second);
                    return;
                }
            }
            pairs.emplace_back(i, j);
        };
        if (use_cache) {
            for (int i = 0; i < n; ++i)
                for (int j = i + 1; j < n; ++j) consider(i, j);
        } else {
            // shortcut (2): only pairs with a fresh member; enumerate from the (few) fresh clouds
            // (AABBs in SoA form: the reject test -- boxes disjoint on some axis, overlap volume 0 -- is the hot loop)
            // Round 6: the fresh clouds of a step -- a frame's masks and what the last step changed -- sit in the camera's view, the
            // list holds the whole scene: ONE pass keeps the clouds whose box meets the fresh clouds' common box, the per-cloud
            // passes then run over those (a tenth of the list at configs[1]); buffers are kept between steps.
            std::vector<double>* lo = fp_lo;
            std::vector<double>* hi = fp_hi;
            for (int a = 0; a < 3; ++a) {
                lo[a].resize((size_t)n);
                hi[a].resize((size_t)n);
            }
            fp_fr.resize((size_t)n);
            fp_cand.assign((size_t)n + 8, 0);
            std::vector<unsigned char>& fr = fp_fr;
            double u[6] = {1e300, -1e300, 1e300, -1e300, 1e300, -1e300};
            for (int j = 0; j < n; ++j) {
                fr[(size_t)j] = L[j].fresh ? 1 : 0;
                for (int a = 0; a < 3; ++a) {
                    lo[a][(size_t)j] = L[j].n ? L[j].mn[a] : 1e300;      // empty clouds never pair
                    hi[a][(size_t)j] = L[j].n ? L[j].mx[a] : -1e300;
                }
                if (fr[(size_t)j] && L[j].n)
                    for (int a = 0; a < 3; ++a) {
                        u[2 * a] = std::min(u[2 * a], L[j].mn[a]);
                        u[2 * a + 1] = std::max(u[2 * a + 1], L[j].mx[a]);
                    }
            }
            if (!(u[0] <= u[1])) return;                        // no fresh cloud with points
            box_mask(n, lo[0].data(), hi[0].data(), lo[1].data(), hi[1].data(), lo[2].data(), hi[2].data(), u, fp_cand.data());
            fp_sub.clear();
            for (int j = 0; j < n; ++j)
                if (fp_cand[(size_t)j]) fp_sub.push_back(j);
            const int m = (int)fp_sub.size();
            for (int a = 0; a < 3; ++a) {
                fp_slo[a].resize((size_t)m);
                fp_shi[a].resize((size_t)m);
                for (int q = 0; q < m; ++q) {
                    fp_slo[a][(size_t)q] = lo[a][(size_t)fp_sub[(size_t)q]];
                    fp_shi[a][(size_t)q] = hi[a][(size_t)fp_sub[(size_t)q]];
                }
            }
            fp_cand.assign((size_t)m + 8, 0);
            for (int i = 0; i < n; ++i) {
                if (!fr[(size_t)i] || L[i].n == 0) continue;
                // branch-free mask pass (vectorised), then a sparse walk over the few survivors
                unsigned char* cm = fp_cand.data();
                const double q[6] = {lo[0][(size_t)i], hi[0][(size_t)i], lo[1][(size_t)i], hi[1][(size_t)i], lo[2][(size_t)i], hi[2][(size_t)i]};
                box_mask(m, fp_slo[0].data(), fp_shi[0].data(), fp_slo[1].data(), fp_shi[1].data(), fp_slo[2].data(), fp_shi[2].data(), q, cm);
                for (int j0 = 0; j0 < m; j0 += 8) {
                    unsigned long long w;
                    std::memcpy(&w, cm + j0, 8);               // cand is padded to a multiple of 8
                    if (!w) continue;
                    for (int jq = j0; jq < std::min(j0 + 8, m); ++jq) {
                        const int j = fp_sub[(size_t)jq];
                        if (!cm[jq] || j == i || (fr[(size_t)j] && j < i)) continue;
                        consider(std::min(i, j), std::max(i, j));
                    }
                }
            }
        }
    }
    // (find_pairs' buffers, kept between steps)
    std::vector<double> fp_lo[3], fp_hi[3], fp_slo[3], fp_shi[3];
    std::vector<unsigned char> fp_fr, fp_cand;
    std::vector<int> fp_sub;

    // ---- components of `overlap > th` (scipy connected_components labels by lowest member index)
    void make_components(const std::vector<Cloud>& L, const std::vector<std::pair<int, int>>& pairs, const std::vector<double>& ratio,
                         const std::vector<std::pair<int, int>>& known_pairs, const std::vector<double>& known, double th,
                         CompList& comps) {
        const int n = (int)L.size();
        std::vector<int> parent(n);
        std::iota(parent.begin(), parent.end(), 0);
        auto find = [&](int x) {
            while (parent[x] != x) x = parent[x] = parent[parent[x]];
            return x;
        };
        auto unite = [&](int a, int b) {
            a = find(a);
            b = find(b);
            if (a != b) parent[std::max(a, b)] = std::min(a, b);
        };
        for (size_t k = 0; k < pairs.size(); ++k) {
            if (use_cache) ratio_cache[L[pairs[k].first].uid * 0x100000000ull + L[pairs[k].second].uid] = ratio[k];
            if (ratio[k] > th) unite(pairs[k].first, pairs[k].second);
        }
        for (size_t k = 0; k < known_pairs.size(); ++k)
            if (known[k] > th) unite(known_pairs[k].first, known_pairs[k].second);
        // (flat CSR: thousands of clouds per step, nearly all singletons -- no per-component allocations)
        {
            std::vector<int> comp_of(n, -1), cid(n);
            int nc = 0;
            for (int i = 0; i < n; ++i) {
                int r = find(i);
                if (comp_of[r] < 0) comp_of[r] = nc++;      // roots are lowest members: components in index order
                cid[i] = comp_of[r];
            }
            comps.off.assign((size_t)nc + 1, 0);
            for (int i = 0; i < n; ++i) ++comps.off[cid[i] + 1];
            for (int c = 0; c < nc; ++c) comps.off[c + 1] += comps.off[c];
            comps.mem.resize(n);
            std::vector<int> cur(comps.off.begin(), comps.off.end() - 1);
            for (int i = 0; i < n; ++i) comps.mem[cur[cid[i]]++] = i;
        }
    }

    // ---- merge_3d_masks (graph_utils.py:918-956)
    std::vector<Cloud> merge_3d_masks(std::vector<Cloud> L, double th) {
        const int n = (int)L.size();
        if (n == 0) return L;
        auto tnow = [] { return std::chrono::steady_clock::now(); };
        auto t0 = tnow();
        auto lap = [&](int k) {
            auto t1 = tnow();
            tphase[k] += std::chrono::duration<double, std::milli>(t1 - t0).count();
            t0 = t1;
        };
        if (want_stats)
            for (auto& c : L) c.has_index_before = c.has_index;
        build_indices(L);
        lap(0);
        // 1. candidate pairs
        std::vector<std::pair<int, int>> pairs;
        std::vector<double> known;                 // cached ratios (hierarchical)
        std::vector<std::pair<int, int>> known_pairs;
        find_pairs(L, pairs, known, known_pairs);
        std::vector<double> ratio;
        lap(1);
        std::vector<unsigned char> second_ran;
        overlap_ratios(L, pairs, ratio, use_cache ? -1.0 : th, want_stats ? &second_ran : nullptr);
        lap(2);
        if (want_stats) {
            st.steps += 1;
            st.clouds += n;
            for (int i = 0; i < n; ++i) {
                if (L[i].fresh && L[i].raw) st.fresh_raw += 1;
                if (L[i].fresh && !L[i].raw) st.fresh_g += 1;
                if (!L[i].has_index_before) st.idx_pts += L[i].n;
            }
            for (size_t k = 0; k < pairs.size(); ++k) {
                const Cloud &a = L[pairs[k].first], &b = L[pairs[k].second];
                const bool g = (a.fresh && !a.raw) || (b.fresh && !b.raw);       // a changed G cloud is involved
                const double n1 = std::min(a.n, b.n), n2 = second_ran[k] ? std::max(a.n, b.n) : 0;
                (g ? st.pairs_g : st.pairs_raw) += 1;
                (g ? st.scan1_g : st.scan1_raw) += n1;
                (g ? st.scan2_g : st.scan2_raw) += n2;
            }
        }
        {   // HMSG_DEBUG_OVERLAP_LOG=<file>: (ratio, threshold) of every pair this call evaluated, float64 pairs appended -- what
            // scripts/fuzz/faiss_form_configs1.py compares between the two distance forms (read once: the fold may run on a worker)
            static const char* const ovlog = getenv("HMSG_DEBUG_OVERLAP_LOG");
            if (ovlog && !pairs.empty()) {
                if (FILE* f = fopen(ovlog, "ab")) {
                    for (size_t k = 0; k < pairs.size(); ++k) {
                        const double rec[2] = {ratio[k], th};
                        fwrite(rec, 8, 2, f);
                    }
                    fclose(f);
                }
            }
        }
        // 2. components of `overlap > th`
        CompList comps;
        make_components(L, pairs, ratio, known_pairs, known, th, comps);
        // 3. merge_point_clouds_list per component: concat in index order + keep-largest DBSCAN
        std::vector<int> seg_of_comp(comps.size(), -1);
        std::vector<SegDesc> segs;
        std::vector<int> seg_cap;                  // per segment: capacity of its output region (0: dense output)
        long long region_total = 0;                // points of the fresh output regions (SegDesc::out_mode 1)
        const bool inplace = use_anchor && inplace_wanted && CloudOps::regions_supported();
        std::vector<CatSeg> cat;
        long long cat_total = 0;
        unsigned cat_blocks = 0;
        for (size_t c = 0; c < comps.size(); ++c) {
            const auto& mem = comps[c];
            if (mem.size() == 1 && (L[mem[0]].fixed || L[mem[0]].n == 0)) continue;   // exact shortcut (1)
            SegDesc sd;
            sd.pt_base = cat_total;
            sd.n = 0;
            bool any = false;
            int anchor_i = -1;                  // the largest member that can serve as the DBSCAN anchor
            if (use_anchor)
                for (int i : mem)
                    if (L[i].anchor && (anchor_i < 0 || L[i].n > L[anchor_i].n)) anchor_i = i;
            for (int i : mem) {
                if (L[i].n == 0) continue;
                cat.push_back(CatSeg{L[i].off, cat_total, L[i].n, i == anchor_i ? 1 : 0, (int)cat_blocks, 0});
                cat_blocks += cdiv((size_t)L[i].n, CAT_CHUNK);
                cat_total += L[i].n;
                if (!any) sd.n_first = L[i].n;             // (its overlap grid may outlive the merge: first_kept)
                sd.n += L[i].n;
                for (int a = 0; a < 3; ++a) {
                    sd.mn[a] = any ? std::min(sd.mn[a], L[i].mn[a]) : L[i].mn[a];
                    sd.mx[a] = any ? std::max(sd.mx[a], L[i].mx[a]) : L[i].mx[a];
                }
                any = true;
            }
            // Crop (SegDesc::forced): the first member is the component's anchor and larger than the rest together -- all of it
            // is kept and its cluster wins; only what lies within 2 eps of the other members' boxes goes into the DBSCAN grid.
            if (use_anchor && anchor_i >= 0 && sd.n_first > 0 && sd.n_first < sd.n && (long long)sd.n_first * 2 > (long long)sd.n) {
                int f = -1;
                for (int i : mem)
                    if (L[i].n > 0) {
                        f = i;
                        break;
                    }
                if (f == anchor_i) {
                    bool any_r = false;
                    for (int i : mem) {
                        if (i == f || L[i].n == 0) continue;
                        for (int a = 0; a < 3; ++a) {
                            sd.cmn[a] = any_r ? std::min(sd.cmn[a], L[i].mn[a]) : L[i].mn[a];
                            sd.cmx[a] = any_r ? std::max(sd.cmx[a], L[i].mx[a]) : L[i].mx[a];
                        }
                        any_r = true;
                    }
                    const double grow = 2.0 * eps * (1.0 + 1e-9) + 1e-6;
                    for (int a = 0; a < 3; ++a) {
                        sd.cmn[a] -= grow;
                        sd.cmx[a] += grow;
                        sd.fmn[a] = L[f].mn[a];
                        sd.fmx[a] = L[f].mx[a];
                    }
                    sd.forced = any_r ? 1 : 0;
                    // Round 5: such a segment's result IS its anchor member with the kept rest appended.  With room behind the member
                    // in the pool (Cloud::cap) the batch runs in place -- the member is neither gathered into the batch nor copied to
                    // a fresh pool range (7.5 MB in and out per step at configs[1]), only its points inside the crop are read; without
                    // room, this step's output goes to a region of its own with half as much again, and the next ones run in place.
                    if (sd.forced && inplace) {
                        if ((long long)L[f].cap >= (long long)sd.n) {
                            sd.out_mode = 2;
                            sd.out_off = L[f].off;
                            seg_cap.push_back(L[f].cap);
                        } else {
                            sd.out_mode = 1;
                            const long long cap = (long long)sd.n + std::max<long long>(sd.n / 2, 1 << 14);
                            sd.out_off = region_total;       // (relative to the first region: fixed up below)
                            region_total += cap;
                            HMSG_REQUIRE(cap <= 0x7fffffffll, HMSG_ERR_UNSUPPORTED, "merge fold: a merged cloud's pool region exceeds 2^31 points");
                            seg_cap.push_back((int)cap);
                        }
                    }
                }
            }
            if (seg_cap.size() <= segs.size()) seg_cap.push_back(0);
            seg_of_comp[c] = (int)segs.size();
            segs.push_back(sd);
        }
        std::vector<DbscanResult> res;
        // pool layout of the step's outputs: [fresh regions of the mode-1 segments | dense outputs of the mode-0 segments]
        long long dense_total = 0;
        for (auto& sd : segs) {
            if (sd.out_mode == 1) sd.out_off += pool_used;
            if (sd.out_mode == 0) dense_total += sd.n;
        }
        long long out_base = pool_used + region_total;
        if (!segs.empty() && cat_total > 0) {
            concat.ensure((size_t)cat_total * 3);
            concat_core.ensure((size_t)cat_total);
            // (the concatenation itself happens inside the DBSCAN batch's binning pass, and the piece table goes up with the
            //  batch's geometry table: DbGather)
            grow(pool, (size_t)pool_used * 3, (size_t)(out_base + dense_total) * 3);
            grow(poolcore, (size_t)pool_used, (size_t)(out_base + dense_total));
            lap(3);
            DbGather ga;
            ga.pool = pool.p;
            ga.host_segs = cat.data();
            ga.nsegs = (int)cat.size();
            ga.poolcore = poolcore.p;
            ga.dstcore = use_anchor ? concat_core.p : nullptr;
            ga.pool_w = pool.p;
            ga.poolcore_w = poolcore.p;
            spec.clear();
            spec_of_seg.assign(segs.size(), -1);
            const bool speculate = spec_wanted && CloudOps::regions_supported();
            ops.dbscan_keep_largest(concat.p, segs, eps, minpts, pool.p + (size_t)out_base * 3, res,
                                    use_anchor ? (const unsigned char*)concat_core.p : nullptr, poolcore.p + out_base, &ga,
                                    speculate ? std::function<void(const unsigned*, int)>([&](const unsigned* d_res, int K) {
                                        speculate_indices(L, comps, seg_of_comp, segs, out_base, d_res, K);
                                    }) : std::function<void(const unsigned*, int)>());
            lap(4);
            if (want_stats) {
                for (size_t c = 0; c < comps.size(); ++c) {
                    if (seg_of_comp[c] < 0) continue;
                    const auto& mem = comps[c];
                    const DbscanResult& r = res[seg_of_comp[c]];
                    long long tot = 0, rest = 0;
                    bool rest_raw = true;
                    for (size_t q = 0; q < mem.size(); ++q) {
                        tot += L[mem[q]].n;
                        if (q) {
                            rest += L[mem[q]].n;
                            rest_raw = rest_raw && L[mem[q]].raw;
                        }
                    }
                    int cls;
                    if (mem.size() == 1) cls = L[mem[0]].raw ? 0 : 1;
                    else if (L[mem[0]].anchor && L[mem[0]].n > rest) cls = rest_raw ? 2 : 3;
                    else cls = 4;
                    st.ccount[cls] += 1;
                    st.cpts[cls] += tot;
                    st.cB[cls] += mem.size() == 1 ? tot : rest;
                    st.cmem[cls] += mem.size();
                    st.cchanged[cls] += r.changed;
                    st.cmulti[cls] += r.n_clusters > 1;
                    st.ccontested[cls] += r.contested != 0;
                    const bool fx = !r.changed || r.n_clusters == 1 || (r.n_clusters > 1 && !r.contested);
                    st.cnonfixed[cls] += !fx;
                }
            }
            static const bool stats_step = getenv("HMSG_DEBUG_MERGESTATS_STEP") != nullptr;   // (read once: the fold may run on a worker thread)
            if (stats_step) {
                long long fixed_pts = 0, big_fixed = 0, removed = 0;
                int nchanged = 0, multi = 0, multicl = 0, multicl_changed = 0;
                long long multicl_pts = 0;
                for (size_t c = 0; c < comps.size(); ++c) {
                    if (seg_of_comp[c] < 0) continue;
                    multi += comps[c].size() > 1;
                    int best = 0;
                    for (int i : comps[c])
                        if (L[i].fixed) {
                            fixed_pts += L[i].n;
                            best = std::max(best, L[i].n);
                        }
                    big_fixed += best;
                    const DbscanResult& r = res[seg_of_comp[c]];
                    nchanged += r.changed;
                    removed += segs[seg_of_comp[c]].n - r.n_out;
                    if (r.n_clusters > 1) {
                        ++multicl;
                        multicl_pts += r.n_out;
                        multicl_changed += r.changed;
                    }
                }
                fprintf(stderr, "[mstat] clouds %d pairs %zu segs %zu multi %d N %lld fixed_pts %lld anchor_pts %lld changed %d removed %lld multicl %d multicl_changed %d multicl_pts %lld\n",
                        n, pairs.size(), segs.size(), multi, cat_total, fixed_pts, big_fixed, nchanged, removed, multicl,
                        multicl_changed, multicl_pts);
            }
        } else {
            res.assign(segs.size(), DbscanResult{});
            spec.clear();
            spec_of_seg.assign(segs.size(), -1);
        }
        // the grids built behind the batch: which of them exist (the device's decisions, replayed from the results) and how much
        // of the arenas they took -- sorted points densely one grid after the other, cells as laid out up to the last grid in use
        std::vector<int> spec_code(spec.size(), 0), spec_first(spec.size(), 0);
        std::vector<long long> spec_pt(spec.size(), 0);
        {
            long long pts = 0, cells_end = 0;
            for (size_t j = 0; j < spec.size(); ++j) {
                const DbscanResult& r = res[(size_t)spec[j].seg];
                int n_idx = 0;
                spec_code[j] = ov_spec_decide(spec[j], r.n_out, r.first_kept, &spec_first[j], &n_idx);
                spec_pt[j] = pts;
                pts += n_idx;
                if (n_idx > 0) {
                    cells_end = (spec[j].g.ix_cell - ix_cells_used) + spec[j].ncell;
                    spec_built += 1;
                    spec_built_pts += n_idx;
                    (spec_code[j] == 2 ? grid_delta : grid_full) += 1;
                    (spec_code[j] == 2 ? grid_delta_pts : grid_full_pts) += n_idx;
                } else {
                    spec_skipped += 1;
                }
            }
            // (all of this step's grids share ix_pt = the arena level before them: cell starts are relative to it)
            for (size_t j = 0; j < spec.size(); ++j) spec_pt[j] = ix_pts_used;
            ix_cells_used += cells_end;
            ix_pts_used += pts;
        }
        // 4. new list in component order
        std::vector<Cloud> out;
        out.reserve(comps.size());
        long long cursor = out_base;
        for (size_t c = 0; c < comps.size(); ++c) {
            const auto& mem = comps[c];
            int sg = seg_of_comp[c];
            if (sg < 0) {                       // untouched singleton
                Cloud k = L[mem[0]];
                k.fresh = false;
                k.fixed = true;
                out.push_back(k);
                continue;
            }
            const DbscanResult& r = res[sg];
            if (mem.size() == 1 && !r.changed) {   // DBSCAN kept every point: same cloud, now known fixed
                Cloud k = L[mem[0]];
                k.fresh = false;
                k.fixed = true;
                k.raw = false;
                k.off = cursor;                 // the identical copy DBSCAN just wrote: it carries the core flags
                k.cap = k.n;
                k.anchor = r.n_clusters == 1;
                out.push_back(k);
                cursor += r.n_out;
                continue;
            }
            Cloud k;
            const int mode = segs[sg].out_mode;
            k.off = mode ? segs[sg].out_off : cursor;      // (an output region of its own / the anchor member's own place)
            k.cap = mode ? seg_cap[sg] : r.n_out;
            k.n = r.n_out;
            for (int a = 0; a < 3; ++a) {
                k.mn[a] = r.mn[a];
                k.mx[a] = r.mx[a];
            }
            // Fixed point of pcd_denoise_dbscan: all points kept, or no dropped point was an eps-neighbour of a
            // core point of the kept cluster.  Neighbours of a core point are core (same cluster) or border, so
            // the only dropped points that can be such neighbours are border points that cores of TWO clusters
            // reach and that went to the other one ("contested"; impossible with a single cluster).  Without
            // them, dropping the rest changes no kept core point's neighbour count: the core graph and every
            // kept border point's witness survive, non-core points stay non-core, a second pass keeps everything
            // ... and the core flags of the kept points are exactly the ones this pass computed.
            k.fixed = !r.changed || r.n_clusters == 1 || (r.n_clusters > 1 && !r.contested);
            k.anchor = k.fixed && r.n_clusters >= 1 && (r.changed || r.n_clusters == 1);
            k.fresh = true;
            k.uid = next_uid++;
            // the first member came through whole: its points are this cloud's first points, in order, and its BASE grid
            // (a sorted copy of exactly its first nb points) goes on answering for them
            int f = -1;
            for (int i : mem)
                if (L[i].n > 0) {
                    f = i;
                    break;
                }
            auto take_base = [&] {
                k.nb = L[f].nb;
                k.ix_cell = L[f].ix_cell;
                k.ix_pt = L[f].ix_pt;
                for (int a = 0; a < 3; ++a) {
                    k.gd[a] = L[f].gd[a];
                    k.bmn[a] = L[f].bmn[a];
                }
            };
            const int sj = spec_of_seg.empty() ? -1 : spec_of_seg[(size_t)sg];
            if (sj >= 0) {
                // its grids exist already (speculate_indices): 1 the first member's base grid covers it, 2 base + delta, 3 a grid of its own
                const OvSpec& sp = spec[(size_t)sj];
                const int code = spec_code[(size_t)sj];
                if (code == 1 || code == 2) take_base();
                if (code == 2) {
                    k.has_delta = true;
                    k.ix_cell2 = sp.g.ix_cell;
                    k.ix_pt2 = spec_pt[(size_t)sj];
                    k.gd2[0] = sp.g.gx, k.gd2[1] = sp.g.gy, k.gd2[2] = sp.g.gz;
                    for (int a = 0; a < 3; ++a) k.dmn[a] = (float)segs[sg].mn[a];
                }
                if (code == 3) {
                    k.nb = k.n;
                    k.ix_cell = sp.g.ix_cell;
                    k.ix_pt = spec_pt[(size_t)sj];
                    k.gd[0] = sp.g.gx, k.gd[1] = sp.g.gy, k.gd[2] = sp.g.gz;
                    for (int a = 0; a < 3; ++a) k.bmn[a] = (float)segs[sg].mn[a];
                }
                k.has_index = code != 0;
            } else if (inherit_grids && r.first_kept >= 0 && f >= 0 && r.first_kept == L[f].n && L[f].has_index && L[f].nb > 0) {
                take_base();
                k.has_index = k.nb == k.n;
            }
            out.push_back(k);
            if (!mode) cursor += r.n_out;
        }
        pool_used = cursor;
        lap(5);
        return out;
    }
};

}  // namespace

#include "hmsg_fold.inl"

namespace {

void merger_init(Merger& m, hmsg_ctx* h) {
    const hmsg_config& c = h->cfg;
    m.h = h;
    m.s = h->stream;
    m.ops.s = h->stream;
    m.ops.prof = &h->prof;
    m.radius = 1.5 * c.voxel_size;
    m.reach = m.radius;
    m.faiss_form = c.overlap_distance_form == HMSG_OVERLAP_FAISS_BLAS ? 1 : 0;
    if (m.faiss_form) {
        // |x|^2 + |y|^2 - 2 x.y in float32: three 3-term sums of magnitude <= M2 = max |p|^2 over the map, each within 3 ulp,
        // then two more roundings of numbers <= 2 M2: the value is within E = 16 * 2^-24 * M2 of the true squared distance, so
        // a point up to sqrt(radius^2 + E) away can still compare below radius^2 -- the grids must reach that far
        const GridGeom& g = h->grid;
        double m2 = 0.0;
        const double lo[3] = {g.ox, g.oy, g.oz}, ext[3] = {g.nx * g.vs, g.ny * g.vs, g.nz * g.vs};
        for (int a = 0; a < 3; ++a) {
            const double v = std::max(std::fabs(lo[a]), std::fabs(lo[a] + ext[a])) + 1.0;
            m2 += v * v;
        }
        m.reach = std::sqrt(m.radius * m.radius + 16.0 * 5.9604644775390625e-08 * m2);
    }
    m.cell = m.reach * (1.0 + 1e-3) + 2e-3;
    m.eps = c.merge_dbscan_eps;
    m.minpts = c.merge_dbscan_min;
    m.iou_thresh = c.iou_thresh;
    m.use_anchor = !getenv("HMSG_DEBUG_NOANCHOR");
    m.want_stats = getenv("HMSG_DEBUG_MERGESTATS") != nullptr;
    HMSG_REQUIRE(c.iou_thresh >= 0.0, HMSG_ERR_UNSUPPORTED, "pipeline.iou_thresh must be >= 0");
}

// the frames' 3-D masks as per-frame cloud lists over a pool seeded with them (frames first .. n_fused-1)
std::vector<std::vector<Cloud>> seed_frames(Merger& m, hmsg_ctx* h, int first, bool legacy_grids = true, long long pool_factor = 2) {
    const int F = h->n_fused;
    const long long total = h->masks3d.total;
    m.pool.alloc((size_t)std::max<long long>(total * pool_factor, 1 << 16) * 3);
    m.poolcore.alloc((size_t)std::max<long long>(total * pool_factor, 1 << 16));
    if (total) HIP_TRY(hipMemcpyAsync(m.pool.p, h->masks3d.pts.p, (size_t)total * 24, hipMemcpyDeviceToDevice, h->stream));
    m.pool_used = total;
    // AABBs of the frame masks (device reduction)
    std::vector<SegDesc> msegs((size_t)h->mask_first[F]);
    for (size_t id = 0; id < msegs.size(); ++id) {
        msegs[id].pt_base = h->masks3d.off[id];
        msegs[id].n = (int)(h->masks3d.off[id + 1] - h->masks3d.off[id]);
    }
    m.ops.bounds(m.pool.p, msegs);
    std::vector<std::vector<Cloud>> frames((size_t)(F - first));
    // Empty masks are left out: an empty cloud never pairs (find_overlapping_ratio_faiss returns 0 for it), so it
    // stays a singleton through every step and is dropped by the min-points filter at the end (graph.py:445-448).
    for (int f = first; f < F; ++f) {
        const int nm = (int)(h->mask_first[f + 1] - h->mask_first[f]);
        auto& fr = frames[(size_t)(f - first)];
        fr.reserve(nm);
        for (int i = 0; i < nm; ++i) {
            const SegDesc& sd = msegs[(size_t)h->mask_first[f] + i];
            if (sd.n == 0) continue;
            fr.emplace_back();
            Cloud& k = fr.back();
            k.off = sd.pt_base;
            k.n = sd.n;
            k.raw = true;
            k.uid = m.next_uid++;
            for (int a = 0; a < 3; ++a) {
                k.mn[a] = sd.mn[a];
                k.mx[a] = sd.mx[a];
            }
        }
    }
    // overlap grids of ALL frame masks in one batch (they are inputs of the fold; only clouds that change during
    // the fold get a new grid later)
    if (!legacy_grids) return frames;
    if (h->cfg.merge_type == HMSG_MERGE_HIERARCHICAL || first != 0) {
        for (size_t a = 0; a < frames.size(); a += Merger::PREBUILD_WINDOW) m.prebuild(frames, a, a + Merger::PREBUILD_WINDOW);
    } else {
        m.prebuild(frames, 0, Merger::PREBUILD_WINDOW);      // (the fold builds the later windows when it gets there)
    }
    return frames;
}

// compact the clouds of `result` with at least `min_points` points into the handle's instance list
void store_instances(Merger& m, hmsg_ctx* h, const std::vector<Cloud>& result, int min_points) {
    long long keep_total = 0;
    unsigned cat_blocks = 0;
    std::vector<CatSeg> cat;
    h->inst.off.assign(1, 0);
    h->inst.box.clear();
    for (auto& k : result) {
        if (k.n < min_points) continue;
        for (int a = 0; a < 3; ++a) h->inst.box.push_back(k.mn[a]);
        for (int a = 0; a < 3; ++a) h->inst.box.push_back(k.mx[a]);
        cat.push_back(CatSeg{k.off, keep_total, k.n, 0, (int)cat_blocks, 0});
        cat_blocks += cdiv((size_t)k.n, CAT_CHUNK);
        keep_total += k.n;
        h->inst.off.push_back(keep_total);
    }
    h->inst.total = keep_total;
    DevBuf<double> fresh;                     // (the pool may alias the handle's current instance buffer: tree join)
    fresh.alloc((size_t)std::max<long long>(keep_total, 1) * 3);
    if (!cat.empty()) {
        m.d_cat.ensure(cat.size());
        HIP_TRY(hipMemcpyAsync(m.d_cat.p, cat.data(), cat.size() * sizeof(CatSeg), hipMemcpyHostToDevice, m.s));
        if (cat_blocks)
            hipLaunchKernelGGL(k_concat, dim3(cat_blocks), dim3(256), 0, m.s, (const double*)m.pool.p, (const CatSeg*)m.d_cat.p,
                               (int)cat.size(), fresh.p, (const unsigned char*)nullptr, (unsigned char*)nullptr, CAT_CHUNK);
        HMSG_CHECK_LAUNCH();
    }
    HIP_TRY(hipStreamSynchronize(m.s));
    fresh.swap(h->inst.pts);
}

// threshold after a level that left `lists` lists (graph_utils.py:1001-1003)
double next_level_threshold(double th, double factor, long long lists) {
    return th - factor * (double)(lists - 2) / (double)std::max<long long>(1, lists - 1);
}

}  // namespace

// Switch a sequential fold to the incremental fold of hmsg_fold.inl: the live clouds (the instance list G and the masks
// of the frames still to come) move to a fresh pool with room to grow, get an id each and are indexed in one batch.
// false: configuration outside what the fold index supports (the batch fold carries on).
static bool fold_begin(Folder& m, hmsg_ctx* h, std::vector<Cloud>& G, std::vector<std::vector<Cloud>>& frames, size_t f_next) {
    const double cs = m.eps / std::sqrt(3.0) * (1.0 - 1e-7);
    if (!(m.radius + 1e-4 < 1.9 * cs)) return false;
    if (m.faiss_form) return false;     // (the incremental fold's overlap test evaluates the direct form only: the batch fold carries on)
    long long live = 0, n_clouds = 0;
    for (auto& k : G) live += k.n, ++n_clouds;
    for (size_t f = f_next; f < frames.size(); ++f)
        for (auto& k : frames[f]) live += k.n, ++n_clouds;
    if (n_clouds == 0 || live >= (1ll << 28) || n_clouds >= (1ll << 23)) return false;
    // (the fold's clouds keep room to grow and are relocated when they outgrow it: ~8 pool points per mask point over a
    //  1000-frame scene -- allocated once, growing a multi-GB buffer costs a fresh hipMalloc and a copy)
    // (the arenas of the fold -- pool, hash, bricks, records: ~120 GB for a 10 000-frame episode -- are cut out of the frame
    //  store's block when hmsg_merge_instances handed a very large one back: DevCache carving.  They all die with `m`.)
    CarveScope carve;
    const size_t keep_gc = m.gc_pool_points;
    m.gc_pool_points = (size_t)live * 11;
    m.collect(G, frames, f_next);
    m.gc_pool_points = keep_gc;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    std::vector<FInsSeg> segs;
    auto take = [&](Cloud& k) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], k.mn[a]);
            hi[a] = std::max(hi[a], k.mx[a]);
        }
        k.id = m.next_id++;
        k.cap = k.n;
        segs.push_back(FInsSeg{k.off, k.id, k.n, 0, 0, k.anchor ? 1 : 0, 0});
    };
    for (auto& k : G) take(k);
    for (size_t f = f_next; f < frames.size(); ++f)
        for (auto& k : frames[f]) take(k);
    m.index_init(live, n_clouds, lo, hi);
    m.index_bulk(segs);
    return true;
}
static void fold_report(Folder& m, hmsg_ctx* h) {
    fprintf(stderr, "[hmsg fold] pairs(host) %.1f  overlap %.1f  tables %.1f  step kernels %.1f  bookkeeping %.1f ms\n", m.tphase[1],
            m.tphase[2], m.tphase[3], m.tphase[4], m.tphase[5]);
    fprintf(stderr, "[hmsg fold] steps %.0f  components: anchor %.0f  plain %.0f  batch %.0f   active points %.0f  relocated %.0f   ids %u\n",
            m.fstat[0], m.fstat[1], m.fstat[2], m.fstat[3], m.fstat[4], m.fstat[5], m.next_id);
    unsigned cnt[FC_N];
    HIP_TRY(hipMemcpy(cnt, m.ix_counters.p, sizeof(cnt), hipMemcpyDeviceToHost));
    fprintf(stderr, "[hmsg fold] index: %u of %u bricks, %u of %u records, hash %u slots\n", cnt[FC_BRICKS], m.ix.brick_cap, cnt[FC_RECS],
            m.ix.rec_cap, m.ix.hmask + 1);
    fprintf(stderr, "[hmsg fold] point pool: %lld points in use (%.2f GB), %lld in the frame masks\n", m.pool_used, m.pool_used * 25e-9, (long long)h->masks3d.total);
    fprintf(stderr, "[hmsg fold] walks: count %u  touch %u  link %u  label %u   touched anchor points %u\n", cnt[FC_STAT], cnt[FC_STAT + 1],
            cnt[FC_STAT + 2], cnt[FC_STAT + 3], cnt[FC_STAT + 4]);
    fprintf(stderr, "[hmsg fold] link walks: anchor comps own %u promoted %u raw %u | plain comps own %u promoted %u raw %u\n", cnt[FC_DBG], cnt[FC_DBG + 1],
            cnt[FC_DBG + 2], cnt[FC_DBG + 4], cnt[FC_DBG + 5], cnt[FC_DBG + 6]);
}

static void merge_report(Folder& m) {
    if (getenv("HMSG_DEBUG_TIMING"))
        fprintf(stderr, "[hmsg merge] index %.1f  pairs(host) %.1f  overlap %.1f  components+concat %.1f  dbscan %.1f  bookkeeping %.1f ms\n",
                m.tphase[0], m.tphase[1], m.tphase[2], m.tphase[3], m.tphase[4], m.tphase[5]);
    if (m.want_stats) {
        const auto& t = m.st;
        const double S = std::max(1.0, t.steps);
        fprintf(stderr, "[mstat] steps %.0f  per step: clouds %.1f  fresh raw %.1f  fresh G %.1f  indexed pts %.0f\n", t.steps, t.clouds / S,
                t.fresh_raw / S, t.fresh_g / S, t.idx_pts / S);
        fprintf(stderr, "[mstat] pairs/step raw %.1f (scan1 %.0f scan2 %.0f pts)   G-fresh %.1f (scan1 %.0f scan2 %.0f pts)\n", t.pairs_raw / S,
                t.scan1_raw / S, t.scan2_raw / S, t.pairs_g / S, t.scan1_g / S, t.scan2_g / S);
        if (m.ovstat.p) {
            unsigned long long hs[16];
            (void)hipMemcpy(hs, m.ovstat.p, 128, hipMemcpyDeviceToHost);
            fprintf(stderr, "[mstat] overlap probes with > 32 candidates: %.1f per step, %.0f candidates per step; longest candidate list of a probe %llu\n",
                    hs[8] / S, hs[9] / S, hs[7]);
            fprintf(stderr, "[mstat] overlap probes/step %.0f: outside the box %.0f  hit own cell %.0f  hit neighbour %.0f  miss %.0f | candidates/step own %.0f  neighbour %.0f\n",
                    hs[0] / S, hs[1] / S, hs[2] / S, hs[3] / S, hs[4] / S, hs[5] / S, hs[6] / S);
        }
        const char* nmc[5] = {"singleton raw", "singleton non-fixed", "anchor+raw", "anchor+any", "other"};
        for (int c = 0; c < 5; ++c)
            fprintf(stderr, "[mstat] dbscan class %-20s comps/step %.2f  pts/step %.0f  B pts/step %.0f  members %.2f  changed %.0f multi %.0f contested %.0f nonfixed %.0f (totals)\n",
                    nmc[c], t.ccount[c] / S, t.cpts[c] / S, t.cB[c] / S, t.ccount[c] ? t.cmem[c] / t.ccount[c] : 0.0, t.cchanged[c],
                    t.cmulti[c], t.ccontested[c], t.cnonfixed[c]);
    }
    if (getenv("HMSG_DEBUG_TIMING"))
        fprintf(stderr, "[hmsg merge] overlap grids: %.0f over whole clouds (%.0f points), %.0f delta grids (%.0f points); of them behind the DBSCAN batches: %.0f (%.0f points), %.0f not needed\n", m.grid_full,
                m.grid_full_pts, m.grid_delta, m.grid_delta_pts, m.spec_built, m.spec_built_pts, m.spec_skipped);
    if (getenv("HMSG_DEBUG_TIMING") && m.n_collects)
        fprintf(stderr, "[hmsg merge] %d collections of the point pool / grid arenas\n", m.n_collects);
    if (getenv("HMSG_DEBUG_TIMING") && m.ops.stat_calls > 0)
        fprintf(stderr, "[hmsg merge] dbscan batches %.0f: mean points %.0f  grid cells %.0f  occupied cells %.0f  counted points %.0f\n",
                m.ops.stat_calls, m.ops.stat_points / m.ops.stat_calls, m.ops.stat_cells / m.ops.stat_calls,
                m.ops.stat_core_cells / m.ops.stat_calls, m.ops.stat_needy / m.ops.stat_calls);
    if (getenv("HMSG_DEBUG_TIMING") && m.ops.stat_calls > 0)
        fprintf(stderr, "[hmsg merge] dbscan segments with a cropped anchor member: %.0f (%.0f anchor points a batch), %.0f of them in place; pool %.2f GB\n",
                m.ops.stat_forced, m.ops.stat_forced_first / m.ops.stat_calls, m.ops.stat_inplace, (double)m.pool_used * 24 / 1e9);
    if (getenv("HMSG_DEBUG_MAXCELL") && m.ops.stat_calls > 0)
        fprintf(stderr, "[hmsg merge] fullest cell of a dbscan batch: mean %.0f points, max %.0f\n", m.ops.stat_maxcell_sum / m.ops.stat_calls, m.ops.stat_maxcell_max);
}

// ---- The sequential fold (graph_utils.py:1015-1038) as a RESUMABLE object: frames are handed in as their 3-D masks
// become available, step() folds the next one.  hmsg_merge drives it in one go; the pipelined path (FoldPipe below)
// drives it from a worker thread while hmsg_fuse_frames is still producing masks.
//   The BATCH fold (re-cluster every touched cloud in full each step) is the cheaper one while the clouds are small; its
//   step grows with the clouds, the INCREMENTAL fold's (hmsg_fold.inl) with the new points only: the fold switches over
//   when the batches' running mean passes `switch_points` points -- and only once every frame has arrived, because the
//   switch indexes the masks of the frames still to come.
//     HMSG_FOLD_LEGACY=1: batch fold throughout;  HMSG_FOLD_INCREMENTAL=1: incremental from the first step;
//     HMSG_FOLD_SWITCH=<points>: the threshold (default 1500000.  Round 3: 600000; with round 4's batch step the merge of the
//     10 000-frame 1280x720 episode takes 4.99 / 4.93 / 4.89 / 4.90 s switching at 0.6 / 1.0 / 1.5 / 3.0 million points and
//     5.46 s without the switch).
//   The instances are identical either way.
struct SeqFold {
    Folder m;
    hmsg_ctx* h = nullptr;
    std::vector<std::vector<Cloud>> frames;    // frame f's non-empty masks (emptied when the frame is folded)
    size_t f_next = 0;                         // next frame to fold
    size_t prebuilt_upto = 0;                  // frames below it have their overlap grids (unless a collection dropped them)
    std::vector<Cloud> G;
    bool incremental = false, never = false;
    int f_switch = -1;
    double batch_mean = 0.0, switch_points = 1500000.0;

    void init(hmsg_ctx* hh, hipStream_t stream, Prof* prof) {
        h = hh;
        merger_init(m, h);
        m.s = stream;
        m.ops.s = stream;
        m.ops.prof = prof;
        if (const char* e = getenv("HMSG_DEBUG_GC_POINTS")) m.gc_pool_points = (size_t)atoll(e);   // (tests: force collections)
        never = getenv("HMSG_FOLD_LEGACY") != nullptr;
        switch_points = getenv("HMSG_FOLD_INCREMENTAL") ? 0.0 : 1500000.0;
        if (const char* e = getenv("HMSG_FOLD_SWITCH")) switch_points = atof(e);
        if (const char* e = getenv("HMSG_FOLD_BIG_ACTIVE")) m.big_active = atoll(e);   // (development: which components take the batch kernels)
    }
    // the 3-D masks of `nm.size()` more frames: npts points at src (device; complete on the fold's stream or synchronised),
    // off[i] .. off[i + 1] = points of mask i (relative to src), nm[f] = masks of frame f.  reserve: pool room per point.
    void ingest(const double* src, long long npts, const long long* off, const std::vector<int>& nm, long long reserve) {
        if (m.pool.n == 0) {
            m.pool.alloc((size_t)std::max<long long>(npts * reserve, 1 << 16) * 3);
            m.poolcore.alloc((size_t)std::max<long long>(npts * reserve, 1 << 16));
        } else {
            m.grow(m.pool, (size_t)m.pool_used * 3, (size_t)(m.pool_used + npts) * 3);
            m.grow(m.poolcore, (size_t)m.pool_used, (size_t)(m.pool_used + npts));
        }
        const long long base = m.pool_used;
        if (npts) HIP_TRY(hipMemcpyAsync(m.pool.p + (size_t)base * 3, src, (size_t)npts * 24, hipMemcpyDeviceToDevice, m.s));
        m.pool_used += npts;
        size_t nmask = 0;
        for (int v : nm) nmask += (size_t)v;
        std::vector<SegDesc> msegs(nmask);
        for (size_t id = 0; id < nmask; ++id) {
            msegs[id].pt_base = base + (off[id] - off[0]);
            msegs[id].n = (int)(off[id + 1] - off[id]);
        }
        m.ops.bounds(m.pool.p, msegs);           // AABBs of the masks (device reduction)
        // Empty masks are left out: an empty cloud never pairs (find_overlapping_ratio_faiss returns 0 for it), so it
        // stays a singleton through every step and is dropped by the min-points filter at the end (graph.py:445-448).
        size_t id = 0;
        for (int v : nm) {
            frames.emplace_back();
            auto& fr = frames.back();
            fr.reserve((size_t)v);
            for (int i = 0; i < v; ++i, ++id) {
                const SegDesc& sd = msegs[id];
                if (sd.n == 0) continue;
                fr.emplace_back();
                Cloud& k = fr.back();
                k.off = sd.pt_base;
                k.n = sd.n;
                k.raw = true;
                k.uid = m.next_uid++;
                for (int a = 0; a < 3; ++a) {
                    k.mn[a] = sd.mn[a];
                    k.mx[a] = sd.mx[a];
                }
            }
        }
    }
    bool pending() const { return f_next < frames.size(); }
    // fold the next frame; input_complete: no further ingest() will follow
    void step(bool input_complete) {
        const hmsg_config& c = h->cfg;
        const size_t f = f_next++;
        if (f == 0) {
            G = std::move(frames[0]);
            std::vector<Cloud>().swap(frames[0]);
            return;
        }
        if (!incremental && !never) {
            // (a single step's batch jumps around: the running mean over ~64 steps decides)
            const double last_batch = m.ops.stat_points - m.stat_points_seen;
            m.stat_points_seen = m.ops.stat_points;
            batch_mean += (last_batch - batch_mean) / 64.0;
            if (input_complete && (switch_points <= 0.0 || batch_mean > switch_points) && fold_begin(m, h, G, frames, f)) {
                incremental = true;
                f_switch = (int)f;
            }
        }
        if (incremental) {
            G.insert(G.end(), frames[f].begin(), frames[f].end());
            std::vector<Cloud>().swap(frames[f]);
            G = m.fold_step(std::move(G), c.init_overlap_thresh);
            return;
        }
        // overlap grids of the frame masks are built ahead, a window at a time (inputs of the fold: only clouds that
        // change during the fold get a new grid later); a collection of the pool drops every grid
        if (m.needs_collect()) {
            m.collect(G, frames, f);
            prebuilt_upto = f;
        }
        if (f >= prebuilt_upto) {
            const size_t b = std::min(frames.size(), f + Merger::PREBUILD_WINDOW);
            m.prebuild(frames, f, b);
            prebuilt_upto = b;
        }
        G.insert(G.end(), frames[f].begin(), frames[f].end());
        std::vector<Cloud>().swap(frames[f]);
        G = m.merge_3d_masks(std::move(G), c.init_overlap_thresh);
    }
    // the last pass (graph_utils.py:1033-1037), the small-cloud drop (graph.py:445-448), instances into the handle
    void finish() {
        const hmsg_config& c = h->cfg;
        std::vector<Cloud> result = incremental ? m.fold_step(std::move(G), c.init_overlap_thresh) : m.merge_3d_masks(std::move(G), c.init_overlap_thresh);
        if (getenv("HMSG_DEBUG_TIMING") && incremental) {
            fprintf(stderr, "[hmsg merge] incremental fold from frame %d of %zu\n", f_switch, frames.size());
            fold_report(m, h);
        }
        store_instances(m, h, result, c.min_instance_points);
        merge_report(m);
    }
};

// ---- The fold runs BESIDE the fusion.  seq_merge consumes the frames' 3-D masks in frame order, hmsg_fuse_frames
// produces them 64 frames at a time, and the fold is a chain of ~25 small latency-bound launches per frame that leaves
// the GPU almost empty -- so a worker thread with its own (high-priority) stream starts folding as soon as the first
// batch of masks exists, while the fusion's wide streaming kernels keep filling the chip.  hmsg_merge_instances then only
// waits for the rest.  Same sequence of merge_3d_masks calls, same instances (tests/test_fold_pipeline.py);
// HMSG_FOLD_NOPIPE=1 runs the fold inside hmsg_merge_instances as before.
// The worker has its own thread-local allocator cache (no block ever moves between the two streams while in flight), its
// own timing list, and owns every buffer of the fold; it gets each batch as a private copy of the batch's mask points.
struct FoldPipe {
    struct Batch {
        DevBuf<double> pts;
        long long npts = 0;
        std::vector<long long> off;
        std::vector<int> nm;
    };
    hmsg_ctx* h = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::unique_ptr<Batch>> q;
    bool closed = false, aborted = false, failed = false;
    hmsg_error err{HMSG_OK, ""};
    Prof prof;
    std::vector<std::unique_ptr<Batch>> spent;   // consumed batches: their buffers go back through the thread that made them

    void run() {
        hipStream_t s = nullptr;
        // the fold's scratch lives in a cache of its own that stays with the handle from scene to scene (a service
        // rebuilds scenes over and over: only the first fold pays for its hipMallocs), apart from the calling thread's
        dev_cache().swap_state(h->fold_cache);
        try {
            HIP_TRY(hipSetDevice(h->cfg.device_id));
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            HIP_TRY(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
            {
                SeqFold sf;
                sf.init(h, s, &prof);
                for (;;) {
                    std::unique_ptr<Batch> b;
                    bool have = false, done = false, complete = false;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        if (!sf.pending()) cv.wait(lk, [&] { return !q.empty() || closed || aborted; });
                        if (aborted) break;
                        if (!q.empty()) {
                            b = std::move(q.front());
                            q.pop_front();
                            have = true;
                        }
                        complete = closed && q.empty();
                        done = complete && !have && !sf.pending();
                    }
                    if (have) {
                        sf.ingest(b->pts.p, b->npts, b->off.data(), b->nm, 64);
                        HIP_TRY(hipStreamSynchronize(s));
                        std::lock_guard<std::mutex> lk(mu);
                        spent.push_back(std::move(b));
                        continue;
                    }
                    if (done) {
                        if (!sf.frames.empty()) sf.finish();
                        break;
                    }
                    sf.step(complete);
                }
                HIP_TRY(hipStreamSynchronize(s));
            }
        } catch (const hmsg_error& e) {
            fail(e);
        } catch (const std::exception& e) {
            fail(hmsg_error{HMSG_ERR_HIP, e.what()});
        } catch (...) {
            fail(hmsg_error{HMSG_ERR_HIP, "merge fold worker: unknown exception"});
        }
        if (s) {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
        h->fold_cache.swap_state(dev_cache());
    }
    void fail(const hmsg_error& e) {
        std::lock_guard<std::mutex> lk(mu);
        failed = true;
        err = e;
    }
    void push(std::unique_ptr<Batch> b) {
        std::vector<std::unique_ptr<Batch>> done;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!failed) q.push_back(std::move(b));      // (a worker that gave up consumes nothing: hmsg_merge_instances reports its error)
            done.swap(spent);
        }
        cv.notify_one();
    }
    // stop == false: let the worker fold everything it was given and store the instances
    void join(bool stop) {
        {
            std::lock_guard<std::mutex> lk(mu);
            closed = true;
            aborted = aborted || stop;
        }
        cv.notify_one();
        if (th.joinable()) th.join();
        spent.clear();
        q.clear();
    }
};

static bool fold_pipe_wanted(const hmsg_ctx* h) {
    // (not beside a very large frame store -- the 157 GB of a 10 000-frame 1280x720 episode: the fold's arenas then come
    //  from fresh hipMallocs of tens of GB, each of which waits for the other thread's device work and stalls it in turn;
    //  measured 18.7 s instead of 15.5 s for that episode.  Such a store is handed back before the merge instead.)
    if (h->rgb.bytes() + h->depth.bytes() + h->bits.bytes() + h->nn.bytes() >= ((size_t)96 << 30)) return false;
    return h->cfg.merge_type != HMSG_MERGE_HIERARCHICAL && h->frame_window == 0 && !getenv("HMSG_FOLD_NOPIPE") &&
           !getenv("HMSG_FOLD_INCREMENTAL") && !getenv("HMSG_FOLD_SWITCH") && !getenv("HMSG_DEBUG_GC_POINTS");
}

// hmsg_fuse_frames finished a batch: frames [f0, f0 + nfr) have their 3-D masks (stream synchronised)
void hmsg_fold_pipe_feed(hmsg_ctx* h, int f0, int nfr) {
    if (h->merged || !fold_pipe_wanted(h)) return;
    if (!h->fold_pipe) {
        if (f0 != 0) return;                     // (fused before without a pipe: the fold runs in hmsg_merge_instances)
        h->fold_pipe = std::make_shared<FoldPipe>();
        h->fold_pipe->h = h;
        h->fold_pipe->prof.enabled = h->prof.enabled;
        FoldPipe* fp = h->fold_pipe.get();
        fp->th = std::thread([fp] { fp->run(); });
    }
    std::unique_ptr<FoldPipe::Batch> bp(new FoldPipe::Batch());
    FoldPipe::Batch& b = *bp;
    const size_t m0 = (size_t)h->mask_first[(size_t)f0], m1 = (size_t)h->mask_first[(size_t)f0 + nfr];
    b.off.assign(h->masks3d.off.begin() + (long)m0, h->masks3d.off.begin() + (long)m1 + 1);
    for (int f = f0; f < f0 + nfr; ++f) b.nm.push_back((int)(h->mask_first[(size_t)f + 1] - h->mask_first[(size_t)f]));
    b.npts = b.off.back() - b.off.front();
    b.pts.alloc((size_t)std::max<long long>(b.npts, 1) * 3);
    if (b.npts) {
        HIP_TRY(hipMemcpyAsync(b.pts.p, h->masks3d.pts.p + (size_t)b.off.front() * 3, (size_t)b.npts * 24, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    h->fold_pipe_frames = f0 + nfr;
    h->fold_pipe->push(std::move(bp));
}

// any call that invalidates a running pipe (reset, destroy, frame windows, the merge tree)
void hmsg_fold_pipe_abort(hmsg_ctx* h) {
    if (!h->fold_pipe) return;
    h->fold_pipe->join(true);
    h->fold_pipe->prof.clear();
    h->fold_pipe.reset();
    h->fold_pipe_frames = 0;
}

void hmsg_merge(hmsg_ctx* h) {
    const hmsg_config& c = h->cfg;
    HMSG_REQUIRE(h->feats_final && h->n_fused > 0, HMSG_ERR_INVALID, "hmsg_merge_instances: run hmsg_fuse_frames first");
    HMSG_REQUIRE(!h->merged, HMSG_ERR_INVALID, "instances already merged");
    HMSG_REQUIRE(h->frame_window == 0, HMSG_ERR_INVALID, "hmsg_merge_instances on a frame window: use hmsg_merge_tree_local / _join");
    const int F = h->n_fused;
    if (h->fold_pipe && h->fold_pipe_frames == F && fold_pipe_wanted(h)) {
        // the fold has been running beside the fusion: wait for the rest
        std::shared_ptr<FoldPipe> fp = h->fold_pipe;
        fp->join(false);
        h->fold_pipe.reset();
        h->fold_pipe_frames = 0;
        for (auto& e : fp->prof.ev) h->prof.ev.push_back(e);      // (events stay valid across threads)
        fp->prof.ev.clear();
        if (fp->failed) throw fp->err;
        h->merged = true;
        return;
    }
    hmsg_fold_pipe_abort(h);
    if (c.merge_type == HMSG_MERGE_HIERARCHICAL) {
        Folder m;
        merger_init(m, h);
        std::vector<std::vector<Cloud>> frames = seed_frames(m, h, 0);
        // graph_utils.py:959-1012
        m.use_cache = true;
        double th = c.init_overlap_thresh;
        std::vector<std::vector<Cloud>> lv = std::move(frames);
        while (lv.size() > 1) {
            std::vector<std::vector<Cloud>> nx;
            for (size_t i = 0; i < lv.size(); i += 2) {
                if (i == lv.size() - 1) {
                    nx.push_back(std::move(lv[i]));
                    break;
                }
                std::vector<Cloud> L = std::move(lv[i]);
                L.insert(L.end(), lv[i + 1].begin(), lv[i + 1].end());
                nx.push_back(m.merge_3d_masks(std::move(L), th));
            }
            lv = std::move(nx);
            if (lv.size() > 1) th = next_level_threshold(th, c.overlap_thresh_factor, (long long)lv.size());
        }
        std::vector<Cloud> result = m.merge_3d_masks(std::move(lv[0]), 0.75);
        // graph.py:445-448: drop clouds with < 10 points; compact the survivors into the handle
        store_instances(m, h, result, c.min_instance_points);
        merge_report(m);
    } else {
        SeqFold sf;
        sf.init(h, h->stream, &h->prof);
        std::vector<int> nm((size_t)F);
        for (int f = 0; f < F; ++f) nm[(size_t)f] = (int)(h->mask_first[(size_t)f + 1] - h->mask_first[(size_t)f]);
        sf.ingest(h->masks3d.pts.p, h->masks3d.total, h->masks3d.off.data(), nm, 2);
        while (sf.pending()) sf.step(true);
        sf.finish();
    }
    h->merged = true;
}

// ---- hierarchical_merge (graph_utils.py:989-1012) sharded over the frames --------------------------
// The merge tree pairs ADJACENT lists level by level, so a handle that holds the frames [first, first + n) of an
// episode of `total_frames` frames -- first a multiple of a power of two >= n -- owns a whole subtree: it reduces its
// frames with the thresholds the GLOBAL tree has at those levels and stops when its list's partner lives on another
// handle.  hmsg_merge_tree_join then plays one cross-handle level: [mine ++ theirs] through merge_3d_masks, and on the
// last level the final pass (threshold 0.75, :1007-1011) and the small-cloud drop of graph.py:445-448.
void hmsg_merge_tree_local_impl(hmsg_ctx* h, int total_frames, double* th_next, long long* lists_now, long long* my_index) {
    const hmsg_config& c = h->cfg;
    HMSG_REQUIRE(h->feats_final && h->n_fused > h->frame_window, HMSG_ERR_INVALID, "hmsg_merge_tree_local: run hmsg_fuse_frames first");
    HMSG_REQUIRE(!h->merged, HMSG_ERR_INVALID, "instances already merged");
    HMSG_REQUIRE(total_frames >= h->n_fused, HMSG_ERR_INVALID, "hmsg_merge_tree_local: total_frames smaller than the window's end");
    Merger m;
    merger_init(m, h);
    m.use_cache = true;
    std::vector<std::vector<Cloud>> lv = seed_frames(m, h, h->frame_window);
    double th = c.init_overlap_thresh;
    long long lists = total_frames, off = h->frame_window;     // global list count / global index of my first list
    while (lists > 1) {
        const long long n = (long long)lv.size();
        // local level: my lists pair among themselves (an odd last one only when it is the global last list).  A single
        // list has nobody to pair with here: whether it is carried up or meets a partner is the cross-handle levels'
        // business (every handle then stops at a level the others can work out, whatever the window lengths).
        if (n == 1 || (off & 1) || ((n & 1) && off + n < lists)) break;
        std::vector<std::vector<Cloud>> nx;
        for (size_t i = 0; i < lv.size(); i += 2) {
            if (i == lv.size() - 1) {
                nx.push_back(std::move(lv[i]));
                break;
            }
            std::vector<Cloud> L = std::move(lv[i]);
            L.insert(L.end(), lv[i + 1].begin(), lv[i + 1].end());
            nx.push_back(m.merge_3d_masks(std::move(L), th));
        }
        lv = std::move(nx);
        off >>= 1;
        lists = (lists + 1) / 2;
        if (lists > 1) th = next_level_threshold(th, c.overlap_thresh_factor, lists);
    }
    HMSG_REQUIRE(lv.size() == 1, HMSG_ERR_UNSUPPORTED,
                 "hmsg_merge_tree_local: the frame window is not a subtree of the merge tree (first frame must be a multiple of a "
                 "power of two >= the window length)");
    store_instances(m, h, lv[0], 0);
    *th_next = th;
    *lists_now = lists;
    *my_index = off;
    h->tree_partial = true;
}

void hmsg_merge_tree_join_impl(hmsg_ctx* h, int n_ext, const long long* ext_sizes, const double* ext_pts, double th, int final_pass) {
    const hmsg_config& c = h->cfg;
    HMSG_REQUIRE(h->tree_partial, HMSG_ERR_INVALID, "hmsg_merge_tree_join: run hmsg_merge_tree_local first");
    HMSG_REQUIRE(n_ext >= 0 && (n_ext == 0 || (ext_sizes && ext_pts)), HMSG_ERR_INVALID, "hmsg_merge_tree_join: bad argument");
    Merger m;
    merger_init(m, h);
    m.use_cache = true;
    long long ext_total = 0;
    for (int k = 0; k < n_ext; ++k) ext_total += ext_sizes[k];
    const long long own = h->inst.total, total = own + ext_total;
    m.pool.alloc((size_t)std::max<long long>(total * 2, 1 << 16) * 3);
    m.poolcore.alloc((size_t)std::max<long long>(total * 2, 1 << 16));
    if (own) HIP_TRY(hipMemcpyAsync(m.pool.p, h->inst.pts.p, (size_t)own * 24, hipMemcpyDeviceToDevice, h->stream));
    if (ext_total) {                               // (the partner's clouds: host memory, or device memory straight from a collective)
        hipPointerAttribute_t pa;
        memset(&pa, 0, sizeof(pa));
        const bool on_dev = hipPointerGetAttributes(&pa, ext_pts) == hipSuccess && pa.type == hipMemoryTypeDevice;
        if (!on_dev) (void)hipGetLastError();
        HIP_TRY(hipMemcpyAsync(m.pool.p + (size_t)own * 3, ext_pts, (size_t)ext_total * 24, on_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    }
    m.pool_used = total;
    const int n_own = (int)h->inst.off.size() - 1;
    std::vector<SegDesc> segs((size_t)(n_own + n_ext));
    for (int k = 0; k < n_own; ++k) {
        segs[(size_t)k].pt_base = h->inst.off[(size_t)k];
        segs[(size_t)k].n = (int)(h->inst.off[(size_t)k + 1] - h->inst.off[(size_t)k]);
    }
    long long at = own;
    for (int k = 0; k < n_ext; ++k) {
        segs[(size_t)(n_own + k)].pt_base = at;
        segs[(size_t)(n_own + k)].n = (int)ext_sizes[k];
        at += ext_sizes[k];
    }
    m.ops.bounds(m.pool.p, segs);
    std::vector<Cloud> L;                      // [my list ++ the partner's list]: I hold the even-indexed list of the pair
    for (auto& sd : segs) {
        if (sd.n == 0) continue;
        Cloud k;
        k.off = sd.pt_base;
        k.n = sd.n;
        k.uid = m.next_uid++;
        for (int a = 0; a < 3; ++a) {
            k.mn[a] = sd.mn[a];
            k.mx[a] = sd.mx[a];
        }
        L.push_back(k);
    }
    std::vector<Cloud> result = n_ext > 0 ? m.merge_3d_masks(std::move(L), th) : std::move(L);
    if (final_pass) result = m.merge_3d_masks(std::move(result), 0.75);
    store_instances(m, h, result, final_pass ? c.min_instance_points : 0);
    if (final_pass) {
        h->merged = true;
        h->tree_partial = false;
    }
}

// A10 first step (graph.py:1589-1591): every instance re-denoised with pcd_denoise_dbscan(eps, min_points),
// in place (the pooled features were computed before, as in the reference).
void hmsg_denoise_inst(hmsg_ctx* h, double eps, int min_points) {
    HMSG_REQUIRE(h->merged, HMSG_ERR_INVALID, "hmsg_denoise_instances: run hmsg_merge_instances first");
    const int K = (int)h->inst.off.size() - 1;
    if (K <= 0 || h->inst.total == 0) return;
    CloudOps ops;
    ops.s = h->stream;
    std::vector<SegDesc> segs(K);
    for (int k = 0; k < K; ++k) {
        segs[k].pt_base = h->inst.off[k];
        segs[k].n = (int)(h->inst.off[k + 1] - h->inst.off[k]);
    }
    ops.bounds(h->inst.pts.p, segs);
    DevBuf<double> out;
    out.alloc((size_t)h->inst.total * 3);
    std::vector<DbscanResult> res;
    long long total = ops.dbscan_keep_largest(h->inst.pts.p, segs, eps, min_points, out.p, res);
    out.swap(h->inst.pts);
    h->inst.off.assign(1, 0);
    h->inst.box.clear();
    long long acc = 0;
    for (int k = 0; k < K; ++k) {
        acc += res[k].n_out;
        h->inst.off.push_back(acc);
        for (int a = 0; a < 3; ++a) h->inst.box.push_back(res[k].mn[a]);
        for (int a = 0; a < 3; ++a) h->inst.box.push_back(res[k].mx[a]);
    }
    h->inst.total = total;
    HIP_TRY(hipStreamSynchronize(h->stream));
}

// ------------------------------------------------------------------------------------------ A10: object -> room share
// segment_hmsg_objects (graph.py:1634-1642) + find_intersection_share (utils/graph_utils.py:160-189): for an
// object and a room, the number of room vertices (2-D, x/z) that have an object point within `radius`, divided
// by the number of object points.  Device version: per-instance 2-D grids (cell = radius) over the instance's
// (x, z) points, one task per (instance, room) whose boxes come within `radius`, float64 distances.
struct ShGrid {
    long long pt_off;        // instance points in inst.pts
    long long cell_off;      // cellstart (ncell + 1) in the concatenated array
    int n, gx, gz, pad;
    double ox, oz, cell;
};
struct ShTask {
    int inst, room;
};
__device__ __forceinline__ long long sh_cell(const ShGrid& g, double x, double z) {
    int ix = (int)floor((x - g.ox) / g.cell), iz = (int)floor((z - g.oz) / g.cell);
    ix = min(max(ix, 0), g.gx - 1);
    iz = min(max(iz, 0), g.gz - 1);
    return g.cell_off + (long long)ix * g.gz + iz;
}
__global__ void k_sh_count(const double* __restrict__ pts, const ShGrid* __restrict__ gr, unsigned* __restrict__ cells) {
    const ShGrid g = gr[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += gridDim.x * blockDim.x) {
        const double* p = pts + (size_t)(g.pt_off + i) * 3;
        atomicAdd(&cells[sh_cell(g, p[0], p[2])], 1u);
    }
}
__global__ void k_sh_fill(const double* __restrict__ pts, const ShGrid* __restrict__ gr, const unsigned* __restrict__ cells,
                          unsigned* __restrict__ cursor, double* __restrict__ sorted /*[P][2]*/) {
    const ShGrid g = gr[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += gridDim.x * blockDim.x) {
        const double* p = pts + (size_t)(g.pt_off + i) * 3;
        long long c = sh_cell(g, p[0], p[2]);
        unsigned pos = cells[c] + atomicAdd(&cursor[c], 1u);
        sorted[(size_t)pos * 2] = p[0];
        sorted[(size_t)pos * 2 + 1] = p[2];
    }
}
__global__ void k_sh_query(const ShGrid* __restrict__ gr, const ShTask* __restrict__ tasks, const long long* __restrict__ vert_off,
                           const double* __restrict__ verts, const unsigned* __restrict__ cells, const double* __restrict__ sorted,
                           double r2, unsigned* __restrict__ counts) {
    const ShTask t = tasks[blockIdx.y];
    const ShGrid g = gr[t.inst];
    const long long v0 = vert_off[t.room], v1 = vert_off[t.room + 1];
    unsigned local = 0;
    for (long long v = v0 + blockIdx.x * blockDim.x + threadIdx.x; v < v1; v += (long long)gridDim.x * blockDim.x) {
        const double x = verts[v * 2], z = verts[v * 2 + 1];
        int cx = (int)floor((x - g.ox) / g.cell), cz = (int)floor((z - g.oz) / g.cell);
        if (cx < -1 || cz < -1 || cx > g.gx || cz > g.gz) continue;
        bool hit = false;
        // The vertex's own cell first, then the rest of its column, then the neighbour columns: inside an object's footprint the
        // witness is in the own cell, and a cell of a surface that has piled up its re-observations holds hundreds of points -- walked
        // to the end in vain when a neighbour comes first (the count does not care which point is the witness).
        auto scan = [&](unsigned k, unsigned k1) {
            for (; k < k1; ++k) {
                const double ddx = __dsub_rn(sorted[(size_t)k * 2], x), ddz = __dsub_rn(sorted[(size_t)k * 2 + 1], z);
                if (__dadd_rn(__dmul_rn(ddx, ddx), __dmul_rn(ddz, ddz)) < r2) return true;
            }
            return false;
        };
        const int z0 = max(cz - 1, 0), z1 = min(cz + 1, g.gz - 1);
        if (z1 >= z0) {
            const bool own = cx >= 0 && cx < g.gx && cz >= 0 && cz < g.gz;
            if (own) {
                const long long c0 = g.cell_off + (long long)cx * g.gz;
                hit = scan(cells[c0 + cz], cells[c0 + cz + 1]);
                if (!hit) hit = scan(cells[c0 + z0], cells[c0 + cz]);              // (below the own cell)
                if (!hit) hit = scan(cells[c0 + cz + 1], cells[c0 + z1 + 1]);      // (above it)
            }
            for (int q = own ? 1 : 0; q < 3 && !hit; ++q) {
                const int jx = cx + (q == 0 ? 0 : (q == 1 ? -1 : 1));
                if (jx < 0 || jx >= g.gx) continue;
                const long long c0 = g.cell_off + (long long)jx * g.gz;
                hit = scan(cells[c0 + z0], cells[c0 + z1 + 1]);
            }
        }
        local += hit ? 1u : 0u;
    }
    if (local) atomicAdd(&counts[blockIdx.y], local);
}

void hmsg_room_share(hmsg_ctx* h, int R, const long long* vert_off, const double* verts_xz, double radius, double* share_out) {
    HMSG_REQUIRE(h->merged, HMSG_ERR_INVALID, "hmsg_instance_room_share: run hmsg_merge_instances first");
    hipStream_t s = h->stream;
    const int N = (int)h->inst.off.size() - 1;
    for (long long i = 0; i < (long long)N * R; ++i) share_out[i] = 0.0;
    if (N <= 0 || R <= 0) return;
    const double cell = radius * (1.0 + 1e-9);
    // room vertex boxes (host)
    std::vector<double> rb((size_t)R * 4);
    for (int r = 0; r < R; ++r) {
        double x0 = 1e300, x1 = -1e300, z0 = 1e300, z1 = -1e300;
        for (long long v = vert_off[r]; v < vert_off[r + 1]; ++v) {
            x0 = std::min(x0, verts_xz[v * 2]);
            x1 = std::max(x1, verts_xz[v * 2]);
            z0 = std::min(z0, verts_xz[v * 2 + 1]);
            z1 = std::max(z1, verts_xz[v * 2 + 1]);
        }
        rb[(size_t)r * 4] = x0; rb[(size_t)r * 4 + 1] = x1; rb[(size_t)r * 4 + 2] = z0; rb[(size_t)r * 4 + 3] = z1;
    }
    std::vector<ShGrid> g(N);
    std::vector<ShTask> tasks;
    long long ncell = 0;
    int maxn = 0;
    for (int i = 0; i < N; ++i) {
        ShGrid& q = g[i];
        q.pt_off = h->inst.off[i];
        q.n = (int)(h->inst.off[i + 1] - h->inst.off[i]);
        q.cell = cell;
        q.pad = 0;
        q.cell_off = ncell;
        const double* bx = &h->inst.box[(size_t)i * 6];
        q.ox = bx[0] - 1e-9;
        q.oz = bx[2] - 1e-9;
        q.gx = q.n ? (int)std::floor((bx[3] - q.ox) / cell) + 1 : 1;
        q.gz = q.n ? (int)std::floor((bx[5] - q.oz) / cell) + 1 : 1;
        ncell += (long long)q.gx * q.gz + 1;
        maxn = std::max(maxn, q.n);
        if (!q.n) continue;
        for (int r = 0; r < R; ++r)
            if (!(rb[(size_t)r * 4] > bx[3] + radius || rb[(size_t)r * 4 + 1] < bx[0] - radius ||
                  rb[(size_t)r * 4 + 2] > bx[5] + radius || rb[(size_t)r * 4 + 3] < bx[2] - radius))
                tasks.push_back(ShTask{i, r});
    }
    if (tasks.empty()) return;
    HMSG_REQUIRE(ncell < (1ll << 31), HMSG_ERR_UNSUPPORTED, "room-share grids too large");
    DevBuf<ShGrid> d_g;
    DevBuf<ShTask> d_t;
    DevBuf<unsigned> cells, cursor, counts;
    DevBuf<double> sorted, d_verts;
    DevBuf<long long> d_voff;
    DevBuf<unsigned> scan_tmp;
    d_g.alloc(N);
    d_t.alloc(tasks.size());
    cells.alloc((size_t)ncell);
    cursor.alloc((size_t)ncell);
    counts.alloc(tasks.size());
    sorted.alloc((size_t)std::max<long long>(h->inst.total, 1) * 2);
    const long long nv = vert_off[R];
    d_verts.alloc((size_t)std::max<long long>(nv, 1) * 2);
    d_voff.alloc(R + 1);
    HIP_TRY(hipMemcpyAsync(d_g.p, g.data(), (size_t)N * sizeof(ShGrid), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_t.p, tasks.data(), tasks.size() * sizeof(ShTask), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_verts.p, verts_xz, (size_t)nv * 16, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_voff.p, vert_off, (size_t)(R + 1) * 8, hipMemcpyHostToDevice, s));
    cells.zero(s);
    cursor.zero(s);
    counts.zero(s);
    dim3 grid(std::max(1u, std::min(cdiv(maxn, 256), 256u)), N);
    hipLaunchKernelGGL(k_sh_count, grid, dim3(256), 0, s, (const double*)h->inst.pts.p, (const ShGrid*)d_g.p, cells.p);
    HMSG_CHECK_LAUNCH();
    hmsg_scan_u32(cells.p, cells.p, (size_t)ncell, s, scan_tmp, nullptr);
    hipLaunchKernelGGL(k_sh_fill, grid, dim3(256), 0, s, (const double*)h->inst.pts.p, (const ShGrid*)d_g.p, (const unsigned*)cells.p,
                       cursor.p, sorted.p);
    for (size_t t0 = 0; t0 < tasks.size(); t0 += 32768) {
        unsigned nt = (unsigned)std::min<size_t>(32768, tasks.size() - t0);
        hipLaunchKernelGGL(k_sh_query, dim3(16, nt), dim3(256), 0, s, (const ShGrid*)d_g.p, (const ShTask*)(d_t.p + t0),
                           (const long long*)d_voff.p, (const double*)d_verts.p, (const unsigned*)cells.p, (const double*)sorted.p,
                           radius * radius, counts.p + t0);
    }
    HMSG_CHECK_LAUNCH();
    std::vector<unsigned> hc(tasks.size());
    HIP_TRY(hipMemcpyAsync(hc.data(), counts.p, tasks.size() * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (size_t k = 0; k < tasks.size(); ++k)
        share_out[(size_t)tasks[k].inst * R + tasks[k].room] = (double)hc[k] / (double)g[tasks[k].inst].n;
}

// ------------------------------------------------------------------------------------------ A9: camera -> room
// compute_room_embeddings (utils/graph_utils.py:244-291): distance of a camera position (x, z) to a room =
// np.min(cdist([pos], room_points, "euclidean")) = sqrt of the smallest dx*dx + dy*dy (float64, that order).
// One workgroup per (query, set) pair.
__global__ void __launch_bounds__(256) k_min_dist_2d(const long long* __restrict__ set_off, const double* __restrict__ pts, int n_sets,
                                                     const double* __restrict__ q, double* __restrict__ out) {
    __shared__ double s_m[4];
    const int qi = blockIdx.y, si = blockIdx.x;
    const double qx = q[(size_t)qi * 2], qy = q[(size_t)qi * 2 + 1];
    double m = 1e308 * 10.0;     // +inf: an empty set answers inf like np.min would refuse to
    for (long long k = set_off[si] + threadIdx.x; k < set_off[si + 1]; k += blockDim.x) {
        const double dx = __dsub_rn(qx, pts[k * 2]), dy = __dsub_rn(qy, pts[k * 2 + 1]);
        const double d2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
        m = d2 < m ? d2 : m;
    }
    m = wave_min_f64(m);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) m = s_m[w] < m ? s_m[w] : m;
        out[(size_t)qi * n_sets + si] = __dsqrt_rn(m);
    }
}

extern "C" int hmsg_points_min_dist_2d(int32_t device_id, int32_t n_sets, const int64_t* set_off, const double* pts_xy, int64_t n_q,
                                       const double* q_xy, double* out) {
    if (n_sets < 0 || n_q < 0 || !set_off || !out) return HMSG_ERR_INVALID;
    if (n_sets == 0 || n_q == 0) return HMSG_OK;
    try {
        HIP_TRY(hipSetDevice(device_id));
        const long long np = set_off[n_sets];
        DevBuf<long long> d_off;
        DevBuf<double> d_p, d_q, d_o;
        d_off.alloc((size_t)n_sets + 1);
        d_p.alloc((size_t)std::max<long long>(np, 1) * 2);
        d_q.alloc((size_t)n_q * 2);
        d_o.alloc((size_t)n_q * n_sets);
        HIP_TRY(hipMemcpy(d_off.p, set_off, ((size_t)n_sets + 1) * 8, hipMemcpyHostToDevice));
        if (np) HIP_TRY(hipMemcpy(d_p.p, pts_xy, (size_t)np * 16, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_q.p, q_xy, (size_t)n_q * 16, hipMemcpyHostToDevice));
        for (long long q0 = 0; q0 < n_q; q0 += 32768) {
            const unsigned nq = (unsigned)std::min<long long>(32768, n_q - q0);
            hipLaunchKernelGGL(k_min_dist_2d, dim3((unsigned)n_sets, nq), dim3(256), 0, 0, (const long long*)d_off.p, (const double*)d_p.p,
                               n_sets, (const double*)d_q.p + q0 * 2, d_o.p + q0 * n_sets);
        }
        HMSG_CHECK_LAUNCH();
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, d_o.p, (size_t)n_q * n_sets * 8, hipMemcpyDeviceToHost));
        return HMSG_OK;
    } catch (const hmsg_error& e) {
        fprintf(stderr, "hmsg_points_min_dist_2d: %s\n", e.msg.c_str());
        return e.code;
    }
}

// ------------------------------------------------------------------------------------------ A8 helper
// Open3D voxel_down_sample of one caller-supplied cloud (segment_floors_manually re-samples the finished map at
// 5 cm before it histograms the heights, graph.py:633): `pts` host or device, `out` host (capacity n points).
long long hmsg_voxel_ds(hmsg_ctx* h, const double* pts, long long n, double vs, double* out) {
    HMSG_REQUIRE(n >= 0 && n < (1ll << 31) && vs > 0, HMSG_ERR_INVALID, "hmsg_voxel_down_sample: bad argument");
    if (n == 0) return 0;
    hipStream_t s = h->stream;
    DevBuf<double> src, dst;
    src.alloc((size_t)n * 3);
    dst.alloc((size_t)n * 3);
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    const bool on_dev = hipPointerGetAttributes(&a, pts) == hipSuccess && a.type == hipMemoryTypeDevice;
    if (!on_dev) (void)hipGetLastError();
    HIP_TRY(hipMemcpyAsync(src.p, pts, (size_t)n * 24, on_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    CloudOps ops;
    ops.s = s;
    std::vector<SegDesc> segs(1);
    segs[0].pt_base = 0;
    segs[0].n = (int)n;
    ops.bounds(src.p, segs);
    std::vector<int> out_n;
    const long long total = ops.voxel_down_sample(src.p, segs, vs, dst.p, out_n);
    if (total) HIP_TRY(hipMemcpyAsync(out, dst.p, (size_t)total * 24, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return total;
}

