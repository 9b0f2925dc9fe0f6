// Segmented exact DBSCAN (keep-largest-cluster wrapper) and segmented voxel_down_sample.
//
// Reference semantics: Open3D 0.18 ClusterDBSCAN / VoxelDownSample as restated in
// oracle/hmsg_oracle.py (o3d_cluster_dbscan, o3d_voxel_down_sample) and the wrapper
// utils/graph_utils.py:827-880 (pcd_denoise_dbscan).
//
// DBSCAN design (grid-exact): cells of side eps/sqrt(3) -- any two points of one cell are neighbours, so
// a cell holding >= min_points points is all-core without a single distance test, and core points of one
// cell always share a cluster.  Clusters are found by a lock-free union-find over CORE CELLS (an edge
// needs one witness pair of core points closer than eps, early exit), which makes the cost independent
// of how many duplicates a heavily re-observed surface has piled up.  Cluster order (smallest core
// index), border assignment (first = smallest-order reaching cluster) and the largest-cluster tie rule
// (first label in point order) follow the reference exactly.
//
// Batch pipeline (one launch each): init tables -> bin points into cells -> scan -> cell-sorted copy -> core
// flags + cell lists (k_db_core) -> per-cell boxes / core counts -> [anchor cells pre-connected] -> box pass and
// witness scans over the ACTIVE cells -> roots, cluster keys and core sizes -> border points (wave per point)
// -> largest cluster per segment -> keep flags -> scan -> compaction + boxes of the kept points.  The
// neighbourhood kernels are serial chains of L2 round trips per lane: they work on a cell-SORTED COPY of the
// points (one load per candidate), visit the nearest cells first and keep several loads in flight.
#include "hmsg_cloudops.h"

#include <algorithm>
#include <cmath>

struct DbSeg {
    double ox, oy, oz, cs;
    int nx, ny, nz, n;
    long long cell_base, pt_base;
    int n_first, forced;    // (SegDesc::n_first, SegDesc::forced)
    double cmn[3], cmx[3];  // (SegDesc::cmn / cmx: the crop of a forced segment's first member)
    long long out_off;      // (SegDesc::out_off, out_mode)
    int out_mode, pad;
};
#define DB_FAR (-1ll)       /* cellid of a point of a forced segment's first member outside the crop: not in the grid */
#define DB_FAR2 (-2ll)      /* ... of a segment that runs IN PLACE (out_mode 2): the point is nowhere in the batch's buffers */
// one workgroup of k_db_compact: points [p0, p0 + cnt) of the batch; tile `tile` of the look-back chain whose status words start at
// state[chain0].  Chain 0 = the mode-0 segments (dense output at dst); every segment with an output region of its own is a chain.
struct DbBlk {
    long long p0;
    int cnt, tile, chain0, seg;     // seg: the segment of an own-region chain (-1: chain 0)
};

#define INF32 0xffffffffu

__device__ __forceinline__ double dist2_f64(const double* __restrict__ a, const double* __restrict__ b) {
    double dx = __dsub_rn(a[0], b[0]), dy = __dsub_rn(a[1], b[1]), dz = __dsub_rn(a[2], b[2]);
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
}

#define DBC_LDS 512
__global__ void k_db_cell(const double* __restrict__ pts, long long N, int* __restrict__ segid, int K,
                          const DbSeg* __restrict__ segs, long long* __restrict__ cellid, unsigned* __restrict__ cnt,
                          DbGather ga, double* __restrict__ pts_out) {
    // the two binary searches below are chains of dependent loads (6 + 4 round trips per point when they go to L2): the keys they
    // search -- first batch position of every piece / of every segment -- come into LDS once per workgroup
    __shared__ long long s_dst[DBC_LDS], s_base[DBC_LDS];
    const bool lds_g = ga.segs && ga.nsegs <= DBC_LDS, lds_s = K <= DBC_LDS;
    if (lds_g)
        for (int q = threadIdx.x; q < ga.nsegs; q += blockDim.x) s_dst[q] = ga.segs[q].dst;
    if (lds_s)
        for (int q = threadIdx.x; q < K; q += blockDim.x) s_base[q] = segs[q].pt_base;
    __syncthreads();
    // (Measured on the MI355X, round 4: neither the two binary searches below -- their keys held in LDS instead: 19.0 -> 18.9 us --
    //  nor the atomics -- the fullest cell of a batch holds ~190 points, a 2 us chain -- bound this kernel; it moves 26.8 MB per
    //  launch (PMC: 8.4 fetched, 18.4 written, byte and 8-byte stores among them) at ~1.4 TB/s.)
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int lo = 0, hi = K - 1;                 // segment of the point (segments tile [0, N) in order)
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if ((lds_s ? s_base[mid] : segs[mid].pt_base) <= i) lo = mid; else hi = mid - 1;
    }
    const DbSeg sg = segs[lo];
    double px, py, pz;
    long long sp = -1;
    unsigned char c0 = 0;
    if (ga.segs) {                          // the batch is assembled here: point i comes from its piece of the pool
        int a = 0, b = ga.nsegs - 1;
        while (a < b) {
            const int mid = (a + b + 1) >> 1;
            if ((lds_g ? s_dst[mid] : ga.segs[mid].dst) <= i) a = mid; else b = mid - 1;
        }
        const CatSeg cs = ga.segs[a];
        sp = cs.src + (i - cs.dst);
        px = ga.pool[sp * 3];
        py = ga.pool[sp * 3 + 1];
        pz = ga.pool[sp * 3 + 2];
        c0 = cs.anchor ? (unsigned char)1 : (unsigned char)0;
    } else {
        px = pts[i * 3];
        py = pts[i * 3 + 1];
        pz = pts[i * 3 + 2];
    }
    const bool far = sg.forced && i < sg.pt_base + sg.n_first &&
                     (px < sg.cmn[0] || px > sg.cmx[0] || py < sg.cmn[1] || py > sg.cmx[1] || pz < sg.cmn[2] || pz > sg.cmx[2]);
    if (far && sg.out_mode == 2) {          // stays where it is: nothing of it enters the batch
        cellid[i] = DB_FAR2;
        return;
    }
    if (ga.segs) {
        pts_out[i * 3] = px;
        pts_out[i * 3 + 1] = py;
        pts_out[i * 3 + 2] = pz;
        if (ga.dstcore) ga.dstcore[i] = c0 ? ga.poolcore[sp] : (unsigned char)0;
    }
    segid[i] = lo;
    if (far) {
        cellid[i] = DB_FAR;
        return;
    }
    int ix = (int)floor((px - sg.ox) / sg.cs), iy = (int)floor((py - sg.oy) / sg.cs), iz = (int)floor((pz - sg.oz) / sg.cs);
    ix = ix < 0 ? 0 : (ix >= sg.nx ? sg.nx - 1 : ix);
    iy = iy < 0 ? 0 : (iy >= sg.ny ? sg.ny - 1 : iy);
    iz = iz < 0 ? 0 : (iz >= sg.nz ? sg.nz - 1 : iz);
    long long c = sg.cell_base + ((long long)ix * sg.ny + iy) * sg.nz + iz;
    cellid[i] = c;
    atomicAdd(&cnt[c], 1u);
}

// Cell starts AND the list of occupied cells in one launch (round 6).  The exclusive prefix sum of the cells' point counts is the
// decoupled look-back of k_scan_lookback (tiles of 1024 cells); the same tiles carry a second look-back -- run by the tile's second
// wave beside the first -- over "the cell holds a point", which gives every occupied cell its position in a compact list in cell
// order: `cells` (hence grouped by segment), cellpos[c], and the cell's entries of the tables the later passes work on (its own
// union-find node, its segment, no core point seen yet ...).  Until round 5 that list was built by the core points themselves
// (atomicMin on the cell's smallest core index, first one registers the cell, slots from an atomic counter per workgroup) in a
// launch of its own, k_db_core, after the neighbour counts; now a cell is listed whether it turns out to hold a core point or not
// -- the passes that walk the list skip a cell whose minidx is still INF (no core) or that is not active -- and the launch is gone.
// The per-cell tables the batch needs are initialised here too (one visit per cell): k_db_init only clears the counts.
struct DbCellTables {
    unsigned *cursor, *minidx, *firstidx, *rootmin, *size;
    unsigned char* hasanchor;
    int *cells, *cellpos, *parent, *cseg;
    unsigned* n_cells;                  // [0] occupied cells
};
#define DBS_LDS 512
__global__ void __launch_bounds__(256) k_db_scan(const unsigned* __restrict__ cnt, unsigned* __restrict__ start, long long NC /* cells; entry NC is the end sentinel */,
                                                 unsigned long long* __restrict__ state, unsigned long long* __restrict__ state_o, unsigned epoch,
                                                 const DbSeg* __restrict__ segs, int K, DbCellTables tb) {
    __shared__ unsigned wsum[4], wsum_o[4];
    __shared__ unsigned s_prefix, s_prefix_o;
    __shared__ long long s_cb[DBS_LDS];
    const bool lds_s = K <= DBS_LDS;
    if (lds_s)
        for (int q = threadIdx.x; q < K; q += blockDim.x) s_cb[q] = segs[q].cell_base;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long long tile = blockIdx.x;
    const long long base = (tile * 256 + tid) * 4;
    unsigned v[4];
    unsigned t = 0, to = 0;
    for (int i = 0; i < 4; ++i) {
        v[i] = base + i <= NC ? cnt[base + i] : 0u;
        t += v[i];
        to += v[i] ? 1u : 0u;
    }
    unsigned incl = t, incl_o = to;   // inclusive wave scans
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(incl, o), uo = __shfl_up(incl_o, o);
        if (lane >= o) {
            incl += u;
            incl_o += uo;
        }
    }
    if (lane == 63) {
        wsum[w] = incl;
        wsum_o[w] = incl_o;
    }
    __syncthreads();
    unsigned woff = 0, woff_o = 0;
    for (int i = 0; i < w; ++i) {
        woff += wsum[i];
        woff_o += wsum_o[i];
    }
    if (w == 0) {
        const unsigned prefix = scan_lookback_prefix(state, tile, epoch, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
        if (lane == 0) s_prefix = prefix;
    } else if (w == 1) {
        const unsigned prefix = scan_lookback_prefix(state_o, tile, epoch, wsum_o[0] + wsum_o[1] + wsum_o[2] + wsum_o[3]);
        if (lane == 0) s_prefix_o = prefix;
    }
    __syncthreads();
    unsigned excl = s_prefix + woff + incl - t, pos = s_prefix_o + woff_o + incl_o - to;
    for (int i = 0; i < 4; ++i) {
        const long long c = base + i;
        if (c <= NC) start[c] = excl;
        excl += v[i];
        if (c < NC) {
            if (!v[i]) tb.minidx[c] = INF32;    // (read for every neighbour cell: INF = no core point there, nothing else of it is looked at;
                                                //  an occupied cell's is written by k_db_cellbox, like its `active`)
            if (v[i]) {
                tb.cursor[c] = 0u;
                tb.firstidx[c] = INF32;
                tb.rootmin[c] = INF32;
                tb.size[c] = 0u;
                tb.hasanchor[c] = 0;
                int lo = 0, hi = K - 1;         // segment of the cell (the segments' cell ranges tile [0, NC) in order)
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if ((lds_s ? s_cb[mid] : segs[mid].cell_base) <= c) lo = mid; else hi = mid - 1;
                }
                tb.cells[pos] = (int)c;
                tb.cellpos[c] = (int)pos;
                tb.parent[c] = (int)c;
                tb.cseg[c] = lo;
                ++pos;
            }
        } else if (c == NC) {
            tb.n_cells[0] = pos;                // (the sentinel holds no point: its position is the number of occupied cells)
        }
    }
}

// Cell-sorted COPY of the points (+ each point's slot, and the point of each slot): every neighbourhood scan below walks
// contiguous runs of it.  Those scans are serial, latency-bound chains per lane (the kernel runs as long as its slowest lane), so
// a candidate must cost one load, not the ord -> point -> flag chain of an index sort.
// It also lists the points whose core status needs a neighbour COUNT (not an anchor core, cell holds fewer than
// min_points): k_db_count gives each of them a whole wave.
// Round 6: every OTHER point's status is known right here -- an anchor core, or a point of a cell that holds min_points points, is
// core; a point a cropped anchor leaves outside its crop keeps the flag it came with -- so this pass writes their flags (per point,
// and per slot of the sorted copy: DB_SLOT_ANCHOR / DB_SLOT_CORE, 0 = to be counted), and what the flags mean for the CELL is worked
// out by the wave that visits the cell anyway (k_db_cellbox) instead of by atomics from every point (until round 5: a separate
// k_db_core launch over all points behind the counts).
#define DB_SLOT_CORE 1
#define DB_SLOT_ANCHOR 2
__global__ void k_db_fill(const double* __restrict__ pts, long long N, const long long* __restrict__ cellid,
                          const unsigned* __restrict__ start, unsigned* __restrict__ cursor, unsigned* __restrict__ rank,
                          double* __restrict__ spts, const unsigned char* __restrict__ core0, int minpts,
                          unsigned* __restrict__ needy, unsigned* __restrict__ n_needy, unsigned* __restrict__ sidx,
                          unsigned char* __restrict__ core, unsigned char* __restrict__ score, unsigned char* __restrict__ hasanchor) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < N;
    bool need = false;
    if (live) {
        const long long c = cellid[i];
        if (c >= 0) {
            const unsigned s0 = start[c];
            unsigned p = s0 + atomicAdd(&cursor[c], 1u);
            rank[i] = p;
            sidx[p] = (unsigned)i;
            for (int a = 0; a < 3; ++a) spts[(size_t)p * 3 + a] = pts[(size_t)i * 3 + a];
            const bool known = core0 != nullptr && core0[i] != 0;
            need = !known && start[c + 1] - s0 < (unsigned)minpts;
            score[p] = known ? DB_SLOT_ANCHOR : (need ? 0 : DB_SLOT_CORE);
            if (!need) core[i] = 1;
            // cells holding anchor cores are all connected: db_anchor_cells (k_db_cellbox's launch) hangs them under the segment's lowest one
            if (known && !hasanchor[c]) hasanchor[c] = 1;
        } else if (c == DB_FAR) {
            // a forced segment's anchor point outside the crop: keeps its flag, takes part in nothing
            core[i] = (core0 != nullptr && core0[i] != 0) ? 1 : 0;
        }
        // (DB_FAR2: a point of an in-place segment outside the crop -- it is nowhere in the batch, no flag to write)
    }
    // (one atomic per wave that holds needy points.  Handing the slots out per 1024-thread block instead was measured on the
    //  MI355X: 18.8 -> 21.3 us per fold step.  This pass is not bound by that counter but by the per-point chains above.)
    const unsigned long long m = __ballot(need);
    if (m) {
        const int lane = threadIdx.x & 63, leader = __ffsll(m) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(n_needy, (unsigned)__popcll(m));
        base = __shfl(base, leader);
        if (need) needy[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (unsigned)i;
    }
}

// number of points of the sorted run [s, e) closer than eps to p, counted until `need` are found; four
// independent loads in flight per step
__device__ __forceinline__ int count_within(const double* __restrict__ spts, unsigned s, unsigned e, const double* __restrict__ p,
                                            double eps2, int have, int need) {
    for (unsigned k = s; k < e && have < need; k += 4) {
        double d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned kk = min(k + (unsigned)j, e - 1u);
            d[j] = dist2_f64(spts + (size_t)kk * 3, p);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) have += (k + (unsigned)j < e && d[j] < eps2) ? 1 : 0;
    }
    return have;
}
// is some CORE point of the sorted run [s, e) closer than eps to p?
__device__ __forceinline__ bool any_core_within(const double* __restrict__ spts, const unsigned char* __restrict__ score, unsigned s,
                                                unsigned e, const double* __restrict__ p, double eps2) {
    for (unsigned k = s; k < e; k += 4) {
        bool h = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned kk = min(k + (unsigned)j, e - 1u);
            const bool c = score[kk] != 0;
            const double d = dist2_f64(spts + (size_t)kk * 3, p);
            h = h || (k + (unsigned)j < e && c && d < eps2);
        }
        if (h) return true;
    }
    return false;
}

__device__ __forceinline__ void cell_xyz(const DbSeg& sg, long long c, int& ix, int& iy, int& iz) {
    long long l = c - sg.cell_base;
    iz = (int)(l % sg.nz);
    l /= sg.nz;
    iy = (int)(l % sg.ny);
    ix = (int)(l / sg.ny);
}

// z-cell range [z0, z1] of grid column (jx, jy) that can hold a point within eps of p (conservative: the x/y
// gaps to the column are shrunk by a slack far above the rounding of the cell arithmetic, and the z cells come
// from the same monotone floor((z - oz) / cs) that k_db_cell bins points with).  false: nothing in reach.
__device__ __forceinline__ bool column_zrange(const DbSeg& sg, const double* __restrict__ p, int jx, int jy, int iz,
                                              double eps2, int& z0, int& z1) {
    const double slack = 1e-9;
    const double x0 = sg.ox + (double)jx * sg.cs, y0 = sg.oy + (double)jy * sg.cs;
    const double gx = fmax(0.0, fmax(x0 - p[0], p[0] - (x0 + sg.cs)) - slack);
    const double gy = fmax(0.0, fmax(y0 - p[1], p[1] - (y0 + sg.cs)) - slack);
    const double rem = eps2 * (1.0 + 1e-9) - gx * gx - gy * gy;
    if (rem <= 0.0) return false;
    const double zr = sqrt(rem) + slack;
    const int a = (int)floor((p[2] - zr - sg.oz) / sg.cs), b = (int)floor((p[2] + zr - sg.oz) / sg.cs);
    z0 = max(max(iz - 2, 0), a);
    z1 = min(min(iz + 2, sg.nz - 1), b);
    return z0 <= z1;
}

// the 25 grid columns around a cell, nearest first: neighbour counts reach min_points, and border points find
// their witness, after far fewer candidates than in raster order
__device__ const signed char DB_COL[25][2] = {{0, 0},  {-1, 0}, {1, 0},  {0, -1},  {0, 1},  {-1, -1}, {-1, 1}, {1, -1}, {1, 1},
                                              {-2, 0}, {2, 0},  {0, -2}, {0, 2},   {-2, -1}, {-2, 1}, {2, -1}, {2, 1},  {-1, -2},
                                              {-1, 2}, {1, -2}, {1, 2},  {-2, -2}, {-2, 2},  {2, -2}, {2, 2}};

// Neighbour counts of the listed points, one WAVE per point: the 25 grid columns around the point's cell are 25
// contiguous ranges of the cell-sorted copy; lanes 0..24 look their range up side by side, the ranges are laid end
// to end (wave prefix sum) and the 64 lanes test 64 candidates per trip -- a handful of L2 round trips per point instead
// of the ~75 of one lane walking the columns one after the other.  core[i] = (neighbours within eps, the point itself
// included, >= min_points), the same predicate as before.
// The points that come out non-core are the entries of THIS list whose flag is 0 (k_db_rootmin compacts them for k_db_label).
__global__ void __launch_bounds__(256) k_db_count(const double* __restrict__ pts, const int* __restrict__ segid,
                                                  const DbSeg* __restrict__ segs, const long long* __restrict__ cellid,
                                                  const unsigned* __restrict__ start, const double* __restrict__ spts, double eps2,
                                                  int minpts, const unsigned* __restrict__ needy,
                                                  const unsigned* __restrict__ n_needy, unsigned char* __restrict__ core) {
    const int lane = threadIdx.x & 63;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6, n = *n_needy;
    for (unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n; w += nwaves) {
        const long long i = needy[w];
        const DbSeg sg = segs[segid[i]];
        int ix, iy, iz;
        cell_xyz(sg, cellid[i], ix, iy, iz);
        const double pi[3] = {pts[(size_t)i * 3], pts[(size_t)i * 3 + 1], pts[(size_t)i * 3 + 2]};
        unsigned s0 = 0, len = 0;
        if (lane < 25) {
            const int jx = ix + DB_COL[lane][0], jy = iy + DB_COL[lane][1];
            int z0, z1;
            if (jx >= 0 && jx < sg.nx && jy >= 0 && jy < sg.ny && column_zrange(sg, pi, jx, jy, iz, eps2, z0, z1)) {
                const long long cb = sg.cell_base + ((long long)jx * sg.ny + jy) * sg.nz;
                s0 = start[cb + z0];
                len = start[cb + z1 + 1] - s0;
            }
        }
        unsigned incl = len;                                  // inclusive prefix sum over the lanes
        for (int d = 1; d < 32; d <<= 1) {
            const unsigned up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        const unsigned total = __shfl(incl, 24);
        int have = 0;
        for (unsigned t0 = 0; t0 < total && have < minpts; t0 += 64) {
            const unsigned t = t0 + lane;
            int lo = 0, hi = 24;                              // column of candidate t: first lane with incl > t
            for (int it = 0; it < 5; ++it) {
                const int mid = (lo + hi) >> 1;
                const unsigned v = __shfl(incl, mid);
                if (v > t) hi = mid; else lo = mid + 1;
            }
            const int col = min(lo, 24);
            const unsigned c_incl = __shfl(incl, col), c_len = __shfl(len, col), c_s0 = __shfl(s0, col);
            bool hit = false;
            if (t < total) hit = dist2_f64(spts + (size_t)(c_s0 + (t - (c_incl - c_len))) * 3, pi) < eps2;
            have += __popcll(__ballot(hit));
        }
        if (lane == 0) core[i] = have >= minpts ? 1 : 0;
    }
}

// Anchor cells start out as ONE component per segment, in one launch: inside a wave the anchor cells of a segment hang
// under the wave's lowest one; the wave minima are chained through rep[segment] -- an atomicMin hands back the lowest
// cell seen so far: whichever of the two is higher hangs under the lower.  Every minimum is written by exactly one wave
// (the one that either brought it in above the current minimum, or took the minimum over from it), parents are always
// lower cells, and the chain ends at the segment's lowest anchor cell.
__device__ __forceinline__ void db_anchor_cells(const int* __restrict__ corecells, const unsigned* __restrict__ ncore, const int* __restrict__ cseg,
                            const unsigned char* __restrict__ hasanchor, unsigned* __restrict__ rep, int* __restrict__ parent) {
    const unsigned n = *ncore;
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned w0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); w0 < n; w0 += stride) {
        const unsigned w = w0 + (threadIdx.x & 63u);
        int seg = -1;
        unsigned c = INF32;
        if (w < n) {
            c = (unsigned)corecells[w];
            if (hasanchor[c]) seg = cseg[c];           // segment of the cell (written when the cell was registered)
        }
        unsigned long long todo = __ballot(seg >= 0);
        while (todo) {
            const int leader = __ffsll(todo) - 1;
            const int key = __shfl(seg, leader);
            const bool mine_b = seg == key;
            const unsigned long long mine = __ballot(mine_b);
            unsigned v = mine_b ? c : INF32;
            for (int o = 32; o > 0; o >>= 1) {
                unsigned t = __shfl_xor(v, o);
                v = t < v ? t : v;
            }
            if (mine_b && c != v) parent[c] = (int)v;
            if ((int)(threadIdx.x & 63) == leader) {
                const unsigned old = atomicMin(&rep[key], v);
                if (old != INF32 && old != v) {
                    if (old > v) parent[old] = (int)v;
                    else parent[v] = (int)old;
                }
            }
            todo &= ~mine;
        }
    }
}

// parent[] is updated by CAS from other workgroups while we walk it: read it with agent-scope atomic loads
// (served by L2, never by this CU's non-coherent L1) or a retry loop could spin on a stale line.
__device__ __forceinline__ int uf_find(int* parent, int x) {
    for (;;) {
        int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        int gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // path halving: any ancestor is a valid parent, so a racy store only shortens chains
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p;
    }
}
// Cached variant: ordinary loads (may be served by this CU's L1, i.e. be STALE).  A stale parent is an older
// ancestor, so whatever this returns is an ancestor of x: equal results for two cells still prove they are
// connected; a stale "root" is caught by the look in front of the CAS in uf_union_from.
// (Thousands of lanes ending their walk on the one hot root word through the L2 was the bottleneck.)
__device__ __forceinline__ int uf_find_cached(const int* parent, int x) {
    for (int hop = 0; hop < 64; ++hop) {
        int p = parent[x];
        if (p == x) return x;
        x = p;
    }
    return x;
}
// read-only walk with agent-scope loads (sees every union that has landed in L2, stores nothing)
__device__ __forceinline__ int uf_find_ro(const int* parent, int x) {
    for (;;) {
        const int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        x = p;
    }
}
// (Hanging the start cell directly under what the walk found -- path compression by plain stores -- was measured: k_db_union
//  37.5 -> 40.2 us per fold step.  The walks are short; the pass is bound by its ~250 scattered cache-line requests per cell.)
// Which of two roots stays a root is a fixed total order over the cells (concurrent CASes cannot close a cycle): cells that
// hold ANCHOR cores come first, then the lower index.  The anchor cells of a segment are one component from the start
// (k_db_anchor, rooted at the lowest of them), so that root never moves and every active cell that joins the anchor's
// cluster CASes its OWN parent word.  (With plain index order every active cell below the anchor's root took the root over
// in turn, thousands of waves retrying on one word: two thirds of k_db_union's time.)
__device__ __forceinline__ void uf_union_from(int* parent, int a, int b, const unsigned char* __restrict__ pri);
__device__ __forceinline__ void uf_union(int* parent, int a, int b, const unsigned char* __restrict__ pri) {
    uf_union_from(parent, uf_find_cached(parent, a), uf_find_cached(parent, b), pri);
}
// a, b: what cached walks from the two cells ended on (ancestors of them, roots unless stale).  Roots are hung under ROOTS
// (a union that hooks under whatever earlier node it holds was measured: the trees get deep, k_db_union_scan's walks went from
// 22 to 65 us per fold step), and the walks after a failed look halve the paths they climb.
__device__ __forceinline__ void uf_union_from(int* parent, int a, int b, const unsigned char* __restrict__ pri) {
    for (;;) {
        if (a == b) return;
        const unsigned char pa = pri[a], pb = pri[b];
        if (pa > pb || (pa == pb && a < b)) {
            int t = a;
            a = b;
            b = t;
        }                                   // b stays a root: hang a under it
        // `a` came from cached loads and may have been hooked by another wave long ago: a CAS on it then fails, and failing
        // CASes are not free -- same-address atomics retire one per ~11 ns on this GPU (scripts/microbench/atom_bench.hip),
        // and every active cell of a cluster that reaches the anchor tries the SAME root.  Atomic LOADS of one word by
        // thousands of lanes cost nothing measurable: look first, CAS only a word that still is a root.
        int up = __hip_atomic_load(&parent[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (up == a) {
            up = atomicCAS(&parent[a], a, b);
            if (up == a) return;
        }
        if (up == b) return;                // somebody else made this very hook (the usual way a look fails)
        a = uf_find(parent, up);
        b = uf_find(parent, b);
    }
}

// One wave per occupied cell: AABB and number of its core points -- and, round 6, what the points' core flags mean for the cell
// (until round 5 every point told its cell by atomics in a launch of its own): the slots of the counted points take their
// flag over from k_db_count, minidx[c] = the cell's smallest core index (INF: none -- the later passes skip the cell), active[c] =
// it holds a core point that is not an anchor (its connections have to be searched); a flag the batch promotes inside the
// anchor member of a segment that runs in place is set where the point lives.  Indices are positions in the BATCH: they order a
// segment's points like positions in the segment do.
__global__ void k_db_cellbox(const double* __restrict__ pts, const int* __restrict__ corecells, const unsigned* __restrict__ ncore,
                             const unsigned* __restrict__ cnt, const unsigned* __restrict__ start, const unsigned* __restrict__ sidx,
                             unsigned char* __restrict__ score, const unsigned char* __restrict__ core, double* __restrict__ cellbox,
                             unsigned* __restrict__ ccore, const int* __restrict__ cseg, const unsigned char* __restrict__ hasanchor,
                             unsigned* __restrict__ rep, int* __restrict__ parent, const DbSeg* __restrict__ segs,
                             unsigned* __restrict__ minidx, unsigned* __restrict__ active, unsigned char* __restrict__ poolcore_w) {
    // (the anchor cells' pre-connection rides in this launch: it needs the cell list and k_db_fill's hasanchor[], and nothing of the boxes)
    if (rep) db_anchor_cells(corecells, ncore, cseg, hasanchor, rep, parent);
    const int lane = threadIdx.x & 63;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6, ncells = *ncore;
    for (unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < ncells; w += nwaves) {
    const long long c = corecells[w];
    const unsigned s0 = start[c], e0 = s0 + cnt[c];
    const int k_seg = cseg[c];
    const int out_mode = segs[k_seg].out_mode;
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    int ncorepts = 0;
    unsigned first = INF32;
    bool act = false;
    for (unsigned k = s0 + lane; k < e0; k += 64) {          // (pts / score: the cell-sorted copies)
        unsigned char sc = score[k];
        const unsigned i = sidx[k];
        if (sc == 0) {                           // a counted point
            sc = core[i] ? DB_SLOT_CORE : 0;
            if (sc) score[k] = sc;
        }
        if (!sc) continue;
        ++ncorepts;
        first = i < first ? i : first;
        if (sc != DB_SLOT_ANCHOR) {
            act = true;
            if (out_mode == 2) {
                const long long pt_base = segs[k_seg].pt_base;
                if ((long long)i < pt_base + segs[k_seg].n_first) poolcore_w[segs[k_seg].out_off + ((long long)i - pt_base)] = 1;
            }
        }
        for (int a = 0; a < 3; ++a) {
            double v = pts[(size_t)k * 3 + a];
            mn[a] = v < mn[a] ? v : mn[a];
            mx[a] = v > mx[a] ? v : mx[a];
        }
    }
    for (int a = 0; a < 3; ++a) {
        mn[a] = wave_min_f64(mn[a]);
        mx[a] = wave_max_f64(mx[a]);
    }
    ncorepts = wave_sum_i32(ncorepts);
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = __shfl_xor(first, o);
        first = t < first ? t : first;
    }
    const bool any_act = __any(act) != 0;
    if (lane == 0) {
        for (int a = 0; a < 3; ++a) {
            cellbox[(size_t)w * 6 + a] = mn[a];
            cellbox[(size_t)w * 6 + 3 + a] = mx[a];
        }
        ccore[w] = (unsigned)ncorepts;           // core points of the cell (cluster sizes are summed per cell)
        minidx[c] = first;
        active[c] = any_act ? 1u : 0u;
    }
    }
}

// Box pass.  One WAVE per ACTIVE core cell (cells whose cores are all anchor points are pre-connected; a pair is
// handled from its active member, from the higher one when both are active), one LANE per neighbour cell.  The
// tight AABBs of the two cells' core points decide most pairs without touching a point: farthest corners closer
// than eps -> every pair of core points is a witness -> union.  (Pairs the boxes cannot decide are left to
// k_db_union_scan.)  Everything a lane needs about its neighbour is loaded up front: the lane is one serial
// chain of L2 round trips, and the kernel lasts as long as the longest chain.
__global__ void k_db_union(const int* __restrict__ corecells, const unsigned* __restrict__ ncore, const int* __restrict__ cseg, const DbSeg* __restrict__ segs,
                           int K, const unsigned* __restrict__ minidx, double eps2, const int* __restrict__ cellpos,
                           const double* __restrict__ cellbox, int* __restrict__ parent, const unsigned* __restrict__ active,
                           const unsigned char* __restrict__ hasanchor) {
    const int lane = threadIdx.x & 63;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6, ncells = *ncore;
    // (two waves per cell, 64 neighbour cells each, was tried: 40 -> 48 us per step -- the per-wave preamble and the
    //  contention on the roots cost more than the second trip through the lane's chain)
    for (unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < ncells; w += nwaves) {
        const long long c = corecells[w];
        if (!active[c]) continue;           // (the list holds every occupied cell: only those with a core point that is not an anchor search)
        const int lo = cseg[c];             // segment of the cell (written when the cell was listed)
        const DbSeg sg = segs[lo];
        int ix, iy, iz;
        cell_xyz(sg, c, ix, iy, iz);
        const int rc = uf_find_cached(parent, (int)c);     // (may go stale: only costs a redundant uf_union)
        double ba[6];
        for (int a = 0; a < 6; ++a) ba[a] = cellbox[(size_t)cellpos[c] * 6 + a];
        // both of the lane's neighbour cells (o = lane and lane + 64) in one straight-line pass: their table entries, their
        // boxes and their root walks are independent chains of L2 round trips, issued side by side instead of one trip after
        // the other (the kernel lasts as long as a lane's chain: 39 -> ~30 us per fold step)
        long long c2[2];
        bool ok[2];
        unsigned mi[2] = {INF32, INF32}, ac[2] = {0u, 0u};
        int p2[2] = {0, 0}, cp2[2] = {0, 0};
        unsigned char ha[2] = {0, 0};
        const unsigned char ha_own = hasanchor[c];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int o = lane + 64 * q;
            const int dx = o / 25 - 2, dy = (o / 5) % 5 - 2, dz = o % 5 - 2;
            const int jx = ix + dx, jy = iy + dy, jz = iz + dz;
            ok[q] = o < 125 && o != 62 && jx >= 0 && jy >= 0 && jz >= 0 && jx < sg.nx && jy < sg.ny && jz < sg.nz;   // 62: the cell itself
            c2[q] = ok[q] ? sg.cell_base + ((long long)jx * sg.ny + jy) * sg.nz + jz : c;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (ok[q]) {
                mi[q] = minidx[c2[q]];
                ac[q] = active[c2[q]];
                p2[q] = parent[c2[q]];
                cp2[q] = cellpos[c2[q]];         // (garbage unless c2 is a core cell; not used then)
                ha[q] = hasanchor[c2[q]];
            }
        double bb[2][6];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ok[q] = ok[q] && mi[q] != INF32 && !(c2[q] < c && ac[q]);
            if (ok[q])
                for (int a = 0; a < 6; ++a) bb[q][a] = cellbox[(size_t)cp2[q] * 6 + a];
        }
        int r2[2] = {rc, rc};
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (ok[q]) r2[q] = uf_find_cached(parent, p2[q]);
        // Which neighbours the boxes connect the cell with (and that the walks do not show connected already).  What is done
        // about them here is at most ONE hook of the cell's own root (below); round 4 measured why: with one union per distinct
        // neighbour root, every cell of the batch at once, three of four unions found their root hooked already when they got
        // there, and each such failure is a look, two walks with atomic loads and another trip (7 900 unions and 5 900 failures
        // a batch on configs[1]; the kernel took 39 us, 12 with the unions switched off).
        bool want[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            want[q] = false;
            if (ok[q] && r2[q] != rc) {
                double mx2 = 0.0;
                for (int a = 0; a < 3; ++a) {
                    double far = fmax(ba[3 + a] - bb[q][a], bb[q][3 + a] - ba[a]);
                    mx2 += far * far;
                }
                want[q] = mx2 < eps2 * (1.0 - 1e-12);
            }
        }
        // A cell that reaches the anchor's cluster makes that ONE hook -- its own root under the anchor's, a word nobody else
        // writes in this pass -- and leaves its other neighbours to k_db_union_scan, which runs when these hooks have all landed.
        // Measured on configs[1] before: 7 900 unions a batch, 5 900 of them finding their root hooked already (every cell of
        // a cluster hooks its neighbours at the same time, and nearly all of them end up under the anchor by their own hook
        // anyway); a failed look costs two walks.  Almost every pair left over is connected by the time the scan pass looks.
        {
            const unsigned long long am0 = __ballot(want[0] && ha[0] != 0), am1 = __ballot(want[1] && ha[1] != 0);
            if ((am0 | am1) != 0ull || ha_own) {
                if (am0 | am1) {
                    const int ra = am0 ? __shfl(r2[0], __ffsll(am0) - 1) : __shfl(r2[1], __ffsll(am1) - 1);
                    // (the cell holds no anchor core, so its root comes after the anchor's in the order: no priority bytes to
                    //  load, and while the cell is its own root nobody else has a reason to write that word -- CAS straight away)
                    if (lane == 0 && !(!ha_own && rc == (int)c && atomicCAS(&parent[rc], rc, ra) == rc)) uf_union_from(parent, rc, ra, hasanchor);
                }
                continue;
            }
        }
        // No anchor in reach: the cell hangs ITSELF under the earliest root among the neighbours the boxes connect it with, if
        // one comes before it -- again a word nobody else writes in this pass (every cell without anchor cores starts the pass
        // as its own root, and a neighbour only ever hooks its own root) -- and leaves the rest to k_db_union_scan.  Measured
        // with every root of the neighbourhood united here, all cells at once: 20 of this kernel's 33 us went into looks that
        // failed and the walks after them; the forest this builds without a single contended word already joins most of a
        // cluster, and the scan pass unites what is left when it has landed.
        int m = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (want[q] && r2[q] < rc) m = min(m, r2[q]);
        for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
        if (lane == 0 && m != 0x7fffffff && rc == (int)c) atomicCAS(&parent[rc], rc, m);
    }
}

// Scan passes (pass 1: Chebyshev distance 1, pass 2: distance 2): for the neighbour pairs the AABBs could not
// decide and that are still unconnected, look for ONE witness pair of core points closer than eps.  The whole
// wave works on one pair: lanes take the points of c in turn (point-to-box pruned), each walks c2 with early
// exit, and the wave stops at the first hit -- a heavily re-observed cell holds hundreds of points.
__global__ void k_db_union_scan(const double* __restrict__ pts, const int* __restrict__ corecells,
                                const unsigned* __restrict__ ncore, const int* __restrict__ cseg, const DbSeg* __restrict__ segs, int K,
                                const unsigned* __restrict__ cnt, const unsigned* __restrict__ start,
                                const unsigned* __restrict__ ord, const unsigned char* __restrict__ core,
                                const unsigned* __restrict__ minidx, double eps2, const int* __restrict__ cellpos,
                                const double* __restrict__ cellbox, int* __restrict__ parent,
                                const unsigned* __restrict__ active, const unsigned char* __restrict__ hasanchor) {
    const int lane = threadIdx.x & 63;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6, ncells = *ncore;
    for (unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < ncells; w += nwaves) {
        const long long c = corecells[w];
        if (!active[c]) continue;           // (see k_db_union)
        const int lo = cseg[c];             // segment of the cell (written when the cell was listed)
        const DbSeg sg = segs[lo];
        int ix, iy, iz;
        cell_xyz(sg, c, ix, iy, iz);
        const unsigned s0 = start[c], e0 = s0 + cnt[c];
        double ba[6];
        for (int a = 0; a < 6; ++a) ba[a] = cellbox[(size_t)cellpos[c] * 6 + a];
        const int rc = uf_find_cached(parent, (int)c);
        // phase A, once for all 124 neighbour cells (the lane's two, o = lane and lane + 64, side by side as in k_db_union: table
        // entries, boxes and root walks are independent chains of L2 round trips): does the pair need a point scan at all?
        // (Four trips through this chain -- one per Chebyshev distance and half of the neighbourhood -- were most of the kernel.)
        long long c2[2];
        bool ok[2];
        int chb[2];
        unsigned mi[2] = {INF32, INF32}, ac[2] = {0u, 0u};
        int p2[2] = {0, 0}, cp2[2] = {0, 0};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int o = lane + 64 * q;
            const int dx = o / 25 - 2, dy = (o / 5) % 5 - 2, dz = o % 5 - 2;
            chb[q] = max(abs(dx), max(abs(dy), abs(dz)));
            const int jx = ix + dx, jy = iy + dy, jz = iz + dz;
            ok[q] = o < 125 && o != 62 && jx >= 0 && jy >= 0 && jz >= 0 && jx < sg.nx && jy < sg.ny && jz < sg.nz;
            c2[q] = ok[q] ? sg.cell_base + ((long long)jx * sg.ny + jy) * sg.nz + jz : c;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (ok[q]) {
                mi[q] = minidx[c2[q]];
                ac[q] = active[c2[q]];
                p2[q] = parent[c2[q]];
                cp2[q] = cellpos[c2[q]];         // (garbage unless c2 is a core cell; not used then)
            }
        double bq[2][6];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ok[q] = ok[q] && mi[q] != INF32 && (c2[q] > c || !ac[q]);
            if (ok[q])
                for (int a = 0; a < 6; ++a) bq[q][a] = cellbox[(size_t)cp2[q] * 6 + a];
        }
        bool need[2] = {false, false}, direct[2] = {false, false};
        int r2[2] = {rc, rc};
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (ok[q] && (r2[q] = uf_find_cached(parent, p2[q])) != rc) {
                double mn2 = 0.0, mx2 = 0.0;
                for (int a = 0; a < 3; ++a) {
                    double gap = fmax(0.0, fmax(ba[a] - bq[q][3 + a], bq[q][a] - ba[3 + a]));
                    double far = fmax(ba[3 + a] - bq[q][a], bq[q][3 + a] - ba[a]);
                    mn2 += gap * gap;
                    mx2 += far * far;
                }
                // (mx2 < eps2: decided by the box pass; mn2 >= eps2: no witness possible)
                direct[q] = mx2 < eps2 * (1.0 - 1e-12);
                need[q] = !direct[q] && mn2 < eps2 * (1.0 + 1e-12);
            }
        // pairs the boxes decide that k_db_union left to this pass (its cell had the anchor to reach) and that are still two sets:
        // one union per distinct root, all at the same time
        {
            bool lead[2] = {false, false};
            unsigned long long todo = __ballot(direct[0]);
            while (todo) {
                const int leader = __ffsll(todo) - 1;
                const int key = __shfl(r2[0], leader);
                if (lane == leader) lead[0] = true;
                todo &= ~__ballot(direct[0] && r2[0] == key);
                direct[1] = direct[1] && r2[1] != key;
            }
            todo = __ballot(direct[1]);
            while (todo) {
                const int leader = __ffsll(todo) - 1;
                const int key = __shfl(r2[1], leader);
                if (lane == leader) lead[1] = true;
                todo &= ~__ballot(direct[1] && r2[1] == key);
            }
            if (lead[0] || lead[1]) uf_union_from(parent, rc, lead[0] ? r2[0] : r2[1], hasanchor);
            if (lead[0] && lead[1]) uf_union_from(parent, rc, r2[1], hasanchor);
        }
        // phase B: the whole wave scans the pairs that need it, one after the other -- Chebyshev distance 1 first (they connect
        // most components), then distance 2; a pair that an earlier scan of this cell (or another wave) has connected meanwhile
        // is dropped when its turn comes
        for (int pass = 1; pass <= 2; ++pass)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          unsigned long long todo = __ballot(need[q] && chb[q] == pass);
          while (todo) {
            const int src_lane = __ffsll(todo) - 1;
            todo &= todo - 1;
            const long long c2s = __shfl(c2[q], src_lane);
            // (one lane looks, every lane follows: the lanes' own walks could see different trees while other waves hook)
            int connected = 0;
            if (lane == 0) connected = uf_find_ro(parent, (int)c) == uf_find_ro(parent, (int)c2s) ? 1 : 0;
            if (__shfl(connected, 0)) continue;
            const double* bb = cellbox + (size_t)cellpos[c2s] * 6;
            const unsigned s1 = start[c2s], e1 = s1 + cnt[c2s];
            bool hit = false;
            for (unsigned a0 = s0; a0 < e0; a0 += 64) {
                unsigned a = a0 + lane;
                if (a < e0) {
                    unsigned ia = a;
                    if (core[ia]) {
                        const double* pa = pts + (size_t)ia * 3;
                        double g2 = 0.0;                       // point-to-box lower bound
                        for (int t = 0; t < 3; ++t) {
                            double gq = fmax(0.0, fmax(bb[t] - pa[t], pa[t] - bb[3 + t]));
                            g2 += gq * gq;
                        }
                        // (eight candidates per step with independent loads: one candidate per step made every step a round trip,
                        //  and a cell where a surface has piled up its re-observations holds a hundred of them)
                        if (g2 < eps2 * (1.0 + 1e-12))
                            for (unsigned b = s1; b < e1 && !hit; b += 8u) {
                                bool h = false;
#pragma unroll
                                for (int j = 0; j < 8; ++j) {
                                    const unsigned ib = min(b + (unsigned)j, e1 - 1u);
                                    const bool cb = core[ib] != 0;
                                    const double d = dist2_f64(pa, pts + (size_t)ib * 3);
                                    h = h || (b + (unsigned)j < e1 && cb && d < eps2);
                                }
                                hit = h;
                            }
                    }
                }
                if (__any(hit)) {
                    hit = true;
                    break;
                }
            }
            if (hit && lane == 0) uf_union(parent, (int)c, (int)c2s, hasanchor);
          }
        }
    }
}

// (the kernels below walk the compact core-cell list with a grid-stride loop: the list length is only
//  known on the device, and launching one thread per grid cell would be dominated by empty cells)
// Final roots + per-cluster keys.  Every core cell looks its root up with a READ-ONLY walk (roots never change
// after the union passes) and stores it: every parent[] must be a root when the kernel ends, labels are read
// straight from it.  (No path halving here: a halving store of one thread may land after another thread has
// written the root into the same word and put an inner node back.)  Then, per cluster: order key = smallest core
// index (kept at the root cell), number of core members, and the number of clusters per segment.
#define DB_ROOTS 8u
__global__ void k_db_rootmin(const int* __restrict__ corecells, const unsigned* __restrict__ ncore, int* __restrict__ parent,
                             const unsigned* __restrict__ minidx, unsigned* __restrict__ rootmin,
                             const int* __restrict__ cseg, const DbSeg* __restrict__ segs, int K,
                             unsigned* __restrict__ ncl, const unsigned* __restrict__ ccore, unsigned* __restrict__ size,
                             int* __restrict__ roots, int* __restrict__ rhead, int* __restrict__ rnext,
                             const unsigned* __restrict__ needy, const unsigned* __restrict__ n_needy, const unsigned char* __restrict__ core,
                             unsigned* __restrict__ nclist, unsigned* __restrict__ n_noncore) {
    // (riding along, on the workgroups from the END of the grid -- the cell list keeps the first few dozen busy --: the counted
    //  points that came out non-core, as a compact list for k_db_label; one atomic per workgroup trip)
    {
        __shared__ unsigned s_w[4], s_base;
        const unsigned nn = *n_needy;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        for (unsigned t0 = (gridDim.x - 1u - blockIdx.x) * blockDim.x; t0 < nn; t0 += gridDim.x * blockDim.x) {   // block-uniform
            const unsigned t = t0 + threadIdx.x;
            unsigned i = 0;
            bool nc = false;
            if (t < nn) {
                i = needy[t];
                nc = core[i] == 0;
            }
            const unsigned long long m = __ballot(nc);
            if (lane == 0) s_w[wv] = (unsigned)__popcll(m);
            __syncthreads();
            if (threadIdx.x == 0) {
                const unsigned tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
                s_base = tot ? atomicAdd(n_noncore, tot) : 0u;
            }
            __syncthreads();
            unsigned pos = s_base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            for (int q = 0; q < wv; ++q) pos += s_w[q];
            if (nc) nclist[pos] = i;
            __syncthreads();                     // (s_w / s_base are rewritten by the next trip)
        }
    }
    const unsigned n = *ncore;
    const unsigned stride = gridDim.x * blockDim.x;
    // wave-uniform trip count; one atomic per (wave, root): a cluster's cells all target the same word
    for (unsigned w0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); w0 < n; w0 += stride) {
        unsigned w = w0 + (threadIdx.x & 63u);
        bool valid = w < n;
        int root = -1;
        unsigned mi = INF32, nc = 0;
        if (valid && minidx[corecells[w]] == INF32) valid = false;      // (an occupied cell without a core point: in no cluster)
        if (valid) {
            int c = corecells[w];
            root = c;
            for (;;) {
                const int q = __hip_atomic_load(&parent[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (q == root) break;
                root = q;
            }
            __hip_atomic_store(&parent[c], root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mi = minidx[c];
            nc = ccore[w];
            if (root == c) {                    // one root cell per cluster: clusters of the segment
                const int lo = cseg[c];             // segment of the cell (written when the cell was registered)
                // ... listed per segment: k_db_compact picks the segment's largest cluster from this list itself (db_best_cluster;
                // until round 5 a launch of its own, k_db_pick, between k_db_label and the compaction).  The first DB_ROOTS roots
                // of a segment sit side by side (one round trip to fetch them all), further ones are chained up.
                const unsigned slot = atomicAdd(&ncl[lo], 1u);
                if (slot < DB_ROOTS) roots[(size_t)lo * DB_ROOTS + slot] = c;
                else rnext[c] = atomicExch(&rhead[lo], c);
            }
        }
        unsigned long long todo = __ballot(valid);
        while (todo) {
            int leader = __ffsll(todo) - 1;
            int key = __shfl(root, leader);
            const bool mine_b = valid && root == key;
            unsigned long long mine = __ballot(mine_b);
            unsigned v = mine_b ? mi : INF32, sum = mine_b ? nc : 0u;
            for (int o = 32; o > 0; o >>= 1) {
                unsigned t = __shfl_xor(v, o);
                v = t < v ? t : v;
                sum += __shfl_xor(sum, o);
            }
            if ((int)(threadIdx.x & 63) == leader) {
                atomicMin(&rootmin[key], v);
                atomicAdd(&size[key], sum);      // core members; k_db_label adds the border points
            }
            todo &= ~mine;
        }
    }
}

__global__ void k_db_label(const double* __restrict__ pts, const int* __restrict__ segid, const DbSeg* __restrict__ segs,
                           const long long* __restrict__ cellid, const unsigned* __restrict__ minidx,
                           const unsigned* __restrict__ start, const double* __restrict__ spts,
                           const unsigned char* __restrict__ score, const int* __restrict__ cellpos,
                           const double* __restrict__ cellbox, const int* __restrict__ parent, const unsigned* __restrict__ rootmin,
                           const unsigned* __restrict__ ncl, double eps2, int* __restrict__ label, unsigned* __restrict__ size,
                           unsigned* __restrict__ firstidx, unsigned* __restrict__ contested, const unsigned* __restrict__ nclist,
                           const unsigned* __restrict__ n_noncore) {
    // Border search for the NON-CORE points only (the counted points that did not reach min_points, listed by k_db_rootmin; a core point's label is simply the
    // root of its cell, k_db_flags looks it up itself).  One WAVE per point, one LANE per neighbour cell (125 cells
    // in two rounds): a lane decides its cell -- core cell? tight box of its core points in reach? a core point
    // within eps? -- and the wave reduces to the cluster with the smallest order key (Open3D: the first cluster
    // that reaches the point) and to "cores of two clusters in reach" (contested, see the merge stage).  A lane
    // walking the 125 cells one after the other was the longest serial chain of the whole batch.
    const int lane = threadIdx.x & 63;
    const unsigned nn = *n_noncore;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
    for (unsigned t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; t < nn; t += nwaves) {
        const long long i = nclist[t];
        const int k = segid[i];
        const DbSeg sg = segs[k];
        int ix, iy, iz;
        cell_xyz(sg, cellid[i], ix, iy, iz);
        const double pi[3] = {pts[(size_t)i * 3], pts[(size_t)i * 3 + 1], pts[(size_t)i * 3 + 2]};
        unsigned mykey = INF32;                 // best (smallest-key) cluster this lane found a witness of
        int myr = -1;
        bool mymulti = false;                   // ... and whether its two cells gave witnesses of different clusters
        for (int round = 0; round < 2; ++round) {
            const int o = round * 64 + lane;
            if (o >= 125) continue;
            const int jx = ix + o / 25 - 2, jy = iy + (o / 5) % 5 - 2, jz = iz + o % 5 - 2;
            if (jx < 0 || jy < 0 || jz < 0 || jx >= sg.nx || jy >= sg.ny || jz >= sg.nz) continue;
            const long long c2 = sg.cell_base + ((long long)jx * sg.ny + jy) * sg.nz + jz;
            if (minidx[c2] == INF32) continue;                               // no core point in the cell
            const int r = parent[c2];
            const unsigned key = rootmin[r];
            const double* bb = cellbox + (size_t)cellpos[c2] * 6;            // tight box of its core points
            double g2 = 0.0;
            for (int q = 0; q < 3; ++q) {
                double gq = fmax(0.0, fmax(bb[q] - pi[q], pi[q] - bb[3 + q]));
                g2 += gq * gq;
            }
            if (g2 >= eps2 * (1.0 + 1e-12)) continue;
            if (!any_core_within(spts, score, start[c2], start[c2 + 1], pi, eps2)) continue;
            if (myr >= 0 && r != myr) mymulti = true;
            if (key < mykey) {
                mykey = key;
                myr = r;
            }
        }
        unsigned best = mykey;
        for (int o = 32; o > 0; o >>= 1) {
            unsigned u = __shfl_xor(best, o);
            best = u < best ? u : best;
        }
        int lab = -1;
        bool contest = false;
        if (best != INF32) {
            const unsigned long long owners = __ballot(mykey == best);      // all of them hold the same cluster (keys are unique per cluster)
            lab = __shfl(myr, __ffsll(owners) - 1);
            contest = __any(mymulti || (myr >= 0 && myr != lab)) != 0;
        }
        if (lane == 0) {
            label[i] = lab;
            if (lab >= 0) {
                // (core members were counted per cell by k_db_rootmin, and their smallest index is the cluster key)
                atomicAdd(&size[lab], 1u);
                atomicMin(&firstidx[lab], (unsigned)i);          // (batch position, like minidx)
                if (contest && ncl[k] > 1u && !contested[k]) contested[k] = 1u;
            }
        }
    }
}

// The segment's largest cluster (ties: first label in point order) as size << 32 | ~first member index, 0 if it has none: a walk
// over the segment's root cells (k_db_rootmin's chain; a segment has a handful of clusters).  Sizes and first border members are
// complete when k_db_label has ended.
__device__ __forceinline__ unsigned long long db_cluster_key(int c, const unsigned* __restrict__ size, const unsigned* __restrict__ firstidx,
                                                            const unsigned* __restrict__ rootmin) {
    const unsigned sz = size[c];
    if (sz == 0u) return 0ull;
    const unsigned first = min(firstidx[c], rootmin[c]);
    return ((unsigned long long)sz << 32) | (unsigned long long)(INF32 - first);
}
__device__ __forceinline__ unsigned long long db_best_cluster(int k, const unsigned* __restrict__ ncl, const int* __restrict__ roots,
                                                              const int* __restrict__ rhead, const int* __restrict__ rnext,
                                                              const unsigned* __restrict__ size, const unsigned* __restrict__ firstidx,
                                                              const unsigned* __restrict__ rootmin) {
    const unsigned n = ncl[k];
    int r[DB_ROOTS];
#pragma unroll
    for (unsigned j = 0; j < DB_ROOTS; ++j) r[j] = j < n ? roots[(size_t)k * DB_ROOTS + j] : -1;
    unsigned long long b = 0ull;
#pragma unroll
    for (unsigned j = 0; j < DB_ROOTS; ++j)
        if (r[j] >= 0) {
            const unsigned long long key = db_cluster_key(r[j], size, firstidx, rootmin);
            b = key > b ? key : b;
        }
    if (n > DB_ROOTS)
        for (int c = rhead[k]; c >= 0; c = rnext[c]) {
            const unsigned long long key = db_cluster_key(c, size, firstidx, rootmin);
            b = key > b ? key : b;
        }
    return b;
}
// cluster roots are core cells: pick the largest cluster per segment (ties: first label in point order)
// (the legacy three-launch compaction's form, HMSG_DB_COMPACT_SPLIT=1)
__global__ void k_db_pick(const int* __restrict__ corecells, const unsigned* __restrict__ ncore, const int* __restrict__ cseg, const DbSeg* __restrict__ segs,
                          int K, const unsigned* __restrict__ size, const unsigned* __restrict__ firstidx,
                          const unsigned* __restrict__ rootmin, unsigned long long* __restrict__ best) {
    const unsigned n = *ncore;
    for (unsigned w = blockIdx.x * blockDim.x + threadIdx.x; w < n; w += gridDim.x * blockDim.x) {
        long long c = corecells[w];
        if (size[c] == 0u) continue;
        const int lo = cseg[c];             // segment of the cell (written when the cell was registered)
        const unsigned first = min(firstidx[c], rootmin[c]);
        unsigned long long key = ((unsigned long long)size[c] << 32) | (unsigned long long)(INF32 - first);
        atomicMax(&best[lo], key);
    }
}
// graph_utils.py:853-880: keep the largest cluster unless there is none or it has < 5 points
// (best[k] = size << 32 | ~first member index of the segment's largest cluster, 0 if it has none: the winner's
//  label is the label of that first member; a core point's label is the root of its cell, a non-core point's
//  was written by k_db_label)
__global__ void k_db_flags(long long N, const int* __restrict__ segid, const DbSeg* __restrict__ segs,
                           const int* __restrict__ label, const unsigned long long* __restrict__ best,
                           unsigned* __restrict__ flags, unsigned* __restrict__ dropped, const unsigned char* __restrict__ core,
                           const long long* __restrict__ cellid, const int* __restrict__ parent) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int k = segid[i];
    const unsigned long long b = best[k];
    bool keep = true;
    if ((unsigned)(b >> 32) >= 5u) {
        const long long fm = (long long)(INF32 - (unsigned)(b & 0xffffffffull));
        const int wl = core[fm] ? parent[cellid[fm]] : label[fm];
        const int my = core[i] ? parent[cellid[i]] : label[i];
        keep = my == wl;
    }
    flags[i] = keep ? 1u : 0u;
    if (!keep && !dropped[k]) dropped[k] = 1u;      // the segment loses points: its box has to be re-reduced
}

// compaction of the kept points; the last point of every segment derives the segment's output count from
// the scan (no per-point atomics).  The AABB of the kept points of every segment comes back with the counts
// (one copy, no second pass): a persistent grid walks the points in contiguous chunks, every lane keeps a
// running box for the segment it is in, and the wave flushes (one set of atomics per (wave, segment)) only
// when some lane crosses into another segment and at the end -- thousands of waves hitting the same six
// words of a dominant segment once per 64 points was the slowest thing in the batch.
// (the block's first flushed segment is combined in LDS and reaches the global box once per block: a dominant
//  segment otherwise draws a dozen same-line L2 atomics from every wave of the grid, which serialise)
__device__ __forceinline__ void db_flush_boxes(bool have, int seg, const double* mn, const double* mx,
                                               unsigned long long* __restrict__ obounds, int* slot_seg,
                                               unsigned long long* slot_box) {
    unsigned long long todo = __ballot(have);
    while (todo) {
        const int leader = __ffsll(todo) - 1;
        const int key = __shfl(seg, leader);
        const bool mine_b = have && seg == key;
        const unsigned long long mine = __ballot(mine_b);
        bool in_lds = false;
        if ((int)(threadIdx.x & 63) == leader) {
            const int owner = atomicCAS(slot_seg, -1, key);
            in_lds = owner == -1 || owner == key;
        }
        for (int a = 0; a < 3; ++a) {
            const double lo = wave_min_f64(mine_b ? mn[a] : 1e300), hi = wave_max_f64(mine_b ? mx[a] : -1e300);
            if ((int)(threadIdx.x & 63) == leader) {
                if (in_lds) {
                    atomicMin(&slot_box[a], enc_f64(lo));
                    atomicMax(&slot_box[3 + a], enc_f64(hi));
                } else {
                    atomicMin(&obounds[(size_t)key * 6 + a], enc_f64(lo));
                    atomicMax(&obounds[(size_t)key * 6 + 3 + a], enc_f64(hi));
                }
            }
        }
        todo &= ~mine;
    }
}
__global__ void k_db_scatter(const double* __restrict__ pts, long long N, const int* __restrict__ segid,
                             const DbSeg* __restrict__ segs, const unsigned* __restrict__ flags,
                             const unsigned* __restrict__ pos, double* __restrict__ dst, int* __restrict__ ocount,
                             const unsigned char* __restrict__ core, unsigned char* __restrict__ dst_core,
                             unsigned long long* __restrict__ obounds, const unsigned* __restrict__ dropped) {
    __shared__ int slot_seg;
    __shared__ unsigned long long slot_box[6];
    if (threadIdx.x == 0) slot_seg = -1;
    if (threadIdx.x < 6) slot_box[threadIdx.x] = threadIdx.x < 3 ? ~0ull : 0ull;
    __syncthreads();
    const long long per_block = ((N + (long long)gridDim.x * blockDim.x - 1) / ((long long)gridDim.x * blockDim.x)) * blockDim.x;
    const long long b0 = (long long)blockIdx.x * per_block, b1 = b0 + per_block < N ? b0 + per_block : N;
    int cur = -1;                            // segment of the running box
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (long long base = b0; base < b1; base += blockDim.x) {      // block-uniform trip count
        const long long i = base + threadIdx.x;
        const bool in_range = i < b1;
        unsigned f = 0u;
        int k = cur;
        double v[3] = {0, 0, 0};
        if (in_range) {
            f = flags[i];
            k = segid[i];
            const unsigned p = pos[i];
            if (f) {
                for (int a = 0; a < 3; ++a) {
                    v[a] = pts[i * 3 + a];
                    dst[(size_t)p * 3 + a] = v[a];
                }
                if (dst_core) dst_core[p] = core[i];
            }
            const DbSeg sg = segs[k];
            if (i == sg.pt_base + sg.n - 1) ocount[k] = (int)(p + f - pos[sg.pt_base]);
        }
        if (f && !dropped[k]) f = 0u;                               // segment keeps every point: its input box stays exact
        if (__any(f != 0u && cur >= 0 && k != cur)) {               // somebody leaves its segment: flush all
            db_flush_boxes(cur >= 0, cur, mn, mx, obounds, &slot_seg, slot_box);
            cur = -1;
            for (int a = 0; a < 3; ++a) {
                mn[a] = 1e300;
                mx[a] = -1e300;
            }
        }
        if (f) {
            cur = k;
            for (int a = 0; a < 3; ++a) {
                mn[a] = v[a] < mn[a] ? v[a] : mn[a];
                mx[a] = v[a] > mx[a] ? v[a] : mx[a];
            }
        }
    }
    db_flush_boxes(cur >= 0, cur, mn, mx, obounds, &slot_seg, slot_box);
    __syncthreads();
    if (threadIdx.x < 6 && slot_seg >= 0) {
        if (threadIdx.x < 3) atomicMin(&obounds[(size_t)slot_seg * 6 + threadIdx.x], slot_box[threadIdx.x]);
        else atomicMax(&obounds[(size_t)slot_seg * 6 + threadIdx.x], slot_box[threadIdx.x]);
    }
}


// k_db_flags + the scan + k_db_scatter as ONE launch (decoupled look-back over the blocks' kept counts).  Every block owns a
// contiguous chunk of the batch: trip by trip it decides its points (bit `trip` of a register per thread), the block's count
// goes through the look-back table, and a second walk over the same chunk writes the kept points behind the block's prefix.
// Whether a segment loses points at all -- its box then has to be re-reduced -- needs no flag from another block: the winning
// cluster's size IS the number of kept points (k_db_rootmin counted its core members per cell, k_db_label its border points).
// A segment's output count comes from the positions of its first and last point (ostart / oend; the host subtracts).
// Round 5: the workgroups follow a table (DbBlk) instead of cutting [0, N) evenly: a segment whose output has a region of its
// own (SegDesc::out_mode 1 / 2) is a look-back chain of its own -- its kept points go to that region at the chain's own
// prefix -- and the anchor member of a segment that runs in place is not visited at all.
#define DBK_TRIPS 32
#define DBK_SEGS 256
__global__ void __launch_bounds__(256) k_db_compact(const double* __restrict__ pts, const DbBlk* __restrict__ blks, const int* __restrict__ segid,
                                                    const DbSeg* __restrict__ segs, const int* __restrict__ label,
                                                    const unsigned* __restrict__ ncl, const int* __restrict__ roots,
                                                    const int* __restrict__ rhead, const int* __restrict__ rnext, const unsigned* __restrict__ size,
                                                    const unsigned* __restrict__ firstidx, const unsigned* __restrict__ rootmin,
                                                    const unsigned* __restrict__ rep, unsigned* __restrict__ flags_dbg,
                                                    const unsigned char* __restrict__ core, const long long* __restrict__ cellid,
                                                    const int* __restrict__ parent, double* __restrict__ dst, int* __restrict__ oend,
                                                    int* __restrict__ ostart, int* __restrict__ ofirst, unsigned char* __restrict__ dst_core,
                                                    unsigned long long* __restrict__ obounds, unsigned long long* __restrict__ state,
                                                    unsigned epoch, double* __restrict__ pool_w, unsigned char* __restrict__ poolcore_w,
                                                    const unsigned long long* __restrict__ best_pre /* k_db_pick's winners (large batches), or nullptr */) {
    __shared__ int slot_seg;
    __shared__ unsigned long long slot_box[6];
    __shared__ unsigned wsum[4], wtot[2][4];
    __shared__ unsigned s_prefix;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) slot_seg = -1;
    if (tid < 6) slot_box[tid] = tid < 3 ? ~0ull : 0ull;
    const DbBlk bk = blks[blockIdx.x];
    const long long b0 = bk.p0, b1 = bk.p0 + bk.cnt;
    if (bk.seg >= 0) {                                       // an output region of its own
        const DbSeg so = segs[bk.seg];
        const long long o = so.out_off + (so.out_mode == 2 ? so.n_first : 0);
        dst = pool_w + (size_t)o * 3;
        if (dst_core) dst_core = poolcore_w + o;
    }
    // the winners of the segments this workgroup's points belong to (segments are contiguous in the batch: those of its first and of
    // its last point and the ones between), one thread per segment, side by side
    __shared__ unsigned long long s_best[DBK_SEGS];
    const int k_lo = bk.cnt > 0 ? segid[b0] : 0, k_hi = bk.cnt > 0 ? segid[b1 - 1] : -1;
    const bool best_in_lds = k_hi - k_lo < DBK_SEGS;
    if (best_in_lds && tid <= k_hi - k_lo)
        s_best[tid] = segs[k_lo + tid].forced ? 0ull : (best_pre ? best_pre[k_lo + tid] : db_best_cluster(k_lo + tid, ncl, roots, rhead, rnext, size, firstidx, rootmin));
    __syncthreads();
    unsigned mine = 0u, kept = 0u;
    int trip = 0;
    for (long long base = b0; base < b1; base += blockDim.x, ++trip) {
        const long long i = base + tid;
        if (i >= b1) continue;
        const int k = segid[i];
        const DbSeg sg = segs[k];
        const unsigned long long b = sg.forced ? 0ull : (best_in_lds ? s_best[k - k_lo] : (best_pre ? best_pre[k] : db_best_cluster(k, ncl, roots, rhead, rnext, size, firstidx, rootmin)));
        bool keep = true;                                    // graph_utils.py:853-880 (see k_db_flags)
        if (sg.forced) {
            // the anchor member is kept whole and its cluster is the winner (SegDesc::forced); the rest is kept where it joined it
            if (i >= sg.pt_base + sg.n_first) {
                const unsigned ra = rep[k];                  // lowest cell with anchor cores (INF32: none inside the crop)
                const int wl = ra != INF32 ? parent[ra] : -2;
                const int my = core[i] ? parent[cellid[i]] : label[i];
                keep = my == wl;
            }
        } else if ((unsigned)(b >> 32) >= 5u) {
            const long long fm = (long long)(INF32 - (unsigned)(b & 0xffffffffull));      // (batch position of the winner's first member)
            const int wl = core[fm] ? parent[cellid[fm]] : label[fm];
            const int my = core[i] ? parent[cellid[i]] : label[i];
            keep = my == wl;
        }
        if (flags_dbg) flags_dbg[i] = keep ? 1u : 0u;
        if (keep) {
            mine |= 1u << trip;
            ++kept;
        }
    }
    for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o);
    if (lane == 0) wsum[w] = kept;
    __syncthreads();
    if (w == 0) {
        const unsigned prefix = scan_lookback_prefix(state + bk.chain0, bk.tile, epoch, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
        if (lane == 0) s_prefix = prefix;
    }
    __syncthreads();
    unsigned running = s_prefix;                             // block-uniform: output slot of the trip's first kept point
    int cur = -1;                                            // segment of the running box
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    trip = 0;
    for (long long base = b0; base < b1; base += blockDim.x, ++trip) {      // block-uniform trip count
        const long long i = base + tid;
        const bool in_range = i < b1;
        unsigned f = in_range ? (mine >> trip) & 1u : 0u;
        const unsigned long long m = __ballot(f != 0u);
        if (lane == 0) wtot[trip & 1][w] = (unsigned)__popcll(m);
        __syncthreads();                                     // (the other buffer is rewritten only after the next trip's barrier)
        unsigned woff = 0u, tot = 0u;
        for (int q = 0; q < 4; ++q) {
            const unsigned t = wtot[trip & 1][q];
            woff += q < w ? t : 0u;
            tot += t;
        }
        const unsigned p = running + woff + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        running += tot;
        int k = cur;
        double v[3] = {0, 0, 0};
        bool drops = false;
        if (in_range) {
            k = segid[i];
            const DbSeg sg = segs[k];
            if (f) {
                for (int a = 0; a < 3; ++a) {
                    v[a] = pts[i * 3 + a];
                    dst[(size_t)p * 3 + a] = v[a];
                }
                if (dst_core) dst_core[p] = core[i];
            }
            if (i == sg.pt_base) ostart[k] = (int)p;
            if (sg.n_first > 0 && sg.n_first < sg.n && i == sg.pt_base + sg.n_first) ofirst[k] = (int)p;   // where the first member's output ends
            if (i == sg.pt_base + sg.n - 1) oend[k] = (int)(p + f);
            if (sg.forced) {
                drops = i >= sg.pt_base + sg.n_first;               // (box of the kept REST: the host unites it with the anchor member's own box)
            } else {
                const unsigned win = (unsigned)((best_in_lds ? s_best[k - k_lo] : (best_pre ? best_pre[k] : db_best_cluster(k, ncl, roots, rhead, rnext, size, firstidx, rootmin))) >> 32);
                drops = win >= 5u && win < (unsigned)sg.n;
            }
        }
        if (f && !drops) f = 0u;                                    // segment keeps every point: its input box stays exact
        if (__any(f != 0u && cur >= 0 && k != cur)) {               // somebody leaves its segment: flush all
            db_flush_boxes(cur >= 0, cur, mn, mx, obounds, &slot_seg, slot_box);
            cur = -1;
            for (int a = 0; a < 3; ++a) {
                mn[a] = 1e300;
                mx[a] = -1e300;
            }
        }
        if (f) {
            cur = k;
            for (int a = 0; a < 3; ++a) {
                mn[a] = v[a] < mn[a] ? v[a] : mn[a];
                mx[a] = v[a] > mx[a] ? v[a] : mx[a];
            }
        }
    }
    db_flush_boxes(cur >= 0, cur, mn, mx, obounds, &slot_seg, slot_box);
    __syncthreads();
    if (tid < 6 && slot_seg >= 0) {
        if (tid < 3) atomicMin(&obounds[(size_t)slot_seg * 6 + tid], slot_box[tid]);
        else atomicMax(&obounds[(size_t)slot_seg * 6 + tid], slot_box[tid]);
    }
}

struct BdSeg {
    long long pt_base;
    int n, pad;
};
__global__ void k_seg_bounds(const double* __restrict__ pts, const BdSeg* __restrict__ segs, unsigned long long* __restrict__ ob) {
    const BdSeg sg = segs[blockIdx.y];
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < sg.n; i += gridDim.x * blockDim.x)
        for (int a = 0; a < 3; ++a) {
            double v = pts[(size_t)(sg.pt_base + i) * 3 + a];
            mn[a] = v < mn[a] ? v : mn[a];
            mx[a] = v > mx[a] ? v : mx[a];
        }
    for (int a = 0; a < 3; ++a) {
        mn[a] = wave_min_f64(mn[a]);
        mx[a] = wave_max_f64(mx[a]);
    }
    if ((threadIdx.x & 63) == 0 && mn[0] <= mx[0])
        for (int a = 0; a < 3; ++a) {
            atomicMin(&ob[(size_t)blockIdx.y * 6 + a], enc_f64(mn[a]));
            atomicMax(&ob[(size_t)blockIdx.y * 6 + 3 + a], enc_f64(mx[a]));
        }
}

void CloudOps::bounds(const double* src, std::vector<SegDesc>& segs) {
    const int K = (int)segs.size();
    if (!K) return;
    std::vector<BdSeg> hs(K);
    std::vector<unsigned long long> hb((size_t)K * 6);
    int maxn = 0;
    for (int k = 0; k < K; ++k) {
        hs[k] = BdSeg{segs[k].pt_base, segs[k].n, 0};
        maxn = std::max(maxn, segs[k].n);
        for (int a = 0; a < 6; ++a) hb[(size_t)k * 6 + a] = a < 3 ? ~0ull : 0ull;
    }
    geom.ensure((size_t)K * sizeof(BdSeg));
    obounds.ensure((size_t)K * 6);
    HIP_TRY(hipMemcpyAsync(geom.p, hs.data(), (size_t)K * sizeof(BdSeg), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(obounds.p, hb.data(), hb.size() * 8, hipMemcpyHostToDevice, s));
    for (int k0 = 0; k0 < K; k0 += 32768) {
        int nk = std::min(32768, K - k0);
        hipLaunchKernelGGL(k_seg_bounds, dim3(std::max(1u, std::min(cdiv(maxn, 256), 64u)), nk), dim3(256), 0, s, src,
                           (const BdSeg*)geom.p + k0, obounds.p + (size_t)k0 * 6);
    }
    HMSG_CHECK_LAUNCH();
    HIP_TRY(hipMemcpyAsync(hb.data(), obounds.p, hb.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int k = 0; k < K; ++k)
        for (int a = 0; a < 3; ++a) {
            segs[k].mn[a] = segs[k].n ? dec_f64(hb[(size_t)k * 6 + a]) : 0.0;
            segs[k].mx[a] = segs[k].n ? dec_f64(hb[(size_t)k * 6 + 3 + a]) : 0.0;
        }
}

__global__ void k_db_maxcell(const unsigned* __restrict__ cnt, long long NC, unsigned* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned v = i < NC ? cnt[i] : 0u;
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
    if ((threadIdx.x & 63) == 0 && v > 16u) atomicMax(out, v);
}
struct DbInit {
    unsigned* cnt;
    unsigned long long* best;
    int *ocount, *ostart, *ofirst, *rhead;
    unsigned long long* obounds;
    unsigned *ncl, *rep, *contested, *dropped, *counters;
    long long NC;
    int K;
    // the batch's tables (segment geometry, gather pieces, compaction blocks) come up from pinned host memory in this launch too
    // (upload_pinned's k_upload16 was a launch of its own in front of it)
    const uint4* up_src;
    uint4* up_dst;
    size_t up_n16;
};
__global__ void k_db_init(DbInit in) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t q = (size_t)i; q < in.up_n16; q += (size_t)gridDim.x * blockDim.x) in.up_dst[q] = in.up_src[q];
    if (i <= in.NC) in.cnt[i] = 0u;          // NC + 1 entries (scan sentinel); the other per-cell tables: k_db_scan
    if (i < in.K) {
        in.best[i] = 0ull;
        for (int a = 0; a < 6; ++a) in.obounds[(size_t)i * 6 + a] = a < 3 ? ~0ull : 0ull;
        in.ocount[i] = 0;
        in.ostart[i] = 0;
        in.ofirst[i] = 0;
        in.rhead[i] = -1;
        in.ncl[i] = 0u;
        in.rep[i] = INF32;
        in.contested[i] = 0u;
        in.dropped[i] = 0u;
    }
    if (i < 8) in.counters[i] = 0u;
}

__global__ void k_publish(const unsigned* __restrict__ src, int n, unsigned* __restrict__ dst_host, unsigned* __restrict__ flag_host,
                          unsigned seq) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst_host[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void Publisher::launch(hipStream_t s, const unsigned* src, size_t n) {
    buf.ensure(n + 32);                                       // (the flag sits behind the payload, on another cache line)
    if (buf.p != inited || buf.n != inited_n) {               // fresh pinned memory: the flag must not look raised
        buf.p[buf.n - 1] = 0u;
        inited = buf.p;
        inited_n = buf.n;
    }
    ++seq;
    stream = s;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, s, src, (int)n, buf.p, buf.p + buf.n - 1, seq);
    HMSG_CHECK_LAUNCH();
}
void Publisher::wait() {
    volatile unsigned* flag = buf.p + buf.n - 1;
    for (unsigned long long it = 1; *flag != seq; ++it) {
        if ((it & 0xfffffull) != 0) continue;
        // about once a millisecond: a stream that failed, or drained without the publish kernel having run, must surface as an
        // error instead of a host that spins for ever
        const hipError_t e = hipStreamQuery(stream);
        if (e == hipErrorNotReady) continue;
        if (e != hipSuccess) HIP_TRY(e);
        if (*flag != seq) throw hmsg_error{HMSG_ERR_HIP, "merge fold: the stream drained but the step's results were never published"};
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

bool CloudOps::regions_supported() {
    static const bool split_compact = getenv("HMSG_DB_COMPACT_SPLIT") != nullptr;
    return !split_compact && getenv("HMSG_DEBUG_NO_CROP") == nullptr;
}

long long CloudOps::dbscan_keep_largest(const double* src, const std::vector<SegDesc>& segs, double eps, int min_points,
                                        double* dst, std::vector<DbscanResult>& res, const unsigned char* core0,
                                        unsigned char* dst_core, const DbGather* gather,
                                        const std::function<void(const unsigned* d_res, int K)>& behind_publish) {
    const int K = (int)segs.size();
    res.assign(K, DbscanResult{});
    if (K == 0) return 0;
    const double cs = eps / std::sqrt(3.0) * (1.0 - 1e-7);
    std::vector<DbSeg> hs(K);
    // HMSG_DB_COMPACT_SPLIT=1: flags, scan and scatter as three launches (the form before round 4; kept for comparison runs).  Read
    // ONCE per process, for both of the decisions that hang on it: the legacy launches know neither cropped anchors nor output
    // regions (they index parent[cellid[i]] for every point).
    static const bool split_compact = getenv("HMSG_DB_COMPACT_SPLIT") != nullptr;
    // (SegDesc::forced needs the one-launch compaction; HMSG_DEBUG_NO_CROP=1 bins every anchor whole, as before round 4)
    const bool forced_ok = !split_compact && getenv("HMSG_DEBUG_NO_CROP") == nullptr;   // (per call: tests switch it)
    const bool regions_ok = forced_ok && gather && gather->pool_w && gather->poolcore_w;
    long long NC = 0, N = 0;
    long long span_lo = segs[0].pt_base, span_hi = segs[0].pt_base;
    for (int k = 0; k < K; ++k) {
        const SegDesc& sd = segs[k];
        DbSeg& g = hs[k];
        g.cs = cs;
        g.n = sd.n;
        g.pt_base = sd.pt_base;
        g.n_first = sd.n_first;
        g.forced = 0;
        g.out_mode = 0;
        g.out_off = 0;
        g.pad = 0;
        for (int a = 0; a < 3; ++a) g.cmn[a] = g.cmx[a] = 0.0;
        g.cell_base = NC;
        if (sd.out_mode != 0) {
            HMSG_REQUIRE(regions_ok && (sd.out_mode == 1 || sd.out_mode == 2), HMSG_ERR_INVALID, "dbscan: output regions need a writable pool (DbGather::pool_w)");
            g.out_mode = sd.out_mode;
            g.out_off = sd.out_off;
        }
        if (sd.n > 0) {
            double lo[3] = {sd.mn[0], sd.mn[1], sd.mn[2]}, hi[3] = {sd.mx[0], sd.mx[1], sd.mx[2]};
            if (sd.forced && core0 && forced_ok && sd.n_first > 0 && sd.n_first < sd.n) {
                // the grid covers the crop only (every point that is binned lies inside it)
                g.forced = 1;
                stat_forced += 1;
                if (g.out_mode == 2) stat_inplace += 1;
                stat_forced_first += sd.n_first;
                for (int a = 0; a < 3; ++a) {
                    g.cmn[a] = sd.cmn[a];
                    g.cmx[a] = sd.cmx[a];
                    lo[a] = std::max(lo[a], sd.cmn[a]);
                    hi[a] = std::max(lo[a], std::min(hi[a], sd.cmx[a]));
                }
            }
            g.ox = lo[0];
            g.oy = lo[1];
            g.oz = lo[2];
            g.nx = (int)std::floor((hi[0] - lo[0]) / cs) + 1;
            g.ny = (int)std::floor((hi[1] - lo[1]) / cs) + 1;
            g.nz = (int)std::floor((hi[2] - lo[2]) / cs) + 1;
        } else {
            g.ox = g.oy = g.oz = 0;
            g.nx = g.ny = g.nz = 1;
        }
        HMSG_REQUIRE(g.out_mode != 2 || g.forced, HMSG_ERR_INVALID, "dbscan: a segment runs in place only with its anchor member cropped (SegDesc::forced)");
        NC += (long long)g.nx * g.ny * g.nz;
        N += sd.n;
        span_lo = std::min(span_lo, sd.pt_base);
        span_hi = std::max(span_hi, sd.pt_base + sd.n);
    }
    // segments must tile [0, N) of src (callers concatenate)
    HMSG_REQUIRE(span_lo == 0 && span_hi == N, HMSG_ERR_INVALID, "dbscan: segments must tile the source buffer");
    HMSG_REQUIRE(NC < (1ll << 31) && N < (1ll << 31), HMSG_ERR_UNSUPPORTED, "dbscan batch too large");
    if (N == 0) return 0;
    // (pinned staging; the previous batch ended with a wait on the stream.  A gather table given on the host rides along.)
    static int n_cu = 0;
    if (!n_cu) {
        hipDeviceProp_t prop;
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        n_cu = std::max(1, prop.multiProcessorCount);
    }
    // The compaction's workgroups (DbBlk): chain 0 covers the mode-0 segments (dense output; a run of consecutive ones is cut into
    // blocks, no block straddles a segment of another mode), every segment with a region of its own is a chain of its own, and
    // the anchor member of a segment that runs in place gets no workgroup at all.  Blocks are numbered in batch order, so every
    // predecessor of a tile has a lower workgroup index (the look-back's progress argument, hmsg_common.h).
    std::vector<DbBlk> hblk;
    if (!split_compact) {
        long long Ns = 0;
        for (int k = 0; k < K; ++k) Ns += hs[k].out_mode == 2 ? hs[k].n - hs[k].n_first : hs[k].n;
        // one grid resident at once while a thread's keep bits fit a register (DBK_TRIPS trips of 256 points), more blocks beyond
        // (two blocks per CU measured best on the MI355X: 28.2 us per fold step with one, 23.8 with two, 24.5 with four, 29.3 with eight)
        const long long gt = std::max<long long>(std::min<long long>(cdiv(std::max<long long>(Ns, 1), 256), (long long)n_cu * 2),
                                                 cdiv(std::max<long long>(Ns, 1), 256ll * DBK_TRIPS));
        const long long per_block = std::min<long long>(256ll * DBK_TRIPS, (cdiv(std::max<long long>(Ns, 1), gt) + 255) / 256 * 256);
        auto cut = [&](long long lo, long long hi, int seg, int* tile) {
            for (long long q = lo; q < hi; q += per_block)
                hblk.push_back(DbBlk{q, (int)std::min<long long>(per_block, hi - q), (*tile)++, seg >= 0 ? -1 : 0, seg});
        };
        int t0 = 0;
        long long run_lo = 0, run_hi = 0;
        for (int k = 0; k < K; ++k) {
            const DbSeg& g = hs[k];
            if (g.n == 0) continue;
            if (g.out_mode == 0) {
                if (run_hi > run_lo && run_hi != g.pt_base) {
                    cut(run_lo, run_hi, -1, &t0);
                    run_lo = run_hi = g.pt_base;
                }
                if (run_hi == run_lo) run_lo = run_hi = g.pt_base;
                run_hi = g.pt_base + g.n;
            } else {
                if (run_hi > run_lo) cut(run_lo, run_hi, -1, &t0);
                run_lo = run_hi = 0;
                int t = 0;
                cut(g.pt_base + (g.out_mode == 2 ? g.n_first : 0), g.pt_base + g.n, k, &t);
            }
        }
        if (run_hi > run_lo) cut(run_lo, run_hi, -1, &t0);
        size_t next = (size_t)t0;                         // status words of the own-region chains behind chain 0's
        for (size_t b = 0; b < hblk.size(); ++b)
            if (hblk[b].seg >= 0 && hblk[b].tile == 0) {
                size_t nt = 1;
                while (b + nt < hblk.size() && hblk[b + nt].seg == hblk[b].seg) ++nt;
                for (size_t q = 0; q < nt; ++q) hblk[b + q].chain0 = (int)next;
                next += nt;
            }
    }
    // (pinned staging; the previous batch ended with a wait on the stream.  A gather table given on the host and the compaction's
    //  block table ride along.)
    const size_t geom_bytes = ((size_t)K * sizeof(DbSeg) + 15) & ~(size_t)15;
    const size_t cat_bytes = ((gather && gather->host_segs ? (size_t)gather->nsegs * sizeof(CatSeg) : 0) + 15) & ~(size_t)15;
    const size_t blk_bytes = (hblk.size() * sizeof(DbBlk) + 15) & ~(size_t)15;
    geom.ensure(geom_bytes + cat_bytes + blk_bytes);
    h_geom.ensure(geom_bytes + cat_bytes + blk_bytes);
    memcpy(h_geom.p, hs.data(), (size_t)K * sizeof(DbSeg));
    if (cat_bytes) memcpy(h_geom.p + geom_bytes, gather->host_segs, (size_t)gather->nsegs * sizeof(CatSeg));
    if (blk_bytes) memcpy(h_geom.p + geom_bytes + cat_bytes, hblk.data(), hblk.size() * sizeof(DbBlk));
    // (the copy itself rides in k_db_init below)
    DbGather ga_dev = gather ? *gather : DbGather{};
    if (cat_bytes) ga_dev.segs = (const CatSeg*)(geom.p + geom_bytes);
    const DbSeg* dsegs = (const DbSeg*)geom.p;
    const DbBlk* dblks = (const DbBlk*)(geom.p + geom_bytes + cat_bytes);
    segid.ensure(N); cellid.ensure(N); ord.ensure(N); core.ensure(N); label.ensure(N); flags.ensure(N); pos.ensure(N);
    cnt.ensure(NC + 1); start.ensure(NC + 1); cursor.ensure(NC); minidx.ensure(NC); firstidx.ensure(NC); size.ensure(NC);
    parent.ensure(NC);
    rootmin.ensure(NC);
    best.ensure(K);
    rep.ensure(K);
    rhead.ensure(K);
    rnext.ensure(NC);
    roots.ensure((size_t)K * DB_ROOTS);
    kres.ensure((size_t)K * 18 + 8);            // per segment: n_out (or output end) | n_clusters | contested | dropped | 6 x u64 box; 8 counter words; per segment: output start | end of the first member's output
    int* const d_ocount = (int*)kres.p;
    unsigned* const d_ncl = kres.p + K;
    unsigned* const d_contested = kres.p + 2 * (size_t)K;
    unsigned* const d_dropped = kres.p + 3 * (size_t)K;
    unsigned long long* const d_obounds = (unsigned long long*)(kres.p + 4 * (size_t)K);
    int* const d_ostart = (int*)(kres.p + (size_t)K * 16 + 8);
    int* const d_ofirst = d_ostart + K;     // per segment: output position of the first point behind the first member
    active.ensure(NC); hasanchor.ensure(NC);
    corelist.ensure((size_t)std::max<long long>(N, 1));
    cellpos.ensure((size_t)NC);
    cseg.ensure((size_t)NC);
    cellbox.ensure((size_t)std::min<long long>(NC, N) * 6);
    ccore.ensure((size_t)std::min<long long>(NC, N));
    const unsigned gN = cdiv(N, 256), gC = cdiv(NC, 256);
    {   // every per-cell / per-segment table initialised by one launch (was a dozen memsets per batch)
        DbInit in;
        in.cnt = cnt.p;
        in.best = best.p; in.obounds = d_obounds; in.ocount = d_ocount; in.ostart = d_ostart; in.ofirst = d_ofirst; in.rhead = rhead.p; in.ncl = d_ncl; in.rep = rep.p; in.contested = d_contested; in.dropped = d_dropped;
        in.counters = kres.p + (size_t)K * 16;   // [0] core cells, [1] active core cells
        in.NC = NC; in.K = K;
        in.up_src = (const uint4*)h_geom.p; in.up_dst = (uint4*)geom.p; in.up_n16 = (geom_bytes + cat_bytes + blk_bytes) / 16;
        hipLaunchKernelGGL(k_db_init, dim3(cdiv(NC + 1, 256)), dim3(256), 0, s, in);
    }
    int maxn = 0;
    for (auto& sd : segs) maxn = std::max(maxn, sd.n);
    hipLaunchKernelGGL(k_db_cell, dim3(gN), dim3(256), 0, s, src, N, segid.p, K, dsegs, cellid.p, cnt.p, ga_dev,
                       const_cast<double*>(src));
    HMSG_CHECK_LAUNCH();
    {   // development: the fullest cell of the batch (HMSG_DEBUG_MAXCELL=1; its points are same-address atomics in k_db_cell / k_db_fill)
        static const bool want = getenv("HMSG_DEBUG_MAXCELL") != nullptr;
        if (want) {
            static DevBuf<unsigned> mx;
            mx.ensure(1);
            HIP_TRY(hipMemsetAsync(mx.p, 0, 4, s));
            hipLaunchKernelGGL(k_db_maxcell, dim3(cdiv(NC, 256)), dim3(256), 0, s, (const unsigned*)cnt.p, NC, mx.p);
            unsigned hm = 0;
            HIP_TRY(hipMemcpyAsync(&hm, mx.p, 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            stat_maxcell_sum += hm;
            stat_maxcell_max = std::max(stat_maxcell_max, (double)hm);
        }
    }
    unsigned* const d_nc = kres.p + (size_t)K * 16;      // counters: [0] occupied cells, [3] counted points
    {   // cell starts (start[NC] = N: end sentinel) + the list of occupied cells + the per-cell tables, one launch
        const size_t ntiles = ((size_t)NC + 1 + 1023) / 1024;
        const unsigned epoch = hmsg_scan_epoch(scan_tmp, 2 * ntiles, s);
        DbCellTables tb;
        tb.cursor = cursor.p; tb.minidx = minidx.p; tb.firstidx = firstidx.p; tb.rootmin = rootmin.p; tb.size = size.p;
        tb.hasanchor = hasanchor.p; tb.cells = corelist.p; tb.cellpos = cellpos.p; tb.parent = parent.p; tb.cseg = cseg.p; tb.n_cells = d_nc;
        unsigned long long* const st = reinterpret_cast<unsigned long long*>(scan_tmp.p);
        hipLaunchKernelGGL(k_db_scan, dim3((unsigned)ntiles), dim3(256), 0, s, (const unsigned*)cnt.p, start.p, NC, st, st + ntiles, epoch, dsegs, K, tb);
        HMSG_CHECK_LAUNCH();
    }
    spts.ensure((size_t)N * 3);
    score.ensure(N);
    needy.ensure((size_t)std::max<long long>(N, 1));
    sidx.ensure((size_t)std::max<long long>(N, 1));
    nclist.ensure((size_t)std::max<long long>(N, 1));
    unsigned char* const pc_w = gather ? gather->poolcore_w : (unsigned char*)nullptr;
    {
    ProfScope ps(prof, s, "k_db_core", (double)N * 33.0, true);    // (k_db_fill + k_db_count: sorted copy, core flags -- one timed unit)
    hipLaunchKernelGGL(k_db_fill, dim3(gN), dim3(256), 0, s, src, N, (const long long*)cellid.p, (const unsigned*)start.p, cursor.p,
                       ord.p, spts.p, core0, min_points, needy.p, d_nc + 3, sidx.p, core.p, score.p, hasanchor.p);   // ord: slot of every point in the cell-sorted copy
    hipLaunchKernelGGL(k_db_count, dim3((unsigned)n_cu * 8u), dim3(256), 0, s, src, (const int*)segid.p, dsegs, (const long long*)cellid.p,
                       (const unsigned*)start.p, (const double*)spts.p, eps * eps, min_points, (const unsigned*)needy.p,
                       (const unsigned*)(d_nc + 3), core.p);
    }
    // persistent grid (8 blocks per CU): waves / threads stride over the cell list
    const unsigned gW = (unsigned)n_cu * 8u;
    hipLaunchKernelGGL(k_db_cellbox, dim3(gW), dim3(256), 0, s, (const double*)spts.p, (const int*)corelist.p,
                       (const unsigned*)d_nc, (const unsigned*)cnt.p, (const unsigned*)start.p, (const unsigned*)sidx.p,
                       score.p, (const unsigned char*)core.p, cellbox.p, ccore.p, (const int*)cseg.p, (const unsigned char*)hasanchor.p,
                       core0 ? rep.p : (unsigned*)nullptr, parent.p, dsegs, minidx.p, active.p, pc_w);
    {
        ProfScope ps(prof, s, "k_db_union/box", (double)N * 24.0, true);
        hipLaunchKernelGGL(k_db_union, dim3(gW), dim3(256), 0, s, (const int*)corelist.p, (const unsigned*)d_nc, (const int*)cseg.p, dsegs, K,
                           (const unsigned*)minidx.p, eps * eps, (const int*)cellpos.p, (const double*)cellbox.p, parent.p,
                           (const unsigned*)active.p, (const unsigned char*)hasanchor.p);
    }
    {
        ProfScope ps(prof, s, "k_db_union/scan", (double)N * 24.0);
        hipLaunchKernelGGL(k_db_union_scan, dim3(gW), dim3(256), 0, s, (const double*)spts.p, (const int*)corelist.p,
                           (const unsigned*)d_nc, (const int*)cseg.p, dsegs, K, (const unsigned*)cnt.p, (const unsigned*)start.p,
                           (const unsigned*)ord.p, (const unsigned char*)score.p, (const unsigned*)minidx.p, eps * eps,
                           (const int*)cellpos.p, (const double*)cellbox.p, parent.p, (const unsigned*)active.p,
                           (const unsigned char*)hasanchor.p);
    }
    hipLaunchKernelGGL(k_db_rootmin, dim3(256), dim3(256), 0, s, (const int*)corelist.p, (const unsigned*)d_nc,
                       parent.p, (const unsigned*)minidx.p, rootmin.p, (const int*)cseg.p, dsegs, K, d_ncl, (const unsigned*)ccore.p, size.p,
                       roots.p, rhead.p, rnext.p, (const unsigned*)needy.p, (const unsigned*)(d_nc + 3), (const unsigned char*)core.p, nclist.p, d_nc + 2);
    {
    ProfScope ps(prof, s, "k_db_label", (double)N * 28.0, true);
    hipLaunchKernelGGL(k_db_label, dim3(std::min(gN, (unsigned)n_cu * 8u)), dim3(256), 0, s, src, (const int*)segid.p, dsegs,
                       (const long long*)cellid.p, (const unsigned*)minidx.p, (const unsigned*)start.p, (const double*)spts.p,
                       (const unsigned char*)score.p, (const int*)cellpos.p, (const double*)cellbox.p, (const int*)parent.p,
                       (const unsigned*)rootmin.p, (const unsigned*)d_ncl, eps * eps, label.p, size.p, firstidx.p, d_contested,
                       (const unsigned*)nclist.p, (const unsigned*)(d_nc + 2));
    }
    static const bool dump_wanted = getenv("HMSG_DEBUG_DUMP") != nullptr;
    if (split_compact) {
    hipLaunchKernelGGL(k_db_pick, dim3(256), dim3(256), 0, s, (const int*)corelist.p, (const unsigned*)d_nc, (const int*)cseg.p, dsegs, K,
                       (const unsigned*)size.p, (const unsigned*)firstidx.p, (const unsigned*)rootmin.p, best.p);
    hipLaunchKernelGGL(k_db_flags, dim3(gN), dim3(256), 0, s, N, (const int*)segid.p, dsegs, (const int*)label.p,
                       (const unsigned long long*)best.p, flags.p, d_dropped, (const unsigned char*)core.p,
                       (const long long*)cellid.p, (const int*)parent.p);
    HMSG_CHECK_LAUNCH();
    hmsg_scan_u32(flags.p, pos.p, (size_t)N, s, scan_tmp, nullptr);
    {
    ProfScope ps(prof, s, "k_db_scatter", (double)N * 56.0, true);
    hipLaunchKernelGGL(k_db_scatter, dim3(std::min(gN, (unsigned)n_cu)), dim3(256), 0, s, src, N, (const int*)segid.p, dsegs, (const unsigned*)flags.p,
                       (const unsigned*)pos.p, dst, d_ocount, (const unsigned char*)core.p, dst_core, d_obounds, (const unsigned*)d_dropped);
    }
    } else {
        const unsigned gK = (unsigned)hblk.size();
        const unsigned epoch = hmsg_scan_epoch(scan_tmp, gK, s);
        // A fold step's segments have a handful of clusters each and the compaction picks the winner itself (db_best_cluster: no launch
        // for it).  A LARGE batch -- the per-object DBSCAN(0.05, 10) over every instance of the scene, the final pass -- is the opposite:
        // an instance of piled-up re-observations falls into hundreds of little clusters, and every one of the compaction's workgroups
        // that touches the segment walked that chain of roots for itself (k_db_compact 2.2 ms of hmsg_build_object_nodes' 7 ms of DBSCAN).
        // There the winners are picked once, by k_db_pick, and the launch it costs is nothing against the batch.
        static const int pick_env = getenv("HMSG_DEBUG_DB_PICK") ? atoi(getenv("HMSG_DEBUG_DB_PICK")) : -1;   // 1 / 0: always / never (tests)
        const bool pick_first = pick_env >= 0 ? pick_env != 0 : (N >= (1ll << 20) || K > 64);
        if (pick_first)
            hipLaunchKernelGGL(k_db_pick, dim3(256), dim3(256), 0, s, (const int*)corelist.p, (const unsigned*)d_nc, (const int*)cseg.p, dsegs, K,
                               (const unsigned*)size.p, (const unsigned*)firstidx.p, (const unsigned*)rootmin.p, best.p);
        ProfScope ps(prof, s, "k_db_scatter", (double)N * 56.0, true);
        if (gK)
            hipLaunchKernelGGL(k_db_compact, dim3(gK), dim3(256), 0, s, src, dblks, (const int*)segid.p, dsegs, (const int*)label.p,
                               (const unsigned*)d_ncl, (const int*)roots.p, (const int*)rhead.p, (const int*)rnext.p, (const unsigned*)size.p, (const unsigned*)firstidx.p, (const unsigned*)rootmin.p,
                               (const unsigned*)rep.p, dump_wanted ? flags.p : (unsigned*)nullptr, (const unsigned char*)core.p,
                               (const long long*)cellid.p, (const int*)parent.p, dst, d_ocount, d_ostart, d_ofirst, dst_core, d_obounds,
                               reinterpret_cast<unsigned long long*>(scan_tmp.p), epoch, gather ? gather->pool_w : (double*)nullptr,
                               gather ? gather->poolcore_w : (unsigned char*)nullptr, pick_first ? (const unsigned long long*)best.p : (const unsigned long long*)nullptr);
    }
    HMSG_CHECK_LAUNCH();
    {   // debug: HMSG_DEBUG_DBCALL=<n> dumps the n-th batch (inputs + per-point results) under HMSG_DEBUG_DUMP
        static long long call_no = 0;
        static const char* const want = getenv("HMSG_DEBUG_DBCALL");      // (read once: batches may run on a worker thread)
        static const char* const wantn = getenv("HMSG_DEBUG_DBMINN");     // ... or the first batch with at least that many points
        static bool dumped = false;
        if ((want && atoll(want) == call_no) || (wantn && !dumped && N >= atoll(wantn) && (dumped = true))) {
            hmsg_dump("db_pts", src, (size_t)N * 24, s);
            hmsg_dump("db_segs", geom.p, (size_t)K * sizeof(DbSeg), s);
            hmsg_dump("db_core", core.p, (size_t)N, s);
            hmsg_dump("db_label", label.p, (size_t)N * 4, s);
            hmsg_dump("db_keep", flags.p, (size_t)N * 4, s);
            hmsg_dump("db_cellid", cellid.p, (size_t)N * 8, s);
            if (core0) hmsg_dump("db_core0", core0, (size_t)N, s);
        }
        ++call_no;
    }
    // one copy brings back counts, cluster counts, contest flags and the boxes of the kept points
    pub.launch(s, (const unsigned*)kres.p, (size_t)K * 18 + 8);
    if (behind_publish && !split_compact) behind_publish((const unsigned*)kres.p, K);
    pub.wait();
    const unsigned* hres = pub.data();
    const unsigned long long* hb = reinterpret_cast<const unsigned long long*>(hres + (size_t)K * 4);
    stat_calls += 1;
    stat_points += N;
    stat_cells += NC;
    stat_core_cells += hres[(size_t)K * 16];
    stat_needy += hres[(size_t)K * 16 + 3];
    long long total = 0;
    for (int k = 0; k < K; ++k) {
        // (one launch: output end - output start, positions in the segment's chain; in place: the anchor member + the kept rest)
        const int n_out = hs[k].out_mode == 2 ? hs[k].n_first + (int)hres[k] : (int)hres[k] - (split_compact ? 0 : (int)hres[(size_t)K * 16 + 8 + k]);
        res[k].n_out = n_out;
        res[k].changed = n_out != segs[k].n;
        res[k].n_clusters = (int)hres[(size_t)K + k];
        res[k].contested = (int)hres[(size_t)2 * K + k];
        res[k].first_kept = -1;
        if (!split_compact && segs[k].n_first > 0)
            res[k].first_kept = hs[k].out_mode == 2 ? segs[k].n_first
                                : (segs[k].n_first < segs[k].n ? (int)hres[(size_t)K * 17 + 8 + k] - (int)hres[(size_t)K * 16 + 8 + k] : n_out);
        for (int a = 0; a < 3; ++a) {
            // unchanged: the input box is exact (and maybe tighter bookkeeping upstream relies on it bit for bit)
            res[k].mn[a] = !n_out ? 0.0 : (res[k].changed ? dec_f64(hb[(size_t)k * 6 + a]) : segs[k].mn[a]);
            res[k].mx[a] = !n_out ? 0.0 : (res[k].changed ? dec_f64(hb[(size_t)k * 6 + 3 + a]) : segs[k].mx[a]);
            if (hs[k].forced && res[k].changed) {
                // the anchor member whole (its own box) + the kept rest (the device's box; none kept: the untouched initial value)
                const bool any_rest = n_out > segs[k].n_first;
                res[k].mn[a] = any_rest ? std::min(segs[k].fmn[a], dec_f64(hb[(size_t)k * 6 + a])) : segs[k].fmn[a];
                res[k].mx[a] = any_rest ? std::max(segs[k].fmx[a], dec_f64(hb[(size_t)k * 6 + 3 + a])) : segs[k].fmx[a];
            }
        }
        total += n_out;
    }
    return total;
}

// ------------------------------------------------------------------------------------------ voxel_down_sample
struct VxSeg {
    double ox, oy, oz;
    int nx, ny, nz, n;
    long long word_base, pt_base;
};
__device__ __forceinline__ long long vx_cell(const VxSeg& g, double vs, const double* __restrict__ p, int& ix, int& iy, int& iz) {
    ix = (int)floor(__ddiv_rn(__dsub_rn(p[0], g.ox), vs));
    iy = (int)floor(__ddiv_rn(__dsub_rn(p[1], g.oy), vs));
    iz = (int)floor(__ddiv_rn(__dsub_rn(p[2], g.oz), vs));
    return ((long long)ix * g.ny + iy) * g.nz + iz;
}

__global__ void k_vx_mark(const double* __restrict__ pts, const VxSeg* __restrict__ segs, double vs,
                          unsigned long long* __restrict__ bitmap) {
    const VxSeg g = segs[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += gridDim.x * blockDim.x) {
        int ix, iy, iz;
        long long lin = vx_cell(g, vs, pts + (size_t)(g.pt_base + i) * 3, ix, iy, iz);
        atomicOr(&bitmap[g.word_base + (lin >> 6)], 1ull << (lin & 63));
    }
}
// sort key of every point = slot of its voxel (value = the point's index: the stable sort keeps input order)
__global__ void k_vx_keys(const double* __restrict__ pts, const VxSeg* __restrict__ segs, double vs,
                          const unsigned long long* __restrict__ bitmap, const unsigned* __restrict__ rank,
                          unsigned* __restrict__ keys, unsigned long long* __restrict__ vals) {
    const VxSeg g = segs[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += gridDim.x * blockDim.x) {
        const double* p = pts + (size_t)(g.pt_base + i) * 3;
        int ix, iy, iz;
        long long lin = vx_cell(g, vs, p, ix, iy, iz);
        long long wd = g.word_base + (lin >> 6);
        keys[g.pt_base + i] = rank[wd] + (unsigned)__popcll(bitmap[wd] & ((1ull << (lin & 63)) - 1ull));
        vals[g.pt_base + i] = (unsigned long long)(g.pt_base + i);
    }
}
// one lane per voxel: sequential float64 sum of its points in input order, / count (Open3D VoxelDownSample,
// o3d_voxel_down_sample in the oracle)
__global__ void k_vx_walk(const unsigned* __restrict__ off, const unsigned long long* __restrict__ idx, long long P,
                          const double* __restrict__ pts, double* __restrict__ out) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= P) return;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    const unsigned a = off[s], b = off[s + 1];
    for (unsigned r = a; r < b; ++r) {
        const double* p = pts + (size_t)idx[r] * 3;
        sx = __dadd_rn(sx, p[0]);
        sy = __dadd_rn(sy, p[1]);
        sz = __dadd_rn(sz, p[2]);
    }
    const double n = (double)(b - a);
    out[(size_t)s * 3 + 0] = __ddiv_rn(sx, n);
    out[(size_t)s * 3 + 1] = __ddiv_rn(sy, n);
    out[(size_t)s * 3 + 2] = __ddiv_rn(sz, n);
}
__global__ void k_vx_gather(const unsigned* __restrict__ rank, const VxSeg* __restrict__ segs, int K, unsigned* __restrict__ out) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) out[k] = segs[k].n ? rank[segs[k].word_base] : 0u;
}

long long CloudOps::voxel_down_sample(const double* src, const std::vector<SegDesc>& segs, double vs, double* dst,
                                      std::vector<int>& out_n) {
    const int K = (int)segs.size();
    out_n.assign(K, 0);
    if (K == 0) return 0;
    std::vector<VxSeg> hs(K);
    long long nwords = 0;
    int maxn = 0;
    for (int k = 0; k < K; ++k) {
        const SegDesc& sd = segs[k];
        VxSeg& g = hs[k];
        g.n = sd.n;
        g.pt_base = sd.pt_base;
        g.word_base = nwords;
        maxn = std::max(maxn, sd.n);
        if (sd.n == 0) {
            g.ox = g.oy = g.oz = 0;
            g.nx = g.ny = g.nz = 0;
            continue;
        }
        g.ox = sd.mn[0] - vs * 0.5;
        g.oy = sd.mn[1] - vs * 0.5;
        g.oz = sd.mn[2] - vs * 0.5;
        g.nx = (int)std::floor((sd.mx[0] - g.ox) / vs) + 2;
        g.ny = (int)std::floor((sd.mx[1] - g.oy) / vs) + 2;
        g.nz = (int)std::floor((sd.mx[2] - g.oz) / vs) + 2;
        nwords += ((long long)g.nx * g.ny * g.nz + 63) / 64;
    }
    if (nwords == 0) return 0;
    HMSG_REQUIRE(nwords < (1ll << 31), HMSG_ERR_UNSUPPORTED, "voxel_down_sample batch too large");
    geom.ensure((size_t)K * sizeof(VxSeg));
    HIP_TRY(hipMemcpyAsync(geom.p, hs.data(), (size_t)K * sizeof(VxSeg), hipMemcpyHostToDevice, s));
    const VxSeg* dsegs = (const VxSeg*)geom.p;
    vbitmap.ensure((size_t)nwords);
    vrank.ensure((size_t)nwords);
    HIP_TRY(hipMemsetAsync(vbitmap.p, 0, (size_t)nwords * 8, s));
    dim3 grid(std::max(1u, std::min(cdiv(maxn, 256), 2048u)), K);
    hipLaunchKernelGGL(k_vx_mark, grid, dim3(256), 0, s, src, dsegs, vs, vbitmap.p);
    HMSG_CHECK_LAUNCH();
    long long P = (long long)hmsg_bitmap_rank(vbitmap.p, vrank.p, (size_t)nwords, s, scan_tmp);
    // (segments tile the source buffer: point index = position in src)
    long long N = 0;
    for (auto& sd : segs) N = std::max(N, sd.pt_base + sd.n);
    HMSG_REQUIRE(N < (1ll << 32), HMSG_ERR_UNSUPPORTED, "voxel_down_sample batch too large");
    vsort.keys.ensure((size_t)N);
    vsort.vals.ensure((size_t)N);
    // (gaps between segments, if any, must not reach the sort: callers pass tiling segments; checked here)
    long long covered = 0;
    for (auto& sd : segs) covered += sd.n;
    HMSG_REQUIRE(covered == N, HMSG_ERR_INVALID, "voxel_down_sample: segments must tile the source buffer");
    hipLaunchKernelGGL(k_vx_keys, grid, dim3(256), 0, s, src, dsegs, vs, (const unsigned long long*)vbitmap.p,
                       (const unsigned*)vrank.p, vsort.keys.p, vsort.vals.p);
    HMSG_CHECK_LAUNCH();
    hmsg_sort_pairs(vsort, (size_t)N, bits_for((unsigned long long)P), s);
    voff.ensure((size_t)P + 1);
    hmsg_sort_segment_starts(vsort.res_keys, (size_t)N, voff.p, (unsigned)P, s);
    hipLaunchKernelGGL(k_vx_walk, dim3(cdiv((size_t)P, 64)), dim3(64), 0, s, (const unsigned*)voff.p,
                       (const unsigned long long*)vsort.res_vals, P, src, dst);
    HMSG_CHECK_LAUNCH();
    pos.ensure(K);
    hipLaunchKernelGGL(k_vx_gather, dim3(cdiv(K, 256)), dim3(256), 0, s, (const unsigned*)vrank.p, dsegs, K, pos.p);
    HMSG_CHECK_LAUNCH();
    std::vector<unsigned> st(K);
    HIP_TRY(hipMemcpyAsync(st.data(), pos.p, (size_t)K * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // empty segments share the next segment's word_base -> rank there equals the next start
    std::vector<long long> startv(K + 1, P);
    for (int k = 0; k < K; ++k)
        if (hs[k].n) startv[k] = st[k];
    for (int k = K - 1; k >= 0; --k)
        if (!hs[k].n) startv[k] = startv[k + 1];
    for (int k = 0; k < K; ++k) out_n[k] = (int)(startv[k + 1] - startv[k]);
    return P;
}

// ---- test hook (include/hmsg.h: hmsg_test_dbscan): the segmented keep-largest DBSCAN on caller-supplied clouds ------
extern "C" int hmsg_test_dbscan(const double* pts, int32_t K, const int64_t* sizes, double eps, int32_t min_points,
                                const uint8_t* core0, double* out_pts, int64_t* out_sizes, uint8_t* out_core, int32_t* out_info) {
    if (K < 0 || (K > 0 && (!sizes || !out_sizes)) || eps <= 0 || min_points <= 0) return HMSG_ERR_INVALID;
    hipStream_t s = nullptr;
    int rc = HMSG_OK;
    try {
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        {
            CloudOps ops;
            ops.s = s;
            std::vector<SegDesc> segs((size_t)K);
            long long N = 0;
            for (int k = 0; k < K; ++k) {
                segs[k].pt_base = N;
                segs[k].n = (int)sizes[k];
                N += sizes[k];
            }
            DevBuf<double> d_src, d_dst;
            DevBuf<unsigned char> d_c0, d_oc;
            d_src.alloc((size_t)std::max<long long>(N, 1) * 3);
            d_dst.alloc((size_t)std::max<long long>(N, 1) * 3);
            d_c0.alloc((size_t)std::max<long long>(N, 1));
            d_oc.alloc((size_t)std::max<long long>(N, 1));
            if (N) HIP_TRY(hipMemcpyAsync(d_src.p, pts, (size_t)N * 24, hipMemcpyHostToDevice, s));
            if (N && core0) HIP_TRY(hipMemcpyAsync(d_c0.p, core0, (size_t)N, hipMemcpyHostToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));
            ops.bounds(d_src.p, segs);
            std::vector<DbscanResult> res;
            const long long total = ops.dbscan_keep_largest(d_src.p, segs, eps, min_points, d_dst.p, res, core0 ? d_c0.p : nullptr, d_oc.p);
            if (total && out_pts) HIP_TRY(hipMemcpyAsync(out_pts, d_dst.p, (size_t)total * 24, hipMemcpyDeviceToHost, s));
            if (total && out_core) HIP_TRY(hipMemcpyAsync(out_core, d_oc.p, (size_t)total, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            for (int k = 0; k < K; ++k) {
                out_sizes[k] = res[(size_t)k].n_out;
                if (out_info) {
                    out_info[k * 3] = res[(size_t)k].changed;
                    out_info[k * 3 + 1] = res[(size_t)k].n_clusters;
                    out_info[k * 3 + 2] = res[(size_t)k].contested;
                }
            }
        }
    } catch (const hmsg_error& e) {
        fprintf(stderr, "hmsg_test_dbscan: %s\n", e.msg.c_str());
        rc = e.code;
    }
    if (s) (void)hipStreamDestroy(s);
    return rc;
}
