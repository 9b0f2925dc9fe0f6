// Exact nearest-neighbour search in the filtered global cloud (replaces scipy cKDTree.query(k=1) of
// graph.py:409, generic.py:181, graph.py:458, no distance cap).
//
//  * query in a cell that still holds a cloud point: ring expansion over the occupancy bitmap; the own
//    centroid bounds the answer, so the search ends after one or two rings.
//  * query in a cell whose voxel was deleted by remove_radius_outlier (every build-frame pixel falls in a
//    cell of the unfiltered cloud): the nearest survivor can be metres away and ring expansion costs
//    O(r^2) bitmap probes per query.  For those cells hmsg_finalize_map precomputes the exact candidate
//    set -- every cloud point p with mindist(p, cell) <= min_p' maxdist(p', cell) -- once per cell; a query
//    scans that short list.  (SURVEY hazard 15.)
#pragma once
#include "hmsg_common.h"

struct NNIndex {
    GridGeom g;
    const unsigned long long* bitmap;     // occupancy of the filtered cloud
    const unsigned* rank;
    const double* pts;
    const unsigned long long* bitmap_rm;  // cells of removed voxels (may be null: generic search only)
    const unsigned* rank_rm;
    const unsigned* cand_off;             // CSR over removed cells
    const int* cand;
};

struct NNBest {
    double d2;
    int idx;
    int ntie;      // candidates at exactly best.d2 (1 = unique minimum)
};

// queries whose nearest neighbour is a BIT-EQUAL tie: the host answers them like scipy's cKDTree would
// (hmsg_ckdtree.h) and k_nn_patch writes the answers back
struct TieRec {
    long long qid;          // position in the index array to patch
    double x, y, z;
};
struct TieList {
    unsigned* count;        // device counter (may run past cap: the batch is then redone with a larger buffer)
    TieRec* recs;
    unsigned cap;
};
__device__ __forceinline__ void tie_push(const TieList& t, long long qid, double x, double y, double z) {
    const unsigned k = atomicAdd(t.count, 1u);
    if (k < t.cap) t.recs[k] = TieRec{qid, x, y, z};
}

// The squared distance is evaluated exactly as scipy's cKDTree does for 3-D points (sqeuclidean_distance_double:
// ((dx*dx) + dy*dy) + dz*dz in float64, no FMA), so a candidate that is closer by even one ulp wins like it does
// there.  BIT-EQUAL distances are counted: the caller hands such queries to the host, which replays cKDTree's
// traversal (the first candidate it meets wins there); until then the lowest index stands in.
__device__ __forceinline__ void nn_consider(NNBest& best, int q, double d2) {
    if (d2 < best.d2) {
        best.d2 = d2;
        best.idx = q;
        best.ntie = 1;
    } else if (d2 == best.d2) {
        ++best.ntie;
        if (q < best.idx) best.idx = q;
    }
}
__device__ __forceinline__ double nn_dist2(const double* __restrict__ p, double qx, double qy, double qz) {
    double dx = __dsub_rn(p[0], qx), dy = __dsub_rn(p[1], qy), dz = __dsub_rn(p[2], qz);
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
}

// `ok` (optional, one byte per cloud point): only points with ok[q] != 0 take part (nearest neighbour inside a SUBSET of
// the cloud, e.g. one storey: graph.py:769-775 crops the floor cloud from the map, :1105 searches in it)
__device__ __forceinline__ void nn_column(const GridGeom& g, const unsigned long long* __restrict__ bitmap,
                                          const unsigned* __restrict__ rank, const double* __restrict__ pts, int ix, int iy,
                                          int z0, int z1, double qx, double qy, double qz, NNBest& best,
                                          const unsigned char* __restrict__ ok = nullptr) {
    if (ix < 0 || iy < 0 || ix >= g.nx || iy >= g.ny) return;
    z0 = z0 < 0 ? 0 : z0;
    z1 = z1 >= g.nz ? g.nz - 1 : z1;
    if (z1 < z0) return;
    long long colw = ((long long)ix * g.ny + iy) * (g.nzp >> 6);
    for (int w = z0 >> 6; w <= z1 >> 6; ++w) {
        unsigned long long word = bitmap[colw + w];
        if (!word) continue;
        int b0 = w * 64;
        int lo = z0 - b0 < 0 ? 0 : z0 - b0, hi = z1 - b0 > 63 ? 63 : z1 - b0;
        unsigned long long m = (hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull);
        unsigned long long sel = word & m;
        if (!sel) continue;
        unsigned base = rank[colw + w];
        while (sel) {
            int b = __ffsll(sel) - 1;
            sel &= sel - 1;
            int q = (int)(base + (unsigned)__popcll(word & ((1ull << b) - 1ull)));
            if (ok && !ok[q]) continue;
            nn_consider(best, q, nn_dist2(pts + (size_t)q * 3, qx, qy, qz));
        }
    }
}

// Ring expansion.  After ring r every cell of the cube [c-r, c+r]^3 has been examined; a point outside the
// cube is at least `m` away, m = distance from q to the nearest cube face that still has cells beyond it,
// so the search stops once best < m.
__device__ inline void nn_rings(const NNIndex& I, int cx, int cy, int cz, double qx, double qy, double qz, NNBest& best,
                                const unsigned char* __restrict__ ok = nullptr) {
    const GridGeom& g = I.g;
    const int rmax = max(g.nx, max(g.ny, g.nz));
    for (int r = 1; r <= rmax; ++r) {
        if (r == 1) {
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy) nn_column(g, I.bitmap, I.rank, I.pts, cx + dx, cy + dy, cz - 1, cz + 1, qx, qy, qz, best, ok);
        } else {
            for (int dx = -r; dx <= r; ++dx)
                for (int dy = -r; dy <= r; ++dy) {
                    bool rim = (dx == -r || dx == r || dy == -r || dy == r);
                    if (rim) {
                        nn_column(g, I.bitmap, I.rank, I.pts, cx + dx, cy + dy, cz - r, cz + r, qx, qy, qz, best, ok);
                    } else {
                        nn_column(g, I.bitmap, I.rank, I.pts, cx + dx, cy + dy, cz - r, cz - r, qx, qy, qz, best, ok);
                        nn_column(g, I.bitmap, I.rank, I.pts, cx + dx, cy + dy, cz + r, cz + r, qx, qy, qz, best, ok);
                    }
                }
        }
        double m = 1e300;
        bool open = false;
        if (cx - r > 0) { m = fmin(m, qx - (g.ox + (cx - r) * g.vs)); open = true; }
        if (cx + r < g.nx - 1) { m = fmin(m, (g.ox + (cx + r + 1) * g.vs) - qx); open = true; }
        if (cy - r > 0) { m = fmin(m, qy - (g.oy + (cy - r) * g.vs)); open = true; }
        if (cy + r < g.ny - 1) { m = fmin(m, (g.oy + (cy + r + 1) * g.vs) - qy); open = true; }
        if (cz - r > 0) { m = fmin(m, qz - (g.oz + (cz - r) * g.vs)); open = true; }
        if (cz + r < g.nz - 1) { m = fmin(m, (g.oz + (cz + r + 1) * g.vs) - qz); open = true; }
        if (!open) break;
        m -= 1e-9;   // centroids sit inside their cell only up to rounding
        if (best.idx >= 0 && m > 0.0 && best.d2 < m * m) break;
    }
}

// nearest neighbour among the points with ok[q] != 0 (ring expansion only: the candidate lists of deleted voxels are
// exact for the WHOLE cloud, not for a subset)
__device__ inline int nn_search_subset(const NNIndex& I, const unsigned char* __restrict__ ok, double qx, double qy, double qz,
                                       int* out_ntie) {
    const GridGeom& g = I.g;
    int cx, cy, cz;
    cell_of(g, qx, qy, qz, cx, cy, cz);
    NNBest best{1e300, -1, 0};
    cx = cx < 0 ? 0 : (cx >= g.nx ? g.nx - 1 : cx);
    cy = cy < 0 ? 0 : (cy >= g.ny ? g.ny - 1 : cy);
    cz = cz < 0 ? 0 : (cz >= g.nz ? g.nz - 1 : cz);
    nn_rings(I, cx, cy, cz, qx, qy, qz, best, ok);
    if (out_ntie) *out_ntie = best.ntie;
    return best.idx;
}

__device__ inline int nn_search(const NNIndex& I, double qx, double qy, double qz, double* out_d2 = nullptr,
                                int* out_ntie = nullptr) {
    const GridGeom& g = I.g;
    int cx, cy, cz;
    cell_of(g, qx, qy, qz, cx, cy, cz);
    NNBest best{1e300, -1, 0};
    const bool inside = cx >= 0 && cy >= 0 && cz >= 0 && cx < g.nx && cy < g.ny && cz < g.nz;
    bool done = false;
    if (inside && I.bitmap_rm) {
        long long lin = lin_of(g, cx, cy, cz);
        unsigned long long bit = 1ull << (lin & 63);
        unsigned long long wr = I.bitmap_rm[lin >> 6];
        if (wr & bit) {          // deleted voxel: precomputed candidate list
            unsigned r = I.rank_rm[lin >> 6] + (unsigned)__popcll(wr & (bit - 1ull));
            for (unsigned k = I.cand_off[r]; k < I.cand_off[r + 1]; ++k) {
                int q = I.cand[k];
                nn_consider(best, q, nn_dist2(I.pts + (size_t)q * 3, qx, qy, qz));
            }
            done = best.idx >= 0;
        }
    }
    if (!done) {
        cx = cx < 0 ? 0 : (cx >= g.nx ? g.nx - 1 : cx);
        cy = cy < 0 ? 0 : (cy >= g.ny ? g.ny - 1 : cy);
        cz = cz < 0 ? 0 : (cz >= g.nz ? g.nz - 1 : cz);
        nn_rings(I, cx, cy, cz, qx, qy, qz, best);
    }
    if (out_d2) *out_d2 = best.d2;
    if (out_ntie) *out_ntie = best.ntie;
    return best.idx;
}

// host side of the tie hand-over (hmsg_api.hip): resolve the listed queries with the restated cKDTree, patch `target`
struct TieBuf {
    DevBuf<unsigned> count;
    DevBuf<TieRec> recs;
    DevBuf<long long> pq;
    DevBuf<int> pv;
    unsigned cap = 0;
    void prepare(hipStream_t s, unsigned want_cap = 1u << 16) {
        if (cap < want_cap) {
            recs.alloc(want_cap);
            cap = want_cap;
        }
        if (!count.p) count.alloc(1);
        HIP_TRY(hipMemsetAsync(count.p, 0, 4, s));
    }
    TieList list() { return TieList{count.p, recs.p, cap}; }
};
// returns false when the list overflowed (cap was doubled: redo the search kernel and call again)
bool hmsg_resolve_ties(hmsg_ctx* h, TieBuf& tb, int* target);

static inline NNIndex hmsg_nn_index(const hmsg_ctx* h) {
    NNIndex I;
    I.g = h->grid;
    I.bitmap = h->bitmap.p;
    I.rank = h->rank.p;
    I.pts = h->pts.p;
    I.bitmap_rm = h->have_cand ? h->bitmap_rm.p : nullptr;
    I.rank_rm = h->rank_rm.p;
    I.cand_off = h->cand_off.p;
    I.cand = h->cand.p;
    return I;
}
