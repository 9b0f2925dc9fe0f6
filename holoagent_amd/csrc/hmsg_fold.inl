// The INCREMENTAL merge fold (A6, sequential merge): included by hmsg_merge.hip (one translation unit).
//
// seq_merge (graph_utils.py:1015-1038) folds one frame's 3-D masks into the instance list per step, and every step
// re-runs pcd_denoise_dbscan (graph_utils.py:667-679, 827-880) on the concatenation of each group of overlapping
// clouds -- clouds that grow to 10^5..10^6 points because nothing is de-duplicated.  The batch kernels of
// hmsg_cloudops.hip re-bin, re-sort and re-label ALL of those points every step (3*10^5 per step at 1000 frames)
// and rebuild the overlap grid of every changed cloud.  Here a step touches only the NEW points and what is within
// eps of them:
//
//   * ONE persistent spatial index for every live cloud, on a FIXED global lattice of cells of side eps/sqrt(3)
//     (two points of one cell are always neighbours): hash (cloud id, 4x4x4-cell brick) -> brick = 64 cell
//     descriptors -> per-cell record block (point copy + index inside the cloud + core flag) that grows by
//     doubling.  Frame masks are indexed in bulk before the fold; a step appends the kept points.
//     The same index answers the float32 overlap test of find_overlapping_ratio_faiss (graph_utils.py:620-662).
//   * a component {members in list order} whose first member A is an ANCHOR (a fixed point of this very DBSCAN
//     with one cluster: every point kept, exact core flags known) and larger than the rest together keeps A
//     INACTIVE: its cores stay core and stay connected (one super-node C_A), A comes first in the concatenation so
//     C_A has the smallest cluster id and wins every contested border point, all of A is kept.  Only the other
//     members' points are ACTIVE: neighbour counts (against all members' indices), re-counts of the anchor's
//     non-core points that have an active point within eps, connections of active cores (lock-free union-find
//     over active points + C_A), labels of active non-core points, cluster sizes / keep-largest, and the kept
//     active points are APPENDED behind A in place (capacity slack in the pool; A is never copied or re-binned).
//     (oracle/incremental_dbscan_proto.py states this step in numpy; tests/test_incremental_proto.py checks it
//      against the batch DBSCAN.)
//   * components without a usable anchor run the same kernels with every member active (no super-node);
//     the few large ones (a big cloud that is not a fixed point) go through the batch kernels.
// Everything is exact: tests/test_fold_incremental.py and the GPU suite compare this fold with the batch fold
// (HMSG_FOLD_LEGACY=1) bit for bit.

namespace {

constexpr unsigned F_NONE = 0xffffffffu;
constexpr unsigned long long F_EMPTY = ~0ull;
constexpr unsigned F_CORE = 1u, F_TOUCHED = 2u;

struct FRec {                // one indexed point
    double x, y, z;
    unsigned lidx;           // index inside its cloud (pool index = cloud offset + lidx: survives a relocation)
    unsigned flags;          // F_CORE (exact for anchors), F_TOUCHED (step scratch)
};
struct FCell {               // one lattice cell of one cloud
    unsigned ptr;            // first record
    unsigned cnt, cap;
    unsigned ncore;          // records flagged F_CORE
    unsigned pend, base;     // insertion in flight: reserved slots / first slot of this batch
    unsigned pad0, pad1;
};
struct FBrick {
    unsigned long long occ;  // cells with cnt > 0 (bit = lx*16 + ly*4 + lz)
    unsigned long long pad[7];
    FCell c[64];
};
struct FIndexDev {           // by value to every kernel
    unsigned long long* keys;
    unsigned* vals;
    unsigned hmask;
    FBrick* bricks;
    unsigned brick_cap;
    FRec* recs;
    unsigned rec_cap;
    unsigned* counters;      // [0] bricks used  [1] records used  [2] error bits  [3] cells touched by the insertion in flight
    double ox, oy, oz, cs;   // lattice
};
enum { FC_BRICKS = 0, FC_RECS = 1, FC_ERR = 2, FC_TOUCHED_CELLS = 3, FC_TOUCHED_RECS = 4, FC_ROOTS = 5, FC_N = 8 };
enum { FERR_BRICKS = 1, FERR_RECS = 2, FERR_TOUCHED = 4, FERR_WINNER = 8, FERR_HASH = 16 };

__device__ __forceinline__ unsigned long long f_key(unsigned id, int bx, int by, int bz) {
    return ((unsigned long long)id << 40) | ((unsigned long long)(unsigned)bx << 27) | ((unsigned long long)(unsigned)by << 14) |
           (unsigned long long)(unsigned)bz;
}
__device__ __forceinline__ unsigned f_hash(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (unsigned)k;
}
__device__ __forceinline__ void f_cell_of(const FIndexDev& ix, double x, double y, double z, int& cx, int& cy, int& cz) {
    cx = (int)floor(__ddiv_rn(__dsub_rn(x, ix.ox), ix.cs));
    cy = (int)floor(__ddiv_rn(__dsub_rn(y, ix.oy), ix.cs));
    cz = (int)floor(__ddiv_rn(__dsub_rn(z, ix.oz), ix.cs));
}
__device__ __forceinline__ unsigned f_find(const FIndexDev& ix, unsigned long long key) {
    unsigned h = f_hash(key) & ix.hmask;
    for (;;) {
        const unsigned long long k = ix.keys[h];
        if (k == key) return ix.vals[h];
        if (k == F_EMPTY) return F_NONE;
        h = (h + 1u) & ix.hmask;
    }
}
// Brick of `key`, created when absent.  The creator publishes the brick number right after its CAS; lanes that lost
// the race for the same key wait for it only AFTER the probe loop, i.e. after every lane of their own wave has left
// it (a creator in the same wave has stored by then; creators in other waves progress on their own).
__device__ __forceinline__ unsigned f_find_or_insert(const FIndexDev& ix, unsigned long long key) {
    unsigned h = f_hash(key) & ix.hmask;
    unsigned res = F_NONE, wait_slot = F_NONE;
    bool done = false;
    for (unsigned probe = 0; !done; ++probe) {
        unsigned long long k = __hip_atomic_load(&ix.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool won = false;
        if (k == F_EMPTY) {
            const unsigned long long old = atomicCAS(&ix.keys[h], F_EMPTY, key);
            won = old == F_EMPTY;
            k = won ? key : old;
        }
        if (won) {
            unsigned b = atomicAdd(&ix.counters[FC_BRICKS], 1u);
            if (b >= ix.brick_cap) {
                atomicOr(&ix.counters[FC_ERR], (unsigned)FERR_BRICKS);
                b = 0u;
            }
            __hip_atomic_store(&ix.vals[h], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            res = b;
            done = true;
        } else if (k == key) {
            wait_slot = h;
            done = true;
        } else {
            h = (h + 1u) & ix.hmask;
            if (probe > ix.hmask) {
                atomicOr(&ix.counters[FC_ERR], (unsigned)FERR_HASH);
                res = 0u;
                done = true;
            }
        }
    }
    if (wait_slot != F_NONE)
        do res = __hip_atomic_load(&ix.vals[wait_slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (res == F_NONE);
    return res;
}

__device__ __forceinline__ double f_dist2(double ax, double ay, double az, double bx, double by, double bz) {
    const double dx = __dsub_rn(ax, bx), dy = __dsub_rn(ay, by), dz = __dsub_rn(az, bz);
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
}

// ---------------------------------------------------------------------------------------------- insertion
// Points to insert are pool segments (bulk: the frame masks, outputs of the batch path) or the kept active points of
// a fold step (cloud id / position come from the step's tables).
struct FInsSeg {
    long long off;           // first point in the pool
    unsigned id;             // cloud id the records go under
    int n;
    unsigned t0;             // first item of this segment (prefix sum of n)
    unsigned lidx0;          // index inside the cloud of the segment's first point
    int use_core, pad;       // copy the pool's core flags into the records
};
struct FInsArgs {
    const FInsSeg* segs;     // bulk mode
    int nsegs;
    unsigned nitems;
    // step mode (segs == nullptr): item t = active point t, inserted when keep[t]
    const unsigned* keep;
    const unsigned* dst;     // pool index the step's emit wrote the point to
    const unsigned* item_id;
    const unsigned* item_lidx;
    const double* pool;
    const unsigned char* poolcore;
    unsigned* cellref;       // per item: brick * 64 + cell
    unsigned* slot;          // per item: slot inside this batch's part of the cell block
    unsigned* touched;       // cells that received points in this batch
};
struct FItem {
    bool valid;
    unsigned id, lidx, core;
    long long p;             // pool index
};
__device__ __forceinline__ FItem f_item(const FInsArgs& a, unsigned i) {
    FItem it;
    it.valid = false;
    it.id = it.lidx = it.core = 0u;
    it.p = 0;
    if (i >= a.nitems) return it;
    if (a.segs) {
        int lo = 0, hi = a.nsegs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (a.segs[mid].t0 <= i) lo = mid; else hi = mid - 1;
        }
        const FInsSeg sg = a.segs[lo];
        const unsigned j = i - sg.t0;
        it.valid = true;
        it.id = sg.id;
        it.lidx = sg.lidx0 + j;
        it.p = sg.off + j;
        it.core = sg.use_core ? (unsigned)a.poolcore[it.p] : 0u;
    } else if (a.keep[i]) {
        it.valid = true;
        it.id = a.item_id[i];
        it.lidx = a.item_lidx[i];
        it.p = (long long)a.dst[i];
        it.core = (unsigned)a.poolcore[it.p];
    }
    return it;
}
__device__ __forceinline__ void f_reserve(const FIndexDev& ix, const FInsArgs& a, unsigned i, unsigned id, double x, double y, double z) {
    int cx, cy, cz;
    f_cell_of(ix, x, y, z, cx, cy, cz);
    const unsigned b = f_find_or_insert(ix, f_key(id, cx >> 2, cy >> 2, cz >> 2));
    const unsigned local = (unsigned)(((cx & 3) << 4) | ((cy & 3) << 2) | (cz & 3));
    const unsigned cr = b * 64u + local;
    const unsigned s = atomicAdd(&ix.bricks[b].c[local].pend, 1u);
    a.cellref[i] = cr;
    a.slot[i] = s;
    if (s == 0u) a.touched[atomicAdd(&ix.counters[FC_TOUCHED_CELLS], 1u)] = cr;
}
__global__ void k_ix_reserve(FIndexDev ix, FInsArgs a) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const FItem it = f_item(a, i);
    if (!it.valid) return;
    f_reserve(ix, a, i, it.id, a.pool[it.p * 3], a.pool[it.p * 3 + 1], a.pool[it.p * 3 + 2]);
}
// one wave per touched cell: make room (blocks double), fix the counts
__global__ void k_ix_grow(FIndexDev ix, const unsigned* __restrict__ touched) {
    const int lane = threadIdx.x & 63;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6, n = ix.counters[FC_TOUCHED_CELLS];
    for (unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n; w += nwaves) {
        const unsigned cr = touched[w];
        FBrick& br = ix.bricks[cr >> 6];
        FCell& c = br.c[cr & 63u];
        const unsigned cnt = c.cnt, need = cnt + c.pend, cap = c.cap, optr = c.ptr;
        unsigned nptr = optr;
        if (need > cap) {
            // first block: exactly what is asked for (a frame mask never grows); later blocks double
            unsigned nc = cap == 0u ? need : max(2u * cap, need);
            if (lane == 0) {
                nptr = atomicAdd(&ix.counters[FC_RECS], nc);
                if ((unsigned long long)nptr + nc > ix.rec_cap) {
                    atomicOr(&ix.counters[FC_ERR], (unsigned)FERR_RECS);
                    nptr = 0u;
                }
            }
            nptr = __shfl(nptr, 0);
            for (unsigned k = lane; k < cnt; k += 64u) ix.recs[nptr + k] = ix.recs[optr + k];
            if (lane == 0) {
                c.ptr = nptr;
                c.cap = nc;
            }
        }
        if (lane == 0) {
            c.base = cnt;
            c.cnt = need;
            c.pend = 0u;
            atomicOr(&br.occ, 1ull << (cr & 63u));
        }
    }
}
__global__ void k_ix_write(FIndexDev ix, FInsArgs a) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0u) ix.counters[FC_TOUCHED_CELLS] = 0u;          // (nobody reads it in this launch)
    const FItem it = f_item(a, i);
    if (!it.valid) return;
    const unsigned cr = a.cellref[i];
    FCell& c = ix.bricks[cr >> 6].c[cr & 63u];
    FRec r;
    r.x = a.pool[it.p * 3];
    r.y = a.pool[it.p * 3 + 1];
    r.z = a.pool[it.p * 3 + 2];
    r.lidx = it.lidx;
    r.flags = it.core ? F_CORE : 0u;
    ix.recs[c.ptr + c.base + a.slot[i]] = r;
    if (it.core) atomicAdd(&c.ncore, 1u);
}

// ---------------------------------------------------------------------------------------------- wave traversal
// cells of brick (bx, by, bz) inside the inclusive cell window [lo, hi]
__device__ __forceinline__ unsigned long long f_window_mask(int bx, int by, int bz, const int* lo, const int* hi) {
    unsigned mx = 0, my = 0, mz = 0;
    for (int i = 0; i < 4; ++i) {
        mx |= (bx * 4 + i >= lo[0] && bx * 4 + i <= hi[0]) ? 1u << i : 0u;
        my |= (by * 4 + i >= lo[1] && by * 4 + i <= hi[1]) ? 1u << i : 0u;
        mz |= (bz * 4 + i >= lo[2] && bz * 4 + i <= hi[2]) ? 1u << i : 0u;
    }
    unsigned long long m = 0ull;
    for (int x = 0; x < 4; ++x)
        if (mx >> x & 1u)
            for (int y = 0; y < 4; ++y)
                if (my >> y & 1u) m |= (unsigned long long)mz << ((x << 4) | (y << 2));
    return m;
}
// squared distance from p to the cube of cell (cx, cy, cz): nearest / farthest corner (a point of the cell can sit a
// rounding error outside the nominal cube: callers compare with a margin)
__device__ __forceinline__ void f_cube_dist(const FIndexDev& ix, const double* p, int cx, int cy, int cz, double& dmin2, double& dmax2) {
    const int c[3] = {cx, cy, cz};
    const double o[3] = {ix.ox, ix.oy, ix.oz};
    dmin2 = 0.0;
    dmax2 = 0.0;
    for (int a = 0; a < 3; ++a) {
        const double lo = o[a] + (double)c[a] * ix.cs, hi = lo + ix.cs;
        const double gap = fmax(0.0, fmax(lo - p[a], p[a] - hi));
        const double far = fmax(p[a] - lo, hi - p[a]);
        dmin2 += gap * gap;
        dmax2 += far * far;
    }
}

// Wave-uniform walk over the cells of the clouds ids[0..k) inside the cell window [lo, hi] (at most 2 bricks per axis:
// hi - lo <= 4) and over the records of the cells `cellfn` selects:
//   cellfn(have_cell, j, cx, cy, cz, desc, cellref) -> records of this lane's cell to scan (0: none), called by ALL lanes
//       (one lane per cell of the current brick; have_cell false on idle lanes) -- it may use wave collectives;
//   recfn(valid, j, cell_lane, rec, rec_index) called by all lanes once per trip of 64 records (cell_lane = the lane
//       whose cell the record belongs to);
//   stop() wave-uniform: end the walk.
template <class CellFn, class RecFn, class StopFn>
__device__ __forceinline__ void f_walk(const FIndexDev& ix, const unsigned* __restrict__ ids, int k, const int* lo, const int* hi,
                                       CellFn cellfn, RecFn recfn, StopFn stop) {
    const int lane = threadIdx.x & 63;
    const int b0x = lo[0] >> 2, b0y = lo[1] >> 2, b0z = lo[2] >> 2;
    for (int g = 0; g < k; g += 8) {
        // probe: lane = (member, brick corner)
        const int j_l = g + (lane >> 3);
        unsigned bi = F_NONE;
        unsigned long long cand = 0ull;
        int bx = 0, by = 0, bz = 0;
        if (j_l < k) {
            bx = b0x + ((lane >> 2) & 1);
            by = b0y + ((lane >> 1) & 1);
            bz = b0z + (lane & 1);
            if (bx * 4 <= hi[0] && by * 4 <= hi[1] && bz * 4 <= hi[2]) {
                bi = f_find(ix, f_key(ids[j_l], bx, by, bz));
                if (bi != F_NONE) cand = ix.bricks[bi].occ & f_window_mask(bx, by, bz, lo, hi);
            }
        }
        unsigned long long owners = __ballot(cand != 0ull);
        while (owners) {
            const int o = __ffsll(owners) - 1;
            owners &= owners - 1ull;
            const unsigned bi_o = __shfl(bi, o);
            const unsigned long long cand_o = __shfl(cand, o);
            const int j_o = g + (o >> 3);
            const int bx_o = b0x + ((o >> 2) & 1), by_o = b0y + ((o >> 1) & 1), bz_o = b0z + (o & 1);
            const bool have = (cand_o >> lane) & 1ull;
            FCell d;
            d.ptr = d.cnt = d.cap = d.ncore = d.pend = d.base = d.pad0 = d.pad1 = 0u;
            if (have) d = ix.bricks[bi_o].c[lane];
            const int cx = bx_o * 4 + (lane >> 4), cy = by_o * 4 + ((lane >> 2) & 3), cz = bz_o * 4 + (lane & 3);
            const unsigned n = cellfn(have, j_o, cx, cy, cz, d, bi_o * 64u + (unsigned)lane);
            unsigned incl = n;                                   // records laid end to end over the lanes' cells
            for (int s = 1; s < 64; s <<= 1) {
                const unsigned up = __shfl_up(incl, s);
                if (lane >= s) incl += up;
            }
            const unsigned total = __shfl(incl, 63);
            for (unsigned t0 = 0; t0 < total; t0 += 64u) {
                const unsigned t = t0 + (unsigned)lane;
                int a = 0, b = 63;                               // first lane whose inclusive sum exceeds t
                for (int it = 0; it < 6; ++it) {
                    const int mid = (a + b) >> 1;
                    const unsigned v = __shfl(incl, mid);
                    if (v > t) b = mid; else a = mid + 1;
                }
                const int cl = min(a, 63);
                const unsigned c_incl = __shfl(incl, cl), c_n = __shfl(n, cl), c_ptr = __shfl(d.ptr, cl);
                const bool valid = t < total;
                const unsigned ri = valid ? c_ptr + (t - (c_incl - c_n)) : 0u;
                FRec r;
                r.x = r.y = r.z = 0.0;
                r.lidx = r.flags = 0u;
                if (valid) r = ix.recs[ri];
                recfn(valid, j_o, cl, r, ri);
                if (stop()) return;
            }
            if (stop()) return;
        }
    }
}

// ---------------------------------------------------------------------------------------------- overlap on the index
struct FOvCloud {
    long long off;           // points in the pool
    unsigned id;
    int n;
    float mn[3], mx[3];      // float32 AABB
};
struct FOvTask {             // count the points of cloud x that have a point of cloud y closer than r
    int x, y;
    int dep_n;               // second direction: points of the pair's smaller cloud (first direction's denominator)
    int blk0;
};
#define FOV_CHUNK 512
// find_overlapping_ratio_faiss (graph_utils.py:645-662): float32 (dx*dx + dy*dy) + dz*dz < r2 against the exact
// nearest neighbour == against SOME point.  Lanes test their point against the first records of its own cell (on a
// re-observed surface the witness sits there); the points that found none are walked by the whole wave, one by one.
__global__ void __launch_bounds__(256) k_f_overlap(FIndexDev ix, const double* __restrict__ pool, const FOvCloud* __restrict__ cl,
                                                   const FOvTask* __restrict__ tasks, int ntasks, float r2, float r,
                                                   unsigned* __restrict__ counts, const unsigned* __restrict__ dep_counts, double th) {
    const int ti = find_entry(tasks, ntasks, blockIdx.x);
    const FOvTask tk = tasks[ti];
    if (dep_counts && (double)dep_counts[ti] / (double)tk.dep_n > th) return;
    const FOvCloud X = cl[tk.x], Y = cl[tk.y];
    const int lane = threadIdx.x & 63;
    const int b0 = (int)(blockIdx.x - (unsigned)tk.blk0) * FOV_CHUNK;
    const int b1 = b0 + FOV_CHUNK < X.n ? b0 + FOV_CHUNK : X.n;
    const double reach = (double)r + 1e-4;                     // float32 rounding of the coordinates is ~1e-6 m
    unsigned local = 0;
    for (int i0 = b0; i0 < b1; i0 += (int)blockDim.x) {        // block-uniform trip count
        const int i = i0 + (int)threadIdx.x;
        double p[3] = {0, 0, 0};
        bool open = false;                                      // still undecided
        if (i < b1) {
            const double* q = pool + (size_t)(X.off + i) * 3;
            p[0] = q[0];
            p[1] = q[1];
            p[2] = q[2];
            const float x = (float)p[0], y = (float)p[1], z = (float)p[2];
            open = !(x < Y.mn[0] - r || x > Y.mx[0] + r || y < Y.mn[1] - r || y > Y.mx[1] + r || z < Y.mn[2] - r || z > Y.mx[2] + r);
        }
        int cx = 0, cy = 0, cz = 0;
        if (open) {
            f_cell_of(ix, p[0], p[1], p[2], cx, cy, cz);
            const unsigned b = f_find(ix, f_key(Y.id, cx >> 2, cy >> 2, cz >> 2));
            if (b != F_NONE) {
                const FCell& c = ix.bricks[b].c[((cx & 3) << 4) | ((cy & 3) << 2) | (cz & 3)];
                const unsigned ptr = c.ptr, m = min(c.cnt, 4u);
                const float x = (float)p[0], y = (float)p[1], z = (float)p[2];
                bool h = false;
                for (unsigned k = 0; k < m; ++k) {
                    const FRec& rc = ix.recs[ptr + k];
                    const float ddx = __fsub_rn(x, (float)rc.x), ddy = __fsub_rn(y, (float)rc.y), ddz = __fsub_rn(z, (float)rc.z);
                    h = h || __fadd_rn(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)), __fmul_rn(ddz, ddz)) < r2;
                }
                if (h) {
                    ++local;
                    open = false;
                }
            }
        }
        unsigned long long todo = __ballot(open);
        while (todo) {
            const int src = __ffsll(todo) - 1;
            todo &= todo - 1ull;
            const double q[3] = {__shfl(p[0], src), __shfl(p[1], src), __shfl(p[2], src)};
            const float x = (float)q[0], y = (float)q[1], z = (float)q[2];
            int lo[3], hi[3];
            const double o[3] = {ix.ox, ix.oy, ix.oz};
            for (int a = 0; a < 3; ++a) {
                lo[a] = (int)floor((q[a] - reach - o[a]) / ix.cs);
                hi[a] = (int)floor((q[a] + reach - o[a]) / ix.cs);
            }
            bool hit = false;
            f_walk(ix, &cl[tk.y].id, 1, lo, hi,
                   [&](bool have, int, int ccx, int ccy, int ccz, const FCell& d, unsigned) -> unsigned {
                       if (!have) return 0u;
                       double dmin2, dmax2;
                       f_cube_dist(ix, q, ccx, ccy, ccz, dmin2, dmax2);
                       return dmin2 > reach * reach ? 0u : d.cnt;
                   },
                   [&](bool valid, int, int, const FRec& rc, unsigned) {
                       bool h = false;
                       if (valid) {
                           const float ddx = __fsub_rn(x, (float)rc.x), ddy = __fsub_rn(y, (float)rc.y), ddz = __fsub_rn(z, (float)rc.z);
                           h = __fadd_rn(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)), __fmul_rn(ddz, ddz)) < r2;
                       }
                       if (__any(h)) hit = true;
                   },
                   [&]() { return hit; });
            if (hit && lane == src) ++local;
        }
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if (lane == 0 && local) atomicAdd(&counts[ti], local);
}

// ---------------------------------------------------------------------------------------------- the fold step
struct FMem {                // one member cloud of a component
    long long off;
    unsigned id;
    int n;
    unsigned t0;             // first active point (F_NONE: the inactive anchor)
    unsigned pad;
};
struct FComp {
    int m0, nm;              // members [m0, m0 + nm) of the member table, in list order
    int has_anchor;          // member m0 is an inactive anchor
    unsigned anchor_n;
    unsigned t0, nt;         // active points [t0, t0 + nt): the other members' points in concatenation order
    unsigned out_id;         // cloud id the kept active points are indexed under
    unsigned out_lidx0;      // index inside the output cloud of the first kept active point (anchor: |A|)
    long long out_off;       // pool position of the first kept active point
};
struct FRes {                // per component, read back
    unsigned n_kept, ncl, contested, pad;
    unsigned long long box[6];   // enc_f64 AABB of the kept active points
};
struct FTouched {
    unsigned rec, cellref, comp, pad;
};
struct FStep {               // by value to the step kernels
    const FComp* comps;
    const FMem* mems;
    const unsigned* mem_ids; // ids of all members, parallel to mems (contiguous per component)
    int ncomp;
    unsigned T;              // active points
    double* pool;
    unsigned char* poolcore;
    unsigned char* acore;    // [T]
    int* parent;             // [ncomp + T]: node c < ncomp = the anchor cluster of component c, ncomp + t = active point t
    unsigned* size;          // [ncomp + T] cluster sizes (at the roots)
    unsigned* first;         // [ncomp + T] 1 + first member (active index) of the cluster; 0 for an anchor cluster
    int* lab;                // [T] root of the point's cluster, -1 noise
    unsigned* keep;          // [T]
    unsigned* pos;           // [T] exclusive scan of keep
    unsigned* dst;           // [T] pool index of the kept point
    unsigned* item_id;       // [T]
    unsigned* item_lidx;     // [T]
    unsigned long long* best;// [ncomp]
    FRes* res;               // [ncomp]
    FTouched* touched;
    unsigned touched_cap;
    unsigned* roots;         // [ncomp + T]
    double eps2;
    int minpts;
};
__device__ __forceinline__ int f_comp_of(const FStep& st, unsigned t) {
    int lo = 0, hi = st.ncomp - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (st.comps[mid].t0 <= t) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// member of component c that holds active point t (members are laid out in order; the anchor has t0 = F_NONE)
__device__ __forceinline__ int f_mem_of(const FStep& st, const FComp& c, unsigned t) {
    int lo = c.m0 + (c.has_anchor ? 1 : 0), hi = c.m0 + c.nm - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (st.mems[mid].t0 <= t) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__device__ __forceinline__ int f_uf_find(int* parent, int x) {
    for (;;) {
        const int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        const int gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // halving: any ancestor is valid
        x = p;
    }
}
__device__ __forceinline__ void f_uf_union(int* parent, int a, int b) {
    a = f_uf_find(parent, a);
    b = f_uf_find(parent, b);
    for (;;) {
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }                                   // a > b: the larger root goes under the smaller (roots = smallest node of the cluster)
        if (atomicCAS(&parent[a], a, b) == a) return;
        a = f_uf_find(parent, a);
        b = f_uf_find(parent, b);
    }
}
__device__ __forceinline__ int f_uf_root_ro(const int* parent, int x) {      // read-only walk (no unions in flight)
    for (;;) {
        const int p = parent[x];
        if (p == x) return x;
        x = p;
    }
}

// (1) the anchor's non-core points that have an active point within eps: the only anchor points whose core status
//     can change.  Also initialises the per-component nodes.
__global__ void __launch_bounds__(256) k_f_touch(FIndexDev ix, FStep st) {
    const int lane = threadIdx.x & 63;
    const unsigned gt = blockIdx.x * blockDim.x + threadIdx.x;
    if (gt < (unsigned)st.ncomp) {
        const FComp c = st.comps[gt];
        st.parent[gt] = (int)gt;
        st.size[gt] = c.has_anchor ? c.anchor_n : 0u;
        st.first[gt] = c.has_anchor ? 0u : 0xffffffffu;
        st.best[gt] = 0ull;
        FRes r;
        r.n_kept = 0u;
        r.ncl = c.has_anchor ? 1u : 0u;
        r.contested = r.pad = 0u;
        for (int a = 0; a < 6; ++a) r.box[a] = a < 3 ? ~0ull : 0ull;
        st.res[gt] = r;
        if (c.has_anchor) st.roots[atomicAdd(&ix.counters[FC_ROOTS], 1u)] = gt;
    }
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
    for (unsigned t = gt >> 6; t < st.T; t += nwaves) {
        const int ci = f_comp_of(st, t);
        const FComp c = st.comps[ci];
        if (!c.has_anchor) continue;
        const FMem m = st.mems[f_mem_of(st, c, t)];
        const size_t pi = (size_t)(m.off + (t - m.t0)) * 3;
        const double p[3] = {st.pool[pi], st.pool[pi + 1], st.pool[pi + 2]};
        int cx, cy, cz;
        f_cell_of(ix, p[0], p[1], p[2], cx, cy, cz);
        const int lo[3] = {cx - 2, cy - 2, cz - 2}, hi[3] = {cx + 2, cy + 2, cz + 2};
        f_walk(ix, st.mem_ids + c.m0, 1, lo, hi,
               [&](bool have, int, int ccx, int ccy, int ccz, const FCell& d, unsigned) -> unsigned {
                   if (!have || d.ncore >= d.cnt) return 0u;
                   double dmin2, dmax2;
                   f_cube_dist(ix, p, ccx, ccy, ccz, dmin2, dmax2);
                   return dmin2 >= st.eps2 * (1.0 + 1e-9) + 1e-12 ? 0u : d.cnt;
               },
               [&](bool valid, int, int cl, const FRec& rc, unsigned ri) {
                   if (!valid || (rc.flags & F_CORE) || !(f_dist2(rc.x, rc.y, rc.z, p[0], p[1], p[2]) < st.eps2)) return;
                   const unsigned old = atomicOr(&ix.recs[ri].flags, F_TOUCHED);
                   if (old & F_TOUCHED) return;
                   const unsigned q = atomicAdd(&ix.counters[FC_TOUCHED_RECS], 1u);
                   if (q >= st.touched_cap) {
                       atomicOr(&ix.counters[FC_ERR], (unsigned)FERR_TOUCHED);
                       return;
                   }
                   int rx, ry, rz;                               // the record's cell (its descriptor holds the core count)
                   f_cell_of(ix, rc.x, rc.y, rc.z, rx, ry, rz);
                   const unsigned b = f_find(ix, f_key(st.mem_ids[c.m0], rx >> 2, ry >> 2, rz >> 2));
                   FTouched tr;
                   tr.rec = ri;
                   tr.cellref = b * 64u + (unsigned)(((rx & 3) << 4) | ((ry & 3) << 2) | (rz & 3));
                   tr.comp = (unsigned)ci;
                   tr.pad = 0u;
                   st.touched[q] = tr;
               },
               [&]() { return false; });
    }
}

// neighbours of p within eps over all members of a component (p's own record included), counted until `minpts`
__device__ __forceinline__ bool f_is_core(const FIndexDev& ix, const FStep& st, const FComp& c, const double* p) {
    const int lane = threadIdx.x & 63;
    int cx, cy, cz;
    f_cell_of(ix, p[0], p[1], p[2], cx, cy, cz);
    const unsigned* ids = st.mem_ids + c.m0;
    // the own cell first: every point of it is a neighbour
    int have = 0;
    for (int j0 = 0; j0 < c.nm; j0 += 64) {
        unsigned n = 0;
        if (j0 + lane < c.nm) {
            const unsigned b = f_find(ix, f_key(ids[j0 + lane], cx >> 2, cy >> 2, cz >> 2));
            if (b != F_NONE) n = ix.bricks[b].c[((cx & 3) << 4) | ((cy & 3) << 2) | (cz & 3)].cnt;
        }
        have += wave_sum_i32((int)n);
    }
    if (have >= st.minpts) return true;
    const int lo[3] = {cx - 2, cy - 2, cz - 2}, hi[3] = {cx + 2, cy + 2, cz + 2};
    f_walk(ix, ids, c.nm, lo, hi,
           [&](bool hv, int, int ccx, int ccy, int ccz, const FCell& d, unsigned) -> unsigned {
               unsigned scan = 0u, sure = 0u;
               if (hv && !(ccx == cx && ccy == cy && ccz == cz)) {
                   double dmin2, dmax2;
                   f_cube_dist(ix, p, ccx, ccy, ccz, dmin2, dmax2);
                   if (dmax2 < st.eps2 * (1.0 - 1e-9) - 1e-12) sure = d.cnt;             // the whole cell is in reach
                   else if (!(dmin2 >= st.eps2 * (1.0 + 1e-9) + 1e-12)) scan = d.cnt;
               }
               have += wave_sum_i32((int)sure);
               return scan;
           },
           [&](bool valid, int, int, const FRec& rc, unsigned) {
               const bool h = valid && f_dist2(rc.x, rc.y, rc.z, p[0], p[1], p[2]) < st.eps2;
               have += __popcll(__ballot(h));
           },
           [&]() { return have >= st.minpts; });
    return have >= st.minpts;
}

// (2) core flags of the active points; re-count (and promotion) of the touched anchor points
__global__ void __launch_bounds__(256) k_f_count(FIndexDev ix, FStep st) {
    const int lane = threadIdx.x & 63;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
    const unsigned n_touched = min(ix.counters[FC_TOUCHED_RECS], st.touched_cap);
    for (unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < st.T + n_touched; w += nwaves) {
        if (w < st.T) {
            const FComp c = st.comps[f_comp_of(st, w)];
            const FMem m = st.mems[f_mem_of(st, c, w)];
            const size_t pi = (size_t)(m.off + (w - m.t0)) * 3;
            const double p[3] = {st.pool[pi], st.pool[pi + 1], st.pool[pi + 2]};
            const bool core = f_is_core(ix, st, c, p);
            if (lane == 0) {
                st.acore[w] = core ? 1 : 0;
                st.parent[st.ncomp + w] = st.ncomp + (int)w;
                st.size[st.ncomp + w] = 0u;
                st.first[st.ncomp + w] = 0xffffffffu;
                st.lab[w] = -1;
            }
        } else {
            const FTouched tr = st.touched[w - st.T];
            const FComp c = st.comps[tr.comp];
            const FRec rc = ix.recs[tr.rec];
            const double p[3] = {rc.x, rc.y, rc.z};
            const bool core = f_is_core(ix, st, c, p);
            if (core && lane == 0) {                             // promoted: a border point of the anchor cluster becomes core
                atomicOr(&ix.recs[tr.rec].flags, F_CORE);
                atomicAdd(&ix.bricks[tr.cellref >> 6].c[tr.cellref & 63u].ncore, 1u);
                st.poolcore[st.mems[c.m0].off + rc.lidx] = 1;
            }
        }
    }
}

// one union per (trip, cell) with a witness: `hit` lanes of the same cell elect their lowest lane
template <class Fn>
__device__ __forceinline__ void f_per_cell_leader(bool hit, int cell_lane, Fn fn) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(hit);
    while (todo) {
        const int l = __ffsll(todo) - 1;
        const int key = __shfl(cell_lane, l);
        const unsigned long long same = __ballot(hit && cell_lane == key);
        if (lane == l) fn();
        todo &= ~same;
    }
}

// (3) connections of the active core points: to the anchor cluster (any anchor core within eps) and to each other.
//     Every core point looks for ONE witness per neighbouring (cloud, cell): core points of one cell are always connected.
__global__ void __launch_bounds__(256) k_f_link(FIndexDev ix, FStep st) {
    const int lane = threadIdx.x & 63;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
    for (unsigned t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; t < st.T; t += nwaves) {
        if (!st.acore[t]) continue;
        const int ci = f_comp_of(st, t);
        const FComp c = st.comps[ci];
        const int mi = f_mem_of(st, c, t);
        const FMem m = st.mems[mi];
        const size_t pi = (size_t)(m.off + (t - m.t0)) * 3;
        const double p[3] = {st.pool[pi], st.pool[pi + 1], st.pool[pi + 2]};
        int cx, cy, cz;
        f_cell_of(ix, p[0], p[1], p[2], cx, cy, cz);
        const int lo[3] = {cx - 2, cy - 2, cz - 2}, hi[3] = {cx + 2, cy + 2, cz + 2};
        const int me = st.ncomp + (int)t;
        bool in_anchor = false;                                  // wave-uniform: already connected to the anchor cluster
        f_walk(ix, st.mem_ids + c.m0, c.nm, lo, hi,
               [&](bool hv, int j, int ccx, int ccy, int ccz, const FCell& d, unsigned) -> unsigned {
                   const bool anch = c.has_anchor && j == 0;
                   unsigned scan = 0u;
                   bool own_anchor = false;
                   if (hv && !(anch && (in_anchor || d.ncore == 0u))) {
                       if (anch && ccx == cx && ccy == cy && ccz == cz) own_anchor = true;     // same cell: within eps
                       else {
                           double dmin2, dmax2;
                           f_cube_dist(ix, p, ccx, ccy, ccz, dmin2, dmax2);
                           if (!(dmin2 >= st.eps2 * (1.0 + 1e-9) + 1e-12)) scan = d.cnt;
                       }
                   }
                   if (__any(own_anchor)) {
                       if (lane == 0) f_uf_union(st.parent, me, ci);
                       in_anchor = true;
                       if (anch) scan = 0u;
                   }
                   return scan;
               },
               [&](bool valid, int j, int cl, const FRec& rc, unsigned) {
                   const bool anch = c.has_anchor && j == 0;
                   bool hit = false;
                   int node = ci;
                   if (valid && f_dist2(rc.x, rc.y, rc.z, p[0], p[1], p[2]) < st.eps2) {
                       if (anch) hit = !in_anchor && (rc.flags & F_CORE);
                       else {
                           const unsigned q = st.mems[c.m0 + j].t0 + rc.lidx;
                           hit = q != t && st.acore[q];
                           node = st.ncomp + (int)q;
                       }
                   }
                   f_per_cell_leader(hit, cl, [&]() { f_uf_union(st.parent, me, node); });
                   if (anch && __any(hit)) in_anchor = true;
               },
               [&]() { return false; });
    }
}

// (4) cluster bookkeeping of the core points, labels of the non-core points (the reaching cluster with the smallest
//     root = Open3D's first cluster; cores of two clusters in reach = contested), contest check of the touched
//     anchor points that stayed non-core.
__global__ void __launch_bounds__(256) k_f_label(FIndexDev ix, FStep st) {
    const int lane = threadIdx.x & 63;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
    const unsigned n_touched = min(ix.counters[FC_TOUCHED_RECS], st.touched_cap);
    for (unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < st.T + n_touched; w += nwaves) {
        const bool is_active = w < st.T;
        int ci;
        double p[3];
        unsigned self = F_NONE;
        if (is_active) {
            ci = f_comp_of(st, w);
            const FComp c0 = st.comps[ci];
            const FMem m = st.mems[f_mem_of(st, c0, w)];
            const size_t pi = (size_t)(m.off + (w - m.t0)) * 3;
            p[0] = st.pool[pi];
            p[1] = st.pool[pi + 1];
            p[2] = st.pool[pi + 2];
            self = w;
            if (st.acore[w]) {
                if (lane == 0) {
                    const int r = f_uf_root_ro(st.parent, st.ncomp + (int)w);
                    st.lab[w] = r;
                    atomicAdd(&st.size[r], 1u);
                    atomicMin(&st.first[r], w + 1u);
                    if (r == st.ncomp + (int)w) {
                        atomicAdd(&st.res[ci].ncl, 1u);
                        st.roots[atomicAdd(&ix.counters[FC_ROOTS], 1u)] = (unsigned)r;
                    }
                }
                continue;
            }
        } else {
            const FTouched tr = st.touched[w - st.T];
            ci = (int)tr.comp;
            const FRec rc = ix.recs[tr.rec];
            if (lane == 0) atomicAnd(&ix.recs[tr.rec].flags, ~F_TOUCHED);
            if (rc.flags & F_CORE) continue;                     // promoted by k_f_count
            p[0] = rc.x;
            p[1] = rc.y;
            p[2] = rc.z;
        }
        const FComp c = st.comps[ci];
        int cx, cy, cz;
        f_cell_of(ix, p[0], p[1], p[2], cx, cy, cz);
        const int lo[3] = {cx - 2, cy - 2, cz - 2}, hi[3] = {cx + 2, cy + 2, cz + 2};
        int best = 0x7fffffff;                                   // per lane: smallest root it saw a witness of
        bool multi = false;                                      // ... and whether it saw two different ones
        bool got_anchor = !is_active;                            // a touched anchor point is a border point of the anchor cluster
        if (!is_active) best = ci;
        f_walk(ix, st.mem_ids + c.m0, c.nm, lo, hi,
               [&](bool hv, int j, int ccx, int ccy, int ccz, const FCell& d, unsigned) -> unsigned {
                   const bool anch = c.has_anchor && j == 0;
                   if (!hv || (anch && (got_anchor || d.ncore == 0u))) return 0u;
                   double dmin2, dmax2;
                   f_cube_dist(ix, p, ccx, ccy, ccz, dmin2, dmax2);
                   return dmin2 >= st.eps2 * (1.0 + 1e-9) + 1e-12 ? 0u : d.cnt;
               },
               [&](bool valid, int j, int, const FRec& rc, unsigned) {
                   const bool anch = c.has_anchor && j == 0;
                   bool hit = false;
                   int r = -1;
                   if (valid && f_dist2(rc.x, rc.y, rc.z, p[0], p[1], p[2]) < st.eps2) {
                       if (anch) {
                           hit = (rc.flags & F_CORE) != 0u;
                           r = ci;
                       } else {
                           const unsigned q = st.mems[c.m0 + j].t0 + rc.lidx;
                           if (q != self && st.acore[q]) {
                               hit = true;
                               r = f_uf_root_ro(st.parent, st.ncomp + (int)q);
                           }
                       }
                   }
                   if (hit) {
                       if (best != 0x7fffffff && r != best) multi = true;
                       if (r < best) best = r;
                   }
                   if (anch && __any(hit)) got_anchor = true;
               },
               [&]() { return false; });
        int wbest = best;
        for (int o = 32; o > 0; o >>= 1) {
            const int u = __shfl_xor(wbest, o);
            wbest = u < wbest ? u : wbest;
        }
        const bool contest = __any(multi || (best != 0x7fffffff && best != wbest)) != 0;
        if (lane == 0) {
            if (is_active) {
                st.lab[w] = wbest == 0x7fffffff ? -1 : wbest;
                if (wbest != 0x7fffffff) {
                    atomicAdd(&st.size[wbest], 1u);
                    atomicMin(&st.first[wbest], w + 1u);
                }
            }
            if (contest && !st.res[ci].contested) st.res[ci].contested = 1u;
        }
    }
}

// (5) largest cluster per component (Counter.most_common: ties go to the cluster that appears first in point order)
__global__ void k_f_pick(FIndexDev ix, FStep st) {
    const unsigned n = ix.counters[FC_ROOTS];
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned r = st.roots[i];
        const unsigned sz = st.size[r];
        if (!sz) continue;
        const int ci = r < (unsigned)st.ncomp ? (int)r : f_comp_of(st, r - (unsigned)st.ncomp);
        atomicMax(&st.best[ci], ((unsigned long long)sz << 32) | (unsigned long long)(0xffffffffu - st.first[r]));
    }
}
// (6) graph_utils.py:853-880: keep the largest cluster unless there is none or it has fewer than 5 points
__global__ void k_f_keep(FIndexDev ix, FStep st) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0u) {                                               // (the lists of this step have been consumed)
        ix.counters[FC_ROOTS] = 0u;
        ix.counters[FC_TOUCHED_RECS] = 0u;
    }
    if (t >= st.T) return;
    const int ci = f_comp_of(st, t);
    const unsigned long long b = st.best[ci];
    bool keep = true;
    if ((unsigned)(b >> 32) >= 5u) {
        const unsigned f = 0xffffffffu - (unsigned)(b & 0xffffffffull);
        const int win = f == 0u ? ci : st.lab[f - 1u];
        keep = st.lab[t] == win;
        if (st.comps[ci].has_anchor && win != ci) atomicOr(&ix.counters[FC_ERR], (unsigned)FERR_WINNER);
    }
    st.keep[t] = keep ? 1u : 0u;
}
// (7) the kept active points go behind the anchor (or to the component's new cloud), their box and count come back,
//     and their index slots are reserved
__global__ void __launch_bounds__(256) k_f_emit(FIndexDev ix, FStep st, FInsArgs ins) {
    const int lane = threadIdx.x & 63;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = t < st.T;
    int ci = -1;
    bool kp = false;
    double v[3] = {0, 0, 0};
    if (in) {
        ci = f_comp_of(st, t);
        const FComp c = st.comps[ci];
        kp = st.keep[t] != 0u;
        const unsigned k = st.pos[t] - st.pos[c.t0];
        if (kp) {
            const FMem m = st.mems[f_mem_of(st, c, t)];
            const size_t pi = (size_t)(m.off + (t - m.t0)) * 3;
            const long long d = c.out_off + (long long)k;
            for (int a = 0; a < 3; ++a) {
                v[a] = st.pool[pi + a];
                st.pool[(size_t)d * 3 + a] = v[a];
            }
            st.poolcore[d] = st.acore[t];
            st.dst[t] = (unsigned)d;
            st.item_id[t] = c.out_id;
            st.item_lidx[t] = c.out_lidx0 + k;
            f_reserve(ix, ins, t, c.out_id, v[0], v[1], v[2]);
        }
        if (t == c.t0 + c.nt - 1u) st.res[ci].n_kept = k + (kp ? 1u : 0u);
    }
    // boxes: one set of atomics per (wave, component)
    unsigned long long todo = __ballot(kp);
    while (todo) {
        const int l = __ffsll(todo) - 1;
        const int key = __shfl(ci, l);
        const bool mine = kp && ci == key;
        const unsigned long long same = __ballot(mine);
        for (int a = 0; a < 3; ++a) {
            const double mn = wave_min_f64(mine ? v[a] : 1e300), mx = wave_max_f64(mine ? v[a] : -1e300);
            if (lane == l) {
                atomicMin(&st.res[key].box[a], enc_f64(mn));
                atomicMax(&st.res[key].box[3 + a], enc_f64(mx));
            }
        }
        todo &= ~same;
    }
}

}  // namespace

// ================================================================================================ host side
namespace {

struct Folder : Merger {
    // the persistent index
    DevBuf<unsigned long long> ix_keys;
    DevBuf<unsigned> ix_vals, ix_counters;
    DevBuf<FBrick> ix_bricks;
    DevBuf<FRec> ix_recs;
    FIndexDev ix;
    unsigned next_id = 1;
    // step scratch
    DevBuf<char> d_pack;             // [comps | mems | ids]
    PinnedBuf<char> h_pack;
    DevBuf<unsigned char> acore;
    DevBuf<int> parent, lab;
    DevBuf<unsigned> size, first, keep, pos, dst, item_id, item_lidx, roots, cellref, slot, touched_cells;
    DevBuf<unsigned long long> best;
    DevBuf<FRes> d_res;
    DevBuf<FTouched> touched;
    PinnedBuf<char> h_res;           // [FRes x ncomp | counters]
    DevBuf<FInsSeg> d_insseg;
    DevBuf<char> d_ovtab;            // overlap step: [clouds | tasks | counts]
    PinnedBuf<char> h_ovtab;
    int n_cu = 0;
    long long big_active = 1 << 16;  // components without an anchor and more active points than this use the batch kernels
    double fstat[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // steps, comps (anchor), comps (plain), comps (batch), active points, relocated points, touched
    static constexpr unsigned TOUCHED_CAP = 1u << 20;

    void index_init(long long total_points, long long n_masks, const double* lo, const double* hi) {
        const double cs = eps / std::sqrt(3.0) * (1.0 - 1e-7);
        // (the lattice origin sits 8 cells below every point: cell and brick coordinates of a 5x5x5 window stay positive)
        ix.cs = cs;
        ix.ox = lo[0] - 8.0 * cs;
        ix.oy = lo[1] - 8.0 * cs;
        ix.oz = lo[2] - 8.0 * cs;
        for (int a = 0; a < 3; ++a) {
            const double cells = (hi[a] - lo[a]) / cs + 24.0;
            HMSG_REQUIRE(cells < (a == 2 ? 4.0 * 16384.0 : 4.0 * 8192.0), HMSG_ERR_UNSUPPORTED, "merge: scene extent exceeds the fold index's lattice");
        }
        const size_t brick_cap = (size_t)(total_points / 5 + n_masks * 8 + 4096);
        size_t H = 1 << 16;
        while (H < brick_cap * 4) H <<= 1;
        const size_t rec_cap = std::min<size_t>((size_t)total_points * 16 + ((size_t)1 << 20), 0xfffffff0u);
        ix_keys.alloc(H);
        ix_vals.alloc(H);
        ix_bricks.alloc(brick_cap);
        ix_recs.alloc(rec_cap);
        ix_counters.alloc(FC_N);
        HIP_TRY(hipMemsetAsync(ix_keys.p, 0xff, H * 8, s));
        HIP_TRY(hipMemsetAsync(ix_vals.p, 0xff, H * 4, s));
        HIP_TRY(hipMemsetAsync(ix_bricks.p, 0, brick_cap * sizeof(FBrick), s));
        HIP_TRY(hipMemsetAsync(ix_counters.p, 0, FC_N * 4, s));
        ix.keys = ix_keys.p;
        ix.vals = ix_vals.p;
        ix.hmask = (unsigned)(H - 1);
        ix.bricks = ix_bricks.p;
        ix.brick_cap = (unsigned)brick_cap;
        ix.recs = ix_recs.p;
        ix.rec_cap = (unsigned)rec_cap;
        ix.counters = ix_counters.p;
        touched.ensure(TOUCHED_CAP);
        hipDeviceProp_t prop;
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        n_cu = std::max(1, prop.multiProcessorCount);
    }

    void ensure_items(size_t n) {
        cellref.ensure(n);
        slot.ensure(n);
        touched_cells.ensure(n);
    }

    // records of whole clouds (the frame masks; outputs of the batch path)
    void index_bulk(const std::vector<FInsSeg>& segs_in) {
        if (segs_in.empty()) return;
        std::vector<FInsSeg> segs = segs_in;
        unsigned total = 0;
        for (auto& sg : segs) {
            sg.t0 = total;
            total += (unsigned)sg.n;
        }
        if (!total) return;
        d_insseg.ensure(segs.size());
        HIP_TRY(hipMemcpyAsync(d_insseg.p, segs.data(), segs.size() * sizeof(FInsSeg), hipMemcpyHostToDevice, s));
        ensure_items(total);
        FInsArgs a;
        memset(&a, 0, sizeof(a));
        a.segs = d_insseg.p;
        a.nsegs = (int)segs.size();
        a.nitems = total;
        a.pool = pool.p;
        a.poolcore = poolcore.p;
        a.cellref = cellref.p;
        a.slot = slot.p;
        a.touched = touched_cells.p;
        hipLaunchKernelGGL(k_ix_reserve, dim3(cdiv(total, 256)), dim3(256), 0, s, ix, a);
        hipLaunchKernelGGL(k_ix_grow, dim3(std::min(cdiv((size_t)total * 64, 256), (unsigned)n_cu * 16u)), dim3(256), 0, s, ix,
                           (const unsigned*)touched_cells.p);
        hipLaunchKernelGGL(k_ix_write, dim3(cdiv(total, 256)), dim3(256), 0, s, ix, a);
        HMSG_CHECK_LAUNCH();
        HIP_TRY(hipStreamSynchronize(s));      // (segs is a stack copy source)
    }

    void check_errors(unsigned err) {
        if (!err) return;
        char msg[160];
        snprintf(msg, sizeof(msg), "merge: fold index error bits 0x%x (1 bricks, 2 records, 4 touched list, 8 winner, 16 hash)", err);
        throw hmsg_error{HMSG_ERR_UNSUPPORTED, msg};
    }

    // ---- overlap ratios on the index (same contract as Merger::overlap_ratios)
    void overlap_ratios_ix(const std::vector<Cloud>& L, const std::vector<std::pair<int, int>>& pairs, std::vector<double>& ratio,
                           double decide_th) {
        ratio.assign(pairs.size(), 0.0);
        if (pairs.empty()) return;
        const size_t P = pairs.size();
        std::vector<int> slot_of(L.size(), -1);
        std::vector<FOvCloud> g;
        std::vector<FOvTask> tasks(P * 2);
        unsigned nblk1 = 0, nblk2 = 0;
        for (size_t k = 0; k < P; ++k) {
            int a = pairs[k].first, b = pairs[k].second;
            for (int v : {a, b})
                if (slot_of[v] < 0) {
                    slot_of[v] = (int)g.size();
                    FOvCloud c;
                    c.off = L[v].off;
                    c.id = L[v].id;
                    c.n = L[v].n;
                    for (int q = 0; q < 3; ++q) {
                        c.mn[q] = (float)L[v].mn[q];
                        c.mx[q] = (float)L[v].mx[q];
                    }
                    g.push_back(c);
                }
            if (L[a].n > L[b].n) std::swap(a, b);
            tasks[k] = FOvTask{slot_of[a], slot_of[b], 0, (int)nblk1};
            tasks[P + k] = FOvTask{slot_of[b], slot_of[a], L[a].n, (int)nblk2};
            nblk1 += cdiv((size_t)L[a].n, FOV_CHUNK);
            nblk2 += cdiv((size_t)L[b].n, FOV_CHUNK);
        }
        const size_t off_t = (g.size() * sizeof(FOvCloud) + 15) & ~(size_t)15, off_c = off_t + tasks.size() * sizeof(FOvTask),
                     pack = off_c + tasks.size() * 4;
        h_ovtab.ensure(pack);
        d_ovtab.ensure(pack);
        memcpy(h_ovtab.p, g.data(), g.size() * sizeof(FOvCloud));
        memcpy(h_ovtab.p + off_t, tasks.data(), tasks.size() * sizeof(FOvTask));
        memset(h_ovtab.p + off_c, 0, tasks.size() * 4);
        HIP_TRY(hipMemcpyAsync(d_ovtab.p, h_ovtab.p, pack, hipMemcpyHostToDevice, s));
        const FOvCloud* const dg = (const FOvCloud*)d_ovtab.p;
        const FOvTask* const dt = (const FOvTask*)(d_ovtab.p + off_t);
        unsigned* const dc = (unsigned*)(d_ovtab.p + off_c);
        const float r = (float)radius;
        const float r2 = (float)(radius * radius);
        const size_t prof_idx = h->prof.ev.size();
        {
            ProfScope ps(h->prof, s, "k_f_overlap", 0.0);
            for (int dir = 0; dir < 2; ++dir) {
                const unsigned nb = dir ? nblk2 : nblk1;
                if (!nb) continue;
                const size_t o = (size_t)dir * P;
                hipLaunchKernelGGL(k_f_overlap, dim3(nb), dim3(256), 0, s, ix, (const double*)pool.p, dg, dt + o, (int)P, r2, r, dc + o,
                                   (dir && decide_th >= 0.0) ? (const unsigned*)dc : (const unsigned*)nullptr, decide_th);
            }
        }
        HMSG_CHECK_LAUNCH();
        h_counts.ensure(tasks.size());
        unsigned* hc = h_counts.p;
        HIP_TRY(hipMemcpyAsync(hc, dc, tasks.size() * 4, hipMemcpyDeviceToHost, s));
        spin.wait(s);
        double ov_work = 0;
        for (size_t k = 0; k < P; ++k) {
            const int na = std::min(L[pairs[k].first].n, L[pairs[k].second].n), nb = std::max(L[pairs[k].first].n, L[pairs[k].second].n);
            ratio[k] = std::max((double)hc[k] / (double)na, (double)hc[P + k] / (double)nb);
            ov_work += 12.0 * na;
            if (!(decide_th >= 0.0 && (double)hc[k] / (double)na > decide_th)) ov_work += 12.0 * nb;
        }
        if (h->prof.enabled && prof_idx < h->prof.ev.size()) h->prof.ev[prof_idx].work = ov_work;
    }

    long long pool_alloc(long long cap) {
        grow(pool, (size_t)pool_used * 3, (size_t)(pool_used + cap) * 3);
        grow(poolcore, (size_t)pool_used, (size_t)(pool_used + cap));
        const long long off = pool_used;
        pool_used += cap;
        HMSG_REQUIRE(pool_used < (1ll << 32), HMSG_ERR_UNSUPPORTED, "merge: point pool exceeds 2^32 points");
        return off;
    }

    // ---- one fold step: merge_3d_masks (graph_utils.py:918-956) on the persistent index
    std::vector<Cloud> fold_step(std::vector<Cloud> L, double th) {
        const int n = (int)L.size();
        if (n == 0) return L;
        auto tnow = [] { return std::chrono::steady_clock::now(); };
        auto t0 = tnow();
        auto lap = [&](int k) {
            auto t1 = tnow();
            tphase[k] += std::chrono::duration<double, std::milli>(t1 - t0).count();
            t0 = t1;
        };
        std::vector<std::pair<int, int>> pairs, known_pairs;
        std::vector<double> known, ratio;
        find_pairs(L, pairs, known, known_pairs);
        lap(1);
        overlap_ratios_ix(L, pairs, ratio, th);
        lap(2);
        CompList comps;
        make_components(L, pairs, ratio, known_pairs, known, th, comps);
        // classify the components that need a DBSCAN
        enum { SKIP = 0, STEP = 1, BATCH = 2 };
        std::vector<unsigned char> kind(comps.size(), SKIP);
        std::vector<FComp> fc;
        std::vector<FMem> fm;
        std::vector<unsigned> fids;
        std::vector<int> comp_slot(comps.size(), -1);           // index into fc / the batch segments
        unsigned T = 0;
        // batch path tables (as Merger::merge_3d_masks)
        std::vector<SegDesc> segs;
        std::vector<CatSeg> cat;
        long long cat_total = 0;
        unsigned cat_blocks = 0;
        std::vector<CatSeg> reloc;                               // anchors that outgrew their capacity
        unsigned reloc_blocks = 0;
        for (size_t c = 0; c < comps.size(); ++c) {
            const auto& mem = comps[c];
            if (mem.size() == 1 && (L[mem[0]].fixed || L[mem[0]].n == 0)) continue;
            long long tot = 0;
            for (int i : mem) tot += L[i].n;
            const Cloud& first = L[mem[0]];
            const bool anch = use_anchor && mem.size() > 1 && first.anchor && first.n > tot - first.n && minpts >= 5;
            const long long active = anch ? tot - first.n : tot;
            if (active == 0) continue;
            if (!anch && active > big_active) {
                kind[c] = BATCH;
                SegDesc sd;
                sd.pt_base = cat_total;
                sd.n = 0;
                bool any = false;
                for (int i : mem) {
                    if (L[i].n == 0) continue;
                    cat.push_back(CatSeg{L[i].off, cat_total, L[i].n, 0, (int)cat_blocks, 0});
                    cat_blocks += cdiv((size_t)L[i].n, CAT_CHUNK);
                    cat_total += L[i].n;
                    sd.n += L[i].n;
                    for (int a = 0; a < 3; ++a) {
                        sd.mn[a] = any ? std::min(sd.mn[a], L[i].mn[a]) : L[i].mn[a];
                        sd.mx[a] = any ? std::max(sd.mx[a], L[i].mx[a]) : L[i].mx[a];
                    }
                    any = true;
                }
                comp_slot[c] = (int)segs.size();
                segs.push_back(sd);
                fstat[3] += 1;
                continue;
            }
            kind[c] = STEP;
            comp_slot[c] = (int)fc.size();
            FComp q;
            memset(&q, 0, sizeof(q));
            q.m0 = (int)fm.size();
            q.has_anchor = anch ? 1 : 0;
            q.anchor_n = anch ? (unsigned)first.n : 0u;
            q.t0 = T;
            for (size_t k = 0; k < mem.size(); ++k) {
                const Cloud& cl = L[mem[k]];
                if (cl.n == 0) continue;
                FMem m;
                m.off = cl.off;
                m.id = cl.id;
                m.n = cl.n;
                m.pad = 0;
                if (anch && k == 0) m.t0 = F_NONE;
                else {
                    m.t0 = T;
                    T += (unsigned)cl.n;
                }
                fm.push_back(m);
                fids.push_back(cl.id);
            }
            q.nm = (int)fm.size() - q.m0;
            q.nt = T - q.t0;
            fc.push_back(q);
            fstat[anch ? 1 : 2] += 1;
            fstat[4] += (double)active;
        }
        // pool space: kept points of an anchor component go behind the anchor (relocated with double the room when
        // they do not fit), the other components get a new cloud with room to grow
        for (size_t c = 0; c < comps.size(); ++c) {
            if (kind[c] != STEP) continue;
            FComp& q = fc[(size_t)comp_slot[c]];
            if (q.has_anchor) {
                Cloud& A = L[comps[c][0]];
                if ((long long)A.n + q.nt > A.cap) {
                    const long long ncap = 2 * ((long long)A.n + q.nt);
                    const long long noff = pool_alloc(ncap);
                    reloc.push_back(CatSeg{A.off, noff, A.n, 1, (int)reloc_blocks, 0});
                    reloc_blocks += cdiv((size_t)A.n, CAT_CHUNK);
                    fstat[5] += A.n;
                    A.off = noff;
                    A.cap = (int)std::min<long long>(ncap, 0x7fffffff);
                    fm[(size_t)q.m0].off = noff;
                }
                q.out_id = A.id;
                q.out_lidx0 = (unsigned)A.n;
                q.out_off = A.off + A.n;
            } else {
                HMSG_REQUIRE(next_id < (1u << 24), HMSG_ERR_UNSUPPORTED, "merge: fold index ran out of cloud ids");
                q.out_id = next_id++;
                q.out_lidx0 = 0;
                q.out_off = pool_alloc(2ll * q.nt);
            }
        }
        if (!reloc.empty()) {
            d_cat.ensure(reloc.size());
            HIP_TRY(hipMemcpyAsync(d_cat.p, reloc.data(), reloc.size() * sizeof(CatSeg), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_concat, dim3(reloc_blocks), dim3(256), 0, s, (const double*)pool.p, (const CatSeg*)d_cat.p, (int)reloc.size(),
                               pool.p, (const unsigned char*)poolcore.p, poolcore.p);
            HMSG_CHECK_LAUNCH();
            HIP_TRY(hipStreamSynchronize(s));                    // (d_cat / reloc are reused below)
        }
        lap(3);
        // ---- the step kernels
        const int NCOMP = (int)fc.size();
        FRes* hres = nullptr;
        if (NCOMP) {
            const size_t off_m = (fc.size() * sizeof(FComp) + 15) & ~(size_t)15, off_i = off_m + ((fm.size() * sizeof(FMem) + 15) & ~(size_t)15),
                         pack = off_i + fids.size() * 4;
            h_pack.ensure(pack);
            d_pack.ensure(pack);
            memcpy(h_pack.p, fc.data(), fc.size() * sizeof(FComp));
            memcpy(h_pack.p + off_m, fm.data(), fm.size() * sizeof(FMem));
            memcpy(h_pack.p + off_i, fids.data(), fids.size() * 4);
            HIP_TRY(hipMemcpyAsync(d_pack.p, h_pack.p, pack, hipMemcpyHostToDevice, s));
            const size_t NT = (size_t)NCOMP + T;
            acore.ensure(T);
            lab.ensure(T);
            keep.ensure(T);
            pos.ensure(T);
            dst.ensure(T);
            item_id.ensure(T);
            item_lidx.ensure(T);
            parent.ensure(NT);
            size.ensure(NT);
            first.ensure(NT);
            roots.ensure(NT);
            best.ensure((size_t)NCOMP);
            d_res.ensure((size_t)NCOMP);
            ensure_items(T);
            FStep st;
            st.comps = (const FComp*)d_pack.p;
            st.mems = (const FMem*)(d_pack.p + off_m);
            st.mem_ids = (const unsigned*)(d_pack.p + off_i);
            st.ncomp = NCOMP;
            st.T = T;
            st.pool = pool.p;
            st.poolcore = poolcore.p;
            st.acore = acore.p;
            st.parent = parent.p;
            st.size = size.p;
            st.first = first.p;
            st.lab = lab.p;
            st.keep = keep.p;
            st.pos = pos.p;
            st.dst = dst.p;
            st.item_id = item_id.p;
            st.item_lidx = item_lidx.p;
            st.best = best.p;
            st.res = d_res.p;
            st.touched = touched.p;
            st.touched_cap = TOUCHED_CAP;
            st.roots = roots.p;
            st.eps2 = eps * eps;
            st.minpts = minpts;
            FInsArgs ins;
            memset(&ins, 0, sizeof(ins));
            ins.nitems = T;
            ins.keep = keep.p;
            ins.dst = dst.p;
            ins.item_id = item_id.p;
            ins.item_lidx = item_lidx.p;
            ins.pool = pool.p;
            ins.poolcore = poolcore.p;
            ins.cellref = cellref.p;
            ins.slot = slot.p;
            ins.touched = touched_cells.p;
            const unsigned gW = std::max(cdiv((size_t)std::max<unsigned>(T, (unsigned)NCOMP) * 64, 256), 1u);     // a wave per active point
            const unsigned gT = cdiv(std::max<unsigned>(T, 1u), 256);
            {
                ProfScope ps(h->prof, s, "k_f_count", (double)T * 24.0);
                hipLaunchKernelGGL(k_f_touch, dim3(gW), dim3(256), 0, s, ix, st);
                hipLaunchKernelGGL(k_f_count, dim3(gW), dim3(256), 0, s, ix, st);
            }
            {
                ProfScope ps(h->prof, s, "k_f_link", (double)T * 24.0);
                hipLaunchKernelGGL(k_f_link, dim3(gW), dim3(256), 0, s, ix, st);
            }
            {
                ProfScope ps(h->prof, s, "k_f_label", (double)T * 24.0);
                hipLaunchKernelGGL(k_f_label, dim3(gW), dim3(256), 0, s, ix, st);
            }
            hipLaunchKernelGGL(k_f_pick, dim3(std::max(1u, std::min(gT, 64u))), dim3(256), 0, s, ix, st);
            hipLaunchKernelGGL(k_f_keep, dim3(gT), dim3(256), 0, s, ix, st);
            HMSG_CHECK_LAUNCH();
            hmsg_scan_u32(keep.p, pos.p, (size_t)T, s, ops.scan_tmp, nullptr);
            hipLaunchKernelGGL(k_f_emit, dim3(gT), dim3(256), 0, s, ix, st, ins);
            hipLaunchKernelGGL(k_ix_grow, dim3(std::min(gW, (unsigned)n_cu * 16u)), dim3(256), 0, s, ix, (const unsigned*)touched_cells.p);
            hipLaunchKernelGGL(k_ix_write, dim3(gT), dim3(256), 0, s, ix, ins);
            HMSG_CHECK_LAUNCH();
            const size_t rb = (size_t)NCOMP * sizeof(FRes);
            h_res.ensure(rb + FC_N * 4);
            HIP_TRY(hipMemcpyAsync(h_res.p, d_res.p, rb, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(h_res.p + rb, ix_counters.p, FC_N * 4, hipMemcpyDeviceToHost, s));
            if (segs.empty()) spin.wait(s);
            hres = (FRes*)h_res.p;
        }
        // ---- the batch path for the few large components without an anchor
        std::vector<DbscanResult> res;
        long long batch_base = 0;
        if (!segs.empty()) {
            concat.ensure((size_t)cat_total * 3);
            d_cat.ensure(cat.size());
            HIP_TRY(hipMemcpyAsync(d_cat.p, cat.data(), cat.size() * sizeof(CatSeg), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_concat, dim3(cat_blocks), dim3(256), 0, s, (const double*)pool.p, (const CatSeg*)d_cat.p, (int)cat.size(),
                               concat.p, (const unsigned char*)nullptr, (unsigned char*)nullptr);
            HMSG_CHECK_LAUNCH();
            batch_base = pool_alloc(cat_total);
            ops.dbscan_keep_largest(concat.p, segs, eps, minpts, pool.p + (size_t)batch_base * 3, res, nullptr, poolcore.p + batch_base);
        }
        if (NCOMP) {
            const unsigned* cnt = (const unsigned*)(h_res.p + (size_t)NCOMP * sizeof(FRes));
            check_errors(cnt[FC_ERR]);
        }
        lap(4);
        // ---- new list in component order
        std::vector<Cloud> out;
        out.reserve(comps.size());
        std::vector<FInsSeg> to_index;
        long long cursor = batch_base;
        auto settle = [&](Cloud& k, bool changed, unsigned ncl, bool contested) {
            k.fixed = !changed || ncl == 1 || (ncl > 1 && !contested);
            k.anchor = k.fixed && ncl >= 1 && (changed || ncl == 1);
            k.fresh = true;
            k.raw = false;
            k.uid = next_uid++;
        };
        for (size_t c = 0; c < comps.size(); ++c) {
            const auto& mem = comps[c];
            if (kind[c] == SKIP) {
                Cloud k = L[mem[0]];
                k.fresh = false;
                k.fixed = true;
                out.push_back(k);
                continue;
            }
            if (kind[c] == BATCH) {
                const DbscanResult& r = res[(size_t)comp_slot[c]];
                Cloud k;
                if (mem.size() == 1 && !r.changed) {
                    k = L[mem[0]];
                    k.fresh = false;
                    k.fixed = true;
                    k.raw = false;
                    k.anchor = r.n_clusters == 1;
                } else {
                    k.n = r.n_out;
                    for (int a = 0; a < 3; ++a) {
                        k.mn[a] = r.mn[a];
                        k.mx[a] = r.mx[a];
                    }
                    settle(k, r.changed != 0, (unsigned)r.n_clusters, r.contested != 0);
                }
                k.off = cursor;
                k.cap = r.n_out;
                HMSG_REQUIRE(next_id < (1u << 24), HMSG_ERR_UNSUPPORTED, "merge: fold index ran out of cloud ids");
                k.id = next_id++;
                to_index.push_back(FInsSeg{k.off, k.id, k.n, 0, 0, 1, 0});
                cursor += r.n_out;
                out.push_back(k);
                continue;
            }
            const FComp& q = fc[(size_t)comp_slot[c]];
            const FRes& r = hres[comp_slot[c]];
            const bool changed = r.n_kept != q.nt;
            double bmn[3] = {0, 0, 0}, bmx[3] = {0, 0, 0};
            if (r.n_kept)
                for (int a = 0; a < 3; ++a) {
                    bmn[a] = dec_f64(r.box[a]);
                    bmx[a] = dec_f64(r.box[3 + a]);
                }
            if (q.has_anchor) {
                Cloud k = L[mem[0]];
                k.n += (int)r.n_kept;
                if (r.n_kept)
                    for (int a = 0; a < 3; ++a) {
                        k.mn[a] = std::min(k.mn[a], bmn[a]);
                        k.mx[a] = std::max(k.mx[a], bmx[a]);
                    }
                settle(k, changed, r.ncl, r.contested != 0);
                out.push_back(k);
                continue;
            }
            Cloud k;
            if (mem.size() == 1 && !changed) {          // DBSCAN kept every point: the same cloud, now known fixed
                k = L[mem[0]];
                k.fresh = false;
                k.fixed = true;
                k.raw = false;
                k.anchor = r.ncl == 1;
            } else {
                k.n = (int)r.n_kept;
                for (int a = 0; a < 3; ++a) {
                    k.mn[a] = bmn[a];
                    k.mx[a] = bmx[a];
                }
                settle(k, changed, r.ncl, r.contested != 0);
            }
            k.off = q.out_off;
            k.cap = (int)std::min<long long>(2ll * q.nt, 0x7fffffff);
            k.id = q.out_id;
            out.push_back(k);
        }
        if (!segs.empty()) {
            pool_used = cursor;                          // (the batch outputs were written consecutively from batch_base)
            index_bulk(to_index);
        }
        fstat[0] += 1;
        lap(5);
        return out;
    }
};

}  // namespace
