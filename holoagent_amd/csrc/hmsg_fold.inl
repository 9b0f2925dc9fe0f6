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
//     (two points of one cell are always neighbours): hash (cloud id, 4x4x4-cell brick) -> brick = occupancy masks +
//     64 cell descriptors -> per-cell record block (point copy + index inside the cloud + core flag) that grows by
//     doubling.  Frame masks are indexed in bulk before the fold; a step appends the kept points.
//     The same index answers the float32 overlap test of find_overlapping_ratio_faiss (graph_utils.py:620-662).
//   * ANCHORS.  A cloud that is a fixed point of this very DBSCAN with one cluster (every point kept, exact core
//     flags known) keeps its cores core (more points only raise counts) and mutually connected: in a component they
//     are ONE node of the union-find.  When the component's first member A is such a cloud and larger than the rest
//     together it is not even looked at: A comes first in the concatenation, so its cluster has the smallest id and
//     wins every contested border point, all of A is kept, and only the OTHER members' points are ACTIVE --
//     neighbour counts (against all members' indices), re-counts of the anchor's non-core points that have an
//     active point within eps, connections of active cores (lock-free union-find over active points + one node per
//     anchor), labels of active non-core points, cluster sizes / keep-largest -- and the kept active points are
//     APPENDED behind A in place (capacity slack in the pool; A is never copied or re-binned).
//     (oracle/incremental_dbscan_proto.py states this step in numpy; tests/test_incremental_proto.py checks it
//      against the batch DBSCAN.)
//   * components without a usable first anchor run the same kernels with every member active;
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
    unsigned long long sub;  // which of the cell's 4x4x4 sub-cells hold a record (bit = sx*16 + sy*4 + sz)
};
struct FBrick {
    unsigned long long occ;  // cells with records (bit = lx*16 + ly*4 + lz)
    unsigned long long cm;   // cells with a record flagged F_CORE
    unsigned long long ncm;  // cells with a record not flagged F_CORE
    unsigned long long pm;   // step scratch: cells of an anchor member with a point promoted in this step (not flagged in its record)
    unsigned long long pad[4];
    FCell c[64];
};
struct __attribute__((aligned(16))) FHashEnt {   // key and brick number come with ONE 16-byte load
    unsigned long long key;
    unsigned val, pad;
};
struct FIndexDev {           // by value to every kernel
    FHashEnt* tab;
    unsigned hmask;
    FBrick* bricks;
    unsigned brick_cap;
    FRec* recs;
    unsigned rec_cap;
    unsigned* counters;
    double ox, oy, oz, cs;   // lattice
};
enum { FC_BRICKS = 0, FC_RECS = 1, FC_ERR = 2, FC_TOUCHED_CELLS = 3, FC_TOUCHED_RECS = 4, FC_ROOTS = 5, FC_L_COUNT = 6, FC_L_TOUCH = 7,
       FC_L_LINK0 = 8, FC_L_LINK = 9, FC_L_LABEL = 10, FC_L_LINK2 = 24, FC_DIRTY = 25, FC_STAT = 11 /* 5 running totals of the lists */, FC_DBG = 32, FC_N = 64 };
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
__device__ __forceinline__ unsigned f_local(int cx, int cy, int cz) { return (unsigned)(((cx & 3) << 4) | ((cy & 3) << 2) | (cz & 3)); }
__device__ __forceinline__ unsigned f_find(const FIndexDev& ix, unsigned long long key) {
    unsigned h = f_hash(key) & ix.hmask;
    for (;;) {
        const FHashEnt e = ix.tab[h];
        if (e.key == key) return e.val;
        if (e.key == F_EMPTY) return F_NONE;
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
        unsigned long long k = __hip_atomic_load(&ix.tab[h].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool won = false;
        if (k == F_EMPTY) {
            const unsigned long long old = atomicCAS(&ix.tab[h].key, F_EMPTY, key);
            won = old == F_EMPTY;
            k = won ? key : old;
        }
        if (won) {
            unsigned b = atomicAdd(&ix.counters[FC_BRICKS], 1u);
            if (b >= ix.brick_cap) {
                atomicOr(&ix.counters[FC_ERR], (unsigned)FERR_BRICKS);
                b = 0u;
            }
            __hip_atomic_store(&ix.tab[h].val, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        do res = __hip_atomic_load(&ix.tab[wait_slot].val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (res == F_NONE);
    return res;
}

__device__ __forceinline__ double f_dist2(double ax, double ay, double az, double bx, double by, double bz) {
    const double dx = __dsub_rn(ax, bx), dy = __dsub_rn(ay, by), dz = __dsub_rn(az, bz);
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
}
// slot of this lane among the lanes of its wave that pass `want`, on a shared counter: ONE atomic per wave
// (call from wave-uniform code)
__device__ __forceinline__ unsigned f_wave_slot(bool want, unsigned* counter) {
    const int lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(want);
    if (!m) return 0u;
    const int leader = __ffsll(m) - 1;
    unsigned base = 0u;
    if (lane == leader) base = atomicAdd(counter, (unsigned)__popcll(m));
    base = __shfl(base, leader);
    return base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
}

// sub-cell of a point inside its cell (a quarter of the cell side per axis)
__device__ __forceinline__ unsigned long long f_sub_bit(const FIndexDev& ix, double x, double y, double z) {
    int cx, cy, cz;
    f_cell_of(ix, x, y, z, cx, cy, cz);
    const double q[3] = {x, y, z}, o[3] = {ix.ox, ix.oy, ix.oz};
    const int c[3] = {cx, cy, cz};
    int sidx = 0;
    for (int a = 0; a < 3; ++a) {
        int sa = (int)floor((q[a] - (o[a] + (double)c[a] * ix.cs)) / (ix.cs * 0.25));
        sa = sa < 0 ? 0 : (sa > 3 ? 3 : sa);
        sidx = sidx * 4 + sa;
    }
    return 1ull << sidx;
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
    // step mode (segs == nullptr): item t = active slot t, inserted when keep[t]
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
// reserve a slot for item i (all lanes call; `valid` lanes insert)
__device__ __forceinline__ void f_reserve(const FIndexDev& ix, const FInsArgs& a, bool valid, unsigned i, unsigned id, double x, double y,
                                          double z) {
    bool fresh = false;
    unsigned cr = 0u;
    if (valid) {
        int cx, cy, cz;
        f_cell_of(ix, x, y, z, cx, cy, cz);
        const unsigned b = f_find_or_insert(ix, f_key(id, cx >> 2, cy >> 2, cz >> 2));
        const unsigned local = f_local(cx, cy, cz);
        cr = b * 64u + local;
        const unsigned s = atomicAdd(&ix.bricks[b].c[local].pend, 1u);
        a.cellref[i] = cr;
        a.slot[i] = s;
        fresh = s == 0u;
    }
    const unsigned q = f_wave_slot(fresh, &ix.counters[FC_TOUCHED_CELLS]);
    if (fresh) a.touched[q] = cr;
}
__global__ void k_ix_reserve(FIndexDev ix, FInsArgs a) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const FItem it = f_item(a, i);
    double x = 0, y = 0, z = 0;
    if (it.valid) {
        x = a.pool[it.p * 3];
        y = a.pool[it.p * 3 + 1];
        z = a.pool[it.p * 3 + 2];
    }
    f_reserve(ix, a, it.valid, i, it.id, x, y, z);
}
// one LANE per touched cell: make room (first block: exactly what is asked for -- a frame mask never grows; later
// blocks double), fix the counts.  One allocation atomic per wave; blocks that move are copied by the whole wave.
__global__ void k_ix_grow(FIndexDev ix, const unsigned* __restrict__ touched) {
    const int lane = threadIdx.x & 63;
    const unsigned n = ix.counters[FC_TOUCHED_CELLS];
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned i0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += stride) {
        const unsigned i = i0 + (unsigned)lane;
        const bool valid = i < n;
        unsigned cr = 0u, cnt = 0u, need = 0u, cap = 0u, optr = 0u, nc = 0u;
        if (valid) {
            cr = touched[i];
            const FCell& c = ix.bricks[cr >> 6].c[cr & 63u];
            cnt = c.cnt;
            need = cnt + c.pend;
            cap = c.cap;
            optr = c.ptr;
            if (need > cap) nc = cap == 0u ? need : max(2u * cap, need);
        }
        unsigned incl = nc;
        for (int s = 1; s < 64; s <<= 1) {
            const unsigned up = __shfl_up(incl, s);
            if (lane >= s) incl += up;
        }
        const unsigned total = __shfl(incl, 63);
        unsigned base = 0u;
        if (total) {
            if (lane == 0) {
                base = atomicAdd(&ix.counters[FC_RECS], total);
                if ((unsigned long long)base + total > ix.rec_cap) {
                    atomicOr(&ix.counters[FC_ERR], (unsigned)FERR_RECS);
                    base = 0u;
                }
            }
            base = __shfl(base, 0);
        }
        const unsigned nptr = nc ? base + (incl - nc) : optr;
        unsigned long long movers = __ballot(nc != 0u && cnt != 0u);
        while (movers) {
            const int l = __ffsll(movers) - 1;
            movers &= movers - 1ull;
            const unsigned src = __shfl(optr, l), dstp = __shfl(nptr, l), m = __shfl(cnt, l);
            for (unsigned k = (unsigned)lane; k < m; k += 64u) ix.recs[dstp + k] = ix.recs[src + k];
        }
        if (valid) {
            FBrick& br = ix.bricks[cr >> 6];
            FCell& c = br.c[cr & 63u];
            if (nc) {
                c.ptr = nptr;
                c.cap = nc;
            }
            c.base = cnt;
            c.cnt = need;
            c.pend = 0u;
            const unsigned long long bit = 1ull << (cr & 63u);
            if (!(br.occ & bit)) atomicOr(&br.occ, bit);
        }
    }
}
__global__ void k_ix_write(FIndexDev ix, FInsArgs a) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0u) ix.counters[FC_TOUCHED_CELLS] = 0u;          // (nobody reads it in this launch)
    const FItem it = f_item(a, i);
    if (!it.valid) return;
    const unsigned cr = a.cellref[i];
    FBrick& br = ix.bricks[cr >> 6];
    FCell& c = br.c[cr & 63u];
    FRec r;
    r.x = a.pool[it.p * 3];
    r.y = a.pool[it.p * 3 + 1];
    r.z = a.pool[it.p * 3 + 2];
    r.lidx = it.lidx;
    r.flags = it.core ? F_CORE : 0u;
    ix.recs[c.ptr + c.base + a.slot[i]] = r;
    {
        const unsigned long long sb = f_sub_bit(ix, r.x, r.y, r.z);
        if (!(c.sub & sb)) atomicOr(&c.sub, sb);
    }
    const unsigned long long bit = 1ull << (cr & 63u);
    if (it.core) {
        atomicAdd(&c.ncore, 1u);
        if (!(br.cm & bit)) atomicOr(&br.cm, bit);
    } else if (!(br.ncm & bit)) atomicOr(&br.ncm, bit);
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
__device__ __forceinline__ int f_nth_bit(unsigned long long m, unsigned r) {
    for (unsigned i = 0; i < r; ++i) m &= m - 1ull;
    return __ffsll(m) - 1;
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
// can a record of the cell be within sqrt(lim2) of p at all?  (nearest corner of the OCCUPIED sub-cells, with a margin of
// a thousandth of a sub-cell for the rounding of the binning)
__device__ __forceinline__ bool f_sub_reach(const FIndexDev& ix, const double* p, int cx, int cy, int cz, unsigned long long sub, double lim2) {
    const double o[3] = {ix.ox + (double)cx * ix.cs, ix.oy + (double)cy * ix.cs, ix.oz + (double)cz * ix.cs};
    const double ss = ix.cs * 0.25, mg = ss * 1e-3;
    // per axis: squared gap to each of the 4 slabs
    double g[3][4];
    for (int a = 0; a < 3; ++a)
        for (int k = 0; k < 4; ++k) {
            const double lo = o[a] + (double)k * ss - mg, hi = lo + ss + 2.0 * mg;
            const double gap = fmax(0.0, fmax(lo - p[a], p[a] - hi));
            g[a][k] = gap * gap;
        }
    while (sub) {
        const int b = __ffsll(sub) - 1;
        sub &= sub - 1ull;
        if (g[0][b >> 4] + g[1][(b >> 2) & 3] + g[2][b & 3] < lim2) return true;
    }
    return false;
}
__device__ __forceinline__ unsigned f_wave_incl_scan(unsigned v) {
    const int lane = threadIdx.x & 63;
    for (int s = 1; s < 64; s <<= 1) {
        const unsigned up = __shfl_up(v, s);
        if (lane >= s) v += up;
    }
    return v;
}
// first lane whose inclusive sum exceeds t (per-lane t)
__device__ __forceinline__ int f_wave_search(unsigned incl, unsigned t) {
    int a = 0, b = 63;
    for (int it = 0; it < 6; ++it) {
        const int mid = (a + b) >> 1;
        const unsigned v = __shfl(incl, mid);
        if (v > t) b = mid; else a = mid + 1;
    }
    return min(a, 63);
}

enum { FSEL_SKIP = 0, FSEL_ALL = 1, FSEL_CORE = 2, FSEL_NONCORE = 3, FSEL_CORE_PROM = 4, FSEL_PROM = 5 };
// Wave-uniform walk over the cells of the clouds ids[0..k) inside the cell window [lo, hi] (at most 2 bricks per axis:
// hi - lo <= 4) and over the records of the cells `cellfn` selects.  The candidate cells of all clouds are compacted
// over the lanes (one lane per cell, 64 per round), their records laid end to end (64 per trip):
//   sel(j) -> FSEL_*: which cells of cloud j are candidates (all occupied / with core records / with non-core records);
//   cellfn(have, j, cx, cy, cz, desc, cellref) -> records of this lane's cell to scan (0: none); called by ALL lanes
//       (have false on idle lanes) -- it may use wave collectives;
//   recfn(valid, j, cell_lane, rec, rec_index): all lanes, once per trip (cell_lane = the lane whose cell the record is in);
//   stop(): wave-uniform, ends the walk.
//   budget > 0: a round takes only as many candidate cells as hold `budget` records (at least one) and re-offers the
//       rest to cellfn in the next round -- a witness found in one heavy cell then lets cellfn drop its neighbours;
//       cellfn must be free of side effects.
#ifndef F_WALK_STATS
#define F_WALK_STATS 0      /* development aid: per-walk statistics (HMSG_DEBUG_LINKSTEP / HMSG_DEBUG_COUNTSTEP) */
#endif
struct FWalkStat {            // development aid: what a walk did (wave-uniform counters)
    unsigned groups, rounds, trips, cells, recs;
};
template <class SelFn, class CellFn, class RecFn, class StopFn>
__device__ __forceinline__ void f_walk(const FIndexDev& ix, const unsigned* __restrict__ ids, int k, const int* lo, const int* hi, SelFn sel,
                                       CellFn cellfn, RecFn recfn, StopFn stop, unsigned budget = 0u, FWalkStat* ws = nullptr) {
    const int lane = threadIdx.x & 63;
    const int b0x = lo[0] >> 2, b0y = lo[1] >> 2, b0z = lo[2] >> 2;
    for (int g = 0; g < k; g += 8) {
        // probe: lane = (cloud, brick corner)
        const int j_l = g + (lane >> 3);
        unsigned bi = F_NONE;
        unsigned long long cand = 0ull;
        if (j_l < k) {
            const int kind = sel(j_l);
            const int bx = b0x + ((lane >> 2) & 1), by = b0y + ((lane >> 1) & 1), bz = b0z + (lane & 1);
            if (kind != FSEL_SKIP && bx * 4 <= hi[0] && by * 4 <= hi[1] && bz * 4 <= hi[2]) {
                bi = f_find(ix, f_key(ids[j_l], bx, by, bz));
                if (bi != F_NONE) {
                    const FBrick& br = ix.bricks[bi];
                    const unsigned long long m = kind == FSEL_ALL ? br.occ : (kind == FSEL_CORE ? br.cm : (kind == FSEL_NONCORE ? br.ncm : (kind == FSEL_PROM ? br.pm : (br.cm | br.pm))));
                    cand = m & f_window_mask(bx, by, bz, lo, hi);
                }
            }
        }
        const unsigned np = (unsigned)__popcll(cand);
        const unsigned cincl = f_wave_incl_scan(np);
        const unsigned ctotal = __shfl(cincl, 63);
        if (F_WALK_STATS && ws) {
            ws->groups += 1u;
            ws->cells += ctotal;
        }
        for (unsigned c0 = 0, took = 64u; c0 < ctotal; c0 += took) {
            const unsigned ci = c0 + (unsigned)lane;
            const bool have = ci < ctotal;
            const int o = f_wave_search(cincl, ci);                  // the probing lane that owns candidate ci
            const unsigned o_incl = __shfl(cincl, o), o_np = __shfl(np, o), o_bi = __shfl(bi, o);
            const unsigned long long o_cand = __shfl(cand, o);
            const int bit = have ? f_nth_bit(o_cand, ci - (o_incl - o_np)) : 0;
            const int j = g + (o >> 3);
            const int cx = (b0x + ((o >> 2) & 1)) * 4 + (bit >> 4), cy = (b0y + ((o >> 1) & 1)) * 4 + ((bit >> 2) & 3),
                      cz = (b0z + (o & 1)) * 4 + (bit & 3);
            FCell d;
            d.ptr = d.cnt = d.cap = d.ncore = d.pend = d.base = 0u;
            d.sub = 0ull;
            if (have) d = ix.bricks[o_bi].c[bit];
            unsigned n = cellfn(have, j, cx, cy, cz, d, o_bi * 64u + (unsigned)bit);
            unsigned rincl = f_wave_incl_scan(n);
            took = 64u;
            if (budget) {
                took = max(1u, (unsigned)__popcll(__ballot(rincl <= budget)));
                if ((unsigned)lane >= took) n = 0u;
                rincl = min(rincl, __shfl(rincl, (int)took - 1));
            }
            const unsigned rtotal = __shfl(rincl, 63);
            if (F_WALK_STATS && ws) {
                ws->rounds += 1u;
                ws->recs += rtotal;
                ws->trips += (rtotal + 63u) >> 6;
            }
            for (unsigned t0 = 0; t0 < rtotal; t0 += 64u) {
                const unsigned t = t0 + (unsigned)lane;
                const int cl = f_wave_search(rincl, t);
                const unsigned c_incl = __shfl(rincl, cl), c_n = __shfl(n, cl), c_ptr = __shfl(d.ptr, cl);
                const int c_j = __shfl(j, cl);
                const bool valid = t < rtotal;
                const unsigned ri = valid ? c_ptr + (t - (c_incl - c_n)) : 0u;
                FRec r;
                r.x = r.y = r.z = 0.0;
                r.lidx = r.flags = 0u;
                if (valid) r = ix.recs[ri];
                recfn(valid, c_j, cl, r, ri);
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
// is any record of [ptr, ptr + cnt) closer than r to (x, y, z) in float32?  four independent loads per step
__device__ __forceinline__ bool f_ov_scan(const FRec* __restrict__ recs, unsigned ptr, unsigned cnt, float x, float y, float z, float r2) {
    for (unsigned k = 0; k < cnt; k += 4u) {
        bool h = false;
#pragma unroll
        for (unsigned q = 0; q < 4u; ++q) {
            const FRec& rc = recs[ptr + min(k + q, cnt - 1u)];
            const float ddx = __fsub_rn(x, (float)rc.x), ddy = __fsub_rn(y, (float)rc.y), ddz = __fsub_rn(z, (float)rc.z);
            h = h || __fadd_rn(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)), __fmul_rn(ddz, ddz)) < r2;
        }
        if (h) return true;
    }
    return false;
}
// find_overlapping_ratio_faiss (graph_utils.py:645-662): float32 (dx*dx + dy*dy) + dz*dz < r2 against the exact
// nearest neighbour == against SOME point.  One lane per point of X: its own cell of Y first (on a re-observed
// surface the witness sits there), then the other cells within reach, brick by brick.
__global__ void __launch_bounds__(256) k_f_overlap(FIndexDev ix, const double* __restrict__ pool, const FOvCloud* __restrict__ cl,
                                                   const FOvTask* __restrict__ tasks, int ntasks, float r2, float r,
                                                   unsigned* __restrict__ counts, const unsigned* __restrict__ dep_counts, double th) {
    const int ti = find_entry(tasks, ntasks, blockIdx.x);
    const FOvTask tk = tasks[ti];
    if (dep_counts && (double)dep_counts[ti] / (double)tk.dep_n > th) return;
    const FOvCloud X = cl[tk.x], Y = cl[tk.y];
    const int b0 = (int)(blockIdx.x - (unsigned)tk.blk0) * FOV_CHUNK;
    const int b1 = b0 + FOV_CHUNK < X.n ? b0 + FOV_CHUNK : X.n;
    const double reach = (double)r + 1e-4;                     // float32 rounding of the coordinates is ~1e-6 m
    const double o[3] = {ix.ox, ix.oy, ix.oz};
    unsigned local = 0;
    for (int i = b0 + (int)threadIdx.x; i < b1; i += (int)blockDim.x) {
        const double* q = pool + (size_t)(X.off + i) * 3;
        const double p[3] = {q[0], q[1], q[2]};
        const float x = (float)p[0], y = (float)p[1], z = (float)p[2];
        if (x < Y.mn[0] - r || x > Y.mx[0] + r || y < Y.mn[1] - r || y > Y.mx[1] + r || z < Y.mn[2] - r || z > Y.mx[2] + r) continue;
        int c[3], lo[3], hi[3];
        f_cell_of(ix, p[0], p[1], p[2], c[0], c[1], c[2]);
        for (int a = 0; a < 3; ++a) {
            lo[a] = (int)floor((p[a] - reach - o[a]) / ix.cs);
            hi[a] = (int)floor((p[a] + reach - o[a]) / ix.cs);
        }
        const int b0x = lo[0] >> 2, b0y = lo[1] >> 2, b0z = lo[2] >> 2;
        const int own = (((c[0] >> 2) - b0x) << 2) | (((c[1] >> 2) - b0y) << 1) | ((c[2] >> 2) - b0z);
        bool hit = false;
        for (int qq = 0; qq < 8 && !hit; ++qq) {
            const int corner = qq ^ own;                       // the point's own brick first
            const int bx = b0x + ((corner >> 2) & 1), by = b0y + ((corner >> 1) & 1), bz = b0z + (corner & 1);
            if (bx * 4 > hi[0] || by * 4 > hi[1] || bz * 4 > hi[2]) continue;
            const unsigned b = f_find(ix, f_key(Y.id, bx, by, bz));
            if (b == F_NONE) continue;
            const FBrick& br = ix.bricks[b];
            unsigned long long m = br.occ & f_window_mask(bx, by, bz, lo, hi);
            if (qq == 0) {                                     // ... and in it the own cell
                const unsigned ob = f_local(c[0], c[1], c[2]);
                if (m >> ob & 1ull) {
                    m &= ~(1ull << ob);
                    hit = f_ov_scan(ix.recs, br.c[ob].ptr, br.c[ob].cnt, x, y, z, r2);
                }
            }
            while (m && !hit) {
                const int bit = __ffsll(m) - 1;
                m &= m - 1ull;
                double dmin2, dmax2;
                f_cube_dist(ix, p, bx * 4 + (bit >> 4), by * 4 + ((bit >> 2) & 3), bz * 4 + (bit & 3), dmin2, dmax2);
                if (dmin2 > reach * reach) continue;
                if (br.c[bit].cnt >= 8u && !f_sub_reach(ix, p, bx * 4 + (bit >> 4), by * 4 + ((bit >> 2) & 3), bz * 4 + (bit & 3), br.c[bit].sub, reach * reach)) continue;
                hit = f_ov_scan(ix.recs, br.c[bit].ptr, br.c[bit].cnt, x, y, z, r2);
            }
        }
        local += hit ? 1u : 0u;
    }
    for (int s = 32; s > 0; s >>= 1) local += __shfl_xor(local, s);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&counts[ti], local);
}

// ---------------------------------------------------------------------------------------------- the fold step
// Active slots: per component, per active member in list order, ONE slot for the member's anchor node followed by one
// slot per point -- slot order = concatenation order, and Open3D numbers clusters by their smallest core index, so
// "smallest node of the cluster" (the union-find's roots) orders clusters exactly like the reference.  (The anchor
// node of a member stands for all its own core points: no other cluster can have its first core inside that member.)
struct FMem {                // one member cloud of a component
    long long off;
    unsigned id;
    int n;
    unsigned t0;             // slot of the member's anchor node, its points follow (F_NONE: the inactive first anchor)
    unsigned am;             // anchor member: exact core flags in the pool / the records, one cluster
    double mn[3], mx[3];     // AABB
};
struct FComp {
    int m0, nm;              // members [m0, m0 + nm) of the member table, in list order
    int has_anchor;          // member m0 is an inactive anchor
    unsigned anchor_n;
    unsigned t0, nt;         // active slots [t0, t0 + nt)
    unsigned n_active;       // active points (nt minus the members' node slots)
    unsigned out_id;         // cloud id the kept active points are indexed under
    unsigned out_lidx0;      // index inside the output cloud of the first kept active point (anchor: |A|)
    unsigned pad;
    long long out_off;       // pool position of the first kept active point
};
struct FRes {                // per component, read back
    unsigned n_kept, ncl, contested, pad;
    unsigned long long box[6];   // enc_f64 AABB of the kept active points
};
struct FTouched {
    unsigned rec, cellref, comp, pad;
};
struct FSlotRec;
struct FStep {               // by value to the step kernels
    const FComp* comps;
    const FMem* mems;
    const unsigned* mem_ids; // ids of all members, parallel to mems (contiguous per component)
    int ncomp;
    unsigned T;              // active slots
    double* pool;
    unsigned char* poolcore;
    unsigned char* acore;    // [T] core flag of the slot's point in this DBSCAN
    int* parent;             // [ncomp + T]: node c < ncomp = the first anchor's cluster of component c, ncomp + t = slot t
    unsigned* size;          // [ncomp + T] cluster sizes (at the roots)
    unsigned* first;         // [ncomp + T] 1 + first member (slot) of the cluster; 0 for a first anchor's cluster
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
    unsigned *list_count, *list_touch, *list_link0, *list_link, *list_link2, *list_label;   // [T] slots that need a walk
    unsigned* dirty;         // bricks with a promoted-point mask to clear
    unsigned dirty_cap;
    struct FSlotRec* slotrec; // [T] written by k_f_pre
    double eps, eps2;
    int minpts, debug;
    unsigned long long* dbgbuf;   // development aid: per listed link slot, what its walk did
};
__device__ __forceinline__ int f_comp_of(const FStep& st, unsigned t) {
    int lo = 0, hi = st.ncomp - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (st.comps[mid].t0 <= t) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// member of component c that holds slot t (members are laid out in order; the inactive anchor has t0 = F_NONE)
__device__ __forceinline__ int f_mem_of(const FStep& st, const FComp& c, unsigned t) {
    int lo = c.m0 + (c.has_anchor ? 1 : 0), hi = c.m0 + c.nm - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (st.mems[mid].t0 <= t) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__device__ __forceinline__ int f_uf_find(int* parent, int x, unsigned* hops = nullptr) {
    for (;;) {
        const int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (hops) ++*hops;
        if (p == x) return x;
        const int gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // halving: any ancestor is valid
        x = p;
    }
}
__device__ __forceinline__ void f_uf_union(int* parent, int a, int b, unsigned* hops = nullptr) {
    a = f_uf_find(parent, a, hops);
    b = f_uf_find(parent, b, hops);
    for (;;) {
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }                                   // a > b: the larger root goes under the smaller (roots = smallest node of the cluster)
        // (look before the CAS: a failing CAS is a same-address atomic, ~11 ns each in a row; atomic loads are not)
        if (__hip_atomic_load(&parent[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a && atomicCAS(&parent[a], a, b) == a) return;
        if (hops) *hops += 1000u;           // (a failed CAS)
        a = f_uf_find(parent, a, hops);
        b = f_uf_find(parent, b, hops);
    }
}
// Ordinary loads (may be served by this CU's L1, i.e. be STALE): a stale parent is an older ancestor, so the result is
// an ancestor of x -- equal results for two nodes still prove that they are connected, unequal ones only cost a real
// union.  (Thousands of walkers ending their walks on one hot root word through the L2 is the bottleneck otherwise.)
__device__ __forceinline__ int f_uf_find_cached(const int* parent, int x) {
    for (int hop = 0; hop < 64; ++hop) {
        const int p = parent[x];
        if (p == x) return x;
        x = p;
    }
    return x;
}
__device__ __forceinline__ int f_uf_root_ro(const int* parent, int x) {      // read-only walk (no unions in flight)
    for (;;) {
        const int p = parent[x];
        if (p == x) return x;
        x = p;
    }
}
// what a slot is: its component, member, point; `own` = a core point of an anchor member (stands in the member's node)
struct FSlot {
    int ci, mi;
    bool is_node, own;
    double p[3];
};
struct __attribute__((aligned(32))) FSlotRec {   // what k_f_pre found out about a slot: one 32-byte load for every later kernel
    double p[3];
    int mi;
    unsigned short ci;
    unsigned char is_node, own;
};
__device__ __forceinline__ FSlot f_slot(const FStep& st, unsigned t) {
    const FSlotRec r = st.slotrec[t];
    FSlot s;
    s.ci = (int)r.ci;
    s.mi = r.mi;
    s.is_node = r.is_node != 0;
    s.own = r.own != 0;
    s.p[0] = r.p[0];
    s.p[1] = r.p[1];
    s.p[2] = r.p[2];
    return s;
}
// first pass: the slot's component and member by binary search over the t0 tables (in LDS when they fit: lds_c / lds_m)
#define F_LDS_COMPS 256
#define F_LDS_MEMS 1024
__device__ __forceinline__ FSlot f_slot_search(const FStep& st, unsigned t, const unsigned* lds_c, const unsigned* lds_m) {
    FSlot s;
    {
        int lo = 0, hi = st.ncomp - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if ((lds_c ? lds_c[mid] : st.comps[mid].t0) <= t) lo = mid; else hi = mid - 1;
        }
        s.ci = lo;
    }
    const FComp& c = st.comps[s.ci];
    {
        int lo = c.m0 + (c.has_anchor ? 1 : 0), hi = c.m0 + c.nm - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if ((lds_m ? lds_m[mid] : st.mems[mid].t0) <= t) lo = mid; else hi = mid - 1;
        }
        s.mi = lo;
    }
    const FMem& m = st.mems[s.mi];
    s.is_node = t == m.t0;
    s.own = false;
    s.p[0] = s.p[1] = s.p[2] = 0.0;
    if (!s.is_node) {
        const long long pi = m.off + (long long)(t - m.t0 - 1u);
        s.p[0] = st.pool[(size_t)pi * 3];
        s.p[1] = st.pool[(size_t)pi * 3 + 1];
        s.p[2] = st.pool[(size_t)pi * 3 + 2];
        s.own = m.am && st.poolcore[pi];
    }
    FSlotRec r;
    r.p[0] = s.p[0];
    r.p[1] = s.p[1];
    r.p[2] = s.p[2];
    r.mi = s.mi;
    r.ci = (unsigned short)s.ci;
    r.is_node = s.is_node ? 1 : 0;
    r.own = s.own ? 1 : 0;
    st.slotrec[t] = r;
    return s;
}
// node of a record of member `m` (its anchor node for the own core points of an anchor member)
__device__ __forceinline__ int f_rec_node(const FStep& st, const FMem& m, const FRec& rc) {
    return st.ncomp + (int)((m.am && (rc.flags & F_CORE)) ? m.t0 : m.t0 + 1u + rc.lidx);
}

// development aid (HMSG_DEBUG_TIMING): 100 MHz ticks a wave spends on an item -> [0] sum, [1] max, [2] items
struct FDbgTimer {
    unsigned* c;
    unsigned long long t0;
    __device__ FDbgTimer(int on, unsigned* counters) : c(on ? counters : nullptr), t0(on ? wall_clock64() : 0ull) {}
    __device__ ~FDbgTimer() {
        if (c && (threadIdx.x & 63) == 0) {
            const unsigned dt = (unsigned)(wall_clock64() - t0);
            atomicAdd(&c[0], dt);
            atomicMax(&c[1], dt);
            atomicAdd(&c[2], 1u);
        }
    }
};

// The step kernels come in pairs: a THREAD per slot settles what a single lane can (most active points re-observe an
// anchored surface: a look at one cell descriptor decides them; a point of an anchor member that is farther than eps
// from every other member keeps its status) and lists the slots that need a walk over a neighbourhood; a WAVE per
// listed slot then does the walks, spread over the whole device.
#define FB 256
#define FWB 1024     /* threads of a walk kernel's workgroup: the dispatcher starts ~20 workgroups per microsecond, so few fat ones */
__device__ __forceinline__ void f_list_push(unsigned* counter, unsigned* list, bool want, unsigned t) {   // wave-uniform call
    const unsigned q = f_wave_slot(want, counter);
    if (want) list[q] = t;
}
// cluster bookkeeping, one set of atomics per (wave, cluster): lanes with `on` add their point to cluster r (size,
// first member).  true on the lane that saw the cluster's size leave zero: it registers the root.
__device__ __forceinline__ bool f_account(const FStep& st, bool on, int r, unsigned t) {
    const int lane = threadIdx.x & 63;
    bool reg = false;
    unsigned long long todo = __ballot(on);
    while (todo) {
        const int l = __ffsll(todo) - 1;
        const int key = __shfl(r, l);
        const bool mine = on && r == key;
        const unsigned long long same = __ballot(mine);
        unsigned f = mine ? t + 1u : 0xffffffffu;
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned u = __shfl_xor(f, o);
            f = u < f ? u : f;
        }
        if (lane == l) {
            atomicMin(&st.first[key], f);
            reg = atomicAdd(&st.size[key], (unsigned)__popcll(same)) == 0u;
        }
        todo &= ~same;
    }
    return reg;
}
__device__ __forceinline__ void f_register_root(const FIndexDev& ix, const FStep& st, int ci, int r) {
    atomicAdd(&st.res[ci].ncl, 1u);
    st.roots[atomicAdd(&ix.counters[FC_ROOTS], 1u)] = (unsigned)r;
}
// number of core records of cloud `id` in the cell of p (0: none / no such cell)
__device__ __forceinline__ unsigned f_own_cell_ncore(const FIndexDev& ix, unsigned id, int cx, int cy, int cz) {
    const unsigned b = f_find(ix, f_key(id, cx >> 2, cy >> 2, cz >> 2));
    return b == F_NONE ? 0u : ix.bricks[b].c[f_local(cx, cy, cz)].ncore;
}
// is p within `d` of the box of some member of component c other than member `jme`?
__device__ __forceinline__ bool f_near_other(const FStep& st, const FComp& c, int jme, const double* p, double d) {
    for (int j = 0; j < c.nm; ++j) {
        if (j == jme) continue;
        const FMem& m = st.mems[c.m0 + j];
        if (p[0] >= m.mn[0] - d && p[0] <= m.mx[0] + d && p[1] >= m.mn[1] - d && p[1] <= m.mx[1] + d && p[2] >= m.mn[2] - d &&
            p[2] <= m.mx[2] + d)
            return true;
    }
    return false;
}
// a point of an anchor member became core in this step: its record does not say so, its cell does
__device__ __forceinline__ void f_mark_promoted(const FIndexDev& ix, const FStep& st, const FSlot& sl) {
    int cx, cy, cz;
    f_cell_of(ix, sl.p[0], sl.p[1], sl.p[2], cx, cy, cz);
    const unsigned b = f_find(ix, f_key(st.mems[sl.mi].id, cx >> 2, cy >> 2, cz >> 2));
    if (b == F_NONE) return;
    const unsigned long long bit = 1ull << f_local(cx, cy, cz);
    if (ix.bricks[b].pm & bit) return;
    const unsigned long long old = atomicOr(&ix.bricks[b].pm, bit);
    if (old == 0ull) {
        const unsigned q = atomicAdd(&ix.counters[FC_DIRTY], 1u);
        if (q < st.dirty_cap) st.dirty[q] = b;
        else atomicOr(&ix.counters[FC_ERR], (unsigned)FERR_TOUCHED);
    }
}

// Connections of the active core points.  Every edge of the eps-graph on core points has to be found from ONE of
// its ends; the own core points of an anchor (member) are one node, so one witness per anchor (member) is enough:
//   (i)   an edge to the first anchor's cluster: looked for by every point that is not in that cluster yet;
//   (ii)  an edge to the node of an anchor member: looked for by every point whose cluster does not hold that node yet;
//   (iii) an edge between two points that are not own core points of an anchor member (frame mask points, points
//         promoted in this step -- not flagged in their records): looked for by BOTH ends, except that a point of
//         the first anchor's cluster need not look (the other end does, or is in that cluster as well).
// An own core point of an anchor member never looks for (iii): the other end does.
// f_link_pre: what one lane can do before the walks -- a point in one cell with a core point of an anchor (member) is
// connected to it without a distance test; an own core point farther than eps from every other member has no edge
// to find (true: settled).
__device__ __forceinline__ bool f_link_pre(const FIndexDev& ix, const FStep& st, const FSlot& sl, unsigned t) {
    const FComp& c = st.comps[sl.ci];
    const int me = st.ncomp + (int)(sl.own ? st.mems[sl.mi].t0 : t);
    int cx, cy, cz;
    f_cell_of(ix, sl.p[0], sl.p[1], sl.p[2], cx, cy, cz);
    if (sl.own && !f_near_other(st, c, sl.mi - c.m0, sl.p, st.eps + 1e-6)) return true;
    if (c.has_anchor && f_uf_find_cached(st.parent, me) != sl.ci && f_own_cell_ncore(ix, st.mem_ids[c.m0], cx, cy, cz)) f_uf_union(st.parent, me, sl.ci);
    for (int j = c.has_anchor ? 1 : 0; j < c.nm; ++j) {
        const FMem& m = st.mems[c.m0 + j];
        if (!m.am || c.m0 + j == sl.mi) continue;
        const double d = 1e-6;
        if (sl.p[0] < m.mn[0] - d || sl.p[0] > m.mx[0] + d || sl.p[1] < m.mn[1] - d || sl.p[1] > m.mx[1] + d || sl.p[2] < m.mn[2] - d ||
            sl.p[2] > m.mx[2] + d)
            continue;
        if (f_own_cell_ncore(ix, m.id, cx, cy, cz) && f_uf_find_cached(st.parent, me) != f_uf_find_cached(st.parent, st.ncomp + (int)m.t0))
            f_uf_union(st.parent, me, st.ncomp + (int)m.t0);
    }
    return false;
}
// f_link_pre2: once those unions are all done -- which of (i), (ii), (iii) is still open for the slot (0: settled).
enum { FL_ANCHOR = 1, FL_MEMBERS = 2, FL_ALL = 4 };
__device__ __forceinline__ unsigned f_link_pre2(const FIndexDev& ix, const FStep& st, const FSlot& sl, unsigned t) {
    const FComp& c = st.comps[sl.ci];
    const int me = st.ncomp + (int)(sl.own ? st.mems[sl.mi].t0 : t);
    const int r = f_uf_root_ro(st.parent, me);                   // (no union in flight in this launch)
    const bool in_anchor = c.has_anchor && r == sl.ci;
    unsigned need = 0u;
    if (!sl.own && !in_anchor) need |= FL_ALL;
    int cx, cy, cz;
    f_cell_of(ix, sl.p[0], sl.p[1], sl.p[2], cx, cy, cz);
    const int lo[3] = {cx - 2, cy - 2, cz - 2}, hi[3] = {cx + 2, cy + 2, cz + 2};
    const double d = st.eps + 1e-6;
    for (int j = 0; j < c.nm; ++j) {
        const FMem& m = st.mems[c.m0 + j];
        if (c.m0 + j == sl.mi) continue;
        const bool first_anchor = c.has_anchor && j == 0;
        if (!first_anchor && !m.am) continue;
        if (sl.p[0] < m.mn[0] - d || sl.p[0] > m.mx[0] + d || sl.p[1] < m.mn[1] - d || sl.p[1] > m.mx[1] + d || sl.p[2] < m.mn[2] - d ||
            sl.p[2] > m.mx[2] + d)
            continue;
        if (first_anchor ? in_anchor : f_uf_root_ro(st.parent, st.ncomp + (int)m.t0) == r) continue;
        // a cell of it with core records in the window may hold the witness
        for (int q = 0; q < 8; ++q) {
            const int bx = (lo[0] >> 2) + ((q >> 2) & 1), by = (lo[1] >> 2) + ((q >> 1) & 1), bz = (lo[2] >> 2) + (q & 1);
            const unsigned b = f_find(ix, f_key(m.id, bx, by, bz));
            if (b != F_NONE && (ix.bricks[b].cm & f_window_mask(bx, by, bz, lo, hi))) {
                need |= first_anchor ? FL_ANCHOR : FL_MEMBERS;
                break;
            }
        }
    }
    return need;
}

// (1) per slot: core flags a lane can decide -- own core points of an anchor member stay core; a point of an anchor
//     member farther than eps from every other member keeps its count; a cell that holds min_points points of the
//     component makes its points core without a distance test -- the rest is listed for k_f_count; points with a
//     cell of the first anchor that holds non-core records in reach are listed for k_f_touch.  Initialises the nodes.
__global__ void __launch_bounds__(FB) k_f_pre(FIndexDev ix, FStep st, int nmem) {
    __shared__ unsigned s_ct0[F_LDS_COMPS], s_mt0[F_LDS_MEMS];
    const bool lds = st.ncomp <= F_LDS_COMPS && nmem <= F_LDS_MEMS;
    if (lds) {
        for (int i = threadIdx.x; i < st.ncomp; i += FB) s_ct0[i] = st.comps[i].t0;
        for (int i = threadIdx.x; i < nmem; i += FB) s_mt0[i] = st.mems[i].t0;
    }
    __syncthreads();
    const unsigned t = blockIdx.x * FB + threadIdx.x;
    if (t < (unsigned)st.ncomp) {
        const FComp c = st.comps[t];
        st.parent[t] = (int)t;
        st.size[t] = c.has_anchor ? c.anchor_n : 0u;
        st.first[t] = c.has_anchor ? 0u : 0xffffffffu;
        st.best[t] = 0ull;
        FRes r;
        r.n_kept = 0u;
        r.ncl = c.has_anchor ? 1u : 0u;
        r.contested = r.pad = 0u;
        for (int a = 0; a < 6; ++a) r.box[a] = a < 3 ? ~0ull : 0ull;
        st.res[t] = r;
        if (c.has_anchor) st.roots[atomicAdd(&ix.counters[FC_ROOTS], 1u)] = t;
    }
    bool want_count = false, want_touch = false, want_link = false;
    if (t < st.T) {
        st.parent[st.ncomp + t] = st.ncomp + (int)t;         // (before any union can name the node)
        st.size[st.ncomp + t] = 0u;
        st.first[st.ncomp + t] = 0xffffffffu;
        st.lab[t] = -1;
    }
    // (connections start in k_f_count: every node has to be initialised before the first union)
    if (t < st.T) {
        const FSlot sl = f_slot_search(st, t, lds ? s_ct0 : nullptr, lds ? s_mt0 : nullptr);
        bool core = sl.own;
        if (!sl.is_node && !sl.own) {
            const FComp& c = st.comps[sl.ci];
            const FMem& m = st.mems[sl.mi];
            if (!m.am || f_near_other(st, c, sl.mi - c.m0, sl.p, st.eps + 1e-6)) {
                int cx, cy, cz;
                f_cell_of(ix, sl.p[0], sl.p[1], sl.p[2], cx, cy, cz);
                unsigned have = 0u;
                for (int j = 0; j < c.nm && have < (unsigned)st.minpts; ++j) {
                    const unsigned b = f_find(ix, f_key(st.mem_ids[c.m0 + j], cx >> 2, cy >> 2, cz >> 2));
                    if (b != F_NONE) have += ix.bricks[b].c[f_local(cx, cy, cz)].cnt;
                }
                core = have >= (unsigned)st.minpts;
                want_count = !core;
            }
        }
        st.acore[t] = core ? 1 : 0;
        want_link = core;
        if (!sl.is_node && st.comps[sl.ci].has_anchor) {
            const FComp& c = st.comps[sl.ci];
            int cx, cy, cz;
            f_cell_of(ix, sl.p[0], sl.p[1], sl.p[2], cx, cy, cz);
            const int lo[3] = {cx - 2, cy - 2, cz - 2}, hi[3] = {cx + 2, cy + 2, cz + 2};
            const unsigned aid = st.mem_ids[c.m0];
            for (int q = 0; q < 8 && !want_touch; ++q) {
                const int bx = (lo[0] >> 2) + ((q >> 2) & 1), by = (lo[1] >> 2) + ((q >> 1) & 1), bz = (lo[2] >> 2) + (q & 1);
                const unsigned b = f_find(ix, f_key(aid, bx, by, bz));
                if (b != F_NONE && (ix.bricks[b].ncm & f_window_mask(bx, by, bz, lo, hi))) want_touch = true;
            }
        }
    }
    f_list_push(&ix.counters[FC_L_COUNT], st.list_count, want_count, t);
    f_list_push(&ix.counters[FC_L_TOUCH], st.list_touch, want_touch, t);
    f_list_push(&ix.counters[FC_L_LINK0], st.list_link0, want_link, t);
}

// (2) the first anchor's non-core points that have an active point within eps: the only points of it whose core
//     status can change
__global__ void __launch_bounds__(FWB) k_f_touch(FIndexDev ix, FStep st) {
    const unsigned n = ix.counters[FC_L_TOUCH];
    const unsigned nw = (gridDim.x * blockDim.x) >> 6;
    for (unsigned i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < n; i += nw) {
        const unsigned t = st.list_touch[i];
        const FSlot sl = f_slot(st, t);
        const int ci = sl.ci;
        const FComp c = st.comps[ci];
        const double* p = sl.p;
        int cx, cy, cz;
        f_cell_of(ix, p[0], p[1], p[2], cx, cy, cz);
        const int lo[3] = {cx - 2, cy - 2, cz - 2}, hi[3] = {cx + 2, cy + 2, cz + 2};
        f_walk(ix, st.mem_ids + c.m0, 1, lo, hi, [&](int) { return (int)FSEL_NONCORE; },
               [&](bool have, int, int ccx, int ccy, int ccz, const FCell& d, unsigned) -> unsigned {
                   if (!have || d.ncore >= d.cnt) return 0u;
                   double dmin2, dmax2;
                   f_cube_dist(ix, p, ccx, ccy, ccz, dmin2, dmax2);
                   if (dmin2 >= st.eps2 * (1.0 + 1e-9) + 1e-12) return 0u;
                   return (d.cnt >= 8u && !f_sub_reach(ix, p, ccx, ccy, ccz, d.sub, st.eps2 * (1.0 + 1e-9) + 1e-12)) ? 0u : d.cnt;
               },
               [&](bool valid, int, int, const FRec& rc, unsigned ri) {
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
                   tr.cellref = b * 64u + f_local(rx, ry, rz);
                   tr.comp = (unsigned)ci;
                   tr.pad = 0u;
                   st.touched[q] = tr;
               },
               [&]() { return false; });
    }
}

// neighbours of p within eps over all members of a component (p's own record included), counted until `minpts`
__device__ __forceinline__ bool f_is_core(const FIndexDev& ix, const FStep& st, const FComp& c, const double* p, FWalkStat* ws = nullptr) {
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
            if (b != F_NONE) n = ix.bricks[b].c[f_local(cx, cy, cz)].cnt;
        }
        have += wave_sum_i32((int)n);
    }
    if (have >= st.minpts) return true;
    const int lo[3] = {cx - 2, cy - 2, cz - 2}, hi[3] = {cx + 2, cy + 2, cz + 2};
    f_walk(ix, ids, c.nm, lo, hi, [&](int) { return (int)FSEL_ALL; },
           [&](bool hv, int, int ccx, int ccy, int ccz, const FCell& d, unsigned) -> unsigned {
               unsigned scan = 0u, sure = 0u;
               if (hv && !(ccx == cx && ccy == cy && ccz == cz)) {
                   double dmin2, dmax2;
                   f_cube_dist(ix, p, ccx, ccy, ccz, dmin2, dmax2);
                   if (dmax2 < st.eps2 * (1.0 - 1e-9) - 1e-12) sure = d.cnt;             // the whole cell is in reach
                   else if (!(dmin2 >= st.eps2 * (1.0 + 1e-9) + 1e-12) &&
                            !(d.cnt >= 8u && !f_sub_reach(ix, p, ccx, ccy, ccz, d.sub, st.eps2 * (1.0 + 1e-9) + 1e-12)))
                       scan = d.cnt;
               }
               have += wave_sum_i32((int)sure);
               return scan;
           },
           [&](bool valid, int, int, const FRec& rc, unsigned) {
               const bool h = valid && f_dist2(rc.x, rc.y, rc.z, p[0], p[1], p[2]) < st.eps2;
               have += __popcll(__ballot(h));
           },
           [&]() { return have >= st.minpts; }, 0u, ws);
    return have >= st.minpts;
}

// (3) neighbour counts of the listed points; re-count and promotion of the touched points of the first anchors; and
//     (thread per entry) the lane-level part of the connections of the points k_f_pre found core
__global__ void __launch_bounds__(256) k_f_linkpre1(FIndexDev ix, FStep st) {
    const int lane = threadIdx.x & 63;
    // entries: the slots k_f_pre found core, then the slots k_f_count counted -- those of them it found core (acore).
    // (k_f_count used to append its core slots to the first list: one atomicAdd per WAVE on one counter, ~8 000 waves a
    //  step, and same-address atomics retire one per ~11 ns on this GPU -- scripts/microbench/atom_bench.hip: that append
    //  alone was the kernel's 88 us.)
    const unsigned n0 = ix.counters[FC_L_LINK0], n = n0 + ix.counters[FC_L_COUNT];
    for (unsigned i0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += gridDim.x * blockDim.x) {
        const unsigned i = i0 + (unsigned)lane;
        bool hard = false;
        unsigned t = 0u;
        bool have = i < n0;
        if (have) t = st.list_link0[i];
        else if (i < n) {
            t = st.list_count[i - n0];
            have = st.acore[t] != 0;
        }
        if (have) {
            const FSlot sl = f_slot(st, t);
            if (!sl.own && st.mems[sl.mi].am) f_mark_promoted(ix, st, sl);            // a point of an anchor member promoted in this step
            hard = !f_link_pre(ix, st, sl, t);
        }
        f_list_push(&ix.counters[FC_L_LINK], st.list_link, hard, t);
    }
}
__global__ void __launch_bounds__(FWB) k_f_count(FIndexDev ix, FStep st, unsigned link_blocks) {
    const int lane = threadIdx.x & 63;
    const unsigned n_count = ix.counters[FC_L_COUNT];
    const unsigned n_touched = min(ix.counters[FC_TOUCHED_RECS], st.touched_cap);
    const unsigned nw = ((gridDim.x - link_blocks) * blockDim.x) >> 6;
    for (unsigned w = ((blockIdx.x - link_blocks) * blockDim.x + threadIdx.x) >> 6; w < n_count + n_touched; w += nw) {
        if (w < n_count) {
            const unsigned t = st.list_count[w];
            const FSlot sl = f_slot(st, t);
            FWalkStat wst = {0u, 0u, 0u, 0u, 0u};
            const bool dbg = F_WALK_STATS && st.dbgbuf != nullptr && st.debug == 2;
            const unsigned long long dbg_t0 = dbg ? wall_clock64() : 0ull;
            const bool core = f_is_core(ix, st, st.comps[sl.ci], sl.p, dbg ? &wst : nullptr);
            if (dbg && lane == 0) {
                unsigned long long* o = st.dbgbuf + (size_t)w * 2;
                o[0] = ((wall_clock64() - dbg_t0) << 32) | ((unsigned long long)min(wst.recs, 0xfffffu) << 12) | ((unsigned long long)min(wst.trips, 0xfffu));
                o[1] = ((unsigned long long)min(wst.cells, 0xffffu) << 48) | ((unsigned long long)min(wst.rounds, 0xffu) << 40) | ((unsigned long long)min(wst.groups, 0xffu) << 32) |
                       ((unsigned long long)min(st.comps[sl.ci].nm, 255) << 24) | ((unsigned long long)(st.comps[sl.ci].has_anchor ? 1 : 0) << 16) | ((unsigned long long)(core ? 1 : 0) << 8) |
                       (unsigned long long)(st.mems[sl.mi].am ? 1 : 0);
            }
            // (its connections start in k_f_linkpre1, a thread per core point: not on one lane of this wave)
            if (core && lane == 0) st.acore[t] = 1;              // (k_f_linkpre1 picks the slot up from list_count + this flag)
        } else {
            const FTouched tr = st.touched[w - n_count];
            const FComp c = st.comps[tr.comp];
            const FRec rc = ix.recs[tr.rec];
            const double p[3] = {rc.x, rc.y, rc.z};
            const bool core = f_is_core(ix, st, c, p);
            if (core && lane == 0) {                             // promoted: a border point of the anchor's cluster becomes core
                atomicOr(&ix.recs[tr.rec].flags, F_CORE);
                FBrick& br = ix.bricks[tr.cellref >> 6];
                FCell& cell = br.c[tr.cellref & 63u];
                const unsigned long long bit = 1ull << (tr.cellref & 63u);
                const unsigned nc = atomicAdd(&cell.ncore, 1u) + 1u;
                if (!(br.cm & bit)) atomicOr(&br.cm, bit);
                if (nc >= cell.cnt) atomicAnd(&br.ncm, ~bit);
                st.poolcore[st.mems[c.m0].off + rc.lidx] = 1;
            }
        }
    }
}

// one action per (trip, cell) with a witness: `hit` lanes of the same cell elect their lowest lane
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

// (3b) the listed core points once more, now that every lane-level union is done
__global__ void __launch_bounds__(256) k_f_linkpre2(FIndexDev ix, FStep st) {
    const int lane = threadIdx.x & 63;
    const unsigned n = ix.counters[FC_L_LINK];
    for (unsigned i0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += gridDim.x * blockDim.x) {
        const unsigned i = i0 + (unsigned)lane;
        bool hard = false;
        unsigned t = 0u;
        if (i < n) {
            t = st.list_link[i];
            const unsigned need = f_link_pre2(ix, st, f_slot(st, t), t);
            st.pos[t] = need;                                    // (scratch until the scan of the keep flags)
            hard = need != 0u;
        }
        f_list_push(&ix.counters[FC_L_LINK2], st.list_link2, hard, t);
    }
}

// (4) the walks of the listed core points: whatever of (i), (ii), (iii) k_f_linkpre2 left open
__global__ void __launch_bounds__(FWB) k_f_link(FIndexDev ix, FStep st) {
    const int lane = threadIdx.x & 63;
    const unsigned n = ix.counters[FC_L_LINK2];
    const unsigned nw = (gridDim.x * blockDim.x) >> 6;
    for (unsigned i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < n; i += nw) {
        const unsigned t = st.list_link2[i];
        const FSlot sl = f_slot(st, t);
        const int ci = sl.ci;
        const FComp c = st.comps[ci];
        const FMem mm = st.mems[sl.mi];
        const int jme = sl.mi - c.m0;
        if (st.debug && lane == 0 && (t & 15u) == 0u) atomicAdd(&ix.counters[FC_DBG + (c.has_anchor ? 0 : 4) + (sl.own ? 0 : (mm.am ? 1 : 2))], 16u);
        const double* p = sl.p;
        int cx, cy, cz;
        f_cell_of(ix, p[0], p[1], p[2], cx, cy, cz);
        const int lo[3] = {cx - 2, cy - 2, cz - 2}, hi[3] = {cx + 2, cy + 2, cz + 2};
        const int me = st.ncomp + (int)(sl.own ? mm.t0 : t);
        const unsigned need = st.pos[t];
        FWalkStat wst = {0u, 0u, 0u, 0u, 0u};
        const unsigned long long dbg_t0 = (F_WALK_STATS && st.dbgbuf) ? wall_clock64() : 0ull;
        if (need & FL_ANCHOR) {                                  // (i): the first core point of the anchor within eps
            bool in_anchor = false;
            f_walk(ix, st.mem_ids + c.m0, 1, lo, hi, [&](int) { return (int)FSEL_CORE; },
                   [&](bool hv, int, int ccx, int ccy, int ccz, const FCell& d, unsigned) -> unsigned {
                       if (!hv || d.ncore == 0u) return 0u;
                       double dmin2, dmax2;
                       f_cube_dist(ix, p, ccx, ccy, ccz, dmin2, dmax2);
                       if (dmin2 >= st.eps2 * (1.0 + 1e-9) + 1e-12) return 0u;
                   return (d.cnt >= 8u && !f_sub_reach(ix, p, ccx, ccy, ccz, d.sub, st.eps2 * (1.0 + 1e-9) + 1e-12)) ? 0u : d.cnt;
                   },
                   [&](bool valid, int, int, const FRec& rc, unsigned) {
                       const bool hit = valid && (rc.flags & F_CORE) && f_dist2(rc.x, rc.y, rc.z, p[0], p[1], p[2]) < st.eps2;
                       if (__any(hit)) in_anchor = true;
                   },
                   [&]() { return in_anchor; }, 128u, (F_WALK_STATS && st.dbgbuf) ? &wst : nullptr);
            if (in_anchor) {
                if (lane == 0) f_uf_union(st.parent, me, ci);
                if (!sl.own) continue;                           // (iii) is the other ends' now; (ii) stays open for an anchor member's node
            }
        }
        if (!(need & (FL_MEMBERS | FL_ALL))) continue;
        // (ii) + (iii).  all: every core point within eps that is not an own core point of an anchor member, one
        // witness per anchor member; else only the witnesses.
        const bool all = (need & FL_ALL) != 0u;
        const int j0 = c.has_anchor ? 1 : 0;
        unsigned long long done_am = 0ull;                       // anchor members (first 64 of the walk) with a witness
        if (sl.own && jme - j0 < 64) done_am |= 1ull << (jme - j0);   // (its own member's node is the walker itself)
        {   // ... or whose node is in the walker's cluster already (a lane per member)
            const int myroot = f_uf_find_cached(st.parent, me);
            bool same = false;
            if (lane < c.nm - j0 && st.mems[c.m0 + j0 + lane].am) same = f_uf_find_cached(st.parent, st.ncomp + (int)st.mems[c.m0 + j0 + lane].t0) == myroot;
            done_am |= __ballot(same);
        }
        f_walk(ix, st.mem_ids + c.m0 + j0, c.nm - j0, lo, hi,
               [&](int jj) {
                   const bool am = st.mems[c.m0 + j0 + jj].am != 0u;
                   if (all) return (sl.own && jj + j0 == jme) ? (int)FSEL_SKIP : (am ? (int)FSEL_CORE_PROM : (int)FSEL_ALL);
                   return (am && jj + j0 != jme) ? (int)FSEL_CORE : (int)FSEL_SKIP;
               },
               [&](bool hv, int jj, int ccx, int ccy, int ccz, const FCell& d, unsigned cr) -> unsigned {
                   if (!hv) return 0u;
                   // an anchor member with a witness: only its cells with a point promoted in this step are still of interest
                   if (jj < 64 && (done_am >> jj & 1ull) && (!all || !(ix.bricks[cr >> 6].pm >> (cr & 63u) & 1ull))) return 0u;
                   double dmin2, dmax2;
                   f_cube_dist(ix, p, ccx, ccy, ccz, dmin2, dmax2);
                   if (dmin2 >= st.eps2 * (1.0 + 1e-9) + 1e-12) return 0u;
                   return (d.cnt >= 8u && !f_sub_reach(ix, p, ccx, ccy, ccz, d.sub, st.eps2 * (1.0 + 1e-9) + 1e-12)) ? 0u : d.cnt;
               },
               [&](bool valid, int jj, int cl, const FRec& rc, unsigned) {
                   bool hit = false, am_hit = false;
                   int node = ci;
                   if (valid && f_dist2(rc.x, rc.y, rc.z, p[0], p[1], p[2]) < st.eps2) {
                       const FMem& m = st.mems[c.m0 + j0 + jj];
                       node = f_rec_node(st, m, rc);
                       const unsigned q = m.t0 + 1u + rc.lidx;
                       am_hit = m.am && (rc.flags & F_CORE);
                       hit = q != t && (st.acore[q] != 0) && (all || am_hit);
                       am_hit = am_hit && hit;
                       if (am_hit && jj < 64 && (done_am >> jj & 1ull)) hit = false;      // (that node is connected already)
                       if (am_hit && jj >= 64 && sl.own && jj + j0 == jme) hit = false;
                   }
                   // the hit lanes look their witnesses' roots up side by side; one union per distinct root
                   // The hit lanes look their witnesses' roots up side by side; every root that is not the smallest of
                   // them (and of the walker's own) is hooked under that one directly, each lane its own word -- a star,
                   // not a chain of unions.  A root that moved in the meantime goes through the general union.
                   const int wr = hit ? f_uf_find_cached(st.parent, node) : 0x7fffffff;
                   const int mr = f_uf_find_cached(st.parent, me);
                   int m = min(wr, mr);
                   for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
                   // (every CAS below is preceded by an atomic LOAD of the word: walkers of one cluster all try the same witness
                   //  roots, and a failing CAS is a same-address atomic -- ~11 ns each, one after the other)
                   if (hit && wr != m) {
                       int old = __hip_atomic_load(&st.parent[wr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                       if (old == wr) old = atomicCAS(&st.parent[wr], wr, m);
                       if (old != wr && old != m) f_uf_union(st.parent, wr, m);
                   }
                   const bool any_hit = __any(hit) != 0;
                   if (lane == 0 && mr != m && any_hit) {
                       int old = __hip_atomic_load(&st.parent[mr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                       if (old == mr) old = atomicCAS(&st.parent[mr], mr, m);
                       if (old != mr && old != m) f_uf_union(st.parent, mr, m);
                   }
                   unsigned long long am_todo = __ballot(am_hit && jj < 64);
                   while (am_todo) {
                       const int l = __ffsll(am_todo) - 1;
                       am_todo &= am_todo - 1ull;
                       done_am |= 1ull << __shfl(jj, l);
                   }
               },
               [&]() { return false; }, 128u, (F_WALK_STATS && st.dbgbuf) ? &wst : nullptr);
        if (F_WALK_STATS && st.dbgbuf && st.debug != 2 && lane == 0) {
            unsigned long long* o = st.dbgbuf + (size_t)i * 2;
            o[0] = ((wall_clock64() - dbg_t0) << 32) | ((unsigned long long)min(wst.recs, 0xfffffu) << 12) | ((unsigned long long)min(wst.trips, 0xfffu));
            o[1] = ((unsigned long long)min(wst.cells, 0xffffu) << 48) | ((unsigned long long)min(wst.rounds, 0xffu) << 40) | ((unsigned long long)min(wst.groups, 0xffu) << 32) |
                   ((unsigned long long)min(c.nm, 255) << 24) | ((unsigned long long)need << 16) | ((unsigned long long)(sl.own ? 1 : 0) << 8) | (unsigned long long)(mm.am ? 1 : 0);
        }
    }
}

// (5) clusters of the core points: root, size, first member; the first member registers the root
__global__ void __launch_bounds__(FB) k_f_acct(FIndexDev ix, FStep st) {
    const unsigned t = blockIdx.x * FB + threadIdx.x;
    bool on = false;
    int r = -1, ci = 0;
    if (t < st.T && st.acore[t]) {
        const FSlot sl = f_slot(st, t);
        ci = sl.ci;
        r = f_uf_root_ro(st.parent, st.ncomp + (int)(sl.own ? st.mems[sl.mi].t0 : t));
        st.lab[t] = r;
        on = true;
    }
    if (f_account(st, on, r, t) && r >= st.ncomp) f_register_root(ix, st, ci, r);
}

// (6) labels of the non-core points (the reaching cluster with the smallest root = Open3D's first cluster; cores of
//     two clusters in reach = contested).  What a lane can settle: no cluster at all -> noise; a non-core point of an
//     anchor member farther than 2 eps from every other member is a border point of its member's cluster and of no
//     other; in a component with ONE cluster, every border point of an anchor (member) and every point in one cell
//     with a core point of the first anchor belongs to it.  The rest is listed for k_f_label.
__global__ void __launch_bounds__(FB) k_f_labelpre(FIndexDev ix, FStep st) {
    const unsigned t = blockIdx.x * FB + threadIdx.x;
    bool hard = false, on = false;
    int r = -1;
    if (t < st.T && !st.acore[t]) {
        const FSlot sl = f_slot(st, t);
        if (!sl.is_node) {
            const FComp& c = st.comps[sl.ci];
            const FMem& m = st.mems[sl.mi];
            const unsigned ncl = st.res[sl.ci].ncl;
            if (ncl != 0u) {
                if (m.am && (ncl == 1u || !f_near_other(st, c, sl.mi - c.m0, sl.p, 2.0 * st.eps + 1e-6)))
                    r = f_uf_root_ro(st.parent, st.ncomp + (int)m.t0);
                else if (ncl == 1u && c.has_anchor) {
                    int cx, cy, cz;
                    f_cell_of(ix, sl.p[0], sl.p[1], sl.p[2], cx, cy, cz);
                    if (f_own_cell_ncore(ix, st.mem_ids[c.m0], cx, cy, cz)) r = sl.ci;
                }
                on = r >= 0;
                if (on) st.lab[t] = r;
                hard = !on;
            }
        }
    }
    f_account(st, on, r, t);                                     // (the cluster exists already: nothing to register)
    f_list_push(&ix.counters[FC_L_LABEL], st.list_label, hard, t);
}
__device__ __forceinline__ void f_label_walk(const FIndexDev& ix, const FStep& st, int ci, const double* p, unsigned self, int jme, bool skip_own_member,
                                             int best0, bool got_anchor0, bool one_cluster, int& wbest_out, bool& contest_out) {
    const int lane = threadIdx.x & 63;
    const FComp c = st.comps[ci];
    int cx, cy, cz;
    f_cell_of(ix, p[0], p[1], p[2], cx, cy, cz);
    const int lo[3] = {cx - 2, cy - 2, cz - 2}, hi[3] = {cx + 2, cy + 2, cz + 2};
    int best = best0;                                            // per lane: smallest root it saw a witness of
    bool multi = false;                                          // this lane saw witnesses of two different clusters
    bool got_anchor = got_anchor0;                               // the first anchor's cluster is known to reach the point
    bool any_hit = best0 != 0x7fffffff;
    if (c.has_anchor && !got_anchor) {
        // does the first anchor's cluster reach the point?  a core point in the point's own cell is within eps;
        // else the first witness ends the search
        unsigned nc = 0u;
        if (lane == 0) nc = f_own_cell_ncore(ix, st.mem_ids[c.m0], cx, cy, cz);
        got_anchor = __shfl(nc, 0) != 0u;
        if (!got_anchor)
            f_walk(ix, st.mem_ids + c.m0, 1, lo, hi, [&](int) { return (int)FSEL_CORE; },
                   [&](bool hv, int, int ccx, int ccy, int ccz, const FCell& d, unsigned) -> unsigned {
                       if (!hv || d.ncore == 0u) return 0u;
                       double dmin2, dmax2;
                       f_cube_dist(ix, p, ccx, ccy, ccz, dmin2, dmax2);
                       if (dmin2 >= st.eps2 * (1.0 + 1e-9) + 1e-12) return 0u;
                   return (d.cnt >= 8u && !f_sub_reach(ix, p, ccx, ccy, ccz, d.sub, st.eps2 * (1.0 + 1e-9) + 1e-12)) ? 0u : d.cnt;
                   },
                   [&](bool valid, int, int, const FRec& rc, unsigned) {
                       const bool hit = valid && (rc.flags & F_CORE) && f_dist2(rc.x, rc.y, rc.z, p[0], p[1], p[2]) < st.eps2;
                       if (__any(hit)) got_anchor = true;
                   },
                   [&]() { return got_anchor; }, 128u);
        if (got_anchor) {
            if (best != 0x7fffffff && best != ci) multi = true;
            best = ci;                                           // (the smallest root there is)
            any_hit = true;
        }
    }
    if (!(one_cluster && any_hit)) {
        // the other members: every core point within eps counts; the own core points of an anchor member are one
        // cluster (one witness is enough, the rest of its all-core cells is skipped)
        const int j0 = c.has_anchor ? 1 : 0;
        unsigned long long seen_am = 0ull;                       // anchor members (first 64 of the walk) an own core point has been seen of
        f_walk(ix, st.mem_ids + c.m0 + j0, c.nm - j0, lo, hi,
               [&](int jj) {
                   if (skip_own_member && jj + j0 == jme) return (int)FSEL_SKIP;
                   return st.mems[c.m0 + j0 + jj].am ? (int)FSEL_CORE_PROM : (int)FSEL_ALL;
               },
               [&](bool hv, int jj, int ccx, int ccy, int ccz, const FCell& d, unsigned cr) -> unsigned {
                   if (!hv) return 0u;
                   if (jj < 64 && (seen_am >> jj & 1ull) && !(ix.bricks[cr >> 6].pm >> (cr & 63u) & 1ull)) return 0u;
                   double dmin2, dmax2;
                   f_cube_dist(ix, p, ccx, ccy, ccz, dmin2, dmax2);
                   if (dmin2 >= st.eps2 * (1.0 + 1e-9) + 1e-12) return 0u;
                   return (d.cnt >= 8u && !f_sub_reach(ix, p, ccx, ccy, ccz, d.sub, st.eps2 * (1.0 + 1e-9) + 1e-12)) ? 0u : d.cnt;
               },
               [&](bool valid, int jj, int, const FRec& rc, unsigned) {
                   bool hit = false, am_hit = false;
                   int r = -1;
                   if (valid && f_dist2(rc.x, rc.y, rc.z, p[0], p[1], p[2]) < st.eps2) {
                       const FMem& m = st.mems[c.m0 + j0 + jj];
                       const unsigned q = m.t0 + 1u + rc.lidx;
                       if (q != self && st.acore[q]) {
                           am_hit = m.am && (rc.flags & F_CORE);
                           if (!(am_hit && jj < 64 && (seen_am >> jj & 1ull))) {       // (that cluster is accounted for)
                               hit = true;
                               r = f_uf_root_ro(st.parent, f_rec_node(st, m, rc));
                           }
                       }
                   }
                   if (hit) {
                       if (best != 0x7fffffff && r != best) multi = true;
                       if (r < best) best = r;
                   }
                   if (__any(hit)) any_hit = true;
                   unsigned long long am_todo = __ballot(am_hit && jj < 64);
                   while (am_todo) {
                       const int l = __ffsll(am_todo) - 1;
                       am_todo &= am_todo - 1ull;
                       seen_am |= 1ull << __shfl(jj, l);
                   }
               },
               [&]() { return one_cluster && any_hit; }, 128u);
    }
    int wbest = best;
    for (int o = 32; o > 0; o >>= 1) {
        const int u = __shfl_xor(wbest, o);
        wbest = u < wbest ? u : wbest;
    }
    wbest_out = wbest;
    contest_out = __any(multi || (best != 0x7fffffff && best != wbest)) != 0;
}
// (7) labels of the listed points; contest check of the touched points of the first anchors that stayed non-core
__global__ void __launch_bounds__(FWB) k_f_label(FIndexDev ix, FStep st) {
    const int lane = threadIdx.x & 63;
    const unsigned n_label = ix.counters[FC_L_LABEL];
    const unsigned n_touched = min(ix.counters[FC_TOUCHED_RECS], st.touched_cap);
    const unsigned nw = (gridDim.x * blockDim.x) >> 6;
    for (unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n_label + n_touched; w += nw) {
        if (w >= n_label) {
            const FTouched tr = st.touched[w - n_label];
            const int ci = (int)tr.comp;
            const FRec rc = ix.recs[tr.rec];
            if (lane == 0) atomicAnd(&ix.recs[tr.rec].flags, ~F_TOUCHED);
            if ((rc.flags & F_CORE) || st.res[ci].ncl <= 1u) continue;     // promoted by k_f_count / no second cluster
            const double p[3] = {rc.x, rc.y, rc.z};
            int wbest;
            bool contest;
            f_label_walk(ix, st, ci, p, F_NONE, -1, false, ci, true, false, wbest, contest);    // a border point of the first anchor's cluster
            if (contest && lane == 0 && !st.res[ci].contested) st.res[ci].contested = 1u;
            continue;
        }
        const unsigned t = st.list_label[w];
        const FSlot sl = f_slot(st, t);
        const FMem& mm = st.mems[sl.mi];
        int best0 = 0x7fffffff;
        if (mm.am) best0 = f_uf_root_ro(st.parent, st.ncomp + (int)mm.t0);       // a border point of its member's one cluster
        int wbest;
        bool contest;
        f_label_walk(ix, st, sl.ci, sl.p, t, sl.mi - st.comps[sl.ci].m0, mm.am != 0u, best0, false, st.res[sl.ci].ncl == 1u, wbest, contest);
        const bool on = wbest != 0x7fffffff;
        if (lane == 0) {
            st.lab[t] = on ? wbest : -1;
            if (contest && !st.res[sl.ci].contested) st.res[sl.ci].contested = 1u;
        }
        f_account(st, on && lane == 0, wbest, t);
    }
}

// (5) largest cluster per component (Counter.most_common: ties go to the cluster that appears first in point order)
__global__ void k_f_pick(FIndexDev ix, FStep st) {
    const unsigned n = ix.counters[FC_ROOTS];
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned r = st.roots[i];
        const unsigned sz = st.size[r];
        if (!sz) continue;
        const int ci = r < (unsigned)st.ncomp ? (int)r : (int)st.slotrec[r - (unsigned)st.ncomp].ci;
        atomicMax(&st.best[ci], ((unsigned long long)sz << 32) | (unsigned long long)(0xffffffffu - st.first[r]));
    }
}
// (6) graph_utils.py:853-880: keep the largest cluster unless there is none or it has fewer than 5 points
__global__ void k_f_keep(FIndexDev ix, FStep st) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0u) {                                               // (the lists of this step have been consumed)
        ix.counters[FC_STAT] += ix.counters[FC_L_COUNT];
        ix.counters[FC_STAT + 1] += ix.counters[FC_L_TOUCH];
        ix.counters[FC_STAT + 2] += ix.counters[FC_L_LINK2];
        ix.counters[FC_STAT + 3] += ix.counters[FC_L_LABEL];
        ix.counters[FC_STAT + 4] += ix.counters[FC_TOUCHED_RECS];
        ix.counters[FC_ROOTS] = 0u;
        ix.counters[FC_TOUCHED_RECS] = 0u;
        ix.counters[FC_L_COUNT] = ix.counters[FC_L_TOUCH] = ix.counters[FC_L_LINK0] = ix.counters[FC_L_LINK] = ix.counters[FC_L_LINK2] = ix.counters[FC_L_LABEL] = 0u;
    }
    {   // the promoted-point masks of this step
        const unsigned nd = min(ix.counters[FC_DIRTY], st.dirty_cap);
        for (unsigned i = t; i < nd; i += gridDim.x * blockDim.x) ix.bricks[st.dirty[i]].pm = 0ull;
    }
    if (t >= st.T) return;
    const FSlotRec sr = st.slotrec[t];
    const int ci = (int)sr.ci;
    const FComp& c = st.comps[ci];
    if (sr.is_node) {
        st.keep[t] = 0u;
        return;
    }
    const unsigned long long b = st.best[ci];
    bool keep = true;
    if ((unsigned)(b >> 32) >= 5u) {
        const unsigned f = 0xffffffffu - (unsigned)(b & 0xffffffffull);
        const int win = f == 0u ? ci : st.lab[f - 1u];
        keep = st.lab[t] == win;
        if (c.has_anchor && win != ci) atomicOr(&ix.counters[FC_ERR], (unsigned)FERR_WINNER);
    }
    st.keep[t] = keep ? 1u : 0u;
}
// (7) the kept active points go behind the first anchor (or to the component's new cloud), their box and count come
//     back, and their index slots are reserved
__global__ void __launch_bounds__(256) k_f_emit(FIndexDev ix, FStep st, FInsArgs ins) {
    const int lane = threadIdx.x & 63;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0u) ix.counters[FC_DIRTY] = 0u;                    // (k_f_keep has cleared the masks)
    const bool in = t < st.T;
    int ci = -1;
    bool kp = false;
    double v[3] = {0, 0, 0};
    unsigned out_id = 0u;
    if (in) {
        const FSlotRec sr = st.slotrec[t];
        ci = (int)sr.ci;
        const FComp c = st.comps[ci];
        kp = st.keep[t] != 0u;
        const unsigned k = st.pos[t] - st.pos[c.t0];
        if (kp) {
            const long long d = c.out_off + (long long)k;
            for (int a = 0; a < 3; ++a) {
                v[a] = sr.p[a];
                st.pool[(size_t)d * 3 + a] = v[a];
            }
            st.poolcore[d] = st.acore[t];
            st.dst[t] = (unsigned)d;
            st.item_id[t] = c.out_id;
            st.item_lidx[t] = c.out_lidx0 + k;
            out_id = c.out_id;
        }
        if (t == c.t0 + c.nt - 1u) st.res[ci].n_kept = k + (kp ? 1u : 0u);
    }
    f_reserve(ix, ins, kp, t, out_id, v[0], v[1], v[2]);
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
    DevBuf<FHashEnt> ix_tab;
    DevBuf<unsigned> ix_counters;
    DevBuf<FBrick> ix_bricks;
    DevBuf<FRec> ix_recs;
    FIndexDev ix;
    unsigned next_id = 1;
    // step scratch
    DevBuf<char> d_pack;             // [comps | mems | ids]
    PinnedBuf<char> h_pack;
    DevBuf<unsigned char> acore;
    DevBuf<int> parent, lab;
    DevBuf<FSlotRec> slotrec;
    DevBuf<unsigned> dirty;
    DevBuf<unsigned long long> dbgbuf;
    DevBuf<unsigned> size, first, keep, pos, dst, item_id, item_lidx, roots, cellref, slot, touched_cells, lists;
    DevBuf<unsigned long long> best;
    DevBuf<FRes> d_res;
    DevBuf<FTouched> touched;
    PinnedBuf<char> h_res;           // [FRes x ncomp | counters]
    DevBuf<FInsSeg> d_insseg;
    DevBuf<CatSeg> d_reloc;
    PinnedBuf<CatSeg> h_reloc;
    DevBuf<char> d_ovtab;            // overlap step: [clouds | tasks | counts]
    PinnedBuf<char> h_ovtab;
    int n_cu = 0;
    double stat_points_seen = 0;     // (batch fold: points of the DBSCAN batches before the last step)
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
        const size_t rec_cap = std::min<size_t>((size_t)total_points * 20 + ((size_t)1 << 20), 0xfffffff0u);
        ix_tab.alloc(H);
        ix_bricks.alloc(brick_cap);
        ix_recs.alloc(rec_cap);
        ix_counters.alloc(FC_N);
        HIP_TRY(hipMemsetAsync(ix_tab.p, 0xff, H * sizeof(FHashEnt), s));
        HIP_TRY(hipMemsetAsync(ix_bricks.p, 0, brick_cap * sizeof(FBrick), s));
        HIP_TRY(hipMemsetAsync(ix_counters.p, 0, FC_N * 4, s));
        ix.tab = ix_tab.p;
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
        const size_t prof_idx = ops.prof->ev.size();
        {
            ProfScope ps(ops.prof, s, "k_f_overlap", 0.0);
            for (int dir = 0; dir < 2; ++dir) {
                const unsigned nb = dir ? nblk2 : nblk1;
                if (!nb) continue;
                const size_t o = (size_t)dir * P;
                hipLaunchKernelGGL(k_f_overlap, dim3(nb), dim3(256), 0, s, ix, (const double*)pool.p, dg, dt + o, (int)P, r2, r, dc + o,
                                   (dir && decide_th >= 0.0) ? (const unsigned*)dc : (const unsigned*)nullptr, decide_th);
            }
        }
        HMSG_CHECK_LAUNCH();
        pub_counts.launch(s, (const unsigned*)dc, tasks.size());
        pub_counts.wait();
        const unsigned* hc = pub_counts.data();
        double ov_work = 0;
        for (size_t k = 0; k < P; ++k) {
            const int na = std::min(L[pairs[k].first].n, L[pairs[k].second].n), nb = std::max(L[pairs[k].first].n, L[pairs[k].second].n);
            ratio[k] = std::max((double)hc[k] / (double)na, (double)hc[P + k] / (double)nb);
            ov_work += 12.0 * na;
            if (!(decide_th >= 0.0 && (double)hc[k] / (double)na > decide_th)) ov_work += 12.0 * nb;
        }
        if (ops.prof->enabled && prof_idx < ops.prof->ev.size()) ops.prof->ev[prof_idx].work = ov_work;
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
                m.am = (use_anchor && cl.anchor && minpts >= 5) ? 1u : 0u;
                for (int a = 0; a < 3; ++a) {
                    m.mn[a] = cl.mn[a];
                    m.mx[a] = cl.mx[a];
                }
                if (anch && k == 0) m.t0 = F_NONE;
                else {
                    m.t0 = T;                               // the member's node slot, then its points
                    T += (unsigned)cl.n + 1u;
                }
                fm.push_back(m);
                fids.push_back(cl.id);
            }
            q.nm = (int)fm.size() - q.m0;
            q.nt = T - q.t0;
            q.n_active = (unsigned)active;
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
                if ((long long)A.n + q.n_active > A.cap) {
                    const long long ncap = 2 * ((long long)A.n + q.n_active);
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
                q.out_off = pool_alloc(2ll * q.n_active);
            }
        }
        if (!reloc.empty()) {
            // (own staging buffers: no wait -- the previous step's copy has completed, every step ends with one)
            h_reloc.ensure(reloc.size());
            d_reloc.ensure(reloc.size());
            memcpy(h_reloc.p, reloc.data(), reloc.size() * sizeof(CatSeg));
            HIP_TRY(hipMemcpyAsync(d_reloc.p, h_reloc.p, reloc.size() * sizeof(CatSeg), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_concat, dim3(reloc_blocks), dim3(256), 0, s, (const double*)pool.p, (const CatSeg*)d_reloc.p, (int)reloc.size(),
                               pool.p, (const unsigned char*)poolcore.p, poolcore.p, CAT_CHUNK);
            HMSG_CHECK_LAUNCH();
        }
        lap(3);
        // ---- the step kernels
        const int NCOMP = (int)fc.size();
        HMSG_REQUIRE(NCOMP < 65536, HMSG_ERR_UNSUPPORTED, "merge: more than 65535 components in one fold step");
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
            lists.ensure((size_t)T * 6);
            slotrec.ensure(T);
            dirty.ensure(1u << 20);
            st.dirty = dirty.p;
            st.dirty_cap = 1u << 20;
            st.slotrec = slotrec.p;
            st.list_link2 = lists.p + 5 * (size_t)T;
            st.list_count = lists.p;
            st.list_touch = lists.p + T;
            st.list_link0 = lists.p + 2 * (size_t)T;
            st.list_link = lists.p + 3 * (size_t)T;
            st.list_label = lists.p + 4 * (size_t)T;
            st.eps = eps;
            st.debug = getenv("HMSG_DEBUG_TIMING") ? 1 : 0;
            st.dbgbuf = nullptr;
            bool dump_step = getenv("HMSG_DEBUG_LINKSTEP") && (int)fstat[0] == atoi(getenv("HMSG_DEBUG_LINKSTEP"));
            if (getenv("HMSG_DEBUG_COUNTSTEP") && (int)fstat[0] == atoi(getenv("HMSG_DEBUG_COUNTSTEP"))) {
                dump_step = true;
                st.debug = 2;
            }
            if (dump_step) {
                dbgbuf.ensure((size_t)T * 2);
                HIP_TRY(hipMemsetAsync(dbgbuf.p, 0, (size_t)T * 16, s));
                st.dbgbuf = dbgbuf.p;
            }
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
            const unsigned gS = std::max(cdiv(std::max<unsigned>(T, (unsigned)NCOMP), FB), 1u);   // a thread per slot
            const unsigned gWmul = getenv("HMSG_DEBUG_GW") ? (unsigned)atoi(getenv("HMSG_DEBUG_GW")) : 2u;
            const unsigned gW = std::max(1u, std::min(cdiv((size_t)T * 64, FWB), (unsigned)n_cu * gWmul)); // a wave per listed slot (grid-stride)
            const unsigned gL = std::max(1u, std::min(gS, 64u));
            const unsigned gT = cdiv(std::max<unsigned>(T, 1u), 256);
            {
                ProfScope ps(ops.prof, s, "k_f_count", (double)T * 24.0);
                hipLaunchKernelGGL(k_f_pre, dim3(gS), dim3(FB), 0, s, ix, st, (int)fm.size());
                hipLaunchKernelGGL(k_f_touch, dim3(gW), dim3(FWB), 0, s, ix, st);
                hipLaunchKernelGGL(k_f_count, dim3(gW), dim3(FWB), 0, s, ix, st, 0u);
                hipLaunchKernelGGL(k_f_linkpre1, dim3(gL), dim3(256), 0, s, ix, st);
            }
            {
                ProfScope ps(ops.prof, s, "k_f_link", (double)T * 24.0);
                hipLaunchKernelGGL(k_f_linkpre2, dim3(gL), dim3(256), 0, s, ix, st);
                hipLaunchKernelGGL(k_f_link, dim3(gW), dim3(FWB), 0, s, ix, st);
            }
            {
                ProfScope ps(ops.prof, s, "k_f_label", (double)T * 24.0);
                hipLaunchKernelGGL(k_f_acct, dim3(gS), dim3(FB), 0, s, ix, st);
                hipLaunchKernelGGL(k_f_labelpre, dim3(gS), dim3(FB), 0, s, ix, st);
                hipLaunchKernelGGL(k_f_label, dim3(gW), dim3(FWB), 0, s, ix, st);
            }
            hipLaunchKernelGGL(k_f_pick, dim3(std::max(1u, std::min(gT, 64u))), dim3(256), 0, s, ix, st);
            hipLaunchKernelGGL(k_f_keep, dim3(gT), dim3(256), 0, s, ix, st);
            HMSG_CHECK_LAUNCH();
            hmsg_scan_u32(keep.p, pos.p, (size_t)T, s, ops.scan_tmp, nullptr);
            hipLaunchKernelGGL(k_f_emit, dim3(gT), dim3(256), 0, s, ix, st, ins);
            hipLaunchKernelGGL(k_ix_grow, dim3(std::min(gT, (unsigned)n_cu * 4u)), dim3(256), 0, s, ix, (const unsigned*)touched_cells.p);
            hipLaunchKernelGGL(k_ix_write, dim3(gT), dim3(256), 0, s, ix, ins);
            HMSG_CHECK_LAUNCH();
            const size_t rb = (size_t)NCOMP * sizeof(FRes);
            h_res.ensure(rb + FC_N * 4);
            HIP_TRY(hipMemcpyAsync(h_res.p, d_res.p, rb, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(h_res.p + rb, ix_counters.p, FC_N * 4, hipMemcpyDeviceToHost, s));
            if (segs.empty()) spin.wait(s);
            hres = (FRes*)h_res.p;
            if (st.dbgbuf) {
                std::vector<unsigned long long> hb((size_t)T * 2);
                HIP_TRY(hipStreamSynchronize(s));
                HIP_TRY(hipMemcpy(hb.data(), dbgbuf.p, (size_t)T * 16, hipMemcpyDeviceToHost));
                std::vector<std::pair<unsigned long long, unsigned long long>> v;
                for (unsigned i = 0; i < T; ++i)
                    if (hb[(size_t)i * 2]) v.emplace_back(hb[(size_t)i * 2], hb[(size_t)i * 2 + 1]);
                std::sort(v.begin(), v.end());
                fprintf(stderr, "[fold dbg] step %d: %zu link walks of %u slots, %d components\n", (int)fstat[0], v.size(), T, NCOMP);
                for (size_t q = 0; q < v.size(); q += std::max<size_t>(1, v.size() / 24)) {
                    const unsigned long long a = v[q].first, b = v[q].second;
                    fprintf(stderr, "[fold dbg]  rank %5zu: %7.1f us  recs %llu trips %llu cells %llu rounds %llu groups %llu members %llu need %llu own %llu am %llu\n", q,
                            (a >> 32) * 0.01, (a >> 12) & 0xfffff, a & 0xfff, b >> 48, (b >> 40) & 0xff, (b >> 32) & 0xff, (b >> 24) & 0xff, (b >> 16) & 0xff,
                            (b >> 8) & 1, b & 1);
                }
                const unsigned long long a = v.empty() ? 0 : v.back().first, b = v.empty() ? 0 : v.back().second;
                fprintf(stderr, "[fold dbg]  slowest    : %7.1f us  recs %llu trips %llu cells %llu rounds %llu groups %llu members %llu need %llu own %llu am %llu\n",
                        (a >> 32) * 0.01, (a >> 12) & 0xfffff, a & 0xfff, b >> 48, (b >> 40) & 0xff, (b >> 32) & 0xff, (b >> 24) & 0xff, (b >> 16) & 0xff, (b >> 8) & 1, b & 1);
            }
        }
        // ---- the batch path for the few large components without an anchor
        std::vector<DbscanResult> res;
        long long batch_base = 0;
        if (!segs.empty()) {
            concat.ensure((size_t)cat_total * 3);
            d_cat.ensure(cat.size());
            HIP_TRY(hipMemcpyAsync(d_cat.p, cat.data(), cat.size() * sizeof(CatSeg), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_concat, dim3(cat_blocks), dim3(256), 0, s, (const double*)pool.p, (const CatSeg*)d_cat.p, (int)cat.size(),
                               concat.p, (const unsigned char*)nullptr, (unsigned char*)nullptr, CAT_CHUNK);
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
            const bool changed = r.n_kept != q.n_active;
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
            k.cap = (int)std::min<long long>(2ll * q.n_active, 0x7fffffff);
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
