// Segmented point-cloud primitives shared by the merge (A6), pooling (A7) and object (A10) stages:
//   * exact DBSCAN of K independent clouds in one batch of launches (Open3D ClusterDBSCAN semantics +
//     the keep-largest-cluster wrapper of graph_utils.py:827-880),
//   * voxel_down_sample of K clouds (Open3D semantics, canonical order),
//   * per-cloud uniform grids + "fraction of X within r of Y" counting (find_overlapping_ratio_faiss).
#pragma once
#include "hmsg_common.h"

#include <functional>

struct SegDesc {                 // one cloud of a batch, points at src[pt_base .. pt_base+n)
    long long pt_base;
    int n;
    double mn[3], mx[3];         // AABB (host knows it: union of member boxes / reduction result)
    int n_first = 0;             // optional: the segment's first n_first points are one member cloud -- the result then says how many of them were kept
    // Optional crop (forced = 1): the first member is an ANCHOR (fixed single-cluster cloud, core flags given in core0) larger than
    // the rest of the segment together, so every point of it is kept and its cluster wins the keep-largest rule whatever the
    // rest does.  Only its points inside [cmn, cmx] -- the other members' boxes grown by 2 eps -- can change status, connect to
    // a new point or be a new point's witness; the others are left out of the grid altogether (kept, core flag as given).
    // fmn / fmx: the first member's own AABB (the result's box is its union with the box of the kept rest).
    int forced = 0;
    double cmn[3] = {0, 0, 0}, cmx[3] = {0, 0, 0}, fmn[3] = {0, 0, 0}, fmx[3] = {0, 0, 0};
    // Where the segment's output goes (gather mode only: the batch is assembled from a pool the caller may write to).
    //   0  densely behind the outputs of the other mode-0 segments at `dst` (the only mode without a gather);
    //   1  to its own region of the pool, DbGather::pool_w + out_off (points): every kept point is copied there;
    //   2  IN PLACE (forced segments only): the first member already sits at pool_w + out_off with room behind it for the whole
    //      rest of the segment -- it is not copied at all, the kept rest is appended behind it, and the core flags the batch
    //      promotes among its points are set where they are (DbGather::poolcore_w).  The members the anchor cloud leaves
    //      outside its crop are then not read beyond their coordinates.
    int out_mode = 0;
    long long out_off = 0;
};

// A batch assembled from pieces of a point pool: piece [src, src + n) of the pool goes to [dst, dst + n) of the batch
// (the pieces tile the batch in order).  The merge concatenates the members of every component this way
// (merge_point_clouds_list, graph_utils.py:667-679); dbscan_keep_largest can do the copy inside its binning pass.
struct CatSeg {
    long long src, dst;
    int n, anchor;          // anchor: copy the member's persisted core flags (else the flags are cleared)
    int blk0, pad;          // first workgroup of this segment (work list of k_concat)
};
struct DbGather {
    const double* pool = nullptr;            // source points
    const CatSeg* segs = nullptr;            // device table -- or
    const CatSeg* host_segs = nullptr;       // the table on the host: it travels with the batch's geometry table (one upload)
    int nsegs = 0;
    const unsigned char* poolcore = nullptr; // persisted core flags of the pool points (anchor pieces)
    unsigned char* dstcore = nullptr;        // core0 of the batch, written by the gather (may be null)
    double* pool_w = nullptr;                // the same pool, writable: output regions of SegDesc::out_mode 1 / 2
    unsigned char* poolcore_w = nullptr;     // ... and their core flags
};

struct DbscanResult {            // per segment
    int n_out;
    double mn[3], mx[3];
    int changed;                 // 0: output == input (all points kept)
    int n_clusters;              // clusters DBSCAN found in the segment
    int contested;               // some border point had cores of two clusters within eps (only looked for when n_clusters > 1)
    int first_kept;              // kept points among the segment's first SegDesc::n_first (they come first in the output, in order); -1: not asked for
};

// A few hundred bytes of results go from the device to the host without the copy engine: one small workgroup writes them
// into pinned host memory, fences at system scope and raises a sequence flag the host spins on (a blit kernel + stream
// event per read-back cost ~7 us of the merge fold's ~400 us step, twice per step).
struct Publisher {
    PinnedBuf<unsigned> buf;         // payload, then (last word) the flag
    unsigned seq = 0;
    const unsigned* inited = nullptr;
    size_t inited_n = 0;
    hipStream_t stream = nullptr;    // of the last launch (wait() asks it whether it is still alive)
    // enqueue on s: copy src[0 .. n) (device) to the host buffer; data() is valid after wait()
    void launch(hipStream_t s, const unsigned* src, size_t n);
    void wait();
    const unsigned* data() const { return buf.p; }
};

struct CloudOps {
    hipStream_t s = nullptr;
    Prof* prof = nullptr;        // optional live timing of the heavy kernels
    // work counters of the DBSCAN batches (debug output of the merge stage)
    double stat_inplace = 0;                            // ... of them run in place (SegDesc::out_mode 2)
    double stat_forced = 0, stat_forced_first = 0;     // segments binned with their anchor member cropped, and those members' points
    double stat_calls = 0, stat_points = 0, stat_cells = 0, stat_core_cells = 0, stat_needy = 0, stat_maxcell_sum = 0, stat_maxcell_max = 0;
    DevBuf<unsigned> scan_tmp;
    // scratch (grown on demand)
    DevBuf<unsigned> cnt, start, cursor, ord, minidx, firstidx, size, flags, pos, rootmin;
    DevBuf<int> parent, label, segid, cellpos, corelist, cseg, roots, rhead, rnext;   // roots / rhead / rnext: a segment's cluster roots (k_db_rootmin)
    DevBuf<double> cellbox;
    DevBuf<long long> cellid;
    DevBuf<unsigned char> core, score;     // core flag per point / per slot of the cell-sorted copy
    DevBuf<double> spts;                   // cell-sorted copy of the batch's points
    DevBuf<unsigned long long> best, obounds;
    DevBuf<unsigned> rep, active, kres, ccore, needy, nclist, sidx;    // sidx: the point of every slot of the cell-sorted copy
    DevBuf<unsigned char> hasanchor;
    DevBuf<char> geom;           // device copy of per-segment geometry tables
    DevBuf<unsigned long long> vbitmap;
    DevBuf<unsigned> vrank;
    Publisher pub;               // per-segment results of a DBSCAN batch, read back every fold step
    PinnedBuf<char> h_geom;      // staging of the DBSCAN batch geometry table (uploaded every fold step)
    SpinWait spin;
    SortBufs vsort;              // ordered voxel sums: (slot, point index) records
    DevBuf<unsigned> voff;

    // fill segs[k].mn / mx from the points (device reduction, one sync)
    void bounds(const double* src, std::vector<SegDesc>& segs);
    // keep-largest-cluster DBSCAN of every segment; outputs are written consecutively to `dst`
    // (capacity >= total input points); returns total output points.
    // core0 (optional, one byte per input point): 1 = ANCHOR point -- a core point of a member cloud that is a
    // known single-cluster fixed point of this very DBSCAN; at most one such member per segment.  Anchor points
    // are core and mutually connected by construction, so they are neither counted nor re-connected; the
    // result is identical to a run without the hint.  dst_core (optional): core flag of every output point.
    long long dbscan_keep_largest(const double* src, const std::vector<SegDesc>& segs, double eps, int min_points,
                                  double* dst, std::vector<DbscanResult>& res, const unsigned char* core0 = nullptr,
                                  unsigned char* dst_core = nullptr, const DbGather* gather = nullptr,
                                  const std::function<void(const unsigned* d_res, int K)>& behind_publish = nullptr);
    // behind_publish (optional): called when every launch of the batch AND the publish of its results are enqueued, before the
    // host waits for them -- work the caller enqueues there runs on the GPU while the host reads the results and does its
    // bookkeeping (the merge fold indexes the batch's output clouds that way: their sizes are read from d_res on the device).
    // d_res = the batch's result words as they will be published: per segment k of K
    //   [k] output end | [K + k] clusters | [2K + k] contested | [3K + k] dropped | [4K ..] 6 x u64 box | [16K .. 16K + 8) counters |
    //   [16K + 8 + k] output start | [17K + 8 + k] end of the first member's output   (positions in the segment's output chain)
    // gather: `src` is an EMPTY buffer of the batch's size; the binning pass fills it (and core0's buffer, gather->dstcore)
    // from the pool pieces while it bins -- one launch and one pass over the points less than a separate concatenation.
    // Open3D voxel_down_sample of every segment; outputs consecutively to dst (capacity >= total input
    // points); out_n[k] = points of segment k.
    // SegDesc::out_mode 1 / 2 are available (not with the legacy three-launch compaction or with HMSG_DEBUG_NO_CROP=1)
    static bool regions_supported();
    long long voxel_down_sample(const double* src, const std::vector<SegDesc>& segs, double vs, double* dst,
                                std::vector<int>& out_n);
};
