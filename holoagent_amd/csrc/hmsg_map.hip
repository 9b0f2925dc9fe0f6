// A1 + A2: RGB-D back-projection of every resident frame, voxel_down_sample into the global cloud,
// remove_radius_outlier, and the occupancy-bitmap + rank index that replaces cKDTree.
//
// Reference: dataloader/generic.py:74-138 (create_pcd), graph/graph.py:339-364 (loop A, filtering,
// tree build).  Open3D semantics restated in oracle/hmsg_oracle.py (o3d_voxel_down_sample,
// o3d_remove_radius_outlier).
//
// MI355X design: all frames stay resident in HBM (depth u16 + rgb u8 = 1.5 MB/frame), so the global
// min bound (which fixes Open3D's voxel grid origin) is one streaming reduction, and voxelisation is
// "sort by counting": pass 1 marks an occupancy bitmap laid out (ix, iy, iz)-major, a popcount prefix
// (rank) turns a cell into its slot in canonical order, pass 2 accumulates per-slot sums with integer
// atomics (fixed-point offsets from the cell corner -> order-independent, bit-reproducible).
#include "hmsg_common.h"

#include <atomic>

#include <algorithm>


// ------------------------------------------------------------------------------------------ scans
// Exclusive prefix sum of u32, ONE launch: tiles of 1024 elements, decoupled look-back.  Every tile publishes
// its aggregate, then its inclusive prefix, in a 64-bit status word {epoch:30 | flag:2 | value:32}; a tile's
// first wave looks back over its predecessors 64 at a time until it meets a published prefix.  Words carry the
// epoch of the scan call, so the status table is never cleared (a stale word reads as "not ready").  Tiles are
// numbered by blockIdx: workgroups are dispatched in index order, so every predecessor of a resident tile is
// resident or finished and the spin cannot starve it.
__global__ void k_scan_lookback(const unsigned* __restrict__ in, unsigned* __restrict__ out, size_t n,
                                unsigned long long* __restrict__ state, unsigned epoch) {
    __shared__ unsigned wsum[4];
    __shared__ unsigned s_prefix;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long long tile = blockIdx.x;
    const size_t base = ((size_t)tile * 256 + tid) * 4;
    unsigned v[4];
    unsigned t = 0;
    for (int i = 0; i < 4; ++i) {
        v[i] = base + i < n ? in[base + i] : 0u;
        t += v[i];
    }
    unsigned incl = t;   // inclusive wave scan
    for (int o = 1; o < 64; o <<= 1) {
        unsigned u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    unsigned woff = 0;
    for (int i = 0; i < w; ++i) woff += wsum[i];
    const unsigned agg = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (w == 0) {
        const unsigned prefix = scan_lookback_prefix(state, tile, epoch, agg);
        if (lane == 0) s_prefix = prefix;
    }
    __syncthreads();
    unsigned excl = s_prefix + woff + incl - t;
    for (int i = 0; i < 4; ++i) {
        if (base + i < n) out[base + i] = excl;
        excl += v[i];
    }
}

unsigned hmsg_scan_epoch(DevBuf<unsigned>& tmp, size_t ntiles, hipStream_t s) {
    static std::atomic<unsigned> g_epoch{0};
    const unsigned epoch = (g_epoch.fetch_add(1) + 1u) & 0x3fffffffu;
    if (tmp.n < ntiles * 2 + 64) {                        // (status words are u64: two u32 slots each)
        tmp.ensure(ntiles * 2 + 64);
        HIP_TRY(hipMemsetAsync(tmp.p, 0, tmp.n * 4, s));  // fresh memory: no word may look like a current epoch
    }
    return epoch;
}

void hmsg_scan_u32(const unsigned* in, unsigned* out, size_t n, hipStream_t s, DevBuf<unsigned>& tmp,
                   unsigned long long* total) {
    if (n == 0) {
        if (total) *total = 0;
        return;
    }
    const size_t ntiles = (n + 1023) / 1024;
    const unsigned epoch = hmsg_scan_epoch(tmp, ntiles, s);
    unsigned last_in = 0, last_out = 0;
    if (total) HIP_TRY(hipMemcpyAsync(&last_in, in + n - 1, 4, hipMemcpyDeviceToHost, s));
    hipLaunchKernelGGL(k_scan_lookback, dim3((unsigned)ntiles), dim3(256), 0, s, in, out, n,
                       reinterpret_cast<unsigned long long*>(tmp.p), epoch);
    HMSG_CHECK_LAUNCH();
    if (total) {
        HIP_TRY(hipMemcpyAsync(&last_out, out + n - 1, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        *total = (unsigned long long)last_in + last_out;
    }
}

__global__ void k_popc_words(const unsigned long long* __restrict__ bm, unsigned* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (unsigned)__popcll(bm[i]);
}

unsigned long long hmsg_bitmap_rank(const unsigned long long* bitmap, unsigned* rank, size_t nwords, hipStream_t s,
                                    DevBuf<unsigned>& tmp) {
    hipLaunchKernelGGL(k_popc_words, dim3(cdiv(nwords, 256)), dim3(256), 0, s, bitmap, rank, nwords);
    HMSG_CHECK_LAUNCH();
    unsigned long long total = 0;
    hmsg_scan_u32(rank, rank, nwords, s, tmp, &total);
    return total;
}

// ------------------------------------------------------------------------------------------ bounds
// Pixel index -> (frame, row, column).  `i / HW` and `p / W` as a 64-bit and a 32-bit integer division per pixel cost these
// streaming passes about as much as the back-projection itself (a 64-bit udiv is ~80 instructions here).  The passes walk the
// pixels in chunks of PIX_CHUNK consecutive indices: the chunk's first frame is one division per chunk, a pixel of the chunk is at
// most a few frames behind it (subtractions), and the row comes from a float32 reciprocal with a one-step correction (exact for
// p < 2^24; larger frames take the integer division).
#define PIX_CHUNK 4096
struct PixWalk {
    unsigned hw, w;
    float inv_w;
    unsigned f0, p0;        // frame / in-frame index of the chunk's first pixel
};
__device__ __forceinline__ PixWalk pix_walk(size_t chunk_first, size_t HW, int W) {
    PixWalk k;
    k.hw = (unsigned)HW;
    k.w = (unsigned)W;
    k.inv_w = 1.0f / (float)W;
    k.f0 = (unsigned)(chunk_first / HW);
    k.p0 = (unsigned)(chunk_first - (size_t)k.f0 * HW);
    return k;
}
// frames [f0, pix_last_frame] are the ones a chunk of n pixels touches (two at most while a frame is larger than a chunk).  The passes
// below walk a chunk once per frame it touches with that frame's pose taken from a WAVE-UNIFORM address: sixteen scalar loads per
// chunk side instead of sixteen vector loads of one and the same address per pixel (which is what bounded k_bounds: its 0.6 GB of depth
// stream in 0.1 ms, its 5 * 10^9 redundant pose loads took 1).
__device__ __forceinline__ unsigned pix_last_frame(const PixWalk& k, unsigned n) {
    return k.f0 + (unsigned)(((unsigned long long)k.p0 + n - 1ull) / k.hw);
}
__device__ __forceinline__ void pix_at(const PixWalk& k, unsigned d /* pixels behind the chunk's first */, int& f, int& x, int& y) {
    unsigned p = k.p0 + d, ff = k.f0;
    while (p >= k.hw) {
        p -= k.hw;
        ++ff;
    }
    unsigned yy;
    if (k.hw <= (1u << 24)) {
        yy = (unsigned)((float)p * k.inv_w);
        while (yy * k.w > p) --yy;
        while ((yy + 1u) * k.w <= p) ++yy;
    } else {
        yy = p / k.w;
    }
    f = (int)ff;
    y = (int)yy;
    x = (int)(p - yy * k.w);
}
__global__ void k_bounds(const unsigned short* __restrict__ depth, const double* __restrict__ pose, CamK cam,
                         float scale, int H, int W, int F, unsigned long long* __restrict__ out /*[6] enc min3,max3*/,
                         unsigned long long* __restrict__ npts) {
    const size_t HW = (size_t)H * W;
    const size_t total = HW * F;
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    unsigned cnt = 0;
    for (size_t c0 = (size_t)blockIdx.x * PIX_CHUNK; c0 < total; c0 += (size_t)gridDim.x * PIX_CHUNK) {
        const PixWalk pw = pix_walk(c0, HW, W);
        const unsigned npix = (unsigned)(total - c0 < (size_t)PIX_CHUNK ? total - c0 : (size_t)PIX_CHUNK);
        const unsigned fl = pix_last_frame(pw, npix);
        for (unsigned ff = pw.f0; ff <= fl; ++ff) {                 // (uniform: the chunk's frames, one or two)
            const double* const T = pose + (size_t)ff * 16;
            for (unsigned d = threadIdx.x; d < npix; d += blockDim.x) {
                const size_t g = c0 + d;
                int f, x, y;
                pix_at(pw, d, f, x, y);
                double w[3];
                if ((unsigned)f == ff && backproject(depth[g], x, y, cam, scale, T, w[0], w[1], w[2])) {
                    for (int a = 0; a < 3; ++a) {
                        mn[a] = w[a] < mn[a] ? w[a] : mn[a];
                        mx[a] = w[a] > mx[a] ? w[a] : mx[a];
                    }
                    cnt++;
                }
            }
        }
    }
    for (int a = 0; a < 3; ++a) {
        mn[a] = wave_min_f64(mn[a]);
        mx[a] = wave_max_f64(mx[a]);
    }
    int c = wave_sum_i32((int)cnt);
    if ((threadIdx.x & 63) == 0) {
        for (int a = 0; a < 3; ++a) {
            atomicMin(&out[a], enc_f64(mn[a]));
            atomicMax(&out[3 + a], enc_f64(mx[a]));
        }
        atomicAdd(npts, (unsigned long long)c);
    }
}

// ------------------------------------------------------------------------------------------ mark + accumulate
__global__ void k_mark(const unsigned short* __restrict__ depth, const double* __restrict__ pose, CamK cam, float scale,
                       int H, int W, int F, GridGeom g, unsigned long long* __restrict__ bitmap) {
    const size_t HW = (size_t)H * W;
    const size_t total = HW * F;
    for (size_t c0 = (size_t)blockIdx.x * PIX_CHUNK; c0 < total; c0 += (size_t)gridDim.x * PIX_CHUNK) {
        const PixWalk pw = pix_walk(c0, HW, W);
        const unsigned npix = (unsigned)(total - c0 < (size_t)PIX_CHUNK ? total - c0 : (size_t)PIX_CHUNK);
        const unsigned fl = pix_last_frame(pw, npix);
        for (unsigned ff = pw.f0; ff <= fl; ++ff) {                 // (uniform: the chunk's frames, one or two)
            const double* const T = pose + (size_t)ff * 16;
            for (unsigned d = threadIdx.x; d < npix; d += blockDim.x) {
                const size_t i = c0 + d;
                int f, x, y;
                pix_at(pw, d, f, x, y);
                double wx, wy, wz;
                if ((unsigned)f != ff || !backproject(depth[i], x, y, cam, scale, T, wx, wy, wz)) continue;
                int ix, iy, iz;
                cell_of(g, wx, wy, wz, ix, iy, iz);
                long long lin = lin_of(g, ix, iy, iz);
                unsigned long long bit = 1ull << (lin & 63);
                unsigned long long* wp = bitmap + (lin >> 6);
                if (!(*wp & bit)) atomicOr(wp, bit);
            }
        }
    }
}

struct VoxAcc {                 // per-slot colour / count accumulators (SoA)
    unsigned long long* sr;
    unsigned long long* sg;
    unsigned long long* sb;
    unsigned* n;
};

// ---- ordered voxel sums (A2) -------------------------------------------------------------------------------
// Open3D's VoxelDownSample adds the points of a voxel in INPUT order (frame-major, pixel row-major here:
// graph.py:339-348) in float64, and the nearest-neighbour decisions downstream hinge on the last bit of the
// centroids, so the sums are taken in exactly that order:
//   k_slots      pixel -> voxel slot (u32, kept for phase 2), colours / counts per run with integer atomics,
//                number of runs per 4096-pixel chunk.  A run = consecutive pixels of one image row (inside one
//                64-pixel wave slice) that fall into the same voxel (~8 pixels).
//   k_emit_runs  run records (key = slot, value = frame << 32 | row << 20 | first column << 8 | length) in input order
//   stable sort by slot (hmsg_sort.hip), segment starts
//   k_accum_ordered  one lane per voxel walks its runs in order: back-project again, s += p, centroid = s / n
#define RUN_CHUNK 4096
__device__ __forceinline__ unsigned pixel_slot(const unsigned short* __restrict__ depth, const double* __restrict__ T /* the frame's pose */,
                                               const CamK& cam, float scale, size_t i, int x, int y, const GridGeom& g,
                                               const unsigned long long* __restrict__ bitmap, const unsigned* __restrict__ rank) {
    double wx, wy, wz;
    if (!backproject(depth[i], x, y, cam, scale, T, wx, wy, wz)) return 0xffffffffu;
    int ix, iy, iz;
    cell_of(g, wx, wy, wz, ix, iy, iz);
    const long long lin = lin_of(g, ix, iy, iz);
    const unsigned long long word = bitmap[lin >> 6];
    return rank[lin >> 6] + (unsigned)__popcll(word & ((1ull << (lin & 63)) - 1ull));
}

__global__ void __launch_bounds__(256) k_slots(const unsigned short* __restrict__ depth, const unsigned char* __restrict__ rgb,
                                               const double* __restrict__ pose, CamK cam, float scale, int H, int W, int F,
                                               GridGeom g, const unsigned long long* __restrict__ bitmap,
                                               const unsigned* __restrict__ rank, VoxAcc acc, unsigned* __restrict__ slots,
                                               unsigned* __restrict__ chunk_runs) {
    __shared__ unsigned s_runs;
    if (threadIdx.x == 0) s_runs = 0u;
    __syncthreads();
    const size_t HW = (size_t)H * W;
    const size_t total = HW * F;
    const int lane = threadIdx.x & 63;
    unsigned myruns = 0;
    const PixWalk pw = pix_walk((size_t)blockIdx.x * RUN_CHUNK, HW, W);
    const size_t left = total - (size_t)blockIdx.x * RUN_CHUNK;
    const unsigned fl = pix_last_frame(pw, (unsigned)(left < (size_t)RUN_CHUNK ? left : (size_t)RUN_CHUNK));
    for (int it = 0; it < RUN_CHUNK / 256; ++it) {
        const size_t i = (size_t)blockIdx.x * RUN_CHUNK + (size_t)it * 256 + threadIdx.x;
        const bool in_range = i < total;
        unsigned slot = 0xffffffffu;
        unsigned cr = 0, cg = 0, cb = 0, cn = 0;
        int x = 0;
        if (in_range) {
            int f, y;
            pix_at(pw, (unsigned)it * 256u + threadIdx.x, f, x, y);
            for (unsigned ff = pw.f0; ff <= fl; ++ff)                // (uniform: the chunk's frames, one or two; pose at a uniform address)
                if ((unsigned)f == ff) slot = pixel_slot(depth, pose + (size_t)ff * 16, cam, scale, i, x, y, g, bitmap, rank);
            slots[i] = slot;
            if (slot != 0xffffffffu) {
                const unsigned char* c = rgb + i * 3;
                cr = c[0];
                cg = c[1];
                cb = c[2];
                cn = 1;
            }
        }
        // segmented inclusive scan over runs of equal slot within an image row (integer sums: exact in any order)
        const unsigned prev = __shfl_up(slot, 1);
        int flag = (lane == 0 || prev != slot || x == 0) ? 1 : 0;
        const int head = flag;
        for (int d = 1; d < 64; d <<= 1) {
            unsigned tr = __shfl_up(cr, d), tg = __shfl_up(cg, d), tb = __shfl_up(cb, d), tn = __shfl_up(cn, d);
            int tf = __shfl_up(flag, d);
            if (lane >= d && !flag) {
                cr += tr; cg += tg; cb += tb; cn += tn;
                flag = tf;
            }
        }
        const int nhead = __shfl_down(head, 1);
        const bool tail = lane == 63 || nhead != 0;
        if (tail && slot != 0xffffffffu) {
            atomicAdd(&acc.sr[slot], (unsigned long long)cr);
            atomicAdd(&acc.sg[slot], (unsigned long long)cg);
            atomicAdd(&acc.sb[slot], (unsigned long long)cb);
            atomicAdd(&acc.n[slot], cn);
            ++myruns;
        }
    }
    myruns = (unsigned)wave_sum_i32((int)myruns);
    if (lane == 0 && myruns) atomicAdd(&s_runs, myruns);
    __syncthreads();
    if (threadIdx.x == 0) chunk_runs[blockIdx.x] = s_runs;
}

__global__ void __launch_bounds__(256) k_emit_runs(const unsigned* __restrict__ slots, size_t total, int W, size_t HW,
                                                   const unsigned* __restrict__ chunk_base, unsigned* __restrict__ keys,
                                                   unsigned long long* __restrict__ vals) {
    __shared__ unsigned s_w[4];
    __shared__ unsigned s_run;
    if (threadIdx.x == 0) s_run = chunk_base[blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const PixWalk pw = pix_walk((size_t)blockIdx.x * RUN_CHUNK, HW, W);
    for (int it = 0; it < RUN_CHUNK / 256; ++it) {
        const size_t i = (size_t)blockIdx.x * RUN_CHUNK + (size_t)it * 256 + threadIdx.x;
        const bool in_range = i < total;
        const unsigned slot = in_range ? slots[i] : 0xffffffffu;
        int pf = 0, x = 0, py = 0;
        if (in_range) pix_at(pw, (unsigned)it * 256u + threadIdx.x, pf, x, py);
        const unsigned prev = __shfl_up(slot, 1);
        const bool head = lane == 0 || prev != slot || x == 0;
        const unsigned long long heads = __ballot(head);
        const int nhead = __shfl_down(head ? 1 : 0, 1);
        const bool tail = (lane == 63 || nhead != 0) && slot != 0xffffffffu;
        const unsigned long long tails = __ballot(tail);
        if (lane == 0) s_w[w] = (unsigned)__popcll(tails);
        __syncthreads();
        unsigned base = s_run;
        for (int q = 0; q < w; ++q) base += s_w[q];
        if (tail) {
            const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
            const int start_lane = 63 - __clzll(below);
            const unsigned len = (unsigned)(lane - start_lane + 1);
            const unsigned pos = base + (unsigned)__popcll(tails & ((1ull << lane) - 1ull));
            keys[pos] = slot;
            // (a run lies inside one image row: its first pixel is this lane's pixel moved left)
            vals[pos] = ((unsigned long long)(unsigned)pf << 32) | ((unsigned long long)(unsigned)py << 20) |
                        ((unsigned long long)(unsigned)(x - (lane - start_lane)) << 8) | (unsigned long long)len;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_run += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
}

// centroid = (sequential float64 sum of the voxel's points in input order) / count  (o3d_voxel_down_sample).
// One WAVE per voxel: the additions are a serial chain by definition, but the back-projections that feed it are
// not.  The wave takes as many consecutive runs as fit its 64 lanes (runs are ~8 pixels), every lane back-projects
// one pixel, and the points are then folded into the sum in order with broadcasts (uniform across lanes, so every
// lane holds the sum).  A voxel seen from close by has thousands of runs: packing them 64 pixels at a time keeps
// that chain -- the kernel lasts as long as the longest one -- short.
// Round 6: the kernel is bound by the instructions it issues, not by the depth lines it fetches nor by the length of a chain (the
// heaviest voxel of configs[1] holds 66 000 of the 3 * 10^8 pixels; profiles/r06_map_accum.txt).  With every lane carrying the sum, a pixel
// costs the wave 6 lane reads and 3 additions.  LDS_SUM: the 64 points of a pack go to LDS and THREE lanes walk them, lane a adding
// coordinate a -- one LDS read and one addition per pixel, issued once for the three chains; the points are read in the order they
// would have been broadcast in, so every sum sees the same additions in the same order.  (LDS_SUM = false: the form until round 5,
// HMSG_DEBUG_ACCUM_WAVES=1, kept as the comparison route of the tests.)
template <bool LDS_SUM>
__global__ void __launch_bounds__(256) k_accum_ordered(const unsigned short* __restrict__ depth, const double* __restrict__ pose,
                                                       CamK cam, float scale, int W, size_t HW, long long V,
                                                       const unsigned* __restrict__ off, const unsigned long long* __restrict__ runs,
                                                       const unsigned* __restrict__ cnt, double* __restrict__ pts) {
    __shared__ double s_pt[LDS_SUM ? 4 : 1][LDS_SUM ? 64 * 3 : 1];
    const int lane = threadIdx.x & 63;
    double* const my_pt = s_pt[LDS_SUM ? (threadIdx.x >> 6) : 0];
    const long long v = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (v >= V) return;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    const unsigned r0 = off[v], r1 = off[v + 1];
    for (unsigned rb = r0; rb < r1; rb += 64) {
        // lane q holds run rb + q of this group of 64 runs; inclusive prefix of the run lengths
        const unsigned long long myrec = rb + lane < r1 ? runs[rb + lane] : 0ull;
        const int mylen = (int)(myrec & 255ull);
        int incl = mylen;
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        const int nrun = (int)min(64u, r1 - rb);
        int q0 = 0;                                   // first run of the next pack
        while (q0 < nrun) {
            const int base = q0 ? __shfl(incl, q0 - 1) : 0;
            // runs q0 .. q1-1 fit into 64 lanes (a run is at most 64 long, so at least one does)
            const unsigned long long fits = __ballot(lane >= q0 && lane < nrun && incl - base <= 64);
            const int q1 = q0 + __popcll(fits);
            const int npix = __shfl(incl, q1 - 1) - base;
            // lane L back-projects pixel L of the pack: find its run (uniform loop over the few runs of the pack)
            double wx = 0.0, wy = 0.0, wz = 0.0;
            unsigned long long rec = 0ull;
            int start = 0;
            for (int q = q0; q < q1; ++q) {
                const unsigned long long rq = __shfl(myrec, q);
                const int eq = __shfl(incl, q) - base;
                const int sq = eq - (int)(rq & 255ull);
                if (lane >= sq && lane < eq) {
                    rec = rq;
                    start = sq;
                }
            }
            if (lane < npix) {
                const int f = (int)(rec >> 32);
                const int y = (int)((rec >> 20) & 0xfffull), x0 = (int)((rec >> 8) & 0xfffull);
                const int x = x0 + (lane - start);
                backproject(depth[(size_t)f * HW + (size_t)y * W + x], x, y, cam, scale, pose + (size_t)f * 16, wx, wy, wz);
            }
            if (LDS_SUM) {
                // (one wave, its own slice of LDS: the LDS queue keeps a wave's accesses in order, the barrier keeps the compiler from
                //  moving them across each other)
                my_pt[lane * 3] = wx;
                my_pt[lane * 3 + 1] = wy;
                my_pt[lane * 3 + 2] = wz;
                __builtin_amdgcn_wave_barrier();
                if (lane < 3)
                    for (int j = 0; j < npix; ++j) sx = __dadd_rn(sx, my_pt[j * 3 + lane]);     // (lane a: coordinate a, in sx)
                __builtin_amdgcn_wave_barrier();
            } else {
                for (int j = 0; j < npix; ++j) {
                    sx = __dadd_rn(sx, wave_bcast_f64(wx, j));
                    sy = __dadd_rn(sy, wave_bcast_f64(wy, j));
                    sz = __dadd_rn(sz, wave_bcast_f64(wz, j));
                }
            }
            q0 = q1;
        }
    }
    if (LDS_SUM) {
        if (lane < 3) pts[v * 3 + lane] = __ddiv_rn(sx, (double)cnt[v]);
    } else if (lane == 0) {
        const double n = (double)cnt[v];
        pts[v * 3 + 0] = __ddiv_rn(sx, n);
        pts[v * 3 + 1] = __ddiv_rn(sy, n);
        pts[v * 3 + 2] = __ddiv_rn(sz, n);
    }
}

// slot -> cell coordinates (one thread per bitmap word)
__global__ void k_slot_cells(const unsigned long long* __restrict__ bitmap, const unsigned* __restrict__ rank,
                             GridGeom g, int* __restrict__ cell /*[V][3]*/) {
    long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= g.nwords) return;
    unsigned long long bits = bitmap[w];
    unsigned s = rank[w];
    while (bits) {
        int b = __ffsll(bits) - 1;
        bits &= bits - 1;
        long long lin = w * 64 + b;
        int iz = (int)(lin % g.nzp);
        long long r = lin / g.nzp;
        int iy = (int)(r % g.ny);
        int ix = (int)(r / g.ny);
        cell[(size_t)s * 3 + 0] = ix;
        cell[(size_t)s * 3 + 1] = iy;
        cell[(size_t)s * 3 + 2] = iz;
        ++s;
    }
}

__global__ void k_finalize(VoxAcc acc, long long V, double* __restrict__ cols) {
    long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= V) return;
    double n = (double)acc.n[s];
    cols[s * 3 + 0] = ((double)acc.sr[s] / 255.0) / n;
    cols[s * 3 + 1] = ((double)acc.sg[s] / 255.0) / n;
    cols[s * 3 + 2] = ((double)acc.sb[s] / 255.0) / n;
}

// ------------------------------------------------------------------------------------------ radius outlier
// Open3D RemoveRadiusOutliers: keep p iff #{q : |p-q| < radius} (incl. p) > nb_points.  One wave per
// point; lanes sweep the (ix', iy') columns of the cube around p.  In a column the cells certainly
// inside the ball are counted with popcounts of the bitmap words, only the cells that straddle the
// sphere are tested point by point.
__device__ __forceinline__ unsigned long long bits_range(int lo, int hi) {   // bits lo..hi inclusive within a word
    if (hi < lo) return 0ull;
    unsigned long long m = (hi >= 63) ? ~0ull : ((1ull << (hi + 1)) - 1ull);
    return m & ~((1ull << lo) - 1ull);
}

__global__ void k_outlier(const double* __restrict__ pts, const int* __restrict__ cell, long long V, GridGeom g,
                          const unsigned long long* __restrict__ bitmap, const unsigned* __restrict__ rank,
                          double radius, int nb_points, unsigned* __restrict__ keep) {
    const int lane = threadIdx.x & 63;
    const long long s = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (s >= V) return;
    const double px = pts[s * 3], py = pts[s * 3 + 1], pz = pts[s * 3 + 2];
    const int cx = cell[s * 3], cy = cell[s * 3 + 1];
    const int R = (int)ceil(radius / g.vs) + 1;
    const int side = 2 * R + 1;
    const double r2 = radius * radius;
    const double slack = 1e-7;
    int count = 0;
    for (int c = lane; c < side * side; c += 64) {
        int ix = cx - R + c / side, iy = cy - R + c % side;
        if (ix < 0 || iy < 0 || ix >= g.nx || iy >= g.ny) continue;
        double xlo = g.ox + ix * g.vs, xhi = xlo + g.vs, ylo = g.oy + iy * g.vs, yhi = ylo + g.vs;
        double dxmin = fmax(0.0, fmax(xlo - px, px - xhi)), dxmax = fmax(fabs(px - xlo), fabs(px - xhi));
        double dymin = fmax(0.0, fmax(ylo - py, py - yhi)), dymax = fmax(fabs(py - ylo), fabs(py - yhi));
        dxmin = fmax(0.0, dxmin - slack);
        dymin = fmax(0.0, dymin - slack);
        dxmax += slack;
        dymax += slack;
        double rem_out = r2 - (dxmin * dxmin + dymin * dymin);
        if (rem_out <= 0.0) continue;
        double ho = sqrt(rem_out) + slack;
        int zo0 = (int)floor((pz - ho - g.oz) / g.vs), zo1 = (int)floor((pz + ho - g.oz) / g.vs);
        zo0 = zo0 < 0 ? 0 : zo0;
        zo1 = zo1 >= g.nz ? g.nz - 1 : zo1;
        if (zo1 < zo0) continue;
        double rem_in = r2 - (dxmax * dxmax + dymax * dymax);
        int zi0 = 1, zi1 = 0;   // empty
        if (rem_in > 0.0) {
            double hi_ = sqrt(rem_in) - slack;
            if (hi_ > 0.0) {
                zi0 = (int)ceil((pz - hi_ - g.oz) / g.vs);
                zi1 = (int)floor((pz + hi_ - g.oz) / g.vs) - 1;
                zi0 = zi0 < zo0 ? zo0 : zi0;
                zi1 = zi1 > zo1 ? zo1 : zi1;
            }
        }
        long long colw = ((long long)ix * g.ny + iy) * (g.nzp >> 6);
        for (int w = zo0 >> 6; w <= zo1 >> 6; ++w) {
            unsigned long long word = bitmap[colw + w];
            if (!word) continue;
            int b0 = w * 64;
            unsigned long long outer = word & bits_range(zo0 - b0 < 0 ? 0 : zo0 - b0, zo1 - b0 > 63 ? 63 : zo1 - b0);
            unsigned long long inner = 0ull;
            if (zi1 >= zi0) inner = outer & bits_range(zi0 - b0 < 0 ? 0 : zi0 - b0, zi1 - b0 > 63 ? 63 : zi1 - b0);
            count += __popcll(inner);
            unsigned long long shell = outer & ~inner;
            unsigned base = rank[colw + w];
            while (shell) {
                int b = __ffsll(shell) - 1;
                shell &= shell - 1;
                unsigned q = base + (unsigned)__popcll(word & ((1ull << b) - 1ull));
                double dx = pts[(size_t)q * 3] - px, dy = pts[(size_t)q * 3 + 1] - py, dz = pts[(size_t)q * 3 + 2] - pz;
                double d2 = dx * dx + dy * dy + dz * dz;
                count += d2 < r2 ? 1 : 0;
            }
        }
    }
    count = wave_sum_i32(count);
    if (lane == 0) keep[s] = count > nb_points ? 1u : 0u;
}

__global__ void k_compact(const unsigned* __restrict__ keep, const unsigned* __restrict__ newidx, long long V0,
                          const double* __restrict__ pts0, const double* __restrict__ cols0, const int* __restrict__ cell,
                          GridGeom g, double* __restrict__ pts, double* __restrict__ cols,
                          unsigned long long* __restrict__ bitmap) {
    long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= V0) return;
    if (keep[s]) {
        unsigned d = newidx[s];
        for (int a = 0; a < 3; ++a) {
            pts[(size_t)d * 3 + a] = pts0[s * 3 + a];
            cols[(size_t)d * 3 + a] = cols0[s * 3 + a];
        }
    } else {
        long long lin = lin_of(g, cell[s * 3], cell[s * 3 + 1], cell[s * 3 + 2]);
        atomicAnd(&bitmap[lin >> 6], ~(1ull << (lin & 63)));
    }
}

// ------------------------------------------------------------------------------------------ NN candidate lists
// (hmsg_nn.h) For every cell whose voxel was deleted: the cloud points that can be the nearest neighbour of
// SOME query inside the cell = { p : mindist(p, cell) <= min_p' maxdist(p', cell) }.  One wave per cell.
__global__ void k_rm_cells(const unsigned* __restrict__ keep, const unsigned* __restrict__ newidx, long long V0,
                           const int* __restrict__ cell, GridGeom g, int* __restrict__ rmcell,
                           unsigned long long* __restrict__ bitmap_rm) {
    long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= V0 || keep[s]) return;
    long long r = s - (long long)newidx[s];
    for (int a = 0; a < 3; ++a) rmcell[r * 3 + a] = cell[s * 3 + a];
    long long lin = lin_of(g, cell[s * 3], cell[s * 3 + 1], cell[s * 3 + 2]);
    atomicOr(&bitmap_rm[lin >> 6], 1ull << (lin & 63));
}

template <typename F>
__device__ __forceinline__ void ring_visit(const GridGeom& g, const unsigned long long* __restrict__ bitmap,
                                           const unsigned* __restrict__ rank, int cx, int cy, int cz, int r, int lane, F&& f) {
    const int side = 2 * r + 1;
    for (int ci = lane; ci < side * side; ci += 64) {
        int dx = ci / side - r, dy = ci % side - r;
        int ix = cx + dx, iy = cy + dy;
        if (ix < 0 || iy < 0 || ix >= g.nx || iy >= g.ny) continue;
        bool rim = (dx == -r || dx == r || dy == -r || dy == r);
        long long colw = ((long long)ix * g.ny + iy) * (g.nzp >> 6);
        for (int part = 0; part < (rim ? 1 : 2); ++part) {
            int z0 = rim ? cz - r : (part == 0 ? cz - r : cz + r);
            int z1 = rim ? cz + r : z0;
            z0 = z0 < 0 ? 0 : z0;
            z1 = z1 >= g.nz ? g.nz - 1 : z1;
            if (!rim && (z0 != z1 || (part == 0 ? cz - r < 0 : cz + r >= g.nz))) continue;
            if (z1 < z0) continue;
            for (int w = z0 >> 6; w <= z1 >> 6; ++w) {
                unsigned long long word = bitmap[colw + w];
                if (!word) continue;
                int b0 = w * 64;
                int lo = z0 - b0 < 0 ? 0 : z0 - b0, hi = z1 - b0 > 63 ? 63 : z1 - b0;
                unsigned long long sel = word & (hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull);
                unsigned base = rank[colw + w];
                while (sel) {
                    int b = __ffsll(sel) - 1;
                    sel &= sel - 1;
                    f((int)(base + (unsigned)__popcll(word & ((1ull << b) - 1ull))));
                }
            }
        }
    }
}

__device__ __forceinline__ void box_dists(const double* __restrict__ p, const double* lo, double vs, double& mn2, double& mx2) {
    mn2 = 0.0;
    mx2 = 0.0;
    for (int a = 0; a < 3; ++a) {
        double hi = lo[a] + vs;
        double dmin = fmax(0.0, fmax(lo[a] - p[a], p[a] - hi));
        double dmax = fmax(fabs(p[a] - lo[a]), fabs(p[a] - hi));
        mn2 += dmin * dmin;
        mx2 += dmax * dmax;
    }
}

__global__ void k_cand(int mode, const int* __restrict__ rmcell, long long VR, GridGeom g,
                       const unsigned long long* __restrict__ bitmap, const unsigned* __restrict__ rank,
                       const double* __restrict__ pts, double* __restrict__ ub2_io, int* __restrict__ rend_io,
                       unsigned* __restrict__ cnt, const unsigned* __restrict__ cand_off, int* __restrict__ cand,
                       unsigned* __restrict__ cursor) {
    const int lane = threadIdx.x & 63;
    const long long r = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= VR) return;
    const int cx = rmcell[r * 3], cy = rmcell[r * 3 + 1], cz = rmcell[r * 3 + 2];
    // the cell box, grown a hair: queries are assigned to cells by a rounded division
    const double pad = 1e-9;
    const double lo[3] = {g.ox + cx * g.vs - pad, g.oy + cy * g.vs - pad, g.oz + cz * g.vs - pad};
    const double vs = g.vs + 2 * pad;
    if (mode == 0) {
        double ub2 = 1e300;
        const int rmax = max(g.nx, max(g.ny, g.nz));
        int rr = 1;
        for (; rr <= rmax; ++rr) {
            ring_visit(g, bitmap, rank, cx, cy, cz, rr, lane, [&](int q) {
                double mn2, mx2;
                box_dists(pts + (size_t)q * 3, lo, vs, mn2, mx2);
                ub2 = mx2 < ub2 ? mx2 : ub2;
            });
            ub2 = wave_min_f64(ub2);
            double reach = rr * g.vs - 1e-6;          // every unexplored point is at least this far from the cell
            if (ub2 < 1e299 && reach > 0.0 && reach * reach > ub2) break;
        }
        if (rr > rmax) rr = rmax;
        const double lim = ub2 * (1.0 + 1e-9) + 1e-12;
        int n = 0;
        for (int q = 1; q <= rr; ++q)
            ring_visit(g, bitmap, rank, cx, cy, cz, q, lane, [&](int k) {
                double mn2, mx2;
                box_dists(pts + (size_t)k * 3, lo, vs, mn2, mx2);
                n += mn2 <= lim ? 1 : 0;
            });
        n = wave_sum_i32(n);
        if (lane == 0) {
            ub2_io[r] = ub2;
            rend_io[r] = rr;
            cnt[r] = (unsigned)n;
        }
    } else {
        const double lim = ub2_io[r] * (1.0 + 1e-9) + 1e-12;
        const int rr = rend_io[r];
        const unsigned base = cand_off[r];
        for (int q = 1; q <= rr; ++q)
            ring_visit(g, bitmap, rank, cx, cy, cz, q, lane, [&](int k) {
                double mn2, mx2;
                box_dists(pts + (size_t)k * 3, lo, vs, mn2, mx2);
                if (mn2 <= lim) cand[base + atomicAdd(&cursor[r], 1u)] = k;
            });
    }
}

// ------------------------------------------------------------------------------------------ host driver
void hmsg_build_map(hmsg_ctx* h) {
    const hmsg_config& c = h->cfg;
    hipStream_t s = h->stream;
    HMSG_REQUIRE(h->n_frames > 0, HMSG_ERR_INVALID, "hmsg_finalize_map: no frames added");
    // pcd_denoise_dbscan(eps=0.01, min_points=100) (graph.py:352): with one point per voxel, at most 27
    // points lie within eps <= voxel_size of any point, so with min_points > 27 there is no core point,
    // no cluster, and the wrapper returns its input (graph_utils.py:853-880).
    HMSG_REQUIRE(c.voxel_size >= 0.01, HMSG_ERR_UNSUPPORTED,
                 "voxel_size < 0.01: the DBSCAN(0.01,100) of graph.py:352 is no longer a provable no-op");
    const int H = c.height, W = c.width, F = h->n_frames;
    const float scale = (float)c.depth_scale;
    const size_t total = (size_t)H * W * F;
    const unsigned nblk = (unsigned)std::min<size_t>(cdiv(total, 256), 256 * 8);

    DevBuf<unsigned long long> d_b;
    d_b.alloc(8);
    unsigned long long init[8] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    HIP_TRY(hipMemcpyAsync(d_b.p, init, sizeof(init), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_bounds, dim3(nblk), dim3(256), 0, s, (const unsigned short*)h->depth.p,
                       (const double*)h->pose.p, h->cam, scale, H, W, F, d_b.p, d_b.p + 6);
    HMSG_CHECK_LAUNCH();
    unsigned long long hb[8];
    HIP_TRY(hipMemcpyAsync(hb, d_b.p, sizeof(hb), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HMSG_REQUIRE(hb[6] > 0, HMSG_ERR_INVALID, "no valid depth pixel in any frame");
    double mn[3], mx[3];
    for (int a = 0; a < 3; ++a) {
        mn[a] = dec_f64(hb[a]);
        mx[a] = dec_f64(hb[3 + a]);
    }
    GridGeom g;
    g.vs = c.voxel_size;
    g.ox = mn[0] - g.vs * 0.5;
    g.oy = mn[1] - g.vs * 0.5;
    g.oz = mn[2] - g.vs * 0.5;
    g.nx = (int)floor((mx[0] - g.ox) / g.vs) + 2;
    g.ny = (int)floor((mx[1] - g.oy) / g.vs) + 2;
    g.nz = (int)floor((mx[2] - g.oz) / g.vs) + 2;
    g.nzp = (g.nz + 63) / 64 * 64;
    g.nwords = (long long)g.nx * g.ny * (g.nzp / 64);
    HMSG_REQUIRE(g.nwords < (1ll << 31), HMSG_ERR_UNSUPPORTED, "scene bounding box too large for the dense occupancy bitmap");
    h->grid = g;

    h->bitmap.alloc((size_t)g.nwords);
    h->rank.alloc((size_t)g.nwords);
    h->bitmap.zero(s);
    hipLaunchKernelGGL(k_mark, dim3(nblk), dim3(256), 0, s, (const unsigned short*)h->depth.p, (const double*)h->pose.p,
                       h->cam, scale, H, W, F, g, h->bitmap.p);
    HMSG_CHECK_LAUNCH();
    unsigned long long V0 = hmsg_bitmap_rank(h->bitmap.p, h->rank.p, (size_t)g.nwords, s, h->scan_tmp);
    h->V0 = (long long)V0;

    DevBuf<unsigned long long> srgb;
    DevBuf<unsigned> sn, slots, chunk_runs;
    srgb.alloc(V0 * 3);
    sn.alloc(V0);
    srgb.zero(s);
    sn.zero(s);
    slots.alloc(total);
    const unsigned nchunks = cdiv(total, RUN_CHUNK);
    chunk_runs.alloc((size_t)nchunks + 1);
    VoxAcc acc{srgb.p, srgb.p + V0, srgb.p + 2 * V0, sn.p};
    {
        ProfScope ps(h->prof, s, "k_slots", (double)total * 9.0 + (double)V0 * 28.0);
        hipLaunchKernelGGL(k_slots, dim3(nchunks), dim3(256), 0, s, (const unsigned short*)h->depth.p,
                           (const unsigned char*)h->rgb.p, (const double*)h->pose.p, h->cam, scale, H, W, F, g,
                           (const unsigned long long*)h->bitmap.p, (const unsigned*)h->rank.p, acc, slots.p, chunk_runs.p);
    }
    HMSG_CHECK_LAUNCH();
    unsigned long long nruns = 0;
    hmsg_scan_u32(chunk_runs.p, chunk_runs.p, (size_t)nchunks, s, h->scan_tmp, &nruns);
    SortBufs sb;
    sb.keys.alloc((size_t)std::max<unsigned long long>(nruns, 1));
    sb.vals.alloc((size_t)std::max<unsigned long long>(nruns, 1));
    HMSG_REQUIRE(W < 4096 && H < 4096, HMSG_ERR_UNSUPPORTED, "image sides must be below 4096 pixels");
    hipLaunchKernelGGL(k_emit_runs, dim3(nchunks), dim3(256), 0, s, (const unsigned*)slots.p, total, W, (size_t)H * W,
                       (const unsigned*)chunk_runs.p, sb.keys.p, sb.vals.p);
    HMSG_CHECK_LAUNCH();
    {
        ProfScope ps(h->prof, s, "sort_runs", (double)nruns * 12.0 * 4.0);
        hmsg_sort_pairs(sb, (size_t)nruns, bits_for(V0), s);
    }
    DevBuf<unsigned> run_off;
    run_off.alloc(V0 + 1);
    hmsg_sort_segment_starts(sb.res_keys, (size_t)nruns, run_off.p, (unsigned)V0, s);
    DevBuf<int> cell;
    cell.alloc(V0 * 3);
    hipLaunchKernelGGL(k_slot_cells, dim3(cdiv((size_t)g.nwords, 256)), dim3(256), 0, s,
                       (const unsigned long long*)h->bitmap.p, (const unsigned*)h->rank.p, g, cell.p);
    HMSG_CHECK_LAUNCH();
    DevBuf<double> pts0, cols0;
    pts0.alloc(V0 * 3);
    cols0.alloc(V0 * 3);
    {
        // HMSG_DEBUG_ACCUM_WAVES=1: every lane carries the sums (the form until round 5)
        static const bool lds_sum = getenv("HMSG_DEBUG_ACCUM_WAVES") == nullptr;
        ProfScope ps(h->prof, s, "k_accum_ordered", (double)total * 2.0 + (double)nruns * 8.0 + (double)V0 * 32.0);
        hipLaunchKernelGGL(lds_sum ? k_accum_ordered<true> : k_accum_ordered<false>, dim3(cdiv(V0 * 64, 256)), dim3(256), 0, s,
                           (const unsigned short*)h->depth.p, (const double*)h->pose.p, h->cam, scale, W, (size_t)H * W, (long long)V0,
                           (const unsigned*)run_off.p, (const unsigned long long*)sb.res_vals, (const unsigned*)sn.p, pts0.p);
        HMSG_CHECK_LAUNCH();
        if (getenv("HMSG_DEBUG_ACCUM_STATS")) {     // development aid: how the pixels are spread over the voxels
            std::vector<unsigned> hc((size_t)V0);
            HIP_TRY(hipMemcpyAsync(hc.data(), sn.p, (size_t)V0 * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            std::sort(hc.begin(), hc.end());
            double tot = 0;
            for (unsigned c : hc) tot += c;
            fprintf(stderr, "[hmsg map] voxels %lld  runs %llu  pixels %.0f  max %u  median %u  p99 %u\n", (long long)V0, nruns, tot, hc.back(),
                    hc[hc.size() / 2], hc[(size_t)(hc.size() * 0.99)]);
            for (unsigned th : {512u, 2048u, 8192u, 32768u, 131072u}) {
                double px = 0;
                long long nv = 0;
                for (unsigned c : hc)
                    if (c > th) px += c, ++nv;
                fprintf(stderr, "[hmsg map]   more than %6u pixels: %7lld voxels holding %.3f of the pixels\n", th, nv, px / tot);
            }
        }
    }
    hipLaunchKernelGGL(k_finalize, dim3(cdiv(V0, 256)), dim3(256), 0, s, acc, (long long)V0, cols0.p);
    HMSG_CHECK_LAUNCH();

    // remove_radius_outlier (graph.py:355-358)
    DevBuf<unsigned> keep, newidx;
    keep.alloc(V0);
    newidx.alloc(V0);
    hipLaunchKernelGGL(k_outlier, dim3(cdiv(V0 * 64, 256)), dim3(256), 0, s, (const double*)pts0.p, (const int*)cell.p,
                       (long long)V0, g, (const unsigned long long*)h->bitmap.p, (const unsigned*)h->rank.p,
                       c.outlier_radius, c.outlier_nb_points, keep.p);
    HMSG_CHECK_LAUNCH();
    unsigned long long V = 0;
    hmsg_scan_u32(keep.p, newidx.p, V0, s, h->scan_tmp, &V);
    h->V = (long long)V;
    h->pts.alloc(V * 3);
    h->cols.alloc(V * 3);
    hipLaunchKernelGGL(k_compact, dim3(cdiv(V0, 256)), dim3(256), 0, s, (const unsigned*)keep.p, (const unsigned*)newidx.p,
                       (long long)V0, (const double*)pts0.p, (const double*)cols0.p, (const int*)cell.p, g, h->pts.p,
                       h->cols.p, h->bitmap.p);
    HMSG_CHECK_LAUNCH();
    unsigned long long V2 = hmsg_bitmap_rank(h->bitmap.p, h->rank.p, (size_t)g.nwords, s, h->scan_tmp);
    HMSG_REQUIRE(V2 == V, HMSG_ERR_INVALID, "internal: filtered bitmap population mismatch");
    // NN candidate lists for the cells of the deleted voxels
    const long long VR = (long long)V0 - (long long)V;
    h->have_cand = false;
    if (VR > 0 && V > 0) {
        h->bitmap_rm.alloc((size_t)g.nwords);
        h->rank_rm.alloc((size_t)g.nwords);
        h->bitmap_rm.zero(s);
        DevBuf<int> rmcell, rend;
        DevBuf<double> ub2;
        DevBuf<unsigned> ccnt, cursor;
        rmcell.alloc((size_t)VR * 3);
        rend.alloc((size_t)VR);
        ub2.alloc((size_t)VR);
        ccnt.alloc((size_t)VR + 1);
        cursor.alloc((size_t)VR);
        ccnt.zero(s);
        cursor.zero(s);
        hipLaunchKernelGGL(k_rm_cells, dim3(cdiv(V0, 256)), dim3(256), 0, s, (const unsigned*)keep.p, (const unsigned*)newidx.p,
                           (long long)V0, (const int*)cell.p, g, rmcell.p, h->bitmap_rm.p);
        HMSG_CHECK_LAUNCH();
        unsigned long long nrm = hmsg_bitmap_rank(h->bitmap_rm.p, h->rank_rm.p, (size_t)g.nwords, s, h->scan_tmp);
        HMSG_REQUIRE((long long)nrm == VR, HMSG_ERR_INVALID, "internal: removed-cell bitmap population mismatch");
        h->cand_off.alloc((size_t)VR + 1);
        {
            ProfScope ps(h->prof, s, "k_cand");
            hipLaunchKernelGGL(k_cand, dim3(cdiv((size_t)VR * 64, 256)), dim3(256), 0, s, 0, (const int*)rmcell.p, VR, g,
                               (const unsigned long long*)h->bitmap.p, (const unsigned*)h->rank.p, (const double*)h->pts.p, ub2.p,
                               rend.p, ccnt.p, (const unsigned*)nullptr, (int*)nullptr, (unsigned*)nullptr);
        }
        HMSG_CHECK_LAUNCH();
        unsigned long long ncand = 0;
        hmsg_scan_u32(ccnt.p, h->cand_off.p, (size_t)VR + 1, s, h->scan_tmp, &ncand);
        h->cand.alloc((size_t)std::max<unsigned long long>(ncand, 1));
        hipLaunchKernelGGL(k_cand, dim3(cdiv((size_t)VR * 64, 256)), dim3(256), 0, s, 1, (const int*)rmcell.p, VR, g,
                           (const unsigned long long*)h->bitmap.p, (const unsigned*)h->rank.p, (const double*)h->pts.p, ub2.p,
                           rend.p, ccnt.p, (const unsigned*)h->cand_off.p, h->cand.p, cursor.p);
        HMSG_CHECK_LAUNCH();
        h->have_cand = true;
    }
    HIP_TRY(hipStreamSynchronize(s));
    hmsg_kd_start(h);
    h->map_ready = true;
}
