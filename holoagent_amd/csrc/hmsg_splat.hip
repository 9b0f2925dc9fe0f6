// Row N3: LiDAR keyframe clouds -> occlusion-aware uint16 depth images, the step right before the HMSG build
// (reference: nav_agent/humble_localization_nav2/lio_mapping_loc/scripts/generate_depth.py
//   process_frame :612-659 = voxel_down_sample(0.02) :626-629 -> project_points :366-396 ->
//   whether_occluded_deoccfast :125-205 -> generate_occ_depth :399-474;  restated in oracle/lidar_depth_oracle.py).
//
// A batch of frames (each with its own local map, CSR over one point array) goes through ONE sequence of launches:
//   voxel_down_sample      the segmented, order-faithful one of hmsg_cloudops.hip (float64 sums in input order)
//   k_sp_project           camera frame, pixel coordinates, culls; key = frame * HW + ROUNDED pixel
//   sort + k_sp_zwalk      the reference's z-buffer compares against a float32 buffer and stores fb / z of the point
//                          that last lowered it, so the outcome depends on the point order: the points of a pixel
//                          are replayed in input order (stable radix sort by pixel, one lane per pixel run)
//   k_sp_dilate_h / _v     cv2.dilate(rect k, iterations 4) = one max filter over [-4a, 4 (k - 1 - a)], a = k / 2;
//                          the vertical pass also does the int16 cast and seeds the union-find
//   k_sp_cc_init / _union / _size / _apply   cv2.filterSpeckles(0, 1000, 1): connected components of the 4-neighbour
//                          graph "both non-zero and |a - b| <= 1" (symmetric, so independent of scan order): lock-free
//                          union-find over pixels seeded with per-wave row runs, only the links that transitivity
//                          does not already give are united; components of <= 1000 pixels zeroed
//   k_sp_flags             per point: occluded if the pixel is empty or |fb / z - disparity| >= 3; the visible ones
//                          race for their TRUNCATED pixel with atomicMax(point index) = numpy's last-writer-wins
//   k_sp_depth             depth = uint16(float32(z * depth_factor)) of the winner
// All of it is HBM / latency bound integer and compare work (no MFMA); per frame the traffic is ~50 B per point and
// ~40 B per pixel.
#include "hmsg_cloudops.h"

#include <cmath>

namespace {

constexpr double SP_FB = 20.0;            // generate_depth.py:147
constexpr float SP_ZINIT = 1000.0f;       // :146
constexpr unsigned SP_MAX_SPECKLE = 1000; // :172
constexpr int SP_MAX_DIFF = 1;            // :172
constexpr double SP_WINDOW = 3.0;         // :203

struct SpGeom {
    int W, H, F, mode;                    // mode 0: world points + poses, 1: (u, v, z) triples
    double fx, fy, cx, cy;
    long long N;
    unsigned HW;
};

__device__ __forceinline__ int sp_frame_of(const long long* __restrict__ off, int F, long long i) {
    int lo = 0, hi = F - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// per-frame counters, one atomic per (wave, frame): lanes of a wave almost always share the frame
__device__ __forceinline__ void sp_count(unsigned* __restrict__ ctr, int f, bool pred) {
    unsigned long long todo = __ballot(pred);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll(todo) - 1;
        const int lf = __shfl(f, leader);
        const unsigned long long same = __ballot(pred && f == lf);
        if (lane == leader) atomicAdd(&ctr[(size_t)lf * 4], (unsigned)__popcll(same));
        todo &= ~same;
    }
}

// state: 0 = candidate (has a rounded pixel), 1 = occluded by the rounding-bounds rule, 2 = culled by the projection
__global__ void __launch_bounds__(256) k_sp_project(const double* __restrict__ pts, const long long* __restrict__ off, SpGeom g,
                                                    const double* __restrict__ poses, double* __restrict__ uvz,
                                                    unsigned char* __restrict__ state, unsigned* __restrict__ pix,
                                                    unsigned* __restrict__ keys, unsigned long long* __restrict__ vals,
                                                    unsigned* __restrict__ stats) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < g.N;
    if (!live) i = g.N - 1;
    const int f = sp_frame_of(off, g.F, i);
    double u, v, z;
    bool ok = true;
    if (g.mode == 0) {
        const double* T = poses + (size_t)f * 12;
        const double x = pts[i * 3], y = pts[i * 3 + 1], w = pts[i * 3 + 2];
        // np.dot(rotation, points.T) + translation, written out left to right without FMA (project_points :375)
        const double xc = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(T[0], x), __dmul_rn(T[1], y)), __dmul_rn(T[2], w)), T[9]);
        const double yc = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(T[3], x), __dmul_rn(T[4], y)), __dmul_rn(T[5], w)), T[10]);
        z = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(T[6], x), __dmul_rn(T[7], y)), __dmul_rn(T[8], w)), T[11]);
        ok = z > 0.0;                                         // :378 (NaN fails too)
        const double xn = __ddiv_rn(xc, z), yn = __ddiv_rn(yc, z), on = __ddiv_rn(z, z);
        // np.dot(intrinsics, .) with the zero entries of K kept in the sum (:382)
        u = __dadd_rn(__dadd_rn(__dmul_rn(g.fx, xn), __dmul_rn(0.0, yn)), __dmul_rn(g.cx, on));
        v = __dadd_rn(__dadd_rn(__dmul_rn(0.0, xn), __dmul_rn(g.fy, yn)), __dmul_rn(g.cy, on));
        ok = ok && u >= 0.0 && u < (double)g.W && v >= 0.0 && v < (double)g.H;   // :386
    } else {
        u = pts[i * 3];
        v = pts[i * 3 + 1];
        z = pts[i * 3 + 2];
        ok = (u == u) && (v == v) && (z == z);                // NaN rows are dropped
    }
    ok = ok && live;
    // whether_occluded_deoccfast :150-152: points whose ROUNDED pixel is outside, or z <= 0, are "occluded"
    const double ur = __dadd_rn(u, 0.5), vr = __dadd_rn(v, 0.5);
    const bool inside = ok && !(z <= 0.0 || ur < 0.0 || ur >= (double)g.W || vr < 0.0 || vr >= (double)g.H);
    unsigned key = (unsigned)g.F * g.HW;                      // sentinel: sorts behind every pixel
    if (inside) key = (unsigned)f * g.HW + (unsigned)(long long)vr * (unsigned)g.W + (unsigned)(long long)ur;
    if (live) {
        uvz[i * 3] = u;
        uvz[i * 3 + 1] = v;
        uvz[i * 3 + 2] = z;
        state[i] = !ok ? 2 : (inside ? 0 : 1);
        pix[i] = key;
        keys[i] = key;
        vals[i] = (unsigned long long)i;
    }
    sp_count(stats + 1, f, ok);                               // projected into the image
}

// one lane per run of equal keys (= one pixel of one frame), points in input order
__global__ void k_sp_zwalk(const unsigned* __restrict__ keys, const unsigned long long* __restrict__ vals, long long n,
                           unsigned sentinel, const double* __restrict__ uvz, float* __restrict__ inv_depth) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const unsigned key = keys[r];
    if (key == sentinel || (r > 0 && keys[r - 1] == key)) return;
    float m = SP_ZINIT, inv = 0.0f;
    for (long long q = r; q < n && keys[q] == key; ++q) {
        const double z = uvz[vals[q] * 3 + 2];
        if ((double)m > z) {                                  // :159 float32 buffer against the float64 depth
            m = (float)z;
            inv = (float)__ddiv_rn(SP_FB, z);
        }
    }
    inv_depth[key] = inv;
}

__global__ void k_sp_dilate_h(const float* __restrict__ src, float* __restrict__ dst, int W, long long rows, int lo, int hi) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= rows * W) return;
    const int x = (int)(p % W);
    const float* row = src + (p - x);
    float m = row[x];
    for (int q = max(0, x - lo); q <= min(W - 1, x + hi); ++q) m = fmaxf(m, row[q]);
    dst[p] = m;
}
// vertical pass + np.int16 cast
__global__ void k_sp_dilate_v(const float* __restrict__ src, int W, int H, long long npix, int lo, int hi, short* __restrict__ s16) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const long long hw = (long long)W * H;
    const long long inf = p % hw;
    const int y = (int)(inf / W);
    const float* col = src + (p - (long long)y * W);
    float m = col[(long long)y * W];
    for (int q = max(0, y - lo); q <= min(H - 1, y + hi); ++q) m = fmaxf(m, col[(long long)q * W]);
    // float32 -> int16 like the C cast numpy performs (through a wider integer, wrapping)
    const short d = (short)(long long)truncf(m);
    s16[p] = d;
}

__device__ __forceinline__ int sp_find(int* parent, int x) {
    for (;;) {
        const int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        const int gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p;
    }
}
__device__ __forceinline__ void sp_union(int* parent, int a, int b) {
    for (;;) {
        a = sp_find(parent, a);
        b = sp_find(parent, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        if (atomicCAS(&parent[a], a, b) == a) return;
    }
}
__device__ __forceinline__ bool sp_linked(int a, int b) { return a != 0 && b != 0 && abs(a - b) <= SP_MAX_DIFF; }

// Seeds: every live pixel starts under the first pixel of its horizontal run INSIDE ITS WAVE (64 consecutive pixels;
// a run never crosses a row start because "linked to the left" is false at x = 0), so the horizontal links of a flat
// region cost one union per wave instead of one per pixel.
__global__ void __launch_bounds__(256) k_sp_cc_init(const short* __restrict__ s16, int W, long long npix, int* __restrict__ parent) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < npix;
    const int d = live ? s16[p] : 0;
    const bool cl = d != 0 && (p % W) > 0 && sp_linked(d, s16[p - 1]);
    const int lane = threadIdx.x & 63;
    const unsigned long long below = __ballot(!cl) & ((2ull << lane) - 1ull);      // run starts at or before this lane
    const int start = below ? 63 - __clzll(below) : 0;
    if (live) parent[p] = d != 0 ? (int)(p - (lane - start)) : -1;
}
// Links: wave-boundary horizontal links, and the vertical links that are not implied by the square to their left
// ((x-1,y)~(x,y), (x-1,y)~(x-1,y+1) and (x-1,y+1)~(x,y+1) already connect (x,y) with (x,y+1)).
__global__ void k_sp_cc_union(const short* __restrict__ s16, int W, int H, long long npix, int* __restrict__ parent) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const int d = s16[p];
    if (d == 0) return;
    const long long inf = p % ((long long)W * H);
    const int x = (int)(inf % W), y = (int)(inf / W);
    const int dl = x > 0 ? s16[p - 1] : 0;
    const bool cl = sp_linked(d, dl);
    if (cl && (p & 63) == 0) sp_union(parent, (int)p, (int)p - 1);
    if (y + 1 < H) {
        const int dd = s16[p + W];
        if (sp_linked(d, dd)) {
            const int dld = x > 0 ? s16[p + W - 1] : 0;
            const bool implied = cl && sp_linked(dl, dld) && sp_linked(dld, dd);
            if (!implied) sp_union(parent, (int)p, (int)p + W);
        }
    }
}
__global__ void k_sp_cc_size(int* __restrict__ parent, long long npix, unsigned* __restrict__ size) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix || parent[p] < 0) return;
    atomicAdd(&size[sp_find(parent, (int)p)], 1u);
}
// (the root is looked up again: a path-halving store of another lane may have moved parent[p] to ANY ancestor)
__global__ void k_sp_cc_apply(int* __restrict__ parent, const unsigned* __restrict__ size, long long npix, short* __restrict__ s16) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix || parent[p] < 0) return;
    if (size[sp_find(parent, (int)p)] <= SP_MAX_SPECKLE) s16[p] = 0;
}

__global__ void __launch_bounds__(256) k_sp_flags(const double* __restrict__ uvz, const long long* __restrict__ off, SpGeom g,
                                                  const unsigned* __restrict__ pix, const short* __restrict__ s16,
                                                  unsigned char* __restrict__ state, unsigned* __restrict__ last,
                                                  unsigned* __restrict__ stats) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < g.N;
    if (!live) i = g.N - 1;
    const int f = sp_frame_of(off, g.F, i);
    bool visible = false;
    if (live && state[i] == 0) {
        const double z = uvz[i * 3 + 2];
        const short noise = s16[pix[i]];
        const bool occ = noise == 0 || fabs(__dsub_rn(__ddiv_rn(SP_FB, z), (double)noise)) >= SP_WINDOW;   // :193-203
        if (occ) state[i] = 1;
        else {
            visible = true;
            // generate_occ_depth :435-441: TRUNCATED pixel, then the bounds test
            const long long tx = (long long)uvz[i * 3], ty = (long long)uvz[i * 3 + 1];
            if (tx >= 0 && tx < g.W && ty >= 0 && ty < g.H)
                atomicMax(&last[(size_t)f * g.HW + (size_t)ty * g.W + (size_t)tx], (unsigned)(i + 1));
        }
    }
    sp_count(stats + 2, f, visible);
}

__global__ void __launch_bounds__(256) k_sp_depth(const unsigned* __restrict__ last, const double* __restrict__ uvz, SpGeom g,
                                                  double factor, long long npix, unsigned short* __restrict__ depth,
                                                  unsigned* __restrict__ stats) {
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < npix;
    if (!live) p = npix - 1;
    const unsigned l = last[p];
    unsigned short d = 0;
    if (l) {
        const float v = (float)__dmul_rn(uvz[(size_t)(l - 1) * 3 + 2], factor);     // :447 float32 image
        d = (unsigned short)((long long)truncf(v) & 0xFFFF);                         // :474 astype(uint16)
    }
    if (live) depth[p] = d;
    sp_count(stats + 3, (int)(p / g.HW), live && l != 0);
}

}  // namespace

extern "C" int hmsg_lidar_depth(int32_t device_id, const hmsg_depth_params* prm, int32_t n_frames, const double* points,
                                const int64_t* cloud_off, const double* poses, uint16_t* depth_out, uint8_t* state_out,
                                int64_t* stats_out, double* device_ms) {
    if (!prm || n_frames < 0 || !cloud_off || !depth_out) return HMSG_ERR_INVALID;
    const int W = prm->width, H = prm->height;
    if (W <= 0 || H <= 0 || prm->image_scale <= 0 || (long long)W * H > (1ll << 26)) return HMSG_ERR_INVALID;
    if (state_out && prm->voxel_size > 0) return HMSG_ERR_INVALID;      // point indices change under down-sampling
    if (n_frames == 0) return HMSG_OK;
    for (int f = 0; f < n_frames; ++f)
        if (cloud_off[f + 1] < cloud_off[f]) return HMSG_ERR_INVALID;
    if (cloud_off[n_frames] > cloud_off[0] && !points) return HMSG_ERR_INVALID;
    const unsigned HW = (unsigned)(W * H);
    const int ksize = std::max(1, 4 / prm->image_scale), a = ksize / 2;
    const int lo = 4 * a, hi = 4 * (ksize - 1 - a);
    hipStream_t s = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int rc = HMSG_OK;
    try {
        HIP_TRY(hipSetDevice(device_id));
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&ev0));
        HIP_TRY(hipEventCreate(&ev1));
        double total_ms = 0.0;
        {
            CloudOps ops;
            ops.s = s;
            SortBufs sort;
            DevBuf<double> d_src, d_vds, d_uvz, d_pose;
            DevBuf<long long> d_off;
            DevBuf<unsigned char> d_state;
            DevBuf<unsigned> d_pix, d_size, d_last, d_stats;
            DevBuf<float> d_inv, d_tmp;
            DevBuf<short> d_s16;
            DevBuf<int> d_parent;
            DevBuf<unsigned short> d_depth;
            // frames per batch: keys are frame * HW + pixel (u32, one sentinel), point indices u32, and the working
            // set is kept to a few GB
            const long long max_pts = 1ll << 27;
            int f0 = 0;
            while (f0 < n_frames) {
                int f1 = f0;
                long long npts = 0;
                while (f1 < n_frames && (f1 - f0 + 1) * (long long)HW < (1ll << 28) &&
                       (f1 == f0 || npts + (cloud_off[f1 + 1] - cloud_off[f1]) <= max_pts)) {
                    npts += cloud_off[f1 + 1] - cloud_off[f1];
                    ++f1;
                }
                HMSG_REQUIRE(npts < (1ll << 31), HMSG_ERR_UNSUPPORTED, "hmsg_lidar_depth: one frame holds more than 2^31 points");
                int F = f1 - f0;
                std::vector<long long> off(F + 1);
                for (int f = 0; f <= F; ++f) off[f] = cloud_off[f0 + f] - cloud_off[f0];
                d_src.ensure((size_t)std::max<long long>(npts, 1) * 3);
                if (npts)
                    HIP_TRY(hipMemcpyAsync(d_src.p, points + cloud_off[f0] * 3, (size_t)npts * 24, hipMemcpyHostToDevice, s));
                HIP_TRY(hipStreamSynchronize(s));
                HIP_TRY(hipEventRecord(ev0, s));
                const double* pts = d_src.p;
                long long N = npts;
                std::vector<long long> n_vds(F);
                for (int f = 0; f < F; ++f) n_vds[f] = off[f + 1] - off[f];
                if (prm->voxel_size > 0 && npts > 0) {
                    std::vector<SegDesc> segs(F);
                    for (int f = 0; f < F; ++f) {
                        segs[f].pt_base = off[f];
                        segs[f].n = (int)(off[f + 1] - off[f]);
                    }
                    ops.bounds(pts, segs);
                    // the down-sampler marks occupied voxels in a dense bitmap over each cloud's bounding box: keep
                    // the batch's bitmaps under 8 GB (a batch shrinks to the frames that fit, at least one)
                    double words = 0;
                    int keep = 0;
                    for (; keep < F; ++keep) {
                        double c = 1;
                        for (int ax = 0; ax < 3; ++ax) c *= std::floor((segs[keep].mx[ax] - segs[keep].mn[ax]) / prm->voxel_size) + 3;
                        if (keep > 0 && words + c / 64 > (double)(1ll << 30)) break;
                        words += c / 64 + 1;
                    }
                    if (keep < F) {
                        F = keep;
                        f1 = f0 + F;
                        segs.resize(F);
                        off.resize(F + 1);
                        n_vds.resize(F);
                        npts = off[F];
                    }
                    d_vds.ensure((size_t)npts * 3);
                    std::vector<int> out_n;
                    N = ops.voxel_down_sample(pts, segs, prm->voxel_size, d_vds.p, out_n);
                    pts = d_vds.p;
                    off[0] = 0;
                    for (int f = 0; f < F; ++f) {
                        n_vds[f] = out_n[f];
                        off[f + 1] = off[f] + out_n[f];
                    }
                }
                const long long npix = (long long)F * HW;
                d_off.ensure((size_t)F + 1);
                HIP_TRY(hipMemcpyAsync(d_off.p, off.data(), ((size_t)F + 1) * 8, hipMemcpyHostToDevice, s));
                if (poses) {
                    d_pose.ensure((size_t)F * 12);
                    HIP_TRY(hipMemcpyAsync(d_pose.p, poses + (size_t)f0 * 12, (size_t)F * 96, hipMemcpyHostToDevice, s));
                }
                d_stats.ensure((size_t)F * 4);
                HIP_TRY(hipMemsetAsync(d_stats.p, 0, (size_t)F * 16, s));
                d_inv.ensure((size_t)npix);
                d_tmp.ensure((size_t)npix);
                d_s16.ensure((size_t)npix);
                d_parent.ensure((size_t)npix);
                d_size.ensure((size_t)npix);
                d_last.ensure((size_t)npix);
                d_depth.ensure((size_t)npix);
                HIP_TRY(hipMemsetAsync(d_inv.p, 0, (size_t)npix * 4, s));
                HIP_TRY(hipMemsetAsync(d_size.p, 0, (size_t)npix * 4, s));
                HIP_TRY(hipMemsetAsync(d_last.p, 0, (size_t)npix * 4, s));
                SpGeom g{W, H, F, poses ? 0 : 1, prm->fx, prm->fy, prm->cx, prm->cy, N, HW};
                const unsigned pblk = cdiv((size_t)npix, 256);
                if (N > 0) {
                    d_uvz.ensure((size_t)N * 3);
                    d_state.ensure((size_t)N);
                    d_pix.ensure((size_t)N);
                    sort.keys.ensure((size_t)N);
                    sort.vals.ensure((size_t)N);
                    const unsigned nblk = cdiv((size_t)N, 256);
                    hipLaunchKernelGGL(k_sp_project, dim3(nblk), dim3(256), 0, s, pts, (const long long*)d_off.p, g,
                                       (const double*)d_pose.p, d_uvz.p, d_state.p, d_pix.p, sort.keys.p, sort.vals.p, d_stats.p);
                    HMSG_CHECK_LAUNCH();
                    hmsg_sort_pairs(sort, (size_t)N, bits_for((unsigned long long)npix + 1), s);
                    hipLaunchKernelGGL(k_sp_zwalk, dim3(nblk), dim3(256), 0, s, (const unsigned*)sort.res_keys,
                                       (const unsigned long long*)sort.res_vals, N, (unsigned)npix, (const double*)d_uvz.p, d_inv.p);
                    HMSG_CHECK_LAUNCH();
                }
                hipLaunchKernelGGL(k_sp_dilate_h, dim3(pblk), dim3(256), 0, s, (const float*)d_inv.p, d_tmp.p, W, (long long)F * H, lo, hi);
                hipLaunchKernelGGL(k_sp_dilate_v, dim3(pblk), dim3(256), 0, s, (const float*)d_tmp.p, W, H, npix, lo, hi, d_s16.p);
                hipLaunchKernelGGL(k_sp_cc_init, dim3(pblk), dim3(256), 0, s, (const short*)d_s16.p, W, npix, d_parent.p);
                hipLaunchKernelGGL(k_sp_cc_union, dim3(pblk), dim3(256), 0, s, (const short*)d_s16.p, W, H, npix, d_parent.p);
                hipLaunchKernelGGL(k_sp_cc_size, dim3(pblk), dim3(256), 0, s, d_parent.p, npix, d_size.p);
                hipLaunchKernelGGL(k_sp_cc_apply, dim3(pblk), dim3(256), 0, s, d_parent.p, (const unsigned*)d_size.p, npix, d_s16.p);
                HMSG_CHECK_LAUNCH();
                if (N > 0) {
                    hipLaunchKernelGGL(k_sp_flags, dim3(cdiv((size_t)N, 256)), dim3(256), 0, s, (const double*)d_uvz.p,
                                       (const long long*)d_off.p, g, (const unsigned*)d_pix.p, (const short*)d_s16.p, d_state.p, d_last.p,
                                       d_stats.p);
                    HMSG_CHECK_LAUNCH();
                }
                hipLaunchKernelGGL(k_sp_depth, dim3(pblk), dim3(256), 0, s, (const unsigned*)d_last.p, (const double*)d_uvz.p, g,
                                   prm->depth_factor, npix, d_depth.p, d_stats.p);
                HMSG_CHECK_LAUNCH();
                HIP_TRY(hipEventRecord(ev1, s));
                HIP_TRY(hipMemcpyAsync(depth_out + (size_t)f0 * HW, d_depth.p, (size_t)npix * 2, hipMemcpyDeviceToHost, s));
                if (state_out && N > 0)
                    HIP_TRY(hipMemcpyAsync(state_out + cloud_off[f0], d_state.p, (size_t)N, hipMemcpyDeviceToHost, s));
                std::vector<unsigned> st((size_t)F * 4);
                HIP_TRY(hipMemcpyAsync(st.data(), d_stats.p, (size_t)F * 16, hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
                total_ms += ms;
                if (stats_out)
                    for (int f = 0; f < F; ++f) {
                        stats_out[(size_t)(f0 + f) * 4 + 0] = n_vds[f];
                        for (int k = 1; k < 4; ++k) stats_out[(size_t)(f0 + f) * 4 + k] = st[(size_t)f * 4 + k];
                    }
                f0 = f1;
            }
        }
        if (device_ms) *device_ms = total_ms;
    } catch (const hmsg_error& e) {
        fprintf(stderr, "hmsg_lidar_depth: %s\n", e.msg.c_str());
        rc = e.code;
    }
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (s) (void)hipStreamDestroy(s);
    return rc;
}
