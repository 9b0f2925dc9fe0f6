// A3 + A4 + A5: per-frame feature fusion math, exact nearest-voxel snapping, last-writer-wins feature
// accumulation and the 3-D mask clouds.
//
// Reference: perception/models/sam_clip_feats_extractor.py:159-191 (fusion math),
// memory/hmsg/graph/graph.py:373-415 (loop B), memory/hmsg/dataloader/generic.py:140-190
// (create_3d_masks).
//
// MI355X design notes
//  * The reference materialises a per-pixel fp16 feature image (315 MB/frame at D=512).  A pixel's
//    feature depends only on WHICH masks cover it, so we keep a mask-membership bitset per pixel (NW 64-bit
//    words, NW = ceil(max_masks / 64) <= 4) and the M x D table F_p per frame, and rebuild the (fp16-rounded)
//    row only for the one pixel per voxel per frame that torch's duplicate-index `+=` actually keeps
//    (graph.py:410).  SAM returns a different number of masks for every frame: the per-frame count bounds the
//    softmax of sam_clip_feats_extractor.py:167-169 and the bitset.
//  * Frames are processed 64 at a time.  stamp[v][j] = 1 + the largest pixel index of frame j (of the
//    batch) whose nearest voxel is v: a wave then owns ONE voxel, ballots its 64 stamps and adds the
//    frames' contributions in frame order in registers -- the same float32 addition order as the
//    reference's sequential loop, and one read-modify-write of the voxel's row per 64 frames instead
//    of one per frame.
//  * Nearest voxel: ring expansion over the occupancy bitmap (z-columns are contiguous bits) with an
//    exact termination bound; no distance cap (graph.py:409, generic.py:181).
//  * 3-D masks (generic.py:181-188): the snapped map points of a mask, WITH their pixel multiplicity, go through
//    Open3D's voxel_down_sample = per voxel a sequential float64 sum in pixel order.  Runs of pixels with the same
//    (voxel, mask set) become records (mask-voxel slot, map voxel, length) in pixel order, a stable sort groups
//    them by slot, one lane per slot replays the additions (hmsg_sort.hip explains why the order matters).
#include "hmsg_common.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <cmath>

#define FB 64                       /* frames per stamp batch = wave width */
#define MFIX_SCALE 17592186044416.0  /* 2^44 fixed point for (pixel-count weighted) mask-cloud centroids */

// ------------------------------------------------------------------------------------------ K_bitset
// masks u8 [M][HW] of one frame -> bits u64 [HW][NW]; 16 pixels per thread, 16-byte loads.  Only the first
// nmask[f] masks of a frame are real (the rest of the M rows is padding of the hand-over layout).
__global__ void k_bitset(const unsigned char* __restrict__ masks, int M, size_t HW, int nfr, size_t mask_stride_frame,
                         const int* __restrict__ nmask, int NW, unsigned long long* __restrict__ bits) {
    const size_t chunks = (HW + 15) / 16;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= chunks * nfr) return;
    int f = (int)(t / chunks);
    size_t p0 = (t - (size_t)f * chunks) * 16;
    const unsigned char* mf = masks + (size_t)f * mask_stride_frame;
    const int nm = min(nmask[f], M);
    unsigned long long* o = bits + ((size_t)f * HW + p0) * NW;
    _Pragma("unroll") for (int w = 0; w < 4; ++w) if (w < NW) {
        unsigned long long b[16];
        for (int j = 0; j < 16; ++j) b[j] = 0ull;
        const int i0 = w * 64, i1 = min(nm, i0 + 64);
        if (p0 + 16 <= HW && (HW & 15) == 0) {
            for (int i = i0; i < i1; ++i) {
                uint4 v = *reinterpret_cast<const uint4*>(mf + (size_t)i * HW + p0);
                unsigned wd[4] = {v.x, v.y, v.z, v.w};
                for (int j = 0; j < 16; ++j) {
                    unsigned byte = (wd[j >> 2] >> ((j & 3) * 8)) & 0xffu;
                    b[j] |= (unsigned long long)(byte != 0) << (i - i0);
                }
            }
        } else {
            for (int i = i0; i < i1; ++i)
                for (int j = 0; j < 16 && p0 + j < HW; ++j) b[j] |= (unsigned long long)(mf[(size_t)i * HW + p0 + j] != 0) << (i - i0);
        }
        for (int j = 0; j < 16 && p0 + j < HW; ++j) o[(size_t)j * NW + w] = b[j];
    }
}

// ------------------------------------------------------------------------------------------ K_fp
// sam_clip_feats_extractor.py:159-175.  One 256-thread block per frame; a wave per mask row.  The softmax runs
// over the frame's OWN nm masks (:167-169); rows >= nm of the table are never referenced.
#define HMSG_MAX_MASKS 256
__global__ void k_fp(const float* __restrict__ Fg, const float* __restrict__ Fm, const float* __restrict__ Fc, int M, int D,
                     float wm, float wc, const int* __restrict__ nmask, int MS, float* __restrict__ Fp /*[nfr][MS][D]*/) {
    __shared__ float phi[HMSG_MAX_MASKS];
    __shared__ float wsm[HMSG_MAX_MASKS];
    const int f = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nm = min(nmask[f], M);
    const float* g = Fg + (size_t)f * D;
    float gn = 0.f;
    for (int e = lane; e < D; e += 64) gn += g[e] * g[e];
    gn = fmaxf(__fsqrt_rn(wave_sum_f32(gn)), 1e-6f);
    for (int i = wv; i < nm; i += 4) {
        const float* a = Fm + ((size_t)f * M + i) * D;
        const float* b = Fc + ((size_t)f * M + i) * D;
        float* o = Fp + ((size_t)f * MS + i) * D;
        float n2 = 0.f;
        for (int e = lane; e < D; e += 64) {
            float v = __fadd_rn(__fmul_rn(wm, a[e]), __fmul_rn(wc, b[e]));
            o[e] = v;                       // scratch: fused crop feature
            n2 += v * v;
        }
        float nl = fmaxf(__fsqrt_rn(wave_sum_f32(n2)), 1e-12f);
        float l2 = 0.f;
        for (int e = lane; e < D; e += 64) {
            float v = __fdiv_rn(o[e], nl);  // F_l
            o[e] = v;
            l2 += v * v;
        }
        float nl2 = fmaxf(__fsqrt_rn(wave_sum_f32(l2)), 1e-6f);
        float dot = 0.f;
        for (int e = lane; e < D; e += 64) dot += __fdiv_rn(o[e], nl2) * __fdiv_rn(g[e], gn);
        dot = wave_sum_f32(dot);
        if (lane == 0) phi[i] = dot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float mx = -3.4e38f;
        for (int i = 0; i < nm; ++i) mx = fmaxf(mx, phi[i]);
        float s = 0.f;
        for (int i = 0; i < nm; ++i) {
            wsm[i] = expf(phi[i] - mx);
            s += wsm[i];
        }
        for (int i = 0; i < nm; ++i) wsm[i] = __fdiv_rn(wsm[i], s);
    }
    __syncthreads();
    for (int i = wv; i < nm; i += 4) {
        float* o = Fp + ((size_t)f * MS + i) * D;
        float w = wsm[i], w1 = __fsub_rn(1.0f, w);
        float n2 = 0.f;
        for (int e = lane; e < D; e += 64) {
            float v = __fadd_rn(__fmul_rn(w, g[e]), __fmul_rn(w1, o[e]));
            o[e] = v;
            n2 += v * v;
        }
        float nn = fmaxf(__fsqrt_rn(wave_sum_f32(n2)), 1e-12f);
        for (int e = lane; e < D; e += 64) o[e] = __fdiv_rn(o[e], nn);
    }
}

#include "hmsg_nn.h"

__global__ void k_nn(const unsigned short* __restrict__ depth, const double* __restrict__ pose, CamK cam, float scale,
                     int H, int W, int f0, int nfr, NNIndex I, int* __restrict__ nn, TieList ties) {
    const size_t HW = (size_t)H * W;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= HW * nfr) return;
    const int fl = (int)(t / HW);
    const int p = (int)(t - (size_t)fl * HW);
    const int f = f0 + fl;
    const int y = p / W, x = p - y * W;
    double wx, wy, wz;
    int idx = -1;
    if (backproject(depth[(size_t)f * HW + p], x, y, cam, scale, pose + (size_t)f * 16, wx, wy, wz)) {
        int ntie = 0;
        idx = nn_search(I, wx, wy, wz, nullptr, &ntie);
        if (ntie > 1) tie_push(ties, (long long)((size_t)f * HW + p), wx, wy, wz);
    }
    nn[(size_t)f * HW + p] = idx;
}

// stamp = 1 + the LARGEST pixel index per (voxel, frame): within a run of consecutive lanes snapping to the same
// voxel only the last lane can win, so only it pays for the atomic (runs are ~8 pixels long)
__global__ void k_stamp(const int* __restrict__ nn, size_t HW, int f0, int nfr, unsigned* __restrict__ stamp /*[V][FB]*/) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = t < HW * nfr;
    if (!in_range) t = HW * nfr - 1;
    const int fl = (int)(t / HW);
    const int p = (int)(t - (size_t)fl * HW);
    const int idx = in_range ? nn[(size_t)(f0 + fl) * HW + p] : -1;
    const int lane = threadIdx.x & 63;
    long long key = idx >= 0 ? ((long long)idx << 8) | fl : -1 - lane;
    long long nxt = __shfl_down(key, 1);
    if (key >= 0 && (lane == 63 || nxt != key)) atomicMax(&stamp[(size_t)idx * FB + fl], (unsigned)p + 1u);
}

// ------------------------------------------------------------------------------------------ K_fuse (A5)
// One wave per voxel.  graph.py:410-411 with torch's last-writer-wins semantics (SURVEY hazard 7): per
// frame the voxel receives old + fp16(F_2D[p*]) for p* = its largest pixel index, counter += 1.
template <int VEC, int NJ>
__global__ void k_fuse(const unsigned* __restrict__ stamp, long long V, int f0, int nfr, int MS, int NW, int D, size_t HW,
                       const unsigned long long* __restrict__ bits, const float* __restrict__ Fp, float* __restrict__ sum,
                       unsigned* __restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const long long v = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (v >= V) return;
    unsigned st = stamp[(size_t)v * FB + lane];
    unsigned long long m = __ballot(st != 0u && lane < nfr);
    if (m == 0ull) return;
    float acc[NJ * VEC];
    float* row = sum + (size_t)v * D;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            int e = j * 64 * VEC + lane * VEC + k;
            acc[j * VEC + k] = e < D ? row[e] : 0.f;
        }
    int nfrm = 0;
    while (m) {
        int fl = __ffsll(m) - 1;
        m &= m - 1;
        ++nfrm;
        unsigned p = __shfl(st, fl) - 1u;
        int f = f0 + fl;
        const float* fpf = Fp + (size_t)f * MS * D;
        float x[NJ * VEC];
#pragma unroll
        for (int q = 0; q < NJ * VEC; ++q) x[q] = 0.f;
        _Pragma("unroll") for (int w = 0; w < 4; ++w) if (w < NW) {          // masks in index order (sam_clip_feats_extractor.py:183-187)
            unsigned long long b = bits[((size_t)f * HW + p) * NW + w];
            while (b) {
                int i = w * 64 + __ffsll(b) - 1;
                b &= b - 1;
                const float* r = fpf + (size_t)i * D;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        int e = j * 64 * VEC + lane * VEC + k;
                        if (e < D) x[j * VEC + k] = __fadd_rn(x[j * VEC + k], r[e]);
                    }
            }
        }
        float n2 = 0.f;
#pragma unroll
        for (int q = 0; q < NJ * VEC; ++q) n2 += x[q] * x[q];
        float nrm = fmaxf(__fsqrt_rn(wave_sum_f32(n2)), 1e-12f);
#pragma unroll
        for (int q = 0; q < NJ * VEC; ++q) {
            float y = __fdiv_rn(x[q], nrm);
            acc[q] = __fadd_rn(acc[q], __half2float(__float2half_rn(y)));
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            int e = j * 64 * VEC + lane * VEC + k;
            if (e < D) row[e] = acc[j * VEC + k];
        }
    if (lane == 0) cnt[v] += (unsigned)nfrm;
}

// graph.py:413-415: counter[counter == 0] = 1e-5; feats = sum / counter
__global__ void k_feats_final(const float* __restrict__ sum, const unsigned* __restrict__ cnt, long long V, int D,
                              float* __restrict__ feats) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)V * D) return;
    unsigned c = cnt[t / D];
    float d = c == 0u ? 1e-5f : (float)c;
    feats[t] = __fdiv_rn(sum[t], d);
}

// ------------------------------------------------------------------------------------------ mask clouds (A4)
struct MaskGeom {           // per (frame, mask) local Open3D voxel grid of the snapped points
    double ox, oy, oz;      // min_bound - vs/2
    int nx, ny, nz;
    int pad;
    long long word_off;     // first bitmap word of this mask in the batch bitmap
};

#define MCHUNK 4096         /* pixels per workgroup of the run kernels (record positions come from per-chunk counts) */

// A run = consecutive pixels of one image row, inside one 64-pixel wave slice, with the same nearest voxel and
// the same mask set.  Every mask kernel below re-detects the runs from (nn, bits) with this routine, so they
// all see the same runs; the lane holding a run's LAST pixel works for the run.
// (the mask words are indexed by fully unrolled loops only -- `for (w < 4) if (w < NW)` -- so that r.b[] stays in registers: with a
//  run-time bound the array went to scratch memory, 120 bytes a lane, and k_mbounds wrote 1.1 - 1.5 GB of it per 64-frame batch,
//  profiles/r05_pmc_traffic.json vs r04)
struct MaskRun {
    bool tail;              // this lane closes a run that has a voxel and at least one mask
    int v, len;
    unsigned long long b[4];
    unsigned dsum;          // sum of the run's u16 depths (create_3d_masks' filter_distance test)
};
__device__ __forceinline__ MaskRun mask_run(const int* __restrict__ nn, const unsigned long long* __restrict__ bits,
                                            const unsigned short* __restrict__ depth, int NW, int W, size_t g, bool in_range,
                                            bool want_depth) {
    MaskRun r;
    const int lane = threadIdx.x & 63;
    r.v = in_range ? nn[g] : -1;
    bool any = false;
    for (int w = 0; w < 4; ++w) {
        r.b[w] = (w < NW && r.v >= 0) ? bits[g * NW + w] : 0ull;
        any = any || r.b[w] != 0ull;
    }
    const int x = in_range ? (int)(g % (size_t)W) : 0;
    const int pv = __shfl_up(r.v, 1);                  // (shuffles outside any short-circuit: all lanes take part)
    bool head = lane == 0 || x == 0 || pv != r.v;
    _Pragma("unroll") for (int w = 0; w < 4; ++w) if (w < NW) {
        const unsigned long long pb = __shfl_up(r.b[w], 1);
        head = head || pb != r.b[w];
    }
    const unsigned long long heads = __ballot(head);
    const int nhead = __shfl_down(head ? 1 : 0, 1);
    const bool tail = lane == 63 || nhead != 0;
    const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
    const int start_lane = 63 - __clzll(below);
    r.len = lane - start_lane + 1;
    r.dsum = 0u;
    if (want_depth) {
        unsigned d = (in_range && r.v >= 0) ? (unsigned)depth[g] : 0u;
        int flag = head ? 1 : 0;
        for (int o = 1; o < 64; o <<= 1) {       // segmented inclusive scan (integer: exact)
            unsigned td = __shfl_up(d, o);
            int tf = __shfl_up(flag, o);
            if (lane >= o && !flag) {
                d += td;
                flag = tf;
            }
        }
        r.dsum = d;
    }
    r.tail = tail && r.v >= 0 && any;
    return r;
}

// per (frame, mask): AABB of the snapped points, number of valid pixels and the sum of their depths.
// grid = (chunks per frame, frames): a workgroup stays inside one frame, combines in LDS (every run of a mask hits
// the same eight words) and touches the global table once per mask it met.
// Round 5: this pass -- the only one that has to look at every pixel -- also writes the chunk's RUNS down, in pixel order: (map
// voxel << 8 | length) and the mask words, 8 (1 + NW) bytes a run, a chunk's runs at runs[chunk * MCHUNK ...] with their number in
// run_count[chunk].  k_mmark and k_memit then walk runs (a run covers ~8 pixels) instead of detecting them again from 12 bytes a
// pixel each: the three passes moved 1.47 + 0.85 + 0.94 GB per 64-frame batch (profiles/r04_pmc_traffic.json).
__global__ void __launch_bounds__(256) k_mbounds(const int* __restrict__ nn, const unsigned long long* __restrict__ bits,
                                                 const unsigned short* __restrict__ depth, size_t HW, int W, int f0,
                                                 int NW, int MS, const double* __restrict__ pts,
                                                 unsigned long long* __restrict__ bounds /*[nfr*MS][6]*/,
                                                 unsigned long long* __restrict__ dstat /*[nfr*MS][2]: depth sum, pixels*/,
                                                 unsigned long long* __restrict__ runs, unsigned* __restrict__ run_count) {
    __shared__ unsigned long long s_b[HMSG_MAX_MASKS][8];
    __shared__ unsigned s_w[4];
    __shared__ unsigned s_run;
    for (int k = threadIdx.x; k < MS * 8; k += 256) s_b[k >> 3][k & 7] = (k & 7) < 3 ? ~0ull : 0ull;
    if (threadIdx.x == 0) s_run = 0u;
    __syncthreads();
    const int fl = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t chunk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    unsigned long long* const my_runs = runs + chunk * (size_t)MCHUNK * (size_t)(1 + NW);
    for (int it = 0; it < MCHUNK / 256; ++it) {
        const size_t t = (size_t)blockIdx.x * MCHUNK + (size_t)it * 256 + threadIdx.x;
        const bool in_range = t < HW;
        const size_t g = (size_t)(f0 + fl) * HW + (in_range ? t : 0);
        const MaskRun r = mask_run(nn, bits, depth, NW, W, g, in_range, true);
        {   // the run list: tails in lane order = pixel order
            const unsigned long long tm = __ballot(r.tail);
            if (lane == 0) s_w[wv] = (unsigned)__popcll(tm);
            __syncthreads();
            unsigned pos = s_run + (unsigned)__popcll(tm & ((1ull << lane) - 1ull));
            for (int q = 0; q < wv; ++q) pos += s_w[q];
            // (the trip's total is read BEFORE the barrier below: behind it the other waves may already be writing the next
            //  trip's s_w while thread 0 has not advanced s_run yet)
            const unsigned trip_total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
            if (r.tail) {
                unsigned long long* rec = my_runs + (size_t)pos * (size_t)(1 + NW);
                rec[0] = ((unsigned long long)(unsigned)r.v << 8) | (unsigned long long)r.len;
                _Pragma("unroll") for (int w = 0; w < 4; ++w) if (w < NW) rec[1 + w] = r.b[w];
            }
            __syncthreads();
            if (threadIdx.x == 0) s_run += trip_total;
        }
        if (!r.tail) continue;
        unsigned long long e[3] = {enc_f64(pts[(size_t)r.v * 3]), enc_f64(pts[(size_t)r.v * 3 + 1]), enc_f64(pts[(size_t)r.v * 3 + 2])};
        _Pragma("unroll") for (int w = 0; w < 4; ++w) if (w < NW) {
            unsigned long long b = r.b[w];
            while (b) {
                const int i = w * 64 + __ffsll(b) - 1;
                b &= b - 1;
                unsigned long long* bd = s_b[i];
                for (int a = 0; a < 3; ++a) {      // (stale reads only cost a redundant atomic)
                    if (e[a] < bd[a]) atomicMin(&bd[a], e[a]);
                    if (e[a] > bd[3 + a]) atomicMax(&bd[3 + a], e[a]);
                }
                atomicAdd(&bd[6], (unsigned long long)r.dsum);
                atomicAdd(&bd[7], (unsigned long long)r.len);
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < MS * 8; k += 256) {
        const int i = k >> 3, a = k & 7;
        if (s_b[i][7] == 0ull) continue;
        const size_t m = (size_t)fl * MS + i;
        if (a < 3) atomicMin(&bounds[m * 6 + a], s_b[i][a]);
        else if (a < 6) atomicMax(&bounds[m * 6 + a], s_b[i][a]);
        else atomicAdd(&dstat[m * 2 + (a - 6)], s_b[i][a]);
    }
    if (threadIdx.x == 0) run_count[chunk] = s_run;
}

__device__ __forceinline__ long long mask_cell(const MaskGeom& mg, double vs, const double* __restrict__ p, int& ix, int& iy,
                                               int& iz) {
    ix = (int)floor(__ddiv_rn(__dsub_rn(p[0], mg.ox), vs));
    iy = (int)floor(__ddiv_rn(__dsub_rn(p[1], mg.oy), vs));
    iz = (int)floor(__ddiv_rn(__dsub_rn(p[2], mg.oz), vs));
    return ((long long)ix * mg.ny + iy) * mg.nz + iz;
}

// occupancy bitmaps of the per-mask Open3D grids + number of records every chunk will emit; one thread per RUN of the chunk's list
__global__ void __launch_bounds__(256) k_mmark(const unsigned long long* __restrict__ runs, const unsigned* __restrict__ run_count, int NW,
                                               int MS, const double* __restrict__ pts, const MaskGeom* __restrict__ geom, double vs,
                                               unsigned long long* __restrict__ mbitmap, unsigned* __restrict__ chunk_recs) {
    __shared__ unsigned s_n;
    if (threadIdx.x == 0) s_n = 0u;
    __syncthreads();
    unsigned mine = 0;
    const int fl = blockIdx.y;
    const size_t chunk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned n = run_count[chunk];
    const unsigned long long* const my_runs = runs + chunk * (size_t)MCHUNK * (size_t)(1 + NW);
    for (unsigned k = threadIdx.x; k < n; k += 256u) {
        const unsigned long long* rec = my_runs + (size_t)k * (size_t)(1 + NW);
        const int v = (int)(rec[0] >> 8);
        _Pragma("unroll") for (int w = 0; w < 4; ++w) if (w < NW) {
            unsigned long long b = rec[1 + w];
            while (b) {
                const int i = w * 64 + __ffsll(b) - 1;
                b &= b - 1;
                const MaskGeom mg = geom[(size_t)fl * MS + i];
                if (mg.nx == 0) continue;                       // mask rejected by filter_distance
                int ix, iy, iz;
                const long long lin = mask_cell(mg, vs, pts + (size_t)v * 3, ix, iy, iz);
                unsigned long long* wp = mbitmap + mg.word_off + (lin >> 6);
                const unsigned long long bit = 1ull << (lin & 63);
                if (!(*wp & bit)) atomicOr(wp, bit);
                ++mine;
            }
        }
    }
    mine = (unsigned)wave_sum_i32((int)mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_n, mine);
    __syncthreads();
    if (threadIdx.x == 0) chunk_recs[chunk] = s_n;
}

// records (key = slot of the mask voxel, value = map voxel << 8 | run length) in pixel order = run order, then mask order
__global__ void __launch_bounds__(256) k_memit(const unsigned long long* __restrict__ runs, const unsigned* __restrict__ run_count, int NW,
                                               int MS, const double* __restrict__ pts, const MaskGeom* __restrict__ geom, double vs,
                                               const unsigned long long* __restrict__ mbitmap, const unsigned* __restrict__ mrank,
                                               const unsigned* __restrict__ chunk_base, unsigned* __restrict__ keys,
                                               unsigned long long* __restrict__ vals) {
    __shared__ unsigned s_w[4];
    __shared__ unsigned s_run;
    const size_t chunk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) s_run = chunk_base[chunk];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int fl = blockIdx.y;
    const unsigned n = run_count[chunk];
    const unsigned long long* const my_runs = runs + chunk * (size_t)MCHUNK * (size_t)(1 + NW);
    for (unsigned k0 = 0; k0 < n; k0 += 256u) {                 // (workgroup-uniform trip count)
        const unsigned k = k0 + threadIdx.x;
        const bool live = k < n;
        unsigned long long head = 0ull, b4[4] = {0ull, 0ull, 0ull, 0ull};
        if (live) {
            const unsigned long long* rec = my_runs + (size_t)k * (size_t)(1 + NW);
            head = rec[0];
            _Pragma("unroll") for (int w = 0; w < 4; ++w) if (w < NW) b4[w] = rec[1 + w];
        }
        // records of this lane: its masks whose cloud was not rejected
        unsigned nrec = 0;
        _Pragma("unroll") for (int w = 0; w < 4; ++w) if (w < NW) {
            unsigned long long b = b4[w];
            while (b) {
                const int i = w * 64 + __ffsll(b) - 1;
                b &= b - 1;
                nrec += geom[(size_t)fl * MS + i].nx != 0 ? 1u : 0u;
            }
        }
        unsigned incl = nrec;                       // inclusive wave scan
        for (int o = 1; o < 64; o <<= 1) {
            unsigned u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        unsigned pos = s_run + incl - nrec;
        for (int q = 0; q < wv; ++q) pos += s_w[q];
        if (nrec) {
            const int v = (int)(head >> 8);
            _Pragma("unroll") for (int w = 0; w < 4; ++w) if (w < NW) {
                unsigned long long b = b4[w];
                while (b) {
                    const int i = w * 64 + __ffsll(b) - 1;
                    b &= b - 1;
                    const MaskGeom mg = geom[(size_t)fl * MS + i];
                    if (mg.nx == 0) continue;
                    int ix, iy, iz;
                    const long long lin = mask_cell(mg, vs, pts + (size_t)v * 3, ix, iy, iz);
                    const long long wd = mg.word_off + (lin >> 6);
                    keys[pos] = mrank[wd] + (unsigned)__popcll(mbitmap[wd] & ((1ull << (lin & 63)) - 1ull));
                    vals[pos] = head;
                    ++pos;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_run += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
}

// one lane per mask voxel: replay Open3D's accumulation -- sequential float64 sum of the snapped points in pixel
// order, then / count (generic.py:188 -> o3d_voxel_down_sample).
// A workgroup owns 256 consecutive voxels, whose records are one contiguous range of the sorted record array.  The
// records and the map points they name are fetched cooperatively (coalesced, all in flight at once) into LDS, a
// chunk at a time; every lane then consumes its voxel's records from LDS in a FLATTENED loop -- one addition per
// trip, a lane that runs out of repetitions picks up its next record -- so a wave runs as long as its busiest
// voxel's pixel count, not (records x longest repetition), and no lane ever waits on a dependent global load.
// Long repetitions of one point are added in closed form (repeat_add).
#define MW_CHUNK 1024
__global__ void __launch_bounds__(256) k_mwalk(const unsigned* __restrict__ off, const unsigned long long* __restrict__ recs,
                                               long long npts, const double* __restrict__ pts, double* __restrict__ out,
                                               unsigned* __restrict__ heavy_cnt, unsigned* __restrict__ heavy_list, unsigned heavy_thr) {
    __shared__ double s_p[MW_CHUNK][3];
    __shared__ unsigned s_v[MW_CHUNK];
    __shared__ unsigned char s_len[MW_CHUNK];
    const long long s0 = (long long)blockIdx.x * 256, s = s0 + threadIdx.x;
    bool live = s < npts;
    const unsigned r1 = live ? off[s + 1] : 0u;
    unsigned r = live ? off[s] : 0u;
    {   // voxels with many records (thousands of pixels of a deleted region snap to the few surviving voxels next to
        // it, SURVEY hazard 15) go to the wave-per-voxel kernel; list slots per wave, one atomic on the counter
        const bool heavy = live && r1 - r >= heavy_thr;
        const unsigned long long hm = __ballot(heavy);
        if (hm) {
            const int lane = threadIdx.x & 63, leader = __ffsll(hm) - 1;
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(heavy_cnt, (unsigned)__popcll(hm));
            base = __shfl(base, leader);
            if (heavy) heavy_list[base + (unsigned)__popcll(hm & ((1ull << lane) - 1ull))] = (unsigned)s;
        }
        if (heavy) {
            live = false;
            r = r1;
        }
    }
    const unsigned blk_r0 = off[s0], blk_r1 = off[s0 + 256 < npts ? s0 + 256 : npts];
    double sx = 0.0, sy = 0.0, sz = 0.0, px = 0.0, py = 0.0, pz = 0.0;
    unsigned long long n = 0;
    int rem = 0;
    for (unsigned c0 = blk_r0; c0 < blk_r1; c0 += MW_CHUNK) {
        const unsigned c1 = c0 + MW_CHUNK < blk_r1 ? c0 + MW_CHUNK : blk_r1;
        for (unsigned k = c0 + threadIdx.x; k < c1; k += 256) {
            const unsigned long long rec = recs[k];
            const double* p = pts + (size_t)(rec >> 8) * 3;
            s_len[k - c0] = (unsigned char)(rec & 255ull);
            s_v[k - c0] = (unsigned)(rec >> 8);
            s_p[k - c0][0] = p[0];
            s_p[k - c0][1] = p[1];
            s_p[k - c0][2] = p[2];
        }
        __syncthreads();
        const unsigned e = r1 < c1 ? r1 : c1;        // this lane's records inside the chunk: [r, e)
        for (;;) {
            if (rem == 0) {
                if (r >= e) break;
                const unsigned i = r - c0;
                rem = (int)s_len[i];
                px = s_p[i][0];
                py = s_p[i][1];
                pz = s_p[i][2];
                ++r;
                // following records of the SAME map point (the voxel on the next image rows) extend the repetition;
                // a long one -- thousands of pixels of a deleted region all snap to the one surviving voxel next to
                // it (SURVEY hazard 15) -- is added in closed form
                while (r < e && s_v[r - c0] == s_v[i]) rem += (int)s_len[r++ - c0];
                n += (unsigned long long)rem;
                if (rem >= 256) {
                    sx = repeat_add(sx, px, rem);
                    sy = repeat_add(sy, py, rem);
                    sz = repeat_add(sz, pz, rem);
                    rem = 0;
                    continue;
                }
            }
            sx = __dadd_rn(sx, px);
            sy = __dadd_rn(sy, py);
            sz = __dadd_rn(sz, pz);
            --rem;
        }
        __syncthreads();
    }
    if (live) {
        const double dn = (double)n;
        out[(size_t)s * 3 + 0] = __ddiv_rn(sx, dn);
        out[(size_t)s * 3 + 1] = __ddiv_rn(sy, dn);
        out[(size_t)s * 3 + 2] = __ddiv_rn(sz, dn);
    }
}

// The heavy voxels: one WAVE per voxel, 64 records side by side.  While the running sum s stays inside one binade
// its ulp u is fixed and every addition of a point p adds the same integer d(p) = round(p / u) to S = s / u (see
// repeat_add): the order of the additions no longer matters, so the lanes compute len * d(p) for their records, a
// prefix sum finds how many records fit before S would leave the binade (or a record needs care: round-half-even
// tie, opposite sign, p larger than s), those are added in ONE step, and the record that did not fit is replayed
// exactly by repeat_add.  A sum crosses ~20 binades in its life, so a voxel with thousands of records costs a few
// dozen rounds.
__global__ void __launch_bounds__(256) k_mwalk_heavy(const unsigned* __restrict__ off, const unsigned long long* __restrict__ recs,
                                                     const double* __restrict__ pts, double* __restrict__ out,
                                                     const unsigned* __restrict__ heavy_cnt, const unsigned* __restrict__ heavy_list) {
    const int lane = threadIdx.x & 63;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6, nheavy = *heavy_cnt;
    // (a wave operation on every path: the kernel simulator of tests/emu classifies a kernel by its first launch)
    if (__ballot(1) == 0ull) return;
    const unsigned long long top = (1ull << 53) - 1ull;
    for (unsigned t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; t < nheavy; t += nwaves) {
        const unsigned slot = heavy_list[t];
        const unsigned r0 = off[slot], r1 = off[slot + 1];
        double res[3];
        unsigned long long n = 0;
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (unsigned rb = r0; rb < r1; rb += 64) {
                const int cnt = (int)min(64u, r1 - rb);
                const unsigned long long rec = lane < cnt ? recs[rb + lane] : 0ull;
                const int len = (int)(rec & 255ull);
                const double p = lane < cnt ? pts[(size_t)(rec >> 8) * 3 + c] : 0.0;
                if (c == 0) {
                    int tl = len;
                    for (int o = 32; o > 0; o >>= 1) tl += __shfl_xor(tl, o);
                    n += (unsigned long long)tl;
                }
                int j0 = 0;
                while (j0 < cnt) {
                    // state of the running sum
                    const double as = fabs(s);
                    const long long bs = __double_as_longlong(as);
                    const int e = (int)(bs >> 52);
                    const unsigned long long S = ((unsigned long long)bs & 0xfffffffffffffull) | (1ull << 52);
                    const bool st_ok = e >= 1 && e < 2046;
                    // this lane's record against it
                    const double ap = fabs(p);
                    bool ok = st_ok && lane >= j0 && lane < cnt && as >= ap && (ap == 0.0 || (p < 0.0) == (s < 0.0));
                    unsigned long long d = 0ull, iq = 0ull;
                    if (ok && ap != 0.0) {
                        const double q = ldexp(ap, 1075 - e);
                        if (q >= 0.5) {
                            const double qi = floor(q), qf = q - qi;
                            iq = (unsigned long long)qi;
                            d = iq + (qf > 0.5 ? 1ull : 0ull);
                            ok = qf != 0.5 && d < (1ull << 44);
                        }
                    }
                    const unsigned long long inc = ok ? (unsigned long long)len * d : 0ull;
                    unsigned long long pre = inc;                    // inclusive prefix over the lanes
                    for (int o = 1; o < 64; o <<= 1) {
                        const unsigned long long u = __shfl_up(pre, o);
                        if (lane >= o) pre += u;
                    }
                    const bool fits = ok && S + pre - d + iq <= top;
                    const unsigned long long todo = ((cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull)) >> j0) << j0;
                    const unsigned long long bad = ~__ballot(fits) & todo;
                    const int f = bad ? __ffsll(bad) - 1 : cnt;      // first record that is not added in bulk
                    if (f > j0) {
                        const unsigned long long add = __shfl(pre, f - 1);
                        const double rr = ldexp((double)(S + add), e - 1075);
                        s = s < 0.0 ? -rr : rr;
                    }
                    if (f < cnt) s = repeat_add(s, wave_bcast_f64(p, f), __builtin_amdgcn_readlane(len, f));
                    j0 = f + 1;
                }
            }
            res[c] = s;
        }
        if (lane == 0) {
            const double dn = (double)n;
            out[(size_t)slot * 3 + 0] = __ddiv_rn(res[0], dn);
            out[(size_t)slot * 3 + 1] = __ddiv_rn(res[1], dn);
            out[(size_t)slot * 3 + 2] = __ddiv_rn(res[2], dn);
        }
    }
}

__global__ void k_gather_u32(const unsigned* __restrict__ src, const long long* __restrict__ idx, int n, unsigned* __restrict__ dst) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = src[idx[t]];
}

// ------------------------------------------------------------------------------------------ host driver
template <int VEC, int NJ>
static void launch_fuse(hmsg_ctx* h, const unsigned* stamp, int f0, int nfr) {
    const size_t HW = (size_t)h->cfg.height * h->cfg.width;
    hipLaunchKernelGGL((k_fuse<VEC, NJ>), dim3(cdiv((size_t)h->V * 64, 256)), dim3(256), 0, h->stream, stamp, h->V, f0, nfr,
                       h->MS, h->NW, h->cfg.feat_dim, HW, (const unsigned long long*)h->bits.p, (const float*)h->fp.p, h->sum.p,
                       h->cnt.p);
    HMSG_CHECK_LAUNCH();
}

static void dispatch_fuse(hmsg_ctx* h, const unsigned* stamp, int f0, int nfr) {
    const int D = h->cfg.feat_dim;
    if (D % 256 == 0 && D <= 1024) {
        switch (D / 256) {
            case 1: launch_fuse<4, 1>(h, stamp, f0, nfr); return;
            case 2: launch_fuse<4, 2>(h, stamp, f0, nfr); return;
            case 3: launch_fuse<4, 3>(h, stamp, f0, nfr); return;
            case 4: launch_fuse<4, 4>(h, stamp, f0, nfr); return;
        }
    }
    if (D <= 64) return launch_fuse<1, 1>(h, stamp, f0, nfr);
    if (D <= 128) return launch_fuse<1, 2>(h, stamp, f0, nfr);
    if (D <= 192) return launch_fuse<1, 3>(h, stamp, f0, nfr);
    if (D <= 256) return launch_fuse<1, 4>(h, stamp, f0, nfr);
    throw hmsg_error{HMSG_ERR_UNSUPPORTED, "feat_dim must be <= 256 or a multiple of 256 up to 1024"};
}

void hmsg_fuse(hmsg_ctx* h) {
    const hmsg_config& c = h->cfg;
    hipStream_t s = h->stream;
    HMSG_REQUIRE(h->map_ready, HMSG_ERR_INVALID, "hmsg_fuse_frames: call hmsg_finalize_map first");
    HMSG_REQUIRE(!h->feats_final || h->n_fused == h->n_feat_frames, HMSG_ERR_INVALID, "feature map already finalised");
    const int H = c.height, W = c.width, D = c.feat_dim, MS = h->MS, NW = h->NW;
    const size_t HW = (size_t)H * W;
    const long long V = h->V;
    const float scale = (float)c.depth_scale;
    if (h->sum.n < (size_t)V * D || h->n_fused == h->frame_window) {
        h->sum.alloc((size_t)V * D);
        h->cnt.alloc((size_t)V);
        h->sum.zero(s);
        h->cnt.zero(s);
    }
    if (h->nn.n < (size_t)c.max_frames * HW) h->nn.alloc((size_t)c.max_frames * HW);
    DevBuf<unsigned> stamp;
    stamp.alloc((size_t)std::max<long long>(V, 1) * FB);
    const int nmask_max = FB * MS;
    DevBuf<unsigned long long> d_bounds, d_dstat;
    d_bounds.alloc((size_t)nmask_max * 6);
    d_dstat.alloc((size_t)nmask_max * 2);
    DevBuf<MaskGeom> d_geom;
    d_geom.alloc((size_t)nmask_max);
    DevBuf<unsigned long long> mbitmap;
    DevBuf<unsigned> mrank, chunk_recs, rec_off, heavy, run_count;
    DevBuf<unsigned long long> runs;                  // the batch's run lists: MCHUNK slots of (1 + NW) words per chunk
    // (HMSG_DEBUG_MWALK_HEAVY: tests push every voxel through the wave-per-voxel replay)
    const unsigned heavy_thr = getenv("HMSG_DEBUG_MWALK_HEAVY") ? (unsigned)atoi(getenv("HMSG_DEBUG_MWALK_HEAVY")) : 32u;
    DevBuf<long long> d_offidx;
    DevBuf<unsigned> d_offval;
    d_offidx.alloc((size_t)nmask_max);
    d_offval.alloc((size_t)nmask_max);
    SortBufs sb;
    TieBuf ties;
    std::vector<unsigned long long> hb((size_t)nmask_max * 6), hd((size_t)nmask_max * 2);
    std::vector<MaskGeom> hg((size_t)nmask_max);
    std::vector<long long> hoffidx((size_t)nmask_max);
    std::vector<unsigned> hoffval((size_t)nmask_max);
    if (h->masks3d.off.empty()) h->masks3d.off.assign(1, 0);
    if (h->mask_first.empty()) h->mask_first.assign(1, 0);
    // create_3d_masks' filter_distance (generic.py:126-127): the mask is dropped when the mean camera depth of its
    // valid pixels exceeds the threshold.  The mean is taken over the u16 depths (exact integer sum); the reference
    // averages float32 metres, so a mask within ~1e-7 (relative) of the threshold can fall on the other side.
    const double filt_mm = c.max_mask_distance * c.depth_scale;

    for (int fb0 = h->n_fused; fb0 < h->n_feat_frames; fb0 += FB) {
        const int nb = std::min(FB, h->n_feat_frames - fb0);
        stamp.zero(s);
        for (;;) {      // (repeated only when the tie list overflowed)
            ties.prepare(s);
            {
                ProfScope ps(h->prof, s, "k_nn", (double)nb * ((double)HW * 6.0 + 128.0));
                hipLaunchKernelGGL(k_nn, dim3(cdiv(HW * nb, 256)), dim3(256), 0, s, (const unsigned short*)h->depth.p,
                                   (const double*)h->pose.p, h->cam, scale, H, W, fb0, nb, hmsg_nn_index(h), h->nn.p, ties.list());
            }
            HMSG_CHECK_LAUNCH();
            if (hmsg_resolve_ties(h, ties, h->nn.p)) break;
        }
        hipLaunchKernelGGL(k_stamp, dim3(cdiv(HW * nb, 256)), dim3(256), 0, s, (const int*)h->nn.p, HW, fb0, nb, stamp.p);
        HMSG_CHECK_LAUNCH();
        {
            ProfScope ps(h->prof, s, "k_fuse", (double)V * 256.0);
            dispatch_fuse(h, stamp.p, fb0, nb);
        }
        // ---- 3-D masks of the batch
        const int nmask = nb * MS;
        const unsigned cpf = cdiv(HW, MCHUNK);           // chunks per frame (a workgroup never straddles two frames)
        const unsigned nchunks = cpf * (unsigned)nb;
        for (int i = 0; i < nmask; ++i)
            for (int a = 0; a < 6; ++a) hb[(size_t)i * 6 + a] = a < 3 ? ~0ull : 0ull;
        HIP_TRY(hipMemcpyAsync(d_bounds.p, hb.data(), (size_t)nmask * 48, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(d_dstat.p, 0, (size_t)nmask * 16, s));
        runs.ensure((size_t)nchunks * MCHUNK * (size_t)(1 + NW));
        run_count.ensure((size_t)nchunks);
        {
            ProfScope ps(h->prof, s, "k_mbounds", (double)nb * (double)HW * (6.0 + 8.0 * NW));
            hipLaunchKernelGGL(k_mbounds, dim3(cpf, nb), dim3(256), 0, s, (const int*)h->nn.p, (const unsigned long long*)h->bits.p,
                               (const unsigned short*)h->depth.p, HW, W, fb0, NW, MS, (const double*)h->pts.p, d_bounds.p,
                               d_dstat.p, runs.p, run_count.p);
        }
        HMSG_CHECK_LAUNCH();
        HIP_TRY(hipMemcpyAsync(hb.data(), d_bounds.p, (size_t)nmask * 48, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(hd.data(), d_dstat.p, (size_t)nmask * 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        long long nwords = 0;
        for (int i = 0; i < nmask; ++i) {
            MaskGeom& mg = hg[i];
            mg.word_off = nwords;
            mg.pad = 0;
            const bool none = hb[(size_t)i * 6] == ~0ull;   // no valid pixel in this mask
            const bool far = !none && (double)hd[(size_t)i * 2] / (double)hd[(size_t)i * 2 + 1] > filt_mm;
            if (none || far) {
                mg.ox = mg.oy = mg.oz = 0.0;
                mg.nx = mg.ny = mg.nz = 0;
                continue;
            }
            double mn[3], mx[3];
            for (int a = 0; a < 3; ++a) {
                mn[a] = dec_f64(hb[(size_t)i * 6 + a]);
                mx[a] = dec_f64(hb[(size_t)i * 6 + 3 + a]);
            }
            mg.ox = mn[0] - c.voxel_size * 0.5;
            mg.oy = mn[1] - c.voxel_size * 0.5;
            mg.oz = mn[2] - c.voxel_size * 0.5;
            mg.nx = (int)std::floor((mx[0] - mg.ox) / c.voxel_size) + 2;
            mg.ny = (int)std::floor((mx[1] - mg.oy) / c.voxel_size) + 2;
            mg.nz = (int)std::floor((mx[2] - mg.oz) / c.voxel_size) + 2;
            nwords += ((long long)mg.nx * mg.ny * mg.nz + 63) / 64;
        }
        long long npts = 0;
        std::vector<long long> start(nmask + 1, 0);
        if (nwords) {
            HMSG_REQUIRE(nwords < (1ll << 31), HMSG_ERR_UNSUPPORTED, "mask grids of one frame batch exceed 2^31 words");
            HIP_TRY(hipMemcpyAsync(d_geom.p, hg.data(), (size_t)nmask * sizeof(MaskGeom), hipMemcpyHostToDevice, s));
            mbitmap.ensure((size_t)nwords);
            mrank.ensure((size_t)nwords);
            chunk_recs.ensure((size_t)nchunks + 1);
            HIP_TRY(hipMemsetAsync(mbitmap.p, 0, (size_t)nwords * 8, s));
            {
                ProfScope ps(h->prof, s, "k_mmark", (double)nb * (double)HW * (4.0 + 8.0 * NW));
                hipLaunchKernelGGL(k_mmark, dim3(cpf, nb), dim3(256), 0, s, (const unsigned long long*)runs.p, (const unsigned*)run_count.p, NW,
                                   MS, (const double*)h->pts.p, (const MaskGeom*)d_geom.p, c.voxel_size, mbitmap.p, chunk_recs.p);
            }
            HMSG_CHECK_LAUNCH();
            npts = (long long)hmsg_bitmap_rank(mbitmap.p, mrank.p, (size_t)nwords, s, h->scan_tmp);
            unsigned long long nrec = 0;
            hmsg_scan_u32(chunk_recs.p, chunk_recs.p, (size_t)nchunks, s, h->scan_tmp, &nrec);
            HMSG_REQUIRE(npts < (1ll << 32) && nrec < (1ull << 32), HMSG_ERR_UNSUPPORTED, "mask batch too large");
            sb.keys.ensure((size_t)std::max<unsigned long long>(nrec, 1));
            sb.vals.ensure((size_t)std::max<unsigned long long>(nrec, 1));
            {
                ProfScope ps(h->prof, s, "k_memit", (double)nb * (double)HW * (4.0 + 8.0 * NW) + (double)nrec * 12.0);
                hipLaunchKernelGGL(k_memit, dim3(cpf, nb), dim3(256), 0, s, (const unsigned long long*)runs.p, (const unsigned*)run_count.p, NW,
                                   MS, (const double*)h->pts.p, (const MaskGeom*)d_geom.p, c.voxel_size,
                                   (const unsigned long long*)mbitmap.p, (const unsigned*)mrank.p, (const unsigned*)chunk_recs.p,
                                   sb.keys.p, sb.vals.p);
            }
            HMSG_CHECK_LAUNCH();
            {
                ProfScope ps(h->prof, s, "sort_mask_recs", (double)nrec * 12.0 * 4.0);
                hmsg_sort_pairs(sb, (size_t)nrec, bits_for((unsigned long long)npts), s);
            }
            rec_off.ensure((size_t)npts + 1);
            hmsg_sort_segment_starts(sb.res_keys, (size_t)nrec, rec_off.p, (unsigned)npts, s);
            // append to the resident mask-cloud store
            size_t need = (size_t)(h->masks3d.total + npts) * 3;
            if (need > h->masks3d.pts.n) {
                DevBuf<double> bigger;
                bigger.alloc(std::max(need * 2, (size_t)1 << 20));
                if (h->masks3d.total)
                    HIP_TRY(hipMemcpyAsync(bigger.p, h->masks3d.pts.p, (size_t)h->masks3d.total * 24, hipMemcpyDeviceToDevice, s));
                HIP_TRY(hipStreamSynchronize(s));
                bigger.swap(h->masks3d.pts);
            }
            {
                ProfScope ps(h->prof, s, "k_mwalk", (double)nrec * 8.0 + (double)npts * 32.0);
                heavy.ensure((size_t)npts + 1);
                HIP_TRY(hipMemsetAsync(heavy.p, 0, 4, s));                        // [0] = counter, the list follows
                hipLaunchKernelGGL(k_mwalk, dim3(cdiv((size_t)npts, 256)), dim3(256), 0, s, (const unsigned*)rec_off.p,
                                   (const unsigned long long*)sb.res_vals, npts, (const double*)h->pts.p,
                                   h->masks3d.pts.p + (size_t)h->masks3d.total * 3, heavy.p, heavy.p + 1, heavy_thr);
                hipLaunchKernelGGL(k_mwalk_heavy, dim3(2048), dim3(256), 0, s, (const unsigned*)rec_off.p,
                                   (const unsigned long long*)sb.res_vals, (const double*)h->pts.p,
                                   h->masks3d.pts.p + (size_t)h->masks3d.total * 3, (const unsigned*)heavy.p,
                                   (const unsigned*)(heavy.p + 1));
            }
            HMSG_CHECK_LAUNCH();
            if (getenv("HMSG_DEBUG_TIMING"))
                fprintf(stderr, "[hmsg fuse] batch at frame %d: mask voxels %lld  records %llu  bitmap words %lld\n", fb0, npts, nrec, nwords);
            // per-mask point offsets = rank at the mask's first word
            int ng = 0;
            for (int i = 0; i < nmask; ++i)
                if (hg[i].nx) hoffidx[ng++] = hg[i].word_off;
            HIP_TRY(hipMemcpyAsync(d_offidx.p, hoffidx.data(), (size_t)ng * 8, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_gather_u32, dim3(cdiv(ng, 256)), dim3(256), 0, s, (const unsigned*)mrank.p,
                               (const long long*)d_offidx.p, ng, d_offval.p);
            HMSG_CHECK_LAUNCH();
            HIP_TRY(hipMemcpyAsync(hoffval.data(), d_offval.p, (size_t)ng * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            // sizes: difference of consecutive starts (empty masks get 0)
            start.assign(nmask + 1, npts);
            int k = 0;
            for (int i = 0; i < nmask; ++i)
                if (hg[i].nx) start[i] = hoffval[k++];
            for (int i = nmask - 1; i >= 0; --i)
                if (!hg[i].nx) start[i] = start[i + 1];
        }
        // the store keeps the first nmask[f] masks of every frame
        for (int fl = 0; fl < nb; ++fl) {
            const int nm = h->nmask[(size_t)fb0 + fl];
            for (int i = 0; i < nm; ++i) h->masks3d.off.push_back(h->masks3d.total + start[(size_t)fl * MS + i + 1]);
            h->mask_first.push_back(h->mask_first.back() + nm);
            // (slots nm .. MS-1 of the frame hold no pixel: their start equals the next one's, nothing is skipped)
        }
        h->masks3d.total += npts;
        h->n_fused = fb0 + nb;
        hmsg_fold_pipe_feed(h, fb0, nb);           // the sequential fold starts on these frames right away (FoldPipe)
    }
    h->feats.alloc((size_t)std::max<long long>(V, 1) * D);
    hipLaunchKernelGGL(k_feats_final, dim3(cdiv((size_t)V * D, 256)), dim3(256), 0, s, (const float*)h->sum.p,
                       (const unsigned*)h->cnt.p, V, D, h->feats.p);
    HMSG_CHECK_LAUNCH();
    HIP_TRY(hipStreamSynchronize(s));
    h->feats_final = true;
}

// hand-over of the encoder outputs of frames [first, first+n): membership bitsets + F_p tables
void hmsg_bitset_and_fp(hmsg_ctx* h, int first, int n, int M, const unsigned char* d_masks, const float* d_fg,
                        const float* d_fm, const float* d_fc, const int* d_nmask) {
    const size_t HW = (size_t)h->cfg.height * h->cfg.width;
    const int D = h->cfg.feat_dim;
    const size_t chunks = (HW + 15) / 16;
    {
        ProfScope ps(h->prof, h->stream, "k_bitset", (double)n * (double)HW * (M + 8.0 * h->NW));
        hipLaunchKernelGGL(k_bitset, dim3(cdiv(chunks * n, 256)), dim3(256), 0, h->stream, d_masks, M, HW, n, (size_t)M * HW,
                           d_nmask, h->NW, h->bits.p + (size_t)first * HW * h->NW);
    }
    HMSG_CHECK_LAUNCH();
    const float wm = (float)h->cfg.clip_masked_weight, wc = (float)(1.0 - h->cfg.clip_masked_weight);
    hipLaunchKernelGGL(k_fp, dim3(n), dim3(256), 0, h->stream, d_fg, d_fm, d_fc, M, D, wm, wc, d_nmask, h->MS,
                       h->fp.p + (size_t)first * h->MS * D);
    HMSG_CHECK_LAUNCH();
}
