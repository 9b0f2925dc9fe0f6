// A3 + A4 + A5: per-frame feature fusion math, exact nearest-voxel snapping, last-writer-wins feature
// accumulation and the 3-D mask clouds.
//
// Reference: perception/models/sam_clip_feats_extractor.py:159-191 (fusion math),
// memory/hmsg/graph/graph.py:373-415 (loop B), memory/hmsg/dataloader/generic.py:140-190
// (create_3d_masks).
//
// MI355X design notes
//  * The reference materialises a per-pixel fp16 feature image (315 MB/frame at D=512).  A pixel's
//    feature depends only on WHICH masks cover it, so we keep a 64-bit mask-membership word per pixel
//    and the M x D table F_p per frame, and rebuild the (fp16-rounded) row only for the one pixel per
//    voxel per frame that torch's duplicate-index `+=` actually keeps (graph.py:410).
//  * Frames are processed 64 at a time.  stamp[v][j] = 1 + the largest pixel index of frame j (of the
//    batch) whose nearest voxel is v: a wave then owns ONE voxel, ballots its 64 stamps and adds the
//    frames' contributions in frame order in registers -- the same float32 addition order as the
//    reference's sequential loop, and one read-modify-write of the voxel's row per 64 frames instead
//    of one per frame.
//  * Nearest voxel: ring expansion over the occupancy bitmap (z-columns are contiguous bits) with an
//    exact termination bound; no distance cap (graph.py:409, generic.py:181).
#include "hmsg_common.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <cmath>

#define FB 64                       /* frames per stamp batch = wave width */
#define MFIX_SCALE 17592186044416.0  /* 2^44 fixed point for (pixel-count weighted) mask-cloud centroids */

// ------------------------------------------------------------------------------------------ K_bitset
// masks u8 [M][HW] of one frame -> bits u64 [HW]; 16 pixels per thread, 16-byte loads.
__global__ void k_bitset(const unsigned char* __restrict__ masks, int M, size_t HW, int nfr, size_t mask_stride_frame,
                         unsigned long long* __restrict__ bits) {
    const size_t chunks = (HW + 15) / 16;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= chunks * nfr) return;
    int f = (int)(t / chunks);
    size_t p0 = (t - (size_t)f * chunks) * 16;
    const unsigned char* mf = masks + (size_t)f * mask_stride_frame;
    unsigned long long b[16];
    for (int j = 0; j < 16; ++j) b[j] = 0ull;
    if (p0 + 16 <= HW && (HW & 15) == 0) {
        for (int i = 0; i < M; ++i) {
            uint4 v = *reinterpret_cast<const uint4*>(mf + (size_t)i * HW + p0);
            unsigned w[4] = {v.x, v.y, v.z, v.w};
            for (int j = 0; j < 16; ++j) {
                unsigned byte = (w[j >> 2] >> ((j & 3) * 8)) & 0xffu;
                b[j] |= (unsigned long long)(byte != 0) << i;
            }
        }
    } else {
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < 16 && p0 + j < HW; ++j) b[j] |= (unsigned long long)(mf[(size_t)i * HW + p0 + j] != 0) << i;
    }
    unsigned long long* o = bits + (size_t)f * HW + p0;
    for (int j = 0; j < 16 && p0 + j < HW; ++j) o[j] = b[j];
}

// ------------------------------------------------------------------------------------------ K_fp
// sam_clip_feats_extractor.py:159-175.  One 256-thread block per frame; a wave per mask row.
__global__ void k_fp(const float* __restrict__ Fg, const float* __restrict__ Fm, const float* __restrict__ Fc, int M, int D,
                     float wm, float wc, float* __restrict__ Fp /*[nfr][M][D]*/) {
    __shared__ float phi[64];
    __shared__ float wsm[64];
    const int f = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float* g = Fg + (size_t)f * D;
    float gn = 0.f;
    for (int e = lane; e < D; e += 64) gn += g[e] * g[e];
    gn = fmaxf(__fsqrt_rn(wave_sum_f32(gn)), 1e-6f);
    for (int i = wv; i < M; i += 4) {
        const float* a = Fm + ((size_t)f * M + i) * D;
        const float* b = Fc + ((size_t)f * M + i) * D;
        float* o = Fp + ((size_t)f * M + i) * D;
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
        for (int i = 0; i < M; ++i) mx = fmaxf(mx, phi[i]);
        float s = 0.f;
        for (int i = 0; i < M; ++i) {
            wsm[i] = expf(phi[i] - mx);
            s += wsm[i];
        }
        for (int i = 0; i < M; ++i) wsm[i] = __fdiv_rn(wsm[i], s);
    }
    __syncthreads();
    for (int i = wv; i < M; i += 4) {
        float* o = Fp + ((size_t)f * M + i) * D;
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

__global__ void k_nn_stamp(const unsigned short* __restrict__ depth, const double* __restrict__ pose, CamK cam, float scale,
                           int H, int W, int f0, int nfr, NNIndex I, int* __restrict__ nn, unsigned* __restrict__ stamp /*[V][FB]*/) {
    const size_t HW = (size_t)H * W;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = t < HW * nfr;
    if (!in_range) t = HW * nfr - 1;
    int fl = (int)(t / HW);
    int p = (int)(t - (size_t)fl * HW);
    int f = f0 + fl;
    int y = p / W, x = p - y * W;
    double wx, wy, wz;
    int idx = -1;
    if (in_range && backproject(depth[(size_t)f * HW + p], x, y, cam, scale, pose + (size_t)f * 16, wx, wy, wz))
        idx = nn_search(I, wx, wy, wz);
    if (in_range) nn[(size_t)f * HW + p] = idx;
    // stamp = LARGEST pixel index per (voxel, frame): within a run of consecutive lanes snapping to the same
    // voxel only the last lane can win, so only it pays for the atomic (runs are ~8 pixels long)
    const int lane = threadIdx.x & 63;
    long long key = in_range && idx >= 0 ? ((long long)idx << 8) | fl : -1 - lane;
    long long nxt = __shfl_down(key, 1);
    if (key >= 0 && (lane == 63 || nxt != key)) atomicMax(&stamp[(size_t)idx * FB + fl], (unsigned)p + 1u);
}

// ------------------------------------------------------------------------------------------ K_fuse (A5)
// One wave per voxel.  graph.py:410-411 with torch's last-writer-wins semantics (SURVEY hazard 7): per
// frame the voxel receives old + fp16(F_2D[p*]) for p* = its largest pixel index, counter += 1.
template <int VEC, int NJ>
__global__ void k_fuse(const unsigned* __restrict__ stamp, long long V, int f0, int nfr, int M, int D, size_t HW,
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
        unsigned long long b = bits[(size_t)f * HW + p];
        const float* fpf = Fp + (size_t)f * M * D;
        float x[NJ * VEC];
#pragma unroll
        for (int q = 0; q < NJ * VEC; ++q) x[q] = 0.f;
        while (b) {
            int i = __ffsll(b) - 1;
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
    long long word_off;     // first bitmap word of this mask in the sub-batch bitmap
};

struct Winner {
    int f;   // frame index within the sub-batch
    int p;
    int v;
};

__global__ void k_mcount(const int* __restrict__ nn, const unsigned long long* __restrict__ bits, size_t HW, int f0, int nfr,
                         long long V, int M, unsigned* __restrict__ mcount /*[nfr][V][M]*/) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = t < HW * nfr;
    if (!in_range) t = HW * nfr - 1;
    int fl = (int)(t / HW);
    size_t g = (size_t)(f0 + fl) * HW + (t - (size_t)fl * HW);
    int v = in_range ? nn[g] : -1;
    unsigned long long b = v >= 0 ? bits[g] : 0ull;
    // runs of consecutive lanes with the same (frame, voxel, mask set): the last lane adds the run length
    const int lane = threadIdx.x & 63;
    long long k1 = v >= 0 ? ((long long)v << 8) | fl : -1 - lane;
    long long p1 = __shfl_up(k1, 1);
    unsigned long long pb = __shfl_up(b, 1);
    const bool head = lane == 0 || p1 != k1 || pb != b;
    unsigned long long heads = __ballot(head);
    long long n1 = __shfl_down(k1, 1);
    unsigned long long nb = __shfl_down(b, 1);
    const bool tail = lane == 63 || n1 != k1 || nb != b;
    if (v >= 0 && tail && b) {
        unsigned long long below = heads & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
        int start_lane = 63 - __clzll(below);
        unsigned len = (unsigned)(lane - start_lane + 1);
        unsigned* row = mcount + ((size_t)fl * V + v) * M;
        while (b) {
            int i = __ffsll(b) - 1;
            b &= b - 1;
            atomicAdd(&row[i], len);
        }
    }
}

// The pixels that define the 3-D masks: for every (voxel, frame) the LAST pixel that snapped to the voxel
// (create_3d_masks keeps one map point per voxel hit, weighted by its pixel count).  They are exactly the
// non-zero stamps, so the list is read off the stamp table with coalesced loads -- testing every pixel against
// its voxel's stamp was a random 4-byte gather per pixel.
__global__ void k_winners(const unsigned* __restrict__ stamp, long long V, int j0, int nfr, Winner* __restrict__ out,
                          unsigned* __restrict__ n_out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // t = v * FB + j, rows padded to FB
    const int j = (int)(t % FB);
    const long long v = t / FB;
    unsigned sv = 0u;
    if (v < V && j >= j0 && j < j0 + nfr) sv = stamp[t];
    const bool win = sv != 0u;
    // list slots per wave (one atomic on the counter per wave)
    const unsigned long long m = __ballot(win);
    if (!m) return;
    const int lane = threadIdx.x & 63, leader = __ffsll(m) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(n_out, (unsigned)__popcll(m));
    base = __shfl(base, leader);
    if (win) out[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = Winner{j - j0, (int)(sv - 1u), (int)v};
}

__global__ void k_mbounds(const Winner* __restrict__ win, unsigned nwin, long long V, int M, const unsigned* __restrict__ mcount,
                          const double* __restrict__ pts, unsigned long long* __restrict__ bounds /*[nfr*M][6]*/) {
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nwin) return;
    Winner w = win[t];
    const unsigned* row = mcount + ((size_t)w.f * V + w.v) * M;
    unsigned long long e[3] = {enc_f64(pts[(size_t)w.v * 3]), enc_f64(pts[(size_t)w.v * 3 + 1]), enc_f64(pts[(size_t)w.v * 3 + 2])};
    for (int i = 0; i < M; ++i) {
        if (!row[i]) continue;
        unsigned long long* b = bounds + ((size_t)w.f * M + i) * 6;
        for (int a = 0; a < 3; ++a) {
            if (e[a] < b[a]) atomicMin(&b[a], e[a]);
            if (e[a] > b[3 + a]) atomicMax(&b[3 + a], e[a]);
        }
    }
}

__device__ __forceinline__ long long mask_cell(const MaskGeom& mg, double vs, const double* __restrict__ p, int& ix, int& iy,
                                               int& iz) {
    ix = (int)floor(__ddiv_rn(__dsub_rn(p[0], mg.ox), vs));
    iy = (int)floor(__ddiv_rn(__dsub_rn(p[1], mg.oy), vs));
    iz = (int)floor(__ddiv_rn(__dsub_rn(p[2], mg.oz), vs));
    return ((long long)ix * mg.ny + iy) * mg.nz + iz;
}

__global__ void k_mmark(const Winner* __restrict__ win, unsigned nwin, long long V, int M, const unsigned* __restrict__ mcount,
                        const double* __restrict__ pts, const MaskGeom* __restrict__ geom, double vs,
                        unsigned long long* __restrict__ mbitmap) {
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nwin) return;
    Winner w = win[t];
    const unsigned* row = mcount + ((size_t)w.f * V + w.v) * M;
    for (int i = 0; i < M; ++i) {
        if (!row[i]) continue;
        const MaskGeom mg = geom[(size_t)w.f * M + i];
        int ix, iy, iz;
        long long lin = mask_cell(mg, vs, pts + (size_t)w.v * 3, ix, iy, iz);
        atomicOr(&mbitmap[mg.word_off + (lin >> 6)], 1ull << (lin & 63));
    }
}

struct MaskAcc {
    long long* sx;
    long long* sy;
    long long* sz;
    unsigned long long* wgt;
};

__global__ void k_maccum(const Winner* __restrict__ win, unsigned nwin, long long V, int M, unsigned* __restrict__ mcount,
                         const double* __restrict__ pts, const MaskGeom* __restrict__ geom, double vs,
                         const unsigned long long* __restrict__ mbitmap, const unsigned* __restrict__ mrank, MaskAcc acc) {
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nwin) return;
    Winner w = win[t];
    unsigned* row = mcount + ((size_t)w.f * V + w.v) * M;
    const double* p = pts + (size_t)w.v * 3;
    for (int i = 0; i < M; ++i) {
        unsigned c = row[i];
        if (!c) continue;
        row[i] = 0u;   // leave the dense counter table clean for the next sub-batch
        const MaskGeom mg = geom[(size_t)w.f * M + i];
        int ix, iy, iz;
        long long lin = mask_cell(mg, vs, p, ix, iy, iz);
        long long wd = mg.word_off + (lin >> 6);
        unsigned long long word = mbitmap[wd];
        unsigned slot = mrank[wd] + (unsigned)__popcll(word & ((1ull << (lin & 63)) - 1ull));
        double cx = __dadd_rn(mg.ox, __dmul_rn((double)ix, vs));
        double cy = __dadd_rn(mg.oy, __dmul_rn((double)iy, vs));
        double cz = __dadd_rn(mg.oz, __dmul_rn((double)iz, vs));
        long long qx = (long long)llrint((p[0] - cx) * MFIX_SCALE) * (long long)c;
        long long qy = (long long)llrint((p[1] - cy) * MFIX_SCALE) * (long long)c;
        long long qz = (long long)llrint((p[2] - cz) * MFIX_SCALE) * (long long)c;
        atomicAdd((unsigned long long*)&acc.sx[slot], (unsigned long long)qx);
        atomicAdd((unsigned long long*)&acc.sy[slot], (unsigned long long)qy);
        atomicAdd((unsigned long long*)&acc.sz[slot], (unsigned long long)qz);
        atomicAdd(&acc.wgt[slot], (unsigned long long)c);
    }
}

// one thread per bitmap word of the sub-batch: emit the points of its set bits
__global__ void k_mfinal(const unsigned long long* __restrict__ mbitmap, const unsigned* __restrict__ mrank, long long nwords,
                         const MaskGeom* __restrict__ geom, int nmasks, double vs, MaskAcc acc, double* __restrict__ out) {
    long long wd = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (wd >= nwords) return;
    unsigned long long bitsw = mbitmap[wd];
    if (!bitsw) return;
    int lo = 0, hi = nmasks - 1;   // last mask with word_off <= wd (empty masks share an offset: take the last)
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (geom[mid].word_off <= wd) lo = mid; else hi = mid - 1;
    }
    const MaskGeom mg = geom[lo];
    unsigned s = mrank[wd];
    while (bitsw) {
        int b = __ffsll(bitsw) - 1;
        bitsw &= bitsw - 1;
        long long lin = (wd - mg.word_off) * 64 + b;
        int iz = (int)(lin % mg.nz);
        long long r = lin / mg.nz;
        int iy = (int)(r % mg.ny);
        int ix = (int)(r / mg.ny);
        double n = (double)acc.wgt[s];
        out[(size_t)s * 3 + 0] = __dadd_rn(mg.ox, __dmul_rn((double)ix, vs)) + ((double)acc.sx[s] / n) / MFIX_SCALE;
        out[(size_t)s * 3 + 1] = __dadd_rn(mg.oy, __dmul_rn((double)iy, vs)) + ((double)acc.sy[s] / n) / MFIX_SCALE;
        out[(size_t)s * 3 + 2] = __dadd_rn(mg.oz, __dmul_rn((double)iz, vs)) + ((double)acc.sz[s] / n) / MFIX_SCALE;
        ++s;
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
                       h->M, h->cfg.feat_dim, HW, (const unsigned long long*)h->bits.p, (const float*)h->fp.p, h->sum.p,
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
    const int H = c.height, W = c.width, D = c.feat_dim, M = h->M;
    const size_t HW = (size_t)H * W;
    const long long V = h->V;
    const float scale = (float)c.depth_scale;
    // create_3d_masks' filter_distance (generic.py:126-127): mean camera depth of a mask can never exceed
    // the u16 depth range, so any threshold >= 65.536 m (the shipped 10000) never fires.
    HMSG_REQUIRE(c.max_mask_distance * c.depth_scale >= 65536.0, HMSG_ERR_UNSUPPORTED,
                 "max_mask_distance below the representable depth range is not implemented in this build");
    if (h->sum.n < (size_t)V * D || h->n_fused == 0) {
        h->sum.alloc((size_t)V * D);
        h->cnt.alloc((size_t)V);
        h->sum.zero(s);
        h->cnt.zero(s);
    }
    if (h->nn.n < (size_t)c.max_frames * HW) h->nn.alloc((size_t)c.max_frames * HW);
    DevBuf<unsigned> stamp;
    stamp.alloc((size_t)std::max<long long>(V, 1) * FB);
    // mask sub-batch size from an 8 GiB budget for the dense per-voxel mask counters (288 GB of HBM: a whole
    // stamp batch at once for maps up to ~1M voxels; every sub-batch costs three host round trips)
    int Bm = FB;
    while (Bm > 1 && (size_t)Bm * V * M * 4 > ((size_t)8 << 30)) Bm >>= 1;
    DevBuf<unsigned> mcount;
    mcount.alloc((size_t)Bm * std::max<long long>(V, 1) * M);
    mcount.zero(s);
    DevBuf<Winner> win;
    win.alloc((size_t)Bm * HW);
    DevBuf<unsigned> d_nwin;
    d_nwin.alloc(1);
    DevBuf<unsigned long long> d_bounds;
    d_bounds.alloc((size_t)Bm * M * 6);
    DevBuf<MaskGeom> d_geom;
    d_geom.alloc((size_t)Bm * M);
    DevBuf<unsigned long long> mbitmap;
    DevBuf<unsigned> mrank;
    DevBuf<long long> macc_xyz;
    DevBuf<unsigned long long> macc_w;
    DevBuf<long long> d_offidx;
    DevBuf<unsigned> d_offval;
    d_offidx.alloc((size_t)Bm * M);
    d_offval.alloc((size_t)Bm * M);
    std::vector<unsigned long long> hb((size_t)Bm * M * 6);
    std::vector<MaskGeom> hg((size_t)Bm * M);
    std::vector<long long> hoffidx((size_t)Bm * M);
    std::vector<unsigned> hoffval((size_t)Bm * M);
    if (h->masks3d.off.empty()) h->masks3d.off.assign(1, 0);

    for (int fb0 = h->n_fused; fb0 < h->n_feat_frames; fb0 += FB) {
        const int nb = std::min(FB, h->n_feat_frames - fb0);
        stamp.zero(s);
        {
            ProfScope ps(h->prof, s, "k_nn_stamp", (double)nb * ((double)HW * 10.0 + 128.0));
            hipLaunchKernelGGL(k_nn_stamp, dim3(cdiv(HW * nb, 256)), dim3(256), 0, s, (const unsigned short*)h->depth.p,
                               (const double*)h->pose.p, h->cam, scale, H, W, fb0, nb, hmsg_nn_index(h), h->nn.p, stamp.p);
        }
        HMSG_CHECK_LAUNCH();
        {
            ProfScope ps(h->prof, s, "k_fuse", (double)V * 256.0);
            dispatch_fuse(h, stamp.p, fb0, nb);
        }
        // ---- 3-D masks, Bm frames at a time
        for (int f0 = fb0; f0 < fb0 + nb; f0 += Bm) {
            const int nfr = std::min(Bm, fb0 + nb - f0);
            const int nmask = nfr * M;
            {
                ProfScope ps(h->prof, s, "k_mcount", (double)nfr * (double)HW * 20.0);
                hipLaunchKernelGGL(k_mcount, dim3(cdiv(HW * nfr, 256)), dim3(256), 0, s, (const int*)h->nn.p,
                                   (const unsigned long long*)h->bits.p, HW, f0, nfr, V, M, mcount.p);
            }
            HMSG_CHECK_LAUNCH();
            HIP_TRY(hipMemsetAsync(d_nwin.p, 0, 4, s));
            hipLaunchKernelGGL(k_winners, dim3(cdiv((size_t)V * FB, 256)), dim3(256), 0, s, (const unsigned*)stamp.p, V,
                               f0 - fb0, nfr, win.p, d_nwin.p);
            HMSG_CHECK_LAUNCH();
            for (int i = 0; i < nmask; ++i)
                for (int a = 0; a < 6; ++a) hb[(size_t)i * 6 + a] = a < 3 ? ~0ull : 0ull;
            HIP_TRY(hipMemcpyAsync(d_bounds.p, hb.data(), (size_t)nmask * 48, hipMemcpyHostToDevice, s));
            unsigned nwin = 0;
            HIP_TRY(hipMemcpyAsync(&nwin, d_nwin.p, 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (nwin) {
                hipLaunchKernelGGL(k_mbounds, dim3(cdiv(nwin, 256)), dim3(256), 0, s, (const Winner*)win.p, nwin, V, M,
                                   (const unsigned*)mcount.p, (const double*)h->pts.p, d_bounds.p);
                HMSG_CHECK_LAUNCH();
            }
            HIP_TRY(hipMemcpyAsync(hb.data(), d_bounds.p, (size_t)nmask * 48, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            long long nwords = 0;
            for (int i = 0; i < nmask; ++i) {
                MaskGeom& mg = hg[i];
                mg.word_off = nwords;
                mg.pad = 0;
                if (hb[(size_t)i * 6] == ~0ull) {   // no valid pixel in this mask
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
            if (nwin && nwords) {
                HIP_TRY(hipMemcpyAsync(d_geom.p, hg.data(), (size_t)nmask * sizeof(MaskGeom), hipMemcpyHostToDevice, s));
                mbitmap.ensure((size_t)nwords);
                mrank.ensure((size_t)nwords);
                HIP_TRY(hipMemsetAsync(mbitmap.p, 0, (size_t)nwords * 8, s));
                hipLaunchKernelGGL(k_mmark, dim3(cdiv(nwin, 256)), dim3(256), 0, s, (const Winner*)win.p, nwin, V, M,
                                   (const unsigned*)mcount.p, (const double*)h->pts.p, (const MaskGeom*)d_geom.p,
                                   c.voxel_size, mbitmap.p);
                HMSG_CHECK_LAUNCH();
                npts = (long long)hmsg_bitmap_rank(mbitmap.p, mrank.p, (size_t)nwords, s, h->scan_tmp);
                macc_xyz.ensure((size_t)npts * 3);
                macc_w.ensure((size_t)npts);
                HIP_TRY(hipMemsetAsync(macc_xyz.p, 0, (size_t)npts * 3 * 8, s));
                HIP_TRY(hipMemsetAsync(macc_w.p, 0, (size_t)npts * 8, s));
                MaskAcc acc{macc_xyz.p, macc_xyz.p + npts, macc_xyz.p + 2 * npts, macc_w.p};
                hipLaunchKernelGGL(k_maccum, dim3(cdiv(nwin, 256)), dim3(256), 0, s, (const Winner*)win.p, nwin, V, M,
                                   mcount.p, (const double*)h->pts.p, (const MaskGeom*)d_geom.p, c.voxel_size,
                                   (const unsigned long long*)mbitmap.p, (const unsigned*)mrank.p, acc);
                HMSG_CHECK_LAUNCH();
                // append to the resident mask-cloud store
                size_t need = (size_t)(h->masks3d.total + npts) * 3;
                if (need > h->masks3d.pts.n) {
                    DevBuf<double> bigger;
                    bigger.alloc(std::max(need * 2, (size_t)1 << 20));
                    if (h->masks3d.total)
                        HIP_TRY(hipMemcpyAsync(bigger.p, h->masks3d.pts.p, (size_t)h->masks3d.total * 24,
                                               hipMemcpyDeviceToDevice, s));
                    HIP_TRY(hipStreamSynchronize(s));
                    bigger.swap(h->masks3d.pts);
                }
                hipLaunchKernelGGL(k_mfinal, dim3(cdiv((size_t)nwords, 256)), dim3(256), 0, s,
                                   (const unsigned long long*)mbitmap.p, (const unsigned*)mrank.p, nwords,
                                   (const MaskGeom*)d_geom.p, nmask, c.voxel_size, acc,
                                   h->masks3d.pts.p + (size_t)h->masks3d.total * 3);
                HMSG_CHECK_LAUNCH();
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
                std::vector<long long> start(nmask + 1, npts);
                int k = 0;
                for (int i = 0; i < nmask; ++i)
                    if (hg[i].nx) start[i] = hoffval[k++];
                for (int i = nmask - 1; i >= 0; --i)
                    if (!hg[i].nx) start[i] = start[i + 1];
                for (int i = 0; i < nmask; ++i) h->masks3d.off.push_back(h->masks3d.total + start[i + 1]);
            } else {
                for (int i = 0; i < nmask; ++i) h->masks3d.off.push_back(h->masks3d.total);
            }
            h->masks3d.total += npts;
        }
        h->n_fused = fb0 + nb;
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
                        const float* d_fm, const float* d_fc) {
    const size_t HW = (size_t)h->cfg.height * h->cfg.width;
    const int D = h->cfg.feat_dim;
    const size_t chunks = (HW + 15) / 16;
    {
        ProfScope ps(h->prof, h->stream, "k_bitset", (double)n * (double)HW * (M + 8.0));
        hipLaunchKernelGGL(k_bitset, dim3(cdiv(chunks * n, 256)), dim3(256), 0, h->stream, d_masks, M, HW, n, (size_t)M * HW,
                           h->bits.p + (size_t)first * HW);
    }
    HMSG_CHECK_LAUNCH();
    const float wm = (float)h->cfg.clip_masked_weight, wc = (float)(1.0 - h->cfg.clip_masked_weight);
    hipLaunchKernelGGL(k_fp, dim3(n), dim3(256), 0, h->stream, d_fg, d_fm, d_fc, M, D, wm, wc,
                       h->fp.p + (size_t)first * M * D);
    HMSG_CHECK_LAUNCH();
}
