// C ABI of libhmsg (see include/hmsg.h).  Thin: argument checks, host<->HBM staging, error capture.
#include "hmsg_common.h"
#include "hmsg_ckdtree.h"
#include "hmsg_nn.h"

#include <algorithm>
#include <cstring>

void hmsg_bitset_and_fp(hmsg_ctx* h, int first, int n, int M, const unsigned char* d_masks, const float* d_fg,
                        const float* d_fm, const float* d_fc, const int* d_nmask);   // hmsg_fuse.hip

// ---- bit-equal nearest-neighbour ties: host side ----------------------------------------------------------------
__global__ void k_nn_patch(const long long* __restrict__ qid, const int* __restrict__ val, unsigned n, int* __restrict__ target) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) target[qid[i]] = val[i];
}

void hmsg_kd_join(hmsg_ctx* h) {
    if (h->kd_thread.joinable()) h->kd_thread.join();
}

void hmsg_kd_start(hmsg_ctx* h) {
    hmsg_kd_join(h);
    h->host_pts.resize((size_t)h->V * 3);
    if (h->V) HIP_TRY(hipMemcpyAsync(h->host_pts.data(), h->pts.p, (size_t)h->V * 24, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->kd = std::make_shared<CKDTree>();
    std::shared_ptr<CKDTree> kd = h->kd;
    const double* pts = h->host_pts.data();
    const long long V = h->V;
    h->kd_thread = std::thread([kd, pts, V] { kd->build(pts, V); });
}

bool hmsg_resolve_ties(hmsg_ctx* h, TieBuf& tb, int* target) {
    unsigned n = 0;
    HIP_TRY(hipMemcpyAsync(&n, tb.count.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (n == 0) return true;
    if (n > tb.cap) {                       // list overflowed: grow, the caller repeats the search
        tb.cap = 0;
        tb.prepare(h->stream, n + n / 2 + 1024);
        return false;
    }
    std::vector<TieRec> recs(n);
    HIP_TRY(hipMemcpy(recs.data(), tb.recs.p, (size_t)n * sizeof(TieRec), hipMemcpyDeviceToHost));
    hmsg_kd_join(h);
    std::vector<long long> q(n);
    std::vector<int> v(n);
    for (unsigned i = 0; i < n; ++i) {
        const double x[3] = {recs[i].x, recs[i].y, recs[i].z};
        q[i] = recs[i].qid;
        v[i] = (int)h->kd->query1(x);
    }
    tb.pq.ensure(n);
    tb.pv.ensure(n);
    HIP_TRY(hipMemcpyAsync(tb.pq.p, q.data(), (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(tb.pv.p, v.data(), (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_nn_patch, dim3(cdiv(n, 256)), dim3(256), 0, h->stream, (const long long*)tb.pq.p, (const int*)tb.pv.p, n,
                       target);
    HMSG_CHECK_LAUNCH();
    HIP_TRY(hipStreamSynchronize(h->stream));     // (q / v are host temporaries)
    h->n_tie_queries += n;
    return true;
}

namespace {

bool is_device_ptr(const void* p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

void copy_in(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, is_device_ptr(src) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
}

template <typename F>
int guard(hmsg_ctx* h, F&& fn) {
    try {
        if (h) HIP_TRY(hipSetDevice(h->cfg.device_id));
        fn();
        return HMSG_OK;
    } catch (const hmsg_error& e) {
        if (h) h->err = e.msg;
        return e.code;
    } catch (const std::exception& e) {
        if (h) h->err = e.what();
        return HMSG_ERR_INVALID;
    } catch (...) {      // nothing may cross the C boundary
        if (h) h->err = "unknown error";
        return HMSG_ERR_INVALID;
    }
}

// A long episode's frame store (colour, depth, mask bitsets, nearest-voxel indices: 17 B per pixel and frame, 157 GB for
// 10 000 frames at 1280x720) is dead weight once every frame is fused: nothing after the fusion reads it, and the merge
// needs the room.  Small stores stay (a service that rebuilds scenes reuses them).
// The frame store of a handle: colour + depth at creation, mask bitsets and nearest-voxel indices when they are first
// needed -- four blocks; or, when all four together pass 96 GB (a long episode), ONE block with the four as views into
// it: handed back before the merge it is a single parked block that the merge's arenas are carved from (DevCache).
static const size_t FRAME_STORE_LARGE = (size_t)96 << 30;
void alloc_frame_store(hmsg_ctx* h) {
    const size_t HW = (size_t)h->cfg.height * h->cfg.width, F = (size_t)h->cfg.max_frames, NW = (size_t)h->NW;
    const size_t b_rgb = HW * 3 * F, b_depth = HW * 2 * F, b_bits = HW * NW * 8 * F, b_nn = HW * 4 * F;
    auto up = [](size_t v) { return (v + ((size_t)1 << 21) - 1) & ~(((size_t)1 << 21) - 1); };
    if (b_rgb + b_depth + b_bits + b_nn >= FRAME_STORE_LARGE) {
        h->frame_arena.alloc(up(b_rgb) + up(b_depth) + up(b_bits) + up(b_nn));
        unsigned char* p = h->frame_arena.p;
        h->rgb.view(p, HW * 3 * F);
        p += up(b_rgb);
        h->depth.view((unsigned short*)p, HW * F);
        p += up(b_depth);
        h->bits.view((unsigned long long*)p, HW * NW * F);
        p += up(b_bits);
        h->nn.view((int*)p, HW * F);
    } else {
        h->rgb.alloc(HW * 3 * F);
        h->depth.alloc(HW * F);
    }
}

void release_frame_store_if_large(hmsg_ctx* h) {
    const size_t bytes = h->rgb.bytes() + h->depth.bytes() + h->bits.bytes() + h->nn.bytes();
    // (handing 60 GB back to the driver costs seconds: only when the merge could not fit next to it)
    if (bytes < FRAME_STORE_LARGE || h->n_fused < h->n_feat_frames) return;
    // (parked in the allocator's cache, not handed back to the driver: hipFree of 150 GB takes seconds; the merge's
    //  arenas re-use the blocks that fit, and the cache frees the rest when an allocation does not fit)
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->rgb.release();
    h->depth.release();
    h->bits.release();
    h->nn.release();
    h->frame_arena.release();      // (arena mode: the four were views, this parks the one block)
    h->frames_released = true;
    // (a fold running on its worker thread allocates from its own cache: hand the parked blocks back to the driver)
    if (h->fold_pipe) dev_cache().trim();
}

}  // namespace

extern "C" {

const char* hmsg_version(void) { return "hmsg-mi355x 0.1 (gfx950)"; }

void hmsg_default_config(hmsg_config* c) {
    memset(c, 0, sizeof(*c));
    c->device_id = 0;
    c->feat_dim = 512;
    c->height = 480;
    c->width = 640;
    c->max_frames = 1024;
    c->max_masks = 64;       /* up to 256 (SAM points_per_side=12 yields at most 144) */
    c->voxel_size = 0.05;
    c->depth_scale = 1000.0;
    c->init_overlap_thresh = 0.75;
    c->overlap_thresh_factor = 0.025;
    c->iou_thresh = 0.05;
    c->clip_masked_weight = 0.4418;
    c->max_mask_distance = 10000.0;
    c->merge_type = HMSG_MERGE_SEQUENTIAL;
    c->outlier_nb_points = 1000;
    c->outlier_radius = 1.0;
    c->pool_max_dist = 0.8;
    c->feat_dbscan_eps = 0.01;
    c->feat_dbscan_min = 100;
    c->merge_dbscan_eps = 0.1;
    c->merge_dbscan_min = 10;
    c->min_instance_points = 10;
    c->skip_frames = 1;
    c->depth_cut = 0.0;
    c->grid_resolution = 0.05;
    c->overlap_distance_form = HMSG_OVERLAP_DIRECT;
}

size_t hmsg_config_size(void) { return sizeof(hmsg_config); }

int hmsg_create(const hmsg_config* cfg, hmsg_t** out) {
    if (!cfg || !out) return HMSG_ERR_INVALID;
    *out = nullptr;
    if (cfg->feat_dim <= 0 || cfg->height <= 0 || cfg->width <= 0 || cfg->max_frames <= 0 || cfg->voxel_size <= 0 ||
        cfg->max_masks <= 0 || cfg->max_masks > 256)
        return HMSG_ERR_INVALID;
    hmsg_ctx* h = new hmsg_ctx();
    h->cfg = *cfg;
    h->NW = (cfg->max_masks + 63) / 64;
    h->MS = h->NW * 64;
    int rc = guard(h, [&] {
        int ndev = 0;
        HIP_TRY(hipGetDeviceCount(&ndev));
        HMSG_REQUIRE(cfg->device_id >= 0 && cfg->device_id < ndev, HMSG_ERR_INVALID, "device_id out of range");
        HIP_TRY(hipSetDevice(cfg->device_id));
        HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        alloc_frame_store(h);
        h->pose.alloc((size_t)16 * cfg->max_frames);
    });
    if (rc != HMSG_OK) {
        fprintf(stderr, "hmsg_create: %s\n", h->err.c_str());
        delete h;
        return rc;
    }
    *out = h;
    return HMSG_OK;
}

void hmsg_destroy(hmsg_t* h) {
    if (!h) return;
    hmsg_kd_join(h);
    (void)hipSetDevice(h->cfg.device_id);
    try {
        hmsg_fold_pipe_abort(h);
    } catch (...) {
    }
    h->fold_cache.trim();
    if (h->stream) {
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamDestroy(h->stream);
    }
    delete h;
}

const char* hmsg_last_error(const hmsg_t* h) { return h ? h->err.c_str() : "null handle"; }

void hmsg_release_cached_memory(void) { dev_cache().trim(); }

int hmsg_reset(hmsg_t* h) {
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HIP_TRY(hipStreamSynchronize(h->stream));
        hmsg_kd_join(h);
        hmsg_fold_pipe_abort(h);
        h->n_tie_queries = 0;
        h->n_frames = h->n_feat_frames = h->n_fused = 0;
        h->n_offered = 0;
        h->room_n = 0;
        h->room_total = 0;
        if (h->frames_released) {              // (hmsg_merge_instances gave a very large frame store back)
            alloc_frame_store(h);
            h->frames_released = false;
        }
        h->nmask.clear();
        h->mask_first.clear();
        h->have_K = false;
        h->map_ready = h->feats_final = h->merged = h->pooled = h->inst_denoised = h->tree_partial = false;
        h->frame_window = 0;
        h->nodes.clear();
        h->V = h->V0 = 0;
        h->masks3d.off.clear();
        h->masks3d.total = 0;
        h->inst.off.clear();
        h->inst.total = 0;
        h->prof.clear();
    });
}

int hmsg_set_profiling(hmsg_t* h, int32_t on) {
    if (!h) return HMSG_ERR_INVALID;
    h->prof.enabled = on != 0;
    h->prof.detail = on >= 2;
    return HMSG_OK;
}

static void prof_aggregate(hmsg_ctx* h, std::vector<std::string>& names, std::vector<long long>& cnt, std::vector<double>& ms,
                           std::vector<double>* work = nullptr) {
    (void)hipStreamSynchronize(h->stream);
    for (auto& e : h->prof.ev) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, e.a, e.b) != hipSuccess) continue;
        size_t k = 0;
        for (; k < names.size(); ++k)
            if (names[k] == e.name) break;
        if (k == names.size()) {
            names.push_back(e.name);
            cnt.push_back(0);
            ms.push_back(0.0);
            if (work) work->push_back(0.0);
        }
        cnt[k]++;
        ms[k] += t;
        if (work) (*work)[k] += e.work;
    }
}

int32_t hmsg_profile_count(hmsg_t* h) {
    if (!h) return 0;
    std::vector<std::string> n;
    std::vector<long long> c;
    std::vector<double> m;
    prof_aggregate(h, n, c, m);
    return (int32_t)n.size();
}

int hmsg_profile_entry(hmsg_t* h, int32_t i, char* name, int64_t* launches, double* total_ms, double* total_work) {
    if (!h || !name || !launches || !total_ms) return HMSG_ERR_INVALID;
    std::vector<std::string> n;
    std::vector<long long> c;
    std::vector<double> m, w;
    prof_aggregate(h, n, c, m, &w);
    if (i < 0 || i >= (int32_t)n.size()) return HMSG_ERR_INVALID;
    snprintf(name, 64, "%s", n[i].c_str());
    *launches = c[i];
    *total_ms = m[i];
    if (total_work) *total_work = w[i];
    return HMSG_OK;
}

__global__ void k_depth_cut(unsigned short* __restrict__ depth, size_t n, double limit) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (double)depth[i] > limit) depth[i] = 0;
}

int hmsg_add_frames(hmsg_t* h, int32_t n, const uint8_t* rgb, const uint16_t* depth, const double* pose, const double* K) {
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE(n >= 0 && rgb && depth && pose && K, HMSG_ERR_INVALID, "hmsg_add_frames: null argument");
        HMSG_REQUIRE(!h->map_ready, HMSG_ERR_INVALID, "hmsg_add_frames after hmsg_finalize_map");
        HMSG_REQUIRE(!h->frames_released, HMSG_ERR_INVALID, "hmsg_add_frames: the frame store was released (hmsg_reset first)");
        double Kh[9];
        if (is_device_ptr(K)) {
            HIP_TRY(hipMemcpy(Kh, K, sizeof(Kh), hipMemcpyDeviceToHost));
        } else {
            memcpy(Kh, K, sizeof(Kh));
        }
        if (h->have_K)
            HMSG_REQUIRE(memcmp(Kh, h->K, sizeof(Kh)) == 0, HMSG_ERR_UNSUPPORTED, "intrinsics must be the same for all frames");
        memcpy(h->K, Kh, sizeof(Kh));
        h->have_K = true;
        h->cam = CamK{Kh[0], Kh[4], Kh[2], Kh[5]};
        const size_t HW = (size_t)h->cfg.height * h->cfg.width;
        const int skip = std::max(1, h->cfg.skip_frames);
        int kept = 0;
        if (skip == 1) {
            HMSG_REQUIRE(h->n_frames + n <= h->cfg.max_frames, HMSG_ERR_INVALID, "frame store full (cfg.max_frames)");
            copy_in(h->rgb.p + (size_t)h->n_frames * HW * 3, rgb, (size_t)n * HW * 3, h->stream);
            copy_in(h->depth.p + (size_t)h->n_frames * HW, depth, (size_t)n * HW * 2, h->stream);
            copy_in(h->pose.p + (size_t)h->n_frames * 16, pose, (size_t)n * 16 * 8, h->stream);
            kept = n;
        } else {
            // graph.py:339 / :373 `range(0, len(dataset), skip_frames)`: the k-th frame OFFERED (counted across calls) is kept
            // iff k % skip_frames == 0
            for (int i = 0; i < n; ++i) {
                if ((h->n_offered + i) % skip != 0) continue;
                HMSG_REQUIRE(h->n_frames + kept < h->cfg.max_frames, HMSG_ERR_INVALID, "frame store full (cfg.max_frames)");
                const size_t f = (size_t)h->n_frames + kept;
                copy_in(h->rgb.p + f * HW * 3, rgb + (size_t)i * HW * 3, HW * 3, h->stream);
                copy_in(h->depth.p + f * HW, depth + (size_t)i * HW, HW * 2, h->stream);
                copy_in(h->pose.p + f * 16, pose + (size_t)i * 16, 16 * 8, h->stream);
                ++kept;
            }
        }
        h->n_offered += n;
        if (h->cfg.depth_cut > 0.0 && kept > 0) {
            // horizon.py:258-261: `depth[depth > depth_cut * scale] = 0` (uint16 image against a float limit)
            const size_t cnt = (size_t)kept * HW;
            hipLaunchKernelGGL(k_depth_cut, dim3(cdiv(cnt, 256)), dim3(256), 0, h->stream, h->depth.p + (size_t)h->n_frames * HW, cnt,
                               h->cfg.depth_cut * h->cfg.depth_scale);
            HMSG_CHECK_LAUNCH();
        }
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->n_frames += kept;
    });
}

int hmsg_finalize_map(hmsg_t* h) {
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE(!h->map_ready, HMSG_ERR_INVALID, "map already finalised");
        hmsg_build_map(h);
    });
}

int64_t hmsg_num_tie_queries(const hmsg_t* h) { return h ? h->n_tie_queries : -1; }
int64_t hmsg_map_size(const hmsg_t* h) { return h && h->map_ready ? h->V : -1; }
int64_t hmsg_map_size_unfiltered(const hmsg_t* h) { return h && h->map_ready ? h->V0 : -1; }

int hmsg_get_map_points(const hmsg_t* hc, double* xyz, double* rgb) {
    hmsg_ctx* h = const_cast<hmsg_ctx*>(hc);
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE(h->map_ready, HMSG_ERR_INVALID, "map not finalised");
        if (xyz) d2h_bounce(xyz, h->pts.p, (size_t)h->V * 24);
        if (rgb) d2h_bounce(rgb, h->cols.p, (size_t)h->V * 24);
    });
}

int hmsg_add_frame_features(hmsg_t* h, int32_t first, int32_t n, int32_t M, const uint8_t* masks, const float* F_g,
                            const float* F_masked, const float* F_crop, const int32_t* n_masks) {
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE(n >= 0 && F_g, HMSG_ERR_INVALID, "hmsg_add_frame_features: null argument");
        HMSG_REQUIRE(M >= 0 && M <= h->cfg.max_masks, HMSG_ERR_INVALID, "M out of range (cfg.max_masks, <= 256)");
        HMSG_REQUIRE(M == 0 || (masks && F_masked && F_crop), HMSG_ERR_INVALID, "hmsg_add_frame_features: null argument");
        HMSG_REQUIRE(first == h->n_feat_frames, HMSG_ERR_INVALID, "frames must be handed over in order (first == #frames so far)");
        HMSG_REQUIRE(first + n <= h->n_frames, HMSG_ERR_INVALID, "features for a frame without geometry");
        HMSG_REQUIRE(!h->frames_released, HMSG_ERR_INVALID, "hmsg_add_frame_features: the frame store was released (hmsg_reset first)");
        if (n == 0) return;
        const size_t HW = (size_t)h->cfg.height * h->cfg.width;
        const int D = h->cfg.feat_dim;
        const size_t NW = (size_t)h->NW, MS = (size_t)h->MS;
        if (h->bits.n < (size_t)h->cfg.max_frames * HW * NW) h->bits.alloc((size_t)h->cfg.max_frames * HW * NW);
        if (h->fp.n < (size_t)h->cfg.max_frames * MS * D) h->fp.alloc((size_t)h->cfg.max_frames * MS * D);
        // per-frame mask counts (host copy kept: the 3-D mask store holds nmask[f] clouds for frame f)
        std::vector<int> nm((size_t)n, M);
        if (n_masks) {
            if (is_device_ptr(n_masks)) {
                HIP_TRY(hipMemcpy(nm.data(), n_masks, (size_t)n * 4, hipMemcpyDeviceToHost));
            } else {
                memcpy(nm.data(), n_masks, (size_t)n * 4);
            }
            for (int v : nm) HMSG_REQUIRE(v >= 0 && v <= M, HMSG_ERR_INVALID, "n_masks[f] must be in [0, M]");
        }
        DevBuf<int> d_nm;
        d_nm.alloc((size_t)n);
        HIP_TRY(hipMemcpyAsync(d_nm.p, nm.data(), (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
        if (M == 0) {   // no mask in any of these frames: empty bitsets, no F_p rows
            HIP_TRY(hipMemsetAsync(h->bits.p + (size_t)first * HW * NW, 0, (size_t)n * HW * NW * 8, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            h->nmask.insert(h->nmask.end(), nm.begin(), nm.end());
            h->n_feat_frames += n;
            return;
        }
        const bool dev = is_device_ptr(masks);
        const int chunk = dev ? n : std::max(1, (int)(((size_t)512 << 20) / ((size_t)M * HW)));
        DevBuf<unsigned char> st_m;
        DevBuf<float> st_f;
        for (int c0 = 0; c0 < n; c0 += chunk) {
            int nc = std::min(chunk, n - c0);
            const unsigned char* dm = masks + (size_t)c0 * M * HW;
            const float *dg = F_g + (size_t)c0 * D, *dfm = F_masked + (size_t)c0 * M * D, *dfc = F_crop + (size_t)c0 * M * D;
            if (!dev) {
                st_m.ensure((size_t)nc * M * HW);
                HIP_TRY(hipMemcpyAsync(st_m.p, dm, (size_t)nc * M * HW, hipMemcpyHostToDevice, h->stream));
                dm = st_m.p;
            }
            if (!is_device_ptr(F_g) || !is_device_ptr(F_masked) || !is_device_ptr(F_crop)) {
                st_f.ensure((size_t)nc * (2 * M + 1) * D);
                float* g = st_f.p;
                float* fm = g + (size_t)nc * D;
                float* fc = fm + (size_t)nc * M * D;
                copy_in(g, dg, (size_t)nc * D * 4, h->stream);
                copy_in(fm, dfm, (size_t)nc * M * D * 4, h->stream);
                copy_in(fc, dfc, (size_t)nc * M * D * 4, h->stream);
                dg = g;
                dfm = fm;
                dfc = fc;
            }
            hmsg_bitset_and_fp(h, first + c0, nc, M, dm, dg, dfm, dfc, d_nm.p + c0);
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
        h->nmask.insert(h->nmask.end(), nm.begin(), nm.end());
        h->n_feat_frames += n;
    });
}

int hmsg_set_frame_window(hmsg_t* h, int32_t first_frame) {
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE(h->map_ready, HMSG_ERR_INVALID, "hmsg_set_frame_window: call hmsg_finalize_map first");
        HMSG_REQUIRE(h->n_feat_frames == 0 && h->n_fused == 0, HMSG_ERR_INVALID, "hmsg_set_frame_window: features already handed over");
        HMSG_REQUIRE(first_frame >= 0 && first_frame <= h->n_frames, HMSG_ERR_INVALID, "hmsg_set_frame_window: frame out of range");
        h->frame_window = first_frame;
        h->n_feat_frames = h->n_fused = first_frame;
        h->nmask.assign((size_t)first_frame, 0);
        h->mask_first.assign((size_t)first_frame + 1, 0);
        h->masks3d.off.assign(1, 0);
        h->masks3d.total = 0;
    });
}

int hmsg_merge_tree_local(hmsg_t* h, int32_t total_frames, double* th_next, int64_t* lists_now, int64_t* my_index) {
    if (!h || !th_next || !lists_now || !my_index) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        long long l = 0, i = 0;
        hmsg_fold_pipe_abort(h);
        release_frame_store_if_large(h);
        hmsg_merge_tree_local_impl(h, total_frames, th_next, &l, &i);
        *lists_now = l;
        *my_index = i;
    });
}

int hmsg_merge_tree_join(hmsg_t* h, int32_t n_ext, const int64_t* ext_sizes, const double* ext_points, double th, int32_t final_pass) {
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] { hmsg_merge_tree_join_impl(h, n_ext, (const long long*)ext_sizes, ext_points, th, final_pass); });
}

int hmsg_fuse_frames(hmsg_t* h) {
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] { hmsg_fuse(h); });
}

int hmsg_get_map_feats(const hmsg_t* hc, float* feats, float* counter) {
    hmsg_ctx* h = const_cast<hmsg_ctx*>(hc);
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE(h->feats_final, HMSG_ERR_INVALID, "hmsg_fuse_frames not run");
        if (feats) d2h_bounce(feats, h->feats.p, (size_t)h->V * h->cfg.feat_dim * 4);
        if (counter) {
            std::vector<unsigned> c((size_t)h->V);
            HIP_TRY(hipMemcpy(c.data(), h->cnt.p, (size_t)h->V * 4, hipMemcpyDeviceToHost));
            for (long long i = 0; i < h->V; ++i) counter[i] = (float)c[i];
        }
    });
}

// graph.py:413-415 again after the sums changed
__global__ void k_feats_refresh(const float* __restrict__ sum, const unsigned* __restrict__ cnt, long long V, int D,
                                float* __restrict__ feats) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)V * D) return;
    unsigned c = cnt[t / D];
    float d = c == 0u ? 1e-5f : (float)c;
    feats[t] = __fdiv_rn(sum[t], d);
}

int hmsg_get_feature_sums(const hmsg_t* hc, float* sum, uint32_t* counter) {
    hmsg_ctx* h = const_cast<hmsg_ctx*>(hc);
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE(h->feats_final, HMSG_ERR_INVALID, "hmsg_fuse_frames not run");
        const size_t n = (size_t)h->V * h->cfg.feat_dim;
        // (on the handle's own stream -- ordered behind whatever produced the sums -- and complete before the call returns:
        //  the caller's stream has no ordering against ours)
        if (sum && n) HIP_TRY(hipMemcpyAsync(sum, h->sum.p, n * 4, is_device_ptr(sum) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
        if (counter && h->V)
            HIP_TRY(hipMemcpyAsync(counter, h->cnt.p, (size_t)h->V * 4, is_device_ptr(counter) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    });
}

int hmsg_set_feature_sums(hmsg_t* h, const float* sum, const uint32_t* counter) {
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE(h->feats_final && sum && counter, HMSG_ERR_INVALID, "hmsg_set_feature_sums: run hmsg_fuse_frames first");
        HMSG_REQUIRE(!h->pooled, HMSG_ERR_INVALID, "hmsg_set_feature_sums after hmsg_pool_instances");
        const size_t n = (size_t)h->V * h->cfg.feat_dim;
        copy_in(h->sum.p, sum, n * 4, h->stream);
        copy_in(h->cnt.p, counter, (size_t)h->V * 4, h->stream);
        if (n) hipLaunchKernelGGL(k_feats_refresh, dim3(cdiv(n, 256)), dim3(256), 0, h->stream, (const float*)h->sum.p,
                                  (const unsigned*)h->cnt.p, (long long)h->V, h->cfg.feat_dim, h->feats.p);
        HMSG_CHECK_LAUNCH();
        HIP_TRY(hipStreamSynchronize(h->stream));
    });
}

int hmsg_get_frame_nn(const hmsg_t* hc, int32_t frame, int32_t* idx) {
    hmsg_ctx* h = const_cast<hmsg_ctx*>(hc);
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE(frame >= 0 && frame < h->n_fused && idx, HMSG_ERR_INVALID, "frame not fused");
        HMSG_REQUIRE(!h->frames_released, HMSG_ERR_INVALID, "hmsg_get_frame_nn: the frame store was released by the merge (very long episode)");
        const size_t HW = (size_t)h->cfg.height * h->cfg.width;
        HIP_TRY(hipMemcpy(idx, h->nn.p + (size_t)frame * HW, HW * 4, hipMemcpyDeviceToHost));
    });
}

int hmsg_get_frame_fp(const hmsg_t* hc, int32_t frame, float* f_p) {
    hmsg_ctx* h = const_cast<hmsg_ctx*>(hc);
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE(frame >= 0 && frame < h->n_feat_frames && f_p, HMSG_ERR_INVALID, "frame has no features");
        const size_t D = (size_t)h->cfg.feat_dim, n = (size_t)h->nmask[frame] * D;
        if (n) HIP_TRY(hipMemcpy(f_p, h->fp.p + (size_t)frame * h->MS * D, n * 4, hipMemcpyDeviceToHost));
    });
}

int32_t hmsg_get_frame_num_masks(const hmsg_t* h, int32_t frame) {
    return (h && frame >= 0 && frame < h->n_feat_frames) ? h->nmask[frame] : -1;
}

int hmsg_get_frame_mask_sizes(const hmsg_t* hc, int32_t frame, int64_t* sizes) {
    hmsg_ctx* h = const_cast<hmsg_ctx*>(hc);
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE(frame >= 0 && frame < h->n_fused && sizes, HMSG_ERR_INVALID, "frame not fused");
        for (int i = 0; i < h->nmask[frame]; ++i) {
            size_t k = (size_t)h->mask_first[frame] + i;
            sizes[i] = h->masks3d.off[k + 1] - h->masks3d.off[k];
        }
    });
}

int hmsg_get_frame_mask_points(const hmsg_t* hc, int32_t frame, double* xyz) {
    hmsg_ctx* h = const_cast<hmsg_ctx*>(hc);
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE(frame >= 0 && frame < h->n_fused && xyz, HMSG_ERR_INVALID, "frame not fused");
        long long a = h->masks3d.off[(size_t)h->mask_first[frame]], b = h->masks3d.off[(size_t)h->mask_first[frame + 1]];
        if (b > a) d2h_bounce(xyz, h->masks3d.pts.p + (size_t)a * 3, (size_t)(b - a) * 24);
    });
}

int hmsg_merge_instances(hmsg_t* h) {
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        release_frame_store_if_large(h);
        hmsg_merge(h);
    });
}

int64_t hmsg_num_instances(const hmsg_t* h) { return h && (h->merged || h->tree_partial) ? (int64_t)h->inst.off.size() - 1 : -1; }

int hmsg_get_instance_sizes(const hmsg_t* hc, int64_t* sizes) {
    hmsg_ctx* h = const_cast<hmsg_ctx*>(hc);
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE((h->merged || h->tree_partial) && sizes, HMSG_ERR_INVALID, "hmsg_merge_instances not run");
        for (size_t i = 0; i + 1 < h->inst.off.size(); ++i) sizes[i] = h->inst.off[i + 1] - h->inst.off[i];
    });
}

int hmsg_get_instance_points(const hmsg_t* hc, double* xyz) {
    hmsg_ctx* h = const_cast<hmsg_ctx*>(hc);
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE((h->merged || h->tree_partial) && xyz, HMSG_ERR_INVALID, "hmsg_merge_instances not run");
        if (h->inst.total) {
            if (is_device_ptr(xyz)) {
                HIP_TRY(hipMemcpyAsync(xyz, h->inst.pts.p, (size_t)h->inst.total * 24, hipMemcpyDeviceToDevice, h->stream));
                HIP_TRY(hipStreamSynchronize(h->stream));
            } else
                d2h_bounce(xyz, h->inst.pts.p, (size_t)h->inst.total * 24);
        }
    });
}

int hmsg_denoise_instances(hmsg_t* h, double eps, int32_t min_points) {
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        hmsg_denoise_inst(h, eps, min_points);
        if (eps == 0.05 && min_points == 10) h->inst_denoised = true;
    });
}

int hmsg_voxel_down_sample(hmsg_t* h, const double* points, int64_t n, double voxel_size, double* out_points, int64_t* out_n) {
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE((points || n == 0) && out_points && out_n, HMSG_ERR_INVALID, "hmsg_voxel_down_sample: null argument");
        *out_n = (int64_t)hmsg_voxel_ds(h, points, (long long)n, voxel_size, out_points);
    });
}

int hmsg_instance_room_share(hmsg_t* h, int32_t n_rooms, const int64_t* vert_off, const double* verts_xz, double radius,
                             double* share) {
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE(n_rooms >= 0 && vert_off && verts_xz && share && radius > 0, HMSG_ERR_INVALID,
                     "hmsg_instance_room_share: bad argument");
        hmsg_room_share(h, n_rooms, (const long long*)vert_off, verts_xz, radius, share);
    });
}

int hmsg_get_instance_boxes(const hmsg_t* hc, double* boxes) {
    hmsg_ctx* h = const_cast<hmsg_ctx*>(hc);
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE(h->merged && boxes, HMSG_ERR_INVALID, "hmsg_merge_instances not run");
        if (!h->inst.box.empty()) memcpy(boxes, h->inst.box.data(), h->inst.box.size() * 8);
    });
}

int hmsg_pool_instances(hmsg_t* h) {
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] { hmsg_pool(h); });
}

int hmsg_get_instance_feats(const hmsg_t* hc, float* feats) {
    hmsg_ctx* h = const_cast<hmsg_ctx*>(hc);
    if (!h) return HMSG_ERR_INVALID;
    return guard(h, [&] {
        HMSG_REQUIRE(h->pooled && feats, HMSG_ERR_INVALID, "hmsg_pool_instances not run");
        size_t n = (h->inst.off.size() - 1) * (size_t)h->cfg.feat_dim;
        if (n) d2h_bounce(feats, h->inst_feats.p, n * 4);
    });
}

// test hook: the host restatement of scipy's cKDTree (hmsg_ckdtree.h) -- index permutation after the build and
// k = 1 answers; no GPU involved
int hmsg_test_ckdtree(const double* pts, int64_t n, const double* queries, int64_t nq, int64_t* out_idx,
                      int64_t* out_indices, int64_t* out_n_nodes) {
    if (!pts || n < 0 || nq < 0) return HMSG_ERR_INVALID;
    CKDTree t;
    t.build(pts, n);
    if (out_indices) memcpy(out_indices, t.indices.data(), (size_t)n * 8);
    if (out_n_nodes) *out_n_nodes = (int64_t)t.nodes.size();
    for (int64_t i = 0; i < nq; ++i) out_idx[i] = t.query1(queries + i * 3);
    return HMSG_OK;
}

}  // extern "C"
