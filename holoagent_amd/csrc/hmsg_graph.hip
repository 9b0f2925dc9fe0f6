// A10 behind the C ABI: objects of the scene graph from the merged + pooled instances -- segment_hmsg_objects
// (fsr_vln/memory/hmsg/graph/graph.py:1582-1736) without the per-view visibility test (that one needs the
// dataset's images and stays with the caller) -- and the node table a host reads back or turns into a resident
// retrieval index (hmsg_index_from_nodes) without a round trip through host memory.
//
// Per instance (after the in-place pcd_denoise_dbscan(0.05, 10) of graph.py:1589-1591, device): skipped when it
// has fewer than 10 points; for every floor whose [zero - 0.2, zero + height + 0.2] contains its y-extent
// (:1611-1620): room = arg-max over the floor's rooms of find_intersection_share (device, hmsg_room_share);
// when every share is 0, the room whose vertex centroid is nearest to the instance's x/z centroid (:1645-1655);
// label = arg-max over the label text features of emb . text^T (identify_object, :1441-1454, float64 MFMA GEMM);
// object id = (room, running counter of that room) (:1696-1700).
#include "hmsg_common.h"

#include <cmath>

extern "C" {

int hmsg_build_object_nodes(hmsg_t* h, int32_t n_floors, const double* floor_zero, const double* floor_height, int32_t n_rooms,
                            const int32_t* room_floor, const int64_t* vert_off, const double* verts_xz, int32_t n_labels,
                            const float* label_feats) {
    if (!h) return HMSG_ERR_INVALID;
    try {
        HIP_TRY(hipSetDevice(h->cfg.device_id));
        HMSG_REQUIRE(h->pooled, HMSG_ERR_INVALID, "hmsg_build_object_nodes: run hmsg_pool_instances first");
        HMSG_REQUIRE(n_floors >= 0 && n_rooms >= 0 && (n_floors == 0 || (floor_zero && floor_height)) &&
                         (n_rooms == 0 || (room_floor && vert_off && verts_xz)) && (n_labels == 0 || label_feats),
                     HMSG_ERR_INVALID, "hmsg_build_object_nodes: bad argument");
        const int D = h->cfg.feat_dim;
        if (!h->inst_denoised) {
            hmsg_denoise_inst(h, 0.05, 10);
            h->inst_denoised = true;
        }
        const int N = (int)h->inst.off.size() - 1;
        h->nodes.clear();
        h->node_label.assign((size_t)std::max(N, 0), -1);
        if (N <= 0) return HMSG_OK;
        // labels: S = emb . text^T on the device (float32 values in float64 arithmetic, as NodeIndex.similarity)
        if (n_labels > 0) {
            hmsg_index_t* ix = nullptr;
            std::vector<int32_t> zero((size_t)n_labels, 0);
            HMSG_REQUIRE(hmsg_index_create(h->cfg.device_id, D, n_labels, label_feats, 0, zero.data(), &ix) == HMSG_OK, HMSG_ERR_HIP,
                         "hmsg_build_object_nodes: label table");
            std::vector<double> S((size_t)N * n_labels);
            int rc = hmsg_similarity(ix, N, h->inst_feats.p, S.data());
            hmsg_index_destroy(ix);
            HMSG_REQUIRE(rc == HMSG_OK, HMSG_ERR_HIP, "hmsg_build_object_nodes: label similarity");
            for (int i = 0; i < N; ++i) {
                int best = 0;
                for (int l = 1; l < n_labels; ++l)
                    if (S[(size_t)i * n_labels + l] > S[(size_t)i * n_labels + best]) best = l;     // np.argmax: first maximum
                h->node_label[(size_t)i] = best;
            }
        }
        std::vector<double> share((size_t)N * std::max(n_rooms, 1), 0.0);
        if (n_rooms > 0) hmsg_room_share(h, n_rooms, (const long long*)vert_off, verts_xz, 0.2, share.data());
        // room vertex centroids (np.mean(axis=0): rows added in order)
        std::vector<double> rc((size_t)n_rooms * 2, 0.0);
        for (int r = 0; r < n_rooms; ++r) {
            double sx = 0.0, sz = 0.0;
            for (long long v = vert_off[r]; v < vert_off[r + 1]; ++v) {
                sx += verts_xz[v * 2];
                sz += verts_xz[v * 2 + 1];
            }
            const double n = (double)(vert_off[r + 1] - vert_off[r]);
            rc[(size_t)r * 2] = sx / n;
            rc[(size_t)r * 2 + 1] = sz / n;
        }
        std::vector<int> counter((size_t)n_rooms, 0);
        const double margin = 0.2;
        std::vector<double> pts;
        for (int f = 0; f < n_floors; ++f) {
            std::vector<int> rooms_f;
            for (int r = 0; r < n_rooms; ++r)
                if (room_floor[r] == f) rooms_f.push_back(r);
            for (int i = 0; i < N; ++i) {
                const long long n_i = h->inst.off[(size_t)i + 1] - h->inst.off[(size_t)i];
                if (n_i < 10) continue;
                const double ymin = h->inst.box[(size_t)i * 6 + 1], ymax = h->inst.box[(size_t)i * 6 + 4];
                if (!(ymin > floor_zero[f] - margin && ymax < floor_zero[f] + floor_height[f] + margin)) continue;
                if (rooms_f.empty()) continue;
                double sum = 0.0;
                for (int r : rooms_f) sum += share[(size_t)i * n_rooms + r];
                int best = rooms_f[0];
                if (sum == 0.0) {          // no room vertex near the object: nearest room centre to the x/z centroid
                    pts.resize((size_t)n_i * 3);
                    HIP_TRY(hipMemcpy(pts.data(), h->inst.pts.p + (size_t)h->inst.off[(size_t)i] * 3, (size_t)n_i * 24, hipMemcpyDeviceToHost));
                    double cx = 0.0, cz = 0.0;
                    for (long long k = 0; k < n_i; ++k) {
                        cx += pts[(size_t)k * 3];
                        cz += pts[(size_t)k * 3 + 2];
                    }
                    cx /= (double)n_i;
                    cz /= (double)n_i;
                    double bv = 0.0;
                    bool have = false;
                    for (int r : rooms_f) {
                        const double dx = rc[(size_t)r * 2] - cx, dz = rc[(size_t)r * 2 + 1] - cz;
                        const double v = -std::sqrt(dx * dx + dz * dz);     // -np.linalg.norm
                        if (!have || v > bv) {
                            bv = v;
                            best = r;
                            have = true;
                        }
                    }
                } else {
                    double bv = share[(size_t)i * n_rooms + best];
                    for (int r : rooms_f)
                        if (share[(size_t)i * n_rooms + r] > bv) {
                            bv = share[(size_t)i * n_rooms + r];
                            best = r;
                        }
                }
                h->nodes.push_back(hmsg_node{i, f, best, counter[(size_t)best]++, h->node_label[(size_t)i], (int64_t)n_i});
            }
        }
        return HMSG_OK;
    } catch (const hmsg_error& e) {
        h->err = e.msg;
        return e.code;
    }
}

int64_t hmsg_num_nodes(const hmsg_t* h) { return h ? (int64_t)h->nodes.size() : -1; }

int hmsg_get_nodes(const hmsg_t* hc, hmsg_node* nodes, float* embeddings) {
    hmsg_ctx* h = const_cast<hmsg_ctx*>(hc);
    if (!h) return HMSG_ERR_INVALID;
    try {
        HIP_TRY(hipSetDevice(h->cfg.device_id));
        const size_t n = h->nodes.size();
        if (nodes && n) memcpy(nodes, h->nodes.data(), n * sizeof(hmsg_node));
        if (embeddings && n) {
            const size_t D = (size_t)h->cfg.feat_dim, NI = h->inst.off.size() - 1;
            std::vector<float> all(NI * D);
            HIP_TRY(hipMemcpy(all.data(), h->inst_feats.p, NI * D * 4, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < n; ++k) memcpy(embeddings + k * D, all.data() + (size_t)h->nodes[k].instance * D, D * 4);
        }
        return HMSG_OK;
    } catch (const hmsg_error& e) {
        h->err = e.msg;
        return e.code;
    }
}

// retrieval index over the node table, built from the device-resident pooled features (node k -> its instance's
// embedding, parent = the node's room)
__global__ void k_gather_rows_f32(const float* __restrict__ src, const int* __restrict__ row, int n, int D, float* __restrict__ dst) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * D) return;
    dst[t] = src[(size_t)row[t / D] * D + t % D];
}

int hmsg_index_from_nodes(hmsg_t* h, hmsg_index_t** out) {
    if (!h || !out) return HMSG_ERR_INVALID;
    *out = nullptr;
    try {
        HIP_TRY(hipSetDevice(h->cfg.device_id));
        const int n = (int)h->nodes.size(), D = h->cfg.feat_dim;
        HMSG_REQUIRE(n > 0, HMSG_ERR_INVALID, "hmsg_index_from_nodes: no nodes (hmsg_build_object_nodes)");
        std::vector<int> inst((size_t)n), room((size_t)n);
        for (int k = 0; k < n; ++k) {
            inst[(size_t)k] = h->nodes[(size_t)k].instance;
            room[(size_t)k] = h->nodes[(size_t)k].room;
        }
        DevBuf<int> d_inst;
        DevBuf<float> emb;
        d_inst.alloc((size_t)n);
        emb.alloc((size_t)n * D);
        HIP_TRY(hipMemcpyAsync(d_inst.p, inst.data(), (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_gather_rows_f32, dim3(cdiv((size_t)n * D, 256)), dim3(256), 0, h->stream, (const float*)h->inst_feats.p,
                           (const int*)d_inst.p, n, D, emb.p);
        HMSG_CHECK_LAUNCH();
        HIP_TRY(hipStreamSynchronize(h->stream));
        return hmsg_index_create(h->cfg.device_id, D, n, emb.p, 0, room.data(), out);
    } catch (const hmsg_error& e) {
        h->err = e.msg;
        return e.code;
    }
}

}  // extern "C"
