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
#include "hmsg_nn.h"
#include <chrono>
#include "hmsg_ckdtree.h"

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
        DbgLaps laps("object nodes", h->stream);
        if (!h->inst_denoised) {
            hmsg_denoise_inst(h, 0.05, 10);
            h->inst_denoised = true;
        }
        laps.lap("per-object DBSCAN");
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
        laps.lap("label similarity");
        std::vector<double> share((size_t)N * std::max(n_rooms, 1), 0.0);
        if (n_rooms > 0) hmsg_room_share(h, n_rooms, (const long long*)vert_off, verts_xz, 0.2, share.data());
        laps.lap("room shares");
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
        laps.lap("floors / rooms (host)");
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


// ------------------------------------------------------------------------------------------ A9: room clouds
// segment_hmsg_room (graph.py:1086-1108): the (x, z) cell centres of a room's 2-D region are extruded over the storey's
// height in 5 cm steps, turned into the map frame (pcd.transform(T1), T1 = Rotation.from_euler("x", 90) as scipy
// gives it: cos(90 deg) is 6.1e-17 there, not 0, so the matrix comes from the caller) and each of those points picks its
// nearest neighbour in the FLOOR cloud = full_pcd.crop(y in [y_lo, y_hi]) (:769-775; cKDTree.query(k = 1), "very slow"
// says the reference's comment: 10^5..10^6 queries per room); the room cloud is floor_pcd.select_by_index(idx): the
// picked points, each once, in the floor cloud's order.  Device: a thread per extruded point, exact nearest neighbour
// among the map points of the storey (ring expansion over the map's occupancy grid, hmsg_nn.h); queries whose two
// nearest candidates are at BIT-EQUAL distance go to the host, which replays scipy's cKDTree built over the floor cloud.
struct RoomTie {
    int room;
    double x, y, z;
};
struct RoomScratch {            // hmsg_room_clouds' work buffers (hmsg_ctx::room_scratch)
    DevBuf<unsigned char> ok, mark;
    DevBuf<unsigned> okw, frank, nt, ccnt, cstart, ccur, fl, pos;
    DevBuf<double> dT, dz, dxz;
    DevBuf<long long> doff;
    DevBuf<RoomTie> ties;
    DevBuf<int> cidx;
    DevBuf<float> cp32;
    DevBuf<unsigned long long> ccol;
};
__global__ void k_floor_mask(const double* __restrict__ pts, long long V, double y_lo, double y_hi, unsigned char* __restrict__ ok,
                             unsigned* __restrict__ okw) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const double y = pts[i * 3 + 1];
    const bool in = y >= y_lo && y <= y_hi;                  // Open3D crop: both bounds inclusive
    ok[i] = in ? 1 : 0;
    okw[i] = in ? 1u : 0u;
}
// The extruded points fill the ROOM'S VOLUME: most of them hang in mid-air, metres from the nearest surviving map point
// (remove_radius_outlier deletes ~90 % of the voxels of a sparsely furnished room), where ring expansion over the 5 cm
// occupancy grid costs O(r^2) probes per ring.  For this stage the storey's points are binned once into a COARSE grid
// (cells of >= 25 cm, at most 64 along the vertical so that one word holds a column's occupancy): a query expands rings of
// coarse cells, tests the points of the occupied ones, and stops when the cube it has covered reaches past its best hit.
struct CoarseGrid {
    double ox, oy, oz, cs;
    int nx, ny, nz;                           // ny <= 64 (vertical)
    const unsigned* start;                    // [nx * nz * ny + 1], cell = (ix * nz + iz) * ny + iy
    const int* idx;                           // map point indices, cell by cell
    const float* p32;                         // the same points as float32 (x, y, z), cell by cell: the pre-filter of k_room_nn
    const unsigned long long* col;            // [nx * nz] occupied iy bits
};
__device__ __forceinline__ void coarse_cell(const CoarseGrid& c, double x, double y, double z, int& ix, int& iy, int& iz) {
    ix = (int)floor((x - c.ox) / c.cs);
    iy = (int)floor((y - c.oy) / c.cs);
    iz = (int)floor((z - c.oz) / c.cs);
    ix = ix < 0 ? 0 : (ix >= c.nx ? c.nx - 1 : ix);
    iy = iy < 0 ? 0 : (iy >= c.ny ? c.ny - 1 : iy);
    iz = iz < 0 ? 0 : (iz >= c.nz ? c.nz - 1 : iz);
}
__global__ void k_coarse_count(const double* __restrict__ pts, const unsigned char* __restrict__ ok, long long V, CoarseGrid c, unsigned* __restrict__ cnt,
                               unsigned long long* __restrict__ col) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V || !ok[i]) return;
    int ix, iy, iz;
    coarse_cell(c, pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2], ix, iy, iz);
    atomicAdd(&cnt[((size_t)ix * c.nz + iz) * c.ny + iy], 1u);
    const unsigned long long bit = 1ull << iy;
    if (!(col[(size_t)ix * c.nz + iz] & bit)) atomicOr(&col[(size_t)ix * c.nz + iz], bit);
}
__global__ void k_coarse_fill(const double* __restrict__ pts, const unsigned char* __restrict__ ok, long long V, CoarseGrid c, unsigned* __restrict__ cursor,
                              int* __restrict__ idx, float* __restrict__ p32) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V || !ok[i]) return;
    int ix, iy, iz;
    coarse_cell(c, pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2], ix, iy, iz);
    const size_t cell = ((size_t)ix * c.nz + iz) * c.ny + iy;
    const size_t k = c.start[cell] + atomicAdd(&cursor[cell], 1u);   // (order inside a cell is irrelevant: minimum + tie COUNT)
    idx[k] = (int)i;
    p32[k * 3] = (float)pts[i * 3];
    p32[k * 3 + 1] = (float)pts[i * 3 + 1];
    p32[k * 3 + 2] = (float)pts[i * 3 + 2];
}
// float32 bound for the pre-filter: everything within sqrt(d2) of the query has a float32 squared distance below this.  With
// coordinates up to ~10^3 m a float32 coordinate is off by <= 6e-5 m, a difference by <= 1.2e-4, the squared distance of points
// d apart by <= ~2 d 2.1e-4 + 1.4e-7 plus the float32 rounding of the sum (relative 2e-7): margin 1e-3 (d + 1) + 1e-6 d2, generous.
__device__ __forceinline__ float best_bound(double d2) {
    if (d2 > 1e30) return 3.0e38f;
    const double d = sqrt(d2);
    return (float)(d2 + 1e-3 * (d + 1.0) + 1e-6 * d2);
}
// One WAVE per (x, z) cell of a room's region, one LANE per extrusion level: the 60-odd queries of a cell lie on one line (the
// storey's height), their nearest neighbours are found by ONE ring traversal of the coarse grid around the box that holds the
// line -- every lane measures every candidate against its own query (the candidate's load is wave-uniform).  A thread per
// query walked the same rings 60 times over: 26 ms for 4.2 million queries, all of it the traversal.
__global__ void __launch_bounds__(256) k_room_nn(CoarseGrid C, const double* __restrict__ map_pts, const unsigned* __restrict__ floor_rank,
                                                 long long NF, const double* __restrict__ T, int n_levels, const double* __restrict__ z_levels,
                                                 int n_rooms, const long long* __restrict__ room_off, const double* __restrict__ room_xz,
                                                 unsigned char* __restrict__ mark, unsigned* __restrict__ n_ties, RoomTie* __restrict__ ties,
                                                 unsigned tie_cap) {
    const long long total_cells = room_off[n_rooms];
    const int lane = threadIdx.x & 63;
    const long long cell = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (cell >= total_cells) return;                        // (wave-uniform)
    int lo = 0, hi = n_rooms - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (room_off[mid] <= cell) lo = mid; else hi = mid - 1;
    }
    const double X = room_xz[cell * 2], Y = room_xz[cell * 2 + 1];
    for (int l0 = 0; l0 < n_levels; l0 += 64) {
        const int lvl = l0 + lane;
        const bool valid = lvl < n_levels;
        const double Z = z_levels[valid ? lvl : l0];
        // Open3D transform: ((X*T0 + Y*T1) + Z*T2) + T3 per row, divided by the homogeneous row; no FMA
        double r[4];
        for (int k = 0; k < 4; ++k)
            r[k] = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(X, T[k * 4]), __dmul_rn(Y, T[k * 4 + 1])), __dmul_rn(Z, T[k * 4 + 2])), T[k * 4 + 3]);
        const double px = __ddiv_rn(r[0], r[3]), py = __ddiv_rn(r[1], r[3]), pz = __ddiv_rn(r[2], r[3]);
        int cx, cy, cz;
        coarse_cell(C, px, py, pz, cx, cy, cz);
        // the box of coarse cells that holds the wave's queries (idle lanes repeat the chunk's first query)
        int bx0 = cx, bx1 = cx, by0 = cy, by1 = cy, bz0 = cz, bz1 = cz;
        for (int o = 32; o > 0; o >>= 1) {
            bx0 = min(bx0, __shfl_xor(bx0, o)); bx1 = max(bx1, __shfl_xor(bx1, o));
            by0 = min(by0, __shfl_xor(by0, o)); by1 = max(by1, __shfl_xor(by1, o));
            bz0 = min(bz0, __shfl_xor(bz0, o)); bz1 = max(bz1, __shfl_xor(bz1, o));
        }
        NNBest best{1e300, -1, 0};
        const float qxf = (float)px, qyf = (float)py, qzf = (float)pz;
        float bestf = 3.0e38f;
        const int rmax = max(C.nx, max(C.ny, C.nz));
        for (int rr = 0; rr <= rmax; ++rr) {
            const int y0 = max(by0 - rr, 0), y1 = min(by1 + rr, C.ny - 1);
            const unsigned long long span = (y1 >= 63 ? ~0ull : ((1ull << (y1 + 1)) - 1ull)) & ~((1ull << y0) - 1ull);
            unsigned long long shell = 0ull;                 // the two new layers of a column that earlier rings covered
            if (rr > 0 && by0 - rr >= 0) shell |= 1ull << (by0 - rr);
            if (rr > 0 && by1 + rr < C.ny) shell |= 1ull << (by1 + rr);
            for (int ix = max(bx0 - rr, 0); ix <= min(bx1 + rr, C.nx - 1); ++ix)
                for (int iz = max(bz0 - rr, 0); iz <= min(bz1 + rr, C.nz - 1); ++iz) {
                    const bool rim = rr == 0 || ix == bx0 - rr || ix == bx1 + rr || iz == bz0 - rr || iz == bz1 + rr;
                    unsigned long long m = C.col[(size_t)ix * C.nz + iz] & (rim ? span : shell);
                    while (m) {
                        const int iy = __ffsll(m) - 1;
                        m &= m - 1ull;
                        const size_t cc = ((size_t)ix * C.nz + iz) * C.ny + iy;
                        for (unsigned k = C.start[cc]; k < C.start[cc + 1]; ++k) {
                            // float32 pre-filter (the float64 vector rate is what this kernel is bound by: ~7 000 candidates per
                            // cell line x 64 lanes): a candidate whose float32 distance exceeds the lane's best by more than the
                            // float32 error of coordinates of a few hundred metres cannot win or tie -- the exact float64 form
                            // (scipy's operation order) is evaluated only for the few that pass
                            const float fx = __fsub_rn(C.p32[(size_t)k * 3], qxf), fy = __fsub_rn(C.p32[(size_t)k * 3 + 1], qyf),
                                        fz = __fsub_rn(C.p32[(size_t)k * 3 + 2], qzf);
                            const float f2 = fx * fx + fy * fy + fz * fz;
                            if (f2 <= bestf) {               // (a wave in which no lane passes skips the body: no vote needed)
                                const int q = C.idx[k];
                                nn_consider(best, q, nn_dist2(map_pts + (size_t)q * 3, px, py, pz));
                                bestf = best_bound(best.d2);
                            }
                        }
                    }
                }
            // every cell within Chebyshev distance rr of the box is done; a point of any other cell is at least rr cell sides
            // away from every query of the wave (they all lie in, or beyond the clamped edge of, the box)
            const double mm = (double)rr * C.cs - 1e-9;
            const bool done = !valid || (best.idx >= 0 && mm > 0.0 && best.d2 < mm * mm);
            if (__all(done)) break;
        }
        if (!valid || best.idx < 0) continue;
        if (best.ntie > 1) {
            const unsigned k = atomicAdd(n_ties, 1u);
            if (k < tie_cap) ties[k] = RoomTie{lo, px, py, pz};
            continue;
        }
        mark[(size_t)lo * NF + floor_rank[best.idx]] = 1;
    }
}

// marks -> compact room lists on the device (what used to be a host loop over rooms x floor points) + the inverse of the floor
// rank (floor point -> map point), so that the room clouds stay usable on the device: hmsg_room_camera_distances
__global__ void k_rc_flags(const unsigned char* __restrict__ mark, size_t n, unsigned* __restrict__ flags) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = mark[i] ? 1u : 0u;
}
__global__ void k_rc_compact(const unsigned char* __restrict__ mark, const unsigned* __restrict__ pos, size_t n, long long NF, int* __restrict__ sel) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && mark[i]) sel[pos[i]] = (int)(i % (size_t)NF);
}
__global__ void k_rc_offsets(const unsigned* __restrict__ pos, const unsigned char* __restrict__ mark, long long NF, int n_rooms, long long* __restrict__ off) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rooms) off[r] = pos[(size_t)r * NF];
    if (r == n_rooms) off[r] = (long long)pos[(size_t)n_rooms * NF - 1] + (mark[(size_t)n_rooms * NF - 1] ? 1 : 0);
}
__global__ void k_rc_floor_map(const unsigned char* __restrict__ ok, const unsigned* __restrict__ frank, long long V, int* __restrict__ fmap) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < V && ok[i]) fmap[frank[i]] = (int)i;
}
__global__ void k_rc_xz(const int* __restrict__ sel, const int* __restrict__ fmap, const double* __restrict__ pts, long long n, double* __restrict__ xz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t q = (size_t)fmap[sel[i]];
    xz[i * 2] = pts[q * 3];
    xz[i * 2 + 1] = pts[q * 3 + 2];
}
// compute_room_embeddings (utils/graph_utils.py:244-291): np.min(cdist([pos], room_points)) = sqrt of the smallest
// dx*dx + dy*dy (float64, that order); one workgroup per (camera, room)  [the statement of hmsg_points_min_dist_2d]
__global__ void __launch_bounds__(256) k_rc_min_dist(const long long* __restrict__ set_off, const double* __restrict__ pts, int n_sets,
                                                     const double* __restrict__ q, double* __restrict__ out) {
    __shared__ double s_m[4];
    const int qi = blockIdx.y, si = blockIdx.x;
    const double qx = q[(size_t)qi * 2], qy = q[(size_t)qi * 2 + 1];
    double m = 1e308 * 10.0;
    for (long long k = set_off[si] + threadIdx.x; k < set_off[si + 1]; k += blockDim.x) {
        const double dx = __dsub_rn(qx, pts[k * 2]), dy = __dsub_rn(qy, pts[k * 2 + 1]);
        const double d2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
        m = d2 < m ? d2 : m;
    }
    m = wave_min_f64(m);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) m = s_m[w] < m ? s_m[w] : m;
        out[(size_t)qi * n_sets + si] = __dsqrt_rn(m);
    }
}

extern "C" int hmsg_room_clouds(hmsg_t* h, double y_lo, double y_hi, const double* T, int32_t n_levels, const double* z_levels,
                                int32_t n_rooms, const int64_t* room_off, const double* room_xz, int64_t* out_sizes,
                                int32_t* out_index, int64_t out_capacity, int64_t* n_floor_points) {
    if (!h) return HMSG_ERR_INVALID;
    try {
        HIP_TRY(hipSetDevice(h->cfg.device_id));
        HMSG_REQUIRE(h->map_ready && T && n_levels >= 0 && (n_levels == 0 || z_levels) && n_rooms >= 0 && room_off && out_sizes && n_floor_points,
                     HMSG_ERR_INVALID, "hmsg_room_clouds: bad argument (finalize the map first)");
        hipStream_t s = h->stream;
        const long long V = h->V, cells = room_off[n_rooms];
        HMSG_REQUIRE(cells == 0 || room_xz, HMSG_ERR_INVALID, "hmsg_room_clouds: room points missing");
        // the resident room lists belong to THIS call from here on: a call that fails or finds nothing must not leave the
        // previous storey's rooms behind for hmsg_room_camera_distances
        h->room_n = 0;
        h->room_total = 0;
        // The work buffers stay with the handle (grow-only): as locals they went through the thread's allocator cache, where the
        // other stages of a build take and return blocks of similar sizes -- the call then cost 10 ms or 40 ms depending on
        // what the cache happened to hold (measured: kernels 9.7 ms every time, library call 11.6 / 27.5 / 38.8 / 10.5 ms over
        // the first scenes of a process).
        if (!h->room_scratch) h->room_scratch = std::make_shared<RoomScratch>();
        RoomScratch& W = *static_cast<RoomScratch*>(h->room_scratch.get());
        DevBuf<unsigned char>&ok = W.ok, &mark = W.mark;
        DevBuf<unsigned>&okw = W.okw, &frank = W.frank, &nt = W.nt, &ccnt = W.ccnt, &cstart = W.cstart, &ccur = W.ccur, &fl = W.fl, &pos = W.pos;
        DevBuf<double>&dT = W.dT, &dz = W.dz, &dxz = W.dxz;
        DevBuf<long long>& doff = W.doff;
        DevBuf<RoomTie>& ties = W.ties;
        DevBuf<int>& cidx = W.cidx;
        DevBuf<float>& cp32 = W.cp32;
        DevBuf<unsigned long long>& ccol = W.ccol;
        ok.ensure((size_t)std::max<long long>(V, 1));
        okw.ensure((size_t)std::max<long long>(V, 1));
        frank.ensure((size_t)std::max<long long>(V, 1));
        unsigned long long NFu = 0;
        if (V) {
            hipLaunchKernelGGL(k_floor_mask, dim3(cdiv((size_t)V, 256)), dim3(256), 0, s, (const double*)h->pts.p, V, y_lo, y_hi, ok.p, okw.p);
            HMSG_CHECK_LAUNCH();
            hmsg_scan_u32(okw.p, frank.p, (size_t)V, s, h->scan_tmp, &NFu);
        }
        const long long NF = (long long)NFu;
        *n_floor_points = NF;
        for (int r = 0; r < n_rooms; ++r) out_sizes[r] = 0;
        if (NF == 0 || cells == 0 || n_levels == 0 || n_rooms == 0) {
            if (n_rooms > 0) {                                      // n_rooms empty rooms: every distance to them is inf
                h->room_off_dev.ensure((size_t)n_rooms + 1);
                HIP_TRY(hipMemsetAsync(h->room_off_dev.p, 0, ((size_t)n_rooms + 1) * 8, s));
                HIP_TRY(hipStreamSynchronize(s));
                h->room_n = n_rooms;
            }
            return HMSG_OK;
        }
        HMSG_REQUIRE((long long)n_rooms * NF < (1ll << 33), HMSG_ERR_UNSUPPORTED, "hmsg_room_clouds: too many rooms x floor points");
        const unsigned tie_cap = 1u << 20;
        dT.ensure(16);
        dz.ensure((size_t)n_levels);
        dxz.ensure((size_t)cells * 2);
        doff.ensure((size_t)n_rooms + 1);
        mark.ensure((size_t)n_rooms * NF);
        nt.ensure(1);
        ties.ensure(tie_cap);
        std::vector<long long> hoff((size_t)n_rooms + 1);
        for (int r = 0; r <= n_rooms; ++r) hoff[(size_t)r] = room_off[r];
        HIP_TRY(hipMemcpyAsync(dT.p, T, 128, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(dz.p, z_levels, (size_t)n_levels * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(dxz.p, room_xz, (size_t)cells * 16, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(doff.p, hoff.data(), hoff.size() * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(mark.p, 0, (size_t)n_rooms * NF, s));
        HIP_TRY(hipMemsetAsync(nt.p, 0, 4, s));
        // the storey's points in a coarse grid over the map's extent (vertical = y; at most 64 cells along it)
        CoarseGrid C;
        {
            const GridGeom& g = h->grid;
            const double ex = g.nx * g.vs, ey = g.ny * g.vs, ez = g.nz * g.vs;
            C.cs = std::max(0.25, ey / 60.0);
            C.ox = g.ox;
            C.oy = g.oy;
            C.oz = g.oz;
            C.nx = (int)std::floor(ex / C.cs) + 1;
            C.ny = std::min(64, (int)std::floor(ey / C.cs) + 1);
            C.nz = (int)std::floor(ez / C.cs) + 1;
        }
        const size_t ncc = (size_t)C.nx * C.ny * C.nz;
        HMSG_REQUIRE(ncc < ((size_t)1 << 31), HMSG_ERR_UNSUPPORTED, "hmsg_room_clouds: map extent too large for the coarse grid");
        ccnt.ensure(ncc + 1);
        cstart.ensure(ncc + 1);
        ccur.ensure(ncc + 1);
        cidx.ensure((size_t)NF);
        cp32.ensure((size_t)NF * 3);
        ccol.ensure((size_t)C.nx * C.nz);
        HIP_TRY(hipMemsetAsync(ccnt.p, 0, (ncc + 1) * 4, s));
        HIP_TRY(hipMemsetAsync(ccur.p, 0, (ncc + 1) * 4, s));
        HIP_TRY(hipMemsetAsync(ccol.p, 0, (size_t)C.nx * C.nz * 8, s));
        C.start = cstart.p;
        C.idx = cidx.p;
        C.p32 = cp32.p;
        C.col = ccol.p;
        hipLaunchKernelGGL(k_coarse_count, dim3(cdiv((size_t)V, 256)), dim3(256), 0, s, (const double*)h->pts.p, (const unsigned char*)ok.p, V, C, ccnt.p, ccol.p);
        HMSG_CHECK_LAUNCH();
        hmsg_scan_u32(ccnt.p, cstart.p, ncc + 1, s, h->scan_tmp, nullptr);
        hipLaunchKernelGGL(k_coarse_fill, dim3(cdiv((size_t)V, 256)), dim3(256), 0, s, (const double*)h->pts.p, (const unsigned char*)ok.p, V, C, ccur.p, cidx.p, cp32.p);
        HMSG_CHECK_LAUNCH();
        const long long nq = cells * n_levels;
        hipLaunchKernelGGL(k_room_nn, dim3(cdiv((size_t)cells * 64, 256)), dim3(256), 0, s, C, (const double*)h->pts.p, (const unsigned*)frank.p, NF,
                           (const double*)dT.p, n_levels, (const double*)dz.p, n_rooms, (const long long*)doff.p, (const double*)dxz.p, mark.p, nt.p,
                           ties.p, tie_cap);
        HMSG_CHECK_LAUNCH();
        const bool dbg = getenv("HMSG_DEBUG_TIMING") != nullptr;
        auto tnow = [] { return std::chrono::steady_clock::now(); };
        auto t_a = tnow();
        if (dbg) HIP_TRY(hipStreamSynchronize(s));
        auto t_b = tnow();
        std::vector<unsigned char> hmark;
        unsigned n_t = 0;
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(&n_t, nt.p, 4, hipMemcpyDeviceToHost));
        if (n_t) {                                                  // (the marks come to the host only to be patched)
            hmark.resize((size_t)n_rooms * NF);
            d2h_bounce(hmark.data(), mark.p, hmark.size());
        }
        auto t_c = tnow();
        if (dbg)
            fprintf(stderr, "[hmsg room clouds] %lld queries, %u bit-equal ties: kernels %.2f ms, read-back %.2f ms\n", nq, n_t,
                    std::chrono::duration<double, std::milli>(t_b - t_a).count(), std::chrono::duration<double, std::milli>(t_c - t_b).count());
        HMSG_REQUIRE(n_t <= tie_cap, HMSG_ERR_UNSUPPORTED, "hmsg_room_clouds: more than 2^20 bit-equal nearest-neighbour ties");
        if (n_t) {
            // scipy's answer for the tied queries: the restated cKDTree over the floor cloud (crop order)
            hmsg_kd_join(h);
            std::vector<double> fpts((size_t)NF * 3);
            {
                std::vector<unsigned char> hok((size_t)V);
                HIP_TRY(hipMemcpy(hok.data(), ok.p, (size_t)V, hipMemcpyDeviceToHost));
                size_t k = 0;
                for (long long i = 0; i < V; ++i)
                    if (hok[(size_t)i]) {
                        for (int a = 0; a < 3; ++a) fpts[k * 3 + a] = h->host_pts[(size_t)i * 3 + a];
                        ++k;
                    }
            }
            CKDTree tree;
            tree.build(fpts.data(), NF);
            std::vector<RoomTie> ht(n_t);
            HIP_TRY(hipMemcpy(ht.data(), ties.p, (size_t)n_t * sizeof(RoomTie), hipMemcpyDeviceToHost));
            for (auto& t : ht) {
                const double x[3] = {t.x, t.y, t.z};
                hmark[(size_t)t.room * NF + (size_t)tree.query1(x)] = 1;
            }
            h->n_tie_queries += n_t;
        }
        if (dbg) fprintf(stderr, "[hmsg room clouds] ties resolved after %.2f ms\n", std::chrono::duration<double, std::milli>(tnow() - t_c).count());
        if (n_t) HIP_TRY(hipMemcpyAsync(mark.p, hmark.data(), hmark.size(), hipMemcpyHostToDevice, s));       // (the patched ties)
        // compact room lists on the device; they stay with the handle (hmsg_room_camera_distances)
        const size_t nm = (size_t)n_rooms * (size_t)NF;
        fl.ensure(nm);
        pos.ensure(nm);
        hipLaunchKernelGGL(k_rc_flags, dim3(cdiv(nm, 256)), dim3(256), 0, s, (const unsigned char*)mark.p, nm, fl.p);
        HMSG_CHECK_LAUNCH();
        hmsg_scan_u32(fl.p, pos.p, nm, s, h->scan_tmp, nullptr);
        h->room_off_dev.ensure((size_t)n_rooms + 1);
        hipLaunchKernelGGL(k_rc_offsets, dim3(cdiv((size_t)n_rooms + 1, 64)), dim3(64), 0, s, (const unsigned*)pos.p, (const unsigned char*)mark.p, NF, n_rooms,
                           h->room_off_dev.p);
        HMSG_CHECK_LAUNCH();
        std::vector<long long> hoff2((size_t)n_rooms + 1);
        HIP_TRY(hipMemcpyAsync(hoff2.data(), h->room_off_dev.p, hoff2.size() * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const long long total = hoff2[(size_t)n_rooms];
        h->room_sel.ensure((size_t)std::max<long long>(total, 1));
        hipLaunchKernelGGL(k_rc_compact, dim3(cdiv(nm, 256)), dim3(256), 0, s, (const unsigned char*)mark.p, (const unsigned*)pos.p, nm, NF, h->room_sel.p);
        h->room_fmap.ensure((size_t)NF);
        hipLaunchKernelGGL(k_rc_floor_map, dim3(cdiv((size_t)V, 256)), dim3(256), 0, s, (const unsigned char*)ok.p, (const unsigned*)frank.p, V, h->room_fmap.p);
        HMSG_CHECK_LAUNCH();
        h->room_n = n_rooms;
        h->room_total = total;
        for (int r = 0; r < n_rooms; ++r) out_sizes[r] = hoff2[(size_t)r + 1] - hoff2[(size_t)r];
        if (out_index && total <= out_capacity && total > 0) {
            HIP_TRY(hipStreamSynchronize(s));
            d2h_bounce(out_index, h->room_sel.p, (size_t)total * 4);
        }
        HIP_TRY(hipStreamSynchronize(s));
        HMSG_REQUIRE(!out_index || total <= out_capacity, HMSG_ERR_INVALID, "hmsg_room_clouds: out_index too small (sum of out_sizes needed)");
        return HMSG_OK;
    } catch (const hmsg_error& e) {
        h->err = e.msg;
        return e.code;
    }
}

// camera -> room distance table of compute_room_embeddings (utils/graph_utils.py:244-291) from the room clouds the last
// hmsg_room_clouds call left on the device: room r = the (x, z) of its selected floor points, in the floor cloud's order
// (np.min over them is order-independent); out f64 [n_q][n_rooms].
extern "C" int hmsg_room_camera_distances(hmsg_t* h, int32_t n_rooms, int64_t n_q, const double* q_xz, double* out) {
    if (!h || n_q < 0 || (n_q > 0 && (!q_xz || !out))) return HMSG_ERR_INVALID;
    try {
        HIP_TRY(hipSetDevice(h->cfg.device_id));
        HMSG_REQUIRE(h->room_n > 0, HMSG_ERR_INVALID, "hmsg_room_camera_distances: call hmsg_room_clouds first");
        HMSG_REQUIRE(n_rooms == h->room_n, HMSG_ERR_INVALID,
                     "hmsg_room_camera_distances: n_rooms is not the room count of the last hmsg_room_clouds call (out is [n_q][n_rooms])");
        if (n_q == 0) return HMSG_OK;
        hipStream_t s = h->stream;
        DevBuf<double> xz, dq, dout;
        xz.alloc((size_t)std::max<long long>(h->room_total, 1) * 2);
        dq.alloc((size_t)n_q * 2);
        dout.alloc((size_t)n_q * (size_t)h->room_n);
        if (h->room_total)
            hipLaunchKernelGGL(k_rc_xz, dim3(cdiv((size_t)h->room_total, 256)), dim3(256), 0, s, (const int*)h->room_sel.p, (const int*)h->room_fmap.p,
                               (const double*)h->pts.p, h->room_total, xz.p);
        HIP_TRY(hipMemcpyAsync(dq.p, q_xz, (size_t)n_q * 16, hipMemcpyHostToDevice, s));
        for (long long q0 = 0; q0 < n_q; q0 += 32768) {
            const unsigned nq = (unsigned)std::min<long long>(32768, n_q - q0);
            hipLaunchKernelGGL(k_rc_min_dist, dim3((unsigned)h->room_n, nq), dim3(256), 0, s, (const long long*)h->room_off_dev.p, (const double*)xz.p, h->room_n,
                               (const double*)dq.p + q0 * 2, dout.p + q0 * h->room_n);
        }
        HMSG_CHECK_LAUNCH();
        HIP_TRY(hipStreamSynchronize(s));
        d2h_bounce(out, dout.p, (size_t)n_q * (size_t)h->room_n * 8);
        return HMSG_OK;
    } catch (const hmsg_error& e) {
        h->err = e.msg;
        return e.code;
    }
}

// ------------------------------------------------------------------------------------------ A10: view <-> object topology
// check_object_in_view (utils/graph_utils.py:95-157) for a batch of (instance, view) pairs, the loop of
// segment_hmsg_objects (graph.py:1712-1734): points into the camera frame (world -> camera matrix), those in front of the
// camera (z > 0) through K, visible = at least `min_ratio` of ALL the cloud's points land inside the image and their mean
// depth is <= max_depth.  The knife edges are not rare here: a point seen at pixel column 0 of a frame re-projects into that
// very frame at u = 0 +- 1e-14, and the frames ARE the views.  The reference's two matmuls go through BLAS dgemm, whose
// micro-kernel keeps one accumulator per output element and feeds it with fused multiply-adds in k order; the chain
// fma(a3, b3, fma(a2, b2, fma(a1, b1, a0 * b0))) reproduces numpy + OpenBLAS of this image bit for bit
// (tests/test_object_views.py holds the kernel to numpy's own matmul results), so the inside / outside decisions are the
// reference run's.  The mean depth is summed in workgroup order (numpy: pairwise): equal to ~1e-16 relative.
// One workgroup per pair; the instance's points are read where the merge left them.
struct ViewPair {
    long long p0;      // first point of the instance
    int n;             // its points
    int view;
};
__global__ void __launch_bounds__(256) k_object_views(const double* __restrict__ pts, const ViewPair* __restrict__ pairs,
                                                      const double* __restrict__ pose_inv, const int* __restrict__ wh, const double* __restrict__ Kmat,
                                                      double min_ratio, double max_depth, unsigned char* __restrict__ visible,
                                                      double* __restrict__ mean_depth) {
    __shared__ double s_z[4];
    __shared__ int s_c[4][2];
    const ViewPair pr = pairs[blockIdx.x];
    const double* P = pose_inv + (size_t)pr.view * 16;
    const double W = (double)wh[(size_t)pr.view * 2], H = (double)wh[(size_t)pr.view * 2 + 1];
    double P_[12], K_[9];
    for (int i = 0; i < 12; ++i) P_[i] = P[i];
    for (int i = 0; i < 9; ++i) K_[i] = Kmat[i];
    int n_front = 0, n_in = 0;
    double zsum = 0.0;
    for (int k = threadIdx.x; k < pr.n; k += blockDim.x) {
        const double* q = pts + (size_t)(pr.p0 + k) * 3;
        const double x = q[0], y = q[1], z = q[2];
        double c[3];
        for (int r = 0; r < 3; ++r)
            c[r] = fma(P_[r * 4 + 3], 1.0, fma(P_[r * 4 + 2], z, fma(P_[r * 4 + 1], y, __dmul_rn(P_[r * 4], x))));
        if (!(c[2] > 0.0)) continue;
        ++n_front;
        double ph[3];
        for (int r = 0; r < 3; ++r)
            ph[r] = fma(K_[r * 3 + 2], c[2], fma(K_[r * 3 + 1], c[1], __dmul_rn(K_[r * 3], c[0])));
        const double u = __ddiv_rn(ph[0], ph[2]), v = __ddiv_rn(ph[1], ph[2]);
        if (u >= 0.0 && u < W && v >= 0.0 && v < H) {
            ++n_in;
            zsum = __dadd_rn(zsum, c[2]);
        }
    }
    n_front = wave_sum_i32(n_front);
    n_in = wave_sum_i32(n_in);
    for (int o = 32; o > 0; o >>= 1) zsum = __dadd_rn(zsum, __shfl_xor(zsum, o));
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_z[w] = zsum;
        s_c[w][0] = n_front;
        s_c[w][1] = n_in;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nf = (s_c[0][0] + s_c[1][0]) + (s_c[2][0] + s_c[3][0]), ni = (s_c[0][1] + s_c[1][1]) + (s_c[2][1] + s_c[3][1]);
        const double zs = __dadd_rn(__dadd_rn(s_z[0], s_z[1]), __dadd_rn(s_z[2], s_z[3]));
        const double inf = 1e308 * 10.0;
        unsigned char vis = 0;
        double md = inf;
        if (pr.n > 0 && nf > 0 && ni > 0 && !((double)ni / (double)pr.n < min_ratio)) {
            md = __ddiv_rn(zs, (double)ni);
            vis = md > max_depth ? 0 : 1;
        }
        visible[blockIdx.x] = vis;
        mean_depth[blockIdx.x] = md;
    }
}

extern "C" int hmsg_object_views(hmsg_t* h, int32_t n_views, const double* pose_inv, const int32_t* wh, const double* K, int64_t n_pairs,
                                 const int32_t* pair_inst, const int32_t* pair_view, double min_visible_ratio, double max_depth,
                                 uint8_t* visible, double* mean_depth) {
    if (!h) return HMSG_ERR_INVALID;
    try {
        HIP_TRY(hipSetDevice(h->cfg.device_id));
        HMSG_REQUIRE(h->merged, HMSG_ERR_INVALID, "hmsg_object_views: run hmsg_merge_instances first");
        HMSG_REQUIRE(n_views >= 0 && n_pairs >= 0 && n_pairs < (1ll << 31) && (n_pairs == 0 || (pose_inv && wh && K && pair_inst && pair_view && visible && mean_depth)),
                     HMSG_ERR_INVALID, "hmsg_object_views: bad argument");
        if (n_pairs == 0) return HMSG_OK;
        const long long NI = (long long)h->inst.off.size() - 1;
        std::vector<ViewPair> hp((size_t)n_pairs);
        for (int64_t k = 0; k < n_pairs; ++k) {
            HMSG_REQUIRE(pair_inst[k] >= 0 && pair_inst[k] < NI && pair_view[k] >= 0 && pair_view[k] < n_views, HMSG_ERR_INVALID,
                         "hmsg_object_views: pair out of range");
            const long long p0 = h->inst.off[(size_t)pair_inst[k]], p1 = h->inst.off[(size_t)pair_inst[k] + 1];
            hp[(size_t)k] = ViewPair{p0, (int)(p1 - p0), pair_view[k]};
        }
        hipStream_t s = h->stream;
        DevBuf<ViewPair> d_pairs;
        DevBuf<double> d_pose, d_K, d_md;
        DevBuf<int> d_wh;
        DevBuf<unsigned char> d_vis;
        d_pairs.alloc((size_t)n_pairs);
        d_pose.alloc((size_t)n_views * 16);
        d_K.alloc(9);
        d_wh.alloc((size_t)n_views * 2);
        d_md.alloc((size_t)n_pairs);
        d_vis.alloc((size_t)n_pairs);
        HIP_TRY(hipMemcpyAsync(d_pairs.p, hp.data(), (size_t)n_pairs * sizeof(ViewPair), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_pose.p, pose_inv, (size_t)n_views * 128, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_K.p, K, 72, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_wh.p, wh, (size_t)n_views * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_object_views, dim3((unsigned)n_pairs), dim3(256), 0, s, (const double*)h->inst.pts.p, (const ViewPair*)d_pairs.p,
                           (const double*)d_pose.p, (const int*)d_wh.p, (const double*)d_K.p, min_visible_ratio, max_depth, d_vis.p, d_md.p);
        HMSG_CHECK_LAUNCH();
        HIP_TRY(hipStreamSynchronize(s));
        d2h_bounce(visible, d_vis.p, (size_t)n_pairs);
        d2h_bounce(mean_depth, d_md.p, (size_t)n_pairs * 8);
        return HMSG_OK;
    } catch (const hmsg_error& e) {
        h->err = e.msg;
        return e.code;
    }
}
