// CPU RESTATEMENT of the HMSG build + retrieval path in C++ -- TEST INFRASTRUCTURE and the `cpu_baseline` leg of bench.py.
//
// The same algorithm as oracle/hmsg_oracle.py (which restates the reference function by function; the file:line
// citations below are the reference's, /root/reference/fsr_vln), compiled: what BASELINE.md section 3 calls the CPU
// baseline that travels to the GPU box.  Algorithmically faithful -- the sequential merge re-clusters EVERY cloud of the
// list every frame like graph_utils.py:918-956 does, the overlap test looks at every point of both clouds, the pooling
// runs the O(n^2 D) cosine DBSCAN -- and parallel (OpenMP) only where the reference itself is: the cKDTree queries
// (`workers=-1`), the radius counts, the BLAS-backed similarity products.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load it; the product (holoagent_amd/) never does.
// tests/test_cpu_restatement.py pins it against the fixtures the REFERENCE's own Python produced (tests/golden/build_seq,
// build_hier, build_ragged: map cloud and merged instances bit for bit, pooled features to 1e-5) and against hmsg_oracle.py on a
// synthetic scene (3-D masks bit for bit too, retrieval indices exactly).  Open3D / faiss / sklearn internals are restated
// from their published algorithms exactly as hmsg_oracle.py does (see its header): unpinned there, unpinned here.
//
//   g++ -O2 -std=c++17 -fopenmp -ffp-contract=off -mf16c -shared -fPIC oracle/hmsg_cpu.cpp -o oracle/libhmsg_cpu.so
#include <immintrin.h>
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <unordered_map>
#include <vector>

#include "../holoagent_amd/csrc/hmsg_ckdtree.h"   // scipy.spatial.cKDTree restated (pinned against scipy: tests/test_ckdtree.py)

namespace {

typedef std::vector<double> VD;
struct Cloud {
    VD p, c;                       // xyz / rgb interleaved
    size_t n() const { return p.size() / 3; }
};

// ---- numpy float32 pairwise summation (np.sum over a contiguous axis; numpy/core/src/umath/loops_utils.h)
float np_pairwise_sum_f32(const float* a, size_t n) {
    if (n < 8) {
        float r = 0.f;
        for (size_t i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        float r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        size_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum_f32(a, n2) + np_pairwise_sum_f32(a + n2, n - n2);
}
float sumsq_f32(const float* x, size_t n, std::vector<float>& tmp) {
    tmp.resize(n);
    for (size_t i = 0; i < n; ++i) tmp[i] = x[i] * x[i];
    return np_pairwise_sum_f32(tmp.data(), n);
}

// ---- Open3D VoxelDownSample (oracle: o3d_voxel_down_sample; call sites graph.py:348, generic.py:188, graph.py:456):
// means of the points of a voxel, summed in input order; output in ascending (ix, iy, iz)
void voxel_down_sample(const VD& pts, const VD* cols, double vs, VD& out, VD* outc) {
    const size_t n = pts.size() / 3;
    out.clear();
    if (outc) outc->clear();
    if (n == 0) return;
    double mn[3] = {pts[0], pts[1], pts[2]};
    for (size_t i = 1; i < n; ++i)
        for (int a = 0; a < 3; ++a) mn[a] = std::min(mn[a], pts[i * 3 + a]);
    double org[3];
    for (int a = 0; a < 3; ++a) org[a] = mn[a] - vs * 0.5;
    std::vector<int64_t> ix(n * 3);
    int64_t dim[3] = {0, 0, 0};
    for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            const int64_t v = (int64_t)std::floor((pts[i * 3 + a] - org[a]) / vs);
            ix[i * 3 + a] = v;
            dim[a] = std::max(dim[a], v + 1);
        }
    std::vector<std::pair<int64_t, size_t>> key(n);
    for (size_t i = 0; i < n; ++i) key[i] = {(ix[i * 3] * dim[1] + ix[i * 3 + 1]) * dim[2] + ix[i * 3 + 2], i};
    std::sort(key.begin(), key.end());              // (lin, input index): groups ascending, input order inside
    const bool hc = cols && cols->size() == pts.size() && outc;
    for (size_t a = 0; a < n;) {
        size_t b = a;
        double s[3] = {0, 0, 0}, sc[3] = {0, 0, 0};
        while (b < n && key[b].first == key[a].first) {
            const size_t i = key[b].second;
            for (int k = 0; k < 3; ++k) s[k] += pts[i * 3 + k];
            if (hc)
                for (int k = 0; k < 3; ++k) sc[k] += (*cols)[i * 3 + k];
            ++b;
        }
        const double cnt = (double)(b - a);
        for (int k = 0; k < 3; ++k) out.push_back(s[k] / cnt);
        if (hc)
            for (int k = 0; k < 3; ++k) outc->push_back(sc[k] / cnt);
        a = b;
    }
}

// ---- uniform grid over a cloud for exact radius searches (stands in for nanoflann / cKDTree ball queries)
struct Grid {
    double org[3], cell;
    int64_t dim[3];
    std::vector<int64_t> start;      // CSR over cells
    std::vector<int> order;
    const double* p;
    void build(const double* pts, size_t n, double cs) {
        p = pts;
        cell = cs;
        double mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
        for (size_t i = 0; i < n; ++i)
            for (int a = 0; a < 3; ++a) {
                if (i == 0 || pts[i * 3 + a] < mn[a]) mn[a] = pts[i * 3 + a];
                if (i == 0 || pts[i * 3 + a] > mx[a]) mx[a] = pts[i * 3 + a];
            }
        for (int a = 0; a < 3; ++a) {
            org[a] = mn[a];
            dim[a] = (int64_t)std::floor((mx[a] - mn[a]) / cs) + 1;
        }
        const int64_t nc = dim[0] * dim[1] * dim[2];
        start.assign((size_t)nc + 1, 0);
        std::vector<int64_t> ci(n);
        for (size_t i = 0; i < n; ++i) {
            ci[i] = cell_of(pts + i * 3);
            ++start[(size_t)ci[i] + 1];
        }
        for (int64_t c = 0; c < nc; ++c) start[(size_t)c + 1] += start[(size_t)c];
        order.resize(n);
        std::vector<int64_t> cur(start.begin(), start.end() - 1);
        for (size_t i = 0; i < n; ++i) order[(size_t)cur[(size_t)ci[i]]++] = (int)i;     // ascending index inside a cell
    }
    int64_t coord(double v, int a) const {
        int64_t c = (int64_t)std::floor((v - org[a]) / cell);
        return std::min(std::max<int64_t>(c, 0), dim[a] - 1);
    }
    int64_t cell_of(const double* q) const { return (coord(q[0], 0) * dim[1] + coord(q[1], 1)) * dim[2] + coord(q[2], 2); }
    // f(j) for every point j in the cells within `reach` cells of q; f returns false to stop
    template <class F>
    void visit(const double* q, int reach, F f) const {
        const int64_t cx = coord(q[0], 0), cy = coord(q[1], 1), cz = coord(q[2], 2);
        for (int64_t x = std::max<int64_t>(cx - reach, 0); x <= std::min(cx + reach, dim[0] - 1); ++x)
            for (int64_t y = std::max<int64_t>(cy - reach, 0); y <= std::min(cy + reach, dim[1] - 1); ++y)
                for (int64_t z = std::max<int64_t>(cz - reach, 0); z <= std::min(cz + reach, dim[2] - 1); ++z) {
                    const int64_t c = (x * dim[1] + y) * dim[2] + z;
                    for (int64_t k = start[(size_t)c]; k < start[(size_t)c + 1]; ++k)
                        if (!f(order[(size_t)k])) return;
                }
    }
};
inline double d2_np(const double* a, const double* b) {     // d0*d0 + d1*d1 + d2*d2, left to right (numpy)
    const double d0 = a[0] - b[0], d1 = a[1] - b[1], dd = a[2] - b[2];
    return (d0 * d0 + d1 * d1) + dd * dd;
}

// ---- Open3D ClusterDBSCAN (oracle: o3d_cluster_dbscan; graph_utils.py:839): strict radius, self included
void dbscan_labels(const VD& pts, double eps, int min_points, std::vector<int64_t>& labels) {
    const size_t n = pts.size() / 3;
    labels.assign(n, -1);
    if (n == 0) return;
    Grid g;
    g.build(pts.data(), n, eps);
    const double e2 = eps * eps;
    std::vector<int> cnt(n, 0);
#pragma omp parallel for schedule(dynamic, 256) if (n > 20000)
    for (long long i = 0; i < (long long)n; ++i) {
        int c = 0;
        g.visit(&pts[(size_t)i * 3], 1, [&](int j) {
            c += d2_np(&pts[(size_t)i * 3], &pts[(size_t)j * 3]) < e2 ? 1 : 0;      // (i itself: 0 < e2)
            return true;
        });
        cnt[(size_t)i] = c;
    }
    std::vector<char> core(n);
    bool any = false;
    for (size_t i = 0; i < n; ++i) {
        core[i] = cnt[i] >= min_points;
        any = any || core[i];
    }
    if (!any) return;
    // components of the core graph; cluster id = rank of the component's smallest core index
    std::vector<int> parent(n);
    std::iota(parent.begin(), parent.end(), 0);
    auto find = [&](int x) {
        while (parent[(size_t)x] != x) x = parent[(size_t)x] = parent[(size_t)parent[(size_t)x]];
        return x;
    };
    for (size_t i = 0; i < n; ++i) {
        if (!core[i]) continue;
        g.visit(&pts[i * 3], 1, [&](int j) {
            if ((size_t)j > i && core[(size_t)j] && d2_np(&pts[i * 3], &pts[(size_t)j * 3]) < e2) {
                const int a = find((int)i), b = find(j);
                if (a != b) parent[(size_t)std::max(a, b)] = std::min(a, b);
            }
            return true;
        });
    }
    std::vector<int64_t> id_of(n, -1);
    int64_t next = 0;
    for (size_t i = 0; i < n; ++i)
        if (core[i]) {
            const int r = find((int)i);
            if (id_of[(size_t)r] < 0) id_of[(size_t)r] = next++;
            labels[i] = id_of[(size_t)r];
        }
    // border points: smallest cluster id over the adjacent core points
    for (size_t i = 0; i < n; ++i) {
        if (core[i]) continue;
        int64_t best = -1;
        g.visit(&pts[i * 3], 1, [&](int j) {
            if (core[(size_t)j] && d2_np(&pts[i * 3], &pts[(size_t)j * 3]) < e2 && (best < 0 || labels[(size_t)j] < best)) best = labels[(size_t)j];
            return true;
        });
        labels[i] = best;
    }
}
// largest cluster by Counter.most_common (first-seen label on ties); -1: none
int64_t largest_label(const std::vector<int64_t>& labels) {
    std::unordered_map<int64_t, std::pair<size_t, size_t>> cnt;   // label -> (count, first position)
    for (size_t i = 0; i < labels.size(); ++i) {
        if (labels[i] < 0) continue;
        auto it = cnt.find(labels[i]);
        if (it == cnt.end()) cnt[labels[i]] = {1, i};
        else ++it->second.first;
    }
    int64_t best = -1;
    size_t bc = 0, bf = 0;
    for (auto& kv : cnt)
        if (kv.second.first > bc || (kv.second.first == bc && kv.second.second < bf)) {
            best = kv.first;
            bc = kv.second.first;
            bf = kv.second.second;
        }
    return best;
}
// pcd_denoise_dbscan (graph_utils.py:827-880)
void pcd_denoise(Cloud& c, double eps, int min_points) {
    std::vector<int64_t> lab;
    dbscan_labels(c.p, eps, min_points, lab);
    const int64_t best = largest_label(lab);
    if (best < 0) return;
    size_t keep = 0;
    for (auto l : lab) keep += l == best;
    if (keep < 5) return;
    Cloud o;
    const bool hc = c.c.size() == c.p.size();
    for (size_t i = 0; i < lab.size(); ++i)
        if (lab[i] == best) {
            o.p.insert(o.p.end(), c.p.begin() + (long)i * 3, c.p.begin() + (long)i * 3 + 3);
            if (hc) o.c.insert(o.c.end(), c.c.begin() + (long)i * 3, c.c.begin() + (long)i * 3 + 3);
        }
    if (!hc) o.c = c.c;
    c = std::move(o);
}

// ---- A1 create_pcd (generic.py:101-138): valid pixels in row-major order; returns world points (+ pixel index list)
void create_pcd(const uint16_t* depth, const uint8_t* rgb, const uint8_t* mask, int H, int W, const double* pose, const double* K,
                double filter_distance, VD& pts, VD* cols, std::vector<int>* pix) {
    pts.clear();
    if (cols) cols->clear();
    if (pix) pix->clear();
    double zsum = 0.0;
    size_t nz = 0;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)y * W + x;
            float z = (float)depth[i] / 1000.0f;
            if (mask) z = z * (float)mask[i];
            if (!(z > 0)) continue;
            const double X = ((double)x - K[2]) * (double)z / K[0], Y = ((double)y - K[5]) * (double)z / K[4], Z = (double)z;
            double r[4];
            for (int k = 0; k < 4; ++k) r[k] = ((X * pose[k * 4] + Y * pose[k * 4 + 1]) + Z * pose[k * 4 + 2]) + pose[k * 4 + 3];
            pts.push_back(r[0] / r[3]);
            pts.push_back(r[1] / r[3]);
            pts.push_back(r[2] / r[3]);
            if (cols && rgb)
                for (int k = 0; k < 3; ++k) cols->push_back((double)rgb[i * 3 + k] / 255.0);
            if (pix) pix->push_back((int)i);
            zsum += z;
            ++nz;
        }
    if (nz && zsum / (double)nz > filter_distance) {      // whole-frame / whole-mask reject (:126-127)
        pts.clear();
        if (cols) cols->clear();
        if (pix) pix->clear();
    }
}

// ---- A6 helpers
double bbox_iou(const double* amn, const double* amx, const double* bmn, const double* bmx) {   // graph_utils.py:883-915
    double ov = 1, va = 1, vb = 1;
    for (int k = 0; k < 3; ++k) {
        const double omin = std::max(amn[k], bmn[k]), omax = std::min(amx[k], bmx[k]);
        ov *= std::max(omax - omin, 0.0);
        va *= amx[k] - amn[k];
        vb *= bmx[k] - bmn[k];
    }
    return ov / (va + vb - ov);          // 0/0 -> NaN: comparisons false, like numpy
}
// fraction-of-points-within-r in float32 (graph_utils.py:620-664): (dx*dx + dy*dy) + dz*dz < r2, float32
size_t count_within(const std::vector<float>& a, const std::vector<float>& b, float r, float r2) {
    const size_t na = a.size() / 3, nb = b.size() / 3;
    if (!na || !nb) return 0;
    // grid over b in float coordinates (as doubles)
    VD bd(b.begin(), b.end());
    Grid g;
    g.build(bd.data(), nb, (double)r * 1.001 + 1e-3);
    size_t hits = 0;
#pragma omp parallel for reduction(+ : hits) schedule(static) if (na > 50000)
    for (long long i = 0; i < (long long)na; ++i) {
        const float x = a[(size_t)i * 3], y = a[(size_t)i * 3 + 1], z = a[(size_t)i * 3 + 2];
        const double q[3] = {x, y, z};
        bool hit = false;
        g.visit(q, 1, [&](int j) {
            const float dx = x - b[(size_t)j * 3], dy = y - b[(size_t)j * 3 + 1], dz = z - b[(size_t)j * 3 + 2];
            if ((dx * dx + dy * dy) + dz * dz < r2) {
                hit = true;
                return false;
            }
            return true;
        });
        hits += hit;
    }
    return hits;
}

struct Ctx {
    int H, W, D;
    VD map_p, map_c;
    CKDTree tree;
    std::vector<float> full_feats;
    std::vector<Cloud> frames_masks;          // all frames' mask clouds, frame-major
    std::vector<int> frame_first;
    std::vector<Cloud> inst;
    std::vector<float> inst_feats;
};

// merge_3d_masks (graph_utils.py:918-956)
void merge_3d_masks(std::vector<Cloud>& L, double th, double radius, double iou_thresh) {
    const size_t n = L.size();
    std::vector<double> mn(n * 3, 0.0), mx(n * 3, 0.0);
    std::vector<std::vector<float>> f32(n);
    for (size_t i = 0; i < n; ++i) {
        const size_t m = L[i].n();
        for (size_t k = 0; k < m; ++k)
            for (int a = 0; a < 3; ++a) {
                const double v = L[i].p[k * 3 + a];
                if (k == 0 || v < mn[i * 3 + a]) mn[i * 3 + a] = v;
                if (k == 0 || v > mx[i * 3 + a]) mx[i * 3 + a] = v;
            }
        f32[i].assign(L[i].p.begin(), L[i].p.end());
    }
    if (n == 0) return;
    const float r = (float)(1.5 * radius), r2 = (float)((1.5 * radius) * (1.5 * radius));
    std::vector<int> parent(n);
    std::iota(parent.begin(), parent.end(), 0);
    auto find = [&](int x) {
        while (parent[(size_t)x] != x) x = parent[(size_t)x] = parent[(size_t)parent[(size_t)x]];
        return x;
    };
    for (size_t i = 0; i < n; ++i)
        for (size_t j = i + 1; j < n; ++j) {
            if (!(bbox_iou(&mn[i * 3], &mx[i * 3], &mn[j * 3], &mx[j * 3]) > iou_thresh)) continue;
            double ratio = 0.0;
            if (L[i].n() && L[j].n()) {
                const double r1 = (double)count_within(f32[i], f32[j], r, r2) / (double)L[i].n();
                const double r2v = (double)count_within(f32[j], f32[i], r, r2) / (double)L[j].n();
                ratio = std::max(r1, r2v);
            }
            if (ratio > th) {
                const int a = find((int)i), b = find((int)j);
                if (a != b) parent[(size_t)std::max(a, b)] = std::min(a, b);
            }
        }
    std::vector<Cloud> out;
    std::vector<int> comp_of(n, -1);
    std::vector<std::vector<size_t>> mem;
    for (size_t i = 0; i < n; ++i) {
        const int rt = find((int)i);
        if (comp_of[(size_t)rt] < 0) {
            comp_of[(size_t)rt] = (int)mem.size();
            mem.emplace_back();
        }
        mem[(size_t)comp_of[(size_t)rt]].push_back(i);
    }
    for (auto& m : mem) {               // merge_point_clouds_list (graph_utils.py:667-679): singletons too
        Cloud c;
        for (size_t i : m) {
            c.p.insert(c.p.end(), L[i].p.begin(), L[i].p.end());
            c.c.insert(c.c.end(), L[i].c.begin(), L[i].c.end());
        }
        pcd_denoise(c, 0.1, 10);
        out.push_back(std::move(c));
    }
    L.swap(out);
}

struct Cfg {
    int32_t H, W, D, outlier_nb, feat_dbscan_min, merge_hierarchical;
    double voxel_size, masked_weight, max_mask_distance, init_overlap_thresh, iou_thresh, outlier_radius, overlap_thresh_factor;
};

inline float f16_round(float v) { return _cvtsh_ss(_cvtss_sh(v, _MM_FROUND_TO_NEAREST_INT)); }

}  // namespace

extern "C" {

// create_feature_map (graph.py:262-491) on F frames: rgb u8 [F][H][W][3], depth u16 [F][H][W], pose f64 [F][16], K f64 [9],
// masks u8 [F][M][H][W], n_masks i32 [F], f_g f32 [F][D], f_masked / f_crop f32 [F][M][D].  Returns a handle.
void* hmsg_cpu_build(const Cfg* cfg, int32_t F, int32_t M, const uint8_t* rgb, const uint16_t* depth, const double* pose, const double* K,
                     const uint8_t* masks, const int32_t* n_masks, const float* f_g, const float* f_masked, const float* f_crop) {
    Ctx* cx = new Ctx();
    const int H = cfg->H, W = cfg->W, D = cfg->D;
    cx->H = H;
    cx->W = W;
    cx->D = D;
    const size_t HW = (size_t)H * W;
    // ---- A1 + A2: graph.py:339-358
    {
        VD all_p, all_c, p, c;
        for (int f = 0; f < F; ++f) {
            create_pcd(depth + (size_t)f * HW, rgb + (size_t)f * HW * 3, nullptr, H, W, pose + (size_t)f * 16, K, 1e300, p, &c, nullptr);
            all_p.insert(all_p.end(), p.begin(), p.end());
            all_c.insert(all_c.end(), c.begin(), c.end());
        }
        Cloud g;
        voxel_down_sample(all_p, &all_c, cfg->voxel_size, g.p, &g.c);
        pcd_denoise(g, 0.01, 100);                                         // graph.py:353 (no-op at the shipped voxel sizes)
        // remove_radius_outlier (graph.py:355-358): keep i iff #{j: d2 < r^2} (self included) > nb
        const size_t n = g.n();
        Grid grid;
        grid.build(g.p.data(), n, cfg->outlier_radius / 4.0);
        const double r2 = cfg->outlier_radius * cfg->outlier_radius;
        std::vector<char> keep(n, 0);
#pragma omp parallel for schedule(dynamic, 64)
        for (long long i = 0; i < (long long)n; ++i) {
            int cnt = 0;
            grid.visit(&g.p[(size_t)i * 3], 4, [&](int j) {
                cnt += d2_np(&g.p[(size_t)i * 3], &g.p[(size_t)j * 3]) < r2 ? 1 : 0;
                return cnt <= cfg->outlier_nb;                              // (the count only matters up to nb + 1)
            });
            keep[(size_t)i] = cnt > cfg->outlier_nb;
        }
        for (size_t i = 0; i < n; ++i)
            if (keep[i])
                for (int k = 0; k < 3; ++k) {
                    cx->map_p.push_back(g.p[i * 3 + k]);
                    cx->map_c.push_back(g.c[i * 3 + k]);
                }
    }
    const size_t V = cx->map_p.size() / 3;
    cx->tree.build(cx->map_p.data(), (int64_t)V);                          // graph.py:362-364
    std::vector<float> sum(V * (size_t)D, 0.f), counter(V, 0.f);
    // ---- loop B: graph.py:373-411
    cx->frame_first.assign(1, 0);
    std::vector<float> tmp;
    for (int f = 0; f < F; ++f) {
        const int nm = n_masks ? n_masks[f] : M;
        const float* fg = f_g + (size_t)f * D;
        // fuse_mask_feats (sam_clip_feats_extractor.py:159-175), float32
        std::vector<float> fp((size_t)nm * D);
        if (nm > 0) {
            std::vector<float> fl((size_t)nm * D), phi((size_t)nm);
            const float wm = (float)cfg->masked_weight, wc = (float)(1 - cfg->masked_weight);
            const float ng = std::max(std::sqrt(sumsq_f32(fg, (size_t)D, tmp)), 1e-6f);
            for (int i = 0; i < nm; ++i) {
                const float* a = f_masked + ((size_t)f * M + i) * D;
                const float* b = f_crop + ((size_t)f * M + i) * D;
                float* l = &fl[(size_t)i * D];
                for (int d = 0; d < D; ++d) l[d] = wm * a[d] + wc * b[d];
                const float nrm = std::max(std::sqrt(sumsq_f32(l, (size_t)D, tmp)), 1e-12f);
                for (int d = 0; d < D; ++d) l[d] = l[d] / nrm;
                const float nl = std::max(std::sqrt(sumsq_f32(l, (size_t)D, tmp)), 1e-6f);
                tmp.resize((size_t)D);
                for (int d = 0; d < D; ++d) tmp[(size_t)d] = (l[d] / nl) * (fg[d] / ng);
                phi[(size_t)i] = np_pairwise_sum_f32(tmp.data(), (size_t)D);
            }
            float mxp = phi[0];
            for (int i = 1; i < nm; ++i) mxp = std::max(mxp, phi[(size_t)i]);
            std::vector<float> e((size_t)nm);
            for (int i = 0; i < nm; ++i) e[(size_t)i] = std::exp(phi[(size_t)i] - mxp);
            const float es = np_pairwise_sum_f32(e.data(), (size_t)nm);
            for (int i = 0; i < nm; ++i) {
                const float w = e[(size_t)i] / es;
                float* o = &fp[(size_t)i * D];
                const float* l = &fl[(size_t)i * D];
                for (int d = 0; d < D; ++d) o[d] = w * fg[d] + (1.0f - w) * l[d];
                const float nrm = std::max(std::sqrt(sumsq_f32(o, (size_t)D, tmp)), 1e-12f);
                for (int d = 0; d < D; ++d) o[d] = o[d] / nrm;
            }
        }
        // the frame's points and their nearest map points (graph.py:390, :409: cKDTree.query, workers=-1)
        VD p;
        std::vector<int> pix;
        create_pcd(depth + (size_t)f * HW, nullptr, nullptr, H, W, pose + (size_t)f * 16, K, 1e300, p, nullptr, &pix);
        const size_t np_ = pix.size();
        std::vector<int64_t> nn(np_);
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < (long long)np_; ++i) nn[(size_t)i] = cx->tree.query1(&p[(size_t)i * 3]);
        std::vector<int64_t> nn_img(HW, -1);
        for (size_t i = 0; i < np_; ++i) nn_img[(size_t)pix[i]] = nn[i];
        // 3-D masks (generic.py:140-190): the masked depth's points snap to the same neighbours, then voxel_down_sample
        const uint8_t* mk = masks + (size_t)f * M * HW;
        for (int i = 0; i < nm; ++i) {
            Cloud snapped;
            double zs = 0;
            size_t nz = 0;
            for (size_t q = 0; q < HW; ++q)
                if (mk[(size_t)i * HW + q] && depth[(size_t)f * HW + q] > 0) {
                    const int64_t v = nn_img[q];
                    for (int k = 0; k < 3; ++k) {
                        snapped.p.push_back(cx->map_p[(size_t)v * 3 + k]);
                        snapped.c.push_back(cx->map_c[(size_t)v * 3 + k]);
                    }
                    zs += (float)depth[(size_t)f * HW + q] / 1000.0f;
                    ++nz;
                }
            Cloud m;
            if (nz && !(zs / (double)nz > cfg->max_mask_distance)) voxel_down_sample(snapped.p, &snapped.c, cfg->voxel_size, m.p, &m.c);
            cx->frames_masks.push_back(std::move(m));
        }
        cx->frame_first.push_back((int)cx->frames_masks.size());
        // per-pixel features (sam_clip_feats_extractor.py:178-190) and sum[idx] += F with the LAST pixel winning (graph.py:403-411)
        std::vector<int64_t> last(V, -1);
        for (size_t i = 0; i < np_; ++i) last[(size_t)nn[i]] = (int64_t)i;
        std::vector<int64_t> touched;
        for (size_t v = 0; v < V; ++v)
            if (last[v] >= 0) touched.push_back((int64_t)v);
        // (the reference builds the whole fp16 [H, W, D] image; every pixel's row is computed here too)
        std::vector<float> img(np_ * (size_t)D);
#pragma omp parallel
        {
            std::vector<float> t2;
#pragma omp for schedule(static)
            for (long long i = 0; i < (long long)np_; ++i) {
                float* o = &img[(size_t)i * D];
                for (int d = 0; d < D; ++d) o[d] = 0.f;
                const size_t q = (size_t)pix[(size_t)i];
                for (int m = 0; m < nm; ++m)
                    if (mk[(size_t)m * HW + q]) {
                        const float* s = &fp[(size_t)m * D];
                        for (int d = 0; d < D; ++d) o[d] += s[d];
                    }
                const float nrm = std::max(std::sqrt(sumsq_f32(o, (size_t)D, t2)), 1e-12f);
                for (int d = 0; d < D; ++d) o[d] = f16_round(o[d] / nrm);
            }
        }
        for (int64_t v : touched) {
            const float* s = &img[(size_t)last[(size_t)v] * D];
            float* o = &sum[(size_t)v * D];
            for (int d = 0; d < D; ++d) o[d] += s[d];
            counter[(size_t)v] += 1.0f;
        }
    }
    cx->full_feats.resize(V * (size_t)D);
    for (size_t v = 0; v < V; ++v) {
        const float c = counter[v] == 0.f ? 1e-5f : counter[v];
        for (int d = 0; d < D; ++d) cx->full_feats[v * D + d] = sum[v * D + d] / c;
    }
    // ---- A6 seq_merge (graph_utils.py:1015-1038)
    std::vector<Cloud> G;
    if (cfg->merge_hierarchical) {
        // hierarchical_merge (graph_utils.py:959-1012): adjacent lists pairwise, level by level, the threshold lowered per level
        std::vector<std::vector<Cloud>> lv((size_t)F);
        for (int f = 0; f < F; ++f)
            for (int i = cx->frame_first[(size_t)f]; i < cx->frame_first[(size_t)f + 1]; ++i) lv[(size_t)f].push_back(cx->frames_masks[(size_t)i]);
        double th = cfg->init_overlap_thresh;
        while (lv.size() > 1) {
            std::vector<std::vector<Cloud>> nx;
            for (size_t i = 0; i < lv.size(); i += 2) {
                if (i == lv.size() - 1) {
                    nx.push_back(std::move(lv[i]));
                    break;
                }
                std::vector<Cloud> L = std::move(lv[i]);
                L.insert(L.end(), lv[i + 1].begin(), lv[i + 1].end());
                merge_3d_masks(L, th, cfg->voxel_size, cfg->iou_thresh);
                nx.push_back(std::move(L));
            }
            lv = std::move(nx);
            if (lv.size() > 1) th -= cfg->overlap_thresh_factor * (double)((long long)lv.size() - 2) / (double)std::max<long long>(1, (long long)lv.size() - 1);
        }
        G = std::move(lv[0]);
        merge_3d_masks(G, 0.75, cfg->voxel_size, cfg->iou_thresh);
    } else {
        for (int f = 0; f < F; ++f) {
            for (int i = cx->frame_first[(size_t)f]; i < cx->frame_first[(size_t)f + 1]; ++i) G.push_back(cx->frames_masks[(size_t)i]);
            if (f > 0) merge_3d_masks(G, cfg->init_overlap_thresh, cfg->voxel_size, cfg->iou_thresh);
        }
        merge_3d_masks(G, cfg->init_overlap_thresh, cfg->voxel_size, cfg->iou_thresh);
    }
    for (auto& c : G)
        if (c.n() >= 10) cx->inst.push_back(c);                            // graph.py:445-448
    // ---- A7 pooling (graph.py:450-488, graph_utils.py:682-728)
    cx->inst_feats.assign(cx->inst.size() * (size_t)D, 0.f);
    for (size_t k = 0; k < cx->inst.size(); ++k) {
        VD dp;
        voxel_down_sample(cx->inst[k].p, nullptr, cfg->voxel_size, dp, nullptr);
        std::vector<const float*> rows;
        for (size_t i = 0; i < dp.size() / 3; ++i) {
            double d2 = 0;
            const int64_t v = cx->tree.query1(&dp[i * 3], &d2);
            if (std::sqrt(d2) <= 0.8) rows.push_back(&cx->full_feats[(size_t)v * D]);
        }
        float* out = &cx->inst_feats[k * D];
        const size_t n = rows.size();
        if (n == 0) continue;                                              // zeros(1, D)
        // sklearn DBSCAN(eps 0.01, min_samples, metric "cosine"): 1 - x^.y^ (float32), neighbour iff d <= eps (self included)
        std::vector<float> xn(n * (size_t)D);
        for (size_t i = 0; i < n; ++i) {
            std::vector<float> t2;
            float nrm = std::sqrt(sumsq_f32(rows[i], (size_t)D, t2));
            if (nrm == 0.f) nrm = 1.f;
            for (int d = 0; d < D; ++d) {
                float v = rows[i][d];
                if (v != v) v = 0.f;                                       // nan_to_num
                xn[i * D + d] = v / nrm;
            }
        }
        std::vector<std::vector<int>> nb(n);
#pragma omp parallel for schedule(dynamic, 16) if (n > 256)
        for (long long i = 0; i < (long long)n; ++i)
            for (size_t j = 0; j < n; ++j) {
                float s = 0.f;
                for (int d = 0; d < D; ++d) s += xn[(size_t)i * D + d] * xn[j * D + d];
                float dist = 1.0f - s;
                dist = std::min(std::max(dist, 0.f), 2.f);
                if ((size_t)i == j) dist = 0.f;
                if (dist <= 0.01f) nb[(size_t)i].push_back((int)j);
            }
        std::vector<int64_t> lab(n, -1);
        std::vector<char> core(n);
        for (size_t i = 0; i < n; ++i) core[i] = (int)nb[i].size() >= cfg->feat_dbscan_min;
        int64_t next = 0;
        std::vector<int> stack;
        for (size_t i = 0; i < n; ++i) {                                   // sklearn dbscan_inner
            if (lab[i] != -1 || !core[i]) continue;
            int cur = (int)i;
            for (;;) {
                if (lab[(size_t)cur] == -1) {
                    lab[(size_t)cur] = next;
                    if (core[(size_t)cur])
                        for (int v : nb[(size_t)cur])
                            if (lab[(size_t)v] == -1) stack.push_back(v);
                }
                if (stack.empty()) break;
                cur = stack.back();
                stack.pop_back();
            }
            ++next;
        }
        const int64_t best = largest_label(lab);
        std::vector<size_t> sel;
        for (size_t i = 0; i < n; ++i)
            if (best < 0 || lab[i] == best) sel.push_back(i);
        for (int d = 0; d < D; ++d) {                                      // np.mean(axis = 0): rows added in order, float32
            float s = 0.f;
            for (size_t i : sel) {
                float v = rows[i][d];
                if (v != v) v = 0.f;
                s += v;
            }
            out[d] = sel.size() > 1 ? s / (float)sel.size() : s;
        }
    }
    return cx;
}

int64_t hmsg_cpu_map_size(void* h) { return (int64_t)(((Ctx*)h)->map_p.size() / 3); }
void hmsg_cpu_get_map(void* h, double* pts) { memcpy(pts, ((Ctx*)h)->map_p.data(), ((Ctx*)h)->map_p.size() * 8); }
void hmsg_cpu_get_full_feats(void* h, float* f) { memcpy(f, ((Ctx*)h)->full_feats.data(), ((Ctx*)h)->full_feats.size() * 4); }
int64_t hmsg_cpu_num_masks(void* h) { return (int64_t)((Ctx*)h)->frames_masks.size(); }
void hmsg_cpu_mask_sizes(void* h, int64_t* s) {
    Ctx* c = (Ctx*)h;
    for (size_t i = 0; i < c->frames_masks.size(); ++i) s[i] = (int64_t)c->frames_masks[i].n();
}
void hmsg_cpu_mask_points(void* h, double* p) {
    Ctx* c = (Ctx*)h;
    for (auto& m : c->frames_masks) {
        memcpy(p, m.p.data(), m.p.size() * 8);
        p += m.p.size();
    }
}
int64_t hmsg_cpu_num_instances(void* h) { return (int64_t)((Ctx*)h)->inst.size(); }
void hmsg_cpu_instance_sizes(void* h, int64_t* s) {
    Ctx* c = (Ctx*)h;
    for (size_t i = 0; i < c->inst.size(); ++i) s[i] = (int64_t)c->inst[i].n();
}
void hmsg_cpu_instance_points(void* h, double* p) {
    Ctx* c = (Ctx*)h;
    for (auto& m : c->inst) {
        memcpy(p, m.p.data(), m.p.size() * 8);
        p += m.p.size();
    }
}
void hmsg_cpu_instance_feats(void* h, float* f) { memcpy(f, ((Ctx*)h)->inst_feats.data(), ((Ctx*)h)->inst_feats.size() * 4); }
// query_hmsg_object (graph.py:3112-3151) over a node table feats f32 [N][D]: text f32 [Q][C][D] (row qid the query, the others
// negatives).  The reference answers one query at a time with `text @ feats.T` (BLAS, all cores); here the queries of a batch
// run side by side on the OpenMP threads -- the same arithmetic per query (float64 dot products in index order), and the more
// generous form for the CPU side of bench.py's queries/s.
void hmsg_cpu_query_table(const float* feats, int64_t N_, int32_t D, int32_t Q, int32_t C, const float* text, int32_t qid, int32_t k,
                          int32_t* out_idx, double* out_score) {
    const size_t N = (size_t)N_;
#pragma omp parallel for schedule(dynamic, 4)
    for (int q = 0; q < Q; ++q) {
        std::vector<double> sim((size_t)C * N);
        for (int r = 0; r < C; ++r)
            for (size_t n = 0; n < N; ++n) {
                double s = 0;
                for (int d = 0; d < D; ++d) s += (double)text[((size_t)q * C + r) * D + d] * (double)feats[n * D + d];
                sim[(size_t)r * N + n] = s;
            }
        std::vector<int> ids;
        for (size_t n = 0; n < N; ++n) {
            int cls = 0;
            for (int r = 1; r < C; ++r)
                if (sim[(size_t)r * N + n] > sim[(size_t)cls * N + n]) cls = r;
            if (cls == qid) ids.push_back((int)n);
        }
        if (ids.empty()) {
            ids.resize(N);
            std::iota(ids.begin(), ids.end(), 0);
        }
        std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) { return sim[(size_t)qid * N + a] > sim[(size_t)qid * N + b]; });
        for (int j = 0; j < k; ++j) {
            out_idx[(size_t)q * k + j] = j < (int)ids.size() ? ids[(size_t)j] : -1;
            out_score[(size_t)q * k + j] = j < (int)ids.size() ? sim[(size_t)qid * N + ids[(size_t)j]] : 0.0;
        }
    }
}
// ... over the instances of a build
void hmsg_cpu_query(void* h, int32_t Q, int32_t C, const float* text, int32_t qid, int32_t k, int32_t* out_idx, double* out_score) {
    Ctx* c = (Ctx*)h;
    hmsg_cpu_query_table(c->inst_feats.data(), (int64_t)c->inst.size(), c->D, Q, C, text, qid, k, out_idx, out_score);
}
void hmsg_cpu_free(void* h) { delete (Ctx*)h; }
int32_t hmsg_cpu_threads(void) { return (int32_t)omp_get_max_threads(); }
}
