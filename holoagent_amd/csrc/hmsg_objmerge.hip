// N2: Room.merge_objects (fsr_vln/memory/hmsg/graph/room.py:62-129) -- the optional post-pass of
// build_hier_multimodal_scene_graph (graph.py:2053-2058, pipeline.merge_objects_graph) that fuses objects of one room which carry
// the same name and whose clouds overlap.
//
//   device : find_overlapping_ratio_faiss (utils/graph_utils.py:620-664) of every same-name pair, both directions in one launch:
//            a point of A overlaps when SOME point of B has float32 (dx*dx + dy*dy) + dz*dz < radius^2 (= its exact nearest
//            neighbour is that close) -- object clouds are a few hundred to a few thousand points, so the pair is evaluated
//            exhaustively, B tiled through LDS, a lane per point of A, early exit per lane and per workgroup.  faiss's BLAS
//            evaluation for 20 or more queries is honoured like in the merge fold (hmsg_config::overlap_distance_form).
//   host   : what the reference's bookkeeping does with the pairs, step by step: `np.where(scores > 0)` in row-major order, the
//            dictionary chaining (an index that was already absorbed can still become a key: kept), the objects not involved
//            appended in index order, and `list(set(j))` -- CPython's set iteration order, restated below.
#include "hmsg_common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

__device__ __forceinline__ float om_norm2(float a, float b, float c) { return __fadd_rn(__fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b)), __fmul_rn(c, c)); }
// (the two distance forms of hmsg_merge.hip: ov_dist2)
template <bool BLAS>
__device__ __forceinline__ float om_dist2(float x, float y, float z, float qx, float qy, float qz, float nx) {
    if (!BLAS) {
        const float ddx = __fsub_rn(x, qx), ddy = __fsub_rn(y, qy), ddz = __fsub_rn(z, qz);
        return __fadd_rn(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)), __fmul_rn(ddz, ddz));
    }
    const float ip = __fmaf_rn(z, qz, __fmaf_rn(y, qy, __fmul_rn(x, qx)));
    const float d = __fsub_rn(__fadd_rn(nx, om_norm2(qx, qy, qz)), __fmul_rn(2.0f, ip));
    return d < 0.f ? 0.f : d;
}

bool om_device_ptr(const void* p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

struct OmTask {             // count the points of cloud x that have a point of cloud y within the radius
    long long x0, y0;       // first points (in the points buffer)
    int nx, ny;
    int blk0, pad;          // first workgroup of the task (work list: ceil(nx / 256) workgroups)
};

#define OM_TILE 256
template <bool BLAS>
__global__ void __launch_bounds__(256) k_om_overlap(const double* __restrict__ pts, const OmTask* __restrict__ tasks, int ntasks, float r2,
                                                    unsigned* __restrict__ counts) {
    __shared__ float sy[OM_TILE][3];
    __shared__ int s_left;
    int lo = 0, hi = ntasks - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((unsigned)tasks[mid].blk0 <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const OmTask t = tasks[lo];
    const int i = (int)(blockIdx.x - (unsigned)t.blk0) * 256 + (int)threadIdx.x;
    const bool live = i < t.nx;
    float x = 0.f, y = 0.f, z = 0.f;
    if (live) {
        const double* p = pts + (size_t)(t.x0 + i) * 3;
        x = (float)p[0], y = (float)p[1], z = (float)p[2];
    }
    const bool blas = BLAS && t.nx >= 20;       // (faiss: the BLAS route from 20 queries on -- the cloud whose points are looked up)
    const float nx = BLAS ? om_norm2(x, y, z) : 0.f;
    bool hit = false;
    for (int y0 = 0; y0 < t.ny; y0 += OM_TILE) {
        __syncthreads();
        if (threadIdx.x == 0) s_left = 0;
        const int j = y0 + (int)threadIdx.x;
        if (j < t.ny) {
            const double* q = pts + (size_t)(t.y0 + j) * 3;
            sy[threadIdx.x][0] = (float)q[0];
            sy[threadIdx.x][1] = (float)q[1];
            sy[threadIdx.x][2] = (float)q[2];
        }
        __syncthreads();
        const int m = min(OM_TILE, t.ny - y0);
        if (live && !hit) {
            for (int k = 0; k < m && !hit; ++k) {
                const float d2 = (BLAS && blas) ? om_dist2<true>(x, y, z, sy[k][0], sy[k][1], sy[k][2], nx)
                                                : om_dist2<false>(x, y, z, sy[k][0], sy[k][1], sy[k][2], nx);
                hit = d2 < r2;
            }
            if (!hit) s_left = 1;               // (benign race: everybody writes 1)
        }
        __syncthreads();
        if (!s_left) break;                     // every point of this workgroup has its witness
    }
    const unsigned long long m = __ballot(live && hit);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&counts[lo], (unsigned)__popcll(m));
}

// list(set(js)) as CPython (3.7 .. 3.12: Objects/setobject.c) iterates it for non-negative ints (hash(i) == i): a table of 8 slots
// that quadruples when fill * 5 >= mask * 3; a key goes to slot hash & mask, then up to LINEAR_PROBES = 9 following slots while they
// stay inside the table, then i = i * 5 + 1 + (perturb >>= 5).  Iteration = the slots in order.
struct PySet {
    std::vector<long long> slot;    // -1 = empty
    size_t mask = 7, used = 0;
    PySet() : slot(8, -1) {}
    static bool put(std::vector<long long>& tab, size_t mask, long long key) {
        size_t perturb = (size_t)key, i = (size_t)key & mask;
        for (;;) {
            const size_t probes = i + 9 <= mask ? 9 : 0;
            for (size_t j = 0; j <= probes; ++j) {
                long long& e = tab[i + j];
                if (e == key) return false;
                if (e < 0) {
                    e = key;
                    return true;
                }
            }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    void add(long long key) {
        if (!put(slot, mask, key)) return;
        ++used;
        if (used * 5 < mask * 3) return;
        // set_table_resize(so, used > 50000 ? used * 2 : used * 4): the smallest power of two above it, from 8
        const size_t want = used > 50000 ? used * 2 : used * 4;
        size_t nsz = 8;
        while (nsz <= want) nsz <<= 1;
        std::vector<long long> nt(nsz, -1);
        for (long long e : slot)
            if (e >= 0) put(nt, nsz - 1, e);
        slot.swap(nt);
        mask = nsz - 1;
    }
    std::vector<int> order() const {
        std::vector<int> o;
        for (long long e : slot)
            if (e >= 0) o.push_back((int)e);
        return o;
    }
};

}  // namespace

// overlap[p] = find_overlapping_ratio_faiss(cloud a[p], cloud b[p], radius) for n_pairs pairs of the clouds
// pts[start[i] .. start[i] + count[i]) (points f64, host or device memory).
void hmsg_pair_overlaps(hmsg_ctx* h, const double* points, const long long* start, const int* count, int n_pairs, const int* pa, const int* pb,
                        double radius, double* overlap) {
    if (n_pairs <= 0) return;
    hipStream_t s = h->stream;
    std::vector<OmTask> tasks;
    unsigned nblk = 0;
    long long lo = -1, hi = 0;
    for (int p = 0; p < n_pairs; ++p)
        for (int dir = 0; dir < 2; ++dir) {
            const int x = dir ? pb[p] : pa[p], y = dir ? pa[p] : pb[p];
            tasks.push_back(OmTask{start[x], start[y], count[x], count[y], (int)nblk, 0});
            nblk += (unsigned)std::max(1, (count[x] + 255) / 256);
            for (int v : {x, y}) {
                lo = lo < 0 ? start[v] : std::min(lo, start[v]);
                hi = std::max(hi, start[v] + count[v]);
            }
        }
    const double* dpts = points;
    DevBuf<double> up;
    if (!om_device_ptr(points)) {               // host clouds: the span the pairs touch goes up once
        up.alloc((size_t)std::max<long long>(hi - lo, 1) * 3);
        HIP_TRY(hipMemcpyAsync(up.p, points + (size_t)lo * 3, (size_t)(hi - lo) * 24, hipMemcpyHostToDevice, s));
        dpts = up.p - (size_t)lo * 3;
    }
    DevBuf<OmTask> dt;
    DevBuf<unsigned> dc;
    dt.alloc(tasks.size());
    dc.alloc(tasks.size());
    HIP_TRY(hipMemcpyAsync(dt.p, tasks.data(), tasks.size() * sizeof(OmTask), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(dc.p, 0, tasks.size() * 4, s));
    const float r2 = (float)(radius * radius);  // `D < radius**2` with a float32 D (graph_utils.py:654-655)
    if (h->cfg.overlap_distance_form == HMSG_OVERLAP_FAISS_BLAS)
        hipLaunchKernelGGL(k_om_overlap<true>, dim3(nblk), dim3(256), 0, s, dpts, (const OmTask*)dt.p, (int)tasks.size(), r2, dc.p);
    else
        hipLaunchKernelGGL(k_om_overlap<false>, dim3(nblk), dim3(256), 0, s, dpts, (const OmTask*)dt.p, (int)tasks.size(), r2, dc.p);
    HMSG_CHECK_LAUNCH();
    std::vector<unsigned> hc(tasks.size());
    HIP_TRY(hipMemcpyAsync(hc.data(), dc.p, hc.size() * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int p = 0; p < n_pairs; ++p) {
        // np.max([np.sum(D1 < r2) / n1, np.sum(D2 < r2) / n2]); an empty cloud: 0 (graph_utils.py:636-637)
        const int na = count[pa[p]], nb = count[pb[p]];
        overlap[p] = (na == 0 || nb == 0) ? 0.0 : std::max((double)hc[(size_t)p * 2] / (double)na, (double)hc[(size_t)p * 2 + 1] / (double)nb);
    }
}

// room.py:62-129 for the n objects of one room: groups in the order the reference's dictionary yields them; group g = members
// [group_off[g], group_off[g + 1]): the key object first, then the objects the reference adds to it, in its order.
void hmsg_merge_groups(hmsg_ctx* h, int n, const double* points, const long long* start, const int* count, const int* name_id,
                       double overlap_threshold, double radius, std::vector<int>& group_off, std::vector<int>& members) {
    std::vector<int> pa, pb;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (name_id[i] == name_id[j]) {
                pa.push_back(i);
                pb.push_back(j);
            }
    std::vector<double> ov(pa.size());
    hmsg_pair_overlaps(h, points, start, count, (int)pa.size(), pa.data(), pb.data(), radius, ov.data());
    std::vector<unsigned char> score((size_t)n * n, 0);
    for (size_t p = 0; p < pa.size(); ++p)
        if (ov[p] > overlap_threshold && ov[p] > 0.0)             // (scores[i, j] = overlap, read back as `scores > 0`)
            score[(size_t)pa[p] * n + pb[p]] = score[(size_t)pb[p] * n + pa[p]] = 1;
    // the dictionary: keys in insertion order
    std::vector<int> keys;
    std::vector<std::vector<int>> lists;
    std::vector<int> key_pos(n, -1);
    std::vector<unsigned char> merging(n, 0);
    auto list_of = [&](int key) -> std::vector<int>& {
        if (key_pos[key] < 0) {
            key_pos[key] = (int)keys.size();
            keys.push_back(key);
            lists.emplace_back();
        }
        return lists[(size_t)key_pos[key]];
    };
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (!score[(size_t)i * n + j]) continue;
            merging[i] = merging[j] = 1;
            if (key_pos[i] < 0 && key_pos[j] < 0) list_of(i).push_back(j);
            else if (key_pos[i] >= 0) list_of(i).push_back(j);
            else list_of(j).push_back(i);
        }
    for (int idx = 0; idx < n; ++idx)
        if (!merging[idx]) list_of(idx).push_back(idx);
    group_off.assign(1, 0);
    members.clear();
    for (size_t g = 0; g < keys.size(); ++g) {
        PySet st;
        for (int v : lists[g]) st.add(v);
        const std::vector<int> js = st.order();
        members.push_back(keys[g]);
        if (!(js.size() == 1 && js[0] == keys[g]))
            for (int v : js) members.push_back(v);
        group_off.push_back((int)members.size());
    }
}

extern "C" int hmsg_merge_room_objects(hmsg_t* h, int32_t n, const double* points, const int64_t* off, const int32_t* name_id,
                                       double overlap_threshold, double radius, int32_t* n_groups, int32_t* group_off, int32_t* group_members,
                                       int32_t members_capacity) {
    if (!h || n < 0 || (n > 0 && (!points || !off || !name_id)) || !n_groups || !group_off || !group_members) return HMSG_ERR_INVALID;
    try {
        HIP_TRY(hipSetDevice(h->cfg.device_id));
        std::vector<long long> start((size_t)n);
        std::vector<int> count((size_t)n);
        for (int i = 0; i < n; ++i) {
            start[(size_t)i] = off[i];
            count[(size_t)i] = (int)(off[i + 1] - off[i]);
        }
        std::vector<int> go, mem;
        hmsg_merge_groups(h, n, points, start.data(), count.data(), name_id, overlap_threshold, radius, go, mem);
        if ((int64_t)mem.size() > (int64_t)members_capacity) {
            h->err = "hmsg_merge_room_objects: group_members too small (n (n + 1) entries always suffice)";
            return HMSG_ERR_INVALID;
        }
        *n_groups = (int32_t)go.size() - 1;
        std::copy(go.begin(), go.end(), group_off);
        std::copy(mem.begin(), mem.end(), group_members);
        return HMSG_OK;
    } catch (const hmsg_error& e) {
        h->err = e.msg;
        return e.code;
    } catch (const std::exception& e) {
        h->err = e.what();
        return HMSG_ERR_INVALID;
    } catch (...) {
        h->err = "unknown error";
        return HMSG_ERR_INVALID;
    }
}
