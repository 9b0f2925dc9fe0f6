// A8 behind the C ABI: the storeys of the map -- Graph.segment_floors_manually
// (fsr_vln/memory/hmsg/graph/graph.py:624-787) for a host that is not Python.
//
// Device: the 5 cm re-sampling of the map (Open3D voxel_down_sample, :633), the height histogram over it
// (np.histogram, :640-642) and the per-storey crop statistics (:769-787).  Host (C++, a few hundred bins): scipy's
// gaussian_filter1d on the int64 counts (:645; int64 in -> int64 out, truncated), np.percentile(.., 90), scipy's
// find_peaks(distance = 0.2 / 0.01, height = that percentile) (:648-651), the 1-D DBSCAN(eps = 1, min_samples = 1)
// chaining of the peak heights and the reference's pick / adjust rules (:680-741).  The Python mirror
// (holoagent_amd/graph.py segment_floors_manually) calls numpy / scipy for the same steps; tests hold the two equal.
//
// One documented freedom: find_peaks orders peaks of equal height with numpy's default argsort (introsort / AVX-512
// network, not stable); here ties are ordered by position.  It matters only when two equally high peaks sit within
// 20 bins of each other.
#include "hmsg_cloudops.h"
#include "hmsg_common.h"

#include <cmath>

namespace {

__device__ __forceinline__ unsigned long long fl_key(double d) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | (1ull << 63));
}
inline double fl_unkey(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & ~(1ull << 63)) : ~k;
    double d;
    memcpy(&d, &u, 8);
    return d;
}
__device__ __forceinline__ unsigned long long fl_wmin(unsigned long long v) {
    for (int o = 32; o; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o);
        v = w < v ? w : v;
    }
    return v;
}
__device__ __forceinline__ unsigned long long fl_wmax(unsigned long long v) {
    for (int o = 32; o; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o);
        v = w > v ? w : v;
    }
    return v;
}

// out[0], out[1] = min / max key of pts[i][1]
__global__ void __launch_bounds__(256) k_fl_yrange(const double* __restrict__ pts, long long n, unsigned long long* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n;
    const unsigned long long k = in ? fl_key(pts[i * 3 + 1]) : 0;
    const unsigned long long lo = fl_wmin(in ? k : ~0ull), hi = fl_wmax(in ? k : 0ull);
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&out[0], lo);
        atomicMax(&out[1], hi);
    }
}
// np.histogram(y, bins): edges = linspace(lo, hi, bins + 1); the last edge belongs to the last bin
__global__ void __launch_bounds__(256) k_fl_hist(const double* __restrict__ pts, long long n, double lo, double hi, int bins,
                                                 unsigned long long* __restrict__ hist) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = pts[i * 3 + 1], step = (hi - lo) / (double)bins;
    int k = (int)floor((v - lo) / step);
    k = k < 0 ? 0 : (k > bins ? bins : k);
    auto edge = [&](int j) { return j == bins ? hi : (double)j * step + lo; };
    while (k > 0 && v < edge(k)) --k;
    while (k < bins && v >= edge(k + 1)) ++k;
    if (k >= bins) k = bins - 1;
    atomicAdd(&hist[k], 1ull);
}
// per storey f (slab [lo, hi] of y, both inclusive): count and AABB keys; st[f * 8 + 0] count, +1..+3 min keys, +4..+6 max keys
__global__ void __launch_bounds__(256) k_fl_crop(const double* __restrict__ pts, long long n, int n_floors, const double* __restrict__ slabs,
                                                 unsigned long long* __restrict__ st) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double p[3] = {0, 0, 0};
    if (i < n)
        for (int a = 0; a < 3; ++a) p[a] = pts[i * 3 + a];
    for (int f = 0; f < n_floors; ++f) {
        const bool in = i < n && p[1] >= slabs[f * 2] && p[1] <= slabs[f * 2 + 1];
        const unsigned long long cnt = __popcll(__ballot(in));
        if (!cnt) continue;
        unsigned long long mn[3], mx[3];
        for (int a = 0; a < 3; ++a) {
            const unsigned long long k = fl_key(p[a]);
            mn[a] = fl_wmin(in ? k : ~0ull);
            mx[a] = fl_wmax(in ? k : 0ull);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&st[f * 8], cnt);
            for (int a = 0; a < 3; ++a) {
                atomicMin(&st[f * 8 + 1 + a], mn[a]);
                atomicMax(&st[f * 8 + 4 + a], mx[a]);
            }
        }
    }
}

// scipy.ndimage.gaussian_filter1d(int64 counts, sigma): radius int(4 sigma + 0.5), weights exp(-x^2 / (2 sigma^2)) / sum,
// correlate1d with mode "reflect" (d c b a | a b c d), symmetric taps paired, result truncated to int64
std::vector<long long> gaussian_filter1d_i64(const std::vector<long long>& x, double sigma) {
    const int n = (int)x.size(), r = (int)(4.0 * sigma + 0.5);
    std::vector<double> w((size_t)(2 * r + 1));
    for (int i = -r; i <= r; ++i) w[(size_t)(i + r)] = std::exp(-0.5 / (sigma * sigma) * (double)(i * i));
    const double sum = np_sum_f64(w.data(), w.size());
    for (auto& v : w) v /= sum;
    auto at = [&](int i) {
        if (n == 1) return (double)x[0];
        while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - 1 - i;
        return (double)x[(size_t)i];
    };
    std::vector<long long> out((size_t)n);
    for (int i = 0; i < n; ++i) {
        double acc = at(i) * w[(size_t)r];
        for (int j = -r; j < 0; ++j) acc += (at(i + j) + at(i - j)) * w[(size_t)(j + r)];
        out[(size_t)i] = (long long)acc;
    }
    return out;
}

// np.percentile(x, q) (method "linear")
double percentile_linear(std::vector<long long> x, double q) {
    std::sort(x.begin(), x.end());
    const double quant = q / 100.0;
    const double virt = (double)x.size() * quant + (1.0 + quant * (1.0 - 1.0 - 1.0)) - 1.0;     // numpy's _compute_virtual_index(alpha = beta = 1)
    const double prev = std::floor(virt), gamma = virt - prev;
    const size_t i0 = (size_t)prev, i1 = std::min(i0 + 1, x.size() - 1);
    const double a = (double)x[i0], b = (double)x[i1], d = b - a;
    return gamma >= 0.5 ? b - d * (1.0 - gamma) : a + d * gamma;
}

// scipy.signal.find_peaks(x, distance, height): local maxima (plateau midpoints), height filter, then the distance rule by
// descending height
std::vector<int> find_peaks(const std::vector<long long>& x, double height, int distance) {
    const int n = (int)x.size();
    std::vector<int> peaks;
    for (int i = 1; i < n - 1; ++i) {
        if (x[(size_t)i - 1] < x[(size_t)i]) {
            int ahead = i + 1;
            while (ahead < n - 1 && x[(size_t)ahead] == x[(size_t)i]) ++ahead;
            if (x[(size_t)ahead] < x[(size_t)i]) {
                peaks.push_back((i + ahead - 1) / 2);
                i = ahead;
            }
        }
    }
    std::vector<int> kept;
    for (int p : peaks)
        if (height <= (double)x[(size_t)p]) kept.push_back(p);
    peaks.swap(kept);
    const int m = (int)peaks.size();
    std::vector<int> order((size_t)m);
    for (int i = 0; i < m; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return x[(size_t)peaks[(size_t)a]] < x[(size_t)peaks[(size_t)b]]; });
    std::vector<char> keep((size_t)m, 1);
    for (int i = m - 1; i >= 0; --i) {
        const int j = order[(size_t)i];
        if (!keep[(size_t)j]) continue;
        for (int k = j - 1; k >= 0 && peaks[(size_t)j] - peaks[(size_t)k] < distance; --k) keep[(size_t)k] = 0;
        for (int k = j + 1; k < m && peaks[(size_t)k] - peaks[(size_t)j] < distance; ++k) keep[(size_t)k] = 0;
    }
    std::vector<int> out;
    for (int i = 0; i < m; ++i)
        if (keep[(size_t)i]) out.push_back(peaks[(size_t)i]);
    return out;
}

}  // namespace

extern "C" int hmsg_segment_floors(hmsg_t* h, hmsg_floor* out, int32_t capacity, int32_t* n_floors) {
    if (!h) return HMSG_ERR_INVALID;
    try {
        HIP_TRY(hipSetDevice(h->cfg.device_id));
        HMSG_REQUIRE(h->map_ready && n_floors && (out || capacity == 0), HMSG_ERR_INVALID, "hmsg_segment_floors: bad argument (finalize the map first)");
        hipStream_t s = h->stream;
        const long long V = h->V;
        *n_floors = 0;
        HMSG_REQUIRE(V > 0 && V < (1ll << 31), HMSG_ERR_INVALID, "hmsg_segment_floors: empty (or too large) map");
        // ---- the map at 5 cm (graph.py:633), on the device
        DevBuf<double> down;
        down.alloc((size_t)V * 3);
        CloudOps ops;
        ops.s = s;
        std::vector<SegDesc> segs(1);
        segs[0].pt_base = 0;
        segs[0].n = (int)V;
        ops.bounds(h->pts.p, segs);
        std::vector<int> out_n;
        const long long nd = ops.voxel_down_sample(h->pts.p, segs, 0.05, down.p, out_n);
        HMSG_REQUIRE(nd > 0, HMSG_ERR_INVALID, "hmsg_segment_floors: empty map");
        // ---- height histogram (:640-642)
        DevBuf<unsigned long long> yr, hist, st;
        yr.alloc(2);
        const unsigned long long yr0[2] = {~0ull, 0ull};
        HIP_TRY(hipMemcpyAsync(yr.p, yr0, 16, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_fl_yrange, dim3(cdiv((size_t)nd, 256)), dim3(256), 0, s, (const double*)down.p, nd, yr.p);
        HMSG_CHECK_LAUNCH();
        unsigned long long hyr[2];
        HIP_TRY(hipMemcpyAsync(hyr, yr.p, 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const double ymin = fl_unkey(hyr[0]), ymax = fl_unkey(hyr[1]);
        const int bins = (int)(std::fabs(ymax - ymin) / 0.01);
        HMSG_REQUIRE(bins >= 1 && bins < (1 << 24), HMSG_ERR_INVALID, "hmsg_segment_floors: the map is less than a centimetre high");
        hist.alloc((size_t)bins);
        hist.zero(s);
        hipLaunchKernelGGL(k_fl_hist, dim3(cdiv((size_t)nd, 256)), dim3(256), 0, s, (const double*)down.p, nd, ymin, ymax, bins, hist.p);
        HMSG_CHECK_LAUNCH();
        std::vector<unsigned long long> hh((size_t)bins);
        HIP_TRY(hipMemcpyAsync(hh.data(), hist.p, (size_t)bins * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        std::vector<long long> counts(hh.begin(), hh.end());
        const double step = (ymax - ymin) / (double)bins;
        auto edge = [&](int j) { return j == bins ? ymax : (double)j * step + ymin; };
        // ---- peaks (:645-651)
        const std::vector<long long> smooth = gaussian_filter1d_i64(counts, 2.0);
        const std::vector<int> peaks = find_peaks(smooth, percentile_linear(smooth, 90.0), (int)std::ceil(0.2 / 0.01));
        const int np_ = (int)peaks.size();
        std::vector<double> locs((size_t)np_);
        for (int i = 0; i < np_; ++i) locs[(size_t)i] = edge(peaks[(size_t)i]);
        // ---- DBSCAN(eps = 1, min_samples = 1) on the peak heights = chains of gaps <= 1; sklearn numbers clusters by first
        // appearance (peaks come in ascending order, so that is the chain order) (:680-683)
        std::vector<int> label((size_t)np_, 0);
        for (int i = 1; i < np_; ++i) label[(size_t)i] = label[(size_t)i - 1] + (locs[(size_t)i] - locs[(size_t)i - 1] > 1.0 ? 1 : 0);
        const int n_lab = np_ ? label[(size_t)np_ - 1] + 1 : 0;
        // ---- per cluster the highest peak (first and last cluster) or the two highest (:690-713)
        std::vector<double> clustered;
        for (int c = 0; c < n_lab; ++c) {
            std::vector<int> pk;
            for (int i = 0; i < np_; ++i)
                if (label[(size_t)i] == c) pk.push_back(peaks[(size_t)i]);
            const int take = (c == 0 || c == n_lab - 1) ? 1 : 2;
            // np.argsort(smooth[pk])[-take:]: ascending, stable for these short lists
            std::vector<int> ord(pk.size());
            for (size_t i = 0; i < pk.size(); ++i) ord[i] = (int)i;
            std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return smooth[(size_t)pk[(size_t)a]] < smooth[(size_t)pk[(size_t)b]]; });
            for (size_t i = pk.size() > (size_t)take ? pk.size() - (size_t)take : 0; i < pk.size(); ++i) clustered.push_back(edge(pk[(size_t)ord[i]]));
        }
        std::sort(clustered.begin(), clustered.end());
        // ---- a 2.5 m gap between picks gets a ceiling 0.2 m under the upper one (:716-727)
        std::vector<double> adj;
        for (size_t i = 0; i + 1 < clustered.size(); ++i) {
            adj.push_back(clustered[i]);
            if (clustered[i + 1] - clustered[i] >= 2.5) adj.push_back(clustered[i + 1] - 0.2);
        }
        if (!clustered.empty()) adj.push_back(clustered.back());
        std::vector<double> slabs;
        for (size_t i = 0; i + 1 < adj.size(); ++i) {
            slabs.push_back(adj[i]);
            slabs.push_back(adj[i + 1]);
        }
        if (slabs.empty()) {
            slabs.push_back(ymin);
            slabs.push_back(ymax);
        }
        slabs[0] = (slabs[0] + ymin) / 2.0;          // (:735-741)
        slabs[slabs.size() - 1] = ymax;
        const int NF = (int)slabs.size() / 2;
        *n_floors = NF;
        if (capacity == 0) return HMSG_OK;
        HMSG_REQUIRE(capacity >= NF, HMSG_ERR_INVALID, "hmsg_segment_floors: capacity too small (n_floors needed)");
        // ---- crops of the FULL map (:769-787)
        DevBuf<double> dsl;
        dsl.alloc(slabs.size());
        st.alloc((size_t)NF * 8);
        std::vector<unsigned long long> st0((size_t)NF * 8, 0ull);
        for (int f = 0; f < NF; ++f)
            for (int a = 0; a < 3; ++a) st0[(size_t)f * 8 + 1 + (size_t)a] = ~0ull;
        HIP_TRY(hipMemcpyAsync(dsl.p, slabs.data(), slabs.size() * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(st.p, st0.data(), st0.size() * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_fl_crop, dim3(cdiv((size_t)V, 256)), dim3(256), 0, s, (const double*)h->pts.p, V, NF, (const double*)dsl.p, st.p);
        HMSG_CHECK_LAUNCH();
        HIP_TRY(hipMemcpyAsync(st0.data(), st.p, st0.size() * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (int f = 0; f < NF; ++f) {
            hmsg_floor& o = out[f];
            o.y_lo = slabs[(size_t)f * 2];
            o.y_hi = slabs[(size_t)f * 2 + 1];
            o.n_points = (int64_t)st0[(size_t)f * 8];
            for (int a = 0; a < 3; ++a) {
                o.bbox_min[a] = o.n_points ? fl_unkey(st0[(size_t)f * 8 + 1 + (size_t)a]) : 0.0;
                o.bbox_max[a] = o.n_points ? fl_unkey(st0[(size_t)f * 8 + 4 + (size_t)a]) : 0.0;
            }
            o.zero_level = o.n_points ? o.bbox_min[1] : o.y_lo;
            o.height = o.y_hi - o.zero_level;
        }
        return HMSG_OK;
    } catch (const hmsg_error& e) {
        h->err = e.msg;
        return e.code;
    }
}
