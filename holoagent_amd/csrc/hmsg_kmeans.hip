// KMeans of compute_room_embeddings (utils/graph_utils.py:329-333) behind the boundary:
//     KMeans(n_clusters = num_views, max_iter = 100, n_init = 5, random_state = 0).fit(room_clip_embeddings)
// The reference calls scikit-learn (1.7.2 in its environment); a C / C++ host has no scikit-learn, so the fit is restated here,
// statement by statement, from sklearn/cluster/_kmeans.py (KMeans.fit, _kmeans_plusplus, _kmeans_single_lloyd, _tolerance),
// sklearn/cluster/_k_means_lloyd.pyx (lloyd_iter_chunked_dense, _update_chunk_dense), _k_means_common.pyx
// (_euclidean_dense_dense, _inertia_dense, _relocate_empty_clusters_dense, _average_centers, _center_shift,
// _is_same_clustering) and sklearn/metrics/pairwise.py (_euclidean_distances for float32 input), with numpy's
// RandomState(0) -- MT19937 seeded by init_genrand, random_sample = (a >> 5, b >> 6) / 2^53, choice(n, p) = searchsorted of
// the normalised cumulative sum (side right), uniform = random_sample -- and numpy's reductions (sequential along axis 0,
// pairwise along the contiguous axis) restated beside it.  Host code: ~10^2 rows of 512..1024 floats per room.
//
// What cannot be restated bit for bit are three BLAS / einsum calls whose summation ORDER is the library's business: the
// sgemm of the E-step (-2 X C^T), einsum("ij,ij->i") for the squared norms of the centres, and the sgemv / dgemm inside the
// k-means++ potentials.  Here those sums are accumulated in float64 and rounded once, i.e. they are within half a float32 ulp
// of the exact value where the libraries are within a few; a point whose two nearest centres tie to ~1e-7 relative can
// therefore land on the other side.  tests/test_kmeans_cabi.py holds this file to scikit-learn itself (labels equal, centres
// to 1e-6) on the room fixtures and on random data; scikit-learn stays the oracle.
#include "../../include/hmsg.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

// ---- numpy.random.RandomState(seed) for 0 <= seed < 2^32 (legacy seeding: init_genrand)
struct MT19937 {
    uint32_t key[624];
    int pos;
    explicit MT19937(uint32_t seed) {
        for (int i = 0; i < 624; ++i) {
            key[i] = seed;
            seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
        }
        pos = 624;
    }
    void gen() {
        const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, A = 0x9908b0dfu;
        int i;
        uint32_t y;
        for (i = 0; i < 624 - 397; ++i) {
            y = (key[i] & UPPER) | (key[i + 1] & LOWER);
            key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        for (; i < 623; ++i) {
            y = (key[i] & UPPER) | (key[i + 1] & LOWER);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        y = (key[623] & UPPER) | (key[0] & LOWER);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        pos = 0;
    }
    uint32_t next32() {
        if (pos == 624) gen();
        uint32_t y = key[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    double next_double() {                       // random_sample()
        const int32_t a = (int32_t)(next32() >> 5), b = (int32_t)(next32() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

// np.add.reduce over a contiguous float32 axis: numpy's pairwise summation (blocks of 128, eight accumulators)
float pairwise_sum_f32(const float* a, size_t n) {
    if (n < 8) {
        float res = 0.f;
        for (size_t i = 0; i < n; ++i) res += a[i];
        return res;
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
    return pairwise_sum_f32(a, n2) + pairwise_sum_f32(a + n2, n - n2);
}

// _euclidean_dense_dense (_k_means_common.pyx:16-43): float32, four terms per step
float euclid_dd(const float* a, const float* b, int n_features, bool squared) {
    const int n = n_features / 4, rem = n_features % 4;
    float result = 0.f;
    for (int i = 0; i < n; ++i) {
        result += ((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]) +
                   (a[3] - b[3]) * (a[3] - b[3]));
        a += 4;
        b += 4;
    }
    for (int i = 0; i < rem; ++i) result += (a[i] - b[i]) * (a[i] - b[i]);
    return squared ? result : std::sqrt(result);
}

// float64 dot product of float32 rows, eight running sums (the compiler keeps them in vector registers; the ORDER of a float64
// sum of ~10^3 float32 products moves its value by ~1e-16 relative, nine digits below the float32 rounding that follows)
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
__attribute__((target("avx2,fma")))
static double dot_f64_avx2(const float* a, const float* b, int n) {
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    for (; i + 8 <= n; i += 8)
        for (int k = 0; k < 8; ++k) s[k] += (double)a[i + k] * (double)b[i + k];
    double r = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    for (; i < n; ++i) r += (double)a[i] * (double)b[i];
    return r;
}
#endif
static double dot_f64(const float* a, const float* b, int n) {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
    static const bool have = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    if (have) return dot_f64_avx2(a, b, n);
#endif
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    for (; i + 8 <= n; i += 8)
        for (int k = 0; k < 8; ++k) s[k] += (double)a[i + k] * (double)b[i + k];
    double r = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    for (; i < n; ++i) r += (double)a[i] * (double)b[i];
    return r;
}

// _euclidean_distances(Y rows, X, squared=True) for float32 input (pairwise.py): chunks upcast to float64,
// d = ((-2 y.x) + |y|^2) + |x|^2 in float64, cast to float32, clipped at 0
void euclid_sq_rows(const float* X, int n, int D, const std::vector<double>& xx, const float* y, float* out) {
    const double yy = dot_f64(y, y, D);
    for (int i = 0; i < n; ++i) {
        const float* x = X + (size_t)i * D;
        const double dot = dot_f64(y, x, D);
        double d = -2.0 * dot;
        d += yy;
        d += xx[(size_t)i];
        const float f = (float)d;
        out[i] = f > 0.f ? f : 0.f;
    }
}

struct Fit {
    std::vector<int> labels;
    std::vector<float> centers;
    float inertia = 0.f;
    int n_iter = 0;
};

// _kmeans_plusplus (_kmeans.py) with sample_weight = ones
void kmeans_plusplus(const float* X, int n, int D, int k, const std::vector<double>& xx, MT19937& rs, std::vector<float>& centers) {
    const int n_local_trials = 2 + (int)std::log((double)k);
    // random_state.choice(n, p = w / w.sum()): p is float32(1 / n), the cdf its float64 running sum normalised by its last entry
    const double p = (double)(1.0f / (float)n);
    std::vector<double> cdf((size_t)n);
    double run = 0.0;
    for (int i = 0; i < n; ++i) {
        run += p;
        cdf[(size_t)i] = run;
    }
    for (int i = 0; i < n; ++i) cdf[(size_t)i] /= run;
    const double u0 = rs.next_double();
    int center_id = (int)(std::upper_bound(cdf.begin(), cdf.end(), u0) - cdf.begin());      // searchsorted(side="right")
    center_id = std::min(center_id, n - 1);
    centers.assign((size_t)k * D, 0.f);
    memcpy(centers.data(), X + (size_t)center_id * D, (size_t)D * 4);
    std::vector<float> closest((size_t)n);
    euclid_sq_rows(X, n, D, xx, centers.data(), closest.data());
    auto pot_of = [&](const float* d) {             // d @ ones (float32 gemv; accumulated in float64 here, see the header)
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += (double)d[i];
        return (float)s;
    };
    float current_pot = pot_of(closest.data());
    std::vector<double> cum((size_t)n);
    std::vector<float> cand_d((size_t)n_local_trials * n);
    std::vector<int> cand((size_t)n_local_trials);
    for (int c = 1; c < k; ++c) {
        double rv[16];
        for (int t = 0; t < n_local_trials; ++t) rv[t] = rs.next_double() * (double)current_pot;
        double acc = 0.0;                             // stable_cumsum(w * closest): float32 products, float64 running sum
        for (int i = 0; i < n; ++i) {
            acc += (double)(1.0f * closest[(size_t)i]);
            cum[(size_t)i] = acc;
        }
        for (int t = 0; t < n_local_trials; ++t) {
            int id = (int)(std::lower_bound(cum.begin(), cum.end(), rv[t]) - cum.begin());   // searchsorted (side left)
            cand[(size_t)t] = std::min(id, n - 1);
        }
        int best = 0;
        float best_pot = 0.f;
        for (int t = 0; t < n_local_trials; ++t) {
            float* d = cand_d.data() + (size_t)t * n;
            euclid_sq_rows(X, n, D, xx, X + (size_t)cand[(size_t)t] * D, d);
            for (int i = 0; i < n; ++i) d[i] = std::min(closest[(size_t)i], d[i]);
            const float pot = pot_of(d);
            if (t == 0 || pot < best_pot) {          // np.argmin: first minimum
                best = t;
                best_pot = pot;
            }
        }
        current_pot = best_pot;
        memcpy(closest.data(), cand_d.data() + (size_t)best * n, (size_t)n * 4);
        memcpy(centers.data() + (size_t)c * D, X + (size_t)cand[(size_t)best] * D, (size_t)D * 4);
    }
}

// lloyd_iter_chunked_dense (one thread).  centers_old -> labels [+ centers_new, center_shift]
void lloyd_iter(const float* X, int n, int D, int k, const float* centers_old, float* centers_new, float* weight, int* labels,
                float* center_shift, bool update_centers) {
    std::vector<float> cn2((size_t)k);
    for (int j = 0; j < k; ++j) {                   // row_norms(centers_old, squared=True): einsum, float64 here (header)
        cn2[(size_t)j] = (float)dot_f64(centers_old + (size_t)j * D, centers_old + (size_t)j * D, D);
    }
    if (update_centers) {
        std::fill(centers_new, centers_new + (size_t)k * D, 0.f);
        std::fill(weight, weight + k, 0.f);
    }
    // chunks of CHUNK_SIZE = 256 samples, one after the other; a chunk's partial sums are added to the totals when it is done
    const int CH = 256;
    std::vector<float> cc, wc;
    if (update_centers) {
        cc.assign((size_t)k * D, 0.f);              // (calloc'ed once per thread: the ONE thread keeps adding into it -- as sklearn does)
        wc.assign((size_t)k, 0.f);
    }
    for (int start = 0; start < n; start += CH) {
        const int end = std::min(n, start + CH);
        for (int i = start; i < end; ++i) {
            const float* x = X + (size_t)i * D;
            int label = 0;
            float min_d = 0.f;
            for (int j = 0; j < k; ++j) {
                // pairwise_distances = |c|^2 - 2 x.c (sgemm with beta = 1; the product accumulated in float64 here, see the header)
                const double dot = dot_f64(x, centers_old + (size_t)j * D, D);
                const float d = (float)((double)cn2[(size_t)j] - 2.0 * dot);
                if (j == 0 || d < min_d) {
                    min_d = d;
                    label = j;
                }
            }
            labels[i] = label;
            if (update_centers) {
                wc[(size_t)label] += 1.0f;
                float* dst = cc.data() + (size_t)label * D;
                for (int q = 0; q < D; ++q) dst[q] += x[q] * 1.0f;
            }
        }
    }
    if (!update_centers) return;
    for (int j = 0; j < k; ++j) {
        weight[j] += wc[(size_t)j];
        for (int q = 0; q < D; ++q) centers_new[(size_t)j * D + q] += cc[(size_t)j * D + q];
    }
    // _relocate_empty_clusters_dense
    std::vector<int> empty;
    for (int j = 0; j < k; ++j)
        if (weight[j] == 0.f) empty.push_back(j);
    if (!empty.empty()) {
        std::vector<float> dist((size_t)n), tmp((size_t)D);
        float dmax = 0.f;
        for (int i = 0; i < n; ++i) {
            const float* x = X + (size_t)i * D;
            const float* c = centers_old + (size_t)labels[i] * D;
            for (int q = 0; q < D; ++q) {
                const float t = x[q] - c[q];
                tmp[(size_t)q] = t * t;
            }
            dist[(size_t)i] = pairwise_sum_f32(tmp.data(), (size_t)D);       // (...**2).sum(axis=1)
            dmax = std::max(dmax, dist[(size_t)i]);
        }
        if (dmax != 0.f) {
            // np.argpartition(distances, -n_empty)[:-n_empty-1:-1]: the n_empty farthest samples.  With one empty cluster that is
            // the farthest sample; with several, numpy's introselect leaves them in an order of its own -- here: farthest first.
            std::vector<int> order((size_t)n);
            for (int i = 0; i < n; ++i) order[(size_t)i] = i;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return dist[(size_t)a] > dist[(size_t)b]; });
            for (size_t e = 0; e < empty.size() && e < (size_t)n; ++e) {
                const int far = order[e], nw = empty[e], old = labels[far];
                const float* x = X + (size_t)far * D;
                for (int q = 0; q < D; ++q) {
                    centers_new[(size_t)old * D + q] -= x[q] * 1.0f;
                    centers_new[(size_t)nw * D + q] = x[q] * 1.0f;
                }
                weight[nw] = 1.0f;
                weight[old] -= 1.0f;
            }
        }
    }
    // _average_centers
    int argmax_w = 0;
    for (int j = 1; j < k; ++j)
        if (weight[j] > weight[argmax_w]) argmax_w = j;
    for (int j = 0; j < k; ++j) {
        if (weight[j] > 0.f) {
            const float alpha = 1.0f / weight[j];
            for (int q = 0; q < D; ++q) centers_new[(size_t)j * D + q] *= alpha;
        } else {
            for (int q = 0; q < D; ++q) centers_new[(size_t)j * D + q] = centers_new[(size_t)argmax_w * D + q];
        }
    }
    // _center_shift
    for (int j = 0; j < k; ++j) center_shift[j] = euclid_dd(centers_new + (size_t)j * D, centers_old + (size_t)j * D, D, false);
}

bool same_clustering(const std::vector<int>& a, const std::vector<int>& b, int k) {
    std::vector<int> map((size_t)k, -1);
    for (size_t i = 0; i < a.size(); ++i) {
        if (map[(size_t)a[i]] == -1) map[(size_t)a[i]] = b[i];
        else if (map[(size_t)a[i]] != b[i]) return false;
    }
    return true;
}

}  // namespace

/* include/hmsg.h: hmsg_kmeans */
extern "C" int hmsg_kmeans(const float* X_in, int64_t n64, int32_t D, int32_t k, int32_t n_init, int32_t max_iter, uint32_t seed,
                           int32_t* out_labels, float* out_centers, float* out_inertia, int32_t* out_n_iter) {
    if (!X_in || n64 <= 0 || n64 > (1 << 24) || D <= 0 || k <= 0 || k > n64 || n_init <= 0 || max_iter <= 0 || k > 65536 || !out_labels ||
        !out_centers)
        return HMSG_ERR_INVALID;
    try {
        const int n = (int)n64;
        // KMeans.fit: X = copy, X_mean = X.mean(axis=0) (float32, rows added one after the other), X -= X_mean
        std::vector<float> X(X_in, X_in + (size_t)n * D), mean((size_t)D, 0.f);
        for (int i = 0; i < n; ++i)
            for (int q = 0; q < D; ++q) mean[(size_t)q] += X[(size_t)i * D + q];
        for (int q = 0; q < D; ++q) mean[(size_t)q] = mean[(size_t)q] / (float)n;
        // _tolerance: np.mean(np.var(X, axis=0)) * 1e-4 on the data as given (fit computes it before centring: _check_params_vs_input)
        float tol;
        {
            std::vector<float> var((size_t)D, 0.f);
            for (int i = 0; i < n; ++i)
                for (int q = 0; q < D; ++q) {
                    const float t = X[(size_t)i * D + q] - mean[(size_t)q];
                    var[(size_t)q] += t * t;
                }
            for (int q = 0; q < D; ++q) var[(size_t)q] = var[(size_t)q] / (float)n;
            tol = (pairwise_sum_f32(var.data(), (size_t)D) / (float)D) * 1e-4f;
        }
        for (int i = 0; i < n; ++i)
            for (int q = 0; q < D; ++q) X[(size_t)i * D + q] -= mean[(size_t)q];
        std::vector<double> xx((size_t)n);          // |x|^2 of the upcast rows (_euclidean_distances_upcast recomputes them in float64)
        for (int i = 0; i < n; ++i) {
            xx[(size_t)i] = dot_f64(X.data() + (size_t)i * D, X.data() + (size_t)i * D, D);
        }
        MT19937 rs(seed);
        Fit best;
        bool have = false;
        for (int run = 0; run < n_init; ++run) {
            Fit f;
            std::vector<float> centers, centers_new((size_t)k * D, 0.f), weight((size_t)k, 0.f), shift((size_t)k, 0.f);
            kmeans_plusplus(X.data(), n, D, k, xx, rs, centers);
            f.labels.assign((size_t)n, -1);
            std::vector<int> labels_old((size_t)n, -1);
            bool strict = false;
            int it = 0;
            for (it = 0; it < max_iter; ++it) {
                lloyd_iter(X.data(), n, D, k, centers.data(), centers_new.data(), weight.data(), f.labels.data(), shift.data(), true);
                centers.swap(centers_new);
                if (f.labels == labels_old) {
                    strict = true;
                    break;
                }
                std::vector<float> sq((size_t)k);
                for (int j = 0; j < k; ++j) sq[(size_t)j] = shift[(size_t)j] * shift[(size_t)j];
                if (pairwise_sum_f32(sq.data(), (size_t)k) <= tol) break;
                labels_old = f.labels;
            }
            f.n_iter = std::min(it + 1, max_iter);
            if (!strict) lloyd_iter(X.data(), n, D, k, centers.data(), nullptr, nullptr, f.labels.data(), nullptr, false);
            float inertia = 0.f;                     // _inertia_dense
            for (int i = 0; i < n; ++i) inertia += euclid_dd(X.data() + (size_t)i * D, centers.data() + (size_t)f.labels[(size_t)i] * D, D, true) * 1.0f;
            f.inertia = inertia;
            f.centers = centers;
            if (!have || (f.inertia < best.inertia && !same_clustering(f.labels, best.labels, k))) {
                best = f;
                have = true;
            }
        }
        for (int j = 0; j < k; ++j)
            for (int q = 0; q < D; ++q) out_centers[(size_t)j * D + q] = best.centers[(size_t)j * D + q] + mean[(size_t)q];
        for (int i = 0; i < n; ++i) out_labels[i] = best.labels[(size_t)i];
        if (out_inertia) *out_inertia = best.inertia;
        if (out_n_iter) *out_n_iter = best.n_iter;
        return HMSG_OK;
    } catch (const std::exception&) {
        return HMSG_ERR_NOMEM;
    } catch (...) {
        return HMSG_ERR_INVALID;
    }
}
