// N1 behind the C ABI: the rooms of one storey as a label image -- Graph.segment_hmsg_room up to its room_vertices
// (fsr_vln/memory/hmsg/graph/graph.py:942-1071) and distance_transform (fsr_vln/memory/hmsg/utils/graph_utils.py:391-487).
//
// The reference runs this on the host with numpy + OpenCV 4.8 (un-vendored): 2-D histograms of the storey's points at
// grid_resolution, normalise / Gaussian blur / threshold, a 10 px border, morphological closings, filled outer contours,
// an exact Euclidean distance transform, Otsu, seed contours above (0.5 m / resolution)^2 px, and a seeded watershed.
// Here every step is a kernel over the (few hundred)^2 grid, the points never leave the device, and what comes back is the
// marker image.  OpenCV is restated, not linked: oracle/rooms_oracle.py states which semantics were taken (float Gaussian
// kernel instead of OpenCV's fixed-point one, filled external contours = components with their holes filled, Pick's
// theorem for contourArea, a layer-synchronous watershed); this file implements exactly that restatement and the tests
// hold the two equal pixel for pixel.  Parity with OpenCV itself is statistical.
#include "hmsg_common.h"

#include <cmath>

namespace {

constexpr int RB = 256;

__device__ __forceinline__ unsigned long long d_key(double d) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | (1ull << 63));
}
inline double key_d(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & ~(1ull << 63)) : ~k;
    double d;
    memcpy(&d, &u, 8);
    return d;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    for (int o = 32; o; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o);
        v = w < v ? w : v;
    }
    return v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    for (int o = 32; o; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o);
        v = w > v ? w : v;
    }
    return v;
}

// rng[0..3] = min x, max x, min z, max z of the wall band (zero + 0.3 <= y < zero + height - 0.3), rng[4..7] the same of
// everything below zero + height - 0.2 (graph.py:946-952); keys are order-preserving encodings of the doubles.
__global__ void __launch_bounds__(RB) k_rm_range(const double* __restrict__ pts, long long V, double y_lo, double y_hi, double wall_lo,
                                                 double wall_hi, double full_hi, unsigned long long* __restrict__ rng) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool a = false, b = false;
    double x = 0, z = 0;
    if (i < V) {
        const double y = pts[i * 3 + 1];
        if (y >= y_lo && y <= y_hi) {
            x = pts[i * 3];
            z = pts[i * 3 + 2];
            b = y < full_hi;
            a = y < wall_hi && y >= wall_lo;
        }
    }
    const unsigned long long kx = d_key(x), kz = d_key(z), hi = ~0ull;
    unsigned long long v[8] = {a ? kx : hi, a ? kx : 0, a ? kz : hi, a ? kz : 0, b ? kx : hi, b ? kx : 0, b ? kz : hi, b ? kz : 0};
    for (int k = 0; k < 8; ++k) v[k] = (k & 1) ? wave_max_u64(v[k]) : wave_min_u64(v[k]);
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 8; ++k) {
            if (k & 1) atomicMax(&rng[k], v[k]);
            else atomicMin(&rng[k], v[k]);
        }
}

// np.histogram2d's bin of v: edges = linspace(lo, hi, n + 1) -- k * step + lo, the last one hi itself --, bin = (number of
// edges <= v) - 1, the last edge belonging to the last bin.
__device__ __forceinline__ int hist_bin(double v, double lo, double hi, int n) {
    const double step = (hi - lo) / (double)n;
    int k = (int)floor((v - lo) / step);
    k = k < 0 ? 0 : (k > n ? n : k);
    auto edge = [&](int j) { return j == n ? hi : (double)j * step + lo; };
    while (k > 0 && v < edge(k)) --k;
    while (k < n && v >= edge(k + 1)) ++k;
    return k >= n ? n - 1 : k;
}
__global__ void __launch_bounds__(RB) k_rm_hist(const double* __restrict__ pts, long long V, double y_lo, double y_hi, double wall_lo,
                                                double wall_hi, double full_hi, const double* __restrict__ r, int nbz, int nbx,
                                                int* __restrict__ hist_wall, int* __restrict__ hist_full) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const double y = pts[i * 3 + 1];
    if (!(y >= y_lo && y <= y_hi)) return;
    const double x = pts[i * 3], z = pts[i * 3 + 2];
    if (y < wall_hi && y >= wall_lo) atomicAdd(&hist_wall[hist_bin(z, r[2], r[3], nbz) * nbx + hist_bin(x, r[0], r[1], nbx)], 1);
    if (y < full_hi) atomicAdd(&hist_full[hist_bin(z, r[6], r[7], nbz) * nbx + hist_bin(x, r[4], r[5], nbx)], 1);
}

__global__ void __launch_bounds__(RB) k_rm_minmax_i32(const int* __restrict__ a, int n, int* __restrict__ mm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int lo = i < n ? a[i] : 0x7fffffff, hi = i < n ? a[i] : (int)0x80000000;
    for (int o = 32; o; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o));
        hi = max(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[0], lo);
        atomicMax(&mm[1], hi);
    }
}
// cv2.normalize(NORM_MINMAX, 0..255) in double (scale = 255 * (1 / (max - min)), shift = -min * scale), then astype(uint8)
__global__ void __launch_bounds__(RB) k_rm_norm_i32(const int* __restrict__ a, int n, const int* __restrict__ mm, unsigned char* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double lo = (double)mm[0], hi = (double)mm[1];
    const double scale = hi > lo ? 255.0 * (1.0 / (hi - lo)) : 0.0, shift = 0.0 - lo * scale;
    const double v = (double)a[i] * scale + shift;
    out[i] = (unsigned char)(int)v;
}
__global__ void __launch_bounds__(RB) k_rm_max_u8(const unsigned char* __restrict__ a, int n, int* __restrict__ mx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int hi = i < n ? (int)a[i] : 0;
    for (int o = 32; o; o >>= 1) hi = max(hi, __shfl_xor(hi, o));
    if ((threadIdx.x & 63) == 0) atomicMax(mx, hi);
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}
// one axis of GaussianBlur in double, BORDER_REFLECT_101; the symmetric taps are paired as scipy's correlate1d pairs them
__global__ void __launch_bounds__(RB) k_rm_blur_axis(const double* __restrict__ in, int rows, int cols, int axis, int ksize,
                                                     const double* __restrict__ w, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int r = i / cols, c = i % cols, h = ksize / 2;
    auto at = [&](int d) { return axis ? in[r * cols + reflect101(c + d, cols)] : in[reflect101(r + d, rows) * cols + c]; };
    double acc = at(0) * w[h];
    for (int j = -h; j < 0; ++j) acc += (at(j) + at(-j)) * w[j + h];
    out[i] = acc;
}
__global__ void __launch_bounds__(RB) k_rm_u8_to_f64(const unsigned char* __restrict__ a, int n, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)a[i];
}
__global__ void __launch_bounds__(RB) k_rm_round_u8(const double* __restrict__ a, int n, unsigned char* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = floor(a[i] + 0.5);
    v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
    out[i] = (unsigned char)(int)v;
}
// binary threshold written into the image with its 10 px zero border: out[(r + pad, c + pad)] = in > t ? 255 : 0
__global__ void __launch_bounds__(RB) k_rm_thresh_pad(const unsigned char* __restrict__ in, int rows, int cols, const int* __restrict__ mx,
                                                      double frac, int pad, unsigned char* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int R = rows + 2 * pad, C = cols + 2 * pad;
    if (i >= R * C) return;
    const int r = i / C - pad, c = i % C - pad;
    unsigned char v = 0;
    if (r >= 0 && r < rows && c >= 0 && c < cols) v = ((double)in[r * cols + c] > frac * (double)mx[0]) ? 255 : 0;
    out[i] = v;
}
// one dilation (grow = 1) or erosion (grow = 0) by a k x k rectangle or cross; outside the image never wins (OpenCV's
// default morphology border)
__global__ void __launch_bounds__(RB) k_rm_morph(const unsigned char* __restrict__ in, int rows, int cols, int k, int cross, int grow,
                                                 unsigned char* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int r = i / cols, c = i % cols, h = k / 2;
    bool any = false, all = true;
    for (int dr = -h; dr <= h; ++dr)
        for (int dc = -h; dc <= h; ++dc) {
            if (cross && dr && dc) continue;
            const int rr = r + dr, cc = c + dc;
            if (rr < 0 || rr >= rows || cc < 0 || cc >= cols) continue;
            const bool f = in[rr * cols + cc] != 0;
            any |= f;
            all &= f;
        }
    out[i] = (grow ? any : all) ? 255 : 0;
}

// ---- connected components by label equivalence: lab[p] = smallest pixel index of p's component (pixels with
// (img != 0) == fg; conn 4 or 8); `border_root`: pixels of the class on the image border start as -1, so that everything
// connected to the border ends as -1.
__global__ void __launch_bounds__(RB) k_cc_init(const unsigned char* __restrict__ img, int rows, int cols, int fg, int border_root,
                                                int* __restrict__ lab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int r = i / cols, c = i % cols;
    const bool in = ((img[i] != 0) ? 1 : 0) == fg;
    const bool edge = r == 0 || c == 0 || r == rows - 1 || c == cols - 1;
    lab[i] = !in ? 0x7fffffff : ((border_root && edge) ? -1 : i);
}
__global__ void __launch_bounds__(RB) k_cc_scan(int rows, int cols, int conn, int* __restrict__ lab, int* __restrict__ changed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int l = lab[i];
    if (l == 0x7fffffff || l == -1) return;
    const int r = i / cols, c = i % cols;
    int m = l;
    for (int dr = -1; dr <= 1; ++dr)
        for (int dc = -1; dc <= 1; ++dc) {
            if ((!dr && !dc) || (conn == 4 && dr && dc)) continue;
            const int rr = r + dr, cc = c + dc;
            if (rr < 0 || rr >= rows || cc < 0 || cc >= cols) continue;
            const int q = lab[rr * cols + cc];
            if (q != 0x7fffffff && q < m) m = q;
        }
    if (m < l) {
        atomicMin(&lab[l], m);          // hook the old representative as well
        atomicMin(&lab[i], m);
        *changed = 1;
    }
}
__global__ void __launch_bounds__(RB) k_cc_flatten(int n, int* __restrict__ lab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int l = lab[i];
    if (l == 0x7fffffff || l == -1) return;
    while (true) {
        const int p = lab[l];
        if (p == l || p == -1) {
            if (p == -1) l = -1;
            break;
        }
        l = p;
    }
    lab[i] = l;
}
// filled outer contours: everything that is not background connected to the border
__global__ void __launch_bounds__(RB) k_rm_fill(const int* __restrict__ bg_lab, int n, unsigned char* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bg_lab[i] == -1 ? 0 : 255;
}
// full_map = walls | ~outside
__global__ void __launch_bounds__(RB) k_rm_full(const unsigned char* __restrict__ walls, const unsigned char* __restrict__ outside, int n,
                                                unsigned char* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (unsigned char)(walls[i] | (unsigned char)~outside[i]);
}

// ---- exact Euclidean distance transform of the free space (full_map == 0) to the nearest non-free pixel
__global__ void __launch_bounds__(RB) k_edt_cols(const unsigned char* __restrict__ full, int rows, int cols, int* __restrict__ g) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    const int INF = 1 << 20;
    int d = INF;
    for (int r = 0; r < rows; ++r) {
        d = full[r * cols + c] ? 0 : (d >= INF ? INF : d + 1);
        g[r * cols + c] = d;
    }
    d = INF;
    for (int r = rows - 1; r >= 0; --r) {
        d = full[r * cols + c] ? 0 : (d >= INF ? INF : d + 1);
        if (d < g[r * cols + c]) g[r * cols + c] = d;
    }
}
__global__ void __launch_bounds__(RB) k_edt_rows(const int* __restrict__ g, int rows, int cols, float* __restrict__ dist, unsigned* __restrict__ mm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float out = 0.f;
    if (i < rows * cols) {
        const int r = i / cols, c = i % cols;
        long long best = 1ll << 60;
        for (int cc = 0; cc < cols; ++cc) {
            const long long gv = g[r * cols + cc];
            if (gv >= (1 << 20)) continue;
            const long long dx = cc - c, d2 = dx * dx + gv * gv;
            best = d2 < best ? d2 : best;
        }
        out = (float)sqrt((double)best);
        dist[i] = out;
    }
    unsigned lo = i < rows * cols ? __float_as_uint(out) : 0xffffffffu, hi = i < rows * cols ? __float_as_uint(out) : 0u;
    for (int o = 32; o; o >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, o));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[0], lo);
        atomicMax(&mm[1], hi);
    }
}
__global__ void __launch_bounds__(RB) k_rm_norm_f32(const float* __restrict__ d, int n, const unsigned* __restrict__ mm, unsigned char* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float lo = __uint_as_float(mm[0]), hi = __uint_as_float(mm[1]);
    if (!(hi > lo)) {
        out[i] = 0;
        return;
    }
    const double sc = 255.0 * (1.0 / ((double)hi - (double)lo));          // cv2.normalize: double scale / shift, float arithmetic
    const float scale = (float)sc, shift = (float)(0.0 - (double)lo * sc);
    out[i] = (unsigned char)(int)(d[i] * scale + shift);
}
__global__ void __launch_bounds__(RB) k_rm_hist256(const unsigned char* __restrict__ a, int n, int* __restrict__ h) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&h[a[i]], 1);
}
// Otsu's threshold (first maximum of the between-class variance), one thread
__global__ void k_rm_otsu(const int* __restrict__ h, int* __restrict__ thr) {
    if (threadIdx.x || blockIdx.x) return;
    double total = 0, wsum = 0;
    for (int t = 0; t < 256; ++t) {
        total += (double)h[t];
        wsum += (double)h[t] * (double)t;
    }
    const double mu_total = wsum / total;
    int best_t = 0;
    double best_v = -1.0, q1 = 0, mu1_sum = 0;
    for (int t = 0; t < 256; ++t) {
        q1 += (double)h[t];
        mu1_sum += (double)t * (double)h[t];
        if (q1 == 0 || q1 == total) continue;
        const double mu1 = mu1_sum / q1, mu2 = (mu_total * total - mu1_sum) / (total - q1);
        const double dm = mu1 - mu2, v = q1 * (total - q1) * (dm * dm);
        if (v > best_v) {
            best_v = v;
            best_t = t;
        }
    }
    thr[0] = best_t;
}
__global__ void __launch_bounds__(RB) k_rm_gt(const unsigned char* __restrict__ a, int n, const int* __restrict__ thr, unsigned char* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int)a[i] > thr[0] ? 255 : 0;
}
// per seed component (lab = its smallest pixel index): pixel count and boundary pixel count (a pixel with a neighbour
// outside the filled component or outside the image)
__global__ void __launch_bounds__(RB) k_seed_stats(const int* __restrict__ lab, int rows, int cols, int* __restrict__ area, int* __restrict__ bnd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int l = lab[i];
    if (l == 0x7fffffff) return;
    const int r = i / cols, c = i % cols;
    bool inner = true;
    for (int dr = -1; dr <= 1; ++dr)
        for (int dc = -1; dc <= 1; ++dc) {
            const int rr = r + dr, cc = c + dc;
            if (rr < 0 || rr >= rows || cc < 0 || cc >= cols || lab[rr * cols + cc] == 0x7fffffff) inner = false;
        }
    atomicAdd(&area[l], 1);
    if (!inner) atomicAdd(&bnd[l], 1);
}
// the kept components' representatives (contourArea > min_area), unordered
__global__ void __launch_bounds__(RB) k_seed_pick(const int* __restrict__ lab, int n, const int* __restrict__ area, const int* __restrict__ bnd,
                                                  double min_area, int* __restrict__ roots, int cap, int* __restrict__ n_roots) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || lab[i] != i) return;
    const double a = (double)area[i] - (double)bnd[i] / 2.0 - 1.0;
    if (a > min_area) {
        const int k = atomicAdd(n_roots, 1);
        if (k < cap) roots[k] = i;
    }
}
// seed number of a kept component = 1 + the number of kept components found later in the raster scan (findContours hands
// the outer contours back in reverse discovery order); the background marker R + 1 is cv2.circle((3, 3), 1)
__global__ void __launch_bounds__(RB) k_seed_markers(const int* __restrict__ lab, int rows, int cols, const int* __restrict__ roots,
                                                     const int* __restrict__ n_roots, int* __restrict__ markers) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int R = n_roots[0], l = lab[i];
    int m = 0;
    if (l != 0x7fffffff) {
        bool kept = false;
        int later = 0;
        for (int k = 0; k < R; ++k) {
            kept |= roots[k] == l;
            later += roots[k] > l;
        }
        if (kept) m = later + 1;
    }
    const int r = i / cols, c = i % cols;
    if ((r - 3) * (r - 3) + (c - 3) * (c - 3) <= 1) m = R + 1;
    if (r == 0 || c == 0 || r == rows - 1 || c == cols - 1) m = -1;
    markers[i] = m;
}
// one synchronous round of the watershed restatement
__global__ void __launch_bounds__(RB) k_ws_round(const unsigned char* __restrict__ colour, const int* __restrict__ in, int rows, int cols,
                                                 int same_colour, int* __restrict__ out, int* __restrict__ changed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    int m = in[i];
    if (m == 0) {
        const int r = i / cols, c = i % cols;
        int lo = 0x7fffffff, hi = 0;
        const int dr[4] = {-1, 1, 0, 0}, dc[4] = {0, 0, -1, 1};
        for (int k = 0; k < 4; ++k) {
            const int rr = r + dr[k], cc = c + dc[k];
            if (rr < 0 || rr >= rows || cc < 0 || cc >= cols) continue;
            const int q = in[rr * cols + cc];
            if (q <= 0 || (same_colour && colour[rr * cols + cc] != colour[i])) continue;
            lo = min(lo, q);
            hi = max(hi, q);
        }
        if (hi > 0) {
            m = lo == hi ? lo : -1;
            *changed = 1;
        }
    }
    out[i] = m;
}

struct Img {
    int rows = 0, cols = 0;
    int n() const { return rows * cols; }
    unsigned grid() const { return cdiv((size_t)n(), RB); }
};

void gauss_weights(int ksize, double sigma, std::vector<double>& w) {
    w.resize((size_t)ksize);
    for (int i = 0; i < ksize; ++i) {
        const double x = (double)i - (double)(ksize - 1) / 2.0;
        w[(size_t)i] = std::exp(-(x * x) / (2.0 * sigma * sigma));
    }
    const double sum = np_sum_f64(w.data(), w.size());       // (the oracle normalises with np.sum)
    for (auto& v : w) v /= sum;
}

// Python's float floor division (float_floor_div): int(a // b) for a > 0, b > 0
long long py_floordiv(double a, double b) {
    double mod = std::fmod(a, b);
    double div = (a - mod) / b;
    if (mod != 0.0 && ((b < 0) != (mod < 0))) div -= 1.0;
    double fl = 0.0;
    if (div != 0.0) {
        fl = std::floor(div);
        if (div - fl > 0.5) fl += 1.0;
    }
    return (long long)fl;
}

struct Rooms {
    hipStream_t s;
    DevBuf<double> fa, fb, w;
    DevBuf<unsigned char> tmp8;
    DevBuf<int> flag;

    void blur(unsigned char* img, Img d, int kx, int ky, double sigma) {
        fa.ensure((size_t)d.n());
        fb.ensure((size_t)d.n());
        hipLaunchKernelGGL(k_rm_u8_to_f64, dim3(d.grid()), dim3(RB), 0, s, (const unsigned char*)img, d.n(), fa.p);
        std::vector<double> hw;
        for (int axis = 1; axis >= 0; --axis) {                 // x first, then y (the oracle's order)
            const int k = axis ? kx : ky;
            if (k <= 1) continue;
            gauss_weights(k, sigma, hw);
            w.ensure(64);
            HIP_TRY(hipMemcpyAsync(w.p, hw.data(), hw.size() * 8, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_rm_blur_axis, dim3(d.grid()), dim3(RB), 0, s, (const double*)fa.p, d.rows, d.cols, axis, k, (const double*)w.p, fb.p);
            HIP_TRY(hipStreamSynchronize(s));                   // (hw is reused)
            fa.swap(fb);
        }
        hipLaunchKernelGGL(k_rm_round_u8, dim3(d.grid()), dim3(RB), 0, s, (const double*)fa.p, d.n(), img);
        HMSG_CHECK_LAUNCH();
    }
    void close(unsigned char* img, Img d, int k, int cross, int iterations) {
        tmp8.ensure((size_t)d.n());
        unsigned char *a = img, *b = tmp8.p;
        for (int grow = 1; grow >= 0; --grow)
            for (int it = 0; it < iterations; ++it) {
                hipLaunchKernelGGL(k_rm_morph, dim3(d.grid()), dim3(RB), 0, s, (const unsigned char*)a, d.rows, d.cols, k, cross, grow, b);
                std::swap(a, b);
            }
        if (a != img) HIP_TRY(hipMemcpyAsync(img, a, (size_t)d.n(), hipMemcpyDeviceToDevice, s));
        HMSG_CHECK_LAUNCH();
    }
    void components(const unsigned char* img, Img d, int fg, int conn, int border_root, int* lab) {
        flag.ensure(1);
        hipLaunchKernelGGL(k_cc_init, dim3(d.grid()), dim3(RB), 0, s, img, d.rows, d.cols, fg, border_root, lab);
        for (int round = 0; round < 4096; ++round) {
            HIP_TRY(hipMemsetAsync(flag.p, 0, 4, s));
            for (int k = 0; k < 4; ++k) {
                hipLaunchKernelGGL(k_cc_scan, dim3(d.grid()), dim3(RB), 0, s, d.rows, d.cols, conn, lab, flag.p);
                hipLaunchKernelGGL(k_cc_flatten, dim3(d.grid()), dim3(RB), 0, s, d.n(), lab);
            }
            int ch = 0;
            HIP_TRY(hipMemcpyAsync(&ch, flag.p, 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (!ch) break;
        }
        HMSG_CHECK_LAUNCH();
    }
};

}  // namespace

extern "C" int hmsg_segment_rooms(hmsg_t* h, double y_lo, double y_hi, double zero_level, double height, double resolution,
                                  int32_t* out_markers, int64_t capacity, int32_t* out_rows, int32_t* out_cols, int32_t* out_n_rooms,
                                  double* out_xz_min) {
    if (!h) return HMSG_ERR_INVALID;
    try {
        HIP_TRY(hipSetDevice(h->cfg.device_id));
        HMSG_REQUIRE(h->map_ready && out_rows && out_cols && out_n_rooms && out_xz_min && resolution > 0, HMSG_ERR_INVALID,
                     "hmsg_segment_rooms: bad argument (finalize the map first)");
        Rooms R;
        R.s = h->stream;
        hipStream_t s = h->stream;
        const long long V = h->V;
        *out_rows = *out_cols = *out_n_rooms = 0;
        HMSG_REQUIRE(V > 0, HMSG_ERR_INVALID, "hmsg_segment_rooms: empty map");
        const double wall_lo = zero_level + 0.3, wall_hi = zero_level + height - 0.3, full_hi = zero_level + height - 0.2;
        // ---- ranges (graph.py:946-957)
        DevBuf<unsigned long long> rng;
        rng.alloc(8);
        const unsigned long long init[8] = {~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0};
        HIP_TRY(hipMemcpyAsync(rng.p, init, 64, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_rm_range, dim3(cdiv((size_t)V, RB)), dim3(RB), 0, s, (const double*)h->pts.p, V, y_lo, y_hi, wall_lo, wall_hi, full_hi, rng.p);
        HMSG_CHECK_LAUNCH();
        unsigned long long hr[8];
        HIP_TRY(hipMemcpyAsync(hr, rng.p, 64, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        HMSG_REQUIRE(hr[0] != ~0ull && hr[4] != ~0ull, HMSG_ERR_INVALID, "hmsg_segment_rooms: no points between the storey's levels");
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = key_d(hr[k]);
        HMSG_REQUIRE(r[1] > r[0] && r[3] > r[2] && r[5] > r[4] && r[7] > r[6], HMSG_ERR_INVALID, "hmsg_segment_rooms: degenerate extent");
        const long long gs0 = (long long)(r[1] - r[0]) + 1, gs1 = (long long)(r[3] - r[2]) + 1;
        const int nbz = (int)py_floordiv((double)gs1, resolution) + 1, nbx = (int)py_floordiv((double)gs0, resolution) + 1;
        HMSG_REQUIRE(nbz > 0 && nbx > 0 && nbz <= 8192 && nbx <= 8192, HMSG_ERR_UNSUPPORTED, "hmsg_segment_rooms: grid larger than 8192 cells a side");
        Img g0{nbz, nbx}, g{nbz + 20, nbx + 20};
        out_xz_min[0] = r[0];
        out_xz_min[1] = r[2];
        *out_rows = g.rows;
        *out_cols = g.cols;
        if (!out_markers) return HMSG_OK;                       // (size query)
        HMSG_REQUIRE(capacity >= (int64_t)g.n(), HMSG_ERR_INVALID, "hmsg_segment_rooms: out_markers too small (rows * cols needed)");
        // ---- histograms, normalise, blur, threshold into the padded images (graph.py:957-1040)
        DevBuf<double> dr;
        DevBuf<int> hw, hf, mm;
        DevBuf<unsigned char> u8a, u8b, walls, outside, full;
        dr.alloc(8);
        hw.alloc((size_t)g0.n());
        hf.alloc((size_t)g0.n());
        mm.alloc(8);
        u8a.alloc((size_t)g.n());
        u8b.alloc((size_t)g.n());
        walls.alloc((size_t)g.n());
        outside.alloc((size_t)g.n());
        full.alloc((size_t)g.n());
        HIP_TRY(hipMemcpyAsync(dr.p, r, 64, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(hw.p, 0, (size_t)g0.n() * 4, s));
        HIP_TRY(hipMemsetAsync(hf.p, 0, (size_t)g0.n() * 4, s));
        hipLaunchKernelGGL(k_rm_hist, dim3(cdiv((size_t)V, RB)), dim3(RB), 0, s, (const double*)h->pts.p, V, y_lo, y_hi, wall_lo, wall_hi, full_hi,
                           (const double*)dr.p, nbz, nbx, hw.p, hf.p);
        HMSG_CHECK_LAUNCH();
        const int mm_init[8] = {0x7fffffff, (int)0x80000000, 0, 0, 0x7fffffff, (int)0x80000000, 0, 0};
        HIP_TRY(hipMemcpyAsync(mm.p, mm_init, 32, hipMemcpyHostToDevice, s));
        for (int which = 0; which < 2; ++which) {
            int* hist = which ? hf.p : hw.p;
            int* m = mm.p + which * 4;
            unsigned char* dst = which ? outside.p : walls.p;
            hipLaunchKernelGGL(k_rm_minmax_i32, dim3(g0.grid()), dim3(RB), 0, s, (const int*)hist, g0.n(), m);
            hipLaunchKernelGGL(k_rm_norm_i32, dim3(g0.grid()), dim3(RB), 0, s, (const int*)hist, g0.n(), (const int*)m, u8a.p);
            HMSG_CHECK_LAUNCH();
            if (which) R.blur(u8a.p, g0, 21, 21, 2.0);
            else R.blur(u8a.p, g0, 5, 5, 1.0);
            hipLaunchKernelGGL(k_rm_max_u8, dim3(g0.grid()), dim3(RB), 0, s, (const unsigned char*)u8a.p, g0.n(), m + 2);
            hipLaunchKernelGGL(k_rm_thresh_pad, dim3(g.grid()), dim3(RB), 0, s, (const unsigned char*)u8a.p, g0.rows, g0.cols, (const int*)(m + 2),
                               which ? 0.0 : 0.25, 10, dst);
            HMSG_CHECK_LAUNCH();
            if (which) R.close(dst, g, 5, 0, 3);
            else R.close(dst, g, 3, 1, 1);
        }
        // ---- filled outer contours of the outside mask, the full map (graph.py:1042-1062)
        DevBuf<int> lab, area, bnd, roots, nroots, mk_a, mk_b;
        lab.alloc((size_t)g.n());
        R.components(outside.p, g, 0, 4, 1, lab.p);
        hipLaunchKernelGGL(k_rm_fill, dim3(g.grid()), dim3(RB), 0, s, (const int*)lab.p, g.n(), outside.p);
        hipLaunchKernelGGL(k_rm_full, dim3(g.grid()), dim3(RB), 0, s, (const unsigned char*)walls.p, (const unsigned char*)outside.p, g.n(), full.p);
        HMSG_CHECK_LAUNCH();
        R.close(full.p, g, 3, 0, 2);
        // ---- distance_transform (graph_utils.py:391-487)
        DevBuf<int> gcol;
        DevBuf<float> dist;
        DevBuf<unsigned> fmm;
        gcol.alloc((size_t)g.n());
        dist.alloc((size_t)g.n());
        fmm.alloc(2);
        const unsigned fmm_init[2] = {0xffffffffu, 0u};
        HIP_TRY(hipMemcpyAsync(fmm.p, fmm_init, 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_edt_cols, dim3(cdiv((size_t)g.cols, RB)), dim3(RB), 0, s, (const unsigned char*)full.p, g.rows, g.cols, gcol.p);
        hipLaunchKernelGGL(k_edt_rows, dim3(g.grid()), dim3(RB), 0, s, (const int*)gcol.p, g.rows, g.cols, dist.p, fmm.p);
        hipLaunchKernelGGL(k_rm_norm_f32, dim3(g.grid()), dim3(RB), 0, s, (const float*)dist.p, g.n(), (const unsigned*)fmm.p, u8a.p);
        HMSG_CHECK_LAUNCH();
        R.blur(u8a.p, g, 11, 1, 10.0);
        DevBuf<int> h256, thr;
        h256.alloc(256);
        thr.alloc(1);
        HIP_TRY(hipMemsetAsync(h256.p, 0, 1024, s));
        hipLaunchKernelGGL(k_rm_hist256, dim3(g.grid()), dim3(RB), 0, s, (const unsigned char*)u8a.p, g.n(), h256.p);
        hipLaunchKernelGGL(k_rm_otsu, dim3(1), dim3(64), 0, s, (const int*)h256.p, thr.p);
        hipLaunchKernelGGL(k_rm_gt, dim3(g.grid()), dim3(RB), 0, s, (const unsigned char*)u8a.p, g.n(), (const int*)thr.p, u8b.p);
        HMSG_CHECK_LAUNCH();
        // seeds: outer contours (holes filled), 8-connected, area filter, numbered in reverse discovery order
        R.components(u8b.p, g, 0, 4, 1, lab.p);
        hipLaunchKernelGGL(k_rm_fill, dim3(g.grid()), dim3(RB), 0, s, (const int*)lab.p, g.n(), u8b.p);
        R.components(u8b.p, g, 1, 8, 0, lab.p);
        const int root_cap = 4096;
        area.alloc((size_t)g.n());
        bnd.alloc((size_t)g.n());
        roots.alloc((size_t)root_cap);
        nroots.alloc(1);
        mk_a.alloc((size_t)g.n());
        mk_b.alloc((size_t)g.n());
        HIP_TRY(hipMemsetAsync(area.p, 0, (size_t)g.n() * 4, s));
        HIP_TRY(hipMemsetAsync(bnd.p, 0, (size_t)g.n() * 4, s));
        HIP_TRY(hipMemsetAsync(nroots.p, 0, 4, s));
        const double q = 0.5 / resolution, min_area = q * q;
        hipLaunchKernelGGL(k_seed_stats, dim3(g.grid()), dim3(RB), 0, s, (const int*)lab.p, g.rows, g.cols, area.p, bnd.p);
        hipLaunchKernelGGL(k_seed_pick, dim3(g.grid()), dim3(RB), 0, s, (const int*)lab.p, g.n(), (const int*)area.p, (const int*)bnd.p, min_area,
                           roots.p, root_cap, nroots.p);
        HMSG_CHECK_LAUNCH();
        int n_rooms = 0;
        HIP_TRY(hipMemcpyAsync(&n_rooms, nroots.p, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        HMSG_REQUIRE(n_rooms <= root_cap, HMSG_ERR_UNSUPPORTED, "hmsg_segment_rooms: more than 4096 seeds");
        hipLaunchKernelGGL(k_seed_markers, dim3(g.grid()), dim3(RB), 0, s, (const int*)lab.p, g.rows, g.cols, (const int*)roots.p, (const int*)nroots.p,
                           mk_a.p);
        HMSG_CHECK_LAUNCH();
        // ---- watershed
        R.flag.ensure(1);
        int *cur = mk_a.p, *nxt = mk_b.p;
        for (int same = 1; same >= 0; --same)
            for (int round = 0; round < 1 << 16; ++round) {
                HIP_TRY(hipMemsetAsync(R.flag.p, 0, 4, s));
                for (int k = 0; k < 8; ++k) {
                    hipLaunchKernelGGL(k_ws_round, dim3(g.grid()), dim3(RB), 0, s, (const unsigned char*)full.p, (const int*)cur, g.rows, g.cols, same, nxt,
                                       R.flag.p);
                    std::swap(cur, nxt);
                }
                int ch = 0;
                HIP_TRY(hipMemcpyAsync(&ch, R.flag.p, 4, hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                if (!ch) break;
            }
        HMSG_CHECK_LAUNCH();
        HIP_TRY(hipMemcpyAsync(out_markers, cur, (size_t)g.n() * 4, hipMemcpyDefault, s));
        HIP_TRY(hipStreamSynchronize(s));
        *out_n_rooms = n_rooms;
        return HMSG_OK;
    } catch (const hmsg_error& e) {
        h->err = e.msg;
        return e.code;
    }
}
