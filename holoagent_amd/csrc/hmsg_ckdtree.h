// Host restatement of scipy.spatial.cKDTree (scipy 1.15.3: spatial/ckdtree/src/build.cxx, query.cxx,
// ordered_pair/heap helpers) for ONE purpose: deciding which of several BIT-EQUAL nearest neighbours
// `tree.query(x, k=1)` returns (graph.py:362-364 builds the tree, :409 / :458 and generic.py:181 query it).
//
// scipy is a pinned dependency of the reference (environment.yaml) that is not vendored in /root/reference; its
// published algorithm is restated here: default construction (leafsize 16, compact_nodes, balanced_tree: median
// split by std::nth_element over the node's indices compared by coordinate, then a partition that sends coordinates
// equal to the split to the right, sliding when a side would be empty) and the k = 1, eps = 0, p = 2 query
// (best-first descent with a binary heap of far children keyed by their lower-bound distance, leaf scan that
// accepts a point only when it is STRICTLY closer than the best so far -- so among bit-equal candidates the first
// one the traversal meets wins).  tests/test_ckdtree.py pins it against scipy itself: node table, index
// permutation and answers on tie-rich data.
//
// The GPU does the exact nearest-neighbour search; only the queries it flags as bit-equal ties (a point that is
// the exact float64 midpoint of two map voxels) come here.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

struct CKDNode {
    int split_dim;          // -1: leaf
    double split;
    int64_t start_idx, end_idx;
    int less, greater;      // node indices
};

struct CKDTree {
    static constexpr int M = 3;
    static constexpr int64_t LEAFSIZE = 16;
    const double* data = nullptr;       // [n][3], caller keeps it alive
    int64_t n = 0;
    std::vector<int64_t> indices;
    std::vector<CKDNode> nodes;
    double mins[M], maxes[M];

    void build(const double* pts, int64_t count) {
        data = pts;
        n = count;
        indices.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) indices[(size_t)i] = i;
        nodes.clear();
        nodes.reserve((size_t)(n / 4 + 16));
        for (int a = 0; a < M; ++a) mins[a] = maxes[a] = n ? data[a] : 0.0;
        for (int64_t i = 1; i < n; ++i)
            for (int a = 0; a < M; ++a) {
                const double v = data[i * M + a];
                maxes[a] = maxes[a] > v ? maxes[a] : v;
                mins[a] = mins[a] < v ? mins[a] : v;
            }
        if (n == 0) return;
        double mx[M], mn[M];
        std::memcpy(mx, maxes, sizeof(mx));
        std::memcpy(mn, mins, sizeof(mn));
        build_node(0, n, mx, mn);
    }

    int build_node(int64_t start_idx, int64_t end_idx, double* mx, double* mn) {
        nodes.push_back(CKDNode{-1, 0.0, start_idx, end_idx, -1, -1});
        const int node_index = (int)nodes.size() - 1;
        if (end_idx - start_idx <= LEAFSIZE) return node_index;
        int64_t* idx = indices.data();
        // compact_nodes: recompute the hyper-rectangle of this node's points
        {
            const double* p0 = data + idx[start_idx] * M;
            for (int a = 0; a < M; ++a) mx[a] = mn[a] = p0[a];
            for (int64_t j = start_idx + 1; j < end_idx; ++j) {
                const double* pj = data + idx[j] * M;
                for (int a = 0; a < M; ++a) {
                    const double t = pj[a];
                    mx[a] = mx[a] > t ? mx[a] : t;
                    mn[a] = mn[a] < t ? mn[a] : t;
                }
            }
        }
        int d = 0;
        double size = 0;
        for (int a = 0; a < M; ++a)
            if (mx[a] - mn[a] > size) {
                d = a;
                size = mx[a] - mn[a];
            }
        const double maxval = mx[d], minval = mn[d];
        if (maxval == minval) return node_index;      // all points identical: leaf
        // balanced_tree: median by nth_element over the indices, compared by the split coordinate
        const double* dat = data;
        const int64_t half = (end_idx - start_idx) / 2;
        std::nth_element(idx + start_idx, idx + start_idx + half, idx + end_idx, [dat, d](int64_t a, int64_t b) {
            return dat[a * M + d] < dat[b * M + d];
        });
        double split = data[idx[start_idx + half] * M + d];
        int64_t p = start_idx, q = end_idx - 1;
        while (p <= q) {
            if (data[idx[p] * M + d] < split) ++p;
            else if (data[idx[q] * M + d] >= split) --q;
            else {
                std::swap(idx[p], idx[q]);
                ++p;
                --q;
            }
        }
        if (p == start_idx) {                 // no point below the split: slide to the smallest value
            int64_t j = start_idx;
            split = data[idx[j] * M + d];
            for (int64_t i = start_idx + 1; i < end_idx; ++i)
                if (data[idx[i] * M + d] < split) {
                    j = i;
                    split = data[idx[j] * M + d];
                }
            std::swap(idx[start_idx], idx[j]);
            p = start_idx + 1;
            q = start_idx;
        } else if (p == end_idx) {            // no point at or above the split: slide to the largest value
            int64_t j = end_idx - 1;
            split = data[idx[j] * M + d];
            for (int64_t i = start_idx; i < end_idx - 1; ++i)
                if (data[idx[i] * M + d] > split) {
                    j = i;
                    split = data[idx[j] * M + d];
                }
            std::swap(idx[end_idx - 1], idx[j]);
            p = end_idx - 1;
            q = end_idx - 2;
        }
        const int less = build_node(start_idx, p, mx, mn);
        const int greater = build_node(p, end_idx, mx, mn);
        CKDNode& nd = nodes[(size_t)node_index];
        nd.less = less;
        nd.greater = greater;
        nd.split_dim = d;
        nd.split = split;
        return node_index;
    }

    // ---- query(x, k=1, eps=0, p=2, distance_upper_bound=inf)
    struct NodeInfo {
        int node;
        double min_distance;
        double side[M];
    };
    struct HeapItem {
        double priority;
        int ni;             // index into the query's NodeInfo pool
    };
    // scipy's own array heap (ordering of equal priorities follows its sift rules)
    struct Heap {
        std::vector<HeapItem> h;
        void push(const HeapItem& it) {
            h.push_back(it);
            size_t i = h.size() - 1;
            while (i > 0 && h[i].priority < h[(i - 1) / 2].priority) {
                std::swap(h[i], h[(i - 1) / 2]);
                i = (i - 1) / 2;
            }
        }
        HeapItem pop() {
            HeapItem top = h[0];
            h[0] = h.back();
            h.pop_back();
            const size_t nn = h.size();
            size_t i = 0, j = 1, k = 2;
            while ((j < nn && h[i].priority > h[j].priority) || (k < nn && h[i].priority > h[k].priority)) {
                const size_t l = (k < nn && h[j].priority > h[k].priority) ? k : j;
                std::swap(h[l], h[i]);
                i = l;
                j = 2 * i + 1;
                k = 2 * i + 2;
            }
            return top;
        }
    };

    static inline double sqdist(const double* u, const double* v) {   // sqeuclidean_distance_double for m = 3
        double s = 0.0;
        for (int i = 0; i < M; ++i) {
            const double d = u[i] - v[i];
            s += d * d;
        }
        return s;
    }

    int64_t query1(const double* x, double* out_d2 = nullptr) const {
        if (n == 0) return -1;
        std::vector<NodeInfo> pool;
        pool.reserve(64);
        Heap q;
        pool.push_back(NodeInfo{0, 0.0, {0.0, 0.0, 0.0}});
        int ni1 = 0;
        for (int i = 0; i < M; ++i) {
            double s = 0.0, t = x[i] - maxes[i];
            if (t > s) s = t;
            else {
                t = mins[i] - x[i];
                if (t > s) s = t;
            }
            const double sd = s * s;
            pool[0].min_distance += sd - pool[0].side[i];
            pool[0].side[i] = sd;
        }
        double bound = __builtin_inf();
        int64_t best = -1;
        for (;;) {
            const CKDNode& node = nodes[(size_t)pool[(size_t)ni1].node];
            if (node.split_dim == -1) {
                for (int64_t i = node.start_idx; i < node.end_idx; ++i) {
                    const int64_t pi = indices[(size_t)i];
                    const double d = sqdist(data + pi * M, x);
                    if (d < bound) {
                        bound = d;
                        best = pi;
                    }
                }
                if (q.h.empty()) break;
                ni1 = q.pop().ni;
            } else {
                if (pool[(size_t)ni1].min_distance > bound) break;
                pool.push_back(pool[(size_t)ni1]);
                int ni2 = (int)pool.size() - 1;
                const int sd = node.split_dim;
                double side;
                if (x[sd] < node.split) {
                    pool[(size_t)ni1].node = node.less;
                    pool[(size_t)ni2].node = node.greater;
                    side = node.split - x[sd];
                } else {
                    pool[(size_t)ni1].node = node.greater;
                    pool[(size_t)ni2].node = node.less;
                    side = x[sd] - node.split;
                }
                side = side * side;
                NodeInfo& far = pool[(size_t)ni2];
                far.min_distance += side - far.side[sd];
                far.side[sd] = side;
                if (pool[(size_t)ni1].min_distance > pool[(size_t)ni2].min_distance) std::swap(ni1, ni2);
                if (pool[(size_t)ni2].min_distance <= bound) q.push(HeapItem{pool[(size_t)ni2].min_distance, ni2});
            }
        }
        if (out_d2) *out_d2 = bound;
        return best;
    }
};
