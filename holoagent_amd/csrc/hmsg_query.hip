// A12: text-query -> node retrieval over a resident node-embedding table.
//
// Reference: Graph.query_hmsg_object (fsr_vln/memory/hmsg/graph/graph.py:3056-3162), and the plain
// similarity GEMVs of query_floor (:2231-2252) / query_hmsg_room (:3204-3272).
//
//   sim = T[C, D] . E[N', D]^T  (float32 text x float64 embeddings -> float64, graph.py:3127)
//   plain top-k by sim[qid] (descending); with negative prompts: objects whose arg-max class (first max)
//   is the query class, ordered by descending score; if there is none, the plain top-k (graph.py:3133-3151).
//   Exact score ties (bit-equal scores: duplicate embeddings) are broken by candidate position = room order, then
//   node order.  That is what np.argsort(-score) of the negative-prompt path gives for them; the plain path's
//   np.argsort(sim)[::-1] has no defined order for equal keys (numpy's default sort is not stable: for three
//   duplicates it returned 5, 0, 2 in tests/test_emu_parity.py::test_query_exact_score_ties), so only the SET of tied
//   nodes and their scores can be compared there.
//
// MI355X design: the table stays in HBM as float64; all Q x C text rows are scored against ALL nodes by
// one f64-MFMA GEMM (v_mfma_f64_16x16x4_f64), then one workgroup per query walks its candidate rooms
// (CSR room -> nodes) and keeps an exact top-k.
#include "hmsg_common.h"

#include <algorithm>
#include <chrono>

typedef double f64x4 __attribute__((ext_vector_type(4)));

struct hmsg_index {
    int device = 0;
    int D = 0;
    long long N = 0;
    int n_rooms = 0;
    hipStream_t stream = nullptr;
    std::string err;
    DevBuf<double> E;            // [N][D]
    DevBuf<int> room_of;         // [N]
    DevBuf<int> room_off;        // [n_rooms + 1]   CSR room -> nodes (ascending node index)
    DevBuf<int> room_nodes;      // [N]
    std::vector<int> h_room_cnt;
    DevBuf<double> T64, S;       // scratch: text rows in f64, similarity matrix
    DevBuf<float> Tf;
    DevBuf<int> d_qid, d_roff, d_rooms, d_oidx, d_oroom;
    DevBuf<double> d_oscore;
    // the hierarchy above the nodes (hmsg_index_set_hierarchy): floors -> rooms, room name / view embeddings
    bool have_hier = false;
    int n_floors = 0, h_rooms = 0;
    long long n_views = 0;
    DevBuf<double> room_name_emb;   // [n_rooms][D]  CLIP text embedding of the room's name (label mode)
    DevBuf<double> view_emb;        // [n_views][D]  room.embeddings (view mode)
    DevBuf<int> view_off;           // [n_rooms + 1]
    DevBuf<int> room_key;           // [n_rooms]     int(room_id.split("_")[-1]): what the view mode returns
    DevBuf<int> floor_room_off;     // [n_floors + 1]
    DevBuf<int> floor_rooms;        // rooms of floor f in floors[f].rooms order (global room ids)
    DevBuf<double> S_room, S_view;  // scratch
    DevBuf<float> Tr;
    DevBuf<double> Tr64;
    DevBuf<int> d_floor, d_mode, d_sel, d_nsel, d_err;
    // hmsg_query_hier: the per-query words in (floor | mode | qid) and every result out (score | sel | nsel | err | idx | room) travel as
    // ONE packed copy each way through pinned memory (round 5: three pageable uploads and six pageable read-backs per call)
    PinnedBuf<int> h_qin;
    DevBuf<int> d_qin;
    PinnedBuf<char> h_qout;
    DevBuf<char> d_qout;
    Prof prof;                   // live timing of the GEMM (hmsg_index_set_profiling)
};

__global__ void k_f32_to_f64(const float* __restrict__ a, double* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = (double)a[i];
}

// S[M][N] = A[M][D] . B[N][D]^T, one wave per 16x16 tile.
// v_mfma_f64_16x16x4_f64: lane l feeds A[i = l&15][k = l>>4], B[k = l>>4][j = l&15];
// result reg r of lane l is C[row = (l>>4) + 4r][col = l&15].
__global__ void __launch_bounds__(256) k_gemm_f64(const double* __restrict__ A, const double* __restrict__ B, int M, long long N,
                                                  int D, double* __restrict__ S) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long tn = (N + 15) / 16;
    long long tile = (long long)blockIdx.x * 4 + wv;
    const long long ntiles = (long long)((M + 15) / 16) * tn;
    const bool active = tile < ntiles;
    if (!active) tile = ntiles - 1;
    const int m0 = (int)(tile / tn) * 16;
    const long long n0 = (tile % tn) * 16;
    const int ar = m0 + (lane & 15);
    const long long br = n0 + (lane & 15);
    const double* ap = A + (size_t)(ar < M ? ar : M - 1) * D;
    const double* bp = B + (size_t)(br < N ? br : N - 1) * D;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    const int kq = lane >> 4;
    for (int k0 = 0; k0 < D; k0 += 4) {
        int k = k0 + kq;
        double a = (k < D && ar < M) ? ap[k] : 0.0;
        double b = (k < D && br < N) ? bp[k] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    if (!active) return;
    const long long col = n0 + (lane & 15);
    for (int r = 0; r < 4; ++r) {
        int row = m0 + (lane >> 4) + 4 * r;
        if (row < M && col < N) S[(size_t)row * N + col] = acc[r];
    }
}

// Tiled version for the batched query path: a 256-thread workgroup owns a 128x128 tile of S, the four waves a
// 2x2 arrangement of 64x64 quadrants = 4x4 accumulators of 16x16 (64 result registers per lane).  Per 16-wide k
// step the workgroup stages a 128x16 panel of A and of B in LDS (k-major, one 64-byte global load per thread and
// matrix, double-buffered: the loads of step t+1 are in flight while the 64 MFMAs of step t run), and every wave
// reads 4 + 4 fragments for 16 MFMAs.  128x128x16 multiply-adds per 32 KB staged = 16 FLOP per byte of L2 traffic
// (the one-wave-per-tile kernel above moves 1 byte per FLOP and has no reuse at all).
#define GT 128
#define GK 16
#define GPAD 2
__global__ void __launch_bounds__(256) k_gemm_f64_tiled(const double* __restrict__ A, const double* __restrict__ B, int M,
                                                        long long N, int D, double* __restrict__ S) {
    __shared__ double sa[2][GK][GT + GPAD];
    __shared__ double sb[2][GK][GT + GPAD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wr = wv >> 1, wc = wv & 1;
    const long long tn = (N + GT - 1) / GT;
    const int m0 = (int)(blockIdx.x / tn) * GT;
    const long long n0 = (long long)(blockIdx.x % tn) * GT;
    // staging: thread t loads 8 consecutive k of row t>>1 (64 bytes) of each matrix
    const int lrow = tid >> 1, lk = (tid & 1) * 8;
    const int ar = m0 + lrow < M ? m0 + lrow : M - 1;
    const long long br = n0 + lrow < N ? n0 + lrow : N - 1;
    const double* ap = A + (size_t)ar * D;
    const double* bp = B + (size_t)br * D;
    f64x4 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    double ra[8], rb[8];
    auto load = [&](int k0) {
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + lk + u;
            ra[u] = k < D ? ap[k] : 0.0;
            rb[u] = k < D ? bp[k] : 0.0;
        }
    };
    auto stage = [&](int buf) {
        for (int u = 0; u < 8; ++u) {
            sa[buf][lk + u][lrow] = ra[u];
            sb[buf][lk + u][lrow] = rb[u];
        }
    };
    load(0);
    stage(0);
    __syncthreads();
    const int nk = (D + GK - 1) / GK;
    const int kq = lane >> 4, li = lane & 15;
    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        if (t + 1 < nk) load((t + 1) * GK);
        for (int k = 0; k < GK; k += 4) {
            double fa[4], fb[4];
            for (int i = 0; i < 4; ++i) fa[i] = sa[buf][k + kq][wr * 64 + i * 16 + li];
            for (int j = 0; j < 4; ++j) fb[j] = sb[buf][k + kq][wc * 64 + j * 16 + li];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nk) stage(buf ^ 1);
        __syncthreads();
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const long long col = n0 + wc * 64 + j * 16 + li;
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr * 64 + i * 16 + kq + 4 * r;
                if (row < M && col < N) S[(size_t)row * N + col] = acc[i][j][r];
            }
        }
}

#define QK_MAX 64
struct TopK {
    double s[QK_MAX];
    int pos[QK_MAX];
    int node[QK_MAX];
    int room[QK_MAX];
    int n;
};
// (score desc, candidate position asc)
__device__ __forceinline__ bool better(double s1, int p1, double s2, int p2) { return s1 > s2 || (s1 == s2 && p1 < p2); }

// One workgroup per query.  Thread-private candidates are reduced through LDS by repeated arg-best
// selection (k is small), which keeps the result exact and deterministic.
__global__ void __launch_bounds__(256) k_query_topk(const double* __restrict__ S, long long N, int C, const int* __restrict__ qid,
                                                    const int* __restrict__ q_room_off, const int* __restrict__ q_rooms,
                                                    const int* __restrict__ room_off, const int* __restrict__ room_nodes,
                                                    int n_rooms, int k, int use_neg, int* __restrict__ out_idx,
                                                    int* __restrict__ out_room, double* __restrict__ out_score) {
    __shared__ double sh_s[256];
    __shared__ int sh_p[256];
    __shared__ int sh_any;
    __shared__ double last_s[2];
    __shared__ int last_p[2];
    const int q = blockIdx.x, tid = threadIdx.x;
    const int myq = qid[q];
    const double* Sq = S + (size_t)q * C * N;
    if (tid == 0) sh_any = 0;
    __syncthreads();
    // pass 0: does any candidate have arg-max class == query class?
    if (use_neg) {
        int pos0 = 0;
        int any = 0;
        for (int j = q_room_off[q]; j < q_room_off[q + 1]; ++j) {
            int r = q_rooms[j];
            if (r < 0 || r >= n_rooms) continue;
            int b = room_off[r], e = room_off[r + 1];
            for (int t = b + tid; t < e; t += 256) {
                int node = room_nodes[t];
                int cls = 0;
                double mx = Sq[node];
                for (int c = 1; c < C; ++c) {
                    double v = Sq[(size_t)c * N + node];
                    if (v > mx) {
                        mx = v;
                        cls = c;
                    }
                }
                any |= (cls == myq);
            }
            pos0 += e - b;
        }
        if (any) sh_any = 1;
    }
    __syncthreads();
    const bool filtered = use_neg && sh_any;
    // k rounds of "best candidate worse than the previous pick"
    if (tid == 0) {
        last_s[0] = 1e308;
        last_p[0] = -1;
    }
    __syncthreads();
    for (int round = 0; round < k; ++round) {
        const double ls = last_s[0];
        const int lp = last_p[0];
        double bs = -1e308;
        int bp = 0x7fffffff;
        int pos0 = 0;
        for (int j = q_room_off[q]; j < q_room_off[q + 1]; ++j) {
            int r = q_rooms[j];
            if (r < 0 || r >= n_rooms) continue;
            int b = room_off[r], e = room_off[r + 1];
            for (int t = b + tid; t < e; t += 256) {
                int node = room_nodes[t];
                int pos = pos0 + (t - b);
                double sc = Sq[(size_t)myq * N + node];
                if (filtered) {
                    int cls = 0;
                    double mx = Sq[node];
                    for (int c = 1; c < C; ++c) {
                        double v = Sq[(size_t)c * N + node];
                        if (v > mx) {
                            mx = v;
                            cls = c;
                        }
                    }
                    if (cls != myq) continue;
                }
                // strictly after the previous pick in (score desc, pos asc) order
                bool after = lp < 0 || sc < ls || (sc == ls && pos > lp);
                if (after && better(sc, pos, bs, bp)) {
                    bs = sc;
                    bp = pos;
                }
            }
            pos0 += e - b;
        }
        sh_s[tid] = bs;
        sh_p[tid] = bp;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o && better(sh_s[tid + o], sh_p[tid + o], sh_s[tid], sh_p[tid])) {
                sh_s[tid] = sh_s[tid + o];
                sh_p[tid] = sh_p[tid + o];
            }
            __syncthreads();
        }
        if (tid == 0) {
            int pos = sh_p[0];
            int oi = -1, orr = -1;
            double os = 0.0;
            if (pos != 0x7fffffff) {
                // position -> (room, node)
                int pos0b = 0;
                for (int j = q_room_off[q]; j < q_room_off[q + 1]; ++j) {
                    int r = q_rooms[j];
                    if (r < 0 || r >= n_rooms) continue;
                    int cnt = room_off[r + 1] - room_off[r];
                    if (pos < pos0b + cnt) {
                        oi = room_nodes[room_off[r] + pos - pos0b];
                        orr = r;
                        break;
                    }
                    pos0b += cnt;
                }
                os = sh_s[0];
            }
            out_idx[(size_t)q * k + round] = oi;
            out_room[(size_t)q * k + round] = orr;
            out_score[(size_t)q * k + round] = os;
            last_s[0] = sh_s[0];
            last_p[0] = pos == 0x7fffffff ? 0x7ffffffe : pos;
        }
        __syncthreads();
    }
}

// query_hmsg_room (graph.py:3164-3272), one workgroup per query.  rooms_list = self.rooms (floor -1) or floors[f].rooms.
//   mode 1 (label, :3204-3232): similarity of the room text with every room NAME of the list; every room within 1e-3 of
//        the best one, in list order; the numbers returned are positions in rooms_list.
//   mode 2 / 3 (view embeddings, :3247-3272): per room the largest similarity over its view embeddings; rooms sorted by
//        it, descending (Python's sorted: stable, ties keep the list order); the first 5 (mode 2) or 10 (mode 3); the
//        numbers returned are int(room_id.split("_")[-1]) -- which the caller then uses as positions in rooms_list.
//   mode 0: no room stage (every room of the list, in order).
// The selected numbers go to sel[q][0 .. nsel[q]); the object stage searches rooms_list[number] in that order
// (query_hmsg_object :3099-3110); a number that is no position of rooms_list raises IndexError there: err[q] = 1.
__global__ void __launch_bounds__(256) k_room_select(int n_rooms, int n_floors, const double* __restrict__ S_room,
                                                     const double* __restrict__ S_view, long long n_views,
                                                     const int* __restrict__ view_off, const int* __restrict__ room_key,
                                                     const int* __restrict__ floor_room_off, const int* __restrict__ floor_rooms,
                                                     const int* __restrict__ floor_id, const int* __restrict__ mode, int max_sel,
                                                     int* __restrict__ sel, int* __restrict__ nsel, int* __restrict__ q_rooms,
                                                     int* __restrict__ err) {
    const int q = blockIdx.x, tid = threadIdx.x;
    const int f = floor_id[q], m = mode[q];
    __shared__ int s_bad;
    if (tid == 0) s_bad = (f >= n_floors) ? 1 : 0;
    __syncthreads();
    const int L = f < 0 ? n_rooms : (s_bad ? 0 : floor_room_off[f + 1] - floor_room_off[f]);
    auto room_at = [&](int i) { return f < 0 ? i : floor_rooms[floor_room_off[f] + i]; };
    __shared__ double s_red[256];
    int* my_sel = sel + (size_t)q * max_sel;
    if (m == 1) {
        double best = -1e308;
        for (int i = tid; i < L; i += 256) best = fmax(best, S_room[(size_t)q * n_rooms + room_at(i)]);
        s_red[tid] = best;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) s_red[tid] = fmax(s_red[tid], s_red[tid + o]);
            __syncthreads();
        }
        best = s_red[0];
        if (tid == 0) {                                         // (a handful of rooms: in list order)
            int n = 0;
            for (int i = 0; i < L && n < max_sel; ++i)
                if (fabs(S_room[(size_t)q * n_rooms + room_at(i)] - best) < 1e-3) my_sel[n++] = i;
            nsel[q] = n;
        }
    } else if (m == 2 || m == 3) {
        // per room: max over its views (np.argmax takes the first maximum; only the value matters here)
        __shared__ double s_max[1024];
        __shared__ unsigned char s_taken[1024];
        const int Lc = min(L, 1024);
        for (int i = tid; i < Lc; i += 256) {
            const int r = room_at(i);
            double mx = -1e308;
            for (int v = view_off[r]; v < view_off[r + 1]; ++v) mx = fmax(mx, S_view[(size_t)q * n_views + v]);
            s_max[i] = mx;
            s_taken[i] = 0;
            if (view_off[r + 1] == view_off[r]) s_bad = 1;      // np.stack([]) raises
        }
        __syncthreads();
        if (tid == 0) {
            if (L > 1024) s_bad = 1;
            // graph.py:3259-3264: `{int(room_id.split("_")[-1]): v for ... in sorted(...)}` -- rooms "0_2" and "1_2" (floor -1 on
            // a multi-storey graph) collapse into ONE key, which keeps the place of its first (best) occurrence; the first
            // 5 / 10 UNIQUE keys are returned.
            const int want = m == 2 ? 5 : 10;
            int n = 0;
            for (int taken = 0; taken < Lc && n < want && n < max_sel; ++taken) {   // selection sort of the top few, first index wins ties
                int bi = -1;
                double bv = -1e308;
                for (int i = 0; i < Lc; ++i)
                    if (!s_taken[i] && (bi < 0 || s_max[i] > bv)) {
                        bi = i;
                        bv = s_max[i];
                    }
                s_taken[bi] = 1;
                const int key = room_key[room_at(bi)];
                bool seen = false;
                for (int j = 0; j < n; ++j) seen = seen || my_sel[j] == key;
                if (!seen) my_sel[n++] = key;
            }
            nsel[q] = n;
        }
    } else if (tid == 0) {
        int n = 0;
        for (int i = 0; i < L && n < max_sel; ++i) my_sel[n++] = i;
        nsel[q] = n;
    }
    __syncthreads();
    // positions of rooms_list -> room ids for the object stage
    if (tid == 0) {
        const int n = nsel[q];
        for (int j = 0; j < max_sel; ++j) {
            int r = -1;
            if (j < n) {
                const int pos = my_sel[j];
                if (pos < 0 || pos >= L) s_bad = 1;
                else r = room_at(pos);
            }
            q_rooms[(size_t)q * max_sel + j] = r;
        }
        err[q] = s_bad;
    }
}
__global__ void k_fill_offsets(int* off, int n, int stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) off[i] = i * stride;
}

namespace {
template <typename F>
int iguard(hmsg_index* ix, F&& fn) {
    try {
        HIP_TRY(hipSetDevice(ix->device));
        fn();
        return HMSG_OK;
    } catch (const hmsg_error& e) {
        ix->err = e.msg;
        return e.code;
    } catch (const std::exception& e) {
        ix->err = e.what();
        return HMSG_ERR_INVALID;
    } catch (...) {
        ix->err = "unknown error";
        return HMSG_ERR_INVALID;
    }
}
bool dev_ptr(const void* p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}
// S[M][N] = A[M][D] . B[N][D]^T  (B = the node table unless given)
void gemm(hmsg_index* ix, const double* A, int M, double* S, const double* B = nullptr, long long N = -1) {
    if (!B) {
        B = ix->E.p;
        N = ix->N;
    }
    if (M >= 64 && N >= 64) {
        const long long tiles = (long long)((M + GT - 1) / GT) * ((N + GT - 1) / GT);
        ProfScope ps(ix->prof, ix->stream, "k_gemm_f64", 2.0 * (double)M * (double)N * (double)ix->D);
        hipLaunchKernelGGL(k_gemm_f64_tiled, dim3((unsigned)tiles), dim3(256), 0, ix->stream, A, B, M, N, ix->D, S);
        HMSG_CHECK_LAUNCH();
        return;
    }
    long long tiles = (long long)((M + 15) / 16) * ((N + 15) / 16);
    hipLaunchKernelGGL(k_gemm_f64, dim3(cdiv((size_t)tiles, 4)), dim3(256), 0, ix->stream, A, B, M, N, ix->D, S);
    HMSG_CHECK_LAUNCH();
}
void text_to_f64(hmsg_index* ix, const float* T, size_t n) {
    ix->T64.ensure(n);
    const float* src = T;
    if (!dev_ptr(T)) {
        ix->Tf.ensure(n);
        h2d_bounce(ix->Tf.p, T, n * 4, ix->stream);
        src = ix->Tf.p;
    }
    hipLaunchKernelGGL(k_f32_to_f64, dim3(cdiv(n, 256)), dim3(256), 0, ix->stream, src, ix->T64.p, n);
    HMSG_CHECK_LAUNCH();
}
}  // namespace

extern "C" {

int hmsg_index_create(int32_t device_id, int32_t dim, int64_t n, const void* emb, int32_t emb_is_f64, const int32_t* room_of_node,
                      hmsg_index_t** out) {
    if (!out) return HMSG_ERR_INVALID;
    *out = nullptr;
    if (dim <= 0 || n <= 0 || !emb || !room_of_node) return HMSG_ERR_INVALID;
    hmsg_index* ix = new hmsg_index();
    ix->device = device_id;
    ix->D = dim;
    ix->N = n;
    int rc = iguard(ix, [&] {
        HIP_TRY(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
        ix->E.alloc((size_t)n * dim);
        const size_t cnt = (size_t)n * dim;
        if (emb_is_f64) {
            if (dev_ptr(emb)) HIP_TRY(hipMemcpyAsync(ix->E.p, emb, cnt * 8, hipMemcpyDeviceToDevice, ix->stream));
            else h2d_bounce(ix->E.p, emb, cnt * 8, ix->stream);
        } else {
            DevBuf<float> tmp;
            tmp.alloc(cnt);
            if (dev_ptr(emb)) HIP_TRY(hipMemcpyAsync(tmp.p, emb, cnt * 4, hipMemcpyDeviceToDevice, ix->stream));
            else h2d_bounce(tmp.p, emb, cnt * 4, ix->stream);
            hipLaunchKernelGGL(k_f32_to_f64, dim3(cdiv(cnt, 256)), dim3(256), 0, ix->stream, (const float*)tmp.p, ix->E.p, cnt);
            HMSG_CHECK_LAUNCH();
            HIP_TRY(hipStreamSynchronize(ix->stream));
        }
        std::vector<int> rooms((size_t)n);
        if (dev_ptr(room_of_node)) {
            HIP_TRY(hipMemcpy(rooms.data(), room_of_node, (size_t)n * 4, hipMemcpyDeviceToHost));
        } else {
            memcpy(rooms.data(), room_of_node, (size_t)n * 4);
        }
        int nr = 0;
        for (int r : rooms) {
            HMSG_REQUIRE(r >= 0, HMSG_ERR_INVALID, "negative room id");
            nr = std::max(nr, r + 1);
        }
        ix->n_rooms = nr;
        std::vector<int> off(nr + 1, 0), nodes((size_t)n);
        for (int r : rooms) off[r + 1]++;
        for (int r = 0; r < nr; ++r) off[r + 1] += off[r];
        std::vector<int> cur(off.begin(), off.end() - 1);
        for (long long i = 0; i < n; ++i) nodes[cur[rooms[i]]++] = (int)i;
        ix->room_of.alloc((size_t)n);
        ix->room_off.alloc(nr + 1);
        ix->room_nodes.alloc((size_t)n);
        HIP_TRY(hipMemcpyAsync(ix->room_of.p, rooms.data(), (size_t)n * 4, hipMemcpyHostToDevice, ix->stream));
        HIP_TRY(hipMemcpyAsync(ix->room_off.p, off.data(), (size_t)(nr + 1) * 4, hipMemcpyHostToDevice, ix->stream));
        HIP_TRY(hipMemcpyAsync(ix->room_nodes.p, nodes.data(), (size_t)n * 4, hipMemcpyHostToDevice, ix->stream));
        HIP_TRY(hipStreamSynchronize(ix->stream));
    });
    if (rc != HMSG_OK) {
        fprintf(stderr, "hmsg_index_create: %s\n", ix->err.c_str());
        delete ix;
        return rc;
    }
    *out = ix;
    return HMSG_OK;
}

void hmsg_index_destroy(hmsg_index_t* ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    if (ix->stream) {
        (void)hipStreamSynchronize(ix->stream);
        (void)hipStreamDestroy(ix->stream);
    }
    ix->prof.clear();
    delete ix;
}

const char* hmsg_index_last_error(const hmsg_index_t* ix) { return ix ? ix->err.c_str() : "null index"; }

int hmsg_index_set_profiling(hmsg_index_t* ix, int32_t on) {
    if (!ix) return HMSG_ERR_INVALID;
    ix->prof.enabled = on != 0;
    return HMSG_OK;
}
// launches, total milliseconds and total FLOP of the similarity GEMM since the index was created
int hmsg_index_profile(hmsg_index_t* ix, int64_t* launches, double* total_ms, double* total_flop) {
    if (!ix || !launches || !total_ms || !total_flop) return HMSG_ERR_INVALID;
    (void)hipStreamSynchronize(ix->stream);
    *launches = 0;
    *total_ms = *total_flop = 0.0;
    for (auto& e : ix->prof.ev) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, e.a, e.b) != hipSuccess) continue;
        ++*launches;
        *total_ms += t;
        *total_flop += e.work;
    }
    return HMSG_OK;
}

int hmsg_query_objects(hmsg_index_t* ix, int32_t Q, int32_t C, const float* T, const int32_t* qid, const int32_t* room_off,
                       const int32_t* rooms, int32_t k, int32_t use_negatives, int32_t* out_idx, int32_t* out_room,
                       double* out_score) {
    if (!ix) return HMSG_ERR_INVALID;
    return iguard(ix, [&] {
        HMSG_REQUIRE(Q >= 0 && C >= 1 && T && qid && room_off && k >= 1 && out_idx && out_room && out_score, HMSG_ERR_INVALID,
                     "hmsg_query_objects: bad argument");
        if (Q == 0) return;
        const size_t nT = (size_t)Q * C * ix->D;
        text_to_f64(ix, T, nT);
        ix->S.ensure((size_t)Q * C * ix->N);
        gemm(ix, ix->T64.p, Q * C, ix->S.p);
        std::vector<int> hoff(Q + 1);
        if (dev_ptr(room_off)) {
            HIP_TRY(hipMemcpy(hoff.data(), room_off, (size_t)(Q + 1) * 4, hipMemcpyDeviceToHost));
        } else {
            memcpy(hoff.data(), room_off, (size_t)(Q + 1) * 4);
        }
        const int nr = hoff[Q];
        HMSG_REQUIRE(nr == 0 || rooms, HMSG_ERR_INVALID, "rooms list missing");
        ix->d_qid.ensure(Q);
        ix->d_roff.ensure(Q + 1);
        ix->d_rooms.ensure(std::max(nr, 1));
        ix->d_oidx.ensure((size_t)Q * k);
        ix->d_oroom.ensure((size_t)Q * k);
        ix->d_oscore.ensure((size_t)Q * k);
        HIP_TRY(hipMemcpyAsync(ix->d_qid.p, qid, (size_t)Q * 4, dev_ptr(qid) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ix->stream));
        HIP_TRY(hipMemcpyAsync(ix->d_roff.p, hoff.data(), (size_t)(Q + 1) * 4, hipMemcpyHostToDevice, ix->stream));
        if (nr) HIP_TRY(hipMemcpyAsync(ix->d_rooms.p, rooms, (size_t)nr * 4, dev_ptr(rooms) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ix->stream));
        hipLaunchKernelGGL(k_query_topk, dim3(Q), dim3(256), 0, ix->stream, (const double*)ix->S.p, ix->N, C, (const int*)ix->d_qid.p,
                           (const int*)ix->d_roff.p, (const int*)ix->d_rooms.p, (const int*)ix->room_off.p,
                           (const int*)ix->room_nodes.p, ix->n_rooms, k, use_negatives, ix->d_oidx.p, ix->d_oroom.p, ix->d_oscore.p);
        HMSG_CHECK_LAUNCH();
        HIP_TRY(hipMemcpyAsync(out_idx, ix->d_oidx.p, (size_t)Q * k * 4, hipMemcpyDeviceToHost, ix->stream));
        HIP_TRY(hipMemcpyAsync(out_room, ix->d_oroom.p, (size_t)Q * k * 4, hipMemcpyDeviceToHost, ix->stream));
        HIP_TRY(hipMemcpyAsync(out_score, ix->d_oscore.p, (size_t)Q * k * 8, hipMemcpyDeviceToHost, ix->stream));
        HIP_TRY(hipStreamSynchronize(ix->stream));
    });
}

int hmsg_index_set_hierarchy(hmsg_index_t* ix, int32_t n_rooms, int32_t n_floors, const int32_t* floor_room_off, const int32_t* floor_rooms,
                             const double* room_name_emb, const int64_t* view_off, const double* view_emb, const int32_t* room_key) {
    if (!ix) return HMSG_ERR_INVALID;
    return iguard(ix, [&] {
        const int R = n_rooms;                   // (rooms without objects included: >= the largest room id of a node + 1)
        HMSG_REQUIRE(R >= ix->n_rooms && n_floors >= 0 && (n_floors == 0 || (floor_room_off && floor_rooms)) && view_off && room_key,
                     HMSG_ERR_INVALID, "hmsg_index_set_hierarchy: bad argument");
        ix->h_rooms = R;
        for (int f = 0; f < n_floors; ++f)
            for (int j = floor_room_off[f]; j < floor_room_off[f + 1]; ++j)
                HMSG_REQUIRE(floor_rooms[j] >= 0 && floor_rooms[j] < R, HMSG_ERR_INVALID, "hmsg_index_set_hierarchy: room id out of range");
        const long long NV = view_off[R];
        HMSG_REQUIRE(NV >= 0 && NV < (1ll << 31) && (NV == 0 || view_emb), HMSG_ERR_INVALID, "hmsg_index_set_hierarchy: bad view table");
        ix->n_floors = n_floors;
        ix->n_views = NV;
        std::vector<int> voff((size_t)R + 1);
        for (int r = 0; r <= R; ++r) voff[(size_t)r] = (int)view_off[r];
        ix->view_off.alloc((size_t)R + 1);
        ix->room_key.alloc((size_t)std::max(R, 1));
        ix->floor_room_off.alloc((size_t)n_floors + 1);
        const int nfr = n_floors ? floor_room_off[n_floors] : 0;
        ix->floor_rooms.alloc((size_t)std::max(nfr, 1));
        HIP_TRY(hipMemcpyAsync(ix->view_off.p, voff.data(), ((size_t)R + 1) * 4, hipMemcpyHostToDevice, ix->stream));
        HIP_TRY(hipMemcpyAsync(ix->room_key.p, room_key, (size_t)R * 4, hipMemcpyHostToDevice, ix->stream));
        std::vector<int> zero(1, 0);
        HIP_TRY(hipMemcpyAsync(ix->floor_room_off.p, n_floors ? floor_room_off : zero.data(), ((size_t)n_floors + 1) * 4, hipMemcpyHostToDevice, ix->stream));
        if (nfr) HIP_TRY(hipMemcpyAsync(ix->floor_rooms.p, floor_rooms, (size_t)nfr * 4, hipMemcpyHostToDevice, ix->stream));
        if (room_name_emb) {
            ix->room_name_emb.alloc((size_t)std::max(R, 1) * ix->D);
            if (dev_ptr(room_name_emb)) HIP_TRY(hipMemcpyAsync(ix->room_name_emb.p, room_name_emb, (size_t)R * ix->D * 8, hipMemcpyDeviceToDevice, ix->stream));
            else h2d_bounce(ix->room_name_emb.p, room_name_emb, (size_t)R * ix->D * 8, ix->stream);
        } else {
            ix->room_name_emb.release();
        }
        ix->view_emb.alloc((size_t)std::max<long long>(NV, 1) * ix->D);
        if (NV) {
            if (dev_ptr(view_emb)) HIP_TRY(hipMemcpyAsync(ix->view_emb.p, view_emb, (size_t)NV * ix->D * 8, hipMemcpyDeviceToDevice, ix->stream));
            else h2d_bounce(ix->view_emb.p, view_emb, (size_t)NV * ix->D * 8, ix->stream);
        }
        HIP_TRY(hipStreamSynchronize(ix->stream));
        ix->have_hier = true;
    });
}

int hmsg_query_hier(hmsg_index_t* ix, int32_t Q, int32_t C, const float* T_obj, const int32_t* qid, const float* T_room,
                    const int32_t* floor_id, const int32_t* room_mode, int32_t k, int32_t use_negatives, int32_t max_rooms,
                    int32_t* out_sel, int32_t* out_nsel, int32_t* out_idx, int32_t* out_room, double* out_score) {
    if (!ix) return HMSG_ERR_INVALID;
    return iguard(ix, [&] {
        HMSG_REQUIRE(ix->have_hier, HMSG_ERR_INVALID, "hmsg_query_hier: hmsg_index_set_hierarchy first");
        HMSG_REQUIRE(Q >= 0 && C >= 1 && T_obj && qid && floor_id && room_mode && k >= 1 && max_rooms >= 1 && out_sel && out_nsel && out_idx &&
                         out_room && out_score,
                     HMSG_ERR_INVALID, "hmsg_query_hier: bad argument");
        if (Q == 0) return;
        const int R = ix->h_rooms;
        // HMSG_DEBUG_TIMING: where a call spends its time (each lap drains the stream first)
        static const bool dbg = getenv("HMSG_DEBUG_TIMING") != nullptr;
        auto t_prev = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!dbg) return;
            (void)hipStreamSynchronize(ix->stream);
            const auto t = std::chrono::steady_clock::now();
            fprintf(stderr, "[hmsg query_hier] %-22s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
            t_prev = t;
        };
        std::vector<int> hm((size_t)Q), hf((size_t)Q);
        memcpy(hm.data(), room_mode, (size_t)Q * 4);
        memcpy(hf.data(), floor_id, (size_t)Q * 4);
        bool need_label = false, need_view = false;
        for (int q = 0; q < Q; ++q) {
            HMSG_REQUIRE(hm[q] >= 0 && hm[q] <= 3 && hf[q] >= -1 && hf[q] < ix->n_floors, HMSG_ERR_INVALID, "hmsg_query_hier: bad floor id / room mode");
            need_label |= hm[q] == 1;
            need_view |= hm[q] >= 2;
        }
        HMSG_REQUIRE(!(need_label || need_view) || T_room, HMSG_ERR_INVALID, "hmsg_query_hier: room text rows missing");
        HMSG_REQUIRE(!need_label || ix->room_name_emb.p, HMSG_ERR_INVALID, "hmsg_query_hier: label mode without room name embeddings");
        // room stage: similarities of the room text with the room names / the view embeddings (float64 MFMA GEMM)
        ix->S_room.ensure((size_t)Q * std::max(R, 1));
        ix->S_view.ensure((size_t)Q * std::max<long long>(ix->n_views, 1));
        if (need_label || need_view) {
            const size_t nT = (size_t)Q * ix->D;
            ix->Tr64.ensure(nT);
            const float* src = T_room;
            if (!dev_ptr(T_room)) {
                ix->Tr.ensure(nT);
                h2d_bounce(ix->Tr.p, T_room, nT * 4, ix->stream);
                src = ix->Tr.p;
            }
            hipLaunchKernelGGL(k_f32_to_f64, dim3(cdiv(nT, 256)), dim3(256), 0, ix->stream, src, ix->Tr64.p, nT);
            HMSG_CHECK_LAUNCH();
            if (need_label) gemm(ix, ix->Tr64.p, Q, ix->S_room.p, ix->room_name_emb.p, R);
            if (need_view && ix->n_views) gemm(ix, ix->Tr64.p, Q, ix->S_view.p, ix->view_emb.p, ix->n_views);
        }
        lap("room text + room GEMM");
        // per-query words: one packed upload from pinned memory
        const bool qid_dev = dev_ptr(qid);
        ix->h_qin.ensure((size_t)Q * 3 + 4);
        ix->d_qin.ensure((size_t)Q * 3 + 4);
        memcpy(ix->h_qin.p, hf.data(), (size_t)Q * 4);
        memcpy(ix->h_qin.p + Q, hm.data(), (size_t)Q * 4);
        if (!qid_dev) memcpy(ix->h_qin.p + 2 * (size_t)Q, qid, (size_t)Q * 4);
        upload_pinned(ix->d_qin.p, ix->h_qin.p, (((size_t)Q * 3 + 3) / 4) * 16, ix->stream);
        int* const d_floor = ix->d_qin.p;
        int* const d_mode = ix->d_qin.p + Q;
        int* const d_qid = ix->d_qin.p + 2 * (size_t)Q;
        if (qid_dev) HIP_TRY(hipMemcpyAsync(d_qid, qid, (size_t)Q * 4, hipMemcpyDeviceToDevice, ix->stream));
        // results: one packed buffer [score f64 Q*k | sel Q*max_rooms | nsel Q | err Q | idx Q*k | room Q*k]
        const size_t o_sel = (size_t)Q * k * 8, o_nsel = o_sel + (size_t)Q * max_rooms * 4, o_err = o_nsel + (size_t)Q * 4,
                     o_idx = o_err + (size_t)Q * 4, o_room = o_idx + (size_t)Q * k * 4, out_bytes = o_room + (size_t)Q * k * 4;
        ix->d_qout.ensure(out_bytes);
        ix->h_qout.ensure(out_bytes);
        double* const d_oscore = (double*)ix->d_qout.p;
        int* const d_sel = (int*)(ix->d_qout.p + o_sel);
        int* const d_nsel = (int*)(ix->d_qout.p + o_nsel);
        int* const d_err = (int*)(ix->d_qout.p + o_err);
        int* const d_oidx = (int*)(ix->d_qout.p + o_idx);
        int* const d_oroom = (int*)(ix->d_qout.p + o_room);
        ix->d_rooms.ensure((size_t)Q * max_rooms);
        ix->d_roff.ensure((size_t)Q + 1);
        hipLaunchKernelGGL(k_room_select, dim3(Q), dim3(256), 0, ix->stream, R, ix->n_floors, (const double*)ix->S_room.p,
                           (const double*)ix->S_view.p, ix->n_views, (const int*)ix->view_off.p, (const int*)ix->room_key.p,
                           (const int*)ix->floor_room_off.p, (const int*)ix->floor_rooms.p, (const int*)d_floor,
                           (const int*)d_mode, max_rooms, d_sel, d_nsel, ix->d_rooms.p, d_err);
        hipLaunchKernelGGL(k_fill_offsets, dim3(cdiv((size_t)Q + 1, 256)), dim3(256), 0, ix->stream, ix->d_roff.p, Q, max_rooms);
        HMSG_CHECK_LAUNCH();
        lap("room select");
        // object stage on the rooms the room stage picked, in that order
        const size_t nT = (size_t)Q * C * ix->D;
        text_to_f64(ix, T_obj, nT);
        lap("object text upload");
        ix->S.ensure((size_t)Q * C * ix->N);
        lap("S alloc");
        gemm(ix, ix->T64.p, Q * C, ix->S.p);
        lap("object GEMM");
        hipLaunchKernelGGL(k_query_topk, dim3(Q), dim3(256), 0, ix->stream, (const double*)ix->S.p, ix->N, C, (const int*)d_qid,
                           (const int*)ix->d_roff.p, (const int*)ix->d_rooms.p, (const int*)ix->room_off.p,
                           (const int*)ix->room_nodes.p, ix->n_rooms, k, use_negatives, d_oidx, d_oroom, d_oscore);
        HMSG_CHECK_LAUNCH();
        lap("top-k");
        HIP_TRY(hipMemcpyAsync(ix->h_qout.p, ix->d_qout.p, out_bytes, hipMemcpyDeviceToHost, ix->stream));
        HIP_TRY(hipStreamSynchronize(ix->stream));
        const int* const herr = (const int*)(ix->h_qout.p + o_err);
        // (results into the caller's arrays: host memory, or device memory -- the reference's callers take numpy arrays)
        bool any_dev = false;
        auto give = [&](void* dst, size_t off, size_t bytes) {
            if (dev_ptr(dst)) {
                HIP_TRY(hipMemcpyAsync(dst, ix->d_qout.p + off, bytes, hipMemcpyDeviceToDevice, ix->stream));
                any_dev = true;
            } else {
                memcpy(dst, ix->h_qout.p + off, bytes);
            }
        };
        give(out_score, 0, (size_t)Q * k * 8);
        give(out_sel, o_sel, (size_t)Q * max_rooms * 4);
        give(out_nsel, o_nsel, (size_t)Q * 4);
        give(out_idx, o_idx, (size_t)Q * k * 4);
        give(out_room, o_room, (size_t)Q * k * 4);
        if (any_dev) HIP_TRY(hipStreamSynchronize(ix->stream));
        lap("read-back");
        for (int q = 0; q < Q; ++q)
            HMSG_REQUIRE(!herr[(size_t)q], HMSG_ERR_INVALID,
                         "hmsg_query_hier: a query's room stage failed like the reference would (a room without view embeddings, or a "
                         "view-mode room number that is no position of the floor's room list)");
    });
}

int hmsg_similarity(hmsg_index_t* ix, int32_t Q, const float* T, double* S) {
    if (!ix) return HMSG_ERR_INVALID;
    return iguard(ix, [&] {
        HMSG_REQUIRE(Q >= 0 && T && S, HMSG_ERR_INVALID, "hmsg_similarity: bad argument");
        if (Q == 0) return;
        text_to_f64(ix, T, (size_t)Q * ix->D);
        ix->S.ensure((size_t)Q * ix->N);
        gemm(ix, ix->T64.p, Q, ix->S.p);
        HIP_TRY(hipMemcpyAsync(S, ix->S.p, (size_t)Q * ix->N * 8, hipMemcpyDeviceToHost, ix->stream));
        HIP_TRY(hipStreamSynchronize(ix->stream));
    });
}

}  // extern "C"
