/* hmsg_test.h -- test hooks and the benchmark's scene renderer of libhmsg.so.
 *
 * NOT part of the drop-in boundary (include/hmsg.h): nothing here replaces a reference interface.  The parity tests
 * use the hooks to pin building blocks of the path in isolation (the keep-largest DBSCAN, the stable radix sort, the
 * restated cKDTree, Python's float repr); bench.py uses hmsg_synth_render to put SURVEY 8d's synthetic stream into
 * HBM.  A host that links the library for the path itself includes hmsg.h only. */
#ifndef HMSG_TEST_H
#define HMSG_TEST_H
#include "hmsg.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Benchmark utility, not part of the path: render the synthetic posed RGB-D + mask stream of SURVEY 8d
 * straight into device buffers (rgb u8 [n][H][W][3], depth u16 [n][H][W], masks u8 [n][M][H][W]);
 * mask_entity (host, i32 [n][M]) tells which scene entity each mask shows. */
int hmsg_synth_render(int32_t device_id, int32_t n_frames, int32_t H, int32_t W, int32_t M, const double* K,
                      const double* poses, const int32_t* room_of_frame, int32_t n_rooms, const double* room_boxes,
                      int32_t n_obj, const double* obj_boxes, const int32_t* room_obj_off, double depth_noise_mm,
                      uint64_t seed, uint8_t* rgb_dev, uint16_t* depth_dev, uint8_t* masks_dev, int32_t* mask_entity_host);

/* test hook: the segmented keep-largest DBSCAN (pcd_denoise_dbscan, graph_utils.py:827-880) on K caller-supplied clouds
 * (sizes[k] points each, concatenated in pts; host pointers).  core0 (optional, one byte per point): anchor hint as the
 * merge fold passes it.  out_pts (capacity = all points), out_sizes [K], out_core (core flag per kept point), out_info
 * [K][3] = changed, clusters found, contested. */
int hmsg_test_dbscan(const double* pts, int32_t K, const int64_t* sizes, double eps, int32_t min_points, const uint8_t* core0,
                     double* out_pts, int64_t* out_sizes, uint8_t* out_core, int32_t* out_info);
/* test hook: Python-repr text of n doubles, newline separated, into out[cap]; returns bytes written or -1 */
int64_t hmsg_test_format_doubles(const double* v, int64_t n, char* out, int64_t cap);

/* ---- diagnostics (tests only): the stable (key, value) radix sort every order-faithful voxel mean is built on
 * (Open3D VoxelDownSample adds points in input order; graph.py:348, generic.py:188, graph.py:456).  Host arrays,
 * sorted in place by the low key_bits bits of the key, equal keys keep their input order. */
int hmsg_test_sort_pairs(uint32_t* keys, uint64_t* vals, int64_t n, int32_t key_bits);
/* out[i] = s[i] after `len[i]` sequential float64 additions of p[i] (how Open3D accumulates a map point that many
 * pixels of a mask snapped to, generic.py:181-188), computed by the closed form the mask kernels use for long
 * repetitions. */
int hmsg_test_repeat_add(const double* s, const double* p, const int32_t* len, double* out, int64_t n);
/* host restatement of scipy.spatial.cKDTree (the reference's NN index, graph.py:362-364; used to answer bit-equal
 * nearest-neighbour ties like scipy does): index permutation after the default build (out_indices i64 [n], may be
 * NULL), node count, and query(x, k=1) answers for nq points. */
int hmsg_test_ckdtree(const double* pts, int64_t n, const double* queries, int64_t nq, int64_t* out_idx,
                      int64_t* out_indices, int64_t* out_n_nodes);

/* the caching allocator's carving of a very large parked block (the frame store of a long episode handed back before the
 * merge: its block serves the merge's arenas instead of fresh hipMallocs): parks one block of `root_gb` GB, carves
 * three requests out of it, checks that they lie inside it and are disjoint, that nothing can be freed while a piece is
 * out, and that the block is whole again (same address) once every piece is back.  0 = ok, otherwise the failed check. */
int hmsg_test_allocator_carving(int32_t device_id, int32_t root_gb);

#ifdef __cplusplus
}
#endif
#endif /* HMSG_TEST_H */
