/*
 * hmsg.h -- C ABI of the MI355X-native HMSG (Hierarchical Multi-modal Scene Graph) build + retrieval
 * path.  This is the drop-in boundary for the hot path of HorizonRobotics/HoloAgent `fsr_vln`:
 * the reference has no FFI today (its boundary is the Python class `Graph`,
 * fsr_vln/memory/hmsg/graph/graph.py:77-219); each entry point below names the reference code it
 * replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every call returns 0 on success, <0 on error; hmsg_last_error(h) gives the message.
 *   - input pointers may be host OR device (HIP) pointers; the library detects which
 *     (hipPointerGetAttributes).  Output pointers are host pointers unless the name ends in `_dev`.
 *   - one handle = one scene = one HIP device + stream.  A handle is not thread-safe; distinct
 *     handles are independent.
 *   - all state (frames, voxel map, feature map, instances) lives in HBM and is owned by the handle.
 */
#ifndef HMSG_H
#define HMSG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hmsg_ctx hmsg_t;

enum {
    HMSG_OK = 0,
    HMSG_ERR_INVALID = -1,     /* bad argument / call order */
    HMSG_ERR_HIP = -2,         /* HIP runtime error */
    HMSG_ERR_UNSUPPORTED = -3, /* configuration outside what this build implements */
    HMSG_ERR_NOMEM = -4
};

enum { HMSG_MERGE_SEQUENTIAL = 0, HMSG_MERGE_HIERARCHICAL = 1 };

/* Hydra `pipeline.*` / `main.*` keys of fsr_vln/config/semantic_scene_reconstruction_hm3d.yaml:1-39
 * that the path reads, plus the constants the reference hard-codes in graph.py (named here so the
 * oracle tests can vary them; defaults = the reference's literals). */
typedef struct hmsg_config {
    int32_t device_id;            /* HIP device ordinal */
    int32_t feat_dim;             /* CLIP_DIM (utils/constants.py:3-7): 512 / 768 / 1024 */
    int32_t height, width;        /* depth image size */
    int32_t max_frames;           /* capacity of the resident frame store */
    int32_t max_masks;            /* masks per frame upper bound (<= 256; SAM at points_per_side=12 yields <= 144) */
    double voxel_size;            /* pipeline.voxel_size */
    double depth_scale;           /* dataset scale, 1000.0 (hm3dsem.py:40, horizon.py:38) */
    double init_overlap_thresh;   /* pipeline.init_overlap_thresh */
    double overlap_thresh_factor; /* pipeline.overlap_thresh_factor */
    double iou_thresh;            /* pipeline.iou_thresh */
    double clip_masked_weight;    /* pipeline.clip_masked_weight */
    double max_mask_distance;     /* pipeline.max_mask_distance (filter_distance of create_3d_masks) */
    int32_t merge_type;           /* pipeline.merge_type: HMSG_MERGE_* */
    int32_t outlier_nb_points;    /* graph.py:355 literal 1000 */
    double outlier_radius;        /* graph.py:356 literal 1.0 */
    double pool_max_dist;         /* graph.py:460 literal 0.8 */
    double feat_dbscan_eps;       /* graph.py:484 literal 0.01 */
    int32_t feat_dbscan_min;      /* graph.py:484 literal 100 */
    double merge_dbscan_eps;      /* graph_utils.py:678 literal 0.1 */
    int32_t merge_dbscan_min;     /* graph_utils.py:678 literal 10 */
    int32_t min_instance_points;  /* graph.py:447 literal 10 */
    int32_t skip_frames;          /* pipeline.skip_frames (config/semantic_scene_reconstruction_hm3d.yaml; graph.py:339,373
                                     `range(0, len(dataset), skip_frames)`): of the frames offered to hmsg_add_frames -- counted
                                     across calls -- every skip_frames-th is kept, the others are ignored; 1 = all */
    double depth_cut;             /* dataset depth_cut in metres (dataloader/horizon.py:258-261: depth > depth_cut * scale -> 0,
                                     applied as the frames come in); 0 = none */
    double grid_resolution;       /* pipeline.grid_resolution of the room level (graph.py:942-974); 0.05 in the shipped configs */
    int32_t overlap_distance_form; /* HMSG_OVERLAP_*: which of faiss's two evaluations of the squared distance
                                     find_overlapping_ratio_faiss (utils/graph_utils.py:645-662) stands for -- see below */
} hmsg_config;

/* find_overlapping_ratio_faiss asks faiss 1.7.2 IndexFlatL2.search(k = 1) for squared float32 distances and counts D < radius**2.
 * faiss evaluates (dx*dx + dy*dy) + dz*dz per pair for fewer than 20 queries and |x|^2 + |y|^2 - 2 x.y (sgemm), clamped at 0, from
 * 20 queries on; the second form's rounding grows with |x|^2 (scenes far from the origin), and a point on the radius can land on
 * either side.  HMSG_OVERLAP_DIRECT (default): the direct form for every cloud -- what the oracle and rounds 1-4 pin.
 * HMSG_OVERLAP_FAISS_BLAS: faiss's switch -- a cloud of 20 or more points is looked up in the BLAS form, stated as
 * |p|^2 = (p0*p0 + p1*p1) + p2*p2, x.y = fma(x2, y2, fma(x1, y1, x0*y0)), dis = (|x|^2 + |y|^2) - 2*(x.y) in float32 (the order inside
 * the reference's BLAS is not ours to know); the overlap grids then reach sqrt(radius^2 + E), E the form's error bound at the
 * scene's coordinates, so no witness the form would accept is missed.  faiss itself is absent from this image:
 * oracle/hmsg_oracle.py carries the same switch, tests/test_faiss_form_switch.py compares the two. */
enum { HMSG_OVERLAP_DIRECT = 0, HMSG_OVERLAP_FAISS_BLAS = 1 };

/* Fill `cfg` with the reference defaults (hm3d yaml + graph.py literals). */
void hmsg_default_config(hmsg_config* cfg);
/* sizeof(hmsg_config) of THIS library: a binding that declares the struct itself (ctypes, cgo, JNA ...) checks its own size against it
 * before the first hmsg_default_config writes through its pointer */
size_t hmsg_config_size(void);

/* Graph.__init__ (graph.py:81-219) for the build pipeline: allocate the HBM-resident scene state. */
int hmsg_create(const hmsg_config* cfg, hmsg_t** out);
void hmsg_destroy(hmsg_t* h);
const char* hmsg_last_error(const hmsg_t* h);
const char* hmsg_version(void);

/* The library parks freed device scratch in a per-thread cache (re-allocating GBs per scene costs hundreds of
 * ms); this returns the calling thread's cached blocks to the driver. */
void hmsg_release_cached_memory(void);

/* Start a new scene on the same handle: state is cleared, the HBM allocations are kept (what a service
 * that rebuilds maps repeatedly does; the reference constructs a new Graph per scene,
 * application/semantic_scene_reconstrucion_offline/offline_mapping_create_hmsg_hm3d_benchmark.py:70-110). */
int hmsg_reset(hmsg_t* h);

/* Live kernel timing with HIP events on the handle's stream (measurement aid for bench.py): when on, the
 * heavy kernels are bracketed by event pairs; hmsg_profile_entry aggregates them per kernel name.  on = 1: the wide kernels and, in
 * the merge fold, the overlap scans and the component pass; on = 2: every phase of the fold's DBSCAN batch as well (a bracket costs
 * that chain of dependent launches ~3 us each: 35 us per fold step with all of them). */
int hmsg_set_profiling(hmsg_t* h, int32_t on);
int32_t hmsg_profile_count(hmsg_t* h);   /* distinct kernel names recorded since the last reset */
int hmsg_profile_entry(hmsg_t* h, int32_t i, char* name /*[64]*/, int64_t* launches, double* total_ms,
                       double* total_work /* algorithmic bytes, or FLOP for the MFMA kernels; may be NULL */);

/* ---- loop A of create_feature_map (graph.py:339-345): hand over posed RGB-D frames ------------
 * rgb u8 [n][H][W][3], depth u16 [n][H][W] (millimetres), pose f64 [n][16] row-major camera-to-world,
 * K f64 [9] row-major intrinsics (dataset[i] tuple contract, horizon.py:217-268). Frames are copied
 * into the handle's resident frame store; frame ids are assigned consecutively from 0. */
int hmsg_add_frames(hmsg_t* h, int32_t n, const uint8_t* rgb, const uint16_t* depth, const double* pose,
                    const double* K);

/* ---- A1+A2: create_pcd over all frames (generic.py:74-138) + voxel_down_sample + the (no-op)
 * DBSCAN + remove_radius_outlier + the NN index that replaces cKDTree (graph.py:348-364). */
int hmsg_finalize_map(hmsg_t* h);
int64_t hmsg_map_size(const hmsg_t* h);              /* V = points of the filtered global cloud */
/* nearest-neighbour queries so far (graph.py:409, :458, generic.py:181) that were bit-equal distance ties and were
 * answered by the host restatement of scipy's cKDTree traversal (statistics) */
int64_t hmsg_num_tie_queries(const hmsg_t* h);
int64_t hmsg_map_size_unfiltered(const hmsg_t* h);   /* voxels before remove_radius_outlier */
int hmsg_get_map_points(const hmsg_t* h, double* xyz /*[V][3]*/, double* rgb /*[V][3] or NULL*/);

/* ---- loop B (graph.py:373-411): per frame the encoder outputs consumed by
 * extract_feats_per_pixel's fusion math (sam_clip_feats_extractor.py:159-191):
 * masks u8 [n][M][H][W] (0/1), F_g f32 [n][D], F_masked f32 [n][M][D], F_crop f32 [n][M][D].
 * M is the row stride of this hand-over (M <= cfg.max_masks; it may differ from call to call); n_masks i32 [n]
 * (host or device, NULL = M for every frame) is the number of masks SAM actually returned for each frame:
 * rows >= n_masks[f] are padding and are ignored -- the softmax of sam_clip_feats_extractor.py:167-169 runs over
 * the frame's own masks only.  A frame with 0 masks makes the reference raise (it unpacks a 3-tuple,
 * :163-164); here such a frame contributes zero features (its touched voxels still count the frame, graph.py:411)
 * and no 3-D mask.  Frames first_frame .. first_frame+n-1 must have been added. */
int hmsg_add_frame_features(hmsg_t* h, int32_t first_frame, int32_t n, int32_t M, const uint8_t* masks,
                            const float* F_g, const float* F_masked, const float* F_crop, const int32_t* n_masks);
/* masks SAM returned for a frame (= number of 3-D masks the frame contributes) */
int32_t hmsg_get_frame_num_masks(const hmsg_t* h, int32_t frame);

/* A3+A4+A5 for every frame handed over so far: per-pixel feature fusion, NN snapping of frame and
 * mask points, last-writer-wins accumulation into the voxel feature map, 3-D mask clouds
 * (graph.py:380-415, generic.py:140-190). */
int hmsg_fuse_frames(hmsg_t* h);
int hmsg_get_map_feats(const hmsg_t* h, float* feats /*[V][D]*/, float* counter /*[V] or NULL*/);
/* The per-voxel feature sums behind hmsg_get_map_feats (graph.py:410-411): sum f32 [V][D], counter u32 [V] (host or
 * device pointers; either may be NULL on get).  Handles that fused disjoint frame windows of ONE episode (same map)
 * all-reduce them and install the result; feats = sum / counter is recomputed (graph.py:413-415).  Frame
 * contributions add commutatively, so the reduced map equals the single-handle one up to float32 summation order. */
int hmsg_get_feature_sums(const hmsg_t* h, float* sum, uint32_t* counter);
int hmsg_set_feature_sums(hmsg_t* h, const float* sum, const uint32_t* counter);
/* test/introspection: NN index of every pixel of a frame (-1 where depth == 0), i32 [H][W].  HMSG_ERR_INVALID after
 * hmsg_merge_instances on an episode whose frame store exceeded 96 GB (the merge hands that store back). */
int hmsg_get_frame_nn(const hmsg_t* h, int32_t frame, int32_t* idx);
/* test/introspection: F_p of a frame (sam_clip_feats_extractor.py:172-175), f32 [n_masks(frame)][D] */
int hmsg_get_frame_fp(const hmsg_t* h, int32_t frame, float* f_p);
/* 3-D masks of a frame (create_3d_masks): sizes i64 [n_masks(frame)] then points f64 [sum][3] */
int hmsg_get_frame_mask_sizes(const hmsg_t* h, int32_t frame, int64_t* sizes);
int hmsg_get_frame_mask_points(const hmsg_t* h, int32_t frame, double* xyz);

/* ---- A6: seq_merge / hierarchical_merge (graph_utils.py:918-1038) + small-cloud drop
 * (graph.py:445-448).
 * Threads: a handle is driven by one host thread at a time; distinct handles may be driven from distinct threads.  With
 * merge_type sequential the library itself starts ONE worker thread per handle inside hmsg_fuse_frames: seq_merge's fold
 * over the frames begins on the first frames' 3-D masks while the fusion is still producing the later ones (own HIP
 * stream, nothing shared with the caller); hmsg_merge_instances waits for it, hmsg_reset / hmsg_destroy stop it.  The
 * instances are the same either way (environment HMSG_FOLD_NOPIPE=1: no worker, the fold runs inside this call). */
int hmsg_merge_instances(hmsg_t* h);
int64_t hmsg_num_instances(const hmsg_t* h);
int hmsg_get_instance_sizes(const hmsg_t* h, int64_t* sizes /*[N]*/);
int hmsg_get_instance_points(const hmsg_t* h, double* xyz /*[sum][3], host or device*/);
int hmsg_get_instance_boxes(const hmsg_t* h, double* boxes /*[N][6]: AABB min xyz, max xyz*/);

/* ---- hierarchical_merge (graph_utils.py:989-1012) sharded over the frames of ONE episode: every handle
 * holds the whole map (all frames' geometry) but the features / masks of its own frame range only.
 *   hmsg_set_frame_window  after hmsg_finalize_map, before any feature: this handle's frames start at first_frame
 *                          (frame indices stay global; hmsg_add_frame_features continues from first_frame).
 *   hmsg_merge_tree_local  the levels of the merge tree that lie inside the window, with the thresholds the global tree
 *                          over total_frames frames has there.  first_frame must be a multiple of a power of two >= the
 *                          window length.  Leaves the unfinished list as the handle's instances (hmsg_get_instance_*),
 *                          reports the threshold of the next level, the number of lists at that level and this list's
 *                          index among them.
 *   hmsg_merge_tree_join   one cross-handle level on the handle that holds the even-indexed list: merge_3d_masks over
 *                          [mine ++ the partner's clouds] (sizes i64 [n_ext] on the host, points f64 [sum][3] on the host
 *                          or on the device -- e.g. the receive buffer of an RCCL collective) at
 *                          threshold th; final_pass != 0 also runs the last pass (threshold 0.75, :1007-1011) and the
 *                          small-cloud drop (graph.py:445-448), after which hmsg_pool_instances may follow.
 *                          n_ext == 0 with final_pass: a single handle held every frame.
 * The result is bit-identical to hmsg_merge_instances over all frames (tests/test_distributed_gloo.py). */
int hmsg_set_frame_window(hmsg_t* h, int32_t first_frame);
int hmsg_merge_tree_local(hmsg_t* h, int32_t total_frames, double* th_next, int64_t* lists_now, int64_t* my_index);
int hmsg_merge_tree_join(hmsg_t* h, int32_t n_ext, const int64_t* ext_sizes, const double* ext_points, double th,
                         int32_t final_pass);

/* ---- A7: per-instance feature pooling (graph.py:450-491, graph_utils.py:682-728). */
int hmsg_pool_instances(hmsg_t* h);
int hmsg_get_instance_feats(const hmsg_t* h, float* feats /*[N][D]*/);

/* ---- A8: the storeys of the map -- Graph.segment_floors_manually (graph.py:624-787): the map re-sampled at 5 cm, the
 * height histogram at 1 cm (device), gaussian_filter1d(sigma = 2) on the int64 counts, find_peaks(distance = 20 bins,
 * height = 90th percentile), the 1-D DBSCAN(eps = 1 m) chaining of the peaks and the reference's pick / adjust rules
 * (numpy / scipy restated in C++ on the host: a few hundred bins), then per storey the crop of the full map by
 * y in [y_lo, y_hi] (both inclusive): point count, axis-aligned box, zero_level = lowest y in the crop, height = y_hi -
 * zero_level (:769-787).  [y_lo, y_hi] is the slab hmsg_segment_rooms / hmsg_room_clouds take.  capacity 0 asks for
 * the count only. */
typedef struct hmsg_floor {
    double y_lo, y_hi;
    double zero_level, height;
    double bbox_min[3], bbox_max[3];
    int64_t n_points;
} hmsg_floor;
int hmsg_segment_floors(hmsg_t* h, hmsg_floor* out, int32_t capacity, int32_t* n_floors);

/* ---- A8 helper: Open3D `PointCloud.voxel_down_sample(voxel_size)` of a caller-supplied cloud, output in ascending
 * (ix, iy, iz) voxel order.  Replaces `self.full_pcd.voxel_down_sample(voxel_size=0.05)` at the top of
 * segment_floors / segment_floors_manually (graph.py:496-497, 633), whose height histogram is taken over the
 * re-sampled cloud.  `points` host or device [n][3]; `out_points` host, capacity n points; *out_n = points written. */
int hmsg_voxel_down_sample(hmsg_t* h, const double* points, int64_t n, double voxel_size, double* out_points,
                           int64_t* out_n);

/* ---- A10 first step: every instance through pcd_denoise_dbscan(eps=0.05, min_points=10)
 * (graph.py:1589-1591), in place; call after hmsg_pool_instances like the reference does. */
int hmsg_denoise_instances(hmsg_t* h, double eps, int32_t min_points);

/* A10 room association (graph.py:1634-1642, find_intersection_share graph_utils.py:160-189): for every
 * instance i and room r, share[i][r] = #{room-r vertices (x,z) with an instance point within `radius` in the
 * x/z plane} / #instance points.  verts_xz f64 [sum][2] (host), vert_off i64 [n_rooms+1], share f64 [N][n_rooms]. */
int hmsg_instance_room_share(hmsg_t* h, int32_t n_rooms, const int64_t* vert_off, const double* verts_xz, double radius,
                             double* share);

/* ---- A10 + node table: segment_hmsg_objects (graph.py:1582-1736) without the per-view visibility test, which needs the
 * dataset's images and stays with the caller.  Runs the per-object pcd_denoise_dbscan(0.05, 10) if it has not run yet,
 * assigns every instance with at least 10 points to each floor whose [zero - 0.2, zero + height + 0.2] contains its
 * y-extent and, there, to the room with the largest find_intersection_share (fallback: nearest room centre), labels
 * it with the arg-max of emb . label_feats^T (identify_object, graph.py:1441-1454) and numbers it per room.  The scores are
 * float32 inputs multiplied and summed in FLOAT64 (exact products, one rounding per sum); the reference's np.dot stays in
 * float32 with BLAS's summation order, so two classes within ~1e-7 of each other could swap -- none does on the
 * reference-made fixture (every object name is compared, tests/test_objects_golden.py).  Nodes come out in the
 * reference's creation order (floors outer, instances inner).  floor_zero / floor_height f64 [n_floors];
 * room_floor i32 [n_rooms]; vert_off i64 [n_rooms + 1]; verts_xz f64 [sum][2]; label_feats f32 [n_labels][D] or NULL. */
typedef struct hmsg_node {
    int32_t instance;      /* index into the instance list (hmsg_get_instance_*) */
    int32_t floor, room;   /* room = index into the caller's room list: object id "<room_id>_<counter>" */
    int32_t counter;
    int32_t label;         /* -1 without a vocabulary */
    int64_t n_points;
} hmsg_node;
int hmsg_build_object_nodes(hmsg_t* h, int32_t n_floors, const double* floor_zero, const double* floor_height, int32_t n_rooms,
                            const int32_t* room_floor, const int64_t* vert_off, const double* verts_xz, int32_t n_labels,
                            const float* label_feats);
int64_t hmsg_num_nodes(const hmsg_t* h);
/* The view <-> object topology of segment_hmsg_objects (graph.py:1712-1734): check_object_in_view
 * (utils/graph_utils.py:95-157) for n_pairs (instance, view) pairs on the instance clouds as they are in the handle (after
 * hmsg_build_object_nodes: denoised, what the reference's mask_pcds hold at that point).  pose_inv f64 [n_views][4][4] =
 * np.linalg.inv(pose) row-major (world -> camera, computed by the caller as the reference does); wh i32 [n_views][2] = image
 * width, height; K f64 [3][3]; pair_inst / pair_view i32 [n_pairs].  visible u8 [n_pairs]: 1 when at least
 * min_visible_ratio (0.5) of ALL the cloud's points are in front of the camera and project inside the image and their
 * mean depth is <= max_depth (10.0); mean_depth f64 [n_pairs]: that mean (inf where the reference returns inf: the caller
 * picks best_view_id = the visible view of smallest mean depth, first on ties).  The two matrix products are evaluated as
 * BLAS dgemm does (one fused multiply-add chain per element, k ascending): a point seen at the image border of a frame
 * re-projects onto that border to ~1e-14, so the inside / outside decisions follow the reference only with its
 * arithmetic.  The mean is summed in a fixed device order (numpy: pairwise), equal to ~1e-16 relative. */
int hmsg_object_views(hmsg_t* h, int32_t n_views, const double* pose_inv, const int32_t* wh, const double* K, int64_t n_pairs,
                      const int32_t* pair_inst, const int32_t* pair_view, double min_visible_ratio, double max_depth,
                      uint8_t* visible, double* mean_depth);

/* ---- A9: the room clouds of segment_hmsg_room (graph.py:1086-1108).  For every room the (x, z) cell centres of its 2-D
 * region (room_xz f64 [sum][2], CSR room_off i64 [n_rooms + 1]: what map_grid_to_point_cloud returns, :1084) are
 * extruded over z_levels (np.arange(zero, zero + height, 0.05) * -1, :1088-1092), transformed by T (row-major 4x4, the
 * reference's T1: Rotation.from_euler("x", 90) as scipy gives it) and every extruded point picks its nearest neighbour
 * in the FLOOR cloud = the map points with y in [y_lo, y_hi] (full_pcd.crop, :769-775; cKDTree.query(k = 1) in the
 * reference, bit-equal ties answered by the restated cKDTree over that cloud).  A room's cloud is
 * floor_pcd.select_by_index(idx): out_index (i32, capacity out_capacity) receives, room after room, the ascending indices
 * INTO THE FLOOR CLOUD (= rank among the map points inside the slab, map order); out_sizes [n_rooms] their counts;
 * n_floor_points the size of the floor cloud.  out_index may be NULL to ask for the sizes only. */
int hmsg_room_clouds(hmsg_t* h, double y_lo, double y_hi, const double* T, int32_t n_levels, const double* z_levels,
                     int32_t n_rooms, const int64_t* room_off, const double* room_xz, int64_t* out_sizes,
                     int32_t* out_index, int64_t out_capacity, int64_t* n_floor_points);
/* The camera -> room distance table of compute_room_embeddings (utils/graph_utils.py:244-291: `np.min(cdist([pos],
 * room_points))` with pos = the camera's (x, z)) from the room clouds the last hmsg_room_clouds call left on the device -- the
 * clouds do not travel to the host and back.  q_xz f64 [n_q][2] (host), out f64 [n_q][n_rooms] (host); n_rooms is what the
 * caller sized `out` for and must be the room count of that hmsg_room_clouds call (HMSG_ERR_INVALID otherwise: a call that
 * found no floor point or failed leaves n_rooms empty rooms / no rooms, never the previous storey's).  The distance to an
 * empty room is +inf. */
int hmsg_room_camera_distances(hmsg_t* h, int32_t n_rooms, int64_t n_q, const double* q_xz, double* out);
/* ---- N1: the rooms of one storey as a label image -- segment_hmsg_room up to room_vertices (graph.py:942-1071) and
 * distance_transform (graph_utils.py:391-487), every image step a kernel (the reference: numpy + OpenCV on the host).
 * The storey's cloud is the map points with y in [y_lo, y_hi] (as hmsg_room_clouds); zero_level / height are the
 * floor's, resolution = pipeline.grid_resolution.  out_markers i32 [rows][cols] (capacity in elements; host or device
 * memory): room i is markers == i + 1 for i < n_rooms, n_rooms + 1 the outside, -1 watershed lines and the image
 * border.  map_grid_to_point_cloud (graph_utils.py:359-388) of a cell (row, col): x = (col - 10.5) * resolution +
 * xz_min[0], z = (row - 10.5) * resolution + xz_min[1].  out_markers NULL asks for rows / cols / xz_min only.
 * OpenCV is restated from its documented semantics (oracle/rooms_oracle.py says where that is not pixel-exact):
 * parity with the reference is statistical here, unlike the rest of the path. */
int hmsg_segment_rooms(hmsg_t* h, double y_lo, double y_hi, double zero_level, double height, double resolution,
                       int32_t* out_markers, int64_t capacity, int32_t* out_rows, int32_t* out_cols, int32_t* out_n_rooms,
                       double* out_xz_min);
/* nodes [hmsg_num_nodes] and/or their embeddings f32 [N][D] (either may be NULL) */
int hmsg_get_nodes(const hmsg_t* h, hmsg_node* nodes, float* embeddings);

/* ---- N2: the object level of the on-disk format written at speed.  Replaces Object.save (memory/hmsg/graph/object.py:
 * 37-57) as driven per node by save_hmsg_graph (graph.py:1801-1824): for every record <dir>/<file_stem>.ply (binary
 * little-endian double x y z of the instance cloud under the header Open3D 0.18 writes for a cloud without colours, its
 * "comment Created by Open3D" line included; the reference's objects also carry colours, which this path does not keep) and
 * <dir>/<file_stem>.json = json.dump of {"object_id", "vertices"
 * (= points[:, [0, 2]], graph.py:1715), "room_id", "name", "embedding" (pooled feature), "view_ids", "best_view_id"} in
 * that order, numbers printed like Python's float repr.  The *_json fields are JSON text supplied by the caller (a
 * quoted string, a list, null ...) and are copied verbatim.  Clouds and features are read back from HBM once;
 * n_threads host threads format and write (<= 0: one per core, at most 32). */
typedef struct {
    int32_t instance;               /* index into the instance list (hmsg_node.instance) */
    const char* file_stem;          /* str(object_id) */
    const char* object_id_json;
    const char* room_id_json;
    const char* name_json;
    const char* view_ids_json;
    const char* best_view_id_json;
} hmsg_object_record;
int hmsg_save_objects(hmsg_t* h, const char* dir, int64_t n, const hmsg_object_record* recs, int32_t n_threads);
/* N2: Room.merge_objects (fsr_vln/memory/hmsg/graph/room.py:62-129; the optional post-pass of build_hier_multimodal_scene_graph,
 * graph.py:2053-2058, `pipeline.merge_objects_graph`) for the n objects of ONE room, in room.objects order: cloud i =
 * points[off[i] .. off[i + 1]) (f64 [.][3], host or device memory), name_id[i] = any numbering of the names (equal ids = equal
 * names).  Every same-name pair i < j is tested on the device (find_overlapping_ratio_faiss(obj_i.pcd, obj_j.pcd, radius) >
 * overlap_threshold; the reference calls it with 0.01 / 0.1), then the reference's bookkeeping is followed step by step:
 * np.where order, the dictionary chaining (an object that was absorbed can still become a key), the untouched objects behind them,
 * list(set(j)) in CPython's order.  Out: n_groups groups in the order the new room.objects list has; group g = group_members[
 * group_off[g] .. group_off[g + 1]): the object that stays (and is re-numbered room_id + "_" + g) first, then the objects the
 * reference adds to it (Object.__add__, object.py:93-103), in that order.  group_off: n + 1 entries, group_members: members_capacity
 * entries (n (n + 1) always suffice: every object can become a key, a key's list holds an object once). */
int hmsg_merge_room_objects(hmsg_t* h, int32_t n, const double* points, const int64_t* off, const int32_t* name_id,
                            double overlap_threshold, double radius, int32_t* n_groups, int32_t* group_off, int32_t* group_members,
                            int32_t members_capacity);
/* N2, load side: the object table of a saved graph straight into a retrieval index.  For every stem <dir>/<stem>.json is
 * read and its "embedding" array parsed as float64 (object.py:75-91 load; graph.py:1892-1987 load_hmsg_graph), rows in
 * the order given; room_of_node as for hmsg_index_create (declared below).  feat_dim (optional) receives the row length.
 * HMSG_ERR_INVALID when a record is missing, has no numeric embedding (saved as "") or the lengths differ.  (Object.load
 * accepts a record saved without an embedding and leaves embedding = None; such a node cannot be scored, and skipping it
 * would shift every node index after it, so the index refuses the whole table instead.) */
struct hmsg_index;
int hmsg_index_load_objects(int32_t device_id, const char* dir, int64_t n, const char* const* stems,
                            const int32_t* room_of_node, int32_t n_threads, struct hmsg_index** out, int32_t* feat_dim);

/* ---- A9 camera -> room assignment of compute_room_embeddings (utils/graph_utils.py:244-291): out[q][s] =
 * np.min(cdist([q], set s, "euclidean")) for n_q 2-D positions (camera x/z) against n_sets 2-D point sets (room
 * clouds projected to x/z): pts_xy f64 [set_off[n_sets]][2], q_xy f64 [n_q][2], out f64 [n_q][n_sets] (inf for an empty
 * set).  Host pointers. */
int hmsg_points_min_dist_2d(int32_t device_id, int32_t n_sets, const int64_t* set_off, const double* pts_xy, int64_t n_q,
                            const double* q_xy, double* out);

/* ---- N3 (the step right before the path): LiDAR keyframe clouds -> occlusion-aware uint16 depth images.  Replaces
 * nav_agent/humble_localization_nav2/lio_mapping_loc/scripts/generate_depth.py process_frame (:612-659) per frame:
 * voxel_down_sample (:626-629), project_points (:366-396), whether_occluded_deoccfast (:125-205: z-buffer at rounded
 * pixels against a float32 buffer in point order, fb / z, cv2.dilate(rect max(1, 4 / image_scale), iterations 4),
 * np.int16, cv2.filterSpeckles(0, 1000, 1), |fb / z - disparity| >= 3) and generate_occ_depth (:399-474: last point
 * in input order wins its truncated pixel, uint16(float32(z * depth_factor))).  Host pointers.
 *   points     f64 [cloud_off[n_frames]][3]: frame f owns rows cloud_off[f] .. cloud_off[f+1]
 *   poses      f64 [n_frames][12]: world -> camera rotation (row-major 3x3) then translation; NULL: `points` are
 *              already (u, v, z) rows = generate_occ_depth's own inputs (points_image[0], points_image[1],
 *              points_camera[2]) and no projection / culling is done
 *   depth_out  u16 [n_frames][height][width]
 *   state_out  optional u8 per input point: 0 visible, 1 occluded, 2 culled by the projection; only without
 *              down-sampling (voxel_size <= 0), HMSG_ERR_INVALID otherwise
 *   stats_out  optional i64 [n_frames][4]: points after down-sampling, projected into the image, visible, depth
 *              pixels written
 *   device_ms  optional: HIP-event time of the device work (transfers excluded)
 * Down-sampled voxels are visited in ascending (ix, iy, iz) order (Open3D's order is its hash map's); a cloud whose
 * bounding box holds more than 2^37 voxels is HMSG_ERR_UNSUPPORTED. */
typedef struct {
    int32_t width, height, image_scale;
    double fx, fy, cx, cy;
    double voxel_size;      /* <= 0: no down-sampling */
    double depth_factor;    /* 1000 in the reference */
} hmsg_depth_params;
int hmsg_lidar_depth(int32_t device_id, const hmsg_depth_params* prm, int32_t n_frames, const double* points,
                     const int64_t* cloud_off, const double* poses, uint16_t* depth_out, uint8_t* state_out,
                     int64_t* stats_out, double* device_ms);

/* ---- N4 (encoder side of the path): crop_all_bounding_boxs (memory/hmsg/utils/sam_utils.py:119-147) for BOTH variants
 * the extractor asks for per frame (perception/models/sam_clip_feats_extractor.py:148-151) in one launch:
 *   out_plain[m]  = cv2.resize(crop_bbox(image, bbox[m], bbox_margin), (S, S))      (sam_utils.py:58-81, 167-183)
 *   out_masked[m] = cv2.resize(crop_image(image, mask m), (S, S))                    (:150-164, no margin)
 * image u8 [H][W][3]; segs u8 [M][H][W] (non-zero = inside; needed for out_masked); bbox f64 [M][4] XYWH (host);
 * out_* u8 [M][S][S][3] (either may be NULL); S = out_size, a multiple of 4 (512 in the reference).  image / segs /
 * out_* may be host or device pointers.  cv2.resize = INTER_LINEAR, OpenCV's 8-bit fixed-point arithmetic.  A mask whose
 * crop is empty is HMSG_ERR_INVALID (cv2.resize raises on it).  device_ms (optional): HIP-event time of the launch. */
int hmsg_crop_resize_batch(int32_t device_id, int32_t H, int32_t W, const uint8_t* image, int32_t M, const uint8_t* segs,
                           const double* bbox, double bbox_margin, int32_t out_size, uint8_t* out_plain, uint8_t* out_masked,
                           double* device_ms);

/* ---- A11: the other node records of save_hmsg_graph (graph.py:1801-1824) -- floors/<f>.{ply,json} (floor.py:37-52),
 * rooms/<f>_<r>.{ply,json} (room.py:309-337), views/<id>.json (view.py:56-74) -- from a C / C++ host, byte for byte what
 * json.dump writes: one JSON object per call, fields in the order given (the reference's key order).  A field is either RAW
 * (value_json: the caller's JSON text for strings, id lists, null ...) or numbers the library prints like Python does
 * (float.__repr__ of the double for F64 / F32, decimal integers for I64): a scalar (ndim 0), [n0] (ndim 1) or [n0][n1]
 * (ndim 2; n0 = 0 gives []).  hmsg_write_ply writes a cloud as Open3D's write_point_cloud does (x y z doubles);
 * hmsg_read_json_numbers reads the numbers of one key back, flattened, as json.load + np.array would see them. */
enum { HMSG_JSON_RAW = 0, HMSG_JSON_F64 = 1, HMSG_JSON_F32 = 2, HMSG_JSON_I64 = 3 };
typedef struct hmsg_json_field {
    const char* key;          /* written between quotes as it is */
    int32_t kind;             /* HMSG_JSON_* */
    int32_t ndim;             /* 0, 1, 2 (numbers only) */
    int64_t n0, n1;
    const void* data;         /* RAW: const char* JSON text;  numbers: host array */
} hmsg_json_field;
int hmsg_write_json(const char* path, int32_t n_fields, const hmsg_json_field* fields);
int hmsg_write_ply(const char* path, const double* xyz, int64_t n);
int hmsg_read_json_numbers(const char* path, const char* key, double* out, int64_t capacity, int64_t* n);

/* ---- A9 / A11 bookkeeping for a C / C++ host (host arrays in, host arrays out; no device work).
 *
 * hmsg_assign_cameras_to_rooms -- compute_room_embeddings' camera -> room step (utils/graph_utils.py:257-291): camera i, if its
 * height cam_height[i] lies inside [y_min, y_max] (the floor cloud's bounds), goes to the room with the smallest
 * dist[i * n_rooms + r] (hmsg_room_camera_distances / hmsg_min_dist_2d; first minimum, as np.argmin); a room that got no camera
 * takes the closest camera OUTSIDE the bounds (sic; camera 0 when there is none, as np.argmin over a table of inf).
 * room_of_cam[n_cams]: the room of every camera inside the bounds, -1 for the others.  The rooms' image lists come back as
 * a CSR table: room_off[n_rooms + 1], room_imgs[room_off[r] .. room_off[r + 1]) in ascending camera order (capacity
 * n_cams + n_rooms entries).
 *
 * hmsg_pick_representative_views -- what follows KMeans in compute_room_embeddings (:334-352): for every label that occurs
 * (ascending, as np.unique), the member whose embedding has the largest dot product with its cluster centre (first maximum).
 * The KMeans fit itself stays on the host side of the boundary: the reference calls scikit-learn's
 * KMeans(n_clusters = num_views, n_init = 5, max_iter = 100, random_state = 0) and so does holoagent_amd/graph.py; a C host brings
 * labels and centres from whatever KMeans it links.  embs [n x D] and centers [k x D] are float32; the products are accumulated
 * in float64.  (A cluster of two members is an exact tie -- both are equally far from their mean -- which the reference
 * decides by float32 rounding inside np.dot; there, and only there, the pick may be the other member.)
 * out_member[<= k]: indices into the room's image list, one per label present; *n_out their number.
 *
 * hmsg_graph_edges -- create_graph_new (graph.py:1752-1775) as an edge list.  Node ids: 0 = the building, then the floors,
 * the rooms (in the order of room_floor), the objects, the views.  Edges in the reference's insertion order: per floor
 * (0, floor); per room of the floor (floor, room); per object of the room, ascending (room, object); then per view
 * (room, view) if view_room[v] >= 0 (a freshly built graph has none: graph.py:1176-1189 compares a string id with an int),
 * and (view, object) for its objects view_obj[view_obj_off[v] .. view_obj_off[v + 1]) in ascending object order, each once.
 * edges: capacity pairs of int64; *n_edges is set even when the capacity is too small (HMSG_ERR_INVALID then). */
/* hmsg_kmeans -- the fit between the two calls above (utils/graph_utils.py:329-333:
 * `KMeans(n_clusters=num_views, max_iter=100, n_init=5, random_state=0).fit(room_clip_embeddings)`), for a host without
 * scikit-learn.  X f32 [n][dim] (n >= k); labels i32 [n], centers f32 [k][dim] = labels_ / cluster_centers_; inertia, n_iter
 * optional.  A restatement of scikit-learn 1.7.2's Lloyd KMeans with k-means++ seeding under numpy's RandomState(seed)
 * (holoagent_amd/csrc/hmsg_kmeans.hip says statement by statement what it follows, and which three BLAS / einsum sums are
 * accumulated in float64 here because their order is the library's own: a point whose two nearest centres tie to ~1e-7
 * relative may be labelled differently).  tests/test_kmeans_cabi.py holds it to scikit-learn itself. */
int hmsg_kmeans(const float* X, int64_t n, int32_t dim, int32_t k, int32_t n_init, int32_t max_iter, uint32_t seed,
                int32_t* labels, float* centers, float* inertia, int32_t* n_iter);
int hmsg_assign_cameras_to_rooms(const double* dist, int64_t n_cams, int32_t n_rooms, const double* cam_height, double y_min,
                                 double y_max, int32_t* room_of_cam, int64_t* room_off, int32_t* room_imgs);
int hmsg_pick_representative_views(const float* embs, int64_t n, int32_t dim, const int32_t* labels, const float* centers,
                                   int32_t k, int32_t* out_member, int32_t* n_out);
int hmsg_graph_edges(int32_t n_floors, int32_t n_rooms, const int32_t* room_floor, int32_t n_objects, const int32_t* obj_room,
                     int32_t n_views, const int32_t* view_room, const int64_t* view_obj_off, const int32_t* view_obj,
                     int64_t* edges, int64_t capacity, int64_t* n_edges);

/* ---- (b) the graph as ONE object behind the boundary: build_hier_multimodal_scene_graph (graph.py:2033-2076), save_hmsg_graph
 * (:1801-1824) and load_hmsg_graph (:1892-1987) without a line of host bookkeeping on the caller's side.  A C / C++ host builds,
 * saves, loads and queries with four calls (tests/host_c/hmsg_host.c); holoagent_amd/graph.py's Graph keeps the reference's method
 * names on top of the same object.
 *
 *   hmsg_build_graph   after hmsg_pool_instances: segment_floors_manually (:624-787) -> per storey segment_hmsg_room (:920-1189:
 *       regions by the device watershed, room clouds, camera -> room assignment, KMeans(num_views) representative views
 *       (hmsg_kmeans), Room and View nodes with the reference's ids: rooms "<floor>_<i>", views "<floor>_<i>_<k>" with k
 *       counting across the storey's rooms, View.room_id the per-floor room INDEX) -> segment_hmsg_objects (:1582-1736: objects
 *       "<room_id>_<counter>", names from the label vocabulary, view_ids / best_view_id by check_object_in_view on the device)
 *       -> create_graph_new (:1752-1775).
 *         poses       f64 [n_frames][16] camera -> world of the PROCESSED frames (frame i of this table = dataset image
 *                     i * skip_frames), row-major;
 *         poses_inv   their np.linalg.inv, or NULL: computed here by LU with partial pivoting (a LAPACK build that orders the
 *                     eliminations differently can differ in the last bit, and a point on an image border decides by it);
 *         view_feats  f32 [n_frames][D]: the frames' global CLIP features (the F_g of loop B; graph.py:1119-1130 recomputes them);
 *         img_paths   optional, [n_frames]: dataset.frameId2imgPath of those frames (View.img_path);
 *         label_feats f32 [n_labels][D] + label_names [n_labels] (get_label_feats), or 0 / NULL: every object is "object".
 *   hmsg_graph_begin / hmsg_graph_finish   the same in two halves, so that the room level runs BESIDE the fusion and the merge
 *       fold: begin right after hmsg_finalize_map (floors, regions, room clouds and the camera table on the device, then the
 *       KMeans fits on host threads), finish after hmsg_pool_instances (joins them; views, objects, edges).
 *   hmsg_save   the whole directory in the reference layout: <dir>/floors/<f>.{ply,json}, rooms/<f>_<r>.{ply,json},
 *       objects/<id>.{ply,json}, views/<id>.json -- byte for byte what the mirror's Floor / Room / Object / View .save() write
 *       (tests/test_scene_graph_cabi.py), which tests/golden/persist.json pins to the reference's own classes.
 *   hmsg_load   the same directory back: nodes in the loader's order (sorted file names: rooms and objects lexicographic),
 *       object embeddings as float64, Room - View edges by id.  The result is a graph without a scene handle.
 *   hmsg_graph_index / hmsg_graph_query   the retrieval index with the upper levels resident (floors -> rooms, view
 *       embeddings, room_key = int(room_id.split("_")[-1]); room_name_emb f64 [rooms][D]: the CLIP embeddings of the rooms' names,
 *       NULL = no label mode) and hmsg_query_hier on it (declared below; the index of hmsg_graph_query belongs to the graph).
 *   hmsg_graph_to_json   ids, names and lists of every node + the edge list as one JSON text (buf NULL: *needed only). */
typedef struct hmsg_graph hmsg_graph_t;
typedef struct hmsg_index hmsg_index_t;
typedef struct hmsg_graph_params {
    int32_t num_views;          /* 24: KMeans clusters per room (graph.py:1136) */
    int32_t kmeans_n_init;      /* 5   (utils/graph_utils.py:329-333) */
    int32_t kmeans_max_iter;    /* 100 */
    uint32_t kmeans_seed;       /* 0   (random_state) */
    int32_t skip_frames;        /* pipeline.skip_frames: image id of processed frame i = i * skip_frames; <= 0: the handle's */
    int32_t image_width, image_height;   /* size of the views' images (check_object_in_view); 0: the handle's */
    double min_visible_ratio;   /* 0.5  (utils/graph_utils.py:95-157) */
    double max_view_depth;      /* 10.0 */
    int32_t host_threads;       /* KMeans fits / object writers; 0: one per core, at most 16 */
    int32_t merge_objects_graph;   /* pipeline.merge_objects_graph (graph.py:2053-2058; false in every shipped config): after the objects, every
                                    * room fuses its same-name objects whose clouds overlap (Room.merge_objects, hmsg_merge_room_objects with
                                    * the reference's 0.01 / 0.1) and re-numbers them; the graph's object list is then the rooms' lists */
    int32_t reserved_;
} hmsg_graph_params;
typedef struct hmsg_graph_counts {
    int32_t floors, rooms, views, objects;
    int64_t edges, view_object_links;
    double begin_ms, finish_ms, kmeans_wait_ms;   /* wall time of the two halves; what finish waited for the KMeans threads */
} hmsg_graph_counts;
typedef struct hmsg_graph_object {
    char object_id[48];
    char name[80];
    int32_t room;               /* global room index (hmsg_graph_get_rooms order) */
    int32_t instance, label;    /* hmsg_node.instance / .label; -1 for a loaded graph */
    int32_t n_views, best_view; /* best_view: global view index, -1 none (loaded graphs: -1, the id is in hmsg_graph_to_json) */
} hmsg_graph_object;
typedef struct hmsg_graph_room {
    char room_id[32];
    char name[80];
    int32_t floor;
    int64_t n_vertices, n_points;
    int32_t n_embeddings, n_sample_images, n_objects, n_views;
} hmsg_graph_room;
/* Lifetime: a BUILT graph (hmsg_build_graph / hmsg_graph_begin) keeps a plain reference to its scene handle -- the object clouds and
 * features stay in the handle's HBM -- so hmsg_graph_finish, hmsg_save, hmsg_graph_index and hmsg_graph_allgather_index need the
 * handle alive and not reset: destroy or reset the graph's scene only after hmsg_graph_destroy (a LOADED graph has no handle). */
void hmsg_graph_default_params(hmsg_graph_params* p);
int hmsg_build_graph(hmsg_t* h, const hmsg_graph_params* prm, int32_t n_frames, const double* poses, const double* poses_inv,
                     const float* view_feats, const char* const* img_paths, int32_t n_labels, const float* label_feats,
                     const char* const* label_names, hmsg_graph_t** out);
int hmsg_graph_begin(hmsg_t* h, const hmsg_graph_params* prm, int32_t n_frames, const double* poses, const double* poses_inv,
                     const float* view_feats, const char* const* img_paths, hmsg_graph_t** out);
int hmsg_graph_finish(hmsg_graph_t* g, int32_t n_labels, const float* label_feats, const char* const* label_names);
void hmsg_graph_destroy(hmsg_graph_t* g);
const char* hmsg_graph_last_error(const hmsg_graph_t* g);
int hmsg_graph_get_counts(const hmsg_graph_t* g, hmsg_graph_counts* counts);
int hmsg_graph_get_edges(const hmsg_graph_t* g, int64_t* edges /*[capacity][2], may be NULL*/, int64_t capacity, int64_t* n_edges);
int hmsg_graph_get_objects(const hmsg_graph_t* g, hmsg_graph_object* out, int64_t capacity);
int hmsg_graph_get_rooms(const hmsg_graph_t* g, hmsg_graph_room* out, int64_t capacity);
int hmsg_graph_get_room_vertices(const hmsg_graph_t* g, int32_t room, double* xz /*[n_vertices][2]: room.vertices*/, int64_t capacity);
int hmsg_graph_get_room_embeddings(const hmsg_graph_t* g, int32_t room, float* emb /*[n_embeddings][D]*/, int64_t capacity);
int hmsg_graph_to_json(const hmsg_graph_t* g, char* buf, int64_t capacity, int64_t* needed);
int hmsg_save(hmsg_graph_t* g, const char* dir);
int hmsg_load(const char* dir, int32_t device_id, hmsg_graph_t** out);
int hmsg_graph_index(hmsg_graph_t* g, const double* room_name_emb, hmsg_index_t** out);
int hmsg_graph_query(hmsg_graph_t* g, const double* room_name_emb, int32_t Q, int32_t C, const float* T_obj, const int32_t* qid,
                     const float* T_room, const int32_t* floor_id, const int32_t* room_mode, int32_t k, int32_t use_negatives,
                     int32_t max_rooms, int32_t* out_sel, int32_t* out_nsel, int32_t* out_idx, int32_t* out_room, double* out_score);

/* ---- A12: retrieval over a node table (graph.py:3056-3162 query_hmsg_object and the GEMV of
 * query_hmsg_room / query_floor).  A table is N node embeddings (f64, as after load_hmsg_graph:
 * object.py:88-89, or f32 right after build) with a parent (room) id per node. */
int hmsg_index_create(int32_t device_id, int32_t dim, int64_t n, const void* emb, int32_t emb_is_f64,
                      const int32_t* room_of_node, hmsg_index_t** out);
/* the node table of a built scene as a resident index (embeddings gathered on the device, parent = node.room) */
int hmsg_index_from_nodes(hmsg_t* h, hmsg_index_t** out);
void hmsg_index_destroy(hmsg_index_t* ix);
const char* hmsg_index_last_error(const hmsg_index_t* ix);
/* live HIP-event timing of the similarity GEMM on the index's stream (measurement aid for bench.py) */
int hmsg_index_set_profiling(hmsg_index_t* ix, int32_t on);
int hmsg_index_profile(hmsg_index_t* ix, int64_t* launches, double* total_ms, double* total_flop);
/* Q queries; query q has C text rows T[q][C][D] f32 (row `qid[q]` is the query itself, the others
 * the negative prompts), searches the nodes whose room id is listed in rooms[room_off[q] ..
 * room_off[q+1]) IN THAT ORDER (candidate order = room order then node order, graph.py:3099-3110),
 * returns up to k node indices (-1 padded), their room ids and float64 scores sim[qid][node]. */
int hmsg_query_objects(hmsg_index_t* ix, int32_t Q, int32_t C, const float* T, const int32_t* qid,
                       const int32_t* room_off, const int32_t* rooms, int32_t k, int32_t use_negatives,
                       int32_t* out_idx, int32_t* out_room, double* out_score);
/* ---- the coarse-to-fine query behind the drivers query_hierarchy_protected{,_icra} (graph.py:3483-3716): floor ->
 * room(s) -> objects, batched, every stage on the device.
 *   hmsg_index_set_hierarchy  makes the levels above the nodes resident: n_rooms rooms (rooms without objects count), the rooms of every floor in floors[f].rooms
 *        order (CSR floor_room_off [n_floors + 1] / floor_rooms; room ids as in room_of_node, where self.rooms order =
 *        ascending id), per room the CLIP embedding of its NAME (room_name_emb f64 [R][D]; NULL: no label mode), its
 *        view embeddings room.embeddings (CSR view_off i64 [R + 1] / view_emb f64 [NV][D]) and room_key[r] =
 *        int(room_id.split("_")[-1]), the number the view mode reports (graph.py:3254-3262).
 *   hmsg_query_hier  per query: floor_id (-1 = all rooms; resolve query_floor :2216-2257 on the host or with
 *        hmsg_similarity), the room text row T_room[q][D] and room_mode[q]:
 *          0  no room stage: every room of the floor's list;
 *          1  query_hmsg_room(..., "label") (:3204-3232): rooms whose name similarity is within 1e-3 of the best;
 *          2 / 3  the view-embedding branch (:3247-3272): the 5 (valid room text) / 10 best rooms by their best view;
 *        then query_hmsg_object (:3056-3162) over those rooms IN THAT ORDER with the C text rows T_obj[q] (row qid[q] the
 *        query, the others negative prompts).  out_sel [Q][max_rooms] / out_nsel [Q]: the numbers query_hmsg_room
 *        returns (positions in the floor's room list; in view mode the room keys, which the reference then uses AS
 *        positions); out_idx / out_room / out_score [Q][k] as hmsg_query_objects (room = global room id).
 *        HMSG_ERR_INVALID where the reference raises (a room without view embeddings in view mode; a key that is no
 *        position of the list). */
int hmsg_index_set_hierarchy(hmsg_index_t* ix, int32_t n_rooms, int32_t n_floors, const int32_t* floor_room_off, const int32_t* floor_rooms,
                             const double* room_name_emb, const int64_t* view_off, const double* view_emb, const int32_t* room_key);
int hmsg_query_hier(hmsg_index_t* ix, int32_t Q, int32_t C, const float* T_obj, const int32_t* qid, const float* T_room,
                    const int32_t* floor_id, const int32_t* room_mode, int32_t k, int32_t use_negatives, int32_t max_rooms,
                    int32_t* out_sel, int32_t* out_nsel, int32_t* out_idx, int32_t* out_room, double* out_score);
/* plain similarity S[Q][N] = T[Q][D] . E[N][D]^T in float64 (query_floor / query_hmsg_room GEMV) */
int hmsg_similarity(hmsg_index_t* ix, int32_t Q, const float* T, double* S);

/* ---- multi-GPU exchange steps: RCCL collectives on device buffers, issued by the host that drives the handle.
 * The reference has no multi-GPU path: its benchmark driver loops scenes one after the other
 * (application/.../offline_mapping_create_hmsg_hm3d_benchmark.py:70-110) -- one process per GPU builds its own scene with no
 * collective, and these calls are the only exchange steps.  librccl is loaded on first use (a one-GPU process never needs
 * it).  One communicator per process / GPU:
 *   hmsg_comm_unique_id   rank 0 makes the 128-byte id (ncclGetUniqueId); the host passes it to the other ranks by its own
 *                         means (MPI, a TCP store, torch.distributed's store ...);
 *   hmsg_comm_create      ncclCommInitRank on device_id.  id == NULL with world == 1: no communicator, every collective is
 *                         the identity (single-GPU runs take the same code path). */
#define HMSG_COMM_ID_BYTES 128
typedef struct hmsg_comm hmsg_comm_t;
int hmsg_comm_unique_id(uint8_t* out_id /* [HMSG_COMM_ID_BYTES] */);
int hmsg_comm_create(const uint8_t* id, int32_t rank, int32_t world, int32_t device_id, hmsg_comm_t** out);
void hmsg_comm_destroy(hmsg_comm_t* c);
const char* hmsg_comm_last_error(const hmsg_comm_t* c);
/* Cross-scene retrieval (configs[3]): all-gather of the ranks' node tables -- embeddings of the object nodes gathered on
 * the device (f32 [n][D]) and the parent room of every node -- into ONE resident index on every rank (counts first, then
 * the payload padded to the largest table, HBM to HBM).  Global node index = node_off[rank] + local index, global room id =
 * room_off[rank] + local room id (n_rooms_local = rooms of this rank's graph, rooms without objects included); the index
 * answers like hmsg_index_create over the concatenated tables.  node_off / room_off: [world + 1], optional. */
int hmsg_allgather_nodes(hmsg_t* h, hmsg_comm_t* c, int32_t n_rooms_local, hmsg_index_t** out_index, int64_t* node_off, int64_t* room_off);
/* The same with the levels above the nodes (configs[3] through the graph object): every rank's graph -> ONE resident index on every
 * rank -- node tables as above, plus floors -> rooms, the rooms' keys and view embeddings and (on every rank or on none) the embeddings of
 * the rooms' names, all with global ids: room = room_off[rank] + local, floor = floor_off[rank] + local ([world + 1] each, optional).
 * hmsg_query_hier on the result addresses a storey by its global floor id.  (The benchmark driver used to gather these tables as
 * pickled Python objects inside its timed step.) */
int hmsg_graph_allgather_index(hmsg_graph_t* g, hmsg_comm_t* c, const double* room_name_emb, hmsg_index_t** out, int64_t* node_off,
                               int64_t* room_off, int64_t* floor_off);
/* Point to point: `bytes` of a DEVICE buffer to / from rank dst / src (ncclSend / ncclRecv on the handle's stream; returns when the
 * transfer has completed).  The cross-rank joins of the sharded merge tree are made of these; a host that schedules them itself can
 * use the pair directly. */
int hmsg_comm_send(hmsg_t* h, hmsg_comm_t* c, const void* dev_buf, int64_t bytes, int32_t dst);
int hmsg_comm_recv(hmsg_t* h, hmsg_comm_t* c, void* dev_buf, int64_t bytes, int32_t src);
/* hierarchical_merge (graph_utils.py:989-1012) of ONE episode whose frame windows are spread over the ranks (configs[4]), one call
 * per rank after hmsg_fuse_frames (+ hmsg_allreduce_feature_sums): the levels inside the rank's window (hmsg_merge_tree_local), the
 * ranks' agreement on the level they meet at (an all-gather of 16 bytes), then level by level the owner of list 2k + 1 sends its
 * clouds -- count, sizes, points, HBM to HBM -- to the owner of list 2k, which merges [mine ++ theirs] (hmsg_merge_tree_join; the
 * last join runs the final pass).  Any number of ranks whose windows are subtrees of the merge tree (hmsg_set_frame_window).
 * *holds_result = 1 on the rank that ends with the episode's instances (the rank of frame 0), bit-identical to a one-process
 * hmsg_merge_instances; hmsg_pool_instances follows there.  Until round 5 this schedule lived in Python over torch.distributed. */
int hmsg_merge_tree_sharded(hmsg_t* h, hmsg_comm_t* c, int32_t total_frames, int32_t* holds_result);
/* One episode fused in disjoint frame windows (configs[4]): the per-voxel feature sums and frame counters of all ranks are
 * all-reduced in place (graph.py:410-415: `sum[idx] += F; cnt[idx] += 1` over the frames commute across windows; counters
 * exact, float32 sums up to summation order) and the voxel features refreshed.  After hmsg_fuse_frames, before pooling. */
int hmsg_allreduce_feature_sums(hmsg_t* h, hmsg_comm_t* c);


#ifdef __cplusplus
}
#endif
#endif /* HMSG_H */
