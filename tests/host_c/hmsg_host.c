/* A plain C99 host of the drop-in boundary: compiled with `gcc -std=c99 -pedantic -Wall -Wextra -Werror` against
 * include/hmsg.h ALONE and linked with libhmsg.so (tests/test_c_host.py).  It drives the whole path the way a C / C++
 * service would -- create, frames, map, encoder outputs, fusion, merge, pooling, storeys, object nodes, resident
 * index, object queries -- on a scene read from a flat binary file, and writes what it got to another one; the test
 * compares that with the same calls made through the Python binding, bit for bit.
 *
 *   hmsg_host <in.bin> <out.bin> [<graph dir>]
 *   in : i32 F H W M D Q k outlier_nb feat_dbscan_min outlier_radius_mm | f64 K[9] | u8 rgb[F][H][W][3] | u16 depth[F][H][W] |
 *        f64 pose[F][16] | u8 masks[F][M][H][W] | i32 n_masks[F] | f32 f_g[F][D] | f32 f_masked[F][M][D] |
 *        f32 f_crop[F][M][D] | f32 text[Q][2][D] | f32 room_text[Q][D] | f64 room_names[8][D]
 *   out: i64 V, N, n_floors, n_nodes | i64 sizes[N] | f32 feats[N][D] | i32 idx[Q][k] | f64 score[Q][k]
 *        | i64 n_rooms, rows, cols, n_nodes2 | i32 markers[rows][cols] | i32 nsel[Q] | i32 sel[Q][8] | i32 hidx[Q][k] | f64 hscore[Q][k]
 *   (second part: rooms of storey 0 by the device room segmentation (hmsg_segment_rooms), their regions as the room
 *    vertices of hmsg_build_object_nodes, create_graph_new's edges of that graph (hmsg_graph_edges), and the coarse-to-fine
 *    query floor -> room by name -> objects, hmsg_query_hier)
 *   with <graph dir>, third part -- THE GRAPH WITH FOUR CALLS: hmsg_build_graph (floors, rooms, views, objects, edges held by the
 *    library), hmsg_save (the reference's directory layout), hmsg_load, hmsg_graph_query (rooms by their view embeddings, then
 *    objects); appended to out: i32 counts[4] (floors, rooms, views, objects) | i64 n_edges | i32 gsel_n[Q] | i32 gsel[Q][16] |
 *    i32 gidx[Q][k] | f64 gscore[Q][k] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hmsg.h"

static void* rd(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p || (bytes && fread(p, 1, bytes, f) != bytes)) {
        fprintf(stderr, "hmsg_host: short read (%lu bytes)\n", (unsigned long)bytes);
        exit(2);
    }
    return p;
}

#define CK(call)                                                                          \
    do {                                                                                  \
        int rc_ = (call);                                                                 \
        if (rc_ != HMSG_OK) {                                                             \
            fprintf(stderr, "hmsg_host: %s -> %d: %s\n", #call, rc_, hmsg_last_error(h)); \
            return 3;                                                                     \
        }                                                                                 \
    } while (0)

int main(int argc, char** argv) {
    FILE *fi, *fo;
    int32_t hd[10];
    int32_t F, H, W, M, D, Q, k;
    double* K;
    uint8_t *rgb, *masks;
    uint16_t* depth;
    double* pose;
    int32_t* n_masks;
    float *f_g, *f_masked, *f_crop, *text, *room_text;
    double* room_names;
    int32_t rows = 0, cols = 0, n_rooms = 0, r, c2, *markers = NULL, *rfl, *froff, *frooms, *rkey, *fid, *mode, *nsel, *sel, *hidx, *hroom;
    int64_t *voff2, *view_off, out_hd2[4], n_nodes2 = 0, cells, n_edges = 0, *edges = NULL;
    hmsg_node* nd2;
    int32_t* obj_room;
    double xz_min[2], *verts2, *hscore;
    hmsg_graph_t *g = NULL, *lg = NULL;
    hmsg_graph_counts gc;
    hmsg_graph_params gprm;
    int32_t *gfid, *gmode, *gqid, *gnsel, *gsel, *gidx, *groom, gcounts[4];
    double* gscore;
    hmsg_config cfg;
    hmsg_t* h = NULL;
    hmsg_index_t* ix = NULL;
    int64_t V, N, n_nodes, out_hd[4];
    int64_t* sizes;
    float* feats;
    int32_t n_floors = 0, q, *qid, *room_off, *rooms, *idx, *room;
    hmsg_floor* floors;
    double *fz, *fh, *score, verts[8];
    int64_t vert_off[2];
    int32_t room_floor[1];
    size_t HW;

    if (argc != 3 && argc != 4) {
        fprintf(stderr, "usage: hmsg_host <in.bin> <out.bin>   (%s)\n", hmsg_version());
        return 1;
    }
    fi = fopen(argv[1], "rb");
    if (!fi) return 1;
    if (fread(hd, 4, 10, fi) != 10) return 2;
    F = hd[0]; H = hd[1]; W = hd[2]; M = hd[3]; D = hd[4]; Q = hd[5]; k = hd[6];
    HW = (size_t)H * (size_t)W;
    K = (double*)rd(fi, 9 * sizeof(double));
    rgb = (uint8_t*)rd(fi, (size_t)F * HW * 3);
    depth = (uint16_t*)rd(fi, (size_t)F * HW * 2);
    pose = (double*)rd(fi, (size_t)F * 16 * sizeof(double));
    masks = (uint8_t*)rd(fi, (size_t)F * (size_t)M * HW);
    n_masks = (int32_t*)rd(fi, (size_t)F * 4);
    f_g = (float*)rd(fi, (size_t)F * (size_t)D * 4);
    f_masked = (float*)rd(fi, (size_t)F * (size_t)M * (size_t)D * 4);
    f_crop = (float*)rd(fi, (size_t)F * (size_t)M * (size_t)D * 4);
    text = (float*)rd(fi, (size_t)Q * 2 * (size_t)D * 4);
    room_text = (float*)rd(fi, (size_t)Q * (size_t)D * sizeof(float));
    room_names = (double*)rd(fi, (size_t)8 * (size_t)D * sizeof(double));
    fclose(fi);

    hmsg_default_config(&cfg);
    cfg.device_id = 0;
    cfg.feat_dim = D;
    cfg.height = H;
    cfg.width = W;
    cfg.max_frames = F;
    cfg.max_masks = M;
    cfg.outlier_radius = (double)hd[9] / 1000.0;
    cfg.outlier_nb_points = hd[7];
    cfg.feat_dbscan_min = hd[8];
    if (hmsg_create(&cfg, &h) != HMSG_OK || !h) {
        fprintf(stderr, "hmsg_host: hmsg_create failed: %s\n", hmsg_last_error(NULL));
        return 4;
    }
    /* loop A, A1 + A2 */
    CK(hmsg_add_frames(h, F, rgb, depth, pose, K));
    CK(hmsg_finalize_map(h));
    V = hmsg_map_size(h);
    /* loop B in two hand-overs (a service streams them), A3 - A5 */
    CK(hmsg_add_frame_features(h, 0, F / 2, M, masks, f_g, f_masked, f_crop, n_masks));
    CK(hmsg_add_frame_features(h, F / 2, F - F / 2, M, masks + (size_t)(F / 2) * (size_t)M * HW, f_g + (size_t)(F / 2) * (size_t)D,
                               f_masked + (size_t)(F / 2) * (size_t)M * (size_t)D, f_crop + (size_t)(F / 2) * (size_t)M * (size_t)D,
                               n_masks + F / 2));
    CK(hmsg_fuse_frames(h));
    /* A6, A7 */
    CK(hmsg_merge_instances(h));
    CK(hmsg_pool_instances(h));
    N = hmsg_num_instances(h);
    sizes = (int64_t*)malloc((size_t)(N ? N : 1) * sizeof(int64_t));
    feats = (float*)malloc((size_t)(N ? N : 1) * (size_t)D * sizeof(float));
    if (N) {
        CK(hmsg_get_instance_sizes(h, sizes));
        CK(hmsg_get_instance_feats(h, feats));
    }
    /* A8: storeys; A10: object nodes with one room that spans the whole map */
    CK(hmsg_segment_floors(h, NULL, 0, &n_floors));
    floors = (hmsg_floor*)malloc((size_t)(n_floors ? n_floors : 1) * sizeof(hmsg_floor));
    CK(hmsg_segment_floors(h, floors, n_floors, &n_floors));
    fz = (double*)malloc((size_t)(n_floors ? n_floors : 1) * sizeof(double));
    fh = (double*)malloc((size_t)(n_floors ? n_floors : 1) * sizeof(double));
    for (q = 0; q < n_floors; ++q) {
        fz[q] = floors[q].zero_level;
        fh[q] = floors[q].height;
    }
    verts[0] = -100.0; verts[1] = -100.0; verts[2] = 100.0; verts[3] = -100.0;
    verts[4] = 100.0;  verts[5] = 100.0;  verts[6] = -100.0; verts[7] = 100.0;
    vert_off[0] = 0;
    vert_off[1] = 4;
    room_floor[0] = 0;
    CK(hmsg_build_object_nodes(h, n_floors, fz, fh, 1, room_floor, vert_off, verts, 0, NULL));
    n_nodes = hmsg_num_nodes(h);
    /* A12: resident index over the node table, Q object queries with one negative prompt each */
    idx = (int32_t*)malloc((size_t)Q * (size_t)k * 4);
    room = (int32_t*)malloc((size_t)Q * (size_t)k * 4);
    score = (double*)malloc((size_t)Q * (size_t)k * sizeof(double));
    memset(idx, 0xff, (size_t)Q * (size_t)k * 4);
    memset(score, 0, (size_t)Q * (size_t)k * sizeof(double));
    if (n_nodes > 0) {
        qid = (int32_t*)calloc((size_t)Q, 4);
        room_off = (int32_t*)malloc((size_t)(Q + 1) * 4);
        rooms = (int32_t*)calloc((size_t)Q, 4);
        for (q = 0; q <= Q; ++q) room_off[q] = q;
        if (hmsg_index_from_nodes(h, &ix) != HMSG_OK) {
            fprintf(stderr, "hmsg_host: hmsg_index_from_nodes: %s\n", hmsg_last_error(h));
            return 5;
        }
        if (hmsg_query_objects(ix, Q, 2, text, qid, room_off, rooms, k, 1, idx, room, score) != HMSG_OK) {
            fprintf(stderr, "hmsg_host: hmsg_query_objects: %s\n", hmsg_index_last_error(ix));
            return 6;
        }
        hmsg_index_destroy(ix);
    }
    /* ---- the room level from the C side: rooms of storey 0 by the device watershed (N1), their regions as room vertices
     * (map_grid_to_point_cloud, graph_utils.py:359-388: cell (x, y) -> ((x - 10.5) res + min_x, (y - 10.5) res + min_z)),
     * object nodes against those rooms (A10), then floor -> room by its name -> objects on the device (A12) */
    nsel = (int32_t*)calloc((size_t)Q, 4);
    sel = (int32_t*)malloc((size_t)Q * 8 * 4);
    hidx = (int32_t*)malloc((size_t)Q * (size_t)k * 4);
    hroom = (int32_t*)malloc((size_t)Q * (size_t)k * 4);
    hscore = (double*)calloc((size_t)Q * (size_t)k, sizeof(double));
    memset(sel, 0xff, (size_t)Q * 8 * 4);
    memset(hidx, 0xff, (size_t)Q * (size_t)k * 4);
    if (n_floors > 0) {
        const double res = 0.1;
        CK(hmsg_segment_rooms(h, floors[0].y_lo, floors[0].y_hi, floors[0].zero_level, floors[0].height, res, NULL, 0, &rows, &cols, &n_rooms, xz_min));
        markers = (int32_t*)malloc((size_t)rows * (size_t)cols * 4 + 4);
        CK(hmsg_segment_rooms(h, floors[0].y_lo, floors[0].y_hi, floors[0].zero_level, floors[0].height, res, markers, (int64_t)rows * cols, &rows, &cols,
                              &n_rooms, xz_min));
        if (n_rooms > 8) n_rooms = 8;
        if (n_rooms > 0) {
            cells = 0;
            for (r = 0; r < rows * cols; ++r) cells += markers[r] >= 1 && markers[r] <= n_rooms;
            verts2 = (double*)malloc((size_t)(cells ? cells : 1) * 2 * sizeof(double));
            voff2 = (int64_t*)calloc((size_t)n_rooms + 1, sizeof(int64_t));
            rfl = (int32_t*)calloc((size_t)n_rooms, 4);
            cells = 0;
            for (r = 0; r < n_rooms; ++r) {                          /* np.where order: rows, then columns */
                int32_t y, x;
                for (y = 0; y < rows; ++y)
                    for (x = 0; x < cols; ++x)
                        if (markers[(size_t)y * (size_t)cols + (size_t)x] == r + 1) {
                            verts2[cells * 2] = ((double)x - 10.5) * res + xz_min[0];
                            verts2[cells * 2 + 1] = ((double)y - 10.5) * res + xz_min[1];
                            ++cells;
                        }
                voff2[r + 1] = cells;
            }
            CK(hmsg_build_object_nodes(h, n_floors, fz, fh, n_rooms, rfl, voff2, verts2, 0, NULL));
            n_nodes2 = hmsg_num_nodes(h);
            if (n_nodes2 > 0) {
                /* create_graph_new's edges of this graph (no views here): building - storeys - rooms - objects */
                nd2 = (hmsg_node*)malloc((size_t)n_nodes2 * sizeof(hmsg_node));
                obj_room = (int32_t*)malloc((size_t)n_nodes2 * 4);
                CK(hmsg_get_nodes(h, nd2, NULL));
                for (c2 = 0; c2 < (int32_t)n_nodes2; ++c2) obj_room[c2] = nd2[c2].room;
                edges = (int64_t*)malloc((size_t)(1 + n_floors + n_rooms + n_nodes2) * 2 * sizeof(int64_t));
                if (hmsg_graph_edges(n_floors, n_rooms, rfl, (int32_t)n_nodes2, obj_room, 0, NULL, NULL, NULL, edges,
                                     (int64_t)(1 + n_floors + n_rooms + n_nodes2), &n_edges) != HMSG_OK)
                    return 9;
                free(nd2);
                free(obj_room);
                froff = (int32_t*)calloc((size_t)n_floors + 1, 4);
                frooms = (int32_t*)malloc((size_t)n_rooms * 4);
                rkey = (int32_t*)malloc((size_t)n_rooms * 4);
                view_off = (int64_t*)calloc((size_t)n_rooms + 1, sizeof(int64_t));
                for (r = 0; r < n_rooms; ++r) frooms[r] = rkey[r] = r;
                for (c2 = 1; c2 <= n_floors; ++c2) froff[c2] = n_rooms;      /* every room lies on storey 0 */
                fid = (int32_t*)calloc((size_t)Q, 4);
                mode = (int32_t*)malloc((size_t)Q * 4);
                qid = (int32_t*)calloc((size_t)Q, 4);
                for (q = 0; q < Q; ++q) mode[q] = 1;
                if (hmsg_index_from_nodes(h, &ix) != HMSG_OK) return 7;
                if (hmsg_index_set_hierarchy(ix, n_rooms, n_floors, froff, frooms, room_names, view_off, NULL, rkey) != HMSG_OK ||
                    hmsg_query_hier(ix, Q, 2, text, qid, room_text, fid, mode, k, 1, 8, sel, nsel, hidx, hroom, hscore) != HMSG_OK) {
                    fprintf(stderr, "hmsg_host: hierarchical query: %s\n", hmsg_index_last_error(ix));
                    return 8;
                }
                hmsg_index_destroy(ix);
            }
        }
    }
    fo = fopen(argv[2], "wb");
    if (!fo) return 1;
    out_hd[0] = V; out_hd[1] = N; out_hd[2] = n_floors; out_hd[3] = n_nodes;
    fwrite(out_hd, sizeof(int64_t), 4, fo);
    fwrite(sizes, sizeof(int64_t), (size_t)N, fo);
    fwrite(feats, sizeof(float), (size_t)N * (size_t)D, fo);
    fwrite(idx, 4, (size_t)Q * (size_t)k, fo);
    fwrite(score, sizeof(double), (size_t)Q * (size_t)k, fo);
    out_hd2[0] = n_rooms; out_hd2[1] = rows; out_hd2[2] = cols; out_hd2[3] = n_nodes2;
    fwrite(out_hd2, sizeof(int64_t), 4, fo);
    if (markers) fwrite(markers, 4, (size_t)rows * (size_t)cols, fo);
    fwrite(nsel, 4, (size_t)Q, fo);
    fwrite(sel, 4, (size_t)Q * 8, fo);
    fwrite(hidx, 4, (size_t)Q * (size_t)k, fo);
    fwrite(hscore, sizeof(double), (size_t)Q * (size_t)k, fo);
    fwrite(&n_edges, sizeof(int64_t), 1, fo);
    if (n_edges) fwrite(edges, sizeof(int64_t), (size_t)n_edges * 2, fo);
    if (argc == 4) {
        /* 1: build -- build_hier_multimodal_scene_graph (the poses' inverses and the label vocabulary are optional) */
        /* (HMSG_HOST_MERGE_OBJECTS=1: with pipeline.merge_objects_graph -- every room fuses its same-name objects whose clouds overlap,
         *  Room.merge_objects room.py:62-129; without a vocabulary every object is named "object") */
        hmsg_graph_default_params(&gprm);
        if (getenv("HMSG_HOST_MERGE_OBJECTS")) gprm.merge_objects_graph = 1;
        CK(hmsg_build_graph(h, &gprm, F, pose, NULL, f_g, NULL, 0, NULL, NULL, &g));
        /* 2: save -- save_hmsg_graph's directory */
        if (hmsg_save(g, argv[3]) != HMSG_OK) {
            fprintf(stderr, "hmsg_host: hmsg_save: %s\n", hmsg_graph_last_error(g));
            return 10;
        }
        if (hmsg_graph_get_counts(g, &gc) != HMSG_OK) return 11;
        hmsg_graph_destroy(g);
        /* 3: load -- load_hmsg_graph; the scene handle is not needed any more */
        if (hmsg_load(argv[3], 0, &lg) != HMSG_OK) return 12;
        /* 4: query -- floor (all) -> rooms by their view embeddings (mode 2) -> objects with one negative prompt */
        gfid = (int32_t*)malloc((size_t)Q * 4);
        gmode = (int32_t*)malloc((size_t)Q * 4);
        gqid = (int32_t*)calloc((size_t)Q, 4);
        gnsel = (int32_t*)calloc((size_t)Q, 4);
        gsel = (int32_t*)malloc((size_t)Q * 16 * 4);
        gidx = (int32_t*)malloc((size_t)Q * (size_t)k * 4);
        groom = (int32_t*)malloc((size_t)Q * (size_t)k * 4);
        gscore = (double*)calloc((size_t)Q * (size_t)k, sizeof(double));
        memset(gsel, 0xff, (size_t)Q * 16 * 4);
        memset(gidx, 0xff, (size_t)Q * (size_t)k * 4);
        for (q = 0; q < Q; ++q) {
            gfid[q] = -1;
            gmode[q] = 2;
        }
        if (gc.objects > 0 && hmsg_graph_query(lg, NULL, Q, 2, text, gqid, room_text, gfid, gmode, k, 1, 16, gsel, gnsel, gidx, groom, gscore) != HMSG_OK) {
            fprintf(stderr, "hmsg_host: hmsg_graph_query: %s\n", hmsg_graph_last_error(lg));
            return 13;
        }
        gcounts[0] = gc.floors; gcounts[1] = gc.rooms; gcounts[2] = gc.views; gcounts[3] = gc.objects;
        fwrite(gcounts, 4, 4, fo);
        fwrite(&gc.edges, sizeof(int64_t), 1, fo);
        fwrite(gnsel, 4, (size_t)Q, fo);
        fwrite(gsel, 4, (size_t)Q * 16, fo);
        fwrite(gidx, 4, (size_t)Q * (size_t)k, fo);
        fwrite(gscore, sizeof(double), (size_t)Q * (size_t)k, fo);
        hmsg_graph_destroy(lg);
    }
    fclose(fo);
    hmsg_destroy(h);
    printf("hmsg_host ok: V %ld instances %ld floors %d nodes %ld\n", (long)V, (long)N, (int)n_floors, (long)n_nodes);
    return 0;
}
