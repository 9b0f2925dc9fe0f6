/* A plain C99 host of the drop-in boundary: compiled with `gcc -std=c99 -pedantic -Wall -Wextra -Werror` against
 * include/hmsg.h ALONE and linked with libhmsg.so (tests/test_c_host.py).  It drives the whole path the way a C / C++
 * service would -- create, frames, map, encoder outputs, fusion, merge, pooling, storeys, object nodes, resident
 * index, object queries -- on a scene read from a flat binary file, and writes what it got to another one; the test
 * compares that with the same calls made through the Python binding, bit for bit.
 *
 *   hmsg_host <in.bin> <out.bin>
 *   in : i32 F H W M D Q k outlier_nb feat_dbscan_min | f64 K[9] | u8 rgb[F][H][W][3] | u16 depth[F][H][W] |
 *        f64 pose[F][16] | u8 masks[F][M][H][W] | i32 n_masks[F] | f32 f_g[F][D] | f32 f_masked[F][M][D] |
 *        f32 f_crop[F][M][D] | f32 text[Q][2][D]
 *   out: i64 V, N, n_floors, n_nodes | i64 sizes[N] | f32 feats[N][D] | i32 idx[Q][k] | f64 score[Q][k]            */
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
    int32_t hd[9];
    int32_t F, H, W, M, D, Q, k;
    double* K;
    uint8_t *rgb, *masks;
    uint16_t* depth;
    double* pose;
    int32_t* n_masks;
    float *f_g, *f_masked, *f_crop, *text;
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

    if (argc != 3) {
        fprintf(stderr, "usage: hmsg_host <in.bin> <out.bin>   (%s)\n", hmsg_version());
        return 1;
    }
    fi = fopen(argv[1], "rb");
    if (!fi) return 1;
    if (fread(hd, 4, 9, fi) != 9) return 2;
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
    fclose(fi);

    hmsg_default_config(&cfg);
    cfg.device_id = 0;
    cfg.feat_dim = D;
    cfg.height = H;
    cfg.width = W;
    cfg.max_frames = F;
    cfg.max_masks = M;
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
    fo = fopen(argv[2], "wb");
    if (!fo) return 1;
    out_hd[0] = V; out_hd[1] = N; out_hd[2] = n_floors; out_hd[3] = n_nodes;
    fwrite(out_hd, sizeof(int64_t), 4, fo);
    fwrite(sizes, sizeof(int64_t), (size_t)N, fo);
    fwrite(feats, sizeof(float), (size_t)N * (size_t)D, fo);
    fwrite(idx, 4, (size_t)Q * (size_t)k, fo);
    fwrite(score, sizeof(double), (size_t)Q * (size_t)k, fo);
    fclose(fo);
    hmsg_destroy(h);
    printf("hmsg_host ok: V %ld instances %ld floors %d nodes %ld\n", (long)V, (long)N, (int)n_floors, (long)n_nodes);
    return 0;
}
