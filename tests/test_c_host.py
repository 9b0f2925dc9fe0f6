"""The drop-in boundary from C: tests/host_c/hmsg_host.c is compiled with gcc as strict C99 against include/hmsg.h ALONE
(and the header is parsed as C++11 too), linked with the library, run on a small scene, and its answers are compared bit
for bit with the same calls made through the Python binding.  CPU: against the kernel simulator build of the same
sources; -m gpu: against libhmsg.so on the MI355X."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import parity_common as PC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_c", "hmsg_host.c")
INC = os.path.join(ROOT, "include")
LIB = os.path.join(ROOT, "holoagent_amd", "libhmsg.so")


def _build(lib_path, out):
    d = os.path.dirname(lib_path)
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", INC, SRC, "-o", out, lib_path,
           "-Wl,-rpath," + d, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def _scene():
    from holoagent_amd.synth import SceneSpec, SynthScene
    # two closed rooms side by side, seen from the inside: the room segmentation finds both
    spec = SceneSpec(seed=40, rooms_x=2, rooms_z=1, room_size=(3.2, 2.6, 3.0), objects_per_room=3, width=96, height=72,
                     n_frames=12, n_masks=5, feat_dim=16, yaw_step_deg=60.0)
    sc = SynthScene(spec)
    frames = [sc.frame(i) for i in range(spec.n_frames)]
    text, _ = sc.text_table(5)
    return spec, frames, np.ascontiguousarray(text, np.float32)


def _run(lib_path, tmp_path, merge=False):
    from holoagent_amd._lib import HmsgLib
    spec, frames, text = _scene()
    S = PC.stack_frames(frames)
    F, (H, W), M, D, Q, k = len(frames), frames[0]["depth"].shape, S["masks"].shape[1], spec.feat_dim, text.shape[0], 3
    over = dict(feat_dim=D, outlier_nb_points=40, feat_dbscan_min=8, outlier_radius=0.5)
    rng = np.random.Generator(np.random.PCG64(5))
    room_names = rng.standard_normal((8, D))
    room_names /= np.linalg.norm(room_names, axis=1, keepdims=True)
    room_text = np.ascontiguousarray(room_names[np.arange(Q) % 2] + 0.05 * rng.standard_normal((Q, D)), np.float32)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        np.array([F, H, W, M, D, Q, k, over["outlier_nb_points"], over["feat_dbscan_min"], int(round(over["outlier_radius"] * 1000))], np.int32).tofile(f)
        for a, t in ((S["K"], np.float64), (S["rgb"], np.uint8), (S["depth"], np.uint16), (S["pose"], np.float64),
                     (S["masks"], np.uint8), (S["n_masks"], np.int32), (S["f_g"], np.float32), (S["f_masked"], np.float32),
                     (S["f_crop"], np.float32), (text, np.float32), (room_text, np.float32), (room_names, np.float64)):
            np.ascontiguousarray(a, t).tofile(f)
    exe = _build(lib_path, str(tmp_path / "hmsg_host"))
    gdir = str(tmp_path / "graph_c")
    env = dict(os.environ)
    if merge:
        env["HMSG_HOST_MERGE_OBJECTS"] = "1"       # hmsg_graph_params::merge_objects_graph from the C host
    r = subprocess.run([exe, fin, fout, gdir], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(fout, "rb").read()
    V, N, n_floors, n_nodes = np.frombuffer(raw, np.int64, 4)
    o = 32
    sizes = np.frombuffer(raw, np.int64, N, o); o += 8 * N
    feats = np.frombuffer(raw, np.float32, N * D, o).reshape(N, D); o += 4 * N * D
    idx = np.frombuffer(raw, np.int32, Q * k, o).reshape(Q, k); o += 4 * Q * k
    score = np.frombuffer(raw, np.float64, Q * k, o).reshape(Q, k); o += 8 * Q * k
    n_rooms, rows, cols, n_nodes2 = np.frombuffer(raw, np.int64, 4, o); o += 32
    markers = np.frombuffer(raw, np.int32, rows * cols, o).reshape(rows, cols); o += 4 * rows * cols
    nsel = np.frombuffer(raw, np.int32, Q, o); o += 4 * Q
    sel = np.frombuffer(raw, np.int32, Q * 8, o).reshape(Q, 8); o += 32 * Q
    hidx = np.frombuffer(raw, np.int32, Q * k, o).reshape(Q, k); o += 4 * Q * k
    hscore = np.frombuffer(raw, np.float64, Q * k, o).reshape(Q, k); o += 8 * Q * k
    n_edges = int(np.frombuffer(raw, np.int64, 1, o)[0]); o += 8
    c_edges = np.frombuffer(raw, np.int64, n_edges * 2, o).reshape(n_edges, 2); o += 16 * n_edges
    gcounts = np.frombuffer(raw, np.int32, 4, o); o += 16
    g_edges = int(np.frombuffer(raw, np.int64, 1, o)[0]); o += 8
    gnsel = np.frombuffer(raw, np.int32, Q, o); o += 4 * Q
    gsel = np.frombuffer(raw, np.int32, Q * 16, o).reshape(Q, 16); o += 64 * Q
    gidx = np.frombuffer(raw, np.int32, Q * k, o).reshape(Q, k); o += 4 * Q * k
    gscore = np.frombuffer(raw, np.float64, Q * k, o).reshape(Q, k); o += 8 * Q * k
    assert o == len(raw)
    # the same calls through the Python binding
    L = HmsgLib(lib_path)
    sc = PC.make_scene(L, frames, over)
    sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
    sc.finalize_map()
    sc.add_frame_features(0, S["masks"], S["f_g"], S["f_masked"], S["f_crop"], S["n_masks"])
    sc.fuse_frames()
    sc.merge_instances()
    sc.pool_instances()
    assert V == sc.map_size() and N == sc.num_instances() and N >= 3
    assert np.array_equal(sizes, [len(c) for c in sc.instances()])
    assert np.array_equal(feats, sc.instance_feats())
    fl = sc.segment_floors()
    assert n_floors == len(fl) >= 1
    box = np.array([[-100.0, -100.0], [100.0, -100.0], [100.0, 100.0], [-100.0, 100.0]])
    nodes = sc.build_object_nodes([f["zero_level"] for f in fl], [f["height"] for f in fl], [0], [box], None)
    assert n_nodes == len(nodes) >= 1
    ix = sc.index_from_nodes()
    idx2, _, score2 = ix.query_objects(text, np.zeros(Q, np.int32), [[0]] * Q, k)
    assert np.array_equal(idx, idx2) and np.array_equal(score, score2)
    ix.close()
    # the room level driven from C: device room segmentation -> regions -> object nodes -> floor / room / object query
    m2, nr2, xz = sc.segment_rooms(fl[0]["y_lo"], fl[0]["y_hi"], fl[0]["zero_level"], fl[0]["height"], 0.1)
    assert (rows, cols) == m2.shape and n_rooms == min(nr2, 8) >= 2 and np.array_equal(markers, m2)
    regions = []
    for i in range(n_rooms):
        y_cells, x_cells = np.where(m2 == i + 1)
        regions.append(np.column_stack(((x_cells - 10.5) * 0.1 + xz[0], (y_cells - 10.5) * 0.1 + xz[1])))
    nodes2 = sc.build_object_nodes([f["zero_level"] for f in fl], [f["height"] for f in fl], [0] * n_rooms, regions, None)
    assert n_nodes2 == len(nodes2) >= 1 and len({int(n["room"]) for n in nodes2}) >= 2        # objects in both rooms
    ix = sc.index_from_nodes()
    ix.set_hierarchy([list(range(n_rooms))] + [[] for _ in fl[1:]], room_names[:n_rooms], [np.zeros((0, D))] * n_rooms, list(range(n_rooms)))
    sel2, hidx2, _, hscore2 = ix.query_hier(text, np.zeros(Q, np.int32), room_text, np.zeros(Q, np.int32), np.ones(Q, np.int32), k)
    for q in range(Q):
        assert list(sel[q][: nsel[q]]) == list(sel2[q]), q
    assert np.array_equal(hidx, hidx2) and np.array_equal(hscore, hscore2)
    ix.close()
    # create_graph_new's edges from C (hmsg_graph_edges): building - storeys - rooms - objects of this graph
    from holoagent_amd._lib import graph_edges
    e2 = graph_edges(len(fl), [0] * n_rooms, [int(n["room"]) for n in nodes2], [], [], lib_=L)
    assert n_edges == len(e2) == len(fl) + n_rooms + len(nodes2) and np.array_equal(c_edges, e2)
    # THE GRAPH WITH FOUR CALLS (hmsg_build_graph, hmsg_save, hmsg_load, hmsg_graph_query) from C == the same four through the binding:
    # same counts, the two saved directories byte for byte, same answers
    from holoagent_amd._lib import SceneGraph
    cg = SceneGraph.build(sc, S["pose"], S["f_g"], merge_objects_graph=1 if merge else 0)
    if merge:
        assert cg.counts()["objects"] < len(sc.nodes()), "no pair of objects was merged"     # (Room.merge_objects did fuse something)
    cnt = cg.counts()
    assert list(gcounts) == [cnt["floors"], cnt["rooms"], cnt["views"], cnt["objects"]] and g_edges == cnt["edges"]
    assert cnt["rooms"] >= 2 and cnt["views"] == F and cnt["objects"] >= 1
    pdir = tmp_path / "graph_py"
    cg.save(pdir)
    for sub in ("floors", "rooms", "objects", "views"):
        a, b = sorted(os.listdir(pdir / sub)), sorted(os.listdir(os.path.join(gdir, sub)))
        assert a == b and a, sub
        for f in a:
            assert open(pdir / sub / f, "rb").read() == open(os.path.join(gdir, sub, f), "rb").read(), (sub, f)
    lg = SceneGraph.load(pdir, lib_=L)
    minus = np.full(Q, -1, np.int32)
    sel3, idx3, _, score3 = lg.query(text, np.zeros(Q, np.int32), room_text, minus, np.full(Q, 2, np.int32), k, max_rooms=16)
    for q in range(Q):
        assert list(gsel[q][: gnsel[q]]) == list(sel3[q]), q
    assert np.array_equal(gidx, idx3) and np.array_equal(gscore, score3) and (gidx >= 0).any()
    lg.close()
    cg.close()
    sc.close()


def test_header_is_plain_c_and_cxx(tmp_path):
    """include/hmsg.h on its own: strict C99 and C++11 front ends, no torch / HIP types in the signatures."""
    src = tmp_path / "t.c"
    src.write_text('#include "hmsg.h"\nint main(void) { hmsg_config c; hmsg_default_config(&c); return c.feat_dim < 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", INC, str(src)], check=True)
    subprocess.run(["g++", "-std=c++11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", "-I", INC, str(src)],
                   check=True)
    import re
    code = re.sub(r"/\*.*?\*/", "", open(os.path.join(INC, "hmsg.h")).read(), flags=re.S)      # declarations only
    assert "torch" not in code and "hipStream" not in code and "at::" not in code and "#include <hip" not in code


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_c_host_equals_binding_emu(tmp_path):
    _run(PC.EMU_PATH, tmp_path)


@pytest.mark.gpu
def test_c_host_equals_binding_gpu(tmp_path):
    _run(LIB, tmp_path)


@pytest.mark.skipif(not os.environ.get("HMSG_EMU_SLOW"), reason="minutes on the kernel simulator (HMSG_EMU_SLOW=1); its twin runs on the MI355X")
@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_c_host_merges_objects_like_the_binding_emu(tmp_path):
    """pipeline.merge_objects_graph from the strict-C99 host: hmsg_graph_params::merge_objects_graph = 1 -> the same merged graph (counts,
    saved directory byte for byte, answers of the loaded graph) as through the binding"""
    _run(PC.EMU_PATH, tmp_path, merge=True)


@pytest.mark.gpu
def test_c_host_merges_objects_like_the_binding_gpu(tmp_path):
    _run(LIB, tmp_path, merge=True)

