"""The C-ABI library loads (no GPU needed) and exports every function include/hmsg.h (the boundary) and
include/hmsg_test.h (test hooks, the bench renderer) declare; the product loader refuses to run without it (no CPU
fallback); a C++ host compiles against hmsg.h alone and drives the path through it."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests import parity_common as PC

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_functions(headers=("hmsg.h", "hmsg_test.h")):
    names = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(hmsg_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_the_boundary_header_holds_no_test_hooks():
    names = declared_functions(("hmsg.h",))
    assert not [n for n in names if n.startswith("hmsg_test_") or n == "hmsg_synth_render"]
    assert "hmsg_test_dbscan" in declared_functions(("hmsg_test.h",))


def test_library_exports_every_declared_symbol():
    path = os.path.join(ROOT, "holoagent_amd", "libhmsg.so")
    if not os.path.exists(path):
        pytest.skip("libhmsg.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(path)
    names = declared_functions()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.hmsg_version.restype = ctypes.c_char_p
    assert b"hmsg" in lib.hmsg_version()


def test_binding_table_matches_header():
    from holoagent_amd._lib import EXPORTED_SYMBOLS
    assert sorted(EXPORTED_SYMBOLS) == declared_functions()


def test_no_cpu_fallback(tmp_path):
    from holoagent_amd._lib import HmsgError, HmsgLib
    with pytest.raises(HmsgError):
        HmsgLib(str(tmp_path / "libhmsg.so"))


def test_allocator_carves_a_large_parked_block():
    """DevCache carving (hmsg_common.h), on the kernel simulator (its hipMalloc is malloc: the 9 GB block is never touched)."""
    import pytest
    from tests import parity_common as PC
    if not os.path.exists(PC.EMU_PATH):
        pytest.skip("kernel simulator not built")
    if os.environ.get("HMSG_DEBUG_EXACT_ALLOC"):
        pytest.skip("the allocator's cache is switched off (sanitizer run)")
    from holoagent_amd._lib import HmsgLib
    L = HmsgLib(PC.EMU_PATH)
    assert L.c.hmsg_test_allocator_carving(0, 9) == 0


@pytest.mark.gpu
def test_allocator_carves_a_large_parked_block_gpu():
    from holoagent_amd._lib import HmsgLib
    assert HmsgLib().c.hmsg_test_allocator_carving(0, 9) == 0


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_config_skip_frames_and_depth_cut():
    """hmsg_config.skip_frames / depth_cut (config/semantic_scene_reconstruction_hm3d.yaml pipeline.skip_frames, graph.py:339;
    dataloader/horizon.py:258-261): offering all frames with skip_frames = 3 and a depth limit builds the map of the frames
    0, 3, 6, ... with the far pixels zeroed."""
    from holoagent_amd._lib import HmsgLib, Scene
    from holoagent_amd.synth import SceneSpec, SynthScene
    L = HmsgLib(PC.EMU_PATH)
    spec = SceneSpec(seed=9, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=3, width=96, height=72, n_frames=9, n_masks=4,
                     feat_dim=16)
    scn = SynthScene(spec)
    frames = [scn.frame(i) for i in range(spec.n_frames)]
    S = PC.stack_frames(frames)
    cut = 2.0
    a = Scene(lib_=L, height=72, width=96, max_frames=9, max_masks=4, feat_dim=16, outlier_nb_points=30, outlier_radius=0.5, skip_frames=3,
              depth_cut=cut)
    a.add_frames(S["rgb"][:5], S["depth"][:5], S["pose"][:5], S["K"])        # two calls: the count runs across them
    a.add_frames(S["rgb"][5:], S["depth"][5:], S["pose"][5:], S["K"])
    a.finalize_map()
    keep = [0, 3, 6]
    dep = S["depth"][keep].copy()
    dep[dep.astype(np.float64) > cut * 1000.0] = 0
    b = Scene(lib_=L, height=72, width=96, max_frames=9, max_masks=4, feat_dim=16, outlier_nb_points=30, outlier_radius=0.5)
    b.add_frames(np.ascontiguousarray(S["rgb"][keep]), np.ascontiguousarray(dep), np.ascontiguousarray(S["pose"][keep]), S["K"])
    b.finalize_map()
    assert a.map_size() == b.map_size() > 0 and np.array_equal(a.map_points(), b.map_points())
    a.close()
    b.close()


def test_integration_md_config_struct_is_the_headers():
    """INTEGRATION.md shows the ctypes struct a maintainer copies into the reference tree: its fields must be struct hmsg_config's
    (include/hmsg.h) -- same names, same order, same C types -- and its size the library's own sizeof (hmsg_config_size()).  (Round 4's
    document stopped three fields short: hmsg_default_config(byref(cfg)) would have written 160 bytes into a 136-byte object.)"""
    import ctypes as C
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    m = re.search(r"class HmsgConfig\(C\.Structure\):.*?_fields_ = \[(.*?)\]\n", doc, flags=re.S)
    assert m, "INTEGRATION.md no longer shows the HmsgConfig binding"
    doc_fields = re.findall(r'\("(\w+)",\s*C\.(c_\w+)\)', m.group(1))
    hdr = open(os.path.join(root, "include", "hmsg.h")).read()
    body = re.search(r"typedef struct hmsg_config \{(.*?)\} hmsg_config;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    hdr_fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype, names = decl.split(None, 1)
        for nme in names.split(","):
            hdr_fields.append((nme.strip(), {"int32_t": "c_int32", "double": "c_double", "int64_t": "c_int64", "uint32_t": "c_uint32"}[ctype]))
    assert doc_fields == hdr_fields
    from holoagent_amd._lib import HmsgConfig, HmsgLib
    assert [(n, t) for n, t in HmsgConfig._fields_] == [(n, getattr(C, t)) for n, t in hdr_fields]     # the shipped binding too

    class DocConfig(C.Structure):
        _fields_ = [(n, getattr(C, t)) for n, t in doc_fields]
    from tests import parity_common as PC
    L = HmsgLib(PC.EMU_PATH if os.path.exists(PC.EMU_PATH) else None)
    assert C.sizeof(DocConfig) == L.c.hmsg_config_size()
    cfg = DocConfig()
    raw = C.CDLL(L.path)                                       # (an untyped handle: the document's own struct goes through)
    raw.hmsg_default_config(C.byref(cfg))
    assert cfg.feat_dim == 512 and cfg.skip_frames == 1 and cfg.grid_resolution == 0.05 and cfg.overlap_distance_form == 0
