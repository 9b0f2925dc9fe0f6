"""The multi-GPU exchange steps behind the C ABI (include/hmsg.h: hmsg_comm_*, hmsg_allgather_nodes,
hmsg_allreduce_feature_sums; holoagent_amd/csrc/hmsg_comm.hip): RCCL collectives on device buffers.

Without a GPU: the library exports the entry points, a one-rank communicator without an id is the identity (the all-gather
leg is checked inside tests/test_emu_parity.py::test_graph_end_to_end_tiny), bad arguments are refused.
On the MI355X (-m gpu): a REAL RCCL communicator with one rank -- ncclGetUniqueId / ncclCommInitRank / ncclAllGather /
ncclAllReduce all run on the box's GPU -- must reproduce the local table and leave the sums as they were.  (N > 1 ranks need
N GPUs: the driver's multi-GPU runs; tests/test_distributed_gloo.py covers the rank arithmetic on CPU.)"""
import os

import numpy as np
import pytest

from tests import parity_common as PC


def _tiny_scene(L):
    from holoagent_amd.synth import SceneSpec, SynthScene
    spec = SceneSpec(seed=3, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4, width=128, height=96, n_frames=12,
                     n_masks=8, feat_dim=64)
    scn = SynthScene(spec)
    frames = [scn.frame(i) for i in range(spec.n_frames)]
    sc = PC.make_scene(L, frames, dict(feat_dim=64, outlier_nb_points=200, feat_dbscan_min=20))
    S = PC.stack_frames(frames)
    sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
    sc.finalize_map()
    sc.add_frame_features(0, S["masks"], S["f_g"], S["f_masked"], S["f_crop"], S["n_masks"])
    sc.fuse_frames()
    return sc, scn


def check_allreduce_identity(L, comm):
    sc, _ = _tiny_scene(L)
    V, D = sc.map_size(), sc.cfg.feat_dim
    s0, c0 = np.empty((V, D), np.float32), np.empty(V, np.uint32)
    sc.feature_sums_into(s0, c0)
    f0 = sc.map_feats().copy()
    sc.allreduce_feature_sums(comm)
    s1, c1 = np.empty((V, D), np.float32), np.empty(V, np.uint32)
    sc.feature_sums_into(s1, c1)
    assert np.array_equal(s0, s1) and np.array_equal(c0, c1) and np.array_equal(f0, sc.map_feats())
    sc.close()


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_single_rank_without_communicator_emu():
    from holoagent_amd._lib import Comm, HmsgError, HmsgLib
    L = HmsgLib(PC.EMU_PATH)
    cm = Comm.single(lib_=L)
    assert (cm.rank, cm.world) == (0, 1)
    check_allreduce_identity(L, cm)
    cm.close()
    with pytest.raises(HmsgError):                       # two ranks need an id
        import ctypes as C
        h = C.c_void_p()
        rc = L.c.hmsg_comm_create(None, 0, 2, 0, C.byref(h))
        assert rc != 0
        raise HmsgError("refused")
    import ctypes as C
    h = C.c_void_p()
    assert L.c.hmsg_comm_create(None, 3, 2, 0, C.byref(h)) != 0          # rank outside the world


@pytest.mark.gpu
def test_rccl_communicator_with_one_rank_gpu(tmp_path):
    from holoagent_amd._lib import Comm, HmsgLib
    L = HmsgLib()
    uid = Comm.unique_id(L)                              # ncclGetUniqueId through the C ABI
    assert len(uid) == 128 and any(uid)
    cm = Comm.create(uid, 0, 1, 0, L)                    # ncclCommInitRank on this GPU
    check_allreduce_identity(L, cm)                      # ncclAllReduce on the handle's own buffers
    # ncclAllGather of the node table: the tiny graph of the end-to-end test, all-gathered through the real communicator
    from holoagent_amd.graph import Graph
    from tests.graph_fixture import SynthDataset, SynthEncoders, tiny_scene
    scn = tiny_scene(4, 32)
    ds = SynthDataset(scn)
    enc = SynthEncoders(ds, ["background", "wall", "office", "kitchen", "chair", "table"])
    cfg = dict(main=dict(device_id=0), models=dict(clip=dict(type="ViT-B/32", feat_dim=32)),
               pipeline=dict(voxel_size=0.05, skip_frames=1, merge_type="sequential", max_masks=8))
    g = Graph(cfg, dataset=ds, encoders=enc, lib=L)
    g.create_feature_map()
    g.set_label_feats(enc.encode_text(["chair", "table"]), ["chair", "table"])
    lo, hi = scn.rooms[0]
    rooms = [dict(floor=0, name="office", vertices=[[x, z] for x in np.arange(lo[0], hi[0], 0.1) for z in np.arange(lo[2], hi[2], 0.1)])]
    g.build_hier_multimodal_scene_graph(None, rooms=rooms)
    assert g.objects
    ixn = g.scene.index_from_nodes()
    ixg, noff, roff = g.scene.allgather_nodes(cm, len(g.rooms))
    assert list(noff) == [0, len(g.objects)] and list(roff) == [0, 1]
    Tq = g.get_text_feats_multiple_templates(["chair", "background"])[None]
    a = ixn.query_objects(Tq, np.zeros(1, np.int32), [[0]], 3)
    b = ixg.query_objects(Tq, np.zeros(1, np.int32), [[0]], 3)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    ixg.close()
    ixn.close()
    cm.close()


def _from_torch_worker(rank, world, port, out):
    import torch.distributed as dist
    from holoagent_amd._lib import Comm, HmsgError, HmsgLib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Comm.from_torch(0, HmsgLib(PC.EMU_PATH))
        res = "created"
    except HmsgError as e:
        res = "refused: %s" % e
    open("%s.%d" % (out, rank), "w").write(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_every_rank_takes_the_same_way_out_without_gpus(tmp_path):
    """Comm.from_torch with two gloo ranks on a box WITHOUT GPUs: the communicator cannot be made (ncclCommInitRank has no
    device), and both ranks learn so together -- bench.py then falls back to the torch.distributed forms on every rank instead
    of hanging half of them in a collective."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "res")
    mp.spawn(_from_torch_worker, args=(2, port, out), nprocs=2, join=True)
    r = [open("%s.%d" % (out, k)).read() for k in range(2)]
    assert r[0].split(":")[0] == r[1].split(":")[0], r
