"""A12 driver: holoagent_amd.graph.Graph.query_hierarchy_protected_icra against what the REFERENCE's own
query_hierarchy_protected_icra (graph.py:3483-3591, LLM parse replaced by a fixed triple) returned on the query
fixture's graph (tests/golden/query.npz `driver_json`, made by oracle/refdrive/gen_golden.py query): same floor,
same rooms, same objects in the same order, same negative labels.  Runs the retrieval on the kernel simulator."""
import json
import os

import numpy as np
import pytest

from tests import golden_io as GI
from tests import parity_common as PC


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_driver_matches_reference():
    from holoagent_amd._lib import HmsgLib
    check_driver_matches_reference(HmsgLib(PC.EMU_PATH))


@pytest.mark.gpu
def test_driver_matches_reference_gpu():
    from holoagent_amd._lib import HmsgLib
    check_driver_matches_reference(HmsgLib())


def check_driver_matches_reference(L):
    from holoagent_amd.graph import Floor, Graph, Object, Room
    z = GI.load("query")
    words = [str(w) for w in z["table_words"]]
    table = {w: z["table"][i] for i, w in enumerate(words)}
    table["Exhibition room1"] = table["room1"]
    D = z["table"].shape[1]
    g = Graph(dict(main=dict(), models=dict(clip=dict(feat_dim=D))), lib=L)
    g.get_text_feats_multiple_templates = lambda ws: np.stack([table[w] for w in ws]).astype(np.float32)
    for f, zero in enumerate(z["floor_zero"]):
        fl = Floor(str(f), name="floor_%d" % f)
        fl.floor_zero_level = float(zero)
        g.floors.append(fl)
    off = z["room_view_off"]
    for r, (fid, name) in enumerate(zip(z["room_floor"], z["room_name"])):
        fl = g.floors[int(fid)]
        room = Room("%s_%d" % (fl.floor_id, len(fl.rooms)), fl.floor_id, name=str(name))
        room.embeddings = [e for e in z["room_view_emb"][off[r]:off[r + 1]]]
        fl.add_room(room)
        g.rooms.append(room)
    for o, (emb, r) in enumerate(zip(z["obj_emb"], z["obj_room"])):
        room = g.rooms[int(r)]
        obj = Object("%s_%d" % (room.room_id, len(room.objects)), room.room_id, name="thing")
        obj.embedding = np.asarray(emb, np.float64)
        room.add_object(obj)
        g.objects.append(obj)
    for case in json.loads(str(z["driver_json"])):
        fl, rooms, objs, res = g.query_hierarchy_protected_icra(tuple(case["triple"]), top_k=3)
        assert (None if fl is None else fl.floor_id) == case["floor"], case["instruction"]
        assert [r.room_id for r in rooms] == case["rooms"], case["instruction"]
        assert [o.object_id for o in objs] == case["objects"], case["instruction"]
        assert [g.objects.index(o) for o in objs] == case["object_index"]
        assert res["negative_labels"] == case["negative_labels"]
