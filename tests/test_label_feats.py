"""Label vocabulary loader (holoagent_amd/label_feats.py) against the reference's get_label_feats
(tests/golden/labels.json, made by oracle/refdrive/gen_golden_labels.py)."""
import hashlib
import json
import os

import numpy as np
import pytest

from holoagent_amd import label_feats as LF

REF_LABELS = "/root/reference/fsr_vln/memory/hmsg/labels"
GOLD = os.path.join(os.path.dirname(__file__), "golden", "labels.json")


def _enc(calls):
    def encode(classes):
        calls.append(len(classes))
        return np.arange(len(classes) * 4, dtype=np.float32).reshape(len(classes), 4)
    return encode


def test_csv_header_eats_the_first_label_and_cache_is_used(tmp_path):
    (tmp_path / "scannet200.csv").write_text("shower head\nspray\ninhaler\nguitar case\n")
    calls = []
    feats, classes = LF.get_label_feats(_enc(calls), "SCANNET200", str(tmp_path))
    assert classes == ["spray", "inhaler", "guitar case"] and feats.shape == (3, 4) and calls == [3]
    assert os.path.exists(tmp_path / "text_feats_SCANNET200_LABELS.npy")
    feats2, classes2 = LF.get_label_feats(_enc(calls), "SCANNET200", str(tmp_path))      # second call: cache, no encoder
    assert calls == [3] and np.array_equal(feats, feats2) and classes2 == classes


def test_registered_and_literal_vocabularies(tmp_path):
    calls = []
    LF.register_label_set("MY_LABELS", ["chair", "table"])
    feats, classes = LF.get_label_feats(_enc(calls), "MY_LABELS", str(tmp_path))
    assert classes == ["chair", "table"] and os.path.exists(tmp_path / "text_feats_MY_LABELS.npy")
    feats, classes = LF.get_label_feats(_enc(calls), ["a", "b", "c"])
    assert feats.shape == (3, 4) and classes == ["a", "b", "c"]
    with pytest.raises(KeyError):
        LF.get_label_feats(_enc(calls), "NO_SUCH_SET", str(tmp_path))
    with pytest.raises(ValueError):
        LF.get_label_feats(_enc(calls), "SCANNET20")


@pytest.mark.skipif(not os.path.isdir(REF_LABELS), reason="the vocabularies are data of the reference checkout")
def test_class_lists_equal_the_reference_loader():
    gold = json.load(open(GOLD))
    for name, (csv, _) in LF.CSV_SETS.items():
        classes = LF.read_label_csv(os.path.join(REF_LABELS, csv))
        g = gold[name]
        assert len(classes) == g["n"] and str(classes[0]) == g["first"] and str(classes[-1]) == g["last"]
        assert hashlib.sha1("\n".join(str(c) for c in classes).encode()).hexdigest() == g["sha1"]


def test_graph_loads_its_vocabulary_from_the_config(tmp_path):
    from holoagent_amd.graph import Graph
    (tmp_path / "final_label.csv").write_text("floor\nwall\nchair\n")

    class Enc:
        def encode_text(self, prompts):
            return np.ones((len(prompts), 6), np.float32)
    g = Graph.__new__(Graph)
    g.cfg = {"pipeline": {"obj_labels": "FINALLABEL", "label_dir": str(tmp_path)}}
    g.encoders, g._text_cache, g._label_feats = Enc(), {}, None
    feats, classes = g.load_label_feats()
    assert classes == ["wall", "chair"] and feats.shape == (2, 6) and g._label_feats[1] == classes
