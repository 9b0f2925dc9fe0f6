"""fsr_vln/memory/hmsg/utils/label_feats.py under its own import path."""
from holoagent_amd.label_feats import *  # noqa: F401,F403
