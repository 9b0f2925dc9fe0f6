"""fsr_vln/memory/hmsg/graph/view.py under its own import path."""
from holoagent_amd.graph import View  # noqa: F401
