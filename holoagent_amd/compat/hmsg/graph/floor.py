"""fsr_vln/memory/hmsg/graph/floor.py under its own import path."""
from holoagent_amd.graph import Floor  # noqa: F401
