"""fsr_vln/memory/hmsg/graph/room.py under its own import path."""
from holoagent_amd.graph import Room  # noqa: F401
