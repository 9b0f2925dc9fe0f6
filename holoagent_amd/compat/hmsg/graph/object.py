"""fsr_vln/memory/hmsg/graph/object.py under its own import path."""
from holoagent_amd.graph import Object  # noqa: F401
