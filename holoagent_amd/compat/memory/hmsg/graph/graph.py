"""fsr_vln/memory/hmsg/graph/graph.py under its own import path: the MI355X-native Graph (holoagent_amd/graph.py)."""
from holoagent_amd.graph import *  # noqa: F401,F403
from holoagent_amd.graph import Floor, Graph, Object, Room, View  # noqa: F401
