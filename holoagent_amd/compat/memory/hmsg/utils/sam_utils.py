"""fsr_vln/memory/hmsg/utils/sam_utils.py: the crop batching of the encoder hand-off (sam_utils.py:119-181) on the device."""
from holoagent_amd._lib import crop_all_bounding_boxs  # noqa: F401
