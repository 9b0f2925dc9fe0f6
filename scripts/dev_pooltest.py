import sys; sys.path.insert(0,'/root/repo')
import numpy as np, os
from tests import golden_io as GI, parity_common as PC
from holoagent_amd._lib import HmsgLib
import oracle.hmsg_oracle as O
L=HmsgLib(PC.EMU_PATH if os.environ.get("USE_EMU") else None)
z=GI.load("build_hier"); frames=GI.unpack_frames(z)[:8]; cfg=GI.unpack_cfg(z); cfg["outlier_nb"]=300
sc=PC.make_scene(L,frames,dict(feat_dim=cfg["feat_dim"],outlier_nb_points=300,feat_dbscan_min=20, merge_type=1))
S,ref_pts,ref_cols=PC.check_map(sc,frames,cfg)
ref_feats,_=PC.check_fuse(sc,frames,S,cfg,ref_pts,ref_cols,check_masks=False)
orig=O.feats_denoise_dbscan
O.feats_denoise_dbscan=lambda f,eps=0.01,min_points=100: orig(f,eps=0.01,min_points=20)
got,feats=PC.check_merge_pool(sc,frames,cfg,ref_pts,ref_feats)
print("pool parity ok", len(got), max(len(g) for g in got))
