"""Development aid: random small synthetic scenes through map / fusion / merge / pooling on the kernel simulator against
the oracle (tests/parity_common.py checks).  A failure of the statistical fp16 bound of the per-voxel features with very
few masks is reported as a tolerance case.  python scripts/fuzz/fuzz_pipeline.py <seed> <seconds>"""
import numpy as np, sys, time
sys.path.insert(0, __import__('os').path.abspath(__import__('os').path.join(__import__('os').path.dirname(__file__), '..', '..')))
from tests import parity_common as PC
from holoagent_amd._lib import HmsgLib
from holoagent_amd.synth import SceneSpec, SynthScene
import oracle.hmsg_oracle as O
L=HmsgLib(PC.EMU_PATH)
seed=int(sys.argv[1]); T=float(sys.argv[2]); t0=time.time(); n=0
rng=np.random.default_rng(seed)
orig=O.feats_denoise_dbscan
O.feats_denoise_dbscan=lambda f, eps=0.01, min_points=100: orig(f, eps=0.01, min_points=20)
while time.time()-t0<T:
    s=int(rng.integers(0,10**6))
    nf=int(rng.integers(3,9)); M=int(rng.integers(1,12)); D=int(rng.choice([8,16,24,40]))
    W,H=int(rng.choice([48,64,80])),int(rng.choice([36,48,60]))
    spec=SceneSpec(seed=s, rooms_x=1, rooms_z=1, room_size=(float(rng.uniform(2.5,4.5)),2.5,float(rng.uniform(2.5,4))), objects_per_room=int(rng.integers(2,7)), width=W, height=H, n_frames=nf, n_masks=M, feat_dim=D, yaw_step_deg=float(rng.uniform(10,40)))
    scn=SynthScene(spec); frames=[scn.frame(i) for i in range(nf)]
    mt=str(rng.choice(["sequential","hierarchical"]))
    nb=int(rng.choice([50,200]))
    cfg=dict(voxel_size=0.05, clip_masked_weight=0.4418, max_mask_distance=10000, feat_dim=D, outlier_nb=nb, init_overlap_thresh=0.75, overlap_thresh_factor=0.025, iou_thresh=0.05, merge_type=mt)
    sc=PC.make_scene(L, frames, dict(feat_dim=D, outlier_nb_points=nb, feat_dbscan_min=20, merge_type=0 if mt=="sequential" else 1))
    try:
        S, ref_pts, ref_cols = PC.check_map(sc, frames, cfg)
        if ref_pts.shape[0]==0: sc.close(); continue
        ref_feats,_=PC.check_fuse(sc, frames, S, cfg, ref_pts, ref_cols)
        got,feats=PC.check_merge_pool(sc, frames, cfg, ref_pts, ref_feats)
    except AssertionError as e:
        import traceback
        tb=traceback.format_exc()
        if '(d > 1e-6).mean() < 1e-3' in tb:
            sc.close(); print('tolerance case', s); continue
        traceback.print_exc(); print('MISMATCH', repr(spec), mt, nb); sys.exit(1)
    sc.close(); n+=1
print('ok',n)
