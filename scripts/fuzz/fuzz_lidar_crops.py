"""Development aid: random inputs through hmsg_lidar_depth (uvz mode) and hmsg_crop_resize_batch on the kernel simulator
against the oracles.  python scripts/fuzz/fuzz_lidar_crops.py <seed> <seconds>"""
import numpy as np, sys, time
sys.path.insert(0, __import__('os').path.abspath(__import__('os').path.join(__import__('os').path.dirname(__file__), '..', '..')))
from tests import parity_common as PC
from holoagent_amd._lib import HmsgLib, lidar_depth, crop_all_bounding_boxs, HmsgError
from oracle import lidar_depth_oracle as LO, crop_oracle as CO
L=HmsgLib(PC.EMU_PATH)
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
t0=time.time(); n3=n4=0
while time.time()-t0 < float(sys.argv[2]) if len(sys.argv)>2 else 120:
    # ---- N3: uvz mode with adversarial values
    W,H=int(rng.integers(4,70)),int(rng.integers(4,50))
    n=int(rng.integers(0,4000))
    u=rng.uniform(-2,W+2,n); v=rng.uniform(-2,H+2,n); z=rng.uniform(-0.5,30,n)
    # quantise some to create ties / pixel-centre cases / identical z in float32
    m=rng.random(n)<0.3; u[m]=np.round(u[m]*2)/2; v[m]=np.round(v[m]*2)/2
    m=rng.random(n)<0.3; z[m]=np.float32(z[m])*(1+rng.integers(-2,3,m.sum())*1e-8)
    m=rng.random(n)<0.1; z[m]=rng.choice([0.0,1.0,2.5,20.0,6.6667],m.sum())
    uvz=np.stack([u,v,z],1)
    scale=int(rng.choice([1,2,4,5]))
    d,st,state,_=lidar_depth([uvz],None,np.eye(3),W,H,voxel_size=0,image_scale=scale,want_state=True,lib_=L)
    pi=np.stack([u,v,np.ones(n)]); pc=np.stack([u,v,z])
    want,flags=LO.occ_depth(pi,pc,W,H,1000,scale)
    if not (np.array_equal(d[0],want) and np.array_equal(state==1,flags)):
        np.savez('/tmp/fuzz_fail_n3.npz',uvz=uvz,W=W,H=H,scale=scale); print('N3 MISMATCH',W,H,n,scale); sys.exit(1)
    n3+=1
    # ---- N4
    H2,W2=int(rng.integers(2,60)),int(rng.integers(2,80))
    img=rng.integers(0,256,(H2,W2,3),dtype=np.uint8)
    M=int(rng.integers(1,5)); masks=[]
    for _ in range(M):
        w,h=int(rng.integers(1,W2+1)),int(rng.integers(1,H2+1)); x,y=int(rng.integers(0,W2-w+1)),int(rng.integers(0,H2-h+1))
        seg=rng.random((H2,W2))<0.5
        masks.append(dict(segmentation=seg,bbox=[x,y,w,h]))
    S=int(rng.choice([4,8,12,32,64])); margin=int(rng.integers(0,20))
    p,mk=crop_all_bounding_boxs(img,masks,margin,size=S,lib_=L)
    for k,mm in enumerate(masks):
        a=CO.resize_linear_u8(CO.crop_bbox(img,mm['bbox'],margin),(S,S)); b=CO.resize_linear_u8(CO.crop_image(img,mm),(S,S))
        if not (np.array_equal(p[k],a) and np.array_equal(mk[k],b)):
            np.savez('/tmp/fuzz_fail_n4.npz',img=img,bbox=np.array(mm['bbox']),seg=mm['segmentation'],S=S,margin=margin); print('N4 MISMATCH',H2,W2,mm['bbox'],S,margin); sys.exit(1)
    n4+=1
print('ok',n3,n4)
