"""Development aid: random clouds through the DBSCAN test hook (plain and with the anchor hint) on the kernel simulator
against the oracle.  python scripts/fuzz/fuzz_dbscan.py <seed> <seconds>"""
import numpy as np, sys, time, ctypes as C
sys.path.insert(0, __import__('os').path.abspath(__import__('os').path.join(__import__('os').path.dirname(__file__), '..', '..')))
from tests import parity_common as PC
from holoagent_amd._lib import HmsgLib
from oracle import hmsg_oracle as O
L=HmsgLib(PC.EMU_PATH)
def run(clouds, eps, mp, core0=None):
    K=len(clouds); sizes=np.array([len(c) for c in clouds],np.int64); N=int(sizes.sum())
    pts=np.ascontiguousarray(np.concatenate(clouds) if N else np.zeros((0,3)))
    outp=np.zeros((max(N,1),3)); outs=np.zeros(K,np.int64); outc=np.zeros(max(N,1),np.uint8); info=np.zeros((K,3),np.int32)
    c0=None if core0 is None else np.ascontiguousarray(core0,np.uint8)
    rc=L.c.hmsg_test_dbscan(pts.ctypes.data,K,sizes.ctypes.data,eps,mp,None if c0 is None else c0.ctypes.data,outp.ctypes.data,outs.ctypes.data,outc.ctypes.data,info.ctypes.data)
    assert rc==0
    off=np.concatenate([[0],np.cumsum(outs)])
    return [outp[off[k]:off[k+1]] for k in range(K)],[outc[off[k]:off[k+1]] for k in range(K)],info
rng=np.random.default_rng(int(sys.argv[1])); T=float(sys.argv[2]); t0=time.time(); n=0; na=0
def cloud(rng, n, kind):
    if kind==0: return rng.uniform(0,rng.uniform(0.3,2.0),(n,3))
    if kind==1:  # planar patches with duplicates (heavily re-observed surface)
        p=np.zeros((n,3)); p[:,0]=rng.uniform(0,1.5,n); p[:,1]=rng.uniform(0,1.0,n); p[:,2]=rng.normal(0,0.004,n)
        q=np.round(p/0.05)*0.05; m=rng.random(n)<0.5; p[m]=q[m]; return p
    # blobs
    c=rng.uniform(0,2,(int(rng.integers(1,5)),3)); return c[rng.integers(0,len(c),n)]+rng.normal(0,rng.uniform(0.02,0.15),(n,3))
while time.time()-t0<T:
    eps=float(rng.choice([0.1,0.05,0.08])); mp=int(rng.choice([10,5,3]))
    K=int(rng.integers(1,5))
    clouds=[cloud(rng,int(rng.integers(0,1500)),int(rng.integers(0,3))) for _ in range(K)]
    got,cores,info=run(clouds,eps,mp)
    for k,c in enumerate(clouds):
        want,_=O.pcd_denoise_dbscan(c,None,eps,mp)
        # the kernel drops everything when no cluster / fewer than 5? mirror the oracle: it returns the input then
        if not np.array_equal(got[k],want):
            np.savez('/tmp/fuzz_db_fail.npz',c=c,eps=eps,mp=mp); print('MISMATCH plain',len(c),len(got[k]),len(want),eps,mp,info[k]); sys.exit(1)
    n+=1
    # anchor path: A' = fixed single-cluster result, B new points near it
    A=got[int(np.argmax([len(g) for g in got]))]
    if len(A)<50: continue
    g2,c2,i2=run([A],eps,mp)
    if i2[0,0]!=0 or i2[0,1]!=1: continue
    flags=c2[0]
    nb=int(rng.integers(1,400))
    B=A[rng.integers(0,len(A),nb)]+rng.normal(0,rng.uniform(0.01,0.2),(nb,3))
    if rng.random()<0.3: B=np.concatenate([B, cloud(rng,int(rng.integers(5,200)),2)+rng.uniform(-1,1,3)])
    cat=np.concatenate([A,B]); c0=np.concatenate([flags,np.zeros(len(B),np.uint8)])
    g3,c3,i3=run([cat, clouds[0]],eps,mp,core0=np.concatenate([c0,np.zeros(len(clouds[0]),np.uint8)]))
    want,_=O.pcd_denoise_dbscan(cat,None,eps,mp)
    g4,c4,i4=run([cat],eps,mp)      # no hint
    if not (np.array_equal(g3[0],want) and np.array_equal(g4[0],want) and np.array_equal(c3[0],c4[0]) and (i3[0]==i4[0]).all()):
        np.savez('/tmp/fuzz_db_fail2.npz',A=A,B=B,eps=eps,mp=mp,flags=flags); print('MISMATCH anchor',len(A),len(B),len(g3[0]),len(g4[0]),len(want),i3[0],i4[0]); sys.exit(1)
    na+=1
print('ok',n,na)
