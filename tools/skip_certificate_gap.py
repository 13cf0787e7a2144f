import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import scenes
from scipy.spatial import cKDTree
from oracle import orc
cfg=int(sys.argv[1]) if len(sys.argv)>1 else 2
sc=scenes.make_scene(cfg=cfg)
o=orc.Oracle(sc["params"],threads=16,use_ref=False)
o.map_build(sc["map"]); o.scan_set(sc["scan"],sc["tables"],sc["temporal_comp"])
states=[]
o.set_pass_hook(lambda k: None)
# states of the passes: run update with hook capturing? use h_share_model at state0 then update to get posterior
r=o.h_share_model(sc["state0"],True); w0=o.scan_get()["world"].copy()
v=o.update_iterated(sc["state0"],sc["P0"])
r=o.h_share_model(v["state"],True); w1=o.scan_get()["world"].copy()
tree=cKDTree(sc["map"][:,:3].astype(np.float64))
d,idx=tree.query(w0.astype(np.float64),k=17)
delta=np.linalg.norm(w1-w0,axis=1)
print("passes",v["passes"],"searches",v["searches"],"delta median %.3f p90 %.3f max %.3f"%(np.median(delta),np.percentile(delta,90),delta.max()))
print("d5 median %.3f ; gap d6-d5 median %.3f p10 %.3f ; d9-d5 median %.3f p10 %.3f; d17-d5 median %.3f p10 %.3f"%(np.median(d[:,4]),np.median(d[:,5]-d[:,4]),np.percentile(d[:,5]-d[:,4],10),np.median(d[:,8]-d[:,4]),np.percentile(d[:,8]-d[:,4],10),np.median(d[:,16]-d[:,4]),np.percentile(d[:,16]-d[:,4],10)))
mp=sc["map"][:,:3].astype(np.float64)
for K in (5,6,8,10,12,16):
    # cached K; outsiders bound L=d[K]; new distances of cached
    nd=np.linalg.norm(mp[idx[:,:K]]-w1[:,None,:].astype(np.float64),axis=2)
    nd.sort(axis=1)
    ok=(nd[:,4] < d[:,K]-delta) & (d[:,4]<np.sqrt(5))
    print("K=%2d kept %.4f ; P(WG of 64 all kept) ~ %.3f"%(K, ok.mean(), ok.mean()**64))
for scale in (0.1,0.3):
    w1s=w0+(w1-w0)*scale; dl=delta*scale
    for K in (5,8):
        nd=np.linalg.norm(mp[idx[:,:K]]-w1s[:,None,:].astype(np.float64),axis=2); nd.sort(axis=1)
        ok=(nd[:,4] < d[:,K]-dl)
        print("motion x%.1f (median %.3f m) K=%d kept %.4f"%(scale,np.median(dl),K,ok.mean()))
