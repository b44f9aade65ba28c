import sys, numpy as np
sys.path.insert(0,'/root/repo')
import matrix_fhe_lattigo_amd as rh
QI60=[0x1fffffffffe00001, 0x1fffffffffc80001, 0x1fffffffffb40001]
for logN,L in [(12,1),(12,3),(13,1),(13,3)]:
    N=1<<logN; mods=QI60[:L]
    ring=rh.Ring(N,mods)
    rng=np.random.default_rng(1)
    B=2
    a=np.stack([np.stack([rng.integers(0,1<<62,size=N,dtype=np.uint64)%np.uint64(q) for q in mods]) for _ in range(B)])
    p=rh.DevicePoly.from_numpy(ring,a); o1=ring.NewPoly(B); o2=ring.NewPoly(B)
    ring.set_tuning("asm_tile",1); ring.NTT(p,o1)
    ring.set_tuning("asm_tile",0); ring.NTT(p,o2)
    x,y=o1.numpy(),o2.numpy()
    bad=np.argwhere(x!=y)
    print(logN,L,"mismatches",len(bad), bad[:8].tolist())
    if len(bad):
        k,i,j=bad[0]; print(hex(int(x[k,i,j])),hex(int(y[k,i,j])), "diff mod q", (int(x[k,i,j])-int(y[k,i,j]))%mods[i])
        js=sorted(set(int(b[2]) for b in bad)); print("first js", js[:20], "count", len(js), "tids(j%256)", sorted(set(j%256 for j in js))[:20])
