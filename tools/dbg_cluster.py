import sys, ctypes as C, numpy as np, collections
sys.path.insert(0,'/root/repo')
import matrix_fhe_lattigo_amd as rh
ring=rh.Ring(1<<13,[0x1fffffffffe00001, 0x1fffffffffc80001])
lib=rh.lib()
cap=512
buf=np.zeros(8+8*cap,dtype=np.uint32)
lib.rh_debug_cluster_dryrun.argtypes=[C.c_void_p,C.c_int,C.c_int,C.c_void_p,C.c_uint]
rc=lib.rh_debug_cluster_dryrun(ring._h,5,2,buf.ctypes.data_as(C.c_void_p),cap)
print("rc",rc,"tickets",buf[0])
rec=buf[8:8+8*min(int(buf[0]),cap)].reshape(-1,8)
raws=collections.Counter(int(r[0]) for r in rec)
print("raw XCC_ID register values:", {hex(k):v for k,v in raws.items()})
rows=collections.defaultdict(list)
for r in rec: rows[int(r[2])].append((int(r[3]), int(r[0])&7, int(r[6])))
for k in sorted(rows): print("row",k,"units",sorted(u for u,_,_ in rows[k]),"xcc",set(x for _,x,_ in rows[k]), "blocks%8", set(b%8 for _,_,b in rows[k]))
print("max row", max(rows), "poly max", rec[:,4].max(), "limb max", rec[:,5].max())
