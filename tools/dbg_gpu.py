import ctypes as C, sys
sys.path.insert(0,'/root/repo')
from oracle import pyref as R
L=C.CDLL('/root/repo/go-ibft_amd/csrc/libibft_devtest.so')
L.devtest_run.argtypes=[C.c_int,C.c_int,C.c_char_p,C.c_char_p,C.c_char_p,C.c_int]
def run(op,xs,ys=None):
    n=len(xs); ys=ys or [0]*n
    a=b"".join(x.to_bytes(32,'big') for x in xs); b=b"".join(y.to_bytes(32,'big') for y in ys)
    out=C.create_string_buffer(32*n); assert L.devtest_run(op,n,a,b,out,32*n)==0
    return [int.from_bytes(out.raw[32*i:32*i+32],'big') for i in range(n)]
ks=[0,1,2,3,7,8,9,16,40,248]
exp=[R.pt_mul(1<<k,R.G)[0] for k in ks]
print("fermat ",[a==b for a,b in zip(run(12,ks),exp)])
print("safegcd",[a==b for a,b in zip(run(13,ks),exp)])
we=[(0,1),(0,2),(1,1),(1,2),(2,1),(5,77),(31,255)]
g=run(14,[w for w,e in we],[e for w,e in we])
print("gtab   ",[a==R.pt_mul(e<<(8*w),R.G)[0] for a,(w,e) in zip(g,we)])
# uniform lanes (all same k) to see if divergence matters
print("uniform k=8 fermat",set(a==exp[5] for a in run(12,[8]*64)), "safegcd", set(a==exp[5] for a in run(13,[8]*64)))
