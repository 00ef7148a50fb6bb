"""BVH builders on hostile input (GPU box): NaN / infinite / huge / denormal / coincident coordinates — the GPU build must
equal the sequential host build (NaN payloads aside) or refuse the input; neither may hang.  python tools/bvh_nan_probe.py"""
import importlib, sys, os
import numpy as np
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path.insert(0,ROOT)
rpt = importlib.import_module('rust-path-tracer_amd'); hip = importlib.import_module('rust-path-tracer_amd.hip'); host = importlib.import_module('rust-path-tracer_amd.host')
ffi = importlib.import_module('rust-path-tracer_amd._ffi')
def soup(n, rng, poison):
    v = rng.normal(size=(n*3,3)).astype(np.float32)
    poison(v, rng)
    v = np.concatenate([v, np.ones((len(v),1),np.float32)],1)
    t = np.zeros(n, ffi.TRIANGLE_DTYPE); idx=np.arange(n*3,dtype=np.uint32).reshape(n,3); nm=t.dtype.names
    t[nm[0]],t[nm[1]],t[nm[2]] = idx[:,0],idx[:,1],idx[:,2]
    return v,t
def same(a,b):
    if len(a)!=len(b): return False
    for f in a.dtype.names:
        x,y=a[f],b[f]
        if x.dtype.kind=='f':
            x=x.view(np.uint32); y=y.view(np.uint32)
            # NaN payloads may differ: compare NaN-ness
            xf=a[f]; yf=b[f]
            nx,ny=np.isnan(xf),np.isnan(yf)
            if not (np.array_equal(nx,ny) and np.array_equal(x[~nx],y[~ny])): return False
        elif not np.array_equal(x,y): return False
    return True
cases = {
 'nan coords 1%': lambda v,r: v.__setitem__(r.random(v.shape)<0.01, np.nan),
 'inf coords 1%': lambda v,r: v.__setitem__(r.random(v.shape)<0.01, np.inf),
 '-inf / inf mix': lambda v,r: (v.__setitem__(r.random(v.shape)<0.005, np.inf), v.__setitem__(r.random(v.shape)<0.005, -np.inf)),
 'all identical': lambda v,r: v.__setitem__(slice(None), 1.5),
 'huge 1e38': lambda v,r: v.__imul__(np.float32(1e38)),
 'denormal': lambda v,r: v.__imul__(np.float32(1e-42)),
 'one nan triangle': lambda v,r: v.__setitem__(slice(0,3), np.nan),
}
for n in (500, 5000):
  for name,p in cases.items():
    rng=np.random.default_rng(3)
    v,t = soup(n, rng, p)
    try:
        hn,ht = host.bvh_build(v,t.copy())
        gn,gt,_ = hip.bvh_build_gpu(v,t.copy())
        print(n, name, 'nodes', len(hn), len(gn), 'same' if same(hn,gn) and np.array_equal(ht,gt) else 'DIFFERENT')
    except Exception as e:
        print(n, name, 'EXC', repr(e)[:200])
