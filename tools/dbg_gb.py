import numpy as np, sys
sys.path.insert(0,'.')
sys.path.insert(0,'tests')
from rust_dataframe_amd import _abi as A, lib
from oracle import oracle
from util import make_chunks
api=lib.api(); ora=oracle.api()
def groups(ok,ov,oc):
    k=ok[0].to_pylist(); v=ov.to_pylist(); c=oc.to_numpy().tolist()
    return {k[i]:(v[i],c[i]) for i in range(oc.length)}
bad=0
for seed in range(40):
    rng=np.random.default_rng(seed)
    lens=[700,0,3000]; nf=0.15; off=13; ngroups=200
    keys=[A.HostArray.from_numpy(rng.integers(0,ngroups,n).astype(np.int64), valid=(rng.uniform(size=n)>=nf), offset=off, rng=rng) for n in lens]
    vals=make_chunks(rng, A.F64, lens, nf, off, kind="special")
    for agg in ("sum","min"):
        for mg in (208, 600, 1200, 2040):
            e=groups(*ora.groupby_agg([keys],vals,agg,mg)); g=groups(*api.groupby_agg([keys],vals,agg,mg))
            d=[(k,g.get(k),e[k]) for k in e if g.get(k,(None,None))[1]!=e[k][1]]
            if d:
                bad+=1
                print("seed",seed,agg,"mg",mg,lib.last_kernel(),"tot",sum(c for _,c in g.values()),sum(c for _,c in e.values()),d[:3])
print("bad",bad)
