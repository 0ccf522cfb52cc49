#!/usr/bin/env python
"""CPU estimate of what the chain pass's deal by predicted depth leaves on the table.
A config-2-like cloud (same 64 clusters, plain Morton octree with the reference's capacity instead of the exact chain
arithmetic: 5 957 nodes against the real 6 073, mean leaf depth 6.56 against 6.53), the leaf depth of the first 1 M points in
input order, and for tiles of 512 / 1 024 / 2 048 points the level steps a wave executes (the deepest of its 64 points)
over the level steps its points need, when the tile is dealt by (a) the classes the 128^3 depth grid can tell apart
(depth <= 7 exact, deeper lumped), (b) the exact depth, (c) not at all.   usage: python tools/deal_sim.py [points]
Result at 100 M points: tile 1 024: 1.076 (grid classes) / 1.038 (exact depth) / 1.48 (input order); tile 512: 1.094 /
1.057; tile 2 048: 1.061 / 1.018 — a perfect depth prediction would save 3.5 % of the executed level steps."""
import numpy as np, sys, time
n=int(float(sys.argv[1])) if len(sys.argv)>1 else 100_000_000
cap=100_000; LV=13
rng=np.random.Generator(np.random.PCG64(12345))
centres=rng.uniform(0,1000,(64,3)); sig=rng.uniform(1,20,64)
g=np.random.default_rng(7)
t0=time.time()
keys=np.empty(n,np.uint64)
mins=np.full(3,np.inf); maxs=np.full(3,-np.inf)
pts=[]
CH=1<<24
for s in range(0,n,CH):
    m=min(CH,n-s)
    w=g.integers(0,64,m)
    p=g.standard_normal((m,3))*sig[w,None]+centres[w]
    pts.append(p.astype(np.float64))
    mins=np.minimum(mins,p.min(0)); maxs=np.maximum(maxs,p.max(0))
edge=(maxs-mins).max()
print('gen',time.time()-t0, 'edge',edge)
def spread(v):
    v=v.astype(np.uint64)&np.uint64(0x1fff)
    out=np.zeros_like(v)
    for b in range(LV):
        out|=((v>>np.uint64(b))&np.uint64(1))<<np.uint64(3*b)
    return out
off=0
for p in pts:
    q=np.clip(((p-mins)/edge*(1<<LV)).astype(np.int64),0,(1<<LV)-1)
    k=(spread(q[:,0])<<np.uint64(2))|(spread(q[:,1])<<np.uint64(1))|spread(q[:,2])
    keys[off:off+len(p)]=k; off+=len(p)
del pts
first=keys[:1<<20].copy()   # input-order sample
print('keys',time.time()-t0)
sk=np.sort(keys); del keys
print('sort',time.time()-t0)
# top-down split: nodes as (level,prefix); leaf depth lookup via dict per level
leaves=[set() for _ in range(LV+1)]
front=[(0,0)]
nn=0
while front:
    nxt=[]
    for (l,pre) in front:
        sh=np.uint64(3*(LV-l))
        lo=np.searchsorted(sk,np.uint64(pre)<<sh,'left'); hi=np.searchsorted(sk,(np.uint64(pre)+np.uint64(1))<<sh,'left') if l>0 else n
        c=hi-lo
        if c==0: continue
        nn+=1
        if (l==0 or c>cap) and l<LV:
            for d in range(8): nxt.append((l+1,pre*8+d))
        else: leaves[l].add(pre)
    front=nxt
print('nodes',nn,'tree',time.time()-t0)
depth=np.zeros(len(first),np.int32)
for l in range(LV+1):
    if not leaves[l]: continue
    arr=np.array(sorted(leaves[l]),dtype=np.uint64)
    pre=first>>np.uint64(3*(LV-l))
    idx=np.searchsorted(arr,pre); idx[idx>=len(arr)]=len(arr)-1
    hit=arr[idx]==pre
    depth[hit]=l
print('depth hist',np.bincount(depth), 'mean',depth.mean())
def sim(depth,tile,classfn,groups_of=64):
    tot_exec=0; tot_need=depth.sum()
    for s in range(0,len(depth),tile):
        d=depth[s:s+tile]
        o=np.argsort(-classfn(d),kind='stable')
        dd=d[o]
        gmax=dd.reshape(-1,groups_of).max(1)
        tot_exec+=gmax.sum()*groups_of
    return tot_exec/tot_need
for tile in (512,1024,2048):
    print('tile',tile,'grid7 classes',sim(depth,tile,lambda d:np.minimum(d,8)),'exact classes',sim(depth,tile,lambda d:d), 'no deal',sim(depth,tile,lambda d:np.zeros_like(d)))
