import numpy as np, sys
a=np.fromfile(sys.argv[1],dtype=np.uint64).reshape(-1,16).astype(np.int64)
w=a.reshape(-1,8,16)
ok=(w[:,:,0]>0).all(axis=1)&(w[:,:,10]>0).all(axis=1)
w=w[ok]
t0=w[:,:,0].min(axis=1,keepdims=True)
for i,n in enumerate(['top','unpack done','fwd done','delay done','bwd done','bar1 passed','refill done','sweep done','-','tail','bar2 passed']):
    rel=w[:,:,i]-t0
    print('%-12s'%n,' '.join('%6d'%np.median(rel[:,k]) for k in range(8)))
