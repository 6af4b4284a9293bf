import itertools
def run(R, E, EXIT, last=False):
    n=1<<R; bd=[E]*n; red=0
    for st in range(R):
        dist=1<<st
        for blk in range(n//(2*dist)):
            for k in range(dist):
                x=blk*2*dist+k; y=x+dist
                if bd[x]+bd[y]>64:
                    for r in (x,y):
                        if bd[r]>4: bd[r]=4; red+=1
                    assert bd[x]+bd[y]<=64
                if last and st==R-1:
                    bd[x]=3; bd[y]=3
                else:
                    bd[x]=bd[x]+bd[y]; bd[y]=3
    for r in range(n):
        if bd[r]>EXIT: bd[r]=4; red+=1
    return max(bd), red, bd
best=None
for xd,xc,xb in itertools.product([8,12,16,24,32,64],repeat=3):
    m,r1,_=run(3,2,xd); 
    m2,r2,_=run(3,m,xc)
    m3,r3,_=run(4,m2,xb)
    m4,r4,_=run(5,m3,64,True)
    tot=r1*4+r2*4+r3*2+r4   # per 32 values: D' 4 groups of 8, C' 4 groups, B' 2 groups of 16, A' 1 group of 32
    if best is None or tot<best[0]: best=(tot,(xd,xc,xb),(m,m2,m3),(r1,r2,r3,r4))
print(best)
for E0 in (1,2):
  for xd,xc,xb in [best[1]]:
    m,r1,b1=run(3,E0,xd); m2,r2,b2=run(3,m,xc); m3,r3,b3=run(4,m2,xb); m4,r4,b4=run(5,m3,64,True)
    print(E0,m,r1,b1,m2,r2,b2,m3,r3,b3,r4)
