import random, itertools
NF=11
pairs=[(i,j) for i in range(NF) for j in range(i+1)]
def cost(assign):  # assign: list of 66 pairs in bx order
    tot=0
    for x in range(8):
        fr=set()
        for bx in range(x,66,8):
            fr.update(assign[bx])
        tot+=len(fr)
    return tot
cur=[None]*66
for bx in range(66):
    fi=0
    while (fi+1)*(fi+2)//2<=bx: fi+=1
    cur[bx]=(fi,bx-fi*(fi+1)//2)
print("current", cost(cur))
best=None
random.seed(1)
for trial in range(40):
    a=pairs[:]; random.shuffle(a); c=cost(a)
    T=1.0
    for it in range(200000):
        i,j=random.randrange(66),random.randrange(66)
        if i%8==j%8: continue
        a[i],a[j]=a[j],a[i]; c2=cost(a)
        if c2<=c or random.random()<pow(2.718,-(c2-c)/T): c=c2
        else: a[i],a[j]=a[j],a[i]
        T=max(0.05,T*0.99997)
    if best is None or c<best[0]: best=(c,a[:]); print(trial,c)
c,a=best
print(c)
for x in range(8): print(x, sorted(set(f for bx in range(x,66,8) for f in a[bx])), [a[bx] for bx in range(x,66,8)])
print("TAB", [p[0]*16+p[1] for p in a])
