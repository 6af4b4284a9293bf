"""LDS bank-conflict checker for the ntt1.hip layouts (lane-group rules of MI355X_MICROARCH.md, section LDS): development tool."""
# LDS bank-conflict checker for the ntt1 layouts (rules: MI355X_MICROARCH.md section LDS)
import itertools
def groups(kind):
    if kind=='r64': return [list(range(0,32)),list(range(32,64))], 8, 32
    if kind=='w64': return [list(range(16*i,16*i+16)) for i in range(4)], 8, 16
    if kind=='r128':
        g0=[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27]
        g1=[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]
        return [g0,g1,[x+32 for x in g0],[x+32 for x in g1]], 16, 16
    if kind=='w128': return [list(range(8*i,8*i+8)) for i in range(8)], 16, 8
def worst(kind, addr_of_lane):  # addr in bytes
    gs, unit, nb = groups(kind)
    w=1
    for g in gs:
        cnt={}
        for l in g:
            b=(addr_of_lane(l)//unit)%nb
            cnt.setdefault(b,set()).add(addr_of_lane(l)//unit)
        w=max(w,max(len(v) for v in cnt.values()))
    return w
def sw(j, variant):
    if variant==0: return j ^ (((j>>6)&7)<<2)
    if variant==1: return j ^ (((j>>6)&7)<<2) ^ (((j>>5)&1)<<1)
    if variant==2: return j ^ (((j>>6)&7)<<2) ^ (((j>>9)&1)<<1) ^ (((j>>5)&1)<<1)
for v in (0,1):
    res={}
    res['B r64']=max(worst('r64',lambda l,reg=reg: 8*sw(64*reg+l,v)) for reg in range(16))
    res['B w64']=max(worst('w64',lambda l,reg=reg: 8*sw(64*reg+l,v)) for reg in range(16))
    res['C r64']=max(worst('r64',lambda l,reg=reg: 8*sw(64*(l>>2)+4*reg+(l&3),v)) for reg in range(16))
    res['C w64']=max(worst('w64',lambda l,reg=reg: 8*sw(64*(l>>2)+4*reg+(l&3),v)) for reg in range(16))
    res['D r128']=max(worst('r128',lambda l,g=g,h=h: 8*sw(256*g+4*l+2*h,v)) for g in range(4) for h in range(2))
    res['D w128']=max(worst('w128',lambda l,g=g,h=h: 8*sw(256*g+4*l+2*h,v)) for g in range(4) for h in range(2))
    res['D r64']=max(worst('r64',lambda l,g=g,e=e: 8*sw(256*g+4*l+e,v)) for g in range(4) for e in range(4))
    res['D w64']=max(worst('w64',lambda l,g=g,e=e: 8*sw(256*g+4*l+e,v)) for g in range(4) for e in range(4))
    res['E r128 (linear 16B per lane)']=max(worst('r128',lambda l,i=i: 8*sw(128*i+2*l,v)) for i in range(8))
    res['E w128']=max(worst('w128',lambda l,i=i: 8*sw(128*i+2*l,v)) for i in range(8))
    print('variant',v,res)
print("search")
import itertools
def mk(src3, b1):  # XOR bits[4:2] with (j>>src3)&7, bit1 with (j>>b1)&1
    return lambda j: j ^ (((j>>src3)&7)<<2) ^ ((((j>>b1)&1)<<1) if b1 is not None else 0)
best=[]
for src3 in (5,6,7):
    for b1 in (None,4,5,6,7,8,9):
        f=mk(src3,b1)
        if len(set(f(j) for j in range(1024)))!=1024: continue
        res={}
        res['Br']=max(worst('r64',lambda l,reg=reg: 8*f(64*reg+l)) for reg in range(16))
        res['Bw']=max(worst('w64',lambda l,reg=reg: 8*f(64*reg+l)) for reg in range(16))
        res['Cr']=max(worst('r64',lambda l,reg=reg: 8*f(64*(l>>2)+4*reg+(l&3))) for reg in range(16))
        res['Cw']=max(worst('w64',lambda l,reg=reg: 8*f(64*(l>>2)+4*reg+(l&3))) for reg in range(16))
        res['Dr128']=max(worst('r128',lambda l,g=g,h=h: 8*f(256*g+4*l+2*h)) for g in range(4) for h in range(2))
        res['Dw128']=max(worst('w128',lambda l,g=g,h=h: 8*f(256*g+4*l+2*h)) for g in range(4) for h in range(2))
        res['Er']=max(worst('r128',lambda l,i=i: 8*f(128*i+2*l)) for i in range(8))
        res['Ew']=max(worst('w128',lambda l,i=i: 8*f(128*i+2*l)) for i in range(8))
        print(src3,b1,res, sum(res.values()))
print("additive search"); cand=[]
def mk2(a,b,c,d):
    return lambda j: j + a*(j>>6) + b*((j>>5)&1) + c*((j>>4)&1) + d*((j>>3)&1)
import itertools
for a,b,c,d in itertools.product((2,4,6,8,12),(0,2,4,6),(0,2,4),(0,2)):
    f=mk2(a,b,c,d)
    img=[f(j) for j in range(1024)]
    if len(set(img))!=1024: continue
    if any(f(j)+1!=f(j+1) for j in range(0,1024,2)): continue   # 16-byte pairs stay together
    if any(f(j)%2 for j in range(0,1024,2)): continue
    res={}
    res['Br']=max(worst('r64',lambda l,reg=reg: 8*f(64*reg+l)) for reg in range(16))
    res['Bw']=max(worst('w64',lambda l,reg=reg: 8*f(64*reg+l)) for reg in range(16))
    res['Cr']=max(worst('r64',lambda l,reg=reg: 8*f(64*(l>>2)+4*reg+(l&3))) for reg in range(16))
    res['Cw']=max(worst('w64',lambda l,reg=reg: 8*f(64*(l>>2)+4*reg+(l&3))) for reg in range(16))
    res['Dr128']=max(worst('r128',lambda l,g=g,h=h: 8*f(256*g+4*l+2*h)) for g in range(4) for h in range(2))
    res['Dw128']=max(worst('w128',lambda l,g=g,h=h: 8*f(256*g+4*l+2*h)) for g in range(4) for h in range(2))
    res['Er']=max(worst('r128',lambda l,i=i: 8*f(128*i+2*l)) for i in range(8))
    # radix-8 variants: C1: j = 64h + 8r + 4it + low (lane=(h,low2)), C2: 8 consecutive per lane: j = 8*(lane+64it) + 2q (b128)
    res['C1r']=max(worst('r64',lambda l,reg=reg,it=it: 8*f(64*(l>>2)+8*reg+4*it+(l&3))) for reg in range(8) for it in range(2))
    res['C2r128']=max(worst('r128',lambda l,it=it,q=q: 8*f(8*(l+64*it)+2*q)) for it in range(2) for q in range(4))
    s=sum(res.values())
    cand.append((s,a,b,c,d,res,max(img)+1))
cand.sort(key=lambda x:x[0])
for c in cand[:12]: print(c)
