"""Bank conflicts of the gather kernel's ds_read_b128 lookups, simulated: four samples share a 16-lane group, a column of
ordinal o occupies the bank class o mod 4 (or o >> 9) of its row part.  Prints the mean LDS cycles per group for the current
slot order and for reordered records.  CPU only: python profiles/sim_lds_conflicts.py"""
import numpy as np
rng=np.random.default_rng(1)
NT=400  # tiles
groups=[[0,3,5,6],[1,2,4,7],[8,11,13,14],[9,10,12,15]]
def cycles(slots_by_sample, cls_fn):
    # slots_by_sample: list of 4 arrays (column lists, already ordered, slot0 excluded) for one LDS lane group; padded with -1 (null)
    L=max(len(a) for a in slots_by_sample)
    tot=0
    for k in range(L):
        cl={}
        for a in slots_by_sample:
            if k<len(a) and a[k]>=0:
                c=cls_fn(a[k]); cl.setdefault(c,set()).add(a[k])
        tot+=max([len(v) for v in cl.values()]+[1])
    return tot, L
def run(order, cls_fn, rotate):
    tot=0; n=0
    for t in range(NT):
        cnt=rng.binomial(2048,0.02/3,64)
        cols=[rng.choice(2048,c,replace=False) for c in cnt]
        idx=np.argsort(cnt,kind='stable')
        for q in range(4):
            recs=[cols[i] for i in idx[16*q:16*q+16]]
            mx=max(len(r) for r in recs)
            nb=(mx+1+3)//4*4  # slots incl slot0
            for g in groups:
                lists=[]
                for s,ri in enumerate(g):
                    r=recs[ri]
                    if order=='sorted': r=np.sort(r)
                    elif order=='mostly':
                        r=np.sort(r); m=rng.random(len(r))<0.18
                        r=np.concatenate([r[~m],rng.permutation(r[m])])
                    if rotate and len(r)>0:
                        sh=(s*len(r))//4
                        r=np.roll(r,-sh)
                    full=np.full(nb-1,-1); full[:len(r)]=r
                    lists.append(full)
                c,L=cycles(lists,cls_fn)
                tot+=c; n+=L
    return tot/n
print('current (random order, ord mod 4):', run('random', lambda o:o%4, False))
print('sorted+rot, class=quartile:', run('sorted', lambda o:o>>9, True))
print('mostly sorted+rot, class=quartile:', run('mostly', lambda o:o>>9, True))
print('sorted no rot, quartile:', run('sorted', lambda o:o>>9, False))

def run_h2():
    tot=0; n=0
    for t in range(NT):
        cnt=rng.binomial(2048,0.02/3,64)
        cols=[rng.choice(2048,c,replace=False) for c in cnt]
        idx=np.argsort(cnt,kind='stable')
        for q in range(4):
            recs=[cols[i] for i in idx[16*q:16*q+16]]
            mx=max(len(r) for r in recs)
            nb=(mx+1+3)//4*4
            for g in groups:
                lists=[]
                for s,ri in enumerate(g):
                    rem={c:[x for x in recs[ri] if x%4==c] for c in range(4)}
                    out=[]
                    for k in range(len(recs[ri])):
                        want=(k+1+s)%4          # slot index k+1 (slot 0 = count)
                        if rem[want]: out.append(rem[want].pop())
                        else:
                            c=max(range(4),key=lambda c:len(rem[c])); out.append(rem[c].pop())
                    full=np.full(nb-1,-1); full[:len(out)]=out
                    lists.append(full)
                c,L=cycles(lists,lambda o:o%4)
                tot+=c; n+=L
    return tot/n
print('H2 greedy class pattern (k+s)%4:', run_h2())

def run_block_local():
    """Reordering only inside each block of four slots: slot 4b + i prefers class (i + s) mod 4."""
    tot=0; n=0
    for t in range(NT):
        cnt=rng.binomial(2048,0.02/3,64)
        cols=[rng.choice(2048,c,replace=False) for c in cnt]
        idx=np.argsort(cnt,kind='stable')
        for q in range(4):
            recs=[cols[i] for i in idx[16*q:16*q+16]]
            mx=max(len(r) for r in recs)
            nb=(mx+1+3)//4*4
            for g in groups:
                lists=[]
                for s,ri in enumerate(g):
                    full=np.full(nb,-1); full[1:1+len(recs[ri])]=recs[ri]     # slot 0 = count
                    out=full.copy()
                    for b in range(nb//4):
                        blk=[x for x in full[4*b:4*b+4]]
                        fixed0 = (b==0)
                        elems=[x for x in (blk[1:] if fixed0 else blk) if x>=0]
                        slots=[i for i in range(4) if not (fixed0 and i==0)]
                        res={i:-1 for i in slots}
                        rest=[]
                        for x in elems:
                            want=[i for i in slots if res[i]<0 and (i+s)%4==x%4]
                            if want: res[want[0]]=x
                            else: rest.append(x)
                        for x in rest:
                            free=[i for i in slots if res[i]<0]; res[free[0]]=x
                        for i in slots: out[4*b+i]=res[i]
                    lists.append(out[1:])
                c,L=cycles(lists,lambda o:o%4)
                tot+=c; n+=L
    return tot/n
print('block-local reorder:', run_block_local())

def run_class_slots():
    """What the compact kernel does: the first four columns of class q in slots q, q+4, q+8, q+12 (class 0: 4, 8, 12, 16), the
    rest from slot 17 on, columns above slot c moved into the holes; sample s reads the slots of a block rotated by s."""
    tot=0; n=0
    for t in range(NT):
        cnt=rng.binomial(2048,0.02/3,64)
        cols=[rng.choice(2048,c,replace=False) for c in cnt]
        idx=np.argsort(cnt,kind='stable')
        for q in range(4):
            recs=[cols[i] for i in idx[16*q:16*q+16]]
            mx=max(len(r) for r in recs)
            nb=(mx+1+3)//4*4
            for g in groups:
                lists=[]
                for s,ri in enumerate(g):
                    slots=np.full(40,-1); ncls=[0,0,0,0]; tail=0
                    for x in recs[ri]:
                        c=x%4
                        if ncls[c]<4:
                            k=4*ncls[c]+(c if c else 4); ncls[c]+=1
                        else:
                            k=17+tail; tail+=1
                        slots[k]=x
                    c=len(recs[ri])
                    holes=[k for k in range(1,c+1) if slots[k]<0]
                    extras=[k for k in range(39,c,-1) if slots[k]>=0]
                    for h,e in zip(holes,extras):
                        slots[h]=slots[e]; slots[e]=-1
                    full=slots[:nb].copy()
                    rot=np.full(nb,-1)
                    for b in range(nb//4):
                        for i in range(4):
                            rot[4*b+i]=full[4*b+((i+s)%4)]
                    lists.append(rot)
                c2,L=cycles(lists,lambda o:o%4)
                tot+=c2; n+=L
    return tot/n
print('class slots + hole filling + rotation (as built):', run_class_slots())
