#!/usr/bin/env python3
"""Exhaustive search for a bank-conflict-free 16-byte-slot swizzle of the conv LDS images
(ds_read_b128 lane groups and the 16x16x32 MFMA fragment map; see csrc/xv_common.h xv_swz)."""
import itertools
groups=[list(range(0,4))+list(range(12,16))+list(range(20,28)),
        list(range(4,12))+list(range(16,20))+list(range(28,32)),
        list(range(32,36))+list(range(44,48))+list(range(52,60)),
        list(range(36,44))+list(range(48,52))+list(range(60,64))]
def conflicts(f, b, kk):
    # returns total extra cycles over the 4 groups
    extra=0
    for g in groups:
        seen={}
        for l in g:
            l15=l&15; lg=l>>4
            hx=b+l15
            gran=((hx&1)<<3)|((kk*4+lg)^f(hx))
            seen[gran]=seen.get(gran,0)+1
        extra+=max(seen.values())-1
    return extra
def score(f):
    tot=0
    for b in range(0,20):
        for kk in (0,1):
            tot+=conflicts(f,b,kk)
    return tot
cur=lambda hx:(hx>>1)&7
print('current', score(cur), [sum(conflicts(cur,b,kk) for kk in (0,1)) for b in range(4)])
best=[]
# linear maps of low 5 bits of hx to 3 bits
for m in itertools.product(range(32), repeat=3):
    def f(hx, m=m):
        v=0
        for i,row in enumerate(m):
            v|=(bin(row & hx & 31).count('1')&1)<<i
        return v
    s=score(f)
    best.append((s,m))
best.sort()
print(best[:10])
