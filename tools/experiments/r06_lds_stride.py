import itertools
GROUPS=[list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32)),
        [32+x for x in list(range(0,4))+list(range(12,16))+list(range(20,28))],[32+x for x in list(range(4,12))+list(range(16,20))+list(range(28,32))]]
def conflicts(T, off, S, E, budget=352):
    TW=T-off
    tpw=min(64//TW, budget//S)
    tot=0; n=0
    for en in range(1):
        for j in range(0, TW):  # row offset j (4g+jj)
            cyc=0
            for G in GROUPS:
                slots={}
                for lane in G:
                    tw=lane//TW; lit=lane-tw*TW
                    if tw>=tpw: continue
                    nl=TW-lit  # terms
                    g=j//4
                    if not (4*g<nl): continue
                    addr=tw*S+en*E+(lit+off+1+j)
                    slots.setdefault(addr%16,set()).add(addr)
                c=max([len(v) for v in slots.values()] or [0])
                cyc+=max(c,1) if slots else 0
                n+= 1 if slots else 0
            tot+=cyc
    return tot, n, tpw
for T in (17,16,14,12,9):
    for off in (0,1,2,3):
        if T-off<2: continue
        E=T+4
        base=conflicts(T,off,4*E,E)
        best=min(((conflicts(T,off,S,E),S) for S in range(4*E, 4*E+17) if (64//(T-off))*S<=352 or S==4*E), key=lambda x:(x[0][0]/max(1,x[0][1])))
        print(f"T={T} off={off} TW={T-off}: base S={4*E}: cycles {base[0]} over {base[1]} group-accesses (x{base[0]/max(1,base[1]):.2f}) teams {base[2]} | best S={best[1]} x{best[0][0]/max(1,best[0][1]):.2f} teams {best[0][2]}")

print("table")
rows=[]
for T in range(0,18):
    r=[]
    for off in range(4):
        TW=T-off
        E=T+4
        if T<1 or TW<1:
            r.append(4*E if T>=0 else 0); continue
        teams=min(64//TW, 352//(4*E))
        best=None
        for S in range(4*E, 4*E+33):
            if teams*S>352: break
            c=conflicts(T,off,S,E)
            if c[2]!=teams: continue
            key=c[0]/max(1,c[1])
            if best is None or key<best[0]-1e-9: best=(key,S)
        r.append(best[1] if best else 4*E)
    rows.append(r)
print(rows)
# summary of improvement
for T in (17,16,15,14,13,12,10,8):
    for off in (0,1,2):
        E=T+4
        a=conflicts(T,off,4*E,E); b=conflicts(T,off,rows[T][off],E)
        print(T,off,'base x%.2f -> x%.2f (S %d -> %d) teams %d/%d'%(a[0]/max(1,a[1]), b[0]/max(1,b[1]),4*E,rows[T][off],a[2],b[2]))
