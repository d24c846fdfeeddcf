"""Lock-step cost model of the record walk (agh_mwalk.hip) on the real c5_as_worded pattern set: wave-instruction slots
per text byte for the shipped layout (one directory keyed by a piece's first two bytes) and for split directories
(two-byte pieces by bigram, longer pieces by trigram).  A model of the control flow only -- skip / advance / examine
rounds with a wave paying its slowest lane; the costs per step are estimates from the ISA.  Not a test: run by hand,
`python tests/sim_mwalk_lockstep.py [tiles]`; result in DESIGN.md (f).  (Lives under tests/ because it uses the oracle's
corpus generator.)"""
import collections
import os
import random
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _oracle as O
rng = random.Random(1024)
pw=set()
while len(pw)<1024:
    pw.add(bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(4,12))))
pw=sorted(pw)
NT=int(sys.argv[1]) if len(sys.argv)>1 else 4
text,_=O.corpus(NT*16+1, seed=5, variants=tuple(pw[:7]), plant_period=500)   # blocks of 4096
t=text.tobytes()
DEL=10
# entries: (piece bytes, po, other side bytes nearest-first, L, behind?)
ents=[]
for p in pw:
    m=len(p); h=m//2   # split: first piece length? guess: len = m//2 first piece
    a,b=p[:h],p[h:]
    ents.append((a,0,b,len(b),False))          # piece a at j; side b behind
    ents.append((b,h,a[::-1],len(a),True))     # piece b at j; side a in front, nearest first
def side_ok(S,B,L):
    # S,B sequences (nearest first), within one edit as in side_within_one_edit
    S=list(S)+[DEL]*(9-len(S))
    i=0
    while i<L and S[i]==B[i]: i+=1
    if i==L: return True
    if list(B[i+1:L])==S[i:L-1]: return True      # pattern byte missing
    if S[i]==DEL: return False
    if list(B[i+1:L])==S[i+1:L]: return True      # replaced
    return list(B[i:L])==S[i+1:L+1]               # extra text byte
bidir=collections.defaultdict(list); tridir=collections.defaultdict(list)
for e in ents:
    bidir[e[0][:2]].append(e)
def entry_hit(e,j):
    pc,po,B,L,front=e
    if t[j:j+len(pc)]!=pc: return False
    if front: S=t[max(0,j-8):j][::-1]
    else: S=t[j+len(pc):j+len(pc)+8]
    return side_ok(S,B,min(L,7)) if L<=7 else False
def precond(e,j):
    pc,po,B,L,front=e
    if front: S=t[max(0,j-8):j][::-1]
    else: S=t[j+len(pc):j+len(pc)+8]
    S=list(S)+[DEL]*4
    return L<2 or S[0]==B[0] or S[0]==B[1] or S[1]==B[0] or S[1]==B[1]
# slot-level fmask (exact bytes, optimistic): candidate at j under design X
def cand_current(j):
    l=bidir.get(t[j:j+2])
    if not l: return False
    for e in l:
        pc,po,B,L,front=e
        if len(pc)>2:
            if t[j+2]==pc[2]: return True
        else:
            if precond(e,j): return True
    return False
def lists_current(j): return bidir.get(t[j:j+2],[])
# split design: bigram dir holds only 2-byte pieces; trigram dir the rest
bi2=collections.defaultdict(list); tri=collections.defaultdict(list)
for e in ents:
    if len(e[0])==2: bi2[e[0]].append(e)
    else: tri[e[0][:3]].append(e)
def cand_split(j):
    for e in bi2.get(t[j:j+2],[]):
        if precond(e,j): return True
    for e in tri.get(t[j:j+3],[]):
        pc=e[0]
        if len(pc)>3:
            if t[j+3]==pc[3]: return True
        elif precond(e,j): return True
    return False
def lists_split(j): return bi2.get(t[j:j+2],[])+tri.get(t[j:j+3],[])

def simulate(cand, lists, adv_cost, name):
    total=0; nb=0; stats=collections.Counter()
    for tile in range(NT):
        base=8+tile*65536
        lanes=[]
        for l in range(64):
            cs=base+l*1024
            lanes.append(dict(j=cs,end=cs+1024,active=True,skip=False,cand=False))
        nb+=65536
        while any(L['active'] for L in lanes):
            # skip loop
            it=0
            while any(L['active'] and L['skip'] for L in lanes):
                it+=1
                for L in lanes:
                    if L['active'] and L['skip']:
                        j=L['j']; d=t.find(b'\n',j,j+16)
                        if d>=0 and d<L['end']:
                            L['j']=d+1; L['skip']=False
                            if L['j']>=L['end']: L['active']=False
                        elif d>=0 or j+16>=L['end']: L['active']=False
                        else: L['j']=j+16
            total+=it*25; stats['skip']+=it*25
            # advance
            steps=0
            for s in range(8):
                adv=[L for L in lanes if L['active'] and not L['cand']]
                if not adv: break
                steps+=1
                for L in adv:
                    j=L['j']
                    if t[j]==DEL: L['j']=j+1
                    elif cand(j): L['cand']=True
                    else: L['j']=j+1
                    if L['j']>=L['end']: L['active']=False
            total+=steps*adv_cost+15; stats['adv']+=steps*adv_cost+15
            cl=[L for L in lanes if L['cand']]
            if cl:
                # entry loops
                outer=0; inner=0
                state=[]
                for L in cl:
                    lst=lists(L['j']); state.append([L,lst,0,False])
                while any(s[2]<len(s[1]) for s in state):
                    outer+=1; mx=0
                    for s in state:
                        L,lst,i,m=s; c=0; pend=False
                        while i<len(lst) and not pend:
                            e=lst[i]; i+=1; c+=1
                            if t[L['j']:L['j']+len(e[0])]==e[0] and precond(e,L['j']): pend=True
                        s[2]=i; mx=max(mx,c)
                        if pend and entry_hit(e,L['j']): s[3]=True; s[2]=len(lst)
                    inner+=mx
                ex=60+inner*22+outer*45
                total+=ex; stats['exam']+=ex; stats['exam_rounds']+=1; stats['cands']+=len(cl)
                for s in state:
                    L=s[0]; L['cand']=False
                    if s[3]: L['skip']=True; stats['hits']+=1
                    else:
                        L['j']+=1
                        if L['j']>=L['end']: L['active']=False
    print(name, "wave-instr/B %.3f  lane-slots/B %.1f" % (total/nb, total*64/nb), dict(stats))
simulate(cand_current, lists_current, 18, "current")
simulate(cand_split, lists_split, 28, "split  ")
