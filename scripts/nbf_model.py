"""Model check of the neighbour-byte filter (AGH_MS_NBF): whenever an entry matches at j by ms_match_k1's rule
(piece verbatim + the other side within one edit), the filter built from the masks lets j through -- at every
alignment of j inside its chunk."""
import random
def side_within_one_edit(S, B, L, delim):
    # S: text bytes nearest first (>= L+1 available), B: pattern bytes nearest first
    x = [S[i] != B[i] for i in range(L)]
    if not any(x): return True
    i = x.index(True)
    if all(S[i + t] == B[i + 1 + t] for t in range(L - i - 1)): return True          # missing
    if S[i] == delim: return False
    if all(S[t] == B[t] for t in range(i + 1, L)): return True                        # replaced
    return all(S[t + 1] == B[t] for t in range(i, L))                                 # extra
rng = random.Random(5)
alpha = b"abcd"
bad = checked = matches = 0
for trial in range(300):
    npat = rng.randint(1, 6)
    pats = [bytes(rng.choice(alpha) for _ in range(rng.randint(8, 14))) for _ in range(npat)]
    ents = []   # (gram, piece, before?, Bside)
    for p in pats:
        h = len(p) // 2
        ents.append((p[:h], False, p[h:]))               # head piece, other side behind (nearest first = as is)
        ents.append((p[h:], True, p[:h][::-1]))          # tail piece, other side in front, nearest first
    masks = {}
    for piece, before, Bs in ents:
        g = piece[:4]
        mx, ml, mb = masks.get(g, (0, 0, 0))
        L = min(len(Bs), 7)
        if len(piece) >= 5: mx |= 1 << (piece[4] & 31)
        else:
            near2 = (1 << (Bs[0] & 31)) | (1 << (Bs[1] & 31)) if L >= 2 else 0xffffffff
            if before: mb |= near2
            else: ml |= near2
        masks[g] = (mx, ml, mb)
    parts = []
    for _ in range(40):
        v = bytearray(rng.choice(pats))
        r = rng.random()
        if r < 0.7:
            at = rng.randrange(len(v)); op = rng.randint(0, 2)
            if op == 0: v[at] = rng.choice(alpha + b"\n")
            elif op == 1: del v[at]
            else: v.insert(at, rng.choice(alpha + b"\n"))
        parts.append(bytes(v)); parts.append(bytes(rng.choice(alpha + b"\n") for _ in range(rng.randint(0, 6))))
    text = b"zzzzzzzzzz" + b"".join(parts) + b"z" * 30
    for j in range(10, len(text) - 24):
        g = text[j:j + 4]
        if g not in masks: continue
        mx, ml, mb = masks[g]
        truth = False
        for piece, before, Bs in ents:
            if text[j:j + len(piece)] != piece: continue
            L = len(Bs)
            if L > 7: continue
            S = text[j - 8:j][::-1] if before else text[j + len(piece):j + len(piece) + 8]
            if side_within_one_edit(S, Bs, L, 10): truth = True
        if not truth: continue
        matches += 1
        S0, S1, T1, T2 = text[j + 4] & 31, text[j + 5] & 31, text[j - 1] & 31, text[j - 2] & 31
        for pc in range(16):
            ok = (mb >> T1) & 1
            ok |= (mb >> T2) & 1 if pc >= 1 else int(mb != 0)
            ok |= ((mx | ml) >> S0) & 1 if pc <= 14 else int((mx | ml) != 0)
            ok |= (ml >> S1) & 1 if pc <= 13 else int(ml != 0)
            checked += 1
            if not ok:
                bad += 1
                if bad < 5: print("MISSED", text[j - 8:j + 16], pats, pc)
print("true matches", matches, "checked", checked, "missed", bad)
