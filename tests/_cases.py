"""Shared random-case generator for the parity / fuzz tests."""
import random  # noqa: F401


def _rand_case(rng, sigma, max_m=29):
    alpha = bytes(rng.sample(range(97, 123), sigma - 1)) + b" "
    m = rng.randint(3, max_m)
    k = rng.randint(0, min(8, m - 1))
    pat = bytes(rng.choice(alpha[:-1]) for _ in range(m))
    recs = []
    for _ in range(rng.randint(1, 60)):
        L = rng.randint(0, 120)
        r = bytearray(rng.choice(alpha) for _ in range(L))
        if rng.random() < 0.4 and L > m + 4:
            # plant a mutated copy
            v = bytearray(pat)
            for _ in range(rng.randint(0, k + 1)):
                op = rng.randint(0, 2)
                pos = rng.randrange(len(v)) if v else 0
                if op == 0 and v:
                    v[pos] = rng.choice(alpha)
                elif op == 1 and len(v) > 1:
                    del v[pos]
                else:
                    v.insert(pos, rng.choice(alpha))
            at = rng.randint(0, L - len(v)) if L > len(v) else 0
            r[at:at + len(v)] = v
        recs.append(bytes(r))
    text = b"\n".join(recs) + (b"\n" if rng.random() < 0.8 else b"")
    return pat, k, text
