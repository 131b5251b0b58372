"""Test-only: a numpy walk over the flattened HBM image exported by the product's host tree
(aic_st_export), following exactly what the HIP matcher does (hash probe -> node record -> edge
tokens; NodeRec.best instead of iterating children).  Lets the CPU suite validate the image the
device will read — layout, hash table, incrementally maintained `best` — against the golden
vectors without a GPU.  It is not part of the product and not an oracle."""
import numpy as np

F = np.float32


def edge_hash(parent: int, token: int) -> int:
    M = 0xFFFFFFFF
    h = ((parent & M) * 0x9E3779B1 & M) ^ ((token & M) * 0x85EBCA77 & M)
    h ^= h >> 15
    h = h * 0xC2B2AE3D & M
    h ^= h >> 13
    return h


class FlatCand:
    def __init__(self):
        self.token_ids, self.parents, self.probs, self.score, self.match_len = [], [], [], 0.0, 0


def lookup(img, parent, token):
    H = img["hash"]
    mask = len(H) - 1
    h = edge_hash(parent, token) & mask
    for _ in range(len(H)):
        p, t, c, st = (int(x) for x in H[h])
        if st == 0:
            return -1
        if st == 1 and p == parent and t == token:
            return c
        h = (h + 1) & mask
    return -1


def speculate(img, pattern, max_spec, factor, offset, min_prob, max_depth):
    nodes, toks, base = img["nodes"], img["tokens"], img["seq_base"]
    n = len(pattern)
    best = FlatCand()
    for s in range(max(n - max_depth, 0), n):
        node, idx, ok = 0, 0, True
        for i in range(s, n):
            if idx >= nodes[node][4]:
                node = lookup(img, node, pattern[i])
                if node < 0:
                    ok = False
                    break
                idx = 0
            r = nodes[node]
            if toks[base[r[2]] + r[3] + idx] != pattern[i]:
                ok = False
                break
            idx += 1
        if not ok:
            continue
        match_len = n - s
        scaled = F(F(match_len) * F(factor)) + F(offset)
        budget = max(min(int(np.float64(F(scaled)) + 1e-6), max_spec), 0)
        c = FlatCand()
        prob, score = F(1.0), F(0.0)
        while len(c.token_ids) < budget and prob >= F(min_prob):
            r = nodes[node]
            if idx < r[4]:
                c.parents.append(len(c.token_ids) - 1)
                c.token_ids.append(int(toks[base[r[2]] + r[3] + idx]))
                c.probs.append(float(prob))
                score = F(score + prob)
                idx += 1
            else:
                b = int(r[5])
                if b < 0:
                    break
                prob = F(prob * F(F(nodes[b][0]) / F(r[0])))
                node, idx = b, 0
        c.score = float(score)
        if c.score > best.score:
            best = c
            best.match_len = match_len
    return best
