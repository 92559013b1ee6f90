"""The oracle's chain levels (>= 5), restated a second time as an executable specification in plain Python and compared
sequence for sequence on small inputs: exact hash chains (4-byte hash, newest first, chainDepth links, best gain),
gain-based lazy rules, greedy parse with bounded extension and 4-byte backward extension, and the window-by-window
repeat-offset aware parse.  Independent of the C code (nothing shared but the profile numbers), so a slip in
oracle/qzstd_oracle.c that still round-trips — a wrong tie rule, an off-by-one in a window edge — shows up here, on CPU.
Pure-Python loops: small cases only."""
import pytest

import qz_bind as B
import qz_corpus as K

P1 = 2654435761


def bitlen(x):  # 31 - clz(x) for x >= 1
    return x.bit_length() - 1


def seg_end(pf, p, n):
    return min(n, ((p >> pf.segLog) + 1) << pf.segLog) if pf.segLog else n


def candidates(pf, src):
    n = len(src)
    nh = n - 3 if n >= 4 else 0
    tbl = {}
    chain = [0] * (n + 1)
    cand = [(0, 0)] * (n + 1)
    for p in range(nh):
        if p + 4 > seg_end(pf, p, n):  # would hash bytes of the next segment: takes no part
            continue
        v = src[p:p + 4]
        slot = ((int.from_bytes(v, "little") * P1 & 0xFFFFFFFF) * pf.tableSize) >> 32
        link = tbl.get(slot, 0)
        chain[p] = link
        tbl[slot] = p + 1
        cap = min(pf.capLen, seg_end(pf, p, n) - p)
        best, bg = (0, 0), 0
        for _ in range(pf.chainDepth):
            if link == 0:
                break
            q = link - 1
            if src[q:q + 4] == v:
                l = 0
                while l < cap and src[q + l] == src[p + l]:
                    l += 1
                g = 4 * l - bitlen(p - q + 1)
                if l >= 4 and (best[0] == 0 or g > bg):
                    best, bg = (l, p - q), g
            link = chain[q]
        cand[p] = best
    return cand, nh


def min_len(pf, off):
    return pf.minMatch + (1 if off >> pf.farLog1 else 0) + (1 if off >> pf.farLog2 else 0)


def take(pf, c):
    return c[0] != 0 and c[0] >= min_len(pf, c[1])


def extend(pf, src, p, off, L):
    lim = min(seg_end(pf, p, len(src)), ((p >> pf.extLog) + 2) << pf.extLog)
    while p + L < lim and src[p + L - off] == src[p + L]:
        L += 1
    return L


def back(pf, src, q, off, anchor):
    b = 0
    if pf.segLog:
        anchor = max(anchor, (q >> pf.segLog) << pf.segLog)  # never backwards across the start of the segment
    while b < pf.backExt and q - b > anchor and q - off - b > 0 and src[q - b - 1] == src[q - off - b - 1]:
        b += 1
    return b


def parse_plain(pf, src, cand, nh):
    def gain(c):
        return 4 * c[0] - bitlen(c[1] + 1)

    def is_start(p):
        if not take(pf, cand[p]):
            return False
        g = gain(cand[p])
        if p + 1 < nh and (p & 63) < 63 and take(pf, cand[p + 1]) and gain(cand[p + 1]) > g + 4:
            return False
        if p + 2 < nh and (p & 63) < 62 and take(pf, cand[p + 2]) and gain(cand[p + 2]) > g + 7:
            return False
        return True

    out, p, anchor = [], 0, 0
    while p < nh:
        if not is_start(p):
            p += 1
            continue
        L, off = cand[p]
        if L == pf.capLen:
            L = extend(pf, src, p, off, L)
        b = back(pf, src, p, off, anchor)
        out.append((off, p - b - anchor, L + b))
        p += L
        anchor = p
    out.append((0, len(src) - anchor, 0))
    return out


def parse_rep(pf, src, cand, nh):
    n = len(src)
    CAP, MIN = 32, 3
    out, cur, anchor, rep, rep_seg = [], 0, 0, [0, 0], 0
    while cur < nh:
        if pf.segLog and (cur >> pf.segLog) != rep_seg:  # a new segment starts without repeat offsets
            rep, rep_seg = [0, 0], cur >> pf.segLog
        start_end = seg_end(pf, cur, n) - 4 + 1  # nothing starts in a segment's last positions that cannot be hashed (4-byte hash here)
        if cur >= start_end:
            cur = seg_end(pf, cur, n)
            continue
        lim = min(((cur >> pf.tileLog) + 1) << pf.tileLog, start_end)
        W = min(pf.repWin, lim - cur)
        V = min(W + 2, lim - cur)
        G, opt = [], []
        for k in range(V):
            p = cur + k
            c = cand[p]
            g, o = (4 * c[0] + 32 - bitlen(c[1] + 1), 0) if take(pf, c) else (0, 0)
            for r in range(2):
                if rep[r]:
                    mx, l = min(seg_end(pf, p, n) - p, CAP), 0
                    while l < mx and src[p - rep[r] + l] == src[p + l]:
                        l += 1
                    rg = 0 if l < MIN else (1000 - r if l >= CAP else 4 * l + 36 - r)
                    if rg > g:
                        g, o = rg, 1 + r
            G.append(g)
            opt.append(o)
        pick = None
        for k in range(W):
            if G[k] == 0 or (k + 1 < V and G[k + 1] > G[k] + 4) or (k + 2 < V and G[k + 2] > G[k] + 11):
                continue
            pick = k
            break
        if pick is None:
            cur += W
            continue
        q = cur + pick
        if opt[pick]:
            off = rep[opt[pick] - 1]
            mx, L = min(seg_end(pf, q, n) - q, CAP), 0
            while L < mx and src[q - off + L] == src[q + L]:
                L += 1
            if L == CAP:
                L = extend(pf, src, q, off, L)
        else:
            L, off = cand[q]
            if L == pf.capLen:
                L = extend(pf, src, q, off, L)
        b = back(pf, src, q, off, anchor)
        out.append((off, q - b - anchor, L + b))
        if off != rep[0]:
            rep = [off, rep[0]]
        cur = anchor = q + L
    out.append((0, n - anchor, 0))
    return out


def blocks():
    yield K.text(21, 2500)
    yield K.weblog(22, 3000)
    yield K.binary_struct(23, 2000)
    yield b"".join(b"record%05d;" % (i % 7) + bytes(53) for i in range(40))  # runs + a 65-byte period: long repeats, capped matches
    yield (b"abcdefgh" * 5 + b"X") * 50
    yield K.text(24, 700) + K.text(24, 700) + K.text(25, 300) + K.text(24, 700)  # far-ish repeats across a tile edge
    yield K.weblog(26, 31000) + K.weblog(26, 4000)  # crosses the 32 KiB segment boundary with repeats on both sides


@pytest.mark.parametrize("level", [5, 6, 9, 10, 12, 0x106])
def test_chain_levels_equal_the_python_specification(oracle, level):
    for blk in blocks():
        pf = oracle.profile(level, len(blk))
        assert pf.chainDepth and pf.hashBytes == 4 and pf.lazy == 4 and not pf.nearTab and not pf.longSize
        cand, nh = candidates(pf, blk)
        want = parse_rep(pf, blk, cand, nh) if pf.repWin else parse_plain(pf, blk, cand, nh)
        n, seqs = oracle.find(pf, blk)
        got = [(seqs[i].offset, seqs[i].litLength, seqs[i].matchLength) for i in range(n)]
        assert got == want, "level %#x, block of %d: first difference at sequence %d" % (
            level, len(blk), next(i for i, (a, b) in enumerate(zip(got + [None], want + [None])) if a != b))
