"""Block phasing (phaser/phaser.py:2107-2324 phase_v3 and helpers) on integer-indexed blocks.

A block is n position-sorted variants; the allele graph is held as Python-int bitmasks
(node 2*i + a = allele a of variant i), which turns the reference's set unions into word operations.
Same decisions as the reference, including its quirks:
  * resolve: flood fill from node (first variant, allele 0); accepted iff the component has exactly n
    nodes (:2198) -- not necessarily one per variant, and the emitted string skips variants without a node
  * split at weak points (:2271-2324), brute-force each fragment over configurations whose first allele is 0
    (the complement is skipped, :2234), ties -> all '-'
  * left-to-right stitching tests the 4 joint configurations over variants[start:start+used] where start is
    ASSIGNED used (:2152), not advanced by it
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple


def _flip(cfg: str) -> str:
    return "".join("-" if a == "-" else ("1" if a == "0" else "0") for a in cfg)


def _popcount(x: int) -> int:
    return bin(x).count("1")


class Block:
    """variants: global variant ids, position-sorted.  edges: (i, j, cfg) local indices, cfg 0 cis / 1 trans / -1 tie."""

    def __init__(self, n: int, edges: Sequence[Tuple[int, int, int]]):
        self.n = n
        self.adj = [0] * (2 * n)            # allele graph
        self.vadj: List[List[int]] = [[] for _ in range(n)]
        for i, j, cfg in edges:
            self.vadj[i].append(j); self.vadj[j].append(i)
            if cfg == 0:
                pairs = ((2 * i, 2 * j), (2 * i + 1, 2 * j + 1))
            elif cfg == 1:
                pairs = ((2 * i, 2 * j + 1), (2 * i + 1, 2 * j))
            else:
                pairs = ()
            for a, b in pairs:
                self.adj[a] |= 1 << b
                self.adj[b] |= 1 << a

    # :2172-2207
    def resolve(self, lo: int, hi: int, clean: bool):
        n = hi - lo
        if clean:
            inside = ((1 << (2 * hi)) - 1) ^ ((1 << (2 * lo)) - 1)
        else:
            inside = -1
        seed = 2 * lo
        comp = (1 << seed) | (self.adj[seed] & inside)
        todo = comp & ~(1 << seed)
        done = 1 << seed
        while todo:
            low = todo & -todo
            x = low.bit_length() - 1
            done |= low
            comp |= self.adj[x] & inside
            todo = comp & ~done
        if _popcount(comp) == n:
            s = ""
            for i in range(lo, hi):
                if (comp >> (2 * i)) & 1:
                    s += "0"
                elif (comp >> (2 * i + 1)) & 1:
                    s += "1"
            return [s, _flip(s)]
        return None

    def _score(self, idx: Sequence[int], cfg: str) -> int:
        chosen = 0
        for i, a in zip(idx, cfg):
            if a != "-":
                chosen |= 1 << (2 * i + (a == "1"))
        s = 0
        for i, a in zip(idx, cfg):
            if a != "-":
                node = 2 * i + (a == "1")
                # supporting edges to OTHER variants of the slice carrying their configured allele
                s += _popcount(self.adj[node] & chosen & ~(3 << (2 * i)))
        return s

    # :2209-2258
    def best(self, idx: Sequence[int], given=None, attempt: bool = False):
        n = len(idx)
        if given is None:
            if attempt:
                got = self.resolve(idx[0], idx[-1] + 1, True)
                if got is not None:
                    return got
            best_s = -1; best_c = None; ties = 0
            adj = self.adj
            base = idx[0]
            # configurations in itertools.product("01") order restricted to first allele 0 (complements skipped)
            for code in range(1 << (n - 1)):
                chosen = 0
                for k in range(n):
                    bit = (code >> (n - 1 - k)) & 1 if k > 0 else 0
                    chosen |= 1 << (2 * (base + k) + bit)
                s = 0
                c = chosen
                while c:
                    low = c & -c
                    node = low.bit_length() - 1
                    s += _popcount(adj[node] & chosen)
                    c ^= low
                if s > best_s:
                    best_s = s; best_c = code; ties = 1
                elif s == best_s:
                    ties += 1
            if ties == 1:
                cfg = "0" + (format(best_c, "0%db" % (n - 1)) if n > 1 else "")
                return [cfg, _flip(cfg)]
            return ["-" * n, "-" * n]
        cfgs = [given[0][0] + given[1][0], given[0][0] + given[1][1], given[0][1] + given[1][0], given[0][1] + given[1][1]]
        score: Dict[str, int] = {}
        for c in cfgs:
            inv = _flip(c)
            if c + "|" + inv in score or inv + "|" + c in score:
                continue
            score[c + "|" + inv] = self._score(idx, c)
        top = max(score.values())
        winners = [k for k, s in score.items() if s == top]
        if len(winners) == 1:
            return winners[0].split("|")
        return ["-" * n, "-" * n]

    # :2271-2324
    def weak_split(self, max_size: int) -> List[List[int]]:
        n = self.n
        weak = {}
        for p in range(2, n - 1):
            c = 0
            for u in range(n):
                if u < p:
                    for w in self.vadj[u]:
                        if w >= p:
                            c += 1
            weak[p] = c
        pts: List[int] = []
        level = 1
        biggest = n
        frags = [list(range(n))]
        while biggest > max_size or level == 1:
            for p in sorted(weak):
                if weak[p] == level and p + 1 not in pts and p - 1 not in pts:
                    pts.append(p)
            if pts:
                sp = sorted(pts)
                frags = [list(range(0, sp[0]))] + [list(range(sp[i - 1], sp[i])) for i in range(1, len(sp))] + [list(range(sp[-1], n))]
            else:
                frags = [list(range(n))]
            biggest = max(len(x) for x in frags)
            level += 1
        return frags

    # :2107-2170  -> list of sub-blocks, each a list of (local index, allele char)
    def phase(self, max_block_size: int):
        n = self.n
        got = self.resolve(0, n, False)
        if got is not None:
            final = [got]
        else:
            xmax = n if max_block_size == 0 else max_block_size
            subs = self.weak_split(xmax)
            if len(subs) == 1:
                ph = [self.best(x) for x in subs]
            else:
                ph = [self.best(x, attempt=True) for x in subs]
            done = []
            cur = ph[0]
            start = 0
            for i in range(1, len(ph)):
                step = [cur, ph[i]]
                used = math.ceil(sum(sum(len(y) for y in x) for x in step) / 2)
                new = self.best(list(range(start, min(n, start + used))), given=step)
                if "-" in new[0]:
                    done.append(cur); start = used; cur = ph[i]
                else:
                    cur = new
            final = done + [cur]
        res = []
        vi = 0
        for blk in final:
            ob = []
            for a in blk[0]:
                ob.append((vi, a))
                vi += 1
            if ob and ob[0][1] != "-":
                res.append(ob)
        return res
