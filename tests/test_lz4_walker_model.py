"""CPU: the ALGORITHM of the GPU LZ4 decoder's walker wave (flagstat_lz4_kernels.hip, lz4wg_walk), restated in Python and
checked against the true token chain of liblz4-written blocks.  The kernel finds which input bytes are tokens with one
lane per 32-byte segment of a 2 KiB tile: (1) backward over a segment's positions, the "exit" of every position -- where
a chain through it lands in the next segment, from the token alone (a sequence whose lengths sit in its token is
3 + literals (+ 1 match-length byte) long); (2) the chain through the tile, segment by segment, from a 12-entry exit table
per segment; (3) forward again, the positions reachable from each segment's entry.  Tokens whose literal length continues
in further bytes (or whose match-length byte is 255) stop the tile.  This file pins that restatement -- the tables' sizes,
the stop rules, the tile-end arithmetic, and (r05) that the prefix SCAN over the exit tables gives what the serial walk gives -- on LZ4-fast and LZ4-HC streams of flag data and on streams full of long matches
and literal runs; the HIP code itself is checked on the GPU (tests/test_gpu_blockfile.py)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import blockfile_tool as bt  # noqa: E402

SEG, SEGS, TAB = 32, 64, 12


def true_chain(comp):
    """token positions and, per token, whether the window form covers it (all lengths in the token, one length byte < 255)"""
    pos, simple, ip, n = [], {}, 0, len(comp)
    while ip < n:
        tok = comp[ip]
        ll, p, ok = tok >> 4, ip + 1, (tok >> 4) < 15
        if ll == 15:
            while True:
                e = comp[p]
                p += 1
                ll += e
                if e != 255:
                    break
        p += ll
        pos.append(ip)
        if p >= n:
            simple[ip] = False          # the last sequence: literals only
            break
        p += 2
        if (tok & 15) == 15:
            first = True
            while True:
                e = comp[p]
                p += 1
                if first and e == 255:
                    ok = False
                first = False
                if e != 255:
                    break
        simple[ip] = ok
        ip = p
    return pos, simple


def walk_tile(comp, ip):
    """one tile at input position ip, as the kernel does it: (members found, positions advanced, stopped at a token for the scalar code)"""
    iend = len(comp)
    nseg = min(SEGS, (iend - ip - 50) // SEG + 1) if iend >= ip + 50 else 0
    if nseg == 0:
        return None
    byte = lambda x: comp[x] if x < iend else 0  # noqa: E731
    tabs = []
    for s in range(SEGS):
        win, tab = [0] * 18, 0
        for i in range(SEG - 1, -1, -1):
            tok = byte(ip + SEG * s + i)
            ll = tok >> 4
            d = 3 + ll + (1 if (tok & 15) == 15 else 0)
            ex = win[d - 1] if d - 1 < 18 else 0
            if d >= SEG - i:
                ex = i + d - SEG
            if ll == 15:
                ex = 31
            win = [ex] + win[:17]
            if i < TAB:
                tab |= ex << (5 * i)
        tabs.append(tab)
    ent, e, nvalid, e_next = [0] * SEGS, 0, nseg, 0
    for sg in range(SEGS):
        if sg % 8 == 0 and sg >= nvalid:
            break
        ent[sg] = e
        e2 = (tabs[sg] >> (5 * e)) & 31
        last = (sg + 1 == nvalid) or e2 >= TAB
        if last and sg < nvalid:
            e_next = e2
        if last:
            nvalid = min(nvalid, sg + 1)
        e = 0 if e2 >= TAB else e2
    # r05: the kernel does not walk that chain any more, it SCANS it -- lane s ends up with F(s) = T(s) o ... o T(0) after six
    # doubling rounds (a lane composes its table with the one 1, 2, 4 ... lanes to its left; an exit of 12 or more stays what it
    # is), F(s)(0) is segment s's exit, the first exit of 12 or more (or the last segment) ends the tile.  Same entries, same end.
    ident = sum(e << (5 * e) for e in range(TAB))
    F = list(tabs)
    d = 1
    while d < SEGS:
        G = [F[s - d] if s >= d else ident for s in range(SEGS)]
        nF = []
        for s in range(SEGS):
            n = 0
            for en in range(TAB):
                v = (G[s] >> (5 * en)) & 31
                t = (F[s] >> ((5 * v) & 63)) & 31
                n |= (v if v >= TAB else t) << (5 * en)
            nF.append(n)
        F, d = nF, d * 2
    exits = [f & 31 for f in F]
    last_seg = min(s for s in range(SEGS) if exits[s] >= TAB or s + 1 >= nseg)
    assert last_seg + 1 == nvalid and exits[last_seg] == e_next, ("scan", ip, last_seg, nvalid, exits[last_seg], e_next)
    assert [0] + exits[:last_seg] == ent[:nvalid], ("scan entries", ip)
    members = []
    for s in range(nvalid):
        reach, stop = 1 << ent[s], SEG
        for i in range(SEG):
            tok = byte(ip + SEG * s + i)
            ll, mlc = tok >> 4, tok & 15
            if not (reach >> i) & 1:
                continue
            if ll == 15 or (mlc == 15 and byte(ip + SEG * s + i + 3 + ll) == 255):
                stop = min(stop, i)
                continue
            reach |= 1 << (i + 3 + ll + (1 if mlc == 15 else 0))
        got = [ip + SEG * s + i for i in range(min(stop, SEG)) if (reach >> i) & 1]
        members += got
        if stop < SEG:
            return members, SEG * s + stop, True
    assert e_next < 18, "the chain saw a token for the scalar code that the members pass did not"
    return members, SEG * nvalid + e_next, False


def check_block(raw, mode, level):
    comp = bt.compress_block(raw, mode, level)
    assert bt.decompress_block_ref(comp, len(raw)) == raw
    pos, simple = true_chain(comp)
    nxt = {p: (pos[k + 1] if k + 1 < len(pos) else len(comp)) for k, p in enumerate(pos)}
    chain = set(pos)
    ip, tiles = 0, 0
    while True:
        t = walk_tile(comp, ip)
        if t is None:
            break
        members, adv, stopped = t
        end = ip + adv
        assert end in chain, (mode, level, "tile ends off the chain", ip, end)
        assert members == [p for p in pos if ip <= p < end], (mode, level, "members", ip)
        assert all(simple[p] for p in members), (mode, level, "a member the window form does not cover", ip)
        if stopped:
            assert not simple[end], (mode, level, "stopped at a token the window form covers", end)
            end = nxt[end]                 # the scalar code takes that sequence
        else:
            assert adv > 0
        ip, tiles = end, tiles + 1
        if ip >= len(comp):
            break
    return tiles


def flags_like(kind, n, seed):
    import oracle
    rs = np.random.RandomState(seed)
    if kind == "na12878":
        return oracle.generate(oracle.GEN_NA12878, seed, 1, 0, n)
    if kind == "uniform":
        return oracle.generate(oracle.GEN_UNIFORM, seed, 0x0FFF, 0, n)
    if kind == "zeros":
        return np.zeros(n, dtype=np.uint16)
    if kind == "runs":
        parts, left = [], n
        while left > 0:
            m = int(min(left, rs.choice([1, 3, 17, 40, 300, 5000])))
            parts.append(np.full(m, rs.randint(0, 4096), dtype=np.uint16) if rs.rand() < 0.7 else rs.randint(0, 65536, size=m).astype(np.uint16))
            left -= m
        return np.concatenate(parts)
    base = rs.randint(0, 4096, size=3000).astype(np.uint16)
    parts, left = [], n
    while left > 0:
        o, m = int(rs.randint(0, 2900)), int(min(left, rs.randint(2, 100)))
        parts.append(base[o:o + m])
        left -= len(parts[-1])
    return np.concatenate(parts)[:n]


@pytest.mark.parametrize("kind", ["na12878", "uniform", "zeros", "runs", "repeats"])
@pytest.mark.parametrize("mode,level", [("fast", 2), ("hc", 9)])
def test_walker_tiles_follow_the_token_chain(kind, mode, level):
    for seed, n in ((1, 60_000), (2, 9_000), (3, 150_000)):
        raw = np.ascontiguousarray(flags_like(kind, n, seed)).tobytes()
        check_block(raw, mode, level)


def test_kernel_constants_match_the_model():
    src = open(os.path.join(ROOT, "libflagstats_amd", "csrc", "flagstat_lz4_kernels.hip")).read()
    assert "kWgSeg = 32, kWgTileSegs = 64" in src and "if (i < 12) {" in src and "e2 >= 12u" in src and "(iend - ip - 50u) / kWgSeg + 1u" in src
