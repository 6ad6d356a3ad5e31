#!/usr/bin/env python3
"""Shape of the LZ4 streams the GPU decoder sees (design evidence, CPU only): sequence mix, offsets, and -- for a decoder
that resolves the output in CHUNKS of C bytes, every byte fetching out[o - off] -- how many pointer-doubling rounds the
in-chunk chains need.   python3 tests/perf/lz4_stream_stats.py [fast:2 hc:9] [--blocks 2]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

import blockfile_tool as bt  # noqa: E402


def parse(comp, usize):
    """-> arrays per sequence: literal length, match length (0 for the last), offset, output position of the literals"""
    ll_a, ml_a, off_a, op_a = [], [], [], []
    ip = op = 0
    n = len(comp)
    while ip < n:
        tok = comp[ip]; ip += 1
        ll = tok >> 4
        if ll == 15:
            while True:
                e = comp[ip]; ip += 1; ll += e
                if e != 255: break
        lit_at = op
        ip += ll; op += ll
        if ip >= n:
            ll_a.append(ll); ml_a.append(0); off_a.append(0); op_a.append(lit_at)
            break
        off = comp[ip] | (comp[ip + 1] << 8); ip += 2
        ml = tok & 15
        if ml == 15:
            while True:
                e = comp[ip]; ip += 1; ml += e
                if e != 255: break
        ml += 4
        ll_a.append(ll); ml_a.append(ml); off_a.append(off); op_a.append(lit_at)
        op += ml
    assert op == usize, (op, usize)
    return np.array(ll_a), np.array(ml_a), np.array(off_a), np.array(op_a)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("modes", nargs="*", default=["fast:2", "hc:9"])
    ap.add_argument("--blocks", type=int, default=2)
    ap.add_argument("--kind", default="na12878")
    args = ap.parse_args()
    import oracle
    per = bt.BLOCK_BYTES // 2
    for mode_level in args.modes:
        mode, level = mode_level.split(":")
        for b in range(args.blocks):
            if args.kind == "na12878":
                f = oracle.generate(oracle.GEN_NA12878, 7, 1, b * per, per)
            else:
                f = oracle.generate(oracle.GEN_UNIFORM, 7, 0x0FFF, b * per, per)
            raw = f.tobytes()
            comp = bt.compress_block(raw, mode, int(level))
            ll, ml, off, op = parse(comp, len(raw))
            ns = len(ll)
            bare = (ll == 0) & (ml <= 18) & (ml > 0)
            lit14 = (ll >= 1) & (ll <= 14) & (ml <= 18) & (ml > 0)
            print("LZ4-%s-%s block %d: %d -> %d bytes (%.2fx), %d sequences (%.2f out bytes, %.2f in bytes each)" %
                  (mode, level, b, len(raw), len(comp), len(raw) / len(comp), ns, len(raw) / ns, len(comp) / ns))
            print("  bare %.1f %%, 1..14 literals + short match %.1f %%, other %.2f %%; literal bytes %.2f %% of output; "
                  "match length: mean %.1f, >18: %.2f %%, max %d" %
                  (100 * bare.mean(), 100 * lit14.mean(), 100 * (1 - bare.mean() - lit14.mean()), 100 * ll.sum() / len(raw),
                   ml[ml > 0].mean(), 100 * (ml > 18).mean(), ml.max()))
            o = off[ml > 0]
            qs = [10, 25, 50, 75, 90, 95, 99]
            print("  offsets: " + ", ".join("p%d %d" % (q, np.percentile(o, q)) for q in qs) +
                  "; <64: %.1f %%, <256: %.1f %%, <1024: %.1f %%, >8128: %.1f %%, off < ml: %.2f %%" %
                  (100 * (o < 64).mean(), 100 * (o < 256).mean(), 100 * (o < 1024).mean(), 100 * (o > 8128).mean(),
                   100 * (off[ml > 0] < ml[ml > 0]).mean()))
            # bare-run lengths (what a 16- or 21-lane batch parse of 3-byte sequences sees)
            runs = np.diff(np.flatnonzero(np.concatenate(([True], ~bare, [True])))) - 1
            runs = runs[runs > 0]
            print("  runs of bare sequences: mean %.1f, median %d; sequences in runs >= 16: %.1f %%" %
                  (runs.mean(), np.median(runs), 100 * runs[runs >= 16].sum() / max(1, bare.sum())))
            # per-byte source pointer
            n = len(raw)
            src = np.arange(n, dtype=np.int64)           # literals point at themselves (resolved)
            m_start = op + ll
            for s, l, d in zip(m_start[ml > 0], ml[ml > 0], off[ml > 0]):
                src[s:s + l] = np.arange(s - d, s - d + l)
            for C in (64, 128, 256, 512, 1024, 4096):
                base = (np.arange(n) // C) * C
                ptr = np.where(src >= base, src, np.arange(n))   # in-chunk pointers only; others are roots
                hops = np.zeros(n, dtype=np.int32)
                # depth by iterating single hops (exact chain length), bounded
                cur = ptr.copy()
                depth = np.zeros(n, dtype=np.int32)
                live = cur != np.arange(n)
                rounds_single = np.zeros(n // C + 1, dtype=np.int32)
                it = 0
                p = ptr.copy()
                # pointer doubling rounds per chunk
                rounds = np.zeros((n + C - 1) // C, dtype=np.int32)
                r = 0
                while True:
                    nxt = p[p]
                    ch = nxt != p
                    if not ch.any():
                        break
                    r += 1
                    idx = np.unique(np.flatnonzero(ch) // C)
                    rounds[idx] = r
                    p = nxt
                # the first hop itself is one gather; rounds = extra doubling steps after it
                inchunk = (ptr != np.arange(n))
                frac_in = inchunk.mean()
                print("  chunk %5d B: %4.1f %% of bytes point inside their chunk; doubling rounds after the first gather: mean %.2f, "
                      "p50 %d, p90 %d, max %d; chunks with 0 rounds %.1f %%" %
                      (C, 100 * frac_in, rounds.mean(), np.percentile(rounds, 50), np.percentile(rounds, 90), rounds.max(),
                       100 * (rounds == 0).mean()))


if __name__ == "__main__":
    main()
