#!/usr/bin/env python3
"""Build-container only: pin row f1's file format to the REFERENCE'S OWN WRITER.

Runs the reference's unmodified `bench compress` (benchmark/flagstats.cpp:110-190, built by
`make -C oracle refbench` into oracle/_ref/bench_ref from the sources where they lie) on small,
seeded, highly compressible FLAG streams and commits what it wrote under tests/golden/blockfiles/
as DATA (block files + a manifest with the input recipe, the expected counters and the reference's
own `decompress -d` / `-s` output on each file).  No reference source is stored.

    python tests/golden/make_blockfiles.py
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "blockfiles")
BENCH = os.path.join(ROOT, "oracle", "_ref", "bench_ref")

# name, flags, seed, writer arguments
CASES = [
    ("tiny", 1000, 11, ["-l", "-f", "-c", "2"]),
    ("ragged", 512000 + 300001, 12, ["-l", "-f", "-c", "2"]),          # 1.59 blocks
    ("exact2", 2 * 512000, 13, ["-l", "-f", "-c", "1"]),               # exact multiple: the writer appends an empty block
    ("hc", 700001, 14, ["-l", "-c", "9"]),                             # LZ4-HC
    # Zstandard frames behind the same block header (benchmark/flagstats.cpp:192-226)
    ("zragged", 512000 + 123457, 15, ["-z", "-c", "3"]),
    ("zexact1", 512000, 16, ["-z", "-c", "1"]),                        # exact multiple: trailing block of 0 flags
    ("ztiny", 777, 17, ["-z", "-c", "19"]),
]


def recipe_input(n, seed):
    """NA12878-like (+eps: some dup / QC-fail) flags in runs of 97 -- compresses ~50x -- with a 4096-flag stretch
    of uniform 12-bit values in the middle (literal runs; bits 12-15 stay clear so the reference's SIMD
    kernels agree with FLAGSTAT_scalar on the live slots)."""
    import oracle
    base = oracle.generate(oracle.GEN_NA12878, seed, 1, 0, (n + 96) // 97)
    a = np.repeat(base, 97)[:n].copy()
    k = min(4096, n // 4)
    a[n // 2: n // 2 + k] = oracle.generate(oracle.GEN_UNIFORM, seed + 1000, 0x0FFF, 0, k)
    return a


def main():
    import oracle
    assert os.path.exists(BENCH), "run `make -C oracle refbench` first"
    os.makedirs(OUT, exist_ok=True)
    manifest = {"writer": "reference bench compress (benchmark/flagstats.cpp:110-226), liblz4 1.9.3 / libzstd 1.4.9", "files": {}}
    with tempfile.TemporaryDirectory() as tmp:
        for name, n, seed, wargs in CASES:
            a = recipe_input(n, seed)
            raw = os.path.join(tmp, name + ".bin")
            a.tofile(raw)
            before = set(os.listdir(tmp))
            subprocess.run([BENCH, "compress", "-i", raw, "-o", os.path.join(tmp, name)] + wargs, check=True,
                           capture_output=True)
            made = sorted(set(os.listdir(tmp)) - before)
            assert len(made) == 1 and made[0].endswith((".lz4", ".zst")), made
            is_zst = made[0].endswith(".zst")
            blob = open(os.path.join(tmp, made[0]), "rb").read()
            with open(os.path.join(OUT, made[0]), "wb") as f:
                f.write(blob)
            # the reference reading its own file: counter table (-d) and samtools text (-s)
            rd = subprocess.run([BENCH, "decompress", "-i", os.path.join(tmp, made[0]), "-d"], capture_output=True, text=True)
            rows = re.findall(r"^(\w+)\t(\d+)\t(\d+)$", rd.stderr, flags=re.M)
            rs = subprocess.run([BENCH, "decompress", "-i", os.path.join(tmp, made[0]), "-s"], capture_output=True, text=True)
            want = oracle.flagstat_hist(a)
            if is_zst:
                # the reference's .zst reader prints no counters (the table is commented out,
                # benchmark/flagstats.cpp:676-679), only "[ZSTD file] Time elapsed .. ms <tot_flags>"
                m = re.search(r"\[ZSTD [^\]]*\] Time elapsed \d+ ms (\d+)", rd.stderr)
                assert m and int(m.group(1)) == n, rd.stderr
                rows, rs_text = None, None
            elif n % 512000 == 0:
                # The reference's reader cannot read its own writer's exact-multiple files: the writer appends a
                # block of 0 flags (benchmark/flagstats.cpp:122-138 loops once more after the last full read),
                # LZ4_decompress_safe returns 0 for it and the reader bails out (:320-321) before printing.
                assert len(rows) == 0 and rd.returncode != 0 and "return 0" in rd.stdout, (rd.stdout, rd.stderr)
                rows, rs_text = None, None
            else:
                assert len(rows) == 15, rd.stderr
                assert "in total" in rs.stdout, rs.stdout + rs.stderr
                rs_text = rs.stdout
            ref_scalar = oracle.ref_call("FLAGSTAT_scalar", a) if n < 2 ** 32 else None
            assert ref_scalar is None or np.array_equal(ref_scalar.astype(np.uint64), want)
            # the reference's own read-back agrees with FLAGSTAT_scalar on every slot the scalar rule writes,
            # except pass-QC slot 9, which only the SIMD kernels fill (libflagstats.h:1843)
            if rows is not None:
                table = {k: (int(p), int(f)) for k, p, f in rows}
                names = oracle.SAM_FLAG_NAMES
                for slot in oracle.LIVE_SLOTS:
                    nm = names[slot % 16]
                    assert table[nm][slot // 16] == int(want[slot]), (name, slot, table[nm], int(want[slot]))
            manifest["files"][made[0]] = {
                "case": name, "n_flags": n, "seed": seed, "writer_args": wargs, "bytes": len(blob),
                "input_sha256": hashlib.sha256(a.tobytes()).hexdigest(),
                "recipe": "recipe_input(n, seed) of tests/golden/make_blockfiles.py",
                "scalar_counters": [int(v) for v in want],
                "reference_decompress_d": rows,
                "reference_decompress_s_stdout": rs_text,
                "reference_reader_exits_on_trailing_empty_block": rows is None and not is_zst,
                "codec": "zstd" if is_zst else "lz4",
                "reference_reader_total_flags": n if is_zst else None,
            }
            print(made[0], len(blob), "bytes for", n, "flags")
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
