"""Report layer: the step AFTER the hot path.

``samtools_counts`` / ``samtools_flagstat_text`` turn the 32 counters into the samtools-flagstat
fields and text the reference's bench prints (``benchmark/flagstats.cpp:577-588``, struct
``bam_flagstat_t`` ``:42-48``, ``percent`` ``:73-78``).  Mapping of counter slots to samtools fields
(w = 0 pass-QC / 1 fail-QC, slot = 16*w + k):

    n_reads[1]  = slot 25;  n_reads[0] = n_values - slot 25
    n_secondary = 8   n_supp = 11   n_dup = 10   n_read1 = 6   n_read2 = 7
    n_pair_good = 12  n_sgltn = 13  n_pair_map = 14
    n_mapped[w] = n_reads[w] - slot 2          (libflagstats counts UNMAP, pyx:34)
    n_pair_all[w] = slot 0 + 16*w              COUNTED by K1 (primary paired reads by QC class) and delivered
                                               by the superset entry points (FLAGSTATS_u16_x64_superset,
                                               FLAGSTATS_hip_device_u16_superset*): exact on any input,
                                               like the reference's samtools loop (benchmark/flagstats.cpp:58).

Counters from the scalar-exact entry points carry no n_pair_all (FLAGSTAT_scalar has no such slot);
``derived_pair_all=True`` then falls back to read1 + read2 -- what the reference's Python module calls
"paired_in_seq" (python/libflagstats.pyx:35), exact only when every primary paired read is exactly
one of read1/read2 -- and the text says so on that line.

The two "mate mapped to a different chr" lines of samtools need RNAME/MAPQ, not FLAG, and are
commented out in the reference as well (``:589-590``).
"""
from __future__ import annotations

import numpy as np


def samtools_counts(counters, n_values: int, derived_pair_all: bool = False) -> dict:
    """samtools fields from 32 SUPERSET counters (or scalar-exact ones with ``derived_pair_all=True``)."""
    c = [int(v) for v in np.asarray(counters).ravel()]
    assert len(c) == 32
    n_reads = [int(n_values) - c[25], c[25]]
    if not derived_pair_all:
        assert c[9] == n_reads[0], "slot 9 (pass-QC reads) missing: these are not superset counters"
    out = {"n_reads": n_reads}
    for name, k in (("n_secondary", 8), ("n_supp", 11), ("n_dup", 10), ("n_read1", 6), ("n_read2", 7),
                    ("n_pair_good", 12), ("n_sgltn", 13), ("n_pair_map", 14)):
        out[name] = [c[k], c[16 + k]]
    out["n_mapped"] = [n_reads[0] - c[2], n_reads[1] - c[18]]
    out["n_pair_all"] = [c[6] + c[7], c[22] + c[23]] if derived_pair_all else [c[0], c[16]]
    return out


def _percent(n: int, total: int) -> str:
    # benchmark/flagstats.cpp:73-78: "%.2f%%" of (float)n / total * 100.0, "N/A" for an empty class
    if total == 0:
        return "N/A"
    return "%.2f%%" % (float(np.float32(n) / np.float32(total)) * 100.0)


def samtools_flagstat_text(counters, n_values: int, derived_pair_all: bool = False) -> str:
    s = samtools_counts(counters, n_values, derived_pair_all)
    two = lambda k: "%d + %d" % (s[k][0], s[k][1])  # noqa: E731
    pct = lambda a, b: "(%s : %s)" % (_percent(s[a][0], s[b][0]), _percent(s[a][1], s[b][1]))  # noqa: E731
    lines = [
        two("n_reads") + " in total (QC-passed reads + QC-failed reads)",
        two("n_secondary") + " secondary",
        two("n_supp") + " supplementary",
        two("n_dup") + " duplicates",
        two("n_mapped") + " mapped " + pct("n_mapped", "n_reads"),
        two("n_pair_all") + " paired in sequencing" + (" (derived: read1 + read2)" if derived_pair_all else ""),
        two("n_read1") + " read1",
        two("n_read2") + " read2",
        two("n_pair_good") + " properly paired " + pct("n_pair_good", "n_pair_all"),
        two("n_pair_map") + " with itself and mate mapped",
        two("n_sgltn") + " singletons " + pct("n_sgltn", "n_pair_all"),
    ]
    return "\n".join(lines) + "\n"


def counter_table_text(counters) -> str:
    """The 15-row name / pass / fail table the reference's bench writes to stderr
    (benchmark/flagstats.cpp:340-342, names :100)."""
    from .pyflagstats import SAM_FLAG_NAMES
    c = np.asarray(counters).ravel()
    return "".join("%s\t%d\t%d\n" % (SAM_FLAG_NAMES[i], int(c[i]), int(c[16 + i])) for i in range(15))


def flagstat_report(values) -> str:
    """samtools-flagstat text of a host ``uint16`` array: one superset count on the GPU + the mapping above."""
    from . import _lib
    v = np.ascontiguousarray(values, dtype=np.uint16)
    out = np.zeros(32, dtype=np.uint64)
    _lib.check(_lib.lib().FLAGSTATS_u16_x64_superset(v.ctypes.data if v.size else None, v.size, out.ctypes.data),
               "FLAGSTATS_u16_x64_superset")
    return samtools_flagstat_text(out, v.size)


def flagstat_report_file(path: str, threads: int = 0) -> str:
    """samtools-flagstat text of a file, as ``bench decompress -i FILE -s`` (``.lz4`` / ``.zst`` block files,
    ``benchmark/flagstats.cpp:360-413,684-736``) or ``-S`` (any other name: a raw ``uint16`` file, ``:470-531``)
    prints it: decode + superset count on the GPU engine, then the mapping above."""
    from . import blockfile
    if str(path).endswith((".lz4", ".zst")):
        c, st = blockfile.flagstat_file(path, threads, superset=True)
    else:
        c, st = blockfile.flagstat_raw_file(path, superset=True)
    return samtools_flagstat_text(c, st["n_flags"])
