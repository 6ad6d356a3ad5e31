"""CPU: report layer (row f2) -- the mapping from 32 counters to the samtools-flagstat text -- against
(i) the samtools output the reference publishes for NA12878 (README.md:178-192), (ii) the reference
bench's OWN `decompress -s` stdout on the golden block files (tests/golden/blockfiles/manifest.json,
written by the reference binary) and (iii) the oracle's restatement of its samtools loop.  The HIP
side of the same row (counters from K1 -> identical text) is tests/test_gpu_report.py."""
import json
import os

import numpy as np

from conftest import GOLDEN

README_NA12878 = """824541892 + 0 in total (QC-passed reads + QC-failed reads)
0 + 0 secondary
5393628 + 0 supplementary
0 + 0 duplicates
805383403 + 0 mapped (97.68% : N/A)
819148264 + 0 paired in sequencing
409574132 + 0 read1
409574132 + 0 read2
781085884 + 0 properly paired (95.35% : N/A)
797950890 + 0 with itself and mate mapped
2038885 + 0 singletons (0.25% : N/A)
"""


def superset_from_oracle(oracle, flags):
    """What the superset entry points must deliver: FLAGSTAT_scalar's 32 slots + slots 0/16 (primary paired
    reads by QC class) + slot 9 (pass-QC reads), built here from the two oracle restatements."""
    c = oracle.flagstat_hist(flags).copy()
    s = oracle.samtools_counts(flags)
    c[0], c[16] = s["n_pair_all"]
    c[9] = s["n_reads"][0]
    return c


def test_text_reproduces_readme_na12878_output():
    from libflagstats_amd.report import samtools_flagstat_text
    n = 824541892
    c = np.zeros(32, dtype=np.uint64)
    c[0] = 819148264          # primary paired reads (superset slot)
    c[9] = n                  # pass-QC reads (superset slot)
    c[2] = n - 805383403      # unmapped = total - mapped
    c[6] = c[7] = 409574132
    c[11] = 5393628
    c[12] = 781085884
    c[13] = 2038885
    c[14] = 797950890
    assert samtools_flagstat_text(c, n) == README_NA12878
    # scalar-exact counters (no superset slots): n_pair_all falls back to read1 + read2 and the line says so
    c[0] = c[9] = 0
    t = samtools_flagstat_text(c, n, derived_pair_all=True)
    assert "819148264 + 0 paired in sequencing (derived: read1 + read2)\n" in t


def test_oracle_samtools_restatements_agree(oracle_mod):
    for seed, hi in ((1, 4096), (2, 65536)):
        flags = np.random.RandomState(seed).randint(0, hi, 5000).astype(np.uint16)
        assert oracle_mod.samtools_counts(flags) == oracle_mod.samtools_counts_python(flags)


def test_counts_match_samtools_loop_on_any_flags(oracle_mod):
    from libflagstats_amd.report import counter_table_text, samtools_counts, samtools_flagstat_text
    for flags in (oracle_mod.generate(oracle_mod.GEN_NA12878, 3, 1, 0, 40000),
                  np.random.RandomState(2).randint(0, 65536, 30000).astype(np.uint16),   # arbitrary bit patterns
                  np.arange(65536, dtype=np.uint16)):
        want = oracle_mod.samtools_counts(flags)
        got = samtools_counts(superset_from_oracle(oracle_mod, flags), flags.size)
        for k, v in got.items():
            assert want[k] == v, k
        assert samtools_flagstat_text(superset_from_oracle(oracle_mod, flags), flags.size) == oracle_mod.samtools_text(want)
    t = counter_table_text(oracle_mod.flagstat_c(flags)).splitlines()
    assert len(t) == 15 and t[2].startswith("FUNMAP\t") and t[14].startswith("n_pair_map\t")


def test_text_equals_the_reference_programs_own_stdout(oracle_mod):
    """`bench decompress -s` (benchmark/flagstats.cpp:577-588) was run by the reference binary on each golden
    block file; the report layer must print the same bytes from superset counters of the same flags."""
    import sys
    sys.path.insert(0, GOLDEN)
    from make_blockfiles import recipe_input
    from libflagstats_amd.report import samtools_flagstat_text
    man = json.load(open(os.path.join(GOLDEN, "blockfiles", "manifest.json")))
    seen = 0
    for name, e in man["files"].items():
        if not e["reference_decompress_s_stdout"]:
            continue
        flags = recipe_input(e["n_flags"], e["seed"])
        assert samtools_flagstat_text(superset_from_oracle(oracle_mod, flags), flags.size) == e["reference_decompress_s_stdout"], name
        seen += 1
    assert seen >= 3
