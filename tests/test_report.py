"""CPU: report layer (row f2) against the samtools output the reference publishes for NA12878
(README.md:178-192) and against a direct restatement of the samtools loop on random flags."""
import numpy as np


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


def test_text_reproduces_readme_na12878_output():
    from libflagstats_amd.report import samtools_flagstat_text
    n = 824541892
    c = np.zeros(32, dtype=np.uint64)
    c[2] = n - 805383403      # unmapped = total - mapped
    c[6] = c[7] = 409574132
    c[11] = 5393628
    c[12] = 781085884
    c[13] = 2038885
    c[14] = 797950890
    assert samtools_flagstat_text(c, n) == README_NA12878


def samtools_loop(flags):
    """The reference's flagstat_loop macro (benchmark/flagstats.cpp:51-70), restated."""
    keys = ["n_reads", "n_mapped", "n_pair_all", "n_pair_map", "n_pair_good", "n_sgltn", "n_read1", "n_read2", "n_dup",
            "n_secondary", "n_supp"]
    s = {k: [0, 0] for k in keys}
    for c in flags:
        c = int(c)
        w = 1 if c & 512 else 0
        s["n_reads"][w] += 1
        if c & 256:
            s["n_secondary"][w] += 1
        elif c & 2048:
            s["n_supp"][w] += 1
        elif c & 1:
            s["n_pair_all"][w] += 1
            if (c & 2) and not (c & 4):
                s["n_pair_good"][w] += 1
            if c & 64:
                s["n_read1"][w] += 1
            if c & 128:
                s["n_read2"][w] += 1
            if (c & 8) and not (c & 4):
                s["n_sgltn"][w] += 1
            if not (c & 4) and not (c & 8):
                s["n_pair_map"][w] += 1
        if not (c & 4):
            s["n_mapped"][w] += 1
        if c & 1024:
            s["n_dup"][w] += 1
    return s


def test_counts_match_samtools_loop(oracle_mod):
    from libflagstats_amd.report import counter_table_text, samtools_counts
    # well-formed paired data (every paired read is read1 xor read2): all fields must agree
    flags = oracle_mod.generate(oracle_mod.GEN_NA12878, 3, 1, 0, 40000)
    got = samtools_counts(oracle_mod.flagstat_c(flags), flags.size)
    want = samtools_loop(flags)
    for k, v in want.items():
        assert got[k] == v, k
    # arbitrary bit patterns: everything but the derived n_pair_all agrees
    flags = np.random.RandomState(2).randint(0, 65536, 30000).astype(np.uint16)
    got = samtools_counts(oracle_mod.flagstat_c(flags), flags.size)
    want = samtools_loop(flags)
    for k, v in want.items():
        if k != "n_pair_all":
            assert got[k] == v, k
    t = counter_table_text(oracle_mod.flagstat_c(flags)).splitlines()
    assert len(t) == 15 and t[2].startswith("FUNMAP\t") and t[14].startswith("n_pair_map\t")
