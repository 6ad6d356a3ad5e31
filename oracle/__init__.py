"""CPU oracle for the flagstat hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the checker.  Nothing under
``libflagstats_amd/`` imports it.

Parity status: PINNED -- the restatements here are checked against the
reference's own ``FLAGSTAT_scalar`` compiled from ``/root/reference``
(``oracle/_ref/libflagstats_ref.so``, recipe in ``oracle/Makefile``) and against
the golden vectors in ``tests/golden/`` that the same build produced.

Reference lines restated: ``libflagstats.h:118-142`` (per-flag rule),
``:170-176`` (loop), ``python/libflagstats.pyx:8-37`` (Python dict).
"""
from .pyoracle import (  # noqa: F401
    GEN_NA12878,
    GEN_RAMP,
    GEN_UNIFORM,
    LIVE_SLOTS,
    SAM_FLAG_NAMES,
    build,
    flagstat_c,
    flagstat_generated,
    flagstat_hist,
    flagstat_mt,
    flagstat_numpy,
    flagstat_python,
    generate,
    load_c,
    load_ref,
    pospopcnt,
    pospopcnt_numpy,
    ref_pospopcnt,
    pyflagstats_dict,
    ref_call,
    samtools_counts,
    samtools_counts_python,
    samtools_text,
)
