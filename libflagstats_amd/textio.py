"""FLAG text -> uint16 array: the reference's ``utility`` input maker (``benchmark/utility.cpp:9-16``,
``samtools view FILE | cut -f 2 | utility > FLAGS.bin``), row f3 of SURVEY.md section 8."""
from __future__ import annotations

import numpy as np

from . import _lib


def flags_from_text(text: bytes) -> np.ndarray:
    """One decimal FLAG per line (``std::getline`` + ``atoi`` rules of the reference) -> ``uint16`` array."""
    if isinstance(text, str):
        text = text.encode()
    lib = _lib.lib()
    n = int(lib.FLAGSTATS_text_count_lines(text, len(text)))
    out = np.empty(n, dtype=np.uint16)
    got = lib.FLAGSTATS_text_to_u16(text, len(text), out.ctypes.data if n else None, n)
    if got != n:
        _lib.check(-1, "FLAGSTATS_text_to_u16")
    return out
