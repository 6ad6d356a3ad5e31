"""``import pyflagstats; pyflagstats.flagstats(values)`` -- the reference's Python
module name (python/setup.py:34, python/libflagstats.pyx:8), served by the
MI355X engine in ``libflagstats_amd``."""
from libflagstats_amd.pyflagstats import SAM_FLAG_NAMES, flagstats, flagstats_x64  # noqa: F401
