"""libflagstats_amd -- MI355X (gfx950) engine for libflagstats' flagstat hot path.

Only what the path needs:

* ``csrc/``          hand-written HIP kernels + the C-ABI shim -> ``libflagstats_hip.so``
                     (declared in ``include/libflagstats_hip.h``)
* ``pyflagstats``    mirror of the reference's Python entry point
                     (``python/libflagstats.pyx``): ``flagstats(values)``
* ``device``         device-resident arrays, on-device input makers, torch interop
* ``dist``           shard + single all-reduce for multi-GPU runs

The hot path has no CPU fallback: importing the compute entry points without the
built extension raises.
"""
from .pyflagstats import SAM_FLAG_NAMES, flagstats, flagstats_x64  # noqa: F401

__all__ = ["flagstats", "flagstats_x64", "SAM_FLAG_NAMES"]
