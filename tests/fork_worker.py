"""Worker functions for tests/test_gpu_fork.py (importable from a spawned child)."""
import numpy as np


def flagstats_in_child(conn, seed, n):
    """Runs pyflagstats.flagstats on a seeded array in the calling (child) process and sends back either the 32
    counters or the exception it got."""
    try:
        from libflagstats_amd import pyflagstats
        a = np.random.RandomState(seed).randint(0, 65536, n).astype(np.uint16)
        conn.send(("ok", [int(v) for v in pyflagstats.counters_u64(a)]))
    except Exception as e:  # noqa: BLE001 -- the text is what the test is about
        conn.send(("error", type(e).__name__, str(e)))
    finally:
        conn.close()
