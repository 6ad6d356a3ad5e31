#!/usr/bin/env python3
"""Counterpart of the reference's `utility` (benchmark/utility.cpp): decimal FLAG text on stdin -> uint16
binary on stdout, e.g.  samtools view FILE | cut -f 2 | python tools/utility.py > FLAGS.bin"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libflagstats_amd import textio  # noqa: E402

if __name__ == "__main__":
    sys.stdout.buffer.write(textio.flags_from_text(sys.stdin.buffer.read()).tobytes())
