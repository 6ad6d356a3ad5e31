"""Identity of the DEVICE code of K1 / K2 inside a built ``libflagstats_hip.so``.

``bench.py`` reports it as ``config.kernel_source_id`` and ``profiles/traffic.json`` carries the id of the build its PMC
figures were measured on: a traffic figure is only reported for the kernel it was measured on.  Until r05 the id hashed three
SOURCE files, so a host-side edit of ``flagstat_kernels.hip`` invalidated measurements of byte-identical device code.  Now it
is the sha256 over the ``.text`` and ``.rodata`` sections (instructions, kernel descriptors, constant tables) of the gfx950
code object that defines ``fsk::flagstat_count`` -- found by walking the library's ``.hip_fatbin`` section (clang offload
bundles, one per translation unit).  No tool of the ROCm installation is needed: plain ELF parsing.
"""
from __future__ import annotations

import hashlib
import os
import struct

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _sections(elf: bytes):
    """{name: (offset, size)} of an ELF64 little-endian image."""
    if elf[:4] != b"\x7fELF" or elf[4] != 2 or elf[5] != 1:
        raise ValueError("not a little-endian ELF64 image")
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    heads = []
    for i in range(shnum):
        name, _type, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", elf, shoff + i * shentsize)
        heads.append((name, off, size, _type))
    stroff, strsize = heads[shstrndx][1], heads[shstrndx][2]
    strtab = elf[stroff:stroff + strsize]
    out = {}
    for name, off, size, typ in heads:
        end = strtab.index(b"\0", name)
        out[strtab[name:end].decode()] = (off, 0 if typ == 8 else size)   # SHT_NOBITS occupies no file bytes
    return out


def _code_objects(so: bytes):
    """Every gfx950 code object (bytes) bundled in the host library."""
    secs = _sections(so)
    if ".hip_fatbin" not in secs:
        raise ValueError("no .hip_fatbin section: not a HIP library")
    off, size = secs[".hip_fatbin"]
    fat = so[off:off + size]
    pos = fat.find(_MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", fat, pos + len(_MAGIC))
        at = pos + len(_MAGIC) + 8
        for _ in range(n):
            eoff, esize, tlen = struct.unpack_from("<QQQ", fat, at)
            triple = fat[at + 24:at + 24 + tlen].decode(errors="replace")
            at += 24 + tlen
            if "gfx950" in triple and esize:
                yield fat[pos + eoff:pos + eoff + esize]
        pos = fat.find(_MAGIC, pos + len(_MAGIC))


def kernel_id(so_path: str | None = None) -> str:
    """16 hex digits identifying the device code of the K1 / K2 translation unit of ``so_path`` (default: the in-tree library)."""
    if so_path is None:
        so_path = os.environ.get("FLAGSTATS_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libflagstats_hip.so")
    with open(so_path, "rb") as f:
        so = f.read()
    for co in _code_objects(so):
        secs = _sections(co)
        names = b""
        for tab in (".strtab", ".dynstr"):
            if tab in secs:
                names += co[secs[tab][0]:secs[tab][0] + secs[tab][1]]
        # the product's K1: fsk::flagstat_count (the measurement build also carries fskt::flagstat_count in ANOTHER code object)
        if b"_ZN3fsk14flagstat_count" not in names:
            continue
        h = hashlib.sha256()
        for sec in (".text", ".rodata"):
            o, s = secs.get(sec, (0, 0))
            h.update(sec.encode() + struct.pack("<Q", s) + co[o:o + s])
        return h.hexdigest()[:16]
    raise ValueError("%s carries no gfx950 code object that defines fsk::flagstat_count" % so_path)


if __name__ == "__main__":
    import sys
    print(kernel_id(sys.argv[1] if len(sys.argv) > 1 else None))
