#!/usr/bin/env python3
"""Compare two `make -C libflagstats_amd/csrc asm` outputs (flagstat_kernels.s) kernel by kernel: for every kernel present in
both, the instruction stream between its label and its .Lfunc_end must be the same text, and so must its resource block
(VGPRs, SGPRs, LDS, scratch).  Proof that a refactoring of the source moved no instruction.

    python3 tools/kernel_isa_diff.py before.s after.s
"""
import hashlib
import re
import sys


def kernels(path):
    out, cur, name = {}, None, None
    meta = {}
    for ln in open(path):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m and cur is None:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if ln.startswith(".Lfunc_end"):
                out[name] = cur
                cur = None
                continue
            s = ln.split(";")[0].rstrip() if not ln.lstrip().startswith(";") else ""
            if s.strip():
                cur.append(s.strip())
    # resource usage from the .amdhsa_ directives of each kernel descriptor
    text = open(path).read()
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", text, re.S):
        meta[m.group(1)] = [x.strip() for x in m.group(2).splitlines() if x.strip()]
    return out, meta


def demangle_hint(name):
    m = re.search(r"flagstat_countILi(\d+)ELb(\d)ELb(\d)ELb(\d)ELi(\d+)E", name)
    if m:
        return "fsk::flagstat_count<%s, %s, %s, %s, %s>" % (m.group(1), "true" if m.group(2) == "1" else "false",
                                                             "true" if m.group(3) == "1" else "false",
                                                             "true" if m.group(4) == "1" else "false", m.group(5))
    if "flagstat_finalize" in name:
        return "fsk::flagstat_finalize"
    return name[:60]


def main():
    a, am = kernels(sys.argv[1])
    b, bm = kernels(sys.argv[2])
    rc = 0
    for name in sorted(set(a) | set(b)):
        if name not in a or name not in b:
            print("%-50s only in %s" % (demangle_hint(name), sys.argv[1] if name in a else sys.argv[2]))
            continue
        same = a[name] == b[name]
        same_meta = am.get(name) == bm.get(name)
        h = hashlib.sha256("\n".join(b[name]).encode()).hexdigest()[:16]
        print("%-50s %6d instructions / labels, body %s, kernel descriptor %s, sha256(body) %s" % (
            demangle_hint(name), len(b[name]), "IDENTICAL" if same else "DIFFERS", "identical" if same_meta else "DIFFERS", h))
        if not same:
            rc = 1
            for i, (x, y) in enumerate(zip(a[name], b[name])):
                if x != y:
                    print("    first difference at line %d:\n      before: %s\n      after:  %s" % (i, x, y))
                    break
            print("    lengths: %d -> %d" % (len(a[name]), len(b[name])))
    return rc


if __name__ == "__main__":
    sys.exit(main())
