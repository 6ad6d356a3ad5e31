#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/gpu_profile.sh (gpurun_out/prof_*) into the small, tracked
summaries under profiles/<round>/ and into profiles/traffic.json (what bench.py reports as
roofline.traffic).  HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md, section HBM:
separate --pmc passes; on gfx950 FETCH_SIZE (KB) reports exactly half of the bytes of a wide coalesced
streaming read, WRITE_SIZE is exact:  bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.

    python tools/summarize_profile.py gpurun_out profiles/r02 [--flags 4294967296]
"""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_id():
    """id of the K1 / K2 device code of the in-tree library (the one that was profiled): libflagstats_amd/kernel_id.py"""
    sys.path.insert(0, ROOT)
    from libflagstats_amd.kernel_id import kernel_id
    return kernel_id()


def find(src, sub, pattern):
    # gpurun merges every call's outputs into the same local directory: take the NEWEST match, not the first
    hits = glob.glob(os.path.join(src, sub, "**", pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--flags", type=int, default=2 ** 32)
    ap.add_argument("--kernel", default="flagstat_count")
    ap.add_argument("--tag", default="")
    ap.add_argument("--sub", default="prof_", help="prefix of the rocprofv3 output directories under src (prof_trace, prof_pmc_fetch, ...)")
    ap.add_argument("--name", default="bench", help="prefix of the summary files written")
    ap.add_argument("--min-grid", type=int, default=256 * 64, help="PMC rows of launches with fewer work-items are ignored")
    args = ap.parse_args()
    os.makedirs(args.dst, exist_ok=True)
    tag = ("_" + args.tag) if args.tag else ""

    # kernel trace: keep the stats table (a few lines) and the full-size launches' own average
    stats = find(args.src, args.sub + "trace", "*kernel_stats.csv")
    if stats:
        rows = list(csv.reader(open(stats)))
        with open(os.path.join(args.dst, "%s_kernel_stats%s.csv" % (args.name, tag)), "w") as f:
            for r in rows[:8]:
                f.write(",".join('"%s"' % c for c in r) + "\n")
    trace = find(args.src, args.sub + "trace", "*kernel_trace.csv")
    launches = None
    if trace:
        durs = []
        name = None
        for r in csv.DictReader(open(trace)):
            if args.kernel in r["Kernel_Name"] and "finalize" not in r["Kernel_Name"]:
                durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                name = r["Kernel_Name"]
        if durs:
            durs.sort()
            big = [d for d in durs if d > 0.5 * durs[-1]]   # full-size launches only (parity/prefix launches are shorter)
            launches = {"kernel": name, "launches": len(durs), "full_size_launches": len(big),
                        "avg_ns_full_size": sum(big) / len(big), "median_ns_full_size": big[len(big) // 2],
                        "min_ns": big[0], "max_ns": big[-1]}

    counters = {}
    vgpr = sgpr = lds = None
    kname = None
    for sub in (args.sub + "pmc_fetch", args.sub + "pmc_write", args.sub + "pmc_sq"):
        cc = find(args.src, sub, "*counter_collection.csv")
        if not cc:
            continue
        for r in csv.DictReader(open(cc)):
            if args.kernel in r["Kernel_Name"] and "finalize" not in r["Kernel_Name"] and int(r["Grid_Size"]) >= args.min_grid:
                counters.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                kname, vgpr, sgpr, lds = r["Kernel_Name"], r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"]
    summary = {}
    for k, v in sorted(counters.items()):
        # the bench's small prefix / parity launches read far less: keep the full-size launches
        top = max(v)
        full = [x for x in v if x > 0.5 * top] if k in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU") else v
        summary[k] = {"launches": len(full), "mean": sum(full) / len(full), "min": min(full), "max": max(full)}
    out = {"kernel": kname, "VGPR_Count": vgpr,
           "VGPR_Count_note": "as rocprofv3 prints it: the kernel descriptor's granule count x 4; gfx950 allocates wave64 VGPRs in granules "
                              "of 8, so the kernel holds twice this many (the ISA metadata's .vgpr_count)",
           "SGPR_Count": sgpr, "LDS_Block_Size": lds, "kernel_source_id": kernel_source_id(),
           "trace": launches, "counters": summary}
    if "SQ_INSTS_VALU" in summary and args.flags and args.kernel == "flagstat_count":
        waves = 256 * 4
        steps = args.flags / 16384
        out["valu_per_wave_step"] = summary["SQ_INSTS_VALU"]["mean"] / (steps * 4)
        out["valu_per_flag_per_lane"] = out["valu_per_wave_step"] / 64.0
        out["note_valu"] = "SQ_INSTS_VALU / (flags / 16384 steps x 4 waves per step); %d waves in the grid" % waves
    if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary and args.flags:
        out["hbm_bytes_per_launch"] = int(round(2 * summary["FETCH_SIZE"]["mean"] * 1024 + summary["WRITE_SIZE"]["mean"] * 1024))
        out["algorithmic_bytes_per_launch"] = 2 * args.flags
        out["traffic_ratio"] = out["hbm_bytes_per_launch"] / (2.0 * args.flags)
        if launches:
            out["achieved_GBs"] = 2.0 * args.flags / launches["avg_ns_full_size"]
            out["frac_of_8TBs"] = out["achieved_GBs"] / 8000.0
    with open(os.path.join(args.dst, "%s_pmc_%s%s.json" % (args.name, args.kernel, tag)), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))

    if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary and args.kernel == "flagstat_count" and args.name == "bench":
        hbm = int(round(2 * summary["FETCH_SIZE"]["mean"] * 1024 + summary["WRITE_SIZE"]["mean"] * 1024))
        t = {"flags_per_launch": args.flags, "kernel": kname, "kernel_source_id": kernel_source_id(),
             "FETCH_SIZE_KB_mean": summary["FETCH_SIZE"]["mean"], "WRITE_SIZE_KB_mean": summary["WRITE_SIZE"]["mean"],
             "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": 2 * args.flags,
             "source": os.path.join(os.path.basename(args.dst.rstrip("/")), "bench_pmc_%s%s.json" % (args.kernel, tag)),
             "derivation": "MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide "
                           "coalesced (16 B/lane) streaming read, WRITE_SIZE is exact: bytes = 2*FETCH_SIZE*1024 + "
                           "WRITE_SIZE*1024. Separate --pmc passes of bench.py (tools/gpu_profile.sh)."}
        with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
            json.dump(t, f, indent=1)
        print("traffic.json:", hbm, "bytes per launch vs", 2 * args.flags, "algorithmic")


if __name__ == "__main__":
    sys.exit(main())
