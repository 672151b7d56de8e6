#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel stats + separate --pmc FETCH_SIZE / WRITE_SIZE passes) into
profiles/<round>/summary.md and profiles/traffic.json.

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE
reports exactly half the bytes of a wide (16 B/lane) coalesced read stream, so it is doubled; WRITE_SIZE is
exact for 16 B/lane streaming stores. Both are collected in their own passes (they do not fit one pass)."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    if "sealhip" not in name:
        return None
    s = name.split("sealhip::(anonymous namespace)::")[1]
    return s.split("(")[0]


def pmc(path):
    agg = collections.defaultdict(list)
    files = glob.glob(path, recursive=True)
    if not files:
        raise SystemExit("no rocprofv3 csv under " + path)
    for r in csv.DictReader(open(files[0])):
        s = short(r["Kernel_Name"])
        if s:
            agg[(s, int(r["Grid_Size"]), int(r["Workgroup_Size"]))].append(float(r["Counter_Value"]))
    return agg


def main(prof_dir, out_dir, rows_per_block_ntt=4 * 512):
    os.makedirs(out_dir, exist_ok=True)
    fetch = pmc(os.path.join(prof_dir, "pmc_fetch/**/*_counter_collection.csv"))
    write = pmc(os.path.join(prof_dir, "pmc_write/**/*_counter_collection.csv"))
    lines = ["| kernel | grid | launches | FETCH_SIZE KB (raw) | read bytes (x2, gfx950) | WRITE_SIZE KB | HBM bytes / launch |",
             "|---|---|---|---|---|---|---|"]
    traffic = {}
    for key in sorted(fetch):
        f = sum(fetch[key]) / len(fetch[key])
        w = sum(write.get(key, [0])) / max(1, len(write.get(key, [0])))
        total = (2 * f + w) * 1024
        lines.append("| %s | %d | %d | %.0f | %.0f | %.0f | %.0f |" % (key[0], key[1], len(fetch[key]), f, 2 * f * 1024,
                                                                   w, total))
        if key[0].startswith("ntt_pass_kernel"):
            tag = "ntt_fwd_pass" if "<0>" in key[0] else "ntt_inv_pass"
            rows = key[1] / key[2] / 4  # N=2^15: 4 tiles per row
            traffic.setdefault(tag, []).append((rows, total / rows))
        elif key[0].startswith("ntt_fwd_half_kernel") or key[0].startswith("ntt_inv_half_kernel"):
            tag = "ntt_fwd_half" if "fwd" in key[0] else "ntt_inv_half"
            rows = key[1] / key[2] / 2  # two workgroups per row (grid rounded up to 8-row groups)
            traffic.setdefault(tag, []).append((rows, total / rows))
    # launches that skip rows (key-switch mod-down: only the special-prime rows are transformed) would dilute the
    # per-row figure: take it from the largest launch of each kernel, where every row is transformed
    out = {k: {"hbm_bytes_per_row_per_launch": max(v)[1], "rows_in_that_launch": max(v)[0]} for k, v in traffic.items()}
    json.dump(out, open(os.path.join(os.path.dirname(out_dir.rstrip("/")), "traffic.json"), "w"), indent=1)
    stats = glob.glob(os.path.join(prof_dir, "stats/**/*_kernel_stats.csv"), recursive=True)
    with open(os.path.join(out_dir, "summary.md"), "w") as fo:
        fo.write("# rocprofv3 summary\n\n## PMC (separate passes: --pmc FETCH_SIZE, --pmc WRITE_SIZE)\n\n")
        fo.write("\n".join(lines) + "\n\n")
        fo.write("NTT kernels, HBM bytes per RNS row per launch (N=2^15; algorithmic = 16*N = 524288 B per row for the "
                 "single-pass kernels, 8*N = 262144 B per row per launch of the two-pass kernel): %s\n\n" % json.dumps(out))
        if stats:
            fo.write("## kernel-trace --stats (same bench command)\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
            for r in csv.DictReader(open(stats[0])):
                s = short(r["Name"]) or r["Name"][:60]
                fo.write("| %s | %s | %.3f | %.1f | %s |\n" % (s, r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                          float(r["AverageNs"]) / 1e3, r["Percentage"]))
    print(open(os.path.join(out_dir, "summary.md")).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
