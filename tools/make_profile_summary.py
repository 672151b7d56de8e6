#!/usr/bin/env python3
"""profiles/<round>/summary.md from the files tools/collect_profiles.sh produced: tools/make_profile_summary.py r02"""
import csv, json, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", tag)
rd = lambda f: open(os.path.join(root, f)).read()
last = lambda f: json.loads(rd(f).strip().splitlines()[-1])
out = ["# Round %s profile summary (1x MI355X, `tools/collect_profiles.sh %s`)" % (tag[1:].lstrip("0"), tag), ""]
out += ["`rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline` (default: batch 4096, 1 warm-up + 3 timed steps + the NTT-only section):", "",
        "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
def short(name):
    m = re.search(r"(\w+_kernel(<[^(]*>)?)\(", name)
    return m.group(1) if m else re.sub(r"^void |\(.*$", "", name)[:100]
rows = list(csv.DictReader(open(os.path.join(root, "bench_default_kernel_stats.csv"))))
for r in rows[:14]:
    out.append("| %s | %s | %.2f | %.1f | %.2f |" % (short(r["Name"]), r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
b = last("bench_with_traffic.jsonl")
d = last("bench_default.jsonl")
rf, cb = b["roofline"], d["cpu_baseline"]
out += ["", "Bench line of the same build (`bench_with_traffic.jsonl`, traffic measured by this run's own `--measure-traffic` PMC passes):",
        "- value **%.0f ct-mul+relin/s**, %.1f ms per step of %d pairs, verified items %s vs the CPU oracle: %s" % (
            b["value"], b["ms_per_step"], b["config"].get("batch", b["config"].get("global_batch", 4096)), b["verified_items"], b["verified_vs_oracle"]),
        "- roofline (dominant kernel `ntt_fwd_half`): achieved %.0f GB/s of %.0f = **%.3f**; %d launches, avg %.3f ms, %.0f rows per launch; PMC traffic / algorithmic bytes = %.3f" % (
            rf["achieved"], rf["peak"], rf["frac"], rf["launches"], rf["avg_launch_ms"], rf["rows_per_launch"],
            (rf["traffic"] or 0) / rf["algorithmic_bytes_per_launch"]),
        "- NTT-only section: %.2f M forward NTT/s = %.3f of the HBM roofline" % (b["ntt"]["forward_ntt_per_s"] / 1e6, b["ntt"]["hbm_roofline_frac"]),
        "- default run (`bench_default.jsonl`): %.0f ct/s, roofline %.3f, traffic source: %s" % (d["value"], d["roofline"]["frac"], d["roofline"].get("traffic_source")),
        "- CPU baseline (oracle built -O3 -march=native on the box): %s; %.1f ct/s on %d threads, %.2f on one; host has %d physical cores -> linear all-core projection %.0f ct/s, GPU/CPU = %.1fx projected, %.0fx against the measured %d threads" % (
            cb["cpu_model"], cb["value"], cb["cores"], cb["value_1thread"], cb["physical_cores"], cb["projected_all_physical_cores_linear"],
            cb["gpu_over_cpu_all_physical_cores_projected"], cb["gpu_over_cpu_%dthreads_measured" % cb["cores"]], cb["cores"]),
        "- kernel time shares in the timed steps: %s" % json.dumps(d["kernel_time_shares"]), ""]
out += ["PMC (`traffic.json`; separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over `bench.py --batch 256 --steps 2 --warmup 0 --ntt-polys 0`; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md), summed over every launch of the kernel in the step:",
        "```", rd("traffic.json").strip(), "```",
        "The inverse figure is what the fused tensor product costs: an output row of c_1 reads four forward-transformed rows, c_0 / c_2 two each (algorithmic 29.3 N bytes per output row instead of 16 N).", ""]
out += ["Per-kernel times of one step (HIP events; `step_profile_b1024.txt`: config 3 over 1024 pairs; `step_profile_side_configs.txt`: config 4 rotate and multiply+relinearize over 1024 ciphertexts, config 5 over 256, config 1 over 4096):",
        "```", "\n".join(l for l in rd("step_profile_b1024.txt").splitlines() if l.startswith("cfg")), rd("step_profile_side_configs.txt").strip(), "```",
        "Round 1, config 3, same kind of box: fwd 18.3, inv 10.9, floor 6.7, tensor 5.45, lift 3.9, mac 3.8, moddown 2.7 = 51.8 ms.", ""]
out += ["Side configs (`configs_1gpu.jsonl`):", "```"]
for l in rd("configs_1gpu.jsonl").splitlines():
    if l.startswith('{"config": "cfg') and "PCIe" not in l and "small batches" not in l:
        j = json.loads(l)
        out.append(j["config"] + ": " + ", ".join("%s %.4g" % (k, v) for k, v in j.items() if k != "config" and isinstance(v, (int, float))))
out += ["```", "", "FP64 against integer NTT instances on 50-bit primes, same box back to back (`ntt_fp64_ab.txt`; inverse = half + top kernels):", "```", rd("ntt_fp64_ab.txt").strip(), "```",
        "Standalone transforms as in round 1 (`ntt_only.txt`: N = 2^15 on config 3's 55-bit primes = integer instances; 2^14 / 2^16 on 50-bit primes = FP64):", "```"]
out += [l[:100] for l in rd("ntt_only.txt").splitlines() if l.startswith("logn")]
out += ["```", "",
        "Other files: `cfg4_rotate_kernel_stats.csv` (rocprofv3 kernel trace of the config-4 rotate step: this is where the 2-D zero fill of apply_galois was found), `ntt_store_pattern.txt` (the final-round store pattern experiment and the FP64 / store-exchange A/Bs), `ubench_fp64.txt` (instruction-rate microbenchmarks behind the FP64 decision), `ntt_twiddle_shuffle_ab.txt` (wavefront-shuffle twiddles: -26 %, dropped), `fuzz_parity_160.txt`, the f1-f4 and PCIe-inclusive rows in `configs_1gpu.jsonl`."]
open(os.path.join(root, "summary.md"), "w").write("\n".join(out) + "\n")
print("wrote", os.path.join(root, "summary.md"))
