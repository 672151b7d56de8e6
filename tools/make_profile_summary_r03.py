#!/usr/bin/env python3
"""profiles/r03/summary.md from the files tools/collect_profiles.sh r03 produced (+ the round's A/B records):
tools/make_profile_summary_r03.py"""
import csv, json, os, re

tag = "r03"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", tag)
rd = lambda f: open(os.path.join(root, f)).read()
last = lambda f: json.loads(rd(f).strip().splitlines()[-1])


def short(name):
    m = re.search(r"(\w+_kernel(<[^(]*>)?)\(", name)
    return m.group(1) if m else re.sub(r"^void |\(.*$", "", name)[:100]


out = ["# Round 3 profile summary (1x MI355X, `tools/collect_profiles.sh r03`)", "",
       "`rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline` (default: config 3, batch 4096, 1 warm-up + 3 timed "
       "steps + the NTT-only section):", "", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
rows = list(csv.DictReader(open(os.path.join(root, "bench_default_kernel_stats.csv"))))
for r in rows[:14]:
    out.append("| %s | %s | %.2f | %.1f | %.2f |" % (short(r["Name"]), r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                   float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
b, d = last("bench_with_traffic.jsonl"), last("bench_default.jsonl")
rf, cb, pr = b["roofline"], d["cpu_baseline"], b["pipeline_roofline"]
tr = json.loads(rd("traffic.json"))
fw = tr["ntt_fwd_half"]
out += ["", "Bench line of the same build (`bench_with_traffic.jsonl`, traffic measured by this run's own `--measure-traffic` PMC passes):",
        "- value **%.0f ct-mul+relin/s**, %.1f ms per step of %d pairs, verified items %s vs the CPU oracle: %s" % (
            b["value"], b["ms_per_step"], b["config"]["global_batch"], b["verified_items"], b["verified_vs_oracle"]),
        "- roofline (dominant kernel `ntt_fwd_half`): achieved %.0f GB/s of %.0f = **%.3f**; %d launches, avg %.3f ms, %.0f rows per launch; "
        "PMC traffic / algorithmic bytes = %.3f" % (rf["achieved"], rf["peak"], rf["frac"], rf["launches"], rf["avg_launch_ms"],
                                                   rf["rows_per_launch"], (rf["traffic"] or 0) / rf["algorithmic_bytes_per_launch"]),
        "- forward NTT per row from the counters: read %.1f KB (x%.3f of 262.1), **written %.1f KB (x%.4f of 262.1)** -- round 2 wrote 317.5 KB "
        "(x1.21): the excess was the 16-byte nontemporal pieces at a 32-byte stride reaching the fabric as partial writes before "
        "their neighbours arrived; with the register transposition every store instruction covers contiguous memory and the "
        "excess is gone. Reads: below the algorithmic 262 KB on average since the key-switch digit launches run four readers of a source "
        "row on one XCD (they find it in L2); the in-place launches read ~292 KB (twiddle tables, the sibling's half where it missed L2)" % (
            fw["read_bytes_per_row"] / 1e3, fw["read_bytes_per_row"] / 262144, fw["write_bytes_per_row"] / 1e3, fw["write_bytes_per_row"] / 262144),
        "- NTT-only section (the BASELINE `forward-NTT/s` metric, canonical `ntt_negacyclic_harvey`, %d rows per launch): %.2f M forward NTT/s = **%.3f** "
        "of the HBM roofline" % (b["ntt"]["rows"], b["ntt"]["forward_ntt_per_s"] / 1e6, b["ntt"]["hbm_roofline_frac"]),
        "- pipeline roofline (SURVEY 8d): compulsory %.1f MB per pair -> %.3f of HBM; NTT-equivalent %.1f MB (%d rows) -> %.3f; measured "
        "(PMC, all kernels of the step) %.1f MB per pair = %.1fx compulsory -> %.3f of HBM" % (
            pr["compulsory_bytes_per_unit"] / 1e6, pr["compulsory_frac_of_hbm"], pr["ntt_equivalent_bytes_per_unit"] / 1e6, pr["ntt_rows_per_unit"],
            pr["ntt_equivalent_frac_of_hbm"], (pr["measured_hbm_bytes_per_unit"] or 0) / 1e6, pr.get("measured_over_compulsory", 0),
            pr.get("measured_frac_of_hbm", 0)),
        "- top kernels against their own algorithmic bytes (HIP events): " + "; ".join(
            "%s %.2f ms/step = %.3f" % (k["kernel"], k["ms_per_step"], k["frac_of_hbm"] or 0) for k in b["kernels"]),
        "- default run (`bench_default.jsonl`, what the driver runs): %.0f ct/s, roofline %.3f, NTT section %.3f, traffic source: %s" % (
            d["value"], d["roofline"]["frac"], d["ntt"]["hbm_roofline_frac"], d["roofline"].get("traffic_source")),
        "- CPU baseline (oracle built -O3 -march=native on the box): %s; %.1f ct/s on %d threads, %.2f on one; host has %d physical cores -> "
        "linear all-core PROJECTION %.0f ct/s, GPU/CPU = %.1fx projected (an upper bound on the host: it assumes perfect scaling of a "
        "memory-heavy workload; the container may use %d CPUs), %.0fx against the measured %d threads" % (
            cb["cpu_model"], cb["value"], cb["cores"], cb["value_1thread"], cb["physical_cores"], cb["projected_all_physical_cores_linear"],
            cb["gpu_over_cpu_all_physical_cores_projected"], cb["cores"], cb["gpu_over_cpu_%dthreads_measured" % cb["cores"]], cb["cores"]),
        "- kernel time shares in the timed steps: %s" % json.dumps(d["kernel_time_shares"]), ""]
out += ["Other BASELINE lines through the same `bench.py` (`--config 4`, `--config 5`; first / middle / last item verified against the oracle):"]
for c in (4, 5):
    j = last("bench_cfg%d.jsonl" % c)
    out.append("- config %d: **%.0f %s**, %.1f ms per step of %d, dominant `%s` at %.3f of HBM, NTT-only section %.3f, verified %s; CPU oracle %.1f/s on %d "
               "threads (%.2f on one)" % (c, j["value"], j["unit"], j["ms_per_step"], j["config"]["global_batch"], j["roofline"]["kernel"],
                                          j["roofline"]["frac"], j["ntt"]["hbm_roofline_frac"], j["verified_vs_oracle"], j["cpu_baseline"]["value"],
                                          j["cpu_baseline"]["cores"], j["cpu_baseline"]["value_1thread"]))
f = last("bench_force_dist.jsonl")
out += ["", "Multi-rank path on the real backend with one rank (`bench_force_dist.jsonl`: `python -m torch.distributed.run --nproc-per-node 1 bench.py "
        "--gpus 1 --force-dist`): dist_initialized %s, gather backend %s, ranks_seen %d, %.1f GB moved at %.0f GB/s (device-local copy: one rank "
        "gathers to itself), %.0f ct/s, verified %s. No multi-GPU curve exists until the driver's SCALE record." % (
            f["dist_initialized"], f["gather"]["backend"], f["gather"]["ranks_seen"], f["gather"]["bytes_per_rank"] / 1e9,
            f["gather"]["GBps_into_root"], f["value"], f["verified_vs_oracle"]), ""]
out += ["PMC (`traffic.json`; separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over `bench.py --batch 256 --steps 2 --warmup 0 --ntt-polys 0`; "
        "FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md), per kernel and for the whole step:", "```", rd("traffic.json").strip(), "```", ""]
out += ["Per-kernel times of one step (HIP events; `step_profile_b1024.txt`: config 3 multiply+relinearize and square+relinearize over 1024 pairs; "
        "`step_profile_side_configs.txt`: config 4 rotate and multiply+relinearize over 1024 ciphertexts, config 5 over 256, config 1 over 4096):",
        "```", "\n".join(l for l in rd("step_profile_b1024.txt").splitlines() if l.startswith("cfg")), rd("step_profile_side_configs.txt").strip(), "```",
        "Round 2, config 3, same kind of box: fwd 17.8-18.3, inv + tensor 14.0, floor 5.7, lift 3.6, mac 3.7, moddown 2.75 = 47.4-48.5 ms.", ""]
out += ["Side configs (`configs_1gpu.jsonl`):", "```"]
for l in rd("configs_1gpu.jsonl").splitlines():
    if l.startswith('{"config": "cfg'):
        j = json.loads(l)
        out.append(j["config"] + ": " + ", ".join("%s %.4g" % (k, v) for k, v in j.items() if k != "config" and isinstance(v, (int, float))))
out += ["```", "", "FP64 against integer NTT instances on 50-bit primes, same box back to back (`ntt_fp64_ab.txt`; inverse at 2^16 = half + top kernels):",
        "```", rd("ntt_fp64_ab.txt").strip(), "```",
        "Standalone transforms (`ntt_only.txt`: N = 2^15 on config 3's 55-bit primes = integer instances; 2^14 / 2^16 on 50-bit primes = FP64):", "```"]
out += [l[:100] for l in rd("ntt_only.txt").splitlines() if l.startswith("logn")]
out += ["```", "", "Round-3 experiment records in this directory:",
        "- `store_pattern_ubench.txt` (`tools/ubench_store_pattern.hip`): the same half rows written by nothing but stores in seven lane -> address "
        "patterns, plain and nontemporal. The slow case is specific to NONTEMPORAL 16-byte pieces at a 32-byte stride (1.9 TB/s; plain stores of the "
        "same pattern 5.7 TB/s); a nontemporal instruction is fast (5.5 TB/s) as soon as the wave as a whole covers contiguous memory, whichever "
        "lane writes which piece -- which is what one `v_permlane32_swap` per dword delivers.",
        "- `store_swap_ab.txt`: config 3, same box, interleaved: r02 stores / register transposition + nontemporal / plain stores / transposition + plain: "
        "forward transforms 18.05-18.13 / **17.46-17.49** / 18.66-18.71 / 18.53-18.66 ms per 1024 pairs; standalone 36.4 / 36.6 / 34.7-35.6 / 34.8-35.6 %.",
        "- `store_swap_ab_fp64_n65536.txt`: the transposition against the LDS trip for the FP64 instances and N = 2^16: forward 2^15 FP64 0.405-0.412 -> "
        "0.424-0.425, 2^16 FP64 0.315 -> 0.348-0.349, 2^16 integer 0.244-0.245 -> 0.257-0.259; config 5 forward 17.0 -> 14.75 ms per 128 pipelines.",
        "- `tensor_inverse_grouped_ab.txt`: the three outputs of an (item, prime) of the fused-tensor inverse enumerated next to each other on one XCD: "
        "inverse + tensor 13.9 -> 11.8-11.95 ms per 1024 pairs, step 46.4-46.5 -> 43.8-44.0 ms.",
        "- `square_relin_step_profile.txt`: square + relinearize 37.5 ms per 1024 against 44.5 for multiply + relinearize (+18.6 % ciphertexts/s).",
        "- `gpu_tests_bench_children.txt`: the `-m gpu` child-process tests of bench.py (RCCL with one rank under torchrun and with its own group; "
        "configs 4 and 5 self-verified)."]
open(os.path.join(root, "summary.md"), "w").write("\n".join(out) + "\n")
print("wrote", os.path.join(root, "summary.md"))
