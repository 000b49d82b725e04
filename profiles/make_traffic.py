#!/usr/bin/env python3
"""traffic_<workload>.json for bench.py's roofline block, from a workload's PMC summary (profiles/summarize_pmc.py) and its
kernel-trace statistics:  make_traffic.py <workload> <pmc_summary.txt> <kernel_stats.csv> [directory of the isa_histogram_*.json files]

HBM-side bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts 64 B per 128-B fabric
read request, so read bytes = 2 x FETCH_SIZE x 1024 (cross-check printed: TCC_MISS_sum x 128 B); WRITE_SIZE x 1024 as is.
valu_issue_frac = SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs): the share of the launch's cycles in
which a SIMD issued a vector instruction (SQ_ACTIVE_INST_* count quad-cycles summed over the chip)."""
import ast
import csv
import json
import os
import re
import sys

# (c4: since round 3 the sweep is the longer phase; `c4_env`, `c5_env` describe the environment kernel of those workloads; the small-pair
# sweeps are k_sweep_duo<slots, lanes per pair, events per pair>: four pairs per wavefront for C3 / C4, two for C2a / C5)
DOMINANT = {"c2a": "k_sweep_duo<12, 32, 480, false, false, true>", "c5": "k_sweep_duo<28, 32, 480, false, false, false>",
            "c4": "k_sweep_duo<8, 16, 240, false, false, false>", "c3": "k_sweep_duo<8, 16, 240, false, false, true>",
            "c2b": "k_dense_fused<12, false>",
            "c4_env": "k_env_group<false, 320", "c5_env": "k_env_group<false, 320"}
# (round 5: the sixth template argument of k_sweep_duo says whether the chunk-start counts come from the environments' prefix-count rows:
#  C2a / C3 yes; C4 -- side B without de-duplication, one pair per environment -- and C5 -- 28 category slots -- no)


def main(workload, summary, stats):
    want = DOMINANT[workload]
    # candidates: the kernels of that family, without the INDIRECT companion; the one with the largest total time in the kernel
    # trace is the dominant one (the first pass of a configuration runs a different instantiation than the steady state)
    total_ns, avg_by_name = {}, {}
    with open(stats, newline="") as fh:
        for row in csv.DictReader(fh):
            kname = row["Name"].split("(")[0].strip()
            if want in kname and "true, true" not in kname.replace(want, ""):
                total_ns[kname] = float(row["TotalDurationNs"])
                avg_by_name[kname] = float(row["AverageNs"])
    name = max(total_ns, key=total_ns.get) if total_ns else None
    counters = {}
    for line in open(summary):
        if "| per-dispatch:" not in line:
            continue
        if name and line.split(" | ")[0].split("(")[0].strip() == name:
            counters.update(ast.literal_eval(line.split("per-dispatch: ")[1]))
    avg_ns = avg_by_name.get(name)
    if not name:
        raise SystemExit(f"no kernel matching {want!r} in {summary}")
    read_b = 2 * counters.get("FETCH_SIZE", 0) * 1024
    write_b = counters.get("WRITE_SIZE", 0) * 1024
    out = {"workload": workload, "kernel": name.strip().replace("void ", "").replace("lchd::", ""), "per_launch": counters,
           "avg_launch_ns_kernel_trace": avg_ns,
           "traffic_bytes_per_launch": read_b + write_b,
           "read_bytes": read_b, "write_bytes": write_b, "tcc_miss_x128_crosscheck": counters.get("TCC_MISS_sum", 0) * 128,
           "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950: 64 B counted per 128-B fabric request), write bytes = WRITE_SIZE x 1024"}
    if counters.get("GRBM_GUI_ACTIVE") and counters.get("SQ_ACTIVE_INST_VALU"):
        cyc = counters["GRBM_GUI_ACTIVE"] / 8.0
        out["valu_issue_frac"] = counters["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc
        out["lds_issue_frac"] = counters.get("SQ_ACTIVE_INST_LDS", 0) * 4.0 / 1024.0 / cyc
        if avg_ns:
            out["clock_ghz_during_counter_pass"] = cyc / avg_ns
        out["binding"] = "valu issue" if out["valu_issue_frac"] > 0.5 else "latency (neither the vector pipes nor HBM are busy half the time)"
    if counters.get("SQ_INSTS_VALU") and counters.get("SQ_WAVES"):
        out["valu_insts_per_wave"] = counters["SQ_INSTS_VALU"] / counters["SQ_WAVES"]
    # class-weighted issue ceiling: the launch's vector instructions x the average cost of the kernel's hot-loop instruction mix
    # (profiles/isa_histogram.py on top of the measured per-class costs of profiles/ubench/issue_rates.hip), against the launch time
    hist_dir = os.path.dirname(os.path.abspath(summary)) if len(sys.argv) < 5 else sys.argv[4]
    kshort = out["kernel"]
    hist = os.path.join(hist_dir, "isa_histogram_" + re.sub(r"[<>, ]+", "_", kshort).strip("_") + ".json")
    if counters.get("SQ_INSTS_VALU") and avg_ns and os.path.exists(hist):
        # (the sweep kernels spend four fifths of their instructions in the per-event loop: its mix; kernels of many phases: the whole kernel's)
        basis = "whole_kernel" if ("k_dense_fused" in kshort or "k_env_" in kshort) else "hot_loop"
        hl = json.load(open(hist))[basis]
        per_simd = counters["SQ_INSTS_VALU"] / 1024.0
        out["valu_avg_ns_per_instr_measured"] = avg_ns / per_simd
        out["valu_avg_ns_per_instr_class_mix"] = hl["avg_ns_per_valu_instr_w4"]
        out["valu_ceiling_frac"] = per_simd * hl["avg_ns_per_valu_instr_w4"] / avg_ns
        out["valu_share_of_2_cycle_class"] = hl["share_of_2_cycle_class"]
        out["valu_ceiling_source"] = os.path.basename(hist) + f" ({basis} instruction mix) x issue_rates.json (measured cost per class, 4 waves/SIMD)"
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
