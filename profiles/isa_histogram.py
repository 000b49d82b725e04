#!/usr/bin/env python3
"""Instruction-class histogram of a kernel's hot loop + its class-weighted VALU issue ceiling.

    python3 profiles/isa_histogram.py <file.hip> '<kernel regex on the demangled name>' profiles/r04/issue_rates.json [--loop N] > out.json

The kernel is compiled to gfx950 assembly with the Makefile's flags (`hipcc -S --cuda-device-only`: the very instruction
stream of the code object in libloco_hd_hip.so -- the same compiler, flags and source; no GPU needed).  Loops are the
backward branches of the function; the HOT loop is the smallest loop that evaluates sqrt(H^2) (the per-event loop of the sweep kernels;
kernels without one: the innermost loop with the most instructions), override with --loop.  Every vector instruction is put into one of the classes that
profiles/ubench/issue_rates.hip measured; the loop's average cost per vector instruction (ns per instruction per SIMD at 4
resident waves) is what bench.py multiplies the measured instruction count of a launch with (roofline.valu_ceiling_frac)."""
import collections
import json
import re
import subprocess
import sys

FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "--cuda-device-only", "-S"]

# mnemonic -> class of profiles/ubench/issue_rates.hip (first match wins); anything unmatched is reported as "other_valu"
RULES = [
    (r"^v_(rsq|rcp|sqrt)_f64", "f64_rsq"),
    (r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_f(16|32)", "f32_transcendental"),
    (r"^v_(add|mul|fma|fmac)_f64", "f64_fma"),
    (r"^v_(max|min)_f64", "f64_minmax"),
    (r"^v_(ldexp|rndne|trunc|floor|ceil|fract|frexp\w*)_f64", "f64_ldexp_rndne"),
    (r"^v_cmp\w*_f64|^v_cmp_class_f64", "cmp_f64"),
    (r"^v_cmp\w*_[ui]64", "cmp_u64"),
    (r"^v_cmp", "cmp_u32"),
    (r"^v_cvt_f64_", "cvt_to_f64"),
    (r"^v_cvt_\w+_f64", "cvt_from_f64"),
    (r"^v_cndmask_b32_e64", "cndmask_e64_sgpr"),
    (r"^v_cndmask_b32", "cndmask_e32_in_mix"),
    (r"^v_(lshlrev|lshrrev|ashrrev)_[bi]64", "b64_shift"),
    (r"^v_lshl_add_u64|^v_(add|sub)_(co_)?u64", "lshl_add_u64"),
    (r"^v_mov_b64", "mov_b64"),
    (r"^v_(readlane|readfirstlane|writelane)", "readlane"),
    (r"^v_mad_u64_u32|^v_mad_i64_i32", "mad_u64_u32"),
    (r"^v_mul_(lo|hi)_[ui]32", "mul_u32"),
    (r"_dpp$|_dpp\b", "dpp_add_u32"),
    (r"_sdwa$|_sdwa\b", "b32_add_sdwa"),
    (r"^v_(addc|subb|subbrev)_co_u32|^v_(add|sub|subrev)_co_u32", "add_co_addc_pair_half"),
    (r"^v_(lshl_add|add_lshl|lshl_or|and_or|or3|add3|bfe|bfi|perm|alignbit|xad|mad)_", "b32_three_operand"),
    (r"^v_(med3|max3|min3)_", "b32_three_operand"),
    (r"^v_(add|sub|subrev|mul|fma|fmac|max|min|mac|madak|madmk)_f32", "f32_alu"),
    (r"^v_(add|sub|subrev|max|min)_[ui]32", "b32_add"),
    (r"^v_(and|or|xor|not|lshlrev|lshrrev|ashrrev|bfrev|ffbh|ffbl|bcnt|mbcnt)\w*_[bui]32", "b32_logic_shift"),
    (r"^v_mov_b32", "mov_b32"),
    (r"^v_cvt", "cvt_from_f64"),
]
# classes without a row of their own in issue_rates.json -> the measured class whose cost they take
ALIAS = {"cndmask_e32_in_mix": None, "f32_transcendental": "f64_rsq", "other_valu": "b32_three_operand"}


def classify(mn: str) -> str:
    if mn.startswith("v_") and mn.endswith(("_dpp", "_sdwa")):
        return "dpp_add_u32" if mn.endswith("_dpp") else "b32_add_sdwa"
    for rx, cls in RULES:
        if re.search(rx, mn):
            return cls
    return "other_valu"


def cost_ns(cls: str, rates: dict, w: str = "w4") -> float:
    c = rates["classes"]
    if cls == "cndmask_e32_in_mix":
        # a select behind a compare: (cmp + 2 cndmask + add) / 4 = mix_cmp_2cndmask_e32_add  ->  solve for the cndmask
        mix = c["mix_cmp_2cndmask_e32_add"][w]["ns_per_instr_per_simd"]
        return (4 * mix - c["cmp_u32"][w]["ns_per_instr_per_simd"] - c["b32_add"][w]["ns_per_instr_per_simd"]) / 2
    return c[ALIAS.get(cls) or cls][w]["ns_per_instr_per_simd"]


def main():
    src, kernel_rx, rates_path = sys.argv[1:4]
    loop_pick = int(sys.argv[sys.argv.index("--loop") + 1]) if "--loop" in sys.argv else None
    rates = json.load(open(rates_path))
    import hashlib, os
    cache = f"/tmp/isa_{hashlib.sha1(open(src, 'rb').read()).hexdigest()[:16]}.s"  # (the kernels file takes 90 s to compile)
    if not os.path.exists(cache):
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, src, "-o", cache], check=True)
    asm = open(cache).read()
    names = sorted(set(re.findall(r"^(_Z\w+):", asm, flags=re.M)))
    dem = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")))
    demangle = lambda s: dem.get(s, s)
    # split into functions
    funcs, cur, name = {}, None, None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            funcs[name] = cur
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
        if cur is not None:
            cur.append(line)
    pick = [n for n in funcs if re.search(kernel_rx, demangle(n))]
    if len(pick) != 1:
        raise SystemExit(f"{len(pick)} kernels match {kernel_rx!r}: {[demangle(n) for n in pick][:8]}")
    body = funcs[pick[0]]
    instrs, labels = [], {}
    for line in body:
        t = line.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        instrs.append(t.split(";")[0].strip())
    loops = []
    for i, ins in enumerate(instrs):
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", ins)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            loops.append((labels[m.group(1)], i))
    loops = sorted(set(loops))
    # the per-event loop of the sweep kernels: the SMALLEST loop that holds the square root of H^2 (v_rsq_f64); kernels without one:
    # the innermost loop with the most instructions; --loop N picks the N-th largest loop instead
    with_rsq = [(a, b) for (a, b) in loops if any(x.startswith("v_rsq_f64") for x in instrs[a:b + 1])]
    innermost = [(a, b) for (a, b) in loops if not any((c, d) != (a, b) and a <= c and d <= b for (c, d) in loops)]
    innermost.sort(key=lambda ab: ab[0] - ab[1])
    if loop_pick is not None:
        lo, hi = sorted(loops, key=lambda ab: ab[0] - ab[1])[loop_pick]
    elif with_rsq:
        lo, hi = min(with_rsq, key=lambda ab: ab[1] - ab[0])
    else:
        lo, hi = innermost[0]

    def histogram(seq):
        h = collections.Counter()
        other = collections.Counter()
        for ins in seq:
            mn = ins.split()[0]
            if mn.startswith("v_"):
                cls = classify(mn)
                h[cls] += 1
                if cls == "other_valu":
                    other[mn] += 1
            elif mn.startswith("ds_"):
                h["lds:" + mn] += 1
            elif mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
                h["vmem:" + mn.split("_")[0]] += 1
            elif mn.startswith("s_"):
                h["scalar"] += 1
        return h, other

    def summarize(seq):
        h, other = histogram(seq)
        valu = {k: v for k, v in h.items() if ":" not in k and k != "scalar"}
        n_valu = sum(valu.values())
        out = {"instructions": len(seq), "valu": n_valu, "scalar": h.get("scalar", 0), "lds": sum(v for k, v in h.items() if k.startswith("lds:")),
               "vmem": sum(v for k, v in h.items() if k.startswith("vmem:")), "valu_by_class": dict(sorted(valu.items(), key=lambda kv: -kv[1]))}
        for w in ("w4", "w8"):
            tot = sum(v * cost_ns(k, rates, w) for k, v in valu.items())
            out[f"avg_ns_per_valu_instr_{w}"] = tot / max(n_valu, 1)
        fast = sum(v for k, v in valu.items() if cost_ns(k, rates) < 1.45)
        out["share_of_2_cycle_class"] = fast / max(n_valu, 1)
        if other:
            out["unclassified"] = dict(other)
        return out

    res = {"kernel": demangle(pick[0]).split("(")[0], "source": f"hipcc {' '.join(FLAGS)} {src}",
           "issue_rates": rates_path, "loops_found": len(loops),
           "hot_loop": {"first_instruction": lo, "last_instruction": hi, **summarize(instrs[lo:hi + 1])},
           "whole_kernel": summarize(instrs),
           "note": "avg_ns_per_valu_instr = sum over classes of count x measured wall-clock cost per instruction per SIMD (profiles/ubench/"
                   "issue_rates.hip, 4 / 8 resident waves per SIMD); classes measured at 1.0-1.35 ns (32-bit VOP1/VOP2 arithmetic, moves, f32) "
                   "issue in about half the time of the 1.75-2.0 ns classes (anything 64-bit or f64, compares, three-operand VOP3, DPP, SDWA, "
                   "conversions); f64 rsq / sqrt / rcp cost 6.8 ns"}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
