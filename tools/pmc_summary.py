#!/usr/bin/env python3
"""Summarise rocprofv3 counter passes of `bench.py` into profiles/<round>/pmc_summary.md and profiles/pmc_traffic.json.

    python tools/pmc_summary.py gpurun_out/prof profiles/r1 [--title "..."] [--name other.md]

reads every `*counter_collection.csv` and `*kernel_stats.csv` below the first directory.  Passes expected (separate runs of
`rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`, counters only,
no trace domains -- /opt/skills/guides/MI355X_MICROARCH.md, HBM and rocprofv3 sections):

    pass A: FETCH_SIZE                       (TCC: 3 of 4 slots)
    pass B: WRITE_SIZE                       (TCC: 2 of 4 slots)
    pass C: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE.  rocprofv3 reports both in KiB (a 32-byte copy reads 3.125e-02), converted
to bytes here; gfx950 tallies a wide coalesced 128-B read request as 64 B, hence the factor 2 on FETCH_SIZE.
Kernel classes are the ones bench.py reports (`conv_x3_128x128`, `conv_gemm_256x32`, `attention`, ...).
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def kernel_class(name: str):
    m = re.search(r"conv_gemm_kernel<(\d+), (\d+), \d+, \d+, (\d+)", name)
    if m:
        bm, bn, mode = m.group(1), m.group(2), int(m.group(3))
        return ("conv_gemm_" if mode == 0 else "conv_x3_") + f"{bm}x{bn}"  # bench.py's classes
    m = re.search(r"resblock_pair_kernel<\d+, (\d+), \d+, \d+, (\d+)", name)
    if m:
        return ("resblock_pair_f32_" if m.group(2) == "0" else "resblock_pair_") + m.group(1)
    m = re.search(r"resblock_chain_kernel<\d+, (\d+)", name)
    if m:
        return "resblock_chain_" + m.group(1)
    # conv_bf16.hip (plain bf16): <MT, NT, WGM, WGN, ...> -> rows x columns of the workgroup tile; the fused forms by their channel count
    # (32 x WGN).  In the 48 kHz stream's windows the pairs and whole ResBlocks run as grouped launches: the engine's class names.
    m = re.search(r"conv_bf16_kernel<(\d+), (\d+), (\d+), (\d+)", name)
    if m:
        mt, nt, wgm, wgn = (int(x) for x in m.groups())
        return f"conv_bf16_{32 * mt * wgm}x{32 * nt * wgn}"
    m = re.search(r"pair_bf16_kernel<\d+, \d+, (\d+)", name)
    if m:
        return f"pair_bf16_group_{32 * int(m.group(1))}"
    m = re.search(r"rb_bf16_kernel<\d+, \d+, (\d+), \d+, (?:true|false|0|1), (true|false|0|1)", name)
    if m:   # the last template argument: the stage form (every ResBlock of the stage per workgroup)
        return ("rb_bf16_stage_" if m.group(2) in ("true", "1") else "rb_bf16_group_") + str(32 * int(m.group(1)))
    m = re.search(r"conv_rows_kernel<\d+, (\d+), \d+, (true|false|1|0)", name)
    if m:  # SPLITK: the phoneme-level K-split form; otherwise the few-rows form of conv_gemm's arithmetic
        if m.group(2) in ("true", "1"):
            return "conv_ksplit"
        return "conv_rows" if m.group(1) == "0" else "conv_x3_rows"
    for key, cls in (("rel_attention_kernel", "rel_attention"), ("attention_x3_split_kernel", "attention_x3"), ("attention_x3_kernel", "attention_x3"),
                     ("attention_split_kernel", "attention"), ("attention_kernel", "attention"),
                     ("layernorm_kernel", "layernorm"), ("conv_post_kernel", "conv_post"), ("dwconv_swish_kernel", "dwconv_swish"),
                     ("glu_kernel", "glu")):
        if key in name:
            return cls
    return None


def main():
    src, dst = sys.argv[1], sys.argv[2]
    title = sys.argv[sys.argv.index("--title") + 1] if "--title" in sys.argv else "PMC summary"
    # --args "...": the bench.py arguments of the passes when they are not the headline's (one workload, one arithmetic mode)
    extra = sys.argv[sys.argv.index("--args") + 1] if "--args" in sys.argv else None
    # --name FILE.md: write the table under that name and leave profiles/pmc_traffic.json (bench.py's `roofline.traffic` source, which
    # belongs to the headline workload) alone -- for side workloads such as `bench.py --blocks conformer`
    name = sys.argv[sys.argv.index("--name") + 1] if "--name" in sys.argv else None
    # --traffic FILE.json: with --name, ALSO write the per-class clock / MFMA busy / HBM bytes as JSON under that name next to
    # pmc_traffic.json (profiles/pmc_traffic_c5.json is what bench.py's c5_longform leg reads for `roofline.traffic`)
    traffic_name = sys.argv[sys.argv.index("--traffic") + 1] if "--traffic" in sys.argv else None
    # --cmd "...": the profiled command line, for the table's header (default: bench.py with --args)
    cmd = sys.argv[sys.argv.index("--cmd") + 1] if "--cmd" in sys.argv else None
    counters = defaultdict(lambda: defaultdict(float))   # class -> counter -> sum over dispatches
    ndisp = defaultdict(lambda: defaultdict(int))        # class -> counter -> dispatches seen
    dur_ns = defaultdict(float)
    dur_n = defaultdict(int)
    for root, _, files in os.walk(src):
        for f in files:
            path = os.path.join(root, f)
            if f.endswith("counter_collection.csv"):
                with open(path, newline="") as fh:
                    for row in csv.DictReader(fh):
                        cls = kernel_class(row.get("Kernel_Name", ""))
                        if cls is None:
                            continue
                        c = row["Counter_Name"]
                        counters[cls][c] += float(row["Counter_Value"]) * (1024.0 if c in ("FETCH_SIZE", "WRITE_SIZE") else 1.0)
                        ndisp[cls][c] += 1
                        if c == "GRBM_GUI_ACTIVE" and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                            dur_ns[cls] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                            dur_n[cls] += 1
    if not counters:
        sys.exit(f"no counter_collection.csv with known kernels under {src}")
    os.makedirs(dst, exist_ok=True)
    traffic = {}
    what = (f"Separate `rocprofv3 --pmc` passes of `{cmd}` -- counters only, no" if cmd else
            f"Separate `rocprofv3 --pmc` passes of `python3 bench.py {extra} --steps 2 --warmup 1 --no-cpu-baseline --no-extras` -- counters only, no"
            if extra else
            "Separate `rocprofv3 --pmc` passes of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras` -- once in the default\n"
            "exact-fp32 arithmetic (classes conv_gemm_*) and once with `--precision bf16x3` (classes conv_x3_*, resblock_*) -- counters only, no")
    lines = [f"# {title}", "", what,
             "trace domains, summarised by tools/pmc_summary.py.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs):",
             "GRBM_GUI_ACTIVE is summed over the 8 XCDs, a bf16 MFMA counts its own 32 cycles, so 1.00 = the dense peak AT THE CLOCK HELD.",
             "clock = GRBM_GUI_ACTIVE / 8 / kernel time.  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 tallies 128-B reads as 64 B).", "",
             "| kernel | launches | clock GHz | MFMA busy | WAIT_ANY/WAVE | WAIT_INST_ANY/WAVE | ACTIVE/WAVE | FETCH_SIZE B/launch (raw) | WRITE_SIZE B/launch | HBM B/launch |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    for cls in sorted(counters, key=lambda k: -counters[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
        c, n = counters[cls], ndisp[cls]
        per = lambda name: (c[name] / n[name]) if n.get(name) else None  # noqa: E731
        gui = per("GRBM_GUI_ACTIVE")
        busy = per("SQ_VALU_MFMA_BUSY_CYCLES")
        clock = (c["GRBM_GUI_ACTIVE"] / 8 / dur_ns[cls]) if dur_ns.get(cls) else None
        mfma_busy = busy / (gui / 8 * 1024) if gui and busy is not None else None
        wave = per("SQ_WAVE_CYCLES")
        frac = lambda name: (per(name) / wave) if wave and per(name) is not None else None  # noqa: E731
        fetch, write = per("FETCH_SIZE"), per("WRITE_SIZE")
        hbm = (2 * fetch + write) if fetch is not None and write is not None else None
        fmt = lambda v, p="{:.2f}": "n/a" if v is None else p.format(v)  # noqa: E731
        launches = max(n.values())
        lines.append(f"| {cls} | {launches} | {fmt(clock)} | {fmt(mfma_busy)} | {fmt(frac('SQ_WAIT_ANY'))} | {fmt(frac('SQ_WAIT_INST_ANY'))} | "
                     f"{fmt(frac('SQ_ACTIVE_INST_ANY'))} | {fmt(fetch, '{:.4g}')} | {fmt(write, '{:.4g}')} | {fmt(hbm, '{:.4g}')} |")
        entry = {}
        if clock is not None:
            entry["clock_ghz"] = clock
        if mfma_busy is not None:
            entry["mfma_busy"] = mfma_busy
        if hbm is not None:
            entry["hbm_bytes_per_launch"] = hbm
        traffic[cls] = entry
    with open(os.path.join(dst, name or "pmc_summary.md"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    if name is None or traffic_name:
        with open(os.path.join(os.path.dirname(os.path.abspath(dst)), traffic_name or "pmc_traffic.json"), "w") as fh:
            json.dump(traffic, fh, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
