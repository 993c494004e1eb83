#!/usr/bin/env python3
"""Print the figures DESIGN.md section 5 / README.md / profiles/<round>/README.md quote, from a collected profile directory:

    python3 tools/profile_numbers.py profiles/r3

(after `bash tools/profile_round.sh <tag>` on the GPU box, `tools/pmc_summary.py` and `bash tools/collect_profiles.sh <tag> <round>`)."""
import csv
import json
import sys


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else "profiles/r3"
    for f in ("fp32", "c3_mixed"):
        rows = [r for r in csv.DictReader(open(f"{d}/{f}_kernel_stats.csv")) if "conv_gemm_kernel<128, 128" in r["Name"]]
        s = [(int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in rows]
        j = last_json(f"{d}/{f}_under_rocprof.json")
        print(f"{f}: conv_gemm 128x128 (calls, avg us) {[(c, round(a, 1)) for c, a in s]} weighted {sum(c * a for c, a in s) / sum(c for c, _ in s):.1f} us; "
              f"live {j['roofline']['avg_launch_us']} us, frac {j['roofline']['frac']}, {j['ms_per_step']:.2f} ms/step under rocprof")
    b = last_json(f"{d}/bench_steps20_warmup5.json")
    r = b["roofline"]
    print(f"headline: {b['value'] / 1e6:.2f} M samples/s, {b['ms_per_step']:.2f} ms/step, {b['real_time_factor']:.0f} x; dominant {r['kernel']} avg {r['avg_launch_us']} us, "
          f"{r['achieved']} TFLOP/s, frac {r['frac']}, traffic {r['traffic']}")
    c = b["c3_mixed"]
    print(f"c3: {c['ms_per_step']:.2f} ms/step, {c['us_per_valid_frame']:.3f} us/valid frame (headline {c['us_per_frame_headline']:.3f}), frac {c['roofline']['frac']}, "
          f"avg {c['roofline']['avg_launch_us']} us, {c['valid_samples_per_s'] / 1e6:.1f} M valid samples/s")
    c4 = b["c4_sharded"]
    print(f"c4: {c4['ms_per_pass']:.1f} ms/pass, {c4['samples_per_s'] / 1e6:.1f} M samples/s, balance {c4['balance_max_over_mean']}")
    c5 = b["c5_longform"]
    print("c5 (chunks of 512):", {k: (round(c5[k]["ms"], 2), round(c5[k]["real_time_factor"])) for k in ("bf16", "bf16x3", "fp32")},
          "chunks of 2048:", {k: (round(v["ms"], 2), round(v["real_time_factor"])) for k, v in c5["chunks_of_2048_frames"].items() if isinstance(v, dict)})
    print("b1 latency ms:", {k: round(v, 2) for k, v in b["latency_b1_ms"].items() if isinstance(v, float)})
    print(f"hbm-resident: {b['hbm_resident']['ms_per_step']:.2f} ms/step; bf16x3 mode: {b['split_precision_mode'].get('ms_per_step')}")
    cb = b["cpu_baseline"]
    print(f"cpu baseline: {cb['value'] / 1e3:.0f} k samples/s on {cb['cores']} cores; b1 {cb['b1']['median_s']} s; batch legs {cb['batch']['legs']}")


if __name__ == "__main__":
    main()
