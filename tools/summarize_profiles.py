"""Markdown table of the records under profiles/ (what DESIGN.md §4 "Other configurations" quotes):
    python tools/summarize_profiles.py [round]      (default r03)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
names = ["davis_854x480", "multseg3_fd2_854x480", "full_854x480", "multseg3_fd5_1920x1080", "full_1920x1080"]
print("| configuration | frames/s | kernel | rocprof avg launch | µs / PCG iteration | frac (§8d, 8 TB/s) | HBM by counters | "
      "VALU issue | wait share | parity_check | CPU oracle (16 thr) |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for n in names:
    pb = os.path.join(ROOT, "profiles", "%s_%s_bench.json" % (rnd, n))
    pc = os.path.join(ROOT, "profiles", "%s_%s_counters.json" % (rnd, n))
    if not (os.path.exists(pb) and os.path.exists(pc)):
        continue
    b, c = json.load(open(pb)), json.load(open(pc))
    rl = b["roofline"]
    k = rl["kernel"]
    ks = c.get("kernel_stats", {}).get(k, {})
    f = lambda v, fmt="%.3f": "—" if v is None else fmt % v
    it = rl.get("us_per_pcg_iteration")
    if it is None and "per_kernel" in rl:
        it = sum(v["avg_us"] for v in rl["per_kernel"].values())
    pcheck = b.get("parity_check", {}).get("bit_equal")
    print("| `%s` | %.3g | `%s` | %s | %s | %.2f | %s | %s | %s | %s | %s |" % (
        n, b["value"], k, f(ks.get("avg_ns", None) and ks["avg_ns"] / 1e3, "%.1f µs"), f(it, "%.2f"), rl["frac"],
        f(rl.get("hbm_frac_by_counters"), "%.3f"), f(rl.get("valu_issue_frac"), "%.2f"), f(rl.get("wait_frac"), "%.2f"),
        {True: "bit equal", False: "DIFFERS", None: "n/a (CPU sample cut)"}[pcheck],
        "%.4f frames/s" % b["cpu_baseline"]["value"] if "cpu_baseline" in b else "—"))
h = os.path.join(ROOT, "profiles", "%s_hosts_para_gen.json" % rnd)
if os.path.exists(h):
    r = json.load(open(h))
    p = r["para_gen"]
    print("\nhost pipeline: %d pairs through para_gen.py --matches: %.1f frames/s wall, %.1f since the worker was ready, "
          "mean batch %.1f; bin/arap_deform list: %.1f; arap_deform.py list: %.1f"
          % (p["frames"], p["frames_per_s_wall"], p["frames_per_s_since_worker_ready"], p["mean_batch"],
             r["cpp_arap_deform_list"]["frames_per_s_wall"], r["python_arap_deform_list"]["frames_per_s_wall"]))
