"""GPU box: the same bench.py command with several builds of the library (ARAPOPT_LIB), one line per build:
   python tools/variant_bench.py LIB[,LIB...] -- <bench.py arguments>
LIB = a path, or 'default'.  Prints frames/s, ms per step and the per-kernel launch times of the kernel-timing pass."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
i = sys.argv.index("--")
libs, args = sys.argv[1:i][0].split(","), sys.argv[i + 1:]
for lib in libs:
    env = dict(os.environ)
    if lib != "default":
        env["ARAPOPT_LIB"] = lib if os.path.isabs(lib) else os.path.join(ROOT, lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not line:
        print(lib, "FAILED", r.stderr[-400:]); continue
    d = json.loads(line[-1]); rl = d.get("roofline", {})
    pk = rl.get("per_kernel")
    print("%-44s %8.3f frames/s %9.2f ms/step  %s" % (lib, d["value"], d["ms_per_step"],
          {k: round(v["avg_us"], 2) for k, v in pk.items()} if pk else {"resident_us": round(rl.get("avg_launch_us", 0), 1)}), flush=True)
