"""GPU box: measured ceilings for the utilisation figures of bench.py's roofline.

    python tools/calibrate_ceilings.py [--round r03]

Builds tools/microbench/issue_peak.hip, runs it plain (cycles per instruction from in-kernel stamps) and under
rocprofv3 --pmc with the SAME counter groups tools/collect_profile.py uses, and writes
gpurun_out/profiles/<round>_ceilings.json:
  valu_active_ceiling   the largest SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles) any VALU kernel reaches
  cycles per v_fma_f32 / v_pk_fma_f32 / phase-A mix, at one and two wavefronts per SIMD
  lds bytes per clock per CU for ds_read_b64 / ds_read_b32, and the counters' reading of a saturated LDS pipe
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = {"sq_valu": ["SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"],
          "sq_wait": ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_THREAD_CYCLES_VALU"],
          "sq_lds": ["SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"],
          "sq_lds2": ["SQ_WAIT_INST_LDS", "SQ_LDS_ADDR_CONFLICT", "SQ_INST_LEVEL_LDS", "SQ_BUSY_CU_CYCLES"],
          "grbm": ["GRBM_GUI_ACTIVE"]}
SHORT = {"0": "fma", "1": "pkfma", "2": "mix", "3": "lds64", "4": "lds32", "5": "mixlds"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r03")
    a = ap.parse_args()
    os.chdir("/tmp")
    os.environ["TMPDIR"] = "/tmp"
    exe = "/tmp/issue_peak"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe,
                           os.path.join(ROOT, "tools", "microbench", "issue_peak.hip")])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    plain = [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]
    rec = {"made_by": "tools/calibrate_ceilings.py", "unix_time": int(time.time()), "plain": plain, "pmc": {}}
    for tag, ctrs in GROUPS.items():
        d = "/tmp/cal_%s" % tag
        shutil.rmtree(d, ignore_errors=True)
        r = subprocess.run(["rocprofv3", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "--", exe],
                           capture_output=True, text=True)
        files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if r.returncode != 0 or not files:
            rec["pmc"][tag] = {"error": (r.stderr or r.stdout)[-300:]}
            continue
        # dispatches come in the order the program launches them: per kernel (warm-up, timed) x (2 per CU, 1 per CU)
        per = collections.OrderedDict()
        for row in csv.DictReader(open(files[0])):
            key = (row["Kernel_Name"], int(row["Dispatch_Id"]))
            cs = per.setdefault(key, {})
            cs[row["Counter_Name"]] = cs.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        by_kernel = collections.OrderedDict()
        for (name, disp), cs in sorted(per.items(), key=lambda kv: kv[0][1]):
            by_kernel.setdefault(name, []).append(cs)
        for name, lst in by_kernel.items():
            idx = name.split("<")[-1].split(">")[0].replace("(int)", "").strip()
            short = SHORT.get(idx, name)
            timed = {"2_per_cu": lst[1] if len(lst) > 1 else None, "1_per_cu": lst[3] if len(lst) > 3 else None}
            rec["pmc"].setdefault(short, {})
            for k, cs in timed.items():
                if cs:
                    rec["pmc"][short].setdefault(k, {}).update(cs)
        shutil.rmtree(d, ignore_errors=True)
    # ---- derived ceilings -------------------------------------------------------------------------------------------
    der = {}
    ms = {(p["kernel"], p["waves_per_simd"]): p for p in plain}
    for short, occ in rec["pmc"].items():
        if short in GROUPS or not isinstance(occ, dict):
            continue
        for k, cs in occ.items():
            w = 2 if k.startswith("2") else 1
            p = ms.get((short, w))
            if not p or not isinstance(cs, dict):
                continue
            cycles = p["ms"] * 1e-3 * p["clock_MHz"] * 1e6          # kernel duration in shader clocks (measured clock)
            e = {"kernel_cycles": cycles}
            if "SQ_ACTIVE_INST_VALU" in cs:
                e["valu_active_x4_per_simd_cycle"] = cs["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles)
                e["valu_insts_per_simd_cycle"] = cs.get("SQ_INSTS_VALU", 0.0) / (1024.0 * cycles)
            if "SQ_ACTIVE_INST_LDS" in cs:
                e["lds_active_x4_per_cu_cycle"] = cs["SQ_ACTIVE_INST_LDS"] * 4.0 / (256.0 * cycles)
                e["lds_idx_active_per_cu_cycle"] = cs.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256.0 * cycles)
                e["lds_insts_per_cu_cycle"] = cs.get("SQ_INSTS_LDS", 0.0) / (256.0 * cycles)
            der.setdefault(short, {})[k] = e
    rec["derived"] = der
    vals = [e["valu_active_x4_per_simd_cycle"] for occ in der.values() for e in occ.values()
            if "valu_active_x4_per_simd_cycle" in e]
    rec["valu_active_ceiling"] = max(vals) if vals else None
    lvals = [e["lds_idx_active_per_cu_cycle"] for k in ("lds64", "lds32") for e in der.get(k, {}).values()
             if "lds_idx_active_per_cu_cycle" in e]
    rec["lds_idx_active_ceiling"] = max(lvals) if lvals else None
    prof = os.path.join(ROOT, "gpurun_out", "profiles")
    os.makedirs(prof, exist_ok=True)
    path = os.path.join(prof, "%s_ceilings.json" % a.round)
    open(path, "w").write(json.dumps(rec, indent=1) + "\n")
    print("wrote", path)
    for p in plain:
        print(p)
    print(json.dumps(der, indent=1)[:6000])
    print("valu_active_ceiling", rec["valu_active_ceiling"], "lds_idx_active_ceiling", rec["lds_idx_active_ceiling"])


if __name__ == "__main__":
    main()
