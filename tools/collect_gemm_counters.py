"""Summarises the two rocprofv3 --pmc passes of tools/gemm_counters.sh into profiles/<round>_gemm_sq_counters.json:
means over the launches of each GEMM kernel; the matrix-pipe busy fraction is SQ_VALU_MFMA_BUSY_CYCLES ÷ (1024 SIMDs ×
duration × shader clock), with the clock taken from GRBM_GUI_ACTIVE ÷ 8 XCDs ÷ duration (reads high on dispatches this
short: MI355X_MICROARCH.md 'DVFS give-back') and, beside it, at the 2.4 GHz maximum."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

tag, out = sys.argv[1], sys.argv[2]
root = Path(__file__).resolve().parent.parent / "gpurun_out" / tag
agg = defaultdict(lambda: defaultdict(list))
for p in root.glob("pmc*/**/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        name = r["Kernel_Name"]
        if "gemm_f32" not in name:
            continue
        short = name.split("(anonymous namespace)::")[-1].split("(")[0]
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and r.get("Counter_Name") in ("SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            agg[short]["duration_ns_" + r["Counter_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
res = {}
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    dur = m.get("duration_ns_GRBM_GUI_ACTIVE") or m.get("duration_ns_SQ_BUSY_CYCLES")
    e = {"launches": len(c.get("SQ_WAVE_CYCLES", [])), "duration_us": round(dur / 1e3, 1) if dur else None}
    for n in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
              "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
        if n in m:
            e[n] = round(m[n])
    if dur and "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        clk = m["GRBM_GUI_ACTIVE"] / 8 / dur  # GHz
        e["shader_clock_GHz_from_grbm"] = round(clk, 3)
        e["mfma_busy_fraction"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * dur * clk), 4)
        e["mfma_busy_fraction_at_2p4GHz"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * dur * 2.4), 4)
    if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_INST_ANY" in m:
        e["wait_inst_any_over_wave_cycles"] = round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 3)
    res[k] = e
json.dump({"command": "bash tools/gemm_counters.sh <tag>  (two rocprofv3 --kernel-trace --pmc passes over python3 tools/gemm_probe.py)",
           "note": "means over the launches of each kernel; SQ_VALU_MFMA_BUSY_CYCLES summed over 1024 SIMDs (12.9 GFLOP = 196 608 cycles per SIMD), "
                   "GRBM_GUI_ACTIVE over 8 XCDs", "kernels": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
