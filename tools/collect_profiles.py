"""Copies the summaries of a tools/profile_bench.sh run from gpurun_out/<tag>/ into profiles/
(tracked) and derives HBM traffic per product from the PMC passes:
    python tools/collect_profiles.py <tag> <round-prefix> <workload>
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reads exactly half the bytes of
16-B-per-lane coalesced reads (MI355X_MICROARCH.md §HBM), so it is doubled; WRITE_SIZE is exact."""
import csv
import glob
import os
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
tag, prefix, workload = sys.argv[1], sys.argv[2], sys.argv[3]
src = REPO / "gpurun_out" / tag
dst = REPO / "profiles"
dst.mkdir(exist_ok=True)
shutil.copy(src / "bench.json", dst / f"{prefix}_bench_{workload}.json")
stats = max(glob.glob(str(src / "stats" / "**" / "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
shutil.copy(stats, dst / f"{prefix}_bench_{workload}_kernel_stats.csv")
bench = json.loads((src / "bench.json").read_text())
launches = bench["roofline"]["launches_per_step"]

per_counter = {}
for name, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = max(glob.glob(str(src / name / "**" / "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    by_kernel = defaultdict(list)
    with open(f) as fi, open(dst / f"{prefix}_{workload}_{name}_counter_collection.csv", "w") as fo:
        for i, line in enumerate(fi):
            if i == 0 or "spmm" in line:
                fo.write(line)
    for r in csv.DictReader(open(f)):
        if "spmm" in r["Kernel_Name"]:
            by_kernel[r["Kernel_Name"].split("(int const")[0].replace("void (anonymous namespace)::", "")].append(float(r["Counter_Value"]))
    # mean per launch of each kernel instantiation; one product = one launch of each
    per_counter[counter] = {k: sum(v) / len(v) for k, v in by_kernel.items()}
fetch_kb = sum(per_counter["FETCH_SIZE"].values())
write_kb = sum(per_counter["WRITE_SIZE"].values())
# the µs-scale helper launches (find_long_rows / the listed-rows kernel with an empty list) ride along
main = [k for k in per_counter["FETCH_SIZE"] if "long_rows" not in k and "staged_rows" not in k]
assert len(main) == launches or launches == 1, (per_counter, launches)
# optional third pass (tools/profile_bench.sh): the DRAM-side share of the reads — requests the L2s sent to HBM
# (TCC_EA0_RDREQ_DRAM: 32-byte and 64-byte requests, the latter counted apart) as opposed to all fabric reads
dram = None
dram_files = glob.glob(str(src / "pmc_dram" / "**" / "*_counter_collection.csv"), recursive=True)
if dram_files:
    by = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(max(dram_files, key=os.path.getmtime))):
        if "spmm" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(int const")[0].replace("void (anonymous namespace)::", "")
            by[r["Counter_Name"]][k].append(float(r["Counter_Value"]))
    dram = {c: {k: sum(v) / len(v) for k, v in d.items()} for c, d in by.items()}
    shutil.copy(max(dram_files, key=os.path.getmtime), dst / f"{prefix}_{workload}_pmc_dram_counter_collection.csv")

sys.path.insert(0, str(REPO))
import bench as _bench  # noqa: E402  (fingerprint of the sources these counters were taken with)

rec_path = dst / "pmc_traffic.json"
rec = json.loads(rec_path.read_text()) if rec_path.exists() else {}
rec[workload] = {
    "round": prefix,
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --output-format csv -- "
               "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline",
    "kernels": per_counter,
    "FETCH_SIZE_KB_raw_per_product": fetch_kb, "WRITE_SIZE_KB_raw_per_product": write_kb,
    "correction": "gfx950: FETCH_SIZE x2 (16-B/lane coalesced reads), WRITE_SIZE exact, unit KB",
    "read_bytes_per_product": fetch_kb * 2 * 1024, "write_bytes_per_product": write_kb * 1024,
    "hbm_bytes_per_launch": (fetch_kb * 2 + write_kb) * 1024 / launches,
    "hbm_bytes_per_product": (fetch_kb * 2 + write_kb) * 1024,
    "algorithmic_bytes_per_product": bench["config"]["algorithmic_bytes_per_step"],
    "note": "fabric-side (L2-miss) bytes: Infinity-Cache hits are included in FETCH_SIZE",
    "source_fingerprint": _bench.source_fingerprint(),
    "dram_counters": dram,
}
if dram and "TCC_EA0_RDREQ_DRAM_sum" in dram and "TCC_EA0_RDREQ_sum" in dram:
    # share of the fabric read REQUESTS that went to DRAM, applied to the (width-corrected) fabric read bytes
    rd = sum(dram["TCC_EA0_RDREQ_sum"].values())
    rd_dram = sum(dram["TCC_EA0_RDREQ_DRAM_sum"].values())
    share = rd_dram / rd if rd else None
    rec[workload]["dram_read_request_share"] = share
    if share is not None:
        wr_share = 1.0
        if "TCC_EA0_WRREQ_DRAM_sum" in dram and "TCC_EA0_WRREQ_sum" in dram and sum(dram["TCC_EA0_WRREQ_sum"].values()):
            wr_share = sum(dram["TCC_EA0_WRREQ_DRAM_sum"].values()) / sum(dram["TCC_EA0_WRREQ_sum"].values())
        rec[workload]["dram_write_request_share"] = wr_share
        if share >= 0.999 and wr_share >= 0.999:
            # what MI355X gives (round 3): every fabric request of the L2s is a "DRAM" request — the counters tell local
            # HBM from remote (GMI) and IO targets, not an Infinity-Cache hit from a miss: the cache sits on the memory
            # side of the fabric, behind the point where the TCC counts.  No DRAM-only figure can be derived from them.
            rec[workload]["dram_bytes_per_product"] = None
            rec[workload]["dram_note"] = ("TCC_EA0_RDREQ_DRAM == TCC_EA0_RDREQ and TCC_EA0_WRREQ_DRAM == TCC_EA0_WRREQ: the L2-side "
                                          "counters classify a request by its TARGET (local HBM / GMI / IO), not by whether the "
                                          "memory-side Infinity Cache served it; HBM-only traffic is not observable from rocprofv3")
        else:
            rec[workload]["dram_bytes_per_product"] = fetch_kb * 2 * 1024 * share + write_kb * 1024 * wr_share
rec_path.write_text(json.dumps(rec, indent=1))
print(json.dumps(rec[workload], indent=1))
