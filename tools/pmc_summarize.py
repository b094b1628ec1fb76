#!/usr/bin/env python
"""Sum the rocprofv3 --pmc passes written by tools/pmc_collect.sh per kernel and apply the gfx950 HBM-byte
correction of MI355X_MICROARCH.md (read bytes = 2 x FETCH_SIZE x 1024, write bytes = WRITE_SIZE x 1024).
    python tools/pmc_summarize.py gpurun_out/pmc profiles/r02_xyz_pmc_counters.json [workload_records [profiles/pmc_traffic.json]]"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return m.group(1) if m else name


def main():
    src, dst = sys.argv[1], sys.argv[2]
    records = int(sys.argv[3]) if len(sys.argv) > 3 else None
    kern = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for path in glob.glob(os.path.join(src, "pass*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = short(row["Kernel_Name"])
            kern[k][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
    out = {}
    for k, c in kern.items():
        if not any(s in k for s in ("sw_", "band_", "prep_")):
            continue
        d = {n: v for n, v in sorted(c.items())}
        d["launches"] = max(len(v) for (kk, _), v in launches.items() if kk == k)
        if "FETCH_SIZE" in c:
            d["hbm_bytes"] = int(2 * c["FETCH_SIZE"] * 1024 + c.get("WRITE_SIZE", 0) * 1024)
        out[k] = d
    doc = {"method": "rocprofv3 --pmc <group> --kernel-trace, one counter group per run (tools/pmc_collect.sh), MI355X; "
                     "bench.py --steps 1 --warmup 0 (one vtx_run)",
           "note": "FETCH_SIZE on gfx950 under-reports wide reads by 2x (MI355X_MICROARCH.md, HBM section): "
                   "read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 as is",
           "workload_records": records, "kernels": out}
    json.dump(doc, open(dst, "w"), indent=1)
    # the per-kernel digest bench.py quotes (only for the code it was measured on: stamped with the kernel sources' hash)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import hashlib
    h = hashlib.sha256()
    for rel in ("vartrix_amd/csrc/vtx_band.hip", "vartrix_amd/csrc/vtx_kernels.hip", "vartrix_amd/csrc/vtx_api.hip"):
        h.update(open(os.path.join(root, rel), "rb").read())
    digest = {"source_hash": h.hexdigest()[:16], "from": os.path.basename(dst),
              "method": "rocprofv3 --pmc, one counter group per pass (FETCH_SIZE and WRITE_SIZE in separate passes; read bytes = "
                        "2 x FETCH_SIZE x 1024 on gfx950, MI355X_MICROARCH.md); per launch = total / launches"}
    for k, d in out.items():
        n = max(d.get("launches", 1), 1)
        e = {"workload_records": records, "launches_profiled": n}
        if "hbm_bytes" in d:
            e["hbm_bytes_per_launch"] = d["hbm_bytes"] // n
        if "SQ_INSTS_VALU" in d:
            e["valu_instructions_per_launch"] = d["SQ_INSTS_VALU"] / n
            e["lds_instructions_per_launch"] = d.get("SQ_INSTS_LDS", 0) / n
            e["wave_cycles_quad_per_launch"] = d.get("SQ_WAVE_CYCLES", 0) / n
        if "SQ_LDS_BANK_CONFLICT" in d:
            e["lds_bank_conflict_cycles"] = d["SQ_LDS_BANK_CONFLICT"] / n
            e["lds_active_cycles"] = d.get("SQ_LDS_IDX_ACTIVE", 0) / n
            if d.get("SQ_LDS_IDX_ACTIVE"):
                e["bank_conflict_frac_of_lds_active"] = round(d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"], 3)
        if "SQ_WAIT_ANY" in d:
            e["wait_any_cycles"] = d["SQ_WAIT_ANY"] / n
            e["wait_inst_any_cycles"] = d.get("SQ_WAIT_INST_ANY", 0) / n
        # several template variants of one kernel (band_run_kernel: the first pass and the small second-chance pass): the
        # digest quotes the one that issues the most instructions, with its variant named
        base = re.sub(r"<.*", "", k)
        e["variant"] = k
        if base not in digest or e.get("valu_instructions_per_launch", 0) > digest[base].get("valu_instructions_per_launch", 0):
            digest[base] = e
    if len(sys.argv) > 4:
        json.dump(digest, open(sys.argv[4], "w"), indent=1)
    for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
        print(k, {n: (round(v) if isinstance(v, float) else v) for n, v in d.items()})


if __name__ == "__main__":
    main()
