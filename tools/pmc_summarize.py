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
    # kernel durations of the passes (kernel trace of the same runs): the effective clock = GRBM_GUI_ACTIVE / duration
    dur = collections.defaultdict(list)
    for path in glob.glob(os.path.join(src, "pass*", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            dur[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
    for k, c in kern.items():
        if not any(s in k for s in ("sw_", "band_", "prep_", "calib_")):
            continue
        d = {n: v for n, v in sorted(c.items())}
        d["launches"] = max(len(v) for (kk, _), v in launches.items() if kk == k)
        if "FETCH_SIZE" in c:
            # tools/fetch_calib.hip (profiles/r03_fetch_calibration.json): every TCC_EA0_RDREQ is a 128-byte line, for 16-, 8-byte
            # and sector-strided loads alike, and FETCH_SIZE tallies it at 64 bytes -> x 2 for every access shape measured
            d["hbm_bytes"] = int(2 * c["FETCH_SIZE"] * 1024 + c.get("WRITE_SIZE", 0) * 1024)
        if "TCC_EA0_RDREQ_sum" in c:
            d["read_bytes_from_rdreq"] = int(c["TCC_EA0_RDREQ_sum"] * 128)
        if dur.get(k):
            d["kernel_ms_in_profiled_runs"] = sum(dur[k]) / len(dur[k])
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
    for rel in ("vartrix_amd/csrc/vtx_band.hip", "vartrix_amd/csrc/vtx_kernels.hip", "vartrix_amd/csrc/vtx_api.hip",
                "vartrix_amd/csrc/vtx_fast_core.h", "vartrix_amd/csrc/vtx_sweep.hip"):
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
        for cn, key in (("SQ_INSTS_SALU", "salu_instructions_per_launch"), ("SQ_THREAD_CYCLES_VALU", "valu_thread_cycles_per_launch"),
                        ("SQ_ACTIVE_INST_VALU", "active_inst_valu_quad_cycles_per_launch"), ("SQ_ACTIVE_INST_ANY", "active_inst_any_quad_cycles_per_launch"),
                        ("SQ_WAVES", "waves_per_launch"), ("GRBM_GUI_ACTIVE", "grbm_gui_active_per_launch"),
                        ("SQ_BUSY_CYCLES", "sq_busy_cycles_per_launch"), ("TCC_EA0_RDREQ_sum", "tcc_ea0_rdreq_per_launch")):
            if cn in d:
                e[key] = d[cn] / n
        if "kernel_ms_in_profiled_runs" in d:
            e["kernel_ms_in_profiled_runs"] = d["kernel_ms_in_profiled_runs"]
        if "SQ_THREAD_CYCLES_VALU" in d and d.get("SQ_ACTIVE_INST_VALU"):
            e["valu_active_lanes_mean"] = round(d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"], 2)
        # several template variants of one kernel (band_run_kernel: the first pass and the small second-chance pass): the
        # digest quotes the one that issues the most instructions, with its variant named
        base = re.sub(r"<.*", "", k)
        e["variant"] = k
        if base not in digest or e.get("valu_instructions_per_launch", 0) > digest[base].get("valu_instructions_per_launch", 0):
            digest[base] = e
    # VALU issue cycles per instruction of the band kernels: static opcode mix x measured per-opcode cycles (tools/isa_mix.py)
    try:
        import subprocess
        asm = "/tmp/vtx_band_pmc.s"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                               "-o", asm, os.path.join(root, "vartrix_amd/csrc/vtx_band.hip")], stderr=subprocess.DEVNULL)
        for base, sym in (("band_diag_kernel", "band_diag_kernel"), ("band_run_kernel", "band_run_kernelILi64ELb1ELi5ELi15"),
                          ("band_tables_kernel", "band_tables_kernel")):
            if base in digest:
                mix = json.loads(subprocess.check_output([sys.executable, os.path.join(root, "tools", "isa_mix.py"), asm, sym]))
                digest[base]["cycles_per_valu_instruction_static_mix"] = mix["cycles_per_valu_instruction_static_mix"]
                digest[base]["static_mix_share_measured"] = mix["share_of_instructions_with_measured_cycles"]
    except Exception as ex:      # no hipcc here: the digest simply lacks the mix
        print("isa_mix skipped:", ex)
    if len(sys.argv) > 4:
        json.dump(digest, open(sys.argv[4], "w"), indent=1)
    for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
        print(k, {n: (round(v) if isinstance(v, float) else v) for n, v in d.items()})


if __name__ == "__main__":
    main()
