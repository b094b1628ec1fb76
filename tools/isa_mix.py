"""VALU issue cycles per instruction of one kernel: its opcode histogram (static, from `hipcc -S`) weighted by the
per-opcode issue cycles measured by tools/valu_peak.hip on the GPU box (profiles/r03_valu_peak_microbench.txt).
On gfx950 a wave64 VALU instruction issues over 2 cycles (SIMD-32: 32-bit add / and / xor / mov / shift ...) or 4 (v_max / v_min,
3-operand ops, packed 16-bit, DPP, SDWA, v_cndmask ...), quarter-rate 32-bit multiplies over 8+ — so "instructions x 4" overstates
and "x 2" understates what the pipe is busy.  The histogram is STATIC (every instruction of the kernel counted once): a proxy
for the dynamic mix, stated as such wherever it is quoted.
    python tools/isa_mix.py <file.s> <kernel-name-substring> [microbench.txt]  ->  JSON on stdout
"""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def microbench(path):
    cyc = {}
    for line in open(path):
        m = re.match(r"(.+?)\s+([\d.]+) ms\s+([\d.]+) T lane-ops/s\s+([\d.]+) cycles", line)
        if m:
            name = m.group(1).strip()
            c = float(m.group(4))
            if "(pair" in name:
                c /= 2                       # two instructions per chain step
            if name == "v_cndmask_b32":
                continue                     # (measured through an unset vcc: an artefact; the cmp + cndmask pairs below are the usable lines)
            cyc[name.split()[0]] = c
    return cyc


def kernel_body(asm, name):
    for m in re.finditer(r"\n(_Z\w*%s\w*):" % re.escape(name), asm):
        start = m.end()
        end = asm.find(".Lfunc_end", start)
        return m.group(1), asm[start:end]
    raise SystemExit("kernel %s not found" % name)


def klass(op, cyc):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.endswith("_sdwa"):
        return cyc.get("v_add_u32_sdwa", cyc.get("v_add_u16_sdwa", 4.0)), "measured (sdwa)"
    if op.endswith("_dpp"):
        return cyc.get("v_mov_b32_dpp", 4.0), "measured (dpp)"
    if base in cyc and base != "v_cndmask_b32":
        return cyc[base], "measured"
    if base.startswith("v_cmp") or base.startswith("v_cmpx"):
        return cyc.get("v_cmp_lt_u32", 4.0), "measured (v_cmp + v_cndmask pair / 2)"
    if base == "v_cndmask_b32":
        return cyc.get("v_cmp_lt_u32", 4.0), "measured (v_cmp + v_cndmask pair / 2)"
    # same-shape relatives of measured ops
    rel = {"v_min_u32": "v_max_u32", "v_max_i32": "v_max_i32", "v_subrev_u32": "v_sub_u32", "v_lshrrev_b32": "v_lshrrev_b32",
           "v_ashrrev_i32": "v_lshrrev_b32", "v_add_co_u32": "v_add_u32", "v_addc_co_u32": "v_add_u32", "v_sub_co_u32": "v_sub_u32",
           "v_subb_co_u32": "v_sub_u32", "v_ffbh_u32": "v_ffbl_b32", "v_mov_b64": "v_mov_b32", "v_and_or_b32": "v_and_or_b32",
           "v_min3_i32": "v_max3_i32", "v_max3_u32": "v_max3_i32", "v_min3_u32": "v_max3_i32", "v_med3_i32": "v_max3_i32",
           "v_mad_i32_i24": "v_mad_u32_u24", "v_mul_i32_i24": "v_mul_u32_u24", "v_mul_hi_u32": "v_mul_lo_u32",
           "v_bfrev_b32": "v_not_b32", "v_accvgpr_write_b32": "v_mov_b32", "v_accvgpr_read_b32": "v_mov_b32"}
    if base in rel and rel[base] in cyc:
        return cyc[rel[base]], "relative of " + rel[base]
    if base in ("v_lshlrev_b64", "v_lshrrev_b64", "v_lshl_add_u64", "v_mad_u64_u32", "v_ashrrev_i64"):
        return 8.0 if base == "v_mad_u64_u32" else 4.0, "assumed (64-bit)"
    if base in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32", "v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32"):
        return 4.0, "assumed (cross-lane)"
    return 4.0, "assumed"


def main():
    asm = open(sys.argv[1]).read()
    cyc = microbench(sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r03_valu_peak_microbench.txt"))
    sym, body = kernel_body(asm, sys.argv[2])
    hist = collections.Counter()
    for l in body.split("\n"):
        l = l.strip()
        if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
            continue
        hist[l.split()[0]] += 1
    valu = {k: v for k, v in hist.items() if k.startswith("v_")}
    n = sum(valu.values())
    total = 0.0
    by_src = collections.Counter()
    rows = []
    for op, cnt in sorted(valu.items(), key=lambda kv: -kv[1]):
        c, src = klass(op, cyc)
        total += c * cnt
        by_src[src.split(" (")[0].split(" of")[0]] += cnt
        rows.append({"op": op, "count": cnt, "cycles": round(c, 2), "source": src})
    out = {"kernel": sym, "valu_instructions_static": n, "salu_instructions_static": sum(v for k, v in hist.items() if k.startswith("s_")),
           "cycles_per_valu_instruction_static_mix": round(total / max(n, 1), 3),
           "share_of_instructions_with_measured_cycles": round((by_src["measured"] + by_src["relative"]) / max(n, 1), 3),
           "top_ops": rows[:25]}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
