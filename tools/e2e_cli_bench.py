#!/usr/bin/env python
"""End-to-end run of the drop-in CLI on authored files (FASTA + VCF + BAM + barcodes) of the synthetic
model, checked byte-for-byte against the packed-batch pipeline.  Usage (GPU box):
    python tools/e2e_cli_bench.py --loci 1000 --reads 256 --out /tmp/e2e
Prints wall times of the CLI phases (its --log-level info lines) and the comparison verdict."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import bamwriter, refpipe  # noqa: E402  (test infrastructure: authoring + comparison only)
from vartrix_amd import hostlib, lib  # noqa: E402
from vartrix_amd.abi import default_config  # noqa: E402


def _compress_blocks(args):
    """BGZF-compress consecutive 60 000-byte slices [lo, hi) of a shared byte file (worker of author_fast)."""
    path, lo, hi, level = args
    import zlib as _z
    out = []
    with open(path, "rb") as fh:
        fh.seek(lo)
        data = fh.read(hi - lo)
    for o in range(0, len(data), 60000):
        blk = data[o:o + 60000]
        comp = _z.compressobj(level, _z.DEFLATED, -15)
        cdata = comp.compress(blk) + comp.flush()
        import struct as _s
        out.append(_s.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(cdata) + 25) + cdata +
                   _s.pack("<II", _z.crc32(blk) & 0xFFFFFFFF, len(blk)))
    return b"".join(out)


def author_fast(out_dir, V, R, B, Lr=150, seed=7, procs=32, chunk_loci=2000):
    """Vectorised authoring of the synthetic model at config-3 scale: every record has the same layout (150M, fixed-size
    name and CB / UB tags), so a chunk of records is one numpy byte matrix; BGZF blocks are compressed by a process pool."""
    import struct
    import multiprocessing as mp
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    codes = rng.integers(0, 4, 1000 * V + 1000, dtype=np.uint8)
    genome = acgt[codes]
    fa = os.path.join(out_dir, "g.fa")
    with open(fa, "wb") as fh:
        fh.write(b">chr1\n")
        body = genome[:len(genome) // 60 * 60].reshape(-1, 60)
        lines = np.concatenate([body, np.full((body.shape[0], 1), 10, np.uint8)], axis=1)
        fh.write(lines.tobytes())
        if len(genome) % 60:
            fh.write(genome[len(genome) // 60 * 60:].tobytes() + b"\n")
    with open(fa + ".fai", "w") as fh:
        fh.write("chr1\t%d\t6\t60\t61\n" % len(genome))
    n_total = int(round(B / 0.95))
    bc_codes = rng.integers(0, 4, (n_total, 16), dtype=np.uint8)
    bc_text = np.concatenate([acgt[bc_codes], np.tile(np.frombuffer(b"-1", np.uint8), (n_total, 1))], axis=1)   # 18 bytes
    listed = [bytes(r) for r in bc_text[:B]]
    assert len(set(listed)) == B
    with open(os.path.join(out_dir, "bcs.tsv"), "wb") as fh:
        fh.write(b"\n".join(listed) + b"\n")
    pos0_all = 500 + 1000 * np.arange(V, dtype=np.int64)
    alt_codes = (codes[pos0_all] + 1 + rng.integers(0, 3, V)) % 4
    with open(os.path.join(out_dir, "v.vcf"), "w") as vf:
        vf.write("##fileformat=VCFv4.2\n##contig=<ID=chr1,length=%d>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n" % len(genome))
        vf.write("".join("chr1\t%d\t.\t%s\t%s\t.\t.\t.\n" % (p + 1, "ACGT"[codes[p]], "ACGT"[a]) for p, a in zip(pos0_all, alt_codes)))
    # fixed record layout
    name_len, n_name = 10, 9                                  # "r" + 8 hex + NUL
    aux_len = 3 + 19 + 3 + 11                                 # CB:Z:<18>\0 UB:Z:<10>\0
    body_len = 32 + name_len + 4 + (Lr + 1) // 2 + Lr + aux_len
    rec_len = 4 + body_len
    nt16 = np.array([1, 2, 4, 8], np.uint8)                  # A C G T in BAM's 4-bit code
    windows = np.lib.stride_tricks.sliding_window_view(codes, Lr)
    raw = os.path.join(out_dir, "r.raw")
    text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:%d\n" % len(genome)
    hdr = b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\x00" + \
        struct.pack("<i", len(genome))
    n_reads = 0
    all_starts = []
    with open(raw, "wb") as fh:
        fh.write(hdr)
        for a in range(0, V, chunk_loci):
            b = min(a + chunk_loci, V)
            nl = b - a
            n = nl * R
            li = np.repeat(np.arange(a, b, dtype=np.int64), R)
            start = pos0_all[li] - rng.integers(0, Lr, n)
            order = np.lexsort((start, li))
            li, start = li[order], start[order]
            seq = windows[start].copy()                                   # (n, Lr) base codes
            carry = rng.random(n) < 0.5
            rows = np.nonzero(carry)[0]
            seq[rows, (pos0_all[li] - start)[rows]] = alt_codes[li[rows]]
            err = rng.random((n, Lr)) < 0.005
            seq[err] = (seq[err] + 1) % 4
            cell = rng.integers(0, n_total, n)
            rec = np.zeros((n, rec_len), np.uint8)
            rec[:, 0:4] = np.frombuffer(struct.pack("<i", body_len), np.uint8)
            rec[:, 4:8] = 0                                               # refID 0
            rec[:, 8:12] = start.astype("<i4").view(np.uint8).reshape(n, 4)
            rec[:, 12] = name_len
            rec[:, 13] = 60                                               # mapq
            rec[:, 14:16] = np.frombuffer(struct.pack("<H", 4680), np.uint8)
            rec[:, 16:18] = np.frombuffer(struct.pack("<H", 1), np.uint8)   # n_cigar_op
            rec[:, 18:20] = 0                                             # flag
            rec[:, 20:24] = np.frombuffer(struct.pack("<i", Lr), np.uint8)
            rec[:, 24:28] = 0xff; rec[:, 28:32] = 0xff                    # next refID / pos = -1
            rec[:, 32:36] = 0
            o = 36
            ids = (n_reads + np.arange(n, dtype=np.int64))
            hexd = np.frombuffer(b"0123456789abcdef", np.uint8)
            rec[:, o] = ord("r")
            for k in range(8):
                rec[:, o + 1 + k] = hexd[(ids >> (4 * (7 - k))) & 15]
            o += name_len                                                 # NUL already 0
            rec[:, o:o + 4] = np.frombuffer(struct.pack("<I", (Lr << 4) | 0), np.uint8)
            o += 4
            c4 = nt16[seq]
            if Lr % 2:
                c4 = np.concatenate([c4, np.zeros((n, 1), np.uint8)], axis=1)
            rec[:, o:o + (Lr + 1) // 2] = (c4[:, 0::2] << 4) | c4[:, 1::2]
            o += (Lr + 1) // 2
            rec[:, o:o + Lr] = 0xff
            o += Lr
            rec[:, o:o + 3] = np.frombuffer(b"CBZ", np.uint8)
            rec[:, o + 3:o + 21] = bc_text[cell]
            o += 22
            rec[:, o:o + 3] = np.frombuffer(b"UBZ", np.uint8)
            rec[:, o + 3:o + 13] = acgt[rng.integers(0, 4, (n, 10), dtype=np.uint8)]
            fh.write(rec.tobytes())
            all_starts.append(start.astype(np.int64))
            n_reads += n
    size = os.path.getsize(raw)
    step = 60000 * 64
    jobs = [(raw, lo, min(lo + step, size), 1) for lo in range(0, size, step)]
    bam = os.path.join(out_dir, "r.bam")
    with mp.Pool(procs) as pool, open(bam, "wb") as fh:
        for blob in pool.imap(_compress_blocks, jobs, chunksize=1):
            fh.write(blob)
        fh.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))      # BGZF EOF block
    os.remove(raw)
    # a real linear index (the packer's index-guided sweep and its streamed ranges jump with it; the bin index stays empty — only
    # the linear part is read): ioffset[w] = virtual offset of the first record that overlaps the 16 kb window w
    coffs = []                                               # compressed offset of every BGZF block (60 000 uncompressed bytes each)
    with open(bam, "rb") as fh:
        data = np.frombuffer(fh.read(), np.uint8)
    o = 0
    while o + 18 <= len(data):
        coffs.append(o)
        o += int(data[o + 16]) + (int(data[o + 17]) << 8) + 1
    coffs = np.array(coffs, np.int64)
    starts = np.concatenate(all_starts) if all_starts else np.zeros(0, np.int64)
    assert np.all(starts[1:] >= starts[:-1])
    u = len(hdr) + np.arange(len(starts), dtype=np.int64) * rec_len      # uncompressed stream offset of record k
    voff = (coffs[u // 60000] << 16) | (u % 60000)
    n_win = (len(genome) >> 14) + 1
    first = np.searchsorted(starts + Lr, np.arange(n_win, dtype=np.int64) << 14, side="right")   # first record with end > window start
    ioff = np.where(first < len(starts), voff[np.minimum(first, max(len(starts) - 1, 0))], 0).astype("<u8")
    with open(bam + ".bai", "wb") as fh:
        fh.write(b"BAI\x01" + struct.pack("<i", 1) + struct.pack("<i", 0) + struct.pack("<i", n_win) + ioff.tobytes())
    return fa, os.path.join(out_dir, "v.vcf"), bam, os.path.join(out_dir, "bcs.tsv"), n_reads


def run_cli_timed(out_dir, fa, vcf, bam, bcs, threads, extra, label, env=None):
    out = os.path.join(out_dir, "out.mtx")
    for f in (out, os.path.join(out_dir, "ref_matrix.mtx")):
        if os.path.exists(f):
            os.remove(f)
    t0 = time.time()
    r = subprocess.run([hostlib.CLI_PATH, "-v", vcf, "-b", bam, "-f", fa, "-c", bcs, "-o", out, "--threads", str(threads),
                        "--log-level", "info"] + extra, cwd=out_dir, capture_output=True, text=True,
                       env=dict(os.environ, **env) if env else None)
    wall = time.time() - t0
    assert r.returncode == 0, r.stdout + r.stderr
    keep = [ln for ln in r.stderr.splitlines() if any(k in ln for k in ("Ingest", "Device", "shard:", "Merge +", "Waited", "Total",
                                                                       "[vtxh]", "alignments evaluated", "device preparation", "device ingest", "Plan of", "packing on the host"))]
    print("%s: CLI wall time %.2f s\n  %s" % (label, wall, "\n  ".join(keep)), flush=True)
    import re
    m = re.search(r"Total since launch: ([\d.]+) s", r.stderr)
    run_cli_timed.totals.append((label, float(m.group(1)) if m else None, wall))
    return out


run_cli_timed.totals = []


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fast", action="store_true", help="vectorised authoring (config-3 scale); timing runs only, with an "
                    "internal consistency check between --prep host and --prep device")
    ap.add_argument("--procs", type=int, default=32)
    ap.add_argument("--more-threads", type=int, nargs="*", default=[], help="--fast: extra CLI runs with these --threads values")
    ap.add_argument("--loci", type=int, default=1000)
    ap.add_argument("--reads", type=int, default=256)
    ap.add_argument("--barcodes", type=int, default=2000)
    ap.add_argument("--out", default="/tmp/vtx_e2e")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--gap-seconds", type=float, default=3.0, help="pause in front of every timed CLI run: a process that starts while the driver "
                    "still clears the ~20 GB of device memory the run before it released waits for that inside its first allocations "
                    "(round 6: submit 0.25 s or 1.2 - 1.7 s, alternating, on back-to-back runs; the timed quantity is one run on an idle device)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    if args.fast:
        t0 = time.time()
        fa, vcf, bam, bcs, n_reads = author_fast(args.out, args.loci, args.reads, args.barcodes, procs=args.procs)
        print("authored %d reads over %d loci in %.1f s (%.1f MB BAM)" % (n_reads, args.loci, time.time() - t0, os.path.getsize(bam) / 1e6), flush=True)
        texts = []
        runs = [(["--ingest", "device"], "--ingest device (BGZF inflate, record split, filters on the GPU)", args.threads),
                (["--ingest", "device"], "--ingest device (second run, page cache warm)", args.threads),
                (["--ingest", "device"], "--ingest device (third run)", args.threads),
                (["--ingest", "device"], "--ingest device (fourth run)", args.threads),
                (["--ingest", "host", "--prep", "device"], "--ingest host --prep device", args.threads),
                (["--ingest", "host", "--prep", "host"], "--ingest host --prep host", args.threads)]
        for th in args.more_threads:
            runs.append((["--prep", "device"], "--prep device, --threads %d" % th, th))
        runs = [r + (None,) for r in runs]
        print("(%.1f s pause in front of every run: --gap-seconds)" % args.gap_seconds, flush=True)
        for extra, label, th, env in runs:
            time.sleep(args.gap_seconds)
            out = run_cli_timed(args.out, fa, vcf, bam, bcs, th, extra, label, env)
            import hashlib
            texts.append(hashlib.sha256(open(out, "rb").read()).hexdigest())
            print("  .mtx %.1f MB, sha256 %s" % (os.path.getsize(out) / 1e6, texts[-1][:16]), flush=True)
        assert len(set(texts)) == 1, "device-ingested / host-packed outputs differ"
        import json
        n_aln = 2 * n_reads
        dev = sorted(t for lab, t, _ in run_cli_timed.totals if lab.startswith("--ingest device") and t)
        host = sorted(t for lab, t, _ in run_cli_timed.totals if lab.startswith("--ingest host") and t)
        summary = {"what": "drop-in CLI from main() to exit on authored files at config-3 scale (tools/e2e_cli_bench.py --fast)",
                   "loci": args.loci, "reads": n_reads, "alignments": n_aln, "bam_bytes": os.path.getsize(bam), "mtx_sha256_16": texts[0][:16],
                   "ingest_device_s": dev, "ingest_host_s": host,
                   "ingest_device_median_s": dev[len(dev) // 2] if dev else None,
                   "alignments_per_s_end_to_end_median": n_aln / dev[len(dev) // 2] if dev else None,
                   "alignments_per_s_end_to_end_host_packer_best": n_aln / host[0] if host else None}
        with open(os.path.join(args.out, "e2e_summary.json"), "w") as fh:
            json.dump(summary, fh, indent=1)
        print("summary: " + json.dumps(summary))
        print("device-ingested, host-packed + device-prepared and host-prepared .mtx are byte-identical")
        return
    rng = np.random.default_rng(7)
    V, R, B, Lr = args.loci, args.reads, args.barcodes, 150
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 1000 * V + 1000)]
    fa = os.path.join(args.out, "g.fa")
    with open(fa, "wb") as fh:
        fh.write(b">chr1\n")
        for o in range(0, len(genome), 60):
            fh.write(genome[o:o + 60].tobytes() + b"\n")
    with open(fa + ".fai", "w") as fh:
        fh.write("chr1\t%d\t6\t60\t61\n" % len(genome))
    bcs = ["".join("ACGT"[c] for c in rng.integers(0, 4, 16)) + "-1" for _ in range(B)]
    bcs = list(dict.fromkeys(bcs))
    with open(os.path.join(args.out, "bcs.tsv"), "w") as fh:
        fh.write("\n".join(bcs) + "\n")
    t0 = time.time()
    recs = []
    with open(os.path.join(args.out, "v.vcf"), "w") as vf:
        vf.write("##fileformat=VCFv4.2\n##contig=<ID=chr1,length=%d>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n" % len(genome))
        for i in range(V):
            pos0 = 500 + 1000 * i
            ref = chr(genome[pos0])
            alt = "ACGT"[("ACGT".index(ref) + 1 + int(rng.integers(0, 3))) % 4]
            vf.write("chr1\t%d\t.\t%s\t%s\t.\t.\t.\n" % (pos0 + 1, ref, alt))
            starts = pos0 - rng.integers(0, Lr, R)
            for k in range(R):
                s = int(starts[k])
                seq = bytearray(genome[s:s + Lr].tobytes())
                if rng.random() < 0.5:
                    seq[pos0 - s] = ord(alt)
                for e in np.nonzero(rng.random(Lr) < 0.005)[0]:
                    seq[e] = ord("ACGT"[(b"ACGT".index(seq[e]) + 1) % 4])
                cb = bcs[int(rng.integers(0, len(bcs)))] if rng.random() < 0.95 else "NNNNNNNNNNNNNNNN-1"
                recs.append((s, bamwriter.record(0, s, "r%d_%d" % (i, k), seq.decode(), "%dM" % Lr, mapq=60,
                                                 tags=[("CB", "Z", cb), ("UB", "Z", "U%05d" % int(rng.integers(0, 50000)))])))
    recs.sort(key=lambda t: t[0])
    bam = os.path.join(args.out, "r.bam")
    bamwriter.write_bam(bam, [("chr1", len(genome))], [r for _, r in recs])
    print("authored %d reads over %d loci in %.1f s (%.1f MB BAM)" % (len(recs), V, time.time() - t0, os.path.getsize(bam) / 1e6))
    out = os.path.join(args.out, "out.mtx")
    texts = {}
    for prep in ("host", "device"):
        for umi in (False, True):
            for f in (out, os.path.join(args.out, "ref_matrix.mtx")):
                if os.path.exists(f):
                    os.remove(f)
            t0 = time.time()
            r = subprocess.run([hostlib.CLI_PATH, "-v", os.path.join(args.out, "v.vcf"), "-b", bam, "-f", fa, "-c",
                                os.path.join(args.out, "bcs.tsv"), "-o", out, "--threads", str(args.threads), "--log-level", "info",
                                "--prep", prep] + (["--umi"] if umi else []), cwd=args.out, capture_output=True, text=True)
            wall = time.time() - t0
            assert r.returncode == 0, r.stdout
            keep = [ln for ln in r.stderr.splitlines() if "Ingest" in ln or "Device" in ln or "shard:" in ln or "Merge +" in ln or "Waited" in ln or "Total" in ln]
            print("--prep %s%s: CLI wall time %.2f s\n  %s" % (prep, " --umi" if umi else "", wall, "\n  ".join(keep)))
            texts[(prep, umi)] = open(out).read()
    assert texts[("host", False)] == texts[("device", False)] and texts[("host", True)] == texts[("device", True)]
    print("host-prepared and device-prepared outputs are byte-identical (with and without --umi)")
    with open(out, "w") as fh:
        fh.write(texts[("host", False)])
    # the same inputs through the library path (C++ packer -> device), rendered with the oracle's MTX text
    batch, metrics, nv, barcodes, _ = hostlib.pack_files(os.path.join(args.out, "v.vcf"), bam, fa, os.path.join(args.out, "bcs.tsv"), threads=args.threads)
    with lib.Context(default_config(aligner="banded", n_barcodes=len(barcodes))) as ctx:
        ctx.submit(batch)
        ctx.run()
        coo = ctx.fetch_coo()
    want = refpipe.mtx_text(nv, len(barcodes), coo["row"], coo["col"], coo["value"])
    assert open(out).read() == want, "CLI .mtx differs from the library path"
    print("CLI .mtx (%d triplets) byte-identical to the library path; %d scored reads, metrics %s" % (len(coo["row"]), batch.n_records, metrics))


if __name__ == "__main__":
    main()
