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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loci", type=int, default=1000)
    ap.add_argument("--reads", type=int, default=256)
    ap.add_argument("--barcodes", type=int, default=2000)
    ap.add_argument("--out", default="/tmp/vtx_e2e")
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
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
