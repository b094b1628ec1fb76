"""Seeded synthetic workloads of the shapes BASELINE.json names.

Generates *packed batches* directly (what the host packer would hand to
``vtx_submit`` after BAM/VCF/FASTA ingest and read filtering), so kernel parity
and the benchmark do not depend on file ingest.  No compute of the hot path
happens here.

Model: one contig of ``1000 * n_loci`` iid-uniform ACGT bases; locus i at
pos0 = 500 + 1000 i; SNV (REF -> random other base) or, with ``indel_frac`` > 0,
insertions/deletions of 1..max_indel bases in VCF anchor-base form; default
padding 100 (reference ``src/main.rs:88``) so haplotypes are 201 bases (SNV).
``reads_per_locus`` reads of ``read_len`` bases each cover the variant (start
uniform in [pos0 - read_len + 1, pos0]); each read belongs to a cell uniform
over ``n_barcodes``; ``unlisted_frac`` of the reads carry a barcode outside the
list and are dropped (the reference's ``num_not_cell_bc`` path,
``src/main.rs:867-876``).  The allele a read carries follows the cell's
genotype at the locus (ref/ref .45, het .45, alt/alt .10); bases then suffer
iid substitution errors.  With ``use_umi`` the reads of a (locus, cell) are
split into UMI families of size 1 + Poisson(1) and ``umi_flip`` of the reads in
a family carry the other allele (exercises the 0.75 rule, ``src/main.rs:1070-1081``).
"""
from __future__ import annotations

from dataclasses import dataclass

import os

import numpy as np

from .abi import LOCUS_DTYPE, RECORD_DTYPE, PackedBatch

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


@dataclass
class SynthSpec:
    n_loci: int
    n_barcodes: int
    reads_per_locus: int = 256
    read_len: int = 150
    padding: int = 100
    seed: int = 20260926
    sub_error: float = 0.005
    unlisted_frac: float = 0.05
    indel_frac: float = 0.0
    max_indel: int = 20
    use_umi: bool = False
    umi_flip: float = 0.02
    read_len_jitter: int = 0     # reads get length read_len - U[0, jitter]
    depth_sigma: float = 0.0     # > 0: reads per locus ~ log-normal, median reads_per_locus, this sigma (>= 1 read)
    genome_fasta: str = ""       # SNV fast path only: loci at random positions of THIS sequence (its ACGT bases) instead of an
                                 # iid genome — real sequence has 2-3x the chance 6-mer matches of iid bases, and repeats

    @property
    def name(self) -> str:
        kind = "SNV" if self.indel_frac == 0 else "SNV+indel<=%d" % self.max_indel
        depth = "%d" % self.reads_per_locus if self.depth_sigma == 0 else "log-normal(median %d, sigma %g)" % (
            self.reads_per_locus, self.depth_sigma)
        return "synthetic %d %s loci x %d barcodes, %s x %dbp reads/locus%s%s" % (
            self.n_loci, kind, self.n_barcodes, depth, self.read_len, ", UMI" if self.use_umi else "",
            ", loci drawn from " + os.path.basename(self.genome_fasta) if self.genome_fasta else "")



def _depths(spec: "SynthSpec", rng, nl: int) -> np.ndarray:
    """Reads generated per locus (before the unlisted-barcode filter)."""
    if spec.depth_sigma <= 0:
        return np.full(nl, spec.reads_per_locus, np.int64)
    d = np.rint(spec.reads_per_locus * np.exp(spec.depth_sigma * rng.standard_normal(nl)))
    return np.clip(d, 1, 64 * spec.reads_per_locus).astype(np.int64)



def _mix(a: np.ndarray, b: np.ndarray, seed: int) -> np.ndarray:
    """splitmix64-style hash of (a, b, seed) -> uint64 (vectorised)."""
    with np.errstate(over="ignore"):
        z = a.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + b.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)
        z = z + np.uint64(seed) * np.uint64(0x165667B19E3779F9)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return z


def _genotype_allele(loc: np.ndarray, cell: np.ndarray, coin: np.ndarray, seed: int) -> np.ndarray:
    """Allele (0 ref / 1 alt) carried by reads of `cell` at `loc`; het reads use `coin`."""
    u = (_mix(loc, cell, seed) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    het = (u >= 0.45) & (u < 0.90)
    return np.where(u >= 0.90, 1, np.where(het, (coin < 0.5).astype(np.int64), 0)).astype(np.uint8)


def _make_batch_snv(spec: SynthSpec, chunk_loci: int = 4096) -> PackedBatch:
    """Fast path of make_batch for SNV-only, fixed-length reads: same statistical model,
    generated with row gathers from sliding windows of the genome (no per-base index arrays)."""
    rng = np.random.default_rng(spec.seed)
    V, B, R, Lr, pad = spec.n_loci, spec.n_barcodes, spec.reads_per_locus, spec.read_len, spec.padding
    if spec.genome_fasta:
        raw = np.frombuffer(b"".join(l.strip().upper() for l in open(spec.genome_fasta, "rb") if not l.startswith(b">")), np.uint8)
        genome = raw[np.isin(raw, _ACGT)]
        if genome.shape[0] < 4 * (Lr + pad):
            raise ValueError("genome_fasta holds too few ACGT bases")
        genome_codes = np.searchsorted(_ACGT, genome).astype(np.uint8)
        all_pos0 = rng.integers(Lr + pad, genome.shape[0] - Lr - pad, size=V)
    else:
        genome_codes = rng.integers(0, 4, size=1000 * V + 1000, dtype=np.uint8)
        genome = _ACGT[genome_codes]
        all_pos0 = 500 + 1000 * np.arange(V, dtype=np.int64)
    read_rows = np.lib.stride_tricks.sliding_window_view(genome, Lr)
    hap_len = 2 * pad + 1
    hap_rows = np.lib.stride_tricks.sliding_window_view(genome, hap_len)
    n_total_cells = int(round(B / (1.0 - spec.unlisted_frac))) if spec.unlisted_frac > 0 else B
    loci_parts, rec_parts, hap_parts, read_parts = [], [], [], []
    n_rec_total = 0
    for a in range(0, V, chunk_loci):
        b = min(a + chunk_loci, V)
        nl = b - a
        li = np.arange(a, b, dtype=np.int64)
        pos0 = all_pos0[a:b]
        snv_alt = _ACGT[(genome_codes[pos0] + 1 + rng.integers(0, 3, size=nl)) % 4]
        haps = np.empty((nl, 2, hap_len), np.uint8)
        haps[:, 0, :] = hap_rows[pos0 - pad]
        haps[:, 1, :] = haps[:, 0, :]
        haps[:, 1, pad] = snv_alt
        hap_parts.append(haps.reshape(-1))
        depth = _depths(spec, rng, nl)
        n = int(depth.sum())
        lidx = np.repeat(np.arange(nl, dtype=np.int64), depth)
        cell = rng.integers(0, n_total_cells, size=n)
        start_rel = rng.integers(-(Lr - 1), 1, size=n)
        coin = rng.random(n, dtype=np.float32)
        # listed barcodes only, ordered by (locus, cell) — stable, like the reference's sort (:932)
        idx = np.nonzero(cell < B)[0]
        idx = idx[np.argsort(lidx[idx] * (B + 1) + cell[idx], kind="stable")]
        lidx, cell, start_rel, coin = lidx[idx], cell[idx], start_rel[idx], coin[idx]
        nk = idx.shape[0]
        umi = np.zeros(nk, np.int64)
        allele = _genotype_allele(li[lidx], cell, coin, spec.seed)
        if spec.use_umi:
            key = lidx * (B + 1) + cell
            head = np.zeros(nk, bool)
            head[0] = True
            head[1:] = key[1:] != key[:-1]
            starts = np.cumsum(1 + rng.poisson(1.0, size=nk))
            head[starts[starts < nk]] = True
            fam = np.cumsum(head) - 1
            fam_allele = allele[np.nonzero(head)[0][fam]]
            flip = rng.random(nk, dtype=np.float32) < spec.umi_flip
            allele = np.where(flip, 1 - fam_allele, fam_allele).astype(np.uint8)
            umi = fam
        seq = read_rows[pos0[lidx] + start_rel]                       # (nk, Lr) row gather (copy)
        alt_rows = np.nonzero(allele == 1)[0]
        seq[alt_rows, -start_rel[alt_rows]] = snv_alt[lidx[alt_rows]]
        flat = seq.reshape(-1)
        n_err = rng.binomial(flat.shape[0], spec.sub_error) if spec.sub_error > 0 else 0
        if n_err:
            epos = rng.integers(0, flat.shape[0], size=n_err)
            codes = np.searchsorted(_ACGT, flat[epos])
            flat[epos] = _ACGT[(codes + 1 + rng.integers(0, 3, size=n_err)) % 4]
        recs = np.zeros(nk, RECORD_DTYPE)
        recs["read_off"] = (n_rec_total + np.arange(nk, dtype=np.int64)) * Lr
        recs["read_len"] = Lr
        recs["cell_index"] = cell
        recs["umi_id"] = (umi % (1 << 31)).astype(np.uint32)
        read_parts.append(flat)
        counts = np.bincount(lidx, minlength=nl)
        loci = np.zeros(nl, LOCUS_DTYPE)
        loci["row"] = li
        loci["rec_begin"] = n_rec_total + np.concatenate([[0], np.cumsum(counts)[:-1]])
        loci["rec_count"] = counts
        loci["ref_off"] = li * (2 * hap_len)
        loci["ref_len"] = hap_len
        loci["alt_off"] = li * (2 * hap_len) + hap_len
        loci["alt_len"] = hap_len
        n_rec_total += nk
        loci_parts.append(loci)
        rec_parts.append(recs)
    if n_rec_total * Lr >= (1 << 32):
        raise ValueError("synthetic batch exceeds the 4 GiB arena limit of one vtx_batch; split the loci")
    return PackedBatch(np.concatenate(loci_parts), np.concatenate(rec_parts), np.concatenate(hap_parts),
                       np.concatenate(read_parts))


def make_batch(spec: SynthSpec, chunk_loci: int = 2048) -> PackedBatch:
    if spec.indel_frac == 0 and spec.read_len_jitter == 0 and spec.read_len <= 499 - spec.padding:
        return _make_batch_snv(spec)
    if spec.genome_fasta:
        raise ValueError("genome_fasta is implemented for the SNV fast path only")
    rng = np.random.default_rng(spec.seed)
    V, B, R, Lr, pad = spec.n_loci, spec.n_barcodes, spec.reads_per_locus, spec.read_len, spec.padding
    margin = 4 * (Lr + pad + spec.max_indel + 8)
    lead = max(0, margin - 500)                      # keep windows of the first locus inside the contig
    genome_codes = rng.integers(0, 4, size=lead + 1000 * V + 1000 + margin, dtype=np.uint8)
    genome = _ACGT[genome_codes]

    loci_parts, rec_parts, hap_parts, read_parts = [], [], [], []
    n_rec_total = 0
    hap_off = 0
    read_off = 0
    for a in range(0, V, chunk_loci):
        b = min(a + chunk_loci, V)
        nl = b - a
        li = np.arange(a, b, dtype=np.int64)
        pos0 = lead + 500 + 1000 * li
        # --- variants -------------------------------------------------------
        kind = np.zeros(nl, np.int64)                      # 0 SNV, 1 INS, 2 DEL
        if spec.indel_frac > 0:
            u = rng.random(nl)
            kind = np.where(u < spec.indel_frac / 2, 1, np.where(u < spec.indel_frac, 2, 0))
        ilen = rng.integers(1, spec.max_indel + 1, size=nl)
        snv_alt = _ACGT[(genome_codes[pos0] + 1 + rng.integers(0, 3, size=nl)) % 4]
        ins_seq = _ACGT[rng.integers(0, 4, size=(nl, spec.max_indel), dtype=np.uint8)]
        ref_allele_len = np.where(kind == 2, 1 + ilen, 1)          # REF column of the VCF
        alt_allele_len = np.where(kind == 1, 1 + ilen, 1)
        # alt "chromosome" window per locus: genome[pos0-W : pos0] + ALT + genome[pos0+reflen : ...]
        W = Lr + pad + spec.max_indel + 8
        win = pos0[:, None] - W + np.arange(2 * W + 2 * spec.max_indel + 8)[None, :]
        ref_win = genome[win]                                        # (nl, WW) reference bases around the locus
        alt_win = ref_win.copy()
        WW = ref_win.shape[1]
        for k in np.nonzero(kind != 0)[0]:                           # indels: rebuild that row
            left = ref_win[k, :W]
            if kind[k] == 1:
                allele = np.concatenate([ref_win[k, W:W + 1], ins_seq[k, :ilen[k]]])
            else:
                allele = ref_win[k, W:W + 1]
            right = ref_win[k, W + ref_allele_len[k]:]
            row = np.concatenate([left, allele, right])
            if row.shape[0] < WW:
                row = np.concatenate([row, np.full(WW - row.shape[0], ord("A"), np.uint8)])
            alt_win[k] = row[:WW]
        snv = kind == 0
        alt_win[snv, W] = snv_alt[snv]
        # --- haplotypes (construct_haplotypes, src/main.rs:958-994) ----------
        ref_len = pad + ref_allele_len + pad
        alt_len = pad + alt_allele_len + pad
        for k in range(nl):
            hap_parts.append(ref_win[k, W - pad:W - pad + ref_len[k]])
            hap_parts.append(alt_win[k, W - pad:W - pad + alt_len[k]])
        ref_offs = hap_off + np.concatenate([[0], np.cumsum(ref_len + alt_len)[:-1]])
        alt_offs = ref_offs + ref_len
        hap_off += int((ref_len + alt_len).sum())
        # --- reads ------------------------------------------------------------
        depth = _depths(spec, rng, nl)
        n = int(depth.sum())
        lidx = np.repeat(np.arange(nl), depth)
        cell = rng.integers(0, int(round(B / (1.0 - spec.unlisted_frac))) if spec.unlisted_frac > 0 else B, size=n)
        keep = cell < B                                               # unlisted barcodes are filtered by the host
        rl = np.full(n, Lr, np.int64)
        if spec.read_len_jitter:
            rl -= rng.integers(0, spec.read_len_jitter + 1, size=n)
        start_rel = rng.integers(-(Lr - 1), 1, size=n)               # read start relative to pos0 (ref coords)
        coin = rng.random(n)
        allele = _genotype_allele(li[lidx], cell, coin, spec.seed)
        umi = np.zeros(n, np.int64)
        if spec.use_umi:
            # UMI families: consecutive reads of a (locus, cell) group, sizes 1 + Poisson(1)
            order = np.lexsort((cell, lidx))
            key = lidx[order] * (B + 1) + np.minimum(cell[order], B)
            head = np.zeros(n, bool)
            head[0] = True
            head[1:] = key[1:] != key[:-1]
            sizes = 1 + rng.poisson(1.0, size=n)
            starts = np.cumsum(sizes)
            head[starts[starts < n]] = True
            fam_sorted = np.cumsum(head) - 1
            first_pos = np.nonzero(head)[0][fam_sorted]            # first read of each read's family
            fam_allele = allele[order][first_pos]
            flip = rng.random(n) < spec.umi_flip
            new_allele = np.where(flip, 1 - fam_allele, fam_allele).astype(np.uint8)
            umi[order] = fam_sorted
            allele = allele.copy()
            allele[order] = new_allele
        col = (W + start_rel)[:, None] + np.arange(Lr)[None, :]
        seq = np.where(allele[:, None] == 1, alt_win[lidx[:, None], col], ref_win[lidx[:, None], col])
        err = rng.random((n, Lr)) < spec.sub_error
        if err.any():
            codes = np.searchsorted(_ACGT, seq[err])                   # A,C,G,T are sorted
            seq[err] = _ACGT[(codes + 1 + rng.integers(0, 3, size=codes.shape[0])) % 4]
        # --- drop filtered reads, sort by (locus, cell, umi), pack -------------
        idx = np.nonzero(keep)[0]
        order = idx[np.lexsort((umi[idx], cell[idx], lidx[idx]))]
        nk = order.shape[0]
        rlo = rl[order]
        recs = np.zeros(nk, RECORD_DTYPE)
        offs = read_off + np.concatenate([[0], np.cumsum(rlo)[:-1]])
        recs["read_off"] = offs
        recs["read_len"] = rlo
        recs["cell_index"] = cell[order]
        # intern UMI ids per locus to small integers (only equality matters)
        recs["umi_id"] = (umi[order] % (1 << 31)).astype(np.uint32)
        if spec.read_len_jitter:
            mask = np.arange(Lr)[None, :] < rlo[:, None]
            read_parts.append(seq[order][mask])
        else:
            read_parts.append(seq[order].reshape(-1))
        read_off += int(rlo.sum())
        counts = np.bincount(lidx[order], minlength=nl)
        loci = np.zeros(nl, LOCUS_DTYPE)
        loci["row"] = li
        loci["rec_begin"] = n_rec_total + np.concatenate([[0], np.cumsum(counts)[:-1]])
        loci["rec_count"] = counts
        loci["ref_off"] = ref_offs
        loci["ref_len"] = ref_len
        loci["alt_off"] = alt_offs
        loci["alt_len"] = alt_len
        n_rec_total += nk
        loci_parts.append(loci)
        rec_parts.append(recs)
    if read_off >= (1 << 32) or hap_off >= (1 << 32):
        raise ValueError("synthetic batch exceeds the 4 GiB arena limit of one vtx_batch; split the loci")
    return PackedBatch(np.concatenate(loci_parts), np.concatenate(rec_parts),
                       np.concatenate(hap_parts) if hap_parts else np.zeros(0, np.uint8),
                       np.concatenate(read_parts) if read_parts else np.zeros(0, np.uint8))


# The configurations BASELINE.json lists.
def config2() -> SynthSpec:   # 10k SNV loci x 5k barcodes, coverage mode, 1 GPU
    return SynthSpec(n_loci=10_000, n_barcodes=5_000)


def config3() -> SynthSpec:   # 100k SNV loci x 10k barcodes, consensus mode, 1 GPU
    return SynthSpec(n_loci=100_000, n_barcodes=10_000)


def config4() -> SynthSpec:   # 100k loci x 50k barcodes, sharded across 8 GPUs
    return SynthSpec(n_loci=100_000, n_barcodes=50_000)


def config5(n_loci: int = 100_000) -> SynthSpec:   # mixed SNV + indel, alt_frac + UMI
    return SynthSpec(n_loci=n_loci, n_barcodes=10_000, indel_frac=0.30, use_umi=True)


def make_raw(batch: PackedBatch, n_barcodes: int, use_umi: bool, seed: int = 1, frac_unlisted: float = 0.05,
             frac_no_umi: float = 0.02, dup_barcodes: int = 0):
    """Raw form of a packed batch (payload of ``vtx_submit_raw``) -> (RawBatch, barcode list).

    Every record gets the tag bytes of its cell (16-mer + "-1") and UMI (10-mer, injective in umi_id);
    ``frac_unlisted`` extra records carry a barcode that is not in the list, ``frac_no_umi`` extra records
    have no UB tag, records are shuffled inside their locus (BAM order is arbitrary with respect to cells),
    and ``dup_barcodes`` list entries are repeated at the end of the list (first index must win)."""
    from .abi import RAW_RECORD_DTYPE, TAG_MISSING, RawBatch
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)

    def kmer(ids: np.ndarray, k: int, salt: int) -> np.ndarray:   # injective base-4 text of (ids * odd + salt) mod 4^k
        v = (ids.astype(np.uint64) * np.uint64(2654435761) + np.uint64(salt)) % np.uint64(4 ** k)
        digits = (v[:, None] >> (np.uint64(2) * np.arange(k, dtype=np.uint64)[None, :])) & np.uint64(3)
        return acgt[digits.astype(np.int64)]

    bc_text = np.concatenate([kmer(np.arange(n_barcodes), 16, 12345), np.tile(np.frombuffer(b"-1", np.uint8), (n_barcodes, 1))], axis=1)
    barcodes = [bytes(row) for row in bc_text]
    if dup_barcodes:
        barcodes += barcodes[:dup_barcodes]
    unlisted = kmer(np.arange(64), 16, 999)                        # 16 bytes, no "-1": never in the list
    nl = batch.n_loci
    counts = batch.loci["rec_count"].astype(np.int64)
    extra_bc = rng.binomial(counts, frac_unlisted) if frac_unlisted > 0 else np.zeros(nl, np.int64)
    extra_umi = rng.binomial(counts, frac_no_umi) if frac_no_umi > 0 else np.zeros(nl, np.int64)
    new_counts = counts + extra_bc + extra_umi
    n_new = int(new_counts.sum())
    loc_of = np.repeat(np.arange(nl), new_counts)
    begin_new = np.concatenate([[0], np.cumsum(new_counts)[:-1]]).astype(np.int64)
    pos = np.arange(n_new) - begin_new[loc_of]                     # position inside the locus, before the shuffle
    kind = np.where(pos < counts[loc_of], 0, np.where(pos < (counts + extra_bc)[loc_of], 1, 2))
    # source record: the original one, or (extras) a random record of the same locus
    src = batch.loci["rec_begin"].astype(np.int64)[loc_of] + np.where(kind == 0, pos, 0)
    has = counts[loc_of] > 0
    rnd = (rng.random(n_new) * np.maximum(counts[loc_of], 1)).astype(np.int64)
    src = np.where(kind == 0, src, batch.loci["rec_begin"].astype(np.int64)[loc_of] + rnd)
    keep = has | (kind == 0)
    loc_of, kind, src = loc_of[keep], kind[keep], src[keep]
    # shuffle inside each locus
    order = np.lexsort((rng.random(loc_of.shape[0]), loc_of))
    loc_of, kind, src = loc_of[order], kind[order], src[order]
    n = loc_of.shape[0]
    rec = batch.records[src]
    # tag arena: per record 18 barcode bytes + 10 UMI bytes
    tags = np.zeros((n, 28), np.uint8)
    tags[:, :18] = bc_text[rec["cell_index"].astype(np.int64)]
    bad = kind == 1
    tags[bad, :16] = unlisted[rng.integers(0, 64, int(bad.sum()))]
    tags[:, 18:] = kmer(rec["umi_id"].astype(np.int64), 10, 777)
    raw = np.zeros(n, RAW_RECORD_DTYPE)
    raw["read_off"], raw["read_len"] = rec["read_off"], rec["read_len"]
    raw["bc_off"] = np.arange(n, dtype=np.int64) * 28
    raw["bc_len"] = np.where(bad, 16, 18)
    raw["umi_off"] = raw["bc_off"] + 18
    raw["umi_len"] = np.where(kind == 2, TAG_MISSING, 10) if use_umi else np.where(rng.random(n) < 0.5, TAG_MISSING, 10)
    loci = batch.loci.copy()
    cnt = np.bincount(loc_of, minlength=nl)
    loci["rec_count"] = cnt
    loci["rec_begin"] = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    return RawBatch(loci, raw, batch.hap_arena, batch.read_arena, tags.reshape(-1)), barcodes
