"""TEST INFRASTRUCTURE ONLY — CPU restatement of the per-read pre-processing the reference does between
the alignment-level filters and the aligner, for checking ``vtx_submit_raw`` (include/vtx.h).

Follows reference ``src/main.rs``:
  * ``load_barcodes`` :697-718 — index = first occurrence of the byte string (:704-710)
  * ``get_cell_barcode`` :737-750 and its use :867-876 — barcode not in the list => ``num_not_cell_bc``
  * UB test :879-888 — only after the barcode test, only with ``--umi`` => ``num_non_umi``
  * sort by cell :932, grouping by UMI byte string inside a cell :1047-1057 (order-free HashMap)

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np

from vartrix_amd.abi import LOCUS_DTYPE, RECORD_DTYPE, TAG_MISSING, PackedBatch, RawBatch


def prep_raw(raw: RawBatch, barcodes, use_umi: bool):
    """-> (PackedBatch, {"num_not_cell_bc", "num_non_umi"}).  UMI ids: first occurrence inside the locus."""
    index = {}
    for j, b in enumerate(barcodes):
        index.setdefault(bytes(b), j)
    tags = raw.tag_arena.tobytes()
    loci = raw.loci.copy()
    out = []
    not_bc = non_umi = 0
    for li in range(raw.n_loci):
        b0, cnt = int(loci["rec_begin"][li]), int(loci["rec_count"][li])
        umi_ids = {}
        rows = []
        for r in raw.records[b0:b0 + cnt]:
            bc = tags[int(r["bc_off"]):int(r["bc_off"]) + int(r["bc_len"])]
            cell = index.get(bc)
            if cell is None:
                not_bc += 1
                continue
            if use_umi:
                if int(r["umi_len"]) == TAG_MISSING:
                    non_umi += 1
                    continue
                umi = tags[int(r["umi_off"]):int(r["umi_off"]) + int(r["umi_len"])]
            else:
                umi = b"\x01"
            rows.append((cell, umi_ids.setdefault(umi, len(umi_ids)), int(r["read_off"]), int(r["read_len"])))
        rows.sort(key=lambda t: (t[0], t[1]))      # stable
        loci["rec_begin"][li] = len(out)
        loci["rec_count"][li] = len(rows)
        out.extend(rows)
    recs = np.zeros(len(out), RECORD_DTYPE)
    if out:
        a = np.array(out, dtype=np.int64)
        recs["cell_index"], recs["umi_id"], recs["read_off"], recs["read_len"] = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    return (PackedBatch(loci.astype(LOCUS_DTYPE), recs, raw.hap_arena, raw.read_arena),
            {"num_not_cell_bc": not_bc, "num_non_umi": non_umi})


def canonical_records(records: np.ndarray, rec_begin, rec_count):
    """Order- and label-independent form of a prepared batch: per locus the records sorted by
    (cell, read_off, read_len) with every UMI group named by its smallest (read_off, position)."""
    out = []
    for b0, cnt in zip(rec_begin, rec_count):
        r = records[int(b0):int(b0) + int(cnt)]
        groups = {}
        for k in range(r.shape[0]):
            key = (int(r["cell_index"][k]), int(r["umi_id"][k]))
            groups.setdefault(key, []).append((int(r["read_off"][k]), int(r["read_len"][k])))
        rows = []
        for (cell, _), members in groups.items():
            label = min(members)
            rows.extend((cell, label, m) for m in members)
        out.append(sorted(rows))
    return out
