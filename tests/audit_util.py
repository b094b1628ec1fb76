"""Helpers of the audit tests: the oracle over a SUBSET of a batch's records, and the stage invariant."""
import os

import numpy as np

from oracle import oracle
from vartrix_amd import abi
from vartrix_amd.abi import PackedBatch, default_config


def sub_batch(batch, record_ids):
    """The same loci and arenas, only the listed records (ascending ids)."""
    ids = np.unique(np.asarray(record_ids, np.int64))
    rec_locus = np.repeat(np.arange(batch.n_loci), batch.loci["rec_count"])
    loci = batch.loci.copy()
    cnt = np.bincount(rec_locus[ids], minlength=batch.n_loci).astype(np.uint32)
    loci["rec_count"] = cnt
    loci["rec_begin"] = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.uint32)
    return PackedBatch(loci, np.ascontiguousarray(batch.records[ids]), batch.hap_arena, batch.read_arena), ids


def oracle_scores_of(batch, record_ids, aligner, n_barcodes):
    sub, ids = sub_batch(batch, record_ids)
    if len(ids) == 0:
        return ids, np.zeros(0, np.int32), np.zeros(0, np.int32)
    r, a = oracle.batch_scores(sub, default_config(aligner=aligner, n_barcodes=n_barcodes), threads=os.cpu_count() or 8)
    return ids, r, a


def stage_report(stage):
    return {abi.STAGE_NAMES.get(int(k), str(k)): int(v) for k, v in zip(*np.unique(stage, return_counts=True))}


def assert_stage_invariant(stage, banded, full, label=""):
    """banded = (ref, alt) of the banded flavour, full = of the full flavour, stage = vtx_fetch_stage of the banded run.
    cert <= banded <= full: an alignment whose two scores differ must have been decided by a DP stage or by the
    band-restricted certificate (abi.BANDED_STAGES)."""
    b = np.empty(2 * len(banded[0]), np.int32)
    f = np.empty_like(b)
    b[0::2], b[1::2] = banded
    f[0::2], f[1::2] = full
    assert np.all(b <= f), "%s: a banded score above the full-matrix score" % label
    known = np.isin(stage, list(abi.STAGE_NAMES))
    assert np.all(known), "%s: unknown stage byte %d" % (label, int(stage[~known][0]))
    differ = b != f
    by_cert = differ & ~np.isin(stage, abi.BANDED_STAGES)
    assert not by_cert.any(), "%s: task %d has banded %d != full %d but was decided by stage %d" % (
        label, int(np.nonzero(by_cert)[0][0]), int(b[by_cert][0]), int(f[by_cert][0]), int(stage[by_cert][0]))
    return differ
