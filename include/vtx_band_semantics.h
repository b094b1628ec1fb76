/*
 * vtx_band_semantics.h — the three recollected details of bio 0.30.0's banded aligner that no vector held by the
 * reference pins, as ONE set of named constants shared by the device code
 * (vartrix_amd/csrc/vtx_band.hip) and the CPU oracle (oracle/vtx_oracle.c, oracle/vtx_certify.c).
 *
 * A maintainer holding bio-0.30.0/src/alignment/pairwise/banded.rs can correct parity here, in one place:
 *
 *   VTX_BAND_LAZY_EXT(k)          Band::set_boundaries with free clipping on both sequences (local mode): how far the
 *                                 band is extended diagonally past the first / last chained k-mer (clipped at the matrix
 *                                 edge).  Recollection: 2 * k.  Alternatives measured in tests/test_band_variants.py:
 *                                 0 (no extension) and VTX_BAND_EXT_TO_EDGE (to the matrix corner).
 *   VTX_BAND_KMER_LAST_ANCHOR(k)  Band::add_kmer: the anchors of a chained k-mer are the cells (r + d, c + d) for
 *                                 d = 0 .. VTX_BAND_KMER_LAST_ANCHOR(k).  Recollection: inclusive of k (k + 1 cells, the
 *                                 cell after the k-mer's last base).
 *   VTX_BAND_NO_SEED_FULL_MATRIX  no exact k-mer match at all: the whole matrix is in band.  Recollection: 1.
 *
 * The call site these implement is src/main.rs:898-901, `banded::Aligner::new(-5, -1, score, 6, 20)`.
 */
#ifndef VTX_BAND_SEMANTICS_H
#define VTX_BAND_SEMANTICS_H

#define VTX_BAND_EXT_TO_EDGE 0x7fffffff
#define VTX_BAND_LAZY_EXT(k) (2 * (k))
#define VTX_BAND_KMER_LAST_ANCHOR(k) (k)
#define VTX_BAND_NO_SEED_FULL_MATRIX 1

#endif
