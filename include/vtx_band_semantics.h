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
 *                                 cell after the k-mer's last base).  The alternative k - 1 changes NO band: that cell is
 *                                 the origin of add_gap / of set_boundaries' extension either way (tests/test_band_kat.py;
 *                                 libvtx_anchor5.so gives the production scores) — kept as a constant for completeness.
 *   VTX_BAND_NO_SEED_FULL_MATRIX  no exact k-mer match at all: the whole matrix is in band.  Recollection: 1.
 * (A fourth recollection, the tie rule of sdpkpp — the larger match index wins — is not a constant: it is the order of the
 * packed words the kernels maximise; oracle/vtx_oracle.c has the switch VTXO_VAR_TIE, tests/golden/band_kat.json 28 vectors
 * whose score tells the two rules apart.)
 *
 * A consequence that rests on the same recollections (round-5 ADVICE): the `whole_read` shortcut of band_diag_kernel / band_diag2_kernel
 * (vtx_fast_core.h) — a read that matches its haplotype base for base on one diagonal is scored m before any k-mer probe — assumes
 * sdpkpp's rules as recalled (a jump always costs gap_open + gap_extend per base, a match's dp never exceeds x + K) and the band's
 * end event at + k.  It agrees with oracle/vtx_oracle.c on every test (tests/golden/band_kat.json carries 24 such vectors, `whole_read`:
 * the Rust replay of INTEGRATION.md section 5 prints what the crate gives on them); should the crate's sdpkpp differ, libvtx_dev.so runs without
 * the shortcut under VTX_DIAG_ABLATE=10 (tests/test_gpu_stress.py::test_whole_read_shortcut_on_and_off) for an A/B comparison.
 *
 * The call site these implement is src/main.rs:898-901, `banded::Aligner::new(-5, -1, score, 6, 20)`.
 */
#ifndef VTX_BAND_SEMANTICS_H
#define VTX_BAND_SEMANTICS_H

#define VTX_BAND_EXT_TO_EDGE 0x7fffffff
/* (every constant may be overridden on the compiler command line: `make -C vartrix_amd/csrc variants` builds
 * libvtx_lazy0.so with -D'VTX_BAND_LAZY_EXT(k)=0' (and libvtx_anchor5.so, libvtx_noseed0.so), and tests/test_gpu_variants.py checks that the device then follows the
 * oracle run with the same override — a maintainer correcting one of these edits this header and nothing else) */
#ifndef VTX_BAND_LAZY_EXT
#define VTX_BAND_LAZY_EXT(k) (2 * (k))
#endif
#ifndef VTX_BAND_KMER_LAST_ANCHOR
#define VTX_BAND_KMER_LAST_ANCHOR(k) (k)
#endif
#ifndef VTX_BAND_NO_SEED_FULL_MATRIX
#define VTX_BAND_NO_SEED_FULL_MATRIX 1
#endif

/* The scoring the kernels, the certificate's proof (oracle/vtx_certify.c) and ub_join_same are derived for: the reference's
 * constants src/main.rs:33-38.  vtx_create rejects any other configuration; the device code static_asserts these values. */
#define VTX_REF_K 6
#define VTX_REF_W 20
#define VTX_REF_MATCH 1
#define VTX_REF_MISMATCH (-5)
#define VTX_REF_GAP_OPEN (-5)
#define VTX_REF_GAP_EXTEND (-1)

#endif
