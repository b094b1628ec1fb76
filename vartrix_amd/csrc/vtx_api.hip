// vtx_api.hip — C-ABI layer of libvtx.so (see include/vtx.h for the contract).
//
// Owns the HIP context state: device buffers sized for the resident batch, the
// per-bucket work lists, one stream, hipEvents for timing.  No CPU compute
// path exists here: without a device every entry point fails.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h>      // types and prototypes only: the library is dlopen'ed on first use (vtx_comm_*)
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "vtx_device.h"
#include "vtx_ingest.h"
#include "../../include/vtx_band_semantics.h"

extern "C" hipError_t vtxk_inclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, void* temp,
                                              size_t temp_bytes, hipStream_t s) {
    if (!n) return hipSuccess;
    return hipcub::DeviceScan::InclusiveSum(temp, temp_bytes, in, out, (int)n, s);
}
extern "C" size_t vtxk_scan_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    hipcub::DeviceScan::InclusiveSum(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)std::max(n, 1u));
    return bytes;
}

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T* as() const { return (T*)p; }
};

struct Bucket { int R, GL; uint32_t offset, count; bool lut; bool duo = false; bool pair = false; };
const uint32_t kLutLociCap = 4;     // loci tables per workgroup of the LUT kernel (6 KiB each at 257 columns)
const uint32_t kDuoLociCap = 8;     // the duo kernel runs 3 workgroups per CU (VGPRs), so up to 52 KiB of tables cost no occupancy

// (rows per lane, lanes per record) choices; capacity = R * GL read bases.
const int kShapes[][2] = {{2, 16}, {4, 16}, {6, 16}, {8, 16}, {10, 16}, {12, 16}, {16, 16}, {8, 64}, {16, 64}};
const int kNumShapes = sizeof(kShapes) / sizeof(kShapes[0]);
const uint32_t kFastReadLen = VTX_FAST_READ_LEN;   // 16 rows x 64 lanes
const uint32_t kFastHapLen = VTX_FAST_HAP_LEN;     // 16 record slots x (len + 35) words must fit 160 KiB of LDS
// Beyond the fast limits a record is scored by slow_align_kernel (exact, one lane per alignment); its 16-bit coordinates
// set the hard limits.
const uint32_t kMaxReadLen = 30000;
const uint32_t kMaxHapLen = 30000;

thread_local std::string g_create_err;

}  // namespace

template <class T>
struct HostArr {
    T* p = nullptr;
    size_t n = 0;
    ~HostArr() { free(p); }
    bool alloc(size_t count) {
        if (count > n || !p) { free(p); p = (T*)malloc(std::max<size_t>(count, 1) * sizeof(T)); n = p ? count : 0; }
        return p != nullptr;
    }
    T* data() { return p; }
};

struct vtx_ctx {
    vtx_config cfg{};
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;           // side stream: the general band kernel runs beside the pending / masked kernels
    hipEvent_t ev2 = nullptr;
    // multi-GPU row gather (vtx_comm_init / vtx_gather_coo)
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    DevBuf d_g_cnt, d_g_row, d_g_col, d_g_alt, d_g_ref, d_g_unk, d_g_val, d_g_refval;
    uint64_t g_nnz = 0;
    uint32_t* h_pin = nullptr;               // pinned words for counters read back asynchronously (a D2H copy into pageable
                                             // memory blocks the host until the stream reaches it)
    hipEvent_t ev[12] = {};
    std::string err;
    bool submitted = false, ran = false;
    uint32_t n_loci = 0, n_records = 0, n_cell_groups = 0, n_umi_groups = 0, max_hap_len = 0;
    uint64_t cells = 0, nnz = 0;
    std::vector<Bucket> buckets;
    vtx_timing timing{};
    DevBuf d_loci, d_records, d_rec_locus, d_hap, d_read, d_work, d_ref, d_alt;
    DevBuf d_head_cell, d_head_umi, d_cell_scan, d_umi_scan, d_grp_row, d_grp_col, d_umi_cellgrp;
    DevBuf d_cell_cnt, d_umi_cnt, d_keep, d_keep_scan, d_scan_tmp;
    DevBuf d_o_row, d_o_col, d_o_alt, d_o_ref, d_o_unk, d_o_val, d_o_refval;
    bool band_long_lists = false;      // (performance feedback between runs: see vtx_run)
    int read_format = VTX_READS_BYTES;     // vtx_set_read_format
    DevBuf d_read_packed;                  // VTX_READS_NIBBLES: the arena as uploaded, unpacked into d_read
    uint64_t gt_used = 0;      // bytes of d_gtables the last banded run's table kernel wrote (vtx_debug_tables)
    DevBuf d_band_ws, d_band_ws2, d_band, d_poly, d_gtables, d_hard, d_over, d_over2, d_pend, d_pend_buf, d_cnt, d_band2, d_hard2, d_fail, d_fail_tmp, d_refine;   // banded flavour
    DevBuf d_tight2, d_tight2_pack;                                          // band_diag2_kernel: tasks whose band is one diagonal stretch after all
    DevBuf d_recheck2, d_recheck2_pack;                                      // ... of which the full-matrix check did not settle (full != certificate); first they hold band_stream_kernel's task list and diagonals
    DevBuf d_sweep_log;                                                      // band_sweep_kernel: the section logs of the resident workgroups (48 MB: 1 536 x 8 x 1 024 words)
    DevBuf d_tight, d_tight_pack, d_dband, d_dband_pack, d_dense, d_stage;                                                 // round 4: tasks with a provisional score (full-matrix check); stage bytes (vtx_fetch_stage)
    bool stage_trace = false, poison = false;                                // test / audit hooks (vtx_set_debug)
    int32_t poison_value = 0;
    DevBuf d_redo, d_redo_cnt;                                               // LUT kernel: records with non-ACGTN bytes
    // raw batches (vtx_submit_raw): barcode table + preparation scratch
    DevBuf d_bc_slots, d_bc_hash, d_bc_off, d_bc_bytes;
    uint32_t bc_mask = 0;
    bool bc_ready = false;
    DevBuf d_raw, d_tags, d_raw_locus, d_key_lc, d_key_lc2, d_key_umi, d_key_umi2, d_idx, d_idx2, d_shape, d_shape2, d_seq,
        d_locus_cnt, d_locus_scan, d_prep_cnt, d_sort_tmp;
    // device-side BAM ingest (vtx_submit_bam, vtx_ingest.hip): compressed range, inflated stream, record offsets, per-record counts / scans
    DevBuf d_bam_comp, d_bam_data, d_bam_blocks, d_bam_seeds, d_bam_seed_cnt, d_bam_seed_scan, d_bam_iv, d_bam_cnt, d_bam_rec, d_bam_nhit,
        d_bam_rsz, d_bam_tsz, d_bam_hscan, d_bam_rscan, d_bam_tscan, d_bam_info;
    // vtx_prefetch_file: bytes [pf_off, pf_off + pf_n) of the BAM on their way into d_bam_comp (a library thread drives the copy workers)
    std::thread pf_thread;
    uint64_t pf_off = 0, pf_n = 0;
    int pf_rc = 0;
    void* pf_map = nullptr;           // the mapping the prefetch reads from (unmapped by the next prefetch / vtx_destroy)
    size_t pf_map_bytes = 0;
    float pf_ms = 0;                  // how long the prefetch's copy took
    bool pf_valid = false;
    uint32_t bam_n_rec = 0, bam_n_raw = 0;
    uint64_t bam_utotal = 0, bam_read_bases = 0, bam_tag_bytes = 0;
    uint32_t max_read_len = 0, fast_overflow = 0;
    uint32_t slow_off = 0, slow_cnt = 0, max_hap_all = 0, max_read_all = 0;   // records of the slow list (d_work[slow_off ..])
    // a batch that mixes haplotypes of <= 255 bases with longer ones (note_long_loci; vtx_run's two banded passes): the longest haplotype
    // among the former, how many loci the latter are, the first and the last of them
    uint32_t hap_short_max = 0, n_long_loci = 0, long_first = 0, long_last = 0;
    DevBuf d_slow_ws, d_slow_retry;
    // host -> device feed: pinned staging buffers + one stream per copy worker (see upload())
    static constexpr int kUpWorkers = 6, kUpSlots = 2;
    static constexpr size_t kUpChunk = 8u << 20;
    void* up_pin[kUpWorkers][kUpSlots] = {};
    hipEvent_t up_ev[kUpWorkers][kUpSlots] = {};
    hipStream_t up_stream[kUpWorkers] = {};
    bool up_ready = false;
    HostArr<uint32_t> h_row, h_col, h_alt, h_ref, h_unk;   // fetched triplets (uninitialised storage: written by the download)
    HostArr<double> h_val, h_refval;
};

namespace {

int fail(vtx_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                       \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail((c), _e == hipErrorOutOfMemory ? VTX_E_NOMEM : VTX_E_HIP, "%s: %s", #expr, \
                        hipGetErrorString(_e));                                                \
    } while (0)

// ---- host -> device feed ------------------------------------------------------------------------------------
// The caller's arrays are pageable: a plain hipMemcpy of 3.65 GB (config 3's read arena) runs at ~17 GB/s through
// the runtime's single staging path.  Here kUpWorkers threads each copy 8 MiB chunks into their own pinned buffers
// (two per worker, so the memcpy of chunk k+1 overlaps the DMA of chunk k) and push them on their own stream: the
// host memcpys and the DMAs of different workers overlap, and the link — not one core's memcpy — is the limit.
struct UploadJob { void* dst; const void* src; size_t bytes; };

int upload_init(vtx_ctx* c) {
    if (c->up_ready) return VTX_OK;
    for (int w = 0; w < vtx_ctx::kUpWorkers; ++w) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->up_stream[w], hipStreamNonBlocking));
        for (int k = 0; k < vtx_ctx::kUpSlots; ++k) {
            HIP_TRY(c, hipHostMalloc(&c->up_pin[w][k], vtx_ctx::kUpChunk, hipHostMallocDefault));
            HIP_TRY(c, hipEventCreateWithFlags(&c->up_ev[w][k], hipEventDisableTiming));
        }
    }
    c->up_ready = true;
    return VTX_OK;
}

void upload_release(vtx_ctx* c) {
    for (int w = 0; w < vtx_ctx::kUpWorkers; ++w) {
        for (int k = 0; k < vtx_ctx::kUpSlots; ++k) {
            if (c->up_pin[w][k]) (void)hipHostFree(c->up_pin[w][k]);
            if (c->up_ev[w][k]) (void)hipEventDestroy(c->up_ev[w][k]);
            c->up_pin[w][k] = nullptr; c->up_ev[w][k] = nullptr;
        }
        if (c->up_stream[w]) (void)hipStreamDestroy(c->up_stream[w]);
        c->up_stream[w] = nullptr;
    }
    c->up_ready = false;
}

// Copies every job to the device and returns when all bytes have landed.
int upload(vtx_ctx* c, const std::vector<UploadJob>& jobs) {
    struct Chunk { char* dst; const char* src; size_t bytes; };
    std::vector<Chunk> chunks;
    size_t total = 0;
    for (const UploadJob& j : jobs)
        for (size_t o = 0; o < j.bytes; o += vtx_ctx::kUpChunk) {
            chunks.push_back(Chunk{(char*)j.dst + o, (const char*)j.src + o, std::min(vtx_ctx::kUpChunk, j.bytes - o)});
            total += chunks.back().bytes;
        }
    if (chunks.empty()) return VTX_OK;
    if (total < (4u << 20)) {                    // small batches: not worth the threads
        for (const Chunk& ch : chunks) HIP_TRY(c, hipMemcpyAsync(ch.dst, ch.src, ch.bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return VTX_OK;
    }
    if (int rc = upload_init(c)) return rc;
    std::atomic<size_t> next{0};
    std::atomic<int> err{(int)hipSuccess};
    const int device = c->cfg.device;
    auto worker = [&](int w) {
        if (hipSetDevice(device) != hipSuccess) { err = (int)hipErrorInvalidDevice; return; }
        bool used[vtx_ctx::kUpSlots] = {};
        int slot = 0;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= chunks.size() || err.load() != (int)hipSuccess) break;
            hipError_t e = used[slot] ? hipEventSynchronize(c->up_ev[w][slot]) : hipSuccess;   // the DMA out of this buffer is done
            if (e == hipSuccess) {
                memcpy(c->up_pin[w][slot], chunks[i].src, chunks[i].bytes);
                e = hipMemcpyAsync(chunks[i].dst, c->up_pin[w][slot], chunks[i].bytes, hipMemcpyHostToDevice, c->up_stream[w]);
            }
            if (e == hipSuccess) e = hipEventRecord(c->up_ev[w][slot], c->up_stream[w]);
            if (e != hipSuccess) { err = (int)e; break; }
            used[slot] = true;
            slot = (slot + 1) % vtx_ctx::kUpSlots;
        }
        const hipError_t e = hipStreamSynchronize(c->up_stream[w]);
        if (e != hipSuccess) err = (int)e;
    };
    const int nw = (int)std::min<size_t>(vtx_ctx::kUpWorkers, chunks.size());
    std::vector<std::thread> th;
    for (int w = 1; w < nw; ++w) th.emplace_back(worker, w);
    worker(0);
    for (auto& t : th) t.join();
    if (err.load() != (int)hipSuccess)
        return fail(c, VTX_E_HIP, "upload: %s", hipGetErrorString((hipError_t)err.load()));
    return VTX_OK;
}

// The way back: device arrays into the caller-visible (pageable, freshly allocated) host arrays.  Same workers and
// pinned buffers as the upload: a worker keeps one DMA in flight while it copies the previous chunk out of its other
// pinned buffer, so the page faults of the fresh host pages are spread over the workers.
int download(vtx_ctx* c, const std::vector<UploadJob>& jobs) {     // dst = host, src = device
    struct Chunk { char* dst; const char* src; size_t bytes; };
    std::vector<Chunk> chunks;
    size_t total = 0;
    for (const UploadJob& j : jobs)
        for (size_t o = 0; o < j.bytes; o += vtx_ctx::kUpChunk) {
            chunks.push_back(Chunk{(char*)j.dst + o, (const char*)j.src + o, std::min(vtx_ctx::kUpChunk, j.bytes - o)});
            total += chunks.back().bytes;
        }
    if (chunks.empty()) return VTX_OK;
    if (total < (4u << 20)) {
        for (const Chunk& ch : chunks) HIP_TRY(c, hipMemcpy(ch.dst, ch.src, ch.bytes, hipMemcpyDeviceToHost));
        return VTX_OK;
    }
    if (int rc = upload_init(c)) return rc;
    std::atomic<size_t> next{0};
    std::atomic<int> err{(int)hipSuccess};
    const int device = c->cfg.device;
    auto worker = [&](int w) {
        if (hipSetDevice(device) != hipSuccess) { err = (int)hipErrorInvalidDevice; return; }
        size_t held[vtx_ctx::kUpSlots];
        bool used[vtx_ctx::kUpSlots] = {};
        auto drain = [&](int slot) -> hipError_t {
            if (!used[slot]) return hipSuccess;
            used[slot] = false;
            const hipError_t e = hipEventSynchronize(c->up_ev[w][slot]);
            if (e == hipSuccess) memcpy(chunks[held[slot]].dst, c->up_pin[w][slot], chunks[held[slot]].bytes);
            return e;
        };
        int slot = 0;
        for (;;) {
            const size_t i = next.fetch_add(1);
            hipError_t e = drain(slot);
            if (e == hipSuccess && i < chunks.size() && err.load() == (int)hipSuccess) {
                e = hipMemcpyAsync(c->up_pin[w][slot], chunks[i].src, chunks[i].bytes, hipMemcpyDeviceToHost, c->up_stream[w]);
                if (e == hipSuccess) e = hipEventRecord(c->up_ev[w][slot], c->up_stream[w]);
                if (e == hipSuccess) { used[slot] = true; held[slot] = i; }
            } else {
                for (int k = 1; k < vtx_ctx::kUpSlots && e == hipSuccess; ++k) e = drain((slot + k) % vtx_ctx::kUpSlots);
                if (e != hipSuccess) err = (int)e;
                break;
            }
            if (e != hipSuccess) { err = (int)e; break; }
            slot = (slot + 1) % vtx_ctx::kUpSlots;
        }
        (void)hipStreamSynchronize(c->up_stream[w]);
    };
    const int nw = (int)std::min<size_t>(vtx_ctx::kUpWorkers, chunks.size());
    std::vector<std::thread> th;
    for (int w = 1; w < nw; ++w) th.emplace_back(worker, w);
    worker(0);
    for (auto& t : th) t.join();
    if (err.load() != (int)hipSuccess)
        return fail(c, VTX_E_HIP, "download: %s", hipGetErrorString((hipError_t)err.load()));
    return VTX_OK;
}

// Device bytes straight into a file: the copy workers pwrite() their pinned buffers at the chunk's file offset — one host copy (pinned
// buffer -> page cache), spread over the workers, instead of two (pinned -> caller's array -> page cache).
int download_to_fd(vtx_ctx* c, int fd, uint64_t file_off, const void* d_src, size_t bytes) {
    if (!bytes) return VTX_OK;
    if (int rc = upload_init(c)) return rc;
    const size_t n_chunks = (bytes + vtx_ctx::kUpChunk - 1) / vtx_ctx::kUpChunk;
    std::atomic<size_t> next{0};
    std::atomic<int> err{(int)hipSuccess};
    std::atomic<bool> io_err{false};
    const int device = c->cfg.device;
    auto worker = [&](int w) {
        if (hipSetDevice(device) != hipSuccess) { err = (int)hipErrorInvalidDevice; return; }
        size_t held[vtx_ctx::kUpSlots];
        bool used[vtx_ctx::kUpSlots] = {};
        auto drain = [&](int slot) -> hipError_t {
            if (!used[slot]) return hipSuccess;
            used[slot] = false;
            const hipError_t e = hipEventSynchronize(c->up_ev[w][slot]);
            if (e != hipSuccess) return e;
            const size_t o = held[slot] * vtx_ctx::kUpChunk, n = std::min(vtx_ctx::kUpChunk, bytes - o);
            const char* p = (const char*)c->up_pin[w][slot];
            size_t done = 0;
            while (done < n) {
                const ssize_t k = pwrite(fd, p + done, n - done, (off_t)(file_off + o + done));
                if (k <= 0) { io_err = true; break; }
                done += (size_t)k;
            }
            return hipSuccess;
        };
        int slot = 0;
        for (;;) {
            const size_t i = next.fetch_add(1);
            hipError_t e = drain(slot);
            if (e == hipSuccess && i < n_chunks && err.load() == (int)hipSuccess && !io_err.load()) {
                const size_t o = i * vtx_ctx::kUpChunk, n = std::min(vtx_ctx::kUpChunk, bytes - o);
                e = hipMemcpyAsync(c->up_pin[w][slot], (const char*)d_src + o, n, hipMemcpyDeviceToHost, c->up_stream[w]);
                if (e == hipSuccess) e = hipEventRecord(c->up_ev[w][slot], c->up_stream[w]);
                if (e == hipSuccess) { used[slot] = true; held[slot] = i; }
            } else {
                for (int k = 1; k < vtx_ctx::kUpSlots && e == hipSuccess; ++k) e = drain((slot + k) % vtx_ctx::kUpSlots);
                if (e != hipSuccess) err = (int)e;
                break;
            }
            if (e != hipSuccess) { err = (int)e; break; }
            slot = (slot + 1) % vtx_ctx::kUpSlots;
        }
        (void)hipStreamSynchronize(c->up_stream[w]);
    };
    const int nw = (int)std::min<size_t>(vtx_ctx::kUpWorkers, n_chunks);
    std::vector<std::thread> th;
    for (int w = 1; w < nw; ++w) th.emplace_back(worker, w);
    worker(0);
    for (auto& t : th) t.join();
    if (err.load() != (int)hipSuccess) return fail(c, VTX_E_HIP, "download: %s", hipGetErrorString((hipError_t)err.load()));
    if (io_err.load()) return fail(c, VTX_E_INVAL, "error writing the output file");
    return VTX_OK;
}

// the seven triplet arrays of n entries, device -> host
int fetch_arrays(vtx_ctx* c, size_t n, const void* row, const void* col, const void* alt, const void* ref, const void* unk,
                 const void* val, const void* refval, vtx_coo* out) {
    if (!c->h_row.alloc(n) || !c->h_col.alloc(n) || !c->h_alt.alloc(n) || !c->h_ref.alloc(n) || !c->h_unk.alloc(n) ||
        !c->h_val.alloc(n) || !c->h_refval.alloc(n))
        return fail(c, VTX_E_NOMEM, "out of host memory for %zu triplets", n);
    if (n) {
        if (int rc = download(c, {{c->h_row.data(), row, n * 4}, {c->h_col.data(), col, n * 4}, {c->h_alt.data(), alt, n * 4},
                                  {c->h_ref.data(), ref, n * 4}, {c->h_unk.data(), unk, n * 4}, {c->h_val.data(), val, n * 8},
                                  {c->h_refval.data(), refval, n * 8}}))
            return rc;
    }
    out->row = c->h_row.data(); out->col = c->h_col.data(); out->alt = c->h_alt.data(); out->ref = c->h_ref.data();
    out->unk = c->h_unk.data(); out->value = c->h_val.data(); out->ref_value = c->h_refval.data();
    out->nnz = n;
    return VTX_OK;
}

#ifdef VTX_DEVTOOLS
extern "C" void* vtxt_comm_test_table(void** fns);          // vtx_comm_test.hip (linked into libvtx_dev.so only)
#endif
// ---- RCCL, opened on first use ---------------------------------------------------------------------------------
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
};
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;          // vtx_comm_init is called from several threads at once (one context per GPU)
    std::call_once(once, [] {
#ifdef VTX_DEVTOOLS
        if (VTX_DEV_ENV("VTX_COMM_TEST_TRANSPORT")) {
            // TEST TRANSPORT (vtx_comm_test.hip, libvtx_dev.so only): ranks = processes that may share one device, payloads over Unix
            // sockets in that directory — the exchange's own logic with world > 1 on a one-GPU box.  Not in the production library.
            void* f[10];
            r.lib = vtxt_comm_test_table(f);
            r.GetUniqueId = (decltype(r.GetUniqueId))f[0]; r.CommInitRank = (decltype(r.CommInitRank))f[1];
            r.CommDestroy = (decltype(r.CommDestroy))f[2]; r.AllGather = (decltype(r.AllGather))f[3];
            r.Send = (decltype(r.Send))f[4]; r.Recv = (decltype(r.Recv))f[5]; r.GroupStart = (decltype(r.GroupStart))f[6];
            r.GroupEnd = (decltype(r.GroupEnd))f[7]; r.GetErrorString = (decltype(r.GetErrorString))f[8]; r.CommCount = (decltype(r.CommCount))f[9];
            return;
        }
#endif
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (r.lib) {
#define VTX_SYM(f) r.f = (decltype(r.f))dlsym(r.lib, "nccl" #f)
            VTX_SYM(GetUniqueId); VTX_SYM(CommInitRank); VTX_SYM(CommDestroy); VTX_SYM(AllGather); VTX_SYM(Send); VTX_SYM(Recv);
            VTX_SYM(GroupStart); VTX_SYM(GroupEnd); VTX_SYM(GetErrorString); VTX_SYM(CommCount);
#undef VTX_SYM
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.Send || !r.Recv || !r.GroupStart ||
                !r.GroupEnd || !r.GetErrorString) { dlclose(r.lib); r.lib = nullptr; }
        }
    });
    return r.lib ? &r : nullptr;
}
#define NCCL_TRY(c, expr)                                                                                   \
    do {                                                                                                    \
        ncclResult_t _r = (expr);                                                                           \
        if (_r != ncclSuccess) return fail((c), VTX_E_HIP, "%s: %s", #expr, rccl()->GetErrorString(_r));    \
    } while (0)

void comm_release(vtx_ctx* c) {
    if (c->comm && rccl()) (void)rccl()->CommDestroy(c->comm);
    c->comm = nullptr; c->comm_world = 0;
}

// per-record device buffers of the resident batch (scores, group structure, COO staging)
int reserve_record_buffers(vtx_ctx* c, uint32_t nr) {
    const size_t u32 = sizeof(uint32_t);
    HIP_TRY(c, c->d_records.reserve((size_t)nr * sizeof(vtx_record)));
    HIP_TRY(c, c->d_rec_locus.reserve(nr * u32));
    HIP_TRY(c, c->d_work.reserve(nr * u32));
    HIP_TRY(c, c->d_redo.reserve(nr * u32));
    HIP_TRY(c, c->d_redo_cnt.reserve(16 * u32));
    HIP_TRY(c, c->d_ref.reserve(nr * sizeof(int32_t)));
    HIP_TRY(c, c->d_alt.reserve(nr * sizeof(int32_t)));
    DevBuf* per_rec[] = {&c->d_head_cell, &c->d_head_umi, &c->d_cell_scan, &c->d_umi_scan, &c->d_grp_row, &c->d_grp_col,
                         &c->d_umi_cellgrp, &c->d_keep, &c->d_keep_scan, &c->d_o_row, &c->d_o_col, &c->d_o_alt,
                         &c->d_o_ref, &c->d_o_unk};
    for (DevBuf* d : per_rec) HIP_TRY(c, d->reserve(nr * u32));
    HIP_TRY(c, c->d_cell_cnt.reserve(3 * (size_t)nr * u32));
    HIP_TRY(c, c->d_umi_cnt.reserve(3 * (size_t)nr * u32));
    HIP_TRY(c, c->d_o_val.reserve(nr * sizeof(double)));
    HIP_TRY(c, c->d_o_refval.reserve(nr * sizeof(double)));
    HIP_TRY(c, c->d_scan_tmp.reserve(vtxk_scan_temp_bytes(nr)));
    return VTX_OK;
}

// (row, cell) / (row, cell, umi) group structure of the resident records: depends only on the records.
// Leaves the two group counts in flight on the stream (the caller synchronises).
int build_groups(vtx_ctx* c, uint32_t nr) {
    hipStream_t s = c->stream;
    const size_t tmp_bytes = vtxk_scan_temp_bytes(nr);
    c->n_cell_groups = c->n_umi_groups = 0;
    if (!nr) return VTX_OK;
    HIP_TRY(c, vtxk_group_heads(c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(), nr,
                                c->d_head_cell.as<uint32_t>(), c->d_head_umi.as<uint32_t>(), s));
    HIP_TRY(c, vtxk_inclusive_scan_u32(c->d_head_cell.as<uint32_t>(), c->d_cell_scan.as<uint32_t>(), nr, c->d_scan_tmp.p, tmp_bytes, s));
    HIP_TRY(c, vtxk_inclusive_scan_u32(c->d_head_umi.as<uint32_t>(), c->d_umi_scan.as<uint32_t>(), nr, c->d_scan_tmp.p, tmp_bytes, s));
    HIP_TRY(c, vtxk_group_table(c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(), c->d_loci.as<vtx_locus>(), nr,
                                c->d_head_cell.as<uint32_t>(), c->d_head_umi.as<uint32_t>(), c->d_cell_scan.as<uint32_t>(),
                                c->d_umi_scan.as<uint32_t>(), c->d_grp_row.as<uint32_t>(), c->d_grp_col.as<uint32_t>(),
                                c->d_umi_cellgrp.as<uint32_t>(), s));
    HIP_TRY(c, hipMemcpyAsync(&c->n_cell_groups, c->d_cell_scan.as<uint32_t>() + (nr - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(&c->n_umi_groups, c->d_umi_scan.as<uint32_t>() + (nr - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    return VTX_OK;
}

// tables per workgroup of the duo kernel for this haplotype length: as many as fit 52 KiB, at most kDuoLociCap
uint32_t duo_loci_cap(uint32_t max_hap) {
    const size_t table_bytes = ((size_t)std::max(max_hap, 16u) + 36) * 6 * 4;
    return (uint32_t)std::min<size_t>(kDuoLociCap, (52 * 1024) / table_bytes);
}

// pair-table mode of the duo kernel (deep data: a 32-record workgroup spans <= 2 loci): columns of the per-locus
// pair table (16 sentinels + prefix) that fit the same 52 KiB next to the two single tables; 0 = mode not available
const uint32_t kPairLociCap = 2;
uint32_t duo_pair_cols(uint32_t max_hap) {
    const size_t lcols = 16 + (size_t)std::max(max_hap, 16u) + 16 + 4;
    const size_t single = kPairLociCap * lcols * 6 * 4;
    if (single + kPairLociCap * (16 + 32) * 38 * 4 > 52 * 1024) return 0;
    const size_t fit = (52 * 1024 - single) / (kPairLociCap * 38 * 4);
    const size_t cols = std::min<size_t>(fit, 16 + (size_t)max_hap);
    // the shared prefix is about half of the haplotype (the padding): a pair table that cannot hold it would cut the
    // sharing short (T1 is clamped to the table), and the two-lookup mode is the better choice then
    return cols >= 16 + (size_t)max_hap / 2 + 4 ? (uint32_t)cols : 0u;
}

// Kernel choice per work list (one list per read-length shape, already in d_work): LUT / shared-prefix (duo) /
// pair-table eligibility is a property of how many loci a workgroup of consecutive list entries spans.
int make_buckets(vtx_ctx* c, const uint32_t* shape_cnt, uint32_t max_hap, uint32_t* d_lut_flag) {
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemsetAsync(d_lut_flag, 0, 48 * sizeof(uint32_t), s));
    c->buckets.clear();
    uint32_t off = 0;
    for (int sh = 0; sh < kNumShapes; ++sh) {
        if (!shape_cnt[sh]) continue;
        const bool lut = kShapes[sh][1] == 16 && (size_t)kLutLociCap * (max_hap + 36) * 6 * 4 <= 96 * 1024;
        const uint32_t dcap = duo_loci_cap(max_hap);
        const bool duo = kShapes[sh][1] == 16 && dcap >= 2;
        if (lut) HIP_TRY(c, vtxk_prep_lut_check(c->d_work.as<uint32_t>() + off, shape_cnt[sh], c->d_rec_locus.as<uint32_t>(), kLutLociCap, 16,
                                                d_lut_flag + c->buckets.size(), s));
        if (duo) HIP_TRY(c, vtxk_prep_lut_check(c->d_work.as<uint32_t>() + off, shape_cnt[sh], c->d_rec_locus.as<uint32_t>(), dcap, 32,
                                                d_lut_flag + 16 + c->buckets.size(), s));
        const bool pair = duo && duo_pair_cols(max_hap) != 0;
        if (pair) HIP_TRY(c, vtxk_prep_lut_check(c->d_work.as<uint32_t>() + off, shape_cnt[sh], c->d_rec_locus.as<uint32_t>(), kPairLociCap, 32,
                                                 d_lut_flag + 32 + c->buckets.size(), s));
        Bucket bk{kShapes[sh][0], kShapes[sh][1], off, shape_cnt[sh], lut};
        bk.duo = duo; bk.pair = pair;
        c->buckets.push_back(bk);
        off += shape_cnt[sh];
    }
    c->slow_off = off; c->slow_cnt = shape_cnt[kNumShapes];          // the slow list sorts last (shape index kNumShapes)
    uint32_t lut_flag[48] = {0};
    HIP_TRY(c, hipMemcpyAsync(lut_flag, d_lut_flag, sizeof lut_flag, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    for (size_t i = 0; i < c->buckets.size(); ++i) {
        if (lut_flag[i]) c->buckets[i].lut = false;
        if (lut_flag[16 + i]) c->buckets[i].duo = false;
        if (lut_flag[16 + i] || lut_flag[32 + i]) c->buckets[i].pair = false;
    }
    return VTX_OK;
}

uint32_t bits_for(uint64_t max_value) {   // bits needed to represent values 0..max_value
    uint32_t b = 1;
    while (b < 64 && (max_value >> b)) ++b;
    return b;
}

}  // namespace

extern "C" {

void vtx_config_default(vtx_config* cfg) {
    if (!cfg) return;
    memset(cfg, 0, sizeof *cfg);
    cfg->abi_version = VTX_ABI_VERSION;
    cfg->device = 0;
    cfg->aligner = VTX_ALIGNER_BANDED;
    cfg->scoring_mode = VTX_MODE_CONSENSUS;
    cfg->use_umi = 0;
    cfg->match_score = 1;
    cfg->mismatch_score = -5;
    cfg->gap_open = -5;
    cfg->gap_extend = -1;
    cfg->min_score = 25;
    cfg->kmer_k = 6;
    cfg->band_w = 20;
}

int vtx_abi_sizes(uint32_t* out, uint32_t n) {
    const uint32_t s[13] = {(uint32_t)sizeof(vtx_config), (uint32_t)sizeof(vtx_locus), (uint32_t)sizeof(vtx_record),
                            (uint32_t)sizeof(vtx_batch), (uint32_t)sizeof(vtx_coo), (uint32_t)sizeof(vtx_timing),
                            (uint32_t)sizeof(vtx_raw_record), (uint32_t)sizeof(vtx_raw_batch), (uint32_t)sizeof(vtx_raw_stats),
                            (uint32_t)sizeof(vtx_bgzf_block), (uint32_t)sizeof(vtx_bam_interval), (uint32_t)sizeof(vtx_bam_ingest),
                            (uint32_t)sizeof(vtx_ingest_stats)};
    for (uint32_t i = 0; i < n && i < 13; ++i) out[i] = s[i];
    return VTX_ABI_VERSION;
}

const char* vtx_status_name(int status) {
    switch (status) {
    case VTX_OK: return "VTX_OK";
    case VTX_E_INVAL: return "VTX_E_INVAL";
    case VTX_E_NODEVICE: return "VTX_E_NODEVICE";
    case VTX_E_HIP: return "VTX_E_HIP";
    case VTX_E_NOMEM: return "VTX_E_NOMEM";
    case VTX_E_UNSUPPORTED: return "VTX_E_UNSUPPORTED";
    case VTX_E_STATE: return "VTX_E_STATE";
    case VTX_E_PEER: return "VTX_E_PEER";
    default: return "VTX_E_?";
    }
}

const char* vtx_strerror(const vtx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int vtx_create(const vtx_config* cfg, vtx_ctx** out) {
    if (!cfg || !out) return fail(nullptr, VTX_E_INVAL, "vtx_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != VTX_ABI_VERSION)
        return fail(nullptr, VTX_E_INVAL, "vtx_create: abi_version %d != %d", cfg->abi_version, VTX_ABI_VERSION);
    if (cfg->aligner != VTX_ALIGNER_BANDED && cfg->aligner != VTX_ALIGNER_FULL)
        return fail(nullptr, VTX_E_INVAL, "vtx_create: unknown aligner %d", cfg->aligner);
    if (cfg->scoring_mode < VTX_MODE_CONSENSUS || cfg->scoring_mode > VTX_MODE_COVERAGE)
        return fail(nullptr, VTX_E_INVAL, "vtx_create: unknown scoring_mode %d", cfg->scoring_mode);
    // The kernels bake the reference's scoring constants (src/main.rs:33-38) in as immediates.
    if (cfg->match_score != VTX_REF_MATCH || cfg->mismatch_score != VTX_REF_MISMATCH || cfg->gap_open != VTX_REF_GAP_OPEN ||
        cfg->gap_extend != VTX_REF_GAP_EXTEND)
        return fail(nullptr, VTX_E_UNSUPPORTED,
                    "vtx_create: only the reference scoring (+1/-5, gap -5/-1; src/main.rs:35-38) is built");
    if (cfg->kmer_k != VTX_REF_K || cfg->band_w != VTX_REF_W)
        return fail(nullptr, VTX_E_UNSUPPORTED, "vtx_create: only K=6, W=20 (src/main.rs:33-34) is built");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, VTX_E_NODEVICE, "vtx_create: no HIP device (%s); this library has no CPU path",
                    e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, VTX_E_INVAL, "vtx_create: device %d out of range (%d devices)", cfg->device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
        return fail(nullptr, VTX_E_HIP, "vtx_create: hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, VTX_E_NODEVICE, "vtx_create: device %d is %s; kernels are built for gfx950 only",
                    cfg->device, prop.gcnArchName);
    vtx_ctx* c = new (std::nothrow) vtx_ctx();
    if (!c) return fail(nullptr, VTX_E_NOMEM, "vtx_create: out of host memory");
    c->cfg = *cfg;
    if (hipSetDevice(cfg->device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(nullptr, VTX_E_HIP, "vtx_create: stream creation failed");
    }
    for (auto& ev : c->ev)
        if (hipEventCreate(&ev) != hipSuccess) { vtx_destroy(c); return fail(nullptr, VTX_E_HIP, "vtx_create: event creation failed"); }
    *out = c;
    return VTX_OK;
}

void vtx_destroy(vtx_ctx* c) {
    if (!c) return;
    if (c->pf_thread.joinable()) c->pf_thread.join();
    if (c->pf_map) { munmap(c->pf_map, c->pf_map_bytes); c->pf_map = nullptr; }
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->d_loci, &c->d_records, &c->d_rec_locus, &c->d_hap, &c->d_read, &c->d_work, &c->d_ref,
                      &c->d_alt, &c->d_head_cell, &c->d_head_umi, &c->d_cell_scan, &c->d_umi_scan, &c->d_grp_row,
                      &c->d_grp_col, &c->d_umi_cellgrp, &c->d_cell_cnt, &c->d_umi_cnt, &c->d_keep, &c->d_keep_scan,
                      &c->d_scan_tmp, &c->d_o_row, &c->d_o_col, &c->d_o_alt, &c->d_o_ref, &c->d_o_unk, &c->d_o_val,
                      &c->d_o_refval, &c->d_band_ws, &c->d_band_ws2, &c->d_band, &c->d_poly, &c->d_gtables, &c->d_hard, &c->d_over, &c->d_over2, &c->d_pend, &c->d_pend_buf, &c->d_band2, &c->d_hard2,
                      &c->d_cnt, &c->d_redo, &c->d_redo_cnt, &c->d_bc_slots, &c->d_bc_hash, &c->d_bc_off, &c->d_bc_bytes,
                      &c->d_raw, &c->d_tags, &c->d_raw_locus, &c->d_key_lc, &c->d_key_lc2, &c->d_key_umi, &c->d_key_umi2,
                      &c->d_idx, &c->d_idx2, &c->d_shape, &c->d_shape2, &c->d_seq, &c->d_locus_cnt, &c->d_locus_scan,
                      &c->d_prep_cnt, &c->d_sort_tmp, &c->d_fail, &c->d_fail_tmp, &c->d_refine, &c->d_tight, &c->d_tight_pack, &c->d_dband, &c->d_dband_pack, &c->d_dense, &c->d_stage, &c->d_sweep_log, &c->d_tight2, &c->d_tight2_pack, &c->d_recheck2, &c->d_recheck2_pack};
    for (DevBuf* b : bufs) b->release();
    c->d_slow_ws.release(); c->d_slow_retry.release(); c->d_read_packed.release();
    DevBuf* ib[] = {&c->d_bam_comp, &c->d_bam_data, &c->d_bam_blocks, &c->d_bam_seeds, &c->d_bam_seed_cnt, &c->d_bam_seed_scan, &c->d_bam_iv, &c->d_bam_cnt,
                    &c->d_bam_rec, &c->d_bam_nhit, &c->d_bam_rsz, &c->d_bam_tsz, &c->d_bam_hscan, &c->d_bam_rscan, &c->d_bam_tscan, &c->d_bam_info};
    for (DevBuf* b : ib) b->release();
    DevBuf* gb[] = {&c->d_g_cnt, &c->d_g_row, &c->d_g_col, &c->d_g_alt, &c->d_g_ref, &c->d_g_unk, &c->d_g_val, &c->d_g_refval};
    for (DevBuf* b : gb) b->release();
    comm_release(c);
    upload_release(c);
    for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->ev2) (void)hipEventDestroy(c->ev2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// Which loci of the batch have a haplotype above 255 bases (and within the fast kernels' limit)?  One of them used to put the WHOLE batch
// on round 3's path (four-byte match entries, no sweep, no second stage); vtx_run now scores them in a pass of their own (round 6).
static void note_long_loci(vtx_ctx* c, const vtx_locus* loci, uint32_t nl) {
    c->hap_short_max = c->n_long_loci = c->long_first = c->long_last = 0;
    for (uint32_t l = 0; l < nl; ++l) {
        const uint32_t hl = std::max(loci[l].ref_len, loci[l].alt_len);
        if (hl <= 255) c->hap_short_max = std::max(c->hap_short_max, hl);
        else if (hl <= kFastHapLen) {
            if (!c->n_long_loci++) c->long_first = l;
            c->long_last = l;
        }
    }
}

// ---- buffers of the banded stage -------------------------------------------------------------------------------
// One launch covers every task unless a test hook asks for chunks.  The scratch of band_run_kernel belongs to its
// RESIDENT lanes (persistent workgroups).  Hard tasks leave band_run_kernel / band_pending_kernel as compact staircase
// records (192 B); the masked DP expands them into band slots (2 x band_stride u16) one slice of `slots` tasks at a
// time.  A quarter of the tasks may be hard (noisy reads: 11 % at 3 % substitution errors) before anything spills to
// the general kernel's list.  gt_bytes: the k-mer tables of every locus in global memory (0: tables in LDS).
// records band_diag_kernel may leave for band_refine_kernel per chunk of tasks (a quarter of the tasks: 22 % are listed at 8 %
// substitution errors; what does not fit goes to band_run_kernel as before)
// (round 6: with a tight list EVERY task that leaves band_diag_kernel with a certificate leaves a record, for band_corridor_kernel: 24 - 30 %
// of the tasks at 8 % errors (39 % measured: 19.1 M of 48.6 M) — half; a task that does not fit takes the masked DP over its band, as all of them did before)
static uint32_t band_refine_cap(uint32_t chunk) { return std::max(65536u, chunk / 2); }
struct BandPlan {
    uint64_t n_tasks = 0;
    uint32_t chunk = 0, band_stride = 0, hard_cap = 0, pend_cap = 0, slots = 0, poly_stride = 0, tasks_per_locus = 0;
    size_t gt_bytes = 0;
    uint32_t gt_loci = 0;      // loci the table buffer holds (< n_loci: the stage runs in chunks of tasks whose loci fit)
};
static BandPlan band_plan(uint32_t nr, uint32_t n_loci, uint32_t max_hap_len) {
    BandPlan p;
    p.n_tasks = 2ull * nr;
    uint64_t chunk_cap = 1u << 31;
    if (VTX_DEV_ENV("VTX_BAND_CHUNK")) chunk_cap = std::max<uint64_t>(256, strtoull(VTX_DEV_ENV("VTX_BAND_CHUNK"), nullptr, 10));   // test hook
    p.chunk = (uint32_t)std::min<uint64_t>(p.n_tasks, chunk_cap);
    p.band_stride = (max_hap_len + 2 + 7) & ~7u;
    p.hard_cap = (uint32_t)std::min<uint64_t>(p.chunk, p.chunk / 4 + (1u << 20));
    p.pend_cap = (uint32_t)std::min<uint64_t>(p.chunk, p.chunk / 8 + (1u << 20));
    p.slots = (uint32_t)std::min<uint64_t>(p.chunk, p.chunk / 16 + (1u << 20));
    if (VTX_DEV_ENV("VTX_BAND_HARD_CAP")) p.hard_cap = p.pend_cap = p.slots = std::max(1u, (uint32_t)atoi(VTX_DEV_ENV("VTX_BAND_HARD_CAP")));   // test hook
    if (VTX_DEV_ENV("VTX_BAND_SLOTS")) p.slots = std::max(1u, (uint32_t)atoi(VTX_DEV_ENV("VTX_BAND_SLOTS")));                                  // test hook
    p.poly_stride = vtxk_band_poly_stride();
    p.tasks_per_locus = (uint32_t)(p.n_tasks / std::max(n_loci, 1u));
    p.gt_bytes = vtxk_band_gtables_bytes(n_loci, max_hap_len, p.tasks_per_locus, &p.gt_loci);
    if (p.gt_bytes && p.gt_loci < n_loci) {
        // chunks of tasks spanning about half the loci the buffer holds (loci differ in depth; a chunk whose loci do not
        // fit after all takes the LDS-table kernel)
        const uint64_t t = std::max<uint64_t>(4096, (uint64_t)p.gt_loci * std::max(p.tasks_per_locus, 1u) / 2);
        p.chunk = (uint32_t)std::min<uint64_t>(p.chunk, t & ~63ull);
        p.hard_cap = std::min(p.hard_cap, p.chunk); p.pend_cap = std::min(p.pend_cap, p.chunk); p.slots = std::min(p.slots, p.chunk);
    }
    return p;
}
// (hipMalloc of these GBs costs ~0.1 s the first time a context runs.  Doing it on a helper thread during the upload of
// vtx_submit was tried: hipMalloc stalls the copy workers, the submit got slower by more than the run got faster.)
static int band_reserve(vtx_ctx* c, BandPlan& p, bool quiet) {
#define RES(buf, bytes)                                                                             \
    do {                                                                                            \
        const hipError_t e_ = c->buf.reserve(bytes);                                                \
        if (e_ != hipSuccess) {                                                                     \
            if (quiet) { (void)hipGetLastError(); return VTX_E_HIP; }                               \
            return fail(c, e_ == hipErrorOutOfMemory ? VTX_E_NOMEM : VTX_E_HIP, "banded stage buffers: %s", hipGetErrorString(e_)); \
        }                                                                                           \
    } while (0)
    RES(d_band_ws, (size_t)vtxk_band_run_lanes() * vtxk_band_task_words() * sizeof(uint32_t));       // per resident lane
    if (p.gt_bytes && c->d_gtables.reserve(p.gt_bytes) != hipSuccess) {     // no room for them: the LDS-table kernels need none
        (void)hipGetLastError();
        p.gt_bytes = 0;
    }
    RES(d_pend, (size_t)p.pend_cap * sizeof(uint32_t));
    RES(d_pend_buf, (size_t)p.pend_cap * vtxk_band_pend_words() * sizeof(uint32_t));
    RES(d_poly, ((size_t)p.hard_cap + p.pend_cap) * p.poly_stride * sizeof(uint16_t));
    RES(d_band, (size_t)p.slots * 2 * p.band_stride * sizeof(uint16_t));
    RES(d_hard, ((size_t)p.hard_cap + p.pend_cap) * sizeof(uint32_t));
    RES(d_over, 4 * (size_t)p.n_tasks * sizeof(uint32_t));   // [0, 2n): band_run_kernel's overflows and their second-chance appends; [2n, 4n): what band_sweep_kernel declines (its own region: the appends of an overflowing chunk cannot reach it)
    RES(d_cnt, 64 * sizeof(uint32_t));
    if (p.gt_bytes) RES(d_fail, 2 * (size_t)p.chunk * sizeof(uint32_t));  // tasks band_diag_kernel leaves to band_run_kernel (as listed, then sorted)
    if (p.gt_bytes) RES(d_refine, (size_t)band_refine_cap(p.chunk) * vtxk_band_refine_words() * sizeof(uint32_t));   // records for band_refine_kernel
    if (p.gt_bytes) RES(d_tight, (size_t)p.chunk * sizeof(uint32_t));       // tasks with a certificate but no verdict ...
    if (p.gt_bytes) RES(d_tight_pack, (size_t)p.chunk * sizeof(uint32_t));  // ... and their bands (one diagonal stretch each: one word)
    if (p.gt_bytes) RES(d_dense, 2 * (size_t)p.chunk * sizeof(uint32_t));   // tasks for band_sweep_kernel (repeats: as listed, then sorted)
    if (p.gt_bytes) RES(d_sweep_log, vtxk_band_sweep_log_bytes());
    if (p.gt_bytes) RES(d_tight2, (size_t)p.chunk * sizeof(uint32_t));      // second stage: one-diagonal bands ...
    if (p.gt_bytes) RES(d_tight2_pack, (size_t)p.chunk * sizeof(uint32_t)); // ... one word each
    if (p.gt_bytes) RES(d_recheck2, (size_t)p.chunk * sizeof(uint32_t));
    if (p.gt_bytes) RES(d_recheck2_pack, (size_t)p.chunk * sizeof(uint32_t));
#undef RES
    return VTX_OK;
}

int vtx_submit(vtx_ctx* c, const vtx_batch* b) {
    if (!c) return VTX_E_INVAL;
    if (!b) return fail(c, VTX_E_INVAL, "vtx_submit: null batch");
    c->submitted = false; c->ran = false;
    const uint32_t nl = b->n_loci, nr = b->n_records;
    if ((nl && !b->loci) || (nr && !b->records) || (b->hap_bytes && !b->hap_arena) || (b->read_bytes && !b->read_arena))
        return fail(c, VTX_E_INVAL, "vtx_submit: null array with non-zero count");
    if (b->hap_bytes > 0xffffffffull || b->read_bytes > 0xffffffffull)
        return fail(c, VTX_E_UNSUPPORTED, "vtx_submit: arenas above 4 GiB need more than one batch");

    // ---- loci: validated on the host (O(loci)); everything per record happens on the device ----
    uint32_t next_rec = 0, max_hap = 0, max_hap_all = 0;
    for (uint32_t l = 0; l < nl; ++l) {
        const vtx_locus& L = b->loci[l];
        if (L.rec_begin != next_rec) return fail(c, VTX_E_INVAL, "vtx_submit: locus %u: records not contiguous (rec_begin %u, expected %u)", l, L.rec_begin, next_rec);
        if ((uint64_t)L.rec_begin + L.rec_count > nr) return fail(c, VTX_E_INVAL, "vtx_submit: locus %u: record range exceeds n_records", l);
        if ((uint64_t)L.ref_off + L.ref_len > b->hap_bytes || (uint64_t)L.alt_off + L.alt_len > b->hap_bytes)
            return fail(c, VTX_E_INVAL, "vtx_submit: locus %u: haplotype outside hap_arena", l);
        if (L.ref_len > kMaxHapLen || L.alt_len > kMaxHapLen)
            return fail(c, VTX_E_UNSUPPORTED, "vtx_submit: locus %u: haplotype longer than %u", l, kMaxHapLen);
        const uint32_t hl = std::max(L.ref_len, L.alt_len);
        max_hap_all = std::max(max_hap_all, hl);
        if (hl <= kFastHapLen) max_hap = std::max(max_hap, hl);      // LDS tables are sized for the loci the fast kernels take
        next_rec = L.rec_begin + L.rec_count;
    }
    if (next_rec != nr) return fail(c, VTX_E_INVAL, "vtx_submit: %u records not covered by any locus", nr - next_rec);
    note_long_loci(c, b->loci, nl);

    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const size_t u32 = sizeof(uint32_t), u64 = sizeof(uint64_t);
    HIP_TRY(c, c->d_loci.reserve((size_t)nl * sizeof(vtx_locus)));
    HIP_TRY(c, c->d_hap.reserve(b->hap_bytes + 16));
    HIP_TRY(c, c->d_read.reserve(b->read_bytes + 16));
    const bool nibbles = c->read_format == VTX_READS_NIBBLES;
    if (nibbles && (b->read_bytes & 1)) return fail(c, VTX_E_INVAL, "vtx_submit: VTX_READS_NIBBLES needs an even read_bytes");
    if (nibbles) HIP_TRY(c, c->d_read_packed.reserve(b->read_bytes / 2 + 16));
    if (int rc = reserve_record_buffers(c, nr)) return rc;
    HIP_TRY(c, c->d_seq.reserve(nr * u32));
    HIP_TRY(c, c->d_shape.reserve(nr + 16));
    HIP_TRY(c, c->d_shape2.reserve(nr + 16));
    HIP_TRY(c, c->d_prep_cnt.reserve(8 * u64 + 64 * u32));
    const size_t sort_tmp = vtxk_prep_sort_temp_bytes(nr);
    HIP_TRY(c, c->d_sort_tmp.reserve(std::max(sort_tmp, vtxk_scan_temp_bytes(std::max(nr, nl)))));
    unsigned long long* d_counters = c->d_prep_cnt.as<unsigned long long>();
    uint32_t* d_shape_cnt = (uint32_t*)(d_counters + 8);
    uint32_t caps[kNumShapes];
    for (int i = 0; i < kNumShapes; ++i) caps[i] = (uint32_t)(kShapes[i][0] * kShapes[i][1]);
    HIP_TRY(c, vtxk_prep_set_shapes(caps, kNumShapes, kFastReadLen, kFastHapLen));

    // ---- feed: descriptors first, then the record checks run on the device while the read bases still stream in ----
    hipStream_t s = c->stream;
    if (int rc = upload(c, {{c->d_loci.p, b->loci, (size_t)nl * sizeof(vtx_locus)},
                            {c->d_records.p, b->records, (size_t)nr * sizeof(vtx_record)},
                            {c->d_hap.p, b->hap_arena, (size_t)b->hap_bytes}})) return rc;
    unsigned long long cnt[8] = {0};
    uint32_t shape_cnt[16] = {0};
    HIP_TRY(c, hipMemsetAsync(c->d_prep_cnt.p, 0, 8 * u64 + 64 * u32, s));
    HIP_TRY(c, hipMemsetAsync(d_counters + 6, 0xff, u64, s));                       // "no offending record"
    if (nr) {
        HIP_TRY(c, vtxk_prep_rec_locus(c->d_loci.as<vtx_locus>(), nl, c->d_rec_locus.as<uint32_t>(), s));
        HIP_TRY(c, vtxk_prep_check(c->d_records.as<vtx_record>(), nr, c->d_rec_locus.as<uint32_t>(), c->d_loci.as<vtx_locus>(),
                                   b->read_bytes, kMaxReadLen | (nibbles ? 0x80000000u : 0u), c->cfg.n_barcodes, kNumShapes, c->d_shape.as<uint8_t>(),
                                   c->d_seq.as<uint32_t>(), d_shape_cnt, d_counters, s));
        // work lists per kernel shape: stable sort of the record numbers by shape
        HIP_TRY(c, vtxk_prep_sort_u8(c->d_shape.as<uint8_t>(), c->d_shape2.as<uint8_t>(), c->d_seq.as<uint32_t>(),
                                     c->d_work.as<uint32_t>(), nr, c->d_sort_tmp.p, sort_tmp, s));
    }
    HIP_TRY(c, hipMemcpyAsync(cnt, d_counters, 8 * u64, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(shape_cnt, d_shape_cnt, sizeof shape_cnt, hipMemcpyDeviceToHost, s));
    if (nibbles) {
        if (int rc = upload(c, {{c->d_read_packed.p, b->read_arena, (size_t)(b->read_bytes / 2)}})) return rc;
        HIP_TRY(c, vtxk_unpack_nibbles(c->d_read_packed.as<uint8_t>(), b->read_bytes / 2, c->d_read.as<uint8_t>(), s));
    } else if (int rc = upload(c, {{c->d_read.p, b->read_arena, (size_t)b->read_bytes}})) return rc;     // overlaps with the kernels above
    HIP_TRY(c, hipStreamSynchronize(s));
    if (cnt[6] != ~0ull) {
        const uint32_t r = (uint32_t)(cnt[6] >> 3), code = (uint32_t)(cnt[6] & 7);
        const vtx_record& R = b->records[r];
        if (code == 1) return fail(c, VTX_E_INVAL, "vtx_submit: record %u: read outside read_arena", r);
        if (code == 2) return fail(c, VTX_E_UNSUPPORTED, "vtx_submit: record %u: read length %u above %u", r, R.read_len, kMaxReadLen);
        if (code == 3) return fail(c, VTX_E_INVAL, "vtx_submit: record %u: cell_index %u >= n_barcodes %u", r, R.cell_index, c->cfg.n_barcodes);
        if (code == 5) return fail(c, VTX_E_INVAL, "vtx_submit: record %u: read_off %u is odd (VTX_READS_NIBBLES: every read starts at an even base)", r, R.read_off);
        return fail(c, VTX_E_INVAL, "vtx_submit: record %u: not sorted by (cell_index, umi_id) within its locus", r);
    }
    if (int rc = make_buckets(c, shape_cnt, max_hap, d_shape_cnt + 16)) return rc;
    if (int rc = build_groups(c, nr)) return rc;
    HIP_TRY(c, hipStreamSynchronize(s));
    c->n_loci = nl; c->n_records = nr; c->max_hap_len = max_hap; c->max_read_len = (uint32_t)cnt[5]; c->cells = cnt[3];
    c->max_hap_all = max_hap_all; c->max_read_all = std::max<uint32_t>((uint32_t)cnt[7], std::max<uint32_t>((uint32_t)cnt[5], 1u));
    c->submitted = true;
    // the banded stage's buffers (a few GB for a config-3 batch) now, not inside the first vtx_run of the context: a drop-in CLI
    // calls vtx_run ONCE per batch, and 0.1 s of hipMalloc inside it was two thirds of that call.  Best effort: vtx_run reserves
    // again (a no-op when this succeeded) and reports a failure there.
    if (c->cfg.aligner == VTX_ALIGNER_BANDED && c->n_records) {
        BandPlan bp = band_plan(c->n_records, c->n_loci, c->max_hap_len);
        (void)band_reserve(c, bp, true);
    }
    return VTX_OK;
}

int vtx_set_barcodes(vtx_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint32_t n) {
    if (!c) return VTX_E_INVAL;
    if (!offsets || (n && !bytes && offsets[n] > offsets[0])) return fail(c, VTX_E_INVAL, "vtx_set_barcodes: null argument");
    if (c->cfg.n_barcodes == 0) c->cfg.n_barcodes = n;          // a context created before the list was read (cfg.n_barcodes 0) takes its width from it
    if (n != c->cfg.n_barcodes) return fail(c, VTX_E_INVAL, "vtx_set_barcodes: %u barcodes, cfg.n_barcodes is %u", n, c->cfg.n_barcodes);
    c->bc_ready = false;
    for (uint32_t j = 0; j < n; ++j) {
        if (offsets[j + 1] < offsets[j]) return fail(c, VTX_E_INVAL, "vtx_set_barcodes: offsets not ascending at %u", j);
        if (offsets[j + 1] - offsets[j] >= VTX_TAG_MISSING) return fail(c, VTX_E_UNSUPPORTED, "vtx_set_barcodes: barcode %u longer than 65534 bytes", j);
    }
    // open addressing, load <= 1/2; slot = index + 1, 0 = empty.  Duplicated byte strings keep the first index (:704-710).
    uint32_t cap = 16;
    while (cap < 2ull * n) cap <<= 1;
    std::vector<uint32_t> slots(cap, 0);
    std::vector<uint64_t> hashes(std::max(n, 1u));
    for (uint32_t j = 0; j < n; ++j) {
        const uint8_t* b = bytes + offsets[j];
        const uint32_t len = (uint32_t)(offsets[j + 1] - offsets[j]);
        const uint64_t h = vtx_hash_bytes(b, len, 0);
        hashes[j] = h;
        bool dup = false;
        uint32_t sidx = (uint32_t)h & (cap - 1);
        for (; slots[sidx]; sidx = (sidx + 1) & (cap - 1)) {
            const uint32_t k = slots[sidx] - 1;
            if (hashes[k] == h && offsets[k + 1] - offsets[k] == len && memcmp(bytes + offsets[k], b, len) == 0) { dup = true; break; }
        }
        if (!dup) slots[sidx] = j + 1;
    }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const uint64_t nbytes = n ? offsets[n] : 0;
    HIP_TRY(c, c->d_bc_slots.reserve(cap * sizeof(uint32_t)));
    HIP_TRY(c, c->d_bc_hash.reserve(hashes.size() * sizeof(uint64_t)));
    HIP_TRY(c, c->d_bc_off.reserve(((size_t)n + 1) * sizeof(uint64_t)));
    HIP_TRY(c, c->d_bc_bytes.reserve(nbytes + 16));
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->d_bc_slots.p, slots.data(), cap * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_bc_hash.p, hashes.data(), hashes.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_bc_off.p, offsets, ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    if (nbytes) HIP_TRY(c, hipMemcpyAsync(c->d_bc_bytes.p, bytes, nbytes, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->bc_mask = cap - 1;
    c->bc_ready = true;
    return VTX_OK;
}

// ---- raw batches: what vtx_submit_raw and vtx_submit_bam share ----
// device buffers of a raw batch of nr records over nl loci (the arenas: d_read / d_read_packed, d_tags; the records: d_raw, d_raw_locus)
static int raw_reserve(vtx_ctx* c, uint32_t nl, uint32_t nr, uint64_t hap_bytes, uint64_t read_bytes, uint64_t tag_bytes, bool nibbles) {
    const size_t u32 = sizeof(uint32_t), u64 = sizeof(uint64_t);
    HIP_TRY(c, c->d_loci.reserve((size_t)nl * sizeof(vtx_locus)));
    HIP_TRY(c, c->d_hap.reserve(hap_bytes + 16));
    HIP_TRY(c, c->d_read.reserve(read_bytes + 16));
    if (nibbles) HIP_TRY(c, c->d_read_packed.reserve(read_bytes / 2 + 16));
    HIP_TRY(c, c->d_tags.reserve(tag_bytes + 16));
    HIP_TRY(c, c->d_raw.reserve((size_t)nr * sizeof(vtx_raw_record)));
    if (int rc = reserve_record_buffers(c, nr)) return rc;
    DevBuf* k64[] = {&c->d_key_lc, &c->d_key_lc2, &c->d_key_umi, &c->d_key_umi2};
    for (DevBuf* d : k64) HIP_TRY(c, d->reserve(nr * u64));
    DevBuf* k32[] = {&c->d_raw_locus, &c->d_idx, &c->d_idx2, &c->d_seq};
    for (DevBuf* d : k32) HIP_TRY(c, d->reserve(nr * u32));
    HIP_TRY(c, c->d_shape.reserve(nr + 16));
    HIP_TRY(c, c->d_shape2.reserve(nr + 16));
    HIP_TRY(c, c->d_locus_cnt.reserve(((size_t)nl + 1) * u32));
    HIP_TRY(c, c->d_locus_scan.reserve(((size_t)nl + 1) * u32));
    HIP_TRY(c, c->d_prep_cnt.reserve(8 * u64 + 64 * u32));
    const size_t sort_tmp = vtxk_prep_sort_temp_bytes(nr);
    HIP_TRY(c, c->d_sort_tmp.reserve(std::max(sort_tmp, vtxk_scan_temp_bytes(std::max(nr, nl)))));
    uint32_t caps[kNumShapes];
    for (int i = 0; i < kNumShapes; ++i) caps[i] = (uint32_t)(kShapes[i][0] * kShapes[i][1]);
    HIP_TRY(c, vtxk_prep_set_shapes(caps, kNumShapes, kFastReadLen, kFastHapLen));
    return VTX_OK;
}

// Barcode lookup, UB test, UMI grouping and the sort by (locus, cell, UMI) of the nr raw records resident in d_raw / d_tags (their
// bases in d_read), then the work lists and the group structure: the state vtx_submit leaves.  locus_from_loci: d_raw_locus is derived
// from the loci's record ranges (vtx_submit_raw: records grouped by locus); otherwise the caller filled it (vtx_submit_bam: records in
// BAM order, a locus per record — the sort's key does not care).
static int raw_prepare(vtx_ctx* c, uint32_t nl, uint32_t nr, uint64_t read_bytes, uint64_t tag_bytes, bool nibbles, uint32_t max_hap,
                       uint32_t max_hap_all, bool locus_from_loci, vtx_raw_stats* stats) {
    hipStream_t s = c->stream;
    const size_t u32 = sizeof(uint32_t), u64 = sizeof(uint64_t);
    const size_t sort_tmp = vtxk_prep_sort_temp_bytes(nr);
    unsigned long long* d_counters = c->d_prep_cnt.as<unsigned long long>();
    uint32_t* d_shape_cnt = (uint32_t*)(d_counters + 8);
    uint32_t* d_lut_flag = d_shape_cnt + 16;
    HIP_TRY(c, hipEventRecord(c->ev[0], s));
    const uint32_t cell_bits = bits_for(c->cfg.n_barcodes ? c->cfg.n_barcodes - 1 : 0);
    const int end_bit = (int)(cell_bits + bits_for(nl));
    if (end_bit > 64) return fail(c, VTX_E_UNSUPPORTED, "raw batch: loci x barcodes exceed a 64-bit sort key");
    const int use_umi = c->cfg.use_umi ? 1 : 0;
    unsigned long long cnt[8] = {0};
    uint32_t n_kept = 0, rounds = 0;
    // test hook: the first N rounds hash every UMI to 0, so that the collision check and the re-seed are exercised
    const uint32_t weak_rounds = VTX_DEV_ENV("VTX_PREP_WEAK_ROUNDS") ? (uint32_t)atoi(VTX_DEV_ENV("VTX_PREP_WEAK_ROUNDS")) : 0;
    if (nr && locus_from_loci) HIP_TRY(c, vtxk_prep_rec_locus(c->d_loci.as<vtx_locus>(), nl, c->d_raw_locus.as<uint32_t>(), s));
    for (uint64_t seed = 0x9e3779b97f4a7c15ull;; seed = seed * 0xd1342543de82ef95ull + 1) {
        ++rounds;
        HIP_TRY(c, hipMemsetAsync(c->d_prep_cnt.p, 0, 8 * u64 + 64 * u32, s));
        HIP_TRY(c, hipMemsetAsync(c->d_locus_cnt.p, 0, ((size_t)nl + 1) * u32, s));      // first record of each locus
        HIP_TRY(c, hipMemsetAsync(c->d_locus_scan.p, 0, ((size_t)nl + 1) * u32, s));     // one past its last record
        HIP_TRY(c, vtxk_prep_resolve(c->d_raw.as<vtx_raw_record>(), nr, c->d_raw_locus.as<uint32_t>(), c->d_tags.as<uint8_t>(),
                                     tag_bytes, read_bytes, kMaxReadLen | (nibbles ? 0x80000000u : 0u), c->d_bc_slots.as<uint32_t>(), c->bc_mask,
                                     c->d_bc_hash.as<uint64_t>(), c->d_bc_off.as<uint64_t>(), c->d_bc_bytes.as<uint8_t>(),
                                     use_umi, seed, rounds <= weak_rounds ? 0ull : ~0ull, cell_bits, nl, c->d_key_lc.as<uint64_t>(), c->d_key_umi.as<uint64_t>(),
                                     c->d_idx.as<uint32_t>(), d_counters, s));
        HIP_TRY(c, hipMemcpyAsync(cnt, d_counters, 3 * u64, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        if (cnt[2]) return fail(c, VTX_E_INVAL, "raw batch: a record points outside its arena, is longer than %u bases, or (VTX_READS_NIBBLES) starts at an odd base", kMaxReadLen);
        n_kept = nr - (uint32_t)cnt[0] - (uint32_t)cnt[1];
        // stable LSD order: UMI hash first, then (locus, cell); dropped records carry the largest key and end up last
        const uint32_t* perm = nullptr;
        const uint64_t* key_sorted = nullptr;
        if (use_umi) {
            HIP_TRY(c, vtxk_prep_sort_u64(c->d_key_umi.as<uint64_t>(), c->d_key_umi2.as<uint64_t>(), c->d_idx.as<uint32_t>(),
                                          c->d_idx2.as<uint32_t>(), nr, 64, c->d_sort_tmp.p, sort_tmp, s));
            HIP_TRY(c, vtxk_prep_gather_u64(c->d_key_lc.as<uint64_t>(), c->d_idx2.as<uint32_t>(), nr, c->d_key_lc2.as<uint64_t>(), s));
            HIP_TRY(c, vtxk_prep_sort_u64(c->d_key_lc2.as<uint64_t>(), c->d_key_umi2.as<uint64_t>(), c->d_idx2.as<uint32_t>(),
                                          c->d_idx.as<uint32_t>(), nr, end_bit, c->d_sort_tmp.p, sort_tmp, s));
            perm = c->d_idx.as<uint32_t>(); key_sorted = c->d_key_umi2.as<uint64_t>();
        } else {
            HIP_TRY(c, vtxk_prep_sort_u64(c->d_key_lc.as<uint64_t>(), c->d_key_lc2.as<uint64_t>(), c->d_idx.as<uint32_t>(),
                                          c->d_idx2.as<uint32_t>(), nr, end_bit, c->d_sort_tmp.p, sort_tmp, s));
            perm = c->d_idx2.as<uint32_t>(); key_sorted = c->d_key_lc2.as<uint64_t>();
        }
        HIP_TRY(c, vtxk_prep_finalize(n_kept, perm, key_sorted, c->d_key_umi.as<uint64_t>(), c->d_raw.as<vtx_raw_record>(),
                                      c->d_tags.as<uint8_t>(), c->d_loci.as<vtx_locus>(), cell_bits, use_umi, kNumShapes,
                                      c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(), c->d_head_umi.as<uint32_t>(),
                                      c->d_shape.as<uint8_t>(), c->d_seq.as<uint32_t>(), c->d_locus_cnt.as<uint32_t>(),
                                      c->d_locus_scan.as<uint32_t>(), d_shape_cnt, d_counters, s));
        HIP_TRY(c, hipMemcpyAsync(cnt, d_counters, 8 * u64, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        if (!cnt[4]) break;                  // no UMI hash collision inside a (locus, cell) group
        if (rounds == 8) return fail(c, VTX_E_STATE, "raw batch: UMI hash collisions with 8 different seeds");
    }
    const size_t scan_tmp = vtxk_scan_temp_bytes(std::max(nr, nl));
    // dense UMI group numbers; per-locus record ranges of the kept, sorted records
    if (n_kept) {
        HIP_TRY(c, vtxk_inclusive_scan_u32(c->d_head_umi.as<uint32_t>(), c->d_umi_scan.as<uint32_t>(), n_kept, c->d_sort_tmp.p, scan_tmp, s));
        HIP_TRY(c, vtxk_prep_umi_ids(c->d_records.as<vtx_record>(), c->d_umi_scan.as<uint32_t>(), n_kept, s));
    }
    if (nl) {
        HIP_TRY(c, vtxk_prep_locus_counts(c->d_locus_cnt.as<uint32_t>(), c->d_locus_scan.as<uint32_t>(), nl, s));
        HIP_TRY(c, vtxk_inclusive_scan_u32(c->d_locus_cnt.as<uint32_t>(), c->d_locus_scan.as<uint32_t>(), nl, c->d_sort_tmp.p, scan_tmp, s));
        HIP_TRY(c, vtxk_prep_locus_ranges(c->d_loci.as<vtx_locus>(), c->d_locus_cnt.as<uint32_t>(), c->d_locus_scan.as<uint32_t>(), nl, s));
    }
    // work lists per kernel shape: stable sort of the record numbers by shape
    HIP_TRY(c, vtxk_prep_sort_u8(c->d_shape.as<uint8_t>(), c->d_shape2.as<uint8_t>(), c->d_seq.as<uint32_t>(), c->d_work.as<uint32_t>(),
                                 n_kept, c->d_sort_tmp.p, sort_tmp, s));
    uint32_t shape_cnt[16] = {0};
    HIP_TRY(c, hipMemcpyAsync(shape_cnt, d_shape_cnt, sizeof shape_cnt, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (int rc = make_buckets(c, shape_cnt, max_hap, d_lut_flag)) return rc;
    if (int rc = build_groups(c, n_kept)) return rc;
    HIP_TRY(c, hipEventRecord(c->ev[1], s));
    HIP_TRY(c, hipStreamSynchronize(s));
    float ms = 0;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->n_loci = nl; c->n_records = n_kept; c->max_hap_len = max_hap; c->max_read_len = (uint32_t)cnt[5]; c->cells = cnt[3];
    c->max_hap_all = max_hap_all; c->max_read_all = std::max<uint32_t>((uint32_t)cnt[7], std::max<uint32_t>((uint32_t)cnt[5], 1u));
    c->submitted = true;
    // the banded stage's buffers (a few GB for a config-3 batch) now, not inside the first vtx_run of the context: a drop-in CLI
    // calls vtx_run ONCE per batch, and 0.1 s of hipMalloc inside it was two thirds of that call.  Best effort: vtx_run reserves
    // again (a no-op when this succeeded) and reports a failure there.
    if (c->cfg.aligner == VTX_ALIGNER_BANDED && c->n_records) {
        BandPlan bp = band_plan(c->n_records, c->n_loci, c->max_hap_len);
        (void)band_reserve(c, bp, true);
    }
    if (stats) {
        stats->num_not_cell_bc = cnt[0]; stats->num_non_umi = cnt[1]; stats->kept = n_kept; stats->prep_ms = ms;
        stats->hash_rounds = rounds;
    }
    return VTX_OK;
}

// loci of a raw batch, host side: haplotypes inside the arena and within the limits; *max_hap: the longest one the fast kernels take
static int check_loci_haps(vtx_ctx* c, const char* who, const vtx_locus* loci, uint32_t nl, uint64_t hap_bytes, uint32_t* max_hap, uint32_t* max_hap_all) {
    *max_hap = *max_hap_all = 0;
    for (uint32_t l = 0; l < nl; ++l) {
        const vtx_locus& L = loci[l];
        if ((uint64_t)L.ref_off + L.ref_len > hap_bytes || (uint64_t)L.alt_off + L.alt_len > hap_bytes)
            return fail(c, VTX_E_INVAL, "%s: locus %u: haplotype outside hap_arena", who, l);
        if (L.ref_len > kMaxHapLen || L.alt_len > kMaxHapLen)
            return fail(c, VTX_E_UNSUPPORTED, "%s: locus %u: haplotype longer than %u", who, l, kMaxHapLen);
        const uint32_t hl = std::max(L.ref_len, L.alt_len);
        *max_hap_all = std::max(*max_hap_all, hl);
        if (hl <= kFastHapLen) *max_hap = std::max(*max_hap, hl);
    }
    return VTX_OK;
}

int vtx_submit_raw(vtx_ctx* c, const vtx_raw_batch* b, vtx_raw_stats* stats) {
    if (!c) return VTX_E_INVAL;
    if (!b) return fail(c, VTX_E_INVAL, "vtx_submit_raw: null batch");
    c->submitted = false; c->ran = false;
    if (!c->bc_ready) return fail(c, VTX_E_STATE, "vtx_submit_raw: no barcode list (vtx_set_barcodes)");
    const uint32_t nl = b->n_loci, nr = b->n_records;
    if ((nl && !b->loci) || (nr && !b->records) || (b->hap_bytes && !b->hap_arena) || (b->read_bytes && !b->read_arena) ||
        (b->tag_bytes && !b->tag_arena))
        return fail(c, VTX_E_INVAL, "vtx_submit_raw: null array with non-zero count");
    if (b->hap_bytes > 0xffffffffull || b->read_bytes > 0xffffffffull || b->tag_bytes > 0xffffffffull)
        return fail(c, VTX_E_UNSUPPORTED, "vtx_submit_raw: arenas above 4 GiB need more than one batch");
    // loci: host validation is O(loci); everything per record happens on the device
    uint32_t next_rec = 0, max_hap = 0, max_hap_all = 0;
    for (uint32_t l = 0; l < nl; ++l) {
        const vtx_locus& L = b->loci[l];
        if (L.rec_begin != next_rec) return fail(c, VTX_E_INVAL, "vtx_submit_raw: locus %u: records not contiguous (rec_begin %u, expected %u)", l, L.rec_begin, next_rec);
        if ((uint64_t)L.rec_begin + L.rec_count > nr) return fail(c, VTX_E_INVAL, "vtx_submit_raw: locus %u: record range exceeds n_records", l);
        next_rec = L.rec_begin + L.rec_count;
    }
    if (next_rec != nr) return fail(c, VTX_E_INVAL, "vtx_submit_raw: %u records not covered by any locus", nr - next_rec);
    if (int rc = check_loci_haps(c, "vtx_submit_raw", b->loci, nl, b->hap_bytes, &max_hap, &max_hap_all)) return rc;
    note_long_loci(c, b->loci, nl);

    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    const bool nibbles = c->read_format == VTX_READS_NIBBLES;
    if (nibbles && (b->read_bytes & 1)) return fail(c, VTX_E_INVAL, "vtx_submit_raw: VTX_READS_NIBBLES needs an even read_bytes");
    if (int rc = raw_reserve(c, nl, nr, b->hap_bytes, b->read_bytes, b->tag_bytes, nibbles)) return rc;
    if (int rc = upload(c, {{c->d_loci.p, b->loci, (size_t)nl * sizeof(vtx_locus)},
                            {c->d_raw.p, b->records, (size_t)nr * sizeof(vtx_raw_record)},
                            {c->d_hap.p, b->hap_arena, (size_t)b->hap_bytes},
                            {c->d_tags.p, b->tag_arena, (size_t)b->tag_bytes},
                            {nibbles ? c->d_read_packed.p : c->d_read.p, b->read_arena, (size_t)(nibbles ? b->read_bytes / 2 : b->read_bytes)}})) return rc;
    if (nibbles) HIP_TRY(c, vtxk_unpack_nibbles(c->d_read_packed.as<uint8_t>(), b->read_bytes / 2, c->d_read.as<uint8_t>(), s));
    return raw_prepare(c, nl, nr, b->read_bytes, b->tag_bytes, nibbles, max_hap, max_hap_all, true, stats);
}

// ---- vtx_prefetch_file: the BAM's bytes start travelling before anybody knows which of them matter ----
int vtx_prefetch_file(vtx_ctx* c, const char* path, uint64_t file_off, uint64_t n) {
    if (!c) return VTX_E_INVAL;
    if (!path) return fail(c, VTX_E_INVAL, "vtx_prefetch_file: null argument");
    if (c->pf_thread.joinable()) c->pf_thread.join();
    c->pf_valid = false;
    if (c->pf_map) { munmap(c->pf_map, c->pf_map_bytes); c->pf_map = nullptr; }
    const int fd = open(path, O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) { if (fd >= 0) close(fd); return fail(c, VTX_E_INVAL, "vtx_prefetch_file: cannot open %s", path); }
    if (file_off > (uint64_t)st.st_size) file_off = (uint64_t)st.st_size;
    if (n == 0 || file_off + n > (uint64_t)st.st_size) n = (uint64_t)st.st_size - file_off;
    if (!n) { close(fd); return VTX_OK; }
    if (hipSetDevice(c->cfg.device) != hipSuccess || c->d_bam_comp.reserve((size_t)n + 64) != hipSuccess) { close(fd); return fail(c, VTX_E_NOMEM, "vtx_prefetch_file: no device memory for %llu bytes", (unsigned long long)n); }
    // the copy workers read the file through a mapping of their own (measured against pread() into the pinned buffers at config-3 scale:
    // the same 80 - 200 ms beside the host's planning threads, and the mapping showed no multi-second outliers)
    const uint64_t map_off = file_off & ~(uint64_t)4095;
    void* mp = mmap(nullptr, (size_t)(file_off - map_off + n), PROT_READ, MAP_PRIVATE, fd, (off_t)map_off);
    close(fd);
    if (mp == MAP_FAILED) return fail(c, VTX_E_INVAL, "vtx_prefetch_file: cannot map %s", path);
    c->pf_map = mp; c->pf_map_bytes = (size_t)(file_off - map_off + n);
    c->pf_off = file_off; c->pf_n = n; c->pf_rc = VTX_OK; c->pf_valid = true;
    void* dst = c->d_bam_comp.p;
    const uint8_t* src = (const uint8_t*)mp + (file_off - map_off);
    c->pf_thread = std::thread([c, dst, src, n] {
        if (hipSetDevice(c->cfg.device) != hipSuccess) { c->pf_rc = VTX_E_HIP; return; }
        const auto t0 = std::chrono::steady_clock::now();
        c->pf_rc = upload(c, {{dst, src, (size_t)n}});
        c->pf_ms = (float)(1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    });
    return VTX_OK;
}

// ---- vtx_submit_bam: the ingest itself on the device (vtx_ingest.hip) ----
int vtx_submit_bam(vtx_ctx* c, const vtx_bam_ingest* g, vtx_ingest_stats* st) {
    if (!c) return VTX_E_INVAL;
    if (!g) return fail(c, VTX_E_INVAL, "vtx_submit_bam: null argument");
    c->submitted = false; c->ran = false;
    if (st) memset(st, 0, sizeof *st);
    if (!c->bc_ready) return fail(c, VTX_E_STATE, "vtx_submit_bam: no barcode list (vtx_set_barcodes)");
    const uint32_t nl = g->n_loci, nb = g->n_blocks;
    if ((nb && (!g->blocks || !g->file)) || (nl && !g->loci) || (g->hap_bytes && !g->hap_arena) || (g->n_seeds && !g->seeds) ||
        (g->n_intervals && !g->intervals) || !g->tid_begin || (g->n_ref && !g->tid_max_span))
        return fail(c, VTX_E_INVAL, "vtx_submit_bam: null array with non-zero count");
    if (g->hap_bytes > 0xffffffffull) return fail(c, VTX_E_UNSUPPORTED, "vtx_submit_bam: hap arena above 4 GiB");
    uint32_t max_hap = 0, max_hap_all = 0;
    if (int rc = check_loci_haps(c, "vtx_submit_bam", g->loci, nl, g->hap_bytes, &max_hap, &max_hap_all)) return rc;
    note_long_loci(c, g->loci, nl);
    if (g->tid_begin[0] != 0 || g->tid_begin[g->n_ref] != g->n_intervals) return fail(c, VTX_E_INVAL, "vtx_submit_bam: tid_begin does not cover the intervals");
    for (uint32_t t = 0; t < g->n_ref; ++t) {
        if (g->tid_begin[t] > g->tid_begin[t + 1]) return fail(c, VTX_E_INVAL, "vtx_submit_bam: tid_begin not ascending at %u", t);
        for (uint32_t k = g->tid_begin[t]; k < g->tid_begin[t + 1]; ++k) {
            const vtx_bam_interval& I = g->intervals[k];
            if (I.locus >= nl || I.end < I.start || (k > g->tid_begin[t] && g->intervals[k - 1].start > I.start) || (int64_t)I.end - I.start > g->tid_max_span[t])
                return fail(c, VTX_E_INVAL, "vtx_submit_bam: interval %u: bad locus / order / span", k);
        }
    }
    // blocks: consecutive in the file; the compressed range [lo, hi) travels as it is
    std::vector<vtxg_block> blocks(nb);
    uint64_t lo = nb ? g->blocks[0].coff : 0, hi = lo, utotal = 0;
    for (uint32_t i = 0; i < nb; ++i) {
        const vtx_bgzf_block& B = g->blocks[i];
        if (B.coff < hi || B.coff + B.clen > g->file_bytes || B.isize > 65536u)
            return fail(c, VTX_E_INVAL, "vtx_submit_bam: block %u: out of order, outside the file or above 64 KiB", i);
        blocks[i] = vtxg_block{B.coff - lo, utotal, B.clen, B.isize};
        hi = B.coff + B.clen;
        utotal += B.isize;
    }
    const uint64_t end_upos = std::min<uint64_t>(g->end_upos, utotal);
    for (uint32_t i = 0; i < g->n_seeds; ++i)
        if (g->seeds[i] >= end_upos || (i && g->seeds[i] <= g->seeds[i - 1])) return fail(c, VTX_E_INVAL, "vtx_submit_bam: seed %u: not ascending or beyond the end", i);

    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    const size_t u32 = sizeof(uint32_t), u64 = sizeof(uint64_t);
    const uint32_t ns = g->n_seeds, ni = g->n_intervals, nref = g->n_ref;
    // bytes a vtx_prefetch_file already brought (or is still bringing) to the device?
    // (Tried in round 6: inflating the blocks of a 128 MB segment as soon as it has landed, on streams of their own.  Every launch of
    //  the latency-bound inflate kernel takes its full ~35 ms whatever the block count, and the copies still in flight queued behind
    //  the kernels: inflate 94 -> 250 ms, the copy 0.09 -> 1.4 s.  The copy is waited for as a whole.)
    const auto t_pf = std::chrono::steady_clock::now();
    if (c->pf_thread.joinable()) c->pf_thread.join();
    const float pf_wait_ms = (float)(1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pf).count());
    const bool prefetched = c->pf_valid && c->pf_rc == VTX_OK && nb && c->pf_off <= lo && hi <= c->pf_off + c->pf_n;
    if (prefetched) { const uint64_t shift = lo - c->pf_off; for (auto& B : blocks) B.coff += shift; }
    else { c->pf_valid = false; HIP_TRY(c, c->d_bam_comp.reserve((size_t)(hi - lo) + 64)); }
    HIP_TRY(c, c->d_bam_data.reserve((size_t)utotal + 64));
    HIP_TRY(c, c->d_bam_blocks.reserve((size_t)nb * sizeof(vtxg_block)));
    HIP_TRY(c, c->d_bam_seeds.reserve((size_t)ns * u64));
    HIP_TRY(c, c->d_bam_seed_cnt.reserve((size_t)ns * u32 + 16));
    HIP_TRY(c, c->d_bam_seed_scan.reserve((size_t)ns * u32 + 16));
    HIP_TRY(c, c->d_bam_iv.reserve((size_t)ni * 3 * u32 + ((size_t)nref + 1) * u32 + (size_t)nref * u32 + 64));
    HIP_TRY(c, c->d_bam_cnt.reserve(VTXG_N_COUNTERS * u64 + 4 * u32));
    HIP_TRY(c, c->d_scan_tmp.reserve(vtxk_scan_temp_bytes(std::max(ns, 1u))));
    // intervals as arrays: start[ni], end[ni], locus[ni], tid_begin[nref + 1], tid_span[nref]
    std::vector<uint32_t> ivh((size_t)ni * 3 + nref + 1 + nref);
    for (uint32_t k = 0; k < ni; ++k) { ivh[k] = (uint32_t)g->intervals[k].start; ivh[ni + k] = (uint32_t)g->intervals[k].end; ivh[2 * (size_t)ni + k] = g->intervals[k].locus; }
    for (uint32_t t = 0; t <= nref; ++t) ivh[3 * (size_t)ni + t] = g->tid_begin[t];
    for (uint32_t t = 0; t < nref; ++t) ivh[3 * (size_t)ni + nref + 1 + t] = (uint32_t)g->tid_max_span[t];
    const int32_t* d_iv_start = c->d_bam_iv.as<int32_t>();
    const int32_t* d_iv_end = d_iv_start + ni;
    const uint32_t* d_iv_locus = (const uint32_t*)(d_iv_end + ni);
    const uint32_t* d_tid_begin = d_iv_locus + ni;
    const int32_t* d_tid_span = (const int32_t*)(d_tid_begin + nref + 1);
    unsigned long long* d_counters = c->d_bam_cnt.as<unsigned long long>();
    uint32_t* d_err = (uint32_t*)(d_counters + VTXG_N_COUNTERS);
    HIP_TRY(c, c->d_hap.reserve(g->hap_bytes + 16));
    HIP_TRY(c, c->d_loci.reserve((size_t)nl * sizeof(vtx_locus)));

    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point t) { return (float)(1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count()); };
    HIP_TRY(c, hipMemsetAsync(d_counters, 0, VTXG_N_COUNTERS * u64 + 4 * u32, s));
    HIP_TRY(c, hipMemsetAsync(d_err + 1, 0xff, u32, s));
    if (int rc = upload(c, {{c->d_bam_comp.p, g->file + lo, prefetched ? (size_t)0 : (size_t)(hi - lo)},
                            {c->d_bam_blocks.p, blocks.data(), (size_t)nb * sizeof(vtxg_block)},
                            {c->d_bam_seeds.p, g->seeds, (size_t)ns * u64},
                            {c->d_bam_iv.p, ivh.data(), ivh.size() * u32},
                            {c->d_loci.p, g->loci, (size_t)nl * sizeof(vtx_locus)},
                            {c->d_hap.p, g->hap_arena, (size_t)g->hap_bytes}})) return rc;
    HIP_TRY(c, hipStreamSynchronize(s));
    const float h2d_ms = since(t0);
    // ---- inflate, record boundaries ----
    HIP_TRY(c, hipEventRecord(c->ev[0], s));
    HIP_TRY(c, vtxg_inflate(c->d_bam_comp.as<uint8_t>(), c->d_bam_blocks.as<vtxg_block>(), nb, c->d_bam_data.as<uint8_t>(), d_err, nullptr, 0, s));
    HIP_TRY(c, hipEventRecord(c->ev[1], s));
    uint32_t n_rec = 0;
    if (ns) {
        HIP_TRY(c, vtxg_chain(c->d_bam_data.as<uint8_t>(), utotal, c->d_bam_seeds.as<uint64_t>(), ns, end_upos, c->d_bam_seed_cnt.as<uint32_t>(), nullptr, nullptr, d_err, s));
        HIP_TRY(c, vtxk_inclusive_scan_u32(c->d_bam_seed_cnt.as<uint32_t>(), c->d_bam_seed_scan.as<uint32_t>(), ns, c->d_scan_tmp.p, vtxk_scan_temp_bytes(ns), s));
        HIP_TRY(c, hipMemcpyAsync(&n_rec, c->d_bam_seed_scan.as<uint32_t>() + (ns - 1), u32, hipMemcpyDeviceToHost, s));
    }
    uint32_t err[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(err, d_err, 2 * u32, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    auto ingest_error = [&](uint32_t e0, uint32_t e1) -> int {
        if (e0 & 0x1ffu) return fail(c, VTX_E_UNSUPPORTED, "vtx_submit_bam: BGZF block %u does not inflate on the device (status bits 0x%x): the host packer decides", e1, e0 & 0x1ffu);
        if (e0 & VTXG_ERR_CHAIN) return fail(c, VTX_E_UNSUPPORTED, "vtx_submit_bam: a record chain does not end on the index's next record start (index and file disagree, or a malformed record)");
        return fail(c, VTX_E_UNSUPPORTED, "vtx_submit_bam: malformed BAM record");
    };
    if (err[0]) return ingest_error(err[0], err[1]);
    if ((uint64_t)n_rec > 0xfffffff0ull) return fail(c, VTX_E_UNSUPPORTED, "vtx_submit_bam: more than 2^32 BAM records in one ingest");
    HIP_TRY(c, c->d_bam_rec.reserve((size_t)n_rec * u64 + 16));
    DevBuf* per_rec[] = {&c->d_bam_nhit, &c->d_bam_rsz, &c->d_bam_tsz, &c->d_bam_hscan, &c->d_bam_rscan, &c->d_bam_tscan};
    for (DevBuf* d : per_rec) HIP_TRY(c, d->reserve((size_t)n_rec * u32 + 16));
    HIP_TRY(c, c->d_bam_info.reserve((size_t)n_rec * sizeof(vtxg_recinfo) + 16));
    HIP_TRY(c, c->d_scan_tmp.reserve(vtxk_scan_temp_bytes(std::max(n_rec, 1u))));
    if (ns) HIP_TRY(c, vtxg_chain(c->d_bam_data.as<uint8_t>(), utotal, c->d_bam_seeds.as<uint64_t>(), ns, end_upos, c->d_bam_seed_cnt.as<uint32_t>(),
                                   c->d_bam_seed_scan.as<uint32_t>(), c->d_bam_rec.as<uint64_t>(), d_err, s));
    HIP_TRY(c, hipEventRecord(c->ev[2], s));
    // ---- fetch + filters per (read, locus) pair: counts, then offsets, then the raw records ----
    const vtxg_filter f{nref, g->min_mapq, g->primary_only ? 1u : 0u, g->no_duplicates ? 1u : 0u, (uint32_t)(uint8_t)g->bam_tag[0] | ((uint32_t)(uint8_t)g->bam_tag[1] << 8)};
    HIP_TRY(c, vtxg_scan(0, c->d_bam_data.as<uint8_t>(), c->d_bam_rec.as<uint64_t>(), n_rec, f, d_iv_start, d_iv_end, d_iv_locus, d_tid_begin, d_tid_span,
                         c->d_bam_nhit.as<uint32_t>(), c->d_bam_rsz.as<uint32_t>(), c->d_bam_tsz.as<uint32_t>(), c->d_bam_info.as<vtxg_recinfo>(),
                         nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d_counters, d_err, s));
    unsigned long long cnt[VTXG_N_COUNTERS] = {0};
    HIP_TRY(c, hipMemcpyAsync(cnt, d_counters, sizeof cnt, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(err, d_err, 2 * u32, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (err[0]) return ingest_error(err[0], err[1]);
    const uint64_t read_bases = cnt[6], tag_bytes = cnt[7], n_pairs = cnt[8];
    if (read_bases > 0xF0000000ull || tag_bytes > 0xF0000000ull || n_pairs > 0x7fffffffull)
        return fail(c, VTX_E_UNSUPPORTED, "vtx_submit_bam: the reads of this range need more than one batch (%llu bases, %llu tag bytes, %llu pairs): pack ranges of loci",
                    (unsigned long long)read_bases, (unsigned long long)tag_bytes, (unsigned long long)n_pairs);
    const uint32_t nr = (uint32_t)n_pairs;
    if (int rc = raw_reserve(c, nl, nr, g->hap_bytes, read_bases, tag_bytes, true)) return rc;
    if (n_rec) {
        const size_t tb = vtxk_scan_temp_bytes(n_rec);
        HIP_TRY(c, vtxk_inclusive_scan_u32(c->d_bam_nhit.as<uint32_t>(), c->d_bam_hscan.as<uint32_t>(), n_rec, c->d_scan_tmp.p, tb, s));
        HIP_TRY(c, vtxk_inclusive_scan_u32(c->d_bam_rsz.as<uint32_t>(), c->d_bam_rscan.as<uint32_t>(), n_rec, c->d_scan_tmp.p, tb, s));
        HIP_TRY(c, vtxk_inclusive_scan_u32(c->d_bam_tsz.as<uint32_t>(), c->d_bam_tscan.as<uint32_t>(), n_rec, c->d_scan_tmp.p, tb, s));
        HIP_TRY(c, vtxg_scan(1, c->d_bam_data.as<uint8_t>(), c->d_bam_rec.as<uint64_t>(), n_rec, f, d_iv_start, d_iv_end, d_iv_locus, d_tid_begin, d_tid_span,
                             c->d_bam_nhit.as<uint32_t>(), c->d_bam_rsz.as<uint32_t>(), c->d_bam_tsz.as<uint32_t>(), c->d_bam_info.as<vtxg_recinfo>(),
                             c->d_bam_hscan.as<uint32_t>(), c->d_bam_rscan.as<uint32_t>(), c->d_bam_tscan.as<uint32_t>(), c->d_raw.as<vtx_raw_record>(),
                             c->d_raw_locus.as<uint32_t>(), c->d_tags.as<uint8_t>(), c->d_read_packed.as<uint8_t>(), d_counters, d_err, s));
    }
    HIP_TRY(c, vtxk_unpack_nibbles(c->d_read_packed.as<uint8_t>(), read_bases / 2, c->d_read.as<uint8_t>(), s));
    HIP_TRY(c, hipEventRecord(c->ev[3], s));
    HIP_TRY(c, hipStreamSynchronize(s));
    float inflate_ms = 0, index_ms = 0, filter_ms = 0;
    HIP_TRY(c, hipEventElapsedTime(&inflate_ms, c->ev[0], c->ev[1]));
    HIP_TRY(c, hipEventElapsedTime(&index_ms, c->ev[1], c->ev[2]));
    HIP_TRY(c, hipEventElapsedTime(&filter_ms, c->ev[2], c->ev[3]));
    c->bam_n_rec = n_rec; c->bam_n_raw = nr; c->bam_utotal = utotal; c->bam_read_bases = read_bases; c->bam_tag_bytes = tag_bytes;
    vtx_raw_stats rs{};
    if (int rc = raw_prepare(c, nl, nr, read_bases, tag_bytes, true, max_hap, max_hap_all, false, &rs)) return rc;
    if (st) {
        st->num_reads = cnt[0]; st->num_low_mapq = cnt[1]; st->num_non_primary = cnt[2]; st->num_duplicates = cnt[3];
        st->num_not_useful = cnt[4]; st->num_no_barcode_tag = cnt[5];
        st->bam_records = n_rec; st->raw_records = nr; st->inflated_bytes = utotal; st->compressed_bytes = hi - lo;
        st->raw = rs; st->h2d_ms = h2d_ms; st->inflate_ms = inflate_ms; st->index_ms = index_ms; st->filter_ms = filter_ms;
        st->prefetch_ms = prefetched ? c->pf_ms : 0.f; st->prefetch_wait_ms = prefetched ? pf_wait_ms : 0.f;
    }
    return VTX_OK;
}

// Test hook: bgzf_inflate_kernel on arbitrary raw-DEFLATE payloads — block i is file[blocks[i].coff .. + clen) and must inflate to
// exactly blocks[i].isize bytes; status[i] = 0 when the device's decoder accepted it (its bytes at out + the sum of the isizes before
// it), else a vtxi::Status.  Blocks need not be ordered or disjoint (the tests feed truncated copies of one stream).
int vtx_debug_inflate(vtx_ctx* c, const uint8_t* file, uint64_t file_bytes, const vtx_bgzf_block* blk, uint32_t n, uint8_t* out, uint64_t out_cap, uint32_t* status) {
    if (!c || (n && (!blk || !status)) || (file_bytes && !file)) return VTX_E_INVAL;
    std::vector<vtxg_block> blocks(n);
    uint64_t utotal = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (blk[i].coff + blk[i].clen > file_bytes || blk[i].isize > 65536u) return fail(c, VTX_E_INVAL, "vtx_debug_inflate: block %u outside the input or above 64 KiB", i);
        blocks[i] = vtxg_block{blk[i].coff, utotal, blk[i].clen, blk[i].isize};
        utotal += blk[i].isize;
    }
    if (utotal > out_cap) return fail(c, VTX_E_INVAL, "vtx_debug_inflate: output buffer too small");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    if (c->pf_thread.joinable()) c->pf_thread.join();
    c->pf_valid = false;
    HIP_TRY(c, c->d_bam_comp.reserve((size_t)file_bytes + 64));
    HIP_TRY(c, c->d_bam_data.reserve((size_t)utotal + 64));
    HIP_TRY(c, c->d_bam_blocks.reserve((size_t)n * sizeof(vtxg_block) + 16));
    HIP_TRY(c, c->d_bam_cnt.reserve(VTXG_N_COUNTERS * sizeof(uint64_t) + 4 * sizeof(uint32_t)));
    HIP_TRY(c, c->d_bam_seed_cnt.reserve((size_t)n * sizeof(uint32_t) + 16));
    uint32_t* d_err = (uint32_t*)(c->d_bam_cnt.as<unsigned long long>() + VTXG_N_COUNTERS);
    HIP_TRY(c, hipMemsetAsync(d_err, 0, 4 * sizeof(uint32_t), s));
    HIP_TRY(c, hipMemsetAsync(c->d_bam_data.p, 0xEE, (size_t)utotal + 64, s));
    if (file_bytes) HIP_TRY(c, hipMemcpyAsync(c->d_bam_comp.p, file, (size_t)file_bytes, hipMemcpyHostToDevice, s));
    if (n) HIP_TRY(c, hipMemcpyAsync(c->d_bam_blocks.p, blocks.data(), (size_t)n * sizeof(vtxg_block), hipMemcpyHostToDevice, s));
    HIP_TRY(c, vtxg_inflate(c->d_bam_comp.as<uint8_t>(), c->d_bam_blocks.as<vtxg_block>(), n, c->d_bam_data.as<uint8_t>(), d_err, c->d_bam_seed_cnt.as<uint32_t>(), 0, s));
    if (n) HIP_TRY(c, hipMemcpyAsync(status, c->d_bam_seed_cnt.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (utotal && out) HIP_TRY(c, hipMemcpyAsync(out, c->d_bam_data.p, (size_t)utotal, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return VTX_OK;
}

// Test / audit hook: the ingest's intermediate arrays of the last vtx_submit_bam (what: VTX_INGEST_*).  *bytes = the array's size;
// min(cap, *bytes) bytes are copied to dst.
int vtx_debug_ingest(vtx_ctx* c, int what, void* dst, uint64_t cap, uint64_t* bytes) {
    if (!c || !bytes) return VTX_E_INVAL;
    const void* src = nullptr;
    uint64_t n = 0;
    switch (what) {
    case VTX_INGEST_INFLATED: src = c->d_bam_data.p; n = c->bam_utotal; break;
    case VTX_INGEST_RECORD_OFFSETS: src = c->d_bam_rec.p; n = (uint64_t)c->bam_n_rec * 8; break;
    case VTX_INGEST_RAW_RECORDS: src = c->d_raw.p; n = (uint64_t)c->bam_n_raw * sizeof(vtx_raw_record); break;
    case VTX_INGEST_RAW_LOCUS: src = c->d_raw_locus.p; n = (uint64_t)c->bam_n_raw * 4; break;
    case VTX_INGEST_TAGS: src = c->d_tags.p; n = c->bam_tag_bytes; break;
    case VTX_INGEST_READS_PACKED: src = c->d_read_packed.p; n = c->bam_read_bases / 2; break;
    default: return fail(c, VTX_E_INVAL, "vtx_debug_ingest: unknown array %d", what);
    }
    *bytes = n;
    const uint64_t k = std::min(cap, n);
    if (k && dst) {
        HIP_TRY(c, hipSetDevice(c->cfg.device));
        HIP_TRY(c, hipMemcpy(dst, src, (size_t)k, hipMemcpyDeviceToHost));
    }
    return VTX_OK;
}

int vtx_fetch_records(vtx_ctx* c, vtx_record* records, uint32_t* rec_begin, uint32_t* rec_count) {
    if (!c) return VTX_E_INVAL;
    if (!c->submitted) return fail(c, VTX_E_STATE, "vtx_fetch_records: no batch submitted");
    if ((c->n_records && !records) || (c->n_loci && (!rec_begin || !rec_count))) return fail(c, VTX_E_INVAL, "vtx_fetch_records: null output");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (c->n_records) HIP_TRY(c, hipMemcpy(records, c->d_records.p, (size_t)c->n_records * sizeof(vtx_record), hipMemcpyDeviceToHost));
    if (c->n_loci) {
        std::vector<vtx_locus> loci(c->n_loci);
        HIP_TRY(c, hipMemcpy(loci.data(), c->d_loci.p, (size_t)c->n_loci * sizeof(vtx_locus), hipMemcpyDeviceToHost));
        for (uint32_t l = 0; l < c->n_loci; ++l) { rec_begin[l] = loci[l].rec_begin; rec_count[l] = loci[l].rec_count; }
    }
    return VTX_OK;
}

int vtx_run(vtx_ctx* c) {
    if (!c) return VTX_E_INVAL;
    if (!c->submitted) return fail(c, VTX_E_STATE, "vtx_run: no batch submitted");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    const uint32_t nr = c->n_records;
    c->ran = false;
    c->fast_overflow = 0;
    // test / audit hooks (vtx_set_debug): poison the score arrays so that a stage that fails to write a task's score cannot hide
    // behind the previous run's value; one byte per task saying which stage decided it (vtx_fetch_stage)
    uint8_t* stage = nullptr;
    if (c->stage_trace && nr) {
        HIP_TRY(c, c->d_stage.reserve(2 * (size_t)nr));
        stage = c->d_stage.as<uint8_t>();
        HIP_TRY(c, hipMemsetAsync(stage, c->cfg.aligner == VTX_ALIGNER_BANDED ? VTX_STAGE_UNKNOWN : VTX_STAGE_FULL_DP, 2 * (size_t)nr, s));
    }
    if (c->poison && nr) {
        HIP_TRY(c, vtxk_fill_i32(c->d_ref.as<int32_t>(), nr, c->poison_value, s));
        HIP_TRY(c, vtxk_fill_i32(c->d_alt.as<int32_t>(), nr, c->poison_value, s));
    }
    HIP_TRY(c, hipEventRecord(c->ev[0], s));
    uint32_t launches = 0;
    bool any_lut = false;
    const bool full_dp = c->cfg.aligner != VTX_ALIGNER_BANDED;      // the banded flavour never runs the full-matrix DP
    for (const Bucket& bk : c->buckets) any_lut |= bk.lut || bk.duo;
    any_lut &= full_dp;
    if (any_lut) HIP_TRY(c, hipMemsetAsync(c->d_redo_cnt.p, 0, 16 * sizeof(uint32_t), s));
    for (size_t b = 0; full_dp && b < c->buckets.size(); ++b) {
        const Bucket& bk = c->buckets[b];
        static const bool no_duo = VTX_DEV_ENV("VTX_DP_KERNEL") && !strcmp(VTX_DEV_ENV("VTX_DP_KERNEL"), "lut");
        static const bool no_pair = VTX_DEV_ENV("VTX_DP_KERNEL") && !strcmp(VTX_DEV_ENV("VTX_DP_KERNEL"), "duo2");   // two-lookup prefix phase
        const bool use_pair = bk.pair && !no_pair;
        if (bk.duo && !no_duo) {
            HIP_TRY(c, vtxk_launch_sw_full_duo(bk.R, bk.count, c->d_work.as<uint32_t>() + bk.offset, c->d_records.as<vtx_record>(),
                                               c->d_rec_locus.as<uint32_t>(), c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(),
                                               c->d_hap.as<uint8_t>(), c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(),
                                               c->max_hap_len, use_pair ? kPairLociCap : duo_loci_cap(c->max_hap_len),
                                               c->d_redo.as<uint32_t>() + bk.offset, c->d_redo_cnt.as<uint32_t>() + b,
                                               use_pair ? duo_pair_cols(c->max_hap_len) : 0u, s));
        } else if (bk.lut) {
            HIP_TRY(c, vtxk_launch_sw_full_lut(bk.R, bk.count, c->d_work.as<uint32_t>() + bk.offset, c->d_records.as<vtx_record>(),
                                               c->d_rec_locus.as<uint32_t>(), c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(),
                                               c->d_hap.as<uint8_t>(), c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(),
                                               c->max_hap_len, kLutLociCap, c->d_redo.as<uint32_t>() + bk.offset,
                                               c->d_redo_cnt.as<uint32_t>() + b, s));
        } else {
            HIP_TRY(c, vtxk_launch_sw_full(bk.R, bk.GL, bk.count, c->d_work.as<uint32_t>() + bk.offset,
                                           c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(), c->d_loci.as<vtx_locus>(),
                                           c->d_read.as<uint8_t>(), c->d_hap.as<uint8_t>(), c->d_ref.as<int32_t>(),
                                           c->d_alt.as<int32_t>(), c->max_hap_len, s));
        }
        ++launches;
    }
    if (any_lut) {
        // records with bytes outside ACGTN: the generic (byte-equality) kernel scores them
        uint32_t redo[16] = {0};
        HIP_TRY(c, hipMemcpyAsync(redo, c->d_redo_cnt.p, sizeof redo, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        for (size_t b = 0; b < c->buckets.size(); ++b) {
            const Bucket& bk = c->buckets[b];
            if (!(bk.lut || bk.duo) || !redo[b]) continue;
            HIP_TRY(c, vtxk_launch_sw_full(bk.R, bk.GL, redo[b], c->d_redo.as<uint32_t>() + bk.offset, c->d_records.as<vtx_record>(),
                                           c->d_rec_locus.as<uint32_t>(), c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(),
                                           c->d_hap.as<uint8_t>(), c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(), c->max_hap_len, s));
            ++launches;
        }
    }
    uint32_t hard_total = 0;
    float band_run_ms = 0;
    HIP_TRY(c, hipEventRecord(c->ev[3], s));
    if (c->cfg.aligner == VTX_ALIGNER_BANDED && nr) {
        // One pass of the banded stages over tasks [t_begin, t_end) whose locus has its longer haplotype in (mh_min, mh] (the others
        // are left alone); chunk_tables: the tables are built per chunk for the loci the chunk spans, whatever the buffer would hold.
        auto band_pass = [&](const uint32_t mh, const uint32_t mh_min, const uint64_t t_begin, const uint64_t t_end, const bool chunk_tables) -> int {
            // Banded flavour.  Per chunk of tasks (task = 2*record + hap), round 4 (the stages are described in vtx_band.hip's header):
            // tables -> band_diag_kernel (+ band_refine_kernel): scores of the certified tasks, and three lists — tasks whose band is one
            // diagonal stretch (masked DP straight from one word), repeats (band_sweep_kernel + masked DP), the others (band_run_kernel:
            // seeds, chain, general certificate; its hard list -> expand -> masked DP).  What overflows band_run_kernel's lists joins the
            // repeats after the last chunk; what band_sweep_kernel declines twice takes the general band kernel (side stream).
            // Counters (d_cnt, 64 words, zeroed once per run unless noted): [0] hard / [1] overflow / [2..7] reasons / [10] / [11] pending of
            // band_run_kernel (0 and 11 per chunk); [8], [9] general kernel; [12] tasks for band_run_kernel, [13] for band_sweep_kernel,
            // [14] refine records, [15] one-diagonal bands (12..15 per chunk); [16..23] band_run_kernel's block counters; [24] full-matrix
            // check; [26] / [27] hard (per slice) / declined of the sweep's first pass, [28] / [29] of its second; [32..47] reasons of
            // band_diag_kernel; [56..63] reasons of band_sweep_kernel.
            BandPlan bp = band_plan(nr, c->n_loci, mh);
            if (int rc = band_reserve(c, bp, false)) return rc;
            const uint64_t n_tasks = bp.n_tasks;
            const uint32_t chunk = bp.chunk, band_stride = bp.band_stride, hard_cap = bp.hard_cap, pend_cap = bp.pend_cap;
            const uint32_t slots = bp.slots, poly_stride = bp.poly_stride, tasks_per_locus = bp.tasks_per_locus;
            const size_t gt_bytes = bp.gt_bytes;
            const bool gt_chunked = gt_bytes && (bp.gt_loci < c->n_loci || chunk_tables);
            uint32_t fast_overflow = 0;
            uint32_t* d_cnt = c->d_cnt.as<uint32_t>();
            int shape = 0;
            while ((uint32_t)(kShapes[shape][0] * kShapes[shape][1]) < c->max_read_len) ++shape;
            // src == nullptr: the band slots already hold arrays (or the full-matrix marker), one slot per task
            auto masked_dp = [&](uint32_t n_hard, uint32_t* hard, const uint16_t* src, uint16_t* band, uint32_t n_slots, hipStream_t st, uint8_t code) -> int {
                if (stage) HIP_TRY(c, vtxk_mark_stage(hard, n_hard, nullptr, code, stage, st));
                for (uint32_t off = 0; off < n_hard; off += n_slots) {
                    const uint32_t cnt_s = std::min(n_slots, n_hard - off);
                    HIP_TRY(c, vtxk_launch_band_expand(hard + off, cnt_s, c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(),
                                                       c->d_loci.as<vtx_locus>(), src ? src + (size_t)off * poly_stride : band,
                                                       src ? poly_stride : 2 * band_stride, band, band_stride, st));
                    HIP_TRY(c, vtxk_launch_sw_banded(kShapes[shape][0], kShapes[shape][1], cnt_s, hard + off, c->d_records.as<vtx_record>(),
                                                     c->d_rec_locus.as<uint32_t>(), c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(),
                                                     c->d_hap.as<uint8_t>(), band, band_stride, c->d_ref.as<int32_t>(),
                                                     c->d_alt.as<int32_t>(), mh, st));
                    launches += 2;
                }
                return VTX_OK;
            };
            // The general band kernel (tasks band_run_kernel could not hold) is a handful of serial lanes: ~4.5 ms of latency
            // for 0.1 % of config 3.  It runs on a side stream, with its own hard list, beside the pending kernel and the
            // masked DP of the last chunk.  Started once the overflow list is complete; finished (host loop: slabs grow
            // until every task fits) after the main path's launches are queued.
            if (!c->stream2) {
                HIP_TRY(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
                HIP_TRY(c, hipEventCreateWithFlags(&c->ev2, hipEventDisableTiming));
                HIP_TRY(c, hipHostMalloc((void**)&c->h_pin, 64 * sizeof(uint32_t), hipHostMallocDefault));
            }
            hipStream_t s2 = c->stream2;
            struct { uint32_t n_over = 0, cap2 = 0, todo = 0, off = 0, total = 0; const uint32_t* tasks = nullptr; bool active = false; uint32_t gcnt[2] = {0, 0}; } fb;
            // A few overflow tasks (shallow data: some hundreds per run) first try the in-LDS variant of the general kernel
            // with a slab for kLdsMatches k-mer matches: their ~2 ms of serial HBM latency were a third of a shallow step.
            const uint32_t kLdsMatches = 512;
            static const uint32_t kLdsTasks = VTX_DEV_ENV("VTX_BAND_LDS_TASKS") ? (uint32_t)atoi(VTX_DEV_ENV("VTX_BAND_LDS_TASKS")) : 4096u;   // experiment knob
            auto fallback_launch = [&]() -> int {
                const uint64_t worst = (uint64_t)c->max_read_len * mh;
                if (fb.cap2 >= worst && fb.cap2 >= 512) return fail(c, VTX_E_STATE, "vtx_run: band kernel overflow with a worst-case slab");
                // first three rounds: the cooperative kernel, everything in LDS (band_coop_kernel: a wavefront per task, up to 512 matches,
                // then 1024, then 4096; reads up to 256 bases); what that cannot hold takes the serial kernel below
                static const bool no_coop = VTX_DEV_ENV("VTX_BAND_NO_COOP") != nullptr;             // experiment / test hook
                const int tier = fb.cap2 < 512 ? 0 : (fb.cap2 == 512 ? 1 : (fb.cap2 == 1024 ? 2 : -1));
                const uint32_t tier_cap = tier == 0 ? 512u : (tier == 1 ? 1024u : 4096u);
                // (haplotypes up to 1000 bases: the kernel walks its Fenwick tree in ten unrolled steps, tn = n + 8 < 1024)
                if (tier >= 0 && !no_coop && c->max_read_len <= 256 && mh <= 1000 &&
                    vtxk_band_coop_lds(mh, tier_cap) <= 64u * 1024) {
                    fb.cap2 = tier_cap;
                    HIP_TRY(c, hipMemsetAsync(d_cnt + 9, 0, sizeof(uint32_t), s2));
                    uint32_t* other = c->d_over2.as<uint32_t>() + ((fb.tasks == c->d_over2.as<uint32_t>()) ? fb.n_over : 0);
                    HIP_TRY(c, vtxk_launch_band_coop(tier, fb.tasks, fb.todo, c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(),
                                                     c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(), c->d_hap.as<uint8_t>(), mh,
                                                     c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(), c->d_band2.as<uint16_t>(), band_stride,
                                                     c->d_hard2.as<uint32_t>(), other, d_cnt + 8, s2));
                    HIP_TRY(c, hipMemcpyAsync(c->h_pin, d_cnt + 8, sizeof fb.gcnt, hipMemcpyDeviceToHost, s2));
                    ++launches;
                    return VTX_OK;
                }
                const bool in_lds = fb.cap2 < 512 && fb.todo <= kLdsTasks && !VTX_DEV_ENV("VTX_BAND_NO_LDS_FALLBACK") &&
                                    vtxk_band_lds_stride(kLdsMatches, mh, c->max_read_len) <= 160 * 1024 - 512;
                fb.cap2 = in_lds ? kLdsMatches : (uint32_t)std::max<uint64_t>(std::min<uint64_t>((uint64_t)fb.cap2 * 16, worst), 512);
                const size_t stride2 = vtxk_band_ws_stride(fb.cap2, mh);
                if (!in_lds) {                                                  // whole wavefronts: the 64 slabs are interleaved, vtxk_band_lanes() of them in use
                    const size_t lanes = vtxk_band_lanes(fb.todo);
                    HIP_TRY(c, c->d_band_ws2.reserve(lanes < 64 ? (size_t)fb.todo * stride2 : ((size_t)fb.todo + 63) / 64 * 64 * stride2));   // (sparse lanes: a contiguous slab per task)
                }
                HIP_TRY(c, hipMemsetAsync(d_cnt + 9, 0, sizeof(uint32_t), s2));
                uint32_t* other = c->d_over2.as<uint32_t>() + ((fb.tasks == c->d_over2.as<uint32_t>()) ? fb.n_over : 0);
                HIP_TRY(c, vtxk_launch_band(fb.tasks, fb.todo, 0, c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(),
                                            c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(), c->d_hap.as<uint8_t>(),
                                            c->d_band_ws2.as<uint8_t>(), stride2, fb.cap2, mh, c->d_ref.as<int32_t>(),
                                            c->d_alt.as<int32_t>(), c->d_band2.as<uint16_t>(), band_stride, c->d_hard2.as<uint32_t>(),
                                            other, d_cnt + 8, in_lds ? 1 : 0, c->max_read_len, s2));
                HIP_TRY(c, hipMemcpyAsync(c->h_pin, d_cnt + 8, sizeof fb.gcnt, hipMemcpyDeviceToHost, s2));   // pinned: does not block
                ++launches;
                return VTX_OK;
            };
            auto fallback_start = [&](uint32_t off, uint32_t total) -> int {   // the overflow list d_over[0, total) is complete and visible
                const uint32_t n_over = std::min(std::max(slots, 1024u), total - off);   // slices (bounds d_band2)
                fb.off = off; fb.total = total;
                fb.n_over = n_over; fb.todo = n_over; fb.cap2 = 512 / 16; fb.tasks = c->d_over.as<uint32_t>() + off; fb.active = true;
                HIP_TRY(c, c->d_over2.reserve(2 * (size_t)n_over * sizeof(uint32_t)));
                HIP_TRY(c, c->d_hard2.reserve((size_t)n_over * sizeof(uint32_t)));
                HIP_TRY(c, c->d_band2.reserve((size_t)n_over * 2 * band_stride * sizeof(uint16_t)));
                HIP_TRY(c, hipMemsetAsync(d_cnt + 8, 0, 2 * sizeof(uint32_t), s2));
                return fallback_launch();
            };
            auto fallback_finish = [&]() -> int {
                if (!fb.active) return VTX_OK;
                for (;;) {
                    for (;;) {
                        HIP_TRY(c, hipStreamSynchronize(s2));
                        fb.gcnt[0] = c->h_pin[0]; fb.gcnt[1] = c->h_pin[1];
                        // tasks that still do not fit were written to the other half of d_over2: rerun them with a larger slab
                        fb.tasks = c->d_over2.as<uint32_t>() + ((fb.tasks == c->d_over2.as<uint32_t>()) ? fb.n_over : 0);
                        fb.todo = fb.gcnt[1];
                        if (!fb.todo) break;
                        if (int rc = fallback_launch()) return rc;
                    }
                    if (int rc = masked_dp(fb.gcnt[0], c->d_hard2.as<uint32_t>(), nullptr, c->d_band2.as<uint16_t>(), std::max(fb.gcnt[0], 1u), s2, VTX_STAGE_GENERAL_DP)) return rc;
                    hard_total += fb.gcnt[0];
                    if (fb.off + fb.n_over >= fb.total) break;
                    if (int rc = fallback_start(fb.off + fb.n_over, fb.total)) return rc;      // next slice (same stream: in order)
                }
                HIP_TRY(c, hipEventRecord(c->ev2, s2));
                HIP_TRY(c, hipStreamWaitEvent(s, c->ev2, 0));             // the reduction kernels read every score
                return VTX_OK;
            };
            HIP_TRY(c, hipMemsetAsync(d_cnt, 0, 64 * sizeof(uint32_t), s));
            uint32_t cnt[12] = {0};
            uint32_t pending_total = 0, over_before = 0;
            uint64_t diag_total = 0, diag_left = 0, refined_total = 0, checked_total = 0, swept_total = 0, diag2_total = 0, diag2_scored = 0, tight2_total = 0, stream_total = 0;
            float diag_ms = 0, check_ms = 0, sweep_ms = 0;
            // the band of every listed task (band_sweep_kernel, tier 0 / 1), one slice of band slots at a time, then the masked DP over
            // the slice (its length — the tasks the sweep did not decline — is read on the device: counters[0]; declined: counters[1])
            // (libvtx_dev.so, VTX_SWEEP_V1=1: round 4's kernel instead — 256 sections per task, then a second pass with 1 024 over what the
            // first declined; the A/B reference of tests/test_gpu_sweep.py and tools/gpu_campaign.sh)
            static const bool sweep_v1 = VTX_DEV_ENV("VTX_SWEEP_V1") != nullptr;
            auto sweep_slices = [&](int tier, const uint32_t* list, uint32_t n, uint32_t* over_out, uint32_t* counters) -> int {
                static const bool sweep_stats = getenv("VTX_DEBUG") != nullptr;
                HIP_TRY(c, c->d_sweep_log.reserve(vtxk_band_sweep_log_bytes()));
                for (uint32_t off = 0; off < n; off += slots) {
                    const uint32_t cnt_s = std::min(slots, n - off);
                    HIP_TRY(c, hipMemsetAsync(counters, 0, sizeof(uint32_t), s));
    #ifdef VTX_DEVTOOLS
                    if (sweep_v1)
                        HIP_TRY(c, vtxk_launch_band_sweep_v1(tier, list + off, cnt_s, nullptr, c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(),
                                                             c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(), c->d_hap.as<uint8_t>(),
                                                             c->d_band.as<uint16_t>(), band_stride, c->d_hard.as<uint32_t>(), over_out,
                                                             counters, sweep_stats ? d_cnt + 56 : nullptr, stage, nullptr, s));
                    else
    #endif
                    HIP_TRY(c, vtxk_launch_band_sweep(list + off, cnt_s, nullptr, c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(),
                                                      c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(), c->d_hap.as<uint8_t>(),
                                                      c->d_band.as<uint16_t>(), band_stride, c->d_hard.as<uint32_t>(), over_out,
                                                      counters, sweep_stats ? d_cnt + 56 : nullptr, stage, nullptr, c->d_sweep_log.as<uint32_t>(), s));
                    (void)tier;
                    HIP_TRY(c, vtxk_launch_sw_banded_dev(kShapes[shape][0], kShapes[shape][1], cnt_s, c->d_hard.as<uint32_t>(), counters,
                                                         c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(), c->d_loci.as<vtx_locus>(),
                                                         c->d_read.as<uint8_t>(), c->d_hap.as<uint8_t>(), c->d_band.as<uint16_t>(), band_stride,
                                                         c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(), mh, s));
                    launches += 2;
                }
                return VTX_OK;
            };
            uint32_t resweep_total = 0;
            // Second stage (round 5): the same single-diagonal logic with a list of 64 entries and the harmless bound from
            // the matches that can really precede a match (band_diag2_kernel), the harmless test alone over a two-row window for what
            // exceeds the list (band_stream_kernel).  They score most of these tasks or prove that their band is one diagonal stretch
            // (full-matrix check, masked DP for what it does not settle; no sweep); what they leave — out[0, *n_out) — takes the sweep.
            // One host round trip for the counts.  (libvtx_dev.so: VTX_BAND_NO_DIAG2=1 sends everything to the sweep, as round 4 did;
            // VTX_BAND_NO_STREAM=1 — what exceeds the second stage's list goes to the sweep.)
            // A short list skips it: one lane per task, a few hundred dependent loads each — the kernels are latency chains with a fixed
            // cost of several milliseconds that the sweep + its DP beat below ~0.65 M tasks.  Measured in round 6 on loci from real sequence
            // (profiles/r06_second_stage_threshold.txt; stage on / off): 33 k tasks 7.9 / 3.4 ms per step, 145 k 12.8 / 8.6, 319 k 28.9 / 19.9,
            // 638 k 36.8 / 36.0, 1.28 M 52.2 / 64.3 — round 5's threshold of 200 k made mid-size batches a third slower.
            // The tables of the tasks' loci have to be resident (gt_l0: first locus of the table buffer).
            auto second_stage_on = [&](uint32_t n) -> bool {
                static const bool no_diag2 = VTX_DEV_ENV("VTX_BAND_NO_DIAG2") != nullptr;
                static const uint32_t diag2_min = VTX_DEV_ENV("VTX_BAND_DIAG2_MIN") ? (uint32_t)strtoul(VTX_DEV_ENV("VTX_BAND_DIAG2_MIN"), nullptr, 10) : 700000u;
                return !no_diag2 && n >= diag2_min && mh <= 255;
            };
            auto second_stage = [&](const uint32_t* list, uint32_t n, uint32_t gt_l0, uint32_t* out, uint32_t* n_out) -> int {
                static const bool no_stream = VTX_DEV_ENV("VTX_BAND_NO_STREAM") != nullptr;
                HIP_TRY(c, hipMemsetAsync(d_cnt + 30, 0, 2 * sizeof(uint32_t), s));
                HIP_TRY(c, hipMemsetAsync(d_cnt + 48, 0, sizeof(uint32_t), s));
                HIP_TRY(c, vtxk_launch_band_diag2(list, n, c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(),
                                                  c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(), mh,
                                                  c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(), bp.tasks_per_locus, gt_l0,
                                                  c->d_gtables.as<uint8_t>(), out, c->d_tight2.as<uint32_t>(),
                                                  c->d_tight2_pack.as<uint32_t>(), d_cnt + 30,
                                                  no_stream ? nullptr : c->d_recheck2.as<uint32_t>(), c->d_recheck2_pack.as<uint32_t>(), d_cnt + 48, stage, s));
                HIP_TRY(c, hipMemcpyAsync(c->h_pin + 14, d_cnt + 30, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                HIP_TRY(c, hipMemcpyAsync(c->h_pin + 16, d_cnt + 48, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                HIP_TRY(c, hipStreamSynchronize(s));
                stream_total += std::min(c->h_pin[16], n);
                const uint32_t n_sweep = std::min(c->h_pin[14], n);
                const uint32_t n_tight2 = std::min(c->h_pin[15], n - n_sweep);
                ++launches;
                diag2_total += n; diag2_scored += n - n_sweep - n_tight2;
                if (n_tight2) {
                    // These tasks hold a certificate (a lower bound of the banded score) and sit in repeat-rich sequence on
                    // clean reads: the full-matrix score equals it for practically all of them (4 223 of 4 223 in the CPU
                    // sample), and cert <= banded <= full then decides the task for 5.9 ns where the masked DP takes 10.
                    // What the check does not settle takes the masked DP over its one-diagonal band (count on the device).
                    HIP_TRY(c, hipMemsetAsync(d_cnt + 25, 0, sizeof(uint32_t), s));
                    HIP_TRY(c, vtxk_launch_sw_check(kShapes[shape][0], kShapes[shape][1], n_tight2, c->d_tight2.as<uint32_t>(),
                                                    c->d_tight2_pack.as<uint32_t>(), nullptr, c->d_records.as<vtx_record>(),
                                                    c->d_rec_locus.as<uint32_t>(), c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(),
                                                    c->d_hap.as<uint8_t>(), c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(), mh,
                                                    c->d_recheck2.as<uint32_t>(), c->d_recheck2_pack.as<uint32_t>(), d_cnt + 25, stage, s));
                    HIP_TRY(c, vtxk_launch_sw_diag_band(kShapes[shape][0], kShapes[shape][1], n_tight2, c->d_recheck2.as<uint32_t>(),
                                                        c->d_recheck2_pack.as<uint32_t>(), d_cnt + 25, c->d_records.as<vtx_record>(),
                                                        c->d_rec_locus.as<uint32_t>(), c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(),
                                                        c->d_hap.as<uint8_t>(), c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(),
                                                        mh, stage, s));
                    // (its grid is sized for n_tight2 although a few hundred tasks remain: the workgroups past the device count leave at once —
                    //  measured in round 6 with an exact-size launch behind a host round trip: no difference.  The 3 x 13 ms of
                    //  sw_banded_kernel<.., 2> in the real-sequence kernel table are ONE launch of 39 ms — the first stage's 0.5 M one-diagonal
                    //  bands on the side stream, stretched by the kernels it runs beside — and two of microseconds.)
                    launches += 2;
                    tight2_total += n_tight2;
                }
                *n_out = n_sweep;
                return 0;
            };
            bool sweep_used = false;                    // some chunk took the round-4 path
            bool sweep_pending = false;                 // the events of a swept chunk have not been read yet
            bool sweep_forked = false;                  // ... and that chunk ran its two branches side by side
            uint32_t fork_nt = 0;
            auto collect_sweep_times = [&]() -> int {
                if (!sweep_pending) return VTX_OK;
                sweep_pending = false;
                HIP_TRY(c, hipEventSynchronize(c->ev[8]));
                float ms = 0;
                // ev[7] .. ev[9]: band_refine_kernel (forked only) + the one-diagonal bands' masked DP; then band_sweep_kernel + its masked
                // DP (the chunk's repeats) — forked: ev[11] .. ev[8] on the main stream, BESIDE the first interval, not after it
                HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[7], c->ev[9])); check_ms += ms;
                HIP_TRY(c, hipEventElapsedTime(&ms, sweep_forked ? c->ev[11] : c->ev[9], c->ev[8])); sweep_ms += ms;
                if (sweep_forked) checked_total += std::min(c->h_pin[12], fork_nt);          // (copied before ev[9], which ev[8] waited for)
                sweep_forked = false;
                return VTX_OK;
            };
            for (uint64_t base = t_begin; base < t_end; base += chunk) {
                const uint32_t nt = (uint32_t)std::min<uint64_t>(chunk, t_end - base);
                if (int rc = collect_sweep_times()) return rc;                              // (a chunk re-records the events)
                HIP_TRY(c, hipMemsetAsync(d_cnt, 0, sizeof(uint32_t), s));                 // hard count of this chunk
                HIP_TRY(c, hipMemsetAsync(d_cnt + 11, 0, sizeof(uint32_t), s));            // pending count of this chunk
                HIP_TRY(c, hipMemsetAsync(d_cnt + 16, 0, 8 * sizeof(uint32_t), s));        // block counters (one per XCD)
                // the loci of this range of tasks (tables in global memory are built per range)
                uint32_t gt_l0 = 0, gt_n = gt_bytes ? c->n_loci : 0;
                if (gt_chunked) {
                    uint32_t ends[2];
                    HIP_TRY(c, hipMemcpyAsync(&ends[0], c->d_rec_locus.as<uint32_t>() + base / 2, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                    HIP_TRY(c, hipMemcpyAsync(&ends[1], c->d_rec_locus.as<uint32_t>() + (base + nt - 1) / 2, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                    HIP_TRY(c, hipStreamSynchronize(s));
                    gt_l0 = ends[0];
                    gt_n = ends[1] - ends[0] + 1;
                    if (gt_n > bp.gt_loci) gt_n = 0;              // does not fit after all: tables in LDS for this chunk
                }
                c->gt_used = (gt_n && bp.gt_loci) ? (uint64_t)gt_n * (gt_bytes / bp.gt_loci) : 0;     // (gt_bytes = gt_loci x bytes per locus)
                HIP_TRY(c, hipEventRecord(c->ev[4], s));
                // Stage 1 (tables in global memory): band_diag_kernel decides the tasks whose alignment lives on one diagonal
                // (vtx_fast_core.h) and lists the others; band_run_kernel then takes that LIST instead of the whole range.
                static const bool no_diag = VTX_DEV_ENV("VTX_BAND_NO_DIAG") != nullptr;          // experiment / test hook: stage 1 off
                static const int diag_stats = getenv("VTX_DEBUG") ? 1 : 0;
                bool diag = false, swept = false;
                uint32_t n_fail = 0;
                const uint32_t* fail_list = c->d_fail.as<uint32_t>();
                // Round 4: what band_diag_kernel / band_refine_kernel leave goes (a) with a certificate: through the full-matrix CHECK
                // (full == cert decides it: cert <= banded <= full), (b) otherwise, or when the check fails: through band_sweep_kernel
                // (the band of ANY task, vtx_sweep.hip) and the masked DP.  VTX_BAND_LEGACY=1: round 3's band_run_kernel / pending /
                // general path instead (kept for A/B tests; also what takes over when a haplotype of the batch exceeds 255 bases).
                static const bool legacy = VTX_DEV_ENV("VTX_BAND_LEGACY") != nullptr;
                static const bool no_tight = VTX_DEV_ENV("VTX_BAND_NO_TIGHT") != nullptr;          // test hook: tasks with a certificate go to the sweep like the others
                static const bool use_check = VTX_DEV_ENV("VTX_BAND_CHECK") != nullptr;            // experiment hook: full-matrix check in front of their DP
                const bool sweep_path = !legacy && mh <= vtxk_band_sweep_max_len() && mh > 0;
                uint32_t* tight_list = (sweep_path && !no_tight) ? c->d_tight.as<uint32_t>() : nullptr;
                uint32_t* tight_pack = tight_list ? c->d_tight_pack.as<uint32_t>() : nullptr;
                // which of the tasks the certificate stages leave skip band_run_kernel (whose piece lists they would overflow) and take
                // band_sweep_kernel at once: bit = vtxf::Why.  Default: W_MATCHES (4: more than 40 off-diagonal k-mer matches — repeats).
                // VTX_BAND_DENSE_MASK: experiment knob (0x3be: everything but shape; 0: nothing — band_run_kernel sees every task first).
                static const uint32_t dense_mask = VTX_DEV_ENV("VTX_BAND_DENSE_MASK") ? (uint32_t)strtoul(VTX_DEV_ENV("VTX_BAND_DENSE_MASK"), nullptr, 0) : (1u << 4);
                uint32_t* dense_list = sweep_path ? c->d_dense.as<uint32_t>() : nullptr;
                if (gt_n && !no_diag) {
                    HIP_TRY(c, hipMemsetAsync(d_cnt + 12, 0, 4 * sizeof(uint32_t), s));   // [12] left for band_run_kernel, [13] for band_sweep_kernel, [14] refine records, [15] tasks with a one-diagonal band
                    // tasks with main pieces only whose bounds do not meet leave a record for band_refine_kernel
                    static const bool no_refine = VTX_DEV_ENV("VTX_BAND_NO_REFINE") != nullptr;        // experiment / test hook
                    const uint32_t refine_cap = band_refine_cap(chunk);
                    uint32_t* refine_list = no_refine ? nullptr : c->d_refine.as<uint32_t>();
                    const hipError_t e = vtxk_launch_band_diag(nt, (uint32_t)base, c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(),
                                                               c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(), c->d_hap.as<uint8_t>(),
                                                               mh, mh_min, c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(),
                                                               c->d_fail.as<uint32_t>(), refine_list, refine_cap, d_cnt, tasks_per_locus, gt_l0, gt_n,
                                                               c->d_gtables.as<uint8_t>(), gt_bytes, diag_stats, tight_list, tight_pack, stage,
                                                               dense_list, dense_mask, c->max_read_len, s);
                    if (e == hipSuccess && sweep_path) {
                        diag = true; swept = true; sweep_used = true;
                        HIP_TRY(c, hipEventRecord(c->ev[6], s));
                        // What the stage left, in two branches that share nothing but the score arrays (disjoint tasks):
                        //   side stream   band_refine_kernel over its records, then the masked DP over the one-diagonal bands (tight list:
                        //                 band_diag_kernel's entries + what the refinement leaves; counted on the device, the grid is
                        //                 sized for the bound known here);
                        //   this stream   band_sweep_kernel + masked DP over the repeats and the short fail list (both final once
                        //                 band_diag_kernel is done: with a tight list the refinement adds nothing to them).
                        // At 16 reads per locus these are five latency-bound launches of 0.1 - 0.2 ms each: side by side 0.31 instead
                        // of 0.55 ms.  One host round trip, right after band_diag_kernel.  (VTX_BAND_NO_TIGHT: the refinement's leftovers
                        // go to the fail list, so everything stays in order on this stream.)
                        static const bool no_fork = VTX_DEV_ENV("VTX_BAND_NO_FORK") != nullptr;          // experiment / test hook
                        const bool fork = tight_list != nullptr && !no_fork;
                        hipStream_t sb = fork ? s2 : s;                                             // the refine / one-diagonal branch
                        // the records band_diag_kernel left: with a tight list (round 6) every task that holds a certificate and a one-diagonal
                        // band, for band_corridor_kernel; else (VTX_BAND_NO_TIGHT, libvtx_dev.so's VTX_BAND_NO_CORRIDOR) round 3's, for band_refine_kernel
                        static const bool no_corridor = VTX_DEV_ENV("VTX_BAND_NO_CORRIDOR") != nullptr;
                        auto second_look = [&](uint32_t n_rec, hipStream_t st) -> hipError_t {
                            if (tight_list && !no_corridor)
                                return vtxk_launch_band_corridor(refine_list, n_rec, c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(),
                                                                 c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(), c->d_hap.as<uint8_t>(),
                                                                 c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(), d_cnt, diag_stats, tight_list, tight_pack,
                                                                 stage, d_cnt + 14, st);
                            return vtxk_launch_band_refine(refine_list, n_rec, c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(),
                                                           c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(), mh,
                                                           c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(), c->d_fail.as<uint32_t>(), d_cnt,
                                                           tasks_per_locus, gt_l0, c->d_gtables.as<uint8_t>(), diag_stats, tight_list, tight_pack, stage, d_cnt + 14, st);
                        };
                        if (!fork && refine_list) HIP_TRY(c, second_look(std::min(refine_cap, nt), s));
                        HIP_TRY(c, hipMemcpyAsync(c->h_pin + 8, d_cnt + 12, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));   // [12] fail, [13] dense, [14] refine, [15] tight
                        HIP_TRY(c, hipStreamSynchronize(s));
                        const uint32_t n_refine = std::min(c->h_pin[10], refine_cap);
                        // (forked: the tight list still grows by what the refinement leaves — at most its records)
                        const uint32_t n_tight = (uint32_t)std::min<uint64_t>((uint64_t)c->h_pin[11] + (fork && refine_list ? n_refine : 0u), nt);
                        refined_total += n_refine;
                        launches += 2;
                        if (fork) HIP_TRY(c, hipStreamWaitEvent(s2, c->ev[6], 0));
                        HIP_TRY(c, hipEventRecord(c->ev[7], sb));
                        if (fork && refine_list && n_refine) HIP_TRY(c, second_look(n_refine, sb));
                        if (n_tight) {
                            // tasks with a certificate but no verdict: their band is one diagonal stretch (tight_pack): the masked DP
                            // expands it itself.  (VTX_BAND_CHECK=1: the full-matrix check first — full == cert decides a task, cert <=
                            // banded <= full; measured: 5.6 ns per task against 10 for the DP it saves on 20 - 30 % of noisy reads.)
                            const uint32_t* dp_list = tight_list;
                            const uint32_t* dp_pack = tight_pack;
                            const uint32_t* dp_cnt = fork ? d_cnt + 15 : nullptr;
                            if (use_check) {
                                HIP_TRY(c, c->d_dband.reserve((size_t)chunk * sizeof(uint32_t)));
                                HIP_TRY(c, c->d_dband_pack.reserve((size_t)chunk * sizeof(uint32_t)));
                                HIP_TRY(c, hipMemsetAsync(d_cnt + 24, 0, sizeof(uint32_t), sb));
                                HIP_TRY(c, vtxk_launch_sw_check(kShapes[shape][0], kShapes[shape][1], n_tight, tight_list, tight_pack, dp_cnt,
                                                                c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(), c->d_loci.as<vtx_locus>(),
                                                                c->d_read.as<uint8_t>(), c->d_hap.as<uint8_t>(), c->d_ref.as<int32_t>(),
                                                                c->d_alt.as<int32_t>(), mh, c->d_dband.as<uint32_t>(),
                                                                c->d_dband_pack.as<uint32_t>(), d_cnt + 24, stage, sb));
                                dp_list = c->d_dband.as<uint32_t>(); dp_pack = c->d_dband_pack.as<uint32_t>(); dp_cnt = d_cnt + 24;
                                ++launches;
                            }
                            HIP_TRY(c, vtxk_launch_sw_diag_band(kShapes[shape][0], kShapes[shape][1], n_tight, dp_list, dp_pack, dp_cnt,
                                                                c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(), c->d_loci.as<vtx_locus>(),
                                                                c->d_read.as<uint8_t>(), c->d_hap.as<uint8_t>(), c->d_ref.as<int32_t>(),
                                                                c->d_alt.as<int32_t>(), mh, stage, sb));
                            ++launches;
                        }
                        if (fork) HIP_TRY(c, hipMemcpyAsync(c->h_pin + 12, d_cnt + 15, sizeof(uint32_t), hipMemcpyDeviceToHost, sb));   // the list's final length (read in collect_sweep_times)
                        HIP_TRY(c, hipEventRecord(c->ev[9], sb));
                        if (fork) HIP_TRY(c, hipEventRecord(c->ev[11], s));                          // (this stream's branch starts here)
                        n_fail = c->h_pin[8];
                        uint32_t n_dense = std::min(c->h_pin[9], nt);
                        if (!fork) checked_total += n_tight;                                    // (forked: the exact count arrives with the events)
                        diag_total += nt; diag_left += (uint64_t)n_fail + n_dense;
                        // repeats: band_sweep_kernel (the band of ANY task) + masked DP, sorted by task (neighbours share their locus'
                        // haplotypes, and the hard list comes out in a fixed order); what it declines waits in d_over[2 n_tasks ..) for the
                        // second pass after the last chunk
                        // (a short list of the other tasks is not worth band_run_kernel's launch — a persistent grid: ~1 ms whatever the
                        // count — and the two host round trips behind it: it joins the repeats, ~25 ns per task)
                        static const uint32_t run_min = VTX_DEV_ENV("VTX_BAND_RUN_MIN") ? (uint32_t)strtoul(VTX_DEV_ENV("VTX_BAND_RUN_MIN"), nullptr, 10) : 65536u;
                        if (n_fail && n_fail < run_min && (uint64_t)n_dense + n_fail <= nt) {
                            HIP_TRY(c, hipMemcpyAsync(dense_list + n_dense, c->d_fail.as<uint32_t>(), (size_t)n_fail * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
                            n_dense += n_fail;
                            n_fail = 0;
                        }
                        if (n_dense) {
                            const uint32_t* dl = dense_list;
                            if (n_dense > 64) {
                                const size_t tb = vtxk_sort_keys_u32_temp_bytes(n_dense);
                                if (c->d_fail_tmp.reserve(tb) == hipSuccess) {
                                    HIP_TRY(c, vtxk_sort_keys_u32(dense_list, dense_list + nt, n_dense, c->d_fail_tmp.p, tb, s));
                                    dl = dense_list + nt;
                                } else (void)hipGetLastError();
                            }
                            // Second stage (round 5): band_diag2_kernel + band_stream_kernel + the full-matrix check (second_stage above)
                            const uint32_t* sl = dl;
                            uint32_t n_sweep = n_dense;
                            if (second_stage_on(n_dense)) {
                                uint32_t* sweep2 = (dl == dense_list) ? dense_list + nt : dense_list;          // (the half of d_dense the list is not in)
                                if (int rc = second_stage(dl, n_dense, gt_l0, sweep2, &n_sweep)) return rc;
                                sl = sweep2;
                            }
                            if (n_sweep) { if (int rc = sweep_slices(0, sl, n_sweep, c->d_over.as<uint32_t>() + 2 * n_tasks, d_cnt + 26)) return rc; }
                            swept_total += n_sweep;
                        }
                        if (n_fail > 64) {
                            const size_t tb = vtxk_sort_keys_u32_temp_bytes(n_fail);
                            if (c->d_fail_tmp.reserve(tb) == hipSuccess) {
                                HIP_TRY(c, vtxk_sort_keys_u32(c->d_fail.as<uint32_t>(), c->d_fail.as<uint32_t>() + nt, n_fail, c->d_fail_tmp.p, tb, s));
                                fail_list = c->d_fail.as<uint32_t>() + nt;
                            } else (void)hipGetLastError();
                        }
                        if (fork) HIP_TRY(c, hipStreamWaitEvent(s, c->ev[9], 0));                    // join: what follows reads every score
                        HIP_TRY(c, hipEventRecord(c->ev[8], s));
                        sweep_pending = true; sweep_forked = fork; fork_nt = nt;
                        // the others: band_run_kernel (task-list mode) below — seeds, chain and the general certificate (a read against the
                        // other allele of an indel lies on TWO diagonals: cert == ub decides nearly all of those without a DP cell)
                    } else if (e == hipSuccess) {
                        diag = true;
                        HIP_TRY(c, hipMemcpyAsync(c->h_pin + 8, d_cnt + 12, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                        HIP_TRY(c, hipMemcpyAsync(c->h_pin + 9, d_cnt + 14, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                        HIP_TRY(c, hipEventRecord(c->ev[6], s));
                        HIP_TRY(c, hipStreamSynchronize(s));
                        n_fail = c->h_pin[8];
                        const uint32_t n_refine = std::min(c->h_pin[9], refine_cap);
                        if (n_refine) {
                            HIP_TRY(c, vtxk_launch_band_refine(refine_list, n_refine, c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(),
                                                               c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(), mh,
                                                               c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(), c->d_fail.as<uint32_t>(), d_cnt,
                                                               tasks_per_locus, gt_l0, c->d_gtables.as<uint8_t>(), diag_stats, nullptr, nullptr, stage, nullptr, s));
                            HIP_TRY(c, hipMemcpyAsync(c->h_pin + 8, d_cnt + 12, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                            HIP_TRY(c, hipStreamSynchronize(s));
                            n_fail = c->h_pin[8];
                            refined_total += n_refine;
                            ++launches;
                        }
                        diag_total += nt; diag_left += n_fail;
                        ++launches;
                        if (n_fail > 64) {
                            // the list comes out in the order the wavefronts finished: eight XCD ranges interleaved.  Sorted by task, the
                            // 64 tasks of a band_run_kernel wavefront share their loci's tables and reads again (12.7 -> GB of L2 misses
                            // for 2 % of the tasks otherwise)
                            const size_t tb = vtxk_sort_keys_u32_temp_bytes(n_fail);
                            if (c->d_fail_tmp.reserve(tb) == hipSuccess) {
                                HIP_TRY(c, vtxk_sort_keys_u32(c->d_fail.as<uint32_t>(), c->d_fail.as<uint32_t>() + nt, n_fail, c->d_fail_tmp.p, tb, s));
                                fail_list = c->d_fail.as<uint32_t>() + nt;
                            } else (void)hipGetLastError();
                        }
                    } else {
                        // (the tables do not fit the buffer for this chunk: band_run_kernel alone, tables in LDS)
                        if (getenv("VTX_DEBUG")) fprintf(stderr, "[vtx] band_diag_kernel not launched for tasks [%llu, +%u): %s\n", (unsigned long long)base, nt, hipGetErrorString(e));
                        (void)hipGetLastError();
                    }
                }
                HIP_TRY(c, hipEventRecord(c->ev[10], s));
                if (!diag || n_fail)
                    HIP_TRY(c, vtxk_launch_band_run(diag ? n_fail : nt, diag ? 0u : (uint32_t)base, c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(),
                                                     c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(), c->d_hap.as<uint8_t>(),
                                                     mh, mh_min, c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(),
                                                     c->d_band_ws.as<uint32_t>(), c->d_poly.as<uint16_t>(), poly_stride / 2,
                                                     c->d_hard.as<uint32_t>(), c->d_over.as<uint32_t>(), c->d_pend.as<uint32_t>(),
                                                     c->d_pend_buf.as<uint32_t>(), hard_cap, pend_cap, d_cnt,
                                                     tasks_per_locus, gt_l0, gt_n, gt_n ? c->d_gtables.as<uint8_t>() : nullptr, gt_bytes,
                                                     diag ? fail_list : nullptr, c->band_long_lists ? 1 : 0, s));
                HIP_TRY(c, hipEventRecord(c->ev[5], s));                  // (complete once the read-back below is: no synchronisation of its own)
                const bool run_skipped = swept && n_fail == 0;             // nothing went to band_run_kernel: its counters are what they were
                if (run_skipped) {
                    cnt[0] = 0; cnt[11] = 0; cnt[1] = over_before;
                } else {
                    HIP_TRY(c, hipMemcpyAsync(cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost, s));
                    HIP_TRY(c, hipStreamSynchronize(s));
                }
                // (task-list mode runs the 15-entry variant: nothing to give a second chance to)
                const bool short_lists = !diag && gt_n && vtxk_band_second_chance(tasks_per_locus, c->band_long_lists ? 1 : 0);
                if (short_lists && nt >= (1u << 20)) {
                    // feedback for the next run of this context: many overflows of the 12-entry lists (noisy reads: 3.3 % of the
                    // tasks at 3 % substitution errors, 0.2 % at 0.5 %) make the 15-entry variant the better first pass
                    if ((uint64_t)(cnt[1] - over_before) * 50 > nt) c->band_long_lists = true;
                }
                if (short_lists && cnt[1] > over_before) {
                    // The six-wavefront variant keeps 12-entry lists: the tasks that overflowed them ([a0, a1) of the overflow
                    // list) get a second chance in the 15-entry variant before the general kernel — what overflows again is
                    // appended behind a1 and then moved down to a0.
                    const uint32_t a0 = over_before, a1 = cnt[1];
                    HIP_TRY(c, hipMemsetAsync(d_cnt + 16, 0, 8 * sizeof(uint32_t), s));
                    HIP_TRY(c, vtxk_launch_band_run(a1 - a0, 0, c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(),
                                                     c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(), c->d_hap.as<uint8_t>(),
                                                     mh, mh_min, c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(),
                                                     c->d_band_ws.as<uint32_t>(), c->d_poly.as<uint16_t>(), poly_stride / 2,
                                                     c->d_hard.as<uint32_t>(), c->d_over.as<uint32_t>(), c->d_pend.as<uint32_t>(),
                                                     c->d_pend_buf.as<uint32_t>(), hard_cap, pend_cap, d_cnt, tasks_per_locus, gt_l0,
                                                     gt_n, c->d_gtables.as<uint8_t>(), gt_bytes, c->d_over.as<uint32_t>() + a0, 0, s));
                    HIP_TRY(c, hipEventRecord(c->ev[5], s));
                    HIP_TRY(c, hipMemcpyAsync(cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost, s));
                    HIP_TRY(c, hipStreamSynchronize(s));
                    const uint32_t again = cnt[1] - a1;          // <= a1 - a0: source and destination do not overlap
                    if (again) HIP_TRY(c, hipMemcpyAsync(c->d_over.as<uint32_t>() + a0, c->d_over.as<uint32_t>() + a1, (size_t)again * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
                    cnt[1] = a0 + again;
                    HIP_TRY(c, hipMemcpyAsync(d_cnt + 1, cnt + 1, sizeof(uint32_t), hipMemcpyHostToDevice, s));
                    ++launches;
                }
                over_before = cnt[1];
                {
                    float ms = 0;
                    if (!run_skipped) {
                        HIP_TRY(c, hipEventElapsedTime(&ms, swept ? c->ev[10] : c->ev[4], c->ev[5]));
                        band_run_ms += ms;
                        if (int rc = collect_sweep_times()) return rc;
                    }
                    if (diag) { HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[4], c->ev[6])); diag_ms += ms; }
                }
                if (!sweep_used && base + chunk >= t_end && cnt[1])      // last chunk: the overflow list is complete
                    if (int rc = fallback_start(0, cnt[1])) return rc;
                if (cnt[0] > hard_cap) {               // the excess went to the general kernel's list: slots in use = hard_cap
                    cnt[0] = hard_cap;
                    HIP_TRY(c, hipMemcpyAsync(d_cnt, cnt, sizeof(uint32_t), hipMemcpyHostToDevice, s));
                }
                cnt[11] = std::min(cnt[11], pend_cap);
                if (cnt[11]) {
                    // tasks whose piece list overflowed its LDS slots: the same certificate, from their pending records
                    HIP_TRY(c, vtxk_launch_band_pending(c->d_pend.as<uint32_t>(), cnt[11], c->d_pend_buf.as<uint32_t>(),
                                                        c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(), c->d_poly.as<uint16_t>(),
                                                        poly_stride / 2, c->d_hard.as<uint32_t>(), d_cnt, s));
                    HIP_TRY(c, hipMemcpyAsync(cnt, d_cnt, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                    HIP_TRY(c, hipStreamSynchronize(s));
                    pending_total += cnt[11];
                    ++launches;
                }
                if (int rc = masked_dp(cnt[0], c->d_hard.as<uint32_t>(), c->d_poly.as<uint16_t>(), c->d_band.as<uint16_t>(), slots, s, VTX_STAGE_RUN_DP)) return rc;
                hard_total += cnt[0];
                ++launches;
                if (sweep_used && base + chunk >= t_end) {
                    // last chunk.  What overflowed band_run_kernel's lists (d_over[0, nA)) takes band_sweep_kernel too; then everything
                    // the sweep declined — here and in the chunks' own sweeps: d_over[2 n_tasks, + nB), counted on the device — takes the
                    // general band kernel (round 4's kernel, libvtx_dev.so: first its second pass, [2 n_tasks + nB, + nC) is what is left).
                    uint32_t* over = c->d_over.as<uint32_t>();
                    const uint32_t nA = cnt[1];
                    if (nA) {
                        // (these tasks left band_diag_kernel for another reason than their number of matches, and then overflowed
                        // band_run_kernel's piece lists: repeats as well — on real-sequence loci 0.9 M tasks.  The second stage first, when
                        // the tables of every locus are still resident.)
                        const uint32_t* sl = over;
                        uint32_t n_sweep = nA;
                        if (second_stage_on(nA) && gt_bytes && !gt_chunked && nA <= chunk && c->d_dense.cap >= (size_t)nA * sizeof(uint32_t)) {
                            if (int rc = second_stage(over, nA, 0, c->d_dense.as<uint32_t>(), &n_sweep)) return rc;
                            sl = c->d_dense.as<uint32_t>();
                        }
                        if (n_sweep) { if (int rc = sweep_slices(0, sl, n_sweep, over + 2 * n_tasks, d_cnt + 26)) return rc; }
                        swept_total += n_sweep;
                    }
                    uint32_t nB = 0, nC = 0;
                    HIP_TRY(c, hipMemcpyAsync(&nB, d_cnt + 27, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                    HIP_TRY(c, hipStreamSynchronize(s));
                    if (int rc = collect_sweep_times()) return rc;
                    if (nB && sweep_v1) {                       // (round 4's kernel only: its second pass with the larger log)
                        resweep_total = nB;
                        if (int rc = sweep_slices(1, over + 2 * n_tasks, nB, over + 2 * n_tasks + nB, d_cnt + 28)) return rc;
                        HIP_TRY(c, hipMemcpyAsync(&nC, d_cnt + 29, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                        HIP_TRY(c, hipStreamSynchronize(s));
                    } else if (nB) {                            // what band_sweep_kernel declines (bytes outside ACGTN, > 255 bases, > 1 024 sections, a
                        nC = nB; nB = 0;                        // full stash bucket) takes the general band kernel
                    }
                    cnt[1] = nC;
                    if (nC) { if (int rc = fallback_start((uint32_t)(2 * n_tasks) + nB, (uint32_t)(2 * n_tasks) + nB + nC)) return rc; }
                }
            }
            fast_overflow = cnt[1];
            if (getenv("VTX_DEBUG")) {
                uint32_t why[11];
                HIP_TRY(c, hipMemcpy(why, d_cnt, sizeof why, hipMemcpyDeviceToHost));
                fprintf(stderr, "[vtx] band_run_kernel: %u tasks hard only because pieces were dropped from a full list\n", why[10]);
                fprintf(stderr, "[vtx] band_run_kernel overflow reasons: bound=%u parked-full=%u log-full=%u other=%u traceback=%u\n", why[3], why[4], why[5], why[6], why[7]);
            }
            if (int rc = fallback_finish()) return rc;
            c->fast_overflow += fast_overflow;
            c->timing.diag_ms += diag_ms; c->timing.diag_left += (uint32_t)std::min<uint64_t>(diag_left, 0xffffffffull);
            c->timing.check_ms += check_ms; c->timing.sweep_ms += sweep_ms;
            c->timing.swept_tasks += (uint32_t)std::min<uint64_t>(swept_total, 0xffffffffull);
            c->timing.resweep_tasks += resweep_total;
            c->timing.diag2_tasks += (uint32_t)std::min<uint64_t>(diag2_total, 0xffffffffull);
            c->timing.diag2_scored += (uint32_t)std::min<uint64_t>(diag2_scored, 0xffffffffull);
            c->timing.diag2_streamed += (uint32_t)std::min<uint64_t>(stream_total, 0xffffffffull);
            checked_total += tight2_total;                     // (one-diagonal bands of the second stage: the same masked DP)
            c->timing.checked_tasks += (uint32_t)std::min<uint64_t>(checked_total, 0xffffffffull);
            if (swept_total) hard_total += (uint32_t)std::min<uint64_t>(swept_total - std::min<uint64_t>(swept_total, fast_overflow), 0xffffffffull);
            hard_total += (uint32_t)std::min<uint64_t>(checked_total, 0xffffffffull);      // (tasks with a certificate: masked DP over their diagonal band; with VTX_BAND_CHECK an upper bound)
            if (getenv("VTX_DEBUG") && diag_total) {
                uint32_t why[16];
                HIP_TRY(c, hipMemcpy(why, d_cnt + 32, sizeof why, hipMemcpyDeviceToHost));
                fprintf(stderr, "[vtx] band_refine_kernel: %llu tasks listed\n", (unsigned long long)refined_total);
                fprintf(stderr, "[vtx] band_diag_kernel: %llu of %llu tasks left to band_run_kernel (%.2f %%), %.2f ms: shape=%u no-diagonal=%u pieces=%u matches=%u not-harmless=%u generic=%u not-tight=%u no-main=%u\n",
                        (unsigned long long)diag_left, (unsigned long long)diag_total, 100.0 * (double)diag_left / (double)diag_total, (double)diag_ms,
                        why[1], why[2], why[3], why[4], why[5], why[7], why[8], why[9]);
            }
            if (getenv("VTX_DEBUG") && diag2_total)
                fprintf(stderr, "[vtx] band_diag2_kernel: %llu tasks looked at, %llu scored, %llu left with a one-diagonal band, %llu to band_sweep_kernel (%llu through band_stream_kernel)\n",
                        (unsigned long long)diag2_total, (unsigned long long)diag2_scored, (unsigned long long)tight2_total,
                        (unsigned long long)(diag2_total - diag2_scored - tight2_total), (unsigned long long)stream_total);
            if (getenv("VTX_DEBUG")) fprintf(stderr, "[vtx] banded: %llu tasks, %u overflowed band_run_kernel, %u bounded by the pending kernel, %u hard\n", (unsigned long long)n_tasks, fast_overflow, pending_total, hard_total);
            return VTX_OK;
        };
        c->fast_overflow = 0;
        c->timing.diag_ms = c->timing.check_ms = c->timing.sweep_ms = 0;
        c->timing.diag_left = c->timing.checked_tasks = c->timing.swept_tasks = c->timing.resweep_tasks = 0;
        c->timing.diag2_tasks = c->timing.diag2_scored = c->timing.diag2_streamed = 0;
        // Shape per task (round 6).  A haplotype above 255 bases (a long deletion or insertion, a larger --padding) does not fit the
        // two-byte match entries of band_diag_kernel<., uint16_t> nor band_sweep_kernel's 256 columns; ONE such locus used to put the
        // whole batch on round 3's path (config 3: 18.9 instead of 15.9 ms; repeat-rich loci lose the sweep and the second stage, 2x).
        // When they are few, two passes: every locus up to 255 bases as if the others were not there, then the stretch of tasks from
        // the first to the last long locus with round 3's kernels, the short loci skipped (measured, config 3 + 1 long locus: 16.45 ms;
        // + 20 spread over the batch: 17.6 ms — the second pass walks the whole stretch: 0.6 ms of skipping, 1 ms of launches).
        // (Many long loci — a batch at --padding 150 — : one pass as before; two would walk the batch twice for nothing.)
        static const bool no_split = VTX_DEV_ENV("VTX_BAND_NO_SPLIT") != nullptr;          // test hook: the single pass of rounds 3 - 5
        const bool split = !no_split && c->max_hap_len > 255 && c->hap_short_max > 0 && c->n_long_loci > 0 &&
                           (uint64_t)c->n_long_loci * 8 <= c->n_loci;
        if (!split) {
            if (int rc = band_pass(c->max_hap_len, 0, 0, 2ull * nr, false)) return rc;
        } else {
            if (int rc = band_pass(c->hap_short_max, 0, 0, 2ull * nr, false)) return rc;
            vtx_locus ends[2];
            HIP_TRY(c, hipMemcpyAsync(&ends[0], c->d_loci.as<vtx_locus>() + c->long_first, sizeof(vtx_locus), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipMemcpyAsync(&ends[1], c->d_loci.as<vtx_locus>() + c->long_last, sizeof(vtx_locus), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            const uint64_t t0 = 2ull * std::min(ends[0].rec_begin, nr), t1 = 2ull * std::min<uint64_t>((uint64_t)ends[1].rec_begin + ends[1].rec_count, nr);
            if (t1 > t0) { if (int rc = band_pass(c->max_hap_len, 255, t0, t1, true)) return rc; }
        }
    }
    if (c->slow_cnt) {
        // records beyond the fast kernels' limits: exact slow path, both flavours (slabs grow until every chain fits)
        const uint32_t n_slow = 2 * c->slow_cnt;
        const int banded = c->cfg.aligner == VTX_ALIGNER_BANDED;
        if (stage) HIP_TRY(c, vtxk_mark_stage_records(c->d_work.as<uint32_t>() + c->slow_off, c->slow_cnt, VTX_STAGE_SLOW, stage, s));
        HIP_TRY(c, c->d_cnt.reserve(64 * sizeof(uint32_t)));
        uint32_t* d_scnt = c->d_cnt.as<uint32_t>() + 13;
        HIP_TRY(c, c->d_slow_retry.reserve(2 * (size_t)n_slow * sizeof(uint32_t)));
        const uint32_t* tasks = nullptr;
        uint32_t todo = n_slow, cap = 1024;
        const uint32_t mh = std::max(c->max_hap_all, 1u);
        for (;;) {
            const uint64_t worst = (uint64_t)c->max_read_all * mh;
            const size_t stride = vtxk_slow_ws_stride(cap, mh, c->max_read_all);
            HIP_TRY(c, c->d_slow_ws.reserve((size_t)todo * stride));
            HIP_TRY(c, hipMemsetAsync(d_scnt, 0, sizeof(uint32_t), s));
            uint32_t* retry = c->d_slow_retry.as<uint32_t>() + ((tasks == c->d_slow_retry.as<uint32_t>()) ? n_slow : 0);
            HIP_TRY(c, vtxk_launch_slow_align(c->d_work.as<uint32_t>() + c->slow_off, tasks, todo, banded, c->d_records.as<vtx_record>(),
                                              c->d_rec_locus.as<uint32_t>(), c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(),
                                              c->d_hap.as<uint8_t>(), c->d_slow_ws.as<uint8_t>(), stride, cap, mh, c->max_read_all,
                                              c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(), retry, d_scnt, s));
            uint32_t left = 0;
            HIP_TRY(c, hipMemcpyAsync(&left, d_scnt, sizeof left, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            ++launches;
            if (!left) break;
            if (cap >= worst) return fail(c, VTX_E_STATE, "vtx_run: slow path overflow with a worst-case slab");
            cap = (uint32_t)std::min<uint64_t>((uint64_t)cap * 16, std::max<uint64_t>(worst, 1024));
            tasks = retry; todo = left;
        }
    }
    HIP_TRY(c, hipEventRecord(c->ev[1], s));
    uint32_t nnz32 = 0;
    const uint32_t ng = c->n_cell_groups, nu = c->n_umi_groups;
    if (nr) {
        const size_t tmp_bytes = vtxk_scan_temp_bytes(nr);
        HIP_TRY(c, hipMemsetAsync(c->d_cell_cnt.p, 0, 3 * (size_t)ng * sizeof(uint32_t), s));
        if (c->cfg.use_umi) {
            HIP_TRY(c, hipMemsetAsync(c->d_umi_cnt.p, 0, 3 * (size_t)nu * sizeof(uint32_t), s));
            HIP_TRY(c, vtxk_count_calls(c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(), nr, c->cfg.min_score,
                                        c->d_umi_scan.as<uint32_t>(), c->d_umi_cnt.as<uint32_t>(), s));
            HIP_TRY(c, vtxk_umi_collapse(c->d_umi_cnt.as<uint32_t>(), nu, c->d_umi_cellgrp.as<uint32_t>(), c->d_cell_cnt.as<uint32_t>(), s));
        } else {
            HIP_TRY(c, vtxk_count_calls(c->d_ref.as<int32_t>(), c->d_alt.as<int32_t>(), nr, c->cfg.min_score,
                                        c->d_cell_scan.as<uint32_t>(), c->d_cell_cnt.as<uint32_t>(), s));
        }
        HIP_TRY(c, vtxk_keep_flags(c->d_cell_cnt.as<uint32_t>(), ng, c->cfg.scoring_mode, c->d_keep.as<uint32_t>(), s));
        HIP_TRY(c, vtxk_inclusive_scan_u32(c->d_keep.as<uint32_t>(), c->d_keep_scan.as<uint32_t>(), ng, c->d_scan_tmp.p, tmp_bytes, s));
        HIP_TRY(c, vtxk_emit_coo(c->d_cell_cnt.as<uint32_t>(), ng, c->cfg.scoring_mode, c->d_keep.as<uint32_t>(),
                                 c->d_keep_scan.as<uint32_t>(), c->d_grp_row.as<uint32_t>(), c->d_grp_col.as<uint32_t>(),
                                 c->d_o_row.as<uint32_t>(), c->d_o_col.as<uint32_t>(), c->d_o_alt.as<uint32_t>(),
                                 c->d_o_ref.as<uint32_t>(), c->d_o_unk.as<uint32_t>(), c->d_o_val.as<double>(),
                                 c->d_o_refval.as<double>(), s));
        if (ng) HIP_TRY(c, hipMemcpyAsync(&nnz32, c->d_keep_scan.as<uint32_t>() + (ng - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipEventRecord(c->ev[2], s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->nnz = nnz32;
    float t01 = 0, t12 = 0, t03 = 0;
    HIP_TRY(c, hipEventElapsedTime(&t03, c->ev[0], c->ev[3]));
    c->timing.full_ms = t03;
    HIP_TRY(c, hipEventElapsedTime(&t01, c->ev[0], c->ev[1]));
    c->timing.band_ms = t01 - t03;
    HIP_TRY(c, hipEventElapsedTime(&t12, c->ev[1], c->ev[2]));
    c->timing.sw_ms = t01; c->timing.reduce_ms = t12; c->timing.total_ms = t01 + t12;
    c->timing.sw_launches = launches; c->timing.hard_tasks = hard_total;
    c->timing.band_run_ms = band_run_ms; c->timing.overflow_tasks = c->fast_overflow;
    c->ran = true;
    return VTX_OK;
}

int vtx_fetch_scores(vtx_ctx* c, int32_t* ref_score, int32_t* alt_score) {
    if (!c) return VTX_E_INVAL;
    if (!c->ran) return fail(c, VTX_E_STATE, "vtx_fetch_scores: no completed vtx_run");
    if (c->n_records && (!ref_score || !alt_score)) return fail(c, VTX_E_INVAL, "vtx_fetch_scores: null output");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (c->n_records) {
        HIP_TRY(c, hipMemcpy(ref_score, c->d_ref.p, (size_t)c->n_records * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(alt_score, c->d_alt.p, (size_t)c->n_records * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return VTX_OK;
}

int vtx_set_debug(vtx_ctx* c, int key, int64_t value) {
    if (!c) return VTX_E_INVAL;
    switch (key) {
        case VTX_DEBUG_STAGE_TRACE: c->stage_trace = value != 0; return VTX_OK;
        case VTX_DEBUG_POISON_SCORES: c->poison = value != 0; return VTX_OK;
        case VTX_DEBUG_POISON_VALUE: c->poison_value = (int32_t)value; return VTX_OK;
        default: return fail(c, VTX_E_INVAL, "vtx_set_debug: unknown key %d", key);
    }
}

int vtx_fetch_stage(vtx_ctx* c, uint8_t* stage) {
    if (!c) return VTX_E_INVAL;
    if (!c->ran) return fail(c, VTX_E_STATE, "vtx_fetch_stage: no completed vtx_run");
    if (!c->stage_trace || (c->n_records && c->d_stage.cap < 2 * (size_t)c->n_records))
        return fail(c, VTX_E_STATE, "vtx_fetch_stage: the last vtx_run was not traced (vtx_set_debug(ctx, VTX_DEBUG_STAGE_TRACE, 1) before it)");
    if (c->n_records && !stage) return fail(c, VTX_E_INVAL, "vtx_fetch_stage: null output");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (c->n_records) HIP_TRY(c, hipMemcpy(stage, c->d_stage.p, 2 * (size_t)c->n_records, hipMemcpyDeviceToHost));
    return VTX_OK;
}

int vtx_set_read_format(vtx_ctx* c, int format) {
    if (!c) return VTX_E_INVAL;
    if (format != VTX_READS_BYTES && format != VTX_READS_NIBBLES) return fail(c, VTX_E_INVAL, "vtx_set_read_format: unknown format %d", format);
    c->read_format = format;
    return VTX_OK;
}

int vtx_debug_tables(vtx_ctx* c, void* dst, uint64_t cap, uint64_t* bytes) {
    if (!c || !bytes) return VTX_E_INVAL;
    *bytes = 0;
    if (!c->submitted) return fail(c, VTX_E_STATE, "vtx_debug_tables: no batch submitted");
    if (cap && !dst) return fail(c, VTX_E_INVAL, "vtx_debug_tables: null destination");
    *bytes = c->gt_used;
    const size_t n = (size_t)std::min<uint64_t>(cap, c->gt_used);
    if (!n) return VTX_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(dst, c->d_gtables.p, n, hipMemcpyDeviceToHost));
    return VTX_OK;
}

int vtx_debug_bands(vtx_ctx* c, const uint32_t* tasks, uint32_t n_tasks, uint32_t stride, uint16_t* lo, uint16_t* hi, uint8_t* status) {
    if (!c) return VTX_E_INVAL;
    if (!c->submitted) return fail(c, VTX_E_STATE, "vtx_debug_bands: no batch submitted");
    if (!n_tasks) return VTX_OK;
    if (!tasks || !lo || !hi || !status) return fail(c, VTX_E_INVAL, "vtx_debug_bands: null array");
    if (stride < c->max_hap_all + 1) return fail(c, VTX_E_INVAL, "vtx_debug_bands: stride %u < longest haplotype + 1 (%u)", stride, c->max_hap_all + 1);
    for (uint32_t i = 0; i < n_tasks; ++i)
        if (tasks[i] >= 2ull * c->n_records) return fail(c, VTX_E_INVAL, "vtx_debug_bands: task %u out of range", tasks[i]);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    // scratch of its own (one task per launch slot; a debug call may be slow): tasks, hard list, overflow list, counters, bands
    DevBuf d_t, d_h, d_o, d_c, d_b, d_d;
    const uint32_t bs = (stride + 7u) & ~7u;
    auto done = [&](int rc) { d_t.release(); d_h.release(); d_o.release(); d_c.release(); d_b.release(); d_d.release(); return rc; };
#define DBG_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return done(fail(c, VTX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_))); } while (0)
    DBG_TRY(d_t.reserve((size_t)n_tasks * 4)); DBG_TRY(d_h.reserve((size_t)n_tasks * 4)); DBG_TRY(d_o.reserve((size_t)n_tasks * 4));
    DBG_TRY(d_c.reserve(64 * 4)); DBG_TRY(d_b.reserve((size_t)n_tasks * 2 * bs * sizeof(uint16_t)));
    const bool dbg_on = VTX_DEV_ENV("VTX_SWEEP_DBG") != nullptr;        // developer aid: the kernel's per-task intermediate state on stderr
    if (dbg_on) DBG_TRY(d_d.reserve((size_t)n_tasks * 64 * 4));
    DBG_TRY(hipMemcpyAsync(d_t.p, tasks, (size_t)n_tasks * 4, hipMemcpyHostToDevice, s));
    DBG_TRY(hipMemsetAsync(d_c.p, 0, 64 * 4, s));
    DBG_TRY(c->d_sweep_log.reserve(vtxk_band_sweep_log_bytes()));
#ifdef VTX_DEVTOOLS
    // (libvtx_dev.so: VTX_SWEEP_V1=1 asks round 4's kernel instead; VTX_SWEEP_TIER=1 its 1024-section variant)
    if (VTX_DEV_ENV("VTX_SWEEP_V1"))
        DBG_TRY(vtxk_launch_band_sweep_v1(VTX_DEV_ENV("VTX_SWEEP_TIER") ? atoi(VTX_DEV_ENV("VTX_SWEEP_TIER")) : 0, d_t.as<uint32_t>(), n_tasks, nullptr,
                                          c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(), c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(),
                                          c->d_hap.as<uint8_t>(), d_b.as<uint16_t>(), bs, d_h.as<uint32_t>(), d_o.as<uint32_t>(), d_c.as<uint32_t>(), nullptr,
                                          nullptr, dbg_on ? d_d.as<uint32_t>() : nullptr, s));
    else
#endif
    DBG_TRY(vtxk_launch_band_sweep(d_t.as<uint32_t>(), n_tasks, nullptr, c->d_records.as<vtx_record>(), c->d_rec_locus.as<uint32_t>(),
                                   c->d_loci.as<vtx_locus>(), c->d_read.as<uint8_t>(), c->d_hap.as<uint8_t>(), d_b.as<uint16_t>(), bs,
                                   d_h.as<uint32_t>(), d_o.as<uint32_t>(), d_c.as<uint32_t>(), nullptr, nullptr, dbg_on ? d_d.as<uint32_t>() : nullptr,
                                   c->d_sweep_log.as<uint32_t>(), s));
    uint32_t cnt[2] = {0, 0};
    DBG_TRY(hipMemcpyAsync(cnt, d_c.p, sizeof cnt, hipMemcpyDeviceToHost, s));
    DBG_TRY(hipStreamSynchronize(s));
    std::vector<uint32_t> hard(cnt[0]);
    std::vector<uint16_t> bands((size_t)cnt[0] * 2 * bs);
    if (cnt[0]) {
        DBG_TRY(hipMemcpy(hard.data(), d_h.p, (size_t)cnt[0] * 4, hipMemcpyDeviceToHost));
        DBG_TRY(hipMemcpy(bands.data(), d_b.p, bands.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
    }
    if (dbg_on) {
        std::vector<uint32_t> dd((size_t)n_tasks * 64);
        DBG_TRY(hipMemcpy(dd.data(), d_d.p, dd.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n_tasks; ++i) {
            const uint32_t* o = dd.data() + (size_t)i * 64;
            fprintf(stderr, "[sweep dbg] task %u: best dp %u x %u y %u, log %u, sections %u, cA %d cB %d, decline %u, m %u n %u\n  sections:", tasks[i],
                    o[0] >> 16, (o[0] >> 8) & 0xff, o[0] & 0xff, o[1], o[2], (int)o[3], (int)o[4], o[5], o[6], o[7]);
            for (uint32_t k = 0; k < o[2] && k < 12; ++k) fprintf(stderr, " (%u,%u,%u)", o[8 + k] >> 16, (o[8 + k] >> 8) & 0xff, o[8 + k] & 0xff);
            fprintf(stderr, "\n  log:");
            for (uint32_t k = 0; k < o[1] && k < 24; ++k) fprintf(stderr, " (%u,%u<-%04x)", o[20 + k] >> 24, (o[20 + k] >> 16) & 0xff, o[20 + k] & 0xffff);
            fprintf(stderr, "\n  rmin:");
            for (int k = 0; k < 10; ++k) fprintf(stderr, " %d", (int)o[44 + k]);
            fprintf(stderr, "  rmax:");
            for (int k = 0; k < 10; ++k) fprintf(stderr, " %d", (int)o[54 + k]);
            fprintf(stderr, "\n");
        }
    }
#undef DBG_TRY
    // slots come out in any order, and a task may be listed more than once: every occurrence of a task gets the band of one of its slots
    std::vector<std::pair<uint32_t, uint32_t>> by_task(cnt[0]);
    for (uint32_t h = 0; h < cnt[0]; ++h) by_task[h] = {hard[h], h};
    std::sort(by_task.begin(), by_task.end());
    for (uint32_t i = 0; i < n_tasks; ++i) {
        auto it = std::lower_bound(by_task.begin(), by_task.end(), std::make_pair(tasks[i], 0u));
        if (it == by_task.end() || it->first != tasks[i]) { status[i] = 1; continue; }
        status[i] = 0;
        const uint16_t* src = bands.data() + (size_t)it->second * 2 * bs;
        memcpy(lo + (size_t)i * stride, src, (size_t)stride * sizeof(uint16_t));
        memcpy(hi + (size_t)i * stride, src + bs, (size_t)stride * sizeof(uint16_t));
    }
    return done(VTX_OK);
}

int vtx_device_scores(vtx_ctx* c, const int32_t** d_ref, const int32_t** d_alt) {
    if (!c) return VTX_E_INVAL;
    if (!c->ran) return fail(c, VTX_E_STATE, "vtx_device_scores: no completed vtx_run");
    if (d_ref) *d_ref = c->d_ref.as<int32_t>();
    if (d_alt) *d_alt = c->d_alt.as<int32_t>();
    return VTX_OK;
}

int vtx_fetch_coo(vtx_ctx* c, vtx_coo* out) {
    if (!c) return VTX_E_INVAL;
    if (!out) return fail(c, VTX_E_INVAL, "vtx_fetch_coo: null output");
    if (!c->ran) return fail(c, VTX_E_STATE, "vtx_fetch_coo: no completed vtx_run");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return fetch_arrays(c, c->nnz, c->d_o_row.p, c->d_o_col.p, c->d_o_alt.p, c->d_o_ref.p, c->d_o_unk.p, c->d_o_val.p, c->d_o_refval.p, out);
}

// sprs::io::write_matrix_market of the last vtx_run's triplets (src/main.rs:381-389: three header lines, then "row+1 col+1 value"
// in insertion order), formatted ON THE DEVICE and streamed into the file by the copy workers: the triplets never become host arrays.
// which: 0 = `value` (the matrix), 1 = `ref_value` (coverage mode's ref matrix).  Only for integral values — consensus 1 / 2 / 3,
// coverage counts — whose Rust `{}` text is their digits; alt_frac's fractions / NaN need shortest-round-trip digits: VTX_E_UNSUPPORTED,
// nothing is left at `path`, the caller formats on the host (vtx_fetch_coo + vtxh_write_mtx).  *sum (optional) = the sum of the values
// (the reference's "matrix has a sum of 0" warning, :410-415).
int vtx_write_mtx(vtx_ctx* c, const char* path, uint32_t n_rows, uint32_t n_cols, int which, double* sum) {
    if (!c) return VTX_E_INVAL;
    if (!path || (which != 0 && which != 1)) return fail(c, VTX_E_INVAL, "vtx_write_mtx: bad argument");
    if (!c->ran) return fail(c, VTX_E_STATE, "vtx_write_mtx: no completed vtx_run");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    const uint64_t nnz = c->nnz;
    const uint32_t* d_row = c->d_o_row.as<uint32_t>();
    const uint32_t* d_col = c->d_o_col.as<uint32_t>();
    const double* d_val = which ? c->d_o_refval.as<double>() : c->d_o_val.as<double>();
    const uint32_t kSlab = 48u << 20;                     // lines per pass: <= 33 bytes each, 32-bit text offsets
    HIP_TRY(c, c->d_bam_cnt.reserve(VTXG_N_COUNTERS * sizeof(uint64_t) + 4 * sizeof(uint32_t)));
    double* d_sum = (double*)c->d_bam_cnt.p;
    uint32_t* d_flag = (uint32_t*)(c->d_bam_cnt.as<unsigned long long>() + VTXG_N_COUNTERS);
    HIP_TRY(c, hipMemsetAsync(c->d_bam_cnt.p, 0, VTXG_N_COUNTERS * sizeof(uint64_t) + 4 * sizeof(uint32_t), s));
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return fail(c, VTX_E_INVAL, "cannot open %s for writing", path);
    auto bail = [&](int rc) { close(fd); unlink(path); return rc; };
    char head[160];
    const int hl = snprintf(head, sizeof head, "%%%%MatrixMarket matrix coordinate real general\n%% written by sprs\n%u %u %llu\n", n_rows, n_cols,
                            (unsigned long long)nnz);
    if (pwrite(fd, head, (size_t)hl, 0) != hl) return bail(fail(c, VTX_E_INVAL, "error writing %s", path));
    uint64_t file_off = (uint64_t)hl;
    for (uint64_t base = 0; base < nnz; base += kSlab) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(kSlab, nnz - base);
        if (hipError_t e = c->d_bam_nhit.reserve((size_t)n * sizeof(uint32_t) + 16)) return bail(fail(c, VTX_E_NOMEM, "vtx_write_mtx: %s", hipGetErrorString(e)));
        if (hipError_t e = c->d_bam_hscan.reserve((size_t)n * sizeof(uint32_t) + 16)) return bail(fail(c, VTX_E_NOMEM, "vtx_write_mtx: %s", hipGetErrorString(e)));
        if (hipError_t e = c->d_scan_tmp.reserve(vtxk_scan_temp_bytes(n))) return bail(fail(c, VTX_E_NOMEM, "vtx_write_mtx: %s", hipGetErrorString(e)));
        uint32_t* d_len = c->d_bam_nhit.as<uint32_t>();
        uint32_t* d_end = c->d_bam_hscan.as<uint32_t>();
        hipError_t e = vtxg_mtx_len(d_row + base, d_col + base, d_val + base, n, d_len, d_sum, d_flag, s);
        if (e == hipSuccess) e = vtxk_inclusive_scan_u32(d_len, d_end, n, c->d_scan_tmp.p, vtxk_scan_temp_bytes(n), s);
        uint32_t total = 0, flag = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&total, d_end + (n - 1), sizeof total, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&flag, d_flag, sizeof flag, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return bail(fail(c, VTX_E_HIP, "vtx_write_mtx: %s", hipGetErrorString(e)));
        if (flag) return bail(fail(c, VTX_E_UNSUPPORTED, "vtx_write_mtx: a value that is not a non-negative integer (alt_frac): format on the host (vtx_fetch_coo + vtxh_write_mtx)"));
        if ((e = c->d_bam_data.reserve((size_t)total + 64)) != hipSuccess) return bail(fail(c, VTX_E_NOMEM, "vtx_write_mtx: %s", hipGetErrorString(e)));
        e = vtxg_mtx_text(d_row + base, d_col + base, d_val + base, n, d_end, c->d_bam_data.as<uint8_t>(), s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return bail(fail(c, VTX_E_HIP, "vtx_write_mtx: %s", hipGetErrorString(e)));
        if (int rc = download_to_fd(c, fd, file_off, c->d_bam_data.p, total)) return bail(rc);
        file_off += total;
    }
    if (sum) {
        *sum = 0.0;
        if (nnz) HIP_TRY(c, hipMemcpy(sum, d_sum, sizeof(double), hipMemcpyDeviceToHost));
    }
    if (close(fd) != 0) { unlink(path); return fail(c, VTX_E_INVAL, "error writing %s", path); }
    return VTX_OK;
}

int vtx_device_coo(vtx_ctx* c, vtx_coo* out) {
    if (!c) return VTX_E_INVAL;
    if (!out) return fail(c, VTX_E_INVAL, "vtx_device_coo: null output");
    if (!c->ran) return fail(c, VTX_E_STATE, "vtx_device_coo: no completed vtx_run");
    out->row = c->d_o_row.as<uint32_t>(); out->col = c->d_o_col.as<uint32_t>(); out->alt = c->d_o_alt.as<uint32_t>();
    out->ref = c->d_o_ref.as<uint32_t>(); out->unk = c->d_o_unk.as<uint32_t>(); out->value = c->d_o_val.as<double>();
    out->ref_value = c->d_o_refval.as<double>();
    out->nnz = c->nnz;
    return VTX_OK;
}

int vtx_comm_id(uint8_t id[VTX_COMM_ID_BYTES]) {
    static_assert(VTX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    if (!id) return fail(nullptr, VTX_E_INVAL, "vtx_comm_id: null argument");
    if (!rccl()) return fail(nullptr, VTX_E_UNSUPPORTED, "vtx_comm_id: librccl.so not found");
    ncclUniqueId u;
    NCCL_TRY(nullptr, rccl()->GetUniqueId(&u));
    memcpy(id, u.internal, VTX_COMM_ID_BYTES);
    return VTX_OK;
}

int vtx_comm_init(vtx_ctx* c, const uint8_t id[VTX_COMM_ID_BYTES], int rank, int world) {
    if (!c) return VTX_E_INVAL;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(c, VTX_E_INVAL, "vtx_comm_init: bad rank %d / world %d", rank, world);
    if (!rccl()) return fail(c, VTX_E_UNSUPPORTED, "vtx_comm_init: librccl.so not found");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    comm_release(c);
    ncclUniqueId u;
    memcpy(u.internal, id, VTX_COMM_ID_BYTES);
    NCCL_TRY(c, rccl()->CommInitRank(&c->comm, world, u, rank));
    c->comm_rank = rank; c->comm_world = world;
    return VTX_OK;
}

// How many ranks the communicator itself reports (ncclCommCount): what bench.py's line quotes next to the launcher's world size.
int vtx_comm_ranks(vtx_ctx* c, int* ranks) {
    if (!c || !ranks) return VTX_E_INVAL;
    if (!c->comm) return fail(c, VTX_E_STATE, "vtx_comm_ranks: no communicator (vtx_comm_init)");
    if (!rccl() || !rccl()->CommCount) return fail(c, VTX_E_UNSUPPORTED, "vtx_comm_ranks: ncclCommCount not available");
    NCCL_TRY(c, rccl()->CommCount(c->comm, ranks));
    return VTX_OK;
}

// The exchange's plan as a pure function of the counts (no device, no communicator): rank r's block lands at
// offsets[r] of the gathered arrays (rank order = row order), *total triplets in all.  VTX_E_UNSUPPORTED when the gathered
// arrays would not fit 32-bit indices.  Every rank computes the same plan from the same all-gathered counts, so every rank
// takes the same decision.
int vtx_gather_plan(int world, const uint64_t* counts, uint64_t* offsets, uint64_t* total) {
    if (world < 1 || !counts || !offsets || !total) return VTX_E_INVAL;
    uint64_t t = 0;
    for (int r = 0; r < world; ++r) { offsets[r] = t; t += counts[r]; }
    *total = t;
    return t > 0xffffffffull ? VTX_E_UNSUPPORTED : VTX_OK;
}

// One all-gather of (count, status) per rank: a rank that cannot take part in the data exchange (its own vtx_run failed,
// the destination could not reserve its buffers) says so HERE, and every rank leaves with VTX_E_PEER before any
// point-to-point call — a Send whose Recv never comes would block for ever.
static int gather_agree(vtx_ctx* c, uint64_t mine, uint64_t status, std::vector<uint64_t>& cnt, std::vector<uint64_t>& st) {
    const int world = c->comm_world;
    hipStream_t s = c->stream;
    Rccl* R = rccl();
    HIP_TRY(c, c->d_g_cnt.reserve(2 * ((size_t)world + 1) * sizeof(uint64_t)));
    uint64_t* d_cnt = c->d_g_cnt.as<uint64_t>();
    const uint64_t pair[2] = {mine, status};
    HIP_TRY(c, hipMemcpyAsync(d_cnt + 2 * world, pair, sizeof pair, hipMemcpyHostToDevice, s));
    NCCL_TRY(c, R->AllGather(d_cnt + 2 * world, d_cnt, 2, ncclUint64, c->comm, s));
    std::vector<uint64_t> both(2 * (size_t)world);
    HIP_TRY(c, hipMemcpyAsync(both.data(), d_cnt, both.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    cnt.resize((size_t)world); st.resize((size_t)world);
    for (int r = 0; r < world; ++r) { cnt[(size_t)r] = both[2 * (size_t)r]; st[(size_t)r] = both[2 * (size_t)r + 1]; }
    return VTX_OK;
}

int vtx_gather_abort(vtx_ctx* c) {
    if (!c) return VTX_E_INVAL;
    if (!c->comm) return fail(c, VTX_E_STATE, "vtx_gather_abort: no communicator (vtx_comm_init)");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    std::vector<uint64_t> cnt, st;
    if (int rc = gather_agree(c, 0, 1, cnt, st)) return rc;
    return VTX_OK;
}

int vtx_gather_coo(vtx_ctx* c, int dst, vtx_coo* out) {
    if (!c) return VTX_E_INVAL;
    if (!out) return fail(c, VTX_E_INVAL, "vtx_gather_coo: null output");
    if (!c->comm) return fail(c, VTX_E_STATE, "vtx_gather_coo: no communicator (vtx_comm_init)");
    const int world = c->comm_world, rank = c->comm_rank;
    if (dst < 0 || dst >= world) return fail(c, VTX_E_INVAL, "vtx_gather_coo: dst %d out of range", dst);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    Rccl* R = rccl();
    memset(out, 0, sizeof *out);
    c->g_nnz = 0;
    // round 1: counts + "my vtx_run completed"
    const uint64_t mine = c->ran ? c->nnz : 0;
    std::vector<uint64_t> cnt, st;
    if (int rc = gather_agree(c, mine, c->ran ? 0 : 1, cnt, st)) return rc;
    for (int r = 0; r < world; ++r)
        if (st[(size_t)r]) return fail(c, r == rank ? VTX_E_STATE : VTX_E_PEER, r == rank ? "vtx_gather_coo: no completed vtx_run"
                                       : "vtx_gather_coo: rank %d reported an error; exchange abandoned", r);
    std::vector<uint64_t> off((size_t)world);
    uint64_t total = 0;
    if (vtx_gather_plan(world, cnt.data(), off.data(), &total) != VTX_OK)      // (the same verdict on every rank)
        return fail(c, VTX_E_UNSUPPORTED, "vtx_gather_coo: more than 2^32 gathered triplets");
    // round 2: the destination has its buffers (the only step before the exchange that can fail on one rank alone)
    DevBuf* dstb[5] = {&c->d_g_row, &c->d_g_col, &c->d_g_alt, &c->d_g_ref, &c->d_g_unk};
    uint64_t ready = 0;
    if (rank == dst) {
        const size_t cap = (size_t)std::max<uint64_t>(total, 1);
        for (DevBuf* b : dstb) if (b->reserve(cap * sizeof(uint32_t)) != hipSuccess) ready = 1;
        if (c->d_g_val.reserve(cap * sizeof(double)) != hipSuccess || c->d_g_refval.reserve(cap * sizeof(double)) != hipSuccess) ready = 1;
        if (ready) (void)hipGetLastError();
    }
    {
        std::vector<uint64_t> c2, s2;
        if (int rc = gather_agree(c, 0, ready, c2, s2)) return rc;
        if (s2[(size_t)dst]) return fail(c, rank == dst ? VTX_E_NOMEM : VTX_E_PEER, "vtx_gather_coo: rank %d could not reserve the gathered arrays", dst);
    }
    const uint32_t* src[5] = {c->d_o_row.as<uint32_t>(), c->d_o_col.as<uint32_t>(), c->d_o_alt.as<uint32_t>(),
                              c->d_o_ref.as<uint32_t>(), c->d_o_unk.as<uint32_t>()};
    if (rank != dst) {
        if (mine) {
            NCCL_TRY(c, R->GroupStart());
            for (int f = 0; f < 5; ++f) NCCL_TRY(c, R->Send(src[f], (size_t)mine, ncclUint32, dst, c->comm, s));
            NCCL_TRY(c, R->GroupEnd());
        }
        HIP_TRY(c, hipStreamSynchronize(s));          // the source arrays may be overwritten by the next vtx_run
        return VTX_OK;
    }
    NCCL_TRY(c, R->GroupStart());
    for (int r = 0; r < world; ++r) {
        if (r == dst || !cnt[(size_t)r]) continue;
        for (int f = 0; f < 5; ++f)
            NCCL_TRY(c, R->Recv(dstb[f]->as<uint32_t>() + off[(size_t)r], (size_t)cnt[(size_t)r], ncclUint32, r, c->comm, s));
    }
    NCCL_TRY(c, R->GroupEnd());
    if (mine)
        for (int f = 0; f < 5; ++f)
            HIP_TRY(c, hipMemcpyAsync(dstb[f]->as<uint32_t>() + off[(size_t)dst], src[f], (size_t)mine * sizeof(uint32_t),
                                      hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, vtxk_values_from_counts(c->d_g_alt.as<uint32_t>(), c->d_g_ref.as<uint32_t>(), c->d_g_unk.as<uint32_t>(), (uint32_t)total,
                                       c->cfg.scoring_mode, c->d_g_val.as<double>(), c->d_g_refval.as<double>(), s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->g_nnz = total;
    out->row = c->d_g_row.as<uint32_t>(); out->col = c->d_g_col.as<uint32_t>(); out->alt = c->d_g_alt.as<uint32_t>();
    out->ref = c->d_g_ref.as<uint32_t>(); out->unk = c->d_g_unk.as<uint32_t>(); out->value = c->d_g_val.as<double>();
    out->ref_value = c->d_g_refval.as<double>();
    out->nnz = total;
    return VTX_OK;
}

int vtx_fetch_gathered(vtx_ctx* c, vtx_coo* out) {
    if (!c) return VTX_E_INVAL;
    if (!out) return fail(c, VTX_E_INVAL, "vtx_fetch_gathered: null output");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return fetch_arrays(c, (size_t)c->g_nnz, c->d_g_row.p, c->d_g_col.p, c->d_g_alt.p, c->d_g_ref.p, c->d_g_unk.p, c->d_g_val.p, c->d_g_refval.p, out);
}

int vtx_last_timing(vtx_ctx* c, vtx_timing* out) {
    if (!c || !out) return VTX_E_INVAL;
    if (!c->ran) return fail(c, VTX_E_STATE, "vtx_last_timing: no completed vtx_run");
    *out = c->timing;
    return VTX_OK;
}

int vtx_last_cells(vtx_ctx* c, uint64_t* cells) {
    if (!c || !cells) return VTX_E_INVAL;
    if (!c->ran) return fail(c, VTX_E_STATE, "vtx_last_cells: no completed vtx_run");
    *cells = c->cells;
    return VTX_OK;
}

}  // extern "C"
