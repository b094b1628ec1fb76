// vartrix — drop-in command line of the reference (same flags, same outputs), hot path on MI355X.
//
// Restates _main (reference src/main.rs:163-418): argument surface of get_args
// (:40-135), check_inputs_exist / validate_output_path (:475-542), then
// ingest + pack (libvtxhost) -> vtx_submit / vtx_run / vtx_fetch_coo (libvtx,
// replacing the rayon map :279-291 and the merge loop :320-348) -> .mtx and the
// optional variants / barcodes files.  `--threads` only sizes the BGZF inflate
// pool here (it never affected results in the reference either, :279-291).
// New, optional: --devices N (shard loci over N GPUs), --aligner banded|full (default banded = the
// reference's banded::Aligner, src/main.rs:899; full = unbanded Smith-Waterman).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/vtx.h"
#include "../../../include/vtx_host.h"

namespace {

int g_log_level = 0;   // 0 error, 1 info, 2 debug  (--log-level, :102-106)

void logmsg(int level, const char* tag, const char* fmt, ...) {
    if (level > g_log_level) return;
    char ts[16];
    time_t t = time(nullptr);
    strftime(ts, sizeof ts, "%H:%M:%S", gmtime(&t));
    fprintf(stderr, "%s [%s] ", ts, tag);
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
#define LOG_ERROR(...) logmsg(0, "ERROR", __VA_ARGS__)
#define LOG_INFO(...) logmsg(1, "INFO", __VA_ARGS__)
#define LOG_DEBUG(...) logmsg(2, "DEBUG", __VA_ARGS__)

bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }

// validate_output_path, :475-491
void validate_output_path(const std::string& p) {
    if (exists(p)) { LOG_ERROR("Output path already exists"); exit(1); }
    size_t slash = p.find_last_of('/');
    std::string parent = slash == std::string::npos ? "" : p.substr(0, slash);
    if (!parent.empty() && !exists(parent)) { LOG_ERROR("Output directory \"%s\" does not exist", parent.c_str()); exit(1); }
}

struct Opt { const char* name; char shrt; bool flag; const char* def; };
const Opt kOpts[] = {
    {"vcf", 'v', false, nullptr}, {"bam", 'b', false, nullptr}, {"fasta", 'f', false, nullptr},
    {"cell-barcodes", 'c', false, nullptr}, {"out-matrix", 'o', false, "out_matrix.mtx"},
    {"out-variants", 0, false, nullptr}, {"out-barcodes", 0, false, nullptr}, {"padding", 'p', false, "100"},
    {"scoring-method", 's', false, "consensus"}, {"ref-matrix", 0, false, "ref_matrix.mtx"},
    {"log-level", 0, false, "error"}, {"threads", 0, false, "1"}, {"mapq", 0, false, "0"},
    {"primary-alignments", 0, true, nullptr}, {"no-duplicates", 0, true, nullptr}, {"umi", 0, true, nullptr},
    {"bam-tag", 0, false, "CB"}, {"valid-chars", 0, false, "ATGCatgc"},
    {"devices", 0, false, "1"}, {"aligner", 0, false, "banded"}, {"prep", 0, false, "host"},
    {"stream-loci", 0, false, "auto"}, {"reads", 0, false, "nibbles"}, {"gather", 0, false, "auto"}, {"ingest", 0, false, "auto"},
};

void usage() {
    fprintf(stderr,
            "vartrix (MI355X-native hot path)\nUSAGE: vartrix --vcf <FILE> --bam <FILE> --fasta <FILE> --cell-barcodes <FILE> [OPTIONS]\n"
            "  -o, --out-matrix <FILE> [out_matrix.mtx]   --out-variants <FILE>   --out-barcodes <FILE>\n"
            "  -p, --padding <INT> [100]   -s, --scoring-method consensus|coverage|alt_frac [consensus]\n"
            "  --ref-matrix <FILE> [ref_matrix.mtx]   --log-level info|debug|error [error]   --threads <INT> [1]\n"
            "  --mapq <INT> [0]   --primary-alignments   --no-duplicates   --umi   --bam-tag <TAG> [CB]\n"
            "  --valid-chars <CHARS> [ATGCatgc]   --devices <INT> [1]   --aligner banded|full [banded]\n"
            "  --ingest auto|device|host [auto]  where the BAM is read: device = BGZF inflate, record split, read filters and tag lookups on the\n"
            "                               GPU (vtx_submit_bam; needs the .bai; implies --prep device for that range); host = the packer\n"
            "                               threads of libvtxhost; auto = device with one GPU when the input allows it, else host\n"
            "  --prep host|device [host]  (device: barcode lookup, UMI grouping and the sort run on the GPU)\n"
            "  --stream-loci <INT>|auto [auto]  VCF records per streamed range (ingest of range k + 1 overlaps the device work on\n"
            "                               range k; host memory follows the range, not the BAM); 0 = the whole input at once;\n"
            "                               auto = at once when the BAM is below 4 GiB (faster: one sweep), ranges of 32768 above\n"
            "  --reads nibbles|bytes [nibbles]  read bases on their way to the device: two per byte as the BAM holds them (the device\n"
            "                               unpacks), or one ASCII byte per base\n"
            "  --gather auto|library [auto]  how the shards' rows meet: auto = through the library's RCCL gather when --devices > 1;\n"
            "                               library = through it even with one device (the same bytes either way)\n");
}

// The shard threads of one batch meet here before each RCCL collective, carrying their status: if any shard has failed,
// NONE enters the collective (a communicator a rank never joins, a Send whose Recv never comes, would block for ever).
struct ShardGate {
    std::mutex mu;
    std::condition_variable cv;
    int world = 1, arrived = 0, generation = 0;
    bool failed = false, verdict = false;
    // returns true when every shard arrived with ok == true
    bool meet(bool ok) {
        std::unique_lock<std::mutex> lk(mu);
        if (!ok) failed = true;
        const int gen = generation;
        if (++arrived == world) { verdict = !failed; arrived = 0; failed = false; ++generation; cv.notify_all(); return verdict; }
        cv.wait(lk, [&] { return generation != gen; });
        return verdict;
    }
};

struct Shard {
    std::vector<vtx_locus> loci;                 // rec_begin rebased to the shard's first record
    const vtx_record* records = nullptr;         // the shard's slice of the pack's arrays (not copied)
    uint32_t n_records = 0;
    const uint8_t *haps, *reads;
    uint64_t hap_bytes, read_bytes;
    // --prep device: raw records (tags as bytes) + the barcode table
    bool raw = false;
    const vtx_raw_record* raw_records = nullptr;
    const uint8_t *tags = nullptr, *bc_bytes = nullptr;
    const uint64_t* bc_offsets = nullptr;
    uint64_t tag_bytes = 0;
    uint32_t n_bcs = 0;
    vtx_raw_stats stats{};
    std::vector<uint32_t> row, col;
    std::vector<double> val, refval;
    // a run of one batch on one shard writes the matrices straight from the context's arrays (ctx kept until exit)
    bool keep_ctx = false;
    vtx_coo kept{};
    vtx_ctx* ctx_pre = nullptr;                 // a context created at launch (its vtx_prefetch_file is bringing the BAM's bytes): used instead of vtx_create
    vtx_ctx* ctx_kept = nullptr;                // keep_ctx with defer_fetch: the triplets stay on the device — vtx_write_mtx formats them there
    bool defer_fetch = false;
    std::string err;
    int rc = 0;
    double t_create = 0, t_submit = 0, t_run = 0, t_fetch = 0;
    // --devices N > 1: the shards' rows meet on device 0 through the library's RCCL gather (vtx_gather_coo)
    const uint8_t* comm_id = nullptr;
    int rank = 0, world = 1;
    ShardGate* gate = nullptr;
    int read_format = VTX_READS_BYTES;          // of the pack's read arenas (vtxh_read_format)
    // --ingest device: the range's reads never exist on the host — the device gets the file's bytes and the plan (vtx_submit_bam)
    const vtx_bam_ingest* ingest = nullptr;
    vtx_ingest_stats istats{};
    bool declined = false;                      // vtx_submit_bam said VTX_E_UNSUPPORTED: the caller packs this range on the host
};

double since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void run_shard(Shard* s, vtx_config cfg) {
    vtx_ctx* ctx = s->ctx_pre;
    double t0 = now_s();
    if (!ctx) {
        s->rc = vtx_create(&cfg, &ctx);
        if (s->rc) s->err = vtx_strerror(nullptr);
    }
    if (s->comm_id) {
        // every shard has a context, or none joins the communicator
        if (!s->gate->meet(s->rc == 0)) {
            if (!s->rc) { s->rc = VTX_E_PEER; s->err = "another shard failed before the communicator was set up"; vtx_destroy(ctx); }
            return;
        }
        s->rc = vtx_comm_init(ctx, s->comm_id, s->rank, s->world);             // collective over the shard threads
        if (s->rc) s->err = vtx_strerror(ctx);
        // ... and every shard has joined it, or none goes on: a shard whose vtx_comm_init failed has no communicator to call
        // vtx_gather_abort on — the others would wait for ever in vtx_gather_coo's all-gather (round-3 ADVICE)
        if (!s->gate->meet(s->rc == 0)) {
            if (!s->rc) { s->rc = VTX_E_PEER; s->err = "another shard failed to join the communicator"; }
            vtx_destroy(ctx);
            return;
        }
    } else if (s->rc) return;
    s->t_create = now_s() - t0; t0 = now_s();
    if ((s->rc = vtx_set_read_format(ctx, s->read_format))) { s->err = vtx_strerror(ctx); if (s->comm_id) (void)vtx_gather_abort(ctx); vtx_destroy(ctx); return; }
    vtx_batch b{s->loci.data(), (uint32_t)s->loci.size(), s->records, s->n_records, s->haps,
                s->hap_bytes, s->reads, s->read_bytes};
    vtx_coo coo{};
    if (s->ingest) {
        if ((s->rc = vtx_set_barcodes(ctx, s->bc_bytes, s->bc_offsets, s->n_bcs)) || (s->rc = vtx_submit_bam(ctx, s->ingest, &s->istats))) {
            s->err = vtx_strerror(ctx);
            s->declined = s->rc == VTX_E_UNSUPPORTED || s->rc == VTX_E_NOMEM;      // (the inflated stream did not fit the device: the host packer needs far less of it)
            vtx_destroy(ctx);
            return;
        }
        s->stats = s->istats.raw;
    } else if (s->raw) {
        vtx_raw_batch rb{s->loci.data(), (uint32_t)s->loci.size(), s->raw_records, s->n_records,
                         s->haps, s->hap_bytes, s->reads, s->read_bytes, s->tags, s->tag_bytes};
        if ((s->rc = vtx_set_barcodes(ctx, s->bc_bytes, s->bc_offsets, s->n_bcs)) || (s->rc = vtx_submit_raw(ctx, &rb, &s->stats))) {
            s->err = vtx_strerror(ctx);
            if (s->comm_id) (void)vtx_gather_abort(ctx);      // the other shards leave their vtx_gather_coo with VTX_E_PEER
            vtx_destroy(ctx);
            return;
        }
    }
    if (!s->raw && !s->ingest && (s->rc = vtx_submit(ctx, &b))) { s->err = vtx_strerror(ctx); if (s->comm_id) (void)vtx_gather_abort(ctx); vtx_destroy(ctx); return; }
    s->t_submit = now_s() - t0; t0 = now_s();
    if ((s->rc = vtx_run(ctx))) { s->err = vtx_strerror(ctx); if (s->comm_id) (void)vtx_gather_abort(ctx); vtx_destroy(ctx); return; }
    s->t_run = now_s() - t0; t0 = now_s();
    if (s->comm_id) {
        // (no host gate here: a shard that failed after joining the communicator says so INSIDE the collective — vtx_gather_abort
        // above takes part in the status round, every vtx_gather_coo returns VTX_E_PEER)
        // every shard's triplets to rank 0 over RCCL (rank order = row order); rank 0 copies the gathered matrix out
        vtx_coo dev{};
        if ((s->rc = vtx_gather_coo(ctx, 0, &dev))) { s->err = vtx_strerror(ctx); vtx_destroy(ctx); return; }
        if (s->rank != 0) { s->t_fetch = now_s() - t0; vtx_destroy(ctx); return; }
        if ((s->rc = vtx_fetch_gathered(ctx, &coo))) { s->err = vtx_strerror(ctx); vtx_destroy(ctx); return; }
    } else if (s->keep_ctx && s->defer_fetch) { s->ctx_kept = ctx; return; }
    else if ((s->rc = vtx_fetch_coo(ctx, &coo))) { s->err = vtx_strerror(ctx); vtx_destroy(ctx); return; }
    if (s->keep_ctx) { s->kept = coo; s->ctx_kept = ctx; s->t_fetch = now_s() - t0; return; }
    s->row.assign(coo.row, coo.row + coo.nnz);
    s->col.assign(coo.col, coo.col + coo.nnz);
    s->val.assign(coo.value, coo.value + coo.nnz);
    s->refval.assign(coo.ref_value, coo.ref_value + coo.nnz);
    s->t_fetch = now_s() - t0;
    vtx_destroy(ctx);
}

// HIP runtime + device context initialisation costs ~0.1 s per process: do it while the host ingests
void warm_device(int device) {
    vtx_config cfg;
    vtx_config_default(&cfg);
    cfg.device = device;
    vtx_ctx* ctx = nullptr;
    if (vtx_create(&cfg, &ctx) == VTX_OK) vtx_destroy(ctx);
}

}  // namespace

int main(int argc, char** argv) {
    const auto t_main = std::chrono::steady_clock::now();
    std::map<std::string, std::string> val;
    std::map<std::string, bool> present;
    for (const Opt& o : kOpts) if (o.def) val[o.name] = o.def;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-h" || a == "--help") { usage(); return 0; }
        const Opt* opt = nullptr;
        std::string inline_val;
        bool has_inline = false;
        if (a.rfind("--", 0) == 0) {
            size_t eq = a.find('=');
            std::string name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            if (eq != std::string::npos) { inline_val = a.substr(eq + 1); has_inline = true; }
            for (const Opt& o : kOpts) if (name == o.name) opt = &o;
        } else if (a.size() >= 2 && a[0] == '-') {
            for (const Opt& o : kOpts) if (o.shrt && a[1] == o.shrt) opt = &o;
            if (opt && a.size() > 2) { inline_val = a.substr(2); has_inline = true; }
        }
        if (!opt) { fprintf(stderr, "error: Found argument '%s' which wasn't expected\n", a.c_str()); usage(); return 1; }
        present[opt->name] = true;
        if (opt->flag) continue;
        if (has_inline) val[opt->name] = inline_val;
        else if (i + 1 < argc) val[opt->name] = argv[++i];
        else { fprintf(stderr, "error: The argument '--%s <value>' requires a value\n", opt->name); return 1; }
    }
    for (const char* req : {"vcf", "bam", "fasta", "cell-barcodes"})
        if (!val.count(req)) { fprintf(stderr, "error: The following required arguments were not provided:\n    --%s <FILE>\n", req); usage(); return 1; }
    const std::string ll = val["log-level"];
    if (ll == "info") g_log_level = 1; else if (ll == "debug") g_log_level = 2;
    else if (ll != "error") { printf("Log level not valid\n"); return 1; }                 // :202-205
    const std::string mode = val["scoring-method"];
    if (mode != "consensus" && mode != "coverage" && mode != "alt_frac") {
        fprintf(stderr, "error: '%s' isn't a valid value for '--scoring-method <scoring_method>'\n", mode.c_str());
        return 1;
    }
    const std::string out_matrix = val["out-matrix"], ref_matrix = val["ref-matrix"];

    // check_inputs_exist, :493-542
    for (const char* k : {"fasta", "vcf", "bam", "cell-barcodes"})
        if (!exists(val[k])) { LOG_ERROR("Input file %s does not exist", val[k].c_str()); return 1; }
    validate_output_path(out_matrix);
    validate_output_path(ref_matrix);     // the default ref_matrix.mtx too, even in consensus mode (:509-511)
    if (!exists(val["fasta"] + ".fai")) { LOG_ERROR("File %s.fai does not exist", val["fasta"].c_str()); return 1; }
    {
        const std::string& bam = val["bam"];
        size_t dot = bam.find_last_of('.');
        std::string ext = dot == std::string::npos ? "" : bam.substr(dot + 1);
        if (ext == "bam") {
            if (!exists(bam + ".bai") && !exists(bam + ".csi")) { LOG_ERROR("BAM index does not exist. Expecting %s.bai or %s.csi", bam.c_str(), bam.c_str()); return 1; }
        } else if (ext == "cram") {
            LOG_ERROR("CRAM input is not supported by this build"); return 1;
        } else { LOG_ERROR("BAM file did not end in .bam or .cram. Unable to validate"); return 1; }
    }

    vtxh_args ha{};
    ha.vcf = val["vcf"].c_str(); ha.bam = val["bam"].c_str(); ha.fasta = val["fasta"].c_str();
    ha.cell_barcodes = val["cell-barcodes"].c_str();
    ha.padding = (uint32_t)strtoul(val["padding"].c_str(), nullptr, 10);
    ha.mapq = (uint32_t)strtoul(val["mapq"].c_str(), nullptr, 10);
    ha.primary_only = present.count("primary-alignments"); ha.no_duplicates = present.count("no-duplicates");
    ha.use_umi = present.count("umi");
    ha.bam_tag = val["bam-tag"].c_str(); ha.valid_chars = val["valid-chars"].c_str();
    ha.threads = std::max(1, atoi(val["threads"].c_str()));
    if (val["reads"] != "nibbles" && val["reads"] != "bytes") { fprintf(stderr, "error: --reads takes nibbles or bytes, not `%s`\n", val["reads"].c_str()); return 1; }
    ha.read_format = val["reads"] == "nibbles" ? VTX_READS_NIBBLES : VTX_READS_BYTES;
    const auto t_start = std::chrono::steady_clock::now();
    if (val["prep"] != "host" && val["prep"] != "device") {
        fprintf(stderr, "error: '%s' isn't a valid value for '--prep <prep>'\n", val["prep"].c_str());
        return 1;
    }
    const bool raw = val["prep"] == "device";
    if (val["gather"] != "auto" && val["gather"] != "library") {
        fprintf(stderr, "error: '%s' isn't a valid value for '--gather <gather>'\n", val["gather"].c_str());
        return 1;
    }
    const int ndev = std::max(1, atoi(val["devices"].c_str()));
    if (val["ingest"] != "auto" && val["ingest"] != "device" && val["ingest"] != "host") {
        fprintf(stderr, "error: '%s' isn't a valid value for '--ingest <ingest>'\n", val["ingest"].c_str());
        return 1;
    }
    if (val["ingest"] == "device" && ndev > 1) { fprintf(stderr, "error: --ingest device runs on one GPU (use --ingest host with --devices %d)\n", ndev); return 1; }
    // the plan of a device-side ingest instead of a pack: with one GPU, unless the host was asked for
    const bool try_device_ingest = val["ingest"] != "host" && ndev == 1 && val["gather"] != "library";
    const bool must_device_ingest = val["ingest"] == "device";
    vtx_config cfg;
    vtx_config_default(&cfg);
    if (val["aligner"] != "banded" && val["aligner"] != "full") {
        fprintf(stderr, "error: '%s' isn't a valid value for '--aligner <aligner>'\n", val["aligner"].c_str());
        return 1;
    }
    cfg.aligner = val["aligner"] == "banded" ? VTX_ALIGNER_BANDED : VTX_ALIGNER_FULL;
    cfg.scoring_mode = mode == "consensus" ? VTX_MODE_CONSENSUS : (mode == "alt_frac" ? VTX_MODE_ALT_FRAC : VTX_MODE_COVERAGE);
    cfg.use_umi = ha.use_umi;
    cfg.n_barcodes = 0;                                    // (known once the barcode file is read: vtx_set_barcodes fills it in)
    // ---- device ingest of the whole input at once: the context is created NOW and the BAM's bytes start travelling to the device
    //      (vtx_prefetch_file) while this thread still reads the VCF, builds the haplotypes and walks the BGZF headers ----
    bool whole_input_at_once = false;
    {
        struct stat st;
        whole_input_at_once = (val["stream-loci"] == "0" || (val["stream-loci"] == "auto" && stat(val["bam"].c_str(), &st) == 0 && (uint64_t)st.st_size <= (4ull << 30)));
    }
    vtx_ctx* early_ctx = nullptr;
    std::thread early;
    if (try_device_ingest && whole_input_at_once)
        early = std::thread([&, cfg0 = cfg] {
            if (vtx_create(&cfg0, &early_ctx) != VTX_OK) { early_ctx = nullptr; return; }
            (void)vtx_prefetch_file(early_ctx, val["bam"].c_str(), 0, 0);      // best effort: vtx_submit_bam uploads itself otherwise
        });
    // (an error return from main must not leave the library's prefetch thread copying while the HIP runtime is torn down: the context
    //  that nobody took over is destroyed — after the early thread has been joined: destructors run in reverse order)
    struct EarlyCtx { vtx_ctx*& c; ~EarlyCtx() { if (c) { vtx_destroy(c); c = nullptr; } } } early_ctx_guard{early_ctx};
    struct EarlyJoin { std::thread& t; ~EarlyJoin() { if (t.joinable()) t.join(); } } early_join{early};
    std::vector<std::thread> warm;
    for (int d = (early.joinable() ? 1 : 0); d < ndev; ++d) warm.emplace_back(warm_device, d);
    struct JoinAll { std::vector<std::thread>& v; ~JoinAll() { for (auto& t : v) if (t.joinable()) t.join(); } } join_warm{warm};
    // ---- streaming: the VCF records are taken in ranges of --stream-loci rows.  A producer thread packs range k + 1 (BGZF
    //      inflate, filters, haplotypes: the host-bound 70 % of a run) while this thread drives the device through range k and
    //      collects its triplets; at most two ranges are in memory.  The reference itself holds one locus' reads at a time
    //      (src/main.rs:822-830); results do not depend on the ranges (tests/test_host.py, tests/test_gpu_cli.py). ----
    uint32_t stream_loci = 0;
    if (val["stream-loci"] == "auto") {
        // measured at config-3 scale (profiles/r03_e2e_cli_*.log): one sweep of a 0.9 GB BAM 2.0 s, four streamed ranges 3.1 s — the
        // packer's threads scale worse on a third of the data; streaming is for inputs whose reads do not fit the host
        struct stat st;
        if (stat(val["bam"].c_str(), &st) == 0 && (uint64_t)st.st_size > (4ull << 30)) stream_loci = 32768;
    } else {
        char* endp = nullptr;
        const unsigned long v = strtoul(val["stream-loci"].c_str(), &endp, 10);
        if (val["stream-loci"].empty() || *endp != '\0' || v > 0xffffffffull) {
            fprintf(stderr, "error: --stream-loci takes a number of VCF records or `auto`, not `%s`\n", val["stream-loci"].c_str());
            return 1;
        }
        stream_loci = (uint32_t)v;                            // (0: one range)
    }
    struct Packed { vtxh_pack* pk = nullptr; int rc = 0; std::string err; double secs = 0; bool last = false; uint32_t begin = 0, end = 0; std::string why_host; };
    std::mutex q_mu;
    std::condition_variable q_cv;
    std::vector<Packed> q;                 // at most one packed range waiting (plus the one being consumed)
    bool consumer_gone = false;
    std::thread producer([&] {
        uint32_t begin = 0, n_total = 0xffffffffu;
        for (;;) {
            Packed pc;
            const auto t0 = std::chrono::steady_clock::now();
            const uint32_t end = stream_loci ? (begin + stream_loci < begin ? 0xffffffffu : begin + stream_loci) : 0xffffffffu;
            pc.begin = begin; pc.end = end;
            if (try_device_ingest) {
                pc.rc = vtxh_plan_ingest(&ha, begin, end, &pc.pk);
                vtx_bam_ingest probe;
                if (!pc.rc && vtxh_get_ingest(pc.pk, &probe) != VTX_OK) {        // no plan for this input: pack on the host (here, beside the device)
                    pc.why_host = vtxh_last_error();
                    vtxh_free(pc.pk); pc.pk = nullptr;
                    if (must_device_ingest) { pc.rc = VTX_E_UNSUPPORTED; pc.err = pc.why_host; }
                    else pc.rc = vtxh_pack_files_range(&ha, raw ? 1 : 0, begin, end, &pc.pk);
                }
            } else pc.rc = vtxh_pack_files_range(&ha, raw ? 1 : 0, begin, end, &pc.pk);
            if (pc.rc) { if (pc.err.empty()) pc.err = vtxh_last_error(); }
            else n_total = vtxh_num_variants(pc.pk);
            pc.secs = since(t0);
            pc.last = pc.rc != 0 || end >= n_total;
            const bool last = pc.last;
            {
                std::unique_lock<std::mutex> lk(q_mu);
                q_cv.wait(lk, [&] { return q.empty() || consumer_gone; });
                if (consumer_gone) { if (pc.pk) vtxh_free(pc.pk); return; }
                q.push_back(std::move(pc));
            }
            q_cv.notify_all();
            if (last) return;
            begin = end;
        }
    });
    struct ProducerJoin { std::thread& t; std::mutex& mu; std::condition_variable& cv; bool& gone;
                          ~ProducerJoin() { { std::lock_guard<std::mutex> lk(mu); gone = true; } cv.notify_all(); if (t.joinable()) t.join(); } }
        producer_join{producer, q_mu, q_cv, consumer_gone};
    auto next_pack = [&]() -> Packed {
        std::unique_lock<std::mutex> lk(q_mu);
        q_cv.wait(lk, [&] { return !q.empty(); });
        Packed pc = std::move(q.front());
        q.erase(q.begin());
        lk.unlock();
        q_cv.notify_all();
        return pc;
    };
    Packed first = next_pack();
    if (first.rc) { printf("Vartrix error.\nError: %s\n", first.err.c_str()); return 1; }
    vtxh_pack* pk = first.pk;
    const uint32_t n_vars = vtxh_num_variants(pk), n_bcs = vtxh_num_barcodes(pk);
    // names for --out-variants / --out-barcodes outlive the packs
    std::vector<std::string> variant_names, barcode_names;
    if (present.count("out-variants")) for (uint32_t i = 0; i < n_vars; ++i) variant_names.emplace_back(vtxh_variant_name(pk, i));
    if (present.count("out-barcodes")) for (uint32_t j = 0; j < n_bcs; ++j) barcode_names.emplace_back(vtxh_barcode(pk, j));
    double t_ingest = first.secs;
    LOG_INFO("Loaded %u barcodes", n_bcs);
    if (n_vars == 0)
        LOG_ERROR("Warning! Zero variants found in input VCF. Output matrices will be by definition empty but will still be generated.");
    LOG_INFO("Initialized a %u variants x %u cell barcodes matrix", n_vars, n_bcs);

    // ---- the hot path: every batch of the pack (one unless the reads span > 4 GiB) is sharded over --devices GPUs
    //      (contiguous row ranges by record count); triplets are appended in batch order = row order ----
    cfg.n_barcodes = n_bcs;
    const uint8_t* bc_bytes = nullptr;
    const uint64_t* bc_offsets = nullptr;
    uint32_t bc_n = 0;
    const auto t_wait = std::chrono::steady_clock::now();
    for (auto& t : warm) t.join();
    LOG_INFO("Waited %.3f s more for the HIP runtime / device initialisation started at launch", since(t_wait));
    std::vector<uint32_t> row, col;
    uint64_t out_nnz = 0;                                  // the matrix to write: either the vectors or one kept context's arrays
    const uint32_t *out_row = nullptr, *out_col = nullptr;
    const double *out_v = nullptr, *out_rv = nullptr;
    std::vector<double> v, rv;
    vtx_ctx* kept_ctx = nullptr;                           // one range, one batch, one device: the matrix is written from the device (vtx_write_mtx)
    vtx_raw_stats raw_total{};
    vtxh_metrics m{};
    double t_device = 0;
    Packed cur = std::move(first);
    for (uint32_t range_idx = 0;; ++range_idx) {
    pk = cur.pk;
    if (!cur.why_host.empty()) LOG_INFO("Range %u: %s — packing on the host", range_idx, cur.why_host.c_str());
    if (vtxh_is_plan(pk)) {
        // ---- --ingest device: the BAM's bytes and the plan go to the device; the reads never exist on the host ----
        vtx_bam_ingest g;
        (void)vtxh_get_ingest(pk, &g);
        vtxh_get_barcode_table(pk, &bc_bytes, &bc_offsets, &bc_n);
        Shard s;
        s.ingest = &g; s.bc_bytes = bc_bytes; s.bc_offsets = bc_offsets; s.n_bcs = bc_n; s.raw = true;
        s.keep_ctx = range_idx == 0 && cur.last;
        s.defer_fetch = s.keep_ctx && mode != "alt_frac";
        if (early.joinable()) early.join();
        if (early_ctx) { s.ctx_pre = early_ctx; early_ctx = nullptr; }
        const auto t_shard = std::chrono::steady_clock::now();
        LOG_INFO("Plan of range %u: %.3f s (%u loci, %u BGZF blocks, %u record-start seeds from the .bai); ingest on the device", range_idx, cur.secs,
                 g.n_loci, g.n_blocks, g.n_seeds);
        run_shard(&s, cfg);
        if (s.rc && s.declined && !must_device_ingest) {
            LOG_INFO("Range %u: the device declined (%s) — packing on the host", range_idx, s.err.c_str());
            vtxh_pack* hp = nullptr;
            const auto t_hp = std::chrono::steady_clock::now();
            if (vtxh_pack_files_range(&ha, raw ? 1 : 0, cur.begin, cur.end, &hp)) { printf("Vartrix error.\nError: %s\n", vtxh_last_error()); return 1; }
            vtxh_free(pk);
            pk = cur.pk = hp;
            cur.secs += since(t_hp);
        } else {
            if (s.rc) { printf("Vartrix error.\nError: %s: %s\n", vtx_status_name(s.rc), s.err.c_str()); return 1; }
            t_device += since(t_shard);
            const vtx_ingest_stats& is = s.istats;
            LOG_INFO("  device ingest: %llu BAM records, %llu (read, locus) pairs; upload %.1f ms (%.1f MB compressed; prefetch %.1f ms, waited %.1f ms for it), inflate %.1f ms (%.1f MB), record index %.1f ms, filters %.1f ms",
                     (unsigned long long)is.bam_records, (unsigned long long)is.raw_records, (double)is.h2d_ms, is.compressed_bytes / 1e6, (double)is.prefetch_ms,
                     (double)is.prefetch_wait_ms, (double)is.inflate_ms, is.inflated_bytes / 1e6, (double)is.index_ms, (double)is.filter_ms);
            LOG_INFO("  shard: create %.3f s, submit (upload + ingest + device preparation) %.3f s, run %.3f s, fetch %.3f s", s.t_create, s.t_submit, s.t_run, s.t_fetch);
            LOG_INFO("  device preparation: %llu reads kept, %.3f ms, %u hash round(s)", (unsigned long long)s.stats.kept, (double)s.stats.prep_ms, s.stats.hash_rounds);
            if (s.keep_ctx && s.ctx_kept && s.defer_fetch) kept_ctx = s.ctx_kept;
            else if (s.keep_ctx) { out_nnz = s.kept.nnz; out_row = s.kept.row; out_col = s.kept.col; out_v = s.kept.value; out_rv = s.kept.ref_value; }
            row.insert(row.end(), s.row.begin(), s.row.end());
            col.insert(col.end(), s.col.begin(), s.col.end());
            v.insert(v.end(), s.val.begin(), s.val.end());
            rv.insert(rv.end(), s.refval.begin(), s.refval.end());
            m.num_reads += is.num_reads; m.num_low_mapq += is.num_low_mapq; m.num_non_primary += is.num_non_primary;
            m.num_duplicates += is.num_duplicates; m.num_not_useful += is.num_not_useful; m.num_not_cell_bc += is.num_no_barcode_tag;
            raw_total.num_not_cell_bc += s.stats.num_not_cell_bc; raw_total.num_non_umi += s.stats.num_non_umi; raw_total.kept += s.stats.kept;
        }
    }
    if (!vtxh_is_plan(pk) || vtxh_num_batches(pk)) {         // the host packed this range: the early context (and the bytes it prefetched) go
        if (early.joinable()) early.join();
        if (early_ctx) { vtx_destroy(early_ctx); early_ctx = nullptr; }
    }
    const bool range_raw = raw && !vtxh_is_plan(pk);             // (a plan that ran has no batches: the loop below is empty)
    if (raw && !vtxh_is_plan(pk)) vtxh_get_barcode_table(pk, &bc_bytes, &bc_offsets, &bc_n);
    (void)range_raw;
    const uint32_t n_batches = vtxh_is_plan(pk) ? 0u : vtxh_num_batches(pk);
    for (uint32_t bi = 0; bi < n_batches; ++bi) {
        vtx_batch full{};
        vtx_raw_batch full_raw{};
        if (raw) {
            vtxh_get_raw_batch_at(pk, bi, &full_raw);
            full.loci = full_raw.loci; full.n_loci = full_raw.n_loci; full.n_records = full_raw.n_records;
            full.hap_arena = full_raw.hap_arena; full.hap_bytes = full_raw.hap_bytes;
            full.read_arena = full_raw.read_arena; full.read_bytes = full_raw.read_bytes;
        } else {
            vtxh_get_batch_at(pk, bi, &full);
        }
        std::vector<Shard> shards((size_t)ndev);
        {
            const uint64_t total = full.n_records;
            std::vector<uint32_t> cuts((size_t)ndev + 1, full.n_loci);
            cuts[0] = 0;
            uint64_t acc = 0;
            uint32_t l = 0;
            for (int d = 1; d < ndev; ++d) {
                const uint64_t target = total * (uint64_t)d / (uint64_t)ndev;
                while (l < full.n_loci && acc < target) acc += full.loci[l++].rec_count;
                cuts[(size_t)d] = l;
            }
            for (int d = 0; d < ndev; ++d) {
                Shard& s = shards[(size_t)d];
                s.loci.assign(full.loci + cuts[(size_t)d], full.loci + cuts[(size_t)d + 1]);
                const uint32_t r0 = s.loci.empty() ? 0 : s.loci.front().rec_begin;
                const uint32_t r1 = s.loci.empty() ? 0 : s.loci.back().rec_begin + s.loci.back().rec_count;
                s.n_records = r1 - r0;
                s.keep_ctx = n_batches == 1 && ndev == 1 && range_idx == 0 && cur.last;
                s.defer_fetch = s.keep_ctx && mode != "alt_frac" && val["gather"] != "library";
                if (raw) {
                    s.raw = true;
                    s.raw_records = full_raw.records + r0;
                    s.tags = full_raw.tag_arena; s.tag_bytes = full_raw.tag_bytes;
                    s.bc_bytes = bc_bytes; s.bc_offsets = bc_offsets; s.n_bcs = bc_n;
                } else {
                    s.records = full.records + r0;
                }
                for (auto& L : s.loci) L.rec_begin -= r0;
                s.haps = full.hap_arena; s.hap_bytes = full.hap_bytes;       // arenas are shared read-only, offsets stay valid
                s.reads = full.read_arena; s.read_bytes = full.read_bytes; s.read_format = vtxh_read_format(pk);
            }
        }
        if (bi == 0)
            LOG_INFO("Ingest + filter + pack of range %u: %.3f s (%u batch(es); first: %u loci, %u %s)", range_idx, cur.secs, n_batches, full.n_loci, full.n_records,
                     raw ? "raw reads; barcode lookup / UMI grouping / sort on the device" : "scored reads");
        // more than one device (or the test hook): the row exchange runs behind the C-ABI over RCCL
        uint8_t comm_id[VTX_COMM_ID_BYTES];
        ShardGate gate;
        gate.world = ndev;
        const bool use_comm = ndev > 1 || val["gather"] == "library";
        if (use_comm) {
            if (int rc = vtx_comm_id(comm_id)) { printf("Vartrix error.\nError: %s: %s\n", vtx_status_name(rc), vtx_strerror(nullptr)); return 1; }
            for (int d = 0; d < ndev; ++d) { shards[(size_t)d].comm_id = comm_id; shards[(size_t)d].rank = d; shards[(size_t)d].world = ndev; shards[(size_t)d].gate = &gate; }
        }
        std::vector<std::thread> th;
        const auto t_shards = std::chrono::steady_clock::now();
        for (int d = 0; d < ndev; ++d) {
            vtx_config c = cfg;
            c.device = d;
            th.emplace_back(run_shard, &shards[(size_t)d], c);
        }
        for (auto& t : th) t.join();
        t_device += since(t_shards);                      // (the shards' work only: not the packer, not the release of a range's reads)
        for (int pass = 0; pass < 2; ++pass)              // the shard that failed on its own first, not the VTX_E_PEER followers
            for (auto& s : shards)
                if (s.rc && (pass == 1 || s.rc != VTX_E_PEER)) { printf("Vartrix error.\nError: %s: %s\n", vtx_status_name(s.rc), s.err.c_str()); return 1; }
        for (auto& s : shards) {
            LOG_INFO("  batch %u shard: create %.3f s, submit (H2D%s) %.3f s, run %.3f s, fetch %.3f s", bi, s.t_create,
                     s.raw ? " + device preparation" : "", s.t_submit, s.t_run, s.t_fetch);
            if (s.keep_ctx && s.ctx_kept && s.defer_fetch) kept_ctx = s.ctx_kept;
            else if (s.keep_ctx) { out_nnz = s.kept.nnz; out_row = s.kept.row; out_col = s.kept.col; out_v = s.kept.value; out_rv = s.kept.ref_value; }
            row.insert(row.end(), s.row.begin(), s.row.end());       // shard (= row) order: the triplet order of the merge loop :320-348
            col.insert(col.end(), s.col.begin(), s.col.end());
            v.insert(v.end(), s.val.begin(), s.val.end());
            rv.insert(rv.end(), s.refval.begin(), s.refval.end());
            if (s.raw) {
                raw_total.num_not_cell_bc += s.stats.num_not_cell_bc;      // the in-list test ran on the device (:870-876)
                raw_total.num_non_umi += s.stats.num_non_umi;              // :879-888
                raw_total.kept += s.stats.kept;
                LOG_INFO("  device preparation: %llu reads kept, %.3f ms, %u hash round(s)", (unsigned long long)s.stats.kept,
                         (double)s.stats.prep_ms, s.stats.hash_rounds);
            }
        }
    }
    {
        vtxh_metrics mr;
        vtxh_get_metrics(pk, &mr);
        m.num_reads += mr.num_reads; m.num_low_mapq += mr.num_low_mapq; m.num_non_primary += mr.num_non_primary;
        m.num_duplicates += mr.num_duplicates; m.num_not_cell_bc += mr.num_not_cell_bc; m.num_not_useful += mr.num_not_useful;
        m.num_non_umi += mr.num_non_umi; m.num_invalid_recs += mr.num_invalid_recs; m.num_multiallelic_recs += mr.num_multiallelic_recs;
    }
    vtxh_free(pk);                                         // this range's reads leave the host before the next range is consumed
    pk = nullptr;
    if (cur.last) break;
    cur = next_pack();
    if (cur.rc) { printf("Vartrix error.\nError: %s\n", cur.err.c_str()); return 1; }
    t_ingest += cur.secs;
    }
    LOG_INFO("Ingest + filter + pack, all ranges: %.3f s of packer time (overlapped with the device work on the range before)", t_ingest);
    LOG_INFO("Device (create + submit + run + fetch) on %d GPU(s): %.3f s", ndev, t_device);
    const auto t_out = std::chrono::steady_clock::now();
    m.num_not_cell_bc += raw_total.num_not_cell_bc;
    m.num_non_umi += raw_total.num_non_umi;
    LOG_INFO("Number of alignments evaluated: %llu", (unsigned long long)m.num_reads);                                        // :350-379
    LOG_INFO("Number of alignments skipped due to low mapping quality: %llu", (unsigned long long)m.num_low_mapq);
    LOG_INFO("Number of alignments skipped due to not being primary: %llu", (unsigned long long)m.num_non_primary);
    LOG_INFO("Number of alignments skipped due to being duplicates: %llu", (unsigned long long)m.num_duplicates);
    LOG_INFO("Number of alignments skipped due to not being associated with a cell barcode: %llu", (unsigned long long)m.num_not_cell_bc);
    LOG_INFO("Number of alignments skipped due to not intersecting variant: %llu", (unsigned long long)m.num_not_useful);
    LOG_INFO("Number of alignments skipped due to not having a UMI: %llu", (unsigned long long)m.num_non_umi);
    LOG_INFO("Number of VCF records skipped due to having invalid characters in the alternative haplotype: %llu", (unsigned long long)m.num_invalid_recs);
    LOG_INFO("Number of VCF records skipped due to being multi-allelic: %llu", (unsigned long long)m.num_multiallelic_recs);

    double sum = 0;
    bool written = false;
    if (kept_ctx) {
        // the matrix straight from the device: Matrix-Market text formatted there, streamed into the file by the copy workers
        double s0 = 0;
        int rc = vtx_write_mtx(kept_ctx, out_matrix.c_str(), n_vars, n_bcs, 0, &s0);
        if (rc == VTX_OK && mode == "coverage") rc = vtx_write_mtx(kept_ctx, ref_matrix.c_str(), n_vars, n_bcs, 1, nullptr);    // :385-389 (see below)
        if (rc == VTX_OK) { written = true; sum = s0; }
        else if (rc != VTX_E_UNSUPPORTED) { printf("Vartrix error.\nError: Error writing out-matrix\nInfo: caused by %s\n", vtx_strerror(kept_ctx)); return 1; }
        else {
            vtx_coo coo{};
            if (vtx_fetch_coo(kept_ctx, &coo)) { printf("Vartrix error.\nError: %s\n", vtx_strerror(kept_ctx)); return 1; }
            out_nnz = coo.nnz; out_row = coo.row; out_col = coo.col; out_v = coo.value; out_rv = coo.ref_value;
        }
    }
    if (!written) {
    if (!out_row) { out_nnz = row.size(); out_row = row.data(); out_col = col.data(); out_v = v.data(); out_rv = rv.data(); }
    if (vtxh_write_mtx(out_matrix.c_str(), n_vars, n_bcs, out_nnz, out_row, out_col, out_v) != 0) {
        printf("Vartrix error.\nError: Error writing out-matrix\nInfo: caused by %s\n", vtxh_last_error());
        return 1;
    }
    // :385-389 `args.is_present("ref_matrix")` is true even without the flag: clap 2.33 reports an argument with a
    // default_value (:100) as present, so coverage mode always writes the REF-count matrix (default ref_matrix.mtx)
    if (mode == "coverage") {
        if (vtxh_write_mtx(ref_matrix.c_str(), n_vars, n_bcs, out_nnz, out_row, out_col, out_rv) != 0) {
            printf("Vartrix error.\nError: Error writing ref-matrix\nInfo: caused by %s\n", vtxh_last_error());
            return 1;
        }
    }
    for (uint64_t k = 0; k < out_nnz; ++k) sum += out_v[k];
    }
    if (present.count("out-variants")) {                                                                                     // :391-398
        validate_output_path(val["out-variants"]);
        FILE* f = fopen(val["out-variants"].c_str(), "wb");
        if (!f) { printf("Vartrix error.\nError: error writing variants file\n"); return 1; }
        for (uint32_t i = 0; i < n_vars; ++i) fprintf(f, "%s\n", variant_names[i].c_str());
        fclose(f);
    }
    if (present.count("out-barcodes")) {                                                                                     // :400-407
        validate_output_path(val["out-barcodes"]);
        FILE* f = fopen(val["out-barcodes"].c_str(), "wb");
        if (!f) { printf("Vartrix error.\nError: error writing barcodes file\n"); return 1; }
        for (uint32_t j = 0; j < n_bcs; ++j) fprintf(f, "%s\n", barcode_names[j].c_str());
        fclose(f);
    }
    LOG_INFO("Merge + output files: %.3f s", since(t_out));
    if (sum == 0.0) LOG_ERROR("The resulting matrix has a sum of 0. Did you use the --umi flag on data without UMIs?");       // :410-415
    LOG_INFO("Total since launch: %.3f s", since(t_main));
    // every output file is closed: skip the teardown of a GB of host arrays and of the HIP runtime (~0.1 s)
    fflush(nullptr);
    if (getenv("VTXH_PROFILE")) exit(0);        // (a profiler attached to the process — rocprofv3 — writes its files from an exit handler)
    _exit(0);
}
