// repaq_hip — host driver with repaq's command line (src/main.cpp:29-51, README.md:136-161) over the C-ABI of
// include/rfq_hip.h.  It is the counterpart of Repaq::compress / compressPE / decompress / decompressPE / compare*
// (src/repaq.cpp): file I/O, batching with carry-over, header-once, line-break thresholds, PE even/odd outputs and the
// compare JSON live here; every byte of codec work happens on the GPU.  Not supported in this build (refused loudly):
// .gz text and the .xz wrapper (external zlib / xz, out of scope per SURVEY.md §2).
#include "rfq_hip.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static void error_exit(const std::string& msg) { fprintf(stderr, "ERROR: %s\n", msg.c_str()); exit(-1); }   // src/util.h:246-249
static bool ends_with(const std::string& s, const std::string& e) { return s.size() >= e.size() && s.compare(s.size() - e.size(), e.size(), e) == 0; }

struct Options {
    std::string in1, out1, in2, out2, rfqCompare, json;
    long chunkKb = 1000; bool compress = false, decompress = false, compare = false, useStdin = false, useStdout = false, interleaved = false;
    int device = 0; size_t batchBytes = (size_t)1 << 30;
};

static bool read_all(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = path == "/dev/stdin" ? stdin : fopen(path.c_str(), "rb");
    if (!f) return false;
    out.clear(); std::vector<uint8_t> buf(1 << 22); size_t n;
    while ((n = fread(buf.data(), 1, buf.size(), f)) > 0) out.insert(out.end(), buf.begin(), buf.begin() + n);
    if (f != stdin) fclose(f);
    return true;
}
static void write_all(const std::string& path, const uint8_t* p, size_t n, bool append) {
    FILE* f = path == "/dev/stdout" ? stdout : fopen(path.c_str(), append ? "ab" : "wb");
    if (!f) error_exit("Failed to open file for writing: " + path);
    if (n && fwrite(p, 1, n, f) != n) error_exit("Failed to write: " + path);
    if (f != stdout) fclose(f); else fflush(stdout);
}
static uint64_t nolb_threshold(const std::vector<uint8_t>& v) {   // SURVEY.md App. C Q10, src/fastqreader.cpp:31-46
    if (v.empty() || v.back() == '\n') return UINT64_MAX;
    if (v.size() % ((size_t)1 << 20) == 0) return 0;
    return ((uint64_t)(v.size() - 1) >> 20) << 20;
}
struct Gpu {
    rfq_ctx* c = nullptr;
    explicit Gpu(int dev) { if (rfq_create(&c, dev) != RFQ_OK) error_exit("no usable MI355X / HIP device (repaq_hip has no CPU fallback)"); }
    ~Gpu() { rfq_destroy(c); }
    void check(int rc) { if (rc != RFQ_OK) error_exit(rfq_last_error(c)); }
    void* put(const uint8_t* p, size_t n) { void* d = nullptr; check(rfq_dev_malloc(c, &d, n + 64)); check(rfq_copy_h2d(c, d, p, n)); return d; }
};

// Repaq::compress / compressPE (src/repaq.cpp:530-762): batches of whole lines, carry the unconsumed tail forward.
static void do_compress(const Options& o) {
    std::vector<uint8_t> t1, t2;
    if (!read_all(o.in1, t1)) error_exit("Failed to open file: " + o.in1);
    const bool two = !o.in2.empty();
    if (two && !read_all(o.in2, t2)) error_exit("Failed to open file: " + o.in2);
    const int paired = two ? RFQ_PE_TWO_FILES : (o.interleaved ? RFQ_PE_INTERLEAVED : RFQ_SE);
    Gpu g(o.device);
    const uint64_t th1 = nolb_threshold(t1), th2 = two ? nolb_threshold(t2) : th1;
    size_t p1 = 0, p2 = 0; bool first = true, wrote = false;
    for (;;) {
        size_t e1 = std::min(t1.size(), p1 + o.batchBytes), e2 = two ? std::min(t2.size(), p2 + o.batchBytes) : 0;
        const bool final = e1 == t1.size() && (!two || e2 == t2.size());
        // the two streams must advance by the same number of records: the device pairs record i with record i and reports
        // consumed bytes per stream, so any surplus on one side is simply carried over.
        void* d1 = g.put(t1.data() + p1, e1 - p1); void* d2 = two ? g.put(t2.data() + p2, e2 - p2) : nullptr;
        rfq_encode_args a; memset(&a, 0, sizeof a);
        a.d_fq1 = (const uint8_t*)d1; a.n1 = e1 - p1; a.d_fq2 = (const uint8_t*)d2; a.n2 = two ? e2 - p2 : 0; a.paired = paired;
        a.chunk_bases = (uint32_t)(std::max(100L, o.chunkKb) * 1000); a.final = final ? 1 : 0; a.emit_header = first ? 1 : 0;
        a.file_off1 = p1; a.file_off2 = p2; a.nolb_from1 = th1; a.nolb_from2 = th2;
        rfq_encode_result r; g.check(rfq_encode_batch(g.c, &a, &r));
        if (r.rfq_len) {
            std::vector<uint8_t> img(r.rfq_len); g.check(rfq_copy_d2h(g.c, img.data(), r.d_rfq, r.rfq_len));
            write_all(o.out1, img.data(), img.size(), wrote); wrote = true;
            if (r.n_chunks) first = false;
        }
        rfq_dev_free(g.c, d1); if (d2) rfq_dev_free(g.c, d2);
        if (final || r.input_ended) break;          // input_ended: the reader stopped at an empty line (src/fastqreader.cpp:180-191)
        if (r.consumed1 == 0 && (!two || r.consumed2 == 0)) {
            // not a single full chunk in this batch: grow it (a chunk can be larger than the batch)
            const_cast<Options&>(o).batchBytes *= 2; continue;
        }
        p1 += r.consumed1; p2 += r.consumed2;
    }
    if (!wrote) write_all(o.out1, nullptr, 0, false);   // empty input -> empty output, like the reference
}

struct Decoded { std::vector<uint8_t> a, b; uint64_t reads = 0, bases = 0; };
static Decoded decode_file(Gpu& g, const std::string& path, bool split) {
    std::vector<uint8_t> img; if (!read_all(path, img)) error_exit("Failed to open file: " + path);
    void* d = g.put(img.data(), img.size());
    rfq_decode_args a; memset(&a, 0, sizeof a);
    a.d_rfq = (const uint8_t*)d; a.n = img.size(); a.has_header = 1; a.split_pe = split ? 1 : 0; a.final = 1;
    rfq_decode_result r; g.check(rfq_decode_batch(g.c, &a, &r));
    Decoded out; out.reads = r.n_reads; out.bases = r.n_bases;
    out.a.resize(r.n1); if (r.n1) g.check(rfq_copy_d2h(g.c, out.a.data(), r.d_fq1, r.n1));
    out.b.resize(r.n2); if (r.n2) g.check(rfq_copy_d2h(g.c, out.b.data(), r.d_fq2, r.n2));
    rfq_dev_free(g.c, d);
    return out;
}
// Repaq::decompress / decompressPE (src/repaq.cpp:262-417)
static void do_decompress(const Options& o) {
    Gpu g(o.device);
    const bool split = !o.out2.empty();
    Decoded d = decode_file(g, o.in1, split);
    write_all(o.out1, d.a.data(), d.a.size(), false);
    if (split) write_all(o.out2, d.b.data(), d.b.size(), false);
}

// ---- compare mode (src/repaq.cpp:36-259): decode on the GPU, compare read by read with the FASTQ text, same JSON
struct Rec { std::string f[4]; };
static bool next_rec(const std::vector<uint8_t>& t, size_t& pos, Rec& r) {
    for (int k = 0; k < 4; k++) {
        if (pos >= t.size()) return false;
        size_t e = pos; while (e < t.size() && t[e] != '\n' && t[e] != '\r') e++;
        r.f[k].assign((const char*)t.data() + pos, e - pos);
        if (e < t.size() && t[e] == '\r' && e + 1 < t.size() && t[e + 1] == '\n') e++;
        pos = e + 1;
        if (r.f[k].empty()) return false;
    }
    return true;
}
static void report(const Options& o, bool passed, const std::string& msg, long fqReads, long fqBases, long rfqReads, long rfqBases) {   // :235-259
    std::string j = "{\n";
    j += passed ? "\t\"result\":\"passed\",\n" : "\t\"result\":\"failed\",\n";
    j += "\t\"msg\":\"" + msg + "\",\n";
    j += "\t\"fastq_reads\":" + std::to_string(fqReads) + ",\n\t\"rfq_reads\":" + std::to_string(rfqReads) + ",\n";
    j += "\t\"fastq_bases\":" + std::to_string(fqBases) + ",\n\t\"rfq_bases\":" + std::to_string(rfqBases) + "\n}\n";
    if (!o.json.empty()) write_all(o.json, (const uint8_t*)j.data(), j.size(), false);
    fputs(j.c_str(), stdout);
}
static void do_compare(const Options& o) {
    Gpu g(o.device);
    const bool pe = !o.in2.empty();
    Decoded d = decode_file(g, o.rfqCompare, pe);
    std::vector<uint8_t> f1, f2;
    if (!read_all(o.in1, f1)) error_exit("Failed to open file: " + o.in1);
    if (pe && !read_all(o.in2, f2)) error_exit("Failed to open file: " + o.in2);
    // decoded text always ends lines with '\n'; restore a dropped final newline so the record splitter sees whole records
    size_t pa = 0, pb = 0, qa = 0, qb = 0; long fqReads = 0, fqBases = 0, rfqReads = 0, rfqBases = 0;
    static const char* what[4] = { "name", "sequence", "strand", "quality" };
    for (;;) {
        Rec r; const bool second = pe && (rfqReads & 1);
        if (!next_rec(second ? d.b : d.a, second ? pb : pa, r)) break;
        rfqReads++; rfqBases += (long)r.f[1].size();
        Rec q;
        if (!next_rec(second ? f2 : f1, second ? qb : qa, q)) {
            report(o, false, "The RFQ file has more reads than the FASTQ file. The RFQ file has >= " + std::to_string(rfqReads) + " reads, while the FASTQ file only has " + std::to_string(fqReads) + " reads", fqReads, fqBases, rfqReads, rfqBases);
            return;
        }
        fqReads++; fqBases += (long)q.f[1].size();
        for (int k = 0; k < 4; k++) if (r.f[k] != q.f[k]) {
            report(o, false, std::string("The RFQ file and FASTQ file have different ") + what[k] + " in the " + std::to_string(rfqReads) + " read. " + r.f[k] + " | " + q.f[k], fqReads, fqBases, rfqReads, rfqBases);
            return;
        }
    }
    Rec q;
    if (next_rec(f1, qa, q) || (pe && next_rec(f2, qb, q))) {
        fqReads++;
        report(o, false, "The FASTQ file has more reads than the RFQ file. The FASTQ file has >= " + std::to_string(fqReads) + " reads, while the RFQ file only has " + std::to_string(rfqReads) + " reads", fqReads, fqBases, rfqReads, rfqBases);
        return;
    }
    report(o, true, "", fqReads, fqBases, rfqReads, rfqBases);
}

static void usage() {
    fputs("repaq_hip: repack FASTQ to .rfq on an MI355X (repaq v0.5.1 compatible)\n"
          "usage: repaq_hip [-c|-d|-p] -i in1 [-I in2] -o out1 [-O out2] [-k chunk_kb] [--stdin] [--stdout] [--interleaved_in]\n"
          "                 [-r rfq_to_compare] [-j json] [--device N] [--batch_mb M]\n", stderr);
}
int main(int argc, char** argv) {
    if (argc == 1) { usage(); return 0; }
    if (argc == 2 && !strcmp(argv[1], "--version")) { printf("repaq_hip 0.5.1-compatible (%s)\n", rfq_version()); return 0; }
    Options o;
    auto val = [&](int& i, const char* name) -> std::string {
        std::string a = argv[i]; const std::string lo = std::string("--") + name + "=";
        if (a.rfind(lo, 0) == 0) return a.substr(lo.size());
        if (i + 1 >= argc) error_exit(std::string("option needs value: --") + name);
        return argv[++i];
    };
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-i" || a == "--in1" || a.rfind("--in1=", 0) == 0) o.in1 = val(i, "in1");
        else if (a == "-o" || a == "--out1" || a.rfind("--out1=", 0) == 0) o.out1 = val(i, "out1");
        else if (a == "-I" || a == "--in2" || a.rfind("--in2=", 0) == 0) o.in2 = val(i, "in2");
        else if (a == "-O" || a == "--out2" || a.rfind("--out2=", 0) == 0) o.out2 = val(i, "out2");
        else if (a == "-c" || a == "--compress") o.compress = true;
        else if (a == "-d" || a == "--decompress") o.decompress = true;
        else if (a == "-p" || a == "--compare") o.compare = true;
        else if (a == "-k" || a == "--chunk" || a.rfind("--chunk=", 0) == 0) o.chunkKb = atol(val(i, "chunk").c_str());
        else if (a == "-r" || a == "--rfq_to_compare" || a.rfind("--rfq_to_compare=", 0) == 0) o.rfqCompare = val(i, "rfq_to_compare");
        else if (a == "-j" || a == "--json_compare_result" || a.rfind("--json_compare_result=", 0) == 0) o.json = val(i, "json_compare_result");
        else if (a == "--stdin") o.useStdin = true;
        else if (a == "--stdout") o.useStdout = true;
        else if (a == "--interleaved_in") o.interleaved = true;
        else if (a == "-v" || a == "--verify" || a == "-f" || a == "--fast_verify") {}        // the reference ignores the verify result (Q15)
        else if (a == "-t" || a == "--thread" || a == "-z" || a == "--compression") { (void)val(i, "thread"); }   // xz only
        else if (a == "--device") o.device = atoi(val(i, "device").c_str());
        else if (a == "--batch_mb") o.batchBytes = (size_t)atol(val(i, "batch_mb").c_str()) << 20;
        else { usage(); error_exit("unknown option: " + a); }
    }
    if ((int)o.compress + (int)o.decompress + (int)o.compare > 1) error_exit("repaq can run in compress/decompress/compare mode, you can only choose any one mode.");
    const bool dec = o.decompress, cmp = o.compare, enc = !dec && !cmp;
    // Options::validate (src/options.cpp:36-111)
    if (o.in1.empty()) {
        if (!o.in2.empty()) error_exit("read2 input is specified by <in2>, but read1 input is not specified by <in1>");
        if (o.useStdin && !cmp) o.in1 = "/dev/stdin"; else if (!cmp) error_exit("Please specify input file by <in1>, or enable --stdin if you want to read STDIN");
    }
    if (o.out1.empty()) {
        if (!o.out2.empty()) error_exit("read2 output is specified by <out2>, but read1 output is not specified by <out1>");
        if (o.useStdout) o.out1 = "/dev/stdout"; else if (!cmp) error_exit("Please specify output file by <out1>, or enable --stdout if you want to read STDIN");
    }
    for (const std::string* s : { &o.in1, &o.in2, &o.out1, &o.out2, &o.rfqCompare })
        if (ends_with(*s, ".gz") || ends_with(*s, ".xz")) error_exit(".gz / .xz streams are outside this build (zlib and xz are external to the .rfq codec): " + *s);
    const long cb = std::max(100L, o.chunkKb) * 1000;
    if (cb < 10000) error_exit("chunk size cannot be less than 10 kb");
    if (cb > 500000000) error_exit("chunk size cannot be greater than 500,000 kb");
    if (enc) {
        if (!o.out2.empty()) error_exit("In compress mode, only one RFQ output file is allowed, but you specified <out2>");
        if (ends_with(o.out1, ".fq") || ends_with(o.out1, ".fastq")) error_exit("In compress mode, the output should not be a FASTQ file. Expect a .rfq or .rfq.xz file, but got " + o.out1);
        if (ends_with(o.in1, ".rfq")) error_exit("In compress mode, the input should not be a RFQ file. Expect a .fq or .fq.gz file, but got " + o.in1);
        do_compress(o);
    } else if (dec) {
        if (!o.in2.empty()) error_exit("In decompress mode, only one RFQ input file is allowed, but you specified <in2>");
        if (ends_with(o.in1, ".fq") || ends_with(o.in1, ".fastq")) error_exit("In decompress mode, the input should not be a FASTQ file. Expect a .rfq or .rfq.xz file, but got " + o.in1);
        if (ends_with(o.out1, ".rfq")) error_exit("In decompress mode, the output should not be a RFQ file. Expect a .fq or .fq.gz file, but got " + o.out1);
        do_decompress(o);
    } else {
        if (o.useStdin) o.rfqCompare = "/dev/stdin";
        if (o.rfqCompare.empty()) error_exit("In compare mode, you should specify the RFQ file to compare by <rfq_to_compare>");
        if (!o.out1.empty() || !o.out2.empty()) error_exit("In compare mode, you cannot specify the output by <out1> or <out2>");
        if (o.in1.empty()) error_exit("Please specify input file by <in1>, or enable --stdin if you want to read STDIN");
        do_compare(o);
    }
    return 0;
}
