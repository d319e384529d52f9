// repaq_hip — host driver with repaq's command line (src/main.cpp:29-51, README.md:136-161) over the C-ABI of
// include/rfq_hip.h.  It is the counterpart of Repaq::compress / compressPE / decompress / decompressPE / compare*
// (src/repaq.cpp): stream I/O, batching with carry-over, header-once, line-break thresholds, PE even/odd outputs and the
// compare JSON live here; every byte of codec work happens on the GPU (there is no CPU codec in this binary).
//
// I/O pipeline: readers fill page-locked staging blocks (16 MB) ahead of the codec - a regular file by several
// threads with pread, block i always before block i+1 is handed out; stdin, .gz (src/fastqreader.cpp:31-37: blocked gzip is inflated on
// many threads, a plain gzip stream by zlib) and .xz through an `xz -d -c` pipe like src/main.cpp:160-177 by one reader, two blocks ahead
// so that the end of the input is known in time.
// The main thread moves blocks into a device-resident batch (256 MB by default: the codec's fixed cost per call is paid per batch, not
// per staging block), runs the codec and copies results out piece by piece into page-locked buffers; writers drain them - a regular
// file by several threads with pwrite at each piece's offset, stdout / .gz (src/writer.cpp:39-51; written as blocked gzip, deflated on many threads) / .rfq.xz (an `xz -z
// -c`
// pipe like src/main.cpp:134-159) by one thread in order.  Inputs and outputs of any size stream through: nothing is slurped.
#include "rfq_hip.h"
#include <zlib.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static void error_exit(const std::string& msg) { fprintf(stderr, "ERROR: %s\n", msg.c_str()); exit(-1); }   // src/util.h:246-249
static bool ends_with(const std::string& s, const std::string& e) { return s.size() >= e.size() && s.compare(s.size() - e.size(), e.size(), e) == 0; }

// Offset from which the reference's reader has its "no line break at the end" flag up, for a file of t > 0 bytes ending in `last` (src/fastqreader.cpp:31-46):
// the start of its final (short) 1 MiB block when the file lacks a final line break.  A size that is an exact multiple of 1 MiB has no short block: the flag
// goes up at the empty read behind the last full block whatever the last byte is (the test reads the byte in front of the buffer there), so the threshold is
// t itself - met by an unterminated last record and by the readers' last, failed attempt (the input's tail chunk).
static inline uint64_t nolb_threshold(uint64_t t, int last) {
    if ((t & ((1ull << 20) - 1)) == 0) return t;
    return last != '\n' ? ((t - 1) >> 20) << 20 : UINT64_MAX;
}
struct Options {
    std::string in1, out1, in2, out2, rfqCompare, json;
    long chunkKb = 1000; bool compress = false, decompress = false, compare = false, useStdin = false, useStdout = false, interleaved = false;
    bool completeCheck = false, fastCheck = false;   // -v / -f: decode what was just encoded and compare it with the input (src/repaq.cpp:430-528)
    std::vector<int> devices;              // --devices a,b,...: chunk-parallel compress of ONE input over several GPUs (host work queue)
    int device = 0; int threads = 1, compression = 3;
    size_t batchBytes = (size_t)256 << 20;  // device-resident text per codec call (per stream)
    size_t blockBytes = (size_t)16 << 20;   // page-locked staging block (never larger than a batch)
    // readers per regular input file (pread; tmpfs -> page-locked blocks -> HBM: 21.6 GB/s with 8, 29.2 with 16, tools/micro/mmap_h2d.cpp)
    int ioThreads = 16;
    int writeThreads = 1;                   // writers per regular output file (pwrite); tmpfs does not scale with more (nor with a mapping: see AsyncWriter), parallel file systems do
    bool serve = false;                     // --serve: jobs (one command line each) from stdin, the HIP runtime and the contexts stay up between them
    bool trace = false;                     // --trace: wall-clock marks of the pipeline on stderr
    // --bug_compat (-d with two outputs): lose what Repaq::decompressPE loses behind a non-last NO_LINE_BREAK chunk (src/repaq.cpp:376-403); Repaq::decompress loses
    // nothing; default: keep every read
    bool bugCompat = false;
    size_t block() const { return std::max<size_t>(std::min(blockBytes, batchBytes), (size_t)1 << 20); }   // >= the reader's 1 MiB block (line-break thresholds)
};
static const std::chrono::steady_clock::time_point g_t0 = std::chrono::steady_clock::now();
static bool g_trace = false;
static bool g_serve = false;               // --serve: many jobs in this process - what a one-shot run leaves to _exit (page-locked staging buffers) is handed back
static void trace_mark(const char* what) { if (g_trace) fprintf(stderr, "[trace] %8.1f ms  %s\n", std::chrono::duration<double,
        std::milli>(std::chrono::steady_clock::now() - g_t0).count(), what); }

// ------------------------------------------------------------------------------------------------ blocked gzip (BGZF)
// A .gz is one deflate stream in the reference (src/writer.cpp:39-51, src/fastqreader.cpp:31-37): one core, ~100 MB/s out, ~300 MB/s in - three
// orders of magnitude under the codec.  gzip members may be concatenated, so the driver WRITES a .gz as independent members of <= 64 KiB, each
// carrying its compressed size in a 'BC' extra field (the BGZF layout of the SAM specification, what bgzip writes): every reader of gzip (zlib's
// gzread, gunzip, the reference) reads it as the same text, and the members deflate on all the I/O threads at once.  READING, a .gz whose
// members carry that field is inflated block-parallel the same way; any other .gz goes through zlib's gzread as before (a plain gzip stream
// cannot be entered in the middle).
static const size_t BGZF_TEXT = 0xff00;                                      // text bytes per member (bgzip's block size)
template <class F> static void parallel_for(size_t n, int threads, F fn) {
    const size_t nt = std::min<size_t>((size_t)std::max(1, threads), n);
    if (nt <= 1) { for (size_t i = 0; i < n; i++) fn(i); return; }
    std::atomic<size_t> next{0}; std::vector<std::thread> th;
    for (size_t t = 0; t < nt; t++) th.emplace_back([&] { for (;;) { const size_t i = next.fetch_add(1); if (i >= n) return; fn(i); } });
    for (auto& t : th) t.join();
}
// threads for deflate / inflate of blocked gzip: the I/O threads, and up to 32 of half the host's cores (the codec leaves the host idle)
static int gz_threads(const Options& o) { const int hw = (int)std::thread::hardware_concurrency();
        return std::max(std::max(1, std::max(o.threads, o.ioThreads)), std::min(32, hw / 2)); }
// one member: 18-byte header (extra field 'B','C',2,BSIZE = member size - 1), raw deflate, crc32, text size
static void bgzf_member(const uint8_t* p, size_t n, int level, std::vector<uint8_t>& out) {
    out.resize(18 + compressBound((uLong)n) + 16 + 8);
    z_stream z; memset(&z, 0, sizeof z);
    if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) error_exit("zlib: deflateInit2 failed");
    z.next_in = (Bytef*)p; z.avail_in = (uInt)n; z.next_out = out.data() + 18; z.avail_out = (uInt)(out.size() - 18 - 8);
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) error_exit("zlib: deflate failed");
    const size_t zn = z.total_out; deflateEnd(&z);
    const size_t total = 18 + zn + 8;
    if (total > 65536) error_exit("internal: a gzip block outgrew 64 KiB");      // (0xff00 bytes of text cannot: stored blocks add 5 bytes per 64 KiB)
    const uint8_t h[18] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)((total - 1) & 0xff), (uint8_t)((total - 1) >> 8) };
    memcpy(out.data(), h, 18);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n), isz = (uint32_t)n;
    uint8_t* t = out.data() + 18 + zn;
    for (int i = 0; i < 4; i++) { t[i] = (uint8_t)(crc >> (8 * i)); t[4 + i] = (uint8_t)(isz >> (8 * i)); }
    out.resize(total);
}
// the member that starts at h (>= 18 bytes visible): its size if it is a BGZF member, else 0
static size_t bgzf_member_size(const uint8_t* h) {
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return 0;
    const size_t xlen = h[10] | ((size_t)h[11] << 8);
    // (bgzip writes exactly this; a member with more subfields takes the serial path)
    if (xlen != 6 || h[12] != 'B' || h[13] != 'C' || h[14] != 2 || h[15] != 0) return 0;
    return (size_t)(h[16] | ((size_t)h[17] << 8)) + 1;
}

// ------------------------------------------------------------------------------------------------ byte sources / sinks
struct ByteSource {                      // sequential bytes of a plain file, stdin, a .gz (block-parallel when blocked, else zlib) or a .xz (xz -d -c pipe)
    FILE* f = nullptr; gzFile gz = nullptr; bool piped = false; std::string path; int zthreads = 1;
    // blocked gzip: compressed bytes are read ahead into zbuf; `left` = text of a member that did not fit the caller's buffer
    bool bgzf = false; int zfd = -1; std::vector<uint8_t> zbuf; size_t zpos = 0, zend = 0; uint64_t zoff = 0; bool zeof = false; std::vector<uint8_t> left;
            size_t left_pos = 0;
    bool open(const std::string& p, int threads = 1) {
        path = p; zthreads = std::max(1, threads);
        if (ends_with(p, ".gz")) {
            zfd = ::open(p.c_str(), O_RDONLY);
            if (zfd >= 0) {
                uint8_t h[18]; const ssize_t k = pread(zfd, h, 18, 0);
                // (RFQ_GZ_BUF: test aid - a small read-ahead makes refills happen on small files)
                if (k == 18 && bgzf_member_size(h) >= 26) { bgzf = true; const char* e = getenv("RFQ_GZ_BUF");
                        zbuf.resize(e ? (size_t)std::max(64, atoi(e)) : (size_t)8 << 20); return true; }
                ::close(zfd); zfd = -1;
            }
            gz = gzopen(p.c_str(), "rb"); if (gz) gzbuffer(gz, 1 << 20); return gz != nullptr;
        }
        if (ends_with(p, ".xz")) { f = popen(("xz -d -c '" + p + "'").c_str(), "r"); piped = true; return f != nullptr; }
        f = p == "/dev/stdin" ? stdin : fopen(p.c_str(), "rb");
        return f != nullptr;
    }
    // more compressed bytes behind zbuf[zpos, zend) (moved to the front first); false at the end of the file
    bool zfill() {
        if (zeof) return false;
        if (zpos) { memmove(zbuf.data(), zbuf.data() + zpos, zend - zpos); zend -= zpos; zpos = 0; }
        if (zend == zbuf.size()) zbuf.resize(zbuf.size() * 2);
        const ssize_t k = pread(zfd, zbuf.data() + zend, zbuf.size() - zend, (off_t)zoff);
        if (k < 0) error_exit("Error to read gzip file");
        if (k == 0) { zeof = true; return false; }
        zend += (size_t)k; zoff += (uint64_t)k; return true;
    }
    size_t read_bgzf(uint8_t* dst, size_t cap) {
        size_t got = 0;
        while (got < cap) {
            if (left_pos < left.size()) { const size_t k = std::min(cap - got, left.size() - left_pos); memcpy(dst + got, left.data() + left_pos, k); got += k;
                    left_pos += k; continue; }
            if (gz) { const int n = gzread(gz, dst + got, (unsigned)std::min<size_t>(cap - got, 1u << 30)); if (n < 0) error_exit("Error to read gzip file");
                    if (n == 0) break; got += (size_t)n; continue; }
            // the members that fit what is left of dst: where each starts (relative to zpos: refills move the buffer's content), and where its text goes
            struct Mem { size_t at, size, text, out; }; std::vector<Mem> ms; size_t out = got, rel = 0; bool spill = false, foreign = false;
            // bytes visible from the next member's start on
            auto need = [&](size_t bytes) -> bool { while (zend - zpos < rel + bytes) if (!zfill()) return false; return true; };
            for (;;) {
                if (!need(1)) break;                                           // nothing behind the last member: the end of the file
                if (!need(18)) error_exit("Error to read gzip file");
                const size_t sz = bgzf_member_size(zbuf.data() + zpos + rel);
                if (sz < 26) { foreign = true; break; }                        // a member of another shape: zlib takes over from here (below)
                if (!need(sz)) error_exit("Error to read gzip file");
                const uint8_t* e = zbuf.data() + zpos + rel + sz - 4;
                        const size_t text = (size_t)e[0] | ((size_t)e[1] << 8) | ((size_t)e[2] << 16) | ((size_t)e[3] << 24);
                if (text > 65536) error_exit("Error to read gzip file");
                if (out + text > cap) { if (ms.empty()) { ms.push_back(Mem{ rel, sz, text, 0 }); rel += sz; spill = true; } break; }
                ms.push_back(Mem{ rel, sz, text, out }); out += text; rel += sz;
                if (ms.size() >= 4096) break;
            }
            for (auto& m : ms) m.at += zpos;
            if (ms.empty() && !foreign) break;                                 // end of the file
            if (spill) { left.resize(ms[0].text); left_pos = 0; }
            const uint8_t* zb = zbuf.data(); uint8_t* lp = left.data(); const bool sp = spill;
            parallel_for(ms.size(), zthreads, [&](size_t i) {
                const Mem& m = ms[i]; uint8_t* o = sp ? lp : dst + m.out;
                z_stream z; memset(&z, 0, sizeof z);
                if (inflateInit2(&z, -15) != Z_OK) error_exit("zlib: inflateInit2 failed");
                z.next_in = (Bytef*)(zb + m.at + 18); z.avail_in = (uInt)(m.size - 26); z.next_out = o; z.avail_out = (uInt)m.text;
                const int rc = m.text || m.size > 26 ? inflate(&z, Z_FINISH) : Z_STREAM_END;
                if (rc != Z_STREAM_END || z.total_out != m.text) error_exit("Error to read gzip file");
                inflateEnd(&z);
                const uint8_t* e = zb + m.at + m.size - 8; const uint32_t want = (uint32_t)e[0] | ((uint32_t)e[1] << 8) | ((uint32_t)e[2] << 16) | ((uint32_t)e[3] << 24);
                if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), o, (uInt)m.text) != want) error_exit("Error to read gzip file");
            });
            zpos += rel;
            if (!spill) got = out;
            if (foreign) {                                                     // the rest of the file is some other gzip: one stream, zlib
                const uint64_t file_at = zoff - (zend - zpos);
                if (lseek(zfd, (off_t)file_at, SEEK_SET) < 0) error_exit("Error to read gzip file");
                gz = gzdopen(zfd, "rb"); if (!gz) error_exit("Error to read gzip file");
                gzbuffer(gz, 1 << 20); zfd = -1;
            }
        }
        return got;
    }
    size_t read(uint8_t* dst, size_t cap) {          // fills `cap` unless the stream ends
        if (bgzf) return read_bgzf(dst, cap);
        size_t got = 0;
        while (got < cap) {
            long n;
            if (gz) { n = gzread(gz, dst + got, (unsigned)std::min<size_t>(cap - got, 1u << 30)); if (n < 0) error_exit("Error to read gzip file"); }
            else n = (long)fread(dst + got, 1, cap - got, f);
            if (n <= 0) break;
            got += (size_t)n;
        }
        return got;
    }
    void close() {
        if (gz) gzclose(gz);
        else if (piped) { if (f && pclose(f) != 0) error_exit("failed to call xz, please confirm that xz is installed in your system"); }
        else if (f && f != stdin) fclose(f);
        if (zfd >= 0) ::close(zfd);
        gz = nullptr; f = nullptr; zfd = -1;
    }
};
struct ByteSink {
    FILE* f = nullptr; bool piped = false, bgzf = false; std::string path; int level = 3, zthreads = 1; std::vector<uint8_t> pend;   // pend: text short of a member
    void open(const std::string& p, const Options& o) {
        path = p;
        if (ends_with(p, ".gz")) {                                          // src/writer.cpp:39-44: the same text at the same level, as blocked gzip (above)
            f = fopen(p.c_str(), "wb"); if (!f) error_exit("Failed to open file for writing: " + p);
            bgzf = true; level = o.compression; zthreads = gz_threads(o); return;
        }
        if (ends_with(p, ".xz")) {                                          // src/main.cpp:134-159
            std::string cmd = "xz -z -c";
            if (o.threads > 1) cmd += " -T" + std::to_string(o.threads);
            if (o.compression <= 4) cmd += " -" + std::to_string(o.compression + 5);
            else { unsigned long dict = (64ul * 1024 * 1024) << (o.compression - 4); if (o.compression == 9) dict = 1536ul * 1024 * 1024;
                    cmd += " --lzma2=\"dict=" + std::to_string(dict) + "\""; }
            if (o.compression >= 4 && o.threads > 1) fprintf(stderr, "WARNING: when repaq compression level is >= 4, only single thread will be used for xz. Your options: compression = %d, thread = %d\n", o.compression, o.threads);
            cmd += " > '" + p + "'";
            f = popen(cmd.c_str(), "w"); piped = true;
            if (!f) error_exit("failed to call xz, please confirm that xz is installed in your system");
            return;
        }
        f = p == "/dev/stdout" ? stdout : fopen(p.c_str(), "wb");
        if (!f) error_exit("Failed to open file for writing: " + p);
    }
    void put(const uint8_t* p, size_t n) { if (n && fwrite(p, 1, n, f) != n) error_exit("Failed to write: " + path); }
    // whole members of p[0, n) deflated on all threads, written in order; what is left (< one member) stays in pend
    void write_bgzf(const uint8_t* p, size_t n) {
        if (!pend.empty()) {                                                 // top the pending text up to one member first
            const size_t k = std::min(n, BGZF_TEXT - pend.size()); pend.insert(pend.end(), p, p + k); p += k; n -= k;
            if (pend.size() < BGZF_TEXT) return;
            std::vector<uint8_t> z; bgzf_member(pend.data(), pend.size(), level, z); put(z.data(), z.size()); pend.clear();
        }
        const size_t nb = n / BGZF_TEXT;
        for (size_t b0 = 0; b0 < nb; b0 += 1024) {                            // (64 MB of text per round)
            const size_t cnt = std::min<size_t>(1024, nb - b0); std::vector<std::vector<uint8_t>> z(cnt);
            const int lv = level;
            parallel_for(cnt, zthreads, [&](size_t i) { bgzf_member(p + (b0 + i) * BGZF_TEXT, BGZF_TEXT, lv, z[i]); });
            for (auto& m : z) put(m.data(), m.size());
        }
        pend.assign(p + nb * BGZF_TEXT, p + n);
    }
    void write(const uint8_t* p, size_t n) {
        if (bgzf) { write_bgzf(p, n); return; }
        while (n) {
            const size_t k = std::min<size_t>(n, 1u << 30);
            if (fwrite(p, 1, k, f) != k) error_exit("Failed to write: " + path);
            p += k; n -= k;
        }
    }
    void close() {
        if (bgzf) {
            if (!pend.empty()) { std::vector<uint8_t> z; bgzf_member(pend.data(), pend.size(), level, z); put(z.data(), z.size()); pend.clear(); }
            // the empty member bgzip ends a file with (an empty input is just this)
            std::vector<uint8_t> z; bgzf_member(nullptr, 0, level, z); put(z.data(), z.size());
            if (fclose(f) != 0) error_exit("Failed to write: " + path);
        }
        else if (piped) { if (f && pclose(f) != 0) error_exit("failed to call xz, please confirm that xz is installed in your system"); }
        else if (f == stdout) fflush(stdout);
        else if (f) fclose(f);
        f = nullptr; bgzf = false;
    }
};

struct Gpu {
    rfq_ctx* c = nullptr; bool leased = false;
    // --serve: the main context of a device is made once and lent to every job (its workspace - several hundred MB of device buffers sized by the batches - stays);
    // one job runs at a time, and a job's other contexts (verifier, per-device workers and ingestion of --devices) are still its own
    explicit Gpu(int dev, bool main_ctx = false) {
        static std::map<int, rfq_ctx*> kept; static std::mutex km;
        // what the switches were when the context was made (the RFQ_* environment at rfq_create): a lent context goes to every job in that state, whatever
        // a job before it set (only the header was reset between jobs)
        static std::map<int, std::vector<std::pair<std::string, std::string>>> made;
        if (g_serve && main_ctx) { std::unique_lock<std::mutex> lk(km); auto it = kept.find(dev);
                if (it == kept.end()) {
                    if (rfq_create(&c, dev) != RFQ_OK) error_exit("no usable MI355X / HIP device (repaq_hip has no CPU fallback)");
                    kept[dev] = c;
                    for (int i = 0; rfq_option_name(i); i++) { char v[64] = ""; if (rfq_get_option(c, rfq_option_name(i), v, sizeof v) == RFQ_OK) made[dev].emplace_back(rfq_option_name(i), v); }
                } else { c = it->second; for (auto& kv : made[dev]) (void)rfq_set_option(c, kv.first.c_str(), kv.second.empty() ? nullptr : kv.second.c_str()); }
                leased = true; rfq_clear_header(c); return; }
        if (rfq_create(&c, dev) != RFQ_OK) error_exit("no usable MI355X / HIP device (repaq_hip has no CPU fallback)");
    }
    ~Gpu() { if (!leased) rfq_destroy(c); }
    void check(int rc) { if (rc != RFQ_OK) error_exit(rfq_last_error(c)); }
    void* dev(size_t n) { void* d = nullptr; check(rfq_dev_malloc(c, &d, n + 64)); return d; }
    uint8_t* pinned(size_t n) { void* h = nullptr; check(rfq_host_alloc(c, &h, n + 64)); return (uint8_t*)h; }
};

// Readers: page-locked blocks of `block` bytes, handed out in file order.
//  * a regular file: its size is known up front (so is whether it ends with a line break), `threads` readers pread block after block;
//    a reader claims a block index together with the buffer it will fill, so block i never waits for a buffer held by a later block;
//  * anything sequential (stdin, .gz, .xz pipe): one reader, two blocks ahead - the end of the input is then known when the block
//    before the last two is handed out (the line-break thresholds and `final` need it).
struct Block { uint8_t* p = nullptr; size_t n = 0; };
class Prefetcher {
    // g: null until attach() - the readers then fill ordinary page-aligned buffers (plain), which attach() page-locks
    // (g is written by attach() on the main thread while the readers run: atomic.  plain: buffers not page-locked yet; owned: every buffer this object made with
    // posix_memalign - page-locked by rfq_host_register at some point - to be unregistered and freed at the end; pinned_: those that came from rfq_host_alloc)
    ByteSource src; std::atomic<Gpu*> g; std::vector<uint8_t*> plain, owned, pinned_; size_t block;
    uint8_t* new_block() {
        if (Gpu* gp = g.load()) { uint8_t* p = gp->pinned(block); std::unique_lock<std::mutex> lk(mu); pinned_.push_back(p); return p; }
        void* p = nullptr; if (posix_memalign(&p, 4096, block + 64) != 0) error_exit("out of memory");
        std::unique_lock<std::mutex> lk(mu); owned.push_back((uint8_t*)p);
        if (Gpu* gp = g.load()) { lk.unlock(); gp->check(rfq_host_register(gp->c, p, block + 64)); } else plain.push_back((uint8_t*)p);
        return (uint8_t*)p; }
    std::vector<std::thread> th; std::mutex mu; std::condition_variable cv;
    // to_alloc: page-locked blocks not allocated yet (a reader allocates its own: pinning runs beside the first reads)
    std::map<uint64_t, Block> ready; uint64_t next_out = 0, next_claim = 0, n_blocks = 0; std::vector<uint8_t*> freeb; int to_alloc = 0;
    bool eof = false, stop = false, regular = false; uint64_t total = 0; int last_byte = -1; int fd = -1; int running = 0; int nbuf_ = 4;
    void run_seq() {
        for (;;) {
            uint8_t* buf;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !freeb.empty() || to_alloc > 0 || stop; }); if (stop) { eof = true; cv.notify_all(); return; }
              if (!freeb.empty()) { buf = freeb.back(); freeb.pop_back(); } else { to_alloc--; buf = nullptr; } }
            if (!buf) buf = new_block();
            const size_t n = src.read(buf, block);
            std::unique_lock<std::mutex> lk(mu);
            if (n) { total += n; last_byte = buf[n - 1]; ready[next_claim++] = Block{ buf, n }; } else freeb.push_back(buf);
            if (n < block) { eof = true; cv.notify_all(); return; }
            cv.notify_all();
        }
    }
    void run_par() {
        for (;;) {
            uint8_t* buf; uint64_t idx;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || next_claim >= n_blocks || !freeb.empty() || to_alloc > 0; });
              if (stop || next_claim >= n_blocks) { if (--running == 0) { eof = true; cv.notify_all(); } return; }
              if (!freeb.empty()) { buf = freeb.back(); freeb.pop_back(); } else { to_alloc--; buf = nullptr; }
              idx = next_claim++; }
            if (!buf) buf = new_block();
            const uint64_t off = idx * (uint64_t)block; const size_t want = (size_t)std::min<uint64_t>(block, total - off); size_t got = 0;
            while (got < want) { const ssize_t k = pread(fd, buf + got, want - got, (off_t)(off + got)); if (k <= 0) break; got += (size_t)k; }
            if (got != want) error_exit("Failed to read file: " + src.path);
            std::unique_lock<std::mutex> lk(mu); ready[idx] = Block{ buf, got }; cv.notify_all();
        }
    }
public:
    // (the context may come later: a compress starts its readers first and creates the context - 0.2 s of HIP start-up - while they fill their first blocks)
    void attach(Gpu& gpu) { std::vector<uint8_t*> todo; { std::unique_lock<std::mutex> lk(mu); g.store(&gpu); todo.swap(plain);
            } for (uint8_t* p : todo) gpu.check(rfq_host_register(gpu.c, p, block + 64)); }
    Prefetcher(Gpu& gpu, const std::string& path, size_t block_bytes, int threads = 1) : Prefetcher(&gpu, path, block_bytes, threads) {}
    Prefetcher(Gpu* gpu, const std::string& path, size_t block_bytes, int threads = 1) : g(gpu), block(block_bytes) {
        struct stat st;
        regular = !ends_with(path, ".gz") && !ends_with(path, ".xz") && path != "/dev/stdin" && stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
        int nbuf = 4;
        if (regular) {
            fd = ::open(path.c_str(), O_RDONLY); if (fd < 0) error_exit("Failed to open file: " + path);
            src.path = path; total = (uint64_t)st.st_size;
            block = std::min(block, std::max<size_t>((size_t)total + 1, (size_t)1 << 20));          // small files: no point in pinning full blocks
            n_blocks = (total + block - 1) / block;
            if (total) { uint8_t c = 0; if (pread(fd, &c, 1, (off_t)(total - 1)) == 1) last_byte = c; }
            threads = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(1, threads), n_blocks));
            nbuf = (int)std::min<uint64_t>((uint64_t)threads + 3, std::max<uint64_t>(n_blocks, 1));
            if (n_blocks == 0) eof = true;
        } else if (!src.open(path, std::max(threads, std::min(32, (int)std::thread::hardware_concurrency() / 2)))) error_exit("Failed to open file: " + path);
        to_alloc = nbuf; nbuf_ = nbuf;
        if (regular) { running = n_blocks ? threads : 0; for (int i = 0; i < running; i++) th.emplace_back([this] { run_par(); }); }
        else th.emplace_back([this] { run_seq(); });
    }
    // (a caller that stops early — an empty line ends the input, a failed compare — leaves blocks unread: wake the readers up)
    ~Prefetcher() { { std::unique_lock<std::mutex> lk(mu); stop = true; cv.notify_all(); } for (auto& t : th) if (t.joinable()) t.join(); if (fd >= 0) ::close(fd);
            else src.close();
            // the staging buffers go back (registered blocks were never unregistered or freed; harmless in a process that _exits, not in one that serves many jobs)
            // (a one-shot run leaves them to _exit: unlocking 38 blocks of 16 MB costs it 0.1 s for nothing)
            Gpu* gp = g.load();
            if (g_serve) { for (uint8_t* p : owned) { if (gp && std::find(plain.begin(), plain.end(), p) == plain.end()) (void)rfq_host_unregister(gp->c, p); free(p); }
                    if (gp) for (uint8_t* p : pinned_) (void)rfq_host_free(gp->c, p); } }
    // next block; false when the input is exhausted.  After it returns, end_known() tells whether the end of the input is known;
    // if not, at least two more full blocks follow the one just returned.
    bool next(Block& b) {
        std::unique_lock<std::mutex> lk(mu);
        if (regular) { if (next_out >= n_blocks) return false; cv.wait(lk, [&] { return ready.count(next_out) != 0; }); }
        else { cv.wait(lk, [&] { return ready.size() >= 3 || eof; }); if (ready.empty()) return false; }
        auto it = ready.find(next_out); b = it->second; ready.erase(it); next_out++;
        return true;
    }
    void release(const Block& b) { std::unique_lock<std::mutex> lk(mu); freeb.push_back(b.p); cv.notify_all(); }
    // staging blocks the consumer may hold back (copies in flight) without starving the readers: half of a regular file's, none of a sequential source's
    // (its reader must be able to run three blocks ahead with the four it has)
    size_t flight_limit() const { return regular ? (size_t)std::max(1, std::min(6, nbuf_ / 2)) : 1u; }
    bool end_known() { std::unique_lock<std::mutex> lk(mu); return regular || eof; }
    bool drained() { std::unique_lock<std::mutex> lk(mu); return regular ? next_out >= n_blocks : (eof && ready.empty()); }
    uint64_t total_bytes() { std::unique_lock<std::mutex> lk(mu); return total; }
    int final_byte() { std::unique_lock<std::mutex> lk(mu); return last_byte; }
};

// Writers: results leave the device piece by piece through a small pool of page-locked buffers.  A regular output file is written by
// several threads with pwrite (every piece knows its offset); anything sequential (stdout, .gz, xz pipe) by one thread in order.
class AsyncWriter {
    ByteSink sink; Gpu& g; std::vector<std::thread> th; std::mutex mu; std::condition_variable cv;
    struct Item { uint8_t* p; size_t n, cap; uint64_t off; }; std::deque<Item> q; std::vector<Item> pool; bool done = false; int in_flight = 0, max_flight = 3;
    size_t piece; bool regular = false; int fd = -1; uint64_t off = 0; std::string path;
    // (Round 5 tried the other way in: the output pre-sized and mapped - per piece, then in 1 GiB windows kept to the end - and the pieces copied into the mapping by 16
    // writers, since pwrite on one tmpfs file serialises on its inode lock.  A micro-benchmark promised 14 GB/s against pwrite's 3 - 5; in the driver it was SLOWER than
    // pwrite, 1.6 - 1.8 s against 1.4 s for 8 GB: what bounds either is the kernel's allocation of 2 M fresh tmpfs pages, and the benchmark's fast run had re-used
    // the pages of the file it had just unlinked.  profiles/r05_io_micro.txt, tools/micro/d2h_out.cpp.)
    void run() {
        for (;;) {
            Item it;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !q.empty() || done; }); if (q.empty()) return; it = q.front(); q.pop_front(); }
            if (regular) { size_t w = 0; while (w < it.n) { const ssize_t k = pwrite(fd, it.p + w, it.n - w, (off_t)(it.off + w)); if (k <= 0) error_exit("Failed to write: " + path); w += (size_t)k; } }
            else sink.write(it.p, it.n);
            std::unique_lock<std::mutex> lk(mu); pool.push_back(it); in_flight--; cv.notify_all();
        }
    }
    // a page-locked buffer of >= n bytes (at most max_flight in flight); fill it, then submit()
    uint8_t* acquire(size_t n, size_t& cap) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return in_flight < max_flight; });
        in_flight++;
        for (size_t i = 0; i < pool.size(); i++) if (pool[i].cap >= n) { Item it = pool[i]; pool.erase(pool.begin() + i); cap = it.cap; return it.p; }
        uint8_t* stale = nullptr; if (!pool.empty()) { stale = pool.back().p; pool.pop_back(); }
        lk.unlock();
        if (stale) rfq_host_free(g.c, stale);
        cap = std::min(piece, std::max<size_t>(n + n / 4 + 4096, (size_t)1 << 20)); if (cap < n) cap = n; return g.pinned(cap);
    }
    void submit(uint8_t* p, size_t n, size_t cap) { std::unique_lock<std::mutex> lk(mu); q.push_back(Item{ p, n, cap, off }); off += n; cv.notify_all(); }
public:
    AsyncWriter(Gpu& gpu, const std::string& p, const Options& o) : g(gpu), piece(o.block()), path(p) {
        struct stat st; int threads = 1;
        regular = !ends_with(p, ".gz") && !ends_with(p, ".xz") && p != "/dev/stdout" && (stat(p.c_str(), &st) != 0 || S_ISREG(st.st_mode));
        if (regular) { fd = ::open(p.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644); if (fd < 0) error_exit("Failed to open file for writing: " + p);
                threads = std::max(1, o.writeThreads); max_flight = threads + 2; }
        else sink.open(p, o);
        for (int i = 0; i < threads; i++) th.emplace_back([this] { run(); });
    }
    // n bytes of device memory -> the output, in order
    void write_dev(const uint8_t* d, size_t n) {
        for (size_t at = 0; at < n; at += piece) {
            const size_t k = std::min(piece, n - at); size_t cap; uint8_t* h = acquire(k, cap);
            g.check(rfq_copy_d2h(g.c, h, d + at, k)); submit(h, k, cap);
        }
    }
    void finish() {
        { std::unique_lock<std::mutex> lk(mu); done = true; cv.notify_all(); }
        for (auto& t : th) if (t.joinable()) t.join();
        th.clear();
        if (g_serve) { for (auto& it : pool) (void)rfq_host_free(g.c, it.p); pool.clear(); }   // (a serving process runs many jobs: the staging buffers go back)
        if (regular) { if (fd >= 0 && ::close(fd) != 0) error_exit("Failed to write: " + path); fd = -1; } else sink.close();
    }
    ~AsyncWriter() { if (!th.empty()) finish(); }
};

// Device text of one input stream: [carry of the previous batch | newly copied block(s)], 16-byte aligned at its base
struct DevStream {
    void* d[2] = { nullptr, nullptr }; size_t cap[2] = { 0, 0 }; int cur = 0; size_t have = 0; uint64_t file_off = 0; bool ended = false;
    uint8_t* base() const { return (uint8_t*)d[cur]; }
    // drop the first `consumed` bytes: the tail moves to the front of the other buffer (device-to-device)
    void advance(Gpu& g, size_t consumed, size_t room) {
        const size_t carry = have - consumed; const int nx = cur ^ 1;
        if (cap[nx] < carry + room) { if (d[nx]) rfq_dev_free(g.c, d[nx]); cap[nx] = carry + room + (carry + room) / 8; d[nx] = g.dev(cap[nx]); }
        if (carry) g.check(rfq_copy_d2d(g.c, d[nx], (uint8_t*)d[cur] + consumed, carry));
        cur = nx; have = carry; file_off += consumed;
    }
    void append(Gpu& g, const Block& b, size_t room = 0) {
        if (cap[cur] < have + b.n) {                                          // grow (first block, or the streams of a pair drift apart)
            const size_t nc = std::max(have + b.n + (have + b.n) / 8, room); void* nd = g.dev(nc);
            if (have) g.check(rfq_copy_d2d(g.c, nd, d[cur], have));
            if (d[cur]) rfq_dev_free(g.c, d[cur]);
            d[cur] = nd; cap[cur] = nc;
        }
        g.check(rfq_copy_h2d(g.c, (uint8_t*)d[cur] + have, b.p, b.n)); have += b.n;
    }
    // the same, the copy only queued (rfq_copy_h2d_async): a run of blocks then crosses the link back to back while the caller waits for the next
    // block to be read; `flight` keeps the staging blocks until their copies are done (the caller hands them back to their reader: drain)
    std::deque<std::pair<Block, uint64_t>> flight;
    void append_async(Gpu& g, const Block& b, size_t room = 0) {
        if (cap[cur] < have + b.n) {                                          // grow: the copies queued into the old buffer must have landed before it is moved
            g.check(rfq_copy_sync(g.c));
            const size_t nc = std::max(have + b.n + (have + b.n) / 8, room); void* nd = g.dev(nc);
            if (have) g.check(rfq_copy_d2d(g.c, nd, d[cur], have));
            if (d[cur]) rfq_dev_free(g.c, d[cur]);
            d[cur] = nd; cap[cur] = nc;
        }
        uint64_t t = 0; g.check(rfq_copy_h2d_async(g.c, (uint8_t*)d[cur] + have, b.p, b.n, &t)); have += b.n; flight.emplace_back(b, t);
    }
    template <class Release> void drain(Gpu& g, bool all, Release rel) {     // hand back the staging blocks whose copies are done (all: wait for every one)
        if (all) g.check(rfq_copy_sync(g.c));
        while (!flight.empty()) { if (!all) { const int d_ = rfq_copy_done(g.c, flight.front().second); if (d_ < 0) g.check(d_); if (!d_) break;
                } rel(flight.front().first); flight.pop_front(); }
    }
    void free_all(Gpu& g) { for (int i = 0; i < 2; i++) if (d[i]) rfq_dev_free(g.c, d[i]); }
};

struct Rec { std::string f[4]; };
struct TextCursor {                      // growing text + a refill callback; records are cut with FastqReader's line rules
    std::vector<uint8_t> t; size_t pos = 0; bool ended = false; std::function<void(TextCursor&)> refill;
    uint64_t off0 = 0;                   // offset of t[0] in its (decompressed) file: the reader works in 1 MiB blocks of it
    void compact() { if (pos > (1u << 24)) { t.erase(t.begin(), t.begin() + pos); off0 += pos; pos = 0; } }
    // FastqReader::getLine + read (src/fastqreader.cpp:94-196): a line ends at '\r' or '\n'; one '\n' right after the terminator is
    // skipped too, unless it is the last byte of the reader's 1 MiB block (for the last block: of the file) or opens the next block
    // (`end < mBufDataLen-1`, :112-114); a record with an empty line ends the input
    bool next(Rec& r) {
        for (;;) {
            size_t p = pos; int k = 0; bool need = false, empty = false;
            for (; k < 4; k++) {
                if (p >= t.size()) { if (!ended) { need = true; break; } r.f[k].clear(); empty = true; continue; }
                size_t e = p; while (e < t.size() && t[e] != '\n' && t[e] != '\r') e++;
                if (e + 2 >= t.size() && !ended) { need = true; break; }      // the rule looks one byte past the candidate '\n'
                r.f[k].assign((const char*)t.data() + p, e - p);
                size_t nx = e + 1;
                if (nx < t.size() && t[nx] == '\n') {
                    const uint64_t at = off0 + nx;
                    const bool opens_block = (at & 0xFFFFF) == 0, closes_block = ((at + 1) & 0xFFFFF) == 0 || nx + 1 == t.size();
                    if (!opens_block && !closes_block) nx++;
                }
                p = std::min(nx, t.size());
                if (r.f[k].empty()) empty = true;
            }
            if (k == 4) { pos = p; return !empty; }
            if (!need) return false;
            compact(); refill(*this);
        }
    }
};
// -v / -f (Repaq::completeCheckAndOutput, src/repaq.cpp:430-528), on the device: the image a batch was just encoded to is decoded on a
// second context and compared byte for byte with the text it came from (rfq_compare_bytes).  Only when the bytes differ (text the
// reader normalises: "\r\n", blank lines, a trailing partial record — or a real codec fault) are both sides cut into records; a
// differing read is reported on stderr in the reference's words and, like there, the output is written regardless (App. C Q15);
// a differing read COUNT ends the run with the reference's error.
struct Verifier {
    Gpu gv; bool have_hdr = false; long pass = 0; const Options& o;
    Verifier(const Options& opt, int device) : gv(device), o(opt) {}
    bool wanted() { const bool w = o.completeCheck || (o.fastCheck && pass % 10 == 0); pass++; return w; }
    void check(Gpu& g, const rfq_encode_result& r, bool image_has_header, bool final, bool two,
               const uint8_t* in1, size_t n1, const uint8_t* in2, size_t n2, uint64_t off1, uint64_t off2) {
        if (!r.n_chunks) return;
        if (!image_has_header && !have_hdr) { uint8_t hb[RFQ_HEADER_MAX]; size_t hn = 0; g.check(rfq_get_header(g.c, hb, &hn)); gv.check(rfq_set_header(gv.c, hb, hn)); }
        have_hdr = true;
        rfq_decode_args a; memset(&a, 0, sizeof a);
        a.d_rfq = r.d_rfq; a.n = r.rfq_len; a.has_header = image_has_header ? 1 : 0; a.split_pe = two ? 1 : 0; a.final = final ? 1 : 0;
        rfq_decode_result d; gv.check(rfq_decode_batch(gv.c, &a, &d));
        const uint8_t* got[2] = { d.d_fq1, d.d_fq2 }; const size_t gn[2] = { d.n1, two ? d.n2 : 0 };
        const uint8_t* exp[2] = { in1, in2 }; const size_t en[2] = { n1, two ? n2 : 0 }; const uint64_t eo[2] = { off1, off2 };
        bool same = true;
        for (int s = 0; s < (two ? 2 : 1) && same; s++) {
            uint64_t at = 0; same = gn[s] == en[s];
            if (same) { gv.check(rfq_compare_bytes(gv.c, got[s], exp[s], en[s], &at)); same = at == en[s]; }
        }
        if (same) return;
        for (int s = 0; s < (two ? 2 : 1); s++) {
            TextCursor ce, cg; ce.ended = cg.ended = true; ce.off0 = eo[s];
            ce.t.resize(en[s]); if (en[s]) gv.check(rfq_copy_d2h(gv.c, ce.t.data(), exp[s], en[s]));
            cg.t.resize(gn[s]); if (gn[s]) gv.check(rfq_copy_d2h(gv.c, cg.t.data(), got[s], gn[s]));
            // (the text handed in may run past the last chunk of a non-final batch; the decoded side says how many reads to draw)
            Rec x, y;
            while (cg.next(y)) {
                if (!ce.next(x)) error_exit("encoding error in chunk, the output will be wrong, quit now!");
                for (int k = 0; k < 4; k++) if (x.f[k] != y.f[k]) {
                    fprintf(stderr, "integrity check failure \nexpected: \n%s\ngot:\n%s\n", x.f[k].c_str(), y.f[k].c_str());
                    return;
                }
            }
        }
    }
};

// Repaq::compress / compressPE (src/repaq.cpp:530-762)
static void do_compress(const Options& o) {
    const bool two = !o.in2.empty();
    const int paired = two ? RFQ_PE_TWO_FILES : (o.interleaved ? RFQ_PE_INTERLEAVED : RFQ_SE);
    const size_t block = o.block(), batch = std::max(o.batchBytes, block);      // staging block (>= the reader's 1 MiB block: see nolb below) / device batch
    // the readers first: they fill their first blocks while the HIP runtime comes up (context, code objects: 0.2 - 0.3 s of an 0.7 s run on 2 x 4 GB)
    Prefetcher* in[2] = { new Prefetcher((Gpu*)nullptr, o.in1, block, o.ioThreads), two ? new Prefetcher((Gpu*)nullptr, o.in2, block, o.ioThreads) : nullptr };
    Gpu g(o.device, true);
    for (int s = 0; s < 2; s++) if (in[s]) in[s]->attach(g);
    AsyncWriter out(g, o.out1, o);
    DevStream ds[2]; const int ns = two ? 2 : 1;
    bool first = true; size_t want = batch;                                    // bytes a stream should hold before a batch is tried
    trace_mark("compress: pipeline up");
    Verifier* ver = (o.completeCheck || o.fastCheck) ? new Verifier(o, o.device) : nullptr;
    for (;;) {
        // (the two files of a pair block by block in turn: filled one after the other, one file's readers sat idle - their buffers full - while
        // the other's worked: 8 GB crossed at the rate of ONE set of readers, 22 - 25 GB/s)
        for (bool more = true; more; ) {
            more = false;
            for (int s = 0; s < ns; s++) {
                if (ds[s].ended || ds[s].have >= want) continue;
                Block b; if (!in[s]->next(b)) { ds[s].ended = true; continue; }
                // (at most flight_limit() staging blocks wait for their copies, so the readers never run dry while this thread waits for a block)
                ds[s].append_async(g, b, batch + block); ds[s].drain(g, ds[s].flight.size() >= in[s]->flight_limit(), [&](const Block& x) { in[s]->release(x); });
                if (in[s]->drained()) ds[s].ended = true;
                more = true;
            }
        }
        for (int s = 0; s < ns; s++) ds[s].drain(g, true, [&](const Block& x) { in[s]->release(x); });      // every queued copy has landed
        trace_mark("compress: batch resident");
        bool final = ds[0].ended && (!two || ds[1].ended);
        rfq_encode_args a; memset(&a, 0, sizeof a);
        a.d_fq1 = ds[0].base(); a.n1 = ds[0].have; a.d_fq2 = two ? ds[1].base() : nullptr; a.n2 = two ? ds[1].have : 0; a.paired = paired;
        a.chunk_bases = (uint32_t)(std::max(100L, o.chunkKb) * 1000); a.emit_header = first ? 1 : 0;
        a.file_off1 = ds[0].file_off; a.file_off2 = ds[1].file_off;
        // one file of a pair is used up, the other goes on: FastqReaderPair::read stops with the shorter file (src/fastqreader.cpp:287-299), so when a final plan of
        // what is resident uses the short stream up, this batch ends the input - the rest of the longer file is never read (it used to be read, uploaded and, holding no
        // whole chunk, re-planned batch after batch)
        if (!final && two && ds[0].ended != ds[1].ended) {
            const int s0 = ds[0].ended ? 0 : 1; rfq_scan_result sf; a.final = 1; g.check(rfq_scan_batch(g.c, &a, &sf));
            if (!sf.input_ended && (s0 == 0 ? sf.consumed1 : sf.consumed2) == ds[s0].have) final = true;
        }
        a.final = final ? 1 : 0;
        // FastqReader::hasNoLineBreakAtEnd (src/fastqreader.cpp:31-46): known once the reader thread has seen
        // the end of the input; until then at least two full blocks (>= 2 MiB) follow this batch, so no chunk of it can reach the
        // reader's final 1 MiB block
        uint64_t th[2] = { UINT64_MAX, UINT64_MAX };
        for (int s = 0; s < ns; s++) if (in[s]->end_known()) { const uint64_t t = in[s]->total_bytes(); if (t) th[s] = nolb_threshold(t, in[s]->final_byte()); }
        a.nolb_from1 = th[0]; a.nolb_from2 = two ? th[1] : th[0];
        rfq_encode_result r; g.check(rfq_encode_batch(g.c, &a, &r));
        if (ver && r.n_chunks && ver->wanted())
            ver->check(g, r, first, final || r.input_ended, two, ds[0].base(), final ? ds[0].have : r.consumed1, two ? ds[1].base() : nullptr,
                    two ? (final ? ds[1].have : r.consumed2) : 0, ds[0].file_off, ds[1].file_off);
        trace_mark("compress: batch encoded");
        if (r.rfq_len) { out.write_dev(r.d_rfq, r.rfq_len); if (r.n_chunks) first = false; }
        if (final || r.input_ended) break;            // input_ended: the reader stopped at an empty line (src/fastqreader.cpp:180-191)
        if (r.consumed1 == 0 && (!two || r.consumed2 == 0)) { want = std::max(ds[0].have, ds[1].have) + batch; continue; }   // a chunk larger than the batch: read on
        ds[0].advance(g, r.consumed1, batch + block); if (two) ds[1].advance(g, r.consumed2, batch + block);
        want = batch;
    }
    trace_mark("compress: all batches done");
    out.finish();                                      // (an input without reads leaves an empty output, like the reference)
    trace_mark("compress: output closed");
    delete ver;
    for (int s = 0; s < ns; s++) { ds[s].free_all(g); delete in[s]; }
}

// Chunk-parallel compress of one input over several GPUs (a host work queue, no collective), PER-DEVICE INGESTION (the first
// version sent all text over the first device's link and planned it there).  The input is dealt out batch by batch, round robin: batch k of every stream
// is uploaded straight to device k mod D, over that device's own link, and never visits another one.  Chunks are cut greedily from the start of the file
// (Repaq::compress, src/repaq.cpp:546-553), so a batch ends inside a chunk: its worker plans its own text (rfq_scan_batch: where every chunk ends), encodes
// the whole chunks and PUBLISHES the rest - the carry, a chunk's worth of text at most - which the worker of batch k + 1 pulls in front of its own bytes
// (rfq_copy_peer: the only text that crosses between devices).  The plan is a chain of scans (~1 ms per GB each) with the encodes running beside it; only
// the <= 272-byte header (made by the first batch that holds a whole chunk) and the carries cross between workers.
struct MBatch {
    uint64_t seq = 0; int w = 0;
    void* buf[2] = { nullptr, nullptr }; size_t cap[2] = { 0, 0 }, room = 0, n[2] = { 0, 0 };   // device buffers: [room for the carry][the batch's own bytes]
    uint64_t file_off[2] = { 0, 0 }, th[2] = { UINT64_MAX, UINT64_MAX }; bool last = false;
    bool eof[2] = { false, false };      // the stream's file ended in this batch or before it: its text here (carry included) is all that is left of it
    // what this batch leaves to the next one (set by its worker, before it encodes)
    bool carry_ready = false, carry_taken = false, ended = false, hdr_promised = false;
    rfq_ctx* owner = nullptr; const uint8_t* carry_ptr[2] = { nullptr, nullptr }; size_t carry_n[2] = { 0, 0 }; uint64_t carry_off[2] = { 0, 0 };
};
static void do_compress_multi(const Options& o) {
    const bool two = !o.in2.empty(); const int ns = two ? 2 : 1, D = (int)o.devices.size();
    const int paired = two ? RFQ_PE_TWO_FILES : (o.interleaved ? RFQ_PE_INTERLEAVED : RFQ_SE);
    const uint32_t chunk_bases = (uint32_t)(std::max(100L, o.chunkKb) * 1000);
    const size_t block = o.block(), batch = std::max(o.batchBytes, block);
    const size_t room = std::max<size_t>((size_t)4 << 20, 4 * (size_t)chunk_bases);          // (a chunk of b bases is ~2.4 b bytes of text per stream)
    Prefetcher* in[2] = { new Prefetcher((Gpu*)nullptr, o.in1, block, o.ioThreads), two ? new Prefetcher((Gpu*)nullptr, o.in2, block, o.ioThreads) : nullptr };
    std::vector<std::unique_ptr<Gpu>> gin((size_t)D);                          // ingestion contexts, one per device (the workers' own contexts are busy encoding)
    gin[0].reset(new Gpu(o.devices[0]));
    for (int s = 0; s < ns; s++) in[s]->attach(*gin[0]);                       // (the staging blocks are page-locked once; every device copies out of them)
    std::mutex mu; std::condition_variable cv;
    std::map<uint64_t, std::shared_ptr<MBatch>> batches; uint64_t n_batches = 0; bool all_ingested = false;
    // a worker has seen the end of the input in front of the files' ends - an empty line (src/fastqreader.cpp:180-191), or the shorter file of a pair used up
    // (FastqReaderPair::read stops there, :287-299): nothing behind it is ever read, so the ingestion stops uploading (it read and uploaded the whole rest)
    bool stop_ingest = false;
    std::vector<uint8_t> header; bool header_ready = false;
    std::map<uint64_t, std::vector<uint8_t>> done; uint64_t next_write = 0;
    ByteSink sink; sink.open(o.out1, o);
    std::thread writer([&] {
        for (;;) {
            std::vector<uint8_t> buf;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done.count(next_write) || (all_ingested && next_write == n_batches); });
              if (!done.count(next_write)) return;
              buf = std::move(done[next_write]); done.erase(next_write); next_write++; cv.notify_all(); }
            if (!buf.empty()) sink.write(buf.data(), buf.size());
        }
    });
    std::vector<std::thread> workers;
    for (int w = 0; w < D; w++) workers.emplace_back([&, w] {
        Gpu g(o.devices[(size_t)w]); bool have_hdr = false;
        std::unique_ptr<Verifier> ver; if (o.completeCheck || o.fastCheck) ver.reset(new Verifier(o, o.devices[(size_t)w]));
        for (uint64_t k = (uint64_t)w; ; k += (uint64_t)D) {
            std::shared_ptr<MBatch> b, prev;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return batches.count(k) || (all_ingested && k >= n_batches); });
              if (!batches.count(k)) break;
              b = batches[k];
              if (k) { cv.wait(lk, [&] { return batches.count(k - 1) && batches[k - 1]->carry_ready; }); prev = batches[k - 1]; } }
            auto publish = [&](bool ended, bool hdr, const uint8_t* p1, size_t n1, uint64_t o1, const uint8_t* p2, size_t n2, uint64_t o2) {
                std::unique_lock<std::mutex> lk(mu);
                b->ended = ended; b->hdr_promised = hdr; b->owner = g.c; b->carry_ptr[0] = p1; b->carry_n[0] = n1; b->carry_off[0] = o1; b->carry_ptr[1] = p2;
                        b->carry_n[1] = n2; b->carry_off[1] = o2;
                b->carry_ready = true; cv.notify_all();
            };
            // the image to the writer (in order), then the buffers go once the next batch has taken its carry
            auto finish = [&](std::vector<uint8_t>&& img) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return done.size() < 4 * (size_t)D || b->seq == next_write; });
                done[b->seq] = std::move(img); cv.notify_all();
                if (!b->last) cv.wait(lk, [&] { return b->carry_taken || (all_ingested && b->seq + 1 >= n_batches); });   // (no batch behind me: nobody will take a carry)
                lk.unlock();
                for (int s = 0; s < ns; s++) if (b->buf[s]) rfq_dev_free(g.c, b->buf[s]);
                lk.lock(); batches.erase(b->seq >= 2 ? b->seq - 2 : UINT64_MAX); cv.notify_all();
            };
            if (prev && prev->ended) {                                         // the reader stopped at an empty line in an earlier batch: nothing of this one is read
                { std::unique_lock<std::mutex> lk(mu); prev->carry_taken = true; cv.notify_all(); }
                publish(true, prev->hdr_promised, nullptr, 0, 0, nullptr, 0, 0);
                finish(std::vector<uint8_t>());
                continue;
            }
            // the carry of the batch in front, pulled in front of my own bytes
            size_t cn[2] = { prev ? prev->carry_n[0] : 0, (prev && two) ? prev->carry_n[1] : 0 };
            const uint8_t* tx[2] = { nullptr, nullptr }; size_t tn[2] = { 0, 0 }; uint64_t toff[2] = { 0, 0 };
            for (int s = 0; s < ns; s++) {
                size_t lead = b->room;                                         // my own bytes start `lead` bytes into the buffer
                // (a carry larger than the room left for it - a chunk of > 4 x chunk_bases bytes, or several batches without a whole chunk: a larger buffer)
                if (cn[s] > lead) {
                    const size_t nc = cn[s] + b->n[s] + 64; void* nb = g.dev(nc); trace_mark("compress: carry larger than its room, buffer grown");
                    if (b->n[s]) g.check(rfq_copy_d2d(g.c, (uint8_t*)nb + cn[s], (uint8_t*)b->buf[s] + lead, b->n[s]));
                    g.check(rfq_dev_free(g.c, b->buf[s])); b->buf[s] = nb; b->cap[s] = nc; lead = cn[s];
                }
                if (cn[s]) g.check(rfq_copy_peer(g.c, (uint8_t*)b->buf[s] + lead - cn[s], prev->owner, prev->carry_ptr[s], cn[s]));
                tx[s] = (const uint8_t*)b->buf[s] + lead - cn[s]; tn[s] = cn[s] + b->n[s]; toff[s] = prev ? prev->carry_off[s] : 0;
            }
            if (prev) { std::unique_lock<std::mutex> lk(mu); prev->carry_taken = true; cv.notify_all(); }
            const bool hdr_before = prev && prev->hdr_promised;
            rfq_encode_args a; memset(&a, 0, sizeof a);
            a.d_fq1 = tx[0]; a.n1 = tn[0]; a.d_fq2 = two ? tx[1] : nullptr; a.n2 = two ? tn[1] : 0; a.paired = paired; a.chunk_bases = chunk_bases;
            a.file_off1 = toff[0]; a.file_off2 = toff[1]; a.nolb_from1 = b->th[0]; a.nolb_from2 = two ? b->th[1] : b->th[0];
            size_t en[2] = { tn[0], tn[1] }; bool encode_final = b->last, nothing = false, ended = false;
            // One file of a pair is used up (all that is left of it is in front of me) and the other one goes on: the pair reader stops with the shorter file, so once
            // every record of the short one finds its mate in MY text this batch ends the input - and the batches behind it, text of the longer file only, would
            // each hold no whole chunk and hand their whole text on as carry (quadratic, unbounded).  A final plan tells: it must use the short stream up.
            if (!b->last && two && (b->eof[0] != b->eof[1])) {
                const int s0 = b->eof[0] ? 0 : 1;
                rfq_scan_result sf; a.final = 1; g.check(rfq_scan_batch(g.c, &a, &sf));
                if (!sf.input_ended && (s0 == 0 ? sf.consumed1 : sf.consumed2) == tn[s0]) {
                    encode_final = true; ended = true;
                    publish(true, true, nullptr, 0, 0, nullptr, 0, 0);
                    std::unique_lock<std::mutex> lk(mu); stop_ingest = true; cv.notify_all();
                }
            }
            if (!encode_final) {                                               // plan: where my last whole chunk ends; the rest is the next batch's
                rfq_scan_result sr; a.final = 0; g.check(rfq_scan_batch(g.c, &a, &sr));
                // (an empty line ends the input inside my text: all of it goes through the encode, which stops there)
                if (sr.input_ended) { ended = true; std::unique_lock<std::mutex> lk(mu); stop_ingest = true; cv.notify_all(); }
                else if (sr.n_chunks == 0) nothing = true;
                else { en[0] = (size_t)sr.h_end1[sr.n_chunks - 1]; en[1] = two ? (size_t)sr.h_end2[sr.n_chunks - 1] : 0; }
                if (ended) publish(true, true, nullptr, 0, 0, nullptr, 0, 0);
                else if (nothing) publish(false, hdr_before, tx[0], tn[0], toff[0], tx[1], tn[1], toff[1]);
                else publish(false, true, tx[0] + en[0], tn[0] - en[0], toff[0] + en[0], two ? tx[1] + en[1] : nullptr, two ? tn[1] - en[1] : 0, toff[1] + en[1]);
            }
            std::vector<uint8_t> img;
            if (!nothing) {
                if (hdr_before && !have_hdr) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return header_ready; }); lk.unlock();
                        g.check(rfq_set_header(g.c, header.data(), header.size())); have_hdr = true; }
                a.n1 = en[0]; a.n2 = two ? en[1] : 0; a.final = encode_final ? 1 : 0; a.flush_all = encode_final ? 0 : 1; a.emit_header = hdr_before ? 0 : 1;
                rfq_encode_result r; g.check(rfq_encode_batch(g.c, &a, &r));
                if (ver && r.n_chunks && ver->wanted()) ver->check(g, r, !hdr_before, encode_final || ended, two, tx[0], en[0], tx[1], two ? en[1] : 0, toff[0], toff[1]);
                img.resize(r.rfq_len); if (r.rfq_len) g.check(rfq_copy_d2h(g.c, img.data(), r.d_rfq, r.rfq_len));
                if (!hdr_before) {                                             // the header every later batch is coded under
                    uint8_t hb[RFQ_HEADER_MAX]; size_t hn = 0; have_hdr = true;
                    if (r.n_chunks) g.check(rfq_get_header(g.c, hb, &hn));
                    std::unique_lock<std::mutex> lk(mu); header.assign(hb, hb + hn); header_ready = true; cv.notify_all();
                }
            }
            finish(std::move(img));
        }
    });
    // ingestion: batch k of every stream to device k mod D
    for (uint64_t k = 0; ; k++) {
        const int w = (int)(k % (uint64_t)D);
        // (at most two batches per device wait for their encode)
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop_ingest || k < next_write + 2 * (uint64_t)D + 1; });
          if (stop_ingest) { all_ingested = true; cv.notify_all(); break; } }
        if (!gin[(size_t)w]) gin[(size_t)w].reset(new Gpu(o.devices[(size_t)w]));
        Gpu& g = *gin[(size_t)w];
        auto b = std::make_shared<MBatch>(); b->seq = k; b->w = w; b->room = room;
        bool ended[2] = { false, !two };
        for (int s = 0; s < ns; s++) { b->cap[s] = room + batch + block + 64; b->buf[s] = g.dev(b->cap[s]); b->file_off[s] = 0; }
        // staging blocks whose copies are queued (a reader must never run out of blocks: flight_limit)
        std::deque<std::pair<Block, int>> flight; size_t held[2] = { 0, 0 };
        auto land = [&] { g.check(rfq_copy_sync(g.c)); for (auto& f : flight) in[f.second]->release(f.first); flight.clear(); held[0] = held[1] = 0; };
        for (bool more = true; more; ) {
            more = false;
            for (int s = 0; s < ns; s++) {
                if (ended[s] || b->n[s] >= batch) continue;
                if (in[s]->drained()) { ended[s] = true; continue; }
                Block blk; if (!in[s]->next(blk)) { ended[s] = true; continue; }
                uint64_t t = 0; g.check(rfq_copy_h2d_async(g.c, (uint8_t*)b->buf[s] + room + b->n[s], blk.p, blk.n, &t)); b->n[s] += blk.n; flight.emplace_back(blk, s);
                if (++held[s] >= in[s]->flight_limit()) land();
                if (in[s]->drained()) ended[s] = true;
                more = true;
            }
        }
        land();
        b->last = ended[0] && ended[1]; b->eof[0] = ended[0]; b->eof[1] = two && ended[1];
        for (int s = 0; s < ns; s++) if (in[s]->end_known()) { const uint64_t t = in[s]->total_bytes(); if (t) b->th[s] = nolb_threshold(t, in[s]->final_byte()); }
        trace_mark("compress: batch resident");
        { std::unique_lock<std::mutex> lk(mu); batches[k] = b; n_batches = k + 1; if (b->last) all_ingested = true; cv.notify_all(); }
        if (b->last) break;
    }
    for (auto& t : workers) t.join();
    { std::unique_lock<std::mutex> lk(mu); cv.notify_all(); }
    writer.join(); sink.close();
    for (int s = 0; s < ns; s++) delete in[s];
}

// RfqChunk::read's chain (src/rfqchunk.cpp:161-228) on the host, as the blocks of the image go by on their way to the GPU: the format has no chunk
// index, chunk c + 1 is found from chunk c's 12-byte header (mSize, mReads, mFlags; mSize is short of the true size by what the writer's size bug
// leaves out, SURVEY.md App. C Q1: a function of the header's and the chunk's flags).  The offsets go to rfq_decode_batch as its optional chunk
// index, which takes the dependent walk off the device; every extent is verified there all the same, a table that does not verify is ignored.
struct ChunkWalker {
    std::vector<uint64_t> off;             // absolute offsets of the chunk starts found so far
    uint64_t fed = 0, next = 0;            // bytes seen; where the next chunk header starts
    uint8_t head[17]; int nhead = 0; uint8_t carry[12]; int ncarry = 0;
    uint32_t hf = 0; bool have_hdr = false, dead = false, ended = false;
    static uint32_t u32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
    void feed(const uint8_t* p, size_t n) {
        const uint64_t base = fed; fed += n;
        if (dead || ended) return;
        size_t i = 0;
        if (!have_hdr) {                                                     // RfqHeader::read (src/rfqheader.cpp:19-43): 17 bytes + the quality table
            while (nhead < 17 && i < n) head[nhead++] = p[i++];
            if (nhead < 17) return;
            hf = (uint32_t)head[10] | ((uint32_t)head[11] << 8); next = 17u + head[16]; have_hdr = true;
        }
        for (;;) {
            if (next + 12 > fed) {                                            // the header is not (all) here yet: keep what there is of it
                if (next < fed) { const uint64_t from = std::max(next, base); const size_t k = (size_t)(fed - from);
                        if (ncarry + k <= 12) { memcpy(carry + ncarry, p + (from - base), k); ncarry += (int)k; } else dead = true; }
                return;
            }
            uint8_t h[12];
            if (ncarry) { const size_t k = 12 - ncarry; memcpy(h, carry, ncarry); memcpy(h + ncarry, p + (next + ncarry - base), k); ncarry = 0; }
            else memcpy(h, p + (next - base), 12);
            const uint32_t ms = u32(h), reads = u32(h + 4), fl = (uint32_t)h[8] | ((uint32_t)h[9] << 8);
            if (reads == 0) { ended = true; return; }                        // a clean end (RfqChunk::read leaves mReads 0)
            const uint32_t half = (fl & (1u << 9)) ? reads / 2 : reads;      // C_PE_INTERLEAVED: lane / tile per pair (constants: rfq_common.h)
            long long total = (long long)ms;
            if (hf & 1u) total += (fl & (1u << 4)) ? 1 : (long long)half;                  // H_LANE, C_LANE_SAME
            if (!(hf & 2u)) total -= (fl & (1u << 5)) ? 2 : 2ll * half;                    // H_TILE, C_TILE_SAME
            if (!(hf & 16u)) total -= (fl & (1u << 2)) ? 1 : (long long)reads;             // H_NAME2, C_NAME2_LEN_SAME
            if (total < 18 || reads > 0x1000000u || fl >= 0x1000u) { dead = true; return; }
            off.push_back(next); next += (uint64_t)total;
        }
    }
    // chunk index of the batch [b0, b0 + n): offsets relative to b0 of every chunk that lies wholly inside, then the end of the last one
    bool table(uint64_t b0, size_t n, std::vector<uint64_t>& t) const {
        t.clear(); if (dead) return false;
        auto it = std::lower_bound(off.begin(), off.end(), b0);
        for (; it != off.end(); ++it) { const uint64_t end = (it + 1 != off.end()) ? *(it + 1) : next; if (end > b0 + n || (it + 1 == off.end() && next > fed)) break;
                if (t.empty()) t.push_back(*it - b0); t.push_back(end - b0); }
        return t.size() >= 2;
    }
};

// Streaming decoder: .rfq blocks -> rfq_decode_batch over whole chunks -> device text handed to `emit_dev(out1, n1, out2, n2)`
struct DecodeTotals { uint64_t reads = 0, bases = 0; };
static DecodeTotals decode_stream(Gpu& g, const Options& o, const std::string& path, bool split,
                                  const std::function<void(const uint8_t*, size_t, const uint8_t*, size_t)>& emit_dev,
                                  const std::function<void(const rfq_decode_result&)>& on_batch = nullptr) {
    // image bytes per call ~ 1/8 of the text batch: .rfq is 7-25 % of its FASTQ, so the text of one call is about one batch
    const size_t batch = std::max<size_t>(o.batchBytes / 8, (size_t)1 << 16), block = std::min(batch, o.block());
    Prefetcher in(g, path, block, o.ioThreads);
    DevStream ds; bool first = true; size_t want = batch; DecodeTotals tot; ChunkWalker walker;
    for (;;) {
        while (!ds.ended && ds.have < want) {
            Block b; if (!in.next(b)) { ds.ended = true; break; }
            walker.feed(b.p, b.n);                                            // (the chunk headers, while the block is on its way)
            ds.append(g, b, batch + block); in.release(b);
            if (in.drained()) ds.ended = true;
        }
        trace_mark("decode: batch resident");
        rfq_decode_args a; memset(&a, 0, sizeof a);
        std::vector<uint64_t> tab;
        if (walker.table(ds.file_off, ds.have, tab)) { a.h_chunk_off = tab.data(); a.n_chunk_off = (uint32_t)(tab.size() - 1); }
        a.d_rfq = ds.base(); a.n = ds.have; a.has_header = first ? 1 : 0; a.split_pe = split ? 1 : 0; a.final = ds.ended ? 1 : 0;
                a.bug_compat = (o.bugCompat && o.decompress) ? 1 : 0;
        rfq_decode_result r; g.check(rfq_decode_batch(g.c, &a, &r));
        first = false;
        trace_mark("decode: batch decoded");
        tot.reads += r.n_reads; tot.bases += r.n_bases;
        if (on_batch) on_batch(r);
        if (r.n1 || r.n2) emit_dev(r.d_fq1, r.n1, r.d_fq2, r.n2);
        if (ds.ended) break;
        if (r.consumed == 0) { want = ds.have + batch; continue; }            // not one whole chunk yet
        ds.advance(g, r.consumed, batch + block); want = batch;
    }
    trace_mark("decode: all batches done");
    ds.free_all(g);
    return tot;
}
// Repaq::decompress / decompressPE (src/repaq.cpp:262-417)
static void do_decompress(const Options& o) {
    Gpu g(o.device, true);
    const bool split = !o.out2.empty();
    AsyncWriter w1(g, o.out1, o); AsyncWriter* w2 = split ? new AsyncWriter(g, o.out2, o) : nullptr;
    decode_stream(g, o, o.in1, split, [&](const uint8_t* d1, size_t n1, const uint8_t* d2, size_t n2) {
        if (n1) w1.write_dev(d1, n1);
        if (split && n2) w2->write_dev(d2, n2);
    });
    w1.finish(); if (w2) { w2->finish(); delete w2; }
    trace_mark("decompress: outputs closed");
}

// Chunk-parallel decompress of one image over several GPUs (--devices a,b,... with -d).  The main thread streams the image, walks the chunk
// headers as the blocks go by (ChunkWalker) and deals RANGES of whole chunks - about one batch of text each - into a queue; one worker per device
// PULLS the next range, uploads it through its OWN device's link (per-device ingestion: nothing crosses between the GPUs), decodes it with the
// range's chunk index and hands the text to an ordered writer.  Only the <= 272-byte header is shared.
struct DecItem { uint64_t seq = 0; std::vector<uint8_t> bytes; std::vector<uint64_t> tab; bool final = false; };
static void do_decompress_multi(const Options& o) {
    if (o.bugCompat) error_exit("--bug_compat follows the reference's loop from chunk to chunk: use a single device");
    const bool split = !o.out2.empty();
    Gpu gs(o.devices[0]);                                                    // (page-locked staging blocks of the reader)
    const size_t target = std::max<size_t>(o.batchBytes / 8, (size_t)1 << 16), block = std::min(target, o.block());
    Prefetcher in(gs, o.in1, block, o.ioThreads);
    std::mutex mu; std::condition_variable cv;
    std::deque<DecItem> queue; bool no_more = false; std::vector<uint8_t> header; bool header_ready = false;
    // a range's text leaves its device into PAGE-LOCKED buffers (two per worker, handed to the writer and back): pageable vectors - 8 GB of fresh pages and a
    // staged copy - were most of the first version's 2.5 s on the 2 x 4 GB input
    struct OutBuf { uint8_t* p1 = nullptr; size_t c1 = 0; uint8_t* p2 = nullptr; size_t c2 = 0; size_t n1 = 0, n2 = 0; bool busy = false; };
    std::map<uint64_t, OutBuf*> done; uint64_t next_write = 0, total_items = 0; bool all_queued = false;
    ByteSink s1; s1.open(o.out1, o); ByteSink s2; if (split) s2.open(o.out2, o);
    std::thread writer([&] {
        for (;;) {
            OutBuf* t = nullptr;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done.count(next_write) || (all_queued && next_write == total_items); });
              if (all_queued && next_write == total_items) return;
              t = done[next_write]; done.erase(next_write); next_write++; cv.notify_all(); }
            if (t->n1) s1.write(t->p1, t->n1);
            if (split && t->n2) s2.write(t->p2, t->n2);
            { std::unique_lock<std::mutex> lk(mu); t->busy = false; cv.notify_all(); }
        }
    });
    std::vector<std::thread> workers;
    for (size_t w = 0; w < o.devices.size(); w++) workers.emplace_back([&, w] {
        Gpu g(o.devices[w]); void* d = nullptr; size_t cap = 0; bool have_hdr = false; OutBuf ob[2]; int turn = 0;
        for (;;) {
            DecItem it;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !queue.empty() || no_more; }); if (queue.empty()) break; it = std::move(queue.front());
                    queue.pop_front(); cv.notify_all(); }
            if (!have_hdr) { { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return header_ready; });
                    } g.check(rfq_set_header(g.c, header.data(), header.size())); have_hdr = true; }
            if (cap < it.bytes.size() + 64) { if (d) rfq_dev_free(g.c, d); cap = it.bytes.size() + it.bytes.size() / 4 + 64; d = g.dev(cap); }
            g.check(rfq_copy_h2d(g.c, d, it.bytes.data(), it.bytes.size()));
            rfq_decode_args a; memset(&a, 0, sizeof a);
            a.d_rfq = (const uint8_t*)d; a.n = it.bytes.size(); a.has_header = 0; a.split_pe = split ? 1 : 0; a.final = it.final ? 1 : 0;
            // (no table: what the host's walk could not index - the library finds the chunks itself, like the one-device path)
            a.h_chunk_off = it.tab.size() >= 2 ? it.tab.data() : nullptr; a.n_chunk_off = it.tab.size() >= 2 ? (uint32_t)(it.tab.size() - 1) : 0u;
            rfq_decode_result r; g.check(rfq_decode_batch(g.c, &a, &r));
            if (it.tab.size() >= 2 && r.consumed != it.bytes.size()) error_exit("internal: a dealt range does not decode as whole chunks");
            OutBuf& t = ob[turn]; turn ^= 1;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !t.busy; }); }               // (the writer is done with what this buffer held two ranges ago)
            if (t.c1 < r.n1) { if (t.p1) rfq_host_free(g.c, t.p1); t.c1 = r.n1 + r.n1 / 4 + 4096; t.p1 = g.pinned(t.c1); }
            if (split && t.c2 < r.n2) { if (t.p2) rfq_host_free(g.c, t.p2); t.c2 = r.n2 + r.n2 / 4 + 4096; t.p2 = g.pinned(t.c2); }
            t.n1 = r.n1; t.n2 = split ? r.n2 : 0;
            if (r.n1) g.check(rfq_copy_d2h(g.c, t.p1, r.d_fq1, r.n1));
            if (split && r.n2) g.check(rfq_copy_d2h(g.c, t.p2, r.d_fq2, r.n2));
            std::unique_lock<std::mutex> lk(mu); t.busy = true; done[it.seq] = &t; cv.notify_all();
        }
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !ob[0].busy && !ob[1].busy; }); }   // (the writer still reads them)
        for (auto& t : ob) { if (t.p1) rfq_host_free(g.c, t.p1); if (t.p2) rfq_host_free(g.c, t.p2); }
        if (d) rfq_dev_free(g.c, d);
    });
    // the dealer: bytes not dealt yet start at a chunk boundary (pend_off) - or, at first, at the file header
    ChunkWalker walker; std::vector<uint8_t> pend; uint64_t pend_off = 0, seq = 0; bool hdr_done = false, ended = false;
    auto deal = [&](bool last) {
        if (!hdr_done) {
            if (!walker.have_hdr || walker.fed < 17u + walker.head[16]) { if (last && walker.fed) error_exit("Not a valid repaq file!"); return; }
            const size_t hl = 17u + walker.head[16];
            { std::unique_lock<std::mutex> lk(mu); header.assign(pend.begin(), pend.begin() + hl); header_ready = true; cv.notify_all(); }
            // (validates it: the reference's messages for a foreign / newer file)
            { Gpu& g = gs; g.check(rfq_set_header(g.c, header.data(), header.size())); }
            pend.erase(pend.begin(), pend.begin() + hl); pend_off = hl; hdr_done = true;
        }
        // what the host's walk cannot index (an implausible chunk header, bytes behind the chain's end): not an error here - the one-device path copes with such
        // images - but one last range without a table: rfq_decode_batch walks it on the device and decides
        auto deal_rest = [&]() {
            if (pend.empty()) return;
            DecItem it; it.seq = seq++; it.bytes.assign(pend.begin(), pend.end()); it.final = true;
            pend_off += pend.size(); pend.clear();
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return queue.size() < 2 * o.devices.size(); }); queue.push_back(std::move(it)); cv.notify_all(); }
        };
        if (walker.dead && !last) return;                                    // (the rest of the image is collected and goes as one range)
        if (walker.dead) { deal_rest(); return; }
        for (;;) {
            // the chunk ends known so far that lie inside pend: cut behind the last one within `target` bytes (at least one chunk; everything at the end)
            auto lo = std::lower_bound(walker.off.begin(), walker.off.end(), pend_off);
            if (lo == walker.off.end()) break;
            std::vector<uint64_t> tab; uint64_t e = pend_off;
            for (auto it = lo; it != walker.off.end(); ++it) {
                const uint64_t end = (it + 1 != walker.off.end()) ? *(it + 1) : walker.next;
                if (end > pend_off + pend.size()) break;
                if (tab.empty()) tab.push_back(*it - pend_off);
                tab.push_back(end - pend_off); e = end;
                if (!last && e - pend_off >= target) break;
            }
            if (tab.size() < 2) break;
            const bool all = last && e >= walker.next;                         // (behind it: nothing, or a tail too short to be a chunk)
            if (!last && e - pend_off < target) break;                         // (wait for more: ranges of about a batch of text)
            DecItem it; it.seq = seq++; it.bytes.assign(pend.begin(), pend.begin() + (e - pend_off)); it.tab = std::move(tab); it.final = all;
            pend.erase(pend.begin(), pend.begin() + (e - pend_off)); pend_off = e;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return queue.size() < 2 * o.devices.size(); }); queue.push_back(std::move(it)); cv.notify_all(); }
            if (all) break;
        }
        // (a tail the chain does not cover: the device's walk says whether it is a chunk, a cut one, or padding)
        if (last && !pend.empty() && pend.size() >= 18) deal_rest();
    };
    while (!ended) {
        Block b; if (!in.next(b)) { ended = true; break; }
        walker.feed(b.p, b.n); pend.insert(pend.end(), b.p, b.p + b.n); in.release(b);
        if (in.drained()) ended = true;
        deal(ended);
    }
    deal(true);
    { std::unique_lock<std::mutex> lk(mu); no_more = true; total_items = seq; all_queued = true; header_ready = true; cv.notify_all(); }
    for (auto& t : workers) t.join();
    writer.join(); s1.close(); if (split) s2.close();
}

// ---- compare mode (src/repaq.cpp:36-259): decode on the GPU, compare read by read with the FASTQ text, same JSON
static void report(const Options& o, bool passed, const std::string& msg, long fqReads, long fqBases, long rfqReads, long rfqBases) {   // :235-259
    std::string j = "{\n";
    j += passed ? "\t\"result\":\"passed\",\n" : "\t\"result\":\"failed\",\n";
    j += "\t\"msg\":\"" + msg + "\",\n";
    j += "\t\"fastq_reads\":" + std::to_string(fqReads) + ",\n\t\"rfq_reads\":" + std::to_string(rfqReads) + ",\n";
    j += "\t\"fastq_bases\":" + std::to_string(fqBases) + ",\n\t\"rfq_bases\":" + std::to_string(rfqBases) + "\n}\n";
    if (!o.json.empty()) { FILE* f = fopen(o.json.c_str(), "wb"); if (!f) error_exit("Failed to open file for writing: " + o.json); fwrite(j.data(), 1, j.size(), f);
            fclose(f); }
    fputs(j.c_str(), stdout); fflush(stdout);
}
static void do_compare(const Options& o) {
    Gpu g(o.device, true);
    const bool pe = !o.in2.empty(); const int ns = pe ? 2 : 1;
    // Fast path, on the device: every decoded batch is compared byte for byte with the same span of the FASTQ
    // text uploaded beside it (rfq_compare_bytes); identical bytes mean identical reads, so only counters move.  The first batch
    // that differs (a real mismatch, or text the reader would have normalised: "\r\n", blank lines) hands both sides over to the
    // record-by-record comparison below, which words the reference's message; so does the end of the image, for the
    // "FASTQ has more reads" test.
    const size_t block = o.block();
    Prefetcher* pf[2] = { new Prefetcher(g, o.in1, block, o.ioThreads), pe ? new Prefetcher(g, o.in2, block, o.ioThreads) : nullptr };
    DevStream fs[2]; bool fast = true;
    long fqReads = 0, fqBases = 0, rfqReads = 0, rfqBases = 0; bool reported = false;
    TextCursor dec[2], fq[2];
    std::mutex hm; std::condition_variable hcv; bool handed = false;
    auto handover = [&] {                                                     // the FASTQ side from here on belongs to the main thread
        for (int s = 0; s < ns; s++) {
            fq[s].off0 = fs[s].file_off;
            if (fs[s].have) { fq[s].t.resize(fs[s].have); g.check(rfq_copy_d2h(g.c, fq[s].t.data(), fs[s].base(), fs[s].have)); }
            fs[s].free_all(g); fs[s] = DevStream();
        }
        fast = false;
        std::unique_lock<std::mutex> lk(hm); handed = true; hcv.notify_all();
    };
    // decoded side: a producer thread decodes batch after batch; once off the fast path it fills two bounded text queues
    struct Q { std::mutex mu; std::condition_variable cv; std::deque<std::vector<uint8_t>> q; bool done = false, discard = false; } dq[2];
    std::thread producer([&] {
        rfq_decode_result cur; memset(&cur, 0, sizeof cur);
        decode_stream(g, o, o.rfqCompare, pe, [&](const uint8_t* d1, size_t n1, const uint8_t* d2, size_t n2) {
            const uint8_t* dp[2] = { d1, d2 }; const size_t dn[2] = { n1, pe ? n2 : 0 };
            if (fast) {
                bool same = true;
                for (int s = 0; s < ns && same; s++) {
                    while (!fs[s].ended && fs[s].have < dn[s]) {
                        Block b; if (!pf[s]->next(b)) { fs[s].ended = true; break; }
                        fs[s].append(g, b); pf[s]->release(b);
                        if (pf[s]->drained()) fs[s].ended = true;
                    }
                    if (!same || fs[s].have < dn[s]) { same = false; break; }
                    uint64_t at = 0; g.check(rfq_compare_bytes(g.c, dp[s], fs[s].base(), dn[s], &at));
                    same = at == dn[s];
                    // text that stops without a line break (the file's last read, NO_LINE_BREAK bit): whether the FASTQ ends there too,
                    // or goes on with a break or with more of the line, is for the record cutter to say
                    if (same && dn[s]) { uint8_t last = 0; g.check(rfq_copy_d2h(g.c, &last, dp[s] + dn[s] - 1, 1)); same = last == '\n'; }
                }
                if (same) {
                    rfqReads += (long)cur.n_reads; fqReads += (long)cur.n_reads; rfqBases += (long)cur.n_bases; fqBases += (long)cur.n_bases;
                    for (int s = 0; s < ns; s++) fs[s].advance(g, dn[s], block);
                    return;
                }
                handover();
            }
            for (int s = 0; s < 2; s++) if (dn[s]) {
                { std::unique_lock<std::mutex> lk(dq[s].mu); if (dq[s].discard) continue; }
                std::vector<uint8_t> v(dn[s]); g.check(rfq_copy_d2h(g.c, v.data(), dp[s], dn[s]));
                std::unique_lock<std::mutex> lk(dq[s].mu); dq[s].cv.wait(lk, [&] { return dq[s].q.size() < 4 || dq[s].discard; });
                if (!dq[s].discard) dq[s].q.push_back(std::move(v));
                dq[s].cv.notify_all();
            }
        }, [&](const rfq_decode_result& r) { cur = r; });
        if (fast) handover();
        for (int s = 0; s < 2; s++) { std::unique_lock<std::mutex> lk(dq[s].mu); dq[s].done = true; dq[s].cv.notify_all(); }
    });
    for (int s = 0; s < 2; s++) dec[s].refill = [&dq, s](TextCursor& c) {
        std::unique_lock<std::mutex> lk(dq[s].mu); dq[s].cv.wait(lk, [&] { return !dq[s].q.empty() || dq[s].done; });
        if (dq[s].q.empty()) { c.ended = true; return; }
        c.t.insert(c.t.end(), dq[s].q.front().begin(), dq[s].q.front().end()); dq[s].q.pop_front(); dq[s].cv.notify_all();
    };
    for (int s = 0; s < ns; s++) fq[s].refill = [&pf, s](TextCursor& c) {
        Block b; if (!pf[s]->next(b)) { c.ended = true; return; }
        c.t.insert(c.t.end(), b.p, b.p + b.n); pf[s]->release(b);
    };
    { std::unique_lock<std::mutex> lk(hm); hcv.wait(lk, [&] { return handed; }); }
    static const char* what[4] = { "name", "sequence", "strand", "quality" };
    // compare (:36-128) counts and words in reads; comparePE (:130-233) draws a whole pair from the two files when it meets the
    // first read of a pair (FastqReaderPair::read: NULL as soon as either file is out of reads) and words in pairs, rfqReads / 2
    const std::string unit = pe ? " pair. " : " read. ", units = pe ? " pairs" : " reads";
    auto cnt = [&](long reads) { return std::to_string(pe ? reads / 2 : reads); };
    Rec pair[2]; bool have_pair = false;
    for (;;) {
        Rec r; const bool second = pe && (rfqReads & 1);
        if (!dec[second ? 1 : 0].next(r)) break;
        rfqReads++; rfqBases += (long)r.f[1].size();
        if (!second) have_pair = fq[0].next(pair[0]) && (!pe || fq[1].next(pair[1]));
        if (!have_pair) {
            report(o, false, "The RFQ file has more reads than the FASTQ file. The RFQ file has >= " + cnt(rfqReads) + units + ", while the FASTQ file only has " + cnt(fqReads) + units, fqReads, fqBases, rfqReads, rfqBases);
            reported = true; break;
        }
        const Rec& q = pair[second ? 1 : 0];
        fqReads++; fqBases += (long)q.f[1].size();
        for (int k = 0; k < 4 && !reported; k++) if (r.f[k] != q.f[k]) {
            report(o, false, std::string("The RFQ file and FASTQ file have different ") + what[k] + " in the " + cnt(rfqReads) + unit + r.f[k] + " | " + q.f[k], fqReads,
                    fqBases, rfqReads, rfqBases);
            reported = true;
        }
        if (reported) break;
    }
    if (!reported) {
        Rec q;
        if (fq[0].next(q) && (!pe || fq[1].next(q))) {
            fqReads++;
            report(o, false, "The FASTQ file has more reads than the RFQ file. The FASTQ file has >= " + cnt(fqReads) + units + ", while the RFQ file only has " + cnt(rfqReads) + units, fqReads, fqBases, rfqReads, rfqBases);
        } else report(o, true, "", fqReads, fqBases, rfqReads, rfqBases);
    }
    // a failed compare stops consuming early: let the producer run to its end without queueing
    for (int s = 0; s < 2; s++) { std::unique_lock<std::mutex> lk(dq[s].mu); dq[s].discard = true; dq[s].q.clear(); dq[s].cv.notify_all(); }
    producer.join();
    for (int s = 0; s < ns; s++) delete pf[s];
}

static void usage() {
    fputs("repaq_hip: repack FASTQ to .rfq on an MI355X (repaq v0.5.1 compatible)\n"
          "usage: repaq_hip [-c|-d|-p] -i in1 [-I in2] -o out1 [-O out2] [-k chunk_kb] [--stdin] [--stdout] [--interleaved_in]\n"
          "                 [-r rfq_to_compare] [-j json] [-t xz_threads] [-z level] [--device N | --devices a,b,...]\n"
          "                 [--batch_mb M] [--block_mb M] [--io_threads N] [--write_threads N] [--trace] [--bug_compat]\n"
          "       repaq_hip --serve : a resident process, one job (the arguments of a command line) per line of stdin\n"
          "       FASTQ may be .gz (written as blocked gzip - bgzip's layout, readable by every gzip tool - on many threads; a blocked .gz is\n"
          "       also read on many threads, any other through zlib); .rfq may be .rfq.xz (external xz)\n", stderr);
}
// one job: a command line as the reference's main() takes it (src/main.cpp:15-184)
static void run_job(int argc, char** argv) {
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
        else if (a == "-v" || a == "--verify") o.completeCheck = true;
        else if (a == "-f" || a == "--fast_verify") o.fastCheck = true;
        else if (a == "-t" || a == "--thread" || a.rfind("--thread=", 0) == 0) o.threads = atoi(val(i, "thread").c_str());
        else if (a == "-z" || a == "--compression" || a.rfind("--compression=", 0) == 0) o.compression = atoi(val(i, "compression").c_str());
        else if (a == "--device") o.device = atoi(val(i, "device").c_str());
        else if (a == "--devices" || a.rfind("--devices=", 0) == 0) { const std::string v = val(i, "devices"); size_t p0 = 0;
                while (p0 <= v.size()) { const size_t q = v.find(',', p0); const std::string t = v.substr(p0, q == std::string::npos ? std::string::npos : q - p0);
                if (!t.empty()) o.devices.push_back(atoi(t.c_str())); if (q == std::string::npos) break; p0 = q + 1; } }
        else if (a == "--bug_compat") o.bugCompat = true;
        else if (a == "--batch_mb") o.batchBytes = (size_t)atol(val(i, "batch_mb").c_str()) << 20;
        else if (a == "--block_mb") o.blockBytes = (size_t)atol(val(i, "block_mb").c_str()) << 20;
        else if (a == "--io_threads") o.ioThreads = std::max(1, atoi(val(i, "io_threads").c_str()));
        else if (a == "--write_threads") o.writeThreads = std::max(1, atoi(val(i, "write_threads").c_str()));
        else if (a == "--trace") { o.trace = true; g_trace = true; }
        else { usage(); error_exit("unknown option: " + a); }
    }
    if ((int)o.compress + (int)o.decompress + (int)o.compare > 1) error_exit("repaq can run in compress/decompress/compare mode, you can only choose any one mode.");
    const bool dec = o.decompress, cmp = o.compare, enc = !dec && !cmp;
    // src/main.cpp:100-113: STDIN / STDOUT override the file names
    if (enc && o.useStdout && !o.out1.empty()) { fprintf(stderr, "Output to STDOUT, ignore --out1 = %s\n", o.out1.c_str()); o.out1.clear(); }
    if (dec && o.useStdin && !o.in1.empty()) { fprintf(stderr, "Input from STDIN, ignore --in1 = %s\n", o.in1.c_str()); o.in1.clear(); }
    if (cmp && o.useStdin && !o.rfqCompare.empty()) { fprintf(stderr, "Input from STDIN, ignore --rfq_to_compare = %s\n", o.rfqCompare.c_str()); o.rfqCompare.clear(); }
    // Options::validate (src/options.cpp:36-111)
    if (o.in1.empty()) {
        if (!o.in2.empty()) error_exit("read2 input is specified by <in2>, but read1 input is not specified by <in1>");
        if (o.useStdin && !cmp) o.in1 = "/dev/stdin"; else if (!cmp) error_exit("Please specify input file by <in1>, or enable --stdin if you want to read STDIN");
    }
    if (o.out1.empty()) {
        if (!o.out2.empty()) error_exit("read2 output is specified by <out2>, but read1 output is not specified by <out1>");
        if (o.useStdout) o.out1 = "/dev/stdout"; else if (!cmp) error_exit("Please specify output file by <out1>, or enable --stdout if you want to read STDIN");
    }
    if (o.compression < 1 || o.compression > 9) error_exit("compression level (-z) should be 1 ~ 9");
    // src/main.cpp:123-131
    if ((ends_with(o.in1, ".xz") || ends_with(o.rfqCompare, ".xz")) && o.useStdin) error_exit("STDIN cannot be read when the input is a .xz file");
    if (ends_with(o.out1, ".xz") && o.useStdout) error_exit("STDOUT cannot be written when the output is a .xz file");
    const long cb = std::max(100L, o.chunkKb) * 1000;
    if (cb < 10000) error_exit("chunk size cannot be less than 10 kb");
    if (cb > 500000000) error_exit("chunk size cannot be greater than 500,000 kb");
    if (o.batchBytes < ((size_t)1 << 20)) o.batchBytes = (size_t)1 << 20;
    if (o.batchBytes > ((size_t)2 << 30)) o.batchBytes = (size_t)2 << 30;
    if (o.blockBytes < ((size_t)1 << 20)) o.blockBytes = (size_t)1 << 20;
    if (enc) {
        if (!o.out2.empty()) error_exit("In compress mode, only one RFQ output file is allowed, but you specified <out2>");
        if (ends_with(o.out1, ".fq") || ends_with(o.out1, ".fastq")) error_exit("In compress mode, the output should not be a FASTQ file. Expect a .rfq or .rfq.xz file, but got " + o.out1);
        if (ends_with(o.in1, ".rfq")) error_exit("In compress mode, the input should not be a RFQ file. Expect a .fq or .fq.gz file, but got " + o.in1);
        if (o.devices.size() > 1) do_compress_multi(o); else { if (o.devices.size() == 1) o.device = o.devices[0]; do_compress(o); }
    } else if (dec) {
        if (!o.in2.empty()) error_exit("In decompress mode, only one RFQ input file is allowed, but you specified <in2>");
        if (ends_with(o.in1, ".fq") || ends_with(o.in1, ".fastq")) error_exit("In decompress mode, the input should not be a FASTQ file. Expect a .rfq or .rfq.xz file, but got " + o.in1);
        if (ends_with(o.out1, ".rfq")) error_exit("In decompress mode, the output should not be a RFQ file. Expect a .fq or .fq.gz file, but got " + o.out1);
        if (o.devices.size() > 1) do_decompress_multi(o); else { if (o.devices.size() == 1) o.device = o.devices[0]; do_decompress(o); }
    } else {
        if (o.useStdin) o.rfqCompare = "/dev/stdin";
        if (o.rfqCompare.empty()) error_exit("In compare mode, you should specify the RFQ file to compare by <rfq_to_compare>");
        if (!o.out1.empty() || !o.out2.empty()) error_exit("In compare mode, you cannot specify the output by <out1> or <out2>");
        if (o.in1.empty()) error_exit("Please specify input file by <in1>, or enable --stdin if you want to read STDIN");
        do_compare(o);
    }
}
// --serve: a resident process.  Every line of stdin is one job - the arguments of a repaq_hip command line, blank-separated, "double quotes" around an argument
// that holds blanks - run one after the other in this process: the HIP runtime (0.2 - 0.33 s of every one-shot run: context, code objects) is paid once, and the main
// context of a device keeps its workspace from job to job.  A line on stderr per job: "[serve] job N: S s".  A job that fails ends the process with the reference's
// error text and status, like a one-shot run.  (--stdin jobs are refused: stdin is the job list.)
static int serve() {
    g_serve = true;
    char* line = nullptr; size_t cap = 0; long n; int job = 0;
    while ((n = getline(&line, &cap, stdin)) >= 0) {
        std::vector<std::string> tok; std::string cur; bool inq = false, any = false;
        for (long i = 0; i < n; i++) { const char ch = line[i];
            if (ch == '"') { inq = !inq; any = true; }
            else if (!inq && (ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r')) { if (any || !cur.empty()) tok.push_back(cur); cur.clear(); any = false; }
            else cur.push_back(ch); }
        if (any || !cur.empty()) tok.push_back(cur);
        if (tok.empty() || tok[0][0] == '#') continue;
        if (tok[0] == "repaq_hip" || ends_with(tok[0], "/repaq_hip")) tok.erase(tok.begin());
        for (auto& t : tok) if (t == "--stdin" || t == "--serve") error_exit("--serve: a job cannot read STDIN (it is the job list) or serve itself");
        std::vector<char*> av; std::string a0 = "repaq_hip"; av.push_back(&a0[0]); for (auto& t : tok) av.push_back(&t[0]);
        const auto t0 = std::chrono::steady_clock::now();
        run_job((int)av.size(), av.data());
        fflush(stdout);
        fprintf(stderr, "[serve] job %d: %.3f s\n", ++job, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); fflush(stderr);
    }
    free(line);
    return 0;
}
int main(int argc, char** argv) {
    if (argc == 1) { usage(); return 0; }
    if (argc == 2 && !strcmp(argv[1], "--version")) { printf("repaq_hip 0.5.1-compatible (%s)\n", rfq_version()); return 0; }
    if (argc == 2 && !strcmp(argv[1], "--serve")) serve(); else run_job(argc, argv);
    trace_mark("done");
    // every output is closed and flushed: leave without the HIP runtime's static teardown (~80 ms of unmapping at exit)
    fflush(stdout); fflush(stderr); _exit(0);
}
