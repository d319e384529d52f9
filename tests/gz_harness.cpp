// Test harness for the host driver's blocked-gzip reader / writer (tests/test_gz_blocks.py): the driver's own source with its main renamed.
//   gz_harness w <file.gz> <piece> [level]   stdin  -> file, handed to the sink in pieces of <piece> bytes
//   gz_harness r <file.gz> <cap>             file   -> stdout, read from the source <cap> bytes at a time
#define main repaq_hip_main_
#include "../repaq_amd/csrc/host/repaq_hip_main.cpp"
#undef main
int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const std::string mode = argv[1], path = argv[2]; const size_t step = (size_t)atoll(argv[3]);
    if (mode == "w") {
        std::vector<uint8_t> text; uint8_t buf[1 << 16]; size_t k;
        while ((k = fread(buf, 1, sizeof buf, stdin)) > 0) text.insert(text.end(), buf, buf + k);
        Options o; o.ioThreads = 3; if (argc > 4) o.compression = atoi(argv[4]);
        ByteSink s; s.open(path, o);
        for (size_t at = 0; at < text.size(); at += step) s.write(text.data() + at, std::min(step, text.size() - at));
        s.close();
        return 0;
    }
    ByteSource r; if (!r.open(path, 3)) return 3;
    std::vector<uint8_t> buf(step);
    for (;;) { const size_t k = r.read(buf.data(), step); if (k && fwrite(buf.data(), 1, k, stdout) != k) return 4; if (k < step) break; }
    r.close();
    return 0;
}
