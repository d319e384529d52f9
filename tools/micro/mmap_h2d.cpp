// Micro-benchmark: a file on tmpfs / in the page cache -> HBM without a CPU copy.  mmap the file, hipHostRegister pieces of the mapping (read-only), hipMemcpyAsync each
// piece; registration of piece i+1 runs on a second thread while piece i crosses the link.  Against: pread into page-locked blocks on T threads + copies (what repaq_hip does).
// build: hipcc -O2 -o /tmp/mmap_h2d tools/micro/mmap_h2d.cpp -lpthread ; run: /tmp/mmap_h2d <file> [piece_mb=256]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <vector>
#include <atomic>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    if (argc < 2) return 1;
    const size_t piece = (size_t)(argc > 2 ? atoi(argv[2]) : 256) << 20;
    int fd = open(argv[1], O_RDONLY); struct stat st; fstat(fd, &st); const size_t n = st.st_size;
    hipFree(0);
    void* d; if (hipMalloc(&d, n) != hipSuccess) return 2;
    hipStream_t s; hipStreamCreate(&s);
    for (int mode = 0; mode < 3; mode++) {
        const double t0 = now();
        uint8_t* m = (uint8_t*)mmap(nullptr, n, PROT_READ, MAP_SHARED | (mode == 2 ? MAP_POPULATE : 0), fd, 0);
        if (m == MAP_FAILED) { perror("mmap"); return 3; }
        const size_t np = (n + piece - 1) / piece; std::vector<int> ok(np, 0); std::atomic<size_t> ready{0};
        double treg = 0;
        std::thread reg([&] { for (size_t i = 0; i < np; i++) { const size_t off = i * piece, len = std::min(piece, n - off); const double a = now();
                                   hipError_t e = hipHostRegister(m + off, len, mode == 0 ? hipHostRegisterDefault : hipHostRegisterReadOnly); treg += now() - a; ok[i] = e == hipSuccess; if (!ok[i] && i == 0) fprintf(stderr, "register: %s\n", hipGetErrorString(e)); ready = i + 1; } });
        for (size_t i = 0; i < np; i++) { while (ready.load() <= i) std::this_thread::yield(); const size_t off = i * piece, len = std::min(piece, n - off);
                                          hipMemcpyAsync((uint8_t*)d + off, m + off, len, hipMemcpyHostToDevice, s); }
        hipStreamSynchronize(s); reg.join();
        const double t1 = now();
        int good = 0; for (int v : ok) good += v;
        printf("mode %d (%s%s): %zu MB in %.3f s = %.1f GB/s  (registered %d / %zu pieces, %.3f s inside hipHostRegister)\n", mode, mode == 0 ? "default" : "read-only", mode == 2 ? ", MAP_POPULATE" : "", n >> 20, t1 - t0, n / (t1 - t0) / 1e9, good, np, treg);
        for (size_t i = 0; i < np; i++) if (ok[i]) hipHostUnregister(m + i * piece);
        munmap(m, n);
    }
    // registration on several threads (pieces handed out in order), copies issued in order as pieces become ready
    for (int T : {2, 4, 8}) {
        const double t0 = now();
        uint8_t* m = (uint8_t*)mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
        const size_t np = (n + piece - 1) / piece; std::vector<std::atomic<int>> st_(np); for (auto& v : st_) v = 0; std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&] { for (;;) { const size_t i = next.fetch_add(1); if (i >= np) break; const size_t off = i * piece, len = std::min(piece, n - off);
                                                          st_[i] = hipHostRegister(m + off, len, hipHostRegisterReadOnly) == hipSuccess ? 1 : 2; } });
        for (size_t i = 0; i < np; i++) { while (st_[i].load() == 0) std::this_thread::yield(); const size_t off = i * piece, len = std::min(piece, n - off);
                                          hipMemcpyAsync((uint8_t*)d + off, m + off, len, hipMemcpyHostToDevice, s); }
        hipStreamSynchronize(s); for (auto& t : th) t.join();
        const double t1 = now();
        printf("register x%d threads, read-only: %.3f s = %.1f GB/s\n", T, t1 - t0, n / (t1 - t0) / 1e9);
        const double u0 = now(); for (size_t i = 0; i < np; i++) if (st_[i] == 1) hipHostUnregister(m + i * piece); munmap(m, n); printf("   (unregister + munmap %.3f s)\n", now() - u0);
    }
    // the driver's way: T pread threads into page-locked blocks, copies behind them
    for (int T : {8, 16}) { const size_t blk = 16u << 20; std::vector<void*> hb(2 * T); for (auto& p : hb) hipHostMalloc(&p, blk);
      const double t0 = now(); std::atomic<size_t> next{0}; const size_t nb = (n + blk - 1) / blk; std::vector<std::thread> th;
      for (int t = 0; t < T; t++) th.emplace_back([&, t] { hipStream_t q; hipStreamCreate(&q); int k = 0; for (;;) { const size_t i = next.fetch_add(1); if (i >= nb) break; void* b = hb[2 * t + (k++ & 1)]; const size_t off = i * blk, len = std::min(blk, n - off);
                                                          hipStreamSynchronize(q); size_t got = 0; while (got < len) { ssize_t r = pread(fd, (char*)b + got, len - got, off + got); if (r <= 0) break; got += r; } hipMemcpyAsync((uint8_t*)d + off, b, len, hipMemcpyHostToDevice, q); } hipStreamSynchronize(q); });
      for (auto& t : th) t.join();
      const double t1 = now(); printf("pread x%d + copies: %.3f s = %.1f GB/s\n", T, t1 - t0, n / (t1 - t0) / 1e9); }
    return 0;
}
