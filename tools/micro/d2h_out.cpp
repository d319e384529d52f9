// Micro-benchmark: HBM -> a file on tmpfs.  (a) what repaq_hip does: D2H into page-locked 16 MB pieces, T writer threads pwrite() them at their offsets; (b) the same pieces
// memcpy'ed by T threads into an mmap of the output, pre-sized with ftruncate (pwrite on ONE tmpfs file serialises on the inode lock; page faults do not); (c) pieces of the
// mapping page-locked (hipHostRegister) and the device copying straight into the page cache.
// build: hipcc -O2 -o /tmp/d2h_out tools/micro/d2h_out.cpp -lpthread ; run: /tmp/d2h_out <file> <MB>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#include <vector>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <deque>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    if (argc < 3) return 1;
    const size_t n = (size_t)atol(argv[2]) << 20, piece = 16u << 20, np = (n + piece - 1) / piece;
    hipFree(0); void* d; if (hipMalloc(&d, n) != hipSuccess) return 2; hipMemset(d, 0x41, n);
    hipStream_t s; hipStreamCreate(&s);
    for (int T : {4, 8, 16}) {
        for (int mode = 0; mode < 2; mode++) {
            unlink(argv[1]); int fd = open(argv[1], O_RDWR | O_CREAT | O_TRUNC, 0644);
            const int NB = T + 3; std::vector<void*> hb(NB); for (auto& p : hb) hipHostMalloc(&p, piece);
            const double t0 = now();
            uint8_t* m = nullptr; if (mode == 1) { if (ftruncate(fd, n)) return 3; m = (uint8_t*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); if (m == MAP_FAILED) { perror("mmap"); return 4; } }
            std::mutex mu; std::condition_variable cv; std::deque<std::pair<size_t, void*>> q; std::vector<void*> freeb(hb); bool done = false;
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([&] { for (;;) { std::pair<size_t, void*> it; { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !q.empty() || done; }); if (q.empty()) return; it = q.front(); q.pop_front(); }
                    const size_t off = it.first * piece, len = std::min(piece, n - off);
                    if (mode == 0) { size_t w = 0; while (w < len) { ssize_t k = pwrite(fd, (char*)it.second + w, len - w, off + w); if (k <= 0) break; w += k; } } else memcpy(m + off, it.second, len);
                    std::unique_lock<std::mutex> lk(mu); freeb.push_back(it.second); cv.notify_all(); } });
            for (size_t i = 0; i < np; i++) { void* b; { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !freeb.empty(); }); b = freeb.back(); freeb.pop_back(); }
                    const size_t off = i * piece, len = std::min(piece, n - off); hipMemcpyAsync(b, (uint8_t*)d + off, len, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
                    std::unique_lock<std::mutex> lk(mu); q.emplace_back(i, b); cv.notify_all(); }
            { std::unique_lock<std::mutex> lk(mu); done = true; cv.notify_all(); } for (auto& t : th) t.join();
            if (m) munmap(m, n); close(fd);
            const double t1 = now(); printf("%s x%d writers: %zu MB in %.3f s = %.1f GB/s\n", mode == 0 ? "pinned pieces + pwrite" : "pinned pieces + memcpy into mmap", T, n >> 20, t1 - t0, n / (t1 - t0) / 1e9);
            for (auto& p : hb) hipHostFree(p);
        }
    }
    // (c) the mapping itself page-locked piece by piece (R registering threads, pieces in order), the device copies into it
    for (int R : {2, 8}) {
        unlink(argv[1]); int fd = open(argv[1], O_RDWR | O_CREAT | O_TRUNC, 0644); const size_t big = 256u << 20, nb = (n + big - 1) / big;
        const double t0 = now(); if (ftruncate(fd, n)) return 3; uint8_t* m = (uint8_t*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        std::vector<std::atomic<int>> st(nb); for (auto& v : st) v = 0; std::atomic<size_t> next{0}; std::vector<std::thread> th;
        for (int t = 0; t < R; t++) th.emplace_back([&] { for (;;) { const size_t i = next.fetch_add(1); if (i >= nb) break; const size_t off = i * big, len = std::min(big, n - off);
                st[i] = hipHostRegister(m + off, len, hipHostRegisterDefault) == hipSuccess ? 1 : 2; } });
        int good = 0;
        for (size_t i = 0; i < nb; i++) { while (st[i].load() == 0) std::this_thread::yield(); const size_t off = i * big, len = std::min(big, n - off); good += st[i] == 1;
                hipMemcpyAsync(m + off, (uint8_t*)d + off, len, hipMemcpyDeviceToHost, s); }
        hipStreamSynchronize(s); for (auto& t : th) t.join();
        const double t1 = now(); for (size_t i = 0; i < nb; i++) if (st[i] == 1) hipHostUnregister(m + i * big); munmap(m, n); close(fd);
        printf("mapping registered by %d threads, device writes into the page cache: %.3f s = %.1f GB/s (registered %d / %zu); with unregister + munmap %.3f s\n", R, t1 - t0, n / (t1 - t0) / 1e9, good, nb, now() - t0);
    }
    unlink(argv[1]);
    return 0;
}
