// measurement only (GPU box host): ways to fill ONE tmpfs file of N GB from T threads — pwrite to disjoint ranges, a shared mapping (plain / MADV_POPULATE_WRITE / fallocate first) —
// behind the JSON writer (svjg_json.cpp: one writer thread).   g++ -O2 -pthread -o tools/ubench/writetest tools/ubench/writetest.cpp ; tools/ubench/writetest /dev/shm/x.bin <GB> <threads> pwrite|mmap|mmap_populate|falloc_mmap
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const char *path = argv[1]; const double gb = atof(argv[2]); const int T = atoi(argv[3]); const char *mode = argv[4];
    const size_t N = (size_t)(gb * (1ull << 30)) / (64 << 20) * (64 << 20), CH = 4 << 20;
    // the source is 2 GB walked once per 2 GB written: cold, like freshly rendered text (argv[5] = "hot": one 4 MB chunk, cache resident)
    const bool hot = argc > 5 && !strcmp(argv[5], "hot");
    const size_t SRC = hot ? CH : (size_t)2 << 30;
    std::vector<char> srcv(SRC); for (size_t i = 0; i < SRC; i += 64) srcv[i] = (char)(i * 31 + 7);
    auto srcp = [&](size_t off) { return srcv.data() + (off % SRC); };
    unlink(path);
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    double t0 = now();
    std::atomic<size_t> next{0};
    auto run = [&](auto body) { std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back([&, t] { for (;;) { size_t c = next.fetch_add(1); if (c * CH >= N) break; body(c * CH); } }); for (auto &x : th) x.join(); };
    if (!strcmp(mode, "falloc_pwrite")) {
        // a helper allocates ahead in 256 MB steps (fallocate holds the inode lock for the length of its call), the T writers only copy
        if (ftruncate(fd, N)) { perror("ftruncate"); return 1; }
        std::thread helper([&] { for (size_t a = 0; a < N; a += 256u << 20) if (fallocate(fd, 0, a, N - a < (256u << 20) ? N - a : (256u << 20))) { perror("fallocate"); break; } });
        run([&](size_t off) { size_t d = 0; while (d < CH) { ssize_t w = pwrite(fd, srcp(off) + d, CH - d, off + d); if (w <= 0) { perror("pwrite"); exit(1); } d += w; } });
        helper.join();
    } else if (!strcmp(mode, "pwrite")) {
        run([&](size_t off) { size_t d = 0; while (d < CH) { ssize_t w = pwrite(fd, srcp(off) + d, CH - d, off + d); if (w <= 0) { perror("pwrite"); exit(1); } d += w; } });
    } else {
        if (ftruncate(fd, N)) { perror("ftruncate"); return 1; }
        if (!strcmp(mode, "falloc_mmap")) { double a = now(); if (fallocate(fd, 0, 0, N)) perror("fallocate"); printf("  fallocate %.2f s\n", now() - a); }
        std::thread helper;
        std::atomic<size_t> ready{0};                          // falloc_ahead_mmap: a helper allocates ahead in 256 MB steps, the copiers wait for their range
        const bool ahead = !strcmp(mode, "falloc_ahead_mmap");
        if (ahead) helper = std::thread([&] { for (size_t a = 0; a < N; a += 256u << 20) { if (fallocate(fd, 0, a, N - a < (256u << 20) ? N - a : (256u << 20))) { perror("fallocate"); } ready.store(a + (256u << 20)); } });
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) { perror("mmap"); return 1; }
        const bool pop = !strcmp(mode, "mmap_populate");
        run([&](size_t off) { if (ahead) while (ready.load() < off + CH) std::this_thread::yield(); if (pop && madvise(m + off, CH, MADV_POPULATE_WRITE)) { perror("madvise"); exit(1); } memcpy(m + off, srcp(off), CH); });
        if (ahead) helper.join();
        munmap(m, N);
    }
    close(fd);
    double dt = now() - t0;
    printf("%s T=%d: %.1f GB in %.2f s = %.2f GB/s\n", mode, T, N / 1e9, dt, N / 1e9 / dt);
    unlink(path);
    return 0;
}
