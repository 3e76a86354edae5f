// Measurement only: parallel pwrite() vs stores into a shared mapping when many threads fill ONE output file (the JSON writer).
//   g++ -O2 -pthread -o /tmp/writetest tools/ubench/writetest.cpp && /tmp/writetest /dev/shm/x.bin <0=pwrite|1=mmap> <threads> <MB>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <thread>
#include <vector>
#include <chrono>
int main(int argc, char **argv) {
    const char *path = argv[1]; int mode = atoi(argv[2]); int T = atoi(argv[3]); size_t total = (size_t)atol(argv[4]) << 20;
    const size_t piece = 4 << 20;
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    char *map = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    if (mode == 1) { if (ftruncate(fd, total)) return 1; map = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); if (map == MAP_FAILED) return 2; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
        std::vector<char> buf(piece, 'a' + t);
        for (size_t off = (size_t)t * piece; off < total; off += (size_t)T * piece) {
            size_t n = total - off < piece ? total - off : piece;
            if (mode == 0) { if (pwrite(fd, buf.data(), n, off) != (ssize_t)n) abort(); }
            else memcpy(map + off, buf.data(), n);
        }
    });
    for (auto &x : th) x.join();
    if (map) munmap(map, total);
    close(fd);
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%s mode=%s threads=%d: %.2f s, %.2f GB/s\n", path, mode ? "mmap" : "pwrite", T, s, total / s / 1e9);
    unlink(path);
}
