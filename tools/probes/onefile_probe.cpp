// ONE output file written by T threads: pwrite() (serialised by the inode lock) against stores through a shared mapping (page faults
// take pages one by one, no inode lock).   g++ -O2 -pthread tools/probes/onefile_probe.cpp -o /tmp/onefile_probe && /tmp/onefile_probe [threads] [GiB] [chunk MiB]
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 12; const double G = argc > 2 ? atof(argv[2]) : 24; const size_t CH = (size_t)(argc > 3 ? atoi(argv[3]) : 8) << 20;
    const size_t n = (size_t)(G * (1ull << 30)) / CH, total = n * CH;
    const char* path = "/dev/shm/onefile_probe.bin";
    std::vector<char*> src(T);
    for (int t = 0; t < T; ++t) { src[t] = (char*)malloc(CH); for (size_t i = 0; i < CH; ++i) src[t][i] = (char)(i * 131 + t); }
    auto run = [&](const char* tag, auto body) {
        std::vector<std::thread> th; const double t0 = now();
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] { for (size_t c = t; c < n; c += T) body(t, c); });
        for (auto& x : th) x.join();
        printf("%-44s %6.1f GB/s (%.2f s)\n", tag, total / (now() - t0) / 1e9, now() - t0); fflush(stdout);
    };
    unlink(path);
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    run("pwrite, new file", [&](int t, size_t c) { size_t o = 0; while (o < CH) { ssize_t w = pwrite(fd, src[t] + o, CH - o, (off_t)(c * CH + o)); if (w <= 0) exit(1); o += (size_t)w; } });
    run("pwrite, pages there", [&](int t, size_t c) { size_t o = 0; while (o < CH) { ssize_t w = pwrite(fd, src[t] + o, CH - o, (off_t)(c * CH + o)); if (w <= 0) exit(1); o += (size_t)w; } });
    close(fd); unlink(path);
    fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (ftruncate(fd, (off_t)total) != 0) return 1;
    char* m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) return 1;
    run("mapping, new file (a fault per page)", [&](int t, size_t c) { memcpy(m + c * CH, src[t], CH); });
    run("mapping, pages there", [&](int t, size_t c) { memcpy(m + c * CH, src[t], CH); });
    munmap(m, total); close(fd); unlink(path);
    fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (ftruncate(fd, (off_t)total) != 0) return 1;
    m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    run("mapping, new file, MADV_POPULATE_WRITE first", [&](int t, size_t c) { madvise(m + c * CH, CH, 23 /* MADV_POPULATE_WRITE */); memcpy(m + c * CH, src[t], CH); });
    munmap(m, total); close(fd); unlink(path);
    // K files, one per thread (what the part files do)
    std::vector<int> fds(T);
    for (int t = 0; t < T; ++t) { char p[256]; snprintf(p, sizeof p, "%s.%d", path, t); fds[t] = open(p, O_RDWR | O_CREAT | O_TRUNC, 0644); }
    run("pwrite, a new file per thread", [&](int t, size_t c) { size_t o = 0; while (o < CH) { ssize_t w = pwrite(fds[t], src[t] + o, CH - o, (off_t)(c / T * CH + o)); if (w <= 0) exit(1); o += (size_t)w; } });
    for (int t = 0; t < T; ++t) { char p[256]; snprintf(p, sizeof p, "%s.%d", path, t); close(fds[t]); unlink(p); }
    return 0;
}
