// Fresh files while the files of the job before are deleted -- who frees the old pages?  T writer threads each write G GiB into a new
// tmpfs file; the previous round's files go away (a) by U unlinking threads beside them (what bench.py's Cleaner does), (b) by the
// writers themselves, a stretch of the old file punched out (fallocate PUNCH_HOLE) before every stretch of the new one is written:
// the pages a CPU frees are the pages it takes next (per-CPU page lists), (c) not at all (the files are kept: the floor).
//   g++ -O2 -pthread tools/probes/recycle_probe.cpp -o /tmp/recycle_probe && /tmp/recycle_probe [writers=12] [GiB each=6] [stretch MiB=1] [unlinkers=4]
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 12; const double G = argc > 2 ? atof(argv[2]) : 6; const size_t ST = (size_t)(argc > 3 ? atoi(argv[3]) : 1) << 20; const int U = argc > 4 ? atoi(argv[4]) : 4;
    const size_t n = (size_t)(G * (1ull << 30)) / ST, per = n * ST; const std::string dir = "/dev/shm/recycle_probe";
    mkdir(dir.c_str(), 0755);
    const size_t SRC = 64u << 20;                                                    // a source larger than the caches, as the pinned slots are
    std::vector<char*> src(T);
    for (int t = 0; t < T; ++t) { src[t] = (char*)malloc(SRC); for (size_t i = 0; i < SRC; i += 64) src[t][i] = (char)(i + t); }
    auto name = [&](int round, int t) { return dir + "/r" + std::to_string(round) + "_" + std::to_string(t); };
    auto write_round = [&](const char* tag, int round, int mode) {                  // mode 0: nothing deleted, 1: U unlinkers beside, 2: the writers punch the old file
        std::atomic<int> next_victim{0}; std::vector<std::thread> th, un; const double t0 = now();
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
            const int fd = open(name(round, t).c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
            const int old = mode == 2 ? open(name(round - 1, t).c_str(), O_RDWR) : -1;
            for (size_t c = 0; c < n; ++c) {
                if (old >= 0) fallocate(old, FALLOC_FL_PUNCH_HOLE | FALLOC_FL_KEEP_SIZE, (off_t)(c * ST), (off_t)ST);
                const char* s = src[t] + (c * ST) % SRC; size_t o = 0;
                while (o < ST) { ssize_t w = pwrite(fd, s + o, ST - o, (off_t)(c * ST + o)); if (w <= 0) exit(1); o += (size_t)w; }
            }
            close(fd); if (old >= 0) { close(old); unlink(name(round - 1, t).c_str()); }
        });
        if (mode == 1) for (int u = 0; u < U; ++u) un.emplace_back([&] { for (;;) { const int v = next_victim.fetch_add(1); if (v >= T) return; unlink(name(round - 1, v).c_str()); } });
        for (auto& x : th) x.join();
        const double t1 = now();
        for (auto& x : un) x.join();
        printf("%-58s %6.1f GB/s (%.2f s; the deleting done %.2f s later)\n", tag, T * per / (t1 - t0) / 1e9, t1 - t0, now() - t1); fflush(stdout);
    };
    write_round("new files, nothing deleted", 0, 0);
    write_round("new files, unlinking threads beside the writers", 1, 1);
    write_round("new files, every writer punches the old file as it goes", 2, 2);
    write_round("new files, unlinking threads beside the writers (again)", 3, 1);
    write_round("new files, every writer punches the old file (again)", 4, 2);
    for (int t = 0; t < T; ++t) unlink(name(4, t).c_str());
    rmdir(dir.c_str());
    return 0;
}
