// Probe: can a batch's text cross PCIe STRAIGHT INTO the page cache of a tmpfs file (the file mapped shared, the mapping registered
// with the HIP runtime, the D2H copy's destination) more cheaply than through a pinned slot + pwrite()?  Prints GB/s of each step,
// for T threads each working on its own file.   hipcc -O2 tools/probes/reg_probe.hip -o /tmp/reg_probe -lpthread ; /tmp/reg_probe [dir] [MB] [threads]
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
#include <string>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Res { double t_alloc = 0, t_reg = 0, t_copy = 0, t_unreg = 0, t_pwrite = 0, t_d2h_pinned = 0; };
int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm";
    const size_t mb = argc > 2 ? atol(argv[2]) : 1024; const int T = argc > 3 ? atoi(argv[3]) : 1;
    const size_t n = mb << 20;
    OK(hipSetDevice(0));
    std::vector<char*> dsrc(T); std::vector<char*> pin(T); std::vector<hipStream_t> st(T);
    for (int t = 0; t < T; ++t) { OK(hipMalloc(&dsrc[t], n)); OK(hipMemset(dsrc[t], 65 + t, n)); OK(hipHostMalloc(&pin[t], n, hipHostMallocDefault)); OK(hipStreamCreate(&st[t])); }
    OK(hipDeviceSynchronize());
    for (int mode = 0; mode < 3; ++mode) {        // 0: pinned slot + pwrite, 1: fallocate + mmap + register + D2H + unregister, 2: the same with MAP_POPULATE instead of fallocate
        std::vector<Res> res(T); std::vector<std::thread> th;
        const double t0 = now();
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
            OK(hipSetDevice(0));
            const std::string path = dir + "/reg_probe_" + std::to_string(mode) + "_" + std::to_string(t);
            int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644); if (fd < 0) { perror("open"); exit(1); }
            Res& r = res[t]; double a = now();
            if (mode == 0) {
                OK(hipMemcpyAsync(pin[t], dsrc[t], n, hipMemcpyDeviceToHost, st[t])); OK(hipStreamSynchronize(st[t])); r.t_d2h_pinned = now() - a; a = now();
                size_t off = 0; while (off < n) { ssize_t w = pwrite(fd, pin[t] + off, n - off, off); if (w <= 0) { perror("pwrite"); exit(1); } off += (size_t)w; }
                r.t_pwrite = now() - a;
            } else {
                if (ftruncate(fd, (off_t)n)) { perror("ftruncate"); exit(1); }
                if (mode == 1 && fallocate(fd, 0, 0, (off_t)n)) { perror("fallocate"); exit(1); }
                void* m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED | (mode == 2 ? MAP_POPULATE : 0), fd, 0); if (m == MAP_FAILED) { perror("mmap"); exit(1); }
                r.t_alloc = now() - a; a = now();
                hipError_t e = hipHostRegister(m, n, hipHostRegisterDefault);
                if (e != hipSuccess) { fprintf(stderr, "hipHostRegister on a shared file mapping: %s\n", hipGetErrorString(e)); exit(2); }
                r.t_reg = now() - a; a = now();
                OK(hipMemcpyAsync(m, dsrc[t], n, hipMemcpyDeviceToHost, st[t])); OK(hipStreamSynchronize(st[t])); r.t_copy = now() - a; a = now();
                OK(hipHostUnregister(m)); munmap(m, n); r.t_unreg = now() - a;
            }
            char chk[4] = {0}; if (pread(fd, chk, 4, (off_t)(n - 4)) != 4 || chk[3] != (char)(65 + t)) { fprintf(stderr, "content check failed (mode %d)\n", mode); exit(3); }
            close(fd); unlink(path.c_str());
        });
        for (auto& x : th) x.join();
        const double wall = now() - t0, gb = (double)n * T / 1e9;
        Res s; for (auto& r : res) { s.t_alloc += r.t_alloc / T; s.t_reg += r.t_reg / T; s.t_copy += r.t_copy / T; s.t_unreg += r.t_unreg / T; s.t_pwrite += r.t_pwrite / T; s.t_d2h_pinned += r.t_d2h_pinned / T; }
        const double g1 = (double)n / 1e9;
        if (mode == 0) printf("%d thread(s) x %zu MB, pinned slot + pwrite : D2H %.1f GB/s per thread, pwrite %.1f GB/s per thread ; all files in %.2f s = %.1f GB/s (incl. unlink)\n", T, mb, g1 / s.t_d2h_pinned, g1 / s.t_pwrite, wall, gb / wall);
        else printf("%d thread(s) x %zu MB, %s + mmap + register : pages %.1f, register %.1f, D2H into the mapping %.1f, unregister+munmap %.1f GB/s per thread ; all files in %.2f s = %.1f GB/s (incl. unlink)\n",
                    T, mb, mode == 1 ? "fallocate" : "MAP_POPULATE", g1 / s.t_alloc, g1 / s.t_reg, g1 / s.t_copy, g1 / s.t_unreg, wall, gb / wall);
        fflush(stdout);
    }
    return 0;
}
