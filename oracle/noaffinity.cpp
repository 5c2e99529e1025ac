// TEST / BENCH INFRASTRUCTURE (oracle/): LD_PRELOAD shim for timing the compiled reference as a CPU baseline.
// The reference pins pool worker i to CPU i (lib/threadpool/ThreadPool.cpp:26-39).  Inside a container whose CPU quota is smaller
// than its affinity mask (a 16-core quota over 256 CPUs on the GPU boxes) that stacks the workers on the first CPUs' hyperthreads
// and understates the reference; with this shim the call succeeds and does nothing, and the kernel places the workers.
// bench.py's cpu_baseline leg reports the reference both ways.
#define _GNU_SOURCE 1
#include <pthread.h>
#include <sched.h>
extern "C" int pthread_setaffinity_np(pthread_t, size_t, const cpu_set_t*) { return 0; }
