// TEST INFRASTRUCTURE (oracle tooling) -- never linked into the product.
//
// LD_PRELOAD shim that pins the wall clock seen by the compiled reference
// binary (oracle/_ref/scssim_ref) so that its RNG seeding becomes
// reproducible.  The reference seeds everything from the clock:
//   srand(time(NULL))                                  src/scssim.cpp:26,47
//   mt19937(chrono::system_clock::now()...count())     lib/threadpool/ThreadPool.cpp:41-47
//   default_random_engine(system_clock::now()...)      lib/profile/Profile.cpp:1406-1408
// With SCS_FIXED_TIME=T:  time() == T  and  system_clock::now() == T seconds.
#include <chrono>
#include <cstdlib>
#include <ctime>

static long long pinned_seconds() {
    const char* s = getenv("SCS_FIXED_TIME");
    return s ? atoll(s) : 1234567890LL;
}

extern "C" time_t time(time_t* out) {
    time_t v = (time_t)pinned_seconds();
    if (out) *out = v;
    return v;
}

namespace std { namespace chrono { inline namespace _V2 {
system_clock::time_point system_clock::now() noexcept {
    return time_point(duration(pinned_seconds() * 1000000000LL));
}
}}}
