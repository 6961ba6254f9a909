// pool_driver.cpp — the drop-in scenario as the reference's processor produces it (parasol_runtime/src/circuit_processor/
// mod.rs:192-253): T host threads, each calling a single-ciphertext operation synchronously, again and again.  Python threads
// cannot generate that load (the interpreter lock serialises the callers), so bench.py's `evaluation_pool` leg loads this
// little driver beside the library: it only uses the public C ABI (include/spf_hip.h).
// Build: g++ -O2 -std=c++17 -shared -fPIC -pthread -I include -o tools/bin/libpool_driver.so tools/pool_driver.cpp
#include "spf_hip.h"

#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

extern "C" {

typedef spf_status (*submit_fn)(spf_pool*, const uint64_t*, double*, uint64_t*);
typedef spf_status (*wait_fn)(spf_pool*, uint64_t);

// `threads` callers loop submit(keyswitch + circuit bootstrap of ONE L1 LWE) + wait for `seconds`; inputs are copies of
// `lwe1` (lwe1_words words), every thread owns its 256 KiB output.  Returns operations completed, -1 on an error status;
// *elapsed_s = wall time from the first submit to the last wait.  all_out (may be null): threads x ggsw_doubles, every caller's
// LAST output, for the test that checks each caller got its own bytes under load (caller t's input is lwe1 with t added to
// word 0).
static long drive(spf_pool* pool, submit_fn submit, wait_fn wait, int threads, double seconds, const uint64_t* lwe1,
                  size_t lwe1_words, size_t ggsw_doubles, double* elapsed_s, double* first_out_checksum, double* all_out)
{
    std::atomic<long> done{0};
    std::atomic<int> failed{0};
    std::vector<std::thread> th;
    std::vector<std::vector<uint64_t>> in((size_t)threads);
    std::vector<std::vector<double>> out((size_t)threads);
    for (int t = 0; t < threads; t++) {
        in[(size_t)t].assign(lwe1, lwe1 + lwe1_words);
        in[(size_t)t][0] += (uint64_t)t; // (a different ciphertext per caller)
        out[(size_t)t].assign(ggsw_doubles, 0.0);
    }
    const auto t0 = std::chrono::steady_clock::now();
    const auto until = t0 + std::chrono::duration<double>(seconds);
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            while (std::chrono::steady_clock::now() < until && !failed.load()) {
                uint64_t ticket = 0;
                if (submit(pool, in[(size_t)t].data(), out[(size_t)t].data(), &ticket) != SPF_OK || wait(pool, ticket) != SPF_OK) {
                    failed.store(1);
                    return;
                }
                done.fetch_add(1);
            }
        });
    for (auto& x : th) x.join();
    *elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double s = 0;
    for (size_t i = 0; i < ggsw_doubles; i += 997) s += out[0][i] * 1e-60;
    *first_out_checksum = s;
    if (all_out)
        for (int t = 0; t < threads; t++) std::memcpy(all_out + (size_t)t * ggsw_doubles, out[(size_t)t].data(), ggsw_doubles * sizeof(double));
    return failed.load() ? -1 : done.load();
}

long spf_pool_drive(spf_pool* pool, submit_fn submit, wait_fn wait, int threads, double seconds, const uint64_t* lwe1,
                    size_t lwe1_words, size_t ggsw_doubles, double* elapsed_s, double* first_out_checksum)
{
    return drive(pool, submit, wait, threads, seconds, lwe1, lwe1_words, ggsw_doubles, elapsed_s, first_out_checksum, nullptr);
}

long spf_pool_drive_collect(spf_pool* pool, submit_fn submit, wait_fn wait, int threads, double seconds, const uint64_t* lwe1,
                            size_t lwe1_words, size_t ggsw_doubles, double* elapsed_s, double* all_out)
{
    double ck;
    return drive(pool, submit, wait, threads, seconds, lwe1, lwe1_words, ggsw_doubles, elapsed_s, &ck, all_out);
}

}
