// pool_driver.cpp — the drop-in scenario as the reference's processor produces it (parasol_runtime/src/circuit_processor/
// mod.rs:192-253): T host threads, each calling a single-ciphertext operation synchronously, again and again.  Python threads
// cannot generate that load (the interpreter lock serialises the callers), so bench.py's `evaluation_pool` leg loads this
// little driver beside the library: it only uses the public C ABI (include/spf_hip.h).
// Build: g++ -O2 -std=c++17 -shared -fPIC -pthread -I include -o tools/bin/libpool_driver.so tools/pool_driver.cpp
#include "spf_hip.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
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

// ---- r06: the same scenario BY HANDLE (device-resident values, include/spf_hip.h "device-resident values") --------------------

#include <climits>
#include <deque>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

extern "C" {

typedef spf_status (*submit_v1_fn)(spf_pool*, const spf_value*, spf_value**, uint64_t*);
typedef spf_status (*submit_cmux_fn)(spf_pool*, const double*, const uint64_t*, const uint64_t*, uint64_t*, uint64_t*);
typedef spf_status (*submit_cmux_v_fn)(spf_pool*, const spf_value*, const spf_value*, const spf_value*, spf_value**, uint64_t*);
typedef spf_status (*submit_op_v_fn)(spf_pool*, spf_graph_op, const spf_value* const*, size_t, uint64_t, spf_value**, uint64_t*);
typedef void (*release_fn)(spf_value*);

// `threads` callers loop submit_v(one input value each: inputs[t]) + wait + release of the result for `seconds`.
// keep_last (may be null): threads slots; every caller's LAST result is kept there (the test downloads and checks them, then
// releases them).  Returns operations completed, -1 on an error status.
long spf_pool_drive_v(spf_pool* pool, submit_v1_fn submit, wait_fn wait, release_fn release, int threads, double seconds,
                      spf_value* const* inputs, double* elapsed_s, spf_value** keep_last)
{
    std::atomic<long> done{0};
    std::atomic<int> failed{0};
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    const auto until = t0 + std::chrono::duration<double>(seconds);
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            spf_value* last = nullptr;
            while (std::chrono::steady_clock::now() < until && !failed.load()) {
                uint64_t ticket = 0;
                spf_value* out = nullptr;
                if (submit(pool, inputs[t], &out, &ticket) != SPF_OK) { failed.store(1); break; }
                if (wait(pool, ticket) != SPF_OK) { release(out); failed.store(1); break; }
                if (last) release(last);
                last = out;
                done.fetch_add(1);
            }
            if (keep_last) keep_last[t] = last;
            else if (last) release(last);
        });
    for (auto& x : th) x.join();
    *elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return failed.load() ? -1 : done.load();
}

// CMUX gates one per call from `threads` callers, each keeping up to `window` tickets open (window 1: the synchronous caller —
// submit, wait, submit; larger: a task that has several independent gates submits them all before it waits for the first, as
// the pool's asynchronous use allows): host-pointer form (every operand and result crosses PCIe) ...
long spf_pool_drive_cmux(spf_pool* pool, submit_cmux_fn submit, wait_fn wait, int threads, int window, double seconds, const double* sel,
                         size_t ggsw_doubles, const uint64_t* a, const uint64_t* b, size_t glwe_words, double* elapsed_s)
{
    std::atomic<long> done{0};
    std::atomic<int> failed{0};
    std::vector<std::thread> th;
    if (window < 1) window = 1;
    std::vector<std::vector<uint64_t>> out((size_t)threads);
    for (auto& o : out) o.assign(glwe_words * (size_t)window, 0);
    // (every caller owns its operands, as every task of the reference owns its ciphertexts)
    std::vector<std::vector<double>> sels((size_t)threads, std::vector<double>(sel, sel + ggsw_doubles));
    std::vector<std::vector<uint64_t>> as((size_t)threads, std::vector<uint64_t>(a, a + glwe_words)), bs((size_t)threads, std::vector<uint64_t>(b, b + glwe_words));
    const auto t0 = std::chrono::steady_clock::now();
    const auto until = t0 + std::chrono::duration<double>(seconds);
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            std::vector<uint64_t> tickets((size_t)window, 0);
            long n = 0; // submitted so far; ticket i of the ring is slot i % window
            long collected = 0;
            bool stop = false;
            while (!stop || collected < n) {
                stop = stop || std::chrono::steady_clock::now() >= until || failed.load();
                if (!stop && n - collected < window) {
                    const size_t k = (size_t)(n % window);
                    if (submit(pool, sels[(size_t)t].data(), as[(size_t)t].data(), bs[(size_t)t].data(), out[(size_t)t].data() + k * glwe_words, &tickets[k]) != SPF_OK) {
                        failed.store(1);
                        stop = true;
                        continue;
                    }
                    n++;
                    continue;
                }
                if (collected < n) {
                    if (wait(pool, tickets[(size_t)(collected % window)]) != SPF_OK) failed.store(1);
                    collected++;
                    done.fetch_add(1);
                }
            }
        });
    for (auto& x : th) x.join();
    *elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return failed.load() ? -1 : done.load();
}

// ... and by handle: sel / a / b are `threads` values each (caller t uses sel[t], a[t], b[t]); a result is released when collected
long spf_pool_drive_cmux_v(spf_pool* pool, submit_cmux_v_fn submit, wait_fn wait, release_fn release, int threads, int window, double seconds,
                           spf_value* const* sel, spf_value* const* a, spf_value* const* b, double* elapsed_s)
{
    std::atomic<long> done{0};
    std::atomic<int> failed{0};
    std::vector<std::thread> th;
    if (window < 1) window = 1;
    const auto t0 = std::chrono::steady_clock::now();
    const auto until = t0 + std::chrono::duration<double>(seconds);
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            std::vector<uint64_t> tickets((size_t)window, 0);
            std::vector<spf_value*> outs((size_t)window, nullptr);
            long n = 0, collected = 0;
            bool stop = false;
            while (!stop || collected < n) {
                stop = stop || std::chrono::steady_clock::now() >= until || failed.load();
                if (!stop && n - collected < window) {
                    const size_t k = (size_t)(n % window);
                    if (submit(pool, sel[t], a[t], b[t], &outs[k], &tickets[k]) != SPF_OK) {
                        failed.store(1);
                        stop = true;
                        continue;
                    }
                    n++;
                    continue;
                }
                if (collected < n) {
                    const size_t k = (size_t)(collected % window);
                    if (wait(pool, tickets[k]) != SPF_OK) failed.store(1);
                    release(outs[k]);
                    outs[k] = nullptr;
                    collected++;
                    done.fetch_add(1);
                }
            }
        });
    for (auto& x : th) x.join();
    *elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return failed.load() ? -1 : done.load();
}

// CMux gates PUSHED: each thread submits bursts of `burst` gates without tickets (nobody waits for a single gate), two bursts in
// flight — while one runs the next is pushed — and waits for the VALUES of a burst before it reuses its slots.
typedef spf_status (*value_wait_fn)(const spf_value*);
long spf_pool_push_cmux_v(spf_pool* pool, submit_cmux_v_fn submit, value_wait_fn value_wait, release_fn release, int threads, int burst,
                          double seconds, spf_value* const* sel, spf_value* const* a, spf_value* const* b, double* elapsed_s)
{
    std::atomic<long> done{0};
    std::atomic<int> failed{0};
    std::vector<std::thread> th;
    if (burst < 1) burst = 1;
    const auto t0 = std::chrono::steady_clock::now();
    const auto until = t0 + std::chrono::duration<double>(seconds);
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            std::vector<spf_value*> outs[2];
            auto collect = [&](std::vector<spf_value*>& v) {
                for (spf_value* x : v) {
                    if (value_wait(x) != SPF_OK) failed.store(1);
                    release(x);
                }
                done.fetch_add((long)v.size());
                v.clear();
            };
            for (int half = 0; std::chrono::steady_clock::now() < until && !failed.load(); half ^= 1) {
                collect(outs[half]);
                for (int i = 0; i < burst; i++) {
                    spf_value* o = nullptr;
                    if (submit(pool, sel[t], a[t], b[t], &o, nullptr) != SPF_OK) { failed.store(1); break; }
                    outs[half].push_back(o);
                }
            }
            collect(outs[0]);
            collect(outs[1]);
        });
    for (auto& x : th) x.join();
    *elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return failed.load() ? -1 : done.load();
}

// A gate graph executed the way the reference executes it (circuit_processor/mod.rs:130-253): one task per node, a task runs as
// soon as its operands exist, every task calls the evaluator with ONE operation and blocks until it is done — here
// spf_pool_submit_op_v + spf_pool_wait from a pool of `threads` workers; a node's value is released when its last consumer has
// finished (the reference drops the task's `Arc`).
//   op[i]      spf_graph_op of node i, or -1: the node is an input / constant whose value is already in values[i]
//   in[3*i..]  operand nodes (n_in[i] of them, earlier nodes), param[i] SampleExtract index / MulXN amount
//   keep[i]    nonzero: the value stays in values[i] for the caller (outputs); otherwise values[i] is released and set to null
// Returns 0, or the first failing status (first-error-wins, :214-223): later tasks become no-ops, as in the reference.
int spf_circuit_drive(spf_pool* pool, submit_op_v_fn submit, wait_fn wait, release_fn release, int threads, uint32_t n_nodes,
                      const int32_t* op, const uint32_t* in, const uint32_t* n_in, const uint64_t* param, spf_value** values,
                      const uint8_t* keep, double* elapsed_s)
{
    std::vector<std::atomic<uint32_t>> deps(n_nodes), users(n_nodes);
    std::vector<std::vector<uint32_t>> dependents(n_nodes);
    uint32_t n_tasks = 0;
    for (uint32_t i = 0; i < n_nodes; i++) { deps[i].store(0); users[i].store(0); }
    for (uint32_t i = 0; i < n_nodes; i++) {
        if (op[i] < 0) continue;
        n_tasks++;
        for (uint32_t k = 0; k < n_in[i]; k++) {
            const uint32_t src = in[3 * i + k];
            users[src].fetch_add(1);
            if (op[src] >= 0) { deps[i].fetch_add(1); dependents[src].push_back(i); }
        }
    }
    // (the ready queue is touched for a few hundred nanoseconds at a time by dozens of workers at once: a spin lock; idle workers
    // sleep on a futex word that counts queue pushes.  A worker that makes operations ready runs one of them itself — the
    // critical path of a circuit never waits for another thread to wake up, as with rayon's local deques.)
    std::atomic_flag qlock = ATOMIC_FLAG_INIT;
    auto lock = [&] { while (qlock.test_and_set(std::memory_order_acquire)) __builtin_ia32_pause(); };
    auto unlock = [&] { qlock.clear(std::memory_order_release); };
    std::deque<uint32_t> ready;
    std::atomic<uint32_t> finished{0};
    std::atomic<uint32_t> bell{0}; // futex word: bumped whenever the queue gets something or everything has finished
    std::atomic<int> sleepers{0};
    std::atomic<int> error{0};
    for (uint32_t i = 0; i < n_nodes; i++)
        if (op[i] >= 0 && deps[i].load() == 0) ready.push_back(i);
    auto ring = [&](int n) {
        bell.fetch_add(1, std::memory_order_release);
        if (sleepers.load(std::memory_order_acquire) > 0) (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(&bell), FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0);
    };
    // (the reference's workers exist before the circuit arrives — a rayon pool: the clock starts when all of them are up)
    std::atomic<int> up{0};
    std::atomic<uint32_t> gate{0};
    std::chrono::steady_clock::time_point t0;
    auto worker = [&] {
        up.fetch_add(1);
        while (gate.load(std::memory_order_acquire) == 0) (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(&gate), FUTEX_WAIT_PRIVATE, 0, nullptr, nullptr, 0);
        bool have = false;
        uint32_t node = 0;
        for (;;) {
            while (!have) {
                const uint32_t seen = bell.load(std::memory_order_acquire);
                lock();
                if (!ready.empty()) { node = ready.front(); ready.pop_front(); have = true; }
                unlock();
                if (have) break;
                if (finished.load(std::memory_order_acquire) == n_tasks) return;
                sleepers.fetch_add(1);
                (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(&bell), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
                sleepers.fetch_sub(1);
            }
            have = false;
            if (!error.load()) { // (after an error the remaining tasks only retire)
                const spf_value* args[3] = {nullptr, nullptr, nullptr};
                for (uint32_t k = 0; k < n_in[node]; k++) args[k] = values[in[3 * node + k]];
                uint64_t ticket = 0;
                spf_value* out = nullptr;
                spf_status st = submit(pool, (spf_graph_op)op[node], args, n_in[node], param[node], &out, &ticket);
                if (st == SPF_OK) {
                    st = wait(pool, ticket);
                    if (st != SPF_OK) { release(out); out = nullptr; }
                }
                if (st != SPF_OK) { int zero = 0; error.compare_exchange_strong(zero, (int)st); }
                values[node] = out;
            }
            for (uint32_t k = 0; k < n_in[node]; k++) { // the operands' last user lets them go
                const uint32_t src = in[3 * node + k];
                if (users[src].fetch_sub(1) == 1 && !keep[src] && values[src]) { release(values[src]); values[src] = nullptr; }
            }
            if (users[node].load() == 0 && !keep[node] && values[node]) { release(values[node]); values[node] = nullptr; }
            uint32_t mine = 0;
            int pushed = 0;
            for (uint32_t d : dependents[node])
                if (deps[d].fetch_sub(1) == 1) {
                    if (!have) { have = true; mine = d; continue; } // this thread goes on with the first one itself
                    lock();
                    ready.push_back(d);
                    unlock();
                    pushed++;
                }
            const uint32_t done = finished.fetch_add(1, std::memory_order_acq_rel) + 1;
            if (done == n_tasks) ring(INT32_MAX);
            else if (pushed) ring(pushed);
            node = mine;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(worker);
    while (up.load() < threads) std::this_thread::yield();
    t0 = std::chrono::steady_clock::now();
    gate.store(1, std::memory_order_release);
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(&gate), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
    for (auto& x : th) x.join();
    *elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return error.load();
}

// The same circuit PUSHED: one thread submits every operation without a ticket and without waiting for anything, in the order
// `order` gives (level by level: the order in which the reference's processor would see the tasks become ready if every task
// returned at once) — the operands of a task are results that are still pending, the pool orders and batches them by level
// (spf_hip.h, "Deferred operands") — then waits for the `n_out` output values only.  Values nobody keeps are released right after
// the last submit that takes them (the pool holds its own references while they are needed).
typedef spf_status (*flush_fn)(spf_pool*);
int spf_circuit_push(spf_pool* pool, submit_op_v_fn submit, value_wait_fn value_wait, release_fn release, flush_fn flush, uint32_t n_nodes,
                     const int32_t* op, const uint32_t* in, const uint32_t* n_in, const uint64_t* param, spf_value** values,
                     const uint8_t* keep, const uint32_t* order, uint32_t n_order, const uint32_t* outputs, uint32_t n_out,
                     double* elapsed_s)
{
    std::vector<uint32_t> users(n_nodes, 0);
    for (uint32_t i = 0; i < n_nodes; i++)
        if (op[i] >= 0)
            for (uint32_t k = 0; k < n_in[i]; k++) users[in[3 * i + k]]++;
    const auto t0 = std::chrono::steady_clock::now();
    int error = 0;
    for (uint32_t j = 0; j < n_order && !error; j++) {
        const uint32_t node = order[j];
        const spf_value* operands[3] = {nullptr, nullptr, nullptr};
        for (uint32_t k = 0; k < n_in[node]; k++) operands[k] = values[in[3 * node + k]];
        spf_value* out = nullptr;
        const spf_status st = submit(pool, (spf_graph_op)op[node], operands, n_in[node], param[node], &out, nullptr);
        if (st != SPF_OK) { error = st; break; }
        values[node] = out;
        for (uint32_t k = 0; k < n_in[node]; k++) {
            const uint32_t src = in[3 * node + k];
            if (--users[src] == 0 && !keep[src] && values[src]) { release(values[src]); values[src] = nullptr; }
        }
        // (flush, may be null: the last of a run of circuit bootstraps has been pushed — the conversions at the head of a circuit
        // are complete and milliseconds long: they start now, under the rest of the push)
        // (once: conversions in the middle of a circuit are better left to gather — the pool runs all of one rank as one batch)
        if (flush && op[node] == SPF_OP_CIRCUIT_BOOTSTRAP && (j + 1 == n_order || op[order[j + 1]] != SPF_OP_CIRCUIT_BOOTSTRAP)) {
            (void)flush(pool);
            flush = nullptr;
        }
    }
    const auto t_pushed = std::chrono::steady_clock::now();
    for (uint32_t j = 0; j < n_out && !error; j++) {
        const spf_status st = value_wait(values[outputs[j]]);
        if (st != SPF_OK) error = st;
    }
    if (getenv("SPF_PUSH_TRACE"))
        fprintf(stderr, "[push] %u operations pushed in %.0f us, outputs after another %.0f us\n", n_order,
                std::chrono::duration<double, std::micro>(t_pushed - t0).count(),
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_pushed).count());
    *elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (uint32_t i = 0; i < n_nodes; i++) // (what never found its last user: unused results, or everything after a failure)
        if (op[i] >= 0 && !keep[i] && values[i]) { release(values[i]); values[i] = nullptr; }
    return error;
}


}
