// spf_pool.hpp — call-coalescing front end.
//
// The reference executes one FHE operation per rayon task and calls `Evaluation` concurrently from
// many threads with ONE ciphertext each (parasol_runtime/src/circuit_processor/mod.rs:192-253:
// `execute_task` -> `rayon::spawn` -> `exec_op`; flow control by a bounded token channel, :130-190).
// A GPU wants thousands of ciphertexts per launch.  The pool bridges the two without touching the
// scheduler: every caller *submits* its single-ciphertext operation and blocks in *wait*, exactly
// like the synchronous call it replaces; a worker thread gathers whatever operations of one kind are
// pending (up to max_batch, or after max_wait_us since the oldest arrived), runs them as ONE batch
// through the ordinary entry points, scatters the outputs and wakes the callers.  Errors follow the
// reference's first-error-wins rule per batch: the batch's status is returned to each of its waiters.
#pragma once
#include "../../include/spf_hip.h"

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <exception>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace spf_pool_impl {

enum Op { OP_KEYSWITCH = 0, OP_CBS = 1, OP_CMUX = 2, OP_GATE_CBS = 3, N_OPS = 4 };

struct Request {
    const void* in0;
    const void* in1;
    const void* in2;
    void* out;
    uint64_t ticket;
    std::chrono::steady_clock::time_point t;
};

} // namespace spf_pool_impl

struct spf_pool {
    spf_ctx* ctx = nullptr;
    spf_params prm{};
    size_t max_batch = 4096;
    std::chrono::microseconds max_wait{200};
    std::mutex mu;
    std::condition_variable cv_work, cv_done, cv_space;
    std::deque<spf_pool_impl::Request> q[spf_pool_impl::N_OPS];
    std::unordered_map<uint64_t, spf_status> done;
    std::unordered_set<uint64_t> open;  // submitted and not yet collected by spf_pool_wait
    std::unordered_set<uint64_t> claimed; // tickets some thread is already waiting for (a ticket has ONE waiter)
    size_t blocked = 0;                 // callers inside submit() / spf_pool_wait(): destroy waits until they have left
    std::condition_variable cv_idle;
    size_t max_inflight = 16384;        // submit blocks while this many tickets are open (back-pressure)
    uint64_t next_ticket = 1;
    uint64_t n_ops = 0, n_launches = 0;
    bool stop = false;
    std::thread worker;
    // gather / scatter staging (host)
    std::vector<uint8_t> h_in0, h_in1, h_in2, h_out;

    size_t lwe0_bytes() const { return ((size_t)prm.lwe_dimension + 1) * 8; }
    size_t lwe1_bytes() const { return ((size_t)prm.glwe_size * prm.polynomial_degree + 1) * 8; }
    size_t glwe_bytes() const { return (size_t)(prm.glwe_size + 1) * prm.polynomial_degree * 8; }
    size_t ggsw_bytes() const
    {
        return (size_t)(prm.glwe_size + 1) * prm.cbs_radix_count * (prm.glwe_size + 1) * (prm.polynomial_degree / 2) * 16;
    }

    void in_out_sizes(int op, size_t (&in)[3], size_t& out) const
    {
        using namespace spf_pool_impl;
        in[0] = in[1] = in[2] = 0;
        switch (op) {
        case OP_KEYSWITCH: in[0] = lwe1_bytes(); out = lwe0_bytes(); break;
        case OP_CBS: in[0] = lwe0_bytes(); out = ggsw_bytes(); break;
        case OP_GATE_CBS: in[0] = lwe1_bytes(); out = ggsw_bytes(); break;
        default: in[0] = ggsw_bytes(); in[1] = glwe_bytes(); in[2] = glwe_bytes(); out = glwe_bytes(); break;
        }
    }

    spf_status run_batch(int op, std::vector<spf_pool_impl::Request>& batch)
    {
        using namespace spf_pool_impl;
        const size_t B = batch.size();
        size_t in[3], out;
        in_out_sizes(op, in, out);
        std::vector<uint8_t>* hin[3] = {&h_in0, &h_in1, &h_in2};
        for (int k = 0; k < 3; k++) {
            if (!in[k]) continue;
            hin[k]->resize(B * in[k]);
            for (size_t i = 0; i < B; i++) {
                const void* src = k == 0 ? batch[i].in0 : (k == 1 ? batch[i].in1 : batch[i].in2);
                std::memcpy(hin[k]->data() + i * in[k], src, in[k]);
            }
        }
        h_out.resize(B * out);
        spf_status st;
        switch (op) {
        case OP_KEYSWITCH:
            st = spf_keyswitch_lwe_l1_lwe_l0_batch(ctx, B, (const uint64_t*)h_in0.data(), (uint64_t*)h_out.data());
            break;
        case OP_CBS:
            st = spf_circuit_bootstrap_batch(ctx, B, (const uint64_t*)h_in0.data(), (double*)h_out.data());
            break;
        case OP_GATE_CBS: // FheOp::KeyswitchL1toL0 -> FheOp::CircuitBootstrap, the level-0 LWE stays in HBM
            st = spf_keyswitch_circuit_bootstrap_batch(ctx, B, (const uint64_t*)h_in0.data(), (double*)h_out.data());
            break;
        default:
            st = spf_cmux_batch(ctx, B, (const double*)h_in0.data(), (const uint64_t*)h_in1.data(),
                                (const uint64_t*)h_in2.data(), (uint64_t*)h_out.data());
            break;
        }
        if (st == SPF_OK)
            for (size_t i = 0; i < B; i++) std::memcpy(batch[i].out, h_out.data() + i * out, out);
        return st;
    }

    void loop()
    {
        using namespace spf_pool_impl;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] {
                if (stop) return true;
                for (auto& d : q) if (!d.empty()) return true;
                return false;
            });
            if (stop) {
                bool any = false;
                for (auto& d : q) any = any || !d.empty();
                if (!any) return;
            }
            // the kind whose oldest request has waited longest
            int op = -1;
            for (int k = 0; k < N_OPS; k++)
                if (!q[k].empty() && (op < 0 || q[k].front().t < q[op].front().t)) op = k;
            // let the batch fill: until max_batch, or max_wait after its oldest member arrived
            auto deadline = q[op].front().t + max_wait;
            while (!stop && q[op].size() < max_batch && std::chrono::steady_clock::now() < deadline)
                cv_work.wait_until(lk, deadline);
            std::vector<Request> batch;
            while (!q[op].empty() && batch.size() < max_batch) {
                batch.push_back(q[op].front());
                q[op].pop_front();
            }
            lk.unlock();
            spf_status st;
            try {
                st = run_batch(op, batch);
            } catch (const std::exception&) { // bad_alloc while staging a batch: its waiters get an error, the pool lives on
                st = SPF_ERR_HIP;
            }
            lk.lock();
            n_launches++;
            n_ops += batch.size();
            for (auto& r : batch) done[r.ticket] = st;
            cv_done.notify_all();
        }
    }

    spf_status submit(int op, const void* a, const void* b, const void* c, void* out, uint64_t* ticket)
    {
        if (!a || !out || !ticket) return SPF_ERR_INVALID_ARGUMENT;
        std::unique_lock<std::mutex> lk(mu);
        // back-pressure: a producer that runs ahead of its own waits blocks here instead of growing the queues
        blocked++;
        cv_space.wait(lk, [&] { return stop || open.size() < max_inflight; });
        blocked--;
        if (stop) {
            cv_idle.notify_all(); // spf_pool_destroy waits for blocked callers to leave before it frees the pool
            return SPF_ERR_INVALID_ARGUMENT;
        }
        try {
            spf_pool_impl::Request r{a, b, c, out, next_ticket, std::chrono::steady_clock::now()};
            q[op].push_back(r);
            open.insert(r.ticket);
            *ticket = r.ticket;
            next_ticket++;
        } catch (const std::exception&) {
            return SPF_ERR_HIP;
        }
        cv_work.notify_one();
        return SPF_OK;
    }
};
