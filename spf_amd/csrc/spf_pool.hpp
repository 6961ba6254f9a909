// spf_pool.hpp — call-coalescing front end.
//
// The reference executes one FHE operation per rayon task and calls `Evaluation` concurrently from
// many threads with ONE ciphertext each (parasol_runtime/src/circuit_processor/mod.rs:192-253:
// `execute_task` -> `rayon::spawn` -> `exec_op`; flow control by a bounded token channel, :130-190).
// A GPU wants thousands of ciphertexts per launch.  The pool bridges the two without touching the
// scheduler: every caller *submits* its single-ciphertext operation and blocks in *wait*, exactly
// like the synchronous call it replaces.
//
// r04: a pipeline instead of one worker that gathered, ran and scattered a batch at a time through pageable vectors
// (r03: the drop-in scenario — T threads, one ciphertext per call — moved 256 KiB of GGSW per gate through ONE thread's
// memcpy, twice, with the GPU idle meanwhile):
//   * the CALLERS move the bytes: `submit` takes a slot of the batch that is filling and copies its input straight into
//     that batch's PINNED staging buffer (outside the lock, all submitting threads in parallel); `wait` copies the
//     caller's own output out of the pinned buffer once the batch is done — T threads scatter T outputs at once;
//   * a launcher thread closes a batch when it is full, or — the pool's kernels off the GPU — at once when it holds one and
//     a half ciphertexts per CU, else when no member has arrived for max_wait_us or an eighth of the last batch's GPU time
//     (everything that arrives during kernel k, and the callers of batch k coming straight back, is batch k + 1), and enqueues
//     host-to-device copy, kernels and device-to-host copy on three streams tied by events: the copies of batch k + 1
//     and k - 1 run under the kernels of batch k;
//   * a completion thread waits for each batch's last event and wakes its waiters — through ONE futex word per batch, not
//     a condition variable of the pool: a thousand sleeping callers on one condition variable + mutex cost 24 s of kernel
//     time per 1.9 s of wall time (a wake storm per batch, then a convoy on the mutex) and ran the process into its CPU
//     quota (tools/pool_probe.py: 15 of 18 scheduler periods throttled); the pool's mutex is held for a few hundred
//     nanoseconds at a time (slot bookkeeping), never across a copy, a sleep or a HIP call;
//   * sixteen staging sets (r04: six; pinned host + device buffers, grown on demand), each with its own kernel stream and its own
//     intermediates.  A set returns to the pool when every ticket of its batch has been collected; tickets nobody waits for
//     are delivered by the launcher after a grace period, so that abandoned tickets cannot wedge the pipeline.
// r05: several batches RESIDENT on the GPU at once.  Synchronous callers make throughput = callers / (kernel latency + host
// turn-around): r04 ran one batch's kernels at a time (one stream, the context's intermediates), so a thousand callers took turns
// as two groups of 512 on the two-per-workgroup shape, and 256 callers idled the GPU for the 2 ms it takes them to copy 64 MB out
// and come back (0.72 of the device-resident rate either way).  Now a batch closes as soon as it holds a QUARTER of the callers
// in the pool (or everybody who can come is there), every batch is launched with the workgroup shape the WHOLE population would
// get (one / two / four ciphertexts per workgroup: launch_blind_rotate's per_wg_hint), so that up to four batches tile the CUs
// side by side, each on its set's stream: while one group copies out and comes back, the others compute.
// r06: operations BY HANDLE (spf_values.hpp).  The same submit / wait convention with device-resident operands and result:
// a handle batch has no staging copies at all — the launcher writes one pointer row per operand into the set's pinned table,
// the CMUX family reads its operands where their producers left them (`CmuxArgs::ptrs`), the small operands of the other kinds
// are packed by `gather_rows_kernel` unless they already lie consecutively (a keyswitch batch feeding a circuit-bootstrap
// batch), and the outputs of the batch are ONE block of the pool's arena that the result values view.  Handle batches use
// lanes of their own (they never mix with host-pointer callers); the cheap kinds take ONE caller group and no pacing — a
// 15 us CMUX level gains nothing from four resident batches.  Completion is per staging set (one completer thread each), so
// a CMUX batch that finishes in 15 us is handed back at once although a 4 ms bootstrap batch was enqueued before it.
// r06 (late): PENDING results as operands.  The blocking convention costs a thread sleep and wake-up per operation and keeps one
// operation in flight per thread; so a `_v` submit also takes operands whose producing operation has not run yet, needs no
// ticket, and spf_value_wait waits for a value.  Operations on pending operands are DEFERRED: batches by (depth, kind,
// parameter) in a table of their own (`deferred`), closed together shallowest first (a wait, or the quiet time), launched in
// that order on ONE in-order stream — a pushed circuit runs as the level plan a gate graph would have, built while it arrives.
// Errors follow the reference's first-error-wins rule per batch: the batch's status is returned to each of its waiters (and to
// every batch whose operands came out of it).
#pragma once
#include "../../include/spf_hip.h"

#include <algorithm>
#include <map>
#include <tuple>
#include <atomic>
#include <chrono>
#include <climits>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <exception>
#include <memory>
#include <pthread.h>
#include <sys/prctl.h>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace spf_pool_impl {

// one kind per `FheOp` that `CircuitProcessor::exec_op` hands to `Evaluation` (circuit_processor/mod.rs:329-540), plus the
// KeyswitchL1toL0 -> CircuitBootstrap chain
enum Op {
    OP_KEYSWITCH = 0, OP_CBS = 1, OP_CMUX = 2, OP_GATE_CBS = 3,
    OP_SAMPLE_EXTRACT = 4, OP_NOT = 5, OP_GLWE_ADD = 6, OP_MUL_XN = 7, OP_MULTIPLY_GGSW_GLWE = 8, OP_GLEV_CMUX = 9, OP_SCHEME_SWITCH = 10,
    N_OPS = 11
};
// the kinds whose batches are worth keeping resident side by side (milliseconds on the GPU): caller groups, pacing and the
// "previous batch still out" rule apply to these; everything else is microseconds per launch
inline bool heavy(int op) { return op == OP_CBS || op == OP_GATE_CBS; }
inline bool cmux_family(int op) { return op == OP_CMUX || op == OP_GLEV_CMUX || op == OP_MULTIPLY_GGSW_GLWE; }
constexpr int kMaxGroups = 4;     // caller groups per operation kind: that many batches of a kind resident on the GPU at once
constexpr int kSets = 4 * kMaxGroups; // per group one batch in flight / being collected and one filling; by handle up to split + 2 bootstrap batches
                                      // resident beside the cheap kinds' batches

constexpr size_t kMaxStagingBytes = (size_t)512 << 20; // per buffer of a set: caps the batch of the operations with 256 KiB outputs

struct Slot {
    void* out;
    uint64_t ticket;
    uint8_t delivered; // 0: output still in the staging set; 1: being copied to `out` by reclaim() on the owner's behalf (the lock is
                       // dropped for the copy: a wait() that arrives meanwhile parks on cv_deliver); 2: in `out`
    uintptr_t who;  // the submitting thread (see `last_members`)
    uint8_t claim;  // 0: nobody waits for the ticket yet; 1: its (one) waiter is inside spf_pool_wait; 2: collected.  Host-pointer
                    // batches change it under the pool's mutex (reclaim() reads it there); by handle it is a lock-free CAS
    spf_value* vin[3]; // by handle: the operands (retained until the batch has run) ...
    spf_value* vout;   // ... and the result (the batch's own reference; the caller holds another)
};

struct Staging {
    void* h_in[3] = {nullptr, nullptr, nullptr};
    void* h_out = nullptr;
    void* d_in[3] = {nullptr, nullptr, nullptr};
    void* d_out = nullptr;
    void* d_mid = nullptr;
    void** h_ptrs = nullptr;  // by handle: pinned pointer table the kernels read (operand rows / CMUX units)
    size_t cap_in[3] = {0, 0, 0}, cap_out = 0, cap_mid = 0, cap_ptrs = 0; // host buffers (bytes; cap_ptrs in pointers)
    size_t dcap_in[3] = {0, 0, 0}, dcap_out = 0;                          // device buffers (bytes)
    bool busy = false;
    hipStream_t sk = nullptr; // this set's kernels (created with the pool)
    Scratch scr;              // ... and their intermediates
    size_t scr_cap = 0;       // operations the intermediates are sized for
};

inline void futex_wait(std::atomic<uint32_t>* w, uint32_t expected);
inline void futex_wake_all(std::atomic<uint32_t>* w);
// The pool's mutex.  Its critical sections are a few hundred nanoseconds (slot bookkeeping) but dozens of callers reach them in
// the same microseconds — the gates of one circuit level returning from their wait and submitting the next (r06: with a plain
// pthread mutex every one of them slept on the futex and was woken in turn, a convoy of ~10 us per caller: 300 us per level of a
// 32-bit adder).  So: spin briefly, then sleep (the three-state futex lock: 0 free, 1 held, 2 held with sleepers).
struct PoolMutex {
    std::atomic<uint32_t> s{0};
    bool try_lock()
    {
        uint32_t z = 0;
        return s.load(std::memory_order_relaxed) == 0 && s.compare_exchange_strong(z, 1, std::memory_order_acquire);
    }
    void lock()
    {
        for (int i = 0; i < 400; i++) {
            if (try_lock()) return;
            __builtin_ia32_pause();
        }
        uint32_t c = 0;
        if (s.compare_exchange_strong(c, 1, std::memory_order_acquire)) return;
        if (c != 2) c = s.exchange(2, std::memory_order_acquire);
        while (c != 0) {
            futex_wait(&s, 2);
            c = s.exchange(2, std::memory_order_acquire);
        }
    }
    void unlock()
    {
        if (s.fetch_sub(1, std::memory_order_release) != 1) {
            s.store(0, std::memory_order_release);
            (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(&s), FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0);
        }
    }
};

inline void futex_wait(std::atomic<uint32_t>* w, uint32_t expected)
{
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0);
}
inline void futex_wake_all(std::atomic<uint32_t>* w)
{
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}

struct Batch {
    int op = 0, set = 0, lane = 0; // lane = ((by handle ? N_OPS : 0) + op) * kMaxGroups + caller group
    bool by_handle = false;
    std::shared_ptr<spf_value_impl::Block> out_blk; // by handle: the outputs of the batch
    size_t cap = 0;               // slots of this batch
    size_t n = 0, n_ready = 0;    // slots taken / inputs copied in
    size_t n_collected = 0;
    size_t n_returning = 0;       // members submitted by a thread that was in the previous batch of this kind
    int per_wg = 0;               // workgroup shape of its bootstrap (the population's, decided when it is enqueued)
    uint64_t param = 0;           // SampleExtract index / MulXN amount: one value per batch (a different one opens a new batch)
    bool closed = false, done = false;
    bool kernels_done = false;    // its kernels have left the GPU (the device-to-host copy may still run): the next batch may go
    // Deferred batches (by handle, operands that are still PENDING results of this pool): not in `filling` but in the pool's
    // `deferred` table under (depth, kind, parameter); `depth` = 1 + the deepest batch an operand comes from, so launching in
    // depth order is a topological order; `deps` = the batches the operands come from (the set's stream waits for their events);
    // the staging set is taken when the batch is launched, not when it is opened (a circuit has more levels than there are sets)
    bool deferred = false;
    int64_t depth = 0;
    int rank = 0; // bootstrap batches on the longest path to the batch's operations (part of its key: see submit_impl)
    std::vector<std::shared_ptr<Batch>> deps;
    // The outputs leave the GPU in up to kMaxChunks copies (each a multiple of kWordSlots slots, all but the last equal), each with
    // its own event.  The waiters sleep on the word of their slot group (futex, 0 -> 1 when the group's bytes are in pinned
    // memory or the batch failed): the callers of the first chunk copy out and come back while the later chunks are still
    // crossing PCIe.  (Words per 64 slots rather than per chunk: a waiter may go to sleep before the batch is closed, when
    // its size — and so the chunk boundaries — is not known yet.)
    static constexpr size_t kWordSlots = 64;
    static constexpr int kMaxWords = 64; // 4096 slots; the last word also takes whatever lies beyond
    static constexpr int kMaxChunks = 16;
    static int word_of(size_t slot) { return (int)std::min<size_t>(slot / kWordSlots, kMaxWords - 1); }
    std::atomic<uint32_t> chunk_word[kMaxWords] = {};
    size_t chunk_slots = 0; // slots per copy (set with n_chunks)
    int n_chunks = 0; // set when the batch is enqueued
    std::atomic<spf_status> st{SPF_OK}; // (atomic: the completion thread may still record a late copy failure while the waiters
                                        // of an earlier chunk read it)
    std::chrono::steady_clock::time_point t0, t_last, t_done, t_close, t_enq, t_ready, t_sync;
    std::vector<Slot> slots;
    hipEvent_t ev_in = nullptr, ev_k = nullptr;
    hipEvent_t ev_chunk[kMaxChunks] = {}; // the last one of a batch is "everything is out"
    void destroy_events()
    {
        for (hipEvent_t* e : {&ev_in, &ev_k})
            if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
        for (hipEvent_t& e : ev_chunk)
            if (e) { (void)hipEventDestroy(e); e = nullptr; }
    }
    void wake_word(int w)
    {
        chunk_word[w].store(1, std::memory_order_release);
        futex_wake_all(&chunk_word[w]);
    }
    // By handle the waiters are woken as a TREE: they sleep in groups of eight on a word per group; whoever completes the batch
    // sets every word, wakes group 0, and every waiter that comes through wakes one child group (group g's member j: group
    // 8 g + 1 + j) before it goes on.  One thread waking a thousand sleepers one by one took 6 us each (r06: 1.5 ms of a 256-caller
    // batch's 5.8 ms cycle); the tree is three levels deep.  The completing thread also walks all groups in order and wakes what
    // nobody has woken yet, so a waiter that never comes (an abandoned ticket) leaves no group asleep.
    static constexpr size_t kTreeGroup = 8;
    std::unique_ptr<std::atomic<uint32_t>[]> gword, gwoken, gsleep; // by handle: [cap / 8 + 1]
    size_t n_groups() const { return (n + kTreeGroup - 1) / kTreeGroup; }
    // a waiter announces itself in gsleep before it looks at the word for the last time; the waker sets the word before it looks at
    // gsleep (both sequentially consistent): a group nobody sleeps on costs no system call — operations pushed without a ticket have
    // no waiters at all, and a batch of 400 of them was 50 futex calls of ~2 us on the launcher's critical path
    void sleep_on_group(size_t g)
    {
        gsleep[g].fetch_add(1, std::memory_order_seq_cst);
        while (gword[g].load(std::memory_order_seq_cst) == 0) futex_wait(&gword[g], 0);
    }
    void wake_group(size_t g)
    {
        if (gwoken[g].exchange(1, std::memory_order_acq_rel) == 0 && gsleep[g].load(std::memory_order_seq_cst) != 0) futex_wake_all(&gword[g]);
    }
    void wake_tree() // by the thread that completed the batch (n is final)
    {
        const size_t ng = n_groups();
        for (size_t g = 0; g < ng; g++) gword[g].store(1, std::memory_order_seq_cst);
        for (size_t g = 0; g < ng; g++) wake_group(g);
    }
    void wake_chunk(int i) // the words of copy i
    {
        // (the last word stands for every slot from 64 * (kMaxWords - 1) on, however many: only the last copy wakes it)
        const int w0 = word_of((size_t)i * chunk_slots);
        const int w1 = i + 1 < n_chunks ? std::min(word_of((size_t)(i + 1) * chunk_slots - 1), kMaxWords - 2) : kMaxWords - 1;
        for (int w = w0; w <= w1; w++)
            if (chunk_word[w].load(std::memory_order_relaxed) == 0) wake_word(w);
    }
};

// open tickets -> (batch, slot), sharded by ticket number: a submit inserts and a wait erases under ONE shard's lock, not the
// pool's (r06: with everything under the pool's mutex 64 callers of a 15 us CMUX took three turns each on one lock per gate)
struct TicketShard {
    PoolMutex mu;
    std::unordered_map<uint64_t, std::pair<std::shared_ptr<Batch>, size_t>> map;
};
constexpr size_t kTicketShards = 64;

// The synchronous caller — submit, then wait for that very ticket on the same thread, what a rayon task of the reference does —
// finds its batch here without any lookup.
struct LastSubmit {
    uint64_t pool_gen = 0, ticket = 0;
    std::shared_ptr<Batch> batch;
    size_t slot = 0;
};
inline LastSubmit& last_submit()
{
    thread_local LastSubmit t;
    return t;
}
inline uint64_t next_pool_generation()
{
    static std::atomic<uint64_t> g{1};
    return g.fetch_add(1);
}

} // namespace spf_pool_impl

struct spf_group;

struct spf_pool {
    using Batch = spf_pool_impl::Batch;
    // A pool over a device group (spf_pool_create_group) is only a dealer: one ordinary pool per member, the calling thread's
    // home member decided on its first submit; tickets carry the member in their top byte.  Nothing below `members` is used then.
    static constexpr int kMemberShift = 56;
    std::vector<spf_pool*> members;
    spf_group* grp = nullptr;
    std::mutex deal_mu;
    std::unordered_map<uintptr_t, int> home; // calling thread -> member
    size_t next_home = 0;
    spf_ctx* ctx = nullptr;
    spf_params prm{};
    size_t max_batch = 4096;
    std::chrono::microseconds max_wait{200};
    using Mutex = spf_pool_impl::PoolMutex;
    Mutex mu;
    std::condition_variable_any cv_work, cv_space, cv_set, cv_idle, cv_deliver;
    static constexpr int kKinds = 2 * spf_pool_impl::N_OPS; // host-pointer kinds, then the same by handle
    static constexpr int kLanes = kKinds * spf_pool_impl::kMaxGroups;
    static int lane_of(int op, bool by_handle, int grp) { return ((by_handle ? spf_pool_impl::N_OPS : 0) + op) * spf_pool_impl::kMaxGroups + grp; }
    std::shared_ptr<spf_value_impl::Arena> arena;                                   // device memory of the values (spf_values.hpp)
    std::shared_ptr<Batch> filling[kLanes];                                           // per (operation kind, caller group)
    std::unordered_map<uintptr_t, size_t> home_group;                               // calling thread -> its arrival number (group = arrival % groups: sticky)
    size_t next_group = 0;
    std::unordered_map<uintptr_t, int> open_by_thread;                              // submitting thread -> its open tickets
    std::deque<std::shared_ptr<Batch>> closing;                                     // closed, waiting for their members' input copies
    std::deque<std::shared_ptr<Batch>> in_flight;                                   // enqueued, each waited for by its set's completer
    std::deque<std::shared_ptr<Batch>> collecting;                                  // done, not yet fully collected
    spf_pool_impl::TicketShard tickets[spf_pool_impl::kTicketShards]; // open tickets -> (batch, slot)
    spf_pool_impl::TicketShard& shard_of(uint64_t ticket) { return tickets[ticket % spf_pool_impl::kTicketShards]; }
    std::atomic<size_t> n_open{0};        // open tickets (back-pressure)
    const uint64_t gen = spf_pool_impl::next_pool_generation(); // (a new pool at a recycled address is another pool: LastSubmit)
    std::atomic<size_t> blocked{0};       // callers inside submit() / wait(): destroy waits until they have left
    std::atomic<size_t> space_waiters{0};  // submitters parked on back-pressure
    size_t set_waiters = 0;                // ... / on a staging set
    size_t max_inflight = 16384;          // submit blocks while this many tickets are open (back-pressure)
    std::atomic<size_t> heavy_open{0};    // open tickets of the bootstrap kinds: the population that shapes their batches
    size_t groups = 0;                    // caller groups in use: 0 = by population (groups_now), else SPF_POOL_GROUPS = 1 .. kMaxGroups
    int pace_div = 0;                     // pacing: a batch starts no sooner than 1 / pace_div of a batch's GPU time after the previous one (0 = groups; SPF_POOL_PACE)
    uint64_t next_ticket = 1;
    uint64_t n_ops = 0, n_launches = 0;
    std::atomic<bool> stop{false};
    std::thread launcher, completers[spf_pool_impl::kSets];
    spf_pool_impl::Staging sets[spf_pool_impl::kSets];
    // Table sets: what a CHEAP deferred batch needs between its launch and its kernels — the pinned pointer table and the rows
    // its scattered operands are packed into; no stream (s_def), no intermediates, no completer thread (the launcher polls its
    // event).  Many more of them than staging sets: the launcher runs ahead of the GPU by as many levels as there are tables, and
    // the sixteen staging sets stay free for the bootstrap batches (several conversions of a circuit in flight side by side).
    static constexpr int kTableSets = 96;
    spf_pool_impl::Staging tsets[kTableSets];
    static bool table_batch(const Batch& b) { return b.deferred && !spf_pool_impl::heavy(b.op); }
    spf_pool_impl::Staging& staging_of(const Batch& b) { return table_batch(b) ? tsets[b.set] : sets[b.set]; }
    std::shared_ptr<Batch> flying[spf_pool_impl::kSets]; // the enqueued batch of each set (one at most)
    std::condition_variable_any cv_fly[spf_pool_impl::kSets];
    uint64_t n_reclaimed = 0;             // outputs delivered on their owners' behalf (reclaim)
    uint64_t n_handle_ops = 0, n_handle_launches = 0;
    int stream_concurrency = 0;           // how many of the sets' streams were seen running at once when the pool was made (0: not measured)
    int n_sets = spf_pool_impl::kSets;    // staging sets in use (SPF_POOL_SETS: tests fill every set with three)
    size_t split = 4;                     // by handle a bootstrap batch goes as soon as it holds 1 / split of the callers (SPF_POOL_SPLIT; see submit_impl)
    int spin_us = 0;                      // how long a waiter of a cheap operation by handle looks before it sleeps (SPF_POOL_SPIN_US; measured on
                                          // the bench's host — 64 callers on 16 CPUs — any spinning loses: 9.7-10.8 ms per 32-bit adder at 0, 10.7-16 at 40-200 us)
    int hot_us = 300;                     // how long the launcher keeps polling after its last piece of work (SPF_POOL_HOT_US)
    std::atomic<uint64_t> work_epoch{0};  // bumped whenever the launcher has something new to look at
    std::atomic<bool> launcher_asleep{false};
    std::vector<std::shared_ptr<Batch>> polling; // the launcher's own: cheap batches by handle it enqueued and completes itself
    // by handle, operations on PENDING operands: open batches by (depth, kind, parameter) — see Batch::deferred, flush_deferred
    std::map<std::tuple<int64_t, int, uint64_t, int>, std::shared_ptr<Batch>> deferred; // (depth, kind, parameter, rank)
    std::vector<std::shared_ptr<Batch>> deferred_full; // ... and the ones that filled up (the next one of their key is bigger): they go with the rest
    std::chrono::steady_clock::time_point t_last_deferred{};
    uint64_t n_deferred_ops = 0;
    uint64_t n_shape[3] = {0, 0, 0};      // bootstrap launches by blind-rotation shape: eight waves per ciphertext / two / four per workgroup
    // The cheap DEFERRED batches' stream: all of them, in the order they were launched — a batch behind the batch its operands come
    // from needs no event (measured: a level of a pushed 32-bit adder on its set's own stream, tied to the previous level's stream
    // by an event, took ~150 us on the GPU; 15-20 us in order on one stream).  A deferred BOOTSTRAP batch is milliseconds long
    // whatever its width and runs on its staging set's own stream, side by side with the others (the conversions inside a 32 x 32
    // multiplication become ready at 32 different depths); nothing is ever enqueued BEHIND one: what takes its results is launched
    // when it has finished (launch_loop) — hipStreamWaitEvent on the event of a running bootstrap batch blocked the calling thread
    // until the batch was through (3.9-4.2 ms per call on this runtime: the launcher ran a pushed multiplication's 32 conversion
    // batches one after the other, 172 ms; the gate graph, which moves them to common levels: 22).
    // Batches of the ordinary lanes keep their sets' streams.
    hipStream_t s_def = nullptr;
    bool create_def_streams() { return hipStreamCreateWithFlags(&s_def, hipStreamNonBlocking) == hipSuccess; }
    // (a stream of the highest priority instead — hardware queues are kept per priority — changed nothing: 96 against 95 ms)
    void destroy_def_streams()
    {
        if (s_def) (void)hipStreamDestroy(s_def);
        s_def = nullptr;
    }
    int n_table_sets = kTableSets; // (SPF_POOL_TABLE_SETS: experiments)
    int max_def_heavy = 3;         // deferred bootstrap batches in flight side by side (SPF_POOL_DEF_HEAVY; 0 = no limit).  A pushed 32 x 32
                                   // multiplication (31 conversion batches of four, at 31 depths): 1: 152-160 ms, 2: 99-102, 3: 79-84, 4: 92-101,
                                   // no limit: 98-104 — beside five or more of them the levels on s_def stop being served (a CMux level enqueued
                                   // beside seven running conversions completed 8 ms later, when the last of them had finished)
    hipStream_t stream_of(const Batch& b) const { return table_batch(b) ? s_def : sets[b.set].sk; }
    std::chrono::milliseconds grace{200}; // after this long an uncollected output is delivered by the launcher
    std::vector<uintptr_t> last_members[kLanes]; // threads of the most recently finished batch of a lane, sorted
    size_t cap_hint[kKinds] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64}; // slots of the next batch of a kind: doubles whenever a batch fills up
                                                              // (pinned staging is sized by what the callers actually produce:
                                                              // 2048 slots of 256 KiB would pin 0.5 GiB per set up front)
    bool preparing[kLanes] = {}; // a submitter is allocating a set for this lane (lock dropped)
    int outstanding[kLanes] = {}; // closed batches of the lane that have not completed yet

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
        const size_t glev = glwe_bytes() * prm.cbs_radix_count;
        switch (op) {
        case OP_KEYSWITCH: in[0] = lwe1_bytes(); out = lwe0_bytes(); break;
        case OP_CBS: in[0] = lwe0_bytes(); out = ggsw_bytes(); break;
        case OP_GATE_CBS: in[0] = lwe1_bytes(); out = ggsw_bytes(); break;
        case OP_SAMPLE_EXTRACT: in[0] = glwe_bytes(); out = lwe1_bytes(); break;
        case OP_NOT: case OP_MUL_XN: in[0] = glwe_bytes(); out = glwe_bytes(); break;
        case OP_GLWE_ADD: in[0] = in[1] = glwe_bytes(); out = glwe_bytes(); break;
        case OP_MULTIPLY_GGSW_GLWE: in[0] = ggsw_bytes(); in[1] = glwe_bytes(); out = glwe_bytes(); break;
        case OP_GLEV_CMUX: in[0] = ggsw_bytes(); in[1] = in[2] = glev; out = glev; break;
        case OP_SCHEME_SWITCH: in[0] = glev; out = ggsw_bytes(); break;
        default: in[0] = ggsw_bytes(); in[1] = glwe_bytes(); in[2] = glwe_bytes(); out = glwe_bytes(); break; // OP_CMUX
        }
    }
    size_t batch_cap(int op) const
    {
        size_t in[3], out;
        in_out_sizes(op, in, out);
        const size_t big = std::max(std::max(in[0], in[1]), std::max(in[2], out));
        return std::max<size_t>(1, std::min(max_batch, spf_pool_impl::kMaxStagingBytes / big));
    }

    // ---- staging sets (prepare_set runs with `mu` DROPPED — hipFree / hipMalloc / hipHostMalloc wait for the device —
    // while `preparing[op]` keeps other submitters of the kind parked; growing a buffer is rare: first use of an operation kind)
    static bool grow_host(void*& p, size_t& cap, size_t bytes)
    {
        if (cap >= bytes) return true;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        if (hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return false; // (the GPU reads / the host reads it in place)
        cap = bytes;
        return true;
    }
    // (the new buffer first, the old one freed afterwards: a failed allocation leaves the pointer and its capacity as they were)
    static bool grow_dev(void*& p, size_t& have, size_t bytes)
    {
        if (have >= bytes) return true;
        void* q = nullptr;
        if (hipMalloc(&q, bytes) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (p) (void)hipFree(p);
        p = q;
        have = bytes;
        return true;
    }
    // pointers of a handle batch's table: one row of `cap` per operand, or four per CMUX unit
    size_t table_pointers(int op, size_t cap) const
    {
        return spf_pool_impl::cmux_family(op) ? 4 * cap * (op == spf_pool_impl::OP_GLEV_CMUX ? prm.cbs_radix_count : 1) + 3 * cap : 3 * cap;
    }
    bool prepare_set(spf_pool_impl::Staging& s, int op, size_t cap, bool by_handle, bool table_set = false)
    {
        size_t in[3], out;
        in_out_sizes(op, in, out);
        if (hipSetDevice(ctx->device) != hipSuccess) return false;
        for (int k = 0; k < 3; k++) {
            if (!in[k]) continue;
            if (by_handle && scattered_cmux(op)) continue; // read in place through the pointer table
            if (!grow_dev(s.d_in[k], s.dcap_in[k], cap * in[k])) return false;
            if (!by_handle && !grow_host(s.h_in[k], s.cap_in[k], cap * in[k])) return false;
        }
        if (by_handle) {
            size_t bytes = s.cap_ptrs * sizeof(void*);
            void* p = s.h_ptrs;
            if (!grow_host(p, bytes, table_pointers(op, cap) * sizeof(void*))) { s.h_ptrs = nullptr; s.cap_ptrs = 0; return false; }
            s.h_ptrs = static_cast<void**>(p);
            s.cap_ptrs = bytes / sizeof(void*);
        } else {
            if (!grow_dev(s.d_out, s.dcap_out, cap * out)) return false;
            if (!grow_host(s.h_out, s.cap_out, cap * out)) return false;
        }
        if (op == spf_pool_impl::OP_GATE_CBS && !grow_dev(s.d_mid, s.cap_mid, cap * lwe0_bytes())) return false;
        if ((op == spf_pool_impl::OP_KEYSWITCH || op == spf_pool_impl::OP_CBS || op == spf_pool_impl::OP_GATE_CBS) && s.scr_cap < cap) {
            // (both: a staging set serves any kind later; a table set only ever sees the keyswitch — no 160 KB of circuit-bootstrap
            // intermediates per slot for it)
            if (scratch_reserve(ctx, s.scr, cap, true, !table_set) != SPF_OK) return false;
            s.scr_cap = cap;
        }
        return true;
    }
    void free_sets()
    {
        (void)hipSetDevice(ctx->device);
        auto free_one = [](spf_pool_impl::Staging& s) {
            for (int k = 0; k < 3; k++) {
                if (s.h_in[k]) (void)hipHostFree(s.h_in[k]);
                if (s.d_in[k]) (void)hipFree(s.d_in[k]);
            }
            if (s.h_out) (void)hipHostFree(s.h_out);
            if (s.d_out) (void)hipFree(s.d_out);
            if (s.d_mid) (void)hipFree(s.d_mid);
            if (s.h_ptrs) (void)hipHostFree(s.h_ptrs);
            scratch_free(s.scr);
            if (s.sk) (void)hipStreamDestroy(s.sk);
            s = spf_pool_impl::Staging{};
        };
        for (auto& s : sets) free_one(s);
        for (auto& s : tsets) free_one(s);
    }

    // output of one slot to its caller's buffer (any thread; the pinned buffer is stable while the batch is collecting)
    void deliver(const Batch& b, size_t slot) const
    {
        size_t in[3], out;
        in_out_sizes(b.op, in, out);
        std::memcpy(b.slots[slot].out, static_cast<const uint8_t*>(sets[b.set].h_out) + slot * out, out);
    }
    // called with `mu` held when a ticket of `b` has been collected (or delivered on its owner's behalf)
    void collected_one(const std::shared_ptr<Batch>& b)
    {
        if (++b->n_collected == b->n) {
            sets[b->set].busy = false;
            for (auto it = collecting.begin(); it != collecting.end(); ++it)
                if (it->get() == b.get()) { collecting.erase(it); break; }
            b->destroy_events();
            if (set_waiters) cv_set.notify_all();
        }
    }

    // the CMUX family reads scattered operands in place where the tuned kernels serve the parameter set (CmuxArgs::ptrs);
    // elsewhere (generic family, another cbs radix) its operands are packed like everybody else's
    bool scattered_cmux(int op) const
    {
        return spf_pool_impl::cmux_family(op) && !ctx->generic && prm.cbs_radix_log == 4 && prm.cbs_radix_count == 4;
    }

    // `mu` held: the launcher has something new to look at.  It is only sent a wake-up (a system call) when it sleeps: while
    // operations keep coming it polls `work_epoch`.
    void poke()
    {
        work_epoch.fetch_add(1, std::memory_order_release);
        if (launcher_asleep.load(std::memory_order_acquire)) cv_work.notify_all();
    }

    spf_status submit(int op, const void* a, const void* b_in, const void* c, void* out, uint64_t* ticket, uint64_t param = 0)
    {
        if (!a || !out || !ticket) return SPF_ERR_INVALID_ARGUMENT;
        const void* src[3] = {a, b_in, c};
        return submit_impl(op, false, src, out, nullptr, nullptr, ticket, param);
    }
    // by handle: `v` are values of this pool, valid or still pending (checked by the caller, pool_submit_v); *result is a new value
    // that becomes valid when its batch has run (spf_pool_wait for the ticket, or spf_value_wait); `ticket` may be null
    spf_status submit_v(int op, spf_value* const* v, spf_value* result, uint64_t* ticket, uint64_t param = 0)
    {
        return submit_impl(op, true, nullptr, nullptr, v, result, ticket, param);
    }

    spf_status submit_impl(int op, bool by_handle, const void* const* src, void* out, spf_value* const* vin, spf_value* vout,
                           uint64_t* ticket, uint64_t param)
    {
        using namespace spf_pool_impl;
        blocked++;
        // (the last one out tells a destroy that waits for everybody to leave; the destroyer re-checks under the mutex)
        struct Leave { spf_pool* p; ~Leave() { if (--p->blocked == 0 && p->stop) { std::lock_guard<Mutex> g(p->mu); p->cv_idle.notify_all(); } } } leave{this};
        std::unique_lock<Mutex> lk(mu);
        // back-pressure: a producer that runs ahead of its own waits blocks here instead of growing the queues
        if (n_open.load() >= max_inflight) {
            space_waiters++;
            cv_space.wait(lk, [&] { return stop || n_open.load() < max_inflight; });
            space_waiters--;
        }
        if (stop) return SPF_ERR_INVALID_ARGUMENT;
        // The caller's group: dealt round-robin on the thread's first submit, kept from then on.  The threads of a group meet in
        // the same batches, call after call — so a group's batch is complete the moment the members of its previous batch are
        // back (everybody_is_back), whatever the other groups are doing, and a straggler only ever delays itself.
        // (By handle only the bootstrap kinds are dealt to groups: the cheap kinds are one batch per launch.)
        const uintptr_t who = (uintptr_t)pthread_self();
        int grp = 0;
        if (!by_handle) {
            size_t arrival;
            auto it = home_group.find(who);
            if (it != home_group.end()) arrival = it->second;
            else {
                arrival = next_group++;
                try {
                    home_group.emplace(who, arrival);
                } catch (const std::exception&) { // (out of memory: dealt again next time)
                }
            }
            grp = (int)(arrival % groups_now());
        }
        const int lane = lane_of(op, by_handle, grp);
        const int kind = (by_handle ? N_OPS : 0) + op;
        std::shared_ptr<Batch> b;
        // By handle an operand may be the PENDING result of an earlier submit to this pool (its batch open, closed or running): the
        // operation is then DEFERRED — it joins the open batch of its (depth, kind, parameter), depth = 1 + the deepest batch an
        // operand comes from.  Nothing is launched for it until flush_deferred: a whole circuit pushed by one thread without a
        // single wait becomes one batch per kind and level, launched in depth order, each ordered behind the batches its operands
        // come from by their events — the level batching of spf_graph_run, built while the operations arrive.
        std::shared_ptr<Batch> dep[3];
        int n_dep = 0;
        if (by_handle) {
            int64_t depth = 0;
            int rank = 0; // (operations behind different numbers of bootstrap batches do not share a batch: what waits for a conversion
                          // — a bootstrap batch is milliseconds — must not hold back the operations of its level that do not; the
                          // conversions inside a 32 x 32 multiplication are 31 levels apart and each was started when the previous
                          // one had finished, 165 ms, because every level held an adder gate that waited for the previous one)
            for (int k = 0; k < 3; k++) {
                if (!vin[k]) continue;
                const int vs = vin[k]->state.load(std::memory_order_acquire);
                if (vs == spf_value_impl::READY) continue;
                if (vs == spf_value_impl::FAILED) return SPF_ERR_INVALID_ARGUMENT;
                const std::shared_ptr<Batch>& p = vin[k]->producer; // (set and cleared under `mu`)
                if (!p || p->done) continue; // its batch is being handed back right now: the kernels have run
                depth = std::max(depth, p->depth + 1);
                rank = std::max(rank, p->rank + (heavy(p->op) ? 1 : 0));
                bool seen = false;
                for (int j = 0; j < n_dep; j++) seen = seen || dep[j] == p;
                if (!seen) dep[n_dep++] = p;
            }
            if (n_dep) {
                // The bootstrap kinds are keyed WITHOUT their depth: every deferred conversion behind the same number of bootstrap
                // batches joins ONE batch, whatever level its operand comes from (the batch's depth is its deepest member's).  A launch
                // of these kinds costs ~4 ms whatever its width, and the 128 conversions inside a 32 x 32 multiplication become ready
                // at 31 depths: 31 launches of four (80 ms pushed) or one of 128.  No cycle can come of it: whatever a bootstrap batch
                // of rank r needs has rank <= r, whatever needs it has rank > r.  (The early members wait for the late ones — the gate
                // graph's planner, which knows every consumer, moves them by their slack; here nothing is known about consumers.)
                const int64_t key_depth = heavy(op) ? -1 : depth;
                const auto key = std::make_tuple(key_depth, op, param, rank);
                try {
                    auto it = deferred.find(key);
                    if (it != deferred.end()) b = it->second;
                    else {
                        const size_t cap = std::min(batch_cap(op), std::max<size_t>(cap_hint[kind], 64));
                        b = std::make_shared<Batch>();
                        b->slots.resize(cap);
                        const size_t ng = cap / Batch::kTreeGroup + 1;
                        b->gword.reset(new std::atomic<uint32_t>[ng]);
                        b->gwoken.reset(new std::atomic<uint32_t>[ng]);
                        b->gsleep.reset(new std::atomic<uint32_t>[ng]);
                        for (size_t g = 0; g < ng; g++) { b->gword[g].store(0); b->gwoken[g].store(0); b->gsleep[g].store(0); }
                        b->op = op; b->set = -1; b->cap = cap; b->lane = lane; b->param = param; b->by_handle = true;
                        b->deferred = true; b->depth = depth; b->rank = rank;
                        const bool first = deferred.empty() && deferred_full.empty();
                        deferred.emplace(key, b);
                        if (first) poke(); // (the launcher arms the quiet time that flushes the table)
                    }
                    b->depth = std::max(b->depth, depth);
                    for (int j = 0; j < n_dep; j++)
                        if (std::find(b->deps.begin(), b->deps.end(), dep[j]) == b->deps.end()) b->deps.push_back(dep[j]);
                } catch (const std::exception&) {
                    return SPF_ERR_HIP;
                }
            }
        }
        while (!b) {
            b = filling[lane];
            if (b && b->param != param) { // (SampleExtract index / MulXN amount: the kernels take one value per launch)
                close_batch(lane);
                b.reset();
                continue;
            }
            if (b) break; // (a batch leaves `filling` the moment it is closed: what is there has room)
            // open a batch on a free staging set
            int set = -1;
            if (!preparing[lane]) // (somebody is allocating a set for this lane right now: its batch is about to appear)
                for (int i = 0; i < n_sets; i++) if (!sets[i].busy) { set = i; break; }
            if (set < 0) {
                // all sets held: wait for collectors; past the grace period deliver the oldest done batch's leftovers here
                set_waiters++;
                const bool timed_out = cv_set.wait_for(lk, std::chrono::milliseconds(20)) == std::cv_status::timeout;
                set_waiters--;
                if (timed_out && !preparing[lane]) reclaim(lk);
                if (stop) return SPF_ERR_INVALID_ARGUMENT;
                continue;
            }
            const size_t cap = std::min(batch_cap(op), cap_hint[kind]);
            sets[set].busy = true;
            preparing[lane] = true;
            lk.unlock(); // (allocation calls wait for the device: never under the pool's mutex)
            const bool prepared = prepare_set(sets[set], op, cap, by_handle);
            lk.lock();
            preparing[lane] = false;
            if (set_waiters) cv_set.notify_all();
            if (!prepared) {
                sets[set].busy = false;
                cv_set.notify_all();
                return SPF_ERR_HIP;
            }
            if (stop) { sets[set].busy = false; return SPF_ERR_INVALID_ARGUMENT; }
            try {
                b = std::make_shared<Batch>();
                b->slots.resize(cap); // (never grows: a slot's address is stable, its owner reads it without the mutex)
                if (by_handle) {
                    const size_t ng = cap / Batch::kTreeGroup + 1;
                    b->gword.reset(new std::atomic<uint32_t>[ng]);
                    b->gwoken.reset(new std::atomic<uint32_t>[ng]);
                        b->gsleep.reset(new std::atomic<uint32_t>[ng]);
                    for (size_t g = 0; g < ng; g++) { b->gword[g].store(0); b->gwoken[g].store(0); b->gsleep[g].store(0); }
                }
            } catch (const std::exception&) {
                sets[set].busy = false;
                cv_set.notify_all();
                return SPF_ERR_HIP;
            }
            b->op = op; b->set = set; b->cap = cap; b->lane = lane; b->param = param; b->by_handle = by_handle;
            filling[lane] = b;
            break;
        }
        const size_t slot = b->n;
        const uint64_t my_ticket = ticket ? next_ticket : 0; // (0: nobody will wait for this operation — spf_value_wait on its result instead)
        b->slots[slot] = Slot{out, my_ticket, 0, who, (uint8_t)(ticket ? 0 : 2), {nullptr, nullptr, nullptr}, nullptr};
        // a caller "comes back" when it submits with nothing else outstanding (the synchronous pattern); a thread that
        // submits many tickets before it waits for any is not waited for — its batches close on the timer or when full.
        // (Only where batches are closed by who is back: not for the cheap kinds by handle.)
        const bool tracks_callers = !by_handle;
        if (tracks_callers) {
            try {
                int& mine_open = open_by_thread[who]; // (may allocate)
                if (mine_open == 0 && std::binary_search(last_members[lane].begin(), last_members[lane].end(), who)) b->n_returning++;
                mine_open++;
            } catch (const std::exception&) {
                return SPF_ERR_HIP;
            }
        }
        if (heavy(op)) heavy_open++; // (taken back by spf_pool_wait, or — no ticket — when the batch is handed back)
        if (ticket) n_open++;
        if (by_handle) { // the operands stay alive until the batch has run; the batch holds its own reference to the result
            Slot& sl = b->slots[slot];
            for (int k = 0; k < 3; k++)
                if (vin[k]) { vin[k]->retain(); sl.vin[k] = vin[k]; }
            vout->retain();
            sl.vout = vout;
            vout->producer = b;
            vout->slot = (uint32_t)slot;
        }
        b->t_last = std::chrono::steady_clock::now();
        if (b->n == 0) b->t0 = b->t_last;
        b->n++;
        if (ticket) *ticket = next_ticket++;
        if (b->deferred) {
            n_deferred_ops++;
            t_last_deferred = b->t_last;
            // (launching what has gathered every N operations, so that the GPU works while a long push goes on, was measured and
            // loses: it cuts the merged conversion batches into pieces — a pushed 32 x 32 multiplication 75-81 ms at 4 096 / 16 384
            // operations per flush, 64 at 65 536, 60-65 with none; launching only the CHEAP batches every 2 048 / 8 192 / 32 768
            // operations: 60-62 ms against 51-54 — the launcher then works the pool's mutex while the pusher pushes)
            if (b->n == b->cap) {
                // full: the next batch of this key is twice as big; this one waits with the rest (launching it now would take every
                // shallower batch with it, half filled: a level of 1 024 gates pushed into a fresh pool is a few launches, once)
                cap_hint[kind] = std::min(batch_cap(op), 2 * b->cap);
                try {
                    deferred_full.push_back(b);
                    deferred.erase(std::make_tuple(heavy(op) ? (int64_t)-1 : b->depth, op, param, b->rank));
                } catch (const std::exception&) {
                    flush_deferred();
                }
            }
        } else if (b->n == b->cap) {
            cap_hint[kind] = std::min(batch_cap(op), 2 * b->cap); // it filled up: the callers can feed a bigger one
            close_batch(lane);
        } else if (by_handle && heavy(op) && ticket && b->n * split >= std::max(heavy_population(b->t_last), n_open.load())) {
            // By handle a caller is back microseconds after its result (nothing to copy out), so the callers need no dealing into
            // groups that meet again: a bootstrap batch simply goes as soon as it holds a quarter of the callers in the pool
            // (all callers count, whatever they are waiting for: the conversions of a circuit's inputs arrive out of its keyswitch
            // batches; or nobody has joined it for max_wait), on the workgroup shape of the whole population — up to four or five
            // batches tile the CUs side by side, each on its set's stream, and whoever comes back meanwhile is the next one.
            // A 32-bit adder's 64 conversions are four launches within the time the callers take to arrive, not four paced
            // quarter batches; 1 024 synchronous callers keep four batches of 256 in flight.  (Operations without a ticket come from
            // a pusher that does not block: its batches close on the quiet time.)
            close_batch(lane);
        } else if (tracks_callers && everybody_is_back(*b)) {
            poke(); // the launcher need not wait for more
        }
        if (by_handle) { // nothing to copy: the slot is ready as it stands
            b->n_ready++;
            if (!b->deferred && (slot == 0 || (b->closed && b->n_ready == b->n))) poke();
        }
        lk.unlock();
        if (!ticket) return SPF_OK;
        // the ticket becomes findable (by any thread) and, for this thread, findable without looking
        bool registered = true;
        {
            TicketShard& sh = shard_of(my_ticket);
            std::lock_guard<PoolMutex> g(sh.mu);
            try {
                sh.map.emplace(my_ticket, std::make_pair(b, slot));
            } catch (const std::exception&) {
                registered = false; // (out of host memory: the ticket can still be waited for by THIS thread, below)
            }
        }
        {
            LastSubmit& ls = last_submit();
            ls.pool_gen = gen; ls.ticket = my_ticket; ls.batch = b; ls.slot = slot;
        }
        (void)registered;
        if (by_handle) return SPF_OK;
        size_t in[3], outsz;
        in_out_sizes(op, in, outsz);
        const Staging& s = sets[b->set]; // (a set's buffers are stable while its batch is open)
        // the caller's own bytes, by the caller's own thread, straight into pinned memory
        for (int k = 0; k < 3; k++)
            if (in[k]) std::memcpy(static_cast<uint8_t*>(s.h_in[k]) + slot * in[k], src[k], in[k]);
        lk.lock();
        b->n_ready++;
        // the launcher sleeps until something changes for it: a batch got its first member (its deadline starts), or a
        // closed batch just got its last input
        if (slot == 0 || (b->closed && b->n_ready == b->n)) poke();
        return SPF_OK;
    }

    // `mu` held.  Synchronous callers come back: the threads of the group's batch that just finished collect their outputs and
    // submit again.  Once every one of them is in the filling batch nobody is left to wait for.  (Threads the group has not seen
    // in its previous batch do not count: for callers arriving for the first time the coalescing window is `max_wait`.)
    bool everybody_is_back(const Batch& b) const
    {
        const std::vector<uintptr_t>& last = last_members[b.lane];
        return b.n > 0 && !last.empty() && b.n_returning >= last.size();
    }

    // `mu` held.  The callers of the bootstrap kinds blocked in the pool right now (open tickets) stand for the population that
    // keeps coming back.  (Tickets of the cheap kinds do not count: a thousand open CMUX gates say nothing about how many
    // bootstraps the next batches will hold.)
    size_t population() const { return heavy_open; }
    // ... and, for the quarter rule of the bootstrap batches by handle, the most callers seen at once lately: while a crowd of
    // callers is still arriving (the conversions of a circuit becoming ready one after the other) the open tickets undercount it
    size_t heavy_peak = 0;
    std::chrono::steady_clock::time_point heavy_peak_at{};
    size_t heavy_population(std::chrono::steady_clock::time_point now)
    {
        const size_t open = heavy_open.load();
        if (open >= heavy_peak || now - heavy_peak_at > std::chrono::milliseconds(100)) {
            heavy_peak = open;
            heavy_peak_at = now;
        }
        return heavy_peak;
    }
    // Caller groups in use.  Measured (tools/pool_bench.py, fraction of the device-resident rate; profiles/r05_pool.md):
    //   callers      32    64    128   256   384   512   768   1024  1536  2048
    //   2 groups    0.88  0.91  0.77  0.71  0.81  0.67  0.77  0.73  0.88  0.53
    //   3 groups    0.96  0.94  0.88  0.76  0.84  0.77  0.81  0.75  0.87  0.58
    //   4 groups    0.93  0.88  0.89  0.79  0.85  0.79  0.82  0.69  0.85  0.65     (6: 0.79 / 0.67 / 0.63, 8: 0.68 / 0.68 / 0.60 at 64 / 256 / 1024)
    // Four groups, except around the population that exactly fills the chip at four ciphertexts per CU (1 024 on 256 CUs): there
    // the bootstraps of four resident batches subscribe every CU and each group's keyswitch, trace and copy kernels queue behind
    // them — three groups leave the fourth quarter of the callers in their host phase.
    // (hysteresis: around 3.5 / 6 times the CU count the population oscillates as callers pass through wait and submit; the count
    // only changes when the other value has been wanted for 50 ms, so that the callers stay in their groups — ADVICE r05)
    mutable size_t groups_cur = 4;
    mutable std::chrono::steady_clock::time_point groups_other_since{};
    size_t groups_now() const
    {
        if (groups) return groups;
        const size_t pop = population(), n_cu = (size_t)ctx->n_cu;
        const size_t want = (2 * pop > 7 * n_cu && pop <= 6 * n_cu) ? 3 : 4;
        if (want == groups_cur) {
            groups_other_since = {};
            return groups_cur;
        }
        const auto now = std::chrono::steady_clock::now();
        if (groups_other_since == std::chrono::steady_clock::time_point{}) groups_other_since = now;
        else if (now - groups_other_since > std::chrono::milliseconds(50)) {
            groups_cur = want;
            groups_other_since = {};
        }
        return groups_cur;
    }
    // ciphertexts per workgroup the bootstrap of a batch should use at least: the shape the whole population would get in one
    // launch, so that the resident batches tile the CUs (one batch of a quarter of 1 024 callers takes 64 CUs, not 256)
    int per_wg_hint() const
    {
        const size_t pop = population(), n_cu = (size_t)ctx->n_cu;
        return pop <= n_cu ? 1 : (pop <= 2 * n_cu ? 2 : 4);
    }
    size_t running() const // host-pointer bootstrap batches whose kernels are on the GPU
    {
        size_t r = 0;
        for (auto& f : in_flight) r += (f->kernels_done || f->by_handle || !spf_pool_impl::heavy(f->op)) ? 0 : 1;
        return r;
    }

    // `mu` held: the batch of `lane` takes no more members; the launcher enqueues it once every member's input is in
    void close_batch(int lane)
    {
        std::shared_ptr<Batch> b = filling[lane];
        if (!b) return;
        b->closed = true;
        b->t_close = std::chrono::steady_clock::now();
        filling[lane].reset();
        // the group's next batch is complete when THESE callers are back (and it does not close on the timer before this batch is
        // done: a member that missed this batch waits for its group instead of becoming a group of one that runs out of phase
        // with it for ever — measured: single-ciphertext launches of 3.7 ms each, alternating with the group's)
        try {
            std::vector<uintptr_t>& last = last_members[lane];
            last.clear();
            for (size_t i = 0; i < b->n; i++) last.push_back(b->slots[i].who);
            std::sort(last.begin(), last.end());
        } catch (const std::exception&) {
            last_members[lane].clear(); // (out of memory: the next batch closes on the timer)
        }
        outstanding[lane]++;
        closing.push_back(b);
        poke();
    }

    // `mu` held: every deferred batch is closed, shallowest first (so that `closing`, which the launcher takes in order, holds every
    // batch behind the ones its operands come from); a batch of the ordinary lanes that an operand comes from and that is still
    // filling is closed ahead of its first user.  Called when a deferred batch is full, when somebody waits for a result that
    // sits in one (spf_pool_wait, spf_value_wait), when nothing has been deferred for the quiet time (launcher), at destroy.
    void flush_deferred()
    {
        if (deferred.empty() && deferred_full.empty()) return;
        const auto now = std::chrono::steady_clock::now();
        auto close_one = [&](const std::shared_ptr<Batch>& b) {
            for (auto& d : b->deps)
                if (!d->deferred && !d->closed && filling[d->lane] == d) close_batch(d->lane);
            b->closed = true;
            b->t_close = now;
            outstanding[b->lane]++;
            closing.push_back(b);
        };
        // (the map is in depth order; the full ones are merged in by depth: they were filled in any order)
        std::stable_sort(deferred_full.begin(), deferred_full.end(), [](const std::shared_ptr<Batch>& x, const std::shared_ptr<Batch>& y) { return x->depth < y->depth; });
        size_t f = 0;
        for (auto& kv : deferred) {
            while (f < deferred_full.size() && deferred_full[f]->depth <= kv.second->depth) close_one(deferred_full[f++]);
            close_one(kv.second);
        }
        while (f < deferred_full.size()) close_one(deferred_full[f++]);
        deferred.clear();
        deferred_full.clear();
        poke();
    }

    // `mu` held.  Deliver the uncollected outputs of done batches that have waited longer than the grace period.
    void reclaim(std::unique_lock<Mutex>& lk)
    {
        const auto now = std::chrono::steady_clock::now();
        for (size_t bi = 0; bi < collecting.size(); bi++) {
            std::shared_ptr<Batch> b = collecting[bi];
            if (now - b->t_done < grace) continue;
            for (size_t i = 0; i < b->n; i++) {
                spf_pool_impl::Slot& sl = b->slots[i];
                if (sl.delivered || sl.claim != 0) continue; // (a claimed ticket's waiter is copying right now, or has collected it)
                if (b->st == SPF_OK) {
                    sl.delivered = 1; // (a wait() for this ticket that arrives during the copy must not return before it ends:
                    lk.unlock();      //  the caller may free or read `out` the moment wait returns, spf_hip.h)
                    deliver(*b, i);
                    lk.lock();
                }
                sl.delivered = 2;
                n_reclaimed++;
                cv_deliver.notify_all();
                collected_one(b); // the ticket stays open (its status is still to be fetched), its bytes are out
                if (!sets[b->set].busy) return;
            }
        }
    }

    spf_status wait(uint64_t ticket)
    {
        using namespace spf_pool_impl;
        blocked++;
        struct Leave { spf_pool* p; ~Leave() { if (--p->blocked == 0 && p->stop) { std::lock_guard<Mutex> g(p->mu); p->cv_idle.notify_all(); } } } leave{this};
        std::shared_ptr<Batch> b;
        size_t slot = 0;
        {
            LastSubmit& ls = last_submit();
            if (ls.pool_gen == gen && ls.ticket == ticket && ls.batch) { // the synchronous caller: its own last submit
                b = std::move(ls.batch);
                slot = ls.slot;
                ls.ticket = 0;
            } else {
                TicketShard& sh = shard_of(ticket);
                std::lock_guard<PoolMutex> g(sh.mu);
                auto it = sh.map.find(ticket);
                if (it == sh.map.end()) return SPF_ERR_INVALID_ARGUMENT; // unknown or already collected: an error, not a hang
                b = it->second.first;
                slot = it->second.second;
            }
        }
        // a ticket can be waited for exactly once: a second waiter (or a wait after it was collected) is an error
        Slot& sl = b->slots[slot];
        if (b->by_handle) {
            uint8_t zero = 0;
            if (!__atomic_compare_exchange_n(&sl.claim, &zero, (uint8_t)1, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) return SPF_ERR_INVALID_ARGUMENT;
        } else {
            std::lock_guard<Mutex> lk(mu); // (reclaim() decides under the mutex whether a ticket has a waiter)
            if (sl.claim != 0) return SPF_ERR_INVALID_ARGUMENT;
            sl.claim = 1;
        }
        // sleep on the batch's own word: no pool-wide condition variable, no mutex on the way out
        const bool is_heavy = heavy(b->op);
        if (b->deferred) { // somebody wants a result of the deferred table: everything pushed so far goes
            std::lock_guard<Mutex> lk(mu);
            if (!b->closed) flush_deferred();
        }
        if (b->by_handle) {
            const size_t g = slot / Batch::kTreeGroup;
            std::atomic<uint32_t>& word = b->gword[g];
            if (!is_heavy && spin_us > 0) {
                // a cheap operation is back in tens of microseconds: look for a moment before going to sleep (a sleeping thread
                // costs its waker a system call and itself a wake-up, ~40 us end to end on the bench's host)
                const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us);
                for (int i = 0; word.load(std::memory_order_acquire) == 0; i++) {
                    for (int k = 0; k < 32; k++) __builtin_ia32_pause();
                    if ((i & 7) == 7) {
                        if (std::chrono::steady_clock::now() >= until) break;
                        (void)sched_yield(); // (more callers than CPUs: the ones that still have to submit go first)
                    }
                }
            }
            if (word.load(std::memory_order_acquire) == 0) b->sleep_on_group(g);
            const size_t child = Batch::kTreeGroup * g + 1 + slot % Batch::kTreeGroup; // this waiter's share of the waking
            if (child < b->n_groups()) b->wake_group(child);
        } else {
            std::atomic<uint32_t>& word = b->chunk_word[Batch::word_of(slot)];
            while (word.load(std::memory_order_acquire) == 0) futex_wait(&word, 0);
        }
        const spf_status st = b->st;
        if (!b->by_handle) {
            // (`delivered` leaves 0 only for an unclaimed ticket, under the mutex this thread's claim went through: what this
            // thread reads here without the lock is either 0 for good, or the 1 / 2 that reclaim() set before the claim)
            if (st == SPF_OK && sl.delivered == 0) deliver(*b, slot); // this caller's output, by this caller's thread
            std::unique_lock<Mutex> lk(mu);
            cv_deliver.wait(lk, [&] { return sl.delivered != 1; }); // a delivery on this ticket's behalf is still copying
            if (sl.delivered == 0) {
                sl.delivered = 2;
                collected_one(b);
            }
            sl.claim = 2;
            auto it = open_by_thread.find(sl.who);
            if (it != open_by_thread.end() && --it->second <= 0) open_by_thread.erase(it);
        } else {
            __atomic_store_n(&sl.claim, (uint8_t)2, __ATOMIC_RELEASE);
        }
        {
            TicketShard& sh = shard_of(ticket);
            std::lock_guard<PoolMutex> g(sh.mu);
            sh.map.erase(ticket);
        }
        if (is_heavy) {
            size_t h = heavy_open.load();
            while (h && !heavy_open.compare_exchange_weak(h, h - 1)) {}
        }
        n_open--;
        if (space_waiters.load()) {
            std::lock_guard<Mutex> lk(mu);
            cv_space.notify_all();
        }
        return st;
    }

    // spf_value_wait: until the value's batch has been handed back (any number of threads, any number of times; no ticket involved)
    spf_status wait_value(const spf_value* v)
    {
        using namespace spf_pool_impl;
        if (v->state.load(std::memory_order_acquire) == spf_value_impl::PENDING) {
            blocked++;
            struct Leave { spf_pool* p; ~Leave() { if (--p->blocked == 0 && p->stop) { std::lock_guard<Mutex> g(p->mu); p->cv_idle.notify_all(); } } } leave{this};
            std::shared_ptr<Batch> b;
            {
                std::lock_guard<Mutex> lk(mu);
                b = v->producer;
                if (b && b->deferred && !b->closed) flush_deferred();
                // (an operation with valid operands sits in an ordinary lane: whoever waits for its VALUE wants it now — the lane is
                // closed instead of sitting out the quiet time; spf_pool_wait, the blocking callers' way, leaves it to gather)
                else if (b && !b->deferred && !b->closed && filling[b->lane] == b) close_batch(b->lane);
            }
            if (b) {
                const size_t g = v->slot / Batch::kTreeGroup;
                if (b->gword[g].load(std::memory_order_acquire) == 0) b->sleep_on_group(g);
            }
        }
        const int st = v->state.load(std::memory_order_acquire);
        return st == spf_value_impl::READY ? SPF_OK : (st == spf_value_impl::FAILED ? SPF_ERR_HIP : SPF_ERR_INVALID_ARGUMENT);
    }

    // ---- launcher: closes batches and enqueues them
    // the kernels of one batch: inputs d[0..2] (contiguous rows), output d_out, on the set's stream and intermediates
    spf_status run_kernels(const Batch& b, const spf_pool_impl::Staging& s, size_t B, void* const d[3], void* d_out)
    {
        using namespace spf_pool_impl;
        hipStream_t sk = stream_of(b);
        Scratch* scr = const_cast<Scratch*>(&s.scr);
        spf_status st;
        switch (b.op) {
        case OP_KEYSWITCH:
            return pool_keyswitch(ctx, sk, B, (const uint64_t*)d[0], (uint64_t*)d_out, scr);
        case OP_CBS:
            return pool_circuit_bootstrap(ctx, sk, B, (const uint64_t*)d[0], (double*)d_out, scr, b.per_wg);
        case OP_GATE_CBS: // FheOp::KeyswitchL1toL0 -> FheOp::CircuitBootstrap, the level-0 LWE stays in HBM
            st = pool_keyswitch(ctx, sk, B, (const uint64_t*)d[0], (uint64_t*)s.d_mid, scr);
            if (st == SPF_OK) st = pool_circuit_bootstrap(ctx, sk, B, (const uint64_t*)s.d_mid, (double*)d_out, scr, b.per_wg);
            return st;
        case OP_SAMPLE_EXTRACT:
            return spf_sample_extract_l1_dev(ctx, sk, B, (const uint64_t*)d[0], (size_t)b.param, (uint64_t*)d_out);
        case OP_NOT:
            return spf_glwe_not_dev(ctx, sk, B, (const uint64_t*)d[0], (uint64_t*)d_out);
        case OP_GLWE_ADD:
            return spf_glwe_xor_dev(ctx, sk, B, (const uint64_t*)d[0], (const uint64_t*)d[1], (uint64_t*)d_out);
        case OP_MUL_XN:
            return spf_glwe_mul_xn_dev(ctx, sk, B, (const uint64_t*)d[0], (size_t)b.param, (uint64_t*)d_out);
        case OP_MULTIPLY_GGSW_GLWE:
            return spf_multiply_glwe_ggsw_dev(ctx, sk, B, (const uint64_t*)d[1], (const double*)d[0], (uint64_t*)d_out);
        case OP_GLEV_CMUX:
            return spf_glev_cmux_dev(ctx, sk, B, (const double*)d[0], (const uint64_t*)d[1], (const uint64_t*)d[2], (uint64_t*)d_out);
        case OP_SCHEME_SWITCH:
            return spf_scheme_switch_dev(ctx, sk, B, (const uint64_t*)d[0], (double*)d_out);
        default:
            return spf_cmux_dev(ctx, sk, B, (const double*)d[0], (const uint64_t*)d[1], (const uint64_t*)d[2], (uint64_t*)d_out);
        }
    }

    // A batch by handle: no copies.  The outputs are one block of the arena; the operands are read where they are (CMUX family)
    // or packed into the set's device rows by gather_rows_kernel — unless they already lie consecutively.
    spf_status enqueue_v(Batch& b)
    {
        using namespace spf_pool_impl;
        size_t in[3], out;
        in_out_sizes(b.op, in, out);
        const Staging& s = staging_of(b);
        const size_t B = b.n;
        if (hipSetDevice(ctx->device) != hipSuccess) return SPF_ERR_HIP;
        b.out_blk = spf_value_impl::Block::make(arena, B * out);
        if (!b.out_blk) return SPF_ERR_HIP;
        char* d_out = static_cast<char*>(b.out_blk->p);
        // (a bootstrap batch is milliseconds on the GPU: its completer sleeps in the driver — hipEventBlockingSync — instead of
        // polling the event from a CPU; the cheap batches are polled by the launcher itself.  SPF_POOL_EVENT_FLAGS overrides.)
        static const unsigned heavy_flags = [] { const char* e = getenv("SPF_POOL_EVENT_FLAGS"); return e ? (unsigned)atoi(e) : (unsigned)hipEventBlockingSync; }();
        if (hipEventCreateWithFlags(&b.ev_k, heavy(b.op) ? heavy_flags : hipEventDefault) != hipSuccess) return SPF_ERR_HIP;
        hipStream_t sk = stream_of(b);
        spf_status st = SPF_OK;
        if (scattered_cmux(b.op)) {
            // units {selector, low (taken when the selector is 0; null = the zero ciphertext), high, out}; units of one selector next
            // to each other (neighbouring workgroups then hit the same 256 KiB in L2: spf_graph.hpp)
            const size_t per = b.op == OP_GLEV_CMUX ? prm.cbs_radix_count : 1, gw = glwe_bytes();
            struct Unit { void* p[4]; };
            Unit* units = reinterpret_cast<Unit*>(s.h_ptrs);
            size_t n_units = 0;
            for (size_t i = 0; i < B; i++) {
                const Slot& sl = b.slots[i];
                for (size_t j = 0; j < per; j++) {
                    Unit u;
                    u.p[0] = sl.vin[0]->ptr();
                    if (b.op == OP_MULTIPLY_GGSW_GLWE) {
                        u.p[1] = nullptr;
                        u.p[2] = sl.vin[1]->ptr();
                    } else {
                        u.p[1] = static_cast<char*>(sl.vin[1]->ptr()) + j * gw;
                        u.p[2] = static_cast<char*>(sl.vin[2]->ptr()) + j * gw;
                    }
                    u.p[3] = d_out + i * out + j * gw;
                    units[n_units++] = u;
                }
            }
            std::stable_sort(units, units + n_units, [](const Unit& x, const Unit& y) { return x.p[0] < y.p[0]; });
            st = spf_cmux_scattered_dev(ctx, sk, n_units, (const void* const*)s.h_ptrs);
        } else {
            void* d[3] = {nullptr, nullptr, nullptr};
            for (int k = 0; k < 3 && st == SPF_OK; k++) {
                if (!in[k]) continue;
                char* first = static_cast<char*>(b.slots[0].vin[k]->ptr());
                bool contiguous = true;
                for (size_t i = 1; i < B && contiguous; i++) contiguous = b.slots[i].vin[k]->ptr() == first + i * in[k];
                if (contiguous) { d[k] = first; continue; }
                void** row = s.h_ptrs + (size_t)k * b.cap;
                for (size_t i = 0; i < B; i++) row[i] = b.slots[i].vin[k]->ptr();
                st = spf_gather_rows_dev(ctx, sk, B, in[k] / 8, (const uint64_t* const*)row, (uint64_t*)s.d_in[k]);
                d[k] = s.d_in[k];
            }
            if (st == SPF_OK) st = run_kernels(b, s, B, d, d_out);
        }
        if (st != SPF_OK) return st;
        if (hipEventRecord(b.ev_k, sk) != hipSuccess) return SPF_ERR_HIP;
        for (size_t i = 0; i < B; i++) { // (published to the callers by the completer: READY is set behind the event)
            b.slots[i].vout->blk = b.out_blk;
            b.slots[i].vout->off = i * out;
        }
        return SPF_OK;
    }

    spf_status enqueue(Batch& b)
    {
        using namespace spf_pool_impl;
        if (b.by_handle) return enqueue_v(b);
        size_t in[3], out;
        in_out_sizes(b.op, in, out);
        const Staging& s = sets[b.set];
        const size_t B = b.n;
        if (hipSetDevice(ctx->device) != hipSuccess) return SPF_ERR_HIP;
        // (timing events on purpose: with hipEventDisableTiming the same run gave 32.8 k instead of 46.2 k operations per second at 256
        // callers — hipEventSynchronize on such an event returned late, by about the batch's remaining work in its stream — r05q)
        constexpr unsigned kEvFlags = hipEventDefault;
        {
            const size_t groups = (B + Batch::kWordSlots - 1) / Batch::kWordSlots;
            b.n_chunks = (int)std::min<size_t>(groups, Batch::kMaxChunks);
            b.chunk_slots = (groups + b.n_chunks - 1) / b.n_chunks * Batch::kWordSlots;
            b.n_chunks = (int)((B + b.chunk_slots - 1) / b.chunk_slots);
        }
        if (hipEventCreateWithFlags(&b.ev_in, kEvFlags) != hipSuccess || hipEventCreateWithFlags(&b.ev_k, kEvFlags) != hipSuccess)
            return SPF_ERR_HIP;
        for (int i = 0; i < b.n_chunks; i++)
            if (hipEventCreateWithFlags(&b.ev_chunk[i], kEvFlags) != hipSuccess) return SPF_ERR_HIP;
        // Everything of a batch — copy in, kernels, copies out — goes on the SET's stream, in order, and nothing waits for an event
        // of another stream: batches of different sets run side by side, a batch's copies run under the other batches' kernels
        // anyway, and no barrier packet of one batch sits in a hardware queue in front of another batch's kernels (r05: with
        // separate copy streams tied to the kernel streams by events, streams that shared a hardware queue with a copy stream
        // stood behind its "wait for batch A's kernels" — the resident batches ran one after the other).
        hipStream_t sk = s.sk;
        for (int k = 0; k < 3; k++) // (a kernel that reads the pinned buffer, not an SDMA copy: spf_kernels.hpp, copy_words_kernel)
            if (in[k] && pool_copy_in(ctx, sk, s.h_in[k], s.d_in[k], B * in[k]) != SPF_OK) return SPF_ERR_HIP;
        if (hipEventRecord(b.ev_in, sk) != hipSuccess) return SPF_ERR_HIP;
        const spf_status st = run_kernels(b, s, B, s.d_in, s.d_out);
        if (st != SPF_OK) return st;
        if (hipEventRecord(b.ev_k, sk) != hipSuccess) return SPF_ERR_HIP;
        hipStream_t so = sk;
        for (int i = 0; i < b.n_chunks; i++) {
            const size_t first = (size_t)i * b.chunk_slots;
            const size_t count = std::min(b.chunk_slots, B - first);
            if (hipMemcpyAsync(static_cast<uint8_t*>(s.h_out) + first * out, static_cast<const uint8_t*>(s.d_out) + first * out, count * out,
                               hipMemcpyDeviceToHost, so) != hipSuccess)
                return SPF_ERR_HIP;
            if (hipEventRecord(b.ev_chunk[i], so) != hipSuccess) return SPF_ERR_HIP;
        }
        return SPF_OK;
    }

    // A batch by handle has completed (or failed): its results become visible, its operands are let go, its staging set is free,
    // its waiters are woken.  Called without `mu` by whoever saw the batch's event: its set's completer, or — the cheap kinds —
    // the launcher itself.
    void finish_handle_batch(const std::shared_ptr<Batch>& b)
    {
        using namespace spf_pool_impl;
        // a batch whose operands came from a batch that failed has computed on nothing: it fails with that status (the stream order
        // puts the producers' completion before this batch's; `deps` is this thread's from here on: the launcher is done with it)
        for (auto& d : b->deps)
            if (b->st == SPF_OK && d->st != SPF_OK) b->st = d->st.load();
        const int state = b->st == SPF_OK ? spf_value_impl::READY : spf_value_impl::FAILED;
        size_t no_ticket_heavy = 0;
        for (size_t i = 0; i < b->n; i++) { // (the state first: a submit that still sees PENDING under the mutex finds `producer` set)
            b->slots[i].vout->state.store(state, std::memory_order_release);
            no_ticket_heavy += (b->slots[i].ticket == 0 && heavy(b->op)) ? 1 : 0;
        }
        {
            std::lock_guard<Mutex> lk(mu);
            for (size_t i = 0; i < b->n; i++) b->slots[i].vout->producer.reset();
        }
        for (size_t i = 0; i < b->n; i++) { // (outside the lock: the last reference to a value gives its block back to the arena)
            Slot& sl = b->slots[i];
            sl.vout->release();
            sl.vout = nullptr;
            for (spf_value*& v : sl.vin)
                if (v) { v->release(); v = nullptr; }
        }
        b->out_blk.reset();
        b->deps.clear();
        for (size_t h = heavy_open.load(); no_ticket_heavy && h;) // (operations without a ticket leave the population here)
            if (heavy_open.compare_exchange_weak(h, h - std::min(h, no_ticket_heavy))) break;
        {
            std::lock_guard<Mutex> lk(mu);
            last_gpu_span[b->op] = b->t_sync - b->t_enq;
            b->kernels_done = true;
            for (auto it = in_flight.begin(); it != in_flight.end(); ++it)
                if (it->get() == b.get()) { in_flight.erase(it); break; }
            b->done = true;
            b->t_done = std::chrono::steady_clock::now();
            last_done = b->t_done;
#ifdef SPF_POOL_TRACE
            {
                auto us = [](auto d) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(d).count(); };
                static const auto epoch = std::chrono::steady_clock::now();
                fprintf(stderr, "[pool] batch op %d (by handle) n %zu: first member at %ld us, filled %ld us, closed->ready %ld us, enqueue %ld us, enqueued->event %ld us, event->marked %ld us\n", b->op, b->n,
                        us(b->t0 - epoch), us(b->t_close - b->t0), us(b->t_ready - b->t_close), us(b->t_enq - b->t_ready), us(b->t_sync - b->t_enq), us(b->t_done - b->t_sync));
            }
#endif
            staging_of(*b).busy = false; // nothing to collect: the staging set is free again
            b->destroy_events();
            if (set_waiters) cv_set.notify_all();
            n_handle_launches++;
            n_handle_ops += b->n;
            outstanding[b->lane]--;
            n_launches++;
            n_ops += b->n;
            poke(); // the launcher closes the batch that filled meanwhile
        }
        b->wake_tree();
    }

    void launch_loop()
    {
        using namespace spf_pool_impl;
        using clock = std::chrono::steady_clock;
        (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0); // the quiet times below are tens of microseconds: the default slack is 50
        std::unique_lock<Mutex> lk(mu);
        auto hot_until = clock::now();
#ifdef SPF_POOL_TRACE
        struct Tally { double t_setdev = 0; long n_deps = 0, n_waits = 0; double t_deps = 0, t_prep = 0, t_enq2 = 0, t_relock = 0; double poll = 0, enq = 0, spin = 0, sleep = 0; long n_poll = 0, n_enq = 0, n_spin = 0, n_sleep = 0, n_noset = 0; ~Tally() { fprintf(stderr, "[pool] launcher: polling %.1f ms in %ld passes, enqueue %.1f ms for %ld batches, spinning %.1f ms (%ld), asleep %.1f ms (%ld), passes without a set for a deferred batch %ld; inside the enqueue block: deps %.1f ms (hipSetDevice %.1f ms; %ld dependencies looked at, %ld event waits), prepare %.1f, enqueue %.1f, relock %.1f\n", poll, n_poll, enq, n_enq, spin, n_spin, sleep, n_sleep, n_noset, t_deps, t_setdev, n_deps, n_waits, t_prep, t_enq2, t_relock); } } tally;
        auto ms_since = [](clock::time_point t) { return std::chrono::duration<double, std::milli>(clock::now() - t).count(); };
#endif
        for (;;) {
            auto now = clock::now();
            // 0. the cheap batches by handle this thread enqueued: it also completes them (a 15 us CMUX level handed to a
            // completer thread costs that thread's wake-up — more than the kernel — before anybody looks at the event)
            if (!polling.empty()) {
#ifdef SPF_POOL_TRACE
                const auto t_poll = clock::now();
                struct P { Tally& t; clock::time_point t0; ~P() { t.poll += std::chrono::duration<double, std::milli>(clock::now() - t0).count(); t.n_poll++; } } p_{tally, t_poll};
#endif
                lk.unlock();
                // (streams are in order: behind a batch that has not finished nothing of its stream has — one query per stream and
                // pass, not one per batch: with dozens of deferred batches parked behind a bootstrap the queries were 200 us a pass)
                hipStream_t waiting[1 + spf_pool_impl::kSets];
                int n_waiting = 0;
                for (size_t i = 0; i < polling.size();) {
                    Batch& b = *polling[i];
                    const hipStream_t bs = stream_of(b);
                    bool behind = false;
                    for (int w = 0; w < n_waiting; w++) behind = behind || waiting[w] == bs;
                    if (behind) { i++; continue; }
                    const hipError_t e = hipEventQuery(b.ev_k);
                    if (e == hipErrorNotReady) {
                        if (n_waiting < (int)(sizeof(waiting) / sizeof(waiting[0]))) waiting[n_waiting++] = bs;
                        i++;
                        continue;
                    }
                    if (e != hipSuccess) {
                        (void)hipGetLastError();
                        b.st = SPF_ERR_HIP;
                        (void)hipStreamSynchronize(stream_of(b));
                    }
                    b.t_sync = clock::now();
                    std::shared_ptr<Batch> done = polling[i];
                    polling.erase(polling.begin() + (long)i);
                    finish_handle_batch(done);
                }
                lk.lock();
                now = clock::now();
                hot_until = now + std::chrono::microseconds(hot_us);
            }
            auto wake = clock::time_point::max();
            // 1. a closed batch: enqueue it as soon as its last members have copied their inputs in (the first one that can go:
            // a batch whose members are still copying does not hold up the ones behind it)
            std::shared_ptr<Batch> go;
            bool no_set_for_deferred = false, no_table_set = false;
            for (auto it = closing.begin(); it != closing.end(); ++it) {
                Batch& b = **it;
                if (b.n_ready < b.n) continue; // (submit wakes this thread when the last input is in)
                if (b.deferred && b.set < 0) {
                    // A deferred batch goes when every batch its operands come from has been ENQUEUED (its own stream is then in
                    // order behind them, or waits for their events) and the bootstrap batches among them have FINISHED (see s_def:
                    // nothing is enqueued behind a running bootstrap batch).  `closing` holds the deferred batches shallowest first,
                    // so what is skipped here is looked at again before anything that depends on it.
                    bool ready = true;
                    for (auto& d : b.deps) {
                        if (d->st != SPF_OK || d->done) continue; // (failed: this batch is launched to fail with it)
                        if (!d->ev_k || heavy(d->op)) { ready = false; break; }
                    }
                    if (!ready) continue;
                    // ... and takes its set now (bootstrap: stream, pointer table, intermediates; cheap: a table set); when every
                    // set of its kind is held it waits for a batch to be handed back (which pokes this thread)
                    const bool table = table_batch(b);
                    if (table ? no_table_set : no_set_for_deferred) continue;
                    if (!table && max_def_heavy > 0) { // (see max_def_heavy)
                        int flying_heavy = 0;
                        for (auto& f : in_flight) flying_heavy += (f->deferred && heavy(f->op)) ? 1 : 0;
                        if (flying_heavy >= max_def_heavy) continue;
                    }
                    int set = -1;
                    if (table) {
                        for (int i = 0; i < n_table_sets; i++) if (!tsets[i].busy) { set = i; break; }
                    } else {
                        for (int i = 0; i < n_sets; i++) if (!sets[i].busy) { set = i; break; }
                    }
                    if (set < 0) { (table ? no_table_set : no_set_for_deferred) = true; continue; }
                    b.set = set;
                    staging_of(b).busy = true;
                }
                // Pacing of the bootstrap kinds: resident batches that start together also finish together — their copies out
                // queue behind each other and their callers come back in one crowd, i.e. they behave as ONE big batch and the GPU
                // idles through the common turn-around.  A batch therefore starts no sooner than a 1 / groups share of a batch's
                // time on the GPU after the previous one: once spread out, the groups keep their phases (each comes back one cycle
                // later), and while one copies out and resubmits the others compute.
                if (!stop && heavy(b.op) && !b.by_handle && running() > 0 && pace_div >= 0) {
                    const auto due = last_enq + std::min<clock::duration>(last_gpu_span[b.op] / (pace_div > 0 ? pace_div : (int)groups_now()), std::chrono::milliseconds(5));
                    if (now < due) { wake = std::min(wake, due); continue; }
                }
                // (By handle no pacing: quarter batches of 1 024 callers do start and finish in convoys — 0.77-0.81 of the device-resident
                // rate — but spacing them by a quarter of a batch's GPU time halved the rate, 34-37 k against 64-68 k circuit
                // bootstraps per second: a caller parked in a closed batch is a caller not computing.  profiles/r06_values.md)
                go = *it;
                closing.erase(it);
                break;
            }
#ifdef SPF_POOL_TRACE
            if (no_set_for_deferred) tally.n_noset++;
#endif
            if (go) {
#ifdef SPF_POOL_TRACE
                struct E { Tally& t; clock::time_point t0; ~E() { t.enq += std::chrono::duration<double, std::milli>(clock::now() - t0).count(); t.n_enq++; } } e_{tally, clock::now()};
#endif
                std::shared_ptr<Batch> b = go;
                b->t_ready = clock::now();
                if (heavy(b->op)) last_enq = b->t_ready;
                b->per_wg = per_wg_hint();
                if (heavy(b->op)) { // (the shape launch_blind_rotate will pick from the batch size and the hint)
                    const size_t n_cu = (size_t)ctx->n_cu;
                    n_shape[(b->n <= n_cu && b->per_wg <= 1) ? 0 : ((b->n <= 2 * n_cu && b->per_wg <= 2) ? 1 : 2)]++;
                }
                // the batches its operands come from: enqueued before it (they were closed before it) — on the same in-order
                // stream (deferred cheap batches), or on a set's stream (the ordinary lanes, deferred bootstraps): this batch's
                // stream waits for THEIR events; one that has failed fails it
                spf_status st = SPF_OK;
                if (!b->deps.empty()) {
#ifdef SPF_POOL_TRACE
                    const auto t_sd = clock::now();
#endif
                    if (hipSetDevice(ctx->device) != hipSuccess) st = SPF_ERR_HIP;
#ifdef SPF_POOL_TRACE
                    tally.t_setdev += ms_since(t_sd);
                    tally.n_deps += (long)b->deps.size();
#endif
                    const hipStream_t mine = stream_of(*b);
                    for (auto& d : b->deps) {
                        if (st != SPF_OK) break;
                        if (d->st != SPF_OK) st = d->st.load();
                        else if (d->done || stream_of(*d) == mine) continue; // (ahead of this batch on the same in-order stream)
                        else {
#ifdef SPF_POOL_TRACE
                            tally.n_waits++;
                            const auto t_w = clock::now();
#endif
                            if (!d->ev_k || hipStreamWaitEvent(mine, d->ev_k, 0) != hipSuccess) st = SPF_ERR_HIP;
#ifdef SPF_POOL_TRACE
                            if (ms_since(t_w) > 0.1) fprintf(stderr, "[pool] launcher: hipStreamWaitEvent took %.0f us (batch op %d waits for op %d deferred %d)\n", ms_since(t_w) * 1e3, b->op, d->op, (int)d->deferred);
#endif
                        }
                    }
                }
#ifdef SPF_POOL_TRACE
                const auto t_a = clock::now();
#endif
                lk.unlock();
#ifdef SPF_POOL_TRACE
                auto t_b = clock::now(), t_c = t_b;
#endif
                try {
                    if (st == SPF_OK && b->deferred && !prepare_set(staging_of(*b), b->op, b->cap, true, table_batch(*b))) st = SPF_ERR_HIP;
#ifdef SPF_POOL_TRACE
                    t_c = clock::now();
#endif
                    if (st == SPF_OK) st = enqueue(*b);
                } catch (const std::exception&) {
                    st = SPF_ERR_HIP;
                }
#ifdef SPF_POOL_TRACE
                const auto t_d = clock::now();
#endif
                lk.lock();
#ifdef SPF_POOL_TRACE
                tally.t_deps += std::chrono::duration<double, std::milli>(t_a - b->t_ready).count();
                tally.t_prep += std::chrono::duration<double, std::milli>(t_c - t_b).count();
                tally.t_enq2 += std::chrono::duration<double, std::milli>(t_d - t_c).count();
                tally.t_relock += std::chrono::duration<double, std::milli>(clock::now() - t_d).count();
                if (ms_since(t_a) > 0.5)
                    fprintf(stderr, "[pool] launcher: op %d n %zu deferred %d: deps %.0f us, prepare %.0f us, enqueue %.0f us, relock %.0f us\n", b->op, b->n, (int)b->deferred,
                            std::chrono::duration<double, std::micro>(t_a - b->t_ready).count(), std::chrono::duration<double, std::micro>(t_c - t_b).count(),
                            std::chrono::duration<double, std::micro>(t_d - t_c).count(), std::chrono::duration<double, std::micro>(clock::now() - t_d).count());
#endif
                b->st = st;
                b->t_enq = clock::now();
                in_flight.push_back(b);
                bool mine = false;
                if (b->by_handle && !heavy(b->op) && st == SPF_OK && (hot_us > 0 || table_batch(*b))) {
                    try {
                        polling.push_back(b);
                        mine = true;
                    } catch (const std::exception&) {
                    }
                }
                if (!mine && table_batch(*b)) {
                    // (no completer thread serves the table sets: a batch that failed before its event was recorded — or that the
                    // polling list has no room for — is handed back here, once whatever was enqueued for it has drained)
                    lk.unlock();
                    if (st != SPF_OK || hipEventSynchronize(b->ev_k) != hipSuccess) {
                        if (b->st == SPF_OK) b->st = SPF_ERR_HIP;
                        (void)hipStreamSynchronize(stream_of(*b));
                    }
                    b->t_sync = clock::now();
                    finish_handle_batch(b);
                    lk.lock();
                } else if (!mine) {
                    flying[b->set] = b;
                    cv_fly[b->set].notify_one();
                }
                hot_until = b->t_enq + std::chrono::microseconds(hot_us);
                continue;
            }
            // 2. time-based closing.  A group's batch is complete when the members of its previous batch are all back (submit
            // wakes this thread then); otherwise — first-time callers, a member that does not come back — it closes when nobody has
            // joined it for the quiet time: max_wait, stretched to an eighth of the last batch OF ITS KIND's time on the GPU (a
            // caller of a bootstrap needs that long to copy 256 KiB out and come back when its CPU is shared), and at most 20 quiet
            // times after its first member arrived, so that a trickle of arrivals cannot hold it open.
            int lane = -1;
            clock::time_point best{};
            for (int k = 0; k < kLanes; k++) {
                if (!filling[k] || filling[k]->n == 0) continue;
                const int op = filling[k]->op;
                // its group's previous batch is still out: the callers are not back yet (the cheap kinds by handle are fed by
                // whoever has an operation ready — the gates of a circuit's next level — not by returning callers)
                if (outstanding[k] > 0 && !stop && !filling[k]->by_handle) continue;
                const auto quiet = filling[k]->by_handle ? std::chrono::duration_cast<clock::duration>(max_wait)
                                   : std::max(std::chrono::duration_cast<clock::duration>(max_wait),
                                              std::min<clock::duration>(last_gpu_span[op] / 8, std::chrono::milliseconds(2)));
                // (the cheap kinds by handle: at most four quiet times — the gates of a circuit level trickle in as their callers
                // wake up, and the ones that are there should not wait for the last)
                const int max_quiets = (filling[k]->by_handle && !heavy(op)) ? 4 : 20;
                const auto due = (stop || everybody_is_back(*filling[k])) ? now : std::min(filling[k]->t_last + quiet, filling[k]->t0 + max_quiets * quiet);
                if (lane < 0 || due < best) { lane = k; best = due; }
            }
            if (lane >= 0 && now >= best) {
                close_batch(lane);
                continue;
            }
            if (lane >= 0) wake = std::min(wake, best);
            // the deferred table goes when nothing has joined it for the quiet time (a pusher that does not wait for anything)
            if (!deferred.empty() || !deferred_full.empty()) {
                const auto due = stop ? now : t_last_deferred + std::chrono::duration_cast<clock::duration>(max_wait);
                if (now >= due) {
                    flush_deferred();
                    continue;
                }
                wake = std::min(wake, due);
            }
            if (stop && closing.empty() && polling.empty() && lane < 0 && deferred.empty() && deferred_full.empty()) return;
            // 3. nothing to do right now.  While cheap operations by handle are in flight or were a moment ago, poll (the event
            // of a 15 us kernel, the next level's submits): going to sleep costs a wake-up per circuit level.  Otherwise sleep
            // until the next deadline or the next poke.
            if (!polling.empty() || now < hot_until) {
                const uint64_t seen = work_epoch.load(std::memory_order_acquire);
#ifdef SPF_POOL_TRACE
                struct S { Tally& t; clock::time_point t0; ~S() { t.spin += std::chrono::duration<double, std::milli>(clock::now() - t0).count(); t.n_spin++; } } s_{tally, clock::now()};
#endif
                lk.unlock();
                const auto until = std::min(wake, polling.empty() ? hot_until : now + std::chrono::microseconds(2));
                for (int i = 0; work_epoch.load(std::memory_order_acquire) == seen; i++) {
                    for (int k = 0; k < 16; k++) __builtin_ia32_pause();
                    if ((i & 3) == 3 && clock::now() >= until) break;
                }
                lk.lock();
                continue;
            }
#ifdef SPF_POOL_TRACE
            struct Z { Tally& t; clock::time_point t0; ~Z() { t.sleep += std::chrono::duration<double, std::milli>(clock::now() - t0).count(); t.n_sleep++; } } z_{tally, clock::now()};
#endif
            launcher_asleep.store(true, std::memory_order_release);
            if (wake == clock::time_point::max()) cv_work.wait(lk);
            else cv_work.wait_until(lk, wake);
            launcher_asleep.store(false, std::memory_order_release);
        }
    }

    // one completer per staging set: each waits for its own batch, so a batch is handed back when IT has finished, whatever was
    // enqueued before it on other sets
    void complete_loop(int si)
    {
        using namespace spf_pool_impl;
        std::unique_lock<Mutex> lk(mu);
        for (;;) {
            cv_fly[si].wait(lk, [&] { return flying[si] || (stop && launcher_gone); });
            if (!flying[si]) return;
            std::shared_ptr<Batch> b = flying[si];
            lk.unlock();
            (void)hipSetDevice(ctx->device);
            if (b->st == SPF_OK && b->by_handle) {
                if (hipEventSynchronize(b->ev_k) != hipSuccess) {
                    b->st = SPF_ERR_HIP;
                    (void)hipStreamSynchronize(stream_of(*b));
                }
                b->t_sync = std::chrono::steady_clock::now();
            } else if (b->st == SPF_OK) {
                // first the kernels: the launcher closes and enqueues the batch that filled meanwhile, so that its copy in and
                // its kernels run under this batch's copy out
                const bool k_ok = hipEventSynchronize(b->ev_k) == hipSuccess;
                lk.lock();
                b->kernels_done = true;
                last_gpu_span[b->op] = std::chrono::steady_clock::now() - b->t_enq;
                poke();
                lk.unlock();
                if (!k_ok) b->st = SPF_ERR_HIP;
                // all chunks but the last: their waiters copy out while the rest is still on its way.  (A failure from here on
                // reaches the waiters of the chunks still asleep; the bytes already handed out were complete.)  The last chunk is
                // woken below, behind the bookkeeping that its waiters' `collected_one` relies on.
                for (int i = 0; i + 1 < b->n_chunks; i++) {
                    if (b->st == SPF_OK && hipEventSynchronize(b->ev_chunk[i]) != hipSuccess) b->st = SPF_ERR_HIP;
                    b->wake_chunk(i);
                }
                if (b->st == SPF_OK && hipEventSynchronize(b->ev_chunk[b->n_chunks - 1]) != hipSuccess) b->st = SPF_ERR_HIP;
                if (b->st != SPF_OK) (void)hipStreamSynchronize(stream_of(*b)); // nothing may still be writing the staging set
                b->t_sync = std::chrono::steady_clock::now();
            } else {
                // something was enqueued before the failure: let it drain before the staging set is reused
                (void)hipStreamSynchronize(stream_of(*b));
            }
            if (b->by_handle) {
                lk.lock();
                flying[si].reset();
                lk.unlock();
                finish_handle_batch(b);
                lk.lock();
                continue;
            }
            lk.lock();
            b->kernels_done = true;
            for (auto it = in_flight.begin(); it != in_flight.end(); ++it)
                if (it->get() == b.get()) { in_flight.erase(it); break; }
            flying[si].reset();
            b->done = true;
            b->t_done = std::chrono::steady_clock::now();
            last_done = b->t_done;
#ifdef SPF_POOL_TRACE
            {
                auto us = [](auto d) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(d).count(); };
                float gpu_ms = -1.f;
                if (b->ev_in && b->n_chunks > 0 && b->ev_chunk[b->n_chunks - 1])
                    (void)hipEventElapsedTime(&gpu_ms, b->ev_in, b->ev_chunk[b->n_chunks - 1]);
                fprintf(stderr, "[pool] batch op %d%s n %zu: filled %ld us, closed->inputs in %ld us, enqueue %ld us, enqueued->event %ld us (gpu h2d..d2h %.0f us), event->marked %ld us\n", b->op, b->by_handle ? " (by handle)" : "", b->n,
                        us(b->t_close - b->t0), us(b->t_ready - b->t_close), us(b->t_enq - b->t_ready), us(b->t_sync - b->t_enq), gpu_ms * 1e3f, us(b->t_done - b->t_sync));
            }
#endif
            collecting.push_back(b);
            outstanding[b->lane]--;
            n_launches++;
            n_ops += b->n;
            for (int w = 0; w < spf_pool_impl::Batch::kMaxWords; w++) // the last chunk — or, for a batch that failed, all of them
                if (b->chunk_word[w].load(std::memory_order_relaxed) == 0) b->wake_word(w);
            poke(); // the launcher closes the batch that filled meanwhile
        }
    }
    bool launcher_gone = false;
    std::chrono::steady_clock::duration last_gpu_span[spf_pool_impl::N_OPS] = {}; // enqueue -> kernels done of the most recent batch of a kind
    std::chrono::steady_clock::time_point last_done;      // when the most recent batch was handed to its waiters
    std::chrono::steady_clock::time_point last_enq;       // when the most recent batch was enqueued (pacing)
};
