// spf_group.hpp — every GPU of a node behind ONE host process (SURVEY.md §8 b / e).
//
// The reference's caller is a single process: one `Evaluation` (crypto/evaluation.rs:144-197) shared by the rayon
// workers of one `CircuitProcessor` (circuit_processor/mod.rs:201-209).  A group gives that process the whole node:
//   * one spf_ctx per listed device, each with a host worker thread that owns the calls made on it (hipSetDevice once,
//     its own stream, no thread creation per call);
//   * keys go up once (host -> member 0) and are replicated inside the library: a single-process RCCL communicator over
//     the distinct devices (`ncclCommInitAll`) and an in-place `ncclBroadcast` per key blob from member 0's HBM over xGMI,
//     device-to-device copies for further members on an already served device; librccl.so is dlopen'ed on first use so
//     that the library itself carries no link dependency on it;
//   * a host batch is cut into contiguous ranges of ceil(B / G) (bootstraps are independent: no data-path collective) and
//     every range runs through the member's ordinary host-pointer entry point — results are word-identical to one context
//     by construction;
//   * a member that fails with SPF_ERR_HIP leaves the rotation and its range is re-queued over the others.
//
// Included at the end of spf_hip.hip: uses its `fail` helper, the key-size helpers and the extern "C" entry points.
#pragma once

#include <rccl/rccl.h> // types and prototypes only: the functions are resolved with dlsym

#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <thread>
#include <unordered_map>

namespace spf_group_impl {

// one host thread per member: the calls of that member run here, in submission order
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false;
    void start()
    {
        th = std::thread([this] {
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return; // (stop: whatever was queued has been run)
                std::function<void()> job = std::move(q.front());
                q.pop_front();
                lk.unlock();
                job();
                lk.lock();
            }
        });
    }
    void post(std::function<void()> job)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            q.push_back(std::move(job));
        }
        cv.notify_one();
    }
    void finish()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

// completion of the jobs of one group call
struct Latch {
    std::mutex mu;
    std::condition_variable cv;
    size_t left = 0;
    void done()
    {
        std::lock_guard<std::mutex> lk(mu);
        if (--left == 0) cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return left == 0; });
    }
};

struct Member {
    spf_ctx* ctx = nullptr;
    int device = 0;
    int leader = 0;              // first member on this device (receives the RCCL broadcast; the others copy from it)
    int rccl_rank = -1;          // rank in the communicator (leaders only)
    std::atomic<bool> enabled{true}, failed{false};
    std::atomic<int> fail_next{0};
    Worker worker;
};

// librccl.so, resolved at run time
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::vector<ncclComm_t> comms;
    bool load(std::string& why)
    {
        if (handle) return true;
        const char* names[] = {getenv("SPF_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) {
            const char* e = dlerror();
            why = std::string("librccl.so cannot be loaded (") + (e ? e : "not found") +
                  "): the key broadcast needs RCCL; set SPF_GROUP_TRANSPORT=peer for plain peer copies";
            return false;
        }
#define SPF_RCCL_SYM(field, name)                                                                  \
        field = reinterpret_cast<decltype(field)>(dlsym(handle, name));                           \
        if (!field) { why = std::string("librccl.so lacks ") + name; dlclose(handle); handle = nullptr; return false; }
        SPF_RCCL_SYM(CommInitAll, "ncclCommInitAll")
        SPF_RCCL_SYM(CommDestroy, "ncclCommDestroy")
        SPF_RCCL_SYM(Broadcast, "ncclBroadcast")
        SPF_RCCL_SYM(GroupStart, "ncclGroupStart")
        SPF_RCCL_SYM(GroupEnd, "ncclGroupEnd")
        SPF_RCCL_SYM(GetErrorString, "ncclGetErrorString")
        SPF_RCCL_SYM(GetVersion, "ncclGetVersion")
#undef SPF_RCCL_SYM
        return true;
    }
};

} // namespace spf_group_impl

struct spf_group {
    using Member = spf_group_impl::Member;
    spf_params prm{};
    std::vector<std::unique_ptr<Member>> m;
    std::mutex key_mu; // key loading and replication: one at a time
    std::string err;
    mutable std::mutex err_mu;
    spf_group_impl::Rccl rccl;
    std::vector<int> leaders; // member index of each communicator rank
    enum Transport { T_NONE = 0, T_RCCL = 1, T_PEER = 2 } transport = T_NONE;
    bool transport_forced = false; // SPF_GROUP_TRANSPORT named it: a missing librccl.so is then an error, never a change of transport
    std::string transport_note;    // why the default transport was changed (librccl.so not loadable)
    double wire_seconds = 0.0, comm_init_seconds = 0.0;
    size_t bytes_per_member = 0;
    int rccl_world = 0;
    // gate-graph jobs (spf_group_run_graphs): per member the merged graph of the jobs it was dealt last time, kept while the same
    // jobs come again (a pool of circuits is usually run more than once: new input contents, same DAGs)
    struct Merged {
        std::vector<spf_graph*> jobs;
        std::vector<std::pair<size_t, size_t>> shape; // (nodes, outputs) of every job when it was merged
        spf_graph* graph = nullptr;
    };
    std::vector<Merged> merged; // [member]
    std::mutex graph_mu;        // one spf_group_run_graphs at a time
};

namespace {

spf_status gfail(spf_group* g, spf_status s, const std::string& msg)
{
    if (g) {
        std::lock_guard<std::mutex> lk(g->err_mu);
        g->err = msg;
    } else {
        g_create_error = msg;
    }
    return s;
}

// run fn(member index) on the worker thread of every listed member, wait for all; returns each member's status
std::vector<spf_status> on_members(spf_group* g, const std::vector<int>& who, const std::function<spf_status(int)>& fn)
{
    std::vector<spf_status> st(who.size(), SPF_OK);
    spf_group_impl::Latch latch;
    latch.left = who.size();
    for (size_t k = 0; k < who.size(); k++) {
        const int i = who[k];
        spf_status* slot = &st[k];
        g->m[i]->worker.post([&fn, &latch, slot, i] {
            spf_status s;
            try {
                s = fn(i);
            } catch (const std::exception&) { // (nothing below throws by design; never across the worker's frame)
                s = SPF_ERR_HIP;
            }
            *slot = s;
            latch.done();
        });
    }
    if (!who.empty()) latch.wait();
    return st;
}

std::vector<int> members_in_rotation(spf_group* g)
{
    std::vector<int> r;
    for (size_t i = 0; i < g->m.size(); i++)
        if (g->m[i]->enabled.load() && !g->m[i]->failed.load()) r.push_back((int)i);
    return r;
}

// The batch split.  `call(ctx, first, count)` runs one contiguous range on one member.  SPF_ERR_HIP from a member takes it
// out of rotation and re-queues its range over the others; any other failure is the caller's and ends the call.
spf_status group_split(spf_group* g, size_t B, const std::function<spf_status(spf_ctx*, size_t, size_t)>& call)
{
    if (!g) return gfail(nullptr, SPF_ERR_INVALID_ARGUMENT, "null group");
    if (B == 0) return SPF_OK;
    struct Range { size_t first, count; };
    std::vector<Range> todo{{0, B}};
    while (!todo.empty()) {
        const std::vector<int> rot = members_in_rotation(g);
        if (rot.empty()) return gfail(g, SPF_ERR_HIP, "no member of the group is in rotation (" + [&] {
            std::lock_guard<std::mutex> lk(g->err_mu);
            return g->err.empty() ? std::string("all disabled") : g->err;
        }() + ")");
        // cut every pending range over the members in rotation: ceil(count / G) contiguous units each
        struct Piece { int member; Range r; };
        std::vector<Piece> pieces;
        for (const Range& r : todo) {
            const size_t per = (r.count + rot.size() - 1) / rot.size();
            for (size_t k = 0; k < rot.size(); k++) {
                const size_t first = std::min(r.count, k * per), count = std::min(per, r.count - first);
                if (count) pieces.push_back({rot[k], {r.first + first, count}});
            }
        }
        todo.clear();
        std::vector<spf_status> st(pieces.size(), SPF_OK);
        spf_group_impl::Latch latch;
        latch.left = pieces.size();
        for (size_t k = 0; k < pieces.size(); k++) {
            const Piece pc = pieces[k];
            spf_status* slot = &st[k];
            spf_group::Member* mem = g->m[pc.member].get();
            mem->worker.post([&call, &latch, slot, pc, mem] {
                spf_status s;
                int f = mem->fail_next.load();
                if (f > 0 && mem->fail_next.compare_exchange_strong(f, f - 1)) {
                    fail(mem->ctx, SPF_ERR_HIP, "injected device failure (spf_group_debug_fail_next)");
                    s = SPF_ERR_HIP;
                } else {
                    try {
                        s = call(mem->ctx, pc.r.first, pc.r.count);
                    } catch (const std::exception&) {
                        s = SPF_ERR_HIP;
                    }
                }
                *slot = s;
                latch.done();
            });
        }
        latch.wait();
        for (size_t k = 0; k < pieces.size(); k++) {
            if (st[k] == SPF_OK) continue;
            spf_group::Member* mem = g->m[pieces[k].member].get();
            const std::string why = "member " + std::to_string(pieces[k].member) + " (device " + std::to_string(mem->device) +
                                    "): " + spf_last_error(mem->ctx);
            if (st[k] != SPF_ERR_HIP) return gfail(g, st[k], why);
            mem->failed.store(true);
            gfail(g, SPF_ERR_HIP, why);
            todo.push_back(pieces[k].r);
        }
    }
    return SPF_OK;
}

bool blob_ready(spf_ctx* c, int which)
{
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    return which == 0 ? c->bsk_ready : which == 1 ? c->ksk_ready : which == 2 ? c->ak_ready : c->ssk_ready;
}

spf_status ensure_rccl(spf_group* g)
{
    if (!g->rccl.comms.empty()) return SPF_OK;
    std::string why;
    if (!g->rccl.load(why)) return gfail(g, SPF_ERR_HIP, why);
    std::vector<int> devs;
    for (int i : g->leaders) devs.push_back(g->m[i]->device);
    g->rccl.comms.assign(devs.size(), nullptr);
    const auto t0 = std::chrono::steady_clock::now();
    const ncclResult_t r = g->rccl.CommInitAll(g->rccl.comms.data(), (int)devs.size(), devs.data());
    g->comm_init_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (r != ncclSuccess) {
        g->rccl.comms.clear();
        return gfail(g, SPF_ERR_HIP, std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(r));
    }
    g->rccl_world = (int)devs.size();
    return SPF_OK;
}

// member 0's blob `which` -> every other member's blob, then every member derives its own images (commit)
struct CallerDevice { // the calling thread's current device is put back when a group call that switches devices returns
    int prev = -1;
    CallerDevice() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
    ~CallerDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
};

spf_status replicate_blob(spf_group* g, int which)
{
    CallerDevice restore;
    const int G = (int)g->m.size();
    if (G == 1 && g->transport != spf_group::T_RCCL) return SPF_OK;
    std::vector<void*> ptr(G, nullptr);
    size_t bytes = 0;
    for (int i = 0; i < G; i++) {
        size_t b = 0;
        spf_status s = spf_key_blob(g->m[i]->ctx, which, &ptr[i], &b);
        if (s != SPF_OK) return gfail(g, s, std::string("member ") + std::to_string(i) + ": " + spf_last_error(g->m[i]->ctx));
        if (i && b != bytes) return gfail(g, SPF_ERR_INVALID_ARGUMENT, "key blobs of the members differ in size");
        bytes = b;
    }
    auto hipfail = [&](const char* what, hipError_t e) { return gfail(g, SPF_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); };
    if (g->transport == spf_group::T_RCCL) {
        spf_status s = ensure_rccl(g); // (communicator set-up is timed by itself: comm_init_seconds)
        if (s != SPF_OK) {
            // RCCL was only the DEFAULT for a group of several members: without a loadable librccl.so plain peer copies do the same
            // job (said in the replication stats).  A transport the environment asked for by name is never replaced.
            if (g->transport_forced || g->rccl.handle) return s;
            g->transport = spf_group::T_PEER;
            {
                std::lock_guard<std::mutex> lk(g->err_mu);
                g->transport_note = g->err;
            }
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (g->transport == spf_group::T_RCCL) {
        // in place: the root sends from its blob, everyone else receives into theirs; one call per rank inside a group
        // (a single thread drives every rank of the communicator)
        ncclResult_t r = g->rccl.GroupStart();
        for (size_t k = 0; r == ncclSuccess && k < g->leaders.size(); k++) {
            spf_group::Member* mem = g->m[g->leaders[k]].get();
            hipError_t e = hipSetDevice(mem->device);
            if (e != hipSuccess) { (void)g->rccl.GroupEnd(); return hipfail("hipSetDevice", e); }
            r = g->rccl.Broadcast(ptr[g->leaders[k]], ptr[g->leaders[k]], bytes, ncclUint8, 0, g->rccl.comms[k], mem->ctx->stream);
        }
        const ncclResult_t r2 = g->rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) return gfail(g, SPF_ERR_HIP, std::string("ncclBroadcast: ") + g->rccl.GetErrorString(r));
        for (int i : g->leaders) {
            hipError_t e = hipSetDevice(g->m[i]->device);
            if (e == hipSuccess) e = hipStreamSynchronize(g->m[i]->ctx->stream);
            if (e != hipSuccess) return hipfail("hipStreamSynchronize after ncclBroadcast", e);
        }
    } else {
        // plain peer copies from member 0's HBM to each other leader (runtime picks xGMI P2P when the devices allow it)
        {
            hipError_t e0 = hipSetDevice(g->m[0]->device); // (the copies are enqueued on member 0's stream: its device is current)
            if (e0 != hipSuccess) return hipfail("hipSetDevice", e0);
        }
        for (int i : g->leaders) {
            if (i == 0) continue;
            hipError_t e = hipMemcpyPeerAsync(ptr[i], g->m[i]->device, ptr[0], g->m[0]->device, bytes, g->m[0]->ctx->stream);
            if (e != hipSuccess) return hipfail("hipMemcpyPeerAsync", e);
        }
        hipError_t e = hipSetDevice(g->m[0]->device);
        if (e == hipSuccess) e = hipStreamSynchronize(g->m[0]->ctx->stream);
        if (e != hipSuccess) return hipfail("hipStreamSynchronize after the peer copies", e);
    }
    // further members on an already served device: device-to-device
    for (int i = 1; i < G; i++) {
        spf_group::Member* mem = g->m[i].get();
        if (mem->leader == i) continue;
        hipError_t e = hipSetDevice(mem->device);
        if (e == hipSuccess) e = hipMemcpyAsync(ptr[i], ptr[mem->leader], bytes, hipMemcpyDeviceToDevice, mem->ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(mem->ctx->stream);
        if (e != hipSuccess) return hipfail("device-to-device key copy", e);
    }
    g->wire_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    g->bytes_per_member += bytes;
    // every other member derives its images from its replica, all at once
    std::vector<int> others;
    for (int i = 1; i < G; i++) others.push_back(i);
    const std::vector<spf_status> st = on_members(g, others, [&](int i) { return spf_key_blob_commit(g->m[i]->ctx, which); });
    for (size_t k = 0; k < st.size(); k++)
        if (st[k] != SPF_OK)
            return gfail(g, st[k], "member " + std::to_string(others[k]) + ": " + spf_last_error(g->m[others[k]]->ctx));
    return SPF_OK;
}

template <class Load>
spf_status group_load(spf_group* g, int which, Load&& load)
{
    if (!g) return gfail(nullptr, SPF_ERR_INVALID_ARGUMENT, "null group");
    std::lock_guard<std::mutex> lk(g->key_mu);
    spf_status s = load(g->m[0]->ctx);
    if (s != SPF_OK) return gfail(g, s, std::string("member 0: ") + spf_last_error(g->m[0]->ctx));
    return replicate_blob(g, which);
}

} // namespace

// The pool a submit of the calling thread goes to.  An ordinary pool: itself.  A group pool: the thread's home member,
// dealt round-robin over the members in rotation on the thread's first submit (and again should its home leave the rotation):
// the callers of one device keep meeting in the same batches, which is what the pool's closing rules count on.
static spf_pool* pool_deal(spf_pool* top, int* member)
{
    *member = 0;
    if (top->members.empty()) return top;
    const uintptr_t who = (uintptr_t)pthread_self();
    std::lock_guard<std::mutex> lk(top->deal_mu);
    auto in_rotation = [&](int i) { return top->grp->m[i]->enabled.load() && !top->grp->m[i]->failed.load(); };
    auto it = top->home.find(who);
    if (it != top->home.end() && in_rotation(it->second)) {
        *member = it->second;
        return top->members[*member];
    }
    const size_t G = top->members.size();
    for (size_t k = 0; k < G; k++) {
        const int i = (int)((top->next_home + k) % G);
        if (!in_rotation(i)) continue;
        top->next_home = (size_t)i + 1;
        try {
            top->home[who] = i;
        } catch (const std::exception&) { // (out of memory: the thread is simply dealt again next time)
        }
        *member = i;
        return top->members[i];
    }
    return nullptr;
}

extern "C" {

spf_status spf_group_create(const spf_params* params, const int* device_ids, int n_devices, spf_group** out)
{
    if (!params || !device_ids || !out || n_devices <= 0 || n_devices > 256)
        return gfail(nullptr, SPF_ERR_INVALID_ARGUMENT, "spf_group_create: null argument or n_devices outside 1..256");
    *out = nullptr;
    spf_group* g = new (std::nothrow) spf_group();
    if (!g) return gfail(nullptr, SPF_ERR_HIP, "out of host memory");
    g->prm = *params;
    for (int i = 0; i < n_devices; i++) {
        std::unique_ptr<spf_group::Member> mem(new (std::nothrow) spf_group::Member());
        spf_status s = mem ? spf_create(params, device_ids[i], &mem->ctx) : SPF_ERR_HIP;
        if (s != SPF_OK) { // (the message of the failing spf_create is already the calling thread's)
            spf_group_destroy(g);
            return s;
        }
        mem->device = device_ids[i];
        mem->leader = i;
        for (int j = 0; j < i; j++)
            if (g->m[j]->device == device_ids[i]) { mem->leader = g->m[j]->leader; break; }
        if (mem->leader == i) {
            mem->rccl_rank = (int)g->leaders.size();
            g->leaders.push_back(i);
        }
        try {
            mem->worker.start();
        } catch (const std::exception& e) {
            spf_destroy(mem->ctx);
            spf_group_destroy(g);
            return gfail(nullptr, SPF_ERR_HIP, std::string("spf_group_create: cannot start a member thread: ") + e.what());
        }
        g->m.push_back(std::move(mem));
    }
    const char* t = getenv("SPF_GROUP_TRANSPORT");
    g->transport_forced = t && *t;
    if (t && !strcmp(t, "peer")) g->transport = spf_group::T_PEER;
    else if (t && !strcmp(t, "rccl")) g->transport = spf_group::T_RCCL;
    else if (t && *t) {
        spf_group_destroy(g);
        return gfail(nullptr, SPF_ERR_INVALID_ARGUMENT, "SPF_GROUP_TRANSPORT must be \"rccl\" or \"peer\"");
    } else g->transport = n_devices > 1 ? spf_group::T_RCCL : spf_group::T_NONE;
    // every member's hipSetDevice happens once, on its own thread
    std::vector<int> all;
    for (int i = 0; i < n_devices; i++) all.push_back(i);
    const std::vector<spf_status> st = on_members(g, all, [&](int i) {
        return hipSetDevice(g->m[i]->device) == hipSuccess ? SPF_OK : SPF_ERR_HIP;
    });
    for (spf_status s : st)
        if (s != SPF_OK) {
            spf_group_destroy(g);
            return gfail(nullptr, SPF_ERR_HIP, "spf_group_create: hipSetDevice failed on a member thread");
        }
    *out = g;
    return SPF_OK;
}

void spf_group_destroy(spf_group* g)
{
    if (!g) return;
    for (auto& mem : g->m) mem->worker.finish();
    for (auto& mg : g->merged)
        if (mg.graph) spf_graph_destroy(mg.graph);
    if (g->rccl.handle)
        for (ncclComm_t c : g->rccl.comms)
            if (c) (void)g->rccl.CommDestroy(c);
    for (auto& mem : g->m) spf_destroy(mem->ctx);
    // (librccl.so stays loaded: unloading a library that owns device state at process teardown is not worth the risk)
    delete g;
}

int spf_group_size(const spf_group* g) { return g ? (int)g->m.size() : 0; }

spf_ctx* spf_group_ctx(spf_group* g, int member)
{
    return (g && member >= 0 && member < (int)g->m.size()) ? g->m[member]->ctx : nullptr;
}

const char* spf_group_last_error(const spf_group* g)
{
    thread_local std::string copy;
    if (!g) return g_create_error.c_str();
    std::lock_guard<std::mutex> lk(g->err_mu);
    copy = g->err;
    return copy.c_str();
}

spf_status spf_group_load_bootstrap_key(spf_group* g, const double* bsk_fft, size_t n_complex)
{
    return group_load(g, 0, [&](spf_ctx* c) { return spf_load_bootstrap_key(c, bsk_fft, n_complex); });
}
spf_status spf_group_load_keyswitch_key(spf_group* g, const uint64_t* ksk, size_t n_words)
{
    return group_load(g, 1, [&](spf_ctx* c) { return spf_load_keyswitch_key(c, ksk, n_words); });
}
spf_status spf_group_load_automorphism_key(spf_group* g, const double* ak_fft, size_t n_complex)
{
    return group_load(g, 2, [&](spf_ctx* c) { return spf_load_automorphism_key(c, ak_fft, n_complex); });
}
spf_status spf_group_load_scheme_switch_key(spf_group* g, const double* ssk_fft, size_t n_complex)
{
    return group_load(g, 3, [&](spf_ctx* c) { return spf_load_scheme_switch_key(c, ssk_fft, n_complex); });
}

spf_status spf_group_replicate_keys(spf_group* g)
{
    if (!g) return gfail(nullptr, SPF_ERR_INVALID_ARGUMENT, "null group");
    std::lock_guard<std::mutex> lk(g->key_mu);
    for (int which = 0; which < 4; which++) {
        if (!blob_ready(g->m[0]->ctx, which)) continue;
        spf_status s = replicate_blob(g, which);
        if (s != SPF_OK) return s;
    }
    return SPF_OK;
}

spf_status spf_group_load_compute_key_bincode(spf_group* g, const uint8_t* bytes, size_t len)
{
    if (!g) return gfail(nullptr, SPF_ERR_INVALID_ARGUMENT, "null group");
    {
        std::lock_guard<std::mutex> lk(g->key_mu);
        spf_status s = spf_load_compute_key_bincode(g->m[0]->ctx, bytes, len);
        if (s != SPF_OK) return gfail(g, s, std::string("member 0: ") + spf_last_error(g->m[0]->ctx));
    }
    return spf_group_replicate_keys(g);
}

spf_status spf_group_replication_stats(spf_group* g, double* wire_seconds, double* comm_init_seconds, size_t* bytes_per_member,
                                       int* rccl_world_size, const char** transport)
{
    if (!g) return gfail(nullptr, SPF_ERR_INVALID_ARGUMENT, "null group");
    std::lock_guard<std::mutex> lk(g->key_mu);
    if (wire_seconds) *wire_seconds = g->wire_seconds;
    if (comm_init_seconds) *comm_init_seconds = g->comm_init_seconds;
    if (bytes_per_member) *bytes_per_member = g->bytes_per_member;
    if (rccl_world_size) *rccl_world_size = g->rccl_world;
    if (transport)
        *transport = g->transport == spf_group::T_RCCL ? "rccl"
                     : g->transport == spf_group::T_PEER ? (g->transport_note.empty() ? "peer" : "peer (librccl.so could not be loaded)") : "none";
    return SPF_OK;
}

spf_status spf_group_set_member_enabled(spf_group* g, int member, int enabled)
{
    if (!g || member < 0 || member >= (int)g->m.size()) return gfail(g, SPF_ERR_INVALID_ARGUMENT, "member out of range");
    g->m[member]->enabled.store(enabled != 0);
    if (enabled) g->m[member]->failed.store(false);
    return SPF_OK;
}

int spf_group_members_in_rotation(spf_group* g) { return g ? (int)members_in_rotation(g).size() : 0; }

spf_status spf_group_debug_fail_next(spf_group* g, int member, int count)
{
    if (!g || member < 0 || member >= (int)g->m.size() || count < 0) return gfail(g, SPF_ERR_INVALID_ARGUMENT, "member out of range");
    g->m[member]->fail_next.store(count);
    return SPF_OK;
}

// ---- the batch forms: the member's own host-pointer entry point on its range

#define SPF_GROUP_NULL(cond)                                                                       \
    if (!g) return gfail(nullptr, SPF_ERR_INVALID_ARGUMENT, "null group");                         \
    if (B && (cond)) return gfail(g, SPF_ERR_INVALID_ARGUMENT, "null argument")

spf_status spf_group_keyswitch_lwe_l1_lwe_l0_batch(spf_group* g, size_t B, const uint64_t* in, uint64_t* out)
{
    SPF_GROUP_NULL(!in || !out);
    const size_t wi = lwe1_words(g->prm), wo = lwe0_words(g->prm);
    return group_split(g, B, [=](spf_ctx* c, size_t at, size_t n) { return spf_keyswitch_lwe_l1_lwe_l0_batch(c, n, in + at * wi, out + at * wo); });
}

spf_status spf_group_generalized_pbs_batch(spf_group* g, size_t B, const uint64_t* lwe, const uint64_t* lut, size_t lut_stride,
                                           uint32_t log_chi, uint32_t log_v, uint64_t body_rotate, uint64_t* out)
{
    SPF_GROUP_NULL(!lwe || !lut || !out);
    const size_t wi = lwe0_words(g->prm), wo = glwe_words(g->prm);
    return group_split(g, B, [=](spf_ctx* c, size_t at, size_t n) {
        return spf_generalized_pbs_batch(c, n, lwe + at * wi, lut + at * lut_stride, lut_stride, log_chi, log_v, body_rotate, out + at * wo);
    });
}

spf_status spf_group_pbs_univariate_batch(spf_group* g, size_t B, const uint64_t* lwe, const uint64_t* lut, size_t lut_stride,
                                          uint64_t* out)
{
    SPF_GROUP_NULL(!lwe || !lut || !out);
    const size_t wi = lwe0_words(g->prm), wo = lwe1_words(g->prm);
    return group_split(g, B, [=](spf_ctx* c, size_t at, size_t n) {
        return spf_pbs_univariate_batch(c, n, lwe + at * wi, lut + at * lut_stride, lut_stride, out + at * wo);
    });
}

spf_status spf_group_circuit_bootstrap_pbs_batch(spf_group* g, size_t B, const uint64_t* lwe, uint64_t* out)
{
    SPF_GROUP_NULL(!lwe || !out);
    const size_t wi = lwe0_words(g->prm), wo = glwe_words(g->prm);
    return group_split(g, B, [=](spf_ctx* c, size_t at, size_t n) { return spf_circuit_bootstrap_pbs_batch(c, n, lwe + at * wi, out + at * wo); });
}

spf_status spf_group_circuit_bootstrap_batch(spf_group* g, size_t B, const uint64_t* lwe, double* out)
{
    SPF_GROUP_NULL(!lwe || !out);
    const size_t wi = lwe0_words(g->prm), wo = 2 * ggsw_fft_complex(g->prm, g->prm.cbs_radix_count);
    return group_split(g, B, [=](spf_ctx* c, size_t at, size_t n) { return spf_circuit_bootstrap_batch(c, n, lwe + at * wi, out + at * wo); });
}

spf_status spf_group_mod_switch_trace_and_rotate_batch(spf_group* g, size_t B, const uint64_t* in, uint64_t* out)
{
    SPF_GROUP_NULL(!in || !out);
    const size_t wi = glwe_words(g->prm), wo = wi * g->prm.cbs_radix_count;
    return group_split(g, B, [=](spf_ctx* c, size_t at, size_t n) { return spf_mod_switch_trace_and_rotate_batch(c, n, in + at * wi, out + at * wo); });
}

spf_status spf_group_scheme_switch_batch(spf_group* g, size_t B, const uint64_t* in, double* out)
{
    SPF_GROUP_NULL(!in || !out);
    const size_t wi = glwe_words(g->prm) * g->prm.cbs_radix_count, wo = 2 * ggsw_fft_complex(g->prm, g->prm.cbs_radix_count);
    return group_split(g, B, [=](spf_ctx* c, size_t at, size_t n) { return spf_scheme_switch_batch(c, n, in + at * wi, out + at * wo); });
}

spf_status spf_group_sample_extract_l1_batch(spf_group* g, size_t B, const uint64_t* in, size_t idx, uint64_t* out)
{
    SPF_GROUP_NULL(!in || !out);
    const size_t wi = glwe_words(g->prm), wo = lwe1_words(g->prm);
    return group_split(g, B, [=](spf_ctx* c, size_t at, size_t n) { return spf_sample_extract_l1_batch(c, n, in + at * wi, idx, out + at * wo); });
}

spf_status spf_group_glwe_not_batch(spf_group* g, size_t B, const uint64_t* in, uint64_t* out)
{
    SPF_GROUP_NULL(!in || !out);
    const size_t w = glwe_words(g->prm);
    return group_split(g, B, [=](spf_ctx* c, size_t at, size_t n) { return spf_glwe_not_batch(c, n, in + at * w, out + at * w); });
}

spf_status spf_group_glwe_xor_batch(spf_group* g, size_t B, const uint64_t* a, const uint64_t* b, uint64_t* out)
{
    SPF_GROUP_NULL(!a || !b || !out);
    const size_t w = glwe_words(g->prm);
    return group_split(g, B, [=](spf_ctx* c, size_t at, size_t n) { return spf_glwe_xor_batch(c, n, a + at * w, b + at * w, out + at * w); });
}

spf_status spf_group_glwe_mul_xn_batch(spf_group* g, size_t B, const uint64_t* in, size_t amount, uint64_t* out)
{
    SPF_GROUP_NULL(!in || !out);
    const size_t w = glwe_words(g->prm);
    return group_split(g, B, [=](spf_ctx* c, size_t at, size_t n) { return spf_glwe_mul_xn_batch(c, n, in + at * w, amount, out + at * w); });
}

spf_status spf_group_cmux_batch(spf_group* g, size_t B, const double* sel, const uint64_t* a, const uint64_t* b, uint64_t* out)
{
    SPF_GROUP_NULL(!sel || !a || !b || !out);
    const size_t w = glwe_words(g->prm), sw = 2 * ggsw_fft_complex(g->prm, g->prm.cbs_radix_count);
    return group_split(g, B, [=](spf_ctx* c, size_t at, size_t n) { return spf_cmux_batch(c, n, sel + at * sw, a + at * w, b + at * w, out + at * w); });
}

spf_status spf_group_glev_cmux_batch(spf_group* g, size_t B, const double* sel, const uint64_t* a, const uint64_t* b, uint64_t* out)
{
    SPF_GROUP_NULL(!sel || !a || !b || !out);
    const size_t w = glwe_words(g->prm) * g->prm.cbs_radix_count, sw = 2 * ggsw_fft_complex(g->prm, g->prm.cbs_radix_count);
    return group_split(g, B, [=](spf_ctx* c, size_t at, size_t n) { return spf_glev_cmux_batch(c, n, sel + at * sw, a + at * w, b + at * w, out + at * w); });
}

spf_status spf_group_multiply_glwe_ggsw_batch(spf_group* g, size_t B, const uint64_t* glwe, const double* ggsw, uint64_t* out)
{
    SPF_GROUP_NULL(!glwe || !ggsw || !out);
    const size_t w = glwe_words(g->prm), sw = 2 * ggsw_fft_complex(g->prm, g->prm.cbs_radix_count);
    return group_split(g, B, [=](spf_ctx* c, size_t at, size_t n) { return spf_multiply_glwe_ggsw_batch(c, n, glwe + at * w, ggsw + at * sw, out + at * w); });
}

spf_status spf_group_gate_bootstrap_batch(spf_group* g, size_t B, const uint64_t* in, uint64_t* out)
{
    SPF_GROUP_NULL(!in || !out);
    const size_t wi = lwe1_words(g->prm), wo = glwe_words(g->prm);
    return group_split(g, B, [=](spf_ctx* c, size_t at, size_t n) { return spf_gate_bootstrap_batch(c, n, in + at * wi, out + at * wo); });
}

spf_status spf_group_keyswitch_circuit_bootstrap_batch(spf_group* g, size_t B, const uint64_t* in, double* out)
{
    SPF_GROUP_NULL(!in || !out);
    const size_t wi = lwe1_words(g->prm), wo = 2 * ggsw_fft_complex(g->prm, g->prm.cbs_radix_count);
    return group_split(g, B, [=](spf_ctx* c, size_t at, size_t n) { return spf_keyswitch_circuit_bootstrap_batch(c, n, in + at * wi, out + at * wo); });
}
#undef SPF_GROUP_NULL

spf_status spf_group_l1ggsw_constant(spf_group* g, int bit, double* out)
{
    if (!g) return gfail(nullptr, SPF_ERR_INVALID_ARGUMENT, "null group");
    const std::vector<int> rot = members_in_rotation(g);
    if (rot.empty()) return gfail(g, SPF_ERR_HIP, "no member of the group is in rotation");
    const std::vector<spf_status> st = on_members(g, {rot[0]}, [&](int i) { return spf_l1ggsw_constant(g->m[i]->ctx, bit, out); });
    if (st[0] != SPF_OK) return gfail(g, st[0], std::string("member ") + std::to_string(rot[0]) + ": " + spf_last_error(g->m[rot[0]]->ctx));
    return SPF_OK;
}

// ---- gate-graph jobs over the group (SURVEY.md §8 e / f3; BASELINE config 5: "32x32-bit encrypted multiply ... 8xMI355X gate pool")
//
// The reference's processor is one per machine and is fed whole `FheCircuit`s (circuit_processor/mod.rs:573-623; the multiplier
// circuits of circuits/mul.rs:90-200).  A gate graph is a chain of dependent CMUX levels hanging off one wide level of
// conversions: cutting ONE graph across GPUs would put a 256 KiB GGSW on the wire for every selector that crosses the cut,
// independent graphs need nothing.  So the unit dealt to a device is the JOB (one graph): longest-processing-time first over
// the members in rotation (the rule of spf_amd/gate_pool.py, here behind the C ABI for a one-process host), every member's
// jobs are lowered into ONE graph on its device — the level-batching executor then sees K jobs x gates per level — and the
// members run side by side on their worker threads.  No data-path collective.

} // extern "C"

namespace {

// relative cost of a job: what its launches are made of (a circuit bootstrap is ~60 us per ciphertext of a batch, a CMUX level
// ~20 us per launch: what matters is only that equal jobs weigh the same and bigger jobs more)
double graph_cost(const spf_graph* g)
{
    double c = 0;
    for (const auto& n : g->nodes)
        c += n.op == SPF_OP_CIRCUIT_BOOTSTRAP ? 50.0 : (n.op >= 0 ? 1.0 : 0.0);
    return c;
}

// `jobs` appended into one graph on `c`, node ids shifted; inputs and outputs keep pointing at the jobs' own host buffers
spf_status merge_jobs(spf_ctx* c, const std::vector<spf_graph*>& jobs, spf_graph** out)
{
    spf_graph* m = nullptr;
    spf_status st = spf_graph_create(c, &m);
    if (st != SPF_OK) return st;
    try {
        for (const spf_graph* j : jobs) {
            const uint32_t base = (uint32_t)m->nodes.size();
            for (spf_graph::Node n : j->nodes) {
                for (uint32_t i = 0; i < n.n_in; i++) n.in[i] += base;
                m->nodes.push_back(n);
            }
            for (const auto& o : j->outputs) m->outputs.emplace_back(o.first + base, o.second);
        }
    } catch (const std::exception&) {
        spf_graph_destroy(m);
        return fail(c, SPF_ERR_HIP, "out of host memory");
    }
    *out = m;
    return SPF_OK;
}

} // namespace

static spf_status group_run_graphs(spf_group* g, spf_graph* const* graphs, size_t n)
{
    if (!g || (n && !graphs)) return gfail(g, SPF_ERR_INVALID_ARGUMENT, "null argument");
    for (size_t i = 0; i < n; i++)
        if (!graphs[i] || graphs[i]->grp != g) return gfail(g, SPF_ERR_INVALID_ARGUMENT, "spf_group_run_graphs: a graph that was not made by spf_group_graph_create of this group");
    if (n == 0) return SPF_OK;
    std::lock_guard<std::mutex> whole(g->graph_mu);
    const int G = (int)g->m.size();
    try {
        g->merged.resize((size_t)G);
    } catch (const std::exception&) {
        return gfail(g, SPF_ERR_HIP, "out of host memory");
    }
    std::vector<size_t> todo(n);
    for (size_t i = 0; i < n; i++) todo[i] = i;
    while (!todo.empty()) {
        const std::vector<int> rot = members_in_rotation(g);
        if (rot.empty()) return gfail(g, SPF_ERR_HIP, "no member of the group is in rotation");
        // longest processing time first: the jobs by falling cost (ties by index: deterministic), each to the member with
        // the least load so far
        std::vector<double> cost(todo.size());
        for (size_t k = 0; k < todo.size(); k++) cost[k] = graph_cost(graphs[todo[k]]);
        std::vector<size_t> order(todo.size());
        for (size_t k = 0; k < order.size(); k++) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });
        std::vector<double> load(rot.size(), 0.0);
        std::vector<std::vector<size_t>> dealt(rot.size()); // per member in rotation: indices into `graphs`
        for (size_t k : order) {
            size_t best = 0;
            for (size_t r = 1; r < rot.size(); r++)
                if (load[r] < load[best]) best = r;
            dealt[best].push_back(todo[k]);
            load[best] += cost[k];
        }
        std::vector<int> who;
        std::vector<size_t> slot_of; // index into dealt for who[k]
        for (size_t r = 0; r < rot.size(); r++)
            if (!dealt[r].empty()) {
                std::sort(dealt[r].begin(), dealt[r].end()); // (the jobs of a member in the caller's order: a stable merged graph)
                who.push_back(rot[r]);
                slot_of.push_back(r);
            }
        const std::vector<spf_status> st = on_members(g, who, [&](int member) -> spf_status {
            size_t r = 0;
            for (size_t k = 0; k < who.size(); k++)
                if (who[k] == member) r = slot_of[k];
            spf_group::Member* mem = g->m[member].get();
            int f = mem->fail_next.load();
            if (f > 0 && mem->fail_next.compare_exchange_strong(f, f - 1))
                return fail(mem->ctx, SPF_ERR_HIP, "injected device failure (spf_group_debug_fail_next)");
            std::vector<spf_graph*> jobs;
            std::vector<std::pair<size_t, size_t>> shape;
            for (size_t i : dealt[r]) {
                jobs.push_back(graphs[i]);
                shape.emplace_back(graphs[i]->nodes.size(), graphs[i]->outputs.size());
            }
            spf_group::Merged& mg = g->merged[(size_t)member];
            if (!mg.graph || mg.jobs != jobs || mg.shape != shape) { // other jobs than last time (or a job changed): lower them again
                if (mg.graph) spf_graph_destroy(mg.graph);
                mg.graph = nullptr;
                spf_status s = merge_jobs(mem->ctx, jobs, &mg.graph);
                if (s != SPF_OK) return s;
                mg.jobs = jobs;
                mg.shape = shape;
            }
            const spf_status s = spf_graph_impl::run(mg.graph);
            if (s == SPF_OK)
                for (spf_graph* j : jobs) { j->member = member; j->n_levels = mg.graph->n_levels; j->n_launches = mg.graph->n_launches; }
            return s;
        });
        todo.clear();
        for (size_t k = 0; k < who.size(); k++) {
            if (st[k] == SPF_OK) continue;
            spf_group::Member* mem = g->m[who[k]].get();
            const std::string why = "member " + std::to_string(who[k]) + " (device " + std::to_string(mem->device) + "): " + spf_last_error(mem->ctx);
            if (st[k] != SPF_ERR_HIP) return gfail(g, st[k], why);
            mem->failed.store(true); // out of rotation; its jobs are dealt again over the others
            gfail(g, SPF_ERR_HIP, why);
            for (size_t i : dealt[slot_of[k]]) todo.push_back(i);
        }
    }
    return SPF_OK;
}

// a job is going away: no merged graph may keep its node list or its buffers
static void group_forget_graph(spf_group* g, spf_graph* graph)
{
    std::lock_guard<std::mutex> whole(g->graph_mu);
    for (auto& mg : g->merged)
        if (std::find(mg.jobs.begin(), mg.jobs.end(), graph) != mg.jobs.end()) {
            if (mg.graph) spf_graph_destroy(mg.graph);
            mg.graph = nullptr;
            mg.jobs.clear();
            mg.shape.clear();
        }
}

extern "C" {

spf_status spf_group_graph_create(spf_group* g, spf_graph** out)
{
    if (!g || !out) return gfail(g, SPF_ERR_INVALID_ARGUMENT, "null argument");
    spf_status s = spf_graph_create(g->m[0]->ctx, out); // (member 0's context: parameters and build-time validation only)
    if (s != SPF_OK) return gfail(g, s, std::string("member 0: ") + spf_last_error(g->m[0]->ctx));
    (*out)->grp = g;
    return SPF_OK;
}

spf_status spf_group_run_graphs(spf_group* g, spf_graph* const* graphs, size_t n) { return group_run_graphs(g, graphs, n); }

int spf_graph_member(const spf_graph* graph) { return graph ? (graph->grp ? graph->member : 0) : -1; }

// ---- call coalescing over the group: one pool per member, threads dealt round-robin on their first submit

spf_status spf_pool_create_group(spf_group* g, size_t max_batch, uint32_t max_wait_us, spf_pool** out)
{
    if (!g || !out || max_batch == 0) return gfail(g, SPF_ERR_INVALID_ARGUMENT, "null argument or max_batch == 0");
    *out = nullptr;
    spf_pool* top = new (std::nothrow) spf_pool();
    if (!top) return gfail(g, SPF_ERR_HIP, "out of host memory");
    top->ctx = g->m[0]->ctx; top->prm = g->prm; top->max_batch = max_batch; top->grp = g;
    for (auto& mem : g->m) {
        spf_pool* p = nullptr;
        spf_status s = spf_pool_create(mem->ctx, max_batch, max_wait_us, &p);
        if (s != SPF_OK) {
            for (spf_pool* q : top->members) spf_pool_destroy(q);
            delete top;
            return gfail(g, s, std::string("spf_pool_create on device ") + std::to_string(mem->device) + ": " + spf_last_error(mem->ctx));
        }
        top->members.push_back(p);
    }
    *out = top;
    return SPF_OK;
}

} // extern "C"
