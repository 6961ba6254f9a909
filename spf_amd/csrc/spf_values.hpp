// spf_values.hpp — device-resident ciphertext values for the per-operation boundary (SURVEY.md §8 b / f3).
//
// The reference keeps every ciphertext in host memory (parasol_runtime/src/crypto/encryption.rs:143-165) and hands
// `Evaluation` one `&L1GgswCiphertext` / `&L1GlweCiphertext` per `FheOp` from a rayon worker (circuit_processor/mod.rs:255-540).
// Taken literally on a GPU that moves a 256 KiB selector and two 32 KiB ciphertexts over PCIe for a 15 us CMUX — and
// ships the GGSW a circuit bootstrap just produced to the host only for the ~45 CMux gates that consume it
// (fhe_circuit.rs:473-494) to send it back.  A *value* keeps the ciphertext where the next operation needs it: in HBM.
//
//   * a value is a reference-counted view (block, offset) into device memory of ONE context; the handle forms of the pool's
//     submits (`spf_pool_submit_*_v`) take values as operands and return a value as result: nothing crosses PCIe unless
//     somebody calls spf_value_download;
//   * the outputs of one batch are ONE block (the kernels write consecutive rows as they always did; no scatter pass);
//     the block returns to the arena when the last value in it is released.  A value that outlives its batch mates
//     therefore pins their block — 288 GB of HBM buys that simplicity; spf_pool_value_stats shows live and cached bytes;
//   * blocks come from a caching arena (size classes, four per doubling): hipMalloc / hipFree are device-wide
//     synchronisation points and never run on the steady-state path; spf_pool_trim gives the cache back to the driver.
//
// Included by spf_hip.hip ahead of spf_pool.hpp.
#pragma once

#include <atomic>
#include <map>
#include <memory>
#include <mutex>

namespace spf_pool_impl {
struct Batch; // spf_pool.hpp: the batch that will produce a pending value
}

namespace spf_value_impl {

struct Arena : std::enable_shared_from_this<Arena> {
    int device = 0;
    std::mutex mu;
    std::multimap<size_t, void*> cache; // free blocks by size class
    size_t cached_bytes = 0;
    size_t cache_limit = (size_t)32 << 30; // beyond this a returned block goes back to the driver (hipFree)
    std::atomic<size_t> live_bytes{0};
    std::atomic<size_t> live_values{0};
    std::atomic<uint64_t> n_malloc{0};
    bool closed = false; // the owning pool is gone: nothing is cached any more

    // four size classes per doubling (at most a quarter wasted), 4 KiB at least
    static size_t size_class(size_t bytes)
    {
        const size_t b = std::max<size_t>(bytes, 4096);
        size_t p = 4096;
        while ((p << 1) <= b) p <<= 1;
        if (b == p) return b;
        const size_t step = p / 4;
        return (b + step - 1) / step * step;
    }

    // the calling thread's current device is preserved
    struct DeviceScope {
        int prev = -1;
        bool ok = false;
        explicit DeviceScope(int device)
        {
            if (hipGetDevice(&prev) != hipSuccess) prev = -1;
            ok = prev == device || hipSetDevice(device) == hipSuccess;
        }
        ~DeviceScope()
        {
            if (ok && prev >= 0) (void)hipSetDevice(prev);
        }
    };

    void* take(size_t cls)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = cache.find(cls);
            if (it != cache.end()) {
                void* p = it->second;
                cache.erase(it);
                cached_bytes -= cls;
                live_bytes += cls;
                return p;
            }
        }
        DeviceScope ds(device);
        if (!ds.ok) return nullptr;
        void* p = nullptr;
        if (hipMalloc(&p, cls) != hipSuccess) {
            (void)hipGetLastError();
            trim(); // the cache may hold what this allocation needs in other classes
            if (hipMalloc(&p, cls) != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
            }
        }
        n_malloc++;
        live_bytes += cls;
        return p;
    }
    void give_back(void* p, size_t cls)
    {
        live_bytes -= cls;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!closed && cached_bytes + cls <= cache_limit) {
                try {
                    cache.emplace(cls, p);
                    cached_bytes += cls;
                    return;
                } catch (const std::exception&) { // (out of host memory for the map node: free it instead)
                }
            }
        }
        DeviceScope ds(device);
        (void)hipFree(p);
    }
    // everything cached goes back to the driver (hipFree waits for the device)
    void trim()
    {
        std::multimap<size_t, void*> mine;
        {
            std::lock_guard<std::mutex> lk(mu);
            mine.swap(cache);
            cached_bytes = 0;
        }
        if (mine.empty()) return;
        DeviceScope ds(device);
        for (auto& kv : mine) {
            const hipError_t e = hipFree(kv.second);
            if (e != hipSuccess) fprintf(stderr, "libspf_hip: hipFree of a cached value block failed: %s\n", hipGetErrorString(e));
        }
    }
    void close()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            closed = true;
        }
        trim();
    }
};

struct Block {
    std::shared_ptr<Arena> arena;
    void* p = nullptr;
    size_t cls = 0;
    ~Block()
    {
        if (p) arena->give_back(p, cls);
    }
    static std::shared_ptr<Block> make(const std::shared_ptr<Arena>& a, size_t bytes)
    {
        try {
            auto b = std::make_shared<Block>();
            b->arena = a;
            b->cls = Arena::size_class(bytes);
            b->p = a->take(b->cls);
            if (!b->p) return nullptr;
            return b;
        } catch (const std::exception&) {
            return nullptr;
        }
    }
};

enum State : int { PENDING = 0, READY = 1, FAILED = 2 };

} // namespace spf_value_impl

struct spf_pool;

struct spf_value {
    std::atomic<int> refs{1};
    int kind = 0;
    size_t bytes = 0;
    int member = 0;                 // member of a group pool the value lives on (0 for a plain pool)
    spf_pool* home = nullptr;       // the (member) pool whose context owns the memory: operands of one operation share it
    std::shared_ptr<spf_value_impl::Arena> arena;
    std::shared_ptr<spf_value_impl::Block> blk; // set at upload, or when the producing batch is enqueued
    size_t off = 0;
    std::atomic<int> state{spf_value_impl::PENDING};
    // a PENDING result: the batch that produces it and its slot there (set and cleared under the pool's mutex; a later operation
    // of the same pool may take the value as an operand right away and is ordered behind that batch — spf_pool.hpp, "deferred")
    std::shared_ptr<spf_pool_impl::Batch> producer;
    uint32_t slot = 0;
    void* ptr() const { return static_cast<char*>(blk->p) + off; }
    bool ready() const { return state.load(std::memory_order_acquire) == spf_value_impl::READY; }
    void retain() { refs.fetch_add(1, std::memory_order_relaxed); }
    void release()
    {
        if (refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
            arena->live_values--;
            delete this;
        }
    }
    static spf_value* make(const std::shared_ptr<spf_value_impl::Arena>& a, spf_pool* home, int member, int kind, size_t bytes)
    {
        spf_value* v = new (std::nothrow) spf_value();
        if (!v) return nullptr;
        v->arena = a; v->home = home; v->member = member; v->kind = kind; v->bytes = bytes;
        a->live_values++;
        return v;
    }
};
