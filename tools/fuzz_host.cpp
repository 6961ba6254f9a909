// fuzz_host.cpp — mutation fuzz of the library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer (no GPU).
//
// the untrusted-bytes parsers and the graph builder had never run under a sanitizer.  GPU ASan / XNACK
// are not available on this pool, so the sanitizers cover what runs on the CPU: this program #includes the library's one
// translation unit (so that it can make a context by hand, without a device, and reach the builders that validate before
// their first HIP call) and is built with -fsanitize=address,undefined for the host side only (tools/asan_host.sh).
// Exercised: spf_ciphertext_{words,from_bincode,to_bincode} (safe_bincode::deserialize of the ciphertext newtypes),
// spf_load_compute_key_bincode's size walk (ComputeKey), spf_generate_lut, params_supported / params_generic, and
// spf_graph_add_{input,trivial,op,output} (the reference validates per task, task.rs:26-31) — every call must return a status,
// never crash, never touch memory outside its buffers (every destination is an exact-size heap allocation).
// Seeds: the reference's own malformed vector (parasol_runtime/src/safe_bincode.rs:58-66, :105-117) and valid serializations.
//
// usage: fuzz_host <cases> [seed]
#include "../spf_amd/csrc/spf_hip.hip"

#include <cinttypes>
#include <random>

namespace {

struct Rng {
    std::mt19937_64 g;
    explicit Rng(uint64_t seed) : g(seed) {}
    uint64_t operator()() { return g(); }
    size_t below(size_t n) { return n ? (size_t)(g() % n) : 0; }
};

std::vector<uint8_t> mutate(const std::vector<uint8_t>& seed, Rng& r)
{
    std::vector<uint8_t> v = seed;
    const int rounds = 1 + (int)r.below(4);
    for (int k = 0; k < rounds; k++) {
        switch (r.below(8)) {
        case 0: if (!v.empty()) v[r.below(v.size())] ^= (uint8_t)(1u << r.below(8)); break;
        case 1: if (!v.empty()) v[r.below(v.size())] = (uint8_t)r(); break;
        case 2: v.resize(r.below(v.size() + 1)); break;                        // truncate
        case 3: v.resize(v.size() + r.below(64), (uint8_t)r()); break;          // trailing bytes
        case 4: {                                                                // a length field of all ones / off by one
            const size_t at = 8 * r.below(v.size() / 8 + 1);
            if (at + 8 <= v.size()) {
                uint64_t n;
                std::memcpy(&n, &v[at], 8);
                const uint64_t choices[6] = {~(uint64_t)0, n + 1, n - 1, 0, (uint64_t)1 << 63, n << 1};
                n = choices[r.below(6)];
                std::memcpy(&v[at], &n, 8);
            }
            break;
        }
        case 5: if (v.size() >= 8) { const uint64_t n = r(); std::memcpy(&v[0], &n, 8); } break; // the first count, random
        case 6: if (!v.empty()) { const size_t a = r.below(v.size()), b = r.below(v.size()); std::swap(v[a], v[b]); } break;
        default: break;
        }
    }
    return v;
}

// a context made by hand: only the fields the host-side validators read (no device, no HIP call)
void fake_ctx(spf_ctx& c, const spf_params& p, bool generic)
{
    c.prm = p;
    c.generic = generic;
    c.log_n = 0;
    while ((1u << c.log_n) < p.polynomial_degree) c.log_n++;
}

} // namespace

int main(int argc, char** argv)
{
    if (argc > 1 && !strcmp(argv[1], "--selftest-overflow")) {
        // the sanitizer really watches the library's host code: a caller that lies about `len` makes the parser read past the
        // allocation, and the run must die with a heap-buffer-overflow report (tests/test_asan_host.py expects exactly that)
        spf_params p;
        spf_default_params(&p);
        const size_t w = spf_ciphertext_words(&p, SPF_VAL_LWE0);
        std::unique_ptr<uint8_t[]> bytes(new uint8_t[16]);
        std::memset(bytes.get(), 0, 16);
        const uint64_t n = w;
        std::memcpy(bytes.get(), &n, 8);
        std::unique_ptr<uint64_t[]> out(new uint64_t[w]);
        (void)spf_ciphertext_from_bincode(&p, SPF_VAL_LWE0, bytes.get(), 8 + 8 * w, out.get());
        printf("selftest: no report?!\n");
        return 0;
    }
    const uint64_t cases = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000;
    Rng r(argc > 2 ? strtoull(argv[2], nullptr, 10) : 0x5EEDF022);
    spf_params dflt;
    spf_default_params(&dflt);
    spf_params tiny = dflt; // a small generic set: the ComputeKey blob stays a few KiB, the walk is the same code
    tiny.lwe_dimension = 2; tiny.polynomial_degree = 16; tiny.ks_radix_count = 2; tiny.tr_radix_count = 2; tiny.ss_radix_count = 2;
    tiny.pbs_radix_log = 4; tiny.cbs_radix_count = 2;

    // seeds
    std::vector<std::vector<uint8_t>> ct_seeds;
    ct_seeds.push_back({253, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0x1, 0x2, 0x3, 0x4}); // the reference's malformed vector
    const spf_value_kind kinds[5] = {SPF_VAL_LWE0, SPF_VAL_LWE1, SPF_VAL_GLWE1, SPF_VAL_GGSW1, SPF_VAL_GLEV1};
    for (const spf_params* p : {&dflt, &tiny})
        for (spf_value_kind k : kinds) {
            const size_t w = spf_ciphertext_words(p, k);
            if (!w) continue;
            std::vector<uint64_t> words(w);
            for (auto& x : words) x = r();
            std::vector<uint8_t> out(8 + 8 * w);
            size_t written = 0;
            if (spf_ciphertext_to_bincode(p, k, words.data(), out.data(), out.size(), &written) == SPF_OK) ct_seeds.push_back(out);
        }
    spf_ctx tiny_ctx, dflt_ctx;
    fake_ctx(tiny_ctx, tiny, true);
    fake_ctx(dflt_ctx, dflt, false);
    std::vector<uint8_t> key_seed;
    {
        const size_t want[4] = {(size_t)tiny.lwe_dimension * ggsw_fft_complex(tiny, tiny.pbs_radix_count),
                                (size_t)tiny.glwe_size * tiny.polynomial_degree * tiny.ks_radix_count * lwe0_words(tiny), ssk_complex(tiny), ak_complex(tiny)};
        const size_t elem[4] = {16, 8, 16, 16};
        for (int i = 0; i < 4; i++) {
            uint64_t n = want[i];
            for (int b = 0; b < 8; b++) key_seed.push_back((uint8_t)(n >> (8 * b)));
            for (size_t j = 0; j < want[i] * elem[i]; j++) key_seed.push_back((uint8_t)r());
        }
    }

    uint64_t ok = 0, refused = 0;
    for (uint64_t it = 0; it < cases; it++) {
        const spf_params* p = r.below(2) ? &dflt : &tiny;
        switch (r.below(6)) {
        case 0: case 1: { // ciphertext wire format
            const std::vector<uint8_t> in = mutate(ct_seeds[r.below(ct_seeds.size())], r);
            // an exact-size copy on the heap: a read past `len` is a heap-buffer-overflow under ASan
            std::unique_ptr<uint8_t[]> bytes(new uint8_t[in.size() ? in.size() : 1]);
            if (!in.empty()) std::memcpy(bytes.get(), in.data(), in.size());
            const spf_value_kind k = (spf_value_kind)(r.below(7)); // (5, 6: unknown kinds)
            const size_t w = spf_ciphertext_words(p, k);
            std::unique_ptr<uint64_t[]> out(new uint64_t[w ? w : 1]);
            const spf_status st = spf_ciphertext_from_bincode(p, k, bytes.get(), in.size(), out.get());
            if (st == SPF_OK) {
                ok++;
                std::unique_ptr<uint8_t[]> back(new uint8_t[8 + 8 * w]);
                size_t written = 0;
                const size_t cap = r.below(4) ? 8 + 8 * w : r.below(8 + 8 * w + 1);
                const spf_status st2 = spf_ciphertext_to_bincode(p, k, out.get(), back.get(), cap, &written);
                if ((st2 == SPF_OK) != (cap >= 8 + 8 * w)) { fprintf(stderr, "to_bincode: capacity check is off\n"); return 1; }
                if (st2 == SPF_OK && std::memcmp(back.get(), bytes.get(), 8 + 8 * w) != 0) { fprintf(stderr, "round trip differs\n"); return 1; }
            } else {
                refused++;
            }
            break;
        }
        case 2: { // ComputeKey blob: the size walk (the loaders behind it fail at their first HIP call without a device)
            const std::vector<uint8_t> in = mutate(key_seed, r);
            std::unique_ptr<uint8_t[]> bytes(new uint8_t[in.size() ? in.size() : 1]);
            if (!in.empty()) std::memcpy(bytes.get(), in.data(), in.size());
            const spf_status st = spf_load_compute_key_bincode(&tiny_ctx, bytes.get(), in.size());
            (st == SPF_OK ? ok : refused)++;
            break;
        }
        case 3: { // generate_lut
            const uint32_t bits = (uint32_t)r.below(14);
            const size_t n_maps = r.below(5);
            const size_t n = n_maps << bits;
            std::unique_ptr<uint64_t[]> maps(new uint64_t[n ? n : 1]);
            for (size_t i = 0; i < n; i++) maps[i] = r.below(8) ? r.below((size_t)1 << bits) : r(); // mostly valid values
            const size_t w = (size_t)(p->glwe_size + 1) * p->polynomial_degree;
            std::unique_ptr<uint64_t[]> lut(new uint64_t[w]);
            const spf_status st = spf_generate_lut(p, maps.get(), n_maps, bits, lut.get());
            (st == SPF_OK ? ok : refused)++;
            break;
        }
        case 4: { // parameter validation
            spf_params q = dflt;
            uint32_t* f = reinterpret_cast<uint32_t*>(&q);
            for (int k = 0; k < 3; k++) f[r.below(sizeof(q) / 4)] = r.below(3) ? (uint32_t)r.below(70) : (uint32_t)r();
            std::string why;
            const bool a = params_supported(q, why), b = params_generic(q, why);
            (void)spf_ciphertext_words(&q, (spf_value_kind)r.below(6));
            ((a || b) ? ok : refused)++;
            break;
        }
        default: { // graph builder: random (mostly malformed) node sequences on a hand-made context; nothing is ever run
            spf_graph* g = nullptr;
            if (spf_graph_create(&dflt_ctx, &g) != SPF_OK) { refused++; break; }
            static uint64_t host_buf[4];
            const int n_calls = 1 + (int)r.below(40);
            for (int k = 0; k < n_calls; k++) {
                uint32_t node = 0;
                const uint32_t have = (uint32_t)g->nodes.size();
                spf_status st;
                switch (r.below(4)) {
                case 0: st = spf_graph_add_input(g, (spf_value_kind)r.below(7), r.below(8) ? host_buf : nullptr, &node); break;
                case 1: st = spf_graph_add_trivial(g, (spf_value_kind)r.below(7), r.below(3), &node); break;
                case 2: {
                    uint32_t in[4];
                    for (auto& x : in) x = r.below(4) ? (uint32_t)r.below(have + 2) : (uint32_t)r();
                    st = spf_graph_add_op(g, (spf_graph_op)r.below(12), r.below(10) ? in : nullptr, r.below(5), r.below(2) ? r() : r.below(5000), &node);
                    break;
                }
                default: st = spf_graph_add_output(g, r.below(4) ? (uint32_t)r.below(have + 2) : (uint32_t)r(), r.below(8) ? host_buf : nullptr); break;
                }
                (st == SPF_OK ? ok : refused)++;
            }
            // the builder's invariants: operands precede their users and carry the kind the operation wants
            for (size_t i = 0; i < g->nodes.size(); i++) {
                const auto& n = g->nodes[i];
                if (n.op < 0) continue;
                spf_graph_impl::OpInfo info{};
                if (!spf_graph_impl::op_info(n.op, &info) || n.n_in != (uint32_t)info.arity) { fprintf(stderr, "graph: bad node accepted\n"); return 1; }
                for (uint32_t j = 0; j < n.n_in; j++)
                    if (n.in[j] >= i || g->nodes[n.in[j]].kind != info.in_kind[j]) { fprintf(stderr, "graph: bad operand accepted\n"); return 1; }
            }
            g->nodes.clear(); // (nothing was planned: no device memory to give back)
            delete g;
            break;
        }
        }
    }
    printf("fuzz_host: %" PRIu64 " cases, %" PRIu64 " calls accepted, %" PRIu64 " refused, no sanitizer report\n", cases, ok, refused);
    return 0;
}
