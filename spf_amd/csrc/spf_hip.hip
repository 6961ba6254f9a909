// spf_hip.hip — host side of the C ABI declared in include/spf_hip.h.
//
// Owns the per-GPU context: device copies of the evaluation keys (reference layouts, 288 GB of
// HBM means every rank simply holds a full replica), the twiddle image, growable staging
// buffers for the host-pointer entry points, and hipEvent timing for bench.py.
// No exception leaves this file; every entry point returns spf_status.
#include "../../include/spf_hip.h"
#include "spf_kernels.hpp"
#include "spf_cbs_tail.hpp"
#include "spf_generic.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

using namespace spf;

// HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams that share a queue
// run their kernels one after the other: with the default, two bootstrap launches on two streams took 7.45 ms, with more queues
// 3.76 (tools/concurrency_probe.py; r06: sixteen staging sets, 24 queues — a 32-bit adder by handles 10.4 ms against 11.5 with 16).
// The pool keeps several batches resident on the GPU at once, each on its own stream, beside
// the streams of the context and of the caller — so the library asks for more queues, unless the environment already says
// otherwise.  The runtime reads the variable when it initialises (first HIP call of the process); this runs when the library is
// loaded.  A process that has initialised HIP before loading the library keeps its setting: export GPU_MAX_HW_QUEUES there.
__attribute__((constructor(101))) static void spf_ask_for_hw_queues() { (void)setenv("GPU_MAX_HW_QUEUES", "24", 0); }
#if defined(SPF_ABL) && SPF_ABL != 0
// a timing-only ablation build (spf_kernels.hpp, SPF_ABL) computes WRONG results on purpose: it says so when it is loaded
__attribute__((constructor(102))) static void spf_ablation_banner()
{
    fprintf(stderr, "libspf_hip: TIMING-ONLY ablation build (SPF_ABL=%d): the blind rotation's results are WRONG by construction\n", (int)SPF_ABL);
}
#endif

namespace {

thread_local std::string g_create_error;

// The three-ciphertexts-per-workgroup blind rotation for 2 x #CU < B <= 3 x #CU (r05, the 512 -> 513 step of the launch time:
// 6.95 -> 9.76 ms).  Built, bit-equal (same output checksums at B = 513 / 520 / 600 / 700 / 768, both instantiations), and no faster
// than four per workgroup — 9.90-9.97 ms against 9.74-9.82 (plain PBS 10.61 against 10.28): two of the four SIMDs still carry two
// waves, and the step of a workgroup is the step of its slowest SIMD.  Off; -DSPF_TRIO_SHAPE=1 re-runs the row
// (profiles/r05_experiments_other_kernels.md).  The alternative "512 on the two-per-workgroup shape + the rest on the eight-wave shape" runs
// back to back (both shapes take a whole CU's LDS): 6.95 + 3.72 ms.
#ifndef SPF_TRIO_SHAPE
#define SPF_TRIO_SHAPE 0
#endif
constexpr size_t kMaxGridRows = 32768; // rows per launch of the one-grid-row-per-ciphertext kernels

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct TimedLaunch {
    hipEvent_t start, stop;
};

// intermediates of the keyswitch (digits [Mpad][K], digit sums [Mpad]) and of the circuit bootstrap (lo-noise GLWE, GLEV): one set
// per stream of work that may run concurrently — the context's own for the entry points, one per staging set of a pool
struct Scratch {
    DevBuf ks_dig, ks_rowsum, cbs_glwe, cbs_glev;
};

// kernels whose launches spf_set_timing brackets with hipEvents on the launch stream (spf_last_kernel_ms)
enum TimedKernel { T_PBS = 0, T_KS, T_TRACE, T_SS, T_CMUX, T_COUNT };
const char* const kTimedNames[T_COUNT] = {"pbs", "keyswitch", "trace", "scheme_switch", "cmux"};

} // namespace

struct spf_ctx {
    spf_params prm{};
    int device = 0;
    std::recursive_mutex mu; // recursive: the host-pointer entry points hold it across the _dev calls they make
    std::string err;               // last error message; guarded by err_mu (read and written from any thread)
    mutable std::mutex err_mu;
    c64* d_tables = nullptr;
    c64* d_bsk = nullptr;        // the caller's spectra (what spf_key_blob hands out and a broadcast replicates)
    c64* d_bsk_scaled = nullptr; // the same times 2^-10: what the blind-rotation kernels read (finish_bootstrap_key)
    size_t bsk_bytes = 0;
    bool bsk_ready = false;
    uint64_t* d_ksk = nullptr;
    size_t ksk_bytes = 0;
    bool ksk_ready = false;
    uint64_t* d_cbs_lut = nullptr; // fill_multifunctional_cbs_decomposition_lut, constant per params
    DevBuf in, out, mid, aux;      // staging for the host-pointer entry points
    int8_t* d_ksk_planes = nullptr; // key byte planes for the int8-MFMA keyswitch [Npad][K]
    size_t ks_npad = 0;
    Scratch scr;                    // intermediates of the entry points (keyswitch digits, circuit-bootstrap GLWE / GLEV)
    c64* d_ak = nullptr;            // automorphism key, FFT'd: [log2 N][l_tr][2][N/2]
    size_t ak_bytes = 0;
    bool ak_ready = false;
    c64* d_ssk = nullptr;           // scheme-switch key, FFT'd: [l_ss][2][N/2]
    size_t ssk_bytes = 0;
    bool ssk_ready = false;
    double* d_ggsw_const = nullptr; // l1ggsw_zero | l1ggsw_one (Evaluation::new, evaluation.rs:161-197), built on first use
    bool ggsw_const_ready = false;  // reset whenever a key of the circuit bootstrap changes
    int n_cu = 256;                // compute units of the device (picks the blind-rotation shape)
    hipStream_t stream = nullptr;  // stream of the host-pointer entry points
    uint64_t buf_epoch = 0;            // bumped whenever a scratch buffer is reallocated (captured gate graphs hold their addresses)
    const char* last_pbs_kernel = "";  // name of the blind-rotation kernel the last launch used
    const char* last_cmux_kernel = ""; // ... and of the CMUX kernel
    hipStream_t copy_stream = nullptr; // device-to-host copies of finished slices, under the next slice's kernel
    std::vector<hipEvent_t> slice_ev;  // one "slice k is computed" event per slice in flight
    bool timing = false;
    std::vector<TimedLaunch> timed[T_COUNT];
    // any other parameter set (spf_generic.hpp): N a power of two in 16 .. 2048, any k and radix that fit the LDS.  One workgroup per ciphertext,
    // the oracle's transform for those sizes; every batch and device-pointer entry point is served (keyswitch in its VALU form);
    // gate graphs are not.
    bool generic = false;
    uint32_t log_n = 11;
    c64* d_gen_tables = nullptr; // [N/2] twist, then [N/4] transform twiddles
};

namespace {

spf_status fail(spf_ctx* c, spf_status s, const std::string& msg)
{
    if (c) {
        std::lock_guard<std::mutex> g(c->err_mu);
        c->err = msg;
    } else {
        g_create_error = msg;
    }
    return s;
}

#define HIPCHK(ctx, expr)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(ctx, SPF_ERR_HIP,                                                         \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                       \
    } while (0)

// e^{+2 pi i num/den}: first-octant cosl/sinl, rounded once to double, mirrored by octant so
// conjugate / quarter-turn partners are exact.  (The build's definition of its twiddles.)
c64 root_of_unity(uint64_t num, uint64_t den)
{
    static const long double TWO_PI = 6.283185307179586476925286766559005768L;
    uint64_t j = num % den, eighth = den / 8, oct = j / eighth, r = j % eighth;
    uint64_t rr = (oct & 1) ? (eighth - r) : r;
    long double th = TWO_PI * (long double)rr / (long double)den;
    double c = (double)cosl(th), s = (double)sinl(th);
    if (rr == 0) { c = 1.0; s = 0.0; }
    switch (oct) {
    case 0: return {c, s};
    case 1: return {s, c};
    case 2: return {-s, c};
    case 3: return {-c, s};
    case 4: return {-c, -s};
    case 5: return {-s, -c};
    case 6: return {s, -c};
    default: return {c, -s};
    }
}

c64 conj(c64 a) { return {a.re, -a.im}; }

void build_tables(std::vector<c64>& t)
{
    t.resize(kTableEntries);
    for (int k1 = 1; k1 < 8; k1++)
        for (int lane = 0; lane < 64; lane++)
            t[kT1Off + (k1 - 1) * 64 + lane] = conj(root_of_unity((uint64_t)(lane * k1), 512));
    for (int c = 1; c < 8; c++)
        for (int b = 0; b < 8; b++) t[kT2Off + (c - 1) * 8 + b] = conj(root_of_unity((uint64_t)(b * c), 64));
    for (int k = 0; k < 512; k++) t[kWCOff + k] = conj(root_of_unity((uint64_t)k, 1024));
    // negacyclic twist e^{+2 pi i j / (2N)}, j = 2n' + par (math/fft/negacyclic/mod.rs:56-65)
    for (int par = 0; par < 2; par++)
        for (int n = 0; n < 512; n++) t[kTWOff + par * 512 + n] = root_of_unity((uint64_t)(2 * n + par), 4096);
}

uint32_t ceil_log2(uint32_t v)
{
    uint32_t l = 0;
    while ((1u << l) < v) l++;
    return l;
}

spf_status ensure(spf_ctx* c, DevBuf& b, size_t bytes)
{
    if (b.cap >= bytes) return SPF_OK;
    c->buf_epoch++;
    if (b.p) HIPCHK(c, hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    HIPCHK(c, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    return SPF_OK;
}

size_t lwe0_words(const spf_params& p) { return (size_t)p.lwe_dimension + 1; }
size_t lwe1_words(const spf_params& p) { return (size_t)p.glwe_size * p.polynomial_degree + 1; }
size_t glwe_words(const spf_params& p) { return (size_t)(p.glwe_size + 1) * p.polynomial_degree; }
size_t ggsw_fft_complex(const spf_params& p, uint32_t count)
{
    return (size_t)(p.glwe_size + 1) * count * (p.glwe_size + 1) * (p.polynomial_degree / 2);
}

size_t ak_complex(const spf_params& p)
{
    uint32_t logn = 0;
    while ((1u << logn) < p.polynomial_degree) logn++;
    return (size_t)logn * p.glwe_size * p.tr_radix_count * (p.glwe_size + 1) * (p.polynomial_degree / 2);
}
size_t ssk_complex(const spf_params& p)
{
    return (size_t)(p.glwe_size * (p.glwe_size + 1) / 2) * p.ss_radix_count * (p.glwe_size + 1) * (p.polynomial_degree / 2);
}

spf_status get_events(spf_ctx* c, hipEvent_t* a, hipEvent_t* b)
{
    HIPCHK(c, hipEventCreate(a));
    HIPCHK(c, hipEventCreate(b));
    return SPF_OK;
}

// hipEvent bracket around the launches of one entry point, when timing is on
struct TimedScope {
    spf_ctx* c; hipStream_t s; int which; TimedLaunch tl{}; bool on = false;
    TimedScope(spf_ctx* c_, hipStream_t s_, int which_) : c(c_), s(s_), which(which_) {}
    spf_status begin()
    {
        if (!c->timing) return SPF_OK;
        spf_status st = get_events(c, &tl.start, &tl.stop);
        if (st != SPF_OK) return st;
        on = true;
        HIPCHK(c, hipEventRecord(tl.start, s));
        return SPF_OK;
    }
    spf_status end()
    {
        if (!on) return SPF_OK;
        on = false;
        hipError_t e = hipEventRecord(tl.stop, s);
        if (e != hipSuccess) {
            (void)hipEventDestroy(tl.start); (void)hipEventDestroy(tl.stop);
            return fail(c, SPF_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(e));
        }
        c->timed[which].push_back(tl);
        return SPF_OK;
    }
    ~TimedScope() // a launch failed between begin() and end(): the pair goes back instead of leaking
    {
        if (on) { (void)hipEventDestroy(tl.start); (void)hipEventDestroy(tl.stop); }
    }
};

GenericShape generic_shape(const spf_ctx* c)
{
    GenericShape g{};
    g.N = c->prm.polynomial_degree; g.logN = c->log_n; g.k = c->prm.glwe_size;
    g.twist = c->d_gen_tables;
    g.w = g.N == (uint32_t)kN ? c->d_tables : c->d_gen_tables + g.N / 2; // N = 2048: DAG-I reads the tuned kernels' table image
    return g;
}

spf_status launch_blind_rotate(spf_ctx* c, hipStream_t s, size_t B, const uint64_t* d_lwe,
                               const uint64_t* d_lut, size_t lut_stride, uint32_t log_chi,
                               uint32_t log_v, uint64_t body_rotate, uint64_t* d_out,
                               size_t out_stride, bool extract, int per_wg_hint = 0)
{
    if (!c->bsk_ready) return fail(c, SPF_ERR_NO_KEY, "bootstrap key not loaded");
    if (B == 0) return SPF_OK;
    if (B > 0x7fffffffu) return fail(c, SPF_ERR_INVALID_ARGUMENT, "batch too large");
    // modulus switch needs log_modulus - log_v >= 1 and shifts below 64
    if (log_v >= c->log_n + 1 || log_chi >= 52) return fail(c, SPF_ERR_INVALID_ARGUMENT, "log_v / log_chi out of range");
    if (c->generic) {
        GenericPbsArgs ga{};
        ga.g = generic_shape(c);
        ga.lwe_in = d_lwe; ga.lut = d_lut; ga.lut_stride = lut_stride; ga.bsk = c->d_bsk; ga.out = d_out; ga.out_stride = out_stride;
        ga.n = c->prm.lwe_dimension; ga.B = (uint32_t)B; ga.radix_log = c->prm.pbs_radix_log; ga.count = c->prm.pbs_radix_count;
        ga.log_chi = log_chi; ga.log_v = log_v; ga.sample_extract = extract ? 1u : 0u; ga.body_rotate = body_rotate;
        c->last_pbs_kernel = "generic_pbs_kernel";
        TimedScope tsg(c, s, T_PBS);
        spf_status stg = tsg.begin();
        if (stg != SPF_OK) return stg;
        hipLaunchKernelGGL(generic_pbs_kernel, dim3((unsigned)B), dim3(kGenericThreads),
                           generic_lds_bytes(ga.g.N, ga.g.k, true), s, ga);
        HIPCHK(c, hipGetLastError());
        return tsg.end();
    }
    BlindRotateArgs a{};
    a.lwe_in = d_lwe; a.lut = d_lut; a.lut_stride = lut_stride; a.bsk = SPF_BSK_PRESCALED ? c->d_bsk_scaled : c->d_bsk;
    a.tables = c->d_tables; a.out = d_out; a.out_stride = out_stride;
    a.n = c->prm.lwe_dimension; a.B = (uint32_t)B; a.log_chi = log_chi; a.log_v = log_v;
    a.body_rotate = body_rotate; a.sample_extract = extract ? 1u : 0u;
    // Shape by batch size: at most one ciphertext per CU -> eight waves per ciphertext (blind_rotate8_kernel, latency);
    // up to two per CU -> the paired schedule with two ciphertexts per workgroup (blind_rotate2p2_kernel);
    // beyond -> four ciphertexts per workgroup, one workgroup per CU, two waves per SIMD (blind_rotate2p_kernel).
    // per_wg_hint (the pool): ciphertexts per workgroup the CALLER wants at least — a batch that shares the chip with other
    // batches in flight takes the shape of the whole population, so that the batches tile the CUs instead of each spreading thin.
    const size_t n_cu = (size_t)c->n_cu;
    const bool quad = B <= n_cu && per_wg_hint <= 1;
    const bool pair2 = !quad && B <= 2 * n_cu && per_wg_hint <= 2;
    const bool trio = !quad && !pair2 && B <= 3 * n_cu && per_wg_hint <= 3 && SPF_TRIO_SHAPE; // three per workgroup: fills the chip up to 3 x #CU
    const size_t per_wg = quad ? 1 : (pair2 ? 2 : (trio ? 3 : 4));
    dim3 grid((unsigned)((B + per_wg - 1) / per_wg)), block(pair2 ? 256 : (trio ? 384 : 512));
    TimedScope ts(c, s, T_PBS);
    {
        spf_status st = ts.begin();
        if (st != SPF_OK) return st;
    }
#ifdef SPF_STAMPS
    static uint64_t* d_stamps = nullptr;
    const size_t stamp_waves = 8;
    const size_t n_stamp = (size_t)grid.x * stamp_waves * 16;
    if (!pair2) {
        if (d_stamps) (void)hipFree(d_stamps);
        HIPCHK(c, hipMalloc(&d_stamps, n_stamp * 8));
        HIPCHK(c, hipMemsetAsync(d_stamps, 0, n_stamp * 8, s));
        a.stamps = d_stamps;
    }
#endif
#define SPF_STR2(x) #x
#define SPF_STR(x) SPF_STR2(x)
#define SPF_LAUNCH(NAME, KERNEL, LDS) do { c->last_pbs_kernel = NAME; hipLaunchKernelGGL(KERNEL, grid, block, LDS, s, a); } while (0)
    // log_v >= 1 makes every rotation amount even: the kernels then skip the hand-overs around the rotation gather (",even")
    if (quad && log_v == 0) SPF_LAUNCH("blind_rotate8_kernel<2,16>", (blind_rotate8_kernel<2, 16, 1>), kBlindRotate8Lds);
    else if (quad) SPF_LAUNCH("blind_rotate8_kernel<2,16,even>", (blind_rotate8_kernel<2, 16, 0>), kBlindRotate8Lds);
    else if (pair2 && log_v == 0) SPF_LAUNCH("blind_rotate2p2_kernel<2,16," SPF_STR(SPF_BR2_OPT) ">", (blind_rotate2p2_kernel<2, 16, SPF_BR2_OPT, 1>), kBlindRotate2p2Lds);
    else if (pair2) SPF_LAUNCH("blind_rotate2p2_kernel<2,16," SPF_STR(SPF_BR2_OPT) ",even>", (blind_rotate2p2_kernel<2, 16, SPF_BR2_OPT, 0>), kBlindRotate2p2Lds);
#if SPF_TRIO_SHAPE
    else if (trio && log_v == 0) SPF_LAUNCH("blind_rotate2p3_kernel<2,16," SPF_STR(SPF_BR_OPT) ">", (blind_rotate2p3_kernel<2, 16, SPF_BR_OPT, 1>), kBlindRotate2p3Lds);
    else if (trio) SPF_LAUNCH("blind_rotate2p3_kernel<2,16," SPF_STR(SPF_BR_OPT) ",even>", (blind_rotate2p3_kernel<2, 16, SPF_BR_OPT, 0>), kBlindRotate2p3Lds);
#endif
    else if (log_v == 0) SPF_LAUNCH("blind_rotate2p_kernel<2,16," SPF_STR(SPF_BR_OPT_MIX) ">", (blind_rotate2p_kernel<2, 16, SPF_BR_OPT_MIX, 1>), kBlindRotate2pLds);
    else SPF_LAUNCH("blind_rotate2p_kernel<2,16," SPF_STR(SPF_BR_OPT) ",even>", (blind_rotate2p_kernel<2, 16, SPF_BR_OPT, 0>), kBlindRotate2pLds);
#undef SPF_LAUNCH
    HIPCHK(c, hipGetLastError());
    {
        spf_status st = ts.end();
        if (st != SPF_OK) return st;
    }
#ifdef SPF_STAMPS
    if (a.stamps) { // diagnostic build: median over waves of the per-phase cycle sums, per CMUX step
        HIPCHK(c, hipStreamSynchronize(s));
        std::vector<uint64_t> h(n_stamp);
        HIPCHK(c, hipMemcpy(h.data(), a.stamps, n_stamp * 8, hipMemcpyDeviceToHost));
        static const char* names[12] = {"stage+rendezvous", "gather+decomp+twist", "rendezvous(gathered) + key-row wait", "fwd transform pair",
            "cross write+key barrier", "cross read+combine", "MAD x2", "ring barrier", "inverse cross exchange",
            "inv transform pair", "untwist+convert+acc", "step head"};
        static const char* names8[12] = {"step head+stage", "barrier A (j=1: A+F)", "gather+decomp+post+F+twist", "fwd transform", "cross write+barrier B",
            "cross read+combine+publish", "barrier C", "MAD+inverse split+post", "barrier D", "inbox read", "inv transform", "untwist+convert+acc"};
        double total = 0;
        std::vector<double> med(12), med_old(12), med_young(12);
        for (int i = 0; i < 12; i++) {
            std::vector<uint64_t> v, vo, vy;
            for (size_t wv = 0; wv < (size_t)grid.x * stamp_waves; wv++) {
                v.push_back(h[wv * 16 + i]);
                ((wv % stamp_waves) < stamp_waves / 2 ? vo : vy).push_back(h[wv * 16 + i]);
            }
            std::sort(v.begin(), v.end());
            std::sort(vo.begin(), vo.end());
            std::sort(vy.begin(), vy.end());
            med[i] = (double)v[v.size() / 2] / a.n;
            med_old[i] = (double)vo[vo.size() / 2] / a.n;
            med_young[i] = (double)vy[vy.size() / 2] / a.n;
            total += med[i];
        }
        fprintf(stderr, "[stamps] per CMUX step, median over %zu waves (cycles, share | waves 0-3 | waves 4-7)\n", (size_t)grid.x * stamp_waves);
        for (int i = 0; i < 12; i++)
            fprintf(stderr, "[stamps] %-30s %8.0f %5.1f%% | %8.0f | %8.0f\n", (quad ? names8 : names)[i], med[i], 100.0 * med[i] / total, med_old[i], med_young[i]);
        fprintf(stderr, "[stamps] %-26s %8.0f\n", "total", total);
    }
#endif
    return SPF_OK;
}

// the int8-MFMA formulation applies when digits fit int8, K is a multiple of the MFMA depth and
// the int32 accumulators cannot overflow
bool ks_mfma_ok(const spf_params& p)
{
    const uint64_t K = (uint64_t)p.glwe_size * p.polynomial_degree * p.ks_radix_count;
    return p.ks_radix_log <= 8 && K % 256 == 0 && K * ((uint64_t)1 << (p.ks_radix_log - 1)) * 128 < ((uint64_t)1 << 31);
}

// (re)build the byte-plane image of the keyswitch key; called whenever the key becomes ready
spf_status build_ks_planes(spf_ctx* c)
{
    if (c->generic || !ks_mfma_ok(c->prm)) return SPF_OK;
    const uint32_t n_in = c->prm.glwe_size * c->prm.polynomial_degree, w = c->prm.lwe_dimension + 1;
    const size_t K = (size_t)n_in * c->prm.ks_radix_count;
    const size_t npad = ((size_t)w * 8 + KSG_TILE - 1) / KSG_TILE * KSG_TILE;
    if (!c->d_ksk_planes) HIPCHK(c, hipMalloc((void**)&c->d_ksk_planes, npad * K));
    c->ks_npad = npad;
    HIPCHK(c, hipMemsetAsync(c->d_ksk_planes, 0, npad * K, c->stream));
    const size_t total = (size_t)n_in * c->prm.ks_radix_count * w;
    hipLaunchKernelGGL(ks_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, c->d_ksk,
                       c->d_ksk_planes, n_in, w, c->prm.ks_radix_count, K);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPF_OK;
}

spf_status launch_keyswitch(spf_ctx* c, hipStream_t s, size_t B, const uint64_t* d_in,
                            uint64_t* d_out, Scratch* sc = nullptr)
{
    if (!sc) sc = &c->scr;
    if (!c->ksk_ready) return fail(c, SPF_ERR_NO_KEY, "keyswitch key not loaded");
    if (B == 0) return SPF_OK;
    if (B > 0x7fffffffu) return fail(c, SPF_ERR_INVALID_ARGUMENT, "batch too large");
    KeyswitchArgs a{};
    a.in = d_in; a.ksk = c->d_ksk; a.out = d_out;
    a.n_in = c->prm.glwe_size * c->prm.polynomial_degree; a.n_out = c->prm.lwe_dimension;
    a.B = (uint32_t)B; a.radix_log = c->prm.ks_radix_log; a.count = c->prm.ks_radix_count;
    const bool mfma = c->d_ksk_planes != nullptr; // (null when the radix does not fit the int8 formulation: keyswitch_kernel)
    const size_t K = (size_t)a.n_in * a.count, mpad = (B + KSG_TILE - 1) / KSG_TILE * KSG_TILE;
    if (mfma) {
        spf_status st = ensure(c, sc->ks_dig, mpad * K);
        if (st != SPF_OK) return st;
        st = ensure(c, sc->ks_rowsum, mpad * sizeof(int));
        if (st != SPF_OK) return st;
    }
    TimedScope ts(c, s, T_KS);
    {
        spf_status st = ts.begin();
        if (st != SPF_OK) return st;
    }
    if (mfma) {
        HIPCHK(c, hipMemsetAsync(sc->ks_rowsum.p, 0, mpad * sizeof(int), s));
        hipLaunchKernelGGL(ks_digits_kernel, dim3((unsigned)B), dim3(256), 0, s, d_in, (int8_t*)sc->ks_dig.p,
                           (int*)sc->ks_rowsum.p, a.n_in, a.B, a.radix_log, a.count);
        KsGemmArgs g{};
        g.A = (const int8_t*)sc->ks_dig.p; g.Bt = c->d_ksk_planes; g.rowsum = (const int*)sc->ks_rowsum.p;
        g.in = d_in; g.out = d_out; g.B = a.B; g.n_in = a.n_in; g.n_out = a.n_out; g.K = (uint32_t)K;
        dim3 grid((unsigned)(c->ks_npad / KSG_TILE), (unsigned)(mpad / KSG_TILE));
        hipLaunchKernelGGL(ks_gemm_lds_kernel, grid, dim3(256), kKsLdsBytes, s, g); // operand tiles staged through LDS by LDS-DMA
    } else {
        dim3 grid((a.n_out + 1 + 255) / 256, (unsigned)((B + KS_CT - 1) / KS_CT)), block(256);
        hipLaunchKernelGGL(keyswitch_kernel, grid, block, 0, s, a);
    }
    HIPCHK(c, hipGetLastError());
    return ts.end();
}

// the parameter sets of the generic kernels (spf_generic.hpp)
bool params_generic(const spf_params& p, std::string& why)
{
    const uint32_t N = p.polynomial_degree;
    if (N < 16 || N > 2048 || (N & (N - 1))) {
        why = "polynomial_degree must be a power of two in 16 .. 2048";
        return false;
    }
    if (p.glwe_size == 0 || p.glwe_size > 8) { why = "glwe_size must be in 1 .. 8"; return false; }
    auto radix_ok = [](uint32_t lg, uint32_t cnt) { return lg >= 1 && cnt >= 1 && lg * cnt < 64; };
    if (!radix_ok(p.pbs_radix_log, p.pbs_radix_count) || !radix_ok(p.cbs_radix_log, p.cbs_radix_count)) {
        why = "a radix decomposition needs 1 <= radix_log * count < 64";
        return false;
    }
    if (generic_trace_lds_bytes(N, p.glwe_size) > 160 * 1024) { why = "(k+1) polynomials of this degree do not fit the generic kernels' LDS"; return false; }
    if (!radix_ok(p.tr_radix_log, p.tr_radix_count) || !radix_ok(p.ss_radix_log, p.ss_radix_count)) {
        why = "a radix decomposition needs 1 <= radix_log * count < 64";
        return false;
    }
    if (p.lwe_dimension == 0 || p.lwe_dimension > 4096) { why = "lwe_dimension out of range"; return false; }
    if (p.ks_radix_log == 0 || p.ks_radix_log * p.ks_radix_count > 32) { why = "ks_radix must satisfy 0 < l*logB <= 32"; return false; }
    if (p.cbs_radix_count >= 8) { why = "cbs_radix.count must be in 1..7"; return false; }
    return true;
}

bool params_supported(const spf_params& p, std::string& why)
{
    if (p.polynomial_degree != kN) { why = "kernels are built for polynomial_degree 2048"; return false; }
    if (p.glwe_size != 1) { why = "kernels are built for glwe_size 1"; return false; }
    if (p.pbs_radix_log != 16 || p.pbs_radix_count != 2) { why = "blind rotation is built for pbs_radix 2 x 16 bits"; return false; }
    if (p.lwe_dimension == 0 || p.lwe_dimension > 4096) { why = "lwe_dimension out of range"; return false; }
    if (p.ks_radix_log == 0 || p.ks_radix_log * p.ks_radix_count > 32) { why = "ks_radix must satisfy 0 < l*logB <= 32"; return false; }
    if (p.cbs_radix_count == 0 || p.cbs_radix_count >= 8 || p.cbs_radix_log == 0) { why = "cbs_radix.count must be in 1..7"; return false; }
    return true;
}

} // namespace

extern "C" {

void spf_default_params(spf_params* o)
{
    if (!o) return;
    *o = spf_params{637, 2048, 1, 16, 2, 4, 4, 2, 6, 7, 6, 3, 15};
}

// the build's identity: version, target, the compile-time options of the blind rotation as built — and, first of all, whether
// this is a TIMING-ONLY ablation build whose results are wrong by construction (SPF_ABL, spf_kernels.hpp)
#define SPF_VSTR2(x) #x
#define SPF_VSTR(x) SPF_VSTR2(x)
const char* spf_version(void)
{
    return "spf_hip 0.6 gfx950"
#if defined(SPF_ABL) && SPF_ABL != 0
           " ABLATION(" SPF_VSTR(SPF_ABL) ": timing only, results are WRONG)"
#endif
           " (blind rotation: BR_OPT=" SPF_VSTR(SPF_BR_OPT) " BR_OPT_MIX=" SPF_VSTR(SPF_BR_OPT_MIX) " BR2_OPT=" SPF_VSTR(SPF_BR2_OPT)
           " BR_NEG=" SPF_VSTR(SPF_BR_NEG) " BSK_PRESCALED=" SPF_VSTR(SPF_BSK_PRESCALED) " TRIO_SHAPE=" SPF_VSTR(SPF_TRIO_SHAPE)
#ifdef SPF_STAMPS
           " STAMPS"
#endif
#ifdef SPF_POOL_TRACE
           " POOL_TRACE"
#endif
           "; int8-MFMA keyswitch; cbs_radix CMUX; values by handle)";
}

// The text is copied into storage of the calling thread, so the pointer stays valid while other
// threads keep using (and failing on) the same context.
const char* spf_last_error(const spf_ctx* ctx)
{
    thread_local std::string copy;
    if (!ctx) return g_create_error.c_str();
    std::lock_guard<std::mutex> g(ctx->err_mu);
    copy = ctx->err;
    return copy.c_str();
}

spf_status spf_create(const spf_params* params, int device_id, spf_ctx** out)
{
    if (!params || !out) return fail(nullptr, SPF_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    std::string why, why_generic;
    const bool specialised = params_supported(*params, why);
    if (!specialised && !params_generic(*params, why_generic))
        return fail(nullptr, SPF_ERR_UNSUPPORTED, why_generic);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, SPF_ERR_HIP, "no HIP device visible: the HIP path is the only path, there is no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, SPF_ERR_INVALID_ARGUMENT, "device_id out of range");
    spf_ctx* c = new (std::nothrow) spf_ctx();
    if (!c) return fail(nullptr, SPF_ERR_HIP, "out of host memory");
    c->prm = *params;
    c->device = device_id;
    c->generic = !specialised;
    c->log_n = ceil_log2(params->polynomial_degree);
    auto bail = [&](spf_status s) { g_create_error = c->err; spf_destroy(c); return s; };
#define CK(expr)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            c->err = std::string(#expr) + ": " + hipGetErrorString(e_);                           \
            return bail(SPF_ERR_HIP);                                                             \
        }                                                                                         \
    } while (0)
    CK(hipSetDevice(device_id));
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    {
        int cu = 0;
        CK(hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device_id));
        if (cu > 0) c->n_cu = cu;
    }
    std::vector<c64> t;
    build_tables(t);
    CK(hipMalloc((void**)&c->d_tables, kTableBytes));
    CK(hipMemcpy(c->d_tables, t.data(), kTableBytes, hipMemcpyHostToDevice));
    // fill_multifunctional_cbs_decomposition_lut (circuit_bootstrapping.rs:430-482)
    {
        std::vector<uint64_t> lut(glwe_words(*params), 0);
        uint64_t levels[16] = {0};
        for (uint32_t i = 0; i < 16; i++) {
            uint32_t lvl = i + 1;
            if (lvl * params->cbs_radix_log + 1 < 64) {
                uint32_t bits = params->cbs_radix_log * lvl + 1;
                uint64_t minus_one = ((uint64_t)1 << bits) - 1;
                levels[i] = minus_one << (64 - bits);
            }
        }
        uint32_t v = 1u << ceil_log2(params->cbs_radix_count);
        uint64_t* b = lut.data() + (size_t)params->glwe_size * params->polynomial_degree;
        for (uint32_t i = 0; i < params->polynomial_degree; i++) {
            uint32_t fn = i % v;
            b[i] = fn < params->cbs_radix_count ? levels[fn] : 0;
        }
        CK(hipMalloc((void**)&c->d_cbs_lut, lut.size() * 8));
        CK(hipMemcpy(c->d_cbs_lut, lut.data(), lut.size() * 8, hipMemcpyHostToDevice));
    }
    if (c->generic) {
        // twist e^{+2 pi i j / (2N)} and transform twiddles e^{+2 pi i j / (N/2)}: the oracle's definitions for N != 2048
        const uint32_t N = params->polynomial_degree, h = N / 2;
        std::vector<c64> gt(h + h / 2);
        for (uint32_t j = 0; j < h; j++) gt[j] = root_of_unity(j, 2 * (uint64_t)N);
        for (uint32_t j = 0; j < h / 2; j++) gt[h + j] = root_of_unity(j, h);
        CK(hipMalloc((void**)&c->d_gen_tables, gt.size() * sizeof(c64)));
        CK(hipMemcpy(c->d_gen_tables, gt.data(), gt.size() * sizeof(c64), hipMemcpyHostToDevice));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&generic_pbs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)generic_lds_bytes(N, params->glwe_size, true)));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&generic_cmux_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)generic_lds_bytes(N, params->glwe_size, false)));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&generic_trace_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)generic_trace_lds_bytes(N, params->glwe_size)));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&generic_scheme_switch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)generic_lds_bytes(N, params->glwe_size, false)));
        *out = c;
        return SPF_OK;
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ks_gemm_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                           kKsLdsBytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&blind_rotate2p_kernel<2, 16, SPF_BR_OPT_MIX, 1>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, kBlindRotate2pLds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&blind_rotate2p_kernel<2, 16, SPF_BR_OPT, 0>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, kBlindRotate2pLds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&blind_rotate2p2_kernel<2, 16, SPF_BR2_OPT, 1>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, kBlindRotate2p2Lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&blind_rotate2p2_kernel<2, 16, SPF_BR2_OPT, 0>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, kBlindRotate2p2Lds));
#if SPF_TRIO_SHAPE
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&blind_rotate2p3_kernel<2, 16, SPF_BR_OPT, 1>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, kBlindRotate2p3Lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&blind_rotate2p3_kernel<2, 16, SPF_BR_OPT, 0>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, kBlindRotate2p3Lds));
#endif
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&blind_rotate8_kernel<2, 16, 1>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, kBlindRotate8Lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&blind_rotate8_kernel<2, 16, 0>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, kBlindRotate8Lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cmux_kernel<4, 4, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                           cmux_lds_bytes(2)));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cmux_kernel<4, 4, 2>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, cmux_lds_bytes(2)));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cmux4_kernel<4, 4>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, kCmux4Lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cbs_trace_kernel<6, 7>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, kTraceLds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&scheme_switch_kernel<15, 3>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, kTraceLds));
#undef CK
    *out = c;
    return SPF_OK;
}

void spf_destroy(spf_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& v : c->timed)
        for (auto& t : v) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    for (void* p : {(void*)c->d_tables, (void*)c->d_bsk, (void*)c->d_bsk_scaled, (void*)c->d_ksk, (void*)c->d_cbs_lut,
                    c->in.p, c->out.p, c->mid.p, c->aux.p, (void*)c->d_ksk_planes, c->scr.ks_dig.p,
                    c->scr.ks_rowsum.p, (void*)c->d_ak, (void*)c->d_ssk, c->scr.cbs_glwe.p, c->scr.cbs_glev.p, (void*)c->d_gen_tables})
        if (p) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->d_ggsw_const) (void)hipFree(c->d_ggsw_const);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (hipEvent_t e : c->slice_ev) (void)hipEventDestroy(e);
    delete c;
}

// The blind-rotation kernels read a copy of the bootstrap key scaled by 2^-10 (the 1/N of the inverse transform travels with
// the key through the multiply-accumulate: exact, and 32 multiplications per polynomial and step are not executed).  Built
// whenever the caller's image is complete; a value no forward transform of a torus polynomial produces (NaN, non-zero magnitude
// outside [2^-900, 2^1000)), where scaling first could round differently from scaling last, makes the key unusable instead.
static spf_status finish_bootstrap_key(spf_ctx* c)
{
    if (c->generic) return SPF_OK; // (the generic kernels read the caller's spectra as they are)
#if SPF_BSK_PRESCALED
    c->bsk_ready = false;
    const size_t n_complex = (size_t)c->prm.lwe_dimension * ggsw_fft_complex(c->prm, c->prm.pbs_radix_count);
    if (!c->d_bsk_scaled) HIPCHK(c, hipMalloc((void**)&c->d_bsk_scaled, n_complex * sizeof(c64)));
    spf_status st = ensure(c, c->aux, sizeof(unsigned int));
    if (st != SPF_OK) return st;
    unsigned int* d_bad = (unsigned int*)c->aux.p;
    HIPCHK(c, hipMemsetAsync(d_bad, 0, sizeof(unsigned int), c->stream));
    hipLaunchKernelGGL(scale_bootstrap_key_kernel, dim3(4 * (unsigned)c->n_cu), dim3(256), 0, c->stream,
                       (const double*)c->d_bsk, (double*)c->d_bsk_scaled, 2 * n_complex, d_bad);
    HIPCHK(c, hipGetLastError());
    unsigned int bad = 0;
    HIPCHK(c, hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (bad)
        return fail(c, SPF_ERR_INVALID_ARGUMENT,
                    "bootstrap key holds a value that is no forward transform of a torus polynomial (NaN, or a non-zero magnitude outside [2^-900, 2^1000))");
#endif
    return SPF_OK;
}

spf_status spf_key_blob(spf_ctx* c, int which, void** dev_ptr, size_t* bytes)
{
    if (!c || !dev_ptr || !bytes) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (which == 0) {
        size_t need = (size_t)c->prm.lwe_dimension * ggsw_fft_complex(c->prm, c->prm.pbs_radix_count) * sizeof(c64);
        if (!c->d_bsk) { HIPCHK(c, hipMalloc((void**)&c->d_bsk, need)); c->bsk_bytes = need; }
        *dev_ptr = c->d_bsk; *bytes = c->bsk_bytes;
    } else if (which == 1) {
        size_t need = (size_t)c->prm.glwe_size * c->prm.polynomial_degree * c->prm.ks_radix_count * lwe0_words(c->prm) * 8;
        if (!c->d_ksk) { HIPCHK(c, hipMalloc((void**)&c->d_ksk, need)); c->ksk_bytes = need; }
        *dev_ptr = c->d_ksk; *bytes = c->ksk_bytes;
    } else if (which == 2) {
        size_t need = ak_complex(c->prm) * sizeof(c64);
        if (!c->d_ak) { HIPCHK(c, hipMalloc((void**)&c->d_ak, need)); c->ak_bytes = need; }
        *dev_ptr = c->d_ak; *bytes = c->ak_bytes;
    } else if (which == 3) {
        size_t need = ssk_complex(c->prm) * sizeof(c64);
        if (!c->d_ssk) { HIPCHK(c, hipMalloc((void**)&c->d_ssk, need)); c->ssk_bytes = need; }
        *dev_ptr = c->d_ssk; *bytes = c->ssk_bytes;
    } else {
        return fail(c, SPF_ERR_INVALID_ARGUMENT, "which must be 0 (bootstrap), 1 (keyswitch), 2 (automorphism) or 3 (scheme switch)");
    }
    return SPF_OK;
}

spf_status spf_key_blob_commit(spf_ctx* c, int which)
{
    if (!c) return fail(nullptr, SPF_ERR_INVALID_ARGUMENT, "null context");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device)); // the byte planes are allocated and built on THIS context's GPU
    // Whatever filled the blob (a copy on the caller's stream, an RCCL broadcast on its own) must have landed before the derived
    // images are built from it on this context's stream, which is ordered with no other stream: once per key load.
    HIPCHK(c, hipDeviceSynchronize());
    if (which == 0 && c->d_bsk) {
        spf_status st = finish_bootstrap_key(c);
        if (st != SPF_OK) return st;
        c->bsk_ready = true, c->ggsw_const_ready = false;
    }
    else if (which == 1 && c->d_ksk) {
        spf_status st = build_ks_planes(c);
        if (st != SPF_OK) return st;
        c->ksk_ready = true;
    }
    else if (which == 2 && c->d_ak) c->ak_ready = true, c->ggsw_const_ready = false;
    else if (which == 3 && c->d_ssk) c->ssk_ready = true, c->ggsw_const_ready = false;
    else return fail(c, SPF_ERR_INVALID_ARGUMENT, "blob not allocated");
    return SPF_OK;
}

spf_status spf_load_bootstrap_key(spf_ctx* c, const double* bsk_fft, size_t n_complex)
{
    if (!c || !bsk_fft) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    size_t want = (size_t)c->prm.lwe_dimension * ggsw_fft_complex(c->prm, c->prm.pbs_radix_count);
    if (n_complex != want)
        return fail(c, SPF_ERR_INVALID_ARGUMENT, "bootstrap key length " + std::to_string(n_complex) + " != " + std::to_string(want));
    void* p; size_t bytes;
    spf_status s = spf_key_blob(c, 0, &p, &bytes);
    if (s != SPF_OK) return s;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    c->bsk_ready = false;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(p, bsk_fft, bytes, hipMemcpyHostToDevice));
    {
        spf_status st = finish_bootstrap_key(c);
        if (st != SPF_OK) return st;
    }
    c->bsk_ready = true, c->ggsw_const_ready = false;
    return SPF_OK;
}

spf_status spf_load_keyswitch_key(spf_ctx* c, const uint64_t* ksk, size_t n_words)
{
    if (!c || !ksk) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    size_t want = (size_t)c->prm.glwe_size * c->prm.polynomial_degree * c->prm.ks_radix_count * lwe0_words(c->prm);
    if (n_words != want)
        return fail(c, SPF_ERR_INVALID_ARGUMENT, "keyswitch key length " + std::to_string(n_words) + " != " + std::to_string(want));
    void* p; size_t bytes;
    spf_status s = spf_key_blob(c, 1, &p, &bytes);
    if (s != SPF_OK) return s;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipMemcpy(p, ksk, bytes, hipMemcpyHostToDevice));
    spf_status st = build_ks_planes(c);
    if (st != SPF_OK) return st;
    c->ksk_ready = true;
    return SPF_OK;
}

// ---------------------------------------------------------------- device buffers for callers without a HIP binding
// (a Rust shim that chains the `_dev` entry points keeps its ciphertexts in HBM between calls; these four spare it a
// HIP binding of its own.  Plain hipMalloc / hipMemcpy on the context's device.)

spf_status spf_device_alloc(spf_ctx* c, size_t bytes, void** dev_ptr)
{
    if (!c || !dev_ptr) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    *dev_ptr = nullptr;
    if (bytes == 0) return SPF_OK;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(dev_ptr, bytes));
    return SPF_OK;
}

spf_status spf_device_free(spf_ctx* c, void* dev_ptr)
{
    if (!c) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null context");
    if (!dev_ptr) return SPF_OK;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipFree(dev_ptr));
    return SPF_OK;
}

spf_status spf_device_upload(spf_ctx* c, void* dev_dst, const void* host_src, size_t bytes)
{
    if (!c || (bytes && (!dev_dst || !host_src))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (bytes == 0) return SPF_OK;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return SPF_OK;
}

spf_status spf_device_download(spf_ctx* c, void* stream, void* host_dst, const void* dev_src, size_t bytes)
{
    if (!c || (bytes && (!host_dst || !dev_src))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize((hipStream_t)stream)); // what was enqueued on `stream` has written dev_src
    if (bytes) HIPCHK(c, hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return SPF_OK;
}

// ---------------------------------------------------------------- device-pointer forms

spf_status spf_keyswitch_lwe_l1_lwe_l0_dev(spf_ctx* c, void* stream, size_t B, const uint64_t* d_in, uint64_t* d_out)
{
    if (!c || (B && (!d_in || !d_out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return launch_keyswitch(c, (hipStream_t)stream, B, d_in, d_out);
}

spf_status spf_generalized_pbs_dev(spf_ctx* c, void* stream, size_t B, const uint64_t* d_lwe, const uint64_t* d_lut,
                                   size_t lut_stride, uint32_t log_chi, uint32_t log_v, uint64_t body_rotate,
                                   uint64_t* d_out)
{
    if (!c || (B && (!d_lwe || !d_lut || !d_out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return launch_blind_rotate(c, (hipStream_t)stream, B, d_lwe, d_lut, lut_stride, log_chi, log_v, body_rotate, d_out,
                               glwe_words(c->prm), false);
}

spf_status spf_pbs_univariate_dev(spf_ctx* c, void* stream, size_t B, const uint64_t* d_lwe, const uint64_t* d_lut,
                                  size_t lut_stride, uint64_t* d_out)
{
    if (!c || (B && (!d_lwe || !d_lut || !d_out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return launch_blind_rotate(c, (hipStream_t)stream, B, d_lwe, d_lut, lut_stride, 0, 0, 0, d_out, lwe1_words(c->prm), true);
}

spf_status spf_circuit_bootstrap_pbs_dev(spf_ctx* c, void* stream, size_t B, const uint64_t* d_lwe, uint64_t* d_out)
{
    if (!c || (B && (!d_lwe || !d_out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    // hi_noise_lwe_to_lo_noise_glwe (circuit_bootstrapping.rs:387-427)
    return launch_blind_rotate(c, (hipStream_t)stream, B, d_lwe, c->d_cbs_lut, 0, 0, ceil_log2(c->prm.cbs_radix_count),
                               (uint64_t)1 << 62, d_out, glwe_words(c->prm), false);
}

static spf_status tail_supported(spf_ctx* c)
{
    const spf_params& p = c->prm;
    if (c->generic) return SPF_OK; // (generic_trace_kernel / generic_scheme_switch_kernel take any radix)
    if (p.cbs_radix_log != 4 || p.cbs_radix_count != 4 || p.tr_radix_log != 7 || p.tr_radix_count != 6 ||
        p.ss_radix_log != 3 || p.ss_radix_count != 15)
        return fail(c, SPF_ERR_UNSUPPORTED, "circuit-bootstrap tail is built for cbs 4x4, tr 6x7, ss 15x3 bits");
    return SPF_OK;
}

static spf_status launch_trace(spf_ctx* c, hipStream_t s, size_t B, const uint64_t* d_glwe, uint64_t* d_glev)
{
    if (!c->ak_ready) return fail(c, SPF_ERR_NO_KEY, "automorphism key not loaded");
    if (c->generic) {
        GenericTraceArgs ga{};
        ga.g = generic_shape(c);
        ga.glwe_in = d_glwe; ga.glev_out = d_glev; ga.ak = c->d_ak;
        ga.units = (uint32_t)(B * c->prm.cbs_radix_count); ga.cbs_count = c->prm.cbs_radix_count; ga.cbs_radix_log = c->prm.cbs_radix_log;
        ga.tr_radix_log = c->prm.tr_radix_log; ga.tr_count = c->prm.tr_radix_count;
        hipLaunchKernelGGL(generic_trace_kernel, dim3(ga.units), dim3(kGenericThreads), generic_trace_lds_bytes(ga.g.N, ga.g.k), s, ga);
        HIPCHK(c, hipGetLastError());
        return SPF_OK;
    }
    TraceArgs a{};
    a.glwe_in = d_glwe; a.glev_out = d_glev; a.ak = c->d_ak; a.tables = c->d_tables;
    a.units = (uint32_t)(B * c->prm.cbs_radix_count); a.cbs_count = c->prm.cbs_radix_count;
    a.cbs_radix_log = c->prm.cbs_radix_log;
    dim3 grid((a.units + kWavesPerBlock - 1) / kWavesPerBlock), block(512);
    TimedScope ts(c, s, T_TRACE);
    spf_status st = ts.begin();
    if (st != SPF_OK) return st;
#ifdef SPF_STAMPS
    static int reported_t = 0;
    if (reported_t < 1 && a.units >= 4096) { // diagnostic build: per-phase cycles per round, median over waves
        const size_t waves = (size_t)grid.x * 8;
        uint64_t* d_st = nullptr;
        HIPCHK(c, hipMalloc(&d_st, waves * 16 * 8));
        HIPCHK(c, hipMemsetAsync(d_st, 0, waves * 16 * 8, s));
        a.stamps = d_st;
        hipLaunchKernelGGL((cbs_trace_kernel<6, 7>), grid, block, kTraceLds, s, a);
        HIPCHK(c, hipStreamSynchronize(s));
        std::vector<uint64_t> h(waves * 16);
        HIPCHK(c, hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(d_st);
        static const char* nm[10] = {"round head: stage, gather, digits 0-1 (+park)", "digits + twist x3", "fwd transform pair x3",
            "cross write + key barrier x3", "cross read + combine + MAD x3", "ring barrier x3", "inverse cross exchange",
            "next rows requested", "inv transform pair", "untwist + convert + acc"};
        fprintf(stderr, "[trace stamps] per automorphism round, median over %zu waves (cycles | waves 0-3 | waves 4-7)\n", waves);
        double tot = 0;
        for (int i = 0; i < 10; i++) {
            std::vector<uint64_t> v, vo, vy;
            for (size_t wv = 0; wv < waves; wv++) {
                v.push_back(h[wv * 16 + i]);
                ((wv % 8) < 4 ? vo : vy).push_back(h[wv * 16 + i]);
            }
            std::sort(v.begin(), v.end()); std::sort(vo.begin(), vo.end()); std::sort(vy.begin(), vy.end());
            fprintf(stderr, "[trace stamps] %-48s %8.0f | %8.0f | %8.0f\n", nm[i], v[v.size() / 2] / 11.0, vo[vo.size() / 2] / 11.0, vy[vy.size() / 2] / 11.0);
            tot += v[v.size() / 2] / 11.0;
        }
        fprintf(stderr, "[trace stamps] total %.0f\n", tot);
        reported_t++;
        return ts.end();
    }
#endif
    hipLaunchKernelGGL((cbs_trace_kernel<6, 7>), grid, block, kTraceLds, s, a);
    HIPCHK(c, hipGetLastError());
    return ts.end();
}

static spf_status launch_scheme_switch(spf_ctx* c, hipStream_t s, size_t B, const uint64_t* d_glev, double* d_ggsw)
{
    if (!c->ssk_ready) return fail(c, SPF_ERR_NO_KEY, "scheme-switch key not loaded");
    if (c->generic) {
        GenericSchemeSwitchArgs ga{};
        ga.g = generic_shape(c);
        ga.glev = d_glev; ga.ggsw_out = reinterpret_cast<c64*>(d_ggsw); ga.ssk = c->d_ssk;
        ga.units = (uint32_t)(B * c->prm.cbs_radix_count); ga.cbs_count = c->prm.cbs_radix_count;
        ga.ss_radix_log = c->prm.ss_radix_log; ga.ss_count = c->prm.ss_radix_count;
        hipLaunchKernelGGL(generic_scheme_switch_kernel, dim3(ga.units), dim3(kGenericThreads), generic_lds_bytes(ga.g.N, ga.g.k, false), s, ga);
        HIPCHK(c, hipGetLastError());
        return SPF_OK;
    }
    SchemeSwitchArgs a{};
    a.glev = d_glev; a.ggsw_out = reinterpret_cast<c64*>(d_ggsw); a.ssk = c->d_ssk; a.tables = c->d_tables;
    a.units = (uint32_t)(B * c->prm.cbs_radix_count); a.cbs_count = c->prm.cbs_radix_count;
    dim3 grid((a.units + kWavesPerBlock - 1) / kWavesPerBlock), block(512);
    TimedScope ts(c, s, T_SS);
    spf_status st = ts.begin();
    if (st != SPF_OK) return st;
    hipLaunchKernelGGL((scheme_switch_kernel<15, 3>), grid, block, kTraceLds, s, a);
    HIPCHK(c, hipGetLastError());
    return ts.end();
}

spf_status spf_mod_switch_trace_and_rotate_dev(spf_ctx* c, void* stream, size_t B, const uint64_t* d_glwe, uint64_t* d_glev)
{
    if (!c || (B && (!d_glwe || !d_glev))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (B == 0) return SPF_OK;
    if (B > 0x0fffffffu) return fail(c, SPF_ERR_INVALID_ARGUMENT, "batch too large");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    spf_status st = tail_supported(c);
    if (st != SPF_OK) return st;
    return launch_trace(c, (hipStream_t)stream, B, d_glwe, d_glev);
}

spf_status spf_scheme_switch_dev(spf_ctx* c, void* stream, size_t B, const uint64_t* d_glev, double* d_ggsw)
{
    if (!c || (B && (!d_glev || !d_ggsw))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (B == 0) return SPF_OK;
    if (B > 0x0fffffffu) return fail(c, SPF_ERR_INVALID_ARGUMENT, "batch too large");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    spf_status st = tail_supported(c);
    if (st != SPF_OK) return st;
    return launch_scheme_switch(c, (hipStream_t)stream, B, d_glev, d_ggsw);
}

// circuit_bootstrap_via_trace_and_scheme_switch (circuit_bootstrapping.rs:342-385) on `s`, intermediates in `sc`
static spf_status circuit_bootstrap_chain(spf_ctx* c, hipStream_t s, size_t B, const uint64_t* d_lwe, double* d_ggsw, Scratch* sc,
                                          int per_wg_hint)
{
    spf_status st = tail_supported(c);
    if (st != SPF_OK) return st;
    st = ensure(c, sc->cbs_glwe, B * glwe_words(c->prm) * 8);
    if (st != SPF_OK) return st;
    st = ensure(c, sc->cbs_glev, B * c->prm.cbs_radix_count * glwe_words(c->prm) * 8);
    if (st != SPF_OK) return st;
    st = launch_blind_rotate(c, s, B, d_lwe, c->d_cbs_lut, 0, 0, ceil_log2(c->prm.cbs_radix_count), (uint64_t)1 << 62,
                             (uint64_t*)sc->cbs_glwe.p, glwe_words(c->prm), false, per_wg_hint);
    if (st != SPF_OK) return st;
    st = launch_trace(c, s, B, (const uint64_t*)sc->cbs_glwe.p, (uint64_t*)sc->cbs_glev.p);
    if (st != SPF_OK) return st;
    return launch_scheme_switch(c, s, B, (const uint64_t*)sc->cbs_glev.p, d_ggsw);
}

spf_status spf_circuit_bootstrap_dev(spf_ctx* c, void* stream, size_t B, const uint64_t* d_lwe, double* d_ggsw)
{
    if (!c || (B && (!d_lwe || !d_ggsw))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (B == 0) return SPF_OK;
    if (B > 0x0fffffffu) return fail(c, SPF_ERR_INVALID_ARGUMENT, "batch too large");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return circuit_bootstrap_chain(c, (hipStream_t)stream, B, d_lwe, d_ggsw, &c->scr, 0);
}

spf_status spf_sample_extract_l1_dev(spf_ctx* c, void* stream, size_t B, const uint64_t* d_glwe, size_t idx, uint64_t* d_out)
{
    if (!c || (B && (!d_glwe || !d_out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (idx >= c->prm.polynomial_degree) return fail(c, SPF_ERR_INVALID_ARGUMENT, "sample_extract index >= polynomial_degree");
    if (B == 0) return SPF_OK;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->generic) {
        const uint32_t N = c->prm.polynomial_degree, k = c->prm.glwe_size;
        for (size_t at = 0; at < B; at += kMaxGridRows) {
            const size_t nb = std::min(B - at, kMaxGridRows);
            hipLaunchKernelGGL(generic_sample_extract_kernel, dim3((k * N + 1 + 255) / 256, (unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                               d_glwe + at * glwe_words(c->prm), d_out + at * lwe1_words(c->prm), (uint32_t)nb, N, k, (uint32_t)idx);
        }
        HIPCHK(c, hipGetLastError());
        return SPF_OK;
    }
    // one grid row per ciphertext; grid.y is limited to 65535, larger batches go in slices
    for (size_t at = 0; at < B; at += kMaxGridRows) {
        const size_t nb = std::min(B - at, kMaxGridRows);
        dim3 grid((kN + 1 + 255) / 256, (unsigned)nb), block(256);
        hipLaunchKernelGGL(sample_extract_kernel, grid, block, 0, (hipStream_t)stream, d_glwe + at * 2 * kN,
                           d_out + at * (kN + 1), (uint32_t)nb, (uint32_t)idx);
    }
    HIPCHK(c, hipGetLastError());
    return SPF_OK;
}

static spf_status glwe_linear_dev(spf_ctx* c, void* stream, size_t B, uint32_t op, const uint64_t* d_a,
                                  const uint64_t* d_b, uint32_t n, uint64_t* d_out)
{
    if (!c || (B && (!d_a || !d_out || (op == GLWE_XOR && !d_b)))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (B == 0) return SPF_OK;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    if (c->generic) {
        const uint32_t N = c->prm.polynomial_degree, k = c->prm.glwe_size;
        for (size_t at = 0; at < B; at += kMaxGridRows) {
            const size_t nb = std::min(B - at, kMaxGridRows), off = at * glwe_words(c->prm);
            hipLaunchKernelGGL(generic_linear_kernel, dim3(((k + 1) * N + 255) / 256, (unsigned)nb), dim3(256), 0, s, d_a + off,
                               d_b ? d_b + off : nullptr, d_out + off, N, c->log_n, k, op == GLWE_NOT ? 0u : (op == GLWE_XOR ? 1u : 2u), n);
        }
        HIPCHK(c, hipGetLastError());
        return SPF_OK;
    }
    for (size_t at = 0; at < B; at += kMaxGridRows) {
        const size_t nb = std::min(B - at, kMaxGridRows), off = at * 2 * kN;
        dim3 grid(2 * kN / 256, (unsigned)nb), block(256);
        const uint64_t* pa = d_a + off;
        const uint64_t* pb = d_b ? d_b + off : nullptr;
        if (op == GLWE_NOT) hipLaunchKernelGGL(glwe_linear_kernel<GLWE_NOT>, grid, block, 0, s, pa, pb, d_out + off, (uint32_t)nb, n);
        else if (op == GLWE_XOR) hipLaunchKernelGGL(glwe_linear_kernel<GLWE_XOR>, grid, block, 0, s, pa, pb, d_out + off, (uint32_t)nb, n);
        else hipLaunchKernelGGL(glwe_linear_kernel<GLWE_MUL_XN>, grid, block, 0, s, pa, pb, d_out + off, (uint32_t)nb, n);
    }
    HIPCHK(c, hipGetLastError());
    return SPF_OK;
}

spf_status spf_glwe_not_dev(spf_ctx* c, void* stream, size_t B, const uint64_t* d_in, uint64_t* d_out)
{
    return glwe_linear_dev(c, stream, B, GLWE_NOT, d_in, nullptr, 0, d_out);
}

spf_status spf_glwe_xor_dev(spf_ctx* c, void* stream, size_t B, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out)
{
    return glwe_linear_dev(c, stream, B, GLWE_XOR, d_a, d_b, 0, d_out);
}

spf_status spf_glwe_mul_xn_dev(spf_ctx* c, void* stream, size_t B, const uint64_t* d_in, size_t n, uint64_t* d_out)
{
    return glwe_linear_dev(c, stream, B, GLWE_MUL_XN, d_in, nullptr, c ? (uint32_t)(n % (2 * (size_t)c->prm.polynomial_degree)) : 0u, d_out);
}

// At most one gate per CU: the four-waves-per-gate latency shape (a level of a gate graph); beyond
// that two gates per workgroup, the streaming shape.
static void launch_cmux_args(spf_ctx* c, hipStream_t s, const CmuxArgs& a)
{
    if (a.B <= (uint32_t)c->n_cu) {
        c->last_cmux_kernel = "cmux4_kernel<4,4>";
#ifdef SPF_STAMPS
        // diagnostic build: per-phase cycles of the first few cmux4 launches (median over waves)
        static int reported = 0;
        static uint64_t* d_st = nullptr;
        if (reported < 6 && a.B <= 256) {
            if (!d_st) (void)hipMalloc(&d_st, 256 * 4 * 16 * 8);
            (void)hipMemsetAsync(d_st, 0, 256 * 4 * 16 * 8, s);
            CmuxArgs b = a;
            b.stamps = d_st;
            hipLaunchKernelGGL((cmux4_kernel<4, 4>), dim3(a.B), dim3(256), kCmux4Lds, s, b);
            (void)hipStreamSynchronize(s);
            std::vector<uint64_t> h((size_t)a.B * 4 * 16);
            (void)hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
            static const char* nm[9] = {"entry: pointers, loads issued, table copy", "barrier (table in place)", "twist tables to registers",
                "decompose + 2 x (transform pair, cross)", "key1 issue + spectra out + 2 barriers", "8-row accumulation chain",
                "inverse cross (3 barriers)", "inverse transform + untwist + store issue", "store drain"};
            fprintf(stderr, "[cmux4 stamps] B=%u\n", a.B);
            double tot = 0;
            for (int i = 0; i < 9; i++) {
                std::vector<uint64_t> v;
                for (size_t wv = 0; wv < (size_t)a.B * 4; wv++) v.push_back(h[wv * 16 + i]);
                std::sort(v.begin(), v.end());
                fprintf(stderr, "[cmux4 stamps] %-44s %8llu\n", nm[i], (unsigned long long)v[v.size() / 2]);
                tot += (double)v[v.size() / 2];
            }
            fprintf(stderr, "[cmux4 stamps] total %.0f cycles\n", tot);
            reported++;
            return;
        }
#endif
        hipLaunchKernelGGL((cmux4_kernel<4, 4>), dim3(a.B), dim3(256), kCmux4Lds, s, a);
    } else {
        // two gates per workgroup, two workgroups per CU: the same eight waves as one workgroup of four gates, but
        // the two halves drift apart, so one half's selector requests fly while the other computes, and a barrier
        // ties four waves instead of eight (0.329 -> 0.316 ms per 4096 gates against four gates per 512-thread workgroup)
#ifdef SPF_STAMPS
        // diagnostic build: per-phase cycles of the first few streaming-shape launches (median over waves)
        static int reported_s = 0;
        if (reported_s < 2 && a.B >= 512) {
            const size_t waves = (size_t)((a.B + 1) / 2) * 4;
            uint64_t* d_st = nullptr;
            (void)hipMalloc(&d_st, waves * 16 * 8);
            (void)hipMemsetAsync(d_st, 0, waves * 16 * 8, s);
            CmuxArgs b = a;
            b.stamps = d_st;
            hipLaunchKernelGGL((cmux_kernel<4, 4, 2>), dim3((a.B + 1) / 2), dim3(256), cmux_lds_bytes(2), s, b);
            (void)hipStreamSynchronize(s);
            std::vector<uint64_t> h(waves * 16);
            (void)hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
            (void)hipFree(d_st);
            static const char* nm[16] = {"entry: operand words in, decomposition", "barrier (table in place)", "digit + twist x8",
                "hand-over (cross data consumed) x7", "forward transform x8", "cross write + hand-over x8", "cross read + combine x8",
                "wait for the round's selector rows x8", "MAD + next rows requested x8", "inverse cross exchange (3 hand-overs)",
                "inverse transform pair", "untwist + d0 + store issue", "entry: kernel arguments + pointer table", "entry: 64 operand loads issued",
                "(unused)", "(unused)"};
            fprintf(stderr, "[cmux stamps] B=%u (cycles per gate, median over %zu waves)\n", a.B, waves);
            double tot = 0;
            for (int i = 0; i < 16; i++) {
                std::vector<uint64_t> v;
                for (size_t wv = 0; wv < waves; wv++) v.push_back(h[wv * 16 + i]);
                std::sort(v.begin(), v.end());
                fprintf(stderr, "[cmux stamps] %-52s %8llu\n", nm[i], (unsigned long long)v[v.size() / 2]);
                tot += (double)v[v.size() / 2];
            }
            fprintf(stderr, "[cmux stamps] total %.0f cycles\n", tot);
            reported_s++;
            return;
        }
#endif
        // selectors of the launch (one per `per_ggsw` units, (k+1) l (k+1) N/2 complex each) that, with the operands, no
        // longer fit the Infinity Cache (256 MB on MI355X; measured cross-over between 768 and 1024 gates of 256 KiB):
        // streaming loads.  Gate graphs (ptrs) share selectors between gates and keep plain loads.
        const size_t sel_bytes = ggsw_fft_complex(c->prm, c->prm.cbs_radix_count) * sizeof(c64);
        const bool stream = !a.ptrs && (size_t)(a.B / (a.per_ggsw ? a.per_ggsw : 1)) * sel_bytes >= ((size_t)224 << 20);
        if (stream) {
            c->last_cmux_kernel = "cmux_kernel<4,4,2,stream>";
            hipLaunchKernelGGL((cmux_kernel<4, 4, 2, true>), dim3((a.B + 1) / 2), dim3(256), cmux_lds_bytes(2), s, a);
        } else {
            c->last_cmux_kernel = "cmux_kernel<4,4,2>";
            hipLaunchKernelGGL((cmux_kernel<4, 4, 2>), dim3((a.B + 1) / 2), dim3(256), cmux_lds_bytes(2), s, a);
        }
    }
}

static spf_status launch_cmux(spf_ctx* c, hipStream_t s, size_t units, uint32_t per_ggsw, const double* d_sel,
                              const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out)
{
    if (c->generic) {
        if (units == 0) return SPF_OK;
        if (units > 0x7fffffffu) return fail(c, SPF_ERR_INVALID_ARGUMENT, "batch too large");
        GenericCmuxArgs ga{};
        ga.g = generic_shape(c);
        ga.ggsw = reinterpret_cast<const c64*>(d_sel); ga.d0 = d_a ? d_a : d_b; ga.d1 = d_b; ga.out = d_out;
        ga.units = (uint32_t)units; ga.per_ggsw = per_ggsw; ga.d0_zero = d_a ? 0u : 1u;
        ga.radix_log = c->prm.cbs_radix_log; ga.count = c->prm.cbs_radix_count;
        c->last_cmux_kernel = "generic_cmux_kernel";
        hipLaunchKernelGGL(generic_cmux_kernel, dim3((unsigned)units), dim3(kGenericThreads), generic_lds_bytes(ga.g.N, ga.g.k, false), s, ga);
        HIPCHK(c, hipGetLastError());
        return SPF_OK;
    }
    if (c->prm.cbs_radix_log != 4 || c->prm.cbs_radix_count != 4)
        return fail(c, SPF_ERR_UNSUPPORTED, "cmux kernel is built for cbs_radix 4 x 4 bits");
    if (units == 0) return SPF_OK;
    if (units > 0x7fffffffu) return fail(c, SPF_ERR_INVALID_ARGUMENT, "batch too large");
    CmuxArgs a{};
    // cmux(c, d_0 = a, d_1 = b, b_fft = sel) (crypto/evaluation.rs:68-83); d_a == nullptr means the
    // zero ciphertext (multiply_glwe_ggsw)
    a.ggsw = reinterpret_cast<const c64*>(d_sel); a.d0 = d_a ? d_a : d_b; a.d1 = d_b; a.out = d_out;
    a.tables = c->d_tables; a.B = (uint32_t)units; a.per_ggsw = per_ggsw; a.d0_zero = d_a ? 0u : 1u;
    TimedScope ts(c, s, T_CMUX);
    spf_status st = ts.begin();
    if (st != SPF_OK) return st;
    launch_cmux_args(c, s, a);
    HIPCHK(c, hipGetLastError());
    return ts.end();
}

spf_status spf_cmux_dev(spf_ctx* c, void* stream, size_t B, const double* d_sel, const uint64_t* d_a, const uint64_t* d_b,
                        uint64_t* d_out)
{
    if (!c || (B && (!d_sel || !d_a || !d_b || !d_out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return launch_cmux(c, (hipStream_t)stream, B, 1, d_sel, d_a, d_b, d_out);
}

// cmux over operands that are not contiguous: d_ptrs is a device array of 4 pointers per unit
// {selector GGSW-FFT, a (null = the zero ciphertext), b, out}; same kernel, same results
spf_status spf_cmux_scattered_dev(spf_ctx* c, void* stream, size_t units, const void* const* d_ptrs)
{
    if (!c || (units && !d_ptrs)) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (!c->generic && (c->prm.cbs_radix_log != 4 || c->prm.cbs_radix_count != 4))
        return fail(c, SPF_ERR_UNSUPPORTED, "cmux kernel is built for cbs_radix 4 x 4 bits");
    if (units == 0) return SPF_OK;
    if (units > 0x7fffffffu) return fail(c, SPF_ERR_INVALID_ARGUMENT, "batch too large");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->generic) {
        GenericCmuxArgs ga{};
        ga.g = generic_shape(c);
        ga.units = (uint32_t)units; ga.per_ggsw = 1; ga.ptrs = d_ptrs;
        ga.radix_log = c->prm.cbs_radix_log; ga.count = c->prm.cbs_radix_count;
        c->last_cmux_kernel = "generic_cmux_kernel";
        hipLaunchKernelGGL(generic_cmux_kernel, dim3((unsigned)units), dim3(kGenericThreads), generic_lds_bytes(ga.g.N, ga.g.k, false),
                           (hipStream_t)stream, ga);
        HIPCHK(c, hipGetLastError());
        return SPF_OK;
    }
    CmuxArgs a{};
    a.tables = c->d_tables; a.B = (uint32_t)units; a.per_ggsw = 1; a.ptrs = d_ptrs;
    launch_cmux_args(c, (hipStream_t)stream, a);
    HIPCHK(c, hipGetLastError());
    return SPF_OK;
}

// dst row r = `words` u64 from d_src_ptrs[r]
spf_status spf_gather_rows_dev(spf_ctx* c, void* stream, size_t rows, size_t words, const uint64_t* const* d_src_ptrs,
                               uint64_t* d_dst)
{
    if (!c || (rows && (!d_src_ptrs || !d_dst))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (rows == 0 || words == 0) return SPF_OK;
    if (words > 0xffffffffu) return fail(c, SPF_ERR_INVALID_ARGUMENT, "gather rows too long");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    for (size_t at = 0; at < rows; at += kMaxGridRows) {
        const size_t nb = std::min(rows - at, kMaxGridRows);
        dim3 grid((unsigned)((words + 255) / 256), (unsigned)nb), block(256);
        hipLaunchKernelGGL(gather_rows_kernel, grid, block, 0, (hipStream_t)stream, d_src_ptrs + at, d_dst + at * words,
                           (uint32_t)nb, (uint32_t)words);
    }
    HIPCHK(c, hipGetLastError());
    return SPF_OK;
}

// KeylessEvaluation::glev_cmux (crypto/evaluation.rs:86-101) = glev_cmux (ops/fft_ops.rs:203-220):
// a cmux over each of the l_cbs GLWEs of two GLEVs with one selector
spf_status spf_glev_cmux_dev(spf_ctx* c, void* stream, size_t B, const double* d_sel, const uint64_t* d_a,
                             const uint64_t* d_b, uint64_t* d_out)
{
    if (!c || (B && (!d_sel || !d_a || !d_b || !d_out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return launch_cmux(c, (hipStream_t)stream, B * c->prm.cbs_radix_count, c->prm.cbs_radix_count, d_sel, d_a, d_b, d_out);
}

// KeylessEvaluation::multiply_glwe_ggsw (crypto/evaluation.rs:104-123): out = IFFT(glwe [*] ggsw)
spf_status spf_multiply_glwe_ggsw_dev(spf_ctx* c, void* stream, size_t B, const uint64_t* d_glwe, const double* d_ggsw,
                                      uint64_t* d_out)
{
    if (!c || (B && (!d_glwe || !d_ggsw || !d_out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return launch_cmux(c, (hipStream_t)stream, B, 1, d_ggsw, nullptr, d_glwe, d_out);
}

// ---------------------------------------------------------------- host-pointer forms

#define STAGE_IN(buf, host, bytes)                                                                \
    do {                                                                                          \
        spf_status s_ = ensure(c, buf, bytes);                                                    \
        if (s_ != SPF_OK) return s_;                                                              \
        HIPCHK(c, hipMemcpyAsync(buf.p, host, bytes, hipMemcpyHostToDevice, c->stream));          \
    } while (0)

spf_status spf_keyswitch_lwe_l1_lwe_l0_batch(spf_ctx* c, size_t B, const uint64_t* in, uint64_t* out)
{
    if (!c || (B && (!in || !out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (B == 0) return SPF_OK;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    STAGE_IN(c->in, in, B * lwe1_words(c->prm) * 8);
    spf_status s = ensure(c, c->out, B * lwe0_words(c->prm) * 8);
    if (s != SPF_OK) return s;
    s = launch_keyswitch(c, c->stream, B, (const uint64_t*)c->in.p, (uint64_t*)c->out.p);
    if (s != SPF_OK) return s;
    HIPCHK(c, hipMemcpyAsync(out, c->out.p, B * lwe0_words(c->prm) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPF_OK;
}

// Bootstrap a device-resident batch and bring the outputs to the caller's host buffer, in SLICES of one
// full round of the chip (4 ciphertexts x #CU): all slices are enqueued on the compute stream at once,
// each followed by an event; the copy stream waits for slice k's event and copies it out while slices
// k+1.. run.  The output is 32 KiB per ciphertext (134 MB per 4096): in one piece behind the kernel it
// added 10-15 % to a call, sliced only the last slice's copy is exposed.  (A pageable destination makes
// hipMemcpyAsync block the host until that copy is done — which is why every kernel is enqueued first.)
static spf_status bootstrap_sliced_to_host(spf_ctx* c, size_t B, const uint64_t* d_lwe, const uint64_t* d_lut,
                                           size_t lut_stride, uint32_t log_chi, uint32_t log_v, uint64_t rot, size_t ow,
                                           bool extract, uint64_t* host_out)
{
    const size_t lw = lwe0_words(c->prm);
    const size_t round = 4 * (size_t)c->n_cu;
    const size_t slice = B > round ? round : B;
    const size_t n_slices = (B + slice - 1) / slice;
    while (c->slice_ev.size() < n_slices) {
        hipEvent_t e;
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->slice_ev.push_back(e);
    }
    uint64_t* d_out = (uint64_t*)c->out.p;
    // Once something is enqueued, no return before both streams are idle: kernels may still be writing c->out and
    // copies may still be landing in the caller's buffer, which the caller is free to release after an error.
    auto enqueue = [&]() -> spf_status {
        for (size_t k = 0; k < n_slices; k++) {
            const size_t off = k * slice, n = std::min(slice, B - off);
            spf_status s = launch_blind_rotate(c, c->stream, n, d_lwe + off * lw, d_lut + off * lut_stride, lut_stride, log_chi,
                                               log_v, rot, d_out + off * ow, ow, extract);
            if (s != SPF_OK) return s;
            HIPCHK(c, hipEventRecord(c->slice_ev[k], c->stream));
        }
        for (size_t k = 0; k < n_slices; k++) {
            const size_t off = k * slice, n = std::min(slice, B - off);
            HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->slice_ev[k], 0));
            HIPCHK(c, hipMemcpyAsync(host_out + off * ow, d_out + off * ow, n * ow * 8, hipMemcpyDeviceToHost, c->copy_stream));
        }
        return SPF_OK;
    };
    const spf_status st = enqueue();
    const hipError_t e1 = hipStreamSynchronize(c->copy_stream), e2 = hipStreamSynchronize(c->stream);
    if (st != SPF_OK) return st; // the message of the first failure stays in place
    HIPCHK(c, e1);
    HIPCHK(c, e2);
    return SPF_OK;
}

static spf_status pbs_host(spf_ctx* c, size_t B, const uint64_t* lwe, const uint64_t* lut, size_t lut_stride,
                           uint32_t log_chi, uint32_t log_v, uint64_t rot, uint64_t* out, bool extract)
{
    if (!c || (B && (!lwe || !out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (B == 0) return SPF_OK;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    STAGE_IN(c->in, lwe, B * lwe0_words(c->prm) * 8);
    const uint64_t* d_lut = c->d_cbs_lut;
    if (lut) {
        size_t luts = lut_stride ? B : 1;
        if (lut_stride && lut_stride < glwe_words(c->prm)) return fail(c, SPF_ERR_INVALID_ARGUMENT, "lut_stride smaller than a GLWE");
        size_t words = lut_stride ? (luts - 1) * lut_stride + glwe_words(c->prm) : glwe_words(c->prm);
        STAGE_IN(c->aux, lut, words * 8);
        d_lut = (const uint64_t*)c->aux.p;
    }
    size_t ow = extract ? lwe1_words(c->prm) : glwe_words(c->prm);
    spf_status s = ensure(c, c->out, B * ow * 8);
    if (s != SPF_OK) return s;
    return bootstrap_sliced_to_host(c, B, (const uint64_t*)c->in.p, d_lut, lut ? lut_stride : 0, log_chi, log_v, rot, ow,
                                    extract, out);
}

spf_status spf_generalized_pbs_batch(spf_ctx* c, size_t B, const uint64_t* lwe, const uint64_t* lut, size_t lut_stride,
                                     uint32_t log_chi, uint32_t log_v, uint64_t body_rotate, uint64_t* out)
{
    if (B && !lut) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null lut");
    return pbs_host(c, B, lwe, lut, lut_stride, log_chi, log_v, body_rotate, out, false);
}

spf_status spf_pbs_univariate_batch(spf_ctx* c, size_t B, const uint64_t* lwe, const uint64_t* lut, size_t lut_stride,
                                    uint64_t* out)
{
    if (B && !lut) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null lut");
    return pbs_host(c, B, lwe, lut, lut_stride, 0, 0, 0, out, true);
}

spf_status spf_circuit_bootstrap_pbs_batch(spf_ctx* c, size_t B, const uint64_t* lwe, uint64_t* out)
{
    if (!c) return SPF_ERR_INVALID_ARGUMENT;
    return pbs_host(c, B, lwe, nullptr, 0, 0, ceil_log2(c->prm.cbs_radix_count), (uint64_t)1 << 62, out, false);
}

spf_status spf_sample_extract_l1_batch(spf_ctx* c, size_t B, const uint64_t* glwe, size_t idx, uint64_t* out)
{
    if (!c || (B && (!glwe || !out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (idx >= c->prm.polynomial_degree) return fail(c, SPF_ERR_INVALID_ARGUMENT, "sample_extract index >= polynomial_degree");
    if (B == 0) return SPF_OK;
    std::lock_guard<std::recursive_mutex> whole(c->mu); // staging buffers are shared: one caller at a time
    {
        HIPCHK(c, hipSetDevice(c->device));
        STAGE_IN(c->in, glwe, B * glwe_words(c->prm) * 8);
        spf_status s = ensure(c, c->out, B * lwe1_words(c->prm) * 8);
        if (s != SPF_OK) return s;
    }
    spf_status s = spf_sample_extract_l1_dev(c, c->stream, B, (const uint64_t*)c->in.p, idx, (uint64_t*)c->out.p);
    if (s != SPF_OK) return s;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipMemcpyAsync(out, c->out.p, B * lwe1_words(c->prm) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPF_OK;
}

static spf_status glwe_linear_host(spf_ctx* c, size_t B, uint32_t op, const uint64_t* a, const uint64_t* b, size_t n,
                                   uint64_t* out)
{
    if (!c || (B && (!a || !out || (op == GLWE_XOR && !b)))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (B == 0) return SPF_OK;
    const size_t gw = glwe_words(c->prm) * 8;
    std::lock_guard<std::recursive_mutex> whole(c->mu); // staging buffers are shared: one caller at a time
    {
        HIPCHK(c, hipSetDevice(c->device));
        STAGE_IN(c->in, a, B * gw);
        if (op == GLWE_XOR) STAGE_IN(c->mid, b, B * gw);
        spf_status s = ensure(c, c->out, B * gw);
        if (s != SPF_OK) return s;
    }
    spf_status s = glwe_linear_dev(c, c->stream, B, op, (const uint64_t*)c->in.p,
                                   op == GLWE_XOR ? (const uint64_t*)c->mid.p : nullptr,
                                   (uint32_t)(n % (2 * (size_t)c->prm.polynomial_degree)), (uint64_t*)c->out.p);
    if (s != SPF_OK) return s;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipMemcpyAsync(out, c->out.p, B * gw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPF_OK;
}

spf_status spf_glwe_not_batch(spf_ctx* c, size_t B, const uint64_t* in, uint64_t* out)
{
    return glwe_linear_host(c, B, GLWE_NOT, in, nullptr, 0, out);
}

spf_status spf_glwe_xor_batch(spf_ctx* c, size_t B, const uint64_t* a, const uint64_t* b, uint64_t* out)
{
    return glwe_linear_host(c, B, GLWE_XOR, a, b, 0, out);
}

spf_status spf_glwe_mul_xn_batch(spf_ctx* c, size_t B, const uint64_t* in, size_t n, uint64_t* out)
{
    return glwe_linear_host(c, B, GLWE_MUL_XN, in, nullptr, n, out);
}

spf_status spf_cmux_batch(spf_ctx* c, size_t B, const double* sel, const uint64_t* a, const uint64_t* b, uint64_t* out)
{
    if (!c || (B && (!sel || !a || !b || !out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (B == 0) return SPF_OK;
    const size_t gw = glwe_words(c->prm) * 8, sw = ggsw_fft_complex(c->prm, c->prm.cbs_radix_count) * 16;
    std::lock_guard<std::recursive_mutex> whole(c->mu); // staging buffers are shared: one caller at a time
    {
        HIPCHK(c, hipSetDevice(c->device));
        STAGE_IN(c->aux, sel, B * sw);
        STAGE_IN(c->in, a, B * gw);
        STAGE_IN(c->mid, b, B * gw);
        spf_status s = ensure(c, c->out, B * gw);
        if (s != SPF_OK) return s;
    }
    spf_status s = spf_cmux_dev(c, c->stream, B, (const double*)c->aux.p, (const uint64_t*)c->in.p,
                                (const uint64_t*)c->mid.p, (uint64_t*)c->out.p);
    if (s != SPF_OK) return s;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipMemcpyAsync(out, c->out.p, B * gw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPF_OK;
}

spf_status spf_glev_cmux_batch(spf_ctx* c, size_t B, const double* sel, const uint64_t* a, const uint64_t* b, uint64_t* out)
{
    if (!c || (B && (!sel || !a || !b || !out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (B == 0) return SPF_OK;
    const size_t ev = glwe_words(c->prm) * 8 * c->prm.cbs_radix_count, sw = ggsw_fft_complex(c->prm, c->prm.cbs_radix_count) * 16;
    std::lock_guard<std::recursive_mutex> whole(c->mu); // staging buffers are shared: one caller at a time
    {
        HIPCHK(c, hipSetDevice(c->device));
        STAGE_IN(c->aux, sel, B * sw);
        STAGE_IN(c->in, a, B * ev);
        STAGE_IN(c->mid, b, B * ev);
        spf_status s = ensure(c, c->out, B * ev);
        if (s != SPF_OK) return s;
    }
    spf_status s = spf_glev_cmux_dev(c, c->stream, B, (const double*)c->aux.p, (const uint64_t*)c->in.p,
                                     (const uint64_t*)c->mid.p, (uint64_t*)c->out.p);
    if (s != SPF_OK) return s;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipMemcpyAsync(out, c->out.p, B * ev, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPF_OK;
}

spf_status spf_multiply_glwe_ggsw_batch(spf_ctx* c, size_t B, const uint64_t* glwe, const double* ggsw, uint64_t* out)
{
    if (!c || (B && (!glwe || !ggsw || !out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (B == 0) return SPF_OK;
    const size_t gw = glwe_words(c->prm) * 8, sw = ggsw_fft_complex(c->prm, c->prm.cbs_radix_count) * 16;
    std::lock_guard<std::recursive_mutex> whole(c->mu); // staging buffers are shared: one caller at a time
    {
        HIPCHK(c, hipSetDevice(c->device));
        STAGE_IN(c->aux, ggsw, B * sw);
        STAGE_IN(c->in, glwe, B * gw);
        spf_status s = ensure(c, c->out, B * gw);
        if (s != SPF_OK) return s;
    }
    spf_status s = spf_multiply_glwe_ggsw_dev(c, c->stream, B, (const uint64_t*)c->in.p, (const double*)c->aux.p,
                                              (uint64_t*)c->out.p);
    if (s != SPF_OK) return s;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipMemcpyAsync(out, c->out.p, B * gw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPF_OK;
}

static spf_status load_fft_key(spf_ctx* c, int which, const double* src, size_t n_complex, size_t want, bool* ready)
{
    if (!c || !src) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (n_complex != want)
        return fail(c, SPF_ERR_INVALID_ARGUMENT, "key length " + std::to_string(n_complex) + " != " + std::to_string(want));
    void* p; size_t bytes;
    spf_status s = spf_key_blob(c, which, &p, &bytes);
    if (s != SPF_OK) return s;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    *ready = true;
    c->ggsw_const_ready = false;
    return SPF_OK;
}

spf_status spf_load_automorphism_key(spf_ctx* c, const double* ak_fft, size_t n_complex)
{
    if (!c) return SPF_ERR_INVALID_ARGUMENT;
    return load_fft_key(c, 2, ak_fft, n_complex, ak_complex(c->prm), &c->ak_ready);
}

spf_status spf_load_scheme_switch_key(spf_ctx* c, const double* ssk_fft, size_t n_complex)
{
    if (!c) return SPF_ERR_INVALID_ARGUMENT;
    return load_fft_key(c, 3, ssk_fft, n_complex, ssk_complex(c->prm), &c->ssk_ready);
}

spf_status spf_mod_switch_trace_and_rotate_batch(spf_ctx* c, size_t B, const uint64_t* glwe_in, uint64_t* glev_out)
{
    if (!c || (B && (!glwe_in || !glev_out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (B == 0) return SPF_OK;
    const size_t gw = glwe_words(c->prm) * 8, ev = gw * c->prm.cbs_radix_count;
    std::lock_guard<std::recursive_mutex> whole(c->mu); // staging buffers are shared: one caller at a time
    {
        HIPCHK(c, hipSetDevice(c->device));
        STAGE_IN(c->in, glwe_in, B * gw);
        spf_status s = ensure(c, c->out, B * ev);
        if (s != SPF_OK) return s;
    }
    spf_status s = spf_mod_switch_trace_and_rotate_dev(c, c->stream, B, (const uint64_t*)c->in.p, (uint64_t*)c->out.p);
    if (s != SPF_OK) return s;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipMemcpyAsync(glev_out, c->out.p, B * ev, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPF_OK;
}

spf_status spf_scheme_switch_batch(spf_ctx* c, size_t B, const uint64_t* glev_in, double* ggsw_out)
{
    if (!c || (B && (!glev_in || !ggsw_out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (B == 0) return SPF_OK;
    const size_t ev = glwe_words(c->prm) * 8 * c->prm.cbs_radix_count;
    const size_t sw = ggsw_fft_complex(c->prm, c->prm.cbs_radix_count) * 16;
    std::lock_guard<std::recursive_mutex> whole(c->mu); // staging buffers are shared: one caller at a time
    {
        HIPCHK(c, hipSetDevice(c->device));
        STAGE_IN(c->in, glev_in, B * ev);
        spf_status s = ensure(c, c->out, B * sw);
        if (s != SPF_OK) return s;
    }
    spf_status s = spf_scheme_switch_dev(c, c->stream, B, (const uint64_t*)c->in.p, (double*)c->out.p);
    if (s != SPF_OK) return s;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipMemcpyAsync(ggsw_out, c->out.p, B * sw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPF_OK;
}

spf_status spf_circuit_bootstrap_batch(spf_ctx* c, size_t B, const uint64_t* lwe0_in, double* ggsw_out)
{
    if (!c || (B && (!lwe0_in || !ggsw_out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (B == 0) return SPF_OK;
    const size_t sw = ggsw_fft_complex(c->prm, c->prm.cbs_radix_count) * 16;
    std::lock_guard<std::recursive_mutex> whole(c->mu); // staging buffers are shared: one caller at a time
    {
        HIPCHK(c, hipSetDevice(c->device));
        STAGE_IN(c->in, lwe0_in, B * lwe0_words(c->prm) * 8);
        spf_status s = ensure(c, c->out, B * sw);
        if (s != SPF_OK) return s;
    }
    spf_status s = spf_circuit_bootstrap_dev(c, c->stream, B, (const uint64_t*)c->in.p, (double*)c->out.p);
    if (s != SPF_OK) return s;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipMemcpyAsync(ggsw_out, c->out.p, B * sw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPF_OK;
}

spf_status spf_keyswitch_circuit_bootstrap_batch(spf_ctx* c, size_t B, const uint64_t* lwe1_in, double* ggsw_out)
{
    if (!c || (B && (!lwe1_in || !ggsw_out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (B == 0) return SPF_OK;
    const size_t sw = ggsw_fft_complex(c->prm, c->prm.cbs_radix_count) * 16;
    std::lock_guard<std::recursive_mutex> whole(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    STAGE_IN(c->in, lwe1_in, B * lwe1_words(c->prm) * 8);
    spf_status s = ensure(c, c->mid, B * lwe0_words(c->prm) * 8); // the level-0 LWE never leaves the device
    if (s != SPF_OK) return s;
    s = ensure(c, c->out, B * sw);
    if (s != SPF_OK) return s;
    s = launch_keyswitch(c, c->stream, B, (const uint64_t*)c->in.p, (uint64_t*)c->mid.p);
    if (s != SPF_OK) return s;
    s = spf_circuit_bootstrap_dev(c, c->stream, B, (const uint64_t*)c->mid.p, (double*)c->out.p);
    if (s != SPF_OK) return s;
    HIPCHK(c, hipMemcpyAsync(ggsw_out, c->out.p, B * sw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPF_OK;
}

spf_status spf_gate_bootstrap_batch(spf_ctx* c, size_t B, const uint64_t* lwe1, uint64_t* glwe_out)
{
    if (!c || (B && (!lwe1 || !glwe_out))) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (B == 0) return SPF_OK;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    STAGE_IN(c->in, lwe1, B * lwe1_words(c->prm) * 8);
    spf_status s = ensure(c, c->mid, B * lwe0_words(c->prm) * 8);
    if (s != SPF_OK) return s;
    s = ensure(c, c->out, B * glwe_words(c->prm) * 8);
    if (s != SPF_OK) return s;
    s = launch_keyswitch(c, c->stream, B, (const uint64_t*)c->in.p, (uint64_t*)c->mid.p);
    if (s != SPF_OK) return s;
    return bootstrap_sliced_to_host(c, B, (const uint64_t*)c->mid.p, c->d_cbs_lut, 0, 0, ceil_log2(c->prm.cbs_radix_count),
                                    (uint64_t)1 << 62, glwe_words(c->prm), false, glwe_out);
}

// ---------------------------------------------------------------- measurement hooks

spf_status spf_set_timing(spf_ctx* c, int enabled)
{
    if (!c) return SPF_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    c->timing = enabled != 0;
    return SPF_OK;
}

spf_status spf_last_kernel_ms(spf_ctx* c, const char* kernel, double* avg_ms, int* launches)
{
    if (!c || !kernel || !avg_ms || !launches) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<TimedLaunch>* v = nullptr;
    for (int k = 0; k < T_COUNT; k++)
        if (!strcmp(kernel, kTimedNames[k])) v = &c->timed[k];
    if (!v) return fail(c, SPF_ERR_INVALID_ARGUMENT, "kernel must be \"pbs\", \"keyswitch\", \"trace\", \"scheme_switch\" or \"cmux\"");
    double total = 0.0;
    int n = 0;
    for (auto& t : *v) {
        HIPCHK(c, hipEventSynchronize(t.stop));
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, t.start, t.stop));
        total += ms; n++;
        (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop);
    }
    v->clear();
    *avg_ms = n ? total / n : 0.0;
    *launches = n;
    return SPF_OK;
}

// ---------------------------------------------------------------- host helpers without a GPU call

// `generate_lut` (ops/bootstrapping/programmable_bootstrapping.rs:129-185) as the trivial GLWE that
// `programmable_bootstrap_univariate` takes (zero mask, body = the table polynomial).  Closed form of the
// reference's fill / negate / rotate: with p = 2^bits inputs, stride = N / p coefficients per input and
// h = stride / 2, body[i] = sign * (f_id(x) << (64 - bits)) for m = (i + h) mod N, x = m / stride,
// id = (m % stride) % 2^ceil(log2 #maps) (0 when id >= #maps), sign = -1 for m < h.
spf_status spf_generate_lut(const spf_params* prm, const uint64_t* map_tables, size_t n_maps, uint32_t plaintext_bits,
                            uint64_t* lut_glwe_out)
{
    if (!prm || !map_tables || !lut_glwe_out || n_maps == 0)
        return fail(nullptr, SPF_ERR_INVALID_ARGUMENT, "spf_generate_lut: null argument or no map");
    const size_t N = prm->polynomial_degree, k = prm->glwe_size;
    if (plaintext_bits == 0 || plaintext_bits >= 64 || ((size_t)1 << plaintext_bits) > N)
        return fail(nullptr, SPF_ERR_INVALID_ARGUMENT, "spf_generate_lut: needs 1 <= plaintext_bits and 2^plaintext_bits <= N");
    const size_t p = (size_t)1 << plaintext_bits, stride = N / p, h = stride / 2;
    size_t ceil_v = 1;
    while (ceil_v < n_maps) ceil_v <<= 1;
    for (size_t i = 0; i < n_maps * p; i++)
        if (map_tables[i] >= p) return fail(nullptr, SPF_ERR_INVALID_ARGUMENT, "spf_generate_lut: a map value is not below 2^plaintext_bits");
    std::memset(lut_glwe_out, 0, k * N * sizeof(uint64_t));
    uint64_t* body = lut_glwe_out + k * N;
    for (size_t i = 0; i < N; i++) {
        const size_t m = (i + h) % N, x = m / stride, id = (m % stride) % ceil_v;
        const uint64_t v = id < n_maps ? map_tables[id * p + x] << (64 - plaintext_bits) : 0;
        body[i] = m < h ? (uint64_t)0 - v : v;
    }
    return SPF_OK;
}

// `safe_bincode::deserialize::<ComputeKey>` (parasol_runtime/src/safe_bincode.rs:16-28, crypto/keys.rs:294-318):
// bincode DefaultOptions + fixint: per field a u64 little-endian element count, then the elements
// (Complex<f64> = two LE f64, Torus<u64> = one LE u64); fields in declaration order bs_key, ks_key, ss_key,
// auto_key; trailing bytes allowed; every count must be exactly what the parameters need (the reference
// bounds the read by GetSize and then runs check_is_valid) — checked for ALL fields before any key is touched.
spf_status spf_load_compute_key_bincode(spf_ctx* c, const uint8_t* bytes, size_t len)
{
    if (!c || !bytes) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    struct Field { const char* name; size_t want, elem; const uint8_t* data; };
    Field f[4] = {{"bs_key", (size_t)c->prm.lwe_dimension * ggsw_fft_complex(c->prm, c->prm.pbs_radix_count), 16, nullptr},
                  {"ks_key", (size_t)c->prm.glwe_size * c->prm.polynomial_degree * c->prm.ks_radix_count * lwe0_words(c->prm), 8, nullptr},
                  {"ss_key", ssk_complex(c->prm), 16, nullptr},
                  {"auto_key", ak_complex(c->prm), 16, nullptr}};
    size_t off = 0;
    for (auto& x : f) {
        if (len - off < 8) return fail(c, SPF_ERR_INVALID_ARGUMENT, std::string("ComputeKey: truncated before the length of ") + x.name);
        uint64_t n = 0;
        for (int b = 7; b >= 0; b--) n = (n << 8) | bytes[off + b]; // little-endian, any alignment
        off += 8;
        if (n != x.want)
            return fail(c, SPF_ERR_INVALID_ARGUMENT, std::string("ComputeKey: ") + x.name + " has " + std::to_string(n) +
                                                         " elements, the parameters need " + std::to_string(x.want));
        if ((len - off) / x.elem < x.want) return fail(c, SPF_ERR_INVALID_ARGUMENT, std::string("ComputeKey: truncated inside ") + x.name);
        x.data = bytes + off;
        off += x.want * x.elem;
    }
    // the loaders only memcpy from these (possibly unaligned) pointers; x86-64 little-endian host = wire order
    spf_status st = spf_load_bootstrap_key(c, reinterpret_cast<const double*>(f[0].data), f[0].want);
    if (st == SPF_OK) st = spf_load_keyswitch_key(c, reinterpret_cast<const uint64_t*>(f[1].data), f[1].want);
    if (st == SPF_OK) st = spf_load_scheme_switch_key(c, reinterpret_cast<const double*>(f[2].data), f[2].want);
    if (st == SPF_OK) st = spf_load_automorphism_key(c, reinterpret_cast<const double*>(f[3].data), f[3].want);
    return st;
}

// Ciphertext wire format: see include/spf_hip.h.  (encryption.rs:23-110, 454-519; safe_bincode.rs:16-28.)
size_t spf_ciphertext_words(const spf_params* p, spf_value_kind kind)
{
    if (!p) return 0;
    switch (kind) {
    case SPF_VAL_LWE0: return lwe0_words(*p);
    case SPF_VAL_LWE1: return lwe1_words(*p);
    case SPF_VAL_GLWE1: return glwe_words(*p);
    case SPF_VAL_GLEV1: return (size_t)p->cbs_radix_count * glwe_words(*p);
    default: return 0;
    }
}

static const char* wire_kind_name(spf_value_kind k)
{
    static const char* n[5] = {"L0LweCiphertext", "L1LweCiphertext", "L1GlweCiphertext", "L1GgswCiphertext", "L1GlevCiphertext"};
    return ((unsigned)k < 5) ? n[k] : "ciphertext";
}

spf_status spf_ciphertext_from_bincode(const spf_params* p, spf_value_kind kind, const uint8_t* bytes, size_t len,
                                       uint64_t* words_out)
{
    if (!p || !bytes || !words_out) return fail(nullptr, SPF_ERR_INVALID_ARGUMENT, "spf_ciphertext_from_bincode: null argument");
    if (kind == SPF_VAL_GGSW1)
        return fail(nullptr, SPF_ERR_UNSUPPORTED, "L1GgswCiphertext has no wire format in the reference (not Serialize)");
    const size_t want = spf_ciphertext_words(p, kind);
    if (want == 0) return fail(nullptr, SPF_ERR_INVALID_ARGUMENT, "spf_ciphertext_from_bincode: unknown ciphertext kind");
    const std::string what = wire_kind_name(kind);
    if (len < 8) return fail(nullptr, SPF_ERR_INVALID_ARGUMENT, what + ": truncated before the element count");
    uint64_t n = 0;
    for (int b = 7; b >= 0; b--) n = (n << 8) | bytes[b]; // little-endian, any alignment
    // the reference bounds the read at (want + 1) * 8 bytes (bincode's SizeLimit error for anything longer) and
    // check_is_valid rejects anything shorter: only the exact count passes
    if (n != want)
        return fail(nullptr, SPF_ERR_INVALID_ARGUMENT, what + ": " + std::to_string(n) + " elements, the parameters need " + std::to_string(want));
    if ((len - 8) / 8 < want) return fail(nullptr, SPF_ERR_INVALID_ARGUMENT, what + ": truncated inside the data");
    for (size_t i = 0; i < want; i++) {
        uint64_t w = 0;
        for (int b = 7; b >= 0; b--) w = (w << 8) | bytes[8 + 8 * i + b];
        words_out[i] = w;
    }
    return SPF_OK;
}

spf_status spf_ciphertext_to_bincode(const spf_params* p, spf_value_kind kind, const uint64_t* words, uint8_t* out, size_t cap,
                                     size_t* written)
{
    if (!p || !words || !out || !written) return fail(nullptr, SPF_ERR_INVALID_ARGUMENT, "spf_ciphertext_to_bincode: null argument");
    if (kind == SPF_VAL_GGSW1)
        return fail(nullptr, SPF_ERR_UNSUPPORTED, "L1GgswCiphertext has no wire format in the reference (not Serialize)");
    const size_t n = spf_ciphertext_words(p, kind);
    if (n == 0) return fail(nullptr, SPF_ERR_INVALID_ARGUMENT, "spf_ciphertext_to_bincode: unknown ciphertext kind");
    if (cap < 8 + 8 * n) return fail(nullptr, SPF_ERR_INVALID_ARGUMENT, "spf_ciphertext_to_bincode: output buffer too small");
    uint64_t v = n;
    for (int b = 0; b < 8; b++) out[b] = (uint8_t)(v >> (8 * b));
    for (size_t i = 0; i < n; i++)
        for (int b = 0; b < 8; b++) out[8 + 8 * i + b] = (uint8_t)(words[i] >> (8 * b));
    *written = 8 + 8 * n;
    return SPF_OK;
}

// l1ggsw_zero / l1ggsw_one as `Evaluation::new` makes them (crypto/evaluation.rs:161-197): the circuit bootstrap
// of the trivial level-0 LWE of the bit, under the loaded keys.  Built once per key set, kept in HBM.
static spf_status ensure_ggsw_constants(spf_ctx* c)
{
    if (c->ggsw_const_ready) return SPF_OK;
    const size_t sw = ggsw_fft_complex(c->prm, c->prm.cbs_radix_count) * 16, lw = lwe0_words(c->prm);
    if (!c->d_ggsw_const) HIPCHK(c, hipMalloc((void**)&c->d_ggsw_const, 2 * sw));
    std::vector<uint64_t> triv(2 * lw, 0);
    triv[2 * lw - 1] = (uint64_t)1 << 63; // trivial_lwe(1, l0, 1 plaintext bit): zero mask, body = 1 << 63
    spf_status st = ensure(c, c->aux, 2 * lw * 8);
    if (st != SPF_OK) return st;
    HIPCHK(c, hipMemcpyAsync(c->aux.p, triv.data(), 2 * lw * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream)); // `triv` goes out of scope
    st = spf_circuit_bootstrap_dev(c, c->stream, 2, (const uint64_t*)c->aux.p, c->d_ggsw_const);
    if (st != SPF_OK) return st;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->ggsw_const_ready = true;
    return SPF_OK;
}

spf_status spf_l1ggsw_constant(spf_ctx* c, int bit, double* ggsw_fft_out)
{
    if (!c || !ggsw_fft_out || (bit != 0 && bit != 1)) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument or bit not 0/1");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    spf_status st = ensure_ggsw_constants(c);
    if (st != SPF_OK) return st;
    const size_t sw = ggsw_fft_complex(c->prm, c->prm.cbs_radix_count) * 16;
    HIPCHK(c, hipMemcpy(ggsw_fft_out, (const char*)c->d_ggsw_const + (size_t)bit * sw, sw, hipMemcpyDeviceToHost));
    return SPF_OK;
}

const char* spf_last_blind_rotate_kernel(spf_ctx* c)
{
    if (!c) return "";
    std::lock_guard<std::recursive_mutex> g(c->mu);
    return c->last_pbs_kernel; // string literals: valid for the life of the library
}

const char* spf_last_cmux_kernel(spf_ctx* c)
{
    if (!c) return "";
    std::lock_guard<std::recursive_mutex> g(c->mu);
    return c->last_cmux_kernel;
}

} // extern "C"

// ---------------------------------------------------------------- call-coalescing pool
// what a pool's launcher thread calls for a batch of a staging set: the set's own stream and intermediates, so that the batches of
// several sets run on the GPU at the same time, and the shape of the whole population of callers (per_wg_hint, launch_blind_rotate)
static spf_status pool_keyswitch(spf_ctx* c, hipStream_t s, size_t B, const uint64_t* d_in, uint64_t* d_out, Scratch* sc)
{
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return launch_keyswitch(c, s, B, d_in, d_out, sc);
}
static spf_status pool_circuit_bootstrap(spf_ctx* c, hipStream_t s, size_t B, const uint64_t* d_lwe, double* d_ggsw, Scratch* sc,
                                         int per_wg_hint)
{
    if (B == 0) return SPF_OK;
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return circuit_bootstrap_chain(c, s, B, d_lwe, d_ggsw, sc, per_wg_hint);
}
// a set's intermediates sized for `cap` operations up front (growing them later would hipFree, i.e. wait for the whole device,
// with other batches in flight)
static spf_status scratch_reserve(spf_ctx* c, Scratch& sc, size_t cap, bool keyswitch, bool circuit_bootstrap)
{
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    spf_status st = SPF_OK;
    if (keyswitch && c->d_ksk_planes) {
        const size_t K = (size_t)c->prm.glwe_size * c->prm.polynomial_degree * c->prm.ks_radix_count;
        const size_t mpad = (cap + KSG_TILE - 1) / KSG_TILE * KSG_TILE;
        st = ensure(c, sc.ks_dig, mpad * K);
        if (st == SPF_OK) st = ensure(c, sc.ks_rowsum, mpad * sizeof(int));
    }
    if (st == SPF_OK && circuit_bootstrap) {
        st = ensure(c, sc.cbs_glwe, cap * glwe_words(c->prm) * 8);
        if (st == SPF_OK) st = ensure(c, sc.cbs_glev, cap * c->prm.cbs_radix_count * glwe_words(c->prm) * 8);
    }
    return st;
}
static spf_status pool_copy_in(spf_ctx* c, hipStream_t s, const void* src, void* dst, size_t bytes)
{
    const size_t words = bytes / 8;
    if (words == 0) return SPF_OK;
    const unsigned blocks = (unsigned)std::min<size_t>((words + 255) / 256, 4 * (size_t)c->n_cu);
    hipLaunchKernelGGL(copy_words_kernel, dim3(blocks), dim3(256), 0, s, (const uint64_t*)src, (uint64_t*)dst, words);
    return hipGetLastError() == hipSuccess ? SPF_OK : SPF_ERR_HIP;
}
static void scratch_free(Scratch& sc)
{
    for (DevBuf* b : {&sc.ks_dig, &sc.ks_rowsum, &sc.cbs_glwe, &sc.cbs_glev}) {
        if (b->p) (void)hipFree(b->p);
        *b = DevBuf{};
    }
}

#include "spf_values.hpp"
#include "spf_pool.hpp"

extern "C" {

spf_status spf_pool_create(spf_ctx* c, size_t max_batch, uint32_t max_wait_us, spf_pool** out)
{
    if (!c || !out || max_batch == 0) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument or max_batch == 0");
    spf_pool* p = new (std::nothrow) spf_pool();
    if (!p) return fail(c, SPF_ERR_HIP, "out of host memory");
    p->ctx = c; p->prm = c->prm; p->max_batch = max_batch;
    p->max_inflight = 4 * max_batch; // flow control: cf. the reference's bounded token channel (circuit_processor/mod.rs:139)
    p->max_wait = std::chrono::microseconds(max_wait_us);
    if (const char* e = getenv("SPF_POOL_GROUPS")) p->groups = (size_t)std::min(std::max(1, atoi(e)), spf_pool_impl::kMaxGroups);
    if (const char* e = getenv("SPF_POOL_PACE")) p->pace_div = atoi(e); // (experiments: -1 = no pacing)
    if (const char* e = getenv("SPF_POOL_SETS")) p->n_sets = std::min(std::max(1, atoi(e)), (int)spf_pool_impl::kSets);
    if (const char* e = getenv("SPF_POOL_SPLIT")) p->split = (size_t)std::min(std::max(1, atoi(e)), 16);
    if (const char* e = getenv("SPF_POOL_SPIN_US")) p->spin_us = std::max(0, atoi(e));
    if (const char* e = getenv("SPF_POOL_DEF_HEAVY")) p->max_def_heavy = std::max(0, atoi(e));
    if (const char* e = getenv("SPF_POOL_TABLE_SETS")) p->n_table_sets = std::min(std::max(1, atoi(e)), (int)spf_pool::kTableSets);
    if (const char* e = getenv("SPF_POOL_HOT_US")) p->hot_us = std::max(0, atoi(e)); // (0: the launcher never polls; completers complete every batch)
    bool streams_ok = hipSetDevice(c->device) == hipSuccess && p->create_def_streams();
    for (auto& set : p->sets)
        streams_ok = streams_ok && hipStreamCreateWithFlags(&set.sk, hipStreamNonBlocking) == hipSuccess;
    if (!streams_ok) {
        p->free_sets();
        p->destroy_def_streams();
        delete p;
        return fail(c, SPF_ERR_HIP, "spf_pool_create: cannot create the pool's streams");
    }
    // Do the sets' streams really run side by side?  The library asks for GPU_MAX_HW_QUEUES=24 when it is loaded, but a host that
    // initialised HIP earlier keeps the runtime's default (4): the pool's resident batches then take turns (0.35-0.50 of the
    // device-resident rate instead of 0.8-0.97, profiles/r05_pool.md).  Measured, not guessed: a 200 us spin kernel on every
    // set's stream at once; streams that share a hardware queue run them one after the other.
    {
        const int n = spf_pool_impl::kSets;
        const uint64_t ticks = 20000; // 200 us of the 100 MHz constant-rate counter
        hipEvent_t e0 = nullptr, e1[spf_pool_impl::kSets] = {};
        bool ok = hipEventCreate(&e0) == hipSuccess;
        for (int i = 0; ok && i < n; i++) ok = hipEventCreate(&e1[i]) == hipSuccess;
        if (ok) {
            for (int i = 0; i < n; i++) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, p->sets[i].sk, (uint64_t)100); // (code object loaded, queues created)
            for (int i = 0; i < n; i++) (void)hipStreamSynchronize(p->sets[i].sk);
            // (the best of three: a device that is busy with somebody else's kernels delays the probe's, which reads as "serialised")
            for (int attempt = 0; attempt < 3 && p->stream_concurrency < n; attempt++) {
                (void)hipEventRecord(e0, p->sets[0].sk);
                (void)hipStreamSynchronize(p->sets[0].sk);
                for (int i = 0; i < n; i++) {
                    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, p->sets[i].sk, ticks);
                    (void)hipEventRecord(e1[i], p->sets[i].sk);
                }
                float worst = 0.f;
                for (int i = 0; i < n; i++) {
                    float ms = 0.f;
                    if (hipEventSynchronize(e1[i]) == hipSuccess && hipEventElapsedTime(&ms, e0, e1[i]) == hipSuccess) worst = std::max(worst, ms);
                }
                if (worst > 0.f) p->stream_concurrency = std::max(p->stream_concurrency, (int)std::min<double>(n, std::max(1.0, std::round(n * 0.2 / worst))));
            }
        }
        (void)hipGetLastError();
        if (e0) (void)hipEventDestroy(e0);
        for (hipEvent_t e : e1)
            if (e) (void)hipEventDestroy(e);
        if (p->stream_concurrency > 0 && p->stream_concurrency <= 5) { // (the runtime's default is four hardware queues)
            const char* q = getenv("GPU_MAX_HW_QUEUES");
            fail(c, SPF_OK, std::string("spf_pool_create: WARNING: only ~") + std::to_string(p->stream_concurrency) + " of the pool's " + std::to_string(n) +
                                " streams run concurrently (GPU_MAX_HW_QUEUES=" + (q ? q : "unset") +
                                " took effect too late or is too small: export GPU_MAX_HW_QUEUES=24 before the process first touches HIP); "
                                "resident batches will take turns");
        }
    }
    try {
        p->arena = std::make_shared<spf_value_impl::Arena>();
        p->arena->device = c->device;
        if (const char* e = getenv("SPF_VALUE_CACHE_MB")) p->arena->cache_limit = (size_t)std::max(0L, atol(e)) << 20;
        p->launcher = std::thread([p] {
            p->launch_loop();
            std::lock_guard<spf_pool::Mutex> lk(p->mu);
            p->launcher_gone = true;
            for (auto& cv : p->cv_fly) cv.notify_all();
        });
        for (int i = 0; i < spf_pool_impl::kSets; i++) p->completers[i] = std::thread([p, i] { p->complete_loop(i); });
    } catch (const std::exception& e) { // std::system_error: no thread to be had — must not cross extern "C"
        {
            std::lock_guard<spf_pool::Mutex> lk(p->mu);
            p->stop = true;
            p->launcher_gone = !p->launcher.joinable();
        }
        p->work_epoch++;
        p->cv_work.notify_all();
        if (p->launcher.joinable()) p->launcher.join();
        for (auto& cv : p->cv_fly) cv.notify_all();
        for (auto& t : p->completers)
            if (t.joinable()) t.join();
        p->free_sets();
        p->destroy_def_streams();
        delete p;
        return fail(c, SPF_ERR_HIP, std::string("spf_pool_create: cannot start the pool threads: ") + e.what());
    }
    *out = p;
    return SPF_OK;
}

spf_status spf_pool_set_max_inflight(spf_pool* p, size_t max_inflight)
{
    if (!p || max_inflight == 0) return SPF_ERR_INVALID_ARGUMENT;
    if (!p->members.empty()) { // a group pool: the bound applies to every member
        for (spf_pool* q : p->members) (void)spf_pool_set_max_inflight(q, max_inflight);
        return SPF_OK;
    }
    std::lock_guard<spf_pool::Mutex> lk(p->mu);
    p->max_inflight = max_inflight;
    p->cv_space.notify_all();
    return SPF_OK;
}

void spf_pool_destroy(spf_pool* p)
{
    if (!p) return;
    if (!p->members.empty()) { // a group pool owns one pool per member and no threads of its own
        for (spf_pool* q : p->members) spf_pool_destroy(q);
        delete p;
        return;
    }
    {
        std::lock_guard<spf_pool::Mutex> lk(p->mu);
        p->stop = true;
    }
    // wake everything that can be parked on this pool: the launcher (it closes and enqueues what is pending, then
    // leaves), producers blocked on back-pressure or on a staging set (they return an error), and waiters (their
    // batches complete: the completion thread outlives the launcher)
    p->work_epoch++;
    p->cv_work.notify_all();
    p->cv_space.notify_all();
    p->cv_set.notify_all();
    if (p->launcher.joinable()) p->launcher.join();
    for (auto& cv : p->cv_fly) cv.notify_all();
    for (auto& t : p->completers)
        if (t.joinable()) t.join();
    {
        // nobody may still be inside submit() / spf_pool_wait() on the mutex and condition variables freed below
        std::unique_lock<spf_pool::Mutex> lk(p->mu);
        p->cv_set.notify_all();
        p->cv_idle.wait(lk, [&] { return p->blocked.load() == 0; });
        for (auto& b : p->collecting) // batches with tickets nobody collected
            b->destroy_events();
    }
    p->free_sets();
    p->destroy_def_streams();
    if (p->arena) p->arena->close(); // cached blocks go back to the driver; values still alive free theirs when they are released
    delete p;
}

} // extern "C"

// the pool an operation of the calling thread goes to: the pool itself, or — a group pool — the thread's home member
// (defined with the group, spf_group.hpp)
static spf_pool* pool_deal(spf_pool* top, int* member);

static spf_status pool_submit(spf_pool* p, int op, const void* a, const void* b, const void* c, void* out, uint64_t* ticket,
                              uint64_t param = 0)
{
    if (!p) return SPF_ERR_INVALID_ARGUMENT;
    int member = 0;
    spf_pool* q = pool_deal(p, &member);
    if (!q) return SPF_ERR_HIP; // no member of the group is in rotation
    const spf_status st = q->submit(op, a, b, c, out, ticket, param);
    if (st == SPF_OK && q != p) *ticket |= (uint64_t)member << spf_pool::kMemberShift;
    return st;
}

extern "C" {

spf_status spf_pool_submit_keyswitch(spf_pool* p, const uint64_t* in, uint64_t* out, uint64_t* ticket)
{
    return pool_submit(p, spf_pool_impl::OP_KEYSWITCH, in, nullptr, nullptr, out, ticket);
}
spf_status spf_pool_submit_circuit_bootstrap(spf_pool* p, const uint64_t* in, double* out, uint64_t* ticket)
{
    return pool_submit(p, spf_pool_impl::OP_CBS, in, nullptr, nullptr, out, ticket);
}
spf_status spf_pool_submit_keyswitch_circuit_bootstrap(spf_pool* p, const uint64_t* in, double* out, uint64_t* ticket)
{
    return pool_submit(p, spf_pool_impl::OP_GATE_CBS, in, nullptr, nullptr, out, ticket);
}
spf_status spf_pool_submit_cmux(spf_pool* p, const double* sel, const uint64_t* a, const uint64_t* b, uint64_t* out,
                                uint64_t* ticket)
{
    if (!p || !a || !b) return SPF_ERR_INVALID_ARGUMENT;
    return pool_submit(p, spf_pool_impl::OP_CMUX, sel, a, b, out, ticket);
}

// the remaining `FheOp` kinds of `CircuitProcessor::exec_op` (circuit_processor/mod.rs:341-540)
spf_status spf_pool_submit_sample_extract(spf_pool* p, const uint64_t* glwe_in, size_t idx, uint64_t* lwe1_out, uint64_t* ticket)
{
    if (!p || idx >= p->prm.polynomial_degree) return SPF_ERR_INVALID_ARGUMENT;
    return pool_submit(p, spf_pool_impl::OP_SAMPLE_EXTRACT, glwe_in, nullptr, nullptr, lwe1_out, ticket, idx);
}
spf_status spf_pool_submit_not(spf_pool* p, const uint64_t* glwe_in, uint64_t* glwe_out, uint64_t* ticket)
{
    return pool_submit(p, spf_pool_impl::OP_NOT, glwe_in, nullptr, nullptr, glwe_out, ticket);
}
spf_status spf_pool_submit_glwe_add(spf_pool* p, const uint64_t* a, const uint64_t* b, uint64_t* glwe_out, uint64_t* ticket)
{
    if (!p || !b) return SPF_ERR_INVALID_ARGUMENT;
    return pool_submit(p, spf_pool_impl::OP_GLWE_ADD, a, b, nullptr, glwe_out, ticket);
}
spf_status spf_pool_submit_mul_xn(spf_pool* p, const uint64_t* glwe_in, size_t n, uint64_t* glwe_out, uint64_t* ticket)
{
    if (!p) return SPF_ERR_INVALID_ARGUMENT;
    return pool_submit(p, spf_pool_impl::OP_MUL_XN, glwe_in, nullptr, nullptr, glwe_out, ticket, n % (2 * (size_t)p->prm.polynomial_degree));
}
spf_status spf_pool_submit_multiply_ggsw_glwe(spf_pool* p, const double* ggsw_fft, const uint64_t* glwe, uint64_t* glwe_out,
                                              uint64_t* ticket)
{
    if (!p || !glwe) return SPF_ERR_INVALID_ARGUMENT;
    return pool_submit(p, spf_pool_impl::OP_MULTIPLY_GGSW_GLWE, ggsw_fft, glwe, nullptr, glwe_out, ticket);
}
spf_status spf_pool_submit_glev_cmux(spf_pool* p, const double* sel_ggsw_fft, const uint64_t* a, const uint64_t* b, uint64_t* glev_out,
                                     uint64_t* ticket)
{
    if (!p || !a || !b) return SPF_ERR_INVALID_ARGUMENT;
    return pool_submit(p, spf_pool_impl::OP_GLEV_CMUX, sel_ggsw_fft, a, b, glev_out, ticket);
}
spf_status spf_pool_submit_scheme_switch(spf_pool* p, const uint64_t* glev_in, double* ggsw_fft_out, uint64_t* ticket)
{
    return pool_submit(p, spf_pool_impl::OP_SCHEME_SWITCH, glev_in, nullptr, nullptr, ggsw_fft_out, ticket);
}

spf_status spf_pool_wait(spf_pool* p, uint64_t ticket)
{
    if (!p) return SPF_ERR_INVALID_ARGUMENT;
    if (!p->members.empty()) {
        const uint64_t member = ticket >> spf_pool::kMemberShift;
        if (member >= p->members.size()) return SPF_ERR_INVALID_ARGUMENT;
        return p->members[member]->wait(ticket & (((uint64_t)1 << spf_pool::kMemberShift) - 1));
    }
    return p->wait(ticket);
}

spf_status spf_pool_stats(spf_pool* p, uint64_t* ops, uint64_t* launches)
{
    if (!p || !ops || !launches) return SPF_ERR_INVALID_ARGUMENT;
    if (!p->members.empty()) {
        *ops = *launches = 0;
        for (spf_pool* q : p->members) {
            uint64_t o = 0, l = 0;
            (void)spf_pool_stats(q, &o, &l);
            *ops += o; *launches += l;
        }
        return SPF_OK;
    }
    std::lock_guard<spf_pool::Mutex> lk(p->mu);
    *ops = p->n_ops; *launches = p->n_launches;
    return SPF_OK;
}

} // extern "C"

// ---------------------------------------------------------------- values: device-resident operands of the per-operation boundary
namespace {

size_t value_bytes(const spf_params& p, int kind)
{
    const size_t k = p.glwe_size, N = p.polynomial_degree, l = p.cbs_radix_count;
    switch (kind) {
    case SPF_VAL_LWE0: return ((size_t)p.lwe_dimension + 1) * 8;
    case SPF_VAL_LWE1: return (k * N + 1) * 8;
    case SPF_VAL_GLWE1: return (k + 1) * N * 8;
    case SPF_VAL_GGSW1: return (k + 1) * l * (k + 1) * (N / 2) * 16;
    case SPF_VAL_GLEV1: return l * (k + 1) * N * 8;
    default: return 0;
    }
}

// operand kinds and result kind of a pool operation (the `FheOp` arms of circuit_processor/mod.rs:255-540)
struct PoolOpInfo { int arity; int in_kind[3]; int out_kind; };
bool pool_op_info(int op, PoolOpInfo* o)
{
    using namespace spf_pool_impl;
    switch (op) {
    case OP_KEYSWITCH: *o = {1, {SPF_VAL_LWE1, -1, -1}, SPF_VAL_LWE0}; return true;
    case OP_CBS: *o = {1, {SPF_VAL_LWE0, -1, -1}, SPF_VAL_GGSW1}; return true;
    case OP_GATE_CBS: *o = {1, {SPF_VAL_LWE1, -1, -1}, SPF_VAL_GGSW1}; return true;
    case OP_CMUX: *o = {3, {SPF_VAL_GGSW1, SPF_VAL_GLWE1, SPF_VAL_GLWE1}, SPF_VAL_GLWE1}; return true;
    case OP_SAMPLE_EXTRACT: *o = {1, {SPF_VAL_GLWE1, -1, -1}, SPF_VAL_LWE1}; return true;
    case OP_NOT: case OP_MUL_XN: *o = {1, {SPF_VAL_GLWE1, -1, -1}, SPF_VAL_GLWE1}; return true;
    case OP_GLWE_ADD: *o = {2, {SPF_VAL_GLWE1, SPF_VAL_GLWE1, -1}, SPF_VAL_GLWE1}; return true;
    case OP_MULTIPLY_GGSW_GLWE: *o = {2, {SPF_VAL_GGSW1, SPF_VAL_GLWE1, -1}, SPF_VAL_GLWE1}; return true;
    case OP_GLEV_CMUX: *o = {3, {SPF_VAL_GGSW1, SPF_VAL_GLEV1, SPF_VAL_GLEV1}, SPF_VAL_GLEV1}; return true;
    case OP_SCHEME_SWITCH: *o = {1, {SPF_VAL_GLEV1, -1, -1}, SPF_VAL_GGSW1}; return true;
    default: return false;
    }
}

// the pool of one context that a value of `member` belongs to: the pool itself, or — a group pool — that member's pool
// (member < 0: the calling thread's home member, dealt as for the host-pointer submits)
spf_pool* value_pool(spf_pool* top, int member, int* which)
{
    *which = 0;
    if (top->members.empty()) return (member <= 0) ? top : nullptr;
    if (member < 0) return pool_deal(top, which);
    if (member >= (int)top->members.size()) return nullptr;
    *which = member;
    return top->members[member];
}

spf_status pool_submit_v(spf_pool* top, int op, const spf_value* const* vals, size_t n_vals, uint64_t param, spf_value** out,
                         uint64_t* ticket)
{
    if (!top || !out || !vals) return SPF_ERR_INVALID_ARGUMENT; // (ticket may be null: nobody will wait for this operation itself)
    PoolOpInfo info{};
    if (!pool_op_info(op, &info) || n_vals != (size_t)info.arity) return fail(top->ctx, SPF_ERR_INVALID_ARGUMENT, "pool operation by handle: wrong number of operands");
    spf_value* v[3] = {nullptr, nullptr, nullptr};
    for (int k = 0; k < info.arity; k++) {
        v[k] = const_cast<spf_value*>(vals[k]);
        if (!v[k]) return fail(top->ctx, SPF_ERR_INVALID_ARGUMENT, "pool operation by handle: null operand");
        if (v[k]->kind != info.in_kind[k]) return fail(top->ctx, SPF_ERR_INVALID_ARGUMENT, "pool operation by handle: operand has the wrong ciphertext type");
        if (v[k]->state.load(std::memory_order_acquire) == spf_value_impl::FAILED)
            return fail(top->ctx, SPF_ERR_INVALID_ARGUMENT, "pool operation by handle: the operation that was to produce this operand failed");
        if (v[k]->home != v[0]->home)
            return fail(top->ctx, SPF_ERR_INVALID_ARGUMENT, "pool operation by handle: operands live on different members of the group (spf_value_copy_to_member moves one)");
    }
    spf_pool* leaf = v[0]->home;
    const int member = v[0]->member;
    const bool mine = top->members.empty() ? leaf == top : (member >= 0 && member < (int)top->members.size() && top->members[member] == leaf);
    if (!mine) return fail(top->ctx, SPF_ERR_INVALID_ARGUMENT, "pool operation by handle: operand belongs to another pool");
    spf_value* res = spf_value::make(leaf->arena, leaf, member, info.out_kind, value_bytes(leaf->prm, info.out_kind));
    if (!res) return fail(leaf->ctx, SPF_ERR_HIP, "out of host memory");
    const spf_status st = leaf->submit_v(op, v, res, ticket, param);
    if (st != SPF_OK) {
        res->release();
        return st == SPF_ERR_INVALID_ARGUMENT ? fail(top->ctx, st, "pool operation by handle: refused (pool closing, or an operand's producing operation failed)") : st;
    }
    if (leaf != top && ticket) *ticket |= (uint64_t)member << spf_pool::kMemberShift;
    *out = res;
    return SPF_OK;
}

} // namespace

extern "C" {

spf_status spf_value_upload(spf_pool* p, int member, spf_value_kind kind, const void* host, spf_value** out)
{
    if (!p || !host || !out) return SPF_ERR_INVALID_ARGUMENT;
    int which = 0;
    spf_pool* leaf = value_pool(p, member, &which);
    if (!leaf) return fail(p->ctx, SPF_ERR_INVALID_ARGUMENT, "spf_value_upload: no such member (or none in rotation)");
    const size_t bytes = value_bytes(leaf->prm, kind);
    if (!bytes) return fail(leaf->ctx, SPF_ERR_INVALID_ARGUMENT, "spf_value_upload: unknown ciphertext kind");
    spf_value* v = spf_value::make(leaf->arena, leaf, which, kind, bytes);
    if (!v) return fail(leaf->ctx, SPF_ERR_HIP, "out of host memory");
    v->blk = spf_value_impl::Block::make(leaf->arena, bytes);
    if (!v->blk) {
        v->release();
        return fail(leaf->ctx, SPF_ERR_HIP, "spf_value_upload: out of device memory");
    }
    spf_value_impl::Arena::DeviceScope ds(leaf->ctx->device);
    const hipError_t e = ds.ok ? hipMemcpy(v->ptr(), host, bytes, hipMemcpyHostToDevice) : hipErrorInvalidDevice;
    if (e != hipSuccess) {
        v->release();
        return fail(leaf->ctx, SPF_ERR_HIP, std::string("spf_value_upload: ") + hipGetErrorString(e));
    }
    v->state.store(spf_value_impl::READY, std::memory_order_release);
    *out = v;
    return SPF_OK;
}

// n ciphertexts of one kind in ONE block and one copy (the bits of an encrypted integer): what spf_value_upload does n times,
// without n copies from pageable memory — and operations that take them in order find them consecutive (no gather pass)
spf_status spf_value_upload_batch(spf_pool* p, int member, spf_value_kind kind, size_t n, const void* host, spf_value** out)
{
    if (!p || !host || !out || n == 0) return SPF_ERR_INVALID_ARGUMENT;
    int which = 0;
    spf_pool* leaf = value_pool(p, member, &which);
    if (!leaf) return fail(p->ctx, SPF_ERR_INVALID_ARGUMENT, "spf_value_upload_batch: no such member (or none in rotation)");
    const size_t bytes = value_bytes(leaf->prm, kind);
    if (!bytes) return fail(leaf->ctx, SPF_ERR_INVALID_ARGUMENT, "spf_value_upload_batch: unknown ciphertext kind");
    if (n > ((size_t)1 << 40) / bytes) return fail(leaf->ctx, SPF_ERR_INVALID_ARGUMENT, "spf_value_upload_batch: too many values");
    std::shared_ptr<spf_value_impl::Block> blk = spf_value_impl::Block::make(leaf->arena, n * bytes);
    if (!blk) return fail(leaf->ctx, SPF_ERR_HIP, "spf_value_upload_batch: out of device memory");
    {
        spf_value_impl::Arena::DeviceScope ds(leaf->ctx->device);
        const hipError_t e = ds.ok ? hipMemcpy(blk->p, host, n * bytes, hipMemcpyHostToDevice) : hipErrorInvalidDevice;
        if (e != hipSuccess) return fail(leaf->ctx, SPF_ERR_HIP, std::string("spf_value_upload_batch: ") + hipGetErrorString(e));
    }
    for (size_t i = 0; i < n; i++) {
        spf_value* v = spf_value::make(leaf->arena, leaf, which, kind, bytes);
        if (!v) {
            for (size_t j = 0; j < i; j++) { out[j]->release(); out[j] = nullptr; }
            return fail(leaf->ctx, SPF_ERR_HIP, "out of host memory");
        }
        v->blk = blk;
        v->off = i * bytes;
        v->state.store(spf_value_impl::READY, std::memory_order_release);
        out[i] = v;
    }
    return SPF_OK;
}

// n valid values of one kind and one context -> host, consecutive; ONE copy when they lie consecutively in one block (the
// results of one batch, or of spf_value_upload_batch), one copy each otherwise
spf_status spf_value_download_batch(size_t n, const spf_value* const* values, void* host)
{
    if (!values || !host || n == 0) return SPF_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < n; i++)
        if (!values[i] || !values[i]->ready() || values[i]->kind != values[0]->kind || values[i]->arena != values[0]->arena) return SPF_ERR_INVALID_ARGUMENT;
    const size_t bytes = values[0]->bytes;
    bool consecutive = true;
    for (size_t i = 1; i < n && consecutive; i++)
        consecutive = values[i]->blk == values[0]->blk && values[i]->off == values[0]->off + i * bytes;
    spf_value_impl::Arena::DeviceScope ds(values[0]->arena->device);
    if (!ds.ok) return SPF_ERR_HIP;
    if (consecutive) return hipMemcpy(host, values[0]->ptr(), n * bytes, hipMemcpyDeviceToHost) == hipSuccess ? SPF_OK : SPF_ERR_HIP;
    bool pool_alive;
    {
        std::lock_guard<std::mutex> lk(values[0]->arena->mu);
        pool_alive = !values[0]->arena->closed; // (values may outlive their pool: its context and stream are then gone)
    }
    if (n >= 4 && pool_alive) {
        // scattered (the outputs of a circuit come from different batches): packed on the device into one scratch block
        // (gather_rows_kernel reading a pointer table at the block's end), then ONE copy — a copy to pageable memory costs ~25 us
        // per call whatever its size
        spf_ctx* c = values[0]->home->ctx;
        const size_t table_at = (n * bytes + 255) / 256 * 256;
        std::shared_ptr<spf_value_impl::Block> tmp = spf_value_impl::Block::make(values[0]->arena, table_at + n * sizeof(void*));
        std::vector<const void*> ptrs;
        try {
            for (size_t i = 0; i < n; i++) ptrs.push_back(values[i]->ptr());
        } catch (const std::exception&) {
            tmp.reset();
        }
        if (tmp) {
            char* base = static_cast<char*>(tmp->p);
            std::lock_guard<std::recursive_mutex> g(c->mu);
            if (hipMemcpyAsync(base + table_at, ptrs.data(), n * sizeof(void*), hipMemcpyHostToDevice, c->stream) != hipSuccess) return SPF_ERR_HIP;
            spf_status st = spf_gather_rows_dev(c, c->stream, n, bytes / 8, (const uint64_t* const*)(base + table_at), (uint64_t*)base);
            if (st != SPF_OK) return st;
            if (hipMemcpyAsync(host, base, n * bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return SPF_ERR_HIP;
            return hipStreamSynchronize(c->stream) == hipSuccess ? SPF_OK : SPF_ERR_HIP;
        }
    }
    for (size_t i = 0; i < n; i++)
        if (hipMemcpy(static_cast<char*>(host) + i * bytes, values[i]->ptr(), bytes, hipMemcpyDeviceToHost) != hipSuccess) return SPF_ERR_HIP;
    return SPF_OK;
}

// FheOp::{Zero,One}{Lwe0,Lwe1,Glwe1,Glev1,Ggsw1} (fhe_circuit.rs:96-116) as values
spf_status spf_value_trivial(spf_pool* p, int member, spf_value_kind kind, uint64_t bit, spf_value** out)
{
    if (!p || !out || bit > 1) return SPF_ERR_INVALID_ARGUMENT;
    int which = 0;
    spf_pool* leaf = value_pool(p, member, &which);
    if (!leaf) return fail(p->ctx, SPF_ERR_INVALID_ARGUMENT, "spf_value_trivial: no such member (or none in rotation)");
    const spf_params& prm = leaf->prm;
    const size_t bytes = value_bytes(prm, kind);
    if (!bytes) return fail(leaf->ctx, SPF_ERR_INVALID_ARGUMENT, "spf_value_trivial: unknown ciphertext kind");
    if (kind != SPF_VAL_GGSW1) {
        // trivial_lwe / trivial_glwe / trivial_glev of a bit (crypto/encryption.rs:345-451): zero mask, body (coefficient 0) =
        // bit << 63, or bit * q / B^(j+1) for GLWE j of a GLEV — the same words spf_graph_add_trivial puts into a graph
        std::vector<uint64_t> w;
        try {
            w.assign(bytes / 8, 0);
        } catch (const std::exception&) {
            return fail(leaf->ctx, SPF_ERR_HIP, "out of host memory");
        }
        const size_t k = prm.glwe_size, N = prm.polynomial_degree;
        if (kind == SPF_VAL_GLEV1)
            for (size_t j = 0; j < prm.cbs_radix_count; j++) w[j * (k + 1) * N + k * N] = bit << (64 - prm.cbs_radix_log * (j + 1));
        else
            w[kind == SPF_VAL_LWE0 ? prm.lwe_dimension : k * N] = bit << 63;
        return spf_value_upload(p, which, kind, w.data(), out);
    }
    // l1ggsw_zero / l1ggsw_one: the context's circuit bootstraps of the trivial level-0 LWE (Evaluation::new, evaluation.rs:161-197)
    spf_ctx* c = leaf->ctx;
    spf_value* v = spf_value::make(leaf->arena, leaf, which, kind, bytes);
    if (!v) return fail(c, SPF_ERR_HIP, "out of host memory");
    v->blk = spf_value_impl::Block::make(leaf->arena, bytes);
    if (!v->blk) {
        v->release();
        return fail(c, SPF_ERR_HIP, "spf_value_trivial: out of device memory");
    }
    spf_status st;
    {
        spf_value_impl::Arena::DeviceScope ds(c->device);
        std::lock_guard<std::recursive_mutex> g(c->mu);
        st = ds.ok ? ensure_ggsw_constants(c) : fail(c, SPF_ERR_HIP, "hipSetDevice failed");
        // (a device-to-device hipMemcpy may return before the copy has run: the value is handed out only after the stream is idle —
        // the kernels that will read it run on the pool's own non-blocking streams, which do not wait for the null stream)
        if (st == SPF_OK && (hipMemcpy(v->ptr(), (const char*)c->d_ggsw_const + (size_t)bit * bytes, bytes, hipMemcpyDeviceToDevice) != hipSuccess ||
                             hipStreamSynchronize(nullptr) != hipSuccess))
            st = fail(c, SPF_ERR_HIP, "spf_value_trivial: device copy failed");
    }
    if (st != SPF_OK) {
        v->release();
        return st;
    }
    v->state.store(spf_value_impl::READY, std::memory_order_release);
    *out = v;
    return SPF_OK;
}

spf_status spf_value_download(const spf_value* v, void* host)
{
    if (!v || !host || !v->ready()) return SPF_ERR_INVALID_ARGUMENT;
    spf_value_impl::Arena::DeviceScope ds(v->arena->device);
    if (!ds.ok || hipMemcpy(host, v->ptr(), v->bytes, hipMemcpyDeviceToHost) != hipSuccess) return SPF_ERR_HIP;
    return SPF_OK;
}

spf_status spf_value_wait(const spf_value* v)
{
    if (!v) return SPF_ERR_INVALID_ARGUMENT;
    // (a value may outlive its pool, a PENDING one cannot: spf_pool_destroy drains every batch first)
    const int st = v->state.load(std::memory_order_acquire);
    if (st == spf_value_impl::READY) return SPF_OK;
    if (st == spf_value_impl::FAILED) return SPF_ERR_HIP;
    return v->home ? v->home->wait_value(v) : SPF_ERR_INVALID_ARGUMENT;
}

spf_status spf_pool_flush(spf_pool* p)
{
    if (!p) return SPF_ERR_INVALID_ARGUMENT;
    auto one = [](spf_pool* q) {
        std::lock_guard<spf_pool::Mutex> lk(q->mu);
        q->flush_deferred();
    };
    if (p->members.empty()) one(p);
    else for (spf_pool* q : p->members) one(q);
    return SPF_OK;
}

spf_status spf_value_retain(spf_value* v)
{
    if (!v) return SPF_ERR_INVALID_ARGUMENT;
    v->retain();
    return SPF_OK;
}

void spf_value_release(spf_value* v)
{
    if (v) v->release();
}

spf_status spf_value_info(const spf_value* v, spf_value_kind* kind, size_t* bytes, int* member, int* ready)
{
    if (!v) return SPF_ERR_INVALID_ARGUMENT;
    if (kind) *kind = (spf_value_kind)v->kind;
    if (bytes) *bytes = v->bytes;
    if (member) *member = v->member;
    if (ready) *ready = v->ready() ? 1 : 0;
    return SPF_OK;
}

spf_status spf_value_device_ptr(const spf_value* v, void** dev_ptr)
{
    if (!v || !dev_ptr || !v->ready()) return SPF_ERR_INVALID_ARGUMENT;
    *dev_ptr = v->ptr();
    return SPF_OK;
}

spf_status spf_value_copy_to_member(spf_pool* p, const spf_value* v, int member, spf_value** out)
{
    if (!p || !v || !out || !v->ready()) return SPF_ERR_INVALID_ARGUMENT;
    int which = 0;
    spf_pool* leaf = value_pool(p, member, &which);
    if (!leaf || member < 0) return fail(p->ctx, SPF_ERR_INVALID_ARGUMENT, "spf_value_copy_to_member: no such member");
    spf_value* w = spf_value::make(leaf->arena, leaf, which, v->kind, v->bytes);
    if (!w) return fail(leaf->ctx, SPF_ERR_HIP, "out of host memory");
    w->blk = spf_value_impl::Block::make(leaf->arena, v->bytes);
    if (!w->blk) {
        w->release();
        return fail(leaf->ctx, SPF_ERR_HIP, "spf_value_copy_to_member: out of device memory");
    }
    spf_value_impl::Arena::DeviceScope ds(leaf->ctx->device);
    hipError_t e = !ds.ok ? hipErrorInvalidDevice
                   : (v->arena->device == leaf->ctx->device ? hipMemcpy(w->ptr(), v->ptr(), v->bytes, hipMemcpyDeviceToDevice)
                                                            : hipMemcpyPeer(w->ptr(), leaf->ctx->device, v->ptr(), v->arena->device, v->bytes));
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr); // (a device-to-device copy may return before it has run; see spf_value_trivial)
    if (e != hipSuccess) {
        w->release();
        return fail(leaf->ctx, SPF_ERR_HIP, std::string("spf_value_copy_to_member: ") + hipGetErrorString(e));
    }
    w->state.store(spf_value_impl::READY, std::memory_order_release);
    *out = w;
    return SPF_OK;
}

spf_status spf_pool_value_stats(spf_pool* p, size_t* live_values, size_t* live_bytes, size_t* cached_bytes)
{
    if (!p) return SPF_ERR_INVALID_ARGUMENT;
    size_t lv = 0, lb = 0, cb = 0;
    auto add = [&](spf_pool* q) {
        lv += q->arena->live_values.load();
        lb += q->arena->live_bytes.load();
        std::lock_guard<std::mutex> lk(q->arena->mu);
        cb += q->arena->cached_bytes;
    };
    if (p->members.empty()) add(p);
    else for (spf_pool* q : p->members) add(q);
    if (live_values) *live_values = lv;
    if (live_bytes) *live_bytes = lb;
    if (cached_bytes) *cached_bytes = cb;
    return SPF_OK;
}

spf_status spf_pool_trim(spf_pool* p)
{
    if (!p) return SPF_ERR_INVALID_ARGUMENT;
    if (p->members.empty()) p->arena->trim();
    else for (spf_pool* q : p->members) q->arena->trim();
    return SPF_OK;
}

spf_status spf_pool_counters_get(spf_pool* p, spf_pool_counters* out)
{
    if (!p || !out) return SPF_ERR_INVALID_ARGUMENT;
    *out = spf_pool_counters{};
    auto add = [&](spf_pool* q) {
        std::lock_guard<spf_pool::Mutex> lk(q->mu);
        out->ops += q->n_ops; out->launches += q->n_launches;
        out->handle_ops += q->n_handle_ops; out->handle_launches += q->n_handle_launches;
        out->reclaimed += q->n_reclaimed;
        for (int i = 0; i < 3; i++) out->bootstrap_launches_by_shape[i] += q->n_shape[i];
        out->staging_sets = (uint64_t)q->n_sets;
        out->stream_concurrency = out->stream_concurrency ? std::min<uint64_t>(out->stream_concurrency, (uint64_t)q->stream_concurrency) : (uint64_t)q->stream_concurrency;
        out->value_mallocs += q->arena->n_malloc.load();
    };
    if (p->members.empty()) add(p);
    else for (spf_pool* q : p->members) add(q);
    return SPF_OK;
}

// ---- the pool's submits by handle
spf_status spf_pool_submit_keyswitch_v(spf_pool* p, const spf_value* lwe1, spf_value** lwe0_out, uint64_t* ticket)
{
    return pool_submit_v(p, spf_pool_impl::OP_KEYSWITCH, &lwe1, 1, 0, lwe0_out, ticket);
}
spf_status spf_pool_submit_circuit_bootstrap_v(spf_pool* p, const spf_value* lwe0, spf_value** ggsw_out, uint64_t* ticket)
{
    return pool_submit_v(p, spf_pool_impl::OP_CBS, &lwe0, 1, 0, ggsw_out, ticket);
}
spf_status spf_pool_submit_keyswitch_circuit_bootstrap_v(spf_pool* p, const spf_value* lwe1, spf_value** ggsw_out, uint64_t* ticket)
{
    return pool_submit_v(p, spf_pool_impl::OP_GATE_CBS, &lwe1, 1, 0, ggsw_out, ticket);
}
spf_status spf_pool_submit_cmux_v(spf_pool* p, const spf_value* sel, const spf_value* a, const spf_value* b, spf_value** out,
                                  uint64_t* ticket)
{
    const spf_value* v[3] = {sel, a, b};
    return pool_submit_v(p, spf_pool_impl::OP_CMUX, v, 3, 0, out, ticket);
}
spf_status spf_pool_submit_sample_extract_v(spf_pool* p, const spf_value* glwe, size_t idx, spf_value** lwe1_out, uint64_t* ticket)
{
    if (!p || idx >= p->prm.polynomial_degree) return SPF_ERR_INVALID_ARGUMENT;
    return pool_submit_v(p, spf_pool_impl::OP_SAMPLE_EXTRACT, &glwe, 1, idx, lwe1_out, ticket);
}
spf_status spf_pool_submit_not_v(spf_pool* p, const spf_value* glwe, spf_value** out, uint64_t* ticket)
{
    return pool_submit_v(p, spf_pool_impl::OP_NOT, &glwe, 1, 0, out, ticket);
}
spf_status spf_pool_submit_glwe_add_v(spf_pool* p, const spf_value* a, const spf_value* b, spf_value** out, uint64_t* ticket)
{
    const spf_value* v[2] = {a, b};
    return pool_submit_v(p, spf_pool_impl::OP_GLWE_ADD, v, 2, 0, out, ticket);
}
spf_status spf_pool_submit_mul_xn_v(spf_pool* p, const spf_value* glwe, size_t n, spf_value** out, uint64_t* ticket)
{
    if (!p) return SPF_ERR_INVALID_ARGUMENT;
    return pool_submit_v(p, spf_pool_impl::OP_MUL_XN, &glwe, 1, n % (2 * (size_t)p->prm.polynomial_degree), out, ticket);
}
spf_status spf_pool_submit_multiply_ggsw_glwe_v(spf_pool* p, const spf_value* ggsw, const spf_value* glwe, spf_value** out,
                                                uint64_t* ticket)
{
    const spf_value* v[2] = {ggsw, glwe};
    return pool_submit_v(p, spf_pool_impl::OP_MULTIPLY_GGSW_GLWE, v, 2, 0, out, ticket);
}
spf_status spf_pool_submit_glev_cmux_v(spf_pool* p, const spf_value* sel, const spf_value* a, const spf_value* b, spf_value** out,
                                       uint64_t* ticket)
{
    const spf_value* v[3] = {sel, a, b};
    return pool_submit_v(p, spf_pool_impl::OP_GLEV_CMUX, v, 3, 0, out, ticket);
}
spf_status spf_pool_submit_scheme_switch_v(spf_pool* p, const spf_value* glev, spf_value** ggsw_out, uint64_t* ticket)
{
    return pool_submit_v(p, spf_pool_impl::OP_SCHEME_SWITCH, &glev, 1, 0, ggsw_out, ticket);
}

// one entry for `exec_op`'s whole match (circuit_processor/mod.rs:255-540): the operation as a spf_graph_op, operands in the
// order spf_graph_add_op takes them
spf_status spf_pool_submit_op_v(spf_pool* p, spf_graph_op op, const spf_value* const* inputs, size_t n_inputs, uint64_t param,
                                spf_value** out, uint64_t* ticket)
{
    using namespace spf_pool_impl;
    if (!p) return SPF_ERR_INVALID_ARGUMENT;
    int pop;
    switch (op) {
    case SPF_OP_SAMPLE_EXTRACT:
        if (param >= p->prm.polynomial_degree) return fail(p->ctx, SPF_ERR_INVALID_ARGUMENT, "sample_extract index >= polynomial_degree");
        pop = OP_SAMPLE_EXTRACT;
        break;
    case SPF_OP_KEYSWITCH_L1_TO_L0: pop = OP_KEYSWITCH; param = 0; break;
    case SPF_OP_NOT: pop = OP_NOT; param = 0; break;
    case SPF_OP_GLWE_ADD: pop = OP_GLWE_ADD; param = 0; break;
    case SPF_OP_CMUX: pop = OP_CMUX; param = 0; break;
    case SPF_OP_GLEV_CMUX: pop = OP_GLEV_CMUX; param = 0; break;
    case SPF_OP_MULTIPLY_GGSW_GLWE: pop = OP_MULTIPLY_GGSW_GLWE; param = 0; break;
    case SPF_OP_CIRCUIT_BOOTSTRAP: pop = OP_CBS; param = 0; break;
    case SPF_OP_SCHEME_SWITCH: pop = OP_SCHEME_SWITCH; param = 0; break;
    case SPF_OP_MUL_XN: pop = OP_MUL_XN; param %= 2 * (uint64_t)p->prm.polynomial_degree; break;
    default: return fail(p->ctx, SPF_ERR_INVALID_ARGUMENT, "unknown operation");
    }
    return pool_submit_v(p, pop, inputs, n_inputs, param, out, ticket);
}

} // extern "C"

#include "spf_graph.hpp"
#include "spf_group.hpp"
