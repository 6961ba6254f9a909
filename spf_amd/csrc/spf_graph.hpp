// spf_graph.hpp — level-batching executor for gate graphs.
//
// The reference runs an `FheCircuit` (parasol_runtime/src/fhe_circuit.rs:34-126: a DAG of `FheOp`
// nodes) by spawning one rayon task per node as its operands become ready
// (circuit_processor/mod.rs:130-253) and calling `Evaluation` with ONE ciphertext per task
// (`exec_op`, :255-560).  On a GPU the same DAG is executed by topological level: all nodes of one
// level and one kind become one batched launch, every intermediate value stays in HBM, and the whole
// graph is enqueued on one HIP stream without a host round trip — inputs go up in one copy, outputs
// come back at the end.  Operands of a level live wherever their producers wrote them: the CMUX
// family reads them through a pointer table (`CmuxArgs::ptrs`; a 256 KiB GGSW per gate is not worth
// copying), the small operands of the other kinds are packed by `gather_rows_kernel` unless they are
// already contiguous (the common KeyswitchL1toL0 -> CircuitBootstrap chain).
//
// Included at the end of spf_hip.hip: uses its `fail` / HIPCHK helpers and the `_dev` entry points.
#pragma once

#include <algorithm>
#include <cstring>
#include <map>
#include <tuple>
#include <vector>

// host memory the copy engines can reach directly (hipHostMalloc): a copy to or from pageable memory is staged by the runtime
// through its own pinned bounce buffer, synchronously, ~20 us per call — 33 of them were 0.68 of a 32-bit addition's 5.5 ms
struct spf_pinned_buf {
    uint8_t* p = nullptr;
    size_t n = 0;
    bool pinned = false;
    uint8_t* data() const { return p; }
    bool resize(size_t bytes)
    {
        if (p) { if (pinned) (void)hipHostFree(p); else std::free(p); }
        p = nullptr;
        n = 0;
        if (!bytes) return true;
        pinned = hipHostMalloc((void**)&p, bytes, hipHostMallocDefault) == hipSuccess;
        if (!pinned) { // (no pinned memory left for a very large graph: pageable still works, the copies are just staged by the runtime)
            (void)hipGetLastError();
            p = static_cast<uint8_t*>(std::malloc(bytes));
            if (!p) return false;
        }
        std::memset(p, 0, bytes);
        n = bytes;
        return true;
    }
    void free_buf() { (void)resize(0); }
};

struct spf_graph {
    struct Node {
        int32_t op;        // spf_graph_op, or -1 input, -2 trivial constant
        int32_t kind;      // spf_value_kind of the value the node produces
        uint64_t param;    // SampleExtract index / MulXN amount / trivial bit
        uint32_t in[3];
        uint32_t n_in;
        uint32_t level;
        size_t off;        // byte offset of the value in the arena
        const void* host;  // inputs: caller's buffer, read at every run
    };
    struct Group {
        int32_t op;
        uint64_t param;
        std::vector<uint32_t> members;
        size_t out_off = 0;
        // per operand slot: offset of the first operand when the members' operands are contiguous
        // in member order, else index of the slot's pointer row in the table
        bool contiguous[3] = {false, false, false};
        size_t first_off[3] = {0, 0, 0};
        size_t ptr_index[3] = {0, 0, 0};
    };
    spf_ctx* ctx = nullptr;
    spf_params prm{};
    struct spf_group* grp = nullptr; // a graph of a device group (spf_group_graph_create): placed on a member when it is run
    int member = -1;                 // ... the member its most recent run took place on
    std::vector<Node> nodes;
    std::vector<std::pair<uint32_t, void*>> outputs;
    bool planned = false;
    std::vector<Group> groups;
    spf_pinned_buf h_inputs;
    // outputs leave in ONE device-to-host copy: gathered on the device (one gather_rows launch per value size) into d_out_stage,
    // copied into pinned h_out_stage, handed to the callers' buffers from there
    struct OutClass { size_t words = 0, count = 0, ptr_index = 0, stage_off = 0; std::vector<size_t> which; };
    std::vector<OutClass> out_classes;
    spf_pinned_buf h_out_stage;
    char* d_out_stage = nullptr;
    void** d_out_ptrs = nullptr;
    size_t out_bytes = 0, outputs_planned = (size_t)-1;
    void release_outputs()
    {
        if (d_out_stage) (void)hipFree(d_out_stage);
        if (d_out_ptrs) (void)hipFree(d_out_ptrs);
        d_out_stage = nullptr;
        d_out_ptrs = nullptr;
        h_out_stage.free_buf();
        out_classes.clear();
        outputs_planned = (size_t)-1;
    }
    size_t inputs_bytes = 0, arena_bytes = 0, stage_bytes = 0;
    char* d_arena = nullptr;
    char* d_stage[2] = {nullptr, nullptr};
    void** d_ptrs = nullptr;
    uint32_t n_levels = 0, n_launches = 0;
    // hipGraph of the planned launch sequence (everything between the input copy and the output copies; opt-in,
    // SPF_GRAPH_CAPTURE=1): captured on the second run after planning (the first one sizes the context's scratch buffers),
    // replayed afterwards; dropped when the graph is re-planned or a captured scratch buffer moved
    hipGraphExec_t exec = nullptr;
    uint64_t exec_epoch = 0;
    uint32_t runs_since_plan = 0, captured_launches = 0;

    size_t value_bytes(int kind) const
    {
        const size_t k = prm.glwe_size, N = prm.polynomial_degree, l = prm.cbs_radix_count;
        switch (kind) {
        case SPF_VAL_LWE0: return ((size_t)prm.lwe_dimension + 1) * 8;
        case SPF_VAL_LWE1: return (k * N + 1) * 8;
        case SPF_VAL_GLWE1: return (k + 1) * N * 8;
        case SPF_VAL_GGSW1: return (k + 1) * l * (k + 1) * (N / 2) * 16;
        case SPF_VAL_GLEV1: return l * (k + 1) * N * 8;
        default: return 0;
        }
    }
    void drop_exec()
    {
        if (exec) (void)hipGraphExecDestroy(exec);
        exec = nullptr;
    }
    void release()
    {
        drop_exec();
        release_outputs();
        runs_since_plan = 0;
        if (d_arena) (void)hipFree(d_arena);
        if (d_stage[0]) (void)hipFree(d_stage[0]);
        if (d_stage[1]) (void)hipFree(d_stage[1]);
        if (d_ptrs) (void)hipFree(d_ptrs);
        d_arena = d_stage[0] = d_stage[1] = nullptr;
        d_ptrs = nullptr;
        planned = false;
    }
};

namespace spf_graph_impl {

struct OpInfo {
    int arity;
    int in_kind[3];
    int out_kind;
};

inline bool op_info(int op, OpInfo* o)
{
    switch (op) {
    case SPF_OP_SAMPLE_EXTRACT: *o = {1, {SPF_VAL_GLWE1, -1, -1}, SPF_VAL_LWE1}; return true;
    case SPF_OP_KEYSWITCH_L1_TO_L0: *o = {1, {SPF_VAL_LWE1, -1, -1}, SPF_VAL_LWE0}; return true;
    case SPF_OP_NOT: *o = {1, {SPF_VAL_GLWE1, -1, -1}, SPF_VAL_GLWE1}; return true;
    case SPF_OP_GLWE_ADD: *o = {2, {SPF_VAL_GLWE1, SPF_VAL_GLWE1, -1}, SPF_VAL_GLWE1}; return true;
    case SPF_OP_CMUX: *o = {3, {SPF_VAL_GGSW1, SPF_VAL_GLWE1, SPF_VAL_GLWE1}, SPF_VAL_GLWE1}; return true;
    case SPF_OP_GLEV_CMUX: *o = {3, {SPF_VAL_GGSW1, SPF_VAL_GLEV1, SPF_VAL_GLEV1}, SPF_VAL_GLEV1}; return true;
    case SPF_OP_MULTIPLY_GGSW_GLWE: *o = {2, {SPF_VAL_GGSW1, SPF_VAL_GLWE1, -1}, SPF_VAL_GLWE1}; return true;
    case SPF_OP_CIRCUIT_BOOTSTRAP: *o = {1, {SPF_VAL_LWE0, -1, -1}, SPF_VAL_GGSW1}; return true;
    case SPF_OP_SCHEME_SWITCH: *o = {1, {SPF_VAL_GLEV1, -1, -1}, SPF_VAL_GGSW1}; return true;
    case SPF_OP_MUL_XN: *o = {1, {SPF_VAL_GLWE1, -1, -1}, SPF_VAL_GLWE1}; return true;
    default: return false;
    }
}

inline bool is_cmux_family(int op)
{
    return op == SPF_OP_CMUX || op == SPF_OP_GLEV_CMUX || op == SPF_OP_MULTIPLY_GGSW_GLWE;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Levels, groups, arena layout, pointer table.  Called by the first run after the graph changed.
inline spf_status plan(spf_graph* g)
{
    spf_ctx* c = g->ctx;
    g->release();
    g->groups.clear();
    HIPCHK(c, hipSetDevice(c->device));
    // inputs and constants first, packed, so one copy brings them up
    size_t off = 0;
    for (auto& n : g->nodes)
        if (n.op < 0) {
            n.level = 0;
            n.off = off;
            off = align_up(off + g->value_bytes(n.kind), 256);
        }
    g->inputs_bytes = off;
    uint32_t max_level = 0;
    for (auto& n : g->nodes)
        if (n.op >= 0) {
            uint32_t lv = 0;
            for (uint32_t i = 0; i < n.n_in; i++) lv = std::max(lv, g->nodes[n.in[i]].level);
            n.level = lv + 1;
            max_level = std::max(max_level, n.level);
        }
    // A circuit bootstrap launch costs as much as ~150 CMUX levels whatever its width (up to one ciphertext per CU),
    // and conversions in the middle of a circuit become ready one by one (the 128 partial-product bits of a 32 x 32
    // multiplication at 32 different depths): batch them.  Within the slack that does not lengthen the graph
    // (latest level = min over consumers of their latest level - 1), bootstraps are moved to common levels —
    // repeatedly take the bootstrap that must run soonest and run every bootstrap that is ready by then with it —
    // and the SampleExtract / KeyswitchL1toL0 nodes feeding them follow.  (Nodes are in topological order: operands
    // precede their users.)
    {
        const size_t nn = g->nodes.size();
        std::vector<uint32_t> alap(nn, max_level);
        for (size_t id = nn; id-- > 0;) {
            const auto& n = g->nodes[id];
            if (n.op < 0) continue;
            for (uint32_t i = 0; i < n.n_in; i++)
                if (g->nodes[n.in[i]].op >= 0) alap[n.in[i]] = std::min(alap[n.in[i]], alap[id] - 1);
        }
        std::vector<uint32_t> cbs;
        for (uint32_t id = 0; id < nn; id++)
            if (g->nodes[id].op == SPF_OP_CIRCUIT_BOOTSTRAP) cbs.push_back(id);
        std::sort(cbs.begin(), cbs.end(), [&](uint32_t a, uint32_t b) { return alap[a] != alap[b] ? alap[a] < alap[b] : a < b; });
        std::vector<uint32_t> fixed(nn, 0);
        std::vector<bool> done(nn, false);
        for (uint32_t lead : cbs) {
            if (done[lead]) continue;
            const uint32_t t = alap[lead];
            for (uint32_t id : cbs)
                if (!done[id] && g->nodes[id].level <= t) { done[id] = true; fixed[id] = t; }
        }
        // forward again with the chosen bootstrap levels as lower bounds (t <= latest level: depth unchanged)
        for (size_t id = 0; id < nn; id++) {
            auto& n = g->nodes[id];
            if (n.op < 0) continue;
            uint32_t lv = 0;
            for (uint32_t i = 0; i < n.n_in; i++) lv = std::max(lv, g->nodes[n.in[i]].level);
            n.level = std::max(lv + 1, fixed[id]);
        }
        // pull the conversion chain's head (SampleExtract -> KeyswitchL1toL0) up against its bootstrap
        std::vector<uint32_t> min_user(nn, 0xffffffffu);
        for (size_t id = nn; id-- > 0;) {
            auto& n = g->nodes[id];
            if (n.op < 0) continue;
            if ((n.op == SPF_OP_KEYSWITCH_L1_TO_L0 || n.op == SPF_OP_SAMPLE_EXTRACT) && min_user[id] != 0xffffffffu)
                n.level = std::max(n.level, min_user[id] - 1);
            for (uint32_t i = 0; i < n.n_in; i++) min_user[n.in[i]] = std::min(min_user[n.in[i]], n.level);
        }
    }
    g->n_levels = max_level;
    // one group per (level, kind, parameter), members in node order
    std::map<std::tuple<uint32_t, int32_t, uint64_t>, size_t> index;
    for (uint32_t id = 0; id < g->nodes.size(); id++) {
        const auto& n = g->nodes[id];
        if (n.op < 0) continue;
        auto key = std::make_tuple(n.level, n.op, n.param);
        auto it = index.find(key);
        if (it == index.end()) {
            it = index.emplace(key, g->groups.size()).first;
            g->groups.emplace_back();
            g->groups.back().op = n.op;
            g->groups.back().param = n.param;
        }
        g->groups[it->second].members.push_back(id);
    }
    // std::map iterates by level first: order the groups the same way
    {
        std::vector<spf_graph::Group> ordered;
        ordered.reserve(g->groups.size());
        for (auto& kv : index) ordered.push_back(std::move(g->groups[kv.second]));
        g->groups.swap(ordered);
    }
    // outputs of a group are consecutive rows
    size_t stage = 0;
    for (auto& gr : g->groups) {
        OpInfo info{};
        op_info(gr.op, &info);
        const size_t ob = g->value_bytes(info.out_kind);
        gr.out_off = off;
        for (size_t i = 0; i < gr.members.size(); i++) g->nodes[gr.members[i]].off = off + i * ob;
        off = align_up(off + gr.members.size() * ob, 256);
    }
    g->arena_bytes = off;
    HIPCHK(c, hipMalloc((void**)&g->d_arena, std::max<size_t>(g->arena_bytes, 256)));
    // operand access per group
    std::vector<void*> table;
    for (auto& gr : g->groups) {
        OpInfo info{};
        op_info(gr.op, &info);
        const size_t B = gr.members.size();
        if (is_cmux_family(gr.op)) {
            // units: one per GLWE pair; {selector, low (a), high (b), out}
            const size_t per = gr.op == SPF_OP_GLEV_CMUX ? g->prm.cbs_radix_count : 1;
            const size_t gw = g->value_bytes(SPF_VAL_GLWE1);
            gr.ptr_index[0] = table.size();
            struct Unit { void* p[4]; };
            std::vector<Unit> units;
            units.reserve(B * per);
            for (size_t i = 0; i < B; i++) {
                const auto& n = g->nodes[gr.members[i]];
                for (size_t j = 0; j < per; j++) {
                    Unit u;
                    u.p[0] = g->d_arena + g->nodes[n.in[0]].off;
                    if (gr.op == SPF_OP_MULTIPLY_GGSW_GLWE) {
                        u.p[1] = nullptr; // zero ciphertext
                        u.p[2] = g->d_arena + g->nodes[n.in[1]].off;
                    } else {
                        u.p[1] = g->d_arena + g->nodes[n.in[1]].off + j * gw;
                        u.p[2] = g->d_arena + g->nodes[n.in[2]].off + j * gw;
                    }
                    u.p[3] = g->d_arena + n.off + j * gw;
                    units.push_back(u);
                }
            }
            // Units that select on the same GGSW go next to each other: a level of a mux_circuits block tests ONE
            // variable (MuxCircuit::from(&[Bdd]), lib.rs:358-445), so a wide level is a handful of selectors with
            // dozens of gates each, and neighbouring workgroups then hit the same 256 KiB in L2 (the streaming kernel
            // does 17.3 M gates/s on four units per selector against 12.9 M/s on distinct ones).  Order inside a
            // level is free: every unit carries its own output pointer.
            std::stable_sort(units.begin(), units.end(), [](const Unit& x, const Unit& y) { return x.p[0] < y.p[0]; });
            for (const Unit& u : units) for (void* q : u.p) table.push_back(q);
            continue;
        }
        for (int s = 0; s < info.arity; s++) {
            const size_t ib = g->value_bytes(info.in_kind[s]);
            bool contig = true;
            const size_t first = g->nodes[g->nodes[gr.members[0]].in[s]].off;
            for (size_t i = 1; i < B && contig; i++)
                contig = g->nodes[g->nodes[gr.members[i]].in[s]].off == first + i * ib;
            gr.contiguous[s] = contig;
            gr.first_off[s] = first;
            if (!contig) {
                gr.ptr_index[s] = table.size();
                for (size_t i = 0; i < B; i++) table.push_back(g->d_arena + g->nodes[g->nodes[gr.members[i]].in[s]].off);
                stage = std::max(stage, B * ib);
            }
        }
    }
    g->stage_bytes = stage;
    if (stage) {
        HIPCHK(c, hipMalloc((void**)&g->d_stage[0], stage));
        HIPCHK(c, hipMalloc((void**)&g->d_stage[1], stage));
    }
    if (!table.empty()) {
        HIPCHK(c, hipMalloc((void**)&g->d_ptrs, table.size() * sizeof(void*)));
        HIPCHK(c, hipMemcpy(g->d_ptrs, table.data(), table.size() * sizeof(void*), hipMemcpyHostToDevice));
    }
    if (getenv("SPF_GRAPH_WIDTHS")) { // diagnostic: CMUX-family launch widths (units) of the plan
        std::map<size_t, size_t> hist;
        for (auto& gr : g->groups)
            if (is_cmux_family(gr.op)) hist[gr.members.size() * (gr.op == SPF_OP_GLEV_CMUX ? g->prm.cbs_radix_count : 1)]++;
        for (auto& kv : hist) fprintf(stderr, "[graph widths] %zu units x %zu launches\n", kv.first, kv.second);
    }
    if (!g->h_inputs.resize(g->inputs_bytes)) return fail(c, SPF_ERR_HIP, "graph: out of host memory for the inputs");
    g->planned = true;
    return SPF_OK;
}

// Everything a run puts on the stream between the input copy and the output copies: the GGSW constants
// (device to device) and one launch per group.  No allocation, no synchronisation: capturable.
inline spf_status enqueue(spf_graph* g, hipStream_t s)
{
    spf_ctx* c = g->ctx;
    for (const auto& n : g->nodes)
        if (n.op == -2 && n.kind == SPF_VAL_GGSW1) {
            const size_t sw = g->value_bytes(SPF_VAL_GGSW1);
            HIPCHK(c, hipMemcpyAsync(g->d_arena + n.off, (const char*)c->d_ggsw_const + (size_t)(n.param & 1) * sw, sw,
                                     hipMemcpyDeviceToDevice, s));
        }
    g->n_launches = 0;
    for (const auto& gr : g->groups) {
        OpInfo info{};
        op_info(gr.op, &info);
        const size_t B = gr.members.size();
        char* out = g->d_arena + gr.out_off;
        if (is_cmux_family(gr.op)) {
            const size_t per = gr.op == SPF_OP_GLEV_CMUX ? g->prm.cbs_radix_count : 1;
            spf_status st = spf_cmux_scattered_dev(c, s, B * per, (const void* const*)(g->d_ptrs + gr.ptr_index[0]));
            if (st != SPF_OK) return st;
            g->n_launches++;
            continue;
        }
        const char* in[2] = {nullptr, nullptr};
        for (int sl = 0; sl < info.arity; sl++) {
            if (gr.contiguous[sl]) {
                in[sl] = g->d_arena + gr.first_off[sl];
            } else {
                spf_status st = spf_gather_rows_dev(c, s, B, g->value_bytes(info.in_kind[sl]) / 8,
                                                    (const uint64_t* const*)(g->d_ptrs + gr.ptr_index[sl]),
                                                    (uint64_t*)g->d_stage[sl]);
                if (st != SPF_OK) return st;
                g->n_launches++;
                in[sl] = g->d_stage[sl];
            }
        }
        spf_status st = SPF_OK;
        switch (gr.op) {
        case SPF_OP_SAMPLE_EXTRACT:
            st = spf_sample_extract_l1_dev(c, s, B, (const uint64_t*)in[0], (size_t)gr.param, (uint64_t*)out);
            break;
        case SPF_OP_KEYSWITCH_L1_TO_L0:
            st = spf_keyswitch_lwe_l1_lwe_l0_dev(c, s, B, (const uint64_t*)in[0], (uint64_t*)out);
            break;
        case SPF_OP_CIRCUIT_BOOTSTRAP:
            st = spf_circuit_bootstrap_dev(c, s, B, (const uint64_t*)in[0], (double*)out);
            break;
        case SPF_OP_SCHEME_SWITCH:
            st = spf_scheme_switch_dev(c, s, B, (const uint64_t*)in[0], (double*)out);
            break;
        case SPF_OP_NOT:
            st = spf_glwe_not_dev(c, s, B, (const uint64_t*)in[0], (uint64_t*)out);
            break;
        case SPF_OP_GLWE_ADD:
            st = spf_glwe_xor_dev(c, s, B, (const uint64_t*)in[0], (const uint64_t*)in[1], (uint64_t*)out);
            break;
        case SPF_OP_MUL_XN:
            st = spf_glwe_mul_xn_dev(c, s, B, (const uint64_t*)in[0], (size_t)gr.param, (uint64_t*)out);
            break;
        default:
            st = fail(c, SPF_ERR_INVALID_ARGUMENT, "unknown graph operation");
        }
        if (st != SPF_OK) return st;
        g->n_launches++;
    }
    return SPF_OK;
}

// Output staging (after plan(): needs the arena offsets): outputs grouped by value size, one pointer row per group
inline spf_status plan_outputs(spf_graph* g)
{
    spf_ctx* c = g->ctx;
    g->release_outputs();
    std::map<size_t, spf_graph::OutClass> by_words;
    for (size_t i = 0; i < g->outputs.size(); i++) {
        const auto& n = g->nodes[g->outputs[i].first];
        auto& oc = by_words[g->value_bytes(n.kind) / 8];
        oc.words = g->value_bytes(n.kind) / 8;
        oc.which.push_back(i);
    }
    std::vector<void*> table;
    size_t off = 0;
    for (auto& kv : by_words) {
        auto& oc = kv.second;
        oc.count = oc.which.size();
        oc.ptr_index = table.size();
        oc.stage_off = off;
        for (size_t i : oc.which) table.push_back(g->d_arena + g->nodes[g->outputs[i].first].off);
        off = align_up(off + oc.count * oc.words * 8, 256);
        g->out_classes.push_back(oc);
    }
    g->out_bytes = off;
    if (off) {
        HIPCHK(c, hipMalloc((void**)&g->d_out_stage, off));
        HIPCHK(c, hipMalloc((void**)&g->d_out_ptrs, table.size() * sizeof(void*)));
        HIPCHK(c, hipMemcpy(g->d_out_ptrs, table.data(), table.size() * sizeof(void*), hipMemcpyHostToDevice));
        if (!g->h_out_stage.resize(off)) return fail(c, SPF_ERR_HIP, "graph: out of host memory for the outputs");
    }
    g->outputs_planned = g->outputs.size();
    return SPF_OK;
}

inline spf_status run(spf_graph* g)
{
    spf_ctx* c = g->ctx;
    // one graph at a time per context: the _dev calls below share the context's scratch buffers and
    // stream, and a run must not interleave its launches with another run's
    std::lock_guard<std::recursive_mutex> whole(c->mu);
    if (!g->planned) {
        spf_status st = plan(g);
        if (st != SPF_OK) return st;
    }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const size_t k = g->prm.glwe_size, N = g->prm.polynomial_degree;
    // inputs: read the callers' buffers now (they may have changed since the last run)
    for (const auto& n : g->nodes) {
        if (n.op == -1) {
            std::memcpy(g->h_inputs.data() + n.off, n.host, g->value_bytes(n.kind));
        } else if (n.op == -2 && n.kind == SPF_VAL_GLEV1) {
            // trivial_glev_l1_{zero,one} (crypto/encryption.rs:434-451 -> trivially_encrypt_glev_ciphertext,
            // ops/encryption/glev_encryption.rs:23-80): GLWE j = zero mask, body = bit * q / B^(j+1) at coefficient 0
            uint64_t* v = reinterpret_cast<uint64_t*>(g->h_inputs.data() + n.off);
            std::memset(v, 0, g->value_bytes(n.kind));
            for (size_t j = 0; j < g->prm.cbs_radix_count; j++)
                v[j * (k + 1) * N + k * N] = (n.param & 1) << (64 - g->prm.cbs_radix_log * (j + 1));
        } else if (n.op == -2 && n.kind != SPF_VAL_GGSW1) {
            // trivial_lwe / trivial_glwe of a bit at one plaintext bit (crypto/encryption.rs:345-412):
            // zero mask, body (coefficient 0) = bit << 63
            uint64_t* v = reinterpret_cast<uint64_t*>(g->h_inputs.data() + n.off);
            std::memset(v, 0, g->value_bytes(n.kind));
            const size_t body = n.kind == SPF_VAL_LWE0 ? g->prm.lwe_dimension : k * N;
            v[body] = (n.param & 1) << 63;
        }
    }
    if (g->inputs_bytes) HIPCHK(c, hipMemcpyAsync(g->d_arena, g->h_inputs.data(), g->inputs_bytes, hipMemcpyHostToDevice, s));
    bool has_ggsw_const = false;
    for (const auto& n : g->nodes) has_ggsw_const = has_ggsw_const || (n.op == -2 && n.kind == SPF_VAL_GGSW1);
    if (has_ggsw_const) { // built (and synchronised) outside any capture
        spf_status st = ensure_ggsw_constants(c);
        if (st != SPF_OK) return st;
    }
    // Off unless SPF_GRAPH_CAPTURE=1: measured on MI355X / ROCm 7.2 the replay is SLOWER than enqueueing the ~100
    // launches (32-bit addition 8.00 ms replayed vs 7.67 ms eager; 16 additions 24.8 vs 24.1 ms) — the launches
    // are already back to back on one stream and hipGraphLaunch adds more than it removes.  (r05 re-measured: 5.11 ms either way
    // for the addition, 42.6 against 43.0 ms for four 32 x 32 multiplications: still opt-in.)
    static const bool capture_on = [] { const char* e = getenv("SPF_GRAPH_CAPTURE"); return e && e[0] == '1'; }();
    g->runs_since_plan++;
    if (g->exec && g->exec_epoch != c->buf_epoch) g->drop_exec(); // a scratch buffer moved since the capture
    if (g->exec) {
        HIPCHK(c, hipGraphLaunch(g->exec, s));
        g->n_launches = g->captured_launches;
    } else {
        const bool capture = capture_on && !c->timing && g->runs_since_plan >= 2;
        const uint64_t epoch0 = c->buf_epoch;
        if (capture) HIPCHK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        spf_status st = enqueue(g, s);
        if (capture) {
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamEndCapture(s, &graph);
            if (st == SPF_OK && e == hipSuccess && graph && epoch0 == c->buf_epoch &&
                hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                g->exec_epoch = c->buf_epoch;
                g->captured_launches = g->n_launches;
            } else {
                g->exec = nullptr;
            }
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            if (st != SPF_OK) return st;
            if (g->exec) HIPCHK(c, hipGraphLaunch(g->exec, s));
            else {                       // the capture did not take (a buffer grew, ...): run it plainly this time
                st = enqueue(g, s);
                if (st != SPF_OK) return st;
            }
        } else if (st != SPF_OK) return st;
    }
    if (g->outputs_planned != g->outputs.size()) {
        spf_status st = plan_outputs(g);
        if (st != SPF_OK) return st;
    }
    for (const auto& oc : g->out_classes) {
        spf_status st = spf_gather_rows_dev(c, s, oc.count, oc.words, (const uint64_t* const*)(g->d_out_ptrs + oc.ptr_index),
                                            (uint64_t*)(g->d_out_stage + oc.stage_off));
        if (st != SPF_OK) return st;
    }
    if (g->out_bytes) HIPCHK(c, hipMemcpyAsync(g->h_out_stage.data(), g->d_out_stage, g->out_bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    for (const auto& oc : g->out_classes)
        for (size_t i = 0; i < oc.count; i++)
            std::memcpy(g->outputs[oc.which[i]].second, g->h_out_stage.data() + oc.stage_off + i * oc.words * 8, oc.words * 8);
    return SPF_OK;
}

} // namespace spf_graph_impl

static spf_status group_run_graphs(struct spf_group* grp, spf_graph* const* graphs, size_t n); // spf_group.hpp
static void group_forget_graph(struct spf_group* grp, spf_graph* graph);

spf_status spf_graph_create(spf_ctx* c, spf_graph** out)
{
    if (!c || !out) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    spf_graph* g = new (std::nothrow) spf_graph();
    if (!g) return fail(c, SPF_ERR_HIP, "out of host memory");
    g->ctx = c;
    g->prm = c->prm;
    *out = g;
    return SPF_OK;
}

void spf_graph_destroy(spf_graph* g)
{
    if (!g) return;
    if (g->grp) group_forget_graph(g->grp, g); // (a merged graph of the group may still read this job's buffers)
    (void)hipSetDevice(g->ctx->device);
    g->release();
    g->h_inputs.free_buf();
    delete g;
}

spf_status spf_graph_add_input(spf_graph* g, spf_value_kind kind, const void* host, uint32_t* node)
{
    if (!g) return SPF_ERR_INVALID_ARGUMENT;
    if (!host || !node || g->value_bytes(kind) == 0) return fail(g->ctx, SPF_ERR_INVALID_ARGUMENT, "graph input: bad kind or null pointer");
    spf_graph::Node n{};
    n.op = -1; n.kind = kind; n.host = host;
    *node = (uint32_t)g->nodes.size();
    g->nodes.push_back(n);
    g->planned = false;
    return SPF_OK;
}

spf_status spf_graph_add_trivial(spf_graph* g, spf_value_kind kind, uint64_t bit, uint32_t* node)
{
    if (!g) return SPF_ERR_INVALID_ARGUMENT;
    if (!node || g->value_bytes(kind) == 0 || bit > 1)
        return fail(g->ctx, SPF_ERR_INVALID_ARGUMENT, "graph constant: LWE0 / LWE1 / GLWE1 / GGSW1 / GLEV1 of bit 0 or 1");
    spf_graph::Node n{};
    n.op = -2; n.kind = kind; n.param = bit;
    *node = (uint32_t)g->nodes.size();
    g->nodes.push_back(n);
    g->planned = false;
    return SPF_OK;
}

// Validation happens here, before anything runs, as the reference's graph-level `RuntimeError`s do
// (task.rs:26-31): wrong arity or operand type is an error of the call, not of the run.
spf_status spf_graph_add_op(spf_graph* g, spf_graph_op op, const uint32_t* inputs, size_t n_inputs, uint64_t param,
                            uint32_t* node)
{
    if (!g) return SPF_ERR_INVALID_ARGUMENT;
    spf_graph_impl::OpInfo info{};
    if (!node || !spf_graph_impl::op_info(op, &info)) return fail(g->ctx, SPF_ERR_INVALID_ARGUMENT, "graph op: unknown operation");
    if (n_inputs != (size_t)info.arity || (n_inputs && !inputs))
        return fail(g->ctx, SPF_ERR_INVALID_ARGUMENT, "graph op: wrong number of operands");
    spf_graph::Node n{};
    n.op = op; n.kind = info.out_kind; n.n_in = (uint32_t)info.arity;
    for (int i = 0; i < info.arity; i++) {
        if (inputs[i] >= g->nodes.size()) return fail(g->ctx, SPF_ERR_INVALID_ARGUMENT, "graph op: operand is not a node of this graph");
        if (g->nodes[inputs[i]].kind != info.in_kind[i]) return fail(g->ctx, SPF_ERR_INVALID_ARGUMENT, "graph op: operand has the wrong ciphertext type");
        n.in[i] = inputs[i];
    }
    if (op == SPF_OP_SAMPLE_EXTRACT && param >= g->prm.polynomial_degree)
        return fail(g->ctx, SPF_ERR_INVALID_ARGUMENT, "graph op: sample_extract index >= polynomial_degree");
    n.param = (op == SPF_OP_SAMPLE_EXTRACT) ? param : (op == SPF_OP_MUL_XN ? param % (2 * (uint64_t)g->prm.polynomial_degree) : 0);
    *node = (uint32_t)g->nodes.size();
    g->nodes.push_back(n);
    g->planned = false;
    return SPF_OK;
}

spf_status spf_graph_add_output(spf_graph* g, uint32_t node, void* host)
{
    if (!g) return SPF_ERR_INVALID_ARGUMENT;
    if (!host || node >= g->nodes.size()) return fail(g->ctx, SPF_ERR_INVALID_ARGUMENT, "graph output: bad node or null pointer");
    g->outputs.emplace_back(node, host);
    return SPF_OK;
}


spf_status spf_graph_run(spf_graph* g)
{
    if (!g) return SPF_ERR_INVALID_ARGUMENT;
    if (g->grp) return group_run_graphs(g->grp, &g, 1); // a group's graph: placed and run by the group
    return spf_graph_impl::run(g);
}

spf_status spf_graph_stats(spf_graph* g, uint32_t* nodes, uint32_t* levels, uint32_t* launches)
{
    if (!g) return SPF_ERR_INVALID_ARGUMENT;
    if (nodes) *nodes = (uint32_t)g->nodes.size();
    if (levels) *levels = g->n_levels;
    if (launches) *launches = g->n_launches;
    return SPF_OK;
}
