// The reference's own tests, restated against the C++ mirror of its interface (include/gbwt_hip.hpp):
//   statistics / extract / backward      src/gbwt/tests.rs:100-214
//   find / extend / bidirectional search src/gbwt/tests.rs:218-350 and the doc-tests src/gbwt.rs:70-83
//   GBZ::path, StateIter                 src/gbz/tests.rs:60-168 and the doc-test src/gbz.rs:1184-1209
// Known answers are the fixture's true paths (src/gbwt/tests.rs:16-40).  Usage: test_reference_api <golden dir>.
// Without a HIP device the library has no fallback: the program then checks for GBWT_HIP_NO_DEVICE and says so.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "gbwt_hip.hpp"

using namespace gbwt_hip;

#define REQUIRE(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } } while (0)

static uint64_t count_occurrences(const std::vector<std::vector<uint32_t>> &paths, const std::vector<uint32_t> &q) {
    uint64_t n = 0;
    for (const auto &p : paths)
        for (size_t i = 0; i + q.size() <= p.size(); i++) {
            bool same = true;
            for (size_t j = 0; j < q.size(); j++) same &= p[i + j] == q[j];
            n += same;
        }
    return n;
}

int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "tests/golden";
    if (gbwt_hip_device_count() == 0) {
        try {
            GBWT index(dir + "/example.gbwt");
            REQUIRE(!"opened an index without a device");
        } catch (const Error &e) {
            REQUIRE(e.status == GBWT_HIP_NO_DEVICE);
            std::printf("no HIP device: GBWT_HIP_NO_DEVICE as documented (no CPU fallback)\n");
            return 0;
        }
    }
    GBWT index(dir + "/example.gbwt");
    // statistics, src/gbwt/tests.rs:100-118
    REQUIRE(index.len() == 68 && index.sequences() == 12 && index.alphabet_size() == 52 && index.alphabet_offset() == 21);
    REQUIRE(index.effective_size() == 31 && index.first_node() == 22 && index.is_bidirectional() && !index.is_empty());
    REQUIRE(!index.has_node(21) && index.has_node(22) && index.has_node(51) && !index.has_node(52));
    // all sequences in both orientations, src/gbwt/tests.rs:120-184
    // the fixture's twelve sequences (true_paths, src/gbwt/tests.rs:116-170, both orientations)
    const std::vector<std::vector<uint32_t>> truth = {
        {22, 24, 28, 30, 34}, {35, 31, 29, 25, 23}, {42, 44, 48, 50}, {51, 49, 45, 43}, {22, 24, 28, 30, 34}, {35, 31, 29, 25, 23},
        {22, 26, 28, 32, 34}, {35, 33, 29, 27, 23}, {42, 44, 48, 47, 43}, {42, 46, 49, 45, 43}, {42, 44, 48, 50}, {51, 49, 45, 43},
    };
    for (uint64_t id = 0; id < index.sequences(); id++) {
        const auto seq = index.sequence(id);
        REQUIRE(seq && *seq == truth[id]);
        // the same through start / forward, src/gbwt/tests.rs:130-150
        std::vector<uint32_t> walked;
        for (auto pos = index.start(id); pos; pos = index.forward(*pos)) walked.push_back(static_cast<uint32_t>(pos->node));
        REQUIRE(walked == truth[id]);
        // and backward from the last position, src/gbwt/tests.rs:191-205
        std::vector<uint32_t> back;
        auto pos = index.start(id), last = pos;
        while (pos) { last = pos; pos = index.forward(*pos); }
        for (pos = last; pos; pos = index.backward(*pos)) back.push_back(static_cast<uint32_t>(pos->node));
        const std::vector<uint32_t> expect(truth[id].rbegin(), truth[id].rend());
        REQUIRE(back == expect);
    }
    REQUIRE(!index.sequence(index.sequences()));
    REQUIRE(!index.start(index.sequences()));
    // the stretches of every row that 1, 2 and 3 GPUs sharing the batch would walk, back to back, are the rows
    {
        std::vector<uint64_t> all(index.sequences());
        for (uint64_t id = 0; id < all.size(); id++) all[id] = id;
        for (uint32_t parts = 1; parts <= 3; parts++) {
            std::vector<std::vector<uint32_t>> joined(all.size());
            for (uint32_t part = 0; part < parts; part++) {
                const gbwt_hip::Rows rows = index.sequences_part(all, part, parts);
                for (size_t k = 0; k < all.size(); k++) { const auto piece = rows.row(k); joined[k].insert(joined[k].end(), piece.begin(), piece.end()); }
            }
            REQUIRE(joined == truth);
        }
    }
    // doc-test src/gbwt.rs:70-83
    auto state = index.find(24);
    REQUIRE(state);
    state = index.extend(*state, 28);
    REQUIRE(state);
    state = index.extend(*state, 30);
    REQUIRE(state && state->node == 30 && state->end - state->start == 2);
    auto bd = index.bd_find(28);
    REQUIRE(bd);
    bd = index.extend_backward(*bd, 24);
    REQUIRE(bd);
    bd = index.extend_forward(*bd, 30);
    REQUIRE(bd && bd->forward.node == 30 && bd->forward.end - bd->forward.start == 2 && bd->reverse.node == 25);
    // every substring of length <= 3 of every sequence against brute-force counts, src/gbwt/tests.rs:294-350
    for (const auto &p : truth)
        for (size_t len = 1; len <= 3; len++)
            for (size_t i = 0; i + len <= p.size(); i++) {
                const std::vector<uint32_t> q(p.begin() + i, p.begin() + i + len);
                auto st = index.find(q[0]);
                for (size_t j = 1; j < q.size() && st; j++) st = index.extend(*st, q[j]);
                REQUIRE(st && st->end - st->start == count_occurrences(truth, q));
            }
    REQUIRE(!index.find(0) && !index.find(21) && !index.find(52));
    REQUIRE(!index.extend(*index.find(22), 28));

    // GBZ::path and StateIter, src/gbz/tests.rs:60-168, doc-test src/gbz.rs:1184-1209
    GBZ gbz(dir + "/example.gbz");
    REQUIRE(gbz.paths() == 6);
    const auto p3 = gbz.path(3, Orientation::Reverse);
    REQUIRE(p3 && p3->size() == 5 && (*p3)[0] == std::make_pair(uint64_t(17), Orientation::Reverse) && (*p3)[4] == std::make_pair(uint64_t(11), Orientation::Reverse));
    REQUIRE(!gbz.path(6, Orientation::Forward));
    const auto s14 = gbz.search_state(14, Orientation::Forward);
    REQUIRE(s14 && s14->forward.end - s14->forward.start == 3);
    const auto successors = gbz.follow_forward(*s14);
    REQUIRE(successors && successors->size() == 2);
    std::vector<std::pair<uint64_t, uint64_t>> preds;
    for (const auto &succ : *successors) {
        const auto back = gbz.follow_backward(succ);
        REQUIRE(back);
        for (const auto &p : *back) preds.emplace_back(node_id(flip_node(p.reverse.node)) * 100 + node_id(p.forward.node), p.forward.end - p.forward.start);
    }
    REQUIRE(preds.size() == 2 && preds[0] == std::make_pair(uint64_t(12 * 100 + 15), uint64_t(2)) && preds[1] == std::make_pair(uint64_t(13 * 100 + 16), uint64_t(1)));
    REQUIRE(gbz.path_lines({0}, 0) == "P\tA\t11+,12+,14+,15+,17+\t*\n");
    REQUIRE(gbz.path_lines({2}, 1) == "W\tsample\t1\tA\t0\t5\t>11>12>14>15>17\n");
    // Metadata::pan_sn_path, doc-test src/gbwt.rs:596-598
    REQUIRE(gbz.pan_sn_path(3) == "sample#2#A");
    REQUIRE(gbz.path_lines({3}, 2) == "P\tsample#2#A\t11+,13+,14+,16+,17+\t*\n");
    // GBZ::segment_path, src/gbz/tests.rs:466-497 (three paths of translation.gbz as segment identifiers; reverse = reversed and flipped; past the
    // end and without a translation: None) and the doc-test src/gbz.rs:1080-1092 (path 2 reversed starts with s17-, s16-)
    {
        GBZ translated(dir + "/translation.gbz");
        const std::vector<std::vector<uint64_t>> truth = {{0, 1, 3, 5, 7}, {0, 1, 3, 5, 7}, {0, 2, 3, 6, 7}};
        REQUIRE(translated.paths() == truth.size());
        for (uint64_t p = 0; p < truth.size(); p++) {
            const auto f = translated.segment_path(p, Orientation::Forward), r = translated.segment_path(p, Orientation::Reverse);
            REQUIRE(f && r && f->size() == truth[p].size() && r->size() == truth[p].size());
            for (size_t k = 0; k < truth[p].size(); k++) {
                REQUIRE((*f)[k] == std::make_pair(truth[p][k], Orientation::Forward));
                REQUIRE((*r)[k] == std::make_pair(truth[p][truth[p].size() - 1 - k], Orientation::Reverse));
            }
        }
        REQUIRE(!translated.segment_path(truth.size(), Orientation::Forward) && !translated.segment_path(truth.size(), Orientation::Reverse));
        REQUIRE(!gbz.segment_path(0, Orientation::Forward));                 // example.gbz has no node-to-segment translation
    }
    // a unidirectional index refuses what the reference asserts on (src/gbwt.rs:237,312)
    try {
        GBWT bad(dir + "/does-not-exist.gbwt");
        REQUIRE(!"opened a missing file");
    } catch (const Error &e) {
        REQUIRE(e.status == GBWT_HIP_IO_ERROR);
    }
    // what an index replica costs (gbwt_hip_memory_usage), and the exchange of a sharded extraction with one rank: the rows of an
    // extraction gathered "from all ranks" in path order are the rows themselves (gbwt_hip_comm_*: RCCL loaded at run time)
    const gbwt_hip_memory mem = gbz.memory_usage();
    REQUIRE(mem.index_device_bytes > 141 && mem.index_host_bytes > 141);
    try {
        Comm comm(Comm::unique_id(), 0, 1, 0);
        REQUIRE(comm.get() != nullptr);
    } catch (const Error &e) {
        REQUIRE(e.status == GBWT_HIP_UNSUPPORTED);                       // a box without RCCL
    }
    std::printf("reference API mirror: all checks passed\n");
    return 0;
}
