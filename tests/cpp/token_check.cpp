// Every GFA node token gfa_tokens.hpp can make, against snprintf: all token lengths, both orientations, the three forms (W-line, P-line,
// first position of a P-line), by both builders, OR-ed in at every byte alignment between neighbours that must stay as they were.
#include "gfa_tokens.hpp"

#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

static unsigned long checked = 0;

static bool check(uint32_t node, bool p_lines, bool first) {
    char want[32];
    const unsigned id = node >> 1, rev = node & 1u;
    int n;
    if (p_lines) n = std::snprintf(want, sizeof(want), "%s%u%c", first ? "" : ",", id, rev ? '-' : '+');
    else n = std::snprintf(want, sizeof(want), "%c%u", rev ? '<' : '>', id);
    for (int form = 0; form < 2; form++) {
        if (form == 1 && id >= 100000000u) continue;             // (make_token_short: ids below 10^8)
        const gbwt_hip::Token t = form == 0 ? gbwt_hip::make_token(node, p_lines, first) : gbwt_hip::make_token_short(node, p_lines, first);
        if (static_cast<int>(t.len) != n) { std::printf("node %u p %d first %d form %d: length %u, expected %d\n", node, p_lines, first, form, t.len, n); return false; }
        uint8_t bytes[12];
        std::memcpy(bytes, t.w, 12);
        for (unsigned k = 0; k < 12; k++)
            if (bytes[k] != (k < t.len ? static_cast<uint8_t>(want[k]) : 0)) { std::printf("node %u p %d first %d form %d: byte %u is %02x\n", node, p_lines, first, form, k, bytes[k]); return false; }
        // placed by OR at every alignment, between neighbours that are already there
        for (unsigned at = 0; at < 16; at++) {
            alignas(4) uint8_t buf[64];
            std::memset(buf, 0, sizeof(buf));
            for (unsigned k = 4; k < 8 + at; k++) buf[k] = 0x5A;                                     // the neighbour in front, up to the token's first byte
            for (unsigned k = 8 + at + t.len; k < 8 + at + t.len + 5; k++) buf[k] = 0x3C;            // the neighbour behind
            uint8_t expect[64];
            std::memcpy(expect, buf, sizeof(buf));
            std::memcpy(expect + 8 + at, want, t.len);
            uint32_t out[4], dwords[16];
            gbwt_hip::spread_token(t, (8 + at) & 3u, out);
            std::memcpy(dwords, buf, sizeof(buf));
            for (unsigned k = 0; k < 4; k++) dwords[(8 + at) / 4 + k] |= out[k];
            if (std::memcmp(dwords, expect, sizeof(buf)) != 0) { std::printf("node %u p %d first %d form %d at %u: placed wrongly\n", node, p_lines, first, form, at); return false; }
        }
    }
    checked++;
    return true;
}

int main() {
    std::vector<uint32_t> ids = {0, 1, 2, 9};
    for (uint64_t p = 10; p <= 1000000000ull; p *= 10) for (int d = -2; d <= 2; d++) ids.push_back(static_cast<uint32_t>(p + d));
    for (uint32_t v : {99u, 100u, 101u, 9999u, 10000u, 10001u, 99999999u, 100000000u, 100000001u, 123456789u, 987654321u, 1999999999u, 2000000000u, 2147483646u, 2147483647u}) ids.push_back(v);
    std::mt19937_64 rng(42);
    for (int k = 0; k < 200000; k++) { const unsigned bits = 1 + rng() % 31; ids.push_back(static_cast<uint32_t>(rng() & ((1ull << bits) - 1))); }
    for (uint32_t x = 0; x < 10000; x++) {                       // the digit arithmetic on its whole domain
        const uint32_t got = gbwt_hip::four_digits(x), want = (x / 1000) | ((x / 100 % 10) << 8) | ((x / 10 % 10) << 16) | ((x % 10) << 24);
        if (got != want) { std::printf("four_digits(%u) = %08x, expected %08x\n", x, got, want); return 1; }
    }
    for (uint32_t id : ids)
        for (uint32_t rev = 0; rev < 2; rev++) {
            const uint32_t node = (id << 1) | rev;
            if ((node >> 1) != id) continue;
            if (!check(node, false, false) || !check(node, false, true) || !check(node, true, false) || !check(node, true, true)) return 1;
        }
    std::printf("tokens checked: %lu\n", checked);
    return 0;
}
