// The token builders of gfa_tokens.hpp ON THE DEVICE (the byte-align instruction instead of its host stand-in), against snprintf: ids of every
// length up to 2^31 - 1, both orientations, the three forms; both builders where both apply.  Built and run by tests/test_gpu_gfa.py.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "gfa_tokens.hpp"

__global__ void k_tokens(const uint32_t *nodes, uint32_t n, uint32_t *out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t node = nodes[k];
    for (uint32_t form = 0; form < 6; form++) {                  // (W, P, first of P) x (general, short)
        const bool p_lines = form % 3 != 0, first = form % 3 == 2, small = form >= 3;
        gbwt_hip::Token t{{0u, 0u, 0u}, 0u};
        if (!small) t = gbwt_hip::make_token(node, p_lines, first);
        else if ((node >> 1) < 100000000u) t = gbwt_hip::make_token_short(node, p_lines, first);
        uint32_t spread[4];
        gbwt_hip::spread_token(t, k & 3u, spread);
        uint32_t *o = out + (static_cast<size_t>(k) * 6 + form) * 8;
        o[0] = t.w[0]; o[1] = t.w[1]; o[2] = t.w[2]; o[3] = t.len;
        o[4] = spread[0]; o[5] = spread[1]; o[6] = spread[2]; o[7] = spread[3];
    }
}

int main() {
    std::vector<uint32_t> ids = {0, 1, 2, 9};
    for (uint64_t p = 10; p <= 1000000000ull; p *= 10) for (int d = -2; d <= 2; d++) ids.push_back(static_cast<uint32_t>(p + d));
    for (uint32_t v : {99999999u, 100000000u, 123456789u, 987654321u, 1999999999u, 2000000000u, 2147483646u, 2147483647u}) ids.push_back(v);
    std::mt19937_64 rng(7);
    for (int k = 0; k < 500000; k++) { const unsigned bits = 1 + rng() % 31; ids.push_back(static_cast<uint32_t>(rng() & ((1ull << bits) - 1))); }
    std::vector<uint32_t> nodes;
    for (uint32_t id : ids) for (uint32_t rev = 0; rev < 2; rev++) nodes.push_back((id << 1) | rev);
    const uint32_t n = static_cast<uint32_t>(nodes.size());
    uint32_t *d_nodes = nullptr, *d_out = nullptr;
    if (hipMalloc(&d_nodes, n * 4) != hipSuccess || hipMalloc(&d_out, static_cast<size_t>(n) * 6 * 8 * 4) != hipSuccess) { std::printf("hipMalloc failed\n"); return 2; }
    if (hipMemcpy(d_nodes, nodes.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess) return 2;
    hipLaunchKernelGGL(k_tokens, dim3((n + 255) / 256), dim3(256), 0, nullptr, d_nodes, n, d_out);
    std::vector<uint32_t> out(static_cast<size_t>(n) * 6 * 8);
    if (hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { std::printf("kernel or copy failed\n"); return 2; }
    unsigned long checked = 0;
    for (uint32_t k = 0; k < n; k++)
        for (uint32_t form = 0; form < 6; form++) {
            const bool p_lines = form % 3 != 0, first = form % 3 == 2, small = form >= 3;
            const unsigned id = nodes[k] >> 1, rev = nodes[k] & 1u;
            if (small && id >= 100000000u) continue;
            char want[32] = {0};
            int len;
            if (p_lines) len = std::snprintf(want, sizeof(want), "%s%u%c", first ? "" : ",", id, rev ? '-' : '+');
            else len = std::snprintf(want, sizeof(want), "%c%u", rev ? '<' : '>', id);
            const uint32_t *o = out.data() + (static_cast<size_t>(k) * 6 + form) * 8;
            uint8_t expect[16] = {0}, placed[16] = {0};
            std::memcpy(expect, want, len);
            std::memcpy(placed + (k & 3u), want, len);
            if (static_cast<int>(o[3]) != len || std::memcmp(o, expect, 12) != 0 || std::memcmp(o + 4, placed, 16) != 0) {
                std::printf("node %u form %u: device token differs from \"%s\"\n", nodes[k], form, want);
                return 1;
            }
            checked++;
        }
    std::printf("device tokens checked: %lu\n", checked);
    return 0;
}
