// mutate_loader.cpp -- the product's file loader (gbwt_rs_amd/csrc/host_index.cpp) under AddressSanitizer + UBSan: every
// 64-bit element of the given files overwritten with a set of hostile values; each variant must either load or throw
// InvalidData.  Built and run by tests/test_capi_cpu.py::test_loader_under_sanitizers (CPU only).
// usage: mutate_loader SCRATCH_FILE FILE...
#include "host_index.hpp"
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
int main(int argc, char **argv) {
    const uint64_t values[] = {~0ull, 1ull << 63, 0xFFFFFFFFFFFFFFC1ull, 1ull << 40, 0x7FFFFFFFFFFFFFFFull, 0, 1, 65, 1ull<<36};
    long ok = 0, bad = 0;
    const char *scratch = argv[1];
    for (int a = 2; a < argc; a++) {
        std::ifstream f(argv[a], std::ios::binary);
        std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        for (uint64_t v : values)
            for (size_t w = 0; w < raw.size() / 8; w++) {
                std::vector<char> m = raw;
                std::memcpy(m.data() + 8 * w, &v, 8);
                std::ofstream(scratch, std::ios::binary).write(m.data(), m.size());
                try { gbwt_hip::load_index_file(scratch); ok++; }
                catch (const gbwt_hip::InvalidData &) { bad++; }
            }
    }
    std::printf("accepted %ld rejected %ld\n", ok, bad);
}
