// The node tokens of GFA P- and W-lines as register arithmetic: a token is put together in three dwords (its digits by a handful of
// multiplications, no loop) and is OR-ed into a zeroed staging buffer as aligned dwords, instead of one store per character.  What is formatted: src/bin/gbunzip.rs:462-476 (P-lines: "<id>+" / "<id>-" joined by ',') and :542-547
// (W-lines: ">" / "<" + id), ids in decimal (Rust's Display for usize).
// Plain C++ apart from one byte-align instruction: tests/test_capi_cpu.py compiles this header for the host and checks every token length
// and alignment against snprintf.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define GBWT_HIP_TOKEN_FN __host__ __device__ __forceinline__
#else
#define GBWT_HIP_TOKEN_FN inline
#endif
#if defined(__clang__)
#define GBWT_HIP_TOKEN_KNOWN(condition) __builtin_assume(condition)     // value ranges the compiler cannot see: 24-bit multiplications are full rate
#else
#define GBWT_HIP_TOKEN_KNOWN(condition) ((void)0)
#endif

namespace gbwt_hip {

// Twelve bytes, byte k of the token = byte k % 4 of w[k / 4]; bytes from `len` on are zero.
struct Token {
    uint32_t w[3];
    uint32_t len;
};

// bytes r .. r + 3 of the eight bytes hi:lo, r = 0 .. 3
GBWT_HIP_TOKEN_FN uint32_t bytes_from(uint32_t hi, uint32_t lo, uint32_t r) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, r);
#else
    return static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> (8 * (r & 3u)));
#endif
}

GBWT_HIP_TOKEN_FN uint32_t token_digits(uint32_t v) {
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) +
           (v >= 100000000u) + (v >= 1000000000u);
}

// The four decimal digits of x < 10 000 as four bytes, the most significant first (values 0 .. 9, not yet characters): x / 100 and x % 100
// side by side in the halves of a dword, their tens by one multiplication ((y * 103) >> 10 = y / 10 for y < 100), their ones by another.
GBWT_HIP_TOKEN_FN uint32_t four_digits(uint32_t x) {
    GBWT_HIP_TOKEN_KNOWN(x < 10000u);
    const uint32_t a = (x * 5243u) >> 19;                       // x / 100 (exact below 43 699)
    const uint32_t pair = a | ((x - 100u * a) << 16);
    GBWT_HIP_TOKEN_KNOWN(pair < (100u << 16));
    const uint32_t tens = ((pair * 103u) >> 10) & 0x000F000Fu;
    return tens + ((pair - 10u * tens) << 8);
}

// The token of one path position.  `node` = GBWT node (2 * id + orientation); `lead`: P-lines ',' (0 for the first position of a path:
// no separator), W-lines '>' / '<'; `trail`: P-lines '+' / '-', W-lines 0.
GBWT_HIP_TOKEN_FN Token make_token(uint32_t node, bool p_lines, bool first_of_path) {
    const uint32_t v = node >> 1, rev = node & 1u;
    const uint32_t digits = token_digits(v);
    // the ten digits with leading zeros in bytes 2 .. 11 of a frame of twelve
    const uint32_t top = v / 100000000u, rest = v - top * 100000000u, high = rest / 10000u, low = rest - high * 10000u;
    const uint32_t top_tens = (top * 103u) >> 10;
    const uint32_t f0 = ((top_tens | ((top - 10u * top_tens) << 8)) + 0x3030u) << 16;
    const uint32_t f1 = four_digits(high) + 0x30303030u, f2 = four_digits(low) + 0x30303030u;
    // the token starts `cut` bytes into the frame: at the byte in front of the first digit (to become the lead character), or at the digit
    const bool no_lead = p_lines && first_of_path;
    const uint32_t cut = (no_lead ? 12u : 11u) - digits;         // 1 .. 11
    const uint32_t q = cut >> 2, r = cut & 3u;
    const uint32_t a0 = q == 0 ? f0 : (q == 1 ? f1 : f2), a1 = q == 0 ? f1 : (q == 1 ? f2 : 0u), a2 = q == 0 ? f2 : 0u;
    Token t;
    t.w[0] = bytes_from(a1, a0, r);
    t.w[1] = bytes_from(a2, a1, r);
    t.w[2] = bytes_from(0u, a2, r);
    if (!no_lead) t.w[0] = (t.w[0] & 0xFFFFFF00u) | (p_lines ? static_cast<uint32_t>(',') : (rev ? static_cast<uint32_t>('<') : static_cast<uint32_t>('>')));
    t.len = digits + (no_lead ? 0u : 1u);
    if (p_lines) {                                               // the byte behind the digits is zero: cut + len = 12
        const uint32_t sign = (rev ? static_cast<uint32_t>('-') : static_cast<uint32_t>('+')) << (8u * (t.len & 3u)), at = t.len >> 2;
        t.w[0] |= at == 0 ? sign : 0u;
        t.w[1] |= at == 1 ? sign : 0u;
        t.w[2] |= at == 2 ? sign : 0u;
        t.len++;
    }
    return t;
}

// The same for ids below 10^8 (every graph of up to a hundred million nodes; make_token takes the rest): the eight digits side by side in
// 64 bits, the leading zeros counted as zero BYTES (one count-trailing-zeros) and shifted out, the lead character shifted in.
GBWT_HIP_TOKEN_FN Token make_token_short(uint32_t node, bool p_lines, bool first_of_path) {
    const uint32_t v = node >> 1, rev = node & 1u;
    const uint32_t high = v / 10000u, low = v - high * 10000u;
    const uint64_t raw = static_cast<uint64_t>(four_digits(high)) | (static_cast<uint64_t>(four_digits(low)) << 32);
    const uint32_t zeros = static_cast<uint32_t>(__builtin_ctzll(raw | (1ull << 56))) >> 3;       // leading zero digits, 0 .. 7 (v = 0 keeps one digit)
    const uint32_t digits = 8u - zeros;
    const uint64_t text = (raw + 0x3030303030303030ull) >> (8u * zeros);                            // the digits from byte 0 on, zero bytes behind them
    uint32_t x0 = static_cast<uint32_t>(text), x1 = static_cast<uint32_t>(text >> 32), x2 = 0;
    Token t;
    t.len = digits;
    if (p_lines) {
        const uint32_t sign = rev ? static_cast<uint32_t>('-') : static_cast<uint32_t>('+'), at = 8u * (digits & 3u);
        x0 |= digits < 4u ? sign << at : 0u;
        x1 |= digits >= 4u && digits < 8u ? sign << at : 0u;
        x2 = digits == 8u ? sign : 0u;
        t.len++;
        if (first_of_path) { t.w[0] = x0; t.w[1] = x1; t.w[2] = x2; return t; }
    }
    const uint32_t lead = p_lines ? static_cast<uint32_t>(',') : static_cast<uint32_t>('>') - 2u * rev;      // '<' = '>' - 2
    t.w[0] = (x0 << 8) | lead;
    t.w[1] = bytes_from(x1, x0, 3u);
    t.w[2] = bytes_from(x2, x1, 3u);
    t.len++;
    return t;
}

// The token as the four aligned dwords it covers when its first byte lies `at` (0 .. 3) bytes into the first of them: what is OR-ed into
// a zeroed buffer -- the bytes around the token stay what the neighbours made them.
GBWT_HIP_TOKEN_FN void spread_token(const Token &t, uint32_t at, uint32_t out[4]) {
    const uint64_t a = ((static_cast<uint64_t>(t.w[1]) << 32) | t.w[0]) << (8u * at);
    const uint64_t b = ((static_cast<uint64_t>(t.w[2]) << 32) | t.w[1]) << (8u * at);
    const uint64_t c = static_cast<uint64_t>(t.w[2]) << (8u * at);
    out[0] = static_cast<uint32_t>(a);
    out[1] = static_cast<uint32_t>(a >> 32);
    out[2] = static_cast<uint32_t>(b >> 32);
    out[3] = static_cast<uint32_t>(c >> 32);
}

}  // namespace gbwt_hip
