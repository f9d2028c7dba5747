"""Oracle codecs vs the reference's byte-level known answers (src/support.rs, src/support/tests.rs)."""
import random

import oracle_lib as O


def test_bytecode_kat():
    # src/support.rs:1042-1045
    assert O.bytecode_encode([123, 456, 789]) == bytes([123, 72 + 128, 3, 21 + 128, 6])
    assert O.bytecode_decode(bytes([123, 200, 3, 149, 6])) == [123, 456, 789]


def test_bytecode_roundtrip_random():
    # src/support/tests.rs:362-384 (random widths up to 64 bits; varints up to 10 bytes, 345-350)
    rng = random.Random(1)
    values = [rng.getrandbits(rng.randint(1, 64)) for _ in range(2000)] + [0, 127, 128, 2**64 - 1]
    data = O.bytecode_encode(values)
    assert O.bytecode_decode(data) == values
    assert len(O.bytecode_encode([2**64 - 1])) == 10


def test_bytecode_truncated():
    # a varint whose continuation never ends yields no value (ByteCodeIter::next returns None)
    assert O.bytecode_decode(bytes([0x80, 0x80])) == []
    assert O.bytecode_decode(bytes([5, 0x81])) == [5]


def test_rle_kat():
    # src/support.rs:1188-1190
    runs = [(3, 12), (2, 721), (0, 34)]
    data = O.rle_encode(4, runs)
    assert data == bytes([3 + 4 * 11, 2 + 4 * 63, 17 + 128, 5, 0 + 4 * 33]) == bytes([47, 254, 145, 5, 132])
    assert O.rle_decode(4, data) == runs


def test_rle_thresholds():
    # src/support/tests.rs:439-469: len threshold-1 -> one byte, len threshold -> two bytes
    for sigma in [1, 4, 5, 128, 129, 254]:
        threshold = 256 // sigma
        if threshold > 1:
            assert len(O.rle_encode(sigma, [(sigma - 1, threshold - 1)])) == 1
        assert len(O.rle_encode(sigma, [(sigma - 1, threshold)])) == 2
        for ln in [1, threshold - 1, threshold, threshold + 1, 1000, 100000]:
            if ln < 1:
                continue
            runs = [(sigma - 1, ln), (0, ln)]
            assert O.rle_decode(sigma, O.rle_encode(sigma, runs)) == runs


def test_rle_roundtrip_random():
    # src/support/tests.rs:425-437: sigma in {4, 254, 255, 14901, 0}
    rng = random.Random(2)
    for sigma in [1, 2, 3, 4, 64, 65, 128, 129, 254, 255, 256, 14901, 0]:
        top = sigma if sigma else 1 << 40
        runs = []
        for _ in range(500):
            ln = rng.choice([1, 2, 3, rng.randint(1, 300), rng.randint(1, 1 << 20)])
            runs.append((rng.randrange(top), ln))
        data = O.rle_encode(sigma, runs)
        assert O.rle_decode(sigma, data) == runs


def test_rle_sigma_large_two_varints():
    # sigma >= 255: value varint, (len - 1) varint (src/support.rs:1239-1241)
    assert O.rle_encode(255, [(254, 1)]) == bytes([254, 1, 0])
    assert O.rle_encode(300, [(299, 130)]) == bytes([299 & 0x7F | 0x80, 299 >> 7, 129 & 0x7F | 0x80, 1])


def test_hand_encoded_record():
    # src/support/tests.rs:471-513 style: a GBWT record written by hand decodes as expected
    # sigma=2; edges (24,0),(26,0); runs (0,2)(1,1)  == record 1 of example.gbwt
    data = bytes.fromhex("02180002000201")
    vals = O.bytecode_decode(data[:5])
    assert vals == [2, 24, 0, 2, 0]
    assert O.rle_decode(2, data[5:]) == [(0, 2), (1, 1)]
