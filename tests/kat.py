"""Known-answer data the reference's own tests hold for the hot path (data only, no reference code).

Sources (file:line into the reference repository):
  * paper example records, unidirectional and bidirectional: src/bwt/tests.rs:10-87
  * fixture paths: src/gbwt/tests.rs:116-162, src/gbz/tests.rs:371-381, test-data/with-empty.txt
  * ByteCode / RLE byte vectors: src/support.rs:1042-1045, 1188-1190
  * gbunzip expected text: SURVEY.md Appendix C (derived from src/bin/gbunzip.rs:193-550)
"""

ENDMARKER = 0

# (edges, runs, invalid_node) -- src/bwt/tests.rs:10-32
PAPER_EDGES = [
    [(1, 0)],
    [(2, 0), (3, 0)],
    [(4, 0), (5, 0)],
    [(4, 1)],
    [(5, 1), (6, 0)],
    [(7, 0)],
    [(7, 2)],
    [(0, 0)],
]
PAPER_RUNS = [
    [(0, 3)],
    [(0, 2), (1, 1)],
    [(0, 1), (1, 1)],
    [(0, 1)],
    [(1, 1), (0, 1)],
    [(0, 2)],
    [(0, 1)],
    [(0, 3)],
]
PAPER_INVALID = 8

# src/bwt/tests.rs:35-87
BD_EDGES = [
    [(2, 0), (15, 0)],
    [(4, 0), (6, 0)],
    [(0, 0)],
    [(8, 0), (10, 0)],
    [(3, 0)],
    [(8, 1)],
    [(3, 2)],
    [(10, 1), (12, 0)],
    [(5, 0), (7, 0)],
    [(14, 0)],
    [(5, 1), (9, 0)],
    [(14, 2)],
    [(9, 1)],
    [(0, 0)],
    [(11, 0), (13, 0)],
]
BD_RUNS = [
    [(0, 3), (1, 3)],
    [(0, 2), (1, 1)],
    [(0, 3)],
    [(0, 1), (1, 1)],
    [(0, 2)],
    [(0, 1)],
    [(0, 1)],
    [(1, 1), (0, 1)],
    [(1, 1), (0, 1)],
    [(0, 2)],
    [(0, 1), (1, 1)],
    [(0, 1)],
    [(0, 1)],
    [(0, 3)],
    [(1, 1), (0, 2)],
]
BD_INVALID = 16


def fwd(n):
    return 2 * n


def rev(n):
    return 2 * n + 1


def flip(n):
    return n ^ 1


def reverse_path(path):
    return [flip(x) for x in reversed(path)]


# src/gbwt/tests.rs:116-162 (GBWT node ids)
def true_paths(with_empty):
    result = [
        [fwd(11), fwd(12), fwd(14), fwd(15), fwd(17)],
        [fwd(21), fwd(22), fwd(24), fwd(25)],
        [fwd(11), fwd(12), fwd(14), fwd(15), fwd(17)],
        [fwd(11), fwd(13), fwd(14), fwd(16), fwd(17)],
    ]
    if with_empty:
        result.append([])
    result.append([fwd(21), fwd(22), fwd(24), rev(23), rev(21)])
    result.append([fwd(21), fwd(22), fwd(24), fwd(25)])
    return result


# src/gbwt/tests.rs:242-250
def true_nodes():
    out = set()
    for n in [11, 12, 13, 14, 15, 16, 17, 21, 22, 23, 24, 25]:
        out.add(fwd(n))
        out.add(rev(n))
    return out


# src/gbwt/tests.rs:252-266
def count_occurrences(paths, subpath):
    result = 0
    r = reverse_path(subpath)
    n = len(subpath)
    for path in paths:
        for i in range(len(path)):
            if path[i:i + n] == subpath:
                result += 1
            if i + 1 >= n and path[i + 1 - n:i + 1] == r:
                result += 1
    return result


# src/gbz/tests.rs:371-381 (node ids, all forward)
TRANSLATION_PATHS = [
    [1, 2, 3, 5, 6, 9, 11],
    [1, 2, 3, 5, 6, 9, 11],
    [1, 2, 4, 5, 6, 10, 11],
]

# SURVEY.md Appendix B
EXAMPLE_STARTS = [0, 19, 26, 30, 34, 38, 42, 46, 53, 60, 64, 68, 72, 76, 80, 87, 88, 89, 90, 91, 92, 93, 101, 105,
                  109, 113, 117, 121, 129, 133, 137]
EXAMPLE_DATA_HEX = (
    "0416000d000700090000010203000100010a030218000200020101000002011c000101170001011c020001170200021e0002000201021900"
    "0200020101220001011d000101220200011d020001000002021f0002000201000000000000022c0002000201000100000301300002012b00"
    "0201310000012b0300022f000300010001012d00020100000101310101"
)
EXAMPLE_SEQUENCES = {
    0: [22, 24, 28, 30, 34], 1: [35, 31, 29, 25, 23], 2: [42, 44, 48, 50], 3: [51, 49, 45, 43],
    6: [22, 26, 28, 32, 34], 7: [35, 33, 29, 27, 23], 8: [42, 44, 48, 47, 43], 9: [42, 46, 49, 45, 43],
}
EXAMPLE_SEQUENCES[4] = EXAMPLE_SEQUENCES[0]
EXAMPLE_SEQUENCES[5] = EXAMPLE_SEQUENCES[1]
EXAMPLE_SEQUENCES[10] = EXAMPLE_SEQUENCES[2]
EXAMPLE_SEQUENCES[11] = EXAMPLE_SEQUENCES[3]

TRANSLATION_STARTS = [0, 11, 15, 19, 26, 30, 34, 38, 42, 46, 50, 57, 64, 68, 69, 70, 71, 72, 76, 80, 84, 88, 92]
TRANSLATION_DATA_HEX = (
    "020200150000010001000101040002010000020206000200020101030002010a000101050001010a020001050200010c0002020700020002"
    "0102120002000201010b00020000000001160001010d000101160200010d02000100000202130002000201"
)

# SURVEY.md Appendix C
EXAMPLE_GFA_SHA256 = "94cc77c1f0d425ebe2a7f257935a02e109914d366075158505e9d9c91891158d"
EXAMPLE_GFA_LEN = 454
TRANSLATION_GFA_SHA256 = "4c4f4449ed08039eb3b546d673a51d426511875c595241327d392c59f9293b21"
TRANSLATION_GFA_LEN = 309
EXAMPLE_PW_LINES = (
    b"P\tA\t11+,12+,14+,15+,17+\t*\n"
    b"P\tB\t21+,22+,24+,25+\t*\n"
    b"W\tsample\t1\tA\t0\t5\t>11>12>14>15>17\n"
    b"W\tsample\t2\tA\t0\t5\t>11>13>14>16>17\n"
    b"W\tsample\t1\tB\t0\t5\t>21>22>24<23<21\n"
    b"W\tsample\t2\tB\t0\t4\t>21>22>24>25\n"
)
