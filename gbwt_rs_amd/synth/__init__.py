"""ctypes wrapper of libgbwt_synth.so: synthetic GBWT/GBZ generator (host-only; see gbwt_synth.h)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgbwt_synth.so")

MOSAIC, IID = 0, 1

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make -C {HERE}` (or __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        p, u64 = C.c_void_p, C.c_uint64
        L.gbwt_synth_chain.restype = p
        L.gbwt_synth_chain.argtypes = [u64, u64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double, u64]
        L.gbwt_synth_chain_indel.restype = p
        L.gbwt_synth_chain_indel.argtypes = [u64, u64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double, u64, C.c_uint32, C.c_uint32]
        L.gbwt_synth_chain_chopped.restype = p
        L.gbwt_synth_chain_chopped.argtypes = [u64, u64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double, u64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.gbwt_synth_chain_labeled.restype = p
        L.gbwt_synth_chain_labeled.argtypes = [u64, u64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double, u64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.gbwt_synth_path_text_stats.restype = None
        L.gbwt_synth_path_text_stats.argtypes = [p, u64, u64 * 3]
        L.gbwt_synth_from_paths.restype = p
        L.gbwt_synth_from_paths.argtypes = [p, p, u64, C.c_int]
        L.gbwt_synth_merge.restype = p
        L.gbwt_synth_merge.argtypes = [p, u64, p, p, u64, p, u64, u64]
        L.gbwt_synth_from_file.restype = p
        L.gbwt_synth_from_file.argtypes = [C.c_char_p, C.c_char_p, u64]
        L.gbwt_synth_attach_gbz.restype = C.c_int
        L.gbwt_synth_attach_gbz.argtypes = [p, p, u64, u64]
        L.gbwt_synth_free.restype = None
        L.gbwt_synth_free.argtypes = [p]
        L.gbwt_synth_data.restype = p
        L.gbwt_synth_data.argtypes = [p, C.POINTER(u64)]
        L.gbwt_synth_starts.restype = p
        L.gbwt_synth_starts.argtypes = [p, C.POINTER(u64)]
        L.gbwt_synth_header.restype = None
        L.gbwt_synth_header.argtypes = [p, u64 * 8]
        L.gbwt_synth_save.restype = C.c_int
        L.gbwt_synth_save.argtypes = [p, C.c_char_p, C.c_int]
        L.gbwt_synth_set_tag.restype = None
        L.gbwt_synth_set_tag.argtypes = [p, C.c_char_p, C.c_char_p]
        L.gbwt_synth_path.restype = u64
        L.gbwt_synth_path.argtypes = [p, u64, p, u64]
        L.gbwt_synth_path_checksum.restype = u64
        L.gbwt_synth_path_checksum.argtypes = [p, u64]
        _lib = L
    return _lib


class Synth:
    """A generated (or loaded) index held on the host: record stream, starts, header, ground-truth paths."""

    def __init__(self, handle):
        if not handle:
            raise ValueError("generator returned NULL (bad parameters)")
        self._L = lib()
        self._h = handle
        hdr = (C.c_uint64 * 8)()
        self._L.gbwt_synth_header(self._h, hdr)
        (self.sequences, self.size, self.alphabet_offset, self.alphabet_size, bd, self.paths, self.sites,
         self.alleles) = [int(x) for x in hdr]
        self.bidirectional = bool(bd)

    @classmethod
    def chain(cls, sites, haplotypes, alleles=2, model=MOSAIC, founders=32, switch_rate=2e-3, zipf=1.2, seed=42, extra=0, indel_every=1, chop=1, labels=0):
        """`extra` > 0: alleles >= 1 are insertions of `extra` more nodes at every `indel_every`-th site (paths of
        different lengths); `chop` > 1: every node is a chain of that many nodes with consecutive ids; `labels` = 1: node labels of
        realistic lengths, 1 .. 1 024 bp, instead of one base each (gbwt_synth.h)."""
        h = lib().gbwt_synth_chain_labeled(sites, haplotypes, alleles, model, founders, switch_rate, zipf, seed, extra, indel_every, chop, labels)
        if not h:
            raise ValueError("gbwt_synth_chain: parameters out of range")
        return cls(h)

    @classmethod
    def from_paths(cls, paths, bidirectional=True):
        """paths: list of lists of GBWT-encoded nodes (2 * id + orientation)."""
        offsets = np.zeros(len(paths) + 1, dtype=np.uint64)
        np.cumsum([len(p) for p in paths], out=offsets[1:])
        flat = np.array([x for p in paths for x in p] or [0], dtype=np.uint64)
        return cls(lib().gbwt_synth_from_paths(offsets.ctypes.data, flat.ctypes.data, len(paths), int(bidirectional)))

    @classmethod
    def merge(cls, parts, path_names, sample_names, contig_names, haplotypes):
        """One index out of several chains (gbwt_synth.h): `parts` = Synth.chain objects (the graph components), `path_names` = one
        (sample, contig, phase, fragment) per path of the merged index (the paths of part 0, then of part 1, ...)."""
        handles = (C.c_void_p * len(parts))(*[q._h for q in parts])
        names = np.ascontiguousarray(path_names, dtype=np.uint32).reshape(-1, 4)
        assert len(names) == sum(q.paths for q in parts)
        samples = (C.c_char_p * len(sample_names))(*[x.encode() for x in sample_names])
        contigs = (C.c_char_p * len(contig_names))(*[x.encode() for x in contig_names])
        merged = cls(lib().gbwt_synth_merge(handles, len(parts), names.ctypes.data, samples, len(sample_names), contigs, len(contig_names), haplotypes))
        merged.path_names = names                              # (sample, contig, phase, fragment) per path
        merged.generic_sample = list(sample_names).index("_gbwt_ref") if "_gbwt_ref" in sample_names else len(sample_names)
        return merged

    @classmethod
    def genome(cls, contigs=24, fragments=3, haplotypes=12, sites=60, seed=1, extra=1, generic_per_contig=1, labels=0, min_walkers=0.5, threads=1,
               wrap_contig=None):
        """Config C4's shape at any scale: `contigs` contigs, each cut into `fragments` graph components that a random
        subset of the `haplotypes` haplotypes (at least `min_walkers` of them; sample s<h/2>, phase h%2+1) walks -- one ragged walk per
        (haplotype, component), with the fragment field = where the walk starts on its haplotype (the running sum of the earlier fragments
        plus gaps) -- and `generic_per_contig` reference paths (sample _gbwt_ref) in the first component of every contig.  `labels` = 1:
        node labels of realistic lengths (chain()).  `wrap_contig` = c: the walks of contig c start just below 2^32, so that the end
        coordinates of its W-lines (fragment + summed label lengths) pass 2^32 while the fragment fields still fit their 32 bits.
        `threads`: the components are generated side by side (the result does not depend on it)."""
        import random
        from concurrent.futures import ThreadPoolExecutor
        rng = random.Random(seed)
        jobs, names = [], []
        n_samples = (haplotypes + 1) // 2
        for c in range(contigs):
            at = [rng.randrange(0, 1000) for _ in range(haplotypes)]         # where every haplotype is on this contig
            for f in range(fragments):
                generic = generic_per_contig if f == 0 else 0
                walkers = sorted(rng.sample(range(haplotypes), rng.randint(max(1, int(haplotypes * min_walkers)), haplotypes)))
                n_sites = max(2, int(sites * rng.uniform(0.5, 2.0)))
                jobs.append(dict(sites=n_sites, haplotypes=generic + len(walkers), alleles=2, model=MOSAIC, founders=max(2, min(8, len(walkers))),
                                 switch_rate=0.02, seed=rng.randrange(1 << 30), extra=extra if (c + f) % 2 else 0, indel_every=3, labels=labels))
                names += [(n_samples, c, 0, 0)] * generic
                for h in walkers:
                    start = at[h]
                    if wrap_contig == c:                                     # the last component's walks start within a thousand bases of 2^32
                        start = (1 << 32) - 1 - (fragments - 1 - f) * 200000000 - at[h] % 1000
                    names.append((h // 2, c, h % 2 + 1, start))
                    at[h] += 3 * n_sites + rng.randrange(0, 500)
        if threads > 1:
            with ThreadPoolExecutor(threads) as pool:
                parts = list(pool.map(lambda kw: cls.chain(**kw), jobs))
        else:
            parts = [cls.chain(**kw) for kw in jobs]
        samples = [f"s{k}" for k in range(n_samples)] + ["_gbwt_ref"]
        merged = cls.merge(parts, names, samples, [f"chr{c + 1}" for c in range(contigs)], haplotypes)
        merged.sample_names = samples
        return merged

    @classmethod
    def from_file(cls, path):
        err = C.create_string_buffer(256)
        h = lib().gbwt_synth_from_file(os.fsencode(path), err, 256)
        if not h:
            raise ValueError(err.value.decode())
        return cls(h)

    def generic_paths(self):
        """Ids of the paths gbunzip writes as P-lines (sample _gbwt_ref); all others are walks (src/bin/gbunzip.rs:343-417)."""
        names = getattr(self, "path_names", None)
        if names is None:
            return [0] if self.paths else []                   # the chain generators: path 0
        return [p for p in range(len(names)) if names[p][0] == self.generic_sample]

    def attach_gbz(self, segment_starts=(), seed=1):
        """Adds path metadata + node labels (+ a node-to-segment translation) so that the index can be saved as a GBZ."""
        starts = np.array(list(segment_starts) or [0], dtype=np.uint64)
        rc = self._L.gbwt_synth_attach_gbz(self._h, starts.ctypes.data, len(segment_starts), seed)
        if rc != 0:
            raise ValueError(f"gbwt_synth_attach_gbz failed ({rc})")
        return self

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.gbwt_synth_free(self._h)
            self._h = None

    def data(self):
        n = C.c_uint64(0)
        p = self._L.gbwt_synth_data(self._h, C.byref(n))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint8)

    def starts(self):
        n = C.c_uint64(0)
        p = self._L.gbwt_synth_starts(self._h, C.byref(n))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint64)

    def save(self, path, as_gbz=False):
        if self._L.gbwt_synth_save(self._h, os.fsencode(path), int(as_gbz)) != 0:
            raise IOError(f"cannot write {path}")

    def set_tag(self, key, value):
        """A tag of the GBWT index as save() writes it (e.g. reference_samples: gbunzip's RS:Z: header field, src/bin/gbunzip.rs:193-203)."""
        self._L.gbwt_synth_set_tag(self._h, key.encode(), value.encode())

    def path(self, path_id):
        n = self._L.gbwt_synth_path(self._h, path_id, None, 0)
        out = np.zeros(max(1, n), dtype=np.uint32)
        self._L.gbwt_synth_path(self._h, path_id, out.ctypes.data, n)
        return out[:n]

    def path_text_stats(self, path_id):
        """(nodes, decimal digits of the node ids, summed label lengths) of a path: what its GFA line is made of (gbwt_synth.h)."""
        out = (C.c_uint64 * 3)()
        self._L.gbwt_synth_path_text_stats(self._h, path_id, out)
        return int(out[0]), int(out[1]), int(out[2])

    def path_checksum(self, path_id):
        return int(self._L.gbwt_synth_path_checksum(self._h, path_id))
