#!/usr/bin/env python3
"""Config C4: GFA path lines of a whole GBZ, sharded over the GPUs of a node, gathered over RCCL.

Every rank opens the same .gbz (the index is replicated), takes the paths `rank, rank + world, ...`
(interleaved: path lengths differ), walks them and formats their W-lines on its GPU in batches
(gbwt_hip_path_lines_device), and the finished text travels to rank 0 (gbwt_rs_amd/dist.py: one group of
point-to-point sends per batch), which consumes the lines in path order: it hashes -- or writes -- the concatenation,
the P/W part of what `gbunzip -t 1` prints.  Launch:  python -m torch.distributed.run --nproc-per-node N tools/gfa_sharded.py ...
(N = 1 works without a launcher; --backend gloo --share-gpu rehearses N > 1 on a one-GPU box).

Stand-in for an HPRC graph (no real file offline): Synth.genome -- --contigs contigs x --fragments graph components each, walked by
random subsets of --haplotypes haplotypes (sample, phase, contig, non-zero fragment offsets; one generic path per contig), --sites
sites per component on average.  24 x 20 x 90 with 62 500 sites = 43 k walks, 90 M node ids: SURVEY 8d's C4; the defaults fit one
GPU in seconds.  --contigs 0: one bubble chain (one contig, fragment 0) as in round 2.
The paths of the generic sample go out as P-lines from rank 0, the walks are sharded (write_paths / write_walks,
src/bin/gbunzip.rs:343-417)."""
import argparse
import hashlib
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=6000)
ap.add_argument("--haplotypes", type=int, default=90)
ap.add_argument("--contigs", type=int, default=24)
ap.add_argument("--fragments", type=int, default=20)
ap.add_argument("--oracle", action="store_true", help="rank 0 also compares the text with the oracle's gbunzip restatement (small sizes)")
ap.add_argument("--founders", type=int, default=16)
ap.add_argument("--batch", type=int, default=16, help="paths per rank and gather round")
ap.add_argument("--backend", default="nccl")
ap.add_argument("--share-gpu", action="store_true", help="all ranks on cuda:0 (rehearsal on a one-GPU box; with --backend gloo)")
ap.add_argument("--check", action="store_true", help="rank 0 also formats all paths alone and compares the hashes")
ap.add_argument("--out", default="", help="write the lines to this file (rank 0)")
ap.add_argument("--torch-gather", action="store_true", help="gather through torch.distributed (dist.gather_parts) instead of the C ABI's gbwt_hip_comm (default with --backend nccl)")
args = ap.parse_args()

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
local = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))

import torch
import torch.distributed as dist

import gbwt_rs_amd as G
from gbwt_rs_amd import dist as D
from gbwt_rs_amd import synth as S

torch.cuda.set_device(local)
device = torch.device("cuda", local)
if world > 1:
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(args.backend)
comm = device if args.backend == "nccl" else torch.device("cpu")
capi = D.Comm(rank, world, local) if (args.backend == "nccl" and world > 1 and not args.torch_gather) else None   # gbwt_hip_comm_*: RCCL behind the C ABI

# the same file on every rank (rank 0 writes it, the others wait for it)
path = os.path.join(tempfile.gettempdir(), f"gfa_sharded_{args.contigs}_{args.fragments}_{args.sites}_{args.haplotypes}.gbz")
if rank == 0:
    t0 = time.perf_counter()
    if args.contigs > 0:
        s = S.Synth.genome(contigs=args.contigs, fragments=args.fragments, haplotypes=args.haplotypes, sites=args.sites, seed=42)
    else:
        s = S.Synth.chain(args.sites, args.haplotypes, alleles=2, model=S.MOSAIC, founders=args.founders, seed=42)
    s.save(path + ".tmp", as_gbz=True)
    np.save(path + ".generic.npy", np.array(s.generic_paths(), dtype=np.uint64))
    os.replace(path + ".tmp", path)
    print(f"generated {path}: {s.paths} paths, {(s.size - s.sequences) // 2} forward LF-steps in {time.perf_counter() - t0:.1f} s", flush=True)
    del s
if world > 1:
    dist.barrier()
generic = np.load(path + ".generic.npy")
t0 = time.perf_counter()
gbz = G.GBZ.load(path, device=local)
open_s = time.perf_counter() - t0
n_paths = gbz.stats.paths
ids = np.setdiff1d(np.arange(n_paths, dtype=np.uint64), generic)    # the walks, ascending
mine = D.shard_ids(ids, rank, world, interleaved=True)
rounds = (len(D.shard_ids(ids, 0, world, interleaved=True)) + args.batch - 1) // args.batch   # rank 0 holds the most

if len(mine):    # untimed: the first call sizes the workspace (hundreds of megabytes of text and node buffers)
    gbz.path_lines_device(mine[:args.batch], 1)
    gbz.path_lines_device(mine[:1], 1)              # ... and leaves another request in the cache
sha, total, out = hashlib.sha256(), 0, open(args.out, "wb") if (args.out and rank == 0) else None
whole = [] if args.oracle else None
walk_ms = gather_ms = 0.0
torch.cuda.synchronize()
t_all = time.perf_counter()
if rank == 0 and len(generic):   # the P-lines: a handful of paths, written by rank 0 before the walks (write_paths)
    chunk = gbz.path_lines(generic, 0)
    sha.update(chunk)
    total += len(chunk)
    if out:
        out.write(chunk)
    if whole is not None:
        whole.append(chunk)
for q in range(rounds):
    batch = mine[q * args.batch:(q + 1) * args.batch]
    t0 = time.perf_counter()
    lines = gbz.path_lines_device(batch, 1)
    offsets, text = D.lines_tensors(lines, device)
    walk_ms += (time.perf_counter() - t0) * 1e3
    lengths = offsets[1:] - offsets[:-1]
    if capi is not None:
        t0 = time.perf_counter()
        got = capi.gather_lines(gbz, root=0, interleaved=True)       # the round's lines of all ranks, in path order on rank 0
        gather_ms += (time.perf_counter() - t0) * 1e3
        if rank == 0:
            _, gathered = D.lines_tensors(got, device)
            chunk = gathered.cpu().numpy().tobytes()
            sha.update(chunk)
            total += len(chunk)
            if out:
                out.write(chunk)
            if whole is not None:
                whole.append(chunk)
        continue
    if world > 1:
        t0 = time.perf_counter()
        if comm.type == "cpu":
            lengths, text = lengths.cpu(), text.cpu()
        len_parts, text_parts = D.gather_parts(lengths, text, dst=0)
        if comm.type == "cuda":
            torch.cuda.synchronize()
        gather_ms += (time.perf_counter() - t0) * 1e3
    else:
        len_parts, text_parts = [lengths], [text]
    if rank == 0:
        # the lines are consumed in path order where they arrived: no interleaved copy of the text is ever built
        host = [t.cpu().numpy() for t in text_parts]
        for r, a, b in D.rows_in_path_order(len_parts):
            chunk = host[r][a:b].tobytes()
            sha.update(chunk)
            total += len(chunk)
            if out:
                out.write(chunk)
            if whole is not None:
                whole.append(chunk)
elapsed = time.perf_counter() - t_all
if out:
    out.close()
if rank == 0:
    nodes = (gbz.len() - gbz.sequences()) // 2
    print(f"{world} rank(s), {n_paths} paths ({len(generic)} generic), {nodes} nodes: {total} bytes of P- and W-lines in {elapsed * 1e3:.1f} ms "
          f"({total / elapsed / 1e9:.2f} GB/s of text at rank 0, {nodes / elapsed / 1e9:.2f} G LF-steps/s; rank 0: walk + format {walk_ms:.1f} ms, "
          f"gather {gather_ms:.1f} ms, the rest is the copy to the host and the hash; open {open_s:.2f} s)  sha256 {sha.hexdigest()[:16]}", flush=True)
    if args.check:
        alone = hashlib.sha256()
        if len(generic):
            alone.update(gbz.path_lines(generic, 0))
        # path order of the interleaved rounds: round r holds walks r*batch*world .. in rank-interleaved order = ascending ids
        for lo in range(0, len(ids), args.batch * world):
            alone.update(gbz.path_lines(ids[lo:lo + args.batch * world], 1))
        same = alone.hexdigest() == sha.hexdigest()
        print(f"single-rank formatting of all paths: sha256 {alone.hexdigest()[:16]}  {'identical' if same else 'DIFFERENT'}", flush=True)
        if not same:
            raise SystemExit(1)
    if whole is not None:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        text = O.OracleGBZ(path).gfa()
        tail = text[text.index(b"\nP\t") + 1:] if b"\nP\t" in text else text[text.index(b"\nW\t") + 1:]
        same = b"".join(whole) == tail
        print(f"oracle (gbunzip restatement, default path mode): {'identical' if same else 'DIFFERENT'} ({len(tail)} bytes of P- and W-lines)", flush=True)
        if not same:
            raise SystemExit(1)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
if rank == 0:
    for leftover in (path, path + ".generic.npy"):
        try:
            os.remove(leftover)
        except OSError:
            pass
