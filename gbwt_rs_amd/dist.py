"""Multi-GPU plumbing for path extraction: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The path set shards by sequence: every rank walks its own shard against its own replica (or its own contig) of the
index, and there is NO collective inside the walk.  The only exchange is the one the reference's writer mutex
stands for (src/bin/gbunzip.rs:421-434): putting the extracted rows back into path order on one rank.  That is

    1. all_gather of the per-rank row lengths                        (8 B per row)
    2. variable-size gather of the row data to the destination rank  (point-to-point sends: every peer has its own
       xGMI link to the root, so a direct gather beats a ring all-gather, which is bound by one link)

Works on CUDA/HIP tensors over RCCL and on CPU tensors over gloo (the CPU form is what the tests run).
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous shard [lo, hi) of n items for `rank`: sizes differ by at most one."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_ids(ids, rank, world, interleaved=False):
    """The sequence ids this rank walks.  Contiguous blocks keep the gathered output in path order with one copy
    per rank; interleaving (id k -> rank k mod world) balances ragged path lengths."""
    if interleaved:
        return ids[rank::world]
    lo, hi = shard_bounds(len(ids), rank, world)
    return ids[lo:hi]


def gather_rows(lengths, values, dst=0, group=None, interleaved=False):
    """Gathers this rank's CSR rows (lengths[k] values each, concatenated in `values`) on rank `dst`.

    Returns (offsets, values) of all rows in global path order on `dst`, (None, None) elsewhere.  `lengths` is an
    int64 tensor, `values` any 1-D tensor; both on the same device."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = lengths.device
    counts = torch.tensor([lengths.numel(), int(lengths.sum().item())], dtype=torch.int64, device=device)
    all_counts = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    rows = [int(c[0]) for c in all_counts]
    sizes = [int(c[1]) for c in all_counts]
    if rank == dst:
        len_parts = [lengths if r == rank else torch.empty(rows[r], dtype=torch.int64, device=device) for r in range(world)]
        val_parts = [values if r == rank else torch.empty(sizes[r], dtype=values.dtype, device=device) for r in range(world)]
        reqs = []
        for r in range(world):
            if r == rank:
                continue
            if rows[r]:
                reqs.append(dist.irecv(len_parts[r], src=r, group=group))
            if sizes[r]:
                reqs.append(dist.irecv(val_parts[r], src=r, group=group))
        for q in reqs:
            q.wait()
        if not interleaved:
            all_len = torch.cat(len_parts)
            all_val = torch.cat(val_parts)
        else:  # row k of rank r is global row k * world + r
            total_rows = sum(rows)
            all_len = torch.zeros(total_rows, dtype=torch.int64, device=device)
            for r in range(world):
                all_len[r::world] = len_parts[r]
            offsets = torch.zeros(total_rows + 1, dtype=torch.int64, device=device)
            torch.cumsum(all_len, 0, out=offsets[1:])
            all_val = torch.empty(int(offsets[-1].item()), dtype=values.dtype, device=device)
            for r in range(world):
                src_off = torch.zeros(rows[r] + 1, dtype=torch.int64, device=device)
                torch.cumsum(len_parts[r], 0, out=src_off[1:])
                for k in range(rows[r]):   # host loop: gather order only matters for the (small) tests and GFA assembly
                    g = k * world + r
                    all_val[offsets[g]:offsets[g + 1]] = val_parts[r][src_off[k]:src_off[k + 1]]
            return offsets, all_val
        offsets = torch.zeros(all_len.numel() + 1, dtype=torch.int64, device=device)
        torch.cumsum(all_len, 0, out=offsets[1:])
        return offsets, all_val
    if lengths.numel():
        dist.send(lengths, dst=dst, group=group)
    if values.numel():
        dist.send(values, dst=dst, group=group)
    return None, None
