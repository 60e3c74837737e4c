"""Sharding of independent image pairs over ranks (one process per GPU) -- SURVEY.md §8(e).

A pair is the unit of work; pairs share nothing, so the data path has NO collective.  Pair k of a batch
goes to rank k mod world; every rank solves its pairs one after another on its own GPU and keeps the
float32 .flo payloads ((ny, nx, 2) each) in device memory.  One gather to rank 0 at the very end (RCCL
over xGMI when the backend is "nccl", gloo in the CPU tests) collects them for writing.  Works with any
torch.distributed backend; with world == 1 nothing is initialised or called.
"""


def pairs_of_rank(n_pairs, world, rank):
    """Pair indices owned by `rank`: k with k % world == rank, in increasing order."""
    return list(range(rank, n_pairs, world))


def pairs_per_rank(n_pairs, world):
    """Largest number of pairs any rank owns (ranks with fewer pad their send buffer)."""
    return (n_pairs + world - 1) // world


def gather_flows(local, n_pairs, world, rank, dist=None):
    """local: tensor (pairs_per_rank, ny, nx, 2) float32 holding this rank's flows in pair order
    (unused tail slots arbitrary).  Returns on rank 0 a list `flows[k]` (views, one per pair of the
    batch, in batch order) and None elsewhere.  Exactly one collective: dist.gather."""
    if world == 1:
        return [local[i] for i in range(n_pairs)]
    out = [local.new_empty(local.shape) for _ in range(world)] if rank == 0 else None
    dist.gather(local, out, dst=0)
    if rank != 0:
        return None
    flows = [None] * n_pairs
    for r in range(world):
        for slot, k in enumerate(pairs_of_rank(n_pairs, world, r)):
            flows[k] = out[r][slot]
    return flows
