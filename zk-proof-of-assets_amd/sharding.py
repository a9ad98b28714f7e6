"""Multi-GPU sharding of the MSMs: contiguous index ranges of (base, scalar) pairs,
one complete Pippenger per rank on its slice, then ONE exchange: an all-gather of each rank's
partial result (one affine point: 64 B for G1, 128 B for G2) followed by a local sum.

RCCL has no reduction operator for elliptic-curve addition, so the "all-reduce of partial sums" is
all-gather + local add; the payload is a few hundred bytes, so the collective is latency-bound and
bucket arrays are never exchanged. One process per GPU; `torch.distributed` (backend "nccl" = RCCL on
ROCm, "gloo" in the CPU tests) provides the collective.
"""


def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_bytes(payload, dist=None, device=None):
    """All-gather a fixed-size byte string over the default process group; returns the list by rank."""
    import torch
    if dist is None or not dist.is_initialized():
        return [bytes(payload)]
    t = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [bytes(o.cpu().numpy().tobytes()) for o in outs]


def combine_partials(group_sum, partials):
    """Sum of the per-rank partial points (wire format) with the library's host-side group sum."""
    return group_sum(b"".join(partials))


def sharded_msm(compute_partial, group_sum, n, dist=None, device=None):
    """compute_partial(lo, hi) -> wire-format point of this rank's slice; returns the full MSM
    (identical bytes on every rank)."""
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    lo, hi = shard_range(n, rank, world)
    part = compute_partial(lo, hi)
    return combine_partials(group_sum, all_gather_bytes(part, dist, device))


def sharded_prove(compute_partials, header_points, sum_partials, assemble, r, s, dist=None, device=None):
    """One Groth16 proof over `world` GPUs: compute_partials() -> this rank's 384-byte partial MSM results
    (A|B1|B2|C|H of its shard); all-gather; component-wise sum; host-side assembly with the SAME r, s on
    every rank (the caller fixes them, e.g. rank 0 draws and broadcasts). Returns proof_points[256]."""
    parts = all_gather_bytes(compute_partials(), dist, device)
    return assemble(header_points, sum_partials(parts), r, s)
