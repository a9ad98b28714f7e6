"""Multi-GPU sharding of the MSMs: contiguous index ranges of (base, scalar) pairs,
one complete Pippenger per rank on its slice, then ONE exchange: an all-gather of each rank's
partial result (one affine point: 64 B for G1, 128 B for G2) followed by a local sum.

RCCL has no reduction operator for elliptic-curve addition, so the "all-reduce of partial sums" is
all-gather + local add; the payload is a few hundred bytes, so the collective is latency-bound and
bucket arrays are never exchanged. One process per GPU; `torch.distributed` (backend "nccl" = RCCL on
ROCm, "gloo" in the CPU tests) provides the collective.

The H-scalar chain (buildABC, 3 x ifft -> coset shift -> fft, joinABC) shards too (NTT row):
every transform is a four-step NTT over world = G ranks with ONE all-to-all, i.e. two exchanges per
polynomial for the whole chain, each rank sending n*32/G^2 bytes to every peer in a single hop (xGMI is
point-to-point, so all 7 links carry their own pair concurrently; no ring). `split_h_chain` below runs the
three local stages (C ABI zkpoa_split_stage1/2/3) around ONE `dist.all_to_all_single` per exchange (the exchange
buffers are laid out [rank][polynomial][slots], so all three polynomials travel in one collective), enqueued on the
library's own stream: two collectives and no host synchronisation for the whole chain.
"""


def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def block_cyclic_blocks(n, rank, world, block_log):
    """The (global start, count) runs of n items that `rank` holds when blocks of 2^block_log consecutive items are
    dealt round-robin (block b to rank b mod world) -- include/zkpoa_prover.h ZKPOA_SHARD_BLOCK_CYCLIC. Contiguous
    ranges inherit the clustering of a real witness (runs of bits, runs of full-width limbs: equal ranges, unequal
    work); small blocks even that out and each is still one contiguous byte range of the zkey file."""
    B = 1 << block_log
    return [(s, min(B, n - s)) for s in range(rank * B, n, world * B)]


def all_gather_bytes(payload, dist=None, device=None):
    """All-gather a fixed-size byte string over the default process group; returns the list by rank.
    On RCCL the ranks' payloads land in ONE tensor and come back with one device-to-host copy (the payload is a
    few hundred bytes: the exchange is pure latency, so every extra synchronisation counts)."""
    import torch
    if dist is None or not dist.is_initialized():
        return [bytes(payload)]
    world = dist.get_world_size()
    t = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
    if device is not None:
        t = t.to(device)
    if t.is_cuda:
        out = torch.empty(world * t.numel(), dtype=torch.uint8, device=t.device)
        dist.all_gather_into_tensor(out, t)
        raw = out.cpu().numpy().tobytes()
        n = t.numel()
        return [raw[i * n:(i + 1) * n] for i in range(world)]
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return [bytes(o.numpy().tobytes()) for o in outs]


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


def split_chain_supported(world, domain):
    """The split H-scalar chain needs world in {2, 4, 8} and world^2 <= domain; otherwise the chain is replicated."""
    return world in (2, 4, 8) and world * world <= domain


def exchange(recv, send, dist=None):
    """One exchange of the split chain: ONE all-to-all with equal splits over the whole buffer. The buffers are laid
    out [rank][polynomial A, B, C][Q * 32 bytes] (include/zkpoa_prover.h), so chunk h of `send` is everything rank h
    needs from this rank and `recv` ends up ordered by source rank. Runs on torch's CURRENT stream: inside
    `library_stream(ctx)` that is the stream the stages were enqueued on, and RCCL's stream is ordered against it with
    events (no host synchronisation)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        recv.copy_(send)
    elif send.is_cuda and dist.get_backend() == "gloo":
        # rehearsal only (ranks sharing a GPU, CPU process group): stage the exchange through the host
        hs = send.cpu()
        hr = hs.new_empty(hs.shape)
        dist.all_to_all_single(hr.view(-1), hs.view(-1))
        recv.copy_(hr)
    else:
        dist.all_to_all_single(recv.view(-1), send.view(-1))


class library_stream:
    """Context manager: makes the library's lane-0 HIP stream torch's current stream, so that the split-chain stages
    (enqueued there by the C ABI) and the collectives between them are ordered on the device. For CPU buffers (the
    gloo tests over the big-int stage model) it does nothing."""

    def __init__(self, ctx, device=None):
        self._cm = None
        if ctx is not None and device is not None and getattr(device, "type", "cpu") == "cuda":
            import torch
            self._cm = torch.cuda.stream(torch.cuda.ExternalStream(ctx.stream(0), device=device))

    def __enter__(self):
        if self._cm is not None:
            self._cm.__enter__()
        return self

    def __exit__(self, *exc):
        if self._cm is not None:
            return self._cm.__exit__(*exc)
        return False


def split_h_chain(stage1, stage2, stage3, buf_a, buf_b, dist=None):
    """The three local stages around the two exchanges. stageN take the buffers as the C ABI does:
    stage1(out=buf_a); exchange a->b; stage2(in=buf_b, out=buf_a); exchange a->b; stage3(in=buf_b).
    After it the rank's H scalars live on its key handle (zkpoa_split_stage3). Per proof: 2 collectives, no host
    synchronisation (call it inside library_stream(ctx))."""
    stage1(buf_a)
    exchange(buf_b, buf_a, dist)
    stage2(buf_b, buf_a)
    exchange(buf_b, buf_a, dist)
    stage3(buf_b)


def exchange_buffers(domain, world, device):
    """The two exchange buffers of the split chain: [world ranks][3 polynomials][Q = domain / world^2 elements x 32 B]."""
    import torch
    buf_a = torch.empty((world, 3, domain // world // world * 32), dtype=torch.uint8, device=device)
    return buf_a, torch.empty_like(buf_a)


def sharded_prove_split(ctx, key, d_witness, header_points, sum_partials, assemble, r, s, dist, device, buffers=None):
    """One Groth16 proof over world GPUs with the H-scalar chain split as well (key: a split shard handle,
    load_zkey_shard_split / set_shard_split; d_witness: device pointer or None for the witness already on the
    handle; buffers: exchange_buffers(...) kept across proofs). Returns proof_points[256], identical on every rank."""
    buf_a, buf_b = buffers if buffers is not None else exchange_buffers(key.info()[2], dist.get_world_size(), device)
    with library_stream(ctx, buf_a.device):
        split_h_chain(lambda a: ctx.split_stage1(key, d_witness, a.data_ptr()),
                      lambda b, a: ctx.split_stage2(key, b.data_ptr(), a.data_ptr()),
                      lambda b: ctx.split_stage3(key, b.data_ptr()), buf_a, buf_b, dist)
    # the witness MSMs (other lanes) start now and overlap whatever of the chain is still in flight on lane 0; the H
    # MSM follows stage 3 in stream order. Host synchronisations per proof: the MSMs' own read-backs + this all-gather.
    return sharded_prove(lambda: ctx.prove_partials_device(key, None), header_points, sum_partials, assemble, r, s,
                         dist, device)
