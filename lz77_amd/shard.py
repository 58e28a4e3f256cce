"""Host-side sharding plan for multi-GPU runs -- pure Python, no GPU needed.

Two ways the path shards, neither needs a data-path collective:

* independent streams (what bench.py measures at N>1): rank r encodes/decodes its own
  stream; `stream_seed` gives each rank a distinct deterministic input.
* one stream cut by position (`plan_positions`): rank r owns the k_match regions whose tile
  starts fall in [begin, end); it reads the halo [begin-SBu, end+SB+LA) of the input
  read-only.  The per-position results (ps, maxlen) are concatenated on the host, which
  runs the sequential stage once over the whole stream.
"""
from __future__ import annotations

from dataclasses import dataclass


def bitof(n: int) -> int:
    """bitio.c:41-43 in integers."""
    b = 0
    while (1 << b) < n:
        b += 1
    return b


def geometry(sb: int, la: int) -> dict:
    """Mirror of lz77x_make_geom (csrc/hoststage.c)."""
    sbu = (sb + 7) & ~7
    rp = 4096
    while rp < 4 * sbu:
        rp <<= 1
    return {"sb": sb, "la": la, "ob": bitof(sb), "lb": bitof(la), "T": bitof(sb) + bitof(la) + 8,
            "SBu": sbu, "RP": rp, "TILE": rp - sbu, "fast": rp <= 16384}


def stream_seed(base_seed: int, rank: int) -> int:
    return (base_seed + rank) & 0xFFFFFFFFFFFFFFFF


@dataclass(frozen=True)
class Shard:
    rank: int
    region0: int        # first k_match region
    nregions: int
    begin: int          # first position produced
    end: int            # one past the last position produced
    halo_begin: int     # first input byte read
    halo_end: int       # one past the last input byte read


def plan_positions(n: int, world: int, sb: int, la: int) -> list:
    """Split the regions of an n-byte stream into `world` contiguous, nearly equal shards."""
    g = geometry(sb, la)
    tile = g["TILE"]
    nreg = (n + tile - 1) // tile
    out = []
    for r in range(world):
        r0 = nreg * r // world
        r1 = nreg * (r + 1) // world
        b, e = min(r0 * tile, n), min(r1 * tile, n)
        # a shard that does not start at 0 also runs the region before its first one (lz77x_internal.h)
        out.append(Shard(r, r0, r1 - r0, b, e, max(0, b - tile) if e > b else b,
                         min(n, e + sb + la) if e > b else b))
    return out


def aggregate_time(dt: float, dist=None) -> float:
    """MAX over ranks of a wall time (the bench contract); identity without a process group."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return dt
    import torch
    t = torch.tensor([dt], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_device_numa(pci_bus_id: str):
    """Pin this process (and the threads it creates later: the library's recurrence thread, the pages
    it first-touches for the pinned rings) to the CPUs local to the GPU with the given PCI address
    ("0000:c1:00.0"), the per-rank equivalent of `numactl --cpunodebind`.  The host recurrence streams
    5 B per position out of pinned memory the GPU writes, so a rank whose threads sit on the other
    socket pays the inter-socket hop on its critical path.  Best effort: returns the CPU count it
    bound to, or 0 when the topology is not exposed / the node has one socket / nothing would change."""
    import os
    try:
        with open("/sys/bus/pci/devices/%s/local_cpulist" % pci_bus_id.lower()) as f:
            local = _parse_cpulist(f.read())
        allowed = os.sched_getaffinity(0)
        want = local & allowed
        if not want or want == allowed:
            return 0
        os.sched_setaffinity(0, want)
        return len(want)
    except (OSError, ValueError, AttributeError):
        return 0
