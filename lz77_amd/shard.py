"""Host side of multi-GPU runs -- bindings of the library's own plan, no GPU needed.

Two ways the path shards, neither needs a data-path collective:

* independent streams (what bench.py measures at N>1, `--mode files`): rank r encodes/decodes its own
  stream; `stream_seed` gives each rank a distinct deterministic input.
* ONE stream cut by position (`plan`, bench.py `--mode shard`): shard d emits the tokens whose position
  lies in [first_token_pos, end_token_pos) and holds only its own bytes (sb of look-back, the look-ahead).
  The two sequential loops of lz77.c cross the cuts as small maps that the host chains:
  `compose_chain` (parse position / token count) and `compose_cells` (priorities of the sb live cells).
  These call the exact functions the library uses inside lz77x_encode when lz77x_set_shards(D) > 1.
"""
from __future__ import annotations

import ctypes
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
    tile = rp - sbu
    if rp >= 262144:                     # large windows: tiles of whole 64 K blocks (shared hierarchical sort)
        tile = tile // 65536 * 65536
    return {"sb": sb, "la": la, "ob": bitof(sb), "lb": bitof(la), "T": bitof(sb) + bitof(la) + 8,
            "SBu": sbu, "RP": rp, "TILE": tile, "fast": rp <= 16384}


def stream_seed(base_seed: int, rank: int) -> int:
    return (base_seed + rank) & 0xFFFFFFFFFFFFFFFF


class _CShard(ctypes.Structure):
    _fields_ = [("first_token_pos", ctypes.c_uint64), ("end_token_pos", ctypes.c_uint64), ("local0", ctypes.c_uint64),
                ("local_bytes", ctypes.c_uint64), ("steps", ctypes.c_uint64), ("lookback", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32)]


@dataclass(frozen=True)
class Shard:
    rank: int
    first_token_pos: int    # the shard emits the tokens whose position lies in [first, end)
    end_token_pos: int
    local0: int             # global position of the first byte the device holds
    local_bytes: int        # look-back + shard + look-ahead
    steps: int              # evictions it simulates: global [local0, end_token_pos - sb)
    lookback: int           # sb, 0 for the first shard


def plan(n: int, shards: int, sb: int, la: int) -> list:
    """lz77x_shard_plan: how the library cuts an n-byte stream for `shards` devices (it may use fewer)."""
    from . import lib
    arr = (_CShard * max(shards, 1))()
    d = lib().lz77x_shard_plan(n, sb, la, shards, ctypes.addressof(arr))
    if d < 1:
        raise ValueError("lz77x_shard_plan(%d, %d, %d, %d) -> %d" % (n, sb, la, shards, d))
    return [Shard(r, int(a.first_token_pos), int(a.end_token_pos), int(a.local0), int(a.local_bytes), int(a.steps), int(a.lookback))
            for r, a in enumerate(arr[:d])]


def compose_cells(dest, loc, cells):
    """lz77x_shard_compose_cells: cells (uint32[sb], in place) <- one shard's boundary map applied to them."""
    import numpy as np
    from . import lib
    dest = np.ascontiguousarray(dest, dtype=np.uint16)
    loc = np.ascontiguousarray(loc, dtype=np.uint32)
    assert cells.dtype == np.uint32 and cells.flags.c_contiguous and dest.size == loc.size == cells.size
    lib().lz77x_shard_compose_cells(dest.ctypes.data, loc.ctypes.data, int(cells.size), cells.ctypes.data)
    return cells


def compose_chain(exit_of, tokens_of, entry: int, tokens: int):
    """lz77x_shard_compose_chain: (entry offset, tokens so far) after one shard's parse-chain map."""
    import numpy as np
    from . import lib
    ex = np.ascontiguousarray(exit_of, dtype=np.uint8)
    tk = np.ascontiguousarray(tokens_of, dtype=np.uint32)
    e = ctypes.c_uint32(entry)
    t = ctypes.c_uint64(tokens)
    lib().lz77x_shard_compose_chain(ex.ctypes.data, tk.ctypes.data, ctypes.byref(e), ctypes.byref(t))
    return int(e.value), int(t.value)


def token_cut(ntok: int, shards: int, d: int) -> int:
    """decode side: first token of shard d (a multiple of eight tokens: a byte boundary of the stream)"""
    from . import lib
    return int(lib().lz77x_shard_token_cut(int(ntok), int(shards), int(d)))


def compose_tail(smap, incoming):
    """decode side: one shard's map (sb 16-bit states: a byte, or 0x8000 | index into the incoming bytes) applied
    to the sb bytes before the shard -> the last sb bytes of its output (what the next shard starts from)"""
    import numpy as np
    from . import lib
    m = np.ascontiguousarray(smap, dtype=np.uint16)
    i = np.ascontiguousarray(incoming, dtype=np.uint8)
    o = np.empty(m.size, dtype=np.uint8)
    lib().lz77x_shard_compose_tail(m.ctypes.data, int(m.size), i.ctypes.data, o.ctypes.data)
    return o


def compose_tail32(smap, incoming):
    """the same for windows above 8192 (32-bit states: a byte, or 0x10000 | index into the incoming bytes)"""
    import numpy as np
    from . import lib
    m = np.ascontiguousarray(smap, dtype=np.uint32)
    i = np.ascontiguousarray(incoming, dtype=np.uint8)
    o = np.empty(m.size, dtype=np.uint8)
    lib().lz77x_shard_compose_tail32(m.ctypes.data, int(m.size), i.ctypes.data, o.ctypes.data)
    return o


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
