"""CPU-side checks of the product: the C ABI loads and exports what the header declares,
host-only entry points agree with the oracle, the CLI keeps the reference's contract
 and the multi-rank plan is consistent (gloo, world_size 2).  No kernels run."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import lz77_amd as L
import oracle_lib as O
from lz77_amd import shard, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_GPU = L.lib().lz77x_device_count() == 0


def test_header_symbols_all_exported():
    hdr = open(os.path.join(ROOT, "include", "lz77_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lz77x_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    raw = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert getattr(raw, name) is not None


def test_library_is_the_hip_build():
    out = subprocess.run(["ldd", L.LIB_PATH], capture_output=True, text=True).stdout
    assert "libamdhip64" in out
    blob = open(L.LIB_PATH, "rb").read()
    assert b"gfx950" in blob                    # embedded code object target
    assert b"lz77o_" not in blob                # the oracle is not linked into the product


def test_product_library_has_one_path_per_stage():
    """The cross-check kernels (pair-scan matcher, round-1 walkers and tie-break, sequential boundary maps) live in
    liblz77_mi355x_variants.so only; both libraries export the same C ABI."""
    import re
    def kernels(path):
        return set(re.findall(rb"_Z\d+k_[a-z0-9_]+(?:ILb[01]ELi\d)?", open(path, "rb").read()))
    prod, var = kernels(L.LIB_PATH), kernels(L.VARIANTS_LIB_PATH)
    cross = [b"_Z10k_walk_big", b"_Z11k_prio_back", b"_Z12k_tokens_big", b"_Z12k_bidx_count", b"_Z7k_matchILb1ELi1", b"_Z7k_matchILb0ELi0"]
    for k in cross:
        assert k not in prod, k
        assert k in var, k
    for k in (b"_Z15k_tokens_sorted", b"_Z12k_prio_back2", b"_Z11k_walk_wave", b"_Z7k_matchILb1ELi3", b"_Z9k_dec_seg"):
        assert k in prod and k in var, k
    V = ctypes.CDLL(L.VARIANTS_LIB_PATH)
    for name in L.SYMBOLS:
        assert hasattr(V, name), name


def test_strerror_and_version():
    lib = L.lib()
    assert lib.lz77x_version().startswith(b"lz77-mi355x")
    assert lib.lz77x_strerror(0) == b"ok"
    assert b"no CPU fallback" in lib.lz77x_strerror(-4)


def test_encode_bound_formula():
    for sb, la, n in ((4095, 15, 0), (4095, 15, 1), (4095, 15, 1000), (65535, 255, 12345), (1000, 10, 7), (255, 7, 9)):
        T = O.bitof(sb) + O.bitof(la) + 8
        assert L.encode_bound(n, la, sb) == 4 + (n * T + 7) // 8
    assert L.encode_bound(5) == 4 + 15                               # defaults 15/4095 -> 24-bit tokens
    assert L.encode_bound(5, 1, 4095) == 0 and L.encode_bound(5, 15, 0) == 0


@pytest.mark.skipif(not NO_GPU, reason="only meaningful without a GPU")
def test_no_cpu_fallback():
    with pytest.raises(L.Lz77Error) as e:
        L.encode(b"abracadabra")
    assert e.value.code == -4
    with pytest.raises(L.Lz77Error) as e:
        L.decode(bytes.fromhex("ff0f0f00000061"))
    assert e.value.code == -4


@pytest.mark.parametrize("kind,seed,n,sb,la", [("text", 71, 60000, 4095, 15), ("random", 72, 30000, 1000, 10),
                                              ("lowent", 73, 40000, 255, 7), ("zeros", 0, 20000, 4095, 15)])
def test_host_priority_stage_matches_oracle(kind, seed, n, sb, la):
    """hoststage.c lz77x_prio_run (ring buffer, branchless) == oracle stage B == live BST"""
    data = synth.make(kind, n, seed)
    P, S, two = O.stage_a(data, sb, la, tree=True)
    xv = L.stage_priorities(P, S, sb)
    assert np.array_equal(xv, O.stage_b(P, S, sb))
    nx = max(n - sb, 0)
    assert np.array_equal(xv[:nx] != 0xFFFFFFFF, two[:nx].astype(bool))


def test_geometry_mirror():
    for sb, la in ((4095, 15), (65535, 255), (1, 2), (5, 3), (1000, 10), (4096, 16), (8191, 15), (8192, 16)):
        g = shard.geometry(sb, la)
        assert g["T"] == O.token_bits(sb, la)
        assert g["TILE"] % 8 == 0 and g["TILE"] >= 3 * g["SBu"] and g["TILE"] + sb <= g["RP"]
        assert g["RP"] & (g["RP"] - 1) == 0
    assert shard.geometry(4095, 15)["TILE"] == 12288 and shard.geometry(4095, 15)["fast"]
    assert not shard.geometry(65535, 255)["fast"]


# ---- CLI contract (main.c:59-180 of the reference; none of these reach the device) ----

def cli(*args):
    r = subprocess.run([L.CLI_PATH, *args], capture_output=True, text=True)
    return r.returncode, r.stdout, r.stderr


def test_cli_messages(tmp_path):
    f = str(tmp_path / "in")
    open(f, "wb").write(b"hello")
    o = str(tmp_path / "out")
    assert cli("-c", "-o", o) == (1, "", "Input file must be provided\n")
    assert cli("-c", "-i", f) == (1, "", "Output file must be provided\n")
    assert cli("-i", f, "-o", o) == (1, "", "Select ENCODE or DECODE mode\n")
    assert cli("-c", "-i", f, "-i", f, "-o", o) == (1, "", "Multiple input files not allowed.\n")
    assert cli("-c", "-i", f, "-o", o, "-o", o) == (1, "", "Multiple output files not allowed.\n")
    for bad in ("1", "256", "abc", "-3"):
        assert cli("-c", "-i", f, "-o", o, "-l", bad) == (1, "", "Bad lookahead size value.\n")
    for bad in ("65536", "-1", "0"):
        assert cli("-c", "-i", f, "-o", o, "-s", bad) == (1, "", "Bad search-buffer size value.\n")
    rc, out, err = cli("-c", "-i", str(tmp_path / "missing"), "-o", o)
    assert rc == 1 and out == "" and err.startswith("Opening input file: ")
    rc, out, err = cli("-c", "-i", f, "-o", str(tmp_path / "nodir" / "x"))
    assert rc == 1 and err.startswith("Opening output file: ")
    rc, out, err = cli("-h")
    assert rc == 1 and out.startswith("Usage: lz77 <options>\n") and out.count("\n") == 9
    assert err == "Input file must be provided\n"


@pytest.mark.skipif(not O.have_ref(), reason="compiled reference only in the build container")
def test_cli_messages_equal_reference(tmp_path):
    f = str(tmp_path / "in")
    open(f, "wb").write(b"hello")
    o = str(tmp_path / "out")
    for args in (["-c", "-o", o], ["-c", "-i", f], ["-i", f, "-o", o], ["-c", "-i", f, "-i", f, "-o", o],
                 ["-c", "-i", f, "-o", o, "-l", "1"], ["-c", "-i", f, "-o", o, "-s", "70000"], ["-h"],
                 ["-c", "-i", str(tmp_path / "missing"), "-o", o]):
        a = subprocess.run([O.REF_BIN, *args], capture_output=True, text=True)
        b = subprocess.run([L.CLI_PATH, *args], capture_output=True, text=True)
        assert (a.returncode, a.stdout, a.stderr) == (b.returncode, b.stdout, b.stderr), args


SHIM_BIN = os.path.join(O.ORACLE_DIR, "_ref", "lz77_shimmed")
SHIM_OBJ = os.path.join(os.path.dirname(L.LIB_PATH), "lz77_shim.o")


def test_shim_object_exports_the_reference_symbols():
    """lz77_shim.o defines encode()/decode() (lz77.h:14-15) and leaves bitIO_write/bitIO_read to the
    reference's own bitio.o -- nothing of tree.c / lz77.c is needed to link the reference's main.c"""
    syms = subprocess.check_output(["nm", SHIM_OBJ], text=True).split("\n")
    defined = {l.split()[-1] for l in syms if " T " in l}
    undefined = {l.split()[-1] for l in syms if l.strip().startswith("U ")}
    assert {"encode", "decode"} <= defined
    assert {"bitIO_write", "bitIO_read", "lz77x_encode_file", "lz77x_decode_file"} <= undefined     # both sides stream
    assert not any(u in undefined for u in ("insert", "find", "delete", "createTree", "updateOffset"))


@pytest.mark.skipif(not (O.have_ref() and os.path.exists(SHIM_BIN)), reason="reference main.c + shim only in the build container")
def test_reference_main_through_shim_cli_contract(tmp_path):
    """the reference's UNMODIFIED main.c linked against the shim: every diagnostic and exit code of the
    paths that end before the device is touched equals the reference binary's"""
    f = str(tmp_path / "in")
    open(f, "wb").write(b"hello")
    o = str(tmp_path / "out")
    for args in (["-c", "-o", o], ["-c", "-i", f], ["-i", f, "-o", o], ["-c", "-i", f, "-i", f, "-o", o],
                 ["-c", "-i", f, "-o", o, "-l", "1"], ["-c", "-i", f, "-o", o, "-s", "70000"], ["-h"],
                 ["-c", "-i", str(tmp_path / "missing"), "-o", o], ["-d", "-i", str(tmp_path / "missing"), "-o", o]):
        a = subprocess.run([O.REF_BIN, *args], capture_output=True, text=True)
        b = subprocess.run([SHIM_BIN, *args], capture_output=True, text=True)
        assert (a.returncode, a.stdout, a.stderr) == (b.returncode, b.stdout, b.stderr), args


# ---- multi-rank plan (gloo, 2 processes) ------------------------------------------------

_WORKER = r"""
# One stream cut over the ranks: every rank derives ITS shard's two hand-over maps from the oracle's view of
# its own range, the maps are all-gathered, and the LIBRARY's host functions (the ones lz77x_encode uses when
# lz77x_set_shards(D) > 1) plan the cut and chain the maps.  Checked against the sequential oracle.
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch, torch.distributed as dist
import oracle_lib as O
from lz77_amd import shard, synth
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n, sb, la = 90_000, 1000, 10
data = synth.text(n, 77)
plan = shard.plan(n, world, sb, la)
assert len(plan) == world and plan[0].first_token_pos == 0 and plan[-1].end_token_pos == n
for a, b in zip(plan, plan[1:]):
    assert a.end_token_pos == b.first_token_pos and b.local0 == b.first_token_pos - sb and b.lookback == sb
    assert a.local0 + a.local_bytes >= a.end_token_pos + la             # look-ahead of the last token
assert plan[-1].local0 + plan[-1].local_bytes == n
mine = plan[rank]

# ---- the oracle's sequential truth (every rank computes it; inputs are tiny) ----
P, S, two = O.stage_a(data, sb, la, tree=True)
P, S = P.astype(np.int64), S.astype(np.int64)
ml = O.maxlen(data, sb, la).astype(np.int64)
nx = n - sb
val = np.arange(n, dtype=np.int64)
gate = np.zeros(nx, dtype=bool)
state_at = {}                                   # step -> cells [step, step+sb) before that step
for x in range(nx):
    if x in {p.local0 for p in plan}:
        state_at[x] = val[x:x + sb].copy()
    if P[x] and S[x]:
        a = val[x]
        gate[x] = a < val[x + P[x]]
        if gate[x] and a < val[x + S[x]]:
            val[x + S[x]] = a
chain = []
p = 0
while p < n:
    chain.append(p)
    p += int(ml[p]) + 1
chain = np.asarray(chain)

# ---- this rank's maps, from its own range only (definition of k_prio_back / k_chain_map) ----
x0, x1 = mine.local0, mine.local0 + mine.steps          # its steps
dest = np.full(sb, 0xFFFF, dtype=np.uint16)
loc = (x1 + np.arange(sb)).astype(np.uint32)
dmap = {}
for x in range(x1 - 1, x0 - 1, -1):
    d = None
    if gate[x]:
        t = x + int(S[x])
        d = t - x1 if t >= x1 else dmap.get(t)
    dmap[x] = d
    if d is not None:
        loc[d] = min(int(loc[d]), x)
for i in range(sb):
    if x0 + i < x1 and dmap.get(x0 + i) is not None:
        dest[i] = dmap[x0 + i]
a0, a1 = mine.first_token_pos, mine.end_token_pos
exit_of = np.zeros(256, dtype=np.uint8)
tokens_of = np.zeros(256, dtype=np.uint32)
for e in range(la):
    q, c = a0 + e, 0
    while q < a1:
        q += int(ml[q]) + 1
        c += 1
    exit_of[e], tokens_of[e] = q - a1, c

# ---- exchange (gloo), then the library's host functions chain the shards ----
def gather(t):
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [o.numpy() for o in out]
dests = gather(torch.from_numpy(dest.astype(np.int32)))
locs = gather(torch.from_numpy(loc.astype(np.int64)))
exits = gather(torch.from_numpy(exit_of.astype(np.int32)))
counts = gather(torch.from_numpy(tokens_of.astype(np.int64)))
cells = np.arange(sb, dtype=np.uint32)
entry, ntok = 0, 0
for d in range(world):
    if d == rank:
        want = state_at[mine.local0]
        assert np.array_equal(cells.astype(np.int64), want), "cells this shard starts from"
        first = chain[np.searchsorted(chain, a0)]
        assert first == a0 + entry and np.searchsorted(chain, a0) == ntok, "parse position / token index at the cut"
    shard.compose_cells(dests[d].astype(np.uint16), locs[d].astype(np.uint32), cells)
    entry, ntok = shard.compose_chain(exits[d].astype(np.uint8), counts[d].astype(np.uint32), entry, ntok)
assert ntok == chain.size and entry == 0
# ---- decode side: the stream cut by token ranges; every rank decodes ITS range with the sb bytes before it as
#      symbols (256 + i = "byte i of what lies before me"), its map = the last sb states; the maps are all-gathered
#      and the library chains them: each rank's incoming bytes must be the true sb bytes before its output ----
z = O.encode_bst(data, sb, la)
_sb, _la, toff, tlen, tnext = O.tokens(z)
ntok_all = len(toff)
k0, k1 = shard.token_cut(ntok_all, world, rank), shard.token_cut(ntok_all, world, rank + 1)
assert k0 %% 8 == 0 and (rank + 1 < world or k1 == ntok_all)
hist = list(range(256, 256 + sb))                # the sb states before my first output byte
outv = []
for k in range(k0, k1):
    for i in range(int(tlen[k])):
        src = len(outv) - int(toff[k])
        outv.append(outv[src] if src >= 0 else hist[sb + src])
    outv.append(int(tnext[k]) & 0xFF)
lastsb = (hist + outv)[-sb:]
mymap = np.array([0x8000 | (v - 256) if v >= 256 else v for v in lastsb], dtype=np.uint16)
maps = gather(torch.from_numpy(mymap.astype(np.int32)))
lens = gather(torch.tensor([len(outv)], dtype=torch.int64))
incoming = np.zeros(sb, dtype=np.uint8)
start = 0
for d in range(world):
    if d == rank:
        truth = np.concatenate((np.zeros(sb, dtype=np.uint8), data))[start:start + sb]      # the sb bytes before output byte `start`
        assert np.array_equal(incoming, truth), "bytes this decode shard starts from"
        mine_out = np.array([incoming[v - 256] if v >= 256 else v for v in outv], dtype=np.uint8)
        assert np.array_equal(mine_out, data[start:start + len(outv)]), "this shard's bytes"
    nxt = shard.compose_tail(maps[d].astype(np.uint16), incoming)
    # the 32-bit form of the same map (windows above 8192: lz77x_shard_compose_tail32) chains to the same bytes
    m32 = np.array([0x10000 | (int(v) & 0x3FFF) if (int(v) & 0xC000) == 0x8000 else int(v) for v in maps[d]], dtype=np.uint32)
    assert np.array_equal(shard.compose_tail32(m32, incoming), nxt)
    incoming = nxt
    start += int(lens[d][0])
assert start == n
seeds = [shard.stream_seed(0x5EED0001, r) for r in range(world)]
assert len(set(seeds)) == world
t = shard.aggregate_time(1.0 + rank, dist)
assert t == float(world), t
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_rank_shard_exchange_gloo(tmp_path):
    """world_size 2 on CPU: the library's own shard plan and its host-side compositions (priority cells, parse
    chain; decode: token cuts and the sb-byte tail map), fed with per-rank maps exchanged over gloo, reproduce the
    sequential oracle at the cut"""
    script = tmp_path / "w.py"
    script.write_text(_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", LZ77X_NO_TORCH="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_shard_plan_properties():
    for n, sb, la, want in ((1_000_000_000, 4095, 15, 8), (100_000, 4095, 15, 3), (50_000, 4095, 15, 1), (10 << 20, 1000, 10, 8)):
        plan = shard.plan(n, 8, sb, la)
        assert len(plan) == want
        assert plan[0].first_token_pos == 0 and plan[-1].end_token_pos == n and plan[0].lookback == 0
        assert sum(p.end_token_pos - p.first_token_pos for p in plan) == n
        assert max(p.local_bytes for p in plan) <= n // len(plan) + sb + la + 65      # memory per device ~ n / D


def test_numa_binding_helper_is_best_effort():
    from lz77_amd.shard import _parse_cpulist, bind_to_device_numa
    assert _parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert _parse_cpulist("") == set()
    assert bind_to_device_numa("ffff:ff:1f.0") == 0          # no such device: nothing changes, no exception


def test_header_is_plain_c_and_links(tmp_path):
    """include/lz77_mi355x.h compiles as C11 with -Wall -Werror -pedantic and a caller links against the library"""
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include "lz77_mi355x.h"
#include <stdio.h>
int main(void)
{
    uint8_t *z = NULL; size_t zn = 0;
    lz77x_stats st;
    (void)st;
    if (lz77x_encode_bound(100, 4095, 15) != 4 + 300) return 2;
    int rc = lz77x_encode((const uint8_t *)"", 0, LZ77X_DEFAULT_SB, LZ77X_DEFAULT_LA, &z, &zn);
    printf("%s %s %d\n", lz77x_version(), lz77x_strerror(rc), lz77x_device_count());
    lz77x_free(z);
    lz77x_shutdown();
    return 0;
}
''')
    exe = tmp_path / "caller"
    libdir = os.path.dirname(L.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
                        "-L", libdir, "-llz77_mi355x", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    # without a GPU the encode call reports LZ77X_E_NODEV (-4) and the program still exits normally
    assert r.returncode == 0, r.stdout + r.stderr
    assert "lz77-mi355x" in r.stdout
