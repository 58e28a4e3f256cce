"""The decoder takes a stream range by range (lz77.c:160-195 decodes any length through 3*SB+LA bytes): token ranges that
start on multiples of eight tokens, the last bytes of the output -- and, for a power-of-two -s, the image of the
reference's staging buffer -- carried from range to range.  LZ77X_DECODE_RANGE (tokens) and LZ77X_DECODE_RANGE_BYTES
(output bytes) force many ranges on small streams; every result is compared with the oracle's decoder, which restates the
reference's buffer including its stale bytes (pinned against the compiled reference in test_oracle_golden.py)."""
import os
import struct

import numpy as np
import pytest

import lz77_amd as L
import oracle_lib as O
from lz77_amd import synth

pytestmark = pytest.mark.gpu

RANGE_ENVS = [{"LZ77X_DECODE_RANGE": "8"}, {"LZ77X_DECODE_RANGE": "1000"}, {"LZ77X_DECODE_RANGE": "12000"},
              {"LZ77X_DECODE_RANGE_BYTES": "5000"}, {"LZ77X_DECODE_RANGE": "30000", "LZ77X_DECODE_RANGE_BYTES": "100000"}]
RANGE_IDS = ["r8", "r1000", "r12000", "b5000", "r30000b100000"]

# (kind, seed, n, sb, la): the headline geometry, T = 19 and T = 22, a window above 8192 (the tile path), the large
# window, power-of-two windows (distance-0 copies: the staging buffer's image is carried), tiny windows
CASES = [("text", 201, 400_000, 4095, 15), ("random", 202, 150_000, 4095, 15), ("code", 203, 300_000, 255, 7),
         ("mixed", 204, 300_000, 1000, 10), ("lowent", 205, 300_000, 8191, 16), ("text", 206, 500_000, 20000, 40),
         ("mixed", 207, 600_000, 65535, 255), ("text", 208, 300_000, 4096, 16), ("lowent", 209, 120_000, 16, 4),
         ("mixed", 210, 250_000, 1024, 15), ("records", 211, 400_000, 32768, 255), ("zeros", 0, 60_000, 8, 7),
         ("random", 212, 40_000, 2, 3), ("text", 213, 60_000, 1, 2), ("zeros", 0, 150_000, 4095, 15)]


@pytest.mark.parametrize("env", RANGE_ENVS, ids=RANGE_IDS)
@pytest.mark.parametrize("kind,seed,n,sb,la", CASES)
def test_ranges_decode_like_the_reference(kind, seed, n, sb, la, env, monkeypatch):
    if env.get("LZ77X_DECODE_RANGE") == "8" and n > 150_000:
        n = 150_000                                        # eight tokens a range: a few thousand ranges are plenty
    data = synth.make(kind, n, seed)
    z = O.encode_bst(data, sb, la)
    want = O.decode(z)
    if sb & (sb - 1):
        assert want == data.tobytes()
    one = L.decode(z)
    assert one == want
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    got = L.decode(z)
    st = L.last_stats()
    if st["ntok"] > 2 * int(env.get("LZ77X_DECODE_RANGE", 1 << 30)) or n > 2 * int(env.get("LZ77X_DECODE_RANGE_BYTES", 1 << 30)):
        assert st["match_launches"] >= 3, "the knob did not cut the stream into ranges"
    assert got == want


@pytest.mark.parametrize("env", RANGE_ENVS[1:4], ids=RANGE_IDS[1:4])
def test_ranges_through_the_file_and_device_entry_points(env, tmp_path, monkeypatch):
    """lz77x_decode_file (what -d calls, main.c:161) and lz77x_decode_device take the same path"""
    import torch
    data = synth.mixed(700_000, 214)
    z = O.encode_bst(data, 4095, 15)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    lz, out = str(tmp_path / "a.lz"), str(tmp_path / "a.out")
    open(lz, "wb").write(z)
    L.decode_path(lz, out)
    assert L.last_stats()["match_launches"] >= 3
    assert open(out, "rb").read() == data.tobytes()
    d_z = torch.from_numpy(np.frombuffer(z, dtype=np.uint8).copy()).cuda()
    assert L.decoded_size_device(d_z.data_ptr(), len(z)) == data.size
    d_back = torch.empty(data.size, dtype=torch.uint8, device="cuda")
    assert L.decode_device(d_z.data_ptr(), len(z), d_back.data_ptr(), data.size) == data.size
    assert d_back.cpu().numpy().tobytes() == data.tobytes()
    # a pipe: the size of the stream is not known up front
    import subprocess
    r = subprocess.run([L.CLI_PATH, "-d", "-i", "/dev/stdin", "-o", out], input=z, capture_output=True,
                       env=dict(os.environ, **env))
    assert r.returncode == 0 and r.stderr == b"" and open(out, "rb").read() == data.tobytes()


@pytest.mark.parametrize("env", [RANGE_ENVS[0], RANGE_ENVS[1], RANGE_ENVS[3]], ids=[RANGE_IDS[0], RANGE_IDS[1], RANGE_IDS[3]])
def test_golden_streams_in_ranges(env, golden, golden_dir, monkeypatch):
    """the streams the compiled reference wrote (tests/golden/*.lz) decoded range by range: the reference's inputs"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for s in golden["small"]:
        z = open(os.path.join(golden_dir, s["stem"] + ".lz"), "rb").read()
        want = open(os.path.join(golden_dir, s["stem"] + ".bin"), "rb").read()
        assert L.decode(z) == want, s["stem"]
        assert L.last_stats()["match_launches"] >= 2, s["stem"]
    for k in golden["kat"]:
        z = bytes.fromhex(k["lz_hex"])
        assert L.decode(z) == bytes.fromhex(k["decoded_hex"]), (k["name"], k["sb"], k["la"])


def _stream(tokens, sb, la):
    ob, lb = O.bitof(sb), O.bitof(la)
    T = ob + lb + 8
    acc, nbits = 0, 0
    out = bytearray(struct.pack("<HH", sb, la))
    for off, ln, ch in tokens:
        acc |= (off | (ln << ob) | (ch << (ob + lb))) << nbits
        nbits += T
        while nbits >= 8:
            out.append(acc & 0xFF)
            acc >>= 8
            nbits -= 8
    if nbits:
        out.append(acc & 0xFF)
    return bytes(out)


@pytest.mark.parametrize("env", [{"LZ77X_DECODE_RANGE": "64"}, {"LZ77X_DECODE_RANGE_BYTES": "3000"}], ids=["r64", "b3000"])
def test_foreign_streams_in_ranges(env, monkeypatch):
    """legal streams no run of the reference's encoder emits, cut into ranges: deep copy chains across the cuts, copies
    that overlap themselves, distances up to the window from the first byte of a range, a lookahead beyond the CLI's
    limit (tokens longer than a tile)"""
    rng = np.random.default_rng(7)
    toks = [(0, 0, 65)] + [(1, 14, 66)] * 50 + [(15, 14, 67)] * 400
    for _ in range(3000):
        ln = int(rng.integers(0, 16))
        toks.append((int(rng.integers(1, 4096)) if ln else 0, ln, int(rng.integers(0, 256))))
    z1 = _stream(toks, 4095, 15)
    toks2 = [(0, 0, 1), (0, 0, 2)] + [(int(rng.integers(1, 200)), int(rng.integers(0, 40000)), 3) for _ in range(300)]
    z2 = _stream(toks2, 255, 40000)
    want1, want2 = O.decode(z1), O.decode(z2)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    assert L.decode(z1) == want1
    assert L.last_stats()["match_launches"] >= 3
    assert L.decode(z2) == want2


def test_distance_beyond_the_window_across_ranges(monkeypatch):
    """the offset field is wider than sb (sb = 3000: 12 bits): a foreign stream may copy from up to 4095 back; the carry
    holds that much"""
    rng = np.random.default_rng(9)
    toks = [(0, 0, int(rng.integers(0, 256))) for _ in range(5000)]
    for _ in range(4000):
        toks.append((int(rng.integers(3001, 4096)), int(rng.integers(1, 16)), int(rng.integers(0, 256))))
    z = _stream(toks, 3000, 15)
    # flat semantics: out[j] = out[j - off]
    out = bytearray()
    for off, ln, ch in toks:
        for _ in range(ln):
            out.append(out[len(out) - off] if 0 < off <= len(out) else 0)
        out.append(ch)
    assert L.decode(z) == bytes(out)
    monkeypatch.setenv("LZ77X_DECODE_RANGE", "512")
    assert L.decode(z) == bytes(out)
    assert L.last_stats()["match_launches"] >= 3


def test_late_distance_0_copy_is_refused(monkeypatch):
    """a distance-0 copy reads the reference's staging buffer (lz77.c:178-181); this decoder follows that buffer from the
    first range on only for power-of-two windows or when the first range already holds such a copy -- a foreign stream that
    brings its first one later is refused, not mis-decoded"""
    toks = [(0, 0, 65)] * 4000 + [(0, 5, 66)] + [(0, 0, 67)] * 100
    z = _stream(toks, 4095, 15)
    assert L.decode(z) == O.decode(z)                      # in one piece: followed
    monkeypatch.setenv("LZ77X_DECODE_RANGE", "1024")
    with pytest.raises(L.Lz77Error) as e:
        L.decode(z)
    assert e.value.code == -5
    toks = [(0, 3, 64)] + toks                             # the first range holds one: followed throughout
    z = _stream(toks, 4095, 15)
    assert L.decode(z) == O.decode(z)


@pytest.mark.parametrize("env", [{}, {"LZ77X_DECODE_RANGE": "4096"}, {"LZ77X_DECODE_RANGE_BYTES": "50000"}, {"LZ77X_DECODE_SEGMENT": "65536"}],
                         ids=["one", "r4096", "b50000", "seg64k"])
@pytest.mark.parametrize("kind,seed,n,sb,la", [("text", 301, 2_000_000, 4095, 15), ("random", 302, 700_000, 4095, 15), ("mixed", 303, 900_000, 1000, 10),
                                              ("lowent", 304, 800_000, 255, 7), ("zeros", 0, 120_000, 4095, 15), ("records", 305, 600_000, 8191, 255),
                                              ("text", 306, 100_000, 3, 2), ("random", 307, 5_000, 4095, 15), ("text", 308, 1, 4095, 15), ("text", 309, 200_000, 4096, 16)])
def test_walk_that_reads_the_stream_equals_the_walk_on_token_words(kind, seed, n, sb, la, env, monkeypatch):
    """lz77.c:260-283: token k is bits [32 + kT, ..) of the stream.  Round 5's segment walk extracts its tokens from the
    stream itself and scans their lengths per 4 KB step (k_dec_sums, k_dec_bounds_fused, k_dec_seg<true>); the walk of
    rounds 2-4 read token words and output offsets that a parse kernel and a grid-wide scan had written
    (LZ77X_DECODE_UNFUSED=1, variants build).  Both give the reference's bytes, in one range and in many, with small
    decode segments, for token widths that are not whole bytes (T = 19, 22) and steps of thousands of tokens (random)."""
    data = synth.make(kind, n, seed)
    z = O.encode_bst(data, sb, la)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    want = O.decode(z)                                    # (== data unless -s is a power of two: lz77.c:172-188 reads stale bytes then)
    assert want == data.tobytes() or (sb & (sb - 1)) == 0
    assert L.decode(z) == want
    monkeypatch.setenv("LZ77X_DECODE_UNFUSED", "1")
    assert L.decode(z) == want
