"""Deterministic synthetic byte streams for tests and bench.

The real corpora named in BASELINE.json (enwik8/enwik9/Silesia) are not available
offline, so every measured input is generated here from a splitmix64 counter
stream.  Generators are pure numpy, chunk-independent (byte i of a stream depends
only on (kind, seed, i-th item)), and cheap enough for 100 MB inside bench.py.

    S1 "enwik8-like" : text(100_000_000, 0x5EED0001)
    S2 "urandom"     : random_bytes(1 << 30, 0x5EED0002)
    S3 "silesia-like": mixed(212_000_000, 0x5EED0003)
    S4 "enwik9-like" : text(1_000_000_000, 0x5EED0004)
"""
from __future__ import annotations

import numpy as np

SEED_S1 = 0x5EED0001
SEED_S2 = 0x5EED0002
SEED_S3 = 0x5EED0003
SEED_S4 = 0x5EED0004

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, start: int, count: int) -> np.ndarray:
    """Outputs start .. start+count-1 of the splitmix64 sequence seeded with `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        v = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _GAMMA
        v = (v ^ (v >> np.uint64(30))) * _M1
        v = (v ^ (v >> np.uint64(27))) * _M2
        v = v ^ (v >> np.uint64(31))
    return v


def random_bytes(n: int, seed: int = SEED_S2) -> np.ndarray:
    """Raw splitmix64 output, little-endian bytes (identical to lz77o_splitmix_fill)."""
    out = np.empty(n, dtype=np.uint8)
    step = 1 << 21                      # 8-byte words per chunk
    pos = 0
    w0 = 0
    while pos < n:
        words = min(step, (n - pos + 7) // 8)
        b = splitmix64(seed, w0, words).view(np.uint8)
        take = min(b.size, n - pos)
        out[pos:pos + take] = b[:take]
        pos += take
        w0 += words
    return out


# ---- text-like ---------------------------------------------------------------

_LETTERS = np.frombuffer(
    b"eeeeeeeeeeeetttttttttaaaaaaaaooooooooiiiiiiinnnnnnnssssssrrrrrrhhhhhhlllldddd"
    b"cccuuummmffppggwwyybbvkxjqz", dtype=np.uint8)
_SEPS = [b" "] * 40 + [b", "] * 4 + [b". "] * 3 + [b"\n"] * 1 + [b" [["] + [b"]] "] + \
        [b" &quot;"] + [b"; "] + [b" ("] + [b") "] + [b".\n\n"] + [b" == "] + [b" <ref>"] + \
        [b"</ref> "] + [b" '''"] + [b"''' "] + [b": "] + [b" - "] + [b"|"] + [b" 19"] + [b" 20"]


class _Vocab:
    def __init__(self, seed: int, nwords: int):
        r = splitmix64(seed ^ 0xA5A5A5A5, 0, nwords)
        lens = (2 + (r % np.uint64(5)) + ((r >> np.uint64(8)) % np.uint64(5))).astype(np.int64)  # 2..10
        cap = (r >> np.uint64(20)) % np.uint64(16) == 0                     # ~6 % capitalised
        self.start = np.zeros(nwords + 1, dtype=np.int64)
        np.cumsum(lens, out=self.start[1:])
        total = int(self.start[-1])
        pick = splitmix64(seed ^ 0x5A5A5A5A, 0, total) % np.uint64(_LETTERS.size)
        flat = _LETTERS[pick.astype(np.int64)].copy()
        first = self.start[:-1][cap]
        flat[first] = flat[first] - 32
        self.flat = flat
        self.len = lens
        seps = _SEPS
        self.sep_len = np.array([len(s) for s in seps], dtype=np.int64)
        self.sep_start = np.zeros(len(seps) + 1, dtype=np.int64)
        np.cumsum(self.sep_len, out=self.sep_start[1:])
        self.sep_flat = np.frombuffer(b"".join(seps), dtype=np.uint8)


_vocab_cache: dict = {}


def _gather(starts: np.ndarray, lens: np.ndarray, flat: np.ndarray, dst_off: np.ndarray, out: np.ndarray):
    total = int(lens.sum())
    if total == 0:
        return
    rep = np.repeat(np.arange(lens.size), lens)
    first = np.cumsum(lens) - lens
    k = np.arange(total) - np.repeat(first, lens)
    out[np.repeat(dst_off, lens) + k] = flat[np.repeat(starts, lens) + k]
    del rep


def text(n: int, seed: int = SEED_S1, nwords: int = 1 << 14, zipf_pow: float = 2.0,
         phrase_every: int = 4, phrase_len: int = 3, reach: int = 300) -> np.ndarray:
    """Zipf-distributed pseudo-words with wiki-ish separators and recurring phrases."""
    key = (seed, nwords)
    voc = _vocab_cache.get(key)
    if voc is None:
        voc = _vocab_cache[key] = _Vocab(seed, nwords)
    out = np.empty(n + 64, dtype=np.uint8)
    pos = 0
    item = 0
    chunk = 1 << 19
    while pos < n:
        r = splitmix64(seed, item, chunk)
        item += chunk
        u = ((r >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)
        wid = np.minimum((np.power(float(nwords), u ** zipf_pow)).astype(np.int64) - 1, nwords - 1)
        wid = np.maximum(wid, 0)
        # local repeats: every so often a run of `phrase_len` words (and their separators)
        # is copied from up to `reach` words back -- the in-window phrase recurrence real
        # text has, and what gives matches longer than LA
        rr = (r & np.uint64(0xFFFF)).astype(np.int64)
        lead = rr % phrase_every == 0
        idx = np.arange(chunk)
        last_lead = np.maximum.accumulate(np.where(lead, idx, -1))
        dist = idx - last_lead
        run = (last_lead >= 0) & (dist < phrase_len)
        back = 1 + ((r >> np.uint64(16)) % np.uint64(reach)).astype(np.int64)
        back = back[np.maximum(last_lead, 0)]
        src = idx - back
        run &= src >= 0
        sid = ((r >> np.uint64(40)) % np.uint64(len(_SEPS))).astype(np.int64)
        wid = np.where(run, wid[np.maximum(src, 0)], wid)
        sid = np.where(run, sid[np.maximum(src, 0)], sid)
        wl = voc.len[wid]
        sl = voc.sep_len[sid]
        tot = wl + sl
        offs = np.cumsum(tot) - tot
        total = int(tot.sum())
        buf = np.empty(total, dtype=np.uint8)
        _gather(voc.start[:-1][wid], wl, voc.flat, offs, buf)
        _gather(voc.sep_start[:-1][sid], sl, voc.sep_flat, offs + wl, buf)
        take = min(total, n - pos)
        out[pos:pos + take] = buf[:take]
        pos += take
    return out[:n].copy()


# ---- other kinds -------------------------------------------------------------

def low_entropy(n: int, seed: int = 7) -> np.ndarray:
    """Small-alphabet runs: geometric run lengths of a handful of byte values."""
    out = np.empty(n, dtype=np.uint8)
    pos = 0
    item = 0
    while pos < n:
        r = splitmix64(seed, item, 1 << 16)
        item += 1 << 16
        val = (97 + (r % np.uint64(4))).astype(np.uint8)
        ln = (1 + ((r >> np.uint64(8)) % np.uint64(7)) * ((r >> np.uint64(16)) % np.uint64(9))).astype(np.int64)
        run = np.repeat(val, ln)
        take = min(run.size, n - pos)
        out[pos:pos + take] = run[:take]
        pos += take
    return out


def records(n: int, seed: int = 11, reclen: int = 48) -> np.ndarray:
    """Record-structured binary: a template row, a little-endian counter, sparse mutations."""
    rows = (n + reclen - 1) // reclen
    tmpl = random_bytes(reclen, seed ^ 0x1234)
    a = np.tile(tmpl, rows).reshape(rows, reclen)
    cnt = np.arange(rows, dtype=np.uint32)
    a[:, 0:4] = cnt.view(np.uint8).reshape(rows, 4)
    r = splitmix64(seed, 0, rows)
    col = (r % np.uint64(reclen - 4)).astype(np.int64) + 4
    a[np.arange(rows), col] = (r >> np.uint64(32)).astype(np.uint8)
    return a.reshape(-1)[:n].copy()


def code_like(n: int, seed: int = 13) -> np.ndarray:
    """ELF-ish: 4-byte words drawn from a skewed dictionary, many zero bytes."""
    words = (n + 3) // 4
    dic = splitmix64(seed ^ 0x77, 0, 512).astype(np.uint32)
    dic[::3] &= np.uint32(0x0000FFFF)
    dic[::5] &= np.uint32(0xFF0000FF)
    out = np.empty(words, dtype=np.uint32)
    pos = 0
    item = 0
    while pos < words:
        c = min(1 << 20, words - pos)
        r = splitmix64(seed, item, c)
        item += c
        u = ((r >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)
        out[pos:pos + c] = dic[np.minimum((512.0 ** u).astype(np.int64) - 1, 511)]
        pos += c
    return out.view(np.uint8)[:n].copy()


def zeros(n: int) -> np.ndarray:
    return np.zeros(n, dtype=np.uint8)


def mixed(n: int, seed: int = SEED_S3, seg: int = 1 << 20) -> np.ndarray:
    """Silesia-like concatenation: text / records / low-entropy / random / code-like segments."""
    out = np.empty(n, dtype=np.uint8)
    kinds = (text, records, low_entropy, random_bytes, code_like, text)
    pos = 0
    k = 0
    while pos < n:
        m = min(seg, n - pos)
        out[pos:pos + m] = kinds[k % len(kinds)](m, (seed + 0x9E37 * k) & 0xFFFFFFFF)
        pos += m
        k += 1
    return out


KINDS = {
    "text": text,
    "random": random_bytes,
    "lowent": low_entropy,
    "records": records,
    "code": code_like,
    "mixed": mixed,
}


def make(kind: str, n: int, seed: int) -> np.ndarray:
    if kind == "zeros":
        return zeros(n)
    return KINDS[kind](n, seed)
