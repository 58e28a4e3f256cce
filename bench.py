#!/usr/bin/env python3
"""bench.py -- encode+decode throughput of the LZ77 hot path on N MI355X GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]; the real enwik8 is not available offline): the S1
"enwik8-like" synthetic text stream, 100,000,000 bytes, s=4095 l=15.  One step = encode
that stream and decode the result, inputs and outputs resident in HBM.  With N>1 every
rank owns an independent stream of the same size (seed + rank): chunks shard with no
data-path collective, so scaling is weak; `value` is the whole-job rate
N * bytes * K / max-over-ranks wall time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CLOCK_HZ = 2.4e9               # MI355X_MICROARCH.md: max clock
N_SIMD = 256 * 4               # 256 CUs x 4 SIMDs
VALU_CYCLES = 3.5              # cycles a wave64 VALU instruction occupies its SIMD: measured 2.8 - 4.5 (tests/ubench/issue_rates.hip)


def cpu_baseline(data, sb, la, budget_bytes):
    """Reference CPU encode()+decode() timed on this host, single thread, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    sample = data[:budget_bytes]
    n = int(sample.size)
    if O.have_ref():
        kind = "reference"
        tmp = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
        fin, flz, fout = (os.path.join(tmp, "lz77bench_%d.%s" % (os.getpid(), e)) for e in ("in", "lz", "out"))
        sample.tofile(fin)
        try:
            t0 = time.perf_counter()
            subprocess.check_call([O.REF_BIN, "-c", "-i", fin, "-o", flz, "-s", str(sb), "-l", str(la)])
            t1 = time.perf_counter()
            subprocess.check_call([O.REF_BIN, "-d", "-i", flz, "-o", fout])
            t2 = time.perf_counter()
        finally:
            for p in (fin, flz, fout):
                if os.path.exists(p):
                    os.unlink(p)
    else:
        kind = "port"
        O.lib()
        t0 = time.perf_counter()
        z = O.encode_bst(sample, sb, la)
        t1 = time.perf_counter()
        O.decode(z)
        t2 = time.perf_counter()
    return {"value": round(n / (t2 - t0) / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": kind,
            "sample": ("first %d bytes of the same S1 stream, encode then decode, single thread; WHOLE-PROCESS wall time of the "
                       "reference CLI (process start + file I/O on tmpfs included) -- like for like with `file_to_file`, not with "
                       "`value` (HBM-resident kernels on the full 100 MB)" if kind == "reference" else
                       "first %d bytes of the same S1 stream, encode then decode, single thread, in-memory port") % n,
            "encode_MBps": round(n / (t1 - t0) / 1e6, 3), "decode_MBps": round(n / (t2 - t1) / 1e6, 3),
            "host": _cpu_model()}


def concurrent_streams(L, synth, torch, a, n, k):
    """Aggregate encode+decode rate of k independent streams sharing the GPU (threads of this process,
    each leasing its own context from the library).  Every stage of a stream runs on the device and its
    latency-bound stages (the recurrence's sweeps, one wavefront per block) leave most of the chip idle, so
    several streams overlap: this is what a multi-file job sees."""
    import threading
    os.environ.setdefault("LZ77X_MAX_CONTEXTS", str(k))
    cap = L.encode_bound(n, a.la, a.sb)
    ins = [torch.from_numpy(synth.make(a.kind, n, synth.SEED_S1 + 100 + i)).cuda() for i in range(k)]
    zs = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(k)]
    backs = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(k)]

    def work(i, reps):
        st = torch.cuda.Stream()
        for _ in range(reps):
            zn = L.encode_device(ins[i].data_ptr(), n, zs[i].data_ptr(), cap, a.la, a.sb, st.cuda_stream)
            L.decode_device(zs[i].data_ptr(), zn, backs[i].data_ptr(), n, st.cuda_stream)

    dt = 0.0
    reps = 3
    for r in (1, reps):                                           # the first round creates the k contexts
        ts = [threading.Thread(target=work, args=(i, r)) for i in range(k)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    ok = all(bool(torch.equal(backs[i], ins[i])) for i in range(k))
    return {"streams": k, "value": round(k * reps * n / dt / 1e6, 3), "unit": "MB/s", "ms_per_step_per_stream": round(dt / reps * 1e3, 3),
            "roundtrip_ok": ok, "note": "not the benchmark value: k independent %d-byte streams on one GPU" % n}


def kernel_source_hash():
    """sha256 over the device sources (csrc/*.hip, kernels_common.h, lz77x_internal.h), in name order: profiles/traffic.json carries
    the hash it was measured on (tools/pmc_summary.py), and a bench line only quotes its counters while the kernels are those."""
    import glob
    import hashlib
    src = os.path.join(ROOT, "lz77_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(src, "*.hip")) + [os.path.join(src, "kernels_common.h"), os.path.join(src, "lz77x_internal.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def file_to_file(L, data, sb, la):
    """what a CLI user sees -- `lz77 -c` then `lz77 -d` as separate processes on a tmpfs file: process
    start, HIP runtime init, file -> pinned slots -> device -> file, process exit.  Three pairs: `first` pays whatever
    the box has not cached yet, `warm` is the best of the three (runtime init alone varies by 50-250 ms from run to
    run, profiles/r04_cli_startup.txt); `process_start_ms` is an encode of an EMPTY file (everything but the data)."""
    import numpy as np
    tmp = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    fin, flz, fout, fempty = (os.path.join(tmp, "lz77f2f_%d.%s" % (os.getpid(), e)) for e in ("in", "lz", "out", "empty"))
    n = int(data.size)
    data.tofile(fin)
    open(fempty, "wb").close()
    geo = ["-s", str(sb), "-l", str(la)]
    runs = []
    try:
        for _ in range(3):
            t0 = time.perf_counter()
            subprocess.check_call([L.CLI_PATH, "-c", "-i", fin, "-o", flz] + geo)
            t1 = time.perf_counter()
            subprocess.check_call([L.CLI_PATH, "-d", "-i", flz, "-o", fout])
            t2 = time.perf_counter()
            runs.append({"encode_ms": round((t1 - t0) * 1e3, 1), "decode_ms": round((t2 - t1) * 1e3, 1)})
        starts = []
        for _ in range(3):
            t0 = time.perf_counter()
            subprocess.check_call([L.CLI_PATH, "-c", "-i", fempty, "-o", flz + ".e"] + geo)
            starts.append((time.perf_counter() - t0) * 1e3)
        ok = bool(np.array_equal(np.fromfile(fout, dtype=np.uint8), data))
    finally:
        for p in (fin, flz, fout, fempty, flz + ".e"):
            if os.path.exists(p):
                os.unlink(p)
    w = {"encode_ms": min(r["encode_ms"] for r in runs), "decode_ms": min(r["decode_ms"] for r in runs)}
    return {"encode_MBps": round(n / w["encode_ms"] / 1e3, 1), "decode_MBps": round(n / w["decode_ms"] / 1e3, 1),
            "encode_plus_decode_MBps": round(n / (w["encode_ms"] + w["decode_ms"]) / 1e3, 1),
            # what the first process of the box pays beyond a later one: the first encode against the best of the OTHER runs
            # (round 5 subtracted the minimum over all three, the first included: 0.0 whenever the first was the fastest)
            "cold_start_ms": round(runs[0]["encode_ms"] - min(r["encode_ms"] for r in runs[1:]), 1), "process_start_ms": round(min(starts), 1),
            "first": runs[0], "warm": w, "runs": runs, "process_start_runs_ms": [round(x, 1) for x in starts],
            "roundtrip_ok": ok, "bytes": n,
            "note": "whole-process wall time of lz77_amd/lz77 on a tmpfs file (process start, HIP runtime init -- 56-250 ms by itself, "
                    "profiles/r04_cli_startup.txt --, PCIe and process exit included); never `value`"}


def other_configs(L, synth, torch):
    """BASELINE.json configs[2] and configs[3] beside the headline line (never `value`): S2 = 1 GiB of raw splitmix64 bytes
    at s=4095 l=15, S3 = 212 MB Silesia-like mixed data at s=65535 l=255.  Each: encode + decode with buffers resident in
    HBM (second of two runs), the digest of the stream against the compiled reference's, the largest kernel's and the whole
    encode's fraction of the HBM roofline (algorithmic bytes n + zn)."""
    import hashlib
    recs = []
    for name, kind, n, seed, sb, la in (("S2", "random", 1 << 30, synth.SEED_S2, 4095, 15), ("S3", "mixed", 212_000_000, synth.SEED_S3, 65535, 255)):
        try:
            data = synth.make(kind, n, seed)
            d_in = torch.from_numpy(data).cuda()
            cap = L.encode_bound(n, la, sb)
            d_z = torch.empty(cap, dtype=torch.uint8, device="cuda")
            d_back = torch.empty(n, dtype=torch.uint8, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                zn = L.encode_device(d_in.data_ptr(), n, d_z.data_ptr(), cap, la, sb, st)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                se = L.last_stats()
                m = L.decode_device(d_z.data_ptr(), zn, d_back.data_ptr(), n, st)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                sd = L.last_stats()
            ok = bool(m == n and torch.equal(d_back, d_in))
            gold = golden_full(kind, n, seed, sb, la)
            sha_ok = None
            if gold is not None:
                h = hashlib.sha256()
                for at in range(0, zn, 1 << 28):
                    h.update(d_z[at:min(at + (1 << 28), zn)].cpu().numpy().tobytes())
                sha_ok = bool(zn == gold["zn"] and h.hexdigest() == gold["sha256_lz"])
            alg = n + zn
            iters = max(int(se["prio_iters"]), 1)
            kern = {"tie-break (k_tokens_sorted / k_tokens_rank_group)": (se["k_tiebreak_ms"], max(int(se["token_launches"]), 1)),
                    "window walkers (k_walk / k_walk_wave)": (se["k_walk_ms"], max(int(se["match_launches"]), 1)),
                    "key sort (k_c1_chunks + k_match / k_big_*)": (se["k_sort_ms"], max(int(se["match_launches"]), 1)),
                    "recurrence forward sweeps (k_prio_fwd / k_pw_fwd), all iterations": (se["k_prio_fwd_ms"], iters)}
            dom = max(kern, key=lambda k: kern[k][0])
            dom_ms, dom_launches = kern[dom]
            t_dev = se["k_match_ms"] + se["k_chain_ms"] + se["k_prio_ms"] + se["k_token_ms"]
            recs.append({"name": name, "workload": "%s %s, %d bytes, s=%d l=%d, buffers resident in HBM" % (name, kind, n, sb, la),
                         "encode_ms": round((t1 - t0) * 1e3, 2), "decode_ms": round((t2 - t1) * 1e3, 2),
                         "encode_MBps": round(n / (t1 - t0) / 1e6, 1), "decode_MBps": round(n / (t2 - t1) / 1e6, 1),
                         "encode_plus_decode_MBps": round(n / (t2 - t0) / 1e6, 1),
                         "stream_sha_ok": sha_ok, "roundtrip_ok": ok, "ratio": round(zn / n, 4), "prio_iters": int(se["prio_iters"]),
                         "host_stageb_ms": round(se["host_stageb_ms"], 2),
                         "dominant_kernel": dom, "dominant_kernel_ms": round(dom_ms, 3), "dominant_kernel_launches": dom_launches,
                         "frac": round(alg / (dom_ms / dom_launches * 1e-3) / 1e9 / HBM_PEAK_GBS / dom_launches, 6) if dom_ms > 0 else None,
                         "frac_op": round(alg / (t_dev * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if t_dev > 0 else None,
                         "frac_decode": round(alg / (sd["k_decode_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if sd["k_decode_ms"] > 0 else None,
                         "kernels_ms": {k: round(se[k], 2) for k in ("k_match_ms", "k_sort_ms", "k_walk_ms", "k_chain_ms", "k_prio_ms", "k_prio_fwd_ms",
                                                                      "k_token_ms", "k_tiebreak_ms")},
                         "k_decode_ms": round(sd["k_decode_ms"], 3)})
            del d_in, d_z, d_back, data
            torch.cuda.empty_cache()
            L.lib().lz77x_shutdown()                               # the next configuration sizes its own buffers
        except Exception as e:                                    # pragma: no cover - must never break the line
            recs.append({"name": name, "error": str(e)[:300]})
    return recs


def shard_record(L, synth, a, shards):
    """BASELINE config 5 inside the default N > 1 run: the 1 GB S4 stream position-sharded over the N devices by ONE
    process (rank 0), host buffers in and out.  No physical scaling is claimed unless the devices are distinct."""
    import hashlib
    n, seed = 1_000_000_000, synth.SEED_S4
    data = synth.make("text", n, seed)
    ndev = L.lib().lz77x_device_count()
    assert L.lib().lz77x_set_shards(max(shards, 1)) == 0
    import numpy as np
    try:
        # the clocks are around the C entry points themselves (encode_c / decode_c: the stream stays in the buffer the library
        # allocated; lz77_amd.encode() would add the copy of a 467 MB stream into a Python bytes object to every figure)
        with L.encode_c(data[:64_000_000], a.la, a.sb) as w:            # contexts, buffers, code objects on every device
            L.decode_c(w.view).close()
        t0 = time.perf_counter()
        zc = L.encode_c(data, a.la, a.sb)
        t1 = time.perf_counter()
        st = L.last_stats()
        t1b = time.perf_counter()
        bc = L.decode_c(zc.view)
        t2 = time.perf_counter()
        z = zc.tobytes()
        roundtrip_ok = bool(np.array_equal(bc.view, data))
        zc.close()
        bc.close()
    finally:
        L.lib().lz77x_set_shards(1)
    gold = golden_full("text", n, seed, a.sb, a.la)
    enc_ms, dec_ms = (t1 - t0) * 1e3, (t2 - t1b) * 1e3
    return {"workload": "S4 enwik9-like text, %d bytes, s=%d l=%d, ONE stream cut into %d position shards on %d physical device(s); "
                        "host buffers in and out (PCIe-inclusive)" % (n, a.sb, a.la, shards, min(shards, ndev)),
            "shards": shards, "physical_devices": min(shards, ndev),
            "encode_ms": round(enc_ms, 1), "decode_ms": round(dec_ms, 1),
            "encode_plus_decode_MBps": round(n / (enc_ms + dec_ms) / 1e3, 1),
            "host_serial_ms": round(st["copy_ms"], 2), "host_serial_frac_of_encode": round(st["copy_ms"] / enc_ms, 4) if enc_ms else None,
            "prio_iters": st["prio_iters"],
            "stream_sha_ok": None if gold is None else bool(len(z) == gold["zn"] and hashlib.sha256(z).hexdigest() == gold["sha256_lz"]),
            "roundtrip_ok": roundtrip_ok,
            "scaling_measured": bool(min(shards, ndev) > 1),
            "note": "strong scaling of one stream (clocks around lz77x_encode / lz77x_decode themselves); `value` above is the weak-scaling "
                    "files mode. host_serial_ms = host time during which no device has work (the parse chain's exchange; per gate iteration, "
                    "from the last shard's device finishing a phase to the first shard's thread enqueuing the next; the pack enqueue)"}


def golden_full(kind, n, seed, sb, la):
    """Reference digest of the full-size stream (tests/golden/golden_full.json, made by the compiled reference)."""
    try:
        recs = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_full.json")))["full"]
    except (OSError, ValueError, KeyError):
        return None
    for r in recs:
        if (r["kind"], r["n"], r["seed"], r["sb"], r["la"]) == (kind, n, seed, sb, la):
            return r
    return None


def stream_sha(torch, d_z, zn):
    import hashlib
    return hashlib.sha256(d_z[:zn].cpu().numpy().tobytes()).hexdigest()


def load_input(a, synth, rank):
    """The step's input: the synthetic S1 stream, or -- when LZ77_CORPUS_DIR holds a file named like the
    workload (enwik8 by default, LZ77_CORPUS_FILE overrides) -- that file.  -> (array, label, seed)"""
    import numpy as np
    cdir = os.environ.get("LZ77_CORPUS_DIR")
    if cdir:
        path = os.path.join(cdir, os.environ.get("LZ77_CORPUS_FILE", "enwik8"))
        if os.path.isfile(path):
            data = np.fromfile(path, dtype=np.uint8)
            if a.bytes and a.bytes < data.size and "--bytes" in sys.argv:
                data = data[:a.bytes]
            return data, "corpus file %s (%d bytes)" % (path, data.size), None
    seed = synth.SEED_S1 + rank
    return synth.make(a.kind, a.bytes, seed), None, seed


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return "%s (%d logical cpus)" % (line.split(":", 1)[1].strip(), os.cpu_count())
    except OSError:
        pass
    return "unknown"


def shard_mode(a):
    """BASELINE config 5: one stream, position-sharded over N devices (lz77x_set_shards), host buffers in and
    out (per-GPU outputs are gathered on the host, as north_star describes) -- so this rate includes PCIe.
    One process (rank 0) drives every device; under torch.distributed.run the other ranks only meet it at the
    barriers."""
    import numpy as np
    import hashlib
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    import lz77_amd as L
    from lz77_amd import synth
    n = a.bytes if "--bytes" in sys.argv else 1_000_000_000
    seed = synth.SEED_S4
    out = None
    if rank == 0:
        ndev = L.lib().lz77x_device_count()
        shards = min(a.gpus, ndev) if not os.environ.get("LZ77X_FAKE_DEVICES") else a.gpus
        data = synth.make(a.kind, n, seed)
        assert L.lib().lz77x_set_shards(max(shards, 1)) == 0
        for _ in range(a.warmup):
            with L.encode_c(data, a.la, a.sb) as zc:
                L.decode(zc.view[:4 + 3 * 1000])
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    if rank == 0:
        # every clock of this mode is around the C entry points themselves (lz77x_encode / lz77x_decode into the buffers the
        # library allocates): L.encode()'s copy of a 467 MB stream into a Python bytes object is not part of the product
        enc_ms, dec_ms, iters, serial_ms = [], [], 0, []
        zc = back = None
        for _ in range(a.steps):
            if zc is not None:
                zc.close()
                back.close()
            t1 = time.perf_counter()
            zc = L.encode_c(data, a.la, a.sb)
            t2 = time.perf_counter()
            st_enc = L.last_stats()
            iters = st_enc["prio_iters"]
            serial_ms.append(st_enc["copy_ms"])
            t2b = time.perf_counter()
            back = L.decode_c(zc.view)
            t3 = time.perf_counter()
            enc_ms.append((t2 - t1) * 1e3)
            dec_ms.append((t3 - t2b) * 1e3)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        K = max(a.steps, 1)
        z = zc.tobytes()
        roundtrip_ok = bool(np.array_equal(back.view, data))
        back.close()
        gold = golden_full(a.kind, n, seed, a.sb, a.la)
        sha_ok = None if gold is None else (len(z) == gold["zn"] and hashlib.sha256(z).hexdigest() == gold["sha256_lz"])
        plan = [p.__dict__ for p in __import__("lz77_amd.shard", fromlist=["plan"]).plan(n, max(shards, 1), a.sb, a.la)]
        # the same stream on ONE context (the device pipeline out of host memory), on the same clock: what Amdahl's law is applied to
        t_one = None
        try:
            assert L.lib().lz77x_set_shards(1) == 0
            L.encode_c(data[:64_000_000], a.la, a.sb).close()
            best = None
            for _ in range(2):
                t4 = time.perf_counter()
                z1 = L.encode_c(data, a.la, a.sb)
                t5 = (time.perf_counter() - t4) * 1e3
                best = t5 if best is None else min(best, t5)
                same = z1.n == len(z) and bool(np.array_equal(z1.view, np.frombuffer(z, dtype=np.uint8)))
                z1.close()
                assert same
            t_one = best
        finally:
            L.lib().lz77x_set_shards(max(shards, 1))
        hs = sum(serial_ms) / K
        amdahl = None
        if t_one:
            amdahl = {"encode_ms_one_context": round(t_one, 1), "host_serial_ms": round(hs, 2),
                      "host_serial_ms_per_gate_iteration": round(hs / max(iters, 1), 3),
                      "bound_encode_ms": {str(d): round(hs + max(t_one - hs, 0.0) / d, 1) for d in (2, 4, 8)},
                      "bound_speedup": {str(d): round(t_one / (hs + max(t_one - hs, 0.0) / d), 2) for d in (2, 4, 8)},
                      "clock": "time.perf_counter around lz77x_encode itself (lz77_amd.encode_c: no copy into a bytes object), for T(1) and for the sharded run alike",
                      "note": "prediction, not a measurement: T(D) >= host_serial + (T(1) - host_serial) / D with host_serial = the host "
                              "time of THIS run during which no device has work: the parse chain's exchange; per gate iteration, twice, from the "
                              "moment the LAST shard's device finished a phase to the moment the FIRST shard's thread enqueues the next "
                              "(the D boundary maps chained on the host, the D flip summaries read, and the wake-ups of the barriers the "
                              "per-shard host threads meet at); the pack enqueue -- it grows with D (one map per shard) and is "
                              "measured here at D = %d on %d physical device(s).  The bound ignores the LATENCY floor of the per-shard stages: a forward sweep "
                              "of the recurrence takes ~0.45 ms per gate iteration whatever the shard's size, the window walkers 0.8 ms (DESIGN.md "
                              "section 2.2), so T(8) on a 1 GB stream is nearer 25 ms than T(1) / 8 = 15" % (len(plan), min(len(plan), L.lib().lz77x_device_count()))}
        out = {"metric": "encode+decode MB/s on enwik9-like synthetic text, s=%d l=%d, ONE stream position-sharded" % (a.sb, a.la),
               "value": round(n * K / dt / 1e6, 3), "unit": "MB/s",
               "n_gpus": min(len(plan), L.lib().lz77x_device_count()),         # one process drives every device: physical devices used
               "contexts": len(plan), "scaling_measured": bool(min(len(plan), L.lib().lz77x_device_count()) > 1),
               "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "u8", "data": "synthetic",
               "config": {"workload": "S4 enwik9-like text (lz77_amd.synth.text, seed 0x5EED0004), %d bytes, s=%d l=%d, cut into %d shards "
                                      "on %d device(s); host buffers in and out (PCIe-inclusive); decode sharded by token ranges "
                                      "(sb-byte maps chained on the host)" %
                                      (n, a.sb, a.la, len(plan), min(len(plan), L.lib().lz77x_device_count())),
                          "mode": "shard", "shards": len(plan), "max_local_bytes": max(p["local_bytes"] for p in plan)},
               "encode_ms": round(sum(enc_ms) / K, 2), "decode_ms": round(sum(dec_ms) / K, 2), "prio_iters": iters,
               "host_serial_ms": round(hs, 2), "amdahl": amdahl,
               "roundtrip_ok": roundtrip_ok, "stream_sha_ok": sha_ok,
               "serial_terms": "per gate iteration one host exchange of 6 KB per shard (priority cells), one of 1.3 KB per shard for "
                               "the parse chain, 16 bytes per cut for packing; decode: one map of sb 16-bit states per shard, chained "
                               "on the host; everything else is shard-local"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def respawn_if_needed(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: files mode is one process per GPU, so re-execute
    under torch.distributed.run with N ranks (the driver's own launch line) instead of printing an n_gpus = 1 line.  A
    launcher whose world size disagrees with --gpus is an error, not a silently different run."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != a.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: launch one rank per GPU "
                             "(python -m torch.distributed.run --nproc-per-node %d ... bench.py --gpus %d)" % (a.gpus, ws, a.gpus, a.gpus))
        return
    if a.gpus <= 1:
        return
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bytes", type=int, default=100_000_000)
    ap.add_argument("--sb", type=int, default=4095)
    ap.add_argument("--la", type=int, default=15)
    ap.add_argument("--kind", default="text")
    ap.add_argument("--cpu-sample", type=int, default=100_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-file-to-file", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the S2 / S3 records (BASELINE configs[2], configs[3])")
    ap.add_argument("--no-shard-record", action="store_true", help="N > 1: skip the S4 one-stream-sharded sub-record")
    ap.add_argument("--streams", type=int, default=4, help="also report k concurrent streams on one GPU (informational; 1 = skip)")
    ap.add_argument("--mode", choices=("files", "shard"), default="files",
                    help="files: an independent stream per GPU (default; weak scaling).  shard: ONE stream (default the 1 GB "
                         "S4 'enwik9-like' input, BASELINE config 5) cut by position over --gpus devices, driven by rank 0")
    a = ap.parse_args()
    if a.mode == "shard":
        return shard_mode(a)
    respawn_if_needed(a)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("LZ77_BENCH_BACKEND", "nccl")     # "gloo": several ranks on one GPU (plumbing test only)
    if local >= ndev and backend == "nccl":
        raise SystemExit("rank %d has no GPU (%d visible)" % (rank, ndev))
    local %= ndev
    torch.cuda.set_device(local)
    numa_cpus = 0
    if os.environ.get("LZ77_BENCH_NUMA", "1") != "0":
        from lz77_amd.shard import bind_to_device_numa
        pr = torch.cuda.get_device_properties(local)
        try:
            numa_cpus = bind_to_device_numa("%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id))
        except AttributeError:
            numa_cpus = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        # host-side rendezvous for the part outside the timed region where rank 0 alone drives every device: an RCCL
        # barrier would park a spinning kernel on the very GPUs rank 0 is using
        cpu_group = dist.new_group(backend="gloo")
    import lz77_amd as L
    from lz77_amd import synth

    data, corpus_label, seed = load_input(a, synth, rank)
    n = int(data.size)
    d_in = torch.from_numpy(data).cuda()
    cap = L.encode_bound(n, a.la, a.sb)
    d_z = torch.empty(cap, dtype=torch.uint8, device="cuda")
    d_back = torch.empty(n, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        zn = L.encode_device(d_in.data_ptr(), n, d_z.data_ptr(), cap, a.la, a.sb, stream)
        se = L.last_stats()
        m = L.decode_device(d_z.data_ptr(), zn, d_back.data_ptr(), n, stream)
        sd = L.last_stats()
        assert m == n
        return zn, se, sd

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    enc_stats, dec_stats = [], []
    zn = 0
    for _ in range(a.steps):
        zn, se, sd = step()
        enc_stats.append(se)
        dec_stats.append(sd)
    barrier()
    dt = time.perf_counter() - t0
    ok = bool(torch.equal(d_back, d_in))                 # round trip checked outside the timed region
    # bit-exactness at the headline size: sha256 of the device stream against the reference's (a round
    # trip cannot see a wrong tie-break offset)
    gold = golden_full(a.kind, n, seed, a.sb, a.la) if seed is not None else None
    sha_ok = None
    if gold is not None:
        sha_ok = zn == gold["zn"] and stream_sha(torch, d_z, zn) == gold["sha256_lz"]
    if world > 1:
        red_dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        okt = torch.tensor([1 if ok else 0, 2 if sha_ok is None else (1 if sha_ok else 0)], device=red_dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt[0].item())
        sha_ok = None if int(okt[1].item()) == 2 else bool(okt[1].item())

    shard_rec = None
    if world > 1 and not a.no_shard_record:
        # BASELINE config 5 (one stream over the N devices), outside the timed region: rank 0 drives every device, the
        # other ranks wait at the barrier
        if rank == 0:
            try:
                del d_z, d_back
                torch.cuda.empty_cache()
                shard_rec = shard_record(L, synth, a, world)
            except Exception as e:                                # pragma: no cover - must never break the line
                shard_rec = {"error": str(e)[:300]}
        dist.barrier(group=cpu_group)

    if rank == 0:
        K = max(a.steps, 1)
        mean = lambda xs, k: sum(x[k] for x in xs) / max(len(xs), 1)
        launches = max(int(mean(enc_stats, "match_launches")), 1)
        tlaunches = max(int(mean(enc_stats, "token_launches")), 1)
        k_match_ms = mean(enc_stats, "k_match_ms")               # sort + window walkers + finalize
        alg_bytes = n + zn                                        # encode reads n, writes zn
        # the big kernels of an encode, each timed by its own hipEvent pair on the stream it runs on
        iters = max(int(round(mean(enc_stats, "prio_iters"))), 1)
        cands = {
            "k_tokens_sorted (offset tie-break: equal-length candidates as runs of the regions' sorted order)": (mean(enc_stats, "k_tiebreak_ms"), tlaunches, "k_tokens_sorted"),
            "k_walk (bitmap window walkers: in-order neighbours of every position)": (mean(enc_stats, "k_walk_ms"), launches, "k_walk"),
            "k_c1_chunks (key sort of every 4 K chunk of positions)": (mean(enc_stats, "k_sort_chunks_ms"), launches, "k_c1_chunks"),
            "k_match<true,3> (a region's last two merge levels + rank export)": (mean(enc_stats, "k_sort_ms") - mean(enc_stats, "k_sort_chunks_ms"), launches, "k_match"),
            "k_prio_fwd (priority recurrence: forward sweep of one gate iteration)": (mean(enc_stats, "k_prio_fwd_ms"), iters, "k_prio_fwd"),
        }
        dom_name = max(cands, key=lambda k: cands[k][0])
        dom_ms, dom_launches, dom_key = cands[dom_name]
        if dom_ms <= 0:
            dom_name, dom_ms, dom_launches, dom_key = "match stage (sort + walkers + finalize)", k_match_ms, launches, None
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        t_device_ms = sum(mean(enc_stats, k) for k in ("k_match_ms", "k_chain_ms", "k_prio_ms", "k_token_ms"))
        traffic = None
        traffic_total = None
        traffic_hash = None
        traffic_current = False
        issue = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile) and dom_key:
            try:
                tj = json.load(open(tfile))
                traffic_hash = tj.get("kernel_source_hash")
                traffic_current = traffic_hash is not None and traffic_hash == kernel_source_hash()
                if traffic_current:
                    traffic = tj["hbm_bytes_per_launch"].get(dom_key)
                    traffic_total = tj.get("encode_hbm_bytes")
                valu = tj.get("valu_wave_insts_per_launch", {}).get(dom_key) if traffic_current else None
                if valu and dom_ms > 0:
                    # none of the encode's kernels is HBM-bound: the view that prices them is instruction issue.  A wave64 VALU
                    # instruction occupies its SIMD for 2.8 (v_add / v_xor) to 4.5 cycles (v_cndmask, v_alignbyte, 64-bit shifts,
                    # v_bfe / v_mbcnt) with two or more waves per SIMD -- tests/ubench/issue_rates.hip, profiles/r05_issue_rates.txt
                    simd_cycles = dom_ms / dom_launches * 1e-3 * CLOCK_HZ * N_SIMD
                    issue = {"valu_wave_insts_per_launch": valu, "cycles_per_valu_inst": VALU_CYCLES, "simd_cycles_per_launch": int(simd_cycles),
                             "issue_frac": round(valu * VALU_CYCLES / simd_cycles, 4),
                             "issue_frac_range": [round(valu * 2.8 / simd_cycles, 4), round(valu * 4.5 / simd_cycles, 4)],
                             "salu_wave_insts_per_launch": tj.get("salu_wave_insts_per_launch", {}).get(dom_key),
                             "wait_any_over_wave_cycles": tj.get("wait_any_over_wave_cycles", {}).get(dom_key),
                             "source": "instruction counts: profiles/traffic.json (SQ_INSTS_VALU of the committed rocprofv3 --pmc run, per launch); "
                                       "duration: this run's hipEvent pair; cycles per instruction: the measured range of the ubench, 3.5 taken for "
                                       "a mix of compares, selects, funnel shifts and adds"}
            except Exception:
                traffic = None
        out = {
            "metric": "encode+decode MB/s on %s, s=%d l=%d" % ("enwik8-like synthetic text" if corpus_label is None else "a supplied corpus file", a.sb, a.la),
            "value": round(world * n * K / dt / 1e6, 3),
            "unit": "MB/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt / K * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" if corpus_label is None else "file",
            "config": {"workload": ("S1 enwik8-like text (lz77_amd.synth.text, seed 0x5EED0001+rank)" if corpus_label is None else corpus_label) +
                                   ", %d bytes per GPU, s=%d l=%d; step = encode then decode, buffers resident in HBM" % (n, a.sb, a.la),
                       "mode": "files", "bytes_per_gpu": n, "sb": a.sb, "la": a.la, "parallelism": "independent stream per GPU",
                       "numa_bound_cpus": numa_cpus},
            "roofline": {"bound": "hbm", "kernel": dom_name + " -- the largest GPU kernel of the step",
                         "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "issue_frac": issue["issue_frac"] if issue else None, "issue": issue,
                         "traffic_total": traffic_total,
                         "traffic_total_def": "HBM bytes of ALL kernels of one encode by the same counters (beside frac_op: the whole operation's traffic against its algorithmic bytes)",
                         "traffic_source": ("profiles/traffic.json (PMC passes of the committed rocprofv3 run, per launch; not measured in this run), taken on the "
                                            "kernel sources of this tree (hash %s)" % traffic_hash) if traffic_current else
                                           ("null: profiles/traffic.json was measured on other kernel sources (hash %s, this tree %s) -- rerun tools/evidence.sh"
                                            % (traffic_hash, kernel_source_hash())),
                         "frac_op": round(alg_bytes / (t_device_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if t_device_ms > 0 else 0.0,
                         "frac_op_def": "SURVEY 8d: algorithmic bytes / SUM of the encode's kernel times (match + chain + recurrence + "
                                        "tie-break with its hand-over lists, and pack) / peak -- the whole operation, beside the dominant kernel's `frac`",
                         "t_device_ms": round(t_device_ms, 3),
                         "launches_per_step": dom_launches,
                         "algorithmic_bytes_per_launch": alg_bytes // dom_launches,
                         "kernel_ms_per_launch": round(dom_ms / dom_launches, 3),
                         "kernels_ms_per_step": {v[2]: round(v[0], 3) for v in cands.values()},
                         "match_stage_ms": round(k_match_ms, 3),
                         "match_stage_GBps": round(alg_bytes / (k_match_ms * 1e-3) / 1e9, 3) if k_match_ms > 0 else 0.0},
            "roofline_decode": {"bound": "hbm", "kernel": "all decode kernels (parse, scan, segment walk with an LDS ring, tail chain, patch)",
                                "achieved": round(alg_bytes / (mean(dec_stats, "k_decode_ms") * 1e-3) / 1e9, 3) if mean(dec_stats, "k_decode_ms") > 0 else 0.0,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(alg_bytes / (mean(dec_stats, "k_decode_ms") * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if mean(dec_stats, "k_decode_ms") > 0 else 0.0,
                                "algorithmic_bytes": alg_bytes, "kernels_ms": round(mean(dec_stats, "k_decode_ms"), 3)},
            "roundtrip_ok": ok,
            "stream_sha_ok": sha_ok,
            "stream_sha_source": "tests/golden/golden_full.json (sha256 of the compiled reference's stream for this input)" if gold else None,
            "ratio": round(zn / n, 4),
            "encode_MBps": round(n / (mean(enc_stats, "total_ms") * 1e-3) / 1e6, 2),
            "decode_MBps": round(n / (mean(dec_stats, "total_ms") * 1e-3) / 1e6, 2),
            "encode_breakdown_ms": {k: round(mean(enc_stats, k), 2) for k in
                                    ("total_ms", "k_match_ms", "k_sort_ms", "k_sort_chunks_ms", "k_walk_ms", "k_token_ms", "k_tiebreak_ms", "k_prio_ms",
                                     "k_prio_fwd_ms", "k_prio_back_ms", "k_prio_scan_ms", "k_chain_ms", "host_chain_ms",
                                     "host_stageb_ms", "copy_ms")},
            "prio_iters": iters,
            "decode_breakdown_ms": {k: round(mean(dec_stats, k), 2) for k in ("total_ms", "k_decode_ms")},
        }
        if world == 1 and not a.no_configs and corpus_label is None:
            # BASELINE configs[2] and [3], driver-observed beside the headline (outside the timed region, never `value`)
            del d_in, d_z, d_back
            torch.cuda.empty_cache()
            L.lib().lz77x_shutdown()
            out["configs"] = other_configs(L, synth, torch)
            d_in = d_z = d_back = None
        if world == 1 and a.streams > 1:
            # informational, never `value`: k independent streams on the one GPU, one thread each
            try:
                out["concurrent_streams"] = concurrent_streams(L, synth, torch, a, n, a.streams)
            except Exception as e:                                # pragma: no cover - must never break the line
                out["concurrent_streams"] = {"error": str(e)[:200]}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(data, a.sb, a.la, min(a.cpu_sample, n))
        if world == 1 and not a.no_file_to_file:
            try:
                out["file_to_file"] = file_to_file(L, data, a.sb, a.la)
            except Exception as e:                                # pragma: no cover - must never break the line
                out["file_to_file"] = {"error": str(e)[:200]}
        if shard_rec is not None:
            out["shard"] = shard_rec
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
