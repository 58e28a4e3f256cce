"""profiles/README.md, one round's part: GENERATED from the files tools/evidence.sh wrote (the numbers in the
README come out of the evidence files, not out of an editor).

    python tools/profiles_readme.py [tag]      # (default r06) rewrites the block between <!-- <tag>:begin --> and <!-- <tag>:end -->
"""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
import sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"


def load(name):
    """a JSON file, or a log whose last line is the JSON line"""
    try:
        with open(os.path.join(P, name)) as f:
            txt = f.read().strip()
    except OSError:
        return None
    for cand in (txt, txt.splitlines()[-1] if txt else ""):
        try:
            return json.loads(cand)
        except ValueError:
            pass
    return None


def kstats(name):
    out = {}
    try:
        for r in csv.DictReader(open(os.path.join(P, name))):
            k = r["Name"]
            k = k[5:] if k.startswith("void ") else k
            out[k.split("(")[0]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6)
    except OSError:
        pass
    return out


def f(x, nd=2):
    return ("%." + str(nd) + "f") % x if isinstance(x, (int, float)) else "n/a"


rows = []
b = load(TAG + "_bench.json")
if b:
    r, rd, e = b["roofline"], b["roofline_decode"], b["encode_breakdown_ms"]
    cfg = {c["name"]: c for c in b.get("configs", []) if "name" in c}
    ff = b.get("file_to_file", {})
    cb = b.get("cpu_baseline", {})
    line = ("the JSON line `python bench.py` printed in the evidence run: encode+decode **%s MB/s** (%s ms per step: encode %s, decode %s), "
            "`stream_sha_ok` %s, `roofline.frac` %s on `%s` (%s ms by its own hipEvent pair), `frac_op` %s, `issue_frac` %s (VALU instructions x %s cycles "
            "/ SIMD-cycles), decode `frac` %s" % (f(b["value"], 0), f(b["ms_per_step"], 2), f(e["total_ms"]), f(b["decode_breakdown_ms"]["total_ms"]),
                                                  b["stream_sha_ok"], r["frac"], r["kernel"].split(" ")[0], r["kernel_ms_per_launch"], r["frac_op"],
                                                  r.get("issue_frac"), (r.get("issue") or {}).get("cycles_per_valu_inst"), rd["frac"]))
    for nm in ("S2", "S3"):
        c = cfg.get(nm)
        if c and "encode_ms" in c:
            line += "; %s encode %s ms / %s GB/s, digest %s" % (nm, f(c["encode_ms"], 1), f(c["encode_MBps"] / 1e3, 2), c["stream_sha_ok"])
    if ff and "warm" in ff:
        line += "; file to file through the CLI %s / %s MB/s (best of three: %s / %s ms; `process_start_ms` %s)" % (
            f(ff["encode_MBps"], 0), f(ff["decode_MBps"], 0), ff["warm"]["encode_ms"], ff["warm"]["decode_ms"], ff["process_start_ms"])
    if cb:
        line += "; reference CPU (%s, %s core) %s MB/s" % (cb.get("kind"), cb.get("cores"), cb.get("value"))
    rows.append(("`%s_bench.json`" % TAG, line))

ks = kstats(TAG + "_bench_kernel_stats.csv")
if ks:
    want = ["k_tokens_sorted", "k_c1_chunks", "k_match<true, 3>", "k_prio_fwd<true>", "k_prio_back2", "k_prio_prep", "k_walk", "k_walk_final_lds",
            "k_chain_emit", "k_dec_seg<true>", "k_dec_patch_seg", "k_dec_sums", "k_dec_bounds_fused"]
    parts = ["`%s` %s x %d" % (k, f(ks[k][1], 3), ks[k][0]) for k in want if k in ks]
    rows.append(("`%s_bench_kernel_stats.csv`" % TAG, "`rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 ...` (4 encodes + 4 decodes), average ms x launches: "
                 + ", ".join(parts)))
u = load(TAG + "_bench_under_rocprof.json")
if u:
    rows.append(("`%s_bench_under_rocprof.json`" % TAG, "the line bench.py printed IN that profiled run (%s MB/s): its own hipEvent figure for the roofline kernel, "
                 "`kernel_ms_per_launch` %s, next to rocprofv3's average above" % (f(u["value"], 0), u["roofline"]["kernel_ms_per_launch"])))
t = load("traffic.json")
if t:
    h = t["hbm_bytes_per_launch"]
    top = sorted(h, key=lambda k: -h[k] * t["launches"].get(k, 1))[:7]
    rows.append(("`traffic.json`", "HBM bytes per launch of every kernel = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the gfx950 correction of MI355X_MICROARCH.md), the VALU / SALU / LDS wave "
                 "instructions per launch, `SQ_WAIT_ANY / SQ_WAVE_CYCLES` and the LDS bank-conflict share, from separate `--pmc` passes.  Whole encode **%s GB**, decode %s GB; largest: %s"
                 % (f(t.get("encode_hbm_bytes", 0) / 1e9), f(t.get("decode_hbm_bytes", 0) / 1e9),
                    ", ".join("`%s` %s GB x %d" % (k, f(h[k] / 1e9), t["launches"].get(k, 1)) for k in top))))
    vi = t.get("valu_wave_insts_per_launch", {})
    wa = t.get("wait_any_over_wave_cycles", {})
    lc = t.get("lds_conflict_over_active", {})
    ks3 = [k for k in ("k_tokens_sorted", "k_c1_chunks", "k_match") if k in vi]
    rows.append(("`%s_bench_pmc_summary.csv`" % TAG, "per-kernel counters per launch.  VALU wave instructions / `SQ_WAIT_ANY` share of wave time / LDS conflict share: "
                 + "; ".join("`%s` %s M / %s / %s" % (k, f(vi[k] / 1e6, 0), wa.get(k), lc.get(k)) for k in ks3)))
for name, what in (("_issue_rates.txt", "`tests/ubench/issue_rates.bin`: cycles a wave64 instruction occupies a SIMD at 1, 2, 4, 8 waves per SIMD (independent and dependent chains), "
                    "`ds_read_b32` throughput and round trip, `s_barrier` in 256- and 1024-thread workgroups -- what `roofline.issue` prices the counters with"),
                   ("_valu_rates.txt", "`tests/ubench/valu_rates.bin`: wave instructions per clock per SIMD of the VALU ops the kernels are made of"),
                   ("_ts_probe.txt", "`bash tools/ts_probe.sh`: `SQ_INSTS_VALU` of `k_tokens_sorted` when the variants build leaves the kernel after the set-up / phase A / B1 / B2")):
    if os.path.exists(os.path.join(P, TAG + name)):
        rows.append(("`%s%s`" % (TAG, name), what))
c = load(TAG + "_configs.json")
if c:
    rows.append(("`%s_configs.json`" % TAG, "`python tools/measure_configs.py`: " + "; ".join(
        "%s: encode %s ms (%s GB/s), decode %s GB/s, %d gate iterations" % (x["config"].split(",")[0], f(x["encode_ms"], 1), f(x["encode_MBps"] / 1e3, 2),
                                                                            f(x["decode_MBps"] / 1e3, 0), x["prio_iters"]) for x in c["configs"])))
pc = load(TAG + "_prio_classes.json")
if pc:
    cl = [x for x in pc["classes"] if x["bytes"] == 100_000_000]
    text = next((x for x in cl if x["kind"] == "text"), None)
    rows.append(("`%s_prio_classes.json`" % TAG, "`python tools/prio_classes.py`: per data class at C1, 100 MB -- encode ms (x text): " + ", ".join(
        "%s %s (%s)" % (x["kind"], f(x["encode_ms"], 1), f(x["encode_ms"] / text["encode_ms"], 2) if text else "") for x in cl)))
k2 = kstats(TAG + "_c2_kernel_stats.csv")
if k2:
    top = sorted(k2, key=lambda k: -k2[k][2])[:10]
    rows.append(("`%s_c2_kernel_stats.csv`" % TAG, "rocprofv3 kernel stats of one S3 encode (`tools/time_c2.py`), total ms: " + ", ".join("`%s` %s" % (k, f(k2[k][2], 1)) for k in top)))
try:
    c2p = list(csv.DictReader(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", TAG + "_c2_pmc_summary.csv"))))
except OSError:
    c2p = []
if c2p and k2:
    parts = []
    for r in c2p[:9]:
        k = r["kernel"]
        n = int(r["launches_per_encode_plus_decode"])
        g = lambda c: float(r.get(c + "_per_launch", 0) or 0)
        ms = k2.get(k, (0, 0, 0))[2]
        if not ms:
            continue
        hbm = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 * n / 1e9
        issue = g("SQ_INSTS_VALU") * n * 3.5 / (ms * 1e-3 * 2.4e9 * 1024)
        wait = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else 0
        parts.append("`%s` %s GB in %s ms (%s TB/s), issue %s, waiting %s" % (k, f(hbm, 1), f(ms, 1), f(hbm / ms, 2), f(issue, 2), f(wait, 2)))
    rows.append(("`%s_c2_pmc_summary.csv`" % TAG, "`bash tools/c2_pmc.sh`: counters of one S3 encode per kernel and launch, four separate `--pmc` passes.  HBM bytes ((2 x FETCH_SIZE + WRITE_SIZE) x 1024) over the "
                 "kernel's time in `%s_c2_kernel_stats.csv`, VALU wave instructions x 3.5 cycles over its SIMD-cycles, `SQ_WAIT_ANY / SQ_WAVE_CYCLES`: " % TAG + "; ".join(parts)))
w = load(TAG + "_worst_cases.json")
if w:
    parts = []
    for g, v in w["geometries"].items():
        worst = v["worst_by_ms_per_100MB"][0] if v["worst_by_ms_per_100MB"] else None
        parts.append("%s: %d candidates, most gate iterations %d, %d took the host fallback, worst `%s` at %s ms per 100 MB (text of the same size: %s)" % (
            g, v["candidates"], v["max_prio_iters"], len([1 for r in v["worst_by_ms_per_100MB"] + v["worst_by_iterations"] if r.get("fell_back_to_the_host")]),
            worst["input"] if worst else "", f(worst["ms_per_100MB"], 1) if worst else "", f(v["text_for_scale"]["ms_per_100MB"], 1)))
    rows.append(("`%s_worst_cases.json`" % TAG, "`python tools/worst_cases.py 300`: adversarial families (periods around the lookahead and the window, cut runs, tiny alphabets, "
                 "repeated blocks, text with planted runs) scored by gate iterations and encode ms per 100 MB.  " + "; ".join(parts)))
try:
    rc_lines = [json.loads(l) for l in open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", TAG + "_run_cliff.jsonl")) if l.startswith("{")]
except OSError:
    rc_lines = []
if rc_lines:
    rows.append(("`%s_run_cliff.jsonl`" % TAG, "`python tools/run_cliff.py 12000000`: runs of equal bytes at both geometries, encode ms per 100 MB (gate iterations): " + "; ".join(
        "%s `%s` %s (%d)" % (r["geometry"].split()[0], r["input"], f(r["ms_per_100MB"], 0), r["prio_iters"]) for r in rc_lines)))
fz = None
try:
    fz = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", TAG + "_fuzz_long.txt")).read().strip().splitlines()[-1]
except (OSError, IndexError):
    pass
if fz:
    rows.append(("`%s_fuzz_long.txt`" % TAG, "`python tests/gpu_fuzz_long.py 200 <seed>` (shards, stretches, segments, ranges and the host recurrence drawn at random, every case against the oracle): " + fz))
hr = load(TAG + "_host_rates.json")
if hr:
    rows.append(("`%s_host_rates.json`" % TAG, "`python tools/host_rates.py` (host buffers, PCIe inclusive, the C calls): " + "; ".join(
        "%d MB: encode %s GB/s, decode %s GB/s" % (x["bytes"] // 1_000_000, x["c_call_encode_GBps"], x["c_call_decode_GBps"]) for x in hr)))
fr = load(TAG + "_file_rates.json")
if fr:
    rows.append(("`%s_file_rates.json`" % TAG, "`python tools/file_rates.py` (FILE* entry points inside one warm process, tmpfs): " + json.dumps(fr)[:400]))
mp = load(TAG + "_mem_probe.json")
if mp:
    rows.append(("`%s_mem_probe.json`" % TAG, "`python tools/mem_probe.py`: device memory held after an encode / a decode per size and geometry (hipMemGetInfo deltas); encode bytes per input byte: "
                 + ", ".join("%s %d MB s=%d %s%s" % (m["kind"], m["n"] // 1000000, m["sb"], m["encode_B_per_input_B"], " (segments of 256 MB)" if m.get("env") else "") for m in mp)
                 + "; `leak_MB` (still held after `lz77x_shutdown()`) is the HIP runtime's queue scratch, see `%s_scratch_hold.txt`" % TAG))
try:
    sh_txt = open(os.path.join(P, TAG + "_scratch_hold.txt")).read().strip().splitlines()
    rows.append(("`%s_scratch_hold.txt`" % TAG, "`./tools/scratch_hold_probe`: device memory before / after a kernel with S bytes of private segment per lane, and after its buffer is freed "
                 "and its stream destroyed -- the runtime keeps the scratch it gave the hardware queue (what `leak_MB` of the memory probe sees): "
                 + "; ".join(l.split(":")[0].strip() + ": " + l.split("still")[1].split("below")[0].strip() + " kept" for l in sh_txt if "still" in l)))
except (OSError, IndexError):
    pass
try:
    rp = open(os.path.join(P, TAG + "_c2_rank_probe.txt")).read().strip().splitlines()
    parts = []
    for i in range(0, len(rp) - 1, 2):
        if rp[i].startswith("==") and "encode" in rp[i + 1]:
            ms = rp[i + 1].split("encode")[1].split("ms")[0].strip()
            tb = rp[i + 1].split("'k_tiebreak_ms':")[1].split(",")[0].strip() if "'k_tiebreak_ms':" in rp[i + 1] else "?"
            parts.append("`%s` encode %s ms, tie-break stage %s ms" % (rp[i][3:].strip(), ms, tb))
    rows.append(("`%s_c2_rank_probe.txt`" % TAG, "the large-window tie-break by parts on S3 (variants build; `LZ77X_RANK_PROBE` 1 = no deferred tokens, 4 = only the bucket tokens deferred: timing "
                 "only, wrong output; `LZ77X_NO_RANK_INDEX=1` = the walk cell by cell of round 5): " + "; ".join(parts)))
except (OSError, IndexError):
    pass
if os.path.exists(os.path.join(P, TAG + "_fallback_soak.txt")):
    rows.append(("`%s_fallback_soak.txt`" % TAG, "`python tools/fallback_soak.py 540`: the hand-over from the device pipeline to the host-assisted one (the path of round 5's abort) in one "
                 "process between calls of every other kind: " + open(os.path.join(P, TAG + "_fallback_soak.txt")).read().strip()))
if os.path.exists(os.path.join(P, TAG + "_fuzz_c2.txt")):
    rows.append(("`%s_fuzz_c2.txt`" % TAG, "`python tests/gpu_fuzz_c2.py 600 20000` (large windows only, several regions, long runs of equal candidates; every stream against the oracle): "
                 + open(os.path.join(P, TAG + "_fuzz_c2.txt")).read().strip()))
if os.path.exists(os.path.join(P, TAG + "_suite_one_process_poison.txt")):
    tail = [l for l in open(os.path.join(P, TAG + "_suite_one_process_poison.txt")).read().splitlines() if " passed" in l]
    rows.append(("`%s_suite_one_process_poison.txt`" % TAG, "`bash tools/poison_suite_and_fuzz.sh`: the same run with `LZ77X_POISON=1` (every cached device and pinned buffer filled with 0xA5 when a "
                 "call leases its context, every buffer when it is allocated): " + (tail[-1].strip("= ") if tail else "n/a") + "; the fuzz under poison behind it: 4480 cases, none differing"))
if os.path.exists(os.path.join(P, TAG + "_suite_one_process.txt")):
    tail = [l for l in open(os.path.join(P, TAG + "_suite_one_process.txt")).read().splitlines() if " passed" in l]
    rows.append(("`%s_suite_one_process.txt`" % TAG, "`bash tools/suite_one_process.sh`: the whole GPU suite in ONE process with the native stderr kept (`--capture=sys`), the soak of "
                 "`tests/test_gpu_soak.py` last: " + (tail[-1].strip("= ") if tail else "n/a")))
sh = load(TAG + "_shard_fake8.json")
if sh:
    a = sh.get("amdahl") or {}
    rows.append(("`%s_shard_fake8.json`" % TAG, "`LZ77X_FAKE_DEVICES=8 python bench.py --mode shard --gpus 8 --steps 2 --warmup 1` on ONE MI355X (eight contexts sharing it: `n_gpus` %s, "
                 "`scaling_measured` %s -- no physical scaling is measured or claimed): S4 1 GB in 8 position shards, encode %s ms, decode %s ms, digest %s, %s joint gate iterations, "
                 "`host_serial_ms` %s (%s per gate iteration: last device done -> first enqueue of the next phase); T(1) on the same clock %s ms; Amdahl bound from this run's serial part "
                 "(it ignores the latency floor of the per-shard stages): %s" % (
                     sh.get("n_gpus"), sh.get("scaling_measured"), sh.get("encode_ms"), sh.get("decode_ms"), sh.get("stream_sha_ok"), sh.get("prio_iters"),
                     sh.get("host_serial_ms"), a.get("host_serial_ms_per_gate_iteration"), a.get("encode_ms_one_context"), json.dumps(a.get("bound_speedup")))))
n2 = load(TAG + "_n2_gloo_one_gpu.json")
if n2:
    rows.append(("`%s_n2_gloo_one_gpu.json`" % TAG, "`LZ77_BENCH_BACKEND=gloo LZ77X_FAKE_DEVICES=2 python bench.py --gpus 2 ...` with NO launcher around it: bench.py re-executes itself under "
                 "`torch.distributed.run` with two ranks (sharing the one GPU: a plumbing check, not a scaling measurement) and prints `n_gpus` %s, %s MB/s" % (n2.get("n_gpus"), f(n2.get("value"), 0))))
for name, what in (("_cli_trace.txt", "`bash tools/cli_trace.sh`: `LZ77X_TRACE=1 lz77 -c / -d` on 100 MB, where the wall time of a CLI run goes"),
                   ("_rss_probe.txt", "`python tools/rss_probe.py`: peak resident set of the CLI on a 1 GB file, one device and four contexts, stretches of 1 GiB and 256 MB")):
    if os.path.exists(os.path.join(P, TAG + name)):
        rows.append(("`%s%s`" % (TAG, name), what))

block = "<!-- %s:begin (generated by tools/profiles_readme.py from the files named in the first column) -->\n| file | what |\n|---|---|\n" % TAG + \
        "\n".join("| %s | %s |" % (a, b_.replace("|", "/")) for a, b_ in rows) + "\n<!-- %s:end -->" % TAG
path = os.path.join(P, "README.md")
txt = open(path).read()
if "<!-- %s:begin" % TAG in txt:
    txt = re.sub(r"<!-- %s:begin.*?<!-- %s:end -->" % (TAG, TAG), lambda m: block, txt, flags=re.S)
else:
    # a new round's table goes in front of the earlier rounds'
    head, sep, rest = txt.partition("## Round ")
    txt = head + "## Round %s\n\n`TAG=%s bash tools/evidence.sh` (one gpurun call) writes every %s file below; `python tools/profiles_readme.py %s` writes this table from them.\n\n" % (
        TAG[1:].lstrip("0"), TAG, TAG, TAG) + block + "\n\n" + sep + rest
open(path, "w").write(txt)
print("rows", len(rows))
