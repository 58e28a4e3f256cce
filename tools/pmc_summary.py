"""rocprofv3 counter_collection csvs of tools/evidence.sh -> <tag>_bench_pmc_summary.csv and traffic.json.
usage: python tools/pmc_summary.py <out dir> <tag>"""
import collections, csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k[5:] if k.startswith("void ") else k
        k = k.split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[(k, f)].add(r["Dispatch_Id"])
nl = collections.defaultdict(int)
for (k, f), ids in launches.items():
    nl[k] = max(nl[k], len(ids))
cols = sorted({c for k in acc for c in acc[k]})
with open("%s/%s_bench_pmc_summary.csv" % (out, tag), "w") as fo:
    w = csv.writer(fo)
    w.writerow(["kernel", "launches_per_encode_plus_decode"] + [c + "_per_launch" for c in cols])
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
        w.writerow([k, nl[k]] + [round(acc[k].get(c, 0) / max(nl[k], 1), 1) for c in cols])


def per_launch(k, c):
    return acc[k].get(c, 0) / max(nl[k], 1)


def hbm(k):
    # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 tallies 128-B read requests as 64 B -> fetch doubled (MI355X_MICROARCH.md)
    return int((2 * per_launch(k, "FETCH_SIZE") + per_launch(k, "WRITE_SIZE")) * 1024)


ks = sorted(k for k in acc if k.startswith("k_"))
short = lambda k: k.split("<")[0]
tr = {"round": tag,
      "command": "rocprofv3 --kernel-trace --pmc <one pass per counter set> -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-configs --no-file-to-file --streams 1",
      "correction": "gfx950 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM): fetch doubled; WRITE_SIZE as reported",
      "hbm_bytes_per_launch": {short(k): hbm(k) for k in ks},
      "launches": {short(k): nl[k] for k in ks},
      "valu_wave_insts_per_launch": {short(k): int(per_launch(k, "SQ_INSTS_VALU")) for k in ks},
      "salu_wave_insts_per_launch": {short(k): int(per_launch(k, "SQ_INSTS_SALU")) for k in ks},
      "lds_wave_insts_per_launch": {short(k): int(per_launch(k, "SQ_INSTS_LDS")) for k in ks},
      "wait_any_over_wave_cycles": {short(k): round(per_launch(k, "SQ_WAIT_ANY") / per_launch(k, "SQ_WAVE_CYCLES"), 3) for k in ks if per_launch(k, "SQ_WAVE_CYCLES")},
      "lds_conflict_over_active": {short(k): round(per_launch(k, "SQ_LDS_BANK_CONFLICT") / per_launch(k, "SQ_LDS_IDX_ACTIVE"), 3) for k in ks if per_launch(k, "SQ_LDS_IDX_ACTIVE")}}
enc = [k for k in ks if not short(k).startswith(("k_dec", "k_scan"))]
dec = [k for k in ks if short(k).startswith(("k_dec", "k_scan"))]
# the device sources these counters were taken on: bench.py quotes them only while csrc/*.hip hash to the same value
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
tr["kernel_source_hash"] = bench.kernel_source_hash()
tr["encode_hbm_bytes"] = sum(hbm(k) * nl[k] for k in enc)
tr["decode_hbm_bytes"] = sum(hbm(k) * nl[k] for k in dec)
json.dump(tr, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps({k: tr[k] for k in ("encode_hbm_bytes", "decode_hbm_bytes")}))
