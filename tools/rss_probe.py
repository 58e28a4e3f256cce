import os, sys, subprocess, resource
sys.path.insert(0, os.getcwd())
import lz77_amd as L
from lz77_amd import synth
d = "/dev/shm/rssprobe"; os.makedirs(d, exist_ok=True)
fin, fout = d + "/in", d + "/out"
synth.make("text", 1_000_000_000, synth.SEED_S4).tofile(fin)
def run(env):
    p = subprocess.Popen([L.CLI_PATH, "-c", "-i", fin, "-o", fout], env=dict(os.environ, **env))
    _, st, ru = os.wait4(p.pid, 0)
    print(env, "status", st, "maxrss_MB", ru.ru_maxrss // 1024, "out", os.path.getsize(fout), flush=True)
run({})
run({"LZ77X_SHARDS": "4", "LZ77X_FAKE_DEVICES": "4"})
run({"LZ77X_SHARDS": "4", "LZ77X_FAKE_DEVICES": "4", "LZ77X_SHARD_STRETCH": str(256 << 20)})
run({"LZ77X_SHARDS": "2", "LZ77X_FAKE_DEVICES": "2", "LZ77X_SHARD_STRETCH": str(256 << 20)})
run({"LZ77X_SHARDS": "4", "LZ77X_FAKE_DEVICES": "4", "LZ77X_SHARD_STRETCH": str(256 << 20), "LZ77X_TRACE": "1"})
import shutil; shutil.rmtree(d)
