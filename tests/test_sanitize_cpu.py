"""Host side under AddressSanitizer + UndefinedBehaviorSanitizer ("host C can be built with
-fsanitize=address,undefined in tests"; GPU sanitizers do not exist on this pool).

`make -C lz77_amd/csrc asan` compiles every host translation unit of the product -- api.cpp, ctx.cpp, hostio.cpp, the pipelines and shard.cpp (the C ABI, contexts,
memory planning, the shard host side), hoststage.c, fileio.c, main.c, shim.c -- with the ROCm clang's sanitizers and
links them with the ordinary kernel objects.  The CPU ABI suite (tests/test_abi_cpu.py: host-only entry points against
the oracle, the no-device error paths of every entry point, the CLI's argument handling, the shard plan and the
world-size-2 gloo exchange) then runs against THAT library and CLI in a child interpreter with the runtime preloaded;
any report fails the test (halt_on_error)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lz77_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang"


def _runtime():
    if not os.path.exists(CLANG):
        return None
    p = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.fixture(scope="module")
def asan_env():
    rt = _runtime()
    if rt is None:
        pytest.skip("no clang AddressSanitizer runtime in this image")
    r = subprocess.run(["make", "-s", "-j8", "-C", CSRC, "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ)
    env.update({"LD_PRELOAD": rt,
                "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=66",
                "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1:exitcode=67",
                "LZ77X_TEST_LIB": os.path.join(ROOT, "lz77_amd", "liblz77_mi355x_asan.so"),
                "LZ77X_TEST_CLI": os.path.join(ROOT, "lz77_amd", "lz77_asan")})
    return env


def test_sanitized_library_is_what_the_child_loads(asan_env):
    code = ("import lz77_amd as L; L.lib(); m = open('/proc/self/maps').read(); "
            "assert 'liblz77_mi355x_asan.so' in m and 'libclang_rt.asan' in m; assert L.CLI_PATH.endswith('lz77_asan'); print('ok')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=asan_env, cwd=ROOT, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_abi_suite_under_asan_ubsan(asan_env):
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_abi_cpu.py"), "-q", "-x", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=asan_env, cwd=ROOT, timeout=1500)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert " passed" in out


def test_sanitizer_catches_a_planted_bug(asan_env, tmp_path):
    """the harness is live: a deliberate out-of-bounds read through the same runtime is reported"""
    src = tmp_path / "oob.c"
    src.write_text("#include <stdlib.h>\nint main(int c, char **v) { char *p = malloc(8); int r = p[8 + c]; free(p); return r & 0; }\n")
    exe = tmp_path / "oob"
    subprocess.check_call([CLANG, "-fsanitize=address", "-shared-libsan", "-g", str(src), "-o", str(exe)])
    env = dict(asan_env)
    env.pop("LD_PRELOAD")
    env["LD_LIBRARY_PATH"] = os.path.dirname(asan_env["LD_PRELOAD"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert r.returncode in (66, 67) and "heap-buffer-overflow" in r.stderr
