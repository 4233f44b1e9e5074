"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer: `make -C oracle asan` builds the same C source
with -fsanitize=address,undefined, and the oracle's own tests (known-answer tests, the numpy / scipy re-derivations,
the golden fixtures) run against that build in a child Python process with the ASan runtime preloaded.  Any report
(heap overflow, misaligned access, signed overflow, out-of-range shift ...) fails the run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_tests_pass_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    lib = os.path.join(ROOT, "oracle", "liborb_oracle_asan.so")
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ)
    env.update({"ORB_ORACLE_LIB": lib, "LD_PRELOAD": asan_rt + (":" + env["LD_PRELOAD"] if env.get("LD_PRELOAD") else ""),
                # the interpreter itself is not instrumented: its allocations are not this test's business
                "ASAN_OPTIONS": "detect_leaks=0:verify_asan_link_order=0:abort_on_error=0:exitcode=97",
                "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1:exitcode=98"})
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle.py"),
                        os.path.join(ROOT, "tests", "test_golden.py")], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    assert " passed" in r.stdout
