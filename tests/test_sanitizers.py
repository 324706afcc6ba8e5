"""Sanitizer runs of the HOST logic, all on the CPU (-m "not gpu"):

* ThreadSanitizer and AddressSanitizer + UBSan builds of csrc/api.hip, dist.hip and multi.hip (host side only:
  `make -C gp_ss_ak_amd/csrc tsan asan`) loaded by tests/san_worker.py, which drives gpak_create_multi's thread-per-rank
  group -- worker threads, in-process rendezvous transport, per-rank error slots, start-up self-check, distributed
  logLikelihood / alpha / gradient, Chol_fail, teardown -- over NumPy engines;
* an AddressSanitizer + UBSan build of the host C++ classes (host/host_selftest_asan): data reader, standardisation,
  model files, and Opt_Algs::LBFGSOptimise on its analytic objectives.
"""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gp_ss_ak_amd", "csrc")
HOST = os.path.join(ROOT, "gp_ss_ak_amd", "host")


def _runtime(name):
    c = sorted(glob.glob(f"/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.{name}-x86_64.so"))
    if not c:
        pytest.skip(f"no clang {name} runtime in this image")
    return c[-1]


def _worker(lib, preload, ranks, n, extra_env):
    env = dict(os.environ, LD_PRELOAD=preload, GPAK_LIB_PATH=os.path.join(CSRC, "san", lib), OPENBLAS_NUM_THREADS="1",
               OMP_NUM_THREADS="1", GPAK_ORACLE_THREADS="1", **extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "san_worker.py"), str(ranks), str(n)], env=env,
                          cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)


@pytest.mark.parametrize("ranks,n", [(2, 400), (4, 700)])
def test_thread_per_gpu_host_logic_under_thread_sanitizer(ranks, n):
    subprocess.check_call(["make", "-s", "-j3", "-C", CSRC, "tsan"])
    out = _worker("libgpak_hip_tsan.so", _runtime("tsan"), ranks, n,
                  {"TSAN_OPTIONS": f"report_signal_unsafe=0 exitcode=66 history_size=4 suppressions={ROOT}/tests/tsan.supp"})
    text = out.stdout.decode(errors="replace")
    assert "WARNING: ThreadSanitizer" not in text, text[-6000:]
    assert out.returncode == 0 and f"san_worker ok: {ranks} ranks" in text, text[-3000:]


def test_thread_per_gpu_host_logic_under_address_and_ub_sanitizers():
    subprocess.check_call(["make", "-s", "-j3", "-C", CSRC, "asan"])
    out = _worker("libgpak_hip_asan.so", _runtime("asan"), 3, 500,
                  {"ASAN_OPTIONS": "detect_leaks=0:exitcode=67", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"})
    text = out.stdout.decode(errors="replace")
    assert "AddressSanitizer" not in text and "runtime error" not in text, text[-6000:]
    assert out.returncode == 0 and "san_worker ok: 3 ranks" in text, text[-3000:]


def test_host_classes_under_address_and_ub_sanitizers(tmp_path):
    from gp_ss_ak_amd import synth
    subprocess.check_call(["make", "-s", "-C", HOST, "host_selftest_asan"])
    exe = os.path.join(HOST, "host_selftest_asan")
    Xr, yr = synth.drillholes_raw(200)
    f = tmp_path / "train.txt"
    with open(f, "w") as fh:
        fh.write("# x, y, z, grade\n")
        for r, v in zip(Xr, yr):
            fh.write("\t".join(f"{t:.17g}" for t in list(r) + [v]) + "\n")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:exitcode=67", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    runs = [[exe, "--logic", str(f)]] + [[exe, "--opt", str(v), "0"] for v in (0, 1, 2)]
    for cmd in runs:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        text = out.stdout.decode(errors="replace")
        assert out.returncode == 0 and "AddressSanitizer" not in text and "runtime error" not in text, (cmd, text[-4000:])
