"""The host protocol of the multi-device handle (2dgaussiansplatting_amd/csrc/s2d_multi.hip: worker threads, the breakable
barrier, pairwise sequence counters, stream event waits, two send buffers, the stop / abort path) under ThreadSanitizer,
without a GPU.

s2d_multi.hip has no kernels: it is compiled here as plain C++ against tests/hostsim -- a thread-backed stand-in for the
handful of HIP / RCCL calls it makes (a stream = an in-order queue with its own thread, an event = a sequence number,
hipStreamWaitEvent = a wait for the record that was the latest when it was called) -- with the single-device contexts
simulated on the CPU oracle.  tests/hostsim/multi_sim_main.cpp then runs 2 / 4 / 8 rank threads through 230 iterations with
three hold-set refreshes, a get_splats / set_splats in the middle and ranks slowed down at random; slab ownership and
replicated state (through the simulated RCCL) must agree bit for bit; a rank that stops answering must give S2D_E_STATE
naming it (both schemes); a rank whose launch or all-reduce submission fails must be reported with ITS error, at once, and a
step right after it refused; a rank inside a runtime call that never returns must be given up on by the caller's watchdog
(the handle abandoned, not freed); a non-finite stop must be reported and survived.  ThreadSanitizer must stay silent throughout.

Test infrastructure only: nothing here is a CPU path of the product.
"""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SIM = os.path.join(HERE, "hostsim")
MULTI = os.path.join(os.path.dirname(HERE), "2dgaussiansplatting_amd", "csrc", "s2d_multi.hip")
# ROCm's clang carries a ThreadSanitizer runtime that knows pthread_cond_clockwait (what libstdc++'s
# condition_variable::wait_for calls); gcc 11's does not and reports every timed wait as a double lock
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _tsan_compiler():
    if os.path.exists(CLANG):
        return CLANG
    pytest.skip("no ThreadSanitizer-capable compiler with a pthread_cond_clockwait interceptor on this machine")


def _run(exe, *args):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    return subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=900, env=env)


@pytest.fixture(scope="module")
def sim():
    cxx = _tsan_compiler()
    subprocess.check_call(["make", "-C", SIM, "CXX=" + cxx], stdout=subprocess.DEVNULL)
    return os.path.join(SIM, "_build", "multi_sim")


def test_multi_device_host_protocol_is_race_free_and_fails_informatively(sim):
    p = _run(sim)
    report = p.stdout + "\n" + p.stderr[-6000:]
    assert "WARNING: ThreadSanitizer" not in p.stderr, report
    assert p.returncode == 0, report
    lines = p.stdout.splitlines()
    assert not [ln for ln in lines if ln.startswith("NOT ok")], report
    for world in (2, 4, 8):
        assert any(ln.startswith("ok: %d ranks, 230 iterations, ownership == replicated bit for bit" % world) for ln in lines), report
    assert any(ln.startswith("ok: 3 ranks sharing one device") for ln in lines), report       # the one-GPU rehearsal mode
    assert sum("stops answering" in ln and "S2D_E_STATE" in ln for ln in lines) == 5, report
    assert sum(ln.startswith("ok:") and "inside a call that never returns -> abandoned" in ln for ln in lines) == 2, report
    assert sum(ln.startswith("ok:") and "fails in its" in ln for ln in lines) == 3, report   # a rank's own error, not the others' timeouts
    assert sum("non-finite stop reported and survived" in ln for ln in lines) == 2, report
    # the simulated runtime really was exercised: peer copies, event waits
    tail = [ln for ln in lines if ln.startswith("simulated runtime:")][0].split()
    assert int(tail[2]) > 50000 and int(tail[5]) > 10000 and int(tail[11]) > 10000, tail


def test_the_harness_sees_a_seeded_race(sim, tmp_path):
    """Sensitivity check: with ONE send buffer instead of two (a rank gathers iteration k + 1 while a neighbour may still be
    copying iteration k) ThreadSanitizer must report the race between the gather and the peer copy."""
    src = open(MULTI).read()
    needle = "const int b = (int)((seq - 1u) & 1u);"
    assert src.count(needle) == 1
    src = src.replace(needle, "const int b = 0;")
    root = os.path.dirname(HERE)
    src = src.replace('#include "../../include/', '#include "%s/include/' % root)
    mut = tmp_path / "s2d_multi_mutant.hip"
    mut.write_text(src)
    exe = str(tmp_path / "multi_sim_mutant")
    build = os.path.join(SIM, "_build")
    subprocess.check_call([_tsan_compiler(), "-O1", "-g", "-std=c++17", "-fPIC", "-pthread", "-fsanitize=thread", "-I" + os.path.join(SIM, "include"),
                           "-o", exe, os.path.join(SIM, "multi_sim_main.cpp"), os.path.join(SIM, "sim_ctx.cpp"), "-x", "c++", str(mut), "-x", "none",
                           os.path.join(build, "s2d_oracle.o"), "-L" + build, "-lsimhip", "-Wl,--no-as-needed", "-l:librccl.so.1", "-Wl,--as-needed",
                           "-ldl", "-lm", "-Wl,--disable-new-dtags", "-Wl,-rpath," + build])
    p = _run(exe, "quick")
    assert "WARNING: ThreadSanitizer: data race" in p.stderr and "s2d_rows_gather" in p.stderr, p.stdout + p.stderr[-3000:]


def test_multi_device_host_protocol_under_address_and_ub_sanitizers():
    """The same scenarios (4 ranks for the long comparison, every failure path) built with -fsanitize=address,undefined: no
    use-after-free on the stop / abort / abandon / destroy paths, no undefined behaviour.  (Leak detection off: an abandoned
    handle and the simulated communicators are leaked on purpose.)"""
    cxx = _tsan_compiler()
    out = "_build_asan"
    subprocess.check_call(["make", "-C", SIM, "CXX=" + cxx, "SAN=-fsanitize=address,undefined -fno-omit-frame-pointer", "OUT=" + out],
                          stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([os.path.join(SIM, out, "multi_sim"), "quick"], capture_output=True, text=True, timeout=900, env=env)
    report = p.stdout + "\n" + p.stderr[-6000:]
    assert p.returncode == 0 and "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, report
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("NOT ok")], report
