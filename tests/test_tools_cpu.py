"""CPU tests of the evidence tooling: what profiles/ is made with must treat the optimistic void launches correctly."""
import csv
import os
import subprocess
import sys

import oracle_lib as O

TOOL = os.path.join(O.ROOT, "tools", "profile_summarise.py")


def _write(path, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writerows(rows)


def test_pmc_summaries_average_over_working_launches_only(tmp_path):
    """The first raster kernel of an iteration is launched optimistically and returns at once when the tile lists turn out
    stale: such a launch reads a few per cent of a working launch's counters (or zero traffic) and must not be averaged in.
    Counters whose value does not depend on the work (SQ_WAVES) follow the verdict of the others of their pass; passes are
    separate processes, so launches are told apart pass by pass."""
    k = "void s2d::raster_fused_kernel<false, false>(int)"
    a, b = str(tmp_path / "sq" / "x" / "1_counter_collection.csv"), str(tmp_path / "tr" / "x" / "2_counter_collection.csv")
    rows = []
    for d in range(10):
        void = d in (3, 7)
        for xcd in range(2):   # one row per XCD: added up per dispatch
            rows.append([d, k, "SQ_INSTS_VALU", 10 if void else 500])
            rows.append([d, k, "SQ_WAVES", 100])
    _write(a, rows)
    _write(b, [[100 + d, k, "FETCH_SIZE", 0 if d == 5 else 4000] for d in range(8)])
    out = str(tmp_path / "o.csv")
    subprocess.check_call([sys.executable, TOOL, "pmc", os.path.dirname(os.path.dirname(a)), os.path.dirname(os.path.dirname(b)), out])
    lines = [l for l in open(out).read().splitlines() if not l.startswith("#")]
    hdr, row = lines[0].split(","), lines[1].split(",")
    got = dict(zip(hdr, row))
    assert got["launches"] == "10" and got["working_launches"] == "8"
    assert float(got["SQ_INSTS_VALU"]) == 1000.0      # 2 XCD rows x 500, the two void launches left out
    assert float(got["SQ_WAVES"]) == 200.0
    assert float(got["FETCH_SIZE"]) == 4000.0         # 7 working launches of the other pass


def test_working_launch_durations(tmp_path):
    p = str(tmp_path / "t" / "x" / "1_kernel_trace.csv")
    os.makedirs(os.path.dirname(p))
    with open(p, "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        t = 0
        for d in range(12):
            dur = 20 if d % 4 == 0 else 2000
            w.writerow(["void s2d::raster_fused_kernel<false>(int)", t, t + dur])
            t += dur + 5
    out = str(tmp_path / "w.csv")
    subprocess.check_call([sys.executable, TOOL, "working", str(tmp_path / "t"), out])
    row = [l for l in open(out).read().splitlines() if not l.startswith("#")][1].split(",")
    assert row[1:5] == ["12", "1505", "3", "9"] and float(row[5]) == 2000.0


def test_one_residual_correction_gives_the_ieee_quotient():
    """csrc/s2d_raster.hip::div_by_recip: q = S*r; q += (S - den*q)*r with r a 1-ulp reciprocal.  What its comment claims, on
    three operand populations of the backward blend's range incl. den = 1e-15 and quotients placed next to rounding midpoints
    (tools/check_recip_division.py, numpy emulation of the fma; the full 3 x 10^8-trial run is profiles/r04/r04_recip_division_check.txt):
    the bare estimate S*r misses the IEEE quotient often, one correction leaves at most a ~1e-7 sliver, and with a correctly
    rounded reciprocal nothing at all."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_recip_division", os.path.join(O.ROOT, "tools", "check_recip_division.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tot = mod.run(total=400_000, procs=1)
    assert set(tot) == {"uniform", "log", "midpoint"}
    for kind, t in tot.items():
        assert t["n"] == 400_000
        for name in mod.NAMES:
            c0, c1, c2, changed_by_second, c3 = t[name]
            assert c0 > 40_000, (kind, name, c0)          # the bare estimate misses the IEEE quotient in > 10 % of the cases
            assert c1 <= 2 and c2 <= c1 + 1 and c3 <= c1 + 1, (kind, name, t[name])
        assert t["r = RN(1/den)"][1] == 0, (kind, t)       # correctly rounded reciprocal: one correction is IEEE division
