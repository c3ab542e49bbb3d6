#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small summaries kept under profiles/.

  kernel stats : profile_summarise.py stats  <dir> <out.csv>      (copies the *_kernel_stats.csv of --kernel-trace --stats)
  counters     : profile_summarise.py pmc    <dir> [<dir> ...] <out.csv>
                 per kernel name: launches, working launches and the per-WORKING-launch mean of every counter found in
                 *_counter_collection.csv (void optimistic launches -- counter < 5 % of the kernel's median -- left out)
  working      : profile_summarise.py working <dir> <out.csv>
                 per kernel, from the kernel TRACE: all launches, and the launches that did work -- the first raster kernel of
                 an iteration is launched optimistically and returns at once (~22 us) when the tile lists turn out stale
                 (csrc/s2d_api.hip queue_raster), so the stats file's AverageNs mixes in one void launch per list rebuild
  traffic      : profile_summarise.py traffic <pmc_traffic.csv> [<pmc_sq.csv>] <traffic.json>
                 HBM bytes per launch of the dominant raster kernel for bench.py's roofline.traffic, corrected as
                 /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950, and tied to the kernel sources
"""
import csv, glob, os, sys
from collections import defaultdict


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, d))
    return hits[0]


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    mode = sys.argv[1]
    if mode == "stats":
        src = find(sys.argv[2], "kernel_stats.csv")
        with open(src) as f, open(sys.argv[3], "w") as g:
            g.write(f.read())
        print("wrote", sys.argv[3])
    elif mode == "pmc":
        import statistics
        dirs, out = sys.argv[2:-1], sys.argv[-1]
        # A launch is 'void' when some counter of its pass reads under 5 % of the kernel's median for that counter: the first
        # raster kernel of an iteration is launched optimistically and returns at once when the tile lists turn out stale
        # (one per list rebuild).  Dispatch ids belong to one pass (one process), so launches are sorted pass by pass.
        # Means are over the launches that WORK; the all-launch count is kept beside them.
        mean, nall, nwork = defaultdict(dict), defaultdict(int), defaultdict(int)
        for d in dirs:
            val = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))  # kernel -> counter -> dispatch -> value
            with open(find(d, "counter_collection.csv")) as f:
                for row in csv.DictReader(f):  # a counter's rows of one dispatch (e.g. one per XCD) are added up
                    val[short(row["Kernel_Name"])][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
            for k in val:
                ids = set().union(*[set(val[k][c]) for c in val[k]])
                void = set()
                for c in val[k]:
                    med = statistics.median(val[k][c].values())
                    if med > 0:
                        void |= {i for i, x in val[k][c].items() if x < 0.05 * med}
                work = ids - void
                for c in val[k]:
                    v = [val[k][c][i] for i in work if i in val[k][c]]
                    mean[k][c] = sum(v) / max(len(v), 1)
                nall[k] = max(nall[k], len(ids))
                nwork[k] = max(nwork[k], len(work))
        counters = sorted({c for k in mean for c in mean[k]})
        val = mean
        with open(out, "w") as g:
            g.write("# per-launch means over the launches that did work (a launch whose counter reads < 5 %% of the kernel's median "
                    "is void and left out); rocprofv3 --pmc, one pass per directory: %s\n" % " ".join(os.path.basename(os.path.normpath(d)) for d in dirs))
            g.write("kernel,launches,working_launches," + ",".join(counters) + "\n")
            for k in sorted(val, key=lambda k: -sum(mean[k].values()) * nwork[k]):
                g.write("%s,%d,%d,%s\n" % (k.replace(",", ";"), nall[k], nwork[k],
                                           ",".join("%.6g" % mean[k][c] if c in mean[k] else "" for c in counters)))
        print("wrote", out)
    elif mode == "working":
        import statistics
        per = defaultdict(list)
        with open(find(sys.argv[2], "kernel_trace.csv")) as f:
            for row in csv.DictReader(f):
                per[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        with open(sys.argv[3], "w") as g:
            g.write("# from the kernel trace; a launch is 'void' when shorter than 5 % of the kernel's median duration\n")
            g.write("kernel,launches,average_ns_all,void_launches,working_launches,average_ns_working,median_ns_working\n")
            for k in sorted(per, key=lambda k: -sum(per[k])):
                d = per[k]
                med = statistics.median(d)
                work = [x for x in d if x >= 0.05 * med]
                g.write("%s,%d,%.0f,%d,%d,%.0f,%.0f\n" % (k.replace(",", ";"), len(d), sum(d) / len(d), len(d) - len(work), len(work),
                                                       sum(work) / len(work), statistics.median(work)))
        print("wrote", sys.argv[3])
    elif mode == "traffic":
        import json
        rows = [r for r in csv.reader(l for l in open(sys.argv[2]) if not l.startswith("#"))]
        hdr, rows = rows[0], rows[1:]
        ifetch, iwrite = hdr.index("FETCH_SIZE"), hdr.index("WRITE_SIZE")
        # the dominant kernel: the fused raster instantiation with the most launches (bench.py's timed block: all nine gradients)
        cand = [r for r in rows if "raster_fused_kernel<" in r[0]] or [r for r in rows if "raster_" in r[0]]
        r = max(cand, key=lambda r: int(r[1]))
        sq = {}
        if len(sys.argv) > 4:   # the SQ / GRBM passes of the same kernel
            srows = [x for x in csv.reader(l for l in open(sys.argv[3]) if not l.startswith("#"))]
            shdr = srows[0]
            hit = [x for x in srows[1:] if x[0] == r[0]]
            if hit:
                v = {c: float(hit[0][k]) for k, c in enumerate(shdr) if k >= 3 and hit[0][k] != ""}
                cycles = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0          # the counter is read per XCD and summed: 8 XCDs
                sq = {"sq_source": "%s (rocprofv3 --pmc, two SQ/GRBM passes)" % os.path.basename(sys.argv[3]),
                      "SQ_INSTS_VALU_per_launch": v.get("SQ_INSTS_VALU"), "SQ_INSTS_SALU_per_launch": v.get("SQ_INSTS_SALU"),
                      "SQ_INSTS_LDS_per_launch": v.get("SQ_INSTS_LDS"), "SQ_LDS_BANK_CONFLICT_per_launch": v.get("SQ_LDS_BANK_CONFLICT"),
                      "gpu_cycles_per_launch": cycles,
                      # 256 CUs x 4 SIMDs issue at most one VALU instruction per cycle each
                      "simd_cycles_per_valu_instruction": (1024.0 * cycles / v["SQ_INSTS_VALU"]) if v.get("SQ_INSTS_VALU") else None}
        fetch_kb, write_kb = float(r[ifetch]), float(r[iwrite])
        import importlib
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        digest = importlib.import_module("2dgaussiansplatting_amd._build").kernel_source_digest()
        out = {
            "source": "%s (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes; tools/gpu_profile_round.sh)" % os.path.basename(sys.argv[2]),
            "workload": "4096x4096 synthetic, 1,000,000 Gaussians, 1 GPU (python3 bench.py)",
            "kernel": r[0], "launches": int(r[1]), "working_launches": int(r[hdr.index("working_launches")]),
            "note": "per-launch means over the launches that did work (void optimistic launches left out)",
            "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
            "correction": "gfx950: FETCH_SIZE tallies a 128-B request of a wide coalesced read as 64 B, WRITE_SIZE is exact for 16-B "
                          "streaming stores and float atomics: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section); "
                          "this kernel also makes 8-B mask accesses and 64-B record gathers, widths the guide calls uncalibrated, so the "
                          "figure is an upper estimate",
            "dominant_kernel_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
            "kernel_source_digest": digest,
        }
        out.update(sq)
        json.dump(out, open(sys.argv[-1], "w"), indent=1)
        print("wrote", sys.argv[-1])
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
