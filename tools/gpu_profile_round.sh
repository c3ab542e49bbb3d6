#!/bin/bash
# GPU-box tool: the rocprofv3 passes behind profiles/rNN (run through gpurun from the repo root):
#   bash tools/gpu_profile_round.sh r03
# 1. --kernel-trace --stats of the default bench command            -> <tag>_kernel_stats.csv
# 2. --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes          -> <tag>_pmc_traffic.csv   (per-launch means)
# 3. two SQ / GRBM counter passes                                    -> <tag>_pmc_sq.csv
# and profiles/traffic.json for bench.py's roofline.traffic (tied to the kernel sources by their digest).
# rocprofv3 is given `python3 bench.py ...` directly (no shell / env hop); counters are collected with --kernel-trace only.
set -o pipefail
tag=${1:-r04}
out=gpurun_out/profile_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
run() { # name, rocprof args...
    local name=$1; shift
    timeout -k 10 420 rocprofv3 "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps "$STEPS" --warmup 10 --no-cpu-baseline > "$out/$name.log" 2>&1 || { echo "$name FAILED"; tail -5 "$out/$name.log"; return 1; }
    echo "$name ok"
}
STEPS=200 run stats --kernel-trace --stats || exit 1
STEPS=24 run fetch --kernel-trace --pmc FETCH_SIZE || exit 1
STEPS=24 run write --kernel-trace --pmc WRITE_SIZE || exit 1
STEPS=24 run sq1 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE || exit 1
STEPS=24 run sq2 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES || exit 1
python3 tools/profile_summarise.py stats "$out/stats" "$out/${tag}_kernel_stats.csv"
python3 tools/profile_summarise.py working "$out/stats" "$out/${tag}_kernel_working_launches.csv"
python3 tools/profile_summarise.py pmc "$out/fetch" "$out/write" "$out/${tag}_pmc_traffic.csv"
python3 tools/profile_summarise.py pmc "$out/sq1" "$out/sq2" "$out/${tag}_pmc_sq.csv"
python3 tools/profile_summarise.py traffic "$out/${tag}_pmc_traffic.csv" "$out/${tag}_pmc_sq.csv" "$out/traffic.json"
grep -h '"metric"' "$out/stats.log" | tail -1 > "$out/${tag}_bench_under_rocprof.json"
head -4 "$out/${tag}_kernel_stats.csv" | cut -c1-60,250-; cat "$out/traffic.json"
