#!/bin/bash
# GPU-box tool: SQ counter passes of the default bench command for several library builds in ONE call
#   bash tools/gpu_pmc_compare.sh <out-dir> <lib.so> [<lib.so> ...]
# -> <out-dir>/<lib basename>_pmc_sq.csv (per-working-launch means, tools/profile_summarise.py pmc)
set -o pipefail
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for lib in "$@"; do
    name=$(basename "$lib" .so)
    export S2D_LIBRARY="$PWD/$lib"
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d "$out/$name.sq1" -- python3 bench.py --steps 24 --warmup 10 --no-cpu-baseline > "$out/$name.sq1.log" 2>&1 || { echo "$name sq1 FAILED"; tail -5 "$out/$name.sq1.log"; exit 1; }
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$out/$name.sq2" -- python3 bench.py --steps 24 --warmup 10 --no-cpu-baseline > "$out/$name.sq2.log" 2>&1 || { echo "$name sq2 FAILED"; tail -5 "$out/$name.sq2.log"; exit 1; }
    python3 tools/profile_summarise.py pmc "$out/$name.sq1" "$out/$name.sq2" "$out/${name}_pmc_sq.csv"
    head -5 "$out/${name}_pmc_sq.csv"
    rm -rf "$out/$name.sq1" "$out/$name.sq2"
done
