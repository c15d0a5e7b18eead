#!/bin/bash
# tools/refresh_profiles.sh <round tag, e.g. r01> : on the GPU box, from the repo root.
# Writes small summaries under gpurun_out/profiles_<tag>/ (copy them into profiles/ afterwards).
set -u
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/profiles_$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/${tag}_bench_default.json" 2> "$out/bench.err"
cd /tmp
args="--steps 20 --warmup 3 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 "$root/bench.py" $args > /dev/null 2>&1
python3 "$root/tools/prof_summarize.py" /tmp/prof_stats "$out/${tag}_bench" > /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- python3 "$root/bench.py" $args > /dev/null 2>&1
python3 "$root/tools/prof_summarize.py" /tmp/prof_fetch "$out/${tag}_fetch" > /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- python3 "$root/bench.py" $args > /dev/null 2>&1
python3 "$root/tools/prof_summarize.py" /tmp/prof_write "$out/${tag}_write" > /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/prof_mfma -- python3 "$root/bench.py" $args > /dev/null 2>&1
python3 "$root/tools/prof_summarize.py" /tmp/prof_mfma "$out/${tag}_mfma" > /dev/null
ls -la "$out"
