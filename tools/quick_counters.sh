#!/bin/bash
# One rocprofv3 counter pass (vector instruction totals) over a short run of the benchmark; run on the GPU box.
# usage: tools/quick_counters.sh <tag> [bench args...]
set -u
TAG=${1:-q}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/quick_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d "$OUT/sq1" -o sq1 --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --cpu-spp 0 --no-roofline-counters "$@" > "$OUT/sq1.log" 2>&1 || exit 1
python3 "$ROOT/tools/summarize_prof.py" "$OUT" | grep trace_kernel
