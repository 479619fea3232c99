#!/bin/bash
# rocprofv3 passes over the ten-million-triangle benchmark configuration (run on the GPU box via gpurun).
# usage: tools/profile_config5.sh <tag>
set -u
TAG=${1:-c5}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
ARGS="--config 5 --steps 1 --warmup 1 --cpu-spp 0 --no-roofline-counters"
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprof options...
  local name=$1; shift
  timeout -s KILL 420 rocprofv3 "$@" -d "$OUT/$name" -o "$name" --output-format csv -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
run stats --kernel-trace --stats || exit 1
run sq1 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY || exit 1
run l2 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE || exit 1
run fetch --kernel-trace --pmc FETCH_SIZE || exit 1
run write --kernel-trace --pmc WRITE_SIZE || exit 1
{
  echo "# rocprofv3 summary of the trace kernel on the ten-million-triangle scene, MI355X; counters are sums over the launches of one pass (count in brackets)"
  echo "# command: tools/profile_config5.sh $TAG  ->  rocprofv3 <pass> -- python3 bench.py $ARGS"
  echo "# kernel_source_sha1: $(cd "$ROOT" && python3 -c 'import bench; print(bench.kernel_source_digest())' 2>/dev/null)"
  python3 "$ROOT/tools/summarize_prof.py" "$OUT"
} > "$ROOT/gpurun_out/prof_${TAG}_summary.txt"
echo "summary: gpurun_out/prof_${TAG}_summary.txt"
