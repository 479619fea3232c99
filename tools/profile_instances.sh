#!/bin/bash
# rocprofv3 passes over the many-node scene (tools/instance_probe.py: scan by groups, node-by-node scan, BVH over the nodes).
# usage: tools/profile_instances.sh <tag>      (run on the GPU box via gpurun)
set -u
TAG=${1:-inst}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprof options...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" -d "$OUT/$name" -o "$name" --output-format csv -- python3 "$ROOT/tools/instance_probe.py" 40 16 > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
run stats --kernel-trace --stats || exit 1
run sq1 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY || exit 1
run l2 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE || exit 1
{
  echo "# rocprofv3 summary of the trace kernels on the 1602-node instance grid (1280x720, 16 spp; two launches per mode), MI355X"
  echo "# trace_kernel<..., true, false, true, false>: bit-exact scan by groups of nodes (run twice: with and without the candidate mask); <..., true, true>: BVH over the nodes (opt-in, not bit-exact)"
  echo "# command: tools/profile_instances.sh $TAG  ->  rocprofv3 <pass> -- python3 tools/instance_probe.py 40 16"
  python3 "$ROOT/tools/summarize_prof.py" "$OUT"
} > "$ROOT/gpurun_out/prof_${TAG}_summary.txt"
echo "summary: gpurun_out/prof_${TAG}_summary.txt"
