#!/bin/bash
# rocprofv3 passes over the headline benchmark (run on the GPU box via gpurun).
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
ARGS="--steps 2 --warmup 1 --cpu-spp 0 --no-roofline-counters --no-secondary $*"
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprof options...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" -d "$OUT/$name" -o "$name" --output-format csv -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
run stats --kernel-trace --stats || exit 1
run sq1 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY || exit 1
run sq2 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR || exit 1
run sq3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 || exit 1
run l2 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE || exit 1
run fetch --kernel-trace --pmc FETCH_SIZE || exit 1
run write --kernel-trace --pmc WRITE_SIZE || exit 1
{
  echo "# rocprofv3 summary of the trace kernel, MI355X; counters are sums over the launches of one pass (count in brackets)"
  echo "# command: tools/profile_bench.sh $TAG $*  ->  rocprofv3 <pass> -- python3 bench.py $ARGS"
  echo "# kernel_source_sha1: $(cd "$ROOT" && python3 -c 'import bench; print(bench.kernel_source_digest())' 2>/dev/null)"
  python3 "$ROOT/tools/summarize_prof.py" "$OUT"
} > "$ROOT/gpurun_out/prof_${TAG}_summary.txt"
echo "summary: gpurun_out/prof_${TAG}_summary.txt"
