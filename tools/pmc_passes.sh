#!/bin/bash
# Counter passes over one stage of the path (tools/prof_stage.py), one rocprofv3 run per pass
# (rocprofv3 does not split counters over passes; see the slot limits in MI355X_MICROARCH.md).
# usage (on the GPU box): bash tools/pmc_passes.sh <stage> <B> <tag> [passes...]
#   passes: sq1 sq2 sq3 fetch write lds  (default: all but sq3)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
STAGE=${1:-slic}; B=${2:-64}; TAG=${3:-pmc}; shift 3 || true
PASSES=${*:-sq1 sq2 fetch write lds}
cd /tmp && export TMPDIR=/tmp
for p in $PASSES; do
  case $p in
    sq1) C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY";;
    sq2) C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR";;
    sq3) C="SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE";;
    lds) C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_LDS SQ_WAVES";;
    fetch) C="FETCH_SIZE";;
    write) C="WRITE_SIZE";;
    *) echo "unknown pass $p"; exit 1;;
  esac
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/${TAG}_$p -- python3 $R/tools/prof_stage.py $STAGE $B > $R/gpurun_out/${TAG}_$p.log 2>&1 || { echo "pass $p failed"; tail -5 $R/gpurun_out/${TAG}_$p.log; exit 1; }
  echo "== pass $p"
  python3 $R/tools/pmc_summary.py "$R/gpurun_out/${TAG}_$p/**/*counter_collection.csv" | grep -v "^slic_\(fill\|finalise\|init\|prepare\|scan\)"
done
