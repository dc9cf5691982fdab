#!/bin/bash
# usage (GPU box): tools/prof_pmc.sh <tag> <targets...>   -> gpurun_out/pmc_<tag>/{sq1,sq2,fetch,write}/...csv
# separate --pmc passes, no trace domains beside them (gpurun refuses pmc + sys/hip traces)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out/sq1 -- python3 $root/tools/prof_target.py "$@" > $out/sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $out/sq2 -- python3 $root/tools/prof_target.py "$@" > $out/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/tools/prof_target.py "$@" > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/tools/prof_target.py "$@" > $out/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/tools/prof_target.py "$@" > $out/trace.log 2>&1
find $out -name "*.csv" | head -20
