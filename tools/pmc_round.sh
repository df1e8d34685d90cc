#!/bin/bash
# Runs on the GPU box (gpurun): SQ counter passes (LDS bank conflicts / LDS busy, VALU busy, MFMA busy, wait states) over
# tools/pmc_driver.py.  Usage: tools/pmc_round.sh <tag> [driver args].  Counters only with --kernel-trace, one group of at
# most 8 SQ counters per pass, as the pool requires.  Summarise with tools/make_pmc_profile.py <tag>.
tag=${1:-r02}; shift
out=/root/repo/gpurun_out/${tag}_pmc
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
C="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
# L2 (TCC): hit rate of the kernels' requests - what share of a sweep's halo gathers the XCD's 4 MB L2 serves
D="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
i=0
for set in "$A" "$B" "$C" "$D"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pass$i" -- python3 /root/repo/tools/pmc_driver.py "$@" > "$out/pass$i.log" 2>&1
done
find "$out" -name "*agent_info.csv" -delete
# keep the rows of the library's kernels only (the torch data generators dispatch thousands of kernels: 20 MB per pass, and
# gpurun copies at most 64 MB back)
for f in $(find "$out" -name "*counter_collection.csv" -o -name "*kernel_trace.csv"); do
  head -1 "$f" > "$f.tmp"; grep "fdx::" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"
done
for l in "$out"/pass*.log; do echo "== $l"; grep -v "rocprofv3\|Opened result" "$l" | tail -3; done
du -sh "$out"; find "$out" -type f | head
