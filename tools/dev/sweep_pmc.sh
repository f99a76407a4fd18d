#!/bin/bash
# dev probe (GPU box): SQ counters of the three landing_sweep_kernel streams, each launched alone (tools/dev/sweep_parts.py B)
#   tools/dev/sweep_pmc.sh OUT [B]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; B=${2:-4096}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
P="python3 $GRAFT_REPO_ROOT/tools/dev/sweep_parts.py $B"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F64" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TOTAL_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_TA_BUSY_sum TA_FLAT_WRITE_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_WRITE_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/sp_$i
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sp_$i -- $P > /dev/null 2> $out/pmc_$i.err
done
python3 - <<'PY' > $out/sweep_pmc.json
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("/tmp/sp_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "landing_sweep_kernel" not in k: continue
        fam = k[k.index("<") + 1:k.index(">")] if "<" in k else k[-40:]
        acc[fam][r["Counter_Name"]] += float(r["Counter_Value"]); n[fam][r["Counter_Name"]].add(r["Dispatch_Id"])
print(json.dumps({fam: {c: acc[fam][c] / max(len(n[fam][c]), 1) for c in sorted(acc[fam])} for fam in sorted(acc)}, indent=1))
PY
cat $out/sweep_pmc.json
