#!/bin/bash
# PMC passes over the conv microbench (shape list index $1, default 0): SQ wait / MFMA-busy / LDS counters per kernel.
# usage (GPU box): bash tools/pmc_conv.sh 0 [lib variant]
S=${1:-0}; R=$(pwd); export TMPDIR=/tmp
[ -n "$2" ] && cp variants/lib_$2.so patchrefinerv2_amd/libprv2_hip.so
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  i=$((i+1)); rm -rf /tmp/pc_$i
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pc_$i -- python3 $R/tools/bench_conv.py bf16x3 27 $S > /tmp/pc_$i.log 2>&1 || tail -3 /tmp/pc_$i.log
done
python3 - <<'PY'
import csv, glob, collections
tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int)); dur=collections.defaultdict(list)
for f in glob.glob('/tmp/pc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'halo' not in k and 'igemm' not in k: continue
        tot[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
        dur[k].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k in tot:
    print(k, 'avg dur us', sum(dur[k])/len(dur[k])/1e3)
    for c in sorted(tot[k]): print(f"   {c:34s} {tot[k][c]/n[k][c]:16.1f} per launch")
PY
