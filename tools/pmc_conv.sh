#!/bin/bash
# PMC passes over the conv microbench: SQ wait / MFMA-busy / LDS counters + effective clock per kernel.
# usage (GPU box): bash tools/pmc_conv.sh <shape index> <lib variant> [<lib variant> ...]   (variants/lib_<v>.so)
S=${1:-0}; shift; R=$(pwd); export TMPDIR=/tmp
cp patchrefinerv2_amd/libprv2_hip.so /tmp/lib_keep.so
for V in "$@"; do
cp $R/variants/lib_$V.so $R/patchrefinerv2_amd/libprv2_hip.so
cd /tmp; rm -rf /tmp/pc_*
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pc_$i -- python3 $R/tools/bench_conv.py bf16x3 27 $S > /tmp/pc_$i.log 2>&1 || tail -3 /tmp/pc_$i.log
done
echo "== $V"
python3 - <<'PY'
import csv, glob, collections
tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int)); dur=collections.defaultdict(list)
for f in glob.glob('/tmp/pc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'halo' not in k and 'gemm16' not in k: continue
        tot[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
        dur[k].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k in tot:
    c={x: tot[k][x]/n[k][x] for x in tot[k]}
    d=sum(dur[k])/len(dur[k])/1e3
    cyc=c.get('GRBM_GUI_ACTIVE',0)/8
    print(f"{k}: {d:.0f} us, {cyc/1e6:.2f} Mcycles/XCD -> {cyc/d/1e3:.2f} GHz; MFMA busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(1024*cyc)*100:.1f} %")
    w=c.get('SQ_WAVE_CYCLES',1)
    print("   wave-cycles: wait_any %.1f %%  wait_inst_any %.1f %%  active %.1f %%  wait_inst_lds %.1f %%;  LDS idx active %.0fM conflicts %.0fM  VALU insts %.0fM" % (
        100*c.get('SQ_WAIT_ANY',0)/w, 100*c.get('SQ_WAIT_INST_ANY',0)/w, 100*c.get('SQ_ACTIVE_INST_ANY',0)/w, 100*c.get('SQ_WAIT_INST_LDS',0)/w,
        c.get('SQ_LDS_IDX_ACTIVE',0)/1e6, c.get('SQ_LDS_BANK_CONFLICT',0)/1e6, c.get('SQ_INSTS_VALU',0)/1e6))
PY
cd $R
done
cp /tmp/lib_keep.so $R/patchrefinerv2_amd/libprv2_hip.so
