#!/bin/bash
# LDS bank conflicts of every kernel of one headline frame (SQ_LDS_BANK_CONFLICT against SQ_LDS_IDX_ACTIVE): a survey -- which kernels
# deserve the per-instruction bank model of MI355X_MICROARCH.md (LDS).  One --pmc pass, no trace domains combined.
# usage (GPU box, repo root): bash tools/pmc_lds_frame.sh > gpurun_out/profiles/rNN_bf16x3_pmc_lds_frame.txt
R=$(pwd); export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/pl_1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pl_1 -- python3 $R/bench.py --steps 1 --warmup 0 --streams 1 --no-roofline --no-cpu-baseline > /tmp/pl_1.log 2>&1 || tail -3 /tmp/pl_1.log
python3 - <<'PY'
import csv, glob, collections
tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter(); dur=collections.Counter()
for f in glob.glob('/tmp/pl_1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('prv2::','')
        tot[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='SQ_LDS_IDX_ACTIVE': n[k]+=1; dur[k]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
print("one v2_zoe_4k_r32 frame (bf16x3, one stream): kernel, launches, total ms, LDS idx-active Mcycles, bank-conflict Mcycles, conflict share")
for k in sorted(tot, key=lambda k: -tot[k].get('SQ_LDS_BANK_CONFLICT',0)):
    a, c = tot[k].get('SQ_LDS_IDX_ACTIVE',0), tot[k].get('SQ_LDS_BANK_CONFLICT',0)
    if a < 1e6: continue
    print(f"{k[:70]:70s} {n[k]:4d} {dur[k]/1e6:8.2f} ms  {a/1e6:9.1f}  {c/1e6:8.1f}  {100*c/a:5.1f} %")
PY
cd $R
