#!/bin/bash
# SQ counters of csrc/upconv.hip (prv2_upconv3x3) and of the loader kernel it replaces (conv3x3_halo16_ups_kernel) on the layers of
# tools/probes/upconv_time.py (14 tiles): wave-cycle split, MFMA pipe busy, effective clock, LDS activity / bank conflicts, HBM bytes.
# Separate --pmc passes, no trace domains combined.
# usage (GPU box, repo root): bash tools/pmc_upconv.sh > gpurun_out/profiles/rNN_bf16x3_pmc_sq_upconv.txt
R=$(pwd); export TMPDIR=/tmp; export PYTHONPATH=$R
cd /tmp; rm -rf /tmp/pu_*
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pu_$i -- python3 $R/tools/probes/upconv_time.py 14 > /tmp/pu_$i.log 2>&1; grep -q "upconv3x3" /tmp/pu_$i.log || tail -5 /tmp/pu_$i.log
done
python3 - <<'PY'
import csv, glob, collections
tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int)); dur=collections.defaultdict(list)
for f in glob.glob('/tmp/pu_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('prv2::','')
        if not any(t in k for t in ('upconv', 'halo16_ups')): continue
        k += " grid %s" % r.get('Grid_Size', '?')   # one line per layer shape
        tot[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
        dur[k].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print("tools/probes/upconv_time.py 14 (bf16x3, 14 tiles): 256->128 @384x512, 256->290 @384x512, 256->322 @96x128, 512->642 @48x64, 128->194 @192x256, 64->98 @384x512; per launch")
for k in sorted(tot, key=lambda k: -sum(dur[k]) / len(dur[k])):
    c={x: tot[k][x]/n[k][x] for x in tot[k]}
    d=sum(dur[k])/len(dur[k])/1e3
    cyc=c.get('GRBM_GUI_ACTIVE',0)/8
    w=c.get('SQ_WAVE_CYCLES',1)
    print(f"{k}: avg {d:.0f} us, {cyc/1e6:.2f} Mcycles/XCD -> {cyc/d/1e3 if d else 0:.2f} GHz; MFMA pipe busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(1024*cyc)*100 if cyc else 0:.1f} %")
    print("    wave-cycles: wait_any %.1f %%  wait_inst_any %.1f %%  active %.1f %%  wait_inst_lds %.1f %%;  LDS idx active %.0fM conflicts %.1fM  VALU insts %.0fM;  HBM-side fetch %.0f MB (2 x FETCH_SIZE) write %.0f MB" % (
        100*c.get('SQ_WAIT_ANY',0)/w, 100*c.get('SQ_WAIT_INST_ANY',0)/w, 100*c.get('SQ_ACTIVE_INST_ANY',0)/w, 100*c.get('SQ_WAIT_INST_LDS',0)/w,
        c.get('SQ_LDS_IDX_ACTIVE',0)/1e6, c.get('SQ_LDS_BANK_CONFLICT',0)/1e6, c.get('SQ_INSTS_VALU',0)/1e6, 2*c.get('FETCH_SIZE',0)/1024, c.get('WRITE_SIZE',0)/1024))
PY
cd $R
