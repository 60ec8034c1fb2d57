#!/bin/bash
# PMC passes over the ViT-L block probe (tools/probes/vit_block_bench.py): per kernel the wave-cycle split, MFMA pipe busy, effective
# clock, L2 (TCC) hit rate, vector-L1 -> L2 read requests and their latency, texture-addresser busy, HBM-side bytes.
# usage (GPU box, from the repo root): bash tools/pmc_vit.sh [batch=14] [ss=1]     (separate --pmc passes: no trace domains combined)
B=${1:-14}; SS=${2:-1}; R=$(pwd); export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/pv_*
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pv_$i -- python3 $R/tools/probes/vit_block_bench.py $B 1 $SS > /tmp/pv_$i.log 2>&1 || tail -3 /tmp/pv_$i.log
done
python3 - <<'PY'
import csv, glob, collections
tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int)); dur=collections.defaultdict(list)
for f in glob.glob('/tmp/pv_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('prv2::','')
        if not any(t in k for t in ('gemm16','gemm_ss','attention_','qkv_split','layernorm')): continue
        tot[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
        dur[k].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k in sorted(tot, key=lambda k: -sum(dur[k])):
    c={x: tot[k][x]/n[k][x] for x in tot[k]}
    d=sum(dur[k])/len(dur[k])/1e3
    cyc=c.get('GRBM_GUI_ACTIVE',0)/8
    w=c.get('SQ_WAVE_CYCLES',1)
    print(f"{k}: {max(n[k].values())} launches (per pass), avg {d:.0f} us, clock {cyc/d/1e3 if d else 0:.2f} GHz, MFMA pipe busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(1024*cyc)*100 if cyc else 0:.1f} %")
    print("    wave-cycles: wait_any %.1f %%  wait_inst_any %.1f %%  active %.1f %%  wait_inst_lds %.1f %%" % (100*c.get('SQ_WAIT_ANY',0)/w, 100*c.get('SQ_WAIT_INST_ANY',0)/w, 100*c.get('SQ_ACTIVE_INST_ANY',0)/w, 100*c.get('SQ_WAIT_INST_LDS',0)/w))
    print("    LDS: idx active %.1fM cycles of which bank conflicts %.2fM;  VALU insts %.0fM" % (c.get('SQ_LDS_IDX_ACTIVE',0)/1e6, c.get('SQ_LDS_BANK_CONFLICT',0)/1e6, c.get('SQ_INSTS_VALU',0)/1e6))
    hit, miss = c.get('TCC_HIT_sum',0), c.get('TCC_MISS_sum',0)
    rd = c.get('TCP_TCC_READ_REQ_sum',0)
    print("    L2: hit rate %.1f %% of %.2f M requests;  L1->L2 read requests %.2f M, mean latency %.0f cycles;  TA busy %.1f %%;  HBM-side: fetch %.1f MB (2 x FETCH_SIZE) write %.1f MB per launch" % (
        100*hit/max(hit+miss,1), c.get('TCC_REQ_sum',0)/1e6, rd/1e6, c.get('TCP_TCC_READ_REQ_LATENCY_sum',0)/max(rd,1), c.get('TA_BUSY_avr',0), 2*c.get('FETCH_SIZE',0)/1024, c.get('WRITE_SIZE',0)/1024))
PY
cd $R
