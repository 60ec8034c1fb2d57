#!/bin/bash
# SQ counters of the dominant kernels on their big layer (14 x 192 x 256, 512 -> 256 [-> 256 gate]): wave-cycle split, MFMA pipe
# busy, effective clock, LDS activity.  Separate --pmc passes over tools/probes/gate_conv_bench.py (no trace domains combined).
# usage (GPU box, repo root): bash tools/pmc_gate.sh > gpurun_out/profiles/rNN_bf16x3_pmc_sq_dominant_kernel.txt
R=$(pwd); export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/pg_*
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pg_$i -- python3 $R/tools/probes/gate_conv_bench.py 14 192 256 512 > /tmp/pg_$i.log 2>&1 || tail -3 /tmp/pg_$i.log
  # round 4: the unit as the frame runs it -- K = 256 gate kernel on the X2 ``out`` + the gathered coarse half (tap_gather_kernel)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pg_t$i -- python3 $R/tools/probes/gate_taps_bench.py 14 192 256 > /tmp/pg_t$i.log 2>&1 || tail -3 /tmp/pg_t$i.log
done
python3 - <<'PY'
import csv, glob, collections
tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int)); dur=collections.defaultdict(list)
for f in glob.glob('/tmp/pg_*/**/*counter_collection.csv', recursive=True):  # (pg_N: the 512-channel concat form; pg_tN: the round-4 form)
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('prv2::','')
        if not any(t in k for t in ('c256', 'halo16', 'gemm16', 'layernorm', 'tap_')): continue
        tot[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
        dur[k].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print("14 x 192 x 256 pixels, bf16x3; per launch.  conv3x3_c256_gate_kernel = the 512 -> 256 concat form (3x3 + LayerNorm + ReLU + 256 -> 256 gate + sigmoid * mul + res);\n"
      "conv3x3_c256_gate_x2_kernel = the round-4 form: K = 256 on the pre-split ``out`` + the coarse half as a pre-LayerNorm addend (tap_gather_kernel);\n"
      "conv3x3_c256_gate_f6_kernel / conv3x3_c256_f6_kernel = the same unit tail / GatedConvUnit.conv (K = 256) in the fp16 + fp6 arithmetic (round 5, csrc/conv3x3_f6.hip)")
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
