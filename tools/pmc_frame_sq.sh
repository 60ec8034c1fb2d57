#!/bin/bash
# SQ counters of every conv / GEMM kernel INSIDE one frame of the default bench workload (real activations, one stream): MFMA pipe busy,
# effective clock (GRBM_GUI_ACTIVE / 8 / duration), wave-cycle split.  Separate --pmc passes over bench.py (no trace domains combined).
# usage (GPU box, repo root): bash tools/pmc_frame_sq.sh > gpurun_out/profiles/rNN_bf16x3_pmc_sq_frame.txt
R=$(pwd); export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/pf_*
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pf_$i -- python3 $R/bench.py --steps 1 --warmup 1 --streams 1 --no-roofline --no-cpu-baseline > /tmp/pf_$i.log 2>&1 || tail -3 /tmp/pf_$i.log
done
python3 - <<'PY'
import csv, glob, collections
tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int)); dur=collections.defaultdict(list)
for f in glob.glob('/tmp/pf_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('prv2::','')
        if not any(t in k for t in ('conv3x3', 'gemm', 'attention', 'igemm')): continue
        tot[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
        dur[k].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print("v2_zoe_4k_r32, bf16x3, one stream, per kernel over all its launches of the profiled frames (duration-weighted): time share, effective clock, MFMA pipe busy")
tsum={k: sum(dur[k]) for k in dur}
for k in sorted(tot, key=lambda k: -tsum[k])[:14]:
    c={x: tot[k][x] for x in tot[k]}   # sums over launches
    launches=n[k].get('GRBM_GUI_ACTIVE', 0) or 1
    d=tsum[k] / max(1, len(dur[k])) * launches / 1e3   # us covered by the GRBM pass
    cyc=c.get('GRBM_GUI_ACTIVE',0)/8
    w=c.get('SQ_WAVE_CYCLES',1)
    npass = len(glob.glob('/tmp/pf_*/'))
    mf = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * cyc) * 100 if cyc else 0
    print(f"{k}: {100 * tsum[k] / sum(tsum.values()):.1f} pct of the time, {cyc / d / 1e3 if d else 0:.2f} GHz; MFMA pipe busy {mf:.1f} pct;  wave-cycles: "
          f"wait_any {100 * c.get('SQ_WAIT_ANY', 0) / w:.1f} pct  wait_inst_any {100 * c.get('SQ_WAIT_INST_ANY', 0) / w:.1f} pct  active {100 * c.get('SQ_ACTIVE_INST_ANY', 0) / w:.1f} pct")
PY
cd $R
