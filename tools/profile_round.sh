#!/bin/bash
# Run ON the GPU box (gpurun -- 'bash tools/profile_round.sh r01'): regenerates the profiles/ evidence of the default
# bench workload into gpurun_out/profiles/ (copy what should be judged into profiles/ afterwards).
set -u
TAG=${1:-r06}; PREC=${2:-f16f6}   # PREC: the bench's arithmetic (its default: bf16x3 + the fp16 / fp6 layers)
R=$(pwd); O=$R/gpurun_out/profiles; mkdir -p $O
WL=$(python3 -c "from patchrefinerv2_amd.workloads import DEFAULT_WORKLOAD as w; print(w)")
export TMPDIR=/tmp
python3 bench.py --prec $PREC --steps 5 --warmup 2 > $O/${TAG}_${PREC}_bench_${WL}.json 2> $O/bench.err
for W2 in v1_zoe_4k_r32 v1_dav2l_4k_r32 v2_zoeda_4k_r32 v2_dav2l_4k_r64; do
  python3 bench.py --prec $PREC --no-alt --workload $W2 --steps 3 --warmup 1 --no-cpu-baseline > $O/${TAG}_${PREC}_bench_${W2}.json 2>> $O/bench.err
done
# (config[1] is a 6.5 ms frame: three steps behind one warm-up measure the process warming up -- 7.5 ms -- not the frame)
python3 bench.py --prec $PREC --no-alt --workload v1_dav2s_1080p_m1 --steps 30 --warmup 5 --no-cpu-baseline > $O/${TAG}_${PREC}_bench_v1_dav2s_1080p_m1.json 2>> $O/bench.err
python3 bench.py --prec f32 --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_f32_bench_${WL}.json 2>> $O/bench.err
python3 bench.py --prec $PREC --no-alt --layer-report $O/${TAG}_${PREC}_layers_${WL}.csv --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>> $O/bench.err
# layer reports of the f32 mode and of the ViT-heavy README example
python3 bench.py --prec f32 --layer-report $O/${TAG}_f32_layers_${WL}.csv --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>> $O/bench.err
python3 bench.py --prec $PREC --no-alt --workload v1_zoe_4k_r32 --layer-report $O/${TAG}_${PREC}_layers_v1_zoe_4k_r32.csv --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>> $O/bench.err
# the bench's cpu_baseline extrapolation against a fully timed oracle frame (m1, 16 tiles)
python3 tools/cpu_baseline_validate.py > $O/${TAG}_cpu_baseline_validation.json 2>> $O/bench.err
python3 tools/shard_model.py > $O/${TAG}_shard_model.json 2>> $O/bench.err
# the power-limited frame's operating range: all-zero frames (upper end) beside the default random frames
python3 bench.py --prec $PREC --no-alt --data zeros --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/${TAG}_${PREC}_bench_${WL}_zeros.json 2>> $O/bench.err
# (the per-feature A/Bs of rounds 4-5 -- coarse taps, upconv, chain32, upconv5, dispatch route, PRV2_F6_GATE -- are in profiles/r04_*, r05_*: git history of this script)
# round 6: the ViT blocks -- persistent gemm_ss workgroups and the attention block without the qkv_split pre-pass -- on the ViT-heavy V1 workload, alternating
for T in 1 0 1 0; do PRV2_GSS_PERSIST=$T PRV2_QKV_SS=$T python3 bench.py --prec $PREC --no-alt --workload v1_zoe_4k_r32 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>> $O/bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('v1_zoe_4k_r32 PRV2_GSS_PERSIST = PRV2_QKV_SS = $T', round(d['ms_per_step'],1), 'ms', round(d['value'],3), 'maps/s')" >> $O/${TAG}_vit_ab.txt; done
python3 tools/probes/vit_ab.py 14 3 2>/dev/null | grep -v amdgpu.ids >> $O/${TAG}_vit_ab.txt
NTOK=769 python3 tools/probes/vit_ab.py 41 2 2>/dev/null | grep -v amdgpu.ids >> $O/${TAG}_vit_ab.txt
bash tools/probes/gss_ablate_run.sh > $O/${TAG}_gemm_ss_ablations.txt 2>&1
# the frame in both arithmetics on this box (alternating), and the bf16x3 mode's own headline line + layer table
for T in f16f6 bf16x3 f16f6 bf16x3; do python3 bench.py --prec $T --no-alt --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>> $O/bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('--prec $T', d['dtype'], round(d['ms_per_step'],2), 'ms', round(d['value'],3), 'maps/s', d['operating_point'])"; done >> $O/${TAG}_f16f6_ab.txt
python3 bench.py --prec bf16x3 --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bf16x3_bench_${WL}.json 2>> $O/bench.err
python3 bench.py --prec bf16x3 --layer-report $O/${TAG}_bf16x3_layers_${WL}.csv --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>> $O/bench.err
# kernel durations only mean something un-overlapped: the traced runs use one stream, like the roofline pass inside bench.py
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --prec $PREC --no-alt --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline > /tmp/kt.log 2>&1
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) $O/${TAG}_${PREC}_kernel_stats_${WL}.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -- python3 $R/bench.py --prec $PREC --no-alt --steps 1 --warmup 0 --streams 1 --no-roofline --no-cpu-baseline > /tmp/pmc_$C.log 2>&1
done
python3 $R/tools/pmc_to_json.py $O/${TAG}_${PREC}_pmc_frame_${WL}.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
# kernel stats of one ViT-heavy workload (BEiT-L on every tile)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python3 $R/bench.py --prec $PREC --no-alt --workload v1_zoe_4k_r32 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline > /tmp/kt2.log 2>&1
cp $(find /tmp/kt2 -name '*kernel_stats.csv' | head -1) $O/${TAG}_${PREC}_kernel_stats_v1_zoe_4k_r32.csv
cd $R
bash tools/pmc_vit.sh 14 1 > $O/${TAG}_${PREC}_pmc_vit_blocks_b14.txt 2>&1
bash tools/pmc_gate.sh > $O/${TAG}_${PREC}_pmc_sq_dominant_kernel.txt 2>&1
ls -la $O
