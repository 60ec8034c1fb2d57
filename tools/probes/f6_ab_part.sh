O=gpurun_out/profiles; mkdir -p $O
python3 tools/probes/c256_bench.py both 2>&1 | grep "256->256" > $O/r05_f16f6_ab_head.txt
python3 tools/probes/gate_taps_bench.py 14 192 256 2>&1 | grep "gate" >> $O/r05_f16f6_ab_head.txt
python3 tools/probes/gate_taps_bench.py 14 96 128 2>&1 | grep "gate" >> $O/r05_f16f6_ab_head.txt
for G in 1 0 1 0; do PRV2_F6_GATE=$G python3 bench.py --prec f16f6 --no-alt --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('--prec f16f6 PRV2_F6_GATE=$G (stage 2: the GatedConvUnit tail kernel)', round(d['ms_per_step'],2), 'ms')"; done >> $O/r05_f16f6_ab_head.txt
cat $O/r05_f16f6_ab_head.txt
