for mb in 41 27 21 81; do for st in 3 2 4; do
python3 bench.py --no-alt --no-cpu-baseline --no-roofline --steps 4 --warmup 2 --max-batch $mb --streams $st 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('max_batch $mb streams $st', round(d['ms_per_step'],2), 'ms')"
done; done
