python bench.py --wsteps 2 --steps 1 --warmup 1 --no-cpu-baseline --no-modconv --no-single-stream --no-end-to-end --no-roofline-events 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['forward_only']; print({k:f['b1'][k] for k in ('latency_ms','encoder_ms','ood_forward_ms','graph_replay_latency_ms')}, {k:f['b8'][k] for k in ('latency_ms','images_per_s','encoder_ms','ood_forward_ms')})"
