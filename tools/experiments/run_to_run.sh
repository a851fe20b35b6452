# run-to-run spread of the headline in fresh processes on one box: step time (two lanes) and the one-stream stage table of each run
for i in 1 2 3 4 5 6 7 8 9 10; do python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('%.3f ms/step two lanes | one stream: conv1 %.3f convs %.3f gemm %.3f attention %.3f | conv1 launch %.3f ms' % (d['ms_per_step'], s['conv1'], s['conv2-fc6+audio_cnn'], s['gemm'], s['attention'], d['roofline']['launch_ms']))"; done
