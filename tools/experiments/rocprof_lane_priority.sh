cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lp in 3 0; do
python3 $R/bench.py --no-cpu-baseline --no-extras --opt lane_priority=$lp 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain   lane_priority=$lp', round(d['value']), round(d['ms_per_step'],3))"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lp$lp -o x -- python3 $R/bench.py --no-cpu-baseline --no-extras --opt lane_priority=$lp 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rocprof lane_priority=$lp', round(d['value']), round(d['ms_per_step'],3))"
done
python3 $R/tools/xlmr_bench.py 256 64 2>&1 | grep engine
