for oa in "" "attn_mfma=0" "xlmr_fold=0"; do
for off in 40000 80000 120000 140000 160000 180000 200000; do
OPTS_A="$oa" OFFSET=$off python tools/experiments/xlmr_race/xl_two_handles_probe.py "" 32 32 300 2>&1 | tail -1 | sed 's/B 32 L 32 layers 2: handle A (folded, MFMA attention)/A/'
done; done
