# which kernels of the LEADING handle (A) must run next to the trailing folded pass (B) for B to differ?  offsets in shader cycles
for oa in "" "attn_mfma=0" "xlmr_fold=0" "xlmr_fold=0,attn_mfma=0"; do
for off in 20000 40000 60000 80000 100000 120000 140000 160000 180000 200000 250000 300000; do
OPTS_A="$oa" OFFSET=$off python tools/experiments/xlmr_race/xl_two_handles_probe.py "${OPTS_B:-}" 32 32 ${RUNS:-300} 2>&1 | tail -1 | sed 's/B 32 L 32 layers 2: handle A (folded, MFMA attention)/A/'
done; done
