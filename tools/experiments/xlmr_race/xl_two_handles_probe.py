"""Debug probe 6: the two-lane difference without run_in_lanes -- TWO handles (one lane each), each on its own stream, encoding their own
batches concurrently.  Handle A is always the folded (implicit-LayerNorm) pass with the MFMA attention; handle B's options come from the
command line, so the kernel families that must run NEXT to A for A to differ can be told apart.
python tools/experiments/xlmr_race/xl_two_handles_probe.py "xlmr_fold=0,attn_mfma=0" [B L runs layers]"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.xlmr import XLMRoberta
optsB = sys.argv[1] if len(sys.argv) > 1 else ""
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
L = int(sys.argv[3]) if len(sys.argv) > 3 else 32
RUNS = int(sys.argv[4]) if len(sys.argv) > 4 else 600
LAYERS = int(sys.argv[5]) if len(sys.argv) > 5 else 2
OFFSET = int(os.environ.get("OFFSET", "0"))
sd = synth.xlmr_state_dict(layers=LAYERS)
engs, xls, ins, streams, base = [], [], [], [], []
for k in range(2):
    e = Engine(0)
    e.set_option("xlmr_lanes", 1)
    if k == 0:
        for kv in os.environ.get("OPTS_A", "").split(","):
            if kv:
                e.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    if k == 1:
        for kv in optsB.split(","):
            if kv:
                e.set_option(kv.split("=")[0], int(kv.split("=")[1]))      # before the weights: xlmr_fold decides how they are packed
    xls.append(XLMRoberta(engine=e).load_state_dict(sd))
    engs.append(e)
    ids, mask = synth.xlmr_inputs(3 + k, B, L)
    ins.append((torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()))
    streams.append(torch.cuda.Stream(priority=-1 if k == 1 else 0))      # B's stream of high priority: a hardware queue of the other pool, so A and B really overlap
torch.cuda.synchronize()
for k in range(2):
    base.append(xls[k](ins[k][0], attention_mask=ins[k][1]).last_hidden_state.clone())
    torch.cuda.synchronize()
for k in range(2):      # alone, a handle repeats itself
    for _ in range(20):
        assert torch.equal(xls[k](ins[k][0], attention_mask=ins[k][1]).last_hidden_state, base[k])
import time
def bench(which, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for k in which:
            with torch.cuda.stream(streams[k]):
                xls[k](ins[k][0], attention_mask=ins[k][1])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print(f"us per iteration: A alone {bench([0]):.0f}, B alone {bench([1]):.0f}, A and B on two streams {bench([0, 1]):.0f}")
GEST = os.environ.get("GESTURE_B")              # handle B runs the GESTURE path instead (one lane): text on one stream, video on the other
if GEST:
    from jegal_amd.gestsync import GestSync
    from jegal_amd.jegal import JEGAL
    engs[1].set_option("dual_stream", 0)
    GestSync(engine=engs[1]).load_state_dict(synth.gestsync_state_dict(include_unused=False))
    JEGAL(engine=engs[1]).load_state_dict(synth.jegal_state_dict())
    gframes = torch.from_numpy(synth.synth_frames(99, 8, 40)).cuda()
    gbase = engs[1].extract_gesture(gframes).clone()
    torch.cuda.synchronize()
bad = [0, 0]
for it in range(RUNS):
    outs = [None, None]
    order = (0, 1) if it & 1 else (1, 0)
    gate = torch.cuda.Event()
    torch.cuda._sleep(3_000_000)                  # ~1.5 ms on the default stream: both handles' launches queue up behind it and start together
    gate.record()
    for k in order:
        with torch.cuda.stream(streams[k]):
            streams[k].wait_event(gate)
            if k == 1 and OFFSET:
                torch.cuda._sleep(OFFSET)         # handle B trails handle A like lane 1 trails lane 0 (the host enqueues lane 0 first)
            if os.environ.get("POISON"):
                engs[k].set_option("ws_poison", 1)
            if k == 1 and GEST:
                outs[k] = engs[1].extract_gesture(gframes)
            else:
                outs[k] = xls[k](ins[k][0], attention_mask=ins[k][1]).last_hidden_state
    torch.cuda.synchronize()
    for k in range(2):
        if not torch.equal(outs[k], gbase if (k == 1 and GEST) else base[k]):
            bad[k] += 1
print(f"{'B = gesture path; ' if GEST else ''}offset {OFFSET} cycles; A options [{os.environ.get('OPTS_A', '')}] B options [{optsB}] B {B} L {L} layers {LAYERS}: handle A (folded, MFMA attention) differs in {bad[0]} of {RUNS} runs, handle B in {bad[1]}")
