"""XLM-RoBERTa text front end: tokens/s of jg_xlmr_encode for the xlm-roberta-base depth (12 layers, seeded random
weights, reduced vocabulary) and, for reference, transformers.XLMRobertaModel on the box's CPU cores -- which is where the
reference runs it (models/jegal.py:116-129).  Usage: python tools/xlmr_bench.py [B L [option=value ...] [--no-cpu]]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.xlmr import XLMRoberta

B, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 64)
OPTS = [a.split("=") for a in sys.argv[3:] if "=" in a]          # engine options, e.g. dual_stream=0 dual_split=4
NO_CPU = "--no-cpu" in sys.argv
sd = synth.xlmr_state_dict(layers=12)
ids, mask = synth.xlmr_inputs(1, B, L)
eng = Engine(0)
for k, v in OPTS:
    eng.set_option(k, int(v))
m = XLMRoberta(engine=eng).load_state_dict(sd)
ids_d, mask_d = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
flop = B * L * 12 * 2 * (768 * 2304 + 768 * 768 + 2 * 768 * 3072) + B * 12 * 4 * L * L * 768
n = 20


def run(tag):
    global out
    for _ in range(3):
        out = m(ids_d, attention_mask=mask_d).last_hidden_state
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = m(ids_d, attention_mask=mask_d).last_hidden_state
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("engine (%s): B=%d L=%d 12 layers: %.3f ms per batch, %.0f tokens/s, %.1f TFLOP/s" % (tag, B, L, dt * 1e3, B * L / dt, flop / dt / 1e12))


run("hi+lo weights: the default until calibrate()")
m.calibrate(ids_d[:8], mask_d[:8])
run("bias-corrected single fp16, calibrated on 8 of these sequences")
try:
    if NO_CPU:
        raise RuntimeError("--no-cpu")
    from transformers import XLMRobertaConfig, XLMRobertaModel
    cfg = XLMRobertaConfig(vocab_size=sd["embeddings.word_embeddings.weight"].shape[0], hidden_size=768, num_hidden_layers=12,
                           num_attention_heads=12, intermediate_size=3072, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1,
                           layer_norm_eps=1e-5, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    hf = XLMRobertaModel(cfg, add_pooling_layer=False).eval()
    hf.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    with torch.no_grad():
        ref = hf(torch.from_numpy(ids).long(), attention_mask=torch.from_numpy(mask).long()).last_hidden_state
        t0 = time.perf_counter()
        for _ in range(3):
            hf(torch.from_numpy(ids).long(), attention_mask=torch.from_numpy(mask).long())
        dc = (time.perf_counter() - t0) / 3
    mm = torch.from_numpy(mask).bool()
    print("transformers on %d CPU threads: %.1f ms per batch, %.0f tokens/s; engine vs it: rel-L2 %.3e" %
          (torch.get_num_threads(), dc * 1e3, B * L / dc, float((out.cpu()[mm] - ref[mm]).norm() / ref[mm].norm())))
except Exception as e:                                        # transformers absent: engine figure only
    print("transformers baseline skipped:", e)
