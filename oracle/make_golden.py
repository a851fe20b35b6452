"""Generate tests/golden/*.npz by running the REAL reference (build container only).

Usage:  python oracle/make_golden.py            (needs /root/reference; never runs on the GPU box)

The reference's ``models.gestsync`` / ``models.jegal`` / ``evaluation.evaluate_*`` modules are
imported from /root/reference with two stubs for the HuggingFace downloads at
``models/jegal.py:13-14`` (no network here; XLM-R is third-party and out of the pinned
scope) and ``Tensor.cuda`` made a no-op (the reference hard-codes .cuda()).
The synthetic state_dicts of ``jegal_amd/synth.py`` are loaded with strict=True, which pins
every key name and shape.  Only inputs-by-seed and the reference's OUTPUTS are stored; no
reference source is copied.
"""
import ast
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from jegal_amd import synth  # noqa: E402


def _import_reference():
    import transformers

    class _Tok:
        cls_token_id, sep_token_id, pad_token_id = 0, 2, 1

    transformers.AutoTokenizer.from_pretrained = staticmethod(lambda *a, **k: _Tok())
    transformers.XLMRobertaModel.from_pretrained = staticmethod(lambda *a, **k: None)
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    from models.gestsync import GestSync
    from models.jegal import JEGAL
    return GestSync, JEGAL


def _load(model, sd):
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return model.eval()


def _import_eval(name):
    """evaluation/*.py parse argv at import (evaluate_spotting.py:11-15)."""
    argv = sys.argv
    sys.argv = [name, "--path", "/tmp", "--file", "/tmp/x.csv"] if name == "evaluate_asd" else [name, "--path", "/tmp"]
    sys.path.insert(0, os.path.join(REF, "evaluation"))
    try:
        mod = __import__(name)
    finally:
        sys.argv = argv
    return mod


def _reference_functions(path, names):
    """Pull plain functions out of a reference script that cannot be imported
    (inference_embs.py imports mediapipe/decord/whisperx) and exec them as-is."""
    src = open(path, encoding="utf-8").read()
    tree = ast.parse(src)
    ns = {"string": __import__("string")}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    return ns


def golden_asd(ea):
    """(vi-b) ASD at dataset level through the reference's own evaluate_asd(df) (evaluate_asd.py:54-125): 48 queries
    written as the .pkl files it loads, candidate lists of 1/3/5/6 clips, positives that lose in some queries.  Stored:
    the per-query argmax of the reference's get_similarity_cos for 2/4/6 speakers and the three counts it prints."""
    import contextlib
    import io
    import pickle
    import re
    import tempfile

    import pandas as pd
    n = 48
    contents, positives, negatives = synth.planted_asd(9007, n)
    preds = np.zeros((n, 3), np.int32)
    with tempfile.TemporaryDirectory() as tmp:
        rows = []
        for i in range(n):
            with open(os.path.join(tmp, f"q{i:03d}__00000.pkl"), "wb") as f:
                pickle.dump({"gesture_emb": positives[i], "content_emb": contents[i]}, f)
            negs = []
            for k, g in enumerate(negatives[i]):
                with open(os.path.join(tmp, f"q{i:03d}n{k}__00000.pkl"), "wb") as f:
                    pickle.dump({"gesture_emb": g, "content_emb": None}, f)
                negs.append(f"q{i:03d}n{k}/00000")
            rows.append({"filename": f"q{i:03d}/00000", "neg_files": str(negs)})
            q = torch.FloatTensor(contents[i].mean(axis=0)).unsqueeze(0)
            allg = torch.cat([torch.FloatTensor(positives[i].mean(axis=0)).unsqueeze(0)] +
                             [torch.FloatTensor(g.mean(axis=0)).unsqueeze(0) for g in negatives[i]])
            preds[i] = [int(np.argmax(ea.get_similarity_cos(q, allg[:k]))) for k in (2, 4, 6)]
        ea.args.path = tmp
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            ea.evaluate_asd(pd.DataFrame(rows))
    counts = [int(x) for x in re.findall(r"spk: Correct: (\d+) \| Total: 48", buf.getvalue())]
    assert len(counts) == 3, buf.getvalue()
    assert counts == [int((preds[:, k] == 0).sum()) for k in range(3)]
    assert 0 < counts[2] < counts[0] < n, counts          # the positive loses in some queries, more often with more speakers
    np.savez_compressed(os.path.join(OUT, "asd.npz"), seed=9007, n=n, preds=preds, correct=np.asarray(counts, np.int32))
    print("asd golden: correct", counts, "of", n)


def golden_xlmr(layers_arg=None, fname="xlmr.npz", seed=9011, B=3, L=24):
    """(vii) XLM-RoBERTa text front end: the third-party model itself -- transformers.XLMRobertaModel with the
    xlm-roberta-base architecture (reduced vocabulary / depth, the released checkpoint is not available offline), strict-loaded
    with the seeded weights of synth.xlmr_state_dict -- run as the reference's call site does (jegal.py:126-127).
    xlmr.npz: 4 layers (round 2); xlmr12.npz: the full depth of xlm-roberta-base, 12 layers, 514 positions (round 3)."""
    import transformers
    from transformers import XLMRobertaConfig, XLMRobertaModel
    sd = synth.xlmr_state_dict() if layers_arg is None else synth.xlmr_state_dict(layers=layers_arg)
    layers = 0
    while f"encoder.layer.{layers}.attention.self.query.weight" in sd:
        layers += 1
    cfg = XLMRobertaConfig(vocab_size=sd["embeddings.word_embeddings.weight"].shape[0], hidden_size=768, num_hidden_layers=layers,
                           num_attention_heads=12, intermediate_size=3072, max_position_embeddings=514, type_vocab_size=1,
                           pad_token_id=1, bos_token_id=0, eos_token_id=2, layer_norm_eps=1e-5, hidden_act="gelu",
                           hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = XLMRobertaModel(cfg, add_pooling_layer=False).eval()
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    missing = [k for k in missing if "position_ids" not in k and "token_type_ids" not in k]      # registered buffers, not weights
    assert not missing and not unexpected, (missing, unexpected)
    ids, mask = synth.xlmr_inputs(seed, B, L)
    with torch.no_grad():
        out = model(torch.from_numpy(ids).long(), attention_mask=torch.from_numpy(mask).long()).last_hidden_state
        out_nomask = model(torch.from_numpy(ids[:1]).long()).last_hidden_state
    np.savez_compressed(os.path.join(OUT, fname), seed=seed, layers=layers, input_ids=ids, attention_mask=mask, last_hidden_state=out.numpy(),
                        last_hidden_state_nomask=out_nomask.numpy(), transformers_version=transformers.__version__)
    print("xlmr golden", fname, ":", out.shape, layers, "layers, transformers", transformers.__version__)


def golden_logmel():
    """(viii) log-mel front end from the reference ITSELF: utils/audio_utils.py imports librosa at module
    level but wav2filterbanks takes `mel_basis` as a parameter (:28,:54-62), so with an empty `librosa` module in sys.modules the
    reference's own load_wav + wav2filterbanks run here on its own samples/sample1.wav.  Stored: the (216,80) features, the
    sample count and a checksum of the raw samples load_wav returned.  mel_basis = jegal_amd.audio.mel_filterbank() (the
    librosa.filters.mel restatement stays unpinned: librosa is absent)."""
    import hashlib
    from jegal_amd import audio
    sys.modules.setdefault("librosa", types.ModuleType("librosa"))
    sys.path.insert(0, REF)
    from utils import audio_utils as ref_audio
    wav = ref_audio.load_wav(os.path.join(REF, "samples", "sample1.wav"))
    mel_basis = torch.from_numpy(audio.mel_filterbank())
    with torch.no_grad():
        feats = ref_audio.wav2filterbanks(torch.FloatTensor(wav).unsqueeze(0), mel_basis=mel_basis)[0]
    np.savez_compressed(os.path.join(OUT, "logmel_sample1.npz"), features=feats[0].numpy(), n_samples=np.int64(wav.shape[0]),
                        wav_dtype=str(wav.dtype), wav_sha256=hashlib.sha256(np.ascontiguousarray(wav).tobytes()).hexdigest())
    print("logmel_sample1.npz:", tuple(feats.shape), wav.dtype, wav.shape)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "logmel":        # regenerate only tests/golden/logmel_sample1.npz
        os.makedirs(OUT, exist_ok=True)
        golden_logmel()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "asd":           # regenerate only tests/golden/asd.npz
        golden_asd(_import_eval("evaluate_asd"))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "xlmr":          # regenerate only tests/golden/xlmr.npz
        os.makedirs(OUT, exist_ok=True)
        torch.manual_seed(0)
        golden_xlmr()
        golden_xlmr(layers_arg=12, fname="xlmr12.npz", seed=9012, B=2, L=40)
        return
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    GestSync, JEGAL = _import_reference()
    gs = _load(GestSync(), synth.gestsync_state_dict())
    jg = _load(JEGAL(), synth.jegal_state_dict())

    with torch.no_grad():
        # (i) GestSync clip: the reference's literal window loop (inference_embs.py:476-522)
        T = 6
        frames = synth.synth_frames(9001, 1, T)[0]                        # (T,270,480,3) u8
        f01 = np.pad(frames / 255., ((12, 12), (0, 0), (0, 0), (0, 0)), "edge")
        f01 = torch.FloatTensor(f01).unsqueeze(0)                         # (1,P,H,W,3)
        wins = [f01[:, i:i + 25] for i in range(f01.shape[1] - 24)]
        x = torch.stack(wins)[:, 0].permute(0, 4, 1, 2, 3)                # (N,3,25,270,480)
        out, out_conv = gs.forward_vid(x, return_feats=True)
        feats = out.mean(-1)
        c1 = gs.net_vid.mp1(torch.relu(gs.net_vid.bn1(gs.net_vid.conv1(x[:1]))))   # (1,64,21,43,78)
        np.savez_compressed(os.path.join(OUT, "gestsync_clip.npz"), seed=9001, T=T,
                            feats=feats.numpy(), out_conv=out_conv.numpy(), out_full=out[:2].numpy(),
                            conv1_pool_t0=c1[0, :, 0, ::6, ::6].numpy())

        # (ii) JEGAL gesture branch, one padded clip
        rng = np.random.default_rng(9002)
        vf = rng.standard_normal((2, 40, 1024)).astype(np.float32)
        vf[1, 30:] = 0
        vm = np.ones((2, 40), np.float32)
        vm[1, 30:] = 0
        g = jg.forward_inference(visual_feats=torch.from_numpy(vf), visual_mask=torch.from_numpy(vm))
        fg = jg.forward_gestures(torch.from_numpy(vf), torch.from_numpy(vm).unsqueeze(1))
        np.savez_compressed(os.path.join(OUT, "jegal_gesture.npz"), seed=9002, gesture=g.numpy(), fwd_gestures=fg.numpy())

        # (iii) audio branch -> content
        mel = synth.synth_mel(9003, 2, 160)
        wb = [[["a", 3, 9], ["b", 10, 10], ["c", 12, 30]], [["d", 0, 5], ["e", 6, 20]]]
        am = torch.ones(2, 40)
        c_a = jg.forward_inference(audio=torch.from_numpy(mel), audio_mask=am, word_boundaries=wb)
        fa = jg.forward_audio(torch.from_numpy(mel))
        np.savez_compressed(os.path.join(OUT, "jegal_audio.npz"), seed=9003, content=c_a.numpy(), fwd_audio=fa.numpy())

        # (iv) text branch: synthetic XLM-R states incl. a multi-sub-word word and a padded row
        rng = np.random.default_rng(9004)
        L = 9
        states = rng.standard_normal((2, L, 768)).astype(np.float32)
        ids = np.array([[0, 11, 12, 13, 14, 15, 16, 2, 1], [0, 21, 22, 23, 2, 1, 1, 1, 1]], np.int64)
        # clip 0: words start at tokens 1,2(+3 continuation),4,5(+6 continuation) ; clip 1: 3 words
        offs = np.zeros((2, L, 2), np.int64)
        offs[0, :, 0] = [0, 0, 0, 3, 0, 0, 2, 0, 0]
        offs[0, :, 1] = [0, 4, 3, 6, 5, 2, 7, 0, 0]
        offs[1, :, 0] = [0, 0, 0, 0, 0, 0, 0, 0, 0]
        offs[1, :, 1] = [0, 3, 3, 3, 0, 0, 0, 0, 0]
        tmask = (ids != 1).astype(np.int64)
        tbatch = [["w0", "w1", "w2", "w3"], ["x0", "x1", "x2"]]
        pack = (torch.from_numpy(states), torch.from_numpy(tmask), tbatch, torch.from_numpy(ids), torch.from_numpy(offs))
        jg.get_roberta_embeddings = lambda text: pack
        c_t = jg.forward_inference(text=["w0 w1 w2 w3", "x0 x1 x2"])
        ft = jg.forward_text(torch.from_numpy(states), torch.from_numpy(tmask).unsqueeze(1))
        np.savez_compressed(os.path.join(OUT, "jegal_text.npz"), seed=9004, states=states, ids=ids, offsets=offs,
                            mask=tmask, content=c_t.numpy(), fwd_text=ft.numpy())

        # (v) vta: all three modalities (W must agree between text and audio per clip: 4 and 3 -> audio boundaries)
        wb2 = [[["w0", 2, 6], ["w1", 7, 12], ["w2", 13, 13], ["w3", 15, 30]], [["x0", 1, 4], ["x1", 5, 9], ["x2", 10, 22]]]
        g2, c2 = jg.forward_inference(visual_feats=torch.from_numpy(vf), visual_mask=torch.from_numpy(vm),
                                      text=["w0 w1 w2 w3", "x0 x1 x2"], audio=torch.from_numpy(mel),
                                      audio_mask=am, word_boundaries=wb2)
        np.savez_compressed(os.path.join(OUT, "jegal_vta.npz"), gesture=g2.numpy(), content=c2.numpy())

    # (vi) metric goldens from evaluation/*.py
    er = _import_eval("evaluate_retrieval")
    es = _import_eval("evaluate_spotting")
    ea = _import_eval("evaluate_asd")
    g_emb, c_emb = synth.planted_retrieval(9005, 64)
    g_emb[7] = g_emb[3]                       # duplicate gallery row -> exact tie with the diagonal in rows 3 and 7
    sim = er.get_similarity_matrix(list(c_emb), list(g_emb))
    m = er.compute_metrics(sim.numpy())
    gest, cont, bounds, targets = synth.planted_spotting(9006, 20, n_frames=60, n_words=10, noise=2.0)
    rows = [types.SimpleNamespace(target_word_boundary=str(b[t])) for b, t in zip(bounds, targets)]
    acc = es.get_spotting_acc(rows, gest, cont, [str(b) for b in bounds])
    attn0, _ = es.get_attn_matrix(0, gest, cont, [str(b) for b in bounds])
    asd = {}
    q = torch.from_numpy(c_emb[:1])
    for P in (2, 4, 6):
        asd[f"asd{P}"] = ea.get_similarity_cos(q, torch.from_numpy(g_emb[:P]))
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), sim=sim.numpy(), R5=m["R5"], R10=m["R10"], R25=m["R25"],
                        R50=m["R50"], MR=m["MR"], spot_acc=acc, attn0=attn0, **asd)

    golden_asd(ea)

    # (vii) text-file grammar: the reference's own load_text on its own sample
    ns = _reference_functions(os.path.join(REF, "inference_embs.py"), {"validate_text_file", "preprocess_text", "load_text"})
    import json
    res = {}
    for s in ("sample1", "sample2"):
        text, wbs = ns["load_text"](os.path.join(REF, "samples", s + ".txt"))
        res[s] = {"text": text, "word_boundaries": wbs, "file": open(os.path.join(REF, "samples", s + ".txt"), encoding="utf-8").read()}
    with open(os.path.join(OUT, "load_text.json"), "w") as f:
        json.dump(res, f, indent=1)
    golden_logmel()
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
