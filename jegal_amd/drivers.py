"""Command-line drivers mirroring the reference's hot-path scripts.

    python -m jegal_amd.drivers extract_gestsync_feats ...   # preprocess/extract_gestsync_feats.py
    python -m jegal_amd.drivers extract_jegal_embs ...       # evaluation/extract_jegal_embs.py
    python -m jegal_amd.drivers evaluate_retrieval --path D  # evaluation/evaluate_retrieval.py
    python -m jegal_amd.drivers evaluate_spotting  --path D  # evaluation/evaluate_spotting.py
    python -m jegal_amd.drivers evaluate_asd --path D --file avs_asd.csv   # evaluation/evaluate_asd.py
    python -m jegal_amd.drivers inference_embs ...            # inference_embs.py (single clip -> <fname>.pkl)

Same flags, file naming and on-disk formats as the reference.  What is upstream of the hot path is
NOT rebuilt: video decoding / mediapipe masking (the drivers read already masked 270x480 crops as
``<frames_dir>/<vid>/<track>.npy`` uint8 (T,270,480,3)) and XLM-RoBERTa's tokenizer; audio is read as the reference reads it
(``<video_dir>/<file>.wav`` -> log-mel on the GPU, or a precomputed ``<file>.mel.npy`` (4T,80) fp32); XLM-RoBERTa states come from
``--text_states_dir`` (``<vid>__<track>.npz`` holding states/mask/ids/offsets) or a ``text_encoder`` passed from Python.
Run under ``torchrun`` to shard over GPUs (contiguous blocks, extract_gestsync_feats.py:366-370).
"""
import argparse
import ast
import glob
import math
import os
import pickle
import sys

import numpy as np
import torch

from . import dist as jdist
from . import synth


def load_checkpoint(path, kind):
    """inference_embs.py:92-119: torch.load(path)['state_dict'] with 'module.' stripped.
    path == 'synthetic' gives the seeded synthetic weights (no checkpoints ship with the reference)."""
    if path == "synthetic":
        return synth.gestsync_state_dict() if kind == "gestsync" else synth.jegal_state_dict()
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    return {k.replace("module.", ""): v for k, v in sd.items()}


def _add_precision_args(p):
    p.add_argument("--precision", type=int, default=None, choices=[0, 1, 2, 3, 5, 6],
                   help="engine precision mode (include/jegal_hip.h); default: 5 (run-time corrected fp16: per-clip, calibration-free), "
                        "or 3 (bias-corrected fp16, ~3 %% faster) when --calibrate_frames is given; 6 = the fp32 audit mode (exact-fp32 "
                        "MFMAs, fp32 activations: the reference's CPU arithmetic, ~50x slower)")
    p.add_argument("--calibrate_frames", default=None,
                   help=".npy of masked uint8 crops (B,T,270,480,3) or (T,270,480,3): re-run the precision-mode-3 calibration on them")


#: engine precision mode (include/jegal_hip.h) a driver selects unless calibration clips are supplied: the library's default,
#: calibration-free by construction.  tests/test_gpu_weight_families.py holds THIS mode to 1e-3 on every weight family.
REAL_CHECKPOINT_PRECISION = 5          # PREC_FP16_RC: per-clip run-time correction (round 5; rounds 3-4: 1 = PREC_FP16_W2, 1.4x slower)


def pick_precision(args, checkpoints, can_calibrate=True):
    """Every checkpoint -- the seeded synthetic one included -- runs the library's default, the calibration-free run-time corrected
    mode (E[x] per clip from the clip's own rows, JG_PREC_FP16_RC; held to 1e-3 on every weight family by
    tests/test_gpu_weight_families.py).  The bias-corrected mode (JG_PREC_FP16_BC: the same term folded into the biases once, ~3 %
    faster) is chosen only when the caller supplies calibration clips AND the command can run the
    calibration on them (can_calibrate: it needs the GestSync model, frames -> features -> JEGAL); its built-in calibration on
    synthetic clips was only ever validated on the seeded weights."""
    from ._lib import PREC_FP16_BC
    if getattr(args, "precision", None) is not None:
        return args.precision
    calibrated = bool(getattr(args, "calibrate_frames", None)) and can_calibrate
    return PREC_FP16_BC if calibrated else REAL_CHECKPOINT_PRECISION


def _models(args, need_gestsync=False, need_jegal=False):
    from ._lib import Engine, PREC_FP16_BC
    from .gestsync import GestSync
    from .jegal import JEGAL
    jdist.init_from_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    eng = Engine.get()
    prec = pick_precision(args, [getattr(args, "checkpoint_path_gestsync", None) if need_gestsync else None,
                                 getattr(args, "checkpoint_path", None) if need_jegal else None], can_calibrate=need_gestsync)
    if getattr(args, "calibrate_frames", None) and not need_gestsync:
        print("Note: --calibrate_frames needs the GestSync model and is ignored by this command (precision mode {})".format(prec))
    if eng.finalized == 0:
        eng.set_precision(prec)
    elif (need_gestsync or need_jegal) and eng.precision != prec:
        raise RuntimeError("this process already holds an engine finalized in precision mode {}; mode {} was asked for "
                           "(close the engine or start a new process)".format(eng.precision, prec))
    gs = jg = None
    if need_gestsync:
        gs = GestSync(engine=eng).load_state_dict(load_checkpoint(args.checkpoint_path_gestsync, "gestsync"))
    if need_jegal:
        jg = JEGAL(engine=eng).load_state_dict(load_checkpoint(args.checkpoint_path, "jegal"))
    if getattr(args, "calibrate_frames", None) and prec == PREC_FP16_BC and need_gestsync:
        clips = np.load(args.calibrate_frames)
        eng.calibrate(torch.from_numpy(clips if clips.ndim == 5 else clips[None]).to(eng.device))
    return eng, gs, jg


# --------------------------------------------------------------------------- extract_gestsync_feats
def cmd_extract_gestsync_feats(argv):
    p = argparse.ArgumentParser(prog="extract_gestsync_feats")
    p.add_argument("--checkpoint_path_gestsync", required=True)
    p.add_argument("--frames_dir", required=True, help="<vid>/<track>.npy masked uint8 crops (T,270,480,3)")
    p.add_argument("--result_dir", required=True)
    p.add_argument("--batch_size", type=int, default=48, help="accepted for compatibility; windows are never materialised")
    p.add_argument("--clips_per_batch", type=int, default=16, help="clips of similar length processed per engine call (padded to the longest with its last frame: exact)")
    p.add_argument("--frames_per_batch", type=int, default=4800, help="upper bound on clips x longest clip per engine call (workspace / host memory bound)")
    p.add_argument("--rank", type=int, default=None)
    p.add_argument("--nshard", type=int, default=None)
    _add_precision_args(p)
    args = p.parse_args(argv)
    eng, gs, _ = _models(args, need_gestsync=True)
    files = sorted(glob.glob(os.path.join(args.frames_dir, "*", "*.npy")))
    lo, hi = jdist.shard_range(len(files), args.rank, args.nshard)
    print("Total videos: {} | Start : End = {} : {}".format(len(files), lo, hi))
    saved = err = 0
    todo = []
    for f in files[lo:hi]:
        out = os.path.join(args.result_dir, f.split("/")[-2], os.path.basename(f))
        if os.path.exists(out):                      # resume: skip-if-exists (extract_gestsync_feats.py:281-284)
            saved += 1
            continue
        todo.append((f, out))
    # Clips of different lengths go through the engine in batches: a clip padded with copies of its LAST frame gives, for its own
    # T frames, exactly the windows of the reference's edge padding (inference_embs.py:283 replicates the last frame 12 times;
    # window t only reaches frame t+12) - so a batch is padded to its longest clip, the first T_i rows of clip i are kept, and
    # the GEMMs see several clips at once instead of one.  Files are sorted by length to keep the padding small.  A batch is
    # bounded by clips AND by padded frames (--frames_per_batch): the engine's workspace grows with clips x Tmax (about 2.9 MB per
    # conv position) and the host array with 388 KB per frame, so long tracks go through in small groups or alone, as the
    # reference's one-video-at-a-time loop bounds them (extract_gestsync_feats.py:314-344).
    def clip_len(f):
        try:
            m = np.load(f, mmap_mode="r")
            if m.dtype != np.uint8:
                return -2
            return m.shape[0] if m.ndim == 4 and tuple(m.shape[1:]) == (270, 480, 3) and m.shape[0] > 0 else -1
        except Exception:
            return -1
    sized = sorted(((clip_len(f), f, out) for f, out in todo), key=lambda x: x[0])
    for T_bad, f, _ in [x for x in sized if x[0] <= 0]:
        err += 1
        print("Error: ", "expected uint8 crops" if T_bad == -2 else "expected a (T,270,480,3) uint8 .npy", " | Video file: ", f)
    sized = [x for x in sized if x[0] > 0]

    def run_group(group):
        """Features of a group of clips (padded to its longest); returns the list of (T,1024) arrays."""
        Tmax = max(t for t, _, _ in group)
        batch = np.empty((len(group), Tmax, 270, 480, 3), np.uint8)
        for i, (t, f, _) in enumerate(group):
            fr = np.load(f)
            batch[i, :t] = fr
            batch[i, t:] = fr[t - 1]
        # (the lengths keep each clip's run-time precision correction on its OWN frames: rows < t are those of the clip alone -- bit for bit
        # from 49 frames on, within the contract below: include/jegal_hip.h, jg_gestsync_clip_ragged)
        feats = gs.extract_clip_feats(torch.from_numpy(batch).to(eng.device), lengths=[t for t, _, _ in group]).cpu().numpy()
        return [feats[i, :t] for i, (t, _, _) in enumerate(group)]

    def save_one(feat, out):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        np.save(out, feat)

    nb, fmax = max(1, args.clips_per_batch), max(1, args.frames_per_batch)
    s0 = 0
    while s0 < len(sized):
        s1 = s0 + 1                                   # sorted by length: the last clip of a group is its longest
        while s1 < len(sized) and s1 - s0 < nb and (s1 - s0 + 1) * sized[s1][0] <= fmax:
            s1 += 1
        group = sized[s0:s1]
        s0 = s1
        try:
            feats = run_group(group)
        except Exception as e:                       # one bad file must not take its neighbours down: retry the group clip by clip
            if len(group) == 1:
                err += 1
                print("Error: ", e, " | Video file: ", group[0][1])
                continue
            feats = []
            for item in group:
                try:
                    feats.append(run_group([item])[0])
                except Exception as e1:              # per-file try/except as in the reference (:347-351)
                    feats.append(None)
                    err += 1
                    print("Error: ", e1, " | Video file: ", item[1])
        for feat, (t, f, out) in zip(feats, group):
            if feat is None:
                continue
            try:
                save_one(feat, out)
                saved += 1
            except Exception as e:
                err += 1
                print("Error: ", e, " | Video file: ", f)
    print("No of files saved = {} | Err = {}".format(saved, err))
    return 0


# --------------------------------------------------------------------------- extract_jegal_embs
def _load_tokenizer(name):
    """The reference's module global `tokenizer = AutoTokenizer.from_pretrained("xlm-roberta-base")` (models/jegal.py:13): third-party,
    host side.  `name` is a HuggingFace name or a local directory (no network on the GPU boxes: pass a directory)."""
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(name)


def _load_xlmr(eng, path):
    """`mroberta = XLMRobertaModel.from_pretrained("xlm-roberta-base")` (models/jegal.py:14) on the engine: `path` = a state_dict of
    transformers.XLMRobertaModel (torch.save) or 'synthetic' (seeded test weights, reduced vocabulary)."""
    from .xlmr import XLMRoberta
    sd = synth.xlmr_state_dict() if path == "synthetic" else torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    return XLMRoberta(engine=eng).load_state_dict({k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in sd.items()})


def _load_text_pack(path):
    z = np.load(path, allow_pickle=True)
    return z["states"], z["mask"], z["ids"], z["offsets"]


def cmd_extract_jegal_embs(argv):
    import pandas as pd
    p = argparse.ArgumentParser(prog="extract_jegal_embs")
    p.add_argument("--file_path", required=True)
    p.add_argument("--checkpoint_path", required=True)
    p.add_argument("--res_dir", required=True)
    p.add_argument("--video_dir", required=True)
    p.add_argument("--feature_dir", required=True)
    p.add_argument("--text_states_dir", default=None, help="precomputed XLM-RoBERTa states <vid>__<track>.npz (states, mask, ids, offsets)")
    p.add_argument("--xlmr_checkpoint", default=None, help="state_dict of transformers.XLMRobertaModel (or 'synthetic'): run XLM-RoBERTa on the engine "
                                                           "from the csv's phrases, as models/jegal.py:116-129 does on the CPU; needs --tokenizer")
    p.add_argument("--tokenizer", default=None, help="HuggingFace tokenizer name or local directory (models/jegal.py:13: xlm-roberta-base)")
    p.add_argument("--modalities", default="vta", choices=["vta", "vt", "va", "ta", "v", "t", "a"])
    p.add_argument("--batch_size", type=int, default=16, help="clips per engine call (the reference: 1; results are those of batch size 1 whatever this is)")
    _add_precision_args(p)
    args = p.parse_args(argv)
    eng, _, jg = _models(args, need_jegal=True)
    xlmr = tok = None
    if "t" in args.modalities and args.xlmr_checkpoint:
        if not args.tokenizer:
            raise SystemExit("--xlmr_checkpoint needs --tokenizer (a HuggingFace name or a local tokenizer directory)")
        xlmr, tok = _load_xlmr(eng, args.xlmr_checkpoint), _load_tokenizer(args.tokenizer)
    df = pd.read_csv(args.file_path)
    print("Total files: {}".format(len(df)))
    res_dir = os.path.join(args.res_dir, args.modalities)
    os.makedirs(res_dir, exist_ok=True)
    lo, hi = jdist.shard_range(len(df))
    mod = args.modalities
    saved = 0
    rows = [df.iloc[i] for i in range(lo, hi)]
    for s in range(0, len(rows), args.batch_size):
        batch = []
        for row in rows[s:s + args.batch_size]:
            item = {"row": row, "file": row.filename}
            if "v" in mod:
                fn = os.path.join(args.feature_dir, row.filename + ".npy")
                if not os.path.exists(fn):
                    print("Visual feats file does not exist: ", fn)
                    continue
                item["feats"] = np.load(fn).astype(np.float32)
                if item["feats"].ndim != 2 or item["feats"].shape[1] != 1024:
                    continue
            if "a" in mod:
                # the reference's dataset reads <video_dir>/<file>.wav and computes the log-mel itself (dataset.py:229-235,279-298);
                # a precomputed <file>.mel.npy (4T,80) takes precedence, else the .wav goes through the GPU front end (jg_logmel)
                fn = os.path.join(args.video_dir, row.filename + ".mel.npy")
                wav_fn = os.path.join(args.video_dir, row.filename + ".wav")
                if os.path.exists(fn):
                    item["mel"] = np.load(fn).astype(np.float32)
                elif os.path.exists(wav_fn):
                    try:
                        from . import audio as jaudio
                        wav = torch.from_numpy(np.asarray(jaudio.load_wav(wav_fn)).astype(np.float32))
                        item["mel"] = jaudio.wav2filterbanks(wav[None].to(eng.device), engine=eng)[0][0].cpu().numpy()
                    except Exception:                      # dataset.py:296-298: a wav that cannot be read drops the sample
                        print("Error loading audio, check the file: ", wav_fn)
                        continue
                else:
                    print("Audio file does not exist: ", wav_fn)
                    continue
            if "a" in mod and item["mel"].shape[0] < 4:
                # under 4 mel frames (< 40 ms) the audio CNN has no output step: the reference's per-sample run raises inside its word
                # pooling and the sample is skipped (jegal.py:230-239; the engine's ragged batch call rejects such a clip, and one
                # bad clip must not abort the other fifteen of the batch)
                print("Audio too short ({} mel frames), skipping: {}".format(item["mel"].shape[0], row.filename))
                continue
            if "t" in mod and xlmr is None:
                fn = os.path.join(args.text_states_dir or args.video_dir, row.filename.replace("/", "__") + ".npz")
                if not os.path.exists(fn):
                    print("Text states file does not exist: ", fn)
                    continue
                item["text"] = _load_text_pack(fn)
            batch.append(item)
        if not batch:
            continue
        n = len(batch)
        vis = mask = audio = text = None
        wbs = [ast.literal_eval(it["row"].word_boundaries) for it in batch] if mod != "v" else None
        if "v" in mod:                                # pad_sequence + mask (dataset.py:336-340)
            T = max(it["feats"].shape[0] for it in batch)
            vis = torch.zeros((n, T, 1024))
            mask = torch.zeros((n, T))
            for i, it in enumerate(batch):
                t = it["feats"].shape[0]
                vis[i, :t] = torch.from_numpy(it["feats"])
                mask[i, :t] = 1
        if "a" in mod:
            Tm = max(it["mel"].shape[0] for it in batch)
            audio = torch.zeros((n, Tm, 80))
            for i, it in enumerate(batch):
                audio[i, :it["mel"].shape[0]] = torch.from_numpy(it["mel"])
        if "t" in mod and xlmr is not None:
            # phrases -> tokenizer (host) -> XLM-RoBERTa on the engine: get_roberta_embeddings (models/jegal.py:116-129) for the whole
            # batch; the key padding mask and the position ids from the non-pad tokens keep every clip's rows independent of the padding
            from .xlmr import roberta_embeddings
            text = roberta_embeddings(xlmr, tok, [str(it["row"].phrase) for it in batch])
        elif "t" in mod:
            L = max(it["text"][0].shape[0] for it in batch)
            st = np.zeros((n, L, 768), np.float32)
            tm = np.zeros((n, L), np.int64)
            ids = np.ones((n, L), np.int64)
            offs = np.zeros((n, L, 2), np.int64)
            for i, it in enumerate(batch):
                l = it["text"][0].shape[0]
                st[i, :l], tm[i, :l], ids[i, :l], offs[i, :l] = it["text"]
            tbatch = [str(it["row"].phrase).split(" ") for it in batch]
            text = (torch.from_numpy(st), torch.from_numpy(tm), tbatch, ids, offs)
        am = alens = None
        if audio is not None:
            alens = [it["mel"].shape[0] for it in batch]
            am = torch.zeros((n, eng.audio_len(audio.shape[1])))
            for i, l in enumerate(alens):
                am[i, :l // 4] = 1                    # dataset.py:291 (unused by the model, kept for the signature)
        # per_clip=True: the reference runs this script with batch_size=1 (extract_jegal_embs.py:141), so a clip's result must not
        # depend on the clips it happens to be batched with here -- the last word's text range ends at the clip's own token
        # count and the audio conv stack sees every clip's own zero padding (jegal_amd/jegal.py, jg_jegal_audio_ragged)
        out = jg.forward_inference(visual_feats=vis, visual_mask=mask, text=text, audio=audio, audio_mask=am, word_boundaries=wbs,
                                   audio_lens=alens, per_clip=True)
        gesture = content = None
        if vis is not None and (text is not None or audio is not None):
            gesture, content = out
        elif vis is not None:
            gesture = out
        else:
            content = out
        for i, it in enumerate(batch):
            g = c = None
            if gesture is not None:                   # strip padded query rows
                g = eng.l2norm(gesture[i, :it["feats"].shape[0]]).cpu().numpy()
            if content is not None:                   # strip padded word rows
                c = eng.l2norm(content[i, :len(wbs[i])]).cpu().numpy()
            parts = it["file"].split("/")
            with open(os.path.join(res_dir, parts[0] + "__" + parts[1] + ".pkl"), "wb") as f:
                pickle.dump({"gesture_emb": g, "content_emb": c, "info": it["row"]}, f)
            saved += 1
    print("Saved {} files".format(saved))
    print("Saved JEGAL features at: ", res_dir)
    return 0


# --------------------------------------------------------------------------- evaluators
def _load_pkls(path):
    files = sorted(glob.glob("{}/*.pkl".format(path)))
    print("No of files = ", len(files))
    feats = []
    for fn in files:
        with open(fn, "rb") as f:
            feats.append(pickle.load(f))
    return files, feats


def cmd_evaluate_retrieval(argv):
    from . import metrics as M
    p = argparse.ArgumentParser(prog="evaluate_retrieval")
    p.add_argument("--path", required=True)
    args = p.parse_args(argv)
    eng, _, _ = _models(args)
    _, feats = _load_pkls(args.path)
    lo, hi = jdist.shard_range(len(feats))
    g = M.video_level(eng, [f["gesture_emb"] for f in feats[lo:hi]])
    c = M.video_level(eng, [f["content_emb"] for f in feats[lo:hi]])
    res = {}
    for name, (a, b) in (("Content to Gesture", (c, g)), ("Gesture to Content", (g, c))):
        m = M.retrieval_metrics(a, b, engine=eng)
        res[name] = m
        if jdist.rank() == 0:
            print("{} Retrieval scores:".format(name))
            print("R@1: {:.2f} - R@5: {:.2f} - R@10: {:.2f} - R@25: {:.2f} - R@50: {:.2f} | Median R: {:.1f}".format(
                m["R1"] * 100, m["R5"] * 100, m["R10"] * 100, m["R25"] * 100, m["R50"] * 100, m["MR"]))
    return res


def cmd_evaluate_spotting(argv):
    from . import metrics as M
    p = argparse.ArgumentParser(prog="evaluate_spotting")
    p.add_argument("--path", required=True)
    p.add_argument("--threshold", type=float, default=0.5)
    p.add_argument("--frame_threshold", type=int, default=9)
    args = p.parse_args(argv)
    eng, _, _ = _models(args)
    _, feats = _load_pkls(args.path)
    lo, hi = jdist.shard_range(len(feats))
    feats = feats[lo:hi]
    acc = M.spotting_accuracy([f["gesture_emb"] for f in feats], [f["content_emb"] for f in feats],
                              [f["info"]["word_boundaries"] for f in feats],
                              [f["info"]["target_word_boundary"] if not hasattr(f["info"], "target_word_boundary") else f["info"].target_word_boundary for f in feats],
                              thresh=args.threshold, frame_thresh=args.frame_threshold, engine=eng)
    if jdist.rank() == 0:
        print("Word Spotting Accuracy: {}".format(acc))
    return acc


def cmd_evaluate_asd(argv):
    import pandas as pd
    from . import metrics as M
    p = argparse.ArgumentParser(prog="evaluate_asd")
    p.add_argument("--path", required=True)
    p.add_argument("--file", required=True)
    args = p.parse_args(argv)
    eng, _, _ = _models(args)
    df = pd.read_csv(args.file)
    print("Total files: {}".format(len(df)))

    def load(fname):
        fn = os.path.join(args.path, fname.split("/")[0] + "__" + fname.split("/")[1] + ".pkl")
        if not os.path.exists(fn):
            return None
        with open(fn, "rb") as f:
            return pickle.load(f)

    queries, cands = [], []
    lo, hi = jdist.shard_range(len(df))           # queries sharded over the ranks (contiguous blocks), counters all-reduced
    for i in range(lo, hi):
        row = df.iloc[i]
        q = load(row.filename)
        if q is None:
            continue
        cs = [np.asarray(q["gesture_emb"], np.float32).mean(axis=0)]
        for neg in ast.literal_eval(row.neg_files):
            d = load(neg)
            if d is not None:
                cs.append(np.asarray(d["gesture_emb"], np.float32).mean(axis=0))
        queries.append(np.asarray(q["content_emb"], np.float32).mean(axis=0))
        cands.append(np.stack(cs))
    a2, a4, a6 = M.asd_accuracy(np.stack(queries) if queries else np.zeros((0, 512), np.float32), cands, engine=eng)
    total = M.reduce_counts([len(queries)], eng.device)[0]
    if jdist.rank() == 0:
        print("Total videos evaluated: {}".format(total))
        for k, a in ((2, a2), (4, a4), (6, a6)):
            print("{} spk: Acc: {:.3f}".format(k, a))
    return a2, a4, a6


COMMANDS = {
    "extract_gestsync_feats": cmd_extract_gestsync_feats,
    "extract_jegal_embs": cmd_extract_jegal_embs,
    "evaluate_retrieval": cmd_evaluate_retrieval,
    "evaluate_spotting": cmd_evaluate_spotting,
    "evaluate_asd": cmd_evaluate_asd,
}
COMMANDS["inference_embs"] = lambda argv: cmd_inference_embs(argv)      # (defined below)


# --------------------------------------------------------------------------- inference_embs
def cmd_inference_embs(argv):
    """inference_embs.py (the reference's single-clip driver, :526-646 + the __main__ checks :648-685): one clip in, one
    ``<res_dir>/<fname>.pkl`` = {"gesture_emb" (T,512) | None, "content_emb" (W,512) | None, "info": {fname, word_boundaries, text}} out.

    Upstream of the hot path and NOT rebuilt: video decoding, the mediapipe face mesh and WhisperX.  So --video_path is a ``.npy``:
    the decoded frames (T,H,W,3) uint8 at source resolution with --mask_y (per frame: last blanked source row = chin y2 + 15, or -1
    for "no face", inference_embs.py:264-270; a ``.npy`` of T ints or one int) -- the face rectangle, the cv2-style resize to
    270 x 480, /255 and the +-12 frame edge padding then run on the GPU -- or already masked (T,270,480,3) crops (no --mask_y).
    --audio_path is a 16 kHz mono ``.wav`` (log-mel on the GPU); --text_path the reference's text-file grammar (:288-377); word
    boundaries must come from it (the WhisperX fallback of :589-601 is upstream).  Text needs XLM-RoBERTa states: --xlmr_checkpoint +
    --tokenizer (encoder on the engine, models/jegal.py:116-129) or --text_states (.npz: states, mask, ids, offsets).

    All seven --modalities work (the reference unpacks two return values and indexes word_boundaries[0] unconditionally, so only
    'vta' / 'vt' / 'va' survive there): outputs follow JEGAL.forward_inference's own return convention (:377-415)."""
    from . import audio as jaudio
    from . import extract
    p = argparse.ArgumentParser(prog="inference_embs")
    p.add_argument("--checkpoint_path_gestsync", required=True)
    p.add_argument("--checkpoint_path_jegal", required=True)
    p.add_argument("--modalities", default="vta", choices=["vta", "vt", "va", "ta", "v", "t", "a"])
    p.add_argument("--video_path", default=None, help=".npy of frames: (T,H,W,3) uint8 source frames (with --mask_y) or masked (T,270,480,3) crops")
    p.add_argument("--mask_y", default=None, help=".npy of T ints or one int: last blanked source row per frame (chin y2 + 15; -1 = no face)")
    p.add_argument("--text_path", default=None)
    p.add_argument("--audio_path", default=None)
    p.add_argument("--res_dir", required=True)
    p.add_argument("--text_states", default=None, help="precomputed XLM-RoBERTa states (.npz: states, mask, ids, offsets) for --text_path's words")
    p.add_argument("--xlmr_checkpoint", default=None)
    p.add_argument("--tokenizer", default=None)
    _add_precision_args(p)
    p.add_argument("--audit", action="store_true",
                   help="run the clip a second time in the fp32 audit mode (precision 6, a second engine with the same checkpoints) and print "
                        "the rel-L2 / max-abs of the embeddings above against it: the 1e-3 contract checked on YOUR clip and checkpoint, where "
                        "no CPU reference is at hand (a network that amplifies operand rounding beyond it shows up here; DESIGN.md section 3)")
    args = p.parse_args(argv)
    mod = args.modalities
    for m, arg in (("v", "video_path"), ("a", "audio_path")):                # inference_embs.py:650-664
        if m in mod and getattr(args, arg) is None:
            raise ValueError(f"--{arg} must be specified when modality '{m}' is used.")
    if "t" in mod and args.text_path is None and args.audio_path is None:
        raise ValueError("For modality 't', you must specify either --text_path or --audio_path (since text can be extracted from audio).")
    if mod != "v" and args.text_path is None:
        raise SystemExit("word boundaries come from --text_path here (the reference's WhisperX fallback, inference_embs.py:589-601, is upstream of the hot path)")
    if "t" in mod and not (args.text_states or (args.xlmr_checkpoint and args.tokenizer)):
        raise SystemExit("modality 't' needs --text_states or --xlmr_checkpoint with --tokenizer")
    ns = argparse.Namespace(checkpoint_path_gestsync=args.checkpoint_path_gestsync, checkpoint_path=args.checkpoint_path_jegal,
                            precision=args.precision, calibrate_frames=args.calibrate_frames)
    eng, gs, jg = _models(ns, need_gestsync="v" in mod, need_jegal=True)
    os.makedirs(args.res_dir, exist_ok=True)
    gesture, content, fname, wbs, text_strs = _inference_embs_clip(args, mod, eng, gs, jg, verbose=True)
    if args.audit:
        rep = audit_against_fp32(args, mod, gesture, content)
        print("Audit (fp32 engine, same clip and checkpoints): " + "; ".join(
            f"{k} rel-L2 {v['rel_l2']:.2e} max-abs {v['max_abs']:.2e}" for k, v in rep.items()))
        bad = [k for k, v in rep.items() if not (v["rel_l2"] < 1e-3 and v["max_abs"] < 1e-3)]
        if bad:
            print("WARNING: " + ", ".join(bad) + " exceed(s) the 1e-3 contract in precision mode {}: this checkpoint amplifies fp16 operand "
                  "rounding (DESIGN.md section 3, `sharp` family); use --precision 6 for the embeddings themselves.".format(eng.precision))
    print("------------------------------------------------")
    feat = {"gesture_emb": gesture, "content_emb": content,
            "info": {"fname": fname, "word_boundaries": None if wbs is None else wbs[0], "text": None if text_strs is None else text_strs[0]}}
    output_fname = os.path.join(args.res_dir, fname + ".pkl")
    with open(output_fname, "wb") as f:
        pickle.dump(feat, f)
    print("Saved the embeddings: ", output_fname)
    return 0


def audit_against_fp32(args, mod, gesture, content):
    """--audit: the same clip through a second engine in JG_PREC_FP32 (exact-fp32 MFMAs, fp32 activations end to end: the arithmetic of the
    reference's CPU path, inference_embs.py:497) with the same checkpoints -> {"gesture" | "content": {"rel_l2", "max_abs"}} of the
    embeddings passed in against it."""
    from ._lib import Engine, PREC_FP32
    from .gestsync import GestSync
    from .jegal import JEGAL
    e32 = Engine(torch.cuda.current_device(), precision=PREC_FP32)
    try:
        gs32 = GestSync(engine=e32).load_state_dict(load_checkpoint(args.checkpoint_path_gestsync, "gestsync")) if "v" in mod else None
        jg32 = JEGAL(engine=e32).load_state_dict(load_checkpoint(args.checkpoint_path_jegal, "jegal"))
        g32, c32, _, _, _ = _inference_embs_clip(args, mod, e32, gs32, jg32, verbose=False)
    finally:
        e32.close()
    rep = {}
    for name, a, b in (("gesture", gesture, g32), ("content", content, c32)):
        if a is not None:
            a64, b64 = np.asarray(a, np.float64), np.asarray(b, np.float64)
            rep[name] = {"rel_l2": float(np.linalg.norm(a64 - b64) / max(np.linalg.norm(b64), 1e-30)), "max_abs": float(np.abs(a64 - b64).max())}
    return rep


def _inference_embs_clip(args, mod, eng, gs, jg, verbose=True):
    """frames / wav / text of one clip -> (gesture (T,512) | None, content (W,512) | None, fname, word boundaries, text): extract_embs of
    inference_embs.py:526-646 on `eng`."""
    from . import audio as jaudio
    from . import extract
    say = print if verbose else (lambda *a, **k: None)
    vis = mask = text = audio = am = wbs = None
    fname = None
    text_strs = None
    if args.video_path is not None and "v" in mod:
        frames = np.load(args.video_path)
        if frames.ndim != 4 or frames.shape[-1] != 3 or frames.dtype != np.uint8:
            raise ValueError("--video_path must hold (T,H,W,3) uint8 frames")
        if args.mask_y is not None:
            my = np.load(args.mask_y) if args.mask_y.endswith(".npy") else np.full(frames.shape[0], int(args.mask_y))
            crops = eng.mask_resize(torch.from_numpy(frames), np.asarray(my, np.int32).reshape(-1))
        else:
            if tuple(frames.shape[1:3]) != (270, 480):
                raise ValueError("frames that are not 270x480 need --mask_y (source-resolution path)")
            crops = torch.from_numpy(frames).to(eng.device)
        say("Input masked frames: ", tuple(crops.shape))
        say("Extracting pre-trained GestSync features...")
        vis = gs.extract_clip_feats(crops)                                   # (1,T,1024): get_gestsync_feats (:476-522)
        mask = torch.ones(vis.shape[:2], device=vis.device)
        fname = os.path.basename(args.video_path).split(".")[0]
        say("Input visual features: ", tuple(vis.shape))
    if args.text_path is not None:
        text_strs, wbs = extract.load_text(args.text_path)
        if fname is None:
            fname = os.path.basename(args.text_path).split(".")[0]
    if args.audio_path is not None and "a" in mod:
        say("Loading audio...")
        wav = torch.from_numpy(np.asarray(jaudio.load_wav(args.audio_path)).astype(np.float32))
        audio = jaudio.wav2filterbanks(wav[None].to(eng.device), engine=eng)[0]       # load_audio (:440-475)
        say("Input audio mel: ", tuple(audio.shape))
        am = torch.ones((1, audio.shape[1] // 4), device=eng.device)
    if fname is None and args.audio_path is not None:
        fname = os.path.basename(args.audio_path).split(".")[0]
    if "t" in mod:
        if args.text_states:
            st, tm, ids, offs = _load_text_pack(args.text_states)
            st, tm, ids, offs = (np.asarray(x) for x in (st, tm, ids, offs))
            if st.ndim == 2:
                st, tm, ids, offs = st[None], tm[None], ids[None], offs[None]
            text = (torch.from_numpy(st.astype(np.float32)), torch.from_numpy(tm), [text_strs[0].split(" ")], ids, offs)
        else:
            from .xlmr import roberta_embeddings
            text = roberta_embeddings(_load_xlmr(eng, args.xlmr_checkpoint), _load_tokenizer(args.tokenizer), text_strs)
    say("Extracting JEGAL embeddings...")
    say("------------------------------------------------")
    out = jg.forward_inference(visual_feats=vis, visual_mask=mask, text=text, audio=audio, audio_mask=am,
                               word_boundaries=wbs if mod != "v" else None)
    gesture = content = None
    if vis is not None and (text is not None or audio is not None):
        gesture, content = out
    elif vis is not None:
        gesture = out
    else:
        content = out
    if gesture is not None:
        gesture = eng.l2norm(gesture[0]).cpu().numpy()                       # F.normalize(p=2, dim=-1), [0], .cpu().numpy() (:629-637)
        say("Extracted gesture embeddings: ", gesture.shape)
    if content is not None:
        content = eng.l2norm(content[0]).cpu().numpy()
        say("Extracted content embeddings: ", content.shape)
    return gesture, content, fname, wbs, text_strs


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    from . import want_hw_queues
    want_hw_queues()            # an application may shape its own process: before the first HIP call (no-op when the caller already set it)
    if not argv or argv[0] not in COMMANDS:
        print(__doc__)
        return 2
    res = COMMANDS[argv[0]](argv[1:])
    return 0 if not isinstance(res, int) else res


if __name__ == "__main__":
    sys.exit(main())
