"""Extraction drivers: the build's counterpart of the reference's per-clip / per-dataset callers.

* ``extract_embs`` -- inference_embs.py:526-646 (one clip -> {"gesture_emb","content_emb","info"})
  and evaluation/extract_jegal_embs.py:56-128 (normalise per clip, pickle, file naming).
* ``gestsync_feats_to_npy`` -- preprocess/extract_gestsync_feats.py:314-344 ((T,1024) .npy, skip-if-exists).
Video decoding, mediapipe masking and WhisperX are upstream of the hot path and out of scope; these
drivers start from (T,270,480,3) crops.
"""
import os
import pickle

import numpy as np
import torch

from . import dist as jdist


def load_text(text_path, fps=25):
    """Text-file grammar of inference_embs.py:288-377 ('WORD, START, END, SCORE' after 4 header lines)."""
    import string
    with open(text_path, "r", encoding="utf-8") as f:
        lines = f.readlines()
    head = [l.strip() for l in lines]
    if len(head) < 4:
        raise ValueError(f"{text_path} is too short to be valid.")
    if not head[0].startswith("Text:"):
        raise ValueError("First line must start with 'Text: '")
    if not head[1].startswith("Lang:"):
        raise ValueError("Second line must start with 'Lang: '")
    if head[2] != "":
        raise ValueError("Third line must be empty.")
    if head[3] != "WORD, START, END, SCORE":
        raise ValueError("Fourth line must be 'WORD, START, END, SCORE'")
    rows = lines[4:]
    text, wbs = "", []
    for i, row in enumerate(rows):
        parts = row.split(", ")
        word = "".join(ch for ch in parts[0].lower() if ch not in string.punctuation)
        if word != "":
            text += word
            if i != len(rows) - 1:
                text += " "
            wbs.append([word, round(float(parts[1]) * fps), round(float(parts[2]) * fps)])
    return [text], [wbs]


def extract_embs(gestsync, jegal, frames, text=None, audio=None, word_boundaries=None, modalities="vta", info=None):
    """frames (T,270,480,3) uint8 (masked crops) -> feature dict of inference_embs.py:642 /
    extract_jegal_embs.py:119 with L2-normalised fp32 arrays (the CPU reference saves fp32 too)."""
    eng = jegal.engine
    gesture = content = None
    vis = mask = None
    if "v" in modalities:
        vis = gestsync.extract_clip_feats(frames)                       # (1,T,1024)
        mask = torch.ones(vis.shape[:2], device=vis.device)
    t = text if "t" in modalities else None
    a = audio if "a" in modalities else None
    am = None if a is None else torch.ones((a.shape[0], eng.audio_len(a.shape[1])), device=eng.device)
    out = jegal.forward_inference(visual_feats=vis, visual_mask=mask, text=t, audio=a, audio_mask=am,
                                  word_boundaries=word_boundaries)
    if vis is not None and (t is not None or a is not None):
        gesture, content = out
    elif vis is not None:
        gesture = out
    else:
        content = out
    feat = {"gesture_emb": None if gesture is None else eng.l2norm(gesture[0]).cpu().numpy(),
            "content_emb": None if content is None else eng.l2norm(content[0]).cpu().numpy(),
            "info": info if info is not None else {"word_boundaries": None if word_boundaries is None else word_boundaries[0]}}
    return feat


def pkl_name(res_dir, filename):
    """<res_dir>/<vid>__<track>.pkl (extract_jegal_embs.py:120)."""
    parts = filename.split("/")
    return os.path.join(res_dir, parts[0] + "__" + parts[1] + ".pkl")


def save_pkl(feat, path):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump(feat, f)


def gestsync_feats_to_npy(gestsync, clips, out_paths, rank=None, nshard=None):
    """Sharded (T,1024) .npy dump (extract_gestsync_feats.py:277-284,344,366-370); skip-if-exists."""
    lo, hi = jdist.shard_range(len(clips), rank, nshard)
    done = 0
    for i in range(lo, hi):
        if os.path.exists(out_paths[i]):
            continue
        feats = gestsync.extract_clip_feats(torch.as_tensor(clips[i]).to(gestsync.engine.device))[0]
        os.makedirs(os.path.dirname(out_paths[i]) or ".", exist_ok=True)
        np.save(out_paths[i], feats.cpu().numpy())
        done += 1
    return done
