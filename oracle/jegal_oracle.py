"""CPU oracle for the JEGAL embedding-extraction hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch fp32 restatement (functional PyTorch on CPU + NumPy) of the
reference algorithm, one function per row of SURVEY.md section 8a; every function cites
the reference file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it -- never the product path
(``jegal_amd/``), which must fail loudly if the HIP library is missing.

Pinning: ``oracle/make_golden.py`` (run in the build container, where /root/reference
exists) imports the real ``models.gestsync`` / ``models.jegal`` / ``evaluation.*``
modules, strict-loads the synthetic state_dicts of ``jegal_amd/synth.py`` and stores
their outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this
oracle against those vectors (<= 2e-5 rel).  Third-party arithmetic outside the
reference tree (XLM-RoBERTa, librosa mel filters) is NOT restated: parity unpinned
for those, see DESIGN.md.
"""
import ast
import math
import string

import numpy as np
import torch
import torch.nn.functional as F


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


def tensors(sd):
    return {k: _t(v) for k, v in sd.items()}


# --------------------------------------------------------------------------- GestSync

#: (name, bn, stride, padding, maxpool)  -- gestsync.py:34-87
VID_LAYERS = [
    ("conv1", "bn1", (1, 3, 3), (0, 0, 0), ((1, 3, 3), (1, 2, 2))),
    ("conv2", "bn2", (1, 2, 2), (0, 0, 0), None),
    ("conv3", "bn3", (1, 2, 2), (0, 1, 1), None),
    ("conv4", "bn4", (1, 1, 2), (0, 1, 1), None),
    ("conv5", "bn5", (1, 1, 1), (0, 1, 1), ((1, 3, 3), (1, 2, 2))),
    ("fc6", "bn6", (1, 1, 1), (0, 0, 0), None),
]


def vgg_vid(sd, x, upto=None):
    """VGGNet.forward for net_vid (gestsync.py:308-325): conv -> BN(eval) -> ReLU -> [maxpool].
    x (N,3,F,270,480) -> (N,512,F-4,1,1)."""
    out = x
    for name, bn, stride, pad, mp in VID_LAYERS:
        p = f"net_vid.{name}"
        out = F.conv3d(out, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=pad)
        b = f"net_vid.{bn}"
        out = F.batch_norm(out, sd[b + ".running_mean"], sd[b + ".running_var"], sd[b + ".weight"],
                           sd[b + ".bias"], training=False, eps=1e-5)
        out = F.relu(out)
        if mp is not None:
            out = F.max_pool3d(out, kernel_size=mp[0], stride=mp[1])
        if upto == name:
            return out
    return out


def layer_norm_std(x, w, b, eps=1e-5):
    """nn.LayerNorm: biased variance, eps inside the sqrt."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def mha_packed(x, in_w, in_b, out_w, out_b, nhead):
    """nn.MultiheadAttention self-attention, batch_first, no mask (used by
    nn.TransformerEncoderLayer, gestsync.py:20)."""
    B, S, D = x.shape
    dk = D // nhead
    qkv = F.linear(x, in_w, in_b)
    q, k, v = qkv.split(D, dim=-1)
    q = q.view(B, S, nhead, dk).transpose(1, 2)
    k = k.view(B, S, nhead, dk).transpose(1, 2)
    v = v.view(B, S, nhead, dk).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, v).transpose(1, 2).reshape(B, S, D)
    return F.linear(o, out_w, out_b)


def gestsync_transformer(sd, x):
    """6 x post-norm nn.TransformerEncoderLayer(d=512, nhead=8, ff=2048, relu, eps=1e-5),
    no final norm (gestsync.py:20-21,153)."""
    for l in range(6):
        p = f"transformer_encoder.layers.{l}"
        a = mha_packed(x, sd[p + ".self_attn.in_proj_weight"], sd[p + ".self_attn.in_proj_bias"],
                       sd[p + ".self_attn.out_proj.weight"], sd[p + ".self_attn.out_proj.bias"], 8)
        x = layer_norm_std(x + a, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])
        f = F.linear(F.relu(F.linear(x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                     sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
        x = layer_norm_std(x + f, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])
    return x


def gestsync_head(sd, out_conv):
    """PE + transformer + ff_vid (gestsync.py:152-156): out_conv (N,512,21) -> (N,1024,21)."""
    x = out_conv.transpose(1, 2)
    x = x + sd["pos_encoder.pe"][:, :x.shape[1]]
    x = gestsync_transformer(sd, x)
    x = F.linear(F.relu(F.linear(x, sd["ff_vid.0.weight"], sd["ff_vid.0.bias"])),
                 sd["ff_vid.2.weight"], sd["ff_vid.2.bias"])
    return x.transpose(1, 2)


def gestsync_forward_vid(sd, x, return_feats=False):
    """GestSync.forward_vid (gestsync.py:148-162)."""
    out_conv = vgg_vid(sd, x).squeeze(-1).squeeze(-1)
    out = gestsync_head(sd, out_conv)
    return (out, out_conv) if return_feats else out


def pad_clip(frames01, pad=12):
    """np.pad(..., ((12,12),(0,0),(0,0),(0,0)), 'edge') -- inference_embs.py:283."""
    f = _t(frames01)
    return torch.cat([f[:1].expand(pad, -1, -1, -1), f, f[-1:].expand(pad, -1, -1, -1)], 0)


def gestsync_clip_feats(sd, frames01, naive=False, batch_size=48, num_frames=25, return_conv=False):
    """Per-clip GestSync features (inference_embs.py:476-522 / extract_gestsync_feats.py:314-344):
    frames01 (T,H,W,3) fp32 in [0,1] -> edge-pad 12 -> T windows of 25, stride 1 ->
    forward_vid -> mean over the 21 output steps -> (T,1024).

    naive=True follows the reference literally (every window through the conv stack);
    naive=False runs the conv stack once over the padded clip and slices [i:i+21], which
    is exactly equal (temporal kernels are (5,1,1,1,1,1), no temporal padding).
    return_conv (naive=False only): also return the conv stack's output (512, P-4) (tests: conditioning of what follows it)."""
    padded = pad_clip(frames01)                         # (P,H,W,3)
    P = padded.shape[0]
    n_win = P - num_frames + 1
    vol = padded.permute(3, 0, 1, 2).unsqueeze(0)       # (1,3,P,H,W)
    feats = []
    if naive:
        for s in range(0, n_win, batch_size):
            e = min(n_win, s + batch_size)
            xs = torch.stack([vol[0, :, i:i + num_frames] for i in range(s, e)])
            feats.append(gestsync_forward_vid(sd, xs).mean(-1))
    else:
        chunks = []
        step = 32
        for s in range(0, P - 4, step):                 # conv positions s..s+step-1 need frames s..s+step+3
            e = min(P - 4, s + step)
            chunks.append(vgg_vid(sd, vol[:, :, s:e + 4]).squeeze(-1).squeeze(-1))
        conv = torch.cat(chunks, dim=2)[0]               # (512, P-4)
        for s in range(0, n_win, batch_size):
            e = min(n_win, s + batch_size)
            oc = torch.stack([conv[:, i:i + num_frames - 4] for i in range(s, e)])
            feats.append(gestsync_head(sd, oc).mean(-1))
        if return_conv:
            return torch.cat(feats, 0), conv
    return torch.cat(feats, 0)


def gestsync_feats_from_conv(sd, conv, batch_size=48, num_frames=25):
    """The tail of gestsync_clip_feats(naive=False): conv (512, P-4) -> windows of 21 positions -> gestsync_head -> mean -> (T,1024)."""
    n_win = conv.shape[1] - (num_frames - 4) + 1
    feats = []
    for s in range(0, n_win, batch_size):
        e = min(n_win, s + batch_size)
        oc = torch.stack([conv[:, i:i + num_frames - 4] for i in range(s, e)])
        feats.append(gestsync_head(sd, oc).mean(-1))
    return torch.cat(feats, 0)


# --------------------------------------------------------------------------- JEGAL blocks

def layer_norm_annotated(x, a_2, b_2, eps=1e-6):
    """modules.py:32-35: unbiased std, eps added to the std."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return a_2 * (x - mean) / (std + eps) + b_2


def mha_annotated(sd, p, x, mask, h):
    """MultiHeadedAttention_Transformer.forward + attention() (modules.py:61-120);
    mask (B,1,S) with 0 = padded key -> masked_fill(-1e9)."""
    B, S, D = x.shape
    dk = D // h
    q, k, v = [F.linear(x, sd[f"{p}.linears.{i}.weight"], sd[f"{p}.linears.{i}.bias"])
               .view(B, S, h, dk).transpose(1, 2) for i in range(3)]
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
    if mask is not None:
        scores = scores.masked_fill(mask.unsqueeze(1) == 0, -1e9)
    pa = torch.softmax(scores, dim=-1)
    o = torch.matmul(pa, v).transpose(1, 2).contiguous().view(B, S, D)
    return F.linear(o, sd[f"{p}.linears.3.weight"], sd[f"{p}.linears.3.bias"])


def encoder_annotated(sd, prefix, x, mask, n_layers, h=8):
    """Encoder_Transformer of pre-norm EncoderLayer_Transformer + final LayerNorm
    (modules.py:11-59)."""
    for l in range(n_layers):
        p = f"{prefix}.layers.{l}"
        n = layer_norm_annotated(x, sd[p + ".sublayer.0.norm.a_2"], sd[p + ".sublayer.0.norm.b_2"])
        x = x + mha_annotated(sd, p + ".self_attn", n, mask, h)
        n = layer_norm_annotated(x, sd[p + ".sublayer.1.norm.a_2"], sd[p + ".sublayer.1.norm.b_2"])
        x = x + F.linear(F.relu(F.linear(n, sd[p + ".feed_forward.w_1.weight"], sd[p + ".feed_forward.w_1.bias"])),
                         sd[p + ".feed_forward.w_2.weight"], sd[p + ".feed_forward.w_2.bias"])
    return layer_norm_annotated(x, sd[prefix + ".norm.a_2"], sd[prefix + ".norm.b_2"])


def _mlp2(sd, name, x):
    return F.linear(F.relu(F.linear(x, sd[name + ".0.weight"], sd[name + ".0.bias"])),
                    sd[name + ".2.weight"], sd[name + ".2.bias"])


def jegal_forward_gestures(sd, x, x_mask=None):
    """JEGAL.forward_gestures (jegal.py:78-92): x (B,T,1024), x_mask (B,1,T)."""
    y = F.linear(x, sd["proj_ip_rgb.0.weight"], sd["proj_ip_rgb.0.bias"])
    y = F.relu(layer_norm_std(y, sd["proj_ip_rgb.1.weight"], sd["proj_ip_rgb.1.bias"]))
    y = F.linear(y, sd["proj_ip_rgb.3.weight"], sd["proj_ip_rgb.3.bias"])
    y = y + sd["position_rgb.pe"][:, :y.shape[1]]
    y = encoder_annotated(sd, "encoder_rgb", y, x_mask, 6)
    return F.linear(y, sd["proj_op_rgb.weight"], sd["proj_op_rgb.bias"])


def jegal_forward_text(sd, x, x_mask=None):
    """JEGAL.forward_text (jegal.py:95-103): x (B,L,768), x_mask (B,1,L) -> (B,L,256)."""
    y = encoder_annotated(sd, "encoder_text", x, x_mask, 3)
    return F.linear(y, sd["proj_op_text.weight"], sd["proj_op_text.bias"])


#: (conv idx, stride, padding, has_bn_relu) -- jegal.py:41-63
AUDIO_CNN = [(0, (1, 1), (2, 2), True), (3, (2, 2), (1, 1), True), (6, (2, 2), (1, 1), True),
             (9, (1, 3), (1, 1), True), (12, (1, 3), (1, 1), True), (15, (1, 3), (0, 0), False)]


def jegal_forward_audio(sd, x, x_mask=None):
    """JEGAL.forward_audio (jegal.py:105-113): mel (B,Tm,80) -> (B,Tm//4,256). x_mask unused."""
    y = x.unsqueeze(1)
    for idx, stride, pad, bnrelu in AUDIO_CNN:
        y = F.conv2d(y, sd[f"cnn.{idx}.weight"], sd[f"cnn.{idx}.bias"], stride=stride, padding=pad)
        if bnrelu:
            b = f"cnn.{idx + 1}"
            y = F.batch_norm(y, sd[b + ".running_mean"], sd[b + ".running_var"], sd[b + ".weight"],
                             sd[b + ".bias"], training=False, eps=1e-5)
            y = F.relu(y)
    y = y.squeeze(-1).permute(0, 2, 1)
    return F.linear(y, sd["proj_op_audio.weight"], sd["proj_op_audio.bias"])


SPECIAL_IDS = (0, 2, 1)   # xlm-roberta cls / sep / pad ids (jegal.py:136)


def word_level_text(text_emb, text_batch, input_ids, offset_mapping):
    """get_word_level_embs, text part (jegal.py:131-211).  A word starts at a token whose
    offset[0]==0 and id is not special; the last word's range runs to input_ids.shape[1]
    (so it swallows </s> and pads).  Samples with more words than starts are dropped."""
    out, invalid = [], []
    L = input_ids.shape[1]
    for b in range(input_ids.shape[0]):
        starts = [i for i in range(L)
                  if int(offset_mapping[b][i][0]) == 0 and int(input_ids[b][i]) not in SPECIAL_IDS]
        embs, ok = [], True
        for idx, _w in enumerate(text_batch[b]):
            if idx >= len(starts):
                ok = False
                invalid.append(b)
                break
            end = starts[idx + 1] if idx < len(starts) - 1 else L
            rows = text_emb[b, starts[idx]:end]
            embs.append(rows.mean(dim=0) if len(rows) > 1 else rows[0])
        if ok:
            if len(embs) == 0:
                invalid.append(b)
            else:
                out.append(torch.stack(embs))
    return out, invalid


def word_level_audio(audio_emb, word_boundaries, invalid=None):
    """get_audio_word_level_embs (jegal.py:213-252): mean of audio_emb[b, s-s0 : e-s0+1]."""
    out = []
    for b in range(audio_emb.shape[0]):
        if invalid is not None and b in invalid:
            continue
        s0 = int(word_boundaries[b][0][1])
        embs = []
        for w in word_boundaries[b]:
            s, e = int(w[1]) - s0, int(w[2]) - s0
            rows = audio_emb[b, s:e + 1]
            embs.append(rows.mean(dim=0) if len(rows) > 1 else rows[0])
        if embs:
            out.append(torch.stack(embs))
    return out


def pad_wordlevel(embs):
    """pad_wordlevel_embs (jegal.py:254-272): zero-pad to the longest, lengths list."""
    m = max(e.shape[0] for e in embs)
    return torch.stack([F.pad(e, (0, 0, 0, m - e.shape[0])) for e in embs]), [e.shape[0] for e in embs]


def jegal_forward_inference(sd, visual_feats=None, visual_mask=None, text=None, audio=None,
                            audio_mask=None, word_boundaries=None):
    """JEGAL.forward_inference (jegal.py:377-420).

    ``text`` here is the tuple (text_feats (B,L,768), text_mask (B,L), text_batch
    (list of word lists), input_ids (B,L), offset_mapping (B,L,2)) that
    get_roberta_embeddings (jegal.py:116-129) would return: XLM-R itself is third-party
    and absent, so the oracle starts after it."""
    gesture = None
    if visual_feats is not None:
        gesture = jegal_forward_gestures(sd, visual_feats, visual_mask.unsqueeze(1))
        gesture = _mlp2(sd, "proj_op_align_gesture", gesture)
        if text is None and audio is None:
            return gesture
    text_attn = audio_attn = None
    if text is not None:
        feats, tmask, tbatch, ids, offs = text
        sub = jegal_forward_text(sd, _t(feats), _t(tmask).unsqueeze(1))
        words, _ = word_level_text(sub, tbatch, ids, offs)
        text_attn, _ = pad_wordlevel(words)
        if audio is None:
            audio_attn = torch.zeros_like(text_attn)
    if audio is not None:
        frames = jegal_forward_audio(sd, audio, None)
        words = word_level_audio(frames, word_boundaries)
        audio_attn, _ = pad_wordlevel(words)
        if text is None:
            text_attn = torch.zeros_like(audio_attn)
    fused = torch.cat((audio_attn, text_attn), dim=-1)            # audio first (jegal.py:408)
    content = _mlp2(sd, "proj_op_align_content", _mlp2(sd, "proj_op_fusion_content", fused))
    return content if visual_feats is None else (gesture, content)


def l2_normalize(x, eps=1e-12):
    """F.normalize(p=2, dim=-1) (inference_embs.py:631,635; extract_jegal_embs.py:111,115)."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


# --------------------------------------------------------------------------- metrics

def similarity_matrix(emb1, emb2):
    """get_similarity_matrix (evaluate_retrieval.py:38-48)."""
    a = l2_normalize(_t(np.asarray(emb1, np.float32)))
    b = l2_normalize(_t(np.asarray(emb2, np.float32)))
    return torch.matmul(a, b.t())


def compute_metrics(x):
    """compute_metrics (evaluate_retrieval.py:51-65) + R1 (added)."""
    x = np.asarray(x)
    sx = np.sort(-x, axis=1)
    d = np.diag(-x)[:, np.newaxis]
    ind = np.where((sx - d) == 0)[1]
    n = len(ind)
    return {"R1": float(np.sum(ind < 1)) / n, "R5": float(np.sum(ind < 5)) / n,
            "R10": float(np.sum(ind < 10)) / n, "R25": float(np.sum(ind < 25)) / n,
            "R50": float(np.sum(ind < 50)) / n, "MR": float(np.median(ind) + 1)}


def attn_matrix(gesture, content, temp=0.07):
    """get_attn_matrix (evaluate_spotting.py:39-57): softmax((G C^T)/temp, dim=1)^T -> (W,T)."""
    g = l2_normalize(_t(np.asarray(gesture, np.float32)))
    c = l2_normalize(_t(np.asarray(content, np.float32)))
    a = torch.softmax(torch.mm(g, c.t()) / temp, dim=1)
    return a.numpy().T


def spotting_correct(gesture, content, word_boundaries, target_idx, thresh=0.5, frame_thresh=9):
    """One iteration of get_spotting_acc (evaluate_spotting.py:59-90)."""
    if isinstance(word_boundaries, str):
        word_boundaries = ast.literal_eval(word_boundaries)
    a = attn_matrix(gesture, content)
    _w, s, e = word_boundaries[target_idx]
    pred = int(np.argmax(a[target_idx]))
    score = a[target_idx][pred]
    s = max(0, s - frame_thresh)
    e = e + frame_thresh
    return bool(s <= pred <= e and score >= thresh), pred, float(score)


def spotting_accuracy(gestures, contents, boundaries, targets, thresh=0.5, frame_thresh=9):
    n = sum(spotting_correct(g, c, wb, t, thresh, frame_thresh)[0]
            for g, c, wb, t in zip(gestures, contents, boundaries, targets))
    return 100.0 * n / len(gestures)


def similarity_cos(query_emb, data_emb, temp=0.07):
    """get_similarity_cos (evaluate_asd.py:43-51): cosine (eps 1e-8), /temp, softmax over P."""
    q, d = _t(np.asarray(query_emb, np.float32)), _t(np.asarray(data_emb, np.float32))
    sim = F.cosine_similarity(q, d, dim=1)
    return torch.softmax(sim / temp, dim=0).numpy()


def asd_predictions(query_content, candidates):
    """Inner loop of evaluate_asd (evaluate_asd.py:94-100): query_content (d,) video-level content embedding,
    candidates (P,d) video-level gesture embeddings with the positive first -> argmax index for the first 2 / 4 / 6."""
    q = np.asarray(query_content, np.float32)[None]
    cand = np.asarray(candidates, np.float32)
    return [int(np.argmax(similarity_cos(q, cand[:k]))) for k in (2, 4, 6)]


def asd_accuracy(contents, positives, negatives):
    """evaluate_asd (evaluate_asd.py:54-125) on in-memory clips: temporal means (load_feats :26-39), candidate list
    [positive] + negatives, accuracy = fraction of queries whose argmax is index 0, for 2 / 4 / 6 speakers."""
    correct = np.zeros(3)
    preds = []
    for c, p, negs in zip(contents, positives, negatives):
        cand = np.stack([np.asarray(p, np.float32).mean(axis=0)] + [np.asarray(g, np.float32).mean(axis=0) for g in negs])
        pr = asd_predictions(np.asarray(c, np.float32).mean(axis=0), cand)
        preds.append(pr)
        correct += np.asarray(pr) == 0
    return tuple(float(x) / len(contents) for x in correct), np.asarray(preds, np.int32)


# --------------------------------------------------------------------------- audio front-end

def wav2filterbanks(wav, mel_basis):
    """utils/audio_utils.py:28-66 with torch.stft on the CPU.  `mel_basis` (80,257) must be supplied: the
    reference takes it from librosa.filters.mel, which is third-party and absent (parity unpinned)."""
    wav = _t(wav).float()
    spect = torch.stft(wav, return_complex=True, n_fft=512, hop_length=160, win_length=320,
                       window=torch.hann_window(320), center=True, pad_mode="reflect", normalized=False, onesided=True)
    spect = torch.view_as_real(spect)[:, :, :-1, :]
    mag = torch.norm(spect, dim=-1)
    feats = torch.log(torch.matmul(_t(mel_basis).float(), mag) + 1e-20)
    return feats.permute(0, 2, 1).contiguous()


# --------------------------------------------------------------------------- text file

def preprocess_text(text):
    """inference_embs.py:320-332: lower-case, strip ASCII punctuation."""
    text = text.lower()
    return "".join(ch for ch in text if ch not in string.punctuation)


def load_text(text_path, fps=25):
    """load_text (inference_embs.py:334-377): skip 4 header lines; rows 'WORD, START, END, SCORE'."""
    with open(text_path, "r", encoding="utf-8") as f:
        lines = f.readlines()
    rows = lines[4:]
    text, wbs = "", []
    for i, row in enumerate(rows):
        parts = row.split(", ")
        word = preprocess_text(parts[0])
        if word != "":
            text += word
            if i != len(rows) - 1:
                text += " "
            wbs.append([word, round(float(parts[1]) * fps), round(float(parts[2]) * fps)])
    return [text], [wbs]


# ------------------------------------------------------------------------------------------------
# Face-mask + resize pre-step (inference_embs.py:235-276), SURVEY 8f-4.  PARITY UNPINNED: cv2 is not
# installed in this image; cv2.resize(img, (480, 270)) (INTER_LINEAR, 8-bit) is restated from OpenCV's
# published generic fixed-point algorithm (modules/imgproc/src/resize.cpp), the pip wheels may use IPP.
def _cv_linear_coeffs(dsize, ssize):
    """per destination index: source index (clamped as OpenCV does for x) and the two short coefficients"""
    scale = float(ssize) / float(dsize)
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _sat_short(x):
    return np.clip(np.rint(x.astype(np.float32)), -32768, 32767).astype(np.int64)      # cvRound: half to even


def cv_resize_linear_u8(img, width=480, height=270):
    """img (H,W,C) uint8 -> (height,width,C) uint8."""
    H, W = img.shape[:2]
    sx, fx = _cv_linear_coeffs(width, W)
    lo = sx < 0
    fx[lo] = 0.0; sx[lo] = 0
    hi = sx + 1 >= W
    fx[hi] = 0.0; sx[hi] = W - 1
    a0 = _sat_short((np.float32(1.0) - fx) * np.float32(2048.0)); a1 = _sat_short(fx * np.float32(2048.0))
    x1 = np.minimum(sx + 1, W - 1)
    sy, fy = _cv_linear_coeffs(height, H)
    b0 = _sat_short((np.float32(1.0) - fy) * np.float32(2048.0)); b1 = _sat_short(fy * np.float32(2048.0))
    y0 = np.clip(sy, 0, H - 1); y1 = np.clip(sy + 1, 0, H - 1)
    src = img.astype(np.int64)
    rows = src[:, sx] * a0[None, :, None] + src[:, x1] * a1[None, :, None]              # horizontal pass (H, width, C)
    v = (((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def mask_resize_frames(frames_u8, mask_y, width=480, height=270):
    """inference_embs.py:255-276 per frame: face found (mask_y >= 0): blank source rows 0..mask_y (cv2.rectangle with
    inclusive corners), then resize; face None (mask_y < 0): resize, then blank rows 0..110.  -> (T,270,480,3) uint8
    (the /255 and the +-12 edge pad of :279-283 are applied downstream)."""
    out = np.empty((len(frames_u8), height, width, 3), dtype=np.uint8)
    for i, img in enumerate(frames_u8):
        img = np.array(img, dtype=np.uint8)
        my = int(mask_y[i])
        if my < 0:
            r = cv_resize_linear_u8(img, width, height)
            r[:111] = 0
        else:
            img[:min(my, img.shape[0] - 1) + 1] = 0
            r = cv_resize_linear_u8(img, width, height)
        out[i] = r
    return out


# --------------------------------------------------------------------------- XLM-RoBERTa
def xlmr_forward(sd, input_ids, attention_mask=None, pad_id=1, heads=12, eps=1e-5):
    """``XLMRobertaModel(input_ids, attention_mask=mask).last_hidden_state`` (call site jegal.py:116-129) restated:
    third-party arithmetic (transformers' modeling_xlm_roberta.py: RobertaEmbeddings + 12 post-norm BERT layers, exact GELU).
    ``sd``: the model's state_dict (keys without prefix).  Pinned against transformers itself in tests/golden/xlmr.npz."""
    ids = torch.as_tensor(input_ids).long()
    B, L = ids.shape
    nonpad = (ids != pad_id).long()
    pos_ids = torch.cumsum(nonpad, dim=1) * nonpad + pad_id                      # create_position_ids_from_input_ids
    g = lambda k: torch.as_tensor(sd[k]).float()
    x = g("embeddings.word_embeddings.weight")[ids] + g("embeddings.token_type_embeddings.weight")[0] + \
        g("embeddings.position_embeddings.weight")[pos_ids]
    x = F.layer_norm(x, (x.shape[-1],), g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"), eps)
    D = x.shape[-1]
    dk = D // heads
    bias = None
    if attention_mask is not None:
        m = torch.as_tensor(attention_mask).float()
        bias = (1.0 - m)[:, None, None, :] * torch.finfo(torch.float32).min
    l = 0
    while f"encoder.layer.{l}.attention.self.query.weight" in sd:
        p = f"encoder.layer.{l}"
        lin = lambda name, t: F.linear(t, g(f"{p}.{name}.weight"), g(f"{p}.{name}.bias"))
        q = lin("attention.self.query", x).view(B, L, heads, dk).transpose(1, 2)
        k = lin("attention.self.key", x).view(B, L, heads, dk).transpose(1, 2)
        v = lin("attention.self.value", x).view(B, L, heads, dk).transpose(1, 2)
        sc = q @ k.transpose(-1, -2) / math.sqrt(dk)
        if bias is not None:
            sc = sc + bias
        ctx = (torch.softmax(sc, dim=-1) @ v).transpose(1, 2).reshape(B, L, D)
        x = F.layer_norm(x + lin("attention.output.dense", ctx), (D,), g(f"{p}.attention.output.LayerNorm.weight"),
                         g(f"{p}.attention.output.LayerNorm.bias"), eps)
        h = F.gelu(lin("intermediate.dense", x))
        x = F.layer_norm(x + lin("output.dense", h), (D,), g(f"{p}.output.LayerNorm.weight"), g(f"{p}.output.LayerNorm.bias"), eps)
        l += 1
    return x
