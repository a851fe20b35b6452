"""JEGAL gesture / content encoders -- MI355X engine behind the reference's interface.

Mirrors ``models/jegal.py``: ``forward_gestures`` (:78), ``forward_text`` (:95), ``forward_audio``
(:105), the word-level pooling helpers (:131-272) and ``forward_inference`` (:377-420) with the same
argument meaning and return conventions (gesture only / content only / tuple; NOT normalised).

XLM-RoBERTa (jegal.py:13-14,116-129) is third-party and outside the hot path: pass
``text_encoder`` (a callable with the return convention of ``get_roberta_embeddings``) or hand
``text`` over as that 5-tuple directly.  The integer segment logic runs on the host in Python
(it is index arithmetic over a handful of words); every tensor op runs in libjegal_hip.
"""
import numpy as np
import torch

from ._lib import Engine

SPECIAL_IDS = (0, 2, 1)      # xlm-roberta <s>, </s>, <pad>  (jegal.py:136)


def text_word_segments(input_ids, offset_mapping, text_batch, lengths=None):
    """Row ranges of get_word_level_embs (jegal.py:141-205).  Returns (per-sample list of
    (start,end_excl) or None for dropped samples, invalid indices).

    The reference ends the LAST word's range at ``input_ids.shape[1]`` (jegal.py:168-171), i.e. at the padded length of the
    batch: it swallows ``</s>`` and, in a batch with a longer sentence, the pad rows too.  ``lengths`` (one token count per
    sample, e.g. ``attention_mask.sum(1)``) ends it at the sample's OWN length instead -- what the reference computes when the
    sample is alone in its batch, which is how its dataset driver runs (evaluation/extract_jegal_embs.py:141, batch_size=1)."""
    ids = np.asarray(input_ids.cpu() if isinstance(input_ids, torch.Tensor) else input_ids)
    offs = np.asarray(offset_mapping.cpu() if isinstance(offset_mapping, torch.Tensor) else offset_mapping)
    L = ids.shape[1]
    segs, invalid = [], []
    for b in range(ids.shape[0]):
        Lb = L if lengths is None else int(lengths[b])
        if not 0 < Lb <= L:
            raise ValueError(f"text length {Lb} of sample {b} outside 1..{L}")
        # word starts: offset[0] == 0 and not <s> / </s> / <pad> (jegal.py:148-150); vectorised per sample
        starts = np.flatnonzero((offs[b, :Lb, 0] == 0) & ~np.isin(ids[b, :Lb], SPECIAL_IDS))
        nw = len(text_batch[b])
        ok = nw <= len(starts)
        cur = []
        if ok and nw:
            ends = np.append(starts[1:], Lb)                       # the LAST detected start runs to the end (jegal.py:168-171)
            cur = list(zip(starts[:nw].tolist(), ends[:nw].tolist()))
        if not ok or len(cur) == 0:
            invalid.append(b)
            segs.append(None)
        else:
            segs.append(cur)
    return segs, invalid


def audio_word_segments(word_boundaries, n_frames, invalid=None):
    """Row ranges of get_audio_word_level_embs (jegal.py:218-245): python slice semantics of
    audio_emb[b, s-s0 : e-s0+1]; an empty slice is an IndexError as in the reference (:239).
    n_frames: audio steps of the batch, or one count per sample (the sample's own steps: the slice of a clip alone in its
    batch is clipped to ITS length, not to a longer neighbour's)."""
    segs = []
    per_sample = None if np.isscalar(n_frames) else [int(n) for n in n_frames]
    for b, wbs in enumerate(word_boundaries):
        if per_sample is not None:
            n_frames = per_sample[b]
        if invalid is not None and b in invalid:
            segs.append(None)
            continue
        s0 = int(wbs[0][1])
        cur = []
        for w in wbs:
            s, e = int(w[1]) - s0, int(w[2]) - s0
            lo, hi, _ = slice(s, e + 1).indices(n_frames)
            if hi <= lo:
                raise IndexError("index 0 is out of bounds for dimension 0 with size 0")
            cur.append((lo, hi))
        segs.append(cur)
    return segs


class JEGAL:
    def __init__(self, fusion_strategy="concat", N=6, N_text=3, d_model=512, d_model_text=768, h=8, dropout=0.1,
                 device=None, engine=None, text_encoder=None):
        if (fusion_strategy, N, N_text, d_model, d_model_text, h) != ("concat", 6, 3, 512, 768, 8):
            raise NotImplementedError("only the released JEGAL configuration (jegal.py:18 defaults) is built")
        self.fusion_strategy = fusion_strategy
        self.engine = engine if engine is not None else Engine.get(device)
        self.text_encoder = text_encoder
        self._loaded = False

    def cuda(self, device=None):
        return self

    def eval(self):
        return self

    def load_state_dict(self, state_dict, strict=True):
        sd = {k.replace("module.", ""): v for k, v in state_dict.items()}
        self.engine.load_tensors(sd)
        self.engine.finalize(2)
        self._loaded = True
        return self

    def _check(self):
        if not self._loaded:
            raise RuntimeError("JEGAL: call load_state_dict() first")

    # ---- branch entry points (same signatures as the reference)
    def forward_gestures(self, x, x_mask=None):
        self._check()
        m = None if x_mask is None else x_mask.reshape(x.shape[0], x.shape[1])
        return self.engine.jegal_gestures(x, m, align=False)

    def forward_text(self, x, x_mask=None):
        self._check()
        m = None if x_mask is None else x_mask.reshape(x.shape[0], x.shape[1])
        return self.engine.jegal_text(x, m)

    def forward_audio(self, x, x_mask=None):
        self._check()
        return self.engine.jegal_audio(x)     # x_mask is unused by the reference too (jegal.py:105)

    def get_roberta_embeddings(self, text):
        if self.text_encoder is None:
            raise RuntimeError("no text_encoder configured: XLM-RoBERTa is third-party (jegal.py:13-14); pass "
                               "JEGAL(text_encoder=...) or give `text` as (text_emb, text_mask, text_batch, input_ids, offset_mapping)")
        return self.text_encoder(text)

    def _pool(self, seq, segs, col, fused, rows_of):
        """seq (B,S,256) device; segs per-sample list of ranges; writes fused[row, w, col:col+256]."""
        B, S, D = seq.shape
        W = fused.shape[1]
        trip = []
        for out_b, b in enumerate(rows_of):
            for w, (lo, hi) in enumerate(segs[b]):
                trip.append((b * S + lo, b * S + hi, out_b * W + w))
        self.engine.word_pool(seq.reshape(B * S, D), np.asarray(trip, np.int32), fused.reshape(-1, 512), col)

    def get_word_level_embs(self, text_emb, text, input_ids, offset_mapping, audio_emb=None, word_boundaries=None):
        """List-of-tensors form of jegal.py:131-211 (kept for drop-in callers)."""
        segs, invalid = text_word_segments(input_ids, offset_mapping, text)
        valid = [b for b in range(len(segs)) if segs[b] is not None]
        out_t, out_a = [], []
        if valid:
            W = max(len(segs[b]) for b in valid)
            buf = torch.zeros((len(valid), W, 512), dtype=torch.float32, device=self.engine.device)
            self._pool(text_emb, segs, 256, buf, valid)
            if audio_emb is not None:
                asegs = audio_word_segments([[wb for wb in word_boundaries[b]][:len(text[b])] for b in range(len(segs))],
                                            audio_emb.shape[1])
                self._pool(audio_emb, asegs, 0, buf, valid)
            for i, b in enumerate(valid):
                out_t.append(buf[i, :len(segs[b]), 256:])
                if audio_emb is not None:
                    out_a.append(buf[i, :len(segs[b]), :256])
        return out_t, out_a, invalid

    def get_audio_word_level_embs(self, audio_emb, word_boundaries, invalid_sample_idx=None):
        segs = audio_word_segments(word_boundaries, audio_emb.shape[1], invalid_sample_idx)
        valid = [b for b in range(len(segs)) if segs[b] is not None]
        W = max(len(segs[b]) for b in valid)
        buf = torch.zeros((len(valid), W, 512), dtype=torch.float32, device=self.engine.device)
        self._pool(audio_emb, segs, 0, buf, valid)
        return [buf[i, :len(segs[b]), :256] for i, b in enumerate(valid)], invalid_sample_idx

    def pad_wordlevel_embs(self, wordlevel_embs):
        m = max(e.shape[0] for e in wordlevel_embs)
        padded = torch.stack([torch.nn.functional.pad(e, (0, 0, 0, m - e.shape[0])) for e in wordlevel_embs])
        return padded, [e.shape[0] for e in wordlevel_embs]

    # ---- the inference entry point (jegal.py:377-420)
    def forward_inference(self, visual_feats=None, visual_mask=None, text=None, audio=None, audio_mask=None,
                          word_boundaries=None, audio_lens=None, per_clip=False):
        """jegal.py:377-420.  Default: the reference's semantics for the batch as given -- including its dependence on the
        batch's padded lengths (the last word's text range runs to the padded L, jegal.py:168-171; the audio conv stack sees a
        shorter clip's zero padding as data, jegal.py:41-63).

        per_clip=True (not in the reference's signature): every clip of a padded batch gets the result it would get ALONE in
        its batch, which is how the reference's dataset driver runs (evaluation/extract_jegal_embs.py:141, batch_size=1):
        text lengths from ``text_mask.sum(1)``, audio lengths from ``audio_lens`` (mel frames per clip, required with audio),
        gesture rows are batch-independent already (key mask).  Rows beyond a clip's own T / W stay for the caller to strip."""
        self._check()
        eng = self.engine
        if per_clip and audio is not None and audio_lens is None:
            raise ValueError("per_clip=True needs audio_lens (mel frames of every clip) when audio is given")
        gesture = None
        if visual_feats is not None:
            gesture = eng.jegal_gestures(visual_feats, visual_mask, align=True)
            if text is None and audio is None:
                return gesture
        t_segs = a_segs = None
        t_rows = a_rows = None
        if text is not None:
            pack = text if isinstance(text, tuple) else self.get_roberta_embeddings(text)
            text_feats, text_mask, text_batch, input_ids, offset_mapping = pack
            sub = eng.jegal_text(text_feats, text_mask)
            t_len = None
            if per_clip:
                tm = text_mask.cpu().numpy() if isinstance(text_mask, torch.Tensor) else np.asarray(text_mask)
                t_len = (tm.reshape(len(text_batch), -1) != 0).sum(1)
            t_segs, _invalid = text_word_segments(input_ids, offset_mapping, text_batch, t_len)
            t_rows = [b for b in range(len(t_segs)) if t_segs[b] is not None]
            if not t_rows:
                raise ValueError("max() arg is an empty sequence")       # pad_wordlevel_embs on an empty list
        if audio is not None:
            frames = eng.jegal_audio(audio, audio_lens if per_clip else None)
            a_segs = audio_word_segments(word_boundaries, [eng.audio_len(n) for n in audio_lens] if per_clip else frames.shape[1])
            a_rows = list(range(len(a_segs)))
        rows = t_rows if t_rows is not None else a_rows
        Wt = max(len(t_segs[b]) for b in t_rows) if t_rows is not None else None
        Wa = max(len(a_segs[b]) for b in a_rows) if a_rows is not None else None
        if Wt is not None and Wa is not None and (Wt != Wa or len(t_rows) != len(a_rows)):
            raise RuntimeError(f"Sizes of tensors must match except in dimension 2 (audio {len(a_rows)}x{Wa}, text {len(t_rows)}x{Wt})")
        W = Wt if Wt is not None else Wa
        fused = torch.zeros((len(rows), W, 512), dtype=torch.float32, device=eng.device)
        if a_segs is not None:
            self._pool(frames, a_segs, 0, fused, a_rows)          # audio first (jegal.py:408)
        if t_segs is not None:
            self._pool(sub, t_segs, 256, fused, t_rows)
        content = eng.fuse_content(fused)
        return content if visual_feats is None else (gesture, content)

    __call__ = forward_inference
