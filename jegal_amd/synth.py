"""Deterministic synthetic weights and inputs.

The reference ships no checkpoints, so parity and the
benchmark run on seeded synthetic ``state_dict``s that carry exactly the keys and
shapes of ``models/gestsync.py`` / ``models/jegal.py`` (strict-loadable into the
reference modules; ``oracle/make_golden.py`` checks that).  Everything is generated
from ``numpy.random.default_rng(seed)`` so the GPU box reproduces it bit-for-bit.

Inputs follow SURVEY.md section 8d (configs 2-5).
"""
import math

import numpy as np

GESTSYNC_SEED = 4101
JEGAL_SEED = 4102


def sinusoid_pe(max_len, d_model):
    """Sin/cos table of gestsync.py:178-183 and modules.py:140-147 (fp32 math as torch does it)."""
    import torch
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0).numpy()


FAMILIES = ("gauss", "heavy", "sharp", "sharp2")


def _relu_moments(mu, sigma):
    """E[relu(z)], E[relu(z)^2] for z ~ N(mu, sigma^2), per element."""
    from math import erf, exp, pi, sqrt
    mu = np.asarray(mu, np.float64)
    sigma = np.maximum(np.asarray(sigma, np.float64), 1e-12)
    r = mu / sigma
    Phi = 0.5 * (1.0 + np.vectorize(erf)(r / sqrt(2.0)))
    phi = np.exp(-0.5 * r * r) / sqrt(2.0 * pi)
    return mu * Phi + sigma * phi, (mu * mu + sigma * sigma) * Phi + mu * sigma * phi


class _Gen:
    """Seeded weight generator.  ``family`` (every parity number of rounds 1-4 came from ONE draw of
    He-initialised Gaussian weights, while the reference runs trained checkpoints, inference_embs.py:92-119):

    "gauss"  the original draw (bit-identical to rounds 1-4 for the default seeds; the goldens depend on it)
    "heavy"  Student-t (nu = 3, rescaled to the Gaussian family's variance) weights -- outliers of tens of sigma; BatchNorm
             running_var log-uniform in [1e-2, 1e1] and BatchNorm / LayerNorm gamma log-uniform in [0.1, 4].  The conv weights in
             front of a BatchNorm are rescaled per output channel so that the pre-BN variance is about running_var (a trained
             net's statistics are self-consistent: without that the activations leave fp16's range after two layers) and
             running_mean sits near the pre-BN mean.
    "sharp"  the Gaussian draw with the q and k projections of every attention x 4 (logits x 16: near one-hot softmax).  This net is
             ILL-CONDITIONED: in float64 a relative input perturbation of 1e-6 moves the JEGAL gesture embedding by 2.4e-5 (x 24; the
             Gaussian draw: x 0.4), so 1e-3 at the output needs 4e-5 at every activation -- beyond any 16-bit operand format, the
             reference's own CUDA autocast included.  Reported, not asserted (tests/test_gpu_weight_families.py measures the factor).
    "sharp2" q and k x 2 (logits x 4): sharp attention that 16-bit operands can still follow
    """

    def __init__(self, seed, family="gauss"):
        if family not in FAMILIES:
            raise ValueError(f"unknown weight family {family!r} (one of {FAMILIES})")
        self.rng = np.random.default_rng(seed)
        self.sd = {}
        self.family = family
        self.in_m1, self.in_m2 = 0.5, 1.0 / 3.0      # first / second moment of the next conv's input (heavy family bookkeeping)

    def _randn(self, shape):
        if self.family == "heavy":
            return self.rng.standard_t(3.0, shape) / math.sqrt(3.0)
        return self.rng.standard_normal(shape)

    def _gamma(self, c, lo, hi):
        if self.family == "heavy":
            return np.exp(self.rng.uniform(math.log(0.1), math.log(4.0), c)).astype(np.float32)
        return self.rng.uniform(lo, hi, c).astype(np.float32)

    def linear(self, name, out_f, in_f, relu_after=False, wname="weight", bname="bias"):
        std = math.sqrt((2.0 if relu_after else 1.0) / in_f)
        self.sd[f"{name}.{wname}"] = (self._randn((out_f, in_f)) * std).astype(np.float32)
        self.sd[f"{name}.{bname}"] = (self.rng.standard_normal(out_f) * 0.05).astype(np.float32)

    def conv(self, name, out_c, in_c, ks):
        fan_in = in_c * int(np.prod(ks))
        std = math.sqrt(2.0 / fan_in)
        self.sd[f"{name}.weight"] = (self._randn((out_c, in_c) + tuple(ks)) * std).astype(np.float32)
        self.sd[f"{name}.bias"] = (self.rng.standard_normal(out_c) * 0.05).astype(np.float32)
        self._last_conv = name

    def bn(self, name, c, pooled=False):
        if self.family != "heavy":
            self.sd[f"{name}.weight"] = self.rng.uniform(0.6, 1.4, c).astype(np.float32)
            self.sd[f"{name}.bias"] = (self.rng.standard_normal(c) * 0.1).astype(np.float32)
            self.sd[f"{name}.running_mean"] = (self.rng.standard_normal(c) * 0.1).astype(np.float32)
            self.sd[f"{name}.running_var"] = self.rng.uniform(0.5, 1.5, c).astype(np.float32)
            self.sd[f"{name}.num_batches_tracked"] = np.array(1000, dtype=np.int64)
            return
        gamma = self._gamma(c, 0.6, 1.4)
        beta = (self.rng.standard_normal(c) * 0.1).astype(np.float32)
        rv = np.exp(self.rng.uniform(math.log(1e-2), math.log(1e1), c)).astype(np.float32)
        # rescale the conv in front: He weights give a pre-BN variance of 2 * E[x^2]; make it running_var per output channel
        w = self.sd[f"{self._last_conv}.weight"]
        w *= np.sqrt(rv / (2.0 * self.in_m2)).astype(np.float32).reshape((c,) + (1,) * (w.ndim - 1))
        pre_mean = self.sd[f"{self._last_conv}.bias"] + self.in_m1 * w.reshape(c, -1).sum(1)
        off = (self.rng.standard_normal(c) * 0.3).astype(np.float32)                 # BN input mean - running_mean, in sigmas
        self.sd[f"{name}.weight"] = gamma
        self.sd[f"{name}.bias"] = beta
        self.sd[f"{name}.running_mean"] = (pre_mean - off * np.sqrt(rv)).astype(np.float32)
        self.sd[f"{name}.running_var"] = rv
        self.sd[f"{name}.num_batches_tracked"] = np.array(1000, dtype=np.int64)
        m1, m2 = _relu_moments(beta + gamma * off, gamma)
        k = 1.6 if pooled else 1.0                   # a 3x3 max-pool behind the ReLU lifts the moments (rough: keeps the next layer in range)
        self.in_m1, self.in_m2 = float(m1.mean()) * k, float(m2.mean()) * k * k

    def ln(self, name, c, wname="weight", bname="bias"):
        self.sd[f"{name}.{wname}"] = self._gamma(c, 0.7, 1.3)
        self.sd[f"{name}.{bname}"] = (self.rng.standard_normal(c) * 0.1).astype(np.float32)

    def sharpen(self, wkey, bkey, rows=None):
        """'sharp' family: q / k projection (rows of a packed in_proj, or a whole Linear) x 4."""
        if self.family not in ("sharp", "sharp2"):
            return
        sl = slice(None) if rows is None else slice(0, rows)
        f = np.float32(4.0 if self.family == "sharp" else 2.0)
        self.sd[wkey][sl] *= f
        self.sd[bkey][sl] *= f


def gestsync_state_dict(seed=GESTSYNC_SEED, include_unused=True, family="gauss"):
    """All keys of ``GestSync().state_dict()`` (gestsync.py:9-32).

    ``include_unused`` adds the audio/LSTM tensors the checkpoint carries but
    ``forward_vid`` never touches (checkpoint row).  ``family``: see _Gen.
    """
    g = _Gen(seed, family)
    vid = [("conv1", 64, 3, (5, 7, 7)), ("conv2", 128, 64, (1, 5, 5)), ("conv3", 256, 128, (1, 3, 3)),
           ("conv4", 256, 256, (1, 3, 3)), ("conv5", 256, 256, (1, 3, 3)), ("fc6", 512, 256, (1, 4, 4))]
    for i, (nm, oc, ic, ks) in enumerate(vid, 1):
        g.conv(f"net_vid.{nm}", oc, ic, ks)
        g.bn(f"net_vid.bn{i}", oc, pooled=nm in ("conv1", "conv5"))
    g.linear("ff_vid.0", 512, 512, relu_after=True)
    g.linear("ff_vid.2", 1024, 512)
    g.sd["pos_encoder.pe"] = sinusoid_pe(50, 512)
    for l in range(6):
        p = f"transformer_encoder.layers.{l}"
        g.linear(f"{p}.self_attn", 1536, 512, wname="in_proj_weight", bname="in_proj_bias")
        g.sharpen(f"{p}.self_attn.in_proj_weight", f"{p}.self_attn.in_proj_bias", rows=1024)
        g.linear(f"{p}.self_attn.out_proj", 512, 512)
        g.linear(f"{p}.linear1", 2048, 512, relu_after=True)
        g.linear(f"{p}.linear2", 512, 2048)
        g.ln(f"{p}.norm1", 512)
        g.ln(f"{p}.norm2", 512)
    if include_unused:
        aud = [("conv1", 64, 1, (3, 3)), ("conv2", 192, 64, (3, 3)), ("conv3", 384, 192, (3, 3)),
               ("conv4", 256, 384, (3, 3)), ("conv5", 256, 256, (3, 3)), ("fc6", 512, 256, (4, 2))]
        for i, (nm, oc, ic, ks) in enumerate(aud, 1):
            g.conv(f"net_aud.{nm}", oc, ic, ks)
            g.bn(f"net_aud.bn{i}", oc)
        for sfx in ("", "_reverse"):
            g.sd[f"lstm.weight_ih_l0{sfx}"] = (g.rng.standard_normal((1024, 512)) * 0.04).astype(np.float32)
            g.sd[f"lstm.weight_hh_l0{sfx}"] = (g.rng.standard_normal((1024, 256)) * 0.04).astype(np.float32)
            g.sd[f"lstm.bias_ih_l0{sfx}"] = np.zeros(1024, np.float32)
            g.sd[f"lstm.bias_hh_l0{sfx}"] = np.zeros(1024, np.float32)
        g.conv("ff_aud.fc7", 512, 512, (1, 1))
        g.bn("ff_aud.bn7", 512)
        g.conv("ff_aud.fc8", 1024, 512, (1, 1))
        g.sd["logits_scale.weight"] = np.ones((1, 1), np.float32)
        g.sd["fc.weight"] = np.ones((1, 1), np.float32)
        g.sd["fc.bias"] = np.zeros(1, np.float32)
    return g.sd


def _annotated_encoder(g, prefix, n_layers, d, d_ff):
    for l in range(n_layers):
        p = f"{prefix}.layers.{l}"
        for i in range(4):
            g.linear(f"{p}.self_attn.linears.{i}", d, d)
            if i < 2:
                g.sharpen(f"{p}.self_attn.linears.{i}.weight", f"{p}.self_attn.linears.{i}.bias")
        g.linear(f"{p}.feed_forward.w_1", d_ff, d, relu_after=True)
        g.linear(f"{p}.feed_forward.w_2", d, d_ff)
        g.ln(f"{p}.sublayer.0.norm", d, "a_2", "b_2")
        g.ln(f"{p}.sublayer.1.norm", d, "a_2", "b_2")
    g.ln(f"{prefix}.norm", d, "a_2", "b_2")


def jegal_state_dict(seed=JEGAL_SEED, family="gauss"):
    """All keys of ``JEGAL().state_dict()`` (jegal.py:18-76).  ``family``: see _Gen."""
    g = _Gen(seed, family)
    g.linear("proj_ip_rgb.0", 512, 1024)
    g.ln("proj_ip_rgb.1", 512)
    g.linear("proj_ip_rgb.3", 512, 512)
    g.sd["position_rgb.pe"] = sinusoid_pe(500, 512)
    _annotated_encoder(g, "encoder_rgb", 6, 512, 2048)
    g.linear("proj_op_rgb", 512, 512)
    _annotated_encoder(g, "encoder_text", 3, 768, 3072)
    g.linear("proj_op_text", 256, 768)
    cnn = [(0, 32, 1, (5, 5)), (3, 64, 32, (3, 3)), (6, 128, 64, (3, 3)), (9, 256, 128, (3, 3)),
           (12, 256, 256, (3, 3)), (15, 256, 256, (1, 1))]
    g.in_m1, g.in_m2 = 8.0, 8.0 * 8.0 + 2.5 * 2.5          # log-mel input ~ N(8, 2.5) (heavy family bookkeeping)
    for idx, oc, ic, ks in cnn:
        g.conv(f"cnn.{idx}", oc, ic, ks)
        if idx != 15:
            g.bn(f"cnn.{idx + 1}", oc)
    g.linear("proj_op_audio", 256, 256)
    for nm in ("proj_op_fusion_content", "proj_op_align_gesture", "proj_op_align_content"):
        g.linear(f"{nm}.0", 512, 512, relu_after=True)
        g.linear(f"{nm}.2", 512, 512)
    return g.sd


XLMR_SEED = 4242


def xlmr_state_dict(seed=XLMR_SEED, vocab=1000, layers=4, max_pos=514, family="gauss"):
    """Keys and shapes of ``transformers.XLMRobertaModel(...).state_dict()`` for the xlm-roberta-base architecture (hidden 768,
    12 heads, intermediate 3072, type vocabulary 1) with a reduced vocabulary / depth: seeded random stand-ins for the released
    checkpoint, which is not available offline (jegal.py:13-14).  Weight scales keep the activations O(1) through the layers."""
    g = _Gen(seed, family)
    D, DFF = 768, 3072
    g.sd["embeddings.word_embeddings.weight"] = (g.rng.standard_normal((vocab, D)) * 0.5).astype(np.float32)
    g.sd["embeddings.position_embeddings.weight"] = (g.rng.standard_normal((max_pos, D)) * 0.3).astype(np.float32)
    g.sd["embeddings.token_type_embeddings.weight"] = (g.rng.standard_normal((1, D)) * 0.1).astype(np.float32)
    g.ln("embeddings.LayerNorm", D)
    for l in range(layers):
        p = f"encoder.layer.{l}"
        for nm in ("query", "key", "value"):
            g.linear(f"{p}.attention.self.{nm}", D, D)
            if nm != "value":
                g.sharpen(f"{p}.attention.self.{nm}.weight", f"{p}.attention.self.{nm}.bias")
        g.linear(f"{p}.attention.output.dense", D, D)
        g.ln(f"{p}.attention.output.LayerNorm", D)
        g.linear(f"{p}.intermediate.dense", DFF, D, relu_after=True)
        g.linear(f"{p}.output.dense", D, DFF)
        g.ln(f"{p}.output.LayerNorm", D)
    return g.sd


def xlmr_inputs(seed, batch, length, vocab=1000):
    """(input_ids, attention_mask) int32 (B, L) in the tokenizer's convention: <s> = 0 first, </s> = 2 last, <pad> = 1 behind it;
    sample 0 has no padding, the others are shorter."""
    rng = np.random.default_rng(seed)
    ids = np.full((batch, length), 1, dtype=np.int32)
    mask = np.zeros((batch, length), dtype=np.int32)
    for b in range(batch):
        n = length if b == 0 else int(rng.integers(max(3, length // 3), length + 1))
        ids[b, 0] = 0
        ids[b, 1:n - 1] = rng.integers(3, vocab, n - 2)
        ids[b, n - 1] = 2
        mask[b, :n] = 1
    return ids, mask


# --------------------------------------------------------------------------- inputs

def synth_frames(seed, n_clips, n_frames, height=270, width=480, mask_rows=110, mask_jitter=None):
    """uint8 (B,T,H,W,3) uniform-noise clips, rows [0,mask_rows) zeroed like the face-mask rectangle of
    inference_embs.py:264.  mask_jitter = (lo, hi): the mask height is drawn PER FRAME from lo..hi
    instead (the reference blanks rows 0..y2+15 with y2 following the chin, inference_embs.py:264-270); the pixels below
    the mask are the same as without jitter."""
    rng = np.random.default_rng(seed)
    out = np.empty((n_clips, n_frames, height, width, 3), np.uint8)
    for b in range(n_clips):
        out[b] = rng.integers(0, 256, (n_frames, height, width, 3), dtype=np.uint8)
    if mask_jitter is None:
        out[:, :, :mask_rows] = 0
    else:
        hts = mask_heights(seed, n_clips, n_frames, *mask_jitter)
        for b in range(n_clips):
            for t in range(n_frames):
                out[b, t, :hts[b, t]] = 0
    return out


def synth_frames_structured(seed, n_clips, n_frames, kind, height=270, width=480, mask_rows=110):
    """Clips that are NOT uniform noise (the reference's inputs are natural crops, inference_embs.py:235-286;
    every oracle comparison of rounds 1-3 used synth_frames).  uint8 (B,T,H,W,3), seeded, three families:

    "smooth"     low-contrast smooth content: two slow sinusoidal gradients per channel plus three Gaussian blobs that drift a
                 few pixels per frame (a torso / arms stand-in), values ~ 60..190, +-2 of pixel noise; rows [0, mask_rows) zero
    "saturated"  hard-edged blocks of 0 / 255 per channel (cells of 24 x 32 pixels, the pattern shifts by 3 pixels per frame) --
                 the largest conv1 sums the u8 path can see; rows [0, mask_rows) zero
    "jitter"     the smooth content with the mask height drawn PER FRAME from 80..140, frame 3 of every clip NOT masked at all
                 and frame 5 masked completely (inference_embs.py:264-270 blanks rows 0..y2+15 per frame; no face = no mask)
    "periodic"   the smooth content under a strong temporal brightness modulation whose periods are the ones the run-time corrected
                 mode's row sample could alias with: 128 / 21 = 6.095 frames (the sampled 16-row runs start every
                 128 token rows = every 6.095 windows), 21 frames and 128 frames, plus a slow horizontal pan; rows [0, mask_rows) zero
    """
    rng = np.random.default_rng([seed, 0x57A7])
    yy, xx = np.meshgrid(np.arange(height, dtype=np.float32), np.arange(width, dtype=np.float32), indexing="ij")
    out = np.empty((n_clips, n_frames, height, width, 3), np.uint8)
    for b in range(n_clips):
        if kind in ("smooth", "jitter", "periodic"):
            ph = rng.uniform(0, 2 * np.pi, (3, 2))
            fx, fy = rng.uniform(0.6, 2.2, 3), rng.uniform(0.6, 2.2, 3)
            blob = rng.uniform([60, 120, 25, 18], [420, 250, 70, 45], (3, 4))          # x0, y0, sx, sy
            vel = rng.uniform(-2.5, 2.5, (3, 2))
            amp = rng.uniform(25, 45, 3)
            for t in range(n_frames):
                img = np.empty((height, width, 3), np.float32)
                bl = np.zeros((height, width), np.float32)
                for k in range(3):
                    cx, cy = blob[k, 0] + vel[k, 0] * t, blob[k, 1] + vel[k, 1] * t
                    bl += amp[k] * np.exp(-0.5 * (((xx - cx) / blob[k, 2]) ** 2 + ((yy - cy) / blob[k, 3]) ** 2))
                for c in range(3):
                    img[..., c] = (118.0 + 8.0 * c + 22.0 * np.sin(2 * np.pi * fx[c] * xx / width + ph[c, 0] + 0.03 * t)
                                   + 16.0 * np.cos(2 * np.pi * fy[c] * yy / height + ph[c, 1] - 0.02 * t) + bl * (1.0 - 0.15 * c))
                if kind == "periodic":
                    gain = 1.0 + 0.30 * np.sin(2 * np.pi * t * 21.0 / 128.0 + ph[0, 0]) + 0.20 * np.sin(2 * np.pi * t / 21.0 + ph[1, 0]) \
                        + 0.15 * np.sin(2 * np.pi * t / 128.0 + ph[2, 0])
                    img = np.roll(img, int(round(1.5 * t)), axis=1) * np.float32(gain)
                img += rng.integers(-2, 3, img.shape).astype(np.float32)
                out[b, t] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        elif kind == "saturated":
            cells = rng.integers(0, 2, (height // 24 + 3, width // 32 + 3, 3), dtype=np.uint8) * np.uint8(255)
            big = np.repeat(np.repeat(cells, 24, axis=0), 32, axis=1)
            for t in range(n_frames):
                dy, dx = (3 * t) % 24, (3 * t) % 32
                out[b, t] = big[dy:dy + height, dx:dx + width]
        else:
            raise ValueError(f"unknown kind {kind!r}")
    if kind == "jitter":
        hts = mask_heights(seed, n_clips, n_frames, 80, 140)
        hts[:, 3 % n_frames] = 0
        hts[:, 5 % n_frames] = height
        for b in range(n_clips):
            for t in range(n_frames):
                out[b, t, :hts[b, t]] = 0
    else:
        out[:, :, :mask_rows] = 0
    return out


def mask_heights(seed, n_clips, n_frames, lo, hi):
    """Per-frame mask heights of synth_frames(..., mask_jitter=(lo, hi)), i.i.d. uniform in lo..hi (harsher than video)."""
    return np.random.default_rng([seed, 0x6A17]).integers(lo, hi + 1, (n_clips, n_frames))


def synth_mel(seed, n_clips, n_mel_frames):
    """log-mel like (B,4T,80) fp32 ~ N(8,2.5)."""
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n_clips, n_mel_frames, 80)) * 2.5 + 8.0).astype(np.float32)


def synth_text(seed, n_clips, n_words, d=768):
    """Stand-in for XLM-R output (third-party, absent: SURVEY 8c): hidden states
    (B,L,768), ids [0, 1000+i.., 2], one sub-word per word, offsets (0,len)."""
    rng = np.random.default_rng(seed)
    L = n_words + 2
    states = rng.standard_normal((n_clips, L, d)).astype(np.float32)
    ids = np.zeros((n_clips, L), np.int64)
    ids[:, 1:1 + n_words] = 1000 + np.arange(n_words)[None]
    ids[:, -1] = 2
    offsets = np.zeros((n_clips, L, 2), np.int64)
    offsets[:, 1:1 + n_words, 1] = 4
    mask = np.ones((n_clips, L), np.int64)
    return states, mask, ids, offsets


def synth_boundaries(n_clips, n_words, stride=15, length=10):
    return [[[f"w{i}", stride * i, stride * i + length] for i in range(n_words)] for _ in range(n_clips)]


def synth_ragged_clip(seed, n_frames, n_words, d=768):
    """One clip of a RAGGED dataset (AVS clips are 25..220 frames, 3..12 words, the last word ends with the clip): GestSync
    features (T,1024) ~ N(0,1), mel (4T,80) ~ N(8,2.5), XLM-R stand-in states for words of 1..3 sub-words (the LAST word always
    has 2 or 3, so its row range matters: jegal.py:168-171 runs it to the end of the padded batch), and word boundaries that
    tile frames 0..T-1 with the last word ending on frame T-1.  Returns a dict with the pieces the drivers read."""
    rng = np.random.default_rng(seed)
    T, W = int(n_frames), int(n_words)
    feats = rng.standard_normal((T, 1024)).astype(np.float32)
    mel = (8.0 + 2.5 * rng.standard_normal((4 * T, 80))).astype(np.float32)
    nsub = rng.integers(1, 4, W)
    nsub[-1] = rng.integers(2, 4)
    L = int(nsub.sum()) + 2
    states = rng.standard_normal((L, d)).astype(np.float32)
    ids = np.zeros(L, np.int64)
    offsets = np.zeros((L, 2), np.int64)
    pos = 1
    for w in range(W):
        for j in range(int(nsub[w])):
            ids[pos] = 1000 + 7 * w + j
            offsets[pos] = (3 * j, 3 * j + 3)          # first sub-word of a word: offset[0] == 0 (jegal.py:148-150)
            pos += 1
    ids[pos] = 2
    mask = np.ones(L, np.int64)
    # word w covers frames cuts[w] .. cuts[w+1]-1 (gaps of 0..2 frames inside the clip, none at the end)
    cuts = np.sort(rng.choice(np.arange(1, T), W - 1, replace=False)) if W > 1 else np.array([], np.int64)
    cuts = np.concatenate(([0], cuts, [T]))
    wbs = []
    for w in range(W):
        s, e = int(cuts[w]), int(cuts[w + 1]) - 1
        if w < W - 1 and e - s >= 3:
            e -= int(rng.integers(0, 3))
        wbs.append([f"w{w}", s, e])
    return {"feats": feats, "mel": mel, "states": states, "mask": mask, "ids": ids, "offsets": offsets, "word_boundaries": wbs,
            "phrase": " ".join(w[0] for w in wbs)}


def planted_retrieval(seed, n, d=512, noise=0.2):
    """g_i ~ N(0,I) normalised, c_i = normalise(g_i + noise*N(0,I)).
    ``noise`` 0.2 keeps Recall@K away from both 0 and 1 at N=10k."""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n, d)).astype(np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    c = g + np.float32(noise) * rng.standard_normal((n, d)).astype(np.float32)
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    return g.astype(np.float32), c.astype(np.float32)


def planted_spotting(seed, n_clips, n_frames=150, n_words=30, d=512, noise=1.2):
    """boundaries [w_j, 5j, 5j+4], target word uniform; gesture rows
    in the target span replaced by normalise(c_target + noise*N)."""
    rng = np.random.default_rng(seed)
    gest, cont, bounds, targets = [], [], [], []
    for _ in range(n_clips):
        c = rng.standard_normal((n_words, d)).astype(np.float32)
        c /= np.linalg.norm(c, axis=1, keepdims=True)
        g = rng.standard_normal((n_frames, d)).astype(np.float32)
        g /= np.linalg.norm(g, axis=1, keepdims=True)
        wb = [[f"w{j}", 5 * j, 5 * j + 4] for j in range(n_words)]
        t = int(rng.integers(0, n_words))
        s, e = wb[t][1], wb[t][2]
        seg = c[t][None] + np.float32(noise / math.sqrt(d) * 4.0) * rng.standard_normal((e - s + 1, d)).astype(np.float32)
        g[s:e + 1] = seg / np.linalg.norm(seg, axis=1, keepdims=True)
        gest.append(g)
        cont.append(c)
        bounds.append(wb)
        targets.append(t)
    return gest, cont, bounds, targets


def planted_asd(seed, n_queries, d=512, n_frames=12, n_words=5, noise=1.6):
    """Active-speaker-detection fixture in the shape evaluate_asd.py:54-113 consumes: per query a temporal content
    embedding (W,d), its own (positive) temporal gesture embedding (T,d) and 0..5 negatives, so candidate lists have
    P = 1, 3, 5 or 6 entries (the reference slices all_gesture_embs[:2|4|6], shorter lists included).  The positive is
    the content direction plus `noise`; every third query also gets a HARD negative closer to the content than the
    positive, so the positive loses in some queries and at different P.  Returns (contents, positives, negatives)."""
    rng = np.random.default_rng(seed)
    contents, positives, negatives = [], [], []
    for i in range(n_queries):
        base = rng.standard_normal(d).astype(np.float32)
        base /= np.linalg.norm(base)
        c = base[None] + np.float32(0.3) * rng.standard_normal((n_words, d)).astype(np.float32) / np.float32(math.sqrt(d))
        pos = base[None] + np.float32(noise) * rng.standard_normal((n_frames, d)).astype(np.float32) / np.float32(math.sqrt(d))
        n_neg = (0, 2, 4, 5)[i % 4]
        negs = []
        for k in range(n_neg):
            g = rng.standard_normal((n_frames, d)).astype(np.float32) / np.float32(math.sqrt(d))
            if i % 3 == 0 and k == (i // 3) % n_neg:
                g = base[None] + np.float32(0.5 * noise) * g          # hard negative: beats the positive
            negs.append(g.astype(np.float32))
        contents.append(c.astype(np.float32))
        positives.append(pos.astype(np.float32))
        negatives.append(negs)
    return contents, positives, negatives
