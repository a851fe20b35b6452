"""GestSync visual feature extractor -- MI355X engine behind the reference's interface.

Mirrors ``models/gestsync.py`` of the reference: ``GestSync()`` (no ctor args, :9),
``forward_vid(x, return_feats=False)`` (:148-162), ``load_state_dict`` with the reference's keys
(the audio / LSTM tensors of the checkpoint are accepted and ignored).  All tensor math runs in
libjegal_hip; there is no PyTorch fallback.
"""
import torch

from ._lib import Engine


class GestSync:
    def __init__(self, device=None, engine=None):
        self.engine = engine if engine is not None else Engine.get(device)
        self._loaded = False

    # nn.Module-style plumbing so reference drivers (inference_embs.py:655-668) read the same
    def cuda(self, device=None):
        return self

    def eval(self):
        return self

    def load_state_dict(self, state_dict, strict=True):
        sd = {k.replace("module.", ""): v for k, v in state_dict.items()}
        self.engine.load_tensors(sd)
        self.engine.finalize(1)     # raises on any missing hot-path key (strict)
        self._loaded = True
        return self

    def _check(self):
        if not self._loaded:
            raise RuntimeError("GestSync: call load_state_dict() first")

    def forward_vid(self, x, return_feats=False):
        """x (N,3,25,270,480) float -> (N,1024,21) [, out_conv (N,512,21)]  (gestsync.py:148-162)."""
        self._check()
        with torch.no_grad():
            return self.engine.gestsync_windows(x, return_feats=return_feats)

    def extract_clip_feats(self, frames, lengths=None):
        """frames (B,T,270,480,3) uint8 (or float in [0,1]) -> (B,T,1024): the padded/windowed loop of
        inference_embs.py:283,476-522 with the conv stack de-duplicated across windows.  lengths (optional): each clip's own
        frame count in a batch padded to T with copies of the clips' last frames (Engine.gestsync_clip)."""
        self._check()
        if frames.dim() == 4:
            frames = frames.unsqueeze(0)
        with torch.no_grad():
            return self.engine.gestsync_clip(frames, lengths)

    __call__ = forward_vid
