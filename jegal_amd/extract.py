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


FRAME_ROW_BYTES = 480 * 3
FRAME_BYTES = 270 * FRAME_ROW_BYTES


class _MaskedPacker:
    """Host side of the masked upload: the kept rows (>= row0) of every frame of a batch, back to back in a pinned buffer."""

    def __init__(self, batch, T, pinned=True):
        self.batch, self.T = batch, T
        pin = (lambda t: t.pin_memory()) if pinned else (lambda t: t)        # pinned=False: host-logic tests without a GPU
        self.buf = pin(torch.empty((batch * T * FRAME_BYTES,), dtype=torch.uint8))
        self.row0 = pin(torch.zeros((batch * T,), dtype=torch.int32))
        self.offs = pin(torch.zeros((batch * T,), dtype=torch.int64))
        self.reset()

    def reset(self):
        self.n, self.used = 0, 0

    def add(self, clip, row0):
        """clip (T,270,480,3) uint8 whose rows < row0 are blank (not checked: they are simply not shipped); row0: int or (T,) ints."""
        clip = np.asarray(clip)
        if clip.shape != (self.T, 270, 480, 3) or clip.dtype != np.uint8:
            raise ValueError(f"clip must be uint8 ({self.T},270,480,3), got {clip.dtype} {clip.shape}")
        if self.n >= self.batch:
            raise ValueError("batch is full")
        r0 = np.broadcast_to(np.asarray(row0, np.int64), (self.T,))
        if r0.min() < 0 or r0.max() > 270:
            raise ValueError("row0 must be in 0..270")
        f0 = self.n * self.T
        kept = (270 - r0) * FRAME_ROW_BYTES
        offs = self.used + np.concatenate(([0], np.cumsum(kept)[:-1]))
        self.row0.numpy()[f0:f0 + self.T] = r0
        self.offs.numpy()[f0:f0 + self.T] = offs
        dst = self.buf.numpy()
        if (r0 == r0[0]).all():                                   # one strided copy for a clip with one mask height
            n = int(kept[0])
            dst[self.used:self.used + self.T * n].reshape(self.T, n)[:] = clip[:, int(r0[0]):].reshape(self.T, n)
        else:
            for t in range(self.T):
                dst[offs[t]:offs[t] + kept[t]] = clip[t, int(r0[t]):].reshape(-1)
        self.used += int(kept.sum())
        self.n += 1


class _SourcePacker:
    """Host side of the source-resolution upload: decoder-resolution frames (T,H,W,3) with their mask rows (mask_y = y2+15 per frame,
    -1 = no face; inference_embs.py:255-270), only the rows BELOW each frame's mask packed back to back in a pinned buffer --
    what jg_mask_resize_packed consumes."""

    def __init__(self, batch, T, H, W, pinned=True):
        self.batch, self.T, self.H, self.W = batch, T, int(H), int(W)
        self.row_bytes = self.W * 3
        pin = (lambda t: t.pin_memory()) if pinned else (lambda t: t)
        self.buf = pin(torch.empty((batch * T * self.H * self.row_bytes,), dtype=torch.uint8))
        self.mask_y = pin(torch.zeros((batch * T,), dtype=torch.int32))
        self.offs = pin(torch.zeros((batch * T,), dtype=torch.int64))
        self.reset()

    def reset(self):
        self.n, self.used = 0, 0

    def add(self, clip, mask_y):
        """clip (T,H,W,3) uint8 source frames (NOT masked: the rows 0..mask_y are simply not shipped); mask_y int or (T,) ints."""
        clip = np.asarray(clip)
        if clip.shape != (self.T, self.H, self.W, 3) or clip.dtype != np.uint8:
            raise ValueError(f"clip must be uint8 ({self.T},{self.H},{self.W},3), got {clip.dtype} {clip.shape}")
        if self.n >= self.batch:
            raise ValueError("batch is full")
        my = np.broadcast_to(np.asarray(mask_y, np.int64), (self.T,))
        if my.min() < -1:
            raise ValueError("mask_y must be >= -1 (-1: no face found)")
        r0 = np.clip(my + 1, 0, self.H)
        f0 = self.n * self.T
        kept = (self.H - r0) * self.row_bytes
        offs = self.used + np.concatenate(([0], np.cumsum(kept)[:-1]))
        self.mask_y.numpy()[f0:f0 + self.T] = my
        self.offs.numpy()[f0:f0 + self.T] = offs
        dst = self.buf.numpy()
        if (r0 == r0[0]).all():
            n = int(kept[0])
            dst[self.used:self.used + self.T * n].reshape(self.T, n)[:] = clip[:, int(r0[0]):].reshape(self.T, n)
        else:
            for t in range(self.T):
                dst[offs[t]:offs[t] + kept[t]] = clip[t, int(r0[t]):].reshape(-1)
        self.used += int(kept.sum())
        self.n += 1


class GestureStreamer:
    """Streams host-resident clips through ``jg_extract_gesture`` with the uploads hidden behind the compute.

    A 32-clip batch of uint8 crops is 1.87 GB; at PCIe Gen5 rates its upload takes longer than the 12 ms
    the GPU needs for it, so a serving / dataset-extraction loop must keep the copy engine busy all the
    time: two pinned host buffers + two device buffers, batch k+1 is packed and copied (copy stream) while
    batch k runs (compute stream), batch k-1's embeddings go back on the copy stream.  torch is plumbing
    only (pinned memory, streams, events); the compute is one library call per batch.

    ``masked=True`` ships fewer bytes: the reference blanks rows 0..y2+15 of every crop (inference_embs.py:264-270), the
    producer knows y2, so only the rows below each frame's mask cross the host link (``run(clips, mask_rows)`` /
    ``run_filled`` with a packer) and ``jg_unpack_masked`` rebuilds the dense batch on the device, in front of the batch's compute.
    Bit-identical to uploading the blanked crops whole.

    ``source_hw=(H, W)`` ships the DECODER's frames: the reference resizes its 228x314 / 294x294 crops up to 270x480 on the host
    (inference_embs.py:255-276), which is up to 1.8x more bytes than the decoder produced.  Clips are (T,H,W,3) uint8 source
    frames, ``mask_rows`` gives mask_y = y2+15 per frame (-1: no face), only the source rows below the mask cross the link and
    ``jg_mask_resize_packed`` (mask + cv2-style bilinear resize) builds the 270x480 crops on the device, in front of the batch's compute.
    Bit-identical to ``load_rgb_masked_frames`` + the resident path.

    ``run(clips)``: ``clips`` iterates over (T,270,480,3) uint8 numpy arrays of one common T; yields
    ``(first_clip_index, embeddings (n,T,512) float32 numpy)`` per batch, in order.
    (The reference's loop does this synchronously per clip: inference_embs.py:476-522,629-637.)
    """

    def __init__(self, engine, batch=32, frames=150, masked=False, source_hw=None):
        self.eng, self.batch, self.T, self.masked = engine, int(batch), int(frames), bool(masked) or source_hw is not None
        q = os.environ.get("GPU_MAX_HW_QUEUES")
        if q is None or not q.isdigit() or int(q) < 8:
            import warnings
            warnings.warn("GestureStreamer uses five streams (H2D, D2H, compute, two engine lanes) and the HIP runtime's default of 4 hardware "
                          "queues can serialise the next upload behind the current compute (2 150 -> 1 230 clips/s measured): call "
                          "jegal_amd.want_hw_queues() -- or export GPU_MAX_HW_QUEUES=8 -- before the process's first HIP call", RuntimeWarning)
        self.source_hw = None if source_hw is None else (int(source_hw[0]), int(source_hw[1]))
        dev = engine.device
        shape = (self.batch, self.T, 270, 480, 3)
        if self.source_hw is not None:
            H, W = self.source_hw
            self.packer = [_SourcePacker(self.batch, self.T, H, W) for _ in range(2)]
            self.d_packed = [torch.empty((self.batch * self.T * H * W * 3,), dtype=torch.uint8, device=dev) for _ in range(2)]
            self.d_row0 = [torch.empty((self.batch * self.T,), dtype=torch.int32, device=dev) for _ in range(2)]      # mask_y here
            self.d_offs = [torch.empty((self.batch * self.T,), dtype=torch.int64, device=dev) for _ in range(2)]
        elif self.masked:
            self.packer = [_MaskedPacker(self.batch, self.T) for _ in range(2)]
            self.d_packed = [torch.empty((self.batch * self.T * FRAME_BYTES,), dtype=torch.uint8, device=dev) for _ in range(2)]
            self.d_row0 = [torch.empty((self.batch * self.T,), dtype=torch.int32, device=dev) for _ in range(2)]
            self.d_offs = [torch.empty((self.batch * self.T,), dtype=torch.int64, device=dev) for _ in range(2)]
        else:
            self.h_in = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.d_in = [torch.empty(shape, dtype=torch.uint8, device=dev) for _ in range(2)]
        self.d_out = [torch.empty((self.batch, self.T, 512), dtype=torch.float32, device=dev) for _ in range(2)]
        self.h_out = [torch.empty((self.batch, self.T, 512), dtype=torch.float32).pin_memory() for _ in range(2)]
        self.copy = torch.cuda.Stream(dev)          # H2D
        self.down = torch.cuda.Stream(dev)          # D2H: on its own stream, or upload k+1 would queue behind `wait computed k`
        self.compute = torch.cuda.Stream(dev)
        self._used = [0, 0]                         # packed bytes of the batch in each slot
        self.uploaded = [torch.cuda.Event() for _ in range(2)]
        self.computed = [torch.cuda.Event() for _ in range(2)]
        self.downloaded = [torch.cuda.Event() for _ in range(2)]

    def _pack(self, it, slot, rows_it=None):
        """Fill pinned buffer `slot` from the iterator; returns the number of clips packed."""
        n = 0
        if self.masked:
            pk = self.packer[slot]
            pk.reset()
            none = -1 if self.source_hw is not None else 0          # no mask given: the whole frame is shipped
            for clip in it:
                pk.add(clip, none if rows_it is None else next(rows_it))
                if pk.n == self.batch:
                    break
            return pk.n
        dst = self.h_in[slot].numpy()
        for clip in it:
            clip = np.asarray(clip)
            if clip.shape != (self.T, 270, 480, 3) or clip.dtype != np.uint8:
                raise ValueError(f"clip must be uint8 ({self.T},270,480,3), got {clip.dtype} {clip.shape}")
            dst[n] = clip
            n += 1
            if n == self.batch:
                break
        return n

    def run(self, clips, mask_rows=None):
        """clips: iterable of (T,270,480,3) uint8 arrays (copied into the pinned staging buffers here).  masked streamer:
        mask_rows iterates alongside and gives each clip's first unmasked row (int, or (T,) ints per frame); None = 0.
        source_hw streamer: clips are (T,H,W,3) source frames and mask_rows gives mask_y = y2+15 (-1: no face; None = -1)."""
        it = iter(clips)
        rows_it = None if mask_rows is None else iter(mask_rows)
        if rows_it is not None and not self.masked:
            raise ValueError("mask_rows needs GestureStreamer(..., masked=True)")
        return self.run_filled(lambda buf, k: self._pack(it, k & 1, rows_it))

    def _upload(self, slot, n):
        """H2D of batch `slot` on the copy stream.  The kernel that rebuilds the dense batch from the packed rows (_unpack) is NOT
        issued here: beside the persistent compute kernels of the previous batch it cost that batch 1.5-2.7 ms (its workgroups
        take CUs the one-workgroup-per-CU kernels were launched for), in front of its own batch on the compute stream it costs its
        stand-alone 1.0 ms (tools/stream_timeline.py: 15.3 -> 13.6 ms per streamed batch)."""
        with torch.cuda.stream(self.copy):
            if self.masked:
                pk = self.packer[slot]
                F = n * self.T
                self.d_packed[slot][:pk.used].copy_(pk.buf[:pk.used], non_blocking=True)
                self.d_row0[slot][:F].copy_((pk.mask_y if self.source_hw is not None else pk.row0)[:F], non_blocking=True)
                self.d_offs[slot][:F].copy_(pk.offs[:F], non_blocking=True)
                self._used[slot] = pk.used
            else:
                self.d_in[slot][:n].copy_(self.h_in[slot][:n], non_blocking=True)
            self.uploaded[slot].record(self.copy)

    def _unpack(self, slot, n):
        """Packed rows -> dense (n,T,270,480,3) batch on the CURRENT (compute) stream: jg_mask_resize_packed / jg_unpack_masked."""
        if not self.masked:
            return
        F, used = n * self.T, self._used[slot]
        if used == 0:                            # every frame masked completely: nothing crossed the link
            self.d_in[slot][:n].zero_()
        elif self.source_hw is not None:
            self.eng.mask_resize_packed(self.d_packed[slot][:used], self.d_offs[slot][:F], self.d_row0[slot][:F], self.source_hw[0],
                                        self.source_hw[1], self.d_in[slot][:n])
        else:
            self.eng.unpack_masked(self.d_packed[slot][:used], self.d_row0[slot][:F], self.d_offs[slot][:F], self.d_in[slot][:n])

    def run_filled(self, fill):
        """fill(buffer, batch_index) -> number of clips written (0 = end), for producers (decoders) that can write their crops
        straight into pinned memory.  buffer: the pinned uint8 array (batch,T,270,480,3); masked streamer: the slot's
        _MaskedPacker -- call reset() and add(clip, row0) per clip, or leave it as it is to re-send its content."""
        pending = []                      # (slot, first index, n) of batches whose embeddings are not yet returned
        first, k = 0, 0
        buf = (lambda s_: self.packer[s_]) if self.masked else (lambda s_: self.h_in[s_].numpy())
        n = fill(buf(0), 0)
        while n > 0 or pending:
            slot = k & 1
            if n > 0:
                self._upload(slot, n)
                with torch.cuda.stream(self.compute):
                    self.compute.wait_event(self.uploaded[slot])
                    self._unpack(slot, n)
                    self.eng.extract_gesture(self.d_in[slot][:n], self.d_out[slot][:n])
                    self.computed[slot].record(self.compute)
                with torch.cuda.stream(self.down):
                    self.down.wait_event(self.computed[slot])
                    self.h_out[slot][:n].copy_(self.d_out[slot][:n], non_blocking=True)
                    self.downloaded[slot].record(self.down)
                pending.append((slot, first, n))
                first += n
            # pack the next batch on the host while the GPU works on this one; its pinned buffer and its
            # device buffers were last used by batch k-1, whose results must be handed out first
            if len(pending) == 2 or n == 0:
                s0, f0, n0 = pending.pop(0)
                self.downloaded[s0].synchronize()
                yield f0, self.h_out[s0][:n0].numpy().copy()
            k += 1
            n = fill(buf(k & 1), k) if n > 0 else 0


# mediapipe face-mesh indices of the face oval (inference_embs.py:248-250)
FACE_OVAL_IDX = (10, 21, 54, 58, 67, 93, 103, 109, 127, 132, 136, 148, 149, 150, 152, 162, 172, 176, 234, 251, 284, 288,
                 297, 323, 332, 338, 356, 361, 365, 377, 378, 379, 389, 397, 400, 454)


def load_rgb_masked_frames(engine, input_frames, kp_dict):
    """Counterpart of inference_embs.py:235-286 (`load_rgb_masked_frames`) with the pixel work on the GPU.

    input_frames: (T,H,W,3) uint8 array/tensor (or list of equally sized frames); kp_dict = {"kps": per-frame dicts with
    "face" = list of {"x","y"} landmarks (normalised) or None, "resolution": (H, W)} as written by the reference's
    mediapipe step.  Returns the masked (T,270,480,3) uint8 crops on the device; the /255 and the +-12 frame edge
    padding of :279-283 are part of `jg_gestsync_clip`.  The per-frame mask row y2+15 (:266-270) is computed here on
    the host exactly as the reference does (int() truncation of landmark*resolution, max over the oval)."""
    kps, res = kp_dict["kps"], kp_dict["resolution"]
    mask_y = []
    for fk in kps:
        face = fk["face"]
        if face is None:
            mask_y.append(-1)
        else:
            ys = [int(face[i]["y"] * res[0]) for i in range(len(face)) if i in FACE_OVAL_IDX]
            mask_y.append(max(ys) + 15)
    frames = torch.as_tensor(np.asarray(input_frames))
    return engine.mask_resize(frames, mask_y)
