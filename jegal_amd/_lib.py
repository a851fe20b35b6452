"""ctypes binding of libjegal_hip.so (include/jegal_hip.h).

The product path has no CPU fallback: if the HIP library is missing or cannot be loaded,
importing the engine raises.  PyTorch is used only for device memory and streams; every
tensor op of the hot path runs inside the library.
"""
import ctypes
import os

import numpy as np
import torch  # imported first so libamdhip64.so.7 resolves to the runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libjegal_hip.so")

JG_F32, JG_F16, JG_I64, JG_U8 = 0, 1, 2, 3
PREC_FP16, PREC_FP16_W2, PREC_FP16_W2_ALL, PREC_FP16_BC, PREC_BF16, PREC_FP16_RC, PREC_FP32 = 0, 1, 2, 3, 4, 5, 6
AUDIT_CONV, AUDIT_GESTSYNC, AUDIT_JEGAL, AUDIT_CONTENT, AUDIT_XLMR = 1, 2, 4, 8, 16      # option audit_stages (include/jegal_hip.h)
STAGES = ["stack_frames", "conv1", "maxpool", "conv2-fc6+audio_cnn", "gemm", "attention", "layernorm", "misc", "conv1_aux"]

_P = ctypes.c_void_p
_I = ctypes.c_int
_SIGS = {
    "jg_create": [_I, ctypes.POINTER(_P)],
    "jg_destroy": [_P],
    "jg_set_stream": [_P, _P],
    "jg_set_precision": [_P, _I],
    "jg_set_chunk": [_P, _I],
    "jg_set_option": [_P, ctypes.c_char_p, _I],
    "jg_sync": [_P],
    "jg_load_tensor": [_P, ctypes.c_char_p, _P, ctypes.POINTER(ctypes.c_int64), _I, _I],
    "jg_finalize_weights": [_P, _I],
    "jg_clear_staged_tensors": [_P],
    "jg_calibrate_gesture": [_P, _P, _I, _I, _I],
    "jg_gestsync_clip": [_P, _P, _I, _I, _I, _P],
    "jg_gestsync_clip_ragged": [_P, _P, _I, _I, _I, ctypes.POINTER(ctypes.c_int32), _P],
    "jg_gestsync_windows": [_P, _P, _I, _P, _P],
    "jg_debug_conv1_pool": [_P, _P, _I, _I, _I, _P],
    "jg_debug_gemm": [_P, _I, _I, _I, _I, _I, ctypes.POINTER(ctypes.c_double)],
    "jg_debug_gemm_ex": [_P, _P, _P, _I, _I, _I, _I, _I, ctypes.POINTER(ctypes.c_double)],
    "jg_debug_conv2_rowskip": [_P, ctypes.POINTER(ctypes.c_int)],
    "jg_debug_conv_rows": [_P, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)],
    "jg_jegal_gestures": [_P, _P, _P, _I, _I, _I, _P],
    "jg_jegal_audio": [_P, _P, _I, _I, _P],
    "jg_jegal_audio_ragged": [_P, _P, _I, _I, ctypes.POINTER(ctypes.c_int32), _P],
    "jg_audio_len": [_I],
    "jg_logmel": [_P, _P, _I, _I, _P, _P],
    "jg_mask_resize": [_P, _P, _I, _I, _I, _P, _P],
    "jg_unpack_masked": [_P, _P, ctypes.c_int64, _P, _P, _I, _P],
    "jg_mask_resize_packed": [_P, _P, ctypes.c_int64, _P, _I, _I, _I, _P, _P],
    "jg_jegal_text": [_P, _P, _P, _I, _I, _P],
    "jg_xlmr_encode": [_P, _P, _P, _I, _I, _P],
    "jg_calibrate_xlmr": [_P, _P, _P, _I, _I],
    "jg_word_pool": [_P, _P, _I, _P, _I, _P, _I, _I],
    "jg_fuse_content": [_P, _P, _I, _P],
    "jg_l2norm": [_P, _P, _P, _I, _I],
    "jg_extract_gesture": [_P, _P, _I, _I, _I, _P],
    "jg_pool_mean": [_P, _P, _P, _I, _I, _P],
    "jg_sim_rank": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "jg_spot": [_P, _P, _P, _P, _P, _P, _I, _I, ctypes.c_float, _P, _P],
    "jg_asd": [_P, _P, _P, _P, _I, _I, ctypes.c_float, _P],
    "jg_comm_get_unique_id": [ctypes.c_char_p],
    "jg_comm_init": [_P, ctypes.c_char_p, _I, _I],
    "jg_comm_destroy": [_P],
    "jg_allgather": [_P, _P, _P, ctypes.c_int64],
    "jg_allreduce_sum_i64": [_P, _P, _I],
    "jg_profile_enable": [_P, _I],
    "jg_profile_get": [_P, _I, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)],
    "jg_profile_reset": [_P],
}
EXPORTS = sorted(list(_SIGS) + ["jg_last_error", "jg_stage_name", "jg_workspace_bytes"])

_lib = None


def load_library():
    """Load libjegal_hip.so or raise.  No fallback by design."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C jegal_amd/csrc`).  jegal_amd has no CPU/PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _I
    lib.jg_last_error.argtypes = [_P]
    lib.jg_last_error.restype = ctypes.c_char_p
    lib.jg_stage_name.argtypes = [_I]
    lib.jg_stage_name.restype = ctypes.c_char_p
    lib.jg_workspace_bytes.argtypes = [_P]
    lib.jg_workspace_bytes.restype = ctypes.c_int64
    _lib = lib
    return lib


class JegalError(RuntimeError):
    pass


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class Engine:
    """One jg_handle on one GPU.  Not thread-safe; one per process/GPU."""

    _cache = {}

    @classmethod
    def get(cls, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("jegal_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if idx not in cls._cache:
            cls._cache[idx] = cls(idx)
        return cls._cache[idx]

    def __init__(self, device_index=0, precision=None):
        self.lib = load_library()
        self.device = torch.device("cuda", device_index)
        torch.cuda.init()
        with torch.cuda.device(self.device):
            torch.zeros(1, device=self.device)          # make sure the HIP context exists
        h = _P()
        rc = self.lib.jg_create(device_index, ctypes.byref(h))
        if rc != 0:
            raise JegalError(f"jg_create({device_index}) failed with {rc}")
        self.h = h
        self.precision = PREC_FP16_RC              # the library's default (jg_handle::precision)
        if precision is not None:
            self.set_precision(precision)
        self.finalized = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.jg_destroy(self.h)
            self.h = None
            for k, v in list(Engine._cache.items()):
                if v is self:
                    del Engine._cache[k]

    def _ck(self, rc):
        if rc != 0:
            raise JegalError(f"libjegal_hip error {rc}: {self.lib.jg_last_error(self.h).decode()}")

    def _bind_stream(self):
        self._ck(self.lib.jg_set_stream(self.h, _P(torch.cuda.current_stream(self.device).cuda_stream)))

    # ---- weights
    def set_precision(self, mode):
        self._ck(self.lib.jg_set_precision(self.h, mode))
        self.precision = int(mode)

    def set_option(self, name, value):
        self._ck(self.lib.jg_set_option(self.h, name.encode(), int(value)))

    def set_chunk(self, clips):
        self._ck(self.lib.jg_set_chunk(self.h, int(clips)))

    def load_tensors(self, state_dict):
        for name, val in state_dict.items():
            if isinstance(val, torch.Tensor):
                val = val.detach().cpu().numpy()
            arr = np.asarray(val)
            if arr.dtype == np.float16:
                code = JG_F16
            elif arr.dtype == np.int64:
                code = JG_I64
            else:
                arr = arr.astype(np.float32, copy=False)
                code = JG_F32
            arr = np.ascontiguousarray(arr)
            shape = (ctypes.c_int64 * max(arr.ndim, 1))(*arr.shape)
            self._ck(self.lib.jg_load_tensor(self.h, name.encode(), arr.ctypes.data_as(_P), shape, arr.ndim, code))

    def finalize(self, which, clear_staged=True):
        """jg_finalize_weights; the Python facades load one model's state_dict and finalize it at once, so whatever that
        finalize did not consume (net_aud.*, lstm.* of a GestSync checkpoint) is dropped afterwards."""
        with torch.cuda.device(self.device):
            self._ck(self.lib.jg_finalize_weights(self.h, which))
        if clear_staged:
            self._ck(self.lib.jg_clear_staged_tensors(self.h))
        self.finalized |= which

    # ---- helpers
    def _f32(self, t):
        return torch.as_tensor(t, device=self.device).to(torch.float32).contiguous()

    def _i32(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.int32).contiguous()
        return torch.as_tensor(np.ascontiguousarray(np.asarray(a, np.int32)), device=self.device)

    def sync(self):
        self._ck(self.lib.jg_sync(self.h))

    def calibrate(self, frames=None):
        """Re-run the bias-correction calibration (precision mode 3) on real clips (B,T,270,480,3);
        None = the built-in deterministic synthetic clips used by finalize()."""
        self._bind_stream()
        if frames is None:
            self._ck(self.lib.jg_calibrate_gesture(self.h, None, JG_U8, 0, 0))
            return
        frames, code, B, T = self._frames_arg(frames)
        self._ck(self.lib.jg_calibrate_gesture(self.h, _ptr(frames), code, B, T))

    def _frames_arg(self, frames):
        """Validate a clip batch and put it on this engine's device: (B,T,270,480,3) uint8 (decoded video) or
        floating point in [0,1].  Returns (contiguous device tensor, dtype code, B, T).  A wrong shape would hand the
        conv1 kernel a raw pointer it reads 270x480x3 bytes per frame from, so it is rejected here."""
        if not isinstance(frames, torch.Tensor):
            frames = torch.as_tensor(frames)
        if frames.dim() != 5 or tuple(frames.shape[2:]) != (270, 480, 3):
            raise ValueError(f"frames must be (B,T,270,480,3), got {tuple(frames.shape)}")
        if frames.shape[0] == 0 or frames.shape[1] == 0:
            raise ValueError("frames must hold at least one clip of at least one frame")
        frames = frames.to(self.device)
        if frames.dtype == torch.uint8:
            code = JG_U8
        elif frames.dtype.is_floating_point:
            frames, code = frames.to(torch.float32), JG_F32
        else:
            raise ValueError(f"frames must be uint8 or floating point, got {frames.dtype}")
        return frames.contiguous(), code, frames.shape[0], frames.shape[1]

    def _out_arg(self, out, shape):
        if out is None:
            return torch.empty(shape, dtype=torch.float32, device=self.device)
        if (not isinstance(out, torch.Tensor) or out.device != self.device or out.dtype != torch.float32
                or tuple(out.shape) != tuple(shape) or not out.is_contiguous()):
            raise ValueError(f"out must be a contiguous float32 tensor of shape {tuple(shape)} on {self.device}")
        return out

    # ---- GestSync
    def gestsync_clip(self, frames, lengths=None):
        """frames (B,T,270,480,3) uint8 or float32 cuda tensor -> (B,T,1024) fp32.  lengths (B ints, optional): frames of each clip
        that are its own in a batch padded to T with copies of the clips' last frames (jg_gestsync_clip_ragged): rows t < lengths[b]
        of clip b are then what the clip gives alone, whatever T is (bit for bit for clips of >= 49 frames; a shorter clip alone takes
        the unfused plan and agrees within the contract, include/jegal_hip.h)."""
        self._bind_stream()
        frames, code, B, T = self._frames_arg(frames)
        out = torch.empty((B, T, 1024), dtype=torch.float32, device=self.device)
        if lengths is None:
            self._ck(self.lib.jg_gestsync_clip(self.h, _ptr(frames), code, B, T, _ptr(out)))
            return out
        v = [int(x) for x in lengths]
        if len(v) != B:
            raise ValueError("lengths needs one entry per clip")
        self._ck(self.lib.jg_gestsync_clip_ragged(self.h, _ptr(frames), code, B, T, (ctypes.c_int32 * B)(*v), _ptr(out)))
        return out

    def debug_gemm(self, M, N, K, mode=0, iters=10, a16=None, w16=None):
        """ms per launch of the production GEMM; a16 (M,K) / w16 (N,K): optional fp16 cuda operands (else constant fill)."""
        self._bind_stream()
        ms = ctypes.c_double()
        for t, shp in ((a16, (M, K)), (w16, (N, K))):
            if t is not None and (t.dtype != torch.float16 or tuple(t.shape) != shp or not t.is_contiguous() or t.device != self.device):
                raise ValueError(f"debug_gemm operand must be a contiguous fp16 {shp} tensor on {self.device}")
        self._ck(self.lib.jg_debug_gemm_ex(self.h, _ptr(a16), _ptr(w16), M, N, K, mode, iters, ctypes.byref(ms)))
        return ms.value

    def debug_conv2_rowskip(self):
        """Leading conv2 output rows per image that the last conv stack copied instead of computing ("conv2_row_skip")."""
        rows = ctypes.c_int()
        self._ck(self.lib.jg_debug_conv2_rowskip(self.h, ctypes.byref(rows)))
        return rows.value

    def debug_conv_rows(self):
        """(computed, full) output pixels of conv2 .. conv5 in the last conv stack (per-position row skip)."""
        c, f = (ctypes.c_int64 * 4)(), (ctypes.c_int64 * 4)()
        self._ck(self.lib.jg_debug_conv_rows(self.h, c, f))
        return list(c), list(f)

    def debug_conv1_pool(self, frames_u8, pad):
        """conv1+BN+ReLU+maxpool only: (B,T,270,480,3) u8 -> (B*(T+2*pad-4),43,78,64) fp16 NHWC."""
        self._bind_stream()
        frames_u8 = frames_u8.to(self.device).contiguous()
        B, T = frames_u8.shape[:2]
        out = torch.empty((B * (T + 2 * pad - 4), 43, 78, 64), dtype=torch.float16, device=self.device)
        self._ck(self.lib.jg_debug_conv1_pool(self.h, _ptr(frames_u8), B, T, pad, _ptr(out)))
        return out

    def gestsync_windows(self, x, return_feats=False):
        self._bind_stream()
        x = self._f32(x)
        if tuple(x.shape[1:]) != (3, 25, 270, 480):
            raise ValueError(f"x must be (N,3,25,270,480), got {tuple(x.shape)}")
        N = x.shape[0]
        out = torch.empty((N, 1024, 21), dtype=torch.float32, device=self.device)
        oc = torch.empty((N, 512, 21), dtype=torch.float32, device=self.device) if return_feats else None
        self._ck(self.lib.jg_gestsync_windows(self.h, _ptr(x), N, _ptr(out), _ptr(oc)))
        return (out, oc) if return_feats else out

    def extract_gesture(self, frames, out=None):
        """frames -> unit-norm gesture embedding (B,T,512), one library call."""
        self._bind_stream()
        frames, code, B, T = self._frames_arg(frames)
        if T > 500:
            raise ValueError("clips are limited to 500 frames (JEGAL's positional table, modules.py:136)")
        out = self._out_arg(out, (B, T, 512))
        self._ck(self.lib.jg_extract_gesture(self.h, _ptr(frames), code, B, T, _ptr(out)))
        return out

    # ---- JEGAL
    def jegal_gestures(self, feats, mask=None, align=False):
        self._bind_stream()
        feats = self._f32(feats)
        B, T, D = feats.shape
        if D != 1024:
            raise ValueError("visual feats must have 1024 channels")
        m = None if mask is None else self._f32(mask).reshape(B, T)
        out = torch.empty((B, T, 512), dtype=torch.float32, device=self.device)
        self._ck(self.lib.jg_jegal_gestures(self.h, _ptr(feats), _ptr(m), B, T, int(bool(align)), _ptr(out)))
        return out

    def audio_len(self, Tm):
        return int(self.lib.jg_audio_len(int(Tm)))

    def jegal_audio(self, mel, valid_len=None):
        """mel (B,Tm,80) -> (B, audio_len(Tm), 256).  valid_len (B ints, optional): mel frames each clip of a zero-padded batch
        really holds -- rows t < audio_len(valid_len[b]) of clip b then equal the clip run alone (jg_jegal_audio_ragged)."""
        self._bind_stream()
        mel = self._f32(mel)
        B, Tm, F = mel.shape
        if F != 80:
            raise ValueError("mel must have 80 bands")
        out = torch.empty((B, self.audio_len(Tm), 256), dtype=torch.float32, device=self.device)
        if valid_len is None:
            self._ck(self.lib.jg_jegal_audio(self.h, _ptr(mel), B, Tm, _ptr(out)))
            return out
        v = [int(x) for x in valid_len]
        if len(v) != B:
            raise ValueError("valid_len needs one entry per clip")
        self._ck(self.lib.jg_jegal_audio_ragged(self.h, _ptr(mel), B, Tm, (ctypes.c_int32 * B)(*v), _ptr(out)))
        return out

    def mask_resize(self, frames_u8, mask_y):
        """load_rgb_masked_frames (inference_embs.py:235-276) minus /255 and the edge pad: (T,H,W,3) uint8 source frames,
        mask_y (T,) int (last blanked source row = y2+15, or -1 for "no face") -> (T,270,480,3) uint8 masked crops."""
        self._bind_stream()
        frames_u8 = frames_u8.to(self.device).contiguous()
        if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[-1] != 3:
            raise ValueError("frames must be uint8 (T,H,W,3)")
        T, H, W, _ = frames_u8.shape
        my = torch.as_tensor(mask_y, dtype=torch.int32).to(self.device).contiguous()
        if my.numel() != T:
            raise ValueError("mask_y needs one entry per frame")
        out = torch.empty((T, 270, 480, 3), dtype=torch.uint8, device=self.device)
        self._ck(self.lib.jg_mask_resize(self.h, _ptr(frames_u8), T, H, W, _ptr(my), _ptr(out)))
        return out

    def unpack_masked(self, packed, row0, offsets, dst):
        """packed uint8 (bytes,) / row0 int32 (F,) / offsets int64 (F,) device tensors -> dst (F,270,480,3) uint8 (written in place).
        offsets must be multiples of 16; the kernel writes a frame as zeros instead of reading it when its metadata points outside
        `packed` (row0 outside 0..270, misaligned / negative offset, rows past the end)."""
        self._bind_stream()
        F = row0.numel()
        if (packed.dtype != torch.uint8 or row0.dtype != torch.int32 or offsets.dtype != torch.int64 or dst.dtype != torch.uint8
                or offsets.numel() != F or dst.numel() != F * 270 * 480 * 3 or not dst.is_contiguous() or not packed.is_contiguous()):
            raise ValueError("unpack_masked: packed uint8, row0 int32 (F), offsets int64 (F), dst uint8 (F,270,480,3)")
        self._ck(self.lib.jg_unpack_masked(self.h, _ptr(packed), packed.numel(), _ptr(row0), _ptr(offsets), F, _ptr(dst)))
        return dst

    def mask_resize_packed(self, packed, offsets, mask_y, H, W, dst):
        """Source-resolution frames with only the rows below each frame's mask shipped (jg_mask_resize_packed): packed uint8 (bytes,),
        offsets int64 (F,), mask_y int32 (F,) device tensors -> dst (F,270,480,3) uint8 (written in place)."""
        self._bind_stream()
        F = mask_y.numel()
        if (packed.dtype != torch.uint8 or mask_y.dtype != torch.int32 or offsets.dtype != torch.int64 or dst.dtype != torch.uint8
                or offsets.numel() != F or dst.numel() != F * 270 * 480 * 3 or not dst.is_contiguous() or not packed.is_contiguous()):
            raise ValueError("mask_resize_packed: packed uint8, offsets int64 (F), mask_y int32 (F), dst uint8 (F,270,480,3)")
        self._ck(self.lib.jg_mask_resize_packed(self.h, _ptr(packed), packed.numel(), _ptr(offsets), F, int(H), int(W), _ptr(mask_y), _ptr(dst)))
        return dst

    def logmel(self, wav, mel_basis):
        """wav (B,n) fp32 (int16 scale) -> log-mel (B, n//160, 80)."""
        self._bind_stream()
        wav = self._f32(wav)
        mb = self._f32(mel_basis)
        B, n = wav.shape
        out = torch.empty((B, n // 160, 80), dtype=torch.float32, device=self.device)
        self._ck(self.lib.jg_logmel(self.h, _ptr(wav), B, n, _ptr(mb), _ptr(out)))
        return out

    def jegal_text(self, states, mask=None):
        self._bind_stream()
        states = self._f32(states)
        B, L, D = states.shape
        if D != 768:
            raise ValueError("text states must have 768 channels")
        m = None if mask is None else self._f32(mask).reshape(B, L)
        out = torch.empty((B, L, 256), dtype=torch.float32, device=self.device)
        self._ck(self.lib.jg_jegal_text(self.h, _ptr(states), _ptr(m), B, L, _ptr(out)))
        return out

    def xlmr_encode(self, input_ids, attention_mask=None):
        """XLM-RoBERTa last_hidden_state: input_ids / attention_mask (B,L) int -> (B,L,768) fp32 (jg_xlmr_encode)."""
        self._bind_stream()
        ids = self._i32(input_ids)
        if ids.ndim != 2:
            raise ValueError("input_ids must be (B, L)")
        B, L = ids.shape
        m = None
        if attention_mask is not None:
            m = self._i32(attention_mask)
            if tuple(m.shape) != (B, L):
                raise ValueError("attention_mask must have the shape of input_ids")
        out = torch.empty((B, L, 768), dtype=torch.float32, device=self.device)
        self._ck(self.lib.jg_xlmr_encode(self.h, _ptr(ids), _ptr(m), B, L, _ptr(out)))
        return out

    def calibrate_xlmr(self, input_ids=None, attention_mask=None):
        """Precision mode 3: switch the XLM-RoBERTa Linears from hi+lo to bias-corrected single fp16, calibrated on these token ids
        ((B,L) ints; None = built-in uniform-random ids, validated on seeded test weights only)."""
        self._bind_stream()
        if input_ids is None:
            self._ck(self.lib.jg_calibrate_xlmr(self.h, None, None, 0, 0))
            return
        ids = self._i32(input_ids)
        if ids.ndim != 2:
            raise ValueError("input_ids must be (B, L)")
        m = None if attention_mask is None else self._i32(attention_mask)
        if m is not None and tuple(m.shape) != tuple(ids.shape):
            raise ValueError("attention_mask must have the shape of input_ids")
        self._ck(self.lib.jg_calibrate_xlmr(self.h, _ptr(ids), _ptr(m), ids.shape[0], ids.shape[1]))

    def word_pool(self, seq, segments, dst, dst_col):
        """seq (rows,D) fp32; segments int (n,3) = (start,end_excl,dst_row); dst (rows_out, ld) fp32."""
        self._bind_stream()
        seg = self._i32(segments).reshape(-1, 3)
        if seg.shape[0] == 0:
            return
        self._ck(self.lib.jg_word_pool(self.h, _ptr(seq), seq.shape[-1], _ptr(seg), seg.shape[0], _ptr(dst), dst.shape[-1], dst_col))

    def fuse_content(self, fused):
        self._bind_stream()
        fused = self._f32(fused)
        rows = fused.numel() // 512
        out = torch.empty_like(fused)
        self._ck(self.lib.jg_fuse_content(self.h, _ptr(fused), rows, _ptr(out)))
        return out

    def l2norm(self, x):
        self._bind_stream()
        x = self._f32(x)
        out = torch.empty_like(x)
        D = x.shape[-1]
        self._ck(self.lib.jg_l2norm(self.h, _ptr(x), _ptr(out), x.numel() // D, D))
        return out

    # ---- metrics
    def pool_mean(self, x, offsets):
        self._bind_stream()
        x = self._f32(x)
        off = self._i32(offsets)
        n = off.numel() - 1
        out = torch.empty((n, x.shape[-1]), dtype=torch.float32, device=self.device)
        self._ck(self.lib.jg_pool_mean(self.h, _ptr(x), _ptr(off), n, x.shape[-1], _ptr(out)))
        return out

    def sim_rank(self, e1, e2, row_offset=0):
        self._bind_stream()
        e1, e2 = self._f32(e1), self._f32(e2)
        n_local, D = e1.shape
        rank = torch.empty(n_local, dtype=torch.int32, device=self.device)
        ties = torch.empty(n_local, dtype=torch.int32, device=self.device)
        self._ck(self.lib.jg_sim_rank(self.h, _ptr(e1), _ptr(e2), n_local, e2.shape[0], row_offset, D, _ptr(rank), _ptr(ties)))
        return rank, ties

    def spot(self, gesture, content, g_offsets, c_offsets, targets, temp=0.07):
        self._bind_stream()
        g, c = self._f32(gesture), self._f32(content)
        goh, coh, tgh = (np.asarray(a, np.int64) for a in (g_offsets, c_offsets, targets))
        n = tgh.size
        if goh.size != n + 1 or coh.size != n + 1:
            raise ValueError("g_offsets / c_offsets need one entry more than targets")
        T, W = np.diff(goh), np.diff(coh)
        if n and (T.min() <= 0 or T.max() > 8192 or W.min() <= 0 or W.max() > 1024 or (tgh < 0).any() or (tgh >= W).any()):
            raise ValueError("jg_spot limits: 1..8192 frames and 1..1024 words per clip, 0 <= target < words")
        go, co, tg = self._i32(g_offsets), self._i32(c_offsets), self._i32(targets)
        pred = torch.empty(n, dtype=torch.int32, device=self.device)
        score = torch.empty(n, dtype=torch.float32, device=self.device)
        self._ck(self.lib.jg_spot(self.h, _ptr(g), _ptr(c), _ptr(go), _ptr(co), _ptr(tg), n, g.shape[-1], temp, _ptr(pred), _ptr(score)))
        return pred, score

    def asd(self, query, cand, c_offsets, temp=0.07):
        self._bind_stream()
        q, c = self._f32(query), self._f32(cand)
        co = self._i32(c_offsets)
        n = q.shape[0]
        pred = torch.empty((n, 3), dtype=torch.int32, device=self.device)
        self._ck(self.lib.jg_asd(self.h, _ptr(q), _ptr(c), _ptr(co), n, q.shape[-1], temp, _ptr(pred)))
        return pred

    # ---- profiling
    # ---- multi-GPU exchange on RCCL through the C ABI (jegal_amd/dist.py uses torch.distributed for the same exchange)
    @staticmethod
    def comm_unique_id():
        """rank 0: a 128-byte RCCL id for the launcher to hand to every rank."""
        buf = ctypes.create_string_buffer(128)
        rc = load_library().jg_comm_get_unique_id(buf)
        if rc != 0:
            raise JegalError(f"jg_comm_get_unique_id failed with {rc} (librccl.so not loadable?)")
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        with torch.cuda.device(self.device):
            self._ck(self.lib.jg_comm_init(self.h, bytes(unique_id), int(rank), int(world)))
        self.comm_world = int(world)

    def comm_destroy(self):
        self._ck(self.lib.jg_comm_destroy(self.h))
        self.comm_world = 1

    def allgather(self, x):
        """(n, ...) per rank -> (world * n, ...) in rank order (ncclAllGather on the engine's stream)."""
        self._bind_stream()
        x = x.to(self.device).contiguous()
        out = torch.empty((getattr(self, "comm_world", 1) * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=self.device)
        self._ck(self.lib.jg_allgather(self.h, _ptr(x), _ptr(out), x.numel() * x.element_size()))
        return out

    def allreduce_sum_i64(self, counts):
        self._bind_stream()
        t = torch.as_tensor(counts, dtype=torch.int64).to(self.device).contiguous().clone()
        self._ck(self.lib.jg_allreduce_sum_i64(self.h, _ptr(t), t.numel()))
        return t

    def profile(self, on, only=None):
        """on: bracket every launch with HIP events; only="conv1": bracket just that stage (the rest of the step runs back to back)."""
        self._ck(self.lib.jg_profile_enable(self.h, 2 + STAGES.index(only) if (on and only) else int(bool(on))))

    def profile_reset(self):
        self._ck(self.lib.jg_profile_reset(self.h))

    def profile_get(self):
        res = {}
        for i, name in enumerate(STAGES):
            ms, n = ctypes.c_double(), ctypes.c_int64()
            self._ck(self.lib.jg_profile_get(self.h, i, ctypes.byref(ms), ctypes.byref(n)))
            res[name] = (ms.value, n.value)
        return res

    def workspace_bytes(self):
        return int(self.lib.jg_workspace_bytes(self.h))
