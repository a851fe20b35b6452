/* libjegal_hip -- C ABI of the MI355X-native JEGAL embedding-extraction engine.
 *
 * The reference (Sindhu-Hegde/jegal) has no FFI layer: its boundary for this path is the Python
 * method surface of two nn.Modules plus two on-disk formats.  Each entry
 * point below names the reference interface it replaces (file:line under the reference tree).
 * The Python facade in jegal_amd/ binds these with ctypes and re-creates the reference signatures
 * (GestSync.forward_vid, JEGAL.forward_inference, ...); INTEGRATION.md shows the binding.
 *
 * Conventions: opaque handle per GPU/process (not thread-safe), caller-owned buffers, every
 * function returns 0 on success or a negative code with jg_last_error() describing it, no
 * exceptions cross the boundary.  All data pointers are DEVICE pointers unless the parameter name
 * ends in _host.  Work is enqueued on the handle's HIP stream (jg_set_stream) and is asynchronous
 * unless stated; jg_sync waits for it.
 */
#ifndef JEGAL_HIP_H
#define JEGAL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jg_handle jg_handle;

enum { JG_OK = 0, JG_ERR_ARG = -1, JG_ERR_HIP = -2, JG_ERR_STATE = -3, JG_ERR_WEIGHT = -4 };

/* dtype codes for jg_load_tensor / frame buffers */
enum { JG_F32 = 0, JG_F16 = 1, JG_I64 = 2, JG_U8 = 3 };

/* operand precision (DESIGN.md "precision"): fp32 accumulate / residual / LN / softmax in all modes */
enum {
    JG_PREC_FP16 = 0,     /* every GEMM/conv operand fp16 */
    JG_PREC_FP16_W2 = 1,  /* Linear weights carried as hi+lo fp16 pair (2 MFMAs) */
    JG_PREC_FP16_W2_ALL = 2, /* conv weights split as well */
    JG_PREC_FP16_BC = 3,  /* single fp16 weights on the gesture path, the systematic part of the weight
                             rounding error (w - fp16(w)).E[x] folded into the bias by a calibration pass run inside
                             jg_finalize_weights (built-in clips: validated on the seeded weights only) or by jg_calibrate_gesture on
                             the caller's clips; content-path Linears keep the hi+lo split.  Opt-in: ~3 % faster than the default */
    JG_PREC_BF16 = 4,     /* REPORTED mode (north_star names bf16): every weight and every 16-bit activation is bf16 and every
                             MFMA is a bf16 MFMA (v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16, the second build of the kernels,
                             namespace bf).  Same MFMA rate as fp16, 8 instead of 11 significant bits: the embeddings come out
                             at ~5e-3 of the reference, outside the 1e-3 contract -- which is why fp16 is the default
                             (tests/test_gpu_precision_uploads.py::test_precision_modes_report prints the measured errors side by
                             side).  conv1 runs as an implicit GEMM over stacked frames and the LayerNorms as separate kernels in
                             this mode (the fused u8 conv1 kernel and the fp16 + fp8 token stream are fp16 constructs). */
    JG_PREC_FP16_RC = 5,  /* DEFAULT (round 5).  Run-time corrected: as JG_PREC_FP16_BC, but the term (w - fp16(w)).E[x] of every GestSync transformer
                             Linear is rebuilt per GEMM call and per clip from a fixed sample of THAT clip's own input rows (two small
                             launches in front of the GEMM, a per-clip bias in its epilogue) -- no calibration pass, nothing depends on
                             calibration data or on the other clips of a batch.  The JEGAL branch and the content path run hi+lo.
                             What the CLI drivers select for a checkpoint they have never seen (jegal_amd/drivers.py). */
    JG_PREC_FP32 = 6      /* AUDIT mode (round 6; SURVEY 8b's precision list names BF16X3 / FP32: this is that exact mode).  Every GEMM /
                             convolution on v_mfma_f32_32x32x2_f32 with fp32 weights, fp32 activations end to end, fp32 softmax and
                             LayerNorm: the on-device stand-in for the reference's CPU path, which is fp32 (inference_embs.py:497: autocast
                             does nothing without CUDA).  ~2e-6 of the fp32 oracle instead of ~6e-4, ~50 clips/s instead of ~2 600: what
                             `python -m jegal_amd.drivers ... --audit` compares the default mode with on the caller's own clips and
                             checkpoint, where no oracle exists.  Option "audit_weights" (before jg_finalize_weights) keeps the fp32
                             matrices next to the fp16 ones in any mode, option "audit_stages" then moves single stages to fp32. */
};

/* ---- lifecycle ------------------------------------------------------------------------------ */
int jg_create(int device, jg_handle** out);
int jg_destroy(jg_handle* h);
const char* jg_last_error(jg_handle* h);
/* stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = the legacy default
 * stream.  Until this is called the handle uses a private non-blocking stream. */
int jg_set_stream(jg_handle* h, void* hip_stream);
int jg_set_precision(jg_handle* h, int mode);
/* clips (or 25-frame windows / 8) of the GestSync conv stack processed per pass; bounds the workspace (default 32: ~14 GB per lane
 * for 150-frame clips - fewer, larger launches are what the 288 GB of the part are for; lower it for long clips on a shared GPU) */
int jg_set_chunk(jg_handle* h, int clips_per_chunk);
/* tuning / A-B switches, per handle (all default to the fast setting; results stay within the parity tolerance either way):
 *   "conv1_direct"    1: fused u8 conv1+pool kernel, 0: temporal stack + implicit GEMM + pool kernel
 *   "conv1_zero_skip" 1: all-zero input bands (the face-mask rows) are skipped / run only the bias slots (bit-identical)
 *   "conv1_mfma16"    1: the fused conv1 kernel's MFMA waves run v_mfma_f32_16x16x32_f16 (0: 32x32x16, the round-1/2 form; the two
 *                     differ in fp32 summation order only)
 *   "conv2_row_skip"  1: conv2 .. conv5 do not compute the leading output rows of a POSITION that the zero-band scan proves to be
 *                     independent of the position (their whole window lies in conv1's constant region); consumers read them from
 *                     images computed once per weight load.  Per position: every position skips what its own five frames allow
 *                     (bit-identical)
 *   "qkv0_linear"     1: the first transformer layer's qkv projection runs over the T+4 distinct conv positions and the 21
 *                     positional rows (linearity of W(conv + pe) + b); the attention kernel gathers and sums the rows
 *   "edge_dedup"      1: evaluate only the T+4 distinct padded-clip positions
 *   "fuse_ln"         1: residual + LayerNorm fused into the GestSync projection GEMMs (tiled token stream)
 *   "stream_fp16"     1: that token stream is the fp16 plane alone (post-norm: the LayerNorm output is rounded to fp16 as the next GEMM's
 *                     operand anyway; -1.5 % per step); 0: fp16 + one e4m3 byte of correction per element (rounds 2-4)
 *   "attn_mfma"       1: MFMA attention kernels for S <= 160, dk = 64
 *   "gemm_glds", "gemm_big_tile", "gemm_small_tile", "gemm_tall_tile", "gemm_persistent", "gemm_counted",
 *   "gemm_stagger":   tile / pipeline choices of the LDS-DMA GEMM (tests/test_gpu_options_scale_multirank.py flips every one of them)
 *   "gemm_tile"       0: plain GEMMs pick their tile by a measured cost estimate; 1 / 2 / 3: force 128x128 / 256x128 / 256x256 (bit-identical)
 *   "dual_stream"     1: jg_extract_gesture and jg_gestsync_clip split a batch of >= 8 clips (and >= 256 frames in the smaller part) into two halves (option "dual_split": eighths of the batch on the first lane, default 4 since round 6; rounds 3-5: 3) and run the two parts concurrently on two internal
 *                     streams (own workspaces; the caller's stream is joined at entry and exit): one part's next kernel fills
 *                     the partly empty last round of the other's persistent kernels.  Bit-identical results.
 *   "num_cu"          workgroups a persistent kernel launches (default: the device's CU count; experiment)
 *   "lane_priority"   3 (default): the two lane streams are created with the device's highest stream priority; 0: normal priority (rounds 3-5);
 *                     1 / 2: only the second / first lane (a change drains and re-creates the lane streams).  HIP deals streams onto four hardware
 *                     queues PER PRIORITY LEVEL in creation order: normal-priority lanes can end up on one queue when the application owns
 *                     other streams, and then the "concurrent" halves run in turn (measured: 2 076 instead of 2 510 clips/s with five other
 *                     streams).  High-priority lanes only compete with the application's own high-priority streams.  (A host-streaming pipeline around the engine -- GestureStreamer's
 *                     H2D / D2H / compute streams -- in a process that owns further streams is the one measured case where 0 was faster:
 *                     tools/experiments/stream_queue_sweep.sh)
 *   "gesture_lanes"   0 (default): two lanes ("dual_split"); 3 / 4: that many EQUAL lanes (experiment: slower, tools/experiments/README.md)
 *   "xlmr_lanes"      2 (default), 1 .. 4: jg_xlmr_encode runs a batch as that many equal parts on as many streams (the first two are the
 *                     lane streams of "dual_stream" / "lane_priority")
 *   "ws_poison"       1 (test aid, default 0): the workspace is filled with 0xff bytes (fp16/fp32 NaN) before every clip chunk, so a
 *                     kernel that reads a row nobody wrote (the row / band skips leave rows unwritten on purpose) shows up as NaN
 *   "gemm_timeline"   1: print a per-tile phase timeline of every GEMM launch to stderr (debug)
 *   "jegal_fp32_ends" 1 (default): the two ends of the JEGAL gesture branch (proj_ip_rgb; final norm + proj_op_rgb + proj_op_align_gesture) and
 *                     of the content path (proj_op_text, fusion / align MLPs) keep fp32 activations and run on the split-operand GEMM
 *                     (three fp16 MFMAs per tile: fp32-grade products); 0: the round-5 arithmetic (fp16 activations, hi+lo weights).  DESIGN.md section 3
 *   "jegal_ffn_x3"    0 (default) / 1: the six feed-forward sub-layers of the JEGAL gesture branch on the split-operand GEMM as well (gesture error
 *                     3.5e-4 -> 2.5e-4 on the Gaussian draw for +3 % step time)
 *   "conv_round_diffuse" 1 (default; before jg_finalize_weights): conv weights are rounded to fp16 with error diffusion across the taps of each
 *                     (output channel, input slot) pair instead of round-to-nearest per weight (the pixel-independent part of the rounding error vanishes)
 *   "rc_layers"       measurement only: mask of the GestSync Linear types that get JG_PREC_FP16_RC's run-time correction (1 qkv, 2 out_proj,
 *                     4 linear1 / ff_vid.0, 8 linear2; default 15); the others run single fp16 WITHOUT a correction
 *   "audit_jegal_parts" measurement only (needs audit_weights): parts of the fp16 JEGAL gesture branch on the fp32 kernels (1 input projection,
 *                     2 attention sub-layers, 4 feed-forward sub-layers, 8 final norm + output projections)
 *   "audit_weights"   1 (before jg_finalize_weights): the fp32 matrices are kept next to the packed fp16 ones (always in JG_PREC_FP32)
 *   "audit_stages"    mask of the stages that run on the fp32 audit kernels (needs audit_weights): 1 GestSync conv stack, 2 GestSync
 *                     transformer + ff_vid, 4 JEGAL gesture branch, 8 JEGAL content path (audio / text / fusion), 16 XLM-RoBERTa.  The
 *                     stage boundaries are fp32 tensors in every mode; this is how DESIGN.md section 3 decomposes the fp16 error by stage */
int jg_set_option(jg_handle* h, const char* name, int value);
int jg_sync(jg_handle* h);

/* ---- weights: replaces model.load_state_dict(sd) (inference_embs.py:92-119,
 *      evaluation/extract_jegal_embs.py:32-53).  `name` is the reference state_dict key with any
 *      "module." prefix already stripped; unknown keys (net_aud.*, lstm.*, ...) are accepted and
 *      ignored, missing hot-path keys make jg_finalize_weights fail (strict). ------------------- */
int jg_load_tensor(jg_handle* h, const char* name, const void* data_host, const int64_t* shape_host, int ndim, int dtype);
/* which: bit 0 = GestSync, bit 1 = JEGAL, bit 2 = XLM-RoBERTa (keys of transformers.XLMRobertaModel under the prefix "xlmr.").
 * Folds BatchNorm, packs k=(kh,kw,c), splits hi/lo.
 * Staged tensors: the ones this call consumed are dropped; tensors staged for a model that is finalized by a LATER call stay
 * (load everything, then finalize(1), finalize(2) works); keys no finalize consumes stay until jg_clear_staged_tensors.
 * JG_PREC_FP16_BC: the built-in calibration runs for the gesture models finalized by THIS call only (never for bit 2 alone);
 * bias corrections of the other model - e.g. from jg_calibrate_gesture on real clips - are kept.  Re-finalizing a model
 * discards ITS corrections: call jg_calibrate_gesture again after re-loading it. */
int jg_finalize_weights(jg_handle* h, int which);
/* Drop every staged host tensor (the unused net_aud / lstm tensors of a GestSync checkpoint, tensors of a model never finalized). */
int jg_clear_staged_tensors(jg_handle* h);

/* Re-run the JG_PREC_FP16_BC calibration on caller-supplied clips (same layout as jg_gestsync_clip; device
 * pointer) instead of the built-in synthetic ones, e.g. a few real videos.  frames == NULL: built-in clips. */
int jg_calibrate_gesture(jg_handle* h, const void* frames, int frames_dtype, int B, int T);

/* ---- XLM-RoBERTa text front end ------------------------------------------------
 * Replaces `mroberta(input_ids, attention_mask=text_mask).last_hidden_state` of JEGAL.get_roberta_embeddings
 * (models/jegal.py:116-129; the reference runs transformers.XLMRobertaModel "xlm-roberta-base" on the CPU).  The tokenizer stays
 * on the host.  input_ids, attention_mask: (B, L) int32 on the device (attention_mask NULL = all ones); out (B, L, 768) fp32.
 * Third-party arithmetic: parity is pinned against transformers.XLMRobertaModel with seeded random weights
 * (tests/golden/xlmr.npz), not against the released checkpoint, which is not available offline. */
int jg_xlmr_encode(jg_handle* h, const int32_t* input_ids, const int32_t* attention_mask, int B, int L, float* out);
/* JG_PREC_FP16_BC and JG_PREC_FP16_RC (the default).  After jg_finalize_weights(h, 4) the XLM-RoBERTa Linears run with hi+lo fp16 weight pairs (calibration-free,
 * two MFMAs per fragment pair).  This call runs one pass over the CALLER's token ids (device (B,L) int32; attention_mask may be
 * NULL), records the input mean of every Linear and switches them to single fp16 weights with the systematic rounding term
 * (w - fp16(w)).E[x] folded into the bias (~1.5x faster encoder).  input_ids == NULL: built-in uniform-random ids -- validated on
 * seeded test weights only, NOT on the released xlm-roberta-base checkpoint (which is not available offline), hence never implicit. */
int jg_calibrate_xlmr(jg_handle* h, const int32_t* input_ids, const int32_t* attention_mask, int B, int L);

/* ---- GestSync (models/gestsync.py) ---------------------------------------------------------- */
/* Per-clip features: frames (B,T,270,480,3) u8 (JG_U8, the /255 of inference_embs.py:282 is applied
 * inside) or fp32 in [0,1] (JG_F32) -> edge-pad 12 (inference_embs.py:283) -> T windows of 25
 * (inference_embs.py:488-492) -> forward_vid -> mean(-1) (inference_embs.py:511) -> (B,T,1024) fp32.
 * The conv stack runs once over the padded clip (window de-duplication, exact). */
int jg_gestsync_clip(jg_handle* h, const void* frames, int frames_dtype, int B, int T, float* out_feats);
/* The same for a batch of clips of DIFFERENT lengths padded to T with copies of each clip's last frame (exact for the clip's own
 * frames: inference_embs.py:283 edge-pads with the last frame and window t only reaches frame t + 12; this is how
 * jegal_amd.drivers extract_gestsync_feats batches preprocess/extract_gestsync_feats.py:314-344, which runs one video at a time).
 * valid_frames_host (host [B], 1..T): frames of clip b that are its own.  JG_PREC_FP16_RC takes each clip's run-time correction from
 * its own rows only, so rows t < valid_frames[b] of clip b do not depend on T or on the other clips, and for clips of >= 49 frames
 * they are bit-identical to the clip run alone.  (A clip of fewer than 49 frames ALONE has fewer than 1 024 token rows and takes the
 * unfused hi+lo plan, in a padded batch the fused one: each within the 1e-3 contract of the reference, 5e-4 apart in the test -- not bit for bit.)  The
 * other modes ignore the lengths.  Rows t >= valid_frames[b] of the output are padding for the caller to strip. */
int jg_gestsync_clip_ragged(jg_handle* h, const void* frames, int frames_dtype, int B, int T, const int32_t* valid_frames_host, float* out_feats);
/* Kernel-level check point: conv1+BN+ReLU+maxpool (gestsync.py:36-46) only.  frames (B,T,270,480,3) u8,
 * pad = temporal edge padding (12 for clips, 0 for a raw 25-frame window) -> out (B*(T+2*pad-4),43,78,64) fp16 NHWC. */
int jg_debug_conv1_pool(jg_handle* h, const void* frames_u8, int B, int T, int pad, void* out_f16);
/* Check point for "conv2_row_skip": the MINIMUM over the positions of the last conv stack of the leading conv2 output rows that
 * were read from the const chain instead of computed (every position skips its own count: jg_debug_conv_rows);
 * 0: none, or the option is off.  Synchronises the stream. */
int jg_debug_conv2_rowskip(jg_handle* h, int* rows);
/* Check point for the per-position form of it: computed[l] / full[l] = output pixels (rows of the implicit GEMM) that conv2 .. conv5
 * (l = 0..3) of the LAST conv stack computed / would compute without the skip (0 / 0: option off or path not taken).  Synchronises. */
int jg_debug_conv_rows(jg_handle* h, int64_t* computed, int64_t* full);
/* Tuning aid: ms per launch of the production GEMM for a shape (mode bit0 hi+lo weights, bit1 fp32 residual in/out, bit2 ReLU). */
int jg_debug_gemm(jg_handle* h, int M, int N, int K, int mode, int iters, double* ms);
/* The same with caller-supplied fp16 operands a16 [M][K] / w16 [N][K] (device pointers; NULL: constant fill).  Constant operands
 * flatter any MFMA kernel (the chip holds a higher clock on them): tools/gemm_yardstick.py times random data on both sides. */
int jg_debug_gemm_ex(jg_handle* h, const void* a16, const void* w16, int M, int N, int K, int mode, int iters, double* ms);
/* Drop-in for GestSync.forward_vid(x, return_feats) (gestsync.py:148-162): x (N,3,25,270,480) fp32
 * -> out (N,1024,21) fp32, optional out_conv (N,512,21) fp32 (NULL to skip). */
int jg_gestsync_windows(jg_handle* h, const float* x, int N, float* out, float* out_conv);

/* Face-mask + resize pre-step, load_rgb_masked_frames (inference_embs.py:235-276) without the /255 and the edge pad
 * (those live in jg_gestsync_clip): src (T,H,W,3) uint8 device frames, mask_y (T) int32 DEVICE array -- per frame the
 * last source row blanked by cv2.rectangle(img,(0,0),(W,y2+15),0,-1) (face found: mask, then resize), or -1 when
 * mediapipe found no face (resize, then rows 0..110 of the result blanked) -> dst (T,270,480,3) uint8, ready for
 * jg_gestsync_clip.  cv2.resize(INTER_LINEAR, 8-bit) restated from OpenCV's generic fixed-point path; parity unpinned
 * (cv2 absent).  The keypoints themselves stay on the host (mediapipe). */
int jg_mask_resize(jg_handle* h, const uint8_t* src, int T, int H, int W, const int32_t* mask_y, uint8_t* dst);

/* Masked crops in fewer bytes (the host link, not the GPU, bounds a streamed extraction: DESIGN.md section 7): the reference blanks
 * rows 0..y2+15 of every crop (inference_embs.py:264-270), so a producer ships only the rows BELOW each frame's mask.
 * packed: those rows of all frames back to back (device), packed_bytes its size; row0 (n_frames) int32 device: first kept row of
 * each frame (0..270); offsets (n_frames) int64 device: byte offset of that row in `packed` (multiples of 16) -> dst
 * (n_frames,270,480,3) uint8 with the rows above row0 zero: exactly the crop load_rgb_masked_frames returns, ready for
 * jg_gestsync_clip / jg_extract_gesture.  Metadata is validated ON THE DEVICE (it lives there): a frame whose row0 is outside
 * 0..270 or whose offset is negative, not a multiple of 16 or runs past packed_bytes comes out all zero, nothing is read. */
int jg_unpack_masked(jg_handle* h, const uint8_t* packed, int64_t packed_bytes, const int32_t* row0, const int64_t* offsets, int n_frames, uint8_t* dst);
/* The same at SOURCE resolution: the reference decodes e.g. 228x314 / 294x294 crops and resizes them to 270x480 on the host
 * (inference_embs.py:255-276) -- shipping the decoder's frames moves up to 1.8x fewer bytes than shipping the resized crops, and the
 * rows the mask blanks (source rows 0..mask_y) need not cross the link at all.  packed: per frame f the source rows
 * max(mask_y[f]+1, 0) .. H-1 (mask_y = -1, no face: the whole frame), (H - row0) * W * 3 bytes starting at offsets[f];
 * -> dst (T,270,480,3) uint8 = jg_mask_resize of the full frames.  A frame whose rows would run past packed_bytes comes out zero. */
int jg_mask_resize_packed(jg_handle* h, const uint8_t* packed, int64_t packed_bytes, const int64_t* offsets, int T, int H, int W,
                          const int32_t* mask_y, uint8_t* dst);

/* ---- JEGAL (models/jegal.py) ---------------------------------------------------------------- */
/* forward_gestures (jegal.py:78-92) [+ proj_op_align_gesture, jegal.py:381 when align != 0]:
 * feats (B,T,1024) fp32, mask (B,T) fp32 (1 valid / 0 pad) or NULL -> out (B,T,512) fp32. */
int jg_jegal_gestures(jg_handle* h, const float* feats, const float* mask, int B, int T, int align, float* out);
/* forward_audio (jegal.py:105-113): mel (B,Tm,80) fp32 -> out (B,Ta,256) fp32, Ta = jg_audio_len(Tm). */
int jg_jegal_audio(jg_handle* h, const float* mel, int B, int Tm, float* out);
int jg_audio_len(int Tm);
/* forward_audio on a zero-padded batch of clips of DIFFERENT lengths with the result each clip would give alone: the reference's
 * dataset driver runs batch_size = 1 (evaluation/extract_jegal_embs.py:141), and the conv stack's zero padding (jegal.py:41-63)
 * makes the last audio steps of a clip depend on what follows it in a padded batch.  valid_tm_host (B) int32 on the HOST: mel
 * frames clip b really holds (4..Tm; rows beyond must be present in `mel` but are never read as data).  Rows t < jg_audio_len(
 * valid_tm_host[b]) of out[b] equal jg_jegal_audio on the clip alone up to fp32 summation order; rows beyond are unspecified.
 * NULL = jg_jegal_audio. */
int jg_jegal_audio_ragged(jg_handle* h, const float* mel, int B, int Tm, const int32_t* valid_tm_host, float* out);
/* wav2filterbanks (utils/audio_utils.py:28-66): wav (B,n_samples) fp32 (int16 scale, NOT normalised: audio_utils.py:20-25),
 * mel_basis (80,257) fp32 = librosa.filters.mel(sr=16000,n_fft=512,n_mels=80,fmin=0,fmax=8000) -> out (B, n_samples/160, 80) log-mel. */
int jg_logmel(jg_handle* h, const float* wav, int B, int n_samples, const float* mel_basis, float* out);
/* forward_text (jegal.py:95-103): states (B,L,768) fp32 (XLM-R last_hidden_state), mask (B,L) -> (B,L,256). */
int jg_jegal_text(jg_handle* h, const float* states, const float* mask, int B, int L, float* out);
/* word pooling (jegal.py:174-180,189-195,233-239): for each int32 triplet (start_row,end_row_excl,dst_row)
 * dst[dst_row][dst_col : dst_col+D] = mean(seq[start:end]).  seg is a DEVICE pointer. */
int jg_word_pool(jg_handle* h, const float* seq, int D, const int32_t* seg, int n_seg, float* dst, int dst_ld, int dst_col);
/* cat((audio,text),-1) -> proj_op_fusion_content -> proj_op_align_content (jegal.py:406-415):
 * fused (rows,512) fp32 (audio cols 0..255, text cols 256..511, zero-padded rows) -> out (rows,512). */
int jg_fuse_content(jg_handle* h, const float* fused, int rows, float* out);
/* F.normalize(p=2,dim=-1) (inference_embs.py:631,635; extract_jegal_embs.py:111,115); in == out allowed */
int jg_l2norm(jg_handle* h, const float* in, float* out, int rows, int D);
/* frames -> unit-norm gesture embedding (B,T,512) without leaving the device (the v-only path of
 * inference_embs.py:526-646): jg_gestsync_clip + jg_jegal_gestures(align=1) + jg_l2norm. */
int jg_extract_gesture(jg_handle* h, const void* frames, int frames_dtype, int B, int T, float* out_emb);

/* ---- metrics (evaluation/evaluate_*.py) ----------------------------------------------------- */
/* temporal mean of ragged blocks (evaluate_retrieval.py:30-31): out[i] = mean(x[off[i]:off[i+1]]) */
int jg_pool_mean(jg_handle* h, const float* x, const int32_t* offsets, int n, int D, float* out);
/* evaluate_retrieval.py:38-65 on already-normalised rows: rank/ties of the diagonal per local row */
int jg_sim_rank(jg_handle* h, const float* e1, const float* e2, int n_local, int n_total, int row_offset, int D,
                int32_t* rank, int32_t* ties);
/* evaluate_spotting.py:39-82: per clip first-argmax frame and its softmax score for word `target`.
 * Limits per clip: <= 1024 words, <= 8192 frames, 0 <= target < words.  The offsets are device arrays (no host sync to
 * validate them): a clip outside the limits gets pred = -1 and score = NaN instead of a result. */
int jg_spot(jg_handle* h, const float* gesture, const float* content, const int32_t* g_offsets, const int32_t* c_offsets,
            const int32_t* target, int n_clips, int D, float temp, int32_t* pred, float* score);
/* evaluate_asd.py:43-51,94-100: pred (n,3) = argmax over the first 2/4/6 candidates */
int jg_asd(jg_handle* h, const float* query, const float* cand, const int32_t* c_offsets, int n, int D, float temp, int32_t* pred);

/* ---- multi-GPU exchange.  The reference is single-process (its only parallelism is the --rank / --nshard file-list split of
 *      preprocess/extract_gestsync_feats.py:366-370); clips shard with no data-path collective, and the ONE exchange of the path is the
 *      gallery all-gather in front of the retrieval similarity matrix (+ a counter all-reduce for R@K / spotting / ASD).  These entries give a
 *      consumer of the C ABI that exchange on RCCL over xGMI without PyTorch: one communicator per handle (= per rank = per GPU), collectives
 *      enqueued on the handle's stream.  librccl.so is bound with dlopen at the first call (no link-time dependency).  jegal_amd/dist.py does
 *      the same through torch.distributed (backend "nccl" = RCCL). ---------------------------------------------------------------------- */
#define JG_COMM_ID_BYTES 128
/* rank 0: ncclGetUniqueId into a 128-byte host buffer, to be handed to every rank by the launcher (a file, MPI, a socket, torchrun's store) */
int jg_comm_get_unique_id(char* id128_host);
/* collective over all ranks: ncclCommInitRank on the handle's device */
int jg_comm_init(jg_handle* h, const char* id128_host, int rank, int world);
int jg_comm_destroy(jg_handle* h);
/* recv (world * bytes_per_rank bytes, device) = the ranks' send buffers (bytes_per_rank bytes each, device) in rank order */
int jg_allgather(jg_handle* h, const void* send, void* recv, int64_t bytes_per_rank);
/* in place sum over the ranks of n int64 counters (device) */
int jg_allreduce_sum_i64(jg_handle* h, int64_t* buf, int n);

/* ---- profiling: HIP-event timing per stage on the handle's stream -------------------------- */
enum { JG_ST_STACK = 0, JG_ST_CONV1, JG_ST_POOL, JG_ST_CONV, JG_ST_GEMM, JG_ST_ATTN, JG_ST_NORM, JG_ST_MISC, JG_ST_CONV1_AUX, JG_ST_COUNT };
/* on: 0 = off, 1 = every launch is bracketed by two events, 2 + stage = only the launches of that stage are (the other
 * launches of the step then run back to back, as in an unprofiled step) */
int jg_profile_enable(jg_handle* h, int on);
/* synchronises, then returns accumulated milliseconds and launch count of a stage since the last reset */
int jg_profile_get(jg_handle* h, int stage, double* ms, int64_t* launches);
int jg_profile_reset(jg_handle* h);
const char* jg_stage_name(int stage);
/* bytes currently held by the workspace arena */
int64_t jg_workspace_bytes(jg_handle* h);

#ifdef __cplusplus
}
#endif
#endif
